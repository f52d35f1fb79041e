// rbm_kernels.hip -- ClassificationRBM hot path on gfx950 (MI355X): block-Gibbs conditionals,
// free-energy evaluation, two-phase (CD-k / discriminative) gradient, parameter update.
// Reference semantics: src/ClassificationRBM.py (file:line cited per kernel), SURVEY.md 8.A.
#include <atomic>
#include <cstdlib>
#include "strip_mma.hip.h"
#include "qbm_host.hpp"

using namespace qbm;

namespace qbm {
static thread_local char g_err[256] = "";

int fail(int code, const char* msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int hip_fail(hipError_t e, const char* where)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return QBM_ERR_HIP;
}
}  // namespace qbm

namespace {

// Strip shapes (rows = 16*MT, columns = 16*NT), chosen so that one launch of the benchmark shape fills the
// 256 CUs in a single round with one workgroup per CU (two workgroups on a CU share its MFMA pipes and its
// L2 -> LDS path, which costs more than the larger strip):
//   hidden  : [256 x 512], K = 784 -> 32 x 16 strips: 8 x 32 = 256 workgroups
//   visible : [256 x 784], K = 512 -> 64 x 16 strips: 4 x 49 = 196 workgroups (+ 32 class workgroups)
//   gradient: [784 x 512], K = 256 x 2 phases -> 64 x 32 strips: 13 x 16 = 208 workgroups (+ 30 for gU/gd/gb/gc)
#ifndef QBM_H_MT   // diagnostic variants (tools/stamp_kernels.py --variant) may override the hidden strip shape
#define QBM_H_MT 2
#define QBM_H_NT 1
#endif
constexpr int H_MT = QBM_H_MT, H_NT = QBM_H_NT;
#ifndef QBM_V_MT
#define QBM_V_MT 4
#endif
constexpr int V_MT = QBM_V_MT, V_NT = 1;
// Gradient strips: 64 rows x 32 hidden columns (NT = 2) for the two-phase form at the benchmark shape; 64 x 16 (NT = 1) for the
// one-phase form of the discriminative step (its fp32 operand then needs 80 KB instead of 96 KB of LDS: TWO workgroups per
// CU, and the kind D workgroups run beside the strips: 784 x 512, B = 256, C = 10: 32.4 -> 26.8 us per step) and wherever the
// wider strips leave half of the CUs without a workgroup (784 x 128: 22.8 -> 21.7 us per CD-1 step).  Measured the other way
// round too: 64 x 16 strips for the two-phase form at 784 x 512 cost 9.6 instead of 8.5 us per launch (every data tile is
// staged twice as often), B = 2048: 135 instead of 126 us per step.
constexpr int G_MT = 4;
// Round 4 built 112 x 16 strips (MT = 7, NT = 1) for the two-phase form where they tile V exactly and fill more CUs than the
// 64 x 32 ones: 784 = 7 x 112, so 784 x 512 is 7 x 32 = 224 full workgroups -- the 64 x 32 strips cut 784 rows into 12 1/4:
// 13 x 16 = 208 workgroups on 256 CUs, the 13th row strip a quarter full (94 % useful MFMAs); per workgroup 7/8 of the MFMA work
// and 88 KB instead of 96 KB staged.  The seven row tiles of a strip share one merged uint8 image (seven 16-byte slots per k:
// 112-byte rows, conflict-free for ds_read_u8), filled by seven 1-KiB transfers per 64 k.  Bit-identical (every GPU parity test)
// -- and SLOWER: k_grad 8.68-8.74 us against 8.39-8.41 with the 64 x 32 strips in three alternating A/B runs of one session
// (profiles/r04_ab_strips_chains.jsonl; CD-1 step 28.13-28.18 against 27.86-27.90 us).  A wavefront of the tall strip stages
// 7 KB of data + 4 KB of probabilities behind 11 transfer instructions for 112 MFMAs, the wide one 4 + 8 KB behind 12 for 128:
// what the launch waits for is the staging of a cold start, not the MFMAs or the 48 idle CUs.  Opt-in for diagnostic builds
// (-DQBM_GRAD_112=1).
constexpr int G_MT7 = 7;
#ifndef QBM_GRAD_112
#define QBM_GRAD_112 0
#endif

// class-count buckets for the kernels that keep one register (or one float4) per class
#define QBM_DISPATCH_CM(C, MACRO)                   \
    do {                                            \
        if ((C) <= 2) { MACRO(2); }                 \
        else if ((C) <= 4) { MACRO(4); }            \
        else if ((C) <= 10) { MACRO(10); }          \
        else { MACRO(QBM_MAX_CLASSES); }            \
    } while (0)

// The status word and the device-side draw counters (HiddenArgs / VisibleArgs::status, ::draw_base) are read by a COMPILE-TIME
// variant of the sampling kernels only (XA, "extra arguments": launched when either pointer is set -- the opt-in fused tail, a
// captured generative epoch).  Read at run time behind a null test the counters cost k_visible 0.7 us per launch, read
// unconditionally from a zero pair still 0.2 us, the status word a few hundredths: the default path carries none of it.  Inside
// the variant a null pointer is replaced by the address of zeros (a select, no branch).
static __device__ unsigned long long g_zero_base[2] = {0ull, 0ull};
template <class T> struct KView;   // K-contiguous A operand of element type T
template <> struct KView<float> { using type = ViewKContig; };
template <> struct KView<uint8_t> { using type = ViewKContigU8; };
template <class T> struct CView;   // M-contiguous A operand of element type T
template <> struct CView<float> { using type = ViewCContig; };
template <> struct CView<uint8_t> { using type = ViewCContigU8; };

// SC1: an L1-bypassing load (the state was written by another workgroup of the XCD during this launch)
template <class T, bool SC1 = false>
__device__ __forceinline__ float ld_state(const void* p, long i)
{
    if constexpr (SC1) return (float)__hip_atomic_load(reinterpret_cast<const T*>(p) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return (float)reinterpret_cast<const T*>(p)[i];
}

// ---------------------------------------------------------------------------------------------
// A-1  sample_hidden  (src/ClassificationRBM.py:43-47):  p = sigma(v.W + d + y.U);  h = (p > u)
// MODE 0: probabilities (+ optional sample);  MODE 1: pre-activation v.W + d only (:65, :106).
// TA: element type of v (float, or uint8 for {0,1} Gibbs states).
// ---------------------------------------------------------------------------------------------
struct HiddenArgs {
    const void* v;   // [B,V] TA
    const float* y;  // [B,C] or null
    const float* W;  // [V,H]
    const float* d;  // [H]
    const float* U;  // [C,H]
    float* p;        // [B,H] or null
    float* s;        // [B,H] or null   sample as fp32 {0,1}
    uint8_t* s8;     // [B,H] or null   sample as uint8 {0,1}
    int B, V, H, C;
    int KP;
    uint64_t seed, offset;
    uint32_t row0;
    bool epiDma;   // the epilogue operands (bias tile, U tile, y rows) of a wavefront come by LDS-DMA transfers
    int ldsCls;    // byte offset of the per-wavefront scratch of those transfers (2 KiB each) in the dynamic LDS
    float* pre;    // [B,H] or null   MODE 0 only: v.W + d WITHOUT the class term (what MODE 1 delivers), for the training
                   // loop's p(y|x) monitor (qbm_rbm_cd_workspace.mon_*): one more store instead of one more contraction
    const uint32_t* status;   // non-null: status word of the step's workspace (sync[512]); non-zero = an earlier fused
                              // launch on it gave up and the host has not looked yet: this launch stores nothing (the
                              // workspace keeps what the host needs to finish that step: k_hgrad)
    const uint64_t* draw_base;   // non-null: [2] device words added to (offset, row0) -- qbm_rbm_cd_workspace.draw_base: the
                                 // launches of an epoch captured in a hipGraph keep their arguments, the counters move on
    int pair;        // k_hidden's grid: column strips 2j, 2j + 1 on XCD j % 8 (H a multiple of 256) instead of strip c on XCD c % 8
};

// 8 wavefronts (512 threads): wavefront q contracts chain q of the 8 the K = V terms are cut into ([Q] with NQ = 8), two
// wavefronts per SIMD; one output column per thread in the epilogue.
constexpr int H_NQ = 8;
// FASTK: 0 = operands through registers (any shape), 1 = LDS-DMA operands, 2 = LDS-DMA operands and epilogue operands
// (a compile-time choice: the hooks below must not hold both variants, or the compiler's merge of the two waits for the
// first operand transfers inside the hook)
// AUX: cache policy of the loads of v and y (kSc1 inside the fused chain kernel: another workgroup of the XCD wrote them
// during this launch; FASTK == 2 only).  The body of strip (ms, ns); the caller is a 512-thread workgroup.
// post: strip_mma's hook between the contraction and the combine (k_hgrad requests the gradient's operands there)
template <class TA, int FASTK, int MODE, int CM, int AUX = 0, class Post = NoPre, bool XA = false>
__device__ __forceinline__ void hidden_body(const HiddenArgs& a, int ms, int ns, char* lds, char* ldsB = nullptr,
                                            bool b_resident = false, Post&& post = NoPre{})
{
    constexpr bool FAST = FASTK >= 1;
    static_assert(AUX == 0 || FASTK == 2, "cache policies apply to the all-DMA variant");
    constexpr int MT = H_MT, NT = H_NT, NQ = H_NQ;
    const int m0 = ms * 16 * MT, n0 = ns * 16 * NT;
    const typename KView<TA>::type A{reinterpret_cast<const TA*>(a.v), a.V, a.B};
    const ViewCContig Bv{a.W, a.H, a.H, a.V};
    // Epilogue map: the 32 x 16 strip has 512 outputs: one per thread.
    using SO = StripOut<MT, NT, 1, 64 * NQ>;
    static_assert(SO::NP == 1, "one output per thread");
    const int row = m0 + SO::row(0), j0 = n0 + SO::cg(0);
    const int i_c = min(row, a.B - 1), j_c = min(j0, a.H - 1);
    const bool with_cls = MODE == 0 && a.y != nullptr;
    uint32_t st_w = 0u;      // (XA: scalar loads issued here; the status word is consumed in front of the stores, the draw
    uint64_t off_b = 0ull;   //  counters in `pre`, behind the first transfers)
    uint32_t row_b = 0u;
    if constexpr (XA) {
        const unsigned long long* db = a.draw_base ? reinterpret_cast<const unsigned long long*>(a.draw_base) : g_zero_base;
        const uint32_t* sp = a.status ? a.status : reinterpret_cast<const uint32_t*>(g_zero_base);
        st_w = *sp; off_b = db[0]; row_b = (uint32_t)db[1];
    }
    // Everything of the epilogue that does not depend on the contraction -- bias and class-term operands, the class chain
    // (classes c >= C enter it as 0 * 0, which is exact), the Philox draw -- is taken out of the tail of the kernel AND out
    // of its head: the operands are requested in `pre0`, i.e. behind the first operand transfers of the contraction, and
    // consumed in `pre`, while those transfers are in flight.
    //   a.epiDma (FAST kernels, aligned operands, C <= 12): a vector-memory instruction of a wavefront costs the kernel's
    //       start ~250 cycles whatever it moves, so instead of 1 + 2*C loads per thread ONE LDS-DMA transfer per wavefront
    //       (global_load_lds_dwordx4: a 64-bit address per lane) fetches the 16-column tile of U (C x 64 B), the y rows of
    //       its 4 strip rows (4 x 4C B, contiguous) and the bias tile (64 B) -- 5C + 4 <= 64 chunks of 16 bytes -- into a
    //       1 KiB scratch of its own, read back with ds_read behind vm_wait.  (Loads into registers that the compiler does
    //       not track are not an option here: it may copy their destination registers -- the hooks sit in the panel loop --
    //       before the data has arrived.)
    //   otherwise: ordinary loads, the compiler places the waits.
    constexpr bool with_dma = FASTK == 2;
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    char* cls_lds = lds + a.ldsCls + wv * 1024;   // chunk of lane L at 16 L: [0, 64C) U tile, [64C, 80C) y rows, [80C, 80C + 64) bias tile
    float d_g = 0.f, y_g[CM], u_g[CM];
    auto pre0 = [&]() {
        if constexpr (with_dma) {   // decided per wavefront (never per lane: a lane masked off by an outer condition would leave its chunk out)
            // lane L < 4C: U chunk (class L/4, columns n0 + 4*(L%4) ..); 4C <= L < 5C: chunk L - 4C of the 4*C floats of rows
            // m0 + 4*wv .. (no class term: none of these); the 4 lanes behind them: the chunks of d[n0 .. n0+15].
            // No range check on this path: lanes whose chunk lies beyond the operand are masked off (their LDS slot keeps
            // its old bytes; only threads of rows >= B / columns >= H would read it).
            const int nU = with_cls ? 4 * a.C : 0, nY = with_cls ? a.C : 0;
            const float* src = nullptr;
            if (lane < nU) {
                const int col = n0 + 4 * (lane & 3);
                if (col < a.H) src = a.U + (long)(lane >> 2) * a.H + col;
            } else if (lane < nU + nY) {
                const long f = (long)(m0 + 4 * wv) * a.C + 4 * (lane - nU);
                if (f < (long)a.B * a.C) src = a.y + f;
            } else if (lane < nU + nY + 4) {
                const int col = n0 + 4 * (lane - nU - nY);
                if (col < a.H) src = a.d + col;
            }
            if (src) __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)cls_lds, 16, 0, AUX);
        } else {
            d_g = a.d[j_c];
            if (with_cls) {
#pragma unroll
                for (int c = 0; c < CM; ++c) {
                    y_g[c] = a.y[(long)i_c * a.C + min(c, a.C - 1)];
                    u_g[c] = a.U[(long)min(c, a.C - 1) * a.H + j_c];
                }
            }
        }
    };
    float dv = 0.f, cls = 0.f, uu = 0.f;
    auto chain = [&](int c, float y, float u) {   // class c of the class term (classes >= C enter as 0 * 0)
        const bool on = c < a.C;
        cls = __builtin_fmaf(on ? y : 0.f, on ? u : 0.f, cls);
    };
    auto pre = [&](int n_younger) {
        if (MODE == 0 && (a.s || a.s8)) {
            // one Philox block serves columns 4*(j0>>2) .. +3: this thread takes word j0 & 3
            uint32_t prow = a.row0 + row_b + (uint32_t)row;
            asm volatile("" : "+v"(prow));   // the draw depends on this statement's position: it cannot be hoisted above the transfers
            const f32x4 u4 = uniform4(a.seed, a.offset + off_b, prow, (uint32_t)(j0 >> 2));
            const int e = j0 & 3;
            uu = e == 0 ? u4[0] : e == 1 ? u4[1] : e == 2 ? u4[2] : u4[3];
            asm volatile("" : "+v"(uu));   // ... nor sunk behind the wait below
        }
        if constexpr (!with_dma) {
            dv = d_g;
            if (with_cls) {
#pragma unroll
                for (int c = 0; c < CM; ++c) chain(c, y_g[c], u_g[c]);
            }
            return;
        }
        vm_wait(n_younger);   // this wavefront's transfer of pre0 (and everything older) has landed: visible to its ds_reads
        // Row r of the wavefront (= lane/16) has its C floats at 64C + 4*C*r; column (lane%16) of class c sits at
        // 64*c + 4*(lane%16), its bias at 80C + 4*(lane%16) (no class term: at 4*(lane%16)).  Reads beyond class C-1
        // return whatever the scratch holds: masked in `chain`.
        const uint32_t ua = lds_off(cls_lds) + 4u * (lane & 15);
        const uint32_t ya = lds_off(cls_lds) + (uint32_t)(64 * a.C + 4 * a.C * (lane >> 4));
        const uint32_t da = ua + (with_cls ? (uint32_t)(80 * a.C) : 0u);
        uint32_t d_l, y_l[CM], u_l[CM];
        asm volatile("ds_read_b32 %0, %1" : "=v"(d_l) : "v"(da) : "memory");
        if (with_cls) {
#pragma unroll
            for (int c = 0; c < CM; ++c) {
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(y_l[c]) : "v"(ya), "n"(4 * c) : "memory");
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(u_l[c]) : "v"(ua), "n"(64 * c) : "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(d_l));
        dv = __uint_as_float(d_l);
        if (with_cls) {
#pragma unroll
            for (int c = 0; c < CM; ++c) {
                asm volatile("" : "+v"(y_l[c]), "+v"(u_l[c]));
                chain(c, __uint_as_float(y_l[c]), __uint_as_float(u_l[c]));
            }
        }
    };
    QBM_STAMP(11);
    const auto o = strip_mma<MT, NT, FAST, FAST, 1, NQ, true, AUX>(A, Bv, m0, n0, a.V, lds, a.KP, pre0, pre, 0, ldsB, b_resident, post);
    if (o.valid(0) && row < a.B && j0 < a.H && st_w == 0u) {
        float x = o.sum[0].x + dv, pv, hv = 0.f;
        if (MODE == 0) {
            if (a.pre) a.pre[(long)row * a.H + j0] = x;
            if (a.y) x = x + cls;
            pv = det_sigmoid(x);
            hv = (pv > uu) ? 1.f : 0.f;
        } else {
            pv = x;
        }
        const long off = (long)row * a.H + j0;
        if (a.p) a.p[off] = pv;
        if (MODE == 0 && a.s) a.s[off] = hv;
        if (MODE == 0 && a.s8) a.s8[off] = (uint8_t)hv;
    }
    QBM_STAMP(12);
}

template <class TA, int FASTK, int MODE, int CM, bool XA = false>
__global__ __launch_bounds__(64 * H_NQ) void k_hidden(HiddenArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    QBM_STAMP(10);
    // every kernel argument in ONE round trip to the (cold) scalar cache: the compiler otherwise fetches them where they
    // are first used, one dependent s_load latency after the other in front of the first operand transfer
    asm volatile("" ::"s"(a.v), "s"(a.y), "s"(a.W), "s"(a.d), "s"(a.U), "s"(a.p), "s"(a.s), "s"(a.s8), "s"(a.B), "s"(a.V),
                 "s"(a.H), "s"(a.C), "s"(a.KP), "s"(a.seed), "s"(a.offset), "s"(a.row0), "s"(a.pre), "s"(a.pair));
    // grid (8, nM, ceil(nN / 8)): workgroups are numbered x + 8 (y + nM z) and dealt to the XCDs round-robin, so XCD x takes the
    // column strips x, x + 8, ... with all their row strips (the W panel of a column strip stays in one L2) -- and the
    // coordinates come without the integer division a linear grid needs (~60 scalar instructions with a v_rcp round trip
    // in front of the first operand request, on every wavefront)
    // a.pair (H a multiple of 256): in PAIRS, strips 2j, 2j + 1 on XCD j % 8 -- the gradient launch deals its column strips
    // (32 wide, or 16 wide in the same pairs) the same way, so it finds p(h), and the next step's first launch the W columns
    // the gradient launch has just updated, in its XCD's L2, which keeps its lines over a kernel boundary (CD-1 at 784 x 512,
    // B = 256: 29.5 -> 28.4 us per step; tools/step_timeline.py)
    const int bx = (int)blockIdx.x, bz = (int)blockIdx.z;
    const int ms = (int)blockIdx.y, ns = a.pair ? 16 * (bz >> 1) + 2 * bx + (bz & 1) : bx + 8 * bz;
    if (ns * 16 * H_NT >= a.H) return;   // (padding of the last group of 8 column strips)
    hidden_body<TA, FASTK, MODE, CM, 0, NoPre, XA>(a, ms, ns, lds);
}

// ---------------------------------------------------------------------------------------------
// A-3  sample_class  (src/ClassificationRBM.py:54-60): exp(h.U^T + c), L1-normalised (eps 1e-12);
// categorical draw: one uniform per row, first class whose running sum exceeds u.
// One workgroup per row: wave w takes the classes c = w, w+4, ... (per class the sum over j is the wave tree [T],
// exactly as one wave would do it), the energies meet in LDS (`e_s`: QBM_MAX_CLASSES floats + one slot for the row's
// uniform) and wavefront 0 normalises and draws, lane c the quotient of class c.  All loads are unconditional.
// ---------------------------------------------------------------------------------------------
// class_row_energies: the 4 wavefronts of a 256-thread team (tid = 0..255) fill e_s[c] = exp(h.U_c + c_c).  class_row = the
// energies, a workgroup barrier, the tail on wavefront 0 -- for a 256-thread team that is alone in its workgroup.
// under: called once by every wavefront while the loads of its first trip are in flight (class_row: the row's uniform)
// TR: columns per lane and trip (8: a row of up to 512 hidden units is ONE round of loads, not two one after the other)
template <class TH, int CM, bool SC1 = false, class Under = NoPre, int TR = 4>
__device__ __forceinline__ void class_row_energies(const void* __restrict__ h, const float* __restrict__ U,
                                                   const float* __restrict__ cb, int i, int H, int C, int tid,
                                                   float* __restrict__ e_s, Under&& under = NoPre{})
{
    constexpr int CW = (CM + 3) / 4;   // classes per wave
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float e[CW], cbv[CW];
#pragma unroll
    for (int t = 0; t < CW; ++t) { e[t] = 0.f; cbv[t] = cb[min(w + 4 * t, C - 1)]; }
    // per class the lane chains j = lane, lane+64, ... ascending ([T]).  64 TR columns per trip: all
    // TR + TR*CW loads of a trip are unconditional (index clamped, h masked to 0: fmaf(0, u, e) == e)
    for (int jb = 0; jb < H; jb += TR * kWave) {
        float hv[TR], uv[TR][CW];
#pragma unroll
        for (int t = 0; t < TR; ++t) {
            const int j = jb + t * kWave + lane;
            const int jj = min(j, H - 1);
            const float x = ld_state<TH, SC1>(h, (long)i * H + jj);
            hv[t] = (j < H) ? x : 0.f;
#pragma unroll
            for (int c = 0; c < CW; ++c) uv[t][c] = U[(long)min(w + 4 * c, C - 1) * H + jj];
        }
        if (jb == 0) under();
#pragma unroll
        for (int t = 0; t < TR; ++t)
#pragma unroll
            for (int c = 0; c < CW; ++c) e[c] = __builtin_fmaf(hv[t], uv[t][c], e[c]);
    }
#pragma unroll
    for (int t = 0; t < CW; ++t) {
        const float v = det_expf(wave_butterfly_sum(e[t]) + cbv[t]);
        if (lane == 0 && w + 4 * t < C) e_s[w + 4 * t] = v;
    }
}

// The same conditional by ONE wavefront (the fused chain kernel: a row per wavefront, no LDS, no workgroup barrier):
// per class the wave tree [T] over j -- lane l chains j = l, l + 64, ... ascending, then the butterfly -- exactly what
// one of class_row_energies' wavefronts does for its classes; lane 0 normalises and draws (denominator, quotients and running sum in class order).
template <class TH, int CM, bool SC1 = false>
__device__ __forceinline__ void class_row_wave(const void* __restrict__ h, const float* __restrict__ U,
                                               const float* __restrict__ cb, float* __restrict__ p_out,
                                               float* __restrict__ y_out, int i, int H, int C, uint64_t seed, uint64_t offset,
                                               uint32_t row0, int lane)
{
    float e[CM];
#pragma unroll
    for (int c = 0; c < CM; ++c) e[c] = 0.f;
    for (int jb = 0; jb < H; jb += 2 * kWave) {   // two columns per lane and trip: 2 + 2*CM loads in flight
        float hv[2], uv[2][CM];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int j = jb + t * kWave + lane;
            const int jj = min(j, H - 1);
            const float x = ld_state<TH, SC1>(h, (long)i * H + jj);
            hv[t] = (j < H) ? x : 0.f;
#pragma unroll
            for (int c = 0; c < CM; ++c) uv[t][c] = U[(long)min(c, C - 1) * H + jj];
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int c = 0; c < CM; ++c) e[c] = __builtin_fmaf(hv[t], uv[t][c], e[c]);
    }
    // (every lane holds every class's butterfly sum: lane c takes the exponential and the quotient of class c -- one of each
    // instead of C in a row --, the running sum in class order is formed from the lanes' quotients: the same operations on the
    // same values as one thread after the other)
    float mine = 0.f;
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        const float t = wave_butterfly_sum(e[c]) + cb[min(c, C - 1)];
        mine = (lane == c || (c == C - 1 && lane >= C)) ? t : mine;
    }
    mine = det_expf(mine);
    float denom = 0.f;
#pragma unroll
    for (int c = 0; c < CM; ++c)
        if (c < C) denom = denom + __builtin_fabsf(__shfl(mine, c, kWave));
    if (!(denom > 1e-12f)) denom = 1e-12f;
    const float u = uniform4(seed, offset, row0 + (uint32_t)i, 0u)[0];
    const float p = mine / denom;
    if (p_out && lane < C) p_out[(long)i * C + lane] = p;
    float cum = 0.f;
    int pick = C - 1;
    bool done = false;
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        if (c < C) {
            cum = cum + __shfl(p, c, kWave);
            if (!done && u < cum) { pick = c; done = true; }
        }
    }
    if (y_out && lane < C) y_out[(long)i * C + lane] = (lane == pick) ? 1.f : 0.f;
}

template <class TH, int CM>
__device__ __forceinline__ void class_row(const void* __restrict__ h, const float* __restrict__ U,
                                          const float* __restrict__ cb, float* __restrict__ p_out,
                                          float* __restrict__ y_out, int i, int H, int C, uint64_t seed,
                                          uint64_t offset, uint32_t row0, float* __restrict__ e_s)
{
    // The row's uniform is drawn by the last wavefront UNDER its first loads and parked behind the energies; the tail is then
    // one wavefront's: lane c divides class c (one division in parallel instead of C in a row on one lane), every lane forms
    // the running sum in class order from the lanes' quotients -- the same operations on the same values as one thread doing
    // them in a row (class_row_wave's tail), a third of the instructions on the path to y.
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    auto under = [&]() {
        if (w == 3) {
            const float u = uniform4(seed, offset, row0 + (uint32_t)i, 0u)[0];
            if (lane == 0) e_s[QBM_MAX_CLASSES] = u;
        }
    };
    if (H > 4 * kWave) class_row_energies<TH, CM, false, decltype(under)&, 8>(h, U, cb, i, H, C, (int)threadIdx.x, e_s, under);
    else class_row_energies<TH, CM, false, decltype(under)&, 4>(h, U, cb, i, H, C, (int)threadIdx.x, e_s, under);
    __syncthreads();
    if (w != 0) return;
    float denom = 0.f;
#pragma unroll
    for (int c = 0; c < CM; ++c)
        if (c < C) denom = denom + __builtin_fabsf(e_s[c]);
    if (!(denom > 1e-12f)) denom = 1e-12f;
    const float u = e_s[QBM_MAX_CLASSES];
    const float p = e_s[min(lane, C - 1)] / denom;
    if (p_out && lane < C) p_out[(long)i * C + lane] = p;
    float cum = 0.f;
    int pick = C - 1;
    bool done = false;
#pragma unroll
    for (int c = 0; c < CM; ++c) {
        if (c < C) {
            cum = cum + __shfl(p, c, kWave);
            if (!done && u < cum) { pick = c; done = true; }
        }
    }
    if (y_out && lane < C) y_out[(long)i * C + lane] = (lane == pick) ? 1.f : 0.f;
}

template <class TH, int CM>
__global__ __launch_bounds__(256) void k_class(const void* __restrict__ h, const float* __restrict__ U,
                                               const float* __restrict__ cb, float* __restrict__ p_out,
                                               float* __restrict__ y_out, int B, int H, int C, uint64_t seed,
                                               uint64_t offset, uint32_t row0)
{
    __shared__ float e_s[QBM_MAX_CLASSES + 1];
    class_row<TH, CM>(h, U, cb, p_out, y_out, (int)blockIdx.x, H, C, seed, offset, row0, e_s);
}

// (defined with k_free_energy below; the training loop's monitor rides in the k_visible launch as aux workgroups)
template <int CM>
__device__ __forceinline__ void free_energy_row(const float* __restrict__ pre, const float* __restrict__ U,
                                                const float* __restrict__ cb, float* __restrict__ negF, float* __restrict__ p_out,
                                                int i, int H, int C, const int* __restrict__ labels, float* __restrict__ diff,
                                                float* __restrict__ rowloss, int tid, int nw, float* __restrict__ sm,
                                                float* __restrict__ sp_lds = nullptr);

// ---------------------------------------------------------------------------------------------
// A-2  sample_visible  (src/ClassificationRBM.py:49-52):  p = sigma(h.W^T + b);  v = (p > u)
// W is read as stored ([V,H], H contiguous): it is the K-contiguous B operand.
// The launch can carry the class conditional of the same hidden state (A-3) as extra workgroups
// placed first in the grid (one row each): both read h only, so one launch serves both and the tiny
// class product hides behind the strips.
// ---------------------------------------------------------------------------------------------
struct VisibleArgs {
    const void* h;   // [B,H] TA
    const float* W;  // [V,H]
    const float* b;  // [V]
    float* p;        // [B,V] or null
    float* s;        // [B,V] or null  fp32 {0,1}
    uint8_t* s8;     // [B,V] or null  uint8 {0,1}
    int B, V, H;
    int KP;
    uint64_t seed, offset;
    uint32_t row0;
    bool vecO;
    // class conditional riding along (all null/0 when not wanted)
    const float* U; const float* cb; float* py; float* ys;
    int C; uint64_t offset_y;
    int n_aux;  // aux workgroups come first in the grid
    const uint32_t* status;   // as HiddenArgs::status
    // the training loop's monitor riding along (qbm_rbm_cd_workspace.mon_*): n_fe = B more aux workgroups behind the class
    // rows, one row each: -F, p(y|x0) and the row's two loss terms from x0.W + d (stored by the launch in front)
    int n_fe; const float* fe_pre; const int* fe_labels; float* fe_p; float* fe_rowloss;
    int fe_stage;             // the softplus terms of a monitor row staged through LDS (they fit behind the row's small arrays)
    unsigned inv_nm;          // div_magic(row strips): the strips' coordinates without an integer division
    const uint64_t* draw_base;   // as HiddenArgs::draw_base (added to offset, offset_y and row0)
    uint32_t* hg_round;       // non-null: the fused tail of the step (k_hgrad) is the next launch: this one advances the word
                              // that launch takes its barrier round from (one writer, a kernel boundary in front of the readers:
                              // the same number in every workgroup whatever shapes earlier launches on the buffer had)
    int stag;                 // the strips stagger their requests (QBM_VIS_STAGGER): where a CU holds ONE strip workgroup and the chains are long
};

// 8 wavefronts (512 threads) per strip workgroup: wavefront q contracts chain q of the 8 the K = H terms are cut into
// ([Q] with NQ = 8); two visible units per thread in the epilogue.
// Request stagger of the visible strips (strip_mma's STAG, units of 64 cycles): with 16 the second wavefront of every SIMD asks
// for its operands ~1000 cycles behind the first, whose MFMAs then start earlier: k_visible 6.55 -> 6.36 us (8: 6.41, 32: 6.48,
// 48: 6.85; A/B in one session through QBM_HIP_LIB).  The same stagger LOSES in k_hidden (5.90 -> 5.97 / 6.11 us at 8 / 16).
// Switched on per launch (VisibleArgs::stag).
#ifndef QBM_VIS_STAGGER
#define QBM_VIS_STAGGER 16
#endif
// (the visible phases of the fused chain: PCD-10 128.8 - 129.2 -> 128.3 us per step with 16, 128.5 with 8)
#ifndef QBM_CHAIN_VIS_STAGGER
#define QBM_CHAIN_VIS_STAGGER 16
#endif
constexpr int V_NQ = 8;
// The body of strip (ms, ns) of (16 MT) x (16 NT) outputs; the caller is a 512-thread workgroup.  AUX: cache policy of the
// loads of h (kSc1 inside the fused chain kernel).
template <class TA, bool FAST, int MT, int NT, int AUX = 0, bool XA = false>
__device__ __forceinline__ void visible_body(const VisibleArgs& a, int ms, int ns, char* lds)
{
    constexpr int NQ = V_NQ;
    const int m0 = ms * 16 * MT, n0 = ns * 16 * NT;
    const typename KView<TA>::type A{reinterpret_cast<const TA*>(a.h), a.H, a.B};
    const ViewKContig Bv{a.W, a.H, a.V};
    // Epilogue map: the strip (64 x 16, or 32 x 32 in the fused chain) has 1024 outputs = 2 consecutive visible units for
    // each of the 512 threads.
    // The bias loads are issued behind the first operand transfers (`pre0`) and consumed in the epilogue; the Philox draw
    // runs in `pre`, while the transfers are in flight.
    using SO = StripOut<MT, NT, 2, 64 * NQ>;
    static_assert(SO::NP == 1, "one output pair per thread");
    const int i = m0 + SO::row(0), k0 = n0 + 2 * SO::cg(0);
    const bool pair = a.vecO && k0 + 1 < a.V;
    float bv[2] = {0.f, 0.f}, uu[2] = {0.f, 0.f};
    uint32_t st_w = 0u;
    uint64_t off_b = 0ull;
    uint32_t row_b = 0u;
    if constexpr (XA) {   // (scalar loads issued here, consumed in `pre` / in front of the stores: see hidden_body)
        const unsigned long long* db = a.draw_base ? reinterpret_cast<const unsigned long long*>(a.draw_base) : g_zero_base;
        const uint32_t* sp = a.status ? a.status : reinterpret_cast<const uint32_t*>(g_zero_base);
        st_w = *sp; off_b = db[0]; row_b = (uint32_t)db[1];
    }
    auto pre0 = [&]() {
        if (pair) {
            const float2 t = *reinterpret_cast<const float2*>(a.b + k0);
            bv[0] = t.x; bv[1] = t.y;
        } else {
            bv[0] = a.b[min(k0, a.V - 1)];
            bv[1] = a.b[min(k0 + 1, a.V - 1)];
        }
    };
    auto pre = [&](int) {
        uint32_t prow = a.row0 + row_b + (uint32_t)i;
        asm volatile("" : "+v"(prow));     // the draw depends on this statement's position: not hoisted above the transfers,
        if (a.s || a.s8) {
            const f32x4 u4 = uniform4(a.seed, a.offset + off_b, prow, (uint32_t)(k0 >> 2));
            uu[0] = (k0 & 2) ? u4[2] : u4[0];   // this thread's pair is the lower or the upper half of the block
            uu[1] = (k0 & 2) ? u4[3] : u4[1];
        }
        asm volatile("" : "+v"(uu[0]), "+v"(uu[1]));   // not sunk into the epilogue
    };
    QBM_STAMP(11);
    const auto o = strip_mma<MT, NT, FAST, FAST, 2, NQ, true, AUX, 0, kWhole, (AUX == 0 ? QBM_VIS_STAGGER : QBM_CHAIN_VIS_STAGGER)>(A, Bv, m0, n0, a.H, lds, a.KP, pre0, pre, 0, nullptr, false, NoPre{}, 0,
                                                                                                          AUX != 0 || a.stag != 0);
    if (o.valid(0) && i < a.B && k0 < a.V && st_w == 0u) {
        const float sv[2] = {o.sum[0].x, o.sum[0].y};
        float pv[2], vv[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            pv[e] = det_sigmoid(sv[e] + bv[e]);
            vv[e] = (pv[e] > uu[e]) ? 1.f : 0.f;
        }
        const long off = (long)i * a.V + k0;
        if (pair) {
            if (a.p) *reinterpret_cast<float2*>(a.p + off) = make_float2(pv[0], pv[1]);
            if (a.s) *reinterpret_cast<float2*>(a.s + off) = make_float2(vv[0], vv[1]);
            if (a.s8) *reinterpret_cast<uint16_t*>(a.s8 + off) = (uint16_t)((uint32_t)vv[0] | ((uint32_t)vv[1] << 8));
        } else {
            for (int e = 0; e < 2 && k0 + e < a.V; ++e) {
                if (a.p) a.p[off + e] = pv[e];
                if (a.s) a.s[off + e] = vv[e];
                if (a.s8) a.s8[off + e] = (uint8_t)vv[e];
            }
        }
    }
    QBM_STAMP(12);
}

template <class TA, bool FAST, int CM, bool XA = false>
__global__ __launch_bounds__(64 * V_NQ) void k_visible(VisibleArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int nMs = (a.B + 16 * V_MT - 1) / (16 * V_MT), nNs = (a.V + 16 * V_NT - 1) / (16 * V_NT);
    // aux workgroups in front of the strips (or none): the strips' part of the grid is padded so that the row strips that
    // read one block of W rows share an XCD for ANY number of visible strips (49 at V = 784: strip_coords_padded)
    const bool padded = a.n_aux == a.B || a.n_aux == 0;
    if constexpr (XA) {   // aux workgroups behind a fused launch that gave up: nothing to do (the strips test the word in front of their stores)
        const bool is_aux = a.n_aux == a.B ? (int)blockIdx.x < a.n_aux + a.n_fe : (int)blockIdx.x >= nMs * nNs;
        if (a.n_aux > 0 && is_aux && a.status && *a.status != 0u) return;
        if (a.hg_round && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0 && *a.status == 0u) *a.hg_round = *a.hg_round + 1u;
    }
    if (a.n_aux == a.B) {
        // aux workgroups dealt first: the class conditional of one row each, its classes dealt to the first 4 wavefronts --
        // the shortest path to y (a wavefront that has ended takes no part in the workgroup's barriers)
        if ((int)blockIdx.x < a.n_aux) {
            if (threadIdx.x >= 256) return;
            uint64_t oy = a.offset_y;
            uint32_t r0 = a.row0;
            if constexpr (XA) {
                const unsigned long long* db = a.draw_base ? reinterpret_cast<const unsigned long long*>(a.draw_base) : g_zero_base;
                oy += db[0]; r0 += (uint32_t)db[1];
            }
            class_row<TA, CM>(a.h, a.U, a.cb, a.py, a.ys, (int)blockIdx.x, a.H, a.C, a.seed, oy, r0, reinterpret_cast<float*>(lds));
            return;
        }
        if ((int)blockIdx.x < a.n_aux + a.n_fe) {   // the monitor: one row per workgroup, its classes dealt to the 8 wavefronts (in
                                                    // front of the strips: behind them -- the ragged end of the launch -- they
                                                    // start late and the launch ends 4.8 us later; measured)
            float* sm = reinterpret_cast<float*>(lds);   // [small arrays: (2 + CM)(CM + 1) <= 272 floats | staged softplus terms C x H, if they fit]
            free_energy_row<CM>(a.fe_pre, a.U, a.cb, nullptr, a.fe_p, (int)blockIdx.x - a.n_aux, a.H, a.C, a.fe_labels, nullptr,
                                a.fe_rowloss, (int)threadIdx.x, V_NQ, sm, a.fe_stage ? sm + 320 : nullptr);
            return;
        }
    } else if (!padded && (int)blockIdx.x >= nMs * nNs) {
        // large minibatches: 8 rows per aux workgroup, one per wavefront, dealt BEHIND the strips (they fill the ragged end of
        // the launch instead of standing in front of it)
        const int i = ((int)blockIdx.x - nMs * nNs) * V_NQ + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
        if (i < a.B) {
            uint64_t oy = a.offset_y;
            uint32_t r0 = a.row0;
            if constexpr (XA) {
                const unsigned long long* db = a.draw_base ? reinterpret_cast<const unsigned long long*>(a.draw_base) : g_zero_base;
                oy += db[0]; r0 += (uint32_t)db[1];
            }
            class_row_wave<TA, CM>(a.h, a.U, a.cb, a.py, a.ys, i, a.H, a.C, a.seed, oy, r0, (int)threadIdx.x & 63);
        }
        return;
    }
    QBM_STAMP(10);
    asm volatile("" ::"s"(a.h), "s"(a.W), "s"(a.b), "s"(a.p), "s"(a.s), "s"(a.s8), "s"(a.B), "s"(a.V), "s"(a.H), "s"(a.KP),
                 "s"(a.seed), "s"(a.offset), "s"(a.row0));   // all arguments in one scalar-cache round trip (see k_hidden)
    int ms, ns;
    if (padded) {
        if (!strip_coords_padded(nMs, nNs, ms, ns, a.n_aux + a.n_fe, a.inv_nm)) return;
    } else {
        strip_coords(nMs, nNs, ms, ns, 0, a.inv_nm);
    }
    visible_body<TA, FAST, V_MT, V_NT, 0, XA>(a, ms, ns, lds);
}

// ---------------------------------------------------------------------------------------------
// A-7 fused: the whole block-Gibbs chain of a CD-k / PCD-k step in ONE launch.
//
// The rows of a minibatch are independent along the chain h_0 -> (v_1, y_1) -> h_1 -> ...: only the gradient couples
// them.  MI355X has 8 XCDs of 32 CUs, each XCD with an L2 of its own that is coherent for its CUs.  So row block g (32
// rows) is given to the 32 workgroups of ONE XCD -- hidden phase: workgroup j owns the 32 x 16 strip of hidden columns
// 16 j ..; visible phase: the 32 x 32 strip of visible columns 32 j .. (the workgroups without a strip take the class
// conditional of the block's rows) -- and the phases of a block are separated by a barrier among those 32 workgroups
// only: every workgroup publishes its round number in a word of the block's 128-byte flag line with a plain store (it
// lands in the XCD's L2) and polls the line with ONE L1-bypassing (`sc1`) 32-lane load per try.  Measured
// (tools/microbench/xcd_barrier.hip): 0.27 us per barrier, against 1.5-1.9 us for a dependent kernel boundary and
// 3.6 us for a barrier built on an atomic counter.  A state handed from phase to phase is written with plain stores,
// drained (s_waitcnt vmcnt(0)) before the flag, and read with `sc1` transfers: served by the L2 the producer wrote
// through, never by the reader's own vector L1 (which no other CU's store refreshes).  W, b, d, U, c are not written
// during the launch: plain loads, and the second and later phases find them in the XCD's L2.
//
// Placement: HIP promises no workgroup -> XCD map.  Workgroups b and b + 8 have shared an XCD in every observation
// (blocks are dealt round-robin); the kernel takes group = b % 8 on that basis but CHECKS it: every workgroup
// publishes HW_REG_XCC_ID with its first flag and compares the 32 ids of its group after the first barrier; a
// mismatch, or a barrier whose bounded spin runs out, sets the status word (sync[512]; read by the host at its next
// synchronisation point: rbm_ops.check_chain_status) and ends the kernel -- never a hang, never a silently wrong result.
// The persistent chains are first written in the last visible phase, i.e. after the check.
// ---------------------------------------------------------------------------------------------
struct ChainArgs {
    const float *W, *b, *d, *U, *c;
    const uint8_t* v0;   // [B,V] data
    const float* y0;     // [B,C]
    uint8_t* chain_v;    // persistent chains, or null
    float* chain_y;
    float* ph0;          // workspace
    uint8_t* h;
    uint8_t* vk;
    float* yk;
    float* phk;
    uint32_t* sync;      // [8][32] round flags | [8][32] XCC ids | status
    int B, V, H, C, k;
    int KPh, KPv, ldsCls;
    int ldsT;            // the W panel of the hidden phases stays in LDS bytes [0, ldsT); everything else is staged behind it
    uint64_t seed, offset;
    uint32_t row0;
    bool vecO;
    float* mon_pre;      // non-null: the first hidden phase also stores v0.W + d (the training loop's monitor)
};

constexpr int kChainSpin = 1 << 20;   // polls (>= 0.3 us each) before a barrier gives up

__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xfu;
}

__device__ __forceinline__ uint32_t load_sc1(const uint32_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Barrier among the 32 workgroups of a group (every thread of the workgroup calls it).  false: the spin ran out.
// Giving up: the status word (sync[512]) and, for the host's replay, the draw offset of the launch (sync[513], [514]).
// (status[3]: the round of the launch, for the fused hidden + gradient launch whose unfinished work the host has repeated
// -- k_hgrad)
__device__ __forceinline__ void chain_give_up(uint32_t* status, uint32_t code, uint64_t offset, uint32_t round = 0u)
{
    status[1] = (uint32_t)offset;
    status[2] = (uint32_t)(offset >> 32);
    status[3] = round;
    __hip_atomic_store(status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// n: workgroups of the group (their flags are the first n words of the line); code: what a timeout records
__device__ __forceinline__ bool group_barrier(uint32_t* flags, int j, uint32_t target, uint32_t* status, uint32_t* s_ok,
                                              uint64_t offset, int n = 32, uint32_t code = 1u)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's stores have reached the L2
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (lane == 0) __hip_atomic_store(&flags[j], target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        bool ok = true;
        for (int spins = 0;; ++spins) {
            const uint32_t f = lane < n ? load_sc1(&flags[lane]) : target;
            if (__all((int)(f - target) >= 0)) break;
            if (spins > kChainSpin) { ok = false; break; }
        }
        if (lane == 0) {
            if (!ok) chain_give_up(status, code, offset, target);
            *s_ok = ok ? 1u : 0u;
        }
    }
    __syncthreads();
    return *s_ok != 0u;
}

template <int CM>
__global__ __launch_bounds__(512) void k_chain(ChainArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ uint32_t s_ok;
    const int g = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
    if (g * 32 >= a.B) return;   // no row block for this group (B < 256): none of its workgroups takes part in anything
    uint32_t* flags = a.sync + g * 32;
    uint32_t* ids = a.sync + 256 + g * 32;
    uint32_t* status = a.sync + 512;
    // A launch that gave up leaves the status word set and the round flags of its group ragged: until the host has looked
    // (rbm_ops.check_chain_status re-zeroes the buffer and replays the steps on the launch-per-phase path) every later
    // launch on this workspace does nothing -- and so do the k_grad / k_update launches behind it, which read the same word
    // (GradArgs::status): a failed chain never reaches the parameters.
    if (load_sc1(status) != 0u) return;
    uint32_t round = flags[j];   // my last round of the previous launch: the same in every word of the line
    QBM_STAMP(5);
    int n_bar = 0;   // (diagnostic builds stamp the first two barriers: slots 6/7 and 8/9)
    (void)n_bar;
    const uint32_t my_xcc = xcc_id();
    if (threadIdx.x == 0) ids[j] = my_xcc;
    bool checked = false;
    bool w_resident = false;   // this workgroup's 16 columns of W are in LDS: staged by its first hidden phase, kept for the others

    const bool pcd = a.chain_v != nullptr;
    const int nNh = (a.H + 15) / 16, nNv = (a.V + 31) / 32;   // <= 32 each
    const int n_free = 32 - nNv;                              // workgroups without a visible strip
    const int n_pre = pcd ? 2 : 1;
    const int nph = n_pre + 2 * a.k;
    for (int ph = 0; ph < nph; ++ph) {
        // ---- decode the phase (the host-side sequence of qbm_rbm_cd_chain)
        bool hidden = true, sync_first = false;
        const uint8_t* vin = a.v0;
        const float* yin = a.y0;
        float* pout = a.ph0;
        uint8_t* hout = pcd ? nullptr : a.h;
        uint8_t* vout = nullptr;
        float* yout = nullptr;
        uint64_t off = a.offset;
        if (ph == 1 && pcd) {          // negative chain starts from the persistent state
            vin = a.chain_v; yin = a.chain_y; pout = nullptr; hout = a.h;
        } else if (ph >= n_pre) {
            const int t = (ph - n_pre) / 2 + 1;
            const uint64_t o = a.offset + 3ull * (uint64_t)t;
            vout = (pcd && t == a.k) ? a.chain_v : a.vk;
            yout = (pcd && t == a.k) ? a.chain_y : a.yk;
            sync_first = true;
            if (((ph - n_pre) & 1) == 0) { hidden = false; off = o; }
            else { vin = vout; yin = yout; pout = a.phk; hout = t < a.k ? a.h : nullptr; off = o; }
        }
        if (sync_first) {
            ++round;
            if (n_bar < 2) QBM_STAMP(6 + 2 * n_bar);
            if (!group_barrier(flags, j, round, status, &s_ok, a.offset)) return;
            if (n_bar < 2) QBM_STAMP(7 + 2 * n_bar);
            ++n_bar;
            if (!checked) {   // all 32 workgroups of the group on one XCD?
                checked = true;
                if (threadIdx.x < 64) {
                    const uint32_t id = threadIdx.x < 32 ? load_sc1(&ids[threadIdx.x]) : my_xcc;
                    const bool same = __all((int)(id == my_xcc));
                    if (threadIdx.x == 0) {
                        if (!same) chain_give_up(status, 2u, a.offset);
                        s_ok = same ? 1u : 0u;
                    }
                }
                __syncthreads();
                if (s_ok == 0u) return;
            }
        }
        if (hidden) {
            if (j < nNh) {
                const HiddenArgs ha{vin, yin, a.W, a.d, a.U, pout, nullptr, hout, a.B, a.V, a.H, a.C, a.KPh, a.seed, off, a.row0,
                                    true, a.ldsCls, ph == 0 ? a.mon_pre : nullptr, nullptr, nullptr};
                hidden_body<uint8_t, 2, 0, CM, kSc1>(ha, g, j, lds + a.ldsT, lds, w_resident);
                w_resident = true;
            }
        } else {
            if (j < nNv) {
                const VisibleArgs va{a.h, a.W, a.b, nullptr, nullptr, vout, a.B, a.V, a.H, a.KPv, a.seed, off - 2, a.row0, a.vecO,
                                     a.U, a.c, nullptr, yout, a.C, off - 1, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0u, nullptr, nullptr};
                visible_body<uint8_t, true, 2, 2, kSc1>(va, g, j, lds + a.ldsT);
            }
            // class conditional of the block's 32 rows, one row per wavefront, on the workgroups that have no strip -- or,
            // when every workgroup has one, on the first wavefront of each behind its strip
            const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
            const int r = n_free > 0 ? (j - nNv) * 8 + wv : (wv == 0 ? j : 32);
            if (j >= nNv || n_free == 0)
                for (int rr = r; rr >= 0 && rr < 32; rr += 8 * (n_free > 0 ? n_free : 32))
                    class_row_wave<uint8_t, CM, true>(a.h, a.U, a.c, nullptr, yout, g * 32 + rr, a.H, a.C, a.seed, off - 1, a.row0,
                                                      (int)threadIdx.x & 63);
        }
    }
    QBM_STAMP(13);
}

// ---------------------------------------------------------------------------------------------
// A-4  sample_class_given_x, second half (src/ClassificationRBM.py:68-82), given pre = x.W + d:
//   -F_y = c_y + sum_j log(1 + exp(pre_j + U_yj));   p_c = 1 / sum_d exp(-(-F_c) + (-F_d))
// One wavefront per row, wave tree [T] over j.
// ---------------------------------------------------------------------------------------------
// One workgroup per row, one wave per class (blockDim = 64 C): wave c does the wave tree [T] over j for class c,
// the -F values meet in LDS, thread (c, d) forms exp(F_c - F_d) and thread c adds them up in d order -> p(y|x).
// With `diff` set, the workgroup goes on to the elementwise stage of discriminative_training for its row while p is
// still in LDS (src/ClassificationRBM.py:106-107, :113-115, :124-126, :134):  o_c = sigma(pre + U_c);  pos = o_{label};
// neg = sum_c o_c * p_c (mul, then add, classes ascending);  diff = pos - neg -- and to the two scalars the training loop
// books per row (`rowloss`): CrossEntropyLoss applied to the probabilities, log(sum_c exp(p_c)) - p_label (:142), and
// -log(p_label + 1e-8) (:188-190).
// All loads of a stage are issued before its arithmetic (a lane's 8 columns of a 512-unit row: a dependent L2 round trip
// per column would otherwise be most of the kernel).
// One row by a team of NW wavefronts (tid = 0 .. 64 NW - 1; every thread of the team calls it: workgroup barriers inside):
// wavefront w takes the classes w, w + NW, ... (k_free_energy: NW = C, one class each; as aux workgroups of k_visible:
// NW = 8).  sm: (2 + CM) (CM + 1) floats of LDS.
template <int CM>
__device__ __forceinline__ void free_energy_row(const float* __restrict__ pre, const float* __restrict__ U,
                                                const float* __restrict__ cb, float* __restrict__ negF, float* __restrict__ p_out,
                                                int i, int H, int C, const int* __restrict__ labels, float* __restrict__ diff,
                                                float* __restrict__ rowloss, int tid, int nw, float* __restrict__ sm,
                                                float* __restrict__ sp_lds)
{
    float* nf_s = sm;
    float* p_s = sm + (CM + 1);
    float* e_s = sm + 2 * (CM + 1);
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lab = labels ? labels[i] : 0;
    // sp_lds ([C][H] floats of LDS, or null): the softplus terms -- the expensive part -- are evaluated by ALL threads of the
    // team into LDS first, and each class's wavefront then only walks its chain over them (same values, same order: [T]).
    // Pays when the team has more wavefronts than classes (C = 2 with 8 wavefronts: 2 evaluations per thread instead of 8 on
    // two wavefronts).
    if (sp_lds) {
        for (int idx = tid; idx < C * H; idx += 64 * nw) {
            const int cls = idx / H, j = idx - cls * H;
            sp_lds[idx] = det_softplus(pre[(long)i * H + j] + U[(long)cls * H + j]);
        }
        __syncthreads();
    }
    for (int cls = w; cls < C; cls += nw) {
        float nf = 0.f;
        const float cbv = cb[cls];
        const float* prow = pre + (long)i * H;
        const float* urow = U + (long)cls * H;
        if (sp_lds) {
            const float* sp = sp_lds + cls * H;
            for (int j = lane; j < H; j += kWave) nf = nf + sp[j];
        } else
        for (int jb = 0; jb < H; jb += 8 * kWave) {   // the lane chains j = lane, lane + 64, ... ascending ([T])
            float pv[8], uv[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int jj = min(jb + t * kWave + lane, H - 1);
                pv[t] = prow[jj];
                uv[t] = urow[jj];
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float sp = det_softplus(pv[t] + uv[t]);
                nf = (jb + t * kWave + lane < H) ? nf + sp : nf;
            }
        }
        const float v = cbv + wave_butterfly_sum(nf);
        if (lane == 0) nf_s[cls] = v;
    }
    __syncthreads();
    for (int t = tid; t < CM * CM; t += 64 * nw) {
        const int c = t / CM, dd = t % CM;
        if (c < C && dd < C) e_s[c * CM + dd] = det_expf((-1.0f * nf_s[c]) + nf_s[dd]);
    }
    __syncthreads();
    if (tid < C) {
        const int c = tid;
        float s = 0.f;
        for (int dd = 0; dd < C; ++dd) s = s + e_s[c * CM + dd];
        const float pc = 1.0f / s;
        p_s[c] = pc;
        if (negF) negF[(long)i * C + c] = nf_s[c];
        if (p_out) p_out[(long)i * C + c] = pc;
    }
    if (!diff && !rowloss) return;
    __syncthreads();
    if (w == nw - 1 && rowloss) {
        // the row's two loss terms by the LAST wavefront (with more wavefronts than 64-column blocks it has no `diff` column
        // to do), lane c the exponential of class c, lanes 0 / 1 one logarithm each -- the same operations on the same values
        // as one thread after the other (C exponentials and two logarithms in a row were the longest path of the workgroup)
        const float ex = det_expf(p_s[min(lane, C - 1)]);
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < CM; ++c)
            if (c < C) se = se + __shfl(ex, c, kWave);
        const float py = p_s[lab];
        const float lg = det_logf(lane == 0 ? se : py + 1e-8f);
        if (lane < 2) rowloss[2 * (long)i + lane] = lane == 0 ? lg - py : -lg;
    }
    if (!diff) return;
    for (int j = tid; j < H; j += 64 * nw) {
        const float pr = pre[(long)i * H + j];
        float uv[CM];
#pragma unroll
        for (int c = 0; c < CM; ++c) uv[c] = U[(long)min(c, C - 1) * H + j];
        float ng = 0.f, ps = 0.f;
#pragma unroll
        for (int c = 0; c < CM; ++c) {
            if (c < C) {
                const float o = det_sigmoid(pr + uv[c]);
                ps = (c == lab) ? o : ps;
                ng = ng + o * p_s[c];
            }
        }
        diff[(long)i * H + j] = ps - ng;
    }
}

// blockDim = 64 * max(C, kFeMinWaves): at least 8 wavefronts per row (with fewer classes the spare ones evaluate softplus
// terms and the discriminative step's `diff` columns); dynamic LDS: C * H floats for the staged softplus terms, or 0
constexpr int kFeMinWaves = 8;
template <int CM>
__global__ __launch_bounds__(64 * (CM > kFeMinWaves ? CM : kFeMinWaves)) void k_free_energy(
    const float* __restrict__ pre, const float* __restrict__ U, const float* __restrict__ cb, float* __restrict__ negF,
    float* __restrict__ p_out, int B, int H, int C, const int* __restrict__ labels, float* __restrict__ diff,
    float* __restrict__ rowloss, const uint32_t* __restrict__ status, int stage_sp)
{
    if (status && *status != 0u) return;   // (the CD step's monitor behind a fused launch that gave up: HiddenArgs::status)
    extern __shared__ __attribute__((aligned(16))) float sp_dyn[];
    __shared__ float sm[(2 + CM) * (CM + 1)];
    free_energy_row<CM>(pre, U, cb, negF, p_out, (int)blockIdx.x, H, C, labels, diff, rowloss, (int)threadIdx.x, (int)(blockDim.x >> 6), sm,
                        stage_sp ? sp_dyn : nullptr);
}

// launch of k_free_energy: 8 wavefronts at least; the softplus terms staged through LDS when the team has spare wavefronts
// (C < 8) and they fit
template <int CM>
void launch_free_energy(const float* pre, const float* U, const float* cb, float* negF, float* p_out, int B, int H, int C,
                        const int* labels, float* diff, float* rowloss, const uint32_t* status, hipStream_t st)
{
    const int waves = C > kFeMinWaves ? C : kFeMinWaves;
    const size_t sp_bytes = (size_t)C * H * sizeof(float);
    const int stage = (C < kFeMinWaves && sp_bytes <= 49152) ? 1 : 0;
    hipLaunchKernelGGL((k_free_energy<CM>), dim3(B), dim3(64 * waves), stage ? sp_bytes : 0, st, pre, U, cb, negF, p_out, B, H, C,
                       labels, diff, rowloss, status, stage);
}

// ---------------------------------------------------------------------------------------------
// A-5 / A-7 gradient (reference association sums: src/ClassificationRBM.py:118-138).
// Every block of the gradient is a two-phase strip contraction over the batch (quartered chain [Q]):
//   kind W : gW = A0^T.P0 - A1^T.P1                    A = data[b][m] (lin image; fp32 or u8), B = P
//   kind U : [gU; gd] = [Y0 | 1]^T.P0 - [Y1 | 1]^T.P1   A = ViewAugY (class rows + a row of ones), B = P
//   kind B : gb = A0^T.1 - A1^T.1                       A = data, B = a column of ones
//   kind C : gc = 1^T.Y0 - 1^T.Y1                       A = a row of ones, B = Y
// The few U/B/C workgroups come first in the grid.  Optional fused A-6 update
// (src/ClassificationRBM.py:88-99) in the epilogue of each kind (single-GPU path; the data-parallel
// path all-reduces the gradients first).
// ---------------------------------------------------------------------------------------------

// B operand with a single column of ones (column sums of the A operand land in output column 0)
struct ViewOnesCol {
    using elem = float;
    static constexpr bool kK4 = false;
    __device__ __forceinline__ rsrc_t rsrc() const { return make_rsrc(nullptr, 0); }
    __device__ __forceinline__ DmaLane lane0(int, int, int) const { return DmaLane{kOOB, 0u, 0, 0}; }
    __device__ __forceinline__ float4 get(int c0, int kb, int kend, int slot) const
    {
        const int c = c0 + 4 * (slot & 3), k = kb + (slot >> 2);
        return make_float4((k < kend && c == 0) ? 1.f : 0.f, 0.f, 0.f, 0.f);
    }
};

struct GradArgs {
    const void* A0; const float* Y0; const float* P0;
    const void* A1; const float* Y1; const float* P1;
    float* gW; float* gb; float* gd; float* gU; float* gc;
    float* W;                                    // non-null: fused W update
    float* ub; float* ud; float* uU; float* uc;  // non-null: fused update of b, d, U, c
    float lr, factor, bs, sparse;
    int B, V, H, C;
    int KP;
    int nU, nB, nC;                              // workgroups of kind U, B, C (they precede kind W)
    bool vecO;
    const float* gUsub;                          // non-null: class-weight gradient = (phase0 - phase1) - gUsub
    int lds_group;                               // bytes of one group's LDS staging region
    int lds_group0;                              // k_hgrad: byte offset of the staging area of phase 0 (behind what the hidden phase uses)
    int lds_sok;                                 // k_hgrad: byte offset of the barrier's result word
    unsigned inv_nm;                             // div_magic(row strips of W): strip coordinates without an integer division
    const uint32_t* status;                      // non-null: status word of the fused chain launch in front (sync[512]);
                                                 // non-zero = that launch gave up: this launch writes nothing (but n_rows)
    float* n_rows;                               // non-null: receives (float)B -- the data-parallel path all-reduces it with the
                                                 // gradients -- or -inf when `status` is set (every rank then skips the update)
    // discriminative step (kind D workgroups, BEHIND the strips in the grid; P1 == null: one phase only)
    int nD;                                      // workgroups of kind D: C x ceil(ceil(H / 16) / dW)
    int dW;                                      // blocks of 16 hidden columns per kind D workgroup
    const float* pre; const float* Up; const float* pyx; const int* labels;   // x.W + d [B,H], U [C,H], p(y|x) [B,C], labels [B]
    const float* rowloss;                        // [B,2] per-row loss terms (k_free_energy) or null
    double* loss_acc;                            // non-null: [2] running sums of the per-minibatch means of the two loss terms
    int nL;                                      // 1: one workgroup of kind L at the very end of the grid (CD step with the
                                                 // training loop's monitor): batch means of `rowloss` -> loss_acc
    float* dgU; float* dgc;                      // gradient outputs of kind D (null = not wanted)
    float* duU; float* duc; float* dub;          // fused updates by kind D: class weights, class bias, visible bias (sparsity only)
};

__device__ __forceinline__ float upd(float th, float g, const GradArgs& a, bool sparse)
{
    float t = a.factor * g;
    t = t * a.lr;
    t = t / a.bs;
    th = th + t;
    return sparse ? th - a.sparse : th;
}

// The two phases of a gradient block are contracted SIDE BY SIDE by a 512-thread workgroup: threads 0..255 ("group 0",
// wavefronts 0..3 = K quarters of phase 0) and threads 256..511 (group 1, phase 1), each with an LDS staging region of
// its own.  Two wavefronts per SIMD: while one is parked on its operand transfers or an LDS read, the other issues MFMAs
// (one after the other, the phases left the MFMA pipe idle for more than half of the kernel).  The groups then swap their
// strips through LDS and share the epilogue.  Same arithmetic as phase after phase: g = o0 - o1 of the same two [Q] sums.
template <int MT_, int NT_>
__device__ __forceinline__ void swap_strips(const StripOut<MT_, NT_>& mine, int grp, char* lds_all, StripOut<MT_, NT_>& o0,
                                            StripOut<MT_, NT_>& o1)
{
    constexpr int NP = StripOut<MT_, NT_>::NP;
    const int t = threadIdx.x & 255;
    float4* x = reinterpret_cast<float4*>(lds_all);   // every thread has left strip_mma: the staging regions are free
#pragma unroll
    for (int p = 0; p < NP; ++p) x[(grp * NP + p) * 256 + t] = mine.sum[p];
    __syncthreads();
    const bool first = grp == 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const float4 other = x[((1 - grp) * NP + p) * 256 + t];
        const float4 own = mine.sum[p];
        o0.sum[p] = make_float4(first ? own.x : other.x, first ? own.y : other.y, first ? own.z : other.z, first ? own.w : other.w);
        o1.sum[p] = make_float4(first ? other.x : own.x, first ? other.y : own.y, first ? other.z : own.z, first ? other.w : own.w);
    }
}

// Kind L (CD-k step with the training loop's monitor): the minibatch means of the two per-row loss terms -- quartered
// chains [Q] over the rows (staged through LDS), / B -- added to the float64 running sums (src/ClassificationRBM.py:187, :191).
__device__ __forceinline__ void loss_block(const GradArgs& a, char* lds_raw)
{
    float* rl = reinterpret_cast<float*>(lds_raw);   // [tile][2], then 8 partial sums
    const int B = a.B, n = (int)blockDim.x, tid = (int)threadIdx.x;
    float* part = rl + 2 * n;
    const int q = tid & 3, k = (tid >> 2) & 1;        // threads 0..7: (quarter q, column k)
    int lo, hi;
    chain_range(B, 4, q, lo, hi);
    float acc = 0.f;
    for (int r0 = 0; r0 < B; r0 += n) {
        const int nr = min(n, B - r0);
        if (tid < nr) {
            rl[2 * tid] = a.rowloss[2 * (long)(r0 + tid)];
            rl[2 * tid + 1] = a.rowloss[2 * (long)(r0 + tid) + 1];
        }
        __syncthreads();
        if (tid < 8) {
            const int a0 = max(lo, r0) - r0, a1 = min(hi, r0 + nr) - r0;
            for (int i = a0; i < a1; ++i) acc = __builtin_fmaf(1.0f, rl[2 * i + k], acc);
        }
        __syncthreads();
    }
    if (tid < 8) part[tid] = acc;   // part[4 k + q]
    __syncthreads();
    if (tid < 2) {
        const float m = (((part[4 * tid] + part[4 * tid + 1]) + part[4 * tid + 2]) + part[4 * tid + 3]) / (float)B;
        a.loss_acc[tid] = a.loss_acc[tid] + (double)m;
    }
}

// Kind D (discriminative step, src/ClassificationRBM.py:113-115, :125, :132, :136, :138, :142): what depends on the
// class index.  Workgroup (class c, 16 hidden columns), 256 threads:
//   gU[c,j] = sum_{i: y_i = c} pos_ij - sum_i o_icj p_ic    with o_icj = sigma(pre_ij + U_cj) (pos_ij = o_icj where y_i = c)
//   gc[c]   = sum_i onehot_ic - sum_i p_ic                  (the workgroup of columns 0..15)
//   the two loss terms of the minibatch: batch means of `rowloss`, added to loss_acc (class 0, columns 0..15)
// All batch sums are quartered chains [Q]: the 256 threads evaluate o for 256 rows x 16 columns into LDS, then thread
// (quarter q, column) walks its quarter's rows in order -- fmaf(onehot, o, acc) for the first sum (bit for bit the chain of
// the [onehot]^T.pos contraction: rows of another class enter as 0 * o), fmaf(1, o * p, acc) for the second.
// Fused update of U, c and of the gradient-free visible bias (:138: grad 0, the sparsity constant still applies).
__device__ __forceinline__ void disc_class_block1(const GradArgs& a, int c, int jb, bool sweep_b, int blk, char* lds_raw)
{
    float (*o_s)[17] = reinterpret_cast<float (*)[17]>(lds_raw);
    float* pc_s = reinterpret_cast<float*>(lds_raw) + 256 * 17;
    float* oh_s = pc_s + 256;
    float* ce_s = oh_s + 256;
    float* nl_s = ce_s + 256;
    float (*part)[40] = reinterpret_cast<float (*)[40]>(nl_s + 256);   // [4 quarters][16 A | 16 B | onehot | p | ce | nll]
    const int B = a.B, H = a.H, C = a.C;
    const int tid = (int)threadIdx.x;
    const int col = tid & 15, slot = tid >> 4;
    const int j = jb * 16 + col, jj = min(j, H - 1);
    const float u = a.Up[(long)c * H + jj];
    const bool with_loss = c == 0 && jb == 0 && a.rowloss != nullptr;
    // roles in the summation stage: threads 0..63 chain A (o * p), 64..127 chain B (onehot * o), 128..143 the four side sums
    const int role = tid < 64 ? 0 : tid < 128 ? 1 : tid < 144 ? 2 + ((tid - 128) >> 2) : 6;
    const int q = role <= 1 ? (tid >> 4) & 3 : tid & 3;
    int lo, hi;
    chain_range(B, 4, q, lo, hi);
    float acc = 0.f;
    for (int r0 = 0; r0 < B; r0 += 256) {
        const int nr = min(256, B - r0);
        float pr[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) pr[t] = a.pre[(long)min(r0 + slot + 16 * t, B - 1) * H + jj];
        const int ir = min(r0 + tid, B - 1);
        const float pcv = a.pyx[(long)ir * C + c];
        const int labv = a.labels[ir];
        float cev = 0.f, nlv = 0.f;
        if (with_loss) { cev = a.rowloss[2 * (long)ir]; nlv = a.rowloss[2 * (long)ir + 1]; }
#pragma unroll
        for (int t = 0; t < 16; ++t) o_s[slot + 16 * t][col] = det_sigmoid(pr[t] + u);
        pc_s[tid] = pcv;
        oh_s[tid] = (labv == c) ? 1.f : 0.f;
        ce_s[tid] = cev;
        nl_s[tid] = nlv;
        __syncthreads();
        const int a0 = max(lo, r0) - r0, a1 = min(hi, r0 + nr) - r0;
        if (role == 0) {
            for (int i = a0; i < a1; ++i) acc = __builtin_fmaf(1.0f, o_s[i][col] * pc_s[i], acc);
        } else if (role == 1) {
            for (int i = a0; i < a1; ++i) acc = __builtin_fmaf(oh_s[i], o_s[i][col], acc);
        } else if (role < 6) {
            const float* src = role == 2 ? oh_s : role == 3 ? pc_s : role == 4 ? ce_s : nl_s;
            for (int i = a0; i < a1; ++i) acc = __builtin_fmaf(1.0f, src[i], acc);
        }
        __syncthreads();
    }
    if (role == 0) part[q][col] = acc;
    else if (role == 1) part[q][16 + col] = acc;
    else if (role < 6) part[q][32 + role - 2] = acc;
    __syncthreads();
    auto qsum = [&](int k) { return ((part[0][k] + part[1][k]) + part[2][k]) + part[3][k]; };
    if (tid < 16 && j < H) {
        const float g = qsum(16 + col) - qsum(col);
        const long off = (long)c * H + j;
        if (a.dgU) a.dgU[off] = g;
        if (a.duU) a.duU[off] = upd(a.duU[off], g, a, false);
    }
    if (jb == 0 && tid == 0) {
        const float g = qsum(32) - qsum(33);
        if (a.dgc) a.dgc[c] = g;
        if (a.duc) a.duc[c] = upd(a.duc[c], g, a, true);
        if (with_loss && a.loss_acc) {   // += (double)(fp32 batch mean): `epoch_error += batch_error.item()` (:187, :191)
            a.loss_acc[0] = a.loss_acc[0] + (double)(qsum(34) / (float)B);
            a.loss_acc[1] = a.loss_acc[1] + (double)(qsum(35) / (float)B);
        }
    }
    if (a.dub && sweep_b)
        for (int m = blk * 256 + tid; m < a.V; m += a.nD * 256) a.dub[m] = upd(a.dub[m], 0.f, a, true);
}

// Workgroup `blk` of kind D: class c, a.dW consecutive blocks of 16 hidden columns, one after the other.  (a.dW > 1 where the
// gradient launch holds ONE workgroup per CU -- fp32 inputs: 96 KB of staging -- and C * H / 16 workgroups of kind D would
// queue up behind the strips for further rounds: fewer, longer ones finish beside the strips, on the CUs the strips leave free.)
__device__ __forceinline__ void disc_class_block(const GradArgs& a, int blk, char* lds_raw)
{
    const int nJ = (a.H + 15) / 16, nJw = (nJ + a.dW - 1) / a.dW;
    const int c = blk / nJw, jb0 = (blk % nJw) * a.dW;
    for (int w = 0; w < a.dW && jb0 + w < nJ; ++w) {
        if (w) __syncthreads();   // the block before has consumed its LDS arrays
        disc_class_block1(a, c, jb0 + w, w == 0, blk, lds_raw);
    }
}

// Epilogues of the small gradient blocks (o0 / o1: the two phases' strips as the finishing thread holds them).
// kind U: rows 0..C-1 = class-weight rows, row C = hidden bias; 16*NT hidden columns from n0 on
template <class SOT>
__device__ __forceinline__ void grad_epi_u(const GradArgs& a, int n0, const SOT& o0, const SOT& o1)
{
#pragma unroll
    for (int op = 0; op < SOT::NP; ++op) {
        if (!o0.valid(op) || o0.row(op) > a.C) continue;
        const int row = o0.row(op);
        const float g[4] = {o0.sum[op].x - o1.sum[op].x, o0.sum[op].y - o1.sum[op].y, o0.sum[op].z - o1.sum[op].z,
                            o0.sum[op].w - o1.sum[op].w};
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + 4 * o0.cg(op) + e;
            if (n >= a.H) break;
            if (row < a.C) {
                const long off = (long)row * a.H + n;
                const float gu = a.gUsub ? g[e] - a.gUsub[off] : g[e];
                if (a.gU) a.gU[off] = gu;
                if (a.uU) a.uU[off] = upd(a.uU[off], gu, a, false);
            } else {
                if (a.gd) a.gd[n] = g[e];
                if (a.ud) a.ud[n] = upd(a.ud[n], g[e], a, true);
            }
        }
    }
}
// kind B: visible bias, rows m0 .. of the data's column sums (output column 0)
template <class SOT>
__device__ __forceinline__ void grad_epi_b(const GradArgs& a, int m0, const SOT& o0, const SOT& o1)
{
#pragma unroll
    for (int op = 0; op < SOT::NP; ++op) {
        if (!o0.valid(op) || o0.cg(op) != 0) continue;
        const int m = m0 + o0.row(op);
        if (m >= a.V) continue;
        const float g = o0.sum[op].x - o1.sum[op].x;
        if (a.gb) a.gb[m] = g;
        if (a.ub) a.ub[m] = upd(a.ub[m], g, a, true);
    }
}
// kind C: class bias, classes n0 .. (output row 0)
template <class SOT>
__device__ __forceinline__ void grad_epi_c(const GradArgs& a, int n0, const SOT& o0, const SOT& o1)
{
#pragma unroll
    for (int op = 0; op < SOT::NP; ++op) {
        if (!o0.valid(op) || o0.row(op) != 0) continue;
        const float g[4] = {o0.sum[op].x - o1.sum[op].x, o0.sum[op].y - o1.sum[op].y, o0.sum[op].z - o1.sum[op].z,
                            o0.sum[op].w - o1.sum[op].w};
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + 4 * o0.cg(op) + e;
            if (n >= a.C) break;
            if (a.gc) a.gc[n] = g[e];
            if (a.uc) a.uc[n] = upd(a.uc[n], g[e], a, true);
        }
    }
}

// One output of a W strip (thread's output `op` of StripOut SO at strip origin (m0, n0)): the gradient, and the fused update
// on the W tile `w_old` requested before the contraction.
template <class SO>
__device__ __forceinline__ void grad_finish_w(const GradArgs& a, int m0, int n0, int op, const float4& s0, const float4& s1,
                                              const float4& w_old)
{
    const int m = m0 + SO::row(op), nb = n0 + 4 * SO::cg(op);
    if (!SO::valid(op) || m >= a.V || nb >= a.H) return;
    const float g[4] = {s0.x - s1.x, s0.y - s1.y, s0.z - s1.z, s0.w - s1.w};
    const long off = (long)m * a.H + nb;
    if (a.vecO && nb + 3 < a.H) {
        if (a.gW) *reinterpret_cast<float4*>(a.gW + off) = make_float4(g[0], g[1], g[2], g[3]);
        if (a.W) {
            float4 w = w_old;
            w.x = upd(w.x, g[0], a, false); w.y = upd(w.y, g[1], a, false);
            w.z = upd(w.z, g[2], a, false); w.w = upd(w.w, g[3], a, false);
            *reinterpret_cast<float4*>(a.W + off) = w;
        }
        return;
    }
    for (int e = 0; e < 4 && nb + e < a.H; ++e) {
        if (a.gW) a.gW[off + e] = g[e];
        if (a.W) a.W[off + e] = upd(a.W[off + e], g[e], a, false);
    }
}

// fused update: the W tile of output `op` (requested behind the first operand transfers)
template <class SO>
__device__ __forceinline__ float4 grad_w_tile(const GradArgs& a, int m0, int n0, int op)
{
    const int m = min(m0 + SO::row(op), a.V - 1), nb = n0 + 4 * SO::cg(op);
    return (a.W && a.vecO && nb + 3 < a.H) ? *reinterpret_cast<const float4*>(a.W + (long)m * a.H + nb) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// The (16 MT) x (16 NT) strip of gW at (m0, n0), both phases side by side (uint8 data, 512 threads: see below).
// The two groups share the epilogue.  NP == 2 outputs per thread: output 0 of every thread is finished by group 0, output 1 by
// group 1.  NP == 1 (a strip of 256 float4 outputs): thread t's output is finished by group t / 128 -- half of each group's
// threads finish, the others only hand their partial over.
template <int MT, int NT, bool FAST>
__device__ __forceinline__ void grad_w_strip_par(const GradArgs& a, int m0, int n0, char* lds_all, int grp)
{
    using SO = StripOut<MT, NT>;
    static_assert(SO::NP == 2 || SO::NP == 1, "the two groups share the epilogue");
    char* lds = lds_all + (size_t)grp * a.lds_group;
    const bool first = grp == 0;
    // ONE instantiation of the contraction for both groups: the phase's operands are picked by scalar selects (two calls --
    // one per phase behind a branch -- kept the views, descriptors and walks of BOTH phases live across the branch: 21 spilled
    // SGPRs in the benchmark's variant, and twice the code for the instruction cache of a cold start)
    const ViewCContig Pm{first ? a.P0 : a.P1, a.H, a.H, a.B};
    const ViewCContigU8 Dm{reinterpret_cast<const uint8_t*>(first ? a.A0 : a.A1), a.V, a.V, a.B};
    const int t = (int)threadIdx.x & 255;
    const bool mine_to_finish = SO::NP == 2 || (t >> 7) == grp;      // (NP == 1) do I finish my output, or hand it over?
    constexpr int kOp1 = SO::NP == 2 ? 1 : 0;                        // group 1's output among a thread's
    float4 w_old = make_float4(0.f, 0.f, 0.f, 0.f);
    const int my_op = first ? 0 : kOp1;
    auto pre0 = [&]() {
        if (mine_to_finish) w_old = grad_w_tile<SO>(a, m0, n0, my_op);
    };
    QBM_STAMP(11);
    // (piece waits: a chain of a minibatch of 256 rows is ONE quad -- strip_mma's PW)
    const SO mine = strip_mma<MT, NT, FAST, FAST, 4, 4, false, 0, 0, kWhole, 0, true>(Dm, Pm, m0, n0, a.B, lds, a.KP, pre0, NoPre{}, first ? 0 : 5);
    float4* x = reinterpret_cast<float4*>(lds_all);   // every thread has left strip_mma: the staging regions are free
    if constexpr (SO::NP == 2) {
        // group 0 finishes output 0 and needs phase 1's output 0; group 1 finishes output 1 and needs phase 0's output 1
        const float4 keep = first ? mine.sum[0] : mine.sum[1], give = first ? mine.sum[1] : mine.sum[0];
        x[grp * 256 + t] = give;
        __syncthreads();
        const float4 got = x[(1 - grp) * 256 + t];
        if (first) grad_finish_w<SO>(a, m0, n0, 0, keep, got, w_old);
        else grad_finish_w<SO>(a, m0, n0, 1, got, keep, w_old);
    } else {
        if (!mine_to_finish) x[t] = mine.sum[0];      // (slot t is written by exactly one of the two threads (0, t), (1, t))
        __syncthreads();
        if (mine_to_finish) {
            const float4 got = x[t];
            if (first) grad_finish_w<SO>(a, m0, n0, 0, mine.sum[0], got, w_old);
            else grad_finish_w<SO>(a, m0, n0, 0, got, mine.sum[0], w_old);
        }
    }
}

// PAR (both data operands uint8: the Gibbs states of the CD-k / PCD step): 512 threads, the phases side by side.
// Otherwise (fp32 data: twice the staging bytes, two regions do not fit the LDS): 256 threads, phase after phase.
// ONE (the discriminative step): phase 0 only -- gW = A0^T.P0 -- and the kind D workgroups behind the strips; 256 threads.
// MT x NT: the W strips' tiles (64 x 32, 64 x 16 or 112 x 16: see G_MT); UNT: tiles of 16 hidden columns per workgroup of kind U
// (and of 16 classes per workgroup of kind C) -- NT, or 2 beside the 112 x 16 strips so that the whole launch stays one round
template <class T0, class T1, bool FAST, bool PAR, bool ONE, int NT, int MT = G_MT>
__global__ __launch_bounds__(PAR ? 512 : 256) void k_grad(GradArgs a)
{
    static_assert(!(PAR && ONE), "the one-phase form takes the 256-thread kernel");
    constexpr int UNT = MT == G_MT7 ? 2 : NT;
    extern __shared__ __attribute__((aligned(16))) char lds_all[];
    QBM_STAMP(10);
    asm volatile("" ::"s"(a.A0), "s"(a.P0), "s"(a.A1), "s"(a.P1), "s"(a.gW), "s"(a.W), "s"(a.B), "s"(a.V), "s"(a.H), "s"(a.C),
                 "s"(a.KP), "s"(a.nU), "s"(a.nB), "s"(a.nC), "s"(a.lds_group), "s"(a.nL), "s"(a.nD), "s"(a.inv_nm), "s"(a.status),
                 "s"(a.n_rows));   // the arguments of the W workgroups in one scalar-cache round trip
    const int grp = PAR ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;   // 0: phase-0 operands, 1: phase-1 operands
    char* lds = lds_all + (size_t)grp * a.lds_group;
    int blk = blockIdx.x;
    if (a.status || a.n_rows) {   // (neither on the launch-per-phase single-GPU path: no load, no branch taken)
        const uint32_t st = a.status ? *a.status : 0u;   // written by an earlier launch: a plain (scalar) load
        if (blk == 0 && threadIdx.x == 0 && a.n_rows) *a.n_rows = st ? -__builtin_inff() : (float)a.B;
        if (st) return;
    }
    const ViewCContig P0{a.P0, a.H, a.H, a.B}, P1{a.P1, a.H, a.H, a.B};
    const typename CView<T0>::type D0{reinterpret_cast<const T0*>(a.A0), a.V, a.V, a.B};
    const typename CView<T1>::type D1{reinterpret_cast<const T1*>(a.A1), a.V, a.V, a.B};
    // both phases of one gradient block: o0, o1 as every finishing thread needs them (PAR: only group 0 finishes the
    // small blocks, group 1 leaves)
#define QBM_TWO_PHASES(SOT, CALL0, CALL1)        \
    SOT o0, o1;                                  \
    if constexpr (PAR) {                         \
        SOT mine;                                \
        if (grp == 0) mine = CALL0;              \
        else mine = CALL1;                       \
        swap_strips(mine, grp, lds_all, o0, o1); \
        if (grp != 0) return;                    \
    } else {                                     \
        o0 = CALL0;                              \
        if constexpr (!ONE) o1 = CALL1;          \
        else                                     \
            for (int op = 0; op < SOT::NP; ++op) o1.sum[op] = make_float4(0.f, 0.f, 0.f, 0.f);   \
    }
    if (blk < a.nU) {  // class-weight rows and the hidden-bias row (C + 1 <= 16 rows: one row tile) x 16*UNT hidden columns
        const int n0 = blk * 16 * UNT;
        const ViewAugY Y0{a.Y0, a.C}, Y1{a.Y1, a.C};
        using SOT = StripOut<1, UNT>;
        QBM_TWO_PHASES(SOT, (strip_mma<1, UNT, false, FAST>(Y0, P0, 0, n0, a.B, lds, a.KP)),
                       (strip_mma<1, UNT, false, FAST>(Y1, P1, 0, n0, a.B, lds, a.KP)))
        grad_epi_u(a, n0, o0, o1);
        return;
    }
    blk -= a.nU;
    if (blk < a.nB) {  // visible bias: column sums of the data = data^T . ones, 16*MT visible units per workgroup
        const int m0 = blk * 16 * MT;
        const ViewOnesCol ones;
        using SOT = StripOut<MT, 1>;
        QBM_TWO_PHASES(SOT, (strip_mma<MT, 1, FAST, false>(D0, ones, m0, 0, a.B, lds, a.KP)),
                       (strip_mma<MT, 1, FAST, false>(D1, ones, m0, 0, a.B, lds, a.KP)))
        grad_epi_b(a, m0, o0, o1);
        return;
    }
    blk -= a.nB;
    if (blk < a.nC) {  // class bias: column sums of the labels = ones^T . Y, 16*UNT classes per workgroup
        const int n0 = blk * 16 * UNT;
        const ViewOnesRow ones;
        const ViewCContig Y0{a.Y0, a.C, a.C, a.B}, Y1{a.Y1, a.C, a.C, a.B};
        using SOT = StripOut<1, UNT>;
        QBM_TWO_PHASES(SOT, (strip_mma<1, UNT, false, false>(ones, Y0, 0, n0, a.B, lds, a.KP)),
                       (strip_mma<1, UNT, false, false>(ones, Y1, 0, n0, a.B, lds, a.KP)))
        grad_epi_c(a, n0, o0, o1);
        return;
    }
    blk -= a.nC;   // index among the strips (and what stands behind them)
#undef QBM_TWO_PHASES
    const int nM = (a.V + 16 * MT - 1) / (16 * MT), nN = (a.H + 16 * NT - 1) / (16 * NT);
    if constexpr (ONE) {
        if (blk >= nM * nN) {   // kind D workgroups stand behind the strips
            disc_class_block(a, blk - nM * nN, lds_all);
            return;
        }
    } else {
        if (a.nL && blk == nM * nN) {   // the last workgroup of the grid (every thread of it comes here)
            loss_block(a, lds_all);
            return;
        }
    }
    int ms, ns;
    strip_coords<true>(nM, nN, ms, ns, a.nU + a.nB + a.nC, a.inv_nm, NT == 1 && (nN & 15) == 0);   // (k_hidden's pairs: H % 256 == 0)
    const int m0 = ms * 16 * MT, n0 = ns * 16 * NT;
    using SO = StripOut<MT, NT>;
    if constexpr (PAR) {
        grad_w_strip_par<MT, NT, FAST>(a, m0, n0, lds_all, grp);
    } else {
        float4 w_old[SO::NP];
        auto pre0 = [&]() {
#pragma unroll
            for (int op = 0; op < SO::NP; ++op) w_old[op] = grad_w_tile<SO>(a, m0, n0, op);
        };
        QBM_STAMP(11);
        const SO o0 = strip_mma<MT, NT, FAST, FAST, 4, 4, false, 0, 0, kWhole, 0, true>(D0, P0, m0, n0, a.B, lds, a.KP, pre0);
        SO o1;   // ONE (the discriminative step: gW = x^T.(pos - neg)): zeros, g = o0 - 0 = o0
        if constexpr (!ONE) o1 = strip_mma<MT, NT, FAST, FAST, 4, 4, false, 0, 0, kWhole, 0, true>(D1, P1, m0, n0, a.B, lds, a.KP, NoPre{}, NoPre{}, 5);
        else
#pragma unroll
            for (int op = 0; op < SO::NP; ++op) o1.sum[op] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int op = 0; op < SO::NP; ++op) grad_finish_w<SO>(a, m0, n0, op, o0.sum[op], o1.sum[op], w_old[op]);
    }
    QBM_STAMP(12);
}

// ---------------------------------------------------------------------------------------------
// A-7 fused tail: the LAST hidden phase of a CD-k / PCD-k step, its two-phase gradient and the parameter update in ONE
// launch (the launch-per-phase step of CD-1 is then three launches: k_hidden, k_visible, k_hgrad).  OPT-IN
// (qbm_rbm_fused_chain_mode(3)): bit-identical, but measured SLOWER than k_hidden + k_grad on every shape tried (784 x 512,
// B = 256: 34.5 us per CD-1 step against 30.2; DESIGN.md section 10 has the in-kernel stamps that say why).
//
// sample_hidden's grid is (row strip ms of 32 rows) x (column strip ns of 16 hidden units), and strip_coords deals the
// workgroups of one ns to ONE XCD.  The gradient of those 16 hidden columns -- gW[:, 16 ns ..], gU[:, 16 ns ..],
// gd[16 ns ..] -- contracts over ALL rows of the minibatch, i.e. over exactly what the nM workgroups of group ns have just
// produced (p(h_k)[:, 16 ns ..]) and what earlier launches wrote (v_0, p(h_0), v_k, y_k).  So the group goes on behind an
// XCD-local barrier (group_barrier: plain stores + sc1 polling, 0.3 us; the p(h_k) panel is then read with sc1 transfers
// from the L2 its producers wrote through) instead of behind a launch boundary (1.5-1.9 us, a cold scalar cache, cold
// operand caches), and all 256 CUs share the gradient (k_grad's 64 x 32 strips fill 208).  Work items of a group: the W
// strips of 128 visible rows x 16 columns (MT = 8, NT = 1; both phases side by side as in k_grad: grad_w_strip_par) and
// the [class rows | hidden bias] tile, dealt round-robin to its nM workgroups; the visible-bias strips (32 rows each), the
// class bias and the training loop's loss terms are dealt over the groups.  Same chains as k_grad ([Q], NQ = 4 over the
// rows): the same bits.
//
// FAIL-SAFE.  The barrier's spin is bounded and the XCD placement is checked (as in k_chain); a workgroup that gives up
// records status 3 (barrier) / 4 (placement), the step's draw offset and the launch's round in sync[512..515] and leaves.
// Other groups may already have applied their part of the update, so a step is never REPEATED from here: per workgroup
// the launch records what it completed -- hdone[blk] = round behind its hidden phase, gdone[blk] = round behind its
// gradient items -- every later launch on the workspace (k_hidden, k_visible, the monitor, k_hgrad, k_chain, k_grad,
// k_update) stores nothing while the status word is set, and the host FINISHES the step (qbm_rbm_cd_step_finish: mode 1 =
// the hidden phases not done, then mode 2 = the gradient items not done, each a launch of its own, no barrier needed) before
// it repeats the later ones on the launch-per-phase path (rbm_ops.HipRbmOps.check_chain_status).  Columns of a group that did
// not get through its barrier have not been touched, so the repeated hidden phases and items see the parameters the
// original launch saw: the finished step has the bits of an undisturbed one.
// ---------------------------------------------------------------------------------------------
constexpr int kHgRound = 516;              // round of the next / current fused tail (sync[512..515]: status, offset, failed round)
constexpr int kHgFlags = 576;              // [32 groups][32] round flags (one 128-byte line per group)
constexpr int kHgIds = kHgFlags + 1024;    // [32][32] XCC ids
constexpr int kHgHdone = kHgIds + 1024;    // [256] per workgroup: round of its last completed hidden phase
constexpr int kHgGdone = kHgHdone + 256;   // [256] per workgroup: round of its last completed gradient items
static_assert(kHgGdone + 256 == QBM_CD_SYNC_WORDS, "layout of qbm_rbm_cd_workspace.sync");
constexpr int HG_MT = 4, HG_BMT = 2;       // a W item: two strips of 64 rows (one per thread group); visible-bias strips of 32 rows
constexpr int HG_ROWS = 2 * 16 * HG_MT;    // visible rows of a W item

struct HGradArgs {
    HiddenArgs h;        // the hidden phase: v = v_k (or the persistent chains), y = y_k, p = p(h_k); no sample
    GradArgs g;          // A0 = v_0, Y0, P0 = p(h_0), A1 = h.v, Y1 = h.y, P1 = h.p; nB / nC != 0: gb / gc wanted; nL: loss terms
    uint32_t* sync;      // qbm_rbm_cd_workspace.sync
    uint64_t offset;     // draw offset of the step (recorded when the launch gives up)
    int mode;            // 0 = the step; 1 / 2 = finishing a step that gave up: its hidden phases / gradient items not done
    uint32_t round;      // modes 1, 2: the round of the launch that gave up; mode 0: non-zero = the round to use (qbm_rbm_cd_tail)
    int fault_wg, fault_kind;   // test hook (qbm_test_hgrad_fault): workgroup fault_wg leaves without a trace at its start (1)
                                // or gives up in front of the barrier (2)
};

// kinds U / B / C of the gradient, both phases side by side (the 512-thread organisation of grad_w_strip_par)
template <int NT, int AUX1>
__device__ __forceinline__ void grad_u_item_par(const GradArgs& a, int n0, char* lds_all, int grp)
{
    char* lds = lds_all + (size_t)grp * a.lds_group;
    const ViewCContig P0{a.P0, a.H, a.H, a.B}, P1{a.P1, a.H, a.H, a.B};
    const ViewAugY Y0{a.Y0, a.C}, Y1{a.Y1, a.C};
    using SOT = StripOut<1, NT>;
    SOT mine, o0, o1;
    if (grp == 0) mine = strip_mma<1, NT, false, true>(Y0, P0, 0, n0, a.B, lds, a.KP);
    else mine = strip_mma<1, NT, false, true, 4, 4, false, 0, AUX1>(Y1, P1, 0, n0, a.B, lds, a.KP);
    swap_strips(mine, grp, lds_all, o0, o1);
    if (grp == 0) grad_epi_u(a, n0, o0, o1);
}
template <int MT>
__device__ __forceinline__ void grad_b_item_par(const GradArgs& a, int m0, char* lds_all, int grp)
{
    char* lds = lds_all + (size_t)grp * a.lds_group;
    const ViewCContigU8 D0{reinterpret_cast<const uint8_t*>(a.A0), a.V, a.V, a.B}, D1{reinterpret_cast<const uint8_t*>(a.A1), a.V, a.V, a.B};
    const ViewOnesCol ones;
    using SOT = StripOut<MT, 1>;
    SOT mine, o0, o1;
    if (grp == 0) mine = strip_mma<MT, 1, true, false>(D0, ones, m0, 0, a.B, lds, a.KP);
    else mine = strip_mma<MT, 1, true, false>(D1, ones, m0, 0, a.B, lds, a.KP);
    swap_strips(mine, grp, lds_all, o0, o1);
    if (grp == 0) grad_epi_b(a, m0, o0, o1);
}
__device__ __forceinline__ void grad_c_item_par(const GradArgs& a, char* lds_all, int grp)
{
    char* lds = lds_all + (size_t)grp * a.lds_group;
    const ViewOnesRow ones;
    const ViewCContig Y0{a.Y0, a.C, a.C, a.B}, Y1{a.Y1, a.C, a.C, a.B};
    using SOT = StripOut<1, 1>;
    SOT mine, o0, o1;
    if (grp == 0) mine = strip_mma<1, 1, false, false>(ones, Y0, 0, 0, a.B, lds, a.KP);
    else mine = strip_mma<1, 1, false, false>(ones, Y1, 0, 0, a.B, lds, a.KP);
    swap_strips(mine, grp, lds_all, o0, o1);
    if (grp == 0) grad_epi_c(a, 0, o0, o1);
}

// The W item of k_hgrad: visible rows [m0w, m0w + 128) x the group's 16 hidden columns.  Thread group g (wavefronts 4g ..
// 4g+3 = the K chains) takes rows m0w + 64 g .. and contracts BOTH phases one after the other -- o0, o1 of an output end up in
// the same thread: no exchange, and what a wavefront keeps live is a 64 x 16 strip (a 128 x 16 strip per wavefront with the
// phases side by side needs more registers than there are).  The two thread groups contract against the same P panel: it is
// staged once, by group 0, into a region both read (strip_mma: ldsB / b_resident).
// PH (strip_mma's two-call form): kIssueAB = the requests of phase 0 (+ the W tile); kIssueA = the data operand of phase 1;
// kConsume = group 0 requests p(h_k), then both contractions and the epilogue; kWhole = everything here.
// LDS (bytes from `lds`): phase 1 at 0: [A group 0 | A group 1 | B]; phase 0 at a.lds_group0, same layout.
template <int PH>
__device__ __forceinline__ void hg_w_item(const GradArgs& a, int m0w, int n0, char* lds, int grp, float4& w_old)
{
    constexpr int MT = HG_MT, NT = H_NT;
    using SO = StripOut<MT, NT>;
    static_assert(SO::NP == 1, "one output per thread");
    const int m0 = m0w + grp * 16 * MT;
    const ViewCContig P0{a.P0, a.H, a.H, a.B}, P1{a.P1, a.H, a.H, a.B};
    const ViewCContigU8 D0{reinterpret_cast<const uint8_t*>(a.A0), a.V, a.V, a.B}, D1{reinterpret_cast<const uint8_t*>(a.A1), a.V, a.V, a.B};
    const int wa = 4 * strip_wave_bytes_a<MT, NT, 1>(a.KP);          // the A tiles (later the partial strips) of one thread group
    char* a1 = lds + grp * wa;
    char* b1 = lds + 2 * wa;
    char* a0 = a1 + a.lds_group0;
    char* b0 = b1 + a.lds_group0;
    const bool stage_b = grp == 0;
    auto pre0 = [&]() { w_old = grad_w_tile<SO>(a, m0, n0, 0); };
    if constexpr (PH == kIssueAB) {
        strip_mma<MT, NT, true, true, 4, 4, false, 0, 0, kIssueAB>(D0, P0, m0, n0, a.B, a0, a.KP, pre0, NoPre{}, 8, b0, !stage_b);
        return;
    } else if constexpr (PH == kIssueA) {
        strip_mma<MT, NT, true, true, 4, 4, false, 0, kSc1, kIssueA>(D1, P1, m0, n0, a.B, a1, a.KP, NoPre{}, NoPre{}, 8, b1, true);
        return;
    }
    SO o0, o1;
    if constexpr (PH == kConsume) {
        int younger = 0;
        if (stage_b) {
            strip_mma<MT, NT, true, true, 4, 4, false, 0, kSc1, kIssueB>(D1, P1, m0, n0, a.B, a1, a.KP, NoPre{}, NoPre{}, 8, b1, false);
            younger = chain_transfers_b<NT>(a.B, a.KP);
        }
        o0 = strip_mma<MT, NT, true, true, 4, 4, false, 0, 0, kConsume>(D0, P0, m0, n0, a.B, a0, a.KP, NoPre{}, NoPre{}, 8, b0, true, NoPre{}, younger);
        // (the combine's barriers have drained group 0's requests of p(h_k) and ordered them for group 1)
        o1 = strip_mma<MT, NT, true, true, 4, 4, false, 0, kSc1, kConsume>(D1, P1, m0, n0, a.B, a1, a.KP, NoPre{}, NoPre{}, 0, b1, true);
    } else {
        char* r = lds + (size_t)grp * strip_lds_bytes<MT, NT, 1>(a.KP);   // everything in one go: a region per thread group
        o0 = strip_mma<MT, NT, true, true>(D0, P0, m0, n0, a.B, r, a.KP, pre0, NoPre{}, 8);
        o1 = strip_mma<MT, NT, true, true, 4, 4, false, 0, kSc1>(D1, P1, m0, n0, a.B, r, a.KP, NoPre{}, NoPre{}, 8);
    }
    grad_finish_w<SO>(a, m0, n0, 0, o0.sum[0], o1.sum[0], w_old);
}

// Barrier among the workgroups of a column group + the placement check: 0 = through, else the status code to record
__device__ __forceinline__ uint32_t hgrad_barrier(uint32_t* flags, const uint32_t* ids, int ms, int nM, uint32_t round, uint32_t my_xcc,
                                                  uint32_t* s_ok)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's stores have reached the L2 (and its transfers the LDS)
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (lane == 0) __hip_atomic_store(&flags[ms], round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint32_t code = 0u;
        for (int spins = 0;; ++spins) {
            const uint32_t f = lane < nM ? load_sc1(&flags[lane]) : round;
            if (__all((int)(f - round) >= 0)) break;
            if (spins > kChainSpin) { code = 3u; break; }
        }
        if (code == 0u) {   // all workgroups of the group on one XCD?  (their stores reached MY L2 only then)
            const uint32_t id = lane < nM ? load_sc1(&ids[lane]) : my_xcc;
            if (!__all((int)(id == my_xcc))) code = 4u;
        }
        if (lane == 0) *s_ok = code;
    }
    __syncthreads();
    return *s_ok;
}

template <int CM>
__global__ __launch_bounds__(512) void k_hgrad(HGradArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int nM = (a.h.B + 16 * H_MT - 1) / (16 * H_MT), nN = (a.h.H + 16 * H_NT - 1) / (16 * H_NT);
    int ms, ns;
    strip_coords(nM, nN, ms, ns);
    uint32_t* flags = a.sync + kHgFlags + ns * 32;
    uint32_t* ids = a.sync + kHgIds + ns * 32;
    uint32_t* status = a.sync + 512;
    uint32_t* hdone = a.sync + kHgHdone + blockIdx.x;
    uint32_t* gdone = a.sync + kHgGdone + blockIdx.x;
    const bool faulty = a.fault_wg == (int)blockIdx.x;
    const GradArgs& ga = a.g;
    const int grp = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);   // 0: phase-0 operands, 1: phase-1 operands
    const int n0 = ns * 16 * H_NT;
    const int nS = (ga.V + HG_ROWS - 1) / HG_ROWS;
    // The workgroup's first item, if it is a W item, goes in several calls around the barrier: the operands of phase 0
    // (written by earlier launches) are requested between the hidden phase's contraction and its combine, into the LDS the
    // hidden phase leaves alone; the data operand of phase 1 behind the hidden phase's epilogue (into its dead staging
    // area); both have landed when the barrier's vmcnt(0) is through.  Only p(h_k) -- 4 transfers per wavefront of thread
    // group 0 -- is requested behind the barrier, and lands under the MFMAs of phase 0.
    const bool split = a.mode == 0 && ms < nS;
    uint32_t* s_ok = reinterpret_cast<uint32_t*>(lds + ga.lds_sok);   // (free whenever the barrier runs)
    float4 w_old = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t round = a.round;
    if (a.mode == 0) {
        const uint32_t st = load_sc1(status);
        if (blockIdx.x == 0 && threadIdx.x == 0 && a.g.n_rows) *a.g.n_rows = st ? -__builtin_inff() : (float)a.g.B;
        if (st != 0u) return;              // an earlier fused launch on this workspace gave up: nothing until the host has looked
        if (faulty && a.fault_kind == 1) return;
        // advanced by the launch in front (VisibleArgs::hg_round), or the caller's (qbm_rbm_cd_tail): the same in every workgroup
        round = a.round ? a.round : a.sync[kHgRound];
        const uint32_t my_xcc = xcc_id();
        if (threadIdx.x == 0) ids[ms] = my_xcc;
        QBM_STAMP(5);
        auto post = [&]() {
            if (split) hg_w_item<kIssueAB>(ga, ms * HG_ROWS, n0, lds, grp, w_old);
        };
        hidden_body<uint8_t, 2, 0, CM>(a.h, ms, ns, lds, nullptr, false, post);
        if (split) hg_w_item<kIssueA>(ga, ms * HG_ROWS, n0, lds, grp, w_old);
        if (threadIdx.x == 0) *hdone = round;
        auto leave = [&](uint32_t code) {  // this workgroup gives up (thread 0 records it)
            if (threadIdx.x == 0) {
                chain_give_up(status, code, a.offset, round);
                if (a.g.n_rows) *a.g.n_rows = -__builtin_inff();   // data-parallel path: every rank skips the update of this step
            }
        };
        if (faulty && a.fault_kind == 2) {   // test hook: as if the barrier had timed out here (the partners get their flag)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (threadIdx.x == 0) __hip_atomic_store(&flags[ms], round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            leave(3u);
            return;
        }
        QBM_STAMP(6);
        const uint32_t code = hgrad_barrier(flags, ids, ms, nM, round, my_xcc, s_ok);
        QBM_STAMP(7);
        if (code != 0u) {
            leave(code);
            return;
        }
    } else if (a.mode == 1) {              // finishing: the hidden phases the launch of round `round` did not complete
        if (*hdone == round) return;
        hidden_body<uint8_t, 2, 0, CM>(a.h, ms, ns, lds);
        if (threadIdx.x == 0) *hdone = round;
        return;
    } else {                               // finishing: the gradient items it did not complete
        if (*gdone == round) return;
    }

    // ---- gradient items of this workgroup (every thread of the workgroup takes the same path through them)
    for (int i = ms; i <= nS; i += nM) {   // the group's items: W items 0 .. nS-1, then the [class rows | hidden bias] tile
        if (i < nS) {
            if (split && i == ms) hg_w_item<kConsume>(ga, i * HG_ROWS, n0, lds, grp, w_old);
            else hg_w_item<kWhole>(ga, i * HG_ROWS, n0, lds, grp, w_old);
        } else {
            grad_u_item_par<H_NT, kSc1>(ga, n0, lds, grp);
        }
        __syncthreads();                   // the strips exchanged through LDS are consumed: the next item may stage
    }
    // items dealt over the groups: visible-bias strips, class bias, loss terms -- to the workgroups with the fewest items
    const int nBs = ga.nB ? (ga.V + 16 * HG_BMT - 1) / (16 * HG_BMT) : 0, nG = nBs + (ga.nC ? 1 : 0) + (ga.nL ? 1 : 0);
    for (int t = ns; t < nG; t += nN) {
        if (ms != nM - 1 - (t / nN) % nM) continue;
        if (t < nBs) grad_b_item_par<HG_BMT>(ga, t * 16 * HG_BMT, lds, grp);
        else if (ga.nC && t == nBs) grad_c_item_par(ga, lds, grp);
        else loss_block(ga, lds);
        __syncthreads();
    }
    if (threadIdx.x == 0) *gdone = round;
    QBM_STAMP(13);
}

// ---------------------------------------------------------------------------------------------
// A-6  update_weights (src/ClassificationRBM.py:88-99): theta += factor*grad*lr/batch_size, evaluated
// left to right; then b, d, c -= sparse_constant.  One launch over the concatenated parameter blocks.
// ---------------------------------------------------------------------------------------------
struct UpdateArgs {
    float* W; float* b; float* d; float* U; float* c;
    const float* gW; const float* gb; const float* gd; const float* gU; const float* gc;
    int V, H, C;
    float lr, factor, bs, sparse;
    long first;  // first element of the concatenated range to touch (V*H when W is already updated)
    const float* bs_dev;      // non-null: the batch size is read from device memory (the all-reduced row count of the
                              // data-parallel step); not a positive finite number (a rank's fused chain gave up: -inf) = no-op
    const uint32_t* status;   // non-null: status word of a fused chain launch; non-zero = no-op
};

__global__ __launch_bounds__(256) void k_update(UpdateArgs a)
{
    const long nW = (long)a.V * a.H, nU = (long)a.C * a.H;
    const long total = nW + a.V + a.H + nU + a.C;
    if (a.status && *a.status != 0u) return;
    if (a.bs_dev) {
        a.bs = *a.bs_dev;
        if (!(a.bs > 0.f) || a.bs > 3.0e38f) return;
    }
    // Which workgroup takes which 128-byte line (a speed-only permutation inside every 8 workgroups x 8 lines): workgroup b sits
    // on XCD b % 8 and takes the lines l = b % 8 (mod 8) -- with H a multiple of 32 that is W's 32-column blocks j = b % 8 (mod 8),
    // the ones k_hidden's workgroups on that XCD read next (the fused update inside k_grad has the same owner)
    const bool perm = (gridDim.x & 7) == 0 && blockDim.x == 256;
    const int t = (int)threadIdx.x, bx = (int)blockIdx.x;
    const long id0 = perm ? (((long)(bx >> 3) * 64 + 8 * (t >> 5) + (bx & 7)) * 32 + (t & 31)) : (long)bx * blockDim.x + t;
    for (long i = a.first + id0; i < total; i += (long)gridDim.x * blockDim.x) {
        float* th; const float* g; long k = i; bool sp = false;
        if (k < nW) { th = a.W; g = a.gW; }
        else if ((k -= nW) < a.V) { th = a.b; g = a.gb; sp = true; }
        else if ((k -= a.V) < a.H) { th = a.d; g = a.gd; sp = true; }
        else if ((k -= a.H) < nU) { th = a.U; g = a.gU; }
        else { k -= nU; th = a.c; g = a.gc; sp = true; }
        const float gi = g ? g[k] : 0.f;   // reference: visible_bias_grad = 0 (:138)
        float t = a.factor * gi;
        t = t * a.lr;
        t = t / a.bs;
        float v = th[k] + t;
        if (sp) v = v - a.sparse;
        th[k] = v;
    }
}

// test hook: det_* functions, bit-compared with the oracle
__global__ void k_math(int which, const float* __restrict__ x, float* __restrict__ y, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    y[i] = which == 0 ? det_expf(v) : which == 1 ? det_logf(v) : which == 2 ? det_sigmoid(v) : det_softplus(v);
}

__global__ void k_uniform(uint64_t seed, uint64_t offset, uint32_t row0, int rows, int cols, float* __restrict__ out)
{
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cgs = (cols + 3) / 4;
    if (idx >= (long)rows * cgs) return;
    const int i = (int)(idx / cgs), cg = (int)(idx % cgs);
    const f32x4 u = uniform4(seed, offset, row0 + (uint32_t)i, (uint32_t)cg);
    for (int e = 0; e < 4 && 4 * cg + e < cols; ++e) out[(long)i * cols + 4 * cg + e] = u[e];
}

// fp32 {0,1} (or any value in 0..255) -> uint8, for callers that hold binary data as floats
__global__ void k_f32_to_u8(const float* __restrict__ x, uint8_t* __restrict__ y, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (uint8_t)x[i];
}

// ---- launch helpers ---------------------------------------------------------------------------
template <int MT, int NT, int ASZ>
int panel_for(int K)
{
    const int Kq = quarter_len(K);
    const int KP = (Kq + 63) / 64 * 64;
    const int mx = strip_max_panel<MT, NT, ASZ>();
    return KP < mx ? KP : mx;
}

template <class Kern>
int set_lds(Kern k, size_t bytes)
{
    if (bytes > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)bytes);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
    }
    return QBM_OK;
}

int check_params(const qbm_rbm_params* P)
{
    if (!P) return fail(QBM_ERR_ARG, "params is null");
    if (P->V <= 0 || P->H <= 0 || P->C <= 0) return fail(QBM_ERR_ARG, "V, H, C must be positive");
    if (P->C > QBM_MAX_CLASSES) return fail(QBM_ERR_ARG, "C exceeds QBM_MAX_CLASSES");
    if (!P->W || !P->b || !P->d || !P->U || !P->c) return fail(QBM_ERR_ARG, "null parameter block");
    return QBM_OK;
}

int check_dtype(int dt) { return (dt == QBM_F32 || dt == QBM_U8) ? QBM_OK : fail(QBM_ERR_ARG, "dtype must be QBM_F32 or QBM_U8"); }

// A call with no rows contributes nothing: its gradient blocks are zeroed so that a data-parallel rank whose
// shard of a short last minibatch is empty adds zeros to the all-reduce (not the previous step's gradients).
int zero_grads(const qbm_rbm_params* P, const qbm_rbm_grads* G, hipStream_t st)
{
    if (!G) return QBM_OK;
    const struct { float* p; size_t n; } blk[5] = {{G->gW, (size_t)P->V * P->H}, {G->gb, (size_t)P->V}, {G->gd, (size_t)P->H},
                                                    {G->gU, (size_t)P->C * P->H}, {G->gc, (size_t)P->C}};
    for (const auto& b : blk)
        if (b.p) {
            hipError_t e = hipMemsetAsync(b.p, 0, b.n * sizeof(float), st);
            if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(gradient block)");
        }
    if (G->n_rows) {   // no rows from this rank
        hipError_t e = hipMemsetAsync(G->n_rows, 0, sizeof(float), st);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(row count)");
    }
    return QBM_OK;
}

inline bool small2g(long elems, int esz) { return elems * esz < (1L << 31); }

template <int MT, int NT, int ASZ, int NQ>
int panel_for_nq(int K, int reserve)
{
    const int Kq = chain_len(K, NQ);
    const int KP = (Kq + 63) / 64 * 64;
    const int mx = strip_max_panel<MT, NT, ASZ, NQ>(reserve);
    return KP < mx ? KP : mx;
}

// Everything a launch of the hidden conditional needs (also the hidden phase inside k_hgrad)
struct HiddenPlan {
    HiddenArgs a;
    size_t lds;
    bool fast;
    int nM, nN;
};

template <class TA>
HiddenPlan plan_hidden(const qbm_rbm_params* P, const void* v, const float* y, int B, float* p, float* s, uint8_t* s8, int mode,
                       uint64_t seed, uint64_t offset, uint32_t row0, float* pre_out, const uint32_t* status,
                       const uint64_t* draw_base = nullptr)
{
    constexpr int ASZ = (int)sizeof(TA), MT = H_MT, NT = H_NT, NQ = H_NQ;
    constexpr int kScratch = NQ * 1024;   // one 1 KiB transfer slot per wavefront for the epilogue operands
    HiddenArgs a{v, y, P->W, P->d, P->U, p, s, s8, B, P->V, P->H, P->C, panel_for_nq<MT, NT, ASZ, NQ>(P->V, kScratch), seed,
                 offset, row0, false, 0, mode == 0 ? pre_out : nullptr, status, draw_base};
    // LDS-DMA path: whole 16-k pieces, 16-byte aligned units, descriptors below 2 GiB
    const bool fast = (P->V % 16 == 0) && aligned16(v) && (P->H % 4 == 0) && aligned16(P->W) &&
                      small2g((long)B * P->V, ASZ) && small2g((long)P->V * P->H, 4);
    const int nM = (B + 16 * MT - 1) / (16 * MT), nN = (P->H + 16 * NT - 1) / (16 * NT);
    a.pair = (nN % 16 == 0) ? 1 : 0;
    // More strips than CUs (B > 256 at H = 512): panels of 64 k, i.e. 56 KB of LDS instead of 104 KB, so that TWO workgroups
    // share a CU and one's transfers run under the other's MFMAs (784x512, uint8 states: B = 512 / 1024 / 2048 / 4096:
    // 48 / 83 / 141 / 262 us per CD-1 step against 51 / 92 / 161 / 306 with whole-chain panels; at B = 256, one workgroup
    // per CU either way, the whole-chain panel is 1-2 % faster).  Measured and dropped for the same sizes: workgroups that
    // keep their W panel in LDS and walk the row strips (50 / 86 / 150 / 281 us; for k_visible, whose 48 KB already allow
    // three workgroups per CU, slower than the plain grid).
    if (ASZ == 1 && fast && nM * nN > 256 && a.KP > 64) a.KP = 64;
    size_t lds = strip_lds_bytes<MT, NT, ASZ, NQ>(a.KP);
    // epilogue operands by LDS-DMA: ONE transfer instruction holds the U tile (4*C chunks of 16 B), the 4 y rows of a
    // wavefront (C chunks) and the bias tile (4 chunks); whole, 16-byte aligned chunks only
    const bool cls = mode == 0 && y;
    a.epiDma = fast && NT == 1 && MT == 2 && aligned16(P->d) &&
               (!cls || (5 * P->C + 4 <= 64 && ((long)B * P->C) % 4 == 0 && aligned16(y) && aligned16(P->U)));
    // ... and without the 8 KB of epilogue scratch three of them fit (B = 2048: 124 us per CD-1 step instead of 128): the
    // epilogue operands of those workgroups come by ordinary loads
    if (ASZ == 1 && fast && nM * nN > 256) a.epiDma = false;
    if (a.epiDma) {
        a.ldsCls = (int)lds;
        lds += kScratch;
    }
    return HiddenPlan{a, lds, fast, nM, nN};
}

template <class TA>
int launch_hidden_t(const qbm_rbm_params* P, const void* v, const float* y, int B, float* p, float* s, uint8_t* s8,
                    int mode, uint64_t seed, uint64_t offset, uint32_t row0, hipStream_t st, float* pre_out,
                    const uint32_t* status, const uint64_t* draw_base)
{
    constexpr int NQ = H_NQ;
    const HiddenPlan hp = plan_hidden<TA>(P, v, y, B, p, s, s8, mode, seed, offset, row0, pre_out, status, draw_base);
    const HiddenArgs& a = hp.a;
    const size_t lds = hp.lds;
    const bool fast = hp.fast;
    const int nM = hp.nM, nN = hp.nN;
#define QBM_LAUNCH(F, MODE, CM)                                                                      \
    {                                                                                                \
        if (MODE == 0 && (a.draw_base || a.status)) {                                                                     \
            if (int rc = set_lds(k_hidden<TA, F, 0, CM, true>, lds)) return rc;                                           \
            hipLaunchKernelGGL((k_hidden<TA, F, 0, CM, true>), dim3(8, nM, (nN + 7) / 8), dim3(64 * NQ), lds, st, a);     \
        } else {                                                                                                          \
            if (int rc = set_lds(k_hidden<TA, F, MODE, CM>, lds)) return rc;                                              \
            hipLaunchKernelGGL((k_hidden<TA, F, MODE, CM>), dim3(8, nM, (nN + 7) / 8), dim3(64 * NQ), lds, st, a);        \
        }                                                                                                                 \
    }
    if (mode != 0 || !y) {  // no class term: the class-count bucket is irrelevant
#define QBM_LF(MODE, CM) { if (a.epiDma) QBM_LAUNCH(2, MODE, CM) else if (fast) QBM_LAUNCH(1, MODE, CM) else QBM_LAUNCH(0, MODE, CM) }
        if (mode == 0) QBM_LF(0, 2) else QBM_LF(1, 2)
    } else {
#define QBM_L(CM) QBM_LF(0, CM)
        QBM_DISPATCH_CM(P->C, QBM_L);
#undef QBM_L
    }
#undef QBM_LF
#undef QBM_LAUNCH
    QBM_CHECK_LAUNCH("k_hidden");
    return QBM_OK;
}

int launch_hidden(const qbm_rbm_params* P, const void* v, int v_dtype, const float* y, int B, float* p, float* s,
                  uint8_t* s8, int mode, uint64_t seed, uint64_t offset, uint32_t row0, hipStream_t st, float* pre_out = nullptr,
                  const uint32_t* status = nullptr, const uint64_t* draw_base = nullptr)
{
    return v_dtype == QBM_U8 ? launch_hidden_t<uint8_t>(P, v, y, B, p, s, s8, mode, seed, offset, row0, st, pre_out, status, draw_base)
                             : launch_hidden_t<float>(P, v, y, B, p, s, s8, mode, seed, offset, row0, st, pre_out, status, draw_base);
}

// sample_visible, optionally with the class conditional riding along as aux workgroups
template <class TA>
int launch_visible_t(const qbm_rbm_params* P, const void* h, int B, float* p_v, float* v_s, uint8_t* v_s8, float* p_y,
                     float* y1hot, uint64_t seed, uint64_t offset_v, uint64_t offset_y, uint32_t row0, hipStream_t st,
                     const uint32_t* status, uint32_t* hg_round, const qbm_rbm_cd_workspace* mon, const uint64_t* draw_base)
{
    constexpr int ASZ = (int)sizeof(TA), MT = V_MT, NT = V_NT, NQ = V_NQ;
    const bool with_class = p_y || y1hot;
    const int nM = (B + 16 * MT - 1) / (16 * MT), nN = (P->V + 16 * NT - 1) / (16 * NT);
    VisibleArgs a{h, P->W, P->b, p_v, v_s, v_s8, B, P->V, P->H, panel_for_nq<MT, NT, ASZ, NQ>(P->H, 0), seed, offset_v, row0,
                  // pairs of visible units: 8-byte accesses to p / v_f32 / b, 2-byte accesses to v_u8
                  (P->V % 2 == 0) && (!p_v || aligned8(p_v)) && (!v_s || aligned8(v_s)) && (!v_s8 || aligned2(v_s8)) &&
                      aligned8(P->b),
                  P->U, P->c, p_y, y1hot, P->C, offset_y, with_class ? (B <= 512 ? B : (B + NQ - 1) / NQ) : 0, status, 0, nullptr, nullptr, nullptr, nullptr, 0, div_magic(nM),
                  draw_base, status ? hg_round : nullptr};
    // the request stagger pays where a CU holds one strip workgroup and a wavefront's chain is a whole quad (784 x 512, B = 256:
    // 6.55 -> 6.36 us); with several workgroups per CU they fill each other's gaps, and it costs (B = 2048: 121.8 -> 122.4 us
    // per step; 784 x 128, B = 128: +0.3 us)
    a.stag = (nM * nN <= 256 && P->H >= 256) ? 1 : 0;
    if (mon && a.n_aux == B) {   // (visible_takes_monitor: the caller has asked)
        a.n_fe = B; a.fe_pre = mon->mon_pre; a.fe_labels = mon->mon_labels; a.fe_p = mon->mon_p; a.fe_rowloss = mon->mon_rowloss;
    }
    const bool fast = (P->H % 16 == 0) && aligned16(h) && aligned16(P->W) && small2g((long)B * P->H, ASZ) &&
                      small2g((long)P->V * P->H, 4);
    const int grid = ((a.n_aux == B || a.n_aux == 0) ? strip_grid_padded(nM, nN) : nM * nN) + a.n_aux + a.n_fe;
    const size_t lds = strip_lds_bytes<MT, NT, ASZ, NQ>(a.KP);   // (>= the aux workgroups' few hundred bytes)
    // (only with fewer classes than wavefronts: at C = 10 the staged form costs the step 0.7 us more than two classes on two of
    // the wavefronts -- measured)
    a.fe_stage = a.n_fe > 0 && P->C < NQ && 1280 + (size_t)P->C * P->H * sizeof(float) <= lds;
#define QBM_LAUNCH(F, CM)                                                                       \
    {                                                                                           \
        if (a.draw_base || a.status) {                                                                \
            if (int rc = set_lds(k_visible<TA, F, CM, true>, lds)) return rc;                         \
            hipLaunchKernelGGL((k_visible<TA, F, CM, true>), dim3(grid), dim3(64 * NQ), lds, st, a);  \
        } else {                                                                                      \
            if (int rc = set_lds(k_visible<TA, F, CM>, lds)) return rc;                               \
            hipLaunchKernelGGL((k_visible<TA, F, CM>), dim3(grid), dim3(64 * NQ), lds, st, a);        \
        }                                                                                             \
    }
#define QBM_L(CM) { if (fast) QBM_LAUNCH(true, CM) else QBM_LAUNCH(false, CM) }
    if (!with_class) { QBM_L(2) }
    else { QBM_DISPATCH_CM(P->C, QBM_L); }
#undef QBM_L
#undef QBM_LAUNCH
    QBM_CHECK_LAUNCH("k_visible");
    return QBM_OK;
}

int launch_visible(const qbm_rbm_params* P, const void* h, int h_dtype, int B, float* p_v, float* v_s, uint8_t* v_s8,
                   float* p_y, float* y1hot, uint64_t seed, uint64_t offset_v, uint64_t offset_y, uint32_t row0,
                   hipStream_t st, const uint32_t* status = nullptr, uint32_t* hg_round = nullptr,
                   const qbm_rbm_cd_workspace* mon = nullptr, const uint64_t* draw_base = nullptr)
{
    return h_dtype == QBM_U8
               ? launch_visible_t<uint8_t>(P, h, B, p_v, v_s, v_s8, p_y, y1hot, seed, offset_v, offset_y, row0, st, status, hg_round, mon, draw_base)
               : launch_visible_t<float>(P, h, B, p_v, v_s, v_s8, p_y, y1hot, seed, offset_v, offset_y, row0, st, status, hg_round, mon, draw_base);
}

// Does the k_visible launch of a CD step carry the training loop's monitor (one aux workgroup per row)?  Only where the
// class conditional rides as one workgroup per row too (B <= 512) and the first Gibbs iteration is a launch of its own.
inline bool visible_takes_monitor(int B) { return B <= 512; }

// what the gradient launch of the discriminative step carries besides the strips (kind D workgroups, one phase)
struct DiscExtras {
    const float* pre; const float* U; const float* pyx; const int* labels; const float* rowloss; double* loss_acc;
    float* dgU; float* dgc; float* duU; float* duc; float* dub;
};

template <class T0, class T1, bool PAR, int NT, int MT = G_MT>
int launch_grad_t(int V, int H, int C, int B, const void* A0, const float* Y0, const float* P0, const void* A1,
                  const float* Y1, const float* P1, const qbm_rbm_grads* G, float* W_fused, const qbm_rbm_params* aux_fused,
                  float lr, float factor, float bs, float sparse, hipStream_t st, const float* gU_sub, const uint32_t* status,
                  const DiscExtras* dx, const float* mon_rowloss, double* mon_loss_acc)
{
    constexpr int UNT = MT == G_MT7 ? 2 : NT;   // tiles per workgroup of kinds U / C (k_grad)
    const int nM = (V + 16 * MT - 1) / (16 * MT), nN = (H + 16 * NT - 1) / (16 * NT);
    const bool want_b = G->gb || (aux_fused && aux_fused->b);
    const bool want_c = G->gc || (aux_fused && aux_fused->c);
    GradArgs a{A0, Y0, P0, A1, Y1, P1, G->gW, G->gb, G->gd, G->gU, G->gc, W_fused,
               aux_fused ? aux_fused->b : nullptr, aux_fused ? aux_fused->d : nullptr,
               aux_fused ? aux_fused->U : nullptr, aux_fused ? aux_fused->c : nullptr,
               lr, factor, bs, sparse, B, V, H, C, panel_for<MT, NT, 4>(B),
               (H + 16 * UNT - 1) / (16 * UNT), want_b ? nM : 0, want_c ? (C + 16 * UNT - 1) / (16 * UNT) : 0,
               (H % 4 == 0) && (!G->gW || aligned16(G->gW)) && (!W_fused || aligned16(W_fused)), gU_sub, 0, 0, 0, div_magic(nM), status, G->n_rows,
               0, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (mon_rowloss && mon_loss_acc) { a.nL = 1; a.rowloss = mon_rowloss; a.loss_acc = mon_loss_acc; }
    if (dx) {
        a.pre = dx->pre; a.Up = dx->U; a.pyx = dx->pyx; a.labels = dx->labels; a.rowloss = dx->rowloss; a.loss_acc = dx->loss_acc;
        a.dgU = dx->dgU; a.dgc = dx->dgc; a.duU = dx->duU; a.duc = dx->duc; a.dub = dx->dub;
    }
    const bool fast = (B % 16 == 0) && (V % 16 == 0) && (H % 4 == 0) && aligned16(A0) && (!P1 || aligned16(A1)) && aligned16(P0) &&
                      aligned16(P1) && small2g((long)B * V, 4) && small2g((long)B * H, 4);
    // PAR: the phases side by side (both operands uint8, two phases); otherwise 256 threads, phase after phase -- or one
    // phase only (P1 == null: the discriminative step)
    static_assert(!PAR || (sizeof(T0) == 1 && sizeof(T1) == 1), "the phase-parallel kernel takes uint8 data");
    if (PAR && (!P1 || dx)) return fail(QBM_ERR_ARG, "two_phase_grad: the one-phase form takes the 256-thread kernel");
    if (dx && P1) return fail(QBM_ERR_ARG, "two_phase_grad: kind D workgroups ride with the one-phase form only");
    size_t lds;
    if constexpr (PAR) {
        // one staging region per phase group, sized for the largest of the four kinds of workgroup (uint8 data; the class
        // rows / ones operands of kinds U, C are synthesised fp32 single-tile operands); the strips the groups swap
        // afterwards (2 x NP float4 per thread of a group) must fit as well
        size_t g = strip_lds_bytes<MT, NT, 1>(a.KP);
        if (g < (size_t)strip_lds_bytes<1, UNT, 4>(a.KP)) g = strip_lds_bytes<1, UNT, 4>(a.KP);
        if (g < (size_t)strip_lds_bytes<MT, 1, 1>(a.KP)) g = strip_lds_bytes<MT, 1, 1>(a.KP);
        a.lds_group = (int)g;
        lds = 2 * g;
        const size_t swap_bytes = 2 * StripOut<MT, NT>::NP * 256 * sizeof(float4);
        if (lds < swap_bytes) lds = swap_bytes;
        if (lds > 163840) return fail(QBM_ERR_ARG, "two_phase_grad: LDS budget exceeded");
    } else {
        constexpr int ASZ = sizeof(T0) == 1 && sizeof(T1) == 1 ? 1 : 4;   // an fp32 operand sets the budget
        lds = strip_lds_bytes<MT, NT, ASZ>(a.KP);
        if (lds < (size_t)strip_lds_bytes<1, UNT, 4>(a.KP)) lds = strip_lds_bytes<1, UNT, 4>(a.KP);   // kinds U, C: synthesised fp32 tile
        if (lds < 24576) lds = 24576;                                                                // kind D / kind L: their LDS arrays
    }
    if (dx) {
        // Kind D workgroups (class c, 16 hidden columns) stand behind the strips.  Where the launch's LDS allows ONE workgroup per
        // CU only, 2 or 4 column blocks per kind D workgroup are taken if that lets them run beside the strips instead of
        // queueing for further rounds (8 blocks take longer than a strip; with two workgroups per CU the queue is cheaper than
        // the wider workgroups: C = 4, fp32: 24.1 us per step with one block each, 27.0 with two -- measured)
        const int nJ = (H + 15) / 16;
        const int slots = 256 * (int)(163840 / lds), others = a.nU + a.nB + a.nC + nM * nN;
        a.dW = 1;
        if (slots == 256 && others < slots && C * nJ > slots - others)
            for (int w = 2; w <= 4; w *= 2)
                if (C * ((nJ + w - 1) / w) <= slots - others) { a.dW = w; break; }
        a.nD = C * ((nJ + a.dW - 1) / a.dW);
    }
    const int grid = a.nU + a.nB + a.nC + nM * nN + a.nD + a.nL;
#define QBM_LAUNCH(F, ONE)                                                                                         \
    {                                                                                                               \
        if (int rc = set_lds(k_grad<T0, T1, F, PAR, ONE, NT, MT>, lds)) return rc;                                  \
        hipLaunchKernelGGL((k_grad<T0, T1, F, PAR, ONE, NT, MT>), dim3(grid), dim3(PAR ? 512 : 256), lds, st, a);   \
    }
    if constexpr (PAR) {
        if (fast) QBM_LAUNCH(true, false) else QBM_LAUNCH(false, false)
    } else if (dx) {   // one phase + kind D: instantiated for A0 = A1 types only (the discriminative step hands x over twice)
        if constexpr (std::is_same<T0, T1>::value) {
            if (fast) QBM_LAUNCH(true, true) else QBM_LAUNCH(false, true)
        } else {
            return fail(QBM_ERR_ARG, "two_phase_grad: the one-phase form takes one operand type");
        }
    } else {
        if (!P1) return fail(QBM_ERR_ARG, "two_phase_grad: second phase missing");
        if (fast) QBM_LAUNCH(true, false) else QBM_LAUNCH(false, false)
    }
#undef QBM_LAUNCH
    QBM_CHECK_LAUNCH("k_grad");
    return QBM_OK;
}

int launch_grad(int V, int H, int C, int B, const void* A0, int d0, const float* Y0, const float* P0, const void* A1, int d1,
                const float* Y1, const float* P1, const qbm_rbm_grads* G, float* W_fused, const qbm_rbm_params* aux_fused,
                float lr, float factor, float bs, float sparse, hipStream_t st, const float* gU_sub = nullptr,
                const uint32_t* status = nullptr, const DiscExtras* dx = nullptr, const float* mon_rowloss = nullptr,
                double* mon_loss_acc = nullptr)
{
    if (C + 1 > 16) return fail(QBM_ERR_ARG, "two_phase_grad: at most 15 classes (class rows + the bias row share one 16-row tile)");
#define QBM_G(T0, T1, PAR, NT) launch_grad_t<T0, T1, PAR, NT>(V, H, C, B, A0, Y0, P0, A1, Y1, P1, G, W_fused, aux_fused, lr, factor, bs, sparse, st, gU_sub, status, dx, mon_rowloss, mon_loss_acc)
    // strips of 64 x 16 for the one-phase form and where 64 x 32 strips would leave half of the CUs idle (see G_MT)
    const int wide = ((V + 16 * G_MT - 1) / (16 * G_MT)) * ((H + 31) / 32);
    const bool narrow = dx || !P1 || wide < 128;
    if (d0 == QBM_U8 && d1 == QBM_U8) {
        if (P1 && !dx) {
            // 112 x 16 strips where they tile V exactly, fill more CUs than the 64 x 32 ones and the launch -- strips, kinds U
            // (32 columns each), B, C, L -- is still ONE round of one workgroup per CU (see G_MT7)
            const int tall = (V / (16 * G_MT7)) * ((H + 15) / 16);
            if (QBM_GRAD_112 && !narrow && V % (16 * G_MT7) == 0 && tall > wide &&
                tall + (H + 31) / 32 + V / (16 * G_MT7) + (C + 31) / 32 + 1 <= 256)
                return launch_grad_t<uint8_t, uint8_t, true, 1, G_MT7>(V, H, C, B, A0, Y0, P0, A1, Y1, P1, G, W_fused, aux_fused, lr, factor,
                                                                       bs, sparse, st, gU_sub, status, dx, mon_rowloss, mon_loss_acc);
            return narrow ? QBM_G(uint8_t, uint8_t, true, 1) : QBM_G(uint8_t, uint8_t, true, 2);
        }
        return QBM_G(uint8_t, uint8_t, false, 1);
    }
    if (d0 == QBM_U8) return QBM_G(uint8_t, float, false, 2);
    if (d1 == QBM_U8) return QBM_G(float, uint8_t, false, 2);
    return (dx || !P1) ? QBM_G(float, float, false, 1) : QBM_G(float, float, false, 2);
#undef QBM_G
}

// 0 = never, 1 = where it is faster (default), 2 = whenever the shape allows, 3 = 1 + the fused tail (k_hgrad) where the shape
// allows; initial value from QBM_FUSED_CHAIN
std::atomic<int>& fused_chain_mode_ref()
{
    static std::atomic<int> mode{[] {
        const char* e = getenv("QBM_FUSED_CHAIN");
        return (e && e[0] >= '0' && e[0] <= '3') ? e[0] - '0' : 1;
    }()};
    return mode;
}
int fused_chain_mode() { return fused_chain_mode_ref().load(); }

// 8 XCDs x 32 CUs: one workgroup per CU, 32 per XCD -- asked of the device the call runs on, cached per device
// (0 = not asked yet, 1 = fits, 2 = does not)
bool device_has_8x32_cus()
{
    static std::atomic<signed char> fits_cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    signed char fc = (dev >= 0 && dev < 64) ? fits_cache[dev].load(std::memory_order_relaxed) : (signed char)0;
    if (fc == 0) {
        int cus = 0;
        fc = (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus == 256) ? 1 : 2;
        if (dev >= 0 && dev < 64) fits_cache[dev].store(fc, std::memory_order_relaxed);
    }
    return fc == 1;
}

// The fused chain kernel for the shapes it is written for; QBM_OK + *done = false when the classic sequence of launches
// has to run instead.
int launch_chain_fused(const qbm_rbm_params* P, const void* v0, int v0_dtype, const float* y0, int B, int k, uint64_t seed,
                       uint64_t offset, uint32_t row0, uint8_t* chain_v, float* chain_y, const qbm_rbm_cd_workspace* ws,
                       hipStream_t st, bool* done, float* mon_pre)
{
    *done = false;
    const bool device_fits = device_has_8x32_cus();
    const int mode = fused_chain_mode();
    // mode 1: where it is faster -- from two Gibbs iterations on (CD-1: the first hidden phase of the fused kernel streams all
    // of W into every XCD, 12.8 MB instead of 1.6, which costs what the two saved launch boundaries gain; measured
    // 30.0 vs 29.9 us per step; PCD-10: 136 vs 160)
    const bool enabled = device_fits && (mode == 2 || ((mode == 1 || mode == 3) && k >= 2));
    if (!enabled || !ws->sync || v0_dtype != QBM_U8 || ws->draw_base) return QBM_OK;   // (device-side draw counters: launch per phase)
    const int V = P->V, H = P->H, C = P->C;
    if (B % 32 != 0 || B > 256 || H % 16 != 0 || H > 512 || V % 16 != 0 || V > 1024 || 5 * C + 4 > 64) return QBM_OK;
    const void* ptrs[] = {v0, y0, P->W, P->b, P->d, P->U, ws->h, ws->vk, ws->yk, chain_v, chain_y};
    for (const void* q : ptrs)
        if (q && !aligned16(q)) return QBM_OK;
    if (!small2g((long)V * H, 4)) return QBM_OK;
    constexpr int NQ = 8;
    constexpr int kScratch = NQ * 1024;
    ChainArgs a{P->W, P->b, P->d, P->U, P->c, (const uint8_t*)v0, y0, chain_v, chain_y, ws->ph0, ws->h, ws->vk, ws->yk, ws->phk,
                ws->sync, B, V, H, C, k, (chain_len(V, NQ) + 63) / 64 * 64, panel_for_nq<2, 2, 1, NQ>(H, 0), 0, 0, seed,
                offset, row0, V % 2 == 0, mon_pre};
    // LDS: [W panel of the hidden phases, resident | transient: A tiles + epilogue scratch of a hidden phase, or the whole
    // staging of a visible phase].  One panel per chain in the hidden phases (the resident image IS the whole chain).
    if (a.KPh < chain_len(V, NQ)) return QBM_OK;
    a.ldsT = NQ * strip_wave_bytes_b<H_NT>(a.KPh);
    const size_t lds_ha = (size_t)NQ * strip_wave_bytes_a<H_MT, H_NT, 1>(a.KPh), lds_v = strip_lds_bytes<2, 2, 1, NQ>(a.KPv);
    a.ldsCls = (int)lds_ha;
    const size_t lds = a.ldsT + (lds_ha + kScratch > lds_v ? lds_ha + kScratch : lds_v);
    if (lds > 160 * 1024) return QBM_OK;
#define QBM_L(CM)                                                                        \
    {                                                                                    \
        if (int rc = set_lds(k_chain<CM>, lds)) return rc;                               \
        hipLaunchKernelGGL((k_chain<CM>), dim3(256), dim3(512), lds, st, a);             \
    }
    QBM_DISPATCH_CM(C, QBM_L);
#undef QBM_L
    QBM_CHECK_LAUNCH("k_chain");
    *done = true;
    return QBM_OK;
}

// test hook (qbm_test_hgrad_fault): the next k_hgrad launch of mode 0 makes one workgroup misbehave
std::atomic<int> g_hgrad_fault_wg{-1}, g_hgrad_fault_kind{0};

// Can the tail of this CD step -- last hidden phase + gradient (+ update) -- be ONE launch (k_hgrad)?
// force: the shape test alone, whatever the process-wide switch says (qbm_rbm_cd_step_finish asks as the step asked, when
// the switch may be off already; ADVICE round 3: the switch itself is not touched)
bool hgrad_eligible(const qbm_rbm_params* P, const qbm_rbm_grads* G, const void* v0, int v0_dtype, const float* y0, int B,
                    const uint8_t* v1, const float* y1, const qbm_rbm_cd_workspace* ws, int fuse_update, bool force = false)
{
    // mode 3 only: measured slower than the launch-per-phase tail on every shape tried -- kept as an
    // opt-in, with its tests, next to the numbers that say why
    if ((!force && fused_chain_mode() != 3) || !device_has_8x32_cus() || !ws || !ws->sync || v0_dtype != QBM_U8 || ws->draw_base) return false;
    const int V = P->V, H = P->H, C = P->C;
    const int nM = (B + 16 * H_MT - 1) / (16 * H_MT), nN = (H + 16 * H_NT - 1) / (16 * H_NT);
    // (B <= 256: one panel of 64 rows per K chain of the gradient -- its operands are requested in one go under the hidden phase)
    if (B <= 0 || B > 256 || B % 16 != 0 || H % 16 != 0 || V % 16 != 0 || nN % 8 != 0 || nN > 32 || nM * nN > 256) return false;
    if (5 * C + 4 > 64 || ((long)B * C) % 4 != 0 || C + 1 > 16) return false;
    const void* ptrs[] = {v0, y0, v1, y1, P->W, P->d, P->U, ws->ph0, ws->phk, G->gW};
    for (const void* q : ptrs)
        if (q && !aligned16(q)) return false;
    if (!small2g((long)B * V, 4) || !small2g((long)V * H, 4) || !small2g((long)B * H, 4)) return false;
    (void)fuse_update;
    // the hidden phase on its all-DMA path, and room behind it for the operands of the gradient's phase 0 (launch_hgrad)
    const HiddenPlan hp = plan_hidden<uint8_t>(P, v1, y1, B, ws->phk, nullptr, nullptr, 0, 0, 0, 0, nullptr, nullptr);
    if (!hp.fast || !hp.a.epiDma) return false;
    const size_t area = 2 * 4 * (size_t)strip_wave_bytes_a<HG_MT, H_NT, 1>(64) + 4 * (size_t)strip_wave_bytes_b<H_NT>(64);
    return (hp.lds > area + 16 ? hp.lds : area + 16) + area <= 163840;
}

// mode 0: the step's tail; 1 / 2: finishing a step whose launch of round `round` gave up (qbm_rbm_cd_step_finish)
int launch_hgrad(const qbm_rbm_params* P, const qbm_rbm_grads* G, const void* v0, const float* y0, int B, uint64_t seed,
                 uint64_t offset, uint64_t off_h, uint32_t row0, const uint8_t* v1, const float* y1, const qbm_rbm_cd_workspace* ws,
                 int fuse_update, float lr, float factor, float bs, float sparse, const float* mon_rowloss, double* mon_loss_acc,
                 hipStream_t st, int mode, uint32_t round)
{
    const int V = P->V, H = P->H, C = P->C;
    HiddenPlan hp = plan_hidden<uint8_t>(P, v1, y1, B, ws->phk, nullptr, nullptr, 0, seed, off_h, row0, nullptr, nullptr);
    if (!hp.fast || !hp.a.epiDma) return fail(QBM_ERR_ARG, "cd_step: fused tail launched on a shape it does not take");
    qbm_rbm_grads g = *G;
    if (fuse_update) g.gW = nullptr;   // the whole update happens inside the launch
    const bool want_b = g.gb || fuse_update, want_c = g.gc || fuse_update;
    GradArgs ga{v0, y0, ws->ph0, v1, y1, ws->phk, g.gW, g.gb, g.gd, g.gU, g.gc, fuse_update ? P->W : nullptr,
                fuse_update ? P->b : nullptr, fuse_update ? P->d : nullptr, fuse_update ? P->U : nullptr,
                fuse_update ? P->c : nullptr, lr, factor, bs, sparse, B, V, H, C, 64,
                0, want_b ? 1 : 0, want_c ? 1 : 0,
                (H % 4 == 0) && (!g.gW || aligned16(g.gW)) && aligned16(P->W), nullptr, 0, 0, 0, 0u, nullptr, G->n_rows,
                0, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (mon_rowloss && mon_loss_acc) { ga.nL = 1; ga.rowloss = mon_rowloss; ga.loss_acc = mon_loss_acc; }
    // LDS.  Items that go in one call: a staging region per thread group, sized for the largest kind (see launch_grad_t).
    size_t grp = strip_lds_bytes<HG_MT, H_NT, 1>(ga.KP);
    if (grp < (size_t)strip_lds_bytes<1, H_NT, 4>(ga.KP)) grp = strip_lds_bytes<1, H_NT, 4>(ga.KP);
    if (grp < (size_t)strip_lds_bytes<HG_BMT, 1, 1>(ga.KP)) grp = strip_lds_bytes<HG_BMT, 1, 1>(ga.KP);
    ga.lds_group = (int)grp;
    // The W item in several calls (hg_w_item): [A tiles of thread group 0 | of group 1 | the P panel] per phase -- phase 1 at
    // the front (the hidden phase's staging area, dead by then), the barrier's result word behind it, phase 0 behind
    // everything the hidden phase uses.
    const size_t area = 2 * 4 * (size_t)strip_wave_bytes_a<HG_MT, H_NT, 1>(ga.KP) + 4 * (size_t)strip_wave_bytes_b<H_NT>(ga.KP);
    ga.lds_sok = (int)area;
    ga.lds_group0 = (int)(hp.lds > area + 16 ? hp.lds : area + 16);
    size_t lds = ga.lds_group0 + area;
    if (lds < 2 * grp) lds = 2 * grp;
    if (lds < 2 * 2 * 256 * sizeof(float4)) lds = 2 * 2 * 256 * sizeof(float4);   // the strips the groups exchange (small items)
    if (lds < 24576) lds = 24576;                                                  // the loss terms' arrays
    if (lds > 163840) return fail(QBM_ERR_ARG, "cd_step: LDS budget of the fused tail exceeded");
    HGradArgs a{hp.a, ga, ws->sync, offset, mode, round, -1, 0};
    if (mode == 0) {
        a.fault_wg = g_hgrad_fault_wg.exchange(-1);
        a.fault_kind = g_hgrad_fault_kind.load();
    }
#define QBM_L(CM)                                                                          \
    {                                                                                      \
        if (int rc = set_lds(k_hgrad<CM>, lds)) return rc;                                 \
        hipLaunchKernelGGL((k_hgrad<CM>), dim3(hp.nM * hp.nN), dim3(512), lds, st, a);     \
    }
    QBM_DISPATCH_CM(C, QBM_L);
#undef QBM_L
    QBM_CHECK_LAUNCH("k_hgrad");
    return QBM_OK;
}


}  // namespace

// =================================================================================================
// C ABI (include/qbm_hip.h)
// =================================================================================================
extern "C" {

int qbm_rbm_fused_chain_mode(int mode)
{
    if (mode < 0 || mode > 3) return fused_chain_mode();
    return fused_chain_mode_ref().exchange(mode);
}


int qbm_abi_version(void) { return QBM_ABI_VERSION; }
const char* qbm_last_error(void) { return g_err; }

int qbm_rbm_sample_hidden(const qbm_rbm_params* P, const float* v, const float* y1hot, int B, float* p_h,
                          float* h_sample, uint64_t seed, uint64_t offset, uint32_t row0, void* stream)
{
    return qbm_rbm_sample_hidden_ex(P, v, QBM_F32, y1hot, B, p_h, h_sample, nullptr, seed, offset, row0, stream);
}

int qbm_rbm_sample_hidden_ex(const qbm_rbm_params* P, const void* v, int v_dtype, const float* y1hot, int B, float* p_h,
                             float* h_f32, uint8_t* h_u8, uint64_t seed, uint64_t offset, uint32_t row0, void* stream)
{
    if (int rc = check_params(P)) return rc;
    if (int rc = check_dtype(v_dtype)) return rc;
    if (B < 0) return fail(QBM_ERR_ARG, "B < 0");
    if (B == 0) return QBM_OK;
    if (!v || (!p_h && !h_f32 && !h_u8)) return fail(QBM_ERR_ARG, "sample_hidden: null input/output");
    return launch_hidden(P, v, v_dtype, y1hot, B, p_h, h_f32, h_u8, 0, seed, offset, row0, (hipStream_t)stream);
}

int qbm_rbm_hidden_preact(const qbm_rbm_params* P, const float* x, int B, float* pre, void* stream)
{
    if (int rc = check_params(P)) return rc;
    if (B < 0) return fail(QBM_ERR_ARG, "B < 0");
    if (B == 0) return QBM_OK;
    if (!x || !pre) return fail(QBM_ERR_ARG, "hidden_preact: null input/output");
    return launch_hidden(P, x, QBM_F32, nullptr, B, pre, nullptr, nullptr, 1, 0, 0, 0, (hipStream_t)stream);
}

int qbm_rbm_sample_visible(const qbm_rbm_params* P, const float* h, int B, float* p_v, float* v_sample,
                           uint64_t seed, uint64_t offset, uint32_t row0, void* stream)
{
    if (int rc = check_params(P)) return rc;
    if (B < 0) return fail(QBM_ERR_ARG, "B < 0");
    if (B == 0) return QBM_OK;
    if (!h || (!p_v && !v_sample)) return fail(QBM_ERR_ARG, "sample_visible: null input/output");
    return launch_visible(P, h, QBM_F32, B, p_v, v_sample, nullptr, nullptr, nullptr, seed, offset, 0, row0,
                          (hipStream_t)stream);
}

int qbm_rbm_sample_visible_class_ex(const qbm_rbm_params* P, const void* h, int h_dtype, int B, float* p_v, float* v_f32,
                                    uint8_t* v_u8, float* p_y, float* y1hot, uint64_t seed, uint64_t offset_v,
                                    uint64_t offset_y, uint32_t row0, void* stream)
{
    if (int rc = check_params(P)) return rc;
    if (int rc = check_dtype(h_dtype)) return rc;
    if (B < 0) return fail(QBM_ERR_ARG, "B < 0");
    if (B == 0) return QBM_OK;
    if (!h || (!p_v && !v_f32 && !v_u8)) return fail(QBM_ERR_ARG, "sample_visible_class: null input/output");
    return launch_visible(P, h, h_dtype, B, p_v, v_f32, v_u8, p_y, y1hot, seed, offset_v, offset_y, row0,
                          (hipStream_t)stream);
}

int qbm_rbm_sample_class(const qbm_rbm_params* P, const float* h, int B, float* p_y, float* y1hot, uint64_t seed,
                         uint64_t offset, uint32_t row0, void* stream)
{
    if (int rc = check_params(P)) return rc;
    if (B < 0) return fail(QBM_ERR_ARG, "B < 0");
    if (B == 0) return QBM_OK;
    if (!h || (!p_y && !y1hot)) return fail(QBM_ERR_ARG, "sample_class: null input/output");
#define QBM_L(CM)                                                                                                  \
    hipLaunchKernelGGL((k_class<float, CM>), dim3(B), dim3(256), 0, (hipStream_t)stream, h, P->U, P->c, p_y, \
                       y1hot, B, P->H, P->C, seed, offset, row0)
    QBM_DISPATCH_CM(P->C, QBM_L);
#undef QBM_L
    QBM_CHECK_LAUNCH("k_class");
    return QBM_OK;
}

int qbm_rbm_free_energy(const qbm_rbm_params* P, const float* x, int B, float* pre_ws, float* neg_free_energy,
                        float* p_y_given_x, void* stream)
{
    if (int rc = check_params(P)) return rc;
    if (B < 0) return fail(QBM_ERR_ARG, "B < 0");
    if (B == 0) return QBM_OK;
    if (!x || !pre_ws || (!neg_free_energy && !p_y_given_x)) return fail(QBM_ERR_ARG, "free_energy: null buffer");
    if (int rc = launch_hidden(P, x, QBM_F32, nullptr, B, pre_ws, nullptr, nullptr, 1, 0, 0, 0, (hipStream_t)stream)) return rc;
#define QBM_L(CM)                                                                                               \
    launch_free_energy<CM>(pre_ws, P->U, P->c, neg_free_energy, p_y_given_x, B, P->H, P->C, nullptr, nullptr, nullptr,  \
                           nullptr, (hipStream_t)stream)
    QBM_DISPATCH_CM(P->C, QBM_L);
#undef QBM_L
    QBM_CHECK_LAUNCH("k_free_energy");
    return QBM_OK;
}

int qbm_rbm_two_phase_grad(int V, int H, int C, int B, const void* A0, int a0_dtype, const float* Y0, const float* P0,
                           const void* A1, int a1_dtype, const float* Y1, const float* P1, const qbm_rbm_grads* G,
                           float* W_fused, const qbm_rbm_params* aux_fused, float lr, float factor, float batch_size,
                           float sparse_constant, void* stream)
{
    if (V <= 0 || H <= 0 || C <= 0 || B <= 0) return fail(QBM_ERR_ARG, "two_phase_grad: non-positive dims");
    if (int rc = check_dtype(a0_dtype)) return rc;
    if (int rc = check_dtype(a1_dtype)) return rc;
    if (!A0 || !P0 || !A1 || !P1 || !G) return fail(QBM_ERR_ARG, "two_phase_grad: null operand");
    if (!G->gW && !W_fused) return fail(QBM_ERR_ARG, "two_phase_grad: neither gW nor fused W given");
    if ((W_fused || aux_fused) && !(batch_size > 0.f)) return fail(QBM_ERR_ARG, "two_phase_grad: batch_size must be positive");
    return launch_grad(V, H, C, B, A0, a0_dtype, Y0, P0, A1, a1_dtype, Y1, P1, G, W_fused, aux_fused, lr, factor,
                       batch_size, sparse_constant, (hipStream_t)stream);
}

int qbm_rbm_update(const qbm_rbm_params* P, const qbm_rbm_grads* G, float lr, float factor, float batch_size,
                   float sparse_constant, void* stream)
{
    return qbm_rbm_update_dev(P, G, lr, factor, batch_size, nullptr, sparse_constant, nullptr, stream);
}

int qbm_rbm_update_dev(const qbm_rbm_params* P, const qbm_rbm_grads* G, float lr, float factor, float batch_size,
                       const float* batch_size_dev, float sparse_constant, const uint32_t* chain_status, void* stream)
{
    if (int rc = check_params(P)) return rc;
    if (!G) return fail(QBM_ERR_ARG, "update: grads is null");
    if (!batch_size_dev && !(batch_size > 0.f)) return fail(QBM_ERR_ARG, "update: batch_size must be positive");
    const long nW = (long)P->V * P->H;
    const long total = nW + P->V + P->H + (long)P->C * P->H + P->C;
    UpdateArgs a{P->W, P->b, P->d, P->U, P->c, G->gW, G->gb, G->gd, G->gU, G->gc, P->V, P->H, P->C,
                 lr, factor, batch_size, sparse_constant, G->gW ? 0 : nW, batch_size_dev, chain_status};
    const long n = total - a.first;
    const int blocks = (int)(((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048) + 7) / 8 * 8;   // whole groups of 8: k_update's line map
    hipLaunchKernelGGL(k_update, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    QBM_CHECK_LAUNCH("k_update");
    return QBM_OK;
}

int qbm_f32_to_u8(const float* x, uint8_t* y, long n, void* stream)
{
    if (n < 0) return fail(QBM_ERR_ARG, "f32_to_u8: n < 0");
    if (n == 0) return QBM_OK;
    if (!x || !y) return fail(QBM_ERR_ARG, "f32_to_u8: null buffer");
    hipLaunchKernelGGL(k_f32_to_u8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    QBM_CHECK_LAUNCH("k_f32_to_u8");
    return QBM_OK;
}

// defer_last: the caller runs the last hidden phase itself (fused with the gradient: k_hgrad) -- unless the one-launch chain
// takes the step (*fused); hg_status: status word every launch tests (HiddenArgs::status), or null
static int cd_chain_impl(const qbm_rbm_params* P, const void* v0, int v0_dtype, const float* y0, int B, int k, uint64_t seed,
                         uint64_t offset, uint32_t row0, uint8_t* chain_v, float* chain_y, const qbm_rbm_cd_workspace* ws,
                         void* stream, bool* fused, float* mon_pre = nullptr, bool defer_last = false,
                         const uint32_t* hg_status = nullptr, bool mon_in_visible = false)
{
    *fused = false;
    if (int rc = check_params(P)) return rc;
    if (int rc = check_dtype(v0_dtype)) return rc;
    if (B <= 0) return B == 0 ? QBM_OK : fail(QBM_ERR_ARG, "B < 0");
    if (k < 1) return fail(QBM_ERR_ARG, "cd_chain: k must be >= 1");
    if (!v0 || !y0 || !ws || !ws->ph0 || !ws->h || !ws->vk || !ws->yk || !ws->phk)
        return fail(QBM_ERR_ARG, "cd_chain: null buffer");
    if ((chain_v == nullptr) != (chain_y == nullptr)) return fail(QBM_ERR_ARG, "cd_chain: chain_v and chain_y go together");
    hipStream_t st = (hipStream_t)stream;
    {   // one launch for the whole chain where the shape allows it (B <= 256 in blocks of 32 rows, uint8 data)
        bool done = false;
        if (int rc = launch_chain_fused(P, v0, v0_dtype, y0, B, k, seed, offset, row0, chain_v, chain_y, ws, st, &done, mon_pre)) return rc;
        *fused = done;
        if (done) return QBM_OK;
    }
    // (mon_pre: the training loop's monitor wants v0.W + d -- one more store of the positive-phase launch)
    const uint64_t* db = ws->draw_base;   // device-side draw counters (added to offset / row0 by every sampling launch), or null
    if (chain_v) {
        // positive phase on the data (probabilities only), negative chain from the persistent state
        if (int rc = launch_hidden(P, v0, v0_dtype, y0, B, ws->ph0, nullptr, nullptr, 0, seed, offset, row0, st, mon_pre, hg_status, db)) return rc;
        if (int rc = launch_hidden(P, chain_v, QBM_U8, chain_y, B, nullptr, nullptr, ws->h, 0, seed, offset, row0, st, nullptr, hg_status, db)) return rc;
    } else {
        if (int rc = launch_hidden(P, v0, v0_dtype, y0, B, ws->ph0, nullptr, ws->h, 0, seed, offset, row0, st, mon_pre, hg_status, db)) return rc;
    }
    for (int t = 1; t <= k; ++t) {
        const uint64_t o = offset + 3ull * (uint64_t)t;
        // PCD: the last iteration writes (v_k, y_k) straight into the persistent chains (no copy afterwards)
        uint8_t* v_out = (chain_v && t == k) ? chain_v : ws->vk;
        float* y_out = (chain_v && t == k) ? chain_y : ws->yk;
        // v_t and y_t both depend on h_{t-1} only: one launch (class conditional as aux workgroups)
        if (int rc = launch_visible(P, ws->h, QBM_U8, B, nullptr, nullptr, v_out, nullptr, y_out, seed, o - 2, o - 1, row0, st, hg_status,
                                    (t == k && defer_last) ? ws->sync + kHgRound : nullptr, (t == 1 && mon_in_visible) ? ws : nullptr, db))
            return rc;
        if (t == k && defer_last) break;
        // the hidden SAMPLE of the last iteration has no consumer (the gradient uses p(h_k)): no draw, no store
        if (int rc = launch_hidden(P, v_out, QBM_U8, y_out, B, ws->phk, nullptr, t < k ? ws->h : nullptr, 0, seed, o, row0, st, nullptr, hg_status, db))
            return rc;
    }
    return QBM_OK;
}

int qbm_rbm_cd_chain(const qbm_rbm_params* P, const void* v0, int v0_dtype, const float* y0, int B, int k, uint64_t seed,
                     uint64_t offset, uint32_t row0, uint8_t* chain_v, float* chain_y, const qbm_rbm_cd_workspace* ws,
                     void* stream)
{
    bool fused;
    return cd_chain_impl(P, v0, v0_dtype, y0, B, k, seed, offset, row0, chain_v, chain_y, ws, stream, &fused);
}

int qbm_rbm_cd_step(const qbm_rbm_params* P, const qbm_rbm_grads* G, const void* v0, int v0_dtype, const float* y0, int B,
                    int k, uint64_t seed, uint64_t offset, uint32_t row0, uint8_t* chain_v, float* chain_y,
                    const qbm_rbm_cd_workspace* ws, int fuse_update, float lr, float factor, float batch_size,
                    float sparse_constant, void* stream)
{
    if (!G) return fail(QBM_ERR_ARG, "cd_step: grads is null");
    if (!fuse_update && (!G->gW || !G->gb || !G->gd || !G->gU || !G->gc))
        return fail(QBM_ERR_ARG, "cd_step: all gradient blocks are required without fuse_update");
    if (int rc = check_params(P)) return rc;
    bool fused = false;
    // the training loop's monitor (ws->mon_*): p(y|x0) under the parameters BEFORE this step's update (what
    // discriminative_training reports too, src/ClassificationRBM.py:108, :142) and its two loss terms
    const bool mon = ws && ws->mon_labels && ws->mon_pre && ws->mon_p && ws->mon_rowloss && ws->mon_loss_acc;
    // The tail of the step -- last hidden phase, gradient, update -- as ONE launch (k_hgrad) where the shape allows; every
    // launch of the step then tests the workspace's status word (a fused tail that gave up is FINISHED by the host, from
    // the workspace as that step left it: qbm_rbm_cd_step_finish)
    const uint8_t* v1 = (ws && chain_v) ? chain_v : (ws ? ws->vk : nullptr);
    const float* y1 = (ws && chain_y) ? chain_y : (ws ? ws->yk : nullptr);
    const bool hg = B > 0 && batch_size > 0.f && v0 && y0 && ws && ws->ph0 && ws->phk && v1 && y1 &&
                    hgrad_eligible(P, G, v0, v0_dtype, y0, B, v1, y1, ws, fuse_update);
    // FAIL-SAFE for the steps the fused launches do NOT take: wherever the switch would hand a step of this
    // kind to a launch that can give up (the caller's record of pending steps, rbm_ops.HipRbmOps._may_fuse, asks the same
    // question and ignores the shape as well), a step whose SHAPE is not eligible -- the short last minibatch of an epoch --
    // runs one launch per phase with every storing launch reading the status word: behind a step that gave up it is a no-op
    // like the eligible ones around it, and the host's replay repeats it in order.
    const int mode = fused_chain_mode();
    const bool may_fuse = ws && ws->sync && v0_dtype == QBM_U8 && !ws->draw_base && (mode >= 2 || (mode == 1 && k >= 2));
    const uint32_t* hg_status = (hg || may_fuse) ? ws->sync + 512 : nullptr;
    // (the monitor's rows ride in the first k_visible launch as aux workgroups where that launch exists and B allows)
    const bool mon_vis = mon && visible_takes_monitor(B);
    if (int rc = cd_chain_impl(P, v0, v0_dtype, y0, B, k, seed, offset, row0, chain_v, chain_y, ws, stream, &fused,
                               mon ? ws->mon_pre : nullptr, hg, hg_status, mon_vis))
        return rc;
    if (B == 0) return zero_grads(P, G, (hipStream_t)stream);
    if (!(batch_size > 0.f)) return fail(QBM_ERR_ARG, "cd_step: batch_size must be positive");
    if (mon && (fused || !mon_vis)) {
        hipStream_t st = (hipStream_t)stream;
        // (the one-launch chain stores v0.W + d in its first hidden phase, like the launch-per-phase path)
#define QBM_L(CM) \
    launch_free_energy<CM>(ws->mon_pre, P->U, P->c, nullptr, ws->mon_p, B, P->H, P->C, ws->mon_labels, nullptr, ws->mon_rowloss, hg_status, st)
        QBM_DISPATCH_CM(P->C, QBM_L);
#undef QBM_L
        QBM_CHECK_LAUNCH("k_free_energy");
    }
    if (hg && !fused)
        return launch_hgrad(P, G, v0, y0, B, seed, offset, offset + 3ull * (uint64_t)k, row0, v1, y1, ws, fuse_update, lr, factor,
                            batch_size, sparse_constant, mon ? ws->mon_rowloss : nullptr, mon ? ws->mon_loss_acc : nullptr,
                            (hipStream_t)stream, 0, 0u);
    qbm_rbm_grads g = *G;
    if (fuse_update) g.gW = nullptr;  // the whole update happens inside the gradient launch
    // PCD: (v_k, y_k) live in the persistent chains.  Behind the one-launch chain the gradient launch reads the chain's status
    // word: a chain that gave up (or an earlier one on this workspace that did) never reaches gradients or parameters.
    return launch_grad(P->V, P->H, P->C, B, v0, v0_dtype, y0, ws->ph0, chain_v ? chain_v : ws->vk, QBM_U8,
                       chain_v ? chain_y : ws->yk, ws->phk, &g, fuse_update ? P->W : nullptr, fuse_update ? P : nullptr, lr,
                       factor, batch_size, sparse_constant, (hipStream_t)stream, nullptr, fused ? ws->sync + 512 : hg_status, nullptr,
                       mon ? ws->mon_rowloss : nullptr, mon ? ws->mon_loss_acc : nullptr);
}

int qbm_rbm_dp_step(const qbm_rbm_params* P, const qbm_rbm_grads* G, const void* v0, int v0_dtype, const float* y0, int B,
                    int k, uint64_t seed, uint64_t offset, uint32_t row0, uint8_t* chain_v, float* chain_y,
                    const qbm_rbm_cd_workspace* ws, int apply_pending, float lr, float factor, float sparse_constant, void* stream)
{
    if (!G || !G->gW || !G->gb || !G->gd || !G->gU || !G->gc || !G->n_rows)
        return fail(QBM_ERR_ARG, "dp_step: all gradient blocks and n_rows are required");
    if (apply_pending)   // the update of the step before, from the all-reduced buffer (its row count read on the device)
        if (int rc = qbm_rbm_update_dev(P, G, lr, factor, 0.f, G->n_rows, sparse_constant, nullptr, stream)) return rc;
    return qbm_rbm_cd_step(P, G, v0, v0_dtype, y0, B, k, seed, offset, row0, chain_v, chain_y, ws, 0, lr, factor, 1.0f,
                           sparse_constant, stream);
}

int qbm_rbm_cd_step_finish(const qbm_rbm_params* P, const qbm_rbm_grads* G, const void* v0, int v0_dtype, const float* y0, int B,
                           int k, uint64_t seed, uint64_t offset, uint32_t row0, uint8_t* chain_v, float* chain_y,
                           const qbm_rbm_cd_workspace* ws, int fuse_update, float lr, float factor, float batch_size,
                           float sparse_constant, uint32_t round, void* stream)
{
    if (!G || !ws) return fail(QBM_ERR_ARG, "cd_step_finish: null argument");
    if (int rc = check_params(P)) return rc;
    if (k < 1 || B <= 0 || !(batch_size > 0.f)) return fail(QBM_ERR_ARG, "cd_step_finish: bad k / B / batch_size");
    const uint8_t* v1 = chain_v ? chain_v : ws->vk;
    const float* y1 = chain_y ? chain_y : ws->yk;
    // (eligibility is asked as the step asked it: the switch may be off by now)
    const bool hg = v0 && y0 && ws->ph0 && ws->phk && v1 && y1 && hgrad_eligible(P, G, v0, v0_dtype, y0, B, v1, y1, ws, fuse_update, true);
    if (!hg) return fail(QBM_ERR_ARG, "cd_step_finish: not a step the fused tail takes");
    const bool mon = ws->mon_labels && ws->mon_pre && ws->mon_p && ws->mon_rowloss && ws->mon_loss_acc;
    for (int mode = 1; mode <= 2; ++mode)
        if (int rc = launch_hgrad(P, G, v0, y0, B, seed, offset, offset + 3ull * (uint64_t)k, row0, v1, y1, ws, fuse_update, lr,
                                  factor, batch_size, sparse_constant, mon ? ws->mon_rowloss : nullptr,
                                  mon ? ws->mon_loss_acc : nullptr, (hipStream_t)stream, mode, round))
            return rc;
    return QBM_OK;
}

int qbm_rbm_cd_tail(const qbm_rbm_params* P, const qbm_rbm_grads* G, const void* v0, int v0_dtype, const float* y0, int B,
                    const uint8_t* vk, const float* yk, const qbm_rbm_cd_workspace* ws, int fuse_update, float lr, float factor,
                    float batch_size, float sparse_constant, uint32_t round, void* stream)
{
    if (!G || !ws) return fail(QBM_ERR_ARG, "cd_tail: null argument");
    if (int rc = check_params(P)) return rc;
    if (!fuse_update && (!G->gW || !G->gb || !G->gd || !G->gU || !G->gc))
        return fail(QBM_ERR_ARG, "cd_tail: all gradient blocks are required without fuse_update");
    if (round == 0u || B <= 0 || !(batch_size > 0.f)) return fail(QBM_ERR_ARG, "cd_tail: bad round / B / batch_size");
    if (!(v0 && y0 && vk && yk && ws->ph0 && ws->phk && hgrad_eligible(P, G, v0, v0_dtype, y0, B, vk, yk, ws, fuse_update)))
        return fail(QBM_ERR_ARG, "cd_tail: not a shape the fused tail takes (or qbm_rbm_fused_chain_mode is 0)");
    return launch_hgrad(P, G, v0, y0, B, 0, 0, 0, 0, vk, yk, ws, fuse_update, lr, factor, batch_size, sparse_constant, nullptr,
                        nullptr, (hipStream_t)stream, 0, round);
}

int qbm_test_hgrad_fault(int workgroup, int kind)
{
    g_hgrad_fault_kind.store(kind);
    g_hgrad_fault_wg.store(workgroup);
    return QBM_OK;
}

int qbm_rbm_disc_step(const qbm_rbm_params* P, const qbm_rbm_grads* G, const float* x, const int32_t* labels, int B,
                      float* p_y, float* neg_free_energy, const qbm_rbm_disc_workspace* ws, int apply_update,
                      int fuse_update, float lr, float factor, float batch_size, float sparse_constant, void* stream)
{
    return qbm_rbm_disc_step_ex(P, G, x, QBM_F32, labels, B, p_y, neg_free_energy, ws, apply_update, fuse_update, lr, factor,
                                batch_size, sparse_constant, stream);
}

int qbm_rbm_disc_step_ex(const qbm_rbm_params* P, const qbm_rbm_grads* G, const void* x, int x_dtype, const int32_t* labels,
                         int B, float* p_y, float* neg_free_energy, const qbm_rbm_disc_workspace* ws, int apply_update,
                         int fuse_update, float lr, float factor, float batch_size, float sparse_constant, void* stream)
{
    if (int rc = check_params(P)) return rc;
    if (int rc = check_dtype(x_dtype)) return rc;
    if (B < 0) return fail(QBM_ERR_ARG, "B < 0");
    if (B == 0) return zero_grads(P, G, (hipStream_t)stream);
    if (!G || !G->gd || !G->gU || !G->gc) return fail(QBM_ERR_ARG, "disc_step: gd, gU, gc are required");
    if (!x || !labels || !p_y || !ws || !ws->pre || !ws->diff || !ws->rowloss)
        return fail(QBM_ERR_ARG, "disc_step: null buffer");
    const bool fuse = apply_update && fuse_update;
    if (!fuse && !G->gW) return fail(QBM_ERR_ARG, "disc_step: gW is required unless the update is fused");
    if (apply_update && !(batch_size > 0.f)) return fail(QBM_ERR_ARG, "disc_step: batch_size must be positive");
    hipStream_t st = (hipStream_t)stream;
    // Three launches:
    //   1. pre = d + x.W                                                          (MFMA strips)
    //   2. per row: -F, p(y|x), diff = pos - neg, the row's two loss terms          (one workgroup per row)
    //   3. gW = x^T.diff (ONE contraction over the batch), gd = colsum(diff) (strips); gU, gc, the minibatch's loss terms
    //      (kind D workgroups, behind the strips: sigma(pre + U_c) is evaluated there once more rather than stored as a
    //      [B,C,H] array); every update fused
    if (int rc = launch_hidden(P, x, x_dtype, nullptr, B, ws->pre, nullptr, nullptr, 1, 0, 0, 0, st)) return rc;
    // ws->status (hybrid training: the CD step behind this one may hand its chain to a launch that can give up): every
    // storing launch reads it -- behind a step that gave up this one is a no-op and the host repeats it in order
#define QBM_L(CM) \
    launch_free_energy<CM>(ws->pre, P->U, P->c, neg_free_energy, p_y, B, P->H, P->C, labels, ws->diff, ws->rowloss, ws->status, st)
    QBM_DISPATCH_CM(P->C, QBM_L);
#undef QBM_L
    QBM_CHECK_LAUNCH("k_free_energy");
    qbm_rbm_grads g = *G;
    g.gb = nullptr;  // reference: visible_bias_grad = 0 (src/ClassificationRBM.py:138)
    g.gU = nullptr;  // class-weight and class-bias gradients come from the kind D workgroups
    g.gc = nullptr;
    if (fuse) g.gW = nullptr;
    const DiscExtras dx{ws->pre, P->U, p_y, labels, ws->rowloss, ws->loss_acc, G->gU, G->gc,
                        fuse ? P->U : nullptr, fuse ? P->c : nullptr, fuse ? P->b : nullptr};
    qbm_rbm_params aux{nullptr, nullptr, P->d, nullptr, nullptr, P->V, P->H, P->C};
    if (int rc = launch_grad(P->V, P->H, P->C, B, x, x_dtype, nullptr, ws->diff, nullptr, x_dtype, nullptr, nullptr, &g,
                             fuse ? P->W : nullptr, fuse ? &aux : nullptr, lr, factor, batch_size, sparse_constant, st,
                             nullptr, ws->status, &dx))
        return rc;
    if (apply_update && !fuse) {
        g.gU = G->gU;
        g.gc = G->gc;
        return qbm_rbm_update_dev(P, &g, lr, factor, batch_size, nullptr, sparse_constant, ws->status, stream);
    }
    return QBM_OK;
}

#ifdef QBM_DIAG_STAMPS
// diagnostic library only (tools/stamp_kernels.py): the s_memtime stamps of the last launches, 16 slots per wave
QBM_API int qbm_diag_read_stamps(unsigned long long* host_out, int n_waves)
{
    if (n_waves > kStampWaves) n_waves = kStampWaves;
    hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16 * (size_t)n_waves);
    return e == hipSuccess ? QBM_OK : hip_fail(e, "hipMemcpyFromSymbol(g_stamps)");
}
QBM_API int qbm_diag_clear_stamps(void)
{
    static unsigned long long zeros[kStampWaves * 16];
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zeros, sizeof(zeros));
    return e == hipSuccess ? QBM_OK : hip_fail(e, "hipMemcpyToSymbol(g_stamps)");
}
#endif

int qbm_test_math(int which, const float* x, float* y, int n, void* stream)
{
    if (n <= 0) return QBM_OK;
    hipLaunchKernelGGL(k_math, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, which, x, y, n);
    QBM_CHECK_LAUNCH("k_math");
    return QBM_OK;
}

int qbm_test_uniform(uint64_t seed, uint64_t offset, uint32_t row0, int rows, int cols, float* out, void* stream)
{
    if (rows <= 0 || cols <= 0) return QBM_OK;
    const long n = (long)rows * ((cols + 3) / 4);
    hipLaunchKernelGGL(k_uniform, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed, offset,
                       row0, rows, cols, out);
    QBM_CHECK_LAUNCH("k_uniform");
    return QBM_OK;
}

}  // extern "C"
