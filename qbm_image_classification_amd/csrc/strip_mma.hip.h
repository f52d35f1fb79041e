// strip_mma.hip.h -- the K-split strip contraction every dense product of the path runs on.
//
// One workgroup = NQ wavefronts (4: one per SIMD of a CU; 8: two per SIMD).  It owns a (16*MT) x (16*NT) strip
// of the output.  Wave q contracts K chain q ([Q] of the numerical contract) with
// v_mfma_f32_16x16x4_f32, MT*NT independent accumulators per wave (>= 2 reaches the MFMA issue
// rate: 32-cycle issue vs 40-cycle dependent latency).  The NQ partial strips meet in LDS and
// are added (((q0+q1)+q2)+...); the caller's epilogue then runs on a row-major thread map over ALL threads
// (StripOut: thread -> one row, CPT consecutive columns) so that one Philox4x32 call serves one thread and
// stores are up to 16 B wide.
//
// What bounds these kernels on MI355X is the L2 -> LDS staging of the operand strips (measured with
// s_memtime stamps, tools/stamp_kernels.py), not the MFMA pipe, so the layout minimises
// staged bytes per workgroup:
//   * the B operand (W, or the probabilities in the gradient) is fp32;
//   * the A operand may be uint8: Gibbs states are {0,1}, a byte per unit is exact and 4x smaller;
//     it is widened to fp32 in registers (v_cvt_f32_ubyte) right before the MFMA;
//   * MT = 2, NT = 1: two row tiles share every staged B fragment.
//
// Operand tiles go through LDS in images filled by LDS-DMA (global_load_lds_dwordx4: 16 B per lane,
// no VGPR round trip, wave-uniform LDS base + lane*16, per-lane global address):
//   fp32 "k4"  image (K contiguous in memory): float4 slot piece*64 + (k4 in piece)*16 + r
//               -> MFMA fragment of k-step s: dword s*64 + (lane&15)*4 + (lane>>4)
//   fp32 "lin" image (M/N contiguous):          dword k*16 + r -> fragment dword s*64 + lane
//   u8   "k4"  image: 16-byte slot piece*16 + r = 16 consecutive k of one row -> one ds_read_b128 per
//               piece, byte (lane>>4) of dword s is k-step s
//   u8   "lin" image: 16-byte slot k*MT + tile = 16 consecutive m of one row tile (the MT row tiles of a strip are
//               adjacent in memory, so ONE transfer instruction covers 64/MT k for all of them with 16*MT-byte segments:
//               a quarter of the cache-line requests of a tile-by-tile transfer at MT = 4)
//               -> ds_read_u8 at ((4s + (lane>>4))*MT + tile)*16 + (lane&15)
// A "piece" is 16 consecutive k (4 MFMA k-steps).  Out-of-range k / rows are staged as zeros from a
// 16-byte zero page, so the MFMA loop needs no masks (fmaf(0, b, acc) == acc).
#pragma once
#include "qbm_device.hip.h"
#include <type_traits>

// Diagnostic builds record s_memtime at phase boundaries: tools/microbench defines its own QBM_STAMP(i), and
// -DQBM_DIAG_STAMPS (tools/stamp_kernels.py builds the real kernels that way into a library of its own, never into
// libqbm_hip.so) stamps slot i of 16 per wave.  In the product QBM_STAMP expands to nothing.
#ifdef QBM_DIAG_STAMPS
#include <hip/hip_runtime.h>
namespace qbm { constexpr int kStampWaves = 8192; static __device__ unsigned long long g_stamps[kStampWaves * 16]; }
#define QBM_STAMP(i)                                                                                          \
    do {                                                                                                      \
        unsigned long long t__;                                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                          \
        const unsigned w__ = (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * (blockDim.x >> 6) + (threadIdx.x >> 6); \
        if ((threadIdx.x & 63) == 0 && w__ < (unsigned)qbm::kStampWaves) qbm::g_stamps[w__ * 16 + (i)] = t__; \
    } while (0)
#endif
#ifndef QBM_STAMP
#define QBM_STAMP(i)
#endif
// diagnostic builds may also define QBM_DIAG_MODE as an int expression: 1 = one vmcnt(0) before the MFMA loop
// instead of the half-way counted wait, 3 = no MFMA loop, 4 = no staging and no waits (the MFMA loop alone, on
// whatever the LDS holds).  0 in the product (the branches fold away).
#ifndef QBM_DIAG_MODE
#define QBM_DIAG_MODE 0
#endif
// diagnostic builds may pin how many quads of LDS-DMA transfers are requested ahead of the quad being contracted (1 or 2);
// 0 = chosen from the chain length (the product)
#ifndef QBM_DMA_AHEAD
#define QBM_DMA_AHEAD 0
#endif


namespace qbm {

// 16 zero bytes: the LDS-DMA source of every out-of-range fragment
static __device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// LDS-DMA through a buffer descriptor: 16 B per lane, global -> LDS with no VGPR destination
// (buffer_load_dwordx4 ... offen lds).  The LDS address is the wave-uniform `lds_piece` (M0) + lane*16;
// `voff` is this lane's byte offset into the buffer.  An offset at or beyond the descriptor's size reads
// as zeros: that is how rows beyond the operand and the tail of a K quarter are staged, with no
// per-lane address arithmetic beyond one add per instruction.
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr unsigned kOOB = 0x80000000u;  // operands are < 2 GiB (checked on the host), so this is out of range

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

// byte offset of an LDS object inside the workgroup's LDS (the address operand of ds_* instructions)
__device__ __forceinline__ uint32_t lds_off(const void* p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

// AUX = cache policy of the transfer: 0 = plain, kSc1 = `sc1` (served by the XCD's L2, never by this CU's vector L1: what a
// workgroup must use for bytes that another CU of its XCD has written during the same launch)
constexpr int kSc1 = 16;
template <int AUX = 0>
__device__ __forceinline__ void blds16(rsrc_t r, unsigned voff, void* lds_piece)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_piece, 16, voff, 0, 0, AUX);
}

// this lane's walk over one 16-wide tile of an operand: byte offset of its 16 B in the first DMA
// instruction, the advance per instruction, and the K index it starts at (for the tail test of u8 quads)
struct DmaLane {
    unsigned voff, step;
    int k, kstep;
};

// ---- operand views -------------------------------------------------------------------------------
// rsrc()/lane0(): buffer descriptor and per-lane walk of the LDS-DMA (FAST) path; the host selects it only
//         when every 16-byte unit is aligned and wholly inside or wholly outside the operand.
// get():  the same 16 bytes as a value, branch-free (clamped indices + selects), for the generic path.
//   `r0` = first row/column of the 16-wide tile, `kb` = first k staged, lane = 0..63.

// fp32, element(row, k) = p[row*ld + k]  (K contiguous: "k4" image)
struct ViewKContig {
    using elem = float;
    static constexpr bool kK4 = true;
    const float* p;
    long ld;
    int rows;
    __device__ __forceinline__ rsrc_t rsrc() const { return make_rsrc(p, (unsigned)(rows * ld * 4)); }
    __device__ __forceinline__ DmaLane lane0(int r0, int kb, int lane) const  // rows beyond the operand are out of range
    {
        const int row = r0 + (lane & 15), k = kb + 4 * (lane >> 4);
        return DmaLane{(unsigned)((row * (int)ld + k) * 4), 64u, k, 16};
    }
    __device__ __forceinline__ float4 get(int r0, int kb, int kend, int slot) const  // slot = piece*64 + lane
    {
        const int row = r0 + (slot & 15), k = kb + 4 * (slot >> 4);
        const float* q = p + (long)min(row, rows - 1) * ld;
        const bool rok = row < rows;
        const int km = max(kend - 1, 0);
        float4 v;
        v.x = q[min(k, km)]; v.y = q[min(k + 1, km)]; v.z = q[min(k + 2, km)]; v.w = q[min(k + 3, km)];
        v.x = (rok && k < kend) ? v.x : 0.f;     v.y = (rok && k + 1 < kend) ? v.y : 0.f;
        v.z = (rok && k + 2 < kend) ? v.z : 0.f; v.w = (rok && k + 3 < kend) ? v.w : 0.f;
        return v;
    }
};

// fp32, element(k, c) = p[k*ld + c]  (M/N contiguous: "lin" image).  p == nullptr reads as zeros.
struct ViewCContig {
    using elem = float;
    static constexpr bool kK4 = false;
    const float* p;
    long ld;
    int cols;
    int krows;  // number of k rows the buffer holds (its size is krows*ld*4 bytes)
    __device__ __forceinline__ rsrc_t rsrc() const { return make_rsrc(p, (unsigned)(krows * ld * 4)); }
    // KPERM ([Q'] of the numerical contract): the image row that MFMA k-step s, lane group g reads (row 4s + g of a piece)
    // holds k = 4g + s instead of k = 4s + g -- a different source row per lane, the same LDS image and reads
    template <bool KPERM = false>
    __device__ __forceinline__ static int krow(int r)   // r = image row 0..15 inside a piece -> k inside the piece
    {
        return KPERM ? 4 * (r & 3) + (r >> 2) : r;
    }
    template <bool KPERM = false>
    __device__ __forceinline__ DmaLane lane0(int c0, int kb, int lane) const
    {
        const int c = c0 + 4 * (lane & 3), k = kb + krow<KPERM>(lane >> 2);
        return DmaLane{c < cols ? (unsigned)((k * (int)ld + c) * 4) : kOOB, (unsigned)(64 * ld), k, 16};
    }
    template <bool KPERM = false>
    __device__ __forceinline__ float4 get(int c0, int kb, int kend, int slot) const
    {
        const int c = c0 + 4 * (slot & 3), k = kb + 16 * (slot >> 6) + krow<KPERM>((slot >> 2) & 15);
        const float* q = p ? p + (long)min(k, max(kend - 1, 0)) * ld : g_zero16;
        const bool kok = p && k < kend;
        const int cm = p ? cols - 1 : 0;
        float4 v;
        v.x = q[min(c, cm)]; v.y = q[min(c + 1, cm)]; v.z = q[min(c + 2, cm)]; v.w = q[min(c + 3, cm)];
        v.x = (kok && c < cols) ? v.x : 0.f;     v.y = (kok && c + 1 < cols) ? v.y : 0.f;
        v.z = (kok && c + 2 < cols) ? v.z : 0.f; v.w = (kok && c + 3 < cols) ? v.w : 0.f;
        return v;
    }
};

// u8, element(row, k) = p[row*ld + k]  (K contiguous).  One DMA instruction stages 4 pieces (64 k).
struct ViewKContigU8 {
    using elem = uint8_t;
    static constexpr bool kK4 = true;
    const uint8_t* p;
    long ld;
    int rows;
    __device__ __forceinline__ rsrc_t rsrc() const { return make_rsrc(p, (unsigned)(rows * ld)); }
    __device__ __forceinline__ DmaLane lane0(int r0, int kb, int lane) const
    {
        const int row = r0 + (lane & 15), k = kb + 16 * (lane >> 4);
        return DmaLane{(unsigned)(row * (int)ld + k), 64u, k, 64};
    }
    __device__ __forceinline__ uint4 get(int r0, int kb, int kend, int slot) const  // slot = quad*64 + lane
    {
        const int row = r0 + (slot & 15), k = kb + 16 * (slot >> 4);
        const uint8_t* q = p + (long)min(row, rows - 1) * ld;
        const bool rok = row < rows;
        const int km = max(kend - 1, 0);
        uint32_t w[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            w[d] = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int kk = k + 4 * d + b;
                const uint32_t v = q[min(kk, km)];
                w[d] |= ((rok && kk < kend) ? v : 0u) << (8 * b);
            }
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// u8, element(k, c) = p[k*ld + c]  (M contiguous).  One DMA instruction stages 64 k of one 16-wide tile.
struct ViewCContigU8 {
    using elem = uint8_t;
    static constexpr bool kK4 = false;
    const uint8_t* p;
    long ld;
    int cols;
    int krows;
    __device__ __forceinline__ rsrc_t rsrc() const { return make_rsrc(p, (unsigned)(krows * ld)); }
    __device__ __forceinline__ DmaLane lane0(int c0, int kb, int lane) const
    {
        const int k = kb + lane;
        return DmaLane{c0 < cols ? (unsigned)(k * (int)ld + c0) : kOOB, (unsigned)(64 * ld), k, 64};
    }
    // transfer instruction j (0..TILES-1) of a quad, all TILES adjacent 16-wide tiles from column c0 on: the quad's image is
    // 64 k x TILES slots of 16 bytes = TILES transfers of 64 slots; lane L of transfer j fills slot s = 64 j + L, i.e.
    // k = kb + s / TILES of tile s % TILES (TILES = 1, 2, 4: k = kb + (64 / TILES) j + L / TILES of tile L % TILES)
    template <int TILES>
    __device__ __forceinline__ DmaLane lane0_tiles(int c0, int kb, int lane, int j) const
    {
        static_assert(TILES == 1 || TILES == 2 || TILES == 4 || TILES == 7, "a k row of TILES * 16 bytes: at 8 tiles the rows k and k + 1 that one ds_read_u8 reads would meet on the same banks (7: 112-byte rows, conflict-free)");
        const int s = 64 * j + lane;
        const int k = kb + s / TILES, c = c0 + 16 * (s % TILES);
        return DmaLane{c < cols ? (unsigned)(k * (int)ld + c) : kOOB, (unsigned)(64 * ld), k, 64};
    }
    __device__ __forceinline__ uint4 get(int c0, int kb, int kend, int slot) const
    {
        const int k = kb + slot;
        const uint8_t* q = p + (long)min(k, max(kend - 1, 0)) * ld;
        const bool kok = k < kend;
        uint32_t w[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            w[d] = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int c = c0 + 4 * d + b;
                const uint32_t v = q[min(c, cols - 1)];
                w[d] |= ((kok && c < cols) ? v : 0u) << (8 * b);
            }
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// Gradient A operand for the class-weight / hidden-bias rows: element(k, m) = Y[k][m] for m < C (Y may
// be null = zeros), 1 for m == C (its product with P is the column sum of P), 0 beyond.  Generic path.
struct ViewAugY {
    using elem = float;
    static constexpr bool kK4 = false;
    const float* Y;  // [B, C] or nullptr
    int C;
    __device__ __forceinline__ rsrc_t rsrc() const { return make_rsrc(nullptr, 0); }
    __device__ __forceinline__ DmaLane lane0(int, int, int) const { return DmaLane{kOOB, 0u, 0, 0}; }
    __device__ __forceinline__ float4 get(int m0, int kb, int kend, int slot) const
    {
        const int m = m0 + 4 * (slot & 3), k = kb + (slot >> 2);
        const float* q = Y ? Y + (long)min(k, max(kend - 1, 0)) * C : g_zero16;
        const bool kok = k < kend;
        const int cm = Y ? C - 1 : 0;
        float e[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float y = q[min(m + t, cm)];
            e[t] = (m + t < C) ? (Y ? y : 0.f) : ((m + t == C) ? 1.f : 0.f);
            e[t] = kok ? e[t] : 0.f;
        }
        return make_float4(e[0], e[1], e[2], e[3]);
    }
};

// Gradient A operand for the column sums of the B operand: row 0 is all ones, the other rows are zero.
struct ViewOnesRow {
    using elem = float;
    static constexpr bool kK4 = false;
    __device__ __forceinline__ rsrc_t rsrc() const { return make_rsrc(nullptr, 0); }
    __device__ __forceinline__ DmaLane lane0(int, int, int) const { return DmaLane{kOOB, 0u, 0, 0}; }
    __device__ __forceinline__ float4 get(int m0, int kb, int kend, int slot) const
    {
        const int m = m0 + 4 * (slot & 3), k = kb + (slot >> 2);
        return make_float4((k < kend && m == 0) ? 1.f : 0.f, 0.f, 0.f, 0.f);
    }
};

// lane walk / register-path value of a view, with the k rows of a piece permuted where the contract asks for it and the
// image is row-per-k (fp32 "lin"); the K-contiguous images hold whole k groups per slot: the permutation is in how they are read
template <bool KPERM, class V>
__device__ __forceinline__ DmaLane view_lane0(const V& v, int r0, int kb, int lane)
{
    if constexpr (KPERM && !V::kK4 && sizeof(typename V::elem) == 4) return v.template lane0<true>(r0, kb, lane);
    else return v.lane0(r0, kb, lane);
}
template <bool KPERM, class V>
__device__ __forceinline__ auto view_get(const V& v, int r0, int kb, int kend, int slot)
{
    if constexpr (KPERM && !V::kK4 && sizeof(typename V::elem) == 4) return v.template get<true>(r0, kb, kend, slot);
    else return v.get(r0, kb, kend, slot);
}

// ---- LDS budget ----------------------------------------------------------------------------------
// bytes of one wave's staging region for panel length KP (multiple of 64): MT A tiles + NT fp32 B tiles,
// and never smaller than the [16*MT][16*NT+4] fp32 partial strip that later reuses it
template <int MT, int NT, int ASZ>
__host__ __device__ constexpr int strip_wave_bytes(int KP)
{
    const int stage = MT * 16 * KP * ASZ + NT * 16 * KP * 4;
    const int red = 16 * MT * (16 * NT + 4) * 4;
    return stage > red ? stage : red;
}
template <int MT, int NT, int ASZ, int NQ = 4>
__host__ __device__ constexpr int strip_lds_bytes(int KP) { return NQ * strip_wave_bytes<MT, NT, ASZ>(KP); }
// the same with the B images in a region of their own (strip_mma's `ldsB`): per wave the A tiles / the partial strip here,
// NT * 16 * KP * 4 bytes of B images there
template <int MT, int NT, int ASZ>
__host__ __device__ constexpr int strip_wave_bytes_a(int KP)
{
    const int stage = MT * 16 * KP * ASZ, red = 16 * MT * (16 * NT + 4) * 4;
    return stage > red ? stage : red;
}
template <int NT>
__host__ __device__ constexpr int strip_wave_bytes_b(int KP) { return NT * 16 * KP * 4; }

// largest panel (multiple of 64) that fits 160 KiB (less `reserve` bytes) with NQ waves and keeps the DMA count under
// vm_wait's range
template <int MT, int NT, int ASZ, int NQ = 4>
__host__ __device__ constexpr int strip_max_panel(int reserve = 0)
{
    int kp = ((163840 - reserve) / NQ) / (MT * 16 * ASZ + NT * 16 * 4) / 64 * 64;
    const int per16 = (ASZ == 4 ? MT : 0) + NT;                     // DMA instructions per piece (u8 A: MT per 4 pieces)
    while (kp > 64 && (kp / 16) * per16 + (ASZ == 1 ? MT * (kp / 64) : 0) > 60) kp -= 64;
    return kp;
}

// What one thread holds after the four partial strips are combined.  CPT = consecutive columns per output (4, or 2
// for strips so small that 4 columns per thread would leave half of the workgroup without an epilogue):
// NP = ceil(256*MT*NT / (CPT*THREADS)) outputs; output p of thread t is row (t + THREADS*p) / (16*NT/CPT), column group
// (t + THREADS*p) % (16*NT/CPT) of the strip (columns CPT*cg .. CPT*cg + CPT - 1), sum[p] = (..((q0+q1)+q2)+..)+q_{NQ-1}
// (CPT == 2: in .x, .y; CPT == 1: in .x).
template <int MT, int NT, int CPT = 4, int THREADS = 256>
struct StripOut {
    static_assert(CPT == 4 || CPT == 2 || CPT == 1, "4, 2 or 1 columns per thread");
    static constexpr int kCG = 16 * NT / CPT;                       // column groups per strip row
    static constexpr int NP = (16 * MT * kCG + THREADS - 1) / THREADS;
    float4 sum[NP];
    // (threadIdx.x & (THREADS - 1): the thread's index inside the group of THREADS/64 wavefronts that runs the contraction;
    // a 512-thread workgroup of k_grad is two groups of 256, each with a contraction of its own)
    __device__ __forceinline__ static int t(int p) { return ((int)threadIdx.x & (THREADS - 1)) + THREADS * p; }
    __device__ __forceinline__ static bool valid(int p) { return t(p) < 16 * MT * kCG; }
    __device__ __forceinline__ static int row(int p) { return t(p) / kCG; }
    __device__ __forceinline__ static int cg(int p) { return t(p) % kCG; }
};

// hook of strip_mma that does nothing
struct NoPre {
    __device__ __forceinline__ void operator()() const {}
    __device__ __forceinline__ void operator()(int) const {}
};

// wait until at most n of this wave's vector-memory operations (LDS-DMA pieces included) are outstanding.
// The compiler does not order a ds_read behind a pending LDS-DMA (only a barrier makes it drain vmcnt),
// so every consumer of DMA-staged data sits behind one of these.
__device__ __forceinline__ void vm_wait(int n)
{
#define QBM_VMW(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        QBM_VMW(0) QBM_VMW(1) QBM_VMW(2) QBM_VMW(3) QBM_VMW(4) QBM_VMW(5) QBM_VMW(6) QBM_VMW(7) QBM_VMW(8) QBM_VMW(9)
        QBM_VMW(10) QBM_VMW(11) QBM_VMW(12) QBM_VMW(13) QBM_VMW(14) QBM_VMW(15) QBM_VMW(16) QBM_VMW(17) QBM_VMW(18)
        QBM_VMW(19) QBM_VMW(20) QBM_VMW(21) QBM_VMW(22) QBM_VMW(23) QBM_VMW(24) QBM_VMW(25) QBM_VMW(26) QBM_VMW(27)
        QBM_VMW(28) QBM_VMW(29) QBM_VMW(30) QBM_VMW(31) QBM_VMW(32) QBM_VMW(33) QBM_VMW(34) QBM_VMW(35) QBM_VMW(36)
        QBM_VMW(37) QBM_VMW(38) QBM_VMW(39) QBM_VMW(40) QBM_VMW(41) QBM_VMW(42) QBM_VMW(43) QBM_VMW(44) QBM_VMW(45)
        QBM_VMW(46) QBM_VMW(47) QBM_VMW(48) QBM_VMW(49) QBM_VMW(50) QBM_VMW(51) QBM_VMW(52) QBM_VMW(53) QBM_VMW(54)
        QBM_VMW(55) QBM_VMW(56) QBM_VMW(57) QBM_VMW(58) QBM_VMW(59) QBM_VMW(60)
    default: break;  // > 60 allowed outstanding: nothing this wave has issued needs to have landed yet
    }
#undef QBM_VMW
}

// KP: panel length, multiple of 64.  lds: the workgroup's dynamic LDS (strip_lds_bytes<MT,NT,sizeof(elem)>(KP)).
// FAST_A / FAST_B: the operand is staged by LDS-DMA (the host guarantees alignment and whole 16-byte units);
// otherwise its values go through registers (any shape, any alignment, synthesised operands).
// Two hooks, each called ONCE by every wave, for the caller's contraction-independent work (bias / class-term loads, the
// Philox draw), so that it runs while the first operand tiles are in flight instead of in front of them -- a kernel's
// first transfers miss every cache and are what its start waits for, so they go out before anything else:
//   pre0()  right after the transfers of the first quad have been issued: the place to ISSUE loads;
//   pre(n)  right after those of the second quad (n = vector-memory instructions this wave has issued since pre0):
//           the place for arithmetic, and for consuming pre0's loads behind vm_wait(n).
// On the register-staged path both run first (n = 0).
//
// NQ wavefronts (4 or 8) contract the strip: wavefront q takes chain q of the contraction, chain_len(K, NQ) terms.  NQ = 8 =
// two wavefronts per SIMD, each with transfers, LDS reads and MFMAs of its own and no synchronisation before the partial
// strips are combined: while one is parked on a transfer or an LDS read the other issues MFMAs, and the VALU work around
// the contraction (Philox, sigmoid) issues at the two-wave rate.  (NQ is part of the numerical contract: [Q] in DESIGN.md.)
//
// KPERM ([Q'] in DESIGN.md, the conditionals): inside every 16-k piece MFMA k-step s multiplies k = 4g + s (g = lane / 16)
// instead of k = 4s + g, i.e. the chain visits k = 0,4,8,12, 1,5,9,13, ... of the piece.  A lane's four k-steps are then
// ADJACENT in a K-contiguous image: one ds_read_b32 (u8: + v_cvt_f32_ubyte0..3) or one ds_read_b128 (fp32) per tile and
// piece instead of four byte reads / two ds_read2st64; a row-per-k fp32 image is filled with its rows permuted (a per-lane
// source address, free) and read as before.
// AUX_A / AUX_B: cache policy of the A / B operand's transfers (kSc1 where the operand is a state another workgroup of the
// XCD has written during this launch: the fused Gibbs chain's A operands, the p(h_k) panel of the fused hidden + gradient
// launch; LDS-DMA operands only).
// PHASE: the contraction in two calls, around other work of the wavefront (k_hgrad: the gradient's operands are requested
// under the tail of the hidden phase and contracted behind the barrier).  LDS-DMA operands, ONE panel per chain (KP >= the
// chain length, at most 60 transfer instructions):
//   kIssueAB / kIssueA  request all of the chain's A and B (only A) transfers, call pre0(), return (the result is unspecified);
//   kIssueB  request the B transfers of a chain whose A transfers are out already, return;
//   kConsume / kIssueBConsume  (request the B transfers,) wait until at most `younger` of this wavefront's transfers are in
//   flight (those requested AFTER this chain's: the caller counts them, chain_transfers_b below), contract, combine.
// `post()`: called once between the contraction and the combine (every transfer of this call has been consumed).
constexpr int kWhole = 0, kIssueAB = 1, kConsume = 2, kIssueA = 3, kIssueBConsume = 4, kIssueB = 5;
// B transfers one wavefront issues for its chain of a K-long contraction in the two-call form (one panel)
template <int NT, int NQ = 4>
__device__ __forceinline__ int chain_transfers_b(int K, int KP)
{
    const int q = __builtin_amdgcn_readfirstlane(((int)threadIdx.x & (64 * NQ - 1)) >> 6);
    int k0, k1;
    chain_range(K, NQ, q, k0, k1);
    const int klen = max(0, min(k1 - k0, KP));
    return NT * ((klen + 15) >> 4);
}
// PW ("piece waits"; the gradient's single-quad chains: a wavefront's whole chain is ONE quad of 64 k, so nothing of its
// staging can hide under its own MFMAs): the quad's transfers are requested piece by piece (the merged uint8 image's transfer
// i IS piece i of all row tiles) and every piece is awaited by itself, so that the MFMAs of piece p run while pieces p + 2 ..
// are still on their way (the fragment reads of piece p + 1 stand in front of the MFMAs of piece p).
#ifndef QBM_PIECE_WAIT   // diagnostic builds: 0 = one wait for the whole quad (A/B timing)
#define QBM_PIECE_WAIT 1
#endif
template <int MT, int NT, bool FAST_A, bool FAST_B, int CPT = 4, int NQ = 4, bool KPERM = false, int AUX_A = 0, int AUX_B = 0,
          int PHASE = kWhole, int STAG = 0, bool PW = false, class AView, class BView, class Pre0 = NoPre, class Pre = NoPre, class Post = NoPre>
__device__ __forceinline__ StripOut<MT, NT, CPT, 64 * NQ> strip_mma(const AView& A, const BView& Bv, int m0, int n0, int K,
                                                                    char* __restrict__ lds, int KP, Pre0&& pre0 = NoPre{},
                                                                    Pre&& pre = NoPre{}, int stamp0 = 0,
                                                                    char* __restrict__ ldsB = nullptr, bool b_resident = false,
                                                                    Post&& post = NoPre{}, int younger = 0,
                                                                    bool stag_on = true)
{
    static_assert(PHASE == kWhole || (FAST_A && FAST_B), "the two-call form takes LDS-DMA operands");
    // ldsB != nullptr: the B images live in a region of their own (NQ * strip_wave_bytes_b<NT>(KP) bytes) instead of behind
    // each wave's A tiles; b_resident: they are already there (the caller ran this very contraction's B operand through
    // this function before, with one panel per chain, and nothing has overwritten the region): no B transfers are issued.
    static_assert(NQ == 4 || NQ == 8, "4 or 8 chains");
    static_assert(AUX_A == 0 || FAST_A, "a cache policy only applies to LDS-DMA transfers");
    static_assert(AUX_B == 0 || FAST_B, "a cache policy only applies to LDS-DMA transfers");
    (void)stamp0;   // first stamp slot of this call (diagnostic builds only)
    using TA = typename AView::elem;
    const int qbm_diag = QBM_DIAG_MODE;   // 0 in the product: the diagnostic branches fold away
    constexpr bool A8 = sizeof(TA) == 1;
    constexpr bool A_K4 = AView::kK4, B_K4 = BView::kK4;
    static_assert(!KPERM || A_K4, "the permuted contract is implemented for K-contiguous A operands (the conditionals)");
    constexpr int ASZ = (int)sizeof(TA);
    static_assert(sizeof(typename BView::elem) == 4, "the B operand is fp32");
    const int tid = threadIdx.x & (64 * NQ - 1);   // the thread's index inside its group of NQ wavefronts
    const int lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave id == K chain (wave-uniform, SGPR)
    int kq0, kend;                                  // this wavefront's chain: k in [kq0, kend), whole pieces ([Q])
    chain_range(K, NQ, q, kq0, kend);
    const int wave_bytes = ldsB ? strip_wave_bytes_a<MT, NT, ASZ>(KP) : strip_wave_bytes<MT, NT, ASZ>(KP);
    char* img = lds + (size_t)q * wave_bytes;
    char* imgA = img;                              // MT tiles of 16*KP*ASZ bytes
    char* imgB = ldsB ? ldsB + (size_t)q * strip_wave_bytes_b<NT>(KP) : img + (size_t)MT * 16 * KP * ASZ;  // NT tiles of 16*KP*4 bytes
    const int nt_dma = b_resident ? 0 : NT;        // B tiles transferred per piece
    // u8 "lin" A image: the MT row tiles interleaved per k (see the header); every other image: one region per tile
    constexpr bool A_TILES = A8 && !A_K4;
    const int a_tile = A_TILES ? 16 : 16 * KP * ASZ, b_tile = 16 * KP * 4;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    QBM_STAMP(stamp0 + 0);

    // byte offset of this lane's fragment in a piece.  KPERM, K-contiguous images: the lane's 4 k-steps are the 4 bytes
    // (u8: dword g of row r's 16-byte slot) or the 4 dwords (fp32: slot g*16 + r) it reads in one instruction
    const int fragB = (KPERM && B_K4) ? ((lane >> 4) * 16 + (lane & 15)) * 16 : (B_K4 ? ((lane & 15) * 4 + (lane >> 4)) : lane) * 4;
    const int fragA = KPERM ? (A8 ? (lane & 15) * 16 + (lane >> 4) * 4 : ((lane >> 4) * 16 + (lane & 15)) * 16)
                    : A8    ? (A_K4 ? (lane & 15) * 16 + (lane >> 4) : (lane >> 4) * 16 * MT + (lane & 15))  // u8: see header
                            : (A_K4 ? ((lane & 15) * 4 + (lane >> 4)) : lane) * 4;

    // ---- MFMA loop over one "quad" (up to 4 pieces = 64 k).
    // f32 MFMA runs at the vector rate and every VALU instruction of the wave displaces ~6-8 cycles of it (measured,
    // tools/microbench/mfma_rate.hip), so the loop keeps VALU work to one widening per u8 value: fragment reads are
    // asm with immediate piece offsets off one base register per tile, a u8 operand is read as single bytes
    // (ds_read_u8 zero-extends: no shift), and the widenings of a piece go as one burst in front of its MFMAs.
    // The reads are asm also so that the compiler's wait-count pass, which orders every LDS read it can see behind
    // ALL pending LDS-DMA (vmcnt(0)), stays out of the loop: the transfers of the following quads land under these
    // MFMAs.  The reads of piece p+1 are issued before the MFMAs of piece p behind counted lgkmcnt waits (LDS
    // returns in order; a wait after the first <= 15 reads of p+1 covers everything older).
    constexpr bool B_PERM = KPERM && B_K4;
    constexpr int kReadsA = KPERM ? 1 : (A8 ? 4 : 2), kReadsB = B_PERM ? 1 : 2;
    constexpr int MT1 = (15 - kReadsB * NT) / kReadsA < MT ? (15 - kReadsB * NT) / kReadsA : MT;  // tiles in chunk 1
    constexpr int kChunk1 = kReadsB * NT + kReadsA * MT1;   // <= 15 = the largest lgkmcnt
    constexpr int kPieceA = A8 ? (A_TILES ? 256 * MT : 256) : 1024;   // bytes from piece to piece at one tile's base address
    struct Frag {
        u32x2 a2[MT][2];     // fp32: k-steps (0,1) and (2,3)
        uint32_t a1[MT][4];  // u8: one zero-extended byte per k-step
        u32x2 b[NT][2];
        uint32_t a4[MT];     // KPERM u8: the four k-steps in one dword
        u32x4 a16[MT];       // KPERM fp32: the four k-steps
        u32x4 b16[NT];       // KPERM, K-contiguous B
    };
    auto readA = [&](auto PC, uint32_t ad, int i, Frag& f) {
        constexpr int pc = decltype(PC)::value;
        if constexpr (KPERM && A8) {
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f.a4[i]) : "v"(ad), "n"(pc * kPieceA) : "memory");
        } else if constexpr (KPERM) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.a16[i]) : "v"(ad), "n"(pc * kPieceA) : "memory");
        } else if constexpr (A8) {
            constexpr int st = A_K4 ? 4 : 64 * MT;   // k4 image: byte 4s+g of the row; lin image: 16-byte slot (4s+g)*MT + tile
            asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(f.a1[i][0]) : "v"(ad), "n"(pc * kPieceA) : "memory");
            asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(f.a1[i][1]) : "v"(ad), "n"(pc * kPieceA + st) : "memory");
            asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(f.a1[i][2]) : "v"(ad), "n"(pc * kPieceA + 2 * st) : "memory");
            asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(f.a1[i][3]) : "v"(ad), "n"(pc * kPieceA + 3 * st) : "memory");
        } else {
            asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(f.a2[i][0]) : "v"(ad), "n"(4 * pc), "n"(4 * pc + 1) : "memory");
            asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(f.a2[i][1]) : "v"(ad), "n"(4 * pc + 2), "n"(4 * pc + 3) : "memory");
        }
    };
    auto readB = [&](auto PC, uint32_t ad, int t, Frag& f) {
        constexpr int pc = decltype(PC)::value;
        if constexpr (B_PERM) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.b16[t]) : "v"(ad), "n"(pc * 1024) : "memory");
        } else {
            asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(f.b[t][0]) : "v"(ad), "n"(4 * pc), "n"(4 * pc + 1) : "memory");
            asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(f.b[t][1]) : "v"(ad), "n"(4 * pc + 2), "n"(4 * pc + 3) : "memory");
        }
    };
    // the registers of `f` are valid after this (placed behind the wait that covers them): the widenings and MFMAs that
    // consume them cannot be scheduled above it
    auto tie = [&](Frag& f) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if constexpr (B_PERM) {
                asm volatile("" : "+v"(f.b16[t]));
            } else {
                asm volatile("" : "+v"(f.b[t][0]));
                asm volatile("" : "+v"(f.b[t][1]));
            }
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if constexpr (KPERM && A8) {
                asm volatile("" : "+v"(f.a4[i]));
            } else if constexpr (KPERM) {
                asm volatile("" : "+v"(f.a16[i]));
            } else if constexpr (A8) {
#pragma unroll
                for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(f.a1[i][s]));
            } else {
                asm volatile("" : "+v"(f.a2[i][0]));
                asm volatile("" : "+v"(f.a2[i][1]));
            }
        }
    };
    auto mma = [&](const Frag& f) {
        float av[MT][4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if constexpr (KPERM && A8) av[i][s] = (float)((f.a4[i] >> (8 * s)) & 0xffu);   // v_cvt_f32_ubyte<s>
                else if constexpr (KPERM) av[i][s] = __uint_as_float(f.a16[i][s]);
                else if constexpr (A8) av[i][s] = (float)f.a1[i][s];
                else av[i][s] = __uint_as_float(f.a2[i][s >> 1][s & 1]);
            }
        if constexpr (A8) {   // the widenings as one burst in front of the MFMAs (no wait states between them)
#pragma unroll
            for (int i = 0; i < MT; ++i) asm volatile("" : "+v"(av[i][0]), "+v"(av[i][1]), "+v"(av[i][2]), "+v"(av[i][3]));
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        av[i][s], __uint_as_float(B_PERM ? f.b16[t][s] : f.b[t][s >> 1][s & 1]), acc[i][t], 0, 0, 0);
    };
    // the NP pieces of quad `qd`, straight-line (no branch inside: a merge point would cost register copies, and
    // copies are VALU work the MFMAs pay for)
    auto run_quad_n = [&](auto NPC, int qd, auto PWC) {
        constexpr int NP = decltype(NPC)::value;
        constexpr bool kPw = decltype(PWC)::value;   // piece p of a 4-piece quad requested piece by piece: (3 - p) (1 + NT) transfers younger than it
        using std::integral_constant;
        uint32_t adA[MT], adB[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) adA[i] = lds_off(imgA) + (uint32_t)(fragA + i * a_tile + qd * 4 * kPieceA);
#pragma unroll
        for (int t = 0; t < NT; ++t) adB[t] = lds_off(imgB) + (uint32_t)(fragB + t * b_tile + qd * 4096);
        auto reads1 = [&](auto PC, Frag& f) {   // chunk 1: at most 15 reads
#pragma unroll
            for (int t = 0; t < NT; ++t) readB(PC, adB[t], t, f);
#pragma unroll
            for (int i = 0; i < MT1; ++i) readA(PC, adA[i], i, f);
        };
        auto reads2 = [&](auto PC, Frag& f) {
#pragma unroll
            for (int i = MT1; i < MT; ++i) readA(PC, adA[i], i, f);
        };
        // one pipeline stage: issue the reads of piece NXT into `fn` (if the quad has it), then the MFMAs of the piece in `fc`
        auto stage = [&](auto NXT, Frag& fn, Frag& fc) {
            constexpr int nxt = decltype(NXT)::value;
            if constexpr (nxt < NP) {
                if constexpr (kPw) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((3 - nxt) * (1 + NT)) : "memory");
                reads1(NXT, fn);
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(kChunk1) : "memory");   // everything older than chunk 1 has landed
                tie(fc);
                reads2(NXT, fn);
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                tie(fc);
            }
            mma(fc);
        };
        Frag f0, f1;
        if constexpr (kPw) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (1 + NT)) : "memory");
        reads1(integral_constant<int, 0>{}, f0);
        reads2(integral_constant<int, 0>{}, f0);
        stage(integral_constant<int, 1>{}, f1, f0);
        if constexpr (NP > 1) stage(integral_constant<int, 2>{}, f0, f1);
        if constexpr (NP > 2) stage(integral_constant<int, 3>{}, f1, f0);
        if constexpr (NP > 3) stage(integral_constant<int, 4>{}, f0, f1);
    };
    auto run_quad = [&](int qd, int np) {   // np = pieces in the quad, 1..4 (wave-uniform)
        using std::integral_constant;
        if (np >= 4) run_quad_n(integral_constant<int, 4>{}, qd, std::false_type{});
        else if (np == 3) run_quad_n(integral_constant<int, 3>{}, qd, std::false_type{});
        else if (np == 2) run_quad_n(integral_constant<int, 2>{}, qd, std::false_type{});
        else if (np == 1) run_quad_n(integral_constant<int, 1>{}, qd, std::false_type{});
    };
    auto run = [&](int p0, int p1) {   // whole quads from p0 (a multiple of 4)
        int pc = p0;
        for (; pc + 4 <= p1; pc += 4) run_quad_n(std::integral_constant<int, 4>{}, pc >> 2, std::false_type{});
        if (pc < p1) run_quad(pc >> 2, p1 - pc);
    };

    if constexpr (FAST_A && FAST_B) {
        // ---- LDS-DMA, one or two quads ahead of the MFMAs: the transfers of the following quads are in flight while quad q
        // is contracted.  The compiler does not order a ds_read behind a pending LDS-DMA, so each quad sits behind an
        // explicit counted vmcnt wait.
        // The first requests of the first panel and the two hooks stand IN FRONT of the panel loop: registers a hook
        // loads into would otherwise be loop-carried, and the compiler would drain every transfer in flight (vmcnt(0))
        // where the loop's paths merge -- in front of the first MFMA.
        const rsrc_t ra = A.rsrc(), rb = Bv.rsrc();
        DmaLane la[MT], lb[NT];
        int pieces = 0, quads = 0;
        auto open_panel = [&](int kb) {   // this lane's walk over panel kb
            const int klen = max(0, min(kend - kb, KP));
            pieces = (klen + 15) >> 4;    // 16 k per piece (klen is a multiple of 16 on this path)
            quads = (pieces + 3) >> 2;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if constexpr (A_TILES) la[i] = A.template lane0_tiles<MT>(m0, kb, lane, i);   // instruction i of a quad
                else la[i] = view_lane0<KPERM>(A, m0 + 16 * i, kb, lane);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) lb[t] = view_lane0<KPERM>(Bv, n0 + 16 * t, kb, lane);
        };
        auto quad_pieces = [&](int qd) { return min(4, pieces - 4 * qd); };
        auto quad_dmas = [&](int qd) { return qd < quads ? (A8 ? MT : MT * quad_pieces(qd)) + nt_dma * quad_pieces(qd) : 0; };
        // PW: a chain that is exactly one full quad, requested piece by piece
        constexpr bool kPwOk = PW && QBM_PIECE_WAIT && A8 && !A_K4 && MT == 4 && PHASE == kWhole;
        auto issue_quad_pw = [&]() {
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) {
                blds16<AUX_A>(ra, la[pc].voff, imgA + pc * 1024);            // transfer pc of the merged image = piece pc of all 4 tiles
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    blds16<AUX_B>(rb, lb[t].voff, imgB + t * b_tile + pc * 1024);
                    lb[t].voff += lb[t].step;
                }
            }
        };
        auto issue_quad = [&](int qd, bool do_a = true, bool do_b = true) {
            const int np = quad_pieces(qd);
            if constexpr (A8) {
                if (do_a)
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    // u8: one instruction covers the 4 pieces (k4 image: of its tile; lin image: a (64/MT)-k slice of
                    // every tile); lanes past the end of the K chain read zeros
                    blds16<AUX_A>(ra, la[i].k < kend ? la[i].voff : kOOB, imgA + (A_TILES ? (qd * MT + i) * 1024 : i * a_tile + qd * 1024));
                    la[i].voff += la[i].step;
                    la[i].k += la[i].kstep;
                }
            }
            for (int pc = 4 * qd; pc < 4 * qd + np; ++pc) {
                if constexpr (!A8) {
                    if (do_a)
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        blds16<AUX_A>(ra, la[i].voff, imgA + i * a_tile + pc * 1024);
                        la[i].voff += la[i].step;
                    }
                }
                if (!b_resident && do_b) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        blds16<AUX_B>(rb, lb[t].voff, imgB + t * b_tile + pc * 1024);
                        lb[t].voff += lb[t].step;
                    }
                }
            }
        };
        // How far the requests run ahead of the quad being contracted: 1 = the next quad is requested when the current one
        // has landed, 2 = two quads in flight.
        open_panel(kq0);
        if constexpr (PHASE != kWhole) {
            // the two-call form: everything of the (single) panel is requested at once, everything is awaited at once
            if constexpr (PHASE == kIssueAB || PHASE == kIssueA || PHASE == kIssueB) {
                for (int qd = 0; qd < quads; ++qd) issue_quad(qd, PHASE != kIssueB, PHASE != kIssueA);
                pre0();
                QBM_STAMP(stamp0 + 1);
                return StripOut<MT, NT, CPT, 64 * NQ>{};
            } else {
                if constexpr (PHASE == kIssueBConsume)
                    for (int qd = 0; qd < quads; ++qd) issue_quad(qd, false, true);
                QBM_STAMP(stamp0 + 1);
                vm_wait(younger);
                QBM_STAMP(stamp0 + 2);
                run(0, pieces);
                QBM_STAMP(stamp0 + 3);
            }
        } else {
        const int ahead = QBM_DMA_AHEAD > 0 ? QBM_DMA_AHEAD : (quads <= 2 ? 1 : 2);
        // STAG > 0 (NQ = 8): the second wavefront of every SIMD requests its operands 64 STAG cycles later -- the first one's
        // land ahead of them and its MFMAs start while the other's are still on their way
        // (stag_on: the caller's run-time say)
        if constexpr (STAG > 0 && NQ == 8)
            if (q >= 4 && stag_on) __builtin_amdgcn_s_sleep(STAG);
        bool pw = false;
        if constexpr (kPwOk) pw = quads == 1 && pieces == 4 && !b_resident && kend - kq0 <= KP && qbm_diag == 0;
        if (pw) issue_quad_pw();
        else if (qbm_diag != 4 && quads > 0) issue_quad(0);
        pre0();
        if (ahead >= 2 && qbm_diag != 4 && quads > 1) issue_quad(1);
        QBM_STAMP(stamp0 + 1);
        pre(ahead >= 2 ? quad_dmas(1) : 0);
        for (int kb = kq0; kb < kend; kb += KP) {
            if (kb != kq0) {
                open_panel(kb);
                if (qbm_diag != 4 && quads > 0) issue_quad(0);
                if (ahead >= 2 && qbm_diag != 4 && quads > 1) issue_quad(1);
            }
            // The wait before quad qd leaves exactly the transfers of the `ahead` quads behind it in flight.  Full quads
            // run in one loop body (one code path: the accumulators stay in place).
            const int full = pieces >> 2;
            int qd = 0;
            if constexpr (kPwOk)
                if (pw) {   // (the hooks' own vector-memory operations -- the W tile of the fused update -- are younger than every
                            //  piece and are awaited by their consumer: the counted waits below leave them in flight... see below)
                    QBM_STAMP(stamp0 + 2);
                    run_quad_n(std::integral_constant<int, 4>{}, 0, std::true_type{});
                    qd = quads;
                }
            for (; qd < full; ++qd) {
                if (qbm_diag != 4) {
                    if (qd + ahead < quads) issue_quad(qd + ahead);
                    vm_wait(qbm_diag == 1 ? 0 : quad_dmas(qd + 1) + (ahead >= 2 ? quad_dmas(qd + 2) : 0));
                }
                if (qd == 0) QBM_STAMP(stamp0 + 2);
                if (qbm_diag != 3) run_quad_n(std::integral_constant<int, 4>{}, qd, std::false_type{});
            }
            if (qd < quads) {   // tail quad: 1..3 pieces, nothing is in flight behind it
                if (qbm_diag != 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (qd == 0) QBM_STAMP(stamp0 + 2);
                if (qbm_diag != 3) run_quad(qd, pieces - 4 * qd);
            }
            QBM_STAMP(stamp0 + 3);
        }
        }   // PHASE == kWhole
    } else {
        // ---- at least one operand goes through registers (unaligned / ragged / synthesised)
        pre0();
        pre(0);
        for (int kb = kq0; kb < kend; kb += KP) {
            const int klen = max(0, min(kend - kb, KP));
            const int pieces = (klen + 15) >> 4;
            const int quads = (pieces + 3) >> 2;
            if constexpr (FAST_A) {
                const rsrc_t ra = A.rsrc();
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    DmaLane la;
                    if constexpr (A_TILES) la = A.template lane0_tiles<MT>(m0, kb, lane, i);
                    else la = view_lane0<KPERM>(A, m0 + 16 * i, kb, lane);
                    for (int g = 0; g < (A8 ? quads : pieces); ++g) {
                        blds16<AUX_A>(ra, (!A8 || la.k < kend) ? la.voff : kOOB, imgA + (A_TILES ? (g * MT + i) * 1024 : i * a_tile + g * 1024));
                        la.voff += la.step;
                        la.k += la.kstep;
                    }
                }
            } else if constexpr (A8) {
                for (int g = 0; g < quads; ++g)
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const uint4 v = view_get<KPERM>(A, m0 + 16 * i, kb, kend, g * 64 + lane);
                        if constexpr (A_TILES) reinterpret_cast<uint4*>(imgA)[(g * 64 + lane) * MT + i] = v;   // slot k*MT + tile
                        else reinterpret_cast<uint4*>(imgA + i * a_tile)[g * 64 + lane] = v;
                    }
            } else {
                for (int pc = 0; pc < pieces; ++pc)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
                        reinterpret_cast<float4*>(imgA + i * a_tile)[pc * 64 + lane] = view_get<KPERM>(A, m0 + 16 * i, kb, kend, pc * 64 + lane);
            }
            if constexpr (FAST_B) {
                const rsrc_t rb = Bv.rsrc();
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    DmaLane lb = view_lane0<KPERM>(Bv, n0 + 16 * t, kb, lane);
                    for (int pc = 0; pc < pieces; ++pc) {
                        blds16<AUX_B>(rb, lb.voff, imgB + t * b_tile + pc * 1024);
                        lb.voff += lb.step;
                    }
                }
            } else {
                for (int pc = 0; pc < pieces; ++pc)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        reinterpret_cast<float4*>(imgB + t * b_tile)[pc * 64 + lane] = view_get<KPERM>(Bv, n0 + 16 * t, kb, kend, pc * 64 + lane);
            }
            QBM_STAMP(stamp0 + 1);
            if constexpr (FAST_A || FAST_B) vm_wait(0);
            QBM_STAMP(stamp0 + 2);
            run(0, pieces);
            QBM_STAMP(stamp0 + 3);
        }
    }

    post();
    // partial strips -> LDS (each wave reuses the head of its own staging region), row-major
    // [16*MT][RS], RS = 16*NT + 4 (the +4 keeps lanes l and l+16, four rows apart, on different banks)
    constexpr int RS = 16 * NT + 4;
    float* red = reinterpret_cast<float*>(img);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(16 * i + (lane >> 4) * 4 + r) * RS + 16 * t + (lane & 15)] = acc[i][t][r];
    // A `post` hook may have put transfers in flight that are to land under what follows: the workgroup barriers of the
    // combine then order LDS accesses only (__syncthreads makes the compiler drain vmcnt as well)
    constexpr bool kKeepVm = !std::is_same<std::decay_t<Post>, NoPre>::value;
    auto wg_barrier = [&]() {
        if constexpr (kKeepVm) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else __syncthreads();
    };
    wg_barrier();

    StripOut<MT, NT, CPT, 64 * NQ> o;
#pragma unroll
    for (int p = 0; p < StripOut<MT, NT, CPT, 64 * NQ>::NP; ++p) {
        o.sum[p] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (o.valid(p)) {
            const char* base = lds + (size_t)(o.row(p) * RS + CPT * o.cg(p)) * 4;
            float part[NQ][CPT];   // all loads first, then the ordered sums (..((q0 + q1) + q2) + ..) + q_{NQ-1}
#pragma unroll
            for (int qq = 0; qq < NQ; ++qq) {
                if constexpr (CPT == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(base + (size_t)qq * wave_bytes);
                    part[qq][0] = v.x; part[qq][1] = v.y; part[qq][2] = v.z; part[qq][3] = v.w;
                } else if constexpr (CPT == 2) {
                    const float2 v = *reinterpret_cast<const float2*>(base + (size_t)qq * wave_bytes);
                    part[qq][0] = v.x; part[qq][1] = v.y;
                } else {
                    part[qq][0] = *reinterpret_cast<const float*>(base + (size_t)qq * wave_bytes);
                }
            }
            float acc[CPT];
#pragma unroll
            for (int e = 0; e < CPT; ++e) {
                acc[e] = part[0][e];
#pragma unroll
                for (int qq = 1; qq < NQ; ++qq) acc[e] = acc[e] + part[qq][e];
            }
            o.sum[p].x = acc[0];
            if constexpr (CPT >= 2) o.sum[p].y = acc[1];
            if constexpr (CPT == 4) { o.sum[p].z = acc[2]; o.sum[p].w = acc[3]; }
        }
    }
    QBM_STAMP(stamp0 + 4);
    wg_barrier();  // the partials are consumed: the caller may run another contraction on this LDS
    return o;
}

// Speed-only remap of blockIdx -> (m strip, n strip): workgroups that read the same B panel
// (same n strip) are dealt to the same XCD (blocks b and b+8 share one), so that each XCD's L2
// holds 1/8 of the B operand.  Falls back to the plain map when nN is not a multiple of 8.
// The same for any nN, on a grid padded to 8 * nM * ceil(nN / 8) strips workgroups (strip_grid_padded): XCD x takes the n
// strips x, x + 8, ... with all their m strips; false = a workgroup of the padding (nothing to do).
__host__ __device__ __forceinline__ int strip_grid_padded(int nM, int nN) { return 8 * nM * ((nN + 7) / 8); }
// j / d without the ~60-instruction integer division (a v_rcp round trip in front of a kernel's first operand request, on
// every wavefront): inv = div_magic(d) = ceil(2^32 / d) from the host, exact for j < 2^32 / d; inv == 0 (d == 1, or a caller
// that passes none): the plain division
inline unsigned div_magic(int d) { return d >= 2 ? (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d) : 0u; }
__device__ __forceinline__ int div_by(int j, int d, unsigned inv) { return inv ? (int)__umulhi((unsigned)j, inv) : j / d; }
__device__ __forceinline__ bool strip_coords_padded(int nM, int nN, int& ms, int& ns, int first = 0, unsigned inv_nM = 0u)
{
    const int id = (int)blockIdx.x - first;
    const int x = id & 7, j = id >> 3, q = div_by(j, nM, inv_nM);
    ns = x + 8 * q;
    ms = j - q * nM;
    return ns < nN;
}

// ABS: n strip ns goes to XCD ns % 8 whatever `first` is (a rotation inside every group of 8 workgroups): a launch whose n
// strips cover what the n strips of the launch before it wrote finds that in its XCD's L2 (k_grad behind k_hidden)
// pair (nN a multiple of 16): n strips 2j, 2j + 1 on XCD j % 8 instead (k_hidden's map when H is a multiple of 256)
template <bool ABS = false>
__device__ __forceinline__ void strip_coords(int nM, int nN, int& ms, int& ns, int first = 0, unsigned inv_nM = 0u, bool pair = false)
{
    const int id = (int)blockIdx.x - first;  // `first` aux workgroups precede the strips in the grid
    if ((nN & 7) == 0) {
        const int x = (ABS ? (int)blockIdx.x : id) & 7, j = id >> 3, q = div_by(j, nM, inv_nM);
        ns = pair ? 16 * (q >> 1) + 2 * x + (q & 1) : x + 8 * q;
        ms = j - q * nM;
    } else {
        ms = id / nN;
        ns = id % nN;
    }
}

}  // namespace qbm
