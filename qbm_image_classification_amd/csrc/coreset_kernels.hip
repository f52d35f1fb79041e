// coreset_kernels.hip -- greedy k-center coreset selection (src/coreset.py:28-106).
// One launch per pick: every workgroup first reduces the previous launch's per-workgroup candidates to the new
// centre p (largest running minimum distance, lowest index on ties = torch.argmax), then updates its slice of the
// running minima with the distances to p and leaves its own candidate for the next launch.
//
// Numerical contract (restated in oracle/qbm_oracle.c: orc_kcenter_greedy), fp32:
//   sq[j]   = chain s = fmaf(x_jk, x_jk, s), k ascending            (euclidean_dist :8-9  pow(x,2).sum(1))
//   dot     = chain t = fmaf(x_pk, x_jk, t), k ascending            (:11 addmm_)
//   d(p,j)  = sqrt(max((sq[p] + sq[j]) - 2*dot, 1e-12))             (:10-12), sqrt correctly rounded
// The embedding matrix is read transposed ([D,N]: lane j streams column j, coalesced).
#include "qbm_device.hip.h"
#include "qbm_host.hpp"
#include "../../include/qbm_hip.h"

using namespace qbm;

namespace {

constexpr int KC_THREADS = 64;
constexpr int KC_MAX_BLOCKS = 1024;

struct Cand {
    float v;
    int idx;
};
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

__device__ __forceinline__ Cand block_argmax(float v, int idx, Cand* sh)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(v, off, kWave);
        const int oi = __shfl_xor(idx, off, kWave);
        if (better(ov, oi, v, idx)) { v = ov; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = Cand{v, idx};
    __syncthreads();
    Cand c = sh[0];
    for (int w = 1; w < KC_THREADS / 64; ++w)
        if (better(sh[w].v, sh[w].idx, c.v, c.idx)) c = sh[w];
    __syncthreads();
    return c;
}

__global__ __launch_bounds__(KC_THREADS) void k_kc_init(const float* __restrict__ embT, int N, int D, float* __restrict__ sq,
                                                        float* __restrict__ mins, uint8_t* __restrict__ selected,
                                                        const int32_t* __restrict__ seeds, int n_seeds)
{
    const int j = blockIdx.x * KC_THREADS + threadIdx.x;
    if (j >= N) return;
    float s = 0.f;
    for (int k = 0; k < D; ++k) {
        const float x = embT[(long)k * N + j];
        s = __builtin_fmaf(x, x, s);
    }
    sq[j] = s;
    mins[j] = __builtin_inff();
    bool sel = false;
    for (int t = 0; t < n_seeds; ++t) sel = sel || seeds[t] == j;
    selected[j] = sel ? 1 : 0;
}

// MODE 0: seed pass (centre given; selected points keep +inf).  MODE 1: pick pass (centre = argmax of the candidates;
// selected points, the new centre included, drop to -1 as in the reference: mins[p] = -1, row[selected] = -1).
template <int MODE>
__global__ __launch_bounds__(KC_THREADS) void k_kc_step(const float* __restrict__ embT, const float* __restrict__ sq, int N,
                                                        int D, float* __restrict__ mins, uint8_t* __restrict__ selected,
                                                        const Cand* __restrict__ cand_in, int n_cand_in,
                                                        Cand* __restrict__ cand_out, int32_t* __restrict__ picks, int pick_no,
                                                        int last, int seed_p)
{
    __shared__ Cand sh[KC_THREADS / 64];
    int p;
    if (MODE == 0) {
        p = seed_p;
    } else {
        float bv = -__builtin_inff();
        int bi = 0x7fffffff;
        for (int t = threadIdx.x; t < n_cand_in; t += KC_THREADS)
            if (better(cand_in[t].v, cand_in[t].idx, bv, bi)) { bv = cand_in[t].v; bi = cand_in[t].idx; }
        p = block_argmax(bv, bi, sh).idx;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            picks[pick_no] = p;
            selected[p] = 1;
        }
        if (last) return;   // src/coreset.py:95-96: the last pick is not followed by an update
    }
    // the centre's coordinates go to LDS once per workgroup; the column of point j is streamed 8 loads at a time
    extern __shared__ float xp[];
    for (int k = threadIdx.x; k < D; k += KC_THREADS) xp[k] = embT[(long)k * N + p];
    __syncthreads();
    const float sqp = sq[p];
    float bv = -__builtin_inff();
    int bi = 0x7fffffff;
    for (int j = blockIdx.x * KC_THREADS + threadIdx.x; j < N; j += gridDim.x * KC_THREADS) {
        const bool sel = selected[j] != 0 || (MODE == 1 && j == p);
        float dot = 0.f;
        int k = 0;
        for (; k + 8 <= D; k += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = embT[(long)(k + u) * N + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) dot = __builtin_fmaf(xp[k + u], v[u], dot);
        }
        for (; k < D; ++k) dot = __builtin_fmaf(xp[k], embT[(long)k * N + j], dot);
        const float d2 = (sqp + sq[j]) - 2.0f * dot;
        const float d = __fsqrt_rn(__builtin_fmaxf(d2, 1e-12f));
        float m = mins[j];
        if (MODE == 0) m = sel ? m : __builtin_fminf(m, d);
        else m = __builtin_fminf(m, sel ? -1.0f : d);
        mins[j] = m;
        if (better(m, j, bv, bi)) { bv = m; bi = j; }
    }
    const Cand c = block_argmax(bv, bi, sh);
    if (threadIdx.x == 0) cand_out[blockIdx.x] = c;
}


// ---------------------------------------------------------------------------------------------
// All picks in ONE launch (round 4).  A pick needs every point's distance to the centre chosen by the pick before it: one
// chip-wide exchange per pick whatever the organisation.  With a launch per pick that exchange is a kernel boundary
// (~1.5 us) in front of a kernel that re-reads the whole embedding matrix (13 MB at 12 665 x 256) with one wavefront per
// workgroup: 10.4 - 11.7 us per pick.  Here 256 persistent workgroups -- one per CU -- keep their slice of the points
// (ceil(N / 256) rows, pitch D + 4 floats: conflict-free ds_read_b128) in LDS for the whole run, with the running minima
// in registers, and exchange per pick ONE 8-byte granule {value, index, round} each way:
//   * inside an XCD: every workgroup stores the granule of its best point with a plain 8-byte store (it lands in the
//     XCD's L2); the XCD's leader (the first workgroup to arrive there) polls the XCD's granules with L1-bypassing loads
//     until all carry the round: the barrier IS the data (0.3 us, tools/microbench/xcd_barrier.hip);
//   * across XCDs: the leader publishes the XCD's best as one `sc1` (write-through) 8-byte store on a line of its own and
//     polls the 8 lines with `sc1` loads (a data-tagged granule: no fence, no ordering needed; MI355X_MICROARCH.md
//     "handoff-1to1": ~1 us); the other workgroups of the chip poll the same 8 lines themselves (the leader handing the
//     winner back through its XCD's L2 -- one more hop for 8 pollers instead of 256 -- was 0.35 us per pick slower);
//   * the new centre's coordinates are read from the row-major matrix (1 KB, contiguous).
// Measured at 12 665 x 256, 1000 picks: 5.5 us per pick with every workgroup polling both levels (the first version);
// 7.0 us with a FLAT exchange (every workgroup publishes write-through and polls all 256 granules: one hop instead of two,
// and 256 pollers on 16 lines -- slower); profiles/r04_other_configs.jsonl has this version's number.  Two dependent
// cross-XCD round trips per pick (the candidates, then the winner's coordinates) at ~1 us each are the floor of the scheme.
// Same contract as the launch-per-pick kernels above (fmaf chains in ascending k, first index on ties): the same picks.
// Placement is not assumed: a workgroup reads HW_REG_XCC_ID, takes a ticket on its XCD, and a start-up rendezvous tells
// everybody how many workgroups every XCD got.  Every spin is bounded: a launch whose workgroups are not all resident (or an
// XCD with more than 64 of them) sets the status word and ends; the caller then runs the launch-per-pick path.
// ---------------------------------------------------------------------------------------------
constexpr int KF_WGS = 256, KF_THREADS = 256, KF_SPIN = 1 << 20;
struct KfCtl {                         // zeroed by the host before the launch
    uint32_t ticket[8];                // workgroups that have arrived per XCD (= rank of the next one)
    uint32_t arrived;                  // start-up rendezvous
    uint32_t status;                   // word 9: 0 fine; 1 a spin ran out; 2 more than 64 workgroups on one XCD
    uint32_t pad[22];
    // All granule arrays are double-buffered by the parity of the round: a workgroup that has finished round r may publish
    // round r + 1 while a slower one is still polling for round r (nobody can be TWO rounds ahead: round r + 1 completes only
    // with everybody's round-(r + 1) granule, which a workgroup stores after it has finished round r).
    unsigned long long xc[2][8][64];   // per XCD: one granule per workgroup (plain stores, sc1 polls: XCD-local)
    unsigned long long gc[2][8][16];   // per XCD: its best, word 0 of a 128-byte line of its own (sc1 store, sc1 polls)
    unsigned long long win[2][8][16];  // (unused since every workgroup polls gc itself; kept: the workspace size is part of the ABI)
};
__device__ __forceinline__ unsigned long long kf_pack(float v, int idx, uint32_t round)
{
    return ((unsigned long long)__float_as_uint(v) << 32) | ((unsigned long long)((uint32_t)idx & 0xffffffu) << 8) | (round & 0xffu);
}
__device__ __forceinline__ void kf_unpack(unsigned long long g, float& v, int& idx)
{
    v = __uint_as_float((uint32_t)(g >> 32));
    idx = (int)((g >> 8) & 0xffffffu);
}
__device__ __forceinline__ uint32_t kf_xcc()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

__global__ __launch_bounds__(KF_THREADS) void k_kc_fused(const float* __restrict__ emb /* [N,D] row-major */, int N, int D, int P /* points per workgroup */,
                                                         const int32_t* __restrict__ seeds, int n_seeds, int n_picks,
                                                         int32_t* __restrict__ picks, uint8_t* __restrict__ selected, KfCtl* __restrict__ ctl)
{
    extern __shared__ __attribute__((aligned(16))) float kf_lds[];
    __shared__ uint32_t s_x, s_rank, s_ok, s_cnt[8];
    __shared__ Cand s_c[KF_THREADS / 64];
    __shared__ int s_p;
    __shared__ float s_sqp;
    const int tid = threadIdx.x, lane = tid & 63;
    const int pitch = D + 4;
    float* xp = kf_lds;                    // [D] the current centre
    float* sl = kf_lds + ((D + 3) & ~3);   // [P][pitch] this workgroup's points

    // ---- who am I, and who else is there
    if (tid == 0) {
        const uint32_t x = kf_xcc();
        s_x = x;
        s_rank = __hip_atomic_fetch_add(&ctl->ticket[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&ctl->arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t ok = 1u;
        for (int spins = 0; __hip_atomic_load(&ctl->arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)KF_WGS; ++spins)
            if (spins > KF_SPIN) { ok = 0u; break; }
        if (ok)
            for (int q = 0; q < 8; ++q) {
                s_cnt[q] = __hip_atomic_load(&ctl->ticket[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (s_cnt[q] > 64u) ok = 2u;
            }
        if (ok != 1u) __hip_atomic_store(&ctl->status, ok == 0u ? 1u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ok = ok;
    }
    __syncthreads();
    if (s_ok != 1u) return;
    const int x = (int)s_x, rank = (int)s_rank;
    int g = rank;                          // global index of this workgroup: XCDs in order, ranks inside
    for (int q = 0; q < x; ++q) g += (int)s_cnt[q];
    const int j0 = g * P;                  // first point of this workgroup
    const int np = max(0, min(P, N - j0));

    // ---- the slice into LDS (rows are contiguous in memory: coalesced), squared norms, state
    for (int e = tid; e < np * D; e += KF_THREADS) {
        const int r = e / D, k = e - r * D;
        sl[r * pitch + k] = emb[(long)(j0 + r) * D + k];
    }
    __syncthreads();
    const bool mine = tid < np;
    const int j = j0 + tid;
    const float* row = sl + tid * pitch;
    float sqj = 0.f, m = __builtin_inff(), dot_keep = 0.f;
    bool sel = false;
    if (mine) {
        for (int k = 0; k < D; ++k) sqj = __builtin_fmaf(row[k], row[k], sqj);
        for (int t = 0; t < n_seeds; ++t) sel = sel || seeds[t] == j;
        selected[j] = sel ? 1 : 0;
    }
    // distances of the own points to centre p (whose coordinates are read here), running minima updated.
    // A point's distance is ONE thread's chain of D dependent fmaf (the contract): what matters is that nothing else sits on
    // that path -- the LDS reads of 32 k are requested together in front of their 32 fmaf (one read per 4 k in a rolled loop
    // paid an LDS round trip per iteration: 4 of the first version's 5.5 us per pick), and the centre's squared norm -- the same
    // chain in every thread -- is walked ONCE, by the second wavefront, beside the distance chains of the first.
    auto visit = [&](int p, bool seed_pass) {
        __syncthreads();                  // (xp, s_sqp may still be read by a slower wavefront of the visit before)
        for (int k = tid; k < D; k += KF_THREADS) xp[k] = emb[(long)p * D + k];
        __syncthreads();
        const bool do_sq = (tid >> 6) == 1 || KF_THREADS == 64;
        if (mine || do_sq) {
            const float* rw = mine ? row : sl;    // (a thread that only walks the norm reads some valid row and drops the result)
            float sqp = 0.f, dot = 0.f;
            int k = 0;
            for (; k + 32 <= D; k += 32) {
                float4 a[8], b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a[u] = *reinterpret_cast<const float4*>(xp + k + 4 * u);      // broadcast reads
                    b[u] = *reinterpret_cast<const float4*>(rw + k + 4 * u);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    dot = __builtin_fmaf(a[u].x, b[u].x, dot); dot = __builtin_fmaf(a[u].y, b[u].y, dot);
                    dot = __builtin_fmaf(a[u].z, b[u].z, dot); dot = __builtin_fmaf(a[u].w, b[u].w, dot);
                }
                if (do_sq) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        sqp = __builtin_fmaf(a[u].x, a[u].x, sqp); sqp = __builtin_fmaf(a[u].y, a[u].y, sqp);
                        sqp = __builtin_fmaf(a[u].z, a[u].z, sqp); sqp = __builtin_fmaf(a[u].w, a[u].w, sqp);
                    }
                }
            }
            for (; k < D; ++k) {
                dot = __builtin_fmaf(xp[k], rw[k], dot);
                if (do_sq) sqp = __builtin_fmaf(xp[k], xp[k], sqp);
            }
            if (tid == 64 || (KF_THREADS == 64 && tid == 0)) s_sqp = sqp;
            dot_keep = dot;
        }
        __syncthreads();
        if (!mine) return;
        const float d2 = (s_sqp + sqj) - 2.0f * dot_keep;
        const float d = __fsqrt_rn(__builtin_fmaxf(d2, 1e-12f));
        if (seed_pass) m = sel ? m : __builtin_fminf(m, d);
        else { sel = sel || j == p; m = __builtin_fminf(m, sel ? -1.0f : d); }
    };
    for (int t = 0; t < n_seeds; ++t) visit(seeds[t], true);          // src/coreset.py:81-89

    // ---- the picks (:91-103)
    for (int i = 0; i < n_picks; ++i) {
        const uint32_t round = (uint32_t)(i + 1), tag = round & 0xffu;
        const int par = (int)(round & 1u);
        // best of this workgroup
        float bv = mine ? m : -__builtin_inff();
        int bi = mine ? j : 0xffffff;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(bv, off, kWave);
            const int oi = __shfl_xor(bi, off, kWave);
            if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) s_c[tid >> 6] = Cand{bv, bi};
        __syncthreads();
        if (tid < 64) {
            Cand c = s_c[0];
            for (int w = 1; w < KF_THREADS / 64; ++w)
                if (better(s_c[w].v, s_c[w].idx, c.v, c.idx)) c = s_c[w];
            // my granule to the XCD (its L2)
            if (lane == 0) __hip_atomic_store(&ctl->xc[par][x][rank], kf_pack(c.v, c.idx, round), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            unsigned long long gr = 0ull;
            bool ok = true;
            float v = -__builtin_inff();
            int idx = 0xffffff;
            if (rank == 0) {
                // the XCD's leader: everybody's granule in (the barrier is the data), the XCD's best out across the chip,
                // the 8 XCDs' in, the winner to the XCD
                const int cnt = (int)s_cnt[x];
                for (int spins = 0;; ++spins) {
                    gr = lane < cnt ? __hip_atomic_load(&ctl->xc[par][x][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned long long)tag;
                    if (__all((int)((gr & 0xffull) == tag))) break;
                    if (spins > KF_SPIN) { ok = false; break; }
                }
                if (lane < cnt) kf_unpack(gr, v, idx);
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const float ov = __shfl_xor(v, off, kWave);
                    const int oi = __shfl_xor(idx, off, kWave);
                    if (better(ov, oi, v, idx)) { v = ov; idx = oi; }
                }
                if (ok && lane == 0) __hip_atomic_store(&ctl->gc[par][x][0], kf_pack(v, idx, round), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool live = lane < 8 && s_cnt[lane & 7] > 0u;
                for (int spins = 0; ok; ++spins) {
                    gr = live ? __hip_atomic_load(&ctl->gc[par][lane & 7][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned long long)tag;
                    if (__all((int)((gr & 0xffull) == tag))) break;
                    if (spins > KF_SPIN) ok = false;
                }
                v = -__builtin_inff();
                idx = 0xffffff;
                if (live) kf_unpack(gr, v, idx);
#pragma unroll
                for (int off = 4; off >= 1; off >>= 1) {
                    const float ov = __shfl_xor(v, off, kWave);
                    const int oi = __shfl_xor(idx, off, kWave);
                    if (better(ov, oi, v, idx)) { v = ov; idx = oi; }
                }
            }
            else {
                // every other workgroup polls the 8 XCDs' granules ITSELF: the leader handing the winner back through the XCD's L2 (one
                // more hop, 8 pollers instead of 256 on the 8 lines) was 0.35 us per pick slower (4.85 against 4.5, alternating runs)
                const bool live = lane < 8 && s_cnt[lane & 7] > 0u;
                for (int spins = 0; ok; ++spins) {
                    gr = live ? __hip_atomic_load(&ctl->gc[par][lane & 7][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned long long)tag;
                    if (__all((int)((gr & 0xffull) == tag))) break;
                    if ((spins & 255) == 255 && __hip_atomic_load(&ctl->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = false; break; }
                    if (spins > KF_SPIN) ok = false;
                }
                v = -__builtin_inff();
                idx = 0xffffff;
                if (live) kf_unpack(gr, v, idx);
#pragma unroll
                for (int off = 4; off >= 1; off >>= 1) {
                    const float ov = __shfl_xor(v, off, kWave);
                    const int oi = __shfl_xor(idx, off, kWave);
                    if (better(ov, oi, v, idx)) { v = ov; idx = oi; }
                }
            }
            if (lane == 0) {
                if (!ok) __hip_atomic_store(&ctl->status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_ok = ok ? 1u : 0u;
                s_p = idx;
            }
        }
        __syncthreads();
        if (s_ok != 1u) return;
        const int p = s_p;
        if (g == 0 && tid == 0) picks[i] = p;
        if (mine && j == p) selected[j] = 1;
        if (i == n_picks - 1) break;      // src/coreset.py:95-96: the last pick is not followed by an update
        visit(p, false);
    }
}

}  // namespace

extern "C" {

long qbm_kcenter_workspace_bytes(int N, int n_seeds)
{
    if (N <= 0 || n_seeds < 0) return 0;
    return (long)N * 8 + 2L * KC_MAX_BLOCKS * (long)sizeof(Cand) + (long)n_seeds * 4 + 64;
}

int qbm_kcenter_greedy(const float* embT, int N, int D, const int32_t* seeds_host, int n_seeds, int n_picks,
                       int32_t* picks, uint8_t* selected, void* workspace, void* stream)
{
    if (N <= 0 || D <= 0 || D > 16384) return fail(QBM_ERR_ARG, "kcenter: bad shape (1 <= D <= 16384)");
    if (n_seeds <= 0 || n_picks < 0) return fail(QBM_ERR_ARG, "kcenter: needs at least one seed centre and n_picks >= 0");
    if (!embT || !seeds_host || !selected || !workspace || (n_picks > 0 && !picks))
        return fail(QBM_ERR_ARG, "kcenter: null buffer");
    for (int t = 0; t < n_seeds; ++t)
        if (seeds_host[t] < 0 || seeds_host[t] >= N) return fail(QBM_ERR_ARG, "kcenter: seed centre out of range");
    hipStream_t st = (hipStream_t)stream;
    float* sq = reinterpret_cast<float*>(workspace);
    float* mins = sq + N;
    Cand* cand[2] = {reinterpret_cast<Cand*>(mins + N), reinterpret_cast<Cand*>(mins + N) + KC_MAX_BLOCKS};
    int32_t* seeds_dev = reinterpret_cast<int32_t*>(cand[1] + KC_MAX_BLOCKS);
    if (hipError_t e = hipMemcpyAsync(seeds_dev, seeds_host, (size_t)n_seeds * 4, hipMemcpyHostToDevice, st); e != hipSuccess)
        return hip_fail(e, "kcenter: copying the seed centres");
    const int nb_all = (N + KC_THREADS - 1) / KC_THREADS;
    const int nb = nb_all < KC_MAX_BLOCKS ? nb_all : KC_MAX_BLOCKS;
    hipLaunchKernelGGL(k_kc_init, dim3(nb_all), dim3(KC_THREADS), 0, st, embT, N, D, sq, mins, selected, seeds_dev, n_seeds);
    QBM_CHECK_LAUNCH("k_kc_init");
    int cur = 0;
    for (int t = 0; t < n_seeds; ++t) {   // distances from the given centres (src/coreset.py:81-89)
        hipLaunchKernelGGL((k_kc_step<0>), dim3(nb), dim3(KC_THREADS), (size_t)D * 4, st, embT, sq, N, D, mins, selected,
                           (const Cand*)nullptr, 0, cand[cur], (int32_t*)nullptr, 0, 0, (int)seeds_host[t]);
        QBM_CHECK_LAUNCH("k_kc_step<seed>");
        // every seed pass leaves candidates of the full current minima; only the last one is consumed
    }
    for (int i = 0; i < n_picks; ++i) {   // :91-103
        hipLaunchKernelGGL((k_kc_step<1>), dim3(nb), dim3(KC_THREADS), (size_t)D * 4, st, embT, sq, N, D, mins, selected,
                           (const Cand*)cand[cur], nb, cand[cur ^ 1], picks, i, i == n_picks - 1 ? 1 : 0, 0);
        QBM_CHECK_LAUNCH("k_kc_step<pick>");
        cur ^= 1;
    }
    return QBM_OK;
}

long qbm_kcenter_fused_workspace_bytes(void) { return (long)sizeof(KfCtl); }

int qbm_kcenter_greedy_fused(const float* emb, int N, int D, const int32_t* seeds_dev, int n_seeds, int n_picks, int32_t* picks,
                             uint8_t* selected, void* workspace, void* stream)
{
    if (N <= 0 || D <= 0) return fail(QBM_ERR_ARG, "kcenter_fused: bad shape");
    if (n_seeds <= 0 || n_picks < 0) return fail(QBM_ERR_ARG, "kcenter_fused: needs at least one seed centre and n_picks >= 0");
    if (!emb || !seeds_dev || !selected || !workspace || (n_picks > 0 && !picks)) return fail(QBM_ERR_ARG, "kcenter_fused: null buffer");
    if (N >= (1 << 24)) return fail(QBM_ERR_ARG, "kcenter_fused: at most 2^24 - 1 points");
    if (D % 4 != 0 || !aligned16(emb)) return fail(QBM_ERR_ARG, "kcenter_fused: D must be a multiple of 4 and emb 16-byte aligned");
    const int P = (N + KF_WGS - 1) / KF_WGS;
    if (P > KF_THREADS) return fail(QBM_ERR_ARG, "kcenter_fused: more than 256 points per workgroup (N > 65536): use qbm_kcenter_greedy");
    const size_t lds = ((size_t)((D + 3) & ~3) + (size_t)P * (D + 4)) * sizeof(float);
    if (lds > 160 * 1024 - 2048) return fail(QBM_ERR_ARG, "kcenter_fused: a workgroup's slice does not fit the LDS: use qbm_kcenter_greedy");
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < KF_WGS)
        return fail(QBM_ERR_ARG, "kcenter_fused: needs 256 compute units (one persistent workgroup each)");
    hipStream_t st = (hipStream_t)stream;
    if (hipError_t e = hipMemsetAsync(workspace, 0, sizeof(KfCtl), st); e != hipSuccess) return hip_fail(e, "kcenter_fused: zeroing the control block");
    if (lds > 65536) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_kc_fused), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_kc_fused)");
    }
    hipLaunchKernelGGL(k_kc_fused, dim3(KF_WGS), dim3(KF_THREADS), lds, st, emb, N, D, P, seeds_dev, n_seeds, n_picks, picks, selected,
                       reinterpret_cast<KfCtl*>(workspace));
    QBM_CHECK_LAUNCH("k_kc_fused");
    return QBM_OK;
}

}  // extern "C"
