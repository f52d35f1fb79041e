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

}  // extern "C"
