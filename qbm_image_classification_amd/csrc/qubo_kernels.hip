// qubo_kernels.hip -- HIP replacement of the QUBO annealer behind Disc_QBM.get_samples
// (src/model/faster_dqbm.py:571-684 -> sample_sa :299-306 -> third-party dwave-neal) and of the
// sample_Q protocol (src/qubo/sampler.py:26-33): batched heat-bath (Gibbs) sweeps on
//     E(s) = sum_i lin_i s_i + sum_{i<j} J_ij s_i s_j,   P(s) ~ exp(-beta E).
// The quadratic part J of Disc_QBM's QUBO is image-independent (faster_dqbm.py:242-243, 256-271);
// only the linear field depends on the image (:239, :279), so one J[n,n] serves a whole minibatch.
#include "qbm_device.hip.h"
#include "qbm_host.hpp"
#include "../../include/qbm_hip.h"

using namespace qbm;

namespace {

// ---------------------------------------------------------------------------------------------
// One wavefront per chain (image b, read r).  Lane l keeps the local field F_i, the threshold and the
// state of spins i = 64 t + l (t < NF) in registers.
//
// The sweep is sequential in i by contract, but it is evaluated speculatively 64 spins at a time: all
// lanes of a tile test their spin against the CURRENT fields at once (one multiply + compare), a ballot
// finds the first lane that wants to flip; every lane before it keeps its state -- exactly what the
// sequential sweep would have decided, because no earlier flip has touched their fields -- the flip is
// applied (every lane adds +-J[i, own spins]: one coalesced row read, wave-uniform control flow), and
// the test is repeated for the lanes behind it.  Cost per tile = (number of flips + 1) wave-wide steps
// instead of 64 serial ones; results are bit-identical to the serial order.
// Uniforms come from Philox (row = chain id, col = spin), one draw id per sweep.
// ---------------------------------------------------------------------------------------------
// J row readers.  The couplings are symmetric (J = Q_offdiag + Q_offdiag^T, include/qbm_hip.h), so a workgroup can
// keep the upper triangle (diagonal included) in LDS: n(n+1)/2 floats, 132.6 KB at n = 257.  Every flip reads one row
// of J; from L2 that read (1 KB, on the critical path of the chain) is what bounds the sweep -- 12 G spin updates/s
// hot vs 156 G cold -- from LDS it costs a few bank-conflicted cycles.
//   T(r, c) = tri[S(r) + c - r],  r <= c,  S(r) = r n - r (r - 1) / 2.
// row<T>(p, ...): the row of spin p, which lies in tile T (compile-time in the unrolled sweep).  Lanes beyond n
// ("phantom" spins of the last tile) get some finite value: their fields are never tested, flipped or stored.
struct JGlobal {
    const float* J;
    int n;
    __device__ __forceinline__ void setup(int) {}
    template <int T, int NF>
    __device__ __forceinline__ void row(int p, int lane, float (&out)[NF]) const
    {
        const float* Jr = J + (long)p * n;
#pragma unroll
        for (int tt = 0; tt < NF; ++tt) out[tt] = Jr[min(64 * tt + lane, n - 1)];
    }
};

template <int NFT>
struct JTri {
    const float* tri;   // LDS
    int n;
    int ci[NFT];        // S(i) - i of this lane's spins: T(i, p) = tri[ci + p] for i < p
    int ic[NFT];        // this lane's spin indices, clamped to n - 1
    __device__ __forceinline__ void setup(int lane)
    {
#pragma unroll
        for (int tt = 0; tt < NFT; ++tt) {
            const int i = min(64 * tt + lane, n - 1);
            ic[tt] = i;
            ci[tt] = i * n - (i * (i - 1)) / 2 - i;
        }
    }
    template <int T, int NF>
    __device__ __forceinline__ void row(int p, int lane, float (&out)[NF]) const
    {
        const int up = p * n - (p * (p - 1)) / 2 - p;   // wave-uniform: T(p, i) = tri[up + i] for i >= p
#pragma unroll
        for (int tt = 0; tt < NF; ++tt) {
            int ad;
            if (tt < T) ad = ci[tt] + p;                 // every spin of an earlier tile is below p
            else if (tt > T) ad = up + ic[tt];           // every spin of a later tile is above p
            else ad = (ic[tt] >= p) ? up + ic[tt] : ci[tt] + p;
            out[tt] = tri[ad];
        }
    }
};

// jr.row<T>() with the (unrolled, hence constant-folded) tile index t of the flipping spin
template <int NF, class JR>
__device__ __forceinline__ void row_of_tile(const JR& jr, int t, int p, int lane, float (&out)[NF])
{
    if constexpr (NF <= 8) {
        switch (t) {
        case 0: jr.template row<0>(p, lane, out); break;
        case 1: if constexpr (NF > 1) jr.template row<1>(p, lane, out); break;
        case 2: if constexpr (NF > 2) jr.template row<2>(p, lane, out); break;
        case 3: if constexpr (NF > 3) jr.template row<3>(p, lane, out); break;
        case 4: if constexpr (NF > 4) jr.template row<4>(p, lane, out); break;
        case 5: if constexpr (NF > 5) jr.template row<5>(p, lane, out); break;
        case 6: if constexpr (NF > 6) jr.template row<6>(p, lane, out); break;
        default: if constexpr (NF > 7) jr.template row<7>(p, lane, out); break;
        }
    } else {
        jr.template row<0>(p, lane, out);   // only the L2 reader is instantiated this wide; it ignores the tile
    }
}

// One chain (image b, read r) on one wavefront.
template <int NF, class JR>
__device__ __forceinline__ void gibbs_chain(const float* __restrict__ lin, const JR& jr, int ch, int R, int n, int sweeps,
                                            float beta0, float beta1, uint64_t seed, uint64_t offset, uint32_t chain0,
                                            uint8_t* __restrict__ states, int lane)
{
    const int b = ch / R;
    const uint32_t row = chain0 + (uint32_t)ch;   // Philox row: the GLOBAL chain index (sharding-invariant draws)
    float F[NF], thr[NF], jrow[NF];
    bool st[NF];   // this lane's spin of tile t

    // start state s_i = (u < 0.5), draw id `offset`
#pragma unroll
    for (int t = 0; t < NF; ++t) {
        const int i = 64 * t + lane;
        st[t] = false;
        F[t] = 0.f;
        if (i < n) {
            const f32x4 u = uniform4(seed, offset, row, (uint32_t)(i >> 2));
            st[t] = u[i & 3] < 0.5f;
            F[t] = lin[(long)b * n + i];
        }
    }
    // initial fields: + J[j, :] for set spins j ascending
#pragma unroll
    for (int t = 0; t < NF; ++t) {
        unsigned long long m = __ballot(st[t]);
        while (m) {
            const int j = 64 * t + __builtin_ctzll(m);
            m &= m - 1;
            row_of_tile<NF>(jr, t, j, lane, jrow);
#pragma unroll
            for (int tt = 0; tt < NF; ++tt) F[tt] = F[tt] + jrow[tt];
        }
    }

    const float lr = (sweeps > 1) ? det_logf(beta1 / beta0) : 0.f;  // geometric schedule exponent
    for (int sw = 1; sw <= sweeps; ++sw) {
        float beta = beta1;
        if (sweeps > 1) beta = beta0 * det_expf(lr * ((float)(sw - 1) / (float)(sweeps - 1)));
        // Uniforms of this sweep.  Spin i takes word i&3 of Philox block i>>2, so the 64 spins of a tile share 16
        // blocks: one pass in which lane L computes block 64*pass + L serves four tiles, and each spin fetches its word
        // from lane 16*(t&3) + (lane>>2) (ds_bpermute) -- a quarter of the Philox work of one call per spin.
        constexpr int NPASS = (NF + 3) / 4;
        f32x4 ub[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
            ub[ps] = uniform4(seed, offset + (uint64_t)sw, row, (uint32_t)(64 * ps + lane));
#pragma unroll
        for (int t = 0; t < NF; ++t) {
            const int i = 64 * t + lane;
            const int src = 16 * (t & 3) + (lane >> 2);
            const float w0 = __shfl(ub[t >> 2][0], src, kWave), w1 = __shfl(ub[t >> 2][1], src, kWave);
            const float w2 = __shfl(ub[t >> 2][2], src, kWave), w3 = __shfl(ub[t >> 2][3], src, kWave);
            const int k = lane & 3;
            const float u = k == 0 ? w0 : k == 1 ? w1 : k == 2 ? w2 : w3;
            thr[t] = (i < n) ? -(det_logf(u) - det_logf(1.0f - u)) : 0.f;
        }
#pragma unroll
        for (int t = 0; t < NF; ++t) {
            const int cnt = min(64, n - 64 * t);
            if (cnt <= 0) break;
            unsigned long long pending = (cnt >= 64) ? ~0ull : ((1ull << cnt) - 1ull);  // spins of this tile not yet visited
            while (true) {
                const bool ns = (beta * F[t]) < thr[t];
                const unsigned long long want = __ballot(ns != st[t]) & pending;
                if (want == 0ull) break;
                const int l = __builtin_ctzll(want);              // first spin that flips; the ones before it stay
                const bool was = (__ballot(st[t]) >> l) & 1ull;
                const float sg = was ? -1.0f : 1.0f;
                row_of_tile<NF>(jr, t, 64 * t + l, lane, jrow);
#pragma unroll
                for (int tt = 0; tt < NF; ++tt) F[tt] = __builtin_fmaf(sg, jrow[tt], F[tt]);
                if (lane == l) st[t] = !was;
                pending = (l >= 63) ? 0ull : (pending & (~0ull << (l + 1)));
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NF; ++t) {
        const int i = 64 * t + lane;
        if (i < n) states[(long)ch * n + i] = st[t] ? 1 : 0;
    }
}

// J rows from L2 (any n <= QBM_MAX_QUBO_VARS): 4 chains per workgroup
template <int NF>
__global__ __launch_bounds__(256) void k_qubo_gibbs(const float* __restrict__ lin, const float* __restrict__ J,
                                                    int nchains, int R, int n, int sweeps, float beta0,
                                                    float beta1, uint64_t seed, uint64_t offset, uint32_t chain0,
                                                    uint8_t* __restrict__ states)
{
    const int lane = threadIdx.x & 63;
    const int ch = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (ch >= nchains) return;
    JGlobal jr{J, n};
    jr.setup(lane);
    gibbs_chain<NF>(lin, jr, ch, R, n, sweeps, beta0, beta1, seed, offset, chain0, states, lane);
}

// J triangle in LDS: one workgroup of 16 wavefronts per CU, every wavefront walks chains ch = wave id, + total waves, ...
constexpr int kGibbsLdsWaves = 16;
template <int NF>
__global__ __launch_bounds__(64 * kGibbsLdsWaves) void k_qubo_gibbs_lds(const float* __restrict__ lin,
                                                                        const float* __restrict__ J, int nchains, int R,
                                                                        int n, int sweeps, float beta0, float beta1,
                                                                        uint64_t seed, uint64_t offset, uint32_t chain0,
                                                                        uint8_t* __restrict__ states)
{
    extern __shared__ float tri[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    for (int r = w; r < n; r += (int)blockDim.x >> 6) {
        const int S = r * n - (r * (r - 1)) / 2;
        for (int c = r + lane; c < n; c += 64) tri[S + c - r] = J[(long)r * n + c];
    }
    __syncthreads();
    JTri<NF> jr{tri, n, {}, {}};
    jr.setup(lane);
    const int waves = (int)blockDim.x >> 6;
    const int total = gridDim.x * waves;
    for (int ch = blockIdx.x * waves + w; ch < nchains; ch += total)
        gibbs_chain<NF>(lin, jr, ch, R, n, sweeps, beta0, beta1, seed, offset, chain0, states, lane);
}

// mean over reads, float64: exact count / R  (np.average(np_samples, axis=0) on {0,1} samples,
// src/model/faster_dqbm.py:780)
__global__ __launch_bounds__(256) void k_read_mean(const uint8_t* __restrict__ states, double* __restrict__ mean,
                                                   int B, int R, int n)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * n) return;
    const int b = idx / n, i = idx % n;
    int cnt = 0;
    for (int r = 0; r < R; ++r) cnt += states[((long)b * R + r) * n + i];
    mean[idx] = (double)cnt / (double)R;
}

// get_average_configuration_batch (src/model/faster_dqbm.py:754-848) from the per-image means: float64,
// sequential over the images, product then sum.  One thread per output element:
//   idx <  (1+D)*H          : W_vh[k][j]   (k < D: pixel rows; k == D: the label row, clamped phase only)
//   idx <  (1+D)*H + D      : W_vo[k]
//   idx <  (1+D)*H + D + H  : b_h[j]
//   last                    : b_o
__global__ __launch_bounds__(256) void k_batch_stats(const double* __restrict__ X, const int* __restrict__ labels,
                                                     const double* __restrict__ mean, int B, int D, int H,
                                                     double* __restrict__ W_vh, double* __restrict__ W_vo,
                                                     double* __restrict__ b_h, double* __restrict__ b_o)
{
    const int n = labels ? H : H + 1, o = labels ? 0 : 1;
    const long nW = (long)(1 + D) * H;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nW + D + H + 1) return;
    double acc = 0.0;
    if (idx < nW) {
        const int k = (int)(idx / H), j = (int)(idx % H);
        if (k < D) {
            for (int i = 0; i < B; ++i) acc += X[(long)i * D + k] * mean[(long)i * n + o + j];
        } else if (labels) {
            for (int i = 0; i < B; ++i) acc += (double)labels[i] * mean[(long)i * n + o + j];
        }
        W_vh[idx] = acc;
    } else if (idx < nW + D) {
        const int k = (int)(idx - nW);
        for (int i = 0; i < B; ++i) acc += X[(long)i * D + k] * (labels ? (double)labels[i] : mean[(long)i * n]);
        W_vo[k] = acc;
    } else if (idx < nW + D + H) {
        const int j = (int)(idx - nW - D);
        for (int i = 0; i < B; ++i) acc += mean[(long)i * n + o + j];
        b_h[j] = acc;
    } else {
        for (int i = 0; i < B; ++i) acc += labels ? (double)labels[i] : mean[(long)i * n];
        b_o[0] = acc;
    }
}

// x-dependent part of create_qubo_matrix_from (src/model/faster_dqbm.py:236-245, 273-283): float64,
// sequential over k, one thread per (image, variable); rounded to fp32 for the sampler.
__global__ __launch_bounds__(256) void k_linear_fields(const double* __restrict__ X, const int* __restrict__ labels,
                                                       const double* __restrict__ W_vh, const double* __restrict__ W_vo,
                                                       const double* __restrict__ b_h, double b_o, int B, int D, int H,
                                                       double beta_eff, float* __restrict__ lin)
{
    const int n = labels ? H : H + 1;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * n) return;
    const int i = idx / n, v = idx % n;
    const double* x = X + (long)i * D;
    double a = 0.0;
    if (!labels && v == 0) {
        for (int k = 0; k < D; ++k) a += x[k] * W_vo[k];
        lin[idx] = (float)((b_o + a) / beta_eff);
        return;
    }
    const int j = labels ? v : v - 1;
    if (labels) a += (double)labels[i] * W_vh[j];
    for (int k = 0; k < D; ++k) a += x[k] * W_vh[(long)(1 + k) * H + j];
    lin[idx] = (float)((b_h[j] + a) / beta_eff);
}

template <int NF>
int launch_gibbs(const float* lin, const float* J, int B, int R, int n, int sweeps, float beta0, float beta1,
                 uint64_t seed, uint64_t offset, uint32_t chain0, uint8_t* states, hipStream_t st)
{
    const int nch = B * R;
    const size_t tri_bytes = (size_t)n * (n + 1) / 2 * sizeof(float);
    if (NF <= 5 && tri_bytes <= 150 * 1024 && nch >= 64) {
        // the triangle of J fits the LDS of a CU (n <= 276): one persistent workgroup per CU, rows of J read from LDS
        if (tri_bytes > 65536) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_qubo_gibbs_lds<NF>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)tri_bytes);
            if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_qubo_gibbs_lds)");
        }
        // one workgroup per CU (256), as many wavefronts as even out the rounds: every wave walks `rounds` chains
        const int wgs = nch < 256 * 4 ? (nch + 3) / 4 : 256;
        const int rounds = (nch + wgs * kGibbsLdsWaves - 1) / (wgs * kGibbsLdsWaves);
        int waves = (nch + rounds * wgs - 1) / (rounds * wgs);
        waves = waves < 1 ? 1 : waves > kGibbsLdsWaves ? kGibbsLdsWaves : waves;
        hipLaunchKernelGGL((k_qubo_gibbs_lds<NF>), dim3(wgs), dim3(64 * waves), tri_bytes, st, lin, J, nch, R, n, sweeps,
                           beta0, beta1, seed, offset, chain0, states);
        QBM_CHECK_LAUNCH("k_qubo_gibbs_lds");
        return QBM_OK;
    }
    hipLaunchKernelGGL((k_qubo_gibbs<NF>), dim3((nch + 3) / 4), dim3(256), 0, st, lin, J, nch, R, n, sweeps, beta0,
                       beta1, seed, offset, chain0, states);
    QBM_CHECK_LAUNCH("k_qubo_gibbs");
    return QBM_OK;
}

}  // namespace

extern "C" {

int qbm_qubo_gibbs(const float* lin, const float* J, int B, int R, int n, int sweeps, float beta0, float beta1,
                   uint64_t seed, uint64_t offset, uint8_t* states, double* mean, void* stream)
{
    return qbm_qubo_gibbs_ex(lin, J, B, R, n, sweeps, beta0, beta1, seed, offset, 0u, states, mean, stream);
}

int qbm_qubo_gibbs_ex(const float* lin, const float* J, int B, int R, int n, int sweeps, float beta0, float beta1,
                      uint64_t seed, uint64_t offset, uint32_t chain0, uint8_t* states, double* mean, void* stream)
{
    if (B < 0 || R < 0) return fail(QBM_ERR_ARG, "qubo_gibbs: negative batch/reads");
    if (B == 0 || R == 0) return QBM_OK;
    if (n <= 0 || n > QBM_MAX_QUBO_VARS) return fail(QBM_ERR_ARG, "qubo_gibbs: n out of range (1..QBM_MAX_QUBO_VARS)");
    if (sweeps < 0) return fail(QBM_ERR_ARG, "qubo_gibbs: sweeps < 0");
    if (!(beta0 > 0.f) || !(beta1 > 0.f)) return fail(QBM_ERR_ARG, "qubo_gibbs: beta must be positive");
    if (!lin || !J || !states) return fail(QBM_ERR_ARG, "qubo_gibbs: null buffer");
    hipStream_t st = (hipStream_t)stream;
    const int NF = (n + 63) / 64;
    int rc;
    switch (NF) {
    case 1: rc = launch_gibbs<1>(lin, J, B, R, n, sweeps, beta0, beta1, seed, offset, chain0, states, st); break;
    case 2: rc = launch_gibbs<2>(lin, J, B, R, n, sweeps, beta0, beta1, seed, offset, chain0, states, st); break;
    case 3: rc = launch_gibbs<3>(lin, J, B, R, n, sweeps, beta0, beta1, seed, offset, chain0, states, st); break;
    case 4: rc = launch_gibbs<4>(lin, J, B, R, n, sweeps, beta0, beta1, seed, offset, chain0, states, st); break;
    case 5: rc = launch_gibbs<5>(lin, J, B, R, n, sweeps, beta0, beta1, seed, offset, chain0, states, st); break;
    case 6: rc = launch_gibbs<6>(lin, J, B, R, n, sweeps, beta0, beta1, seed, offset, chain0, states, st); break;
    case 7: case 8: rc = launch_gibbs<8>(lin, J, B, R, n, sweeps, beta0, beta1, seed, offset, chain0, states, st); break;
    default: rc = launch_gibbs<16>(lin, J, B, R, n, sweeps, beta0, beta1, seed, offset, chain0, states, st); break;
    }
    if (rc) return rc;
    if (mean) {
        hipLaunchKernelGGL(k_read_mean, dim3((B * n + 255) / 256), dim3(256), 0, st, states, mean, B, R, n);
        QBM_CHECK_LAUNCH("k_read_mean");
    }
    return QBM_OK;
}

int qbm_qbm_batch_stats(const double* X, const int32_t* labels, const double* mean, int B, int D, int H, double* W_vh,
                        double* W_vo, double* b_h, double* b_o, void* stream)
{
    if (B < 0) return fail(QBM_ERR_ARG, "batch_stats: B < 0");
    if (D <= 0 || H <= 0) return fail(QBM_ERR_ARG, "batch_stats: non-positive dims");
    if (!W_vh || !W_vo || !b_h || !b_o || (B > 0 && (!X || !mean))) return fail(QBM_ERR_ARG, "batch_stats: null buffer");
    const long total = (long)(1 + D) * H + D + H + 1;
    hipLaunchKernelGGL(k_batch_stats, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, labels,
                       mean, B, D, H, W_vh, W_vo, b_h, b_o);
    QBM_CHECK_LAUNCH("k_batch_stats");
    return QBM_OK;
}

int qbm_qbm_linear_fields(const double* X, const int32_t* labels, const double* W_vh, const double* W_vo,
                          const double* b_h, double b_o, int B, int D, int H, double beta_eff, float* lin,
                          void* stream)
{
    if (B < 0) return fail(QBM_ERR_ARG, "linear_fields: B < 0");
    if (B == 0) return QBM_OK;
    if (D <= 0 || H <= 0) return fail(QBM_ERR_ARG, "linear_fields: non-positive dims");
    if (!X || !W_vh || !W_vo || !b_h || !lin) return fail(QBM_ERR_ARG, "linear_fields: null buffer");
    if (beta_eff == 0.0) return fail(QBM_ERR_ARG, "linear_fields: beta_eff == 0");
    const int n = labels ? H : H + 1;
    hipLaunchKernelGGL(k_linear_fields, dim3((B * n + 255) / 256), dim3(256), 0, (hipStream_t)stream, X, labels, W_vh,
                       W_vo, b_h, b_o, B, D, H, beta_eff, lin);
    QBM_CHECK_LAUNCH("k_linear_fields");
    return QBM_OK;
}

}  // extern "C"
