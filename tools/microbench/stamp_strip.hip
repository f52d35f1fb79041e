// Diagnostic build: s_memtime stamps at the phase boundaries of strip_mma inside a k_hidden-like kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <type_traits>
__device__ unsigned long long g_stamps[1024 * 4 * 8];
#define QBM_STAMP(i)                                                                         \
    do {                                                                                      \
        unsigned long long t__;                                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");           \
        if ((threadIdx.x & 63) == 0) g_stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (i)] = t__; \
    } while (0)
__device__ int g_diag;
#define QBM_DIAG_MODE g_diag
#include "strip_mma.hip.h"
using namespace qbm;

template <class TA, int MT, int NT, bool BK4>
__global__ __launch_bounds__(256) void k_h(const TA* v, const float* y, const float* W, const float* d, const float* U, float* p, float* s, int B, int V, int H, int C, int KP)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int nM = (B + 16 * MT - 1) / (16 * MT), nN = (H + 16 * NT - 1) / (16 * NT);
    int ms, ns;
    strip_coords(nM, nN, ms, ns);
    const int m0 = ms * 16 * MT, n0 = ns * 16 * NT;
    using AV = typename std::conditional<sizeof(TA) == 1, ViewKContigU8, ViewKContig>::type;
    const AV A{v, V, B};
    using BV = typename std::conditional<BK4, ViewKContig, ViewCContig>::type;
    BV Bv;
    if constexpr (BK4) Bv = BV{W, V, H}; else Bv = BV{W, H, H, V};   // k4: W^T as [H rows][V contiguous k]
    const auto o = strip_mma<MT, NT, true, true>(A, Bv, m0, n0, V, lds, KP);
    QBM_STAMP(5);
    if (o.valid(0)) {
        const int i = m0 + o.row(0), j0 = n0 + 4 * o.cg(0);
        const float sv[4] = {o.sum[0].x, o.sum[0].y, o.sum[0].z, o.sum[0].w};
        const float4 dv = *(const float4*)(d + j0);
        float cls[4] = {0, 0, 0, 0};
        float yv[10]; float4 uv[10];
        for (int c = 0; c < 10; ++c) { yv[c] = y[(long)i * C + c]; uv[c] = *(const float4*)(U + (long)c * H + j0); }
        for (int c = 0; c < 10; ++c) { cls[0] = __builtin_fmaf(yv[c], uv[c].x, cls[0]); cls[1] = __builtin_fmaf(yv[c], uv[c].y, cls[1]); cls[2] = __builtin_fmaf(yv[c], uv[c].z, cls[2]); cls[3] = __builtin_fmaf(yv[c], uv[c].w, cls[3]); }
        f32x4 u = uniform4(1, 2, i, j0 >> 2);
        const float dd[4] = {dv.x, dv.y, dv.z, dv.w};
        float pv[4], hv[4];
        for (int e = 0; e < 4; ++e) { pv[e] = det_sigmoid((sv[e] + dd[e]) + cls[e]); hv[e] = pv[e] > u[e] ? 1.f : 0.f; }
        *(float4*)(p + (long)i * H + j0) = make_float4(pv[0], pv[1], pv[2], pv[3]);
        *(float4*)(s + (long)i * H + j0) = make_float4(hv[0], hv[1], hv[2], hv[3]);
    }
    QBM_STAMP(6);
}

template <class TA, int MT = 2, int NT = 1, bool BK4 = false>
void run(const char* name, int diag, int B = 256, int V = 784, int H = 512)
{
    hipMemcpyToSymbol(HIP_SYMBOL(g_diag), &diag, sizeof(int));
    const int C = 10;
    TA* v; float *y, *W, *d, *U, *p, *s;
    hipMalloc(&v, sizeof(TA) * B * V); hipMalloc(&y, sizeof(float) * B * C); hipMalloc(&W, sizeof(float) * V * H);
    hipMalloc(&d, sizeof(float) * H); hipMalloc(&U, sizeof(float) * C * H); hipMalloc(&p, sizeof(float) * B * H); hipMalloc(&s, sizeof(float) * B * H);
    hipMemset(v, 0, sizeof(TA) * B * V); hipMemset(y, 0, sizeof(float) * B * C); hipMemset(W, 0, sizeof(float) * V * H);
    hipMemset(d, 0, sizeof(float) * H); hipMemset(U, 0, sizeof(float) * C * H);
    const int Kq = quarter_len(V); int KP = (Kq + 63) / 64 * 64;
    if (KP > strip_max_panel<MT, NT, sizeof(TA)>()) KP = strip_max_panel<MT, NT, sizeof(TA)>();
    const size_t lds = strip_lds_bytes<MT, NT, sizeof(TA)>(KP);
    hipFuncSetAttribute((const void*)k_h<TA, MT, NT, BK4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = ((B + 16 * MT - 1) / (16 * MT)) * (H / (16 * NT));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k_h<TA, MT, NT, BK4>), dim3(grid), dim3(256), lds, 0, v, y, W, d, U, p, s, B, V, H, C, KP);
    hipEventRecord(e0);
    for (int it = 0; it < 200; ++it) hipLaunchKernelGGL((k_h<TA, MT, NT, BK4>), dim3(grid), dim3(256), lds, 0, v, y, W, d, U, p, s, B, V, H, C, KP);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("== %s: grid %d, LDS %zu B/block, %.2f us/launch back-to-back (stamped build)\n", name, grid, lds, ms / 200 * 1000);
    std::vector<unsigned long long> st(1024 * 4 * 8);
    hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamps), st.size() * 8);
    const char* names[] = {"entry->dma issued", "dma issued->first piece landed+read", "compute loop", "reduce (LDS write, barrier, sum)", "trailing barrier", "epilogue"};
    for (int ph = 0; ph < 6; ++ph) {
        std::vector<double> dts;
        for (int w = 0; w < grid * 4; ++w) dts.push_back((double)(st[w * 8 + ph + 1] - st[w * 8 + ph]));
        std::sort(dts.begin(), dts.end());
        printf("  %-40s median %7.0f  p10 %7.0f  p90 %7.0f cycles\n", names[ph], dts[dts.size() / 2], dts[dts.size() / 10], dts[dts.size() * 9 / 10]);
    }
    std::vector<double> tot;
    for (int w = 0; w < grid * 4; ++w) tot.push_back((double)(st[w * 8 + 6] - st[w * 8]));
    std::sort(tot.begin(), tot.end());
    printf("  wave lifetime entry->exit: median %.0f p90 %.0f cycles\n", tot[tot.size() / 2], tot[tot.size() * 9 / 10]);
}

int main()
{
#if defined(QBM_DIAG_NOCVT) || defined(QBM_DIAG_NOREAD)
    run<uint8_t>("u8 A, MFMA loop alone, diagnostic build (see -D flags)", 4);
#else
    run<uint8_t>("u8 A, product path (DMA two quads ahead)", 0);
    run<uint8_t>("u8 A, every DMA landed before the MFMA loop", 1);
    run<uint8_t>("u8 A, no MFMA loop", 3);
    run<uint8_t>("u8 A, MFMA loop alone (no staging, no waits)", 4);
    run<float>("fp32 A, product path", 0);
    // visible-like: K = 512 (hidden units), N = 784 -> 800 (visible units, padded here to whole strips), B operand K-contiguous
    run<uint8_t, 4, 1, true>("visible-like 64x16 strips, K=512, product path", 0, 256, 512, 784);
    run<uint8_t, 4, 1, true>("visible-like 64x16 strips, K=512, no MFMA loop", 3, 256, 512, 784);
    run<uint8_t, 4, 1, true>("visible-like 64x16 strips, K=512, MFMA loop alone", 4, 256, 512, 784);
#endif
    return 0;
}
