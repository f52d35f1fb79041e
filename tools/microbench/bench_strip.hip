// Microbenchmark: where does k_hidden's time go?  Variants of the strip kernel with phases removed.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I qbm_image_classification_amd/csrc tools/microbench/bench_strip.hip -o gpurun_out/bench_strip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "strip_mma.cuh"
using namespace qbm;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// VAR: 0 full; 1 no staging; 2 no compute; 3 no epilogue math; 4 empty (launch + LDS alloc only); 5 staging only (no compute, no epilogue)
template <int NT, int VAR>
__global__ __launch_bounds__(256) void k_var(const float* v, const float* W, const float* d, float* p, float* s, int B, int V, int H, int KP)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (VAR == 4) { if (threadIdx.x == 0 && B < 0) p[0] = lds[0]; return; }
    const int nM = (B + 15) / 16, nN = (H + 16 * NT - 1) / (16 * NT);
    int ms, ns;
    strip_coords(nM, nN, ms, ns);
    const int m0 = ms * 16, n0 = ns * 16 * NT;
    const ViewKContig A{v, V, B, true};
    const ViewCContig Bv{W, H, H, true};
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Kq = quarter_len(V);
    const int kq0 = q * Kq, kend = min(V, kq0 + Kq);
    float* img = lds + (size_t)q * (NT + 1) * 16 * KP;
    f32x4 acc[NT];
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int klen = max(0, min(kend - kq0, KP));
    const int pieces = (klen + 15) >> 4;
    if (VAR != 1) {
        for (int pc = 0; pc < pieces; ++pc) {
            const int slot = pc * 64 + lane;
            glds16(A.addr(m0 + (slot & 15), kq0 + 4 * (slot >> 4), kend), img + pc * 256);
            for (int t = 0; t < NT; ++t)
                glds16(Bv.addr(kq0 + (slot >> 2), n0 + 16 * t + 4 * (slot & 3), kend), img + (size_t)(1 + t) * 16 * KP + pc * 256);
        }
    }
    const int fragA = (lane & 15) * 4 + (lane >> 4), fragB = lane;
    if (VAR != 2 && VAR != 5) {
        for (int pc = 0; pc < pieces; ++pc) {
            if (VAR != 1) vm_wait((pieces - 1 - pc) * (NT + 1));
            float a[4], b[NT][4];
            for (int s4 = 0; s4 < 4; ++s4) {
                a[s4] = img[(pc * 4 + s4) * 64 + fragA];
                for (int t = 0; t < NT; ++t) b[t][s4] = img[(size_t)(1 + t) * 16 * KP + (pc * 4 + s4) * 64 + fragB];
            }
            for (int s4 = 0; s4 < 4; ++s4)
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s4], b[t][s4], acc[t], 0, 0, 0);
        }
    } else {
        vm_wait(0);
    }
    constexpr int RS = 16 * NT + 4;
    for (int t = 0; t < NT; ++t)
        for (int r = 0; r < 4; ++r) img[((lane >> 4) * 4 + r) * RS + 16 * t + (lane & 15)] = acc[t][r];
    __syncthreads();
    if (tid >= 64 * NT) return;
    const int row = tid / (4 * NT), cg = tid % (4 * NT);
    const size_t wstride = (size_t)(NT + 1) * 16 * KP;
    const float* base = lds + row * RS + 4 * cg;
    const float4 p0 = *(const float4*)base, p1 = *(const float4*)(base + wstride), p2 = *(const float4*)(base + 2 * wstride), p3 = *(const float4*)(base + 3 * wstride);
    float sv[4] = {((p0.x + p1.x) + p2.x) + p3.x, ((p0.y + p1.y) + p2.y) + p3.y, ((p0.z + p1.z) + p2.z) + p3.z, ((p0.w + p1.w) + p2.w) + p3.w};
    const int i = m0 + row, j0 = n0 + 4 * cg;
    if (i >= B || j0 >= H) return;
    float pv[4], hv[4];
    if (VAR == 3 || VAR == 5) {
        for (int e = 0; e < 4; ++e) { pv[e] = sv[e]; hv[e] = sv[e]; }
    } else {
        f32x4 u = uniform4(1, 2, i, j0 >> 2);
        for (int e = 0; e < 4; ++e) { pv[e] = det_sigmoid(sv[e] + d[j0 + e]); hv[e] = pv[e] > u[e] ? 1.f : 0.f; }
    }
    *(float4*)(p + (long)i * H + j0) = make_float4(pv[0], pv[1], pv[2], pv[3]);
    *(float4*)(s + (long)i * H + j0) = make_float4(hv[0], hv[1], hv[2], hv[3]);
}

template <int NT, int VAR>
float run(const float* v, const float* W, const float* d, float* p, float* s, int B, int V, int H, int reps)
{
    const int Kq = quarter_len(V);
    int KP = (Kq + 15) / 16 * 16;
    const size_t lds = sizeof(float) * strip_lds_floats<NT>(KP);
    hipFuncSetAttribute((const void*)k_var<NT, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int nM = (B + 15) / 16, nN = (H + 16 * NT - 1) / (16 * NT);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_var<NT, VAR>), dim3(nM * nN), dim3(256), lds, 0, v, W, d, p, s, B, V, H, KP);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_var<NT, VAR>), dim3(nM * nN), dim3(256), lds, 0, v, W, d, p, s, B, V, H, KP);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps * 1000.f;
}

__global__ void k_empty(float* p) { if (threadIdx.x == 1234567) p[0] = 0; }

int main()
{
    const int B = 256, V = 784, H = 512;
    float *v, *W, *d, *p, *s;
    CK(hipMalloc(&v, sizeof(float) * B * V)); CK(hipMalloc(&W, sizeof(float) * V * H)); CK(hipMalloc(&d, sizeof(float) * H));
    CK(hipMalloc(&p, sizeof(float) * B * H)); CK(hipMalloc(&s, sizeof(float) * B * H));
    std::vector<float> hv(B * V), hW(V * H), hd(H, 0.1f);
    for (size_t i = 0; i < hv.size(); ++i) hv[i] = (float)((i * 2654435761u >> 13) & 1);
    for (size_t i = 0; i < hW.size(); ++i) hW[i] = (float)((int)((i * 2246822519u >> 8) & 255) - 128) / 1280.f;
    CK(hipMemcpy(v, hv.data(), sizeof(float) * hv.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(W, hW.data(), sizeof(float) * hW.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d, hd.data(), sizeof(float) * hd.size(), hipMemcpyHostToDevice));
    const int reps = 300;
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, p);
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, p);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("empty kernel, no LDS            : %7.2f us/launch\n", ms / reps * 1000.f);
    }
    printf("NT=2 full                       : %7.2f us\n", run<2, 0>(v, W, d, p, s, B, V, H, reps));
    printf("NT=2 no staging                 : %7.2f us\n", run<2, 1>(v, W, d, p, s, B, V, H, reps));
    printf("NT=2 no compute                 : %7.2f us\n", run<2, 2>(v, W, d, p, s, B, V, H, reps));
    printf("NT=2 no epilogue math           : %7.2f us\n", run<2, 3>(v, W, d, p, s, B, V, H, reps));
    printf("NT=2 empty (150 KB LDS launch)  : %7.2f us\n", run<2, 4>(v, W, d, p, s, B, V, H, reps));
    printf("NT=2 staging only               : %7.2f us\n", run<2, 5>(v, W, d, p, s, B, V, H, reps));
    printf("NT=4 full (128 blocks)          : %7.2f us\n", run<4, 0>(v, W, d, p, s, B, V, H, reps));
    printf("NT=1 full (512 blocks)          : %7.2f us\n", run<1, 0>(v, W, d, p, s, B, V, H, reps));
    CK(hipDeviceSynchronize());
    return 0;
}
