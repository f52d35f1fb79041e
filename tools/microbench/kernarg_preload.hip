// kernarg_preload.hip -- does a wavefront of this firmware find its leading kernel arguments in SGPRs (kernel-argument
// preloading, -mllvm -amdgpu-kernarg-preload-count=N), and what does the (cold) scalar-cache round trip of the ordinary
// argument load cost?  One wavefront per workgroup stamps s_memtime at entry and again after it has USED its first pointer
// argument's value (an s_waitcnt on the argument load if there is one); the kernel is launched behind a cache-cold
// neighbour, 256 workgroups, medians printed.  Build both ways and compare:
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/kernarg_preload.hip -o tools/microbench/bin/kp_plain
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=8 tools/microbench/kernarg_preload.hip -o tools/microbench/bin/kp_preload
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void k_stamp(unsigned long long* out, const float* a, int n, int m)
{
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    // use the arguments on the scalar unit: the wait for an argument load (if any) lands here
    unsigned av = (unsigned)(unsigned long long)a + (unsigned)n + (unsigned)m;
    asm volatile("s_add_u32 %0, %0, 0" : "+s"(av));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (threadIdx.x == 0) out[blockIdx.x] = (t1 - t0) + (av & 0u);
}

__global__ void k_trash(float* p, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}

int main()
{
    const int G = 256;
    unsigned long long* d;
    float* buf;
    (void)hipMalloc(&d, G * sizeof(unsigned long long));
    (void)hipMalloc(&buf, 64 << 20);
    std::vector<unsigned long long> all;
    for (int it = 0; it < 50; ++it) {
        hipLaunchKernelGGL(k_trash, dim3((16 << 20) / 256), dim3(256), 0, 0, buf, 16 << 20);
        hipLaunchKernelGGL(k_stamp, dim3(G), dim3(64), 0, 0, d, buf, it, 3);
        std::vector<unsigned long long> h(G);
        (void)hipMemcpy(h.data(), d, G * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        if (it >= 5) all.insert(all.end(), h.begin(), h.end());
    }
    std::sort(all.begin(), all.end());
    printf("{\"what\": \"cycles (s_memtime) from a wavefront's first instruction to the use of its first kernel arguments\", "
           "\"median\": %llu, \"p10\": %llu, \"p90\": %llu, \"waves\": %zu}\n",
           all[all.size() / 2], all[all.size() / 10], all[all.size() * 9 / 10], all.size());
    return 0;
}
