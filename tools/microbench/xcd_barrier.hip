// xcd_barrier.hip -- feasibility of an XCD-local phase barrier (diagnostic; GPU box).
//   * which XCD does a workgroup land on (HW_REG_XCC_ID), how many workgroups of a 256-WG grid per XCD;
//   * cost of a barrier among the workgroups of ONE XCD: plain stores -> s_waitcnt vmcnt(0) -> __syncthreads -> one relaxed
//     atomic add on the XCD's counter -> sc1 (L1-bypassing, L2-served) poll; bounded spin;
//   * visibility: after the barrier every workgroup reads, with sc1 loads, what its neighbour on the same XCD stored with
//     plain stores in this round (stale reads counted).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/xcd_barrier.hip -o tools/microbench/bin/xcd_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

struct Ctl {
    unsigned ticket[8];       // roles handed out per XCD
    unsigned arrive[8];       // monotonic arrival counters per XCD
    unsigned error;           // a spin ran out
    unsigned flags[8][32];    // variant 4: one arrival word per workgroup, one 128-byte line per XCD
};

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

__device__ __forceinline__ unsigned load_sc1(const unsigned* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int VAR>
__global__ __launch_bounds__(512) void k_probe(Ctl* ctl, unsigned* slots /* [8][32][128] */, int rounds, unsigned* out_xcc,
                                                unsigned* out_role, unsigned long long* out_cycles, unsigned* out_stale)
{
    extern __shared__ char lds[];
    __shared__ unsigned s_x, s_j;
    if (threadIdx.x == 0) {
        const unsigned x = xcc_id();
        s_x = x;
        s_j = atomicAdd(&ctl->ticket[x], 1u);
        out_xcc[blockIdx.x] = x;
        out_role[blockIdx.x] = s_j;
    }
    __syncthreads();
    const unsigned x = s_x, j = s_j;
    if (j >= 32) return;    // this XCD received more than 32 workgroups: no role (its partner XCD will time out)
    unsigned stale = 0;
    unsigned long long t0 = 0, t1 = 0;
    for (int it = 0; it < rounds; ++it) {
        // phase work: 512 B of plain stores into this workgroup's slot
        // (double-buffered: a workgroup is at most one barrier ahead of its neighbour)
        unsigned* my = slots + (it & 1) * 8 * 32 * 128;
        if (threadIdx.x < 128) my[(x * 32 + j) * 128 + threadIdx.x] = (unsigned)it * 1000u + threadIdx.x;
        if (it == 1 && threadIdx.x == 0) t0 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (VAR == 4) {
            // flag barrier: no read-modify-write.  Lane 0 publishes this workgroup's round number; the first wavefront polls the
            // XCD's 32 words with ONE L1-bypassing load per try until every word has reached the round.
            if (threadIdx.x == 0) __hip_atomic_store(&ctl->flags[x][j], (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (threadIdx.x < 64) {
                int spins = 0;
                for (;;) {
                    const unsigned f = threadIdx.x < 32 ? load_sc1(&ctl->flags[x][threadIdx.x]) : 0xffffffffu;
                    if (__all((int)(f >= (unsigned)(it + 1)))) break;
                    if (++spins > (1 << 22)) { ctl->error = 1; break; }
                }
            }
        }
        if (VAR != 0 && VAR != 4 && threadIdx.x == 0) {
            if (VAR == 1) __hip_atomic_fetch_add(&ctl->arrive[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(&ctl->arrive[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // no sc1: at this XCD's L2
            const unsigned want = 32u * (unsigned)(it + 1);
            int spins = 0;
            while (load_sc1(&ctl->arrive[x]) < want) {
                if (++spins > (1 << 22)) { ctl->error = 1; break; }
                if (VAR == 3) __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        // read the neighbour's slot with L1-bypassing loads
        if (threadIdx.x < 128) {
            const unsigned v = load_sc1(&my[(x * 32 + (j + 1) % 32) * 128 + threadIdx.x]);
            if (v != (unsigned)it * 1000u + threadIdx.x) ++stale;
        }
        // a second barrier so that the slot is not overwritten before everybody has read it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        t1 = __builtin_readcyclecounter();
        out_cycles[blockIdx.x] = (t1 - t0);
    }
    if (stale) atomicAdd(out_stale, stale);
}

int main()
{
    const int rounds = 201;
    Ctl* ctl; unsigned *slots, *xcc, *role, *stale; unsigned long long* cyc;
    hipMalloc(&ctl, sizeof(Ctl)); hipMalloc(&slots, 2 * 8 * 32 * 128 * 4); hipMalloc(&xcc, 256 * 4); hipMalloc(&role, 256 * 4);
    hipMalloc(&stale, 4); hipMalloc(&cyc, 256 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 10; ++rep) {
        const int var = rep % 5;   // 0: no cross-workgroup barrier (baseline; stale reads expected), 1: agent-scope atomic, 2: L2-local atomic, 3: 2 + s_sleep
        hipMemset(ctl, 0, sizeof(Ctl)); hipMemset(stale, 0, 4); hipMemset(slots, 0xff, 2 * 8 * 32 * 128 * 4); hipMemset(cyc, 0, 256 * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (var == 0) hipLaunchKernelGGL(k_probe<0>, dim3(256), dim3(512), 100 * 1024, 0, ctl, slots, rounds, xcc, role, cyc, stale);
        if (var == 1) hipLaunchKernelGGL(k_probe<1>, dim3(256), dim3(512), 100 * 1024, 0, ctl, slots, rounds, xcc, role, cyc, stale);
        if (var == 2) hipLaunchKernelGGL(k_probe<2>, dim3(256), dim3(512), 100 * 1024, 0, ctl, slots, rounds, xcc, role, cyc, stale);
        if (var == 4) hipLaunchKernelGGL(k_probe<4>, dim3(256), dim3(512), 100 * 1024, 0, ctl, slots, rounds, xcc, role, cyc, stale);
        if (var == 3) hipLaunchKernelGGL(k_probe<3>, dim3(256), dim3(512), 100 * 1024, 0, ctl, slots, rounds, xcc, role, cyc, stale);
        hipEventRecord(e1);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned> hx(256), hr(256); std::vector<unsigned long long> hc(256); Ctl hctl; unsigned hs;
        hipMemcpy(hx.data(), xcc, 1024, hipMemcpyDeviceToHost); hipMemcpy(hr.data(), role, 1024, hipMemcpyDeviceToHost);
        hipMemcpy(hc.data(), cyc, 2048, hipMemcpyDeviceToHost); hipMemcpy(&hctl, ctl, sizeof(Ctl), hipMemcpyDeviceToHost);
        hipMemcpy(&hs, stale, 4, hipMemcpyDeviceToHost);
        int cnt[16] = {0}, rr = 0;
        for (int b = 0; b < 256; ++b) { cnt[hx[b] & 15]++; if (hx[b] == hx[b % 8]) ++rr; }
        std::sort(hc.begin(), hc.end());
        printf("variant %d: per-XCD workgroups:", var);
        for (int i = 0; i < 8; ++i) printf(" %d", cnt[i]);
        printf(" | xcc(b)==xcc(b%%8) for %d/256 | block0 on xcc %u | error=%u stale=%u | kernel %.1f us = %.2f us per round\n",
               rr, hx[0], hctl.error, hs, ms * 1e3, ms * 1e3 / rounds);
    }
    return 0;
}
