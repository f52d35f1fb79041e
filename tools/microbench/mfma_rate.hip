// Issue-rate probe for v_mfma_f32_16x16x4_f32 from ONE wave per SIMD: bare, with the u8 widening VALU pair in
// front of every MFMA, with 1/2/4 accumulators, and with 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
using f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ unsigned long long g_t[4096];

template <int NACC, int MODE>
__global__ __launch_bounds__(512) void k(float* out, const unsigned* in, int iters)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    unsigned raw[4];
    for (int i = 0; i < 4; ++i) raw[i] = in[threadIdx.x * 4 + i];
    const float b = __uint_as_float(in[threadIdx.x]);
    const unsigned sh = 8 * ((threadIdx.x & 63) >> 4);
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                float a;
                if (MODE == 0) a = __uint_as_float(raw[s]);
                else if (MODE == 1) a = (float)((raw[s] >> sh) & 255u);
                else a = (float)(raw[s] & 255u);   // one VALU (cvt_ubyte0)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            }
        asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]));
    }
    asm volatile("s_nop 0" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) g_t[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// MODE 3: the 4*NACC widenings of a group as one VALU burst, then the MFMAs back to back (static byte select)
// MODE 4: as 3 with shift + cvt (per-lane byte select);  MODE 5: bare MFMAs + 4 independent VALU adds per group
template <int NACC, int MODE>
__global__ __launch_bounds__(512) void kb(float* out, const unsigned* in, int iters)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    unsigned raw[NACC];
    for (int i = 0; i < NACC; ++i) raw[i] = in[threadIdx.x * 4 + (i & 3)];
    const float b = __uint_as_float(in[threadIdx.x]);
    const unsigned sh = 8 * ((threadIdx.x & 63) >> 4);
    unsigned x0 = threadIdx.x, x1 = 3, x2 = 5, x3 = 7;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        float a[NACC][4];
        if (MODE == 5) {
            asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %1, %1, %2\n\tv_add_u32 %2, %2, %3\n\tv_add_u32 %3, %3, %0" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (MODE == 3) {
                a[i][0] = (float)(raw[i] & 255u); a[i][1] = (float)((raw[i] >> 8) & 255u);
                a[i][2] = (float)((raw[i] >> 16) & 255u); a[i][3] = (float)(raw[i] >> 24);
            } else if (MODE == 4) {
                for (int s = 0; s < 4; ++s) a[i][s] = (float)(((raw[i] + s) >> sh) & 255u);
            } else {
                for (int s = 0; s < 4; ++s) a[i][s] = __uint_as_float(raw[i]);
            }
        }
        if (MODE == 3 || MODE == 4) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("" : "+v"(a[i][0]), "+v"(a[i][1]), "+v"(a[i][2]), "+v"(a[i][3]));
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b, acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("" : "+v"(raw[i]));
    }
    asm volatile("s_nop 0" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = (float)(x0 + x1 + x2 + x3);
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) g_t[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// LDS variants: per 8 MFMAs (2 accumulators x 4 k-steps) NU8 ds_read_u8 + NB32 ds_read2st64_b32 + NB128 ds_read_b128
// issued a group ahead, with CVT widenings per group
template <int NU8, int N2ST, int NB128, int CVT>
__global__ __launch_bounds__(256) void kl(float* out, const unsigned* in, int iters)
{
    __shared__ unsigned sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = in[i & 2047] & 0x01010101u;
    __syncthreads();
    f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    const unsigned ad = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)sm + (threadIdx.x & 63) * 4;
    unsigned u8v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    __attribute__((ext_vector_type(2))) unsigned b2[2] = {{0, 0}, {0, 0}};
    __attribute__((ext_vector_type(4))) unsigned b4[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        float a[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                unsigned r = u8v[i * 4 + s] + b2[i][s & 1] + b4[i][s];
                a[i][s] = (CVT > i * 4 + s) ? (float)r : __uint_as_float(r);
            }
        asm volatile("" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[0][3]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]), "+v"(a[1][3]));
#pragma unroll
        for (int j = 0; j < NU8; ++j) asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(u8v[j]) : "v"(ad), "n"(j * 4) : "memory");
#pragma unroll
        for (int j = 0; j < N2ST; ++j) asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(b2[j]) : "v"(ad), "n"(2 * j), "n"(2 * j + 1) : "memory");
#pragma unroll
        for (int j = 0; j < NB128; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b4[j]) : "v"(ad * 4), "n"(j * 1024) : "memory");
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], 1.0f, acc[i], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(u8v[j]));
        asm volatile("" : "+v"(b2[0]), "+v"(b2[1]), "+v"(b4[0]), "+v"(b4[1]));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1];
    if ((threadIdx.x & 63) == 0) g_t[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

template <int NU8, int N2ST, int NB128, int CVT>
void gol(const char* name)
{
    float* out; unsigned* in;
    hipMalloc(&out, 4 * 256 * 512); hipMalloc(&in, 4 * 2048); hipMemset(in, 0, 4 * 2048);
    const int iters = 64;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((kl<NU8, N2ST, NB128, CVT>), dim3(256), dim3(256), 0, 0, out, in, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> t(1024);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_t), 1024 * 8);
    std::sort(t.begin(), t.end());
    printf("%-60s  1 waves/SIMD: median %6.1f cycles per MFMA per wave\n", name, t[512] / (64.0 * 8));
    hipFree(out); hipFree(in);
}

// accumulators forced into AGPRs ("+a"), as the compiler allocates them in the strip kernels
__global__ __launch_bounds__(256) void ka(float* out, const unsigned* in, int iters)
{
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const float a = __uint_as_float(in[threadIdx.x]), b = __uint_as_float(in[threadIdx.x + 1]);
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b));
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc1[1];
    if ((threadIdx.x & 63) == 0) g_t[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}
// straight-line code: N MFMAs unrolled, executed once (instruction fetch of cold code)
template <int N>
__global__ __launch_bounds__(256) void ks(float* out, const unsigned* in)
{
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const float a = __uint_as_float(in[threadIdx.x]), b = __uint_as_float(in[threadIdx.x + 1]);
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
    for (int s = 0; s < N / 2; ++s) {
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc1[1];
    if ((threadIdx.x & 63) == 0) g_t[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}
template <int N>
void gos()
{
    float* out; unsigned* in;
    hipMalloc(&out, 4 * 256 * 512); hipMalloc(&in, 4 * 2048); hipMemset(in, 0, 4 * 2048);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(ks<N>, dim3(256), dim3(256), 0, 0, out, in);
    hipDeviceSynchronize();
    std::vector<unsigned long long> t(1024);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_t), 1024 * 8);
    std::sort(t.begin(), t.end());
    printf("straight-line, %4d MFMAs executed once (3rd launch)             1 waves/SIMD: median %6.1f cycles per MFMA per wave (p10 %.1f p90 %.1f)\n", N, t[512] / (double)N, t[100] / (double)N, t[920] / (double)N);
}

void goa()
{
    float* out; unsigned* in;
    hipMalloc(&out, 4 * 256 * 512); hipMalloc(&in, 4 * 2048); hipMemset(in, 0, 4 * 2048);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(ka, dim3(256), dim3(256), 0, 0, out, in, 64);
    hipDeviceSynchronize();
    std::vector<unsigned long long> t(1024);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_t), 1024 * 8);
    std::sort(t.begin(), t.end());
    printf("%-60s  1 waves/SIMD: median %6.1f cycles per MFMA per wave\n", "2 accumulators in AGPRs, bare", t[512] / (64.0 * 8));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(ka, dim3(256), dim3(256), 98304, 0, out, in, 64);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_t), 1024 * 8);
    std::sort(t.begin(), t.end());
    printf("%-60s  1 waves/SIMD: median %6.1f cycles per MFMA per wave\n", "2 accumulators in AGPRs, bare, 96 KB dynamic LDS", t[512] / (64.0 * 8));
}

template <int NACC, int MODE>
void go(const char* name, int threads)
{
    float* out; unsigned* in;
    hipMalloc(&out, 4 * 256 * 512); hipMalloc(&in, 4 * 512 * 4); hipMemset(in, 0, 4 * 512 * 4);
    const int iters = 64;
    for (int r = 0; r < 3; ++r) { if (MODE >= 3) hipLaunchKernelGGL((kb<NACC, MODE>), dim3(256), dim3(threads), 0, 0, out, in, iters); else hipLaunchKernelGGL((k<NACC, MODE>), dim3(256), dim3(threads), 0, 0, out, in, iters); }
    hipDeviceSynchronize();
    const int nw = 256 * threads / 64;
    std::vector<unsigned long long> t(nw);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_t), nw * 8);
    std::sort(t.begin(), t.end());
    const double n = (double)iters * 4 * NACC;
    printf("%-60s %2d waves/SIMD: median %6.1f cycles per MFMA per wave\n", name, threads / 256, t[nw / 2] / n);
    hipFree(out); hipFree(in);
}

int main()
{
    go<1, 0>("1 accumulator, bare", 256);
    go<2, 0>("2 accumulators, bare", 256);
    go<4, 0>("4 accumulators, bare", 256);
    go<2, 2>("2 accumulators, cvt_ubyte0 before each MFMA", 256);
    go<2, 1>("2 accumulators, lshr + cvt before each MFMA", 256);
    go<4, 1>("4 accumulators, lshr + cvt before each MFMA", 256);
    go<2, 3>("2 acc, 8 cvt_ubyteN in one burst per 8 MFMAs", 256);
    go<2, 4>("2 acc, 8 (add+lshr+and+cvt) in one burst per 8 MFMAs", 256);
    go<4, 3>("4 acc, 16 cvt_ubyteN in one burst per 16 MFMAs", 256);
    go<2, 5>("2 acc, bare + 4 independent VALU adds per 8 MFMAs", 256);
    goa();
    gos<104>(); gos<512>(); gos<2048>();
    gol<0, 0, 0, 0>("LDS loop skeleton: 8 MFMAs, wait, no reads");
    gol<0, 2, 0, 0>("+ 2 ds_read2st64_b32 per 8 MFMAs");
    gol<8, 2, 0, 0>("+ 8 ds_read_u8 + 2 ds_read2st64_b32");
    gol<8, 2, 0, 8>("+ 8 ds_read_u8 + 2 ds_read2st64_b32 + 8 cvt");
    gol<0, 2, 2, 0>("+ 2 ds_read_b128 + 2 ds_read2st64_b32");
    gol<0, 0, 2, 0>("+ 2 ds_read_b128");
    go<2, 0>("2 accumulators, bare", 512);
    go<2, 1>("2 accumulators, lshr + cvt before each MFMA", 512);
    return 0;
}
