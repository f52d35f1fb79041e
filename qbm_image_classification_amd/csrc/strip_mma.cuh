// strip_mma.cuh -- the K-quartered strip contraction every dense product of the path runs on.
//
// One workgroup = 4 wavefronts = the 4 SIMDs of one CU.  It owns a 16 x (16*NT) strip of the
// output.  Wave q contracts K-quarter q ([Q] of the numerical contract) with
// v_mfma_f32_16x16x4_f32, NT independent accumulators per wave (NT >= 2 reaches the MFMA issue
// rate: 32-cycle issue vs 40-cycle dependent latency).  The four partial strips meet in LDS and
// are added ((q0+q1)+q2)+q3; the caller's epilogue then runs on a row-major thread map
// (thread -> one row, 4 consecutive columns) so that one Philox4x32 call serves one thread and
// stores are 16 B wide.
//
// Operand tiles go through LDS in two images chosen so that the per-lane MFMA fragment read is a
// conflict-free, lane-linear ds_read_b32 of 64 consecutive dwords:
//   "k4"  image (operand whose K index is contiguous in memory: activations [B, features], and W
//         when it is applied transposed): float4 slot (k/4)*16 + r  -> fragment dword
//         s*64 + (lane&15)*4 + (lane>>4)
//   "lin" image (operand whose M/N index is contiguous: W[k][n], P[b][n], data[b][m] for the
//         gradient): dword k*16 + r -> fragment dword s*64 + lane
// Both images are filled 1 KiB per wave-instruction in lane order (slot = piece*64 + lane), i.e.
// in the order an LDS-DMA (global_load_lds_dwordx4) writes.
#pragma once
#include "qbm_device.cuh"

namespace qbm {

// 16 zero bytes: the LDS-DMA source of every out-of-range fragment (rows beyond M/N, k beyond the quarter)
static __device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// LDS-DMA: 16 B per lane, global -> LDS with no VGPR destination.  The LDS address is the wave-uniform
// `lds_piece` + lane*16; the global source is per lane (cdna_hip_programming.md section 5, Caveat).
__device__ __forceinline__ void glds16(const float* g, float* lds_piece)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds_piece, 16, 0, 0);
}

// ---- operand views: get() returns 4 consecutive elements, zero outside the operand -----------

// element(row, k) = p[row*ld + k]; 4 consecutive k
struct ViewKContig {
    const float* p;
    long ld;
    int rows;
    bool vec;  // ld % 4 == 0 and p 16-byte aligned
    // LDS-DMA staging is legal for this strip: every float4 is wholly inside or wholly outside
    __device__ __forceinline__ bool dma(int /*c0*/, int /*width*/, int K) const { return vec && (K & 3) == 0; }
    __device__ __forceinline__ const float* addr(int row, int k, int kend) const
    {
        return (row < rows && k < kend) ? p + (long)row * ld + k : g_zero16;
    }
    __device__ __forceinline__ float4 get(int row, int k, int kend) const
    {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rows && k < kend) {
            const float* q = p + (long)row * ld + k;
            if (vec && k + 3 < kend) {
                v = *reinterpret_cast<const float4*>(q);
            } else {
                v.x = q[0];
                if (k + 1 < kend) v.y = q[1];
                if (k + 2 < kend) v.z = q[2];
                if (k + 3 < kend) v.w = q[3];
            }
        }
        return v;
    }
};

// element(k, c) = p[k*ld + c]; 4 consecutive c
struct ViewCContig {
    const float* p;
    long ld;
    int cols;
    bool vec;
    __device__ __forceinline__ bool dma(int /*c0*/, int /*width*/, int /*K*/) const { return vec && (cols & 3) == 0; }
    __device__ __forceinline__ const float* addr(int k, int c, int kend) const
    {
        return (k < kend && c < cols) ? p + (long)k * ld + c : g_zero16;
    }
    __device__ __forceinline__ float4 get(int k, int c, int kend) const
    {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < kend && c < cols) {
            const float* q = p + (long)k * ld + c;
            if (vec && c + 3 < cols) {
                v = *reinterpret_cast<const float4*>(q);
            } else {
                v.x = q[0];
                if (c + 1 < cols) v.y = q[1];
                if (c + 2 < cols) v.z = q[2];
                if (c + 3 < cols) v.w = q[3];
            }
        }
        return v;
    }
};

// Gradient A operand, augmented: m in [0,V) -> data[k][m]; [V,V+C) -> onehot/prob Y[k][m-V] (if Y);
// m == V+C -> 1 (column sums of the B operand).  element(k, m), 4 consecutive m.
struct ViewAugData {
    const float* data;  // [B, V]
    const float* Y;     // [B, C] or nullptr
    int V, C;
    bool vec;
    __device__ __forceinline__ bool dma(int c0, int width, int /*K*/) const { return vec && c0 + width <= V; }
    __device__ __forceinline__ const float* addr(int k, int m, int kend) const
    {
        return (k < kend) ? data + (long)k * V + m : g_zero16;
    }
    __device__ __forceinline__ float one(int k, int m) const
    {
        if (m < V) return data[(long)k * V + m];
        if (m < V + C) return Y ? Y[(long)k * C + (m - V)] : 0.f;
        return (m == V + C) ? 1.f : 0.f;
    }
    __device__ __forceinline__ float4 get(int k, int m, int kend) const
    {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < kend) {
            if (vec && m + 3 < V) {
                v = *reinterpret_cast<const float4*>(data + (long)k * V + m);
            } else {
                v.x = one(k, m); v.y = one(k, m + 1); v.z = one(k, m + 2); v.w = one(k, m + 3);
            }
        }
        return v;
    }
};

// Gradient B operand, augmented: n in [0,H) -> P[k][n]; n == H -> 1 (column sums of the A operand).
struct ViewAugProb {
    const float* P;  // [B, H]
    int H;
    bool ones;
    bool vec;
    __device__ __forceinline__ bool dma(int c0, int width, int /*K*/) const { return vec && c0 + width <= H; }
    __device__ __forceinline__ const float* addr(int k, int n, int kend) const
    {
        return (k < kend) ? P + (long)k * H + n : g_zero16;
    }
    __device__ __forceinline__ float one(int k, int n) const
    {
        if (n < H) return P[(long)k * H + n];
        return (ones && n == H) ? 1.f : 0.f;
    }
    __device__ __forceinline__ float4 get(int k, int n, int kend) const
    {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < kend) {
            if (vec && n + 3 < H) {
                v = *reinterpret_cast<const float4*>(P + (long)k * H + n);
            } else {
                v.x = one(k, n); v.y = one(k, n + 1); v.z = one(k, n + 2); v.w = one(k, n + 3);
            }
        }
        return v;
    }
};

// LDS floats needed by one workgroup for panel length KP (multiple of 16)
template <int NT>
__host__ __device__ constexpr int strip_lds_floats(int KP) { return 4 * (NT + 1) * 16 * KP; }

// largest panel that fits 160 KiB with 4 waves, multiple of 16
template <int NT>
__host__ __device__ constexpr int strip_max_panel() { return (163840 / 4 / ((NT + 1) * 64)) / 16 * 16; }

struct StripOut {
    float4 sum;  // ((q0+q1)+q2)+q3 for columns n0+4*cg .. +3 of row m0+row
    int row;     // 0..15
    int cg;      // 0..4*NT-1
    bool valid;  // this thread holds an output
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

// A_K4 / B_K4: the operand's K index is the contiguous one (k4 image), else lin image.
//   A_K4: A.get(row, k, kend)   else A.get(k, m, kend)
//   B_K4: B.get(col, k, kend)   else B.get(k, n, kend)
template <int NT, bool A_K4, bool B_K4, class AView, class BView>
__device__ __forceinline__ StripOut strip_mma(const AView& A, const BView& Bv, int m0, int n0, int K,
                                              float* __restrict__ lds, int KP)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave id == K quarter (wave-uniform, SGPR)
    const int Kq = quarter_len(K);
    const int kq0 = q * Kq;
    const int kend = min(K, kq0 + Kq);
    float* img = lds + (size_t)q * (NT + 1) * 16 * KP;

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool dmaA = A.dma(m0, 16, K), dmaB = Bv.dma(n0, 16 * NT, K);  // workgroup-uniform
    const int fragA = A_K4 ? ((lane & 15) * 4 + (lane >> 4)) : lane;
    const int fragB = B_K4 ? ((lane & 15) * 4 + (lane >> 4)) : lane;

    for (int kb = kq0; kb < kq0 + Kq; kb += KP) {
        const int klen = max(0, min(kend - kb, KP));
        const int pieces = (klen + 15) >> 4;  // 16 k per piece = 1 KiB per image = one LDS-DMA instruction
        // ---- stage: every piece of this wave's K range is issued before anything is waited for.
        // Issue order is piece-major (A, B_0.. of piece 0, then piece 1, ...) so that a counted vmcnt
        // releases the pieces in the order the MFMA loop consumes them.
        for (int pc = 0; pc < pieces; ++pc) {
            const int slot = pc * 64 + lane;
            if (dmaA) {
                const float* g = A_K4 ? A.addr(m0 + (slot & 15), kb + 4 * (slot >> 4), kend)
                                      : A.addr(kb + (slot >> 2), m0 + 4 * (slot & 3), kend);
                glds16(g, img + pc * 256);
            } else {
                float4 va;
                if (A_K4) va = A.get(m0 + (slot & 15), kb + 4 * (slot >> 4), kend);
                else      va = A.get(kb + (slot >> 2), m0 + 4 * (slot & 3), kend);
                reinterpret_cast<float4*>(img)[slot] = va;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float* imgB = img + (size_t)(1 + t) * 16 * KP;
                if (dmaB) {
                    const float* g = B_K4 ? Bv.addr(n0 + 16 * t + (slot & 15), kb + 4 * (slot >> 4), kend)
                                          : Bv.addr(kb + (slot >> 2), n0 + 16 * t + 4 * (slot & 3), kend);
                    glds16(g, imgB + pc * 256);
                } else {
                    float4 vb;
                    if (B_K4) vb = Bv.get(n0 + 16 * t + (slot & 15), kb + 4 * (slot >> 4), kend);
                    else      vb = Bv.get(kb + (slot >> 2), n0 + 16 * t + 4 * (slot & 3), kend);
                    reinterpret_cast<float4*>(imgB)[slot] = vb;
                }
            }
        }
        const bool counted = dmaA && dmaB;  // pure LDS-DMA staging: release piece by piece
        if (!counted) vm_wait(0);
        // ---- contract: one piece = 4 MFMA k-steps per accumulator; out-of-range k was staged as zeros.
        // The staged data is wave-private, so no workgroup barrier is needed here.
        for (int pc = 0; pc < pieces; ++pc) {
            if (counted) vm_wait((pieces - 1 - pc) * (NT + 1));
            float a[4], b[NT][4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                a[s] = img[(pc * 4 + s) * 64 + fragA];
#pragma unroll
                for (int t = 0; t < NT; ++t) b[t][s] = img[(size_t)(1 + t) * 16 * KP + (pc * 4 + s) * 64 + fragB];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[t][s], acc[t], 0, 0, 0);
        }
    }

    // partial strips -> LDS (each wave reuses the head of its own staging region), row-major
    // [16][RS], RS = 16*NT + 4 (the +4 keeps lanes l and l+16, four rows apart, on different banks)
    constexpr int RS = 16 * NT + 4;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) img[((lane >> 4) * 4 + r) * RS + 16 * t + (lane & 15)] = acc[t][r];
    __syncthreads();

    StripOut o;
    o.valid = tid < 64 * NT;
    o.row = tid / (4 * NT);
    o.cg = tid % (4 * NT);
    o.sum = make_float4(0.f, 0.f, 0.f, 0.f);
    if (o.valid) {
        const size_t wstride = (size_t)(NT + 1) * 16 * KP;
        const float* base = lds + o.row * RS + 4 * o.cg;
        const float4 p0 = *reinterpret_cast<const float4*>(base);
        const float4 p1 = *reinterpret_cast<const float4*>(base + wstride);
        const float4 p2 = *reinterpret_cast<const float4*>(base + 2 * wstride);
        const float4 p3 = *reinterpret_cast<const float4*>(base + 3 * wstride);
        o.sum.x = ((p0.x + p1.x) + p2.x) + p3.x;
        o.sum.y = ((p0.y + p1.y) + p2.y) + p3.y;
        o.sum.z = ((p0.z + p1.z) + p2.z) + p3.z;
        o.sum.w = ((p0.w + p1.w) + p2.w) + p3.w;
    }
    __syncthreads();  // the partials are consumed: the caller may run another contraction on this LDS
    return o;
}

// Speed-only remap of blockIdx -> (m strip, n strip): workgroups that read the same B panel
// (same n strip) are dealt to the same XCD (blocks b and b+8 share one), so that each XCD's L2
// holds 1/8 of the B operand.  Falls back to the plain map when nN is not a multiple of 8.
__device__ __forceinline__ void strip_coords(int nM, int nN, int& ms, int& ns, int first = 0)
{
    const int id = (int)blockIdx.x - first;  // `first` aux workgroups precede the strips in the grid
    if ((nN & 7) == 0) {
        const int x = id & 7, j = id >> 3;
        ns = x + 8 * (j / nM);
        ms = j % nM;
    } else {
        ms = id / nN;
        ns = id % nN;
    }
}

}  // namespace qbm
