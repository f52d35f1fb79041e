// qbm_device.hip.h -- device-side building blocks for the gfx950 Gibbs / CD kernels.
//
// Numerical contract (DESIGN.md "Numerical contract"): IEEE binary32, round-to-nearest-even,
// compiled with -ffp-contract=off; an FMA happens only where __builtin_fmaf is written or an
// MFMA instruction is issued.  The same operation sequences are restated independently in
// oracle/qbm_oracle.c; tests/test_gpu_parity.py checks them bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qbm {

constexpr int kWave = 64;

using f32x4 = __attribute__((ext_vector_type(4))) float;

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  key = (seed lo, seed hi);
// counter = (col >> 2, row, offset lo, offset hi); element (row, col) takes word col & 3.
// ---------------------------------------------------------------------------------------------
struct Philox4 {
    uint32_t w[4];
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1,
                                                  uint32_t c2, uint32_t c3)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a mul_hi + mul_lo pair: integer multiplies run at a
        // quarter of the vector rate, and the 40 of a Philox call were a tenth of a Gibbs sweep
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{{c0, c1, c2, c3}};
}

__device__ __forceinline__ float u01(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-08f; }

// the four uniforms of columns 4*cg .. 4*cg+3 of one row
__device__ __forceinline__ f32x4 uniform4(uint64_t seed, uint64_t offset, uint32_t row, uint32_t cg)
{
    const Philox4 p = philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), cg, row, (uint32_t)offset,
                                    (uint32_t)(offset >> 32));
    return f32x4{u01(p.w[0]), u01(p.w[1]), u01(p.w[2]), u01(p.w[3])};
}

// ---------------------------------------------------------------------------------------------
// Deterministic fp32 elementary functions (fixed operation sequences).
// ---------------------------------------------------------------------------------------------
// Branch-free: the polynomial is always evaluated and the special cases are selected at the end, so
// that several independent evaluations interleave and never split the wave (same values as the
// early-return formulation restated in the oracle).
__device__ __forceinline__ float det_expf(float x)
{
    const float xc = __builtin_fminf(__builtin_fmaxf(x, -87.0f), 88.72283935546875f);  // NaN -> -87: discarded below
    const float n = __builtin_rintf(xc * 1.44269502162933349609375f);
    float r = __builtin_fmaf(n, -0.693145751953125f, xc);
    r = __builtin_fmaf(n, -1.428606765330187045037746429443359375e-06f, r);
    float p = 1.98412701138295233249664306640625e-04f;
    p = __builtin_fmaf(p, r, 1.38888892251998186111450195312500e-03f);
    p = __builtin_fmaf(p, r, 8.33333376795053482055664062500000e-03f);
    p = __builtin_fmaf(p, r, 4.16666679084300994873046875000000e-02f);
    p = __builtin_fmaf(p, r, 1.66666671633720397949218750000000e-01f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    const int ni = (int)n;
    const int h = ni / 2;
    const float s1 = __uint_as_float((uint32_t)(h + 127) << 23);
    const float s2 = __uint_as_float((uint32_t)(ni - h + 127) << 23);
    float v = (p * s1) * s2;
    v = (x > 88.72283935546875f) ? __builtin_inff() : v;
    v = (x < -87.0f) ? 0.0f : v;
    return (x != x) ? x : v;
}

__device__ __forceinline__ float det_logf(float x)
{
    uint32_t ix = __float_as_uint(x);
    ix += 0x3f800000u - 0x3f3504f3u;
    const int k = (int)(ix >> 23) - 0x7f;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    const float f = __uint_as_float(ix) - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float w = z * z;
    const float t1 = w * __builtin_fmaf(w, 0.24279078841f, 0.40000972152f);
    const float t2 = z * __builtin_fmaf(w, 0.28498786688f, 0.66666662693f);
    const float R = t2 + t1;
    const float hfsq = (0.5f * f) * f;
    const float dk = (float)k;
    float r = __builtin_fmaf(s, hfsq + R, dk * 9.0580006145e-06f);
    r = r - hfsq;
    r = r + f;
    float v = __builtin_fmaf(dk, 6.9313812256e-01f, r);
    v = (x > 0.0f) ? v : __builtin_nanf("");
    v = (x == __builtin_inff()) ? __builtin_inff() : v;
    return (x == 0.0f) ? -__builtin_inff() : v;
}

// reference: src/ClassificationRBM.py:148-149  1 / (1 + exp(-x))
__device__ __forceinline__ float det_sigmoid(float x) { return 1.0f / (1.0f + det_expf(-x)); }
// reference: src/ClassificationRBM.py:72  log(1 + exp(z)), unstabilised like the reference
__device__ __forceinline__ float det_softplus(float z) { return det_logf(1.0f + det_expf(z)); }

// ---------------------------------------------------------------------------------------------
// Wave tree [T]: butterfly p[l] = p[l] + p[l ^ off], off = 32..1.  fp32 addition is commutative,
// so every lane ends with the same bits.
// ---------------------------------------------------------------------------------------------
// On the vector ALU only -- no ds_bpermute round trips (a __shfl_xor each: ~130 cycles of LDS-crossbar latency per step, six
// dependent steps per sum, ten sums per row of the class conditional):
//   off 32 / 16  v_permlane32_swap / v_permlane16_swap of the value with a copy of itself leave {lower, lower} and {upper,
//                upper} halves (even / odd rows) in the two registers: their sum IS p[l] + p[l ^ off] in every lane;
//   off 8        row_ror:8;   off 4   row_shl:4 into banks 0, 2 and row_shr:4 into banks 1, 3;   off 2, 1   quad_perm.
// The same pairs in the same order as the shuffles: the same bits.
__device__ __forceinline__ float wave_butterfly_sum(float p)
{
    {
        const unsigned x = __float_as_uint(p);
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        p = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        const unsigned x = __float_as_uint(p);
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        p = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    p = p + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(p), 0x128, 0xF, 0xF, false));   // row_ror:8
    {
        const unsigned x = __float_as_uint(p);
        unsigned t = __builtin_amdgcn_update_dpp(0u, x, 0x104, 0xF, 0x5, false);   // row_shl:4: lanes with bit 2 clear read l + 4
        t = __builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);            // row_shr:4: lanes with bit 2 set read l - 4
        p = p + __uint_as_float(t);
    }
    p = p + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(p), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    p = p + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(p), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    return p;
}

// [Q] the NQ chains a contraction of length K is cut into: contiguous chains of Kq = roundup16(ceil(K / NQ)) terms -- a whole
// number of 16-k staging pieces; the last ones may be short or empty.  Chain q covers k in [k0, k1).  NQ = 4 ("quarters": the
// gradient, contracted over the batch) or 8 (the conditionals).  Restated in oracle/qbm_oracle.c::chain_range.
// Round 4 measured the alternative the stamps suggested -- the pieces dealt as evenly as they go (K = 784, NQ = 8: 49 pieces =
// 6,6,6,6,6,6,6,7 instead of 7,7,7,7,7,7,7,0: no idle wavefront, the critical SIMD a piece shorter) -- bit-exact against an
// oracle restated in lock-step, goldens within tolerance, and 0.05 us per k_hidden launch SLOWER in three alternating A/B runs
// of one session (5.91-5.94 against 5.86-5.91 us; k_visible, K = 512, is the same under both rules): the MFMA loop gets shorter
// (2.6 k instead of 3.0 k cycles per wavefront) but eight wavefronts asking for their first quad at once land it later (2.8 k
// instead of 2.5 k).  Kept as a diagnostic build (-DQBM_DIAG_BALANCED_CHAINS: NOT the contract the oracle restates) with its
// numbers (profiles/r04_ab_strips_chains.jsonl).
__host__ __device__ __forceinline__ void chain_range(int K, int nq, int q, int& k0, int& k1)
{
#ifdef QBM_DIAG_BALANCED_CHAINS
    const int P = (K + 15) >> 4, base = P / nq, rem = P - base * nq;
    const int extra = q - (nq - rem);                  // chains in front of q that hold one piece more (if positive)
    const int p0 = q * base + (extra > 0 ? extra : 0);
    const int np = base + (q >= nq - rem ? 1 : 0);
    k0 = 16 * p0 < K ? 16 * p0 : K;
    k1 = 16 * (p0 + np) < K ? 16 * (p0 + np) : K;
#else
    const int Kq = (((K + nq - 1) / nq) + 15) / 16 * 16;
    k0 = q * Kq < K ? q * Kq : K;
    k1 = (q + 1) * Kq < K ? (q + 1) * Kq : K;
#endif
}
// length of the longest chain (a whole number of pieces): what panels and LDS images are sized for
__host__ __device__ __forceinline__ int chain_len(int K, int nq) { return (((K + nq - 1) / nq) + 15) / 16 * 16; }
__host__ __device__ __forceinline__ int quarter_len(int K) { return chain_len(K, 4); }

}  // namespace qbm
