// mfma_bf16_accum.hip -- can the accumulate of v_mfma_f32_16x16x32_bf16 be restated bit for bit on the CPU?  (diagnostic; GPU box)
//
// VERDICT round 3, item 7 / SURVEY.md section 7: the gradient's MFMA loops are MFMA-bound (fp32 MFMA runs at the vector rate, 1/16
// of the bf16 rate on gfx950), one operand of every contraction of the path is {0,1} (exact in any format), so the other one
// could go as THREE bf16 terms (hi + mid + lo of the fp32 value: 24 mantissa bits) on the bf16 MFMA -- IF the oracle can say
// what that instruction computes, bit for bit: the parity contract of this build is "GPU == oracle on every bit".
//
// The probe: D = C + sum_{k<32} A[i][k] * B[k][j] with A = ones (bf16 1.0), C fp32, B random bf16 of widely different
// exponents (what hi / mid / lo terms look like).  Every product is then exactly a bf16 value; what the hardware does with the
// 33-term sum is compared with candidate models evaluated in exact integer arithmetic on the host:
//   exact_rne   the exact sum, rounded once to fp32, round-to-nearest-even
//   exact_rtz   the exact sum, rounded once toward zero
//   seq_k       fp32 fmaf chain over k ascending (what v_mfma_f32_16x16x4_f32 is)
//   tree        pairwise fp32 tree over k, then + C
//   chunk4/8    exact sums of 4 / 8 consecutive k, rounded to fp32 (RNE), then chained into C in order
//   fchunkN     acc = RNE(acc + exact sum of N consecutive k) -- ONE rounding per block of N products, blocks in ascending k
//               (fchunk8_rev: blocks in descending order; fchunk8_rtz: each step rounded toward zero)
//   align_trunc the terms aligned to the largest exponent of the 33 with G extra bits kept (G = 0..6), truncated, summed, rounded
// and the share of outputs each model reproduces is printed as one JSON line.  A model at 100 % over the random cases AND the
// crafted ones (cancellation, ties, a large C swallowing small products) is a restatement; anything less is not.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/microbench/mfma_bf16_accum.hip -o tools/microbench/bin/mfma_bf16_accum
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// one wave, one MFMA per case: A = ones; B[k][j], C[i][j] from memory; D out.  Operand layout of the 16x16x32 shapes:
// A[i][k] in lane i + 16 (k / 8), element k % 8;  B[k][j] in lane j + 16 (k / 8), element k % 8;  C/D[i][j] in lane j + 16 (i / 4),
// register i % 4 -- with A all ones and C[i][j] the same for every i, a wrong guess about A's or C's layout cannot change the sum.
__global__ void k_probe(const uint16_t* __restrict__ Bm /* [cases][32][16] bf16 bits */, const float* __restrict__ Cm /* [cases][16] per column */,
                        float* __restrict__ Dm /* [cases][16][16] */, int cases)
{
    const int lane = threadIdx.x;
    for (int c = blockIdx.x; c < cases; c += gridDim.x) {
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) {
            a[e] = (__bf16)1.0f;
            const int k = 8 * (lane >> 4) + e, j = lane & 15;
            const uint16_t bits = Bm[((long)c * 32 + k) * 16 + j];
            __bf16 v;
            memcpy(&v, &bits, 2);
            b[e] = v;
        }
        f32x4 acc;
        for (int r = 0; r < 4; ++r) acc[r] = Cm[(long)c * 16 + (lane & 15)];
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        for (int r = 0; r < 4; ++r) Dm[((long)c * 16 + 4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[r];
    }
}

static float bf16_to_f32(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f32_to_bf16_rne(float f)
{
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// exact arithmetic on sums of floats: every term is m * 2^e with |m| < 2^24; scaled to a common exponent they fit __int128 when the
// exponent spread is below ~90 bits (the generator keeps it below 64)
struct Exact {
    __int128 m; int e;   // value = m * 2^e
};
static Exact exact_sum(const std::vector<float>& t)
{
    int emin = 1 << 30;
    for (float x : t) if (x != 0.f) { int e; frexpf(x, &e); emin = std::min(emin, e - 24); }
    if (emin == (1 << 30)) return {0, 0};
    __int128 s = 0;
    for (float x : t) if (x != 0.f) { int e; const float fr = frexpf(x, &e); const long long mi = (long long)ldexpf(fr, 24); s += (__int128)mi << (e - 24 - emin); }
    return {s, emin};
}
static float round_exact(Exact v, int mode /* 0 RNE, 1 RTZ */)
{
    if (v.m == 0) return 0.f;
    const bool neg = v.m < 0;
    unsigned __int128 a = neg ? (unsigned __int128)(-v.m) : (unsigned __int128)v.m;
    int hb = 127;
    while (!((a >> hb) & 1)) --hb;
    int e = v.e;
    if (hb > 23) {
        const int sh = hb - 23;
        const unsigned __int128 rem = a & (((unsigned __int128)1 << sh) - 1), half = (unsigned __int128)1 << (sh - 1);
        a >>= sh;
        e += sh;
        if (mode == 0 && (rem > half || (rem == half && (a & 1)))) ++a;   // (a carry to 2^24 is still exact as a float)
    }
    const float r = ldexpf((float)(unsigned long long)a, e);
    return neg ? -r : r;
}
static float aligned_trunc(const std::vector<float>& t, int guard)
{
    int emax = -(1 << 30);
    for (float x : t) if (x != 0.f) { int e; frexpf(x, &e); emax = std::max(emax, e); }
    if (emax == -(1 << 30)) return 0.f;
    const int lsb = emax - 24 - guard;          // bits below 2^lsb are dropped (toward zero) term by term
    __int128 s = 0;
    for (float x : t) {
        if (x == 0.f) continue;
        int e; const float fr = frexpf(x, &e);
        long long mi = (long long)ldexpf(fr, 24);
        const int sh = (e - 24) - lsb;
        __int128 v = sh >= 0 ? ((__int128)mi << sh) : (mi >= 0 ? ((__int128)mi >> -sh) : -((__int128)(-mi) >> -sh));
        if (sh < -60) v = 0;
        s += v;
    }
    return round_exact({s, lsb}, 0);
}

// acc <- round(sum of the block's 8 products and acc, every term first aligned to the block's largest exponent with `guard` bits
// below an fp32 ulp of that exponent kept; floor_mode: dropped bits are dropped toward -inf (two's complement) instead of toward zero;
// final: 0 RNE, 1 RTZ
static float block_aligned(const float* p, int n, float acc, int guard, bool floor_mode, int final_mode)
{
    std::vector<float> t(p, p + n);
    t.push_back(acc);
    int emax = -(1 << 30);
    for (float x : t) if (x != 0.f) { int e; frexpf(x, &e); emax = std::max(emax, e); }
    if (emax == -(1 << 30)) return 0.f;
    const int lsb = emax - 24 - guard;
    __int128 s = 0;
    for (float x : t) {
        if (x == 0.f) continue;
        int e; const float fr = frexpf(x, &e);
        const long long mi = (long long)ldexpf(fr, 24);
        const int sh = (e - 24) - lsb;
        __int128 v;
        if (sh >= 0) v = (__int128)mi << sh;
        else if (sh < -62) v = (floor_mode && mi < 0) ? -1 : 0;
        else if (floor_mode) v = (__int128)(mi >> -sh);                       // arithmetic shift: floor
        else v = mi >= 0 ? (__int128)(mi >> -sh) : -(__int128)((-mi) >> -sh);   // toward zero
        s += v;
    }
    return round_exact({s, lsb}, final_mode);
}

int main()
{
    const int random_cases = 4096;
    std::vector<uint16_t> B;
    std::vector<float> C;
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 0.1f);
    std::uniform_real_distribution<float> ud(0.f, 1.f);
    auto add_case = [&](auto gen_b, auto gen_c) {
        for (int k = 0; k < 32; ++k)
            for (int j = 0; j < 16; ++j) B.push_back(gen_b(k, j));
        for (int j = 0; j < 16; ++j) C.push_back(gen_c(j));
    };
    // random: what the path would feed -- hi / mid / lo bf16 terms of N(0, 0.1^2) weights gated by a Bernoulli {0,1} operand,
    // accumulators of the size a chain reaches
    for (int c = 0; c < random_cases; ++c) {
        const int kind = c % 4;
        add_case([&](int, int) -> uint16_t {
            if (ud(rng) < 0.5f) return 0;          // the {0,1} operand was 0
            const float w = nd(rng);
            const uint16_t hi = f32_to_bf16_rne(w);
            const float r1 = w - bf16_to_f32(hi);
            const uint16_t mid = f32_to_bf16_rne(r1);
            const float r2 = r1 - bf16_to_f32(mid);
            const uint16_t lo = f32_to_bf16_rne(r2);
            return kind == 0 ? hi : kind == 1 ? mid : kind == 2 ? lo : (ud(rng) < 0.33f ? hi : ud(rng) < 0.5f ? mid : lo); },
                 [&](int) -> float { return kind == 3 ? nd(rng) * 30.f : nd(rng) * (kind == 0 ? 3.f : kind == 1 ? 0.01f : 1e-4f); });
    }
    // crafted: a large accumulator swallowing ones; exact cancellation; ties at half an ulp; one big term among small ones
    add_case([&](int, int) { return f32_to_bf16_rne(1.0f); }, [&](int j) { return ldexpf(1.f, 24 + (j & 3)); });
    add_case([&](int k, int) { return f32_to_bf16_rne(k & 1 ? -1.0f : 1.0f); }, [&](int j) { return (float)j * 0.125f; });
    add_case([&](int k, int j) { return f32_to_bf16_rne(ldexpf(1.f, -24 + (j & 3)) * (k < 2 ? 1.f : 0.f)); }, [&](int) { return 1.0f; });
    add_case([&](int k, int j) { return f32_to_bf16_rne(k == (j % 32) ? 256.f : ldexpf(1.f, -16)); }, [&](int) { return 0.f; });
    add_case([&](int k, int) { return f32_to_bf16_rne(ldexpf(1.f, -k)); }, [&](int j) { return j & 1 ? 1.0f : -1.0f; });
    add_case([&](int k, int) { return f32_to_bf16_rne(ldexpf(k & 1 ? 1.f : -1.f, -(k / 2))); }, [&](int) { return 3.f; });
    // pair map: C = 2^24, B[k1] = B[k2] = 1, all else 0.  2^24 + 1 is a tie that rounds back to 2^24, 2^24 + 2 is exact: the result is
    // 2^24 + 2 iff the two ones were added TOGETHER (exactly) before or when they met the accumulator.  Column j of case c tests the
    // pair (k1, k2) = (c', j') of the flattened upper triangle.
    const int pair_case0 = (int)(C.size() / 16);
    std::vector<std::pair<int, int>> pairs;
    for (int k1 = 0; k1 < 32; ++k1) for (int k2 = k1 + 1; k2 < 32; ++k2) pairs.push_back({k1, k2});
    for (size_t q0 = 0; q0 < pairs.size(); q0 += 16)
        add_case([&](int k, int j) -> uint16_t { const size_t q = q0 + j; return (q < pairs.size() && (k == pairs[q].first || k == pairs[q].second)) ? f32_to_bf16_rne(1.0f) : 0; },
                 [&](int) { return ldexpf(1.f, 24); });
    // ... and the same with the accumulator at 2^25 and two 2s / a 2 and two 1s (is a block's sum added to C in one rounding?)
    const int cases = (int)(C.size() / 16);

    uint16_t* dB; float *dC, *dD;
    hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, (size_t)cases * 256 * 4);
    hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(256), dim3(64), 0, 0, dB, dC, dD, cases);
    std::vector<float> D((size_t)cases * 256);
    if (hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("{\"error\": \"hip\"}\n"); return 1; }

    const char* names[] = {"exact_rne", "exact_rtz", "seq_k", "tree", "chunk4", "chunk8", "chunk16", "align_trunc_g0", "align_trunc_g1", "align_trunc_g2",
                           "align_trunc_g3", "align_trunc_g4", "align_trunc_g6", "fchunk4", "fchunk8", "fchunk16", "fchunk8_rev", "fchunk8_rtz",
                           "b8_g0_z_rne", "b8_g1_z_rne", "b8_g2_z_rne", "b8_g3_z_rne", "b8_g4_z_rne", "b8_g8_z_rne",
                           "b8_g0_f_rne", "b8_g1_f_rne", "b8_g2_f_rne", "b8_g3_f_rne", "b8_g4_f_rne", "b8_g8_f_rne",
                           "b8_g2_z_rtz", "b8_g3_z_rtz", "b8_g3_f_rtz", "b4_g3_z_rne", "b16_g3_z_rne", "b32_g3_z_rne"};
    constexpr int NM = 36;
    long hits[NM] = {0}, hits_crafted[NM] = {0}, total = 0, total_crafted = 0, rows_differ = 0;
    // print the pair map: row k1, '#' where the pair (k1, k2) was added together
    {
        char map[32][33];
        for (int a = 0; a < 32; ++a) { for (int b = 0; b < 32; ++b) map[a][b] = a == b ? '\\' : '.'; map[a][32] = 0; }
        for (size_t q = 0; q < pairs.size(); ++q) {
            const int c = pair_case0 + (int)(q / 16), j = (int)(q % 16);
            const float got = D[((size_t)c * 16 + 0) * 16 + j];
            const char ch = got == ldexpf(1.f, 24) + 2.f ? '#' : (got == ldexpf(1.f, 24) ? '.' : '?');
            map[pairs[q].first][pairs[q].second] = map[pairs[q].second][pairs[q].first] = ch;
        }
        fprintf(stderr, "pair map (k1 down, k2 across; # = the two products were added exactly together):\n");
        for (int a = 0; a < 32; ++a) fprintf(stderr, "%2d %s\n", a, map[a]);
    }
    for (int c = 0; c < pair_case0; ++c)
        for (int j = 0; j < 16; ++j) {
            const float got = D[((size_t)c * 16 + 0) * 16 + j];
            for (int i = 1; i < 16; ++i) if (memcmp(&D[((size_t)c * 16 + i) * 16 + j], &got, 4) != 0) ++rows_differ;   // all rows see the same sum
            std::vector<float> t(33);
            float p[32];
            for (int k = 0; k < 32; ++k) t[k] = p[k] = bf16_to_f32(B[((size_t)c * 32 + k) * 16 + j]);
            t[32] = C[(size_t)c * 16 + j];
            float m[NM];
            const Exact ex = exact_sum(t);
            m[0] = round_exact(ex, 0);
            m[1] = round_exact(ex, 1);
            { float a = t[32]; for (int k = 0; k < 32; ++k) a = fmaf(1.0f, p[k], a); m[2] = a; }
            { float q[32]; memcpy(q, p, sizeof(q)); for (int w = 16; w >= 1; w >>= 1) for (int k = 0; k < w; ++k) q[k] = q[2 * k] + q[2 * k + 1]; m[3] = q[0] + t[32]; }
            int mi = 4;
            for (int ch : {4, 8, 16}) {
                float a = t[32];
                for (int k0 = 0; k0 < 32; k0 += ch) a = a + round_exact(exact_sum(std::vector<float>(p + k0, p + k0 + ch)), 0);
                m[mi++] = a;
            }
            for (int g : {0, 1, 2, 3, 4, 6}) m[mi++] = aligned_trunc(t, g);
            auto fchunk = [&](int ch, bool rev, int mode) {
                float a = t[32];
                for (int b = 0; b < 32 / ch; ++b) {
                    const int k0 = (rev ? 32 / ch - 1 - b : b) * ch;
                    std::vector<float> u(p + k0, p + k0 + ch);
                    u.push_back(a);
                    a = round_exact(exact_sum(u), mode);
                }
                return a;
            };
            m[mi++] = fchunk(4, false, 0); m[mi++] = fchunk(8, false, 0); m[mi++] = fchunk(16, false, 0);
            m[mi++] = fchunk(8, true, 0); m[mi++] = fchunk(8, false, 1);
            auto blocks = [&](int ch, int guard, bool fl, int fin) {
                float a = t[32];
                for (int k0 = 0; k0 < 32; k0 += ch) a = block_aligned(p + k0, ch, a, guard, fl, fin);
                return a;
            };
            for (int g : {0, 1, 2, 3, 4, 8}) m[mi++] = blocks(8, g, false, 0);
            for (int g : {0, 1, 2, 3, 4, 8}) m[mi++] = blocks(8, g, true, 0);
            m[mi++] = blocks(8, 2, false, 1); m[mi++] = blocks(8, 3, false, 1); m[mi++] = blocks(8, 3, true, 1);
            m[mi++] = blocks(4, 3, false, 0); m[mi++] = blocks(16, 3, false, 0); m[mi++] = blocks(32, 3, false, 0);
            const bool crafted = c >= random_cases;   // (the pair-map cases are not scored)
            {   // dump the first mismatches of the best model so far (fchunk8: index 14)
                static int dumped = 0;
                if (dumped < 12 && memcmp(&m[14], &got, 4) != 0 && !(m[14] == 0.f && got == 0.f)) {
                    ++dumped;
                    uint32_t gb, mb; memcpy(&gb, &got, 4); memcpy(&mb, &m[14], 4);
                    fprintf(stderr, "mismatch case %d col %d: hw %.9g (%08x) fchunk8 %.9g (%08x) exact_rne %.9g  C %.9g\n", c, j, got, gb, m[14], mb, m[0], t[32]);
                    float a = t[32];
                    for (int b = 0; b < 4; ++b) {
                        std::vector<float> u(p + 8 * b, p + 8 * b + 8);
                        const float sb = round_exact(exact_sum(u), 0);
                        u.push_back(a);
                        const float an = round_exact(exact_sum(u), 0);
                        fprintf(stderr, "   block %d: sum ~ %.9g  acc %.9g -> %.9g   terms:", b, sb, a, an);
                        for (int k = 0; k < 8; ++k) fprintf(stderr, " %.4g", p[8 * b + k]);
                        fprintf(stderr, "\n");
                        a = an;
                    }
                }
            }
            for (int q = 0; q < NM; ++q) {
                const bool same = memcmp(&m[q], &got, 4) == 0 || (m[q] == 0.f && got == 0.f);
                if (same) (crafted ? hits_crafted : hits)[q]++;
            }
            (crafted ? total_crafted : total)++;
        }
    printf("{\"what\": \"v_mfma_f32_16x16x32_bf16: D = C + sum_k 1 * B[k][j] against host models, share of outputs reproduced bit for bit\", "
           "\"random_outputs\": %ld, \"crafted_outputs\": %ld, \"rows_that_differ_from_row_0\": %ld", total, total_crafted, rows_differ);
    for (int q = 0; q < NM; ++q) printf(", \"%s\": [%.5f, %.5f]", names[q], (double)hits[q] / total, (double)hits_crafted[q] / total_crafted);
    printf("}\n");
    return 0;
}
