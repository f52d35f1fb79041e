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
    static constexpr bool kExactTiles = false;   // instantiated for NF >= ceil(n / 64): a tile other than the last may be partial
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
    static constexpr bool kExactTiles = true;    // NFT == ceil(n / 64): every tile but the last holds 64 spins
    const float* tri;   // LDS
    int n;
    int ci[NFT];        // BYTE offset of S(i) - i for this lane's spins: T(i, p) at tri + ci + 4 p for i < p
    int ic[NFT];        // BYTE offset of this lane's spin indices (clamped to n - 1)
    __device__ __forceinline__ void setup(int lane)
    {
#pragma unroll
        for (int tt = 0; tt < NFT; ++tt) {
            const int i = min(64 * tt + lane, n - 1);
            ic[tt] = 4 * i;
            // (+ the triangle's LDS address: every read below adds ci -- or `up`, which is some lane's ci -- exactly once, so the
            // address needs no further add; `tri + offset` cost one v_add_u32 per flip for a base the linker resolves to 0)
            ci[tt] = 4 * (i * n - (i * (i - 1)) / 2 - i) + (int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)tri;
            asm volatile("" : "+v"(ci[tt]));   // (opaque: the compiler otherwise keeps the offset unscaled and rebuilds (x + p) << 2
                                               // in the flip loop: two vector instructions for one add)
        }
    }
    template <int T, int NF>
    __device__ __forceinline__ void row(int p, int lane, float (&out)[NF]) const
    {
        // wave-uniform: T(p, i) at tri + up + 4 i for i >= p; one vector add per tile in front of its read.
        // up = 4 (p n - p (p - 1) / 2 - p) = 2 p (2 n - 1 - p) is what lane p % 64 holds as ci[T]: ONE v_readlane instead of
        // four scalar instructions -- the flip loop is bound by the NUMBER of instructions it issues (a SIMD hands its 3.75
        // wavefronts one scalar and one vector slot every 4 cycles), not by latency: two chains in lockstep in one wavefront,
        // same count per flip plus null steps, were 18 % SLOWER
        const int up = __builtin_amdgcn_readlane(ci[T], p & 63), p4 = 4 * p;
        // the later tiles lie 256 bytes apart behind ONE address (up + 4 lane): immediate offsets, and the compiler pairs the
        // reads (ds_read2st64_b32).  Phantom lanes of the last tile read up to 252 bytes beyond the row -- the next rows, or the
        // 256 zeroed bytes the kernel keeps behind the triangle: finite values for fields nobody tests.
        const int rb = up + 4 * lane;
#pragma unroll
        for (int tt = 0; tt < NF; ++tt) {
            int ad;
            if (tt < T) ad = ci[tt] + p4;                // every spin of an earlier tile is below p
            // every spin of a later tile is above p.  (After a flip in tile 0 the compiler pairs the reads of tiles 1 and 2 and moves
            // both halves to where the packed updates of (0,1), (2,3) take them; addressing tile 1 through its own index register --
            // one v_add for two v_mov -- was 0.6 % SLOWER in alternating runs: left as it is.)
            else if (tt > T) ad = rb + 256 * tt;
            else ad = min(up + ic[tt], ci[tt] + p4);     // the spin's own tile: T(min(i, p), max(i, p)) is the SMALLER of the two
                                                         // candidates (S(r) - r grows by n - 1 - r >= 1 words per row, a column
                                                         // by one): a v_min instead of a compare and a select
            out[tt] = *(__attribute__((address_space(3))) const float*)(uintptr_t)(uint32_t)ad;
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

// The threshold of a heat-bath update: s_i = 1 iff F_i < fl(-logit(u_i) * fl(1 / beta))  (== sigma(-beta F_i) > u_i as ONE compare
// of the field; until round 4 the field was multiplied by beta at every test -- a vector instruction per flip in a loop that is
// bound by their number; the rule is this build's own, oracle and kernels state it the same way).
// -logit(u) = log((1 - u) / u): ONE logarithm of a quotient (round 4; rounds 1-3: log(1 - u) - log(u), two of them -- the
// thresholds were ~40 % of a sweep's fixed work).  u = 0 gives +inf (the spin is set), u < 1 always.
// The logarithm is the sampler's own (round 4): exponent + a degree-7 polynomial of log(1 + t) / t on [sqrt(1/2) - 1, sqrt(2) - 1]
// (absolute error 7e-8 -- a spin's probability moves by 2e-8), a fixed sequence of fp32 operations without the division,
// the range tests and the special cases of det_logf: 29 vector instructions per threshold instead of 58, four thresholds per
// lane and sweep (a sixth of a sweep's instructions at n = 256).  oracle/qbm_oracle.c states the same sequence.
__device__ __forceinline__ float neg_logit(float u)
{
    const float q = (1.0f - u) / u;   // in [2^-24, 2^24]: normal;  u = 0 -> +inf, selected below
    uint32_t ix = __float_as_uint(q);
    ix += 0x3f800000u - 0x3f3504f3u;
    const int k = (int)(ix >> 23) - 0x7f;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    const float t = __uint_as_float(ix) - 1.0f;
    float p = -0.0991365612f;
    p = __builtin_fmaf(p, t, 0.163385555f);
    p = __builtin_fmaf(p, t, -0.173631534f);
    p = __builtin_fmaf(p, t, 0.198853612f);
    p = __builtin_fmaf(p, t, -0.249577805f);
    p = __builtin_fmaf(p, t, 0.333356887f);
    p = __builtin_fmaf(p, t, -0.500006974f);
    p = __builtin_fmaf(p, t, 0.999999821f);
    const float v = __builtin_fmaf((float)k, 0.693147182f, t * p);
    return (u == 0.0f) ? __builtin_inff() : v;
}

// One chain (image b, read r) on one wavefront.
// XS ("extra spin", n = 64 NF + 1: the unclamped QUBO of a Disc_QBM with 64 k hidden nodes -- 257 variables at 256): the LAST
// spin does not open a tile of its own (63 idle lanes in every ballot, threshold and row update, and a triangle whose rows
// start on odd addresses: the n = 257 phase cost 13-15 % more than n = 256 in round 3).  The n - 1 spins in front of it fill NF
// whole tiles; spin n - 1 is a wave-uniform scalar: its row J[n-1, :] sits in NF registers (jx), its field Fx is updated by
// every flip with the flipping lane's jx (one v_readlane + one fmaf instead of an LDS read + fmaf on a fifth tile), and it is
// visited behind the last tile -- the same visit order, the same draws (Philox column n - 1), the same fmaf sequence on every
// field as the tiled form: bit-identical to the oracle, which knows nothing of tiles.
template <int NF, class JR, bool XS = false>
__device__ __forceinline__ void gibbs_chain(const float* __restrict__ lin, const JR& jr, const float* __restrict__ Jx, int ch,
                                            int R, int n, int sweeps, float beta0, float beta1,
                                            const float* __restrict__ beta_dev, uint64_t seed, uint64_t offset, uint32_t chain0,
                                            uint8_t* __restrict__ states, int lane)
{
    const int b = ch / R;
    if (beta_dev) {   // per-image annealing range [B,2] written by k_qbm_beta (no host round trip)
        beta0 = beta_dev[2 * b];
        beta1 = beta_dev[2 * b + 1];
    }
    const uint32_t row = chain0 + (uint32_t)ch;   // Philox row: the GLOBAL chain index (sharding-invariant draws)
    const int nt = XS ? n - 1 : n;                // spins held in tiles
    float F[NF], thr[NF], jrow[NF];
    // The states of a tile's 64 spins are ONE wave-uniform 64-bit mask (scalar registers): "which lanes want to flip" is a
    // ballot XOR the mask, the flip one scalar XOR -- no per-lane state, no second ballot.  (The counters said what bounds
    // this loop: vector-instruction issue, 4 cycles per instruction and wavefront, 3.75 wavefronts per SIMD; per-lane bools
    // cost a dozen vector instructions per flip in packing and unpacking alone: profiles/r03_qubo_pmc.json.)
    unsigned long long sm[NF];
    float jx[NF], Fx = 0.f;                       // XS: J[n-1, tile spins], the extra spin's field (the same in every lane)
    bool sx = false;

    // Which uniform a spin takes (the sampler's own layout, round 4): spin i = 64 t + l reads word t & 3 of Philox block
    // 64 (t >> 2) + l of the draw id -- Philox column gibbs_col(i) = 256 (t >> 2) + 4 l + (t & 3).  One Philox call of lane l
    // (block 64 pass + l) then IS the uniforms of its spins in four tiles: no exchange between lanes.  (Rounds 1-3 gave spin i
    // column i, as the RBM's draws do: a tile's 64 spins shared 16 blocks, and every sweep fetched each spin's word from
    // another lane -- four ds_bpermute and three selects per tile.)
    constexpr int NPASS = (NF + 3) / 4;
    // the extra spin n - 1 = 64 NF (XS): tile NF, lane 0
    constexpr uint32_t kXsBlock = 64u * (uint32_t)(NF >> 2);
    constexpr int kXsWord = NF & 3;
    // start state s_i = (u < 0.5), draw id `offset`
    {
        f32x4 u0[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) u0[ps] = uniform4(seed, offset, row, (uint32_t)(64 * ps + lane));
#pragma unroll
        for (int t = 0; t < NF; ++t) {
            const int i = 64 * t + lane;
            bool s0 = false;
            F[t] = 0.f;
            jx[t] = 0.f;
            if (i < nt) {
                s0 = u0[t >> 2][t & 3] < 0.5f;
                F[t] = lin[(long)b * n + i];
                if constexpr (XS) jx[t] = Jx[i];
            }
            sm[t] = __ballot(s0);
        }
    }
    if constexpr (XS) {
        const f32x4 u = uniform4(seed, offset, row, kXsBlock);
        sx = u[kXsWord] < 0.5f;
        Fx = lin[(long)b * n + n - 1];
    }
    // initial fields: + J[j, :] for set spins j ascending
#pragma unroll
    for (int t = 0; t < NF; ++t) {
        unsigned long long m = sm[t];
        while (m) {
            const int l = __builtin_ctzll(m), j = 64 * t + l;
            m &= m - 1;
            row_of_tile<NF>(jr, t, j, lane, jrow);
#pragma unroll
            for (int tt = 0; tt < NF; ++tt) F[tt] = F[tt] + jrow[tt];
            if constexpr (XS) Fx = Fx + __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(jx[t]), l));
        }
    }
    if constexpr (XS) {
        if (sx) {
#pragma unroll
            for (int tt = 0; tt < NF; ++tt) F[tt] = F[tt] + jx[tt];
            Fx = Fx + 0.0f;                       // (+ J[n-1, n-1]: the oracle's row walk includes the diagonal)
        }
    }

    const float lr = (sweeps > 1) ? det_logf(beta1 / beta0) : 0.f;  // geometric schedule exponent
    float thrx_v = 0.f;   // XS: the extra spin's thresholds of 64 sweeps, one per lane
    float ibv = 0.f;      // 1 / beta of 64 sweeps, one per lane
    for (int sw = 1; sw <= sweeps; ++sw) {
        // 1 / beta of the sweep (the thresholds are divided by beta once per sweep, not the fields multiplied at every test).  The
        // schedule is the same number in every lane: lane L evaluates it for sweep sw + L, once every 64 sweeps (an exponential
        // and two divisions per sweep were 3 % of a sweep's instructions); per sweep that leaves one v_readlane.
        if (((sw - 1) & 63) == 0) {
            const int s2 = sw + lane;
            float bt = beta1;
            if (sweeps > 1) bt = beta0 * det_expf(lr * ((float)(s2 - 1) / (float)(sweeps - 1)));
            ibv = 1.0f / bt;
        }
        const float ib = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ibv), (sw - 1) & 63));
        // Uniforms of this sweep: one Philox call per lane and four tiles (the layout above)
        f32x4 ub[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps)
            ub[ps] = uniform4(seed, offset + (uint64_t)sw, row, (uint32_t)(64 * ps + lane));
#pragma unroll
        for (int t = 0; t < NF; ++t) {
            const int i = 64 * t + lane;
            const float u = ub[t >> 2][t & 3];
            if constexpr (XS && JR::kExactTiles) thr[t] = neg_logit(u) * ib;      // whole tiles: every lane holds a spin
            else thr[t] = (i < nt) ? neg_logit(u) * ib : 0.f;
        }
        float thrx = 0.f;
        if constexpr (XS) {
            // The extra spin's uniform (word kXsWord of block kXsBlock of the sweep's draw id) is one number per sweep: lane L
            // evaluates Philox and the threshold for sweep sw + L, once every 64 sweeps -- per sweep that leaves one v_readlane
            // (a whole Philox call and a logarithm per sweep for ONE spin were a twentieth of the launch).  Same counters, same
            // function of them: the same thresholds.
            if (((sw - 1) & 63) == 0) {
                const f32x4 u = uniform4(seed, offset + (uint64_t)sw + (uint64_t)lane, row, kXsBlock);
                thrx_v = neg_logit(u[kXsWord]);
            }
            thrx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(thrx_v), (sw - 1) & 63)) * ib;
        }
#pragma unroll
        for (int t = 0; t < NF; ++t) {
            // (rotated: one test per iteration, at the bottom; an empty `pending` gives an empty `want`)
            // (the LDS reader's tiles are full but for the last: the spins behind l are -2 << l, no running mask; with the extra
            // spin outside the tiles ALL of them are full: no `pending` mask at all -- the scalar registers it takes were what
            // pushed this variant into spilling them to vector lanes inside the flip loop)
            constexpr bool all_full = XS && JR::kExactTiles;
            const int cnt = all_full ? 64 : min(64, nt - 64 * t);   // <= 0: a tile beyond n (no spin to visit: `pending` is empty)
            unsigned long long pending = (cnt >= 64) ? ~0ull : (cnt <= 0 ? 0ull : ((1ull << cnt) - 1ull));  // spins of this tile not yet visited
            const bool full = all_full || (JR::kExactTiles && t + 1 < NF);
            unsigned long long want = __ballot(F[t] < thr[t]) ^ sm[t];
            if constexpr (!all_full) want &= pending;
            // (Reading the row of the NEXT wanting lane while the current flip is applied -- the re-test nearly always confirms it
            // -- was 45 % SLOWER, 5.21 against 3.60 ms, bit-identical: the loop is bound by the instructions its four wavefronts per
            // SIMD issue, not by the LDS round trip; profiles/r04_ab_gibbs_flip.jsonl.)
            while (want) {
                const int l = __builtin_ctzll(want);              // first spin that flips; the ones before it stay
                // +-1 by the spin's state before the flip: one scalar bit test and select
                float sg;
                asm("s_bitcmp1_b64 %1, %2\n\ts_cselect_b32 %0, 0xbf800000, 0x3f800000" : "=s"(sg) : "s"(sm[t]), "s"(l) : "scc");
                row_of_tile<NF>(jr, t, 64 * t + l, lane, jrow);
#pragma unroll
                for (int tt = 0; tt < NF; ++tt) F[tt] = __builtin_fmaf(sg, jrow[tt], F[tt]);
                // (fl(Fx +- j) is what fmaf(+-1, j, Fx) rounds to: the sign goes into j on the scalar unit -- one vector add, and no
                // v_mov to get a second scalar operand into a vector register)
                if constexpr (XS)
                    Fx = Fx + __uint_as_float((uint32_t)__builtin_amdgcn_readlane(__float_as_uint(jx[t]), l) ^ (__float_as_uint(sg) & 0x80000000u));
                sm[t] ^= 1ull << l;
                if (all_full || full) {
                    want = (__ballot(F[t] < thr[t]) ^ sm[t]) & ((~0ull << l) << 1);   // the spins behind l
                } else {
                    pending &= (~0ull << l) << 1;
                    want = (__ballot(F[t] < thr[t]) ^ sm[t]) & pending;
                }
            }
        }
        if constexpr (XS) {   // spin n - 1, behind the last tile (every lane holds the same Fx, thrx: a uniform decision)
            const bool ns = __ballot(Fx < thrx) != 0ull;
            if (ns != sx) {
                const float sg = ns ? 1.0f : -1.0f;
#pragma unroll
                for (int tt = 0; tt < NF; ++tt) F[tt] = __builtin_fmaf(sg, jx[tt], F[tt]);
                Fx = __builtin_fmaf(sg, 0.0f, Fx);   // (its own diagonal entry: the oracle's row walk includes it)
                sx = ns;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NF; ++t) {
        const int i = 64 * t + lane;
        if (i < nt) states[(long)ch * n + i] = (uint8_t)((sm[t] >> lane) & 1ull);
    }
    if constexpr (XS)
        if (lane == 0) states[(long)ch * n + n - 1] = sx ? 1 : 0;
}

// J rows from L2 (any n <= QBM_MAX_QUBO_VARS): 4 chains per workgroup
template <int NF, bool XS = false>
__global__ __launch_bounds__(256) void k_qubo_gibbs(const float* __restrict__ lin, const float* __restrict__ J,
                                                    int nchains, int R, int n, int sweeps, float beta0,
                                                    float beta1, const float* __restrict__ beta_dev, uint64_t seed,
                                                    uint64_t offset, uint32_t chain0, uint8_t* __restrict__ states)
{
    const int lane = threadIdx.x & 63;
    const int ch = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (ch >= nchains) return;
    JGlobal jr{J, n};
    jr.setup(lane);
    gibbs_chain<NF, JGlobal, XS>(lin, jr, J + (long)(n - 1) * n, ch, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, lane);
}

// J triangle in LDS: one workgroup of 16 wavefronts per CU, every wavefront walks chains ch = wave id, + total waves, ...
constexpr int kGibbsLdsWaves = 16;
// XS: the triangle covers the n - 1 spins held in tiles (whole tiles: rows start on multiples of 4 bytes x 64 k); the last
// spin's row comes from J itself, once per chain (gibbs_chain)
template <int NF, bool XS = false>
__global__ __launch_bounds__(64 * kGibbsLdsWaves) void k_qubo_gibbs_lds(const float* __restrict__ lin,
                                                                        const float* __restrict__ J, int nchains, int R,
                                                                        int n, int sweeps, float beta0, float beta1,
                                                                        const float* __restrict__ beta_dev, uint64_t seed,
                                                                        uint64_t offset, uint32_t chain0,
                                                                        uint8_t* __restrict__ states)
{
    extern __shared__ float tri[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int nt = XS ? n - 1 : n;   // spins in tiles = rows and columns of the triangle
    for (int r = w; r < nt; r += (int)blockDim.x >> 6) {
        const int S = r * nt - (r * (r - 1)) / 2;
        for (int c = r + lane; c < nt; c += 64) tri[S + c - r] = J[(long)r * n + c];
    }
    if (threadIdx.x < 64) tri[nt * (nt + 1) / 2 + threadIdx.x] = 0.f;   // (the pad JTri::row's phantom lanes may read)
    __syncthreads();
    JTri<NF> jr{tri, nt, {}, {}};
    jr.setup(lane);
    const int waves = (int)blockDim.x >> 6;
    // chains are dealt to the WORKGROUPS round-robin (workgroup b: chains b, b + grid, b + 2 grid, ...) and inside a workgroup
    // to its wavefronts: every CU gets the same number of chains +-1.  (Dealt to the wavefronts of the whole grid instead,
    // 7300 chains on 4096 wavefronts leave the last 56 workgroups without a second chain: a fifth of the CUs idle for half
    // of the launch.)  Draws depend on the chain index only, not on who runs it.
    for (int ch = (int)blockIdx.x + (int)gridDim.x * w; ch < nchains; ch += (int)gridDim.x * waves)
        gibbs_chain<NF, JTri<NF>, XS>(lin, jr, J + (long)(n - 1) * n, ch, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, lane);
}

// ---------------------------------------------------------------------------------------------
// Sparse couplings (round 4): a QUBO whose triangle does not fit the LDS (n > 276) but whose NON-ZERO couplings do -- the
// conv-QBM QUBOs of src/qubo/sampler.py's sample_Q protocol: n = 717, 10 092 non-zero couplings of 513 372, every row of J a
// 2.9 KB read from L2 of which 16 words matter.  The rows live in LDS as CSR entries {byte offset of the column, value}; the
// chain's FIELDS live in LDS as well (n floats per wavefront), because a flip now touches the fields of the row's few
// neighbours, whichever lanes would hold them: one ds_read_b64 of the row's entries (lane e takes entry e) and one
// ds_add_f32 into F[column] -- no return value, different addresses within a row, and the LDS executes a wavefront's
// operations in order, so the re-test of the tile (a ds_read_b32 of its 64 fields) sees them.  A flip costs two LDS round
// trips instead of an L2 one (~800 cycles) plus NF fused multiply-adds.
// Same visit order, same draws, same thresholds as gibbs_chain; a field receives sg * J_pk for the flips p in sweep order:
// fl(F +- J) is what fmaf(+-1, J, F) rounds to, and the entries the sparse row leaves out are the oracle's fmaf(+-1, 0, F) = F
// (a field of -0 stays -0 instead of becoming +0: equal in every comparison made of it).  Bit-identical states.
// (ds_add_f32 rounds to nearest even and keeps denormals under this build's FP mode: tests/test_gpu_parity.py holds a QUBO of
// denormal couplings against the oracle.)
// ---------------------------------------------------------------------------------------------
struct CsrHead {        // front of the CSR workspace (device memory): include/qbm_hip.h, qbm_qubo_csr_build
    int status;         // 1: more non-zero couplings than the workspace (or the LDS) holds -- the sampler launch does nothing
    int nnz;
    int cap;
    int pad;
};

// non-zero entries per row: one wavefront per row
__global__ __launch_bounds__(64) void k_csr_count(const float* __restrict__ J, int n, int* __restrict__ cnt)
{
    const int r = blockIdx.x, lane = threadIdx.x;
    int c = 0;
    for (int k0 = 0; k0 < n; k0 += 64) {
        const int k = k0 + lane;
        const bool nz = k < n && k != r && J[(long)r * n + k] != 0.0f;
        c += __builtin_popcountll(__ballot(nz));
    }
    if (lane == 0) cnt[r] = c;
}

// rowptr and the entries in ascending column order: one wavefront per row (its start = the sum of the counts in front of it)
__global__ __launch_bounds__(64) void k_csr_fill(const float* __restrict__ J, int n, const int* __restrict__ cnt,
                                                 CsrHead* __restrict__ head, int* __restrict__ rowptr, uint2* __restrict__ ent, int cap)
{
    const int r = blockIdx.x, lane = threadIdx.x;
    int s = 0;
    for (int i = lane; i < r; i += 64) s += cnt[i];
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    int pos = s;
    if (lane == 0) rowptr[r] = s;
    for (int k0 = 0; k0 < n; k0 += 64) {
        const int k = k0 + lane;
        const float v = (k < n && k != r) ? J[(long)r * n + k] : 0.0f;
        const unsigned long long m = __ballot(v != 0.0f);
        const int e = pos + __builtin_popcountll(m & ((1ull << lane) - 1ull));
        if (v != 0.0f && e < cap) ent[e] = make_uint2((unsigned)(4 * k), __float_as_uint(v));
        pos += __builtin_popcountll(m);
    }
    if (r == n - 1 && lane == 0) {
        rowptr[n] = pos;
        head->nnz = pos;
        head->cap = cap;
        head->status = pos > cap ? 1 : 0;
    }
}

constexpr int kCsrMaxWaves = 8;
constexpr int kCsrLdsBytes = 160 * 1024;

// HELP (one chain per workgroup: at most 256 chains, sample_Q's 100 reads): a SECOND wavefront beside the chain's computes the
// thresholds of sweep s + 1 (Philox, 12 logarithms per lane at n = 717: 3.2 us, an eighth of a sweep) into LDS while the first
// one walks sweep s -- the chain's wavefront has the CU to itself and nothing else to fill its waits with.  One workgroup barrier
// per sweep; the same thresholds, the same states.
template <int NF, bool HELP>
__global__ __launch_bounds__(64 * kCsrMaxWaves) void k_qubo_gibbs_csr(const float* __restrict__ lin, const CsrHead* __restrict__ head,
                                                                       const int* __restrict__ rowptr, const uint2* __restrict__ ent,
                                                                       int ent_cap /* entries the LDS holds */, int nchains, int R, int n,
                                                                       int sweeps, float beta0, float beta1,
                                                                       const float* __restrict__ beta_dev, uint64_t seed,
                                                                       uint64_t offset, uint32_t chain0, uint8_t* __restrict__ states)
{
    extern __shared__ uint2 s_ent[];
    if (head->status != 0) return;                       // (uniform: every workgroup reads the same word)
    const int nnz = rowptr[n];
    if (nnz > ent_cap) return;                           // (the host sized the LDS for head->cap: cannot happen with status == 0)
    for (int i = threadIdx.x; i < nnz; i += blockDim.x) s_ent[i] = ent[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const bool helper = HELP && wv == 1;                  // (HELP: wavefront 0 walks the chain, wavefront 1 prepares its thresholds)
    const int w = HELP ? 0 : wv;
    const int waves = HELP ? 1 : (int)blockDim.x >> 6;
    float* Fw = reinterpret_cast<float*>(s_ent + ent_cap) + w * 64 * NF;   // this wavefront's fields
    float* thb = reinterpret_cast<float*>(s_ent + ent_cap) + 64 * NF;      // HELP: thresholds [2][64 NF], by the sweep's parity
    // lane l: where the entries of the row of spin 64 t + l lie (byte offset in s_ent, 24 bits) and how many (8 bits, 255 = more)
    unsigned rinfo[NF];
#pragma unroll
    for (int t = 0; t < NF; ++t) {
        const int i = 64 * t + lane;
        const int r0 = i < n ? rowptr[i] : 0, r1 = i < n ? rowptr[i + 1] : 0;
        rinfo[t] = (unsigned)(8 * r0) | ((unsigned)min(r1 - r0, 255) << 24);
    }
    // sg * J[p, :] into the fields: lane e takes entry rs + e (rows of more than 64 neighbours go round again)
    auto add_row = [&](int rs, int re, unsigned sgbit) {
        for (int e0 = rs; e0 < re; e0 += 64) {
            const int e = e0 + lane;
            if (e < re) {
                const uint2 en = s_ent[e];
                float* f = reinterpret_cast<float*>(reinterpret_cast<char*>(Fw) + en.x);
                __hip_atomic_fetch_add(f, __uint_as_float(en.y ^ sgbit), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
    };
    for (int ch = (int)blockIdx.x + (int)gridDim.x * w; ch < nchains; ch += (int)gridDim.x * waves) {
        const int b = ch / R;
        float b0 = beta0, b1 = beta1;
        if (beta_dev) {
            b0 = beta_dev[2 * b];
            b1 = beta_dev[2 * b + 1];
        }
        const uint32_t row = chain0 + (uint32_t)ch;
        float thr[NF];
        unsigned long long sm[NF];
        // start state s_i = (u < 0.5), draw id `offset` (the draw layout of gibbs_chain: lane l's Philox block 64 pass + l holds the
        // uniforms of its spins in four tiles);  F = lin
        constexpr int NPASS = (NF + 3) / 4;
#pragma unroll
        for (int t = 0; t < NF; ++t) sm[t] = 0ull;
        if (!helper) {
            f32x4 u0[NPASS];
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) u0[ps] = uniform4(seed, offset, row, (uint32_t)(64 * ps + lane));
#pragma unroll
            for (int t = 0; t < NF; ++t) {
                const int i = 64 * t + lane;
                bool s0 = false;
                float f0 = 0.f;
                if (i < n) {
                    s0 = u0[t >> 2][t & 3] < 0.5f;
                    f0 = lin[(long)b * n + i];
                }
                Fw[i] = f0;
                sm[t] = __ballot(s0);
            }
        }
        // initial fields: + J[j, :] for set spins j ascending
#pragma unroll
        for (int t = 0; t < NF; ++t) {
            unsigned long long m = sm[t];
            while (m) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                add_row(rowptr[64 * t + l], rowptr[64 * t + l + 1], 0u);
            }
        }
        const float lr = (sweeps > 1) ? det_logf(b1 / b0) : 0.f;
        float ibv = 0.f;
        // thresholds of sweep sw (called for sw = 1, 2, ... in order): uniforms as in gibbs_chain (one Philox call per lane and four
        // tiles); into registers, or (HELP) into the LDS buffer of the sweep's parity
        auto thresholds = [&](int sw) {
            if (((sw - 1) & 63) == 0) {   // (as in gibbs_chain: lane L holds 1 / beta of sweep sw + L)
                const int s2 = sw + lane;
                float bt = b1;
                if (sweeps > 1) bt = b0 * det_expf(lr * ((float)(s2 - 1) / (float)(sweeps - 1)));
                ibv = 1.0f / bt;
            }
            const float ib = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ibv), (sw - 1) & 63));
            f32x4 ub[NPASS];
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps)
                ub[ps] = uniform4(seed, offset + (uint64_t)sw, row, (uint32_t)(64 * ps + lane));
#pragma unroll
            for (int t = 0; t < NF; ++t) {
                const int i = 64 * t + lane;
                const float th = (i < n) ? neg_logit(ub[t >> 2][t & 3]) * ib : 0.f;
                if constexpr (HELP) thb[(sw & 1) * 64 * NF + i] = th;
                else thr[t] = th;
            }
        };
        if constexpr (HELP) {
            if (helper && sweeps >= 1) thresholds(1);
            __syncthreads();
        }
        for (int sw = 1; sw <= sweeps; ++sw) {
            if constexpr (HELP) {
                if (helper) {
                    if (sw < sweeps) thresholds(sw + 1);
                } else {
#pragma unroll
                    for (int t = 0; t < NF; ++t) thr[t] = thb[(sw & 1) * 64 * NF + 64 * t + lane];
                }
            } else {
                thresholds(sw);
            }
            if (!helper) {
#pragma unroll
            for (int t = 0; t < NF; ++t) {
                const int cnt = min(64, n - 64 * t);
                const unsigned long long tile = (cnt >= 64) ? ~0ull : (cnt <= 0 ? 0ull : ((1ull << cnt) - 1ull));
                // (an LDS pointer by type: through a generic volatile pointer the re-read is a FLAT load with the system-coherence
                // bits set: 7 % of the launch)
                const volatile __attribute__((address_space(3))) float* Ft =
                    (const volatile __attribute__((address_space(3))) float*)(Fw + 64 * t + lane);
                unsigned long long want = (__ballot(*Ft < thr[t]) ^ sm[t]) & tile;
                // One flip = readlane (where the row's entries lie, how many) -> ds_read_b64 (lane e: entry e) -> ds_add_f32 into
                // the neighbours' fields -> ds_read_b32 of the tile's fields -> ballot.  What a flip costs on a wavefront that has
                // its CU to itself is the NUMBER of instructions (each waits for the one before) rather than the two LDS round trips:
                // reading the next candidate's entries one flip ahead (one round trip instead of two) took 20 more instructions and
                // was 40 % SLOWER (profiles/r04_ab_csr_sampler.jsonl).
                while (want) {
                    const int l = __builtin_ctzll(want);
                    const unsigned sgbit = ((sm[t] >> l) & 1ull) ? 0x80000000u : 0u;   // -J if the spin was set
                    const unsigned info = (unsigned)__builtin_amdgcn_readlane((int)rinfo[t], l);
                    const unsigned deg = info >> 24;
                    if (lane < (int)deg) {
                        const uint2 en = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(s_ent) + (info & 0xffffffu) + 8 * lane);
                        __hip_atomic_fetch_add(reinterpret_cast<float*>(reinterpret_cast<char*>(Fw) + en.x), __uint_as_float(en.y ^ sgbit),
                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
                    if (deg > 64u) {   // (a row of more than 64 neighbours: the rest of it)
                        const int p = 64 * t + l;
                        add_row(rowptr[p] + 64, rowptr[p + 1], sgbit);
                    }
                    sm[t] ^= 1ull << l;
                    want = (__ballot(*Ft < thr[t]) ^ sm[t]) & tile & ((~0ull << l) << 1);
                }
            }
            }
            if constexpr (HELP) __syncthreads();   // sweep sw walked, the thresholds of sweep sw + 1 written
        }
        if (!helper) {
#pragma unroll
            for (int t = 0; t < NF; ++t) {
                const int i = 64 * t + lane;
                if (i < n) states[(long)ch * n + i] = (uint8_t)((sm[t] >> lane) & 1ull);
            }
        }
    }
}

// mean over reads, float64: exact count / R  (np.average(np_samples, axis=0) on {0,1} samples,
// src/model/faster_dqbm.py:780)
__global__ __launch_bounds__(256) void k_read_mean(const uint8_t* __restrict__ states, double* __restrict__ mean,
                                                   int B, int R, int n, const int* __restrict__ skip /* NULL, or a status word: != 0 -> nothing is written */)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * n) return;
    if (skip && *skip != 0) return;   // (the sparse sampler gave up: its states are not there, the means stay untouched too)
    const int b = idx / n, i = idx % n;
    int cnt = 0;   // (an integer count: any order; 10 reads requested together instead of one round trip per read)
    const uint8_t* s = states + (long)b * R * n + i;
    int r = 0;
    for (; r + 10 <= R; r += 10) {
        int v[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) v[u] = s[(long)(r + u) * n];
#pragma unroll
        for (int u = 0; u < 10; ++u) cnt += v[u];
    }
    for (; r < R; ++r) cnt += s[(long)r * n];
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
// Round 4: the sum is a chain of 2 D dependent float64 operations per output -- 5 us of arithmetic at D = 784 -- but with one
// thread per output fetching its own operands the kernel waited for a round trip to L2 per 8 terms: 58 us, twice per training
// minibatch.  Now a workgroup owns 64 variables x 4 images (one wavefront per image): the weights of its 64 columns and the
// pixels of its 4 images pass through LDS in chunks of 56 terms, the next three chunks on their way (42 loads per thread in
// flight) while the current one is summed: operands come from LDS (the pixel is one broadcast read, the weights 512 contiguous bytes).
// Same terms, same order, product then sum.
constexpr int kLfKC = 56, kLfIm = 4;
__global__ __launch_bounds__(64 * kLfIm) void k_linear_fields(const double* __restrict__ X, const int* __restrict__ labels,
                                                              const double* __restrict__ W_vh, const double* __restrict__ W_vo,
                                                              const double* __restrict__ b_h, double b_o_val,
                                                              const double* __restrict__ b_o_dev, int B, int D, int H,
                                                              double beta_eff, float* __restrict__ lin)
{
    __shared__ double wbuf[2][kLfKC][64];
    __shared__ double xbuf[2][kLfIm][kLfKC];
    const double b_o = b_o_dev ? b_o_dev[0] : b_o_val;   // device-resident parameters: the output bias by address
    const int n = labels ? H : H + 1;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int v = 64 * (int)blockIdx.x + lane;            // this lane's variable: the column it stages AND the one it sums
    const int i = kLfIm * (int)blockIdx.y + w;            // this wavefront's image
    const bool outnode = !labels && v == 0;               // unclamped variable 0 = the output node: its "column" is W_vo
    const int j = labels ? v : v - 1;
    const double* wcol = outnode ? W_vo : W_vh + (long)H + j;   // term k: wcol[k * wstride]
    const long wstride = outnode ? 1 : H;
    constexpr int NR = kLfKC / kLfIm;                     // rows of a chunk each thread stages (its own column: k = w + 4 r)
    const int xi = tid / kLfKC, xk = tid % kLfKC;         // the pixel it stages (threads 0 .. 4 KC - 1)
    const int ximg = kLfIm * (int)blockIdx.y + xi;
    // kLfAhead chunks are in flight at any time (registers are where they land: a chunk's loads are issued kLfAhead - 1
    // iterations before it is written to LDS -- one chunk is summed in 0.6 us, a first touch of the weights takes ~2 us)
    constexpr int kLfAhead = 4;
    double wr[kLfAhead][NR], xr[kLfAhead];
    auto fetch = [&](int slot, int k0) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int k = k0 + w + kLfIm * r;
            wr[slot][r] = (v < n && k < D) ? wcol[(long)k * wstride] : 0.0;
        }
        xr[slot] = (tid < kLfIm * kLfKC && ximg < B && k0 + xk < D) ? X[(long)ximg * D + k0 + xk] : 0.0;
    };
    auto stash = [&](int slot, int buf) {
#pragma unroll
        for (int r = 0; r < NR; ++r) wbuf[buf][w + kLfIm * r][lane] = wr[slot][r];
        if (tid < kLfIm * kLfKC) xbuf[buf][xi][xk] = xr[slot];
    };
    double a = 0.0;
    if (labels && v < n && i < B) a += (double)labels[i] * W_vh[j];
    const int nc = (D + kLfKC - 1) / kLfKC;
#pragma unroll
    for (int u = 0; u < kLfAhead - 1; ++u)
        if (u < nc) fetch(u, u * kLfKC);
    stash(0, 0);
    __syncthreads();
    for (int c0 = 0; c0 < nc; c0 += kLfAhead) {
#pragma unroll
        for (int u = 0; u < kLfAhead; ++u) {     // (chunk c = c0 + u lives in register slot u: the slots are compile-time)
            const int c = c0 + u;
            if (c >= nc) break;
            const int cur = c & 1;
            if (c + kLfAhead - 1 < nc) fetch((u + kLfAhead - 1) % kLfAhead, (c + kLfAhead - 1) * kLfKC);
            const int kmax = min(kLfKC, D - c * kLfKC);
            int kk = 0;
            if (kmax == kLfKC) {
                // a whole chunk: the operands of the NEXT 8 terms are read from LDS while the current 8 are added (the sum is a
                // chain of dependent adds: nothing but that chain should be waited for)
                double xv[8], wv[8], xn[8], wn[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { xv[q] = xbuf[cur][w][q]; wv[q] = wbuf[cur][q][lane]; }
#pragma unroll
                for (int g8 = 0; g8 < kLfKC / 8; ++g8) {
                    if (g8 + 1 < kLfKC / 8) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) { xn[q] = xbuf[cur][w][8 * g8 + 8 + q]; wn[q] = wbuf[cur][8 * g8 + 8 + q][lane]; }
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) a += xv[q] * wv[q];
#pragma unroll
                    for (int q = 0; q < 8; ++q) { xv[q] = xn[q]; wv[q] = wn[q]; }
                }
                kk = kLfKC;
            }
            for (; kk + 8 <= kmax; kk += 8) {
                double xv[8], wv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) { xv[q] = xbuf[cur][w][kk + q]; wv[q] = wbuf[cur][kk + q][lane]; }
#pragma unroll
                for (int q = 0; q < 8; ++q) a += xv[q] * wv[q];
            }
            for (; kk < kmax; ++kk) a += xbuf[cur][w][kk] * wbuf[cur][kk][lane];
            if (c + 1 < nc) stash((u + 1) % kLfAhead, cur ^ 1);
            __syncthreads();
        }
    }
    if (v < n && i < B) lin[(long)i * n + v] = (float)(((outnode ? b_o : b_h[j]) + a) / beta_eff);
}

// ---------------------------------------------------------------------------------------------
// Device-resident Disc_QBM training state.  theta = [W_vh (1+D)*H | W_vo D | b_h H | b_o 1],
// float64 like the reference's numpy arrays (src/model/faster_dqbm.py:196-216); statistics use the same layout.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline long qbm_theta_len(int D, int H) { return (long)(1 + D) * H + D + H + 1; }

// sum_b |J_ab| (ascending b, float64) and the smallest non-zero |J_ab| of one row, from its magnitudes in LDS: ONE thread (the
// sum's order is the contract).  Eight magnitudes are read together in front of their eight adds.  (Not what bounds these
// launches.)
__device__ __forceinline__ void row_stats(const float* mag /* LDS, 16-byte aligned */, int n, double* rowabs, double* rowmin)
{
    double sum = 0.0, mn = 0.0;
    int b = 0;
    for (; b + 8 <= n; b += 8) {
        const float4 m0 = *reinterpret_cast<const float4*>(mag + b), m1 = *reinterpret_cast<const float4*>(mag + b + 4);
        const float m[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const double av = (double)m[q];
            sum += av;
            if (av > 0.0 && (mn == 0.0 || av < mn)) mn = av;
        }
    }
    for (; b < n; ++b) {
        const double av = (double)mag[b];
        sum += av;
        if (av > 0.0 && (mn == 0.0 || av < mn)) mn = av;
    }
    *rowabs = sum;
    *rowmin = mn;
}

// Image-independent couplings of the QUBO (create_qubo_matrix_from, :242-243 clamped, :256-271 unclamped) as the sampler
// wants them: symmetric fp32 J with zero diagonal, J_ab = fp32(Q_ab / beta_eff) of the upper-triangular Q.  One wavefront
// per row: the lanes write the row's entries, lane 0 then adds their magnitudes in ascending column order (float64) for
// the row's sum of |J|; the smallest non-zero |J| of the row (0 if none) is order-independent.  Both feed k_qbm_beta.
__global__ __launch_bounds__(64) void k_qbm_couplings(const double* __restrict__ w_label /* W_vh row 0: [H], unclamped only */,
                                                      const double* __restrict__ W_hh /* [H,H] upper triangle, or null */,
                                                      int H, int clamped, double beta_eff, float* __restrict__ J,
                                                      double* __restrict__ rowabs, double* __restrict__ rowmin)
{
    __shared__ __attribute__((aligned(16))) float mag[QBM_MAX_QUBO_VARS];
    const int n = clamped ? H : H + 1, o = clamped ? 0 : 1;
    const int a = blockIdx.x, lane = threadIdx.x;
    for (int b = lane; b < n; b += 64) {
        float v = 0.f;
        if (a != b) {
            if (!clamped && (a == 0 || b == 0)) v = (float)(w_label[(a == 0 ? b : a) - 1] / beta_eff);
            else if (W_hh) {
                const int i = min(a, b) - o, j = max(a, b) - o;
                v = (float)(W_hh[(long)i * H + j] / beta_eff);
            }
        }
        J[(long)a * n + b] = v;
        mag[b] = __builtin_fabsf(v);
    }
    __syncthreads();
    if (lane != 0) return;
    row_stats(mag, n, rowabs + a, rowmin + a);
}

// sample_Q's host side on the device (src/qubo/sampler.py:26-33 hands over ONE n x n QUBO per call): from the matrix as the
// caller holds it -- float64 or float32, upper-triangular by protocol, any square matrix by construction -- the sampler's
// operands in one launch:  lin_i = fp32(Q_ii);  J_ab = fp32(Q_ab + Q_ba) for a != b (the sum in the matrix's own precision, like
// numpy's (off + off.T).astype(float32));  rowabs / rowmin as k_qbm_couplings delivers them (ascending column order, float64);
// any_off[0] = 1 if some off-diagonal entry is non-zero (the reference short-circuits a QUBO without quadratic terms).
// (round 4: in 64 x 64 tiles through LDS -- a wavefront per row read the column Q[b, a] one cache line per element: 50 us at
// n = 717 -- and the row statistics from the finished J in a second launch)
template <class TQ>
__global__ __launch_bounds__(256) void k_qubo_from_Q(const TQ* __restrict__ Q, int n, float* __restrict__ lin, float* __restrict__ J,
                                                     int* __restrict__ any_off)
{
    __shared__ TQ tA[64][65], tB[64][65];   // Q[a0 + r, b0 + c] and Q[b0 + r, a0 + c]
    const int a0 = 64 * (int)blockIdx.y, b0 = 64 * (int)blockIdx.x;
    const int c = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int r = w; r < 64; r += 4) {
        tA[r][c] = (a0 + r < n && b0 + c < n) ? Q[(long)(a0 + r) * n + b0 + c] : (TQ)0;
        tB[r][c] = (b0 + r < n && a0 + c < n) ? Q[(long)(b0 + r) * n + a0 + c] : (TQ)0;
    }
    __syncthreads();
    bool nz = false;
    for (int r = w; r < 64; r += 4) {
        const int a = a0 + r, b = b0 + c;
        if (a >= n || b >= n) continue;
        const TQ qab = tA[r][c];
        float v = 0.f;
        if (a != b) {
            v = (float)(qab + tB[c][r]);
            nz = nz || (qab != (TQ)0);        // numpy's (Q - diag(diag(Q))).any()
        } else {
            lin[a] = (float)qab;
        }
        J[(long)a * n + b] = v;
    }
    if (__ballot(nz) != 0ull && (threadIdx.x & 63) == 0) atomicOr(any_off, 1);
}

// rowabs / rowmin of a finished J (one wavefront per row)
__global__ __launch_bounds__(64) void k_qubo_row_stats(const float* __restrict__ J, int n, double* __restrict__ rowabs, double* __restrict__ rowmin)
{
    __shared__ __attribute__((aligned(16))) float mag[QBM_MAX_QUBO_VARS];
    const int a = blockIdx.x, lane = threadIdx.x;
    for (int b = lane; b < n; b += 64) mag[b] = __builtin_fabsf(J[(long)a * n + b]);
    __syncthreads();
    if (lane != 0) return;
    row_stats(mag, n, rowabs + a, rowmin + a);
}

// uint8 {0,1} states -> float32 (sample_Q returns float32[num_reads, n], src/qubo/sampler.py:33)
__global__ __launch_bounds__(256) void k_u8_to_f32(const uint8_t* __restrict__ x, float* __restrict__ y, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = (float)x[i];
}

// Annealing range of every image's QUBO (what simulated annealers derive from the coefficients; sampler.auto_beta_range is
// the host statement of the same rule):  hot = ln 2 / max_i (|lin_i| + sum_j |J_ij|)  (half of the largest single flip is
// accepted),  cold = ln 100 / (smallest non-zero |coefficient|),  cold clamped to [10 hot, 1e4 hot];  (0.1, 1) if the
// QUBO has no usable coefficient.  float64 (max / min are exact: any reduction order gives the same bits), rounded to
// fp32 for the sampler.  One wavefront per image.
__global__ __launch_bounds__(64) void k_qbm_beta(const float* __restrict__ lin, const double* __restrict__ rowabs,
                                                 const double* __restrict__ rowmin, int B, int n, float* __restrict__ beta)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    double big = 0.0, mn = 0.0;
    for (int i = lane; i < n; i += 64) {
        const double al = fabs((double)lin[(long)b * n + i]);
        big = fmax(big, al + rowabs[i]);
        if (al > 0.0 && (mn == 0.0 || al < mn)) mn = al;
        const double rm = rowmin[i];
        if (rm > 0.0 && (mn == 0.0 || rm < mn)) mn = rm;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const double ob = __shfl_xor(big, off, 64), om = __shfl_xor(mn, off, 64);
        big = fmax(big, ob);
        if (om > 0.0 && (mn == 0.0 || om < mn)) mn = om;
    }
    if (lane != 0) return;
    double hot = 0.1, cold = 1.0;
    if (mn > 0.0 && big > 0.0 && big < __builtin_inf()) {
        hot = 0.6931471805599453 / big;
        cold = 4.605170185988092 / mn;
        cold = fmin(fmax(cold, 10.0 * hot), 1e4 * hot);
    }
    beta[2 * b] = (float)hot;
    beta[2 * b + 1] = (float)cold;
}

// Both phases' statistics (get_average_configuration_batch, :754-848) and their difference in one pass:
// diff = stat(clamped means) - stat(unclamped means), each accumulated sequentially over the images like k_batch_stats
// (same element order: W_vh | W_vo | b_h | b_o).
// The weight block -- (1 + D) x H elements, each a sum over the B images of x_ik * mean_ij -- in tiles of kSdKB rows x 256
// columns per workgroup (round 4): a thread keeps kSdKB pairs of sums in registers and reads its column's two means ONCE per
// image; the pixels of the tile's rows pass through LDS.  One thread per element, as until then, read both means again for
// every one of the 785 rows: 235 MB through the L2 for 150 KB of means, 27-35 us of a launch that computes for 2.  Same terms,
// same order (images ascending, product then sum).  Workgroups behind the tiles take the remaining D + H + 1 (+ 1) elements.
constexpr int kSdKB = 4, kSdIm = 128;
__global__ __launch_bounds__(256) void k_qbm_stats_diff(const double* __restrict__ X, const int* __restrict__ labels,
                                                        const double* __restrict__ mean_c /* [B,H] */,
                                                        const double* __restrict__ mean_u /* [B,1+H] */, int B, int D,
                                                        int H, double* __restrict__ diff, int with_nll, int n_tiles, int n_jb)
{
    const long nW = (long)(1 + D) * H;
    const int nu = H + 1;
    if ((int)blockIdx.x < n_tiles) {
        __shared__ double xs[kSdIm][kSdKB];
        const int tid = threadIdx.x;
        const int jb = (int)blockIdx.x % n_jb, kb = (int)blockIdx.x / n_jb;
        const int j = 256 * jb + tid, k0 = kSdKB * kb;     // rows k0 .. k0 + kSdKB - 1 (row D: the label row, clamped phase only)
        double c[kSdKB], u[kSdKB];
#pragma unroll
        for (int q = 0; q < kSdKB; ++q) c[q] = u[q] = 0.0;
        for (int i0 = 0; i0 < B; i0 += kSdIm) {
            const int nb = min(kSdIm, B - i0);
            __syncthreads();
            for (int e = tid; e < nb * kSdKB; e += 256) {
                const int ii = e / kSdKB, q = e % kSdKB, k = k0 + q;
                xs[ii][q] = k < D ? X[(long)(i0 + ii) * D + k] : (k == D ? (double)labels[i0 + ii] : 0.0);
            }
            __syncthreads();
            if (j < H) {
                int ii = 0;
                for (; ii + 4 <= nb; ii += 4) {           // (the means of 4 images requested together)
                    double mc[4], mu[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        mc[r] = mean_c[(long)(i0 + ii + r) * H + j];
                        mu[r] = mean_u[(long)(i0 + ii + r) * nu + 1 + j];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int q = 0; q < kSdKB; ++q) {
                            const double x = xs[ii + r][q];
                            c[q] += x * mc[r];
                            if (k0 + q < D) u[q] += x * mu[r];
                        }
                }
                for (; ii < nb; ++ii) {
                    const double mc = mean_c[(long)(i0 + ii) * H + j], mu = mean_u[(long)(i0 + ii) * nu + 1 + j];
#pragma unroll
                    for (int q = 0; q < kSdKB; ++q) {
                        const double x = xs[ii][q];
                        c[q] += x * mc;
                        if (k0 + q < D) u[q] += x * mu;
                    }
                }
            }
        }
        if (j < H) {
#pragma unroll
            for (int q = 0; q < kSdKB; ++q)
                if (k0 + q <= D) diff[(long)(k0 + q) * H + j] = c[q] - u[q];
        }
        return;
    }
    const long idx = nW + (long)((int)blockIdx.x - n_tiles) * blockDim.x + threadIdx.x;
    if (with_nll && idx == nW + D + H + 1) {
        // diff[theta length] = sum over the images of compute_nll's term (src/model/faster_dqbm.py:986-991):
        // -log(float32([1 - p1, p1])[y] + 1e-12), p1 = mean of the output node; float32 terms, float64 sum in image order
        // (a monitor, compared with a tolerance: a dozen small torch launches per minibatch until round 4)
        double sum = 0.0;
        for (int i = 0; i < B; ++i) {
            const double p1 = mean_u[(long)i * (H + 1)];
            const float p = (float)(labels[i] ? p1 : 1.0 - p1);
            sum += (double)(-logf(p + 1e-12f));
        }
        diff[idx] = sum;
        return;
    }
    if (idx >= nW + D + H + 1) return;
    double c = 0.0, u = 0.0;
    if (idx < nW + D) {
        const int k = (int)(idx - nW);
        for (int i = 0; i < B; ++i) {
            const double x = X[(long)i * D + k];
            c += x * (double)labels[i];
            u += x * mean_u[(long)i * nu];
        }
    } else if (idx < nW + D + H) {
        const int j = (int)(idx - nW - D);
        for (int i = 0; i < B; ++i) {
            c += mean_c[(long)i * H + j];
            u += mean_u[(long)i * nu + 1 + j];
        }
    } else {
        for (int i = 0; i < B; ++i) {
            c += (double)labels[i];
            u += mean_u[(long)i * nu];
        }
    }
    diff[idx] = c - u;
}

// train_for_one_iteration's update (:1023-1059): theta -= lr * (diff / B), element by element, in this order of operations
__global__ __launch_bounds__(256) void k_qbm_apply(double* __restrict__ theta, const double* __restrict__ diff, long n,
                                                   double lr, double batch)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) theta[i] = theta[i] - lr * (diff[i] / batch);
}

template <int NF, bool XS = false>
int launch_gibbs(const float* lin, const float* J, int B, int R, int n, int sweeps, float beta0, float beta1,
                 const float* beta_dev, uint64_t seed, uint64_t offset, uint32_t chain0, uint8_t* states, hipStream_t st)
{
    const int nch = B * R;
    const int nt = XS ? n - 1 : n;
    const size_t tri_bytes = (size_t)nt * (nt + 1) / 2 * sizeof(float);
    if constexpr (NF <= 5) if (tri_bytes <= 150 * 1024 && nch >= 64) {
        // the triangle of J fits the LDS of a CU (n <= 276): one persistent workgroup per CU, rows of J read from LDS
        if (tri_bytes + 256 > 65536) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_qubo_gibbs_lds<NF, XS>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)tri_bytes + 256);
            if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_qubo_gibbs_lds)");
        }
        // one workgroup per CU (256), as many wavefronts as even out the rounds: every wave walks `rounds` chains
        const int wgs = nch < 256 * 4 ? (nch + 3) / 4 : 256;
        const int rounds = (nch + wgs * kGibbsLdsWaves - 1) / (wgs * kGibbsLdsWaves);
        int waves = (nch + rounds * wgs - 1) / (rounds * wgs);
        waves = waves < 1 ? 1 : waves > kGibbsLdsWaves ? kGibbsLdsWaves : waves;
        hipLaunchKernelGGL((k_qubo_gibbs_lds<NF, XS>), dim3(wgs), dim3(64 * waves), tri_bytes + 256, st, lin, J, nch, R, n, sweeps,
                           beta0, beta1, beta_dev, seed, offset, chain0, states);
        QBM_CHECK_LAUNCH("k_qubo_gibbs_lds");
        return QBM_OK;
    }
    hipLaunchKernelGGL((k_qubo_gibbs<NF, XS>), dim3((nch + 3) / 4), dim3(256), 0, st, lin, J, nch, R, n, sweeps, beta0,
                       beta1, beta_dev, seed, offset, chain0, states);
    QBM_CHECK_LAUNCH("k_qubo_gibbs");
    return QBM_OK;
}

// workspace layout of the sparse sampler: [CsrHead | cnt int[n] | rowptr int[n+1] | (pad to 8) | entries uint2[cap]]
struct CsrLayout {
    size_t cnt, rowptr, ent, total;
};
inline CsrLayout csr_layout(int n, long cap)
{
    CsrLayout L;
    L.cnt = sizeof(CsrHead);
    L.rowptr = L.cnt + sizeof(int) * (size_t)n;
    L.ent = (L.rowptr + sizeof(int) * (size_t)(n + 1) + 7) & ~(size_t)7;
    L.total = L.ent + sizeof(uint2) * (size_t)cap;
    return L;
}
// wavefronts per workgroup and the entries the LDS then holds
inline int csr_waves(int nch) { const int w = (nch + 255) / 256; return w < 1 ? 1 : w > kCsrMaxWaves ? kCsrMaxWaves : w; }
inline int csr_nf(int n) { const int NF = (n + 63) / 64; return NF <= 6 ? 6 : NF <= 8 ? 8 : NF <= 10 ? 10 : NF <= 12 ? 12 : 16; }
// (one chain per workgroup: + two threshold buffers for the helper wavefront, k_qubo_gibbs_csr<NF, true>)
inline long csr_lds_cap(int n, int waves) { return ((long)kCsrLdsBytes - 256 - (long)(waves == 1 ? 3 : waves) * 256 * csr_nf(n)) / 8; }

template <int NF>
int launch_gibbs_csr(const float* lin, const char* ws, int cap, int B, int R, int n, int sweeps, float beta0, float beta1,
                     const float* beta_dev, uint64_t seed, uint64_t offset, uint32_t chain0, uint8_t* states, hipStream_t st)
{
    const int nch = B * R;
    const int waves = csr_waves(nch);
    const bool help = waves == 1;                        // one chain per workgroup: a second wavefront prepares its thresholds
    const long lds_cap = csr_lds_cap(n, waves);
    const int ent_cap = (int)(cap < lds_cap ? cap : lds_cap);
    const size_t lds = (size_t)ent_cap * 8 + (size_t)(help ? 3 : waves) * 256 * NF;
    const CsrLayout L = csr_layout(n, cap);
    const void* fn = help ? reinterpret_cast<const void*>(k_qubo_gibbs_csr<NF, true>) : reinterpret_cast<const void*>(k_qubo_gibbs_csr<NF, false>);
    if (lds > 65536) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_qubo_gibbs_csr)");
    }
    int wgs = (nch + waves - 1) / waves;
    if (wgs > 256) wgs = 256;
    if (help)
        hipLaunchKernelGGL((k_qubo_gibbs_csr<NF, true>), dim3(wgs), dim3(128), lds, st, lin, (const CsrHead*)ws, (const int*)(ws + L.rowptr),
                           (const uint2*)(ws + L.ent), ent_cap, nch, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states);
    else
        hipLaunchKernelGGL((k_qubo_gibbs_csr<NF, false>), dim3(wgs), dim3(64 * waves), lds, st, lin, (const CsrHead*)ws,
                           (const int*)(ws + L.rowptr), (const uint2*)(ws + L.ent), ent_cap, nch, R, n, sweeps, beta0, beta1, beta_dev,
                           seed, offset, chain0, states);
    QBM_CHECK_LAUNCH("k_qubo_gibbs_csr");
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
    if (!(beta0 > 0.f) || !(beta1 > 0.f)) return fail(QBM_ERR_ARG, "qubo_gibbs: beta must be positive");
    return qbm_qubo_gibbs_dev(lin, J, B, R, n, sweeps, nullptr, beta0, beta1, seed, offset, chain0, states, mean, stream);
}

int qbm_qubo_gibbs_dev(const float* lin, const float* J, int B, int R, int n, int sweeps, const float* beta_dev,
                       float beta0, float beta1, uint64_t seed, uint64_t offset, uint32_t chain0, uint8_t* states,
                       double* mean, void* stream)
{
    if (B < 0 || R < 0) return fail(QBM_ERR_ARG, "qubo_gibbs: negative batch/reads");
    if (B == 0 || R == 0) return QBM_OK;
    if (n <= 0 || n > QBM_MAX_QUBO_VARS) return fail(QBM_ERR_ARG, "qubo_gibbs: n out of range (1..QBM_MAX_QUBO_VARS)");
    if (sweeps < 0) return fail(QBM_ERR_ARG, "qubo_gibbs: sweeps < 0");
    if (!beta_dev && (!(beta0 > 0.f) || !(beta1 > 0.f))) return fail(QBM_ERR_ARG, "qubo_gibbs: beta must be positive");
    if (!lin || !J || !states) return fail(QBM_ERR_ARG, "qubo_gibbs: null buffer");
    hipStream_t st = (hipStream_t)stream;
    const int NF = (n + 63) / 64;
    int rc;
    // n = 64 k + 1 (the unclamped QUBO of a Disc_QBM with 64 k hidden nodes): k whole tiles + the last spin as a scalar
    // (gibbs_chain's XS form) instead of a (k+1)-th tile with one spin in it
#ifndef QBM_DIAG_NO_XS   // diagnostic builds (A/B timing): the extra spin as a tile of its own, as in rounds 1-3
    const bool xs = n % 64 == 1 && n > 64 && n <= 321;
#else
    const bool xs = false;
#endif
    if (xs) {
        switch (n / 64) {
        case 1: rc = launch_gibbs<1, true>(lin, J, B, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, st); break;
        case 2: rc = launch_gibbs<2, true>(lin, J, B, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, st); break;
        case 3: rc = launch_gibbs<3, true>(lin, J, B, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, st); break;
        case 4: rc = launch_gibbs<4, true>(lin, J, B, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, st); break;
        default: rc = launch_gibbs<5, true>(lin, J, B, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, st); break;
        }
    } else
    switch (NF) {
    case 1: rc = launch_gibbs<1>(lin, J, B, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, st); break;
    case 2: rc = launch_gibbs<2>(lin, J, B, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, st); break;
    case 3: rc = launch_gibbs<3>(lin, J, B, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, st); break;
    case 4: rc = launch_gibbs<4>(lin, J, B, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, st); break;
    case 5: rc = launch_gibbs<5>(lin, J, B, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, st); break;
    case 6: rc = launch_gibbs<6>(lin, J, B, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, st); break;
    case 7: case 8: rc = launch_gibbs<8>(lin, J, B, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, st); break;
    case 9: case 10: rc = launch_gibbs<10>(lin, J, B, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, st); break;
    case 11: case 12: rc = launch_gibbs<12>(lin, J, B, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, st); break;   // conv-QBM QUBOs (n ~ 700)
    default: rc = launch_gibbs<16>(lin, J, B, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, st); break;
    }
    if (rc) return rc;
    if (mean) {
        hipLaunchKernelGGL(k_read_mean, dim3((B * n + 255) / 256), dim3(256), 0, st, states, mean, B, R, n, (const int*)nullptr);
        QBM_CHECK_LAUNCH("k_read_mean");
    }
    return QBM_OK;
}

long qbm_qubo_csr_max_entries(int n, int chains)
{
    if (n <= 0 || n > QBM_MAX_QUBO_VARS || chains <= 0) return 0;
    return csr_lds_cap(n, csr_waves(chains));
}

size_t qbm_qubo_csr_workspace_bytes(int n, long cap)
{
    if (n <= 0 || n > QBM_MAX_QUBO_VARS || cap < 0) return 0;
    return csr_layout(n, cap).total;
}

int qbm_qubo_csr_build(const float* J, int n, long cap, void* workspace, void* stream)
{
    if (n <= 0 || n > QBM_MAX_QUBO_VARS) return fail(QBM_ERR_ARG, "qubo_csr_build: n out of range (1..QBM_MAX_QUBO_VARS)");
    if (cap < 0 || cap > (1l << 24)) return fail(QBM_ERR_ARG, "qubo_csr_build: capacity out of range");
    if (!J || !workspace) return fail(QBM_ERR_ARG, "qubo_csr_build: null buffer");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const CsrLayout L = csr_layout(n, cap);
    hipLaunchKernelGGL(k_csr_count, dim3(n), dim3(64), 0, st, J, n, (int*)(ws + L.cnt));
    QBM_CHECK_LAUNCH("k_csr_count");
    hipLaunchKernelGGL(k_csr_fill, dim3(n), dim3(64), 0, st, J, n, (const int*)(ws + L.cnt), (CsrHead*)ws, (int*)(ws + L.rowptr),
                       (uint2*)(ws + L.ent), (int)cap);
    QBM_CHECK_LAUNCH("k_csr_fill");
    return QBM_OK;
}

int qbm_qubo_gibbs_csr(const float* lin, const void* workspace, long cap, int B, int R, int n, int sweeps, const float* beta_dev,
                       float beta0, float beta1, uint64_t seed, uint64_t offset, uint32_t chain0, uint8_t* states, double* mean,
                       void* stream)
{
    if (B < 0 || R < 0) return fail(QBM_ERR_ARG, "qubo_gibbs_csr: negative batch/reads");
    if (B == 0 || R == 0) return QBM_OK;
    if (n <= 0 || n > QBM_MAX_QUBO_VARS) return fail(QBM_ERR_ARG, "qubo_gibbs_csr: n out of range (1..QBM_MAX_QUBO_VARS)");
    if (sweeps < 0) return fail(QBM_ERR_ARG, "qubo_gibbs_csr: sweeps < 0");
    if (!beta_dev && (!(beta0 > 0.f) || !(beta1 > 0.f))) return fail(QBM_ERR_ARG, "qubo_gibbs_csr: beta must be positive");
    if (!lin || !workspace || !states) return fail(QBM_ERR_ARG, "qubo_gibbs_csr: null buffer");
    if (cap < 0 || cap > (1l << 24)) return fail(QBM_ERR_ARG, "qubo_gibbs_csr: capacity out of range");
    if (cap > csr_lds_cap(n, csr_waves(B * R)))
        return fail(QBM_ERR_ARG, "qubo_gibbs_csr: capacity beyond qbm_qubo_csr_max_entries(n, B * R) -- the entries must fit the LDS");
    hipStream_t st = (hipStream_t)stream;
    const char* ws = (const char*)workspace;
    int rc;
    switch (csr_nf(n)) {
    case 6: rc = launch_gibbs_csr<6>(lin, ws, (int)cap, B, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, st); break;
    case 8: rc = launch_gibbs_csr<8>(lin, ws, (int)cap, B, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, st); break;
    case 10: rc = launch_gibbs_csr<10>(lin, ws, (int)cap, B, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, st); break;
    case 12: rc = launch_gibbs_csr<12>(lin, ws, (int)cap, B, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, st); break;
    default: rc = launch_gibbs_csr<16>(lin, ws, (int)cap, B, R, n, sweeps, beta0, beta1, beta_dev, seed, offset, chain0, states, st); break;
    }
    if (rc) return rc;
    if (mean) {
        hipLaunchKernelGGL(k_read_mean, dim3((B * n + 255) / 256), dim3(256), 0, st, states, mean, B, R, n, (const int*)ws);   // (ws[0]: the status word)
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

int qbm_qbm_couplings(const double* w_label, const double* W_hh, int H, int clamped, double beta_eff, float* J,
                      double* rowabs, double* rowmin, void* stream)
{
    if (H <= 0) return fail(QBM_ERR_ARG, "couplings: H <= 0");
    if (!J || !rowabs || !rowmin || (!clamped && !w_label)) return fail(QBM_ERR_ARG, "couplings: null buffer");
    if (beta_eff == 0.0) return fail(QBM_ERR_ARG, "couplings: beta_eff == 0");
    const int n = clamped ? H : H + 1;
    if (n > QBM_MAX_QUBO_VARS) return fail(QBM_ERR_ARG, "couplings: more than QBM_MAX_QUBO_VARS variables");
    hipLaunchKernelGGL(k_qbm_couplings, dim3(n), dim3(64), 0, (hipStream_t)stream, w_label, W_hh, H, clamped,
                       beta_eff, J, rowabs, rowmin);
    QBM_CHECK_LAUNCH("k_qbm_couplings");
    return QBM_OK;
}

int qbm_qubo_from_Q(const void* Q, int q_is_f64, int n, float* lin, float* J, double* rowabs, double* rowmin, int* any_offdiag,
                    void* stream)
{
    if (n <= 0 || n > QBM_MAX_QUBO_VARS) return fail(QBM_ERR_ARG, "qubo_from_Q: n out of range (1..QBM_MAX_QUBO_VARS)");
    if (!Q || !lin || !J || !rowabs || !rowmin || !any_offdiag) return fail(QBM_ERR_ARG, "qubo_from_Q: null buffer");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(any_offdiag, 0, sizeof(int), st);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(any_offdiag)");
    const dim3 tiles((n + 63) / 64, (n + 63) / 64);
    if (q_is_f64)
        hipLaunchKernelGGL((k_qubo_from_Q<double>), tiles, dim3(256), 0, st, (const double*)Q, n, lin, J, any_offdiag);
    else
        hipLaunchKernelGGL((k_qubo_from_Q<float>), tiles, dim3(256), 0, st, (const float*)Q, n, lin, J, any_offdiag);
    QBM_CHECK_LAUNCH("k_qubo_from_Q");
    hipLaunchKernelGGL(k_qubo_row_stats, dim3(n), dim3(64), 0, st, (const float*)J, n, rowabs, rowmin);
    QBM_CHECK_LAUNCH("k_qubo_row_stats");
    return QBM_OK;
}

int qbm_u8_to_f32(const uint8_t* x, float* y, long n, void* stream)
{
    if (n < 0) return fail(QBM_ERR_ARG, "u8_to_f32: n < 0");
    if (n == 0) return QBM_OK;
    if (!x || !y) return fail(QBM_ERR_ARG, "u8_to_f32: null buffer");
    hipLaunchKernelGGL(k_u8_to_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    QBM_CHECK_LAUNCH("k_u8_to_f32");
    return QBM_OK;
}

int qbm_qbm_beta_range(const float* lin, const double* rowabs, const double* rowmin, int B, int n, float* beta, void* stream)
{
    if (B < 0 || n <= 0) return fail(QBM_ERR_ARG, "beta_range: bad dims");
    if (B == 0) return QBM_OK;
    if (!lin || !rowabs || !rowmin || !beta) return fail(QBM_ERR_ARG, "beta_range: null buffer");
    hipLaunchKernelGGL(k_qbm_beta, dim3(B), dim3(64), 0, (hipStream_t)stream, lin, rowabs, rowmin, B, n, beta);
    QBM_CHECK_LAUNCH("k_qbm_beta");
    return QBM_OK;
}

int qbm_qbm_stats_diff(const double* X, const int32_t* labels, const double* mean_c, const double* mean_u, int B, int D,
                       int H, double* diff, void* stream)
{
    if (B < 0) return fail(QBM_ERR_ARG, "stats_diff: B < 0");
    if (D <= 0 || H <= 0) return fail(QBM_ERR_ARG, "stats_diff: non-positive dims");
    if (!diff || (B > 0 && (!X || !labels || !mean_c || !mean_u))) return fail(QBM_ERR_ARG, "stats_diff: null buffer");
    const int n_jb = (H + 255) / 256, n_tiles = n_jb * ((D + 1 + kSdKB - 1) / kSdKB);
    const int n_tail = (D + H + 1 + 255) / 256;
    hipLaunchKernelGGL(k_qbm_stats_diff, dim3((unsigned)(n_tiles + n_tail)), dim3(256), 0, (hipStream_t)stream, X, labels,
                       mean_c, mean_u, B, D, H, diff, 0, n_tiles, n_jb);
    QBM_CHECK_LAUNCH("k_qbm_stats_diff");
    return QBM_OK;
}

int qbm_qbm_stats_diff_nll(const double* X, const int32_t* labels, const double* mean_c, const double* mean_u, int B, int D,
                           int H, double* diff, void* stream)
{
    if (B < 0) return fail(QBM_ERR_ARG, "stats_diff_nll: B < 0");
    if (D <= 0 || H <= 0) return fail(QBM_ERR_ARG, "stats_diff_nll: non-positive dims");
    if (!diff || (B > 0 && (!X || !labels || !mean_c || !mean_u))) return fail(QBM_ERR_ARG, "stats_diff_nll: null buffer");
    const int n_jb = (H + 255) / 256, n_tiles = n_jb * ((D + 1 + kSdKB - 1) / kSdKB);
    const int n_tail = (D + H + 2 + 255) / 256;
    hipLaunchKernelGGL(k_qbm_stats_diff, dim3((unsigned)(n_tiles + n_tail)), dim3(256), 0, (hipStream_t)stream, X, labels,
                       mean_c, mean_u, B, D, H, diff, 1, n_tiles, n_jb);
    QBM_CHECK_LAUNCH("k_qbm_stats_diff");
    return QBM_OK;
}

int qbm_qbm_apply_update(double* theta, const double* diff, int D, int H, double lr, double batch, void* stream)
{
    if (D <= 0 || H <= 0) return fail(QBM_ERR_ARG, "apply_update: non-positive dims");
    if (!theta || !diff) return fail(QBM_ERR_ARG, "apply_update: null buffer");
    if (!(batch > 0.0)) return fail(QBM_ERR_ARG, "apply_update: batch must be positive");
    const long total = qbm_theta_len(D, H);
    hipLaunchKernelGGL(k_qbm_apply, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, theta, diff, total,
                       lr, batch);
    QBM_CHECK_LAUNCH("k_qbm_apply");
    return QBM_OK;
}

int qbm_qbm_linear_fields_dev(const double* X, const int32_t* labels, const double* theta, int B, int D, int H,
                              double beta_eff, float* lin, void* stream)
{
    if (B < 0) return fail(QBM_ERR_ARG, "linear_fields: B < 0");
    if (B == 0) return QBM_OK;
    if (D <= 0 || H <= 0) return fail(QBM_ERR_ARG, "linear_fields: non-positive dims");
    if (!X || !theta || !lin) return fail(QBM_ERR_ARG, "linear_fields: null buffer");
    if (beta_eff == 0.0) return fail(QBM_ERR_ARG, "linear_fields: beta_eff == 0");
    const double* W_vh = theta;
    const double* W_vo = W_vh + (long)(1 + D) * H;
    const double* b_h = W_vo + D;
    const int n = labels ? H : H + 1;
    hipLaunchKernelGGL(k_linear_fields, dim3((n + 63) / 64, (B + kLfIm - 1) / kLfIm), dim3(64 * kLfIm), 0, (hipStream_t)stream, X, labels,
                       W_vh, W_vo, b_h, 0.0, b_h + H, B, D, H, beta_eff, lin);
    QBM_CHECK_LAUNCH("k_linear_fields");
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
    hipLaunchKernelGGL(k_linear_fields, dim3((n + 63) / 64, (B + kLfIm - 1) / kLfIm), dim3(64 * kLfIm), 0, (hipStream_t)stream, X, labels,
                       W_vh, W_vo, b_h, b_o, (const double*)nullptr, B, D, H, beta_eff, lin);
    QBM_CHECK_LAUNCH("k_linear_fields");
    return QBM_OK;
}

}  // extern "C"
