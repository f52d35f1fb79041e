// data_kernels.hip -- device-resident dataset pipeline behind the data_loader outputs:
// the training set lives in HBM (fp32 greyscale, uint8 {0,1}, or bit-packed 98 B per 28x28 image), the epoch
// permutation is made on the device and a minibatch is one gather launch -- no host copy per step.
// Byte / integer work only: results are exact, the oracle restates them in numpy.
#include "qbm_device.hip.h"
#include "qbm_host.hpp"
#include "../../include/qbm_hip.h"

using namespace qbm;

namespace {

// reference: the {0,1} states the QBM / RBM fixtures use: x > threshold  (data_loader.binarise)
__global__ void k_binarise(const float* __restrict__ x, long n, float thr, uint8_t* __restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[i] > thr ? 1 : 0;
}

// [N,V] {0,1} -> [N, ceil(V/8)], MSB first (numpy.packbits): element e of a row is bit 7-(e&7) of byte e>>3
__global__ void k_pack_bits(const uint8_t* __restrict__ x, int N, int V, uint8_t* __restrict__ bits)
{
    const int nb = (V + 7) / 8;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)N * nb) return;
    const int i = (int)(t / nb), b = (int)(t % nb);
    const uint8_t* row = x + (long)i * V;
    unsigned v = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int e = 8 * b + k;
        v |= (e < V && row[min(e, V - 1)] != 0) ? (0x80u >> k) : 0u;
    }
    bits[t] = (uint8_t)v;
}

// sort keys of the epoch permutation: 31 bits of Philox word i (counter = (i>>2, 0, epoch)), non-negative int32;
// the permutation is the stable argsort of the keys
__global__ void k_shuffle_keys(uint64_t seed, uint64_t epoch, int N, int32_t* __restrict__ keys)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;   // group of 4 indices
    if (4 * g >= N) return;
    const Philox4 p = philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)g, 0u, (uint32_t)epoch,
                                    (uint32_t)(epoch >> 32));
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (4 * g + e < N) keys[4 * g + e] = (int32_t)(p.w[e] >> 1);
}

// minibatch gather: row r of the batch = row idx[r] of the set.  One workgroup per row.
//   SRC 0 = fp32, 1 = uint8, 2 = bit-packed;  DST 0 = fp32, 1 = uint8.  Labels ride along (int32 and one-hot).
template <int SRC, int DST>
__global__ __launch_bounds__(128) void k_gather(const void* __restrict__ src, const int32_t* __restrict__ idx, int V,
                                                void* __restrict__ dst, const int32_t* __restrict__ labels,
                                                int32_t* __restrict__ labels_out, float* __restrict__ y1hot, int C)
{
    const int r = blockIdx.x;
    const long s = idx[r];
    if (threadIdx.x == 0 && labels) {
        const int lab = labels[s];
        if (labels_out) labels_out[r] = lab;
        if (y1hot)
            for (int c = 0; c < C; ++c) y1hot[(long)r * C + c] = (c == lab) ? 1.f : 0.f;
    }
    if constexpr (SRC == 2) {
        const int nb = (V + 7) / 8;
        const uint8_t* row = reinterpret_cast<const uint8_t*>(src) + s * nb;
        for (int b = threadIdx.x; b < nb; b += blockDim.x) {
            const unsigned v = row[b];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int e = 8 * b + k;
                if (e < V) {
                    const unsigned bit = (v >> (7 - k)) & 1u;
                    if constexpr (DST == 0) reinterpret_cast<float*>(dst)[(long)r * V + e] = (float)bit;
                    else reinterpret_cast<uint8_t*>(dst)[(long)r * V + e] = (uint8_t)bit;
                }
            }
        }
    } else {
        for (int e = threadIdx.x; e < V; e += blockDim.x) {
            if constexpr (SRC == 0 && DST == 0)
                reinterpret_cast<float*>(dst)[(long)r * V + e] = reinterpret_cast<const float*>(src)[s * V + e];
            else if constexpr (SRC == 0 && DST == 1)
                reinterpret_cast<uint8_t*>(dst)[(long)r * V + e] = (uint8_t)reinterpret_cast<const float*>(src)[s * V + e];
            else if constexpr (SRC == 1 && DST == 0)
                reinterpret_cast<float*>(dst)[(long)r * V + e] = (float)reinterpret_cast<const uint8_t*>(src)[s * V + e];
            else
                reinterpret_cast<uint8_t*>(dst)[(long)r * V + e] = reinterpret_cast<const uint8_t*>(src)[s * V + e];
        }
    }
}

}  // namespace

extern "C" {

int qbm_data_binarise(const float* x, long n, float threshold, uint8_t* out, void* stream)
{
    if (n < 0) return fail(QBM_ERR_ARG, "binarise: n < 0");
    if (n == 0) return QBM_OK;
    if (!x || !out) return fail(QBM_ERR_ARG, "binarise: null buffer");
    hipLaunchKernelGGL(k_binarise, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, threshold, out);
    QBM_CHECK_LAUNCH("k_binarise");
    return QBM_OK;
}

int qbm_data_pack_bits(const uint8_t* x, int N, int V, uint8_t* bits, void* stream)
{
    if (N < 0 || V <= 0) return fail(QBM_ERR_ARG, "pack_bits: bad shape");
    if (N == 0) return QBM_OK;
    if (!x || !bits) return fail(QBM_ERR_ARG, "pack_bits: null buffer");
    const long n = (long)N * ((V + 7) / 8);
    hipLaunchKernelGGL(k_pack_bits, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, N, V, bits);
    QBM_CHECK_LAUNCH("k_pack_bits");
    return QBM_OK;
}

int qbm_data_shuffle_keys(uint64_t seed, uint64_t epoch, int N, int32_t* keys, void* stream)
{
    if (N < 0) return fail(QBM_ERR_ARG, "shuffle_keys: N < 0");
    if (N == 0) return QBM_OK;
    if (!keys) return fail(QBM_ERR_ARG, "shuffle_keys: null buffer");
    const int g = (N + 3) / 4;
    hipLaunchKernelGGL(k_shuffle_keys, dim3((g + 255) / 256), dim3(256), 0, (hipStream_t)stream, seed, epoch, N, keys);
    QBM_CHECK_LAUNCH("k_shuffle_keys");
    return QBM_OK;
}

int qbm_data_gather(const void* src, int src_kind, const int32_t* idx, int B, int V, void* dst, int dst_kind,
                    const int32_t* labels, int32_t* labels_out, float* y1hot, int C, void* stream)
{
    if (B < 0 || V <= 0) return fail(QBM_ERR_ARG, "gather: bad shape");
    if (B == 0) return QBM_OK;
    if (!src || !idx || !dst) return fail(QBM_ERR_ARG, "gather: null buffer");
    if (src_kind < QBM_F32 || src_kind > QBM_BITS) return fail(QBM_ERR_ARG, "gather: src_kind must be QBM_F32, QBM_U8 or QBM_BITS");
    if (dst_kind != QBM_F32 && dst_kind != QBM_U8) return fail(QBM_ERR_ARG, "gather: dst_kind must be QBM_F32 or QBM_U8");
    if ((labels_out || y1hot) && !labels) return fail(QBM_ERR_ARG, "gather: label outputs need the label array");
    if (y1hot && C <= 0) return fail(QBM_ERR_ARG, "gather: one-hot output needs C > 0");
    hipStream_t st = (hipStream_t)stream;
#define QBM_GATHER(S, D) \
    hipLaunchKernelGGL((k_gather<S, D>), dim3(B), dim3(128), 0, st, src, idx, V, dst, labels, labels_out, y1hot, C)
    if (src_kind == QBM_F32) { if (dst_kind == QBM_F32) QBM_GATHER(0, 0); else QBM_GATHER(0, 1); }
    else if (src_kind == QBM_U8) { if (dst_kind == QBM_F32) QBM_GATHER(1, 0); else QBM_GATHER(1, 1); }
    else { if (dst_kind == QBM_F32) QBM_GATHER(2, 0); else QBM_GATHER(2, 1); }
#undef QBM_GATHER
    QBM_CHECK_LAUNCH("k_gather");
    return QBM_OK;
}

}  // extern "C"
