// qbm_host.hpp -- host-side helpers shared by the translation units of libqbm_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/qbm_hip.h"

namespace qbm {
int fail(int code, const char* msg);             // records msg for qbm_last_error(), returns code
int hip_fail(hipError_t e, const char* where);   // same, with hipGetErrorString; returns QBM_ERR_HIP
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }
inline bool aligned2(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 1u) == 0; }
}  // namespace qbm

#define QBM_CHECK_LAUNCH(where)                                      \
    do {                                                             \
        hipError_t e__ = hipGetLastError();                          \
        if (e__ != hipSuccess) return ::qbm::hip_fail(e__, where);   \
    } while (0)
