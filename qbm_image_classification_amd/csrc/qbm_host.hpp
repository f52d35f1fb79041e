// qbm_host.hpp -- host-side helpers shared by the translation units of libqbm_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/qbm_hip.h"

namespace qbm {
int fail(int code, const char* msg);             // records msg for qbm_last_error(), returns code
int hip_fail(hipError_t e, const char* where);   // same, with hipGetErrorString; returns QBM_ERR_HIP
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }
inline bool aligned2(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 1u) == 0; }
}  // namespace qbm

// QBM_DEBUG_SYNC=1 in the environment: every launch is followed by a device synchronisation and a line on stderr, so that a
// faulting kernel is the last one named (diagnosis only: never set inside a stream capture)
inline bool debug_sync()
{
    static const bool on = [] { const char* e = getenv("QBM_DEBUG_SYNC"); return e && e[0] == '1'; }();
    return on;
}
#define QBM_CHECK_LAUNCH(where)                                      \
    do {                                                             \
        hipError_t e__ = hipGetLastError();                          \
        if (e__ != hipSuccess) return ::qbm::hip_fail(e__, where);   \
        if (::debug_sync()) {                                        \
            fprintf(stderr, "[qbm] launched %s\n", where);           \
            fflush(stderr);                                          \
            e__ = hipDeviceSynchronize();                            \
            fprintf(stderr, "[qbm] %s done (%d)\n", where, (int)e__); \
            fflush(stderr);                                          \
        }                                                            \
    } while (0)
