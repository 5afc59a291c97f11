// common.h -- error plumbing and launch helpers shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/muzero_amd.h"

void mz_set_error(const char *fmt, ...);

#define MZ_REQUIRE(cond, ...)                  \
    do {                                       \
        if (!(cond)) {                         \
            mz_set_error(__VA_ARGS__);         \
            return MZ_E_ARG;                   \
        }                                      \
    } while (0)

#define MZ_HIP(call)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            mz_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return MZ_E_HIP;                                                             \
        }                                                                                \
    } while (0)

#define MZ_LAUNCH_CHECK(name)                                                            \
    do {                                                                                 \
        hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess) {                                                          \
            mz_set_error("launch of %s failed: %s", name, hipGetErrorString(e_));        \
            return MZ_E_HIP;                                                             \
        }                                                                                \
    } while (0)

static inline unsigned mz_blocks(int64_t n, int threads) { return (unsigned)((n + threads - 1) / threads); }
