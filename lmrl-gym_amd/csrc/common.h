// common.h — shared host-side helpers for liblmrl_amd.so (gfx950 only; no CUDA/compat paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

namespace lmrl {

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

#define LMRL_CHECK_HIP(expr)                                                                  \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ::lmrl::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return LMRL_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

#define LMRL_CHECK_LAUNCH()  LMRL_CHECK_HIP(hipGetLastError())

#define LMRL_REQUIRE(cond, msg)                                   \
    do {                                                          \
        if (!(cond)) {                                            \
            ::lmrl::set_error("%s:%d: %s", __FILE__, __LINE__, msg); \
            return LMRL_ERR_ARG;                                  \
        }                                                         \
    } while (0)

constexpr int kWave = 64;  // CDNA wavefront

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace lmrl
