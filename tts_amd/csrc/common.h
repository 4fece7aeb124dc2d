// Shared host-side helpers for the tts_amd C ABI (gfx950 only; no CUDA/compat paths).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/tts_amd.h"

namespace ttsamd {

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

#define TTSAMD_CHECK_ARG(cond, ...)          \
    do {                                     \
        if (!(cond)) {                       \
            ::ttsamd::set_error(__VA_ARGS__); \
            return TTSAMD_ERR_INVALID;       \
        }                                    \
    } while (0)

#define TTSAMD_HIP(call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            ::ttsamd::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                                __LINE__);                                                       \
            return TTSAMD_ERR_HIP;                                                               \
        }                                                                                        \
    } while (0)

#define TTSAMD_LAUNCH_CHECK() TTSAMD_HIP(hipGetLastError())

constexpr int kWave = 64;  // CDNA wavefront width

}  // namespace ttsamd
