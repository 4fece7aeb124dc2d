// Shared host-side helpers for the tts_amd C ABI (gfx950 only; no CUDA/compat paths).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <exception>
#include <new>

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

// Exception barrier of the C ABI (include/tts_amd.h: "never throws/aborts across the ABI").  The kernel-level entries are plain
// C-style code; the model-level handles (hifigan_model.hip, vits_model.hip, glow_model.hip) use the standard library, so every
// extern "C" entry of theirs runs its body through abi_guard: an exception becomes TTSAMD_ERR_INTERNAL + ttsamd_last_error().
template <class F>
inline int abi_guard(const char *where, F &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        set_error("%s: out of host memory", where);
    } catch (const std::exception &e) {
        set_error("%s: %s", where, e.what());
    } catch (...) {
        set_error("%s: unknown C++ exception", where);
    }
    return TTSAMD_ERR_INTERNAL;
}

extern std::atomic<unsigned long long> g_launches;       // kernel launches issued through the ABI (ttsamd_launch_count)
#define TTSAMD_LAUNCH_CHECK()                                        \
    do {                                                             \
        ::ttsamd::g_launches.fetch_add(1, std::memory_order_relaxed); \
        TTSAMD_HIP(hipGetLastError());                               \
    } while (0)

constexpr int kWave = 64;  // CDNA wavefront width

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel instantiation, device).  `done` is a per-call-site
// bit mask of devices already configured (one static per template instantiation); the attribute call is idempotent, so a
// race between two host threads costs a redundant call, never a missed one.
inline hipError_t ensure_dynamic_lds(const void *kern, int bytes, std::atomic<unsigned long long> &done)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
constexpr int kBufOob = 0x7FFFFFF0;  // buffer offset of an invalid lane: the hardware range check drops it / reads 0
// Buffer-resource helpers (raw buffer, stride 0; dword 3 = 0x00020000 as on gfx90a/gfx94x/gfx950).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, long bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float ld_buf(__amdgpu_buffer_rsrc_t r, int voffset, int soffset)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voffset, soffset, 0));
}
__device__ __forceinline__ void st_buf(__amdgpu_buffer_rsrc_t r, float v, int voffset, int soffset)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voffset, soffset, 0);
}

#endif

}  // namespace ttsamd
