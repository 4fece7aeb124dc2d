#include "common.h"

#include <cstring>

namespace ttsamd {
static thread_local char g_err[512] = "";
std::atomic<unsigned long long> g_launches{0};

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ttsamd

extern "C" const char *ttsamd_last_error(void) { return ttsamd::g_err; }
extern "C" int ttsamd_abi_version(void) { return 4; }   // 4: model-level handles of VITS / Glow-TTS, TTSAMD_ERR_INTERNAL; 2: ttsamd_resblock_args carries the weight image sizes; 3: w_h2 / w1_h2 / w2_h2 (three-product arithmetic)
extern "C" const char *ttsamd_arch(void) { return "gfx950"; }
extern "C" uint64_t ttsamd_launch_count(void) { return ttsamd::g_launches.load(std::memory_order_relaxed); }

extern "C" int ttsamd_stream_create(int priority, void **stream_out)
{
    TTSAMD_CHECK_ARG(stream_out != nullptr, "stream_create: stream_out is NULL");
    int lo = 0, hi = 0;                                   // lo = least priority (largest number), hi = greatest
    TTSAMD_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    const int pr = priority < hi ? hi : (priority > lo ? lo : priority);
    hipStream_t st = nullptr;
    TTSAMD_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, pr));
    *stream_out = reinterpret_cast<void *>(st);
    return TTSAMD_OK;
}

extern "C" int ttsamd_stream_destroy(void *stream)
{
    if (stream) TTSAMD_HIP(hipStreamDestroy(ttsamd::as_stream(stream)));
    return TTSAMD_OK;
}
