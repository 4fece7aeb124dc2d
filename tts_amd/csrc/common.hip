#include "common.h"

#include <cstring>

namespace ttsamd {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ttsamd

extern "C" const char *ttsamd_last_error(void) { return ttsamd::g_err; }
extern "C" int ttsamd_abi_version(void) { return 1; }
extern "C" const char *ttsamd_arch(void) { return "gfx950"; }
