// Library-level entry points of include/ngan.h: version string and the per-thread error message.
#include <cstdarg>
#include <cstdio>
#include "../../include/ngan.h"

namespace ngan {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ngan

extern "C" const char* ngan_version(void) { return "ngan-hip 0.1.0 (gfx950)"; }
extern "C" const char* ngan_last_error(void) { return ngan::g_err; }
