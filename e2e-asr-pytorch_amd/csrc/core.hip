// Error plumbing + identification entry points of libasr_hip.so.
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void asr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* asr_last_error(void) { return g_err; }
extern "C" int asr_version(void) { return 100; }
extern "C" const char* asr_device_arch(void) { return "gfx950"; }
