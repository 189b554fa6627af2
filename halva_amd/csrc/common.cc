// Host-side error reporting shared by every entry point of libhalva_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/halva_hip.h"

static thread_local char g_err[512] = "";

void halva_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* halva_last_error(void) { return g_err; }
extern "C" int halva_abi_version(void) { return HALVA_ABI_VERSION; }
