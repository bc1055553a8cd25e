// Thread-local error string + version for libowlhip's C ABI (include/owl_hip.h).
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void owl_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* owl_last_error(void) { return g_err; }
extern "C" int owl_abi_version(void) { return 1; }
