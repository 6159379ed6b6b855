// libeeseg: error reporting + version.
#include "eeseg_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void eeseg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* eeseg_last_error(void) { return g_err; }
extern "C" int eeseg_version(void) { return 106; }   // = _lib.ABI_VERSION; bump with every signature / struct change
