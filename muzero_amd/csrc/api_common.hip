// api_common.hip -- error string, version.
#include <stdarg.h>
#include "common.h"

static thread_local char g_err[512] = "";

void mz_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *mz_last_error(void) { return g_err; }
extern "C" int mz_version(void) { return 1; }
