// Error text + version for the C ABI (include/mmr.h).
#include <stdarg.h>
#include <stdio.h>

#include "mmr_common.h"

namespace mmr {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
}  // namespace mmr

extern "C" const char *mmr_last_error(void) { return mmr::g_err; }
extern "C" int mmr_version(void) { return 1; }
