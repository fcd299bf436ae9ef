// library-wide state of the C ABI: thread-local error string, version.
#include "focr_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

extern "C" void focr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* focr_last_error(void) { return g_err; }
extern "C" int focr_version(void) { return 100; }
