// Error plumbing of the C-ABI: thread-local message, int return codes.
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void dy_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* dy_last_error(void) { return g_err; }
extern "C" int dy_version(void) { return 1; }
