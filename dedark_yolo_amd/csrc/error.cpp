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

// Name of the GPU kernel the last C-ABI call of this thread launched ("" when the entry does not report one).  bench.py keys its
// per-kernel roofline on it, so that the line names ONE kernel symbol (as rocprofv3 prints it), not a C-ABI entry.
static thread_local const char* g_kernel = "";
void dy_note_kernel(const char* name) { g_kernel = name; }
extern "C" const char* dy_last_kernel(void) { return g_kernel; }
extern "C" void dy_clear_last_kernel(void) { g_kernel = ""; }
extern "C" int dy_version(void) { return 2; }
