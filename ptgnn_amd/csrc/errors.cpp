// Thread-local error message + version for libptgnn_amd.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/ptgnn_amd.h"

namespace ptgnn_amd {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace ptgnn_amd

extern "C" int ptgnn_amd_version(void) { return PTGNN_AMD_VERSION; }
extern "C" const char *ptgnn_amd_last_error(void) { return ptgnn_amd::g_err; }
