// api.hip -- version / error reporting of libpvcnn_hip.so (see include/pvcnn_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace pvcnn {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char *what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  set_error("%s: kernel launch failed: %s", what, hipGetErrorString(e));
  return static_cast<int>(e);
}

}  // namespace pvcnn

extern "C" int pvcnn_version(void) { return PVCNN_ABI_VERSION; }
extern "C" const char *pvcnn_last_error_string(void) { return pvcnn::g_err; }
