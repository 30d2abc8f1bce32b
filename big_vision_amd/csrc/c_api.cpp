// libbvhip: error plumbing shared by all kernels' C entry points.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "bv_common.h"
#include "bvhip_internal.h"

static thread_local char g_err[512] = "";

void bv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int bv_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    bv_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return BV_ERR_HIP;
  }
  return BV_OK;
}

extern "C" const char* bv_last_error(void) { return g_err; }
extern "C" int bv_version(void) { return BVHIP_VERSION; }
