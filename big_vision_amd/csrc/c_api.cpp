// libbvhip: error plumbing shared by all kernels' C entry points.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdint.h>

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

// ---- bv_ctx (include/bvhip.h "Context"): host memory only; the defaults below are what a NULL context means
static void ctx_defaults(bv_ctx* c) {
  for (int i = 0; i < BV_OPT_COUNT; ++i) c->opt[i] = 0;
  c->opt[BV_OPT_FAST_PATH] = 1;
  c->opt[BV_OPT_GEMM_SKEW_MODE] = 1;
  c->opt[BV_OPT_GEMM_ROLL] = 1;
  c->opt[BV_OPT_SGEMM_MFMA] = 1;
  c->ws = nullptr;
  c->ws_bytes = 0;
  for (auto& a : c->calls) a.store(0, std::memory_order_relaxed);
}
const bv_ctx* bv_ctx_or_default(const bv_ctx* c) {
  if (c) return c;
  static const bv_ctx* const dflt = [] { bv_ctx* d = new bv_ctx; ctx_defaults(d); return d; }();   // never written again
  return dflt;
}
extern "C" bv_ctx* bv_ctx_create(void) {
  bv_ctx* c = new bv_ctx;
  ctx_defaults(c);
  return c;
}
extern "C" void bv_ctx_destroy(bv_ctx* ctx) { delete ctx; }
extern "C" long bv_ctx_get(const bv_ctx* ctx, int opt) {
  const bv_ctx* c = bv_ctx_or_default(ctx);
  if (opt >= 0 && opt < BV_OPT_COUNT) return c->opt[opt];
  if (opt >= BV_STAT_GEMM256_CALLS && opt <= BV_STAT_GEMM256_FUSED)
    return c->calls[opt - BV_STAT_GEMM256_CALLS].load(std::memory_order_relaxed);
  bv_set_error("bv_ctx_get: unknown option %d", opt);
  return BV_ERR_INVALID_ARG;
}
extern "C" long bv_ctx_set(bv_ctx* ctx, int opt, long value) {
  if (!ctx || opt < 0 || opt >= BV_OPT_COUNT) {
    bv_set_error("bv_ctx_set: %s", ctx ? "unknown option" : "NULL context (the defaults are immutable)");
    return BV_ERR_INVALID_ARG;
  }
  const long old = ctx->opt[opt];
  if (value >= 0) {
    if (opt == BV_OPT_GEMM_RESERVE_CUS && value > 128) value = 128;
    if (opt == BV_OPT_FAST_PATH || opt == BV_OPT_SGEMM_MFMA || opt == BV_OPT_GEMM_PRE_ISSUE) value = value != 0;
    ctx->opt[opt] = value;
  }
  return old;
}
extern "C" int bv_ctx_set_workspace(bv_ctx* ctx, void* ptr, long bytes) {
  if (!ctx || bytes < 0 || ((uintptr_t)ptr & 15)) {
    bv_set_error("bv_ctx_set_workspace: NULL context, negative size or a pointer that is not 16-byte aligned");
    return BV_ERR_INVALID_ARG;
  }
  ctx->ws = ptr;
  ctx->ws_bytes = ptr ? bytes : 0;
  return BV_OK;
}
