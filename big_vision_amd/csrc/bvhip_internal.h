// Internal: pulls in the public C ABI so kernels and the header cannot drift, and defines the
// context object behind the opaque `bv_ctx` of include/bvhip.h.
#pragma once
#include "../../include/bvhip.h"
#include <atomic>

// One caller's options, split-K workspace and launch counters.  Entry points read it through
// bv_opt() / bv_ctx_ws(); a NULL context reads the immutable defaults (no workspace).  Nothing in the
// library keeps a mutable context of its own.
struct bv_ctx {
  long opt[BV_OPT_COUNT];
  void* ws;
  long ws_bytes;
  mutable std::atomic<long> calls[3];
};
const bv_ctx* bv_ctx_or_default(const bv_ctx* c);
inline long bv_opt(const bv_ctx* c, int o) { return bv_ctx_or_default(c)->opt[o]; }
