// Probes own ONE context: libbvhip has no process-global option setters any more (include/bvhip.h "Context").
// Include AFTER the kernel source the probe pulls in (it needs bvhip_internal.h); link big_vision_amd/csrc/c_api.cpp.
#pragma once
static bv_ctx* probe_ctx() {
  static bv_ctx* c = bv_ctx_create();
  return c;
}
static inline int bv_gemm_roll(int mask) { return (int)bv_ctx_set(probe_ctx(), BV_OPT_GEMM_ROLL, mask); }
static inline int bv_attn_tune(int cfg) { return (int)bv_ctx_set(probe_ctx(), BV_OPT_ATTN_CFG, cfg); }
