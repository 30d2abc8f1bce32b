// BigVision Adafactor (big_vision/optax.py:187-216: optax.scale_by_factored_rms(factored=True,
// decay_rate 0.8 as 1 - t^-0.8 capped at 0.999, min_dim_size_to_factor 32, eps 1e-30) ->
// [clip_by_block_rms(clipping_threshold): per-leaf, bv_adafactor_step only] -> optax.ema(0.9, debias=False, bf16 accumulator)) fused with
// the rest of the bv_optax chain (optax.py:100-149: global-norm clip, lr, lr_mults, decoupled weight
// decay, schedule, sign, apply) - per parameter LEAF, because the second-moment statistics are
// factored along the two largest axes of each leaf.
//
// A leaf is addressed as a strided 4-D view [B1][B2][R][C] of the flat fp32 buffers (params, grads,
// momentum share one layout): C = its largest axis ("d0" of optax), R = the second largest ("d1"),
// B1/B2 = the remaining axes.  The fused q/k/v tensors of libbvhip ([D][3][H][64]) present their
// Flax leaves (query/kernel [D][H][64], ...) as such views.  State per factored leaf:
// v_row[B][R] = EMA of mean_C(g^2 + eps), v_col[B][C] = EMA of mean_R(g^2 + eps), rcm[B] =
// mean_R(v_row); unfactored leaves keep v per element.
//   u = g / sqrt(v_row / rcm) / sqrt(v_col)      (factored)      u = g / sqrt(v)   (unfactored)
//   m = 0.9 m + 0.1 u (stored bf16, used in fp32) ; p += -sched (lr_eff m + wd p)
// All HBM-bound and tiny next to the step (3 passes over the gradients); one launch group per leaf.
#include "bv_common.h"
#include "bvhip_internal.h"

namespace {

struct AfView {
  long off;              // element offset of the view origin in params / grads / momentum / shadow
  int B1, B2, R, C;
  long sB1, sB2, sR, sC;
};

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float clip_factor(const double* gsq, float clip_norm) {
  if (clip_norm <= 0.f) return 1.f;
  const float norm = (float)sqrt(gsq[0]);
  return norm > clip_norm ? clip_norm / norm : 1.f;     // optax.clip_by_global_norm
}

// v_row[b][r] = d v_row + (1 - d) mean_c((cf g)^2 + eps): one workgroup per (b, r)
__global__ __launch_bounds__(256) void af_rows_kernel(const float* __restrict__ g, AfView v,
                                                      float* __restrict__ v_row, const double* gsq,
                                                      float clip_norm, float decay, float eps) {
  __shared__ float sh[4];
  const int br = blockIdx.x;
  const int r = br % v.R, b = br / v.R;
  const int b2 = b % v.B2, b1 = b / v.B2;
  const float cf = clip_factor(gsq, clip_norm);
  const float* base = g + v.off + b1 * v.sB1 + b2 * v.sB2 + r * v.sR;
  float acc = 0.f;
  for (int c = threadIdx.x; c < v.C; c += 256) {
    const float x = base[c * v.sC] * cf;
    acc += x * x + eps;
  }
  const float s = block_sum_256(acc, sh);
  if (threadIdx.x == 0) v_row[br] = decay * v_row[br] + (1.f - decay) * (s / (float)v.C);
}

// v_col[b][c] = d v_col + (1 - d) mean_r(...): one thread per (b, c)
__global__ __launch_bounds__(256) void af_cols_kernel(const float* __restrict__ g, AfView v,
                                                      float* __restrict__ v_col, const double* gsq,
                                                      float clip_norm, float decay, float eps) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long nbc = (long)v.B1 * v.B2 * v.C;
  if (i >= nbc) return;
  const int c = (int)(i % v.C), b = (int)(i / v.C);
  const int b2 = b % v.B2, b1 = b / v.B2;
  const float cf = clip_factor(gsq, clip_norm);
  const float* base = g + v.off + b1 * v.sB1 + b2 * v.sB2 + c * v.sC;
  float acc = 0.f;
  for (int r = 0; r < v.R; ++r) {
    const float x = base[r * v.sR] * cf;
    acc += x * x + eps;
  }
  v_col[i] = decay * v_col[i] + (1.f - decay) * (acc / (float)v.R);
}

// rcm[b] = mean_r v_row[b][r]
__global__ __launch_bounds__(256) void af_rcm_kernel(const float* __restrict__ v_row, float* __restrict__ rcm, int R) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) acc += v_row[(long)blockIdx.x * R + r];
  const float s = block_sum_256(acc, sh);
  if (threadIdx.x == 0) rcm[blockIdx.x] = s / (float)R;
}

// Elementwise update of one leaf.  FACT: factored statistics (v_row, v_col, rcm) else per-element v.
// r_fast: walk the view with r as the fastest index (when sR < sC), else c fastest.
template <bool FACT, bool MOM_BF16>
__global__ __launch_bounds__(256) void af_update_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        void* __restrict__ mom, bf16* __restrict__ shadow,
                                                        AfView v, const float* __restrict__ v_row,
                                                        const float* __restrict__ v_col,
                                                        const float* __restrict__ rcm, float* __restrict__ vfull,
                                                        const double* gsq, float clip_norm, float decay,
                                                        float eps, float momentum, float lr_eff, float wd,
                                                        float sched, int r_fast, double* __restrict__ stats) {
  __shared__ float sh[4];
  const long total = (long)v.B1 * v.B2 * v.R * v.C;
  const float cf = clip_factor(gsq, clip_norm);
  float sp = 0.f, su = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int r, c;
    long b;
    if (r_fast) { r = (int)(i % v.R); const long t = i / v.R; c = (int)(t % v.C); b = t / v.C; }
    else { c = (int)(i % v.C); const long t = i / v.C; r = (int)(t % v.R); b = t / v.R; }
    const int b2 = (int)(b % v.B2), b1 = (int)(b / v.B2);
    const long e = v.off + b1 * v.sB1 + b2 * v.sB2 + r * v.sR + c * v.sC;
    const float gc = g[e] * cf;
    float u;
    if constexpr (FACT) {
      const float rf = rsqrtf(v_row[b * v.R + r] / rcm[b]);
      const float cfac = rsqrtf(v_col[b * v.C + c]);
      u = gc * rf * cfac;
    } else {
      const long vi = (b * v.R + r) * v.C + c;     // canonical order of the per-element state
      const float nv = decay * vfull[vi] + (1.f - decay) * (gc * gc + eps);
      vfull[vi] = nv;
      u = gc * rsqrtf(nv);
    }
    if (momentum > 0.f) {
      float m;
      if constexpr (MOM_BF16) m = (float)reinterpret_cast<bf16*>(mom)[e];
      else m = reinterpret_cast<float*>(mom)[e];
      m = momentum * m + (1.f - momentum) * u;
      if constexpr (MOM_BF16) reinterpret_cast<bf16*>(mom)[e] = (bf16)m;
      else reinterpret_cast<float*>(mom)[e] = m;
      u = m;                                       // optax.ema returns the fp32 EMA, stores the cast
    }
    const float pv = p[e];
    const float upd = -sched * (lr_eff * u + wd * pv);
    const float pn = pv + upd;
    p[e] = pn;
    if (shadow) shadow[e] = (bf16)pn;
    sp += pn * pn;
    su += upd * upd;
  }
  const float a = block_sum_256(sp, sh);
  const float b_ = block_sum_256(su, sh);
  if (threadIdx.x == 0 && stats) {
    atomicAdd(stats + 0, (double)a);
    atomicAdd(stats + 1, (double)b_);
  }
}

// ---- batched form: ALL leaves of a step in four launches (blockIdx.y = leaf).  A B/16 + text-B model has
// ~440 leaves; one launch group per leaf is ~1.5 k tiny launches per step (6 ms of launch overhead: 6 % of a
// 512-pair step).  The table lives in device memory; a workgroup whose blockIdx.x lies beyond its leaf's extent
// (the grid is sized for the largest leaf) or whose leaf is of the other kind returns at once.
struct AfSched { float v[BV_MAX_SCHED]; };

__device__ __forceinline__ AfView af_view_of(const bv_af_leaf& L) {
  AfView v;
  v.off = L.off; v.B1 = L.B1; v.B2 = L.B2; v.R = L.R; v.C = L.C;
  v.sB1 = L.sB1; v.sB2 = L.sB2; v.sR = L.sR; v.sC = L.sC;
  return v;
}

__global__ __launch_bounds__(256) void af_rows_batched(const float* __restrict__ g, const bv_af_leaf* __restrict__ leaves,
                                                       float* __restrict__ state, const double* gsq,
                                                       float clip_norm, float decay, float eps) {
  __shared__ float sh[4];
  const bv_af_leaf L = leaves[blockIdx.y];
  const long B = (long)L.B1 * L.B2;
  if (!L.factored || blockIdx.x >= B * L.R) return;
  const AfView v = af_view_of(L);
  float* v_row = state + L.soff;
  const int br = blockIdx.x;
  const int r = br % v.R, b = br / v.R;
  const int b2 = b % v.B2, b1 = b / v.B2;
  const float cf = clip_factor(gsq, clip_norm);
  const float* base = g + v.off + b1 * v.sB1 + b2 * v.sB2 + r * v.sR;
  float acc = 0.f;
  for (int c = threadIdx.x; c < v.C; c += 256) {
    const float x = base[c * v.sC] * cf;
    acc += x * x + eps;
  }
  const float s = block_sum_256(acc, sh);
  if (threadIdx.x == 0) v_row[br] = decay * v_row[br] + (1.f - decay) * (s / (float)v.C);
}

__global__ __launch_bounds__(256) void af_cols_batched(const float* __restrict__ g, const bv_af_leaf* __restrict__ leaves,
                                                       float* __restrict__ state, const double* gsq,
                                                       float clip_norm, float decay, float eps) {
  const bv_af_leaf L = leaves[blockIdx.y];
  if (!L.factored) return;
  const AfView v = af_view_of(L);
  const long B = (long)v.B1 * v.B2;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= B * v.C) return;
  float* v_col = state + L.soff + B * v.R;
  const int c = (int)(i % v.C), b = (int)(i / v.C);
  const int b2 = b % v.B2, b1 = b / v.B2;
  const float cf = clip_factor(gsq, clip_norm);
  const float* base = g + v.off + b1 * v.sB1 + b2 * v.sB2 + c * v.sC;
  float acc = 0.f;
  for (int r = 0; r < v.R; ++r) {
    const float x = base[r * v.sR] * cf;
    acc += x * x + eps;
  }
  v_col[i] = decay * v_col[i] + (1.f - decay) * (acc / (float)v.R);
}

__global__ __launch_bounds__(256) void af_rcm_batched(const bv_af_leaf* __restrict__ leaves, float* __restrict__ state) {
  __shared__ float sh[4];
  const bv_af_leaf L = leaves[blockIdx.y];
  const long B = (long)L.B1 * L.B2;
  if (!L.factored || blockIdx.x >= B) return;
  const float* v_row = state + L.soff;
  float* rcm = state + L.soff + B * L.R + B * L.C;
  float acc = 0.f;
  for (int r = threadIdx.x; r < L.R; r += 256) acc += v_row[(long)blockIdx.x * L.R + r];
  const float s = block_sum_256(acc, sh);
  if (threadIdx.x == 0) rcm[blockIdx.x] = s / (float)L.R;
}

// optax.clip_by_block_rms (scale_by_adafactor(clipping_threshold=...), optax.py:208): per LEAF, u /= max(1, rms(u) /
// threshold) with u the output of scale_by_factored_rms.  This pass adds every leaf's sum of u^2 (from the statistics
// the row / column / rcm passes just updated; an unfactored leaf's v is advanced in registers only - the update pass
// stores it) into usq[leaf]; the update pass turns it into the scale.
__global__ __launch_bounds__(256) void af_usq_batched(const float* __restrict__ g, const bv_af_leaf* __restrict__ leaves,
                                                      const float* __restrict__ state, const double* gsq, float clip_norm,
                                                      float decay, float eps, double* __restrict__ usq) {
  __shared__ float sh[4];
  const bv_af_leaf L = leaves[blockIdx.y];
  const AfView v = af_view_of(L);
  const long B = (long)v.B1 * v.B2;
  const long total = B * v.R * v.C;
  if ((long)blockIdx.x * 256 >= total) return;
  const float* v_row = state + L.soff;
  const float* v_col = v_row + B * v.R;
  const float* rcm = v_col + B * v.C;
  const float* vfull = state + L.soff;
  const float cf = clip_factor(gsq, clip_norm);
  float acc = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int r, c;
    long b;
    if (L.r_fast) { r = (int)(i % v.R); const long t = i / v.R; c = (int)(t % v.C); b = t / v.C; }
    else { c = (int)(i % v.C); const long t = i / v.C; r = (int)(t % v.R); b = t / v.R; }
    const int b2 = (int)(b % v.B2), b1 = (int)(b / v.B2);
    const long e = v.off + b1 * v.sB1 + b2 * v.sB2 + r * v.sR + c * v.sC;
    const float gc = g[e] * cf;
    float u;
    if (L.factored) {
      u = gc * rsqrtf(v_row[b * v.R + r] / rcm[b]) * rsqrtf(v_col[b * v.C + c]);
    } else {
      const long vi = (b * v.R + r) * v.C + c;
      u = gc * rsqrtf(decay * vfull[vi] + (1.f - decay) * (gc * gc + eps));
    }
    acc += u * u;
  }
  const float s = block_sum_256(acc, sh);
  if (threadIdx.x == 0) atomicAdd(usq + blockIdx.y, (double)s);
}

template <bool MOM_BF16>
__global__ __launch_bounds__(256) void af_update_batched(float* __restrict__ p, const float* __restrict__ g,
                                                         void* __restrict__ mom, bf16* __restrict__ shadow,
                                                         const bv_af_leaf* __restrict__ leaves,
                                                         float* __restrict__ state, const double* gsq,
                                                         float clip_norm, float decay, float eps, float momentum,
                                                         AfSched sched, double* __restrict__ stats,
                                                         float block_rms_clip, const double* __restrict__ usq) {
  __shared__ float sh[4];
  const bv_af_leaf L = leaves[blockIdx.y];
  const AfView v = af_view_of(L);
  const long B = (long)v.B1 * v.B2;
  const long total = B * v.R * v.C;
  if ((long)blockIdx.x * 256 >= total) return;          // (uniform per workgroup: the reductions below are safe)
  // clip_by_block_rms: u / max(1, rms(u) / threshold), rms over the whole leaf (af_usq_batched)
  const float bscale = block_rms_clip > 0.f
                           ? 1.f / fmaxf(1.f, sqrtf((float)(usq[blockIdx.y] / (double)total)) / block_rms_clip) : 1.f;
  const float* v_row = state + L.soff;
  const float* v_col = v_row + B * v.R;
  const float* rcm = v_col + B * v.C;
  float* vfull = state + L.soff;
  const float cf = clip_factor(gsq, clip_norm);
  const float sc = sched.v[L.sched_idx];
  float sp = 0.f, su = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int r, c;
    long b;
    if (L.r_fast) { r = (int)(i % v.R); const long t = i / v.R; c = (int)(t % v.C); b = t / v.C; }
    else { c = (int)(i % v.C); const long t = i / v.C; r = (int)(t % v.R); b = t / v.R; }
    const int b2 = (int)(b % v.B2), b1 = (int)(b / v.B2);
    const long e = v.off + b1 * v.sB1 + b2 * v.sB2 + r * v.sR + c * v.sC;
    const float gc = g[e] * cf;
    float u;
    if (L.factored) {
      const float rf = rsqrtf(v_row[b * v.R + r] / rcm[b]);
      const float cfac = rsqrtf(v_col[b * v.C + c]);
      u = gc * rf * cfac;
    } else {
      const long vi = (b * v.R + r) * v.C + c;
      const float nv = decay * vfull[vi] + (1.f - decay) * (gc * gc + eps);
      vfull[vi] = nv;
      u = gc * rsqrtf(nv);
    }
    u *= bscale;
    if (momentum > 0.f) {
      float m;
      if constexpr (MOM_BF16) m = (float)reinterpret_cast<bf16*>(mom)[e];
      else m = reinterpret_cast<float*>(mom)[e];
      m = momentum * m + (1.f - momentum) * u;
      if constexpr (MOM_BF16) reinterpret_cast<bf16*>(mom)[e] = (bf16)m;
      else reinterpret_cast<float*>(mom)[e] = m;
      u = m;
    }
    const float pv = p[e];
    const float upd = -sc * (L.lr_eff * u + L.wd * pv);
    const float pn = pv + upd;
    p[e] = pn;
    if (shadow) shadow[e] = (bf16)pn;
    sp += pn * pn;
    su += upd * upd;
  }
  const float a = block_sum_256(sp, sh);
  const float b_ = block_sum_256(su, sh);
  if (threadIdx.x == 0 && stats) {
    atomicAdd(stats + 0, (double)a);
    atomicAdd(stats + 1, (double)b_);
  }
}

}  // namespace

// The whole Adafactor step of a model: the leaves of bv_adafactor_leaf as a device table, four launches.
// max_rows = max over factored leaves of B*R, max_cols = max B*C, max_b = max B, max_total = max B*R*C over
// all leaves (grid extents; host-computed once per model).
extern "C" int bv_adafactor_step(float* params, const float* grads, void* momentum, int mom_bf16, void* shadow_bf16,
                                 const bv_af_leaf* leaves, int nleaves, long max_rows, long max_cols, long max_b,
                                 long max_total, float* state, const double* gsq, float clip_norm, float decay,
                                 float eps, float mom, const float* sched, int nsched, double* stats,
                                 float block_rms_clip, double* block_usq, void* stream) {
  BV_REQUIRE(nleaves > 0 && nleaves <= 65535 && leaves != nullptr, "bv_adafactor_step: bad leaf table (%d)", nleaves);
  BV_REQUIRE(block_rms_clip <= 0.f || block_usq != nullptr, "bv_adafactor_step: clip_by_block_rms needs the usq scratch");
  BV_REQUIRE(nsched >= 1 && nsched <= BV_MAX_SCHED && sched != nullptr, "bv_adafactor_step: 1..%d schedule values", BV_MAX_SCHED);
  BV_REQUIRE(clip_norm <= 0.f || gsq != nullptr, "bv_adafactor_step: clipping needs gsq");
  BV_REQUIRE(max_total > 0, "bv_adafactor_step: empty model");
  hipStream_t s = (hipStream_t)stream;
  AfSched sc;
  for (int i = 0; i < BV_MAX_SCHED; ++i) sc.v[i] = i < nsched ? sched[i] : 0.f;
  if (max_rows > 0) {
    hipLaunchKernelGGL(af_rows_batched, dim3((unsigned)max_rows, nleaves), dim3(256), 0, s, grads, leaves, state, gsq,
                       clip_norm, decay, eps);
    hipLaunchKernelGGL(af_cols_batched, dim3((unsigned)((max_cols + 255) / 256), nleaves), dim3(256), 0, s, grads, leaves,
                       state, gsq, clip_norm, decay, eps);
    hipLaunchKernelGGL(af_rcm_batched, dim3((unsigned)max_b, nleaves), dim3(256), 0, s, leaves, state);
  }
  long gsz = (max_total + 1023) / 1024;
  if (gsz > 1024) gsz = 1024;
  if (block_rms_clip > 0.f) {
    (void)hipMemsetAsync(block_usq, 0, sizeof(double) * (size_t)nleaves, s);
    hipLaunchKernelGGL(af_usq_batched, dim3((unsigned)gsz, nleaves), dim3(256), 0, s, grads, leaves, (const float*)state, gsq,
                       clip_norm, decay, eps, block_usq);
  }
  if (mom_bf16)
    hipLaunchKernelGGL((af_update_batched<true>), dim3((unsigned)gsz, nleaves), dim3(256), 0, s, params, grads, momentum,
                       (bf16*)shadow_bf16, leaves, state, gsq, clip_norm, decay, eps, mom, sc, stats, block_rms_clip,
                       (const double*)block_usq);
  else
    hipLaunchKernelGGL((af_update_batched<false>), dim3((unsigned)gsz, nleaves), dim3(256), 0, s, params, grads, momentum,
                       (bf16*)shadow_bf16, leaves, state, gsq, clip_norm, decay, eps, mom, sc, stats, block_rms_clip,
                       (const double*)block_usq);
  return bv_check_launch("bv_adafactor_step");
}

// One leaf of the Adafactor step.  view = {off, B1, B2, R, C, sB1, sB2, sR, sC} (9 longs; unfactored
// leaves pass any view that enumerates their elements).  state: factored -> v_row [B*R], v_col [B*C],
// rcm [B] consecutively; unfactored -> v [numel] in the view's canonical [B1][B2][R][C] order.
extern "C" int bv_adafactor_leaf(float* params, const float* grads, void* momentum, int mom_bf16,
                                 void* shadow_bf16, const long* view, float* state, int factored,
                                 const double* gsq, float clip_norm, float decay, float eps, float mom,
                                 float lr_eff, float wd, float sched, double* stats, void* stream) {
  AfView v;
  v.off = view[0];
  v.B1 = (int)view[1]; v.B2 = (int)view[2]; v.R = (int)view[3]; v.C = (int)view[4];
  v.sB1 = view[5]; v.sB2 = view[6]; v.sR = view[7]; v.sC = view[8];
  BV_REQUIRE(v.B1 > 0 && v.B2 > 0 && v.R > 0 && v.C > 0, "bv_adafactor_leaf: empty view");
  BV_REQUIRE(clip_norm <= 0.f || gsq != nullptr, "bv_adafactor_leaf: clipping needs gsq");
  hipStream_t s = (hipStream_t)stream;
  const long B = (long)v.B1 * v.B2, total = B * v.R * v.C;
  float* v_row = state;
  float* v_col = state + B * v.R;
  float* rcm = v_col + B * v.C;
  if (factored) {
    hipLaunchKernelGGL(af_rows_kernel, dim3((unsigned)(B * v.R)), dim3(256), 0, s, grads, v, v_row, gsq,
                       clip_norm, decay, eps);
    hipLaunchKernelGGL(af_cols_kernel, dim3((unsigned)((B * v.C + 255) / 256)), dim3(256), 0, s, grads, v, v_col,
                       gsq, clip_norm, decay, eps);
    hipLaunchKernelGGL(af_rcm_kernel, dim3((unsigned)B), dim3(256), 0, s, (const float*)v_row, rcm, v.R);
  }
  long gsz = (total + 1023) / 1024;
  if (gsz > 2048) gsz = 2048;
  const int r_fast = v.sR < v.sC;
  const dim3 grid((unsigned)gsz), block(256);
#define BV_AF_LAUNCH(F, MB)                                                                              \
  hipLaunchKernelGGL((af_update_kernel<F, MB>), grid, block, 0, s, params, grads, momentum, (bf16*)shadow_bf16, \
                     v, (const float*)v_row, (const float*)v_col, (const float*)rcm, state, gsq, clip_norm,   \
                     decay, eps, mom, lr_eff, wd, sched, r_fast, stats)
  if (factored) { if (mom_bf16) BV_AF_LAUNCH(true, true); else BV_AF_LAUNCH(true, false); }
  else { if (mom_bf16) BV_AF_LAUNCH(false, true); else BV_AF_LAUNCH(false, false); }
#undef BV_AF_LAUNCH
  return bv_check_launch("bv_adafactor_leaf");
}
