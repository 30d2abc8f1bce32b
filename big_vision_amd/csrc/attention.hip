// Entry points of the Dh = 64 self-attention core and the single-query attention of the MAP head, gfx950.
//
// Replaces the attention core inside flax nn.MultiHeadDotProductAttention as used by the reference at
// big_vision/models/vit.py:93-98 (encoder blocks of the image tower and, via vit.Encoder, of the text tower,
// models/proj/image_text/text_transformer.py:72-75):
//   q = q / sqrt(Dh);  S = q k^T;  P = softmax_rows(S);  O = P v
// per (sample, head); no bias, no dropout on this path; optional key-padding lengths (NaFlex, BERT).
//
// The kernels live in attention3.hip (forward; two-launch backward for masked sequences and L > 208) and
// attention5.hip (the backward in one launch for the step's shapes); this file validates arguments, dispatches
// and holds the MAP head's kernels (models/vit.py:176-178: ONE query per sample and head, a wave per pair).
// Rounds 1-3 also kept a general global-memory kernel set and the first LDS-resident set (attention2.hip) behind
// A/B switches; both were superseded and left the library in round 5 (git history has them).
#include "bv_common.h"
#include "bvhip_internal.h"

namespace {

constexpr int DH = 64;

// ------------------------------------------------- MAP head (single query) --
// One wave per (sample, head).  kv packed [n*L][2][H][64].
__global__ __launch_bounds__(256) void map_attn_fwd_kernel(const bf16* __restrict__ q,
                                                           const bf16* __restrict__ kv,
                                                           bf16* __restrict__ o, float* __restrict__ p,
                                                           const int* __restrict__ kv_len,
                                                           int n, int L, int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sp = reinterpret_cast<float*>(smem) + (threadIdx.x >> 6) * L;
  const int lane = threadIdx.x & 63;
  const long pair = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= (long)n * H) return;
  const long i = pair / H;
  const int h = (int)(pair - i * H);
  const int Lk = kv_len ? min(kv_len[i], L) : L;   // keys >= Lk are padding (NaFlex pool mask): p = 0
  const long ld = 2L * H * DH;
  const bf16* kb_ = kv + i * L * ld + h * DH;
  const bf16* vb_ = kb_ + (long)H * DH;
  const uint4* qp = reinterpret_cast<const uint4*>(q + (i * H + h) * DH);
  uint4 qv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) qv[c] = qp[c];
  float mx = -INFINITY;
  for (int l = lane; l < L; l += 64) {
    const uint4* kp = reinterpret_cast<const uint4*>(kb_ + (long)l * ld);
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint4 a = kp[c], b = qv[c];
      acc += bflo(a.x) * bflo(b.x) + bfhi(a.x) * bfhi(b.x) + bflo(a.y) * bflo(b.y) +
             bfhi(a.y) * bfhi(b.y) + bflo(a.z) * bflo(b.z) + bfhi(a.z) * bfhi(b.z) +
             bflo(a.w) * bflo(b.w) + bfhi(a.w) * bfhi(b.w);
    }
    acc = l < Lk ? acc * scale : -INFINITY;
    sp[l] = acc;
    mx = fmaxf(mx, acc);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int l = lane; l < L; l += 64) {
    const float e = __expf(sp[l] - mx);
    sp[l] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  float* prow = p + (i * H + h) * L;
  for (int l = lane; l < L; l += 64) {
    const float pv = sp[l] * inv;
    sp[l] = pv;
    prow[l] = pv;
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  float acc = 0.f;
  for (int l = 0; l < L; ++l) acc += sp[l] * bf2f(vb_[(long)l * ld + lane]);
  o[(i * H + h) * DH + lane] = f2bf(acc);
}

__global__ __launch_bounds__(256) void map_attn_bwd_kernel(const bf16* __restrict__ q,
                                                           const bf16* __restrict__ kv,
                                                           const float* __restrict__ p,
                                                           const bf16* __restrict__ d_o,
                                                           bf16* __restrict__ dq, bf16* __restrict__ dkv,
                                                           int n, int L, int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sp = reinterpret_cast<float*>(smem) + (threadIdx.x >> 6) * 2 * L;  // p
  float* sd = sp + L;                                                       // ds
  const int lane = threadIdx.x & 63;
  const long pair = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= (long)n * H) return;
  const long i = pair / H;
  const int h = (int)(pair - i * H);
  const long ld = 2L * H * DH;
  const bf16* kb_ = kv + i * L * ld + h * DH;
  const bf16* vb_ = kb_ + (long)H * DH;
  bf16* dkb_ = dkv + i * L * ld + h * DH;
  bf16* dvb_ = dkb_ + (long)H * DH;
  const uint4* gp = reinterpret_cast<const uint4*>(d_o + (i * H + h) * DH);
  uint4 gv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) gv[c] = gp[c];
  const float* prow = p + (i * H + h) * L;
  float dsum = 0.f;
  for (int l = lane; l < L; l += 64) {
    const uint4* vp = reinterpret_cast<const uint4*>(vb_ + (long)l * ld);
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint4 a = vp[c], b = gv[c];
      acc += bflo(a.x) * bflo(b.x) + bfhi(a.x) * bfhi(b.x) + bflo(a.y) * bflo(b.y) +
             bfhi(a.y) * bfhi(b.y) + bflo(a.z) * bflo(b.z) + bfhi(a.z) * bfhi(b.z) +
             bflo(a.w) * bflo(b.w) + bfhi(a.w) * bfhi(b.w);
    }
    const float pv = prow[l];
    sp[l] = pv;
    sd[l] = acc;  // dp
    dsum += pv * acc;
  }
  dsum = wave_sum(dsum);
  for (int l = lane; l < L; l += 64) sd[l] = sp[l] * (sd[l] - dsum);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  const float g = bf2f(d_o[(i * H + h) * DH + lane]);
  const float qd = bf2f(q[(i * H + h) * DH + lane]) * scale;
  float dqa = 0.f;
  for (int l = 0; l < L; ++l) {
    const float ds = sd[l];
    dvb_[(long)l * ld + lane] = f2bf(sp[l] * g);
    dkb_[(long)l * ld + lane] = f2bf(ds * qd);
    dqa += ds * bf2f(kb_[(long)l * ld + lane]);
  }
  dq[(i * H + h) * DH + lane] = f2bf(dqa * scale);
}

}  // namespace

// attention3.hip: one query fragment per wave iteration, prefetched fragments, exact delta, optional key-padding
// length per sample; its backward hands the shapes attention5.hip covers to that kernel
int bv_attn3_fwd(const void* qkv, void* o, float* lse, const int* kv_len, int n, int L, int H, void* stream,
                 const bv_ctx* ctx);
int bv_attn3_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                 float* dbias, const int* kv_len, int n, int L, int H, void* stream, const bv_ctx* ctx);

extern "C" int bv_attn_fwd(const void* qkv, void* o, float* lse, int n, int L, int H, void* stream, const bv_ctx* ctx) {
  BV_REQUIRE(n > 0 && L > 0 && H > 0, "bv_attn_fwd: bad shape n=%d L=%d H=%d", n, L, H);
  BV_REQUIRE(L <= 576, "bv_attn_fwd: L=%d > 576 not supported", L);
  BV_REQUIRE((uintptr_t)qkv % 16 == 0 && (uintptr_t)o % 16 == 0, "bv_attn_fwd: unaligned pointers");
  return bv_attn3_fwd(qkv, o, lse, nullptr, n, L, H, stream, ctx);
}

extern "C" int bv_attn_bwd(const void* qkv, const void* o, const void* d_o, const float* lse,
                           float* delta, void* dqkv, float* dbias_rows, int n, int L, int H, void* stream,
                           const bv_ctx* ctx) {
  BV_REQUIRE(n > 0 && L > 0 && H > 0, "bv_attn_bwd: bad shape n=%d L=%d H=%d", n, L, H);
  BV_REQUIRE(L <= 576, "bv_attn_bwd: L=%d > 576 not supported", L);
  return bv_attn3_bwd(qkv, o, d_o, lse, delta, dqkv, dbias_rows, nullptr, n, L, H, stream, ctx);
}

// Self-attention with a key-padding length per sample (kv_len[i] valid keys, 1 <= kv_len[i] <= L;
// keys >= kv_len[i] get zero probability, their dK / dV rows are written as zeros; query rows are
// all computed).  Replaces nn.MultiHeadDotProductAttention(mask=...) of the NaFlex tower
// (models/proj/image_text/naflex_vit.py:84-293) for masks that are a valid PREFIX of the sequence
// (NaFlex pads at the end).  kv_len = NULL: no mask.
extern "C" int bv_attn_fwd_masked(const void* qkv, void* o, float* lse, const int* kv_len, int n, int L, int H,
                                  void* stream, const bv_ctx* ctx) {
  BV_REQUIRE(n > 0 && L > 0 && H > 0, "bv_attn_fwd_masked: bad shape n=%d L=%d H=%d", n, L, H);
  BV_REQUIRE(L <= 576, "bv_attn_fwd_masked: L=%d > 576 not supported", L);
  BV_REQUIRE((uintptr_t)qkv % 16 == 0 && (uintptr_t)o % 16 == 0, "bv_attn_fwd_masked: unaligned pointers");
  return bv_attn3_fwd(qkv, o, lse, kv_len, n, L, H, stream, ctx);
}
extern "C" int bv_attn_bwd_masked(const void* qkv, const void* d_o, const float* lse, const int* kv_len,
                                  float* delta, void* dqkv, float* dbias_rows, int n, int L, int H, void* stream,
                                  const bv_ctx* ctx) {
  BV_REQUIRE(n > 0 && L > 0 && H > 0, "bv_attn_bwd_masked: bad shape n=%d L=%d H=%d", n, L, H);
  BV_REQUIRE(L <= 576, "bv_attn_bwd_masked: L=%d > 576 not supported", L);
  return bv_attn3_bwd(qkv, nullptr, d_o, lse, delta, dqkv, dbias_rows, kv_len, n, L, H, stream, ctx);
}

// kv_len (optional, int32 [n]): keys >= kv_len[i] get probability 0 - the pool mask of the NaFlex MAP
// head (models/proj/image_text/naflex_vit.py:183-199).  The backward needs no mask: it works from the
// saved probabilities, which are exactly 0 there.
extern "C" int bv_map_attn_fwd_masked(const void* q, const void* kv, void* o, float* p, const int* kv_len,
                                      int n, int L, int H, void* stream) {
  BV_REQUIRE(n > 0 && L > 0 && H > 0 && L <= 2048, "bv_map_attn_fwd: bad shape n=%d L=%d H=%d", n, L, H);
  const long pairs = (long)n * H;
  hipLaunchKernelGGL(map_attn_fwd_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256),
                     4 * L * sizeof(float), (hipStream_t)stream, (const bf16*)q, (const bf16*)kv, (bf16*)o, p,
                     kv_len, n, L, H, 0.125f);
  return bv_check_launch("bv_map_attn_fwd");
}
extern "C" int bv_map_attn_fwd(const void* q, const void* kv, void* o, float* p, int n, int L, int H,
                               void* stream) {
  return bv_map_attn_fwd_masked(q, kv, o, p, nullptr, n, L, H, stream);
}

extern "C" int bv_map_attn_bwd(const void* q, const void* kv, const float* p, const void* d_o, void* dq,
                               void* dkv, int n, int L, int H, void* stream) {
  BV_REQUIRE(n > 0 && L > 0 && H > 0 && L <= 2048, "bv_map_attn_bwd: bad shape n=%d L=%d H=%d", n, L, H);
  const long pairs = (long)n * H;
  hipLaunchKernelGGL(map_attn_bwd_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256),
                     8 * L * sizeof(float), (hipStream_t)stream, (const bf16*)q, (const bf16*)kv, p,
                     (const bf16*)d_o, (bf16*)dq, (bf16*)dkv, n, L, H, 0.125f);
  return bv_check_launch("bv_map_attn_bwd");
}
