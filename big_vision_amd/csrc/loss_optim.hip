// Sigmoid loss, softmax cross-entropy, fp32 strided GEMM, grad-norm and the
// fused optimizer chain for gfx950.  fp32 arithmetic throughout (the logits are
// scaled by t = exp(t') ~ 10..100, so bf16 would be visible in the loss).
#include "bv_common.h"
#include "bvhip_internal.h"

namespace {

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// ------------------------------------------------------------- siglip loss --
// trainers/proj/image_text/siglip.py:291-306.  raw[n][B] -> G in place.
__global__ __launch_bounds__(256) void siglip_loss_kernel(float* __restrict__ raw,
                                                          const float* __restrict__ t_param,
                                                          const float* __restrict__ b_param,
                                                          double* __restrict__ stats, int n, int B,
                                                          int row_offset, float inv_bg) {
  __shared__ float sh[4];
  const float t = __expf(t_param[0]);
  const float b = b_param ? b_param[0] : 0.f;
  const long total = (long)n * B;
  float l_acc = 0.f, dt_acc = 0.f, db_acc = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int row = (int)(i / B);
    const int col = (int)(i - (long)row * B);
    const float r = raw[i];
    const float s = t * r + b;
    const float m = (col == row_offset + row) ? 1.f : -1.f;
    const float x = m * s;
    const float e = __expf(-fabsf(x));
    // -log_sigmoid(x) = max(-x, 0) + log1p(exp(-|x|))
    l_acc += fmaxf(-x, 0.f) + log1pf(e);
    // sigmoid(-x)
    const float sig = x >= 0.f ? e / (1.f + e) : 1.f / (1.f + e);
    const float g = -inv_bg * m * sig;
    raw[i] = g;
    dt_acc += g * (s - b);
    db_acc += g;
  }
  const float l = block_sum_256(l_acc, sh);
  const float dt = block_sum_256(dt_acc, sh);
  const float db = block_sum_256(db_acc, sh);
  if (threadIdx.x == 0) {
    atomicAdd(stats + 0, (double)l * (double)inv_bg);
    atomicAdd(stats + 1, (double)dt);
    atomicAdd(stats + 2, (double)db);
  }
}

// ------------------------------------------------------------- logit stats --
// The per-device logit statistics the pmap trainer logs next to the sigmoid loss
// (trainers/proj/image_text/_deprecated_contrastive.py:143-160): logits = t raw + b of this rank's
// n images against all B texts; "me" = the n x n block of this rank's own texts (columns
// row_offset .. row_offset + n), whose diagonal holds the positives.
//   part[block][0..5] = min / max of {positives, local negatives, all negatives}
//   part[block][6..8] = sums of the same three sets
// logit_stats_finish reduces the partials to out[9] = {pos_min, pos_max, pos_avg, local_neg_min,
// local_neg_max, local_neg_avg, neg_min, neg_max, neg_avg}.
__device__ __forceinline__ float block_red_256(float v, float* sh, int op) {   // op 0 min, 1 max, 2 sum
  v = op == 0 ? -wave_max(-v) : (op == 1 ? wave_max(v) : wave_sum(v));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  if (op == 0) return fminf(fminf(sh[0], sh[1]), fminf(sh[2], sh[3]));
  if (op == 1) return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  return sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void logit_stats_kernel(const float* __restrict__ raw,
                                                          const float* __restrict__ t_param,
                                                          const float* __restrict__ b_param,
                                                          float* __restrict__ part, int n, int B,
                                                          int row_offset) {
  __shared__ float sh[4];
  const float t = __expf(t_param[0]);
  const float b = b_param ? b_param[0] : 0.f;
  const long total = (long)n * B;
  float v[9] = {INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int row = (int)(i / B);
    const int col = (int)(i - (long)row * B) - row_offset;
    const float s = t * raw[i] + b;
    if (col == row) {
      v[0] = fminf(v[0], s); v[1] = fmaxf(v[1], s); v[6] += s;
    } else {
      if (col >= 0 && col < n) { v[2] = fminf(v[2], s); v[3] = fmaxf(v[3], s); v[7] += s; }
      v[4] = fminf(v[4], s); v[5] = fmaxf(v[5], s); v[8] += s;
    }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const float r = block_red_256(v[k], sh, k >= 6 ? 2 : (k & 1));
    if (threadIdx.x == 0) part[blockIdx.x * 9 + k] = r;
  }
}
__global__ __launch_bounds__(64) void logit_stats_finish(const float* __restrict__ part, int nblocks,
                                                         float* __restrict__ out, int n, int B) {
  const int k = threadIdx.x;
  if (k >= 9) return;
  const int src = k < 2 ? k : (k == 2 ? 6 : (k < 5 ? k - 1 : (k == 5 ? 7 : (k < 8 ? k - 2 : 8))));
  // out order {pos_min, pos_max, pos_avg, lneg_min, lneg_max, lneg_avg, neg_min, neg_max, neg_avg}
  const bool is_avg = k == 2 || k == 5 || k == 8;
  const bool is_min = !is_avg && (src & 1) == 0;
  double acc = is_avg ? 0.0 : (is_min ? INFINITY : -INFINITY);
  for (int j = 0; j < nblocks; ++j) {
    const double p = part[j * 9 + src];
    acc = is_avg ? acc + p : (is_min ? fmin(acc, p) : fmax(acc, p));
  }
  if (is_avg) {
    const double cnt = k == 2 ? (double)n : (k == 5 ? (double)n * n - n : (double)n * B - n);
    acc = cnt > 0 ? acc / cnt : 0.0;
  }
  out[k] = (float)acc;
}

// out[0] += sum_i a[i] b[i] (double accumulator): dL/dt' of the softmax contrastive loss
__global__ __launch_bounds__(256) void dot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  long count, double* __restrict__ out) {
  __shared__ float sh[4];
  float acc = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) acc += a[i] * b[i];
  const float r = block_sum_256(acc, sh);
  if (threadIdx.x == 0) atomicAdd(out, (double)r);
}

// ------------------------------------------------------------ softmax xent --
// utils.py:276-281; one workgroup per row.
__global__ __launch_bounds__(256) void softmax_xent_kernel(const float* __restrict__ logits,
                                                           const float* __restrict__ labels,
                                                           double* __restrict__ loss_sum,
                                                           float* __restrict__ dlogits, int n, int C,
                                                           float inv_n) {
  __shared__ float sh[4];
  const int r = blockIdx.x;
  const float* lr = logits + (long)r * C;
  const float* yr = labels + (long)r * C;
  float mx = -INFINITY;
  for (int c = threadIdx.x; c < C; c += 256) mx = fmaxf(mx, lr[c]);
  mx = wave_max(mx);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  float se = 0.f, sy = 0.f, syl = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    se += __expf(lr[c] - mx);
    sy += yr[c];
    syl += yr[c] * (lr[c] - mx);
  }
  se = block_sum_256(se, sh);
  sy = block_sum_256(sy, sh);
  syl = block_sum_256(syl, sh);
  const float lse = logf(se);
  // -sum_c y (l - mx - lse) = -(syl - sy*lse)
  if (threadIdx.x == 0) atomicAdd(loss_sum, (double)(-(syl - sy * lse)) * (double)inv_n);
  if (dlogits) {
    for (int c = threadIdx.x; c < C; c += 256) {
      const float p = __expf(lr[c] - mx - lse);
      dlogits[(long)r * C + c] = (p * sy - yr[c]) * inv_n;
    }
  }
}

// ------------------------------------------------------------ sigmoid xent --
// utils.py:236-243: nll_i = -sum_c [y log sigmoid(l) + (1 - y) log sigmoid(-l)], mean over i.
// log sigmoid(x) = min(x, 0) - log1p(exp(-|x|)) (stable); d nll / d l = sigmoid(l) - y.
// One workgroup per row.
__global__ __launch_bounds__(256) void sigmoid_xent_kernel(const float* __restrict__ logits,
                                                           const float* __restrict__ labels,
                                                           double* __restrict__ loss_sum,
                                                           float* __restrict__ dlogits, int n, int C,
                                                           float inv_n) {
  __shared__ float sh[4];
  const int r = blockIdx.x;
  const float* lr = logits + (long)r * C;
  const float* yr = labels + (long)r * C;
  float acc = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float l = lr[c], y = yr[c];
    const float sp = log1pf(__expf(-fabsf(l)));          // softplus(-|l|)
    const float log_p = fminf(l, 0.f) - sp;               // log sigmoid(l)
    const float log_np = fminf(-l, 0.f) - sp;             // log sigmoid(-l)
    acc -= y * log_p + (1.f - y) * log_np;
    if (dlogits) dlogits[(long)r * C + c] = (1.f / (1.f + __expf(-l)) - y) * inv_n;
  }
  acc = block_sum_256(acc, sh);
  if (threadIdx.x == 0) atomicAdd(loss_sum, (double)acc * (double)inv_n);
}

// ----------------------------------------------------------- strided sgemm --
// 64x64 tile, 16x16 threads, 4x4 micro-tile, BK = 16.
__global__ __launch_bounds__(256) void sgemm_strided_kernel(const float* __restrict__ A, long sam,
                                                            long sak, const float* __restrict__ B,
                                                            long sbk, long sbn, float* __restrict__ C,
                                                            long ldc, int M, int N, int K, float alpha,
                                                            float beta, const float* __restrict__ log_alpha) {
  if (log_alpha) alpha *= __expf(log_alpha[0]);
  __shared__ float As[16][68];
  __shared__ float Bs[16][68];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    // each thread loads 4 elements of A (64 m x 16 k) and 4 of B (16 k x 64 n)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int idx = threadIdx.x + e * 256;
      int mm, kk;
      if (sak == 1) { kk = idx & 15; mm = idx >> 4; } else { mm = idx & 63; kk = idx >> 6; }
      const int gm = m0 + mm, gk = k0 + kk;
      As[kk][mm] = (gm < M && gk < K) ? A[(long)gm * sam + (long)gk * sak] : 0.f;
      int nn, kb;
      if (sbk == 1) { kb = idx & 15; nn = idx >> 4; } else { nn = idx & 63; kb = idx >> 6; }
      const int gn = n0 + nn, gkb = k0 + kb;
      Bs[kb][nn] = (gn < N && gkb < K) ? B[(long)gkb * sbk + (long)gn * sbn] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= N) continue;
      float* c = C + (long)m * ldc + n;
      *c = alpha * acc[i][j] + (beta != 0.f ? beta * (*c) : 0.f);
    }
  }
}


// The same product on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, bit-for-bit a k-ordered
// fmaf chain - the results are IDENTICAL to sgemm_strided_kernel's, MI355X_MICROARCH.md "FP32-input MFMA"), for the
// B x B logits of the sigmoid loss and their two gradient products (trainers/proj/image_text/siglip.py:291): 3 x 0.64 ms
// per step on the VALU kernel at B = 4096 (40 TFLOP/s).  TILE x TILE outputs per 256-thread workgroup (4 waves as
// 2 x 2, each (TILE/64)^2 MFMA tiles of 32 x 32), BK = 16 through the LDS as [k][m] / [k][n] rows (a lane reads one
// float: lanes 0-31 k, lanes 32-63 k + 1), the next K-step's global loads in flight during the MFMAs.
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <int TILE>
__global__ __launch_bounds__(256) void sgemm_mfma_kernel(const float* __restrict__ A, long sam, long sak,
                                                         const float* __restrict__ B, long sbk, long sbn,
                                                         float* __restrict__ C, long ldc, int M, int N, int K,
                                                         float alpha, float beta, const float* __restrict__ log_alpha) {
  constexpr int BK = 16, LD = TILE + 4, PER = TILE * BK / 256, WT = TILE / 64;   // WT x WT MFMA tiles per wave
  if (log_alpha) alpha *= __expf(log_alpha[0]);
  __shared__ float As[BK][LD];
  __shared__ float Bs[BK][LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.y * TILE, n0 = blockIdx.x * TILE;
  f32x16_t acc[WT][WT];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float ra[PER], rb[PER];
  auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const int idx = tid + e * 256;
      int mm, kk;
      if (sak == 1) { kk = idx & (BK - 1); mm = idx / BK; } else { mm = idx % TILE; kk = idx / TILE; }
      const int gm = m0 + mm, gk = k0 + kk;
      ra[e] = (gm < M && gk < K) ? A[(long)gm * sam + (long)gk * sak] : 0.f;
      int nn, kb;
      if (sbk == 1) { kb = idx & (BK - 1); nn = idx / BK; } else { nn = idx % TILE; kb = idx / TILE; }
      const int gn = n0 + nn, gkb = k0 + kb;
      rb[e] = (gn < N && gkb < K) ? B[(long)gkb * sbk + (long)gn * sbn] : 0.f;
    }
  };
  auto lstore = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const int idx = tid + e * 256;
      int mm, kk;
      if (sak == 1) { kk = idx & (BK - 1); mm = idx / BK; } else { mm = idx % TILE; kk = idx / TILE; }
      As[kk][mm] = ra[e];
      int nn, kb;
      if (sbk == 1) { kb = idx & (BK - 1); nn = idx / BK; } else { nn = idx % TILE; kb = idx / TILE; }
      Bs[kb][nn] = rb[e];
    }
  };
  gload(0);
  for (int k0 = 0; k0 < K; k0 += BK) {
    lstore();
    __syncthreads();
    if (k0 + BK < K) gload(k0 + BK);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a[WT], b[WT];
#pragma unroll
      for (int i = 0; i < WT; ++i) a[i] = As[kk + (lane >> 5)][wr * (TILE / 2) + i * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < WT; ++j) b[j] = Bs[kk + (lane >> 5)][wc * (TILE / 2) + j * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // C layout of the 32 x 32 tile: register r of lane l = row (r / 4) * 8 + (l / 32) * 4 + r % 4, column l % 32
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j) {
      const int n = n0 + wc * (TILE / 2) + j * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * (TILE / 2) + i * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
        if (m < M && n < N) {
          float* c = C + (long)m * ldc + n;
          *c = alpha * acc[i][j][r] + (beta != 0.f ? beta * (*c) : 0.f);
        }
      }
    }
}

// ------------------------------------------------------------------ sqnorm --
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ x, long count,
                                                     double* __restrict__ out) {
  __shared__ float sh[4];
  float acc = 0.f;
  const long n4 = count / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 a = *reinterpret_cast<const float4*>(x + i * 4);
    acc += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
  }
  if (blockIdx.x == 0) {
    for (long i = n4 * 4 + threadIdx.x; i < count; i += 256) acc += x[i] * x[i];
  }
  const float s = block_sum_256(acc, sh);
  if (threadIdx.x == 0) atomicAdd(out, (double)s);
}

// -------------------------------------------------------------- adam chain --
// optax.py:143-149.  One workgroup per 1024-element chunk; chunk_seg maps the
// chunk to its hyper-parameter segment.
struct AdamArgs {
  float* p;
  const float* g;
  void* mu;
  float* nu;
  bf16* shadow;
  const bv_adam_seg* segs;
  const int* chunk_seg;
  const double* gsq;
  double* stats;
  float clip_norm, b1, b2, eps, bc1, bc2;
  int mu_bf16;
  float sched[BV_MAX_SCHED];
};

// Streaming accesses of the optimizer pass: master weights, gradients and moments are each touched once per step
// and are far larger than any cache (the bf16 shadow is what the next forward reads: plain stores).
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_nt __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ldnt4(const float* p) {
  const f32x4_nt t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p));
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ uint2 ldnt2(const uint32_t* p) {
  const u32x2_nt t = __builtin_nontemporal_load(reinterpret_cast<const u32x2_nt*>(p));
  return make_uint2(t.x, t.y);
}
__device__ __forceinline__ void stnt4(float* p, const float4& v) {
  f32x4_nt t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  __builtin_nontemporal_store(t, reinterpret_cast<f32x4_nt*>(p));
}
__device__ __forceinline__ void stnt2(uint32_t* p, const uint2& v) {
  u32x2_nt t; t.x = v.x; t.y = v.y;
  __builtin_nontemporal_store(t, reinterpret_cast<u32x2_nt*>(p));
}

__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a, long nchunks) {
  __shared__ double shd[2][256];
  float clip = 1.f;
  if (a.clip_norm > 0.f) {
    const float gn = (float)sqrt(*a.gsq);
    clip = gn > a.clip_norm ? a.clip_norm / gn : 1.f;
  }
  // Grid-stride over the 1024-element chunks: the two norm statistics are accumulated per
  // thread over all its chunks and leave the workgroup as ONE pair of fp64 atomics (a
  // workgroup per chunk put ~400 k atomics on the same two addresses and ran at 1.2 TB/s).
  double sp = 0.0, su = 0.0;   // per-thread over all its chunks (fp32 within a chunk)
  for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
    float cp = 0.f, cu = 0.f;
    const bv_adam_seg hp = a.segs[a.chunk_seg[c]];
    const float sched = a.sched[hp.sched_idx & (BV_MAX_SCHED - 1)];
    const long i = c * 1024 + threadIdx.x * 4;
    const float4 p4 = ldnt4(a.p + i);
    const float4 g4 = ldnt4(a.g + i);
    const float4 v4 = ldnt4(a.nu + i);
    float m[4];
    if (a.mu_bf16) {
      const uint2 u = ldnt2(reinterpret_cast<const uint32_t*>(reinterpret_cast<const bf16*>(a.mu) + i));
      m[0] = bflo(u.x); m[1] = bfhi(u.x); m[2] = bflo(u.y); m[3] = bfhi(u.y);
    } else {
      const float4 m4 = ldnt4(reinterpret_cast<const float*>(a.mu) + i);
      m[0] = m4.x; m[1] = m4.y; m[2] = m4.z; m[3] = m4.w;
    }
    float p[4] = {p4.x, p4.y, p4.z, p4.w};
    const float g[4] = {g4.x * clip, g4.y * clip, g4.z * clip, g4.w * clip};
    float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      m[e] = a.b1 * m[e] + (1.f - a.b1) * g[e];
      v[e] = a.b2 * v[e] + (1.f - a.b2) * g[e] * g[e];
      float u = (m[e] / a.bc1) / (sqrtf(v[e] / a.bc2) + a.eps);
      u = hp.lr_eff * u + hp.wd_eff * p[e];
      u *= sched;
      p[e] -= u;
      cp += p[e] * p[e];
      cu += u * u;
    }
    sp += (double)cp;
    su += (double)cu;
    stnt4(a.p + i, make_float4(p[0], p[1], p[2], p[3]));
    stnt4(a.nu + i, make_float4(v[0], v[1], v[2], v[3]));
    if (a.mu_bf16) {
      uint2 u;
      u.x = pack_bf2(m[0], m[1]);
      u.y = pack_bf2(m[2], m[3]);
      stnt2(reinterpret_cast<uint32_t*>(reinterpret_cast<bf16*>(a.mu) + i), u);
    } else {
      stnt4(reinterpret_cast<float*>(a.mu) + i, make_float4(m[0], m[1], m[2], m[3]));
    }
    if (a.shadow) {
      uint2 u;
      u.x = pack_bf2(p[0], p[1]);
      u.y = pack_bf2(p[2], p[3]);
      *reinterpret_cast<uint2*>(a.shadow + i) = u;
    }
  }
  if (a.stats) {
    shd[0][threadIdx.x] = sp;
    shd[1][threadIdx.x] = su;
    __syncthreads();
    if (threadIdx.x < 2) {
      double t = 0.0;
      for (int k = 0; k < 256; ++k) t += shd[threadIdx.x][k];
      atomicAdd(a.stats + threadIdx.x, t);
    }
  }
}

}  // namespace

extern "C" int bv_siglip_loss(float* raw, const float* t_param, const float* b_param, double* stats,
                              int n, int B, int row_offset, int B_global, void* stream) {
  BV_REQUIRE(n > 0 && B > 0 && B_global > 0, "bv_siglip_loss: bad shape n=%d B=%d B_global=%d", n, B, B_global);
  BV_REQUIRE(row_offset >= 0 && row_offset + n <= B, "bv_siglip_loss: positive diagonal [%d,%d) outside B=%d", row_offset, row_offset + n, B);
  long g = ((long)n * B + 1023) / 1024;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(siglip_loss_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, raw, t_param,
                     b_param, stats, n, B, row_offset, 1.0f / (float)B_global);
  return bv_check_launch("bv_siglip_loss");
}

extern "C" int bv_logit_stats(const float* raw, const float* t_param, const float* b_param, float* part,
                              float* out9, int n, int B, int row_offset, void* stream) {
  BV_REQUIRE(n > 0 && B >= n, "bv_logit_stats: bad shape n=%d B=%d", n, B);
  BV_REQUIRE(row_offset >= 0 && row_offset + n <= B, "bv_logit_stats: diagonal [%d,%d) outside B=%d", row_offset, row_offset + n, B);
  long g = ((long)n * B + 1023) / 1024;
  if (g > BV_LOGIT_STATS_BLOCKS) g = BV_LOGIT_STATS_BLOCKS;
  hipLaunchKernelGGL(logit_stats_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, raw, t_param, b_param,
                     part, n, B, row_offset);
  hipLaunchKernelGGL(logit_stats_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)part, (int)g,
                     out9, n, B);
  return bv_check_launch("bv_logit_stats");
}

extern "C" int bv_dot_f32(const float* a, const float* b, long count, double* out, void* stream) {
  BV_REQUIRE(count > 0, "bv_dot_f32: empty input");
  long g = (count + 1023) / 1024;
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(dot_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, a, b, count, out);
  return bv_check_launch("bv_dot_f32");
}

extern "C" int bv_softmax_xent(const float* logits, const float* labels, double* loss_sum,
                               float* dlogits, int n, int C, int n_global, void* stream) {
  BV_REQUIRE(n > 0 && C > 0 && n_global >= n, "bv_softmax_xent: bad shape n=%d C=%d n_global=%d", n, C, n_global);
  hipLaunchKernelGGL(softmax_xent_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, logits, labels,
                     loss_sum, dlogits, n, C, 1.0f / (float)n_global);
  return bv_check_launch("bv_softmax_xent");
}

extern "C" int bv_sigmoid_xent(const float* logits, const float* labels, double* loss_sum,
                               float* dlogits, int n, int C, int n_global, void* stream) {
  BV_REQUIRE(n > 0 && C > 0 && n_global >= n, "bv_sigmoid_xent: bad shape n=%d C=%d n_global=%d", n, C, n_global);
  hipLaunchKernelGGL(sigmoid_xent_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, logits, labels,
                     loss_sum, dlogits, n, C, 1.0f / (float)n_global);
  return bv_check_launch("bv_sigmoid_xent");
}

extern "C" int bv_sgemm_strided(const float* A, long sam, long sak, const float* B, long sbk, long sbn,
                                float* C, long ldc, int M, int N, int K, float alpha, float beta,
                                const float* log_alpha, void* stream, const bv_ctx* ctx) {
  BV_REQUIRE(M > 0 && N > 0 && K > 0, "bv_sgemm_strided: empty problem");
  const bool g_sgemm_mfma = bv_opt(ctx, BV_OPT_SGEMM_MFMA) != 0;   // 0: always the VALU kernel (A/B and bit-equality test)
  // the fp32 matrix pipe for everything that fills at least one 64 x 64 tile (bit-identical results: both kernels
  // are k-ordered fmaf chains); 128 x 128 tiles once those alone give every CU two workgroups
  const long t128 = (long)((N + 127) / 128) * ((M + 127) / 128);
  if (g_sgemm_mfma && M >= 64 && N >= 64 && K >= 16) {
    if (t128 >= 512)
      hipLaunchKernelGGL(sgemm_mfma_kernel<128>, dim3((N + 127) / 128, (M + 127) / 128), dim3(256), 0, (hipStream_t)stream,
                         A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, alpha, beta, log_alpha);
    else
      hipLaunchKernelGGL(sgemm_mfma_kernel<64>, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                         A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, alpha, beta, log_alpha);
    return bv_check_launch("bv_sgemm_strided(mfma)");
  }
  dim3 grid((N + 63) / 64, (M + 63) / 64);
  hipLaunchKernelGGL(sgemm_strided_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, sam, sak, B, sbk,
                     sbn, C, ldc, M, N, K, alpha, beta, log_alpha);
  return bv_check_launch("bv_sgemm_strided");
}

extern "C" int bv_sqnorm(const float* x, long count, double* sqnorm_out, void* stream) {
  BV_REQUIRE(count > 0 && (uintptr_t)x % 16 == 0, "bv_sqnorm: empty or unaligned input");
  long g = (count / 4 + 255) / 256;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(sqnorm_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, x, count, sqnorm_out);
  return bv_check_launch("bv_sqnorm");
}

extern "C" int bv_adam_step(float* params, const float* grads, void* mu, int mu_bf16, float* nu,
                            void* shadow_bf16, const bv_adam_seg* segs, const int* chunk_seg,
                            long count, const float* sched, int nsched, const double* gsq,
                            float clip_norm, float b1, float b2, float eps, float bc1, float bc2,
                            double* stats, void* stream) {
  BV_REQUIRE(count > 0 && count % 1024 == 0, "bv_adam_step: count=%ld must be a positive multiple of 1024", count);
  BV_REQUIRE(clip_norm <= 0.f || gsq != nullptr, "bv_adam_step: clipping needs gsq");
  BV_REQUIRE(sched != nullptr && nsched >= 1 && nsched <= BV_MAX_SCHED, "bv_adam_step: 1..%d schedule values required", BV_MAX_SCHED);
  AdamArgs a;
  for (int i = 0; i < BV_MAX_SCHED; ++i) a.sched[i] = i < nsched ? sched[i] : 0.f;
  a.p = params; a.g = grads; a.mu = mu; a.nu = nu; a.shadow = (bf16*)shadow_bf16;
  a.segs = segs; a.chunk_seg = chunk_seg; a.gsq = gsq; a.stats = stats;
  a.clip_norm = clip_norm; a.b1 = b1; a.b2 = b2; a.eps = eps; a.bc1 = bc1; a.bc2 = bc2;
  a.mu_bf16 = mu_bf16;
  const long nchunks = count / 1024;
  const unsigned grid = (unsigned)(nchunks < 4096 ? nchunks : 4096);   // 16 workgroups per CU
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, nchunks);
  return bv_check_launch("bv_adam_step");
}
