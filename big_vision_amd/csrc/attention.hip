// Fused multi-head self-attention forward / backward for gfx950 (Dh = 64).
//
// Replaces the attention core inside flax nn.MultiHeadDotProductAttention as
// used by the reference at big_vision/models/vit.py:93-98 (encoder blocks of the
// image tower and, via vit.Encoder, of the text tower,
// models/proj/image_text/text_transformer.py:72-75):
//   q = q / sqrt(Dh);  S = q k^T;  P = softmax_rows(S);  O = P v
// per (sample, head); no mask, no bias, no dropout on this path.
//
// Sequence lengths on this path are short (L = 64 text, 196/197 at 224 px,
// 441 at 336 px, 576 at 384 px), so one workgroup owns one (sample, head) and
// a wave owns 16 query rows against ALL keys: the whole score row lives in MFMA
// accumulators and the softmax is exact (no online rescaling).
//
// MFMA operand plan (v_mfma_f32_16x16x32_bf16, D = A*B, lane l supplies
// A[l&15][8*(l>>4)..+7], B[8*(l>>4)..+7][l&15], receives D[4*(l>>4)+r][l&15]):
//   S^T[key][q]  = K Q^T        A = K rows (global/L2), B = Q rows (registers)
//   O^T[d][q]   += V^T P^T      A = V^T (LDS, transposed at staging),
//                               B = P^T straight from the S^T accumulators: two
//                               16-key fragments supply the 8 k-slots of a lane
// The contraction index of the second product is a dummy, so the "slot -> key"
// map only has to agree between A and B; no cross-lane shuffle of P is needed.
// Row-major operands (16-byte rows per lane) are read straight from global
// memory — K/V/Q of one head are 8..72 KiB and stay L2 resident — only the
// transposed operands are staged in LDS.
//
// Backward = three kernels: delta = rowsum(dO * O); a query-owned pass for dQ
// (same structure as the forward with K^T in LDS); a key-owned pass for dK, dV
// (Q^T and dO^T in LDS).  P is recomputed from the saved row log-sum-exp.
#include "bv_common.h"
#include "bvhip_internal.h"

namespace {

constexpr int DH = 64;

__device__ __forceinline__ bf16x8 gfrag(const bf16* base, long ld, int row, int L, int col) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (row < L) v = *reinterpret_cast<const uint4*>(base + (long)row * ld + col);
  return __builtin_bit_cast(bf16x8, v);
}

// Stage src[L][64] (row stride ld elements) transposed into LDS T[64][LV]
// (LV = Lpad + 4 elements per row), rows >= L zero-filled.  256 threads.
__device__ __forceinline__ void stage_transposed(char* T, const bf16* src, long ld, int L, int Lpad,
                                                 int tid) {
  const int lvb = (Lpad + 4) * 2;
  const int dg = tid & 7;
  for (int l0 = (tid >> 3) * 4; l0 < Lpad; l0 += 128) {
    uint32_t w[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (l0 + j < L) v = *reinterpret_cast<const uint4*>(src + (long)(l0 + j) * ld + dg * 8);
      w[j][0] = v.x; w[j][1] = v.y; w[j][2] = v.z; w[j][3] = v.w;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int d = e >> 1;
      uint2 o;
      if ((e & 1) == 0) {
        o.x = (w[0][d] & 0xffffu) | (w[1][d] << 16);
        o.y = (w[2][d] & 0xffffu) | (w[3][d] << 16);
      } else {
        o.x = (w[0][d] >> 16) | (w[1][d] & 0xffff0000u);
        o.y = (w[2][d] >> 16) | (w[3][d] & 0xffff0000u);
      }
      *reinterpret_cast<uint2*>(T + (dg * 8 + e) * lvb + l0 * 2) = o;
    }
  }
}

// A-operand fragment from a transposed LDS image: row d, k-slots = rows
// (f0*16 + 4*lg .. +3) and (f1*16 + 4*lg .. +3) of the original matrix.
__device__ __forceinline__ bf16x8 tfrag(const char* T, int lvb, int d, int f0, int f1, int lg) {
  const uint2 a = *reinterpret_cast<const uint2*>(T + d * lvb + f0 * 32 + lg * 8);
  const uint2 b = *reinterpret_cast<const uint2*>(T + d * lvb + f1 * 32 + lg * 8);
  return __builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, b.x, b.y));
}

__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  return __builtin_bit_cast(bf16x8, make_uint4(pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]),
                                               pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3])));
}

__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------ forward --
template <int KF>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16* __restrict__ qkv,
                                                       bf16* __restrict__ o, float* __restrict__ lse,
                                                       int L, int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LPAD = KF * 16;
  constexpr int LVB = (LPAD + 4) * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int i = blockIdx.x / H, h = blockIdx.x % H;
  const long ld = 3L * H * DH;
  const bf16* qb_ = qkv + (long)i * L * ld + h * DH;
  const bf16* kb_ = qb_ + (long)H * DH;
  const bf16* vb_ = qb_ + 2L * H * DH;
  stage_transposed(smem, vb_, ld, L, LPAD, tid);
  __syncthreads();

  const int nqb = (L + 15) / 16;
  for (int qb = wave; qb < nqb; qb += 4) {
    const int qrow = qb * 16 + lr;
    const bf16x8 q0 = gfrag(qb_, ld, qrow, L, lg * 8);
    const bf16x8 q1 = gfrag(qb_, ld, qrow, L, 32 + lg * 8);
    f32x4 s[KF];
    float mx = -INFINITY;
#pragma unroll
    for (int f = 0; f < KF; ++f) {
      const int krow = f * 16 + lr;
      const bf16x8 k0 = gfrag(kb_, ld, krow, L, lg * 8);
      const bf16x8 k1 = gfrag(kb_, ld, krow, L, 32 + lg * 8);
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
      a = mfma16(k0, q0, a);
      a = mfma16(k1, q1, a);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = f * 16 + lg * 4 + r;
        a[r] = key < L ? a[r] * scale : -INFINITY;
        mx = fmaxf(mx, a[r]);
      }
      s[f] = a;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int f = 0; f < KF; ++f) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(s[f][r] - mx);
        s[f][r] = p;
        sum += p;
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    f32x4 oa[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) oa[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int fp = 0; fp < KF / 2; ++fp) {
      const bf16x8 pf = pack8(s[2 * fp], s[2 * fp + 1]);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const bf16x8 vf = tfrag(smem, LVB, d * 16 + lr, 2 * fp, 2 * fp + 1, lg);
        oa[d] = mfma16(vf, pf, oa[d]);
      }
    }
    if (qrow < L) {
      const float inv = 1.0f / sum;
      bf16* orow = o + ((long)i * L + qrow) * H * DH + h * DH;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        uint2 u;
        u.x = pack_bf2(oa[d][0] * inv, oa[d][1] * inv);
        u.y = pack_bf2(oa[d][2] * inv, oa[d][3] * inv);
        *reinterpret_cast<uint2*>(orow + d * 16 + lg * 4) = u;
      }
      if (lg == 0) lse[((long)i * H + h) * L + qrow] = mx + __logf(sum);
    }
  }
}

// -------------------------------------------------------------------- delta --
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16* __restrict__ o,
                                                         const bf16* __restrict__ d_o,
                                                         float* __restrict__ delta, int n, int L,
                                                         int H) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // (t, h)
  const long total = (long)n * L * H;
  if (idx >= total) return;
  const long t = idx / H;
  const int h = (int)(idx - t * H);
  const uint4* po = reinterpret_cast<const uint4*>(o + idx * DH);
  const uint4* pd = reinterpret_cast<const uint4*>(d_o + idx * DH);
  float acc = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const uint4 a = po[c], b = pd[c];
    acc += bflo(a.x) * bflo(b.x) + bfhi(a.x) * bfhi(b.x) + bflo(a.y) * bflo(b.y) +
           bfhi(a.y) * bfhi(b.y) + bflo(a.z) * bflo(b.z) + bfhi(a.z) * bfhi(b.z) +
           bflo(a.w) * bflo(b.w) + bfhi(a.w) * bfhi(b.w);
  }
  const long i = t / L;
  const int l = (int)(t - i * L);
  delta[(i * H + h) * L + l] = acc;
}

// ------------------------------------------------------------- backward: dQ --
template <int KF>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const bf16* __restrict__ qkv,
                                                          const bf16* __restrict__ d_o,
                                                          const float* __restrict__ lse,
                                                          const float* __restrict__ delta,
                                                          bf16* __restrict__ dqkv, int L, int H,
                                                          float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LPAD = KF * 16;
  constexpr int LVB = (LPAD + 4) * 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int i = blockIdx.x / H, h = blockIdx.x % H;
  const long ld = 3L * H * DH, ldo = (long)H * DH;
  const bf16* qb_ = qkv + (long)i * L * ld + h * DH;
  const bf16* kb_ = qb_ + (long)H * DH;
  const bf16* vb_ = qb_ + 2L * H * DH;
  const bf16* dob_ = d_o + (long)i * L * ldo + h * DH;
  stage_transposed(smem, kb_, ld, L, LPAD, tid);  // K^T
  __syncthreads();

  const int nqb = (L + 15) / 16;
  for (int qb = wave; qb < nqb; qb += 4) {
    const int qrow = qb * 16 + lr;
    const bf16x8 q0 = gfrag(qb_, ld, qrow, L, lg * 8);
    const bf16x8 q1 = gfrag(qb_, ld, qrow, L, 32 + lg * 8);
    const bf16x8 g0 = gfrag(dob_, ldo, qrow, L, lg * 8);
    const bf16x8 g1 = gfrag(dob_, ldo, qrow, L, 32 + lg * 8);
    float lse_q = INFINITY, del_q = 0.f;
    if (qrow < L) {
      lse_q = lse[((long)i * H + h) * L + qrow];
      del_q = delta[((long)i * H + h) * L + qrow];
    }
    f32x4 dq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int fp = 0; fp < KF / 2; ++fp) {
      f32x4 ds[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int f = 2 * fp + u;
        const int krow = f * 16 + lr;
        const bf16x8 k0 = gfrag(kb_, ld, krow, L, lg * 8);
        const bf16x8 k1 = gfrag(kb_, ld, krow, L, 32 + lg * 8);
        const bf16x8 v0 = gfrag(vb_, ld, krow, L, lg * 8);
        const bf16x8 v1 = gfrag(vb_, ld, krow, L, 32 + lg * 8);
        f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
        st = mfma16(k0, q0, st);
        st = mfma16(k1, q1, st);
        dp = mfma16(v0, g0, dp);
        dp = mfma16(v1, g1, dp);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = f * 16 + lg * 4 + r;
          const float p = key < L ? __expf(st[r] * scale - lse_q) : 0.f;
          ds[u][r] = p * (dp[r] - del_q);
        }
      }
      const bf16x8 dsf = pack8(ds[0], ds[1]);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const bf16x8 kt = tfrag(smem, LVB, d * 16 + lr, 2 * fp, 2 * fp + 1, lg);
        dq[d] = mfma16(kt, dsf, dq[d]);
      }
    }
    if (qrow < L) {
      bf16* row = dqkv + ((long)i * L + qrow) * ld + h * DH;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        uint2 u;
        u.x = pack_bf2(dq[d][0] * scale, dq[d][1] * scale);
        u.y = pack_bf2(dq[d][2] * scale, dq[d][3] * scale);
        *reinterpret_cast<uint2*>(row + d * 16 + lg * 4) = u;
      }
    }
  }
}

// --------------------------------------------------------- backward: dK, dV --
template <int QF>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const bf16* __restrict__ qkv,
                                                           const bf16* __restrict__ d_o,
                                                           const float* __restrict__ lse,
                                                           const float* __restrict__ delta,
                                                           bf16* __restrict__ dqkv, int L, int H,
                                                           float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LPAD = QF * 16;
  constexpr int LVB = (LPAD + 4) * 2;
  char* Qt = smem;
  char* dOt = smem + 64 * LVB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int i = blockIdx.x / H, h = blockIdx.x % H;
  const long ld = 3L * H * DH, ldo = (long)H * DH;
  const bf16* qb_ = qkv + (long)i * L * ld + h * DH;
  const bf16* kb_ = qb_ + (long)H * DH;
  const bf16* vb_ = qb_ + 2L * H * DH;
  const bf16* dob_ = d_o + (long)i * L * ldo + h * DH;
  const float* lse_ = lse + ((long)i * H + h) * L;
  const float* del_ = delta + ((long)i * H + h) * L;
  stage_transposed(Qt, qb_, ld, L, LPAD, tid);
  stage_transposed(dOt, dob_, ldo, L, LPAD, tid);
  __syncthreads();

  const int nkb = (L + 15) / 16;
  for (int kb = wave; kb < nkb; kb += 4) {
    const int krow = kb * 16 + lr;
    // B operands: B[k = d][col = key = lr]
    const bf16x8 k0 = gfrag(kb_, ld, krow, L, lg * 8);
    const bf16x8 k1 = gfrag(kb_, ld, krow, L, 32 + lg * 8);
    const bf16x8 v0 = gfrag(vb_, ld, krow, L, lg * 8);
    const bf16x8 v1 = gfrag(vb_, ld, krow, L, 32 + lg * 8);
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      dk[d] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 2
    for (int ip = 0; ip < QF / 2; ++ip) {
      f32x4 pp[2], ds[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int f = 2 * ip + u;
        const int qrow = f * 16 + lr;
        const bf16x8 q0 = gfrag(qb_, ld, qrow, L, lg * 8);
        const bf16x8 q1 = gfrag(qb_, ld, qrow, L, 32 + lg * 8);
        const bf16x8 g0 = gfrag(dob_, ldo, qrow, L, lg * 8);
        const bf16x8 g1 = gfrag(dob_, ldo, qrow, L, 32 + lg * 8);
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
        s = mfma16(q0, k0, s);    // D[q = 4lg+r][key = lr]
        s = mfma16(q1, k1, s);
        dp = mfma16(g0, v0, dp);  // dP[q][key] = sum_d dO[q][d] V[key][d]
        dp = mfma16(g1, v1, dp);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = f * 16 + lg * 4 + r;
          float p = 0.f, del = 0.f;
          if (q < L) {
            p = __expf(s[r] * scale - lse_[q]);
            del = del_[q];
          }
          pp[u][r] = p;
          ds[u][r] = p * (dp[r] - del);
        }
      }
      const bf16x8 pf = pack8(pp[0], pp[1]);
      const bf16x8 dsf = pack8(ds[0], ds[1]);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const bf16x8 gt = tfrag(dOt, LVB, d * 16 + lr, 2 * ip, 2 * ip + 1, lg);
        const bf16x8 qt = tfrag(Qt, LVB, d * 16 + lr, 2 * ip, 2 * ip + 1, lg);
        dv[d] = mfma16(gt, pf, dv[d]);   // D[d = 4lg+r][key = lr]
        dk[d] = mfma16(qt, dsf, dk[d]);
      }
    }
    if (krow < L) {
      bf16* rowk = dqkv + ((long)i * L + krow) * ld + (long)H * DH + h * DH;
      bf16* rowv = rowk + (long)H * DH;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        uint2 u;
        u.x = pack_bf2(dk[d][0] * scale, dk[d][1] * scale);
        u.y = pack_bf2(dk[d][2] * scale, dk[d][3] * scale);
        *reinterpret_cast<uint2*>(rowk + d * 16 + lg * 4) = u;
        uint2 w;
        w.x = pack_bf2(dv[d][0], dv[d][1]);
        w.y = pack_bf2(dv[d][2], dv[d][3]);
        *reinterpret_cast<uint2*>(rowv + d * 16 + lg * 4) = w;
      }
    }
  }
}

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

template <int KF>
int launch_fwd(const void* qkv, void* o, float* lse, int n, int L, int H, hipStream_t s) {
  const size_t shmem = 64 * (KF * 16 + 4) * 2;
  if (shmem > 65536)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<KF>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
  hipLaunchKernelGGL(attn_fwd_kernel<KF>, dim3(n * H), dim3(256), shmem, s, (const bf16*)qkv, (bf16*)o,
                     lse, L, H, 0.125f);
  return bv_check_launch("bv_attn_fwd");
}
template <int KF>
int launch_bwd(const void* qkv, const void* d_o, const float* lse, const float* delta, void* dqkv,
               int n, int L, int H, hipStream_t s) {
  const size_t sh1 = 64 * (KF * 16 + 4) * 2;
  if (sh1 > 65536)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<KF>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh1);
  hipLaunchKernelGGL(attn_bwd_dq_kernel<KF>, dim3(n * H), dim3(256), sh1, s, (const bf16*)qkv,
                     (const bf16*)d_o, lse, delta, (bf16*)dqkv, L, H, 0.125f);
  int rc = bv_check_launch("bv_attn_bwd(dq)");
  if (rc) return rc;
  const size_t sh2 = 2 * sh1;
  if (sh2 > 65536)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<KF>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh2);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel<KF>, dim3(n * H), dim3(256), sh2, s, (const bf16*)qkv,
                     (const bf16*)d_o, lse, delta, (bf16*)dqkv, L, H, 0.125f);
  return bv_check_launch("bv_attn_bwd(dkv)");
}

}  // namespace

// attention2.hip (LDS-resident operands); the kernels above remain as the general
// fallback selected by bv_gemm_fast_path(0).
int bv_attn2_fwd(const void* qkv, void* o, float* lse, int n, int L, int H, void* stream);
int bv_attn2_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta,
                 void* dqkv, float* dbias, int n, int L, int H, void* stream);
// attention3.hip: one query fragment per wave iteration, 8 waves, prefetched fragments, exact delta,
// optional key-padding length per sample (the default fast path)
int bv_attn3_fwd(const void* qkv, void* o, float* lse, const int* kv_len, int n, int L, int H, void* stream);
int bv_attn3_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                 float* dbias, const int* kv_len, int n, int L, int H, void* stream);
int bv_fast_path_enabled();
static int g_attn_impl = 3;
// diagnostics / A-B benchmarking: 3 = attention3.hip (default), 2 = attention2.hip.  impl < 0 only
// queries; returns the old value.
extern "C" int bv_attn_impl(int impl) {
  const int old = g_attn_impl;
  if (impl == 2 || impl == 3) g_attn_impl = impl;
  return old;
}
extern "C" int bv_colsum(const void* x, int x_is_f32, long ldx, float* out, int rows, int cols, void* stream);

extern "C" int bv_attn_fwd(const void* qkv, void* o, float* lse, int n, int L, int H, void* stream) {
  BV_REQUIRE(n > 0 && L > 0 && H > 0, "bv_attn_fwd: bad shape n=%d L=%d H=%d", n, L, H);
  BV_REQUIRE(L <= 576, "bv_attn_fwd: L=%d > 576 not supported", L);
  BV_REQUIRE((uintptr_t)qkv % 16 == 0 && (uintptr_t)o % 16 == 0, "bv_attn_fwd: unaligned pointers");
  if (bv_fast_path_enabled())
    return g_attn_impl == 3 ? bv_attn3_fwd(qkv, o, lse, nullptr, n, L, H, stream) : bv_attn2_fwd(qkv, o, lse, n, L, H, stream);
  hipStream_t s = (hipStream_t)stream;
  if (L <= 64) return launch_fwd<4>(qkv, o, lse, n, L, H, s);
  if (L <= 224) return launch_fwd<14>(qkv, o, lse, n, L, H, s);
  if (L <= 448) return launch_fwd<28>(qkv, o, lse, n, L, H, s);
  return launch_fwd<36>(qkv, o, lse, n, L, H, s);
}

extern "C" int bv_attn_bwd(const void* qkv, const void* o, const void* d_o, const float* lse,
                           float* delta, void* dqkv, float* dbias_rows, int n, int L, int H, void* stream) {
  BV_REQUIRE(n > 0 && L > 0 && H > 0, "bv_attn_bwd: bad shape n=%d L=%d H=%d", n, L, H);
  BV_REQUIRE(L <= 576, "bv_attn_bwd: L=%d > 576 not supported", L);
  if (bv_fast_path_enabled())
    return g_attn_impl == 3 ? bv_attn3_bwd(qkv, o, d_o, lse, delta, dqkv, dbias_rows, nullptr, n, L, H, stream)
                            : bv_attn2_bwd(qkv, o, d_o, lse, delta, dqkv, dbias_rows, n, L, H, stream);
  hipStream_t s = (hipStream_t)stream;
  const long total = (long)n * L * H;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                     (const bf16*)o, (const bf16*)d_o, delta, n, L, H);
  int rc = bv_check_launch("bv_attn_bwd(delta)");
  if (rc) return rc;
  if (L <= 64) rc = launch_bwd<4>(qkv, d_o, lse, delta, dqkv, n, L, H, s);
  else if (L <= 224) rc = launch_bwd<14>(qkv, d_o, lse, delta, dqkv, n, L, H, s);
  else if (L <= 448) rc = launch_bwd<28>(qkv, d_o, lse, delta, dqkv, n, L, H, s);
  else rc = launch_bwd<36>(qkv, d_o, lse, delta, dqkv, n, L, H, s);
  if (rc || !dbias_rows) return rc;
  // general path: per-sample column sums with the stand-alone reduction
  (void)hipMemsetAsync(dbias_rows, 0, sizeof(float) * (size_t)n * 3 * H * 64, s);
  for (int i = 0; i < n && !rc; ++i)
    rc = bv_colsum((const bf16*)dqkv + (long)i * L * 3 * H * 64, 0, 3L * H * 64, dbias_rows + (long)i * 3 * H * 64, L,
                   3 * H * 64, stream);
  return rc;
}

// Self-attention with a key-padding length per sample (kv_len[i] valid keys, 1 <= kv_len[i] <= L;
// keys >= kv_len[i] get zero probability, their dK / dV rows are written as zeros; query rows are
// all computed).  Replaces nn.MultiHeadDotProductAttention(mask=...) of the NaFlex tower
// (models/proj/image_text/naflex_vit.py:84-293) for masks that are a valid PREFIX of the sequence
// (NaFlex pads at the end).  kv_len = NULL: no mask.
extern "C" int bv_attn_fwd_masked(const void* qkv, void* o, float* lse, const int* kv_len, int n, int L, int H,
                                  void* stream) {
  BV_REQUIRE(n > 0 && L > 0 && H > 0, "bv_attn_fwd_masked: bad shape n=%d L=%d H=%d", n, L, H);
  BV_REQUIRE(L <= 576, "bv_attn_fwd_masked: L=%d > 576 not supported", L);
  BV_REQUIRE((uintptr_t)qkv % 16 == 0 && (uintptr_t)o % 16 == 0, "bv_attn_fwd_masked: unaligned pointers");
  return bv_attn3_fwd(qkv, o, lse, kv_len, n, L, H, stream);
}
extern "C" int bv_attn_bwd_masked(const void* qkv, const void* d_o, const float* lse, const int* kv_len,
                                  float* delta, void* dqkv, float* dbias_rows, int n, int L, int H, void* stream) {
  BV_REQUIRE(n > 0 && L > 0 && H > 0, "bv_attn_bwd_masked: bad shape n=%d L=%d H=%d", n, L, H);
  BV_REQUIRE(L <= 576, "bv_attn_bwd_masked: L=%d > 576 not supported", L);
  return bv_attn3_bwd(qkv, nullptr, d_o, lse, delta, dqkv, dbias_rows, kv_len, n, L, H, stream);
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
