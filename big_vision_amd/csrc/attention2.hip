// Fused self-attention forward / backward for gfx950, Dh = 64, L <= 576 — LDS-resident
// K/V (forward, dQ) or Q/dO (dK,dV) tiles, every MFMA operand read from LDS or
// registers.  Replaces the attention core of flax nn.MultiHeadDotProductAttention
// (reference big_vision/models/vit.py:93-98; text tower via vit.Encoder,
// models/proj/image_text/text_transformer.py:72-75):
//   S = (q/sqrt(Dh)) k^T,  P = softmax_rows(S),  O = P v      (no mask / dropout)
// and its backward (jax.value_and_grad, trainers/proj/image_text/siglip.py:311):
//   dV = P^T dO, dP = dO V^T, dS = P o (dP - rowsum(dO o O)), dQ = dS K/sqrt(Dh),
//   dK = dS^T Q/sqrt(Dh).
//
// One workgroup (4 waves) owns one (sample, head).  Sequences are short (64 text
// tokens, 196/197 patches at 224 px, 441 at 336 px, 576 at 384 px), so the tiles
// of one head fit in LDS ([rows][64] bf16 = 128 B per row, <= 72 KiB per tensor)
// and a wave keeps a whole score row in MFMA accumulators: the softmax is exact,
// no online rescaling.  A wave works on QF (1 or 2) 16-row fragments at a time so
// every LDS fragment read feeds QF MFMAs.
//
// LDS tile layout ("T64"): row r = 8 chunks of 16 B, chunk c stored at position
// c ^ (((r >> 1) & 3) << 1).  The same image serves
//   * row-operand reads (ds_read_b128: lane = row, 8 consecutive d), and
//   * transposed reads (ds_read_b64_tr_b16: contraction over rows, lane = d),
// both bank-conflict free (brute-forced over the gfx950 lane groups).
//
// MFMA plan (v_mfma_f32_16x16x32_bf16, D = A*B: lane l supplies A[l&15][8*(l>>4)..+7],
// B[8*(l>>4)..+7][l&15], receives D[4*(l>>4)+r][l&15]):
//   forward   S^T[key][q] = K Q^T ; O^T[d][q] += V^T P^T   (P^T straight from the
//             S^T accumulators: two key fragments supply a lane's 8 k-slots)
//   dQ pass   S^T, dP^T[key][q] = V dO^T ; dQ^T[d][q] += K^T dS^T
//   dK,dV     S[q][key] = Q K^T, dP[q][key] = dO V^T ; dV^T[d][key] += dO^T P,
//             dK^T[d][key] += Q^T dS
#include "attn_common.h"
#include "bvhip_internal.h"

namespace {
using namespace bvattn;

// ------------------------------------------------------------------ forward --
template <int KF, int QF>
__global__ __launch_bounds__(256, 2) void attn2_fwd_kernel(const bf16* __restrict__ qkv,
                                                           bf16* __restrict__ o,
                                                           float* __restrict__ lse, int L, int H,
                                                           float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kt = smem;
  char* Vt = smem + KF * 16 * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int i = blockIdx.x / H, h = blockIdx.x % H;
  const long ld = 3L * H * DH;
  const bf16* qb_ = qkv + (long)i * L * ld + h * DH;
  const bf16* kb_ = qb_ + (long)H * DH;
  const bf16* vb_ = qb_ + 2L * H * DH;
  t64_stage2<KF * 16>(Kt, kb_, ld, Vt, vb_, ld, L, tid);
  __syncthreads();
  const float c = scale * LOG2E;

  for (int blk = wave; blk * QF * 16 < L; blk += 4) {
    bf16x8 q[QF][2];
#pragma unroll
    for (int u = 0; u < QF; ++u) {
      const int qrow = (blk * QF + u) * 16 + lr;
      q[u][0] = gfrag(qb_, ld, qrow, L, lg * 8);
      q[u][1] = gfrag(qb_, ld, qrow, L, 32 + lg * 8);
    }
    f32x4 s[QF][KF];
#pragma unroll
    for (int f = 0; f < KF; ++f) {
      const bf16x8 k0 = t64_row(Kt, f * 16 + lr, lg);
      const bf16x8 k1 = t64_row(Kt, f * 16 + lr, 4 + lg);
#pragma unroll
      for (int u = 0; u < QF; ++u) {
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        a = mfma16(k0, q[u][0], a);
        a = mfma16(k1, q[u][1], a);
        s[u][f] = a;
      }
      if (f & 1) __builtin_amdgcn_sched_barrier(0);   // bound load hoisting (register pressure)
    }
    float inv[QF], lsev[QF];
#pragma unroll
    for (int u = 0; u < QF; ++u) {
      float mx = -INFINITY;
#pragma unroll
      for (int f = 0; f < KF; ++f) {
        if (f * 16 + 16 > L) {   // fragment straddles / lies beyond the sequence end: mask keys >= L
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (f * 16 + lg * 4 + r >= L) s[u][f][r] = -INFINITY;
        }
        mx = fmaxf(mx, fmaxf(fmaxf(s[u][f][0], s[u][f][1]), fmaxf(s[u][f][2], s[u][f][3])));
      }
      mx = xmax4(mx);
      const float mc = mx * c;
      float sum = 0.f;
#pragma unroll
      for (int f = 0; f < KF; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(s[u][f][r] * c - mc);
          s[u][f][r] = p;
          sum += p;
        }
      sum = xsum4(sum);
      inv[u] = 1.0f / sum;
      lsev[u] = mx * scale + __logf(sum);
    }
    f32x4 oa[QF][4];
#pragma unroll
    for (int u = 0; u < QF; ++u)
#pragma unroll
      for (int d = 0; d < 4; ++d) oa[u][d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int fp = 0; fp < KF / 2; ++fp) {
      bf16x8 pf[QF];
#pragma unroll
      for (int u = 0; u < QF; ++u) pf[u] = pack8(s[u][2 * fp], s[u][2 * fp + 1]);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const bf16x8 vf = t64_trpair(Vt, (2 * fp) * 16 + 4 * lg, (2 * fp + 1) * 16 + 4 * lg, d, lr);
#pragma unroll
        for (int u = 0; u < QF; ++u) oa[u][d] = mfma16(vf, pf[u], oa[u][d]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 0; u < QF; ++u) {
      const int qrow = (blk * QF + u) * 16 + lr;
      if (qrow < L) {
        bf16* orow = o + ((long)i * L + qrow) * H * DH + h * DH;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          uint2 w;
          w.x = pack_bf2(oa[u][d][0] * inv[u], oa[u][d][1] * inv[u]);
          w.y = pack_bf2(oa[u][d][2] * inv[u], oa[u][d][3] * inv[u]);
          *reinterpret_cast<uint2*>(orow + d * 16 + lg * 4) = w;
        }
        if (lg == 0) lse[((long)i * H + h) * L + qrow] = lsev[u];
      }
    }
  }
}

// ------------------------------------------------------------- backward: dQ --
template <int KF, int QF>
__global__ __launch_bounds__(256, 2) void attn2_bwd_dq_kernel(const bf16* __restrict__ qkv,
                                                              const bf16* __restrict__ o,
                                                              const bf16* __restrict__ d_o,
                                                              const float* __restrict__ lse,
                                                              float* __restrict__ delta,
                                                              bf16* __restrict__ dqkv,
                                                              float* __restrict__ dbias, int L, int H,
                                                              float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kt = smem;
  char* Vt = smem + KF * 16 * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int i = blockIdx.x / H, h = blockIdx.x % H;
  const long ld = 3L * H * DH, ldo = (long)H * DH;
  const bf16* qb_ = qkv + (long)i * L * ld + h * DH;
  const bf16* kb_ = qb_ + (long)H * DH;
  const bf16* vb_ = qb_ + 2L * H * DH;
  const bf16* dob_ = d_o + (long)i * L * ldo + h * DH;
  const bf16* ob_ = o + (long)i * L * ldo + h * DH;
  t64_stage2<KF * 16>(Kt, kb_, ld, Vt, vb_, ld, L, tid);
  __syncthreads();
  const float c = scale * LOG2E;
  f32x4 cq[4];   // per-lane partial column sums of dQ (query-bias gradient), rows >= L contribute 0
#pragma unroll
  for (int d = 0; d < 4; ++d) cq[d] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int blk = wave; blk * QF * 16 < L; blk += 4) {
    bf16x8 q[QF][2], g[QF][2];
    float lse2[QF], del[QF];
#pragma unroll
    for (int u = 0; u < QF; ++u) {
      const int qrow = (blk * QF + u) * 16 + lr;
      q[u][0] = gfrag(qb_, ld, qrow, L, lg * 8);
      q[u][1] = gfrag(qb_, ld, qrow, L, 32 + lg * 8);
      g[u][0] = gfrag(dob_, ldo, qrow, L, lg * 8);
      g[u][1] = gfrag(dob_, ldo, qrow, L, 32 + lg * 8);
      lse2[u] = INFINITY;
      if (qrow < L) lse2[u] = lse[((long)i * H + h) * L + qrow] * LOG2E;
      // delta = rowsum(dO o O) of this query row, computed here from the dO fragments the lane
      // holds anyway (4 lanes x 16 columns per row) and published for the dK,dV pass.
      const bf16x8 o0 = gfrag(ob_, ldo, qrow, L, lg * 8), o1 = gfrag(ob_, ldo, qrow, L, 32 + lg * 8);
      float dsum = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        dsum += (float)g[u][0][e] * (float)o0[e] + (float)g[u][1][e] * (float)o1[e];
      del[u] = xsum4(dsum);
      if (lg == 0 && qrow < L) delta[((long)i * H + h) * L + qrow] = del[u];
    }
    f32x4 dq[QF][4];
#pragma unroll
    for (int u = 0; u < QF; ++u)
#pragma unroll
      for (int d = 0; d < 4; ++d) dq[u][d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int fp = 0; fp < KF / 2; ++fp) {
      f32x4 ds[QF][2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int f = 2 * fp + e;
        const bf16x8 k0 = t64_row(Kt, f * 16 + lr, lg), k1 = t64_row(Kt, f * 16 + lr, 4 + lg);
        const bf16x8 v0 = t64_row(Vt, f * 16 + lr, lg), v1 = t64_row(Vt, f * 16 + lr, 4 + lg);
#pragma unroll
        for (int u = 0; u < QF; ++u) {
          f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
          st = mfma16(k0, q[u][0], st);
          st = mfma16(k1, q[u][1], st);
          dp = mfma16(v0, g[u][0], dp);
          dp = mfma16(v1, g[u][1], dp);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = __builtin_amdgcn_exp2f(st[r] * c - lse2[u]);
            if (f * 16 + 16 > L && f * 16 + lg * 4 + r >= L) p = 0.f;
            ds[u][e][r] = p * (dp[r] - del[u]);
          }
        }
      }
      bf16x8 dsf[QF];
#pragma unroll
      for (int u = 0; u < QF; ++u) dsf[u] = pack8(ds[u][0], ds[u][1]);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const bf16x8 kt = t64_trpair(Kt, (2 * fp) * 16 + 4 * lg, (2 * fp + 1) * 16 + 4 * lg, d, lr);
#pragma unroll
        for (int u = 0; u < QF; ++u) dq[u][d] = mfma16(kt, dsf[u], dq[u][d]);
      }
    }
#pragma unroll
    for (int u = 0; u < QF; ++u) {
      const int qrow = (blk * QF + u) * 16 + lr;
#pragma unroll
      for (int d = 0; d < 4; ++d) cq[d] += dq[u][d];   // padded query rows are exactly 0
      if (qrow < L) {
        bf16* row = dqkv + ((long)i * L + qrow) * ld + h * DH;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          uint2 w;
          w.x = pack_bf2(dq[u][d][0] * scale, dq[u][d][1] * scale);
          w.y = pack_bf2(dq[u][d][2] * scale, dq[u][d][3] * scale);
          *reinterpret_cast<uint2*>(row + d * 16 + lg * 4) = w;
        }
      }
    }
  }
  if (dbias) {
    // per-(sample, head) column sums of dQ (fp32, before the bf16 rounding) -> dbias[i][0][h][:];
    // the 4 waves are combined through LDS (the K tile is dead), the host sums over samples.
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);   // [4 waves][64]
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t = rowsum16(cq[d][r]) * scale;
        if (lr == 0) red[wave * 64 + d * 16 + lg * 4 + r] = t;
      }
    __syncthreads();
    if (tid < 64)
      dbias[((long)i * 3 * H + h) * DH + tid] = red[tid] + red[64 + tid] + red[128 + tid] + red[192 + tid];
  }
}

// --------------------------------------------------------- backward: dK, dV --
// QN = number of 16-row query fragments (even), KB = key fragments per wave iteration.
template <int QN, int KB>
__global__ __launch_bounds__(256, 2) void attn2_bwd_dkv_kernel(const bf16* __restrict__ qkv,
                                                               const bf16* __restrict__ d_o,
                                                               const float* __restrict__ lse,
                                                               const float* __restrict__ delta,
                                                               bf16* __restrict__ dqkv,
                                                               float* __restrict__ dbias, int L, int H,
                                                               float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qt = smem;
  char* Gt = smem + QN * 16 * 128;
  float* lse_s = reinterpret_cast<float*>(smem + 2 * QN * 16 * 128);
  float* del_s = lse_s + QN * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int i = blockIdx.x / H, h = blockIdx.x % H;
  const long ld = 3L * H * DH, ldo = (long)H * DH;
  const bf16* qb_ = qkv + (long)i * L * ld + h * DH;
  const bf16* kb_ = qb_ + (long)H * DH;
  const bf16* vb_ = qb_ + 2L * H * DH;
  const bf16* dob_ = d_o + (long)i * L * ldo + h * DH;
  t64_stage2<QN * 16>(Qt, qb_, ld, Gt, dob_, ldo, L, tid);
  for (int idx = tid; idx < QN * 16; idx += 256) {
    // rows >= L: lse = +inf makes P = exp2(-inf) = 0 without an explicit mask
    lse_s[idx] = idx < L ? lse[((long)i * H + h) * L + idx] * LOG2E : INFINITY;
    del_s[idx] = idx < L ? delta[((long)i * H + h) * L + idx] : 0.f;
  }
  __syncthreads();
  const float c = scale * LOG2E;
  f32x4 ck[4], cv[4];   // per-lane partial column sums of dK / dV (key / value bias gradients)
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    ck[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    cv[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  for (int blk = wave; blk * KB * 16 < L; blk += 4) {
    bf16x8 k[KB][2], v[KB][2];
#pragma unroll
    for (int e = 0; e < KB; ++e) {
      const int krow = (blk * KB + e) * 16 + lr;
      k[e][0] = gfrag(kb_, ld, krow, L, lg * 8);
      k[e][1] = gfrag(kb_, ld, krow, L, 32 + lg * 8);
      v[e][0] = gfrag(vb_, ld, krow, L, lg * 8);
      v[e][1] = gfrag(vb_, ld, krow, L, 32 + lg * 8);
    }
    f32x4 dk[KB][4], dv[KB][4];
#pragma unroll
    for (int e = 0; e < KB; ++e)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        dk[e][d] = f32x4{0.f, 0.f, 0.f, 0.f};
        dv[e][d] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll 1
    for (int ip = 0; ip < QN / 2; ++ip) {
      f32x4 pp[KB][2], ds[KB][2];
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const int f = 2 * ip + w;
        const bf16x8 q0 = t64_row(Qt, f * 16 + lr, lg), q1 = t64_row(Qt, f * 16 + lr, 4 + lg);
        const bf16x8 g0 = t64_row(Gt, f * 16 + lr, lg), g1 = t64_row(Gt, f * 16 + lr, 4 + lg);
        const float4 l4 = *reinterpret_cast<const float4*>(lse_s + f * 16 + lg * 4);
        const float4 d4 = *reinterpret_cast<const float4*>(del_s + f * 16 + lg * 4);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int e = 0; e < KB; ++e) {
          f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
          s = mfma16(q0, k[e][0], s);     // D[q = 4lg+r][key = lr]
          s = mfma16(q1, k[e][1], s);
          dp = mfma16(g0, v[e][0], dp);   // dP[q][key] = sum_d dO[q][d] V[key][d]
          dp = mfma16(g1, v[e][1], dp);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f(s[r] * c - lv[r]);
            pp[e][w][r] = p;
            ds[e][w][r] = p * (dp[r] - dl[r]);
          }
        }
      }
      bf16x8 pf[KB], dsf[KB];
#pragma unroll
      for (int e = 0; e < KB; ++e) {
        pf[e] = pack8(pp[e][0], pp[e][1]);
        dsf[e] = pack8(ds[e][0], ds[e][1]);
      }
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const bf16x8 gt = t64_trpair(Gt, (2 * ip) * 16 + 4 * lg, (2 * ip + 1) * 16 + 4 * lg, d, lr);
        const bf16x8 qt = t64_trpair(Qt, (2 * ip) * 16 + 4 * lg, (2 * ip + 1) * 16 + 4 * lg, d, lr);
#pragma unroll
        for (int e = 0; e < KB; ++e) {
          dv[e][d] = mfma16(gt, pf[e], dv[e][d]);    // D[d = 4lg+r][key = lr]
          dk[e][d] = mfma16(qt, dsf[e], dk[e][d]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < KB; ++e) {
      const int krow = (blk * KB + e) * 16 + lr;
      if (krow < L) {   // (padded key rows hold garbage: P = exp2(-lse) != 0 there)
        bf16* rowk = dqkv + ((long)i * L + krow) * ld + (long)H * DH + h * DH;
        bf16* rowv = rowk + (long)H * DH;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          ck[d] += dk[e][d];
          cv[d] += dv[e][d];
          uint2 a, b;
          a.x = pack_bf2(dk[e][d][0] * scale, dk[e][d][1] * scale);
          a.y = pack_bf2(dk[e][d][2] * scale, dk[e][d][3] * scale);
          *reinterpret_cast<uint2*>(rowk + d * 16 + lg * 4) = a;
          b.x = pack_bf2(dv[e][d][0], dv[e][d][1]);
          b.y = pack_bf2(dv[e][d][2], dv[e][d][3]);
          *reinterpret_cast<uint2*>(rowv + d * 16 + lg * 4) = b;
        }
      }
    }
  }
  if (dbias) {   // dbias[i][1][h][:] (key) and dbias[i][2][h][:] (value), see the dQ kernel
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);   // [4 waves][2][64]
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float tk = rowsum16(ck[d][r]) * scale, tv = rowsum16(cv[d][r]);
        if (lr == 0) {
          red[wave * 128 + d * 16 + lg * 4 + r] = tk;
          red[wave * 128 + 64 + d * 16 + lg * 4 + r] = tv;
        }
      }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, d = tid & 63;
      dbias[((long)i * 3 * H + (long)(1 + which) * H + h) * DH + d] =
          red[tid] + red[128 + tid] + red[256 + tid] + red[384 + tid];
    }
  }
}

template <typename K>
void set_lds(K kernel, size_t bytes) {
  if (bytes > 65536)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int KF, int QF>
int launch_fwd2(const void* qkv, void* o, float* lse, int n, int L, int H, hipStream_t s) {
  const size_t sh = (size_t)KF * 4096;
  set_lds(attn2_fwd_kernel<KF, QF>, sh);
  hipLaunchKernelGGL((attn2_fwd_kernel<KF, QF>), dim3(n * H), dim3(256), sh, s, (const bf16*)qkv,
                     (bf16*)o, lse, L, H, 0.125f);
  return bv_check_launch("bv_attn_fwd");
}
template <int KF, int QF>
int launch_bwd2(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                float* dbias, int n, int L, int H, hipStream_t s) {
  const size_t sh1 = (size_t)KF * 4096;
  set_lds(attn2_bwd_dq_kernel<KF, QF>, sh1);
  hipLaunchKernelGGL((attn2_bwd_dq_kernel<KF, QF>), dim3(n * H), dim3(256), sh1, s, (const bf16*)qkv,
                     (const bf16*)o, (const bf16*)d_o, lse, delta, (bf16*)dqkv, dbias, L, H, 0.125f);
  int rc = bv_check_launch("bv_attn_bwd(dq)");
  if (rc) return rc;
  const size_t sh2 = (size_t)KF * 4096 + (size_t)KF * 16 * 8;
  set_lds(attn2_bwd_dkv_kernel<KF, QF>, sh2);
  hipLaunchKernelGGL((attn2_bwd_dkv_kernel<KF, QF>), dim3(n * H), dim3(256), sh2, s, (const bf16*)qkv,
                     (const bf16*)d_o, lse, delta, (bf16*)dqkv, dbias, L, H, 0.125f);
  return bv_check_launch("bv_attn_bwd(dkv)");
}

}  // namespace

// Entry points used by bv_attn_fwd / bv_attn_bwd (attention.hip).
int bv_attn2_fwd(const void* qkv, void* o, float* lse, int n, int L, int H, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (L <= 64) return launch_fwd2<4, 1>(qkv, o, lse, n, L, H, s);
  if (L <= 224) return launch_fwd2<14, 2>(qkv, o, lse, n, L, H, s);
  if (L <= 448) return launch_fwd2<28, 1>(qkv, o, lse, n, L, H, s);
  return launch_fwd2<36, 1>(qkv, o, lse, n, L, H, s);
}

int bv_attn2_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta,
                 void* dqkv, float* dbias, int n, int L, int H, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  // delta = rowsum(dO o O) is produced by the dQ pass (no separate pass over O and dO)
  if (L <= 64) return launch_bwd2<4, 1>(qkv, o, d_o, lse, delta, dqkv, dbias, n, L, H, s);
  if (L <= 224) return launch_bwd2<14, 2>(qkv, o, d_o, lse, delta, dqkv, dbias, n, L, H, s);
  if (L <= 448) return launch_bwd2<28, 1>(qkv, o, d_o, lse, delta, dqkv, dbias, n, L, H, s);
  return launch_bwd2<36, 1>(qkv, o, d_o, lse, delta, dqkv, dbias, n, L, H, s);
}
