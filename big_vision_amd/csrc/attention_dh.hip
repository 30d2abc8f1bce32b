// Self-attention and MAP-head attention for head dims OTHER than 64 and for any sequence length:
// the general path behind bv_attn_fwd_dh / bv_attn_bwd_dh / bv_map_attn_*_dh.  gfx950.
//
// The LDS-resident kernels (attention3.hip) are specialised for Dh = 64, L <= 576 - every model of
// BASELINE.json.  big_vision's variant table (models/vit.py:284-303) also holds So400m (width 1152,
// 16 heads: Dh = 72, the flagship SigLIP size), `mu` (width 32, 2 heads: Dh = 16, the variant the
// reference's own tests instantiate) and Ti / S at other widths; those run here.  Same mathematics
// and call sites as attention3.hip (flax nn.MultiHeadDotProductAttention inside
// models/vit.py:93-98, MAPHead :176-178; backward = jax.value_and_grad, siglip.py:311):
//   S = (q / sqrt(Dh)) k^T,  P = softmax_rows(S),  O = P v       optional key-padding length per sample
//   dV = P^T dO, dP = dO V^T, dS = P o (dP - delta), dQ = dS K / sqrt(Dh), dK = dS^T Q / sqrt(Dh)
//
// Structure: flash-style.  A workgroup = 4 waves owns 64 query (or key) rows of one (sample, head),
// a wave 16 of them; the other side streams through LDS in blocks of 32 rows, row-major and (for the
// operands contracted over rows) transposed.  Head dim is padded to DP = 16 * NS (zero columns);
// all products are v_mfma_f32_16x16x16_bf16 (K = 16 steps, lane l supplies A[l&15][4*(l>>4)..+3],
// B[4*(l>>4)..+3][l&15], receives D[4*(l>>4)+r][l&15]):
//   S^T[key][q] = K Q^T puts P^T in B-operand layout for O^T[d][q] += V^T[d][key] P^T[key][q];
//   the backward mirrors attention3 (query-owned pass with two sweeps: delta = rowsum(P o dP) exactly,
//   then dQ; key-owned pass for dK, dV).  Online softmax over key blocks (any L).
// Not a tuned kernel (16-row tiles, 8-byte LDS reads): it exists so that every variant of the table
// can be instantiated and trained; the Dh = 64 models never come here.
#include "attn_common.h"
#include "bvhip_internal.h"

namespace {
using namespace bvattn;

constexpr int KB = 32;        // streamed rows per block
constexpr int TP = KB + 4;    // row pitch (elements) of the transposed tiles [DP][KB]

__device__ __forceinline__ s16x4 lds4(const bf16* p) { return *reinterpret_cast<const s16x4*>(p); }
// Wait states between the last MFMAs of a loop and the VALU code that reads the accumulators behind
// the loop exit (hipcc pads this hazard inside a basic block only, see attention3.hip): the nops sit
// in front of empty asm statements that every accumulator passes through (volatile asms keep their order).
template <int NS>
__device__ __forceinline__ void drain(f32x4 (&a)[NS]) {
  asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
#pragma unroll
  for (int d = 0; d < NS; ++d) asm volatile("" : "+v"(a[d]));
}
__device__ __forceinline__ s16x4 zero4() { return s16x4{0, 0, 0, 0}; }

// Stage rows [r0, r0 + KB) of src[L][Dh] (row stride ld) into T[KB][PK] row-major (PK = DP + 8; rows >= Lv
// and columns >= Dh are zero) and, optionally, transposed into Tt[DP][TP].  256 threads.
template <int NS, bool TR>
__device__ __forceinline__ void stage_block(bf16* T, bf16* Tt, const bf16* src, long ld, int r0, int Lv, int Dh,
                                            int tid) {
  constexpr int DP = NS * 16, PK = DP + 8, CH = DP / 8;
  for (int idx = tid; idx < KB * CH; idx += 256) {
    const int row = idx / CH, c = idx - row * CH;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r0 + row < Lv && c * 8 < Dh) v = *reinterpret_cast<const uint4*>(src + (long)(r0 + row) * ld + c * 8);
    *reinterpret_cast<uint4*>(T + row * PK + c * 8) = v;
    if constexpr (TR) {
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 8; ++e)
        reinterpret_cast<unsigned short*>(Tt)[(c * 8 + e) * TP + row] = (unsigned short)(w[e >> 1] >> ((e & 1) * 16));
    }
  }
}

// Row fragment of a [rows][Dh] global matrix for the B operand: lane (lr, lg) gets row (row0 + lr),
// columns ks*16 + 4*lg .. +3 of every k-step.
template <int NS>
__device__ __forceinline__ void load_rowfrag(s16x4 (&f)[NS], const bf16* base, long ld, int row, int Lv, int Dh,
                                             int lg) {
#pragma unroll
  for (int ks = 0; ks < NS; ++ks) {
    f[ks] = zero4();
    if (row < Lv && ks * 16 + 4 * lg < Dh) f[ks] = *reinterpret_cast<const s16x4*>(base + (long)row * ld + ks * 16 + 4 * lg);
  }
}

// ------------------------------------------------------------------ forward --
template <int NS>
__global__ __launch_bounds__(256) void adh_fwd_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ o,
                                                      float* __restrict__ lse, const int* __restrict__ kv_len,
                                                      int L, int H, int Dh, float scale) {
  constexpr int DP = NS * 16, PK = DP + 8;
  __shared__ __attribute__((aligned(16))) bf16 Ks[KB * PK];
  __shared__ __attribute__((aligned(16))) bf16 Vt[DP * TP];
  __shared__ __attribute__((aligned(16))) bf16 Vs[KB * PK];   // row-major copy (only staged, keeps stage_block simple)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int i = blockIdx.x / H, h = blockIdx.x % H;
  const int Lk = kv_len ? max(1, min(kv_len[i], L)) : L;
  const long ld = 3L * H * Dh;
  const bf16* qb_ = qkv + (long)i * L * ld + (long)h * Dh;
  const bf16* kb_ = qb_ + (long)H * Dh;
  const bf16* vb_ = qb_ + 2L * H * Dh;
  const int q0 = blockIdx.y * 64 + wave * 16;
  s16x4 qf[NS];
  load_rowfrag<NS>(qf, qb_, ld, q0 + lr, L, Dh, lg);
  const float c = scale * LOG2E;
  float m = -INFINITY, lsum = 0.f;
  f32x4 oa[NS];
#pragma unroll
  for (int d = 0; d < NS; ++d) oa[d] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < Lk; k0 += KB) {
    __syncthreads();
    stage_block<NS, false>(Ks, nullptr, kb_, ld, k0, Lk, Dh, tid);
    stage_block<NS, true>(Vs, Vt, vb_, ld, k0, Lk, Dh, tid);
    __syncthreads();
    f32x4 s[2];
#pragma unroll
    for (int kf = 0; kf < 2; ++kf) {
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NS; ++ks) a = mfma16k16(lds4(Ks + (kf * 16 + lr) * PK + ks * 16 + 4 * lg), qf[ks], a);
      const int lim = Lk - k0 - kf * 16 - lg * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) s[kf][r] = r < lim ? a[r] * c : -INFINITY;
    }
    float mx = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3])),
                     fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
    mx = fmaxf(m, xmax4(mx));          // finite: every block holds a valid key (k0 < Lk)
    const float alpha = __builtin_amdgcn_exp2f(m - mx);
    m = mx;
    lsum *= alpha;
#pragma unroll
    for (int kf = 0; kf < 2; ++kf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[kf][r] - mx);
        s[kf][r] = p;
        lsum += p;
      }
    const s16x4 p0 = pack4(s[0]), p1 = pack4(s[1]);
#pragma unroll
    for (int d = 0; d < NS; ++d) {
#pragma unroll
      for (int r = 0; r < 4; ++r) oa[d][r] *= alpha;
      oa[d] = mfma16k16(lds4(Vt + (d * 16 + lr) * TP + 4 * lg), p0, oa[d]);
      oa[d] = mfma16k16(lds4(Vt + (d * 16 + lr) * TP + 16 + 4 * lg), p1, oa[d]);
    }
  }
  drain<NS>(oa);
  lsum = xsum4(lsum);
  const float inv = 1.0f / lsum;
  const int qrow = q0 + lr;
  if (qrow < L) {
    bf16* orow = o + ((long)i * L + qrow) * H * Dh + (long)h * Dh;
#pragma unroll
    for (int d = 0; d < NS; ++d) {
      if (d * 16 + 4 * lg < Dh) {
        uint2 w;
        w.x = pack_bf2(oa[d][0] * inv, oa[d][1] * inv);
        w.y = pack_bf2(oa[d][2] * inv, oa[d][3] * inv);
        *reinterpret_cast<uint2*>(orow + d * 16 + 4 * lg) = w;
      }
    }
    if (lg == 0) lse[((long)i * H + h) * L + qrow] = (m + __log2f(lsum)) * 0.6931471805599453f;
  }
}

// ------------------------------------------------------- backward: delta, dQ --
template <int NS>
__global__ __launch_bounds__(256) void adh_bwd_dq_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ d_o,
                                                         const float* __restrict__ lse, float* __restrict__ delta,
                                                         bf16* __restrict__ dqkv, float* __restrict__ dbias,
                                                         const int* __restrict__ kv_len, int L, int H, int Dh,
                                                         float scale) {
  constexpr int DP = NS * 16, PK = DP + 8;
  __shared__ __attribute__((aligned(16))) bf16 Ks[KB * PK];
  __shared__ __attribute__((aligned(16))) bf16 Vs[KB * PK];
  __shared__ __attribute__((aligned(16))) bf16 Kt[DP * TP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int i = blockIdx.x / H, h = blockIdx.x % H;
  const int Lk = kv_len ? max(1, min(kv_len[i], L)) : L;
  const long ld = 3L * H * Dh, ldo = (long)H * Dh;
  const bf16* qb_ = qkv + (long)i * L * ld + (long)h * Dh;
  const bf16* kb_ = qb_ + (long)H * Dh;
  const bf16* vb_ = qb_ + 2L * H * Dh;
  const bf16* dob_ = d_o + (long)i * L * ldo + (long)h * Dh;
  const int q0 = blockIdx.y * 64 + wave * 16, qrow = q0 + lr;
  s16x4 qf[NS], gf[NS];
  load_rowfrag<NS>(qf, qb_, ld, qrow, L, Dh, lg);
  load_rowfrag<NS>(gf, dob_, ldo, qrow, L, Dh, lg);
  const float lse2 = qrow < L ? lse[((long)i * H + h) * L + qrow] * LOG2E : INFINITY;   // rows >= L: P = 0
  const float c = scale * LOG2E;
  f32x4 dq[NS];
#pragma unroll
  for (int d = 0; d < NS; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float del = 0.f;

  for (int sweep = 0; sweep < 2; ++sweep) {
    float dacc = 0.f;
    for (int k0 = 0; k0 < Lk; k0 += KB) {
      __syncthreads();
      stage_block<NS, true>(Ks, Kt, kb_, ld, k0, Lk, Dh, tid);
      stage_block<NS, false>(Vs, nullptr, vb_, ld, k0, Lk, Dh, tid);
      __syncthreads();
      f32x4 ds[2];
#pragma unroll
      for (int kf = 0; kf < 2; ++kf) {
        f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
          st = mfma16k16(lds4(Ks + (kf * 16 + lr) * PK + ks * 16 + 4 * lg), qf[ks], st);
          dp = mfma16k16(lds4(Vs + (kf * 16 + lr) * PK + ks * 16 + 4 * lg), gf[ks], dp);
        }
        const int lim = Lk - k0 - kf * 16 - lg * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c, -lse2));
          const float p = r < lim ? e : 0.f;
          dacc = __builtin_fmaf(p, dp[r], dacc);
          ds[kf][r] = p * (dp[r] - del);
        }
      }
      if (sweep == 1) {   // block-uniform
        const s16x4 d0 = pack4(ds[0]), d1 = pack4(ds[1]);
#pragma unroll
        for (int d = 0; d < NS; ++d) {
          dq[d] = mfma16k16(lds4(Kt + (d * 16 + lr) * TP + 4 * lg), d0, dq[d]);
          dq[d] = mfma16k16(lds4(Kt + (d * 16 + lr) * TP + 16 + 4 * lg), d1, dq[d]);
        }
      }
    }
    if (sweep == 0) {
      del = xsum4(dacc);
      if (lg == 0 && qrow < L) delta[((long)i * H + h) * L + qrow] = del;
    }
  }
  drain<NS>(dq);
  if (qrow < L) {
    bf16* row = dqkv + ((long)i * L + qrow) * ld + (long)h * Dh;
#pragma unroll
    for (int d = 0; d < NS; ++d)
      if (d * 16 + 4 * lg < Dh) {
        uint2 w;
        w.x = pack_bf2(dq[d][0] * scale, dq[d][1] * scale);
        w.y = pack_bf2(dq[d][2] * scale, dq[d][3] * scale);
        *reinterpret_cast<uint2*>(row + d * 16 + 4 * lg) = w;
      }
  }
  if (dbias) {   // column sums over this wave's 16 queries (rows >= L are exactly 0) -> dbias[i][0][h][:]
#pragma unroll
    for (int d = 0; d < NS; ++d)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t = rowsum16(dq[d][r]);
        const int dd = d * 16 + 4 * lg + r;
        if (lr == 0 && dd < Dh) atomicAdd(dbias + ((long)i * 3 * H + h) * Dh + dd, t * scale);
      }
  }
}

// --------------------------------------------------------- backward: dK, dV --
template <int NS>
__global__ __launch_bounds__(256) void adh_bwd_dkv_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ d_o,
                                                          const float* __restrict__ lse,
                                                          const float* __restrict__ delta, bf16* __restrict__ dqkv,
                                                          float* __restrict__ dbias, const int* __restrict__ kv_len,
                                                          int L, int H, int Dh, float scale) {
  constexpr int DP = NS * 16, PK = DP + 8;
  __shared__ __attribute__((aligned(16))) bf16 Qs[KB * PK];
  __shared__ __attribute__((aligned(16))) bf16 Gs[KB * PK];
  __shared__ __attribute__((aligned(16))) bf16 Qt[DP * TP];
  __shared__ __attribute__((aligned(16))) bf16 Gt[DP * TP];
  __shared__ float lse_s[KB], del_s[KB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int i = blockIdx.x / H, h = blockIdx.x % H;
  const int Lk = kv_len ? max(1, min(kv_len[i], L)) : L;
  const long ld = 3L * H * Dh, ldo = (long)H * Dh;
  const bf16* qb_ = qkv + (long)i * L * ld + (long)h * Dh;
  const bf16* kb_ = qb_ + (long)H * Dh;
  const bf16* vb_ = qb_ + 2L * H * Dh;
  const bf16* dob_ = d_o + (long)i * L * ldo + (long)h * Dh;
  const int krow = blockIdx.y * 64 + wave * 16 + lr;
  const bool live = krow < Lk;
  s16x4 kf_[NS], vf_[NS];
  load_rowfrag<NS>(kf_, kb_, ld, krow, Lk, Dh, lg);
  load_rowfrag<NS>(vf_, vb_, ld, krow, Lk, Dh, lg);
  const float c = scale * LOG2E;
  f32x4 dk[NS], dv[NS];
#pragma unroll
  for (int d = 0; d < NS; ++d) {
    dk[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    dv[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (blockIdx.y * 64 < Lk) {   // block-uniform: key blocks beyond the valid keys only write zeros
    for (int r0 = 0; r0 < L; r0 += KB) {
      __syncthreads();
      stage_block<NS, true>(Qs, Qt, qb_, ld, r0, L, Dh, tid);
      stage_block<NS, true>(Gs, Gt, dob_, ldo, r0, L, Dh, tid);
      if (tid < KB) {
        const int q = r0 + tid;
        lse_s[tid] = q < L ? lse[((long)i * H + h) * L + q] * LOG2E : INFINITY;
        del_s[tid] = q < L ? delta[((long)i * H + h) * L + q] : 0.f;
      }
      __syncthreads();
      f32x4 pp[2], ds[2];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
          s = mfma16k16(lds4(Qs + (f * 16 + lr) * PK + ks * 16 + 4 * lg), kf_[ks], s);     // D[q = 4lg+r][key = lr]
          dp = mfma16k16(lds4(Gs + (f * 16 + lr) * PK + ks * 16 + 4 * lg), vf_[ks], dp);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = f * 16 + 4 * lg + r;
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c, -lse_s[q]));
          const float p = live ? e : 0.f;
          pp[f][r] = p;
          ds[f][r] = p * (dp[r] - del_s[q]);
        }
      }
      const s16x4 p0 = pack4(pp[0]), p1 = pack4(pp[1]), d0 = pack4(ds[0]), d1 = pack4(ds[1]);
#pragma unroll
      for (int d = 0; d < NS; ++d) {
        dv[d] = mfma16k16(lds4(Gt + (d * 16 + lr) * TP + 4 * lg), p0, dv[d]);        // D[d = 4lg+r][key = lr]
        dv[d] = mfma16k16(lds4(Gt + (d * 16 + lr) * TP + 16 + 4 * lg), p1, dv[d]);
        dk[d] = mfma16k16(lds4(Qt + (d * 16 + lr) * TP + 4 * lg), d0, dk[d]);
        dk[d] = mfma16k16(lds4(Qt + (d * 16 + lr) * TP + 16 + 4 * lg), d1, dk[d]);
      }
    }
  }
  drain<NS>(dk);
  drain<NS>(dv);
  if (krow < L) {   // rows >= Lk get zeros (the dX GEMM reads every row of dqkv)
    bf16* rowk = dqkv + ((long)i * L + krow) * ld + (long)H * Dh + (long)h * Dh;
    bf16* rowv = rowk + (long)H * Dh;
#pragma unroll
    for (int d = 0; d < NS; ++d)
      if (d * 16 + 4 * lg < Dh) {
        uint2 a, b;
        a.x = pack_bf2(dk[d][0] * scale, dk[d][1] * scale);
        a.y = pack_bf2(dk[d][2] * scale, dk[d][3] * scale);
        b.x = pack_bf2(dv[d][0], dv[d][1]);
        b.y = pack_bf2(dv[d][2], dv[d][3]);
        *reinterpret_cast<uint2*>(rowk + d * 16 + 4 * lg) = a;
        *reinterpret_cast<uint2*>(rowv + d * 16 + 4 * lg) = b;
      }
  }
  if (dbias) {
#pragma unroll
    for (int d = 0; d < NS; ++d)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float tk = rowsum16(dk[d][r]), tv = rowsum16(dv[d][r]);
        const int dd = d * 16 + 4 * lg + r;
        if (lr == 0 && dd < Dh) {
          atomicAdd(dbias + ((long)i * 3 * H + H + h) * Dh + dd, tk * scale);
          atomicAdd(dbias + ((long)i * 3 * H + 2 * H + h) * Dh + dd, tv);
        }
      }
  }
}

// --------------------------------------------------------------- MAP head ----
// One probe query per (sample, head) (models/vit.py:168-183): a wave per pair, VALU dot products.
__global__ __launch_bounds__(256) void adh_map_fwd_kernel(const bf16* __restrict__ q, const bf16* __restrict__ kv,
                                                          bf16* __restrict__ o, float* __restrict__ p,
                                                          const int* __restrict__ kv_len, int n, int L, int H,
                                                          int Dh, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sp = reinterpret_cast<float*>(smem) + (threadIdx.x >> 6) * L;
  const int lane = threadIdx.x & 63;
  const long pair = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= (long)n * H) return;
  const long i = pair / H;
  const int h = (int)(pair - i * H);
  const int Lk = kv_len ? max(1, min(kv_len[i], L)) : L;
  const long ld = 2L * H * Dh;
  const bf16* kb_ = kv + i * L * ld + (long)h * Dh;
  const bf16* vb_ = kb_ + (long)H * Dh;
  const bf16* qp = q + (i * H + h) * Dh;
  float mx = -INFINITY;
  for (int l = lane; l < L; l += 64) {
    float acc = 0.f;
    for (int d = 0; d < Dh; d += 8) {
      const uint4 a = *reinterpret_cast<const uint4*>(kb_ + (long)l * ld + d);
      const uint4 b = *reinterpret_cast<const uint4*>(qp + d);
      acc += bflo(a.x) * bflo(b.x) + bfhi(a.x) * bfhi(b.x) + bflo(a.y) * bflo(b.y) + bfhi(a.y) * bfhi(b.y) +
             bflo(a.z) * bflo(b.z) + bfhi(a.z) * bfhi(b.z) + bflo(a.w) * bflo(b.w) + bfhi(a.w) * bfhi(b.w);
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
  for (int d = lane; d < Dh; d += 64) {
    float acc = 0.f;
    for (int l = 0; l < L; ++l) acc += sp[l] * bf2f(vb_[(long)l * ld + d]);
    o[(i * H + h) * Dh + d] = f2bf(acc);
  }
}

__global__ __launch_bounds__(256) void adh_map_bwd_kernel(const bf16* __restrict__ q, const bf16* __restrict__ kv,
                                                          const float* __restrict__ p, const bf16* __restrict__ d_o,
                                                          bf16* __restrict__ dq, bf16* __restrict__ dkv, int n,
                                                          int L, int H, int Dh, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sp = reinterpret_cast<float*>(smem) + (threadIdx.x >> 6) * 2 * L;  // p
  float* sd = sp + L;                                                       // ds
  const int lane = threadIdx.x & 63;
  const long pair = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= (long)n * H) return;
  const long i = pair / H;
  const int h = (int)(pair - i * H);
  const long ld = 2L * H * Dh;
  const bf16* kb_ = kv + i * L * ld + (long)h * Dh;
  const bf16* vb_ = kb_ + (long)H * Dh;
  bf16* dkb_ = dkv + i * L * ld + (long)h * Dh;
  bf16* dvb_ = dkb_ + (long)H * Dh;
  const bf16* gp = d_o + (i * H + h) * Dh;
  const float* prow = p + (i * H + h) * L;
  float dsum = 0.f;
  for (int l = lane; l < L; l += 64) {
    float acc = 0.f;
    for (int d = 0; d < Dh; d += 8) {
      const uint4 a = *reinterpret_cast<const uint4*>(vb_ + (long)l * ld + d);
      const uint4 b = *reinterpret_cast<const uint4*>(gp + d);
      acc += bflo(a.x) * bflo(b.x) + bfhi(a.x) * bfhi(b.x) + bflo(a.y) * bflo(b.y) + bfhi(a.y) * bfhi(b.y) +
             bflo(a.z) * bflo(b.z) + bfhi(a.z) * bfhi(b.z) + bflo(a.w) * bflo(b.w) + bfhi(a.w) * bfhi(b.w);
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
  for (int d = lane; d < Dh; d += 64) {
    const float g = bf2f(gp[d]);
    const float qd = bf2f(q[(i * H + h) * Dh + d]) * scale;
    float dqa = 0.f;
    for (int l = 0; l < L; ++l) {
      const float ds = sd[l];
      dvb_[(long)l * ld + d] = f2bf(sp[l] * g);
      dkb_[(long)l * ld + d] = f2bf(ds * qd);
      dqa += ds * bf2f(kb_[(long)l * ld + d]);
    }
    dq[(i * H + h) * Dh + d] = f2bf(dqa * scale);
  }
}

template <int NS>
int launch_fwd(const void* qkv, void* o, float* lse, const int* kv_len, int n, int L, int H, int Dh, float scale,
               hipStream_t s) {
  hipLaunchKernelGGL((adh_fwd_kernel<NS>), dim3(n * H, (L + 63) / 64), dim3(256), 0, s, (const bf16*)qkv, (bf16*)o,
                     lse, kv_len, L, H, Dh, scale);
  return bv_check_launch("bv_attn_fwd_dh");
}
template <int NS>
int launch_bwd(const void* qkv, const void* d_o, const float* lse, const int* kv_len, float* delta, void* dqkv,
               float* dbias, int n, int L, int H, int Dh, float scale, hipStream_t s) {
  hipLaunchKernelGGL((adh_bwd_dq_kernel<NS>), dim3(n * H, (L + 63) / 64), dim3(256), 0, s, (const bf16*)qkv,
                     (const bf16*)d_o, lse, delta, (bf16*)dqkv, dbias, kv_len, L, H, Dh, scale);
  int rc = bv_check_launch("bv_attn_bwd_dh(dq)");
  if (rc) return rc;
  hipLaunchKernelGGL((adh_bwd_dkv_kernel<NS>), dim3(n * H, (L + 63) / 64), dim3(256), 0, s, (const bf16*)qkv,
                     (const bf16*)d_o, lse, (const float*)delta, (bf16*)dqkv, dbias, kv_len, L, H, Dh, scale);
  return bv_check_launch("bv_attn_bwd_dh(dkv)");
}

}  // namespace

#define BV_ADH_DISPATCH(FN, ...)                            \
  switch ((Dh + 15) / 16) {                                 \
    case 1: return FN<1>(__VA_ARGS__);                      \
    case 2: return FN<2>(__VA_ARGS__);                      \
    case 3: return FN<3>(__VA_ARGS__);                      \
    case 4: return FN<4>(__VA_ARGS__);                      \
    case 5: return FN<5>(__VA_ARGS__);                      \
    case 6: return FN<6>(__VA_ARGS__);                      \
    case 7: return FN<7>(__VA_ARGS__);                      \
    default: return FN<8>(__VA_ARGS__);                     \
  }

// qkv [n*L][3][H][Dh] bf16 -> o [n*L][H][Dh] bf16, lse [n][H][L] fp32 (natural log).  kv_len optional
// (int32 [n], valid keys per sample, >= 1).  Dh % 8 == 0, Dh <= 128; any L.
extern "C" int bv_attn_fwd_dh(const void* qkv, void* o, float* lse, const int* kv_len, int n, int L, int H, int Dh,
                              void* stream) {
  BV_REQUIRE(n > 0 && L > 0 && H > 0, "bv_attn_fwd_dh: bad shape n=%d L=%d H=%d", n, L, H);
  BV_REQUIRE(Dh >= 8 && Dh <= 128 && Dh % 8 == 0, "bv_attn_fwd_dh: head dim %d must be a multiple of 8, <= 128", Dh);
  BV_REQUIRE((uintptr_t)qkv % 16 == 0 && (uintptr_t)o % 8 == 0, "bv_attn_fwd_dh: unaligned pointers");
  const float scale = 1.0f / sqrtf((float)Dh);
  hipStream_t s = (hipStream_t)stream;
  BV_ADH_DISPATCH(launch_fwd, qkv, o, lse, kv_len, n, L, H, Dh, scale, s)
}

// dqkv [n*L][3][H][Dh] bf16 (every row written), delta [n][H][L] fp32 scratch/output, dbias_rows optional
// [n][3][H][Dh] fp32: per-sample column sums of dq / dk / dv (zeroed here, then accumulated).
extern "C" int bv_attn_bwd_dh(const void* qkv, const void* d_o, const float* lse, const int* kv_len, float* delta,
                              void* dqkv, float* dbias_rows, int n, int L, int H, int Dh, void* stream) {
  BV_REQUIRE(n > 0 && L > 0 && H > 0, "bv_attn_bwd_dh: bad shape n=%d L=%d H=%d", n, L, H);
  BV_REQUIRE(Dh >= 8 && Dh <= 128 && Dh % 8 == 0, "bv_attn_bwd_dh: head dim %d must be a multiple of 8, <= 128", Dh);
  const float scale = 1.0f / sqrtf((float)Dh);
  hipStream_t s = (hipStream_t)stream;
  if (dbias_rows) {
    hipError_t e = hipMemsetAsync(dbias_rows, 0, (size_t)n * 3 * H * Dh * sizeof(float), s);
    BV_REQUIRE(e == hipSuccess, "bv_attn_bwd_dh: memset failed: %s", hipGetErrorString(e));
  }
  BV_ADH_DISPATCH(launch_bwd, qkv, d_o, lse, kv_len, delta, dqkv, dbias_rows, n, L, H, Dh, scale, s)
}

// MAP head (models/vit.py:168-183): q [n][H][Dh] (the probe's projection), kv [n*L][2][H][Dh] ->
// o [n][H][Dh], p [n][H][L] fp32 probabilities (saved for the backward).
extern "C" int bv_map_attn_fwd_dh(const void* q, const void* kv, void* o, float* p, const int* kv_len, int n, int L,
                                  int H, int Dh, void* stream) {
  BV_REQUIRE(n > 0 && L > 0 && H > 0 && L <= 4096, "bv_map_attn_fwd_dh: bad shape n=%d L=%d H=%d", n, L, H);
  BV_REQUIRE(Dh >= 8 && Dh <= 128 && Dh % 8 == 0, "bv_map_attn_fwd_dh: head dim %d must be a multiple of 8, <= 128", Dh);
  const long pairs = (long)n * H;
  hipLaunchKernelGGL(adh_map_fwd_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 4 * L * sizeof(float),
                     (hipStream_t)stream, (const bf16*)q, (const bf16*)kv, (bf16*)o, p, kv_len, n, L, H, Dh,
                     1.0f / sqrtf((float)Dh));
  return bv_check_launch("bv_map_attn_fwd_dh");
}
extern "C" int bv_map_attn_bwd_dh(const void* q, const void* kv, const float* p, const void* d_o, void* dq,
                                  void* dkv, int n, int L, int H, int Dh, void* stream) {
  BV_REQUIRE(n > 0 && L > 0 && H > 0 && L <= 4096, "bv_map_attn_bwd_dh: bad shape n=%d L=%d H=%d", n, L, H);
  BV_REQUIRE(Dh >= 8 && Dh <= 128 && Dh % 8 == 0, "bv_map_attn_bwd_dh: head dim %d must be a multiple of 8, <= 128", Dh);
  const long pairs = (long)n * H;
  hipLaunchKernelGGL(adh_map_bwd_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), 8 * L * sizeof(float),
                     (hipStream_t)stream, (const bf16*)q, (const bf16*)kv, p, (const bf16*)d_o, (bf16*)dq, (bf16*)dkv,
                     n, L, H, Dh, 1.0f / sqrtf((float)Dh));
  return bv_check_launch("bv_map_attn_bwd_dh");
}
