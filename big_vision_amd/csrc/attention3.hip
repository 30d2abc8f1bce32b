// Fused self-attention forward / backward for gfx950, Dh = 64, L <= 576: third generation of the
// LDS-resident kernels.  Same mathematics and reference call sites as its predecessor attention2.hip (rounds 2-4, git history)
// (flax nn.MultiHeadDotProductAttention inside big_vision/models/vit.py:93-98, text tower via
// vit.Encoder, models/proj/image_text/text_transformer.py:72-75; backward = jax.value_and_grad,
// trainers/proj/image_text/siglip.py:311):
//   S = (q/sqrt(Dh)) k^T,  P = softmax_rows(S),  O = P v
//   dV = P^T dO, dP = dO V^T, dS = P o (dP - delta), dQ = dS K/sqrt(Dh), dK = dS^T Q/sqrt(Dh)
// plus an optional KEY-PADDING length per sample (kv_len[i] <= L: keys >= kv_len[i] are masked,
// the NaFlex path, models/proj/image_text/naflex_vit.py:84-293 passes mask = valid patches).
//
// Why a third version.  attention2 (4 waves, 2 query fragments per iteration, 8 waves per CU) ran at
// 3.1 TB/s and 12 % of the MFMA rate: neither roof.  rocprof + arithmetic: per workgroup ~6 k
// cycles of staging and ~14 k cycles per query block, of which the MFMAs are 1.8 k - the rest is
// exposed latency (the block's Q fragments are fetched from HBM after the previous block's stores,
// every LDS fragment read is waited for by the only wave that could hide it).  The shape is
// bandwidth/latency-bound (97 FLOP per HBM byte at L = 196), so the lever is memory-level
// parallelism, not MFMA efficiency:
//   * ONE query fragment per wave iteration: a third of the registers (the score row of a
//     fragment is KF x 4 VGPRs), 8 waves per workgroup and 2 workgroups per CU -> 16 waves per CU
//     hide each other's LDS / HBM latencies; each LDS fragment feeds 1 MFMA instead of 2, which
//     the LDS affords (65 MB per CU and launch at 256 B/clk = 0.12 ms, the HBM floor is 0.42 ms);
//   * the NEXT query fragment's global loads are issued before the current one is computed, the
//     first one before the K/V staging;
//   * key fragments = ceil(L/16) exactly (13 for 196/197 tokens, not 14): the odd last fragment
//     of the P V / dS K / P^T dO contractions uses the K = 16 MFMA;
//   * delta = rowsum(P o dP) is computed EXACTLY from the fp32 P and dP of the row (a second
//     sweep over the keys recomputes them; MFMAs are free here) instead of rowsum(dO o O) with the
//     bf16-rounded O: that shortcut costs the q/k gradients of nearly-uniform attention rows
//     (random init, repeated tokens) a relative error of 5-10 % at small batch, where dP - delta
//     cancels to a fraction of delta; the backward no longer reads O at all.
// MFMA plan as attention2: S^T[key][q] = K Q^T and dP^T = V dO^T put P^T / dS^T straight into the
// B-operand layout of O^T += V^T P^T and dQ^T += K^T dS^T; the key-owned pass computes S = Q K^T,
// dP = dO V^T and accumulates dV^T += dO^T P, dK^T += Q^T dS.
#include <type_traits>
#include "attn_common.h"
#include "bvhip_internal.h"
// scheduling fences of the forward loops at 3 workgroups per CU (168 VGPRs): one behind every A3_SB_S-th score
// fragment / every A3_SB_PV-th pair of P V fragments bounds how many LDS operand reads hipcc hoists (0 = none)
#ifndef A3_PIPE
#define A3_PIPE 1   // 1: software-pipelined S / P V loops of the 3-workgroups-per-CU forward (0: the plain loops, A/B)
#endif
#ifndef A3_PIPE_LONG
#define A3_PIPE_LONG 1   // 1 (ships since round 6): the K-row prefetch of that pipeline also in the one-workgroup-per-CU forward of the
#endif                   //    long sequences (28 / 36 key fragments, 2 waves per SIMD): bit-identical, L = 441: 441 -> 425 us at n = 256,
                         //    L = 576: 678 -> 657 (profiles/r06_attn_long_prefetch_ab.txt); 2 = the P V loop too (spills at 256 VGPRs)
#ifndef A3_DQ_PIPE
#define A3_DQ_PIPE 0     // 1: the one-sweep dQ kernel of the long sequences requests the K / V rows of the next PAIR of key fragments
#endif                   //    before the current pair is computed: backward -2.4 % at L = 441, +0.9 % at L = 576 (same file), 72 of 10.8 M
                         //    gradient elements move by one bf16 ulp (contraction of the delta sum) - not adopted
#ifndef A4_DKV_PIPE
#define A4_DKV_PIPE 0    // 1: the 32-key-block dK / dV kernel requests the Q / dO rows of the NEXT query fragment before the current
#endif                   //    fragment's MFMAs / exponentials: bit-identical, +0.5..1 % SLOWER (same file) - not adopted
#ifndef A3_FWD_LONG_NW
#define A3_FWD_LONG_NW 8   // waves of the 28-fragment forward's workgroup: 8 (2 per SIMD, K-row prefetch) or 12 (3 per SIMD, plain loops,
#endif                     // 166 VGPRs): 12 is bit-identical and 2 % faster (433 vs 442 us, r06_attn_long_prefetch_ab.txt) - inside the spread, off
#ifndef A3_DQ_LONG_NW
#define A3_DQ_LONG_NW 16   // waves of the UNMASKED 28-fragment one-sweep dQ kernel: 16 (4 per SIMD: its 126 VGPRs fit the 128 line) ships since
#endif                     // round 6 - bit-identical, backward at L = 441 1424-1430 -> 1384-1400 us (same file); 8 = the former 2 per SIMD
#ifndef A3_SB_S
#define A3_SB_S 4
#endif
#ifndef A3_SB_PV
#define A3_SB_PV 2
#endif
#ifndef A3_PROBE
#define A3_PROBE 0   // != 0 only in tools/probes/attn3_probe.hip (ablations of the forward kernel)
#endif
#if A3_PROBE == 3    // s_memtime stamps of waves 0 and NW-1 of a few mid-launch workgroups
__device__ long* g_a3_stamps;
#define A3_STAMP(k)                                                                              \
  do {                                                                                           \
    if (lane == 0 && (wave == 0 || wave == NW - 1) && blockIdx.x >= 12000 && blockIdx.x < 12004) \
      g_a3_stamps[((blockIdx.x - 12000) * 2 + (wave != 0)) * 32 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define A3_STAMP(k)
#endif

namespace {
using namespace bvattn;

// hipcc pads the MFMA-result -> VALU-read hazard inside a basic block; across a taken branch right
// behind the MFMA it was seen to pad nothing (see sdp below).  Wherever the last MFMAs of a loop
// feed VALU code behind control flow, the wait states are spelled out.
__device__ __forceinline__ void mfma_drain(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  // the operands pin the wait states BEHIND the MFMAs that produce them
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

// ------------------------------------------------------------------ forward --
// TAIL: the host guarantees Lk > (KF - 1) * 16 for every sample (no kv_len, L in the last fragment), so
// only the last key fragment is masked.  (With a runtime straddling fragment hipcc if-converts the
// mask of ALL KF x 4 scores: 52 compares hoisted out of the loop into SGPR pairs that spill to VGPR
// lanes, 107 v_cndmask + 50 v_readlane per query fragment - a third of the loop's VALU work.)
template <int KF, int NW, int WPS, bool TAIL = false>
__global__ __launch_bounds__(NW * 64, WPS) void attn3_fwd_kernel(const bf16* __restrict__ qkv,
                                                                 bf16* __restrict__ o,
                                                                 float* __restrict__ lse,
                                                                 const int* __restrict__ kv_len, int L,
                                                                 int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kt = smem;
  char* Vt = smem + KF * 16 * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int i = blockIdx.x / H, h = blockIdx.x % H;
  const int Lk = kv_len ? min(kv_len[i], L) : L;
  const long ld = 3L * H * DH;
  const bf16* qb_ = qkv + (long)i * L * ld + h * DH;
  const bf16* kb_ = qb_ + (long)H * DH;
  const bf16* vb_ = qb_ + 2L * H * DH;
  int qf = wave;
  A3_STAMP(0);
  bf16x8 q0 = gfrag(qb_, ld, qf * 16 + lr, L, lg * 8), q1 = gfrag(qb_, ld, qf * 16 + lr, L, 32 + lg * 8);
#if A3_PROBE == 2   // probe: no K/V staging (LDS holds garbage)
#else
  t64_stage2<KF * 16, NW * 64>(Kt, kb_, ld, Vt, vb_, ld, Lk, tid);
#endif
  __syncthreads();
  A3_STAMP(1);
  const float c = scale * LOG2E;
#if A3_PROBE == 1     // probe: staging only
  if (L > 0) return;
#endif
  int it_ = 0;
  // (software pipeline of the S loop, below: the first pair of key fragments of an iteration is requested at the end of
  //  the previous iteration's S loop, the first V operands of the P V loop in front of the softmax)
  // groups of two key fragments, the last one of three when KF is odd: KF / 2 groups, an EVEN number for every
  // instantiation that runs this path (6, 14, 18), so the last group of an iteration works from register set 1 and
  // set 0 is free for the next iteration's first group
  constexpr int NG = KF / 2;
  constexpr bool PIPE = A3_PIPE && ((WPS == 3 && KF <= 17) || (A3_PIPE_LONG && KF >= 28 && WPS == 2));        // the S loop
  constexpr bool PIPE_PV = A3_PIPE && ((WPS == 3 && KF <= 17) || (A3_PIPE_LONG == 2 && KF >= 28 && WPS == 2));   // the P V loop too (A3_PIPE_LONG = 2 spills: 256 VGPRs)
  static_assert(!PIPE || (NG % 2 == 0 && NG >= 2), "attn3_fwd: the K prefetch ping-pong needs an even group count");
  bf16x8 kk[2][3][2];
  auto ldk = [&](int f, bf16x8 (&k)[2]) __attribute__((always_inline)) {
    k[0] = t64_row(Kt, f * 16 + lr, lg);
    k[1] = t64_row(Kt, f * 16 + lr, 4 + lg);
  };
  auto ldg = [&](int g, bf16x8 (&k)[3][2]) __attribute__((always_inline)) {   // group g: fragments 2 g, 2 g + 1 (, KF - 1)
    ldk(2 * g, k[0]);
    ldk(2 * g + 1, k[1]);
    if ((KF & 1) && g == NG - 1) ldk(KF - 1, k[2]);
  };
  if constexpr (PIPE) ldg(0, kk[0]);

  for (; qf * 16 < L; qf += NW, ++it_) {
    A3_STAMP(2 + it_ * 5);
    f32x4 s[KF];
    bf16x8 vv[2][4];
    auto ldv = [&](int fp, bf16x8 (&v)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int d = 0; d < 4; ++d) v[d] = t64_trpair(Vt, (2 * fp) * 16 + 4 * lg, (2 * fp + 1) * 16 + 4 * lg, d, lr);
    };
    if constexpr (PIPE) {
      // software pipeline over PAIRS of key fragments: the K rows of the next pair are requested from the LDS before the
      // MFMAs of the current pair are issued (two register sets, ping-pong; 3 workgroups per CU leave 168 VGPRs).  In
      // the plain loop below every group of MFMAs waits for its own operands' LDS round trip with only the two other
      // waves of the SIMD to cover it.
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int c_ = g & 1;
        if (g + 1 < NG) {
          ldg(g + 1, kk[c_ ^ 1]);
        } else {               // last group (set 1): the NEXT iteration's first group into set 0, and this
          ldg(0, kk[0]);       // iteration's first V operands
          if constexpr (PIPE_PV) ldv(0, vv[0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
        s[2 * g] = mfma16(kk[c_][0][1], q1, mfma16(kk[c_][0][0], q0, z));
        s[2 * g + 1] = mfma16(kk[c_][1][1], q1, mfma16(kk[c_][1][0], q0, z));
        if ((KF & 1) && g == NG - 1) s[KF - 1] = mfma16(kk[c_][2][1], q1, mfma16(kk[c_][2][0], q0, z));
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int f = 0; f < KF; ++f) {
      const bf16x8 k0 = t64_row(Kt, f * 16 + lr, lg);
      const bf16x8 k1 = t64_row(Kt, f * 16 + lr, 4 + lg);
      f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
      a = mfma16(k0, q0, a);
      a = mfma16(k1, q1, a);
      s[f] = a;
      // bound load hoisting (register pressure): 2 fragments' operands in flight at 128 VGPRs, 4 at 168
      if (KF > 13 || (WPS >= 4 ? (f & 1) : (A3_SB_S > 0 && f % A3_SB_S == A3_SB_S - 1))) __builtin_amdgcn_sched_barrier(0);
    }
    }
    // the NEXT fragment of this wave goes straight into the registers the scores no longer need
    // (rows >= L load nothing); its latency hides behind the softmax and the P V products
    const int qrow = qf * 16 + lr;
    {
      const int qn = (qf + NW) * 16 + lr;
      q0 = gfrag(qb_, ld, qn, L, lg * 8);
      q1 = gfrag(qb_, ld, qn, L, 32 + lg * 8);
    }
    A3_STAMP(3 + it_ * 5);
    float mx = -INFINITY;
    if constexpr (TAIL) {
      const int lim = Lk - (KF - 1) * 16 - lg * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) s[KF - 1][r] = r < lim ? s[KF - 1][r] : -INFINITY;
    }
#pragma unroll
    for (int f = 0; f < KF; ++f) {
      if constexpr (!TAIL) {
        if (f * 16 + 16 > Lk) {   // wave-uniform: fragment straddles / lies beyond the last valid key
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (f * 16 + lg * 4 + r >= Lk) s[f][r] = -INFINITY;
        }
      }
      if constexpr (TAIL) {
        mx = fmaxf(fmaxf(mx, s[f][0]), s[f][1]);   // v_max3_f32
        mx = fmaxf(fmaxf(mx, s[f][2]), s[f][3]);
      } else {
        mx = fmaxf(mx, fmaxf(fmaxf(s[f][0], s[f][1]), fmaxf(s[f][2], s[f][3])));
      }
    }
    mx = xmax4(mx);
    const float mc = mx * c;
    float sum;
    if constexpr (TAIL) {
      // packed fp32 (v_pk_fma_f32 / v_pk_add_f32: two elements per VALU slot) around the exp2
      const f32x2 c2 = f32x2{c, c}, nmc2 = f32x2{-mc, -mc};
      f32x2 sum2 = f32x2{0.f, 0.f};
#pragma unroll
      for (int f = 0; f < KF; ++f)
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
          const f32x2 a = __builtin_elementwise_fma(f32x2{s[f][2 * hp], s[f][2 * hp + 1]}, c2, nmc2);
          const f32x2 p = f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
          s[f][2 * hp] = p[0];
          s[f][2 * hp + 1] = p[1];
          sum2 += p;
        }
      sum = sum2[0] + sum2[1];
    } else {
      sum = 0.f;
#pragma unroll
      for (int f = 0; f < KF; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[f][r], c, -mc));
          s[f][r] = p;
          sum += p;
        }
    }
    sum = xsum4(sum);
    const float inv = 1.0f / sum;
    const float lsev = mx * scale + __logf(sum);
    f32x4 oa[4];
    A3_STAMP(4 + it_ * 5);
#pragma unroll
    for (int d = 0; d < 4; ++d) oa[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (PIPE_PV) {
      // the same for the P V products: the transposed V operands of the next pair of key fragments are in flight
      // while the current pair's four MFMAs issue
#pragma unroll
      for (int fp = 0; fp < KF / 2; ++fp) {
        const int c_ = fp & 1;
        if (fp + 1 < KF / 2) ldv(fp + 1, vv[c_ ^ 1]);
        const bf16x8 pf = pack8(s[2 * fp], s[2 * fp + 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < 4; ++d) oa[d] = mfma16(vv[c_][d], pf, oa[d]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int fp = 0; fp < KF / 2; ++fp) {
      const bf16x8 pf = pack8(s[2 * fp], s[2 * fp + 1]);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const bf16x8 vf = t64_trpair(Vt, (2 * fp) * 16 + 4 * lg, (2 * fp + 1) * 16 + 4 * lg, d, lr);
        oa[d] = mfma16(vf, pf, oa[d]);
      }
      if (WPS >= 4 || (A3_SB_PV > 0 && fp % A3_SB_PV == A3_SB_PV - 1)) __builtin_amdgcn_sched_barrier(0);
    }
    }
    if constexpr (KF & 1) {
      const s16x4 pf = pack4(s[KF - 1]);
#pragma unroll
      for (int d = 0; d < 4; ++d) oa[d] = mfma16k16(t64_tr(Vt, (KF - 1) * 16 + 4 * lg, d, lr), pf, oa[d]);
    }
    mfma_drain(oa[0], oa[1], oa[2], oa[3]);
    A3_STAMP(5 + it_ * 5);
    bf16* orow = o + ((long)i * L + qrow) * H * DH + h * DH;
    if constexpr (KF == 17 && !TAIL) {
      // (the masked 17-fragment instantiation sits at its 128-VGPR line: with the lane swaps of store_ot_rows hipcc
      //  spills 3-9 registers inside the loop; it keeps the 8-byte pieces)
      if (qrow < L) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          uint2 w;
          w.x = pack_bf2(oa[d][0] * inv, oa[d][1] * inv);
          w.y = pack_bf2(oa[d][2] * inv, oa[d][3] * inv);
          *reinterpret_cast<uint2*>(orow + d * 16 + lg * 4) = w;
        }
      }
    } else {
      store_ot_rows(orow, oa, inv, lg, qrow < L);   // 16-byte pieces: 16 rows x 64 B per store instruction
    }
    if (qrow < L && lg == 0) lse[((long)i * H + h) * L + qrow] = lsev;
    A3_STAMP(6 + it_ * 5);
  }
}

// ------------------------------------------------------------- backward: dQ --
template <int KF, int NW, int WPS>
__global__ __launch_bounds__(NW * 64, WPS) void attn3_bwd_dq_kernel(const bf16* __restrict__ qkv,
                                                                    const bf16* __restrict__ d_o,
                                                                    const float* __restrict__ lse,
                                                                    float* __restrict__ delta,
                                                                    bf16* __restrict__ dqkv,
                                                                    float* __restrict__ dbias,
                                                                    const int* __restrict__ kv_len, int L,
                                                                    int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kt = smem;
  char* Vt = smem + KF * 16 * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int i = blockIdx.x / H, h = blockIdx.x % H;
  const int Lk = kv_len ? min(kv_len[i], L) : L;
  const long ld = 3L * H * DH, ldo = (long)H * DH;
  const bf16* qb_ = qkv + (long)i * L * ld + h * DH;
  const bf16* kb_ = qb_ + (long)H * DH;
  const bf16* vb_ = qb_ + 2L * H * DH;
  const bf16* dob_ = d_o + (long)i * L * ldo + h * DH;
  const float* lse_ = lse + ((long)i * H + h) * L;
  float* red = reinterpret_cast<float*>(smem + 2 * KF * 16 * 128);   // [NW waves][64] column sums of dQ
  int qf = wave;
  bf16x8 q0 = gfrag(qb_, ld, qf * 16 + lr, L, lg * 8), q1 = gfrag(qb_, ld, qf * 16 + lr, L, 32 + lg * 8);
  bf16x8 g0 = gfrag(dob_, ldo, qf * 16 + lr, L, lg * 8), g1 = gfrag(dob_, ldo, qf * 16 + lr, L, 32 + lg * 8);
  float lse2 = qf * 16 + lr < L ? lse_[qf * 16 + lr] * LOG2E : INFINITY;
  t64_stage2<KF * 16, NW * 64>(Kt, kb_, ld, Vt, vb_, ld, Lk, tid);
  if (dbias && tid < NW * 64) red[tid] = 0.f;
  __syncthreads();
  const float c = scale * LOG2E;

  // S^T and dP^T fragment f of the current query fragment; p = P^T, dp = dP^T (keys >= Lk: p = 0)
  auto sdp = [&](int f, f32x4& p, f32x4& dp) {
    const bf16x8 k0 = t64_row(Kt, f * 16 + lr, lg), k1 = t64_row(Kt, f * 16 + lr, 4 + lg);
    const bf16x8 v0 = t64_row(Vt, f * 16 + lr, lg), v1 = t64_row(Vt, f * 16 + lr, 4 + lg);
    f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f};
    dp = f32x4{0.f, 0.f, 0.f, 0.f};
    st = mfma16(k0, q0, st);
    st = mfma16(k1, q1, st);
    dp = mfma16(v0, g0, dp);
    dp = mfma16(v1, g1, dp);
    // keys >= Lk: branch-free select.  (A wave-uniform `if (fragment straddles Lk)` around the mask put
    // the consumers of dp - the delta FMAs - first in the block after a loop back-edge, straight
    // behind the MFMA that writes dp in the predecessor block; hipcc (ROCm 7.2) emitted no wait
    // states on that edge and the first two FMAs read stale registers: delta was wrong by 30 %.)
    const int lim = Lk - f * 16 - lg * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c, -lse2));
      p[r] = r < lim ? e : 0.f;
    }
  };

  for (; qf * 16 < L; qf += NW) {
    const int qn = (qf + NW) * 16 + lr;
    const bf16x8 nq0 = gfrag(qb_, ld, qn, L, lg * 8), nq1 = gfrag(qb_, ld, qn, L, 32 + lg * 8);
    const bf16x8 ng0 = gfrag(dob_, ldo, qn, L, lg * 8), ng1 = gfrag(dob_, ldo, qn, L, 32 + lg * 8);
    const float nlse = qn < L ? lse_[qn] * LOG2E : INFINITY;
    // ---- sweep 1: delta = sum_keys P o dP of this query row (exact: fp32 P and dP)
    float dacc = 0.f;
#pragma unroll 1
    for (int f = 0; f < KF; ++f) {
      f32x4 p, dp;
      sdp(f, p, dp);
#pragma unroll
      for (int r = 0; r < 4; ++r) dacc = __builtin_fmaf(p[r], dp[r], dacc);
    }
    const float del = xsum4(dacc);
    const int qrow = qf * 16 + lr;
    if (lg == 0 && qrow < L) delta[((long)i * H + h) * L + qrow] = del;
    // ---- sweep 2: dS^T = P^T o (dP^T - delta), dQ^T += K^T dS^T
    f32x4 dq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int fp = 0; fp < KF / 2; ++fp) {
      f32x4 ds[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        f32x4 p, dp;
        sdp(2 * fp + e, p, dp);
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[e][r] = p[r] * (dp[r] - del);
      }
      const bf16x8 dsf = pack8(ds[0], ds[1]);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const bf16x8 kt = t64_trpair(Kt, (2 * fp) * 16 + 4 * lg, (2 * fp + 1) * 16 + 4 * lg, d, lr);
        dq[d] = mfma16(kt, dsf, dq[d]);
      }
    }
    if constexpr (KF & 1) {
      f32x4 p, dp, ds;
      sdp(KF - 1, p, dp);
#pragma unroll
      for (int r = 0; r < 4; ++r) ds[r] = p[r] * (dp[r] - del);
      const s16x4 dsf = pack4(ds);
#pragma unroll
      for (int d = 0; d < 4; ++d) dq[d] = mfma16k16(t64_tr(Kt, (KF - 1) * 16 + 4 * lg, d, lr), dsf, dq[d]);
    }
    mfma_drain(dq[0], dq[1], dq[2], dq[3]);
    if (dbias) {
      // column sums of this fragment (fp32, before the bf16 rounding; padded query rows are exactly
      // 0: q = dO = 0, P = 0) into the wave's private LDS row
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = rowsum16(dq[d][r]);
          if (lr == 0) red[wave * 64 + d * 16 + lg * 4 + r] += t * scale;
        }
    }
    store_ot_rows(dqkv + ((long)i * L + qrow) * ld + h * DH, dq, scale, lg, qrow < L);   // 16-byte pieces (attn_common.h)
    q0 = nq0; q1 = nq1; g0 = ng0; g1 = ng1; lse2 = nlse;
  }
  if (dbias) {   // per-(sample, head) column sums of dQ -> dbias[i][0][h][:]; the host sums over samples
    __syncthreads();
    if (tid < 64) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red[w * 64 + tid];
      dbias[((long)i * 3 * H + h) * DH + tid] = t;
    }
  }
}

// ------------------------------------------- backward: dQ, one sweep (with O) --
// When the forward output O is at hand (the training step keeps it: it is the operand of the
// out-projection's dW), one sweep over the keys is enough and delta stays exact:
//   delta~_i = rowsum(dO_i o O_i)               (bf16 O: off by eps_i = delta_i - delta~_i, ~2^-9 |dO||O|)
//   dS~ = P o (dP - delta~)                      fp32, then bf16 for the MFMA - as accurate as dS itself
//   eps_i = rowsum_j dS~_ij = delta_i - delta~_i (rows of P sum to 1), fp32, exact
//   dQ_i = sum_j dS~_ij K_j - eps_i * sum_j P_ij K_j
// The correction term is second order (eps times a bf16-rounded P K), so nothing is lost against the
// two-sweep kernel above, which recomputes S, dP and the exponentials a second time: per key pair
// 16 MFMAs and ~40 VALU slots instead of 20 and ~110.  delta = delta~ + eps is written for the
// key-owned pass.  TAIL as in the forward: only the last key fragment can hold masked keys.
template <int KF, int NW, int WPS, bool TAIL>
__global__ __launch_bounds__(NW * 64, WPS) void attn3_bwd_dq1_kernel(const bf16* __restrict__ qkv,
                                                                     const bf16* __restrict__ o,
                                                                     const bf16* __restrict__ d_o,
                                                                     const float* __restrict__ lse,
                                                                     float* __restrict__ delta,
                                                                     bf16* __restrict__ dqkv,
                                                                     float* __restrict__ dbias,
                                                                     const int* __restrict__ kv_len, int L,
                                                                     int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Kt = smem;
  char* Vt = smem + KF * 16 * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int i = blockIdx.x / H, h = blockIdx.x % H;
  const int Lk = kv_len ? min(kv_len[i], L) : L;
  const long ld = 3L * H * DH, ldo = (long)H * DH;
  const bf16* qb_ = qkv + (long)i * L * ld + h * DH;
  const bf16* kb_ = qb_ + (long)H * DH;
  const bf16* vb_ = qb_ + 2L * H * DH;
  const bf16* dob_ = d_o + (long)i * L * ldo + h * DH;
  const bf16* ob_ = o + (long)i * L * ldo + h * DH;
  const float* lse_ = lse + ((long)i * H + h) * L;
  float* red = reinterpret_cast<float*>(smem + 2 * KF * 16 * 128);   // [NW waves][64] column sums of dQ
  int qf = wave;
  bf16x8 q0 = gfrag(qb_, ld, qf * 16 + lr, L, lg * 8), q1 = gfrag(qb_, ld, qf * 16 + lr, L, 32 + lg * 8);
  bf16x8 g0 = gfrag(dob_, ldo, qf * 16 + lr, L, lg * 8), g1 = gfrag(dob_, ldo, qf * 16 + lr, L, 32 + lg * 8);
  bf16x8 o0 = gfrag(ob_, ldo, qf * 16 + lr, L, lg * 8), o1 = gfrag(ob_, ldo, qf * 16 + lr, L, 32 + lg * 8);
  float lse2 = qf * 16 + lr < L ? lse_[qf * 16 + lr] * LOG2E : INFINITY;
  t64_stage2<KF * 16, NW * 64>(Kt, kb_, ld, Vt, vb_, ld, Lk, tid);
  if (dbias && tid < NW * 64) red[tid] = 0.f;
  __syncthreads();
  const float c = scale * LOG2E;

  for (; qf * 16 < L; qf += NW) {
    // delta~ of this lane's query row (the row's 64 products are spread over its 4 lane groups)
    float dt = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      dt = __builtin_fmaf((float)g0[e], (float)o0[e], dt);
      dt = __builtin_fmaf((float)g1[e], (float)o1[e], dt);
    }
    dt = xsum4(dt);
    const int qn = (qf + NW) * 16 + lr;
    o0 = gfrag(ob_, ldo, qn, L, lg * 8);      // next fragment's O goes into the registers just consumed
    o1 = gfrag(ob_, ldo, qn, L, 32 + lg * 8);
    const f32x2 c2 = f32x2{c, c}, nl2 = f32x2{-lse2, -lse2}, dt2 = f32x2{dt, dt};
    f32x2 eps2 = f32x2{0.f, 0.f};

    // P^T and dS~^T of key fragment f for this query fragment (MASK: keys >= Lk get p = 0)
    // (the K / V rows of fragment f: rows[0..1] = K halves, rows[2..3] = V halves)
    auto ldrows = [&](int f, bf16x8 (&rows)[4]) __attribute__((always_inline)) {
      rows[0] = t64_row(Kt, f * 16 + lr, lg); rows[1] = t64_row(Kt, f * 16 + lr, 4 + lg);
      rows[2] = t64_row(Vt, f * 16 + lr, lg); rows[3] = t64_row(Vt, f * 16 + lr, 4 + lg);
    };
    auto pds_of = [&](int f, bf16x8 k0, bf16x8 k1, bf16x8 v0, bf16x8 v1, auto maskc, f32x4& p, f32x4& ds) __attribute__((always_inline)) {
      constexpr bool MASK = decltype(maskc)::value;
      f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
      st = mfma16(k0, q0, st);
      st = mfma16(k1, q1, st);
      dp = mfma16(v0, g0, dp);
      dp = mfma16(v1, g1, dp);
      const int lim = Lk - f * 16 - lg * 4;
#pragma unroll
      for (int hp = 0; hp < 2; ++hp) {
        const f32x2 a = __builtin_elementwise_fma(f32x2{st[2 * hp], st[2 * hp + 1]}, c2, nl2);
        f32x2 e = f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
        if constexpr (MASK) {   // branch-free (see the hazard note in the two-sweep kernel)
          e[0] = 2 * hp < lim ? e[0] : 0.f;
          e[1] = 2 * hp + 1 < lim ? e[1] : 0.f;
        }
        const f32x2 t = e * (f32x2{dp[2 * hp], dp[2 * hp + 1]} - dt2);
        eps2 += t;
        p[2 * hp] = e[0]; p[2 * hp + 1] = e[1];
        ds[2 * hp] = t[0]; ds[2 * hp + 1] = t[1];
      }
    };
    using NoMask = std::integral_constant<bool, !TAIL>;   // general path masks every fragment
    using Mask = std::integral_constant<bool, true>;

    f32x4 dq[4], bq[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
      bq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // (plain path: a fragment's rows are read right in front of its MFMAs, as before the operands became arguments)
    auto pds = [&](int f, auto maskc, f32x4& p, f32x4& ds) __attribute__((always_inline)) {
      const bf16x8 k0 = t64_row(Kt, f * 16 + lr, lg), k1 = t64_row(Kt, f * 16 + lr, 4 + lg);
      const bf16x8 v0 = t64_row(Vt, f * 16 + lr, lg), v1 = t64_row(Vt, f * 16 + lr, 4 + lg);
      pds_of(f, k0, k1, v0, v1, maskc, p, ds);
    };
    auto pair_tail = [&](int fp, f32x4 (&p)[2], f32x4 (&ds)[2]) __attribute__((always_inline)) {
      const bf16x8 dsf = pack8(ds[0], ds[1]), pf = pack8(p[0], p[1]);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const bf16x8 kt = t64_trpair(Kt, (2 * fp) * 16 + 4 * lg, (2 * fp + 1) * 16 + 4 * lg, d, lr);
        dq[d] = mfma16(kt, dsf, dq[d]);
        bq[d] = mfma16(kt, pf, bq[d]);
      }
    };
    auto pair_of = [&](int fp, const bf16x8 (&ra)[4], const bf16x8 (&rb)[4], auto mask_hi) __attribute__((always_inline)) {
      f32x4 p[2], ds[2];
      pds_of(2 * fp, ra[0], ra[1], ra[2], ra[3], NoMask{}, p[0], ds[0]);
      pds_of(2 * fp + 1, rb[0], rb[1], rb[2], rb[3], mask_hi, p[1], ds[1]);
      pair_tail(fp, p, ds);
    };
    auto pair = [&](int fp, auto mask_hi) __attribute__((always_inline)) {
      f32x4 p[2], ds[2];
      pds(2 * fp, NoMask{}, p[0], ds[0]);
      pds(2 * fp + 1, mask_hi, p[1], ds[1]);
      pair_tail(fp, p, ds);
    };
    constexpr int NPAIR = KF / 2, NLOOP = (KF & 1) ? NPAIR : NPAIR - 1;   // even KF: the last pair holds the tail
    // Long sequences (one workgroup per CU, 2 waves per SIMD, 256 VGPRs): two operand sets, ping-pong - the rows of the next
    // pair are requested from the LDS before the current pair's MFMAs / exponentials, which then cover the round trip (in the
    // plain loop every pair waits for its own operands with ONE other wave on the SIMD to fill the gap).
    constexpr bool DQ_PIPE = A3_DQ_PIPE && WPS == 2 && KF >= 28 && !(KF & 1) && (NLOOP & 1);
    if constexpr (DQ_PIPE) {
      bf16x8 a0[4], a1[4], b0[4], b1[4];
      ldrows(0, a0); ldrows(1, a1);
#pragma unroll 1
      for (int fp = 0; fp + 1 < NLOOP; fp += 2) {
        ldrows(2 * fp + 2, b0); ldrows(2 * fp + 3, b1);
        __builtin_amdgcn_sched_barrier(0);
        pair_of(fp, a0, a1, NoMask{});
        __builtin_amdgcn_sched_barrier(0);
        ldrows(2 * fp + 4, a0); ldrows(2 * fp + 5, a1);
        __builtin_amdgcn_sched_barrier(0);
        pair_of(fp + 1, b0, b1, NoMask{});
        __builtin_amdgcn_sched_barrier(0);
      }
      // (NLOOP odd: set a holds pair NLOOP - 1; the masked last pair follows)
      ldrows(2 * NPAIR - 2, b0); ldrows(2 * NPAIR - 1, b1);
      __builtin_amdgcn_sched_barrier(0);
      pair_of(NLOOP - 1, a0, a1, NoMask{});
      __builtin_amdgcn_sched_barrier(0);
      pair_of(NPAIR - 1, b0, b1, Mask{});
    } else {
#pragma unroll 1
    for (int fp = 0; fp < NLOOP; ++fp) pair(fp, NoMask{});
    }
    if constexpr (DQ_PIPE) {
    } else if constexpr (KF & 1) {
      f32x4 p, ds;
      pds(KF - 1, Mask{}, p, ds);
      const s16x4 dsf = pack4(ds), pf = pack4(p);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const s16x4 kt = t64_tr(Kt, (KF - 1) * 16 + 4 * lg, d, lr);
        dq[d] = mfma16k16(kt, dsf, dq[d]);
        bq[d] = mfma16k16(kt, pf, bq[d]);
      }
    } else {
      pair(NPAIR - 1, Mask{});
    }
    mfma_drain(dq[0], dq[1], dq[2], dq[3]);
    mfma_drain(bq[0], bq[1], bq[2], bq[3]);
    // the NEXT fragment's q / dO rows go into the registers of the finished one (no second set: the
    // kernel sits at the 128-VGPR line of 4 waves per SIMD); the epilogue below covers part of the latency
    q0 = gfrag(qb_, ld, qn, L, lg * 8); q1 = gfrag(qb_, ld, qn, L, 32 + lg * 8);
    g0 = gfrag(dob_, ldo, qn, L, lg * 8); g1 = gfrag(dob_, ldo, qn, L, 32 + lg * 8);
    lse2 = qn < L ? lse_[qn] * LOG2E : INFINITY;
    const float eps = xsum4(eps2[0] + eps2[1]);
    const int qrow = qf * 16 + lr;
    if (lg == 0 && qrow < L) delta[((long)i * H + h) * L + qrow] = dt + eps;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int r = 0; r < 4; ++r) dq[d][r] = __builtin_fmaf(-eps, bq[d][r], dq[d][r]);
    if (dbias) {
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = rowsum16(dq[d][r]);
          if (lr == 0) red[wave * 64 + d * 16 + lg * 4 + r] += t * scale;
        }
    }
    store_ot_rows(dqkv + ((long)i * L + qrow) * ld + h * DH, dq, scale, lg, qrow < L);   // 16-byte pieces (attn_common.h)
  }
  if (dbias) {
    __syncthreads();
    if (tid < 64) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red[w * 64 + tid];
      dbias[((long)i * 3 * H + h) * DH + tid] = t;
    }
  }
}

// --------------------------------------------------------- backward: dK, dV --
// QN = number of 16-row query fragments = ceil(L/16); one key fragment per wave iteration.
template <int QN, int NW, int WPS>
__global__ __launch_bounds__(NW * 64, WPS) void attn3_bwd_dkv_kernel(const bf16* __restrict__ qkv,
                                                                     const bf16* __restrict__ d_o,
                                                                     const float* __restrict__ lse,
                                                                     const float* __restrict__ delta,
                                                                     bf16* __restrict__ dqkv,
                                                                     float* __restrict__ dbias,
                                                                     const int* __restrict__ kv_len, int L,
                                                                     int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qt = smem;
  char* Gt = smem + QN * 16 * 128;
  float* lse_s = reinterpret_cast<float*>(smem + 2 * QN * 16 * 128);
  float* del_s = lse_s + QN * 16;
  float* red = del_s + QN * 16;   // [NW waves][2][64] column sums of dK / dV
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int i = blockIdx.x / H, h = blockIdx.x % H;
  const int Lk = kv_len ? min(kv_len[i], L) : L;
  const long ld = 3L * H * DH, ldo = (long)H * DH;
  const bf16* qb_ = qkv + (long)i * L * ld + h * DH;
  const bf16* kb_ = qb_ + (long)H * DH;
  const bf16* vb_ = qb_ + 2L * H * DH;
  const bf16* dob_ = d_o + (long)i * L * ldo + h * DH;
  int kf = wave;
  bf16x8 k0 = gfrag(kb_, ld, kf * 16 + lr, Lk, lg * 8), k1 = gfrag(kb_, ld, kf * 16 + lr, Lk, 32 + lg * 8);
  bf16x8 v0 = gfrag(vb_, ld, kf * 16 + lr, Lk, lg * 8), v1 = gfrag(vb_, ld, kf * 16 + lr, Lk, 32 + lg * 8);
  t64_stage2<QN * 16, NW * 64>(Qt, qb_, ld, Gt, dob_, ldo, L, tid);
  for (int idx = tid; idx < QN * 16; idx += NW * 64) {
    // rows >= L: lse = +inf makes P = exp2(-inf) = 0 without an explicit mask
    lse_s[idx] = idx < L ? lse[((long)i * H + h) * L + idx] * LOG2E : INFINITY;
    del_s[idx] = idx < L ? delta[((long)i * H + h) * L + idx] : 0.f;
  }
  if (dbias)
    for (int idx = tid; idx < NW * 128; idx += NW * 64) red[idx] = 0.f;
  __syncthreads();
  const float c = scale * LOG2E;

  // S and dP of query fragment f against this wave's key fragment: p = P[q][key], ds = dS[q][key]
  auto pds = [&](int f, f32x4& p, f32x4& ds) {
    const bf16x8 q0 = t64_row(Qt, f * 16 + lr, lg), q1 = t64_row(Qt, f * 16 + lr, 4 + lg);
    const bf16x8 g0 = t64_row(Gt, f * 16 + lr, lg), g1 = t64_row(Gt, f * 16 + lr, 4 + lg);
    const float4 l4 = *reinterpret_cast<const float4*>(lse_s + f * 16 + lg * 4);
    const float4 d4 = *reinterpret_cast<const float4*>(del_s + f * 16 + lg * 4);
    const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
    s = mfma16(q0, k0, s);     // D[q = 4lg+r][key = lr]
    s = mfma16(q1, k1, s);
    dp = mfma16(g0, v0, dp);   // dP[q][key] = sum_d dO[q][d] V[key][d]
    dp = mfma16(g1, v1, dp);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c, -lv[r]));
      ds[r] = p[r] * (dp[r] - dl[r]);
    }
  };

  for (; kf * 16 < L; kf += NW) {
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      dk[d] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int ip = 0; ip < QN / 2; ++ip) {
      f32x4 pp[2], ds[2];
      pds(2 * ip, pp[0], ds[0]);
      __builtin_amdgcn_sched_barrier(0);   // one fragment's operand reads at a time (register pressure)
      pds(2 * ip + 1, pp[1], ds[1]);
      const bf16x8 pf = pack8(pp[0], pp[1]), dsf = pack8(ds[0], ds[1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const bf16x8 gt = t64_trpair(Gt, (2 * ip) * 16 + 4 * lg, (2 * ip + 1) * 16 + 4 * lg, d, lr);
        const bf16x8 qt = t64_trpair(Qt, (2 * ip) * 16 + 4 * lg, (2 * ip + 1) * 16 + 4 * lg, d, lr);
        dv[d] = mfma16(gt, pf, dv[d]);    // D[d = 4lg+r][key = lr]
        dk[d] = mfma16(qt, dsf, dk[d]);
      }
    }
    if constexpr (QN & 1) {
      f32x4 pp, ds;
      pds(QN - 1, pp, ds);
      const s16x4 pf = pack4(pp), dsf = pack4(ds);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        dv[d] = mfma16k16(t64_tr(Gt, (QN - 1) * 16 + 4 * lg, d, lr), pf, dv[d]);
        dk[d] = mfma16k16(t64_tr(Qt, (QN - 1) * 16 + 4 * lg, d, lr), dsf, dk[d]);
      }
    }
    mfma_drain(dk[0], dk[1], dk[2], dk[3]);
    mfma_drain(dv[0], dv[1], dv[2], dv[3]);
    const int krow = kf * 16 + lr;
    {
      // the wave's NEXT key fragment goes straight into the registers of the finished one (rows >= Lk
      // load nothing); the stores and the column sums below cover part of its latency
      const int kn = (kf + NW) * 16 + lr;
      k0 = gfrag(kb_, ld, kn, Lk, lg * 8); k1 = gfrag(kb_, ld, kn, Lk, 32 + lg * 8);
      v0 = gfrag(vb_, ld, kn, Lk, lg * 8); v1 = gfrag(vb_, ld, kn, Lk, 32 + lg * 8);
    }
    // key rows >= Lk (masked or padded): k = v = 0 there, so S = 0 and P = exp2(-lse) != 0 - their
    // columns are garbage and are replaced by zeros (rows < L must still be written: the dX GEMM
    // reads every row of dqkv)
    const bool live = krow < Lk;
    {
      bf16* rowk = dqkv + ((long)i * L + krow) * ld + (long)H * DH + h * DH;
      store_ot_rows(rowk, dk, scale, lg, krow < L, live);
      store_ot_rows(rowk + (long)H * DH, dv, 1.0f, lg, krow < L, live);
    }
    if (dbias) {
      // column sums over this fragment's live keys (fp32, before the bf16 rounding), wave-private LDS row
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float tk = rowsum16(live ? dk[d][r] : 0.f), tv = rowsum16(live ? dv[d][r] : 0.f);
          if (lr == 0) {
            red[wave * 128 + d * 16 + lg * 4 + r] += tk * scale;
            red[wave * 128 + 64 + d * 16 + lg * 4 + r] += tv;
          }
        }
    }
  }
  if (dbias) {   // dbias[i][1][h][:] (key) and dbias[i][2][h][:] (value)
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, d = tid & 63;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red[w * 128 + tid];
      dbias[((long)i * 3 * H + (long)(1 + which) * H + h) * DH + d] = t;
    }
  }
}

// ----------------------------------------- backward: dK, dV with 32-key blocks --
// Same mathematics and LDS tiles as attn3_bwd_dkv_kernel; a wave owns TWO key fragments (32 keys, K / V rows in
// registers) and sweeps the query fragments in pairs, so every LDS operand fragment (Q / dO rows, Q^T / dO^T
// transposed pairs) feeds two MFMAs instead of one: half the LDS bytes per (query, key) block - the resource
// the 16-key kernel spends 40 % of its time on.  ~200 VGPRs: two waves per SIMD, NW waves per
// workgroup walk the ceil(QN / 2) key blocks.
template <int QN, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn4_bwd_dkv_kernel(const bf16* __restrict__ qkv,
                                                                   const bf16* __restrict__ d_o,
                                                                   const float* __restrict__ lse,
                                                                   const float* __restrict__ delta,
                                                                   bf16* __restrict__ dqkv,
                                                                   float* __restrict__ dbias,
                                                                   const int* __restrict__ kv_len, int L,
                                                                   int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qt = smem;
  char* Gt = smem + QN * 16 * 128;
  float* lse_s = reinterpret_cast<float*>(smem + 2 * QN * 16 * 128);
  float* del_s = lse_s + QN * 16;
  float* red = del_s + QN * 16;   // [NW waves][2][64] column sums of dK / dV
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int i = blockIdx.x / H, h = blockIdx.x % H;
  const int Lk = kv_len ? min(kv_len[i], L) : L;
  const long ld = 3L * H * DH, ldo = (long)H * DH;
  const bf16* qb_ = qkv + (long)i * L * ld + h * DH;
  const bf16* kb_ = qb_ + (long)H * DH;
  const bf16* vb_ = qb_ + 2L * H * DH;
  const bf16* dob_ = d_o + (long)i * L * ldo + h * DH;
  constexpr int NB = (QN + 1) / 2;   // key blocks of 32
  int kb = wave;
  bf16x8 k0[2], k1[2], v0[2], v1[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int kr = (2 * kb + e) * 16 + lr;
    k0[e] = gfrag(kb_, ld, kr, Lk, lg * 8); k1[e] = gfrag(kb_, ld, kr, Lk, 32 + lg * 8);
    v0[e] = gfrag(vb_, ld, kr, Lk, lg * 8); v1[e] = gfrag(vb_, ld, kr, Lk, 32 + lg * 8);
  }
  t64_stage2<QN * 16, NW * 64>(Qt, qb_, ld, Gt, dob_, ldo, L, tid);
  for (int idx = tid; idx < QN * 16; idx += NW * 64) {
    lse_s[idx] = idx < L ? lse[((long)i * H + h) * L + idx] * LOG2E : INFINITY;   // rows >= L: P = exp2(-inf) = 0
    del_s[idx] = idx < L ? delta[((long)i * H + h) * L + idx] : 0.f;
  }
  if (dbias)
    for (int idx = tid; idx < NW * 128; idx += NW * 64) red[idx] = 0.f;
  __syncthreads();
  const float c = scale * LOG2E;

  // P and dS of query fragment f against this wave's two key fragments: p[e][r] = P[q = 4lg+r][key = lr of fragment e]
  auto ldrows = [&](int f, bf16x8 (&rows)[4]) __attribute__((always_inline)) {
    rows[0] = t64_row(Qt, f * 16 + lr, lg); rows[1] = t64_row(Qt, f * 16 + lr, 4 + lg);
    rows[2] = t64_row(Gt, f * 16 + lr, lg); rows[3] = t64_row(Gt, f * 16 + lr, 4 + lg);
  };
  auto pds_of = [&](int f, bf16x8 q0, bf16x8 q1, bf16x8 g0, bf16x8 g1, f32x4 (&p)[2], f32x4 (&ds)[2]) __attribute__((always_inline)) {
    const float4 l4 = *reinterpret_cast<const float4*>(lse_s + f * 16 + lg * 4);
    const float4 d4 = *reinterpret_cast<const float4*>(del_s + f * 16 + lg * 4);
    const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
      sv = mfma16(q0, k0[e], sv);
      sv = mfma16(q1, k1[e], sv);
      dp = mfma16(g0, v0[e], dp);
      dp = mfma16(g1, v1[e], dp);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p[e][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[r], c, -lv[r]));
        ds[e][r] = p[e][r] * (dp[r] - dl[r]);
      }
    }
  };
  auto pds = [&](int f, f32x4 (&p)[2], f32x4 (&ds)[2]) __attribute__((always_inline)) {
    const bf16x8 q0 = t64_row(Qt, f * 16 + lr, lg), q1 = t64_row(Qt, f * 16 + lr, 4 + lg);
    const bf16x8 g0 = t64_row(Gt, f * 16 + lr, lg), g1 = t64_row(Gt, f * 16 + lr, 4 + lg);
    pds_of(f, q0, q1, g0, g1, p, ds);
  };
  constexpr bool DKV_PIPE = A4_DKV_PIPE && QN >= 28 && !(QN & 1);

  for (; kb < NB; kb += NW) {
    f32x4 dk[2][4], dv[2][4];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        dk[e][d] = f32x4{0.f, 0.f, 0.f, 0.f};
        dv[e][d] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    bf16x8 r0[4], r1[4];
    if constexpr (DKV_PIPE) ldrows(0, r0);
#pragma unroll 1
    for (int ip = 0; ip < QN / 2; ++ip) {
      f32x4 pa[2], dsa[2], pb[2], dsb[2];
      if constexpr (DKV_PIPE) {
        // 7 waves per CU hide little: the rows of fragment 2 ip + 1 are requested before fragment 2 ip is computed, those
        // of the next pair's first fragment before 2 ip + 1 (the transposed products below cover that round trip)
        ldrows(2 * ip + 1, r1);
        __builtin_amdgcn_sched_barrier(0);
        pds_of(2 * ip, r0[0], r0[1], r0[2], r0[3], pa, dsa);
        __builtin_amdgcn_sched_barrier(0);
        ldrows(min(2 * ip + 2, QN - 1), r0);
        __builtin_amdgcn_sched_barrier(0);
        pds_of(2 * ip + 1, r1[0], r1[1], r1[2], r1[3], pb, dsb);
      } else {
      pds(2 * ip, pa, dsa);
      __builtin_amdgcn_sched_barrier(0);   // one query fragment's operand reads at a time (register pressure)
      pds(2 * ip + 1, pb, dsb);
      }
      bf16x8 pf[2], dsf[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        pf[e] = pack8(pa[e], pb[e]);
        dsf[e] = pack8(dsa[e], dsb[e]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const bf16x8 gt = t64_trpair(Gt, (2 * ip) * 16 + 4 * lg, (2 * ip + 1) * 16 + 4 * lg, d, lr);
        const bf16x8 qt = t64_trpair(Qt, (2 * ip) * 16 + 4 * lg, (2 * ip + 1) * 16 + 4 * lg, d, lr);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          dv[e][d] = mfma16(gt, pf[e], dv[e][d]);    // D[d = 4lg+r][key = lr]
          dk[e][d] = mfma16(qt, dsf[e], dk[e][d]);
        }
      }
    }
    if constexpr (QN & 1) {
      f32x4 pp[2], ds[2];
      pds(QN - 1, pp, ds);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const s16x4 gt = t64_tr(Gt, (QN - 1) * 16 + 4 * lg, d, lr), qt = t64_tr(Qt, (QN - 1) * 16 + 4 * lg, d, lr);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          dv[e][d] = mfma16k16(gt, pack4(pp[e]), dv[e][d]);
          dk[e][d] = mfma16k16(qt, pack4(ds[e]), dk[e][d]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      mfma_drain(dk[e][0], dk[e][1], dk[e][2], dk[e][3]);
      mfma_drain(dv[e][0], dv[e][1], dv[e][2], dv[e][3]);
    }
    // the wave's NEXT key block goes straight into the registers of the finished one
    const int kcur = kb;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int kn = (2 * (kb + NW) + e) * 16 + lr;
      k0[e] = gfrag(kb_, ld, kn, Lk, lg * 8); k1[e] = gfrag(kb_, ld, kn, Lk, 32 + lg * 8);
      v0[e] = gfrag(vb_, ld, kn, Lk, lg * 8); v1[e] = gfrag(vb_, ld, kn, Lk, 32 + lg * 8);
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int krow = (2 * kcur + e) * 16 + lr;
      // key rows >= Lk (masked or padded): k = v = 0 there, so S = 0 and P = exp2(-lse) != 0 - their columns are
      // garbage and are replaced by zeros (rows < L must still be written: the dX GEMM reads every row of dqkv)
      const bool live = krow < Lk;
      {
        bf16* rowk = dqkv + ((long)i * L + krow) * ld + (long)H * DH + h * DH;
        store_ot_rows(rowk, dk[e], scale, lg, krow < L, live);
        store_ot_rows(rowk + (long)H * DH, dv[e], 1.0f, lg, krow < L, live);
      }
      if (dbias) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float tk = rowsum16(live ? dk[e][d][r] : 0.f), tv = rowsum16(live ? dv[e][d][r] : 0.f);
            if (lr == 0) {
              red[wave * 128 + d * 16 + lg * 4 + r] += tk * scale;
              red[wave * 128 + 64 + d * 16 + lg * 4 + r] += tv;
            }
          }
      }
    }
  }
  if (dbias) {   // dbias[i][1][h][:] (key) and dbias[i][2][h][:] (value)
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, d = tid & 63;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += red[w * 128 + tid];
      dbias[((long)i * 3 * H + (long)(1 + which) * H + h) * DH + d] = t;
    }
  }
}

// A/B switches of a launch, decoded from the caller's BV_OPT_ATTN_CFG (include/bvhip.h)
struct A3Cfg {
  bool fwd8;        // bit 8:   forward of the L <= 208 kernels as 8 waves x 2 workgroups
  bool one_sweep;   // !bit 16: one-sweep dQ kernel (default); bit 16 = always the two-sweep kernel
  int dkv32;        // bits 32 / 64: 32-key-block dK/dV kernel, 1 = 4 waves x 2 workgroups per CU, 2 = 7 waves x 1
  bool a5_off;      // bit 128: keep the two-launch backward also where the one-launch kernel (attention5.hip) applies
  bool a5_bias_dpp; // bit 256: attention5.hip reduces the bias gradients with DPP column sums instead of the identities
  bool dkv_classic; // bit 1024: the 16-key-fragment dK/dV kernel also for the long sequences (28+ fragments), see launch_bwd3
};
A3Cfg a3cfg(const bv_ctx* ctx) {
  const long c = bv_opt(ctx, BV_OPT_ATTN_CFG);
  return A3Cfg{(c & 15) == 8, !(c & 16), (c & 64) ? 2 : (c & 32) ? 1 : 0, (c & 128) != 0, (c & 256) != 0, (c & 1024) != 0};
}

template <typename K>
void set_lds(K kernel, size_t bytes) {
  if (bytes > 65536)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int KF, int NW, int WPS>
int launch_fwd3(const void* qkv, void* o, float* lse, const int* kv_len, int n, int L, int H, hipStream_t s) {
  const size_t sh = (size_t)KF * 4096;
  if (!kv_len && L > (KF - 1) * 16) {
    set_lds(attn3_fwd_kernel<KF, NW, WPS, true>, sh);
    hipLaunchKernelGGL((attn3_fwd_kernel<KF, NW, WPS, true>), dim3(n * H), dim3(NW * 64), sh, s, (const bf16*)qkv,
                       (bf16*)o, lse, kv_len, L, H, 0.125f);
  } else {
    set_lds(attn3_fwd_kernel<KF, NW, WPS, false>, sh);
    hipLaunchKernelGGL((attn3_fwd_kernel<KF, NW, WPS, false>), dim3(n * H), dim3(NW * 64), sh, s, (const bf16*)qkv,
                       (bf16*)o, lse, kv_len, L, H, 0.125f);
  }
  return bv_check_launch("bv_attn_fwd");
}
// NW / WPS: the dQ kernel, NW2 / WPS2: the dK,dV kernel (waves per workgroup / per SIMD)
template <int KF, int NW, int WPS, int WPS2, int NW2 = NW>
int launch_bwd3(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                float* dbias, const int* kv_len, int n, int L, int H, hipStream_t s, const A3Cfg& cfg) {
  const bool g_a3_one_sweep = cfg.one_sweep;
  // Long sequences (28+ key fragments: L/16 at 336 px = 441 tokens, 576 at 384 px), unmasked: the 32-key-block dK/dV
  // kernel with 7 waves is the default since round 6 - bit-identical results, whole backward 1360 -> 1249 us at n = 256,
  // L = 441, H = 16 and 716 -> 686 us at L = 576 (tools/attn_longseq_cfg_ab.py, profiles/r06_attn_longseq_cfg_ab.txt);
  // at 13-17 fragments it loses (L = 256: 366 -> 461 us) and stays opt-in
  const int g_a4_dkv = cfg.dkv32 ? cfg.dkv32 : ((KF >= 28 && !kv_len && !cfg.dkv_classic) ? 2 : 0);
  const size_t sh1 = (size_t)KF * 4096 + (size_t)NW * 64 * 4;
  if (o && g_a3_one_sweep) {
    if (!kv_len && L > (KF - 1) * 16) {
      constexpr int NWT = (KF == 28 && A3_DQ_LONG_NW == 16) ? 16 : NW, WPST = (KF == 28 && A3_DQ_LONG_NW == 16) ? 4 : WPS;
      const size_t sh1 = (size_t)KF * 4096 + (size_t)NWT * 64 * 4;
      set_lds(attn3_bwd_dq1_kernel<KF, NWT, WPST, true>, sh1);
      hipLaunchKernelGGL((attn3_bwd_dq1_kernel<KF, NWT, WPST, true>), dim3(n * H), dim3(NWT * 64), sh1, s,
                         (const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse, delta, (bf16*)dqkv, dbias, kv_len,
                         L, H, 0.125f);
    } else {
      set_lds(attn3_bwd_dq1_kernel<KF, NW, WPS, false>, sh1);
      hipLaunchKernelGGL((attn3_bwd_dq1_kernel<KF, NW, WPS, false>), dim3(n * H), dim3(NW * 64), sh1, s,
                         (const bf16*)qkv, (const bf16*)o, (const bf16*)d_o, lse, delta, (bf16*)dqkv, dbias, kv_len,
                         L, H, 0.125f);
    }
  } else {
    set_lds(attn3_bwd_dq_kernel<KF, NW, WPS>, sh1);
    hipLaunchKernelGGL((attn3_bwd_dq_kernel<KF, NW, WPS>), dim3(n * H), dim3(NW * 64), sh1, s, (const bf16*)qkv,
                       (const bf16*)d_o, lse, delta, (bf16*)dqkv, dbias, kv_len, L, H, 0.125f);
  }
  int rc = bv_check_launch("bv_attn_bwd(dq)");
  if (rc) return rc;
  if (g_a4_dkv && KF >= 13) {   // 32-key blocks (A/B: BV_OPT_ATTN_CFG bits 32 = 4 waves x 2 workgroups, 64 = 7 waves x 1)
    if (g_a4_dkv == 2) {
      const size_t sh = (size_t)KF * 4096 + (size_t)KF * 16 * 8 + (size_t)7 * 128 * 4;
      set_lds(attn4_bwd_dkv_kernel<KF, 7>, sh);
      hipLaunchKernelGGL((attn4_bwd_dkv_kernel<KF, 7>), dim3(n * H), dim3(7 * 64), sh, s, (const bf16*)qkv,
                         (const bf16*)d_o, lse, delta, (bf16*)dqkv, dbias, kv_len, L, H, 0.125f);
    } else {
      const size_t sh = (size_t)KF * 4096 + (size_t)KF * 16 * 8 + (size_t)4 * 128 * 4;
      set_lds(attn4_bwd_dkv_kernel<KF, 4>, sh);
      hipLaunchKernelGGL((attn4_bwd_dkv_kernel<KF, 4>), dim3(n * H), dim3(4 * 64), sh, s, (const bf16*)qkv,
                         (const bf16*)d_o, lse, delta, (bf16*)dqkv, dbias, kv_len, L, H, 0.125f);
    }
    return bv_check_launch("bv_attn_bwd(dkv32)");
  }
  const size_t sh2 = (size_t)KF * 4096 + (size_t)KF * 16 * 8 + (size_t)NW2 * 128 * 4;
  set_lds(attn3_bwd_dkv_kernel<KF, NW2, WPS2>, sh2);
  hipLaunchKernelGGL((attn3_bwd_dkv_kernel<KF, NW2, WPS2>), dim3(n * H), dim3(NW2 * 64), sh2, s, (const bf16*)qkv,
                     (const bf16*)d_o, lse, delta, (bf16*)dqkv, dbias, kv_len, L, H, 0.125f);
  return bv_check_launch("bv_attn_bwd(dkv)");
}

}  // namespace

// Entry points used by bv_attn_fwd / bv_attn_bwd (attention.hip).  Key fragments = ceil(L/16) for
// the common sequence lengths (64 text tokens; 196/197 at 224 px; 256/257; 441 at 336 px; 576 at
// 384 px), the next instantiated size otherwise.
// attention5.hip: the backward in one launch (unmasked, L <= 64 or 193..208); -100 = shape not covered
int bv_attn5_bwd(const void* qkv, const void* d_o, const float* lse, float* delta, void* dqkv, float* dbias, int n,
                 int L, int H, void* stream, bool bias_dpp);

int bv_attn3_fwd(const void* qkv, void* o, float* lse, const int* kv_len, int n, int L, int H, void* stream,
                 const bv_ctx* ctx) {
  hipStream_t s = (hipStream_t)stream;
  const A3Cfg cfg = a3cfg(ctx);
  if (L <= 64) return launch_fwd3<4, 4, 4>(qkv, o, lse, kv_len, n, L, H, s);
  // 13 key fragments: 4 waves per workgroup and 3 workgroups per CU (the third one computes while
  // another stages its K/V: 605-650 us instead of 670-730 at n = 2048); 8 waves x 2 under BV_OPT_ATTN_CFG = 8
  if (L <= 208 && cfg.fwd8) return launch_fwd3<13, 8, 4>(qkv, o, lse, kv_len, n, L, H, s);
  if (L <= 208) return launch_fwd3<13, 4, 3>(qkv, o, lse, kv_len, n, L, H, s);
  if (L <= 272) return launch_fwd3<17, 8, 4>(qkv, o, lse, kv_len, n, L, H, s);
  if (L <= 448) return launch_fwd3<28, A3_FWD_LONG_NW, A3_FWD_LONG_NW == 12 ? 3 : 2>(qkv, o, lse, kv_len, n, L, H, s);
  return launch_fwd3<36, 8, 2>(qkv, o, lse, kv_len, n, L, H, s);
}

int bv_attn3_bwd(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                 float* dbias, const int* kv_len, int n, int L, int H, void* stream, const bv_ctx* ctx) {
  hipStream_t s = (hipStream_t)stream;
  const A3Cfg cfg = a3cfg(ctx);
  if (!kv_len && !cfg.a5_off) {
    const int rc = bv_attn5_bwd(qkv, d_o, lse, delta, dqkv, dbias, n, L, H, stream, cfg.a5_bias_dpp);
    if (rc != -100) return rc;
  }
  if (L <= 64) return launch_bwd3<4, 4, 4, 4>(qkv, o, d_o, lse, delta, dqkv, dbias, kv_len, n, L, H, s, cfg);
  if (L <= 208) return launch_bwd3<13, 8, 4, 4>(qkv, o, d_o, lse, delta, dqkv, dbias, kv_len, n, L, H, s, cfg);
  if (L <= 272) return launch_bwd3<17, 8, 4, 4>(qkv, o, d_o, lse, delta, dqkv, dbias, kv_len, n, L, H, s, cfg);
  if (L <= 448) return launch_bwd3<28, 8, 2, 2>(qkv, o, d_o, lse, delta, dqkv, dbias, kv_len, n, L, H, s, cfg);
  return launch_bwd3<36, 8, 2, 2>(qkv, o, d_o, lse, delta, dqkv, dbias, kv_len, n, L, H, s, cfg);
}
