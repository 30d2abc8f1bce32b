// Fused self-attention BACKWARD in ONE launch for gfx950, Dh = 64, L <= 208 (fourth generation; the forward and
// the L > 208 / key-padded cases stay in attention3.hip).  Same mathematics and reference call sites as
// attention3.hip (flax nn.MultiHeadDotProductAttention inside big_vision/models/vit.py:93-98, text tower
// models/proj/image_text/text_transformer.py:72-75; backward = jax.value_and_grad,
// trainers/proj/image_text/siglip.py:311):
//   P = softmax_rows((q/sqrt(Dh)) k^T), dV = P^T dO, dP = dO V^T, delta_i = sum_j P_ij dP_ij,
//   dS = P o (dP - delta), dQ = dS K/sqrt(Dh), dK = dS^T Q/sqrt(Dh)
//
// Why.  attention3 runs the backward as two launches (query-owned dQ sweep, key-owned dK/dV sweep): q, k, v, dO
// are staged twice, S / dP / the exponentials are computed twice or three times, and each launch sits at
// 3.5 TB/s = neither roof (1059 + 1107 us at n = 2048, L = 196, H = 12 against a 0.62 ms HBM floor,
// profiles/r03_bench_kernel_stats.csv).  Here one persistent workgroup per CU walks the (sample, head) pairs and
// reads every operand from HBM exactly once (175 KB per pair at L = 196: q, k, v, dO in, dq, dk, dv out; O is not
// read at all), with the NEXT pair's tiles in flight while the current one is computed:
//   * KF = ceil(L / 16) waves, wave w OWNS key fragment w (its K / V rows live in registers for the whole pair)
//     and, in the last phase, query fragment w.  Q and dO of the pair are LDS tiles (T64 images, attn_common.h).
//   * phase 1a  S^T = K Q^T, dP^T = V dO^T, P^T: the wave's partial of delta_i = sum_j P_ij dP_ij (fp32, exact -
//               the cancellation-safe form, see attention3.hip) for every query -> LDS partials -> barrier ->
//               fixed-order sum (deterministic, no atomics).
//   * phase 1b  S = Q K^T, dP = dO V^T again (MFMAs are not the bound), dS = P o (dP - delta) with the exact
//               delta, dV^T += dO^T P and dK^T += Q^T dS in the wave's accumulators; dS^T goes to LDS as bf16
//               (one [keys][16 q] tile per query fragment, plane layout below).
//   * phase 2   the K fragments are written over the Q tile; wave w computes dQ^T of query fragment w
//               = K^T dS^T from the two LDS images (every operand a ds_read_b64_tr_b16) - the reduction over the
//               key-owning waves happens inside the MFMA accumulators, not through LDS atomics.
//   7 matmuls (attention3: 8 incl. the P K correction) and two exponentials per score, six workgroup barriers
//   per pair.  Column sums of dq / dk / dv (the q/k/v bias gradients) as in attention3 (DPP + LDS rows).
// dS^T tile of query fragment f ("plane layout"): byte address f * TS + (q & 15) / 4 * PL + key * 8 + (q & 3) * 2,
// PL = R * 8 + 64.  The 1b store (lane = key, 4 consecutive q) is a ds_write_b64 whose 16-lane groups cover 128
// contiguous bytes; the phase-2 transposed read (ds_read_b64_tr_b16: 16 lanes fetch a [4 keys][16 q] block) touches
// 4 planes x 32 contiguous bytes per lane group of 16, planes 16 dwords apart mod 64 banks: conflict-free both ways.
#include "attn_common.h"
#include "bvhip_internal.h"
#ifndef A5_UNROLL_1A
#define A5_UNROLL_1A 1
#endif

namespace {
using namespace bvattn;

__device__ __forceinline__ void a5_drain(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  // wait states between the last MFMAs of a loop and VALU reads of their results behind control flow
  // (hipcc pads the hazard inside a basic block only, see attention3.hip)
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

// transposed read of a dS^T plane tile: lane lr of a 16-lane group receives keys (row4 .. row4 + 3) of query
// column lr (row4 includes the lane group's 4 * lg)
template <int PL>
__device__ __forceinline__ s16x4 a5_tr(const char* T, int row4, int lr) {
  const char* p = T + (lr & 3) * PL + (row4 + (lr >> 2)) * 8;
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}
template <int PL>
__device__ __forceinline__ bf16x8 a5_trpair(const char* T, int ra, int rb, int lr) {
  const s16x4 a = a5_tr<PL>(T, ra, lr), b = a5_tr<PL>(T, rb, lr);
  const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <int KF, int LW>
struct A5 {
  static constexpr int R = KF * 16;            // padded rows of a tile
  static constexpr int NTC = KF * 64;          // compute threads (wave w < KF owns key / query fragment w)
  static constexpr int NTL = LW * 64;          // loader threads (waves KF .. KF + LW - 1)
  static constexpr int NT = NTC + NTL;
  static constexpr int NPL = (R * 8 + NTL - 1) / NTL;   // 16-byte pieces of one tile per loader lane
  static constexpr int PL = R * 8 + 64;        // bytes of one q-quad plane of a dS^T tile
  static constexpr int TS = 4 * PL;            // bytes of one dS^T tile (one query fragment)
  static constexpr int OFF_G = R * 128;
  static constexpr int OFF_DS = 2 * R * 128;
  static constexpr int OFF_LSE = OFF_DS + KF * TS;
  static constexpr int OFF_DEL = OFF_LSE + R * 4;
  static constexpr int OFF_RED = OFF_DEL + R * 4;
  static constexpr int LDS = OFF_RED + KF * 192 * 4;
  static_assert(KF * 4 * R * 4 <= KF * TS, "delta partials overlay the dS^T tiles");
  static_assert(NT <= 1024 && R <= NTC && 192 <= NTC, "workgroup shape");
};

// Workgroup = KF compute waves + LW loader waves.  The loader waves own the NEXT pair's Q / dO tiles (in registers,
// loaded right after the current pair's tiles became visible, i.e. in flight during the whole pair) and write them
// into the LDS tiles as soon as those are free (dO after phase 1b, Q after phase 2): the compute waves sit at the
// 128-VGPR line of four waves per SIMD and cannot hold 17 prefetch registers on top of K, V, two accumulator sets
// and the fragment operands (first version: 29-74 spilled VGPRs).  Every wave passes the same six barriers per pair.
template <int KF, int LW, bool DBIAS>
__global__ __launch_bounds__((KF + LW) * 64) void attn5_bwd_kernel(const bf16* __restrict__ qkv,
                                                                   const bf16* __restrict__ d_o,
                                                                   const float* __restrict__ lse,
                                                                   float* __restrict__ delta,
                                                                   bf16* __restrict__ dqkv,
                                                                   float* __restrict__ dbias, int L, int H,
                                                                   int npairs, float scale) {
  using C = A5<KF, LW>;
  constexpr int R = C::R, PL = C::PL, TS = C::TS, NTL = C::NTL, NPL = C::NPL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qt = smem;                 // Q tile; K tile in phase 2
  char* Gt = smem + C::OFF_G;      // dO tile
  char* dSt = smem + C::OFF_DS;    // KF dS^T tiles; the delta partials [KF][4][R] of phase 1a overlay them
  float* dpart = reinterpret_cast<float*>(dSt);
  float* lse_s = reinterpret_cast<float*>(smem + C::OFF_LSE);
  float* del_s = reinterpret_cast<float*>(smem + C::OFF_DEL);
  float* red = reinterpret_cast<float*>(smem + C::OFF_RED);   // [KF waves][3][64] column sums of dq / dk / dv
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const long ld = 3L * H * DH, ldo = (long)H * DH;
  const float c = scale * LOG2E;
  const int stride = gridDim.x;

  if (wave >= KF) {
    // ================================================================ loader waves ==========================
    const int ll = tid - C::NTC;
    uint4 pq[NPL], pg[NPL];
    float pl[(R + NTL - 1) / NTL];
    auto load_tiles = [&](int pair) {
      const int i = pair / H, h = pair % H;
      const bf16* qb_ = qkv + (long)i * L * ld + h * DH;
      const bf16* dob_ = d_o + (long)i * L * ldo + h * DH;
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        const int idx = ll + j * NTL, row = idx >> 3, pc = idx & 7;
        pq[j] = make_uint4(0, 0, 0, 0);
        pg[j] = make_uint4(0, 0, 0, 0);
        if (row < L) {   // rows >= L (and pieces beyond the tile) stay zero
          pq[j] = *reinterpret_cast<const uint4*>(qb_ + (long)row * ld + pc * 8);
          pg[j] = *reinterpret_cast<const uint4*>(dob_ + (long)row * ldo + pc * 8);
        }
      }
#pragma unroll
      for (int j = 0; j < (R + NTL - 1) / NTL; ++j) {
        const int q = ll + j * NTL;
        // rows >= L: lse = +inf makes P = exp2(-inf) = 0 without an explicit query mask
        pl[j] = q < L ? lse[(long)pair * L + q] * LOG2E : INFINITY;
      }
    };
    auto put_tile = [&](char* T, const uint4 (&pc_)[NPL]) {
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        const int idx = ll + j * NTL, row = idx >> 3, pc = idx & 7;
        if (idx < R * 8) *reinterpret_cast<uint4*>(T + row * 128 + ((pc ^ t64_swz(row)) << 4)) = pc_[j];
      }
    };
    auto put_lse = [&]() {
#pragma unroll
      for (int j = 0; j < (R + NTL - 1) / NTL; ++j) {
        const int q = ll + j * NTL;
        if (q < R) lse_s[q] = pl[j];
      }
    };
    int pair = blockIdx.x;
    if (pair < npairs) {
      load_tiles(pair);
      put_tile(Gt, pg);
      put_lse();
      put_tile(Qt, pq);
    }
    for (; pair < npairs; pair += stride) {
      __syncthreads();   // B1: this pair's tiles are visible
      const int nxt = pair + stride;
      if (nxt < npairs) load_tiles(nxt);
      __syncthreads();   // B2
      __syncthreads();   // B3
      __syncthreads();   // B4: the dO tile and lse are free
      if (nxt < npairs) {
        put_tile(Gt, pg);
        put_lse();
      }
      __syncthreads();   // B5
      __syncthreads();   // B6: the K tile (= Q tile) is free
      if (nxt < npairs) put_tile(Qt, pq);
    }
    return;
  }

  // ================================================================== compute waves ==========================
  bf16x8 k0, k1, v0, v1;
  auto load_k = [&](int pair) {
    const int i = pair / H, h = pair % H;
    const bf16* kb_ = qkv + (long)i * L * ld + (long)H * DH + h * DH;
    const int kr = wave * 16 + lr;
    k0 = gfrag(kb_, ld, kr, L, lg * 8); k1 = gfrag(kb_, ld, kr, L, 32 + lg * 8);
  };
  auto load_v = [&](int pair) {
    const int i = pair / H, h = pair % H;
    const bf16* vb_ = qkv + (long)i * L * ld + 2L * H * DH + h * DH;
    const int kr = wave * 16 + lr;
    v0 = gfrag(vb_, ld, kr, L, lg * 8); v1 = gfrag(vb_, ld, kr, L, 32 + lg * 8);
  };

  int pair = blockIdx.x;
  if (pair < npairs) {
    load_k(pair);
    load_v(pair);
  }
  for (; pair < npairs; pair += stride) {
    const int i = pair / H, h = pair % H;
    const int nxt = pair + stride;
    __syncthreads();   // B1

    // ---- phase 1a: partials of delta.  S^T[key = 4 lg + r][q = lr] of (key fragment `wave`, query fragment f)
    {
      const int lim = L - wave * 16 - lg * 4;   // key 4 lg + r of this wave's fragment exists for r < lim
#pragma unroll A5_UNROLL_1A
      for (int f = 0; f < KF; ++f) {
        const bf16x8 q0 = t64_row(Qt, f * 16 + lr, lg), q1 = t64_row(Qt, f * 16 + lr, 4 + lg);
        const bf16x8 g0 = t64_row(Gt, f * 16 + lr, lg), g1 = t64_row(Gt, f * 16 + lr, 4 + lg);
        const float nl = -lse_s[f * 16 + lr];
        f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
        st = mfma16(k0, q0, st);
        st = mfma16(k1, q1, st);
        dp = mfma16(v0, g0, dp);
        dp = mfma16(v1, g1, dp);
        float x = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // padded keys have k = v = 0: S = 0, dP = 0, but exp2(-lse) may overflow: select, branch-free
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c, nl));
          x = __builtin_fmaf(r < lim ? e : 0.f, dp[r], x);
        }
        dpart[(wave * 4 + lg) * R + f * 16 + lr] = x;
        if (f & 1) __builtin_amdgcn_sched_barrier(0);   // bound operand-read hoisting (register pressure)
      }
    }
    __syncthreads();   // B2
    if (tid < R) {     // fixed-order sum of the KF x 4 partials of query row tid: deterministic
      float t = 0.f;
#pragma unroll 4
      for (int w = 0; w < KF * 4; ++w) t += dpart[w * R + tid];
      del_s[tid] = t;
      if (tid < L) delta[(long)pair * L + tid] = t;
    }
    __syncthreads();   // B3: delta complete; the partials are dead, the dS^T tiles may be written

    // ---- phase 1b: dV^T, dK^T of this wave's key fragment; dS^T of every (query fragment, key fragment) -> LDS
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      dk[d] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    {
      const bool klive = wave * 16 + lr < L;
      // P and dS of query fragment f against the wave's key fragment: p[r] = P[q = 4 lg + r][key = lr]
      auto pds = [&](int f, f32x4& p, f32x4& ds) {
        const bf16x8 q0 = t64_row(Qt, f * 16 + lr, lg), q1 = t64_row(Qt, f * 16 + lr, 4 + lg);
        const bf16x8 g0 = t64_row(Gt, f * 16 + lr, lg), g1 = t64_row(Gt, f * 16 + lr, 4 + lg);
        const float4 l4 = *reinterpret_cast<const float4*>(lse_s + f * 16 + lg * 4);
        const float4 d4 = *reinterpret_cast<const float4*>(del_s + f * 16 + lg * 4);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
        s = mfma16(q0, k0, s);     // D[q = 4 lg + r][key = lr]
        s = mfma16(q1, k1, s);
        dp = mfma16(g0, v0, dp);   // dP[q][key] = sum_d dO[q][d] V[key][d]
        dp = mfma16(g1, v1, dp);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c, -lv[r]));
          p[r] = klive ? e : 0.f;          // padded key rows (see phase 1a)
          ds[r] = p[r] * (dp[r] - dl[r]);
        }
        // dS^T[key][q = f * 16 + 4 lg .. + 3]: plane lg of tile f, 8 bytes per key
        *reinterpret_cast<s16x4*>(dSt + f * TS + lg * PL + (wave * 16 + lr) * 8) = pack4(ds);
      };
#pragma unroll 1
      for (int ip = 0; ip < KF / 2; ++ip) {
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
          dv[d] = mfma16(gt, pf, dv[d]);    // D[d = 4 lg + r][key = lr]
          dk[d] = mfma16(qt, dsf, dk[d]);
        }
      }
      if constexpr (KF & 1) {
        f32x4 pp, ds;
        pds(KF - 1, pp, ds);
        const s16x4 pf = pack4(pp), dsf = pack4(ds);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          dv[d] = mfma16k16(t64_tr(Gt, (KF - 1) * 16 + 4 * lg, d, lr), pf, dv[d]);
          dk[d] = mfma16k16(t64_tr(Qt, (KF - 1) * 16 + 4 * lg, d, lr), dsf, dk[d]);
        }
      }
    }
    a5_drain(dk[0], dk[1], dk[2], dk[3]);
    a5_drain(dv[0], dv[1], dv[2], dv[3]);
    if (nxt < npairs) load_v(nxt);   // V is dead: the next pair's rows are in flight from here
    __syncthreads();   // B4: every dS^T tile complete; nobody reads the Q / dO tiles any more

    // ---- K fragments -> the Q tile (T64 image, the A operand of phase 2), then this wave's dK / dV rows
    {
      const int row = wave * 16 + lr;
      *reinterpret_cast<uint4*>(Qt + row * 128 + ((lg ^ t64_swz(row)) << 4)) = __builtin_bit_cast(uint4, k0);
      *reinterpret_cast<uint4*>(Qt + row * 128 + (((4 + lg) ^ t64_swz(row)) << 4)) = __builtin_bit_cast(uint4, k1);
      if (nxt < npairs) load_k(nxt);   // the next pair's K rows replace the finished ones
      const bool live = row < L;
      if (live) {
        bf16* rowk = dqkv + ((long)i * L + row) * ld + (long)H * DH + h * DH;
        bf16* rowv = rowk + (long)H * DH;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          uint2 a, b;
          a.x = pack_bf2(dk[d][0] * scale, dk[d][1] * scale);
          a.y = pack_bf2(dk[d][2] * scale, dk[d][3] * scale);
          *reinterpret_cast<uint2*>(rowk + d * 16 + lg * 4) = a;
          b.x = pack_bf2(dv[d][0], dv[d][1]);
          b.y = pack_bf2(dv[d][2], dv[d][3]);
          *reinterpret_cast<uint2*>(rowv + d * 16 + lg * 4) = b;
        }
      }
      if constexpr (DBIAS) {
        // column sums over this fragment's live keys (fp32, before the bf16 rounding): one row of `red` per wave
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float tk = rowsum16(live ? dk[d][r] : 0.f), tv = rowsum16(live ? dv[d][r] : 0.f);
            if (lr == 0) {
              red[wave * 192 + 64 + d * 16 + lg * 4 + r] = tk * scale;
              red[wave * 192 + 128 + d * 16 + lg * 4 + r] = tv;
            }
          }
      }
    }
    __syncthreads();   // B5: K tile complete

    // ---- phase 2: dQ^T[d][q] of query fragment `wave` = sum_key K^T[d][key] dS^T[key][q]
    {
      const char* T = dSt + wave * TS;
      f32x4 dq[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
      for (int fp = 0; fp < KF / 2; ++fp) {
        const int ra = (2 * fp) * 16 + 4 * lg, rb = ra + 16;
        const bf16x8 dsf = a5_trpair<PL>(T, ra, rb, lr);
#pragma unroll
        for (int d = 0; d < 4; ++d) dq[d] = mfma16(t64_trpair(Qt, ra, rb, d, lr), dsf, dq[d]);
      }
      if constexpr (KF & 1) {
        const int ra = (KF - 1) * 16 + 4 * lg;
        const s16x4 dsf = a5_tr<PL>(T, ra, lr);
#pragma unroll
        for (int d = 0; d < 4; ++d) dq[d] = mfma16k16(t64_tr(Qt, ra, d, lr), dsf, dq[d]);
      }
      a5_drain(dq[0], dq[1], dq[2], dq[3]);
      const int qrow = wave * 16 + lr;
      if (qrow < L) {
        bf16* row = dqkv + ((long)i * L + qrow) * ld + h * DH;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          uint2 w;
          w.x = pack_bf2(dq[d][0] * scale, dq[d][1] * scale);
          w.y = pack_bf2(dq[d][2] * scale, dq[d][3] * scale);
          *reinterpret_cast<uint2*>(row + d * 16 + lg * 4) = w;
        }
      }
      if constexpr (DBIAS) {
        // padded query rows are exactly 0 (their dS^T columns are)
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float t = rowsum16(dq[d][r]);
            if (lr == 0) red[wave * 192 + d * 16 + lg * 4 + r] = t * scale;
          }
      }
    }
    __syncthreads();   // B6: the K tile, the dS^T tiles and `red` are free / complete
    if constexpr (DBIAS) {
      // per-(sample, head) column sums -> dbias[i][which][h][:]; the host sums over samples
      if (tid < 192) {
        const int which = tid >> 6, d = tid & 63;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < KF; ++w) t += red[w * 192 + tid];
        dbias[((long)i * 3 * H + (long)which * H + h) * DH + d] = t;
      }
    }
  }
}

template <int KF, int LW>
int launch_bwd5(const void* qkv, const void* d_o, const float* lse, float* delta, void* dqkv, float* dbias, int n,
                int L, int H, hipStream_t s) {
  using C = A5<KF, LW>;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  // workgroups per CU: LDS (160 KiB) and 4 waves per SIMD at <= 128 VGPRs
  int per_cu = (160 * 1024) / C::LDS;
  if (per_cu > 16 / (KF + LW)) per_cu = 16 / (KF + LW);
  if (per_cu < 1) per_cu = 1;
  const int npairs = n * H;
  const int grid = npairs < cus * per_cu ? npairs : cus * per_cu;
  if (dbias) {
    auto kern = attn5_bwd_kernel<KF, LW, true>;
    if (C::LDS > 65536)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS, s, (const bf16*)qkv, (const bf16*)d_o, lse, delta,
                       (bf16*)dqkv, dbias, L, H, npairs, 0.125f);
  } else {
    auto kern = attn5_bwd_kernel<KF, LW, false>;
    if (C::LDS > 65536)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS, s, (const bf16*)qkv, (const bf16*)d_o, lse, delta,
                       (bf16*)dqkv, dbias, L, H, npairs, 0.125f);
  }
  return bv_check_launch("bv_attn_bwd(one launch)");
}

}  // namespace

// Entry used by bv_attn3_bwd (attention3.hip) for unmasked sequences of at most 208 tokens; returns -100 when the
// shape is not covered (the caller keeps the two-launch path).
int bv_attn5_bwd(const void* qkv, const void* d_o, const float* lse, float* delta, void* dqkv, float* dbias, int n,
                 int L, int H, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (L <= 64) return launch_bwd5<4, 1>(qkv, d_o, lse, delta, dqkv, dbias, n, L, H, s);
  if (L > 192 && L <= 208) return launch_bwd5<13, 3>(qkv, d_o, lse, delta, dqkv, dbias, n, L, H, s);
  return -100;
}
