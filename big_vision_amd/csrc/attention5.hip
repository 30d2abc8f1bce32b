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
#include <type_traits>
#include "attn_common.h"
#include "bvhip_internal.h"
#ifndef A5_UNROLL_1A
#define A5_UNROLL_1A 16   // fully unrolled (like 1b and phase 2): every LDS address is a base register + an immediate
#endif
// A5_ABL != 0 only in tools/probes/attn5_probe.hip (ablations, results are garbage): 1 = no phase 1a loop, 2 = no
// phase 1b loop, 4 = no phase 2 loop, 16 = no global stores, 32 = the loader waves load nothing, 64 = every second
// fragment step re-uses the previous step's Q / dO / K operand reads (HALF the operand LDS reads of all three phases,
// same MFMA / VALU counts: the upper bound of what two key fragments per wave could save); A5_STAMPS: s_memtime
// stamps of waves 0 / KF-1 / KF (first loader) of four mid-launch workgroups
#ifndef A5_ABL
#define A5_ABL 0
#endif
// schedule knobs (swept by tools/probes/attn5_probe.hip -> profiles/r04_attn5_schedule_sweep.txt; the defaults are the
// measured best: all three loops fully unrolled, -5 % against rolled loops with running address adds)
#ifndef A5_UNROLL_1B
#define A5_UNROLL_1B 16
#endif
#ifndef A5_UNROLL_P2
#define A5_UNROLL_P2 16
#endif
#ifndef A5_SB_1B
#define A5_SB_1B 0      // 1: sched_barrier between the two fragments of a phase-1b pair and in front of its MFMA block
#endif
#ifdef A5_STAMPS
__device__ long* g_a5_stamps;
#define A5_STAMP(k)                                                                                   \
  do {                                                                                                \
    if (lane == 0 && (wave == 0 || wave == KF - 1 || wave == KF) && blockIdx.x >= 100 && blockIdx.x < 104 && \
        pair == blockIdx.x + 2 * stride)                                                              \
      g_a5_stamps[((blockIdx.x - 100) * 3 + (wave == 0 ? 0 : wave == KF - 1 ? 1 : 2)) * 16 + (k)] =   \
          __builtin_amdgcn_s_memtime();                                                               \
  } while (0)
#else
#define A5_STAMP(k)
#endif
#ifdef A5_WSTAMPS   // per-wave stamps of ONE workgroup (probe): [wave][8]
__device__ long* g_a5_wstamps;
#define A5_WSTAMP(k)                                                                     \
  do {                                                                                   \
    if (lane == 0 && blockIdx.x == 100 && pair == blockIdx.x + 2 * stride)               \
      g_a5_wstamps[wave * 8 + (k)] = __builtin_amdgcn_s_memtime();                       \
  } while (0)
#else
#define A5_WSTAMP(k)
#endif

namespace {
using namespace bvattn;
// register arrays of 16-byte pieces: a first-class vector type (arrays of HIP's uint4 STRUCT were left in scratch memory)
typedef unsigned int a5_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void a5_drain(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  // wait states between the last MFMAs of a loop and VALU reads of their results behind control flow
  // (hipcc pads the hazard inside a basic block only, see attention3.hip)
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}

// An opaque copy of a lane-dependent value.  Every phase derives its LDS / global addresses from its own copy of the
// lane id, so hipcc cannot hoist two dozen address registers out of the pair loop and keep them live through the
// phases that sit at the 128-VGPR line (it spilled them to scratch and reloaded them in front of MFMAs).
__device__ __forceinline__ int a5_opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}
// The same for the lane id, with its RANGE handed back to the compiler: behind the opaque copy hipcc no longer knows
// that lr < 16 and lg < 4, cannot prove that a fragment's row offset (f * 16) leaves the swizzle bits of a row alone,
// and re-derives every LDS address of every fragment with VALU instructions (7 of the 24 per tile of phase 1a, 230
// per pair in phase 2) instead of folding f into the instruction's immediate offset.
__device__ __forceinline__ int a5_lane(int lane) { return a5_opaque(lane) & 63; }

// Sum over the four 16-lane rows of a wave (every lane gets the total): two lane-swap VALU operations instead of two
// ds_bpermute round trips through the LDS crossbar.  Inline asm: the builtins of ROCm 7.2 drop the second result
// (__builtin_amdgcn_permlane32_swap(x, y)[1] reads the FIRST destination register - checked in the ISA).
//   v_permlane32_swap a, b: a' = {a.lanes 0-31, b.lanes 0-31}, b' = {a.lanes 32-63, b.lanes 32-63}
//   v_permlane16_swap a, b: a' = {a.row0, b.row0, a.row2, b.row2}, b' = {a.row1, b.row1, a.row3, b.row3}
__device__ __forceinline__ float a5_xsum4(float x) {
#ifdef A5_XSUM_SHFL
  return xsum4(x);
#else
  float a = x, b = x;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  x = a + b;
  a = x; b = x;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
#endif
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
  static constexpr int OFF_DEL = OFF_LSE + R * 8;   // lse: one (-lse2, -lse2) PAIR per query (an operand of v_pk_fma_f32)
  static constexpr int OFF_RED = OFF_DEL + R * 4;
  static constexpr int RED = KF * 192 * 4 > KF * R * 4 ? KF * 192 * 4 : KF * R * 4;   // column sums | delta partials [KF][R]
  static constexpr int OFF_CSO = OFF_RED + RED;
  static constexpr int OFF_CST = OFF_CSO + NTL * 8 * 4;   // per loader lane: column sums of its 8-column chunk of dO
  static constexpr int LDS = OFF_CST + 64 * 4;            // their totals of the current pair, [64]
  static_assert(NT <= 1024 && R * 4 == NTC && 192 <= NTC && TS >= 2048, "workgroup shape");
  static_assert(NTL % 8 == 0, "a loader lane keeps one column chunk");
};

// 16 rows x 64 columns of fp32 accumulators in MFMA C layout (lane (lr, lg): row lr, columns d * 16 + 4 lg .. + 3 of
// acc[d]) -> bf16 -> a wave-private 2 KiB LDS image -> whole 128-byte rows -> global, 16 bytes per lane.  The scattered
// form (sixteen 8-byte stores per lane, 32 bytes per row and instruction) is store-ISSUE bound: ~10 k cycles per
// (sample, head) pair, a quarter of the first version of this kernel (profiles/r04_attn5_probe.txt).
struct A5Rows {
  char* w[4];       // where the lane puts its 8-byte piece of column block d
  const char* r[2]; // the two 16-byte row pieces it reads back
  uint32_t g[2];    // their byte offsets inside the (sample, head) block of dqkv
  bool ok[2];       // row < L
};
__device__ __forceinline__ void a5_store_rows(const A5Rows& a, const f32x4 (&acc)[4], float mul, char* base) {
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    uint2 w;
    w.x = pack_bf2(acc[d][0] * mul, acc[d][1] * mul);
    w.y = pack_bf2(acc[d][2] * mul, acc[d][3] * mul);
    *reinterpret_cast<uint2*>(a.w[d]) = w;
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const uint4 v = *reinterpret_cast<const uint4*>(a.r[it]);
    if (a.ok[it]) *reinterpret_cast<uint4*>(base + a.g[it]) = v;
  }
}

// Workgroup = KF compute waves + LW loader waves.  The loader waves own the NEXT pair's Q / dO tiles (in registers,
// loaded right after the current pair's tiles became visible, i.e. in flight during the whole pair) and write them
// into the LDS tiles as soon as those are free (dO after phase 1b, Q after phase 2): the compute waves sit at the
// 128-VGPR line of four waves per SIMD and cannot hold 17 prefetch registers on top of K, V, two accumulator sets
// and the fragment operands (first version: 29-74 spilled VGPRs).  Every wave passes the same six barriers per pair.
// BM: 0 = no bias gradients; 1 = column sums of dq / dk / dv by DPP reductions (any L; A/B only); 3 = the k and v
// identities below + DPP column sums of dq (L % 16 == 0); 2 = all three by identities, for L % 16 != 0 (a padded
// query column exists):
//   sum_j dK_j = 0 exactly (shift invariance of the softmax: what autodiff produces there is rounding noise);
//   sum_j dV_j = sum_i dO_i (rows of P sum to 1): column sums of the dO tile, taken by the loader waves from the
//               registers they hold it in;
//   sum_i dQ_i = scale * sum_j cs_j K_j with cs_j = sum_i dS_ij: every key-owning wave keeps cs of its keys in one
//               register and plants it (bf16) in the LAST, padded, query column of the dS^T image - phase 2 then
//               computes exactly this product as column R - 1 of dQ^T, for free.
template <int KF, int LW, int BM>
__global__ __launch_bounds__((KF + LW) * 64) void attn5_bwd_kernel(const bf16* __restrict__ qkv,
                                                                   const bf16* __restrict__ d_o,
                                                                   const float* __restrict__ lse,
                                                                   bf16* __restrict__ dqkv,
                                                                   float* __restrict__ dbias, int L, int H,
                                                                   int npairs, float scale) {
  using C = A5<KF, LW>;
  constexpr int R = C::R, PL = C::PL, TS = C::TS, NTL = C::NTL, NPL = C::NPL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qt = smem;                 // Q tile; K tile in phase 2
  char* Gt = smem + C::OFF_G;      // dO tile
  char* dSt = smem + C::OFF_DS;    // KF tiles: P (bf16, [q][key] planes) after phase 1a, dS^T ([key][q] planes) after 1b
  float* lse_s = reinterpret_cast<float*>(smem + C::OFF_LSE);
  float* del_s = reinterpret_cast<float*>(smem + C::OFF_DEL);
  float* red = reinterpret_cast<float*>(smem + C::OFF_RED);   // BM 1: [KF waves][3][64] column sums; BM 2: [64] (dq)
  float* dpart = red;              // phase 1a -> reduction: [KF][R] partials of delta (the column sums come later)
  float* cso = reinterpret_cast<float*>(smem + C::OFF_CSO);   // BM 2: [loader lane][8] column sums of its dO pieces
  float* cst = reinterpret_cast<float*>(smem + C::OFF_CST);   // BM 2: [64] column sums of the pair's dO = v-bias gradient
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const long ld = 3L * H * DH, ldo = (long)H * DH;
  const float c = scale * LOG2E;
  const int stride = gridDim.x;

  if (wave >= KF) {
    // ================================================================ loader waves ==========================
    const int ll0 = tid - C::NTC;
    a5_u32x4 pq[NPL], pg[NPL];
    float pl[(R + NTL - 1) / NTL];
    auto load_tiles = [&](int pair) __attribute__((always_inline)) {
      const int i = pair / H, h = pair % H;
      // wave-uniform 64-bit bases + 32-bit lane offsets (a tile spans < 1 MB): one address register per piece
      const char* qb_ = reinterpret_cast<const char*>(qkv + (long)i * L * ld + h * DH);
      const char* dob_ = reinterpret_cast<const char*>(d_o + (long)i * L * ldo + h * DH);
      const uint32_t ldb = (uint32_t)ld * 2u, ldob = (uint32_t)ldo * 2u;
      const int ll = a5_opaque(ll0);   // per-pair address arithmetic instead of 40 hoisted (and spilled) registers
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        // branch-free: rows >= L re-read row L - 1 and are zeroed when they are USED (put_tile / put_cso) - with a
        // branch per piece hipcc spilled the first pieces right behind their loads (a vmcnt wait in front of B2)
        const int idx = ll + j * NTL, row = min(idx >> 3, L - 1), pc = idx & 7;
        pq[j] = *reinterpret_cast<const a5_u32x4*>(qb_ + ((uint32_t)row * ldb + (uint32_t)pc * 16u));
        pg[j] = *reinterpret_cast<const a5_u32x4*>(dob_ + ((uint32_t)row * ldob + (uint32_t)pc * 16u));
      }
#pragma unroll
      for (int j = 0; j < (R + NTL - 1) / NTL; ++j) {
        const int q = ll + j * NTL;
        // rows >= L: lse = +inf makes P = exp2(-inf) = 0 without an explicit query mask
        pl[j] = q < L ? lse[(long)pair * L + q] : INFINITY;
      }
    };
    // rows >= L of a tile are zeroed ONCE (zero_pad_rows) and then left alone - nobody else writes them (the K rows
    // that overlay the Q tile in phase 2 are zero there too), so the steady-state put needs no select (a select per
    // piece cost the loader its registers: spills right behind the loads, i.e. a vmcnt wait in front of B2)
    // (a macro, one expansion per register array: an array handed to a lambda BY REFERENCE stays in scratch memory)
#define A5_PUT_TILE(T, arr)                                                                               \
  do {                                                                                                    \
    const int ll_ = a5_opaque(ll0);                                                                       \
    _Pragma("unroll") for (int j = 0; j < NPL; ++j) {                                                     \
      const int idx = ll_ + j * NTL, row = idx >> 3, pc = idx & 7;                                        \
      if (row < L) *reinterpret_cast<a5_u32x4*>((T) + row * 128 + ((pc ^ t64_swz(row)) << 4)) = arr[j];  \
    }                                                                                                     \
  } while (0)
    auto zero_pad_rows = [&](char* T) __attribute__((always_inline)) {   // once, before the first put
      for (int idx = ll0; idx < R * 8; idx += NTL) {
        const int row = idx >> 3, pc = idx & 7;
        if (row >= L) *reinterpret_cast<uint4*>(T + row * 128 + ((pc ^ t64_swz(row)) << 4)) = make_uint4(0, 0, 0, 0);
      }
    };
    auto put_lse = [&]() __attribute__((always_inline)) {
      const int ll = a5_opaque(ll0);
#pragma unroll
      for (int j = 0; j < (R + NTL - 1) / NTL; ++j) {
        const int q = ll + j * NTL;
        if (q < R) {   // minus lse in base-2 units, twice: phase 1a reads the pair with one ds_read_b64
          const float v = -pl[j] * LOG2E;
          *reinterpret_cast<float2*>(lse_s + 2 * q) = make_float2(v, v);
        }
      }
    };
    // column sums of the dO tile held in pg: every piece of a lane is the same 8-column chunk (NTL % 8 == 0), so a
    // lane sums its pieces in registers and leaves 8 partial sums; the compute side adds the NTL / 8 lanes of a chunk
    auto put_cso = [&]() __attribute__((always_inline)) {
      float a[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] = 0.f;
      const int ll = a5_opaque(ll0);
#pragma unroll
      for (int j = 0; j < NPL; ++j) {
        const float m = ((ll + j * NTL) >> 3) < L ? 1.f : 0.f;   // pieces of rows >= L hold a copy of row L - 1
        const uint32_t w[4] = {pg[j][0], pg[j][1], pg[j][2], pg[j][3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[2 * e] = __builtin_fmaf(m, bflo(w[e]), a[2 * e]);
          a[2 * e + 1] = __builtin_fmaf(m, bfhi(w[e]), a[2 * e + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);   // piece by piece: no hoisted unpacking
      }
      float* dst = cso + ll * 8;
      *reinterpret_cast<float4*>(dst) = make_float4(a[0], a[1], a[2], a[3]);
      *reinterpret_cast<float4*>(dst + 4) = make_float4(a[4], a[5], a[6], a[7]);
    };
    // totals of the partial sums (one loader wave, behind a barrier that follows put_cso): cst[64].  Called right
    // behind B1 (the partial sums of THIS pair were written in front of it), off every critical path: the compute waves
    // read cst behind B6, the partial sums are overwritten behind B5.
    auto put_cst = [&]() __attribute__((always_inline)) {
      if (wave == KF) {
        const int d = a5_lane(lane), ch = d >> 3, e = d & 7;
        float t = 0.f;
#pragma unroll
        for (int m = 0; m < NTL / 8; ++m) t += cso[(ch + 8 * m) * 8 + e];
        cst[d] = t;
      }
    };
    int pair = blockIdx.x;
    if (pair < npairs) {
      load_tiles(pair);
      zero_pad_rows(Gt);
      zero_pad_rows(Qt);
      A5_PUT_TILE(Gt, pg);
      put_lse();
      A5_PUT_TILE(Qt, pq);
      if constexpr (BM >= 2) put_cso();
    }
    for (; pair < npairs; pair += stride) {
      __syncthreads();   // B1: this pair's tiles are visible
      const int nxt = pair + stride;
      A5_STAMP(0);
      // issued here, first USED behind B5: a loader wave must not wait for HBM in front of a barrier the compute
      // waves arrive at earlier (the first version multiplied lse here: every B2 waited ~8 k cycles for the loads)
      if (nxt < npairs && !(A5_ABL & 32)) load_tiles(nxt);
      if constexpr (BM >= 2) put_cst();
      A5_STAMP(1);
      __syncthreads();   // B2
      __syncthreads();   // B3
      __syncthreads();   // B4: the dO tile and lse are free
      A5_WSTAMP(3);
      A5_WSTAMP(4);
      __syncthreads();   // B5
      A5_WSTAMP(5);
      A5_STAMP(5);
      if (nxt < npairs) {   // under phase 2 of the compute waves
        A5_PUT_TILE(Gt, pg);
        put_lse();
        if constexpr (BM >= 2) put_cso();   // partial column sums of the NEXT pair's dO
      }
      A5_STAMP(6);
      A5_WSTAMP(6);
      __syncthreads();   // B6: the K tile (= Q tile) is free
      A5_WSTAMP(7);
      A5_STAMP(9);
      if (nxt < npairs) A5_PUT_TILE(Qt, pq);
      A5_STAMP(10);
    }
    return;
  }

  // ================================================================== compute waves ==========================
  bf16x8 k0, k1, v0, v1;
  auto load_k = [&](int pair) __attribute__((always_inline)) {
    const int i = pair / H, h = pair % H;
    const bf16* kb_ = qkv + (long)i * L * ld + (long)H * DH + h * DH;
    const int ln = a5_lane(lane), lr = ln & 15, lg = ln >> 4;
    const int kr = wave * 16 + lr;
    k0 = gfrag(kb_, ld, kr, L, lg * 8); k1 = gfrag(kb_, ld, kr, L, 32 + lg * 8);
  };
  auto load_v = [&](int pair) __attribute__((always_inline)) {
    const int i = pair / H, h = pair % H;
    const bf16* vb_ = qkv + (long)i * L * ld + 2L * H * DH + h * DH;
    const int ln = a5_lane(lane), lr = ln & 15, lg = ln >> 4;
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
    A5_STAMP(0);

    // ---- phase 1a: P^T and the partials of delta.  S^T[key = 4 lg + r][q = lr] of (key fragment `wave`, query
    // fragment f); the operands of fragment f + 1 are read from the LDS before fragment f is computed.  P goes to the
    // LDS as bf16 - block (f, wave) of tile f, [q = lr][4 keys of lane group lg] - where phase 1b fetches it back
    // TRANSPOSED (ds_read_b64_tr_b16: P[q = 4 lg + r][key = lr]) instead of computing S and the exponentials again.
    {
      const int ln = a5_lane(lane), lr = ln & 15, lg = ln >> 4;
      // Lane bases, each an OPAQUE register: every per-fragment address below is `base + compile-time constant`, i.e.
      // the offset field of the LDS instruction.  (Written as t64_row(Qt, f * 16 + lr, lg) hipcc merged f * 16 into
      // the row before the swizzle and re-derived the address of every fragment with VALU instructions: 7 of the 24
      // per tile, on the SIMD whose issue time bounds the kernel.)  f * 16 leaves the swizzle bits of a row alone.
      const int sw = t64_swz(lr);
      const char* qa0 = smem + a5_opaque(lr * 128 + ((lg ^ sw) << 4));         // Q row lr, chunk lg (dO: + OFF_G)
      const char* qa1 = smem + a5_opaque(lr * 128 + (((4 + lg) ^ sw) << 4));   // chunk 4 + lg
      const char* la = smem + a5_opaque(C::OFF_LSE + lr * 8);                  // (-lse, -lse) of query lr
      char* pwr = smem + a5_opaque(C::OFF_DS + lg * PL + (wave * 16 + lr) * 8);   // this wave's block of tile f: + f * TS
      struct Ops { bf16x8 q0, q1, g0, g1; f32x2 nl; };
      auto rd = [&](int f) __attribute__((always_inline)) {
        Ops o;
        o.q0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(qa0 + f * 2048));
        o.q1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(qa1 + f * 2048));
        o.g0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(qa0 + C::OFF_G + f * 2048));
        o.g1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(qa1 + C::OFF_G + f * 2048));
        o.nl = *reinterpret_cast<const f32x2*>(la + f * 128);
        return o;
      };
      // Padded keys (rows >= L of the last fragment; k = v = 0) are masked for free: their S^T accumulators START at
      // -1e30 instead of 0, so exp2 gives P = 0 exactly - for any lse, without a select per score.  Everything
      // downstream (P in the LDS, dS, cs) is then exactly 0 for them.
      const int lim = L - wave * 16 - lg * 4;   // key 4 lg + r of this wave's fragment exists for r < lim
      const f32x4 st0 = f32x4{0 < lim ? 0.f : -1e30f, 1 < lim ? 0.f : -1e30f, 2 < lim ? 0.f : -1e30f,
                              3 < lim ? 0.f : -1e30f};
      const f32x2 c2 = f32x2{c, c};
      // one tile: returns this lane's partial of delta (its four keys of query lr); packed fp32 around the exponentials
      auto go = [&](int f, const Ops& o) __attribute__((always_inline)) {
        f32x4 st = st0, dp = f32x4{0.f, 0.f, 0.f, 0.f};
        st = mfma16(k0, o.q0, st);
        st = mfma16(k1, o.q1, st);
        dp = mfma16(v0, o.g0, dp);
        dp = mfma16(v1, o.g1, dp);
        const f32x2 a01 = __builtin_elementwise_fma(f32x2{st[0], st[1]}, c2, o.nl);
        const f32x2 a23 = __builtin_elementwise_fma(f32x2{st[2], st[3]}, c2, o.nl);
        const f32x4 e = f32x4{__builtin_amdgcn_exp2f(a01[0]), __builtin_amdgcn_exp2f(a01[1]),
                              __builtin_amdgcn_exp2f(a23[0]), __builtin_amdgcn_exp2f(a23[1])};
        f32x2 x2 = f32x2{e[0], e[1]} * f32x2{dp[0], dp[1]};
        x2 = __builtin_elementwise_fma(f32x2{e[2], e[3]}, f32x2{dp[2], dp[3]}, x2);
        *reinterpret_cast<s16x4*>(pwr + f * TS) = pack4(e);
        return x2[0] + x2[1];
      };
      // Partials of FOUR tiles are summed over the four 16-lane rows together: two v_permlane32_swap, one
      // v_permlane16_swap and three adds leave the total of tile f0 + {0, 2, 1, 3}[lg] in lane row lg (a' = {a.lo32,
      // b.lo32}, b' = {a.hi32, b.hi32}; a' = {a.row0, b.row0, a.row2, b.row2}, b' = {a.row1, b.row1, a.row3, b.row3}),
      // every lane writes one partial: 1.5 VALU operations per tile instead of 6 and no masked write.
      float* dw4 = reinterpret_cast<float*>(smem + a5_opaque(C::OFF_RED + (wave * R + ((((lg & 1) << 1) | (lg >> 1)) * 16) + lr) * 4));
      float* dw1 = reinterpret_cast<float*>(smem + a5_opaque(C::OFF_RED + (wave * R + lr) * 4));
      auto red4 = [&](int f0, float xa, float xb, float xc, float xd) __attribute__((always_inline)) {
        asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(xa), "+v"(xb));
        asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(xc), "+v"(xd));
        float t1 = xa + xb, t2 = xc + xd;
        asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(t1), "+v"(t2));
        dw4[f0 * 16] = t1 + t2;
      };
      constexpr int NF = (A5_ABL & 1) ? 1 : KF;
      float xs[4] = {0.f, 0.f, 0.f, 0.f};
      auto done = [&](int f, float x) __attribute__((always_inline)) {
        if (f < (NF & ~3)) {
          xs[f & 3] = x;
          if ((f & 3) == 3) red4(f - 3, xs[0], xs[1], xs[2], xs[3]);
        } else {   // the one to three tiles behind the last group of four
          x = a5_xsum4(x);
          if (lg == 0) dw1[f * 16] = x;
        }
      };
      Ops a = rd(0), b = a;
#pragma unroll A5_UNROLL_1A
      for (int f = 0; f < NF; f += 2) {
        if (f + 1 < NF) b = (A5_ABL & 64) ? a : rd(f + 1);
        done(f, go(f, a));
        if (f + 1 < NF) {
          if (f + 2 < NF) a = rd(f + 2);
          done(f + 1, go(f + 1, b));
        }
      }
    }
    A5_STAMP(1);
    __syncthreads();   // B2
    A5_STAMP(2);
    {   // fixed-order sum of the KF partials of every query row (four lanes per row, quad reduction): deterministic
      const int tq = wave * 64 + a5_lane(lane), row = tq >> 2, part = tq & 3;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < (KF + 3) / 4; ++w)
        if (w * 4 + part < KF) t += dpart[(w * 4 + part) * R + row];
      t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, true));
      t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4E, 0xF, 0xF, true));
      if (part == 0) del_s[row] = -t;   // minus delta: phase 1b starts its dP accumulators from it
    }
    __syncthreads();   // B3: delta complete; the partials are dead, the dS^T tiles may be written
    A5_WSTAMP(0);
    A5_STAMP(3);

    // ---- phase 1b: dV^T, dK^T of this wave's key fragment; dS^T of every (query fragment, key fragment) -> LDS
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      dk[d] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[d] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // BM 2: cs[key = lr] = sum over the queries of the bf16 dS the other products use, accumulated by the matrix
    // pipe: ones[16 x q] . dS[q][key] - every row of the result is the column sum (one MFMA per fragment pair
    // instead of four VALU adds per fragment and a cross-lane sum at the end)
    f32x4 csacc = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      const int ln = a5_lane(lane), lr = ln & 15, lg = ln >> 4;
      // Lane bases (opaque registers, see phase 1a): every address of the loop is base + compile-time constant.
      const int sw = t64_swz(lr);
      const char* ga0 = smem + a5_opaque(C::OFF_G + lr * 128 + ((lg ^ sw) << 4));         // dO row lr, chunk lg
      const char* ga1 = smem + a5_opaque(C::OFF_G + lr * 128 + (((4 + lg) ^ sw) << 4));   // chunk 4 + lg
      const char* da = smem + a5_opaque(C::OFF_DEL + lg * 16);                            // -delta of rows 4 lg .. + 3
      const char* ptr_ = smem + a5_opaque(C::OFF_DS + (lr & 3) * PL + (wave * 16 + 4 * lg + (lr >> 2)) * 8);   // a5_tr of this wave's P block
      char* pwr = smem + a5_opaque(C::OFF_DS + lg * PL + (wave * 16 + lr) * 8);           // dS^T block (over the P block)
      // transposed operand reads of the Q / dO tiles (t64_tr): the chunk position is (2 d + b) ^ swizzle - the four d
      // blocks sit at a lane-dependent permutation of {0, 32, 64, 96} bytes, so each gets its own base register
      const int trow = 4 * lg + (lr >> 2), tsw = t64_swz(trow);
      const char* tq[4];
#pragma unroll
      for (int d = 0; d < 4; ++d)
        tq[d] = smem + a5_opaque(trow * 128 + (((d * 2 + ((lr >> 1) & 1)) ^ tsw) << 4) + ((lr & 1) << 3));
      auto trd = [&](const char* p) __attribute__((always_inline)) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
      };
      auto trd2 = [&](const char* p) __attribute__((always_inline)) {   // rows 4 lg .. of two consecutive fragments
        const s16x4 a = trd(p), b = trd(p + 2048);
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
      };
      // P and dS of query fragment f against the wave's key fragment: pw = bf16 P[q = 4 lg + r][key = lr] (from phase
      // 1a, transposed read), ds[r] = dS of the same elements.  dP - delta comes out of the MFMA: the accumulator starts
      // at -delta of its rows.
      auto pds = [&](int f, s16x4& pw, f32x4& ds) __attribute__((always_inline)) {
        const bf16x8 g0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ga0 + f * 2048));
        const bf16x8 g1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ga1 + f * 2048));
        const float4 d4 = *reinterpret_cast<const float4*>(da + f * 64);
        pw = trd(ptr_ + f * TS);
        f32x4 dp = f32x4{d4.x, d4.y, d4.z, d4.w};   // -delta of the fragment's rows
        dp = mfma16(g0, v0, dp);   // dP[q][key] - delta[q],  dP = sum_d dO[q][d] V[key][d]
        dp = mfma16(g1, v1, dp);
        const uint2 w = __builtin_bit_cast(uint2, pw);
        const f32x2 s01 = f32x2{bflo(w.x), bfhi(w.x)} * f32x2{dp[0], dp[1]};
        const f32x2 s23 = f32x2{bflo(w.y), bfhi(w.y)} * f32x2{dp[2], dp[3]};
        ds = f32x4{s01[0], s01[1], s23[0], s23[1]};
        // dS^T[key][q = f * 16 + 4 lg .. + 3]: plane lg of tile f, 8 bytes per key - the bytes this wave's P block of
        // the tile occupied (read above; LDS operations of one wave complete in order)
        *reinterpret_cast<s16x4*>(pwr + f * TS) = pack4(ds);
      };
      bf16x8 gtr[4], qtr[4];   // transposed dO / Q operands of the current fragment pair
#pragma unroll A5_UNROLL_1B
      for (int ip = 0; ip < ((A5_ABL & 2) ? 1 : KF / 2); ++ip) {
        s16x4 pp[2];
        f32x4 ds[2];
        pds(2 * ip, pp[0], ds[0]);
        if (A5_SB_1B) __builtin_amdgcn_sched_barrier(0);   // one fragment's operand reads at a time (register pressure)
        pds(2 * ip + 1, pp[1], ds[1]);
        const bf16x8 pf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(pp[0], pp[1], 0, 1, 2, 3, 4, 5, 6, 7));
        const bf16x8 dsf = pack8(ds[0], ds[1]);
        if constexpr (BM == 2) {
          const s16x8 ones = s16x8{0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};   // bf16 1.0
          csacc = mfma16(__builtin_bit_cast(bf16x8, ones), dsf, csacc);
        }
        if (A5_SB_1B) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          if (!(A5_ABL & 64) || !(ip & 1)) {
            gtr[d] = trd2(tq[d] + C::OFF_G + (2 * ip) * 2048);
            qtr[d] = trd2(tq[d] + (2 * ip) * 2048);
          }
          dv[d] = mfma16(gtr[d], pf, dv[d]);    // D[d = 4 lg + r][key = lr]
          dk[d] = mfma16(qtr[d], dsf, dk[d]);
        }
      }
      if constexpr (KF & 1) {
        s16x4 pf;
        f32x4 ds;
        pds(KF - 1, pf, ds);
        const s16x4 dsf = pack4(ds);
        if constexpr (BM == 2) csacc = mfma16k16(s16x4{0x3f80, 0x3f80, 0x3f80, 0x3f80}, dsf, csacc);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          dv[d] = mfma16k16(trd(tq[d] + C::OFF_G + (KF - 1) * 2048), pf, dv[d]);
          dk[d] = mfma16k16(trd(tq[d] + (KF - 1) * 2048), dsf, dk[d]);
        }
      }
      if constexpr (BM == 2) {
        // cs of key lr (all four lane groups) -> the padded query column R - 1 of the dS^T image: element 3 of
        // plane 3 of the last tile.  cs is rounded to bf16 HERE (advisor r4: documented, include/bvhip.h at
        // bv_attn_bwd): it is the same 2^-9 rounding every dS element of this image carries into dQ, so the q-bias
        // gradient (column R - 1 of dQ^T) has the dQ rows' precision; the k-bias gradient is exactly 0 by identity.  This wave wrote that word itself (zeros: query R - 1 does not exist), LDS
        // operations of one wave complete in order.
        asm volatile("s_nop 15\n\ts_nop 3" : "+v"(csacc));   // MFMA result -> VALU read behind control flow
        const float cs = csacc[0];
        if (lg == 0)
          *reinterpret_cast<bf16*>(dSt + (KF - 1) * TS + 3 * PL + (wave * 16 + lr) * 8 + 6) = (bf16)cs;
      }
    }
    A5_WSTAMP(1);
    a5_drain(dk[0], dk[1], dk[2], dk[3]);
    a5_drain(dv[0], dv[1], dv[2], dv[3]);
    // V is dead: the next pair's rows are in flight from here.  (With global loads in flight B4 and B5 release 1.7 k /
    // 2.8 k cycles after their last arrival instead of ~0.4 k - per-wave stamps, profiles/r04_attn5_probe.txt - but every
    // later position of the two requests measured SLOWER end to end: +2..8 %, profiles/NOTES_r04.md.)
    if (nxt < npairs) load_v(nxt);
    A5_STAMP(4);
    A5_WSTAMP(2);
    __syncthreads();   // B4: every dS^T tile complete; nobody reads the Q / dO tiles any more
    A5_WSTAMP(3);
    A5_STAMP(5);

    // ---- K fragments -> the Q tile (T64 image, the A operand of phase 2)
    {
      const int ln = a5_lane(lane), lr = ln & 15, lg = ln >> 4;
      const int row = wave * 16 + lr;
      *reinterpret_cast<uint4*>(Qt + row * 128 + ((lg ^ t64_swz(row)) << 4)) = __builtin_bit_cast(uint4, k0);
      *reinterpret_cast<uint4*>(Qt + row * 128 + (((4 + lg) ^ t64_swz(row)) << 4)) = __builtin_bit_cast(uint4, k1);
      if (nxt < npairs) load_k(nxt);   // the next pair's K rows replace the finished ones
      if constexpr (BM == 1) {
        // column sums over this fragment's live keys (fp32, before the bf16 rounding): one row of `red` per wave
        const bool live = row < L;
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
    A5_STAMP(6);
    A5_WSTAMP(4);
    __syncthreads();   // B5: K tile complete
    A5_WSTAMP(5);
    A5_STAMP(7);

    // ---- phase 2: dQ^T[d][q] of query fragment `wave` = sum_key K^T[d][key] dS^T[key][q]
    {
      const int ln = a5_lane(lane), lr = ln & 15, lg = ln >> 4;
      char* T = dSt + wave * TS;
      // lane bases (opaque registers, see phase 1a): the dS^T planes of this wave's tile and the four d blocks of the K
      // tile (transposed reads, lane-dependent chunk permutation as in phase 1b)
      const char* sa = smem + a5_opaque(C::OFF_DS + wave * TS + (lr & 3) * PL + (4 * lg + (lr >> 2)) * 8);
      const int trow = 4 * lg + (lr >> 2), tsw = t64_swz(trow);
      const char* tk[4];
#pragma unroll
      for (int d = 0; d < 4; ++d)
        tk[d] = smem + a5_opaque(trow * 128 + (((d * 2 + ((lr >> 1) & 1)) ^ tsw) << 4) + ((lr & 1) << 3));
      auto trd = [&](const char* p) __attribute__((always_inline)) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
      };
      auto trd2 = [&](const char* p, int step) __attribute__((always_inline)) {   // the same rows of two consecutive fragments
        const s16x4 a = trd(p), b = trd(p + step);
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
      };
      f32x4 dq[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) dq[d] = f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x8 ktr[4];   // transposed K operands of the current fragment pair
#pragma unroll A5_UNROLL_P2
      for (int fp = 0; fp < ((A5_ABL & 4) ? 1 : KF / 2); ++fp) {
        const bf16x8 dsf = trd2(sa + fp * 256, 128);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          if (!(A5_ABL & 64) || !(fp & 1)) ktr[d] = trd2(tk[d] + fp * 4096, 2048);
          dq[d] = mfma16(ktr[d], dsf, dq[d]);
        }
      }
      if constexpr (KF & 1) {
        const s16x4 dsf = trd(sa + (KF - 1) * 128);
#pragma unroll
        for (int d = 0; d < 4; ++d) dq[d] = mfma16k16(trd(tk[d] + (KF - 1) * 2048), dsf, dq[d]);
      }
      a5_drain(dq[0], dq[1], dq[2], dq[3]);
      if constexpr (BM == 1 || BM == 3) {
        // padded query rows are exactly 0 (their dS^T columns are)
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float t = rowsum16(dq[d][r]);
            if (lr == 0) red[wave * 192 + d * 16 + lg * 4 + r] = t * scale;
          }
      }
      if constexpr (BM == 2) {
        // column R - 1 of dQ^T (lane lr = 15 of the last wave) = sum_key cs_key K[key][:] = the q-bias gradient / scale
        if (wave == KF - 1 && lr == 15) {
#pragma unroll
          for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[d * 16 + lg * 4 + r] = dq[d][r] * scale;
        }
      }
      // ---- the wave's dQ rows, then its dK / dV rows, through its own (now dead) dS^T tile as whole rows (a5_store_rows;
      // the lane's LDS positions and its offset inside the (sample, head) block are the same for all three)
      const bool st = !(A5_ABL & 16);
      char* outb = reinterpret_cast<char*>(dqkv + (long)i * L * ld + h * DH);   // wave-uniform
      A5Rows rw;
#pragma unroll
      for (int d = 0; d < 4; ++d)
        rw.w[d] = T + a5_opaque(lr * 128 + (((d * 2 + (lg >> 1)) ^ (lr & 7)) << 4) + (lg & 1) * 8);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int row = it * 8 + (ln >> 3), ch = ln & 7;
        rw.r[it] = T + a5_opaque(row * 128 + ((ch ^ (row & 7)) << 4));
        rw.ok[it] = st && wave * 16 + row < L;
        rw.g[it] = (uint32_t)a5_opaque((wave * 16 + row) * (int)(ld * 2) + ch * 16);
      }
      a5_store_rows(rw, dq, scale, outb);
      a5_store_rows(rw, dk, scale, outb + (long)H * DH * 2);
      a5_store_rows(rw, dv, 1.0f, outb + 4L * H * DH);
    }
    A5_STAMP(8);
    A5_WSTAMP(6);
    __syncthreads();   // B6: the K tile, the dS^T tiles and `red` are free / complete
    A5_WSTAMP(7);
    A5_STAMP(9);
    if constexpr (BM != 0) {
      // per-(sample, head) column sums -> dbias[i][which][h][:]; the host sums over samples
      if (wave < 3) {
        const int which = wave, d = a5_lane(lane), tid = wave * 64 + d;
        float t = 0.f;
        if constexpr (BM == 1) {
#pragma unroll
          for (int w = 0; w < KF; ++w) t += red[w * 192 + tid];
        } else if constexpr (BM == 3) {   // q: column sums of the dQ rows; k: 0; v: column sums of dO
          if (which == 0) {
#pragma unroll
            for (int w = 0; w < KF; ++w) t += red[w * 192 + d];
          }
          if (which == 2) t = cst[d];
        } else {   // q: column R - 1 of dQ^T; k: 0; v: column sums of dO (the loader waves' totals)
          t = which == 0 ? red[d] : which == 2 ? cst[d] : 0.f;
        }
        dbias[((long)i * 3 * H + (long)which * H + h) * DH + d] = t;
      }
    }
    A5_STAMP(10);
  }
}

template <int KF, int LW>
int launch_bwd5(const void* qkv, const void* d_o, const float* lse, void* dqkv, float* dbias, int n, int L, int H,
                hipStream_t s, bool bias_dpp) {
  using C = A5<KF, LW>;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  // workgroups per CU: LDS (160 KiB) and 5 waves per SIMD (<= 96 VGPRs: the KF = 4 instantiations; 13 + 3 waves fill a CU)
  int per_cu = (160 * 1024) / C::LDS;
  if (per_cu > 20 / (KF + LW)) per_cu = 20 / (KF + LW);
  if (per_cu < 1) per_cu = 1;
  const int npairs = n * H;
  const int grid = npairs < cus * per_cu ? npairs : cus * per_cu;
  auto go = [&](auto kern) __attribute__((always_inline)) {
    if (C::LDS > 65536)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS, s, (const bf16*)qkv, (const bf16*)d_o, lse, (bf16*)dqkv,
                       dbias, L, H, npairs, 0.125f);
  };
  if (!dbias) go(attn5_bwd_kernel<KF, LW, 0>);
  else if (bias_dpp) go(attn5_bwd_kernel<KF, LW, 1>);   // A/B (BV_OPT_ATTN_CFG bit 256): DPP column sums where the identities apply
  else if (L & 15) go(attn5_bwd_kernel<KF, LW, 2>);
  else go(attn5_bwd_kernel<KF, LW, 3>);
  return bv_check_launch("bv_attn_bwd(one launch)");
}

}  // namespace

// Entry used by bv_attn3_bwd (attention3.hip) for unmasked sequences of at most 208 tokens; returns -100 when the
// shape is not covered (the caller keeps the two-launch path).
int bv_attn5_bwd(const void* qkv, const void* d_o, const float* lse, float* delta, void* dqkv, float* dbias, int n,
                 int L, int H, void* stream, bool bias_dpp) {
  hipStream_t s = (hipStream_t)stream;
  (void)delta;   // scratch of the two-launch path; the exact delta never leaves the LDS here
  if (L <= 64) return launch_bwd5<4, 1>(qkv, d_o, lse, dqkv, dbias, n, L, H, s, bias_dpp);
  if (L > 192 && L <= 208) return launch_bwd5<13, 3>(qkv, d_o, lse, dqkv, dbias, n, L, H, s, bias_dpp);
  return -100;
}
