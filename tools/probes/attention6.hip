// NOT IN THE LIBRARY (round 6): built, bit-for-bit deterministic, as accurate as attention5.hip (q/k/v gradient rel-L2
// vs fp64 0.0027, bias gradients equal to 2e-6) - and within +-3 % of its time (profiles/r06_attn6_ab.txt: 0.95-1.05x);
// superseded / non-winning kernels do not ship.  For the parity run it sat in libbvhip behind BV_OPT_ATTN_CFG bit 512
// (tools/attn6_ab.py).  What the phase stamps say (same file): with 8 waves the operand LDS reads fall from 13 to 8
// wave-reads per tile, but phases 1a / 1b take the SAME time on the SIMD that hosts 4 of the 13 fragments (9 k / 12 k
// cycles; attention5: 8.5 k / 11.5 k) and the SIMDs with 3 fragments are barely faster (7.6 k / 11 k): the phases are
// bound by what the waves share, not by the per-wave operand reads the -20 % ablation removed (that ablation also
// halved the LDS INSTRUCTIONS of a 4-wave SIMD).  Static priority moves the straggler from the younger to the older
// wave of a SIMD without shortening the phase.  Lessons kept: no load inside a branch and no select right behind a
// load (vmcnt(0) stalls); request the next pair's tiles behind the last first-use of loaded registers.
//
// Fused self-attention BACKWARD in one launch, fifth generation: EIGHT SYMMETRIC WAVES, one or two key fragments per
// wave, no loader waves.  gfx950, Dh = 64, 193 <= L <= 208 (13 key fragments: ViT-B/16 at 224 px), unmasked.  Same
// mathematics, reference call sites and phase structure as attention5.hip (flax nn.MultiHeadDotProductAttention inside
// big_vision/models/vit.py:93-98; backward = jax.value_and_grad, trainers/proj/image_text/siglip.py:311) - read that file
// first: phases 1a / 1b / 2, the P -> dS^T image in the LDS ("plane layout"), the exact in-kernel delta, the bias
// gradients by identities (BM 2), whole-row stores through a wave-private LDS image.
//
// Why.  attention5 is LDS-BANDWIDTH bound: 13 key-owning waves each read the whole Q and dO tiles as row operands
// (phase 1a), again transposed (phase 1b) and the whole K tile (phase 2): ~2.3 MB of LDS reads per (sample, head) pair =
// 18 k of its 33.5 k cycles at 128 B/clk; halving the operand reads measured -18..22 % by ablation
// (profiles/r06_attn5_lds_share_ablation.txt).  A wave that owns TWO key fragments feeds every operand it reads to two
// MFMA chains.  Two fragments need ~190 VGPRs, i.e. at most 8 waves per CU (256 each) - which leaves no room for the
// three loader waves of attention5 (10 waves -> 168 VGPRs), and one loader wave cannot hand 53 KB over per pair.  So:
//   * 8 waves; waves 0-4 own key fragments (2w, 2w + 1), waves 5-7 own fragments 10, 11, 12 (13 = 5 x 2 + 3); the same
//     map assigns the QUERY fragments of phase 2.  8 wave-reads of every operand tile instead of 13.
//   * every wave also prefetches 1/8 of the NEXT pair's Q / dO tiles into 32 spare registers right after barrier B1 (in
//     flight during the whole pair, like the loader waves' registers) and writes its pieces into the LDS tiles when they
//     are free: dO / lse after B5 (nobody reads them after B4), Q after B6.
//   * the two kinds of wave run the same barrier sequence from two instantiations of one body (NF = 1 | 2): the branch is
//     taken once per launch, not per phase.
// Bias gradients: BM 0 = none, BM 2 = by the identities of attention5.hip (needs a padded query column: L % 16 != 0).
// Everything else (L = 208 with bias gradients, L <= 64, key padding) stays on attention5.hip / attention3.hip.
#include <type_traits>
#include "../../big_vision_amd/csrc/attn_common.h"
#include "../../big_vision_amd/csrc/bvhip_internal.h"

// A6_STAMPS (tools/probes/attn6_probe.hip only): s_memtime of every wave of workgroup 100 at the phase boundaries of
// its third pair, [wave][16]
#ifdef A6_STAMPS
__device__ long* g_a6_stamps;
#define A6_STAMP(k)                                                                        \
  do {                                                                                     \
    if (lane == 0 && blockIdx.x == 100 && pair == blockIdx.x + 2 * stride)                 \
      g_a6_stamps[wave * 16 + (k)] = __builtin_amdgcn_s_memtime();                         \
  } while (0)
#else
#define A6_STAMP(k)
#endif

namespace {
using namespace bvattn;
typedef unsigned int a6_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void a6_drain(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  // wait states between the last MFMAs of a loop and VALU reads of their results behind control flow (attention3.hip)
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// opaque copies of lane-dependent values: every phase derives its addresses from its own copy, so hipcc cannot hoist
// dozens of address registers out of the pair loop (attention5.hip)
__device__ __forceinline__ int a6_opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}
__device__ __forceinline__ int a6_lane(int lane) { return a6_opaque(lane) & 63; }
// Row fragment of K / V WITHOUT control flow and without a select: rows >= L read row L - 1.  A load inside a branch
// makes hipcc wait with vmcnt(0) at the first use of ANY loaded register (the first version stalled every phase 1a on
// the next pair's tile prefetch issued a moment earlier); a select right behind the load waits for it on the spot.  The
// copies of row L - 1 are harmless: padded keys are masked by the -1e30 start of their S^T accumulators (P = dS = 0
// exactly, so their K / V values never reach a sum), their dK / dV rows are not stored, and the rows they leave in the
// K tile (= the next pair's padded QUERY rows) meet lse = +inf (P = 0) and zero dO rows.
__device__ __forceinline__ bf16x8 a6_gfrag(const bf16* base, long ld, int row, int L, int col) {
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(base + (long)min(row, L - 1) * ld + col));
}
// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() drains vmcnt too: with the next pair's tile / K / V
// loads in flight every barrier would wait for HBM (that is what made B4 / B5 of attention5.hip release 1.7 k / 2.8 k
// cycles after their last arrival).  The compiler still waits for a loaded register where it is first used.
__device__ __forceinline__ void a6_barrier() {
#ifdef A6_SYNCTHREADS
  __syncthreads();
#else
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#endif
}
__device__ __forceinline__ float a6_xsum4(float x) {   // sum over the four 16-lane rows (attention5.hip a5_xsum4)
  float a = x, b = x;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  x = a + b;
  a = x; b = x;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}

struct A6 {
  static constexpr int KF = 13;                // key / query fragments
  static constexpr int NW = 8;                 // waves
  static constexpr int NTWO = KF - NW;         // waves 0 .. NTWO - 1 own two fragments
  static constexpr int R = KF * 16;            // padded rows of a tile
  static constexpr int NT = NW * 64;
  static constexpr int NP = (R * 8 + NT - 1) / NT;   // 16-byte pieces of one tile per lane (the next pair's prefetch)
  static constexpr int PL = R * 8 + 64;        // bytes of one q-quad plane of a dS^T tile
  static constexpr int TS = 4 * PL;            // bytes of one dS^T tile (one query fragment)
  static constexpr int OFF_G = R * 128;
  static constexpr int OFF_DS = 2 * R * 128;
  static constexpr int OFF_LSE = OFF_DS + KF * TS;
  static constexpr int OFF_DEL = OFF_LSE + R * 8;
  static constexpr int OFF_RED = OFF_DEL + R * 4;      // delta partials [NW][R]; later [64]: column R - 1 of dQ^T
  static constexpr int OFF_CSO = OFF_RED + NW * R * 4; // [NW][64]: per-wave column sums of the NEXT pair's dO pieces
  static constexpr int OFF_CST = OFF_CSO + NW * 64 * 4;   // [64]: their totals for the current pair
  static constexpr int LDS = OFF_CST + 64 * 4;
  static_assert(LDS <= 160 * 1024 && NT % 8 == 0 && 2 * R <= NT, "workgroup shape");
};

struct A6Rows {       // attention5.hip A5Rows: whole 128-byte rows through a wave-private 2 KiB LDS image
  char* w[4];
  const char* r[2];
  uint32_t g[2];
  bool ok[2];
};
__device__ __forceinline__ void a6_store_rows(const A6Rows& a, const f32x4 (&acc)[4], float mul, char* base) {
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

template <int BM>
__global__ __launch_bounds__(A6::NT) void attn6_bwd_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ d_o,
                                                            const float* __restrict__ lse, bf16* __restrict__ dqkv,
                                                            float* __restrict__ dbias, int L, int H, int npairs,
                                                            float scale) {
  using C = A6;
  constexpr int KF = C::KF, R = C::R, PL = C::PL, TS = C::TS, NT = C::NT, NP = C::NP, NW = C::NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qt = smem;                 // Q tile; K tile in phase 2
  char* Gt = smem + C::OFF_G;      // dO tile
  char* dSt = smem + C::OFF_DS;    // KF tiles: P after phase 1a, dS^T after 1b (plane layout, attention5.hip)
  float* lse_s = reinterpret_cast<float*>(smem + C::OFF_LSE);
  float* del_s = reinterpret_cast<float*>(smem + C::OFF_DEL);
  float* red = reinterpret_cast<float*>(smem + C::OFF_RED);
  float* cso = reinterpret_cast<float*>(smem + C::OFF_CSO);
  float* cst = reinterpret_cast<float*>(smem + C::OFF_CST);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long ld = 3L * H * DH, ldo = (long)H * DH;
  const float c = scale * LOG2E;
  const int stride = gridDim.x;

  // ---- the next pair's tiles: this lane's pieces (prefetch registers) ------------------------------------------------
  a6_u32x4 pq[NP], pg[NP];
  float pl = 0.f;
  auto load_tiles = [&](int pair) __attribute__((always_inline)) {
    const int i = pair / H, h = pair % H;
    const char* qb_ = reinterpret_cast<const char*>(qkv + (long)i * L * ld + h * DH);
    const char* dob_ = reinterpret_cast<const char*>(d_o + (long)i * L * ldo + h * DH);
    const uint32_t ldb = (uint32_t)ld * 2u, ldob = (uint32_t)ldo * 2u;
    const int ll = a6_opaque(tid);
    pl = lse[(long)pair * L + min(ll, L - 1)];   // FIRST (put_lse needs it while younger K loads are in flight); no branch around the load, no use before put_lse
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      // branch-free: rows >= L re-read row L - 1 and are never written to the tile
      const int idx = ll + j * NT, row = min(idx >> 3, L - 1), pc = idx & 7;
      pq[j] = *reinterpret_cast<const a6_u32x4*>(qb_ + ((uint32_t)row * ldb + (uint32_t)pc * 16u));
      pg[j] = *reinterpret_cast<const a6_u32x4*>(dob_ + ((uint32_t)row * ldob + (uint32_t)pc * 16u));
    }
  };
#define A6_PUT_TILE(T, arr)                                                                               \
  do {                                                                                                    \
    const int ll_ = a6_opaque(tid);                                                                       \
    _Pragma("unroll") for (int j = 0; j < NP; ++j) {                                                      \
      const int idx = ll_ + j * NT, row = idx >> 3, pc = idx & 7;                                         \
      if (row < L) *reinterpret_cast<a6_u32x4*>((T) + row * 128 + ((pc ^ t64_swz(row)) << 4)) = arr[j];  \
    }                                                                                                     \
  } while (0)
  auto zero_pad_rows = [&](char* T) __attribute__((always_inline)) {   // once: nobody ever writes rows >= L again
    for (int idx = tid; idx < R * 8; idx += NT) {
      const int row = idx >> 3, pc = idx & 7;
      if (row >= L) *reinterpret_cast<uint4*>(T + row * 128 + ((pc ^ t64_swz(row)) << 4)) = make_uint4(0, 0, 0, 0);
    }
  };
  auto put_lse = [&]() __attribute__((always_inline)) {
    const int q = a6_opaque(tid);
    if (q < R) {
      const float v = q < L ? -pl * LOG2E : -INFINITY;   // rows >= L: lse = +inf makes P = exp2(-inf) = 0 without a query mask
      *reinterpret_cast<float2*>(lse_s + 2 * q) = make_float2(v, v);
    }
  };
  // column sums of the dO pieces in pg (the v-bias gradient, sum_j dV_j = sum_i dO_i): every piece of a lane is the same
  // 8-column chunk (NT % 8 == 0); the 8 lanes of a wave that hold one chunk are summed by lane exchanges
  auto put_cso = [&]() __attribute__((always_inline)) {
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = 0.f;
    const int ll = a6_opaque(tid);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const float m = ((ll + j * NT) >> 3) < L ? 1.f : 0.f;   // pieces of rows >= L hold a copy of row L - 1
      const uint32_t w[4] = {pg[j][0], pg[j][1], pg[j][2], pg[j][3]};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[2 * e] = __builtin_fmaf(m, bflo(w[e]), a[2 * e]);
        a[2 * e + 1] = __builtin_fmaf(m, bfhi(w[e]), a[2 * e + 1]);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[e] += __shfl_xor(a[e], 8, 64);
      a[e] += __shfl_xor(a[e], 16, 64);
      a[e] += __shfl_xor(a[e], 32, 64);
    }
    const int ln = a6_lane(lane);
    if (ln < 8) {
      float* dst = cso + wave * 64 + ln * 8;
      *reinterpret_cast<float4*>(dst) = make_float4(a[0], a[1], a[2], a[3]);
      *reinterpret_cast<float4*>(dst + 4) = make_float4(a[4], a[5], a[6], a[7]);
    }
  };

  int pair = blockIdx.x;
  if (pair < npairs) {
    load_tiles(pair);
    zero_pad_rows(Gt);
    zero_pad_rows(Qt);
    A6_PUT_TILE(Gt, pg);
    put_lse();
    A6_PUT_TILE(Qt, pq);
    if constexpr (BM == 2) put_cso();
  }

  // ---- one body for both kinds of wave: NF key (and query) fragments, the first one is fragment f0 ------------------
  auto run = [&](auto nf_c) __attribute__((always_inline)) {
    constexpr int NF = decltype(nf_c)::value;
    const int f0 = NF == 2 ? 2 * wave : wave + C::NTWO;
    bf16x8 k0[NF], k1[NF], v0[NF], v1[NF];
    auto load_k = [&](int pr) __attribute__((always_inline)) {
      const int i = pr / H, h = pr % H;
      const bf16* kb_ = qkv + (long)i * L * ld + (long)H * DH + h * DH;
      const int ln = a6_lane(lane), lr = ln & 15, lg = ln >> 4;
#pragma unroll
      for (int x = 0; x < NF; ++x) {
        const int kr = (f0 + x) * 16 + lr;
        k0[x] = a6_gfrag(kb_, ld, kr, L, lg * 8); k1[x] = a6_gfrag(kb_, ld, kr, L, 32 + lg * 8);
      }
    };
    auto load_v = [&](int pr) __attribute__((always_inline)) {
      const int i = pr / H, h = pr % H;
      const bf16* vb_ = qkv + (long)i * L * ld + 2L * H * DH + h * DH;
      const int ln = a6_lane(lane), lr = ln & 15, lg = ln >> 4;
#pragma unroll
      for (int x = 0; x < NF; ++x) {
        const int kr = (f0 + x) * 16 + lr;
        v0[x] = a6_gfrag(vb_, ld, kr, L, lg * 8); v1[x] = a6_gfrag(vb_, ld, kr, L, 32 + lg * 8);
      }
    };
    int pair = blockIdx.x;
    if (pair < npairs) {
      load_k(pair);
      load_v(pair);
    }
    for (; pair < npairs; pair += stride) {
      const int i = pair / H, h = pair % H;
      const int nxt = pair + stride;
      a6_barrier();   // B1: this pair's tiles (and the per-wave dO column sums) are visible
      A6_STAMP(0);
      if constexpr (BM == 2) {
        if (wave == NW - 1) {              // totals of the dO column sums (read behind B6; cso is rewritten behind B5)
          const int d = a6_lane(lane);
          float t = 0.f;
#pragma unroll
          for (int w = 0; w < NW; ++w) t += cso[w * 64 + d];
          cst[d] = t;
        }
      }

      // ---- phase 1a: P^T of the wave's key fragments against every query fragment, partials of delta ----------------
      {
        const int ln = a6_lane(lane), lr = ln & 15, lg = ln >> 4;
        const int sw = t64_swz(lr);
        const char* qa0 = smem + a6_opaque(lr * 128 + ((lg ^ sw) << 4));         // Q row lr, chunk lg (dO: + OFF_G)
        const char* qa1 = smem + a6_opaque(lr * 128 + (((4 + lg) ^ sw) << 4));   // chunk 4 + lg
        const char* la = smem + a6_opaque(C::OFF_LSE + lr * 8);                  // (-lse, -lse) of query lr
        char* pwr = smem + a6_opaque(C::OFF_DS + lg * PL + (f0 * 16 + lr) * 8);  // P block of (tile f, fragment f0): + f * TS (+ 128: f0 + 1)
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
        // padded keys (rows >= L; k = v = 0): their S^T accumulators start at -1e30, so P = 0 exactly for any lse
        f32x4 st0[NF];
#pragma unroll
        for (int x = 0; x < NF; ++x) {
          const int lim = L - (f0 + x) * 16 - lg * 4;
          st0[x] = f32x4{0 < lim ? 0.f : -1e30f, 1 < lim ? 0.f : -1e30f, 2 < lim ? 0.f : -1e30f, 3 < lim ? 0.f : -1e30f};
        }
        const f32x2 c2 = f32x2{c, c};
        // T query fragments (f, f + 1) x NF key fragments = 2 .. 4 independent chains, written breadth first: with only
        // two waves per SIMD the matrix pipe and the transcendental unit are kept busy by interleaving the chains of ONE
        // wave (the first version computed tile after tile and spent 14 k cycles here, attention5's 4 waves per SIMD 8.5 k)
        auto goT = [&](int f, const Ops (&o)[2], float (&part)[2], auto t_c) __attribute__((always_inline)) {
          constexpr int T = decltype(t_c)::value;
          f32x4 st[T][NF], dp[T][NF];
#pragma unroll
          for (int t = 0; t < T; ++t)
#pragma unroll
            for (int x = 0; x < NF; ++x) st[t][x] = mfma16(k0[x], o[t].q0, st0[x]);
#pragma unroll
          for (int t = 0; t < T; ++t)
#pragma unroll
            for (int x = 0; x < NF; ++x) dp[t][x] = mfma16(v0[x], o[t].g0, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
          for (int t = 0; t < T; ++t)
#pragma unroll
            for (int x = 0; x < NF; ++x) st[t][x] = mfma16(k1[x], o[t].q1, st[t][x]);
#pragma unroll
          for (int t = 0; t < T; ++t)
#pragma unroll
            for (int x = 0; x < NF; ++x) dp[t][x] = mfma16(v1[x], o[t].g1, dp[t][x]);
#pragma unroll
          for (int t = 0; t < T; ++t) {
            part[t] = 0.f;
#pragma unroll
            for (int x = 0; x < NF; ++x) {
              const f32x2 a01 = __builtin_elementwise_fma(f32x2{st[t][x][0], st[t][x][1]}, c2, o[t].nl);
              const f32x2 a23 = __builtin_elementwise_fma(f32x2{st[t][x][2], st[t][x][3]}, c2, o[t].nl);
              const f32x4 e = f32x4{__builtin_amdgcn_exp2f(a01[0]), __builtin_amdgcn_exp2f(a01[1]),
                                    __builtin_amdgcn_exp2f(a23[0]), __builtin_amdgcn_exp2f(a23[1])};
              f32x2 x2 = f32x2{e[0], e[1]} * f32x2{dp[t][x][0], dp[t][x][1]};
              x2 = __builtin_elementwise_fma(f32x2{e[2], e[3]}, f32x2{dp[t][x][2], dp[t][x][3]}, x2);
              *reinterpret_cast<s16x4*>(pwr + (f + t) * TS + x * 128) = pack4(e);
              part[t] += x2[0] + x2[1];
            }
          }
        };
        // partials of FOUR tiles summed over the four lane rows together (attention5.hip red4)
        float* dw4 = reinterpret_cast<float*>(smem + a6_opaque(C::OFF_RED + (wave * R + ((((lg & 1) << 1) | (lg >> 1)) * 16) + lr) * 4));
        float* dw1 = reinterpret_cast<float*>(smem + a6_opaque(C::OFF_RED + (wave * R + lr) * 4));
        auto red4 = [&](int fq, float xa, float xb, float xc, float xd) __attribute__((always_inline)) {
          asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(xa), "+v"(xb));
          asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(xc), "+v"(xd));
          float t1 = xa + xb, t2 = xc + xd;
          asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(t1), "+v"(t2));
          dw4[fq * 16] = t1 + t2;
        };
        float xs[4] = {0.f, 0.f, 0.f, 0.f};
        auto done = [&](int f, float x) __attribute__((always_inline)) {
          if (f < (KF & ~3)) {
            xs[f & 3] = x;
            if ((f & 3) == 3) red4(f - 3, xs[0], xs[1], xs[2], xs[3]);
          } else {
            x = a6_xsum4(x);
            if (lg == 0) dw1[f * 16] = x;
          }
        };
        Ops cur[2] = {rd(0), rd(1)};
#pragma unroll
        for (int f = 0; f < KF; f += 2) {
          Ops nx[2] = {cur[0], cur[1]};
          if (f + 2 < KF) nx[0] = rd(f + 2);
          if (f + 3 < KF) nx[1] = rd(f + 3);
          float part[2];
          if (f + 1 < KF) {
            goT(f, cur, part, std::integral_constant<int, 2>{});
            done(f, part[0]);
            done(f + 1, part[1]);
          } else {
            goT(f, cur, part, std::integral_constant<int, 1>{});
            done(f, part[0]);
          }
          cur[0] = nx[0]; cur[1] = nx[1];
        }
      }
      A6_STAMP(1);
      a6_barrier();   // B2
      A6_STAMP(2);
      {   // fixed-order sum of the NW partials of every query row (two lanes per row): deterministic
        const int tq = a6_opaque(tid), row = tq >> 1, part = tq & 1;
        if (row < R) {
          float t = 0.f;
#pragma unroll
          for (int w = 0; w < NW / 2; ++w) t += red[(2 * w + part) * R + row];
          t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
          if (part == 0) del_s[row] = -t;   // minus delta: phase 1b starts its dP accumulators from it
        }
      }
      a6_barrier();   // B3: delta complete; the partials are dead, the dS^T tiles may be written
      A6_STAMP(3);
      // The next pair's tiles: requested HERE - behind the last first-use of this pair's K / V registers (phase 1a), so
      // that no wait of phase 1a covers them (hipcc's vmcnt bookkeeping across the loop is coarse: requested at B1 they
      // cost every phase 1a an HBM round trip) - and in flight during phase 1b; first used behind B4.
      if (nxt < npairs) load_tiles(nxt);

      // ---- phase 1b: dV^T, dK^T of the wave's key fragments; dS^T of every (query fragment, key fragment) -> LDS -----
      f32x4 dk[NF][4], dv[NF][4], csacc[NF];
#pragma unroll
      for (int x = 0; x < NF; ++x) {
        csacc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          dk[x][d] = f32x4{0.f, 0.f, 0.f, 0.f};
          dv[x][d] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      {
        const int ln = a6_lane(lane), lr = ln & 15, lg = ln >> 4;
        const int sw = t64_swz(lr);
        const char* ga0 = smem + a6_opaque(C::OFF_G + lr * 128 + ((lg ^ sw) << 4));         // dO row lr, chunk lg
        const char* ga1 = smem + a6_opaque(C::OFF_G + lr * 128 + (((4 + lg) ^ sw) << 4));   // chunk 4 + lg
        const char* da = smem + a6_opaque(C::OFF_DEL + lg * 16);                            // -delta of rows 4 lg .. + 3
        const char* ptr_ = smem + a6_opaque(C::OFF_DS + (lr & 3) * PL + (f0 * 16 + 4 * lg + (lr >> 2)) * 8);   // transposed read of the P block (+ 128: f0 + 1)
        char* pwr = smem + a6_opaque(C::OFF_DS + lg * PL + (f0 * 16 + lr) * 8);             // dS^T block (over the P block)
        const int trow = 4 * lg + (lr >> 2), tsw = t64_swz(trow);
        const char* tq[4];
#pragma unroll
        for (int d = 0; d < 4; ++d)
          tq[d] = smem + a6_opaque(trow * 128 + (((d * 2 + ((lr >> 1) & 1)) ^ tsw) << 4) + ((lr & 1) << 3));
        auto trd = [&](const char* p) __attribute__((always_inline)) {
          return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
        };
        auto trd2 = [&](const char* p) __attribute__((always_inline)) {   // rows 4 lg .. of two consecutive fragments
          const s16x4 a = trd(p), b = trd(p + 2048);
          return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        // P and dS of query fragment f against the wave's key fragments: the dO row operands and -delta are read ONCE
        auto pds = [&](int f, s16x4 (&pw)[NF], f32x4 (&ds)[NF]) __attribute__((always_inline)) {
          const bf16x8 g0 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ga0 + f * 2048));
          const bf16x8 g1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(ga1 + f * 2048));
          const float4 d4 = *reinterpret_cast<const float4*>(da + f * 64);
#pragma unroll
          for (int x = 0; x < NF; ++x) {
            pw[x] = trd(ptr_ + f * TS + x * 128);
            f32x4 dp = f32x4{d4.x, d4.y, d4.z, d4.w};   // -delta of the fragment's rows
            dp = mfma16(g0, v0[x], dp);                 // dP[q][key] - delta[q]
            dp = mfma16(g1, v1[x], dp);
            const uint2 w = __builtin_bit_cast(uint2, pw[x]);
            const f32x2 s01 = f32x2{bflo(w.x), bfhi(w.x)} * f32x2{dp[0], dp[1]};
            const f32x2 s23 = f32x2{bflo(w.y), bfhi(w.y)} * f32x2{dp[2], dp[3]};
            ds[x] = f32x4{s01[0], s01[1], s23[0], s23[1]};
            *reinterpret_cast<s16x4*>(pwr + f * TS + x * 128) = pack4(ds[x]);   // over the P block just read (in-order LDS)
          }
        };
#pragma unroll
        for (int ip = 0; ip < KF / 2; ++ip) {
          s16x4 pa[NF], pb[NF];
          f32x4 dsa[NF], dsb[NF];
          pds(2 * ip, pa, dsa);
          pds(2 * ip + 1, pb, dsb);
          bf16x8 pf[NF], dsf[NF];
#pragma unroll
          for (int x = 0; x < NF; ++x) {
            pf[x] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(pa[x], pb[x], 0, 1, 2, 3, 4, 5, 6, 7));
            dsf[x] = pack8(dsa[x], dsb[x]);
            if constexpr (BM == 2) {
              const s16x8 ones = s16x8{0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};   // bf16 1.0
              csacc[x] = mfma16(__builtin_bit_cast(bf16x8, ones), dsf[x], csacc[x]);
            }
          }
#pragma unroll
          for (int d = 0; d < 4; ++d) {   // the transposed dO / Q operands feed both key fragments
            const bf16x8 gt = trd2(tq[d] + C::OFF_G + (2 * ip) * 2048);
            const bf16x8 qt = trd2(tq[d] + (2 * ip) * 2048);
#pragma unroll
            for (int x = 0; x < NF; ++x) {
              dv[x][d] = mfma16(gt, pf[x], dv[x][d]);    // D[d = 4 lg + r][key = lr]
              dk[x][d] = mfma16(qt, dsf[x], dk[x][d]);
            }
          }
        }
        if constexpr (KF & 1) {
          s16x4 pa[NF];
          f32x4 dsa[NF];
          pds(KF - 1, pa, dsa);
          s16x4 dsf[NF];
#pragma unroll
          for (int x = 0; x < NF; ++x) {
            dsf[x] = pack4(dsa[x]);
            if constexpr (BM == 2) csacc[x] = mfma16k16(s16x4{0x3f80, 0x3f80, 0x3f80, 0x3f80}, dsf[x], csacc[x]);
          }
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const s16x4 gt = trd(tq[d] + C::OFF_G + (KF - 1) * 2048), qt = trd(tq[d] + (KF - 1) * 2048);
#pragma unroll
            for (int x = 0; x < NF; ++x) {
              dv[x][d] = mfma16k16(gt, pa[x], dv[x][d]);
              dk[x][d] = mfma16k16(qt, dsf[x], dk[x][d]);
            }
          }
        }
        if constexpr (BM == 2) {
          // cs of key lr -> the padded query column R - 1 of the dS^T image (element 3 of plane 3 of the last tile),
          // rounded to bf16 like every dS element of this image (attention5.hip)
#pragma unroll
          for (int x = 0; x < NF; ++x) {
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(csacc[x]));
            const float cs = csacc[x][0];
            if (lg == 0)
              *reinterpret_cast<bf16*>(dSt + (KF - 1) * TS + 3 * PL + ((f0 + x) * 16 + lr) * 8 + 6) = (bf16)cs;
          }
        }
      }
#pragma unroll
      for (int x = 0; x < NF; ++x) {
        a6_drain(dk[x][0], dk[x][1], dk[x][2], dk[x][3]);
        a6_drain(dv[x][0], dv[x][1], dv[x][2], dv[x][3]);
      }
      A6_STAMP(4);
      a6_barrier();   // B4: every dS^T tile complete; nobody reads the Q / dO tiles any more
      A6_STAMP(5);

      A6_STAMP(6);
      // ---- K fragments -> the Q tile (T64 image, the A operand of phase 2) -------------------------------------------
      {
        const int ln = a6_lane(lane), lr = ln & 15, lg = ln >> 4;
#pragma unroll
        for (int x = 0; x < NF; ++x) {
          const int row = (f0 + x) * 16 + lr;
          *reinterpret_cast<uint4*>(Qt + row * 128 + ((lg ^ t64_swz(row)) << 4)) = __builtin_bit_cast(uint4, k0[x]);
          *reinterpret_cast<uint4*>(Qt + row * 128 + (((4 + lg) ^ t64_swz(row)) << 4)) = __builtin_bit_cast(uint4, k1[x]);
        }
      }
      A6_STAMP(7);
      a6_barrier();   // B5: K tile complete
      // Under phase 2: the next pair's dO tile and lse (nobody reads them after B4) and the column sums of that dO - no
      // younger load is in flight, so the waits for the prefetch registers are exact - THEN the next pair's K / V rows
      // (both dead since the K write; requested in front of the put, the put's last wait covered them: +2 k cycles).
      if (nxt < npairs) {
        A6_PUT_TILE(Gt, pg);
        put_lse();
        if constexpr (BM == 2) put_cso();
        load_k(nxt);
        load_v(nxt);
      }
      A6_STAMP(8);
      // ---- phase 2: dQ^T[d][q] of the wave's query fragments = sum_key K^T[d][key] dS^T[key][q] -----------------------
      {
        const int ln = a6_lane(lane), lr = ln & 15, lg = ln >> 4;
        const char* sa = smem + a6_opaque(C::OFF_DS + f0 * TS + (lr & 3) * PL + (4 * lg + (lr >> 2)) * 8);   // tile f0 (+ TS: f0 + 1)
        const int trow = 4 * lg + (lr >> 2), tsw = t64_swz(trow);
        const char* tk[4];
#pragma unroll
        for (int d = 0; d < 4; ++d)
          tk[d] = smem + a6_opaque(trow * 128 + (((d * 2 + ((lr >> 1) & 1)) ^ tsw) << 4) + ((lr & 1) << 3));
        auto trd = [&](const char* p) __attribute__((always_inline)) {
          return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
        };
        auto trd2 = [&](const char* p, int step) __attribute__((always_inline)) {
          const s16x4 a = trd(p), b = trd(p + step);
          return __builtin_bit_cast(bf16x8, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        f32x4 dq[NF][4];
#pragma unroll
        for (int x = 0; x < NF; ++x)
#pragma unroll
          for (int d = 0; d < 4; ++d) dq[x][d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int fp = 0; fp < KF / 2; ++fp) {
          bf16x8 dsf[NF];
#pragma unroll
          for (int x = 0; x < NF; ++x) dsf[x] = trd2(sa + x * TS + fp * 256, 128);
#pragma unroll
          for (int d = 0; d < 4; ++d) {   // the transposed K operand feeds both query fragments
            const bf16x8 kt = trd2(tk[d] + fp * 4096, 2048);
#pragma unroll
            for (int x = 0; x < NF; ++x) dq[x][d] = mfma16(kt, dsf[x], dq[x][d]);
          }
        }
        if constexpr (KF & 1) {
          s16x4 dsf[NF];
#pragma unroll
          for (int x = 0; x < NF; ++x) dsf[x] = trd(sa + x * TS + (KF - 1) * 128);
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const s16x4 kt = trd(tk[d] + (KF - 1) * 2048);
#pragma unroll
            for (int x = 0; x < NF; ++x) dq[x][d] = mfma16k16(kt, dsf[x], dq[x][d]);
          }
        }
#pragma unroll
        for (int x = 0; x < NF; ++x) a6_drain(dq[x][0], dq[x][1], dq[x][2], dq[x][3]);
        if constexpr (BM == 2) {
          // column R - 1 of dQ^T (lane lr = 15 of the owner of the last query fragment) = the q-bias gradient / scale
          if (f0 + NF - 1 == KF - 1 && lr == 15) {
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
              for (int r = 0; r < 4; ++r) red[d * 16 + lg * 4 + r] = dq[NF - 1][d][r] * scale;
          }
        }
        A6_STAMP(9);
        // ---- the wave's dQ rows, then its dK / dV rows, as whole rows through the (now dead) dS^T tile of the fragment
        char* outb = reinterpret_cast<char*>(dqkv + (long)i * L * ld + h * DH);   // wave-uniform
#pragma unroll
        for (int x = 0; x < NF; ++x) {
          char* T = dSt + (f0 + x) * TS;
          A6Rows rw;
#pragma unroll
          for (int d = 0; d < 4; ++d)
            rw.w[d] = T + a6_opaque(lr * 128 + (((d * 2 + (lg >> 1)) ^ (lr & 7)) << 4) + (lg & 1) * 8);
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int row = it * 8 + (ln >> 3), ch = ln & 7;
            rw.r[it] = T + a6_opaque(row * 128 + ((ch ^ (row & 7)) << 4));
            rw.ok[it] = (f0 + x) * 16 + row < L;
            rw.g[it] = (uint32_t)a6_opaque(((f0 + x) * 16 + row) * (int)(ld * 2) + ch * 16);
          }
          a6_store_rows(rw, dq[x], scale, outb);
          a6_store_rows(rw, dk[x], scale, outb + (long)H * DH * 2);
          a6_store_rows(rw, dv[x], 1.0f, outb + 4L * H * DH);
        }
      }
      A6_STAMP(10);
      a6_barrier();   // B6: the K tile, the dS^T tiles and `red` are free / complete
      A6_STAMP(11);
      if (nxt < npairs) A6_PUT_TILE(Qt, pq);
      if constexpr (BM == 2) {
        // per-(sample, head) column sums -> dbias[i][which][h][:]: q = column R - 1 of dQ^T, k = 0 (shift invariance of
        // the softmax), v = column sums of dO
        if (wave < 3) {
          const int which = wave, d = a6_lane(lane);
          const float t = which == 0 ? red[d] : which == 2 ? cst[d] : 0.f;
          dbias[((long)i * 3 * H + (long)which * H + h) * DH + d] = t;
        }
      }
      A6_STAMP(12);
    }
  };
#ifndef A6_NO_SETPRIO
  // the second-dispatched half (waves 4-7) loses every VALU / MFMA arbitration against its older SIMD partner: wave 4
  // (two fragments, beside wave 0's two) was the straggler of every phase (12.7 k against 9.4 k cycles in phase 1b);
  // static priority for that half equalises the pair (MI355X_MICROARCH.md, "Static priority for the younger half")
  if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
#endif
  if (wave < C::NTWO) run(std::integral_constant<int, 2>{});
  else run(std::integral_constant<int, 1>{});
#undef A6_PUT_TILE
}

}  // namespace

// Entry used by bv_attn3_bwd (attention3.hip) under BV_OPT_ATTN_CFG bit 512; -100 = shape not covered here.
int bv_attn6_bwd(const void* qkv, const void* d_o, const float* lse, void* dqkv, float* dbias, int n, int L, int H,
                 void* stream) {
  if (L <= 192 || L > 208) return -100;
  if (dbias && !(L & 15)) return -100;   // the identities need a padded query column
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  const int npairs = n * H;
  const int grid = npairs < cus ? npairs : cus;
  hipStream_t s = (hipStream_t)stream;
  auto go = [&](auto kern) __attribute__((always_inline)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, A6::LDS);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(A6::NT), A6::LDS, s, (const bf16*)qkv, (const bf16*)d_o, lse, (bf16*)dqkv,
                       dbias, L, H, npairs, 0.125f);
  };
  if (!dbias) go(attn6_bwd_kernel<0>);
  else go(attn6_bwd_kernel<2>);
  return bv_check_launch("bv_attn_bwd(one launch, 8 waves)");
}
