// 256x128x32 bf16 MFMA GEMM for gfx950 (MI355X), k-major operands, TWO co-resident workgroups per CU.
// Round-5 experiment (VERDICT r4 "next" #2), a measured NEGATIVE: 0.67-0.85x of gemm256 on every step shape and epilogue
// (profiles/r05_gemm_pair_ab.txt), because the kernel is bound by the CU's global->LDS fill path, which a half-size tile
// loads 1.5x harder (profiles/r05_gemm_pair_probe.txt, profiles/NOTES_r05.md).  It ran inside libbvhip behind a context
// option for its parity run - every epilogue bit-identical to gemm256 (profiles/r05_pair_kernel_parity.txt) - and left
// the library afterwards (the per-epilogue A/B and the parity run are reproducible at commit 1890027, where the kernel sat
// behind BV_OPT_GEMM_PAIR with tools/gemm_pair_ab.py): this file is included by tools/probes/gemm_pair_probe.hip only.
// Reference call sites as gemm256.hip (big_vision/models/vit.py:72,77,93-98 and their backward transposes).
//
// Why.  gemm256.hip fills a CU with ONE 8-wave workgroup (160 KiB of LDS, 256x256 tile): at every tile boundary all
// eight waves convert and store their accumulators and the CU's matrix pipes idle - for the fused epilogues of the MLP
// (two [M, N] output streams, an auxiliary [M, N] operand) that phase is as long as the K loop at K = 768
// (profiles/NOTES_r04.md: ~20 us of MFMAs + 18-24 us of epilogue traffic per tile, back to back).  Every way of
// overlapping the two INSIDE that workgroup lost (rolling epilogue, second accumulator set, stores inside the MFMA
// segments: profiles/NOTES_r01-r03.md).  Here the overlap is left to the hardware: two INDEPENDENT 4-wave workgroups
// per CU (80 KiB of LDS each, one wave of each per SIMD, 256 registers per wave), each with the same 128x64
// accumulator block per wave as gemm256 - while one workgroup streams its epilogue, the other owns the matrix pipes.
// No barrier couples the two, so their phases drift apart by themselves.
//
//   tile      256 (M) x 128 (N), waves 2 (M) x 2 (N), acc[8][4] of v_mfma_f32_16x16x32_bf16 per wave
//   K-tile    32 (one MFMA k-step): stage = A 256 rows x 64 B + B 128 rows x 64 B = 24 KiB, 3 stages = 72 KiB
//   feed      global_load_lds_dwordx4, 24 pieces of 1 KiB per stage = 6 per wave, issued two K-tiles ahead; the
//             stream runs across tile boundaries (persistent workgroups, 512 = 2 per CU)
//   sync      ONE workgroup barrier per K-tile: behind it every wave's pieces of the next K-tile have landed
//             (counted s_waitcnt vmcnt(6)) and everybody is done with the stage the next request overwrites
//   LDS image [row][4 x 16 B]; 16-byte chunk c of row r sits at position c ^ (((r >> 3) & 1) << 1): conflict-free for
//             ds_read_b128 under the lane groups of MI355X_MICROARCH.md (LDS table) for both fragment-row maps
//             (rows i*16 + lr, and the bf16-output map (j>>1)*32 + (lr>>2)*8 + (j&1)*4 + (lr&3) that gives a lane 8
//             consecutive output columns per fragment pair); the DMA writes lane-linearly, so the swizzle is applied
//             to the per-lane global source address
//   results   bit-identical to gemm256.hip: same fragment maps, same k order per accumulator, same epilogue arithmetic
//             (profiles/r05_pair_kernel_parity.txt)
#include "../../big_vision_amd/csrc/bv_common.h"
#include "../../big_vision_amd/csrc/bvhip_internal.h"

namespace {

constexpr int P_STAGE_A = 256 * 64;               // bytes
constexpr int P_STAGE_B = 128 * 64;
constexpr int P_STAGE = P_STAGE_A + P_STAGE_B;    // 24 KiB
constexpr int P_NSTAGE = 3;
constexpr int P_SMEM = P_NSTAGE * P_STAGE;        // 72 KiB: two workgroups per CU

struct PairParams {
  const bf16* A;
  const bf16* B;
  void* C;
  void* C2;
  const float* bias;
  const void* aux;
  float* colsum;
  long lda, ldb, ldc, ldaux;
  int M, N, K;
  int aux_rows;
  int tiles_n;   // N / 128
  int ntiles;
  float alpha;
  int nt;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gl_void;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int OFF>
__device__ __forceinline__ bf16x8 p_lds_read128(uint32_t addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ void p_glds16(const bf16* src, char* dst_wave_base) {
  __builtin_amdgcn_global_load_lds((gl_void*)src, (lds_void*)dst_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void p_st16(void* ptr, u32x4 v, bool nt) {
  if (nt) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(ptr));
  else *reinterpret_cast<u32x4*>(ptr) = v;
}
__device__ __forceinline__ u32x4 p_ld16(const void* ptr, bool nt) {
  if (nt) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ptr));
  return *reinterpret_cast<const u32x4*>(ptr);
}

// The k-major epilogue of gemm256_kernel for one wave's 128 x 64 block at (m0w, n0w): the same arithmetic, statement
// by statement (bias, alpha, +residual / +posemb, GELU with two outputs, GELU' / MUL with fused column sums), so that
// both kernels produce the same bits.  A lane holds, per row fragment i (row m0w + i*16 + lr) and B fragment j, the 4
// columns nc[j] .. nc[j]+3 (fp32 output: j*16 + lg*4; bf16 output: (j>>1)*32 + lg*8 + (j&1)*4).
template <int EPI, bool OUTF32>
__device__ __forceinline__ void pair_epilogue(const PairParams& p, f32x4 (&acc)[8][4], int m0w, int n0w, int lr, int lg) {
  int nc[4], ncl[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ncl[j] = OUTF32 ? j * 16 + lg * 4 : (j >> 1) * 32 + lg * 8 + (j & 1) * 4;
    nc[j] = n0w + ncl[j];
  }
  constexpr int ESZ = OUTF32 ? 4 : 2;
  const long crow = (long)m0w * p.ldc + n0w;
  char* const Cb = reinterpret_cast<char*>(p.C) + crow * ESZ;
  char* const C2b = reinterpret_cast<char*>(p.C2) + crow * 2;
  const long cstep = 16L * p.ldc;
  uint32_t vc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) vc[j] = (uint32_t)(lr * (int)p.ldc + ncl[j]) * ESZ;
  f32x2 bv[8], cs[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) bv[q] = cs[q] = pk_splat(0.f);
  if (p.bias) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 b = *reinterpret_cast<const float4*>(p.bias + nc[j]);
      bv[j * 2 + 0] = f32x2{b.x, b.y}; bv[j * 2 + 1] = f32x2{b.z, b.w};
    }
  }
  const f32x2 alpha2 = pk_splat(p.alpha);
  const int mrow0 = m0w + lr;
  const bool nts = (p.nt & 1), ntl = p.nt & 2;
  constexpr bool GBWD = EPI == BV_EPI_GELU_BWD || EPI == BV_EPI_GELU_BWD_EMIT || EPI == BV_EPI_MUL;
  constexpr bool RESBF = EPI == BV_EPI_RESIDUAL && !OUTF32;
  constexpr int IB = GBWD ? 2 : 4;
#pragma unroll
  for (int ib = 0; ib < 8; ib += IB) {
    float4 ax[IB][4];
    uint4 hx[IB][2];
    if constexpr (EPI == BV_EPI_POS) {
#pragma unroll
      for (int ii = 0; ii < IB; ++ii) {
        const int m = mrow0 + (ib + ii) * 16;
        const float* x = reinterpret_cast<const float*>(p.aux) + (long)(m % p.aux_rows) * p.ldaux;
#pragma unroll
        for (int j = 0; j < 4; ++j) ax[ii][j] = __builtin_bit_cast(float4, p_ld16(x + nc[j], ntl));
      }
    } else if constexpr (EPI == BV_EPI_RESIDUAL && OUTF32) {
      const char* Ab = reinterpret_cast<const char*>(p.aux) + ((long)m0w * p.ldaux + n0w) * 4;
#pragma unroll
      for (int ii = 0; ii < IB; ++ii) {
        const char* x = Ab + (long)(ib + ii) * 16 * p.ldaux * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          ax[ii][j] = __builtin_bit_cast(float4, p_ld16(x + (uint32_t)(lr * (int)p.ldaux + ncl[j]) * 4u, ntl));
      }
    } else if constexpr (GBWD || RESBF) {
      const char* Ab = reinterpret_cast<const char*>(p.aux) + ((long)m0w * p.ldaux + n0w) * 2;
      const uint32_t va0 = (uint32_t)(lr * (int)p.ldaux + ncl[0]) * 2u, va1 = (uint32_t)(lr * (int)p.ldaux + ncl[2]) * 2u;
#pragma unroll
      for (int ii = 0; ii < IB; ++ii) {
        const char* h = Ab + (long)(ib + ii) * 16 * p.ldaux * 2;
        hx[ii][0] = __builtin_bit_cast(uint4, p_ld16(h + va0, ntl));
        hx[ii][1] = __builtin_bit_cast(uint4, p_ld16(h + va1, ntl));
      }
    }
#pragma unroll
    for (int ii = 0; ii < IB; ++ii) {
      const int i = ib + ii;
      if constexpr (GBWD) __builtin_amdgcn_sched_barrier(0);
      f32x2 v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
          v[j * 2 + h2] = __builtin_elementwise_fma(f32x2{acc[i][j][h2 * 2], acc[i][j][h2 * 2 + 1]}, alpha2, bv[j * 2 + h2]);
      if constexpr (OUTF32) {
        if constexpr (EPI == BV_EPI_RESIDUAL || EPI == BV_EPI_POS) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j * 2 + 0] += f32x2{ax[ii][j].x, ax[ii][j].y};
            v[j * 2 + 1] += f32x2{ax[ii][j].z, ax[ii][j].w};
          }
        }
        char* c = Cb + (long)i * cstep * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          p_st16(c + vc[j], __builtin_bit_cast(u32x4, make_float4(v[j * 2].x, v[j * 2].y, v[j * 2 + 1].x, v[j * 2 + 1].y)), nts);
      } else {
        char* c = Cb + (long)i * cstep * 2;
        char* c2 = C2b + (long)i * cstep * 2;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          uint32_t aw[4] = {0, 0, 0, 0};
          if constexpr (GBWD || RESBF) { aw[0] = hx[ii][hh].x; aw[1] = hx[ii][hh].y; aw[2] = hx[ii][hh].z; aw[3] = hx[ii][hh].w; }
          uint32_t cw[4], gw[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f32x2& x = v[hh * 4 + e];
            if constexpr (RESBF) {
              x += bf2_unpack(aw[e]);
            } else if constexpr (EPI == BV_EPI_MUL) {
              x *= bf2_unpack(aw[e]);
            } else if constexpr (EPI == BV_EPI_GELU_BWD || EPI == BV_EPI_GELU_BWD_EMIT) {
              uint32_t dw;
              mlp_act_from_h(aw[e], gw[e], dw);
              x *= bf2_unpack(dw);
            }
            if constexpr (GBWD) {
              cs[hh * 4 + e] += x;
              asm volatile("" : "+v"(cs[hh * 4 + e]));
            }
            if constexpr (EPI == BV_EPI_GELU_GD) {
              uint32_t hw;
              mlp_act_words(x, hw, cw[e], gw[e]);
            } else {
              cw[e] = bf2_pack(x);
              if constexpr (EPI == BV_EPI_GELU) gw[e] = bf2_pack(gelu_tanh_pk(bf2_unpack(cw[e])));
            }
          }
          p_st16(c + vc[hh * 2], u32x4{cw[0], cw[1], cw[2], cw[3]}, nts);
          if constexpr (EPI == BV_EPI_GELU || EPI == BV_EPI_GELU_GD || EPI == BV_EPI_GELU_BWD_EMIT)
            p_st16(c2 + vc[hh * 2], u32x4{gw[0], gw[1], gw[2], gw[3]}, nts);
        }
      }
    }
  }
  if constexpr (GBWD) {
    if (p.colsum) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float sum = cs[e >> 1][e & 1];
        sum += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sum), 0xB1, 0xF, 0xF, true));
        sum += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sum), 0x4E, 0xF, 0xF, true));
        sum += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sum), 0x141, 0xF, 0xF, true));
        sum += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sum), 0x140, 0xF, 0xF, true));
        if (lr == 0) unsafeAtomicAdd(p.colsum + nc[e >> 2] + (e & 3), sum);
      }
    }
  }
}

struct PCursor {   // position of the load stream: item j of this workgroup, K-tile t
  int j, t;
  long offA, offB;   // element offsets of the item's first rows
};

// The wait for the second half of a K-tile's fragment reads names the 16 accumulators of the first half: hipcc moves an
// untied asm wait up across MFMAs (they touch no memory), which would make the whole K-tile wait for all 12 reads.
#define P_WAIT_SECOND_HALF()                                                                                          \
  do {                                                                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                               \
                 : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]), \
                   "+v"(acc[1][2]), "+v"(acc[1][3]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[2][2]), "+v"(acc[2][3]), \
                   "+v"(acc[3][0]), "+v"(acc[3][1]), "+v"(acc[3][2]), "+v"(acc[3][3])                                  \
                 :                                                                                                    \
                 : "memory");                                                                                         \
    asm volatile("" : "+v"(af[4]), "+v"(af[5]), "+v"(af[6]), "+v"(af[7]));                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
  } while (0)

// PROBE != 0 exists only for tools/probes/gemm_pair_probe.hip (bottleneck ablation, results are garbage): bit 0 = no
// global->LDS requests after the prologue, bit 1 = fragment reads only for the first K-tile, bit 2 = no epilogue.
// The library instantiates PROBE = 0 only.
template <int EPI, bool OUTF32, int PROBE = 0>
__global__ __launch_bounds__(256, 2) void gemm_pair_kernel(PairParams p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 15, lg = lane >> 4;

  // ---- XCD-aware work distribution (as gemm256_kernel: block b runs on XCD b % 8, every XCD takes a contiguous
  // range of tile ids, column tile fastest, its workgroups walk it round after round)
  const int bid = blockIdx.x, G = gridDim.x;
  const int nwork = p.ntiles;
  const int xcd = bid & 7, idx = bid >> 3;
  const int q8 = nwork >> 3, r8 = nwork & 7;
  const int cs0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int cl = q8 + (xcd < r8 ? 1 : 0);
  const int bpx = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
  const int nmy = idx < cl ? (cl - idx + bpx - 1) / bpx : 0;
  if (nmy == 0) return;
  const int nk = p.K >> 5;

  auto item_mn = [&](int j, int& m0, int& n0) {
    const int tile = cs0 + idx + j * bpx;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    m0 = tm * 256; n0 = tn * 128;
  };
  PCursor ld{};
  auto load_item = [&](PCursor& c) {
    int m0, n0;
    item_mn(c.j, m0, n0);
    c.offA = (long)m0 * p.lda;
    c.offB = (long)n0 * p.ldb;
  };
  auto advance = [&](PCursor& c) {
    if (++c.t == nk) {
      c.t = 0;
      ++c.j;
      if (c.j < nmy) load_item(c);
    }
  };

  // ---- per-lane global sources of the DMA.  A piece = 16 rows x 64 B, written lane-linearly: lane -> row lane >> 2,
  // position lane & 3, which must receive the 16-byte k-chunk pos ^ swz(row), swz(row) = ((row >> 3) & 1) << 1 =
  // ((lane >> 5) & 1) << 1 for every piece (pieces start at multiples of 16 rows).  Wave w issues A pieces w, w+4, w+8,
  // w+12 (rows 16 w + 64 q) and B pieces w, w+4.
  const int chunk = (lane & 3) ^ (((lane >> 5) & 1) << 1);
  const bf16* const srcA = p.A + (long)(wave * 16 + (lane >> 2)) * p.lda + chunk * 8;
  const bf16* const srcB = p.B + (long)(wave * 16 + (lane >> 2)) * p.ldb + chunk * 8;
  const long qA = 64 * p.lda, qB = 64 * p.ldb;
  bool probe_quiet = false;   // PROBE: past the prologue / the first K-tile
  auto issue = [&](const PCursor& c, int stage) {
    if ((PROBE & 1) && probe_quiet) return;
    char* d = smem + stage * P_STAGE + wave * 1024;
    const bf16* a = srcA + c.offA + c.t * 32;
    p_glds16(a, d);
    p_glds16(a + qA, d + 4096);
    p_glds16(a + 2 * qA, d + 8192);
    p_glds16(a + 3 * qA, d + 12288);
    const bf16* b = srcB + c.offB + c.t * 32;
    p_glds16(b, d + P_STAGE_A);
    p_glds16(b + qB, d + P_STAGE_A + 4096);
  };

  // ---- per-lane LDS read addresses (stage 0)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t ra = lds0 + (wr * 128 + lr) * 64 + ((lg ^ (((lr >> 3) & 1) << 1)) << 4);
  const int brow = OUTF32 ? lr : (lr >> 2) * 8 + (lr & 3);
  const uint32_t rb = lds0 + P_STAGE_A + (wc * 64 + brow) * 64 + ((lg ^ (((brow >> 3) & 1) << 1)) << 4);
  constexpr int O1 = OUTF32 ? 1024 : 256, O2 = 2048, O3 = OUTF32 ? 3072 : 2304;   // B fragments 1, 2, 3

  f32x4 acc[8][4];
  bf16x8 af[8], bfg[4];

  // ---- prologue: K-tiles 0 and 1 of the stream in flight, K-tile 0 landed
  ld.j = 0; ld.t = 0;
  load_item(ld);
  issue(ld, 0);
  advance(ld);
  int gk = 0;   // K-tiles consumed by this workgroup so far (ring position)
  if (ld.j < nmy) {
    issue(ld, 1);
    advance(ld);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

  for (int jt = 0; jt < nmy; ++jt) {
    for (int t = 0; t < nk; ++t) {
      const int st = gk % 3;
      const uint32_t so = st * P_STAGE;
      const uint32_t a0 = ra + so, b0 = rb + so;
      // fragment reads of this K-tile (12 x 1 KiB), then the request for the K-tile two ahead into the stage the
      // previous K-tile used (everybody left it before the barrier that ended that iteration)
      if (!((PROBE & 2) && probe_quiet)) {
        bfg[0] = p_lds_read128<0>(b0);  bfg[1] = p_lds_read128<O1>(b0);
        bfg[2] = p_lds_read128<O2>(b0); bfg[3] = p_lds_read128<O3>(b0);
        af[0] = p_lds_read128<0>(a0);    af[1] = p_lds_read128<1024>(a0);
        af[2] = p_lds_read128<2048>(a0); af[3] = p_lds_read128<3072>(a0);
        af[4] = p_lds_read128<4096>(a0); af[5] = p_lds_read128<5120>(a0);
        af[6] = p_lds_read128<6144>(a0); af[7] = p_lds_read128<7168>(a0);
      }
      const bool more = ld.j < nmy;
      if (more) issue(ld, (gk + 2) % 3);
      // (waits are untied; the empty statements behind them name the registers the reads fill, so that no consumer can
      // be placed above its wait - hipcc treats the asm reads' outputs as ready at once)
      asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      asm volatile("" : "+v"(bfg[0]), "+v"(bfg[1]), "+v"(bfg[2]), "+v"(bfg[3]), "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]));
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      if (t == 0) {   // first K-tile of a work item: C = 0 as an inline constant, no accumulator clearing
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfg[j], af[i], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        P_WAIT_SECOND_HALF();
#pragma unroll
        for (int i = 4; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfg[j], af[i], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfg[j], af[i], acc[i][j], 0, 0, 0);
        P_WAIT_SECOND_HALF();
#pragma unroll
        for (int i = 4; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfg[j], af[i], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      // the next K-tile's pieces (requested one iteration ago) have landed once at most this iteration's six are
      // still in flight; the queue is in order, so this also retires an epilogue's stores issued before them
      if (more) {
        advance(ld);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      ++gk;
      probe_quiet = true;
    }
    int m0, n0;
    item_mn(jt, m0, n0);
    if constexpr (PROBE & 4) {   // keep the math live, skip conversion and stores
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    } else {
      pair_epilogue<EPI, OUTF32>(p, acc, m0 + wr * 128, n0 + wc * 64, lr, lg);
    }
  }
}

template <int EPI, bool OUTF32>
void launch_pair(const PairParams& p, int grid, hipStream_t s) {
  auto kern = gemm_pair_kernel<EPI, OUTF32>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, P_SMEM);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), P_SMEM, s, p);
}

#undef P_WAIT_SECOND_HALF

}  // namespace

// Dispatcher (was called by bv_gemm_bf16_colsum in front of the 256x256 path during the parity run).  Returns 1 if the
// problem was launched here: both operands k-major, M a multiple of 256, N of 128, K of 32 and >= 64, and the
// epilogue's bit set in `mask`.
int bv_gemm_pair_try(int a_kmajor, int b_kmajor, const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                     int out_f32, int M, int N, int K, int epilogue, const float* bias, const void* aux, long ldaux,
                     int aux_rows, void* C2, float alpha, float* colsum, void* stream, const bv_ctx* ctx_, long mask) {
  const bv_ctx* ctx = bv_ctx_or_default(ctx_);
  if (!a_kmajor || !b_kmajor || epilogue == BV_EPI_ATOMIC || !((mask >> epilogue) & 1)) return 0;
  if ((M & 255) || (N & 127) || (K & 31) || K < 64) return 0;
  if ((lda & 7) || (ldb & 7) || (ldc & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15)) return 0;
  if (aux && ((ldaux & 7) || ((uintptr_t)aux & 15))) return 0;
  if (bias && ((uintptr_t)bias & 15)) return 0;
  if (C2 && ((uintptr_t)C2 & 15)) return 0;
  PairParams p;
  p.A = (const bf16*)A; p.B = (const bf16*)B; p.C = C; p.C2 = C2;
  p.bias = bias; p.aux = aux; p.colsum = colsum;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldaux = ldaux;
  p.M = M; p.N = N; p.K = K; p.aux_rows = aux_rows > 0 ? aux_rows : 1;
  p.tiles_n = N >> 7;
  p.ntiles = (M >> 8) * p.tiles_n;
  p.alpha = alpha;
  p.nt = (int)ctx->opt[BV_OPT_GEMM_NT];
  const int cus = 256 - (int)ctx->opt[BV_OPT_GEMM_RESERVE_CUS];
  const int grid = p.ntiles < 2 * cus ? p.ntiles : 2 * cus;   // persistent: two workgroups per CU
  hipStream_t s = (hipStream_t)stream;
  if (epilogue == BV_EPI_RESIDUAL && !out_f32) launch_pair<BV_EPI_RESIDUAL, false>(p, grid, s);
  else if (epilogue == BV_EPI_RESIDUAL) launch_pair<BV_EPI_RESIDUAL, true>(p, grid, s);
  else if (epilogue == BV_EPI_POS) launch_pair<BV_EPI_POS, true>(p, grid, s);
  else if (epilogue == BV_EPI_GELU) launch_pair<BV_EPI_GELU, false>(p, grid, s);
  else if (epilogue == BV_EPI_GELU_BWD) launch_pair<BV_EPI_GELU_BWD, false>(p, grid, s);
  else if (epilogue == BV_EPI_GELU_BWD_EMIT) launch_pair<BV_EPI_GELU_BWD_EMIT, false>(p, grid, s);
  else if (epilogue == BV_EPI_GELU_GD) launch_pair<BV_EPI_GELU_GD, false>(p, grid, s);
  else if (epilogue == BV_EPI_MUL) launch_pair<BV_EPI_MUL, false>(p, grid, s);
  else if (out_f32) launch_pair<BV_EPI_NONE, true>(p, grid, s);
  else launch_pair<BV_EPI_NONE, false>(p, grid, s);
  return 1;
}
