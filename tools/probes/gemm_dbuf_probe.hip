// EXPERIMENT (not part of libbvhip.so): k-major bf16 GEMM with DOUBLE-BUFFERED accumulators and a
// trickled epilogue, the candidate fix for the tile-boundary cost of gemm256.hip (DESIGN.md 4.1:
// 9-10 k cycles of 43 k per K = 768 tile go to the epilogue's store burst, which all 32
// workgroups of an XCD push through the L2 write path at the same time).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I big_vision_amd/csrc tools/probes/gemm_dbuf_probe.hip \
//         big_vision_amd/csrc/c_api.cpp -o tools/probes/gemm_dbuf_probe.out && tools/probes/gemm_dbuf_probe.out
//
// Differences to gemm256.hip:
//   * workgroup tile 256 x 128 (8 waves as 4(M) x 2(N), wave tile 64 x 64 = acc[4][4]): half the
//     accumulator registers, so a wave holds TWO sets - the tile being computed and the finished
//     tile whose results are still being written;
//   * no epilogue phase: while tile t+1 runs its K loop, one 16-row fragment of tile t is
//     converted and stored per K-tile (2 x 16-B stores per lane), inside an MFMA section.  The
//     store traffic of a workgroup is spread over its whole next tile, the chip never sees the
//     synchronized burst, and no wave ever waits for its own stores: the counted vmcnt of every
//     K-tile leaves the youngest stores outstanding;
//   * ring of 3 stages x (A 2 half-tiles + B 1 half-tile) = 144 KiB, loads 2 K-tiles ahead;
//     a K-tile is 2 phases of 16 MFMAs (rows 0-31 / 32-63 of the wave tile x all 64 columns).
//     B is restaged in phase a and A in phase b of K-tile k (their slots were last read in
//     phase a / phase b of K-tile k-1: two phases earlier, the WAR distance of the 8-phase
//     template).
// LDS images, swizzle (kswz), DMA lane mapping, fragment row maps and the store pattern are the
// ones of gemm256.hip, which this file includes (also as the bit-exact reference: both kernels
// accumulate K in the same order).
//
// STATUS (end of round 1).  Two runs on the MI355X with the FIRST form of the trickle (a runtime
// switch over the four fragments): v1 and v2 BIT-EXACT against gemm256 on every shape below (incl.
// 100352 x 3072 x 768), but at half its speed (qkv forward 453-469 TF/s vs 947; the same without
// the stores, the same with v2's deeper A ring - so neither the stores nor the load latency).  The
// cause was found afterwards in the ISA: the optimiser merges the switch cases into one block
// with a computed fragment index, which moves a whole accumulator set into SCRATCH memory
// (ScratchSize 272 B/lane, 72 scratch_load/store in the loop - VMEM instructions in the same
// in-order queue as the DMA, each followed by a wait that drains the load pipeline).  The code
// below peels the first four K-tiles of a tile into separate instantiations (compile-time
// fragment index): ScratchSize 0, 192 VGPRs, no wait in front of the trickled stores in the ISA.
// THIS FORM HAS NOT RUN ON A GPU YET (the round's GPU budget was spent) - first thing to do next
// round: run this probe (bit-exactness check + timing are built in).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <vector>
#define BV_GEMM256_PROBES   // compiles the PROBE != 0 ablation paths of gemm256_kernel (absent from the library build)
#include "../../big_vision_amd/csrc/gemm256.hip"

namespace {

constexpr int DB_NS = 3;                    // ring stages
constexpr int DB_STAGE = 3 * HALF;          // A half 0, A half 1, B: 48 KiB
constexpr int DB_SMEM = DB_NS * DB_STAGE;   // 147456 B

struct DbParams {
  const bf16* A;     // [M][K]
  const bf16* B;     // [N][K]
  bf16* C;           // [M][N]
  long lda, ldb, ldc;
  int M, N, K;
  int tiles_n;       // N / 128
  int ntiles;        // (M / 256) * (N / 128)
  float alpha;
  long* dbg;
};

struct DbCursor {   // position of the load stream: item j of this block, K-tile t
  int j, t, nk;
  long offA, offB;
  int m0, n0;
};

template <int PROBE = 0>
__global__ __launch_bounds__(512, 2) void gemm_dbuf_kernel(DbParams p) {
  __shared__ __attribute__((aligned(1024))) char smem[DB_SMEM];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;            // which A half-tile (rows grp*128..) and which ping-pong group
  const int wsub = (wave >> 1) & 1;     // 64-row block inside the half
  const int wc = wave & 1;              // 64-column block of the 128-wide tile
  const int lr = lane & 15, lg = lane >> 4;

  // ---- XCD-aware persistent work distribution (same scheme as gemm256.hip)
  const int bid = blockIdx.x, G = gridDim.x;
  const int nwork = p.ntiles;
  const int xcd = bid & 7, idx = bid >> 3;
  const int q8 = nwork >> 3, r8 = nwork & 7;
  const int cs = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int cl = q8 + (xcd < r8 ? 1 : 0);
  const int bpx = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
  const int nmy = idx < cl ? (cl - idx + bpx - 1) / bpx : 0;
  if (nmy == 0) return;
  const int nk = p.K >> 6;
  if (PROBE != 0 && p.dbg && tid == 0) {
    p.dbg[bid * 4 + 0] = __builtin_amdgcn_s_memtime();
    p.dbg[bid * 4 + 1] = __builtin_amdgcn_s_memrealtime();
  }

  auto load_item = [&](DbCursor& c) __attribute__((always_inline)) {
    const int w = cs + idx + c.j * bpx;
    const int tm = w / p.tiles_n, tn = w - tm * p.tiles_n;
    c.m0 = tm * 256; c.n0 = tn * 128;
    c.nk = nk;
    c.offA = (long)c.m0 * p.lda;
    c.offB = (long)c.n0 * p.ldb;
  };
  auto advance = [&](DbCursor& c) __attribute__((always_inline)) {
    if (++c.t == c.nk) {
      c.t = 0;
      ++c.j;
      if (c.j < nmy) load_item(c);
    }
  };

  // ---- DMA source pointers (lane -> row / 16-B chunk of a [128 rows][64 k] half-tile)
  const int r = wave * 8 + (lane >> 3);
  const int pos = lane & 7;
  const int ch = pos ^ kswz(r);
  const bf16* srcA = p.A + (long)r * p.lda + ch * 8;
  const bf16* srcB = p.B + (long)r * p.ldb + ch * 8;
  const long gA = 64 * p.lda, gB = 64 * p.ldb, hA = 128 * p.lda;
  const int wave_off = wave * 1024;
  auto issueA = [&](const DbCursor& c, int slot) __attribute__((always_inline)) {   // 4 DMA instructions per thread
    char* d = smem + slot * DB_STAGE + wave_off;
    const bf16* s = srcA + c.offA + (long)c.t * 64;
    glds16(s, d);
    glds16(s + gA, d + 8192);
    glds16(s + hA, d + HALF);
    glds16(s + hA + gA, d + HALF + 8192);
  };
  auto issueB = [&](const DbCursor& c, int slot) __attribute__((always_inline)) {   // 2 DMA instructions per thread
    char* d = smem + slot * DB_STAGE + 2 * HALF + wave_off;
    const bf16* s = srcB + c.offB + (long)c.t * 64;
    glds16(s, d);
    glds16(s + gB, d + 8192);
  };

  // ---- fragment read addresses (relative to the stage base)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int brow = (lr >> 2) * 8 + (lr & 3);      // bf16-output column map (see gemm256.hip)
  const uint32_t ra0 = lds0 + grp * HALF + (wsub * 64 + lr) * 128 + ((lg ^ kswz(lr)) << 4);
  const uint32_t ra1 = ra0 ^ 64;
  const uint32_t rb0 = lds0 + 2 * HALF + (wc * 64 + brow) * 128 + ((lg ^ kswz(brow)) << 4);
  const uint32_t rb1 = rb0 ^ 64;

  f32x4 acc[2][4][4];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[q][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[4][2], bfg[4][2];

  // odd fragments (row bit 4 / row bit 2 set) swap the two k-step bases (kswz bit 2 flips)
  auto readA01 = [&](uint32_t sb) __attribute__((always_inline)) {
    const uint32_t a0 = ra0 + sb, a1 = ra1 + sb;
    af[0][0] = lds_read128<0>(a0);     af[0][1] = lds_read128<0>(a1);
    af[1][0] = lds_read128<2048>(a1);  af[1][1] = lds_read128<2048>(a0);
  };
  auto readA23 = [&](uint32_t sb) __attribute__((always_inline)) {
    const uint32_t a0 = ra0 + sb, a1 = ra1 + sb;
    af[2][0] = lds_read128<4096>(a0);  af[2][1] = lds_read128<4096>(a1);
    af[3][0] = lds_read128<6144>(a1);  af[3][1] = lds_read128<6144>(a0);
  };
  auto readB = [&](uint32_t sb) __attribute__((always_inline)) {
    const uint32_t b0 = rb0 + sb, b1 = rb1 + sb;
    bfg[0][0] = lds_read128<0>(b0);     bfg[0][1] = lds_read128<0>(b1);
    bfg[1][0] = lds_read128<512>(b1);   bfg[1][1] = lds_read128<512>(b0);
    bfg[2][0] = lds_read128<4096>(b0);  bfg[2][1] = lds_read128<4096>(b1);
    bfg[3][0] = lds_read128<4608>(b1);  bfg[3][1] = lds_read128<4608>(b0);
  };

#define DB_MFMA(Q, I0)                                                                        \
  do {                                                                                        \
    __builtin_amdgcn_s_setprio(1);                                                            \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                          \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                           \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                         \
          acc[Q][(I0) + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                      \
              bfg[j][ks], af[(I0) + i][ks], acc[Q][(I0) + i][j], 0, 0, 0);                    \
    __builtin_amdgcn_s_setprio(0);                                                            \
  } while (0)
#define DB_MID()                                          \
  do {                                                    \
    __builtin_amdgcn_s_barrier();                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);                    \
  } while (0)
#define DB_END()                             \
  do {                                       \
    __builtin_amdgcn_sched_barrier(0);       \
    __builtin_amdgcn_s_barrier();            \
  } while (0)

  // ---- trickled epilogue: row fragment U of accumulator set Q -> C (2 x 16-B stores per lane)
  int pm0 = 0, pn0 = 0;  // origin of the previous tile
  // per-lane part of the output address (elements); the tile / fragment part is wave-uniform
  const long lane_off = (long)(grp * 128 + wsub * 64 + lr) * p.ldc + wc * 64 + lg * 8;
  auto store_unit = [&](auto Qtag, auto Utag) __attribute__((always_inline)) {
    constexpr int Q = decltype(Qtag)::value, U = decltype(Utag)::value;
    const long tile_off = (long)(pm0 + U * 16) * p.ldc + pn0;          // SGPRs
    bf16* c = p.C + tile_off + lane_off;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const f32x4 x = acc[Q][U][2 * hh], y = acc[Q][U][2 * hh + 1];
      u32x4 o;
      o[0] = pack_bf2(x[0] * p.alpha, x[1] * p.alpha);
      o[1] = pack_bf2(x[2] * p.alpha, x[3] * p.alpha);
      o[2] = pack_bf2(y[0] * p.alpha, y[1] * p.alpha);
      o[3] = pack_bf2(y[2] * p.alpha, y[3] * p.alpha);
      if (PROBE != 5) *reinterpret_cast<u32x4*>(c + hh * 32) = o;
      else asm volatile("" ::"v"(o));
      acc[Q][U][2 * hh] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[Q][U][2 * hh + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  // The fragment index must be a COMPILE-TIME constant all the way: a runtime switch over the
  // four fragments is merged by the optimiser into one block with a computed index, which puts a
  // whole accumulator set into scratch memory (272 B/lane, scratch loads/stores in the same
  // in-order VMEM queue as the DMA: the first GPU run of this probe, at half speed).  The first
  // four K-tiles of a tile are therefore separate instantiations of the K-tile body.
  bool has_prev = false;   // a finished tile is waiting in the other accumulator set
  // ---- prologue: K-tiles 0 and 1 of the stream in flight, K-tile 0 landed
  DbCursor cur{};
  cur.j = 0; cur.t = 0;
  load_item(cur);
  DbCursor ld = cur;               // load stream, runs 2 K-tiles ahead of the math
  issueB(ld, 0); issueA(ld, 0);
  advance(ld);
  if (ld.j < nmy) {
    issueB(ld, 1); issueA(ld, 1);
    advance(ld);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier behind waves 0-3

  int slot = 0;           // ring slot of the K-tile being consumed
  int stores_prev = 0;    // trickle stores issued in the previous K-tile (after its DMA)

  // one K-tile of the tile in accumulator set P; U >= 0: also write fragment U of the previous tile
  auto ktile = [&](auto Ptag, auto Utag) __attribute__((always_inline)) {
    constexpr int P = decltype(Ptag)::value, U = decltype(Utag)::value;
    const uint32_t sb = slot * DB_STAGE;
    const int slot2 = slot == 0 ? 2 : slot - 1;      // (slot + 2) % 3: slot of K-tile k+2 = K-tile k-1's
    const bool more = ld.j < nmy;
    // -------- phase a: rows 0-31 of the wave tile
    readA01(sb);
    readB(sb);
    if (more) issueB(ld, slot2);
    DB_MID();
    DB_MFMA(P, 0);
    DB_END();
    // -------- phase b: rows 32-63; retire K-tile k+1's loads; one fragment of the previous tile
    readA23(sb);
    if (more) issueA(ld, slot2);
    // in-order queue: ... [A(k+1)] [stores of K-tile k-1] [B(k+2) x2, A(k+2) x4]: retire through A(k+1)
    if (more) {
      if (stores_prev) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      if (stores_prev) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    DB_MID();
    stores_prev = 0;
    if constexpr (U >= 0) {
      if (has_prev) {
        store_unit(std::integral_constant<int, 1 - P>{}, std::integral_constant<int, U>{});
        stores_prev = 2;
      }
    }
    DB_MFMA(P, 2);
    DB_END();
    if (more) advance(ld);
    slot = slot == 2 ? 0 : slot + 1;
  };
  auto flush = [&](auto Qtag, int from) __attribute__((always_inline)) {   // fragments `from`..3 of set Q
    if (from <= 0) store_unit(Qtag, std::integral_constant<int, 0>{});
    if (from <= 1) store_unit(Qtag, std::integral_constant<int, 1>{});
    if (from <= 2) store_unit(Qtag, std::integral_constant<int, 2>{});
    if (from <= 3) store_unit(Qtag, std::integral_constant<int, 3>{});
  };
  auto tile = [&](auto Ptag) __attribute__((always_inline)) {
    constexpr int P = decltype(Ptag)::value;
    using NoU = std::integral_constant<int, -1>;
    if (nk > 0) ktile(Ptag, std::integral_constant<int, 0>{});
    if (nk > 1) ktile(Ptag, std::integral_constant<int, 1>{});
    if (nk > 2) ktile(Ptag, std::integral_constant<int, 2>{});
    if (nk > 3) ktile(Ptag, std::integral_constant<int, 3>{});
    for (int t = 4; t < nk; ++t) ktile(Ptag, NoU{});
    // K loops shorter than 4 K-tiles: the rest of the PREVIOUS tile goes out now
    if (has_prev && nk < 4) flush(std::integral_constant<int, 1 - P>{}, nk);
    has_prev = true;
    pm0 = cur.m0; pn0 = cur.n0;
    ++cur.j;
    if (cur.j < nmy) {
      const int w = cs + idx + cur.j * bpx;
      const int tm = w / p.tiles_n, tn = w - tm * p.tiles_n;
      cur.m0 = tm * 256; cur.n0 = tn * 128;
    }
  };

  for (int jt = 0; jt < nmy; jt += 2) {
    tile(std::integral_constant<int, 0>{});
    if (jt + 1 < nmy) tile(std::integral_constant<int, 1>{});
  }
  // the last tile's results
  if ((nmy - 1) & 1) flush(std::integral_constant<int, 1>{}, 0);
  else flush(std::integral_constant<int, 0>{}, 0);
  if (grp == 0) __builtin_amdgcn_s_barrier();   // balance the extra barrier of waves 4-7
#undef DB_MFMA
#undef DB_MID
#undef DB_END
  if (PROBE != 0 && p.dbg && tid == 0) {
    p.dbg[bid * 4 + 2] = __builtin_amdgcn_s_memtime();
    p.dbg[bid * 4 + 3] = __builtin_amdgcn_s_memrealtime();
  }
}

// =====================================================================================
// v2 (NOT YET RUN ON A GPU): same idea, shaped for the load LATENCY that v1 exposed.  With
// K-tiles of ~1300 cycles a 3-stage ring gives the loads one K-tile of lead, less than the
// latency of the HBM-streamed operand (A = activations: every row panel is a first touch).
// Tile 128 (M) x 256 (N): the streamed operand is only a third of the bytes of a stage, so it
// gets a FOUR-stage ring (16 KiB stages, loads 3 K-tiles ahead, >= 2 K-tiles of lead) while the
// L2-resident weights keep three 32 KiB stages (2 ahead): 64 + 96 = 160 KiB.  8 waves as
// 2(M) x 4(N), wave tile 64 x 64, two accumulator sets, trickled epilogue as in v1.
constexpr int D2_NA = 4, D2_NB = 3;
constexpr int D2_A_BYTES = D2_NA * HALF;                    // 65536
constexpr int D2_SMEM = D2_A_BYTES + D2_NB * 2 * HALF;      // 163840

template <int PROBE = 0>
__global__ __launch_bounds__(512, 2) void gemm_dbuf2_kernel(DbParams p) {   // tiles_n = N/256, ntiles = (M/128)*(N/256)
  __shared__ __attribute__((aligned(1024))) char smem[D2_SMEM];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2;             // 64-row block of the tile AND ping-pong group
  const int wn = wave & 3;              // 64-column block
  const int lr = lane & 15, lg = lane >> 4;

  const int bid = blockIdx.x, G = gridDim.x;
  const int nwork = p.ntiles;
  const int xcd = bid & 7, idx = bid >> 3;
  const int q8 = nwork >> 3, r8 = nwork & 7;
  const int cs = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int cl = q8 + (xcd < r8 ? 1 : 0);
  const int bpx = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
  const int nmy = idx < cl ? (cl - idx + bpx - 1) / bpx : 0;
  if (nmy == 0) return;
  const int nk = p.K >> 6;

  auto load_item = [&](DbCursor& c) __attribute__((always_inline)) {
    const int w = cs + idx + c.j * bpx;
    const int tm = w / p.tiles_n, tn = w - tm * p.tiles_n;
    c.m0 = tm * 128; c.n0 = tn * 256;
    c.nk = nk;
    c.offA = (long)c.m0 * p.lda;
    c.offB = (long)c.n0 * p.ldb;
  };
  auto advance = [&](DbCursor& c) __attribute__((always_inline)) {
    if (++c.t == c.nk) {
      c.t = 0;
      ++c.j;
      if (c.j < nmy) load_item(c);
    }
  };

  const int r = wave * 8 + (lane >> 3);
  const int pos = lane & 7;
  const int ch = pos ^ kswz(r);
  const bf16* srcA = p.A + (long)r * p.lda + ch * 8;
  const bf16* srcB = p.B + (long)r * p.ldb + ch * 8;
  const long gA = 64 * p.lda, gB = 64 * p.ldb, hB = 128 * p.ldb;
  const int wave_off = wave * 1024;
  auto issueA = [&](const DbCursor& c, int slot) __attribute__((always_inline)) {   // 2 DMA instructions per thread
    char* d = smem + slot * HALF + wave_off;
    const bf16* s = srcA + c.offA + (long)c.t * 64;
    glds16(s, d);
    glds16(s + gA, d + 8192);
  };
  auto issueB = [&](const DbCursor& c, int slot) __attribute__((always_inline)) {   // 4 DMA instructions per thread
    char* d = smem + D2_A_BYTES + slot * 2 * HALF + wave_off;
    const bf16* s = srcB + c.offB + (long)c.t * 64;
    glds16(s, d);
    glds16(s + gB, d + 8192);
    glds16(s + hB, d + HALF);
    glds16(s + hB + gB, d + HALF + 8192);
  };

  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int brow = (lr >> 2) * 8 + (lr & 3);
  const uint32_t ra0 = lds0 + (wm * 64 + lr) * 128 + ((lg ^ kswz(lr)) << 4);
  const uint32_t ra1 = ra0 ^ 64;
  const uint32_t rb0 = lds0 + D2_A_BYTES + (wn >> 1) * HALF + ((wn & 1) * 64 + brow) * 128 + ((lg ^ kswz(brow)) << 4);
  const uint32_t rb1 = rb0 ^ 64;

  f32x4 acc[2][4][4];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[q][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[4][2], bfg[4][2];

  auto readA01 = [&](uint32_t sa) __attribute__((always_inline)) {
    const uint32_t a0 = ra0 + sa, a1 = ra1 + sa;
    af[0][0] = lds_read128<0>(a0);     af[0][1] = lds_read128<0>(a1);
    af[1][0] = lds_read128<2048>(a1);  af[1][1] = lds_read128<2048>(a0);
  };
  auto readA23 = [&](uint32_t sa) __attribute__((always_inline)) {
    const uint32_t a0 = ra0 + sa, a1 = ra1 + sa;
    af[2][0] = lds_read128<4096>(a0);  af[2][1] = lds_read128<4096>(a1);
    af[3][0] = lds_read128<6144>(a1);  af[3][1] = lds_read128<6144>(a0);
  };
  auto readB = [&](uint32_t sb) __attribute__((always_inline)) {
    const uint32_t b0 = rb0 + sb, b1 = rb1 + sb;
    bfg[0][0] = lds_read128<0>(b0);     bfg[0][1] = lds_read128<0>(b1);
    bfg[1][0] = lds_read128<512>(b1);   bfg[1][1] = lds_read128<512>(b0);
    bfg[2][0] = lds_read128<4096>(b0);  bfg[2][1] = lds_read128<4096>(b1);
    bfg[3][0] = lds_read128<4608>(b1);  bfg[3][1] = lds_read128<4608>(b0);
  };

#define D2_MFMA(Q, I0)                                                                        \
  do {                                                                                        \
    __builtin_amdgcn_s_setprio(1);                                                            \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                          \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                           \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                         \
          acc[Q][(I0) + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                      \
              bfg[j][ks], af[(I0) + i][ks], acc[Q][(I0) + i][j], 0, 0, 0);                    \
    __builtin_amdgcn_s_setprio(0);                                                            \
  } while (0)
#define D2_MID()                                          \
  do {                                                    \
    __builtin_amdgcn_s_barrier();                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);                    \
  } while (0)
#define D2_END()                             \
  do {                                       \
    __builtin_amdgcn_sched_barrier(0);       \
    __builtin_amdgcn_s_barrier();            \
  } while (0)
#define D2_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

  int pm0 = 0, pn0 = 0;
  const long lane_off = (long)(wm * 64 + lr) * p.ldc + wn * 64 + lg * 8;
  auto store_unit = [&](auto Qtag, auto Utag) __attribute__((always_inline)) {
    constexpr int Q = decltype(Qtag)::value, U = decltype(Utag)::value;
    const long tile_off = (long)(pm0 + U * 16) * p.ldc + pn0;
    bf16* c = p.C + tile_off + lane_off;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const f32x4 x = acc[Q][U][2 * hh], y = acc[Q][U][2 * hh + 1];
      u32x4 o;
      o[0] = pack_bf2(x[0] * p.alpha, x[1] * p.alpha);
      o[1] = pack_bf2(x[2] * p.alpha, x[3] * p.alpha);
      o[2] = pack_bf2(y[0] * p.alpha, y[1] * p.alpha);
      o[3] = pack_bf2(y[2] * p.alpha, y[3] * p.alpha);
      if (PROBE != 5) *reinterpret_cast<u32x4*>(c + hh * 32) = o;
      else asm volatile("" ::"v"(o));
      acc[Q][U][2 * hh] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc[Q][U][2 * hh + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  bool has_prev = false;

  // ---- prologue: B(0) A(0) | B(1) A(1) A(2) in flight; K-tile 0 landed
  DbCursor cur{};
  cur.j = 0; cur.t = 0;
  load_item(cur);
  DbCursor ca = cur, cb = cur;     // A stream runs 3 K-tiles ahead of the math, B stream 2
  issueB(cb, 0); issueA(ca, 0);
  advance(ca); advance(cb);
  int a_prev = 0;                  // A DMA issued after the most recent B issue (see the wait below)
  {
    int after = 0;
    if (cb.j < nmy) { issueB(cb, 1); advance(cb); after += 4; }
    if (ca.j < nmy) { issueA(ca, 1); advance(ca); after += 2; }
    if (ca.j < nmy) { issueA(ca, 2); advance(ca); after += 2; a_prev = 2; }
    if (after == 8) D2_WAIT(8);
    else if (after == 6) D2_WAIT(6);
    else D2_WAIT(0);
  }
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier behind waves 0-3

  int slotA = 0, slotB = 0, stores_prev = 0;

  auto ktile = [&](auto Ptag, auto Utag) __attribute__((always_inline)) {
    constexpr int P = decltype(Ptag)::value, U = decltype(Utag)::value;
    const uint32_t sa = slotA * HALF, sb = slotB * 2 * HALF;
    const int slotA3 = (slotA + 3) & 3;                 // K-tile k+3 -> the slot K-tile k-1 used
    const int slotB2 = slotB == 0 ? 2 : slotB - 1;      // K-tile k+2 -> the slot K-tile k-1 used
    const bool moreA = ca.j < nmy, moreB = cb.j < nmy;
    // -------- phase a
    readA01(sa);
    readB(sb);
    if (moreB) issueB(cb, slotB2);
    D2_MID();
    D2_MFMA(P, 0);
    D2_END();
    // -------- phase b
    readA23(sa);
    if (moreA) issueA(ca, slotA3);
    // in-order queue: ... B(k+1)x4 | A(k+2)x2 [a_prev] | stores(k-1) | B(k+2)x4 | A(k+3)x2 :
    // K-tile k+1 needs everything up to and including B(k+1)
    {
      const int allow = a_prev + stores_prev + (moreB ? 4 : 0) + (moreA ? 2 : 0);
      if (allow >= 10) D2_WAIT(10);
      else if (allow >= 8) D2_WAIT(8);
      else if (allow >= 6) D2_WAIT(6);
      else if (allow >= 4) D2_WAIT(4);
      else if (allow >= 2) D2_WAIT(2);
      else D2_WAIT(0);
    }
    D2_MID();
    stores_prev = 0;
    if constexpr (U >= 0) {
      if (has_prev) {
        store_unit(std::integral_constant<int, 1 - P>{}, std::integral_constant<int, U>{});
        stores_prev = 2;
      }
    }
    D2_MFMA(P, 2);
    D2_END();
    a_prev = moreA ? 2 : 0;
    if (moreA) advance(ca);
    if (moreB) advance(cb);
    slotA = (slotA + 1) & 3;
    slotB = slotB == 2 ? 0 : slotB + 1;
  };
  auto flush = [&](auto Qtag, int from) __attribute__((always_inline)) {
    if (from <= 0) store_unit(Qtag, std::integral_constant<int, 0>{});
    if (from <= 1) store_unit(Qtag, std::integral_constant<int, 1>{});
    if (from <= 2) store_unit(Qtag, std::integral_constant<int, 2>{});
    if (from <= 3) store_unit(Qtag, std::integral_constant<int, 3>{});
  };
  auto tile = [&](auto Ptag) __attribute__((always_inline)) {
    constexpr int P = decltype(Ptag)::value;
    using NoU = std::integral_constant<int, -1>;
    if (nk > 0) ktile(Ptag, std::integral_constant<int, 0>{});
    if (nk > 1) ktile(Ptag, std::integral_constant<int, 1>{});
    if (nk > 2) ktile(Ptag, std::integral_constant<int, 2>{});
    if (nk > 3) ktile(Ptag, std::integral_constant<int, 3>{});
    for (int t = 4; t < nk; ++t) ktile(Ptag, NoU{});
    if (has_prev && nk < 4) flush(std::integral_constant<int, 1 - P>{}, nk);
    has_prev = true;
    pm0 = cur.m0; pn0 = cur.n0;
    ++cur.j;
    if (cur.j < nmy) {
      const int w = cs + idx + cur.j * bpx;
      const int tm = w / p.tiles_n, tn = w - tm * p.tiles_n;
      cur.m0 = tm * 128; cur.n0 = tn * 256;
    }
  };

  for (int jt = 0; jt < nmy; jt += 2) {
    tile(std::integral_constant<int, 0>{});
    if (jt + 1 < nmy) tile(std::integral_constant<int, 1>{});
  }
  if ((nmy - 1) & 1) flush(std::integral_constant<int, 1>{}, 0);
  else flush(std::integral_constant<int, 0>{}, 0);
  if (wm == 0) __builtin_amdgcn_s_barrier();
#undef D2_MFMA
#undef D2_MID
#undef D2_END
#undef D2_WAIT
}

}  // namespace

static void fill(void* d, size_t n_bf16, unsigned seed) {
  std::vector<unsigned short> h(n_bf16);
  unsigned s = seed;
  for (size_t i = 0; i < n_bf16; ++i) {
    s = s * 1664525u + 1013904223u;
    float f = ((s >> 8) & 0xffff) / 65536.0f * 2.f - 1.f;   // U(-1,1)
    unsigned u; memcpy(&u, &f, 4);
    h[i] = (unsigned short)(u >> 16);
  }
  hipMemcpy(d, h.data(), n_bf16 * 2, hipMemcpyHostToDevice);
}

template <typename F>
static float time_ms(F launch, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("HIP error: %s\n", hipGetErrorString(e));
  return ms / iters;
}

int main() {
  struct Shape { const char* name; int M, N, K; } shapes[] = {
      {"check 512x256x128", 512, 256, 128},   {"check 768x512x64 (1 K-tile)", 768, 512, 64},
      {"check 1024x768x320 (5 K-tiles)", 1024, 768, 320},
      {"qkv  fwd", 100352, 2304, 768},        {"out  dx ", 100352, 768, 768},
      {"fc1  fwd", 100352, 3072, 768},        {"fc1  dx ", 100352, 768, 3072},
      {"text qkv", 32768, 2304, 768},         {"text out", 32768, 768, 768}};
  const size_t maxA = (size_t)100352 * 3072, maxB = (size_t)3072 * 3072, maxC = (size_t)100352 * 3072;
  void *a, *b, *c0, *c1;
  hipMalloc(&a, maxA * 2); hipMalloc(&b, maxB * 2); hipMalloc(&c0, maxC * 2); hipMalloc(&c1, maxC * 2);
  fill(a, maxA, 12345u); fill(b, maxB, 999u);
  for (auto& s : shapes) {
    G256Params r{};
    r.A = (const bf16*)a; r.B = (const bf16*)b; r.C = c0; r.lda = s.K; r.ldb = s.K; r.ldc = s.N;
    r.M = s.M; r.N = s.N; r.K = s.K; r.aux_rows = 1; r.tiles_n = s.N / 256;
    r.ntiles = (s.M / 256) * r.tiles_n; r.epi = BV_EPI_NONE; r.out_f32 = 0; r.alpha = 1.f;
    r.ktiles_per_split = s.K / 64; r.splits = 1;
    const bool ref_ok = (s.N % 256) == 0;
    DbParams d{};
    d.A = (const bf16*)a; d.B = (const bf16*)b; d.C = (bf16*)c1; d.lda = s.K; d.ldb = s.K; d.ldc = s.N;
    d.M = s.M; d.N = s.N; d.K = s.K; d.tiles_n = s.N / 128; d.ntiles = (s.M / 256) * d.tiles_n; d.alpha = 1.f;
    const int g0 = r.ntiles < 256 ? r.ntiles : 256, g1 = d.ntiles < 256 ? d.ntiles : 256;
    hipMemset(c0, 0xff, (size_t)s.M * s.N * 2); hipMemset(c1, 0xee, (size_t)s.M * s.N * 2);
    if (ref_ok) hipLaunchKernelGGL((gemm256_kernel<true, 0, BV_EPI_NONE, false>), dim3(g0), dim3(512), 0, 0, r);
    hipLaunchKernelGGL((gemm_dbuf_kernel<0>), dim3(g1), dim3(512), 0, 0, d);
    hipDeviceSynchronize();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("%s: HIP error %s\n", s.name, hipGetErrorString(e)); return 1; }
    size_t bad = 0;
    if (ref_ok) {
      std::vector<unsigned short> h0((size_t)s.M * s.N), h1((size_t)s.M * s.N);
      hipMemcpy(h0.data(), c0, h0.size() * 2, hipMemcpyDeviceToHost);
      hipMemcpy(h1.data(), c1, h1.size() * 2, hipMemcpyDeviceToHost);
      for (size_t i = 0; i < h0.size(); ++i) bad += h0[i] != h1[i];
    }
    const double fl = 2.0 * s.M * s.N * s.K;
    const float t0 = ref_ok ? time_ms([&] { hipLaunchKernelGGL((gemm256_kernel<true, 0, BV_EPI_NONE, false>), dim3(g0), dim3(512), 0, 0, r); }, 5) : 0.f;
    const float t1 = time_ms([&] { hipLaunchKernelGGL((gemm_dbuf_kernel<0>), dim3(g1), dim3(512), 0, 0, d); }, 5);
    const float t5 = time_ms([&] { hipLaunchKernelGGL((gemm_dbuf_kernel<5>), dim3(g1), dim3(512), 0, 0, d); }, 5);
    printf("%-34s mismatches vs gemm256: %zu%s | gemm256 %.3f ms %6.0f TF | dbuf %.3f ms %6.0f TF | dbuf no-stores %.3f ms\n",
           s.name, bad, ref_ok ? "" : " (no reference: N % 256)", t0, t0 > 0 ? fl / t0 / 1e9 : 0.0, t1, fl / t1 / 1e9, t5);
    // ---- v2: 128 x 256 tiles
    if (ref_ok && s.M % 128 == 0) {
      DbParams d2 = d;
      d2.tiles_n = s.N / 256; d2.ntiles = (s.M / 128) * d2.tiles_n;
      const int g2 = d2.ntiles < 256 ? d2.ntiles : 256;
      hipMemset(c1, 0xee, (size_t)s.M * s.N * 2);
      hipLaunchKernelGGL((gemm_dbuf2_kernel<0>), dim3(g2), dim3(512), 0, 0, d2);
      hipDeviceSynchronize();
      std::vector<unsigned short> h0((size_t)s.M * s.N), h1((size_t)s.M * s.N);
      hipMemcpy(h0.data(), c0, h0.size() * 2, hipMemcpyDeviceToHost);
      hipMemcpy(h1.data(), c1, h1.size() * 2, hipMemcpyDeviceToHost);
      size_t bad2 = 0;
      for (size_t i = 0; i < h0.size(); ++i) bad2 += h0[i] != h1[i];
      const float u1 = time_ms([&] { hipLaunchKernelGGL((gemm_dbuf2_kernel<0>), dim3(g2), dim3(512), 0, 0, d2); }, 5);
      const float u5 = time_ms([&] { hipLaunchKernelGGL((gemm_dbuf2_kernel<5>), dim3(g2), dim3(512), 0, 0, d2); }, 5);
      printf("%-34s   v2 (128x256, A 4 stages): mismatches %zu | %.3f ms %6.0f TF | no-stores %.3f ms\n", "", bad2, u1,
             fl / u1 / 1e9, u5);
    }
  }
  return 0;
}
