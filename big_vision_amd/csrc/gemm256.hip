// 256x256x64 bf16 MFMA GEMM for gfx950 (MI355X): 8 waves, direct-to-LDS loads,
// ping-pong wave groups.  The fast path behind bv_gemm_bf16 for the large
// transformer projections (reference call sites: big_vision/models/vit.py:72,77,
// 93-98 and their backward transposes, trainers/proj/image_text/siglip.py:311).
//
//   KM = true  ("NT"): A [M][K] and B [N][K], reduction dim contiguous.
//        forward  Y = X W    with W^T taken from the transposed bf16 weight shadow
//        dX = dY W^T         with W in its natural Flax (in,out) layout
//   KM = false ("TN"): A [K][M] and B [K][N], reduction dim is the slow axis.
//        dW = X^T dY (split-K, fp32 atomics into the flat gradient buffer)
//
// Structure (one workgroup = one 256x256 C tile, 512 threads = 8 waves as 2(M) x 4(N),
// each wave owns 128x64 = acc[8][4] fragments of v_mfma_f32_16x16x32_bf16):
//   * operands are staged by global_load_lds_dwordx4 (no VGPR round trip) in
//     16 KiB half-tiles (128 rows x 64 k); A ring = 2 K-tiles, B ring = 3 K-tiles
//     -> 160 KiB LDS, one workgroup per CU;
//   * a K-tile is consumed in 4 phases (one 64x32 quadrant of the wave's C tile
//     x K=64 = 16 MFMAs each).  Every phase = {LDS fragment reads + ONE half-tile
//     of global->LDS loads} barrier {16 MFMAs} barrier.  Waves 4-7 run one
//     barrier behind waves 0-3, so on every SIMD one wave issues MFMAs while its
//     partner issues LDS reads / loads (the matrix pipe never waits for memory);
//   * loads run 6 half-tiles ahead of the math; the only VMEM wait is one counted
//     s_waitcnt vmcnt(4) per K-tile (never 0 in steady state);
//   * LDS images are written lane-linearly by the DMA, so the bank-conflict
//     swizzle is applied to the per-lane GLOBAL source address and again on the
//     fragment read (same involution on both sides).
//
// LDS images.  KM: half-tile = [128 rows][8 x 16 B]; position p of row r holds
// global k-chunk p ^ f(r); fA(r) = (r>>1)&7, fB(r) = ((r>>4)&3)<<1 | (r>>1)&1 —
// both make every ds_read_b128 lane group hit 16 distinct 16-B bank slots.
// B fragment j of a wave maps lane-row c to column (c>>2)*16 + j*4 + (c&3), so a
// lane ends up with 16 CONTIGUOUS output columns per row (32/64-byte stores).
// !KM: half-tile = [64 k][128 rows] in natural order (256 B per k-row); 32-B slot q
// of k-row k lives at slot q ^ swzk(k); fragments come out of ds_read_b64_tr_b16.
#include "bv_common.h"
#include "bvhip_internal.h"
#include <type_traits>

namespace {

constexpr int HALF = 16384;                // one half-tile, bytes
constexpr int A_STAGES = 2, B_STAGES = 3;
constexpr int A_BYTES = A_STAGES * 2 * HALF;
constexpr int SMEM = (A_STAGES + B_STAGES) * 2 * HALF;  // 163840 = all of the CU's LDS

struct G256Params {
  const bf16* A;
  const bf16* B;
  void* C;
  void* C2;
  const float* bias;
  const void* aux;
  float* colsum;       // GELU_BWD epilogues: colsum[n] += sum_m C[m][n] (the Dense_0 bias gradient)
  long lda, ldb, ldc, ldaux;
  int M, N, K;
  int aux_rows;
  int tiles_n;
  int ntiles;          // tiles_m * tiles_n
  int group_n;         // k-major tile order: column tiles are walked in groups of this many (tile_mn below)
  int splits;          // split-K factor (work items = ntiles * splits)
  int ktiles_per_split;
  int epi;
  int out_f32;
  float alpha;
  float* slab;         // split-K partial slabs (TN), or nullptr
  int skew_cycles;     // > 0: spread the workgroups' start over this many cycles (see kernel)
  int skew_mode;       // 0: one phase per XCD; 1: one phase per workgroup (idx-major, so the
                       //    workgroups that own one tile fewer start last)
  int nt;              // bit 0: nontemporal (streaming) stores of C / C2; bit 1: nontemporal aux loads
  int pre_issue;       // 1: issue the next tile's K-tile 1/2 loads ahead of the epilogue stores
  long* dbg;           // probes only: [bid*4 + {0,1,2,3}] = shader-clock / 100 MHz REFCLK stamps at start / end
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gl_void;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ int swzk(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

// ---- LDS reads: inline asm so the compiler attaches no implicit vmcnt/lgkmcnt
// waits to them (it would drain the in-flight DMA before every read); every
// consumer is fenced by an explicit s_waitcnt lgkmcnt(0) + sched_barrier.
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read128(uint32_t addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return __builtin_bit_cast(bf16x8, v);
}
template <int OFF>
__device__ __forceinline__ s16x4 lds_read_tr64(uint32_t addr) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

// Cache policy of the operand DMA (the instruction's aux field: 1 = sc0, 2 = nt, 16 = sc1), per operand.  0 / 0 ships;
// tools/gemm_cachepolicy_ab.py builds the other combinations (-DBV_GLDS_AUX_A=.. -DBV_GLDS_AUX_B=..) and times them.
#ifndef BV_GLDS_AUX_A
#define BV_GLDS_AUX_A 0
#endif
#ifndef BV_GLDS_AUX_B
#define BV_GLDS_AUX_B 0
#endif
template <int AUX = 0>
__device__ __forceinline__ void glds16(const bf16* src, char* dst_wave_base) {
  __builtin_amdgcn_global_load_lds((gl_void*)src, (lds_void*)dst_wave_base, 16, 0, AUX);
}

// 16-byte global stores / loads with an optional streaming (nontemporal) hint: the outputs of
// the big projections (hundreds of MB) are never re-read before they fall out of the 4 MiB
// L2, so allocating them there only evicts the weight panel every workgroup of the XCD shares.
__device__ __forceinline__ void st16(void* ptr, u32x4 v, bool nt) {
  if (nt) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(ptr));
  else *reinterpret_cast<u32x4*>(ptr) = v;
}
__device__ __forceinline__ u32x4 ld16(const void* ptr, bool nt) {
  if (nt) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ptr));
  return *reinterpret_cast<const u32x4*>(ptr);
}

// Tile index -> (row tile, column tile) of the k-major kernels.  The ids an XCD's workgroups hold at one time are
// consecutive, so their order decides what that XCD's 4 MiB L2 has to keep: with the column tile fastest over ALL
// tiles_n columns (N = 3072: 12 weight panels = 4.7 MB) the weight panels fall out between two rounds and every
// tile re-fetches its 393 KB panel from the fabric.  In groups of `g` column tiles (id order: column inside the
// group, then row tile, then group) the XCD works on g weight panels that stay resident while the activation row
// panels stream through, each shared by g workgroups.  g >= tiles_n is the plain order.
__device__ __forceinline__ void tile_mn(int tile, int tiles_n, int ntiles, int g, int& tm, int& tn) {
  if (g <= 0 || g >= tiles_n) { tm = tile / tiles_n; tn = tile - tm * tiles_n; return; }
  const int tiles_m = ntiles / tiles_n;
  const int per = tiles_m * g;
  const int grp = tile / per, rem = tile - grp * per;
  const int w = min(g, tiles_n - grp * g);    // the last group may be narrower
  tm = rem / w;
  tn = grp * g + (rem - tm * w);
}

// PROBE != 0 variants (bottleneck ablations; most of them produce garbage results) exist ONLY when a
// translation unit under tools/probes/ defines BV_GEMM256_PROBES before including this file: 1 = LDS fragment
// reads only for the first K-tile, 2 = (TN) plain b128 reads instead of transpose reads, 3 = no DMA after the
// prologue, 5 = no epilogue stores (bf16 outputs), ...  In the library build the macro is not defined: every
// probe branch below is the constant `false`, the probe-only code blocks are not compiled, and instantiating
// PROBE != 0 is a compile error (tests/test_codegen_cpu.py also checks the exported kernel names).
#ifdef BV_GEMM256_PROBES
#define BV_PROBE(n) (PROBE == (n))
#define BV_PROBING (PROBE != 0)
#else
#define BV_PROBE(n) false
#define BV_PROBING false
#endif
//
// The kernel is PERSISTENT: the grid is one workgroup per CU (<= 256); each workgroup
// walks a list of work items (tile, split) and keeps the DMA pipeline running across
// item boundaries, so the loads of the next tile's first K-tiles are in flight while
// the current tile's epilogue runs (no per-tile pipeline fill).
struct Cursor {   // position of a load stream / of the math: item j (of this block), K-tile t
  int j, t, nk;
  long offA, offB;   // element offsets of K-tile 0 of item j (added to the per-lane pointers)
  int m0, n0, tile, split;
};

// 16-B chunk swizzle of the k-major LDS images: chunk c of row r sits at c ^ kswz(r).
// Conflict-free for ds_read_b128 under BOTH fragment-row maps used below (rows
// j*16 + c and (j>>1)*32 + (c>>2)*8 + (j&1)*4 + (c&3)), brute-forced over the lane groups.
__device__ __forceinline__ int kswz(int r) {
  return ((r >> 1) & 1) | (((r >> 3) & 1) << 1) | ((((r >> 2) ^ (r >> 4)) & 1) << 2);
}

template <bool KM, int PROBE = 0, int EPI = BV_EPI_NONE, bool OUTF32 = false>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(G256Params p) {
  __shared__ __attribute__((aligned(1024))) char smem[SMEM];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int lr = lane & 15, lg = lane >> 4;
#ifndef BV_GEMM256_PROBES
  static_assert(PROBE == 0, "probe variants are compiled only under tools/probes/ (BV_GEMM256_PROBES)");
#endif

  // ---- XCD-aware work distribution.  Block b runs on XCD b%8.  Work ids (tile index
  // fastest, n fastest inside it, then split) are cut into 8 contiguous chunks, one per
  // XCD; the blocks of an XCD take consecutive ids of their chunk, round after round, so
  // co-resident blocks share A row panels (NT) or the same K-chunk of both operands (TN
  // split-K) through that XCD's L2.
  const int bid = blockIdx.x, G = gridDim.x;
  const int nwork = p.ntiles * p.splits;
  const int xcd = bid & 7, idx = bid >> 3;
  const int q8 = nwork >> 3, r8 = nwork & 7;
  const int cs = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int cl = q8 + (xcd < r8 ? 1 : 0);
  const int bpx = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
  const int nmy = idx < cl ? (cl - idx + bpx - 1) / bpx : 0;
  if (nmy == 0) return;
  const int nk_all = p.K >> 6;
  if (BV_PROBING && p.skew_mode >= 2) {   // probe: only a subset of the workgroups runs (its usual work list)
    const int m = p.skew_mode;
    const bool on = m == 2 ? xcd == 0 : m == 3 ? (idx & 7) == 0 : m == 4 ? bid == 0 : m == 5 ? idx == 0
                  : m == 6 ? (xcd == 0 && idx < 8) : m == 7 ? xcd < 4 : true;
    if (!on) return;
  }
  if (BV_PROBING && p.dbg && tid == 0) {
    p.dbg[bid * 4 + 0] = __builtin_amdgcn_s_memtime();
    p.dbg[bid * 4 + 1] = __builtin_amdgcn_s_memrealtime();
  }
  // PROBE 9: fine-grained s_memtime stamps of wave 0 (blocks 0 and 1): one per K-tile end, one
  // at the epilogue start and one at its end.
  int stamp_n = 0;
  auto stamp = [&]() {
    if (BV_PROBE(9) && p.dbg && tid == 0 && bid < 2 && stamp_n < 1000)
      p.dbg[1024 + bid * 1024 + stamp_n++] = __builtin_amdgcn_s_memtime();
  };
  auto tstamp = [&](int jt) {
    if (BV_PROBE(10) && p.dbg && tid == 0 && (bid & 7) == 0 && jt < 30)
      p.dbg[1024 + (bid >> 3) * 32 + jt] = __builtin_amdgcn_s_memrealtime();
  };
  // All workgroups of a launch run identical tiles, so left alone they stay in lockstep
  // and hit their epilogues together: a chip-wide store burst during which nobody
  // computes (the next tile's first counted vmcnt wait also retires the older stores).
  // For launches of many rounds the host asks for a start skew of one tile period spread
  // over the workgroups of each XCD; the phases persist, the store traffic becomes steady.
  if (p.skew_cycles > 0) {
    const int rank = p.skew_mode ? idx * 8 + xcd : xcd * (G >> 3);
    const int n = (int)(((long)p.skew_cycles * rank / G) >> 10);   // s_sleep 16 ~ 1024 cycles
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(16);
  }

  auto load_item = [&](Cursor& c) {
    const int w = cs + idx + c.j * bpx;
    c.split = w / p.ntiles;
    c.tile = w - c.split * p.ntiles;
    int tm, tn;
    tile_mn(c.tile, p.tiles_n, p.ntiles, KM ? p.group_n : 0, tm, tn);
    c.m0 = tm * 256; c.n0 = tn * 256;
    const int kt0 = c.split * p.ktiles_per_split;
    c.nk = min(p.ktiles_per_split, nk_all - kt0);
    if constexpr (KM) {
      c.offA = (long)c.m0 * p.lda + (long)kt0 * 64;
      c.offB = (long)c.n0 * p.ldb + (long)kt0 * 64;
    } else {
      c.offA = (long)kt0 * 64 * p.lda + c.m0;
      c.offB = (long)kt0 * 64 * p.ldb + c.n0;
    }
  };
  auto advance = [&](Cursor& c) {
    if (++c.t == c.nk) {
      c.t = 0;
      ++c.j;
      if (c.j < nmy) load_item(c);
    }
  };

  // ---- per-lane global source pointers of the DMA (item offset 0, half 0, piece g = 0)
  const bf16* srcA;
  const bf16* srcB;
  long stepA, stepB;      // advance per K-tile
  long gA, gB;            // offset of piece g = 1
  long hA, hB;            // offset of half 1
  if constexpr (KM) {
    const int r = wave * 8 + (lane >> 3);           // row within the half-tile (g = 0)
    const int pos = lane & 7;
    const int cA = pos ^ kswz(r);
    const int cB = pos ^ kswz(r);
    srcA = p.A + (long)r * p.lda + cA * 8;
    srcB = p.B + (long)r * p.ldb + cB * 8;
    stepA = 64; stepB = 64;
    gA = 64 * p.lda; gB = 64 * p.ldb;
    hA = 128 * p.lda; hB = 128 * p.ldb;
  } else {
    const int k = wave * 4 + (lane >> 4);           // k-row within the tile (g = 0)
    const int piece = lane & 15;
    const int qs = (piece >> 1) ^ swzk(k);
    const int off = qs * 16 + (piece & 1) * 8;
    srcA = p.A + (long)k * p.lda + off;
    srcB = p.B + (long)k * p.ldb + off;
    stepA = 64 * p.lda; stepB = 64 * p.ldb;
    gA = 32 * p.lda; gB = 32 * p.ldb;
    hA = 128; hB = 128;
  }
  char* const ldsA = smem;
  char* const ldsB = smem + A_BYTES;
  const int wave_off = wave * 1024;
  bool probe_no_dma = false;

  // issue one half-tile (2 DMA instructions per thread) of the K-tile under cursor c
  auto issueA = [&](const Cursor& c, int slot, int h) {
    if (BV_PROBE(3) && probe_no_dma) return;
    char* d = ldsA + (slot * 2 + h) * HALF + wave_off;
    const bf16* s = srcA + c.offA + (long)c.t * stepA + (h ? hA : 0);
    glds16<BV_GLDS_AUX_A>(s, d);
    glds16<BV_GLDS_AUX_A>(s + gA, d + 8192);
  };
  auto issueB = [&](const Cursor& c, int slot, int h) {
    if (BV_PROBE(3) && probe_no_dma) return;
    char* d = ldsB + (slot * 2 + h) * HALF + wave_off;
    const bf16* s = srcB + c.offB + (long)c.t * stepB + (h ? hB : 0);
    glds16<BV_GLDS_AUX_B>(s, d);
    glds16<BV_GLDS_AUX_B>(s + gB, d + 8192);
  };

#ifdef BV_GEMM256_PROBES
  // PROBE 16 (tools/probes/gemm_vf_probe.hip; bit-identical to PROBE 0, timing): both operand streams go through
  // VGPRs - global_load_dwordx4 into staging registers where the DMA instruction would be issued, ds_write_b128 into
  // the same LDS positions two phases later - instead of through global_load_lds (whose issue costs a wave ~60 cycles
  // per instruction).  Two instructions per phase in a fixed order, so every wait is vmcnt(4) in the steady state.
  u32x4 vstA[2][2], vstB[2][2];          // [half][piece]
  uint32_t vdA[2] = {0, 0}, vdB[2] = {0, 0};   // LDS byte offsets of the staged halves (wave-uniform)
  bool vpA[2] = {false, false}, vpB[2] = {false, false};   // a staged half waits for its ds_write
  auto vloadA = [&](const Cursor& c, int slot, int h) __attribute__((always_inline)) {
    const bf16* s0 = srcA + c.offA + (long)c.t * stepA + (h ? hA : 0);
    const bf16* s1 = s0 + gA;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(vstA[h][0]) : "v"(s0) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(vstA[h][1]) : "v"(s1) : "memory");
    vdA[h] = (slot * 2 + h) * HALF + wave_off;
    vpA[h] = true;
  };
  auto vloadB = [&](const Cursor& c, int slot, int h) __attribute__((always_inline)) {
    const bf16* s0 = srcB + c.offB + (long)c.t * stepB + (h ? hB : 0);
    const bf16* s1 = s0 + gB;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(vstB[h][0]) : "v"(s0) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(vstB[h][1]) : "v"(s1) : "memory");
    vdB[h] = (slot * 2 + h) * HALF + wave_off;
    vpB[h] = true;
  };
  // steady: the two phases since the staged half was requested issued their two loads each (4 younger requests)
  auto vstoreA = [&](int h, bool steady) __attribute__((always_inline)) {
    if (!vpA[h]) return;
    if (steady) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    char* d = ldsA + vdA[h] + lane * 16;
    *reinterpret_cast<u32x4*>(d) = vstA[h][0];
    *reinterpret_cast<u32x4*>(d + 8192) = vstA[h][1];
    asm volatile("" ::: "memory");
    vpA[h] = false;
  };
  auto vstoreB = [&](int h, bool steady) __attribute__((always_inline)) {
    if (!vpB[h]) return;
    if (steady) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    char* d = ldsB + vdB[h] + lane * 16;
    *reinterpret_cast<u32x4*>(d) = vstB[h][0];
    *reinterpret_cast<u32x4*>(d + 8192) = vstB[h][1];
    asm volatile("" ::: "memory");
    vpB[h] = false;
  };
#endif

  // ---- per-lane LDS read addresses (relative to the stage base)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  uint32_t ra0, ra1, rb0, rb1;   // A/B fragment base for k-step 0 / 1 (KM) or ka/kb rows (!KM)
  if constexpr (KM) {
    // A fragment i: rows i*16 + lr.  B fragment j: fp32 output -> rows j*16 + lr (a lane ends
    // up with 4 consecutive columns per fragment, 16-B stores, 64 B contiguous per row and
    // instruction); bf16 output -> rows (j>>1)*32 + (lr>>2)*8 + (j&1)*4 + (lr&3) (a lane ends
    // up with 8 consecutive columns per fragment PAIR: again 16-B stores, 64 B per row).
    // Bit 4 of the row (odd fragments) flips chunk bit 2 = byte 64: odd fragments swap the
    // two k-step bases instead of recomputing the address.
    const int brow = OUTF32 ? lr : (lr >> 2) * 8 + (lr & 3);
    ra0 = lds0 + wr * HALF + lr * 128 + ((lg ^ kswz(lr)) << 4);
    ra1 = ra0 ^ 64;
    rb0 = lds0 + A_BYTES + (wc >> 1) * HALF + ((wc & 1) * 64 + brow) * 128 + ((lg ^ kswz(brow)) << 4);
    rb1 = rb0 ^ 64;
  } else {
    // tr read: lane supplies the 8-byte chunk [k = lg*8 + (lr>>2) (+4)][rows 4*(lr&3)..+3]
    const int ka = lg * 8 + (lr >> 2), kb = ka + 4;
    ra0 = lds0 + wr * HALF + ka * 256 + ((lr & 3) << 3);
    ra1 = lds0 + wr * HALF + kb * 256 + ((lr & 3) << 3);
    rb0 = lds0 + A_BYTES + (wc >> 1) * HALF + ka * 256 + ((lr & 3) << 3);
    rb1 = lds0 + A_BYTES + (wc >> 1) * HALF + kb * 256 + ((lr & 3) << 3);
  }
  // !KM: 32-B slot swizzle depends on the k-row: slot (rb ^ swzk(k)); k-step ks adds 32 to k
  // (swzk unchanged), so the per-lane xor masks are constants:
  const int xka = KM ? 0 : swzk(lg * 8 + (lr >> 2));
  const int xkb = KM ? 0 : swzk(lg * 8 + (lr >> 2) + 4);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef BV_GEMM256_PROBES
  // PROBE 11 (tools/probes/gemm_r3_probe.hip, TIMING ONLY - the fragments are not re-mapped, results are
  // garbage): the same main loop issuing v_mfma_f32_32x32x16_bf16 - per quadrant 2 x 1 fragments of 32 x 32
  // x 4 k-steps of 16 = 8 MFMAs instead of 16, the same 32 accumulator and 48 operand registers.
  f32x16 acc32[4][2];
  if constexpr (BV_PROBE(11)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc32[i][j][r] = 0.f;
  }
#define BV_ACC(i, j, r) (BV_PROBE(11) ? acc32[(i) >> 1][(j) >> 1][(((i) & 1) * 2 + ((j) & 1)) * 4 + (r)] : acc[i][j][r])
#else
#define BV_ACC(i, j, r) acc[i][j][r]
#endif
  bf16x8 af[4][2], bfg[4][2];

  // A fragments of 64-row sub-tile `sub` (4 frags x 2 k-steps) of stage base `sa`
  bool probe_skip_reads = false;
  auto readA = [&](uint32_t sa, int sub) {
    if (BV_PROBE(1) && probe_skip_reads) return;
    if constexpr (KM || BV_PROBE(2)) {
      const uint32_t a0 = (ra0 + sa) & (BV_PROBE(2) ? ~15u : ~0u), a1 = (ra1 + sa) & (BV_PROBE(2) ? ~15u : ~0u);
      if (sub == 0) {
        af[0][0] = lds_read128<0>(a0);     af[0][1] = lds_read128<0>(a1);
        af[1][0] = lds_read128<2048>(a1);  af[1][1] = lds_read128<2048>(a0);
        af[2][0] = lds_read128<4096>(a0);  af[2][1] = lds_read128<4096>(a1);
        af[3][0] = lds_read128<6144>(a1);  af[3][1] = lds_read128<6144>(a0);
      } else {
        af[0][0] = lds_read128<8192>(a0);  af[0][1] = lds_read128<8192>(a1);
        af[1][0] = lds_read128<10240>(a1); af[1][1] = lds_read128<10240>(a0);
        af[2][0] = lds_read128<12288>(a0); af[2][1] = lds_read128<12288>(a1);
        af[3][0] = lds_read128<14336>(a1); af[3][1] = lds_read128<14336>(a0);
      }
    } else {
      // 16-row block rb = sub*4 + i lives at 32-B slot rb ^ swzk(k)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int rbk = sub * 4 + i;
        const uint32_t pa = ra0 + sa + ((rbk ^ xka) << 5);
        const uint32_t pb = ra1 + sa + ((rbk ^ xkb) << 5);
        const s16x4 x0 = lds_read_tr64<0>(pa), y0 = lds_read_tr64<0>(pb);
        const s16x4 x1 = lds_read_tr64<8192>(pa), y1 = lds_read_tr64<8192>(pb);
        af[i][0] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(x0, y0, 0, 1, 2, 3, 4, 5, 6, 7));
        af[i][1] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(x1, y1, 0, 1, 2, 3, 4, 5, 6, 7));
      }
    }
  };
  // B fragments j = 2*sub, 2*sub+1 (2 frags x 2 k-steps) of stage base `sb`
  auto readB = [&](uint32_t sb, int sub) {
    if (BV_PROBE(1) && probe_skip_reads) return;
    if constexpr (KM || BV_PROBE(2)) {
      const uint32_t b0 = (rb0 + sb) & (BV_PROBE(2) ? ~15u : ~0u), b1 = (rb1 + sb) & (BV_PROBE(2) ? ~15u : ~0u);
      constexpr int O1 = OUTF32 ? 2048 : 512, O2 = OUTF32 ? 4096 : 4096, O3 = OUTF32 ? 6144 : 4608;
      if (sub == 0) {
        bfg[0][0] = lds_read128<0>(b0);    bfg[0][1] = lds_read128<0>(b1);
        bfg[1][0] = lds_read128<O1>(b1);   bfg[1][1] = lds_read128<O1>(b0);
      } else {
        bfg[2][0] = lds_read128<O2>(b0);   bfg[2][1] = lds_read128<O2>(b1);
        bfg[3][0] = lds_read128<O3>(b1);   bfg[3][1] = lds_read128<O3>(b0);
      }
    } else {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = sub * 2 + jj;
        const int rbk = (wc & 1) * 4 + j;
        const uint32_t pa = rb0 + sb + ((rbk ^ xka) << 5);
        const uint32_t pb = rb1 + sb + ((rbk ^ xkb) << 5);
        const s16x4 x0 = lds_read_tr64<0>(pa), y0 = lds_read_tr64<0>(pb);
        const s16x4 x1 = lds_read_tr64<8192>(pa), y1 = lds_read_tr64<8192>(pb);
        const bf16x8 f0 = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(x0, y0, 0, 1, 2, 3, 4, 5, 6, 7));
        const bf16x8 f1 = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(x1, y1, 0, 1, 2, 3, 4, 5, 6, 7));
        if (sub == 0) { bfg[jj][0] = f0; bfg[jj][1] = f1; }
        else { bfg[2 + jj][0] = f0; bfg[2 + jj][1] = f1; }
      }
    }
  };

  // PROBE 14 / 15 (timing only): 14 = STATIC priority (waves 4-7 raise theirs once, no per-segment flips:
  // MI355X_MICROARCH.md, VALU arbitration item 4), 15 = no s_setprio at all
  if (BV_PROBE(14) && wr == 1) __builtin_amdgcn_s_setprio(1);
#ifdef BV_GEMM256_PROBES
#define BV_MFMA_QUAD_32(I0, J0)                                                                \
  do {                                                                                         \
      if (first) {                                                                             \
        _Pragma("unroll") for (int r2 = 0; r2 < 2; ++r2)                                       \
          acc32[((I0) >> 1) + r2][(J0) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(        \
              bfg[(J0)][0], af[r2 * 2][0], f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0); \
      } else {                                                                                 \
        _Pragma("unroll") for (int r2 = 0; r2 < 2; ++r2)                                       \
          acc32[((I0) >> 1) + r2][(J0) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(        \
              bfg[(J0)][0], af[r2 * 2][0], acc32[((I0) >> 1) + r2][(J0) >> 1], 0, 0, 0);       \
      }                                                                                        \
      _Pragma("unroll") for (int k4 = 1; k4 < 4; ++k4)                                         \
        _Pragma("unroll") for (int r2 = 0; r2 < 2; ++r2)                                       \
          acc32[((I0) >> 1) + r2][(J0) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(        \
              bfg[(J0) + (k4 >> 1)][k4 & 1], af[r2 * 2 + (k4 >> 1)][k4 & 1],                   \
              acc32[((I0) >> 1) + r2][(J0) >> 1], 0, 0, 0);                                    \
  } while (0)
#else
#define BV_MFMA_QUAD_32(I0, J0) do {} while (0)
#endif
#define BV_MFMA_QUAD(I0, J0)                                                                   \
  do {                                                                                         \
    if (!BV_PROBE(14) && !BV_PROBE(15)) __builtin_amdgcn_s_setprio(1);                             \
    if constexpr (BV_PROBE(11)) {                                                              \
      BV_MFMA_QUAD_32(I0, J0);                                                                 \
    } else                                                                                     \
    if (!BV_PROBE(4)) {                                                                          \
      /* first K-tile of a work item: the k-step-0 MFMAs take C = 0 (an inline constant), so   \
         the 128 accumulator registers are never cleared by VALU moves (512 cycles per wave    \
         and tile during which the SIMD's matrix pipe had nothing to do) */                    \
      if (first) {                                                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                          \
          _Pragma("unroll") for (int j = 0; j < 2; ++j)                                        \
            acc[(I0) + i][(J0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                 \
                bfg[(J0) + j][0], af[i][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);               \
      } else {                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                          \
          _Pragma("unroll") for (int j = 0; j < 2; ++j)                                        \
            acc[(I0) + i][(J0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                 \
                bfg[(J0) + j][0], af[i][0], acc[(I0) + i][(J0) + j], 0, 0, 0);                 \
      }                                                                                        \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                            \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                          \
          acc[(I0) + i][(J0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                   \
              bfg[(J0) + j][1], af[i][1], acc[(I0) + i][(J0) + j], 0, 0, 0);                   \
    }                                                                                          \
    if (!BV_PROBE(14) && !BV_PROBE(15)) __builtin_amdgcn_s_setprio(0);                             \
  } while (0)

  // PROBE 12 / 13 (timing only, tools/probes/gemm_r3_probe.hip): the same loop with HALF the workgroup
  // barriers - 12 drops the one behind every MFMA segment, 13 the one in front of it
#define BV_MID()                                          \
  do {                                                    \
    if (!BV_PROBE(13)) __builtin_amdgcn_s_barrier();        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);                    \
  } while (0)
#define BV_END()                                          \
  do {                                                    \
    __builtin_amdgcn_sched_barrier(0);                    \
    if (!BV_PROBE(12)) __builtin_amdgcn_s_barrier();        \
  } while (0)

  // ---- prologue: B(0), A(0), B(1) in flight; K-tile 0 must have landed.
  Cursor cur{};            // the math
  cur.j = 0; cur.t = 0;
  load_item(cur);
  Cursor ca = cur, cb = cur;   // A stream runs 1 K-tile ahead, B stream 2 K-tiles ahead
  issueB(cb, 0, 0); issueB(cb, 0, 1);
  issueA(ca, 0, 0); issueA(ca, 0, 1);
  advance(ca);
  advance(cb);
  if (cb.j < nmy) {
    issueB(cb, 1, 0); issueB(cb, 1, 1);
    advance(cb);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    cb.j = nmy;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

  int gk = 0;  // K-tiles consumed so far (ring position)
  int bs = 0;  // B ring slot of the current K-tile
  // Stores of one tile's epilogue per wave (they sit in the same in-order VMEM queue as the
  // DMA): with `pre` set, the loads K-tile 0 of the next tile would issue were issued BEFORE
  // the stores (into the ring slots the finished tile freed), and K-tile 0's counted wait
  // leaves the stores outstanding (vmcnt(4 + NSTORE)) instead of draining them.
  constexpr int NSTORE = KM ? ((OUTF32 || EPI == BV_EPI_GELU || EPI == BV_EPI_GELU_BWD_EMIT || EPI == BV_EPI_GELU_GD) ? 32 : 16) : 32;
  bool pre = false;
#ifdef BV_GEMM256_PROBES
  bool vf_prev = false;   // PROBE 16: the previous K-tile requested its B halves
#endif
  for (int jt = 0; jt < nmy; ++jt) {
    if (wr == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier behind
    const int nkc = cur.nk;
    for (int t = 0; t < nkc; ++t) {
      const uint32_t sa = (gk & 1) * 2 * HALF;
      const uint32_t sb = bs * 2 * HALF;
      const int bs1 = (bs == 2) ? 0 : bs + 1;          // slot of K-tile gk+1
      const int bs2 = (bs1 == 2) ? 0 : bs1 + 1;        // slot of K-tile gk+2
      const bool moreA = ca.j < nmy, moreB = cb.j < nmy;
      const bool doA = moreA && !pre, doB = moreB && !pre;
      const bool first = t == 0;
#ifdef BV_GEMM256_PROBES
      if constexpr (BV_PROBE(16)) {
        // VGPR-fed streams: A(k+1) requested in phases 0 / 1 and written in phases 2 / 3, B(k+2) requested in phases
        // 2 / 3 and written in phases 0 / 1 of the next K-tile.  The phase-3 write is read right behind the next
        // barrier by the wave group that runs one barrier ahead: it is waited for in front of BV_MID.
        const bool st_now = doA && doB, st_prev = vf_prev && doA;
        readA(sa, 0);
        readB(sb, 0);
        if (doA) vloadA(ca, (gk + 1) & 1, 0);
        vstoreB(0, st_prev);
        BV_MID();
        BV_MFMA_QUAD(0, 0);
        BV_END();
        readB(sb, 1);
        if (doA) vloadA(ca, (gk + 1) & 1, 1);
        vstoreB(1, st_prev);
        BV_MID();
        BV_MFMA_QUAD(0, 2);
        BV_END();
        readA(sa, 1);
        if (doB) vloadB(cb, bs2, 0);
        vstoreA(0, st_now);
        BV_MID();
        BV_MFMA_QUAD(4, 2);
        BV_END();
        if (doB) vloadB(cb, bs2, 1);
        vstoreA(1, st_now);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        BV_MID();
        BV_MFMA_QUAD(4, 0);
        BV_END();
        vf_prev = doB;
      } else
#endif
      {
      // -------- phase 0: quadrant (0,0)
      readA(sa, 0);
      readB(sb, 0);
      if (doA) issueA(ca, (gk + 1) & 1, 0);
      BV_MID();
      BV_MFMA_QUAD(0, 0);
      BV_END();
      // -------- phase 1: quadrant (0,1)
      readB(sb, 1);
      if (doA) issueA(ca, (gk + 1) & 1, 1);
      BV_MID();
      BV_MFMA_QUAD(0, 2);
      BV_END();
      // -------- phase 2: quadrant (1,1)
      readA(sa, 1);
      if (doB) issueB(cb, bs2, 0);
      BV_MID();
      BV_MFMA_QUAD(4, 2);
      BV_END();
      // -------- phase 3: quadrant (1,0); retire K-tile gk+1's loads for the next iteration
      if (pre) {
        // in-order queue: [A(k+1) x4] [B(k+2) x4 if moreB] [NSTORE stores]; retire through A(k+1)
        if (moreB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + NSTORE) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
        pre = false;
      } else if (moreB) {
        issueB(cb, bs2, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      BV_MID();
      BV_MFMA_QUAD(4, 0);
      BV_END();
      }
      if (moreA) advance(ca);
      if (moreB) advance(cb);
      bs = bs1;
      ++gk;
      stamp();
      probe_skip_reads = true;
      probe_no_dma = true;
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();   // re-align both wave groups for the epilogue
    stamp();
    // Every wave has finished reading the last K-tile: its ring slots are free.  Issue the next
    // tile's K-tile 1 (A) / K-tile 2 (B) now, ahead of the epilogue's stores.
    if ((KM || p.slab) && EPI != BV_EPI_GELU_BWD && EPI != BV_EPI_GELU_BWD_EMIT && EPI != BV_EPI_MUL && p.pre_issue) {
      const bool mA = ca.j < nmy, mB = cb.j < nmy;
      if (mA || mB) {
        const int b1 = (bs == 2) ? 0 : bs + 1, b2 = (b1 == 2) ? 0 : b1 + 1;
        if (mA) { issueA(ca, (gk + 1) & 1, 0); issueA(ca, (gk + 1) & 1, 1); }
        if (mB) { issueB(cb, b2, 0); issueB(cb, b2, 1); }
        pre = true;
      }
    }
    const int m0 = cur.m0, n0 = cur.n0, tile = cur.tile, split = cur.split;

  // ---- epilogue
  const int epi = p.epi;
  if constexpr (KM) {
    // lane holds, per row fragment i (m = m0 + wr*128 + i*16 + lr) and B fragment j, the 4
    // columns nc[j] .. nc[j]+3:  fp32 output: nc[j] = n0 + wc*64 + j*16 + lg*4;  bf16 output:
    // nc[j] = n0 + wc*64 + (j>>1)*32 + lg*8 + (j&1)*4 (fragments 2jp, 2jp+1 are adjacent: 8
    // columns = one 16-byte bf16 store).  Either way one store instruction covers 16 rows x
    // 64 contiguous bytes.  EPI is a compile-time constant and the auxiliary loads of 4 row
    // fragments are issued together before their first use (one memory round trip per
    // batch: the whole CU is in its epilogue at the same time, nothing else hides it).
    int nc[4], ncl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ncl[j] = OUTF32 ? j * 16 + lg * 4 : (j >> 1) * 32 + lg * 8 + (j & 1) * 4;   // inside the wave's 64 columns
      nc[j] = n0 + wc * 64 + ncl[j];
    }
    // Addresses: a wave-uniform 64-bit base per row fragment (scalar registers, scalar arithmetic) + one 32-bit lane
    // offset per column piece, the `saddr + voffset` form of the global instructions.  (Built per element as
    // (long)m * ldc + n the epilogue spent a third of its VALU instructions - quarter-rate 64-bit multiplies among
    // them - on addresses: 70 of 200 per tile and lane in the plain bf16 epilogue.)
    constexpr int ESZ = OUTF32 ? 4 : 2;
    const long crow = (long)(m0 + wr * 128) * p.ldc + n0 + wc * 64;     // wave-uniform: first element of the wave's block
    char* const Cb = reinterpret_cast<char*>(p.C) + crow * ESZ;
    char* const C2b = reinterpret_cast<char*>(p.C2) + crow * 2;
    const long cstep = 16L * p.ldc;                                     // elements between row fragments
    uint32_t vc[4];                                                     // lane offsets (bytes) of its four column pieces
#pragma unroll
    for (int j = 0; j < 4; ++j) vc[j] = (uint32_t)(lr * (BV_PROBE(7) ? 256 : (int)p.ldc) + ncl[j]) * ESZ;
    // Everything below works on PAIRS of adjacent columns (f32x2: v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32,
    // two fp32 lanes per VALU instruction): pair q of a row fragment = columns nc[q >> 1] + 2 (q & 1) + {0, 1}.
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
    const int mrow0 = m0 + wr * 128 + lr;
    const bool nts = (p.nt & 1) || BV_PROBE(6), ntl = p.nt & 2;
    // GBWD: epilogues that take a bf16 auxiliary operand of C's shape and feed the fused column sums
    constexpr bool GBWD = EPI == BV_EPI_GELU_BWD || EPI == BV_EPI_GELU_BWD_EMIT || EPI == BV_EPI_MUL;
    constexpr bool RESBF = EPI == BV_EPI_RESIDUAL && !OUTF32;   // bf16 residual stream: aux bf16, C bf16
    // row fragments per batch of auxiliary loads (GELU': 2, the variants are at the VGPR limit)
    constexpr int IB = GBWD ? 2 : 4;
#pragma unroll
    for (int ib = 0; ib < 8; ib += IB) {
      float4 ax[IB][4];
      uint4 hx[IB][2];
      if constexpr (EPI == BV_EPI_POS) {   // rows wrap around the table (m % aux_rows): per-element addresses
#pragma unroll
        for (int ii = 0; ii < IB; ++ii) {
          const int m = mrow0 + (ib + ii) * 16;
          const float* x = reinterpret_cast<const float*>(p.aux) + (long)(m % p.aux_rows) * p.ldaux;
#pragma unroll
          for (int j = 0; j < 4; ++j) ax[ii][j] = __builtin_bit_cast(float4, ld16(x + nc[j], ntl));
        }
      } else if constexpr (EPI == BV_EPI_RESIDUAL && OUTF32) {
        const char* Ab = reinterpret_cast<const char*>(p.aux) + ((long)(m0 + wr * 128) * p.ldaux + n0 + wc * 64) * 4;
#pragma unroll
        for (int ii = 0; ii < IB; ++ii) {
          const char* x = Ab + (long)(ib + ii) * 16 * p.ldaux * 4;   // wave-uniform
#pragma unroll
          for (int j = 0; j < 4; ++j)
            ax[ii][j] = __builtin_bit_cast(float4, ld16(x + (uint32_t)(lr * (int)p.ldaux + ncl[j]) * 4u, ntl));
        }
      } else if constexpr (GBWD || RESBF) {
        const char* Ab = reinterpret_cast<const char*>(p.aux) + ((long)(m0 + wr * 128) * p.ldaux + n0 + wc * 64) * 2;
        const uint32_t va0 = (uint32_t)(lr * (int)p.ldaux + ncl[0]) * 2u, va1 = (uint32_t)(lr * (int)p.ldaux + ncl[2]) * 2u;
#pragma unroll
        for (int ii = 0; ii < IB; ++ii) {
          const char* h = Ab + (long)(ib + ii) * 16 * p.ldaux * 2;   // wave-uniform
          hx[ii][0] = __builtin_bit_cast(uint4, ld16(h + va0, ntl));
          hx[ii][1] = __builtin_bit_cast(uint4, ld16(h + va1, ntl));
        }
      }
#pragma unroll
      for (int ii = 0; ii < IB; ++ii) {
        const int i = ib + ii;
        const int m = mrow0 + i * 16;
        // the GELU' variants sit at the VGPR limit: keep the scheduler from interleaving the (register-
        // hungry) evaluation of two row fragments
        if constexpr (GBWD) __builtin_amdgcn_sched_barrier(0);
        f32x2 v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2)
            v[j * 2 + h2] = __builtin_elementwise_fma(f32x2{BV_ACC(i, j, h2 * 2), BV_ACC(i, j, h2 * 2 + 1)}, alpha2, bv[j * 2 + h2]);
        if constexpr (OUTF32) {
          if constexpr (EPI == BV_EPI_RESIDUAL || EPI == BV_EPI_POS) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              v[j * 2 + 0] += f32x2{ax[ii][j].x, ax[ii][j].y};
              v[j * 2 + 1] += f32x2{ax[ii][j].z, ax[ii][j].w};
            }
          }
          char* c = Cb + (long)i * cstep * 4;   // wave-uniform
#pragma unroll
          for (int j = 0; j < 4; ++j)
            st16(c + vc[j], __builtin_bit_cast(u32x4, make_float4(v[j * 2].x, v[j * 2].y, v[j * 2 + 1].x, v[j * 2 + 1].y)), nts);
        } else {
          char* c = Cb + (long)i * cstep * 2;     // wave-uniform
          char* c2 = C2b + (long)i * cstep * 2;
          if (BV_PROBE(7))   // probe: every tile of a block overwrites the same 64 KiB (L2-resident, no HBM write-back; vc: pitch 256)
            c = reinterpret_cast<char*>(p.C) + ((long)bid * 128 + ((wr * 128 + i * 16) & 127)) * 512 + wc * 128;
          if (BV_PROBE(5)) {   // probe: keep ALL the math live (no DCE of MFMAs), skip the stores
#pragma unroll
            for (int q = 0; q < 8; ++q) asm volatile("" ::"v"(v[q]));
            continue;
          }
          // one 16-byte half (8 columns = 4 pairs) at a time: evaluate, reduce, convert, store - short live ranges
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            // bf16 auxiliary operand: word e of half hh holds the pair q = hh * 4 + e
            uint32_t aw[4] = {0, 0, 0, 0};
            if constexpr (GBWD || RESBF) { aw[0] = hx[ii][hh].x; aw[1] = hx[ii][hh].y; aw[2] = hx[ii][hh].z; aw[3] = hx[ii][hh].w; }
            uint32_t cw[4], gw[4];   // first / second bf16 output of this half, packed pairs
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              f32x2& x = v[hh * 4 + e];
              if constexpr (RESBF) {
                x += bf2_unpack(aw[e]);
              } else if constexpr (EPI == BV_EPI_MUL) {
                x *= bf2_unpack(aw[e]);   // dH = dG o gelu'(h): the derivative was written (bf16) by the forward's GELU_GD epilogue
              } else if constexpr (EPI == BV_EPI_GELU_BWD || EPI == BV_EPI_GELU_BWD_EMIT) {
                // EMIT: also C2 = gelu(aux), recomputed here (from the same bf16 pre-activation the forward
                // applied it to) instead of being kept from the forward
                // (gelu' is rounded to bf16 before the product, as GELU_GD stores it: all context kinds agree)
                uint32_t dw;
                mlp_act_from_h(aw[e], gw[e], dw);
                x *= bf2_unpack(dw);
              }
              if constexpr (GBWD) {
                cs[hh * 4 + e] += x;
                asm volatile("" : "+v"(cs[hh * 4 + e]));   // accumulate HERE (hipcc otherwise defers the 64 adds of a tile to its end: 40+ spilled pairs)
              }
              if constexpr (EPI == BV_EPI_GELU_GD) {
                // C = gelu(h), C2 = gelu'(h) of the bf16-rounded pre-activation h (which is not stored)
                uint32_t hw;
                mlp_act_words(x, hw, cw[e], gw[e]);
              } else {
                cw[e] = bf2_pack(x);
                // g = gelu(h) of the bf16-ROUNDED pre-activation h that is stored (and that the
                // backward differentiates / can recompute g from): forward and backward agree.
                if constexpr (EPI == BV_EPI_GELU) gw[e] = bf2_pack(gelu_tanh_pk(bf2_unpack(cw[e])));
                if constexpr (EPI == BV_EPI_GELU_G) cw[e] = bf2_pack(gelu_tanh_pk(bf2_unpack(cw[e])));   // the only output
              }
            }
            st16(c + vc[hh * 2], u32x4{cw[0], cw[1], cw[2], cw[3]}, nts);
            if constexpr (EPI == BV_EPI_GELU || EPI == BV_EPI_GELU_GD || EPI == BV_EPI_GELU_BWD_EMIT)
              st16(c2 + vc[hh * 2], u32x4{gw[0], gw[1], gw[2], gw[3]}, nts);
          }
        }
      }
    }
    if constexpr (GBWD) {
      // column sums of this wave's 128 x 64 block of dX (fp32, before the bf16 rounding): reduce
      // the 16 row lanes with DPP, one atomic per column and wave.
      if (p.colsum) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float sum = cs[e >> 1][e & 1];
          sum += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sum), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
          sum += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sum), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
          sum += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sum), 0x141, 0xF, 0xF, true));  // row_half_mirror
          sum += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, sum), 0x140, 0xF, 0xF, true));  // row_mirror
          if (lr == 0) unsafeAtomicAdd(p.colsum + nc[e >> 2] + (e & 3), sum);
        }
      }
    }
  } else {
    // TN (dW): lane holds C[m][n .. n+3], m = m0 + wr*128 + i*16 + lr, n = n0 + wc*64 + j*16 + lg*4.
    if (p.slab) {
      // split-K partial: fully coalesced 16-B stores into this block's 256 KiB slab,
      // layout [wave][i][j][lane][4]; gemm256_reduce_kernel sums the slabs into C.
      float* sl = p.slab + ((long)split * p.ntiles + tile) * 65536 + (wave * 32) * 256 + lane * 4;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<float4*>(sl + (i * 4 + j) * 256) =
              make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = m0 + wr * 128 + i * 16 + lr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = n0 + wc * 64 + j * 16 + lg * 4;
          float* c = reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n;
          if (epi == BV_EPI_ATOMIC) {
#pragma unroll
            for (int r = 0; r < 4; ++r) unsafeAtomicAdd(c + r, acc[i][j][r] * p.alpha);
          } else {
            *reinterpret_cast<float4*>(c) = make_float4(acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha,
                                                        acc[i][j][2] * p.alpha, acc[i][j][3] * p.alpha);
          }
        }
      }
    }
  }
    stamp();
    tstamp(jt);
    // ---- next work item: move the math cursor (the accumulators restart from C = 0 in the MFMAs)
    if constexpr (BV_PROBE(4)) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    cur.t = cur.nk - 1;
    advance(cur);
  }
#undef BV_MFMA_QUAD
#undef BV_MFMA_QUAD_32
#undef BV_MID
#undef BV_END
#undef BV_ACC
  if (BV_PROBING && p.dbg && tid == 0) {
    p.dbg[bid * 4 + 2] = __builtin_amdgcn_s_memtime();
    p.dbg[bid * 4 + 3] = __builtin_amdgcn_s_memrealtime();
  }
}

// ---------------------------------------------------------------------------------------------
// ROLLING-EPILOGUE variant of the k-major kernel (same tiles, LDS images, DMA ring and phase
// structure as gemm256_kernel<true>; reference call sites as there).
//
// gemm256_kernel stops at every tile boundary: all 8 waves convert and store their 128x64 block,
// the whole XCD pushes its C tiles through the L2 write path at the same time and the next tile's
// first K-tile waits behind the stores (measured: 9-10 k of the 43 k cycles of a K = 768 tile).
// Here there is NO epilogue phase.  The phases of a K-tile visit the wave's C quadrants in the
// order (0,0) (0,1) (1,1) (1,0); quadrant q of tile t is final after phase q of t's LAST K-tile
// and its registers are next written in phase q of tile t+1's FIRST K-tile.  Its epilogue
// (scale, bias, activation, convert, 4-8 stores per lane) runs in the LOAD segment of the phase
// after its last MFMA - while the partner wave group (one barrier behind / ahead) owns the matrix
// pipe: (0,0) (0,1) (1,1) in phases 1-3 of the last K-tile, (1,0) in phase 0 of the next tile.
// The K loop never drains: every VMEM wait is a counted s_waitcnt that leaves the youngest stores
// (and the loads behind them) in flight.
//   * first K-tile of a tile: the k-step-0 MFMAs take C = 0 (no accumulator clearing);
//   * +residual (BV_EPI_RESIDUAL, alpha = 1): the fp32 residual tile of the NEXT tile is loaded
//     straight INTO the accumulator registers of a quadrant right after that quadrant was stored
//     (three phases before its first MFMA), so x + f(x) costs no extra registers, no VALU adds and
//     its HBM read overlaps the K loop;
//   * bias: 16 floats per lane, reloaded in phase 1 of every tile's first K-tile.
// Loads with VGPR destinations are inline asm (hipcc would drain the DMA queue for any load it
// counts itself); their waits name the destinations ("+v") so no consumer moves above them.
// In-order VMEM queue per wave and K-tile kind (g = 2 DMA instructions, S / X = NS stores / NX
// residual loads of one quadrant, BL = 4 bias loads):
//   last  K-tile:  ph0 gA0 | ph1 gA1 S00 X00 | ph2 gB0 S01 X01 | ph3 gB1 WAIT(a) S11 X11
//   first K-tile:  ph0 gA0 S10 X10 W(X00) | ph1 gA1 BL W(X01) | ph2 gB0 W(X11) | ph3 gB1 vmcnt(4)
//   WAIT(a) retires gA1 of the last K-tile: 4 + 2 NS + 2 NX younger operations stay in flight.
// Whenever a K-tile does not issue its full set of DMA (the last two K-tiles of a workgroup's
// list) the counted waits fall back to vmcnt(0).  Requires K >= 128 (first != last K-tile).
__device__ __forceinline__ void gl_load16(f32x4& d, const void* ptr) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(ptr) : "memory");
}
// 16-byte load at (wave-uniform base in SGPRs) + (32-bit per-lane byte offset): no 64-bit VALU
// address arithmetic, one VGPR of address per lane
__device__ __forceinline__ void gl_load16s(f32x4& d, const void* sbase, uint32_t voff) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(sbase) : "memory");
}
// Waiting for asm loads: a counted s_waitcnt (no operands), THEN one empty statement that names the
// destinations read-write.  The wait itself must not carry the "+v" ties: hipcc satisfies a tie
// with register copies placed BEFORE the statement, i.e. before the wait - it did exactly that on
// one of two branches here and copied bias registers whose loads had not landed.  With the tie on
// a separate empty statement every such copy sits after the wait, and no consumer can be moved
// above it.
template <int N>
__device__ __forceinline__ void vm_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void vm_tie4(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3) {
  asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
}
__device__ __forceinline__ void vm_tie8(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, f32x4& a4, f32x4& a5,
                                        f32x4& a6, f32x4& a7) {
  asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
}

// STM = 1 (bf16 outputs): the stores of a quadrant are issued INSIDE the MFMA segment of the phase (one
// store behind every 4th / 2nd MFMA via sched_group_barrier) instead of in its load segment: the wave
// pushes store data while its own MFMAs execute.
template <int EPI, bool OUTF32, int STM = 0>
__global__ __launch_bounds__(512, 2) void gemm256r_kernel(G256Params p) {
  static_assert(EPI == BV_EPI_NONE || EPI == BV_EPI_GELU || EPI == BV_EPI_RESIDUAL, "epilogue");
  static_assert(OUTF32 == (EPI == BV_EPI_RESIDUAL), "fp32 output only with the residual epilogue");
  __shared__ __attribute__((aligned(1024))) char smem[SMEM];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int lr = lane & 15, lg = lane >> 4;
  constexpr int NS = EPI == BV_EPI_NONE ? 4 : 8;        // stores per quadrant and lane
  constexpr int NX = EPI == BV_EPI_RESIDUAL ? 8 : 0;    // residual loads per quadrant and lane
  constexpr int W_A = 4 + 2 * NS + 2 * NX;
  constexpr int W_X00 = 6 + 3 * NS + 3 * NX, W_X01 = 10 + 2 * NS + 2 * NX, W_X11 = 10 + NS + NX;
  static_assert(W_X00 < 64, "vmcnt is a 6-bit field");

  // ---- XCD-aware work distribution (as gemm256_kernel)
  const int bid = blockIdx.x, G = gridDim.x;
  const int nwork = p.ntiles;
  const int xcd = bid & 7, idx = bid >> 3;
  const int q8 = nwork >> 3, r8 = nwork & 7;
  const int cs = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  const int cl = q8 + (xcd < r8 ? 1 : 0);
  const int bpx = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
  const int nmy = idx < cl ? (cl - idx + bpx - 1) / bpx : 0;
  if (nmy == 0) return;
  const int nk_all = p.K >> 6;

  auto load_item = [&](Cursor& c) {
    c.tile = cs + idx + c.j * bpx;
    c.split = 0;
    int tm, tn;
    tile_mn(c.tile, p.tiles_n, p.ntiles, p.group_n, tm, tn);
    c.m0 = tm * 256; c.n0 = tn * 256;
    c.nk = nk_all;
    c.offA = (long)c.m0 * p.lda;
    c.offB = (long)c.n0 * p.ldb;
  };
  auto advance = [&](Cursor& c) {
    if (++c.t == c.nk) {
      c.t = 0;
      ++c.j;
      if (c.j < nmy) load_item(c);
    }
  };

  // ---- per-lane global source pointers of the DMA
  const bf16* srcA;
  const bf16* srcB;
  {
    const int r = wave * 8 + (lane >> 3);
    const int pos = lane & 7;
    const int cc = pos ^ kswz(r);
    srcA = p.A + (long)r * p.lda + cc * 8;
    srcB = p.B + (long)r * p.ldb + cc * 8;
  }
  const long gA = 64 * p.lda, gB = 64 * p.ldb, hA = 128 * p.lda, hB = 128 * p.ldb;
  char* const ldsA = smem;
  char* const ldsB = smem + A_BYTES;
  const int wave_off = wave * 1024;
  auto issueA = [&](const Cursor& c, int slot, int h) {
    char* d = ldsA + (slot * 2 + h) * HALF + wave_off;
    const bf16* s = srcA + c.offA + (long)c.t * 64 + (h ? hA : 0);
    glds16<BV_GLDS_AUX_A>(s, d);
    glds16<BV_GLDS_AUX_A>(s + gA, d + 8192);
  };
  auto issueB = [&](const Cursor& c, int slot, int h) {
    char* d = ldsB + (slot * 2 + h) * HALF + wave_off;
    const bf16* s = srcB + c.offB + (long)c.t * 64 + (h ? hB : 0);
    glds16<BV_GLDS_AUX_B>(s, d);
    glds16<BV_GLDS_AUX_B>(s + gB, d + 8192);
  };

  // ---- per-lane LDS read addresses (relative to the stage base), as gemm256_kernel<true>
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int brow = OUTF32 ? lr : (lr >> 2) * 8 + (lr & 3);
  const uint32_t ra0 = lds0 + wr * HALF + lr * 128 + ((lg ^ kswz(lr)) << 4);
  const uint32_t ra1 = ra0 ^ 64;
  const uint32_t rb0 = lds0 + A_BYTES + (wc >> 1) * HALF + ((wc & 1) * 64 + brow) * 128 + ((lg ^ kswz(brow)) << 4);
  const uint32_t rb1 = rb0 ^ 64;

  f32x4 acc[8][4];
  bf16x8 af[4][2], bfg[4][2];
  f32x4 bq[4];   // bias of the tile whose epilogue is running: bq[j][r] = bias[col(j) + r]
#pragma unroll
  for (int j = 0; j < 4; ++j) bq[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto readA = [&](uint32_t sa, int sub) {
    const uint32_t a0 = ra0 + sa, a1 = ra1 + sa;
    if (sub == 0) {
      af[0][0] = lds_read128<0>(a0);     af[0][1] = lds_read128<0>(a1);
      af[1][0] = lds_read128<2048>(a1);  af[1][1] = lds_read128<2048>(a0);
      af[2][0] = lds_read128<4096>(a0);  af[2][1] = lds_read128<4096>(a1);
      af[3][0] = lds_read128<6144>(a1);  af[3][1] = lds_read128<6144>(a0);
    } else {
      af[0][0] = lds_read128<8192>(a0);  af[0][1] = lds_read128<8192>(a1);
      af[1][0] = lds_read128<10240>(a1); af[1][1] = lds_read128<10240>(a0);
      af[2][0] = lds_read128<12288>(a0); af[2][1] = lds_read128<12288>(a1);
      af[3][0] = lds_read128<14336>(a1); af[3][1] = lds_read128<14336>(a0);
    }
  };
  auto readB = [&](uint32_t sb, int sub) {
    const uint32_t b0 = rb0 + sb, b1 = rb1 + sb;
    constexpr int O1 = OUTF32 ? 2048 : 512, O2 = 4096, O3 = OUTF32 ? 6144 : 4608;
    if (sub == 0) {
      bfg[0][0] = lds_read128<0>(b0);    bfg[0][1] = lds_read128<0>(b1);
      bfg[1][0] = lds_read128<O1>(b1);   bfg[1][1] = lds_read128<O1>(b0);
    } else {
      bfg[2][0] = lds_read128<O2>(b0);   bfg[2][1] = lds_read128<O2>(b1);
      bfg[3][0] = lds_read128<O3>(b1);   bfg[3][1] = lds_read128<O3>(b0);
    }
  };

  // Epilogue addressing: (wave-uniform 64-bit base, SALU) + (per-lane 32-bit byte offset, loop
  // invariant).  Row fragment i of the wave sits at rows m0 + wr*128 + i*16 + lr; column fragment j
  // at element n0 + wc*64 + {fp32 out: j*16 + lg*4 | bf16 out: (j>>1)*32 + lg*8 + (j&1)*4}
  // (see the epilogue comment of gemm256_kernel).
  constexpr int ESZ = OUTF32 ? 4 : 2;
  const uint32_t lane_c = (uint32_t)lr * (uint32_t)p.ldc * ESZ + (OUTF32 ? lg * 16 : lg * 16);
  const uint32_t lane_x = (uint32_t)lr * (uint32_t)p.ldaux * 4 + lg * 16;
  const uint32_t lane_b = OUTF32 ? lg * 16 : lg * 32;
  auto ucol = [&](int n0, int j) {   // wave-uniform part of fragment j's column
    return n0 + wc * 64 + (OUTF32 ? j * 16 : (j >> 1) * 32 + (j & 1) * 4);
  };
  // issue the NX residual loads of quadrant (IH, JH) of the tile at (xm0, xn0) into its accumulators
  auto xload = [&](auto IHc, auto JHc, int xm0, int xn0) {
    constexpr int IH = decltype(IHc)::value, JH = decltype(JHc)::value;
    if constexpr (NX > 0) {
      const float* x0 = reinterpret_cast<const float*>(p.aux) + (long)(xm0 + wr * 128 + IH * 64) * p.ldaux;
#pragma unroll
      for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          gl_load16s(acc[IH * 4 + ii][JH * 2 + jj], x0 + (long)ii * 16 * p.ldaux + ucol(xn0, JH * 2 + jj), lane_x);
    }
  };
  // issue the 4 bias loads of the tile with columns n0.. (no bias: any valid address, zeroed later)
  auto bload = [&](int n0) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      gl_load16s(bq[j], p.bias ? (const void*)(p.bias + ucol(n0, j)) : (const void*)p.A, p.bias ? lane_b : 0u);
  };
  // epilogue of quadrant (IH, JH) of the tile at (m0, n0): NS stores per lane
  auto chunk = [&](auto IHc, auto JHc, int m0, int n0) {
    constexpr int IH = decltype(IHc)::value, JH = decltype(JHc)::value;
    const long mrow = m0 + wr * 128 + IH * 64;
    if constexpr (OUTF32) {
      char* c0 = reinterpret_cast<char*>(reinterpret_cast<float*>(p.C) + mrow * p.ldc);
#pragma unroll
      for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int i = IH * 4 + ii, j = JH * 2 + jj;
          const f32x4 v = acc[i][j] + bq[j];       // alpha = 1; the residual is already in acc
          char* cb = c0 + ((long)ii * 16 * p.ldc + ucol(n0, j)) * 4;
          *reinterpret_cast<f32x4*>(cb + lane_c) = v;
        }
    } else {
      char* c0 = reinterpret_cast<char*>(reinterpret_cast<bf16*>(p.C) + mrow * p.ldc + ucol(n0, JH * 2));
      char* g0 = reinterpret_cast<char*>(reinterpret_cast<bf16*>(p.C2) + mrow * p.ldc + ucol(n0, JH * 2));
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int i = IH * 4 + ii;
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[jj * 4 + r] = acc[i][JH * 2 + jj][r] * p.alpha + bq[JH * 2 + jj][r];
        uint32_t hw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) hw[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
        *reinterpret_cast<u32x4*>(c0 + (long)ii * 32 * p.ldc + lane_c) = u32x4{hw[0], hw[1], hw[2], hw[3]};
        if constexpr (EPI == BV_EPI_GELU) {
          uint32_t gw[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) gw[e] = pack_bf2(gelu_tanh_f(bflo(hw[e])), gelu_tanh_f(bfhi(hw[e])));
          *reinterpret_cast<u32x4*>(g0 + (long)ii * 32 * p.ldc + lane_c) = u32x4{gw[0], gw[1], gw[2], gw[3]};
        }
      }
    }
  };

  // STM: the same epilogue split in two - VALU part (load segment) and the stores (MFMA segment)
  u32x4 pk[4], pk2[4];
  char* st_c = nullptr;
  char* st_g = nullptr;
  auto chunk_prep = [&](auto IHc, auto JHc, int m0, int n0) {
    constexpr int IH = decltype(IHc)::value, JH = decltype(JHc)::value;
    if constexpr (!OUTF32) {
      const long mrow = m0 + wr * 128 + IH * 64;
      st_c = reinterpret_cast<char*>(reinterpret_cast<bf16*>(p.C) + mrow * p.ldc + ucol(n0, JH * 2));
      st_g = reinterpret_cast<char*>(reinterpret_cast<bf16*>(p.C2) + mrow * p.ldc + ucol(n0, JH * 2));
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const int i = IH * 4 + ii;
        float v[8];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[jj * 4 + r] = acc[i][JH * 2 + jj][r] * p.alpha + bq[JH * 2 + jj][r];
        uint32_t hw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) hw[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
        pk[ii] = u32x4{hw[0], hw[1], hw[2], hw[3]};
        if constexpr (EPI == BV_EPI_GELU) {
          uint32_t gw[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) gw[e] = pack_bf2(gelu_tanh_f(bflo(hw[e])), gelu_tanh_f(bfhi(hw[e])));
          pk2[ii] = u32x4{gw[0], gw[1], gw[2], gw[3]};
        }
      }
    }
  };
  auto chunk_store = [&]() {
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      *reinterpret_cast<u32x4*>(st_c + (long)ii * 32 * p.ldc + lane_c) = pk[ii];
      if constexpr (EPI == BV_EPI_GELU) *reinterpret_cast<u32x4*>(st_g + (long)ii * 32 * p.ldc + lane_c) = pk2[ii];
    }
  };

// 16 MFMAs with NS stores spread between them: [16/NS MFMAs][1 store] x NS
#define BVR_MFMA_QUAD_ST(I0, J0, ZERO)                                                         \
  do {                                                                                         \
    __builtin_amdgcn_s_setprio(1);                                                             \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                           \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                            \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                          \
          acc[(I0) + i][(J0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                   \
              bfg[(J0) + j][ks], af[i][ks],                                                    \
              ((ZERO) && ks == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[(I0) + i][(J0) + j], 0, 0, 0); \
    chunk_store();                                                                             \
    _Pragma("unroll") for (int g_ = 0; g_ < NS; ++g_) {                                        \
      __builtin_amdgcn_sched_group_barrier(0x008, 16 / NS, 0);                                 \
      __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);                                       \
    }                                                                                          \
    __builtin_amdgcn_s_setprio(0);                                                             \
  } while (0)
#define BVR_MFMA_QUAD(I0, J0, ZERO)                                                            \
  do {                                                                                         \
    __builtin_amdgcn_s_setprio(1);                                                             \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                           \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                            \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                          \
          acc[(I0) + i][(J0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                   \
              bfg[(J0) + j][ks], af[i][ks],                                                    \
              ((ZERO) && ks == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[(I0) + i][(J0) + j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                             \
  } while (0)
#define BVR_MID()                                         \
  do {                                                    \
    __builtin_amdgcn_sched_barrier(0);                    \
    __builtin_amdgcn_s_barrier();                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);                    \
  } while (0)
#define BVR_END()                            \
  do {                                       \
    __builtin_amdgcn_sched_barrier(0);       \
    __builtin_amdgcn_s_barrier();            \
  } while (0)
#define BVR_PIN() __builtin_amdgcn_sched_barrier(0)

  // ---- prologue: B(0), A(0), B(1) in flight; K-tile 0 must have landed.
  Cursor cur{};
  cur.j = 0; cur.t = 0;
  load_item(cur);
  Cursor ca = cur, cb = cur;   // A stream runs 1 K-tile ahead, B stream 2 K-tiles ahead
  issueB(cb, 0, 0); issueB(cb, 0, 1);
  issueA(ca, 0, 0); issueA(ca, 0, 1);
  advance(ca);
  advance(cb);
  issueB(cb, 1, 0); issueB(cb, 1, 1);     // nk >= 2: K-tile 1 of the first tile always exists
  advance(cb);
  BVR_PIN();
  if constexpr (NX > 0) {
    // residual of the first tile into all four quadrants; drained here, once
    xload(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, cur.m0, cur.n0);
    xload(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, cur.m0, cur.n0);
    xload(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, cur.m0, cur.n0);
    xload(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, cur.m0, cur.n0);
    vm_wait<0>();
#pragma unroll
    for (int i = 0; i < 8; ++i) vm_tie4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
  } else {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  BVR_PIN();
  __builtin_amdgcn_s_barrier();

  int gk = 0;  // K-tiles consumed so far (ring position)
  int bs = 0;  // B ring slot of the current K-tile
  int em0 = 0, en0 = 0;     // tile whose epilogue is rolling
  int xm0 = 0, xn0 = 0;     // tile after it (target of the residual loads issued during its last K-tile)
  bool have_prev = false, have_next = false;
  using T_ = std::integral_constant<bool, true>;
  using F_ = std::integral_constant<bool, false>;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  auto ktile = [&](auto FIRSTc, auto LASTc) {
    constexpr bool FIRST = decltype(FIRSTc)::value, LAST = decltype(LASTc)::value;
    constexpr bool ZERO = FIRST && NX == 0;
    const uint32_t sa = (gk & 1) * 2 * HALF;
    const uint32_t sb = bs * 2 * HALF;
    const int bs1 = (bs == 2) ? 0 : bs + 1;          // slot of K-tile gk+1
    const int bs2 = (bs1 == 2) ? 0 : bs1 + 1;        // slot of K-tile gk+2
    const bool moreA = ca.j < nmy, moreB = cb.j < nmy;
    const bool full = moreA && moreB;
    // -------- phase 0: quadrant (0,0)
    readA(sa, 0);
    readB(sb, 0);
    if (moreA) issueA(ca, (gk + 1) & 1, 0);
    BVR_PIN();
    if constexpr (FIRST) {
      if (have_prev) {
        chunk(I1{}, I0{}, em0, en0);
        BVR_PIN();
        xload(I1{}, I0{}, cur.m0, cur.n0);
      }
      if constexpr (NX > 0) {
        if (full) vm_wait<W_X00>();
        else vm_wait<0>();
        vm_tie8(acc[0][0], acc[0][1], acc[1][0], acc[1][1], acc[2][0], acc[2][1], acc[3][0], acc[3][1]);
      }
    }
    BVR_MID();
    BVR_MFMA_QUAD(0, 0, ZERO);
    BVR_END();
    // -------- phase 1: quadrant (0,1)
    readB(sb, 1);
    if (moreA) issueA(ca, (gk + 1) & 1, 1);
    BVR_PIN();
    if constexpr (LAST) {
      if constexpr (STM) chunk_prep(I0{}, I0{}, em0, en0); else chunk(I0{}, I0{}, em0, en0);
      BVR_PIN();
      if (have_next) xload(I0{}, I0{}, xm0, xn0);
    }
    if constexpr (FIRST) {
      // bias of this tile (always 4 loads, so that the counts are static; no bias: any valid address)
      bload(cur.n0);
      if constexpr (NX > 0) {
        if (full) vm_wait<W_X01>();
        else vm_wait<0>();
        vm_tie8(acc[0][2], acc[0][3], acc[1][2], acc[1][3], acc[2][2], acc[2][3], acc[3][2], acc[3][3]);
      }
    }
    BVR_MID();
    if constexpr (LAST && STM) BVR_MFMA_QUAD_ST(0, 2, ZERO); else BVR_MFMA_QUAD(0, 2, ZERO);
    BVR_END();
    // -------- phase 2: quadrant (1,1)
    readA(sa, 1);
    if (moreB) issueB(cb, bs2, 0);
    BVR_PIN();
    if constexpr (LAST) {
      if constexpr (STM) chunk_prep(I0{}, I1{}, em0, en0); else chunk(I0{}, I1{}, em0, en0);
      BVR_PIN();
      if (have_next) xload(I0{}, I1{}, xm0, xn0);
    }
    if constexpr (FIRST && NX > 0) {
      if (full) vm_wait<W_X11>();
      else vm_wait<0>();
      vm_tie8(acc[4][2], acc[4][3], acc[5][2], acc[5][3], acc[6][2], acc[6][3], acc[7][2], acc[7][3]);
    }
    BVR_MID();
    if constexpr (LAST && STM) BVR_MFMA_QUAD_ST(4, 2, ZERO); else BVR_MFMA_QUAD(4, 2, ZERO);
    BVR_END();
    // -------- phase 3: quadrant (1,0); retire K-tile gk+1's loads for the next iteration
    if (moreB) issueB(cb, bs2, 1);
    BVR_PIN();
    if constexpr (LAST) {
      if (full) vm_wait<W_A>();
      else vm_wait<0>();
      BVR_PIN();
      if constexpr (STM) chunk_prep(I1{}, I1{}, em0, en0); else chunk(I1{}, I1{}, em0, en0);
      BVR_PIN();
      if (have_next) xload(I1{}, I1{}, xm0, xn0);
    } else if constexpr (FIRST) {
      // also retires this tile's bias loads and the residual loads of quadrant (1,0)
      if (moreB) vm_wait<4>();
      else vm_wait<0>();
      vm_tie4(bq[0], bq[1], bq[2], bq[3]);
      if constexpr (NX > 0) vm_tie8(acc[4][0], acc[4][1], acc[5][0], acc[5][1], acc[6][0], acc[6][1], acc[7][0], acc[7][1]);
      if (!p.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    } else {
      if (moreB) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    BVR_MID();
    if constexpr (LAST && STM) BVR_MFMA_QUAD_ST(4, 0, ZERO); else BVR_MFMA_QUAD(4, 0, ZERO);
    BVR_END();
    if (moreA) advance(ca);
    if (moreB) advance(cb);
    bs = bs1;
    ++gk;
  };

  if (wr == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier behind, for the whole launch
  for (int jt = 0; jt < nmy; ++jt) {
    const int nkc = cur.nk;
    ktile(T_{}, F_{});
    for (int t = 1; t < nkc - 1; ++t) ktile(F_{}, F_{});
    // the last K-tile rolls this tile's epilogue; the residual loads it issues belong to the next tile
    em0 = cur.m0; en0 = cur.n0;
    have_next = jt + 1 < nmy;
    cur.t = cur.nk - 1;
    advance(cur);
    xm0 = cur.m0; xn0 = cur.n0;
    ktile(F_{}, T_{});
    have_prev = true;
  }
  BVR_PIN();
  chunk(I1{}, I0{}, em0, en0);   // quadrant (1,0) of the last tile
  if (wr == 0) __builtin_amdgcn_s_barrier();
#undef BVR_MFMA_QUAD
#undef BVR_MFMA_QUAD_ST
#undef BVR_MID
#undef BVR_END
#undef BVR_PIN
}

// C[m][n..n+3] += alpha * sum_s slab[s][tile][...]  (deterministic split-K combine).
// One thread per float4 of a tile; slab layout [wave][i][j][lane][4] as written above.
__global__ __launch_bounds__(256) void gemm256_reduce_kernel(const float* __restrict__ slab,
                                                             float* __restrict__ C, long ldc,
                                                             int ntiles, int tiles_n, int splits,
                                                             float alpha, int accumulate) {
  const int tile = blockIdx.x >> 6;                      // 64 blocks of 256 float4 per tile
  const int q = ((blockIdx.x & 63) << 8) + threadIdx.x;  // float4 index inside the tile, 0..16383
  const int lane = q & 63, ij = (q >> 6) & 31, wave = q >> 11;
  const int i = ij >> 2, j = ij & 3, wr = wave >> 2, wc = wave & 3, lr = lane & 15, lg = lane >> 4;
  const float* s = slab + (long)tile * 65536 + (long)q * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < splits; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(s + (long)k * ntiles * 65536);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m = tm * 256 + wr * 128 + i * 16 + lr;
  const int n = tn * 256 + wc * 64 + j * 16 + lg * 4;
  float4* c = reinterpret_cast<float4*>(C + (long)m * ldc + n);
  float4 o = make_float4(a.x * alpha, a.y * alpha, a.z * alpha, a.w * alpha);
  if (accumulate) {
    const float4 old = *c;
    o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
  }
  *c = o;
}

}  // namespace

// Bytes of split-K scratch a dW GEMM (a_kmajor = b_kmajor = 0, EPI_ATOMIC) of this shape uses
// with the automatic split choice; 0 = the shape takes no workspace.
extern "C" long bv_gemm_workspace_bytes(int M, int N, int K) {
  if ((M & 255) || (N & 255) || (K & 63)) return 0;
  const int ntiles = (M >> 8) * (N >> 8), nk = K >> 6;
  int splits = 256 / (ntiles > 0 ? ntiles : 1);
  const int max_splits = nk / 8 > 0 ? nk / 8 : 1;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  const int per = (nk + splits - 1) / splits;
  splits = (nk + per - 1) / per;
  return splits > 1 ? (long)ntiles * splits * 65536 * 4 : 0;
}

// Internal entry used by bv_gemm_bf16 (gemm_bf16.hip).  Returns 1 if the problem
// was launched on the 256x256 path, 0 if it does not qualify (caller falls back).
int bv_gemm256_try(int a_kmajor, int b_kmajor, const void* A, long lda, const void* B, long ldb,
                   void* C, long ldc, int out_f32, int M, int N, int K, int epilogue,
                   const float* bias, const void* aux, long ldaux, int aux_rows, void* C2,
                   float alpha, int split_k, float* colsum, void* stream, const bv_ctx* ctx_) {
  // every option, the split-K workspace and the launch counters come from the caller's context (NULL: defaults)
  const bv_ctx* ctx = bv_ctx_or_default(ctx_);
  const int g_reserve = (int)ctx->opt[BV_OPT_GEMM_RESERVE_CUS], g_roll = (int)ctx->opt[BV_OPT_GEMM_ROLL];
  const int g_skew_pct = (int)ctx->opt[BV_OPT_GEMM_SKEW_PCT], g_skew_mode = (int)ctx->opt[BV_OPT_GEMM_SKEW_MODE];
  if (a_kmajor != b_kmajor) return 0;
  if ((M & 255) || (N & 255) || (K & 63)) return 0;
  const bool km = a_kmajor != 0;
  if (km && epilogue == BV_EPI_ATOMIC) return 0;
  if (!km && !(epilogue == BV_EPI_ATOMIC || (epilogue == BV_EPI_NONE && out_f32 && !bias))) return 0;
  if ((lda & 7) || (ldb & 7) || (ldc & 7) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) ||
      ((uintptr_t)C & 15))
    return 0;
  if (aux && ((ldaux & 7) || ((uintptr_t)aux & 15))) return 0;
  if (bias && ((uintptr_t)bias & 15)) return 0;
  if (C2 && ((uintptr_t)C2 & 15)) return 0;

  G256Params p;
  p.A = (const bf16*)A; p.B = (const bf16*)B; p.C = C; p.C2 = C2;
  p.bias = bias; p.aux = aux; p.colsum = colsum;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldaux = ldaux;
  p.M = M; p.N = N; p.K = K; p.aux_rows = aux_rows > 0 ? aux_rows : 1;
  p.epi = epilogue; p.out_f32 = out_f32; p.alpha = alpha;
  const int tiles_m = M >> 8;
  p.tiles_n = N >> 8;
  p.ntiles = tiles_m * p.tiles_n;
  const int nk = K >> 6;
  int splits = 1;
  p.slab = nullptr;
  if (epilogue == BV_EPI_ATOMIC) {
    // split-K so that tiles x splits ~ one workgroup per CU (256): the K loop is the
    // whole cost, every extra split adds a 256 KiB partial tile of output traffic.
    if (split_k > 0) {
      splits = split_k;
    } else {
      splits = (256 - g_reserve) / p.ntiles;   // work items <= CUs in use (4 reserved CUs keep B/16's choices: 252 = 36 x 7 = 9 x 28)
      const int max_splits = nk / 8 > 0 ? nk / 8 : 1;
      if (splits > max_splits) splits = max_splits;
    }
    if (splits < 1) splits = 1;
    if (splits > nk) splits = nk;
  }
  p.ktiles_per_split = (nk + splits - 1) / splits;
  splits = (nk + p.ktiles_per_split - 1) / p.ktiles_per_split;
  const long slab_bytes = (long)p.ntiles * splits * 65536 * 4;
  void* const ws_ptr = epilogue == BV_EPI_ATOMIC ? ctx->ws : nullptr;
  const bool use_slab = epilogue == BV_EPI_ATOMIC && splits > 1 && ws_ptr && slab_bytes <= ctx->ws_bytes &&
                        (ldc & 3) == 0;
  if (use_slab) p.slab = (float*)ws_ptr;
  p.splits = splits;
  const int nwork = p.ntiles * splits;
  // start skew (see kernel): a percentage of one tile period (~3600 cycles per K-tile + epilogue)
  p.skew_cycles = 0;
  p.skew_mode = 0;
  if (km && g_skew_pct > 0 && nwork > 256) {
    p.skew_cycles = (int)((long)(nk * 3600 + 12000) * g_skew_pct / 100);
    p.skew_mode = g_skew_mode;
  }
  p.nt = (int)ctx->opt[BV_OPT_GEMM_NT];
  p.group_n = km ? (int)ctx->opt[BV_OPT_GEMM_GROUP_N] : 0;
  p.pre_issue = (int)ctx->opt[BV_OPT_GEMM_PRE_ISSUE];
  p.dbg = nullptr;
  const int cus = 256 - g_reserve;
  dim3 grid(nwork < cus ? nwork : cus), block(512);   // persistent: one workgroup per CU
  hipStream_t s = (hipStream_t)stream;
  ctx->calls[0].fetch_add(1, std::memory_order_relaxed);
  if (nwork > (int)grid.x) {
    ctx->calls[1].fetch_add(1, std::memory_order_relaxed);
    if ((epilogue != BV_EPI_NONE && epilogue != BV_EPI_ATOMIC) || colsum) ctx->calls[2].fetch_add(1, std::memory_order_relaxed);
  }
  // rolling-epilogue kernel: k-major, at least two K-tiles per tile, the epilogues it implements
  const bool roll = km && nk >= 2 && !colsum &&
                    (((g_roll & 2) && epilogue == BV_EPI_NONE && !out_f32) ||
                     ((g_roll & 4) && epilogue == BV_EPI_GELU && !out_f32) ||
                     ((g_roll & 1) && epilogue == BV_EPI_RESIDUAL && out_f32 && alpha == 1.0f));
  if (roll) {
    if (epilogue == BV_EPI_RESIDUAL) hipLaunchKernelGGL((gemm256r_kernel<BV_EPI_RESIDUAL, true>), grid, block, 0, s, p);
    else if (epilogue == BV_EPI_GELU && (g_roll & 8)) hipLaunchKernelGGL((gemm256r_kernel<BV_EPI_GELU, false, 1>), grid, block, 0, s, p);
    else if (epilogue == BV_EPI_GELU) hipLaunchKernelGGL((gemm256r_kernel<BV_EPI_GELU, false>), grid, block, 0, s, p);
    else if (g_roll & 8) hipLaunchKernelGGL((gemm256r_kernel<BV_EPI_NONE, false, 1>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm256r_kernel<BV_EPI_NONE, false>), grid, block, 0, s, p);
    return 1;
  }
  if (!km) hipLaunchKernelGGL((gemm256_kernel<false>), grid, block, 0, s, p);
  else if (epilogue == BV_EPI_RESIDUAL && !out_f32) hipLaunchKernelGGL((gemm256_kernel<true, 0, BV_EPI_RESIDUAL, false>), grid, block, 0, s, p);
  else if (epilogue == BV_EPI_RESIDUAL) hipLaunchKernelGGL((gemm256_kernel<true, 0, BV_EPI_RESIDUAL, true>), grid, block, 0, s, p);
  else if (epilogue == BV_EPI_POS) hipLaunchKernelGGL((gemm256_kernel<true, 0, BV_EPI_POS, true>), grid, block, 0, s, p);
  else if (epilogue == BV_EPI_GELU) hipLaunchKernelGGL((gemm256_kernel<true, 0, BV_EPI_GELU, false>), grid, block, 0, s, p);
  else if (epilogue == BV_EPI_GELU_G) hipLaunchKernelGGL((gemm256_kernel<true, 0, BV_EPI_GELU_G, false>), grid, block, 0, s, p);
  else if (epilogue == BV_EPI_GELU_BWD) hipLaunchKernelGGL((gemm256_kernel<true, 0, BV_EPI_GELU_BWD, false>), grid, block, 0, s, p);
  else if (epilogue == BV_EPI_GELU_BWD_EMIT) hipLaunchKernelGGL((gemm256_kernel<true, 0, BV_EPI_GELU_BWD_EMIT, false>), grid, block, 0, s, p);
  else if (epilogue == BV_EPI_GELU_GD) hipLaunchKernelGGL((gemm256_kernel<true, 0, BV_EPI_GELU_GD, false>), grid, block, 0, s, p);
  else if (epilogue == BV_EPI_MUL) hipLaunchKernelGGL((gemm256_kernel<true, 0, BV_EPI_MUL, false>), grid, block, 0, s, p);
  else if (out_f32) hipLaunchKernelGGL((gemm256_kernel<true, 0, BV_EPI_NONE, true>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((gemm256_kernel<true, 0, BV_EPI_NONE, false>), grid, block, 0, s, p);
  if (use_slab)
    hipLaunchKernelGGL(gemm256_reduce_kernel, dim3(p.ntiles * 64), dim3(256), 0, s, (const float*)ws_ptr,
                       (float*)C, ldc, p.ntiles, p.tiles_n, splits, alpha, 1);
  return 1;
}
