// Shared device helpers of the LDS-resident attention kernels (attention3.hip, attention5.hip; attention2.hip of rounds 2-4 left the library in round 5):
// the "T64" tile image, fragment loads and the small cross-lane reductions.  gfx950 only.
#pragma once
#include "bv_common.h"

namespace bvattn {

constexpr int DH = 64;
constexpr float LOG2E = 1.4426950408889634f;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// LDS tile layout ("T64"): row r = 8 chunks of 16 B, chunk c stored at position
// c ^ (((r >> 1) & 3) << 1).  The same image serves row-operand reads (ds_read_b128: lane = row,
// 8 consecutive d) and transposed reads (ds_read_b64_tr_b16: contraction over rows, lane = d),
// both bank-conflict free (brute-forced over the gfx950 lane groups).
__device__ __forceinline__ int t64_swz(int row) { return ((row >> 1) & 3) << 1; }

// Stage two [L][64] matrices (row strides ld0/ld1 elements) into T64 tiles of ROWS rows (rows >= L
// are zero) with NT threads.  All global loads of a batch are issued before the first LDS store,
// so a tile costs one memory round trip per batch, not one per piece.
template <int ROWS, int NT = 256>
__device__ __forceinline__ void t64_stage2(char* T0, const bf16* src0, long ld0, char* T1,
                                           const bf16* src1, long ld1, int L, int tid) {
  constexpr int N = (ROWS * 8 + NT - 1) / NT;     // 16-byte pieces per thread and tile
  constexpr int B = N < 8 ? N : 8;
  static_assert(ROWS % 16 == 0, "tile rows must be a multiple of 16");
#pragma unroll
  for (int b0 = 0; b0 < N; b0 += B) {
    uint4 v0[B], v1[B];
#pragma unroll
    for (int j = 0; j < B; ++j) {
      const int idx = tid + (b0 + j) * NT;
      const int row = idx >> 3, pc = idx & 7;
      v0[j] = make_uint4(0, 0, 0, 0);
      v1[j] = make_uint4(0, 0, 0, 0);
      if (b0 + j < N && row < L) {
        v0[j] = *reinterpret_cast<const uint4*>(src0 + (long)row * ld0 + pc * 8);
        v1[j] = *reinterpret_cast<const uint4*>(src1 + (long)row * ld1 + pc * 8);
      }
    }
#pragma unroll
    for (int j = 0; j < B; ++j) {
      const int idx = tid + (b0 + j) * NT;
      const int row = idx >> 3, pc = idx & 7;
      if (b0 + j < N && row < ROWS) {
        const int off = row * 128 + ((pc ^ t64_swz(row)) << 4);
        *reinterpret_cast<uint4*>(T0 + off) = v0[j];
        *reinterpret_cast<uint4*>(T1 + off) = v1[j];
      }
    }
  }
}
// Row operand: 8 consecutive d (chunk) of row `row`.
__device__ __forceinline__ bf16x8 t64_row(const char* T, int row, int chunk) {
  const uint4 v = *reinterpret_cast<const uint4*>(T + row * 128 + ((chunk ^ t64_swz(row)) << 4));
  return __builtin_bit_cast(bf16x8, v);
}
// Transposed operand: lane (d = db*16 + lr) receives rows (row4 + 0..3) of column d; row4 already
// includes the lane group's 4*lg.
__device__ __forceinline__ s16x4 t64_tr(const char* T, int row4, int db, int lr) {
  const int row = row4 + (lr >> 2);
  const char* p = T + row * 128 + (((db * 2 + ((lr >> 1) & 1)) ^ t64_swz(row)) << 4) + ((lr & 1) << 3);
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}
__device__ __forceinline__ bf16x8 t64_trpair(const char* T, int ra, int rb, int db, int lr) {
  const s16x4 a = t64_tr(T, ra, db, lr), b = t64_tr(T, rb, db, lr);
  const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 gfrag(const bf16* base, long ld, int row, int L, int col) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (row < L) v = *reinterpret_cast<const uint4*>(base + (long)row * ld + col);
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  return __builtin_bit_cast(bf16x8, make_uint4(pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]),
                                               pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3])));
}
__device__ __forceinline__ s16x4 pack4(const f32x4& a) {
  return __builtin_bit_cast(s16x4, make_uint2(pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3])));
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// K = 16 form (lane l supplies A[l&15][4*(l>>4)..+3], B[4*(l>>4)..+3][l&15]): the odd last
// fragment of a contraction over 16-row fragments
__device__ __forceinline__ f32x4 mfma16k16(s16x4 a, s16x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
// Store of a TRANSPOSED 64 x 16 result (oa[d][r] = element (column d * 16 + 4 * lg + r) of row `lr`, the layout the
// O^T / dQ^T accumulators have) as bf16 into 16 rows of 64 contiguous values, `row` = this lane's row pointer.
// Straight from the accumulators a lane owns four 8-byte pieces of its row, and a wave store instruction writes 16
// rows x 32 bytes: twice the instructions and half-empty write requests.  One v_permlane16_swap per dword first
// pairs up the neighbouring pieces of two lane groups (a' = {a.row0, b.row0, a.row2, b.row2}, b' = {a.row1, b.row1,
// a.row3, b.row3}; inline asm, see attention5.hip on the builtin), so a lane stores two 16-byte pieces and an
// instruction covers 16 rows x 64 bytes.  ALL 64 lanes must call it (the swaps); `valid` masks the stores.
// `live` = false writes zeros instead of the row (a per-ROW flag: the lanes that swap share their row).
__device__ __forceinline__ void store_ot_rows(bf16* row, const f32x4 (&oa)[4], float mul, int lg, bool valid,
                                              bool live = true) {
  typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int dp = 0; dp < 4; dp += 2) {    // one pair of 16-column blocks at a time: four packed registers live
    unsigned int a0 = pack_bf2(oa[dp][0] * mul, oa[dp][1] * mul), a1 = pack_bf2(oa[dp][2] * mul, oa[dp][3] * mul);
    unsigned int b0 = pack_bf2(oa[dp + 1][0] * mul, oa[dp + 1][1] * mul), b1 = pack_bf2(oa[dp + 1][2] * mul, oa[dp + 1][3] * mul);
    if (!live) a0 = a1 = b0 = b1 = 0u;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a0), "+v"(b0));
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a1), "+v"(b1));
    if (valid) *reinterpret_cast<u32x4_*>(row + (dp + (lg & 1)) * 16 + (lg >> 1) * 8) = u32x4_{a0, a1, b0, b1};
  }
}

__device__ __forceinline__ float xmax4(float v) {   // max over the 4 lane groups (same lr)
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xsum4(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}
// sum over the 16 lanes of a DPP row (lanes with the same lane>>4); every lane gets the total
__device__ __forceinline__ float rowsum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

}  // namespace bvattn
