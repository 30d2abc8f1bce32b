// bf16 MFMA GEMM for gfx950 (MI355X): C[M,N] = A(M x K) * B(K x N), fp32 accumulate.
//
// Replaces the XLA-lowered lax.dot_general calls behind flax nn.Dense /
// nn.DenseGeneral / nn.Conv(stem) on the hot path
// (reference big_vision/models/vit.py:72,77,93-98,212-214,261,272;
//  models/proj/image_text/text_transformer.py:98) and their transposes in the
// backward pass (jax.value_and_grad, trainers/proj/image_text/siglip.py:311).
//
// Tile: 128x128x64 per 256-thread workgroup (4 waves as 2x2, each 64x64 =
// 4x4 fragments of v_mfma_f32_16x16x32_bf16), LDS double-buffered (64 KiB, two
// workgroups per CU), register-staged global->LDS so that either operand may
// be "K-major" (reduction dim contiguous: 16-byte loads along K) or
// "K-minor" (reduction dim is the slow axis: the tile is staged in its natural
// [k][row] layout and MFMA fragments are gathered with the LDS transpose read
// ds_read_b64_tr_b16 — no register transposes).  K-major LDS image:
//   tile[row][64 k] bf16, 128 B per row, 16-byte chunk c stored at
//   c ^ swz(row), swz(row) = (row ^ (row >> 3)) & 7.
// The MFMA is issued "swapped" (first operand = B rows, second = A rows) so a
// lane ends up with 4 consecutive output columns of one output row, giving 8/16
// byte epilogue accesses.
#include "bv_common.h"
#include "bvhip_internal.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;  // 16 KiB per operand tile

struct GemmParams {
  const bf16* A;
  const bf16* B;
  void* C;
  void* C2;
  const float* bias;
  const void* aux;
  float* colsum;       // optional: colsum[n] += sum_m C[m][n] (fp32, before the output rounding)
  long lda, ldb, ldc, ldaux;
  int M, N, K;
  int aux_rows;
  int k_chunk;
  int epi;
  int out_f32;
  float alpha;
};

__device__ __forceinline__ int swz(int row) { return (row ^ (row >> 3)) & 7; }

// ---- global -> registers ---------------------------------------------------
// K-major operand: memory P[row][k], k contiguous.  Thread t owns chunk (t&7)
// of rows (t>>3) + 32*i.
__device__ __forceinline__ void gload_kmajor(uint4 (&r)[4], const bf16* P, long ld, int R,
                                             int row0, int k0, int kend, int tid) {
  const int c = tid & 7;
  const int gk = k0 + c * 8;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int grow = row0 + (tid >> 3) + 32 * i;
    if (grow < R && gk < kend) {
      r[i] = *reinterpret_cast<const uint4*>(P + (long)grow * ld + gk);
    } else {
      r[i] = make_uint4(0, 0, 0, 0);
    }
  }
}
// K-minor operand: memory P[k][row], row contiguous.  The tile is kept in its
// natural [64 k][128 rows] layout (256 B per k-row); thread t owns the 16-byte
// chunk (t&15) of k-rows (t>>4) + 16*i.
__device__ __forceinline__ void gload_kminor(uint4 (&r)[4], const bf16* P, long ld, int R,
                                             int row0, int k0, int kend, int tid) {
  const int grow = row0 + (tid & 15) * 8;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gk = k0 + (tid >> 4) + 16 * i;
    if (grow < R && gk < kend) {
      r[i] = *reinterpret_cast<const uint4*>(P + (long)gk * ld + grow);
    } else {
      r[i] = make_uint4(0, 0, 0, 0);
    }
  }
}

// ---- registers -> LDS -------------------------------------------------------
__device__ __forceinline__ void sstore_kmajor(const uint4 (&r)[4], char* tile, int tid) {
  const int c = tid & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (tid >> 3) + 32 * i;
    *reinterpret_cast<uint4*>(tile + row * 128 + ((c ^ swz(row)) << 4)) = r[i];
  }
}
// K-minor LDS image: tile[k][128 rows], 32-byte slot q (16 rows) of k-row k is
// stored at slot q ^ swzk(k), swzk(k) = (k & 3) | (((k >> 3) & 1) << 2): the 8
// k-rows {8g..8g+3, 8g+8..8g+11} one half-wave touches in a transpose read land
// on 8 distinct 32-byte bank groups.
__device__ __forceinline__ int swzk(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

__device__ __forceinline__ void sstore_kminor(const uint4 (&r)[4], char* tile, int tid) {
  const int c = tid & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = (tid >> 4) + 16 * i;
    *reinterpret_cast<uint4*>(tile + k * 256 + ((((c >> 1) ^ swzk(k)) << 5) | ((c & 1) << 4))) = r[i];
  }
}

template <bool KM>
__device__ __forceinline__ void gload(uint4 (&r)[4], const bf16* P, long ld, int R, int row0,
                                      int k0, int kend, int tid) {
  if constexpr (KM) gload_kmajor(r, P, ld, R, row0, k0, kend, tid);
  else gload_kminor(r, P, ld, R, row0, k0, kend, tid);
}
template <bool KM>
__device__ __forceinline__ void sstore(const uint4 (&r)[4], char* tile, int tid) {
  if constexpr (KM) sstore_kmajor(r, tile, tid);
  else sstore_kminor(r, tile, tid);
}

__device__ __forceinline__ bf16x8 lds_frag(const char* tile, int row, int chunk) {
  const uint4 v = *reinterpret_cast<const uint4*>(tile + row * 128 + ((chunk ^ swz(row)) << 4));
  return __builtin_bit_cast(bf16x8, v);
}

// Fragment of a K-minor tile through the LDS transpose read
// (ds_read_b64_tr_b16): within a 16-lane group, lane i supplies the address of
// the 4-element chunk [k = i>>2][rows 4*(i&3)..+3] of a 4(k) x 16(row) block and
// receives the 4 k-values of row i (measured on gfx950, tools/probes/tr_probe.hip).
// rb = 16-row block index (0..7), ks = 32-wide k-step, lg = lane>>4, lr = lane&15.
typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ bf16x8 lds_frag_tr(const char* tile, int rb, int ks, int lg, int lr) {
  const int ka = ks * 32 + lg * 8 + (lr >> 2);
  const int kb = ka + 4;
  const char* pa = tile + ka * 256 + ((rb ^ swzk(ka)) << 5) + ((lr & 3) << 3);
  const char* pb = tile + kb * 256 + ((rb ^ swzk(kb)) << 5) + ((lr & 3) << 3);
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pa);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pb);
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <bool A_KM, bool B_KM>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lg = lane >> 4;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * p.k_chunk;
  const int kend = min(p.K, kbeg + p.k_chunk);
  const int nk = (kend - kbeg + BK - 1) / BK;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rb[4];
  gload<A_KM>(ra, p.A, p.lda, p.M, m0, kbeg, kend, tid);
  gload<B_KM>(rb, p.B, p.ldb, p.N, n0, kbeg, kend, tid);
  sstore<A_KM>(ra, smem, tid);
  sstore<B_KM>(rb, smem + TILE_BYTES, tid);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const char* sA = smem + (kt & 1) * 2 * TILE_BYTES;
    const char* sB = sA + TILE_BYTES;
    const bool more = (kt + 1 < nk);
    if (more) {
      const int k0 = kbeg + (kt + 1) * BK;
      gload<A_KM>(ra, p.A, p.lda, p.M, m0, k0, kend, tid);
      gload<B_KM>(rb, p.B, p.ldb, p.N, n0, k0, kend, tid);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (A_KM) af[i] = lds_frag(sA, wm * 64 + i * 16 + lr, ks * 4 + lg);
        else af[i] = lds_frag_tr(sA, wm * 4 + i, ks, lg, lr);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (B_KM) bfr[j] = lds_frag(sB, wn * 64 + j * 16 + lr, ks * 4 + lg);
        else bfr[j] = lds_frag_tr(sB, wn * 4 + j, ks, lg, lr);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (more) {
      char* dA = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
      sstore<A_KM>(ra, dA, tid);
      sstore<B_KM>(rb, dA + TILE_BYTES, tid);
    }
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m][n..n+3], m = m_base + lr, n = n_base + 4*lg
  const int epi = p.epi;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + lr;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + lg * 4;
      if (n >= p.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * p.alpha;
      if (epi == BV_EPI_ATOMIC) {
        float* c = reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) unsafeAtomicAdd(c + r, v[r]);
        continue;
      }
      if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      }
      if (epi == BV_EPI_RESIDUAL && !p.out_f32) {   // bf16 residual stream: aux has C's dtype
        const uint2 x = *reinterpret_cast<const uint2*>(
            reinterpret_cast<const bf16*>(p.aux) + (long)m * p.ldaux + n);
        v[0] += bflo(x.x); v[1] += bfhi(x.x); v[2] += bflo(x.y); v[3] += bfhi(x.y);
      } else if (epi == BV_EPI_RESIDUAL) {
        const float4 x = *reinterpret_cast<const float4*>(
            reinterpret_cast<const float*>(p.aux) + (long)m * p.ldaux + n);
        v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w;
      } else if (epi == BV_EPI_POS) {
        const float4 x = *reinterpret_cast<const float4*>(
            reinterpret_cast<const float*>(p.aux) + (long)(m % p.aux_rows) * p.ldaux + n);
        v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w;
      } else if (epi == BV_EPI_MUL) {
        const uint2 d = *reinterpret_cast<const uint2*>(
            reinterpret_cast<const bf16*>(p.aux) + (long)m * p.ldaux + n);
        v[0] *= bflo(d.x); v[1] *= bfhi(d.x); v[2] *= bflo(d.y); v[3] *= bfhi(d.y);
      } else if (epi == BV_EPI_GELU_BWD || epi == BV_EPI_GELU_BWD_EMIT) {
        const uint2 h = *reinterpret_cast<const uint2*>(
            reinterpret_cast<const bf16*>(p.aux) + (long)m * p.ldaux + n);
        uint2 go, dw;
        mlp_act_from_h(h.x, go.x, dw.x);
        mlp_act_from_h(h.y, go.y, dw.y);
        const float d[4] = {bflo(dw.x), bfhi(dw.x), bflo(dw.y), bfhi(dw.y)};   // gelu' rounded to bf16 (as GELU_GD stores it)
        if (epi == BV_EPI_GELU_BWD_EMIT)
          *reinterpret_cast<uint2*>(reinterpret_cast<bf16*>(p.C2) + (long)m * p.ldc + n) = go;
        v[0] *= d[0]; v[1] *= d[1]; v[2] *= d[2]; v[3] *= d[3];
      }
      if (p.colsum) {   // small / ragged problems only: one atomic per element
#pragma unroll
        for (int r = 0; r < 4; ++r) unsafeAtomicAdd(p.colsum + n + r, v[r]);
      }
      if (epi == BV_EPI_GELU_GD) {   // C = gelu(h), C2 = gelu'(h) of the bf16-rounded pre-activation
        uint2 hw, gw, dw;
        mlp_act_words(f32x2{v[0], v[1]}, hw.x, gw.x, dw.x);
        mlp_act_words(f32x2{v[2], v[3]}, hw.y, gw.y, dw.y);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16*>(p.C) + (long)m * p.ldc + n) = gw;
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16*>(p.C2) + (long)m * p.ldc + n) = dw;
        continue;
      }
      if (p.out_f32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n) =
            make_float4(v[0], v[1], v[2], v[3]);
      } else {
        uint2 o;
        o.x = pack_bf2(v[0], v[1]);
        o.y = pack_bf2(v[2], v[3]);
        if (epi == BV_EPI_GELU_G) {   // only gelu of the bf16-rounded pre-activation leaves the kernel
          o.x = pack_bf2(gelu_tanh_f(bflo(o.x)), gelu_tanh_f(bfhi(o.x)));
          o.y = pack_bf2(gelu_tanh_f(bflo(o.y)), gelu_tanh_f(bfhi(o.y)));
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16*>(p.C) + (long)m * p.ldc + n) = o;
        if (epi == BV_EPI_GELU) {   // gelu of the bf16-rounded pre-activation that is stored
          uint2 g;
          g.x = pack_bf2(gelu_tanh_f(bflo(o.x)), gelu_tanh_f(bfhi(o.x)));
          g.y = pack_bf2(gelu_tanh_f(bflo(o.y)), gelu_tanh_f(bfhi(o.y)));
          *reinterpret_cast<uint2*>(reinterpret_cast<bf16*>(p.C2) + (long)m * p.ldc + n) = g;
        }
      }
    }
  }
}

}  // namespace

// gemm256.hip
int bv_gemm256_try(int a_kmajor, int b_kmajor, const void* A, long lda, const void* B, long ldb,
                   void* C, long ldc, int out_f32, int M, int N, int K, int epilogue,
                   const float* bias, const void* aux, long ldaux, int aux_rows, void* C2,
                   float alpha, int split_k, float* colsum, void* stream, const bv_ctx* ctx);

// See include/bvhip.h for the contract.
extern "C" int bv_gemm_bf16_colsum(int a_kmajor, int b_kmajor, const void* A, long lda, const void* B,
                                   long ldb, void* C, long ldc, int out_f32, int M, int N, int K,
                                   int epilogue, const float* bias, const void* aux, long ldaux,
                                   int aux_rows, void* C2, float alpha, int split_k, float* colsum,
                                   void* stream, const bv_ctx* ctx);
extern "C" int bv_gemm_bf16(int a_kmajor, int b_kmajor, const void* A, long lda, const void* B,
                            long ldb, void* C, long ldc, int out_f32, int M, int N, int K,
                            int epilogue, const float* bias, const void* aux, long ldaux,
                            int aux_rows, void* C2, float alpha, int split_k, void* stream, const bv_ctx* ctx) {
  return bv_gemm_bf16_colsum(a_kmajor, b_kmajor, A, lda, B, ldb, C, ldc, out_f32, M, N, K, epilogue, bias,
                             aux, ldaux, aux_rows, C2, alpha, split_k, nullptr, stream, ctx);
}
extern "C" int bv_gemm_bf16_colsum(int a_kmajor, int b_kmajor, const void* A, long lda, const void* B,
                                   long ldb, void* C, long ldc, int out_f32, int M, int N, int K,
                                   int epilogue, const float* bias, const void* aux, long ldaux,
                                   int aux_rows, void* C2, float alpha, int split_k, float* colsum,
                                   void* stream, const bv_ctx* ctx) {
  BV_REQUIRE(M > 0 && N > 0 && K > 0, "bv_gemm_bf16: empty problem M=%d N=%d K=%d", M, N, K);
  BV_REQUIRE(N % 8 == 0, "bv_gemm_bf16: N=%d must be a multiple of 8", N);
  BV_REQUIRE(a_kmajor ? (K % 8 == 0 && lda % 8 == 0) : (M % 8 == 0 && lda % 8 == 0),
             "bv_gemm_bf16: A contiguous dim / lda must be multiples of 8 (M=%d K=%d lda=%ld)", M, K, lda);
  BV_REQUIRE(b_kmajor ? (K % 8 == 0 && ldb % 8 == 0) : (ldb % 8 == 0),
             "bv_gemm_bf16: B contiguous dim / ldb must be multiples of 8 (N=%d K=%d ldb=%ld)", N, K, ldb);
  BV_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 8 == 0),
             "bv_gemm_bf16: operand pointers must be 16-byte aligned");
  BV_REQUIRE(epilogue >= BV_EPI_NONE && epilogue <= BV_EPI_GELU_G, "bv_gemm_bf16: bad epilogue %d", epilogue);
  BV_REQUIRE(ldc % 4 == 0, "bv_gemm_bf16: ldc=%ld must be a multiple of 4", ldc);
  if (epilogue == BV_EPI_RESIDUAL || epilogue == BV_EPI_POS || epilogue == BV_EPI_GELU_BWD ||
      epilogue == BV_EPI_GELU_BWD_EMIT || epilogue == BV_EPI_MUL)
    BV_REQUIRE(aux != nullptr && ldaux % 4 == 0, "bv_gemm_bf16: epilogue %d needs aux (ldaux %% 4 == 0)", epilogue);
  if (epilogue == BV_EPI_POS) BV_REQUIRE(aux_rows > 0, "bv_gemm_bf16: POS epilogue needs aux_rows > 0");
  if (epilogue == BV_EPI_GELU || epilogue == BV_EPI_GELU_BWD_EMIT || epilogue == BV_EPI_GELU_GD)
    BV_REQUIRE(C2 != nullptr && !out_f32, "bv_gemm_bf16: epilogue %d needs bf16 C and C2", epilogue);
  if (epilogue == BV_EPI_POS || epilogue == BV_EPI_ATOMIC)   // RESIDUAL: aux and C share one dtype (fp32 or bf16)
    BV_REQUIRE(out_f32, "bv_gemm_bf16: epilogue %d writes fp32", epilogue);
  if (epilogue == BV_EPI_GELU_BWD || epilogue == BV_EPI_MUL || epilogue == BV_EPI_GELU_G)
    BV_REQUIRE(!out_f32, "bv_gemm_bf16: epilogue %d writes bf16 (out_f32 must be 0)", epilogue);
  if (epilogue == BV_EPI_ATOMIC) BV_REQUIRE(bias == nullptr, "bv_gemm_bf16: ATOMIC epilogue takes no bias");
  if (colsum)
    BV_REQUIRE(epilogue == BV_EPI_GELU_BWD || epilogue == BV_EPI_GELU_BWD_EMIT || epilogue == BV_EPI_MUL,
               "bv_gemm_bf16_colsum: column sums are fused into the GELU_BWD / MUL epilogues only (got %d)", epilogue);

  if (bv_opt(ctx, BV_OPT_FAST_PATH) &&
      bv_gemm256_try(a_kmajor, b_kmajor, A, lda, B, ldb, C, ldc, out_f32, M, N, K, epilogue, bias, aux, ldaux, aux_rows,
                     C2, alpha, split_k, colsum, stream, ctx))
    return bv_check_launch("bv_gemm_bf16(256x256)");

  GemmParams p;
  p.A = (const bf16*)A; p.B = (const bf16*)B; p.C = C; p.C2 = C2;
  p.bias = bias; p.aux = aux; p.colsum = colsum;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldaux = ldaux;
  p.M = M; p.N = N; p.K = K; p.aux_rows = aux_rows > 0 ? aux_rows : 1;
  p.epi = epilogue; p.out_f32 = out_f32; p.alpha = alpha;

  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  int splits = 1;
  if (epilogue == BV_EPI_ATOMIC) {
    const int ksteps = (K + BK - 1) / BK;
    if (split_k > 0) {
      splits = split_k;
    } else {
      // aim for >= 1024 workgroups (256 CUs x 2 resident x 2 waves of work)
      // while keeping >= 8 K-steps per split.
      splits = (1024 + tiles_m * tiles_n - 1) / (tiles_m * tiles_n);
      const int max_splits = ksteps / 8 > 0 ? ksteps / 8 : 1;
      if (splits > max_splits) splits = max_splits;
    }
    if (splits < 1) splits = 1;
    if (splits > ksteps) splits = ksteps;
    const int steps_per = (ksteps + splits - 1) / splits;
    p.k_chunk = steps_per * BK;
    splits = (K + p.k_chunk - 1) / p.k_chunk;
  } else {
    p.k_chunk = ((K + BK - 1) / BK) * BK;
  }
  BV_REQUIRE(tiles_m <= 65535 && splits <= 65535, "bv_gemm_bf16: grid too large");
  dim3 grid(tiles_n, tiles_m, splits), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (a_kmajor && b_kmajor) hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, block, 0, s, p);
  else if (a_kmajor && !b_kmajor) hipLaunchKernelGGL((gemm_bf16_kernel<true, false>), grid, block, 0, s, p);
  else if (!a_kmajor && !b_kmajor) hipLaunchKernelGGL((gemm_bf16_kernel<false, false>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((gemm_bf16_kernel<false, true>), grid, block, 0, s, p);
  return bv_check_launch("bv_gemm_bf16");
}
