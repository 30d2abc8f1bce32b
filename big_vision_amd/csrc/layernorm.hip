// LayerNorm forward / backward for gfx950 — HBM-bound, one 64-lane wave per row.
//
// Replaces flax nn.LayerNorm() on the hot path (reference
// big_vision/models/vit.py:92,103 (block LNs), :160 (encoder_norm), :181 (MAP
// head LN)): eps = 1e-6, statistics in fp32 with var = E[x^2] - E[x]^2 clamped
// at 0 (Flax use_fast_variance), y = (x - mean) * rstd * scale + bias.
// The residual stream x stays fp32; the normalised output feeds a bf16 MFMA
// GEMM so it is emitted as bf16 (and optionally fp32).  The backward fuses the
// residual-gradient add and the bf16 copy needed by the next GEMM, and reduces
// dscale/dbias per workgroup before one fp32 atomic per column.
#include "bv_common.h"
#include "bvhip_internal.h"

namespace {

constexpr int MAXV = 8;  // float4 per lane -> D <= 2048

__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x,
                                                     const float* __restrict__ scale,
                                                     const float* __restrict__ bias,
                                                     bf16* __restrict__ y_bf, float* __restrict__ y_f,
                                                     float* __restrict__ mean_o,
                                                     float* __restrict__ rstd_o, int rows, int D,
                                                     long row_stride, long row_offset, float eps) {
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  const float inv_d = 1.0f / (float)D;
  for (int r = wave_global; r < rows; r += nwaves) {
    const float* xr = x + ((long)r * row_stride + row_offset) * D;
    float4 v[MAXV];
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int it = 0; it < MAXV; ++it) {
      const int c = lane * 4 + it * 256;
      if (c < D) {
        v[it] = *reinterpret_cast<const float4*>(xr + c);
        s += v[it].x + v[it].y + v[it].z + v[it].w;
        ss += v[it].x * v[it].x + v[it].y * v[it].y + v[it].z * v[it].z + v[it].w * v[it].w;
      }
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    const float mean = s * inv_d;
    const float var = fmaxf(ss * inv_d - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps);
    if (lane == 0) {
      if (mean_o) mean_o[r] = mean;
      if (rstd_o) rstd_o[r] = rstd;
    }
#pragma unroll
    for (int it = 0; it < MAXV; ++it) {
      const int c = lane * 4 + it * 256;
      if (c < D) {
        const float4 g = *reinterpret_cast<const float4*>(scale + c);
        const float4 b = *reinterpret_cast<const float4*>(bias + c);
        float4 o;
        o.x = (v[it].x - mean) * rstd * g.x + b.x;
        o.y = (v[it].y - mean) * rstd * g.y + b.y;
        o.z = (v[it].z - mean) * rstd * g.z + b.z;
        o.w = (v[it].w - mean) * rstd * g.w + b.w;
        if (y_bf) {
          uint2 p;
          p.x = pack_bf2(o.x, o.y);
          p.y = pack_bf2(o.z, o.w);
          *reinterpret_cast<uint2*>(y_bf + (long)r * D + c) = p;
        }
        if (y_f) *reinterpret_cast<float4*>(y_f + (long)r * D + c) = o;
      }
    }
  }
}

// NV = float4 per lane and row (D <= 256 * NV): the per-column partials live in registers, so
// the ViT widths (768 -> 3, 1024 -> 4) get their own instantiation (3x fewer VGPRs than the
// generic NV = 8, more rows in flight per CU).
template <bool DY_F32, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy_,
                                                     const float* __restrict__ x,
                                                     const float* __restrict__ scale,
                                                     const float* __restrict__ mean_i,
                                                     const float* __restrict__ rstd_i,
                                                     const float* __restrict__ dres,
                                                     float* __restrict__ dx, bf16* __restrict__ dx_bf,
                                                     float* __restrict__ dscale,
                                                     float* __restrict__ dbias,
                                                     float* __restrict__ dxsum, int rows, int D,
                                                     long row_stride, long row_offset) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [3][4][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_global = blockIdx.x * 4 + wave;
  const int nwaves = gridDim.x * 4;
  const float inv_d = 1.0f / (float)D;
  float4 ps[NV], pb[NV], po[NV];
#pragma unroll
  for (int it = 0; it < NV; ++it) {
    ps[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    pb[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    po[it] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int r = wave_global; r < rows; r += nwaves) {
    const long xrow = (long)r * row_stride + row_offset;
    const float* xr = x + xrow * D;
    const float mean = mean_i[r], rstd = rstd_i[r];
    float4 g[NV], xh[NV], dr[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int c = lane * 4 + it * 256;
      if (c < D) {
        float4 d;
        if constexpr (DY_F32) {
          d = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dy_) + (long)r * D + c);
        } else {
          const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16*>(dy_) + (long)r * D + c);
          d = make_float4(bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y));
        }
        const float4 xv = *reinterpret_cast<const float4*>(xr + c);
        const float4 sc = *reinterpret_cast<const float4*>(scale + c);
        dr[it] = dres ? *reinterpret_cast<const float4*>(dres + xrow * D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        xh[it] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd,
                             (xv.w - mean) * rstd);
        g[it] = make_float4(d.x * sc.x, d.y * sc.y, d.z * sc.z, d.w * sc.w);
        s1 += g[it].x + g[it].y + g[it].z + g[it].w;
        s2 += g[it].x * xh[it].x + g[it].y * xh[it].y + g[it].z * xh[it].z + g[it].w * xh[it].w;
        ps[it].x += d.x * xh[it].x; ps[it].y += d.y * xh[it].y;
        ps[it].z += d.z * xh[it].z; ps[it].w += d.w * xh[it].w;
        pb[it].x += d.x; pb[it].y += d.y; pb[it].z += d.z; pb[it].w += d.w;
      }
    }
    s1 = wave_sum(s1) * inv_d;
    s2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int c = lane * 4 + it * 256;
      if (c < D) {
        float4 o;
        o.x = rstd * (g[it].x - s1 - xh[it].x * s2);
        o.y = rstd * (g[it].y - s1 - xh[it].y * s2);
        o.z = rstd * (g[it].z - s1 - xh[it].z * s2);
        o.w = rstd * (g[it].w - s1 - xh[it].w * s2);
        o.x += dr[it].x; o.y += dr[it].y; o.z += dr[it].z; o.w += dr[it].w;
        *reinterpret_cast<float4*>(dx + xrow * D + c) = o;
        po[it].x += o.x; po[it].y += o.y; po[it].z += o.z; po[it].w += o.w;
        if (dx_bf) {
          uint2 p;
          p.x = pack_bf2(o.x, o.y);
          p.y = pack_bf2(o.z, o.w);
          *reinterpret_cast<uint2*>(dx_bf + xrow * D + c) = p;
        }
      }
    }
  }
  // cross-wave reduction of the per-column partials, then one atomic per column.
  float* rs = red;               // [4][D]
  float* rb = red + 4 * D;       // [4][D]
  float* ro = red + 8 * D;       // [4][D]
#pragma unroll
  for (int it = 0; it < NV; ++it) {
    const int c = lane * 4 + it * 256;
    if (c < D) {
      *reinterpret_cast<float4*>(rs + wave * D + c) = ps[it];
      *reinterpret_cast<float4*>(rb + wave * D + c) = pb[it];
      *reinterpret_cast<float4*>(ro + wave * D + c) = po[it];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    const float a = rs[c] + rs[D + c] + rs[2 * D + c] + rs[3 * D + c];
    const float b = rb[c] + rb[D + c] + rb[2 * D + c] + rb[3 * D + c];
    if (dscale) unsafeAtomicAdd(dscale + c, a);
    if (dbias) unsafeAtomicAdd(dbias + c, b);
    if (dxsum) unsafeAtomicAdd(dxsum + c, ro[c] + ro[D + c] + ro[2 * D + c] + ro[3 * D + c]);
  }
}

}  // namespace

extern "C" int bv_layernorm_fwd(const float* x, const float* scale, const float* bias, void* y_bf16,
                                float* y_f32, float* mean, float* rstd, int rows, int D,
                                long row_stride, long row_offset, float eps, void* stream) {
  BV_REQUIRE(rows > 0 && D > 0, "bv_layernorm_fwd: empty input rows=%d D=%d", rows, D);
  BV_REQUIRE(D % 4 == 0 && D <= 256 * MAXV, "bv_layernorm_fwd: D=%d must be a multiple of 4 and <= %d", D, 256 * MAXV);
  BV_REQUIRE(row_stride >= 1 && row_offset >= 0 && row_offset < row_stride, "bv_layernorm_fwd: bad row_stride/offset");
  BV_REQUIRE(y_bf16 || y_f32, "bv_layernorm_fwd: no output requested");
  int grid = (rows + 3) / 4;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, scale, bias,
                     (bf16*)y_bf16, y_f32, mean, rstd, rows, D, row_stride, row_offset, eps);
  return bv_check_launch("bv_layernorm_fwd");
}

extern "C" int bv_layernorm_bwd(const void* dy, int dy_is_f32, const float* x, const float* scale,
                                const float* mean, const float* rstd, const float* dres, float* dx,
                                void* dx_bf16, float* dscale, float* dbias, float* dx_colsum, int rows,
                                int D, long row_stride, long row_offset, void* stream) {
  BV_REQUIRE(rows > 0 && D > 0, "bv_layernorm_bwd: empty input rows=%d D=%d", rows, D);
  BV_REQUIRE(D % 4 == 0 && D <= 256 * MAXV, "bv_layernorm_bwd: D=%d must be a multiple of 4 and <= %d", D, 256 * MAXV);
  BV_REQUIRE(row_stride >= 1 && row_offset >= 0 && row_offset < row_stride, "bv_layernorm_bwd: bad row_stride/offset");
  BV_REQUIRE(mean && rstd && dx, "bv_layernorm_bwd: mean/rstd/dx required");
  const int nv = D <= 768 ? 3 : (D <= 1024 ? 4 : MAXV);
  int grid = (rows + 3) / 4;
  const int max_grid = nv <= 4 ? 1024 : 512;   // workgroups resident per CU: 4 / 2
  if (grid > max_grid) grid = max_grid;
  const size_t shmem = sizeof(float) * 12 * D;
#define BV_LN_BWD(F32, NV)                                                                              \
  hipLaunchKernelGGL((ln_bwd_kernel<F32, NV>), dim3(grid), dim3(256), shmem, (hipStream_t)stream, dy, x, \
                     scale, mean, rstd, dres, dx, (bf16*)dx_bf16, dscale, dbias, dx_colsum, rows, D,     \
                     row_stride, row_offset)
  if (dy_is_f32) {
    if (nv == 3) BV_LN_BWD(true, 3); else if (nv == 4) BV_LN_BWD(true, 4); else BV_LN_BWD(true, MAXV);
  } else {
    if (nv == 3) BV_LN_BWD(false, 3); else if (nv == 4) BV_LN_BWD(false, 4); else BV_LN_BWD(false, MAXV);
  }
#undef BV_LN_BWD
  return bv_check_launch("bv_layernorm_bwd");
}
