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

// Streaming accesses of the fp32-stream kernels.  Every activation these kernels touch is read or written once
// per launch.  NT bit 0 marks the loads, bit 1 the stores non-temporal (tools/probes/ln_probe.hip, all four
// combinations on the step's shapes): streams far larger than the 256 MB memory-side cache gain 3-5 % from both
// (NT = 3: fwd 5.5 -> 5.8 TB/s, bwd 5.3 -> 5.6 on 401 408 x 768); when the inputs can still be cache-resident
// (32 768 x 768: 100 MB) non-temporal LOADS lose 15 %, so the host picks NT = 2 there (ln_nt_for).
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
template <int NT>
__device__ __forceinline__ float4 ld_stream4(const float* p) {
  if constexpr ((NT & 1) != 0) {
    const f32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
  } else {
    return *reinterpret_cast<const float4*>(p);
  }
}
template <int NT>
__device__ __forceinline__ uint2 ld_stream_bf4(const bf16* p) {
  if constexpr ((NT & 1) != 0) {
    const u32x2_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p));
    return make_uint2(t.x, t.y);
  } else {
    return *reinterpret_cast<const uint2*>(p);
  }
}
template <int NT>
__device__ __forceinline__ void st_stream4(float* p, const float4& v) {
  if constexpr ((NT & 2) != 0) {
    f32x4_t t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4_t*>(p));
  } else {
    *reinterpret_cast<float4*>(p) = v;
  }
}
template <int NT>
__device__ __forceinline__ void st_stream_bf4(bf16* p, const uint2& v) {
  if constexpr ((NT & 2) != 0) {
    u32x2_t t; t.x = v.x; t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<u32x2_t*>(p));
  } else {
    *reinterpret_cast<uint2*>(p) = v;
  }
}
// rows x D fp32 activations: larger than what the memory-side cache can still hold from the producer?
inline int ln_nt_for(long rows, int D) { return rows * (long)D * 4 > (192L << 20) ? 3 : 2; }

// NV = float4 per lane and row (D <= 256 * NV).  A wave normalises TWO rows per trip: both rows' loads are
// issued before either reduction (tools/probes/ln_fwd_probe.hip on 401 408 x 768: one row per trip 5.2 TB/s,
// two 5.5, two with streaming accesses 5.8; four rows, or scale / bias held in registers across the trips, cost
// occupancy and fall to 3.8-4.2).  The grid cap matters as much: 2048 workgroups 4.4 TB/s, 8192 5.7.
template <int NV, int NT>
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
  for (int r0 = wave_global; r0 < rows; r0 += 2 * nwaves) {
    float4 v[2][NV];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = r0 + j * nwaves;
      if (r >= rows) continue;
      const float* xr = x + ((long)r * row_stride + row_offset) * D;
#pragma unroll
      for (int it = 0; it < NV; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < D) v[j][it] = ld_stream4<NT>(xr + c);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = r0 + j * nwaves;
      if (r >= rows) continue;
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int it = 0; it < NV; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < D) {
          s += v[j][it].x + v[j][it].y + v[j][it].z + v[j][it].w;
          ss += v[j][it].x * v[j][it].x + v[j][it].y * v[j][it].y + v[j][it].z * v[j][it].z + v[j][it].w * v[j][it].w;
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
      for (int it = 0; it < NV; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < D) {
          const float4 g = *reinterpret_cast<const float4*>(scale + c);
          const float4 b = *reinterpret_cast<const float4*>(bias + c);
          float4 o;
          o.x = (v[j][it].x - mean) * rstd * g.x + b.x;
          o.y = (v[j][it].y - mean) * rstd * g.y + b.y;
          o.z = (v[j][it].z - mean) * rstd * g.z + b.z;
          o.w = (v[j][it].w - mean) * rstd * g.w + b.w;
          if (y_bf) {
            uint2 p;
            p.x = pack_bf2(o.x, o.y);
            p.y = pack_bf2(o.z, o.w);
            st_stream_bf4<NT>(y_bf + (long)r * D + c, p);
          }
          if (y_f) st_stream4<NT>(y_f + (long)r * D + c, o);
        }
      }
    }
  }
}

// NV = float4 per lane and row (D <= 256 * NV): the per-column partials live in registers, so
// the ViT widths (768 -> 3, 1024 -> 4) get their own instantiation (3x fewer VGPRs than the
// generic NV = 8, more rows in flight per CU).
template <bool DY_F32, int NV, int NT>
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
                                                     long row_stride, long row_offset,
                                                     const float* __restrict__ bias, bf16* __restrict__ y_out) {
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
          d = ld_stream4<NT>(reinterpret_cast<const float*>(dy_) + (long)r * D + c);
        } else {
          const uint2 u = ld_stream_bf4<NT>(reinterpret_cast<const bf16*>(dy_) + (long)r * D + c);
          d = make_float4(bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y));
        }
        const float4 xv = ld_stream4<NT>(xr + c);
        const float4 sc = *reinterpret_cast<const float4*>(scale + c);
        dr[it] = dres ? ld_stream4<NT>(dres + xrow * D + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        xh[it] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd,
                             (xv.w - mean) * rstd);
        if (y_out) {
          // the forward's bf16 output, re-derived from the x this kernel reads anyway (same expression as
          // ln_fwd_kernel): "light" contexts do not keep it, and a separate re-normalisation pass would
          // read x a second time (4 of its 6 bytes per element)
          const float4 bb = *reinterpret_cast<const float4*>(bias + c);
          float4 o;
          o.x = (xv.x - mean) * rstd * sc.x + bb.x;
          o.y = (xv.y - mean) * rstd * sc.y + bb.y;
          o.z = (xv.z - mean) * rstd * sc.z + bb.z;
          o.w = (xv.w - mean) * rstd * sc.w + bb.w;
          uint2 pk;
          pk.x = pack_bf2(o.x, o.y);
          pk.y = pack_bf2(o.z, o.w);
          st_stream_bf4<NT>(y_out + (long)r * D + c, pk);
        }
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
        st_stream4<NT>(dx + xrow * D + c, o);
        po[it].x += o.x; po[it].y += o.y; po[it].z += o.z; po[it].w += o.w;
        if (dx_bf) {
          uint2 p;
          p.x = pack_bf2(o.x, o.y);
          p.y = pack_bf2(o.z, o.w);
          st_stream_bf4<NT>(dx_bf + xrow * D + c, p);
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

// ---- bf16 residual stream (config.residual_stream = "bfloat16"): x, the residual gradient dres and
// dx are bf16; statistics, scale / bias gradients and all arithmetic stay fp32.  A lane handles 8
// consecutive elements (16-byte accesses): NV8 = ceil(D / 512) pieces per lane and row.
constexpr int MAXV8 = 4;   // D <= 2048

__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
  f[0] = bflo(u.x); f[1] = bfhi(u.x); f[2] = bflo(u.y); f[3] = bfhi(u.y);
  f[4] = bflo(u.z); f[5] = bfhi(u.z); f[6] = bflo(u.w); f[7] = bfhi(u.w);
}
__device__ __forceinline__ uint4 pack8f(const float (&f)[8]) {
  return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
}

// Two rows per wave iteration: a bf16 row is half the bytes of an fp32 one, and with one row in flight per
// wave the kernel sat at 3.7 TB/s (latency x bytes in flight); NV pieces of 256 elements per row,
// 8-byte accesses (4 elements per lane and piece: every lane is busy at D = 768).
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_bfx_kernel(const bf16* __restrict__ x,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ bias,
                                                         bf16* __restrict__ y_bf, float* __restrict__ y_f,
                                                         float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                         int rows, int D, long row_stride, long row_offset,
                                                         float eps) {
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform
  const int nwaves = gridDim.x * 4;
  const float inv_d = 1.0f / (float)D;
  for (int r0 = 2 * wave_global; r0 < rows; r0 += 2 * nwaves) {
    uint2 raw[2][NV];
    float s[2] = {0.f, 0.f}, ss[2] = {0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const bf16* xr = x + ((long)(r0 + rr) * row_stride + row_offset) * D;
#pragma unroll
      for (int it = 0; it < NV; ++it) {
        const int c = lane * 4 + it * 256;
        raw[rr][it] = make_uint2(0, 0);
        if (c < D && r0 + rr < rows) raw[rr][it] = *reinterpret_cast<const uint2*>(xr + c);
      }
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int it = 0; it < NV; ++it) {
        const float a = bflo(raw[rr][it].x), b = bfhi(raw[rr][it].x), c_ = bflo(raw[rr][it].y), d = bfhi(raw[rr][it].y);
        s[rr] += a + b + c_ + d;
        ss[rr] += a * a + b * b + c_ * c_ + d * d;
      }
    float mean[2], rstd[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      s[rr] = wave_sum(s[rr]);
      ss[rr] = wave_sum(ss[rr]);
      mean[rr] = s[rr] * inv_d;
      rstd[rr] = rsqrtf(fmaxf(ss[rr] * inv_d - mean[rr] * mean[rr], 0.f) + eps);
      if (lane == 0 && r0 + rr < rows) {
        if (mean_o) mean_o[r0 + rr] = mean[rr];
        if (rstd_o) rstd_o[r0 + rr] = rstd[rr];
      }
    }
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int c = lane * 4 + it * 256;
      if (c < D) {
        const float4 g = *reinterpret_cast<const float4*>(scale + c);
        const float4 b = *reinterpret_cast<const float4*>(bias + c);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          if (r0 + rr >= rows) continue;
          float4 o;
          o.x = (bflo(raw[rr][it].x) - mean[rr]) * rstd[rr] * g.x + b.x;
          o.y = (bfhi(raw[rr][it].x) - mean[rr]) * rstd[rr] * g.y + b.y;
          o.z = (bflo(raw[rr][it].y) - mean[rr]) * rstd[rr] * g.z + b.z;
          o.w = (bfhi(raw[rr][it].y) - mean[rr]) * rstd[rr] * g.w + b.w;
          if (y_bf) *reinterpret_cast<uint2*>(y_bf + (long)(r0 + rr) * D + c) = make_uint2(pack_bf2(o.x, o.y), pack_bf2(o.z, o.w));
          if (y_f) *reinterpret_cast<float4*>(y_f + (long)(r0 + rr) * D + c) = o;
        }
      }
    }
  }
}

// bf16 dy: the three bf16 streams of two rows stay PACKED in registers (6 VGPRs per piece and row) and are
// unpacked twice (statistics pass, output pass) instead of being held as fp32.  Branch-free loads (row and
// column indices are clamped, contributions of clamped elements are multiplied by 0): every load of the
// two rows is in flight before the first one is consumed.  FULL: D == NV * 256 (no column tail).
template <int NV, bool FULL, bool HAS_RES>
__global__ __launch_bounds__(256) void ln_bwd_bfx2_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ mean_i,
                                                          const float* __restrict__ rstd_i,
                                                          const bf16* __restrict__ dres, bf16* __restrict__ dx,
                                                          float* __restrict__ dscale, float* __restrict__ dbias,
                                                          float* __restrict__ dxsum, int rows, int D,
                                                          long row_stride, long row_offset) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [3][4][D]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: row bases live in SGPRs
  const int wave_global = blockIdx.x * 4 + wave;
  const int nwaves = gridDim.x * 4;
  const float inv_d = 1.0f / (float)D;
  float ps[NV][4], pb[NV][4], po[NV][4], sc[NV][4];
  int col[NV];
  float cm[NV];   // 1 for a real column, 0 for a clamped one
#pragma unroll
  for (int it = 0; it < NV; ++it) {
    const int c = lane * 4 + it * 256;
    col[it] = FULL ? c : min(c, D - 4);
    cm[it] = (FULL || c < D) ? 1.f : 0.f;
    const float4 s4 = *reinterpret_cast<const float4*>(scale + col[it]);
    sc[it][0] = s4.x; sc[it][1] = s4.y; sc[it][2] = s4.z; sc[it][3] = s4.w;
#pragma unroll
    for (int e = 0; e < 4; ++e) ps[it][e] = pb[it][e] = po[it][e] = 0.f;
  }
  for (int r0 = 2 * wave_global; r0 < rows; r0 += 2 * nwaves) {
    uint2 rd[2][NV], rx[2][NV], rr_[2][NV];
    float mean[2], rstd[2], rm[2], s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    long xrow[2];
    int rq[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      rq[q] = __builtin_amdgcn_readfirstlane(min(r0 + q, rows - 1));   // uniform: SGPR row bases
      rm[q] = r0 + q < rows ? 1.f : 0.f;
      xrow[q] = (long)rq[q] * row_stride + row_offset;
      mean[q] = mean_i[rq[q]];
      rstd[q] = rstd_i[rq[q]];
      const bf16* dyr = dy + (long)rq[q] * D;
      const bf16* xr = x + xrow[q] * D;
      const bf16* rr = HAS_RES ? dres + xrow[q] * D : nullptr;
#pragma unroll
      for (int it = 0; it < NV; ++it) {
        rd[q][it] = *reinterpret_cast<const uint2*>(dyr + col[it]);
        rx[q][it] = *reinterpret_cast<const uint2*>(xr + col[it]);
        if constexpr (HAS_RES) rr_[q][it] = *reinterpret_cast<const uint2*>(rr + col[it]);
        else rr_[q][it] = make_uint2(0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int it = 0; it < NV; ++it) {
        const float w = FULL ? rm[q] : rm[q] * cm[it];
        const float d[4] = {bflo(rd[q][it].x) * w, bfhi(rd[q][it].x) * w, bflo(rd[q][it].y) * w, bfhi(rd[q][it].y) * w};
        const float xv[4] = {bflo(rx[q][it].x), bfhi(rx[q][it].x), bflo(rx[q][it].y), bfhi(rx[q][it].y)};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (xv[e] - mean[q]) * rstd[q];
          const float g = d[e] * sc[it][e];
          s1[q] += g;
          s2[q] += g * xh;
          ps[it][e] += d[e] * xh;
          pb[it][e] += d[e];
        }
      }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      s1[q] = wave_sum(s1[q]) * inv_d;
      s2[q] = wave_sum(s2[q]) * inv_d;
    }
    // the packed registers pass through an empty asm: the output pass unpacks them AGAIN instead of keeping
    // the statistics pass's fp32 values (and x-hat) alive across the reduction
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int it = 0; it < NV; ++it) {
        unsigned a0 = rd[q][it].x, a1 = rd[q][it].y, b0 = rx[q][it].x, b1 = rx[q][it].y;
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
        rd[q][it] = make_uint2(a0, a1);
        rx[q][it] = make_uint2(b0, b1);
      }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int it = 0; it < NV; ++it) {
        const float w = FULL ? rm[q] : rm[q] * cm[it];
        const float d[4] = {bflo(rd[q][it].x), bfhi(rd[q][it].x), bflo(rd[q][it].y), bfhi(rd[q][it].y)};
        const float xv[4] = {bflo(rx[q][it].x), bfhi(rx[q][it].x), bflo(rx[q][it].y), bfhi(rx[q][it].y)};
        const float dr[4] = {bflo(rr_[q][it].x), bfhi(rr_[q][it].x), bflo(rr_[q][it].y), bfhi(rr_[q][it].y)};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (xv[e] - mean[q]) * rstd[q];
          o[e] = rstd[q] * (d[e] * sc[it][e] - s1[q] - xh * s2[q]) + dr[e];
          po[it][e] += o[e] * w;
        }
        bf16* dxr = dx + xrow[q] * D;
        if (w != 0.f) *reinterpret_cast<uint2*>(dxr + col[it]) = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
      }
    }
  }
  float* rs = red;               // [4][D]
  float* rb = red + 4 * D;       // [4][D]
  float* ro = red + 8 * D;       // [4][D]
#pragma unroll
  for (int it = 0; it < NV; ++it) {
    const int c = lane * 4 + it * 256;
    if (FULL || c < D) {
      *reinterpret_cast<float4*>(rs + wave * D + c) = make_float4(ps[it][0], ps[it][1], ps[it][2], ps[it][3]);
      *reinterpret_cast<float4*>(rb + wave * D + c) = make_float4(pb[it][0], pb[it][1], pb[it][2], pb[it][3]);
      *reinterpret_cast<float4*>(ro + wave * D + c) = make_float4(po[it][0], po[it][1], po[it][2], po[it][3]);
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

template <bool DY_F32, int NV8>
__global__ __launch_bounds__(256) void ln_bwd_bfx_kernel(const void* __restrict__ dy_, const bf16* __restrict__ x,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ mean_i,
                                                         const float* __restrict__ rstd_i,
                                                         const bf16* __restrict__ dres, bf16* __restrict__ dx,
                                                         float* __restrict__ dscale, float* __restrict__ dbias,
                                                         float* __restrict__ dxsum, int rows, int D,
                                                         long row_stride, long row_offset) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [3][4][D]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_global = blockIdx.x * 4 + wave;
  const int nwaves = gridDim.x * 4;
  const float inv_d = 1.0f / (float)D;
  float ps[NV8][8], pb[NV8][8], po[NV8][8];
#pragma unroll
  for (int it = 0; it < NV8; ++it)
#pragma unroll
    for (int e = 0; e < 8; ++e) ps[it][e] = pb[it][e] = po[it][e] = 0.f;
  for (int r = wave_global; r < rows; r += nwaves) {
    const long xrow = (long)r * row_stride + row_offset;
    const float mean = mean_i[r], rstd = rstd_i[r];
    float g[NV8][8], xh[NV8][8], dr[NV8][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NV8; ++it) {
      const int c = lane * 8 + it * 512;
      if (c < D) {
        float d[8], xv[8];
        if constexpr (DY_F32) {
          const float* dp = reinterpret_cast<const float*>(dy_) + (long)r * D + c;
          const float4 a = *reinterpret_cast<const float4*>(dp), b = *reinterpret_cast<const float4*>(dp + 4);
          d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
        } else {
          unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(dy_) + (long)r * D + c), d);
        }
        unpack8(*reinterpret_cast<const uint4*>(x + xrow * D + c), xv);
        if (dres) unpack8(*reinterpret_cast<const uint4*>(dres + xrow * D + c), dr[it]);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) dr[it][e] = 0.f;
        }
        float sc[8];
        *reinterpret_cast<float4*>(sc) = *reinterpret_cast<const float4*>(scale + c);
        *reinterpret_cast<float4*>(sc + 4) = *reinterpret_cast<const float4*>(scale + c + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[it][e] = (xv[e] - mean) * rstd;
          g[it][e] = d[e] * sc[e];
          s1 += g[it][e];
          s2 += g[it][e] * xh[it][e];
          ps[it][e] += d[e] * xh[it][e];
          pb[it][e] += d[e];
        }
      }
    }
    s1 = wave_sum(s1) * inv_d;
    s2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int it = 0; it < NV8; ++it) {
      const int c = lane * 8 + it * 512;
      if (c < D) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o[e] = rstd * (g[it][e] - s1 - xh[it][e] * s2) + dr[it][e];
          po[it][e] += o[e];
        }
        *reinterpret_cast<uint4*>(dx + xrow * D + c) = pack8f(o);
      }
    }
  }
  float* rs = red;               // [4][D]
  float* rb = red + 4 * D;       // [4][D]
  float* ro = red + 8 * D;       // [4][D]
#pragma unroll
  for (int it = 0; it < NV8; ++it) {
    const int c = lane * 8 + it * 512;
    if (c < D) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        rs[wave * D + c + e] = ps[it][e];
        rb[wave * D + c + e] = pb[it][e];
        ro[wave * D + c + e] = po[it][e];
      }
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

// bf16 residual stream: x bf16 [rows*row_stride][D] -> y (bf16 and / or fp32), fp32 statistics.
extern "C" int bv_layernorm_fwd_bf16x(const void* x_bf16, const float* scale, const float* bias, void* y_bf16,
                                      float* y_f32, float* mean, float* rstd, int rows, int D, long row_stride,
                                      long row_offset, float eps, void* stream) {
  BV_REQUIRE(rows > 0 && D > 0, "bv_layernorm_fwd_bf16x: empty input rows=%d D=%d", rows, D);
  BV_REQUIRE(D % 8 == 0 && D <= 2048, "bv_layernorm_fwd_bf16x: D=%d must be a multiple of 8 and <= 2048", D);
  BV_REQUIRE(row_stride >= 1 && row_offset >= 0 && row_offset < row_stride, "bv_layernorm_fwd_bf16x: bad row_stride/offset");
  BV_REQUIRE(y_bf16 || y_f32, "bv_layernorm_fwd_bf16x: no output requested");
  int grid = (rows + 7) / 8;   // 4 waves x 2 rows per workgroup iteration
  if (grid > 4096) grid = 4096;
  const int nv = (D + 255) / 256;
#define BV_LN_FWDX(NV)                                                                                         \
  hipLaunchKernelGGL((ln_fwd_bfx_kernel<NV>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)x_bf16, \
                     scale, bias, (bf16*)y_bf16, y_f32, mean, rstd, rows, D, row_stride, row_offset, eps)
  if (nv <= 3) BV_LN_FWDX(3); else if (nv == 4) BV_LN_FWDX(4); else if (nv == 5) BV_LN_FWDX(5); else BV_LN_FWDX(8);
#undef BV_LN_FWDX
  return bv_check_launch("bv_layernorm_fwd_bf16x");
}

// dx (bf16) = dres (bf16, optional) + LN_bwd(dy); dscale / dbias / dx_colsum accumulated in fp32.
extern "C" int bv_layernorm_bwd_bf16x(const void* dy, int dy_is_f32, const void* x_bf16, const float* scale,
                                      const float* mean, const float* rstd, const void* dres_bf16, void* dx_bf16,
                                      float* dscale, float* dbias, float* dx_colsum, int rows, int D,
                                      long row_stride, long row_offset, void* stream) {
  BV_REQUIRE(rows > 0 && D > 0, "bv_layernorm_bwd_bf16x: empty input rows=%d D=%d", rows, D);
  BV_REQUIRE(D % 8 == 0 && D <= 512 * MAXV8, "bv_layernorm_bwd_bf16x: D=%d must be a multiple of 8 and <= %d", D, 512 * MAXV8);
  BV_REQUIRE(row_stride >= 1 && row_offset >= 0 && row_offset < row_stride, "bv_layernorm_bwd_bf16x: bad row_stride/offset");
  BV_REQUIRE(mean && rstd && dx_bf16, "bv_layernorm_bwd_bf16x: mean/rstd/dx required");
  const size_t shmem = sizeof(float) * 12 * D;
  if (!dy_is_f32) {   // the hot case (dy = a bf16 dX GEMM output): packed two-row kernel
    const int nv4 = (D + 255) / 256;
    int grid2 = (rows + 7) / 8;
    if (grid2 > 1280) grid2 = 1280;   // 5 workgroups per CU
#define BV_LN_BWDX2(NV, FULL, RES)                                                                            \
  hipLaunchKernelGGL((ln_bwd_bfx2_kernel<NV, FULL, RES>), dim3(grid2), dim3(256), shmem, (hipStream_t)stream,  \
                     (const bf16*)dy, (const bf16*)x_bf16, scale, mean, rstd, (const bf16*)dres_bf16,          \
                     (bf16*)dx_bf16, dscale, dbias, dx_colsum, rows, D, row_stride, row_offset)
#define BV_LN_BWDX2_NV(NV)                                                                  \
  do {                                                                                      \
    if (D == NV * 256) { if (dres_bf16) BV_LN_BWDX2(NV, true, true); else BV_LN_BWDX2(NV, true, false); }   \
    else { if (dres_bf16) BV_LN_BWDX2(NV, false, true); else BV_LN_BWDX2(NV, false, false); }               \
  } while (0)
    if (nv4 <= 3) BV_LN_BWDX2_NV(3); else if (nv4 == 4) BV_LN_BWDX2_NV(4); else if (nv4 == 5) BV_LN_BWDX2_NV(5); else BV_LN_BWDX2_NV(8);
#undef BV_LN_BWDX2_NV
#undef BV_LN_BWDX2
    return bv_check_launch("bv_layernorm_bwd_bf16x");
  }
  const int nv = D <= 1024 ? 2 : MAXV8;
  int grid = (rows + 3) / 4;
  const int max_grid = nv <= 2 ? 1024 : 512;
  if (grid > max_grid) grid = max_grid;
#define BV_LN_BWDX(F32, NV)                                                                                  \
  hipLaunchKernelGGL((ln_bwd_bfx_kernel<F32, NV>), dim3(grid), dim3(256), shmem, (hipStream_t)stream, dy,     \
                     (const bf16*)x_bf16, scale, mean, rstd, (const bf16*)dres_bf16, (bf16*)dx_bf16, dscale,  \
                     dbias, dx_colsum, rows, D, row_stride, row_offset)
  if (dy_is_f32) { if (nv == 2) BV_LN_BWDX(true, 2); else BV_LN_BWDX(true, MAXV8); }
  else { if (nv == 2) BV_LN_BWDX(false, 2); else BV_LN_BWDX(false, MAXV8); }
#undef BV_LN_BWDX
  return bv_check_launch("bv_layernorm_bwd_bf16x");
}

extern "C" int bv_layernorm_fwd(const float* x, const float* scale, const float* bias, void* y_bf16,
                                float* y_f32, float* mean, float* rstd, int rows, int D,
                                long row_stride, long row_offset, float eps, void* stream) {
  BV_REQUIRE(rows > 0 && D > 0, "bv_layernorm_fwd: empty input rows=%d D=%d", rows, D);
  BV_REQUIRE(D % 4 == 0 && D <= 256 * MAXV, "bv_layernorm_fwd: D=%d must be a multiple of 4 and <= %d", D, 256 * MAXV);
  BV_REQUIRE(row_stride >= 1 && row_offset >= 0 && row_offset < row_stride, "bv_layernorm_fwd: bad row_stride/offset");
  BV_REQUIRE(y_bf16 || y_f32, "bv_layernorm_fwd: no output requested");
  int grid = (rows + 7) / 8;          // 4 waves per workgroup, 2 rows per wave and trip
  if (grid > 8192) grid = 8192;
  const int nt = ln_nt_for(rows, D);
#define BV_LN_FWD(NV, NT)                                                                                        \
  hipLaunchKernelGGL((ln_fwd_kernel<NV, NT>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, scale, bias,      \
                     (bf16*)y_bf16, y_f32, mean, rstd, rows, D, row_stride, row_offset, eps)
  if (D <= 768) { if (nt == 3) BV_LN_FWD(3, 3); else BV_LN_FWD(3, 2); }
  else if (D <= 1024) { if (nt == 3) BV_LN_FWD(4, 3); else BV_LN_FWD(4, 2); }
  else { if (nt == 3) BV_LN_FWD(MAXV, 3); else BV_LN_FWD(MAXV, 2); }
#undef BV_LN_FWD
  return bv_check_launch("bv_layernorm_fwd");
}

extern "C" int bv_layernorm_bwd_y(const void* dy, int dy_is_f32, const float* x, const float* scale,
                                  const float* mean, const float* rstd, const float* dres, float* dx,
                                  void* dx_bf16, float* dscale, float* dbias, float* dx_colsum, int rows,
                                  int D, long row_stride, long row_offset, const float* bias, void* y_bf16,
                                  void* stream);
extern "C" int bv_layernorm_bwd(const void* dy, int dy_is_f32, const float* x, const float* scale,
                                const float* mean, const float* rstd, const float* dres, float* dx,
                                void* dx_bf16, float* dscale, float* dbias, float* dx_colsum, int rows,
                                int D, long row_stride, long row_offset, void* stream) {
  return bv_layernorm_bwd_y(dy, dy_is_f32, x, scale, mean, rstd, dres, dx, dx_bf16, dscale, dbias, dx_colsum, rows, D,
                            row_stride, row_offset, nullptr, nullptr, stream);
}
extern "C" int bv_layernorm_bwd_y(const void* dy, int dy_is_f32, const float* x, const float* scale,
                                  const float* mean, const float* rstd, const float* dres, float* dx,
                                  void* dx_bf16, float* dscale, float* dbias, float* dx_colsum, int rows,
                                  int D, long row_stride, long row_offset, const float* bias, void* y_bf16,
                                  void* stream) {
  BV_REQUIRE(!y_bf16 || (bias && row_stride == 1), "bv_layernorm_bwd_y: y_bf16 needs the LayerNorm bias and row_stride 1");
  BV_REQUIRE(rows > 0 && D > 0, "bv_layernorm_bwd: empty input rows=%d D=%d", rows, D);
  BV_REQUIRE(D % 4 == 0 && D <= 256 * MAXV, "bv_layernorm_bwd: D=%d must be a multiple of 4 and <= %d", D, 256 * MAXV);
  BV_REQUIRE(row_stride >= 1 && row_offset >= 0 && row_offset < row_stride, "bv_layernorm_bwd: bad row_stride/offset");
  BV_REQUIRE(mean && rstd && dx, "bv_layernorm_bwd: mean/rstd/dx required");
  const int nv = D <= 768 ? 3 : (D <= 1024 ? 4 : MAXV);
  int grid = (rows + 3) / 4;
  const int max_grid = nv <= 4 ? 768 : 512;   // workgroups per CU: 3 / 2 (ln_probe: 768 5.6 TB/s, 1024 5.5, 1280 4.3)
  if (grid > max_grid) grid = max_grid;
  const size_t shmem = sizeof(float) * 12 * D;
  const int nt = ln_nt_for(rows, D);
#define BV_LN_BWD(F32, NV, NT)                                                                              \
  hipLaunchKernelGGL((ln_bwd_kernel<F32, NV, NT>), dim3(grid), dim3(256), shmem, (hipStream_t)stream, dy, x, \
                     scale, mean, rstd, dres, dx, (bf16*)dx_bf16, dscale, dbias, dx_colsum, rows, D,         \
                     row_stride, row_offset, bias, (bf16*)y_bf16)
#define BV_LN_BWD_NV(F32, NT)                                                                               \
  do { if (nv == 3) BV_LN_BWD(F32, 3, NT); else if (nv == 4) BV_LN_BWD(F32, 4, NT); else BV_LN_BWD(F32, MAXV, NT); } while (0)
  if (dy_is_f32) { if (nt == 3) BV_LN_BWD_NV(true, 3); else BV_LN_BWD_NV(true, 2); }
  else { if (nt == 3) BV_LN_BWD_NV(false, 3); else BV_LN_BWD_NV(false, 2); }
#undef BV_LN_BWD_NV
#undef BV_LN_BWD
  return bv_check_launch("bv_layernorm_bwd");
}
