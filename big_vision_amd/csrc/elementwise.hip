// HBM-bound helper kernels of the SigLIP/ViT step for gfx950: patchify, token
// embedding gather / scatter-add, column / batch reductions, casts, pooling,
// L2 normalisation.  All accesses are 16-byte vectors, coalesced along the
// fastest axis; reductions pre-reduce per workgroup before fp32 atomics.
// Reference call sites are cited at each entry point (paths relative to
// big_vision/ in the reference tree).
#include "bv_common.h"
#include "bvhip_internal.h"

namespace {

// ---------------------------------------------------------------- patchify --
// out[(i*h*w + py*w + px)][r*P*3 + c*3 + ch] = image[i][py*P + r][px*P + c][ch]
// One thread converts 8 consecutive output elements (they are contiguous in
// the input as well because P*3 % 8 == 0).
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img,
                                                       bf16* __restrict__ out, int n, int Hi,
                                                       int Wi, int P, int h, int w) {
  const int K = P * P * 3;
  const int P3 = P * 3;
  const long total8 = (long)n * h * w * K / 8;
  for (long i8 = (long)blockIdx.x * blockDim.x + threadIdx.x; i8 < total8;
       i8 += (long)gridDim.x * blockDim.x) {
    const long o = i8 * 8;
    const long row = o / K;
    const int within = (int)(o - row * K);
    const int r = within / P3, cc = within - r * P3;
    const int px = (int)(row % w);
    const long t = row / w;
    const int py = (int)(t % h);
    const long i = t / h;
    const float* src = img + ((i * Hi + (long)py * P + r) * Wi + (long)px * P) * 3 + cc;
    const float4 a = *reinterpret_cast<const float4*>(src);
    const float4 b = *reinterpret_cast<const float4*>(src + 4);
    uint4 p;
    p.x = pack_bf2(a.x, a.y);
    p.y = pack_bf2(a.z, a.w);
    p.z = pack_bf2(b.x, b.y);
    p.w = pack_bf2(b.z, b.w);
    *reinterpret_cast<uint4*>(out + o) = p;
  }
}

// generic (any P): one thread per output element
__global__ __launch_bounds__(256) void patchify_scalar_kernel(const float* __restrict__ img,
                                                              bf16* __restrict__ out, int n, int Hi,
                                                              int Wi, int P, int h, int w, int ldo) {
  // ldo >= K: row pitch of the output; columns [K, ldo) are written as zeros (patch sizes whose K is
  // not a multiple of 8, e.g. 14 x 14 x 3 = 588, are padded for the GEMM's 16-byte operand loads)
  const int K = P * P * 3;
  const int P3 = P * 3;
  const long total = (long)n * h * w * ldo;
  for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < total;
       o += (long)gridDim.x * blockDim.x) {
    const long row = o / ldo;
    const int within = (int)(o - row * ldo);
    if (within >= K) { out[o] = f2bf(0.f); continue; }
    const int r = within / P3, cc = within - r * P3;
    const int px = (int)(row % w);
    const long t = row / w;
    const int py = (int)(t % h);
    const long i = t / h;
    out[o] = f2bf(img[((i * Hi + (long)py * P + r) * Wi + (long)px * P) * 3 + cc]);
  }
}

// ---------------------------------------------------------------- embedding --
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int* __restrict__ ids,
                                                        const float* __restrict__ table,
                                                        const float* __restrict__ pos,
                                                        float* __restrict__ x, long rows, int L,
                                                        int D, int vocab) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nw = (long)gridDim.x * 4;
  for (long r = wave; r < rows; r += nw) {
    int id = ids[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const float* src = table + (long)id * D;
    const float* ps = pos + (long)(r % L) * D;
    float* dst = x + r * D;
    for (int c = lane * 4; c < D; c += 256) {
      const float4 a = *reinterpret_cast<const float4*>(src + c);
      const float4 b = *reinterpret_cast<const float4*>(ps + c);
      *reinterpret_cast<float4*>(dst + c) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
  }
}

// dtable[ids[r]] += dx[r].  Token ids repeat heavily (sticky EOS / padding: pp/ops_text.py:143-155
// makes about half of all positions the same id), so plain per-row atomics serialise on a few
// table rows.  Each workgroup takes EB_ROWS consecutive rows, links the rows that share an id
// into chains (LDS), sums every chain in registers and issues ONE atomic add per distinct id,
// column and workgroup.
constexpr int EB_ROWS = 64;
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int* __restrict__ ids,
                                                        const float* __restrict__ dx,
                                                        float* __restrict__ dtable, long rows,
                                                        int D, int vocab) {
  __shared__ int sid[EB_ROWS];
  __shared__ int nxt[EB_ROWS];    // next row of the chunk with the same id (-1: end of chain)
  __shared__ int head[EB_ROWS];   // 1: first row of its chain
  const int tid = threadIdx.x;
  const long r0 = (long)blockIdx.x * EB_ROWS;
  const int nr = (int)min((long)EB_ROWS, rows - r0);
  if (tid < EB_ROWS) {
    int id = tid < nr ? ids[r0 + tid] : -1;
    if (tid < nr) id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    sid[tid] = id;
    nxt[tid] = -1;
  }
  __syncthreads();
  if (tid < nr) {
    const int me = sid[tid];
    int prev = -1;
    for (int j = tid - 1; j >= 0; --j)
      if (sid[j] == me) { prev = j; break; }
    head[tid] = prev < 0;
    if (prev >= 0) nxt[prev] = tid;   // unique writer: a row has at most one successor
  }
  __syncthreads();
  const float* base = dx + r0 * D;
  for (int c = tid * 4; c < D; c += 1024) {
    for (int r = 0; r < nr; ++r) {
      if (!head[r]) continue;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      int q = r;
      while (q >= 0) {   // up to 4 independent row loads in flight per step
        const int q1 = nxt[q], q2 = q1 >= 0 ? nxt[q1] : -1, q3 = q2 >= 0 ? nxt[q2] : -1;
        const float4 a0 = *reinterpret_cast<const float4*>(base + (long)q * D + c);
        float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f), a2 = a1, a3 = a1;
        if (q1 >= 0) a1 = *reinterpret_cast<const float4*>(base + (long)q1 * D + c);
        if (q2 >= 0) a2 = *reinterpret_cast<const float4*>(base + (long)q2 * D + c);
        if (q3 >= 0) a3 = *reinterpret_cast<const float4*>(base + (long)q3 * D + c);
        acc.x += (a0.x + a1.x) + (a2.x + a3.x);
        acc.y += (a0.y + a1.y) + (a2.y + a3.y);
        acc.z += (a0.z + a1.z) + (a2.z + a3.z);
        acc.w += (a0.w + a1.w) + (a2.w + a3.w);
        q = q3 >= 0 ? nxt[q3] : -1;
      }
      float* dst = dtable + (long)sid[r] * D + c;
      unsafeAtomicAdd(dst + 0, acc.x);
      unsafeAtomicAdd(dst + 1, acc.y);
      unsafeAtomicAdd(dst + 2, acc.z);
      unsafeAtomicAdd(dst + 3, acc.w);
    }
  }
}

// ------------------------------------------------------------------ colsum --
// out[c] += sum_r x[r][c].  Workgroup = 64 columns x ROWS_PER_BLOCK rows;
// thread (tx = t&7 -> 8 columns, ty = t>>3 -> row lane).
constexpr int CS_ROWS = 1024;
template <bool F32>
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ x_, long ldx,
                                                     float* __restrict__ out, int rows, int cols) {
  __shared__ float red[32][65];
  const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const int c0 = blockIdx.x * 64 + tx * 8;
  const int r0 = blockIdx.y * CS_ROWS;
  const int r1 = min(rows, r0 + CS_ROWS);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < cols) {
    for (int r = r0 + ty; r < r1; r += 32) {
      if constexpr (F32) {
        const float* p = reinterpret_cast<const float*>(x_) + (long)r * ldx + c0;
        const float4 a = *reinterpret_cast<const float4*>(p);
        const float4 b = *reinterpret_cast<const float4*>(p + 4);
        acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
        acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
      } else {
        const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(x_) + (long)r * ldx + c0);
        acc[0] += bflo(u.x); acc[1] += bfhi(u.x); acc[2] += bflo(u.y); acc[3] += bfhi(u.y);
        acc[4] += bflo(u.z); acc[5] += bfhi(u.z); acc[6] += bflo(u.w); acc[7] += bfhi(u.w);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ty][tx * 8 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) s += red[j][threadIdx.x];
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c < cols) unsafeAtomicAdd(out + c, s);
  }
}

// ---------------------------------------------------------------- batchsum --
// out[j] += sum_i x[i][j], j over L*D (fp32), i over n split across blockIdx.y
__global__ __launch_bounds__(256) void batchsum_kernel(const float* __restrict__ x,
                                                       float* __restrict__ out, int n, long LD,
                                                       int n_per) {
  const long j = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (j >= LD) return;
  const int i0 = blockIdx.y * n_per, i1 = min(n, i0 + n_per);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = i0; i < i1; ++i) {
    const float4 a = *reinterpret_cast<const float4*>(x + (long)i * LD + j);
    s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
  }
  unsafeAtomicAdd(out + j + 0, s.x);
  unsafeAtomicAdd(out + j + 1, s.y);
  unsafeAtomicAdd(out + j + 2, s.z);
  unsafeAtomicAdd(out + j + 3, s.w);
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x,
                                                        bf16* __restrict__ y, long count) {
  const long n8 = count / 8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const float4 a = *reinterpret_cast<const float4*>(x + i * 8);
    const float4 b = *reinterpret_cast<const float4*>(x + i * 8 + 4);
    uint4 p;
    p.x = pack_bf2(a.x, a.y); p.y = pack_bf2(a.z, a.w);
    p.z = pack_bf2(b.x, b.y); p.w = pack_bf2(b.z, b.w);
    *reinterpret_cast<uint4*>(y + i * 8) = p;
  }
  if (blockIdx.x == 0) {
    for (long i = n8 * 8 + threadIdx.x; i < count; i += 256) y[i] = f2bf(x[i]);
  }
}

__global__ __launch_bounds__(256) void concat_cls_kernel(const float* __restrict__ cls,
                                                         const float* __restrict__ x,
                                                         float* __restrict__ y, int n, int L, int D) {
  const long total4 = (long)n * (L + 1) * D / 4;
  const int D4 = D / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const long row = i / D4;
    const int c = (int)(i - row * D4) * 4;
    const long b = row / (L + 1);
    const int l = (int)(row - b * (L + 1));
    float4 v;
    if (l == 0) v = *reinterpret_cast<const float4*>(cls + c);
    else v = *reinterpret_cast<const float4*>(x + (b * L + (l - 1)) * D + c);
    *reinterpret_cast<float4*>(y + row * D + c) = v;
  }
}

__global__ __launch_bounds__(256) void pool_gap_fwd_kernel(const float* __restrict__ x,
                                                           float* __restrict__ y, int n, int L, int D) {
  const int D4 = D / 4;
  const long total = (long)n * D4;
  const float inv = 1.0f / (float)L;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / D4;
    const int c = (int)(i - b * D4) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 0; l < L; ++l) {
      const float4 a = *reinterpret_cast<const float4*>(x + (b * L + l) * D + c);
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    *reinterpret_cast<float4*>(y + b * D + c) = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
  }
}
__global__ __launch_bounds__(256) void pool_gap_bwd_kernel(const float* __restrict__ dy,
                                                           float* __restrict__ dx, int n, int L, int D) {
  const int D4 = D / 4;
  const long total = (long)n * L * D4;
  const float inv = 1.0f / (float)L;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / D4;
    const int c = (int)(i - row * D4) * 4;
    const long b = row / L;
    const float4 a = *reinterpret_cast<const float4*>(dy + b * D + c);
    *reinterpret_cast<float4*>(dx + row * D + c) = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
  }
}

// pool_type "max" / "gmp" of the text tower (text_transformer.py:89-90): y[b][c] = max_l x[b][l][c]; the index of the
// (first) maximum is kept for the backward, which routes dy[b][c] to that one position (ties have measure zero in
// float activations; jnp.max's VJP would split the cotangent between exact ties).
// len (nullable): only the first len[b] positions of sample b take part (NaFlex, naflex_vit.py:267-271: padded positions
// count as finfo.min; a sample without any valid token yields -FLT_MAX at position 0)
__global__ __launch_bounds__(256) void pool_max_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           int* __restrict__ arg, const int* __restrict__ len, int n, int L,
                                                           int D) {
  const long total = (long)n * D;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / D;
    const int c = (int)(i - b * D);
    const int Lb = len ? min(max(len[b], 0), L) : L;
    float best = Lb > 0 ? x[(b * L) * D + c] : -3.402823466e+38f;
    int at = 0;
    for (int l = 1; l < Lb; ++l) {
      const float v = x[(b * L + l) * D + c];
      // a NaN takes the maximum and keeps it (x.max(axis=1) propagates NaN: a diverged tower must stay visible to
      // the trainer's finiteness check); `best == best` stops later values from replacing a NaN already held
      if ((v > best || v != v) && best == best) { best = v; at = l; }
    }
    y[i] = best;
    arg[i] = at;
  }
}
__global__ __launch_bounds__(256) void pool_max_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ arg,
                                                           float* __restrict__ dx, int n, int L, int D) {
  const long total = (long)n * L * D;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / D;
    const int c = (int)(i - row * D);
    const long b = row / L;
    const int l = (int)(row - b * L);
    dx[i] = arg[b * D + c] == l ? dy[b * D + c] : 0.f;
  }
}

// gap over the first len[b] rows only (NaFlex: padding excluded, naflex_vit.py:262-264)
__global__ __launch_bounds__(256) void pool_gap_masked_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  const int* __restrict__ len, int n, int L, int D) {
  const int D4 = D / 4;
  const long total = (long)n * D4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / D4;
    const int c = (int)(i - b * D4) * 4;
    const int lb = min(len[b], L);
    const float inv = 1.0f / (float)lb;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 0; l < lb; ++l) {
      const float4 a = *reinterpret_cast<const float4*>(x + (b * L + l) * D + c);
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    *reinterpret_cast<float4*>(y + b * D + c) = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
  }
}
__global__ __launch_bounds__(256) void pool_gap_masked_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                                  const int* __restrict__ len, int n, int L, int D) {
  const int D4 = D / 4;
  const long total = (long)n * L * D4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long row = i / D4;
    const int c = (int)(i - row * D4) * 4;
    const long b = row / L;
    const int l = (int)(row - b * L), lb = min(len[b], L);
    const float inv = l < lb ? 1.0f / (float)lb : 0.f;
    const float4 a = *reinterpret_cast<const float4*>(dy + b * D + c);
    *reinterpret_cast<float4*>(dx + row * D + c) = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
  }
}

// ------------------------------------------------- NaFlex position embedding --
// models/proj/image_text/naflex_vit.py:38-83 (`_pos_emb_resize`): every example resizes the learned
// [P, P, D] grid to its own patch grid (h_e, w_e) = max coordinate + 1 with
// jax.image.scale_and_translate(method="bilinear", antialias=True) and gathers one embedding per
// token at (yabs, xabs).  Resize + gather are one linear map per token,
//   emb[t] = sum_ij Wy_e[y_t, i] Wx_e[x_t, j] pos[i, j],
// so the kernel below only writes the token weights W[t][i*P + j] (bf16, [n*N][P*P]); the product
// with pos (and, in the backward, W^T dtok) runs on the MFMA GEMMs.  Weights as in
// jax._src.image.scale.compute_weight_mat: sample position s = (y + 0.5) P / h - 0.5, triangle
// kernel of width max(P / h, 1), normalised over the input index, zero when s leaves [-0.5, P - 0.5].
__device__ __forceinline__ float naflex_w(int out_pos, int in_pos, int in_size, int out_size) {
  const float inv_scale = (float)in_size / (float)out_size;
  const float kscale = fmaxf(inv_scale, 1.f);
  const float sf = ((float)out_pos + 0.5f) * inv_scale - 0.5f;
  if (sf < -0.5f || sf > (float)in_size - 0.5f) return 0.f;
  float total = 0.f;
  for (int i = 0; i < in_size; ++i) total += fmaxf(0.f, 1.f - fabsf(sf - (float)i) / kscale);
  const float w = fmaxf(0.f, 1.f - fabsf(sf - (float)in_pos) / kscale);
  return fabsf(total) > 1000.f * 1.1920929e-07f ? w / total : 0.f;
}
__global__ __launch_bounds__(256) void naflex_posw_kernel(const int* __restrict__ yabs, const int* __restrict__ xabs,
                                                          bf16* __restrict__ W, int N, int P) {
  __shared__ int sh[8];
  __shared__ float wy[64], wx[64];
  const int e = blockIdx.x;
  const int* ye = yabs + (long)e * N;
  const int* xe = xabs + (long)e * N;
  int my = 0, mx = 0;
  for (int t = threadIdx.x; t < N; t += 256) { my = max(my, ye[t]); mx = max(mx, xe[t]); }
  my = (int)wave_max((float)my); mx = (int)wave_max((float)mx);
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = my; sh[4 + (threadIdx.x >> 6)] = mx; }
  __syncthreads();
  const int h = max(max(sh[0], sh[1]), max(sh[2], sh[3])) + 1;       // shapes = coords.max(axis=1) + 1
  const int w = max(max(sh[4], sh[5]), max(sh[6], sh[7])) + 1;
  for (int t = 0; t < N; ++t) {
    __syncthreads();
    if (threadIdx.x < P) wy[threadIdx.x] = naflex_w(ye[t], threadIdx.x, P, h);
    else if (threadIdx.x >= 64 && threadIdx.x < 64 + P) wx[threadIdx.x - 64] = naflex_w(xe[t], threadIdx.x - 64, P, w);
    __syncthreads();
    bf16* row = W + ((long)e * N + t) * P * P;
    for (int k = threadIdx.x; k < P * P; k += 256) row[k] = (bf16)(wy[k / P] * wx[k % P]);
  }
}

// ------------------------------------------------------------------ l2norm --
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ z,
                                                         float* __restrict__ zn,
                                                         float* __restrict__ norm, int rows, int D,
                                                         float eps) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* zr = z + (long)r * D;
  float ss = 0.f;
  for (int c = lane; c < D; c += 64) ss += zr[c] * zr[c];
  ss = wave_sum(ss);
  const float s = sqrtf(ss);
  if (lane == 0) norm[r] = s;
  const float inv = 1.0f / (s + eps);
  for (int c = lane; c < D; c += 64) zn[(long)r * D + c] = zr[c] * inv;
}
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ z,
                                                         const float* __restrict__ norm,
                                                         const float* __restrict__ dzn,
                                                         float* __restrict__ dz, int rows, int D,
                                                         float eps) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* zr = z + (long)r * D;
  const float* gr = dzn + (long)r * D;
  const float s = norm[r];
  const float inv = 1.0f / (s + eps);
  float dot = 0.f;  // zn . dzn
  for (int c = lane; c < D; c += 64) dot += zr[c] * inv * gr[c];
  dot = wave_sum(dot);
  const float k = s > 0.f ? dot / (s * (s + eps)) : 0.f;
  for (int c = lane; c < D; c += 64) dz[(long)r * D + c] = gr[c] * inv - zr[c] * k;
}

inline int grid_for(long work_items, int per_block, int cap) {
  long g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// ----------------------------------------------------------- transpose -----
// dst[c][r] = src[r][c], bf16, 64x64 tiles through LDS ([64][66] halfwords:
// the odd word pitch makes the column reads conflict-free).
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const uint16_t* __restrict__ src,
                                                             uint16_t* __restrict__ dst, int rows,
                                                             int cols, long lds, long ldd) {
  __shared__ uint16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = ty + 4 * i;
    if (r0 + r < rows && c0 + tx < cols) tile[r][tx] = src[(long)(r0 + r) * lds + c0 + tx];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = ty + 4 * i;
    if (c0 + c < cols && r0 + tx < rows) dst[(long)(c0 + c) * ldd + r0 + tx] = tile[tx][c];
  }
}

// Table-driven form: one launch transposes every listed matrix (the ~100 projection kernels of the two
// towers after an optimizer step: one launch instead of one per weight).  A workgroup finds its matrix by
// bisection over the tile prefix `tile0`.
__global__ __launch_bounds__(256) void transpose_bf16_batched_kernel(const bv_tr_leaf* __restrict__ tab, int nleaves) {
  __shared__ uint16_t tile[64][66];
  int lo = 0, hi = nleaves - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const bv_tr_leaf e = tab[lo];
  const int t = blockIdx.x - e.tile0;
  const int r0 = (t / e.tiles_x) * 64, c0 = (t % e.tiles_x) * 64;
  const uint16_t* src = (const uint16_t*)e.src;
  uint16_t* dst = (uint16_t*)e.dst;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = ty + 4 * i;
    if (r0 + r < e.rows && c0 + tx < e.cols) tile[r][tx] = src[(long)(r0 + r) * e.lds + c0 + tx];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = ty + 4 * i;
    if (c0 + c < e.cols && r0 + tx < e.rows) dst[(long)(c0 + c) * e.ldd + r0 + tx] = tile[tx][c];
  }
}

__global__ __launch_bounds__(256) void tanh_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long count) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) y[i] = tanhf(x[i]);
}
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                       float* __restrict__ dx, long count) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256)
    dx[i] = dy[i] * (1.f - y[i] * y[i]);
}
__global__ __launch_bounds__(256) void mixup_kernel(const float* __restrict__ x, float* __restrict__ out, float a,
                                                    int n, long row_elems) {
  const long total = (long)n * row_elems;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / row_elems;
    const long j = r == 0 ? i + (long)(n - 1) * row_elems : i - row_elems;   // row r-1 (mod n): jnp.roll(x, 1, 0)
    out[i] = a * x[i] + (1.f - a) * x[j];
  }
}

}  // namespace

// bf16 transpose: dst[cols][rows] = src[rows][cols]^T (row strides lds / ldd
// elements).  Keeps the [out][in] image of the Flax (in,out) kernels that the
// forward projections (models/vit.py:72,77,93-98) consume on the k-major GEMM path.
extern "C" int bv_transpose_bf16(const void* src, void* dst, int rows, int cols, long lds, long ldd,
                                 void* stream) {
  BV_REQUIRE(rows > 0 && cols > 0 && lds >= cols && ldd >= rows,
             "bv_transpose_bf16: bad shape rows=%d cols=%d lds=%ld ldd=%ld", rows, cols, lds, ldd);
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  BV_REQUIRE(grid.y <= 65535, "bv_transpose_bf16: too many rows");
  hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)src, (uint16_t*)dst, rows, cols, lds, ldd);
  return bv_check_launch("bv_transpose_bf16");
}

extern "C" int bv_transpose_bf16_batched(const bv_tr_leaf* leaves, int nleaves, int total_tiles, void* stream) {
  BV_REQUIRE(leaves && nleaves > 0 && total_tiles > 0, "bv_transpose_bf16_batched: empty table (nleaves=%d tiles=%d)",
             nleaves, total_tiles);
  hipLaunchKernelGGL(transpose_bf16_batched_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, leaves,
                     nleaves);
  return bv_check_launch("bv_transpose_bf16_batched");
}

// models/vit.py:212-217 — im2col of the stride-P VALID patch conv.
extern "C" int bv_patchify_ld(const float* image, void* patches, int n, int Hi, int Wi, int P, int ldo,
                              void* stream) {
  BV_REQUIRE(n > 0 && Hi >= P && Wi >= P && P > 0, "bv_patchify: bad shape n=%d Hi=%d Wi=%d P=%d", n, Hi, Wi, P);
  BV_REQUIRE(ldo >= P * P * 3, "bv_patchify: row pitch %d < P*P*3 = %d", ldo, P * P * 3);
  const int h = Hi / P, w = Wi / P;
  const long total = (long)n * h * w * P * P * 3;
  if (ldo == P * P * 3 && (P * 3) % 8 == 0 && (Wi * 3) % 4 == 0 && ((uintptr_t)image % 16 == 0)) {
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(total / 8, 256, 8192)), dim3(256), 0,
                       (hipStream_t)stream, image, (bf16*)patches, n, Hi, Wi, P, h, w);
  } else {
    hipLaunchKernelGGL(patchify_scalar_kernel, dim3(grid_for((long)n * h * w * ldo, 256, 8192)), dim3(256), 0,
                       (hipStream_t)stream, image, (bf16*)patches, n, Hi, Wi, P, h, w, ldo);
  }
  return bv_check_launch("bv_patchify");
}
extern "C" int bv_patchify(const float* image, void* patches, int n, int Hi, int Wi, int P, void* stream) {
  return bv_patchify_ld(image, patches, n, Hi, Wi, P, P * P * 3, stream);
}

// models/proj/image_text/text_transformer.py:63-70
extern "C" int bv_embed_fwd(const int* ids, const float* table, const float* pos, float* x, int n,
                            int L, int D, int vocab, void* stream) {
  BV_REQUIRE(n > 0 && L > 0 && D % 4 == 0 && vocab > 0, "bv_embed_fwd: bad shape n=%d L=%d D=%d", n, L, D);
  const long rows = (long)n * L;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid_for(rows, 4, 4096)), dim3(256), 0, (hipStream_t)stream,
                     ids, table, pos, x, rows, L, D, vocab);
  return bv_check_launch("bv_embed_fwd");
}
extern "C" int bv_embed_bwd(const int* ids, const float* dx, float* dtable, int rows, int D, int vocab,
                            void* stream) {
  BV_REQUIRE(rows > 0 && D % 4 == 0 && vocab > 0, "bv_embed_bwd: bad shape rows=%d D=%d", rows, D);
  hipLaunchKernelGGL(embed_bwd_kernel, dim3((rows + EB_ROWS - 1) / EB_ROWS), dim3(256), 0, (hipStream_t)stream,
                     ids, dx, dtable, (long)rows, D, vocab);
  return bv_check_launch("bv_embed_bwd");
}

extern "C" int bv_colsum(const void* x, int x_is_f32, long ldx, float* out, int rows, int cols,
                         void* stream) {
  BV_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && ldx % 8 == 0, "bv_colsum: rows=%d cols=%d ldx=%ld (cols, ldx %% 8)", rows, cols, ldx);
  dim3 grid((cols + 63) / 64, (rows + CS_ROWS - 1) / CS_ROWS);
  if (x_is_f32) hipLaunchKernelGGL(colsum_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, out, rows, cols);
  else hipLaunchKernelGGL(colsum_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, out, rows, cols);
  return bv_check_launch("bv_colsum");
}

extern "C" int bv_batchsum(const float* x, float* out, int n, int L, int D, void* stream) {
  const long LD = (long)L * D;
  BV_REQUIRE(n > 0 && LD > 0 && LD % 4 == 0, "bv_batchsum: bad shape n=%d L=%d D=%d", n, L, D);
  const int ny = n >= 64 ? 8 : 1;
  const int n_per = (n + ny - 1) / ny;
  dim3 grid((unsigned)((LD / 4 + 255) / 256), ny);
  hipLaunchKernelGGL(batchsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, out, n, LD, n_per);
  return bv_check_launch("bv_batchsum");
}

// bf16 -> fp32 (the boundary of a bf16 residual stream: embedding / posemb gradients are summed in fp32)
__global__ __launch_bounds__(256) void cast_f32_kernel(const bf16* __restrict__ x, float* __restrict__ y, long count) {
  const long n8 = count / 8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const uint4 u = *reinterpret_cast<const uint4*>(x + i * 8);
    *reinterpret_cast<float4*>(y + i * 8) = make_float4(bflo(u.x), bfhi(u.x), bflo(u.y), bfhi(u.y));
    *reinterpret_cast<float4*>(y + i * 8 + 4) = make_float4(bflo(u.z), bfhi(u.z), bflo(u.w), bfhi(u.w));
  }
  if (blockIdx.x == 0 && threadIdx.x < count - n8 * 8) y[n8 * 8 + threadIdx.x] = bf2f(x[n8 * 8 + threadIdx.x]);
}
extern "C" int bv_cast_f32(const void* x_bf16, float* y, long count, void* stream) {
  BV_REQUIRE(count > 0, "bv_cast_f32: empty");
  BV_REQUIRE((uintptr_t)x_bf16 % 16 == 0 && (uintptr_t)y % 16 == 0, "bv_cast_f32: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(cast_f32_kernel, dim3(grid_for(count / 8 + 1, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)x_bf16, y, count);
  return bv_check_launch("bv_cast_f32");
}

extern "C" int bv_cast_bf16(const float* x, void* y, long count, void* stream) {
  BV_REQUIRE(count > 0, "bv_cast_bf16: empty");
  BV_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0, "bv_cast_bf16: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(count / 8 + 1, 256, 8192)), dim3(256), 0,
                     (hipStream_t)stream, x, (bf16*)y, count);
  return bv_check_launch("bv_cast_bf16");
}

// pre_logits tanh (models/vit.py:259-262) and its backward dx = dy (1 - y^2)
extern "C" int bv_tanh_fwd(const float* x, float* y, long count, void* stream) {
  BV_REQUIRE(count > 0, "bv_tanh_fwd: empty");
  hipLaunchKernelGGL(tanh_fwd_kernel, dim3(grid_for(count, 256, 4096)), dim3(256), 0, (hipStream_t)stream, x, y, count);
  return bv_check_launch("bv_tanh_fwd");
}
extern "C" int bv_tanh_bwd(const float* y, const float* dy, float* dx, long count, void* stream) {
  BV_REQUIRE(count > 0, "bv_tanh_bwd: empty");
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3(grid_for(count, 256, 4096)), dim3(256), 0, (hipStream_t)stream, y, dy, dx, count);
  return bv_check_launch("bv_tanh_bwd");
}

// utils.py:1146-1154 (get_mixup): out[i] = a x[i] + (1 - a) x[i - 1 mod n] over rows of row_elems floats
extern "C" int bv_mixup(const float* x, float* out, float a, int n, long row_elems, void* stream) {
  BV_REQUIRE(n > 0 && row_elems > 0 && x != out, "bv_mixup: bad arguments (in-place is not supported)");
  hipLaunchKernelGGL(mixup_kernel, dim3(grid_for((long)n * row_elems, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                     x, out, a, n, row_elems);
  return bv_check_launch("bv_mixup");
}

// models/vit.py:223-225
extern "C" int bv_concat_cls(const float* cls, const float* x, float* y, int n, int L, int D, void* stream) {
  BV_REQUIRE(n > 0 && L > 0 && D % 4 == 0, "bv_concat_cls: bad shape");
  hipLaunchKernelGGL(concat_cls_kernel, dim3(grid_for((long)n * (L + 1) * D / 4, 256, 8192)), dim3(256), 0,
                     (hipStream_t)stream, cls, x, y, n, L, D);
  return bv_check_launch("bv_concat_cls");
}

// models/vit.py:246
extern "C" int bv_pool_gap_fwd(const float* x, float* y, int n, int L, int D, void* stream) {
  BV_REQUIRE(n > 0 && L > 0 && D % 4 == 0, "bv_pool_gap_fwd: bad shape");
  hipLaunchKernelGGL(pool_gap_fwd_kernel, dim3(grid_for((long)n * D / 4, 256, 8192)), dim3(256), 0,
                     (hipStream_t)stream, x, y, n, L, D);
  return bv_check_launch("bv_pool_gap_fwd");
}
extern "C" int bv_pool_gap_bwd(const float* dy, float* dx, int n, int L, int D, void* stream) {
  BV_REQUIRE(n > 0 && L > 0 && D % 4 == 0, "bv_pool_gap_bwd: bad shape");
  hipLaunchKernelGGL(pool_gap_bwd_kernel, dim3(grid_for((long)n * L * D / 4, 256, 8192)), dim3(256), 0,
                     (hipStream_t)stream, dy, dx, n, L, D);
  return bv_check_launch("bv_pool_gap_bwd");
}

// models/proj/image_text/text_transformer.py:89-90
extern "C" int bv_pool_max_fwd(const float* x, float* y, int* argmax, int n, int L, int D, void* stream) {
  BV_REQUIRE(n > 0 && L > 0 && D > 0 && argmax != nullptr, "bv_pool_max_fwd: bad arguments");
  hipLaunchKernelGGL(pool_max_fwd_kernel, dim3(grid_for((long)n * D, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, y,
                     argmax, (const int*)nullptr, n, L, D);
  return bv_check_launch("bv_pool_max_fwd");
}
extern "C" int bv_pool_max_masked_fwd(const float* x, float* y, int* argmax, const int* len, int n, int L, int D, void* stream) {
  BV_REQUIRE(n > 0 && L > 0 && D > 0 && argmax != nullptr && len != nullptr, "bv_pool_max_masked_fwd: bad arguments");
  hipLaunchKernelGGL(pool_max_fwd_kernel, dim3(grid_for((long)n * D, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, y,
                     argmax, len, n, L, D);
  return bv_check_launch("bv_pool_max_masked_fwd");
}
extern "C" int bv_pool_max_bwd(const float* dy, const int* argmax, float* dx, int n, int L, int D, void* stream) {
  BV_REQUIRE(n > 0 && L > 0 && D > 0 && argmax != nullptr, "bv_pool_max_bwd: bad arguments");
  hipLaunchKernelGGL(pool_max_bwd_kernel, dim3(grid_for((long)n * L * D, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                     dy, argmax, dx, n, L, D);
  return bv_check_launch("bv_pool_max_bwd");
}

extern "C" int bv_pool_gap_masked_fwd(const float* x, float* y, const int* len, int n, int L, int D, void* stream) {
  BV_REQUIRE(n > 0 && L > 0 && D % 4 == 0 && len != nullptr, "bv_pool_gap_masked_fwd: bad arguments");
  hipLaunchKernelGGL(pool_gap_masked_fwd_kernel, dim3(grid_for((long)n * D / 4, 256, 8192)), dim3(256), 0,
                     (hipStream_t)stream, x, y, len, n, L, D);
  return bv_check_launch("bv_pool_gap_masked_fwd");
}
extern "C" int bv_pool_gap_masked_bwd(const float* dy, float* dx, const int* len, int n, int L, int D, void* stream) {
  BV_REQUIRE(n > 0 && L > 0 && D % 4 == 0 && len != nullptr, "bv_pool_gap_masked_bwd: bad arguments");
  hipLaunchKernelGGL(pool_gap_masked_bwd_kernel, dim3(grid_for((long)n * L * D / 4, 256, 8192)), dim3(256), 0,
                     (hipStream_t)stream, dy, dx, len, n, L, D);
  return bv_check_launch("bv_pool_gap_masked_bwd");
}
// naflex_vit.py:38-83; W [n*N][P*P] bf16, P <= 64
extern "C" int bv_naflex_posemb_weights(const int* yabs, const int* xabs, void* W, int n, int N, int P, void* stream) {
  BV_REQUIRE(n > 0 && N > 0 && P > 0 && P <= 64, "bv_naflex_posemb_weights: bad shape n=%d N=%d P=%d", n, N, P);
  hipLaunchKernelGGL(naflex_posw_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, yabs, xabs, (bf16*)W, N, P);
  return bv_check_launch("bv_naflex_posemb_weights");
}

// models/proj/image_text/two_towers.py:60-61,73-74
extern "C" int bv_l2norm_fwd(const float* z, float* zn, float* norm, int rows, int D, float eps, void* stream) {
  BV_REQUIRE(rows > 0 && D > 0, "bv_l2norm_fwd: bad shape");
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, z, zn, norm, rows, D, eps);
  return bv_check_launch("bv_l2norm_fwd");
}
extern "C" int bv_l2norm_bwd(const float* z, const float* norm, const float* dzn, float* dz, int rows,
                             int D, float eps, void* stream) {
  BV_REQUIRE(rows > 0 && D > 0, "bv_l2norm_bwd: bad shape");
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, z, norm, dzn, dz, rows, D, eps);
  return bv_check_launch("bv_l2norm_bwd");
}
