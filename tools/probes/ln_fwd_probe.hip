// LayerNorm-forward streaming probe (standalone): rows per wave and trip, scale / bias hoisting, non-temporal
// accesses and the grid cap, on the step's two shapes (401 408 x 768 image rows, 131 072 x 768 text rows;
// fp32 in, bf16 out + mean / rstd).  Prints the time and the algorithmic HBM rate of every variant.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/ln_fwd_probe.hip -o tools/probes/ln_fwd_probe.out
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../big_vision_amd/csrc/bv_common.h"

template <int NV, int R, bool HOIST, bool NT>
__global__ __launch_bounds__(256) void ln_fwd_v(const float* __restrict__ x, const float* __restrict__ scale,
                                                const float* __restrict__ bias, bf16* __restrict__ y_bf,
                                                float* __restrict__ mean_o, float* __restrict__ rstd_o, int rows,
                                                int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  const float inv_d = 1.0f / (float)D;
  float4 gs[HOIST ? NV : 1], bs[HOIST ? NV : 1];
  if (HOIST) {
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int c = lane * 4 + it * 256;
      if (c < D) {
        gs[it] = *reinterpret_cast<const float4*>(scale + c);
        bs[it] = *reinterpret_cast<const float4*>(bias + c);
      }
    }
  }
  for (int r0 = wave_global; r0 < rows; r0 += R * nwaves) {
    float4 v[R][NV];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int r = r0 + j * nwaves;
      if (r >= rows) continue;
      const float* xr = x + (long)r * D;
#pragma unroll
      for (int it = 0; it < NV; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < D) {
          if (NT) {
            typedef float f4v __attribute__((ext_vector_type(4)));
            const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(xr + c));
            v[j][it] = make_float4(t.x, t.y, t.z, t.w);
          } else {
            v[j][it] = *reinterpret_cast<const float4*>(xr + c);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
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
      const float mean = wave_sum(s) * inv_d;
      const float var = fmaxf(wave_sum(ss) * inv_d - mean * mean, 0.f);
      const float rstd = rsqrtf(var + eps);
      if (lane == 0) {
        mean_o[r] = mean;
        rstd_o[r] = rstd;
      }
#pragma unroll
      for (int it = 0; it < NV; ++it) {
        const int c = lane * 4 + it * 256;
        if (c < D) {
          const float4 g = HOIST ? gs[it] : *reinterpret_cast<const float4*>(scale + c);
          const float4 b = HOIST ? bs[it] : *reinterpret_cast<const float4*>(bias + c);
          float4 o;
          o.x = (v[j][it].x - mean) * rstd * g.x + b.x;
          o.y = (v[j][it].y - mean) * rstd * g.y + b.y;
          o.z = (v[j][it].z - mean) * rstd * g.z + b.z;
          o.w = (v[j][it].w - mean) * rstd * g.w + b.w;
          uint2 p;
          p.x = pack_bf2(o.x, o.y);
          p.y = pack_bf2(o.z, o.w);
          typedef unsigned u2v __attribute__((ext_vector_type(2)));
          if (NT) {
            u2v t; t.x = p.x; t.y = p.y;
            __builtin_nontemporal_store(t, reinterpret_cast<u2v*>(y_bf + (long)r * D + c));
          } else {
            *reinterpret_cast<uint2*>(y_bf + (long)r * D + c) = p;
          }
        }
      }
    }
  }
}

__global__ void fill(float* d, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned s = (unsigned)(i * 2654435761u);
    s ^= s >> 13; s *= 0x5bd1e995u; s ^= s >> 15;
    d[i] = ((s >> 8) & 0xffff) / 65536.0f - 0.5f;
  }
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

template <int R, bool HOIST, bool NT>
static int run(const char* tag, const float* x, const float* sc, const float* bi, bf16* y, float* mean, float* rstd,
               int rows, int D, int cap) {
  int grid = (rows + 4 * R - 1) / (4 * R);
  if (grid > cap) grid = cap;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((ln_fwd_v<3, R, HOIST, NT>), dim3(grid), dim3(256), 0, 0, x, sc, bi, y, mean, rstd, rows, D, 1e-6f);
  CK(hipEventRecord(e0, 0));
  const int iters = 20;
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((ln_fwd_v<3, R, HOIST, NT>), dim3(grid), dim3(256), 0, 0, x, sc, bi, y, mean, rstd, rows, D, 1e-6f);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters;
  const double bytes = (double)rows * (D * 6.0 + 8.0);
  printf("%-34s rows %7d grid %5d  %8.1f us  %6.2f TB/s\n", tag, rows, grid, us, bytes / us * 1e-6);
  return 0;
}

int main() {
  const int D = 768;
  const int shapes[2] = {401408, 131072};
  float *x, *sc, *bi, *mean, *rstd;
  bf16* y;
  CK(hipMalloc(&x, (size_t)shapes[0] * D * 4));
  CK(hipMalloc(&y, (size_t)shapes[0] * D * 2));
  CK(hipMalloc(&sc, D * 4)); CK(hipMalloc(&bi, D * 4));
  CK(hipMalloc(&mean, shapes[0] * 4)); CK(hipMalloc(&rstd, shapes[0] * 4));
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, x, (size_t)shapes[0] * D);
  hipLaunchKernelGGL(fill, dim3(4), dim3(256), 0, 0, sc, (size_t)D);
  hipLaunchKernelGGL(fill, dim3(4), dim3(256), 0, 0, bi, (size_t)D);
  CK(hipDeviceSynchronize());
  for (int s = 0; s < 2; ++s) {
    const int rows = shapes[s];
#define RUN(R, H, N, cap) if (run<R, H, N>("R=" #R " hoist=" #H " nt=" #N " cap=" #cap, x, sc, bi, y, mean, rstd, rows, D, cap)) return 1
    RUN(1, false, false, 4096);      // the product kernel of rounds 1-3
    RUN(1, false, false, 2048);
    RUN(1, false, false, 16384);
    RUN(1, false, false, 1 << 20);   // one trip per wave
    RUN(1, true, false, 4096);
    RUN(1, false, true, 4096);
    RUN(2, false, false, 2048);
    RUN(2, true, false, 2048);
    RUN(2, true, false, 1024);
    RUN(2, true, false, 4096);
    RUN(2, true, true, 2048);
    RUN(2, false, true, 2048);
    RUN(4, false, false, 1024);
    RUN(4, false, false, 2048);
    RUN(4, true, false, 1024);
    RUN(4, false, true, 1024);
  }
  return 0;
}
