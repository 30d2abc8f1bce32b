// LayerNorm forward / backward kernels of the product (csrc/layernorm.hip, included as is) on the step's
// shapes, swept over the grid cap and the streaming mode NT (bit 0: non-temporal loads, bit 1: stores):
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I big_vision_amd/csrc tools/probes/ln_probe.hip \
//       big_vision_amd/csrc/c_api.cpp -o tools/probes/ln_probe.out
//
// Backward = the "light context" form of the fp32 stream: bf16 dy, fp32 x / residual gradient in, fp32 dx and
// the re-emitted bf16 forward output out, scale / bias gradients accumulated.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../big_vision_amd/csrc/layernorm.hip"

__global__ void fill(float* d, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned s = (unsigned)(i * 2654435761u) ^ seed;
    s ^= s >> 13; s *= 0x5bd1e995u; s ^= s >> 15;
    d[i] = ((s >> 8) & 0xffff) / 65536.0f - 0.5f;
  }
}
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

template <typename F>
static double time_us(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  hipEventRecord(e0, 0);
  const int iters = 20;
  for (int i = 0; i < iters; ++i) launch();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / iters;
}

int main() {
  const int D = 768, R0 = 401408;
  float *x, *dres, *dx, *sc, *bi, *mean, *rstd, *dsc, *dbi;
  bf16 *y, *dy;
  CK(hipMalloc(&x, (size_t)R0 * D * 4)); CK(hipMalloc(&dres, (size_t)R0 * D * 4)); CK(hipMalloc(&dx, (size_t)R0 * D * 4));
  CK(hipMalloc(&y, (size_t)R0 * D * 2)); CK(hipMalloc(&dy, (size_t)R0 * D * 2));
  CK(hipMalloc(&sc, D * 4)); CK(hipMalloc(&bi, D * 4)); CK(hipMalloc(&dsc, D * 4)); CK(hipMalloc(&dbi, D * 4));
  CK(hipMalloc(&mean, R0 * 4)); CK(hipMalloc(&rstd, R0 * 4));
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, x, (size_t)R0 * D, 1u);
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, dres, (size_t)R0 * D, 2u);
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (float*)dy, (size_t)R0 * D / 2, 3u);
  hipLaunchKernelGGL(fill, dim3(4), dim3(256), 0, 0, sc, (size_t)D, 4u);
  hipLaunchKernelGGL(fill, dim3(4), dim3(256), 0, 0, bi, (size_t)D, 5u);
  CK(hipMemset(dsc, 0, D * 4)); CK(hipMemset(dbi, 0, D * 4));
  CK(hipDeviceSynchronize());
  const int shapes[4] = {401408, 131072, 100352, 32768};
  printf("%-4s %7s %6s   %s\n", "", "rows", "grid", "us / TB/s at NT = 0 | 1 | 2 | 3");
  for (int s = 0; s < 4; ++s) {
    const int rows = shapes[s];
    const int fcaps[5] = {1024, 2048, 4096, 8192, 1 << 20};
    for (int k = 0; k < 5; ++k) {
      int grid = (rows + 7) / 8;
      if (grid > fcaps[k]) grid = fcaps[k];
      printf("fwd  %7d %6d ", rows, grid);
#define FWD(NT) { const double us = time_us([&] { hipLaunchKernelGGL((ln_fwd_kernel<3, NT>), dim3(grid), dim3(256), 0, 0, x, sc, bi, y, \
      (float*)nullptr, mean, rstd, rows, D, 1L, 0L, 1e-6f); }); printf(" | %7.1f %5.2f", us, rows * (D * 6.0 + 8) / us * 1e-6); }
      FWD(0) FWD(1) FWD(2) FWD(3)
      printf("\n");
    }
    const int bcaps[5] = {512, 768, 1024, 1280, 2048};
    for (int k = 0; k < 5; ++k) {
      int grid = (rows + 3) / 4;
      if (grid > bcaps[k]) grid = bcaps[k];
      printf("bwd  %7d %6d ", rows, grid);
#define BWD(NT) { const double us = time_us([&] { hipLaunchKernelGGL((ln_bwd_kernel<false, 3, NT>), dim3(grid), dim3(256),  \
      sizeof(float) * 12 * D, 0, (const void*)dy, x, sc, mean, rstd, dres, dx, (bf16*)nullptr, dsc, dbi, (float*)nullptr, rows, D, \
      1L, 0L, bi, y); }); printf(" | %7.1f %5.2f", us, rows * (D * 16.0 + 8) / us * 1e-6); }
      BWD(0) BWD(1) BWD(2) BWD(3)
      printf("\n");
    }
  }
  CK(hipDeviceSynchronize());
  return 0;
}
