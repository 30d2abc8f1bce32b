// Probe: sustained rate and shader clock of bare MFMA streams on all 256 CUs - v_mfma_f32_16x16x32_bf16
// (what gemm256 issues) vs v_mfma_f32_32x32x16_bf16, one or two waves per SIMD.  No memory traffic: the
// number is the ceiling the GEMM's main loop is measured against, at the clock the chip actually grants
// an all-MFMA kernel (s_memtime vs the 100 MHz s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma_rate_probe.hip -o tools/probes/mfma_rate_probe.out
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>   // 0: 16x16x32, 8 independent accumulators; 1: 32x32x16, 4 independent accumulators
__global__ __launch_bounds__(512) void mfma_kernel(float* out, long* clk, int iters, float seed) {
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + threadIdx.x * 0.001f + e); b[e] = (__bf16)(seed - e * 0.01f); }
  const long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float r = 0.f;
  if constexpr (SHAPE == 0) {
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[i], 0, 0, 0);
    for (int i = 0; i < 8; ++i) r += c[i][0] + c[i][3];
  } else {
    f32x16 c[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[i], 0, 0, 0);
    for (int i = 0; i < 4; ++i) r += c[i][0] + c[i][15];
  }
  const long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
  if (r == 12345.678f) out[0] = r;
}

template <int SHAPE>
static void run(const char* name, int threads, float* out, long* clk) {
  const int iters = 200000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((mfma_kernel<SHAPE>), dim3(256), dim3(threads), 0, 0, out, clk, 1000, 1.0f);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((mfma_kernel<SHAPE>), dim3(256), dim3(threads), 0, 0, out, clk, iters, 1.0f);
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double per = SHAPE == 0 ? 8 * 2.0 * 16 * 16 * 32 : 4 * 2.0 * 32 * 32 * 16;
  const double fl = per * iters * (threads / 64) * 256.0;
  const double mhz = (double)h[0] / h[1] * 100.0;
  printf("%-34s %d waves/SIMD: %7.1f TFLOP/s at %4.0f MHz = %5.1f %% of the 1024 FLOP/clk/SIMD rate at that clock\n", name,
         threads / 256, fl / ms / 1e9, mhz, 100.0 * (fl / (ms * 1e-3)) / (256.0 * 4 * 1024 * mhz * 1e6));
}

int main() {
  float* out; long* clk;
  (void)hipMalloc(&out, 64); (void)hipMalloc(&clk, 64);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("v_mfma_f32_16x16x32_bf16", 256, out, clk);
    run<0>("v_mfma_f32_16x16x32_bf16", 512, out, clk);
    run<1>("v_mfma_f32_32x32x16_bf16", 256, out, clk);
    run<1>("v_mfma_f32_32x32x16_bf16", 512, out, clk);
  }
  return 0;
}
