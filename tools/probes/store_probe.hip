// Store-path probe: every workgroup (512 threads) writes 128 KiB (16 x dwordx4 per lane, row
// pattern of the GEMM epilogue) then waits vmcnt(0); cycles per workgroup vs number of
// active workgroups.  hipcc --offload-arch=gfx950 -O3 tools/probes/store_probe.hip -o /tmp/sp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ __launch_bounds__(512) void k(uint4* out, long* cyc, int ldc_u4, int reps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3, 4);
  long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    // tile r of this block: rows wave*... 256 rows x 256 bf16 cols = 32 uint4 per row
    uint4* base = out + ((long)(blockIdx.x * reps + r) * 256) * ldc_u4;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int row = (wave >> 2) * 128 + i * 16 + lr;
        const int col_u4 = (wave & 3) * 8 + hh * 4 + lg;     // 64 B contiguous per row per instr
        base[(long)row * ldc_u4 + col_u4] = v;
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  const int reps = 8, ldc_u4 = 32;
  uint4* out; long* cyc;
  hipMalloc(&out, (size_t)1024 * reps * 256 * ldc_u4 * 16);
  hipMalloc(&cyc, 1024 * 8);
  for (int nb : {1, 8, 32, 64, 128, 256, 512}) {
    hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, out, cyc, ldc_u4, reps);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, out, cyc, ldc_u4, reps);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long> h(nb);
    hipMemcpy(h.data(), cyc, nb * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (long c : h) avg += c; avg /= nb;
    printf("blocks %4d: %.1f us total, %.0f cycles(100MHz ticks?) per block for %d x 128 KiB -> %.2f us per 128 KiB, %.1f GB/s aggregate\n",
           nb, ms * 1e3, avg, reps, ms * 1e3 / reps / ((nb + 255) / 256), nb * reps * 131072.0 / ms / 1e6);
  }
  return 0;
}
