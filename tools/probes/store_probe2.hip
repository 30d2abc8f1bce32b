// Store-path probe 2: per-CU store throughput of 128 KiB tiles for different row patterns.
// Each workgroup (512 threads) writes `reps` tiles of 256 rows x 512 B (bf16 256x256), 16 dwordx4
// stores per lane per tile.  PAT = bytes contiguous per row per wave-instruction:
//   64: 16 rows x 64 B (the GEMM epilogue today), 128: 8 rows x 128 B, 256: 4 rows x 256 B,
//   512: 2 rows x 512 B, 1024: lane-linear 1 KiB.   NT = nontemporal stores.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u4;
template <int PAT, bool NT>
__global__ __launch_bounds__(512) void k(u4* out, long* cyc, int reps, long tile_stride_u4, int ld_u4 = 32) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
  constexpr int LPR = PAT / 16;          // lanes per row
  constexpr int RPI = 64 / LPR;          // rows per instruction
  const int lrow = lane / LPR, lcol = lane % LPR;
  long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    // tile: 256 rows x 32 u4.  ld_u4 == 32: tiles are contiguous 128 KiB blocks; otherwise the tiles sit
    // in a row-major [M][ld_u4] matrix (the GEMM's C: row stride N*2 bytes), 9 tiles per row of tiles
    const long tix = (long)blockIdx.x * reps + r;
    u4* base = ld_u4 == 32 ? out + tix * tile_stride_u4 : out + (tix / 9) * 256 * ld_u4 + (tix % 9) * 32;
    // wave owns 128 rows x 8 u4 (128 B) when PAT<=128 [2x4 wave grid]; for wider patterns the
    // wave owns 32 rows x 32 u4 (full 512-B rows) [8x1 wave grid].
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      int row, col;
      if (PAT <= 128) {
        const int wr = wave >> 2, wc = wave & 3;
        // 128 rows x 8 u4 per wave = 1024 u4 = 16 instr x 64 lanes
        const int chunks_per_row = 8 / LPR;              // instr needed per row group
        const int rg = s / chunks_per_row, cc = s % chunks_per_row;
        row = wr * 128 + rg * RPI + lrow;
        col = wc * 8 + cc * LPR + lcol;
      } else {
        const int chunks_per_row = 32 / LPR;
        const int rg = s / chunks_per_row, cc = s % chunks_per_row;
        row = wave * 32 + rg * RPI + lrow;
        col = cc * LPR + lcol;
      }
      u4* p = base + (long)row * ld_u4 + col;
      if (NT) __builtin_nontemporal_store(v, p); else *p = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int PAT, bool NT>
void run(u4* out, long* cyc, const char* name, int ld_u4 = 32) {
  const int reps = 16;
  for (int nb : {1, 32, 256}) {
    hipLaunchKernelGGL((k<PAT, NT>), dim3(nb), dim3(512), 0, 0, out, cyc, reps, 8192L, ld_u4);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<PAT, NT>), dim3(nb), dim3(512), 0, 0, out, cyc, reps, 8192L, ld_u4);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long> h(nb);
    hipMemcpy(h.data(), cyc, nb * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (long c : h) avg += c; avg /= nb;
    printf("%-10s blocks %4d: %7.1f us, %.2f us per 128 KiB tile per block, %7.1f GB/s aggregate, %.0f ticks/tile\n", name, nb,
           ms * 1e3, ms * 1e3 / reps, nb * reps * 131072.0 / ms / 1e6, avg / reps);
  }
}
int main() {
  u4* out; long* cyc;
  hipMalloc(&out, (size_t)256 * 16 * 131072 * 2);
  hipMalloc(&cyc, 1024 * 8);
  run<64, false>(out, cyc, "64B");
  run<128, false>(out, cyc, "128B");
  run<256, false>(out, cyc, "256B");
  run<512, false>(out, cyc, "512B");
  run<1024, false>(out, cyc, "1024B");
  run<64, true>(out, cyc, "64B nt");
  run<128, true>(out, cyc, "128B nt");
  run<1024, true>(out, cyc, "1024B nt");
  // rows strided as in C[M][2304] bf16 (4608 B per row)
  run<64, false>(out, cyc, "64B ld288", 288);
  run<128, false>(out, cyc, "128B ld288", 288);
  run<256, false>(out, cyc, "256B ld288", 288);
  run<512, false>(out, cyc, "512B ld288", 288);
  return 0;
}
