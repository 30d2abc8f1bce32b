// Where does gemm_pair.hip (two 4-wave workgroups per CU, 256x128 tiles, K-tiles of 32) lose against gemm256 (one
// 8-wave workgroup, 256x256 tile, K-tiles of 64)?  Ablations of its main loop on the step's plain shapes, timing only
// (PROBE bits of gemm_pair_kernel: 1 = no global->LDS requests after the prologue, 2 = fragment reads only for the
// first K-tile, 4 = no epilogue), once with 512 workgroups (two per CU) and once with 256 (one per CU), next to
// gemm256 and ITS no-DMA ablation (PROBE 3 of gemm256_kernel) on the same shape.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I big_vision_amd/csrc -I include tools/probes/gemm_pair_probe.hip \
//         big_vision_amd/csrc/c_api.cpp -o tools/probes/gemm_pair_probe.out && tools/probes/gemm_pair_probe.out
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define BV_GEMM256_PROBES   // compiles the PROBE != 0 ablation paths of gemm256_kernel (absent from the library build)
#include "../../big_vision_amd/csrc/gemm256.hip"
#include "gemm_pair.hip"
#include "probe_ctx.h"

__global__ void fill_bf16(unsigned short* d, size_t n, unsigned seed, float scale) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned s = (unsigned)(i * 2654435761u) ^ seed;
    s ^= s >> 13; s *= 0x5bd1e995u; s ^= s >> 15;
    const float f = (((s >> 8) & 0xffff) / 65536.0f * 2.f - 1.f) * scale;
    d[i] = (unsigned short)(__float_as_uint(f) >> 16);
  }
}

namespace {
template <int PROBE>
void run_pair(const PairParams& p, int grid) {
  auto kern = gemm_pair_kernel<BV_EPI_NONE, false, PROBE>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, P_SMEM);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), P_SMEM, 0, p);
}
template <int PROBE>
void run_256(const G256Params& p, int grid) {
  hipLaunchKernelGGL((gemm256_kernel<true, PROBE, BV_EPI_NONE, false>), dim3(grid), dim3(512), 0, 0, p);
}
}  // namespace

int main() {
  struct Shape { const char* name; int M, N, K; } shapes[] = {
      {"qkv   401408x2304x768", 401408, 2304, 768}, {"dx fc1 401408x768x3072", 401408, 768, 3072},
      {"dx out 401408x768x768", 401408, 768, 768}, {"long-K 8192x2048x16384", 8192, 2048, 16384}};
  unsigned short *a, *b; void* c;
  (void)hipMalloc(&a, (size_t)401408 * 3072 * 2); (void)hipMalloc(&b, (size_t)3072 * 16384 * 2);
  (void)hipMalloc(&c, (size_t)401408 * 2304 * 2);
  fill_bf16<<<2048, 256>>>(a, (size_t)401408 * 3072, 1u, 1.0f);
  fill_bf16<<<2048, 256>>>(b, (size_t)3072 * 16384, 2u, 0.05f);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto time_us = [&](auto fn) {
    fn(); fn();
    (void)hipDeviceSynchronize();
    const int it = 4;
    float t;
    (void)hipEventRecord(e0, 0); for (int i = 0; i < it; ++i) fn(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&t, e0, e1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); exit(1); }
    return t * 1e3f / it;
  };
  for (auto& s : shapes) {
    PairParams p{};
    p.A = (const bf16*)a; p.B = (const bf16*)b; p.C = c; p.lda = s.K; p.ldb = s.K; p.ldc = s.N;
    p.M = s.M; p.N = s.N; p.K = s.K; p.aux_rows = 1; p.tiles_n = s.N / 128; p.ntiles = (s.M / 256) * p.tiles_n; p.alpha = 1.f;
    G256Params q{};
    q.A = (const bf16*)a; q.B = (const bf16*)b; q.C = c; q.lda = s.K; q.ldb = s.K; q.ldc = s.N;
    q.M = s.M; q.N = s.N; q.K = s.K; q.aux_rows = 1; q.tiles_n = s.N / 256; q.ntiles = (s.M / 256) * q.tiles_n;
    q.splits = 1; q.ktiles_per_split = s.K / 64; q.alpha = 1.f;
    const double tf = 2.0 * s.M * s.N * s.K / 1e6;   // TFLOP/s = tf / us
    const float g0 = time_us([&] { run_256<0>(q, 256); });
    const float g3 = time_us([&] { run_256<3>(q, 256); });
    const float g5 = time_us([&] { run_256<5>(q, 256); });
    printf("%-26s gemm256 %8.1f us %6.0f TF | no DMA %6.0f TF | no stores %6.0f TF\n", s.name, g0, tf / g0, tf / g3, tf / g5);
    for (int grid : {512, 256}) {
      const float t0 = time_us([&] { run_pair<0>(p, grid); });
      const float t1 = time_us([&] { run_pair<1>(p, grid); });
      const float t2 = time_us([&] { run_pair<2>(p, grid); });
      const float t3 = time_us([&] { run_pair<3>(p, grid); });
      const float t4 = time_us([&] { run_pair<4>(p, grid); });
      const float t7 = time_us([&] { run_pair<7>(p, grid); });
      printf("   pair, %d workgroups: %8.1f us %6.0f TF | no DMA %6.0f | no LDS reads %6.0f | neither %6.0f | no epilogue %6.0f | "
             "MFMAs + barriers only %6.0f TF\n", grid, t0, tf / t0, tf / t1, tf / t2, tf / t3, tf / t4, tf / t7);
    }
  }
  return 0;
}
