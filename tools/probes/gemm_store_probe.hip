// Where does the tile boundary of gemm256 go?  The plain bf16 k-major GEMM on the K = 768 shapes of the step (where the
// ablation of profiles/r05_gemm_pair_probe.txt attributes 35-53 % of the time to the stores), with the store variants
// gemm256_kernel already carries as PROBE modes, in one process:
//   0 product | 5 no stores (math kept live) | 6 nontemporal stores | 7 every tile of a workgroup overwrites the same 64 KiB
//   (L2-resident: the stores are issued and reach the L2, nothing is written back to HBM) | 3 no global->LDS requests
// and the product kernel on a HALF grid (128 workgroups: the other half of the chip idle).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I big_vision_amd/csrc -I include tools/probes/gemm_store_probe.hip \
//         big_vision_amd/csrc/c_api.cpp -o tools/probes/gemm_store_probe.out && tools/probes/gemm_store_probe.out
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define BV_GEMM256_PROBES   // compiles the PROBE != 0 ablation paths of gemm256_kernel (absent from the library build)
#include "../../big_vision_amd/csrc/gemm256.hip"
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
void run(const G256Params& p, int grid) {
  hipLaunchKernelGGL((gemm256_kernel<true, PROBE, BV_EPI_NONE, false>), dim3(grid), dim3(512), 0, 0, p);
}
}  // namespace

int main() {
  struct Shape { const char* name; int M, N, K; } shapes[] = {
      {"qkv    401408x2304x768", 401408, 2304, 768}, {"fc1    401408x3072x768", 401408, 3072, 768},
      {"dx out 401408x768x768", 401408, 768, 768}, {"dx fc1 401408x768x3072", 401408, 768, 3072},
      {"qkv    100352x2304x768", 100352, 2304, 768}};
  unsigned short *a, *b; void* c;
  (void)hipMalloc(&a, (size_t)401408 * 3072 * 2); (void)hipMalloc(&b, (size_t)3072 * 3072 * 2);
  (void)hipMalloc(&c, (size_t)401408 * 3072 * 2);
  fill_bf16<<<2048, 256>>>(a, (size_t)401408 * 3072, 1u, 1.0f);
  fill_bf16<<<2048, 256>>>(b, (size_t)3072 * 3072, 2u, 0.05f);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto time_us = [&](auto fn) {
    fn(); fn(); fn();
    (void)hipDeviceSynchronize();
    const int it = 5;
    float t;
    (void)hipEventRecord(e0, 0); for (int i = 0; i < it; ++i) fn(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&t, e0, e1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); exit(1); }
    return t * 1e3f / it;
  };
  for (int rep = 0; rep < 2; ++rep)     // (the first pass of a process runs on a ramping clock: read the second)
    for (auto& s : shapes) {
      G256Params q{};
      q.A = (const bf16*)a; q.B = (const bf16*)b; q.C = c; q.lda = s.K; q.ldb = s.K; q.ldc = s.N;
      q.M = s.M; q.N = s.N; q.K = s.K; q.aux_rows = 1; q.tiles_n = s.N / 256; q.ntiles = (s.M / 256) * q.tiles_n;
      q.splits = 1; q.ktiles_per_split = s.K / 64; q.alpha = 1.f;
      const double tf = 2.0 * s.M * s.N * s.K / 1e6;
      const float g0 = time_us([&] { run<0>(q, 256); });
      const float g5 = time_us([&] { run<5>(q, 256); });
      const float g6 = time_us([&] { run<6>(q, 256); });
      const float g7 = time_us([&] { run<7>(q, 256); });
      const float g3 = time_us([&] { run<3>(q, 256); });
      const float gh = time_us([&] { run<0>(q, 128); });
      const float gh5 = time_us([&] { run<5>(q, 128); });
      printf("%s%-24s product %7.1f us %5.0f TF | no stores %5.0f | nontemporal %5.0f | stores to an L2-resident 64 KiB %5.0f | no requests %5.0f |"
             " 128 workgroups: %5.0f TF (no stores %5.0f) = %.2f of 256's\n", rep ? "" : "(warm-up pass) ", s.name, g0, tf / g0, tf / g5, tf / g6,
             tf / g7, tf / g3, tf / gh, tf / gh5, g0 / gh);
    }
  return 0;
}
