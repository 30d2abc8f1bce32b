// Probe: gemm256's k-major main loop with both operand streams fed through VGPRs (PROBE 16 of gemm256_kernel:
// global_load_dwordx4 into staging registers + ds_write_b128 two phases later) against the product loop (PROBE 0:
// global_load_lds DMA).  Same LDS images and arithmetic: the outputs must be BIT-IDENTICAL.  Plain bf16 epilogue, no
// pre-issue across the epilogue in either variant (p.pre_issue = 0).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I big_vision_amd/csrc tools/probes/gemm_vf_probe.hip \
//         big_vision_amd/csrc/c_api.cpp -o tools/probes/gemm_vf_probe.out && tools/probes/gemm_vf_probe.out
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define BV_GEMM256_PROBES   // compiles the PROBE != 0 ablation paths of gemm256_kernel (absent from the library build)
#include "../../big_vision_amd/csrc/gemm256.hip"

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
__global__ void cmp_words(const unsigned* a, const unsigned* b, size_t n, unsigned long long* bad) {
  unsigned long long c = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
  if (c) atomicAdd(bad, c);
}

template <int PROBE>
static void launch_plain(const void* a, const void* b, void* c, int M, int N, int K) {
  G256Params p{};
  p.A = (const bf16*)a; p.B = (const bf16*)b; p.C = c; p.lda = K; p.ldb = K; p.ldc = N;
  p.M = M; p.N = N; p.K = K; p.aux_rows = 1; p.tiles_n = N / 256;
  p.ntiles = (M / 256) * p.tiles_n; p.epi = BV_EPI_NONE; p.out_f32 = 0; p.alpha = 1.f;
  p.ktiles_per_split = K / 64; p.splits = 1;
  const int grid = p.ntiles < 256 ? p.ntiles : 256;
  hipLaunchKernelGGL((gemm256_kernel<true, PROBE, BV_EPI_NONE, false>), dim3(grid), dim3(512), 0, 0, p);
}

template <typename F>
static float time_ms(F launch, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) launch();
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

int main() {
  struct Shape { const char* name; int M, N, K; } shapes[] = {
      {"check 512x768x192", 512, 768, 192},     {"check 2048x2304x768", 2048, 2304, 768},
      {"qkv 131072x2304x768", 131072, 2304, 768}, {"out 131072x768x768", 131072, 768, 768},
      {"fc1 131072x3072x768", 131072, 3072, 768}, {"dfc1 131072x768x3072", 131072, 768, 3072},
      {"long-K 8192x2048x16384", 8192, 2048, 16384}};
  unsigned short *a, *b; void *c0, *c1; unsigned long long* bad;
  (void)hipMalloc(&a, (size_t)131072 * 3072 * 2); (void)hipMalloc(&b, (size_t)3072 * 16384 * 2);
  (void)hipMalloc(&c0, (size_t)131072 * 3072 * 2); (void)hipMalloc(&c1, (size_t)131072 * 3072 * 2);
  (void)hipMalloc(&bad, 8);
  fill_bf16<<<2048, 256>>>(a, (size_t)131072 * 3072, 12345u, 1.0f);
  fill_bf16<<<2048, 256>>>(b, (size_t)3072 * 16384, 999u, 0.05f);
  (void)hipDeviceSynchronize();
  for (int rep = 0; rep < 2; ++rep)
    for (auto& s : shapes) {
      const size_t cb = (size_t)s.M * s.N * 2;
      (void)hipMemset(c0, 0xff, cb); (void)hipMemset(c1, 0xee, cb);
      launch_plain<0>(a, b, c0, s.M, s.N, s.K);
      launch_plain<16>(a, b, c1, s.M, s.N, s.K);
      (void)hipDeviceSynchronize();
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) { printf("%s: HIP error %s\n", s.name, hipGetErrorString(e)); return 1; }
      (void)hipMemset(bad, 0, 8);
      cmp_words<<<2048, 256>>>((const unsigned*)c0, (const unsigned*)c1, cb / 4, bad);
      unsigned long long hb; (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
      const int it = 6;
      const float t0 = time_ms([&] { launch_plain<0>(a, b, c0, s.M, s.N, s.K); }, it);
      const float t1 = time_ms([&] { launch_plain<16>(a, b, c1, s.M, s.N, s.K); }, it);
      const double fl = 2.0 * s.M * s.N * s.K;
      printf("%-26s words differing %llu | DMA %.3f ms %6.0f TF | VGPR-fed %.3f ms %6.0f TF | x%.3f\n", s.name, hb, t0,
             fl / t0 / 1e9, t1, fl / t1 / 1e9, t0 / t1);
    }
  return 0;
}
