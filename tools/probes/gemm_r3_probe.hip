// Round-3 GEMM probe (standalone, no torch):
//   (1) every epilogue variant of the 256x256 k-major kernel on the training step's shapes, through the
//       library's own dispatcher (bv_gemm256_try): TFLOP/s per shape and epilogue - in particular the
//       forward GELU pair (BV_EPI_GELU vs BV_EPI_GELU_GD) and the backward trio (GELU_BWD / _EMIT / MUL);
//   (2) the main loop with v_mfma_f32_32x32x16_bf16 (PROBE 11: same LDS images, DMA ring, barriers and
//       register counts, 8 MFMAs of 32x32x16 per phase instead of 16 of 16x16x32; results are garbage, the
//       fragments are not re-mapped) against the product main loop (PROBE 0), plain epilogue, same launch.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I big_vision_amd/csrc tools/probes/gemm_r3_probe.hip \
//         big_vision_amd/csrc/c_api.cpp -o tools/probes/gemm_r3_probe.out && tools/probes/gemm_r3_probe.out
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
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
__global__ void fill_f32(float* d, size_t n, unsigned seed, float scale) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned s = (unsigned)(i * 2246822519u) ^ seed;
    s ^= s >> 13; s *= 0x5bd1e995u; s ^= s >> 15;
    d[i] = (((s >> 8) & 0xffff) / 65536.0f * 2.f - 1.f) * scale;
  }
}

template <typename F>
static float time_ms(F launch, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms / iters;
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

int main(int argc, char** argv) {
  const bool quick = argc > 1 && !strcmp(argv[1], "quick");
  struct Shape { const char* name; int M, N, K; } shapes[] = {
      {"img qkv fwd   N=2304 K=768 ", 401408, 2304, 768},  {"img out / dO  N=768  K=768 ", 401408, 768, 768},
      {"img fc1 / dfc2 N=3072 K=768", 401408, 3072, 768},  {"img fc2 / dfc1 N=768 K=3072", 401408, 768, 3072},
      {"img dqkv dx   N=768  K=2304", 401408, 768, 2304},  {"txt qkv fwd   N=2304 K=768 ", 131072, 2304, 768},
      {"txt out / dO  N=768  K=768 ", 131072, 768, 768},   {"txt fc1 / dfc2 N=3072 K=768", 131072, 3072, 768},
      {"txt fc2 / dfc1 N=768 K=3072", 131072, 768, 3072},  {"txt dqkv dx   N=768  K=2304", 131072, 768, 2304},
      {"n512 img fc1  N=3072 K=768 ", 100352, 3072, 768},  {"n512 txt fc1  N=3072 K=768 ", 32768, 3072, 768},
      {"n512 txt out  N=768  K=768 ", 32768, 768, 768},    {"n512 txt fc2  N=768  K=3072", 32768, 768, 3072}};
  const size_t maxM = quick ? 32768 : 401408;
  unsigned short *a, *b, *auxb;
  void *c0, *c1;
  float *bias, *auxf, *colsum;
  (void)hipMalloc(&a, maxM * 3072 * 2); (void)hipMalloc(&b, (size_t)3072 * 3072 * 2);
  (void)hipMalloc(&c0, maxM * 3072 * 2); (void)hipMalloc(&c1, maxM * 3072 * 2);
  (void)hipMalloc(&auxb, maxM * 3072 * 2); (void)hipMalloc(&auxf, maxM * 768 * 4);
  (void)hipMalloc(&bias, 3072 * 4); (void)hipMalloc(&colsum, 3072 * 4);
  fill_bf16<<<2048, 256>>>(a, maxM * 3072, 12345u, 1.0f);
  fill_bf16<<<2048, 256>>>(b, (size_t)3072 * 3072, 999u, 0.05f);
  fill_bf16<<<2048, 256>>>(auxb, maxM * 3072, 77u, 2.0f);
  fill_f32<<<2048, 256>>>(auxf, maxM * 768, 31u, 2.0f);
  fill_f32<<<64, 256>>>(bias, 3072, 7u, 0.5f);
  (void)hipMemset(colsum, 0, 3072 * 4);
  (void)hipDeviceSynchronize();
  printf("%-28s | %s\n", "shape (TFLOP/s)", " bias  +res32 +res16   gelu gelu_gd  gelu' gelu'emit  mul  | main loop: 16x16x32 32x32x16 ratio");
  for (auto& s : shapes) {
    if ((size_t)s.M > maxM) continue;
    const double fl = 2.0 * s.M * s.N * s.K;
    const int it = s.M > 200000 ? 4 : 8;
    auto tf = [&](int epi, int f32, const void* aux, void* c2, float* cs) {
      auto run = [&] {
        const int ok = bv_gemm256_try(1, 1, a, s.K, b, s.K, c0, s.N, f32, s.M, s.N, s.K, epi, bias, aux, s.N, 0, c2, 1.0f, 0, cs, nullptr, probe_ctx());
        if (!ok) { printf("not dispatched\n"); exit(1); }
      };
      float t = 1e30f;
      for (int rep = 0; rep < 2; ++rep) t = fminf(t, time_ms(run, it));
      return fl / t / 1e9;
    };
    const double t_bias = tf(BV_EPI_NONE, 0, nullptr, nullptr, nullptr);
    const double t_r32 = s.N == 768 ? tf(BV_EPI_RESIDUAL, 1, auxf, nullptr, nullptr) : 0;
    const double t_r16 = s.N == 768 ? tf(BV_EPI_RESIDUAL, 0, auxb, nullptr, nullptr) : 0;
    double t_g = 0, t_gd = 0, t_gb = 0, t_ge = 0, t_mul = 0;
    if (s.N == 3072) {
      t_g = tf(BV_EPI_GELU, 0, nullptr, c1, nullptr);
      t_gd = tf(BV_EPI_GELU_GD, 0, nullptr, c1, nullptr);
      t_gb = tf(BV_EPI_GELU_BWD, 0, auxb, nullptr, colsum);
      t_ge = tf(BV_EPI_GELU_BWD_EMIT, 0, auxb, c1, colsum);
      t_mul = tf(BV_EPI_MUL, 0, auxb, nullptr, colsum);
    }
    float p0 = 1e30f, p11 = 1e30f, p12 = 1e30f, p13 = 1e30f, p14 = 1e30f, p15 = 1e30f;
    for (int rep = 0; rep < 2; ++rep) {
      p0 = fminf(p0, time_ms([&] { launch_plain<8>(a, b, c0, s.M, s.N, s.K); }, it));
      p11 = fminf(p11, time_ms([&] { launch_plain<11>(a, b, c0, s.M, s.N, s.K); }, it));
      p12 = fminf(p12, time_ms([&] { launch_plain<12>(a, b, c0, s.M, s.N, s.K); }, it));
      p13 = fminf(p13, time_ms([&] { launch_plain<13>(a, b, c0, s.M, s.N, s.K); }, it));
      p14 = fminf(p14, time_ms([&] { launch_plain<14>(a, b, c0, s.M, s.N, s.K); }, it));
      p15 = fminf(p15, time_ms([&] { launch_plain<15>(a, b, c0, s.M, s.N, s.K); }, it));
    }
    printf("%-28s | %5.0f %6.0f %6.0f %6.0f %7.0f %6.0f %8.0f %6.0f | %8.0f %8.0f  x%.3f | half barriers: no END %5.0f  no MID %5.0f | setprio: static %5.0f  none %5.0f\n", s.name, t_bias, t_r32, t_r16, t_g,
           t_gd, t_gb, t_ge, t_mul, fl / p0 / 1e9, fl / p11 / 1e9, p0 / p11, fl / p12 / 1e9, fl / p13 / 1e9, fl / p14 / 1e9, fl / p15 / 1e9);
    fflush(stdout);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); return 1; }
  return 0;
}
