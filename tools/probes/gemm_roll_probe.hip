// A/B of the rolling-epilogue kernel (gemm256r_kernel) against gemm256_kernel<true> through the
// library's own dispatcher (bv_gemm256_try with bv_gemm_roll(0/1)): bit-exactness for the bf16
// epilogues (same accumulation order, same epilogue arithmetic), max-abs difference for the
// residual epilogue (the residual is summed first instead of last), run-to-run bit-equality of the
// new kernel (race screen) and timing of both on the training step's shapes.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I big_vision_amd/csrc tools/probes/gemm_roll_probe.hip \
//         big_vision_amd/csrc/c_api.cpp -o tools/probes/gemm_roll_probe.out && tools/probes/gemm_roll_probe.out
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
// res[0] = number of differing 32-bit words, res[1] = max |a - b| as float bits (fp32 compare only)
__global__ void cmp_words(const unsigned* a, const unsigned* b, size_t nwords, int as_f32, unsigned long long* res) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned long long bad = 0;
  float mx = 0.f;
  for (; i < nwords; i += stride) {
    bad += a[i] != b[i];
    if (as_f32) mx = fmaxf(mx, fabsf(__uint_as_float(a[i]) - __uint_as_float(b[i])));
  }
  if (bad) atomicAdd(res, bad);
  if (as_f32) atomicMax((unsigned*)(res + 1), __float_as_uint(mx));
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
  return ms / iters;
}

int main(int argc, char** argv) {
  bool quick = false;
  int pad = 0;
  int mask = 7;                      // bv_gemm_roll bits: 1 RESIDUAL, 2 NONE, 4 GELU, 8 stores in the MFMA segment
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "quick")) quick = true;
    if (!strncmp(argv[i], "mask=", 5)) mask = atoi(argv[i] + 5);
    if (!strncmp(argv[i], "pad=", 4)) pad = atoi(argv[i] + 4);       // extra elements in the C row pitch
  }
  printf("roll mask %d, C row pitch N + %d\n", mask, pad);
  struct Shape { const char* name; int M, N, K, epi; } shapes[] = {
      {"check 256x256x128 (1 tile, 2 K-tiles)", 256, 256, 128, BV_EPI_NONE},
      {"check 512x256x192 none", 512, 256, 192, BV_EPI_NONE},
      {"check 1024x768x320 none", 1024, 768, 320, BV_EPI_NONE},
      {"check 66816x768x128 none (261 tiles x3)", 66816, 768, 128, BV_EPI_NONE},
      {"check 66816x768x128 gelu", 66816, 768, 128, BV_EPI_GELU},
      {"check 66816x768x128 resid", 66816, 768, 128, BV_EPI_RESIDUAL},
      {"check 2048x512x256 gelu", 2048, 512, 256, BV_EPI_GELU},
      {"check 256x256x128 resid", 256, 256, 128, BV_EPI_RESIDUAL},
      {"check 2304x768x768 resid", 2304, 768, 768, BV_EPI_RESIDUAL},
      {"img qkv  fwd  (bias)", 401408, 2304, 768, BV_EPI_NONE},
      {"img out  fwd  (+resid)", 401408, 768, 768, BV_EPI_RESIDUAL},
      {"img fc1  fwd  (gelu)", 401408, 3072, 768, BV_EPI_GELU},
      {"img fc2  fwd  (+resid)", 401408, 768, 3072, BV_EPI_RESIDUAL},
      {"img d_fc1 dx  (none)", 401408, 768, 3072, BV_EPI_NONE},
      {"img d_qkv dx  (none)", 401408, 768, 2304, BV_EPI_NONE},
      {"img d_o  dx   (none)", 401408, 768, 768, BV_EPI_NONE},
      {"txt qkv  fwd  (bias)", 131072, 2304, 768, BV_EPI_NONE},
      {"txt fc1  fwd  (gelu)", 131072, 3072, 768, BV_EPI_GELU},
      {"txt fc2  fwd  (+resid)", 131072, 768, 3072, BV_EPI_RESIDUAL},
      {"n512 qkv fwd  (bias)", 100352, 2304, 768, BV_EPI_NONE},
      {"n512 fc1 fwd  (gelu)", 100352, 3072, 768, BV_EPI_GELU},
      {"n512 out fwd  (+resid)", 100352, 768, 768, BV_EPI_RESIDUAL}};
  const size_t maxM = 401408;
  const size_t nA = maxM * 3072, nB = (size_t)3072 * 3072, nC = maxM * 3328;
  unsigned short *a, *b;
  void *c0, *c1, *g0, *g1;
  float *bias, *aux;
  unsigned long long* res;
  (void)hipMalloc(&a, nA * 2); (void)hipMalloc(&b, nB * 2);
  (void)hipMalloc(&c0, nC * 2); (void)hipMalloc(&c1, nC * 2);
  (void)hipMalloc(&g0, nC * 2); (void)hipMalloc(&g1, nC * 2);
  (void)hipMalloc(&bias, 3072 * 4); (void)hipMalloc(&aux, maxM * 768 * 4);
  (void)hipMalloc(&res, 16);
  fill_bf16<<<2048, 256>>>(a, nA, 12345u, 1.0f);
  fill_bf16<<<2048, 256>>>(b, nB, 999u, 0.05f);
  fill_f32<<<64, 256>>>(bias, 3072, 7u, 0.5f);
  fill_f32<<<2048, 256>>>(aux, maxM * 768, 31u, 2.0f);
  (void)hipDeviceSynchronize();
  int rc_all = 0;
  for (auto& s : shapes) {
    if (quick && s.M > 70000) continue;
    const bool f32 = s.epi == BV_EPI_RESIDUAL;
    const int ldc = s.N + pad;
    const size_t cbytes = (size_t)s.M * ldc * (f32 ? 4 : 2);
    auto run = [&](int roll, void* c, void* c2) {
      bv_gemm_roll(roll ? mask : 0);
      const int ok = bv_gemm256_try(1, 1, a, s.K, b, s.K, c, ldc, f32 ? 1 : 0, s.M, s.N, s.K, s.epi, bias,
                                    f32 ? aux : nullptr, s.N, 0, s.epi == BV_EPI_GELU ? c2 : nullptr, 1.0f, 0,
                                    nullptr, nullptr, probe_ctx());
      if (!ok) { printf("%s: not dispatched to the 256x256 path\n", s.name); exit(1); }
    };
    (void)hipMemset(c0, 0xff, cbytes); (void)hipMemset(c1, pad ? 0xff : 0xee, cbytes);
    (void)hipMemset(g0, 0xff, cbytes); (void)hipMemset(g1, pad ? 0xff : 0xee, cbytes);
    run(0, c0, g0);
    run(1, c1, g1);
    (void)hipDeviceSynchronize();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("%s: HIP error %s\n", s.name, hipGetErrorString(e)); return 1; }
    unsigned long long h[2];
    (void)hipMemset(res, 0, 16);
    cmp_words<<<2048, 256>>>((const unsigned*)c0, (const unsigned*)c1, cbytes / 4, f32, res);
    (void)hipMemcpy(h, res, 16, hipMemcpyDeviceToHost);
    unsigned long long bad_c = h[0];
    float maxd;
    { unsigned u = (unsigned)h[1]; memcpy(&maxd, &u, 4); }
    unsigned long long bad_g = 0;
    if (s.epi == BV_EPI_GELU) {
      (void)hipMemset(res, 0, 16);
      cmp_words<<<2048, 256>>>((const unsigned*)g0, (const unsigned*)g1, cbytes / 4, 0, res);
      (void)hipMemcpy(h, res, 16, hipMemcpyDeviceToHost);
      bad_g = h[0];
    }
    // race screen: the new kernel twice more, bitwise against its first run
    unsigned long long bad_rr = 0;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipMemset(c0, pad ? 0xff : 0x11, cbytes);
      run(1, c0, g0);
      (void)hipMemset(res, 0, 16);
      cmp_words<<<2048, 256>>>((const unsigned*)c0, (const unsigned*)c1, cbytes / 4, 0, res);
      (void)hipMemcpy(h, res, 16, hipMemcpyDeviceToHost);
      bad_rr += h[0];
    }
    const double fl = 2.0 * s.M * s.N * s.K;
    const int it = s.M > 70000 ? 5 : 3;
    float t0 = 0, t1 = 0;
    for (int rep = 0; rep < 2; ++rep) {   // interleaved A/B
      t0 += time_ms([&] { run(0, c0, g0); }, it) / 2;
      t1 += time_ms([&] { run(1, c1, g1); }, it) / 2;
    }
    const bool pass = f32 ? (maxd <= 1e-3f && bad_rr == 0) : (bad_c == 0 && bad_g == 0 && bad_rr == 0);
    if (!pass) rc_all = 1;
    printf("%-42s %s | old-vs-new words differing C %llu C2 %llu maxabs %.3g | rerun diffs %llu | old %.3f ms %6.0f TF | roll %.3f ms %6.0f TF | x%.3f\n",
           s.name, pass ? "OK  " : "FAIL", bad_c, bad_g, maxd, bad_rr, t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, t0 / t1);
    fflush(stdout);
  }
  printf(rc_all ? "RESULT: FAIL\n" : "RESULT: PASS\n");
  return rc_all;
}
