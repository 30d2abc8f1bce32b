// Ablation / phase timing of attn5_bwd_kernel (attention5.hip).  Build one executable per variant:
//   for a in 0 1 2 4 16 32 7; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DA5_ABL=$a -DA5_STAMPS -I big_vision_amd/csrc \
//     -I include tools/probes/attn5_probe.hip big_vision_amd/csrc/c_api.cpp -o tools/probes/attn5_probe_$a.out; done
// A5_ABL bits: 1 no phase 1a, 2 no phase 1b, 4 no phase 2, 16 no global stores, 32 no tile loads (results are garbage).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../big_vision_amd/csrc/attention5.hip"

__global__ void fill(unsigned short* p, size_t n, unsigned seed, float amp) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned s = (unsigned)i * 2654435761u + seed;
    s ^= s >> 15; s *= 2246822519u; s ^= s >> 13;
    float f = ((s & 0xffff) / 65536.0f - 0.5f) * 2.0f * amp;
    unsigned u; memcpy(&u, &f, 4);
    p[i] = (unsigned short)(u >> 16);
  }
}
__global__ void fillf(float* p, size_t n, float v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 2048, L = argc > 2 ? atoi(argv[2]) : 196, H = 12;
  unsigned short *qkv, *d_o, *dqkv; float *lse, *delta, *dbias;
  (void)hipMalloc(&qkv, (size_t)n * L * 3 * H * 64 * 2);
  (void)hipMalloc(&dqkv, (size_t)n * L * 3 * H * 64 * 2);
  (void)hipMalloc(&d_o, (size_t)n * L * H * 64 * 2);
  (void)hipMalloc(&lse, (size_t)n * H * L * 4);
  (void)hipMalloc(&delta, (size_t)n * H * L * 4);
  (void)hipMalloc(&dbias, (size_t)n * 3 * H * 64 * 4);
  fill<<<2048, 256>>>(qkv, (size_t)n * L * 3 * H * 64, 1u, 1.0f);
  fill<<<2048, 256>>>(d_o, (size_t)n * L * H * 64, 7u, 1.0f);
  fillf<<<2048, 256>>>(lse, (size_t)n * H * L, 5.3f);   // ~log(196): probabilities of order 1/L
  (void)hipDeviceSynchronize();
#ifdef A5_STAMPS
  long* st; (void)hipMalloc(&st, 4 * 3 * 16 * 8); (void)hipMemset(st, 0, 4 * 3 * 16 * 8);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_a5_stamps), &st, sizeof(st));
#endif
#ifdef A5_WSTAMPS
  long* wst; (void)hipMalloc(&wst, 16 * 8 * 8); (void)hipMemset(wst, 0, 16 * 8 * 8);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_a5_wstamps), &wst, sizeof(wst));
#endif
  for (int rep = 0; rep < 4; ++rep) {   // 1, 0, 1, 0: the first measurement of a process also warms the clocks up
    const int withb = (rep & 1) ^ 1;
    for (int i = 0; i < 3; ++i) bv_attn5_bwd(qkv, d_o, lse, delta, dqkv, withb ? dbias : nullptr, n, L, H, nullptr, false);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) bv_attn5_bwd(qkv, d_o, lse, delta, dqkv, withb ? dbias : nullptr, n, L, H, nullptr, false);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("A5_ABL=%d n=%d L=%d dbias=%d: bwd %.1f us  (%s)\n", A5_ABL, n, L, withb, ms * 100.f, hipGetErrorString(hipGetLastError()));
#ifdef A5_STAMPS
    long h[4 * 3 * 16];
    (void)hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    printf("  cycles: compute waves  B1->1a | wait B2 | reduce+B3 | 1b | wait B4 | K-write,stores,sums | wait B5 | phase 2 | wait B6 | finalize   (total)\n");
    printf("          loader wave    issue loads | wait ..B4 | put dO,lse | wait ..B6 | put Q\n");
    for (int b = 0; b < 4; ++b)
      for (int w = 0; w < 3; ++w) {
        const long* q = h + (b * 3 + w) * 16;
        if (w < 2) {
          printf("  wg %d wave %s:", 100 + b, w ? "KF-1" : "0   ");
          for (int k = 1; k <= 10; ++k) printf(" %6ld", q[k] - q[k - 1]);
          printf("   (%ld)\n", q[10] - q[0]);
        } else {
          printf("  wg %d loader   : %6ld %6ld %6ld %6ld %6ld\n", 100 + b, q[1] - q[0], q[5] - q[1], q[6] - q[5], q[9] - q[6], q[10] - q[9]);
        }
      }
    (void)hipMemset(st, 0, 4 * 3 * 16 * 8);
#endif
#ifdef A5_WSTAMPS
    {
      long w[16 * 8];
      (void)hipMemcpy(w, wst, sizeof(w), hipMemcpyDeviceToHost);
      long t0 = 0;
      for (int v = 0; v < 16; ++v) if (w[v * 8 + 0] && (!t0 || w[v * 8 + 0] < t0)) t0 = w[v * 8 + 0];
      printf("  workgroup 100, cycles since the first wave left B3: end 1b | at B4 | left B4 | at B5 | left B5 | at B6 | left B6\n");
      for (int v = 0; v < 16; ++v) {
        printf("   wave %2d:", v);
        for (int k = 1; k < 8; ++k) printf(" %7ld", w[v * 8 + k] ? w[v * 8 + k] - t0 : -1L);
        printf("\n");
      }
      (void)hipMemset(wst, 0, 16 * 8 * 8);
    }
#endif
  }
  return 0;
}
