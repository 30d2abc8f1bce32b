// Phase stamps of attn6_bwd_kernel (tools/probes/attention6.hip: the 8-symmetric-wave attention backward of round 6, a
// measured non-win that left the library - profiles/r06_attn6_ab.txt), every wave of one mid-launch workgroup:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DA6_STAMPS -I big_vision_amd/csrc -I include tools/probes/attn6_probe.hip \
//     big_vision_amd/csrc/c_api.cpp -o tools/probes/attn6_probe.out && ./tools/probes/attn6_probe.out [n] [L]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "attention6.hip"

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
  unsigned short *qkv, *d_o, *dqkv; float *lse, *dbias;
  (void)hipMalloc(&qkv, (size_t)n * L * 3 * H * 64 * 2);
  (void)hipMalloc(&dqkv, (size_t)n * L * 3 * H * 64 * 2);
  (void)hipMalloc(&d_o, (size_t)n * L * H * 64 * 2);
  (void)hipMalloc(&lse, (size_t)n * H * L * 4);
  (void)hipMalloc(&dbias, (size_t)n * 3 * H * 64 * 4);
  fill<<<2048, 256>>>(qkv, (size_t)n * L * 3 * H * 64, 1u, 1.0f);
  fill<<<2048, 256>>>(d_o, (size_t)n * L * H * 64, 7u, 1.0f);
  fillf<<<2048, 256>>>(lse, (size_t)n * H * L, 5.3f);
  (void)hipDeviceSynchronize();
  long* st; (void)hipMalloc(&st, 8 * 16 * 8); (void)hipMemset(st, 0, 8 * 16 * 8);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_a6_stamps), &st, sizeof(st));
  for (int rep = 0; rep < 4; ++rep) {
    const int withb = (rep & 1) ^ 1;
    for (int i = 0; i < 3; ++i) bv_attn6_bwd(qkv, d_o, lse, dqkv, withb ? dbias : nullptr, n, L, H, nullptr);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) bv_attn6_bwd(qkv, d_o, lse, dqkv, withb ? dbias : nullptr, n, L, H, nullptr);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("attn6 n=%d L=%d dbias=%d: bwd %.1f us  (%s)\n", n, L, withb, ms * 100.f, hipGetErrorString(hipGetLastError()));
    long h[8 * 16];
    (void)hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    if (rep < 2) {
      printf("  cycles per wave: 1a | wait B2 | delta+B3 | 1b | wait B4 | (-) | K write | wait B5, put dO,lse,cso, K/V loads | phase 2 mfma | stores | wait B6 | put Q, dbias  (total)\n");
      for (int w = 0; w < 8; ++w) {
        const long* q = h + w * 16;
        printf("   wave %d (%d frag):", w, w < 5 ? 2 : 1);
        for (int k = 1; k <= 12; ++k) printf(" %6ld", q[k] - q[k - 1]);
        printf("   (%ld)\n", q[12] - q[0]);
      }
    }
    (void)hipMemset(st, 0, sizeof(h));
  }
  return 0;
}
