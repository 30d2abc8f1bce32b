// Probe: semantics of ds_read_b64_tr_b16 on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  int lane = threadIdx.x;
  // mode 0: lane l address = l*8 bytes (contiguous chunks)
  // mode 1: lane l -> row (l&3) + 4*(l>>4)... custom: address = ((l>>2)*64 + (l&3)*4) elements
  int elem = mode == 0 ? lane * 4 : ((lane >> 2) * 64 + (lane & 3) * 4);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + elem));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  return 0;
}
