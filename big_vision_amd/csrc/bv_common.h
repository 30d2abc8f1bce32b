// Common device/host helpers for libbvhip (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define BV_OK 0
#define BV_ERR_INVALID_ARG (-1)
#define BV_ERR_UNSUPPORTED (-2)
#define BV_ERR_HIP (-3)

// Defined in c_api.cpp
void bv_set_error(const char* fmt, ...);
int bv_check_launch(const char* what);

#define BV_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) {                            \
      bv_set_error(__VA_ARGS__);              \
      return BV_ERR_INVALID_ARG;              \
    }                                         \
  } while (0)

#ifdef __HIPCC__
__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }

// low / high bf16 of a packed dword -> float (exact)
__device__ __forceinline__ float bflo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bfhi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  bf16x2 v;
  v[0] = (bf16)lo;
  v[1] = (bf16)hi;
  return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// flax nn.gelu(approximate=True): 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3).
// Written as x * sigmoid(2u) = x / (1 + exp2(-2 log2(e) u)): one v_exp_f32 + one v_rcp_f32,
// no IEEE division (these run inside GEMM epilogues, 128 evaluations per lane and tile).
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float k = -2.0f * 1.4426950408889634f * 0.7978845608028654f;   // -2 log2(e) sqrt(2/pi)
  const float z = k * x * (1.0f + 0.044715f * x * x);
  const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));   // sigmoid(2u)
  return x * sg;
}
// value g = x sigmoid(2u) and derivative d/dx g = s + x s (1 - s) 2 u',
// u' = sqrt(2/pi) (1 + 3*0.044715 x^2), in one pass (shared exp/rcp).  The explicit fmaf /
// products pin the rounding so that every caller (with or without the value) gets the same bits.
__device__ __forceinline__ void gelu_tanh_val_grad_f(float x, float& g, float& dg) {
  const float c = 0.7978845608028654f;
  const float x2 = x * x;
  const float z = (-2.0f * 1.4426950408889634f * c) * x * __builtin_fmaf(0.044715f, x2, 1.0f);
  const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
  g = x * sg;
  const float up = (2.0f * c) * __builtin_fmaf(3.0f * 0.044715f, x2, 1.0f);
  dg = __builtin_fmaf(g * (1.0f - sg), up, sg);
}
__device__ __forceinline__ float gelu_tanh_grad_f(float x) {
  float g, dg;
  gelu_tanh_val_grad_f(x, g, dg);
  return dg;
}
#endif
