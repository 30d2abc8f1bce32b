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

// flax nn.gelu(approximate=True): 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3), written as
// x * sigmoid(2u) = x / (1 + exp2(z)), z = x (k0 + k1 x^2), k0 = -2 log2(e) sqrt(2/pi), k1 = 0.044715 k0:
// one v_exp_f32 + one v_rcp_f32 per element, no IEEE division.  These run inside GEMM epilogues (128
// evaluations per lane and 256x256 tile) where both waves of a SIMD are in their epilogue at the same time,
// i.e. they are bound by the SIMD's VALU throughput: everything but the two transcendentals is written on
// PAIRS of elements (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two fp32 lanes per instruction, full rate).
// The scalar entry points evaluate the SAME operation sequence (same fused / unfused roundings), so a value
// computed by any kernel through any of them has the same bits (forward g vs the backward's re-emitted g).
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define BV_GELU_K0 (-2.0f * 1.4426950408889634f * 0.7978845608028654f)
#define BV_GELU_K1 (BV_GELU_K0 * 0.044715f)
#define BV_GELU_U0 (2.0f * 0.7978845608028654f)
#define BV_GELU_U1 (BV_GELU_U0 * 3.0f * 0.044715f)
__device__ __forceinline__ f32x2 pk_splat(float a) { return f32x2{a, a}; }
__device__ __forceinline__ f32x2 gelu_sigmoid_pk(f32x2 x, f32x2 x2) {   // sigmoid(2u) of both elements
  const f32x2 z = x * __builtin_elementwise_fma(x2, pk_splat(BV_GELU_K1), pk_splat(BV_GELU_K0));
  const f32x2 den = f32x2{__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y)} + pk_splat(1.0f);
  return f32x2{__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
}
__device__ __forceinline__ f32x2 gelu_tanh_pk(f32x2 x) { return x * gelu_sigmoid_pk(x, x * x); }
// value g = x s and derivative dg = s + x s (1 - s) 2 u', 2 u' = U0 + U1 x^2, sharing exp / rcp
__device__ __forceinline__ void gelu_tanh_val_grad_pk(f32x2 x, f32x2& g, f32x2& dg) {
  const f32x2 x2 = x * x;
  const f32x2 s = gelu_sigmoid_pk(x, x2);
  g = x * s;
  const f32x2 up = __builtin_elementwise_fma(x2, pk_splat(BV_GELU_U1), pk_splat(BV_GELU_U0));
  dg = __builtin_elementwise_fma(g * (pk_splat(1.0f) - s), up, s);
}
__device__ __forceinline__ float gelu_tanh_f(float x) { return gelu_tanh_pk(pk_splat(x)).x; }
__device__ __forceinline__ void gelu_tanh_val_grad_f(float x, float& g, float& dg) {
  f32x2 gg, dd;
  gelu_tanh_val_grad_pk(pk_splat(x), gg, dd);
  g = gg.x; dg = dd.x;
}
__device__ __forceinline__ float gelu_tanh_grad_f(float x) {
  float g, dg;
  gelu_tanh_val_grad_f(x, g, dg);
  return dg;
}
// both bf16 halves of a packed dword as an fp32 pair (exact), and back (round to nearest even)
__device__ __forceinline__ f32x2 bf2_unpack(uint32_t w) { return f32x2{bflo(w), bfhi(w)}; }
__device__ __forceinline__ uint32_t bf2_pack(f32x2 v) { return pack_bf2(v.x, v.y); }
// The MLP activation as every epilogue applies it (one definition, so that the context kinds of the
// trainer - full: g and gelu' kept; gelu(h)-free / light: h kept - feed the backward the SAME bits):
//   h  = bf16(pre-activation)          the only form of h that is ever stored or differentiated
//   g  = bf16(gelu(h)),  d = bf16(gelu'(h))
// mlp_act_words: packed (h, g, d) of a pair of fp32 pre-activations; mlp_act_from_h: (g, d) from a stored h word.
__device__ __forceinline__ void mlp_act_from_h(uint32_t hw, uint32_t& gw, uint32_t& dw) {
  f32x2 g, d;
  gelu_tanh_val_grad_pk(bf2_unpack(hw), g, d);
  gw = bf2_pack(g); dw = bf2_pack(d);
}
__device__ __forceinline__ void mlp_act_words(f32x2 pre, uint32_t& hw, uint32_t& gw, uint32_t& dw) {
  hw = bf2_pack(pre);
  mlp_act_from_h(hw, gw, dw);
}
#endif
