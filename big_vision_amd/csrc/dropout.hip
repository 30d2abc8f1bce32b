// Dropout for the encoder (models/vit.py:76,100,109,228: nn.Dropout(rate)(x, deterministic) after the GELU,
// after each residual branch and behind the position embedding): y = keep * x / (1 - rate), keep ~ Bernoulli(1 - rate).
//
// HBM-bound element-wise kernels.  The keep bits are NOT stored: they are a pure function of (key, element index)
// - Philox-4x32-10 keyed by the 64-bit site key, counter = index of a group of four consecutive elements, element j
// of the group keeps iff word j < floor((1 - rate) 2^32) - so the backward (and a micro-batch forward that is re-run)
// regenerates exactly the forward's mask from the key alone.  bv_dropout_mask exports the bits for the parity tests
// (the oracle takes the masks as inputs: JAX's own random stream cannot be reproduced here).
#include "bv_common.h"

namespace {

struct Keep4 { bool k[4]; };

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t r[4]) {
  uint32_t c2 = 0, c3 = 0;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

__device__ __forceinline__ Keep4 keep_of(unsigned long long key, long group, unsigned long long thr) {
  uint32_t r[4];
  philox4x32_10((uint32_t)group, (uint32_t)((unsigned long long)group >> 32), (uint32_t)key, (uint32_t)(key >> 32), r);
  Keep4 k;
#pragma unroll
  for (int j = 0; j < 4; ++j) k.k[j] = (unsigned long long)r[j] < thr;
  return k;
}

// out_f32 / out_bf16 (either may be null) = addend (or 0) + keep * scale * x   (out_f32 may be x or addend: element-wise)
__global__ __launch_bounds__(256) void dropout_f32_kernel(const float* x, const float* addend, float* out_f32, bf16* out_bf16,
                                                          long groups,
                                                          unsigned long long key, unsigned long long thr, float scale) {
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (long)gridDim.x * 256) {
    const Keep4 k = keep_of(key, g, thr);
    const float4 v = *reinterpret_cast<const float4*>(x + g * 4);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (addend) a = *reinterpret_cast<const float4*>(addend + g * 4);
    const float4 y = make_float4(a.x + (k.k[0] ? v.x * scale : 0.f), a.y + (k.k[1] ? v.y * scale : 0.f),
                                 a.z + (k.k[2] ? v.z * scale : 0.f), a.w + (k.k[3] ? v.w * scale : 0.f));
    if (out_f32) *reinterpret_cast<float4*>(out_f32 + g * 4) = y;
    if (out_bf16) *reinterpret_cast<uint2*>(out_bf16 + g * 4) = make_uint2(pack_bf2(y.x, y.y), pack_bf2(y.z, y.w));
  }
}

// in place on one or two bf16 tensors of the same shape (the SAME mask on both: gelu(h) and gelu'(h))
__global__ __launch_bounds__(256) void dropout_bf16_kernel(bf16* __restrict__ a, bf16* __restrict__ b, long groups,
                                                           unsigned long long key, unsigned long long thr, float scale) {
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (long)gridDim.x * 256) {
    const Keep4 k = keep_of(key, g, thr);
    uint2 u = *reinterpret_cast<const uint2*>(a + g * 4);
    u.x = pack_bf2(k.k[0] ? bflo(u.x) * scale : 0.f, k.k[1] ? bfhi(u.x) * scale : 0.f);
    u.y = pack_bf2(k.k[2] ? bflo(u.y) * scale : 0.f, k.k[3] ? bfhi(u.y) * scale : 0.f);
    *reinterpret_cast<uint2*>(a + g * 4) = u;
    if (b) {
      uint2 w = *reinterpret_cast<const uint2*>(b + g * 4);
      w.x = pack_bf2(k.k[0] ? bflo(w.x) * scale : 0.f, k.k[1] ? bfhi(w.x) * scale : 0.f);
      w.y = pack_bf2(k.k[2] ? bflo(w.y) * scale : 0.f, k.k[3] ? bfhi(w.y) * scale : 0.f);
      *reinterpret_cast<uint2*>(b + g * 4) = w;
    }
  }
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(unsigned char* __restrict__ out, long groups, unsigned long long key,
                                                           unsigned long long thr) {
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < groups; g += (long)gridDim.x * 256) {
    const Keep4 k = keep_of(key, g, thr);
    *reinterpret_cast<uchar4*>(out + g * 4) = make_uchar4(k.k[0], k.k[1], k.k[2], k.k[3]);
  }
}

inline int grid_of(long groups) {
  const long b = (groups + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

inline bool rate_ok(float rate) { return rate >= 0.f && rate < 1.f; }
inline unsigned long long thr_of(float rate) {
  const double q = 1.0 - (double)rate;
  const double t = q * 4294967296.0;
  return t >= 4294967296.0 ? 4294967296ull : (unsigned long long)t;
}

}  // namespace

extern "C" int bv_dropout_f32(const float* x, const float* addend, float* out_f32, void* out_bf16, long count,
                              unsigned long long key, float rate, void* stream) {
  BV_REQUIRE(count > 0 && count % 4 == 0, "bv_dropout_f32: count must be a positive multiple of 4");
  BV_REQUIRE(rate_ok(rate), "bv_dropout_f32: rate must be in [0, 1)");
  BV_REQUIRE(x && (out_f32 || out_bf16), "bv_dropout_f32: x and at least one output are required");
  BV_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)addend % 16 == 0 && (uintptr_t)out_f32 % 16 == 0 && (uintptr_t)out_bf16 % 8 == 0,
             "bv_dropout_f32: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(dropout_f32_kernel, dim3(grid_of(count / 4)), dim3(256), 0, (hipStream_t)stream, x, addend, out_f32,
                     (bf16*)out_bf16, count / 4, key, thr_of(rate), (float)(1.0 / (1.0 - (double)rate)));
  return bv_check_launch("bv_dropout_f32");
}

extern "C" int bv_dropout_bf16(void* a, void* b, long count, unsigned long long key, float rate, void* stream) {
  BV_REQUIRE(count > 0 && count % 4 == 0, "bv_dropout_bf16: count must be a positive multiple of 4");
  BV_REQUIRE(rate_ok(rate), "bv_dropout_bf16: rate must be in [0, 1)");
  BV_REQUIRE(a && (uintptr_t)a % 8 == 0 && (uintptr_t)b % 8 == 0, "bv_dropout_bf16: pointers must be 8-byte aligned");
  hipLaunchKernelGGL(dropout_bf16_kernel, dim3(grid_of(count / 4)), dim3(256), 0, (hipStream_t)stream, (bf16*)a, (bf16*)b,
                     count / 4, key, thr_of(rate), (float)(1.0 / (1.0 - (double)rate)));
  return bv_check_launch("bv_dropout_bf16");
}

extern "C" int bv_dropout_mask(void* keep_u8, long count, unsigned long long key, float rate, void* stream) {
  BV_REQUIRE(count > 0 && count % 4 == 0, "bv_dropout_mask: count must be a positive multiple of 4");
  BV_REQUIRE(rate_ok(rate) && keep_u8 && (uintptr_t)keep_u8 % 4 == 0, "bv_dropout_mask: rate in [0, 1), 4-byte aligned output");
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_of(count / 4)), dim3(256), 0, (hipStream_t)stream, (unsigned char*)keep_u8,
                     count / 4, key, thr_of(rate));
  return bv_check_launch("bv_dropout_mask");
}
