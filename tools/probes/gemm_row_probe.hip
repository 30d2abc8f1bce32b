// Main-loop rate of a FULL-ROW tile for the N = 768 GEMMs (VERDICT r4 "next" #3: a 128 x 768 tile would see whole rows
// of the residual stream, so the out-proj / fc2 "+residual" epilogue could emit LayerNorm(x) and the QKV / fc1 dX
// epilogue could run the LayerNorm backward's row reductions - no separate LayerNorm passes).  Before any of that: can
// such a tile feed the matrix pipes?  This probe is the K loop only, with a plain bf16 store epilogue:
//   tile 128 (M) x 768 (N), 8 waves as 2 (M) x 4 (N), wave tile 64 x 192 = acc[4][12] (192 accumulator VGPRs),
//   K-tile 32: stage = A 128 rows x 64 B + B 768 rows x 64 B = 56 KiB, 2 stages = 112 KiB, one workgroup per CU,
//   global_load_lds_dwordx4 feed (7 pieces per wave and K-tile, requested one K-tile ahead), one barrier per K-tile,
//   the LDS image / swizzle / fragment maps of gemm_pair.hip -> results bit-identical to gemm256 (checked).
// Prints TFLOP/s next to gemm256 (256 x 256 tile) on the same shapes.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I big_vision_amd/csrc -I include tools/probes/gemm_row_probe.hip \
//         big_vision_amd/csrc/c_api.cpp -o tools/probes/gemm_row_probe.out && tools/probes/gemm_row_probe.out
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
__global__ void cmp_words(const unsigned* a, const unsigned* b, size_t n, unsigned long long* bad) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned long long c = 0;
  for (; i < n; i += stride) c += a[i] != b[i];
  if (c) atomicAdd(bad, c);
}

namespace {

constexpr int R_STAGE_A = 128 * 64, R_STAGE_B = 768 * 64, R_STAGE = R_STAGE_A + R_STAGE_B;   // 56 KiB
constexpr int R_SMEM = 2 * R_STAGE;

struct RowParams {
  const bf16* A;
  const bf16* B;
  bf16* C;
  long lda, ldb, ldc;
  int M, K, ntiles;   // N = 768, ntiles = M / 128
};

template <int OFF>
__device__ __forceinline__ bf16x8 r_read(uint32_t addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return __builtin_bit_cast(bf16x8, v);
}

template <int PROBE>
__global__ __launch_bounds__(512, 2) void gemm_row_kernel(RowParams p) {
  extern __shared__ __attribute__((aligned(1024))) char rsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int lr = lane & 15, lg = lane >> 4;
  const int bid = blockIdx.x, G = gridDim.x;
  const int nk = p.K >> 5;
  const int nmy = bid < p.ntiles ? (p.ntiles - bid + G - 1) / G : 0;
  if (nmy == 0) return;
  // DMA: piece q = wave + 8 k, k = 0..6: k = 0 -> A piece `wave` (rows 16 wave ..), k >= 1 -> B piece wave + 8 (k - 1)
  const int chunk = (lane & 3) ^ (((lane >> 5) & 1) << 1);
  const bf16* const srcA = p.A + (long)(wave * 16 + (lane >> 2)) * p.lda + chunk * 8;
  const bf16* const srcB = p.B + (long)(wave * 16 + (lane >> 2)) * p.ldb + chunk * 8;
  const long qB = 128 * p.ldb;
  bool quiet = false;
  auto issue = [&](int j, int t, int stage) {
    if ((PROBE & 1) && quiet) return;
    const long m0 = (long)(bid + j * G) * 128;
    char* d = rsm + stage * R_STAGE + wave * 1024;
    __builtin_amdgcn_global_load_lds((gl_void*)(srcA + m0 * p.lda + t * 32), (lds_void*)d, 16, 0, 0);
    const bf16* b = srcB + t * 32;
#pragma unroll
    for (int k = 0; k < 6; ++k)
      __builtin_amdgcn_global_load_lds((gl_void*)(b + k * qB), (lds_void*)(d + R_STAGE_A + k * 8192), 16, 0, 0);
  };
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)rsm;
  const uint32_t ra = lds0 + (wr * 64 + lr) * 64 + ((lg ^ (((lr >> 3) & 1) << 1)) << 4);
  const int brow = (lr >> 2) * 8 + (lr & 3);
  const uint32_t rb = lds0 + R_STAGE_A + (wc * 192 + brow) * 64 + ((lg ^ (((brow >> 3) & 1) << 1)) << 4);

  f32x4 acc[4][12];
  bf16x8 af[4], bq[2][4];
  int lj = 0, lt = 0;   // load cursor
  issue(0, 0, 0);
  lt = 1;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int gk = 0;
  for (int jt = 0; jt < nmy; ++jt) {
    for (int t = 0; t < nk; ++t) {
      const uint32_t so = (gk & 1) * R_STAGE;
      const uint32_t a0 = ra + so, b0 = rb + so;
      // B fragment j of the wave: rows (j >> 1) * 32 + (j & 1) * 4 + brow -> byte offset (j >> 1) * 2048 + (j & 1) * 256
#define R_READ_B(set, jb)                                                                  \
  do {                                                                                     \
    bq[set][0] = r_read<((jb) * 4 + 0) / 2 * 2048 + 0>(b0);                                \
    bq[set][1] = r_read<((jb) * 4 + 0) / 2 * 2048 + 256>(b0);                              \
    bq[set][2] = r_read<((jb) * 4 + 2) / 2 * 2048 + 0>(b0);                                \
    bq[set][3] = r_read<((jb) * 4 + 2) / 2 * 2048 + 256>(b0);                              \
  } while (0)
#define R_MFMA(set, jb, ZERO)                                                                             \
  do {                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                         \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                       \
        acc[i][(jb) * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                   \
            bq[set][j], af[i], (ZERO) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][(jb) * 4 + j], 0, 0, 0);      \
  } while (0)
      if (!((PROBE & 2) && quiet)) {
        af[0] = r_read<0>(a0); af[1] = r_read<1024>(a0); af[2] = r_read<2048>(a0); af[3] = r_read<3072>(a0);
        R_READ_B(0, 0);
        R_READ_B(1, 1);
      }
      // next K-tile of the stream into the other stage (everybody left it before the last barrier)
      const bool more = lj < nmy;
      if (more) {
        issue(lj, lt, (gk + 1) & 1);
        if (++lt == nk) { lt = 0; ++lj; }
      }
      asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      asm volatile("" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[0][2]), "+v"(bq[0][3]));
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
      if (t == 0) R_MFMA(0, 0, true); else R_MFMA(0, 0, false);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(acc[0][0]), "+v"(acc[1][1]), "+v"(acc[2][2]), "+v"(acc[3][3])::"memory");
      asm volatile("" : "+v"(bq[1][0]), "+v"(bq[1][1]), "+v"(bq[1][2]), "+v"(bq[1][3]));
      __builtin_amdgcn_sched_barrier(0);
      if (!((PROBE & 2) && quiet)) R_READ_B(0, 2);     // third batch into set 0 (its MFMAs were issued above)
      if (t == 0) R_MFMA(1, 1, true); else R_MFMA(1, 1, false);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(acc[0][4]), "+v"(acc[1][5]), "+v"(acc[2][6]), "+v"(acc[3][7])::"memory");
      asm volatile("" : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[0][2]), "+v"(bq[0][3]));
      __builtin_amdgcn_sched_barrier(0);
      if (t == 0) R_MFMA(0, 2, true); else R_MFMA(0, 2, false);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      ++gk;
      quiet = true;
    }
#undef R_READ_B
#undef R_MFMA
    // plain bf16 epilogue: lane holds row m0 + wr*64 + i*16 + lr, columns wc*192 + (j>>1)*32 + lg*8 + (j&1)*4 .. +3
    const long m0 = (long)(bid + jt * G) * 128 + wr * 64;
    if (PROBE & 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 12; ++j) asm volatile("" ::"v"(acc[i][j]));
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bf16* c = p.C + (m0 + i * 16 + lr) * p.ldc + wc * 192 + lg * 8;
#pragma unroll
        for (int jp = 0; jp < 6; ++jp) {
          const f32x4 x = acc[i][jp * 2], y = acc[i][jp * 2 + 1];
          *reinterpret_cast<u32x4*>(c + jp * 32) = u32x4{pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3]), pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3])};
        }
      }
    }
  }
}

template <int PROBE>
void run_row(const RowParams& p, int grid) {
  auto kern = gemm_row_kernel<PROBE>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, R_SMEM);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), R_SMEM, 0, p);
}

}  // namespace

int main() {
  struct Shape { const char* name; int M, K; } shapes[] = {
      {"check 1024x768x64", 1024, 64}, {"check 12544x768x192", 12544, 192},
      {"out-proj 401408x768x768", 401408, 768}, {"fc2 / dx fc1 401408x768x3072", 401408, 3072},
      {"dx qkv 401408x768x2304", 401408, 2304}, {"out-proj 100352x768x768", 100352, 768}};
  unsigned short *a, *b; void *c0, *c1; unsigned long long* bad;
  (void)hipMalloc(&a, (size_t)401408 * 3072 * 2); (void)hipMalloc(&b, (size_t)768 * 3072 * 2);
  (void)hipMalloc(&c0, (size_t)401408 * 768 * 2); (void)hipMalloc(&c1, (size_t)401408 * 768 * 2);
  (void)hipMalloc(&bad, 8);
  fill_bf16<<<2048, 256>>>(a, (size_t)401408 * 3072, 1u, 1.0f);
  fill_bf16<<<2048, 256>>>(b, (size_t)768 * 3072, 2u, 0.05f);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto time_us = [&](auto fn) {
    fn(); fn();
    (void)hipDeviceSynchronize();
    const int it = 4;
    float t;
    (void)hipEventRecord(e0, 0); for (int i = 0; i < it; ++i) fn(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&t, e0, e1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); exit(1); }
    return t * 1e3f / it;
  };
  bv_gemm_roll(0);
  for (auto& s : shapes) {
    const int N = 768;
    const size_t cb = (size_t)s.M * N * 2;
    (void)hipMemset(c0, 0xff, cb); (void)hipMemset(c1, 0xee, cb);
    RowParams p{(const bf16*)a, (const bf16*)b, (bf16*)c1, s.K, s.K, N, s.M, s.K, s.M / 128};
    const int grid = p.ntiles < 256 ? p.ntiles : 256;
    auto ref = [&]() {
      if (!bv_gemm256_try(1, 1, a, s.K, b, s.K, c0, N, 0, s.M, N, s.K, BV_EPI_NONE, nullptr, nullptr, 0, 0, nullptr, 1.0f, 0, nullptr,
                          nullptr, probe_ctx())) { printf("ref not dispatched\n"); exit(1); }
    };
    ref(); run_row<0>(p, grid);
    (void)hipDeviceSynchronize();
    (void)hipMemset(bad, 0, 8);
    cmp_words<<<2048, 256>>>((const unsigned*)c0, (const unsigned*)c1, cb / 4, bad);
    unsigned long long hb; (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
    const double tf = 2.0 * s.M * N * s.K / 1e6;
    const float g = time_us(ref);
    const float r0 = time_us([&] { run_row<0>(p, grid); });
    const float r1 = time_us([&] { run_row<1>(p, grid); });
    const float r2 = time_us([&] { run_row<2>(p, grid); });
    const float r4 = time_us([&] { run_row<4>(p, grid); });
    const float r7 = time_us([&] { run_row<7>(p, grid); });
    printf("%-30s words differing %llu | gemm256 %8.1f us %6.0f TF | row tile %8.1f us %6.0f TF (x%.3f) | no DMA %6.0f | no LDS reads %6.0f | "
           "no epilogue %6.0f | MFMAs + barriers only %6.0f TF\n", s.name, hb, g, tf / g, r0, tf / r0, g / r0, tf / r1, tf / r2, tf / r4, tf / r7);
  }
  return 0;
}
