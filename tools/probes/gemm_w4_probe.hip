// Probe: 4-wave variant of the 256x256x64 k-major GEMM.
//
// gemm256_kernel runs 8 waves (2 x 4, 128 x 64 per wave): per K-tile they read 192 KiB of fragments
// from the LDS next to 64 KiB of DMA writes = 2048 LDS cycles, exactly the 2062 MFMA cycles of the
// K-tile - the main loop is co-limited by both (measured 2670 cycles, 77 %).  Here 4 waves (2 x 2) own
// 128 x 128 each: 128 KiB of fragment reads per K-tile (1536 LDS cycles, 75 % of the MFMA time), the
// 256 accumulator registers of a wave live in the unified VGPR/AGPR file of a one-wave-per-SIMD kernel.
// There is no partner wave to hide a wave's LDS latency, so the wave software-pipelines itself:
//   P0(t):  64 MFMAs of k-step 0  ||  16 ds_read_b128 of k-step 1's fragments
//   --- s_waitcnt (DMA of K-tile t+1 landed), s_barrier: the ONE barrier per K-tile ---
//   P1(t):  64 MFMAs of k-step 1  ||  16 DMA instructions of K-tile t+2  ||  16 ds_read_b128 of (t+1, k-step 0)
// Same LDS images, swizzle and fragment / output maps as gemm256 (k-major, bf16 C, no bias), so the result
// must be BIT-IDENTICAL to gemm256_kernel<true> (same accumulation order per element).
// This probe answers one question: what does the main loop of this layout reach?  (No cross-tile DMA
// pipelining, no epilogue variants - a tile pays one pipeline fill.)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I big_vision_amd/csrc -I include tools/probes/gemm_w4_probe.hip \
//         big_vision_amd/csrc/c_api.cpp -o tools/probes/gemm_w4_probe.out && tools/probes/gemm_w4_probe.out
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define BV_GEMM256_PROBES   // compiles the PROBE != 0 ablation paths of gemm256_kernel (absent from the library build)
#include "../../big_vision_amd/csrc/gemm256.hip"
#include "probe_ctx.h"

namespace {

struct W4Params {
  const bf16* A;   // [M][K] k-major
  const bf16* B;   // [N][K] k-major
  bf16* C;         // [M][N]
  long lda, ldb, ldc;
  int M, N, K, tiles_n, ntiles;
};

constexpr int W4_STAGE = 4 * HALF;   // A half 0, A half 1, B half 0, B half 1 of one K-tile: 64 KiB
constexpr int W4_SMEM = 2 * W4_STAGE;

__global__ __launch_bounds__(256) void gemm_w4_kernel(W4Params p) {
  __shared__ __attribute__((aligned(1024))) char smem[W4_SMEM];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 15, lg = lane >> 4;
  const int nk = p.K >> 6;

  // ---- DMA: wave w moves rows [w*32, w*32 + 32) of each of the 4 half-tiles: 4 instructions of 8 rows
  const int drow = lane >> 3, dpos = lane & 7;
  int voffA[4], voffB[4];   // per-lane element offsets of the 4 row groups (uniform bases are added per issue)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = wave * 32 + q * 8 + drow;            // row inside the half
    const int chunk = dpos ^ kswz(r);
    voffA[q] = r * (int)p.lda + chunk * 8;
    voffB[q] = r * (int)p.ldb + chunk * 8;
  }
  // DMA instructions 2*idx, 2*idx + 1 of the 16 a wave issues per K-tile (h = idx >> 1, q = 2 * (idx & 1) + {0, 1})
  auto issue2 = [&](int stage, long offA, long offB, int kt, int idx) {
    char* sbase = smem + stage * W4_STAGE;
    const int h = idx >> 1;
    const bool isB = h >= 2;
    const bf16* base = (isB ? p.B + offB + (long)(h & 1) * 128 * p.ldb : p.A + offA + (long)(h & 1) * 128 * p.lda) +
                       (long)kt * 64;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
      const int q = 2 * (idx & 1) + qq;
      glds16(base + (isB ? voffB[q] : voffA[q]), sbase + h * HALF + (wave * 32 + q * 8) * 128);
    }
  };
  auto issue = [&](int stage, long offA, long offB, int kt) {
    char* sbase = smem + stage * W4_STAGE;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const bool isB = h >= 2;
      const bf16* base = (isB ? p.B + offB + (long)(h & 1) * 128 * p.ldb : p.A + offA + (long)(h & 1) * 128 * p.lda) +
                         (long)kt * 64;   // wave-uniform
#pragma unroll
      for (int q = 0; q < 4; ++q)
        glds16(base + (isB ? voffB[q] : voffA[q]), sbase + h * HALF + (wave * 32 + q * 8) * 128);
    }
  };

  // ---- fragment read addresses (relative to the stage base), as in gemm256_kernel<true> (bf16 output map)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int brow = (lr >> 2) * 8 + (lr & 3);
  const uint32_t ra0 = wr * HALF + lr * 128 + ((lg ^ kswz(lr)) << 4), ra1 = ra0 ^ 64;
  const uint32_t rb0 = 2 * HALF + wc * HALF + brow * 128 + ((lg ^ kswz(brow)) << 4), rb1 = rb0 ^ 64;

  f32x4 acc[8][8];
  bf16x8 fa[2][8], fb[2][8];   // [set][fragment]: set s holds k-step s

#define W4_READ_A(SET, KS, I, BASE)                                                     \
  fa[SET][I] = lds_read128<(I) * 2048>((BASE) + ((((I) & 1) ^ (KS)) ? ra1 : ra0))
#define W4_READ_B(SET, KS, J, BASE)                                                     \
  fb[SET][J] = lds_read128<((J) >> 1) * 4096 + ((J) & 1) * 512>((BASE) + ((((J) & 1) ^ (KS)) ? rb1 : rb0))
// The accumulators must LIVE in AGPRs: the builtin selects the VGPR form of the MFMA and uses the AGPRs as
// spill space (8 v_accvgpr moves around every MFMA); the inline-asm form names an AGPR destination.
#define W4_MFMA_ROW(SET, I)                                                             \
  _Pragma("unroll") for (int j = 0; j < 8; ++j)                                         \
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[I][j]) : "v"(fb[SET][j]), "v"(fa[SET][I]))
#define W4_PIN() __builtin_amdgcn_sched_barrier(0)

  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const long offA = (long)tm * 256 * p.lda, offB = (long)tn * 256 * p.ldb;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: K-tiles 0 and 1 in flight, K-tile 0 landed, its k-step-0 fragments in set 0
    issue(0, offA, offB, 0);
    if (nk > 1) issue(1, offA, offB, 1);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
      const uint32_t sb = lds0;
      W4_READ_A(0, 0, 0, sb); W4_READ_A(0, 0, 1, sb); W4_READ_A(0, 0, 2, sb); W4_READ_A(0, 0, 3, sb);
      W4_READ_A(0, 0, 4, sb); W4_READ_A(0, 0, 5, sb); W4_READ_A(0, 0, 6, sb); W4_READ_A(0, 0, 7, sb);
      W4_READ_B(0, 0, 0, sb); W4_READ_B(0, 0, 1, sb); W4_READ_B(0, 0, 2, sb); W4_READ_B(0, 0, 3, sb);
      W4_READ_B(0, 0, 4, sb); W4_READ_B(0, 0, 5, sb); W4_READ_B(0, 0, 6, sb); W4_READ_B(0, 0, 7, sb);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      W4_PIN();
    }

    for (int t = 0; t < nk; ++t) {
      const uint32_t sb = lds0 + (t & 1) * W4_STAGE, sn = lds0 + ((t + 1) & 1) * W4_STAGE;
      // ---- P0: k-step 0 MFMAs (set 0), k-step 1 fragments -> set 1 (two reads behind every MFMA row)
      W4_MFMA_ROW(0, 0); W4_PIN(); W4_READ_A(1, 1, 0, sb); W4_READ_B(1, 1, 0, sb); W4_PIN();
      W4_MFMA_ROW(0, 1); W4_PIN(); W4_READ_A(1, 1, 1, sb); W4_READ_B(1, 1, 1, sb); W4_PIN();
      W4_MFMA_ROW(0, 2); W4_PIN(); W4_READ_A(1, 1, 2, sb); W4_READ_B(1, 1, 2, sb); W4_PIN();
      W4_MFMA_ROW(0, 3); W4_PIN(); W4_READ_A(1, 1, 3, sb); W4_READ_B(1, 1, 3, sb); W4_PIN();
      W4_MFMA_ROW(0, 4); W4_PIN(); W4_READ_A(1, 1, 4, sb); W4_READ_B(1, 1, 4, sb); W4_PIN();
      W4_MFMA_ROW(0, 5); W4_PIN(); W4_READ_A(1, 1, 5, sb); W4_READ_B(1, 1, 5, sb); W4_PIN();
      W4_MFMA_ROW(0, 6); W4_PIN(); W4_READ_A(1, 1, 6, sb); W4_READ_B(1, 1, 6, sb); W4_PIN();
      W4_MFMA_ROW(0, 7); W4_PIN(); W4_READ_A(1, 1, 7, sb); W4_READ_B(1, 1, 7, sb); W4_PIN();
      // ---- the K-tile's one barrier: K-tile t+1 has landed (only its 16 DMAs are outstanding), every wave
      //      has read all of K-tile t (its k-step-1 fragments are in registers once lgkmcnt drains)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      W4_PIN();
      const bool more2 = t + 2 < nk;   // (the reads of a K-tile behind the last one fetch bytes nobody uses)
      // ---- P1: k-step 1 MFMAs (set 1), DMA of K-tile t+2 into the stage just released, (t+1, k-step 0) -> set 0
      // (two DMA instructions behind every MFMA row: issued in one block they cost the wave ~1000 cycles of
      //  issue time with an idle matrix pipe - first version of this probe: 1034 TFLOP/s at long K)
      W4_MFMA_ROW(1, 0); W4_PIN(); if (more2) issue2(t & 1, offA, offB, t + 2, 0); W4_READ_A(0, 0, 0, sn); W4_READ_B(0, 0, 0, sn); W4_PIN();
      W4_MFMA_ROW(1, 1); W4_PIN(); if (more2) issue2(t & 1, offA, offB, t + 2, 1); W4_READ_A(0, 0, 1, sn); W4_READ_B(0, 0, 1, sn); W4_PIN();
      W4_MFMA_ROW(1, 2); W4_PIN(); if (more2) issue2(t & 1, offA, offB, t + 2, 2); W4_READ_A(0, 0, 2, sn); W4_READ_B(0, 0, 2, sn); W4_PIN();
      W4_MFMA_ROW(1, 3); W4_PIN(); if (more2) issue2(t & 1, offA, offB, t + 2, 3); W4_READ_A(0, 0, 3, sn); W4_READ_B(0, 0, 3, sn); W4_PIN();
      W4_MFMA_ROW(1, 4); W4_PIN(); if (more2) issue2(t & 1, offA, offB, t + 2, 4); W4_READ_A(0, 0, 4, sn); W4_READ_B(0, 0, 4, sn); W4_PIN();
      W4_MFMA_ROW(1, 5); W4_PIN(); if (more2) issue2(t & 1, offA, offB, t + 2, 5); W4_READ_A(0, 0, 5, sn); W4_READ_B(0, 0, 5, sn); W4_PIN();
      W4_MFMA_ROW(1, 6); W4_PIN(); if (more2) issue2(t & 1, offA, offB, t + 2, 6); W4_READ_A(0, 0, 6, sn); W4_READ_B(0, 0, 6, sn); W4_PIN();
      W4_MFMA_ROW(1, 7); W4_PIN(); if (more2) issue2(t & 1, offA, offB, t + 2, 7); W4_READ_A(0, 0, 7, sn); W4_READ_B(0, 0, 7, sn); W4_PIN();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      W4_PIN();
    }

    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // MFMA results -> VALU reads behind the loop exit
    // ---- epilogue: bf16 C, a lane holds 8 consecutive columns per fragment pair (16-byte stores)
    const int m_base = tm * 256 + wr * 128 + lr, n_base = tn * 256 + wc * 128 + lg * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      bf16* crow = p.C + (long)(m_base + i * 16) * p.ldc + n_base;
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        const f32x4 a = acc[i][2 * jp], b = acc[i][2 * jp + 1];
        const u32x4 o = u32x4{pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3])};
        *reinterpret_cast<u32x4*>(crow + jp * 32) = o;
      }
    }
    __builtin_amdgcn_s_barrier();   // the next tile's prologue overwrites both stages
  }
#undef W4_READ_A
#undef W4_READ_B
#undef W4_MFMA_ROW
#undef W4_PIN
}

__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned s = (unsigned)i * 2654435761u + seed;
    s ^= s >> 15; s *= 2246822519u; s ^= s >> 13;
    const float f = ((s & 0xffff) / 65536.0f - 0.5f) * 2.0f * scale;
    unsigned u; memcpy(&u, &f, 4);
    p[i] = (unsigned short)(u >> 16);
  }
}
__global__ void cmp_words(const unsigned* a, const unsigned* b, size_t n, unsigned long long* bad) {
  unsigned long long c = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    c += a[i] != b[i];
  if (c) atomicAdd(bad, c);
}

}  // namespace

int main() {
  struct Shape { const char* name; int M, N, K; } shapes[] = {
      {"check 256x256x64", 256, 256, 64},   {"check 256x256x128", 256, 256, 128}, {"check 512x768x192", 512, 768, 192},
      {"qkv 100352x2304x768", 100352, 2304, 768}, {"dx 100352x768x3072", 100352, 768, 3072},
      {"long-K 8192x2048x16384", 8192, 2048, 16384}};
  unsigned short *a, *b; void *c0, *c1; unsigned long long* bad;
  (void)hipMalloc(&a, (size_t)100352 * 3072 * 2); (void)hipMalloc(&b, (size_t)3072 * 16384 * 2);
  (void)hipMalloc(&c0, (size_t)100352 * 2304 * 2); (void)hipMalloc(&c1, (size_t)100352 * 2304 * 2);
  (void)hipMalloc(&bad, 8);
  fill_bf16<<<2048, 256>>>(a, (size_t)100352 * 3072, 1u, 1.0f);
  fill_bf16<<<2048, 256>>>(b, (size_t)3072 * 16384, 2u, 0.05f);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  bv_gemm_roll(0);
  for (auto& s : shapes) {
    const size_t cb = (size_t)s.M * s.N * 2;
    (void)hipMemset(c0, 0xff, cb); (void)hipMemset(c1, 0xee, cb);
    W4Params p{(const bf16*)a, (const bf16*)b, (bf16*)c1, s.K, s.K, s.N, s.M, s.N, s.K, s.N / 256, (s.M / 256) * (s.N / 256)};
    const int grid = p.ntiles < 256 ? p.ntiles : 256;
    auto ref = [&]() {
      if (!bv_gemm256_try(1, 1, a, s.K, b, s.K, c0, s.N, 0, s.M, s.N, s.K, BV_EPI_NONE, nullptr, nullptr, 0, 0, nullptr, 1.0f,
                          0, nullptr, nullptr, probe_ctx())) { printf("ref not dispatched\n"); exit(1); }
    };
    auto w4 = [&]() { hipLaunchKernelGGL(gemm_w4_kernel, dim3(grid), dim3(256), 0, 0, p); };
    ref(); w4();
    (void)hipDeviceSynchronize();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("%s: HIP error %s\n", s.name, hipGetErrorString(e)); return 1; }
    (void)hipMemset(bad, 0, 8);
    cmp_words<<<2048, 256>>>((const unsigned*)c0, (const unsigned*)c1, cb / 4, bad);
    unsigned long long hb; (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
    const int it = 5;
    float t0, t1;
    (void)hipEventRecord(e0, 0); for (int i = 0; i < it; ++i) ref(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&t0, e0, e1);
    (void)hipEventRecord(e0, 0); for (int i = 0; i < it; ++i) w4(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&t1, e0, e1);
    const double fl = 2.0 * s.M * s.N * s.K * it;
    printf("%-28s words differing %llu | gemm256 %.3f ms %6.0f TF | w4 %.3f ms %6.0f TF | x%.3f\n", s.name, hb, t0 / it,
           fl / t0 / 1e9, t1 / it, fl / t1 / 1e9, t0 / t1);
  }
  return 0;
}
