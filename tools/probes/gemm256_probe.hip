// Bottleneck ablation of the 256x256 GEMM main loop (standalone, no torch):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I big_vision_amd/csrc tools/probes/gemm256_probe.hip \
//         big_vision_amd/csrc/c_api.cpp -o /tmp/gemm256_probe && /tmp/gemm256_probe
// Variants (see PROBE in gemm256.hip): 0 full, 1 no LDS reads after tile 0,
// 2 (TN) b128 reads instead of transpose reads, 3 no DMA after the prologue, 4 no MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define BV_GEMM256_PROBES   // compiles the PROBE != 0 ablation paths of gemm256_kernel (absent from the library build)
#include "../../big_vision_amd/csrc/gemm256.hip"

template <bool KM, int PROBE>
static float run(G256Params p, int grid, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((gemm256_kernel<KM, PROBE>), dim3(grid), dim3(512), 0, 0, p);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((gemm256_kernel<KM, PROBE>), dim3(grid), dim3(512), 0, 0, p);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("HIP error: %s\n", hipGetErrorString(e));
  return ms / iters;
}

static long* g_dbg = nullptr;
// average shader clock over the last launch (block 0): d(s_memtime) / d(s_memrealtime @100 MHz)
static void report_clock(const char* what) {
  long h[4];
  hipMemcpy(h, g_dbg, sizeof(h), hipMemcpyDeviceToHost);
  const double cyc = (double)(h[2] - h[0]), ref = (double)(h[3] - h[1]);
  printf("      [%s] block 0: %.0f shader cycles in %.1f us -> %.0f MHz\n", what, cyc, ref / 100.0, cyc / ref * 100.0);
}

static void fill(void* d, size_t n_bf16) {
  std::vector<unsigned short> h(n_bf16);
  unsigned s = 12345;
  for (size_t i = 0; i < n_bf16; ++i) {
    s = s * 1664525u + 1013904223u;
    float f = ((s >> 8) & 0xffff) / 65536.0f * 2.f - 1.f;   // U(-1,1)
    unsigned u; memcpy(&u, &f, 4);
    h[i] = (unsigned short)(u >> 16);
  }
  hipMemcpy(d, h.data(), n_bf16 * 2, hipMemcpyHostToDevice);
}

int main() {
  const int T = 100352;
  struct Shape { const char* name; int in, out; } shapes[] = {{"qkv", 768, 2304}, {"out", 768, 768}, {"fc1", 768, 3072}, {"fc2", 3072, 768}};
  void *x, *y, *w, *c;
  hipError_t e1 = hipMalloc(&x, (size_t)T * 3072 * 2), e2 = hipMalloc(&y, (size_t)T * 3072 * 2);
  hipError_t e3 = hipMalloc(&w, (size_t)3072 * 3072 * 2), e4 = hipMalloc(&c, (size_t)T * 3072 * 4);
  printf("malloc: %d %d %d %d  %p %p %p %p\n", e1, e2, e3, e4, x, y, w, c); fflush(stdout);
  fill(x, (size_t)T * 3072); fill(y, (size_t)T * 3072); fill(w, (size_t)3072 * 3072);
  hipMemset(c, 0, (size_t)T * 3072 * 4);
  hipMalloc(&g_dbg, 4096 * sizeof(long)); hipMemset(g_dbg, 0, 4096 * sizeof(long));
  for (auto& s : shapes) {
    const double fl = 2.0 * T * s.in * s.out;
    {  // TN: dW[in][out] = X^T dY, split-K atomics
      G256Params p{};
      p.A = (const bf16*)x; p.B = (const bf16*)y; p.C = c; p.lda = s.in; p.ldb = s.out; p.ldc = s.out;
      p.M = s.in; p.N = s.out; p.K = T; p.aux_rows = 1; p.tiles_n = s.out / 256;
      p.ntiles = (s.in / 256) * p.tiles_n; p.epi = BV_EPI_ATOMIC; p.out_f32 = 1; p.alpha = 1.f;
      const int nk = T / 64;
      for (int rounds = 1; rounds <= 2; ++rounds) {
        int splits = rounds * 256 / p.ntiles; if (splits < 1) splits = 1;
        p.ktiles_per_split = (nk + splits - 1) / splits;
        splits = (nk + p.ktiles_per_split - 1) / p.ktiles_per_split;
        p.splits = splits;
        const int grid = p.ntiles * splits < 256 ? p.ntiles * splits : 256;
        float t0 = run<false, 0>(p, grid, 5), t1 = run<false, 1>(p, grid, 5), t2 = run<false, 2>(p, grid, 5),
              t3 = run<false, 3>(p, grid, 5), t4 = 0;
        printf("TN %-4s grid %4d (splits %3d): full %.3f ms %6.0f TF | noLDSread %.3f | b128reads %.3f | noDMA %.3f | noMFMA %.3f\n",
               s.name, grid, splits, t0, fl / t0 / 1e9, t1, t2, t3, t4);
      }
    }
    {  // NT: Y[T][out] = X[T][in] W^T[out][in]
      G256Params p{};
      p.A = (const bf16*)x; p.B = (const bf16*)w; p.C = c; p.lda = s.in; p.ldb = s.in; p.ldc = s.out;
      p.M = T; p.N = s.out; p.K = s.in; p.aux_rows = 1; p.tiles_n = s.out / 256;
      p.ntiles = (T / 256) * p.tiles_n; p.epi = BV_EPI_NONE; p.out_f32 = 0; p.alpha = 1.f;
      p.ktiles_per_split = s.in / 64;
      p.splits = 1;
      const int grid = p.ntiles < 256 ? p.ntiles : 256;
      float t0 = run<true, 0>(p, grid, 5), t1 = run<true, 1>(p, grid, 5), t3 = run<true, 3>(p, grid, 5), t4 = run<true, 5>(p, grid, 5);
      if (s.in == 768 && s.out == 2304) {
        const int period = (s.in / 64) * 2700 + 11000;
        for (int pc = 0; pc <= 100; pc += 50) {
          p.dbg = g_dbg; p.skew_mode = 1; p.skew_cycles = period * pc / 100; p.pre_issue = 0;
          hipMemset(g_dbg, 0, 4096 * sizeof(long));
          float tt = run<true, 10>(p, grid, 1);
          std::vector<long> st(4096);
          hipMemcpy(st.data(), g_dbg, 4096 * sizeof(long), hipMemcpyDeviceToHost);
          printf("   skew %d%% (%.3f ms): tile-end times (us since block 0 tile 0 end) of XCD-0 blocks idx 0,8,16,24,31\n", pc, tt);
          const long base = st[1024];
          for (int idx : {0, 8, 16, 24, 31}) {
            printf("     idx %2d:", idx);
            for (int j = 0; j < 14; ++j) printf(" %6.1f", (st[1024 + idx * 32 + j] - base) / 100.0);
            printf("\n");
          }
        }
        p.skew_cycles = 0; p.dbg = nullptr;
      }
      p.dbg = g_dbg;
      float t6 = run<true, 6>(p, grid, 5);
      printf("   nontemporal stores: %.3f ms %6.0f TF\n", t6, fl / t6 / 1e9);
      report_clock("nt stores");
      run<true, 8>(p, grid, 5); report_clock("full");
      if (s.in == 768) {
        run<true, 9>(p, grid, 1);
        std::vector<long> st(3072);
        hipMemcpy(st.data(), g_dbg, 3072 * sizeof(long), hipMemcpyDeviceToHost);
        const int nk = s.in / 64, per = nk + 2;
        for (int b = 0; b < 2; ++b) {
          const long* q = st.data() + 1024 + b * 1024;
          printf("      [stamps block %d] per tile: K-tile durations | epilogue | (cycles)\n", b);
          long prev = st[b * 4 + 0];
          for (int t = 0; t < 6; ++t) {
            printf("        tile %d (start +%ld):", t, q[t * per] - st[b * 4 + 0]);
            for (int i = 0; i < per; ++i) { printf(" %ld", q[t * per + i] - prev); prev = q[t * per + i]; }
            printf("\n");
          }
        }
      }
      {
        const char* names[] = {"", "", "XCD 0 only (32 WGs)", "every 8th WG of each XCD (32 WGs)", "WG 0 alone", "one WG per XCD (8)",
                               "8 WGs of XCD 0", "XCDs 0-3 (128 WGs)"};
        for (int m = 2; m <= 7; ++m) {
          p.skew_mode = m;
          float ta = run<true, 8>(p, grid, 3), tb = run<true, 5>(p, grid, 3);
          printf("      subset %-36s: full %.3f ms | no stores %.3f ms | stores cost %.3f ms\n", names[m], ta, tb, ta - tb);
        }
        p.skew_mode = 0;
      }
      run<true, 5>(p, grid, 5); report_clock("no stores");
      run<true, 3>(p, grid, 5); report_clock("no DMA");
      run<true, 1>(p, grid, 5); report_clock("no LDS reads");
      p.dbg = nullptr;
      p.skew_cycles = 0; p.skew_mode = 0;
    }
  }
  return 0;
}
