"""Vendor yardstick for the dominant GEMM shapes (tools only - the product never calls a vendor library).

SURVEY 7 allows hipBLASLt / rocBLAS "only as A/B performance yardsticks".  For every k-major shape of the
training step (A [M,K] x B[N,K]^T, bf16 in, bf16 out, fp32 accumulate) this times, in ONE process and back
to back (same box, same clocks):
  * ours:     ops.gemm -> bv_gemm_bf16 (the 256x256 direct-to-LDS kernel), plain and with the bias epilogue;
  * hipBLASLt default heuristic: torch.matmul / torch.addmm with preferred_blas_library("hipblaslt");
  * the best of ALL hipBLASLt + rocBLAS solutions: PyTorch TunableOp (PYTORCH_TUNABLEOP_*), which times every
    solution hipblaslt-ext / rocblas expose for the shape and keeps the fastest.
Prints one table; run as `python tools/gemm_yardstick.py > profiles/r03_gemm_yardstick.txt`.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TUNE = "--no-tune" not in sys.argv
if TUNE:
  os.environ.setdefault("PYTORCH_TUNABLEOP_ENABLED", "1")
  os.environ.setdefault("PYTORCH_TUNABLEOP_TUNING", "1")
  os.environ.setdefault("PYTORCH_TUNABLEOP_VERBOSE", "0")
  os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "8")
  os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS", "3")
  os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS", "2")
  os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", "/tmp/bv_tunableop_results.csv")

import torch  # noqa: E402

BF16 = torch.bfloat16
# M = 131 072 (text tower, 2048 pairs) first and TUNED (every vendor solution timed: ~3x cheaper to sweep than
# the image shapes and the same N, K); M = 401 408 (image tower) with the library's default heuristic only.
SHAPES = [(131072, 2304, 768), (131072, 768, 768), (131072, 3072, 768), (131072, 768, 3072), (131072, 768, 2304),
          (401408, 2304, 768), (401408, 768, 768), (401408, 3072, 768), (401408, 768, 3072), (401408, 768, 2304)]
TUNE_MAX_M = 131072
if "--quick" in sys.argv:
  SHAPES = [s for s in SHAPES if s[0] <= 131072]


def timeit(fn, iters):
  fn()
  torch.cuda.synchronize()
  best = 1e30
  for _ in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
      fn()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / iters)
  return best


def main():
  from big_vision_amd import ops
  dev = torch.device("cuda", 0)
  torch.backends.cuda.preferred_blas_library("hipblaslt")
  g = torch.Generator(device=dev).manual_seed(0)
  print(f"# torch {torch.__version__}, hip {torch.version.hip}, device {torch.cuda.get_device_name(0)}, tunableop {'on' if TUNE else 'off'}")
  print(f"# {'M':>7} {'N':>5} {'K':>5} | {'ours':>7} {'ours+bias':>9} | {'hipblaslt':>9} {'+bias':>7} | {'tuned best':>10} {'+bias':>7} | ours/best   (TFLOP/s, bf16 in/out)")
  rows = []
  for (M, N, K) in SHAPES:
    a = (torch.rand((M, K), device=dev, generator=g) * 2 - 1).to(BF16)
    b = ((torch.rand((N, K), device=dev, generator=g) * 2 - 1) * 0.05).to(BF16)
    bias = torch.rand((N,), device=dev, generator=g)
    bias16 = bias.to(BF16)
    out = torch.empty((M, N), device=dev, dtype=BF16)
    fl = 2.0 * M * N * K
    it = 4 if M > 200000 else 8
    tf = lambda ms: fl / ms / 1e9
    ours = tf(timeit(lambda: ops.gemm(a, b, a_kmajor=True, b_kmajor=True, out=out), it))
    ours_b = tf(timeit(lambda: ops.gemm(a, b, a_kmajor=True, b_kmajor=True, out=out, bias=bias), it))
    bt = b.t()
    if TUNE:
      torch.cuda.tunable.enable(False)
    lt = tf(timeit(lambda: torch.matmul(a, bt, out=out), it))
    lt_b = tf(timeit(lambda: torch.addmm(bias16, a, bt, out=out), it))
    best = best_b = float("nan")
    tuned = TUNE and M <= TUNE_MAX_M
    if tuned:
      torch.cuda.tunable.enable(True)
      torch.cuda.tunable.tuning_enable(True)
      t0 = time.time()
      torch.matmul(a, bt, out=out)          # tunes this shape (all hipBLASLt + rocBLAS solutions)
      torch.cuda.synchronize()
      tune_s = time.time() - t0
      torch.cuda.tunable.tuning_enable(False)
      best = tf(timeit(lambda: torch.matmul(a, bt, out=out), it))
    ref = max(x for x in (lt, best) if x == x)
    print(f"  {M:>7} {N:>5} {K:>5} | {ours:7.0f} {ours_b:9.0f} | {lt:9.0f} {lt_b:7.0f} | {best:10.0f} {best_b:7.0f} | {ours / ref:6.2f}"
          + (f"   (tuning {tune_s:.0f} s)" if tuned else ""), flush=True)
    rows.append((M, N, K, ours, ref))
    del a, b, out
  if TUNE:
    try:
      res = torch.cuda.tunable.get_results()
      print("# TunableOp picks:")
      for r in res:
        print("#  ", r)
    except Exception as e:   # noqa: BLE001
      print("# (no tunable results:", e, ")")
  # ---- weight-gradient GEMMs (reduction over the tokens): dW[in, out] = X^T dY, ours = split-K slabs + reduce
  # into an fp32 gradient buffer (accumulating), vendor = torch.matmul(x.t(), dy) -> bf16 [in, out] (less work:
  # no fp32 output, no accumulation), default heuristic.
  print(f"# dW = X^T dY   {'T':>7} {'in':>5} {'out':>5} | {'ours (fp32 +=)':>14} | {'hipblaslt (bf16 out)':>20} | ours/vendor")
  for (T, din, dout) in [(401408, 768, 2304), (401408, 768, 768), (401408, 768, 3072), (401408, 3072, 768),
                         (131072, 768, 2304), (131072, 768, 3072), (131072, 3072, 768)]:
    if "--quick" in sys.argv and T > 131072:
      continue
    x = (torch.rand((T, din), device=dev, generator=g) * 2 - 1).to(BF16)
    dy = ((torch.rand((T, dout), device=dev, generator=g) * 2 - 1) * 0.05).to(BF16)
    grad = torch.zeros((din, dout), device=dev, dtype=torch.float32)
    fl = 2.0 * T * din * dout
    tf = lambda ms: fl / ms / 1e9
    ours = tf(timeit(lambda: ops.gemm(x, dy, a_kmajor=False, b_kmajor=False, out=grad, epilogue=ops.EPI_ATOMIC), 4))
    xt = x.t()
    o16 = torch.empty((din, dout), device=dev, dtype=BF16)
    if TUNE:
      torch.cuda.tunable.enable(False)
    lt = tf(timeit(lambda: torch.matmul(xt, dy, out=o16), 4))
    print(f"                {T:>7} {din:>5} {dout:>5} | {ours:14.0f} | {lt:20.0f} | {ours / lt:6.2f}", flush=True)
    del x, dy
  w = sum(2.0 * m * n * k for m, n, k, _, _ in rows)
  ours_t = sum(2.0 * m * n * k / o for m, n, k, o, _ in rows)
  ref_t = sum(2.0 * m * n * k / r for m, n, k, _, r in rows)
  print(f"# FLOP-weighted over the shapes: ours {w / ours_t:.0f} TFLOP/s, best vendor solution {w / ref_t:.0f} TFLOP/s")


if __name__ == "__main__":
  main()
