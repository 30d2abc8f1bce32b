"""A/B sweep of the k-major GEMM tuning knobs (BV_OPT_GEMM_NT / _SKEW_MODE / _SKEW_PCT / _PRE_ISSUE of the context) over every (shape, epilogue)
instance of the training step.  GPU only.  Prints one row per case, one column per variant;
also checks that every variant's output is bit-identical to the baseline's."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from big_vision_amd import ops, _lib

dev = torch.device("cuda:0")
BF16, F32 = torch.bfloat16, torch.float32
# (pre_issue, nt, skew_mode, skew_pct)
VARIANTS = [(0, 0, 1, 0), (1, 0, 1, 0), (1, 0, 1, 30), (0, 0, 1, 0), (1, 0, 1, 0)]


def timeit(fn, iters=6, warm=2):
  for _ in range(warm): fn()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters


def main():
  lib = _lib.load()
  D, M = 768, 3072
  print("variants (pre_issue, nt, skew_mode, skew_pct):", VARIANTS)
  sums = {}
  for T in (100352, 32768):
    x = torch.randn(T, D, device=dev).to(BF16)
    hM = torch.randn(T, M, device=dev).to(BF16)
    x3 = torch.randn(T, 3 * D, device=dev).to(BF16)
    res = torch.randn(T, D, device=dev)
    bias = {n: torch.randn(n, device=dev) for n in (D, 3 * D, M)}
    w = {(n, k): (torch.randn(n, k, device=dev) * 0.02).to(BF16) for n, k in ((3 * D, D), (D, D), (M, D), (D, M), (D, 3 * D))}
    cases = [
        ("fwd qkv bias->bf16", x, w[(3 * D, D)], dict(bias=bias[3 * D]), BF16),
        ("fwd out +resid->f32", x, w[(D, D)], dict(bias=bias[D], epilogue=ops.EPI_RESIDUAL, aux=res), F32),
        ("fwd fc1 gelu->2xbf16", x, w[(M, D)], dict(bias=bias[M], epilogue=ops.EPI_GELU, out2=torch.empty(T, M, device=dev, dtype=BF16)), BF16),
        ("fwd fc2 +resid->f32", hM, w[(D, M)], dict(bias=bias[D], epilogue=ops.EPI_RESIDUAL, aux=res), F32),
        ("dx fc2 gelu_bwd->bf16", x, w[(M, D)], dict(epilogue=ops.EPI_GELU_BWD, aux=hM), BF16),
        ("dx fc1 ->bf16", hM, w[(D, M)], dict(), BF16),
        ("dx out ->bf16", x, w[(D, D)], dict(), BF16),
        ("dx qkv ->bf16", x3, w[(D, 3 * D)], dict(), BF16),
    ]
    for name, a, b, kw, odt in cases:
      N, K = b.shape
      out = torch.empty(T, N, device=dev, dtype=odt)
      ref = None
      row = []
      for v in VARIANTS:
        for k_, v_ in zip(("gemm_pre_issue", "gemm_nt", "gemm_skew_mode", "gemm_skew_pct"), v):
          ops.ctx_set(k_, v_)
        out.zero_()
        ms = timeit(lambda: ops.gemm(a, b, a_kmajor=True, b_kmajor=True, out=out, **kw))
        if ref is None:
          ref = out.clone()
        elif not torch.equal(ref, out):
          print("MISMATCH", name, v)
        row.append(ms)
        sums[(T, v)] = sums.get((T, v), 0.0) + ms
      for k_, v_ in zip(("gemm_pre_issue", "gemm_nt", "gemm_skew_mode", "gemm_skew_pct"), (0, 0, 1, 0)):
        ops.ctx_set(k_, v_)
      print(f"T={T:6d} {name:24s} N={N:4d} K={K:4d} " + " ".join(f"{m*1e3:7.1f}" for m in row) +
            f"   best {2*T*N*K/min(row)/1e9:7.1f} TF/s")
    print(f"T={T} sums: " + " ".join(f"{sums[(T, v)]*1e3:7.1f}" for v in VARIANTS))


if __name__ == "__main__":
  main()
