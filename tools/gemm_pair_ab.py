"""A/B of the k-major GEMM kernels per step shape and epilogue: gemm256 (one 8-wave workgroup per CU, 256x256 tiles)
vs gemm_pair (two 4-wave workgroups per CU, 256x128 tiles; BV_OPT_GEMM_PAIR).  us per launch, TFLOP/s.  GPU only.

  python tools/gemm_pair_ab.py [T_image T_text]      (default 401408 131072: micro-batches of 2048 pairs)
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from big_vision_amd import ops

dev = torch.device("cuda:0")
BF16, F32 = torch.bfloat16, torch.float32
PAIR_ALL = sum(1 << e for e in (0, 1, 2, 3, 4, 6, 7, 8))


def timeit(fn, iters=6, warm=2):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters):
    fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters


def main():
  D, M = 768, 3072
  Ts = [int(a) for a in sys.argv[1:3]] or [401408, 131072]
  tot = {0: 0.0, 1: 0.0}
  for T in Ts:
    x = torch.randn(T, D, device=dev).to(BF16)
    hM = torch.randn(T, M, device=dev).to(BF16)
    x3 = torch.randn(T, 3 * D, device=dev).to(BF16)
    res = torch.randn(T, D, device=dev)
    bias = {n: torch.randn(n, device=dev) for n in (D, 3 * D, M)}
    w = {(n, k): (torch.randn(n, k, device=dev) * 0.02).to(BF16) for n, k in ((3 * D, D), (D, D), (M, D), (D, M), (D, 3 * D))}
    o2 = torch.empty(T, M, device=dev, dtype=BF16)
    cs = torch.zeros(M, device=dev)
    cases = [
        ("fwd qkv  bias->bf16", x, w[(3 * D, D)], dict(bias=bias[3 * D], out_dtype=BF16)),
        ("fwd out  bias+resid->f32", x, w[(D, D)], dict(bias=bias[D], out_dtype=F32, epilogue=ops.EPI_RESIDUAL, aux=res)),
        ("fwd fc1  bias+gelu->2xbf16", x, w[(M, D)], dict(bias=bias[M], out_dtype=BF16, epilogue=ops.EPI_GELU, out2=o2)),
        ("fwd fc1  gelu_gd->2xbf16", x, w[(M, D)], dict(bias=bias[M], out_dtype=BF16, epilogue=ops.EPI_GELU_GD, out2=o2)),
        ("fwd fc2  bias+resid->f32", hM, w[(D, M)], dict(bias=bias[D], out_dtype=F32, epilogue=ops.EPI_RESIDUAL, aux=res)),
        ("dx  fc2  gelu'-emit+colsum", x, w[(M, D)], dict(out_dtype=BF16, epilogue=ops.EPI_GELU_BWD_EMIT, aux=hM, out2=o2, colsum=cs)),
        ("dx  fc2  mul+colsum", x, w[(M, D)], dict(out_dtype=BF16, epilogue=ops.EPI_MUL, aux=hM, colsum=cs)),
        ("dx  fc1  ->bf16", hM, w[(D, M)], dict(out_dtype=BF16)),
        ("dx  out  ->bf16", x, w[(D, D)], dict(out_dtype=BF16)),
        ("dx  qkv  ->bf16", x3, w[(D, 3 * D)], dict(out_dtype=BF16)),
    ]
    for name, a, b, kw in cases:
      N, K = b.shape
      kw = dict(kw)
      kw["out"] = torch.empty(T, N, device=dev, dtype=kw.pop("out_dtype"))
      us = {}
      for rep in range(2):          # interleaved: 256, pair, 256, pair
        for pair in (0, 1):
          with ops.option("gemm_pair", PAIR_ALL if pair else 0):
            ms = timeit(lambda: ops.gemm(a, b, a_kmajor=True, b_kmajor=True, **kw))
          us[pair] = min(us.get(pair, 1e9), ms * 1e3)
      fl = 2.0 * T * N * K
      for pair in (0, 1):
        tot[pair] += us[pair]
      print(f"T={T:6d} {name:30s} N={N:4d} K={K:4d}  gemm256 {us[0]:8.1f} us {fl / us[0] / 1e6:7.1f} TF/s   pair {us[1]:8.1f} us "
            f"{fl / us[1] / 1e6:7.1f} TF/s   pair/256 time {us[1] / us[0]:.3f}", flush=True)
  print(f"sum over the cases: gemm256 {tot[0]:.0f} us, pair {tot[1]:.0f} us ({tot[1] / tot[0]:.3f})")


if __name__ == "__main__":
  main()
