"""Times every (shape, epilogue) instance of the k-major GEMM as the training step uses it
(image tower T=100352, text tower T=32768, ViT-B widths).  GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from big_vision_amd import ops

dev = torch.device("cuda:0")
BF16, F32 = torch.bfloat16, torch.float32


def timeit(fn, iters=8, warm=2):
  for _ in range(warm): fn()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters


def main():
  D, M = 768, 3072
  tot = {}
  for T in (100352, 32768):
    x = torch.randn(T, D, device=dev).to(BF16)
    hM = torch.randn(T, M, device=dev).to(BF16)
    x3 = torch.randn(T, 3 * D, device=dev).to(BF16)
    res = torch.randn(T, D, device=dev)
    bias = {n: torch.randn(n, device=dev) for n in (D, 3 * D, M)}
    w = {(n, k): (torch.randn(n, k, device=dev) * 0.02).to(BF16) for n, k in ((3 * D, D), (D, D), (M, D), (D, M), (D, 3 * D))}
    cases = [
        ("fwd qkv  bias->bf16", x, w[(3 * D, D)], dict(bias=bias[3 * D], out_dtype=BF16)),
        ("fwd out  bias+resid->f32", x, w[(D, D)], dict(bias=bias[D], out_dtype=F32, epilogue=ops.EPI_RESIDUAL, aux=res)),
        ("fwd fc1  bias+gelu->2xbf16", x, w[(M, D)], dict(bias=bias[M], out_dtype=BF16, epilogue=ops.EPI_GELU, out2=torch.empty(T, M, device=dev, dtype=BF16))),
        ("fwd fc2  bias+resid->f32", hM, w[(D, M)], dict(bias=bias[D], out_dtype=F32, epilogue=ops.EPI_RESIDUAL, aux=res)),
        ("dx  fc2  gelu_bwd->bf16", x, w[(M, D)], dict(out_dtype=BF16, epilogue=ops.EPI_GELU_BWD, aux=hM)),
        ("dx  fc1  ->bf16", hM, w[(D, M)], dict(out_dtype=BF16)),
        ("dx  out  ->bf16", x, w[(D, D)], dict(out_dtype=BF16)),
        ("dx  qkv  ->bf16", x3, w[(D, 3 * D)], dict(out_dtype=BF16)),
    ]
    s = 0.0
    for name, a, b, kw in cases:
      N, K = b.shape
      kw = dict(kw)
      if "out" not in kw:
        kw["out"] = torch.empty(T, N, device=dev, dtype=kw.pop("out_dtype"))
      else:
        kw.pop("out_dtype", None)
      ms = timeit(lambda: ops.gemm(a, b, a_kmajor=True, b_kmajor=True, **kw))
      from big_vision_amd import _lib
      lib = _lib.load()
      if hasattr(lib, "bv_gemm_skew"):
        lib.bv_gemm_skew(0)
        ms0 = timeit(lambda: ops.gemm(a, b, a_kmajor=True, b_kmajor=True, **kw))
        lib.bv_gemm_skew(2)
        ms = timeit(lambda: ops.gemm(a, b, a_kmajor=True, b_kmajor=True, **kw))
        lib.bv_gemm_skew(1)
        name = f"{name} [noskew {ms0*1e3:.1f}]"
      s += ms
      print(f"T={T:6d} {name:44s} N={N:4d} K={K:4d} {ms*1e3:8.1f} us {2*T*N*K/ms/1e9:8.1f} TF/s  tiles={T//256*(N//256):5d} ({T//256*(N//256)/256:.2f} rounds)")
    tot[T] = s
    print(f"T={T}: sum {s*1e3:.1f} us")


if __name__ == "__main__":
  main()
