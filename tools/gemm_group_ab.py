"""A/B of the k-major tile order (BV_OPT_GEMM_GROUP_N): per step shape and epilogue, us per launch for groups of
g column tiles, bit-compared with the plain order; `--step g [bench args]` runs bench.py's main with the knob set.  GPU only."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from big_vision_amd import _lib, ops

BF16, F32 = torch.bfloat16, torch.float32


def timeit(fn, iters=6, warm=2):
  for _ in range(warm): fn()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e3


def main():
  lib = _lib.load()
  if len(sys.argv) > 2 and sys.argv[1] == "--step":
    ops.ctx_set("gemm_group_n", int(sys.argv[2]))
    import bench
    sys.argv = ["bench.py"] + sys.argv[3:]
    return bench.main()
  dev = torch.device("cuda:0")
  D, M = 768, 3072
  groups = (0, 2, 3, 4, 6)
  print(f"# us per launch; columns: group_n = {groups}; '=' bit-identical to the plain order")
  for T in (401408, 131072, 100352, 32768):
    x = torch.randn(T, D, device=dev).to(BF16)
    hM = torch.randn(T, M, device=dev).to(BF16)
    res = torch.randn(T, D, device=dev)
    bias = {n: torch.randn(n, device=dev) for n in (D, 3 * D, M)}
    w = {(n, k): (torch.randn(n, k, device=dev) * 0.02).to(BF16) for n, k in ((3 * D, D), (M, D), (D, M), (D, D))}
    cases = [
        ("qkv bias N=2304 K=768", x, w[(3 * D, D)], dict(bias=bias[3 * D]), 1),
        ("fc1 gelu 2 outs N=3072 K=768", x, w[(M, D)], dict(bias=bias[M], epilogue=ops.EPI_GELU), 2),
        ("dfc2 gelu' N=3072 K=768", x, w[(M, D)], dict(epilogue=ops.EPI_GELU_BWD, aux=hM), 1),
        ("dfc2 gelu'emit N=3072 K=768", x, w[(M, D)], dict(epilogue=ops.EPI_GELU_BWD_EMIT, aux=hM), 2),
        ("fc2 +res f32 N=768 K=3072", hM, w[(D, M)], dict(bias=bias[D], epilogue=ops.EPI_RESIDUAL, aux=res, out_dtype=F32), 1),
        ("out +res f32 N=768 K=768", x, w[(D, D)], dict(bias=bias[D], epilogue=ops.EPI_RESIDUAL, aux=res, out_dtype=F32), 1),
    ]
    for name, a, b, kw, nout in cases:
      N = b.shape[0]
      kw = dict(kw)
      od = kw.pop("out_dtype", BF16)
      out = torch.empty(T, N, device=dev, dtype=od)
      out2 = torch.empty(T, N, device=dev, dtype=BF16) if nout == 2 else None
      run = lambda: ops.gemm(a, b, a_kmajor=True, b_kmajor=True, out=out, out2=out2, **kw)
      cells, ref = [], None
      for g in groups:
        ops.ctx_set("gemm_group_n", g)
        us = timeit(run)
        got = (out.clone(), out2.clone() if out2 is not None else None)
        if ref is None:
          ref = got
          same = ""
        else:
          same = "=" if torch.equal(ref[0], got[0]) and (ref[1] is None or torch.equal(ref[1], got[1])) else "!"
        cells.append(f"{us:8.1f}{same}")
      ops.ctx_set("gemm_group_n", 0)
      flops = 2.0 * T * N * a.shape[1]
      print(f"T={T:6d} {name:32s} | " + " ".join(cells) + f" | plain {flops / float(cells[0].strip('=!')) / 1e6:6.0f} TFLOP/s", flush=True)
    del x, hM, res


if __name__ == "__main__":
  main()
