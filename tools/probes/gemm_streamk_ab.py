"""A/B of the stream-K tail of the k-major 256x256 GEMMs (BV_OPT_GEMM_STREAMK) on the step's shapes: us per launch with the
option off / on (interleaved), the launch's round count, and the largest difference between the two outputs.  GPU only.

  python tools/gemm_streamk_ab.py [threshold_pct]      (default 100: every launch with a ragged last round)
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from big_vision_amd import ops

BF16, F32 = torch.bfloat16, torch.float32
# (name, rows, N, K, epilogue): image tower at 256 / 512 / 2048 samples, text tower at 512 / 2048 samples (ViT-B widths)
ROWS = {"img256": 256 * 196, "img512": 512 * 196, "img2048": 2048 * 196, "txt512": 512 * 64, "txt2048": 2048 * 64}
KINDS = [("qkv", 2304, 768, "none"), ("out+res", 768, 768, "res"), ("fc1 gelu", 3072, 768, "gelu"), ("fc2+res", 768, 3072, "res"),
         ("dX qkv", 768, 2304, "none"), ("dX out", 768, 768, "none"), ("dX fc2 gelu'emit", 3072, 768, "emit"), ("dX fc1", 768, 3072, "none")]


def launch(kind, x, w, b, aux32, aux16, outs):
  kw = dict(a_kmajor=True, b_kmajor=True)
  if kind == "none":
    return (ops.gemm(x, w, bias=b, out=outs[0], **kw),)
  if kind == "res":
    return (ops.gemm(x, w, bias=b, out=outs[2], epilogue=ops.EPI_RESIDUAL, aux=aux32, **kw),)
  if kind == "gelu":
    return (ops.gemm(x, w, bias=b, out=outs[0], epilogue=ops.EPI_GELU, out2=outs[1], **kw), outs[1])
  if kind == "emit":
    outs[3].zero_()
    return (ops.gemm(x, w, out=outs[0], epilogue=ops.EPI_GELU_BWD_EMIT, aux=aux16, out2=outs[1], colsum=outs[3], **kw), outs[1], outs[3])
  raise ValueError(kind)


def timed(fn, reps):
  fn(); torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps * 1e3


def main():
  thr = int(sys.argv[1]) if len(sys.argv) > 1 else 100
  dev = torch.device("cuda:0")
  g = torch.Generator(device="cpu").manual_seed(0)
  print(f"stream-K threshold {thr} %   (us per launch: off / on, best of 3 interleaved rounds of 10 launches)")
  only = sys.argv[2].split(",") if len(sys.argv) > 2 else list(ROWS)
  for rname, M in ROWS.items():
    if rname not in only:
      continue
    for name, N, K, kind in KINDS:
      x = (torch.randn((M, K), generator=g) ).to(dev).to(BF16) if M * K < 2 ** 28 else torch.randn((M, K), device=dev).to(BF16)
      w = (torch.randn((N, K), generator=g) / K ** 0.5).to(dev).to(BF16)
      b = torch.randn((N,), generator=g).to(dev)
      aux32 = torch.randn((M, N), device=dev) if kind == "res" else None
      aux16 = torch.randn((M, N), device=dev).to(BF16) if kind == "emit" else None
      outs = [torch.empty((M, N), device=dev, dtype=BF16), torch.empty((M, N), device=dev, dtype=BF16),
              torch.empty((M, N), device=dev, dtype=F32) if kind == "res" else None, torch.zeros((N,), device=dev)]
      tiles = (M // 256) * (N // 256)
      res = {}
      best = {0: 1e30, 1: 1e30}
      for rep in range(3):
        for on in (0, 1):
          with ops.option("gemm_streamk", thr if on else 0):
            best[on] = min(best[on], timed(lambda: launch(kind, x, w, b, aux32, aux16, outs), 10))
            if rep == 0:
              res[on] = [t.clone() for t in launch(kind, x, w, b, aux32, aux16, outs)]
              again = launch(kind, x, w, b, aux32, aux16, outs)
              res[on, "same"] = all(torch.equal(p, q) for p, q in zip(res[on], again) if p.numel() > 4096)
      diff = max((p.float() - q.float()).abs().max().item() / max(1e-30, q.float().abs().max().item()) for p, q in zip(res[1], res[0]))
      ident = all(torch.equal(p, q) for p, q in zip(res[1], res[0]))
      print(f"{rname:8s} {name:18s} M={M:6d} N={N:4d} K={K:4d} tiles {tiles:5d} = {tiles / 256:6.2f} rounds: "
            f"{best[0]:8.1f} / {best[1]:8.1f} us  ({best[1] / best[0]:.3f})  max rel diff {diff:.2e}{' identical' if ident else ''}"
            f"  run-to-run {'ok' if res[1, 'same'] else 'DIFFERS'}", flush=True)
      del x, w, aux32, aux16, outs, res
  print("done")


if __name__ == "__main__":
  main()
