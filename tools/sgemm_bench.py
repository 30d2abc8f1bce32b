"""A/B of bv_sgemm_strided's two kernels (VALU 64x64 tiles vs fp32 MFMA) on the loss's three products.  GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from big_vision_amd import ops, _lib

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
  for _ in range(warm): fn()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e3


def main():
  lib = _lib.load()
  E = 768
  for n, B in ((4096, 4096), (2048, 4096), (512, 4096), (512, 512)):
    zi = torch.randn(n, E, device=dev); zt = torch.randn(B, E, device=dev); G = torch.randn(n, B, device=dev) * 0.01
    raw = torch.empty(n, B, device=dev); dzi = torch.empty(n, E, device=dev); dzt = torch.empty(B, E, device=dev)
    row = [f"n={n} B={B}"]
    for path in (0, 1):
      ops.ctx_set("sgemm_mfma", path)
      t1 = timeit(lambda: ops.sgemm(zi, E, 1, zt, 1, E, raw, n, B, E))
      t2 = timeit(lambda: ops.sgemm(G, B, 1, zt, E, 1, dzi, n, E, B))
      t3 = timeit(lambda: ops.sgemm(G, 1, B, zi, E, 1, dzt, B, E, n))
      fl = 2.0 * n * B * E / 1e6
      row.append(f"{'mfma' if path else 'valu'}: logits {t1:7.1f} us ({fl / t1:5.1f} TF) dzimg {t2:7.1f} ({fl / t2:5.1f}) dztxt {t3:7.1f} ({fl / t3:5.1f})")
    ops.ctx_set("sgemm_mfma", 1)
    print(" | ".join(row), flush=True)


if __name__ == "__main__":
  main()
