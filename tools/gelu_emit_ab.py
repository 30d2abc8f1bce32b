"""A/B for VERDICT r5 item 6 (ii): the fc2 dX GEMM of a LIGHT context re-emits gelu(h) in its epilogue
(BV_EPI_GELU_BWD_EMIT: reads h, writes dH and g = gelu(h): three [M, 3072] streams beside the matmul).  Alternative:
the plain GELU' epilogue (BV_EPI_GELU_BWD: reads h, writes dH) + a separate HBM-roof elementwise kernel for g.

  python tools/gelu_emit_ab.py [M ...]        (default 401408 131072: image / text tokens of one 2048-pair micro-batch)

The elementwise stand-in is torch's F.gelu(h, approximate="tanh") on bf16 (one read + one write of [M, 3072]; its
achieved TB/s is printed so that "HBM-roof" can be checked) - not bit-identical to mlp_act_words, which does not matter
for a timing A/B."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from big_vision_amd import ops  # noqa: E402

BF16 = torch.bfloat16


def timeit(fn, iters=10):
  fn(); fn()
  torch.cuda.synchronize()
  best = 1e30
  for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
      fn()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / iters)
  return best * 1e3   # us


def main():
  dev = torch.device("cuda", 0)
  Ms = [int(a) for a in sys.argv[1:]] or [401408, 131072]
  D, Mh = 768, 3072
  print(f"# fc2 dX of the MLP backward, [M, {D}] x [{D}, {Mh}], aux = bf16 pre-activation h, fused Dense_0 bias-gradient column sums")
  print("#      M | EMIT (one launch) us | GELU' (no emit) us | elementwise gelu us (TB/s) | GELU' + elementwise us | EMIT / (GELU' + elementwise)")
  for M in Ms:
    g = torch.Generator(device=dev).manual_seed(0)
    dy = (torch.randn((M, D), generator=g, device=dev) * 0.1).to(BF16)
    w2 = (torch.randn((D, Mh), generator=g, device=dev) * 0.03).to(BF16)   # dX = dy [M, D] x W2^T: B = W2^T as [K = D][N = Mh]
    h = torch.randn((M, Mh), generator=g, device=dev).to(BF16)
    dh = torch.empty((M, Mh), device=dev, dtype=BF16)
    gout = torch.empty((M, Mh), device=dev, dtype=BF16)
    cs = torch.zeros(Mh, device=dev)
    w2t = w2.t().contiguous()           # [N = Mh][K = D]: the k-major operand the engine hands over (natural Flax layout of Dense_1)
    emit = lambda: ops.gemm(dy, w2t, a_kmajor=True, b_kmajor=True, out=dh, epilogue=ops.EPI_GELU_BWD_EMIT, aux=h, out2=gout, colsum=cs)
    plain = lambda: ops.gemm(dy, w2t, a_kmajor=True, b_kmajor=True, out=dh, epilogue=ops.EPI_GELU_BWD, aux=h, colsum=cs)
    ew = lambda: F.gelu(h, approximate="tanh", out=None)
    t_emit, t_plain, t_ew = timeit(emit), timeit(plain), timeit(ew)
    tbs = 2 * M * Mh * 2 / (t_ew * 1e-6) / 1e12
    print(f"{M:8d} | {t_emit:10.1f} | {t_plain:10.1f} | {t_ew:8.1f} ({tbs:.2f}) | {t_plain + t_ew:10.1f} | {t_emit / (t_plain + t_ew):.3f}", flush=True)
    del dy, w2, h, dh, gout


if __name__ == "__main__":
  main()
