"""tools/probes/attention6.hip (8 symmetric waves, one or two key fragments per wave) against attention5.hip (the
one-launch backward of round 4): results and time at the image tower's shapes.  GPU only.  HISTORICAL: it ran while
the kernel sat in libbvhip behind BV_OPT_ATTN_CFG bit 512 (commit "attention6 ..." of round 6); the kernel left the
library after this A/B (profiles/r06_attn6_ab.txt), so bit 512 is a no-op in the current library.

  python tools/attn6_ab.py [n ...]        (default 2048 512)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from big_vision_amd import ops  # noqa: E402
from attn_bench import timeit, dev, BF16  # noqa: E402


def ref_grads(qkv, d_o, n, L, H, rows):
  """fp64 autograd of softmax attention on the first `rows` samples: dqkv [rows * L, 3 H 64] and the bias gradients."""
  x = qkv[:rows * L].double().view(rows, L, 3, H, 64).clone().requires_grad_(True)
  q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
  p = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1)
  o = (p @ v).transpose(1, 2).reshape(rows * L, H * 64)
  o.backward(d_o[:rows * L].double())
  return x.grad.view(rows * L, 3 * H * 64)


def main():
  ns = [int(a) for a in sys.argv[1:]] or [2048, 512]
  for n in ns:
    for L in (196, 197, 208, 193):
      H = 12
      g = torch.Generator(device=dev).manual_seed(L)
      qkv = torch.randn(n * L, 3 * H * 64, device=dev, generator=g).to(BF16)
      d_o = torch.randn(n * L, H * 64, device=dev, generator=g).to(BF16)
      res = {}
      for cfg in (0, 512):
        ops.ctx_set("attn_cfg", cfg)
        o, lse = ops.attn_fwd(qkv, n, L, H)
        for with_bias in (True, False):
          if with_bias and L % 16 == 0 and cfg == 512:
            continue          # (falls back to attention5: nothing new to compare)
          dq = torch.zeros_like(qkv)
          db = torch.zeros(3 * H * 64, device=dev)
          ops.attn_bwd(qkv, o, d_o, lse, n, L, H, dqkv=dq, dbias=db if with_bias else None)
          dq2 = torch.zeros_like(qkv)
          ops.attn_bwd(qkv, o, d_o, lse, n, L, H, dqkv=dq2, dbias=torch.zeros_like(db) if with_bias else None)
          t = timeit(lambda: ops.attn_bwd(qkv, o, d_o, lse, n, L, H, dqkv=dq2, dbias=db.clone() if with_bias else None))
          res[(cfg, with_bias)] = (dq, db, t, bool((dq == dq2).all()))
      ops.ctx_set("attn_cfg", 0)
      rows = 2
      ref = ref_grads(qkv, d_o, n, L, H, rows)
      for with_bias in (True, False):
        if (512, with_bias) not in res:
          continue
        d5, b5, t5, _ = res[(0, with_bias)]
        d6, b6, t6, rep = res[(512, with_bias)]
        rel = lambda a: float((a[:rows * L].double() - ref).norm() / ref.norm())
        relb = float((b6 - b5).norm() / (b5.norm() + 1e-30)) if with_bias else 0.0
        same = float((d6.float() - d5.float()).abs().max())
        print(f"n={n:5d} L={L} bias={int(with_bias)}: attn5 {t5:7.1f} us  attn6 {t6:7.1f} us  ({t6 / t5:.3f}x)   rel-L2 vs fp64: attn5 {rel(d5):.4f} attn6 {rel(d6):.4f}   "
              f"max |attn6 - attn5| {same:.3e}  dbias rel diff {relb:.2e}  run-to-run identical {rep}", flush=True)


if __name__ == "__main__":
  main()
