"""What would a bf16 residual stream cost against the parity bounds?  CPU only (the fp64 oracle).

The product keeps the residual stream in fp32 like the reference (models/vit.py: fp32 activations; only
the matmul inputs are cast).  A bf16 stream would halve the LayerNorm traffic and the +residual GEMM
epilogues (~65 ms of the 750 ms step).  This script measures, on the ViT-B/16 +
text-B model, the per-tensor gradient error of three arithmetics against fp64:
  (a) bf16 GEMM / attention operands, fp32 everything else          = the product (its noise floor)
  (b) (a) + the residual stream rounded to bf16 after every add     = the design option
and prints the worst tensors next to SURVEY 8c's bounds (cosine >= 0.999, rel-L2 <= 3e-2).

    python tools/bf16_residual_budget.py [n]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bv_oracle as O  # noqa: E402


def grads(params, image, text, okw, modes):
  p = O.tree_map(lambda v: v.detach().clone().requires_grad_(True), params)
  import contextlib
  with contextlib.ExitStack() as st:
    for m in modes:
      st.enter_context(m())
    loss, (zimg, ztxt, logits, _) = O.siglip_step_loss(p, image, text, **okw)
    loss.backward()
  return float(loss.detach()), {k: v.grad for k, v in O.tree_flatten_with_names(p)}, (zimg.detach(), ztxt.detach(), logits.detach())


def main():
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
  image_cfg = dict(variant="B/16", pool_type="map")
  text_cfg = dict(variant="B", vocab_size=32_000)
  okw = dict(image_cfg=image_cfg, text_cfg=text_cfg, out_dim=(None, 768))
  params = O.init_two_towers(0, (224, 224), 64, image_cfg=image_cfg, text_cfg=text_cfg, out_dim=(None, 768),
                             temperature_init=10.0, bias_init=-10.0, dtype=torch.float64)
  gen = torch.Generator().manual_seed(7)
  params = O.recover_tree([(k, v + 0.05 * torch.randn(v.shape, generator=gen, dtype=v.dtype) if k.endswith(("bias", "scale")) else v)
                           for k, v in O.tree_flatten_with_names(params)])
  image, text = O.synthetic_batch(1, n, 224, 64, 32_000, dtype=torch.float64)
  l0, g0, f0 = grads(params, image, text, okw, [])
  gnorm = sum(float((g * g).sum()) for g in g0.values()) ** 0.5
  for name, modes in (("(a) bf16 operands", [O.bf16_operands]),
                      ("(b) bf16 operands + bf16 residual stream", [O.bf16_operands, O.bf16_residual])):
    l1, g1, f1 = grads(params, image, text, okw, modes)
    print(f"{name}: forward max-abs errors: zimg {float((f1[0]-f0[0]).abs().max()):.2e} ztxt {float((f1[1]-f0[1]).abs().max()):.2e} "
          f"logits {float((f1[2]-f0[2]).abs().max()):.3f} (bounds 2e-2 / 2e-2 / 0.25), loss rel {abs(l1-l0)/abs(l0):.1e} (1e-2)")
    rows = []
    for k in g0:
      a, b = g0[k].flatten(), g1[k].flatten()
      if float(a.norm()) < 1e-3 * gnorm:
        continue
      rel = float((a - b).norm() / a.norm())
      cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
      rows.append((rel, cos, k))
    rows.sort(reverse=True)
    over = sum(1 for r in rows if r[0] > 3e-2 or r[1] < 0.999)
    print(f"{name}: loss {l1:.6f} (fp64 {l0:.6f}); {len(rows)} tensors, worst rel-L2 {rows[0][0]:.4f}, "
          f"min cosine {min(r[1] for r in rows):.5f}, {over} outside the bounds (rel-L2 <= 0.03, cos >= 0.999)")
    for rel, cos, k in rows[:5]:
      print(f"    {rel:.4f}  cos {cos:.5f}  {k}")


if __name__ == "__main__":
  main()
