"""BASELINE configs[1] ("C2"): the ViT-B/16 image tower ALONE, forward + backward, against the fp64
oracle restatement of models/vit.py:206-276.  The tower is driven the way bench.py's C2 workload
drives it: executor.fwd(save=True) -> embeddings z, a synthetic upstream gradient dL/dz of the
loss L = 0.5 * mean_n |z|^2 ("use a sum z^2-style synthetic upstream grad"),
executor.bwd.  Checked: z (max-abs 2e-2 at unit scale), every parameter gradient (tests/_parity.py
bounds, bf16-operand floor measured for the case)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _tower_case(dev, cfg, n, res, case):
  import bv_oracle as O
  import _parity
  from big_vision_amd import utils as u
  from big_vision_amd.models import vit
  from big_vision_amd.params import ParamStore

  model = vit.Model(None, **cfg)
  hw = model.grid((n, res, res, 3))
  store = ParamStore(model.entries("", hw), dev, scan_prefixes=model.scan_prefixes())
  store.init_random(0)
  g = torch.Generator().manual_seed(11)
  for name in store.leaf_names():
    if name.endswith(("bias", "scale", "cls")):
      leaf = store.leaf(name)
      leaf.add_((0.05 * torch.randn(leaf.shape, generator=g)).to(dev))
  store.mark_dirty(); store.refresh_shadow()
  store.want_grads = True
  store.zero_grad()
  image = torch.rand((n, res, res, 3), generator=g) * 2 - 1
  params64 = O.recover_tree([(k, v.detach().cpu().double().clone().requires_grad_(True))
                             for k, v in u.tree_flatten_with_names(store.tree())[0]])
  ocfg = {**O.decode_variant(cfg.get("variant")), **{k: v for k, v in cfg.items() if k != "variant"}}

  def loss_of(p):
    z, _ = O.vit_forward(p, image.double(), num_classes=None, **ocfg)
    return 0.5 * (z ** 2).sum() / n, z

  loss_ref, z_ref = loss_of(params64)
  ex = model.executor(store, "", hw)
  z, _, ctx = ex.fwd(image.to(dev), save=True)
  scale = max(1.0, z_ref.abs().max().item())
  assert (z.cpu().double() - z_ref.detach()).abs().max().item() <= 2e-2 * scale
  ex.bwd(ctx, (z / n).contiguous())
  torch.cuda.synchronize()
  loss_ref.backward()
  gref = {k: v.grad for k, v in u.tree_flatten_with_names(params64)[0]}
  gours = {k: v.detach().cpu().double() for k, v in u.tree_flatten_with_names(store.tree("grad"))[0]}
  fl = _parity.bf16_floor(lambda p: loss_of(p)[0], params64)
  _parity.compare_grads(case, gref, gours, floor=fl)


def test_vit_b16_tower_fwd_bwd(dev):
  """ViT-B/16@224, MAP pooling (the SigLIP image tower), n = 8."""
  _tower_case(dev, dict(variant="B/16", pool_type="map"), 8, 224, "C2 ViT-B/16 MAP tower n=8")


def test_vit_b16_tower_gap_sincos(dev):
  """Same tower with gap pooling and the sincos2d position table (no pos_embedding parameter)."""
  _tower_case(dev, dict(variant="B/16", pool_type="gap", posemb="sincos2d", depth=4), 4, 224,
              "C2 ViT-B/16 gap/sincos2d depth4 n=4")
