"""Dropout > 0 on the accelerated path (models/vit.py:76,100,109,228; text_transformer.py:72-75) vs the oracle.

JAX's random stream cannot be reproduced, so parity is stated GIVEN THE MASKS: the product's kernels derive the keep
bits of every site from (site key, element index); the tests read those bits back (`ops.dropout_mask`, checked bit for bit
against the oracle's Philox restatement, which Random123's known answers pin in tests/test_dropout_cpu.py) and hand the same
masks to `bv_oracle` - whose dropout PLACEMENT is pinned by the executed reference (tests/golden/refwiring_*dropout*).
Bounds are the end-to-end ones of tests/_parity.py (per-tensor gradient cosine >= 0.999, rel-L2 <= 3e-2)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


def test_mask_kernel_equals_the_philox_restatement(dev):
  import bv_oracle as O
  from big_vision_amd import ops
  for key, count, rate in [(0x0123456789ABCDEF, 4096, 0.1), (2 ** 64 - 1, 8 * 197 * 64, 0.25), (5, 4, 0.5), (77, 1 << 20, 0.0)]:
    got = ops.dropout_mask((count,), key, rate, dev).cpu().numpy()
    want = O.dropout_keep_mask(key, count, rate)
    assert np.array_equal(got, want), (key, count, rate, int((got != want).sum()))
  m = ops.dropout_mask((1 << 22,), 99, 0.3, dev).float()
  assert abs(m.mean().item() - 0.7) < 2e-3
  # neighbouring elements / groups are uncorrelated
  assert abs(((m[1:] * m[:-1]).mean() - 0.49).item()) < 2e-3


def test_dropout_kernels_apply_the_exported_mask(dev):
  from big_vision_amd import ops
  g = torch.Generator().manual_seed(3)
  rows, cols, rate, key = 394, 768, 0.2, 0xDEADBEEF12345678
  x = torch.randn((rows, cols), generator=g).to(dev)
  res = torch.randn((rows, cols), generator=g).to(dev)
  keep = ops.dropout_mask((rows, cols), key, rate, dev)
  scale = 1.0 / (1.0 - rate)
  want = res + torch.where(keep, x * scale, torch.zeros_like(x))
  y = ops.dropout_f32(x, key, rate, addend=res)
  assert torch.allclose(y, want, rtol=1e-6, atol=1e-6)
  yb = ops.dropout_f32(x, key, rate, out_bf16=torch.empty((rows, cols), device=dev, dtype=BF16))
  assert torch.equal(yb, torch.where(keep, x * scale, torch.zeros_like(x)).to(BF16))
  x2 = x.clone()
  assert ops.dropout_f32(x2, key, rate, out=x2) is x2                     # in place
  assert torch.equal(x2, torch.where(keep, x * scale, torch.zeros_like(x)))
  a, b = x.to(BF16), res.to(BF16)
  a0, b0 = a.clone(), b.clone()
  ops.dropout_bf16_(a, key, rate, b=b)
  assert torch.equal(a, torch.where(keep, a0.float() * scale, torch.zeros_like(x)).to(BF16))
  assert torch.equal(b, torch.where(keep, b0.float() * scale, torch.zeros_like(x)).to(BF16))
  assert torch.equal(ops.dropout_f32(x, key, 0.0), x)                     # rate 0 keeps everything
  with pytest.raises(RuntimeError):
    ops.dropout_f32(x, key, 1.0)


def tower_masks(drop, n, L, D, M, depth, posemb, dev):
  """{oracle site name: keep mask} of one tower from the product's site keys (engine.DROP_*, Dropout.fold("block", i))."""
  from big_vision_amd import engine as E
  from big_vision_amd import ops
  out = {}
  if posemb:
    out["posemb"] = ops.dropout_mask((n, L, D), drop.key(E.DROP_POSEMB), drop.rate, dev).cpu()
  for i in range(depth):
    b = drop.fold("block", i)
    out[f"block{i}/sa"] = ops.dropout_mask((n, L, D), b.key(E.DROP_SA), drop.rate, dev).cpu()
    out[f"block{i}/gelu"] = ops.dropout_mask((n, L, M), b.key(E.DROP_GELU), drop.rate, dev).cpu()
    out[f"block{i}/mlp"] = ops.dropout_mask((n, L, D), b.key(E.DROP_MLP), drop.rate, dev).cpu()
  return out


def _jitter(store, dev):
  g = torch.Generator().manual_seed(7)
  for name in store.leaf_names():
    if name.endswith(("bias", "scale", "cls")):
      leaf = store.leaf(name)
      leaf.add_((0.05 * torch.randn(leaf.shape, generator=g)).to(dev))
  store.mark_dirty()
  store.refresh_shadow()


@pytest.mark.parametrize("pool,scan", [("tok", False), ("map", True)])
def test_vit_tower_with_dropout_matches_the_oracle(dev, pool, scan):
  """Forward and every parameter gradient of an image tower in train mode, dropout 0.25, through the executor (what
  train.update_fn runs); rate 0 / train=False stay deterministic; the same key repeats the same masks."""
  import bv_oracle as O
  import _parity
  from big_vision_amd import engine as E
  from big_vision_amd import utils as u
  from big_vision_amd.models import vit
  from big_vision_amd.params import ParamStore
  cfg = dict(num_classes=10, width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(8, 8), pool_type=pool, dropout=0.25,
             scan=scan, head_zeroinit=False)
  n, res = 8, 32
  model = vit.Model(**cfg)
  hw = model.grid((n, res, res, 3))
  store = ParamStore(model.entries("", hw), dev, scan_prefixes=model.scan_prefixes())
  store.init_random(0)
  store.want_grads = True
  store.ensure_grad()
  _jitter(store, dev)
  store.zero_grad()
  image = (torch.rand((n, res, res, 3), generator=torch.Generator().manual_seed(1)) * 2 - 1)
  L = hw[0] * hw[1] + (1 if pool == "tok" else 0)
  drop = E.Dropout(cfg["dropout"], 0xABCDEF)
  ex = model.executor(store, "", hw)
  logits, _, ctx = ex.fwd(image.to(dev), save=True, drop=drop)
  logits2, _, _ = ex.fwd(image.to(dev), save=False, drop=E.Dropout(cfg["dropout"], 0xABCDEF))
  assert torch.equal(logits, logits2), "the same key must repeat the same masks"
  logits3, _, _ = ex.fwd(image.to(dev), save=False, drop=E.Dropout(cfg["dropout"], 0xABCDF0))
  assert not torch.equal(logits, logits3)
  w = torch.randn(logits.shape, generator=torch.Generator().manual_seed(2)).to(dev)
  ex.bwd(ctx, w.clone())

  masks = tower_masks(drop, n, L, cfg["width"], cfg["mlp_dim"], cfg["depth"], True, dev)
  params64 = O.recover_tree([(k, v.detach().cpu().double().clone().requires_grad_(True))
                             for k, v in u.tree_flatten_with_names(store.tree())[0]])
  dm = O.DropMasks(cfg["dropout"], masks)
  okw = {k: v for k, v in cfg.items() if k not in ("scan", "head_zeroinit")}
  y_ref, out_ref = O.vit_forward(params64, image.double(), **okw, drop=dm)
  assert sorted(dm.used) == sorted(masks), "the oracle did not consume every mask"
  # the `out` dict of a train-mode pass (8b contract): "sa" / "mlp" are the branch outputs BEFORE their dropout
  # (vit.py:98,108), "with_posemb" the input of the dropout behind the position embedding (:220,228) - and collecting
  # them changes nothing downstream
  logits_c, out_c, _ = ex.fwd(image.to(dev), save=False, collect=True, drop=E.Dropout(cfg["dropout"], 0xABCDEF))
  assert torch.equal(logits_c, logits)
  def close(a, b, what, rel=6e-2, floor=2e-3):
    a, b = a.detach().double().cpu().reshape(b.shape), b.detach().double()
    assert (a - b).abs().max().item() <= rel * b.pow(2).mean().sqrt().item() + floor, what
  close(out_c["with_posemb"], out_ref["with_posemb"], "with_posemb")
  for i in range(cfg["depth"]):
    for k in ("sa", "+sa", "mlp", "+mlp"):
      close(out_c["encoder"][f"block{i:02d}"][k], out_ref["encoder"][f"block{i:02d}"][k], f"block{i:02d}/{k}")
  # (the dropped branch is NOT what is published: about a quarter of its elements are zero)
  assert (out_c["encoder"]["block00"]["sa"] == 0).float().mean().item() < 0.01
  assert (logits.cpu().double() - y_ref.detach()).abs().max() <= 5e-2 * max(1.0, y_ref.abs().max().item())
  (y_ref * w.cpu().double()).sum().backward()
  gref = {k: v.grad for k, v in u.tree_flatten_with_names(params64)[0] if v.grad is not None}
  gours = {k: v.detach().cpu().double() for k, v in u.tree_flatten_with_names(store.tree("grad"))[0]}
  _parity.compare_grads(f"vit tower dropout 0.25 pool={pool} scan={scan}", gref, gours)
  # deterministic modes are untouched: train=False ignores the rate
  y_eval, _ = model.apply({"params": store.tree()}, image.to(dev), train=False)
  y_eval_ref, _ = O.vit_forward(params64, image.double(), **okw)
  assert (y_eval.cpu().double() - y_eval_ref.detach()).abs().max() <= 5e-2 * max(1.0, y_eval_ref.abs().max().item())
  y_tr, _ = model.apply({"params": store.tree()}, image.to(dev), train=True, rngs={"dropout": 11}, collect=False)
  assert not torch.equal(y_tr, y_eval)


@pytest.mark.parametrize("micro,keep", [(0, "auto"), (4, "all"), (4, 0)])
def test_siglip_step_with_dropout_matches_the_oracle(dev, micro, keep):
  """A whole two-tower training step (image tower dropout 0.1, text tower 0.3) through siglip.update_fn - one pass,
  micro-batches with kept contexts, micro-batches whose forward is RE-RUN in pass 2 (the masks must come out again) -
  against the oracle given the masks of every micro-batch."""
  import bv_oracle as O
  import _parity
  from big_vision_amd import engine as E
  from big_vision_amd import utils as u
  from big_vision_amd.compat.ml_collections import ConfigDict
  from big_vision_amd.models import vit
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  image_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(8, 8), pool_type="map", dropout=0.1)
  text_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, vocab_size=100, dropout=0.3)
  Eo, n, res, seq = 128, 8, 32, 16
  model = two_towers.Model(image=image_cfg, text=text_cfg, out_dim=(None, Eo), temperature_init=10.0, bias_init=-10.0)
  c = ConfigDict()
  c.lr, c.wd = 1e-3, 1e-2
  c.schedule = dict(decay_type="cosine", warmup_steps=2)
  c.optax_name = "scale_by_adam"
  c.grad_clip_norm = 1.0
  c.total_steps = 10
  if micro:
    c.microbatch, c.microbatch_keep = micro, keep
  image, text = O.synthetic_batch(1, n, res, seq, 100)
  state, _ = siglip.make_train_state(model, c, tuple(image.shape), tuple(text.shape), rng=0, total_steps=10)
  store = state["params"].store
  _jitter(store, dev)
  params64 = O.recover_tree([(k, v.detach().cpu().double().clone().requires_grad_(True))
                             for k, v in u.tree_flatten_with_names(state["params"])[0]])
  rng = 1234
  with pytest.raises(ValueError):
    siglip.make_update_fn(model, c)(state, None, {"image": image.to(dev), "labels": text.to(dev)})
  update_fn = siglip.make_update_fn(model, c)
  state, meas = update_fn(state, rng, {"image": image.to(dev), "labels": text.to(dev)})

  # the masks of this step: step key (rng, step count 0, rank 0) -> micro-batch -> tower -> site
  step_key = E.Dropout(0.0, vit._seed_of(rng)).key("step", 0, "rank", 0)
  starts = list(range(0, n, micro)) if micro else [0]
  mb = micro or n
  Li, Lt = (res // 8) ** 2, seq
  per = {"img": [], "txt": []}
  for s in starts:
    dk = E.Dropout(0.0, step_key).key("microbatch", s)
    per["img"].append(tower_masks(E.Dropout(image_cfg["dropout"], dk).fold("img"), mb, Li, 128, 256, 2, True, dev))
    per["txt"].append(tower_masks(E.Dropout(text_cfg["dropout"], dk).fold("txt"), mb, Lt, 128, 256, 2, False, dev))
  full = {t: {k: torch.cat([m[k] for m in per[t]], 0) for k in per[t][0]} for t in per}
  drop = {"img": O.DropMasks(image_cfg["dropout"], full["img"]), "txt": O.DropMasks(text_cfg["dropout"], full["txt"])}
  loss_ref, _ = O.siglip_step_loss(params64, image.double(), text, image_cfg=image_cfg, text_cfg=text_cfg, out_dim=(None, Eo),
                                   drop=drop)
  assert sorted(drop["img"].used) == sorted(full["img"]) and sorted(drop["txt"].used) == sorted(full["txt"])
  loss_ref.backward()
  assert abs(meas["training_loss"].item() - loss_ref.item()) <= 1e-2 * abs(loss_ref.item())
  gref = {k: v.grad for k, v in u.tree_flatten_with_names(params64)[0] if v.grad is not None}
  gours = {k: v.detach().cpu().double() for k, v in u.tree_flatten_with_names(store.tree("grad"))[0]}
  gnorm, _ = _parity.compare_grads(f"siglip step dropout img 0.1 / txt 0.3 micro={micro} keep={keep}", gref, gours)
  assert abs(meas["l2_grads"].item() - gnorm) <= 2e-2 * gnorm
  # WITHOUT the masks the oracle's gradients are far away: the step really dropped something
  p0 = O.recover_tree([(k, v.detach().clone().requires_grad_(True)) for k, v in u.tree_flatten_with_names(params64)[0]])
  O.siglip_step_loss(p0, image.double(), text, image_cfg=image_cfg, text_cfg=text_cfg, out_dim=(None, Eo))[0].backward()
  k = "img/Transformer/encoderblock_0/MlpBlock_0/Dense_1/kernel"
  plain = dict(u.tree_flatten_with_names(p0)[0])[k].grad
  assert (plain - gref[k]).norm() > 0.1 * gref[k].norm()
