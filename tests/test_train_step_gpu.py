"""Classification trainer parity (BASELINE configs[0], big_vision/train.py:275-315): the HIP
`big_vision_amd.train.update_fn` vs the fp64 oracle restatement on identical weights and a
synthetic batch.  Tolerances as in test_siglip_step_gpu.py (bf16 MFMA operands / fp32 accumulate
vs fp64): loss rel <= 1e-2, logits max-abs <= 5e-2, per-tensor gradient bounds of tests/_parity.py
(cosine >= 0.999, rel-L2 <= 3e-2, or 2x the measured bf16-operand floor of the tensor)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(**kw):
  from big_vision_amd.compat.ml_collections import ConfigDict
  c = ConfigDict()
  c.lr, c.wd = 1e-3, 1e-4
  c.schedule = dict(warmup_steps=2, decay_type="cosine")
  c.optax_name = "scale_by_adam"
  c.optax = dict(mu_dtype="bfloat16")
  c.grad_clip_norm = 1.0
  c.total_steps = 10
  c.model_name = "vit"
  for k, v in kw.items():
    c[k] = v
  return c


def _run(dev, model_cfg, num_classes, n, res, loss, mixup_a, **extra):
  import bv_oracle as O
  from big_vision_amd import train, utils as u
  config = _cfg(model=model_cfg, num_classes=num_classes, loss=loss, **extra)
  if mixup_a is not None:
    config.mixup = dict(p=0.2, fold_in=None)
  _, model = train.get_model(config)
  g = torch.Generator().manual_seed(3)
  image = torch.rand((n, res, res, 3), generator=g) * 2 - 1
  labels = torch.nn.functional.one_hot(torch.randint(0, num_classes, (n,), generator=g), num_classes).float()
  state, _ = train.make_train_state(model, config, tuple(image.shape), rng=0, total_steps=config.total_steps)
  store = state["params"].store
  for name in store.leaf_names():   # break zero / unit inits (head kernel is zero-init) so all paths carry signal
    leaf = store.leaf(name)
    std = 0.02 if name.endswith("kernel") and "head" in name else (0.05 if name.endswith(("bias", "scale")) else 0.0)
    if std:
      leaf.add_((std * torch.randn(leaf.shape, generator=g)).to(dev))
  store.mark_dirty(); store.refresh_shadow()
  params64 = O.recover_tree([(k, v.detach().cpu().double().clone().requires_grad_(True))
                             for k, v in u.tree_flatten_with_names(state["params"])[0]])
  ocfg = {**O.decode_variant(model_cfg.get("variant")), **{k: v for k, v in model_cfg.items() if k != "variant"}}
  loss_ref, logits_ref = O.classification_step_loss(params64, image.double(), labels.double(), model_cfg=ocfg,
                                                    num_classes=num_classes, loss=loss, mixup_a=mixup_a)
  # forward-only paths
  logits, out = model.apply({"params": state["params"]}, (O.mixup(mixup_a, image)[0] if mixup_a else image).to(dev))
  assert (logits.cpu().double() - logits_ref.detach()).abs().max() <= 5e-2
  assert "pre_logits" in out and out["logits"] is logits
  batch = {"image": image.to(dev), "labels": labels.to(dev)}
  if mixup_a is not None:
    batch["mixup_a"] = mixup_a
  else:
    lf = train.loss_fn(model, state["params"], batch["image"], batch["labels"], config)
    assert abs(lf.item() - loss_ref.item()) <= 1e-2 * abs(loss_ref.item())
  state, meas = train.make_update_fn(model, config)(state, 0, batch)
  assert abs(meas["training_loss"].item() - loss_ref.item()) <= 1e-2 * abs(loss_ref.item())
  loss_ref.backward()
  gref = {k: v.grad for k, v in u.tree_flatten_with_names(params64)[0]}
  gours = {k: v.detach().cpu().double() for k, v in u.tree_flatten_with_names(store.tree("grad"))[0]}
  import _parity
  fl = _parity.bf16_floor(
      lambda p: O.classification_step_loss(p, image.double(), labels.double(), model_cfg=ocfg,
                                           num_classes=num_classes, loss=loss, mixup_a=mixup_a)[0], params64)
  gnorm, _ = _parity.compare_grads(f"classification {model_cfg.get('variant', 'tiny')} {loss} n={n}", gref, gours,
                                   floor=fl)
  assert abs(meas["l2_grads"].item() - gnorm) <= 2e-2 * gnorm
  train.check_finite(meas)


def test_vit_s16_i1k_step(dev):
  """configs/vit_s16_i1k.py: ViT-S/16, rep_size pre_logits, gap pooling, sincos2d posemb,
  softmax_xent on mixed-up one-hot labels, batch 8 of 224x224 (BASELINE configs[0])."""
  _run(dev, dict(variant="S/16", rep_size=True, pool_type="gap", posemb="sincos2d"), 1000, 8, 224,
       "softmax_xent", 0.8)


def test_tiny_sigmoid_xent_step(dev):
  """The trainer's default loss (sigmoid_xent, train.py:299), learned posemb, no mixup."""
  _run(dev, dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="gap"), 24, 6, 64,
       "sigmoid_xent", None)


def test_tiny_step_on_the_bf16_residual_stream(dev):
  """config.residual_stream = "bfloat16" through big_vision.train's update_fn (gap pooling, mixup)."""
  _run(dev, dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="gap"), 24, 6, 64,
       "softmax_xent", 0.7, residual_stream="bfloat16")


def test_ragged_class_count_and_rep_size(dev):
  """num_classes = 10 and rep_size = 12: output widths that are not multiples of 8 (found by running the product on
  the executed-reference fixtures, tests/test_reference_wiring_gpu.py: the GEMM entry point takes N % 8 == 0 only).
  The engine pads such heads with zero columns (engine._W.npad); forward, loss and every gradient vs the oracle."""
  _run(dev, dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="gap", rep_size=12), 10, 6,
       64, "softmax_xent", None)


def test_unknown_loss_raises(dev):
  from big_vision_amd import train
  config = _cfg(model=dict(variant="S/16"), num_classes=10, loss="hinge")
  _, model = train.get_model(config)
  with pytest.raises(AttributeError):
    train.make_update_fn(model, config)
