"""Classification training step on gfx950 kernels + RCCL, mirroring `big_vision/train.py`.

Reference: big_vision/train.py — `update_fn` (:275-315): optional mixup
(`u.get_mixup`, utils.py:1146-1154), `loss_fn` = `getattr(u, config.get("loss",
"sigmoid_xent"))(logits, labels)` on the logits of `model.apply` (:295-300),
`jax.value_and_grad`, `tx.update` + `optax.apply_updates`, and the measurements
`training_loss, l2_grads, l2_params, l2_updates` (:307-313).  This is BASELINE
configs[0] (ViT-S/16 on ImageNet-1k, `configs/vit_s16_i1k.py`).

Kept: names and signatures (`update_fn(train_state, rng, batch) -> (train_state,
measurements)`), the config fields consumed (`model_name, model, num_classes, loss,
mixup, optax_name, optax, lr, wd, schedule, grad_clip_norm`), error behaviour.
New: forward, backward, loss and optimizer are explicit libbvhip kernel sequences
(big_vision_amd.models.vit.VitExec); no jit, no autograd.

Data parallelism (sharding.py:83-101, replicated parameters, batch split on dim 0):
every rank normalises its loss / dlogits by the GLOBAL batch size, so parameter
gradients are partial sums and one all-reduce(SUM) finishes them (dp.GradSync).
Mixup rolls the batch along dim 0 on every device independently, exactly like the
reference's shard_map (:283-289).
"""
from __future__ import annotations

import importlib

import torch

from big_vision_amd import dp
from big_vision_amd import engine as E
from big_vision_amd import ops
from big_vision_amd import optax as bv_optax
from big_vision_amd import utils as u
from big_vision_amd.params import ParamStore

F32 = torch.float32
LOSSES = {"softmax_xent": ops.softmax_xent, "sigmoid_xent": ops.sigmoid_xent}


def get_model(config):
  """Model registry by module path (train.py:191-192)."""
  model_mod = importlib.import_module(f"big_vision_amd.models.{config.model_name}")
  return model_mod, model_mod.Model(num_classes=config.num_classes, **config.get("model", {}))


def _loss_kernel(config):
  name = config.get("loss", "sigmoid_xent")
  if name not in LOSSES:
    raise AttributeError(f"module 'big_vision.utils' has no loss '{name}' on the accelerated path "
                         f"(available: {sorted(LOSSES)})")
  return LOSSES[name]


def make_train_state(model, config, image_shape, *, rng=0, comm=None, total_steps=None, device=None):
  """Parameter store + optimizer laid out for `config` (replaces train.py:191-262)."""
  from big_vision_amd.models.vit import _seed_of
  comm = comm or dp.Comm()
  device = device or torch.device("cuda", torch.cuda.current_device())
  hw = model.grid(tuple(image_shape))
  from big_vision_amd.params import external_leaf_names, scan_name
  ents = model.entries("", hw)
  sp = model.scan_prefixes()
  leaves = external_leaf_names([leaf for e in ents for leaf, _ in e.flax_leaves()], sp)
  frozen_leaves = bv_optax.frozen_leaves(config, leaves)
  frozen = set()
  for e in ents:
    hits = [scan_name(leaf, sp)[0] in frozen_leaves for leaf, _ in e.flax_leaves()]
    if any(hits) and not all(hits):
      raise NotImplementedError(f"fused tensor {e.name} is only partially frozen")
    if all(hits):
      frozen.add(e.name)
  store = ParamStore(ents, device, frozen=frozen, scan_prefixes=sp)
  store.init_random(_seed_of(rng))
  store.refresh_shadow()
  store.want_grads = True
  from big_vision_amd import sharding     # train.py:201-203: replicate, or fsdp = sharded optimizer state / update
  fsdp = sharding.is_sharded(sharding.check_config(config, store.tree(), mesh=comm))
  batch_size = config.get("input", {}).get("batch_size", image_shape[0] * comm.size)
  total_steps = total_steps if total_steps is not None else u.steps("total", config, None, batch_size)
  opt, sched_fns = bv_optax.make(config, store, sched_kw=dict(total_steps=total_steps, batch_size=batch_size,
                                                               data_size=None), comm=comm, shard=fsdp)
  return {"params": store.tree(), "opt": opt}, sched_fns


def loss_fn(model, params, images, labels, config, comm=None):
  """Forward-only loss, the `loss_fn(params)` closure of train.py:295-300 (no mixup)."""
  comm = comm or dp.Comm()
  store = params.store
  store.refresh_shadow()
  ex = model.executor(store, "", model.grid(tuple(images.shape)))
  logits, _, _ = ex.fwd(images, save=False)
  acc = torch.zeros(1, device=images.device, dtype=torch.float64)
  _loss_kernel(config)(logits.contiguous(), labels.to(F32).contiguous(), acc, want_grad=False,
                       n_global=images.shape[0] * comm.size)
  comm.all_reduce_scalars_(acc)
  return acc[0]


def make_update_fn(model, config, comm=None):
  """Builds `update_fn(train_state, rng, batch)` (train.py:275-315)."""
  comm = comm or dp.Comm()
  loss_kernel = _loss_kernel(config)
  mix_p = float(config.get("mixup", {}).get("p", 0.0) or 0.0) if config.get("mixup") else 0.0

  def update_fn(train_state, rng, batch):
    images, labels = batch["image"], batch["labels"]
    params, opt = train_state["params"], train_state["opt"]
    store = params.store
    store.want_grads = True
    store.refresh_shadow()   # no-op when clean; picks up load_tree() / in-place master edits
    store.zero_grad()
    images = images.to(F32).contiguous()
    labels = labels.to(F32).contiguous()
    if mix_p:
      # train.py:283-289: one coefficient per step (fold_in(rng, step)), the roll is per device
      a = batch.get("mixup_a") if isinstance(batch, dict) else None
      if a is None:
        a = u.get_mixup_coefficient(rng, bv_optax.get_count(opt), mix_p)
      images, labels = ops.mixup(images, a), ops.mixup(labels, a)
    n = images.shape[0]
    ex = model.executor(store, "", model.grid(tuple(images.shape)))
    # dropout (vit.py:228,100,109,76; train.py:298 rngs={"dropout": rng_model}): one key per step and rank
    drop = None
    if float(getattr(model, "dropout", 0.0) or 0.0) > 0.0:
      if rng is None:
        raise ValueError("the model has dropout > 0: update_fn needs an rng")
      drop = E.Dropout(model.dropout, _seed_of(rng)).fold("step", int(bv_optax.get_count(opt)), "rank", int(comm.rank))
    logits, _, ctx = ex.fwd(images, save=True, drop=drop)
    acc = torch.zeros(1, device=images.device, dtype=torch.float64)
    dlogits = loss_kernel(logits.contiguous(), labels, acc, want_grad=True, n_global=n * comm.size)
    # ("fsdp" placement: every range is summed onto its owner only, dp.GradShardSync)
    sync = None
    if comm.size > 1:
      sync = opt.grad_sync() if getattr(opt, "sharded", False) else dp.GradSync(comm, store.grad)
    ex.bwd(ctx, dlogits)
    if sync is not None:
      sync.finish()
    comm.all_reduce_scalars_(acc)
    measurements = {"training_loss": acc[0]}
    measurements.update(opt.step())
    return {"params": params, "opt": opt}, measurements

  # config.residual_stream ("float32" = the reference's arithmetic, "bfloat16"): set for the duration of
  # each step and restored afterwards, so model.apply / other trainers in the process keep their own
  stream = config.get("residual_stream", "float32")
  inner = update_fn

  def update_fn(train_state, rng, batch):
    old = E.set_residual_stream(stream)
    try:
      return inner(train_state, rng, batch)
    finally:
      E.set_residual_stream(old)
  return update_fn


def check_finite(measurements):
  """NaN/Inf abort of train.py:452-454 (synchronises)."""
  for k, v in measurements.items():
    if not torch.isfinite(torch.as_tensor(v)).all():
      raise RuntimeError(f"measurement '{k}' is not finite: {v}")
