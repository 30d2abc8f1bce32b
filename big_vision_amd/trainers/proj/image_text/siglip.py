"""SigLIP training step on gfx950 kernels + RCCL, mirroring the reference trainer.

Reference: big_vision/trainers/proj/image_text/siglip.py — `update_fn`
(:271-323) with its inner `loss_fn` (:287-308), and the explicit-collective
statement of the same algorithm in _deprecated_contrastive.py (:117-160,
:308-355).  What is kept: the function names and signatures
(`update_fn(train_state, rng, batch) -> (train_state, measurements)`), the
config fields consumed, the measurement names, the error behaviour.  What is
new: forward, backward, loss, collectives and optimizer are explicit sequences
of libbvhip kernels and RCCL calls (no jit, no autograd).

Data-parallel algorithm ("convention A"): rank r owns
rows [r*n, (r+1)*n) of zimg and ztxt.  all_gather(ztxt) -> local logits
[n, B] with the positive diagonal at column r*n + i -> G = dL/dS with the GLOBAL
1/B -> dzimg local, dztxt_partial [B, E] -> reduce_scatter -> all parameter
gradients are partial sums of the global gradient -> all_reduce(SUM).
"""
from __future__ import annotations

import importlib

import torch

from big_vision_amd import dp
from big_vision_amd import engine as E
from big_vision_amd import ops
from big_vision_amd import optax as bv_optax
from big_vision_amd import utils as u

F32 = torch.float32


def _img_shape(images):
  """Shape that keys the image tower's layout: [n, H, W, 3] pixels, or - NaFlex
  (models/proj/image_text/naflex_vit.py:227) - the patches of the (patches, ptype, yabs, xabs) tuple."""
  return tuple(images[0].shape) if isinstance(images, (tuple, list)) else tuple(images.shape)


def _img_slice(images, a, b):
  return tuple(t[a:b] for t in images) if isinstance(images, (tuple, list)) else images[a:b]


def sigmoid_loss_fwd_bwd(zimg, ztxt, t_param, b_param, comm: dp.Comm):
  """Global-batch pairwise sigmoid loss (siglip.py:291-306) and its gradients.

  zimg, ztxt: this rank's L2-normalised embeddings [n, E] (fp32).
  Returns (stats f64[3] = [loss share, dL/dt', dL/db], dzimg [n,E], dztxt [n,E]).
  """
  n, E = zimg.shape
  B = n * comm.size
  ztxt_all = comm.all_gather_rows(ztxt)
  raw = torch.empty((n, B), device=zimg.device, dtype=F32)
  ops.sgemm(zimg, E, 1, ztxt_all, 1, E, raw, n, B, E)                       # zimg . ztxt_all^T
  stats = torch.zeros(3, device=zimg.device, dtype=torch.float64)
  ops.siglip_loss_(raw, t_param, b_param, stats, comm.rank * n, B)            # raw := G
  dzimg = torch.empty((n, E), device=zimg.device, dtype=F32)
  ops.sgemm(raw, B, 1, ztxt_all, E, 1, dzimg, n, E, B, log_alpha=t_param)    # t G ztxt_all
  dztxt_all = torch.empty((B, E), device=zimg.device, dtype=F32)
  ops.sgemm(raw, 1, B, zimg, E, 1, dztxt_all, B, E, n, log_alpha=t_param)    # t G^T zimg
  dztxt = comm.reduce_scatter_rows(dztxt_all)
  return stats, dzimg, dztxt


def loss_fn(model, params, images, labels, comm=None):
  """Forward-only loss, the `loss_fn(params)` closure of siglip.py:287-308."""
  comm = comm or dp.Comm()
  zimg, ztxt, extras = model.apply({"params": params}, images, labels, train=True, collect=False)
  stats, _, _ = sigmoid_loss_fwd_bwd(zimg, ztxt, extras["t/parameter"], extras.get("b"), comm)
  loss = stats[:1].clone()
  comm.all_reduce_scalars_(loss)
  return loss[0]


def make_train_state(model, config, image_shape, text_shape, *, rng=0, comm=None, total_steps=None,
                     device=None):
  """Parameter store + optimizer laid out for `config` (replaces siglip.py:190-260)."""
  from big_vision_amd.models.vit import _seed_of
  comm = comm or dp.Comm()
  frozen = bv_optax.frozen_leaves(config, model.leaf_names(image_shape, text_shape))
  store = model.make_store(image_shape, text_shape, device=device, frozen_leaves=frozen)
  store.init_random(_seed_of(rng))
  store.refresh_shadow()
  store.want_grads = True
  # parameter placement (siglip.py:196, 232-237): replicated; any other strategy is refused, not ignored
  from big_vision_amd import sharding
  specs = sharding.check_config(config, store.tree(), mesh=comm)
  fsdp = sharding.is_sharded(specs)   # "fsdp" somewhere in config.sharding_strategy: sharded optimizer state / update
  batch_size = config.get("input", {}).get("batch_size", image_shape[0] * comm.size)
  total_steps = total_steps if total_steps is not None else u.steps(
      "total", config, None, batch_size)
  sched_kw = dict(total_steps=total_steps, batch_size=batch_size, data_size=None)
  opt, sched_fns = bv_optax.make(config, store, sched_kw=sched_kw, comm=comm, shard=fsdp)
  return {"params": store.tree(), "opt": opt}, sched_fns


def make_update_fn(model, config, comm=None, loss_fwd_bwd=None, measure=None):
  """Builds `update_fn(train_state, rng, batch)` (siglip.py:271-323).

  loss_fwd_bwd(zimg, ztxt, t_param, b_param, comm) -> (stats f64[3], dzimg, dztxt[, extras]): the
  loss on the local embeddings and its gradients (default: the sigmoid loss of this trainer);
  measure(extras, norms, t_param) -> dict of additional measurements, norms = list of
  (img_norm, txt_norm) per micro-batch.  Both hooks exist for trainers.proj.image_text.contrastive
  (config.loss_fn switch + the pmap trainer's measurement dict)."""
  comm = comm or dp.Comm()
  loss_fwd_bwd = loss_fwd_bwd or sigmoid_loss_fwd_bwd
  assert "mixup" not in config, "Mixup is not supported for SigLIP."
  micro = int(config.get("microbatch", 0) or 0)
  state_cache = {"keep_n": 0, "light": None, "per_ctx": {}}
  if comm.active and config.get("overlap_grad_sync", True):
    dp.warn_if_overlap_is_uncapped()   # the persistent GEMMs leave dp.RESERVED_CUS CUs to RCCL: cap its channels

  towers_drop = any(float(getattr(t, "dropout", 0.0) or 0.0) > 0.0
                    for t in (getattr(model, "image_tower", None), getattr(model, "text_tower", None)))

  def update_fn(train_state, rng, batch):
    images, labels = batch["image"], batch["labels"]
    params, opt = train_state["params"], train_state["opt"]
    # dropout (towers configured with dropout > 0; siglip.py:281-290: the step's key is `rng` folded with the step
    # count, the model draws from rngs={"dropout": rng_model}): one 64-bit key per step, rank and micro-batch; the
    # kernels derive the keep bits of every site from it, so a re-run micro-batch forward repeats its masks
    step_key = None
    if towers_drop:
      if rng is None:
        raise ValueError("the model has dropout > 0: update_fn needs an rng")
      from big_vision_amd.models import vit as _vit
      step_key = E.Dropout(0.0, _vit._seed_of(rng)).key("step", int(opt.count), "rank", int(comm.rank))
    drop_key = lambda s: None if step_key is None else E.Dropout(0.0, step_key).key("microbatch", int(s))
    store = params.store
    store.want_grads = True
    # a checkpoint loaded with store.load_tree() / an in-place edit of the master weights only
    # marks the bf16 shadow dirty; forward and backward read the shadow (no-op when clean)
    store.refresh_shadow()
    store.zero_grad()
    n = _img_shape(images)[0]
    ex = model.executor(store, "", _img_shape(images), tuple(labels.shape))
    # default since round 6: the text tower on a side stream beside the image tower (TwoTowersExec; the towers share
    # nothing until the loss, two_towers.py:56-75; same kernels on the same inputs, identical results).
    # config.tower_streams = 1 puts both towers back on the caller's stream.
    ex.streams = int(config.get("tower_streams", 2) or 1)
    img_frozen = all(e in store.frozen for e in store.entries if e.startswith("img/"))
    txt_frozen = all(e in store.frozen for e in store.entries if e.startswith("txt/"))
    t_param = store.t("t")
    b_param = store.t("b") if "b" in store.entries else None
    # N > 1: the gradient all-reduce overlaps the (last) backward, see dp.GradSync
    # "fsdp" placement: every gradient range is summed onto its owner only (dp.GradShardSync), same overlap
    sharded = getattr(opt, "sharded", False)
    # (comm.active: N > 1 ranks, or a ONE-rank group with forced collectives - bench.py's rank512_rccl and
    # tests/test_dp_nccl_gpu.py run the overlapped path, CU reservation included, exactly as N > 1 does)
    overlap = comm.active and config.get("overlap_grad_sync", True)
    sync = (opt.grad_sync() if sharded else dp.GradSync(comm, store.grad)) if overlap else None

    if micro and n > micro:
      assert n % micro == 0, f"per-device batch {n} not divisible by microbatch {micro}"
      # Pass 1: embeddings of all micro-batches.  The sigmoid loss couples the whole batch,
      # so every tower backward has to wait for all embeddings.  The activations of as many
      # micro-batches as fit in HBM (288 GB on MI355X) are KEPT; only the rest is recomputed
      # in pass 2 (a pure memory/compute trade: results are identical either way).  When the
      # full contexts do not all fit, "light" contexts are kept instead (engine.Block.fwd:
      # LayerNorm outputs and gelu(h) are re-derived by the backward, 1/3 fewer bytes).
      starts = list(range(0, n, micro))
      dev = labels.device
      keep_cfg = config.get("microbatch_keep", "auto")
      keep_max = len(starts) if keep_cfg in ("all", "auto") else (0 if keep_cfg in (0, "none") else int(keep_cfg))
      light_cfg = config.get("microbatch_light", "auto")
      total_mem = torch.cuda.mem_get_info(dev)[1]
      margin = int(0.06 * total_mem)   # pass-2 transients + allocator slack

      def headroom():
        free, _ = torch.cuda.mem_get_info(dev)
        return free + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)

      def fits(per_ctx, count):
        return headroom() - margin >= count * (per_ctx + per_ctx // 32)

      # state_cache["light"]: None = undecided ("auto", first step), False = full contexts, "g" = contexts
      # without gelu(h) (re-emitted by the fc2 dX GEMM), True / "light" = also without the LayerNorm outputs
      if state_cache["light"] is None:
        state_cache["light"] = ({"g": "g", "light": "light"}.get(light_cfg, bool(light_cfg))
                                if light_cfg != "auto" else None)
      mode_of = lambda l: "light" if l in (True, "light") else ("g" if l == "g" else True)
      zi, zt, kept, norms = [], [], {}, []
      for k, s in enumerate(starts):
        mode = mode_of(state_cache["light"])
        per_ctx = state_cache["per_ctx"].get(mode)
        keep = len(kept) < keep_max and (per_ctx is None or keep_cfg == "all" or fits(per_ctx, 1))
        before = torch.cuda.memory_allocated(dev)
        a, b, o_, c = ex.fwd(_img_slice(images, s, s + micro), labels[s:s + micro], save=(mode if keep else False),
                             drop_key=drop_key(s))
        norms.append((o_.get("img/norm"), o_.get("txt/norm")))
        if keep and per_ctx is None:
          per_ctx = state_cache["per_ctx"][mode] = max(1, torch.cuda.memory_allocated(dev) - before)
          if state_cache["light"] is None:
            # first step, "auto": the richest kind of context of which ALL micro-batches fit - full, then
            # without gelu(h), then light (each trial re-runs this micro-batch's forward once)
            others = min(keep_max, len(starts)) - 1
            state_cache["light"] = False
            for trial in ("g", "light"):
              if keep_max <= 1 or fits(per_ctx, others):
                break
              del c
              state_cache["light"] = mode = trial
              before = torch.cuda.memory_allocated(dev)
              a, b, _, c = ex.fwd(_img_slice(images, s, s + micro), labels[s:s + micro], save=mode, drop_key=drop_key(s))
              per_ctx = state_cache["per_ctx"][mode] = max(1, torch.cuda.memory_allocated(dev) - before)
        if keep:
          kept[s] = c
        del c
        zi.append(a); zt.append(b)
      state_cache["keep_n"] = len(kept)
      zimg, ztxt = torch.cat(zi), torch.cat(zt)
      stats, dzimg, dztxt, *lx = loss_fwd_bwd(zimg, ztxt, t_param, b_param, comm)
      # Pass 2: back-propagate every micro-batch's slice of the embedding gradients (grads
      # accumulate in the flat buffer); recompute the forward where it was not kept (those are
      # the LAST micro-batches: by then the kept contexts have been consumed and freed).
      for s in starts:
        ctx = kept.pop(s, None)
        if ctx is None:
          _, _, _, ctx = ex.fwd(_img_slice(images, s, s + micro), labels[s:s + micro],
                                save=mode_of(state_cache["light"]), drop_key=drop_key(s))
        with dp.reserve_cus_for_collectives(comm if (sync is not None and s == starts[-1]) else None):
          ex.bwd(ctx, None if img_frozen else dzimg[s:s + micro].contiguous(),
                 None if txt_frozen else dztxt[s:s + micro].contiguous(),
                 sync=(sync if s == starts[-1] else None))   # gradients are final in the LAST backward only
        del ctx
    else:
      # one pass over the whole per-device batch; config.microbatch_light also applies here (the same context kinds
      # as the two-pass path: True / "light" = LayerNorm outputs and gelu(h) re-derived by the backward, "g" =
      # gelu(h) only) - a memory knob that changes no result
      light_cfg = config.get("microbatch_light", "auto")
      save = "light" if light_cfg in (True, "light") else ("g" if light_cfg == "g" else True)
      zimg, ztxt, o_, ctx = ex.fwd(images, labels, save=save, drop_key=drop_key(0))
      norms = [(o_.get("img/norm"), o_.get("txt/norm"))]
      stats, dzimg, dztxt, *lx = loss_fwd_bwd(zimg, ztxt, t_param, b_param, comm)
      with dp.reserve_cus_for_collectives(comm if sync is not None else None):
        ex.bwd(ctx, None if img_frozen else dzimg, None if txt_frozen else dztxt, sync=sync)

    # dL/dt', dL/db (scalars computed by the loss kernel) into the flat grad buffer.
    gt = store.g("t")
    if gt is not None:
      gt += stats[1].to(F32)
    if b_param is not None and store.g("b") is not None:
      store.g("b").add_(stats[2].to(F32))

    # DP: sum partial gradients and the loss shares (pmean of the reference).
    if sync is not None:
      sync.finish()
    elif sharded:
      late = opt.grad_sync()       # overlap_grad_sync = False: the whole buffer after the backward
      if late is not None:
        late.finish()
    else:
      comm.all_reduce_sum_(store.grad)
    loss = stats[:1].clone()
    comm.all_reduce_scalars_(loss)

    measurements = {"training_loss": loss[0]}
    if measure is not None:
      measurements.update(measure(lx[0] if lx else {}, norms, t_param))
    # deferred input validation of the towers (BERT input_mask): raise BEFORE the update is applied - the weights of a
    # refused batch stay untouched (advisor r4: the last batch of a run was never validated)
    ex.check_inputs()
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
  update_fn.state_cache = state_cache   # diagnostics: how many micro-batches keep their activations
  return update_fn


def make_predict_fn(model):
  """`predict_fn(train_state, batch)` handed to the evaluators (siglip.py:388-392): either of
  batch["image"] / batch["labels"] may be missing."""
  def predict_fn(train_state, batch):
    zimg, ztxt, out = model.apply({"params": train_state["params"]}, batch.get("image", None),
                                  batch.get("labels", None), collect=False)
    return zimg, ztxt, out
  return predict_fn


def check_finite(measurements):
  """NaN/Inf abort of siglip.py:468-470 (synchronises)."""
  for k, v in measurements.items():
    if not torch.isfinite(torch.as_tensor(v)).all():
      raise RuntimeError(f"measurement '{k}' is not finite: {v}")


def get_model(config):
  """Model registry by module path (siglip.py:190-191)."""
  # the reference spells it big_vision.models.<name>; the `big_vision` package of this repo maps that
  # to the same module object
  model_mod = importlib.import_module(f"big_vision_amd.models.{config.model_name}")
  return model_mod, model_mod.Model(**config.get("model", {}))
