"""Contrastive image-text training step with the reference's `config.loss_fn` switch.

Reference: big_vision/trainers/proj/image_text/_deprecated_contrastive.py (historical name
`contrastive.py`, README.md:83, configs/proj/clippo/train_clippo.py:34): `loss_fn` :317-339 selects
`softmax_loss` :80-101 / `sigmoid_loss` :117-160 / `chunked_sigmoid_loss` :168-200 by
`config.get("loss_fn", "softmax")` and returns, next to the loss, the measurement dict
{t, t/parameter, train/nimg, train/ntxt, train/<loss extras>}; `update_fn` :308-355 pmeans loss,
measurements and gradients over the devices.

Here (one process per GPU, convention A of SURVEY.md App. A):
  * sigmoid / chunked_sigmoid: the kernels of trainers.proj.image_text.siglip (fp32 [n, B] logits
    per rank, fused loss + dL/dS kernel).  The chunked variant of the reference is the same
    function of the same inputs computed in n x n chunks to bound a TPU core's memory; a rank here
    already holds only its [n, B] slice, so both names run the same launch sequence and differ in
    the extras they report (the chunked variant of the reference reports no global negative
    statistics, :194-200).
  * softmax (CLIP): i2t and t2i cross-entropies over the gathered axis with the positives on the
    diagonal, each direction on its own [n, B] logits (all_gather of the other modality),
    bv_softmax_xent for loss + dL/dlogits, strided fp32 GEMMs and reduce_scatter for the embedding
    gradients, dL/dt' = sum G o logits.
  * measurements: per-device logit statistics from bv_logit_stats (one pass over the logits before
    the loss kernel overwrites them), averaged over ranks as the reference's pmean does.
"""
from __future__ import annotations

import torch

from big_vision_amd import dp
from big_vision_amd import ops
from big_vision_amd.trainers.proj.image_text import siglip as _siglip

F32 = torch.float32
LOSSES = ("softmax", "sigmoid", "chunked_sigmoid")


def sigmoid_loss(zimg, ztxt, t_param, b_param, comm: dp.Comm, chunked=False):
  """-> (stats f64[3] = [loss share, dL/dt', dL/db], dzimg, dztxt, extras) on this rank's embeddings."""
  n, E = zimg.shape
  B = n * comm.size
  ztxt_all = comm.all_gather_rows(ztxt)
  raw = torch.empty((n, B), device=zimg.device, dtype=F32)
  ops.sgemm(zimg, E, 1, ztxt_all, 1, E, raw, n, B, E)
  ls = ops.logit_stats(raw, t_param, b_param, comm.rank * n)        # before raw becomes dL/dS
  names = ops.LOGIT_STATS_NAMES[:6] if chunked else ops.LOGIT_STATS_NAMES
  extras = {k: ls[i] for i, k in enumerate(names)}
  stats = torch.zeros(3, device=zimg.device, dtype=torch.float64)
  ops.siglip_loss_(raw, t_param, b_param, stats, comm.rank * n, B)
  dzimg = torch.empty((n, E), device=zimg.device, dtype=F32)
  ops.sgemm(raw, B, 1, ztxt_all, E, 1, dzimg, n, E, B, log_alpha=t_param)
  dztxt_all = torch.empty((B, E), device=zimg.device, dtype=F32)
  ops.sgemm(raw, 1, B, zimg, E, 1, dztxt_all, B, E, n, log_alpha=t_param)
  return stats, dzimg, comm.reduce_scatter_rows(dztxt_all), extras


def chunked_sigmoid_loss(zimg, ztxt, t_param, b_param, comm: dp.Comm):
  return sigmoid_loss(zimg, ztxt, t_param, b_param, comm, chunked=True)


_onehot_cache = {}


def _diag_labels(n, B, offset, device):
  key = (n, B, offset, str(device))
  if key not in _onehot_cache:
    y = torch.zeros((n, B), device=device, dtype=F32)
    y[torch.arange(n, device=device), offset + torch.arange(n, device=device)] = 1.0
    _onehot_cache.clear()
    _onehot_cache[key] = y
  return _onehot_cache[key]


def softmax_loss(zimg, ztxt, t_param, b_param, comm: dp.Comm):
  """CLIP loss, 0.5 (i2t + t2i) (:80-101); the bias is not part of this loss."""
  del b_param
  n, E = zimg.shape
  B = n * comm.size
  dev = zimg.device
  y = _diag_labels(n, B, comm.rank * n, dev)
  stats = torch.zeros(3, device=dev, dtype=torch.float64)
  loss2 = torch.zeros(1, device=dev, dtype=torch.float64)
  extras, grads = {}, {}
  for name, row, col in (("i2t", zimg, ztxt), ("t2i", ztxt, zimg)):
    col_all = comm.all_gather_rows(col)
    logits = torch.empty((n, B), device=dev, dtype=F32)
    ops.sgemm(row, E, 1, col_all, 1, E, logits, n, B, E, log_alpha=t_param)       # t row . col_all^T
    l_dir = torch.zeros(1, device=dev, dtype=torch.float64)
    g = ops.softmax_xent(logits, y, l_dir, n_global=B)                            # dL_dir / dlogits (mean over B)
    ops.dot_(g, logits, stats[1:2])                                               # dL/dt' (x 0.5 below)
    drow = torch.empty((n, E), device=dev, dtype=F32)
    ops.sgemm(g, B, 1, col_all, E, 1, drow, n, E, B, alpha=0.5, log_alpha=t_param)
    dcol_all = torch.empty((B, E), device=dev, dtype=F32)
    ops.sgemm(g, 1, B, row, E, 1, dcol_all, B, E, n, alpha=0.5, log_alpha=t_param)
    grads[name] = (drow, comm.reduce_scatter_rows(dcol_all))
    loss2 += l_dir
    # diagnostics only (host-side reductions of an [n, B] matrix the loss already produced)
    acc = (logits.argmax(1) == comm.rank * n + torch.arange(n, device=dev)).float().mean()
    extras[f"{name}_acc"] = acc
    extras[f"{name}_loss"] = l_dir[0] * comm.size      # per-device mean, as the reference reports it
  stats[0] = 0.5 * loss2[0]
  stats[1] *= 0.5
  dzimg = grads["i2t"][0] + grads["t2i"][1]
  dztxt = grads["t2i"][0] + grads["i2t"][1]
  return stats, dzimg, dztxt, extras


def _loss_impl(config):
  name = config.get("loss_fn", "softmax")
  if name == "softmax":
    return softmax_loss
  if name == "sigmoid":
    return sigmoid_loss
  if name == "chunked_sigmoid":
    return chunked_sigmoid_loss
  raise NotImplementedError(f"Unrecognized loss config.loss_fn={name!r}")


def _measurements(extras, norms, t_param, comm):
  """{t, t/parameter, train/nimg, train/ntxt, train/<extras>} averaged over ranks (:333-342)."""
  out = {"t": torch.exp(t_param[0]), "t/parameter": t_param[0].clone()}
  ni = [a.float().mean() for a, _ in norms if a is not None]
  nt = [b.float().mean() for _, b in norms if b is not None]
  loc = {}
  if ni:
    loc["train/nimg"] = torch.stack(ni).mean()
  if nt:
    loc["train/ntxt"] = torch.stack(nt).mean()
  loc.update({f"train/{k}": v.float() for k, v in extras.items()})
  if loc:
    keys = sorted(loc)
    vec = torch.stack([loc[k].reshape(()) for k in keys]).double()
    comm.all_reduce_scalars_(vec)
    vec /= comm.size
    out.update({k: vec[i].float() for i, k in enumerate(keys)})
  return out


def loss_fn(model, params, images, labels, config, comm=None):
  """Forward-only `loss_fn(params, images, labels)` (:317-339): (loss, measurements)."""
  comm = comm or dp.Comm()
  zimg, ztxt, extras = model.apply({"params": params}, images, labels, train=True, collect=False)
  stats, _, _, lx = _loss_impl(config)(zimg, ztxt, extras["t/parameter"], extras.get("b"), comm)
  loss = stats[:1].clone()
  comm.all_reduce_scalars_(loss)
  meas = _measurements(lx, [(extras.get("img/norm"), extras.get("txt/norm"))], extras["t/parameter"], comm)
  return loss[0], meas


def make_update_fn(model, config, comm=None):
  """`update_fn(train_state, rng, batch) -> (train_state, measurements)` (:308-355) with the loss of
  `config.loss_fn`; micro-batching, frozen towers, gradient sync and the optimizer are the ones of
  trainers.proj.image_text.siglip."""
  comm = comm or dp.Comm()
  impl = _loss_impl(config)
  if impl is not softmax_loss and getattr(model, "bias_init", None) is None:
    raise ValueError("the sigmoid losses need the model's bias parameter (config.model.bias_init)")
  return _siglip.make_update_fn(model, config, comm=comm, loss_fwd_bwd=impl,
                                measure=lambda lx, norms, t: _measurements(lx, norms, t, comm))


make_train_state = _siglip.make_train_state
make_predict_fn = _siglip.make_predict_fn
check_finite = _siglip.check_finite
get_model = _siglip.get_model
