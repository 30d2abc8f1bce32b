"""trainers.proj.image_text.contrastive on the GPU: every `config.loss_fn` against the oracle's
restatement of _deprecated_contrastive.py (:80-101 softmax, :117-160 sigmoid + measurement dict,
:168-200 chunked) - loss, the gradients w.r.t. both embeddings, dL/dt', dL/db, the extras - and
the trainer's measurement names (:333-339).  Tolerances: fp32 kernels vs fp64, rtol 1e-4."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _unit(n, e, seed):
  g = torch.Generator().manual_seed(seed)
  z = torch.randn((n, e), generator=g, dtype=torch.float64)
  return z / z.norm(dim=-1, keepdim=True)


@pytest.mark.parametrize("loss", ["sigmoid", "chunked_sigmoid", "softmax"])
def test_losses_match_the_oracle(dev, loss):
  import bv_oracle as O
  from big_vision_amd import dp
  from big_vision_amd.trainers.proj.image_text import contrastive
  n, E = 24, 64
  zi, zt = _unit(n, E, 1).requires_grad_(True), _unit(n, E, 2).requires_grad_(True)
  tp = torch.tensor([math.log(7.0)], dtype=torch.float64, requires_grad=True)
  bp = torch.tensor([-3.0], dtype=torch.float64, requires_grad=True)
  t = torch.exp(tp[0])
  if loss == "softmax":
    ref = O.softmax_loss_per_device(zi, zt, [zi], [zt], 0, t)
  elif loss == "sigmoid":
    ref = O.sigmoid_loss_per_device(zi, [zt], 0, t, bp[0])
  else:
    ref = O.chunked_sigmoid_loss_per_device(zi, [zt], 0, t, bp[0])
  ref.backward()
  impl = contrastive._loss_impl({"loss_fn": loss})
  stats, dzi, dzt, extras = impl(zi.detach().float().to(dev), zt.detach().float().to(dev),
                                 tp.detach().float().to(dev), bp.detach().float().to(dev), dp.Comm())
  torch.cuda.synchronize()
  assert abs(stats[0].item() - ref.item()) <= 1e-4 * abs(ref.item()), (stats[0].item(), ref.item())
  assert (dzi.cpu().double() - zi.grad).abs().max() <= 1e-4 * zi.grad.abs().max()
  assert (dzt.cpu().double() - zt.grad).abs().max() <= 1e-4 * zt.grad.abs().max()
  assert abs(stats[1].item() - tp.grad.item()) <= 1e-4 * abs(tp.grad.item()) + 1e-7
  if loss != "softmax":
    assert abs(stats[2].item() - bp.grad.item()) <= 1e-4 * abs(bp.grad.item())
    want = O.sigmoid_logit_stats_per_device(zi.detach(), [zt.detach()], 0, t.detach(), bp[0].detach())
    keys = list(want)[:6] if loss == "chunked_sigmoid" else list(want)
    assert set(extras) == set(keys)
    for k in keys:
      assert abs(extras[k].item() - want[k].item()) <= 1e-4 * max(1.0, abs(want[k].item())), k
  else:
    assert set(extras) == {"i2t_acc", "i2t_loss", "t2i_acc", "t2i_loss"}
    logits = (zi.detach() @ zt.detach().T) * t.detach()
    assert abs(extras["i2t_acc"].item() - (logits.argmax(1) == torch.arange(n)).double().mean().item()) < 1e-6
    l_i2t = -(torch.diagonal(logits) - torch.logsumexp(logits, -1)).mean()
    assert abs(extras["i2t_loss"].item() - l_i2t.item()) <= 1e-4 * abs(l_i2t.item())


@pytest.mark.parametrize("loss", ["sigmoid", "softmax"])
def test_update_fn_measurements(dev, loss):
  """One step of the contrastive trainer on a toy two-tower model: measurement names of the
  reference (:333-351) and the loss against the oracle forward."""
  import bv_oracle as O
  from big_vision_amd import utils as u
  from big_vision_amd.compat.ml_collections import ConfigDict
  from big_vision.models.proj.image_text import two_towers          # the reference's import path
  from big_vision.trainers.proj.image_text import contrastive
  image_cfg = dict(width=128, depth=1, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="map")
  text_cfg = dict(width=128, depth=1, mlp_dim=256, num_heads=2, vocab_size=64)
  model = two_towers.Model(image=image_cfg, text=text_cfg, out_dim=(None, 128), temperature_init=10.0, bias_init=-10.0)
  c = ConfigDict()
  c.lr, c.wd, c.optax_name, c.total_steps, c.grad_clip_norm = 1e-3, 1e-2, "scale_by_adam", 10, 1.0
  c.schedule = dict(decay_type="cosine", warmup_steps=2)
  c.loss_fn = loss
  image, text = O.synthetic_batch(3, 8, 48, 8, 64)
  state, _ = contrastive.make_train_state(model, c, tuple(image.shape), tuple(text.shape), rng=0, total_steps=10)
  p64 = O.recover_tree([(k, v.detach().cpu().double()) for k, v in u.tree_flatten_with_names(state["params"])[0]])
  zi, zt, out = O.two_towers_forward(p64, image.double(), text, image_cfg=image_cfg, text_cfg=text_cfg, out_dim=(None, 128))
  t = torch.exp(p64["t"][0])
  ref = O.softmax_loss_per_device(zi, zt, [zi], [zt], 0, t) if loss == "softmax" else \
      O.sigmoid_loss_per_device(zi, [zt], 0, t, p64["b"][0])
  state, meas = contrastive.make_update_fn(model, c)(state, None, {"image": image.to(dev), "labels": text.to(dev)})
  assert abs(meas["training_loss"].item() - ref.item()) <= 1e-2 * abs(ref.item())
  base = {"training_loss", "t", "t/parameter", "train/nimg", "train/ntxt", "l2_grads", "l2_params", "l2_updates"}
  extra = {"train/i2t_acc", "train/i2t_loss", "train/t2i_acc", "train/t2i_loss"} if loss == "softmax" else \
      {f"train/{k}" for k in O.sigmoid_logit_stats_per_device(zi, [zt], 0, t, p64["b"][0])}
  assert set(meas) == base | extra, set(meas) ^ (base | extra)
  assert abs(meas["t"].item() - 10.0) < 1e-4 and abs(meas["train/nimg"].item() - out["img/norm"].mean().item()) < 2e-2
  with pytest.raises(NotImplementedError):
    contrastive.make_update_fn(model, ConfigDict(dict(c.to_dict(), loss_fn="triplet")))
