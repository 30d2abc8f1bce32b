"""BigVision Adafactor (big_vision/optax.py:187-216) - host side and oracle, no GPU: which axes get
factored (optax `_factored_dims`), the reference's own known answer for the state size
(optax_test.py:320-337), the leaf views handed to bv_adafactor_leaf, and the oracle restatement
against closed forms of the first two steps."""
import math

import numpy as np
import pytest
import torch

import bv_oracle as O
from big_vision_amd import optax as bv_optax
from big_vision_amd.compat.ml_collections import ConfigDict
from big_vision_amd.engine import init_zeros
from big_vision_amd.params import Entry, ParamStore


def test_factored_dims_follow_optax():
  fd = bv_optax.factored_dims
  assert fd((768, 3072)) == (0, 1) and fd((3072, 768)) == (1, 0)
  assert fd((768, 12, 64)) == (2, 0)           # q/k/v kernels: 64 and 768
  assert fd((12, 64, 768)) == (1, 2)           # out kernel
  assert fd((1, 196, 768)) == (1, 2)           # pos_embedding
  assert fd((16, 16, 3, 768)) is None          # stem conv: second largest axis 16 < 32
  assert fd((32, 32, 3, 768)) == (1, 3)
  assert fd((768,)) is None and fd((1, 1, 768)) is None and fd((31, 31)) is None
  for shape in [(768, 3072), (768, 12, 64), (12, 64, 768), (16, 16, 3, 768), (5, 40, 33), (64, 64)]:
    assert fd(shape) == O.adafactor_factored_dims(shape)


def _cfg(**kw):
  c = ConfigDict()
  c.optax_name = "big_vision.scale_by_adafactor"
  c.lr = 0.01
  c.schedule = dict(decay_type="linear")
  for k, v in kw.items():
    c[k] = v
  return c


def test_reference_known_answer_state_size():
  """optax_test.py:320-337: one 1024 x 1024 kernel -> FactoredState holds 2 * 1024 + 2 numbers."""
  store = ParamStore([Entry("Dense_0/kernel", (1024, 1024), init_zeros)], "cpu")
  opt, _ = bv_optax.make(_cfg(), store, sched_kw=dict(global_batch_size=1, total_steps=1))
  assert opt.adafactor_state_numel() == 2 * 1024 + 2
  assert opt.mu.dtype == torch.bfloat16 and opt.mu.numel() == store.trainable_count      # ema accumulator


def test_leaf_views_of_fused_tensors():
  """query/key/value kernels are strided views of the fused [D,3,H,64] tensor: each gets its own
  factored statistics over (Dh = 64, D) with the head axis as batch."""
  D, H = 128, 2
  views = {f"attn/{n}/kernel": (1, i) for i, n in enumerate(("query", "key", "value"))}
  store = ParamStore([Entry("attn/qkv/kernel", (D, 3, H, 64), init_zeros, views),
                      Entry("attn/out/kernel", (H, 64, D), init_zeros),
                      Entry("ln/scale", (D,), init_zeros)], "cpu")
  opt, _ = bv_optax.make(_cfg(), store, sched_kw=dict(global_batch_size=1, total_steps=1))
  by = {l["leaf"]: l for l in opt.af_leaves}
  q, k = by["attn/query/kernel"], by["attn/key/kernel"]
  assert q["factored"] and q["dims"] == (2, 0) and (q["B"], q["R"], q["C"]) == (H, 64, D)
  v = list(q["view"])
  assert v[0] == 0 and v[1:5] == [1, H, 64, D] and v[6:9] == [64, 1, 3 * H * 64]       # strides: head, d_head, D
  assert list(k["view"])[0] == H * 64                                                 # key slice starts one [H,64] block later
  o = by["attn/out/kernel"]
  assert o["dims"] == (1, 2) and list(o["view"])[1:5] == [1, H, 64, D] and list(o["view"])[6:9] == [64 * D, D, 1]
  ln = by["ln/scale"]
  assert not ln["factored"] and list(ln["view"])[1:5] == [1, 1, 1, D]
  vr, vc, vf = opt.adafactor_state_of("attn/query/kernel")
  assert tuple(vr.shape) == (H, 64) and tuple(vc.shape) == (D, H) and tuple(vf.shape) == (1,)
  vr, vc, vf = opt.adafactor_state_of("ln/scale")
  assert tuple(vf.shape) == (D,) and tuple(vr.shape) == (1,)


def test_oracle_first_steps_closed_form():
  g = torch.Generator().manual_seed(0)
  params = {"w": torch.randn((40, 48), generator=g, dtype=torch.float64), "b": torch.randn((48,), generator=g, dtype=torch.float64)}
  grads = {"w": torch.randn((40, 48), generator=g, dtype=torch.float64), "b": torch.randn((48,), generator=g, dtype=torch.float64)}
  cfg = dict(optax_name="big_vision.scale_by_adafactor", lr=0.1, schedule=dict(decay_type="cosine", warmup_steps=0),
             optax=dict(dtype_momentum="float32"))
  tx = O.OptaxOracle(cfg, params, sched_kw=dict(total_steps=10, batch_size=1))
  u0 = tx.update(grads, params)
  gw = grads["w"]
  g2 = gw * gw + 1e-30
  vrow, vcol = g2.mean(1), g2.mean(0)               # step 0: decay = 1 - 1^-0.8 = 0
  want = gw / torch.sqrt(vrow / vrow.mean())[:, None] / torch.sqrt(vcol)[None, :]
  lr0 = 0.1 * 0.5 * (1 + math.cos(0.0))
  assert torch.allclose(u0["w"], -lr0 * 0.1 * want, rtol=1e-12)          # ema from zero: 0.1 * u
  assert torch.allclose(u0["b"], -lr0 * 0.1 * torch.sign(grads["b"]), rtol=1e-9)
  # step 1: decay = 1 - 2^-0.8, same gradients -> the statistics are unchanged, the EMA grows
  u1 = tx.update(grads, params)
  lr1 = 0.1 * 0.5 * (1 + math.cos(math.pi * 1 / 10))
  assert torch.allclose(u1["w"], -lr1 * (0.9 * 0.1 + 0.1) * want, rtol=1e-12)
  assert abs((1 - 2 ** -0.8) - 0.42565) < 1e-4


def test_unsupported_options_raise():
  store = ParamStore([Entry("k", (64, 64), init_zeros)], "cpu")
  with pytest.raises(NotImplementedError, match="clip_by_block_rms"):
    bv_optax.make(_cfg(optax=dict(clipping_threshold=1.0)), store, sched_kw=dict(global_batch_size=1, total_steps=1))
