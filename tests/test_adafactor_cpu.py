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


def test_clipping_threshold_is_taken():
  """scale_by_adafactor(clipping_threshold=...) (optax.py:190,208: optax.clip_by_block_rms per leaf) raised until round 5;
  the batched step now applies it (GPU parity: tests/test_adafactor_gpu.py[...-clip])."""
  store = ParamStore([Entry("k", (64, 64), init_zeros)], "cpu")
  opt, _ = bv_optax.make(_cfg(optax=dict(clipping_threshold=1.0)), store, sched_kw=dict(global_batch_size=1, total_steps=1))
  assert opt.af["block_rms_clip"] == 1.0
  opt0, _ = bv_optax.make(_cfg(), ParamStore([Entry("k", (64, 64), init_zeros)], "cpu"), sched_kw=dict(global_batch_size=1, total_steps=1))
  assert opt0.af["block_rms_clip"] == 0.0


def _af_store(scan_depth=0):
  D, H = 64, 2
  views = {f"blk/attn/{n}/kernel": (1, i) for i, n in enumerate(("query", "key", "value"))}
  ents = [Entry("blk/attn/qkv/kernel", (D, 3, H, 32), init_zeros, views),
          Entry("blk/attn/out/kernel", (H, 32, D), init_zeros),
          Entry("blk/mlp/kernel", (D, 96), init_zeros),
          Entry("blk/ln/scale", (D,), init_zeros),
          Entry("t", (1,), init_zeros)]
  return ParamStore(ents, "cpu")


def test_adafactor_state_tree_roundtrip(tmp_path):
  """`state_tree()` / `load_state_tree()` / `u.save_train_state` for the Adafactor
  optimizer.  The reference's optax state is masked(chain(scale_by_factored_rms, identity, ema)):
  `1/0/0/{0 count, 1 v_row, 2 v_col, 3 v}/<leaf>` + `1/0/2/{0 count, 1 ema}` (optax.py:187-216, leaf naming
  utils.py:616-641).  Random state -> tree (shapes as optax keeps them) -> .npz -> fresh optimizer: every
  statistic, the bf16 momentum and the count come back bit for bit."""
  from big_vision_amd import utils as u
  store = _af_store()
  opt, _ = bv_optax.make(_cfg(), store, sched_kw=dict(global_batch_size=1, total_steps=4))
  g = torch.Generator().manual_seed(3)
  opt.af_state.copy_(torch.rand(opt.af_state.shape, generator=g))
  # the padding between leaves is never read or written by the kernels; zero it for the equality below
  mask = torch.zeros_like(opt.af_state, dtype=torch.bool)
  for lf in opt.af_leaves:   # (the trailing rcm[B] of a factored leaf is per-step scratch: mean_R(v_row), rebuilt by the row pass)
    n_real = lf["B"] * (lf["R"] + lf["C"]) if lf["factored"] else lf["n_state"]
    mask[lf["soff"]:lf["soff"] + n_real] = True
  opt.af_state.mul_(mask)
  opt.mu.copy_(torch.randn(opt.mu.shape, generator=g).to(opt.mu.dtype))
  opt.count = 7
  tree = opt.state_tree()
  flat = dict(u.tree_flatten_with_names(tree)[0])
  # shapes as optax keeps them: v_row drops d0, v_col drops d1, placeholders are (1,)
  assert tuple(flat["1/0/0/1/blk/attn/query/kernel"].shape) == (2, 32)       # [H, Dh]  (D dropped)
  assert tuple(flat["1/0/0/2/blk/attn/query/kernel"].shape) == (64, 2)       # [D, H]   (Dh dropped)
  assert tuple(flat["1/0/0/3/blk/attn/query/kernel"].shape) == (1,)
  assert tuple(flat["1/0/0/1/blk/mlp/kernel"].shape) == (64,) and tuple(flat["1/0/0/2/blk/mlp/kernel"].shape) == (96,)
  assert tuple(flat["1/0/0/3/blk/ln/scale"].shape) == (64,) and tuple(flat["1/0/0/1/blk/ln/scale"].shape) == (1,)
  assert int(flat["1/0/0/0"]) == 7 and int(flat["1/0/2/0"]) == 7
  assert tuple(flat["1/0/2/1/blk/attn/key/kernel"].shape) == (64, 2, 32) and flat["1/0/2/1/blk/attn/key/kernel"].dtype == torch.bfloat16
  f = str(tmp_path / "af_state.npz")
  u.save_train_state(f, {"params": store.tree(), "opt": opt})
  store2 = _af_store()
  opt2, _ = bv_optax.make(_cfg(), store2, sched_kw=dict(global_batch_size=1, total_steps=4))
  # (u.load_train_state = store.load_tree + refresh of the bf16 shadow on the GPU + this call)
  opt2.load_state_tree({k[len("opt/"):]: v for k, v in u.npload(f).items() if k.startswith("opt/")})
  assert opt2.count == 7
  assert torch.equal(opt2.af_state, opt.af_state)
  mu_a, mu_b = (dict(u.tree_flatten_with_names(o._moment_tree(o.mu))[0]) for o in (opt, opt2))   # (the flat buffer has alignment gaps)
  assert mu_a.keys() == mu_b.keys() and all(torch.equal(mu_a[k], mu_b[k]) for k in mu_a)
  # state_dict form too
  opt3, _ = bv_optax.make(_cfg(), _af_store(), sched_kw=dict(global_batch_size=1, total_steps=4))
  opt3.load_state_dict(opt.state_dict())
  assert torch.equal(opt3.af_state, opt.af_state) and opt3.count == 7
  # a wrong shape is refused, not reshaped
  bad = dict(flat); bad["1/0/0/1/blk/mlp/kernel"] = torch.zeros(96)
  with pytest.raises(ValueError, match="Shape mismatch"):
    opt2.load_state_tree(bad)


def test_adafactor_scan_leaves_factor_like_the_stacked_leaf():
  """optax factors the STACKED leaf of a scan=True model.  depth < min_dim_size_to_factor
  never enters the factored pair (same statistics per block; the state tree stacks them over depth); a depth
  that would (>= 32 with a [depth, D] bias) is refused instead of silently diverging."""
  def store_of(depth):
    ents = [e for i in range(depth) for e in (Entry(f"enc/encoderblock_{i}/k/kernel", (64, 96), init_zeros),
                                              Entry(f"enc/encoderblock_{i}/k/bias", (96,), init_zeros))]
    return ParamStore(ents, "cpu", scan_prefixes=("enc",))
  st = store_of(3)
  opt, _ = bv_optax.make(_cfg(), st, sched_kw=dict(global_batch_size=1, total_steps=4))
  from big_vision_amd import utils as u
  flat = dict(u.tree_flatten_with_names(opt.state_tree())[0])
  assert tuple(flat["1/0/0/1/enc/encoderblock/k/kernel"].shape) == (3, 64)      # v_row of [3, 64, 96]: drops d0 = 96
  assert tuple(flat["1/0/0/2/enc/encoderblock/k/kernel"].shape) == (3, 96)
  assert tuple(flat["1/0/0/3/enc/encoderblock/k/bias"].shape) == (3, 96)        # unfactored [3, 96]
  assert tuple(flat["1/0/0/1/enc/encoderblock/k/bias"].shape) == (1,)
  opt.load_state_tree(flat)
  with pytest.raises(NotImplementedError, match="depth axis"):
    bv_optax.make(_cfg(), store_of(40), sched_kw=dict(global_batch_size=1, total_steps=4))


def test_per_example_clipping_is_refused_with_the_reference_line():
  """optax.py:100-103: grad_clip_per_example needs per-example gradient trees (clip_by_per_example_global_norm,
  optax.py:54-72); no trainer on this path produces them, so the option raises instead of clipping the batch
  gradient and calling it per-example.  The math it would apply (known answer, 2 examples): clip each
  example's global norm to max_norm, then average."""
  store = ParamStore([Entry("k", (64, 64), init_zeros)], "cpu")
  for name in ("big_vision.scale_by_adafactor", "scale_by_adam"):
    with pytest.raises(NotImplementedError, match=r"optax.py:100-103"):
      bv_optax.make(_cfg(optax_name=name, grad_clip_norm=1.0, grad_clip_per_example=True), store,
                    sched_kw=dict(global_batch_size=1, total_steps=1))
  g = torch.tensor([[3.0, 4.0], [0.3, 0.4]])                       # norms 5 and 0.5, max_norm 1
  want = (g[0] / 5.0 + g[1]) / 2
  assert torch.allclose(O.clip_by_per_example_global_norm([g], 1.0)[0], want)
