"""Regression tests for the round-1 advisor findings (host logic, CPU):
  * update_fn refreshes a dirty bf16 shadow (checkpoint loaded after make_train_state);
  * resample_posemb only reads the SHAPE of the model-side tensor (a device view of the store
    cannot be converted to numpy);
  * ad-hoc `apply` trees are re-loaded on every call (no id()-keyed stale cache, one store per
    geometry);
  * pool_type="none" is a forward-only pass-through (vit.py:252-253) instead of a late ValueError;
  * `load` falls back to model_cfg.scan when there is no init tree;
  * optimizer state round-trips under the reference's optax state names (train-state .npz)."""
import collections
import os

import numpy as np
import pytest
import torch

from big_vision_amd import _lib, ops, params as P, utils as u
from big_vision_amd.compat.ml_collections import ConfigDict
from big_vision_amd.models import vit
from big_vision_amd.models.proj.image_text import text_transformer, two_towers
from big_vision_amd.trainers.proj.image_text import siglip

IMG = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="tok")
TXT = dict(width=128, depth=1, mlp_dim=256, num_heads=2, vocab_size=50)


@pytest.fixture()
def dry(monkeypatch):
  calls = []
  monkeypatch.setattr(_lib, "call", lambda name, *a: calls.append((name, a)))
  monkeypatch.setattr(ops, "_chk", lambda t, dtype, name: t)
  monkeypatch.setattr(ops, "_stream", lambda: 0)
  monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a: (1 << 40, 1 << 40))
  monkeypatch.setattr(torch.cuda, "memory_reserved", lambda *a: 0)
  monkeypatch.setattr(torch.cuda, "memory_allocated", lambda *a: 0)
  monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
  return calls


def _cfg(**kw):
  c = ConfigDict()
  c.lr, c.wd, c.optax_name, c.total_steps, c.grad_clip_norm = 1e-3, 1e-2, "scale_by_adam", 10, 1.0
  c.schedule = dict(decay_type="cosine", warmup_steps=2)
  for k, v in kw.items():
    c[k] = v
  return c


def test_update_fn_refreshes_a_dirty_shadow(dry):
  """LiT flow: make_train_state, then load a checkpoint into the (frozen) image tower.  The next
  update_fn must cast master -> bf16 shadow before any forward kernel reads it."""
  model = two_towers.Model(image=IMG, text=TXT, out_dim=(None, 64), temperature_init=10.0, bias_init=-2.71)
  cfg = _cfg(schedule=[("img/.*", None), (".*", dict(decay_type="cosine", warmup_steps=2))])
  image = torch.zeros((4, 32, 32, 3)); text = torch.ones((4, 8), dtype=torch.int32)
  state, _ = siglip.make_train_state(model, cfg, tuple(image.shape), tuple(text.shape), rng=0, total_steps=10,
                                     device="cpu")
  store = state["params"].store
  assert not store._shadow_dirty
  fn = siglip.make_update_fn(model, cfg)
  dry.clear()
  fn(state, None, {"image": image, "labels": text})
  assert not any(n == "bv_cast_bf16" and a[2] == store.count for n, a in dry), "clean shadow must not be re-cast"
  tree = u.tree_map(lambda v: np.asarray(v) + 1.0, dict(state["params"]))
  store.load_tree(tree)
  assert store._shadow_dirty
  dry.clear()
  fn(state, None, {"image": image, "labels": text})
  names = [n for n, _ in dry]
  cast_full = [i for i, (n, a) in enumerate(dry) if n == "bv_cast_bf16" and a[2] == store.count]
  assert cast_full, "update_fn did not refresh the bf16 shadow of the loaded weights"
  first_fwd = min(i for i, n in enumerate(names) if n in ("bv_patchify", "bv_patchify_ld", "bv_embed_fwd", "bv_gemm_bf16"))
  assert cast_full[0] < first_fwd
  assert not store._shadow_dirty


class _ShapeOnly:
  """Stands in for a CUDA view of the flat store: has a shape, refuses numpy conversion."""

  def __init__(self, shape):
    self.shape = shape

  def __array__(self, *a, **k):
    raise TypeError("can't convert cuda:0 device type tensor to numpy")


def test_resample_posemb_reads_only_the_shape_of_the_model_tensor():
  old = np.random.RandomState(0).randn(1, 16, 8).astype(np.float32)
  out = vit.resample_posemb(old, _ShapeOnly((1, 64, 8)))
  assert out.shape == (1, 64, 8)
  same = vit.resample_posemb(old, _ShapeOnly((1, 16, 8)))
  assert same is old or np.array_equal(same, old)


def test_vit_load_with_a_store_bound_init_tree(tmp_path):
  """vit.load(init_params=<tree of the flat store>, ...) with a hi-res posemb change."""
  m_lo = vit.Model(10, **dict(IMG, pool_type="gap"))
  m_hi = vit.Model(10, **dict(IMG, pool_type="gap"))
  s_lo = P.ParamStore(m_lo.entries("", (2, 2)), "cpu"); s_lo.init_random(1)
  s_hi = P.ParamStore(m_hi.entries("", (4, 4)), "cpu"); s_hi.init_random(2)
  f = os.path.join(tmp_path, "ckpt.npz")
  u.save_params_npz(f, dict(s_lo.tree()))
  init = s_hi.tree()
  assert isinstance(init, P.ParamTree) and init.store is s_hi
  loaded = vit.load(init, f, {"pool_type": "gap"})
  assert np.asarray(loaded["pos_embedding"]).shape == (1, 16, 128)
  s_hi.load_tree(loaded)
  np.testing.assert_allclose(s_hi.leaf("head/kernel").numpy(), s_lo.leaf("head/kernel").numpy())


def test_load_without_init_tree_uses_model_cfg_scan(tmp_path):
  m = vit.Model(10, **dict(IMG, pool_type="gap"))
  s = P.ParamStore(m.entries("", (2, 2)), "cpu"); s.init_random(1)
  f = os.path.join(tmp_path, "ckpt.npz")
  u.save_params_npz(f, dict(s.tree()))
  assert "encoderblock" in vit.load(None, f, {"scan": True})["Transformer"]
  assert "encoderblock_0" in vit.load(None, f, {"scan": False})["Transformer"]
  assert "encoderblock_0" in vit.load(None, f, None)["Transformer"]
  mt = text_transformer.Model(16, **TXT)
  st = P.ParamStore(mt.entries("", 8), "cpu"); st.init_random(1)
  ft = os.path.join(tmp_path, "txt.npz")
  u.save_params_npz(ft, dict(st.tree()))
  assert "encoderblock" in text_transformer.load(None, ft, {"scan": True})["Encoder_0"]


def test_adhoc_apply_reloads_the_tree_every_call(dry, monkeypatch):
  """Two different plain trees (and an in-place edit of one) must each reach the device store;
  only one store per geometry is kept."""
  monkeypatch.setattr(torch, "device", lambda *a, **k: torch.empty(0).device)   # "cuda" -> cpu for the dry run
  m = vit.Model(10, **dict(IMG, pool_type="gap"))
  ref = P.ParamStore(m.entries("", (2, 2)), "cpu"); ref.init_random(3)
  tree_a = u.tree_map(lambda v: v.numpy().copy(), dict(ref.tree()))
  tree_b = u.tree_map(lambda v: v + 1.0, tree_a)
  image = torch.zeros((2, 32, 32, 3))
  m.apply({"params": tree_a}, image, collect=False)
  stores = [v for k, v in m._execs.items() if k[0] == "adhoc"]
  assert len(stores) == 1
  st = stores[0]
  np.testing.assert_array_equal(st.leaf("head/bias").numpy(), tree_a["head"]["bias"])
  m.apply({"params": tree_b}, image, collect=False)
  np.testing.assert_array_equal(st.leaf("head/bias").numpy(), tree_b["head"]["bias"])
  tree_b["head"]["bias"][:] = 7.0
  m.apply({"params": tree_b}, image, collect=False)
  assert float(st.leaf("head/bias")[0]) == 7.0
  assert len([k for k in m._execs if k[0] == "adhoc"]) == 1


def test_pool_none_is_forward_only_passthrough(dry):
  m = vit.Model(10, **dict(IMG, pool_type="none"))
  st = P.ParamStore(m.entries("", (2, 2)), "cpu"); st.init_random(0)
  x, out = m.apply({"params": st.tree()}, torch.zeros((2, 32, 32, 3)), collect=True)
  assert tuple(x.shape) == (2, 4, 10) and tuple(out["logits"].shape) == (2, 4, 10)
  assert "head_input" not in out and tuple(out["encoded"].shape) == (2, 4, 128)
  with pytest.raises(NotImplementedError):
    m.executor(st, "", (2, 2)).fwd(torch.zeros((2, 32, 32, 3)), save=True)
  with pytest.raises(ValueError):
    vit.Model(10, **dict(IMG, pool_type="bogus"))


def test_optimizer_state_roundtrip_with_reference_names(dry, tmp_path):
  """`opt/1/0/{0,1,2}` = masked(scale_by_adam) -> ScaleByAdamState(count, mu, nu) at chain position 1
  (optax.py:143-149), `opt/<j>/0/0` = the scale_by_schedule counts; frozen leaves carry no state."""
  model = two_towers.Model(image=IMG, text=dict(TXT, scan=True), out_dim=(None, 64), temperature_init=10.0,
                           bias_init=-2.71)
  cfg = _cfg(schedule=[("img/.*", None), (".*", dict(decay_type="cosine", warmup_steps=2))])
  ishape, tshape = (4, 32, 32, 3), (4, 8)
  state, _ = siglip.make_train_state(model, cfg, ishape, tshape, rng=0, total_steps=10, device="cpu")
  opt = state["opt"]
  g = torch.Generator().manual_seed(0)
  opt.mu.copy_(torch.randn(opt.mu.shape, generator=g)); opt.nu.copy_(torch.rand(opt.nu.shape, generator=g))
  opt.count = 7
  f = os.path.join(tmp_path, "state.npz")
  u.save_train_state(f, state)
  flat = u.npload(f)
  assert int(flat["opt/1/0/0"]) == 7
  # chain = [clip, masked(adam), scale(lr), wd (1 mask), sched (1 group), set_to_zero, scale(-1)]
  assert int(flat["opt/4/0/0"]) == 7
  assert "opt/1/0/1/txt/head/kernel" in flat and "opt/1/0/2/t" in flat
  assert flat["opt/1/0/1/txt/Encoder_0/encoderblock/MlpBlock_0/Dense_0/kernel"].shape == (1, 128, 256)
  assert not any(k.startswith("opt/") and "/img/" in k for k in flat), "frozen leaves have no optimizer state"
  assert "params/img/cls" in flat and "params/txt/Embed_0/embedding" in flat
  # resume into a fresh state
  state2, _ = siglip.make_train_state(model, cfg, ishape, tshape, rng=5, total_steps=10, device="cpu")
  u.load_train_state(f, state2)
  s1, s2 = state["params"].store, state2["params"].store
  assert torch.equal(s1.master, s2.master) and s2._shadow_dirty is False
  assert state2["opt"].count == 7
  for e in s1.entries.values():
    if e.name in s1.frozen:
      continue
    sl = slice(e.offset, e.offset + e.numel)
    assert torch.equal(opt.mu[sl], state2["opt"].mu[sl]) and torch.equal(opt.nu[sl], state2["opt"].nu[sl]), e.name
  # params-only consumers (model load) read the same file
  assert "img" in u.load_params(f) and "cls" in u.load_params(f + ":img")
