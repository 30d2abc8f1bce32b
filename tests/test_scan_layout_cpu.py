"""scan=True models (models/vit.py:129-148: nn.scan over the encoder blocks) PRESENT their
parameters stacked - `Transformer/encoderblock/<leaf>` with a leading depth axis - exactly like
the reference (vit.py:363-405 converts between the two layouts).  Here the stacked leaves are
strided views over the per-block storage the kernels use, so only the naming layer is under
test: names, shapes, aliasing, checkpoint loading in both directions, and regex addressing by
the optimizer config (schedule / freezing)."""
import numpy as np
import pytest
import torch

from big_vision_amd import optax as bv_optax
from big_vision_amd import utils as u
from big_vision_amd.compat.ml_collections import ConfigDict
from big_vision_amd.models import vit
from big_vision_amd.models.proj.image_text import two_towers
from big_vision_amd.params import ParamStore

IMG = dict(width=128, depth=3, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="map")
TXT = dict(width=128, depth=2, mlp_dim=256, num_heads=2, vocab_size=50)


def _store(scan_img, scan_txt, frozen_leaves=()):
  m = two_towers.Model(image=dict(IMG, scan=scan_img), text=dict(TXT, scan=scan_txt), out_dim=(None, 32),
                       temperature_init=10.0, bias_init=-10.0)
  st = m.make_store((2, 32, 32, 3), (2, 8), device="cpu", frozen_leaves=frozen_leaves)
  st.init_random(0)
  return m, st


def _np_tree(store):
  return u.tree_map(lambda v: v.detach().clone().numpy(), dict(store.tree()))


def test_scan_tree_is_the_stacked_pyloop_tree():
  _, loop = _store(False, False)
  m, scan = _store(True, False)
  t_loop, t_scan = _np_tree(loop), _np_tree(scan)
  want = dict(t_loop)
  want["img"] = vit.pyloop_to_scan(t_loop["img"])
  fa, fb = dict(u.tree_flatten_with_names(want)[0]), dict(u.tree_flatten_with_names(t_scan)[0])
  assert fa.keys() == fb.keys()
  for k in fa:
    assert fa[k].shape == fb[k].shape, k
    np.testing.assert_array_equal(fa[k], fb[k], err_msg=k)      # same seed -> same per-block values
  assert "encoderblock" in t_scan["img"]["Transformer"] and "encoderblock_0" in t_scan["txt"]["Encoder_0"]
  q = scan.leaf("img/Transformer/encoderblock/MultiHeadDotProductAttention_0/query/kernel")
  assert tuple(q.shape) == (3, 128, 2, 64)
  assert m.leaf_names((2, 32, 32, 3), (2, 8)) == scan.leaf_names()


def test_stacked_leaf_aliases_block_storage():
  _, st = _store(True, True)
  stacked = st.leaf("txt/Encoder_0/encoderblock/MlpBlock_0/Dense_0/kernel")
  stacked[1].fill_(7.0)
  assert torch.all(st.leaf("txt/Encoder_0/encoderblock_1/MlpBlock_0/Dense_0/kernel") == 7.0)
  assert not torch.any(st.leaf("txt/Encoder_0/encoderblock_0/MlpBlock_0/Dense_0/kernel") == 7.0)
  assert torch.all(st.t("txt/Encoder_0/encoderblock_1/MlpBlock_0/Dense_0/kernel") == 7.0)   # what the kernels read
  g = st.leaf("txt/Encoder_0/encoderblock/LayerNorm_0/scale", "grad")
  assert tuple(g.shape) == (2, 128)


def test_load_tree_accepts_both_layouts():
  _, src = _store(False, False)
  _, dst = _store(True, True)
  dst.master.zero_()
  dst.load_tree(dict(src.tree()))                       # per-block checkpoint into a scanned model
  np.testing.assert_array_equal(dst.master.numpy(), src.master.numpy())
  _, dst2 = _store(False, False)
  dst2.master.zero_()
  dst2.load_tree(dict(dst.tree()), strict=False)        # stacked names are unknown to a pyloop model ...
  with pytest.raises(ValueError):
    dst2.load_tree(dict(dst.tree()))                    # ... and strict loading says so


def test_vit_load_converts_to_the_presented_layout(tmp_path):
  m_scan = vit.Model(num_classes=None, **dict(IMG, scan=True))
  m_loop = vit.Model(num_classes=None, **IMG)
  s_scan = ParamStore(m_scan.entries("", (2, 2)), "cpu", scan_prefixes=m_scan.scan_prefixes()); s_scan.init_random(1)
  s_loop = ParamStore(m_loop.entries("", (2, 2)), "cpu"); s_loop.init_random(2)
  f = str(tmp_path / "loop.npz")
  u.save_params_npz(f, dict(s_loop.tree()))
  got = vit.load(_np_tree(s_scan), f, IMG)              # pyloop checkpoint -> scan model
  assert "encoderblock" in got["Transformer"] and "encoderblock_0" not in got["Transformer"]
  np.testing.assert_array_equal(got["Transformer"]["encoderblock"]["LayerNorm_0"]["scale"][2],
                                _np_tree(s_loop)["Transformer"]["encoderblock_2"]["LayerNorm_0"]["scale"])
  f2 = str(tmp_path / "scan.npz")
  u.save_params_npz(f2, dict(s_scan.tree()))
  back = vit.load(_np_tree(s_loop), f2, IMG)            # scan checkpoint -> pyloop model
  assert "encoderblock_1" in back["Transformer"] and "encoderblock" not in back["Transformer"]


def test_optimizer_regexes_address_stacked_names():
  cfg = ConfigDict()
  cfg.lr, cfg.wd, cfg.optax_name = 1e-3, 1e-2, "scale_by_adam"
  cfg.schedule = [("img/Transformer/encoderblock/.*", None), (".*", dict(decay_type="cosine", warmup_steps=2))]
  m = two_towers.Model(image=dict(IMG, scan=True), text=TXT, out_dim=(None, 32), temperature_init=10.0, bias_init=-10.0)
  leaves = m.leaf_names((2, 32, 32, 3), (2, 8))
  frozen = bv_optax.frozen_leaves(cfg, leaves)
  assert frozen and all(n.startswith("img/Transformer/encoderblock/") for n in frozen)
  st = m.make_store((2, 32, 32, 3), (2, 8), device="cpu", frozen_leaves=frozen)
  assert all(("encoderblock_" in e) == (e in st.frozen) for e in st.entries if e.startswith("img/Transformer/"))
  opt, _ = bv_optax.make(cfg, st, sched_kw=dict(total_steps=10, batch_size=2))
  assert opt.mu.numel() == st.trainable_count
  assert "img/Transformer/encoderblock/LayerNorm_0/scale" not in dict(u.tree_flatten_with_names(st.tree("grad"))[0])
  # per-block names do NOT match a scanned model (they would in the pyloop layout)
  cfg2 = ConfigDict(cfg.to_dict())
  cfg2.schedule = [("img/Transformer/encoderblock_0/.*", None), (".*", dict(decay_type="cosine", warmup_steps=2))]
  assert not bv_optax.frozen_leaves(cfg2, leaves)
