"""Checkpoint loading = SURVEY.md §8(f) rank 1: npz I/O (utils.py:133-227), `vit.load` with its
backward-compat fix-ups (models/vit.py:324-433), `two_towers.load` (two_towers.py:93-137) and
`common.merge_params` (models/common.py:24-92).  Pure host code: the trees are numpy."""
import numpy as np
import pytest
import torch

import bv_oracle as O
from big_vision_amd import utils as u
from big_vision_amd.models import vit
from big_vision_amd.models.proj.image_text import two_towers

CFG = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16))


def _np(tree):
  return u.tree_map(lambda v: v.numpy() if torch.is_tensor(v) else np.asarray(v), tree)


def _vit_tree(seed, hw=(64, 64), **kw):
  g = torch.Generator().manual_seed(seed)
  return _np(O.init_vit(g, hw, **{**CFG, **kw}))


def _same(a, b):
  fa, fb = dict(u.tree_flatten_with_names(a)[0]), dict(u.tree_flatten_with_names(b)[0])
  assert fa.keys() == fb.keys(), (sorted(fa.keys() ^ fb.keys()))
  for k in fa:
    np.testing.assert_array_equal(np.asarray(fa[k]), np.asarray(fb[k]), err_msg=k)


def test_npz_roundtrip_and_subkey(tmp_path):
  tree = {"img": _vit_tree(0, pool_type="map"), "t": np.array([2.3], np.float32)}
  f = str(tmp_path / "ckpt.npz")
  u.save_params_npz(f, tree)
  _same(u.load_params(f), tree)
  _same(u.load_params(f + ":img"), tree["img"])
  _same(u.load_params(f + ":img/Transformer/encoder_norm"), tree["img"]["Transformer"]["encoder_norm"])
  # train-state checkpoints: {"params": ...} and legacy {"opt": {"target": ...}} wrappers
  u.save_params_npz(f, {"params": tree, "opt": {"count": np.zeros(1)}})
  _same(u.load_params(f), tree)
  u.save_params_npz(f, {"opt": {"target": tree}})
  _same(u.load_params(f), tree)


def test_npload_bf16_as_void16(tmp_path):
  x = np.array([1.0, -2.5, 3.140625], np.float32)
  raw = (x.view(np.uint32) >> 16).astype(np.uint16).view(np.dtype("V2"))
  f = str(tmp_path / "bf16.npz")
  np.savez(f, w=raw)
  np.testing.assert_array_equal(u.npload(f)["w"], x)   # these values are exact in bf16


def test_vit_load_same_layout_and_dont_load(tmp_path):
  init, ckpt = _vit_tree(1, num_classes=10), _vit_tree(2, num_classes=10)
  f = str(tmp_path / "vit.npz")
  u.save_params_npz(f, ckpt)
  _same(vit.load(init, f, CFG), ckpt)
  got = vit.load(init, f, CFG, dont_load=("head/.*",))
  np.testing.assert_array_equal(got["head"]["bias"], init["head"]["bias"])
  np.testing.assert_array_equal(got["head"]["kernel"], init["head"]["kernel"])
  np.testing.assert_array_equal(got["embedding"]["kernel"], ckpt["embedding"]["kernel"])


def test_vit_load_mismatch_raises_with_diff(tmp_path):
  init, ckpt = _vit_tree(1, num_classes=10), _vit_tree(2)        # checkpoint has no head
  f = str(tmp_path / "vit.npz")
  u.save_params_npz(f, ckpt)
  with pytest.raises(ValueError) as e:
    vit.load(init, f, CFG)
  assert "head/kernel" in str(e.value) and "not in checkpoint" in str(e.value)


def test_vit_load_old_checkpoint_fixups(tmp_path):
  """posembed_input under Transformer, MAP head params at top level, cls-token posemb folded."""
  new = _vit_tree(3, pool_type="map")
  old = u.tree_map(lambda x: x, new)
  old["Transformer"]["posembed_input"] = {"pos_embedding": old.pop("pos_embedding")}
  for k in ("probe", "MlpBlock_0", "MultiHeadDotProductAttention_0", "LayerNorm_0"):
    old[k] = old["MAPHead_0"].pop(k)
  del old["MAPHead_0"]
  f = str(tmp_path / "old.npz")
  u.save_params_npz(f, old)
  _same(vit.load(new, f, CFG), new)
  # (gs*gs + 1)-long posemb of a token-pooled checkpoint: first row belongs to the cls token
  tok = _vit_tree(4, pool_type="tok")
  oldtok = u.tree_map(lambda x: x, tok)
  pe_cls = np.full((1, 1, CFG["width"]), 0.25, np.float32)
  oldtok["pos_embedding"] = np.concatenate([pe_cls, tok["pos_embedding"]], axis=1)
  oldtok["cls"] = tok["cls"] - pe_cls
  u.save_params_npz(f, oldtok)
  got = vit.load(tok, f, CFG)
  np.testing.assert_allclose(got["cls"], tok["cls"], atol=1e-7)
  np.testing.assert_array_equal(got["pos_embedding"], tok["pos_embedding"])


def test_vit_load_scan_layout_and_posemb_resample(tmp_path):
  ckpt = _vit_tree(5, hw=(64, 64))                  # 4x4 grid
  f = str(tmp_path / "scan.npz")
  u.save_params_npz(f, vit.pyloop_to_scan(ckpt))
  assert "encoderblock" in u.load_params(f)["Transformer"]
  _same(vit.load(_vit_tree(6, hw=(64, 64)), f, CFG), ckpt)
  _same(vit.scan_to_pyloop(vit.pyloop_to_scan(ckpt)), ckpt)
  hires = _vit_tree(7, hw=(128, 128))               # 8x8 grid: bilinear resize of the 4x4 table
  got = vit.load(hires, f, CFG)
  assert got["pos_embedding"].shape == (1, 64, CFG["width"])
  ref = vit.resample_posemb(ckpt["pos_embedding"], hires["pos_embedding"])
  np.testing.assert_array_equal(got["pos_embedding"], ref)
  np.testing.assert_allclose(got["pos_embedding"].mean(), ckpt["pos_embedding"].mean(), atol=2e-2)
  same = vit.resample_posemb(ckpt["pos_embedding"], ckpt["pos_embedding"])
  np.testing.assert_array_equal(same, ckpt["pos_embedding"])


def test_two_towers_load_single_file(tmp_path):
  g = torch.Generator().manual_seed(8)
  image_cfg = dict(CFG, pool_type="map")
  text_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, vocab_size=50)
  mk = lambda seed: _np(O.init_two_towers(seed, (64, 64), 8, image_cfg=image_cfg, text_cfg=text_cfg,
                                          out_dim=(None, 64), temperature_init=10.0, bias_init=-10.0))
  init, ckpt = mk(0), mk(1)
  f = str(tmp_path / "tt.npz")
  u.save_params_npz(f, ckpt)
  model_cfg = dict(image=image_cfg, text=text_cfg, out_dim=(None, 64), temperature_init=10.0, bias_init=-10.0)
  got = two_towers.load(init, f, model_cfg)
  _same(got, ckpt)
  only_img = two_towers.load(init, {"img": f + ":img"}, model_cfg)
  _same(only_img["img"], ckpt["img"])
  _same(only_img["txt"], init["txt"])


def test_bert_load_big_vision_checkpoint_and_lit_model_init(tmp_path):
  """models/proj/flaxformer/bert.py:67-94 `load` on a big_vision checkpoint (the second branch; the TensorFlow
  branch needs tensorflow + flaxformer and raises), used as the literal LiT config does through
  two_towers.load: `model_init = {'image': ..., 'text': ...}` with `txt_load_kw = {'dont_load': ['head/kernel',
  'head/bias']}` (configs/proj/image_text/siglip_lit_coco.py:69-74)."""
  from big_vision_amd.models.proj.flaxformer import bert
  cfg = dict(hidden_size=64, intermediate_dim=128, num_hidden_layers=2, num_attention_heads=2, vocab_size=50,
             max_length=16, num_segments=2)
  ck = _np(O.init_bert(torch.Generator().manual_seed(1), config={**cfg, "max_length": 32}, num_classes=24, head_zeroinit=False))
  init = _np(O.init_bert(torch.Generator().manual_seed(2), config=cfg, num_classes=24, head_zeroinit=False))
  f = str(tmp_path / "bert.npz")
  u.save_params_npz(f, ck)
  got = bert.load(init, f, None, dont_load=("head/kernel", "head/bias"))
  pos = got["BertEncoder_0"]["embedder"]["embedders_position_ids"]["embedding"]
  assert pos.shape == (16, 64)                                              # cropped to the model's max_length (:81-88)
  np.testing.assert_array_equal(pos, ck["BertEncoder_0"]["embedder"]["embedders_position_ids"]["embedding"][:16])
  np.testing.assert_array_equal(got["BertEncoder_0"]["encoder_block_1"]["mlp_block"]["mlp"]["wi"]["kernel"],
                                ck["BertEncoder_0"]["encoder_block_1"]["mlp_block"]["mlp"]["wi"]["kernel"])
  np.testing.assert_array_equal(got["head"]["kernel"], init["head"]["kernel"])       # dont_load keeps the init
  # a tree that does not match is refused with the diff, not silently merged
  bad = dict(ck); bad["BertEncoder_0"] = dict(ck["BertEncoder_0"]); bad["BertEncoder_0"].pop("layer_norm")
  fb = str(tmp_path / "bad.npz")
  u.save_params_npz(fb, bad)
  with pytest.raises(ValueError, match="layer_norm"):
    bert.load(init, fb, None)
  # the TensorFlow-checkpoint branch is refused with the reference lines
  tf_dir = tmp_path / "tf_ckpt"; tf_dir.mkdir()
  (tf_dir / "bert_model.ckpt.index").write_bytes(b"")
  with pytest.raises(NotImplementedError, match="bert_checkpoint_converter"):
    bert.load(init, str(tf_dir), None)
  # through two_towers.load, as config.model_init = {'image': ..., 'text': ...} does
  img_ck = _vit_tree(3, pool_type="tok")
  fi = str(tmp_path / "img.npz")
  u.save_params_npz(fi, img_ck)
  init_tt = {"img": _vit_tree(4, pool_type="tok"), "txt": init, "t": np.zeros(1, np.float32), "b": np.zeros(1, np.float32)}
  model_cfg = dict(image_model="vit", text_model="proj.flaxformer.bert", image=dict(CFG, pool_type="tok"),
                   text=dict(config=cfg), bias_init=-2.71)
  out = two_towers.load(init_tt, {"image": fi, "text": f}, model_cfg,
                        txt_load_kw={"dont_load": ["head/kernel", "head/bias"]})
  np.testing.assert_array_equal(out["txt"]["BertEncoder_0"]["layer_norm"]["scale"], ck["BertEncoder_0"]["layer_norm"]["scale"])
  np.testing.assert_array_equal(out["txt"]["head"]["bias"], init["head"]["bias"])
  np.testing.assert_array_equal(out["img"]["embedding"]["kernel"], img_ck["embedding"]["kernel"])
