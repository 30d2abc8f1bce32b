"""Host-side logic that needs no GPU: Flax-compatible parameter naming / layout,
regex masks, durations and schedules (the reference's known answers,
big_vision/utils_test.py:228-281), config-driven freezing, ConfigDict."""
import os
import sys

import numpy as np
import pytest
import torch

import bv_oracle as O
from big_vision_amd import optax as bv_optax
from big_vision_amd import utils as u
from big_vision_amd.compat.ml_collections import ConfigDict
from big_vision_amd.models import vit
from big_vision_amd.models.proj.image_text import two_towers
from big_vision_amd.params import ParamStore, make_masks

IMG = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="map")
TXT = dict(width=128, depth=2, mlp_dim=256, num_heads=2, vocab_size=100)


def _model(**kw):
  return two_towers.Model(image=IMG, text=TXT, out_dim=(None, 128), temperature_init=10.0,
                          bias_init=-10.0, **kw)


def test_param_tree_matches_flax_names_and_shapes():
  """Leaf names/shapes == the reference's Flax tree, as restated by the oracle init."""
  model = _model()
  store = model.make_store((4, 32, 32, 3), (4, 16), device="cpu")
  ours = {k: tuple(v.shape) for k, v in u.tree_flatten_with_names(store.tree())[0]}
  ref = O.init_two_towers(0, (32, 32), 16, image_cfg=IMG, text_cfg=TXT, out_dim=(None, 128),
                          temperature_init=10.0, bias_init=-10.0)
  want = {k: tuple(v.shape) for k, v in O.tree_flatten_with_names(ref)}
  assert ours == want
  assert "img/Transformer/encoderblock_0/MultiHeadDotProductAttention_0/query/kernel" in ours
  assert ours["img/Transformer/encoderblock_0/MultiHeadDotProductAttention_0/query/kernel"] == (128, 2, 64)
  assert ours["img/MAPHead_0/probe"] == (1, 1, 128) and ours["t"] == (1,) and ours["b"] == (1,)


def test_b16_variant_table():
  """models/vit.py:284-303."""
  assert vit.decode_variant("B/16") == dict(width=768, depth=12, mlp_dim=3072, num_heads=12, patch_size=(16, 16))
  assert vit.decode_variant("S/16")["mlp_dim"] == 1536 and vit.decode_variant("L/16")["depth"] == 24
  assert vit.decode_variant("mu/16")["width"] == 32 and vit.decode_variant(None) == {}


def test_store_roundtrip_and_fused_views():
  """load_tree(tree) -> tree() is the identity; q/k/v leaves are views of the fused tensor."""
  model = _model()
  store = model.make_store((4, 32, 32, 3), (4, 16), device="cpu")
  ref = O.init_two_towers(3, (32, 32), 16, image_cfg=IMG, text_cfg=TXT, out_dim=(None, 128),
                          temperature_init=10.0, bias_init=-10.0)
  store.load_tree(ref)
  back = dict(u.tree_flatten_with_names(store.tree())[0])
  for k, v in O.tree_flatten_with_names(ref):
    assert torch.equal(back[k], v), k
  A = "img/Transformer/encoderblock_1/MultiHeadDotProductAttention_0"
  fused = store.t(f"{A}/qkv/kernel")
  assert fused.shape == (128, 3, 2, 64)
  assert torch.equal(fused[:, 1], back[f"{A}/key/kernel"])
  # every storage tensor starts on a 1024-element boundary (fused optimizer addressing)
  assert all(e.offset % 1024 == 0 for e in store.entries.values())
  with pytest.raises(ValueError):
    store.load_tree({"img": {}}, strict=True)


def test_frozen_towers_are_laid_out_last():
  """config.schedule [("img/.*", None), ...] (siglip_lit_coco.py:101-104): no grads / Adam state."""
  model = _model()
  cfg = ConfigDict(dict(schedule=[("img/.*", None), (".*", dict(decay_type="cosine", warmup_steps=1))]))
  names = model.leaf_names((4, 32, 32, 3), (4, 16))
  frozen = bv_optax.frozen_leaves(cfg, names)
  assert frozen and all(n.startswith("img/") for n in frozen)
  store = model.make_store((4, 32, 32, 3), (4, 16), device="cpu", frozen_leaves=frozen)
  offs = {e.name: e.offset for e in store.entries.values()}
  assert max(o for n, o in offs.items() if not n.startswith("img/")) < \
      min(o for n, o in offs.items() if n.startswith("img/"))
  assert store.trainable_count < store.count
  assert store.g("img/embedding/kernel") is None and store.g("t") is not None


def test_make_masks_first_match_wins():
  """utils.py:1195-1212."""
  names = ["a/kernel", "a/bias", "b/kernel"]
  m1, m2 = make_masks(names, ["a/.*", ".*/kernel"])
  assert m1 == {"a/kernel": True, "a/bias": True, "b/kernel": False}
  assert m2 == {"a/kernel": False, "a/bias": False, "b/kernel": True}
  with pytest.raises(AssertionError):
    make_masks(names, ["/a"])


def test_tree_names_sorted_traversal():
  """utils_test.py:144-225 style: '/'-joined names, sorted keys."""
  tree = {"b": {"y": 1, "x": 2}, "a": 3}
  names = [n for n, _ in u.tree_flatten_with_names(tree)[0]]
  assert names == ["a", "b/x", "b/y"]


@pytest.mark.parametrize("data_size,batch_size,total,cfg,expected", [
    (1000, None, None, dict(foo_steps=3), 3), (1000, 100, None, dict(foo_epochs=3), 30),
    (None, 100, None, dict(foo_examples=300), 3), (None, None, 10, dict(foo_percent=0.30), 3),
    (None, None, 10, dict(foo_percent=0.0), 0), (1001, 100, None, dict(foo_epochs=3), 30),
    (None, 101, None, dict(foo_examples=300), 3), (None, None, 11, dict(foo_percent=0.30), 3)])
def test_steps_known_answers(data_size, batch_size, total, cfg, expected):
  """big_vision/utils_test.py:228-255 on the PRODUCT's utils.steps."""
  assert u.steps("foo", cfg, data_size=data_size, batch_size=batch_size, total_steps=total) == expected
  with pytest.raises(ValueError):
    u.steps("bar", cfg, data_size=data_size, batch_size=batch_size, total_steps=total)


@pytest.mark.parametrize("decay_type,extra,step,expected", [
    ("linear", {}, 13, .5), ("polynomial", {"end": .1, "power": 2}, 13, .325), ("cosine", {}, 13, .5),
    ("rsqrt", {"timescale": 1}, 13, 0.3333333), ("stair", {"steps": [10], "mults": [.5]}, 5, 1.),
    ("stair", {"steps": [10], "mults": [.5]}, 10, .5), ("rsqrt", {"timescale": 1}, 3, .6),
    ("rsqrt", {"timescale": 1}, 20, .05)])
def test_lr_schedule_known_answers(decay_type, extra, step, expected):
  """big_vision/utils_test.py:258-281 on the PRODUCT's schedule factory."""
  fn = u.create_learning_rate_schedule(total_steps=21, batch_size=512, base=.5, decay_type=decay_type,
                                       scale_with_batchsize=True, warmup_steps=5, cooldown_steps=5, **extra)
  assert abs(float(fn(step)) - expected) < 5e-7


def test_error_conventions():
  with pytest.raises(ValueError):
    vit.Model(num_classes=None, variant="B/16", pool_type="nope")
  with pytest.raises(ValueError):
    vit.Model(num_classes=None, variant="B/16", posemb="nope")
  with pytest.raises(NotImplementedError):
    vit.Model(num_classes=None, width=100, num_heads=2)   # head_dim 50: not a multiple of 8
  vit.Model(num_classes=None, width=96, num_heads=3)     # head_dim 32: the general attention kernels
  assert vit.Model(num_classes=None, variant="So400m/14").width // 16 == 72


def test_posemb_sincos_2d_matches_oracle():
  a = vit.posemb_sincos_2d(3, 5, 64)
  b = O.posemb_sincos_2d(3, 5, 64).numpy()
  assert a.shape == (1, 15, 64) and np.allclose(a, b, atol=1e-6)


def test_configdict_surface():
  c = ConfigDict()
  c.lr = 1e-3
  c.model = dict(image=dict(variant="B/16"), out_dim=(None, 768))
  assert c.model.image.variant == "B/16" and c["model"]["out_dim"] == (None, 768)
  assert c.get("missing", 7) == 7 and "lr" in c
  assert c.to_dict()["model"]["image"] == {"variant": "B/16"}


def test_grad_ranges_for_overlapped_all_reduce():
  """dp.GradSync hands contiguous runs of the flat gradient buffer to RCCL while the backward
  still runs: the text tower and the upper half of the image tower must be contiguous runs
  that do not overlap, and frozen entries must not be part of any run."""
  from big_vision_amd.models.proj.image_text import two_towers
  m = two_towers.Model(image=dict(width=128, depth=4, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="map"),
                       text=dict(width=128, depth=2, mlp_dim=256, num_heads=2, vocab_size=50),
                       out_dim=(None, 32), temperature_init=10.0, bias_init=-10.0)
  store = m.make_store((2, 32, 32, 3), (2, 8), device="cpu")
  txt = store.grad_range(lambda n: n.startswith("txt/"))
  up = store.grad_range(lambda n: n.startswith("img/MAPHead") or "encoder_norm" in n and n.startswith("img/")
                        or any(f"img/Transformer/encoderblock_{i}/" in n for i in (2, 3)))
  assert txt is not None and up is not None
  assert up[1] <= txt[0] and txt[1] <= store.trainable_count
  for name, e in store.entries.items():
    inside = txt[0] <= e.offset < txt[1]
    assert inside == name.startswith("txt/"), name
  # a non-contiguous selection is refused
  assert store.grad_range(lambda n: n in ("img/embedding/kernel", "t")) is None
  assert store.grad_range(lambda n: False) is None


def test_parse_arg_contract():
  """Docstring examples of big_vision/configs/common.py:29-60."""
  from big_vision_amd.configs import common as bvcc
  spec = dict(res=(224, int), runlocal=False, schedule="short")
  a = bvcc.parse_arg("runlocal,schedule=long,res=128", **spec)
  assert (a.res, a.runlocal, a.schedule) == (128, True, "long")
  assert bvcc.parse_arg("res=128", **spec).res == 128
  assert bvcc.parse_arg("runlocal", **spec).runlocal is True
  assert bvcc.parse_arg("runlocal=False", **spec).runlocal is False
  assert bvcc.parse_arg("128", **spec).res == 128          # first spec entry may be passed unnamed
  assert bvcc.parse_arg(None, **spec).res == 224
  with pytest.raises(ValueError):
    bvcc.parse_arg("nope=1", **spec)
  lazy = bvcc.parse_arg("nope=1,x=2.5,y=abc,z=true", lazy=True, **spec)
  assert (lazy.nope, lazy.x, lazy.y, lazy.z) == (1, 2.5, "abc", True)
  assert bvcc.arg(res=256, foo="bar") == {"config_arg": "res=256,foo=bar", "res": 256, "foo": "bar"}


@pytest.mark.skipif(not os.path.isdir("/root/reference/big_vision/configs"), reason="needs the reference checkout")
def test_reference_config_files_load_unchanged():
  """`existing configs load unchanged` (BASELINE north star): the two in-repo configs of the hot
  path are executed as they are and drive our model registry."""
  from big_vision_amd.configs.loader import load_config
  from big_vision_amd import train
  c = load_config("/root/reference/big_vision/configs/vit_s16_i1k.py")
  assert c.model_name == "vit" and c.loss == "softmax_xent" and c.mixup.p == 0.2
  _, model = train.get_model(c)
  assert (model.width, model.depth, model.mlp_dim, model.num_heads, model.num_classes) == (384, 12, 1536, 6, 1000)
  assert model.rep_size is True and model.pool_type == "gap" and model.posemb == "sincos2d"
  names = {e.name for e in model.entries("", (14, 14))}
  assert "pre_logits/kernel" in names and "head/kernel" in names and "pos_embedding" not in names
  lit = load_config("/root/reference/big_vision/configs/proj/image_text/siglip_lit_coco.py",
                    reference_root="/root/reference")
  assert lit.model_name == "proj.image_text.two_towers" and lit.model.bias_init == -2.71
  assert lit.model.image.pool_type == "tok" and tuple(lit.model.out_dim) == (None, 768)
  assert lit.schedule[0] == ("img/.*", None)               # frozen image tower (LiT)
  # the loader cleans up its temporary namespace: afterwards `big_vision` is either absent or the
  # repo's own alias package again (never the bare namespace rooted at the reference checkout)
  bv = sys.modules.get("big_vision")
  assert bv is None or "/root/reference" not in str(getattr(bv, "__path__", ""))
  assert "big_vision.configs.proj.image_text.common" not in sys.modules


def test_oracle_sigmoid_xent_and_mixup():
  import bv_oracle as O
  logits = torch.tensor([[100.0, -100.0, 0.3]], dtype=torch.float64)
  labels = torch.tensor([[1.0, 0.0, 0.25]], dtype=torch.float64)
  naive = -(labels * torch.log(torch.sigmoid(logits)) + (1 - labels) * torch.log(torch.sigmoid(-logits))).sum(-1).mean()
  assert abs(O.sigmoid_xent(logits, labels).item() - naive.item()) < 1e-12
  assert torch.isfinite(O.sigmoid_xent(torch.tensor([[800.0, -800.0]]), torch.tensor([[0.0, 1.0]])))
  x = torch.arange(6.0).view(3, 2)
  (m,) = O.mixup(0.75, x)
  assert torch.equal(m, 0.75 * x + 0.25 * x[[2, 0, 1]])     # roll(x, shift=1, axis=0)
  from big_vision_amd import utils as u
  for s in range(20):
    a = u.get_mixup_coefficient(0, s, 0.2)
    assert 0.5 <= a <= 1.0
  assert u.get_mixup_coefficient(0, 3, 0.2) == u.get_mixup_coefficient(0, 3, 0.2)


def test_initialisers_have_the_flax_statistics():
  """Initialisers of the reference modules (SURVEY.md §8c "Flax facts"): xavier_uniform for the
  attention / MLP kernels (vit.py:66-69,93-98), normal(1e-6) MLP biases, lecun_normal (truncated,
  variance 1/fan_in) for the stem conv and heads, normal(1/sqrt(D)) position embeddings and token
  table, ones / zeros for LayerNorm, log(temperature_init) and bias_init for t / b."""
  import math
  m = two_towers.Model(image=dict(width=256, depth=1, mlp_dim=1024, num_heads=4, patch_size=(16, 16), pool_type="map"),
                       text=dict(width=256, depth=1, mlp_dim=1024, num_heads=4, vocab_size=2000),
                       out_dim=(None, 256), temperature_init=10.0, bias_init=-10.0)
  st = m.make_store((2, 64, 64, 3), (2, 16), device="cpu")
  st.init_random(0)
  leaf = lambda n: st.leaf(n).double()

  def close(x, want, rel):
    assert abs(x - want) <= rel * abs(want), (x, want)

  D, M = 256, 1024
  blk = "img/Transformer/encoderblock_0/"
  q = leaf(blk + "MultiHeadDotProductAttention_0/query/kernel")
  lim = math.sqrt(6.0 / (D + D))
  assert q.abs().max() <= lim and q.abs().max() >= 0.98 * lim
  close(q.var().item(), lim * lim / 3.0, 0.05)                      # U(-lim, lim)
  w1 = leaf(blk + "MlpBlock_0/Dense_0/kernel")
  close(w1.var().item(), (6.0 / (D + M)) / 3.0, 0.05)
  b1 = leaf(blk + "MlpBlock_0/Dense_0/bias")
  close(b1.std().item(), 1e-6, 0.15)
  assert torch.all(leaf(blk + "LayerNorm_0/scale") == 1) and torch.all(leaf(blk + "LayerNorm_0/bias") == 0)
  assert torch.all(leaf(blk + "MultiHeadDotProductAttention_0/out/bias") == 0)
  stem = leaf("img/embedding/kernel")
  close(stem.var().item(), 1.0 / (16 * 16 * 3), 0.05)               # lecun_normal: variance 1 / fan_in
  assert stem.abs().max() <= 2.0 / 0.87962566103423978 * math.sqrt(1.0 / (16 * 16 * 3)) + 1e-9   # truncated at 2 sigma
  close(leaf("img/pos_embedding").std().item(), 1 / math.sqrt(D), 0.05)
  close(leaf("txt/Embed_0/embedding").std().item(), 1 / math.sqrt(D), 0.02)
  close(leaf("txt/head/kernel").var().item(), 1.0 / D, 0.05)
  close(leaf("t").item(), math.log(10.0), 1e-6)
  close(leaf("b").item(), -10.0, 1e-6)
  probe = leaf("img/MAPHead_0/probe")             # (1, 1, D): fan_in 1, fan_out D (vit.py:172-173)
  lim_p = math.sqrt(6.0 / (1 + D))
  assert 0.9 * lim_p <= probe.abs().max() <= lim_p
