"""The reference's import surface (`big_vision.*`) served by this repo: module paths a
big_vision trainer uses verbatim - importlib.import_module(f"big_vision.models.{config.model_name}")
(trainers/proj/image_text/siglip.py:190-191), big_vision.optax / utils / sharding, the historical
names of BASELINE.json (trainers.proj.image_text.contrastive, configs.proj.image_text.lit_coco) -
must resolve to the accelerated implementation; names outside the hot path must fail loudly.
CPU only: kernels are replaced by a recorder (dry run), so this checks plumbing, not numbers."""
import collections
import numpy as np
import importlib

import pytest
import torch


@pytest.fixture()
def dry(monkeypatch):
  from big_vision_amd import _lib, ops
  calls = collections.Counter()
  monkeypatch.setattr(_lib, "call", lambda name, *a: calls.update([name]))
  monkeypatch.setattr(ops, "_chk", lambda t, dtype, name: t)
  monkeypatch.setattr(ops, "_stream", lambda: 0)
  monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a: (1 << 40, 1 << 40))
  monkeypatch.setattr(torch.cuda, "memory_reserved", lambda *a: 0)
  monkeypatch.setattr(torch.cuda, "memory_allocated", lambda *a: 0)
  return calls


def test_module_paths_are_the_same_modules():
  import big_vision_amd.models.vit as vit_amd
  assert importlib.import_module("big_vision.models.vit") is vit_amd
  from big_vision.models.proj.image_text import two_towers, text_transformer   # noqa: F401
  import big_vision.optax as bv_optax
  import big_vision.utils as u
  import big_vision.sharding as sh
  import big_vision.train as train
  from big_vision.trainers.proj.image_text import siglip, contrastive, _deprecated_contrastive
  from big_vision.evaluators.proj.image_text import retrieval            # noqa: F401
  assert contrastive.loss_fn is _deprecated_contrastive.loss_fn
  assert callable(bv_optax.make) and callable(u.steps) and callable(sh.infer_sharding) and callable(train.make_update_fn)
  assert callable(siglip.make_update_fn)
  with pytest.raises(ModuleNotFoundError, match="outside the accelerated hot path"):
    importlib.import_module("big_vision.input_pipeline")
  with pytest.raises(ModuleNotFoundError):
    importlib.import_module("big_vision.models.bit")


def test_lit_coco_config_under_both_names():
  a = importlib.import_module("big_vision.configs.proj.image_text.lit_coco").get_config("batch_size=64")
  b = importlib.import_module("big_vision.configs.proj.image_text.siglip_lit_coco").get_config("batch_size=64")
  assert a.to_dict() == b.to_dict()
  assert a.input.batch_size == 64 and a.model_name == "proj.image_text.two_towers"
  assert a.schedule[0] == ("img/.*", None) and a.schedule[1][1]["warmup_steps"] == 150
  assert a.model.bias_init == -2.71 and a.model.image.pool_type == "tok" and tuple(a.model.out_dim) == (None, 768)
  assert a.lr == 1e-3 and a.wd == 1e-2 and a.grad_clip_norm == 1.0


@pytest.mark.parametrize("txt", ["bert_base", "transformer_b"])
def test_reference_style_trainer_snippet(dry, txt):
  """What trainers/proj/image_text/siglip.py:180-323 does, with big_vision.* names only - with the config's
  default BERT text tower (`text_model='proj.flaxformer.bert'`, siglip_lit_coco.py:78,84-87; cut to a
  test-sized encoder) and with the in-repo text transformer."""
  import big_vision.optax as bv_optax
  import big_vision.sharding as bv_sharding
  import big_vision.utils as u
  from big_vision.trainers.proj.image_text import siglip as trainer
  config = importlib.import_module("big_vision.configs.proj.image_text.lit_coco").get_config(
      f"batch_size=4,res=32,token_len=8,txt={txt}")
  config.model.image = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="tok",
                            head_zeroinit=False)
  if txt == "bert_base":
    assert config.model.text_model == "proj.flaxformer.bert" and config.model.text.config == "base"
    assert config.optax_name == "scale_by_adam"
    config.model.text = dict(config=dict(hidden_size=128, intermediate_dim=256, num_hidden_layers=2, num_attention_heads=2,
                                         vocab_size=50, max_length=16, num_segments=2), head_zeroinit=False)
  else:
    config.model.text = dict(width=128, depth=2, mlp_dim=256, num_heads=2, vocab_size=50)
  config.model.out_dim = (None, 128)
  model_mod = importlib.import_module(f"big_vision.models.{config.model_name}")      # siglip.py:190
  model = model_mod.Model(**config.model)
  image = torch.zeros((4,) + tuple(config.init_shapes[0][1:]))
  text = torch.ones((4,) + tuple(config.init_shapes[1][1:]), dtype=torch.int32)
  state, sched_fns = trainer.make_train_state(model, config, tuple(image.shape), tuple(text.shape), rng=0, device="cpu")
  specs = bv_sharding.infer_sharding(state["params"], config.sharding_strategy, None)     # siglip.py:232-237
  def leaves(t):
    return [t] if isinstance(t, tuple) else [x for v in t.values() for x in leaves(v)]
  flat = leaves(specs)
  assert len(flat) == len(u.tree_flatten_with_names(state["params"])[0]) and all(all(a is None for a in spec) for spec in flat)
  names = [n for n, _ in u.tree_flatten_with_names(state["params"])[0]]
  assert set(bv_optax.frozen_leaves(config, names)) == {n for n in names if n.startswith("img/")}
  update_fn = trainer.make_update_fn(model, config)
  dry.clear()
  state, meas = update_fn(state, None, {"image": image, "labels": text})
  assert {"training_loss", "l2_grads", "l2_params", "l2_updates"} <= set(meas)
  if txt == "bert_base":   # BERT masks the padded keys (input_mask = text != 0): the *_masked entry points; image tower frozen
    assert (dry["bv_attn_fwd"], dry["bv_attn_fwd_masked"], dry["bv_attn_bwd"], dry["bv_attn_bwd_masked"]) == (2, 2, 0, 2)
  else:
    assert dry["bv_attn_fwd"] == 4 and dry["bv_attn_bwd"] == 2       # image tower frozen: text-only backward
  assert len(sched_fns) >= 1


def test_contrastive_trainer_loss_switch(dry):
  from big_vision.trainers.proj.image_text import contrastive
  from big_vision.models.proj.image_text import two_towers
  from ml_collections import ConfigDict
  tower = dict(width=128, depth=1, mlp_dim=256, num_heads=2)
  model = two_towers.Model(image=dict(tower, patch_size=(16, 16), pool_type="map"), text=dict(tower, vocab_size=50),
                           out_dim=(None, 64), temperature_init=10.0, bias_init=-10.0)
  image, text = torch.zeros((4, 32, 32, 3)), torch.ones((4, 8), dtype=torch.int32)
  for loss, kernel in (("sigmoid", "bv_siglip_loss"), ("chunked_sigmoid", "bv_siglip_loss"), ("softmax", "bv_softmax_xent")):
    c = ConfigDict(dict(lr=1e-3, wd=1e-2, optax_name="scale_by_adam", total_steps=10, grad_clip_norm=1.0,
                        schedule=dict(decay_type="cosine", warmup_steps=2), loss_fn=loss))
    state, _ = contrastive.make_train_state(model, c, tuple(image.shape), tuple(text.shape), rng=0, device="cpu")
    dry.clear()
    state, meas = contrastive.make_update_fn(model, c)(state, None, {"image": image, "labels": text})
    assert dry[kernel] >= 1, (loss, dict(dry))
    assert {"t", "t/parameter", "train/nimg", "train/ntxt", "training_loss"} <= set(meas), (loss, set(meas))
    if loss != "softmax":
      assert dry["bv_logit_stats"] == 1 and "train/pos_avg_logit" in meas
      assert ("train/neg_avg_logit" in meas) == (loss == "sigmoid")
  bad = ConfigDict(dict(c.to_dict(), loss_fn="nce"))
  with pytest.raises(NotImplementedError, match="Unrecognized loss"):
    contrastive.make_update_fn(model, bad)


def test_sharding_replicate_fsdp_and_the_refused_rules():
  import numpy as np
  import big_vision.sharding as sh
  params = {"img": {"embedding": {"kernel": np.zeros((16, 16, 3, 8)), "bias": np.zeros(8)}}, "t": np.zeros(1)}
  specs = sh.infer_sharding(params, [(".*", "replicate")], None)
  assert specs == {"img": {"embedding": {"kernel": (None,) * 4, "bias": (None,)}}, "t": (None,)}
  assert sh.infer_sharding(params) == specs                                       # default strategy
  assert sh.infer_sharding(params, [("img/.*", "replicate")]) == specs            # unmatched leaves stay replicated
  assert not sh.is_sharded(specs)
  for bad in ("shard_dim('data', 0)", "logical_partitioning"):
    with pytest.raises(NotImplementedError, match="replicated"):
      sh.infer_sharding(params, [(".*", bad)])
  with pytest.raises(KeyError):
    sh.infer_sharding(params, [(".*", "bogus")])
  # fsdp (sharding.py:104-139): the largest axis the device count divides, tensors above min_size_to_shard_mb only
  class Mesh:   # the rank group stands in for the device mesh: only its size matters
    size = 8
  big = {"mlp": {"kernel": np.zeros((768, 3072), np.float32), "bias": np.zeros(3072, np.float32)},
         "odd": np.zeros((1023, 1025), np.float32), "qkv": np.zeros((768, 12, 64), np.float32)}
  sp = sh.infer_sharding(big, [(".*", "fsdp(axis='data')")], Mesh())
  assert sp == {"mlp": {"kernel": (None, "data"), "bias": (None,)},      # 9 MiB: cut along 3072; the bias is 12 KiB
                "odd": (None, None),                                      # nothing divisible by 8: stays replicated
                "qkv": (None, None, None)}                                # 2.25 MiB <= 4 MiB: stays replicated
  assert sh.is_sharded(sp)
  sp1 = sh.infer_sharding(big, [(".*", "fsdp(axis='data', min_size_to_shard_mb=1)")], Mesh())
  assert sp1["qkv"] == ("data", None, None) and sp1["odd"] == (None, None)
  with pytest.raises(ValueError, match="can't be fully replicated"):          # replicate after fsdp on the same leaf
    sh.infer_sharding(big, [(".*", "fsdp(axis='data')|replicate")], Mesh())
  from ml_collections import ConfigDict
  assert sh.is_sharded(sh.check_config(ConfigDict(dict(sharding_strategy=[(".*", "fsdp(axis='data')")])), big, Mesh()))


def test_fsdp_placement_builds_the_sharded_optimizer(dry):
  """config.sharding_strategy = [(".*", "fsdp(axis='data')")] -> optimizer moments only for the own 1/N slice of
  the flat buffer, the update through reduce_scatter / all_gather (dp.Comm; identities in a one-rank job), no
  gradient all-reduce by the trainer.  Host logic on the dry-run kernels; arithmetic: tests/test_dp_two_ranks_gpu.py."""
  from big_vision.trainers.proj.image_text import siglip as trainer
  from big_vision.models.proj.image_text import two_towers
  from ml_collections import ConfigDict
  tower = dict(width=128, depth=1, mlp_dim=256, num_heads=2)
  model = two_towers.Model(image=dict(tower, patch_size=(16, 16), pool_type="map"), text=dict(tower, vocab_size=50),
                           out_dim=(None, 64), temperature_init=10.0, bias_init=-10.0)
  image, text = torch.zeros((4, 32, 32, 3)), torch.ones((4, 8), dtype=torch.int32)
  c = ConfigDict(dict(lr=1e-3, wd=1e-2, optax_name="scale_by_adam", total_steps=10, grad_clip_norm=1.0,
                      schedule=dict(decay_type="cosine", warmup_steps=2),
                      sharding_strategy=[(".*", "fsdp(axis='data', min_size_to_shard_mb=0)")]))
  state, _ = trainer.make_train_state(model, c, tuple(image.shape), tuple(text.shape), rng=0, device="cpu")
  opt = state["opt"]
  assert opt.sharded and (opt.lo, opt.hi) == (0, state["params"].store.trainable_count) and opt.mu.numel() == opt.S
  dry.clear()
  state, meas = trainer.make_update_fn(model, c)(state, None, {"image": image, "labels": text})
  assert dry["bv_adam_step"] == 1 and opt.count == 1 and {"l2_grads", "l2_params", "l2_updates"} <= set(meas)
  # checkpoints hold WHOLE moments (gathered over the ranks on save - a collective: every rank calls it - and
  # sliced on load), in the same optax naming as the replicated optimizer
  import big_vision.utils as u
  opt.mu.copy_(torch.arange(opt.mu.numel(), dtype=torch.float32).to(opt.mu.dtype) % 7)
  flat = dict(u.tree_flatten_with_names(opt.state_tree())[0])
  n_tr = state["params"].store.trainable_count
  assert sum(v.numel() for k, v in flat.items() if k.startswith("1/0/1/")) <= n_tr and int(flat["1/0/0"]) == 1
  before = opt.mu.clone()
  opt.mu.zero_()
  opt.load_state_tree(flat)
  keep = dict(u.tree_flatten_with_names(opt._moment_tree(opt._full_moment(opt.mu)))[0])
  want = dict(u.tree_flatten_with_names(opt._moment_tree(opt._full_moment(before)))[0])
  assert all(torch.equal(keep[k], want[k]) for k in want)
  c2 = ConfigDict(dict(c.to_dict(), optax_name="big_vision.scale_by_adafactor"))
  # Adafactor under fsdp (round 4): ownership by whole tensors - on one rank, every leaf is this rank's
  state2, _ = trainer.make_train_state(model, c2, tuple(image.shape), tuple(text.shape), rng=0, device="cpu")
  opt2 = state2["opt"]
  assert opt2.sharded and opt2.bounds == [0, state2["params"].store.trainable_count]
  assert opt2.af_nown == len(opt2.af_leaves) and all(lf["own"] for lf in opt2.af_leaves)
  dry.clear()
  state2, meas2 = trainer.make_update_fn(model, c2)(state2, None, {"image": image, "labels": text})
  # one bv_adafactor_step call per SIZE CLASS of the leaf table (a single table made every bias pay for
  # the largest leaf's grid); the classes partition the table and no leaf sits in a class sized > 4x its own extent
  assert dry["bv_adafactor_step"] == len(opt2.af_classes) >= 2 and opt2.count == 1
  assert [c["first"] for c in opt2.af_classes] == [sum(d["n"] for d in opt2.af_classes[:i]) for i in range(len(opt2.af_classes))]
  assert sum(c["n"] for c in opt2.af_classes) == opt2.af_nown
  by_off = {int(lf["view"][0]): lf for lf in opt2.af_leaves}
  rows = opt2.af_table.cpu().numpy().view(np.uint8).reshape(opt2.af_nown, 88)
  for c in opt2.af_classes:
    for r in rows[c["first"]:c["first"] + c["n"]]:
      lf = by_off[int(r[:8].view(np.int64)[0])]
      tot = lf["B"] * lf["R"] * lf["C"]
      assert lf["factored"] == c["factored"] and tot <= c["total"] < 4 * tot + 4, (lf["leaf"], tot, c)
      if lf["factored"]:
        assert lf["B"] * lf["R"] <= c["rows"] <= 4 * lf["B"] * lf["R"], (lf["leaf"], c)
