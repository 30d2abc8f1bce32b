"""Dry run of the trainers' HOST logic on CPU: every libbvhip entry point is replaced by a recorder
(no arithmetic), tensors live on the CPU, and the CUDA memory probes are faked.  What is under
test is control flow that the GPU parity tests only reach in one configuration: how many times
each kernel group is launched per micro-batch scheme (keep / recompute / light contexts), the
memory-driven switch to light contexts, and the order in which gradient ranges are handed to the
all-reduce on N > 1 ranks (dp.GradSync hooks).  Values are garbage by construction."""
import collections

import pytest
import torch

from big_vision_amd import _lib, dp, ops
from big_vision_amd.compat.ml_collections import ConfigDict
from big_vision_amd.models.proj.image_text import two_towers
from big_vision_amd.trainers.proj.image_text import siglip

IMG = dict(width=128, depth=3, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="map")
TXT = dict(width=128, depth=2, mlp_dim=256, num_heads=2, vocab_size=50)


@pytest.fixture()
def dry(monkeypatch):
  calls = collections.Counter()
  monkeypatch.setattr(_lib, "call", lambda name, *a: calls.update([name]))
  monkeypatch.setattr(ops, "_chk", lambda t, dtype, name: t)
  monkeypatch.setattr(ops, "_stream", lambda: 0)
  mem = {"free": 1 << 40, "total": 1 << 40, "alloc": 0}
  monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a: (mem["free"], mem["total"]))
  monkeypatch.setattr(torch.cuda, "memory_reserved", lambda *a: 0)

  def fake_allocated(*a):       # every probe sees 1 GiB more "allocated": a forward "costs" 1 GiB
    mem["alloc"] += 1 << 30
    return mem["alloc"]
  monkeypatch.setattr(torch.cuda, "memory_allocated", fake_allocated)
  return calls, mem


def _cfg(**kw):
  c = ConfigDict()
  c.lr, c.wd, c.optax_name, c.total_steps, c.grad_clip_norm = 1e-3, 1e-2, "scale_by_adam", 10, 1.0
  c.schedule = dict(decay_type="cosine", warmup_steps=2)
  for k, v in kw.items():
    c[k] = v
  return c


def _setup(config, comm=None, n=8):
  model = two_towers.Model(image=IMG, text=TXT, out_dim=(None, 64), temperature_init=10.0, bias_init=-10.0)
  image = torch.zeros((n, 32, 32, 3))
  text = torch.ones((n, 8), dtype=torch.int32)
  state, _ = siglip.make_train_state(model, config, tuple(image.shape), tuple(text.shape), rng=0, comm=comm,
                                     total_steps=10, device="cpu")
  fn = siglip.make_update_fn(model, config, comm=comm)
  return fn, state, {"image": image, "labels": text}


BLOCKS = IMG["depth"] + TXT["depth"]


def test_single_pass_launch_counts(dry):
  calls, _ = dry
  fn, state, batch = _setup(_cfg())
  calls.clear()
  state, meas = fn(state, None, batch)
  assert calls["bv_attn_fwd"] == BLOCKS and calls["bv_attn_bwd"] == BLOCKS
  assert calls["bv_siglip_loss"] == 1 and calls["bv_adam_step"] == 1 and calls["bv_sqnorm"] == 1
  assert calls["bv_embed_fwd"] == 1 and calls["bv_embed_bwd"] == 1 and calls["bv_patchify_ld"] == 1
  assert calls["bv_gemm_bf16_colsum"] == BLOCKS + 1            # fc2 dX with fused Dense_0 bias sums (+ MAP head MLP)
  assert set(meas) == {"training_loss", "l2_grads", "l2_params", "l2_updates"}


@pytest.mark.parametrize("keep,light,fwd_passes", [
    ("all", False, 4), (0, False, 8), (1, False, 7), (2, True, 6), ("all", True, 4)])
def test_microbatch_schemes_recompute_exactly_what_was_not_kept(dry, keep, light, fwd_passes):
  calls, _ = dry
  fn, state, batch = _setup(_cfg(microbatch=2, microbatch_keep=keep, microbatch_light=light))
  calls.clear()
  fn(state, None, batch)
  assert calls["bv_attn_fwd"] == BLOCKS * fwd_passes          # 4 micro-batches + re-run forwards
  assert calls["bv_attn_bwd"] == BLOCKS * 4
  assert calls["bv_siglip_loss"] == 1 and calls["bv_adam_step"] == 1
  assert fn.state_cache["keep_n"] == (4 if keep == "all" else keep)
  # light contexts (fp32 stream): the LayerNorm outputs are re-emitted by the LayerNorm BACKWARD kernel
  # (bv_layernorm_bwd_y, 2 per block and kept micro-batch) - no extra forward normalisation passes - and the
  # fc2 dX GEMM re-emits gelu(h)
  if light:
    kept = fn.state_cache["keep_n"]
    assert calls["bv_layernorm_bwd_y"] >= 2 * BLOCKS * kept
    assert calls["bv_layernorm_fwd"] <= (2 * BLOCKS + 6) * fwd_passes
  else:
    assert calls["bv_layernorm_bwd_y"] == 0


def test_auto_switches_to_light_contexts_when_full_ones_do_not_fit(dry):
  calls, mem = dry
  fn, state, batch = _setup(_cfg(microbatch=2))               # keep "auto", light "auto"
  mem["free"] = int(2.5 * (1 << 30)) + int(0.06 * mem["total"])   # room for ~2 more 1 GiB contexts, not 3
  calls.clear()
  fn(state, None, batch)
  # the dry-run memory model charges every context kind the same 1 GiB, so both trials ("g" = without
  # gelu(h), then "light") are made and the lightest one is kept
  assert fn.state_cache["light"] == "light"
  first = calls["bv_attn_fwd"]
  assert first >= BLOCKS * 6                                   # 4 micro-batches + two re-runs of micro-batch 0
  mem["free"] = 1 << 40                                        # plenty of memory from now on: everything is kept
  calls.clear()
  fn(state, None, batch)
  assert fn.state_cache["light"] == "light" and fn.state_cache["keep_n"] == 4
  assert calls["bv_attn_fwd"] == BLOCKS * 4
  # a model whose full contexts fit never switches
  fn2, state2, batch2 = _setup(_cfg(microbatch=2))
  fn2(state2, None, batch2)
  assert fn2.state_cache["light"] is False and fn2.state_cache["keep_n"] == 4


class _FakeComm(dp.Comm):
  """Two 'ranks' without a process group: collectives are recorded, data is passed through."""

  def __init__(self, log):
    self.enabled, self.group, self.rank, self.size, self.log, self.active = False, None, 1, 2, log, True

  def all_gather_rows(self, x):
    self.log.append("all_gather")
    return torch.cat([x, x])

  def reduce_scatter_rows(self, x):
    self.log.append("reduce_scatter")
    return x[: x.shape[0] // 2].contiguous()

  def all_reduce_sum_(self, flat, bucket_bytes=0):
    self.log.append(("all_reduce", flat.storage_offset(), flat.numel()))

  def all_reduce_scalars_(self, t):
    self.log.append("scalars")

  def barrier(self):
    pass


def test_gradient_ranges_are_reduced_in_backward_order_on_two_ranks(dry):
  calls, _ = dry
  log = []
  comm = _FakeComm(log)
  fn, state, batch = _setup(_cfg(), comm=comm, n=4)
  store = state["params"].store
  log.clear()
  fn(state, None, batch)
  reds = [(a, a + b) for e in log if isinstance(e, tuple) for _, a, b in [e]]
  txt = store.grad_range(lambda n: n.startswith("txt/"))
  assert log.index("all_gather") < log.index("reduce_scatter") < log.index(next(e for e in log if isinstance(e, tuple)))
  blk = lambda tower, enc, i: store.grad_range(lambda n: n.startswith(f"{tower}/{enc}/encoderblock_{i}/"))
  # text tower first (its backward runs first): its blocks last-to-first, one message per block,
  # then whatever is left of the tower (embedding table, final norm, head) ...
  assert reds[0] == blk("txt", "Encoder_0", TXT["depth"] - 1) and reds[TXT["depth"] - 1] == blk("txt", "Encoder_0", 0)
  k = TXT["depth"]
  rest_txt = []
  while reds[k][0] >= txt[0] and reds[k][1] <= txt[1]:
    rest_txt.append(reds[k]); k += 1
  assert rest_txt and sum(b - a for a, b in reds[:k]) == txt[1] - txt[0]         # the whole text tower, exactly once
  # ... then the image tower: final norm + MAP head together with its last block, blocks last-to-first
  tail = store.grad_range(lambda n: n.startswith("img/") and "/encoderblock_" not in n
                          and not n.startswith(("img/embedding", "img/pos_embedding")))
  assert reds[k] == tail and reds[k + 1] == blk("img", "Transformer", IMG["depth"] - 1)
  assert reds[k + IMG["depth"]] == blk("img", "Transformer", 0)
  covered = sorted(reds)
  assert covered[0][0] == 0 and covered[-1][1] == store.trainable_count      # every element exactly once
  assert all(x[1] == y[0] for x, y in zip(covered, covered[1:]))
  # with the overlap switched off it is one plain all-reduce of the whole buffer
  log.clear()
  fn2, state2, batch2 = _setup(_cfg(overlap_grad_sync=False), comm=comm, n=4)
  fn2(state2, None, batch2)
  reds = [e for e in log if isinstance(e, tuple)]
  assert reds == [("all_reduce", 0, state2["params"].store.trainable_count)]


def test_classification_step_launches(dry):
  """big_vision_amd.train: mixup on images AND labels, pre_logits tanh, the configured loss."""
  from big_vision_amd import train
  calls, _ = dry
  cfg = _cfg(model_name="vit", num_classes=10, loss="softmax_xent", mixup=dict(p=0.2, fold_in=None),
             model=dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="gap",
                        rep_size=True, posemb="sincos2d"))
  _, model = train.get_model(cfg)
  state, _ = train.make_train_state(model, cfg, (4, 32, 32, 3), rng=0, total_steps=10, device="cpu")
  fn = train.make_update_fn(model, cfg)
  calls.clear()
  state, meas = fn(state, 0, {"image": torch.zeros((4, 32, 32, 3)), "labels": torch.zeros((4, 10))})
  assert calls["bv_mixup"] == 2 and calls["bv_softmax_xent"] == 1 and calls["bv_sigmoid_xent"] == 0
  assert calls["bv_tanh_fwd"] == 1 and calls["bv_tanh_bwd"] == 1
  assert calls["bv_attn_fwd"] == 2 and calls["bv_attn_bwd"] == 2 and calls["bv_adam_step"] == 1
  assert "pos_embedding" not in state["params"]                 # sincos2d: no learned table
  cfg2 = ConfigDict(cfg.to_dict()); cfg2.loss = "sigmoid_xent"; del cfg2["mixup"]
  fn2 = train.make_update_fn(model, cfg2)
  calls.clear()
  fn2(state, 0, {"image": torch.zeros((4, 32, 32, 3)), "labels": torch.zeros((4, 10))})
  assert calls["bv_mixup"] == 0 and calls["bv_sigmoid_xent"] == 1


def _flat(d, prefix=""):
  out = {}
  for k, v in d.items():
    if isinstance(v, dict):
      out.update(_flat(v, prefix + k + "/"))
    else:
      out[prefix + k] = v
  return out


@pytest.mark.parametrize("img_kw,num_classes", [
    (dict(pool_type="map"), None), (dict(pool_type="tok"), 7), (dict(pool_type="gap", rep_size=True, posemb="sincos2d"), 10),
    (dict(pool_type="0"), None)])
def test_out_dict_has_the_reference_keys_and_shapes(dry, img_kw, num_classes):
  """The `out` dictionary of `apply` is API (two_towers.py:56-57,69-70; vit.py:207-274;
  text_transformer.py:56-98): same keys and shapes as the oracle restatement produces."""
  import bv_oracle as O
  from big_vision_amd.models import vit
  base = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16))
  cfg = {**base, **img_kw}
  image = torch.zeros((3, 48, 32, 3))
  m = vit.Model(num_classes, **cfg)
  from big_vision_amd.params import ParamStore
  st = ParamStore(m.entries("", m.grid(tuple(image.shape))), "cpu")
  st.init_random(0)
  _, out = m.apply({"params": st.tree()}, image)
  gen = torch.Generator().manual_seed(0)
  p = O.init_vit(gen, (48, 32), num_classes=num_classes, **cfg)
  _, ref = O.vit_forward(p, image, num_classes=num_classes, **cfg)
  ours, want = _flat(out), _flat(ref)
  assert set(ours) == set(want), (sorted(set(want) - set(ours)), sorted(set(ours) - set(want)))
  for k in want:
    assert tuple(ours[k].shape) == tuple(want[k].shape), (k, tuple(ours[k].shape), tuple(want[k].shape))


def test_two_towers_out_dict_keys(dry):
  import bv_oracle as O
  model = two_towers.Model(image=IMG, text=TXT, out_dim=(None, 64), temperature_init=10.0, bias_init=-10.0)
  st = model.make_store((2, 32, 32, 3), (2, 8), device="cpu")
  st.init_random(0)
  image, text = torch.zeros((2, 32, 32, 3)), torch.ones((2, 8), dtype=torch.int32)
  _, _, out = model.apply({"params": st.tree()}, image, text)
  p = O.init_two_towers(0, (32, 32), 8, image_cfg=IMG, text_cfg=TXT, out_dim=(None, 64), temperature_init=10.0,
                        bias_init=-10.0)
  _, _, ref = O.two_towers_forward(p, image, text.long(), image_cfg=IMG, text_cfg=TXT, out_dim=(None, 64))
  ours, want = _flat(out), _flat(ref)
  assert set(ours) == set(want), (sorted(set(want) - set(ours)), sorted(set(ours) - set(want)))
  for k in want:
    assert tuple(ours[k].shape) == tuple(want[k].shape), (k, tuple(ours[k].shape), tuple(want[k].shape))
  # one input only (two_towers.py:43)
  zimg, ztxt, out_i = model.apply({"params": st.tree()}, image, None)
  assert ztxt is None and not any(k.startswith("txt/") for k in out_i)


def test_bf16_residual_stream_routes_the_encoder_through_the_bf16x_kernels(dry):
  """config.residual_stream = "bfloat16": block LayerNorms run on the bf16x entry points (both towers),
  the gradient leaving each encoder stack is cast back to fp32 once per tower, and the setting does
  not leak out of the step (model.apply afterwards sees the default fp32 stream)."""
  from big_vision_amd import engine as E
  calls, _ = dry
  fn, state, batch = _setup(_cfg(residual_stream="bfloat16"))
  assert E.residual_stream() == torch.float32
  fn(state, None, batch)
  assert E.residual_stream() == torch.float32
  # 2 LayerNorms per block + encoder_norm per tower on the bf16 stream; the MAP head's LayerNorm (fp32 [n, D]) is not
  assert calls["bv_layernorm_fwd_bf16x"] == 2 * BLOCKS + 2
  assert calls["bv_layernorm_bwd_bf16x"] == 2 * BLOCKS + 2
  assert calls["bv_layernorm_fwd"] == 1 and calls["bv_layernorm_bwd"] == 1
  assert calls["bv_cast_f32"] == 2
  calls.clear()
  fn32, state32, batch32 = _setup(_cfg())
  fn32(state32, None, batch32)
  assert calls["bv_layernorm_fwd_bf16x"] == 0 and calls["bv_cast_f32"] == 0
  assert calls["bv_layernorm_fwd"] == 2 * BLOCKS + 3


def test_context_kinds_full_then_gelu_free_then_light(dry):
  """microbatch_light: "g" keeps the LayerNorm outputs and drops gelu(h) (the fc2 dX GEMM re-emits it);
  "light" drops both - on the fp32 stream the LayerNorm outputs then come out of the LayerNorm BACKWARD kernel
  (bv_layernorm_bwd_y), on the bf16 stream out of a re-normalisation pass."""
  calls, _ = dry
  for kind, renorm in (("g", False), ("light", True), (False, False)):
    fn, state, batch = _setup(_cfg(microbatch=2, microbatch_keep="all", microbatch_light=kind))
    calls.clear()
    fn(state, None, batch)
    assert fn.state_cache["light"] == kind
    base = (2 * BLOCKS + 3) * 4                       # 4 micro-batches of forward LayerNorms
    assert calls["bv_layernorm_fwd"] == base, (kind, calls["bv_layernorm_fwd"])
    assert calls["bv_layernorm_bwd_y"] == (2 * BLOCKS * 4 if renorm else 0), (kind, calls["bv_layernorm_bwd_y"])
  fn, state, batch = _setup(_cfg(microbatch=2, microbatch_keep="all", microbatch_light="light", residual_stream="bfloat16"))
  calls.clear()
  fn(state, None, batch)
  assert calls["bv_layernorm_fwd_bf16x"] == (2 * BLOCKS + 2) * 4 + 2 * BLOCKS * 4 and calls["bv_layernorm_bwd_y"] == 0


def test_weight_images_refresh_in_one_launch_and_frozen_ones_once(dry):
  """The [out][in] bf16 images of the projection kernels (engine._W.bf_t): transposed one by one when first used,
  then all trainable ones in ONE table-driven launch per optimizer step.  LiT (schedule [("img/.*", None), ...]):
  the frozen image tower's images are rebuilt only when the store is re-cast (init / load)."""
  calls, _ = dry
  fn, state, batch = _setup(_cfg(schedule=[("img/.*", None), (".*", dict(decay_type="cosine", warmup_steps=2))]))
  calls.clear()
  state, _ = fn(state, None, batch)
  first = calls["bv_transpose_bf16"]
  assert first > BLOCKS * 4 and calls["bv_transpose_bf16_batched"] == 0      # >= qkv, out, fc1, fc2 per block
  for _ in range(2):
    calls.clear()
    state, _ = fn(state, None, batch)
    assert calls["bv_transpose_bf16"] == 0 and calls["bv_transpose_bf16_batched"] == 1     # the text tower's
  state["params"].store.mark_dirty()          # what load_tree() does: every image is stale again
  calls.clear()
  state, _ = fn(state, None, batch)
  assert calls["bv_transpose_bf16"] == 0 and calls["bv_transpose_bf16_batched"] == 2       # trainable + frozen


def test_batched_transpose_table_layout(dry):
  """ops.transpose_table packs struct bv_tr_leaf (include/bvhip.h): 48 bytes, ascending tile prefix."""
  assert ops.TR_LEAF.itemsize == 48
  a, at = torch.zeros((100, 130), dtype=torch.bfloat16), torch.zeros((130, 104), dtype=torch.bfloat16)
  b, bt = torch.zeros((64, 64), dtype=torch.bfloat16), torch.zeros((64, 64), dtype=torch.bfloat16)
  table, n, tiles = ops.transpose_table([(a, at[:, :100]), (b, bt)], "cpu")
  rec = table.numpy().view(ops.TR_LEAF)
  assert (n, tiles) == (2, 2 * 3 + 1)
  assert [int(x) for x in rec["tile0"]] == [0, 6] and [int(x) for x in rec["tiles_x"]] == [3, 1]
  assert (int(rec["lds"][0]), int(rec["ldd"][0]), int(rec["rows"][0]), int(rec["cols"][0])) == (130, 104, 100, 130)
  assert int(rec["src"][1]) == b.data_ptr() and int(rec["dst"][1]) == bt.data_ptr()
  with pytest.raises(ValueError):
    ops.transpose_table([(a, bt)], "cpu")
