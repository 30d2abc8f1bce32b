"""The BERT text tower (`text_model='proj.flaxformer.bert'`, configs/proj/image_text/siglip_lit_coco.py:78,84-87)
on the GPU against the fp64 oracle restatement (bv_oracle.bert_forward, pinned to HuggingFace BertModel -
PARITY UNPINNED against flaxformer itself, which is not vendored): post-LN blocks on the kernels of the
pre-LN ViT blocks, key-padding lengths from input_mask = (text != 0), CLS pooling, `head`.

  tower forward (model.apply)   CLS / logits max-abs <= 2e-2 of unit-scale activations
  LiT step (frozen ViT image tower + trainable BERT, siglip_lit_coco.py:79-104): the bounds of
  tests/test_siglip_step_gpu.py (per-tensor gradient cosine >= 0.999, rel-L2 <= 3e-2; optimizer rtol 1e-5)
"""
import pytest
import torch

from test_siglip_step_gpu import LIT_SCHEDULE, _cfg, _run_case

pytestmark = pytest.mark.gpu

TINY = dict(hidden_size=128, intermediate_dim=256, num_hidden_layers=2, num_attention_heads=2, vocab_size=60,
            max_length=24, num_segments=2)


def test_bert_tower_forward_matches_oracle(dev):
  import bv_oracle as O
  from big_vision_amd import utils as u
  from big_vision_amd.models.proj.flaxformer import bert
  gen = torch.Generator().manual_seed(3)
  p = O.init_bert(gen, config=TINY, num_classes=64, head_zeroinit=False, dtype=torch.float64)
  p = O.tree_map(lambda t: t + 0.05 * torch.randn(t.shape, generator=gen, dtype=torch.float64), p)
  text = torch.tensor([[7, 8, 9, 3, 0, 0, 0, 0, 0, 0, 0, 0], [5, 5, 6, 2, 4, 9, 11, 12, 13, 14, 15, 16],
                       [2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0], [9, 8, 7, 6, 5, 4, 3, 2, 0, 0, 0, 0]])
  ref, rout = O.bert_forward(p, text, config=TINY, num_classes=64)
  model = bert.Model(TINY, num_classes=64, head_zeroinit=False)
  params = u.tree_map(lambda t: t.float().to(dev), p)
  got, out = model.apply({"params": params}, text.to(dev))
  assert (got.cpu().double() - ref).abs().max() <= 2e-2, (got.cpu().double() - ref).abs().max()
  assert (out["pre_logits"].cpu().double() - rout["pre_logits"]).abs().max() <= 2e-2
  valid = (text != 0)[..., None]
  err = ((out["transformed"].cpu().double() - rout["transformed"]).abs() * valid).max()
  assert err <= 3e-2, err                                   # padded positions differ by construction (module docstring)
  with pytest.raises(NotImplementedError, match="prefix"):  # a hole in the mask is refused, not mis-masked
    bert.Model(TINY, num_classes=64).apply({"params": params}, torch.tensor([[5, 0, 6, 0]]).to(dev))


def test_bad_input_mask_of_the_last_batch_is_refused_before_the_update(dev):
  """Advisor r4: the mask verdict of batch k used to be read when batch k + 1 arrived - the LAST batch of a run (or a
  bad second batch) was trained on.  Now the trainer reads it before `opt.step()`: the bad batch raises in its own
  step and leaves every weight bit untouched; `apply` (the evaluators' predict_fn) raises before it returns."""
  import bv_oracle as O
  from big_vision_amd import dp
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  image_cfg = dict(width=128, depth=1, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="tok", head_zeroinit=False)
  model = two_towers.Model(image=image_cfg, text=dict(config=dict(TINY, num_hidden_layers=1), head_zeroinit=False),
                           text_model="proj.flaxformer.bert", out_dim=(None, 128), temperature_init=10.0, bias_init=-2.71)
  config = _cfg()
  image, text = O.synthetic_batch(5, 4, 32, 12, 60)
  text = text.clamp(min=1)                      # no pad id at all: a valid (full-length) mask
  image, text = image.to(dev), text.to(dev)
  state, _ = siglip.make_train_state(model, config, tuple(image.shape), tuple(text.shape), rng=0, comm=dp.Comm(), total_steps=10)
  fn = siglip.make_update_fn(model, config, comm=dp.Comm())
  state, _ = fn(state, None, {"image": image, "labels": text})            # first batch: fine
  torch.cuda.synchronize()
  before = state["params"].store.master.detach().clone()
  bad = text.clone(); bad[1, 3] = 0                                         # an inner pad id in the SECOND (= last) batch
  with pytest.raises(NotImplementedError, match="prefix"):
    fn(state, None, {"image": image, "labels": bad})
  torch.cuda.synchronize()
  assert torch.equal(state["params"].store.master, before), "a refused batch must not move the weights"
  allpad = text.clone(); allpad[2] = 0
  with pytest.raises(ValueError, match="without any token"):
    fn(state, None, {"image": image, "labels": allpad})
  with pytest.raises(NotImplementedError, match="prefix"):                  # eval path: raised by the call that saw the batch
    siglip.make_predict_fn(model)(state, {"labels": bad})
  state, _ = fn(state, None, {"image": image, "labels": text})            # and a good batch afterwards trains again
  torch.cuda.synchronize()
  assert not torch.equal(state["params"].store.master, before)


def test_lit_step_with_bert_text_tower_tiny(dev):
  """LiT on a toy width: frozen `tok`-pooled ViT + trainable BERT (2 post-LN blocks), ragged padding."""
  image_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="tok", head_zeroinit=False)
  _run_case(dev, image_cfg, dict(config=TINY, head_zeroinit=False), E=128, n=8, res=64, seq=16, vocab=60, bias_init=-2.71,
            config=_cfg(schedule=LIT_SCHEDULE), frozen=("img/",), floor=True, text_model="proj.flaxformer.bert", pad_id=0,
            case="LiT tiny: frozen ViT + BERT text tower")


def test_lit_step_with_bert_base_shapes(dev):
  """BASELINE configs[4] as the reference writes it: ViT-B/16 (frozen, pool 'tok') + BERT-base text tower (width 768,
  12 heads, intermediate 3072, vocab 30522, 16 tokens; depth cut to 2 blocks for the CPU oracle), n = 8, nothing else
  changed (siglip_lit_coco.py:33,46,79-104)."""
  bert_base_d2 = dict(hidden_size=768, intermediate_dim=3072, num_hidden_layers=2, num_attention_heads=12,
                      vocab_size=30522, max_length=512, num_segments=2)
  image_cfg = dict(variant="B/16", pool_type="tok", head_zeroinit=False, depth=2)
  _run_case(dev, image_cfg, dict(config=bert_base_d2, head_zeroinit=False), E=768, n=8, res=224, seq=16, vocab=30522,
            bias_init=-2.71, config=_cfg(schedule=LIT_SCHEDULE), frozen=("img/",), floor=True,
            text_model="proj.flaxformer.bert", pad_id=0, case="LiT B/16 (depth 2) + BERT-base shapes (depth 2) n=8")
