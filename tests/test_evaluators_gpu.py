"""The retrieval evaluator driven by the accelerated predict_fn: the
embeddings that reach the recall computation are the HIP model's, batched and padded to one
shape; similarities must match the fp64 oracle's and the metric names the reference's."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_retrieval_evaluator_on_tiny_towers(dev):
  import bv_oracle as O
  from big_vision_amd import utils as u
  from big_vision_amd.compat.ml_collections import ConfigDict
  from big_vision_amd.evaluators.proj.image_text import retrieval
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  image_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="map")
  text_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, vocab_size=100)
  model = two_towers.Model(image=image_cfg, text=text_cfg, out_dim=(None, 128), temperature_init=10.0,
                           bias_init=-10.0)
  config = ConfigDict()
  config.lr, config.wd, config.optax_name, config.total_steps = 1e-3, 0.0, "scale_by_adam", 10
  config.schedule = dict(decay_type="cosine", warmup_steps=2)
  n_img, caps = 10, 3
  image, _ = O.synthetic_batch(1, n_img, 64, 16, 100)
  _, text = O.synthetic_batch(2, n_img * caps, 64, 16, 100)
  state, _ = siglip.make_train_state(model, config, (4, 64, 64, 3), (4, 16), rng=0, total_steps=10)
  ev = retrieval.Evaluator(siglip.make_predict_fn(model), images=image, texts=text, batch_size=4)
  res = ev.evaluate(state)
  assert res["images"]["embeddings"].shape == (n_img, 128) and res["texts"]["embeddings"].shape == (n_img * caps, 128)
  params64 = O.recover_tree([(k, v.detach().cpu().double()) for k, v in u.tree_flatten_with_names(state["params"])[0]])
  zi, zt, _ = O.two_towers_forward(params64, image.double(), text, image_cfg=image_cfg, text_cfg=text_cfg,
                                   out_dim=(None, 128))
  sims = (zi @ zt.T).numpy()
  assert np.abs(res["similarities"] - sims).max() <= 3e-2
  names = [k for k, _ in ev.run(state)]
  assert names == [f"{d}_recall@{k}" for d in ("img2txt", "txt2img") for k in (1, 5, 10)]
  for _, v in ev.run(state):
    assert 0.0 <= v <= 1.0
  # image-only / text-only calls of predict_fn (two_towers.py:43)
  zimg, ztxt, _ = siglip.make_predict_fn(model)(state, {"image": image[:4].to(dev)})
  assert ztxt is None and zimg.shape == (4, 128)


def test_zeroshot_classifier_on_tiny_towers(dev):
  """discriminative_classifier.Evaluator on the accelerated predict_fn vs the fp64 oracle forward +
  the reference's decision rule (:305-318) in numpy: same correct count, metric name of `run`,
  multi-label targets, padded last batch."""
  import bv_oracle as O
  from big_vision_amd import utils as u
  from big_vision_amd.compat.ml_collections import ConfigDict
  from big_vision_amd.evaluators.proj.image_text import discriminative_classifier as dc
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  image_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="map")
  text_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, vocab_size=100)
  model = two_towers.Model(image=image_cfg, text=text_cfg, out_dim=(None, 128), temperature_init=10.0, bias_init=-10.0)
  config = ConfigDict()
  config.lr, config.wd, config.optax_name, config.total_steps = 1e-3, 0.0, "scale_by_adam", 10
  config.schedule = dict(decay_type="cosine", warmup_steps=2)
  n_img, n_cls, n_tpl = 22, 5, 3
  image, _ = O.synthetic_batch(1, n_img, 64, 16, 100)
  _, prompts = O.synthetic_batch(2, n_cls * n_tpl, 64, 16, 100)
  prompt_labels = np.repeat(np.arange(n_cls), n_tpl)
  rng = np.random.RandomState(3)
  labels = np.stack([rng.randint(0, n_cls, n_img), rng.randint(0, n_cls, n_img)], 1)     # multi-label [N, 2]
  state, _ = siglip.make_train_state(model, config, (8, 64, 64, 3), (8, 16), rng=0, total_steps=10)
  ev = dc.Evaluator(siglip.make_predict_fn(model), batch_size=8,
                    datasets={"toy": dict(images=image, labels=labels, prompts=prompts, prompt_labels=prompt_labels)})
  res = ev.evaluate(state, "toy", return_embeddings=True)
  params64 = O.recover_tree([(k, v.detach().cpu().double()) for k, v in u.tree_flatten_with_names(state["params"])[0]])
  zi, zt, _ = O.two_towers_forward(params64, image.double(), prompts, image_cfg=image_cfg, text_cfg=text_cfg,
                                   out_dim=(None, 128))
  avg = dc._average_embeddings(zt.numpy(), labels=prompt_labels, num_classes=n_cls, normalize=True)
  sims = zi.numpy() @ avg.T
  best = sims.argmax(1)
  # examples whose top-2 margin is below the bf16 embedding noise may legitimately flip
  srt = np.sort(sims, 1)
  sure = (srt[:, -1] - srt[:, -2]) > 2e-2
  want = ((best[:, None] == labels).sum(1) > 0)
  got_best = (res["images"]["embedding"] @ res["texts"]["average_embedding"].T).argmax(1)
  assert (got_best[sure] == best[sure]).all()
  assert res["count"] == n_img and abs(res["correct"] - int(want.sum())) <= int((~sure).sum())
  assert abs(res["accuracy"] - res["correct"] / n_img) < 1e-12
  assert np.abs(res["texts"]["average_embedding"] - avg).max() <= 3e-2
  assert ev.run(state)[0][0] == "toy_accuracy"
