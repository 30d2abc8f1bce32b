"""SURVEY.md §8(f) rank 1 on the GPU: a checkpoint written in the reference's npz format is loaded
through `two_towers.load` (two_towers.py:93-137 -> vit.load / text_transformer.load with their
fix-ups) INTO THE STORE-BOUND PARAMETER TREE of a freshly initialised model, and the HIP forward on
the loaded weights must equal the oracle's forward on the checkpoint's weights - i.e. the weights
that reach the kernels (master copy, bf16 shadow, transposed shadows) are the loaded ones, including
a posemb resample and the scan-layout conversion; then a train-state round trip (save -> step ->
load -> same step)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

IMG = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="map")
TXT = dict(width=128, depth=2, mlp_dim=256, num_heads=2, vocab_size=100)


def _np_tree(tree):
  import bv_oracle as O
  return O.tree_map(lambda v: v.numpy() if torch.is_tensor(v) else np.asarray(v), tree)


@pytest.mark.parametrize("scan", [False, True])
def test_loaded_checkpoint_reaches_the_kernels(dev, tmp_path, scan):
  import bv_oracle as O
  from big_vision_amd import utils as u
  from big_vision_amd.compat.ml_collections import ConfigDict
  from big_vision_amd.models.proj.image_text import two_towers
  res_ckpt, res, seq = 32, 64, 16        # checkpoint trained at 32 px (2x2 posemb), model runs at 64 px (4x4)
  ckpt = O.init_two_towers(7, (res_ckpt, res_ckpt), seq, image_cfg=IMG, text_cfg=TXT, out_dim=(None, 128),
                           temperature_init=5.0, bias_init=-3.0, dtype=torch.float32)
  g = torch.Generator().manual_seed(11)
  ckpt = O.recover_tree([(n, v + 0.05 * torch.randn(v.shape, generator=g) if n.endswith(("bias", "scale")) else v)
                         for n, v in O.tree_flatten_with_names(ckpt)])
  f = str(tmp_path / "siglip.npz")
  u.save_params_npz(f, {"params": _np_tree(ckpt)})
  image_cfg = dict(IMG, scan=scan)
  text_cfg = dict(TXT, scan=scan)
  model = two_towers.Model(image=image_cfg, text=text_cfg, out_dim=(None, 128), temperature_init=10.0, bias_init=-10.0)
  image, text = O.synthetic_batch(1, 4, res, seq, 100)
  variables = model.init(0, image.to(dev), text.to(dev))                     # store-bound ParamTree on the GPU
  store = variables["params"].store
  model_cfg = ConfigDict(dict(image=image_cfg, text=text_cfg, bias_init=-10.0))
  loaded = two_towers.load(variables["params"], f, model_cfg)                # init_params are CUDA views
  store.load_tree(loaded)
  zimg, ztxt, out = model.apply({"params": store.tree()}, image.to(dev), text.to(dev), collect=False)
  # oracle on the checkpoint's weights with the posemb resampled the reference's way (vit.py:305-321)
  from big_vision_amd.models import vit
  want = O.tree_map(lambda v: v.double(), ckpt)
  want["img"]["pos_embedding"] = torch.from_numpy(np.asarray(vit.resample_posemb(
      ckpt["img"]["pos_embedding"].numpy(), np.zeros((1, (res // 16) ** 2, 128), np.float32)))).double()
  zi, zt, _ = O.two_towers_forward(want, image.double(), text, image_cfg=IMG, text_cfg=TXT, out_dim=(None, 128))
  assert (zimg.cpu().double() - zi).abs().max() <= 2e-2 and (ztxt.cpu().double() - zt).abs().max() <= 2e-2
  assert abs(out["t"].item() - 5.0) < 1e-4 and abs(out["b"].item() + 3.0) < 1e-6
  # the loaded values are what the master buffer holds, under the presented (possibly stacked) names
  names = dict(u.tree_flatten_with_names(store.tree())[0])
  k = "img/Transformer/encoderblock/MlpBlock_0/Dense_0/kernel" if scan else "img/Transformer/encoderblock_1/MlpBlock_0/Dense_0/kernel"
  ref = ckpt["img"]["Transformer"]
  if scan:
    ref = torch.stack([ref[f"encoderblock_{i}"]["MlpBlock_0"]["Dense_0"]["kernel"] for i in range(2)])
  else:
    ref = ref["encoderblock_1"]["MlpBlock_0"]["Dense_0"]["kernel"]
  assert torch.equal(names[k].cpu(), ref)


def test_train_state_roundtrip_resumes(dev, tmp_path):
  import bv_oracle as O
  from big_vision_amd import utils as u
  from big_vision_amd.compat.ml_collections import ConfigDict
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  model = two_towers.Model(image=IMG, text=TXT, out_dim=(None, 128), temperature_init=10.0, bias_init=-10.0)
  c = ConfigDict(dict(lr=1e-3, wd=1e-2, optax_name="scale_by_adam", total_steps=10, grad_clip_norm=1.0,
                      schedule=dict(decay_type="cosine", warmup_steps=2)))
  image, text = O.synthetic_batch(1, 8, 64, 16, 100)
  batch = {"image": image.to(dev), "labels": text.to(dev)}
  state, _ = siglip.make_train_state(model, c, tuple(image.shape), tuple(text.shape), rng=0, total_steps=10)
  fn = siglip.make_update_fn(model, c)
  state, _ = fn(state, None, batch)
  f = str(tmp_path / "state.npz")
  u.save_train_state(f, state)
  saved = {k: v.detach().clone() for k, v in u.tree_flatten_with_names(state["params"])[0]}
  saved_mu, saved_nu, saved_count = state["opt"].mu.clone(), state["opt"].nu.clone(), state["opt"].count
  state, m2 = fn(state, None, batch)
  after = {k: v.detach().clone() for k, v in u.tree_flatten_with_names(state["params"])[0]}
  # fresh state (other seed), resumed from the file: parameters, both Adam moments and the step count
  # come back bit for bit
  state_b, _ = siglip.make_train_state(model, c, tuple(image.shape), tuple(text.shape), rng=5, total_steps=10)
  u.load_train_state(f, state_b)
  for k, v in u.tree_flatten_with_names(state_b["params"])[0]:
    assert torch.equal(v, saved[k]), k
  assert torch.equal(state_b["opt"].mu, saved_mu) and torch.equal(state_b["opt"].nu, saved_nu)
  assert state_b["opt"].count == saved_count == 1
  # ... and the next step is the step the original run took: identical loss (the forward is
  # deterministic); parameters equal up to the fp32-atomic bias-gradient sums, which may flip the
  # sign of an Adam update of a near-zero gradient (one update = lr * schedule = 5e-4 here)
  state_b, m2b = siglip.make_update_fn(model, c)(state_b, None, batch)
  assert m2["training_loss"].item() == m2b["training_loss"].item()
  worst = max((v - after[k]).abs().max().item() for k, v in u.tree_flatten_with_names(state_b["params"])[0])
  assert worst <= 2 * 5e-4 + 1e-6, worst


def test_train_state_of_a_parameter_sharded_run_resumes(dev, tmp_path):
  """"fsdp" placement (sharding.py:104-139): the optimizer holds its slice of the fp32 master and moments only
  (ParamStore.shard_master_).  A checkpoint of such a state holds WHOLE fp32 parameters and moments (gathered on save);
  it resumes a replicated run and a sharded run alike, and the sharded run's next step is the replicated run's."""
  import bv_oracle as O
  from big_vision_amd import utils as u
  from big_vision_amd.compat.ml_collections import ConfigDict
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  base = dict(lr=1e-3, wd=1e-2, optax_name="scale_by_adam", total_steps=10, grad_clip_norm=1.0,
              schedule=dict(decay_type="cosine", warmup_steps=2))
  c_rep = ConfigDict(base)
  c_fsdp = ConfigDict(dict(base, sharding_strategy=[(".*", "fsdp(axis='data', min_size_to_shard_mb=0)")]))
  image, text = O.synthetic_batch(1, 8, 64, 16, 100)
  batch = {"image": image.to(dev), "labels": text.to(dev)}

  def fresh(c, rng):
    model = two_towers.Model(image=IMG, text=TXT, out_dim=(None, 128), temperature_init=10.0, bias_init=-10.0)
    state, _ = siglip.make_train_state(model, c, tuple(image.shape), tuple(text.shape), rng=rng, total_steps=10)
    return state, siglip.make_update_fn(model, c)

  state, fn = fresh(c_fsdp, 0)
  store = state["params"].store
  assert store.master_sharded and store.master is None
  # the live tree is complete: fp32 views for the replicated leaves, the bf16 compute copy for matmul kernels
  live = dict(u.tree_flatten_with_names(state["params"])[0])
  assert live["txt/head/bias"].dtype == torch.float32 and live["txt/head/kernel"].dtype == torch.bfloat16
  state, _ = fn(state, None, batch)
  f = str(tmp_path / "state.npz")
  u.save_train_state(f, state)
  saved = {k: v.detach().clone() for k, v in u.tree_flatten_with_names(store.full_tree())[0]}
  assert all(v.dtype == torch.float32 for v in saved.values())
  sd = state["opt"].state_dict()
  saved_mu, saved_nu = sd["mu"].clone(), sd["nu"].clone()
  state, m2 = fn(state, None, batch)
  after = {k: v.detach().clone() for k, v in u.tree_flatten_with_names(store.full_tree())[0]}
  # (a) into a REPLICATED state of another seed: whole fp32 parameters and moments, bit for bit
  state_r, fn_r = fresh(c_rep, 5)
  u.load_train_state(f, state_r)
  for k, v in u.tree_flatten_with_names(state_r["params"])[0]:
    assert torch.equal(v, saved[k]), k
  assert state_r["opt"].count == 1 and torch.equal(state_r["opt"].mu, saved_mu) and torch.equal(state_r["opt"].nu, saved_nu)
  state_r, m2r = fn_r(state_r, None, batch)
  assert m2["training_loss"].item() == m2r["training_loss"].item()
  worst = max((v - after[k]).abs().max().item() for k, v in u.tree_flatten_with_names(state_r["params"])[0])
  assert worst <= 2 * 5e-4 + 1e-6, worst
  # (b) into a SHARDED state of another seed: the slice, the replicated entries and the bf16 compute copy are the file's
  state_s, fn_s = fresh(c_fsdp, 7)
  u.load_train_state(f, state_s)
  st_s = state_s["params"].store
  assert st_s.master_sharded
  for k, v in u.tree_flatten_with_names(st_s.full_tree())[0]:
    assert torch.equal(v, saved[k]), k
  state_s, m2s = fn_s(state_s, None, batch)
  assert m2["training_loss"].item() == m2s["training_loss"].item()


def test_train_state_of_a_parameter_sharded_adafactor_run_resumes(dev, tmp_path):
  """The same under BigVision Adafactor (optax.py:187-216): the sharded optimizer owns runs of WHOLE tensors, holds the
  fp32 master and the momentum of its run only (base pointers shifted by -lo into the kernel's flat index space) and
  the factored statistics; a checkpoint holds whole parameters / momentum / statistics and resumes either placement."""
  import bv_oracle as O
  from big_vision_amd import utils as u
  from big_vision_amd.compat.ml_collections import ConfigDict
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  base = dict(lr=1e-3, wd=1e-2, optax_name="big_vision.scale_by_adafactor", total_steps=10, grad_clip_norm=1.0,
              schedule=dict(decay_type="cosine", warmup_steps=2))
  c_rep = ConfigDict(base)
  c_fsdp = ConfigDict(dict(base, sharding_strategy=[(".*", "fsdp(axis='data', min_size_to_shard_mb=0)")]))
  image, text = O.synthetic_batch(1, 8, 64, 16, 100)
  batch = {"image": image.to(dev), "labels": text.to(dev)}

  def fresh(c, rng):
    model = two_towers.Model(image=IMG, text=TXT, out_dim=(None, 128), temperature_init=10.0, bias_init=-10.0)
    state, _ = siglip.make_train_state(model, c, tuple(image.shape), tuple(text.shape), rng=rng, total_steps=10)
    return state, siglip.make_update_fn(model, c)

  state, fn = fresh(c_fsdp, 0)
  store, opt = state["params"].store, state["opt"]
  assert store.master_sharded and store.master is None and opt.sharded
  state, _ = fn(state, None, batch)
  f = str(tmp_path / "state.npz")
  u.save_train_state(f, state)
  saved = {k: v.detach().clone() for k, v in u.tree_flatten_with_names(store.full_tree())[0]}
  sd = opt.state_dict()
  assert sd["mu"].numel() == store.trainable_count
  saved_mu = sd["mu"].clone()
  saved_tree = {k: torch.as_tensor(v).detach().clone() for k, v in u.tree_flatten_with_names(opt.state_tree())[0]}
  assert saved_mu.float().abs().sum().item() > 0
  state, m2 = fn(state, None, batch)
  after = {k: v.detach().clone() for k, v in u.tree_flatten_with_names(store.full_tree())[0]}
  # (a) a replicated state of another seed takes the file bit for bit and makes the same next step
  state_r, fn_r = fresh(c_rep, 5)
  u.load_train_state(f, state_r)
  for k, v in u.tree_flatten_with_names(state_r["params"])[0]:
    assert torch.equal(v, saved[k]), k
  assert state_r["opt"].count == 1 and torch.equal(state_r["opt"].mu, saved_mu)
  for k, v in u.tree_flatten_with_names(state_r["opt"].state_tree())[0]:       # v_row / v_col / v / ema of every leaf
    assert torch.equal(torch.as_tensor(v).to(saved_tree[k].device), saved_tree[k]), k
  state_r, m2r = fn_r(state_r, None, batch)
  assert m2["training_loss"].item() == m2r["training_loss"].item()
  # (the gradients of two runs differ in their last bits: one update moves a weight by at most lr (1 - momentum) here)
  worst = max((v - after[k]).abs().max().item() for k, v in u.tree_flatten_with_names(state_r["params"])[0])
  assert worst <= 2 * 2.2e-4 + 1e-6, worst
  # (b) a sharded state of another seed
  state_s, fn_s = fresh(c_fsdp, 7)
  u.load_train_state(f, state_s)
  assert state_s["params"].store.master_sharded
  for k, v in u.tree_flatten_with_names(state_s["params"].store.full_tree())[0]:
    assert torch.equal(v, saved[k]), k
  assert torch.equal(state_s["opt"].state_dict()["mu"], saved_mu)
  state_s, m2s = fn_s(state_s, None, batch)
  assert m2["training_loss"].item() == m2s["training_loss"].item()
  # and state_dict() -> load_state_dict() between the placements
  state_r["opt"].load_state_dict(state_s["opt"].state_dict())
  assert torch.equal(state_r["opt"].mu, state_s["opt"].state_dict()["mu"])
