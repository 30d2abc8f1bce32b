"""BigVision Adafactor on the GPU (bv_adafactor_leaf per parameter leaf, fused with clip / lr / wd /
schedule / apply) against the oracle's restatement of optax.scale_by_factored_rms + optax.ema fed
the SAME gradients: isolates the optimizer.  Two consecutive steps (second-moment decay, bf16
momentum carry-over), with and without a frozen tower.  fp32 kernels vs the fp32 oracle: max-abs
parameter error <= 2e-5 of the largest parameter."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("frozen,clip", [(False, None), (True, None), (False, 0.7)], ids=["all", "frozen_img", "all-clip"])
def test_two_adafactor_steps_match_the_oracle(dev, frozen, clip):
  import bv_oracle as O
  from big_vision_amd import utils as u
  from big_vision_amd.compat.ml_collections import ConfigDict
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  image_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="map")
  text_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, vocab_size=100)
  model = two_towers.Model(image=image_cfg, text=text_cfg, out_dim=(None, 128), temperature_init=10.0, bias_init=-10.0)
  c = ConfigDict()
  c.lr, c.wd, c.total_steps, c.grad_clip_norm = 1e-2, 1e-2, 10, 1.0
  c.optax_name = "big_vision.scale_by_adafactor"
  if clip:   # scale_by_adafactor(clipping_threshold=...): clip_by_block_rms per leaf (u of a leaf has rms ~1: 0.7 clips most)
    c.optax = dict(clipping_threshold=clip)
  sched = dict(decay_type="cosine", warmup_steps=2)
  c.schedule = [("img/.*", None), (".*", sched)] if frozen else sched
  c.lr_mults = [("txt/.*", 0.5), (".*", 1.0)]
  image, text = O.synthetic_batch(1, 8, 64, 16, 100)
  state, _ = siglip.make_train_state(model, c, tuple(image.shape), tuple(text.shape), rng=0, total_steps=10)
  store = state["params"].store
  assert state["opt"].mu.dtype == torch.bfloat16
  snap = lambda t: {k: v.detach().cpu().double().clone() for k, v in u.tree_flatten_with_names(t)[0]}
  orc = O.OptaxOracle(c.to_dict(), O.recover_tree(list(snap(state["params"]).items())),
                      sched_kw=dict(total_steps=10, batch_size=8))
  update_fn = siglip.make_update_fn(model, c)
  batch = {"image": image.to(dev), "labels": text.to(dev)}
  for step in range(2):
    p_before = snap(state["params"])
    state, meas = update_fn(state, None, batch)
    torch.cuda.synchronize()
    gours = snap(store.tree("grad"))
    g_all = {k: gours.get(k, torch.zeros_like(v)) for k, v in p_before.items()}
    # the oracle runs in fp32 like the kernel (the bf16 momentum makes fp64 vs fp32 a different function)
    upd = orc.update(O.recover_tree([(k, v.float()) for k, v in g_all.items()]),
                     O.recover_tree([(k, v.float()) for k, v in p_before.items()]))
    upd = dict(O.tree_flatten_with_names(upd))
    p_after = snap(state["params"])
    worst = 0.0
    for k, v in p_after.items():
      ref = p_before[k] + upd[k].double()
      err = (v - ref).abs().max().item()
      worst = max(worst, err / max(1.0, ref.abs().max().item()))
      if frozen and k.startswith("img/"):
        assert torch.equal(v, p_before[k]), f"frozen leaf {k} changed"
    assert worst <= 2e-5, f"step {step}: parameter error {worst:.3e}"
    l2u = math.sqrt(sum((upd[k].double() ** 2).sum().item() for k in p_after))
    assert abs(meas["l2_updates"].item() - l2u) <= 2e-3 * l2u
    assert math.isfinite(meas["training_loss"].item())


def test_batched_step_equals_the_per_leaf_entry(dev):
  """bv_adafactor_step (all leaves of a model in four launches, device leaf table) against bv_adafactor_leaf called
  leaf by leaf on cloned buffers: parameters, momentum, statistics and the bf16 shadow must be bit-identical (same
  arithmetic per element; only the fp64 statistics atomics are summed in another order)."""
  from big_vision_amd import ops
  from big_vision_amd.compat.ml_collections import ConfigDict
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  image_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="map")
  text_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, vocab_size=100)
  model = two_towers.Model(image=image_cfg, text=text_cfg, out_dim=(None, 128), temperature_init=10.0, bias_init=-10.0)
  c = ConfigDict()
  c.lr, c.wd, c.total_steps, c.grad_clip_norm = 1e-2, 1e-2, 10, 1.0
  c.optax_name = "big_vision.scale_by_adafactor"
  c.schedule = dict(decay_type="cosine", warmup_steps=0)
  c.lr_mults = [("txt/.*", 0.5), (".*", 1.0)]
  state, _ = siglip.make_train_state(model, c, (8, 64, 64, 3), (8, 16), rng=0, total_steps=10)
  opt, st = state["opt"], state["params"].store
  g = torch.Generator(device=dev).manual_seed(5)
  st.ensure_grad().copy_(torch.randn(st.grad.shape, generator=g, device=dev) * 1e-2)
  # a second, independent copy of every buffer for the per-leaf path
  m2, mu2, sh2, af2 = st.master.clone(), opt.mu.clone(), st.shadow.clone(), opt.af_state.clone()
  for step in range(2):
    k = opt.count
    sched = [fn(k) for fn in opt.schedule_fns]
    decay = min(opt.af["beta2_cap"], 1.0 - (float(k - opt.af["decay_offset"]) + 1.0) ** (-opt.af["decay_rate"]))
    gsq = torch.zeros(1, device=dev, dtype=torch.float64)
    ops.sqnorm_(st.grad, gsq)
    stats2 = torch.zeros(2, device=dev, dtype=torch.float64)
    for lf in opt.af_leaves:
      ops.adafactor_leaf_(m2, st.grad, mu2, sh2, lf["view"], af2[lf["soff"]:lf["soff"] + lf["n_state"]], lf["factored"], gsq,
                          opt.clip_norm, decay, opt.af["eps"], opt.af["momentum"], lf["lr_eff"], lf["wd"], sched[lf["sched"]],
                          stats2)
    meas = opt.step()
    torch.cuda.synchronize()
    n = st.trainable_count
    assert torch.equal(st.master[:n], m2[:n]), f"step {step}: parameters differ"
    assert torch.equal(opt.mu, mu2) and torch.equal(opt.af_state, af2) and torch.equal(st.shadow[:n], sh2[:n])
    assert abs(meas["l2_updates"].item() - math.sqrt(stats2[1].item())) <= 1e-9 * math.sqrt(stats2[1].item())
