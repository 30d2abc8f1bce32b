"""The N > 1 trainer path on ONE GPU: two (and, for two cases, FOUR) rank processes share cuda:0 and talk over `gloo`
(RCCL refuses two ranks on one device; the collectives' semantics are the same).  Each rank runs
the product `update_fn` on its half of the batch - all_gather of ztxt, sharded sigmoid loss with
the positive diagonal at rank*n, reduce_scatter of dztxt, overlapped gradient all-reduce
(dp.GradSync on a side stream) - and the result must match the single-process step on the
whole batch (trainers/proj/image_text/siglip.py:271-323 under a 2-device mesh).  The 4-rank cases execute the row
offsets rank * n for rank >= 2 and 4-way slice / tensor ownership of the sharded optimizer."""
import math
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


IMAGE_CFG = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="map")
TEXT_CFG = dict(width=128, depth=2, mlp_dim=256, num_heads=2, vocab_size=100)


def _config(**kw):
  from big_vision_amd.compat.ml_collections import ConfigDict
  c = ConfigDict()
  c.lr, c.wd = 1e-3, 1e-2
  c.schedule = dict(decay_type="cosine", warmup_steps=2)
  c.optax_name = "scale_by_adam"
  c.grad_clip_norm = 1.0
  c.total_steps = 10
  for k, v in kw.items():
    if k != "extra_steps":   # test-only knob of _step
      c[k] = v
  return c


def _step(comm, image, text, **kw):
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  from big_vision_amd import utils as u
  if "loss_fn" in kw:     # the contrastive trainer (config.loss_fn switch) over the same DP machinery
    from big_vision_amd.trainers.proj.image_text import contrastive as siglip
  model = two_towers.Model(image=IMAGE_CFG, text=TEXT_CFG, out_dim=(None, 128), temperature_init=10.0,
                           bias_init=-10.0)
  config = _config(**kw)
  state, _ = siglip.make_train_state(model, config, tuple(image.shape), tuple(text.shape), rng=0, comm=comm,
                                     total_steps=config.total_steps)
  fn = siglip.make_update_fn(model, config, comm=comm)
  if "sharding_strategy" in kw:
    # "fsdp" shards the PARAMETERS too (round 6): no rank holds the flat fp32 master, each holds its slice (Adam: equal
    # slices; Adafactor: its run of whole tensors) and the replicated entries (everything that is not a matmul kernel)
    st0, opt0 = state["params"].store, state["opt"]
    assert st0.master_sharded and st0.master is None and st0.master_own.numel() == opt0.hi - opt0.lo
    adafactor = "adafactor" in kw.get("optax_name", "")
    assert st0.master_own.numel() + st0.master_small.numel() < ((0.75 if adafactor else 0.6) if comm.size > 1 else 1.3) * st0.count
    if adafactor and comm.size > 1:     # the momentum exists for the own run only
      assert opt0.mu.numel() == max(4, opt0.hi - opt0.lo) < 0.75 * st0.trainable_count
  state, meas = fn(state, None, {"image": image, "labels": text})
  torch.cuda.synchronize()
  store = state["params"].store
  grads = {k: v.detach().cpu().double().clone() for k, v in u.tree_flatten_with_names(store.tree("grad"))[0]}
  # (fsdp: the fp32 parameters are sharded - full_tree() gathers the owners' slices; a plain tree otherwise)
  params = {k: v.detach().cpu().double().clone() for k, v in u.tree_flatten_with_names(store.full_tree())[0]}
  more = int(kw.get("extra_steps", 0))
  if more:   # further steps on the same batch; `more_digest` = a hash of every parameter's BITS after them
    for _ in range(more):
      state, meas2 = fn(state, None, {"image": image, "labels": text})
    torch.cuda.synchronize()
    import hashlib
    h = hashlib.sha256()
    for k, v in u.tree_flatten_with_names(store.full_tree())[0]:
      h.update(k.encode()); h.update(v.detach().cpu().contiguous().numpy().tobytes())
    sh = store.shadow.detach().cpu().view(torch.int16).numpy().tobytes()
    params["__bits__"] = (h.hexdigest(), hashlib.sha256(sh).hexdigest(), float(meas2["training_loss"].item()))
  return meas["training_loss"].item(), meas["l2_grads"].item(), grads, params


def _worker(rank, world, port, out, kw):
  sys.path.insert(0, ROOT)
  sys.path.insert(0, os.path.join(ROOT, "oracle"))
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                    LOCAL_RANK="0")
  import bv_oracle as O
  from big_vision_amd import dp
  torch.cuda.set_device(0)
  comm = dp.init_from_env(backend="gloo")
  image, text = O.synthetic_batch(1, 8, 64, 16, 100)
  n = 8 // world
  dev = torch.device("cuda:0")
  loss, gn, grads, params = _step(comm, image[rank * n:(rank + 1) * n].to(dev), text[rank * n:(rank + 1) * n].to(dev), **kw)
  bits = params.pop("__bits__", None)
  digest = {k: (v.sum().item(), v.abs().sum().item()) for k, v in params.items()}
  digest["__bits__"] = bits
  # numpy arrays are pickled by value (torch tensors would travel as shared-memory handles that
  # die with this process)
  out.put((rank, loss, gn, {k: v.numpy() for k, v in grads.items()} if rank == 0 else None, digest,
           {k: v.numpy() for k, v in params.items()} if (rank == 0 and "sharding_strategy" in kw) else None))
  comm.barrier()
  torch.distributed.destroy_process_group()


FSDP = dict(sharding_strategy=[(".*", "fsdp(axis='data', min_size_to_shard_mb=0)")],
            schedule=dict(decay_type="cosine", warmup_steps=0),      # no warm-up: the first step really moves the weights
            extra_steps=2)      # 3 steps in all: the ranks' fp32 masters AND bf16 shadows must stay bit-identical


# Adafactor under the fsdp placement: ownership by whole tensors (the factored statistics are per tensor)
FSDP_AF = dict(FSDP, optax_name="big_vision.scale_by_adafactor")


# LiT: the image tower frozen - the forward still forks (text tower on the side stream), the one-stream backward of the text tower
# hands its blocks to the overlapped gradient sync
LIT = dict(schedule=[("img/.*", None), (".*", dict(decay_type="cosine", warmup_steps=2))])


@pytest.mark.parametrize("world,kw", [(2, dict()), (2, dict(overlap_grad_sync=False)), (2, dict(microbatch=2)),
                                      (2, dict(loss_fn="softmax")), (2, dict(loss_fn="sigmoid")), (2, FSDP), (2, FSDP_AF),
                                      (4, dict()), (4, FSDP), (2, dict(tower_streams=1)), (2, LIT)],
                         ids=["w2-plain", "w2-no_overlap", "w2-microbatch", "w2-softmax", "w2-sigmoid", "w2-fsdp",
                              "w2-fsdp_adafactor", "w4-plain", "w4-fsdp", "w2-one_stream", "w2-lit"])
def test_two_ranks_match_single_process(dev, world, kw):
  import torch.multiprocessing as mp
  sys.path.insert(0, os.path.join(ROOT, "oracle"))
  import bv_oracle as O
  from big_vision_amd import dp
  image, text = O.synthetic_batch(1, 8, 64, 16, 100)
  single_kw = {k: kw[k] for k in ("loss_fn", "schedule", "optax_name") if k in kw}   # same trainer / schedule / optimizer, replicated, one process
  kw = dict(kw)
  loss1, gn1, g1, p1 = _step(dp.Comm(), image.to(dev), text.to(dev), **single_kw)
  ctx = mp.get_context("spawn")
  out = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, world, port, out, kw)) for r in range(world)]
  for p in procs:
    p.start()
  res = {}
  for _ in range(world):
    r = out.get(timeout=300)
    res[r[0]] = r[1:]
  for p in procs:
    p.join(60)
    assert p.exitcode == 0, f"rank process failed (exit {p.exitcode})"
  loss2, gn2, g2, p2, params2 = res[0]
  for r in range(1, world):
    assert abs(res[0][0] - res[r][0]) <= 1e-6 * abs(loss1), "ranks disagree on the global loss"
    assert res[0][3] == res[r][3], "replicated parameters diverged between the ranks after the update"
  assert abs(loss2 - loss1) <= 1e-4 * abs(loss1), (loss1, loss2)
  assert abs(gn2 - gn1) <= 2e-2 * gn1, (gn1, gn2)
  if "sharding_strategy" in kw:
    # three sharded steps (reduce-to-owner overlapped with the backward, Adam on the own slice, in-place exchange,
    # partial shadow refresh): every parameter bit and every bf16 shadow bit agrees between the ranks
    b0 = res[0][3]["__bits__"]
    for r in range(1, world):
      b1 = res[r][3]["__bits__"]
      assert b0 is not None and b0[0] == b1[0], "fp32 parameters differ between the ranks after 3 sharded steps"
      assert b0[1] == b1[1], "bf16 shadows differ between the ranks after 3 sharded steps"
      assert abs(b0[2] - b1[2]) <= 1e-6 * abs(b0[2]) and b0[2] < loss1, (b0[2], loss1)
    # "fsdp" placement: each rank's gradient buffer holds its PARTIAL sums (the optimizer reduce-scatters them),
    # each rank updated its own slice of the flat buffer and all-gathered the rest: the parameters after the
    # step must be the single-process step's.  Adam's first update is lr * g / (|g| + eps): a gradient that is
    # ~0 up to summation order may flip its sign, one update = lr = 1e-3.
    assert all(abs(res[0][1] - res[r][1]) <= 1e-9 * gn1 for r in range(1, world)), "ranks disagree on the global gradient norm"
    gnorm = math.sqrt(sum((v ** 2).sum().item() for v in g1.values()))
    adafactor = "adafactor" in kw.get("optax_name", "")
    for k, v in p1.items():
      d = (v - torch.from_numpy(params2[k])).abs()
      # one update moves a weight by at most lr (Adam) / lr (1 - momentum) (Adafactor: unfactored 1-D leaves are sign-like
      # on the first step too); a ~0 gradient may flip its sign between the two summation orders
      step_max = 2.2e-4 if adafactor else 2e-3 + 1e-6
      assert d.max().item() <= step_max, (k, d.max().item())
      if g1[k].norm().item() >= 1e-3 * gnorm:     # (tensors whose gradient is noise - the key bias - move by noise / (noise + eps))
        assert (d > 1e-6).double().mean().item() <= 0.05, (k, (d > 1e-6).double().mean().item())
    return
  gnorm = math.sqrt(sum((v ** 2).sum().item() for v in g1.values()))
  for k, v in g1.items():
    assert (v - torch.from_numpy(g2[k])).norm().item() <= 2e-2 * max(v.norm().item(), 1e-2 * gnorm), k


def _contract_worker(rank, world, port, out):
  sys.path.insert(0, ROOT)
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
  from big_vision_amd import dp
  torch.cuda.set_device(0)
  comm = dp.init_from_env(backend="gloo")
  out.put((rank,) + _bare_sharded_step(comm, rank, world))
  comm.barrier()
  torch.distributed.destroy_process_group()


def _bare_sharded_step(comm, rank, world):
  """A caller that is NOT one of the trainers: builds Optimizer(shard=True), writes its partial gradients and calls
  step() - never touching grad_sync().  Returns (bits of the updated masters, whole-state digest of state_dict())."""
  import hashlib
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  model = two_towers.Model(image=IMAGE_CFG, text=TEXT_CFG, out_dim=(None, 128), temperature_init=10.0, bias_init=-10.0)
  config = _config(schedule=dict(decay_type="cosine", warmup_steps=0),
                   **(dict(sharding_strategy=FSDP["sharding_strategy"]) if world > 1 else {}))
  state, _ = siglip.make_train_state(model, config, (8, 64, 64, 3), (8, 16), rng=0, comm=comm, total_steps=10)
  opt, store = state["opt"], state["params"].store
  assert bool(getattr(opt, "sharded", False)) == (world > 1)
  store.zero_grad()          # allocates the flat gradient buffer
  g = torch.Generator(device="cpu").manual_seed(7)
  full = torch.randn(store.trainable_count, generator=g) * 1e-2
  # rank r contributes (r + 1) / sum of the global gradient: the partial sums add up to `full` (not bit-exactly)
  share = (rank + 1) / sum(r + 1 for r in range(world))
  store.grad[:store.trainable_count].copy_((full * share).to(store.grad.device))
  opt.step()                                   # no grad_sync(): the step must reduce onto the owners itself
  store.grad[:store.trainable_count].copy_((full * share).to(store.grad.device))
  opt.step()                                   # and again (the stamp of the first step must not leak into the second)
  torch.cuda.synchronize()
  sd = opt.state_dict()                        # whole moments on every rank (a collective under fsdp)
  assert sd["mu"].numel() == store.trainable_count and sd["nu"].numel() == store.trainable_count
  dig = hashlib.sha256(sd["mu"].float().cpu().numpy().tobytes() + sd["nu"].float().cpu().numpy().tobytes()).hexdigest()
  return store.gather_master().detach().cpu().numpy(), dig, float(sd["mu"].float().abs().sum().item())   # (fsdp: the owners' fp32 slices gathered)


def test_sharded_step_without_a_driven_grad_sync_still_sums_onto_the_owners(dev):
  """Advisor r4: Optimizer(shard=True).step() used to assume that the trainer had driven grad_sync() to completion
  and otherwise updated from this rank's partial gradients.  Now finish() stamps the optimizer and a step without
  the stamp reduces [0, n_tr) itself.  Also: state_dict() of a sharded optimizer returns the WHOLE moments on every
  rank (it used to return the own slice and stale zeros elsewhere)."""
  import numpy as np
  import torch.multiprocessing as mp
  from big_vision_amd import dp
  ref_master, _, ref_mu = _bare_sharded_step(dp.Comm(), 0, 1)
  ctx = mp.get_context("spawn")
  out = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_contract_worker, args=(r, 2, port, out)) for r in range(2)]
  for p in procs:
    p.start()
  res = {}
  for _ in range(2):
    r = out.get(timeout=300)
    res[r[0]] = r[1:]
  for p in procs:
    p.join(60)
    assert p.exitcode == 0
  assert np.array_equal(res[0][0], res[1][0]), "ranks hold different parameters after two bare sharded steps"
  assert res[0][1] == res[1][1], "state_dict() must return the same WHOLE moments on every rank"
  # two Adam steps of lr <= 1e-3 from partial sums that differ from `full` in the last bits
  # (a ~0 gradient may flip the sign of its first updates: 2 lr per step)
  assert np.abs(res[0][0] - ref_master).max() <= 4.2e-3
  assert (np.abs(res[0][0] - ref_master) > 1e-6).mean() <= 0.05, "the update was not computed from the summed gradient"
  assert abs(res[0][2] - ref_mu) <= 1e-3 * ref_mu
