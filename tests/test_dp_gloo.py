"""N>1 path on CPU: 2, 4 and 8 `gloo` processes drive big_vision_amd.dp.Comm
(the RCCL layer on GPUs) through the sharded sigmoid-loss exchange of
trainers/proj/image_text/siglip.py ("convention A"):

  all_gather(ztxt) -> local rows [n,B] with the positive diagonal at r*n+i ->
  G with the GLOBAL 1/B -> dzimg local, dztxt partial [B,E] -> reduce_scatter
  -> parameter-gradient all-reduce(SUM) of partial sums.

The per-rank arithmetic here is plain torch-CPU (the HIP kernels need a GPU);
what is under test is the collective choreography: its result must equal the
oracle's single-program global-batch loss and autograd gradients
(siglip.py:287-308) and the reference pmap convention (1/n + pmean,
_deprecated_contrastive.py:139-141,343-344).
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _worker(rank, world, port, n, E, out):
  sys.path.insert(0, ROOT)
  sys.path.insert(0, os.path.join(ROOT, "oracle"))
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                    WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
  from big_vision_amd import dp
  import bv_oracle as O
  comm = dp.init_from_env(backend="gloo")
  assert comm.size == world and comm.rank == rank
  B = n * world
  g = torch.Generator().manual_seed(0)
  zimg_all = torch.nn.functional.normalize(torch.randn(B, E, generator=g, dtype=torch.float64), dim=1)
  ztxt_all = torch.nn.functional.normalize(torch.randn(B, E, generator=g, dtype=torch.float64), dim=1)
  w = torch.randn(E, generator=g, dtype=torch.float64)       # a "parameter" shared by all ranks
  tp, b = torch.tensor([2.3], dtype=torch.float64), torch.tensor([-10.0], dtype=torch.float64)
  zimg, ztxt = zimg_all[rank * n:(rank + 1) * n], ztxt_all[rank * n:(rank + 1) * n]

  # ---- convention A with the product's collectives
  gathered = comm.all_gather_rows(ztxt.contiguous())
  assert torch.equal(gathered, ztxt_all), "all_gather_rows must place rank r at rows [r*n,(r+1)*n)"
  t = torch.exp(tp)
  S = t * (zimg * w) @ gathered.T + b                       # embeddings depend on the shared param w
  m = -torch.ones(n, B, dtype=torch.float64)
  m[torch.arange(n), rank * n + torch.arange(n)] = 1.0
  loss_share = -(O.log_sigmoid(m * S)).sum() / B
  G = -(1.0 / B) * m * torch.sigmoid(-m * S)
  dzimg = t * G @ gathered                                    # d/d(zimg*w)
  dztxt_partial = t * G.T @ (zimg * w)
  dztxt = comm.reduce_scatter_rows(dztxt_partial.contiguous())
  flat = torch.cat([(dzimg * zimg).sum(0), (G * (S - b)).sum().view(1), G.sum().view(1)])  # [dw, dt', db]
  comm.all_reduce_sum_(flat, bucket_bytes=64)                 # several small buckets
  loss = loss_share.view(1).clone()
  comm.all_reduce_scalars_(loss)

  # ---- oracle: single-program global loss + autograd
  wr, tr, br = w.clone().requires_grad_(True), tp.clone().requires_grad_(True), b.clone().requires_grad_(True)
  zt = ztxt_all.clone().requires_grad_(True)
  ref, _ = O.siglip_loss_global(zimg_all * wr, zt, torch.exp(tr), br)
  ref.backward()
  assert abs(loss.item() - ref.item()) < 1e-12
  assert torch.allclose(flat[:E], wr.grad, atol=1e-12)
  assert abs(flat[E].item() - tr.grad.item()) < 1e-12 and abs(flat[E + 1].item() - br.grad.item()) < 1e-12
  assert torch.allclose(dztxt, zt.grad[rank * n:(rank + 1) * n], atol=1e-12)

  # ---- reference pmap convention B: local 1/n loss, then pmean == convention A
  shards = list(ztxt_all.split(n))
  lb = O.sigmoid_loss_per_device(zimg * w, shards, rank, t, b).view(1).clone()
  comm.all_reduce_scalars_(lb)
  assert abs(lb.item() / world - ref.item()) < 1e-12
  # ---- GradSync: ranges handed over early + the complement in finish() == one all-reduce
  base = torch.arange(1000, dtype=torch.float64)
  buf = base * (rank + 1)
  sync = dp.GradSync(comm, buf, bucket_bytes=256)
  sync.launch(100, 300)
  sync.launch(600, 900)
  sync.finish()
  assert torch.equal(buf, base * sum(r + 1 for r in range(world))), "GradSync must sum every element exactly once"
  buf2 = base * (rank + 1)
  sync.launch(0, 1000); sync.finish()      # (state was reset by finish)
  buf2_sync = dp.GradSync(comm, buf2); buf2_sync.finish()
  assert torch.equal(buf2, base * sum(r + 1 for r in range(world)))
  # ---- sharded optimizer ("fsdp" placement): reduce_scatter of the flat gradient buffer into per-rank slices,
  # all_gather of the updated slices; the buffer length is not a multiple of the slice length
  nflat, S = 1003, 1024 // world          # world * S >= nflat; the last slice is cut short by the end of the buffer
  ranks_sum = sum(r + 1 for r in range(world))
  part = torch.arange(nflat, dtype=torch.float64) * (rank + 1)
  mine = comm.reduce_scatter_flat(part, S)
  tot = torch.zeros(world * S, dtype=torch.float64)
  tot[:nflat] = torch.arange(nflat, dtype=torch.float64) * ranks_sum
  assert torch.equal(mine, tot[rank * S:(rank + 1) * S]), "reduce_scatter_flat: slice r = elements [r*S, (r+1)*S) of the sum"
  lo, hi = rank * S, min((rank + 1) * S, nflat)
  params = torch.full((nflat,), -1.0, dtype=torch.float64)
  params[lo:hi] = mine[:hi - lo] + 1.0          # "update" of this rank's slice only
  comm.all_gather_flat_(params, lo, hi, S)
  assert torch.equal(params, tot[:nflat] + 1.0), "all_gather_flat_: every rank must end with every slice"
  # ---- the same through the trainer-facing objects: GradShardSync sums every range onto its owner only (in
  # place, ranges handed over early + the complement in finish()), broadcast_slices_ exchanges the updated slices
  part = torch.arange(nflat, dtype=torch.float64) * (rank + 1)
  buf = torch.cat([part, torch.full((7,), -5.0, dtype=torch.float64)])   # 7 frozen elements behind the trainable prefix
  ssync = dp.GradShardSync(comm, buf, [min(nflat, r * S) for r in range(world + 1)])
  ssync.launch(400, 700)          # straddles the slice boundary at 512 (world 2: two owners; world 8: slices 3, 4, 5)
  ssync.launch(0, 100)
  ssync.finish()
  assert torch.equal(buf[lo:hi], tot[lo:hi]), "GradShardSync: the own slice must hold the global sum"
  assert torch.equal(buf[nflat:], torch.full((7,), -5.0, dtype=torch.float64)), "frozen tail touched"
  params = torch.full((nflat,), -1.0, dtype=torch.float64)
  params[lo:hi] = buf[lo:hi] + 1.0
  comm.broadcast_slices_(params, S)
  assert torch.equal(params, tot[:nflat] + 1.0), "broadcast_slices_: every rank must end with every slice"
  # uneven, tensor-aligned ownership (Adafactor under fsdp): world 2 -> [0, 355), [355, 1003); world 8 -> eight
  # ranges of growing length; from 4 ranks on one rank owns NOTHING (a model with fewer large tensors than ranks)
  ub = [int(round(nflat * (r / world) ** 1.5)) for r in range(world + 1)]
  if world >= 4:
    ub[2] = ub[1]
  assert ub[0] == 0 and ub[-1] == nflat and all(a <= b for a, b in zip(ub, ub[1:]))
  buf = torch.arange(nflat, dtype=torch.float64) * (rank + 1)
  us = dp.GradShardSync(comm, buf, ub)
  us.launch(ub[1] - 50, ub[1] + 50)     # across the first owner boundary (and the empty range behind it)
  if world > 2:
    us.launch(ub[-2] - 10, nflat)       # across the last one
  us.finish()
  assert torch.equal(buf[ub[rank]:ub[rank + 1]], tot[ub[rank]:ub[rank + 1]])
  pr = torch.full((nflat,), -1.0, dtype=torch.float64)
  pr[ub[rank]:ub[rank + 1]] = buf[ub[rank]:ub[rank + 1]]
  comm.broadcast_ranges_(pr, ub)
  assert torch.equal(pr, tot[:nflat])
  comm.barrier()
  out.put((rank, loss.item()))
  dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_sigmoid_loss_equals_global(world):
  """world 4 / 8: row offsets r * n for r >= 2, slices that straddle several owners, 8-way tensor-aligned (uneven,
  one empty) ownership - the shapes of the headline's 8-GPU job (reference
  _deprecated_contrastive.py:67-77,117-141,343-344)."""
  ctx = mp.get_context("spawn")
  out = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, world, port, 5, 16, out)) for r in range(world)]
  for p in procs:
    p.start()
  for p in procs:
    p.join(240)
    assert p.exitcode == 0, f"rank process failed (exit {p.exitcode})"
  res = dict(out.get(timeout=5) for _ in range(world))
  assert len(res) == world and all(abs(res[0] - res[r]) < 1e-15 for r in range(world))


def test_single_process_comm_is_identity():
  sys.path.insert(0, ROOT)
  from big_vision_amd import dp
  c = dp.Comm()
  x = torch.arange(6.0).view(3, 2)
  assert c.size == 1 and c.rank == 0
  assert c.all_gather_rows(x) is x and c.reduce_scatter_rows(x) is x
  c.all_reduce_sum_(x); c.all_reduce_scalars_(x); c.barrier()
  assert torch.equal(x, torch.arange(6.0).view(3, 2))
  flat = torch.arange(5.0)
  assert torch.equal(c.reduce_scatter_flat(flat, 8), torch.tensor([0., 1., 2., 3., 4., 0., 0., 0.]))
  c.all_gather_flat_(flat, 0, 5, 8)
  assert torch.equal(flat, torch.arange(5.0))
