"""The N > 1 path over RCCL (torch.distributed backend "nccl"), which the gloo tests cannot reach:
  * ONE rank on one GPU with the collectives forced on (BV_DP_FORCE_COLLECTIVES=1): every RCCL call of
    the step is issued for real - all_gather_into_tensor / reduce_scatter_tensor of the embeddings, the
    per-block gradient all-reduces on dp.GradSync's side stream, the scalar all-reduce - and the
    result must be the plain single-process step (sums over one rank are the identity);
  * TWO ranks on two GPUs when the box has them (skipped otherwise): must match the single-process
    step on the whole batch, like tests/test_dp_two_ranks_gpu.py does over gloo."""
import math
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out, force, kw=None):
  sys.path.insert(0, ROOT)
  sys.path.insert(0, os.path.join(ROOT, "oracle"))
  sys.path.insert(0, os.path.join(ROOT, "tests"))
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                    LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
  if force:
    os.environ["BV_DP_FORCE_COLLECTIVES"] = "1"
  import bv_oracle as O
  import test_dp_two_ranks_gpu as T
  from big_vision_amd import dp
  torch.cuda.set_device(rank)
  comm = dp.init_from_env(backend="nccl", overlap_channels=dp.RESERVED_CUS)   # the benchmarked configuration
  assert comm.active and comm.size == world
  image, text = O.synthetic_batch(1, 8, 64, 16, 100)
  n = 8 // world
  dev = torch.device("cuda", rank)
  loss, gn, grads, params = T._step(comm, image[rank * n:(rank + 1) * n].to(dev), text[rank * n:(rank + 1) * n].to(dev), **(kw or {}))
  params.pop("__bits__", None)   # (bit digests of the multi-step FSDP case of test_dp_two_ranks_gpu)
  digest = {k: (v.sum().item(), v.abs().sum().item()) for k, v in params.items()}
  out.put((rank, loss, gn, {k: v.numpy() for k, v in grads.items()} if rank == 0 else None, digest,
           {k: v.numpy() for k, v in params.items()} if (rank == 0 and kw) else None))
  comm.barrier()
  torch.distributed.destroy_process_group()


def _run(world, force, kw=None):
  import torch.multiprocessing as mp
  sys.path.insert(0, os.path.join(ROOT, "tests"))
  import test_dp_two_ranks_gpu as T
  ctx = mp.get_context("spawn")
  out = ctx.Queue()
  port = T._free_port()
  procs = [ctx.Process(target=_worker, args=(r, world, port, out, force, kw)) for r in range(world)]
  for p in procs:
    p.start()
  res = {}
  for _ in range(world):
    r = out.get(timeout=600)
    res[r[0]] = r[1:]
  for p in procs:
    p.join(120)
    assert p.exitcode == 0, f"rank process failed (exit {p.exitcode})"
  return res


def _check(res, dev):
  import bv_oracle as O
  import test_dp_two_ranks_gpu as T
  from big_vision_amd import dp
  image, text = O.synthetic_batch(1, 8, 64, 16, 100)
  loss1, gn1, g1, _ = T._step(dp.Comm(), image.to(dev), text.to(dev))
  loss2, gn2, g2 = res[0][:3]
  assert abs(loss2 - loss1) <= 1e-4 * abs(loss1), (loss1, loss2)
  assert abs(gn2 - gn1) <= 2e-2 * gn1, (gn1, gn2)
  gnorm = math.sqrt(sum((v ** 2).sum().item() for v in g1.values()))
  for k, v in g1.items():
    assert (v - torch.from_numpy(g2[k])).norm().item() <= 2e-2 * max(v.norm().item(), 1e-2 * gnorm), k
  for r in res:
    assert res[r][3] == res[0][3], "replicated parameters diverged between the ranks after the update"


def test_rccl_call_path_on_one_gpu(dev):
  sys.path.insert(0, os.path.join(ROOT, "tests"))
  _check(_run(1, force=True), dev)


def test_fsdp_placement_over_rccl_on_one_gpu(dev):
  """config.sharding_strategy fsdp on a ONE-rank RCCL group with the collectives forced on: the per-range `reduce`
  of the gradients onto their owner (dp.GradShardSync, on the side stream during the backward) and the `broadcast`
  of the updated slices really go through ProcessGroupNCCL (over one rank both are the identity), the sharded
  Adam step runs on the slice [0, P) for three steps (the comparison below is after the first): the parameters after the
  step must be the replicated single-process step's (up to the sign of an Adam update of a ~0 gradient: the bias
  gradients are summed with fp32 atomics, run-to-run order noise; one update = lr = 1e-3)."""
  sys.path.insert(0, os.path.join(ROOT, "tests"))
  import bv_oracle as O
  import test_dp_two_ranks_gpu as T
  from big_vision_amd import dp
  res = _run(1, force=True, kw=T.FSDP)
  image, text = O.synthetic_batch(1, 8, 64, 16, 100)
  loss1, gn1, g1, p1 = T._step(dp.Comm(), image.to(dev), text.to(dev), schedule=T.FSDP["schedule"])
  loss2, gn2, _, _, params2 = res[0]
  assert abs(loss2 - loss1) <= 1e-6 * abs(loss1) and abs(gn2 - gn1) <= 1e-5 * gn1
  gnorm = math.sqrt(sum((v ** 2).sum().item() for v in g1.values()))
  moved = 0
  for k, v in p1.items():
    d = (v - torch.from_numpy(params2[k])).abs()
    assert d.max().item() <= 2e-3 + 1e-6, (k, d.max().item())
    if g1[k].norm().item() >= 1e-3 * gnorm:
      assert (d > 1e-6).double().mean().item() <= 0.02, (k, (d > 1e-6).double().mean().item())
    moved += int((d <= 1e-9).sum().item())
  assert moved > 0.9 * sum(v.numel() for v in p1.values())      # >90 % of all parameters agree to the last bit


def test_two_ranks_over_rccl(dev):
  if torch.cuda.device_count() < 2:
    pytest.skip("needs two GPUs (the driver's multi-GPU box); the one-GPU test above issues the same RCCL calls")
  sys.path.insert(0, os.path.join(ROOT, "tests"))
  _check(_run(2, force=False), dev)


def test_bench_stdout_is_one_json_line_with_rccl_in_the_loop():
  """bench.py contract: rank 0 prints ONE JSON line.  RCCL writes a version banner to stdout when a
  communicator is created; bench.py points fd 1 at stderr and writes its line to the saved descriptor.
  Runs the real benchmark command (small batch) on a one-rank RCCL group with the collectives forced on."""
  import json
  import subprocess
  import test_dp_two_ranks_gpu as T
  env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
             MASTER_PORT=str(T._free_port()), BV_DP_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
  r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--global-batch", "32", "--steps", "1",
                      "--warmup", "1", "--no-cpu-baseline", "--no-roofline"], env=env, capture_output=True,
                     text=True, timeout=600)
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [l for l in r.stdout.splitlines() if l.strip()]
  assert len(lines) == 1, r.stdout
  line = json.loads(lines[0])
  assert line["n_gpus"] == 1 and line["config"]["global_batch"] == 32 and math.isfinite(line["value"])


def test_bv_comm_c_layer_on_one_rank(dev):
  """include/bvhip.h `bv_comm_*` (init / all_gather / reduce_scatter / all_reduce_bucket / destroy):
  the collectives of the step as C entry points over RCCL, bound at run time to the RCCL already in the process.
  One-rank communicator on the MI355X box: every call really goes through RCCL and must be the identity."""
  import ctypes
  from big_vision_amd import _lib
  lib = _lib.load()
  ver = ctypes.c_int(0)
  assert lib.bv_comm_version(ctypes.byref(ver)) == 0 and ver.value >= 20000, lib.bv_last_error()
  uid = ctypes.create_string_buffer(128)
  assert lib.bv_comm_unique_id(uid) == 0, lib.bv_last_error()
  comm = ctypes.c_void_p()
  torch.cuda.set_device(dev)
  assert lib.bv_comm_init(uid, 0, 1, ctypes.byref(comm)) == 0, lib.bv_last_error()
  try:
    s = torch.cuda.current_stream().cuda_stream
    x = torch.randn(4096, device=dev)
    y = torch.empty_like(x)
    assert lib.bv_comm_all_gather(comm, x.data_ptr(), y.data_ptr(), x.numel(), 0, s) == 0, lib.bv_last_error()
    z = torch.empty_like(x)
    assert lib.bv_comm_reduce_scatter(comm, x.data_ptr(), z.data_ptr(), x.numel(), 0, s) == 0, lib.bv_last_error()
    w = x.clone()
    assert lib.bv_comm_all_reduce_bucket(comm, w.data_ptr(), w.numel(), 1000, 0, s) == 0, lib.bv_last_error()   # 5 buckets
    b = x.to(torch.bfloat16)
    b2 = torch.empty_like(b)
    assert lib.bv_comm_all_gather(comm, b.data_ptr(), b2.data_ptr(), b.numel(), 1, s) == 0, lib.bv_last_error()
    torch.cuda.synchronize()
    assert torch.equal(y, x) and torch.equal(z, x) and torch.equal(w, x) and torch.equal(b2, b)
    assert lib.bv_comm_all_gather(comm, x.data_ptr(), y.data_ptr(), 0, 0, s) != 0      # empty message is refused
    assert b"bad arguments" in lib.bv_last_error()
  finally:
    assert lib.bv_comm_destroy(comm) == 0


def test_bench_line_of_two_ranks_sharing_one_gpu():
  """The N > 1 path of bench.py end to end on a one-GPU box: `python bench.py --gpus 2` spawns its two ranks (both on
  GPU 0, gloo: RCCL refuses two ranks per device), rank 0 times the CPU oracle before the rendezvous, both ranks run the
  sharded-loss step with the overlapped gradient sync, EVERY rank takes part in the closing all-reduce of ones (round 6
  found it issued by rank 0 alone: a hang on the first multi-GPU run) and rank 0 prints ONE complete line."""
  import json
  import subprocess
  env = dict(os.environ, BV_BENCH_SHARE_GPU="1", BV_DP_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
  for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
    env.pop(k, None)
  r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--global-batch", "64", "--steps", "2",
                      "--warmup", "1", "--cpu-sample", "2"], env=env, capture_output=True, text=True, timeout=900)
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [l for l in r.stdout.splitlines() if l.strip()]
  assert len(lines) == 1, r.stdout[-1000:]
  d = json.loads(lines[0])
  assert d["n_gpus"] == 2 and d["config"]["per_gpu_batch"] == 32 and d["config"]["parallelism"] == "dp2"
  assert d["rccl"]["ranks"] == 2 and d["rccl"]["allreduce_of_ones"] == 2.0 and d["rccl"]["backend"] == "gloo"
  assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
  assert d["roofline"]["bound"] == "mfma" and d["roofline"]["launches"] > 0 and "measured_on" in d["roofline"]
  assert "shared_gpu" in d["config"] and d["value"] > 0 and math.isfinite(d["config"]["final_loss"])


def test_bench_line_under_torchrun_sharing_one_gpu():
  """The driver's own N > 1 launch command - `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr
  127.0.0.1 --master-port P bench.py --gpus 2 ...` - on a one-GPU box: the launcher numbers LOCAL_RANK 0 / 1, BV_BENCH_SHARE_GPU
  pins both ranks to GPU 0 (gloo).  Exactly one line on stdout, complete, rc 0."""
  import json
  import socket
  import subprocess
  with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
  env = dict(os.environ, BV_BENCH_SHARE_GPU="1", BV_DP_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
  for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
    env.pop(k, None)
  r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                      "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--global-batch",
                      "64", "--steps", "2", "--warmup", "1", "--cpu-sample", "2"], env=env, capture_output=True, text=True,
                     timeout=900)
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [l for l in r.stdout.splitlines() if l.strip()]
  assert len(lines) == 1, r.stdout[-1000:]
  d = json.loads(lines[0])
  assert d["n_gpus"] == 2 and d["config"]["per_gpu_batch"] == 32 and d["config"]["parallelism"] == "dp2"
  assert d["rccl"]["ranks"] == 2 and d["rccl"]["allreduce_of_ones"] == 2.0
  assert d["cpu_baseline"]["value"] > 0 and d["roofline"]["launches"] > 0 and "shared_gpu" in d["config"]
  assert d["value"] > 0 and math.isfinite(d["config"]["final_loss"])
