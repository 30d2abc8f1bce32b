"""Data-parallel communication layer: one process per GPU over RCCL (xGMI).

Re-expresses the reference's replicated-parameter data parallelism
(big_vision/sharding.py:83-101 `replicate`, batch split on dim 0 by
utils.py:1388-1409) and the explicit collectives of the pmap trainer
(trainers/proj/image_text/_deprecated_contrastive.py):

  all_gather(ztxt)                    :67-77,122   -> all_gather_rows
  transpose of that gather (psum_scatter inserted by JAX AD) -> reduce_scatter_rows
  pmean(grads) / pmean(loss)          :343-344     -> all_reduce_sum_ on the flat
                                                      grad buffer (sum: every rank
                                                      already uses the GLOBAL 1/B)

`torch.distributed` backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU
tests (world_size 2).  With world_size 1 every method is the identity.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


class Comm:
  def __init__(self, group=None, force=None):
    """force (default: env BV_DP_FORCE_COLLECTIVES=1): issue the collectives even in a group of ONE
    rank - a one-GPU box can then run the whole RCCL call path (all_gather_into_tensor,
    reduce_scatter_tensor, bucketed all-reduce on the side stream) that N > 1 takes."""
    self.enabled = dist.is_available() and dist.is_initialized()
    self.group = group
    self.rank = dist.get_rank(group) if self.enabled else 0
    self.size = dist.get_world_size(group) if self.enabled else 1
    if force is None:
      force = os.environ.get("BV_DP_FORCE_COLLECTIVES", "0") == "1"
    self.active = self.enabled and (self.size > 1 or bool(force))

  # ------------------------------------------------------------ embeddings --
  def all_gather_rows(self, x: torch.Tensor) -> torch.Tensor:
    """[n, E] on every rank -> [size*n, E], rank r's rows at [r*n, (r+1)*n)."""
    if not self.active:
      return x
    out = torch.empty((self.size * x.shape[0],) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
    dist.all_gather_into_tensor(out, x.contiguous(), group=self.group)
    return out

  def reduce_scatter_rows(self, x: torch.Tensor) -> torch.Tensor:
    """[size*n, E] partial sums -> this rank's [n, E] block of the total."""
    if not self.active:
      return x
    n = x.shape[0] // self.size
    out = torch.empty((n,) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
    if dist.get_backend(self.group) == "gloo":  # gloo has no reduce_scatter_tensor
      tmp = x.contiguous().clone()
      dist.all_reduce(tmp, group=self.group)
      out.copy_(tmp[self.rank * n:(self.rank + 1) * n])
    else:
      dist.reduce_scatter_tensor(out, x.contiguous(), group=self.group)
    return out

  # ------------------------------------------------ sharded optimizer ("fsdp") --
  def reduce_scatter_flat(self, flat: torch.Tensor, S: int) -> torch.Tensor:
    """flat [n] partial sums on every rank -> this rank's slice [rank*S, (rank+1)*S) of the total, length S
    (zero-padded beyond n; size * S >= n)."""
    n = flat.numel()
    if not self.active:
      out = torch.zeros(S, device=flat.device, dtype=flat.dtype)
      out[:min(n, S)] = flat[:min(n, S)]
      return out
    pad = torch.zeros(self.size * S, device=flat.device, dtype=flat.dtype)
    pad[:n] = flat
    if dist.get_backend(self.group) == "gloo":  # gloo has no reduce_scatter_tensor
      dist.all_reduce(pad, group=self.group)
      return pad[self.rank * S:(self.rank + 1) * S].clone()
    out = torch.empty(S, device=flat.device, dtype=flat.dtype)
    dist.reduce_scatter_tensor(out, pad, group=self.group)
    return out

  def all_gather_flat_(self, flat: torch.Tensor, lo: int, hi: int, S: int):
    """In place: every rank contributes flat[lo:hi] (its slice, slices are S apart) and receives all others."""
    if not self.active:
      return
    n = flat.numel()
    mine = torch.zeros(S, device=flat.device, dtype=flat.dtype)
    mine[:hi - lo] = flat[lo:hi]
    if dist.get_backend(self.group) == "gloo":
      parts = [torch.empty_like(mine) for _ in range(self.size)]
      dist.all_gather(parts, mine, group=self.group)
      full = torch.cat(parts)
    else:
      full = torch.empty(self.size * S, device=flat.device, dtype=flat.dtype)
      dist.all_gather_into_tensor(full, mine, group=self.group)
    flat.copy_(full[:n])

  def broadcast_ranges_(self, flat: torch.Tensor, bounds):
    """In place: the range [bounds[r], bounds[r + 1]) of `flat` is sent by rank r to everybody - the parameter
    exchange of the sharded optimizer without staging buffers (size ranks, size broadcasts)."""
    if not self.active:
      return
    for r in range(self.size):
      a, b = int(bounds[r]), int(bounds[r + 1])
      if b > a:
        src = dist.get_global_rank(self.group, r) if self.group is not None else r
        dist.broadcast(flat[a:b], src=src, group=self.group)

  def broadcast_slices_(self, flat: torch.Tensor, S: int):
    """broadcast_ranges_ with equal slices of length S (cut at the end of `flat`)."""
    n = flat.numel()
    self.broadcast_ranges_(flat, [min(n, r * S) for r in range(self.size + 1)])

  # ----------------------------------------------------------------- grads --
  def all_reduce_sum_(self, flat: torch.Tensor, bucket_bytes: int = 256 << 20):
    """In-place sum over ranks of a flat buffer, in large buckets.

    xGMI is point-to-point (7 links/GPU); RCCL picks its own ring/tree/direct
    algorithm per message, large messages amortise launch latency.
    """
    if not self.active:
      return
    step = max(1, bucket_bytes // flat.element_size())
    for off in range(0, flat.numel(), step):
      dist.all_reduce(flat[off:off + step], group=self.group)

  def all_reduce_scalars_(self, t: torch.Tensor):
    if self.active:
      dist.all_reduce(t, group=self.group)

  def barrier(self):
    if self.active:
      dist.barrier(group=self.group)


_COLLECTIVE_STREAMS = {}


def _collective_stream(device):
  key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
  if key not in _COLLECTIVE_STREAMS:
    _COLLECTIVE_STREAMS[key] = torch.cuda.Stream(device=device)
  return _COLLECTIVE_STREAMS[key]


class GradSync:
  """Overlaps the gradient all-reduce with the backward pass.

  The backward finishes parameter gradients tower by tower, last layer first (the reference
  leaves this overlap to XLA's scheduler, _deprecated_contrastive.py:343).  `launch(lo, hi)`
  is called on the compute stream once elements [lo, hi) of the flat gradient buffer are
  final: the range is summed over ranks on a side stream (RCCL kernels run next to the
  remaining backward GEMMs); `finish()` reduces whatever was not launched and makes the
  compute stream wait for the side stream.  Every rank issues the same ranges in the same
  order.  CPU tensors (gloo tests) are reduced synchronously."""

  def __init__(self, comm: Comm, flat: torch.Tensor, bucket_bytes: int = 256 << 20):
    self.comm, self.flat, self.bucket_bytes = comm, flat, bucket_bytes
    self.done = []   # disjoint [lo, hi) already handed to the collective
    # ONE collective stream per device and process (a GradSync is built every step: torch.cuda.Stream() would walk its pool
    # of 32 streams, which HIP maps onto a handful of hardware queues - a step whose collective stream shares a queue with the
    # text tower's side stream serialises its all-reduces behind that tower's kernels)
    self.stream = _collective_stream(flat.device) if (flat.is_cuda and comm.active) else None

  def launch(self, lo: int, hi: int):
    lo, hi = max(0, int(lo)), min(int(hi), self.flat.numel())
    if not self.comm.active or hi <= lo:
      return
    for a, b in self.done:
      assert hi <= a or lo >= b, f"gradient range [{lo},{hi}) overlaps an already reduced range [{a},{b})"
    self.done.append((lo, hi))
    if self.stream is None:
      self._collective(lo, hi)
      return
    ready = torch.cuda.Event()
    ready.record()                         # on the compute stream: the range is final after this point
    with torch.cuda.stream(self.stream):
      self.stream.wait_event(ready)
      self._collective(lo, hi)

  def _collective(self, lo: int, hi: int):
    """What happens to a final range of the flat gradient buffer: here, the sum over ranks on every rank."""
    self.comm.all_reduce_sum_(self.flat[lo:hi], self.bucket_bytes)

  def launch_gaps(self, lo: int, hi: int):
    """Hands every not yet launched part of [lo, hi) to the collective (e.g. what is left of a tower
    once its backward is enqueued: embeddings, final norm, head)."""
    lo, hi = max(0, int(lo)), min(int(hi), self.flat.numel())
    pos = lo
    for a, b in sorted(self.done) + [(hi, hi)]:
      if b <= pos:
        continue
      if a >= hi:
        a = hi
      if a > pos:
        self.launch(pos, a)
      pos = max(pos, b)
      if pos >= hi:
        break

  def finish(self):
    """Reduces the complement of the launched ranges, then joins the side stream."""
    if not self.comm.active:
      return
    pos = 0
    for a, b in sorted(self.done) + [(self.flat.numel(), self.flat.numel())]:
      if a > pos:
        self._collective(pos, a)
      pos = max(pos, b)
    if self.stream is not None:
      torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)
    self.done = []


class GradShardSync(GradSync):
  """GradSync for the sharded optimizer ("fsdp" placement, optax.Optimizer(shard=True)): rank r OWNS the range
  [bounds[r], bounds[r + 1]) of the flat trainable gradient buffer (Adam: equal slices of whole 1024-element chunks;
  Adafactor: runs of whole tensors), and a final range is summed onto its owner(s) only (`reduce` per owner, in
  place) instead of onto every rank - half the bytes of an all-reduce, the other half being the parameter exchange
  after the update.  Same launch / launch_gaps / finish protocol, so the ranges the backward hands over block by
  block overlap the remaining GEMMs exactly like the replicated path; no staging copy (round 3 reduce-scattered a
  zero-padded 813 MB copy after the whole backward).  After finish() the own range of `flat` holds the global
  sum; the rest of the buffer holds partial sums nobody reads."""

  def __init__(self, comm: Comm, flat: torch.Tensor, bounds, bucket_bytes: int = 256 << 20, on_finish=None):
    bounds = [int(b) for b in bounds]
    assert len(bounds) == comm.size + 1 and bounds[0] == 0 and all(a <= b for a, b in zip(bounds, bounds[1:]))
    super().__init__(comm, flat[:bounds[-1]], bucket_bytes)
    self.bounds = bounds
    self.on_finish = on_finish     # optax.Optimizer stamps itself: the sharded step checks that the owners' sums exist

  def finish(self):
    super().finish()
    if self.on_finish is not None:
      self.on_finish()

  def _collective(self, lo: int, hi: int):
    import bisect
    comm = self.comm
    pos = lo
    while pos < hi:
      owner = bisect.bisect_right(self.bounds, pos) - 1
      end = min(hi, self.bounds[owner + 1])
      dst = dist.get_global_rank(comm.group, owner) if comm.group is not None else owner
      dist.reduce(self.flat[pos:end], dst=dst, group=comm.group)
      pos = end


def init_from_env(backend: str | None = None, overlap_channels: int | None = None) -> Comm:
  """Initialises torch.distributed from torchrun-style env vars (if present).

  overlap_channels: cap RCCL at this many channels (NCCL_MAX_NCHANNELS, only if the variable is not already
  set).  Pass it ONLY for a job whose gradient all-reduces overlap the persistent GEMMs of the backward
  (bench.py / the SigLIP trainer with overlap_grad_sync: one channel = one workgroup = one CU taken from the
  GEMMs, see RESERVED_CUS); the default leaves RCCL's channel count alone, so other nccl jobs in the
  process (large or multi-node all-reduces) are not throttled.  BV_NCCL_MAX_NCHANNELS overrides the value."""
  world = int(os.environ.get("WORLD_SIZE", "1"))
  force = os.environ.get("BV_DP_FORCE_COLLECTIVES", "0") == "1" and "RANK" in os.environ
  if (world > 1 or force) and not dist.is_initialized():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:     # BV_DP_BACKEND=gloo: ranks that share ONE GPU (RCCL refuses that; tests, bench smoke runs)
      backend = os.environ.get("BV_DP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
      torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
      # The collectives of a step move < 1 GB per 90 ms: a few channels are plenty, and every RCCL
      # channel is a workgroup that needs a CU of its own beside the persistent GEMMs (see RESERVED_CUS).
      cap = os.environ.get("BV_NCCL_MAX_NCHANNELS", overlap_channels)
      if cap and str(cap).lower() not in ("0", "none", "uncapped"):     # "uncapped": leave RCCL's own choice (A/B)
        os.environ.setdefault("NCCL_MAX_NCHANNELS", str(int(cap)))
    dist.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=world)
  return Comm()


_warned_uncapped = False


def warn_if_overlap_is_uncapped():
  """The overlapped gradient all-reduce is sized for RESERVED_CUS free CUs; an RCCL communicator created without
  a channel cap (init_from_env(overlap_channels=...) / NCCL_MAX_NCHANNELS) launches more workgroups than that,
  which then wait for - or delay - whole GEMM launches.  Same results, not the benchmarked configuration
  (advisor r3): say so once."""
  global _warned_uncapped
  if _warned_uncapped or not (dist.is_available() and dist.is_initialized()) or dist.get_backend() != "nccl":
    return
  if "NCCL_MAX_NCHANNELS" not in os.environ:
    _warned_uncapped = True
    import warnings
    warnings.warn("overlap_grad_sync with an uncapped RCCL: create the process group with "
                  f"dp.init_from_env(overlap_channels=dp.RESERVED_CUS) or set NCCL_MAX_NCHANNELS={RESERVED_CUS} "
                  "(the persistent GEMMs reserve that many CUs for the collectives)")


# CUs the persistent 256x256 GEMM leaves to RCCL while gradient all-reduces overlap the backward
# (BV_OPT_GEMM_RESERVE_CUS of the compute stream's bv_ctx): its workgroups fill a CU, so a collective launched beside it would otherwise
# wait for - or delay - a whole GEMM launch.  4 keeps the split-K choices of the B/16 shapes intact
# (252 = 36 x 7 = 9 x 28 work items) and costs the k-major GEMMs 1.6 % of the chip during the backward.
# BV_RESERVED_CUS overrides it (A/B on a multi-GPU node without a code change; 0 = reserve nothing).
RESERVED_CUS = int(os.environ.get("BV_RESERVED_CUS", "4"))


class reserve_cus_for_collectives:
  """`with reserve_cus_for_collectives(comm):` around the part of a step whose kernels overlap RCCL
  traffic (the backward).  No-op on an inactive communicator (one rank without forced collectives) and without a
  GPU library."""

  def __init__(self, comm):
    self.on = bool(comm is not None and comm.active and torch.cuda.is_available())

  def __enter__(self):
    if self.on:
      from big_vision_amd import ops
      self.ctx = ops.ctx()          # the context of the stream the backward's GEMMs are enqueued on
      self.old = self.ctx.set("gemm_reserve_cus", RESERVED_CUS)
    return self

  def __exit__(self, *exc):
    if self.on:
      self.ctx.set("gemm_reserve_cus", self.old)
