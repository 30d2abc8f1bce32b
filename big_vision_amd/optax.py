"""Optimizer chain of big_vision/optax.py:75-149 as ONE fused HIP kernel launch.

`make(config, store, sched_kw=...)` reads the same config fields as
`bv_optax.make` (schedule, grad_clip_norm, optax_name, optax, lr, lr_mults, wd,
wd_mults) and resolves them per Flax leaf name with the reference's first-match
regex semantics (utils.py:1195-1212).  The chain

  clip_by_global_norm(not frozen) -> scale_by_adam -> scale(lr) -> lr_mults ->
  add_decayed_weights -> scale_by_schedule / set_to_zero(frozen) -> scale(-1)
  -> optax.apply_updates

runs as bv_sqnorm + bv_adam_step over the flat parameter buffer.  Frozen
parameters (schedule None) live past the trainable prefix of the store: no
gradient, no Adam state (optax_test.py:301-318), excluded from the clip norm
(optax.py:105) and from l2_grads (trainers/proj/image_text/siglip.py:316).
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch

from big_vision_amd import ops
from big_vision_amd import utils as u
from big_vision_amd.params import ParamStore, make_masks

MAX_SCHED = 8
ADAFACTOR_NAMES = ("big_vision.scale_by_adafactor", "scale_by_adafactor")



def _refuse_per_example_clip(config):
  """`config.grad_clip_per_example` (reference optax.py:100-103 -> `clip_by_per_example_global_norm`,
  optax.py:54-72) clips PER-EXAMPLE gradients: its update_fn takes a gradient tree whose leaves carry a
  leading batch axis (`grads_flat[0].shape[0]`), clips every example's global norm and averages.  Neither
  trainer on this path produces such a tree - `train.py:275-315` and `trainers/proj/image_text/siglip.py:
  271-323` differentiate the batch-mean loss - and the fused step here accumulates the batch-summed
  gradient in the kernels (dW GEMMs reduce over the batch), so the option cannot be honoured; it is
  refused instead of silently clipping the wrong quantity."""
  if config.get("grad_clip_per_example"):
    raise NotImplementedError(
        "config.grad_clip_per_example (big_vision/optax.py:100-103, clip_by_per_example_global_norm :54-72) "
        "needs per-example gradient trees [batch, ...]; this step produces the batch gradient only "
        "(as big_vision's own train.py / siglip.py trainers do). Use grad_clip_norm alone.")


def factored_dims(shape, min_dim_size_to_factor=32):
  """optax/_src/factorized.py `_factored_dims` (factored=True): None, or (d1, d0) = the axes of the
  second-largest and the largest dimension (numpy argsort order on ties)."""
  if len(shape) < 2:
    return None
  order = np.argsort(np.asarray(shape), kind="stable")
  if shape[order[-2]] < min_dim_size_to_factor:
    return None
  return int(order[-2]), int(order[-1])


def frozen_patterns(config) -> List[str]:
  """Regexes of config.schedule whose schedule is None (= frozen params).

  Only valid as a pre-filter when earlier patterns cannot shadow later ones;
  `make` re-validates against the store with full first-match semantics.
  """
  schedule = config.get("schedule", {})
  if not isinstance(schedule, (tuple, list)):
    return []
  return [p for p, s in schedule if s is None]


def frozen_leaves(config, leaf_names) -> set:
  """Exact first-match resolution of frozen leaves (optax.py:82-84,152-174)."""
  schedule = config.get("schedule", {})
  if not isinstance(schedule, (tuple, list)):
    return set()
  pats, scheds = zip(*schedule)
  masks = make_masks(leaf_names, pats)
  return {n for n in leaf_names for m, s in zip(masks, scheds) if s is None and m[n]}


class Optimizer:
  """State + update of the fused chain (the `opt` half of the train state)."""

  def __init__(self, config, store: ParamStore, *, sched_kw, comm=None, shard=False):
    """comm / shard: placement of the optimizer.  shard=False (config.sharding_strategy "replicate"): every rank
    holds the whole state and applies the whole update to gradients the trainer has all-reduced.  shard=True
    ("fsdp", sharding.py): rank r owns the 1/N slice [lo, hi) of the flat trainable buffer - its Adam moments
    exist only there; `store.grad` holds this rank's PARTIAL sums, the trainer sums every range onto its owner with
    the object `grad_sync()` returns (overlapped with the backward; `step()` does it itself if nobody did), then
    `step()` updates its slice and exchanges the parameters (the reference's FSDP rule shards parameters and optimizer state, sharding.py:104-139; see
    big_vision_amd/sharding.py for how the per-tensor axis rule maps onto flat slices)."""
    self.store = store
    self.comm, self.sharded = comm, bool(shard)
    self._cfg = {k: config.get(k) for k in ("lr_mults", "wd", "wd_mults", "grad_clip_norm") if config.get(k) is not None}
    dev = store.device
    leaves = store.leaf_names()
    # ---- schedules (optax.py:79-97)
    schedule = config.get("schedule", {})
    if not isinstance(schedule, (tuple, list)):
      schedule = [(".*", schedule)]
    pats, scheds = zip(*schedule)
    masks = make_masks(leaves, pats)
    not_covered = [n for n in leaves if not any(m[n] for m in masks)]
    assert not not_covered, f"All params must be covered (use `None` for freezing): {not_covered}"
    kw = dict(sched_kw)
    if "global_batch_size" in kw:
      kw["batch_size"] = kw.pop("global_batch_size")
    self.schedule_fns = []
    sched_idx_of_leaf: Dict[str, int] = {}
    frozen = set()
    for m, s in zip(masks, scheds):
      names = [n for n in leaves if m[n]]
      if s is None:
        frozen.update(names)
        continue
      s = dict(s)
      assert "base" not in s, s
      mult = s.pop("mult", 1.0)
      self.schedule_fns.append(u.create_learning_rate_schedule(base=mult, **kw, **s))
      for n in names:
        sched_idx_of_leaf[n] = len(self.schedule_fns) - 1
    assert len(self.schedule_fns) <= MAX_SCHED, "too many schedule groups"
    frozen_entries = {e for n in frozen for e in store.entries_of(n)}
    if frozen_entries != set(store.frozen):
      raise ValueError("The parameter store was not laid out for this config.schedule: "
                       f"frozen in config {sorted(frozen_entries)[:4]}... vs store {sorted(store.frozen)[:4]}...")
    # ---- lr multipliers, weight decay (optax.py:114-138)
    lr_mult = {n: 1.0 for n in leaves}
    if config.get("lr_mults"):
      p2, mults = zip(*config["lr_mults"])
      assert all(mu > 0 for mu in mults), (
          f"Use schedule=None for parameter freezing instead of lr_mults={mults}")
      for m, mu in zip(make_masks(leaves, p2), mults):
        for n in leaves:
          if m[n]:
            lr_mult[n] = mu
    assert "weight_decay" not in config, "Deprecated option. Use wd and schedule."
    assert config.get("weight_decay_decouple", True), "Coupled weight decay not supported anymore."
    wd = {n: 0.0 for n in leaves}
    if config.get("wd"):
      wd_mults = config.get("wd_mults", [(".*/kernel$", 1.0)])
      p3, mults = zip(*wd_mults)
      for m, mu in zip(make_masks(leaves, p3), mults):
        for n in leaves:
          if m[n]:
            wd[n] = config["wd"] * mu
    # ---- optimizer proper (optax.py:108-112)
    assert "optim" not in config, "Deprecated option, use config.optax."
    self.name = config["optax_name"]
    okw = dict(config.get("optax", {}) or {})
    if self.name == "scale_by_adam":
      self.b1, self.b2, self.eps = okw.get("b1", 0.9), okw.get("b2", 0.999), okw.get("eps", 1e-8)
      assert okw.get("eps_root", 0.0) == 0.0, "eps_root is not supported"
      mu_dtype = okw.get("mu_dtype")
      mu_dtype = torch.bfloat16 if str(mu_dtype) in ("bfloat16", "torch.bfloat16") else torch.float32
    elif self.name in ADAFACTOR_NAMES:
      self.clip_norm = float(config.get("grad_clip_norm") or 0.0)
      _refuse_per_example_clip(config)
      self.lr = float(config["lr"])
      self._init_adafactor(okw, lr_mult, wd, sched_idx_of_leaf)
      # "fsdp": the PARAMETERS are sharded too, as under Adam - by OWNERSHIP of whole tensors here (the factored
      # statistics are per tensor): the store keeps the fp32 master of this rank's run of entries plus the replicated
      # non-kernel entries; the momentum exists for the own run only (`_af_base` hands the kernel base pointers
      # shifted by -lo, its leaf table addresses the flat index space)
      if self.sharded and config.get("fsdp_shard_params", True) and not store.master_sharded:
        store.refresh_shadow()
        store.shard_master_(self.lo, self.hi, 0, self.comm, bounds=self.bounds)
      return
    else:
      raise NotImplementedError(f"optax_name={self.name!r}: scale_by_adam and big_vision.scale_by_adafactor "
                                "are on the fused path")
    self.clip_norm = float(config.get("grad_clip_norm") or 0.0)
    _refuse_per_example_clip(config)
    self.lr = float(config["lr"])
    # ---- per-entry hyper-parameter table + chunk map
    n_tr = store.trainable_count
    seg_rows, chunk_seg = [], np.zeros(n_tr // 1024, np.int32)
    for e in store.entries.values():
      if e.name in store.frozen:
        continue
      lv = [store.ext_of[leaf] for leaf, _ in e.flax_leaves()]   # names as presented (stacked if scanned)
      hp = {(lr_mult[l], wd[l], sched_idx_of_leaf[l]) for l in lv}
      if len(hp) != 1:
        raise NotImplementedError(f"fused tensor {e.name} has mixed optimizer hyper-parameters {hp}")
      (lm, w, si), = hp
      seg_rows.append((self.lr * lm, w, si))
      c0, c1 = e.offset // 1024, (e.offset + e.numel + 1023) // 1024
      chunk_seg[c0:c1] = len(seg_rows) - 1
    segs = np.zeros((len(seg_rows), 4), np.float32)
    for i, (le, w, si) in enumerate(seg_rows):
      segs[i, 0], segs[i, 1] = le, w
      segs[i, 2:3].view(np.int32)[0] = si
    self.segs = torch.from_numpy(segs).to(dev)
    self.chunk_seg = torch.from_numpy(chunk_seg).to(dev)
    if self.sharded:
      from big_vision_amd import dp
      self.comm = self.comm or dp.Comm()
      N, r = self.comm.size, self.comm.rank
      self.S = (n_tr + N * 1024 - 1) // (N * 1024) * 1024           # slice length, a whole number of 1024-chunks
      self.lo = min(n_tr, r * self.S)
      self.hi = min(n_tr, self.lo + self.S)
      self.bounds = [min(n_tr, i * self.S) for i in range(N + 1)]
      n_own = self.S
    else:
      self.lo, self.hi, n_own = 0, n_tr, n_tr
    self.mu = torch.zeros(n_own, device=dev, dtype=mu_dtype)
    self.nu = torch.zeros(n_own, device=dev, dtype=torch.float32)
    # "fsdp": the PARAMETERS are sharded too (reference sharding.py:104-139) - the store keeps this rank's slice of the
    # fp32 master plus the replicated entries from here on (ParamStore.shard_master_); config.fsdp_shard_params = False
    # keeps the round-4 form (state and update sharded, fp32 master replicated)
    if self.sharded and config.get("fsdp_shard_params", True) and not store.master_sharded:
      store.refresh_shadow()
      store.shard_master_(self.lo, self.hi, self.S, self.comm)
    self.count = 0
    self.gsq = torch.zeros(1, device=dev, dtype=torch.float64)
    self.stats = torch.zeros(2, device=dev, dtype=torch.float64)
    self._frozen_sq = None

  # ------------------------------------------------------------------ Adafactor --
  def _init_adafactor(self, okw, lr_mult, wd, sched_idx_of_leaf):
    """BigVision Adafactor (optax.py:187-216) per Flax leaf: which two axes optax factors
    (optax/_src/factorized.py `_factored_dims`: the two largest, if the second largest is >=
    min_dim_size_to_factor), the leaf as a strided [B1][B2][R][C] view of the flat buffers, and
    the layout of the second-moment state."""
    import ctypes
    st, dev = self.store, self.store.device
    self.af = dict(min_dim=int(okw.get("min_dim_size_to_factor", 32)), decay_rate=float(okw.get("decay_rate", 0.8)),
                   decay_offset=int(okw.get("decay_offset", 0)), beta2_cap=float(okw.get("beta2_cap", 0.999)),
                   momentum=float(okw.get("momentum", 0.9) or 0.0), eps=float(okw.get("eps", 1e-30)))
    # scale_by_adafactor(clipping_threshold=...) = optax.clip_by_block_rms on every leaf between the factored RMS
    # scaling and the momentum (optax.py:190,208): one more launch of the batched step
    self.af["block_rms_clip"] = float(okw.get("clipping_threshold") or 0.0)
    mdt = okw.get("dtype_momentum", "bfloat16")
    mom_dtype = torch.float32 if str(mdt) in ("float32", "torch.float32") else torch.bfloat16
    self.af_leaves = []
    off_state = 0
    # "fsdp" placement: rank r owns a run of whole store entries (the factored statistics are per tensor, so the
    # flat buffer is cut at tensor boundaries: entry e belongs to the rank whose equal share of the trainable
    # prefix its first element falls into); bounds[r] .. bounds[r + 1] is that run as a flat range
    if self.sharded:
      from big_vision_amd import dp
      self.comm = self.comm or dp.Comm()
      N, n_tr = self.comm.size, st.trainable_count
      share = (n_tr + N - 1) // N
      starts = sorted(e.offset for e in st.entries.values() if e.name not in st.frozen)
      self.bounds = [0] + [next((o for o in starts if o >= r * share), n_tr) for r in range(1, N)] + [n_tr]
      self.lo, self.hi = self.bounds[self.comm.rank], self.bounds[self.comm.rank + 1]
    for leaf, (sname, sl) in st.leaf_index.items():
      if sname in st.frozen:
        continue
      e = st.entries[sname]
      t = torch.empty(e.shape, device="meta")
      v = t if sl is None else t.select(sl[0], sl[1])
      shape, strides = tuple(v.shape), tuple(v.stride())
      fd = factored_dims(shape, self.af["min_dim"])
      # scan=True models present their blocks as ONE stacked Flax leaf [depth, ...] and optax factors THAT
      # leaf (optax/_src/factorized.py:_factored_dims on the stacked shape).  Per-block statistics are the
      # same arithmetic as long as the depth axis is not one of the two factored axes, i.e. depth is not
      # among the two largest extents >= min_dim_size_to_factor (B/16, L/16: depth 12 / 24 < 32).  Where it
      # would be (g/14 depth 40, G/14 depth 48: [depth, D] biases factor over (depth, D)) the update rule
      # would silently differ from the reference's - refuse.
      group = st.ext_index[st.ext_of[leaf]]
      if len(group) > 1:
        fd_st = factored_dims((len(group),) + shape, self.af["min_dim"])
        fd_shift = None if fd_st is None else tuple(a - 1 for a in fd_st)
        if fd_shift != fd:
          raise NotImplementedError(
              f"{st.ext_of[leaf]}: stacked (scan) leaf {(len(group),) + shape} factors over axes {fd_st} in optax "
              f"(the depth axis {len(group)} >= min_dim_size_to_factor enters the factored pair); per-block "
              "Adafactor statistics would differ from big_vision's - not implemented")
      if fd is None:
        d1 = d0 = None
        rest = list(range(len(shape)))
        R = C = 1
        sR = sC = 0
      else:
        d1, d0 = fd
        rest = [a for a in range(len(shape)) if a not in fd]
        R, C, sR, sC = shape[d1], shape[d0], strides[d1], strides[d0]
      rest = [a for a in rest if shape[a] > 1]
      if fd is None:
        # unfactored: enumerate the elements through (up to) four axes, last one as "C"
        axes = rest[-4:] if len(rest) <= 4 else None
        if axes is None:
          raise NotImplementedError(f"{leaf}: more than 4 non-trivial axes")
        ext = [1] * (4 - len(axes)) + [shape[a] for a in axes]
        strd = [0] * (4 - len(axes)) + [strides[a] for a in axes]
        B1, B2, R, C = ext
        sB1, sB2, sR, sC = strd
      else:
        if len(rest) > 2:
          raise NotImplementedError(f"{leaf} {shape}: more than two axes besides the factored pair")
        ext = [1] * (2 - len(rest)) + [shape[a] for a in rest]
        strd = [0] * (2 - len(rest)) + [strides[a] for a in rest]
        (B1, B2), (sB1, sB2) = ext, strd
      B = B1 * B2
      n_state = (B * R + B * C + B) if fd is not None else B * R * C
      view = (ctypes.c_long * 9)(e.offset + v.storage_offset(), B1, B2, R, C, sB1, sB2, sR, sC)
      extn = st.ext_of[leaf]
      self.af_leaves.append(dict(leaf=leaf, view=view, factored=fd is not None, dims=fd, shape=shape, rest=rest,
                                 soff=off_state, n_state=n_state, lr_eff=self.lr * lr_mult[extn], wd=wd[extn],
                                 sched=sched_idx_of_leaf[extn], B=B, R=R, C=C,
                                 own=(not self.sharded) or (self.lo <= e.offset < self.hi)))
      off_state += (n_state + 3) // 4 * 4
    self.af_state = torch.zeros(max(4, off_state), device=dev, dtype=torch.float32)
    # device table of all leaves (struct bv_af_leaf, include/bvhip.h) for the batched step: four launches per
    # step instead of up to four per leaf
    AF_LEAF = np.dtype([("off", np.int64), ("sB1", np.int64), ("sB2", np.int64), ("sR", np.int64), ("sC", np.int64),
                        ("soff", np.int64), ("B1", np.int32), ("B2", np.int32), ("R", np.int32), ("C", np.int32),
                        ("factored", np.int32), ("sched_idx", np.int32), ("r_fast", np.int32), ("pad_", np.int32),
                        ("lr_eff", np.float32), ("wd", np.float32)], align=True)
    assert AF_LEAF.itemsize == 88, AF_LEAF.itemsize
    own = [lf for lf in self.af_leaves if lf["own"]]   # the leaves this rank updates (all of them when replicated)
    if self.sharded:
      # the own-run buffers (fp32 master, momentum) reach the kernel as base pointers shifted by -lo (_adafactor_step):
      # every row of this rank's table must address elements of [lo, hi) only
      for lf in own:
        e = st.entries[st.leaf_index[lf["leaf"]][0]]
        assert self.lo <= e.offset and e.offset + e.numel <= self.hi, (lf["leaf"], e.offset, e.numel, self.lo, self.hi)
    self.af_nown = len(own)
    # bv_adafactor_step launches a 2-D grid (extent of the LARGEST leaf of the table) x (leaves), and a workgroup beyond
    # its own leaf's extent returns at once: with one table for the whole model the 300 biases / LayerNorm scales would
    # each pay for the embedding table's 32 000 rows (~10 M empty workgroups per step at B/16 + text).  The table is
    # therefore sorted by extent and cut into SIZE CLASSES (a new class where the row count or the element count
    # drops below a quarter of the class's largest; unfactored leaves apart, they skip the three statistics launches);
    # one bv_adafactor_step call per class, sized for that class.  The update does not depend on the order.
    def extents(lf):
      B = lf["B"]
      return (B * lf["R"], B * lf["C"], B, B * lf["R"] * lf["C"])
    own.sort(key=lambda lf: (not lf["factored"], -extents(lf)[3], -extents(lf)[0]))
    tab = np.zeros(max(1, len(own)), AF_LEAF)
    self.af_classes = []     # (first row of the table, rows, {rows, cols, b, total} grid extents)
    for i, lf in enumerate(own):
      off, B1, B2, R, C, sB1, sB2, sR, sC = (int(x) for x in lf["view"])
      tab[i] = (off, sB1, sB2, sR, sC, lf["soff"], B1, B2, R, C, int(lf["factored"]), lf["sched"], int(sR < sC), 0,
                lf["lr_eff"], lf["wd"])
      rows, cols, b, total = extents(lf)
      cur = self.af_classes[-1] if self.af_classes else None
      if (cur is None or cur["factored"] != lf["factored"] or total * 4 < cur["total"]
          or (lf["factored"] and rows * 4 < cur["rows"])):
        cur = dict(first=i, n=0, factored=lf["factored"], rows=0, cols=0, b=0, total=0)
        self.af_classes.append(cur)
      cur["n"] += 1
      if lf["factored"]:
        cur["rows"], cur["cols"], cur["b"] = max(cur["rows"], rows), max(cur["cols"], cols), max(cur["b"], b)
      cur["total"] = max(cur["total"], total)
    self.af_table = torch.from_numpy(tab.view(np.uint8).copy()).to(dev).view(-1, AF_LEAF.itemsize)
    # (sharded: the momentum of the OWN run of tensors only - elements [lo, hi) of the flat index space)
    n_mu = max(4, self.hi - self.lo) if self.sharded else st.trainable_count
    self.mu = torch.zeros(n_mu, device=dev, dtype=mom_dtype) if self.af["momentum"] > 0 else None
    self.nu = None
    self.count = 0
    self.gsq = torch.zeros(1, device=dev, dtype=torch.float64)
    self.stats = torch.zeros(2, device=dev, dtype=torch.float64)
    self._frozen_sq = None

  def _adafactor_step(self):
    """One fused Adafactor step.  "fsdp" placement: the trainer summed every gradient range onto the rank that owns
    it (grad_sync(), ranges cut at tensor boundaries); this rank updates its own tensors (its rows of the leaf
    table), the ranks exchange the updated fp32 ranges in place and cast what they do not own into the bf16 shadow -
    the same protocol as the sharded Adam step."""
    st, af, k = self.store, self.af, self.count
    sched = [fn(k) for fn in self.schedule_fns]
    t = float(k - af["decay_offset"]) + 1.0
    decay = min(af["beta2_cap"], 1.0 - t ** (-af["decay_rate"]))     # optax.py:196-199
    self._owner_sums_ready()
    self.gsq.zero_()
    lo, hi = (self.lo, self.hi) if self.sharded else (0, st.trainable_count)
    if hi > lo:
      ops.sqnorm_(st.grad[lo:hi], self.gsq)
    if self.sharded:
      self.comm.all_reduce_scalars_(self.gsq)
    self.stats.zero_()
    # sharded: this rank's leaves live in [lo, hi) of the flat index space; the buffers that exist for that run only
    # (fp32 master when the parameters are sharded, momentum) go in as base pointers shifted by -lo
    master = ops.ShiftedBase(st.master_own, lo) if st.master_sharded else st.master
    mu = ops.ShiftedBase(self.mu, lo) if (self.sharded and self.mu is not None) else self.mu
    for c in self.af_classes:     # one call per size class of the leaf table (four launches each, three for unfactored)
      ops.adafactor_step_(master, st.grad, mu, st.shadow, self.af_table[c["first"]:c["first"] + c["n"]], c["n"],
                          c["rows"], c["cols"], c["b"], c["total"], self.af_state, self.gsq, self.clip_norm, decay,
                          af["eps"], af["momentum"], sched, self.stats, block_rms_clip=af["block_rms_clip"],
                          block_usq=self._af_usq(c["n"]) if af["block_rms_clip"] > 0 else None)
    if self.sharded:
      comm, n_tr = self.comm, st.trainable_count
      comm.all_reduce_scalars_(self.stats)
      if st.master_sharded:
        # sharded PARAMETERS: every rank needs the bf16 compute copy of the others' tensors (the kernel wrote the own
        # run's) and the fp32 of the replicated entries - half the bytes of the fp32 exchange below
        comm.broadcast_ranges_(st.shadow[:n_tr], self.bounds)
        st.exchange_small_()
      else:
        comm.broadcast_ranges_(st.master[:n_tr], self.bounds)
        if comm.active:
          for a, b in ((0, lo), (hi, n_tr)):
            if b > a:
              ops.cast_bf16(st.master[a:b], st.shadow[a:b])
    self.count = k + 1
    st.shadow_version += 1
    return {"l2_grads": torch.sqrt(self.gsq[0]),
            "l2_params": torch.sqrt(self.stats[0] + self.frozen_sqnorm()[0]),
            "l2_updates": torch.sqrt(self.stats[1])}

  def _af_usq(self, n):
    """Per-leaf sum-of-squares scratch of clip_by_block_rms (float64, the largest size class)."""
    buf = getattr(self, "_af_usq_buf", None)
    if buf is None or buf.numel() < n:
      buf = self._af_usq_buf = torch.zeros(max(n, 1), device=self.store.device, dtype=torch.float64)
    return buf

  def _gather_af_state(self):
    """"fsdp" placement, before the state is read as a whole (checkpoint): every rank's statistics and momentum of
    the tensors it owns, broadcast in place.  A COLLECTIVE: every rank must enter state_tree()."""
    if not (self.sharded and self.comm.active):
      return
    sb = [0] * (self.comm.size + 1)      # af_state is laid out in leaf order = flat order: owners hold contiguous runs
    for lf in self.af_leaves:
      off = int(lf["view"][0])
      r = max(i for i in range(self.comm.size) if self.bounds[i] <= off)
      sb[r + 1] = max(sb[r + 1], lf["soff"] + (lf["n_state"] + 3) // 4 * 4)
    for r in range(1, len(sb)):
      sb[r] = max(sb[r], sb[r - 1])
    self.comm.broadcast_ranges_(self.af_state, sb)

  def _full_mu(self):
    """The momentum over the whole flat index space (what a replicated optimizer holds): under the "fsdp" placement a
    TEMPORARY tensor assembled from the owners' runs - a COLLECTIVE on N > 1 ranks, like _gather_af_state."""
    if self.mu is None or not self.sharded:
      return self.mu
    n_tr = self.store.trainable_count
    full = torch.zeros(n_tr, device=self.mu.device, dtype=self.mu.dtype)
    full[self.lo:self.hi] = self.mu[:self.hi - self.lo]
    self.comm.broadcast_ranges_(full, self.bounds)
    return full

  def _set_own_mu(self, full):
    """Inverse of _full_mu: keep the own run of a whole momentum buffer."""
    self.mu.zero_()
    self.mu[:self.hi - self.lo].copy_(full[self.lo:self.hi].to(self.mu.dtype))

  def adafactor_state_numel(self):
    """Elements of the optax FactoredState (count, v_row, v_col, v) this optimizer stands for - the
    number optax_test.py:320-337 checks (2 * 1024 + 2 for one 1024 x 1024 kernel)."""
    n = 1
    for lf in self.af_leaves:
      n += (lf["B"] * lf["R"] + lf["B"] * lf["C"] + 1) if lf["factored"] else (1 + 1 + lf["B"] * lf["R"] * lf["C"])
    return n

  def adafactor_state_of(self, leaf):
    """(v_row, v_col, v) of a storage leaf in the axis order optax keeps them (the leaf's shape without
    d0 / without d1 / the full shape; (1,) zeros where optax keeps a placeholder)."""
    lf = next(l for l in self.af_leaves if l["leaf"] == leaf)
    buf = self.af_state[lf["soff"]:lf["soff"] + lf["n_state"]]
    one = torch.zeros(1, device=buf.device)
    shape, rest = lf["shape"], lf["rest"]
    if not lf["factored"]:
      full = [shape[a] for a in rest[-4:]]
      return one, one, buf.view(full if full else (1,)).reshape(shape)
    d1, d0 = lf["dims"]
    B, R, C = lf["B"], lf["R"], lf["C"]

    def to_axes(t, axes_canon):    # t: canonical [rest..., X] -> leaf axis order (axes sorted), size-1 axes re-inserted
      order = sorted(range(len(axes_canon)), key=lambda i: axes_canon[i])
      t = t.permute(order)
      full = [shape[a] if a in axes_canon else 1 for a in range(len(shape)) if a in axes_canon or shape[a] == 1]
      return t.reshape(full)
    v_row = buf[:B * R].view([shape[a] for a in rest] + [R])
    v_col = buf[B * R:B * R + B * C].view([shape[a] for a in rest] + [C])
    keep_row = [a for a in range(len(shape)) if a != d0]
    keep_col = [a for a in range(len(shape)) if a != d1]
    v_row = to_axes(v_row, rest + [d1]).reshape([shape[a] for a in keep_row])
    v_col = to_axes(v_col, rest + [d0]).reshape([shape[a] for a in keep_col])
    return v_row, v_col, one

  def frozen_sqnorm(self):
    if self._frozen_sq is None:
      acc = torch.zeros(1, device=self.store.device, dtype=torch.float64)
      st = self.store
      # (sharded parameters: the frozen tensors are the replicated tail of master_small; its padding is zero)
      tail = st.master_small[st.small_trainable:] if st.master_sharded else st.master[st.trainable_count:]
      if tail.numel():
        ops.sqnorm_(tail, acc)
      self._frozen_sq = acc
    return self._frozen_sq

  def step(self):
    """tx.update + optax.apply_updates on the store; returns device scalars
    (l2_grads, l2_params, l2_updates) without synchronising."""
    if self.name in ADAFACTOR_NAMES:
      return self._adafactor_step()
    if self.sharded:
      return self._sharded_adam_step()
    st = self.store
    n_tr = st.trainable_count
    k = self.count
    sched = [fn(k) for fn in self.schedule_fns]   # scale_by_schedule uses the pre-increment count
    self.gsq.zero_()
    ops.sqnorm_(st.grad, self.gsq)
    self.stats.zero_()
    ops.adam_step_(st.master, st.grad, self.mu, self.nu, st.shadow, self.segs, self.chunk_seg, n_tr,
                   sched, self.gsq, self.clip_norm, self.b1, self.b2, self.eps,
                   1.0 - self.b1 ** (k + 1), 1.0 - self.b2 ** (k + 1), self.stats)
    self.count = k + 1
    st.shadow_version += 1     # the kernel refreshed the bf16 shadow of the trainable prefix
    return {"l2_grads": torch.sqrt(self.gsq[0]),
            "l2_params": torch.sqrt(self.stats[0] + self.frozen_sqnorm()[0]),
            "l2_updates": torch.sqrt(self.stats[1])}

  def grad_sync(self):
    """The gradient reduction object a trainer drives during the backward of a sharded step (dp.GradShardSync:
    every final range is summed onto the rank that owns it); None on one rank.  Its finish() stamps this optimizer
    (`_reduced_for` = the step count the reduced gradients belong to): the sharded step() checks the stamp."""
    from big_vision_amd import dp
    if not self.sharded or self.comm is None or not self.comm.active:
      return None
    return dp.GradShardSync(self.comm, self.store.grad, self.bounds, on_finish=self._mark_reduced)

  def mark_grads_reduced(self):
    """Public stamp: "`store.grad` already holds, on every owner, the SUM over ranks of its ranges" - good for the
    next step() only.  Callers that reduced the gradients by other means than grad_sync() (an all-reduce of the
    whole buffer, a second step() on gradients a previous step() already reduced) must call it, otherwise step()
    sums the owners' ranges over the ranks once more.  Every rank must agree on whether it stamps: the fallback
    reduction of step() is a collective."""
    self._reduced_for = self.count

  _mark_reduced = mark_grads_reduced

  def _owner_sums_ready(self):
    """Contract of step() under shard=True (advisor r4, r5): step() CONSUMES partial sums - `store.grad` holds this
    rank's PARTIAL sums and every range must have been summed onto its owner - normally by the trainer's grad_sync() object during the backward.  A
    caller that never drove one (its finish() leaves the stamp) gets the whole trainable range reduced here, after
    the backward, instead of an update from partial gradients that nothing would flag."""
    if not (self.sharded and self.comm is not None and self.comm.active):
      return
    if getattr(self, "_reduced_for", None) != self.count:
      self.grad_sync().finish()
    self._reduced_for = None      # the stamp is good for ONE step

  def _sharded_adam_step(self):
    """"fsdp" placement.  The trainer has summed every gradient range onto its OWNER during the backward
    (grad_sync() / dp.GradShardSync, overlapped with the remaining GEMMs like the all-reduce of the replicated
    path): st.grad[lo:hi] is this rank's slice of the global gradient, in place.  Then: global clip norm from the
    slices' square norms -> the fused Adam kernel on the slice (same kernel, same per-chunk hyper-parameter table,
    offset pointers; it also refreshes the bf16 shadow of the slice) -> in-place exchange of the updated fp32
    slices -> bf16 shadow of the slices this rank does NOT own (two casts: the frozen tail and the own slice are
    left alone, static_version does not move, so transposed images of frozen towers are not rebuilt - advisor r3).
    Per step and rank this moves (N-1)/N x 4 B x P each way - the bytes of the all-reduce it replaces - runs 1/N
    of the optimizer kernel and holds 1/N of its state; no staging copies."""
    st, comm = self.store, self.comm
    n_tr, S, lo, hi = st.trainable_count, self.S, self.lo, self.hi
    n_own = hi - lo
    k = self.count
    sched = [fn(k) for fn in self.schedule_fns]
    self._owner_sums_ready()
    self.gsq.zero_()
    if n_own:
      ops.sqnorm_(st.grad[lo:hi], self.gsq)
    comm.all_reduce_scalars_(self.gsq)
    self.stats.zero_()
    master_slice = st.master_own[:n_own] if st.master_sharded else st.master[lo:hi]
    if n_own:
      ops.adam_step_(master_slice, st.grad[lo:hi], self.mu[:n_own], self.nu[:n_own], st.shadow[lo:hi], self.segs,
                     self.chunk_seg[lo // 1024:], n_own, sched, self.gsq, self.clip_norm, self.b1, self.b2, self.eps,
                     1.0 - self.b1 ** (k + 1), 1.0 - self.b2 ** (k + 1), self.stats)
    comm.all_reduce_scalars_(self.stats)
    self.count = k + 1
    if st.master_sharded:
      # sharded PARAMETERS: nobody holds the other ranks' fp32 kernels.  What every rank needs after the update is the
      # bf16 compute copy - all-gathered in place (the Adam kernel wrote this rank's slice of it): HALF the bytes of
      # the fp32 exchange below - and the fp32 values of the replicated entries (biases, LayerNorm, embeddings, t, b)
      comm.all_gather_flat_(st.shadow[:n_tr], lo, hi, S)
      st.exchange_small_()
    else:
      comm.broadcast_ranges_(st.master[:n_tr], self.bounds)   # every rank's updated slice into every rank's master
      if comm.active:
        for a, b in ((0, lo), (hi, n_tr)):
          if b > a:
            ops.cast_bf16(st.master[a:b], st.shadow[a:b])
    st.shadow_version += 1     # the trainable prefix changed; frozen tensors (static_version) did not
    return {"l2_grads": torch.sqrt(self.gsq[0]),
            "l2_params": torch.sqrt(self.stats[0] + self.frozen_sqnorm()[0]),
            "l2_updates": torch.sqrt(self.stats[1])}

  def _full_moment(self, t):
    """Sharded moments as the full flat buffer every rank would hold when replicated (checkpointing).  A
    COLLECTIVE on N > 1 ranks: `state_tree()` / `u.save_train_state` must be entered by every rank (write the
    file on one)."""
    if not self.sharded:
      return t
    n_tr = self.store.trainable_count
    full = torch.zeros(n_tr, device=t.device, dtype=t.dtype)
    full[self.lo:self.hi] = t[:self.hi - self.lo]
    self.comm.all_gather_flat_(full, self.lo, self.hi, self.S)
    return full

  # ---------------------------------------------------------------- checkpointing --
  def _chain_layout(self):
    """Positions inside the reference's optax.chain (optax.py:143-149) whose state is not empty:
    index of masked(optimizer) and of every masked(scale_by_schedule).  chain = [clip | identity,
    masked(opt), scale(lr), *lr_mults, *weight_decay, *schedules, set_to_zero, scale(-1)]."""
    cfg = self._cfg
    n_lr = 1 + (len(cfg["lr_mults"]) if cfg.get("lr_mults") else 0)
    n_wd = len(cfg.get("wd_mults", [(".*/kernel$", 1.0)])) if cfg.get("wd") else 0
    first_sched = 2 + n_lr + n_wd
    return 1, [first_sched + i for i in range(len(self.schedule_fns))]

  def _moment_tree(self, flat):
    st = self.store
    names = [n for n in st.leaf_names() if not any(e in st.frozen for e in st.entries_of(n))]
    views = {}
    for n in names:
      group = st.ext_index[n]
      per = []
      for leaf in group:
        sname, sl = st.leaf_index[leaf]
        e = st.entries[sname]
        t = flat[e.offset:e.offset + e.numel].view(e.shape)
        per.append(t if sl is None else t.select(sl[0], sl[1]))
      views[n] = per[0] if (len(per) == 1 and group[0] == n) else torch.stack(per)
    return u.recover_tree(list(views.keys()), list(views.values()))

  def state_tree(self):
    """The optimizer state with the names `u.tree_flatten_with_names` gives the reference's
    optax state (tuples are indexed, utils.py:616-641): `<i>/0/0` = count, `<i>/0/1/<leaf>` = mu,
    `<i>/0/2/<leaf>` = nu of masked(scale_by_adam) at chain position i (MaskedState.inner_state ->
    ScaleByAdamState(count, mu, nu); frozen leaves are MaskedNode()s and emit nothing), and
    `<j>/0/0` = count of every masked(scale_by_schedule).  Adafactor (scale_by_factored_rms +
    ema, optax.py:187-216): `<i>/0/0/{0: count, 1: v_row, 2: v_col, 3: v}` and `<i>/0/2/{0: count,
    1: ema}` (_af_state_tree).  Tensors are copies (stacked for scan-layout leaves)."""
    i_opt, i_sched = self._chain_layout()
    cnt = np.asarray(self.count, np.int32)
    tree = {str(j): {"0": {"0": cnt}} for j in i_sched}
    tree[str(i_opt)] = {"0": self._opt_state_tree(cnt)}
    return tree

  def _opt_state_tree(self, cnt):
    if self.name in ADAFACTOR_NAMES:
      self._gather_af_state()
      return self._af_state_tree(cnt)
    return {"0": cnt, "1": self._moment_tree(self._full_moment(self.mu)), "2": self._moment_tree(self._full_moment(self.nu))}

  def _trainable_ext_names(self):
    st = self.store
    return [n for n in st.leaf_names() if not any(e in st.frozen for e in st.entries_of(n))]

  def _af_state_tree(self, cnt):
    """State of masked(chain(scale_by_factored_rms, identity | clip, ema)) (optax.py:187-216) in the
    reference's naming: inner_state = (FactoredState(count, v_row, v_col, v), EmptyState(), EmaState(count,
    ema)) -> `0/{0: count, 1: v_row, 2: v_col, 3: v}` and `2/{0: count, 1: ema}` (no `2` when momentum is
    off: optax.identity has an empty state).  optax keeps a zeros((1,)) placeholder in v_row / v_col of an
    unfactored leaf and in v of a factored one; scan-layout leaves are stacked over depth (placeholders are
    not: optax builds them from the stacked leaf)."""
    st = self.store
    trees = ({}, {}, {})
    for n in self._trainable_ext_names():
      group = st.ext_index[n]
      stacked = not (len(group) == 1 and group[0] == n)
      per = [self.adafactor_state_of(leaf) for leaf in group]
      factored = next(l for l in self.af_leaves if l["leaf"] == group[0])["factored"]
      for k in range(3):
        real = (k < 2) == factored
        ts = [p[k].detach().clone() for p in per]
        trees[k][n] = (torch.stack(ts) if (stacked and real) else ts[0])
    rt = lambda d: u.recover_tree(list(d.keys()), list(d.values()))
    out = {"0": {"0": cnt, "1": rt(trees[0]), "2": rt(trees[1]), "3": rt(trees[2])}}
    if self.mu is not None:
      out["2"] = {"0": cnt, "1": self._moment_tree(self._full_mu())}
    return out

  def _af_assign(self, leaf, v_row, v_col, v):
    """Inverse of adafactor_state_of: optax-ordered (v_row, v_col, v) of one storage leaf into af_state."""
    lf = next(l for l in self.af_leaves if l["leaf"] == leaf)
    buf = self.af_state[lf["soff"]:lf["soff"] + lf["n_state"]]
    shape = lf["shape"]
    if not lf["factored"]:
      if tuple(v.shape) != tuple(shape):
        raise ValueError(f"Shape mismatch for Adafactor v of {leaf}: {tuple(v.shape)} vs {tuple(shape)}")
      # canonical order = the (up to four) non-trivial axes in leaf order: a plain flatten
      buf.copy_(v.to(buf.dtype).to(buf.device).reshape(-1))
      return
    d1, d0 = lf["dims"]
    B, R, C = lf["B"], lf["R"], lf["C"]
    others = [a for a in range(len(shape)) if a not in (d1, d0)]
    for t, drop, keep_last, lo, n in ((v_row, d0, d1, 0, B * R), (v_col, d1, d0, B * R, B * C)):
      keep = [a for a in range(len(shape)) if a != drop]
      want = tuple(shape[a] for a in keep)
      if tuple(t.shape) != want:
        raise ValueError(f"Shape mismatch for Adafactor statistics of {leaf}: {tuple(t.shape)} vs {want}")
      perm = [keep.index(a) for a in others + [keep_last]]
      buf[lo:lo + n].copy_(t.to(buf.dtype).to(buf.device).permute(perm).reshape(-1))

  def load_state_tree(self, tree):
    """Inverse of `state_tree` (accepts the flat `{name: array}` form too)."""
    flat = dict(u.tree_flatten_with_names(tree)[0])   # flat '/'-joined keys pass through unchanged
    i_opt, _ = self._chain_layout()
    self._load_opt_state(flat, f"{i_opt}/0/")

  def _assign_moment(self, flat_buf, flat, prefix):
    st = self.store
    for n in st.leaf_names():
      if any(e in st.frozen for e in st.entries_of(n)):
        continue
      key = prefix + n
      if key not in flat:
        raise ValueError(f"optimizer state is missing '{key}'")
      v = torch.as_tensor(np.asarray(flat[key], np.float32) if not torch.is_tensor(flat[key]) else flat[key])
      group = st.ext_index[n]
      stacked = not (len(group) == 1 and group[0] == n)
      for k, leaf in enumerate(group):
        sname, sl = st.leaf_index[leaf]
        e = st.entries[sname]
        t = flat_buf[e.offset:e.offset + e.numel].view(e.shape)
        dst = t if sl is None else t.select(sl[0], sl[1])
        src = v[k] if stacked else v
        if tuple(src.shape) != tuple(dst.shape):
          raise ValueError(f"Shape mismatch for optimizer state {key}: {tuple(src.shape)} vs {tuple(dst.shape)}")
        dst.copy_(src.to(dst.dtype).to(dst.device))

  def _load_opt_state(self, flat, pre):
    if self.name in ADAFACTOR_NAMES:
      return self._load_af_state(flat, pre)
    self.count = int(np.asarray(flat[pre + "0"]))
    if self.sharded:   # the checkpoint holds whole moments: lay them out in a full buffer, keep the own slice
      n_tr = self.store.trainable_count
      for buf, key in ((self.mu, "1/"), (self.nu, "2/")):
        full = torch.zeros(n_tr, device=buf.device, dtype=buf.dtype)
        self._assign_moment(full, flat, pre + key)
        buf.zero_()
        buf[:self.hi - self.lo] = full[self.lo:self.hi]
      return
    self._assign_moment(self.mu, flat, pre + "1/")
    self._assign_moment(self.nu, flat, pre + "2/")

  def _load_af_state(self, flat, pre):
    st = self.store
    as_t = lambda x: x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x, np.float32))
    self.count = int(np.asarray(flat[pre + "0/0"]))
    for n in self._trainable_ext_names():
      keys = [f"{pre}0/{k}/{n}" for k in (1, 2, 3)]
      missing = [k for k in keys if k not in flat]
      if missing:
        raise ValueError(f"optimizer state is missing '{missing[0]}'")
      vr, vc, vv = (as_t(flat[k]) for k in keys)
      group = st.ext_index[n]
      stacked = not (len(group) == 1 and group[0] == n)
      factored = next(l for l in self.af_leaves if l["leaf"] == group[0])["factored"]
      for i, leaf in enumerate(group):
        pick = lambda t, real: t[i] if (stacked and real) else t
        self._af_assign(leaf, pick(vr, factored), pick(vc, factored), pick(vv, not factored))
    if self.mu is not None:
      if self.sharded:      # the checkpoint holds the whole momentum: lay it out in a full buffer, keep the own run
        full = torch.zeros(st.trainable_count, device=self.mu.device, dtype=self.mu.dtype)
        self._assign_moment(full, flat, pre + "2/1/")
        self._set_own_mu(full)
      else:
        self._assign_moment(self.mu, flat, pre + "2/1/")

  def state_dict(self):
    """The raw state buffers.  Under the "fsdp" placement this is a COLLECTIVE like state_tree() (every rank must
    call it): each rank holds the moments / statistics of what it owns only, so the owners' parts are exchanged
    first and every rank returns the WHOLE state (advisor r4: a rank-0 state_dict used to lose (N-1)/N of it);
    what load_state_dict of a replicated or sharded optimizer takes back."""
    if self.name in ADAFACTOR_NAMES:
      self._gather_af_state()
      return {"mu": self._full_mu(), "af_state": self.af_state, "count": self.count}
    # always moments of exactly `trainable_count` elements, whatever the placement (advisor r5: a sharded
    # optimizer on an inactive one-rank group used to hand out its padded own-slice buffers)
    if self.sharded:
      return {"mu": self._full_moment(self.mu), "nu": self._full_moment(self.nu), "count": self.count}
    return {"mu": self.mu, "nu": self.nu, "count": self.count}

  def load_state_dict(self, d):
    if self.name in ADAFACTOR_NAMES:
      if self.mu is not None:
        if d["mu"].numel() != self.store.trainable_count:
          raise ValueError(f"Adafactor momentum of {d['mu'].numel()} elements does not fit this model's "
                           f"{self.store.trainable_count} trainable parameters (state_dict() holds the whole momentum)")
        if self.sharded:
          self._set_own_mu(d["mu"].to(self.mu.device))
        else:
          self.mu.copy_(d["mu"].to(self.mu.dtype))
      self.af_state.copy_(d["af_state"]); self.count = int(d["count"])
      return
    mu, nu = d["mu"], d["nu"]
    n_tr = self.store.trainable_count
    if mu.numel() != n_tr or nu.numel() != n_tr:
      raise ValueError(f"optimizer moments of {mu.numel()} / {nu.numel()} elements do not fit this model's "
                       f"{n_tr} trainable parameters (state_dict() of any placement holds whole moments)")
    if self.sharded:                                       # whole moments -> own slice
      n_own = self.hi - self.lo
      self.mu.zero_(); self.nu.zero_()
      self.mu[:n_own].copy_(mu[self.lo:self.hi].to(self.mu.dtype)); self.nu[:n_own].copy_(nu[self.lo:self.hi])
    else:
      self.mu.copy_(mu.to(self.mu.dtype)); self.nu.copy_(nu)
    self.count = int(d["count"])


def replace_frozen(schedule, pytree, replacement, log=None):
  """Replaces the values of `pytree` whose parameter is frozen (a `None` entry of `schedule`) with `replacement`
  (optax.py:44-51; trainers use it for `l2_grads`: the norm of the gradients that are applied, train.py:307)."""
  del log
  if not isinstance(schedule, (list, tuple)):
    return pytree
  patterns, scheds = zip(*schedule)
  masks = u.make_mask_trees(pytree, patterns)
  flat_masks = [dict(u.tree_flatten_with_names(m)[0]) for m in masks]
  names = [n for n, _ in u.tree_flatten_with_names(pytree)[0]]
  not_covered = [n for n in names if not any(m[n] for m in flat_masks)]
  assert not not_covered, f"All params must be covered (use `None` for freezing): {not_covered}"
  frozen = {n: any(m[n] for m, s_ in zip(flat_masks, scheds) if s_ is None) for n in names}
  return u.tree_map_with_names(lambda n, v: replacement if frozen[n] else v, pytree)


def make(config, store: ParamStore, *, sched_kw, comm=None, shard=False):
  """Returns (optimizer, schedule_fns) like bv_optax.make returns (tx, sched_fns).  comm / shard: see Optimizer."""
  opt = Optimizer(config, store, sched_kw=sched_kw, comm=comm, shard=shard)
  return opt, opt.schedule_fns


def get_count(opt: Optimizer, jittable=False):
  """optax.py:30-41."""
  del jittable
  return opt.count
