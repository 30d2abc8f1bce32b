"""Flat parameter storage with Flax-compatible named views.

The reference keeps parameters as an immutable pytree of arrays whose leaf
names ('/'-joined, utils.py:616-641) are the checkpoint / regex-addressing
contract.  Here all parameters of a model live in ONE flat fp32
device buffer (plus a same-shaped grad buffer, a bf16 shadow for the MFMA
GEMMs and the Adam moments), laid out for the kernels:

  * q/k/v projection kernels are stored fused as [D, 3, H, Dh] so the QKV
    projection is a single [D, 3D] GEMM; the Flax leaves `query/kernel`,
    `key/kernel`, `value/kernel` (D,H,Dh) are strided VIEWS of that tensor
    (same for the MAP head's key/value).
  * every tensor starts at a multiple of 1024 elements so the fused optimizer
    kernel can address hyper-parameters per 1024-element chunk.

`ParamStore.tree()` returns the nested dict with exactly the reference's leaf
names and shapes; `load_tree()` accepts such a tree (e.g. from an .npz).
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

ALIGN = 1024


class Entry:
  """One storage tensor; `views` maps Flax leaf names to (dim, index) slices."""

  def __init__(self, name: str, shape: Sequence[int], init, views: Optional[Dict[str, Tuple[int, int]]] = None):
    self.name = name
    self.shape = tuple(int(s) for s in shape)
    self.init = init          # callable(gen, flax_shape) -> cpu tensor, applied per Flax leaf
    self.views = views        # None: the Flax leaf IS the storage tensor (same name)
    self.offset = -1
    self.numel = int(np.prod(self.shape))

  def flax_leaves(self):
    if self.views is None:
      return [(self.name, None)]
    return list(self.views.items())


_BLOCK = re.compile(r"^(?P<pre>.*/)encoderblock_(?P<i>\d+)/(?P<rest>.+)$")


def scan_name(name: str, scan_prefixes: Sequence[str]):
  """`<enc>/encoderblock_<i>/<rest>` -> (`<enc>/encoderblock/<rest>`, i) for the encoders
  listed in `scan_prefixes` (models built with scan=True: the reference stacks the blocks of
  nn.scan on a leading `depth` axis, vit.py:129-148,363-405); (name, None) otherwise."""
  m = _BLOCK.match(name)
  if m and m.group("pre").rstrip("/") in scan_prefixes:
    return f"{m.group('pre')}encoderblock/{m.group('rest')}", int(m.group("i"))
  return name, None


def external_leaf_names(leaves: Sequence[str], scan_prefixes: Sequence[str]) -> List[str]:
  return sorted({scan_name(n, scan_prefixes)[0] for n in leaves})


class ParamTree(dict):
  """Nested dict of parameter views that remembers the store it came from."""
  store = None
  buf = "master"
  prefix = ""   # '/'-terminated path of this node below the store root


def _nest(flat: Dict[str, torch.Tensor], store, buf) -> ParamTree:
  root = ParamTree()
  root.store, root.buf = store, buf
  for name, v in flat.items():
    node = root
    parts = name.split("/")
    for p in parts[:-1]:
      if p not in node:
        child = ParamTree()
        child.store, child.buf = store, buf
        child.prefix = f"{node.prefix}{p}/"
        dict.__setitem__(node, p, child)
      node = node[p]
    dict.__setitem__(node, parts[-1], v)
  return root


def flatten_tree(tree, prefix="") -> Dict[str, object]:
  """'/'-joined leaf names, sorted traversal (utils.py:616-641)."""
  out = {}
  if isinstance(tree, dict):
    for k in sorted(tree.keys()):
      out.update(flatten_tree(tree[k], f"{prefix}{k}/"))
    return out
  out[prefix.rstrip("/")] = tree
  return out


class ParamStore:
  def __init__(self, entries: List[Entry], device, frozen: Optional[Sequence[str]] = None,
               scan_prefixes: Sequence[str] = ()):
    """`frozen`: storage-entry names placed at the END of the flat buffers so
    the optimizer state / kernels only cover the trainable prefix.
    `scan_prefixes`: encoders (e.g. "img/Transformer") whose blocks are PRESENTED stacked,
    `<enc>/encoderblock/<leaf>` with a leading depth axis, like a reference model built with
    scan=True.  Storage and kernels are unchanged (one tensor per block); the stacked leaves
    are strided views over the same flat buffers (all blocks of an encoder have the same
    layout, so leaf i sits at offset0 + i * block_stride)."""
    self.device = torch.device(device)
    self.scan_prefixes = tuple(scan_prefixes)
    frozen = set(frozen or ())
    self.entries: Dict[str, Entry] = {}
    order = [e for e in entries if e.name not in frozen] + [e for e in entries if e.name in frozen]
    off = 0
    self.trainable_count = 0
    for e in order:
      if e.name in self.entries:
        raise ValueError(f"duplicate parameter {e.name}")
      e.offset = off
      off += (e.numel + ALIGN - 1) // ALIGN * ALIGN
      if e.name not in frozen:
        self.trainable_count = off
      self.entries[e.name] = e
    self.frozen = frozen
    self.count = off
    self.master = torch.zeros(off, device=self.device, dtype=torch.float32)
    self.shadow = torch.zeros(off, device=self.device, dtype=torch.bfloat16)
    self.grad = None  # allocated by ensure_grad()
    self.leaf_index: Dict[str, Tuple[str, Optional[Tuple[int, int]]]] = {}
    for e in order:
      for leaf, sl in e.flax_leaves():
        self.leaf_index[leaf] = (e.name, sl)
    # external (presentation) names: identical to the internal ones unless an encoder is scanned
    self.ext_of: Dict[str, str] = {}
    self.ext_index: Dict[str, List[str]] = {}
    for leaf in self.leaf_index:
      ext, i = scan_name(leaf, self.scan_prefixes)
      self.ext_of[leaf] = ext
      self.ext_index.setdefault(ext, []).append((i if i is not None else 0, leaf))
    self.ext_index = {k: [l for _, l in sorted(v)] for k, v in self.ext_index.items()}
    # parameter sharding ("fsdp" placement, shard_master_): fp32 master split into this rank's slice + the replicated entries
    self.master_sharded = False
    self.master_own = self.master_small = None
    self.small_off: Dict[str, int] = {}
    self.own = (0, self.trainable_count)
    self._shadow_dirty = True
    self.shadow_version = 0   # bumped whenever the bf16 shadow changes (cast / optimizer step)
    self.static_version = 0   # bumped only by a full cast (init / load): the version of the FROZEN tensors, which an optimizer step never touches

  # ------------------------------------------------------------ accessors --
  def _buf(self, buf):
    if buf == "grad":
      self.ensure_grad()
    return getattr(self, buf)

  def ensure_grad(self):
    if self.grad is None:
      self.grad = torch.zeros(self.trainable_count, device=self.device, dtype=torch.float32)
    return self.grad

  def t(self, name: str, buf: str = "master") -> torch.Tensor:
    """Storage tensor `name` (kernel layout) from one of the flat buffers.  With a SHARDED fp32 master
    (`shard_master_`, the "fsdp" placement) `buf="master"` resolves a replicated entry (everything kernels read in fp32,
    and every frozen tensor) to its view of `master_small`, a matmul kernel that lies inside this rank's slice to its
    view of `master_own`, and raises KeyError for a kernel another rank owns (`full_tree()` gathers)."""
    e = self.entries[name]
    if buf == "master" and self.master_sharded:
      if name in self.small_off:
        o = self.small_off[name]
        return self.master_small[o:o + e.numel].view(e.shape)
      lo, hi = self.own
      if lo <= e.offset and e.offset + e.numel <= hi:
        return self.master_own[e.offset - lo:e.offset - lo + e.numel].view(e.shape)
      raise KeyError(f"{name}: the fp32 master of this kernel lives on its owner rank (sharded parameters); "
                     "ParamStore.full_tree() gathers it")
    b = self._buf(buf)
    if e.offset + e.numel > b.numel():
      raise KeyError(f"{name} is frozen: it has no entry in buffer '{buf}'")
    return b[e.offset:e.offset + e.numel].view(e.shape)

  def grad_range(self, pred) -> Optional[Tuple[int, int]]:
    """[lo, hi) of the flat gradient buffer covered by the trainable entries whose name satisfies
    `pred`, or None if they are not one contiguous run (or there are none)."""
    hit = [e for n, e in self.entries.items() if n not in self.frozen and pred(n)]
    if not hit:
      return None
    lo = min(e.offset for e in hit)
    hi = max(e.offset + (e.numel + ALIGN - 1) // ALIGN * ALIGN for e in hit)
    for n, e in self.entries.items():
      if n not in self.frozen and not pred(n) and lo <= e.offset < hi:
        return None
    return lo, min(hi, self.trainable_count)

  def has_grad(self, name: str) -> bool:
    return name not in self.frozen

  def g(self, name: str) -> Optional[torch.Tensor]:
    """Grad view of a storage tensor, or None if it is frozen."""
    return self.t(name, "grad") if name not in self.frozen else None

  def _leaf_internal(self, leaf_name: str, buf: str) -> torch.Tensor:
    sname, sl = self.leaf_index[leaf_name]
    t = self.t(sname, buf)
    return t if sl is None else t.select(sl[0], sl[1])

  def leaf(self, leaf_name: str, buf: str = "master") -> torch.Tensor:
    """View of one (external) Flax leaf in buffer `buf`; scanned encoders give the stacked
    `[depth, ...]` view.  Internal per-block names are accepted too."""
    group = self.ext_index.get(leaf_name)
    if group is None or (len(group) == 1 and group[0] == leaf_name):
      return self._leaf_internal(leaf_name, buf)
    views = [self._leaf_internal(l, buf) for l in group]
    v0 = views[0]
    if len(views) == 1:
      return v0.unsqueeze(0)
    if buf == "master" and self.master_sharded:
      return torch.stack(views)     # (a copy: the replicated entries are packed, not at their flat offsets)
    step = views[1].storage_offset() - v0.storage_offset()
    for k, v in enumerate(views):
      if (tuple(v.shape), v.stride(), v.storage_offset()) != (tuple(v0.shape), v0.stride(), v0.storage_offset() + k * step):
        raise RuntimeError(f"blocks of {leaf_name} are not laid out uniformly; cannot present them stacked")
    return torch.as_strided(self._buf(buf), (len(views),) + tuple(v0.shape), (step,) + tuple(v0.stride()),
                            v0.storage_offset())

  def leaf_names(self) -> List[str]:
    """External leaf names (what checkpoints and config regexes address)."""
    return sorted(self.ext_index.keys())

  def entries_of(self, leaf_name: str) -> List[str]:
    """Storage entries behind an external leaf (depth of them for a stacked leaf)."""
    return [self.leaf_index[l][0] for l in self.ext_index[leaf_name]]

  def tree(self, buf: str = "master") -> ParamTree:
    """Leaves of buffer `buf` as a nested ParamTree.  With a sharded fp32 master the tree stays COMPLETE and LIVE without
    holding a gathered copy: replicated leaves (biases, LayerNorm, embeddings, t, b, frozen tensors) are views of their
    fp32 values, matmul kernels are views of the replicated bf16 compute copy (what every forward reads); the fp32 kernels
    are a collective away: `full_tree()`."""
    flat = {}
    for n in self.leaf_names():
      if buf == "grad" and any(e in self.frozen for e in self.entries_of(n)):
        continue
      if buf == "master" and self.master_sharded and not all(e in self.small_off for e in self.entries_of(n)):
        flat[n] = self.leaf(n, "shadow")
        continue
      flat[n] = self.leaf(n, buf)
    return _nest(flat, self, buf)

  # ------------------------------------------------- sharded fp32 master --
  def shard_master_(self, lo: int, hi: int, S: int, comm, bounds=None):
    """"fsdp" placement (reference sharding.py:104-139: parameters AND optimizer state 1/N per device): from here on
    this rank keeps the fp32 master of its slice [lo, hi) of the flat trainable buffer only (`master_own`), plus
    `master_small`: every entry a kernel reads in fp32 (everything that is not a `.../kernel`: biases, LayerNorm,
    position / token embeddings, cls, t, b) and every frozen tensor, replicated.  The full flat master is released.
    Matmul kernels reach the GEMMs through the replicated bf16 shadow, which the sharded optimizer step all-gathers
    (optax.Optimizer._sharded_adam_step); `exchange_small_` carries the updated fp32 of the replicated entries.
    slices are S apart (a whole number of 1024-element chunks) - or, with `bounds` (rank r owns [bounds[r],
    bounds[r + 1]): the sharded Adafactor cuts the flat buffer at TENSOR boundaries, its statistics are per tensor),
    of unequal length and exchanged by one broadcast per owner instead of an all-gather."""
    assert self.master is not None and not self.master_sharded
    self.own, self.own_S, self._comm = (int(lo), int(hi)), int(S or 0), comm
    self.own_bounds = None if bounds is None else [int(b) for b in bounds]
    self.master_own = self.master[lo:hi].clone()
    small = [e for e in self.entries.values() if e.name in self.frozen or not e.name.endswith("/kernel")]
    small.sort(key=lambda e: (e.name in self.frozen, e.offset))      # trainable ones first: only they are exchanged
    off, self.small_off = 0, {}
    for e in small:
      self.small_off[e.name] = off
      off += (e.numel + ALIGN - 1) // ALIGN * ALIGN
      if e.name not in self.frozen:
        self.small_trainable = off
    if not any(e.name not in self.frozen for e in small):
      self.small_trainable = 0
    self.master_small = torch.zeros(off, device=self.device, dtype=torch.float32)
    for e in small:
      o = self.small_off[e.name]
      self.master_small[o:o + e.numel] = self.master[e.offset:e.offset + e.numel]
    # pieces of trainable replicated entries inside the own slice: (offset in master_own, offset in master_small, length)
    self._own_small = []
    for e in small:
      if e.name in self.frozen:
        continue
      a, b = max(lo, e.offset), min(hi, e.offset + e.numel)
      if b > a:
        self._own_small.append((a - lo, self.small_off[e.name] + (a - e.offset), b - a))
    self._small_tmp = None
    self.master = None
    self.master_sharded = True

  def exchange_small_(self):
    """After a sharded optimizer step: the updated fp32 values of the replicated entries - each was updated by the
    rank whose slice holds it - reach every rank's `master_small` (in place: kernel-side views stay valid).  One
    all-reduce over the trainable part of `master_small` with every rank contributing the pieces it owns."""
    comm = self._comm
    n = self.small_trainable
    if n == 0:
      return
    if comm is None or not comm.active:
      for so, do, ln in self._own_small:
        self.master_small[do:do + ln] = self.master_own[so:so + ln]
      return
    if self._small_tmp is None:
      self._small_tmp = torch.zeros(n, device=self.device, dtype=torch.float32)
    tmp = self._small_tmp
    tmp.zero_()
    for so, do, ln in self._own_small:
      tmp[do:do + ln] = self.master_own[so:so + ln]
    comm.all_reduce_sum_(tmp)
    self.master_small[:n].copy_(tmp)

  def _gather_slices_(self, flat: torch.Tensor):
    """In place on a flat tensor over the trainable prefix (any dtype): every rank's own range reaches every rank."""
    if self._comm is None:
      return
    if self.own_bounds is not None:
      self._comm.broadcast_ranges_(flat, self.own_bounds)
    else:
      self._comm.all_gather_flat_(flat, self.own[0], self.own[1], self.own_S)

  def gather_master(self) -> torch.Tensor:
    """The whole flat fp32 master as a TEMPORARY tensor.  A COLLECTIVE on N > 1 ranks (checkpointing, tests)."""
    if not self.master_sharded:
      return self.master
    lo, hi = self.own
    full = torch.zeros(self.count, device=self.device, dtype=torch.float32)
    full[lo:hi] = self.master_own
    self._gather_slices_(full[:self.trainable_count])
    for name, o in self.small_off.items():      # replicated entries (the frozen ones exist nowhere else)
      e = self.entries[name]
      full[e.offset:e.offset + e.numel] = self.master_small[o:o + e.numel]
    return full

  def full_tree(self) -> ParamTree:
    """fp32 values of EVERY leaf (a copy when the master is sharded; a collective then, see gather_master)."""
    if not self.master_sharded:
      return self.tree()
    full = self.gather_master()
    flat = {}
    for n in self.leaf_names():
      per = []
      for leaf in self.ext_index[n]:
        sname, sl = self.leaf_index[leaf]
        e = self.entries[sname]
        t = full[e.offset:e.offset + e.numel].view(e.shape)
        per.append(t if sl is None else t.select(sl[0], sl[1]))
      group = self.ext_index[n]
      flat[n] = per[0] if (len(per) == 1 and group[0] == n) else torch.stack(per)
    return _nest(flat, None, "master")

  # ---------------------------------------------------------------- I / O --
  def init_random(self, seed: int):
    """Fills the master buffer from each entry's initialiser (CPU RNG -> device)."""
    gen = torch.Generator().manual_seed(int(seed))
    for e in self.entries.values():
      for leaf, sl in e.flax_leaves():
        dst = self._leaf_internal(leaf, "master")
        val = e.init(gen, tuple(dst.shape))
        dst.copy_(val.to(torch.float32).to(self.device))
    self._shadow_dirty = True

  def load_tree(self, tree, strict: bool = True):
    if self.master_sharded:      # materialise the flat master (collective), load into it, shard again
      (lo, hi), S, comm, bounds = self.own, self.own_S, self._comm, self.own_bounds
      self.master = self.gather_master()
      self.master_sharded, self.master_own, self.master_small = False, None, None
      try:
        self.load_tree(tree, strict)
        self.refresh_shadow()
      finally:
        self.shard_master_(lo, hi, S, comm, bounds=bounds)
      return
    flat = flatten_tree(tree)
    # accept the presented layout (stacked for scanned encoders) and the per-block one
    known = set(self.ext_index) | set(self.leaf_index)
    covered = {l for n in flat if n in self.ext_index for l in self.ext_index[n]} | \
              {n for n in flat if n in self.leaf_index}
    missing = [n for n in self.leaf_index if n not in covered]
    extra = [n for n in flat if n not in known]
    if strict and (missing or extra):
      raise ValueError(f"Parameter tree mismatch. Missing: {missing[:8]} Unexpected: {extra[:8]}")
    for n, v in flat.items():
      if n not in known:
        continue
      dst = self.leaf(n)
      v = torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v)
      if tuple(v.shape) != tuple(dst.shape):
        raise ValueError(f"Shape mismatch for {n}: got {tuple(v.shape)}, expected {tuple(dst.shape)}")
      dst.copy_(v.to(torch.float32).to(self.device))
    self._shadow_dirty = True

  def refresh_shadow(self, force: bool = False):
    """bf16 shadow <- master (HIP cast kernel).  The optimizer kernel keeps the
    trainable prefix in sync itself; this is for init / load / frozen tensors."""
    if (self._shadow_dirty or force) and self.master_sharded:
      # (an in-place edit of the sharded master: own slice and replicated entries are cast, the slices gathered)
      from big_vision_amd import ops
      lo, hi = self.own
      if hi > lo:
        ops.cast_bf16(self.master_own, self.shadow[lo:hi])
      self._gather_slices_(self.shadow[:self.trainable_count])
      for name, o in self.small_off.items():
        e = self.entries[name]
        n8 = (e.numel + ALIGN - 1) // ALIGN * ALIGN
        ops.cast_bf16(self.master_small[o:o + n8], self.shadow[e.offset:e.offset + n8])
      self._shadow_dirty = False
      self.shadow_version += 1
      self.static_version += 1
      return
    if self._shadow_dirty or force:
      from big_vision_amd import ops
      ops.cast_bf16(self.master, self.shadow)
      self._shadow_dirty = False
      self.shadow_version += 1
      self.static_version += 1

  def mark_dirty(self):
    self._shadow_dirty = True

  def zero_grad(self):
    self.ensure_grad().zero_()


def adhoc_store(cache: dict, key, params, factory) -> "ParamStore":
  """Store for `model.apply({"params": tree}, ...)` with a plain (numpy / torch) tree that is not
  bound to a ParamStore.  ONE store per (model, input geometry) is kept in `cache` and re-used
  for its device allocations only: the values are re-loaded from `tree` on every call, so a
  different tree (CPython re-uses ids after GC) or an in-place edit of the same tree can never
  run with stale weights, and repeated ad-hoc applies do not accumulate device copies."""
  key = ("adhoc",) + tuple(key)
  store = cache.get(key)
  if store is None:
    store = cache[key] = factory()
  store.load_tree(params)
  return store


# ------------------------------------------------------------ regex masks ----
def make_masks(names: Sequence[str], patterns: Sequence[str]) -> List[Dict[str, bool]]:
  """First-match-wins masks with fullmatch (reference utils.py:1169-1212)."""
  comp = []
  for p in patterns:
    assert not p.startswith("/"), f"Big vision parameter names never start with '/': '{p}"
    comp.append(re.compile(p))
  masks = [dict() for _ in patterns]
  for n in names:
    hit = False
    for i, c in enumerate(comp):
      m = (not hit) and bool(c.fullmatch(n))
      masks[i][n] = m
      hit = hit or m
  return masks
