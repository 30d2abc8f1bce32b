"""Parameter placement for the data-parallel hot path: the `sharding` surface of the reference.

Reference: big_vision/sharding.py - `infer_sharding(params, strategy, mesh)` (:38-71) walks
`config.sharding_strategy = [(regex, "op|op(...)"), ...]`, each parameter is matched by the FIRST
pattern that hits it and the ops of that entry rewrite its partition spec; ops are registered as
`shardings.replicate` (:83-101), `shardings.fsdp` (:104-139), `shardings.logical_partitioning`,
`shardings.shard_dim`.  The trainers default to `[(".*", "replicate")]`
(trainers/proj/image_text/siglip.py:196) and split the batch over all devices
(utils.py:1388-1409).

North star of this repo: REPLICATED data parallelism - every rank holds the whole flat parameter
buffer (203 M parameters of B/16 + text-B = 0.8 GB fp32 of 288 GB), gradients are summed with an
RCCL all-reduce (big_vision_amd/dp.py).  Two placements exist:

  replicate  every axis unsharded on every rank (the default strategy).
  fsdp       `fsdp(axis=..., min_size_to_shard_mb=4)` (:104-139).  `infer_sharding` returns the spec the
             reference would return - the largest axis divisible by the device count of every tensor above the
             size threshold carries the mesh axis name.  WHAT IS BUILT BEHIND IT: the state that the
             reference's FSDP spreads over the devices - parameters' fp32 masters' UPDATE and the optimizer
             moments - is owned in contiguous ranges of the flat buffer (`optax.Optimizer(shard=True)`): every
             gradient range is summed onto its owner during the backward (dp.GradShardSync), the owner updates
             its range, the ranks exchange the updated ranges in place.  Adam: equal 1/N slices of whole
             1024-element chunks (a slice of the FLAT buffer, not a cut along each tensor's axis: the same bytes
             per rank, no per-tensor bookkeeping, identical arithmetic - Adam is elementwise, the clip norm is
             all-reduced).  Adafactor (round 4): runs of WHOLE tensors (its factored statistics are per tensor).
             Round 6, Adam: the fp32 PARAMETERS are sharded as well (`ParamStore.shard_master_`): a rank keeps
             the fp32 master of its own slice, plus - replicated - every entry a kernel reads in fp32 (biases,
             LayerNorm, position / token embeddings, cls, t, b: everything that is not a `.../kernel`; the
             reference keeps arrays under `min_size_to_shard_mb` replicated too, the embedding TABLE is this
             implementation's exception to its rule) and every frozen tensor.  The matmul kernels reach the GEMMs
             through the bf16 compute copy, which stays resident and is all-gathered after every update (half
             the bytes of the fp32 exchange it replaces) where XLA would re-gather per use; the updated fp32 of
             the replicated entries travels in one all-reduce.  `store.tree()` stays complete and live (bf16
             views for sharded kernels), `store.full_tree()` / `u.save_train_state` gather the fp32 slices.
             `config.fsdp_shard_params = False` keeps the round-4 form.  Adafactor: the same with runs of whole
             tensors (fp32 master and momentum of the own run only; one broadcast of the bf16 copy per owner).
`shard_dim` and `logical_partitioning` raise NotImplementedError naming the parameter - instead of silently
running replicated under a config that asked for something else.
A spec is a tuple with one entry per array axis (None = not sharded), like the reference's
intermediate `specs` tree; there is no device mesh object: `mesh` is the rank group (dp.Comm) or None.
"""
from __future__ import annotations

import re

from big_vision_amd import utils as u

DEFAULT_STRATEGY = [(".*", "replicate")]
_OPS = {}


def register(name):
  def deco(fn):
    _OPS[name] = fn
    return fn
  return deco


def _parse(op_str):
  """`name` or `name(arg, key=val)` -> (name, args, kwargs); literals only."""
  m = re.fullmatch(r"\s*([A-Za-z_][\w.]*)\s*(?:\((.*)\))?\s*", op_str)
  if not m:
    raise ValueError(f"malformed sharding op {op_str!r}")
  name, argstr = m.group(1), m.group(2)
  args, kwargs = (), {}
  if argstr and argstr.strip():
    import ast
    call = ast.parse(f"f({argstr})", mode="eval").body
    args = tuple(ast.literal_eval(a) for a in call.args)
    kwargs = {k.arg: ast.literal_eval(k.value) for k in call.keywords}
  return name, args, kwargs


@register("replicate")
def replicate():
  """Full replication (:83-101): keeps the spec, refuses a spec that is already sharded."""
  def update(cur_spec, mesh, name, x):
    del mesh, x
    if not all(axis is None for axis in cur_spec):
      raise ValueError(f"Inconsistent sharding instructions: parameter {name} has spec {cur_spec}, "
                       "so it can't be fully replicated.")
    return cur_spec
  return update


def _unsupported(rule):
  def factory(*args, **kwargs):
    def update(cur_spec, mesh, name, x):
      raise NotImplementedError(
          f"sharding rule '{rule}{args or ''}' on parameter {name}: this build keeps every parameter "
          "replicated on every rank (RCCL all-reduce of the gradients); parameter sharding "
          "(big_vision/sharding.py:104-197) is not implemented - use [('.*', 'replicate')]")
    return update
  return factory


@register("fsdp")
def fsdp(axis, min_size_to_shard_mb=4):
  """FSDP rule (:104-139): the largest not-yet-sharded dimension that the device count divides gets `axis`;
  tensors of at most `min_size_to_shard_mb` MiB stay replicated."""
  axis = axis if isinstance(axis, str) else tuple(axis)

  def update(cur_spec, mesh, name, x):
    del name
    shape = tuple(x.shape)
    axis_size = int(getattr(mesh, "size", 1) or 1)
    itemsize = x.element_size() if hasattr(x, "element_size") else getattr(getattr(x, "dtype", None), "itemsize", 4)
    numel = 1
    for d in shape:
      numel *= int(d)
    if numel * itemsize <= min_size_to_shard_mb * (2 ** 20):
      return cur_spec
    for i in sorted(range(len(shape)), key=lambda j: shape[j])[::-1]:   # np.argsort(shape)[::-1]
      if shape[i] % axis_size == 0 and cur_spec[i] is None:
        return cur_spec[:i] + (axis,) + cur_spec[i + 1:]
    return cur_spec      # nothing divisible / free: stays as it is (the reference logs and moves on)
  return update


for _r in ("shard_dim", "logical_partitioning"):
  _OPS[_r] = _unsupported(_r)


def is_sharded(specs) -> bool:
  """True if any leaf of a spec tree names a mesh axis (i.e. the strategy asked for fsdp somewhere)."""
  if isinstance(specs, dict):
    return any(is_sharded(v) for v in specs.values())
  return any(a is not None for a in specs)


def infer_sharding(params, strategy=None, mesh=None):
  """-> tree of specs (tuples of None) with the structure of `params` (:38-71).  Every leaf is
  matched by at most one strategy entry (first pattern wins, utils.make_mask_trees semantics); an
  unmatched leaf stays replicated, as in the reference.  Raises on shard_dim / logical_partitioning."""
  strategy = list(strategy if strategy is not None else DEFAULT_STRATEGY)
  flat, names = u.tree_flatten_with_names(params)
  by_name = dict(flat)
  specs = {n: (None,) * len(getattr(v, "shape", ())) for n, v in flat}
  taken = set()
  for pattern, tactic in strategy:
    rx = re.compile(pattern)
    hits = [n for n in names if n not in taken and rx.fullmatch(n)]
    taken.update(hits)
    for op_str in tactic.split("|"):
      opname, args, kwargs = _parse(op_str)
      opname = opname[len("shardings."):] if opname.startswith("shardings.") else opname
      if opname not in _OPS:
        raise KeyError(f"unknown sharding rule {opname!r} (known: {sorted(_OPS)})")
      op = _OPS[opname](*args, **kwargs)
      for n in hits:
        specs[n] = op(specs[n], mesh, n, by_name[n])
  return u.recover_tree(names, [specs[n] for n in names])


def check_config(config, params, mesh=None):
  """What a trainer calls once: validates `config.sharding_strategy` / `config.sharding_rules`
  and returns the spec tree (`is_sharded(specs)` tells the trainer to build the sharded optimizer)."""
  if config.get("sharding_rules"):
    raise NotImplementedError("config.sharding_rules (logical axis partitioning, sharding.py:142-166) "
                              "is not implemented: parameters are replicated")
  return infer_sharding(params, config.get("sharding_strategy", DEFAULT_STRATEGY), mesh)
