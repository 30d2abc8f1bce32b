"""Host-side utilities: the subset of big_vision/utils.py (Copyright 2024 Big Vision Authors, Apache-2.0) that
the hot path's callers use - leaf naming (utils.py:616-862), regex masks (:1169-1212), durations -> steps
(:1002-1067), learning-rate schedules (:1070-1143), .npz parameter loading (:133-227), mixup (:1146-1154).

What is contract here is BEHAVIOUR: configs, checkpoints and schedules written for the reference must mean the
same thing, and the reference's own known-answer tests (utils_test.py:144-281) are restated in
tests/test_host_cpu.py.  The tree helpers and the pattern compiler are written in this project's own structure;
`steps` and `create_learning_rate_schedule` follow the reference's branch order closely because every branch,
default and assertion message of theirs is observable through a config.  Pure Python / numpy, no device code;
nothing of the hot path's arithmetic lives here.
"""
from __future__ import annotations

import math
import re
from collections.abc import Mapping

import numpy as np


# ------------------------------------------------------------- tree utils ----
def _traverse_with_names(tree, with_inner_nodes=False):
  """(name, leaf) pairs of a nested dict / list / tuple.  The naming contract of utils.py:616-641: children are
  visited in sorted-key order (sequences by index), a leaf's name is its keys joined by '/', `None` nodes do not
  exist, and with `with_inner_nodes` every container is also yielded AFTER its children under its own path."""
  out = []

  def walk(node, prefix):
    if node is None:
      return
    if isinstance(node, Mapping):
      children = [(str(k), node[k]) for k in sorted(node.keys())]
    elif isinstance(node, (list, tuple)):
      children = [(str(i), v) for i, v in enumerate(node)]
    else:
      out.append(("/".join(prefix), node))
      return
    for key, child in children:
      walk(child, prefix + (key,))
    if with_inner_nodes:
      out.append(("/".join(prefix), node))

  walk(tree, ())
  return iter(out)


def tree_flatten_with_names(tree):
  """[(name, value), ...] in sorted-name order, plus a treedef stand-in."""
  nv = list(_traverse_with_names(tree))
  return nv, [n for n, _ in nv]


def recover_tree(keys, values):
  """Inverse of the flattening: '/'-joined names back into nested dicts (utils.py:836-862; keys appear in
  first-seen order - every consumer walks trees in sorted-name order, so only the nesting is contract)."""
  root = {}
  for name, value in zip(keys, values):
    *parents, last = name.split("/")
    node = root
    for part in parents:
      node = node.setdefault(part, {})
    node[last] = value
  return root


def tree_unflatten(names_and_vals):
  return recover_tree(*zip(*names_and_vals))


def tree_map_with_names(f, tree, *rest):
  """f(name, leaf, *leaves of `rest` at the same position) over the leaves; the result has the STRUCTURE of `tree`
  (utils.py:676-696: `tree_def.unflatten`) - dicts stay dicts, lists / tuples keep their type, `None` stays `None`."""
  names_and_vals, _ = tree_flatten_with_names(tree)
  rest_vals = [[v for _, v in tree_flatten_with_names(t)[0]] for t in rest]
  assert all(len(rv) == len(names_and_vals) for rv in rest_vals), "trees of different structure"
  mapped = {n: f(n, v, *[rv[i] for rv in rest_vals]) for i, (n, v) in enumerate(names_and_vals)}

  def rebuild(node, prefix):
    if node is None:
      return None
    if isinstance(node, Mapping):
      return {k: rebuild(node[k], prefix + (str(k),)) for k in node}
    if isinstance(node, (list, tuple)):
      items = [rebuild(v, prefix + (str(i),)) for i, v in enumerate(node)]
      return type(node)(*items) if hasattr(node, "_fields") else type(node)(items)
    return mapped["/".join(prefix)]

  return rebuild(tree, ())


def tree_map(f, tree, *rest):
  return tree_map_with_names(lambda _n, v, *r: f(v, *r), tree, *rest)


def check_and_compile_patterns(patterns):
  """One regex string or a list / tuple of them -> compiled regexes (utils.py:1169-1192).  Leaf names here never
  begin with '/', so a pattern that does can match nothing: refused with the reference's message."""
  if isinstance(patterns, str):
    patterns = (patterns,)
  assert isinstance(patterns, (list, tuple)), patterns
  for pat in patterns:
    assert not pat.startswith("/"), f"Big vision parameter names never start with '/': '{pat}"
  return [re.compile(pat) for pat in patterns]


def make_mask_trees(tree, patterns, *, log=None):
  """One boolean mask tree per pattern, only the first match counts (utils.py:1195-1212)."""
  compiled = check_and_compile_patterns(patterns)

  def matchfirst(name, _):
    matches = []
    for pattern in compiled:
      matches.append(not any(matches) and bool(pattern.fullmatch(name)))
    return np.array(matches)

  multimask = tree_map_with_names(matchfirst, tree)
  return [tree_map(lambda m, i=idx: bool(m[i]), multimask) for idx in range(len(patterns))]


# --------------------------------------------------------------- durations ---
def steps(prefix, config, data_size=None, batch_size=None, total_steps=None, default=ValueError):
  """Duration `prefix` from config in steps — utils.py:1002-1067."""
  suffixes = {"steps", "examples", "epochs", "percent"}
  matches = {f"{prefix}_{s}" for s in suffixes
             if (x := config.get(f"{prefix}_{s}")) is not None and x >= 0}
  assert len(matches) <= 1, f"Only one of '{matches}' should be defined."
  if f"{prefix}_steps" in matches:
    return config[f"{prefix}_steps"]

  def to_integer(x):
    return max(1, round(x)) if x else 0

  if batch_size and f"{prefix}_examples" in matches:
    return to_integer(config[f"{prefix}_examples"] / batch_size)
  if batch_size and data_size and f"{prefix}_epochs" in matches:
    return to_integer(config[f"{prefix}_epochs"] * (data_size / batch_size))
  if total_steps and f"{prefix}_percent" in matches:
    pct = config[f"{prefix}_percent"]
    assert 0.0 <= pct <= 1.0, (
        f"Percents should lie in [0.0, 1.0], but {prefix}_percent is {pct}")
    return to_integer(pct * total_steps)
  if default is ValueError:
    raise ValueError(
        f"Cannot convert {prefix} to steps, due to missing batch_size "
        f"({batch_size}), data_size ({data_size}), total_steps ({total_steps})"
        ", or corresponding entry in config:\n" + "\n".join(config.keys()))
  return default


def create_learning_rate_schedule(total_steps, batch_size=None, data_size=None, base=1.0,
                                  decay_type="stair", scale_with_batchsize=False, **kw):
  """step -> learning-rate multiplier (float32-rounded) — utils.py:1070-1143."""

  def to_steps(name, default=0):
    return steps(name, kw, data_size, batch_size, total_steps, default=default)

  warmup_steps = to_steps("warmup")
  cooldown_steps = to_steps("cooldown")
  assert (total_steps <= 1) or (warmup_steps < total_steps), "warmup_steps is >= total_steps"

  def step_fn(step):
    lr = base
    if scale_with_batchsize:
      lr = lr * batch_size / 256.0
    progress = (step - warmup_steps) / float(total_steps - warmup_steps)
    progress = min(max(progress, 0.0), 1.0)
    if decay_type in ("linear", "polynomial"):
      power = kw.get("power", 1)
      zero = kw.get("end", kw.get("linear_end", 0))
      lr = zero + (lr - zero) * (1.0 - progress) ** power
    elif decay_type == "cosine":
      lr = lr * 0.5 * (1.0 + math.cos(math.pi * progress))
    elif decay_type == "rsqrt":
      t = to_steps("timescale", default=kw.get("timescale", 10_000))
      shift = to_steps("shift", default=kw.get("shift", 0))
      if warmup_steps <= step:
        lr = lr / math.sqrt(1 + (step + shift - warmup_steps) / t)
      else:
        lr = lr / math.sqrt(1 + shift / t)
    elif decay_type == "stair":
      i = int(np.searchsorted(np.array(kw.get("steps", [])), step + 1))
      lr = lr * ([1.0] + list(kw.get("mults", [])))[i]
    else:
      raise ValueError(f"Unknown lr type {decay_type}")
    if warmup_steps:
      lr = lr * min(1.0, step / warmup_steps)
    if cooldown_steps:
      lr = lr * min(1.0, (total_steps - step) / cooldown_steps)
    return float(np.float32(lr))

  return step_fn


# ---------------------------------------------------------------- npz I/O ----
def npload(fname):
  """Loads an .npz (or .npy) file into {name: array} — utils.py:133-151."""
  loaded = np.load(fname, allow_pickle=False)
  if isinstance(loaded, np.ndarray):
    return loaded
  out = {}
  for k in loaded.files:
    v = loaded[k]
    if v.dtype.kind == "V" and v.dtype.itemsize == 2:  # bf16 stored as 2-byte void (:827-833)
      v = (v.view(np.uint16).astype(np.uint32) << 16).view(np.float32)
    out[k] = v
  return out


def load_checkpoint_np(npz, tree=None):
  """utils.py:154-169: flat npz keys -> nested tree."""
  if isinstance(npz, str):
    npz = npload(npz)
  keys, values = zip(*list(npz.items()))
  return recover_tree(keys, values)


def load_params(ckpt, **kw):
  """Loads a parameter tree from `path.npz[:sub/key]` — utils.py:172-227."""
  del kw
  if isinstance(ckpt, str) and ":" in ckpt and not ckpt.startswith(("gs:", "http")) or \
     isinstance(ckpt, str) and ckpt.count(":") > (1 if ckpt.startswith(("gs:", "http")) else 0):
    ckpt, key = ckpt.rsplit(":", 1)
  else:
    key = None
  params = load_checkpoint_np(ckpt) if isinstance(ckpt, str) else ckpt
  if isinstance(params, Mapping):
    if "params" in params:
      params = params["params"]
    elif "opt" in params and "target" in params["opt"]:
      params = params["opt"]["target"]
  if key is not None:
    for k in key.split("/"):
      params = params[k]
  return params


def _to_numpy(v):
  """bfloat16 tensors (Adam / Adafactor momentum with a bf16 accumulator) are written the way the
  reference's checkpoints hold them: raw 2-byte void, which `npload` widens back exactly (utils.py:827-833)."""
  if hasattr(v, "detach"):
    v = v.detach().cpu()
    if str(v.dtype) == "torch.bfloat16":
      import torch
      return v.contiguous().view(torch.int16).numpy().view(np.dtype("V2"))
    return v.numpy()
  return np.asarray(v)


def save_params_npz(fname, tree):
  names_and_vals, _ = tree_flatten_with_names(tree)
  np.savez(fname, **{n: _to_numpy(v) for n, v in names_and_vals})


def save_train_state(fname, train_state):
  """Writes {"params": ..., "opt": ...} as a flat .npz with the reference's '/'-joined key naming
  (`params/<leaf>`, `opt/<chain index>/...`, utils.py:616-641 + trainers/.../siglip.py:263-268): what
  `load_params("file.npz")` / `load_train_state` read back.

  Under the "fsdp" placement on N > 1 ranks `opt.state_tree()` and the parameter gather are COLLECTIVES (the owners'
  fp32 parameter slices, moments / statistics are gathered first): EVERY rank must call this function - a call on rank 0 only deadlocks - and the ranks that should not
  write pass `fname=None` (advisor r4)."""
  params = train_state["params"]
  store = getattr(params, "store", None)
  if store is not None and getattr(store, "master_sharded", False):
    params = store.full_tree()     # sharded parameters: the owners' fp32 slices are gathered first (a collective, like the moments)
  tree = {"params": params, "opt": train_state["opt"].state_tree()}
  if fname is not None:
    save_params_npz(fname, tree)


def load_train_state(fname, train_state):
  """Resumes in place: parameters into the store (master + bf16 shadow), optimizer moments and
  step count into the fused optimizer.  Returns the train_state."""
  flat = npload(fname) if isinstance(fname, str) else dict(fname)
  params = recover_tree(*zip(*[(k[len("params/"):], v) for k, v in flat.items() if k.startswith("params/")]))
  store = train_state["params"].store
  store.load_tree(params)
  store.refresh_shadow()
  train_state["opt"].load_state_tree({k[len("opt/"):]: v for k, v in flat.items() if k.startswith("opt/")})
  return train_state


# ------------------------------------------------------------------ mixup ----
def get_mixup_coefficient(rng, step, p):
  """a ~ Beta(p, p), a := max(a, 1 - a) (utils.py:1148-1150).  The reference draws it from
  jax.random with the step-folded key; here a host torch.Generator seeded from (rng, step)
  plays that role (same distribution, not the same bits - tests pass `mixup_a` explicitly)."""
  import torch
  from big_vision_amd.models.vit import _seed_of
  g = torch.Generator().manual_seed((_seed_of(rng if rng is not None else 0) * 1000003 + int(step)) % (2 ** 63 - 1))
  # Beta(p, p) = first coordinate of Dirichlet(p, p); the only sampler that takes a Generator
  a = float(torch._sample_dirichlet(torch.tensor([float(p), float(p)]), generator=g)[0].item())
  return max(a, 1.0 - a)


def get_mixup(rng, p, step=0):
  """Mirror of utils.get_mixup (utils.py:1146-1154): returns `_mixup(*things, **more_things)` which gives back
  `(rng, things, more_things)` - a TUPLE of the mixed positional things and a DICT of the mixed keyword things, exactly the
  shape train.py:285-289 unpacks (`rng, (images, labels), _ = ...`).  Every thing is a [n, ...] fp32 GPU tensor, mixed with
  its roll by one along dim 0 (bv_mixup kernel) under one coefficient a = max(a, 1 - a), a ~ Beta(p, p)."""
  from big_vision_amd import ops
  a = get_mixup_coefficient(rng, step, p)
  mix = lambda t: ops.mixup(t.contiguous(), a)

  def _mixup(*things, **more_things):
    return rng, tuple(mix(t) for t in things), {k: mix(t) for k, t in more_things.items()}
  _mixup.a = a
  return _mixup


def mixup(rng, *things, p, **more_things):
  """The legacy spelling (utils.py:1158-1159)."""
  return get_mixup(rng, p)(*things, **more_things)
