"""flax.linen stand-in (see ../../README.md): the Module machinery and the layers the reference's model files use.

RESTATED here (from upstream knowledge of Flax; pinned by HuggingFace cross-checks, not by this file):
  * Module = a dataclass of its annotated fields plus `name` / `parent`; a module created inside a parent's
    `@compact` method becomes its child; NAMING RULE: explicit `name=`, else `<ClassName>_<k>` where k counts the
    UNNAMED instances of that class created so far in the parent's current compact call; a module instance called
    twice shares its parameters; `self.param(name, init, *args)` creates or reads `<path>/<name>`;
  * `init(rng, *args, **kw)` -> {"params": tree}; `apply({"params": tree}, *args, **kw)`;
  * the layers' arithmetic (Dense, DenseGeneral, Conv VALID-strided, LayerNorm eps 1e-6 with the fast variance,
    MultiHeadDotProductAttention, Embed, Dropout (identity at rate 0 / deterministic; in train mode a recorded
    Bernoulli mask from the run's "dropout" rng), gelu tanh form);
  * `scan` over a leading parameter axis with a carried first argument, `remat` as the identity.
Everything computes in numpy float64."""
import numpy as np

from . import initializers  # noqa: F401
from jax import nn as _jnn

_MISSING = object()
_stack = []          # modules whose compact method is executing (innermost last)
_run = None          # the active init / apply run: {"mode", "params", "key"}
_scan = []           # active scans: [index, length]
last_dropout_masks = []   # the train-mode Dropout masks of the most recent init / apply run (see Dropout)

broadcast = "broadcast"


def gelu(x, approximate=True):
  return _jnn.gelu(x, approximate)


def tanh(x):
  return np.tanh(x)


def softmax(x, axis=-1):
  return _jnn.softmax(x, axis)


def with_logical_constraint(x, names):
  return x


class Partitioned:
  """flax.linen.Partitioned / LogicallyPartitioned boxes: no model on this path creates one; big_vision/sharding.py only
  tests for them (`is_leaf`, `isinstance`)."""


class LogicallyPartitioned(Partitioned):
  pass


def unbox(tree):
  return tree


def logical_to_mesh_axes(names):
  raise NotImplementedError("logical partitioning is out of scope (DESIGN.md 4.5)")


def compact(fn):
  def wrapped(self, *a, **kw):
    self._autonames, self._children = {}, set()
    _stack.append(self)
    try:
      return fn(self, *a, **kw)
    finally:
      _stack.pop()
      self._autonames, self._children = {}, set()
  wrapped._compact = True
  wrapped.__name__ = getattr(fn, "__name__", "compact")
  return wrapped


def _fields_of(cls):
  """(name, default) of the dataclass fields in definition order, base classes first."""
  out = {}
  for c in reversed(cls.__mro__):
    if c is Module or c is object:
      continue
    for k in getattr(c, "__annotations__", {}):
      if k in ("name", "parent"):
        continue
      out[k] = c.__dict__.get(k, out.get(k, _MISSING))
  return list(out.items())


class Module:
  def __init__(self, *args, name=None, parent=_MISSING, **kw):
    fields = _fields_of(type(self))
    if len(args) > len(fields):
      raise TypeError(f"{type(self).__name__}: {len(args)} positional arguments for {len(fields)} fields")
    vals = dict(zip([k for k, _ in fields], args))
    for k, v in kw.items():
      if k not in dict(fields):
        raise TypeError(f"{type(self).__name__} has no field '{k}'")
      if k in vals:
        raise TypeError(f"{type(self).__name__}: field '{k}' given twice")
      vals[k] = v
    for k, d in fields:
      if k not in vals:
        if d is _MISSING:
          raise TypeError(f"{type(self).__name__}: missing field '{k}'")
        vals[k] = d
    for k, v in vals.items():
      object.__setattr__(self, k, v)
    self._autonames, self._children = {}, set()
    par = (_stack[-1] if _stack else None) if parent is _MISSING else parent
    self.parent = par
    if par is not None:
      if name is None:
        prefix = type(self).__name__
        k = par._autonames.get(prefix, 0)
        par._autonames[prefix] = k + 1
        name = f"{prefix}_{k}"
      if name in par._children:
        raise ValueError(f"two sub-modules named '{name}' in one call of {type(par).__name__}")
      par._children.add(name)
      self._path = par._path + (name,)
    else:
      self._path = ()
    self.name = name

  # ---------------------------------------------------------------- variables --
  def param(self, name, init_fn, *init_args):
    assert _run is not None, "param() outside init / apply"
    path = self._path + (name,)
    node = _run["params"]
    for p in path[:-1]:
      node = node.setdefault(p, {}) if _run["mode"] == "init" else node[p]
    if _scan:
      i, n = _scan[-1]
      if _run["mode"] == "init":
        slot = node.setdefault(path[-1], [])
        if len(slot) == i:
          slot.append(np.asarray(init_fn(_run["key"].fold("/".join(path) + f"#{i}"), *init_args), np.float64))
        return slot[i]
      return np.asarray(node[path[-1]])[i]
    if _run["mode"] == "init":
      if path[-1] not in node:
        node[path[-1]] = np.asarray(init_fn(_run["key"].fold("/".join(path)), *init_args), np.float64)
      return node[path[-1]]
    if path[-1] not in node:
      raise KeyError(f"parameter '{'/'.join(path)}' is not in the variables handed to apply()")
    return np.asarray(node[path[-1]])

  def _run_root(self, mode, params, key, args, kw):
    global _run
    assert _run is None and not _stack, "nested init / apply"
    rngs = kw.pop("rngs", None) or {}
    kw.pop("mutable", None)
    global last_dropout_masks
    _run = {"mode": mode, "params": params, "key": key, "rngs": rngs, "dropout_masks": []}
    try:
      return self(*args, **kw)
    finally:
      last_dropout_masks = _run["dropout_masks"]     # [(module path incl. a scan index, keep mask)] of this run, in call order
      _run = None

  def init(self, rng, *args, **kw):
    params = {}
    self._run_root("init", params, rng, args, kw)
    return {"params": params}

  def init_with_output(self, rng, *args, **kw):
    params = {}
    out = self._run_root("init", params, rng, args, kw)
    return out, {"params": params}

  def apply(self, variables, *args, **kw):
    return self._run_root("apply", variables["params"], None, args, kw)


class _JArray(np.ndarray):
  """A numpy array with jax's VALUE semantics for augmented assignment: `x += y` rebinds x to a new array instead of
  writing into the buffer other names still refer to (naflex_vit.py:243 does `x += posemb` on the tensor it has just
  published as out["stem"]; under plain numpy that entry would silently become the sum).  Layer outputs are handed out
  as this type; it propagates through numpy arithmetic."""

  def __iadd__(self, other):
    return np.add(self, other)

  def __isub__(self, other):
    return np.subtract(self, other)

  def __imul__(self, other):
    return np.multiply(self, other)

  def __itruediv__(self, other):
    return np.true_divide(self, other)


def _value(x):
  return np.asarray(x).view(_JArray)


# ------------------------------------------------------------------------ layers --
class Dense(Module):
  features: int
  use_bias: bool = True
  dtype: object = None
  kernel_init: object = None
  bias_init: object = None

  def __call__(self, x):
    kinit = self.kernel_init or initializers.lecun_normal()
    binit = self.bias_init or initializers.zeros
    kernel = self.param("kernel", kinit, (x.shape[-1], self.features), np.float32)
    y = x @ kernel
    if self.use_bias:
      y = y + self.param("bias", binit, (self.features,), np.float32)
    return _value(y)


class DenseGeneral(Module):
  features: object
  axis: object = -1
  use_bias: bool = True
  dtype: object = None
  kernel_init: object = None
  bias_init: object = None

  def __call__(self, x):
    feats = (self.features,) if isinstance(self.features, int) else tuple(self.features)
    axes = (self.axis,) if isinstance(self.axis, int) else tuple(self.axis)
    axes = tuple(a % x.ndim for a in axes)
    assert axes == tuple(range(x.ndim - len(axes), x.ndim)), "contraction over trailing axes only"
    in_shape = tuple(x.shape[a] for a in axes)
    kinit = self.kernel_init or initializers.lecun_normal()

    def kernel_init_2d(key, shape, dtype):     # Flax initialises the kernel on its flattened (fan_in, fan_out) shape
      flat = (int(np.prod(in_shape)), int(np.prod(feats)))
      return np.reshape(kinit(key, flat, dtype), shape)

    kernel = self.param("kernel", kernel_init_2d, in_shape + feats, np.float32)
    n_in = len(in_shape)
    y = np.tensordot(x, kernel, axes=(list(range(x.ndim - n_in, x.ndim)), list(range(n_in))))
    if self.use_bias:
      y = y + self.param("bias", self.bias_init or initializers.zeros, feats, np.float32)
    return y


class Conv(Module):
  features: int
  kernel_size: object
  strides: object = 1
  padding: object = "SAME"
  use_bias: bool = True
  dtype: object = None
  kernel_init: object = None
  bias_init: object = None

  def __call__(self, x):
    kh, kw = self.kernel_size
    sh, sw = (self.strides, self.strides) if isinstance(self.strides, int) else tuple(self.strides)
    assert self.padding == "VALID", "only the VALID-padded patch-embedding conv is restated"
    n, H, W, C = x.shape
    kernel = self.param("kernel", self.kernel_init or initializers.lecun_normal(), (kh, kw, C, self.features), np.float32)
    oh, ow = (H - kh) // sh + 1, (W - kw) // sw + 1
    out = np.zeros((n, oh, ow, self.features), np.float64)
    for i in range(kh):          # NHWC x HWIO, VALID, stride (sh, sw): out[n,y,x,:] = sum_ijc x[n, y sh + i, x sw + j, c] k[i,j,c,:]
      for j in range(kw):
        out += x[:, i:i + (oh - 1) * sh + 1:sh, j:j + (ow - 1) * sw + 1:sw, :] @ kernel[i, j]
    if self.use_bias:
      out = out + self.param("bias", self.bias_init or initializers.zeros, (self.features,), np.float32)
    return out


class LayerNorm(Module):
  epsilon: float = 1e-6
  dtype: object = None
  use_bias: bool = True
  use_scale: bool = True

  def __call__(self, x):
    mean = np.mean(x, axis=-1, keepdims=True)
    var = np.maximum(0.0, np.mean(x * x, axis=-1, keepdims=True) - mean * mean)    # use_fast_variance
    y = (x - mean) / np.sqrt(var + self.epsilon)
    if self.use_scale:
      y = y * self.param("scale", initializers.ones, (x.shape[-1],), np.float32)
    if self.use_bias:
      y = y + self.param("bias", initializers.zeros, (x.shape[-1],), np.float32)
    return _value(y)


class Dropout(Module):
  rate: float = 0.0
  deterministic: object = None

  def __call__(self, x, deterministic=None):
    det = self.deterministic if deterministic is None else deterministic
    if self.rate == 0.0 or det:
      return x
    # train mode: keep ~ Bernoulli(1 - rate) drawn from the run's "dropout" rng folded with the module path (flax:
    # make_rng("dropout") per module), x * keep / (1 - rate).  The masks are recorded so that a caller can hand the
    # same bits to another implementation (JAX's own stream is not reproduced: only the PLACEMENT and the arithmetic
    # of the reference's dropout are executed here).
    if "dropout" not in _run["rngs"]:
      raise ValueError("Dropout in train mode needs rngs={'dropout': key}")
    path = "/".join(self._path) + (f"#{_scan[-1][0]}" if _scan else "")
    keep = _run["rngs"]["dropout"].fold(path).generator().random(np.shape(x)) < (1.0 - self.rate)
    _run["dropout_masks"].append((path, keep))
    return np.where(keep, np.asarray(x, np.float64) / (1.0 - self.rate), 0.0)


class Embed(Module):
  num_embeddings: int
  features: int
  dtype: object = None
  embedding_init: object = None

  def _table(self):
    init = self.embedding_init or initializers.variance_scaling(1.0, "fan_in", "normal", out_axis=0)
    return self.param("embedding", init, (self.num_embeddings, self.features), np.float32)

  def __call__(self, ids):
    return np.take(self._table(), np.asarray(ids), axis=0)

  def attend(self, query):
    return query @ self._table().T


class MultiHeadDotProductAttention(Module):
  num_heads: int
  dtype: object = None
  qkv_features: object = None
  out_features: object = None
  deterministic: object = None
  kernel_init: object = None
  bias_init: object = None
  use_bias: bool = True
  dropout_rate: float = 0.0

  @compact
  def __call__(self, inputs_q, inputs_kv=None, mask=None, deterministic=None):
    inputs_kv = inputs_q if inputs_kv is None else inputs_kv
    feats = self.qkv_features or inputs_q.shape[-1]
    out_feats = self.out_features or inputs_q.shape[-1]
    assert feats % self.num_heads == 0
    hd = feats // self.num_heads
    kw = dict(kernel_init=self.kernel_init, bias_init=self.bias_init, use_bias=self.use_bias, dtype=self.dtype)
    q = DenseGeneral(features=(self.num_heads, hd), axis=-1, name="query", **kw)(inputs_q)
    k = DenseGeneral(features=(self.num_heads, hd), axis=-1, name="key", **kw)(inputs_kv)
    v = DenseGeneral(features=(self.num_heads, hd), axis=-1, name="value", **kw)(inputs_kv)
    q = q / np.sqrt(hd)
    w = np.einsum("...qhd,...khd->...hqk", q, k)
    if mask is not None:
      w = np.where(mask, w, np.finfo(np.float64).min)
    w = softmax(w, axis=-1)
    assert self.dropout_rate == 0.0
    x = np.einsum("...hqk,...khd->...qhd", w, v)
    return DenseGeneral(features=out_feats, axis=(-2, -1), name="out", **kw)(x)


SelfAttention = MultiHeadDotProductAttention


# -------------------------------------------------------------------- transforms --
def remat(target, **_kw):
  """nn.remat / nn.checkpoint change memory, not results."""
  return target


checkpoint = remat


def scan(target, variable_axes=None, split_rngs=None, in_axes=0, out_axes=0, length=None, **_kw):
  """nn.scan(Block, variable_axes={"params": 0}, in_axes=nn.broadcast, length=L): ONE module whose parameters carry a
  leading axis of length L; calling it threads the first argument through L applications of Block (block i sees
  slice i of every parameter) and stacks the second element of each result along a new leading axis."""
  assert variable_axes == {"params": 0} and in_axes == broadcast and out_axes == 0 and length

  class Scanned(target):   # same class name rules do not matter: the reference names the scanned module explicitly
    def __call__(self, carry, *bcast):
      fields = {k: getattr(self, k) for k, _ in _fields_of(target)}
      ys = []
      for i in range(length):
        _scan.append([i, length])
        try:
          inner = target(**fields, parent=None)
          inner._path = self._path                 # the block's parameters live directly under the scanned module
          carry, y = inner(carry, *bcast)
        finally:
          _scan.pop()
        ys.append(y)
      if _run["mode"] == "init":                   # stack the per-block parameter lists created above
        node = _run["params"]
        for p in self._path:
          node = node[p]
        _stack_lists(node)
      import jax
      return carry, jax.tree.map(lambda *v: np.stack(v), *ys)

  Scanned.__name__ = "Scan" + target.__name__
  return Scanned


def _stack_lists(node):
  for k, v in node.items():
    if isinstance(v, dict):
      _stack_lists(v)
    elif isinstance(v, list):
      node[k] = np.stack(v)
