"""optax stand-in (TEST INFRASTRUCTURE, see ../README.md): enough of optax for the reference's OWN
`big_vision/optax.py` - `make()` (:75-149), `scale_by_adafactor` (:187-216), `get_count`, `replace_frozen` - to be
imported UNMODIFIED and executed (oracle/run_reference_optax.py).

optax is an un-vendored dependency of the reference (big_vision/requirements.txt names `optax` without a version); what
follows RESTATES the published algorithms of the transformations the reference composes - `chain`, `masked` (with its
`MaskedNode` placeholders), `scale`, `scale_by_schedule`, `clip_by_global_norm`, `add_decayed_weights`, `set_to_zero`,
`identity`, `scale_by_adam`, `scale_by_factored_rms`, `clip_by_block_rms`, `ema`, `trace`, `apply_updates` - and the
NamedTuple states whose field ORDER gives the optimizer state its checkpoint names (`u.tree_flatten_with_names` indexes
tuples).  What executing the reference over it pins is the reference's WIRING: the order of the chain, which mask goes
where, how frozen parameters, `lr_mults`, `wd_mults` and several schedules combine, which arguments the BigVision
Adafactor hands to the factored RMS / clip / momentum stages.  The arithmetic of the stages stays restated (and is pinned
by the reference's own known answers, optax_test.py:103-318, in tests/test_oracle.py).

Everything computes in numpy float64 (the policy of ../jax/numpy) with ONE exception: an accumulator dtype of bfloat16
(`mu_dtype`, `ema(accumulator_dtype=...)`, `trace(...)`) rounds the STORED accumulator to bfloat16 - that rounding is
part of the algorithm the product and the oracle implement (2^-9 relative), not a promotion detail."""
from typing import Any, Callable, NamedTuple

import numpy as np

import jax


# ------------------------------------------------------------------ base --
class GradientTransformation(NamedTuple):
  init: Callable
  update: Callable


GradientTransformationExtraArgs = GradientTransformation


class EmptyState(NamedTuple):
  pass


class MaskedNode(NamedTuple):
  """Placeholder that optax.masked puts where the mask is False: a pytree node WITHOUT leaves."""


class MaskedState(NamedTuple):
  inner_state: Any


class ScaleByScheduleState(NamedTuple):
  count: Any


class ScaleByAdamState(NamedTuple):
  count: Any
  mu: Any
  nu: Any


class FactoredState(NamedTuple):
  count: Any
  v_row: Any
  v_col: Any
  v: Any


class EmaState(NamedTuple):
  count: Any
  ema: Any


class TraceState(NamedTuple):
  trace: Any


ScaleState = ClipByGlobalNormState = AddDecayedWeightsState = IdentityState = EmptyState

_tmap = jax.tree.map


def _zero_count():
  return np.zeros([], np.int32)


def _inc(count):
  """optax.numerics.safe_int32_increment."""
  return np.asarray(min(int(count) + 1, np.iinfo(np.int32).max), np.int32)


def _is_bf16(dtype):
  return dtype is not None and str(dtype) in ("bfloat16", "<class 'jax.numpy.bfloat16'>", "jnp.bfloat16")


def _round_bf16(x):
  """float64 -> float32 -> bfloat16 (round to nearest even), returned as float64."""
  u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
  u = (u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))) & np.uint64(0xFFFF0000)
  return u.astype(np.uint32).view(np.float32).astype(np.float64)


def _cast(tree, dtype):
  """optax.tree_utils.tree_cast under this directory's dtype policy (module docstring)."""
  if _is_bf16(dtype):
    return _tmap(_round_bf16, tree)
  return tree


def _zeros_like(tree):
  return _tmap(lambda p: np.zeros(np.shape(p), np.float64), tree)


# ------------------------------------------------------------ combinators --
def chain(*transforms):
  """optax.chain: states as a TUPLE in the order of the arguments, updates threaded left to right."""
  def init(params):
    return tuple(t.init(params) for t in transforms)

  def update(updates, state, params=None):
    assert len(state) == len(transforms)
    new_state = []
    for t, s in zip(transforms, state):
      updates, s = t.update(updates, s, params)
      new_state.append(s)
    return updates, tuple(new_state)

  return GradientTransformation(init, update)


def masked(inner, mask):
  """optax.masked: the inner transformation sees `MaskedNode()` where the mask is False; those updates pass through."""
  def mask_pytree(tree, mask_tree):
    return _tmap(lambda m, p: p if m else MaskedNode(), mask_tree, tree)

  def init(params):
    mask_tree = mask(params) if callable(mask) else mask
    return MaskedState(inner_state=inner.init(mask_pytree(params, mask_tree)))

  def update(updates, state, params=None):
    mask_tree = mask(updates) if callable(mask) else mask
    masked_updates = mask_pytree(updates, mask_tree)
    masked_params = None if params is None else mask_pytree(params, mask_tree)
    new_masked, new_inner = inner.update(masked_updates, state.inner_state, masked_params)
    new_updates = _tmap(lambda m, new_u, old_u: new_u if m else old_u, mask_tree, new_masked, updates)
    return new_updates, MaskedState(inner_state=new_inner)

  return GradientTransformation(init, update)


# ------------------------------------------------------- stateless stages --
def identity():
  return GradientTransformation(lambda params: EmptyState(), lambda updates, state, params=None: (updates, state))


def set_to_zero():
  return GradientTransformation(lambda params: EmptyState(),
                                lambda updates, state, params=None: (_zeros_like(updates), state))


def scale(step_size):
  return GradientTransformation(lambda params: EmptyState(),
                                lambda updates, state, params=None: (_tmap(lambda g: step_size * g, updates), state))


def global_norm(updates):
  return np.sqrt(sum(np.sum(np.square(np.asarray(x, np.float64))) for x in jax.tree.leaves(updates)))


def clip_by_global_norm(max_norm):
  """optax.clip_by_global_norm: untouched below max_norm, else (t / ||g||) * max_norm."""
  def update(updates, state, params=None):
    g_norm = global_norm(updates)
    if g_norm < max_norm:
      return updates, state
    return _tmap(lambda t: (t / g_norm) * max_norm, updates), state

  return GradientTransformation(lambda params: EmptyState(), update)


def clip_by_block_rms(threshold):
  """optax.clip_by_block_rms: every leaf divided by max(1, rms(leaf) / threshold)."""
  def update(updates, state, params=None):
    def clip(u):
      return u / np.maximum(1.0, np.sqrt(np.mean(np.square(u))) / threshold)
    return _tmap(clip, updates), state

  return GradientTransformation(lambda params: EmptyState(), update)


def add_decayed_weights(weight_decay=0.0, mask=None):
  def update(updates, state, params=None):
    if params is None:
      raise ValueError("add_decayed_weights needs params")
    return _tmap(lambda g, p: g + weight_decay * p, updates, params), state

  tx = GradientTransformation(lambda params: EmptyState(), update)
  return tx if mask is None else masked(tx, mask)


# -------------------------------------------------------- stateful stages --
def scale_by_schedule(step_size_fn):
  """optax.scale_by_schedule: the schedule is read at the PRE-increment count."""
  def update(updates, state, params=None):
    step_size = step_size_fn(state.count)
    return _tmap(lambda g: np.asarray(step_size, np.float64) * g, updates), ScaleByScheduleState(count=_inc(state.count))

  return GradientTransformation(lambda params: ScaleByScheduleState(count=_zero_count()), update)


def scale_by_adam(b1=0.9, b2=0.999, eps=1e-8, eps_root=0.0, mu_dtype=None, *, nesterov=False):
  assert not nesterov, "not restated"

  def init(params):
    return ScaleByAdamState(count=_zero_count(), mu=_zeros_like(params), nu=_zeros_like(params))

  def update(updates, state, params=None):
    mu = _tmap(lambda g, t: (1 - b1) * g + b1 * t, updates, state.mu)
    nu = _tmap(lambda g, t: (1 - b2) * np.square(g) + b2 * t, updates, state.nu)
    count = _inc(state.count)
    mu_hat = _tmap(lambda t: t / (1 - b1 ** int(count)), mu)
    nu_hat = _tmap(lambda t: t / (1 - b2 ** int(count)), nu)
    new = _tmap(lambda m, v: m / (np.sqrt(v + eps_root) + eps), mu_hat, nu_hat)
    return new, ScaleByAdamState(count=count, mu=_cast(mu, mu_dtype), nu=nu)

  return GradientTransformation(init, update)


def _factored_dims(shape, factored, min_dim_size_to_factor):
  """optax/_src/factorized.py: the two largest axes, if the second largest is at least min_dim_size_to_factor."""
  if not factored or len(shape) < 2:
    return None
  sorted_dims = np.argsort(shape)
  if shape[sorted_dims[-2]] < min_dim_size_to_factor:
    return None
  return int(sorted_dims[-2]), int(sorted_dims[-1])


class _UpdateResult(NamedTuple):
  update: Any
  v_row: Any
  v_col: Any
  v: Any


def scale_by_factored_rms(factored=True, decay_rate=0.8, step_offset=0, min_dim_size_to_factor=128, epsilon=1e-30,
                          decay_rate_fn=None):
  """optax.scale_by_factored_rms (Adafactor's second-moment scaling).  State leaves that a parameter does not use are
  zeros((1,)) placeholders."""
  if decay_rate_fn is None:
    decay_rate_fn = lambda i, exponent: 1.0 - (np.asarray(i, np.float64) + 1.0) ** (-exponent)
  is_res = lambda x: isinstance(x, _UpdateResult)

  def to_state(count, results):
    pick = lambda k: _tmap(lambda r: getattr(r, k), results, is_leaf=is_res)
    return FactoredState(count=count, v_row=pick("v_row"), v_col=pick("v_col"), v=pick("v"))

  def init(params):
    def one(param):
      shape = tuple(np.shape(param))
      fd = _factored_dims(shape, factored, min_dim_size_to_factor)
      one_ = lambda: np.zeros((1,), np.float64)
      if fd is not None:
        d1, d0 = fd
        return _UpdateResult(one_(), np.zeros(np.delete(shape, d0), np.float64), np.zeros(np.delete(shape, d1), np.float64),
                             one_())
      return _UpdateResult(one_(), one_(), one_(), np.zeros(shape, np.float64))
    return to_state(_zero_count(), _tmap(one, params))

  def update(grads, state, params=None):
    if params is None:
      raise ValueError("scale_by_factored_rms needs params")

    def one(grad, v_row, v_col, v, param):
      shape = tuple(np.shape(param))
      decay_rate_t = decay_rate_fn(state.count - step_offset, decay_rate)
      new_v_row, new_v_col, new_v = np.zeros((1,)), np.zeros((1,)), np.zeros((1,))
      fd = _factored_dims(shape, factored, min_dim_size_to_factor)
      grad_sqr = np.square(grad) + epsilon
      if fd is not None:
        d1, d0 = fd
        new_v_row = decay_rate_t * v_row + (1.0 - decay_rate_t) * np.mean(grad_sqr, axis=d0)
        new_v_col = decay_rate_t * v_col + (1.0 - decay_rate_t) * np.mean(grad_sqr, axis=d1)
        reduced_d1 = d1 - 1 if d1 > d0 else d1
        row_col_mean = np.mean(new_v_row, axis=reduced_d1, keepdims=True)
        row_factor = (new_v_row / row_col_mean) ** -0.5
        col_factor = new_v_col ** -0.5
        upd = grad * np.expand_dims(row_factor, axis=d0) * np.expand_dims(col_factor, axis=d1)
      else:
        new_v = decay_rate_t * v + (1.0 - decay_rate_t) * grad_sqr
        upd = grad * new_v ** -0.5
      return _UpdateResult(upd, new_v_row, new_v_col, new_v)

    out = _tmap(one, grads, state.v_row, state.v_col, state.v, params)
    return _tmap(lambda r: r.update, out, is_leaf=is_res), to_state(_inc(state.count), out)

  return GradientTransformation(init, update)


def ema(decay, debias=True, accumulator_dtype=None):
  """optax.ema: the un-rounded new average is the update, the STORED average is cast to accumulator_dtype."""
  def init(params):
    return EmaState(count=_zero_count(), ema=_zeros_like(params))

  def update(updates, state, params=None):
    new = _tmap(lambda g, t: (1 - decay) * g + decay * t, updates, state.ema)
    count = _inc(state.count)
    out = _tmap(lambda t: t / (1 - decay ** int(count)), new) if debias else new
    return out, EmaState(count=count, ema=_cast(new, accumulator_dtype))

  return GradientTransformation(init, update)


def trace(decay, nesterov=False, accumulator_dtype=None):
  def init(params):
    return TraceState(trace=_zeros_like(params))

  def update(updates, state, params=None):
    new = _tmap(lambda g, t: g + decay * t, updates, state.trace)
    out = _tmap(lambda g, t: g + decay * t, updates, new) if nesterov else new
    return out, TraceState(trace=_cast(new, accumulator_dtype))

  return GradientTransformation(init, update)


def apply_updates(params, updates):
  return _tmap(lambda p, u: np.asarray(p + u, np.float64), params, updates)


def per_example_global_norm_clip(grads, l2_norm_clip):
  """optax.per_example_global_norm_clip: leaves [B, ...]; returns (sum over the batch of the clipped gradients, number
  of clipped examples)."""
  bsize = grads[0].shape[0]
  norms = np.sqrt(sum(np.sum(np.square(g.reshape(bsize, -1)), axis=1) for g in grads))
  divisors = np.maximum(norms / l2_norm_clip, 1.0)
  clipped = [np.einsum("i,i...", 1.0 / divisors, g) for g in grads]
  return clipped, int(np.sum(norms > l2_norm_clip))
