"""jax.lax collectives over a named axis, emulated for the reference's pmap-style per-device code: `spmd(fn, n, args)`
runs fn once per virtual device, each in its own thread; a collective is a rendezvous (every device deposits its value,
a barrier, every device reads all of them).  all_gather / psum / pmean / axis_index only - what
trainers/proj/image_text/_deprecated_contrastive.py:67-200 calls."""
import threading

import numpy as np

_tls = threading.local()


class _Group:
  def __init__(self, n):
    self.n = n
    self.barrier = threading.Barrier(n)
    self.slots = [None] * n

  def exchange(self, rank, value):
    self.slots[rank] = value
    self.barrier.wait()
    vals = list(self.slots)
    self.barrier.wait()          # nobody overwrites a slot before everyone has read
    return vals


def spmd(fn, n, per_device_args):
  """[fn(*per_device_args[r]) for r in range(n)] with the collectives below connecting the n calls."""
  group, out, err = _Group(n), [None] * n, []

  def body(r):
    _tls.rank, _tls.group = r, group
    try:
      out[r] = fn(*per_device_args[r])
    except BaseException as e:   # pylint: disable=broad-except
      err.append(e)
      group.barrier.abort()

  ts = [threading.Thread(target=body, args=(r,)) for r in range(n)]
  for t in ts:
    t.start()
  for t in ts:
    t.join()
  if err:
    raise err[0]
  return out


def device_count():
  g = getattr(_tls, "group", None)
  return g.n if g is not None else 1


def axis_index(axis_name):
  return _tls.rank


def all_gather(x, axis_name):
  return np.stack(_tls.group.exchange(_tls.rank, np.asarray(x)))


def _tree(f, x):
  if isinstance(x, dict):
    return {k: _tree(f, v) for k, v in x.items()}
  if isinstance(x, (list, tuple)):
    return type(x)(_tree(f, v) for v in x)
  return f(x)


def psum(x, axis_name):
  return _tree(lambda v: np.sum(np.stack(_tls.group.exchange(_tls.rank, np.asarray(v, np.float64))), axis=0), x)


def pmean(x, axis_name):
  return _tree(lambda v: np.mean(np.stack(_tls.group.exchange(_tls.rank, np.asarray(v, np.float64))), axis=0), x)


class GatherDimensionNumbers:
  def __init__(self, offset_dims, collapsed_slice_dims, start_index_map, **_kw):
    self.offset_dims, self.collapsed_slice_dims, self.start_index_map = tuple(offset_dims), tuple(collapsed_slice_dims), tuple(start_index_map)


def gather(operand, start_indices, dimension_numbers, slice_sizes, mode=None, **_kw):
  """jax.lax.gather for the ONE pattern naflex_vit.py:69-79 uses: operand [A, B, C], indices [L, 3] = (a, b, 0), slices
  [1, 1, C] with the first two dims collapsed -> [L, C] = operand[a, b, :]; mode="fill": an index outside the operand
  yields NaN (the documented failure mode of grids beyond the resize canvas)."""
  dn = dimension_numbers
  assert dn.offset_dims == (1,) and dn.collapsed_slice_dims == (0, 1) and dn.start_index_map == (0, 1, 2), vars(dn)
  operand, idx = np.asarray(operand), np.asarray(start_indices)
  assert list(slice_sizes) == [1, 1, operand.shape[-1]] and mode == "fill" and idx.shape[-1] == 3
  a, b = idx[:, 0], idx[:, 1]
  ok = (a >= 0) & (a < operand.shape[0]) & (b >= 0) & (b < operand.shape[1]) & (idx[:, 2] == 0)
  out = operand[np.clip(a, 0, operand.shape[0] - 1), np.clip(b, 0, operand.shape[1] - 1), :].astype(np.float64)
  out[~ok] = np.nan
  return out
