"""ParamStore with a SHARDED fp32 master (the parameter half of the "fsdp" placement, reference sharding.py:104-139):
host-side bookkeeping on the CPU - which entries stay replicated, what `t()` / `tree()` resolve to, that gathering the
slices of all "ranks" gives back the master, that the exchange of the replicated entries is exact.  The device side (the
sharded Adam step, the bf16 all-gather) is covered by tests/test_dp_two_ranks_gpu.py [fsdp]."""
import numpy as np
import pytest
import torch

from big_vision_amd import utils as u
from big_vision_amd.models.proj.image_text import two_towers

IMG = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="map")
TXT = dict(width=128, depth=2, mlp_dim=256, num_heads=2, vocab_size=50)


class _FakeComm:
  """The collectives of dp.Comm over a list of stores that play the ranks of one job (sequential, in-process)."""
  active = True

  def __init__(self, world):
    self.size, self.peers = world, []

  def all_gather_flat_(self, flat, lo, hi, S):
    n = flat.numel()
    for st in self.peers:
      a, b = st.own
      b = min(b, n)
      if b > a:
        flat[a:b] = st.master_own[:b - a]

  def broadcast_ranges_(self, flat, bounds):
    for r, st in enumerate(self.peers):
      a, b = bounds[r], bounds[r + 1]
      assert (a, b) == st.own
      if b > a:
        flat[a:b] = st.master_own[:b - a].to(flat.dtype)

  def all_reduce_sum_(self, t):
    raise AssertionError("exchanged through _exchange_all below")


def _stores(world, frozen_leaves=()):
  out = []
  for r in range(world):
    m = two_towers.Model(image=IMG, text=TXT, out_dim=(None, 32), temperature_init=10.0, bias_init=-10.0)
    st = m.make_store((2, 32, 32, 3), (2, 8), device="cpu", frozen_leaves=frozen_leaves)
    st.init_random(0)
    st._shadow_dirty = False
    out.append(st)
  return out


@pytest.mark.parametrize("world", [1, 2, 4])
def test_sharded_master_round_trips(world):
  stores = _stores(world)
  ref = stores[0].master.clone()
  n_tr = stores[0].trainable_count
  S = (n_tr + world * 1024 - 1) // (world * 1024) * 1024
  comm = _FakeComm(world)
  comm.peers = stores
  for r, st in enumerate(stores):
    lo = min(n_tr, r * S)
    st.shard_master_(lo, min(n_tr, lo + S), S, comm)
    assert st.master is None and st.master_sharded
  st = stores[-1]
  # replicated: everything that is not a matmul kernel
  assert all(not n.endswith("/kernel") for n in st.small_off)
  assert "txt/Embed_0/embedding" in st.small_off and "t" in st.small_off and "img/pos_embedding" in st.small_off
  assert sum(st.entries[n].numel for n in st.small_off) < 0.25 * n_tr
  # a kernel of another rank's slice is not here; the tree stays complete (bf16 compute copy for kernels)
  foreign = [n for n, e in st.entries.items() if n.endswith("/kernel") and not (st.own[0] <= e.offset and e.offset + e.numel <= st.own[1])]
  if world > 1:
    assert foreign
    with pytest.raises(KeyError, match="owner rank"):
      st.t(foreign[0])
  tree = dict(u.tree_flatten_with_names(dict(st.tree()))[0])
  assert set(tree) == set(stores[0].leaf_names())
  assert tree["img/Transformer/encoderblock_0/MlpBlock_0/Dense_0/kernel"].dtype == torch.bfloat16
  assert tree["img/Transformer/encoderblock_0/MlpBlock_0/Dense_0/bias"].dtype == torch.float32
  # gathering gives the master back, on every rank
  for s2 in stores:
    assert torch.equal(s2.gather_master(), ref)
  full = dict(u.tree_flatten_with_names(dict(stores[0].full_tree()))[0])
  e = stores[0].entries["txt/head/kernel"]
  assert torch.equal(full["txt/head/kernel"], ref[e.offset:e.offset + e.numel].view(e.shape))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_master_cut_at_tensor_boundaries_round_trips(world):
  """The sharded Adafactor owns whole tensors (optax.Optimizer._init_adafactor): unequal runs, exchanged by one
  broadcast per owner (`bounds`) instead of the equal-slice all-gather."""
  stores = _stores(world)
  ref = stores[0].master.clone()
  st0 = stores[0]
  n_tr = st0.trainable_count
  share = (n_tr + world - 1) // world
  starts = sorted(e.offset for e in st0.entries.values() if e.name not in st0.frozen)
  bounds = [0] + [next((o for o in starts if o >= r * share), n_tr) for r in range(1, world)] + [n_tr]
  assert len(set(b1 - b0 for b0, b1 in zip(bounds, bounds[1:]))) > 1, "the toy model happens to cut evenly"
  comm = _FakeComm(world)
  comm.peers = stores
  for r, st in enumerate(stores):
    st.shard_master_(bounds[r], bounds[r + 1], 0, comm, bounds=bounds)
    assert st.master is None and st.master_sharded and st.master_own.numel() == bounds[r + 1] - bounds[r]
  for st in stores:
    assert torch.equal(st.gather_master(), ref)
  # every entry lies inside exactly one owner's run
  for e in st0.entries.values():
    if e.name not in st0.frozen:
      assert sum(b0 <= e.offset and e.offset + e.numel <= b1 for b0, b1 in zip(bounds, bounds[1:])) == 1, e.name


def test_exchange_of_the_replicated_entries_is_exact():
  world = 2
  stores = _stores(world)
  n_tr = stores[0].trainable_count
  S = (n_tr + world * 1024 - 1) // (world * 1024) * 1024
  comm = _FakeComm(world)
  comm.peers = stores
  for r, st in enumerate(stores):
    st.shard_master_(min(n_tr, r * S), min(n_tr, r * S + S), S, comm)
  # every owner "updates" its slice; then the all-reduce of the owned pieces (done by hand over the two stores)
  for st in stores:
    st.master_own.add_(1.0)
  tmp = torch.zeros(stores[0].small_trainable)
  for st in stores:
    for so, do, ln in st._own_small:
      assert torch.all(tmp[do:do + ln] == 0), "two ranks own the same piece"
      tmp[do:do + ln] = st.master_own[so:so + ln]
  for st in stores:
    st.master_small[:st.small_trainable].copy_(tmp)
  want = stores[0].gather_master()
  for st in stores:
    for name, o in st.small_off.items():
      e = st.entries[name]
      assert torch.equal(st.master_small[o:o + e.numel], want[e.offset:e.offset + e.numel]), name
  b = stores[0].t("img/Transformer/encoderblock_1/LayerNorm_0/scale")
  assert torch.allclose(b, torch.full_like(b, 2.0))      # ones at init, + 1


def test_frozen_tower_stays_replicated():
  m = two_towers.Model(image=IMG, text=TXT, out_dim=(None, 32), temperature_init=10.0, bias_init=-10.0)
  frozen = [n for n in m.leaf_names((2, 32, 32, 3), (2, 8)) if n.startswith("img/")]
  st, = _stores(1, frozen_leaves=frozen)
  ref = st.master.clone()
  n_tr = st.trainable_count
  st.shard_master_(0, n_tr, n_tr, None)
  assert "img/embedding/kernel" in st.small_off          # frozen kernels keep their fp32 values on every rank
  assert st.small_off["img/embedding/kernel"] >= st.small_trainable
  assert torch.equal(st.gather_master(), ref)
