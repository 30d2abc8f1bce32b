"""The reference's OWN optimizer factory, executed: tests/golden/refoptax_* are written by oracle/run_reference_optax.py,
which imports /root/reference/big_vision/optax.py and utils.py UNMODIFIED over the stand-ins of oracle/refshim (optax
itself is restated there: what is pinned is the reference's WIRING of the chain, see oracle/refshim/optax/__init__.py).

* `bv_oracle.OptaxOracle` (the checker every optimizer GPU test trusts) reproduces the executed reference: the updates of
  three steps, the optimizer state after them, the schedule values;
* the PRODUCT's host logic against the same fixtures: `utils.create_learning_rate_schedule`, and the names and shapes of
  `Optimizer.state_tree()` (the checkpoint contract) for real model trees under Adam and Adafactor;
* when /root/reference is present the fixtures are regenerated and must equal the committed ones.

The reference's schedule functions return float32 (utils.py:1141); the stand-ins compute in float64 throughout and the
oracle / product round the value to float32, hence 3e-7 relative on anything a schedule value enters."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bv_oracle as O  # noqa: E402
import run_reference_optax as RO  # noqa: E402  (case tables only; nothing of the reference is imported here)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REL = 3e-7


def _load(name):
  z = np.load(os.path.join(GOLDEN, f"refoptax_{name}.npz"))
  return z, json.loads(bytes(z["meta"]).decode())


def _tree(z, prefix):
  return O.recover_tree([(k[len(prefix):], torch.from_numpy(np.asarray(z[k]))) for k in z.files if k.startswith(prefix)])


def _close(got, want, what, rel=REL):
  got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
  assert got.shape == want.shape, (what, got.shape, want.shape)
  scale = max(1e-30, float(np.max(np.abs(want))))
  assert float(np.max(np.abs(got - want))) <= rel * scale, (what, float(np.max(np.abs(got - want))), scale)


def _run_oracle(z, meta, mutate=None):
  cfg = json.loads(json.dumps(meta["config"]))
  if mutate:
    mutate(cfg)
  params = _tree(z, "param/")
  opt = O.OptaxOracle(cfg, params, sched_kw=dict(meta["sched_kw"]))
  updates = []
  for step in range(meta["steps"]):
    upd = opt.update(_tree(z, f"grad/{step}/"), params)
    updates.append(dict(O.tree_flatten_with_names(upd)))
    params = O.tree_map(lambda p, u_: p + u_, params, upd)
  return opt, updates


@pytest.mark.parametrize("name", sorted(RO.CASES))
def test_oracle_chain_reproduces_the_executed_reference(name):
  z, meta = _load(name)
  opt, updates = _run_oracle(z, meta)
  names = [k[len("param/"):] for k in z.files if k.startswith("param/")]
  for step, upd in enumerate(updates):
    for n in names:
      _close(upd[n].numpy(), z[f"update/{step}/{n}"], f"step {step} {n}")
  assert meta["count"] == opt.count == meta["steps"]
  assert meta["n_schedules"] == len(opt.schedule_fns)
  for i, fn in enumerate(opt.schedule_fns):
    _close([fn(s) for s in range(meta["sched_kw"]["total_steps"] + 1)], z[f"sched/{i}"], f"schedule {i}")
  # every gradient clip case triggers on some steps and not on others (otherwise the clip stage would be untested)
  if meta["config"].get("grad_clip_norm"):
    norms = [np.sqrt(sum(float(np.sum(z[k] ** 2)) for k in z.files if k.startswith(f"grad/{s}/"))) for s in range(meta["steps"])]
    assert min(norms) < meta["config"]["grad_clip_norm"] < max(norms), norms


def _state_index(meta):
  """Chain position of masked(optimizer): the first component of the state names that hold per-leaf arrays."""
  idx = {n.split("/")[0] for n in meta["state_names"] if n.count("/") >= 3}
  assert len(idx) == 1, idx
  return idx.pop()


@pytest.mark.parametrize("name", [c for c in sorted(RO.CASES) if c != "sgd"])
def test_oracle_state_equals_the_reference_state(name):
  """The optimizer state after three steps under the reference's checkpoint names (utils.py:616-668: tuples are indexed;
  MaskedState.inner_state -> ScaleByAdamState(count, mu, nu) | (FactoredState(count, v_row, v_col, v), clip, EmaState))."""
  z, meta = _load(name)
  opt, _ = _run_oracle(z, meta)
  i = _state_index(meta)
  seen = set()
  if "adam" in meta["config"]["optax_name"]:
    for n, v in opt.mu.items():
      _close(v.double().numpy(), z[f"state/{i}/0/1/{n}"], f"mu {n}"); seen.add(f"{i}/0/1/{n}")
      _close(opt.nu[n].numpy(), z[f"state/{i}/0/2/{n}"], f"nu {n}"); seen.add(f"{i}/0/2/{n}")
    seen.add(f"{i}/0/0")
  else:
    mom = meta["config"].get("optax", {}).get("momentum", 0.9)
    for n, st in opt.af.items():
      one = np.zeros((1,))
      want = (st["v_row"].numpy(), st["v_col"].numpy(), one) if st["fd"] is not None else (one, one, st["v"].numpy())
      for k, w in zip((1, 2, 3), want):
        _close(w, z[f"state/{i}/0/0/{k}/{n}"], f"factored state {k} {n}"); seen.add(f"{i}/0/0/{k}/{n}")
      if mom:
        _close(st["ema"].double().numpy(), z[f"state/{i}/0/2/1/{n}"], f"ema {n}"); seen.add(f"{i}/0/2/1/{n}")
    seen.update({f"{i}/0/0/0"} | ({f"{i}/0/2/0"} if mom else set()))
  # what is left are the counts of the scale_by_schedule stages, nothing else
  rest = set(meta["state_names"]) - seen
  assert rest and all(re.fullmatch(r"\d+/0/0", n) for n in rest), sorted(rest)
  assert all(int(z[f"state/{n}"]) == meta["steps"] for n in rest | {f"{i}/0/0" if "adam" in meta["config"]["optax_name"] else f"{i}/0/0/0"})
  # frozen leaves carry no state at all (optax_test.py:301-318)
  frozen = [n for n, f in opt.frozen.items() if f]
  assert not any(n.endswith("/" + f) for n in meta["state_names"] for f in frozen)


def test_the_comparison_bites():
  """A chain with another mask or another match order must not pass: a decay mask that also covers biases, the
  lr_mults listed in another order (first match wins), a leaf that is frozen in the reference's run and live here."""
  z, meta = _load("adam_frozen_mults")

  def worst_rel(mutate=None):
    _, updates = _run_oracle(z, meta, mutate)
    worst = 0.0
    for step, upd in enumerate(updates):
      for n, v in upd.items():
        ref = z[f"update/{step}/{n}"]
        worst = max(worst, float(np.max(np.abs(v.numpy() - ref))) / max(1e-30, float(np.max(np.abs(ref)))))
    return worst

  assert worst_rel() <= REL
  assert worst_rel(lambda c: c.update(wd_mults=[(".*", 1.0)])) > 1e-3
  assert worst_rel(lambda c: c.update(lr_mults=[(".*/bias", 0.5), ("txt/head/.*", 3.0), (".*", 1.0)])) > 1e-3
  assert worst_rel(lambda c: c["schedule"].__setitem__(0, ["img/embedding/.*", dict(decay_type="cosine")])) > 1e-3


@pytest.mark.parametrize("who", ["oracle", "product"])
def test_schedules_equal_the_executed_reference(who):
  """utils.py:1070-1143 executed on a grid of decay types / warm-up / cool-down spellings vs the oracle's restatement
  and the PRODUCT's `utils.create_learning_rate_schedule` (host logic of every training step)."""
  z = np.load(os.path.join(GOLDEN, "refoptax_schedules.npz"))
  meta = json.loads(bytes(z["meta"]).decode())
  if who == "oracle":
    make = O.create_learning_rate_schedule
  else:
    from big_vision_amd import utils as u
    make = u.create_learning_rate_schedule
  assert set(meta["schedules"]) == set(RO.SCHEDULES)
  for name, kw in meta["schedules"].items():
    fn = make(**meta["kw"], **kw)
    _close([float(fn(s)) for s in range(meta["kw"]["total_steps"] + 1)], z[name], name, rel=2e-7)


@pytest.fixture()
def dry(monkeypatch):
  """Kernels replaced by a recorder: building a train state on the CPU is host logic only (names, shapes, layouts)."""
  import collections
  from big_vision_amd import _lib, ops
  calls = collections.Counter()
  monkeypatch.setattr(_lib, "call", lambda name, *a: calls.update([name]))
  monkeypatch.setattr(ops, "_chk", lambda t, dtype, name: t)
  monkeypatch.setattr(ops, "_stream", lambda: 0)
  return calls


def test_product_replace_frozen_equals_the_executed_reference():
  """optax.py:44-51 (what a trainer's `l2_grads` goes through, train.py:307)."""
  from big_vision_amd import optax as bv_optax
  from big_vision_amd import utils as u
  want = json.load(open(os.path.join(GOLDEN, "refoptax_state_names.json")))["__replace_frozen__"]
  tree = RO._nest({k: i + 1 for i, k in enumerate(RO.SHAPES)})
  for cname, w in want.items():
    sched = w["schedule"] if isinstance(w["schedule"], dict) else [tuple(x) for x in w["schedule"]]
    got = [[k, int(v)] for k, v in u.tree_flatten_with_names(bv_optax.replace_frozen(sched, tree, 0))[0]]
    assert got == w["result"], cname
  assert any(v == 0 for _, v in want["adam_frozen_mults"]["result"]) and all(v for _, v in want["adam_clip_wd"]["result"])
  with pytest.raises(AssertionError, match="All params must be covered"):
    bv_optax.replace_frozen([("img/.*", None)], tree, 0)


@pytest.mark.parametrize("case", sorted(RO.MODEL_STATE_CASES))
def test_product_optimizer_state_has_the_reference_names_and_shapes(dry, case):
  """`Optimizer.state_tree()` of the product for a real two-tower parameter tree = the names and shapes of the state the
  reference's `make(...).init(params)` builds for the same tree and config (chain positions, MaskedNode gaps for frozen
  leaves, FactoredState placeholders of shape (1,), stacked leaves of scan models)."""
  from big_vision_amd import utils as u
  from big_vision_amd.compat.ml_collections import ConfigDict
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  want = json.load(open(os.path.join(GOLDEN, "refoptax_state_names.json")))[case]      # ("__replace_frozen__" is not a case)
  zf = np.load(os.path.join(GOLDEN, f"{want['fixture']}.npz"))
  mcfg = json.loads(bytes(zf["meta"]).decode())["config"]
  mcfg["image"]["patch_size"] = tuple(mcfg["image"]["patch_size"])
  if not isinstance(mcfg["out_dim"], int):
    mcfg["out_dim"] = tuple(mcfg["out_dim"])
  model = two_towers.Model(**mcfg)
  cfg = ConfigDict(dict(want["config"], total_steps=RO.SCHED_KW["total_steps"]))
  state, _ = siglip.make_train_state(model, cfg, tuple(zf["in/image"].shape), tuple(zf["in/text"].shape), rng=0, device="cpu",
                                     total_steps=RO.SCHED_KW["total_steps"])
  got = {k: list(np.shape(v)) for k, v in u.tree_flatten_with_names(state["opt"].state_tree())[0]}
  ref = {k: s for k, s in want["state"]}
  assert set(got) == set(ref), (sorted(set(got) - set(ref))[:6], sorted(set(ref) - set(got))[:6])
  bad = {k: (got[k], ref[k]) for k in ref if got[k] != ref[k]}
  assert not bad, dict(list(bad.items())[:6])


@pytest.mark.skipif(not os.path.isdir(os.path.join(RO.REFERENCE, "big_vision")), reason="the reference tree is not on this host")
def test_committed_fixtures_are_what_the_reference_produces_now(tmp_path):
  """Regenerates every fixture from /root/reference (a subprocess: `big_vision` must resolve to the reference there) and
  compares with the committed files."""
  r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "run_reference_optax.py"), str(tmp_path)],
                     capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stderr[-2000:]
  for f in sorted(os.listdir(tmp_path)):
    if f.endswith(".json"):
      assert json.load(open(tmp_path / f)) == json.load(open(os.path.join(GOLDEN, f))), f
      continue
    a, b = np.load(tmp_path / f), np.load(os.path.join(GOLDEN, f))
    assert sorted(a.files) == sorted(b.files), f
    for k in a.files:
      assert np.array_equal(a[k], b[k]), (f, k)
