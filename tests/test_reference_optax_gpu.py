"""The PRODUCT's optimizer kernels (bv_adam_step / bv_adafactor_step through `big_vision_amd.optax.make`) on the fixtures the
reference's own `big_vision/optax.py::make` produced (tests/golden/refoptax_*.npz, oracle/run_reference_optax.py:
"reference wiring over restated optax stages", oracle/refshim/optax/__init__.py): same initial parameters, same
gradients step by step, and after every step the parameters must be the reference's - fp32 kernels against a float64
reference, so 2e-6 of the largest parameter plus what ONE rounding of a bf16 accumulator (momentum, mu) moves an update by
(2^-9 of the update).  Frozen leaves must not move at all.  The optimizer's state tree after the last step carries the
reference's names and values."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import run_reference_optax as RO  # noqa: E402  (case tables only)

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(ROOT, "tests", "golden")
# "big_vision.sgd" (= optax.identity) is not one of the fused optimizers (Optimizer raises NotImplementedError for it)
CASES = [c for c in sorted(RO.CASES) if c != "sgd"]


@pytest.mark.parametrize("name", CASES)
def test_product_optimizer_follows_the_executed_reference(dev, name):
  from big_vision_amd import optax as bv_optax
  from big_vision_amd import utils as u
  from big_vision_amd.compat.ml_collections import ConfigDict
  from big_vision_amd.params import Entry, ParamStore
  z = np.load(os.path.join(GOLDEN, f"refoptax_{name}.npz"))
  meta = json.loads(bytes(z["meta"]).decode())
  cfg = ConfigDict(meta["config"])
  names = [k[len("param/"):] for k in z.files if k.startswith("param/")]
  frozen = set(bv_optax.frozen_leaves(cfg, names))
  init0 = lambda gen, shape: torch.zeros(shape)
  store = ParamStore([Entry(n, z[f"param/{n}"].shape, init0) for n in names], dev, frozen=frozen)
  store.load_tree({n: torch.from_numpy(np.asarray(z[f"param/{n}"], np.float32)) for n in names})
  store.refresh_shadow()
  opt, sched_fns = bv_optax.make(cfg, store, sched_kw=dict(meta["sched_kw"]))
  assert len(sched_fns) == meta["n_schedules"]
  ref = {n: np.asarray(z[f"param/{n}"], np.float64) for n in names}
  bf16_acc = "adafactor" in meta["config"]["optax_name"] and str(meta["config"].get("optax", {}).get("dtype_momentum", "bfloat16")) == "bfloat16" \
      or meta["config"].get("optax", {}).get("mu_dtype") == "bfloat16"
  store.want_grads = True
  store.ensure_grad()
  for step in range(meta["steps"]):
    store.zero_grad()
    for n in names:
      if n not in frozen:
        store.leaf(n, "grad").copy_(torch.from_numpy(np.asarray(z[f"grad/{step}/{n}"], np.float32)))
    opt.step()
    torch.cuda.synchronize()
    assert bv_optax.get_count(opt) == step + 1
    for n in names:
      upd = np.asarray(z[f"update/{step}/{n}"], np.float64)
      ref[n] = ref[n] + upd
      got = store.leaf(n).detach().cpu().double().numpy()
      if n in frozen:
        assert np.array_equal(got, np.asarray(z[f"param/{n}"], np.float32).astype(np.float64)), f"frozen leaf {n} moved"
        continue
      tol = 2e-6 * max(1.0, float(np.max(np.abs(ref[n])))) + (2.0 ** -8 * float(np.max(np.abs(upd))) * (step + 1) if bf16_acc else 0.0)
      assert float(np.max(np.abs(got - ref[n]))) <= tol, (step, n, float(np.max(np.abs(got - ref[n]))), tol)
  # the state tree after the last step: the reference's names, and its values within the accumulators' precision
  got_state = {k: np.asarray(v.detach().cpu().float().numpy() if torch.is_tensor(v) else v, np.float64)
               for k, v in u.tree_flatten_with_names(opt.state_tree())[0]}
  assert set(got_state) == set(meta["state_names"]), (sorted(set(got_state) ^ set(meta["state_names"]))[:8])
  for k in meta["state_names"]:
    want = np.asarray(z[f"state/{k}"], np.float64)
    assert got_state[k].shape == want.shape, (k, got_state[k].shape, want.shape)
    scale = max(1e-30, float(np.max(np.abs(want))))
    assert float(np.max(np.abs(got_state[k] - want))) <= (1e-2 if bf16_acc else 2e-5) * scale, (k, float(np.max(np.abs(got_state[k] - want))), scale)
