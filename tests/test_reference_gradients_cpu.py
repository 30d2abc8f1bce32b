"""value_and_grad of the reference's own step losses against the oracle's backward.

oracle/run_reference_gradients.py executes the reference's files - `two_towers.py` -> `vit.py` / `text_transformer.py`
followed by `_deprecated_contrastive.py::sigmoid_loss` / `softmax_loss` (one device: the function siglip.py:287-308
differentiates), and `vit.py` followed by `utils.py::sigmoid_xent` / `softmax_xent` (train.py:281-299) - over the
stand-ins of oracle/refshim and takes, for EVERY parameter leaf, the derivative of that loss along one stored direction by
4th-order central differences in float64 (no automatic differentiation on that side).  Here `bv_oracle`'s autograd
gradient - what every gradient test of the product is checked against - must satisfy <grad[leaf], v> = dF/dv leaf by
leaf, and the loss itself must agree to 1e-10.  What this does and does not pin: oracle/refshim/README.md (the wiring and
the composition are the reference's, the primitives' arithmetic is restated on both sides)."""
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bv_oracle as O  # noqa: E402
import run_reference_gradients as RG  # noqa: E402  (the case table only; nothing of the reference is imported here)
from test_reference_wiring_cpu import _nest, _unstack_scan  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = sorted(RG.CASES)


def load_case(name, folder=GOLDEN):
  z = np.load(os.path.join(folder, f"refgrad_{name}.npz"))
  return z, json.loads(bytes(z["meta"]).decode())


def oracle_loss_and_leaves(z, meta):
  """(loss, {leaf name: float64 leaf tensor with requires_grad}) of the oracle's restatement of the same composition."""
  cfg, kind, loss_name = meta["config"], meta["kind"], meta["loss"]
  leaves = {l: torch.from_numpy(np.asarray(z[f"param/{l}"], np.float64)).requires_grad_(True) for l in meta["leaves"]}
  params = _unstack_scan(_nest(leaves))
  image = torch.from_numpy(z["in/image"])
  if kind == "two":
    text = torch.from_numpy(z["in/text"]).long()
    image_cfg = dict(cfg["image"])
    image_cfg["patch_size"] = tuple(image_cfg["patch_size"])
    out_dim = cfg["out_dim"] if isinstance(cfg["out_dim"], int) else tuple(cfg["out_dim"])
    if loss_name == "sigmoid":
      loss, _ = O.siglip_step_loss(params, image, text, image_cfg=image_cfg, text_cfg=cfg["text"], out_dim=out_dim)
    else:
      zi, zt, out = O.two_towers_forward(params, image, text, image_cfg=image_cfg, text_cfg=cfg["text"], out_dim=out_dim)
      loss = O.softmax_loss_per_device(zi, zt, [zi], [zt], 0, out["t"])
  else:
    kw = {**O.decode_variant(cfg.get("variant")), **{k: v for k, v in cfg.items() if k not in ("variant", "num_classes")}}
    kw["patch_size"] = tuple(kw["patch_size"])
    loss, _ = O.classification_step_loss(params, image, torch.from_numpy(z["in/labels"]), model_cfg=kw,
                                         num_classes=cfg["num_classes"], loss=loss_name)
  return loss, leaves


def mismatches(z, meta, grads):
  """[(leaf, <grad, v>, dF/dv, tolerance)] of the leaves outside the tolerance.  Tolerance per leaf: 1e-9 of the case's
  gradient scale (median |dF/dv|) + ten times the finite difference's own h-vs-2h disagreement."""
  scale = float(np.median([abs(float(z[f"dd/{l}"])) for l in meta["leaves"]]))
  bad = []
  for l in meta["leaves"]:
    v = torch.from_numpy(np.asarray(z[f"dir/{l}"], np.float64))
    got = float((grads[l] * v).sum())
    want, want2 = float(z[f"dd/{l}"]), float(z[f"dd2/{l}"])
    tol = 1e-9 * scale + 10.0 * abs(want - want2)
    if not abs(got - want) <= tol:
      bad.append((l, got, want, tol))
  return bad


@pytest.mark.parametrize("name", CASES)
def test_oracle_gradient_is_the_derivative_of_the_executed_reference(name):
  z, meta = load_case(name)
  loss, leaves = oracle_loss_and_leaves(z, meta)
  assert abs(float(loss.detach()) - float(z["loss"])) <= 1e-10 * max(1.0, abs(float(z["loss"]))), (float(loss.detach()), float(z["loss"]))
  names = list(leaves)
  grads = dict(zip(names, torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)))
  grads = {n: (g if g is not None else torch.zeros_like(leaves[n])) for n, g in grads.items()}
  bad = mismatches(z, meta, grads)
  assert not bad, bad[:5]
  # the finite differences themselves are tight: h vs 2h agree to 1e-6 of the gradient scale on every leaf
  scale = float(np.median([abs(float(z[f"dd/{l}"])) for l in meta["leaves"]]))
  assert max(abs(float(z[f"dd/{l}"]) - float(z[f"dd2/{l}"])) for l in meta["leaves"]) <= 1e-5 * scale


def test_the_comparison_bites():
  """A gradient that is off by 1e-6 relative on one leaf, or a backward that drops one term, is reported."""
  z, meta = load_case("siglip_map_last_bias")
  loss, leaves = oracle_loss_and_leaves(z, meta)
  names = list(leaves)
  grads = dict(zip(names, torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)))
  grads = {n: (g if g is not None else torch.zeros_like(leaves[n])) for n, g in grads.items()}
  assert not mismatches(z, meta, grads)
  leaf = "img/Transformer/encoderblock_0/MlpBlock_0/Dense_0/kernel"
  off = dict(grads)
  off[leaf] = grads[leaf] * (1.0 + 1e-6)
  assert [b[0] for b in mismatches(z, meta, off)] == [leaf]
  # the temperature's gradient without the chain rule through exp (two_towers.py:81: t = exp(t'))
  off = dict(grads)
  off["t"] = grads["t"] / torch.exp(leaves["t"].detach())
  assert [b[0] for b in mismatches(z, meta, off)] == ["t"]


def test_every_leaf_of_the_reference_tree_is_probed():
  """The fixture's leaves are the executed reference's whole parameter tree (names from its own module definitions)."""
  for name in CASES:
    z, meta = load_case(name)
    assert sorted(meta["leaves"]) == sorted(k[len("param/"):] for k in z.files if k.startswith("param/"))
    assert len(meta["leaves"]) >= 25
  z, meta = load_case("siglip_map_last_bias")
  assert {"t", "b", "img/embedding/kernel", "txt/Embed_0/embedding", "img/MAPHead_0/probe"} <= set(meta["leaves"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/big_vision"), reason="the reference is not on this host")
def test_fixtures_are_what_the_reference_files_produce_today(tmp_path):
  env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
  subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "run_reference_gradients.py"), str(tmp_path)], check=True,
                 env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
  fresh = sorted(os.path.basename(p) for p in glob.glob(str(tmp_path / "refgrad_*.npz")))
  assert fresh == sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "refgrad_*.npz")))
  for f in fresh:
    a, b = np.load(tmp_path / f), np.load(os.path.join(GOLDEN, f))
    assert sorted(a.files) == sorted(b.files), f
    for k in a.files:
      assert np.array_equal(a[k], b[k]), (f, k)
