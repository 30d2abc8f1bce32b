"""Executes the reference's own classification losses and mixup and writes what they return as a golden fixture.

TEST INFRASTRUCTURE (oracle/): run by hand or by tests/test_reference_utils_cpu.py, never by the product.

  python oracle/run_reference_utils.py [out_dir]          (default tests/golden/)

`/root/reference/big_vision/utils.py` is imported UNMODIFIED over the stand-ins of `oracle/refshim/` and these functions of
the classification step (train.py:281-299) are executed: `sigmoid_xent` (:236-243), `softmax_xent` (:276-281, with and
without `kl`), `bidirectional_contrastive_loss` (:246-273, no mask), `get_mixup` / `mixup` (:1146-1159).  `jax.nn.log_sigmoid` / `log_softmax` and `jnp.*` are the stand-ins' (numpy float64, stable forms): what is
pinned is which of them the reference composes and how - sums over the class axis, mean over the batch, `max(a, 1 - a)`,
`roll(shift=1, axis=0)` - not their arithmetic.  The mixup coefficient comes from the stand-in generator (JAX's random
stream is not reproduced); the fixture stores it (`mixup/a`, recovered from the mixed arrays) so that other implementations
can be fed the same one.

`refutils.npz`: inputs (`<case>/in/...`) and outputs (`<case>/out/...`) in float64."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REFERENCE = os.environ.get("BV_REFERENCE_ROOT", "/root/reference")


def _isolate_imports():
  drop = {REPO, ""}
  sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != REPO and p not in drop]
  sys.path.insert(0, REFERENCE)
  sys.path.insert(0, os.path.join(HERE, "refshim"))
  for m in list(sys.modules):
    if m == "big_vision" or m.startswith("big_vision.") or m in ("optax", "jax", "flax") or m.startswith(("jax.", "flax.", "optax.")):
      del sys.modules[m]


def inputs():
  """Seeded inputs (numpy only; the tests rebuild nothing: everything is stored in the fixture)."""
  g = np.random.default_rng(41)
  n, C = 12, 10
  logits = g.normal(0.0, 3.0, (n, C))
  logits[0] *= 20.0                       # a saturated row: the stable forms matter
  hard = np.eye(C)[g.integers(0, C, n)]
  soft = g.dirichlet(np.ones(C), n)
  multi = (g.uniform(size=(n, C)) < 0.3).astype(np.float64)     # multi-label targets of sigmoid_xent
  z = g.normal(size=(8, 16))
  zimg = z / np.linalg.norm(z, axis=1, keepdims=True)
  z = g.normal(size=(8, 16))
  ztxt = z / np.linalg.norm(z, axis=1, keepdims=True)
  images = g.uniform(-1.0, 1.0, (6, 4, 4, 3))
  labels = np.eye(C)[g.integers(0, C, 6)]
  return dict(logits=logits, hard=hard, soft=soft, multi=multi, zimg=zimg, ztxt=ztxt, images=images, labels=labels)


def main():
  out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "tests", "golden")
  x = inputs()
  _isolate_imports()
  import jax
  import big_vision.utils as u
  a = {f"in/{k}": v for k, v in x.items()}
  for lab in ("hard", "soft", "multi"):
    a[f"sigmoid_xent/{lab}/mean"] = np.asarray(u.sigmoid_xent(logits=x["logits"], labels=x[lab]))
    a[f"sigmoid_xent/{lab}/per_example"] = np.asarray(u.sigmoid_xent(logits=x["logits"], labels=x[lab], reduction=False))
  for lab in ("hard", "soft"):
    a[f"softmax_xent/{lab}/mean"] = np.asarray(u.softmax_xent(logits=x["logits"], labels=x[lab]))
    a[f"softmax_xent/{lab}/per_example"] = np.asarray(u.softmax_xent(logits=x["logits"], labels=x[lab], reduction=False))
    a[f"softmax_xent/{lab}/kl_mean"] = np.asarray(u.softmax_xent(logits=x["logits"], labels=x[lab], kl=True))
  for red in (False, True):
    l, extra = u.bidirectional_contrastive_loss(x["zimg"], x["ztxt"], 7.5, reduction=red)
    a[f"bidirectional/{'mean' if red else 'per_example'}"] = np.asarray(l, np.float64)
    a[f"bidirectional/ncorrect_{'mean' if red else 'per_example'}"] = np.asarray(extra["ncorrect"], np.float64)
  # mixup as train.py:281-289 calls it: positional things, the rng comes back first
  rng = jax.random.PRNGKey(7)
  out = u.get_mixup(rng, 0.2)(x["images"], x["labels"])
  _, (images, labels), more = out[0], out[1], out[2]
  assert more == {}
  a["mixup/images"], a["mixup/labels"] = np.asarray(images), np.asarray(labels)
  # the legacy spelling with keyword things (utils.py:1158-1159)
  out2 = u.mixup(rng, x["images"], p=0.2, labels=x["labels"])
  a["mixup_kw/images"], a["mixup_kw/labels"] = np.asarray(out2[1][0]), np.asarray(out2[2]["labels"])
  # the coefficient, recovered from one element: y = a x + (1 - a) roll(x)
  xi, ri, yi = x["images"].reshape(6, -1)[:, 0], np.roll(x["images"], 1, axis=0).reshape(6, -1)[:, 0], images.reshape(6, -1)[:, 0]
  coef = float(np.median((yi - ri) / (xi - ri)))
  assert 0.5 <= coef <= 1.0
  a["mixup/a"] = np.asarray(coef)
  a["meta"] = np.frombuffer(json.dumps(dict(temperature=7.5, mixup_p=0.2), sort_keys=True).encode(), np.uint8)
  np.savez_compressed(os.path.join(out_dir, "refutils.npz"), **a)
  print(len(a) - 1, "arrays; mixup a =", coef)


if __name__ == "__main__":
  main()
