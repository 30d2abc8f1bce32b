"""Executes the reference's own tree / duration helpers and writes what they return as a golden fixture.

TEST INFRASTRUCTURE (oracle/): run by hand or by tests/test_reference_treeutils_cpu.py, never by the product.

  python oracle/run_reference_treeutils.py [out_dir]          (default tests/golden/)

`/root/reference/big_vision/utils.py` imported UNMODIFIED over `oracle/refshim/`: `tree_flatten_with_names` (:642-668),
`recover_tree` (:836-862), `tree_map_with_names` (:676-696), `make_mask_trees` (:1195-1212), `steps` (:1002-1067).  Pure Python on trees and dicts (the tree definition comes from the stand-in `jax.tree`, whose dict order is
jax's: sorted keys): the fixture is the reference's answer.  `reftreeutils.json`."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REFERENCE = os.environ.get("BV_REFERENCE_ROOT", "/root/reference")

# leaves are integers so that the trees are JSON; None = an absent leaf (frozen parameters' optimizer state)
TREES = {
    "nested": {"b": {"y": 2, "x": 1}, "a": 0, "c": {"d": {"e": 3}}},
    "with_none": {"img": {"kernel": 1, "bias": None}, "txt": None, "t": 2},
    "lists_and_tuples": {"chain": [{"count": 0}, [1, 2], {"mu": {"w": 3}}], "z": 4},
    "unsorted_numeric_keys": {"encoderblock_10": {"k": 1}, "encoderblock_2": {"k": 2}, "encoderblock_1": {"k": 3}},
    "flat": {"w": 5},
}
PATTERNS = {
    "nested": [["a", "b/.*"], [".*/x", "b/.*", ".*"], ["c/d/e"]],
    "unsorted_numeric_keys": [[".*_1/.*", ".*"], ["encoderblock_1.*"]],
}
STEPS = [
    ("warmup", {"warmup_steps": 7}, {}), ("warmup", {"warmup_steps": 0}, {}), ("warmup", {"warmup_examples": 100}, {"batch_size": 8}),
    ("warmup", {"warmup_examples": 1}, {"batch_size": 8}), ("warmup", {"warmup_epochs": 0.5}, {"batch_size": 8, "data_size": 100}),
    ("warmup", {"warmup_epochs": 0}, {"batch_size": 8, "data_size": 100}), ("warmup", {"warmup_percent": 0.25}, {"total_steps": 10}),
    ("warmup", {"warmup_percent": 0.0001}, {"total_steps": 10}), ("warmup", {"warmup_percent": 1.5}, {"total_steps": 10}),
    ("warmup", {"warmup_steps": 3, "warmup_epochs": 1}, {"batch_size": 8, "data_size": 100}), ("warmup", {}, {}),
    ("warmup", {}, {"default": 11}), ("warmup", {"warmup_epochs": 2}, {"batch_size": 8}), ("warmup", {"warmup_steps": -1}, {"default": 5}),
    ("total", {"total_epochs": 90}, {"batch_size": 1024, "data_size": 1281167}), ("log_training", {"log_training_steps": 50}, {}),
]


def _isolate_imports():
  drop = {REPO, ""}
  sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != REPO and p not in drop]
  sys.path.insert(0, REFERENCE)
  sys.path.insert(0, os.path.join(HERE, "refshim"))
  for m in list(sys.modules):
    if m == "big_vision" or m.startswith("big_vision.") or m in ("optax", "jax", "flax") or m.startswith(("jax.", "flax.", "optax.")):
      del sys.modules[m]


def attempt(fn):
  try:
    return {"value": fn()}
  except Exception as e:     # recorded: the product must fail the same way
    return {"error": type(e).__name__, "message": str(e)}


def run(u, to_bool=bool):
  """Everything the fixture holds, computed with the utils module `u` (the reference's here, the product's in the test)."""
  out = {"flatten": {}, "recover": {}, "map_with_names": {}, "masks": {}, "steps": []}
  for name, tree in TREES.items():
    flat = u.tree_flatten_with_names(tree)[0]
    out["flatten"][name] = [[k, v] for k, v in flat]
    keys, vals = zip(*flat)
    out["recover"][name] = u.recover_tree(keys, vals)
    out["map_with_names"][name] = u.tree_map_with_names(lambda n, v: f"{n}={v}", tree)
  for name, pats in PATTERNS.items():
    out["masks"][name] = [[u.tree_map(to_bool, m) for m in u.make_mask_trees(TREES[name], p)] for p in pats]
  for prefix, cfg, kw in STEPS:
    kw = dict(kw)
    out["steps"].append(attempt(lambda: u.steps(prefix, cfg, **kw)))
  return out


def main():
  out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "tests", "golden")
  _isolate_imports()
  import jax
  import big_vision.utils as u
  u.tree_map = jax.tree.map          # (`run` converts numpy bools of the mask trees through it)
  res = run(u)
  with open(os.path.join(out_dir, "reftreeutils.json"), "w") as f:
    json.dump(res, f, indent=1, sort_keys=True, default=lambda o: o.item() if hasattr(o, "item") else str(o))
  print({k: len(v) for k, v in res.items()}, sum("error" in s for s in res["steps"]), "steps cases raise")


if __name__ == "__main__":
  main()
