"""Executes the reference's own sharding inference and writes what it produces as a golden fixture.

TEST INFRASTRUCTURE (oracle/): run by hand or by tests/test_reference_sharding_cpu.py, never by the product.

  python oracle/run_reference_sharding.py [out_dir]          (default tests/golden/)

`/root/reference/big_vision/sharding.py` (`infer_sharding` :38-71, the `replicate` :83-101 and `fsdp` :104-139 rules),
`big_vision/pp/registry.py` (the `name(args)` parser the rules are looked up through) and `big_vision/utils.py`
(`make_mask_trees`, `tree_flatten_with_names`) are imported UNMODIFIED over the stand-ins of `oracle/refshim/`
(`jax.sharding.{PartitionSpec, NamedSharding}` are plain records there; a mesh is a name -> size mapping).  This is
pure host logic - shapes in, partition specs out - so NOTHING of it is restated: the fixture is the reference's answer.

`refsharding.json`: for every (tree, strategy, mesh size) case the spec of every leaf - in the order of the tree's sorted
leaf names, each as the digits of its sharded axes ("-" = replicated; the mesh axis is always 'data') - or the error the
reference raises.  Trees: the parameter trees of the model fixtures (names and shapes from
tests/golden/refwiring_*.npz) and the REAL ViT-B/16 + text-B shapes (placeholders with .shape / .dtype only)."""
import json
import os
import sys

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


def b16_shapes():
  """Leaf names and shapes of SigLIP ViT-B/16 (MAP head) + text-B (vocab 32k, 64 tokens), the reference's naming (pinned at
  toy width by tests/golden/refwiring_two_map_last_bias.npz)."""
  D, H, Dh, M, E = 768, 12, 64, 3072, 768
  out = {}

  def block(p):
    out.update({f"{p}/LayerNorm_0/scale": (D,), f"{p}/LayerNorm_0/bias": (D,), f"{p}/LayerNorm_1/scale": (D,), f"{p}/LayerNorm_1/bias": (D,)})
    for n in ("query", "key", "value"):
      out.update({f"{p}/MultiHeadDotProductAttention_0/{n}/kernel": (D, H, Dh), f"{p}/MultiHeadDotProductAttention_0/{n}/bias": (H, Dh)})
    out.update({f"{p}/MultiHeadDotProductAttention_0/out/kernel": (H, Dh, D), f"{p}/MultiHeadDotProductAttention_0/out/bias": (D,)})
    out.update({f"{p}/MlpBlock_0/Dense_0/kernel": (D, M), f"{p}/MlpBlock_0/Dense_0/bias": (M,),
                f"{p}/MlpBlock_0/Dense_1/kernel": (M, D), f"{p}/MlpBlock_0/Dense_1/bias": (D,)})

  out.update({"img/embedding/kernel": (16, 16, 3, D), "img/embedding/bias": (D,), "img/pos_embedding": (1, 196, D)})
  for i in range(12):
    block(f"img/Transformer/encoderblock_{i}")
  out.update({"img/Transformer/encoder_norm/scale": (D,), "img/Transformer/encoder_norm/bias": (D,), "img/MAPHead_0/probe": (1, 1, D)})
  block("img/MAPHead_0")     # (same leaf shapes: attention + LayerNorm_0 + MlpBlock_0; the extra LayerNorm_1 is dropped below)
  for k in [k for k in out if k.startswith("img/MAPHead_0/LayerNorm_1")]:
    del out[k]
  out.update({"img/head/kernel": (D, E), "img/head/bias": (E,)})
  out.update({"txt/Embed_0/embedding": (32000, D), "txt/pos_embedding": (1, 64, D)})
  for i in range(12):
    block(f"txt/Encoder_0/encoderblock_{i}")
  out.update({"txt/Encoder_0/encoder_norm/scale": (D,), "txt/Encoder_0/encoder_norm/bias": (D,), "txt/head/kernel": (D, E), "txt/head/bias": (E,)})
  out.update({"t": (1,), "b": (1,)})
  return out


FSDP0 = "fsdp(axis='data', min_size_to_shard_mb=0)"
STRATEGIES = {
    "replicate": [(".*", "replicate")],
    "fsdp_default": [(".*", "fsdp(axis='data')")],                       # 4 MiB threshold
    "fsdp_all": [(".*", FSDP0)],
    "fsdp_1kb": [(".*", "fsdp(axis='data', min_size_to_shard_mb=0.001)")],
    "fsdp_img_only": [("img/.*", FSDP0), (".*", "replicate")],
    "first_match_wins": [(".*/bias", "replicate"), ("txt/.*", FSDP0), (".*/kernel", "fsdp(axis='data', min_size_to_shard_mb=0.01)")],
    "unmatched_stay_replicated": [("txt/head/.*", FSDP0)],
    "fsdp_twice": [(".*", FSDP0 + "|" + FSDP0)],                         # the second application takes the next free axis
    "fsdp_then_replicate": [(".*", FSDP0 + "|replicate")],               # the reference raises (inconsistent instructions)
}
TREES = ("refwiring_two_map_last_bias", "refwiring_two_scan", "b16")
MESHES = (1, 2, 8, 3)


def _code(spec):
  """A spec as the digits of its sharded axes ("-" = replicated): (None, 'data') -> "1"."""
  return "".join(str(i) for i, a in enumerate(spec) if a is not None) or "-"


class _Leaf:
  """What the rules read of a parameter: .shape, .ndim, .dtype.itemsize."""

  def __init__(self, shape):
    import numpy as np
    self.shape, self.ndim, self.dtype = tuple(shape), len(shape), np.dtype(np.float32)


def _nest(flat):
  tree = {}
  for k, v in flat.items():
    node = tree
    *parents, last = k.split("/")
    for p in parents:
      node = node.setdefault(p, {})
    node[last] = v
  return tree


def tree_shapes(name):
  import numpy as np
  if name == "b16":
    return b16_shapes()
  z = np.load(os.path.join(REPO, "tests", "golden", f"{name}.npz"))
  return {k[len("param/"):]: tuple(z[k].shape) for k in z.files if k.startswith("param/")}


def main():
  out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "tests", "golden")
  shapes = {t: tree_shapes(t) for t in TREES}
  _isolate_imports()
  import jax
  import big_vision.sharding as bv_sharding
  import big_vision.utils as u
  out = {"trees": {t: {k: list(v) for k, v in s.items()} for t, s in shapes.items()}, "strategies": STRATEGIES, "cases": {}}
  for t in TREES:
    params = _nest({k: _Leaf(s) for k, s in shapes[t].items()})
    for sname, strategy in STRATEGIES.items():
      for n in MESHES:
        mesh = jax.sharding.Mesh({"data": n})
        try:
          sh = bv_sharding.infer_sharding(params, strategy, mesh)
          flat = dict(u.tree_flatten_with_names(sh)[0])
          assert all(a in (None, "data") for v in flat.values() for a in v.spec)
          assert all(len(v.spec) == len(shapes[t][k]) for k, v in flat.items())
          res = {"sharded_axes": [_code(flat[k].spec) for k in sorted(shapes[t])]}
        except Exception as e:     # recorded: the product must raise the same kind of error
          res = {"error": type(e).__name__, "message": str(e)[:200]}
        out["cases"][f"{t}|{sname}|{n}"] = res
  with open(os.path.join(out_dir, "refsharding.json"), "w") as f:
    json.dump(out, f, sort_keys=True, separators=(",", ":"))
  n_err = sum("error" in c for c in out["cases"].values())
  print(len(out["cases"]), "cases,", n_err, "raise")


if __name__ == "__main__":
  main()
