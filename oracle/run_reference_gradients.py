"""Directional derivatives of the reference's OWN step losses, taken by finite differences of the executed reference.

TEST INFRASTRUCTURE (oracle/): run by hand or by tests/test_reference_gradients_cpu.py, never by the product.

  python oracle/run_reference_gradients.py [out_dir]          (default tests/golden/)

What `jax.value_and_grad(loss_fn)` differentiates in the reference's trainers is a composition of the reference's own
files: `model.apply` (models/proj/image_text/two_towers.py -> vit.py / text_transformer.py) followed by the sigmoid loss
(trainers/proj/image_text/_deprecated_contrastive.py:308-330 `loss_fn` -> `sigmoid_loss` :117-160; on one device the same
function as siglip.py:287-308), or `model.apply` followed by `getattr(u, config.loss)` (train.py:281-299 -> utils.py
`sigmoid_xent` / `softmax_xent`).  Those files are imported UNMODIFIED over the stand-ins of `oracle/refshim/` (numpy
float64; see its README for what that does and does not pin) exactly as in `run_reference_wiring.py`, the loss F(params)
is evaluated as the reference composes it, and for EVERY parameter leaf the derivative of F along one seeded direction v
is taken by a 4th-order central difference in float64:

    dF/dv = ( -F(p + 2hv) + 8 F(p + hv) - 8 F(p - hv) + F(p - 2hv) ) / (12 h)

(the same with 2h is stored beside it: their difference bounds the truncation error).  No automatic differentiation is
involved on this side, so the fixture is "the gradient of the executed reference function" independent of any backward
pass written here: tests hold `bv_oracle`'s autograd gradient - the checker of every gradient test of the product - to
<grad[leaf], v> == dF/dv, leaf by leaf.

`refgrad_<case>.npz`: `param/<leaf>` (float32-representable values), `in/...`, `dir/<leaf>` (values exactly representable in float16, so the file stays
small), `dd/<leaf>` (h), `dd2/<leaf>` (2h), `loss`, `meta`."""
import json
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import run_reference_wiring as RW  # noqa: E402  (case configs, import isolation, parameter jitter)

H = 1e-3

CASES = {
    # (kind, model config, loss)
    "siglip_map_last_bias": ("two", RW.CASES["two_map_last_bias"][1], "sigmoid"),
    "siglip_scan": ("two", RW.CASES["two_scan"][1], "sigmoid"),
    "contrastive_tok_softmax": ("two", RW.CASES["two_tok_int_outdim"][1], "softmax"),     # no bias parameter: config.loss_fn = "softmax"
    "cls_rep16_sigmoid_xent": ("vit", RW.CASES["vit_rep16"][1], "sigmoid_xent"),
    "cls_map_softmax_xent": ("vit", RW.CASES["vit_map"][1], "softmax_xent"),
}
N = 4   # pairs / images per case


def _leaves(tree, prefix=""):
  for k in tree:
    name = f"{prefix}{k}"
    if isinstance(tree[k], dict):
      yield from _leaves(tree[k], name + "/")
    else:
      yield name, tree, k


def direction(name, shape, np):
  """Seeded unit-scale direction with float16-representable entries (stored exactly, compresses well)."""
  g = np.random.default_rng([23, zlib.crc32(name.encode())])
  v = g.standard_normal(shape) / max(1.0, float(np.sqrt(np.prod(shape, dtype=np.float64))))
  return v.astype(np.float16).astype(np.float64)


def run_case(name, out_dir):
  import jax
  import numpy as np
  kind, cfg, loss_name = CASES[name]
  g = np.random.default_rng([29, zlib.crc32(name.encode())])
  image = g.uniform(-1.0, 1.0, (N, 32, 32, 3))
  text = g.integers(2, 50, (N, 8)).astype(np.int32)
  text[:, 6:] = 1
  arrays = {"in/image": image}
  if kind == "two":
    from big_vision.models.proj.image_text import two_towers
    RW._stub_reference_modules()
    from big_vision.trainers.proj.image_text import _deprecated_contrastive as C
    model = two_towers.Model(**cfg)
    params = model.init(jax.random.PRNGKey(0), image, text)["params"]
    arrays["in/text"] = text

    def F(p):
      # _deprecated_contrastive.py:316-327 on ONE device (sigmoid: = siglip.py:288-306): apply, then the loss config.loss_fn names
      zimg, ztxt, extras = model.apply({"params": p}, image, text, train=False)
      if loss_name == "sigmoid":
        per_device = lambda zi, zt: C.sigmoid_loss(zi, zt, extras["t"], bias=extras["b"])
      else:
        per_device = lambda zi, zt: C.softmax_loss(zi, zt, extras["t"])
      (l, _), = jax.lax.spmd(per_device, 1, [(zimg, ztxt)])
      return float(l)
  else:
    from big_vision.models import vit
    import big_vision.utils as u
    model = vit.Model(**cfg)
    params = model.init(jax.random.PRNGKey(0), image)["params"]
    nc = cfg["num_classes"]
    labels = (g.uniform(size=(N, nc)) < 0.3).astype(np.float64)
    labels[np.arange(N), g.integers(0, nc, N)] = 1.0
    if loss_name == "softmax_xent":
      labels = labels / labels.sum(-1, keepdims=True)
    arrays["in/labels"] = labels

    def F(p):
      # train.py:281-299: logits, _ = model.apply(...); getattr(u, config.loss)(logits=logits, labels=labels)
      logits, _ = model.apply({"params": p}, image, train=False)
      return float(np.mean(getattr(u, loss_name)(logits=logits, labels=labels)))
  RW._jitter(params, np)
  for _, node, key in list(_leaves(params)):       # float32-representable values: stored as float32, used as float64
    node[key] = np.asarray(node[key], np.float32).astype(np.float64)
  meta = {"case": name, "kind": kind, "config": cfg, "loss": loss_name, "h": H, "leaves": []}
  arrays["loss"] = np.float64(F(params))
  for leaf, node, key in list(_leaves(params)):
    base = np.asarray(node[key], np.float64)
    v = direction(leaf, base.shape, np)

    def at(s):
      node[key] = base + s * v
      return F(params)

    try:
      f = {s: at(s * H) for s in (-4, -2, -1, 1, 2, 4)}
    finally:
      node[key] = base
    dd = (-f[2] + 8 * f[1] - 8 * f[-1] + f[-2]) / (12 * H)
    dd2 = (-f[4] + 8 * f[2] - 8 * f[-2] + f[-4]) / (24 * H)
    arrays[f"param/{leaf}"], arrays[f"dir/{leaf}"] = base.astype(np.float32), v.astype(np.float16)
    arrays[f"dd/{leaf}"], arrays[f"dd2/{leaf}"] = np.float64(dd), np.float64(dd2)
    meta["leaves"].append(leaf)
  arrays["meta"] = np.frombuffer(json.dumps(meta, sort_keys=True).encode(), np.uint8)
  np.savez_compressed(os.path.join(out_dir, f"refgrad_{name}.npz"), **arrays)
  return meta, arrays


def main():
  out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "tests", "golden")
  if not os.path.isdir(os.path.join(RW.REFERENCE, "big_vision")):
    raise SystemExit(f"{RW.REFERENCE}/big_vision not found: the reference's files are needed to run them")
  RW._isolate_imports()
  os.makedirs(out_dir, exist_ok=True)
  import big_vision
  assert os.path.abspath(os.path.dirname(big_vision.__file__ or big_vision.__path__[0])).startswith(RW.REFERENCE), big_vision
  import numpy as np
  for name in CASES:
    meta, arrays = run_case(name, out_dir)
    worst = max(abs(float(arrays[f"dd/{l}"]) - float(arrays[f"dd2/{l}"])) for l in meta["leaves"])
    scale = float(np.median([abs(float(arrays[f"dd/{l}"])) for l in meta["leaves"]]))
    print(f"{name:26s} loss {float(arrays['loss']):.6f}  {len(meta['leaves']):3d} leaves, median |dF/dv| {scale:.3e}, worst |h - 2h| {worst:.2e}", flush=True)


if __name__ == "__main__":
  main()
