"""Executes the reference's own model files and writes what they produce as golden fixtures.

TEST INFRASTRUCTURE (oracle/): run by hand or by tests/test_reference_wiring_cpu.py, never by the product.

  python oracle/run_reference_wiring.py [out_dir]          (default tests/golden/)

`/root/reference/big_vision/models/vit.py`, `models/proj/image_text/text_transformer.py`, `two_towers.py`,
`models/common.py`, `utils.py` and (for the three loss functions, run_losses) `trainers/proj/image_text/
_deprecated_contrastive.py` are imported UNMODIFIED, from where they lie, over the stand-ins of
`oracle/refshim/` (jax / flax / absl / ... are not installed; see oracle/refshim/README.md for what that does and does
not pin).  For every case below the reference `Model` is built from a config, initialised, its parameters are jittered
(so that zero-initialised leaves - biases, cls token, zero-init head - carry signal) and applied to a seeded input.
One `refwiring_<case>.npz` per case: `param/<leaf name>` (the reference's '/'-joined names), `in/image`, `in/text`,
`out/<key>` for every array of the returned `out` dict (nested dicts flattened with '/'), `z/img`, `z/txt` (or `y`
for a single tower) and `meta` (JSON: the config, the ordered key lists).  float64 throughout."""
import json
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REFERENCE = os.environ.get("BV_REFERENCE_ROOT", "/root/reference")


def _isolate_imports():
  """`big_vision` must resolve to the REFERENCE, not to this repository's alias package of the same name."""
  drop = {REPO, ""}
  sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != REPO and p not in drop]
  sys.path.insert(0, REFERENCE)
  sys.path.insert(0, os.path.join(HERE, "refshim"))
  for m in list(sys.modules):
    if m == "big_vision" or m.startswith("big_vision."):
      del sys.modules[m]


TINY_IMG = dict(width=32, depth=2, mlp_dim=64, num_heads=2, patch_size=(8, 8))
TINY_TXT = dict(width=32, depth=2, mlp_dim=64, num_heads=2, vocab_size=50)
TINY_NAFLEX = dict(width=32, depth=2, mlp_dim=64, num_heads=2, nposemb=4, posemb="learn_2d(8)")
ONE_IMG, ONE_TXT = dict(TINY_IMG, depth=1), dict(TINY_TXT, depth=1)     # branch variants: one block keeps the fixtures small

CASES = {
    # --- models/vit.py:186-276: every pool_type / posemb / rep_size / head branch, loop and scan layouts
    "vit_gap_learn": ("vit", dict(num_classes=10, **ONE_IMG, pool_type="gap", posemb="learn")),
    "vit_gap_headinit": ("vit", dict(num_classes=10, **ONE_IMG, pool_type="gap", head_zeroinit=False)),
    "vit_map": ("vit", dict(num_classes=12, **TINY_IMG, pool_type="map")),
    "vit_map_nohead": ("vit", dict(num_classes=None, **ONE_IMG, pool_type="map")),
    "vit_tok": ("vit", dict(num_classes=None, **ONE_IMG, pool_type="tok")),
    "vit_0": ("vit", dict(num_classes=7, **ONE_IMG, pool_type="0")),
    "vit_sincos_rep": ("vit", dict(num_classes=10, **ONE_IMG, pool_type="gap", posemb="sincos2d", rep_size=True)),
    "vit_rep16": ("vit", dict(num_classes=10, **ONE_IMG, pool_type="gap", rep_size=16)),
    "vit_variant_mu16": ("vit", dict(num_classes=5, variant="mu/16", pool_type="map")),
    "vit_scan": ("vit", dict(num_classes=12, **TINY_IMG, pool_type="map", scan=True)),
    # --- train mode with dropout > 0 (vit.py:76,100,109,228): the masks the stand-in Dropout drew are stored (`mask/<path>`)
    "vit_dropout_tok": ("vit", dict(num_classes=10, **TINY_IMG, pool_type="tok", dropout=0.25), dict(train=True)),
    "vit_dropout_scan": ("vit", dict(num_classes=None, **TINY_IMG, pool_type="map", dropout=0.1, scan=True), dict(train=True)),
    # --- models/proj/image_text/text_transformer.py:55-99: every pool_type
    "txt_last": ("txt", dict(num_classes=16, **TINY_TXT, pool_type="last")),
    "txt_first": ("txt", dict(num_classes=16, **ONE_TXT, pool_type="first")),
    "txt_mean": ("txt", dict(num_classes=16, **ONE_TXT, pool_type="mean")),
    "txt_max": ("txt", dict(num_classes=16, **ONE_TXT, pool_type="max")),
    "txt_map": ("txt", dict(num_classes=16, **ONE_TXT, pool_type="map")),
    "txt_nohead": ("txt", dict(num_classes=0, **ONE_TXT, pool_type="last")),
    "txt_scan": ("txt", dict(num_classes=16, **TINY_TXT, pool_type="last", scan=True)),
    "txt_dropout": ("txt", dict(num_classes=16, **TINY_TXT, pool_type="last", dropout=0.2), dict(train=True)),
    # --- models/proj/image_text/two_towers.py:28-90
    "two_map_last_bias": ("two", dict(image=dict(**TINY_IMG, pool_type="map"), text=dict(**TINY_TXT), out_dim=(None, 32),
                                      temperature_init=10.0, bias_init=-10.0)),
    "two_tok_int_outdim": ("two", dict(image=dict(**ONE_IMG, pool_type="tok"), text=dict(**ONE_TXT), out_dim=24,
                                       temperature_init=2.0)),
    "two_image_only": ("two", dict(image=dict(**ONE_IMG, pool_type="map"), text=dict(**ONE_TXT), out_dim=(None, 32),
                                   temperature_init=10.0, bias_init=-2.71), dict(text=False)),
    "two_text_only": ("two", dict(image=dict(**ONE_IMG, pool_type="map"), text=dict(**ONE_TXT), out_dim=(None, 32),
                                  temperature_init=10.0, bias_init=-2.71), dict(image=False)),
    "two_scan": ("two", dict(image=dict(**TINY_IMG, pool_type="map", scan=True), text=dict(**TINY_TXT, scan=True),
                             out_dim=(None, 32), temperature_init=10.0, bias_init=-10.0)),
    # --- models/proj/image_text/naflex_vit.py:196-285: pre-patchified ragged inputs (patches, ptype, yabs, xabs), the
    # learned position grid resized per example, key-padding masks, masked pooling
    "naflex_gap": ("naflex", dict(num_classes=7, **TINY_NAFLEX, pool_type="gap")),
    "naflex_map_patchln": ("naflex", dict(num_classes=None, **TINY_NAFLEX, pool_type="map", patchln_pre=True, patchln_post=True)),
    "naflex_max_rep": ("naflex", dict(num_classes=5, **TINY_NAFLEX, pool_type="max", rep_size=16)),
    "naflex_none_scan": ("naflex", dict(num_classes=None, **TINY_NAFLEX, pool_type="none", scan=True)),
    "naflex_gap_holes": ("naflex", dict(num_classes=7, **TINY_NAFLEX, pool_type="gap"), dict(holes=True)),
    "two_naflex": ("two", dict(image=dict(**TINY_NAFLEX, pool_type="map"), text=dict(**TINY_TXT), image_model="proj.image_text.naflex_vit",
                               out_dim=(None, 32), temperature_init=10.0, bias_init=-10.0), dict(naflex=True)),
    "two_dropout": ("two", dict(image=dict(**TINY_IMG, pool_type="map", dropout=0.1), text=dict(**TINY_TXT, dropout=0.3),
                                out_dim=(None, 32), temperature_init=10.0, bias_init=-10.0), dict(train=True)),
}


def _flatten(tree, prefix=""):
  out = []
  for k, v in tree.items():
    name = f"{prefix}{k}"
    if isinstance(v, dict):
      out.extend(_flatten(v, name + "/"))
    elif v is not None:
      out.append((name, v))
  return out


def _jitter(params, np):
  def walk(t, prefix):
    for k in t:
      name = f"{prefix}{k}"
      if isinstance(t[k], dict):
        walk(t[k], name + "/")
      else:
        g = np.random.default_rng([7, zlib.crc32(name.encode())])
        t[k] = np.asarray(t[k], np.float64) + 0.05 * g.standard_normal(np.shape(t[k]))
  walk(params, "")


def naflex_inputs(g, holes=False):
  """(patches, ptype, yabs, xabs) of two examples with 12 token slots: a 3 x 4 grid that fills them and a 2 x 3 grid with
  six padding slots (ptype 0, coordinates 0) - at the end, or (holes) scattered between the patches."""
  import numpy as np
  n, L, pd = 2, 12, 48
  patches = g.uniform(-1.0, 1.0, (n, L, pd))
  ptype = np.zeros((n, L), np.int32)
  yabs = np.zeros((n, L), np.int32)
  xabs = np.zeros((n, L), np.int32)
  slots = [list(range(12)), [0, 2, 3, 6, 9, 11] if holes else list(range(6))]
  for i, (gh, gw) in enumerate(((3, 4), (2, 3))):
    for t, s_ in enumerate(slots[i]):
      ptype[i, s_], yabs[i, s_], xabs[i, s_] = 1, t // gw, t % gw
  patches[ptype == 0] = 0.0
  return patches, ptype, yabs, xabs


def run_case(name, out_dir):
  import jax
  import numpy as np
  kind, cfg, *rest = CASES[name]
  use = dict(image=True, text=True, train=False, holes=False, naflex=False)
  use.update(rest[0] if rest else {})
  train, holes, two_naflex = use.pop("train"), use.pop("holes"), use.pop("naflex")
  tkw = dict(train=True, rngs={"dropout": jax.random.PRNGKey(5)}) if train else dict(train=False)
  g = np.random.default_rng([11, zlib.crc32(name.encode())])
  res = 32
  image = g.uniform(-1.0, 1.0, (2, res, res, 3))
  text = g.integers(2, 50, (2, 8)).astype(np.int32)
  text[:, 6:] = 1                                   # sticky EOS / padding id 1
  arrays, meta = {}, {"case": name, "kind": kind, "config": cfg}
  if kind == "vit":
    from big_vision.models import vit
    model = vit.Model(**cfg)
    params = model.init(jax.random.PRNGKey(0), image)["params"]
    _jitter(params, np)
    y, out = model.apply({"params": params}, image, **tkw)
    arrays.update({"in/image": image, "y": y})
  elif kind == "naflex":
    from big_vision.models.proj.image_text import naflex_vit
    nf = naflex_inputs(g, holes)
    model = naflex_vit.Model(**cfg)
    params = model.init(jax.random.PRNGKey(0), nf)["params"]
    _jitter(params, np)
    y, out = model.apply({"params": params}, nf, **tkw)
    arrays.update({"in/patches": nf[0], "in/ptype": nf[1], "in/yabs": nf[2], "in/xabs": nf[3], "y": y})
  elif kind == "txt":
    from big_vision.models.proj.image_text import text_transformer
    model = text_transformer.Model(**cfg)
    params = model.init(jax.random.PRNGKey(0), text)["params"]
    _jitter(params, np)
    y, out = model.apply({"params": params}, text, **tkw)
    arrays.update({"in/text": text, "y": y})
  else:
    from big_vision.models.proj.image_text import two_towers
    model = two_towers.Model(**cfg)
    if two_naflex:     # the image tower takes (patches, ptype, yabs, xabs)
      image = naflex_inputs(g)
    params = model.init(jax.random.PRNGKey(0), image, text)["params"]      # both towers exist in the tree
    _jitter(params, np)
    im = image if use["image"] else None
    tx = text if use["text"] else None
    zimg, ztxt, out = model.apply({"params": params}, im, tx, **(tkw if train else {}))
    if im is not None and two_naflex:
      arrays.update({"in/patches": im[0], "in/ptype": im[1], "in/yabs": im[2], "in/xabs": im[3], "z/img": zimg})
    elif im is not None:
      arrays.update({"in/image": image, "z/img": zimg})
    if tx is not None:
      arrays.update({"in/text": text, "z/txt": ztxt})
    meta["inputs"] = use
  if train:
    import flax.linen as nn
    meta["dropout_sites"] = [p for p, _ in nn.last_dropout_masks]          # module paths in call order (`#i`: scan index)
    arrays.update({f"mask/{p}": np.asarray(k, np.uint8) for p, k in nn.last_dropout_masks})
  flat_p, flat_o = _flatten(params), _flatten(out)
  meta["param_names"] = [n for n, _ in flat_p]
  meta["param_shapes"] = {n: list(np.shape(v)) for n, v in flat_p}
  meta["out_keys"] = [n for n, _ in flat_o]
  arrays.update({f"param/{n}": np.asarray(v, np.float64) for n, v in flat_p})
  arrays.update({f"out/{n}": np.asarray(v, np.float64) for n, v in flat_o})
  arrays["meta"] = np.frombuffer(json.dumps(meta, sort_keys=True).encode(), np.uint8)
  np.savez_compressed(os.path.join(out_dir, f"refwiring_{name}.npz"), **arrays)
  return meta


def scan_roundtrip(out_dir):
  """The reference's own layout converters (vit.py:363-405) on the reference's own scan model: applying the LOOP model
  to scan_to_pyloop(params of the scan model) must give the scan model's output.  Written next to the fixtures."""
  import jax
  import numpy as np
  from big_vision.models import vit
  cfg = dict(CASES["vit_scan"][1])
  image = np.random.default_rng(5).uniform(-1, 1, (2, 32, 32, 3))
  scan_model = vit.Model(**cfg)
  p_scan = scan_model.init(jax.random.PRNGKey(0), image)["params"]
  _jitter(p_scan, np)
  y_scan, _ = scan_model.apply({"params": p_scan}, image)
  cfg["scan"] = False
  loop_model = vit.Model(**cfg)
  p_loop = vit.scan_to_pyloop(p_scan)
  y_loop, _ = loop_model.apply({"params": p_loop}, image)
  back = vit.pyloop_to_scan(p_loop)
  fb, fs = dict(_flatten(back)), dict(_flatten(p_scan))
  same = fb.keys() == fs.keys() and all(np.array_equal(fb[k], fs[k]) for k in fs)
  return {"max_abs_diff_scan_vs_loop": float(np.max(np.abs(y_scan - y_loop))), "pyloop_to_scan_inverts": bool(same),
          "loop_names": [n for n, _ in _flatten(p_loop)]}


def _stub_reference_modules():
  """`_deprecated_contrastive.py` imports three sibling REFERENCE modules at its top that drag in tensorflow / tf.data /
  optax (`big_vision.evaluators.common`, `big_vision.input_pipeline`, `big_vision.optax`).  None of them is touched by
  the loss functions executed here, so empty modules stand in for them (the trainer file itself stays unmodified)."""
  import types
  import big_vision
  for name in ("big_vision.evaluators", "big_vision.evaluators.common", "big_vision.input_pipeline", "big_vision.optax"):
    if name not in sys.modules:
      m = types.ModuleType(name)
      m.__path__ = []
      sys.modules[name] = m
      parent, _, leaf = name.rpartition(".")
      setattr(sys.modules[parent], leaf, m)


LOSS_WORLDS = (1, 2, 4)


def run_losses(out_dir):
  """trainers/proj/image_text/_deprecated_contrastive.py:80-200 - softmax_loss, sigmoid_loss, chunked_sigmoid_loss - the
  reference's own per-device functions, run as N virtual devices (jax.lax collectives emulated, refshim/jax/lax.py) on
  seeded unit-norm embeddings: per device the loss and every entry of its measurement dict."""
  import jax
  import numpy as np
  _stub_reference_modules()
  from big_vision.trainers.proj.image_text import _deprecated_contrastive as C
  g = np.random.default_rng(123)
  B, E = 16, 24
  zimg = g.standard_normal((B, E)); zimg /= np.linalg.norm(zimg, axis=1, keepdims=True)
  ztxt = zimg * 0.6 + 0.8 * g.standard_normal((B, E)); ztxt /= np.linalg.norm(ztxt, axis=1, keepdims=True)
  t, b = 7.5, -4.25
  arrays = {"zimg": zimg, "ztxt": ztxt, "t": np.float64(t), "b": np.float64(b)}
  keys = {}
  for world in LOSS_WORLDS:
    n = B // world
    shards = [(zimg[r * n:(r + 1) * n], ztxt[r * n:(r + 1) * n]) for r in range(world)]
    runs = {
        "sigmoid": lambda zi, zt: C.sigmoid_loss(zi, zt, t, bias=b),
        "chunked_sigmoid": lambda zi, zt: C.chunked_sigmoid_loss(zi, zt, t, bias=b),
        "softmax": lambda zi, zt: C.softmax_loss(zi, zt, t),
    }
    for kind, fn in runs.items():
      res = jax.lax.spmd(fn, world, shards)
      for r, (l, extras) in enumerate(res):
        arrays[f"{kind}/w{world}/r{r}/loss"] = np.float64(l)
        for k, v in extras.items():
          arrays[f"{kind}/w{world}/r{r}/{k}"] = np.float64(v)
        keys[kind] = sorted(extras)
  arrays["meta"] = np.frombuffer(json.dumps({"worlds": LOSS_WORLDS, "extras": keys, "B": B, "E": E}, sort_keys=True).encode(), np.uint8)
  np.savez_compressed(os.path.join(out_dir, "refwiring_losses.npz"), **arrays)
  return keys


def model_fields():
  """The dataclass fields (name, default) of the reference's model classes, in definition order: the constructor surface a
  config reaches through `Model(**config.model)` (8b contract)."""
  import flax.linen as nn
  from big_vision.models import vit
  from big_vision.models.proj.image_text import naflex_vit, text_transformer, two_towers
  out = {}
  for key, cls in (("vit", vit._Model), ("proj.image_text.text_transformer", text_transformer._Model),
                   ("proj.image_text.two_towers", two_towers.Model), ("proj.image_text.naflex_vit", naflex_vit._Model)):
    out[key] = [[k, "<required>" if d is nn._MISSING else (list(d) if isinstance(d, tuple) else d)] for k, d in nn._fields_of(cls)]
  return out


def main():
  out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "tests", "golden")
  if not os.path.isdir(os.path.join(REFERENCE, "big_vision")):
    raise SystemExit(f"{REFERENCE}/big_vision not found: the reference's files are needed to run them")
  _isolate_imports()
  os.makedirs(out_dir, exist_ok=True)
  import big_vision
  assert os.path.abspath(os.path.dirname(big_vision.__file__ or big_vision.__path__[0])).startswith(REFERENCE), big_vision
  summary = {}
  for name in CASES:
    meta = run_case(name, out_dir)
    summary[name] = {"params": len(meta["param_names"]), "out_keys": len(meta["out_keys"])}
    print(f"{name:22s} {len(meta['param_names']):3d} parameters, {len(meta['out_keys']):3d} out entries", flush=True)
  summary["losses"] = run_losses(out_dir)
  print("losses:", summary["losses"])
  summary["scan_roundtrip"] = scan_roundtrip(out_dir)
  print("scan round trip:", {k: v for k, v in summary["scan_roundtrip"].items() if k != "loop_names"})
  summary["model_fields"] = model_fields()
  with open(os.path.join(out_dir, "refwiring_summary.json"), "w") as f:
    json.dump(summary, f, indent=1, sort_keys=True)


if __name__ == "__main__":
  main()
