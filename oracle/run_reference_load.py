"""Executes the reference's own checkpoint loaders and writes what they return as a golden fixture.

TEST INFRASTRUCTURE (oracle/): run by hand or by tests/test_reference_load_cpu.py, never by the product.

  python oracle/run_reference_load.py [out_dir]          (default tests/golden/)

`/root/reference/big_vision/models/vit.py` (`load` :408-433 with `fix_old_checkpoints` :324-361, `pyloop_to_scan` /
`scan_to_pyloop` :364-405, `resample_posemb` :306-321), `models/proj/image_text/text_transformer.py` (`load` :107-119),
`models/proj/image_text/two_towers.py` (`load` :93-137), `models/common.py` (`merge_params` :24-92) and `utils.py`
(`load_params` :170-227, `npload`, `tree_get`, `recover_tree`) are imported UNMODIFIED over the stand-ins of
`oracle/refshim/` and run on .npz checkpoints written here.  This path is host code on numpy arrays - file I/O, renames,
stacking, a `scipy.ndimage.zoom` (the REAL scipy: it is installed) - so nothing of it is restated: the fixture is the
reference's answer.

Every scenario of `SCENARIOS` is a pure function of its name (numpy only - tests/test_reference_load_cpu.py imports this
module for them and feeds the SAME checkpoints to the product's loaders): `build(name, tmp_dir)` writes the checkpoint
file(s) and returns what to call.  `refload.npz`: `<scenario>/<leaf name>` = the returned tree, `meta` = per scenario the
ordered leaf names or the error (type, message)."""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REFERENCE = os.environ.get("BV_REFERENCE_ROOT", "/root/reference")

D, H, M = 8, 2, 16


def _isolate_imports():
  drop = {REPO, ""}
  sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != REPO and p not in drop]
  sys.path.insert(0, REFERENCE)
  sys.path.insert(0, os.path.join(HERE, "refshim"))
  for m in list(sys.modules):
    if m == "big_vision" or m.startswith("big_vision.") or m in ("optax", "jax", "flax") or m.startswith(("jax.", "flax.", "optax.")):
      del sys.modules[m]


# ------------------------------------------------------------- seeded trees --
def _gen(*key):
  return np.random.default_rng([31] + [zlib.crc32(str(k).encode()) for k in key])


def _r(g, *shape):
  return g.normal(0.0, 1.0, shape).astype(np.float32)


def _mha(g):
  return {**{n: {"kernel": _r(g, D, H, D // H), "bias": _r(g, H, D // H)} for n in ("query", "key", "value")},
          "out": {"kernel": _r(g, H, D // H, D), "bias": _r(g, D)}}


def _ln(g):
  return {"scale": _r(g, D), "bias": _r(g, D)}


def _mlp(g):
  return {"Dense_0": {"kernel": _r(g, D, M), "bias": _r(g, M)}, "Dense_1": {"kernel": _r(g, M, D), "bias": _r(g, D)}}


def _block(g):
  return {"LayerNorm_0": _ln(g), "MultiHeadDotProductAttention_0": _mha(g), "LayerNorm_1": _ln(g), "MlpBlock_0": _mlp(g)}


def _stack(blocks):
  if isinstance(blocks[0], dict):
    return {k: _stack([b[k] for b in blocks]) for k in blocks[0]}
  return np.stack(blocks)


def _encoder(g, depth, scan):
  blocks = [_block(g) for _ in range(depth)]
  enc = {"encoderblock": _stack(blocks)} if scan else {f"encoderblock_{i}": b for i, b in enumerate(blocks)}
  enc["encoder_norm"] = _ln(g)
  return enc


def vit_tree(seed, grid=4, depth=2, pool="map", num_classes=5, scan=False):
  """A ViT parameter tree in the reference's naming (vit.py:206-276; pinned by tests/golden/refwiring_vit_*.npz)."""
  g = _gen("vit", seed)
  t = {"embedding": {"kernel": _r(g, 2, 2, 3, D), "bias": _r(g, D)}, "pos_embedding": _r(g, 1, grid * grid, D),
       "Transformer": _encoder(g, depth, scan)}
  if pool == "tok":
    t["cls"] = _r(g, 1, 1, D)
  if pool == "map":
    t["MAPHead_0"] = {"probe": _r(g, 1, 1, D), "MultiHeadDotProductAttention_0": _mha(g), "LayerNorm_0": _ln(g), "MlpBlock_0": _mlp(g)}
  if num_classes:
    t["head"] = {"kernel": _r(g, D, num_classes), "bias": _r(g, num_classes)}
  return t


def txt_tree(seed, length=6, depth=2, vocab=11, num_classes=5):
  g = _gen("txt", seed)
  t = {"Embed_0": {"embedding": _r(g, vocab, D)}, "pos_embedding": _r(g, 1, length, D), "Encoder_0": _encoder(g, depth, False)}
  if num_classes:
    t["head"] = {"kernel": _r(g, D, num_classes), "bias": _r(g, num_classes)}
  return t


def _copy(t):
  return {k: _copy(v) for k, v in t.items()} if isinstance(t, dict) else np.array(t)


def _flatten(tree, prefix=""):
  out = {}
  for k, v in tree.items():
    if isinstance(v, dict):
      out.update(_flatten(v, f"{prefix}{k}/"))
    else:
      out[f"{prefix}{k}"] = v
  return out


def _save(path, tree):
  np.savez(path, **_flatten(tree))


# ---------------------------------------------------------------- scenarios --
def _vit(name, tmp, ckpt, init, cfg=None, kw=None, suffix="", wrap=None):
  f = os.path.join(tmp, f"{name}.npz")
  _save(f, wrap(ckpt) if wrap else ckpt)
  return dict(kind="vit", init=init, init_file=f + suffix, model_cfg=dict(cfg or {}), kw=dict(kw or {}))


def build(name, tmp):
  """-> dict(kind, init, init_file | init_files, model_cfg, kw) after writing the scenario's checkpoint file(s) into tmp."""
  ck, init = vit_tree(1), vit_tree(2)
  if name == "vit_same_layout":
    return _vit(name, tmp, ck, init)
  if name == "vit_dont_load_head":
    return _vit(name, tmp, ck, init, kw=dict(dont_load=("head/.*",)))
  if name == "vit_init_none":
    return _vit(name, tmp, ck, None)
  if name == "vit_old_posemb_in_transformer":          # vit.py:337-340
    old = _copy(ck)
    old["Transformer"]["pos_embedding"] = old.pop("pos_embedding")
    return _vit(name, tmp, old, init)
  if name == "vit_very_old_posembed_input":            # vit.py:331-335
    old = _copy(ck)
    old["Transformer"]["posembed_input"] = {"pos_embedding": old.pop("pos_embedding")}
    return _vit(name, tmp, old, init)
  if name == "vit_combined_cls_posemb":                # vit.py:342-352: 4 x 4 + 1 position embeddings, cls += the first
    ck, init = vit_tree(3, pool="tok"), vit_tree(4, pool="tok")
    old = _copy(ck)
    old["pos_embedding"] = np.concatenate([_r(_gen("pe_cls"), 1, 1, D), old["pos_embedding"]], axis=1)
    return _vit(name, tmp, old, init)
  if name == "vit_inlined_map_head":                   # vit.py:354-359
    old = _copy(ck)
    old.update(old.pop("MAPHead_0"))
    return _vit(name, tmp, old, init)
  if name == "vit_loop_ckpt_into_scan_model":          # vit.py:417-419
    return _vit(name, tmp, ck, vit_tree(2, scan=True), cfg=dict(scan=True))
  if name == "vit_scan_ckpt_into_loop_model":          # vit.py:420-422
    return _vit(name, tmp, vit_tree(1, scan=True), init, cfg=dict(scan=False))
  if name == "vit_posemb_upsample":                    # vit.py:428-431 -> resample_posemb (scipy.ndimage.zoom, order 1)
    return _vit(name, tmp, ck, vit_tree(2, grid=6))
  if name == "vit_posemb_downsample":
    return _vit(name, tmp, vit_tree(1, grid=6), vit_tree(2, grid=3))
  if name == "vit_wrapper_params":                     # utils.py:206-208
    return _vit(name, tmp, ck, init, wrap=lambda t: {"params": t, "opt": {"count": np.zeros(1, np.float32)}})
  if name == "vit_wrapper_opt_target":                 # utils.py:209-211
    return _vit(name, tmp, ck, init, wrap=lambda t: {"opt": {"target": t}})
  if name == "vit_subkey":                             # utils.py:194-199,224-225: "file.npz:img"
    return _vit(name, tmp, {"img": ck, "txt": txt_tree(1), "t": _r(_gen("t"), 1)}, init, suffix=":img")
  if name == "vit_missing_leaf_raises":                # common.py:78-90
    bad = _copy(ck)
    del bad["head"]["bias"]
    del bad["MAPHead_0"]["probe"]
    return _vit(name, tmp, bad, init)
  if name == "vit_extra_leaf_raises":
    bad = _copy(ck)
    bad["extra"] = {"kernel": _r(_gen("x"), 2, 2)}
    return _vit(name, tmp, bad, init)
  if name == "vit_mismatch_covered_by_dont_load":
    bad = _copy(ck)
    del bad["head"]
    bad["extra"] = {"kernel": _r(_gen("x"), 2, 2)}
    return _vit(name, tmp, bad, init, kw=dict(dont_load=("head/.*", "extra/.*")))
  if name in ("txt_same_layout", "txt_posemb_added_twice", "txt_dont_load"):
    ck, init = txt_tree(1), txt_tree(2)
    if name == "txt_posemb_added_twice":               # text_transformer.py:114-117
      ck["Encoder_0"]["pos_embedding"] = _r(_gen("pe2"), 1, 6, D)
    f = os.path.join(tmp, f"{name}.npz")
    _save(f, ck)
    return dict(kind="txt", init=init, init_file=f, model_cfg={}, kw=dict(dont_load=("head/bias",)) if name == "txt_dont_load" else {})
  if name.startswith("two_"):
    full = {"img": vit_tree(5, num_classes=7), "txt": txt_tree(5, num_classes=7), "t": _r(_gen("t"), 1), "b": _r(_gen("b"), 1)}
    init = {"img": vit_tree(6, num_classes=7), "txt": txt_tree(6, num_classes=7), "t": _r(_gen("t0"), 1), "b": _r(_gen("b0"), 1)}
    cfg = dict(image=dict(scan=False), text=dict())
    f = os.path.join(tmp, f"{name}.npz")
    _save(f, full)
    if name == "two_single_file_with_bias":            # two_towers.py:99-104
      return dict(kind="two", init=init, init_files=f, model_cfg=dict(cfg, bias_init=-10.0), kw={})
    if name == "two_single_file_without_bias":         # :105-107: b is not read
      return dict(kind="two", init=init, init_files=f, model_cfg=cfg, kw={})
    if name == "two_dict_of_files":                    # :108-133, the long key spellings and load kwargs
      f2 = os.path.join(tmp, f"{name}_img.npz")
      _save(f2, vit_tree(7, num_classes=7))
      return dict(kind="two", init=init, init_files={"image": f2, "text": f + ":txt", "temperature": f + ":t"}, model_cfg=cfg,
                  kw=dict(img_load_kw=dict(dont_load=("head/.*",))))
    if name == "two_txt_only":
      return dict(kind="two", init=init, init_files={"txt": f + ":txt"}, model_cfg=cfg, kw={})
    if name == "two_typo_key_raises":                  # :135-137
      return dict(kind="two", init=init, init_files={"img": f + ":img", "imagee": f}, model_cfg=cfg, kw={})
  if name.startswith("merge_"):
    loaded, inited = vit_tree(8, pool="gap"), vit_tree(9, pool="gap")
    if name == "merge_plain":
      return dict(kind="merge", loaded=loaded, inited=inited, kw={})
    if name == "merge_dont_load_two_patterns":
      return dict(kind="merge", loaded=loaded, inited=inited, kw=dict(dont_load=("head/.*", "Transformer/encoder_norm/.*")))
    if name == "merge_inited_none":
      return dict(kind="merge", loaded=loaded, inited=None, kw={})
    if name == "merge_both_sides_differ_raises":
      del loaded["embedding"]["bias"]
      del inited["Transformer"]["encoder_norm"]
      return dict(kind="merge", loaded=loaded, inited=inited, kw=dict(dont_load=("head/bias",)))
  raise KeyError(name)


SCENARIOS = [
    "vit_same_layout", "vit_dont_load_head", "vit_init_none", "vit_old_posemb_in_transformer", "vit_very_old_posembed_input",
    "vit_combined_cls_posemb", "vit_inlined_map_head", "vit_loop_ckpt_into_scan_model", "vit_scan_ckpt_into_loop_model",
    "vit_posemb_upsample", "vit_posemb_downsample", "vit_wrapper_params", "vit_wrapper_opt_target", "vit_subkey",
    "vit_missing_leaf_raises", "vit_extra_leaf_raises", "vit_mismatch_covered_by_dont_load",
    "txt_same_layout", "txt_posemb_added_twice", "txt_dont_load",
    "two_single_file_with_bias", "two_single_file_without_bias", "two_dict_of_files", "two_txt_only", "two_typo_key_raises",
    "merge_plain", "merge_dont_load_two_patterns", "merge_inited_none", "merge_both_sides_differ_raises",
]


VARIANTS = [n + p for n in ("mu", "Ti", "S", "M", "B", "L", "So400m", "H", "g", "g-opt", "G", "G-opt", "e") for p in ("", "/16", "/14", "/32", "/8")]


def call(sc, vit, text_transformer, two_towers, common, config_of):
  """Runs one scenario through a set of loader modules (the reference's here, the product's in the test)."""
  if sc["kind"] == "vit":
    return vit.load(sc["init"], sc["init_file"], config_of(sc["model_cfg"]), **sc["kw"])
  if sc["kind"] == "txt":
    return text_transformer.load(sc["init"], sc["init_file"], config_of(sc["model_cfg"]), **sc["kw"])
  if sc["kind"] == "two":
    return two_towers.load(sc["init"], sc["init_files"], config_of(sc["model_cfg"]), **sc["kw"])
  return common.merge_params(sc["loaded"], sc["inited"], **sc["kw"])


def scrub(msg, tmp):
  """Error messages name the temporary directory: replaced by a token so that two runs compare."""
  return msg.replace(tmp, "<tmp>")


def main():
  import tempfile
  out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "tests", "golden")
  _isolate_imports()
  from ml_collections import ConfigDict
  from big_vision.models import common, vit
  from big_vision.models.proj.image_text import text_transformer, two_towers

  def config_of(d):
    return ConfigDict({k: (ConfigDict(v) if isinstance(v, dict) else v) for k, v in d.items()})

  arrays, meta = {}, {}
  with tempfile.TemporaryDirectory() as tmp:
    for name in SCENARIOS:
      sc = build(name, tmp)
      try:
        flat = _flatten(call(sc, vit, text_transformer, two_towers, common, config_of))
        arrays.update({f"{name}/{k}": np.asarray(v) for k, v in flat.items()})
        meta[name] = {"leaves": list(flat), "dtypes": sorted({str(np.asarray(v).dtype) for v in flat.values()})}
      except Exception as e:     # recorded: the product must fail the same way
        meta[name] = {"error": type(e).__name__, "message": scrub(str(e), tmp)}
  # the variant table (vit.py:284-303), every name x a few patch sizes, as the reference decodes them
  meta["__variants__"] = {v: {k: (list(x) if isinstance(x, tuple) else x) for k, x in vit.decode_variant(v).items()}
                          for v in VARIANTS}
  arrays["meta"] = np.frombuffer(json.dumps(meta, sort_keys=True).encode(), np.uint8)
  np.savez_compressed(os.path.join(out_dir, "refload.npz"), **arrays)
  print(len(SCENARIOS), "scenarios,", sum("error" in m for m in meta.values()), "raise")
  for n, m in meta.items():
    if "error" in m and n != "__variants__":
      print(" ", n, m["error"], m["message"][:90].replace("\n", " | "))


if __name__ == "__main__":
  main()
