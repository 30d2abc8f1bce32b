"""Loads big_vision config FILES unchanged.

A reference config is a Python file with `get_config(arg=None)` that imports
`ml_collections` and, often, `big_vision.configs.common as bvcc` (e.g.
configs/vit_s16_i1k.py, configs/proj/image_text/siglip_lit_coco.py).  Neither package is
installed here, so the loader executes the file with those names mapped to
`big_vision_amd.compat.ml_collections` and `big_vision_amd.configs.common`; any further
`big_vision.*` import (dataset / evaluator helper tables, outside the hot path) is served
from a reference checkout when one is given, and is an ImportError otherwise.

  config = load_config("big_vision/configs/vit_s16_i1k.py")
  config = load_config(".../siglip_lit_coco.py:txt=bert_base", reference_root="/path/to/big_vision")
"""
import importlib
import importlib.util
import os
import sys
import types


def _is_bv(name):
  return name == "big_vision" or name.startswith("big_vision.")


def _alias_modules(reference_root):
  """Installs the names a reference config file imports and returns a restore() callback.  While a
  file is being loaded, `big_vision` is a bare namespace rooted at the reference checkout (so its
  helper tables resolve there) with `big_vision.configs.common` served from this repo; the
  repo's own `big_vision` alias package (if imported) is parked and comes back afterwards."""
  from big_vision_amd.compat import ml_collections as mlc
  from big_vision_amd.configs import common
  parked = {n: m for n, m in sys.modules.items() if _is_bv(n)}
  for n in parked:
    del sys.modules[n]
  finders = [f for f in sys.meta_path if type(f).__name__ == "_AliasFinder"]
  for f in finders:
    sys.meta_path.remove(f)
  had_mlc = "ml_collections" in sys.modules
  if not had_mlc:
    sys.modules["ml_collections"] = mlc
  for pkg in ("big_vision", "big_vision.configs"):
    m = types.ModuleType(pkg)
    m.__path__ = [os.path.join(reference_root, *pkg.split("."))] if reference_root else []
    sys.modules[pkg] = m
  sys.modules["big_vision.configs.common"] = common
  sys.modules["big_vision.configs"].common = common

  def restore():
    for n in [n for n in sys.modules if _is_bv(n)]:
      del sys.modules[n]
    if not had_mlc:
      sys.modules.pop("ml_collections", None)
    sys.modules.update(parked)
    for f in finders:
      sys.meta_path.insert(0, f)
  return restore


def load_config(path_and_arg, reference_root=None):
  """`path[:arg]` like `--config file.py:arg` of the reference launcher (train.py:63-64)."""
  path, _, arg = path_and_arg.partition(":")
  restore = _alias_modules(reference_root)
  try:
    spec = importlib.util.spec_from_file_location("_bv_config_file", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    get_config = mod.get_config
    try:
      takes_arg = get_config.__code__.co_argcount > 0
    except AttributeError:
      takes_arg = False
    return get_config(arg or None) if takes_arg else get_config()
  finally:
    restore()
