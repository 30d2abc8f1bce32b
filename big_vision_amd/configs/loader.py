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


def _alias_modules(reference_root):
  from big_vision_amd.compat import ml_collections as mlc
  from big_vision_amd.configs import common
  added = {}

  def put(name, mod):
    if name not in sys.modules:
      sys.modules[name] = mod
      added[name] = mod

  put("ml_collections", mlc)
  for pkg in ("big_vision", "big_vision.configs"):
    m = types.ModuleType(pkg)
    m.__path__ = [os.path.join(reference_root, *pkg.split("."))] if reference_root else []
    put(pkg, m)
  put("big_vision.configs.common", common)
  sys.modules["big_vision.configs"].common = common
  return added


def load_config(path_and_arg, reference_root=None):
  """`path[:arg]` like `--config file.py:arg` of the reference launcher (train.py:63-64)."""
  path, _, arg = path_and_arg.partition(":")
  added = _alias_modules(reference_root)
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
    for name in added:
      sys.modules.pop(name, None)
    for name in [n for n in sys.modules if n.startswith("big_vision.") and not n.startswith("big_vision_amd")]:
      if reference_root and name not in added:
        sys.modules.pop(name, None)
