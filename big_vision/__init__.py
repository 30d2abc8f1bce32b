"""`big_vision` import surface of the MI355X-native hot path.

The reference resolves everything by module path: trainers import
`big_vision.optax`, `big_vision.utils`, `big_vision.sharding`, and build models with
`importlib.import_module(f"big_vision.models.{config.model_name}")`
(trainers/proj/image_text/siglip.py:190-191, models/proj/image_text/two_towers.py:51-53);
configs import `big_vision.configs.common`.  This package makes those spellings resolve to the
implementation in `big_vision_amd` WITHOUT a second copy of any module: `big_vision.X` IS
`big_vision_amd.X` (same module object, registered under both names), for every X that exists
there - including the historical names BASELINE.json / README.md use
(`trainers.proj.image_text.contrastive`, `configs.proj.image_text.lit_coco`).  A name that has
no counterpart in `big_vision_amd` (the reference's input pipeline, other model families, ...)
raises ModuleNotFoundError naming the missing implementation - it is out of the hot-path scope,
not silently stubbed.
"""
import importlib
import importlib.abc
import importlib.util
import sys

_SRC = "big_vision_amd"


class _AliasLoader(importlib.abc.Loader):
  def __init__(self, target):
    self.target = target

  def create_module(self, spec):
    return importlib.import_module(self.target)     # the implementation module itself

  def exec_module(self, module):
    pass                                             # already executed under its own name


class _AliasFinder(importlib.abc.MetaPathFinder):
  def find_spec(self, fullname, path=None, target=None):
    if not fullname.startswith(__name__ + "."):
      return None
    real = _SRC + fullname[len(__name__):]
    try:
      found = importlib.util.find_spec(real)
    except ModuleNotFoundError:
      found = None
    if found is None:
      raise ModuleNotFoundError(
          f"No module named '{fullname}': '{real}' does not exist - this part of big_vision is outside "
          "the accelerated hot path (SURVEY.md §8)", name=fullname)
    spec = importlib.util.spec_from_loader(fullname, _AliasLoader(real), is_package=found.submodule_search_locations is not None)
    return spec


def _provide_ml_collections():
  """Reference config files `import ml_collections`.  When that package is not installed, the
  subset big_vision configs use (ConfigDict / FieldReference, big_vision_amd.compat.ml_collections)
  answers to the name; an installed ml_collections always wins."""
  if "ml_collections" in sys.modules:
    return
  try:
    if importlib.util.find_spec("ml_collections") is not None:
      return
  except (ImportError, ValueError):
    pass
  compat = importlib.import_module(_SRC + ".compat.ml_collections")
  sys.modules["ml_collections"] = compat
  if hasattr(compat, "config_dict"):
    sys.modules["ml_collections.config_dict"] = compat.config_dict


_provide_ml_collections()
if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
  sys.meta_path.insert(0, _AliasFinder())
__path__ = []          # a package whose submodules all come from the finder above
