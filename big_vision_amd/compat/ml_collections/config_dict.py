"""ConfigDict work-alike (see package docstring)."""
import json


class FieldReference:
  """Tiny stand-in: holds a value; `.get()` returns it."""

  def __init__(self, default, field_type=None):
    self._value = default

  def get(self):
    return self._value

  def set(self, v):
    self._value = v


class ConfigDict:
  def __init__(self, initial_dictionary=None, type_safe=True, convert_dict=True):
    object.__setattr__(self, "_fields", {})
    object.__setattr__(self, "_locked", False)
    object.__setattr__(self, "_convert_dict", convert_dict)
    if initial_dictionary is not None:
      items = initial_dictionary.items() if hasattr(initial_dictionary, "items") else initial_dictionary
      for k, v in items:
        self[k] = v

  # -- mapping protocol
  def _wrap(self, v):
    if isinstance(v, FieldReference):
      v = v.get()
    if isinstance(v, dict) and self._convert_dict:
      return ConfigDict(v)
    return v

  def __setitem__(self, key, value):
    if self._locked and key not in self._fields:
      raise KeyError(f"Key '{key}' does not exist and the ConfigDict is locked")
    self._fields[key] = self._wrap(value)

  def __getitem__(self, key):
    if "." in key and key not in self._fields:
      node = self
      for part in key.split("."):
        node = node[part]
      return node
    return self._fields[key]

  def __delitem__(self, key):
    del self._fields[key]

  def __setattr__(self, key, value):
    self[key] = value

  def __getattr__(self, key):
    if key.startswith("__"):
      raise AttributeError(key)
    try:
      return self._fields[key]
    except KeyError as e:
      raise AttributeError(f"'ConfigDict' has no attribute '{key}'") from e

  def __delattr__(self, key):
    del self._fields[key]

  def __contains__(self, key):
    return key in self._fields

  def __iter__(self):
    return iter(self._fields)

  def __len__(self):
    return len(self._fields)

  def keys(self):
    return self._fields.keys()

  def values(self):
    return self._fields.values()

  def items(self):
    return self._fields.items()

  def get(self, key, default=None):
    return self._fields.get(key, default)

  def get_ref(self, key):
    return FieldReference(self._fields[key])

  def update(self, *other, **kwargs):
    for o in other:
      for k, v in (o.items() if hasattr(o, "items") else o):
        if k in self._fields and isinstance(self._fields[k], ConfigDict) and hasattr(v, "items"):
          self._fields[k].update(v)
        else:
          self[k] = v
    for k, v in kwargs.items():
      self[k] = v

  def setdefault(self, key, default=None):
    if key not in self._fields:
      self[key] = default
    return self._fields[key]

  def lock(self):
    object.__setattr__(self, "_locked", True)
    for v in self._fields.values():
      if isinstance(v, ConfigDict):
        v.lock()
    return self

  def unlock(self):
    object.__setattr__(self, "_locked", False)
    for v in self._fields.values():
      if isinstance(v, ConfigDict):
        v.unlock()
    return self

  def unlocked(self):
    cfg = self

    class _Ctx:
      def __enter__(self_inner):
        cfg.unlock()
        return cfg

      def __exit__(self_inner, *a):
        cfg.lock()
    return _Ctx()

  @property
  def is_locked(self):
    return self._locked

  def to_dict(self):
    return {k: (v.to_dict() if isinstance(v, ConfigDict) else v) for k, v in self._fields.items()}

  def to_json(self, **kw):
    return json.dumps(self.to_dict(), default=str, **kw)

  def to_json_best_effort(self, **kw):
    return self.to_json(**kw)

  def copy_and_resolve_references(self):
    return ConfigDict(self.to_dict())

  def __eq__(self, other):
    if isinstance(other, ConfigDict):
      return self.to_dict() == other.to_dict()
    if isinstance(other, dict):
      return self.to_dict() == other
    return NotImplemented

  def __repr__(self):
    return f"ConfigDict({self.to_dict()!r})"


class FrozenConfigDict(ConfigDict):
  def __init__(self, initial_dictionary=None):
    super().__init__(initial_dictionary)
    self.lock()


def create(**kwargs):
  return ConfigDict(kwargs)


def placeholder(field_type=None):
  return None
