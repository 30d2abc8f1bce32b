"""Config-file helpers under the reference's module path `big_vision.configs.common`
(imported by config files as `bvcc`): the single-string argument parser and its friends.
Behaviour follows big_vision/configs/common.py:22-160 (docstring examples are the contract:
`runlocal`, `res=128`, a lone unnamed first value, strict bools, lazy extras)."""
from big_vision_amd.compat import ml_collections as mlc


def input_for_quicktest(config_input, quicktest):
  if quicktest:
    config_input.batch_size = 8
    config_input.shuffle_buffer_size = 10
    config_input.cache_raw = False


def _strict_bool(x):
  if x.lower() not in ("true", "false", ""):
    raise AssertionError(f"not a boolean: {x!r}")
  return x.lower() == "true"


def get_type_with_default(v):
  """(default, string -> value) for one spec entry; bools parse strictly."""
  if isinstance(v, bool):
    return v, _strict_bool
  if isinstance(v, (tuple, list)):
    assert len(v) == 2 and isinstance(v[1], type), "spec tuples are (default, type)"
    return v[0], v[1]
  return v, type(v)


def autotype(x):
  """str -> bool / int / float when it parses as one."""
  assert isinstance(x, str)
  if x.lower() in ("true", "false"):
    return x.lower() == "true"
  for conv in (int, float):
    try:
      return conv(x)
    except ValueError:
      pass
  return x


def parse_arg(arg, lazy=False, **spec):
  arg = arg or ""
  spec = {k: get_type_with_default(v) for k, v in spec.items()}
  result = mlc.ConfigDict(type_safe=False)
  if arg and "," not in arg and "=" not in arg:
    arg = f"{arg}=True" if (arg in spec or not spec) else f"{next(iter(spec))}={arg}"
  raw = {}
  for piece in arg.split(","):
    if piece:
      k, _, v = piece.partition("=")
      raw[k] = v if "=" in piece else "True"
  for name, (default, conv) in spec.items():
    val = raw.pop(name, None)
    result[name] = conv(val) if val is not None else default
  if raw:
    if not lazy:
      raise ValueError(f"Unhandled config args remain: {raw}")
    for k, v in raw.items():
      result[k] = autotype(v)
  return result


def pack_arg(**kw):
  for v in kw.values():
    assert "," not in f"{v}", f"Can't use `,` in config_arg value: {v}"
  return ",".join(f"{k}={v}" for k, v in kw.items())


def arg(**kw):
  return {"config_arg": pack_arg(**kw), **kw}
