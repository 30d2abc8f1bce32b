def to_state_dict(x):
  import dataclasses
  return dataclasses.asdict(x) if dataclasses.is_dataclass(x) else x
