"""Minimal work-alike of `ml_collections` (not installed in this image) so the
reference's config files (`get_config()` returning a ConfigDict) import
unchanged.  Only the surface used by big_vision configs and trainers:
attribute / item access, nested dict promotion, get, keys/items, to_dict,
to_json, update, lock/unlock, FieldReference passthrough, config_dict module.
"""
from big_vision_amd.compat.ml_collections.config_dict import ConfigDict, FieldReference, FrozenConfigDict
from big_vision_amd.compat.ml_collections import config_dict

__all__ = ["ConfigDict", "FieldReference", "FrozenConfigDict", "config_dict"]
