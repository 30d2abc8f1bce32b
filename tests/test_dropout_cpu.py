"""Host side of dropout > 0 (models/vit.py:76,100,109,228), no GPU: the Philox restatement the mask tests compare the
kernels with is pinned by Random123's published known-answer vectors; the site keys of engine.Dropout are distinct."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bv_oracle as O  # noqa: E402


# Random123 (kat_vectors, philox4x32 with 10 rounds): counter, key -> output
KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


@pytest.mark.parametrize("ctr,key,want", KAT)
def test_philox_known_answers(ctr, key, want):
  got = O.philox4x32_10(np.array([ctr], np.uint32), np.array([key], np.uint32))[0]
  assert tuple(int(v) for v in got) == want


def test_keep_mask_is_bernoulli_and_keyed():
  a = O.dropout_keep_mask(0x1234567890ABCDEF, 1 << 16, 0.25)
  b = O.dropout_keep_mask(0x1234567890ABCDEE, 1 << 16, 0.25)
  assert abs(a.mean() - 0.75) < 0.01 and abs(b.mean() - 0.75) < 0.01
  assert abs((a == b).mean() - (0.75 ** 2 + 0.25 ** 2)) < 0.01      # independent streams
  assert O.dropout_keep_mask(7, 64, 0.0).all()


def test_site_keys_are_distinct_and_reproducible():
  from big_vision_amd import engine as E
  d = E.Dropout(0.1, 42)
  keys = set()
  for tower in ("img", "txt"):
    t = d.fold(tower)
    keys.add(t.key(E.DROP_POSEMB))
    for i in range(24):
      b = t.fold("block", i)
      keys.update(b.key(s) for s in (E.DROP_SA, E.DROP_GELU, E.DROP_MLP))
  assert len(keys) == 2 * (1 + 24 * 3)
  assert E.Dropout(0.1, 42).fold("img").fold("block", 3).key(E.DROP_GELU) == d.fold("img").fold("block", 3).key(E.DROP_GELU)
  assert E.Dropout(0.1, 43).fold("img").key(E.DROP_POSEMB) != d.fold("img").key(E.DROP_POSEMB)
  with pytest.raises(ValueError):
    E.Dropout(1.0, 0)


def test_models_accept_dropout_and_need_an_rng_in_train_mode():
  from big_vision_amd.models import vit
  from big_vision_amd.models.proj.image_text import text_transformer
  m = vit.Model(num_classes=10, variant="mu/16", dropout=0.1)
  assert m.dropout == 0.1
  assert text_transformer.Model(num_classes=16, width=32, depth=1, mlp_dim=64, num_heads=2, dropout=0.2).dropout == 0.2
  with pytest.raises(ValueError):
    vit.Model(num_classes=10, variant="mu/16", dropout=1.5)
  assert vit.dropout_for(0.1, False, None) is None and vit.dropout_for(0.0, True, None) is None
  with pytest.raises(ValueError):
    vit.dropout_for(0.1, True, None)
  d = vit.dropout_for(0.1, True, {"dropout": 5})
  assert d.rate == 0.1
