"""The reference's OWN classification losses and mixup, executed (oracle/run_reference_utils.py: big_vision/utils.py imported
unmodified over oracle/refshim; `sigmoid_xent`, `softmax_xent`, `bidirectional_contrastive_loss`, `get_mixup` / `mixup`) vs the
oracle's restatements that the classification-step GPU tests trust (bv_oracle.sigmoid_xent / softmax_xent / mixup,
softmax_loss_per_device on one device)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bv_oracle as O  # noqa: E402
import run_reference_utils as RU  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden", "refutils.npz")
T = lambda a: torch.from_numpy(np.asarray(a, np.float64))


@pytest.fixture(scope="module")
def z():
  return np.load(GOLDEN)


def test_oracle_losses_equal_the_executed_reference(z):
  logits = T(z["in/logits"])
  for lab in ("hard", "soft", "multi"):
    got = O.sigmoid_xent(logits, T(z[f"in/{lab}"])).item()
    assert abs(got - float(z[f"sigmoid_xent/{lab}/mean"])) <= 1e-12 * abs(got), lab
    assert abs(float(np.mean(z[f"sigmoid_xent/{lab}/per_example"])) - float(z[f"sigmoid_xent/{lab}/mean"])) <= 1e-12 * abs(got)
  for lab in ("hard", "soft"):
    got = O.softmax_xent(logits, T(z[f"in/{lab}"])).item()
    assert abs(got - float(z[f"softmax_xent/{lab}/mean"])) <= 1e-12 * abs(got), lab
  # the kl form only shifts by the labels' negative entropy (0 for one-hot labels)
  assert abs(float(z["softmax_xent/hard/kl_mean"]) - float(z["softmax_xent/hard/mean"])) <= 1e-6
  soft = z["in/soft"]
  ent = float(np.mean(np.sum(soft * np.log(np.clip(soft, 1e-8, None)), -1)))
  assert abs(float(z["softmax_xent/soft/kl_mean"]) - float(z["softmax_xent/soft/mean"]) - ent) <= 1e-12
  assert np.isfinite(z["sigmoid_xent/hard/per_example"]).all() and z["sigmoid_xent/hard/per_example"][0] > 50   # the saturated row


def test_oracle_softmax_contrastive_loss_equals_the_executed_utils_function(z):
  """utils.bidirectional_contrastive_loss on one device = the oracle's per-device softmax loss with one shard."""
  zimg, ztxt = T(z["in/zimg"]), T(z["in/ztxt"])
  t = json.loads(bytes(z["meta"]).decode())["temperature"]
  got = O.softmax_loss_per_device(zimg, ztxt, [zimg], [ztxt], 0, torch.tensor(float(t), dtype=torch.float64))
  got = got[0] if isinstance(got, tuple) else got
  assert abs(float(got) - float(z["bidirectional/mean"])) <= 1e-12 * abs(float(got))
  assert abs(float(np.mean(z["bidirectional/per_example"])) - float(z["bidirectional/mean"])) <= 1e-12


def test_oracle_mixup_equals_the_executed_reference(z):
  a = float(z["mixup/a"])
  img, lab = O.mixup(a, T(z["in/images"]), T(z["in/labels"]))
  assert np.max(np.abs(img.numpy() - z["mixup/images"])) <= 1e-12 and np.max(np.abs(lab.numpy() - z["mixup/labels"])) <= 1e-12
  # the keyword spelling mixes the same way with the same coefficient (same key)
  assert np.array_equal(z["mixup_kw/images"], z["mixup/images"]) and np.array_equal(z["mixup_kw/labels"], z["mixup/labels"])
  # a = max(a, 1 - a): the un-rolled sample dominates
  assert 0.5 <= a <= 1.0


@pytest.mark.skipif(not os.path.isdir(os.path.join(RU.REFERENCE, "big_vision")), reason="the reference tree is not on this host")
def test_committed_fixture_is_what_the_reference_produces_now(tmp_path):
  r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "run_reference_utils.py"), str(tmp_path)],
                     capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stderr[-2000:]
  a, b = np.load(tmp_path / "refutils.npz"), np.load(GOLDEN)
  assert sorted(a.files) == sorted(b.files)
  for k in a.files:
    assert np.array_equal(a[k], b[k]), k
