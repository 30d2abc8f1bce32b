"""The reference's OWN tree and duration helpers, executed (oracle/run_reference_treeutils.py: utils.py unmodified) vs the
PRODUCT's `big_vision_amd.utils`: `tree_flatten_with_names` (names AND order: sorted dict keys, indexed sequences, `None`
leaves dropped), `recover_tree`, `tree_map_with_names`, `make_mask_trees` (first match wins), `steps` with every spelling of a
duration and the errors it raises.  These name every checkpoint entry and every config duration."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import run_reference_treeutils as RT  # noqa: E402  (tables and the `run` driver; nothing of the reference)

GOLDEN = os.path.join(ROOT, "tests", "golden", "reftreeutils.json")


def test_product_helpers_return_what_the_reference_returns():
  from big_vision_amd import utils as u
  want = json.load(open(GOLDEN))
  got = json.loads(json.dumps(RT.run(u), default=lambda o: o.item() if hasattr(o, "item") else str(o)))
  for part in ("flatten", "recover", "map_with_names", "masks"):
    assert got[part] == want[part], part
  assert len(got["steps"]) == len(want["steps"]) == len(RT.STEPS)
  for (prefix, cfg, kw), g, w in zip(RT.STEPS, got["steps"], want["steps"]):
    assert ("error" in g) == ("error" in w), (prefix, cfg, kw, g, w)
    if "error" not in w:
      assert g["value"] == w["value"], (prefix, cfg, kw, g, w)
      continue
    assert g["error"] == w["error"], (prefix, cfg, kw, g, w)
    if w["message"].startswith("Only one of"):      # (the message prints a SET: its order is not defined)
      assert g["message"].startswith("Only one of") and sorted(g["message"]) == sorted(w["message"])
    else:
      assert g["message"] == w["message"], (prefix, cfg, kw)
  # the cases cover every branch
  assert sum("error" in w for w in want["steps"]) == 4
  assert want["flatten"]["with_none"] == [["img/kernel", 1], ["t", 2]]
  assert [k for k, _ in want["flatten"]["unsorted_numeric_keys"]] == ["encoderblock_1/k", "encoderblock_10/k", "encoderblock_2/k"]


@pytest.mark.skipif(not os.path.isdir(os.path.join(RT.REFERENCE, "big_vision")), reason="the reference tree is not on this host")
def test_committed_fixture_is_what_the_reference_produces_now(tmp_path):
  r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "run_reference_treeutils.py"), str(tmp_path)],
                     capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stderr[-2000:]
  a, b = json.load(open(tmp_path / "reftreeutils.json")), json.load(open(GOLDEN))
  for s_ in (a, b):      # (one message prints a set)
    for c in s_["steps"]:
      if c.get("message", "").startswith("Only one of"):
        c["message"] = "".join(sorted(c["message"]))
  assert a == b
