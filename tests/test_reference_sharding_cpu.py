"""The reference's OWN sharding inference, executed (oracle/run_reference_sharding.py: big_vision/sharding.py, pp/registry.py
and utils.py imported unmodified; pure host logic, nothing restated) vs the PRODUCT's `big_vision_amd.sharding.infer_sharding`:
for every parameter tree (two model fixtures, the real ViT-B/16 + text-B shapes), strategy (replicate, fsdp with its size
threshold, per-tower patterns, first-match-wins, unmatched leaves, a rule applied twice, contradictory rules) and mesh size
(1, 2, 8 and the non-divisor 3) the product returns the reference's partition spec for every leaf - or raises where the
reference raises.  These specs decide which placement `make_train_state` builds (reference sharding.py:38-139)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import run_reference_sharding as RS  # noqa: E402  (tables only; nothing of the reference is imported here)

GOLDEN = os.path.join(ROOT, "tests", "golden", "refsharding.json")


class _Mesh:
  def __init__(self, size):
    self.size = size


def _nest(flat):
  tree = {}
  for k, v in flat.items():
    node = tree
    *parents, last = k.split("/")
    for p in parents:
      node = node.setdefault(p, {})
    node[last] = v
  return tree


def _flat_specs(tree, prefix=""):
  """{leaf name: spec tuple} of a spec tree (the specs are tuples: a generic tree flatten would walk into them)."""
  out = {}
  for k, v in tree.items():
    if isinstance(v, dict):
      out.update(_flat_specs(v, f"{prefix}{k}/"))
    else:
      out[f"{prefix}{k}"] = v
  return out


@pytest.fixture(scope="module")
def golden():
  return json.load(open(GOLDEN))


@pytest.mark.parametrize("tree", RS.TREES)
def test_product_specs_equal_the_executed_reference(golden, tree):
  from big_vision_amd import sharding
  shapes = {k: tuple(v) for k, v in golden["trees"][tree].items()}
  params = _nest({k: torch.empty(s, device="meta") for k, s in shapes.items()})
  names = sorted(shapes)
  checked = raised = 0
  for sname, strategy in golden["strategies"].items():
    for n in RS.MESHES:
      want = golden["cases"][f"{tree}|{sname}|{n}"]
      strategy = [tuple(x) for x in strategy]
      if "error" in want:
        with pytest.raises(ValueError, match="Inconsistent sharding instructions"):
          sharding.infer_sharding(params, strategy, _Mesh(n))
        assert want["error"] == "ValueError" and "Inconsistent sharding instructions" in want["message"]
        raised += 1
        continue
      got = _flat_specs(sharding.infer_sharding(params, strategy, _Mesh(n)))
      assert set(got) == set(names)
      for k, code in zip(names, want["sharded_axes"]):
        spec = tuple(got[k])
        assert len(spec) == len(shapes[k]) and all(a in (None, "data") for a in spec), (k, spec)
        assert RS._code(spec) == code, (tree, sname, n, k, spec, code)
        checked += 1
  assert checked > 1000 and raised == 4      # fsdp|replicate raises on every mesh size (also on one device)


def test_the_fixture_has_teeth(golden):
  """The cases are not all trivially replicated: the size threshold, divisibility and the second application matter."""
  c = golden["cases"]
  names = sorted(golden["trees"]["b16"])
  at = lambda case, leaf: c[case]["sharded_axes"][names.index(leaf)]
  k = "img/Transformer/encoderblock_0/MlpBlock_0/Dense_0/kernel"      # [768, 3072] fp32 = 9 MiB
  q = "img/Transformer/encoderblock_0/MultiHeadDotProductAttention_0/query/kernel"   # [768, 12, 64] = 2.25 MiB
  assert at("b16|fsdp_default|8", k) == "1" and at("b16|fsdp_default|8", q) == "-"   # 4 MiB threshold
  assert at("b16|fsdp_all|8", q) == "0" and at("b16|fsdp_all|3", q) == "0"           # 768 = 3 x 256
  assert at("b16|fsdp_all|8", "txt/Embed_0/embedding") == "0" and at("b16|fsdp_all|3", "txt/Embed_0/embedding") == "1"   # 32000 % 3 != 0
  assert at("b16|fsdp_twice|2", k) == "01" and at("b16|fsdp_img_only|2", "txt/head/kernel") == "-"
  assert at("b16|unmatched_stay_replicated|2", k) == "-" and at("b16|unmatched_stay_replicated|2", "txt/head/kernel") in ("0", "1")


@pytest.mark.skipif(not os.path.isdir(os.path.join(RS.REFERENCE, "big_vision")), reason="the reference tree is not on this host")
def test_committed_fixture_is_what_the_reference_produces_now(tmp_path):
  r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "run_reference_sharding.py"), str(tmp_path)],
                     capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stderr[-2000:]
  assert json.load(open(tmp_path / "refsharding.json")) == json.load(open(GOLDEN))
