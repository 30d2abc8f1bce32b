"""bench.py host logic that does not need a GPU: the launch observer's filter and byte model,
and the PMC traffic lookup (profiles/r01_pmc_traffic.json belongs to one configuration)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_pmc_traffic_is_tied_to_its_configuration():
  t, src = bench.pmc_traffic(1, bench.MICRO)
  with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
    d = json.load(f)
  assert d["microbatch"] == bench.MICRO and d["n_gpus"] == 1
  assert src == "profiles/r01_pmc_traffic.json" and t == d["kernels"][bench.DOMINANT_KERNEL]["hbm_bytes"]
  assert bench.pmc_traffic(8, bench.MICRO) == (None, None)        # other world size: no number
  assert bench.pmc_traffic(1, bench.MICRO // 2) == (None, None)   # other micro-batch: no number


def test_observer_filters_and_counts_bytes():
  obs = bench.GemmObserver()
  args = [1, 1, 0, 768, 0, 768, 0, 2304, 0, 100352, 2304, 768, 0, 0, 0, 0, 0, 0, 1.0, 0, 0]
  assert obs.begin("bv_gemm_bf16", args) is None                  # inactive
  obs.active = True
  assert obs.begin("bv_layernorm_fwd", args) is None
  assert obs.begin("bv_gemm_bf16", [0, 0] + args[2:]) is None     # dW (k-minor) is not the dominant kernel
  ragged = list(args); ragged[9] = 100352 + 8
  assert obs.begin("bv_gemm_bf16", ragged) is None                # falls to the general 128x128 kernel


def test_config_matches_the_baseline_workload():
  c = bench.make_config(20_000)
  assert c.optax_name == "scale_by_adam" and c.grad_clip_norm == 1.0 and c.schedule["decay_type"] == "cosine"
  assert bench.GLOBAL_BATCH == 4096 and (bench.RES, bench.SEQ, bench.VOCAB, bench.EMB) == (224, 64, 32_000, 768)
  assert bench.IMAGE_CFG == dict(variant="B/16", pool_type="map") and bench.TEXT_CFG["variant"] == "B"
