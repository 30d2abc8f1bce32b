"""bench.py host logic that does not need a GPU: the launch observer's filter and byte model,
and the PMC traffic lookup (profiles/r01_pmc_traffic.json belongs to one configuration)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_pmc_traffic_is_tied_to_its_configuration():
  """The committed counter profile a line may quote is picked by what ONE GPU runs (pairs per GPU, micro-batch): the
  headline finds this round's headline passes, a rank of N = 8 the passes taken at its 512-pair shape, anything else
  nothing (the rest of the selection logic: tests/test_bench_line_cpu.py)."""
  first = bench.PMC_PROFILES[0]
  path = os.path.join(ROOT, "profiles", first)
  t, src = bench.pmc_traffic(1, bench.GLOBAL_BATCH, bench.MICRO)
  if os.path.exists(path):
    with open(path) as f:
      d = json.load(f)
    assert d["microbatch"] == bench.MICRO and d["n_gpus"] == 1 and d.get("per_gpu_batch", bench.GLOBAL_BATCH) == bench.GLOBAL_BATCH
    assert src == "profiles/" + first and t == d["kernels"][bench.DOMINANT_KERNEL]["hbm_bytes"]
  t8, src8 = bench.pmc_traffic(8, bench.GLOBAL_BATCH // 8, bench.MICRO)
  if os.path.exists(os.path.join(ROOT, "profiles", "r06_pmc_traffic_rank512.json")):
    assert t8 and t8 < t and "rank512" in src8
  assert bench.pmc_traffic(1, bench.GLOBAL_BATCH, bench.MICRO // 2) == (None, None)   # other micro-batch: no number
  assert bench.pmc_traffic(3, 1365, bench.MICRO) == (None, None)                      # no profile of that shape


def test_plain_multi_gpu_invocation_spawns_its_own_ranks(monkeypatch):
  """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset) must start N ranks itself."""
  import subprocess
  started = []

  class P:
    returncode = 0
    def __init__(self, cmd, env): started.append((cmd, env))
    def wait(self): return 0
    def poll(self): return 0
  monkeypatch.setattr(subprocess, "Popen", lambda cmd, env=None: P(cmd, env))
  monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: False)
  monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
  monkeypatch.delenv("WORLD_SIZE", raising=False)
  bench.main()
  assert len(started) == 4
  ranks = sorted(int(e["RANK"]) for _, e in started)
  assert ranks == [0, 1, 2, 3] and all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" for _, e in started)
  assert len({e["MASTER_PORT"] for _, e in started}) == 1 and all(e["LOCAL_RANK"] == e["RANK"] for _, e in started)
  assert all(c[-4:] == ["--gpus", "4", "--steps", "2"] for c, _ in started)


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


def test_pmc_family_totals_reads_a_counter_file(tmp_path):
  """bench.live_pmc_traffic: the per-launch traffic of the dominant family comes from rocprofv3's
  counter_collection.csv of a child run - k-major gemm256 / gemm256r rows only, KiB summed, launches counted."""
  import bench
  hdr = "Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp"
  rows = [("void (anonymous namespace)::gemm256_kernel<true, 0, 3, false>((anonymous namespace)::G256Params)", 1000.0),
          ("void (anonymous namespace)::gemm256_kernel<true, 0, 0, false>((anonymous namespace)::G256Params)", 500.0),
          ("void (anonymous namespace)::gemm256r_kernel<1, true, 0>((anonymous namespace)::G256Params)", 250.0),
          ("void (anonymous namespace)::gemm256_kernel<false, 0, 0, false>((anonymous namespace)::G256Params)", 9999.0),   # dW: not in the family
          ("void (anonymous namespace)::ln_bwd_kernel<false, 3, 3>(void const*)", 7777.0)]
  p = tmp_path / "1_counter_collection.csv"
  with open(p, "w") as f:
    f.write(hdr + "\n")
    for i, (name, val) in enumerate(rows):
      f.write(f'{i},{i},0,1,1,1,131072,{i},"{name}",512,163840,0,240,0,96,"FETCH_SIZE",{val},{1000 * i},{1000 * i + 500}\n')
  kib, n = bench.pmc_family_totals(str(p))
  assert n == 3 and kib == 1750.0
