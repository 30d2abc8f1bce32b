"""FETCH_SIZE of the wide k-major GEMMs per tile order (BV_OPT_GEMM_GROUP_N).  Run under
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -- python tools/gemm_group_pmc.py run
then  python tools/gemm_group_pmc.py parse <counter_collection.csv>.  GPU only."""
import csv
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

GROUPS = (0, 3, 4, 6)
TS = (401408, 131072)
CASES = ("qkv bias N=2304", "fc1 gelu N=3072", "dfc2 gelu'emit N=3072")
REPS = 2


def run():
  import torch
  from big_vision_amd import _lib, ops
  lib = _lib.load()
  dev = torch.device("cuda:0")
  BF16 = torch.bfloat16
  D, M = 768, 3072
  for T in TS:
    x = torch.randn(T, D, device=dev).to(BF16)
    hM = torch.randn(T, M, device=dev).to(BF16)
    wq = (torch.randn(3 * D, D, device=dev) * 0.02).to(BF16)
    w1 = (torch.randn(M, D, device=dev) * 0.02).to(BF16)
    bq, b1 = torch.randn(3 * D, device=dev), torch.randn(M, device=dev)
    oq = torch.empty(T, 3 * D, device=dev, dtype=BF16)
    o1, o2 = torch.empty(T, M, device=dev, dtype=BF16), torch.empty(T, M, device=dev, dtype=BF16)
    torch.cuda.synchronize()
    for g in GROUPS:
      ops.ctx_set("gemm_group_n", g)
      for _ in range(REPS):
        ops.gemm(x, wq, a_kmajor=True, b_kmajor=True, out=oq, bias=bq)
      for _ in range(REPS):
        ops.gemm(x, w1, a_kmajor=True, b_kmajor=True, out=o1, out2=o2, bias=b1, epilogue=ops.EPI_GELU)
      for _ in range(REPS):
        ops.gemm(x, w1, a_kmajor=True, b_kmajor=True, out=o1, out2=o2, epilogue=ops.EPI_GELU_BWD_EMIT, aux=hM)
      torch.cuda.synchronize()
    ops.ctx_set("gemm_group_n", 0)


def parse(path):
  rows = [r for r in csv.DictReader(open(path)) if "gemm256" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]]
  rows.sort(key=lambda r: int(r["Dispatch_Id"]))
  assert len(rows) == len(TS) * len(GROUPS) * len(CASES) * REPS, len(rows)
  print("# FETCH_SIZE x 2 (gfx950 correction), GB per launch | us under the counter pass; columns: group_n =", GROUPS)
  it = iter(rows)
  tab = {}
  for T in TS:
    for g in GROUPS:
      for c in CASES:
        rs = [next(it) for _ in range(REPS)]
        gb = sum(float(r["Counter_Value"]) for r in rs) * 1024 * 2 / REPS / 1e9
        us = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / REPS / 1e3
        tab[(T, c, g)] = (gb, us)
  for T in TS:
    for c in CASES:
      N = 2304 if "2304" in c else 3072
      alg = (T * 768 * 2 + N * 768 * 2 + (T * N * 2 if "emit" in c else 0)) / 1e9
      print(f"T={T:6d} {c:24s} algorithmic reads {alg:5.2f} GB | " +
            "  ".join(f"{tab[(T, c, g)][0]:5.2f} GB {tab[(T, c, g)][1]:7.1f} us" for g in GROUPS))


if __name__ == "__main__":
  run() if sys.argv[1] == "run" else parse(sys.argv[2])
