"""Per-kernel roofline table of the headline step from the committed profiles of one round:
`rNN_bench_kernel_stats.csv` (rocprofv3 --kernel-trace --stats) for time shares and `rNN_pmc_traffic.json`
(FETCH_SIZE x 2 + WRITE_SIZE per launch, tools/pmc_summary.py) for the HBM-side bytes.  Prints markdown.

  python tools/step_roofline_table.py r05 [steps_in_trace] [workload tag: c4 | c5b -> rNN_<tag>_kernel_stats.csv, rNN_<tag>_pmc_traffic.json]
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc_summary

HBM_PEAK = 8.0   # TB/s, MI355X_MICROARCH.md


def main():
  rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
  steps = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
  tag = sys.argv[3] if len(sys.argv) > 3 else None
  stats = f"{rnd}_{tag}_kernel_stats.csv" if tag else f"{rnd}_bench_kernel_stats.csv"
  pmc = f"{rnd}_{tag}_pmc_traffic.json" if tag else f"{rnd}_pmc_traffic.json"
  rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", stats))))
  traffic = json.load(open(os.path.join(ROOT, "profiles", pmc)))["kernels"]
  tot = sum(float(r["TotalDurationNs"]) for r in rows)
  print(f"| kernel | share of GPU time | launches / step | avg us | HBM-side bytes / launch (PMC) | TB/s | of the {HBM_PEAK:.0f} TB/s peak |")
  print("|---|---|---|---|---|---|---|")
  for r in rows:
    share = float(r["TotalDurationNs"]) / tot
    if share < 0.004:
      continue
    k = pmc_summary.short(r["Name"])
    t = traffic.get(k)
    us = float(r["AverageNs"]) / 1e3
    if t:
      gb = t["hbm_bytes"] / 1e9
      tbs = t["hbm_bytes"] / (us * 1e-6) / 1e12
      extra = f"{gb:.2f} GB | {tbs:.2f} | {tbs / HBM_PEAK:.2f}"
    else:
      extra = "- | - | -"
    print(f"| `{k}` | {100 * share:.1f} % | {int(r['Calls']) / steps:.0f} | {us:.0f} | {extra} |")
  print(f"\n(total GPU time in the trace: {tot / 1e6:.0f} ms over {steps:g} steps = {tot / 1e6 / steps:.0f} ms per step)")


if __name__ == "__main__":
  main()
