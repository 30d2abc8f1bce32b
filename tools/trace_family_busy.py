"""Exclusive busy time of the dominant kernel family from a rocprofv3 --kernel-trace CSV: the union of the family's
dispatch intervals (towers on two streams: dispatches overlap, so the per-kernel average durations of `--stats` are no
longer exclusive).  The figure bench.py's live `roofline` (HIP events, bench.GemmObserver) must agree with.

  python tools/trace_family_busy.py <kernel_trace.csv> [steps_in_trace] > profiles/rNN_family_busy.json
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import pmc_summary  # noqa: E402


def union_ns(iv):
  busy, a0, b0 = 0, None, None
  for a, b in sorted(iv):
    if b0 is None or a > b0:
      if b0 is not None:
        busy += b0 - a0
      a0, b0 = a, b
    elif b > b0:
      b0 = b
  return busy + ((b0 - a0) if b0 is not None else 0)


def main():
  steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
  fam, other, per = [], [], {}
  with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
      k = pmc_summary.short(r["Kernel_Name"])
      iv = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
      (fam if (k.startswith("gemm256_kernel<true") or k.startswith("gemm256r_kernel")) else other).append(iv)
      per.setdefault(k, []).append(iv)
  every = fam + other
  out = {"family": pmc_summary.FAMILY, "launches": len(fam),
         "sum_of_durations_ms": sum(b - a for a, b in fam) / 1e6, "exclusive_busy_ms": union_ns(fam) / 1e6,
         "avg_exclusive_us_per_launch": union_ns(fam) / 1e3 / max(1, len(fam)),
         "avg_duration_us_per_launch": sum(b - a for a, b in fam) / 1e3 / max(1, len(fam)),
         "all_kernels": {"launches": len(every), "sum_of_durations_ms": sum(b - a for a, b in every) / 1e6,
                         "exclusive_busy_ms": union_ns(every) / 1e6,
                         "span_ms": (max(b for _, b in every) - min(a for a, _ in every)) / 1e6 if every else 0.0},
         "steps_in_trace": steps,
         "by_kernel": {k: {"launches": len(v), "sum_ms": sum(b - a for a, b in v) / 1e6, "exclusive_ms": union_ns(v) / 1e6}
                       for k, v in sorted(per.items(), key=lambda kv: -sum(b - a for a, b in kv[1]))[:24]}}
  json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
  main()
