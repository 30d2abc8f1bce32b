"""Per-kernel means of the SQ counters of one `rocprofv3 --pmc ... --kernel-trace` pass (counter_collection.csv).

    python tools/pmc_sq_summary.py <counter_collection.csv> > profiles/r02_pmc_sq.json

Reported per kernel (mean over its launches): every counter, plus
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE  (extra LDS cycles per LDS cycle)
  mfma_busy_per_wave_cycle = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES)   (WAVE_CYCLES counts quad-cycles)
Units follow /opt/skills/guides/MI355X_MICROARCH.md (rocprofv3 PMC slots): no derived gfx950 metrics exist."""
import collections
import csv
import json
import re
import sys


def short(name):
  name = re.sub(r"\(anonymous namespace\)::", "", name)
  name = re.sub(r"^void ", "", name)
  return name.split("(")[0][:90]


def main():
  rows = list(csv.DictReader(open(sys.argv[1])))
  per = collections.defaultdict(lambda: collections.defaultdict(float))
  disp = collections.defaultdict(set)
  for r in rows:
    k = short(r["Kernel_Name"])
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
  out = {}
  for k, c in per.items():
    n = max(1, len(disp[k]))
    d = {"launches": n, **{name: v / n for name, v in sorted(c.items())}}
    if d.get("SQ_LDS_IDX_ACTIVE"):
      d["lds_conflict_frac"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
    if d.get("SQ_WAVE_CYCLES"):
      d["mfma_busy_per_wave_cycle"] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * d["SQ_WAVE_CYCLES"])
    out[k] = d
  top = dict(sorted(out.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0) * kv[1]["launches"])[:24])
  json.dump({"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT "
                        "SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -- python bench.py "
                        "--steps 1 --warmup 0 --no-roofline --no-cpu-baseline",
             "kernels": top}, sys.stdout, indent=1)


if __name__ == "__main__":
  main()
