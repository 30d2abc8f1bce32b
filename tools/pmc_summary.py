"""Per-kernel HBM traffic from rocprofv3 --pmc passes (counter_collection.csv files).

  python tools/pmc_summary.py <FETCH_SIZE csv> <WRITE_SIZE csv> > profiles/rNN_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  MI355X_MICROARCH.md (HBM section):
on gfx950 FETCH_SIZE counts 128-B requests at 64 B, i.e. reports exactly HALF of the bytes of a
wide coalesced (16 B/lane) read stream -> doubled here; WRITE_SIZE is taken as reported
(uncalibrated by the guide; the adam_kernel row, whose write volume is known exactly, is the
in-run calibration point and is printed with its expected byte count).
"""
import csv, json, re, sys, collections


# the k-major GEMM family bench.py reports as its dominant kernel (same string as bench.DOMINANT_KERNEL)
FAMILY = "gemm256_kernel<true> + gemm256r_kernel (256x256 k-major bf16 MFMA GEMM, all epilogues)"


def short(n):
  n = re.sub(r"\(anonymous namespace\)::", "", n)
  n = re.sub(r"^void ", "", n)
  m = re.match(r"([A-Za-z0-9_:]+(<[^()]*?>)?)", n)
  s = m.group(1) if m else n[:40]
  if s.startswith("_ZN12_GLOBAL__N_1"):
    s = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", s)
    s = re.sub(r"(E|IL).*", "", s)
  if s.startswith("at::native"):
    s = "torch:" + re.sub(r"<.*", "", s.split("::")[-1])[:30]
  return s[:60]


def load(path):
  tot = collections.defaultdict(float); cnt = collections.Counter(); dur = collections.Counter()
  with open(path) as f:
    for r in csv.DictReader(f):
      k = short(r["Kernel_Name"])
      tot[k] += float(r["Counter_Value"]); cnt[k] += 1
      dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
  return tot, cnt, dur


def main():
  ft, fc, fd = load(sys.argv[1])
  wt, wc, _ = load(sys.argv[2])
  opts = dict(a.lstrip("-").split("=", 1) for a in sys.argv[3:])   # --microbatch=2048 --n_gpus=1
  out = {"n_gpus": int(opts.get("n_gpus", 1)), "microbatch": int(opts.get("microbatch", 2048)),
         "per_gpu_batch": int(opts.get("per_gpu_batch", 4096)), "workload": opts.get("workload", "headline (bench.py)"),
         "command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python bench.py --steps 1 --warmup 0 "
                    "--no-roofline --no-cpu-baseline",
         "units": "bytes per launch (mean over the launches of one training step)",
         "corrections": "FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count of 16 B/lane streams); WRITE_SIZE KiB x 1024",
         "kernels": {}}
  fam = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
  for k in sorted(ft, key=lambda k: -ft[k]):
    n = fc[k]
    rd = ft[k] * 1024 * 2 / n
    wr = wt.get(k, 0.0) * 1024 / max(1, wc.get(k, 0))
    out["kernels"][k] = {"launches": n, "read_bytes": rd, "write_bytes": wr, "hbm_bytes": rd + wr,
                         "avg_us_under_pmc": fd[k] / n / 1e3}
    if k.startswith("gemm256_kernel<true") or k.startswith("gemm256r_kernel"):
      f = fam[FAMILY]
      f[0] += n; f[1] += ft[k] * 1024 * 2; f[2] += wt.get(k, 0.0) * 1024; f[3] += fd[k]
  for k, (n, rd, wr, d) in fam.items():
    out["kernels"][k] = {"launches": n, "read_bytes": rd / n, "write_bytes": wr / n, "hbm_bytes": (rd + wr) / n,
                         "avg_us_under_pmc": d / n / 1e3}
  json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
  main()
