"""Gap analysis of a rocprofv3 --kernel-trace CSV: per-kernel busy time, idle gaps
between consecutive dispatches, and which kernel FOLLOWS the large gaps."""
import csv, re, sys, collections
def short(n):
  n = re.sub(r"\(anonymous namespace\)::", "", n)
  n = re.sub(r"^void ", "", n)
  m = re.match(r"([A-Za-z0-9_:]+(<[^()]*?>)?)", n)
  s = m.group(1) if m else n[:40]
  if s.startswith("at::native"): s = "torch:" + re.sub(r"<.*", "", s.split("::")[-1])[:30]
  return s[:60]
rows = []
with open(sys.argv[1]) as f:
  for r in csv.DictReader(f):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
# steady region: find adam_kernel launches as step delimiters
adam = [i for i, r in enumerate(rows) if r[2].startswith("adam_kernel")]
print("rows", len(rows), "adam at", adam)
if len(adam) >= 2:
  lo, hi = adam[-2] + 1, adam[-1] + 1
else:
  lo, hi = 0, len(rows)
seg = rows[lo:hi]
span = seg[-1][1] - seg[0][0]
busy = sum(e - s for s, e, _ in seg)
print("last step: %d kernels, span %.1f ms, busy %.1f ms, idle %.1f ms" % (len(seg), span / 1e6, busy / 1e6, (span - busy) / 1e6))
gap_after = collections.Counter(); gap_cnt = collections.Counter(); bt = collections.Counter(); bc = collections.Counter()
hist = collections.Counter()
for a, b in zip(seg, seg[1:]):
  g = max(0, b[0] - a[1])
  gap_after[(a[2], b[2])] += g; gap_cnt[(a[2], b[2])] += 1
  hist[min(9, g // 10000)] += 1
for s, e, n in seg:
  bt[n] += e - s; bc[n] += 1
print("gap histogram (10us bins):", sorted(hist.items()))
print("--- busy by kernel")
for n, t in bt.most_common(25): print("%9.2f ms %6d  avg %8.1f us  %s" % (t / 1e6, bc[n], t / 1e3 / bc[n], n))
print("--- idle by (prev -> next)")
for k, t in gap_after.most_common(30): print("%9.2f ms %6d  avg %7.1f us  %s -> %s" % (t / 1e6, gap_cnt[k], t / 1e3 / gap_cnt[k], k[0], k[1]))
