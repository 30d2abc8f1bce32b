"""ISA audit: an MFMA result consumed by a non-MFMA instruction straight across a control-flow edge.
hipcc (ROCm 7.2) pads the XDL-write -> VALU-read hazard inside a basic block, but was seen to emit
NO wait states when the reader is the first instruction of a block entered by a taken branch placed
right behind the MFMA (attention3 dQ, sweep 1: the delta FMAs read stale registers).  Flags every
branch whose preceding <= WINDOW instructions contain a v_mfma whose destination is read by a
non-MFMA instruction among the first <= WINDOW instructions of the branch target (and of the
fall-through block).  Usage: python tools/audit_mfma_edges.py file.s [kernel-substring]"""
import re
import sys

WINDOW = 4


def vregs(tok):
  m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
  if m:
    return set(range(int(m.group(1)), int(m.group(2)) + 1))
  m = re.fullmatch(r"-?v(\d+)", tok)
  return {int(m.group(1))} if m else set()


def parse(path, want):
  kernels, name, ins = {}, None, []
  for ln, line in enumerate(open(path), 1):
    t = line.split(";")[0].strip()
    m = re.match(r"^(_Z\w+):", t)
    if m:
      name = m.group(1)
      ins = kernels.setdefault(name, [])
      continue
    if name is None or want not in name or not t:
      continue
    m = re.match(r"^(\.LBB\w+):", t)
    if m:
      ins.append((ln, "label", m.group(1), []))
      continue
    if t.startswith("."):
      continue
    op, _, rest = t.partition(" ")
    toks = [x.strip() for x in rest.replace(",", " ").split()]
    ins.append((ln, op, rest, toks))
  return kernels


def audit(path, want=""):
  bad = 0
  for name, ins in parse(path, want).items():
    labels = {x[2]: i for i, x in enumerate(ins) if x[1] == "label"}
    for i, (ln, op, rest, toks) in enumerate(ins):
      if not op.startswith(("s_cbranch", "s_branch")):
        continue
      # MFMA destinations written within WINDOW real instructions before the branch
      dests, k, j, pre = {}, 0, i - 1, 0   # dest register -> wait states between its MFMA and the branch
      while j >= 0 and k < WINDOW:
        if ins[j][1] == "label":
          break
        if ins[j][1].startswith("v_mfma"):
          for r in vregs(ins[j][3][0]):
            dests.setdefault(r, pre)
          pre += 4     # an MFMA behind it occupies the pipe for at least one pass group
        elif ins[j][1].startswith("s_nop"):
          pre += (int(ins[j][3][0]) + 1) if ins[j][3] else 1
        elif not ins[j][1].startswith("s_waitcnt"):
          k += 1
          pre += 1
        j -= 1
      if not dests:
        continue
      starts = [labels.get(toks[0] if toks else "", None), i + 1]
      for st in starts:
        if st is None:
          continue
        k, j, budget = 0, st, 0
        while j < len(ins) and k < WINDOW:
          l2, op2, rest2, toks2 = ins[j]
          j += 1
          if op2 == "label":
            continue
          if op2.startswith("s_nop"):
            budget += int(toks2[0]) + 1 if toks2 else 1
            continue
          if op2.startswith(("s_", "ds_", "global_", "buffer_", "scratch_")) and not op2.startswith("s_waitcnt"):
            k += 1
            continue
          k += 1
          srcs = set()
          for x in toks2[1:]:
            srcs |= vregs(x)
          if op2.startswith("v_mfma"):
            continue     # MFMA -> MFMA chains are interlocked
          hit = {r for r in srcs if r in dests and dests[r] + budget < 12}
          if hit:
            bad += 1
            print(f"{path}:{ln}: {name[:50]}: branch `{op} {rest}` follows an MFMA writing v{sorted(hit)[:4]}; "
                  f"line {l2} `{op2} {rest2}` reads it after {min(dests[r] for r in hit) + budget} wait states")
          break_all = False
  return bad


if __name__ == "__main__":
  n = audit(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
  print("audit:", "clean" if n == 0 else f"{n} suspicious edge(s)")
  sys.exit(1 if n else 0)
