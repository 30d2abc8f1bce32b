"""ISA audit for kernels that hide VGPR-destination loads from hipcc in inline asm (gemm256r_kernel):
between an asm `global_load_dwordx4 vdst, ...` and the next `s_waitcnt vmcnt` nothing may READ or
COPY vdst (hipcc does not know the load is in flight; a tie-induced v_mov before the wait copies
stale registers).  Linear scan of `hipcc -S` output, conservative: a register stays "in flight" until
the next vmcnt wait of any count on the fall-through path; labels keep the set (branch targets are
visited in text order).  Usage: python tools/audit_asm_loads.py file.s [kernel-substring]"""
import re
import sys


def regs(tok):
  m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
  if m:
    return set(range(int(m.group(1)), int(m.group(2)) + 1))
  m = re.fullmatch(r"v(\d+)", tok)
  return {int(m.group(1))} if m else set()


def audit(path, want=""):
  bad = 0
  name, inflight, in_asm = None, {}, False
  for ln, line in enumerate(open(path), 1):
    t = line.strip()
    m = re.match(r"^(_Z\w+):", t)
    if m:
      name, inflight = m.group(1), {}
      continue
    if not name or want not in name:
      continue
    if t.startswith(";;#ASMSTART"):
      in_asm = True
      continue
    if t.startswith(";;#ASMEND"):
      in_asm = False
      continue
    if not t or t.startswith((";", ".")):
      continue
    op, _, rest = t.partition(" ")
    toks = [x.strip() for x in rest.split(",")]
    if op == "s_waitcnt" and "vmcnt" in rest:
      inflight = {}
      continue
    if in_asm and op == "global_load_dwordx4":
      for r in regs(toks[0]):
        inflight[r] = ln
      continue
    used = set()
    for x in toks:
      for y in x.split():
        used |= regs(y)
    hit = used & set(inflight)
    if hit and not op.startswith("global_load_dwordx4"):
      bad += 1
      print(f"{path}:{ln}: {name[:60]}: `{t}` touches v{sorted(hit)[:4]} loaded by asm at line {inflight[sorted(hit)[0]]} with no vmcnt wait in between")
  return bad


if __name__ == "__main__":
  n = audit(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
  print("audit:", "clean" if n == 0 else f"{n} suspicious instruction(s)")
  sys.exit(1 if n else 0)
