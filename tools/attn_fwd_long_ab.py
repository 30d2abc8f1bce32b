"""A/B of the operand prefetch in the long-sequence attention kernels (csrc/attention3.hip, 28 / 36 key fragments, one
workgroup per CU and 2 waves per SIMD): A3_PIPE_LONG = 1 (forward: the K rows of the next pair of key fragments are requested
before the current pair's MFMAs) A3_DQ_PIPE = 1 (one-sweep dQ kernel: K / V rows of the next pair) and A4_DKV_PIPE = 1 (32-key-block
dK / dV kernel: Q / dO rows of the next query fragment).  Compile-time
switches, so every variant is its own library build:

  python tools/attn_fwd_long_ab.py build     (CPU, hipcc: tools/probes/build/libbvhip_a3<variant>.so)
  python tools/attn_fwd_long_ab.py           (GPU: forward and two-launch backward at L = 441 / 576 / 577, interleaved; checksums)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "probes", "build")


VARIANTS = {"base": [], "nopipe": ["-DA3_PIPE_LONG=0"], "fwd12": ["-DA3_FWD_LONG_NW=12"], "dq8": ["-DA3_DQ_LONG_NW=8"]}
if os.environ.get("BV_AB_ALL"):   # the measured-and-shelved prefetch variants too
  VARIANTS.update({"dq": ["-DA3_DQ_PIPE=1"], "dkv": ["-DA4_DKV_PIPE=1"]})


def lib_of(v):
  return os.path.join(OUT, f"libbvhip_a3{v}.so")


def build():
  from big_vision_amd import build as B
  B.build(force=False, verbose=False)
  os.makedirs(OUT, exist_ok=True)
  objdir = os.path.join(B.HERE, "build")
  others = [os.path.join(objdir, os.path.splitext(s)[0] + ".o") for s in B.SOURCES if s != "attention3.hip"]
  for v, flags in VARIANTS.items():
    obj = os.path.join(OUT, f"attention3_{v}.o")
    subprocess.check_call(["hipcc", *B.FLAGS, *flags, "-x", "hip", "-c", os.path.join(B.CSRC, "attention3.hip"), "-o", obj])
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj, *others, "-o", lib_of(v)])
    print("built", lib_of(v))


def child(v):
  from big_vision_amd import _lib
  _lib.LIB_PATH = lib_of(v)
  import torch
  from big_vision_amd import ops
  dev = torch.device("cuda:0")
  res = {}
  for n, L, H in ((256, 441, 16), (256, 576, 16), (128, 577, 12)):
    g = torch.Generator(device="cpu").manual_seed(L)
    qkv = torch.randn(n * L, 3 * H * 64, generator=g).to(torch.bfloat16).to(dev)
    for _ in range(3):
      o, lse = ops.attn_fwd(qkv, n, L, H)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
      o, lse = ops.attn_fwd(qkv, n, L, H)
    e1.record(); torch.cuda.synchronize()
    d_o = torch.randn(n * L, H * 64, generator=g).to(torch.bfloat16).to(dev)
    dq = torch.empty_like(qkv)
    db = torch.zeros(3 * H * 64, device=dev)
    for _ in range(3):
      ops.attn_bwd(qkv, o, d_o, lse, n, L, H, dqkv=dq, dbias=db)
    torch.cuda.synchronize()
    b0 = torch.cuda.Event(enable_timing=True); b1 = torch.cuda.Event(enable_timing=True)
    b0.record()
    for _ in range(10):
      ops.attn_bwd(qkv, o, d_o, lse, n, L, H, dqkv=dq, dbias=db)
    b1.record(); torch.cuda.synchronize()
    res[f"n={n} L={L} H={H}"] = dict(us=1e2 * e0.elapsed_time(e1), bwd_us=1e2 * b0.elapsed_time(b1),
                                     ohash=int(o.view(torch.int16).to(torch.int64).sum().item()), lsesum=float(lse.double().sum().item()),
                                     dhash=int(dq.view(torch.int16).to(torch.int64).sum().item()))
  print("RESULT " + json.dumps(res), flush=True)


def main():
  if sys.argv[1:2] == ["build"]:
    return build()
  if sys.argv[1:2] == ["child"]:
    return child(sys.argv[2])
  rows = {}
  for rep in range(3):
    for v in VARIANTS:
      out = subprocess.run([sys.executable, __file__, "child", v], capture_output=True, text=True)
      line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
      if not line:
        print(f"variant {v} failed:\n{out.stderr[-2000:]}", flush=True)
        continue
      for k, d in json.loads(line[0][7:]).items():
        rows.setdefault(k, {}).setdefault(v, []).append(d)
  print("# us per launch, three interleaved repetitions; variants: " + ", ".join(f"{v} ({' '.join(f) or 'what ships'})" for v, f in VARIANTS.items()))
  for k, r in rows.items():
    base = r.get("base", [{}])[0]
    for v in VARIANTS:
      ds = r.get(v, [])
      if not ds:
        continue
      same = ds[0]["ohash"] == base.get("ohash") and ds[0]["lsesum"] == base.get("lsesum") and ds[0]["dhash"] == base.get("dhash")
      print(f"{k:20s} {v:5s} | fwd " + " / ".join(f"{d['us']:7.1f}" for d in ds) + " | bwd (dQ + dK,dV launches) "
            + " / ".join(f"{d['bwd_us']:7.1f}" for d in ds) + f" | bits equal to base: {same}")


if __name__ == "__main__":
  main()
