"""A/B of the cache policy of gemm256's operand DMA (`global_load_lds` aux field, BV_GLDS_AUX_A / _B in csrc/gemm256.hip:
0 = default, 2 = nt) on the GEMM instances of the headline step.  The policy is an instruction immediate, so every
combination is its own library build:

  python tools/gemm_cachepolicy_ab.py build      (CPU, hipcc: tools/probes/build/libbvhip_a<A>_b<B>.so)
  python tools/gemm_cachepolicy_ab.py            (GPU: every variant in its own process, interleaved, table of us per launch)
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "probes", "build")
VARIANTS = [(0, 0), (2, 0), (0, 2), (2, 2)]


def lib_of(a, b):
  return os.path.join(OUT, f"libbvhip_a{a}_b{b}.so")


def build():
  from big_vision_amd import build as B
  B.build(force=False, verbose=False)       # the product's objects are the other translation units of every variant
  os.makedirs(OUT, exist_ok=True)
  objdir = os.path.join(B.HERE, "build")
  others = [os.path.join(objdir, os.path.splitext(s)[0] + ".o") for s in B.SOURCES if s != "gemm256.hip"]
  procs = []
  for a, b in VARIANTS:
    obj = os.path.join(OUT, f"gemm256_a{a}_b{b}.o")
    cmd = ["hipcc", *B.FLAGS, f"-DBV_GLDS_AUX_A={a}", f"-DBV_GLDS_AUX_B={b}", "-x", "hip", "-c",
           os.path.join(B.CSRC, "gemm256.hip"), "-o", obj]
    procs.append((a, b, obj, subprocess.Popen(cmd)))
  for a, b, obj, p in procs:
    assert p.wait() == 0, (a, b)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj, *others, "-o", lib_of(a, b)])
    print("built", lib_of(a, b))


def child(a, b):
  from big_vision_amd import _lib
  _lib.LIB_PATH = lib_of(a, b)
  import torch
  from big_vision_amd import ops
  dev = torch.device("cuda:0")
  BF16, F32 = torch.bfloat16, torch.float32

  def timeit(fn, iters=6, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters

  D, M = 768, 3072
  res = {}
  for T in (401408, 131072):
    x = torch.randn(T, D, device=dev).to(BF16)
    hM = torch.randn(T, M, device=dev).to(BF16)
    x3 = torch.randn(T, 3 * D, device=dev).to(BF16)
    r32 = torch.randn(T, D, device=dev)
    bias = {n: torch.randn(n, device=dev) for n in (D, 3 * D, M)}
    w = {(n, k): (torch.randn(n, k, device=dev) * 0.02).to(BF16) for n, k in ((3 * D, D), (D, D), (M, D), (D, M), (D, 3 * D))}
    o = {n: torch.empty(T, n, device=dev, dtype=BF16) for n in (D, 3 * D, M)}
    o2 = torch.empty(T, M, device=dev, dtype=BF16)
    of = torch.empty(T, D, device=dev, dtype=F32)
    gw = {(k, n): torch.zeros(k, n, device=dev, dtype=F32) for k, n in ((D, 3 * D), (D, D), (D, M), (M, D))}
    cases = [
        ("fwd qkv bias", lambda: ops.gemm(x, w[(3 * D, D)], b_kmajor=True, bias=bias[3 * D], out=o[3 * D])),
        ("fwd out +res f32", lambda: ops.gemm(x, w[(D, D)], b_kmajor=True, bias=bias[D], out=of, epilogue=ops.EPI_RESIDUAL, aux=r32)),
        ("fwd fc1 gelu 2 outs", lambda: ops.gemm(x, w[(M, D)], b_kmajor=True, bias=bias[M], out=o[M], epilogue=ops.EPI_GELU, out2=o2)),
        ("fwd fc2 +res f32", lambda: ops.gemm(hM, w[(D, M)], b_kmajor=True, bias=bias[D], out=of, epilogue=ops.EPI_RESIDUAL, aux=r32)),
        ("dx fc2 gelu'emit", lambda: ops.gemm(x, w[(M, D)], b_kmajor=True, out=o[M], epilogue=ops.EPI_GELU_BWD_EMIT, aux=hM, out2=o2)),
        ("dx fc1", lambda: ops.gemm(hM, w[(D, M)], b_kmajor=True, out=o[D])),
        ("dx out", lambda: ops.gemm(x, w[(D, D)], b_kmajor=True, out=o[D])),
        ("dx qkv", lambda: ops.gemm(x3, w[(D, 3 * D)], b_kmajor=True, out=o[D])),
        ("dW qkv", lambda: ops.gemm(x, x3, a_kmajor=False, b_kmajor=False, out=gw[(D, 3 * D)], epilogue=ops.EPI_ATOMIC)),
        ("dW fc1", lambda: ops.gemm(x, hM, a_kmajor=False, b_kmajor=False, out=gw[(D, M)], epilogue=ops.EPI_ATOMIC)),
        ("dW fc2", lambda: ops.gemm(hM, x, a_kmajor=False, b_kmajor=False, out=gw[(M, D)], epilogue=ops.EPI_ATOMIC)),
    ]
    for name, fn in cases:
      res[f"T={T} {name}"] = timeit(fn)
    del x, hM, x3, r32, o, o2, of
    torch.cuda.empty_cache()
  print("RESULT " + json.dumps(res), flush=True)


def main():
  if sys.argv[1:2] == ["build"]:
    return build()
  if sys.argv[1:2] == ["child"]:
    return child(int(sys.argv[2]), int(sys.argv[3]))
  reps = int(os.environ.get("BV_AB_REPS", "2"))
  table = {}
  for rep in range(reps):
    for a, b in VARIANTS:
      out = subprocess.run([sys.executable, __file__, "child", str(a), str(b)], capture_output=True, text=True)
      line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
      if not line:
        print(f"variant a={a} b={b} failed:\n{out.stderr[-2000:]}", flush=True)
        continue
      for k, v in json.loads(line[0][7:]).items():
        table.setdefault(k, {}).setdefault((a, b), []).append(v)
  print("# us per launch (each repetition); columns: (aux A, aux B) = " + "  ".join(f"({a},{b})" for a, b in VARIANTS) + "; 0 = default, 2 = nt")
  tot = {v: 0.0 for v in VARIANTS}
  for k, row in table.items():
    cells = []
    for v in VARIANTS:
      xs = row.get(v, [])
      cells.append(" / ".join(f"{x:7.1f}" for x in xs))
      tot[v] += min(xs) if xs else float("nan")
    print(f"{k:34s} | " + " | ".join(cells))
  print(f"{'sum of the best of each':34s} | " + " | ".join(f"{tot[v]:9.1f}" for v in VARIANTS))


if __name__ == "__main__":
  main()
