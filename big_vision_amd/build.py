"""Build libbvhip.so (hand-written HIP kernels for gfx950) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only
build container; the resulting big_vision_amd/libbvhip.so travels to the GPU box
with the repo snapshot.

Build-time requirements: hipcc and the ROCm headers, including <rccl/rccl.h> (csrc/comm.cpp takes the
ncclComm_t / ncclDataType_t types and the prototypes of the eight entry points it binds from it).  There
is NO link-time RCCL dependency: comm.cpp dlopen()s librccl.so at the first bv_comm_* call, so the
library loads on hosts without RCCL as long as nothing calls those entry points.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbvhip.so")
SOURCES = ["c_api.cpp", "comm.cpp", "gemm_bf16.hip", "gemm256.hip", "attention.hip", "attention3.hip", "attention5.hip", "attention_dh.hip", "layernorm.hip",
           "elementwise.hip", "loss_optim.hip", "adafactor.hip", "dropout.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
         "-DNDEBUG"]


def _stale():
  if not os.path.exists(LIB):
    return True
  t = os.path.getmtime(LIB)
  deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
  deps.append(os.path.join(HERE, "..", "include", "bvhip.h"))
  return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
  hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
  if not force and not _stale():
    return LIB
  if not os.path.exists(hipcc):
    raise RuntimeError("hipcc not found: cannot build libbvhip.so")
  objdir = os.path.join(HERE, "build")
  os.makedirs(objdir, exist_ok=True)
  objs = []
  procs = []
  headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
  headers.append(os.path.join(HERE, "..", "include", "bvhip.h"))
  hdr_t = max(os.path.getmtime(h) for h in headers)
  for src in SOURCES:
    obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
    objs.append(obj)
    # per-object staleness: only the sources that changed (or everything, after a header edit)
    if (not force and os.path.exists(obj) and
        os.path.getmtime(obj) > max(hdr_t, os.path.getmtime(os.path.join(CSRC, src)))):
      continue
    cmd = [hipcc, *FLAGS, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
      print(" ".join(cmd), flush=True)
    procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
  for src, p in procs:
    out, _ = p.communicate()
    if p.returncode != 0:
      sys.stderr.write(out.decode())
      raise RuntimeError(f"hipcc failed on {src}")
  cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
  if verbose:
    print(" ".join(cmd), flush=True)
  subprocess.check_call(cmd)
  return LIB


if __name__ == "__main__":
  build(force="--force" in sys.argv)
  print("built", LIB)
