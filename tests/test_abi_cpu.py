"""The C-ABI library loads on a CPU-only host and exports exactly what
include/bvhip.h declares; the product path refuses to run without a GPU
(no CPU fallback).  No compute calls are made here."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bvhip.h")


def _header_symbols():
  src = open(HEADER).read()
  src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
  return sorted(set(re.findall(r"^(?:int|long|const char\*)\s+(bv_\w+)\s*\(", src, re.M)))


@pytest.fixture(scope="module")
def lib():
  from big_vision_amd import build, _lib
  build.build(verbose=False)          # hipcc cross-compiles gfx950 without a GPU
  return _lib.load()


def test_header_declares_entry_points():
  syms = _header_symbols()
  assert "bv_gemm_bf16" in syms and "bv_siglip_loss" in syms and "bv_adam_step" in syms
  assert len(syms) >= 20


def test_library_exports_every_declared_symbol(lib):
  raw = ctypes.CDLL(os.path.join(ROOT, "big_vision_amd", "libbvhip.so"))
  for s in _header_symbols():
    assert hasattr(raw, s), f"libbvhip.so does not export {s} (declared in include/bvhip.h)"


def test_python_prototypes_cover_the_header(lib):
  from big_vision_amd import _lib
  assert sorted(list(_lib.PROTOTYPES) + ["bv_last_error"]) == _header_symbols()
  assert lib.bv_version() == 1


def test_header_cites_reference_call_sites():
  src = open(HEADER).read()
  for cite in ("models/vit.py:", "trainers/proj/image_text/siglip.py:", "optax.py:",
               "models/proj/image_text/two_towers.py:", "models/proj/image_text/text_transformer.py:"):
    assert cite in src, cite


def test_product_path_has_no_cpu_fallback():
  from big_vision_amd import ops
  a = torch.zeros(8, 8, dtype=torch.bfloat16)
  with pytest.raises(RuntimeError, match="GPU"):
    ops.gemm(a, a)
  with pytest.raises(RuntimeError, match="GPU"):
    ops.layernorm_fwd(torch.zeros(4, 8), torch.ones(8), torch.zeros(8), rows=4, D=8)


def test_product_never_imports_the_oracle():
  pkg = os.path.join(ROOT, "big_vision_amd")
  for d, _, files in os.walk(pkg):
    for f in files:
      if f.endswith((".py", ".hip", ".cpp", ".h")):
        txt = open(os.path.join(d, f)).read()
        assert "bv_oracle" not in txt and "import oracle" not in txt, os.path.join(d, f)


def test_comm_layer_binds_rccl_at_run_time():
  """bv_comm_* (csrc/comm.cpp) has no link-time RCCL dependency: the library loads on a host without a GPU and the
  first call binds the RCCL that is already in the process (PyTorch's) - one RCCL per process."""
  import ctypes
  import subprocess
  from big_vision_amd import _lib
  lib = _lib.load()
  v = ctypes.c_int(0)
  assert lib.bv_comm_version(ctypes.byref(v)) == 0 and v.value >= 20000, lib.bv_last_error()
  needed = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
  assert "rccl" not in needed.lower(), "libbvhip.so must not link RCCL"
  assert lib.bv_comm_init(None, 0, 1, None) != 0 and b"bad arguments" in lib.bv_last_error()
