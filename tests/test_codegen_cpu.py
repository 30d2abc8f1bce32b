"""Code-generation guard for the hot kernels (cross-compiled here, no GPU needed): no register
spills and NO SCRATCH MEMORY.  A private array that the optimiser ends up indexing at run time
(e.g. accumulator fragments selected by a merged `switch`) silently moves to scratch: the kernel
stays bit-exact and runs at half speed, with scratch loads/stores in the same in-order VMEM queue
as the LDS DMA (this happened to tools/probes/gemm_dbuf_probe.hip).  hipcc's resource remarks
make it visible at build time."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "big_vision_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# file -> kernels that may use scratch (general fallbacks outside the default dispatch)
ALLOWED_SCRATCH = {}


def _resources(src, tmp_path):
  cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DNDEBUG", "-x", "hip", "-c",
         os.path.join(CSRC, src), "-o", str(tmp_path / "x.o"), "-Rpass-analysis=kernel-resource-usage"]
  out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True).stdout
  res, name = {}, None
  for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
      name = m.group(1)
      res[name] = {}
    for key in ("VGPRs Spill", "SGPRs Spill", "ScratchSize [bytes/lane]", "VGPRs"):
      m = re.search(re.escape(key) + r": (\d+)", line)
      if m and name and key not in res[name]:
        res[name][key] = int(m.group(1))
  return res


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src", ["gemm256.hip", "attention3.hip", "attention_dh.hip", "layernorm.hip", "gemm_bf16.hip", "attention.hip",
                                 "elementwise.hip", "loss_optim.hip", "dropout.hip"])
def test_no_spills_no_scratch(src, tmp_path):
  res = _resources(src, tmp_path)
  assert res, "no kernels found in the compiler remarks"
  for name, r in res.items():
    if any(a in name for a in ALLOWED_SCRATCH.get(src, ())):
      continue   # long-sequence general fallbacks (L > 224 with the fast path switched off)
    # (SGPR spills go to VGPR lanes, not to memory: tolerated - attn2_fwd_kernel<14,2> has 160)
    assert r.get("VGPRs Spill", 0) == 0, f"{src}:{name} spills vector registers: {r}"
    assert r.get("ScratchSize [bytes/lane]", 0) == 0, f"{src}:{name} uses scratch memory: {r}"
    assert r.get("VGPRs", 0) <= 256, (name, r)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_asm_loads_are_not_touched_before_their_wait(tmp_path):
  """gemm256r_kernel hides its bias / residual loads from hipcc in inline asm (a load hipcc counts
  itself would drain the LDS-DMA queue).  hipcc does not know such a destination is in flight: a
  register copy placed between the asm load and the counted wait reads stale data (this happened:
  the copies hipcc emits to satisfy a "+v" tie landed BEFORE the tied s_waitcnt on one branch).
  tools/audit_asm_loads.py scans the ISA for any instruction that touches an asm-loaded register
  before the next vmcnt wait."""
  import sys
  sys.path.insert(0, os.path.join(ROOT, "tools"))
  import audit_asm_loads
  out = tmp_path / "g256.s"
  subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DNDEBUG", "-x", "hip", "-S",
                  "--cuda-device-only", os.path.join(CSRC, "gemm256.hip"), "-o", str(out)],
                 check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
  assert audit_asm_loads.audit(str(out), "gemm256r") == 0


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src", ["attention3.hip", "attention_dh.hip", "gemm256.hip"])
def test_no_mfma_result_is_read_straight_across_a_branch(src, tmp_path):
  """hipcc (ROCm 7.2) pads the MFMA-write -> VALU-read hazard inside a basic block, but emitted no wait
  states when the reader was the first instruction of a block entered by a taken branch right behind
  the MFMA (attention3 dQ sweep 1: delta came out 30 % wrong, on hardware only).  The kernels avoid
  the pattern (branch-free masks, explicit s_nop drains behind the last MFMAs of a loop);
  tools/audit_mfma_edges.py checks the ISA for it."""
  import sys
  sys.path.insert(0, os.path.join(ROOT, "tools"))
  import audit_mfma_edges
  out = tmp_path / "k.s"
  subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DNDEBUG", "-x", "hip", "-S",
                  "--cuda-device-only", os.path.join(CSRC, src), "-o", str(out)],
                 check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
  assert audit_mfma_edges.audit(str(out)) == 0


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_library_build_holds_no_probe_variant_of_the_gemm(tmp_path):
  """The bottleneck-ablation modes of gemm256_kernel (template argument PROBE != 0: timing-only paths, several of
  them with wrong results by design) are compiled only when a translation unit under tools/probes/ defines
  BV_GEMM256_PROBES.  The library build must (a) instantiate PROBE = 0 only and (b) refuse any other value."""
  res = _resources("gemm256.hip", tmp_path)
  names = [n for n in res if "gemm256_kernel" in n]
  assert names
  for n in names:   # _ZN..14gemm256_kernelILb<KM>ELi<PROBE>ELi<EPI>ELb<OUTF32>EEEv...
    m = re.search(r"gemm256_kernelILb[01]ELi(\d+)E", n)
    assert m and m.group(1) == "0", n
  src = open(os.path.join(CSRC, "gemm256.hip")).read()
  assert "#define BV_GEMM256_PROBES" not in src
  tu = tmp_path / "probe_without_macro.hip"
  tu.write_text('#include "gemm256.hip"\n'
                "template __global__ void gemm256_kernel<true, 5, BV_EPI_NONE, false>(G256Params);\n")
  r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DNDEBUG", "--cuda-device-only",
                      "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-x", "hip", "-c", str(tu), "-o",
                      str(tmp_path / "p.o")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
  assert r.returncode != 0 and "probe variants are compiled only under tools/probes" in r.stdout, r.stdout[-2000:]
