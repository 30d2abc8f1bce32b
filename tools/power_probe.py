"""Shader clock and package power while the k-major GEMM runs on all 256 CUs vs on 128 (BV_OPT_GEMM_RESERVE_CUS = 128):
evidence for the power-envelope reading of profiles/NOTES_r05.md (half the grid delivers two thirds of the throughput).
Polls `rocm-smi --showclocks --showpower --json` from a thread while a stream of launches keeps the GPU busy.  GPU only.

  python tools/power_probe.py > profiles/rNN_power_probe.txt
"""
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from big_vision_amd import ops

dev = torch.device("cuda:0")
BF16 = torch.bfloat16


def poll(stop, out):
  while not stop.is_set():
    try:
      r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10)
      d = json.loads(r.stdout)
      card = d[sorted(d)[0]]
      sclk = next((v for k, v in card.items() if "sclk" in k.lower()), None)
      pwr = next((v for k, v in card.items() if "power" in k.lower() and "W" in k), None)
      out.append((time.time(), sclk, pwr))
    except Exception as e:   # keep polling
      out.append((time.time(), f"error {type(e).__name__}", None))
    time.sleep(0.2)


def phase(name, seconds, fn):
  stop, out = threading.Event(), []
  th = threading.Thread(target=poll, args=(stop, out))
  torch.cuda.synchronize()
  th.start()
  t0 = time.time()
  n = 0
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  while time.time() - t0 < seconds:
    for _ in range(20):
      fn()
    n += 20
    torch.cuda.synchronize()
  e1.record(); torch.cuda.synchronize()
  stop.set(); th.join()
  us = e0.elapsed_time(e1) * 1e3 / max(1, n)
  print(f"{name}: {us:8.1f} us per launch")
  for t, sclk, pwr in out[1:]:
    print(f"    t+{t - t0:4.1f}s  sclk {sclk}  power {pwr}")
  return us


def main():
  T, N, K = 401408, 2304, 768
  a = torch.randn(T, K, device=dev).to(BF16)
  w = (torch.randn(N, K, device=dev) * 0.02).to(BF16)
  out = torch.empty(T, N, device=dev, dtype=BF16)
  fn = lambda: ops.gemm(a, w, a_kmajor=True, b_kmajor=True, out=out)
  fl = 2.0 * T * N * K
  phase("idle (no launches)", 1.5, lambda: None)
  u256 = phase("k-major GEMM 401408 x 2304 x 768 on 256 CUs", 4.0, fn)
  with ops.option("gemm_reserve_cus", 128):
    u128 = phase("the same on 128 CUs (128 reserved)", 4.0, fn)
  print(f"TFLOP/s: 256 CUs {fl / u256 / 1e6:.0f}, 128 CUs {fl / u128 / 1e6:.0f} = {u256 / u128:.2f} of the full grid's throughput")


if __name__ == "__main__":
  main()
