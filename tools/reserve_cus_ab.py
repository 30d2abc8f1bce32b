"""A/B: the persistent GEMMs of BOTH towers leave R CUs free (ctx option gemm_reserve_cus) while the towers run on two
streams - do the other tower's HBM-bound kernels (LayerNorm, attention) on the free CUs beside a GEMM buy more than
the GEMMs lose?  ms per step, interleaved, headline (4096 pairs in micro-batches of 2048) and rank shapes.  GPU only.

  python tools/reserve_cus_ab.py [n ...]      (default 512 4096)
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from big_vision_amd import ops


def run(dev, n, reserve, streams=2, steps=4):
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  model = two_towers.Model(image=bench.IMAGE_CFG, text=bench.TEXT_CFG, out_dim=(None, bench.EMB), temperature_init=10.0, bias_init=-10.0)
  config = bench.make_config(20_000)
  config.tower_streams = streams
  config.microbatch = bench.MICRO
  image, text = bench.synthetic_batch(n, dev, seed=1)
  state, _ = siglip.make_train_state(model, config, (n, bench.RES, bench.RES, 3), (n, bench.SEQ), rng=0, total_steps=20_000, device=dev)
  fn = siglip.make_update_fn(model, config)
  batch = {"image": image, "labels": text}
  old = ops.ctx_set("gemm_reserve_cus", reserve)
  try:
    for _ in range(2):
      state, meas = fn(state, None, batch)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
      state, meas = fn(state, None, batch)
    e1.record(); torch.cuda.synchronize()
  finally:
    ops.ctx_set("gemm_reserve_cus", old)
  ms = e0.elapsed_time(e1) / steps
  loss = meas["training_loss"].item()
  del state, fn, model, batch, image, text, meas
  import gc
  gc.collect()
  torch.cuda.empty_cache()
  return ms, loss


def main():
  dev = torch.device("cuda:0")
  sweep = [int(r) for r in os.environ.get("RESERVE_SWEEP", "0,8,16,32,48").split(",")]
  for n in [int(a) for a in sys.argv[1:]] or [512, 4096]:
    for rep in range(2):
      for reserve in sweep:
        ms, loss = run(dev, n, reserve, steps=6 if n <= 1024 else 4)
        print(f"n = {n:5d}  two streams, gemm_reserve_cus = {reserve:3d}: {ms:8.2f} ms per step  loss {loss:.6f}", flush=True)


if __name__ == "__main__":
  main()
