"""A/B of config.tower_streams (1 = one stream, 2 = text tower on a side stream) at the rank shapes of the headline
(512 / 1024 pairs per GPU, full model): ms per step, interleaved.  GPU only.

  python tools/tower_streams_ab.py [n ...]      (default 512 1024)
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def run(dev, n, streams, steps=6):
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  model = two_towers.Model(image=bench.IMAGE_CFG, text=bench.TEXT_CFG, out_dim=(None, bench.EMB), temperature_init=10.0, bias_init=-10.0)
  config = bench.make_config(20_000)
  config.tower_streams = streams
  image, text = bench.synthetic_batch(n, dev, seed=1)
  state, _ = siglip.make_train_state(model, config, (n, bench.RES, bench.RES, 3), (n, bench.SEQ), rng=0, total_steps=20_000, device=dev)
  fn = siglip.make_update_fn(model, config)
  batch = {"image": image, "labels": text}
  for _ in range(2):
    state, meas = fn(state, None, batch)
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(steps):
    state, meas = fn(state, None, batch)
  e1.record(); torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / steps
  loss = meas["training_loss"].item()
  peak = torch.cuda.max_memory_allocated(dev) / 1e9
  del state, fn, model
  torch.cuda.empty_cache()
  return ms, loss, peak


def main():
  dev = torch.device("cuda:0")
  for n in [int(a) for a in sys.argv[1:]] or [512, 1024]:
    for rep in range(2):
      for streams in (1, 2):
        ms, loss, peak = run(dev, n, streams)
        print(f"n = {n:5d}  tower_streams = {streams}: {ms:8.2f} ms per step  loss {loss:.6f}  peak {peak:.1f} GB", flush=True)


if __name__ == "__main__":
  main()
