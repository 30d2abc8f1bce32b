"""A/B of HIP stream priorities for the two-stream towers (config.tower_streams = 2): does it matter which tower's
launches the dispatcher prefers when both streams have ready workgroups?  Variants, interleaved, full model:

  eq        both streams at the default priority (what ships)
  txt_high  the text tower's side stream at high priority (BV_SIDE_STREAM_PRIORITY=-1)
  img_high  the whole step enqueued from a high-priority main stream (image tower, loss, optimizer), side stream default

  python tools/stream_priority_ab.py [n ...]      (default 512 4096; GPU only)
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def run(dev, n, variant, steps):
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  os.environ["BV_SIDE_STREAM_PRIORITY"] = "-1" if variant == "txt_high" else "0"
  main = torch.cuda.Stream(device=dev, priority=-1) if variant == "img_high" else torch.cuda.current_stream(dev)
  with torch.cuda.stream(main):
    model = two_towers.Model(image=bench.IMAGE_CFG, text=bench.TEXT_CFG, out_dim=(None, bench.EMB), temperature_init=10.0, bias_init=-10.0)
    config = bench.make_config(20_000)
    config.tower_streams = 2
    config.microbatch = bench.MICRO
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
  del state, fn, model, batch, image, text
  torch.cuda.empty_cache()
  return ms, loss


def main():
  dev = torch.device("cuda:0")
  torch.cuda.set_device(dev)
  print("priority range (least, greatest):", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a", flush=True)
  for n in [int(a) for a in sys.argv[1:]] or [512, 4096]:
    steps = int(os.environ.get("BV_AB_STEPS", "0")) or (8 if n <= 1024 else 4)
    for rep in range(int(os.environ.get("BV_AB_REPS", "2"))):
      for variant in os.environ.get("BV_AB_VARIANTS", "eq,txt_high,img_high").split(","):
        ms, loss = run(dev, n, variant, steps)
        print(f"n = {n:5d}  {variant:9s}: {ms:8.2f} ms per step  loss {loss:.6f}", flush=True)


if __name__ == "__main__":
  main()
