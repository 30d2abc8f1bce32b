"""Bench lines for the BASELINE.json configs other than the headline one (bench.py = configs[2]):

  c2  ViT-B/16 image tower alone, forward + backward, batch 256                     (configs[1])
  c4  SigLIP ViT-L/16 @336 px + text-L, global batch 8192 on 8 GPUs = 1024 pairs per GPU: one
      rank's share of the step (its 1024 pairs, micro-batches of 256) on ONE GPU    (configs[3])
  c5  LiT: frozen ViT-B/16 (cls token) + trainable text-B at 16 tokens, config batch 512 on one
      GPU (text-only backward, no image-tower gradients / optimizer state)          (configs[4])
  c5b the same with the BERT-base text tower the reference config names (text_model='proj.flaxformer.bert')

One JSON line per workload (same fields as bench.py where they apply; `value` is per-GPU here because
these lines are measured on one device).  GPU only; synthetic data resident in HBM; K timed steps
between synchronisations.     python tools/bench_configs.py [c2 c4 c5] [--steps K]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def timed(fn, steps, warmup):
  """Seconds per step, and the `roofline` object of bench.py for the same timed region (the k-major
  256x256 GEMM family bracketed by HIP events on the launch stream)."""
  from big_vision_amd import _lib
  for _ in range(warmup):
    fn()
  obs = bench.GemmObserver()
  _lib.observer = obs
  torch.cuda.synchronize()
  obs.active = True
  t0 = time.perf_counter()
  for _ in range(steps):
    fn()
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / steps
  obs.active = False
  _lib.observer = None
  launches, ms, flops, nbytes = obs.summary()
  ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
  ROOF.clear()
  ROOF.update({"bound": "mfma", "kernel": bench.DOMINANT_KERNEL, "achieved": ach, "peak": bench.BF16_DENSE_PEAK_TFLOPS,
               "unit": "TFLOP/s", "frac": ach / bench.BF16_DENSE_PEAK_TFLOPS, "traffic": None,
               "algorithmic_bytes_per_launch": nbytes / max(1, launches), "launches": launches,
               "avg_launch_us": 1e3 * ms / max(1, launches), "share_of_step_time": ms / (1e3 * dt * steps)})
  return dt


ROOF = {}   # roofline object of the last timed() call


STREAM = "float32"   # --residual-stream


def c2(dev, steps):
  from big_vision_amd import engine as E
  from big_vision_amd.models import vit
  from big_vision_amd.params import ParamStore
  E.set_residual_stream(STREAM)
  n, res = 256, 224
  model = vit.Model(None, variant="B/16", pool_type="map")
  hw = model.grid((n, res, res, 3))
  store = ParamStore(model.entries("", hw), dev)
  store.init_random(0); store.refresh_shadow(); store.want_grads = True
  image = torch.rand((n, res, res, 3), device=dev) * 2 - 1
  ex = model.executor(store, "", hw)

  def step():
    store.zero_grad()
    z, _, ctx = ex.fwd(image, save=True)
    ex.bwd(ctx, (z / n).contiguous())          # dL/dz of L = 0.5 mean |z|^2
  dt = timed(step, steps, 2)
  flops = 3 * 35.42e9 * n                      # fwd + bwd matmul FLOPs of the tower
  return {"metric": "images/sec, ViT-B/16 image tower forward+backward, batch 256 (BASELINE configs[1])",
          "value": n / dt, "unit": "images/s", "ms_per_step": 1e3 * dt, "tflops_algorithmic": flops / dt / 1e12,
          "config": {"workload": "ViT-B/16@224 MAP tower, fwd+bwd, no optimizer", "batch": n, "residual_stream": STREAM}}


def _siglip(dev, steps, image_cfg, text_cfg, emb, n, res, seq, micro, schedule=None, label="", text_model=None, vocab=32_000):
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  model = two_towers.Model(image=image_cfg, text=text_cfg, out_dim=(None, emb), temperature_init=10.0,
                           bias_init=-10.0 if schedule is None else -2.71,
                           **({"text_model": text_model} if text_model else {}))
  config = bench.make_config(20_000)
  config.microbatch = micro
  config.residual_stream = STREAM
  if schedule is not None:
    config.schedule = schedule
  g = torch.Generator(device=dev).manual_seed(1)
  image = torch.rand((n, res, res, 3), generator=g, device=dev) * 2 - 1
  text = torch.randint(2, vocab, (n, seq), generator=g, device=dev, dtype=torch.int32)
  state, _ = siglip.make_train_state(model, config, (n, res, res, 3), (n, seq), rng=0, total_steps=20_000, device=dev)
  fn = siglip.make_update_fn(model, config)
  box = {"s": state}

  def step():
    box["s"], box["m"] = fn(box["s"], None, {"image": image, "labels": text})
  dt = timed(step, steps, 2)
  siglip.check_finite(box["m"])
  return {"value": n / dt, "unit": "pairs/s", "ms_per_step": 1e3 * dt,
          "config": {"workload": label, "per_gpu_batch": n, "microbatch": micro, "residual_stream": STREAM,
                     "final_loss": float(box["m"]["training_loss"].item())}}


def c4(dev, steps):
  r = _siglip(dev, steps, dict(variant="L/16", pool_type="map"), dict(variant="L", vocab_size=32_000), 1024,
              n=1024, res=336, seq=64, micro=256,
              label="SigLIP ViT-L/16@336 + text-L: one rank's 1024 pairs of the global batch 8192 (loss over the local "
                    "1024 only: no peers on a single device), micro-batches of 256, Adam+clip+wd+cosine")
  r["metric"] = "image-text pairs/sec per GPU, SigLIP ViT-L/16@336 training step at 1024 pairs per GPU (BASELINE configs[3])"
  return r


def c5(dev, steps):
  sched = [("img/.*", None), (".*", dict(decay_type="cosine", warmup_steps=150))]
  r = _siglip(dev, steps, dict(variant="B/16", pool_type="tok", head_zeroinit=False), dict(variant="B", vocab_size=32_000),
              768, n=512, res=224, seq=16, micro=2048, schedule=sched,
              label="LiT (siglip_lit_coco.py): frozen ViT-B/16 cls-token tower + trainable text-B, 16 tokens, batch 512, "
                    "text-only backward")
  r["metric"] = "image-text pairs/sec, LiT locked-image step, batch 512 (BASELINE configs[4])"
  return r


def c5b(dev, steps):
  """The literal siglip_lit_coco.py: text_model='proj.flaxformer.bert', config 'base' (:78,84-87)."""
  sched = [("img/.*", None), (".*", dict(decay_type="cosine", warmup_steps=150))]
  r = _siglip(dev, steps, dict(variant="B/16", pool_type="tok", head_zeroinit=False), dict(config="base", head_zeroinit=False),
              768, n=512, res=224, seq=16, micro=2048, schedule=sched, text_model="proj.flaxformer.bert", vocab=30522,
              label="LiT (siglip_lit_coco.py as written): frozen ViT-B/16 cls-token tower + trainable BERT-base text tower, "
                    "16 tokens, batch 512, text-only backward")
  r["metric"] = "image-text pairs/sec, LiT locked-image step with the BERT-base text tower, batch 512 (BASELINE configs[4])"
  return r


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("workloads", nargs="*", default=["c2", "c4", "c5", "c5b"])
  ap.add_argument("--steps", type=int, default=5)
  ap.add_argument("--residual-stream", default="float32", choices=("float32", "bfloat16"))
  a = ap.parse_args()
  global STREAM
  STREAM = a.residual_stream
  dev = torch.device("cuda", 0)
  torch.cuda.set_device(dev)
  for w in a.workloads:
    r = {"c2": c2, "c4": c4, "c5": c5, "c5b": c5b}[w](dev, a.steps)
    r.update(n_gpus=1, steps=a.steps, dtype="bf16", data="synthetic", higher_is_better=True, roofline=dict(ROOF))
    print(json.dumps(r), flush=True)
    torch.cuda.empty_cache()


if __name__ == "__main__":
  main()
