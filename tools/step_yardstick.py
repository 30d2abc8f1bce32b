"""Whole-step yardstick (tools only): the same SigLIP ViT-B/16 + text-B training step - forward, sigmoid loss,
backward, global-norm clip, AdamW - written the way a PyTorch user would run it on this GPU: HuggingFace
`SiglipModel` (the port the oracle is pinned to; default config = B/16@224 MAP + 12L/64tok/vocab 32k), eager
PyTorch-ROCm with `torch.autocast(bfloat16)`, SDPA attention, `clip_grad_norm_` + `AdamW(fused=True)`, i.e. the vendor's
library stack (hipBLASLt GEMMs, flash attention, ATen LayerNorm / elementwise kernels) behind autograd.  No
torch.compile (this image's torch has no working inductor backend for it), random-init weights, synthetic batch.

It is NOT the reference (JAX / XLA cannot run here) and not a target: it answers "what does the stock PyTorch stack
give for this step on the same MI355X", next to `python bench.py --global-batch N`.  Prints one JSON line per batch.
  python tools/step_yardstick.py [--batch 512 256] [--steps 5] [--warmup 2]"""
import argparse
import json
import sys
import time

import torch


def run(n, steps, warmup, attn):
  from transformers import SiglipConfig, SiglipModel
  dev = torch.device("cuda:0")
  cfg = SiglipConfig()
  cfg._attn_implementation = attn
  cfg.vision_config._attn_implementation = attn
  cfg.text_config._attn_implementation = attn
  torch.manual_seed(0)
  model = SiglipModel(cfg).to(dev).train()
  params = [p for p in model.parameters() if p.requires_grad]
  opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=1e-4, fused=True)
  image = torch.rand(n, 3, 224, 224, device=dev) * 2 - 1
  text = torch.randint(2, 32000, (n, 64), device=dev)

  def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
      out = model(input_ids=text, pixel_values=image, return_loss=True)
    out.loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 1.0)
    opt.step()
    return out.loss

  for _ in range(warmup):
    loss = step()
  torch.cuda.synchronize()
  torch.cuda.reset_peak_memory_stats(dev)
  t0 = time.perf_counter()
  for _ in range(steps):
    loss = step()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  return {"stack": f"HF SiglipModel, eager torch {torch.__version__}, autocast bf16, attention={attn}, AdamW(fused)",
          "batch": n, "steps": steps, "ms_per_step": 1e3 * dt / steps, "pairs_per_s": n * steps / dt,
          "loss": float(loss.item()), "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 1e9, 1),
          "params_m": round(sum(p.numel() for p in params) / 1e6, 1)}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--batch", type=int, nargs="+", default=[512])
  ap.add_argument("--steps", type=int, default=5)
  ap.add_argument("--warmup", type=int, default=2)
  ap.add_argument("--attn", default="sdpa")
  args = ap.parse_args()
  for n in args.batch:
    try:
      print(json.dumps(run(n, args.steps, args.warmup, args.attn)), flush=True)
    except Exception as e:   # out of memory at a batch is a result too
      print(json.dumps({"batch": n, "error": f"{type(e).__name__}: {str(e)[:200]}"}), flush=True)
    torch.cuda.empty_cache()


if __name__ == "__main__":
  sys.exit(main())
