"""Bench lines for the BASELINE.json configs other than the headline one (bench.py = configs[2]); the workloads
themselves live in bench.py (its N = 1 line carries them as the `configs` object), this is the stand-alone CLI:

  c2  ViT-B/16 image tower alone, forward + backward, batch 256                     (configs[1])
  c4  SigLIP ViT-L/16 @336 px + text-L, global batch 8192 on 8 GPUs = 1024 pairs per GPU: one
      rank's share of the step (its 1024 pairs, micro-batches of 256) on ONE GPU    (configs[3])
  c5  LiT: frozen ViT-B/16 (cls token) + trainable text-B at 16 tokens, config batch 512 on one
      GPU (text-only backward, no image-tower gradients / optimizer state)          (configs[4])
  c5b the same with the BERT-base text tower the reference config names (text_model='proj.flaxformer.bert')
  rank512 / rank1024   the pairs one rank of the headline owns at N = 8 / 4

One JSON line per workload (same fields as bench.py where they apply; `value` is per-GPU here because
these lines are measured on one device).  GPU only; synthetic data resident in HBM; K timed steps
between synchronisations.     python tools/bench_configs.py [c2 c4 c5 c5b rank512] [--steps K]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("workloads", nargs="*", default=["c2", "c4", "c5", "c5b"])
  ap.add_argument("--steps", type=int, default=5)
  ap.add_argument("--residual-stream", default="float32", choices=("float32", "bfloat16"))
  a = ap.parse_args()
  dev = torch.device("cuda", 0)
  torch.cuda.set_device(dev)
  table = {"c2": bench.workload_c2, "c4": bench.workload_c4, "c5": bench.workload_c5, "c5b": bench.workload_c5b,
           "rank512": lambda d, k, stream: bench.workload_rank_shape(d, k, 512, stream),
           "rank1024": lambda d, k, stream: bench.workload_rank_shape(d, k, 1024, stream)}
  for w in a.workloads:
    r = table[w](dev, a.steps, stream=a.residual_stream)
    r.update(n_gpus=1, steps=a.steps, dtype="bf16", data="synthetic", higher_is_better=True)
    print(json.dumps(r), flush=True)
    torch.cuda.empty_cache()


if __name__ == "__main__":
  main()
