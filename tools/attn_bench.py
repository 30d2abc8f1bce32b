"""A/B of the attention backward paths (BV_OPT_ATTN_CFG bit 128: two launches / one launch) at the training step's shapes. GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from big_vision_amd import ops, _lib

dev = torch.device("cuda:0")
BF16 = torch.bfloat16


def timeit(fn, iters=10, warm=3):
  for _ in range(warm): fn()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
  lib = _lib.load()
  for name, n, L, H in (("img n=2048 L=196", 2048, 196, 12), ("txt n=2048 L=64", 2048, 64, 12),
                        ("img n=512 L=196", 512, 196, 12), ("LiT n=512 L=197", 512, 197, 12),
                        ("L/16@336 n=256 L=441", 256, 441, 16)):
    qkv = (torch.randn(n * L, 3 * H * 64, device=dev) * 1.0).to(BF16)
    d_o = torch.randn(n * L, H * 64, device=dev).to(BF16)
    db = torch.zeros(3 * H * 64, device=dev)
    gb = (4 * n * L * H * 64 * 2 + n * H * L * 4) / 1e9           # fwd: q,k,v read + o write + lse
    gbb = (3 * n * L * H * 64 * 2 * 2 + 2 * n * L * H * 64 * 2) / 1e9   # bwd (att3): qkv x2 passes, dO x2, dqkv write
    fl = 4 * n * H * L * L * 64 / 1e12
    row = [name]
    for impl in (3, 5):   # 3 = two-launch backward of attention3.hip, 5 = the one-launch backward of attention5.hip (the default)
      ops.ctx_set("attn_cfg", 0 if impl == 5 else 128)
      o, lse = ops.attn_fwd(qkv, n, L, H)
      dq = torch.empty_like(qkv)
      tf = timeit(lambda: ops.attn_fwd(qkv, n, L, H))
      tb = timeit(lambda: ops.attn_bwd(qkv, o, d_o, lse, n, L, H, dqkv=dq, dbias=db))
      row.append(f"impl{impl}: fwd {tf:7.1f} us ({gb / tf * 1e3:5.2f} TB/s, {fl / tf * 1e6:5.0f} TF)  bwd {tb:7.1f} us ({gbb / tb * 1e3:5.2f} TB/s)")
    ops.ctx_set("attn_cfg", 0)
    print(" | ".join(row), flush=True)


if __name__ == "__main__":
  main()
