"""A/B of the 32-key-block dK/dV kernel (attn4_bwd_dkv_kernel, BV_OPT_ATTN_CFG bits 32 / 64) against the 16-key
kernel of attention3.hip: bit-equality of dqkv and the bias-gradient rows, and time per backward. GPU only."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from big_vision_amd import ops, _lib

dev = torch.device("cuda:0")
BF16 = torch.bfloat16


def timeit(fn, iters=10, warm=3):
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters):
    fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
  lib = _lib.load()
  base = ops.ctx_get("attn_cfg")
  for name, n, L, H in (("img n=2048 L=196", 2048, 196, 12), ("img n=512 L=196", 512, 196, 12), ("LiT n=512 L=197", 512, 197, 12),
                        ("L=256 n=256", 256, 256, 12), ("L/16@336 n=256 L=441", 256, 441, 16)):
    qkv = torch.randn(n * L, 3 * H * 64, device=dev).to(BF16)
    d_o = torch.randn(n * L, H * 64, device=dev).to(BF16)
    o, lse = ops.attn_fwd(qkv, n, L, H)
    res = {}
    for cfg in (0, 32, 64):
      ops.ctx_set("attn_cfg", cfg)
      dq = torch.zeros_like(qkv)
      db = torch.zeros(3 * H * 64, device=dev)
      ops.attn_bwd(qkv, o, d_o, lse, n, L, H, dqkv=dq, dbias=db)
      first = (dq.clone(), db.clone())
      t = timeit(lambda: ops.attn_bwd(qkv, o, d_o, lse, n, L, H, dqkv=dq, dbias=db))
      res[cfg] = (first[0], first[1], t)
    ok32 = torch.equal(res[0][0], res[32][0]), (res[0][1] - res[32][1]).abs().max().item()
    ok64 = torch.equal(res[0][0], res[64][0]), (res[0][1] - res[64][1]).abs().max().item()
    print(f"{name:24s} bwd us: 16-key {res[0][2]:7.1f} | 32-key 4x2 {res[32][2]:7.1f} (dqkv equal {ok32[0]}, dbias maxdiff {ok32[1]:.2e})"
          f" | 32-key 7x1 {res[64][2]:7.1f} (dqkv equal {ok64[0]}, dbias maxdiff {ok64[1]:.2e})", flush=True)
  ops.ctx_set("attn_cfg", base)


if __name__ == "__main__":
  main()
