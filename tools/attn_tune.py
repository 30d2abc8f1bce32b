"""A/B of the BV_OPT_ATTN_CFG switches at the image tower's shape. GPU only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from big_vision_amd import ops, _lib
from attn_bench import timeit, dev, BF16


def main():
  lib = _lib.load()
  cfgs = [int(a) for a in sys.argv[1:]] or [16, 0, 8]   # +16: two-sweep dQ kernel, 8: forward with 8 waves x 2 workgroups
  for name, n, L, H in (("img n=2048 L=196", 2048, 196, 12), ("img n=512 L=196", 512, 196, 12)):
    qkv = torch.randn(n * L, 3 * H * 64, device=dev).to(BF16)
    d_o = torch.randn(n * L, H * 64, device=dev).to(BF16)
    db = torch.zeros(3 * H * 64, device=dev)
    ops.ctx_set("attn_cfg", 16)
    o0, lse0 = ops.attn_fwd(qkv, n, L, H)
    dq0 = torch.empty_like(qkv)
    ops.attn_bwd(qkv, o0, d_o, lse0, n, L, H, dqkv=dq0, dbias=db)
    for cfg in cfgs:
      ops.ctx_set("attn_cfg", cfg)
      o, lse = ops.attn_fwd(qkv, n, L, H)
      dq = torch.empty_like(qkv)
      ops.attn_bwd(qkv, o, d_o, lse, n, L, H, dqkv=dq, dbias=db)
      same = bool((o == o0).all()) and bool((dq == dq0).all())
      rel = float((dq.float() - dq0.float()).norm() / dq0.float().norm())
      tf = timeit(lambda: ops.attn_fwd(qkv, n, L, H))
      tb = timeit(lambda: ops.attn_bwd(qkv, o, d_o, lse, n, L, H, dqkv=dq, dbias=db))
      tn = timeit(lambda: ops.attn_bwd(qkv, o, d_o, lse, n, L, H, dqkv=dq))
      print(f"{name} cfg {cfg}: fwd {tf:7.1f} us  bwd {tb:7.1f} us (without the bias sums {tn:7.1f})  bit-equal to the two-sweep default: {same} (dqkv rel-L2 {rel:.2e})", flush=True)
    ops.ctx_set("attn_cfg", 0)


if __name__ == "__main__":
  main()
