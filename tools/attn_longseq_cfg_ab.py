"""A/B of the BV_OPT_ATTN_CFG switches (include/bvhip.h) at the LONG sequences - L = 441 (L/16 at 336 px, BASELINE configs[3]),
256, 576 - where the two-launch backward of attention3.hip runs.  GPU only.   python tools/attn_longseq_cfg_ab.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from big_vision_amd import ops
from attn_bench import timeit, dev, BF16
for name, n, L, H in (("L/16@336 n=256 L=441", 256, 441, 16), ("n=256 L=256 H=12", 256, 256, 12), ("L=576 n=128 H=12", 128, 576, 12)):
  qkv = torch.randn(n * L, 3 * H * 64, device=dev).to(BF16)
  d_o = torch.randn(n * L, H * 64, device=dev).to(BF16)
  db = torch.zeros(3 * H * 64, device=dev)
  ops.ctx_set("attn_cfg", 0)
  o0, lse0 = ops.attn_fwd(qkv, n, L, H)
  dq0 = torch.empty_like(qkv); ops.attn_bwd(qkv, o0, d_o, lse0, n, L, H, dqkv=dq0, dbias=db)
  for cfg in (0, 16, 32, 64, 48, 80, 8):
    ops.ctx_set("attn_cfg", cfg)
    o, lse = ops.attn_fwd(qkv, n, L, H)
    dq = torch.empty_like(qkv)
    ops.attn_bwd(qkv, o, d_o, lse, n, L, H, dqkv=dq, dbias=db)
    rel = float((dq.float() - dq0.float()).norm() / dq0.float().norm())
    tf = timeit(lambda: ops.attn_fwd(qkv, n, L, H))
    tb = timeit(lambda: ops.attn_bwd(qkv, o, d_o, lse, n, L, H, dqkv=dq, dbias=db))
    print(f"{name} cfg {cfg:3d}: fwd {tf:7.1f} us  bwd {tb:7.1f} us  dqkv rel diff vs cfg 0 {rel:.2e}", flush=True)
  ops.ctx_set("attn_cfg", 0)
