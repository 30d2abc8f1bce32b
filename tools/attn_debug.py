import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from big_vision_amd import ops, _lib
from big_vision_amd.ops import _p, _stream
dev = torch.device("cuda:0")
BF16 = torch.bfloat16
lib = _lib.load()

def ref(qkv, n, L, H):
  q, k, v = qkv.double().view(n, L, 3, H, 64).unbind(2)
  s = torch.einsum("nqhd,nkhd->nhqk", q / 8.0, k)
  p = torch.softmax(s, -1)
  return torch.einsum("nhqk,nkhd->nqhd", p, v).reshape(n * L, H * 64), p, v

for (n, L, H) in ((1, 64, 1), (1, 196, 1)):
  g = torch.Generator(device="cpu").manual_seed(1)
  qkv = (torch.randn((n * L, 3 * H * 64), generator=g) * 1.5).to(dev).to(BF16)
  d_o = torch.randn((n * L, H * 64), generator=g).to(dev).to(BF16)
  o_ref, p, v = ref(qkv, n, L, H)
  dref = (d_o.double() * o_ref).view(n, L, H, 64).sum(-1).permute(0, 2, 1)   # [n, H, L]
  dP = torch.einsum("nqhd,nkhd->nhqk", d_o.double().view(n, L, H, 64), v)
  dref2 = (p * dP).sum(-1)
  print("ref consistency", (dref - dref2).abs().max().item())
  for impl in (3, 2):
    lib.bv_attn_impl(impl)
    o, lse = ops.attn_fwd(qkv, n, L, H)
    delta = torch.zeros((n, H, L), device=dev)
    dq = torch.empty_like(qkv)
    _lib.call("bv_attn_bwd", _p(qkv), _p(o), _p(d_o), _p(lse), _p(delta), _p(dq), _p(None), n, L, H, _stream())
    torch.cuda.synchronize()
    e = (delta.double() - dref).abs()
    print(f"L={L} impl{impl}: delta max err {e.max().item():.3e} (ref max {dref.abs().max().item():.3e})")
    print("  delta[:8]", delta[0, 0, :8].tolist())
    print("  ref  [:8]", dref[0, 0, :8].tolist())
lib.bv_attn_impl(3)
