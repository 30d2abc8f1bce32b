"""Vendor yardstick for the attention core (tools only, like tools/gemm_yardstick.py): our fused kernels
(bv_attn_fwd / bv_attn_bwd through ops.attn_*) against PyTorch's scaled_dot_product_attention on ROCm (the
flash / memory-efficient kernels the vendor ships: AOTriton / CK, whichever backend torch picks), on the step's
shapes, bf16, no mask, no dropout.  SDPA takes [n, H, L, Dh] views of the same packed qkv rows; its backward is
timed as (forward + backward) - forward.  Prints us per call and the algorithmic TFLOP/s (4 n H L^2 Dh forward,
10 n H L^2 Dh backward: 2 and 5 matmuls).  GPU only."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from big_vision_amd import ops

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


def backends():
  try:
    from torch.nn.attention import SDPBackend, sdpa_kernel
    return [("flash", lambda: sdpa_kernel(SDPBackend.FLASH_ATTENTION)),
            ("mem_efficient", lambda: sdpa_kernel(SDPBackend.EFFICIENT_ATTENTION)),
            ("math", lambda: sdpa_kernel(SDPBackend.MATH))]
  except Exception:   # older layout of the module
    return []


def main():
  print(f"torch {torch.__version__}; device {torch.cuda.get_device_name(0)}")
  for name, n, L, H in (("img  n=2048 L=196", 2048, 196, 12), ("text n=2048 L=64", 2048, 64, 12), ("img  n=512  L=196", 512, 196, 12),
                        ("text n=512  L=64", 512, 64, 12), ("L/16@336 n=256 L=441", 256, 441, 16)):
    Dh = 64
    qkv = torch.randn(n * L, 3 * H * Dh, device=dev).to(BF16)
    d_o = torch.randn(n * L, H * Dh, device=dev).to(BF16)
    f_fwd, f_bwd = 4.0 * n * H * L * L * Dh, 10.0 * n * H * L * L * Dh
    # ---- ours
    o, lse = ops.attn_fwd(qkv, n, L, H)
    dq = torch.zeros_like(qkv)
    db = torch.zeros(3 * H * Dh, device=dev)
    t_f = timeit(lambda: ops.attn_fwd(qkv, n, L, H))
    t_b = timeit(lambda: ops.attn_bwd(qkv, o, d_o, lse, n, L, H, dqkv=dq, dbias=db))
    print(f"{name:22s} ours           fwd {t_f:8.1f} us {f_fwd / t_f * 1e-6:6.0f} TF/s | bwd {t_b:8.1f} us {f_bwd / t_b * 1e-6:6.0f} TF/s", flush=True)
    # ---- SDPA on [n, H, L, Dh] views (strided: what a drop-in caller would pass) and on contiguous copies
    v5 = qkv.view(n, L, 3, H, Dh)
    for layout in ("strided", "contiguous"):
      q, k, v = (v5[:, :, j].permute(0, 2, 1, 3) for j in range(3))
      g = d_o.view(n, L, H, Dh).permute(0, 2, 1, 3)
      if layout == "contiguous":
        q, k, v, g = (t.contiguous() for t in (q, k, v, g))
      q, k, v = (t.detach().requires_grad_(True) for t in (q, k, v))
      for bname, ctx in backends():
        try:
          with ctx():
            ref = F.scaled_dot_product_attention(q, k, v)
            err = (ref.permute(0, 2, 1, 3).reshape(n * L, H * Dh).float() - o.float()).abs().max().item()
            tf = timeit(lambda: F.scaled_dot_product_attention(q, k, v))

            def fb():
              out = F.scaled_dot_product_attention(q, k, v)
              out.backward(g)
              q.grad = k.grad = v.grad = None
            tfb = timeit(fb)
          tb = tfb - tf
          print(f"{'':22s} sdpa {bname:13s} {layout:10s} fwd {tf:8.1f} us {f_fwd / tf * 1e-6:6.0f} TF/s | bwd {tb:8.1f} us "
                f"{f_bwd / tb * 1e-6:6.0f} TF/s | max |o - ours| {err:.3e}", flush=True)
        except Exception as e:
          print(f"{'':22s} sdpa {bname:13s} {layout:10s} unavailable: {type(e).__name__}: {str(e)[:90]}", flush=True)
      del q, k, v, g
    torch.cuda.empty_cache()


if __name__ == "__main__":
  main()
