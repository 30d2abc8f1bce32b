"""Vendor yardstick for the HBM-bound kernels of the step (tools only): our LayerNorm forward / backward and fused
Adam against what PyTorch-ROCm ships for the same work on the same MI355X.

LayerNorm (fp32 residual stream, 401 408 x 768 and 131 072 x 768):
  ours      bv_layernorm_fwd: fp32 x -> bf16 y + mean / rstd;  bv_layernorm_bwd_y as the step calls it (bf16 dy, fp32 x and
            residual gradient in; fp32 dx + its bf16 copy + the re-emitted bf16 y out; dscale / dbias accumulated)
  torch     F.layer_norm on the fp32 x followed by the bf16 cast the GEMM needs; backward = autograd of that +
            the residual-gradient add and the bf16 copy (the ops a PyTorch port of the block would run)
Adam (203 M parameters, fp32 master / moments, bf16 shadow, global-norm clip, decoupled weight decay):
  ours      sqnorm + bv_adam_step (one pass over the flat buffers)
  torch     clip_grad_norm_ + torch.optim.AdamW(fused=True) + the bf16 cast of the parameters
Prints us per call and the algorithmic HBM rate of OUR byte count for both (so the columns compare time).  GPU only."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from big_vision_amd import ops

dev = torch.device("cuda:0")
BF16, F32 = torch.bfloat16, torch.float32


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


def layernorm(rows, D=768):
  x = torch.randn(rows, D, device=dev) * 2 + 0.5
  scale, bias = 1 + 0.1 * torch.randn(D, device=dev), 0.1 * torch.randn(D, device=dev)
  dy = torch.randn(rows, D, device=dev).to(BF16)
  dres = torch.randn(rows, D, device=dev)
  y_bf, _, mean, rstd = ops.layernorm_fwd(x, scale, bias, rows=rows, D=D)
  t_f = timeit(lambda: ops.layernorm_fwd(x, scale, bias, rows=rows, D=D))
  dx, dx_bf, y_re = torch.empty_like(x), torch.empty_like(y_bf), torch.empty_like(y_bf)
  ds, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
  t_b = timeit(lambda: ops.layernorm_bwd(dy, x, scale, mean, rstd, rows=rows, D=D, dres=dres, dx=dx, dx_bf16=dx_bf,
                                         dscale=ds, dbias=db, bias=bias, y_out=y_re))
  b_f, b_b = rows * (D * 6.0 + 8), rows * (D * 18.0 + 8)
  print(f"LayerNorm {rows:7d} x {D}  ours   fwd {t_f:7.1f} us {b_f / t_f * 1e-6:5.2f} TB/s | bwd {t_b:7.1f} us {b_b / t_b * 1e-6:5.2f} TB/s", flush=True)
  # torch: what a port of the block would run
  xs = x.clone().requires_grad_(True)
  sc, bi = scale.clone().requires_grad_(True), bias.clone().requires_grad_(True)
  t_tf = timeit(lambda: F.layer_norm(xs, (D,), sc, bi, eps=1e-6).to(BF16))

  def fb():
    y = F.layer_norm(xs, (D,), sc, bi, eps=1e-6).to(BF16)
    y.backward(dy)
    g = xs.grad + dres          # residual-gradient add
    gb = g.to(BF16)             # the copy the next GEMM reads
    xs.grad = sc.grad = bi.grad = None
    return g, gb
  t_tfb = timeit(fb)
  t_tb = t_tfb - t_tf
  print(f"{'':22s}  torch  fwd {t_tf:7.1f} us {b_f / t_tf * 1e-6:5.2f} TB/s | bwd {t_tb:7.1f} us {b_b / t_tb * 1e-6:5.2f} TB/s  "
        f"(ours {t_tf / t_f:.2f}x / {t_tb / t_b:.2f}x faster)", flush=True)


def adam(n=203_000_000 // 1024 * 1024):
  from big_vision_amd import _lib
  p = torch.randn(n, device=dev) * 0.02
  g = torch.randn(n, device=dev) * 1e-3
  mu, nu = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
  shadow = torch.empty(n, device=dev, dtype=BF16)
  segs = torch.tensor([[1e-3, 1e-4, 0.0, 0.0]], device=dev, dtype=F32)
  chunk_seg = torch.zeros(n // 1024, device=dev, dtype=torch.int32)
  gsq = torch.zeros(1, device=dev, dtype=torch.float64)
  stats = torch.zeros(2, device=dev, dtype=torch.float64)

  def ours():
    gsq.zero_()
    ops.sqnorm_(g, gsq)
    stats.zero_()
    ops.adam_step_(p, g, mu, nu, shadow, segs, chunk_seg, n, [1.0], gsq, 1.0, 0.9, 0.999, 1e-8, 0.1, 0.001, stats)
  t_o = timeit(ours, iters=5, warm=2)
  nbytes = n * 34.0       # sqnorm 4 + step: read p g m v 16, write p m v 12 + shadow 2
  print(f"Adam {n / 1e6:.0f} M params       ours   {t_o:8.1f} us {nbytes / t_o * 1e-6:5.2f} TB/s", flush=True)
  del mu, nu
  pt = torch.nn.Parameter(p.clone())
  pt.grad = g.clone()
  opt = torch.optim.AdamW([pt], lr=1e-3, weight_decay=1e-4, fused=True)

  def theirs():
    torch.nn.utils.clip_grad_norm_([pt], 1.0)
    opt.step()
    return pt.detach().to(BF16)
  t_t = timeit(theirs, iters=5, warm=2)
  print(f"{'':22s}  torch  {t_t:8.1f} us {nbytes / t_t * 1e-6:5.2f} TB/s  (clip_grad_norm_ + AdamW(fused=True) + bf16 cast; ours {t_t / t_o:.2f}x faster)",
        flush=True)


if __name__ == "__main__":
  print(f"torch {torch.__version__}; device {torch.cuda.get_device_name(0)}")
  for rows in (401408, 131072):
    layernorm(rows)
  adam()
