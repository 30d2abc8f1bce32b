"""Vendor yardstick for the FUSED-epilogue GEMMs of the MLP / attention-output path (tools only; the plain NT and
dW shapes are in tools/gemm_yardstick.py).  For each fused launch of the step: our kernel, and the cheapest way
the PyTorch-ROCm stack gets the same tensors (hipBLASLt's own GELU / bias epilogue where it has one, otherwise
GEMM + the elementwise kernels), one process, back to back.  TFLOP/s count the matmul only (2 M N K), so the
columns compare time.

  fc1 forward   [M,768] x [768,3072] + bias -> h (bf16) and g = gelu_tanh(h) (bf16)          ours: BV_EPI_GELU
                vendor a: torch._addmm_activation(use_gelu=True)  (hipBLASLt GELU epilogue; returns g ONLY - the
                          backward then has to recompute h or keep fp32 pre-activations)
                vendor b: addmm -> h, then F.gelu(h, approximate="tanh") -> g                 (two launches)
  fc2 forward   [M,3072] x [3072,768] + bias + fp32 residual -> fp32 stream                   ours: BV_EPI_RESIDUAL
                vendor:   addmm -> bf16, then residual + out.float()                          (two launches)
  fc2 dX        [M,768] x [768,3072] * gelu'(h) (kept, bf16) -> dh (bf16), bias-grad colsum   ours: BV_EPI_MUL + colsum
                vendor:   matmul -> bf16, then out * d, then sum(0)                           (three launches)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32


def timeit(fn, iters):
  fn()
  torch.cuda.synchronize()
  best = 1e30
  for _ in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
      fn()
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / iters)
  return best


def main():
  from big_vision_amd import ops
  dev = torch.device("cuda", 0)
  torch.backends.cuda.preferred_blas_library("hipblaslt")
  g = torch.Generator(device=dev).manual_seed(0)
  rnd = lambda shape, s=1.0: ((torch.rand(shape, device=dev, generator=g) * 2 - 1) * s)
  print(f"# torch {torch.__version__}, hip {torch.version.hip}, device {torch.cuda.get_device_name(0)}")
  print(f"# {'launch':12s} {'M':>7} | {'ours':>6} | vendor variants (TFLOP/s of the 2MNK matmul; higher = less time)")
  D, Mh = 768, 3072
  for M in (131072, 401408):
    it = 4 if M > 200000 else 8
    x = rnd((M, D)).to(BF16)
    w1t = rnd((Mh, D), 0.05).to(BF16)          # [out][in] image (k-major B operand); the vendor gets its transpose view
    b1 = rnd((Mh,))
    b1h = b1.to(BF16)
    tf = lambda ms, fl: fl / ms / 1e9
    fl1 = 2.0 * M * D * Mh
    # ---- fc1 forward
    h = torch.empty((M, Mh), device=dev, dtype=BF16)
    gg = torch.empty_like(h)
    ours = tf(timeit(lambda: ops.gemm(x, w1t, a_kmajor=True, b_kmajor=True, bias=b1, out=h, epilogue=ops.EPI_GELU, out2=gg), it), fl1)
    w1 = w1t.t()
    try:
      va = tf(timeit(lambda: torch._addmm_activation(b1h, x, w1, use_gelu=True), it), fl1)
    except Exception as e:   # noqa: BLE001
      va = float("nan")
      print("#   _addmm_activation unavailable:", type(e).__name__, str(e)[:80])

    def vb():
      hh = torch.addmm(b1h, x, w1)
      return hh, F.gelu(hh, approximate="tanh")
    vbt = tf(timeit(vb, it), fl1)
    print(f"  {'fc1 + GELU':12s} {M:>7} | {ours:6.0f} | hipBLASLt GELU epilogue (g only) {va:5.0f} | addmm + F.gelu (h and g) {vbt:5.0f}", flush=True)
    # ---- fc2 forward with the fp32 residual stream
    w2t = rnd((D, Mh), 0.05).to(BF16)
    b2 = rnd((D,))
    res = rnd((M, D))
    out = torch.empty((M, D), device=dev, dtype=F32)
    fl2 = 2.0 * M * Mh * D
    ours = tf(timeit(lambda: ops.gemm(gg, w2t, a_kmajor=True, b_kmajor=True, bias=b2, out=out, epilogue=ops.EPI_RESIDUAL, aux=res), it), fl2)
    w2 = w2t.t()
    b2h = b2.to(BF16)
    v = tf(timeit(lambda: res + torch.addmm(b2h, gg, w2).float(), it), fl2)
    print(f"  {'fc2 + resid':12s} {M:>7} | {ours:6.0f} | addmm (bf16) + fp32 residual add {v:5.0f}", flush=True)
    # ---- fc2 dX with the kept gelu'(h)
    dy = rnd((M, D), 0.05).to(BF16)
    dker = w2t.t().contiguous()                 # dh = dy [M,768] x W2^T: B operand [3072][768] k-major
    d = rnd((M, Mh)).to(BF16)
    dh = torch.empty((M, Mh), device=dev, dtype=BF16)
    cs = torch.zeros((Mh,), device=dev)
    ours = tf(timeit(lambda: ops.gemm(dy, dker, a_kmajor=True, b_kmajor=True, out=dh, epilogue=ops.EPI_MUL, aux=d, colsum=cs), it), fl1)
    dk = dker.t()

    def vd():
      t = torch.matmul(dy, dk) * d
      return t, t.sum(0, dtype=F32)
    v = tf(timeit(vd, it), fl1)
    print(f"  {'fc2 dX * gelu_prime':12s} {M:>7} | {ours:6.0f} | matmul + mul + column sum {v:5.0f}", flush=True)
    del x, h, gg, res, out, dy, d, dh
    torch.cuda.empty_cache()


if __name__ == "__main__":
  main()
