"""Per-kernel micro-benchmarks at the ViT-B/16 (n=512) shapes. GPU only."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from big_vision_amd import ops

dev = torch.device("cuda:0")
BF16, F32 = torch.bfloat16, torch.float32


def timeit(fn, iters=10, warm=3):
  for _ in range(warm): fn()
  torch.cuda.synchronize()
  e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters


def main():
  n = int(os.environ.get("N", 512)); L = 196; D = 768; M = 3072; H = 12
  T = n * L
  res = {}
  x = torch.randn(T, D, device=dev).to(BF16)
  h = torch.randn(T, M, device=dev).to(BF16)
  for name, K, N in (("qkv", D, 3 * D), ("out", D, D), ("fc1", D, M), ("fc2", M, D)):
    a = x if K == D else h
    w = (torch.randn(K, N, device=dev) * 0.02).to(BF16)
    out = torch.empty(T, N, device=dev, dtype=BF16)
    ms = timeit(lambda: ops.gemm(a, w, a_kmajor=True, b_kmajor=False, out=out))
    res[f"fwd_{name}"] = (ms, 2 * T * N * K / ms / 1e9)
    dy = x if N == D else (h if N == M else torch.randn(T, N, device=dev).to(BF16))
    dx = torch.empty(T, K, device=dev, dtype=BF16)
    ms = timeit(lambda: ops.gemm(dy, w, a_kmajor=True, b_kmajor=True, out=dx))
    res[f"dx_{name}"] = (ms, 2 * T * N * K / ms / 1e9)
    dw = torch.zeros(K, N, device=dev)
    ms = timeit(lambda: ops.gemm(a, dy, a_kmajor=False, b_kmajor=False, out=dw, epilogue=ops.EPI_ATOMIC))
    res[f"dw_{name}"] = (ms, 2 * T * N * K / ms / 1e9)
  # 256x256 direct-to-LDS path (NT with pre-transposed weights) vs the general kernel
  for name, K, N in (("qkv", D, 3 * D), ("out", D, D), ("fc1", D, M), ("fc2", M, D)):
    a = x if K == D else h
    wt = (torch.randn(N, K, device=dev) * 0.02).to(BF16)     # W^T [out][in]
    out = torch.empty(T, N, device=dev, dtype=BF16)
    for fast in (1, 0):
      ops.ctx_set("fast_path", fast)
      ms = timeit(lambda: ops.gemm(a, wt, a_kmajor=True, b_kmajor=True, out=out))
      res[f"nt{'256' if fast else '128'}_{name}"] = (ms, 2 * T * N * K / ms / 1e9)
    dy = x if N == D else (h if N == M else torch.randn(T, N, device=dev).to(BF16))
    dw = torch.zeros(K, N, device=dev)
    for fast in (1, 0):
      ops.ctx_set("fast_path", fast)
      ms = timeit(lambda: ops.gemm(a, dy, a_kmajor=False, b_kmajor=False, out=dw, epilogue=ops.EPI_ATOMIC))
      res[f"tn{'256' if fast else '128'}_{name}"] = (ms, 2 * T * N * K / ms / 1e9)
    ops.ctx_set("fast_path", 1)
    ops.ctx().use_workspace = False     # fp32-atomic split-K for comparison
    ms = timeit(lambda: ops.gemm(a, dy, a_kmajor=False, b_kmajor=False, out=dw, epilogue=ops.EPI_ATOMIC))
    res[f"tn256atomic_{name}"] = (ms, 2 * T * N * K / ms / 1e9)
    ops.ctx().use_workspace = True
  ops.ctx_set("fast_path", 1)
  qkv = torch.randn(T, 3 * D, device=dev).to(BF16)
  ms = timeit(lambda: ops.attn_fwd(qkv, n, L, H))
  fl = 4 * n * H * L * L * 64
  res["attn_fwd"] = (ms, fl / ms / 1e9)
  o, lse = ops.attn_fwd(qkv, n, L, H)
  do = torch.randn(T, D, device=dev).to(BF16)
  ms = timeit(lambda: ops.attn_bwd(qkv, o, do, lse, n, L, H))
  res["attn_bwd"] = (ms, 2.5 * fl / ms / 1e9)
  xf = torch.randn(T, D, device=dev)
  sc = torch.ones(D, device=dev); bi = torch.zeros(D, device=dev)
  ms = timeit(lambda: ops.layernorm_fwd(xf, sc, bi, rows=T, D=D))
  res["ln_fwd"] = (ms, T * D * 6 / ms / 1e6)  # GB/s: 4B read + 2B write
  _, _, mean, rstd = ops.layernorm_fwd(xf, sc, bi, rows=T, D=D)
  dyb = x; dres = xf.clone(); dx = torch.empty_like(xf); dxb = torch.empty_like(x)
  ds = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
  ms = timeit(lambda: ops.layernorm_bwd(dyb, xf, sc, mean, rstd, rows=T, D=D, dres=dres, dx=dx, dx_bf16=dxb, dscale=ds, dbias=db))
  res["ln_bwd"] = (ms, T * D * (2 + 4 + 4 + 4 + 2) / ms / 1e6)
  acc = torch.zeros(M, device=dev)
  ms = timeit(lambda: ops.colsum(h, acc))
  res["colsum_bf16_M"] = (ms, T * M * 2 / ms / 1e6)
  for k, (ms, rate) in res.items():
    unit = "GB/s" if k.startswith(("ln", "colsum")) else "TFLOP/s"
    print(f"{k:16s} {ms:8.3f} ms  {rate:9.1f} {unit}")
  print(json.dumps({k: v for k, v in res.items()}))


if __name__ == "__main__":
  main()
