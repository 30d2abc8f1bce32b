"""Parity + race screen for the 256x256x64 direct-to-LDS GEMM path (gemm256.hip),
reached through bv_gemm_bf16 when M,N % 256 == 0, K % 64 == 0 and both operands
share a layout.  Reference = fp32 matmul on the same bf16-rounded inputs (the
only differences are accumulation order and output rounding).  Each case is
launched several times back-to-back and must be bit-identical run to run (a
staging race shows up as run-to-run differences) and match the general
128x128 kernel to accumulation-order noise."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


def rnd(shape, dev, seed, scale=1.0, dtype=F32):
  g = torch.Generator(device="cpu").manual_seed(seed)
  return (torch.randn(shape, generator=g, dtype=torch.float32) * scale).to(dev).to(dtype)


def close(a, b, rtol, atol, name):
  a = a.double(); b = b.double()
  err = (a - b).abs()
  bad = err > atol + rtol * b.abs()
  assert not bad.any(), f"{name}: {int(bad.sum())}/{bad.numel()} bad, max err {err.max().item():.3e}"


@pytest.fixture()
def fast(dev):
  from big_vision_amd import ops
  ops.ctx_set("fast_path", 1)
  yield
  ops.ctx_set("fast_path", 1)


def _general(fn):
  from big_vision_amd import ops
  with ops.option("fast_path", 0):
    return fn()


NT_SHAPES = [(256, 256, 64), (256, 256, 128), (512, 768, 768), (1024, 2304, 768), (768, 768, 3072),
             (2048, 3072, 768), (256, 512, 192)]


@pytest.mark.parametrize("M,N,K", NT_SHAPES)
def test_nt_matches_reference(dev, fast, M, N, K):
  """dX layout: A [M][K], B [N][K] (asymmetric operands catch transposes)."""
  from big_vision_amd import ops
  a = rnd((M, K), dev, 1, dtype=BF16)
  b = rnd((N, K), dev, 2, 0.05, dtype=BF16)
  bias = rnd((N,), dev, 3)
  ref = a.float() @ b.float().T + bias
  outs = [ops.gemm(a, b, a_kmajor=True, b_kmajor=True, bias=bias, out_dtype=F32) for _ in range(4)]
  close(outs[0], ref, 1e-4, 2e-3, "nt f32")
  for o in outs[1:]:
    assert torch.equal(o, outs[0]), "run-to-run difference (staging race?)"
  gen = _general(lambda: ops.gemm(a, b, a_kmajor=True, b_kmajor=True, bias=bias, out_dtype=F32))
  close(outs[0], gen, 1e-5, 1e-3, "nt vs general kernel")
  o16 = ops.gemm(a, b, a_kmajor=True, b_kmajor=True, bias=bias, out_dtype=BF16)
  close(o16, ref, 1e-2, 1e-2, "nt bf16")


def test_nt_epilogues(dev, fast):
  from big_vision_amd import ops
  M, N, K, L = 784 * 0 + 512, 512, 256, 128
  x = rnd((M, K), dev, 8, dtype=BF16)
  w = rnd((N, K), dev, 9, 0.1, dtype=BF16)
  b = rnd((N,), dev, 10)
  pre = x.float() @ w.float().T + b
  res = rnd((M, N), dev, 11)
  kw = dict(a_kmajor=True, b_kmajor=True)
  y = ops.gemm(x, w, bias=b, out_dtype=F32, epilogue=ops.EPI_RESIDUAL, aux=res, **kw)
  close(y, pre + res, 1e-4, 2e-3, "residual")
  # bf16 residual stream: aux and C are bf16 (same epilogue id, out_f32 = 0)
  resb = res.to(BF16)
  yb = ops.gemm(x, w, bias=b, out_dtype=BF16, epilogue=ops.EPI_RESIDUAL, aux=resb, **kw)
  close(yb, pre + resb.float(), 1e-2, 1e-2, "residual (bf16 stream)")
  assert torch.equal(yb, ops.gemm(x, w, bias=b, out_dtype=BF16, epilogue=ops.EPI_RESIDUAL, aux=resb, **kw))
  pos = rnd((L, N), dev, 12)
  y = ops.gemm(x, w, bias=b, out_dtype=F32, epilogue=ops.EPI_POS, aux=pos, aux_rows=L, **kw)
  close(y, pre + pos.repeat(M // L, 1), 1e-4, 2e-3, "pos")
  g = torch.empty((M, N), device=dev, dtype=BF16)
  h = ops.gemm(x, w, bias=b, out_dtype=BF16, epilogue=ops.EPI_GELU, out2=g, **kw)
  close(h, pre, 1e-2, 1e-2, "gelu pre")
  close(g, torch.nn.functional.gelu(pre, approximate="tanh"), 1e-2, 1e-2, "gelu out")
  hh = rnd((M, N), dev, 15, dtype=BF16)
  hf = hh.float().requires_grad_(True)
  torch.nn.functional.gelu(hf, approximate="tanh").sum().backward()
  ref = (x.float() @ w.float().T) * hf.grad
  out = ops.gemm(x, w, out_dtype=BF16, epilogue=ops.EPI_GELU_BWD, aux=hh, **kw)
  close(out, ref, 1e-2, 2e-2, "gelu bwd")
  # GELU_BWD_EMIT: same dX, and C2 = gelu(aux) with the bits the forward epilogue produces
  g2 = torch.empty((M, N), device=dev, dtype=BF16)
  out2 = ops.gemm(x, w, out_dtype=BF16, epilogue=ops.EPI_GELU_BWD_EMIT, aux=hh, out2=g2, **kw)
  close(out2, ref, 1e-2, 2e-2, "gelu bwd (emit)")
  close(g2, torch.nn.functional.gelu(hh.float(), approximate="tanh"), 1e-2, 1e-2, "emitted gelu")
  g3 = torch.empty((M, N), device=dev, dtype=BF16)
  ops.gemm(x, w, out_dtype=BF16, epilogue=ops.EPI_GELU_BWD_EMIT, aux=h, out2=g3, **kw)
  assert torch.equal(g3, g), "gelu(h) re-emitted by the backward differs from the forward's"
  # fused column sums (Dense_0 bias gradient) of both GELU_BWD epilogues, accumulated in place
  for epi, extra in ((ops.EPI_GELU_BWD, {}), (ops.EPI_GELU_BWD_EMIT, dict(out2=g2))):
    cs = torch.ones((N,), device=dev, dtype=F32)
    o3 = ops.gemm(x, w, out_dtype=BF16, epilogue=epi, aux=hh, colsum=cs, **extra, **kw)
    assert torch.equal(o3, out), "colsum changed the GEMM result"
    close(cs, 1.0 + ref.sum(0), 1e-3, 1e-3 * ref.abs().sum(0).max().item(), "fused colsum")
  # GELU_GD (forward, full contexts): C = gelu(pre), C2 = gelu'(pre) from the fp32 pre-activation; MUL
  # (backward): C = (x w^T) o aux with the fused column sums.  The pair must reproduce GELU_BWD's dX when
  # fed the derivative GELU_GD emitted.
  g4 = torch.empty((M, N), device=dev, dtype=BF16)
  d4 = torch.empty((M, N), device=dev, dtype=BF16)
  ret = ops.gemm(x, w, bias=b, out=g4, epilogue=ops.EPI_GELU_GD, out2=d4, **kw)
  assert ret is g4
  pf = pre.detach().clone().requires_grad_(True)
  torch.nn.functional.gelu(pf, approximate="tanh").sum().backward()
  close(g4, torch.nn.functional.gelu(pre, approximate="tanh"), 1e-2, 1e-2, "gelu_gd value")
  close(d4, pf.grad, 1e-2, 1e-2, "gelu_gd derivative")
  # one definition of the activation for every context kind: GELU_GD's g has the bits of GELU's g, and
  # MUL fed GELU_GD's derivative reproduces GELU_BWD / GELU_BWD_EMIT fed the stored h, bit for bit
  assert torch.equal(g4, g), "GELU_GD and GELU disagree on gelu(h)"
  o_mul = ops.gemm(x, w, out_dtype=BF16, epilogue=ops.EPI_MUL, aux=d4, **kw)
  o_bwd = ops.gemm(x, w, out_dtype=BF16, epilogue=ops.EPI_GELU_BWD, aux=h, **kw)
  g5 = torch.empty((M, N), device=dev, dtype=BF16)
  o_emit = ops.gemm(x, w, out_dtype=BF16, epilogue=ops.EPI_GELU_BWD_EMIT, aux=h, out2=g5, **kw)
  assert torch.equal(o_mul, o_bwd) and torch.equal(o_mul, o_emit) and torch.equal(g5, g)
  dd = hf.grad.to(BF16)
  cs = torch.ones((N,), device=dev, dtype=F32)
  o5 = ops.gemm(x, w, out_dtype=BF16, epilogue=ops.EPI_MUL, aux=dd, colsum=cs, **kw)
  ref5 = (x.float() @ w.float().T) * dd.float()
  close(o5, ref5, 1e-2, 2e-2, "mul")
  close(cs, 1.0 + ref5.sum(0), 1e-3, 1e-3 * ref5.abs().sum(0).max().item(), "mul: fused colsum")
  assert torch.equal(o5, ops.gemm(x, w, out_dtype=BF16, epilogue=ops.EPI_MUL, aux=dd, **kw))
  # the 256x256 kernel and the general kernel evaluate the same operation sequence: same bits
  gg, dg = torch.empty_like(g4), torch.empty_like(d4)
  _general(lambda: ops.gemm(x, w, bias=b, out=gg, epilogue=ops.EPI_GELU_GD, out2=dg, **kw))
  close(gg, g4, 1e-2, 1e-2, "gelu_gd vs general kernel"); close(dg, d4, 1e-2, 1e-2, "gelu_gd' vs general kernel")
  y = ops.gemm(x, w, out_dtype=BF16, alpha=0.5, **kw)
  close(y, 0.5 * (x.float() @ w.float().T), 1e-2, 1e-2, "alpha")


def test_nt_strided_views(dev, fast):
  """Operands / outputs that are column slices of wider buffers (lda/ldb/ldc > width)."""
  from big_vision_amd import ops
  M, N, K = 512, 256, 128
  abuf = rnd((M, 3 * K), dev, 1, dtype=BF16)
  bbuf = rnd((N, 2 * K), dev, 2, dtype=BF16)
  cbuf = torch.zeros((M, 2 * N), device=dev, dtype=F32)
  a, b = abuf[:, K:2 * K], bbuf[:, K:]
  ops.gemm(a, b, a_kmajor=True, b_kmajor=True, out=cbuf[:, N:])
  close(cbuf[:, N:], a.float() @ b.float().T, 1e-4, 2e-3, "strided")
  assert torch.count_nonzero(cbuf[:, :N]) == 0


TN_SHAPES = [(256, 256, 64, 1), (256, 256, 1024, 0), (768, 768, 4096, 0), (768, 2304, 6272, 0),
             (3072, 768, 2048, 3), (256, 512, 320, 5)]


@pytest.mark.parametrize("Din,Dout,T,split", TN_SHAPES)
def test_tn_dw_matches_reference(dev, fast, Din, Dout, T, split):
  """dW += X^T dY: both operands k-minor, split-K fp32 atomics, accumulating."""
  from big_vision_amd import ops
  x = rnd((T, Din), dev, 5, dtype=BF16)
  dy = rnd((T, Dout), dev, 6, 0.25, dtype=BF16)
  base = rnd((Din, Dout), dev, 7)
  ref = base.double() + x.double().T @ dy.double()
  tol = 2e-5 * (T ** 0.5) * 4
  for _ in range(3):
    out = base.clone()
    ops.gemm(x, dy, a_kmajor=False, b_kmajor=False, out=out, epilogue=ops.EPI_ATOMIC, split_k=split)
    close(out, ref, 1e-4, tol, "tn dw")
  out1 = torch.zeros((Din, Dout), device=dev)
  ops.gemm(x, dy, a_kmajor=False, b_kmajor=False, out=out1, epilogue=ops.EPI_ATOMIC, split_k=1)
  out2 = torch.zeros((Din, Dout), device=dev)
  ops.gemm(x, dy, a_kmajor=False, b_kmajor=False, out=out2, epilogue=ops.EPI_ATOMIC, split_k=1)
  assert torch.equal(out1, out2), "split_k=1 must be deterministic (staging race?)"


def test_big_shape_spot_check(dev, fast):
  """ViT-B/16 fc1 at n=64 (T=12544): row subset against fp64."""
  from big_vision_amd import ops
  T, D, Mlp = 12544, 768, 3072
  x = rnd((T, D), dev, 21, dtype=BF16)
  wt = rnd((Mlp, D), dev, 22, 0.03, dtype=BF16)
  y = ops.gemm(x, wt, a_kmajor=True, b_kmajor=True, out_dtype=F32)
  rows = torch.arange(0, T, 97, device=dev)
  ref = x[rows].double() @ wt.double().T
  close(y[rows], ref, 1e-4, 2e-3, "fc1 rows")
  y2 = ops.gemm(x, wt, a_kmajor=True, b_kmajor=True, out_dtype=F32)
  assert torch.equal(y, y2)


ROLL_SHAPES = [(256, 256, 128), (512, 256, 192), (1024, 768, 320), (66816, 768, 128), (2304, 768, 768),
               (33024, 768, 768)]


@pytest.mark.parametrize("M,N,K", ROLL_SHAPES)
def test_rolling_epilogue_kernel_matches_the_full_epilogue_kernel(dev, fast, M, N, K):
  """gemm256r_kernel (epilogue folded into the K loop, residual loaded into the accumulators) vs
  gemm256_kernel through the same dispatcher (BV_OPT_GEMM_ROLL mask): bit-identical for the bf16
  epilogues (same accumulation order and epilogue arithmetic), fp32 rounding-order noise for
  +residual; run-to-run bit-equality as the race screen.  66816 x 768 x 128 = 783 tiles of two
  K-tiles: every workgroup rolls 3-4 tiles whose first K-tile is also the one before the last
  (the shape that exposed a tie-induced register copy ahead of an asm wait)."""
  from big_vision_amd import ops, _lib
  lib = _lib.load()
  a = rnd((M, K), dev, 21, dtype=BF16)
  w = rnd((N, K), dev, 22, 0.05, dtype=BF16)
  b = rnd((N,), dev, 23)
  res = rnd((M, N), dev, 24, 2.0)
  kw = dict(a_kmajor=True, b_kmajor=True)

  def run_all():
    y0 = ops.gemm(a, w, bias=b, out_dtype=BF16, **kw)
    g = torch.empty((M, N), device=dev, dtype=BF16)
    h = ops.gemm(a, w, bias=b, out_dtype=BF16, epilogue=ops.EPI_GELU, out2=g, **kw)
    y1 = ops.gemm(a, w, bias=b, out_dtype=F32, epilogue=ops.EPI_RESIDUAL, aux=res, **kw)
    y2 = ops.gemm(a, w, out_dtype=BF16, alpha=0.5, **kw)          # no bias, alpha != 1
    g1 = ops.gemm(a, w, bias=b, out_dtype=BF16, epilogue=ops.EPI_GELU_G, **kw)   # gelu(h) alone (context-free forward)
    assert torch.equal(g1, g), "BV_EPI_GELU_G differs from the activation BV_EPI_GELU writes (256 x 256 kernel)"
    return y0, h, g, y1, y2

  with ops.option("gemm_roll", 0):
    ref = run_all()
  with ops.option("gemm_roll", 7):
    new = [run_all() for _ in range(3)]
  for k, name in enumerate(("bias bf16", "gelu h", "gelu g", "residual f32", "alpha bf16")):
    for r in new[1:]:
      assert torch.equal(r[k], new[0][k]), f"{name}: run-to-run difference"
    if name == "residual f32":
      close(new[0][k], ref[k], 1e-6, 1e-4, name)
    else:
      assert torch.equal(new[0][k], ref[k]), f"{name}: rolling kernel differs from the full-epilogue kernel"


def test_reserved_cus_change_nothing_but_the_grid(dev, fast):
  """BV_OPT_GEMM_RESERVE_CUS = 4: the persistent grid leaves 4 CUs to RCCL during an overlapped backward
  (dp.reserve_cus_for_collectives).  k-major results are bit-identical (a tile's arithmetic does not
  depend on which workgroup runs it); the split-K choice of the weight-gradient GEMM follows the CUs
  in use, so dW matches to accumulation order."""
  from big_vision_amd import ops, _lib
  lib = _lib.load()
  a = rnd((2048, 768), dev, 21, dtype=BF16)
  w = rnd((2304, 768), dev, 22, 0.1, dtype=BF16)
  dy = rnd((2048, 2304), dev, 23, dtype=BF16)
  y0 = ops.gemm(a, w, a_kmajor=True, b_kmajor=True, out_dtype=BF16)
  g0 = torch.zeros((768, 2304), device=dev)
  ops.gemm(a, dy, a_kmajor=False, b_kmajor=False, out=g0, epilogue=ops.EPI_ATOMIC)
  with ops.option("gemm_reserve_cus", 4) as o:
    assert o.old == 0
    y1 = ops.gemm(a, w, a_kmajor=True, b_kmajor=True, out_dtype=BF16)
    g1 = torch.zeros((768, 2304), device=dev)
    ops.gemm(a, dy, a_kmajor=False, b_kmajor=False, out=g1, epilogue=ops.EPI_ATOMIC)
  assert torch.equal(y0, y1)
  close(g1, g0, 1e-5, 1e-3, "dW with 4 reserved CUs")
  close(g0, a.float().T @ dy.float(), 1e-4, 2e-2, "dW")


@pytest.mark.parametrize("g", [2, 4, 5, 7])
def test_grouped_tile_order_changes_no_bit(dev, fast, g):
  """BV_OPT_GEMM_GROUP_N = g: the k-major tiles are walked in groups of g column tiles (an A/B knob for the L2 traffic of
  the wide GEMMs, profiles/NOTES_r04.md; default off).  Which workgroup computes a tile, and when, does not enter
  its arithmetic: the outputs of a multi-tile walk (12 x 49 and 9 x 49 tiles; groups that divide the column count
  and ragged last groups; plain kernel, two-output GELU on the full-epilogue kernel, fp32 +residual on the rolling
  kernel) are bit-identical to the plain order, and every tile is visited exactly once (no row left unwritten)."""
  from big_vision_amd import ops, _lib
  lib = _lib.load()
  M = 12544
  a = rnd((M, 768), dev, 31, dtype=BF16)
  w12 = rnd((3072, 768), dev, 32, 0.05, dtype=BF16)
  w9 = rnd((2304, 768), dev, 33, 0.05, dtype=BF16)
  w3 = rnd((768, 768), dev, 34, 0.05, dtype=BF16)
  b12, b3 = rnd((3072,), dev, 35), rnd((768,), dev, 36)
  res = rnd((M, 768), dev, 37)

  def run():
    y9 = ops.gemm(a, w9, a_kmajor=True, b_kmajor=True, out=torch.full((M, 2304), float("nan"), device=dev, dtype=BF16))
    h = torch.full((M, 3072), float("nan"), device=dev, dtype=BF16)
    gl = torch.full((M, 3072), float("nan"), device=dev, dtype=BF16)
    ops.gemm(a, w12, a_kmajor=True, b_kmajor=True, out=h, out2=gl, bias=b12, epilogue=ops.EPI_GELU)
    x1 = ops.gemm(a, w3, a_kmajor=True, b_kmajor=True, out=torch.full((M, 768), float("nan"), device=dev),
                  bias=b3, epilogue=ops.EPI_RESIDUAL, aux=res)
    return y9, h, gl, x1

  ref = run()
  with ops.option("gemm_group_n", g) as o:
    assert o.old == 0
    got = run()
  for r, o, name in zip(ref, got, ("plain N=2304", "gelu h", "gelu g", "+residual fp32 N=768")):
    assert not torch.isnan(o.float()).any(), f"{name}: a tile was never written with groups of {g}"
    assert torch.equal(r, o), f"{name}: groups of {g} column tiles changed the result"


# ---- round 4: every fused epilogue the step uses, at shapes where each persistent
# workgroup walks 2-4 tiles (588-783 tiles on 256 CUs: DMA ring across tile boundaries, XCD work order,
# rolling epilogue, atomics-based column sums over 49-261 M-tiles), against fp64 - full matrices, not spot
# checks.  The shapes are the image tower's at n = 64 (12 544 = 49 x 256 token rows) and the longest
# rolling walk of the suite (66 816 rows, N = K = 768).  With 4 CUs reserved the same again (the grid an
# overlapped backward leaves to the GEMMs, dp.reserve_cus_for_collectives).
MULTI_TILE_SHAPES = [(12544, 3072, 768), (12544, 768, 3072), (12544, 2304, 768), (66816, 768, 768)]


def _gelu_tanh64(h):
  return torch.nn.functional.gelu(h, approximate="tanh")


def _dgelu_tanh64(h):
  h = h.detach().clone().requires_grad_(True)
  _gelu_tanh64(h).sum().backward()
  return h.grad


def _rel_l2(a, b):
  return ((a.double() - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("reserve", [0, 4])
@pytest.mark.parametrize("M,N,K", MULTI_TILE_SHAPES)
def test_fused_epilogues_multi_tile_vs_fp64(dev, fast, M, N, K, reserve):
  from big_vision_amd import ops, _lib
  lib = _lib.load()
  x = rnd((M, K), dev, 31, dtype=BF16)
  w = rnd((N, K), dev, 32, 1.0 / K ** 0.5, dtype=BF16)
  b = rnd((N,), dev, 33)
  kw = dict(a_kmajor=True, b_kmajor=True)
  acc = x.double() @ w.double().T            # fp64 product of the bf16-rounded operands
  pre = acc + b.double()
  res = rnd((M, N), dev, 34, 2.0)
  hh = rnd((M, N), dev, 35, dtype=BF16)       # a stored pre-activation (light contexts)
  L = 196
  pos = rnd((L, N), dev, 36)
  ctx_ = ops.ctx()
  old = ctx_.set("gemm_reserve_cus", reserve)
  try:
    # plain +bias (QKV / dX shapes), bf16 and fp32 outputs
    close(ops.gemm(x, w, bias=b, out_dtype=F32, **kw), pre, 1e-4, 2e-3, "bias f32")
    close(ops.gemm(x, w, bias=b, out_dtype=BF16, **kw), pre, 1e-2, 1e-2, "bias bf16")
    # +residual on the fp32 stream (the rolling kernel) and on the bf16 stream
    y = ops.gemm(x, w, bias=b, out_dtype=F32, epilogue=ops.EPI_RESIDUAL, aux=res, **kw)
    close(y, pre + res.double(), 1e-4, 2e-3, "residual f32")
    assert torch.equal(y, ops.gemm(x, w, bias=b, out_dtype=F32, epilogue=ops.EPI_RESIDUAL, aux=res, **kw)), \
        "residual f32: run-to-run difference"
    resb = res.to(BF16)
    close(ops.gemm(x, w, bias=b, out_dtype=BF16, epilogue=ops.EPI_RESIDUAL, aux=resb, **kw), pre + resb.double(),
          1e-2, 2e-2, "residual bf16")
    # + position embedding (stem epilogue): aux row = token row mod L
    if M % L == 0:
      y = ops.gemm(x, w, bias=b, out_dtype=F32, epilogue=ops.EPI_POS, aux=pos, aux_rows=L, **kw)
      close(y, pre + pos.double().repeat(M // L, 1), 1e-4, 2e-3, "pos")
    # forward GELU: both outputs (h rounded to bf16, g = gelu of that rounded h)
    g = torch.empty((M, N), device=dev, dtype=BF16)
    h = ops.gemm(x, w, bias=b, out_dtype=BF16, epilogue=ops.EPI_GELU, out2=g, **kw)
    close(h, pre, 1e-2, 1e-2, "gelu: h")
    close(g, _gelu_tanh64(h.double()), 1e-2, 1e-2, "gelu: g")
    # forward GELU_GD: g and gelu'(pre) from the fp32 pre-activation
    g4 = torch.empty((M, N), device=dev, dtype=BF16)
    d4 = torch.empty((M, N), device=dev, dtype=BF16)
    ops.gemm(x, w, bias=b, out=g4, epilogue=ops.EPI_GELU_GD, out2=d4, **kw)
    close(g4, _gelu_tanh64(pre), 1e-2, 1e-2, "gelu_gd: g")
    close(d4, _dgelu_tanh64(pre), 1e-2, 1e-2, "gelu_gd: g'")
    # backward GELU' x aux (+ emitted gelu, + fused column sums = the Dense_0 bias gradient)
    ref_bwd = acc * _dgelu_tanh64(hh.double())
    csref = ref_bwd.sum(0)
    cstol = 2e-3 * ref_bwd.abs().sum(0).max().item()
    cs = torch.ones((N,), device=dev, dtype=F32)
    g2 = torch.empty((M, N), device=dev, dtype=BF16)
    o = ops.gemm(x, w, out_dtype=BF16, epilogue=ops.EPI_GELU_BWD_EMIT, aux=hh, out2=g2, colsum=cs, **kw)
    close(o, ref_bwd, 1e-2, 2e-2, "gelu' emit: dX")
    close(g2, _gelu_tanh64(hh.double()), 1e-2, 1e-2, "gelu' emit: g")
    close(cs, 1.0 + csref, 1e-3, cstol, "gelu' emit: column sums")
    assert _rel_l2(cs - 1.0, csref) < 5e-3, "gelu' emit: column sums rel-L2"
    cs = torch.zeros((N,), device=dev, dtype=F32)
    o2 = ops.gemm(x, w, out_dtype=BF16, epilogue=ops.EPI_GELU_BWD, aux=hh, colsum=cs, **kw)
    assert torch.equal(o2, o), "GELU_BWD and GELU_BWD_EMIT disagree on dX"
    close(cs, csref, 1e-3, cstol, "gelu': column sums")
    # the emitted activation has the forward's bits
    g3 = torch.empty((M, N), device=dev, dtype=BF16)
    ops.gemm(x, w, out_dtype=BF16, epilogue=ops.EPI_GELU_BWD_EMIT, aux=h, out2=g3, **kw)
    assert torch.equal(g3, g), "re-emitted gelu(h) differs from the forward's"
    # backward MUL (full contexts) + fused column sums
    dd = _dgelu_tanh64(hh.double()).to(BF16)
    ref_mul = acc * dd.double()
    cs = torch.zeros((N,), device=dev, dtype=F32)
    o5 = ops.gemm(x, w, out_dtype=BF16, epilogue=ops.EPI_MUL, aux=dd, colsum=cs, **kw)
    close(o5, ref_mul, 1e-2, 2e-2, "mul: dX")
    close(cs, ref_mul.sum(0), 1e-3, 2e-3 * ref_mul.abs().sum(0).max().item(), "mul: column sums")
    assert _rel_l2(cs, ref_mul.sum(0)) < 5e-3, "mul: column sums rel-L2"
  finally:
    ctx_.set("gemm_reserve_cus", old)

