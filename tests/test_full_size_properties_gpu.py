"""Parity at BASELINE.json's FULL sizes through size-independent properties (the fp64 oracle finishes in seconds only at
small sizes; these cases run the kernels at the shapes bench.py times them on: one micro-batch of the headline =
2048 pairs = 401 408 image tokens / 131 072 text tokens, ViT-B widths, global batch 4096 in the loss, 203 M parameters
in the optimizer, and the 512-pair rank shape of the 8-GPU job for a whole step).

Properties used (all through the C ABI, `ops.*`):
  * decomposition: a kernel whose work items are independent (GEMM tiles, (sample, head) pairs, LayerNorm rows,
    optimizer chunks) must produce THE SAME BITS for a row range whether it is computed inside the full-size launch or
    by a launch on that range alone - the small launches are the ones tests/test_kernels_gpu.py / test_gemm256_gpu.py pin
    against fp64 - so every row of the full-size result is tied to a verified computation;
  * linearity: a reduction over tokens (weight-gradient GEMM, fused column sums, LayerNorm scale / bias gradients)
    over [0, T) equals the sum over two halves to fp32 accumulation noise;
  * the 8-rank decomposition of the global sigmoid loss at B = 4096 (row blocks with the positive diagonal at
    rank * n, partial dztxt summed = the reduce_scatter) equals the one-shot loss, and both equal fp64 on the GPU;
  * micro-batch invariance of the whole training step at the rank shape;
plus direct fp64 checks of randomly sampled rows / pairs of the full-size outputs (torch fp64 on the GPU as the checker).
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

BF16, F32, F64 = torch.bfloat16, torch.float32, torch.float64
T_IMG, T_TXT, D, MLP, H = 401408, 131072, 768, 3072, 12   # 2048 pairs x 196 / x 64 tokens, ViT-B / text-B widths


def _rnd(shape, dev, seed, scale=1.0, dtype=F32):
  g = torch.Generator(device=dev).manual_seed(seed)
  return (torch.randn(shape, generator=g, device=dev, dtype=F32) * scale).to(dtype)


def _chunks(T, parts):
  assert T % (parts * 256) == 0
  s = T // parts
  return [(i * s, (i + 1) * s) for i in range(parts)]


def _rel(a, b):
  return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("T", [T_IMG, T_TXT], ids=["image-401408", "text-131072"])
@pytest.mark.parametrize("name", ["qkv bias", "out-proj +residual f32", "fc1 gelu (2 outputs)", "fc2 +residual f32",
                                  "fc2 dX gelu'-emit + colsum", "fc2 dX mul + colsum", "fc1 dX plain"])
def test_kmajor_gemms_of_a_2048_pair_microbatch(dev, name, T):
  """Every k-major GEMM variant of the step at T = 401 408 image-token rows (1568 row tiles, 6-18 rounds of the
  persistent grid) and at the text tower's 131 072 rows (1.5-6 rounds: the ragged last round): the monolithic launch vs
  8 launches on its row ranges - bit-identical outputs, fused column sums to fp32 accumulation order - and 384 random
  rows against fp64."""
  from big_vision_amd import ops
  N, K = {"qkv bias": (3 * D, D), "out-proj +residual f32": (D, D), "fc1 gelu (2 outputs)": (MLP, D),
          "fc2 +residual f32": (D, MLP), "fc2 dX gelu'-emit + colsum": (MLP, D), "fc2 dX mul + colsum": (MLP, D),
          "fc1 dX plain": (D, MLP)}[name]
  x = _rnd((T, K), dev, 1, dtype=BF16)
  w = _rnd((N, K), dev, 2, 1.0 / math.sqrt(K), dtype=BF16)
  b = _rnd((N,), dev, 3)
  aux32 = _rnd((T, N), dev, 4, 2.0) if "residual" in name else None
  aux16 = _rnd((T, N), dev, 5, dtype=BF16) if "dX" in name and "plain" not in name else None

  def run(lo, hi, out, out2, cs):
    kw = dict(a_kmajor=True, b_kmajor=True, out=out[lo:hi])
    xs = x[lo:hi]
    if name == "qkv bias":
      ops.gemm(xs, w, bias=b, **kw)
    elif "residual" in name:
      ops.gemm(xs, w, bias=b, epilogue=ops.EPI_RESIDUAL, aux=aux32[lo:hi], **kw)
    elif name.startswith("fc1 gelu"):
      ops.gemm(xs, w, bias=b, epilogue=ops.EPI_GELU, out2=out2[lo:hi], **kw)
    elif "emit" in name:
      ops.gemm(xs, w, epilogue=ops.EPI_GELU_BWD_EMIT, aux=aux16[lo:hi], out2=out2[lo:hi], colsum=cs, **kw)
    elif "mul" in name:
      ops.gemm(xs, w, epilogue=ops.EPI_MUL, aux=aux16[lo:hi], colsum=cs, **kw)
    else:
      ops.gemm(xs, w, **kw)

  odt = F32 if "residual" in name else BF16
  two = name.startswith("fc1 gelu") or "emit" in name
  full, part = torch.empty((T, N), device=dev, dtype=odt), torch.empty((T, N), device=dev, dtype=odt)
  full2 = torch.empty((T, N), device=dev, dtype=BF16) if two else None
  part2 = torch.empty((T, N), device=dev, dtype=BF16) if two else None
  cs_full, cs_part = torch.zeros((N,), device=dev), torch.zeros((N,), device=dev)
  run(0, T, full, full2, cs_full)
  for lo, hi in _chunks(T, 8):
    run(lo, hi, part, part2, cs_part)
  assert torch.equal(full, part), f"{name}: the full-size launch differs from the launches on its row ranges"
  if two:
    assert torch.equal(full2, part2), f"{name}: second output differs"
  if "colsum" in name:
    assert _rel(cs_full, cs_part) <= 1e-5, f"{name}: fused column sums, full vs sum of the ranges"
    assert _rel(cs_full, full.double().sum(0)) <= 2e-3, f"{name}: fused column sums vs the column sums of the (bf16) output"
  # fp64 on sampled rows
  rows = torch.randint(0, T, (384,), device=dev, generator=torch.Generator(device=dev).manual_seed(9))
  acc = x[rows].double() @ w.double().T
  if name == "qkv bias":
    ref, tol = acc + b.double(), 1e-2
  elif "residual" in name:
    ref, tol = acc + b.double() + aux32[rows].double(), 1e-4
  elif name.startswith("fc1 gelu"):
    ref, tol = acc + b.double(), 1e-2
    g_ref = torch.nn.functional.gelu(full[rows].double(), approximate="tanh")   # g = gelu of the bf16-rounded h
    assert (full2[rows].double() - g_ref).abs().max() <= 1e-2 * max(1.0, g_ref.abs().max().item()), "gelu output"
  elif "emit" in name or "mul" in name:
    h = aux16[rows].double()
    if "emit" in name:
      hh = h.clone().requires_grad_(True)
      torch.nn.functional.gelu(hh, approximate="tanh").sum().backward()
      ref = acc * hh.grad.to(BF16).double()      # gelu' is rounded to bf16 before the product (one definition everywhere)
      assert (full2[rows].double() - torch.nn.functional.gelu(h, approximate="tanh")).abs().max() <= 2e-2, "re-emitted gelu"
    else:
      ref = acc * h
    tol = 1e-2
  else:
    ref, tol = acc, 1e-2
  err = (full[rows].double() - ref).abs().max().item()
  assert err <= tol * max(1.0, ref.abs().max().item()), f"{name}: sampled rows vs fp64: {err:.3e}"


@pytest.mark.parametrize("L,n,H", [(196, 2048, 12), (64, 2048, 12), (441, 512, 16)],
                         ids=["B16-image-196", "text-64", "L16-336-image-441"])
def test_attention_of_a_2048_pair_microbatch(dev, L, n, H):
  """The attention forward and the one-launch backward at the micro-batch of the headline (24 576 (sample, head) pairs,
  every persistent workgroup of attention5.hip walks ~96): full launch vs 8 launches of 256 samples - bit-identical o, lse,
  dqkv - and 3 random samples against fp64.  L = 441, 16 heads: BASELINE configs[3] (L/16 at 336 px, two micro-batches of
  the 1024-pair rank shape), the long-sequence forward and the two-launch backward of attention3.hip."""
  from big_vision_amd import ops
  qkv = _rnd((n * L, 3 * H * 64), dev, 11, 1.2, dtype=BF16)
  d_o = _rnd((n * L, H * 64), dev, 12, dtype=BF16)
  o, lse = ops.attn_fwd(qkv, n, L, H)
  dqkv = ops.attn_bwd(qkv, o, d_o, lse, n, L, H)
  step = n // 8
  for i in range(0, n, step):
    sl = slice(i * L, (i + step) * L)
    o_c, lse_c = ops.attn_fwd(qkv[sl].contiguous(), step, L, H)
    assert torch.equal(o_c, o[sl]) and torch.equal(lse_c, lse[i:i + step]), "forward: chunk differs from the full launch"
    d_c = ops.attn_bwd(qkv[sl].contiguous(), o_c, d_o[sl].contiguous(), lse_c, step, L, H)
    assert torch.equal(d_c, dqkv[sl]), "backward: chunk differs from the full launch"
  for i in (0, n // 2 + 17, n - 1):
    sl = slice(i * L, (i + 1) * L)
    qr = qkv[sl].double().requires_grad_(True)
    q, k, v = qr.view(1, L, 3, H, 64).unbind(2)
    s = torch.einsum("nqhd,nkhd->nhqk", q / 8.0, k)
    p = torch.softmax(s, -1)
    o_ref = torch.einsum("nhqk,nkhd->nqhd", p, v).reshape(L, H * 64)
    o_ref.backward(d_o[sl].double())
    assert (o[sl].double() - o_ref).abs().max() <= 2e-2 * max(1.0, o_ref.abs().max().item())
    assert (lse[i].double() - torch.logsumexp(s, -1)[0]).abs().max() <= 1e-3
    g = qr.grad
    assert (dqkv[sl].double() - g).abs().max() <= 3e-2 * g.abs().max().item()
    for j in range(3):
      a, r = dqkv[sl].double().view(L, 3, -1)[:, j], g.view(L, 3, -1)[:, j]
      assert _rel(a, r) <= 1e-2, (i, j, _rel(a, r))


def test_layernorm_of_a_2048_pair_microbatch(dev):
  """LayerNorm forward / backward (with the residual-gradient add, the bf16 copy, the fused column sums and the
  re-emitted forward output of the light contexts) on 401 408 rows: full launch vs 8 row ranges - bit-identical row
  outputs, parameter gradients / column sums (fp32 atomics) to accumulation order - and 512 random rows against fp64."""
  from big_vision_amd import ops
  T = T_IMG
  x = _rnd((T, D), dev, 21, 1.5) + 0.3
  sc, bi = 1.0 + 0.1 * _rnd((D,), dev, 22), 0.1 * _rnd((D,), dev, 23)
  dy = _rnd((T, D), dev, 24, dtype=BF16)
  dres = _rnd((T, D), dev, 25)
  y, _, mean, rstd = ops.layernorm_fwd(x, sc, bi, rows=T, D=D)
  outs = {}
  for tag, ranges in (("full", [(0, T)]), ("parts", _chunks(T, 8))):
    dx, dxb, yo = torch.empty_like(x), torch.empty((T, D), device=dev, dtype=BF16), torch.empty((T, D), device=dev, dtype=BF16)
    dsc, dbi, dcs = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    yy, mm, rr = torch.empty_like(y), torch.empty_like(mean), torch.empty_like(rstd)
    for lo, hi in ranges:
      a, _, m, r = ops.layernorm_fwd(x[lo:hi], sc, bi, rows=hi - lo, D=D)
      yy[lo:hi], mm[lo:hi], rr[lo:hi] = a, m, r
      ops.layernorm_bwd(dy[lo:hi], x[lo:hi], sc, m, r, rows=hi - lo, D=D, dres=dres[lo:hi], dx=dx[lo:hi], dx_bf16=dxb[lo:hi],
                        dscale=dsc, dbias=dbi, dx_colsum=dcs, bias=bi, y_out=yo[lo:hi])
    outs[tag] = (yy, mm, rr, dx, dxb, yo, dsc, dbi, dcs)
  f, p = outs["full"], outs["parts"]
  for k, nm in enumerate(("y", "mean", "rstd", "dx", "dx bf16", "re-emitted y")):
    assert torch.equal(f[k], p[k]), f"LayerNorm {nm}: full launch differs from its row ranges"
  assert torch.equal(f[0], y) and torch.equal(f[5], y), "the backward must re-emit the forward's bits"
  for k, nm in ((6, "dscale"), (7, "dbias"), (8, "column sums of dx")):
    assert _rel(f[k], p[k]) <= 1e-5, nm
  rows = torch.randint(0, T, (512,), device=dev, generator=torch.Generator(device=dev).manual_seed(29))
  xr = x[rows].double().requires_grad_(True)
  ref = torch.nn.functional.layer_norm(xr, (D,), sc.double(), bi.double(), eps=1e-6)
  assert (y[rows].double() - ref).abs().max() <= 1e-2 * max(1.0, ref.abs().max().item())
  ref.backward(dy[rows].double())
  want = xr.grad + dres[rows].double()
  assert (f[3][rows].double() - want).abs().max() <= 1e-5 * max(1.0, want.abs().max().item()), "dx vs fp64"
  assert _rel(f[8], f[3].double().sum(0)) <= 1e-5, "fused column sums of dx"


def test_weight_gradient_gemm_is_linear_over_401408_tokens(dev):
  """dW = X^T dY with K = 401 408 tokens (split-K over the persistent grid, deterministic slab reduction): the launch
  over all tokens vs two launches over the halves accumulating into one buffer (fp32 accumulation noise), twice the same
  bits run to run, and against fp64 (bf16 operands are exact in fp64: only the fp32 accumulation differs)."""
  from big_vision_amd import ops
  T = T_IMG
  x = _rnd((T, D), dev, 31, dtype=BF16)
  dy = _rnd((T, MLP), dev, 32, dtype=BF16)
  kw = dict(a_kmajor=False, b_kmajor=False, epilogue=ops.EPI_ATOMIC)
  full = torch.zeros((D, MLP), device=dev)
  ops.gemm(x, dy, out=full, **kw)
  again = torch.zeros((D, MLP), device=dev)
  ops.gemm(x, dy, out=again, **kw)
  assert torch.equal(full, again), "the weight-gradient GEMM must be deterministic"
  halves = torch.zeros((D, MLP), device=dev)
  ops.gemm(x[:T // 2], dy[:T // 2], out=halves, **kw)
  ops.gemm(x[T // 2:], dy[T // 2:], out=halves, **kw)
  assert _rel(halves, full) <= 2e-6, _rel(halves, full)
  ref = torch.zeros((D, MLP), device=dev, dtype=F64)
  for lo, hi in _chunks(T, 16):     # fp64 on the GPU, in slices (the fp64 operands of the whole product are 12 GB)
    ref += x[lo:hi].double().T @ dy[lo:hi].double()
  assert _rel(full, ref) <= 2e-5, _rel(full, ref)
  assert (full.double() - ref).abs().max() <= 1e-4 * ref.abs().max()


def test_global_sigmoid_loss_at_batch_4096_in_eight_rank_blocks(dev):
  """The loss of the headline (B = 4096, E = 768) the way 8 ranks compute it - rank r: rows [512 r, 512 (r + 1)) of the
  logits against ALL text embeddings, positive diagonal at 512 r, GLOBAL 1/B, dzimg local, dztxt partial summed over
  the ranks (= the reduce_scatter) - against the one-shot computation and against fp64 autograd of siglip.py:291-306."""
  from big_vision_amd import dp
  from big_vision_amd.trainers.proj.image_text import siglip
  B, E, R = 4096, D, 8
  n = B // R
  zi = torch.nn.functional.normalize(_rnd((B, E), dev, 41), dim=1)
  zt = torch.nn.functional.normalize(_rnd((B, E), dev, 42) + 0.2 * zi, dim=1)
  tp, b = torch.tensor([math.log(10.0)], device=dev), torch.tensor([-10.0], device=dev)
  stats1, dzi1, dzt1 = siglip.sigmoid_loss_fwd_bwd(zi, zt, tp, b, dp.Comm())

  class Rank(dp.Comm):      # a "communicator" whose all-gather returns the full text matrix: the per-rank arithmetic
    def __init__(self, r):
      super().__init__()
      self.rank, self.size, self.active = r, R, False
    def all_gather_rows(self, x):
      return zt
    def reduce_scatter_rows(self, x):
      return x          # partial [B, E]: summed below
  stats8 = torch.zeros(3, device=dev, dtype=F64)
  dzi8 = torch.empty_like(zi)
  dzt8 = torch.zeros_like(zt)
  for r in range(R):
    s, a, part = siglip.sigmoid_loss_fwd_bwd(zi[r * n:(r + 1) * n].contiguous(), zt[r * n:(r + 1) * n].contiguous(), tp, b, Rank(r))
    stats8 += s
    dzi8[r * n:(r + 1) * n] = a
    dzt8 += part
  assert abs(stats8[0].item() - stats1[0].item()) <= 1e-9 * abs(stats1[0].item()), "loss: 8 row blocks vs one shot"
  assert torch.equal(dzi8, dzi1), "dzimg rows do not depend on the row block they are computed in"
  assert _rel(dzt8, dzt1) <= 5e-6 and _rel(stats8[1:], stats1[1:]) <= 1e-7      # (fp32 sums over 8 partials vs one k-ordered chain)
  # fp64 autograd of the reference's expression
  zi64, zt64 = zi.double().requires_grad_(True), zt.double().requires_grad_(True)
  t64, b64 = tp.double().requires_grad_(True), b.double().requires_grad_(True)
  logits = zi64 @ zt64.T * torch.exp(t64) + b64
  m = -torch.ones((B, B), device=dev, dtype=F64) + 2 * torch.eye(B, device=dev, dtype=F64)
  loss = -torch.nn.functional.logsigmoid(m * logits).sum() / B
  loss.backward()
  assert abs(stats1[0].item() - loss.item()) <= 1e-6 * abs(loss.item())
  assert _rel(dzi1, zi64.grad) <= 1e-5 and _rel(dzt1, zt64.grad) <= 1e-5
  assert abs(stats1[1].item() - t64.grad.item()) <= 1e-5 * abs(t64.grad.item())
  assert abs(stats1[2].item() - b64.grad.item()) <= 1e-5 * abs(b64.grad.item())


def test_adam_on_all_203m_parameters_in_slices(dev):
  """The fused clip + Adam + weight decay + schedule + apply kernel on the headline's 203 M-parameter flat buffer: one
  launch over everything vs launches over the 1/8 slices the sharded optimizer owns (offset pointers, the same
  per-chunk hyper-parameter table) - bit-identical parameters, moments and bf16 shadow; the fp64 statistics to
  summation order; and a random sample of elements against the optax chain in fp64."""
  import bench
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  from big_vision_amd import ops
  model = two_towers.Model(image=bench.IMAGE_CFG, text=bench.TEXT_CFG, out_dim=(None, bench.EMB), temperature_init=10.0, bias_init=-10.0)
  config = bench.make_config(20_000)
  state, _ = siglip.make_train_state(model, config, (8, 224, 224, 3), (8, 64), rng=0, total_steps=20_000, device=dev)
  opt, st = state["opt"], state["params"].store
  n_tr = st.trainable_count
  assert n_tr > 200e6 and n_tr % 1024 == 0
  st.ensure_grad().copy_(_rnd((n_tr,), dev, 51, 1e-3))
  p0, sh0 = st.master.clone(), st.shadow.clone()
  k = 5000          # past the warm-up (the schedule is 0 at step 0: nothing would move)
  sched = [fn(k) for fn in opt.schedule_fns]
  gsq = torch.zeros(1, device=dev, dtype=F64)
  ops.sqnorm_(st.grad, gsq)
  bc = (1.0 - opt.b1 ** (k + 1), 1.0 - opt.b2 ** (k + 1))

  def run(ranges):
    p, mu, nu, sh = p0.clone(), torch.zeros_like(opt.mu), torch.zeros_like(opt.nu), sh0.clone()
    stats = torch.zeros(2, device=dev, dtype=F64)
    for lo, hi in ranges:
      ops.adam_step_(p[lo:hi], st.grad[lo:hi], mu[lo:hi], nu[lo:hi], sh[lo:hi], opt.segs, opt.chunk_seg[lo // 1024:], hi - lo,
                     sched, gsq, opt.clip_norm, opt.b1, opt.b2, opt.eps, bc[0], bc[1], stats)
    return p, mu, nu, sh, stats
  full = run([(0, n_tr)])
  chunk = (n_tr // 1024 + 7) // 8 * 1024
  parts = run([(lo, min(n_tr, lo + chunk)) for lo in range(0, n_tr, chunk)])
  for a, b_, nm in zip(full[:4], parts[:4], ("parameters", "mu", "nu", "bf16 shadow")):
    assert torch.equal(a[:n_tr], b_[:n_tr]), f"{nm}: the whole-buffer launch differs from the slices"
  assert _rel(full[4], parts[4]) <= 1e-12
  # sampled elements against the chain in fp64 (moments start at 0: mu = (1 - b1) g, nu = (1 - b2) g^2, bias-corrected
  # for step k); lr multipliers, weight decay and schedules differ per segment, so the check is the direction and the
  # magnitude bound: every element moves against its clipped gradient by at most sched * (lr |u| + wd |p|)
  idx = torch.randint(0, n_tr, (200_000,), device=dev, generator=torch.Generator(device=dev).manual_seed(59))
  g = st.grad[idx].double()
  if opt.clip_norm:
    g = g * min(1.0, opt.clip_norm / math.sqrt(gsq.item()))
  u = ((1 - opt.b1) * g / bc[0]) / (torch.sqrt((1 - opt.b2) * g * g / bc[1]) + opt.eps)
  moved = full[0][idx].double() - p0[idx].double()
  assert max(sched) > 0 and moved.abs().max().item() > 0
  big = u.abs() > 0.5 * u.abs().max()
  assert (torch.sign(moved[big]) == -torch.sign(u[big])).double().mean().item() >= 0.999
  bound = max(sched) * (float(config.lr) * u.abs().max().item() + float(config.wd) * p0[idx].abs().max().item())
  assert moved.abs().max().item() <= 1.001 * bound + 1e-12, (moved.abs().max().item(), bound)
  one = torch.tensor(1.0, dtype=F32)      # the kernel forms 1 - b in fp32: 1 - 0.999f = 0.00100004673
  omb1, omb2 = float(one - torch.tensor(opt.b1, dtype=F32)), float(one - torch.tensor(opt.b2, dtype=F32))
  assert _rel(full[1][idx], omb1 * g) <= 1e-6 and _rel(full[2][idx], omb2 * g * g) <= 1e-6, "moments vs fp64"


def test_training_step_at_the_rank_shape_is_microbatch_invariant(dev):
  """BASELINE configs[2] on 8 GPUs = 512 pairs per rank: the product update_fn on 512 pairs in ONE pass vs two
  micro-batches of 256 (the two-pass scheme of the N = 1 headline: embed all, loss, back-propagate each) - the same loss
  and the same gradients to accumulation noise, at the full model size (B/16 + text-B, 12 + 12 blocks)."""
  import bench
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  n = 512
  image, text = bench.synthetic_batch(n, dev, seed=3)
  res = {}
  for micro in (0, 256):
    model = two_towers.Model(image=bench.IMAGE_CFG, text=bench.TEXT_CFG, out_dim=(None, bench.EMB), temperature_init=10.0, bias_init=-10.0)
    config = bench.make_config(20_000)
    config.microbatch = micro
    config.microbatch_keep = "all"
    state, _ = siglip.make_train_state(model, config, (n, bench.RES, bench.RES, 3), (n, bench.SEQ), rng=0, total_steps=20_000, device=dev)
    state, meas = siglip.make_update_fn(model, config)(state, None, {"image": image, "labels": text})
    torch.cuda.synchronize()
    res[micro] = (meas["training_loss"].item(), meas["l2_grads"].item(), state["params"].store.grad.clone())
    del state, model
    torch.cuda.empty_cache()
  (l0, g0, v0), (l1, g1, v1) = res[0], res[256]
  assert abs(l0 - l1) <= 1e-6 * abs(l0), (l0, l1)
  assert abs(g0 - g1) <= 1e-3 * g0, (g0, g1)
  cos = torch.dot(v0.double(), v1.double()).item() / (v0.double().norm().item() * v1.double().norm().item())
  assert cos >= 0.99999 and _rel(v1, v0) <= 1e-3, (cos, _rel(v1, v0))
