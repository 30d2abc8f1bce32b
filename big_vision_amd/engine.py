"""Explicit forward/backward executors for the transformer pieces of the hot path.

The reference builds these with flax.linen modules and lets jax.value_and_grad
derive the backward (big_vision/models/vit.py:57-183,
trainers/proj/image_text/siglip.py:311).  Here forward and backward are spelled
out as sequences of libbvhip kernels so that every fusion is explicit:

  forward block  (vit.py:88-112):   LN -> QKV GEMM -> attention -> out-proj GEMM
     (+bias +residual epilogue) -> LN -> fc1 GEMM (+bias, GELU epilogue) ->
     fc2 GEMM (+bias +residual epilogue)
  backward block: dW GEMMs (split-K fp32 atomics straight into the flat grad
     buffer), dX GEMMs (GELU' fused into the fc2 dX epilogue), attention
     backward, LayerNorm backward fused with the residual-gradient add and the
     bf16 copy the next GEMM needs.

The residual stream and all reductions are fp32; GEMM / attention operands are
bf16 (fp32 MFMA accumulate).  No autograd graph is built.
"""
from __future__ import annotations

import math
import weakref
from typing import List, Optional

import torch

from big_vision_amd import ops
from big_vision_amd.params import Entry, ParamStore

BF16, F32 = torch.bfloat16, torch.float32


# ----------------------------------------------------------------- inits -----
def init_zeros(gen, shape):
  return torch.zeros(shape)


def init_ones(gen, shape):
  return torch.ones(shape)


def init_normal(std):
  return lambda gen, shape: torch.randn(shape, generator=gen) * std


def init_const(val):
  return lambda gen, shape: torch.full(shape, float(val))


def init_xavier_uniform(fan_in, fan_out):
  lim = math.sqrt(6.0 / (fan_in + fan_out))
  return lambda gen, shape: (torch.rand(shape, generator=gen) * 2 - 1) * lim


def init_lecun_normal(fan_in):
  # flax default kernel_init = variance_scaling(1.0, "fan_in", "truncated_normal")
  std = math.sqrt(1.0 / fan_in) / 0.87962566103423978

  def f(gen, shape):
    t = torch.empty(shape)
    torch.nn.init.trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=gen)
    return t * std
  return f


# --------------------------------------------------------------- entries -----
def ln_entries(prefix):
  return lambda d: [Entry(f"{prefix}/scale", (d,), init_ones), Entry(f"{prefix}/bias", (d,), init_zeros)]


def mha_entries(prefix, D, H, fused="qkv"):
  """Flax MultiHeadDotProductAttention params; q/k/v stored fused.

  fused="qkv": self-attention, one [D,3,H,Dh] tensor.  fused="kv": MAP head,
  query separate and key/value fused as [D,2,H,Dh].
  """
  Dh = D // H
  xav = init_xavier_uniform(D, D)
  ents = []
  if fused == "qkv":
    names = ("query", "key", "value")
    ents.append(Entry(f"{prefix}/qkv/kernel", (D, 3, H, Dh), xav,
                      {f"{prefix}/{n}/kernel": (1, i) for i, n in enumerate(names)}))
    ents.append(Entry(f"{prefix}/qkv/bias", (3, H, Dh), init_zeros,
                      {f"{prefix}/{n}/bias": (0, i) for i, n in enumerate(names)}))
  else:
    ents.append(Entry(f"{prefix}/query/kernel", (D, H, Dh), xav))
    ents.append(Entry(f"{prefix}/query/bias", (H, Dh), init_zeros))
    names = ("key", "value")
    ents.append(Entry(f"{prefix}/kv/kernel", (D, 2, H, Dh), xav,
                      {f"{prefix}/{n}/kernel": (1, i) for i, n in enumerate(names)}))
    ents.append(Entry(f"{prefix}/kv/bias", (2, H, Dh), init_zeros,
                      {f"{prefix}/{n}/bias": (0, i) for i, n in enumerate(names)}))
  ents.append(Entry(f"{prefix}/out/kernel", (H, Dh, D), xav))
  ents.append(Entry(f"{prefix}/out/bias", (D,), init_zeros))
  return ents


def mlp_entries(prefix, D, M):
  """MlpBlock params (vit.py:63-78): xavier_uniform kernels, normal(1e-6) biases."""
  return [Entry(f"{prefix}/Dense_0/kernel", (D, M), init_xavier_uniform(D, M)),
          Entry(f"{prefix}/Dense_0/bias", (M,), init_normal(1e-6)),
          Entry(f"{prefix}/Dense_1/kernel", (M, D), init_xavier_uniform(M, D)),
          Entry(f"{prefix}/Dense_1/bias", (D,), init_normal(1e-6))]


def encoder_entries(prefix, depth, D, H, M):
  """vit.Encoder params, python-loop layout (vit.py:149-160)."""
  ents = []
  for i in range(depth):
    P = f"{prefix}/encoderblock_{i}"
    ents += ln_entries(f"{P}/LayerNorm_0")(D)
    ents += mha_entries(f"{P}/MultiHeadDotProductAttention_0", D, H, "qkv")
    ents += ln_entries(f"{P}/LayerNorm_1")(D)
    ents += mlp_entries(f"{P}/MlpBlock_0", D, M)
  ents += ln_entries(f"{prefix}/encoder_norm")(D)
  return ents


def map_entries(prefix, D, H, M):
  """MAPHead params (vit.py:163-183)."""
  # xavier_uniform on shape (1, 1, D): Flax takes fan_in from axis -2 (= 1) and fan_out from axis -1
  ents = [Entry(f"{prefix}/probe", (1, 1, D), init_xavier_uniform(1, D))]
  ents += mha_entries(f"{prefix}/MultiHeadDotProductAttention_0", D, H, "kv")
  ents += ln_entries(f"{prefix}/LayerNorm_0")(D)
  ents += mlp_entries(f"{prefix}/MlpBlock_0", D, M)
  return ents


# ------------------------------------------------------------- helpers -------
# Residual-stream dtype of the encoder stacks (trainer option config.residual_stream).  fp32 is the
# reference's arithmetic (models/vit.py keeps activations in fp32; only matmul inputs are cast).  bf16
# halves the LayerNorm traffic and the +residual GEMM epilogues; measured against the parity bounds
# by tools/bf16_residual_budget.py (oracle) and the -m gpu step tests.  Everything outside
# Encoder.fwd / Encoder.bwd sees fp32 tensors in both modes.
import threading

_tls = threading.local()   # per host thread: two trainers driving two streams from two threads do not race


def set_residual_stream(dtype):
  """dtype: torch.float32 / torch.bfloat16 or their names; returns the previous setting (of this thread)."""
  old = residual_stream()
  if isinstance(dtype, str):
    dtype = {"float32": F32, "fp32": F32, "f32": F32, "bfloat16": BF16, "bf16": BF16}[dtype]
  if dtype not in (F32, BF16):
    raise ValueError(f"residual_stream must be float32 or bfloat16, got {dtype}")
  _tls.stream = dtype
  return old


def residual_stream():
  return getattr(_tls, "stream", F32)


class _W:
  """A weight tensor resolved against a store: bf16 shadow (2-D), fp32 master,
  and the fp32 grad view (None if frozen)."""

  def __init__(self, store: ParamStore, name: str, shape2d=None):
    self.name = name
    sh = store.t(name, "shadow")
    try:
      ma = store.t(name, "master")
    except KeyError:       # sharded parameters: the fp32 master of a matmul kernel another rank owns (never read by a kernel)
      ma = None
    g = store.g(name) if getattr(store, "want_grads", False) else None
    if shape2d is not None:
      sh = sh.view(shape2d)
      ma = ma.view(shape2d) if ma is not None else None
      g = g.view(shape2d) if g is not None else None
    self.bf, self.f32, self.grad = sh, ma, g
    self.store = store
    self._t, self._t_ver = None, -1
    # frozen tensors (config.schedule None: the image tower of LiT) change only on init / load, not at every
    # optimizer step: their transposed image is rebuilt against store.static_version
    self.frozen = name in store.frozen
    # input width not a multiple of 8 (the 14 x 14 x 3 = 588 stem of So400m/14): the GEMM operands are
    # padded to kpad columns / rows of zeros (16-byte operand loads); the activations arrive padded
    self.kpad = None
    if self.bf.dim() == 2 and self.bf.shape[0] % 8:
      self.kpad = (self.bf.shape[0] + 7) // 8 * 8
    # output width not a multiple of 8 (a classification head with 10 or 1000 + 1 classes, rep_size = 12): the
    # GEMMs run on operands padded with zero columns to npad (linear_fwd / linear_bwd_w / linear_bwd_x below) and
    # the results are cut back; only heads and pre_logits ever take this path, their GEMMs are tiny
    self.npad = None
    if self.bf.dim() == 2 and self.bf.shape[1] % 8:
      self.npad = (self.bf.shape[1] + 7) // 8 * 8
    self._np, self._np_ver = None, -1

  def bf_t(self):
    """[out][in] bf16 image of a 2-D (in,out) kernel, re-transposed only when the
    shadow changed (once per optimizer step): lets the forward projections run
    on the k-major ("NT") GEMM path.  The first use transposes this weight alone and enrols it in the
    store's batch; after that the first stale weight of a step refreshes every enrolled image in ONE
    launch (_Twins)."""
    ver = self.store.static_version if self.frozen else self.store.shadow_version
    if self._t_ver != ver:
      if self._t is None:
        K, N = self.bf.shape
        self._t = (torch.zeros((self.npad or N, self.kpad or K), device=self.bf.device, dtype=BF16)
                   if (self.kpad or self.npad) else torch.empty((N, K), device=self.bf.device, dtype=BF16))
        ops.transpose_bf16(self.bf, self._t[:N, :K])
        self._t_ver = ver
        _Twins.of(self.store).add(self)
      else:
        _Twins.of(self.store).refresh(self.frozen, ver)
    return self._t

  def bf_np(self):
    """[in (padded to kpad)][out (padded to npad)] bf16 copy of a kernel with a ragged width (zero padding): the k-major
    B operand of the dX GEMM.  Re-copied when the shadow changed."""
    ver = self.store.static_version if self.frozen else self.store.shadow_version
    if self._np_ver != ver:
      K, N = self.bf.shape
      if self._np is None:
        self._np = torch.zeros((self.kpad or K, self.npad or N), device=self.bf.device, dtype=BF16)
      self._np[:K, :N].copy_(self.bf)
      self._np_ver = ver
    return self._np


class _Twins:
  """The transposed bf16 images (_W.bf_t) of one store's projection kernels, refreshed together: ~100
  weights of the two towers in one table-driven launch (ops.transpose_bf16_batched) instead of one 10 us
  launch each - 1 ms of a 96 ms step at n = 512.  Trainable and frozen weights form separate batches (a
  frozen image changes only with store.static_version).  The device table holds raw addresses: the batch
  keeps the tensors it names alive and is rebuilt when a member joins or dies."""

  def __init__(self):
    self.members = {False: [], True: []}     # frozen? -> [weakref to _W]
    self.batch = {False: None, True: None}   # frozen? -> (table, n, tiles, tensors kept alive)

  @staticmethod
  def of(store) -> "_Twins":
    tw = getattr(store, "_twins", None)
    if tw is None:
      tw = store._twins = _Twins()
    return tw

  def add(self, w: _W):
    self.members[w.frozen].append(weakref.ref(w))
    self.batch[w.frozen] = None

  def refresh(self, frozen: bool, ver: int):
    live = [w for w in (r() for r in self.members[frozen]) if w is not None]
    if self.batch[frozen] is None or len(live) != len(self.members[frozen]):
      self.members[frozen] = [weakref.ref(w) for w in live]
      pairs = [(w.bf, w._t[:w.bf.shape[1], :w.bf.shape[0]]) for w in live]
      self.batch[frozen] = ops.transpose_table(pairs, live[0].bf.device) + (pairs,)
    table, n, tiles, _ = self.batch[frozen]
    ops.transpose_bf16_batched(table, n, tiles)
    for w in live:
      w._t_ver = ver


def refresh_twins(store):
  """Brings every enrolled transposed weight image of `store` up to date NOW, on the current stream.  The lazy refresh
  in _W.bf_t() runs on whichever stream first touches a stale weight and rewrites ALL images in one launch; a caller
  that is about to run the towers on two streams (TwoTowersExec, config.tower_streams = 2) calls this before the fork
  so that no stream reads an image another stream is rewriting."""
  tw = getattr(store, "_twins", None)
  if tw is None:
    return
  for frozen in (False, True):
    ver = store.static_version if frozen else store.shadow_version
    live = [w for w in (r() for r in tw.members[frozen]) if w is not None]
    if any(w._t_ver != ver for w in live):
      tw.refresh(frozen, ver)


def record_stream_tree(obj, stream):
  """`record_stream(stream)` on every CUDA tensor inside a saved context (tuples / lists / dicts / objects with
  __dict__): the allocator must not hand the blocks back to the pool of the stream that produced them while
  `stream` still reads them."""
  seen = set()

  def walk(o):
    if id(o) in seen:
      return
    seen.add(id(o))
    if torch.is_tensor(o):
      if o.is_cuda:
        o.record_stream(stream)
    elif isinstance(o, (list, tuple)):
      for v in o:
        walk(v)
    elif isinstance(o, dict):
      for v in o.values():
        walk(v)
    elif hasattr(o, "__dict__") and not isinstance(o, (type, _W)):
      walk(vars(o))

  walk(obj)


def _pad_cols(t, width):
  """[rows, N] -> [rows, width] with zero columns on the right (heads / pre_logits only: tiny tensors)."""
  return torch.nn.functional.pad(t, (0, width - t.shape[-1]))


def _ragged(x_bf, w: _W):
  """True when this product needs host-side padding: a ragged OUTPUT width, or a ragged INPUT width whose activation
  arrives unpadded (the head behind a rep_size = 12 pre_logits; the So400m/14 stem hands over padded patches)."""
  return bool(w.npad or (w.kpad and x_bf is not None and x_bf.shape[-1] != w.kpad))


def linear_fwd(x_bf, w: _W, b: Optional[_W], **kw):
  if _ragged(x_bf, w):
    assert "out" not in kw and "out2" not in kw and "aux" not in kw, "padded widths: plain epilogue only"
    N = w.bf.shape[1]
    if w.kpad and x_bf.shape[-1] != w.kpad:
      x_bf = _pad_cols(x_bf, w.kpad)
    bias = None if b is None else (_pad_cols(b.f32, w.npad) if w.npad else b.f32)
    y = ops.gemm(x_bf, w.bf_t(), a_kmajor=True, b_kmajor=True, bias=bias, **kw)
    return y[:, :N].contiguous() if w.npad else y
  return ops.gemm(x_bf, w.bf_t(), a_kmajor=True, b_kmajor=True, bias=None if b is None else b.f32, **kw)


def linear_bwd_w(x_bf, dy_bf, w: _W, b: Optional[_W], dy_for_bias=None):
  """dW += x^T dy (split-K atomics into the grad buffer), db += colsum(dy)."""
  if _ragged(x_bf, w):
    # ragged widths: the product of the zero-padded operands lands in a [K (padded)][N (padded)] scratch, its real
    # block is added to the gradient; the bias gradient is a column sum of the (tiny) cotangent
    K, N = w.bf.shape
    if w.grad is not None:
      xp = _pad_cols(x_bf, w.kpad) if (w.kpad and x_bf.shape[-1] != w.kpad) else x_bf
      dyp = _pad_cols(dy_bf, w.npad) if w.npad else dy_bf
      tmp = torch.zeros((w.kpad or K, w.npad or N), device=x_bf.device, dtype=F32)
      ops.gemm(xp, dyp, a_kmajor=False, b_kmajor=False, out=tmp, epilogue=ops.EPI_ATOMIC)
      w.grad.add_(tmp[:K, :N])
    if b is not None and b.grad is not None:
      if w.npad:
        b.grad.add_((dy_bf if dy_for_bias is None else dy_for_bias).float().sum(0))
      else:
        ops.colsum(dy_bf if dy_for_bias is None else dy_for_bias, b.grad)
    return
  if w.grad is not None and w.kpad:
    # padded input width: the product lands in a [kpad][out] scratch (its last rows are exactly 0: the
    # pad columns of x are) and the real rows are added to the gradient
    K, N = w.grad.shape
    tmp = torch.zeros((w.kpad, N), device=w.grad.device, dtype=F32)
    ops.gemm(x_bf, dy_bf, a_kmajor=False, b_kmajor=False, out=tmp, epilogue=ops.EPI_ATOMIC)
    ops.batchsum(tmp, w.grad, 1, K, N)
  elif w.grad is not None:
    ops.gemm(x_bf, dy_bf, a_kmajor=False, b_kmajor=False, out=w.grad, epilogue=ops.EPI_ATOMIC)
  if b is not None and b.grad is not None:
    ops.colsum(dy_bf if dy_for_bias is None else dy_for_bias, b.grad)


def linear_bwd_x(dy_bf, w: _W, out_dtype=BF16, **kw):
  if w.npad or w.kpad:
    # ragged widths: the contraction runs over the padded output width and produces the padded input width (zero rows
    # / columns on the operands), cut back to the real one
    K = w.bf.shape[0]
    dyp = _pad_cols(dy_bf, w.npad) if w.npad else dy_bf
    dx = ops.gemm(dyp, w.bf_np(), a_kmajor=True, b_kmajor=True, out_dtype=out_dtype, **kw)
    return dx[:, :K].contiguous() if w.kpad else dx
  return ops.gemm(dy_bf, w.bf, a_kmajor=True, b_kmajor=True, out_dtype=out_dtype, **kw)


class Dropout:
  """Dropout state of one forward pass: the rate and a 64-bit key from which every dropout site (models/vit.py:76,
  100, 109, 228) derives its own key.  The kernels regenerate the keep bits from (site key, element index), so a
  saved context only remembers keys; `fold(...)` derives the state of a sub-computation (tower, micro-batch)."""
  M64 = (1 << 64) - 1

  def __init__(self, rate, key=0):
    if not 0.0 <= float(rate) < 1.0:
      raise ValueError(f"dropout rate must be in [0, 1), got {rate}")
    self.rate, self.base = float(rate), int(key) & self.M64

  @classmethod
  def _mix(cls, h, v):
    # splitmix64 finaliser over the running key and the next id
    z = (h + 0x9E3779B97F4A7C15 + (int(v) & cls.M64) * 0xBF58476D1CE4E5B9) & cls.M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & cls.M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & cls.M64
    return z ^ (z >> 31)

  def key(self, *ids):
    h = self.base
    for v in ids:
      if isinstance(v, str):
        for ch in v.encode():
          h = self._mix(h, ch)
      else:
        h = self._mix(h, v)
    return h

  def fold(self, *ids):
    return Dropout(self.rate, self.key(*ids))


# ids of the dropout sites of one encoder block (Block._fwd_drop) and of the tower stem
DROP_SA, DROP_GELU, DROP_MLP, DROP_POSEMB = 1, 2, 3, 4


class LN:
  def __init__(self, store, prefix):
    self.scale = _W(store, f"{prefix}/scale")
    self.bias = _W(store, f"{prefix}/bias")

  def fwd(self, x, rows, D, **kw):
    return ops.layernorm_fwd(x, self.scale.f32, self.bias.f32, rows=rows, D=D, **kw)

  def bwd(self, dy, x, mean, rstd, rows, D, y_out=None, **kw):
    """y_out (bf16 [rows, D]): also re-emit this LayerNorm's forward output (fp32 x only)."""
    if y_out is not None:
      kw.update(bias=self.bias.f32, y_out=y_out)
    return ops.layernorm_bwd(dy, x, self.scale.f32, mean, rstd, rows=rows, D=D,
                             dscale=self.scale.grad, dbias=self.bias.grad, **kw)


class MLP:
  def __init__(self, store, prefix, D, M):
    self.w1 = _W(store, f"{prefix}/Dense_0/kernel"); self.b1 = _W(store, f"{prefix}/Dense_0/bias")
    self.w2 = _W(store, f"{prefix}/Dense_1/kernel"); self.b2 = _W(store, f"{prefix}/Dense_1/bias")
    self.M = M

  def fwd_drop(self, y_bf, resid, rate, k_gelu, k_out, keep_branch=False):
    """resid + drop(fc2(drop(gelu(fc1(y))))) (models/vit.py:72-77,109) on the fp32 stream; returns (out, hd, g, branch)
    in the form of full contexts: g = drop(gelu(h)) is fc2's operand and hd = gelu'(h) scaled by the SAME keep bits, so
    the unchanged backward (dW2 = g^T dout, dH = (dout W2^T) o hd, BV_EPI_MUL) differentiates the dropped activation.
    branch: the MLP output BEFORE its dropout (the reference's out["mlp"], vit.py:108) when keep_branch, else None."""
    g = torch.empty((y_bf.shape[0], self.M), device=y_bf.device, dtype=BF16)
    hd = torch.empty_like(g)
    linear_fwd(y_bf, self.w1, self.b1, out=g, epilogue=ops.EPI_GELU_GD, out2=hd)
    ops.dropout_bf16_(g, k_gelu, rate, b=hd)
    branch = linear_fwd(g, self.w2, self.b2, out_dtype=F32)
    if keep_branch:
      return ops.dropout_f32(branch, k_out, rate, addend=resid), hd, g, branch
    return ops.dropout_f32(branch, k_out, rate, addend=resid, out=branch), hd, g, None

  def fwd(self, y_bf, resid, keep_g=True, ctx=True):
    """resid + fc2(gelu(fc1(y))) ; returns (out, hd bf16, g bf16 or None).

    ctx = False (a forward that saves nothing: a frozen tower, inference): fc1's epilogue writes gelu(h) only
    (BV_EPI_GELU_G, bit-identical to the activation the saving epilogues write) and (out, None, None) is returned.

    keep_g (full contexts): the fc1 epilogue BV_EPI_GELU_GD evaluates gelu AND its derivative on the fp32
    pre-activation - they share the exp / rcp - and writes g = gelu(h) and hd = gelu'(h); the pre-activation
    itself is never stored and the backward's fc2 dX GEMM only multiplies by hd (BV_EPI_MUL).
    not keep_g (light contexts): hd = the bf16 pre-activation h (BV_EPI_GELU: g is applied to the rounded h
    that is stored) and g is dropped after fc2; the backward re-derives both g and gelu'(h) from h
    (BV_EPI_GELU_BWD_EMIT, bit-identical g)."""
    if not ctx:
      g = linear_fwd(y_bf, self.w1, self.b1, out_dtype=BF16, epilogue=ops.EPI_GELU_G)
      return linear_fwd(g, self.w2, self.b2, out_dtype=resid.dtype, epilogue=ops.EPI_RESIDUAL, aux=resid), None, None
    g = torch.empty((y_bf.shape[0], self.M), device=y_bf.device, dtype=BF16)
    if keep_g:
      hd = torch.empty_like(g)
      linear_fwd(y_bf, self.w1, self.b1, out=g, epilogue=ops.EPI_GELU_GD, out2=hd)
    else:
      hd = linear_fwd(y_bf, self.w1, self.b1, out_dtype=BF16, epilogue=ops.EPI_GELU, out2=g)
    out = linear_fwd(g, self.w2, self.b2, out_dtype=resid.dtype, epilogue=ops.EPI_RESIDUAL, aux=resid)
    return out, hd, (g if keep_g else None)

  def bwd(self, dout_f32, dout_bf, y_bf, hd, g, bias2_done=False, dx_kw=None):
    """Returns dy (bf16) = gradient w.r.t. the MLP input y.  bias2_done: the Dense_1 bias
    gradient (column sums of dout) was already accumulated by the LayerNorm-backward kernel
    that produced dout (bv_layernorm_bwd dx_colsum).  g given: hd = gelu'(h) from the forward (see fwd).
    g = None ("light" context): hd = the pre-activation h; the dX GEMM that needs gelu'(h) anyway
    re-emits the activation (BV_EPI_GELU_BWD_EMIT, bit-identical to the forward's g)."""
    # The Dense_0 bias gradient (column sums of dh) is reduced inside the same epilogue.
    if g is None:
      g = torch.empty_like(hd)
      dh = linear_bwd_x(dout_bf, self.w2, epilogue=ops.EPI_GELU_BWD_EMIT, aux=hd, out2=g, colsum=self.b1.grad)
      linear_bwd_w(g, dout_bf, self.w2, None if bias2_done else self.b2, dy_for_bias=dout_f32)
      del g
    else:
      linear_bwd_w(g, dout_bf, self.w2, None if bias2_done else self.b2, dy_for_bias=dout_f32)
      dh = linear_bwd_x(dout_bf, self.w2, epilogue=ops.EPI_MUL, aux=hd, colsum=self.b1.grad)
    # dx_kw: epilogue of the last dX GEMM (post-LN blocks add the residual branch there, fp32 out)
    if y_bf is None:   # the caller gets y from the LayerNorm backward and runs the Dense_0 dW GEMM afterwards
      return linear_bwd_x(dh, self.w1, **(dx_kw or {})), dh
    linear_bwd_w(y_bf, dh, self.w1, None)
    return linear_bwd_x(dh, self.w1, **(dx_kw or {}))


# ------------------------------------------------------------- encoder -------
class Block:
  """Encoder1DBlock (vit.py:81-112)."""

  def __init__(self, store, P, D, H, M):
    A = f"{P}/MultiHeadDotProductAttention_0"
    self.D, self.H, self.M = D, H, M
    self.ln0 = LN(store, f"{P}/LayerNorm_0")
    self.ln1 = LN(store, f"{P}/LayerNorm_1")
    self.wqkv = _W(store, f"{A}/qkv/kernel", (D, 3 * D))
    self.bqkv = _W(store, f"{A}/qkv/bias", (3 * D,))
    self.wo = _W(store, f"{A}/out/kernel", (D, D))
    self.bo = _W(store, f"{A}/out/bias")
    self.mlp = MLP(store, f"{P}/MlpBlock_0", D, M)

  def fwd(self, x, n, L, light=False, kv_len=None, drop=None, collect=False, ctx=True):
    """kv_len (int32 [n], optional): key-padding length per sample (NaFlex, naflex_vit.py:84-113).
    ctx = False: the caller keeps no context of this block (Encoder.fwd with save falsy): the MLP writes gelu(h) only.
    drop (Dropout of THIS block, rate > 0): the dropout sites of vit.py:100,109 and :76, see _fwd_drop.
    light (True, or "g" = only the second item): the saved context drops what the backward can re-derive cheaply - the two
    LayerNorm outputs (re-normalised from x / x1) and gelu(h) (re-emitted by the fc2 dX
    GEMM) - one third of the block's activation bytes."""
    T, D, H = n * L, self.D, self.H
    if drop is not None and drop.rate > 0.0:
      return self._fwd_drop(x, n, L, kv_len, drop, collect)
    y0, _, mean0, rstd0 = self.ln0.fwd(x, T, D)
    qkv = linear_fwd(y0, self.wqkv, self.bqkv, out_dtype=BF16)
    o, lse = ops.attn_fwd(qkv, n, L, H, kv_len=kv_len)
    x1 = linear_fwd(o, self.wo, self.bo, out_dtype=x.dtype, epilogue=ops.EPI_RESIDUAL, aux=x)   # fp32 or bf16 stream
    y1, _, mean1, rstd1 = self.ln1.fwd(x1, T, D)
    x2, h, g = self.mlp.fwd(y1, x1, keep_g=not light, ctx=ctx)
    if light is True:     # light == "g": only gelu(h) is dropped, the LayerNorm outputs stay
      y0 = y1 = None
    return x2, (x, mean0, rstd0, y0, qkv, o, lse, x1, mean1, rstd1, y1, h, g)

  def _fwd_drop(self, x, n, L, kv_len, drop, collect=False):
    """Encoder1DBlock in train mode with dropout > 0 (vit.py:90-111): x1 = x + drop(attention branch), x2 = x1 +
    drop(MLP branch), drop(gelu(h)) inside the MLP.  fp32 stream, full contexts.  The residual adds cannot ride in
    the GEMM epilogues here: each branch GEMM writes fp32 and ONE element-wise kernel applies the mask and adds the
    stream (bv_dropout_f32).  The context carries the site keys as a 14th entry (rate, k_sa, k_mlp, pre); pre = the
    two branch outputs BEFORE their dropout when `collect` (the reference publishes those as out["sa"] / out["mlp"],
    vit.py:98,108), which Encoder.fwd takes out of the context again."""
    if x.dtype != F32:
      raise NotImplementedError("dropout > 0 runs on the float32 residual stream only")
    T, D, H = n * L, self.D, self.H
    k_sa, k_gelu, k_mlp = drop.key(DROP_SA), drop.key(DROP_GELU), drop.key(DROP_MLP)
    y0, _, mean0, rstd0 = self.ln0.fwd(x, T, D)
    qkv = linear_fwd(y0, self.wqkv, self.bqkv, out_dtype=BF16)
    o, lse = ops.attn_fwd(qkv, n, L, H, kv_len=kv_len)
    branch = linear_fwd(o, self.wo, self.bo, out_dtype=F32)
    x1 = ops.dropout_f32(branch, k_sa, drop.rate, addend=x, out=None if collect else branch)
    y1, _, mean1, rstd1 = self.ln1.fwd(x1, T, D)
    x2, hd, g, mlp_pre = self.mlp.fwd_drop(y1, x1, drop.rate, k_gelu, k_mlp, keep_branch=collect)
    pre = (branch, mlp_pre) if collect else None
    return x2, (x, mean0, rstd0, y0, qkv, o, lse, x1, mean1, rstd1, y1, hd, g, (drop.rate, k_sa, k_mlp, pre))

  def _bwd_drop(self, saved, dx2, n, L, kv_len):
    """Backward of _fwd_drop.  The gradient of a dropped branch is the stream's gradient under the branch's mask:
    the bf16 operand of the branch's dX / dW GEMMs is written by the dropout kernel (instead of the LayerNorm
    backward's plain bf16 copy) and the branch biases take ITS column sums (the fused column sums of the LayerNorm
    backward would be those of the unmasked stream)."""
    x, mean0, rstd0, y0, qkv, o, lse, x1, mean1, rstd1, y1, hd, g, (rate, k_sa, k_mlp, _) = saved
    T, D, H = n * L, self.D, self.H
    dmlp_bf = ops.dropout_f32(dx2, k_mlp, rate, out_bf16=torch.empty((T, D), device=dx2.device, dtype=BF16))
    dy1 = self.mlp.bwd(None, dmlp_bf, y1, hd, g, bias2_done=False)
    del dmlp_bf
    dx1 = self.ln1.bwd(dy1, x1, mean1, rstd1, T, D, dres=dx2)
    dsa_bf = ops.dropout_f32(dx1, k_sa, rate, out_bf16=torch.empty((T, D), device=dx2.device, dtype=BF16))
    linear_bwd_w(o, dsa_bf, self.wo, self.bo)
    d_o = linear_bwd_x(dsa_bf, self.wo)
    del dsa_bf
    dqkv = ops.attn_bwd(qkv, o, d_o, lse, n, L, H, dbias=self.bqkv.grad, kv_len=kv_len)
    linear_bwd_w(y0, dqkv, self.wqkv, None)
    dy0 = linear_bwd_x(dqkv, self.wqkv)
    dx_bf = torch.empty((T, D), device=dx2.device, dtype=BF16)
    dx = self.ln0.bwd(dy0, x, mean0, rstd0, T, D, dres=dx1, dx_bf16=dx_bf)
    return dx, dx_bf

  def bwd(self, saved, dx2, dx2_bf, n, L, b2_done=False, next_b2=None, kv_len=None):
    """b2_done: this block's MlpBlock Dense_1 bias gradient was fused into the producer of dx2;
    next_b2: gradient buffer of the PREVIOUS block's Dense_1 bias, to be fused into the
    LayerNorm_0 backward that produces that block's dx2 (bias grads = column sums of dx)."""
    if len(saved) == 14:   # saved by _fwd_drop (b2_done / next_b2 are never set for such contexts, see Encoder.bwd)
      assert not b2_done and next_b2 is None
      return self._bwd_drop(saved, dx2, n, L, kv_len)
    x, mean0, rstd0, y0, qkv, o, lse, x1, mean1, rstd1, y1, h, g = saved
    T, D, H = n * L, self.D, self.H
    if x.dtype == BF16:   # bf16 residual stream: the gradient stream IS the GEMM operand
      dx2 = dx2_bf
    # Light contexts keep neither LayerNorm output.  On the fp32 stream the LayerNorm BACKWARD kernel re-emits
    # it (bv_layernorm_bwd_y: it reads x anyway; same expression, same bits as the forward) and the weight-
    # gradient GEMM that needs it runs right after; on the bf16 stream a forward pass re-normalises first.
    emit_y = x.dtype == F32
    if y1 is None and not emit_y:
      y1 = self.ln1.fwd(x1, T, D)[0]
    dh = None
    if y1 is None:
      dy1, dh = self.mlp.bwd(dx2, dx2_bf, None, h, g, bias2_done=b2_done)
      y1 = torch.empty((T, D), device=dx2.device, dtype=BF16)
      y1_out = y1
    else:
      dy1 = self.mlp.bwd(dx2, dx2_bf, y1, h, g, bias2_done=b2_done)
      y1_out = None
    dx1_bf = torch.empty((T, D), device=dx2.device, dtype=BF16)
    dx1 = self.ln1.bwd(dy1, x1, mean1, rstd1, T, D, dres=dx2, dx_bf16=dx1_bf, dx_colsum=self.bo.grad, y_out=y1_out)
    if dh is not None:
      linear_bwd_w(y1, dh, self.mlp.w1, None)
    del y1, dh
    linear_bwd_w(o, dx1_bf, self.wo, None)      # out-proj bias grad = colsum(dx1): fused above
    d_o = linear_bwd_x(dx1_bf, self.wo)
    dqkv = ops.attn_bwd(qkv, o, d_o, lse, n, L, H, dbias=self.bqkv.grad, kv_len=kv_len)   # q/k/v bias grads fused
    if y0 is None and not emit_y:
      y0 = self.ln0.fwd(x, T, D)[0]
    y0_out = None
    if y0 is None:
      y0_out = torch.empty((T, D), device=dx2.device, dtype=BF16)
    else:
      linear_bwd_w(y0, dqkv, self.wqkv, None)
    dy0 = linear_bwd_x(dqkv, self.wqkv)
    dx_bf = torch.empty((T, D), device=dx2.device, dtype=BF16)
    dx = self.ln0.bwd(dy0, x, mean0, rstd0, T, D, dres=dx1, dx_bf16=dx_bf, dx_colsum=next_b2, y_out=y0_out)
    if y0_out is not None:
      linear_bwd_w(y0_out, dqkv, self.wqkv, None)
    return dx, dx_bf


class Encoder:
  """vit.Encoder without the final encoder_norm (the caller applies it, because
  which rows it must cover depends on the pooling)."""

  def __init__(self, store, prefix, depth, D, H, M, scan=False):
    self.blocks = [Block(store, f"{prefix}/encoderblock_{i}", D, H, M) for i in range(depth)]
    self.norm = LN(store, f"{prefix}/encoder_norm")
    self.D = D
    self.scan = bool(scan)   # presentation only: stacked leaf names and the `out` keys of the reference's scan branch

  def fwd(self, x, n, L, save, out=None, kv_len=None, drop=None):
    """x: fp32 [n*L, D].  Returns the last block's output in the stream dtype (the callers hand it to
    encoder_norm, whose kernel takes either) and the saved contexts.
    drop (Dropout, rate > 0; train mode of vit.py:100,109,76): every block gets its own fold of it; contexts are full
    ones whatever `save` asks for (the light kinds re-derive activations the mask would have to be re-applied to)."""
    saved = []
    dropping = drop is not None and drop.rate > 0.0
    if dropping and residual_stream() == BF16:
      raise NotImplementedError("dropout > 0 runs on the float32 residual stream only")
    if residual_stream() == BF16 and x.dtype == F32:
      x = ops.cast_bf16(x)
    for i, blk in enumerate(self.blocks):
      x_in = x
      x, s = blk.fwd(x, n, L, light=(True if save == "light" else ("g" if save == "g" else False)), kv_len=kv_len,
                     drop=(drop.fold("block", i) if dropping else None), collect=out is not None, ctx=bool(save))
      pre = None
      if dropping and out is not None:       # the branch outputs before their dropout leave the context again
        pre, s = s[13][3], s[:13] + (s[13][:3] + (None,),)
      if save:
        saved.append(s)
      if out is not None:
        x1 = s[7]
        v = lambda t: t.view(n, L, -1)   # the reference's activations are [n, L, D]
        sa, mlp = pre if pre is not None else (x1 - x_in, x - x1)     # (vit.py:98,108: published BEFORE the dropout)
        out[f"block{i:02d}"] = {"sa": v(sa), "+sa": v(x1), "mlp": v(mlp), "+mlp": v(x)}
    if out is not None and not self.scan:   # (the reference's scan branch publishes no `pre_ln` alias, vit.py:129-157)
      out["pre_ln"] = x.view(n, L, -1)
    return x, saved

  @staticmethod
  def dropped(saved):
    """True for contexts saved by a forward with dropout > 0 (Block._fwd_drop)."""
    return bool(saved) and len(saved[0]) == 14

  def last_b2_grad(self, saved=None):
    """Gradient buffer of the last block's Dense_1 bias: the kernel that produces the
    encoder's incoming dx (encoder_norm backward) accumulates its column sums there.  None for contexts saved under
    dropout: that bias takes the column sums of the MASKED gradient (Block._bwd_drop)."""
    if saved is not None and self.dropped(saved):
      return None
    return self.blocks[-1].mlp.b2.grad if self.blocks else None

  def bwd(self, saved, dx, dx_bf, n, L, b2_done=False, on_block=None, kv_len=None):
    """on_block(i): called after block i's backward is enqueued; the gradients of blocks >= i
    are final at that point (block i's Dense_1 bias was accumulated earlier, by the kernel
    that produced its incoming dx; block i's backward also finishes block i-1's Dense_1 bias)."""
    last = len(self.blocks) - 1
    bf_stream = bool(saved) and saved[0][0].dtype == BF16   # the contexts remember the stream they were built on
    dropped = self.dropped(saved)
    assert not (dropped and b2_done), "contexts saved under dropout: the caller must not fuse the Dense_1 bias gradient"
    for i in range(last, -1, -1):
      nb2 = self.blocks[i - 1].mlp.b2.grad if (i > 0 and not dropped) else None
      dx, dx_bf = self.blocks[i].bwd(saved[i], dx, dx_bf, n, L,
                                     b2_done=(False if dropped else (b2_done if i == last else True)),
                                     next_b2=nb2, kv_len=kv_len)
      if on_block is not None:
        on_block(i)
    if bf_stream:   # callers (stem / embedding / posemb gradients) take the fp32 tensor and its bf16 copy
      dx = ops.cast_f32(dx_bf)
    return dx, dx_bf


# ------------------------------------------------------------- MAP head ------
class MAPHead:
  """MAPHead (vit.py:163-183) on top of the encoder_norm output."""

  def __init__(self, store, prefix, D, H, M):
    A = f"{prefix}/MultiHeadDotProductAttention_0"
    self.D, self.H = D, H
    self.probe = _W(store, f"{prefix}/probe", (1, D))
    self.wq = _W(store, f"{A}/query/kernel", (D, D)); self.bq = _W(store, f"{A}/query/bias", (D,))
    self.wkv = _W(store, f"{A}/kv/kernel", (D, 2 * D)); self.bkv = _W(store, f"{A}/kv/bias", (2 * D,))
    self.wo = _W(store, f"{A}/out/kernel", (D, D)); self.bo = _W(store, f"{A}/out/bias")
    self.ln = LN(store, f"{prefix}/LayerNorm_0")
    self.mlp = MLP(store, f"{prefix}/MlpBlock_0", D, M)

  def fwd(self, y_bf, n, L, kv_len=None):
    D, H = self.D, self.H
    probe_t = self.probe.bf.expand(n, D).contiguous()
    q = linear_fwd(probe_t, self.wq, self.bq, out_dtype=BF16)
    kv = linear_fwd(y_bf, self.wkv, self.bkv, out_dtype=BF16)
    o, p = ops.map_attn_fwd(q, kv, n, L, H, kv_len=kv_len)
    a = linear_fwd(o, self.wo, self.bo, out_dtype=F32)
    yl, _, mean, rstd = self.ln.fwd(a, n, D)
    z, h, g = self.mlp.fwd(yl, a)
    return z, (y_bf, probe_t, q, kv, o, p, a, mean, rstd, yl, h, g)

  def bwd(self, saved, dz, n, L):
    y_bf, probe_t, q, kv, o, p, a, mean, rstd, yl, h, g = saved
    D, H = self.D, self.H
    dz_bf = ops.cast_bf16(dz)
    dyl = self.mlp.bwd(dz, dz_bf, yl, h, g)
    da_bf = torch.empty((n, D), device=dz.device, dtype=BF16)
    da = self.ln.bwd(dyl, a, mean, rstd, n, D, dres=dz, dx_bf16=da_bf, dx_colsum=self.bo.grad)
    linear_bwd_w(o, da_bf, self.wo, None)       # out-proj bias grad fused into the LN backward
    d_o = linear_bwd_x(da_bf, self.wo)
    dq, dkv = ops.map_attn_bwd(q, kv, p, d_o, n, L, H)
    linear_bwd_w(probe_t, dq, self.wq, self.bq)
    if self.probe.grad is not None:
      dprobe_t = linear_bwd_x(dq, self.wq)
      ops.colsum(dprobe_t, self.probe.grad.view(-1))
    linear_bwd_w(y_bf, dkv, self.wkv, self.bkv)
    return linear_bwd_x(dkv, self.wkv)   # dy (bf16 [T, D])
