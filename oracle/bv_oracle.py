"""CPU oracle: a torch-CPU (fp32 / fp64) restatement of big_vision's SigLIP/ViT step.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.
The product (`big_vision_amd/`) never imports anything from `oracle/`.

PARITY STATUS: **parity unpinned against the JAX reference itself** — jax/flax/
optax are not installed in this image and the reference ships no golden vectors
or tests for models/vit.py, two_towers.py, text_transformer.py or the sigmoid
loss.  What pins this oracle instead:
  * `oracle/make_golden.py` cross-checks the forward pass and the loss against
    HuggingFace `transformers.SiglipModel` (an independent PyTorch port of the
    same big_vision model) with weights copied across; the resulting fixtures
    live in `tests/golden/` and are re-checked by `tests/test_oracle.py`.
  * the reference's own known-answer tests for the pieces that HAVE them are
    restated verbatim in `tests/test_oracle.py`: LR schedules
    (big_vision/utils_test.py:258-281), `steps()` (utils_test.py:228-255) and
    the optimizer chain closed forms (big_vision/optax_test.py:103-299).

Every function cites the reference file:line it follows (paths relative to
/root/reference/big_vision/).  Parameters are nested dicts of torch tensors
with the exact Flax names/shapes of the reference.
"""
from __future__ import annotations

import math
import os
import re
from typing import Any, Dict, Optional, Sequence, Tuple

import numpy as np
import torch

Tree = Dict[str, Any]

# -----------------------------------------------------------------------------
# Variant table — models/vit.py:284-303 (decode_variant)
# -----------------------------------------------------------------------------
_WIDTH = {"mu": 32, "Ti": 192, "S": 384, "M": 512, "B": 768, "L": 1024, "So400m": 1152, "H": 1280, "g": 1408, "g-opt": 1536, "G": 1664, "G-opt": 1536, "e": 1792}
_DEPTH = {"mu": 1, "Ti": 12, "S": 12, "M": 12, "B": 12, "L": 24, "So400m": 27, "H": 32, "g": 40, "g-opt": 40, "G": 48, "G-opt": 48, "e": 56}
_MLP = {"mu": 128, "Ti": 768, "S": 1536, "M": 2048, "B": 3072, "L": 4096, "So400m": 4304, "H": 5120, "g": 6144, "g-opt": 6144, "G": 8192, "G-opt": 8192, "e": 15360}
_HEADS = {"mu": 2, "Ti": 3, "S": 6, "M": 8, "B": 12, "L": 16, "So400m": 16, "H": 16, "g": 16, "g-opt": 16, "G": 16, "G-opt": 16, "e": 16}


def decode_variant(variant: Optional[str]) -> dict:
  """models/vit.py:284-303."""
  if variant is None:
    return {}
  v, patch = variant, {}
  if "/" in variant:
    v, p = variant.split("/")
    patch = {"patch_size": (int(p), int(p))}
  return {"width": _WIDTH[v], "depth": _DEPTH[v], "mlp_dim": _MLP[v],
          "num_heads": _HEADS[v], **patch}


# -----------------------------------------------------------------------------
# Primitive layers (Flax semantics, SURVEY.md §8c "Flax facts")
# -----------------------------------------------------------------------------
def layernorm(x, p, eps=1e-6):
  """flax.linen.LayerNorm as used at models/vit.py:92,103,160,181.

  eps=1e-6, stats over the last axis, var = E[x^2] - E[x]^2 clamped at 0.
  """
  if _FUSED_BACKWARD and torch.is_grad_enabled() and x.dtype == torch.float64:
    return _LayerNormFn.apply(x, p["scale"], p["bias"], eps)
  return _layernorm_formula(x, p["scale"], p["bias"], eps)


def _layernorm_formula(x, scale, bias, eps):
  mu = x.mean(-1, keepdim=True)
  var = ((x * x).mean(-1, keepdim=True) - mu * mu).clamp_min(0.0)
  return (x - mu) * torch.rsqrt(var + eps) * scale + bias


# --- hand-written backward passes of the two element-wise layers that dominate the oracle's host time ------------
# The FORWARD values are the formulas above, evaluated as written.  What these Functions replace is autograd's tape
# for them (8-10 saved [n, L, 4D] / [n, L, D] temporaries and ~20 element-wise passes per layer in float64: 60 % of
# the oracle's time on the B/16 and L/16 cases of the GPU suite) by the closed-form derivative of the same formula.
# BV_ORACLE_AUTOGRAD=1 switches back to plain autograd; tests/test_oracle.py holds the two to 1e-12 of each other
# and tests/test_reference_gradients_cpu.py holds the result to finite differences of the EXECUTED reference.
_FUSED_BACKWARD = os.environ.get("BV_ORACLE_AUTOGRAD") != "1"


class _LayerNormFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x, scale, bias, eps):
    mu = x.mean(-1, keepdim=True)
    var_raw = (x * x).mean(-1, keepdim=True) - mu * mu
    rstd = torch.rsqrt(var_raw.clamp_min(0.0) + eps)
    ctx.save_for_backward(x, scale, mu, rstd, var_raw >= 0.0)
    return (x - mu) * rstd * scale + bias

  @staticmethod
  def backward(ctx, g):
    x, scale, mu, rstd, live = ctx.saved_tensors
    xhat = (x - mu).mul_(rstd)
    red = tuple(range(g.dim() - scale.dim()))
    dscale = (g * xhat).sum(red) if ctx.needs_input_grad[1] else None
    dbias = g.sum(red) if ctx.needs_input_grad[2] else None
    dx = None
    if ctx.needs_input_grad[0]:
      gy = g * scale
      # y = (x - mu) rstd(var), var = E[x^2] - mu^2 (its derivative vanishes where the clamp is active)
      c = (gy * xhat).mean(-1, keepdim=True).mul_(live)
      dx = gy.sub_(gy.mean(-1, keepdim=True)).sub_(xhat.mul_(c)).mul_(rstd)
    return dx, dscale, dbias, None


# --- optional emulation of the product's arithmetic (NOT the reference's) ---------------------
# `with bf16_operands():` rounds BOTH operands of every tower contraction (and the incoming
# cotangent in the backward) to bfloat16 and accumulates in the working dtype: the arithmetic the
# north star prescribes for the MFMA GEMMs / attention (bf16 operands, fp32 accumulate).  The
# tests use it to MEASURE the noise floor of that arithmetic per gradient tensor (oracle-bf16 vs
# oracle-fp64), so a stated tolerance above SURVEY.md §8c's proposal carries a measured reason.
# Outside the context manager nothing changes (plain torch matmul / einsum).
_BF16_OPERANDS = False


class bf16_operands:
  def __enter__(self):
    global _BF16_OPERANDS
    self.prev, _BF16_OPERANDS = _BF16_OPERANDS, True

  def __exit__(self, *exc):
    global _BF16_OPERANDS
    _BF16_OPERANDS = self.prev


def _rb(t):
  return t.to(torch.bfloat16).to(t.dtype)


# `with bf16_residual():` additionally rounds the RESIDUAL STREAM to bfloat16 after every residual add
# (forward value and backward cotangent): the arithmetic a bf16 residual stream would have.  Only used by
# tools/bf16_residual_budget.py to measure what that design option costs against the parity bounds.
_BF16_RESIDUAL = False


class bf16_residual:
  def __enter__(self):
    global _BF16_RESIDUAL
    self.prev, _BF16_RESIDUAL = _BF16_RESIDUAL, True

  def __exit__(self, *exc):
    global _BF16_RESIDUAL
    _BF16_RESIDUAL = self.prev


class _RoundStream(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x):
    return _rb(x)

  @staticmethod
  def backward(ctx, g):
    return _rb(g)


def _stream(x):
  return _RoundStream.apply(x) if _BF16_RESIDUAL else x


class _Bf16Contract(torch.autograd.Function):
  """einsum(eq, a, b) with operands (forward) and cotangent (backward) rounded to bf16.  Every
  index of an operand appears in the output or in the other operand (true for all contractions of
  the model), so the two gradient contractions are einsums over the permuted equation."""

  @staticmethod
  def forward(ctx, eq, a, b):
    ra, rb = _rb(a), _rb(b)
    ctx.save_for_backward(ra, rb)
    ctx.eq = eq
    return torch.einsum(eq, ra, rb)

  @staticmethod
  def backward(ctx, g):
    ra, rb = ctx.saved_tensors
    lhs, out = ctx.eq.split("->")
    ia, ib = lhs.split(",")
    rg = _rb(g)
    return None, torch.einsum(f"{out},{ib}->{ia}", rg, rb), torch.einsum(f"{ia},{out}->{ib}", ra, rg)


def contract(eq, a, b):
  if _BF16_OPERANDS:
    return _Bf16Contract.apply(eq, a, b)
  return torch.einsum(eq, a, b)


def dense(x, p):
  """flax.linen.Dense: y = x @ kernel + bias (models/vit.py:72,77)."""
  if _BF16_OPERANDS:
    return _Bf16Contract.apply("...d,df->...f", x, p["kernel"]) + p["bias"]
  return x @ p["kernel"] + p["bias"]


def gelu_tanh(x):
  """flax.linen.gelu default approximate=True (models/vit.py:75)."""
  if _FUSED_BACKWARD and torch.is_grad_enabled() and x.requires_grad and x.dtype == torch.float64:
    return _GeluTanhFn.apply(x)
  return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


class _GeluTanhFn(torch.autograd.Function):
  """gelu_tanh with the closed-form derivative  0.5 (1 + t) + 0.5 x (1 - t^2) c (1 + 3 a x^2),  t = tanh(c (x + a x^3))."""

  @staticmethod
  def forward(ctx, x):
    ctx.save_for_backward(x)
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))

  @staticmethod
  def backward(ctx, g):
    x, = ctx.saved_tensors
    c, a = math.sqrt(2.0 / math.pi), 0.044715
    x2 = x * x
    t = torch.tanh_((x2 * a).add_(1.0).mul_(x).mul_(c))          # tanh(c (x + a x^3))
    du = x2.mul_(3.0 * a).add_(1.0).mul_(c)                       # d/dx of the tanh argument
    d = (t * t).neg_().add_(1.0).mul_(du).mul_(x).add_(t).add_(1.0).mul_(0.5)
    return d.mul_(g)


def mha(xq, xkv, p, num_heads, mask=None):
  """flax.linen.MultiHeadDotProductAttention (models/vit.py:93-98, :176-178).

  q/k/v kernels (D,H,Dh) + bias (H,Dh); out kernel (H,Dh,D) + bias (D).
  query scaled by 1/sqrt(Dh) before the dot; softmax over keys.  mask (bool, broadcastable to
  [n, heads, q, k], flax.linen.attention.dot_product_attention_weights): masked logits are replaced
  by the most negative finite value, so a fully masked row attends uniformly (naflex_vit.py:96-104).
  """
  q = contract("nld,dhk->nlhk", xq, p["query"]["kernel"]) + p["query"]["bias"]
  k = contract("nld,dhk->nlhk", xkv, p["key"]["kernel"]) + p["key"]["bias"]
  v = contract("nld,dhk->nlhk", xkv, p["value"]["kernel"]) + p["value"]["bias"]
  dh = q.shape[-1]
  if _BF16_OPERANDS:   # the kernels keep q unscaled in bf16 and fold 1/sqrt(dh) into the exponent
    s = contract("nqhd,nkhd->nhqk", q, k) / math.sqrt(dh)
  else:
    q = q / math.sqrt(dh)
    s = contract("nqhd,nkhd->nhqk", q, k)
  if mask is not None:
    s = torch.where(mask, s, torch.full_like(s, torch.finfo(s.dtype).min))
  a = torch.softmax(s, dim=-1)
  o = contract("nhqk,nkhd->nqhd", a, v)
  return contract("nlhk,hkd->nld", o, p["out"]["kernel"]) + p["out"]["bias"]


class DropMasks:
  """Dropout in train mode (flax nn.Dropout(rate)(x, deterministic=False): x * keep / (1 - rate), keep ~
  Bernoulli(1 - rate)) with the keep masks GIVEN: JAX's random stream cannot be reproduced, so the parity tests draw
  the masks once (from the product's own generator, ops.dropout_mask) and hand the same bits to both sides.
  `masks`: {site name: bool tensor of the activation's shape}; site names are `<prefix>posemb` (vit.py:228) and
  `<prefix>block<i>/{sa, gelu, mlp}` (vit.py:100, :76, :109).  `used` records the sites that were applied."""

  def __init__(self, rate, masks, prefix=""):
    self.rate, self.masks, self.prefix, self.used = float(rate), masks, prefix, []

  def sub(self, name):
    d = DropMasks(self.rate, self.masks, self.prefix + name)
    d.used = self.used
    return d

  def __call__(self, site, x):
    name = self.prefix + site
    m = self.masks[name]
    self.used.append(name)
    return x * (m.reshape(x.shape).to(x.dtype) / (1.0 - self.rate))


def philox4x32_10(counter, key):
  """Philox-4x32 with 10 rounds (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11;
  Random123), vectorised over a leading axis: counter uint32 [..., 4], key uint32 [..., 2] -> uint32 [..., 4].
  The product's dropout kernels (csrc/dropout.hip) draw their keep bits from it; pinned by Random123's published
  known-answer vectors in tests/test_dropout_cpu.py."""
  import numpy as np
  c = [np.asarray(counter[..., i], np.uint64) for i in range(4)]
  k0, k1 = np.asarray(key[..., 0], np.uint64), np.asarray(key[..., 1], np.uint64)
  M0, M1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
  for _ in range(10):
    p0, p1 = M0 * c[0], M1 * c[2]
    c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & MASK, p1 & MASK, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & MASK, p0 & MASK]
    k0, k1 = (k0 + np.uint64(0x9E3779B9)) & MASK, (k1 + np.uint64(0xBB67AE85)) & MASK
  return np.stack(c, -1).astype(np.uint32)


def dropout_keep_mask(key, count, rate):
  """The keep bits bv_dropout_* derive from a 64-bit site key (include/bvhip.h): element 4 g + j keeps iff word j of
  philox4x32_10(counter = (g mod 2^32, g >> 32, 0, 0), key = (key mod 2^32, key >> 32)) < floor((1 - rate) 2^32)."""
  import numpy as np
  assert count % 4 == 0
  g = np.arange(count // 4, dtype=np.uint64)
  ctr = np.stack([g & np.uint64(0xFFFFFFFF), g >> np.uint64(32), 0 * g, 0 * g], -1).astype(np.uint32)
  k = np.broadcast_to(np.array([key & 0xFFFFFFFF, (key >> 32) & 0xFFFFFFFF], np.uint32), (len(g), 2))
  thr = min(int((1.0 - float(np.float32(rate))) * 4294967296.0), 4294967296)
  return (philox4x32_10(ctr, k).astype(np.uint64) < np.uint64(thr)).reshape(-1)


def _drop(drop, site, x):
  return x if drop is None else drop(site, x)


def mlp_block(x, p, drop=None):
  """models/vit.py:57-78 (MlpBlock); drop: the dropout behind the GELU (:76)."""
  return dense(_drop(drop, "gelu", gelu_tanh(dense(x, p["Dense_0"]))), p["Dense_1"])


def encoder_block(x, p, num_heads, mask=None, drop=None):
  """models/vit.py:81-112 (Encoder1DBlock); mask [n, q, k] as naflex_vit.py:92-94; drop (DropMasks of this block,
  train mode with dropout > 0): the branch dropouts of :100 and :109 and the MlpBlock's."""
  out = {}
  y = layernorm(x, p["LayerNorm_0"])
  y = out["sa"] = mha(y, y, p["MultiHeadDotProductAttention_0"], num_heads,
                      mask=None if mask is None else mask[:, None])
  y = _drop(drop, "sa", y)
  x = out["+sa"] = _stream(x + y)
  y = layernorm(x, p["LayerNorm_1"])
  y = out["mlp"] = mlp_block(y, p["MlpBlock_0"], drop)
  y = _drop(drop, "mlp", y)
  x = out["+mlp"] = _stream(x + y)
  return x, out


def encoder(x, p, depth, num_heads, mask=None, drop=None):
  """models/vit.py:115-160 (Encoder); accepts loop and scan param layouts."""
  out = {}
  x = _stream(x)
  sub = lambda lyr: None if drop is None else drop.sub(f"block{lyr}/")
  if "encoderblock" in p:  # scan layout: leading depth axis (vit.py:129-148)
    for lyr in range(depth):
      pl = tree_map(lambda t, l=lyr: t[l], p["encoderblock"])
      x, out[f"block{lyr:02d}"] = encoder_block(x, pl, num_heads, mask, sub(lyr))
  else:
    for lyr in range(depth):
      x, out[f"block{lyr:02d}"] = encoder_block(x, p[f"encoderblock_{lyr}"], num_heads, mask, sub(lyr))
    out["pre_ln"] = x
  return layernorm(x, p["encoder_norm"]), out


def map_head(x, p, num_heads, mask=None):
  """models/vit.py:163-183 (MAPHead); mask [n, k] = pool mask of naflex_vit.py:183-199."""
  n = x.shape[0]
  probe = p["probe"].expand(n, -1, -1)
  x = mha(probe, x, p["MultiHeadDotProductAttention_0"], num_heads,
          mask=None if mask is None else mask[:, None, None, :])
  y = layernorm(x, p["LayerNorm_0"])
  x = x + mlp_block(y, p["MlpBlock_0"])
  return x[:, 0]


def scale_and_translate_weights(input_size, output_size, scale, translation=0.0, antialias=True,
                                dtype=torch.float64):
  """jax._src.image.scale.compute_weight_mat for the triangle ("bilinear") kernel: [input_size,
  output_size] (jax is an un-vendored dependency of the reference; this restates its published
  algorithm, pinned in tests against torch's antialiased bilinear interpolate)."""
  inv_scale = 1.0 / scale
  kernel_scale = max(inv_scale, 1.0) if antialias else 1.0
  sample_f = (torch.arange(output_size, dtype=dtype) + 0.5) * inv_scale - translation * inv_scale - 0.5
  x = (sample_f[None, :] - torch.arange(input_size, dtype=dtype)[:, None]).abs() / kernel_scale
  w = torch.clamp(1.0 - x, min=0.0)
  total = w.sum(0, keepdim=True)
  w = torch.where(total.abs() > 1000.0 * 1.1920929e-07, w / torch.where(total != 0, total, torch.ones_like(total)),
                  torch.zeros_like(w))
  ok = (sample_f >= -0.5) & (sample_f <= input_size - 0.5)
  return torch.where(ok[None, :], w, torch.zeros_like(w))


def naflex_pos_emb_resize(pos_emb, yabs, xabs, l=64):
  """models/proj/image_text/naflex_vit.py:38-83: per example resize the [P, P, D] grid to its patch
  grid (coords.max + 1) inside an l x l canvas and gather at (yabs, xabs)."""
  P = pos_emb.shape[0]
  outs = []
  for e in range(yabs.shape[0]):
    h, w = int(yabs[e].max()) + 1, int(xabs[e].max()) + 1
    wy = scale_and_translate_weights(P, l, h / P, dtype=pos_emb.dtype)
    wx = scale_and_translate_weights(P, l, w / P, dtype=pos_emb.dtype)
    emb = torch.einsum("iy,jx,ijd->yxd", wy, wx, pos_emb)
    outs.append(emb[yabs[e].long(), xabs[e].long()])
  return torch.stack(outs)


def naflex_vit_forward(params, image, *, num_classes=None, width=768, depth=12, mlp_dim=None, num_heads=12,
                       rep_size=False, pool_type="gap", posemb="learn_2d(64)", nposemb=None, patchln_pre=False,
                       patchln_post=False, variant=None, **unused):
  """models/proj/image_text/naflex_vit.py:200-285 (_Model.__call__), dropout 0."""
  del unused, mlp_dim, nposemb
  if variant is not None:
    dv = decode_variant(variant)
    width, depth, num_heads = dv["width"], dv["depth"], dv["num_heads"]
  patches, ptype, yabs, xabs = image
  out = {}
  x = patches
  if patchln_pre:
    x = layernorm(x, params["patchln_pre"])
  tokens = out["stem"] = dense(x, params["embedding"])
  if patchln_post:
    tokens = layernorm(tokens, params["patchln_post"])
  grid = int(posemb[len("learn_2d("):-1]) if posemb.startswith("learn_2d(") else 64
  x = out["with_posemb"] = tokens + naflex_pos_emb_resize(params["pos_embedding"], yabs, xabs, grid)
  valid = ptype == 1
  sa_mask = valid[:, :, None] & valid[:, None, :]
  x, out["encoder"] = encoder(x, params["Transformer"], depth, num_heads, mask=sa_mask)
  out["encoded"] = x
  if pool_type == "map":
    x = map_head(x, params["MAPHead_0"], num_heads, mask=valid)
  elif pool_type == "gap":
    pm = valid[..., None].to(x.dtype)
    x = (x * pm).sum(1) / pm.sum(1)
  elif pool_type == "max":
    pm = valid[..., None]
    x = torch.where(pm, x, torch.full_like(x, torch.finfo(x.dtype).min)).max(dim=1).values
  elif pool_type != "none":
    raise ValueError(pool_type)
  out["head_input"] = x
  if rep_size:
    x = torch.tanh(dense(x, params["pre_logits"]))
  out["pre_logits"] = x
  if num_classes:
    x = out["logits"] = dense(x, params["head"])
  return x, out


def posemb_sincos_2d(h, w, width, temperature=10_000.0, dtype=torch.float32):
  """models/vit.py:34-44."""
  y, x = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
  assert width % 4 == 0
  omega = torch.arange(width // 4, dtype=torch.float64) / (width // 4 - 1)
  omega = 1.0 / (temperature ** omega)
  y = y.flatten().to(torch.float64)[:, None] * omega[None, :]
  x = x.flatten().to(torch.float64)[:, None] * omega[None, :]
  pe = torch.cat([torch.sin(x), torch.cos(x), torch.sin(y), torch.cos(y)], dim=1)
  return pe.to(dtype)[None]


def extract_patches(image, patch):
  """NHWC image -> [n, h*w, ph*pw*3] in HWIO flattening order (row, col, chan).

  Equivalent to the strided VALID conv of models/vit.py:212-217 (also
  models/proj/flexi/vit_test.py:40-42: NHWC x HWIO).
  """
  n, H, W, C = image.shape
  ph, pw = patch
  h, w = H // ph, W // pw
  x = image[:, :h * ph, :w * pw, :].reshape(n, h, ph, w, pw, C)
  x = x.permute(0, 1, 3, 2, 4, 5).reshape(n, h * w, ph * pw * C)
  return x, (h, w)


# -----------------------------------------------------------------------------
# Models
# -----------------------------------------------------------------------------
def vit_forward(params, image, *, num_classes=None, patch_size=(16, 16), width=768,
                depth=12, mlp_dim=None, num_heads=12, posemb="learn",
                rep_size=False, pool_type="gap", drop=None, **_unused):
  """models/vit.py:206-276 (_Model.__call__); drop (DropMasks): train mode with dropout > 0."""
  out = {}
  patches, (h, w) = extract_patches(image, patch_size)
  kern = params["embedding"]["kernel"]
  if _BF16_OPERANDS:
    x = _Bf16Contract.apply("nlp,pd->nld", patches, kern.reshape(-1, kern.shape[-1])) + params["embedding"]["bias"]
  else:
    x = patches @ kern.reshape(-1, kern.shape[-1]) + params["embedding"]["bias"]
  out["stem"] = x.reshape(x.shape[0], h, w, -1)
  if posemb == "learn":
    pe = params["pos_embedding"]
  elif posemb == "sincos2d":
    pe = posemb_sincos_2d(h, w, width, dtype=x.dtype)
  else:
    raise ValueError(f"Unknown posemb type: {posemb}")
  x = out["with_posemb"] = x + pe
  n = x.shape[0]
  if pool_type == "tok":
    x = torch.cat([params["cls"].expand(n, -1, -1), x], dim=1)
  x = _drop(drop, "posemb", x)    # vit.py:228
  x, out["encoder"] = encoder(x, params["Transformer"], depth, num_heads, drop=drop)
  encoded = out["encoded"] = x
  if pool_type == "map":
    x = out["head_input"] = map_head(x, params["MAPHead_0"], num_heads)
  elif pool_type == "gap":
    x = out["head_input"] = x.mean(dim=1)
  elif pool_type == "0":
    x = out["head_input"] = x[:, 0]
  elif pool_type == "tok":
    x = out["head_input"] = x[:, 0]
    encoded = encoded[:, 1:]
  elif pool_type == "none":
    pass
  else:
    raise ValueError(f"Unknown pool type: '{pool_type}'")
  x_2d = encoded.reshape(n, h, w, -1)
  if rep_size:
    x_2d = torch.tanh(dense(x_2d, params["pre_logits"]))
    x = torch.tanh(dense(x, params["pre_logits"]))
  out["pre_logits_2d"] = x_2d
  out["pre_logits"] = x
  if num_classes:
    x_2d = out["logits_2d"] = dense(x_2d, params["head"])
    x = out["logits"] = dense(x, params["head"])
  return x, out


def text_forward(params, text, *, num_classes, width=512, depth=12, mlp_dim=2048,
                 num_heads=8, vocab_size=32_000, pool_type="last", drop=None, **_unused):
  """models/proj/image_text/text_transformer.py:55-99; drop (DropMasks): train mode with dropout > 0 (:72-75)."""
  out = {}
  emb = params["Embed_0"]["embedding"]
  x = out["embedded"] = emb[text]
  x = x + params["pos_embedding"]
  x, enc_out = encoder(x, params["Encoder_0"], depth, num_heads, drop=drop)
  out.update({"transformed": x, **enc_out})
  out["vocab_logits"] = x @ emb.T
  if pool_type == "last":
    x = out["pre_logits"] = x[:, -1, :]
  elif pool_type == "first":
    x = out["pre_logits"] = x[:, 0, :]
  elif pool_type in ("mean", "gap"):
    x = out["pre_logits"] = x.mean(dim=1)
  elif pool_type in ("max", "gmp"):
    x = out["pre_logits"] = x.max(dim=1).values
  elif pool_type == "map":
    x = out["pre_logits"] = map_head(x, params["MAPHead_0"], num_heads)
  else:
    raise NotImplementedError(f"Cannot do pooling '{pool_type}'")
  if num_classes:
    x = out["logits"] = dense(x, params["head"])
  return x, out


# -----------------------------------------------------------------------------
# BERT text tower - models/proj/flaxformer/bert.py:33-64 on flaxformer's BertEncoder
# -----------------------------------------------------------------------------
# PARITY UNPINNED against the reference's own arithmetic: the encoder lives in the un-vendored, un-pinned
# `flaxformer` package (flaxformer/architectures/bert/bert.py; big_vision/requirements.txt lists
# `flaxformer` without a version).  This restates its published algorithm = the original BERT encoder
# (Devlin et al. 2018, google-research/bert modeling.py): token + position + segment embeddings ->
# LayerNorm(eps 1e-12) -> num_hidden_layers POST-LayerNorm blocks {self-attention (+bias, key mask from
# input_mask) -> +residual -> LayerNorm -> Dense(intermediate) -> gelu (tanh approximation: flax nn.gelu
# default, the original TF BERT's formula) -> Dense -> +residual -> LayerNorm}.  bert.py:45-56 fixes the
# call: position_ids = arange(max_len), segment_ids = 0, input_mask = (text != 0).  What pins it here is
# HuggingFace `BertModel(hidden_act="gelu_new")`, an independent implementation of the same published
# network (tests/test_oracle.py, tests/golden/bert_hf_tiny.npz, generator oracle/make_golden_bert.py).
BERT_CONFIGS = {   # flaxformer/architectures/bert/configs.py: BertBaseConfig / BertLargeConfig
    "base": dict(hidden_size=768, intermediate_dim=3072, num_hidden_layers=12, num_attention_heads=12,
                 vocab_size=30522, max_length=512, num_segments=2),
    "large": dict(hidden_size=1024, intermediate_dim=4096, num_hidden_layers=24, num_attention_heads=16,
                  vocab_size=30522, max_length=512, num_segments=2),
}
BERT_LN_EPS = 1e-12


def bert_config(config):
  """`config`: "base" / "large" (bert.py:46-49) or an explicit dict of the same fields (tests)."""
  return dict(BERT_CONFIGS[config]) if isinstance(config, str) else dict(config)


def bert_encoder(p, text, cfg):
  """flaxformer BertEncoder.__call__(token_ids, position_ids, segment_ids, input_mask) as bert.py:50-57
  calls it.  Masked logits get the large negative bias flax's attention uses (a fully masked QUERY row -
  a padded position - attends uniformly; those rows never reach the CLS output)."""
  n, L = text.shape
  H = cfg["num_attention_heads"]
  emb = p["embedder"]
  x = (emb["embedders_token_ids"]["embedding"][text]
       + emb["embedders_position_ids"]["embedding"][:L][None]
       + emb["embedders_segment_ids"]["embedding"][0][None, None])
  x = layernorm(x, p["layer_norm"], eps=BERT_LN_EPS)
  valid = text != 0
  mask = (valid[:, :, None] & valid[:, None, :])[:, None]          # make_attention_mask(input_mask, input_mask)
  for i in range(cfg["num_hidden_layers"]):
    b = p[f"encoder_block_{i}"]
    a = mha(x, x, b["attention_block"]["attention_layer"], H, mask=mask)
    x = layernorm(_stream(x + a), b["attention_block"]["layer_norm"], eps=BERT_LN_EPS)
    m = dense(gelu_tanh(dense(x, b["mlp_block"]["mlp"]["wi"])), b["mlp_block"]["mlp"]["wo"])
    x = layernorm(_stream(x + m), b["mlp_block"]["layer_norm"], eps=BERT_LN_EPS)
  return x


def bert_forward(params, text, *, config, num_classes=None, **_unused):
  """models/proj/flaxformer/bert.py:40-64: encoder -> CLS token -> optional `head` Dense."""
  cfg = bert_config(config)
  out = {}
  x = out["transformed"] = bert_encoder(params["BertEncoder_0"], text.long(), cfg)
  x = out["pre_logits"] = x[:, 0]
  if num_classes:
    x = out["logits"] = dense(x, params["head"])
  return x, out


def init_bert(gen, *, config, num_classes=None, head_zeroinit=True, dtype=torch.float32, **_):
  """Random init in the flaxformer tree layout (truncated-normal(0.02)-like scale; the reference always
  loads a checkpoint, the init only has to be a valid point)."""
  cfg = bert_config(config)
  D, M, H = cfg["hidden_size"], cfg["intermediate_dim"], cfg["num_attention_heads"]
  Dh = D // H
  rn = lambda *shape: torch.randn(shape, generator=gen, dtype=dtype) * 0.02
  ln = lambda: {"scale": torch.ones(D, dtype=dtype), "bias": torch.zeros(D, dtype=dtype)}
  enc = {"embedder": {"embedders_token_ids": {"embedding": rn(cfg["vocab_size"], D)},
                      "embedders_position_ids": {"embedding": rn(cfg["max_length"], D)},
                      "embedders_segment_ids": {"embedding": rn(cfg["num_segments"], D)}},
         "layer_norm": ln()}
  for i in range(cfg["num_hidden_layers"]):
    att = {k: {"kernel": rn(D, H, Dh), "bias": torch.zeros((H, Dh), dtype=dtype)} for k in ("query", "key", "value")}
    att["out"] = {"kernel": rn(H, Dh, D), "bias": torch.zeros(D, dtype=dtype)}
    enc[f"encoder_block_{i}"] = {
        "attention_block": {"attention_layer": att, "layer_norm": ln()},
        "mlp_block": {"mlp": {"wi": {"kernel": rn(D, M), "bias": torch.zeros(M, dtype=dtype)},
                              "wo": {"kernel": rn(M, D), "bias": torch.zeros(D, dtype=dtype)}},
                      "layer_norm": ln()}}
  p = {"BertEncoder_0": enc}
  if num_classes:
    p["head"] = {"kernel": (torch.zeros((D, num_classes), dtype=dtype) if head_zeroinit
                            else _lecun_normal(gen, (D, num_classes), D, dtype)),
                 "bias": torch.zeros(num_classes, dtype=dtype)}
  return p


def two_towers_forward(params, image, text, *, image_cfg, text_cfg, out_dim, text_model=None, image_model=None, drop=None,
                       **_unused):
  """models/proj/image_text/two_towers.py:39-90; drop: {"img": DropMasks, "txt": DropMasks} (either may be missing) for
  a train-mode pass of towers configured with dropout > 0 (:56, :69 hand `train` down to both towers)."""
  drop = drop or {}
  out = {}
  out_dims = (out_dim, out_dim) if isinstance(out_dim, int) else tuple(out_dim)
  zimg = ztxt = None
  if text is not None:
    if text_model == "proj.flaxformer.bert":            # two_towers.py:51-53: towers by module path
      ztxt, o = bert_forward(params["txt"], text, num_classes=out_dims[1], **text_cfg)
    else:
      kw = {**decode_variant(text_cfg.get("variant")),
            **{k: v for k, v in text_cfg.items() if k != "variant"}}
      ztxt, o = text_forward(params["txt"], text, num_classes=out_dims[1], **{**kw, "drop": drop.get("txt")})
    out.update({f"txt/{k}": v for k, v in o.items()})
    out["txt/norm"] = torch.linalg.norm(ztxt, dim=1, keepdim=True)
    out["txt/normalized"] = ztxt = ztxt / (out["txt/norm"] + 1e-8)
  if image is not None:
    kw = {**decode_variant(image_cfg.get("variant")),
          **{k: v for k, v in image_cfg.items() if k != "variant"}}
    if image_model == "proj.image_text.naflex_vit":     # two_towers.py:64-66: towers by module path
      zimg, o = naflex_vit_forward(params["img"], image, num_classes=out_dims[0], **kw)
    else:
      zimg, o = vit_forward(params["img"], image, num_classes=out_dims[0], **{**kw, "drop": drop.get("img")})
    out.update({f"img/{k}": v for k, v in o.items()})
    out["img/norm"] = torch.linalg.norm(zimg, dim=1, keepdim=True)
    out["img/normalized"] = zimg = zimg / (out["img/norm"] + 1e-8)
  out["t"] = torch.exp(params["t"])
  out["t/parameter"] = params["t"]
  if "b" in params:
    out["b"] = params["b"]
  return zimg, ztxt, out


# -----------------------------------------------------------------------------
# Losses
# -----------------------------------------------------------------------------
def log_sigmoid(x):
  """jax.nn.log_sigmoid = -softplus(-x), stable form."""
  return torch.clamp(x, max=0.0) - torch.log1p(torch.exp(-x.abs()))


def siglip_loss_global(zimg, ztxt, t, b):
  """trainers/proj/image_text/siglip.py:291-306 (global-batch sigmoid loss)."""
  logits = zimg @ ztxt.T * t + b
  eye = torch.eye(zimg.shape[0], dtype=logits.dtype)
  m1_diag1 = -torch.ones_like(logits) + 2 * eye
  loglik = log_sigmoid(m1_diag1 * logits)
  nll = -loglik.sum(dim=-1)
  return nll.mean(), logits


def sigmoid_loss_per_device(zimg_r, ztxt_shards, r, t, b=0.0):
  """trainers/proj/image_text/_deprecated_contrastive.py:117-141 on "device" r.

  `ztxt_shards` is the list of every device's local ztxt; device r keeps its
  own chunk as "me" and sees the others rolled so that r's chunk would come
  first and is dropped (all_gather(only_others=True), :67-77).
  """
  N = len(ztxt_shards)
  ztxt_me = ztxt_shards[r]
  others = [ztxt_shards[(r + k) % N] for k in range(1, N)]
  logits_me = zimg_r @ ztxt_me.T * t + b
  eye = torch.eye(zimg_r.shape[0], dtype=logits_me.dtype)
  m1_diag1 = -torch.ones_like(logits_me) + 2 * eye
  nll_me = -log_sigmoid(m1_diag1 * logits_me).sum(dim=-1)
  l = nll_me.mean()
  if others:
    ztxt_ot = torch.cat(others, 0)
    logits_ot = zimg_r @ ztxt_ot.T * t + b
    l = l + (-log_sigmoid(-logits_ot).sum(dim=-1)).mean()
  return l


def sigmoid_logit_stats_per_device(zimg_r, ztxt_shards, r, t, b=0.0):
  """The measurement dict of _deprecated_contrastive.py:143-160 on "device" r (before the pmean)."""
  N = len(ztxt_shards)
  logits_me = zimg_r @ ztxt_shards[r].T * t + b
  others = [ztxt_shards[(r + k) % N] for k in range(1, N)]
  logits_ot = zimg_r @ torch.cat(others, 0).T * t + b if others else logits_me.new_zeros((logits_me.shape[0], 0))
  n = logits_me.shape[0]
  eye = torch.eye(n, dtype=logits_me.dtype)
  diag = torch.diagonal(logits_me)

  def avg_neg(x_me, x_ot=None):      # :104-111
    nom = x_me.sum() - torch.diagonal(x_me).sum()
    den = x_me.numel() - len(x_me)
    if x_ot is not None and x_ot.numel():
      nom = nom + x_ot.sum()
      den += x_ot.numel()
    return nom / den

  inf = torch.tensor(float("inf"), dtype=logits_me.dtype)
  return {
      "pos_min_logit": diag.min(), "pos_max_logit": diag.max(), "pos_avg_logit": diag.mean(),
      "local_neg_min_logit": (logits_me + 1e9 * eye).min(),
      "local_neg_max_logit": (logits_me - 1e9 * eye).max(),
      "local_neg_avg_logit": avg_neg(logits_me),
      "neg_min_logit": torch.minimum((logits_me + 1e9 * eye).min(), logits_ot.min() if logits_ot.numel() else inf),
      "neg_max_logit": torch.maximum((logits_me - 1e9 * eye).max(), logits_ot.max() if logits_ot.numel() else -inf),
      "neg_avg_logit": avg_neg(logits_me, logits_ot),
  }


def chunked_sigmoid_loss_per_device(zimg_r, ztxt_shards, r, t, b=0.0):
  """_deprecated_contrastive.py:168-200 (one broadcast per other device)."""
  logits_me = zimg_r @ ztxt_shards[r].T * t + b
  eye = torch.eye(zimg_r.shape[0], dtype=logits_me.dtype)
  l = (-log_sigmoid((-torch.ones_like(logits_me) + 2 * eye) * logits_me).sum(-1)).mean()
  for d, z in enumerate(ztxt_shards):
    if d == r:
      continue
    logits_ot = zimg_r @ z.T * t + b
    l = l + (-log_sigmoid(-logits_ot).sum(-1)).mean()
  return l


def softmax_loss_per_device(zimg_r, ztxt_r, zimg_shards, ztxt_shards, r, t):
  """_deprecated_contrastive.py:80-101 (CLIP loss; before the pmean)."""
  N = len(ztxt_shards)

  def uni(z1, shards):
    z2 = torch.cat([shards[(r + k) % N] for k in range(N)], 0)
    logits = z1 @ z2.T * t
    return -(torch.diagonal(logits) - torch.logsumexp(logits, dim=-1)).mean()

  return 0.5 * uni(zimg_r, ztxt_shards) + 0.5 * uni(ztxt_r, zimg_shards)


def softmax_xent(logits, labels):
  """utils.py:276-281."""
  return (-(labels * torch.log_softmax(logits, dim=-1)).sum(-1)).mean()


def sigmoid_xent(logits, labels):
  """utils.py:236-243 (stable log-sigmoid form)."""
  log_p = log_sigmoid(logits)
  log_not_p = log_sigmoid(-logits)
  return (-(labels * log_p + (1.0 - labels) * log_not_p).sum(-1)).mean()


def classification_step_loss(params, image, labels, *, model_cfg, num_classes, loss="sigmoid_xent",
                             mixup_a=None):
  """Loss of the classification trainer (big_vision/train.py:281-300): optional mixup with a
  GIVEN coefficient a (utils.py:1146-1154; the reference draws it from jax.random.beta), then
  getattr(u, config.loss)(logits, labels) on the ViT logits.  Returns (loss, logits)."""
  if mixup_a is not None:
    image, labels = mixup(mixup_a, image, labels)
  logits, _ = vit_forward(params, image, num_classes=num_classes, **model_cfg)
  fn = {"sigmoid_xent": sigmoid_xent, "softmax_xent": softmax_xent}[loss]
  return fn(logits, labels), logits


def mixup(a, *things):
  """utils.py:1146-1154 with a given mixing coefficient a (already max(a,1-a))."""
  return tuple(a * t + (1 - a) * torch.roll(t, shifts=1, dims=0) for t in things)


# -----------------------------------------------------------------------------
# Tree helpers — utils.py:616-700, :1169-1212
# -----------------------------------------------------------------------------
def tree_map(f, tree, *rest):
  if isinstance(tree, dict):
    return {k: tree_map(f, tree[k], *[r[k] for r in rest]) for k in tree}
  return f(tree, *rest)


def tree_flatten_with_names(tree, prefix=""):
  """utils.py:616-641: sorted-key traversal, '/'-joined names."""
  if isinstance(tree, dict):
    res = []
    for k in sorted(tree.keys()):
      res += tree_flatten_with_names(tree[k], f"{prefix}{k}/")
    return res
  return [(prefix.rstrip("/"), tree)]


def recover_tree(names_and_vals):
  tree = {}
  for name, v in names_and_vals:
    node = tree
    parts = name.split("/")
    for p in parts[:-1]:
      node = node.setdefault(p, {})
    node[parts[-1]] = v
  return tree


def make_masks(names: Sequence[str], patterns: Sequence[str]):
  """utils.py:1195-1212: first-match-wins boolean masks, regex fullmatch."""
  comp = []
  for p in patterns:
    assert not p.startswith("/"), p
    comp.append(re.compile(p))
  masks = [dict() for _ in patterns]
  for n in names:
    hit = False
    for i, c in enumerate(comp):
      m = (not hit) and bool(c.fullmatch(n))
      masks[i][n] = m
      hit = hit or m
  return masks


# -----------------------------------------------------------------------------
# Schedules — utils.py:1002-1143
# -----------------------------------------------------------------------------
def steps(prefix, config, data_size=None, batch_size=None, total_steps=None,
          default=ValueError):
  """utils.py:1002-1067."""
  suffixes = {"steps", "examples", "epochs", "percent"}
  matches = {f"{prefix}_{s}" for s in suffixes
             if (x := config.get(f"{prefix}_{s}")) is not None and x >= 0}
  assert len(matches) <= 1, f"Only one of '{matches}' should be defined."
  if f"{prefix}_steps" in matches:
    return config[f"{prefix}_steps"]

  def to_integer(x):
    return max(1, round(x)) if x else 0

  if batch_size and f"{prefix}_examples" in matches:
    return to_integer(config[f"{prefix}_examples"] / batch_size)
  if batch_size and data_size and f"{prefix}_epochs" in matches:
    return to_integer(config[f"{prefix}_epochs"] * data_size / batch_size)
  if total_steps and f"{prefix}_percent" in matches:
    pct = config[f"{prefix}_percent"]
    assert 0.0 <= pct <= 1.0
    return to_integer(pct * total_steps)
  if default is ValueError:
    raise ValueError(f"Cannot convert {prefix} to steps")
  return default


def create_learning_rate_schedule(total_steps, batch_size=None, data_size=None,
                                  base=1.0, decay_type="stair",
                                  scale_with_batchsize=False, **kw):
  """utils.py:1070-1143; returns step -> float (float64 python math)."""

  def to_steps(name, default=0):
    return steps(name, kw, data_size, batch_size, total_steps, default=default)

  warmup_steps = to_steps("warmup")
  cooldown_steps = to_steps("cooldown")
  assert (total_steps <= 1) or (warmup_steps < total_steps)

  def step_fn(step):
    lr = base
    if scale_with_batchsize:
      lr = lr * batch_size / 256.0
    progress = (step - warmup_steps) / float(total_steps - warmup_steps)
    progress = min(max(progress, 0.0), 1.0)
    if decay_type in ("linear", "polynomial"):
      power = kw.get("power", 1)
      zero = kw.get("end", kw.get("linear_end", 0))
      lr = zero + (lr - zero) * (1.0 - progress) ** power
    elif decay_type == "cosine":
      lr = lr * 0.5 * (1.0 + math.cos(math.pi * progress))
    elif decay_type == "rsqrt":
      t = to_steps("timescale", default=kw.get("timescale", 10_000))
      shift = to_steps("shift", default=kw.get("shift", 0))
      if warmup_steps <= step:
        lr = lr / math.sqrt(1 + (step + shift - warmup_steps) / t)
      else:
        lr = lr / math.sqrt(1 + shift / t)
    elif decay_type == "stair":
      i = int(np.searchsorted(np.array(kw.get("steps", [])), step + 1))
      lr = lr * ([1.0] + list(kw.get("mults", [])))[i]
    else:
      raise ValueError(f"Unknown lr type {decay_type}")
    if warmup_steps:
      lr = lr * min(1.0, step / warmup_steps)
    if cooldown_steps:
      lr = lr * min(1.0, (total_steps - step) / cooldown_steps)
    return float(np.float32(lr))

  return step_fn


# -----------------------------------------------------------------------------
# Optimizer chain — optax.py:75-149 (bv_optax.make) restated on flat name->tensor
# -----------------------------------------------------------------------------
def clip_by_per_example_global_norm(per_example_grads, max_norm):
  """optax.py:54-72 (`clip_by_per_example_global_norm`): every leaf carries a leading batch axis; each
  example's gradient is scaled to a global norm of at most max_norm over ALL leaves
  (optax.per_example_global_norm_clip: scale_i = min(1, max_norm / ||g_i||)), the clipped gradients are
  summed over the batch and divided by the batch size.  List of [B, ...] tensors in, list of [...] out."""
  B = per_example_grads[0].shape[0]
  sq = sum((g.reshape(B, -1) ** 2).sum(1) for g in per_example_grads)
  norm = torch.sqrt(sq)
  scale = torch.clamp(max_norm / torch.clamp(norm, min=1e-30), max=1.0)
  out = []
  for g in per_example_grads:
    out.append((g * scale.view((B,) + (1,) * (g.dim() - 1))).sum(0) / B)
  return out


class OptaxOracle:
  """Functional restatement of `bv_optax.make(config, params, sched_kw=...)`.

  Chain order (optax.py:143-149): clip_by_global_norm(not frozen) ->
  optimizer(not frozen) -> scale(lr) -> lr_mults -> add_decayed_weights ->
  per-group scale_by_schedule / set_to_zero(frozen) -> scale(-1).
  Supports optax_name in {"scale", "scale_by_adam", "identity"}.
  """

  def __init__(self, config: dict, params: Tree, *, sched_kw: dict):
    self.config = config
    flat = tree_flatten_with_names(params)
    self.names = [n for n, _ in flat]
    schedule = config.get("schedule", {})
    if not isinstance(schedule, (tuple, list)):
      schedule = [(".*", schedule)]
    pats, scheds = zip(*schedule)
    masks = make_masks(self.names, pats)
    not_covered = [n for n in self.names if not any(m[n] for m in masks)]
    assert not not_covered, f"All params must be covered: {not_covered}"
    self.frozen = {n: any(m[n] for m, s in zip(masks, scheds) if s is None)
                   for n in self.names}
    self.sched_of = {}
    self.schedule_fns = []
    kw = dict(sched_kw)
    if "global_batch_size" in kw:  # name used by optax_test.py:120
      kw["batch_size"] = kw.pop("global_batch_size")
    for m, s in zip(masks, scheds):
      if s is None:
        continue
      s = dict(s)
      mult = s.pop("mult", 1.0)
      fn = create_learning_rate_schedule(base=mult, **kw, **s)
      self.schedule_fns.append(fn)
      for n in self.names:
        if m[n]:
          self.sched_of[n] = fn
    self.lr_mult = {n: 1.0 for n in self.names}
    if config.get("lr_mults"):
      pats, mults = zip(*config["lr_mults"])
      assert all(m > 0 for m in mults)
      for m, mult in zip(make_masks(self.names, pats), mults):
        for n in self.names:
          if m[n]:
            self.lr_mult[n] = mult
    self.wd = {n: 0.0 for n in self.names}
    if config.get("wd"):
      wd_mults = config.get("wd_mults", [(".*/kernel$", 1.0)])
      pats, mults = zip(*wd_mults)
      for m, mult in zip(make_masks(self.names, pats), mults):
        for n in self.names:
          if m[n]:
            self.wd[n] = config["wd"] * mult
    self.name = config["optax_name"]
    self.okw = dict(config.get("optax", {}))
    self.count = 0
    self.mu = {}
    self.nu = {}
    if self.name == "scale_by_adam":
      for n, p in flat:
        if not self.frozen[n]:  # optax_test.py:301-318: no state for frozen
          self.mu[n] = torch.zeros_like(p)
          self.nu[n] = torch.zeros_like(p)
    if self.name in ("big_vision.scale_by_adafactor", "scale_by_adafactor"):
      # optax.py:187-216 -> optax.scale_by_factored_rms state (v_row, v_col, v) + optax.ema state
      self.af = {}
      mind = self.okw.get("min_dim_size_to_factor", 32)
      for n, p in flat:
        if self.frozen[n]:
          continue
        fd = adafactor_factored_dims(tuple(p.shape), mind)
        if fd is None:
          st = dict(fd=None, v=torch.zeros_like(p))
        else:
          d1, d0 = fd
          st = dict(fd=fd, v_row=torch.zeros([s_ for a, s_ in enumerate(p.shape) if a != d0], dtype=p.dtype),
                    v_col=torch.zeros([s_ for a, s_ in enumerate(p.shape) if a != d1], dtype=p.dtype))
        st["ema"] = torch.zeros_like(p)
        self.af[n] = st

  def update(self, grads: Tree, params: Optional[Tree] = None) -> Tree:
    g = dict(tree_flatten_with_names(grads))
    p = dict(tree_flatten_with_names(params)) if params is not None else None
    step = self.count
    # clip_by_global_norm over not-frozen leaves (optax.py:100-105)
    if (c := self.config.get("grad_clip_norm")):
      sq = sum((g[n].double() ** 2).sum() for n in self.names if not self.frozen[n])
      norm = torch.sqrt(sq)
      factor = torch.clamp(c / norm, max=1.0).to(next(iter(g.values())).dtype) \
          if float(norm) > 0 else torch.tensor(1.0)
      g = {n: (g[n] if self.frozen[n] else g[n] * factor) for n in self.names}
    upd = {}
    for n in self.names:
      if self.frozen[n]:
        upd[n] = torch.zeros_like(g[n])
        continue
      u = g[n]
      if self.name == "scale":
        u = u * self.okw["step_size"]
      elif self.name == "scale_by_adam":
        b1 = self.okw.get("b1", 0.9)
        b2 = self.okw.get("b2", 0.999)
        eps = self.okw.get("eps", 1e-8)
        mu = b1 * self.mu[n].to(u.dtype) + (1 - b1) * u
        nu = b2 * self.nu[n] + (1 - b2) * u * u
        k = step + 1
        mu_hat = mu / (1 - b1 ** k)
        nu_hat = nu / (1 - b2 ** k)
        mu_dtype = self.okw.get("mu_dtype")
        self.mu[n] = mu.to(torch.bfloat16) if mu_dtype == "bfloat16" else mu
        self.nu[n] = nu
        u = mu_hat / (torch.sqrt(nu_hat) + eps)
      elif self.name in ("big_vision.scale_by_adafactor", "scale_by_adafactor"):
        u = self._adafactor(n, u, step)
      elif self.name in ("identity", "big_vision.sgd"):      # optax.py:227: big_vision.sgd = optax.identity
        pass
      else:
        raise NotImplementedError(self.name)
      u = u * self.config["lr"] * self.lr_mult[n]
      if self.wd[n]:
        u = u + self.wd[n] * p[n]
      u = u * self.sched_of[n](step)
      upd[n] = -u
    self.count += 1
    return recover_tree([(n, upd[n]) for n in self.names])


def adafactor_factored_dims(shape, min_dim_size_to_factor=32):
  """optax/_src/factorized.py `_factored_dims` with factored=True (optax is an un-vendored
  dependency of the reference, requirements.txt; this restates its published algorithm)."""
  import numpy as np
  if len(shape) < 2:
    return None
  sorted_dims = np.argsort(shape)
  if shape[sorted_dims[-2]] < min_dim_size_to_factor:
    return None
  return int(sorted_dims[-2]), int(sorted_dims[-1])


def _adafactor(self, n, g, step):
  """big_vision/optax.py:187-216: chain(scale_by_factored_rms(decay 1 - t^-0.8 capped 0.999, eps 1e-30),
  [clip_by_block_rms if clipping_threshold], ema(momentum, debias=False, accumulator bf16))."""
  kw = self.okw
  decay_rate, offset = kw.get("decay_rate", 0.8), kw.get("decay_offset", 0)
  cap, eps = kw.get("beta2_cap", 0.999), kw.get("eps", 1e-30)
  mom = kw.get("momentum", 0.9)
  t = float(step - offset) + 1.0
  beta2 = min(cap, 1.0 - t ** (-decay_rate))
  st = self.af[n]
  g2 = g * g + eps
  if st["fd"] is None:
    st["v"] = beta2 * st["v"] + (1 - beta2) * g2
    u = g * st["v"] ** -0.5
  else:
    d1, d0 = st["fd"]
    st["v_row"] = beta2 * st["v_row"] + (1 - beta2) * g2.mean(dim=d0)
    st["v_col"] = beta2 * st["v_col"] + (1 - beta2) * g2.mean(dim=d1)
    reduced_d1 = d1 - 1 if d1 > d0 else d1
    rcm = st["v_row"].mean(dim=reduced_d1, keepdim=True)
    row_factor = (st["v_row"] / rcm) ** -0.5
    col_factor = st["v_col"] ** -0.5
    u = g * row_factor.unsqueeze(d0) * col_factor.unsqueeze(d1)
  if kw.get("clipping_threshold"):
    thr = kw["clipping_threshold"]
    u = u / torch.clamp(torch.sqrt((u ** 2).mean()) / thr, min=1.0)      # optax.clip_by_block_rms
  if mom:
    ema = mom * st["ema"].to(u.dtype) + (1 - mom) * u
    acc = kw.get("dtype_momentum", "bfloat16")
    st["ema"] = ema.to(torch.bfloat16) if str(acc) in ("bfloat16", "torch.bfloat16") else ema
    u = ema
  return u


OptaxOracle._adafactor = _adafactor


# -----------------------------------------------------------------------------
# Parameter initialisation following the Flax initialisers
# -----------------------------------------------------------------------------
def _xavier_uniform(gen, shape, fan_in, fan_out, dtype):
  lim = math.sqrt(6.0 / (fan_in + fan_out))
  return (torch.rand(shape, generator=gen, dtype=dtype) * 2 - 1) * lim


def _lecun_normal(gen, shape, fan_in, dtype):
  # variance_scaling(1.0, "fan_in", "truncated_normal"); plain normal is used
  # here (same variance) — initial values never enter a parity comparison.
  return torch.randn(shape, generator=gen, dtype=dtype) * math.sqrt(1.0 / fan_in)


def _init_mha(gen, d, h, dtype):
  dh = d // h
  p = {}
  for name in ("query", "key", "value"):
    p[name] = {"kernel": _xavier_uniform(gen, (d, h, dh), d, d, dtype),
               "bias": torch.zeros(h, dh, dtype=dtype)}
  p["out"] = {"kernel": _xavier_uniform(gen, (h, dh, d), d, d, dtype),
              "bias": torch.zeros(d, dtype=dtype)}
  return p


def _init_mlp(gen, d, m, dtype):
  return {
      "Dense_0": {"kernel": _xavier_uniform(gen, (d, m), d, m, dtype),
                  "bias": torch.randn(m, generator=gen, dtype=dtype) * 1e-6},
      "Dense_1": {"kernel": _xavier_uniform(gen, (m, d), m, d, dtype),
                  "bias": torch.randn(d, generator=gen, dtype=dtype) * 1e-6},
  }


def _init_ln(d, dtype):
  return {"scale": torch.ones(d, dtype=dtype), "bias": torch.zeros(d, dtype=dtype)}


def _init_encoder(gen, depth, d, m, h, dtype):
  p = {}
  for i in range(depth):
    p[f"encoderblock_{i}"] = {
        "LayerNorm_0": _init_ln(d, dtype),
        "MultiHeadDotProductAttention_0": _init_mha(gen, d, h, dtype),
        "LayerNorm_1": _init_ln(d, dtype),
        "MlpBlock_0": _init_mlp(gen, d, m, dtype),
    }
  p["encoder_norm"] = _init_ln(d, dtype)
  return p


def _init_map(gen, d, m, h, dtype):
  # nn.initializers.xavier_uniform() on (1, 1, d): fan_in = shape[-2] = 1, fan_out = shape[-1] = d
  return {"probe": _xavier_uniform(gen, (1, 1, d), 1, d, dtype),
          "MultiHeadDotProductAttention_0": _init_mha(gen, d, h, dtype),
          "LayerNorm_0": _init_ln(d, dtype),
          "MlpBlock_0": _init_mlp(gen, d, m, dtype)}


def init_vit(gen, image_hw, *, num_classes=None, patch_size=(16, 16), width=768,
             depth=12, mlp_dim=None, num_heads=12, posemb="learn", rep_size=False,
             pool_type="gap", head_zeroinit=True, dtype=torch.float32, **_):
  mlp_dim = mlp_dim or 4 * width
  ph, pw = patch_size
  h, w = image_hw[0] // ph, image_hw[1] // pw
  p = {"embedding": {
      "kernel": _lecun_normal(gen, (ph, pw, 3, width), ph * pw * 3, dtype),
      "bias": torch.zeros(width, dtype=dtype)}}
  if posemb == "learn":
    p["pos_embedding"] = torch.randn((1, h * w, width), generator=gen, dtype=dtype) / math.sqrt(width)
  if pool_type == "tok":
    p["cls"] = torch.zeros(1, 1, width, dtype=dtype)
  p["Transformer"] = _init_encoder(gen, depth, width, mlp_dim, num_heads, dtype)
  if pool_type == "map":
    p["MAPHead_0"] = _init_map(gen, width, mlp_dim, num_heads, dtype)
  feat = width
  if rep_size:
    rs = width if rep_size is True else rep_size
    p["pre_logits"] = {"kernel": _lecun_normal(gen, (width, rs), width, dtype),
                       "bias": torch.zeros(rs, dtype=dtype)}
    feat = rs
  if num_classes:
    k = (torch.zeros(feat, num_classes, dtype=dtype) if head_zeroinit
         else _lecun_normal(gen, (feat, num_classes), feat, dtype))
    p["head"] = {"kernel": k, "bias": torch.zeros(num_classes, dtype=dtype)}
  return p


def init_text(gen, seq_len, *, num_classes, width=512, depth=12, mlp_dim=2048,
              num_heads=8, vocab_size=32_000, pool_type="last",
              dtype=torch.float32, **_):
  p = {"Embed_0": {"embedding": torch.randn((vocab_size, width), generator=gen, dtype=dtype) / math.sqrt(width)},
       "pos_embedding": torch.randn((1, seq_len, width), generator=gen, dtype=dtype) / math.sqrt(width),
       "Encoder_0": _init_encoder(gen, depth, width, mlp_dim, num_heads, dtype)}
  if pool_type == "map":
    p["MAPHead_0"] = _init_map(gen, width, mlp_dim, num_heads, dtype)
  if num_classes:
    p["head"] = {"kernel": _lecun_normal(gen, (width, num_classes), width, dtype),
                 "bias": torch.zeros(num_classes, dtype=dtype)}
  return p


def init_two_towers(seed, image_hw, seq_len, *, image_cfg, text_cfg, out_dim,
                    temperature_init=1.0, bias_init=None, dtype=torch.float32):
  gen = torch.Generator().manual_seed(seed)
  out_dims = (out_dim, out_dim) if isinstance(out_dim, int) else tuple(out_dim)
  ikw = {**decode_variant(image_cfg.get("variant")), **{k: v for k, v in image_cfg.items() if k != "variant"}}
  tkw = {**decode_variant(text_cfg.get("variant")), **{k: v for k, v in text_cfg.items() if k != "variant"}}
  p = {"img": init_vit(gen, image_hw, num_classes=out_dims[0], dtype=dtype, **ikw),
       "txt": init_text(gen, seq_len, num_classes=out_dims[1], dtype=dtype, **tkw),
       "t": torch.full((1,), math.log(temperature_init), dtype=dtype)}
  if bias_init is not None:
    p["b"] = torch.full((1,), float(bias_init), dtype=dtype)
  return p


def synthetic_batch(seed, n, res, seq_len, vocab_size=32_000, dtype=torch.float32):
  """U(-1,1) NHWC images; sticky-EOS tokens (pp/ops_text.py:143-155)."""
  gen = torch.Generator().manual_seed(seed)
  image = torch.rand((n, res, res, 3), generator=gen, dtype=torch.float32) * 2 - 1
  text = torch.randint(2, vocab_size, (n, seq_len), generator=gen, dtype=torch.int64)
  lens = torch.randint(min(4, seq_len - 1), seq_len, (n,), generator=gen)
  pos = torch.arange(seq_len)[None, :]
  text = torch.where(pos >= lens[:, None], torch.ones_like(text), text)
  return image.to(dtype), text.to(torch.int32)


def siglip_step_loss(params, image, text, *, image_cfg, text_cfg, out_dim, text_model=None, drop=None):
  """siglip.py:287-308 loss_fn: model.apply -> global sigmoid loss (drop: see two_towers_forward)."""
  zimg, ztxt, out = two_towers_forward(
      params, image, text.long(), image_cfg=image_cfg, text_cfg=text_cfg, out_dim=out_dim, text_model=text_model,
      drop=drop)
  loss, logits = siglip_loss_global(zimg, ztxt, out["t"], out["b"])
  return loss, (zimg, ztxt, logits, out)
