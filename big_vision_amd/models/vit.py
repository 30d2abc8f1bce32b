"""ViT on hand-written gfx950 kernels, behind big_vision's `models.vit` surface.

Mirrors the module-level API of the reference big_vision/models/vit.py:
`Model(num_classes=None, *, variant=None, **kw)` (:279-281), `.init` /
`.apply` with Flax-named parameter trees, `decode_variant` (:284-303),
`load` (:408-433), `VANITY_NAMES`.  The forward math follows
`_Model.__call__` (:206-276); every tensor op is a libbvhip kernel
(see big_vision_amd/engine.py), nothing runs in eager PyTorch.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Union

import numpy as np
import torch

from big_vision_amd import engine as E
from big_vision_amd import ops
from big_vision_amd import utils
from big_vision_amd.models import common
from big_vision_amd.params import Entry, ParamStore, ParamTree, adhoc_store

BF16, F32 = torch.bfloat16, torch.float32


def posemb_sincos_2d(h, w, width, temperature=10_000.0):
  """Constant sin-cos table [1, h*w, width] (reference vit.py:34-44), float32."""
  y, x = np.mgrid[:h, :w]
  assert width % 4 == 0, "Width must be mult of 4 for sincos posemb"
  omega = np.arange(width // 4) / (width // 4 - 1)
  omega = 1.0 / (temperature ** omega)
  y = np.einsum("m,d->md", y.flatten(), omega)
  x = np.einsum("m,d->md", x.flatten(), omega)
  pe = np.concatenate([np.sin(x), np.cos(x), np.sin(y), np.cos(y)], axis=1)
  return np.asarray(pe, np.float32)[None, :, :]


def decode_variant(variant):
  """Converts a string like "B" or "B/32" into a params dict (vit.py:284-303)."""
  if variant is None:
    return {}
  v, patch = variant, {}
  if "/" in variant:
    v, patch = variant.split("/")
    patch = {"patch_size": (int(patch), int(patch))}
  return {
      "width": {"mu": 32, "Ti": 192, "S": 384, "M": 512, "B": 768, "L": 1024, "So400m": 1152, "H": 1280, "g": 1408, "g-opt": 1536, "G": 1664, "G-opt": 1536, "e": 1792}[v],
      "depth": {"mu": 1, "Ti": 12, "S": 12, "M": 12, "B": 12, "L": 24, "So400m": 27, "H": 32, "g": 40, "g-opt": 40, "G": 48, "G-opt": 48, "e": 56}[v],
      "mlp_dim": {"mu": 128, "Ti": 768, "S": 1536, "M": 2048, "B": 3072, "L": 4096, "So400m": 4304, "H": 5120, "g": 6144, "g-opt": 6144, "G": 8192, "G-opt": 8192, "e": 15360}[v],
      "num_heads": {"mu": 2, "Ti": 3, "S": 6, "M": 8, "B": 12, "L": 16, "So400m": 16, "H": 16, "g": 16, "g-opt": 16, "G": 16, "G-opt": 16, "e": 16}[v],
      **patch
  }


def _seed_of(rng) -> int:
  if isinstance(rng, torch.Generator):
    return int(rng.initial_seed())
  a = np.asarray(rng).astype(np.uint64).ravel()
  s = 0
  for v in a:
    s = (s * 1000003 + int(v)) % (2 ** 63)
  return int(s)


def dropout_for(rate, train, rngs):
  """engine.Dropout of an `apply(..., train=, rngs=)` call, None when the call is deterministic (flax: a module with
  dropout > 0 applied with train=True and no "dropout" rng raises)."""
  if not train or not rate:
    return None
  if not rngs or "dropout" not in rngs:
    raise ValueError("train=True with dropout > 0 needs rngs={'dropout': key}")
  return E.Dropout(rate, _seed_of(rngs["dropout"]))


class VitExec:
  """Forward / backward of one image tower bound to a ParamStore at `prefix`."""

  def __init__(self, m: "_Model", store: ParamStore, prefix: str, hw):
    self.m, self.store = m, store
    D, H, M = m.width, m.num_heads, m.mlp_dim
    self.hw = hw
    self.wemb = E._W(store, f"{prefix}embedding/kernel", (m.patch_size[0] * m.patch_size[1] * 3, D))
    self.bemb = E._W(store, f"{prefix}embedding/bias")
    if m.posemb == "learn":
      self.pos = E._W(store, f"{prefix}pos_embedding", (hw[0] * hw[1], D))
      self.pos_const = None
    else:
      self.pos = None
      self.pos_const = torch.from_numpy(posemb_sincos_2d(hw[0], hw[1], D)[0]).to(store.device)
    self.cls = E._W(store, f"{prefix}cls", (1, D)) if m.pool_type == "tok" else None
    self.enc = E.Encoder(store, f"{prefix}Transformer", m.depth, D, H, M, scan=getattr(m, "scan", False))
    self.map = E.MAPHead(store, f"{prefix}MAPHead_0", D, H, M) if m.pool_type == "map" else None
    self.pre = None
    if m.rep_size:   # pre_logits = tanh(Dense(x)), vit.py:259-262
      self.pre = (E._W(store, f"{prefix}pre_logits/kernel"), E._W(store, f"{prefix}pre_logits/bias"))
    self.head = None
    if m.num_classes:
      self.head = (E._W(store, f"{prefix}head/kernel"), E._W(store, f"{prefix}head/bias"))

  # -------------------------------------------------------------- forward --
  def fwd(self, image, save=False, collect=False, drop=None):
    """drop (engine.Dropout, train mode with the model's dropout > 0): vit.py:228 behind the position embedding
    (+ cls) and the encoder blocks' sites; None = deterministic."""
    m = self.m
    D = m.width
    out = {}
    if drop is not None and not (drop.rate > 0.0):
      drop = None
    image = image.to(F32).contiguous()
    n = image.shape[0]
    patches, (h, w) = ops.patchify(image, m.patch_size[0])
    assert (h, w) == tuple(self.hw), f"image grid {(h, w)} != initialised grid {self.hw}"
    L0 = h * w
    pos = self.pos.f32 if self.pos is not None else self.pos_const
    x = E.linear_fwd(patches, self.wemb, self.bemb, out_dtype=F32, epilogue=ops.EPI_POS, aux=pos, aux_rows=L0)
    if collect:
      # `out` is part of the API (vit.py:207-274).  The stem output without the position
      # embedding is not materialised on the training path (the add is the GEMM's epilogue):
      # recompute it for the diagnostics dict.
      out["stem"] = E.linear_fwd(patches, self.wemb, self.bemb, out_dtype=F32).view(n, h, w, D)
      out["with_posemb"] = x.view(n, L0, D)
    L = L0
    if m.pool_type == "tok":
      x = ops.concat_cls(self.cls.f32, x, n, L0, D)
      L = L0 + 1
    enc_out = {} if collect else None
    k_pos = None
    if drop is not None:
      k_pos = (drop.rate, drop.key(E.DROP_POSEMB))
      x = ops.dropout_f32(x, k_pos[1], drop.rate, out=None if collect else x)   # vit.py:228 (out["with_posemb"] is the input)
    xL, saved = self.enc.fwd(x, n, L, save, enc_out, drop=drop)
    if collect:
      out["encoder"] = enc_out
    ctx = dict(n=n, L=L, L0=L0, patches=patches, enc=saved, xL=xL, k_pos=k_pos)
    T = n * L
    if m.pool_type == "map":
      y, _, mean, rstd = self.enc.norm.fwd(xL, T, D)
      z, msaved = self.map.fwd(y, n, L)
      ctx.update(norm=(mean, rstd), map=msaved)
      if collect:
        out["encoded"] = self.enc.norm.fwd(xL, T, D, want_bf16=False, want_f32=True)[1].view(n, L, D)
    elif m.pool_type == "gap":
      _, yf, mean, rstd = self.enc.norm.fwd(xL, T, D, want_bf16=False, want_f32=True)
      z = ops.pool_gap_fwd(yf, n, L, D)
      ctx.update(norm=(mean, rstd))
      if collect:
        out["encoded"] = yf.view(n, L, D)
    elif m.pool_type in ("0", "tok"):
      if collect:
        _, yf, mean, rstd = self.enc.norm.fwd(xL, T, D, want_bf16=False, want_f32=True)
        enc = yf.view(n, L, D)
        out["encoded"] = enc   # all L (+1 cls) tokens, as in the reference (vit.py:240); x_2d drops the cls row
        z = enc[:, 0].contiguous()
        mean, rstd = mean.view(n, L)[:, 0].contiguous(), rstd.view(n, L)[:, 0].contiguous()
      else:
        _, z, mean, rstd = self.enc.norm.fwd(xL, n, D, row_stride=L, row_offset=0, want_bf16=False, want_f32=True)
      ctx.update(norm=(mean, rstd))
    elif m.pool_type == "none":
      # vit.py:252-253: no pooling, the tail (pre_logits / head) runs on every token; forward only
      # (no trainer on the accelerated path back-propagates through an un-pooled tower)
      if save:
        raise NotImplementedError("pool_type='none' is forward-only on the accelerated path")
      _, yf, mean, rstd = self.enc.norm.fwd(xL, T, D, want_bf16=False, want_f32=True)
      z = yf
      if collect:
        out["encoded"] = yf.view(n, L, D)
    else:
      raise ValueError(f"Unknown pool type: '{m.pool_type}'")
    if m.pool_type != "none":
      out["head_input"] = z
    if self.pre is not None:
      zb0 = ops.cast_bf16(z)
      z = ops.tanh_fwd(E.linear_fwd(zb0, self.pre[0], self.pre[1], out_dtype=F32))
      ctx["pre"] = (zb0, z)
    out["pre_logits"] = z
    x = z
    if self.head is not None:
      zb = ops.cast_bf16(z)
      x = E.linear_fwd(zb, self.head[0], self.head[1], out_dtype=F32)
      out["logits"] = x
      ctx["head_in"] = zb
    if m.pool_type == "none":   # [n, L, features], like the reference's un-pooled x
      for k in ("pre_logits", "logits"):
        if k in out:
          out[k] = out[k].view(n, L, -1)
      x = x.view(n, L, -1)
    if collect and "encoded" in out:
      # the same tail applied to every patch token (vit.py:257-273: x_2d; unused by training)
      x2 = (out["encoded"][:, 1:] if m.pool_type == "tok" else out["encoded"]).contiguous().view(n * L0, D)
      if self.pre is not None:
        x2 = ops.tanh_fwd(E.linear_fwd(ops.cast_bf16(x2), self.pre[0], self.pre[1], out_dtype=F32))
      out["pre_logits_2d"] = x2.view(n, h, w, -1)
      if self.head is not None:
        out["logits_2d"] = E.linear_fwd(ops.cast_bf16(x2), self.head[0], self.head[1], out_dtype=F32).view(n, h, w, -1)
    return x, out, (ctx if save else None)

  # ------------------------------------------------------------- backward --
  def bwd(self, ctx, dx, on_block=None):
    m = self.m
    D = m.width
    n, L, L0 = ctx["n"], ctx["L"], ctx["L0"]
    T = n * L
    dz = dx.contiguous()
    if self.head is not None:
      dzb = ops.cast_bf16(dz)
      E.linear_bwd_w(ctx["head_in"], dzb, self.head[0], self.head[1], dy_for_bias=dz)
      dz = E.linear_bwd_x(dzb, self.head[0], out_dtype=F32)
    if self.pre is not None:
      zb0, y = ctx["pre"]
      dpl = ops.tanh_bwd(y, dz.contiguous())
      dplb = ops.cast_bf16(dpl)
      E.linear_bwd_w(zb0, dplb, self.pre[0], self.pre[1], dy_for_bias=dpl)
      dz = E.linear_bwd_x(dplb, self.pre[0], out_dtype=F32)
    mean, rstd = ctx["norm"]
    xL = ctx["xL"]
    dxL_bf = torch.empty((T, D), device=xL.device, dtype=BF16)
    if m.pool_type == "map":
      dy = self.map.bwd(ctx["map"], dz, n, L)
      dxL = self.enc.norm.bwd(dy, xL, mean, rstd, T, D, dx_bf16=dxL_bf, dx_colsum=self.enc.last_b2_grad(ctx["enc"]))
    elif m.pool_type == "gap":
      dyf = ops.pool_gap_bwd(dz, n, L, D)
      dxL = self.enc.norm.bwd(dyf, xL, mean, rstd, T, D, dx_bf16=dxL_bf, dx_colsum=self.enc.last_b2_grad(ctx["enc"]))
    else:
      dxL = torch.zeros((T, D), device=xL.device, dtype=F32)
      dxL_bf.zero_()
      self.enc.norm.bwd(dz, xL, mean, rstd, n, D, dx=dxL, dx_bf16=dxL_bf, row_stride=L, row_offset=0,
                        dx_colsum=self.enc.last_b2_grad(ctx["enc"]))
    dx0, dx0_bf = self.enc.bwd(ctx["enc"], dxL, dxL_bf, n, L, b2_done=not self.enc.dropped(ctx["enc"]), on_block=on_block)
    if ctx.get("k_pos") is not None:    # backward of the dropout behind the position embedding: the same keep bits
      ops.dropout_f32(dx0, ctx["k_pos"][1], ctx["k_pos"][0], out=dx0, out_bf16=dx0_bf)
    if m.pool_type == "tok":
      if self.cls.grad is not None:
        ops.colsum(dx0.view(n, L * D)[:, :D], self.cls.grad.view(-1))
      dx0 = dx0.view(n, L, D)[:, 1:].contiguous().view(n * L0, D)
      dx0_bf = ops.cast_bf16(dx0)
    E.linear_bwd_w(ctx["patches"], dx0_bf, self.wemb, self.bemb, dy_for_bias=dx0)
    if self.pos is not None and self.pos.grad is not None:
      ops.batchsum(dx0, self.pos.grad, n, L0, D)


class _Model:
  """ViT model (configuration holder + Flax-like init/apply).

  Two reference arguments are accepted for config compatibility and have no effect here, by design:
  `remat_policy` (models/vit.py:129-148: XLA rematerialisation of scanned blocks) - what the backward
  keeps or re-derives is decided by the trainer's explicit contexts (full / light);
  `dtype_mm` (models/vit.py:209: the dtype matmul inputs are cast to) - the contractions always run
  on bf16 MFMA operands with fp32 accumulation, the precision `dtype_mm="bfloat16"` asks for and
  what `float32` + XLA's default TPU matmul precision amounts to; parameters, residual stream,
  LayerNorm, softmax and the loss stay fp32 in both.
  """

  def __init__(self, num_classes: Optional[int] = None, patch_size: Sequence[int] = (16, 16),
               width: int = 768, depth: int = 12, mlp_dim: Optional[int] = None, num_heads: int = 12,
               posemb: str = "learn", rep_size: Union[int, bool] = False, dropout: float = 0.0,
               pool_type: str = "gap", head_zeroinit: bool = True, scan: bool = False,
               remat_policy: str = "nothing_saveable", dtype_mm: str = "float32", name=None):
    if posemb not in ("learn", "sincos2d"):
      raise ValueError(f"Unknown posemb type: {posemb}")
    if pool_type not in ("map", "gap", "0", "tok", "none"):
      raise ValueError(f"Unknown pool type: '{pool_type}'")
    if not 0.0 <= float(dropout) < 1.0:
      raise ValueError(f"dropout must be in [0, 1), got {dropout}")
    self.dropout = float(dropout)
    if width % num_heads or (width // num_heads) % 8 or width // num_heads > 128:
      raise NotImplementedError(f"attention kernels need a head_dim that is a multiple of 8 and <= 128 (64 is the "
                                f"fast path), got {width}/{num_heads}")
    self.num_classes, self.patch_size = num_classes, tuple(patch_size)
    self.width, self.depth, self.mlp_dim = width, depth, mlp_dim or 4 * width
    self.num_heads, self.posemb, self.rep_size = num_heads, posemb, rep_size
    self.pool_type, self.head_zeroinit, self.scan = pool_type, head_zeroinit, scan
    self.name = name
    self._execs = {}

  # -------------------------------------------------------------- params ---
  def entries(self, prefix, hw):
    D, H, M = self.width, self.num_heads, self.mlp_dim
    ph, pw = self.patch_size
    ents = [Entry(f"{prefix}embedding/kernel", (ph, pw, 3, D), E.init_lecun_normal(ph * pw * 3)),
            Entry(f"{prefix}embedding/bias", (D,), E.init_zeros)]
    if self.posemb == "learn":
      ents.append(Entry(f"{prefix}pos_embedding", (1, hw[0] * hw[1], D), E.init_normal(1 / math.sqrt(D))))
    if self.pool_type == "tok":
      ents.append(Entry(f"{prefix}cls", (1, 1, D), E.init_zeros))
    ents += E.encoder_entries(f"{prefix}Transformer", self.depth, D, H, M)
    if self.pool_type == "map":
      ents += E.map_entries(f"{prefix}MAPHead_0", D, H, M)
    feat = D
    if self.rep_size:
      rs = D if self.rep_size is True else self.rep_size
      ents += [Entry(f"{prefix}pre_logits/kernel", (D, rs), E.init_lecun_normal(D)),
               Entry(f"{prefix}pre_logits/bias", (rs,), E.init_zeros)]
      feat = rs
    if self.num_classes:
      kinit = E.init_zeros if self.head_zeroinit else E.init_lecun_normal(feat)
      ents += [Entry(f"{prefix}head/kernel", (feat, self.num_classes), kinit),
               Entry(f"{prefix}head/bias", (self.num_classes,), E.init_zeros)]
    return ents

  def grid(self, image_shape):
    return (image_shape[1] // self.patch_size[0], image_shape[2] // self.patch_size[1])

  def scan_prefixes(self, prefix=""):
    """Encoders presented with stacked blocks (scan=True, vit.py:129-148)."""
    return (f"{prefix}Transformer",) if self.scan else ()

  def init(self, rng, image, **kw):
    del kw
    shape = tuple(image.shape)
    dev = image.device if torch.is_tensor(image) and image.is_cuda else torch.device("cuda", torch.cuda.current_device())
    store = ParamStore(self.entries("", self.grid(shape)), dev, scan_prefixes=self.scan_prefixes())
    store.init_random(_seed_of(rng))
    store.refresh_shadow()
    return {"params": store.tree()}

  def executor(self, store, prefix, hw):
    key = (id(store), prefix, tuple(hw), getattr(store, "want_grads", False))
    if key not in self._execs:
      self._execs[key] = VitExec(self, store, prefix, hw)
    return self._execs[key]

  def _store_for(self, params, hw):
    if isinstance(params, ParamTree) and params.store is not None:
      return params.store, params.prefix
    dev = torch.device("cuda", torch.cuda.current_device())
    return adhoc_store(self._execs, ("vit", tuple(hw), dev.index), params,
                       lambda: ParamStore(self.entries("", hw), dev, scan_prefixes=self.scan_prefixes())), ""

  def apply(self, variables, image, *, train=False, rngs=None, collect=True, **kw):
    """train=True with dropout > 0 needs rngs={"dropout": key} (vit.py:228, train.py:298), as in the reference."""
    del kw
    hw = self.grid(tuple(image.shape))
    store, prefix = self._store_for(variables["params"], hw)
    store.refresh_shadow()
    x, out, _ = self.executor(store, prefix, hw).fwd(image, save=False, collect=collect,
                                                     drop=dropout_for(self.dropout, train, rngs))
    return x, out

  __call__ = None  # Flax-style direct calls are not supported; use .apply


def Model(num_classes=None, *, variant=None, **kw):  # pylint: disable=invalid-name
  """Factory function (reference vit.py:279-281)."""
  return _Model(num_classes, **{**decode_variant(variant), **kw})


# ----------------------------------------------------------- checkpoint I/O --
def resample_posemb(old, new):
  """"High-res finetuning": bilinear resize of the posemb grid (vit.py:306-321)."""
  import scipy.ndimage
  # `new` is only consulted for its shape: it may be a device view of the flat parameter store
  old, new_shape = np.asarray(old), tuple(new.shape)
  if old.shape == new_shape:
    return old
  gs_old = int(np.sqrt(old.shape[1]))
  gs_new = int(np.sqrt(new_shape[1]))
  grid = old.reshape(gs_old, gs_old, -1)
  zoom = (gs_new / gs_old, gs_new / gs_old, 1)
  grid = scipy.ndimage.zoom(grid, zoom, order=1)
  return grid.reshape(1, gs_new * gs_new, -1)


def fix_old_checkpoints(params):
  """Small backward-compat fix-ups that names alone cannot express (vit.py:324-360)."""
  params = utils.tree_map(lambda x: x, params)  # structural copy
  t = params["Transformer"]
  if "posembed_input" in t:
    params["pos_embedding"] = t.pop("posembed_input")["pos_embedding"]
  if "pos_embedding" in t:
    params["pos_embedding"] = t.pop("pos_embedding")
  if "pos_embedding" in params:
    pe = np.asarray(params["pos_embedding"])
    if int(np.sqrt(pe.shape[1])) ** 2 + 1 == int(pe.shape[1]):
      pe_cls, params["pos_embedding"] = pe[:, :1], pe[:, 1:]
      if "cls" in params:
        params["cls"] = np.asarray(params["cls"]) + pe_cls
  if "probe" in params:
    params["MAPHead_0"] = {k: params.pop(k) for k in
                           ["probe", "MlpBlock_0", "MultiHeadDotProductAttention_0", "LayerNorm_0"]}
  return params


def pyloop_to_scan(params_pyloop):
  """encoderblock_{i} -> stacked `encoderblock` (vit.py:363-386)."""
  p = utils.tree_map(lambda x: x, params_pyloop)
  t = p["Transformer"]
  depth = 1 + max(int(k.split("_")[-1]) for k in t if k.startswith("encoderblock_"))
  blocks = [t.pop(f"encoderblock_{i}") for i in range(depth)]
  t["encoderblock"] = utils.tree_map(lambda *v: np.stack([np.asarray(a) for a in v]), *blocks)
  return p


def scan_to_pyloop(params_scan):
  """Stacked `encoderblock` -> encoderblock_{i} (vit.py:389-405)."""
  p = utils.tree_map(lambda x: x, params_scan)
  t = p["Transformer"]
  stacked = t.pop("encoderblock")
  depth = len(stacked["LayerNorm_0"]["bias"])
  for i in range(depth):
    t[f"encoderblock_{i}"] = utils.tree_map(lambda x, i=i: np.asarray(x)[i], stacked)
  return p


def _wants_stacked_blocks(init_params, model_cfg):
  """Layout the MODEL presents: the init tree says it when there is one, otherwise `model_cfg.scan` (vit.py:416-424)."""
  if init_params:
    return "encoderblock" in init_params.get("Transformer", {})
  return bool((model_cfg or {}).get("scan", False))


def load(init_params, init_file, model_cfg, dont_load=()):  # pylint: disable=invalid-name
  """Checkpoint (current or legacy layout, loop or scan blocks, any posemb grid) -> tree shaped like `init_params`
  (contract of vit.py:408-433): legacy names fixed, blocks re-laid-out to the model's loop / scan form, leaves matching
  `dont_load` kept at their init value, position embeddings resampled to the model's grid."""
  ckpt = fix_old_checkpoints(utils.load_params(VANITY_NAMES.get(init_file, init_file)))
  stacked = "encoderblock" in ckpt["Transformer"]
  if stacked != _wants_stacked_blocks(init_params, model_cfg):
    ckpt = scan_to_pyloop(ckpt) if stacked else pyloop_to_scan(ckpt)
  restored_params = common.merge_params(ckpt, init_params, dont_load)
  target_posemb = (init_params or {}).get("pos_embedding")
  if target_posemb is not None:
    restored_params["pos_embedding"] = resample_posemb(old=restored_params["pos_embedding"], new=target_posemb)
  return restored_params


# Shortcut names for some canonical paper checkpoints (paths are gs:// buckets,
# unreachable offline; kept so configs referring to them resolve the same way).
VANITY_NAMES = {
    "howto-i21k-Ti/16": "gs://vit_models/augreg/Ti_16-i21k-300ep-lr_0.001-aug_none-wd_0.03-do_0.0-sd_0.0.npz",
    "howto-i21k-S/16": "gs://vit_models/augreg/S_16-i21k-300ep-lr_0.001-aug_light1-wd_0.03-do_0.0-sd_0.0.npz",
    "howto-i21k-B/16": "gs://vit_models/augreg/B_16-i21k-300ep-lr_0.001-aug_medium1-wd_0.1-do_0.0-sd_0.0.npz",
    "howto-i21k-L/16": "gs://vit_models/augreg/L_16-i21k-300ep-lr_0.001-aug_strong1-wd_0.1-do_0.0-sd_0.0.npz",
}
