"""NaFlex ViT image tower behind `models.proj.image_text.naflex_vit` (native aspect ratio,
variable number of patches per image, padding masked out).

Mirrors big_vision/models/proj/image_text/naflex_vit.py:200-293: `Model(num_classes, variant=...,
posemb="learn_2d(64)", nposemb=P, patchln_pre/post, pool_type in {map, gap, none})`; the input is
the tuple `(patches [n, N, ph*pw*3], ptype [n, N], yabs [n, N], xabs [n, N])` (:227), ptype == 1
marks a real patch, 0 padding; parameters: `embedding` (a Dense on the flattened patch, :233-234),
`pos_embedding` [P, P, width] (:243-246), optional `patchln_pre` / `patchln_post`, `Transformer`,
`MAPHead_0`, `pre_logits`, `head`; `load = vit.load` (:296).

What runs here (all libbvhip kernels, explicit forward / backward):
  * stem = k-major bf16 GEMM on the pre-patchified input with bias;
  * position embedding (`_pos_emb_resize` :38-83): per token the bilinear-antialias resize weights
    from the learned grid to the example's own patch grid, gathered at the token's coordinates
    (bv_naflex_posemb_weights), then ONE GEMM  tok_pos = W [tokens, P*P] . pos [P*P, width]  whose
    +residual epilogue adds it to the stem output; the backward is the transposed GEMM W^T d tok;
  * encoder blocks with the key-padding mask (bv_attn_fwd/bwd_masked: keys of padding tokens get
    zero probability and zero dK / dV; the reference also masks padded QUERY rows (:255-256), which
    only changes the values of padding rows - nothing a valid token, the pooled output or any
    gradient depends on);
  * pooling over the valid tokens only: MAP head with the pool mask (bv_map_attn_fwd_masked) or
    masked mean (bv_pool_gap_masked_*); pool_type "max" is not implemented.
The mask must be a PREFIX of the sequence (padding at the end, as the NaFlex preprocessing
produces it); this is verified on the first batch an executor sees.
"""
from __future__ import annotations

import math
import re
from typing import Optional, Union

import torch

from big_vision_amd import engine as E
from big_vision_amd import ops
from big_vision_amd.models import vit
from big_vision_amd.params import Entry, ParamStore, ParamTree, adhoc_store

F32, BF16 = torch.float32, torch.bfloat16


def _decode_posemb(posemb):
  """'learn_2d(64)' -> ('learn_2d', 64) (:30-34)."""
  m = re.fullmatch(r"learn_2d(\(\d+\))", posemb)
  if m:
    return "learn_2d", int(m.groups()[0][1:-1])
  return posemb, None


class NaflexExec:
  """Forward / backward of one NaFlex tower bound to a ParamStore at `prefix`."""

  def __init__(self, m: "_Model", store: ParamStore, prefix: str, patch_dim: int):
    self.m, self.store = m, store
    D, H, M, P = m.width, m.num_heads, m.mlp_dim, m.nposemb
    if patch_dim % 8:
      # The GEMM operands move in 16-byte (8 x bf16) pieces.  The pixel-input ViT pads its im2col rows to a
      # multiple of 8 inside bv_patchify_ld (14 x 14 x 3 = 588 -> 592); here the patches ARRIVE flattened
      # and go through patchln_pre / a cast unpadded, so the stem's K would not match the padded weight.
      raise NotImplementedError(
          f"NaFlex stem with patch_dim = {patch_dim} (not a multiple of 8): the flattened patches are consumed "
          "as they arrive (naflex_vit.py:227-236) and the stem GEMM needs 16-byte rows; use a patch size whose "
          "P*P*3 is a multiple of 8 (16x16, 32x32) or zero-pad the patches and the embedding kernel's input rows")
    self.wemb = E._W(store, f"{prefix}embedding/kernel", (patch_dim, D))
    self.bemb = E._W(store, f"{prefix}embedding/bias")
    self.pos = E._W(store, f"{prefix}pos_embedding", (P * P, D))
    self.ln_pre = E.LN(store, f"{prefix}patchln_pre") if m.patchln_pre else None
    self.ln_post = E.LN(store, f"{prefix}patchln_post") if m.patchln_post else None
    self.enc = E.Encoder(store, f"{prefix}Transformer", m.depth, D, H, M, scan=getattr(m, "scan", False))
    self.map = E.MAPHead(store, f"{prefix}MAPHead_0", D, H, M) if m.pool_type == "map" else None
    self.pre = (E._W(store, f"{prefix}pre_logits/kernel"), E._W(store, f"{prefix}pre_logits/bias")) if m.rep_size else None
    self.head = (E._W(store, f"{prefix}head/kernel"), E._W(store, f"{prefix}head/bias")) if m.num_classes else None
    self.patch_dim = patch_dim
    self._mask_checked = False

  def _lengths(self, ptype):
    """Valid tokens per example (ptype == 1, naflex_vit.py:84-94) and, when they are not a prefix of the
    sequence, the per-example permutation that moves them to the front.  The attention kernels mask a SUFFIX
    of the keys; the tower is equivariant to a permutation of an example's tokens (the position embedding is
    gathered per token from its own (yabs, xabs), masked MAP / gap / max pooling are permutation invariant),
    so a mask with holes is served by reordering the example's tokens - outputs per token are put back in
    the caller's order.  The prefix test is one host sync per call of a NEW mask layout."""
    valid = ptype == 1
    lens = valid.sum(dim=1).to(torch.int32).contiguous()
    N = ptype.shape[1]
    prefix = torch.arange(N, device=ptype.device)[None, :] < lens[:, None]
    if int(lens.min()) < 1:
      raise ValueError("an example without any real patch")
    if torch.equal(valid, prefix):
      return lens, None
    perm = torch.argsort((~valid).to(torch.int8), dim=1, stable=True)   # valid tokens first, original order kept
    return lens, perm

  def fwd(self, image, save=False, collect=False):
    m = self.m
    D, P = m.width, m.nposemb
    patches, ptype, yabs, xabs = image
    n, N, pd = patches.shape
    assert pd == self.patch_dim, (pd, self.patch_dim)
    T = n * N
    out = {}
    lens, perm = self._lengths(ptype)
    if perm is not None:   # holes in the mask: reorder every example's tokens (input-side gather, like patchify)
      if m.pool_type == "none":
        raise NotImplementedError("pool_type='none' with a non-prefix NaFlex mask")
      patches = torch.gather(patches, 1, perm[:, :, None].expand(-1, -1, pd))
      yabs, xabs = torch.gather(yabs, 1, perm), torch.gather(xabs, 1, perm)
    x_in = patches.to(F32).contiguous().view(T, pd)
    ctx = dict(n=n, N=N, lens=lens)
    if self.ln_pre is not None:
      pb, _, mean, rstd = self.ln_pre.fwd(x_in, T, pd)
      ctx["ln_pre"] = (x_in, mean, rstd)
    else:
      pb = ops.cast_bf16(x_in)
    W = ops.naflex_posemb_weights(yabs.to(torch.int32).contiguous(), xabs.to(torch.int32).contiguous(), P)
    if self.ln_post is None:
      tokpos = ops.gemm(W, self.pos.bf_t(), a_kmajor=True, b_kmajor=True, out_dtype=F32)
      x = E.linear_fwd(pb, self.wemb, self.bemb, out_dtype=F32, epilogue=ops.EPI_RESIDUAL, aux=tokpos)
      if collect:
        out["stem"] = (x - tokpos).view(n, N, D)
    else:
      t0 = E.linear_fwd(pb, self.wemb, self.bemb, out_dtype=F32)
      _, t1, mean, rstd = self.ln_post.fwd(t0, T, D, want_bf16=False, want_f32=True)
      x = ops.gemm(W, self.pos.bf_t(), a_kmajor=True, b_kmajor=True, out_dtype=F32, epilogue=ops.EPI_RESIDUAL, aux=t1)
      ctx["ln_post"] = (t0, mean, rstd)
      if collect:
        out["stem"] = t0.view(n, N, D)
    if collect:
      out["with_posemb"] = x.view(n, N, D)
    enc_out = {} if collect else None
    xL, saved = self.enc.fwd(x, n, N, save, enc_out, kv_len=lens)
    if collect:
      out["encoder"] = enc_out
    ctx.update(pb=pb, W=W, enc=saved, xL=xL)
    if m.pool_type == "map":
      y, _, mean, rstd = self.enc.norm.fwd(xL, T, D)
      z, msaved = self.map.fwd(y, n, N, kv_len=lens)
      ctx.update(norm=(mean, rstd), map=msaved)
      if collect:
        out["encoded"] = self.enc.norm.fwd(xL, T, D, want_bf16=False, want_f32=True)[1].view(n, N, D)
    elif m.pool_type == "gap":
      _, yf, mean, rstd = self.enc.norm.fwd(xL, T, D, want_bf16=False, want_f32=True)
      z = ops.pool_gap_fwd(yf, n, N, D, lens=lens)
      ctx.update(norm=(mean, rstd))
      if collect:
        out["encoded"] = yf.view(n, N, D)
    elif m.pool_type == "max":        # naflex_vit.py:267-271: padded tokens never win the maximum
      _, yf, mean, rstd = self.enc.norm.fwd(xL, T, D, want_bf16=False, want_f32=True)
      z, argmax = ops.pool_max_fwd(yf, n, N, D, lens=lens)
      ctx.update(norm=(mean, rstd), argmax=argmax)
      if collect:
        out["encoded"] = yf.view(n, N, D)
    elif m.pool_type == "none":
      if save:
        raise NotImplementedError("pool_type='none' is forward-only on the accelerated path")
      _, yf, mean, rstd = self.enc.norm.fwd(xL, T, D, want_bf16=False, want_f32=True)
      z = yf
      if collect:
        out["encoded"] = yf.view(n, N, D)
    else:
      raise NotImplementedError(f"pool_type '{m.pool_type}' (naflex_vit.py:267-271) is not implemented")
    out["head_input"] = z.view(n, N, D) if m.pool_type == "none" else z      # (naflex_vit.py:276: set for every pool_type)
    if self.pre is not None:
      zb0 = ops.cast_bf16(z)
      z = ops.tanh_fwd(E.linear_fwd(zb0, self.pre[0], self.pre[1], out_dtype=F32))
      ctx["pre"] = (zb0, z)
    per_token = (lambda t: t.view(n, N, -1)) if m.pool_type == "none" else (lambda t: t)     # [n, N, .] like the reference's
    out["pre_logits"] = per_token(z)
    x = z
    if self.head is not None:
      zb = ops.cast_bf16(z)
      x = E.linear_fwd(zb, self.head[0], self.head[1], out_dtype=F32)
      out["logits"] = per_token(x)
      ctx["head_in"] = zb
    x = per_token(x)
    if perm is not None and collect:   # per-token diagnostics back in the caller's token order
      inv = torch.argsort(perm, dim=1)

      def unperm(v):
        if isinstance(v, dict):
          return {k: unperm(t) for k, t in v.items()}
        if torch.is_tensor(v) and v.dim() == 3 and v.shape[:2] == (n, N):
          return torch.gather(v, 1, inv[:, :, None].expand(-1, -1, v.shape[2]))
        return v
      out = unperm(out)
    return x, out, (ctx if save else None)

  def bwd(self, ctx, dx, on_block=None):
    m = self.m
    D = m.width
    n, N, lens = ctx["n"], ctx["N"], ctx["lens"]
    T = n * N
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
      dy = self.map.bwd(ctx["map"], dz, n, N)          # works from the saved (masked) probabilities
      dxL = self.enc.norm.bwd(dy, xL, mean, rstd, T, D, dx_bf16=dxL_bf, dx_colsum=self.enc.last_b2_grad())
    else:
      dyf = ops.pool_max_bwd(dz, ctx["argmax"], n, N, D) if m.pool_type == "max" else ops.pool_gap_bwd(dz, n, N, D, lens=lens)
      dxL = self.enc.norm.bwd(dyf, xL, mean, rstd, T, D, dx_bf16=dxL_bf, dx_colsum=self.enc.last_b2_grad())
    dx0, dx0_bf = self.enc.bwd(ctx["enc"], dxL, dxL_bf, n, N, b2_done=True, on_block=on_block, kv_len=lens)
    # x = stem (+ patchln_post) + W . pos : d pos = W^T d x
    E.linear_bwd_w(ctx["W"], dx0_bf, self.pos, None)
    if self.ln_post is not None:
      t0, mean, rstd = ctx["ln_post"]
      dt0_bf = torch.empty((T, D), device=xL.device, dtype=BF16)
      dt0 = self.ln_post.bwd(dx0, t0, mean, rstd, T, D, dx_bf16=dt0_bf)
    else:
      dt0, dt0_bf = dx0, dx0_bf
    E.linear_bwd_w(ctx["pb"], dt0_bf, self.wemb, self.bemb, dy_for_bias=dt0)
    if self.ln_pre is not None:
      x_in, mean, rstd = ctx["ln_pre"]
      dpb = E.linear_bwd_x(dt0_bf, self.wemb)
      self.ln_pre.bwd(dpb, x_in, mean, rstd, T, self.patch_dim)     # only its scale / bias gradients are needed


class _Model:
  """NaFlex ViT (configuration holder + Flax-like init / apply)."""

  def __init__(self, num_classes: Optional[int] = None, width: int = 768, depth: int = 12,
               mlp_dim: Optional[int] = None, num_heads: int = 12, rep_size: Union[int, bool] = False,
               pool_type: str = "gap", head_zeroinit: bool = True, scan: bool = False,
               remat_policy: str = "nothing_saveable", dtype_mm: str = "float32", posemb: str = "learn_2d(64)",
               nposemb: Optional[int] = None, patchln_pre: bool = False, patchln_post: bool = False,
               patch_size=(16, 16), name=None):
    kind, grid = _decode_posemb(posemb)
    if kind != "learn_2d":
      raise ValueError(f"Unknown posemb: '{posemb}'")
    if nposemb is None:
      raise ValueError("nposemb (side of the learned position grid) needs to be set")
    if nposemb > 64:
      raise NotImplementedError("nposemb > 64")
    if pool_type not in ("map", "gap", "max", "none"):
      raise ValueError(f"Unknown pool type: '{pool_type}'")
    if width % num_heads or (width // num_heads) % 8 or width // num_heads > 128:
      raise NotImplementedError(f"attention kernels need a head_dim that is a multiple of 8 and <= 128 (64 is the "
                                f"fast path), got {width}/{num_heads}")
    self.num_classes, self.width, self.depth = num_classes, width, depth
    self.mlp_dim, self.num_heads, self.rep_size = mlp_dim or 4 * width, num_heads, rep_size
    self.pool_type, self.head_zeroinit, self.scan = pool_type, head_zeroinit, scan
    self.posemb, self.posemb_grid, self.nposemb = posemb, grid or 64, int(nposemb)
    self.patchln_pre, self.patchln_post = patchln_pre, patchln_post
    self.patch_size = tuple(patch_size)     # only used by decode_variant-driven factories; the input is pre-patchified
    self.name = name
    self._execs = {}

  def entries(self, prefix, patch_dim):
    D, H, M, P = self.width, self.num_heads, self.mlp_dim, self.nposemb
    ents = []
    if self.patchln_pre:
      ents += E.ln_entries(f"{prefix}patchln_pre")(patch_dim)
    ents += [Entry(f"{prefix}embedding/kernel", (patch_dim, D), E.init_lecun_normal(patch_dim)),
             Entry(f"{prefix}embedding/bias", (D,), E.init_zeros)]
    if self.patchln_post:
      ents += E.ln_entries(f"{prefix}patchln_post")(D)
    ents.append(Entry(f"{prefix}pos_embedding", (P, P, D), E.init_normal(1 / math.sqrt(D))))
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
    """two_towers asks the image tower for its 'grid'; for NaFlex the layout key is the patch dimension."""
    return int(image_shape[-1])

  def scan_prefixes(self, prefix=""):
    return (f"{prefix}Transformer",) if self.scan else ()

  def init(self, rng, image, **kw):
    del kw
    patches = image[0]
    dev = patches.device if torch.is_tensor(patches) and patches.is_cuda else torch.device("cuda", torch.cuda.current_device())
    store = ParamStore(self.entries("", int(patches.shape[-1])), dev, scan_prefixes=self.scan_prefixes())
    store.init_random(vit._seed_of(rng))
    store.refresh_shadow()
    return {"params": store.tree()}

  def executor(self, store, prefix, patch_dim):
    key = (id(store), prefix, int(patch_dim), getattr(store, "want_grads", False))
    if key not in self._execs:
      self._execs[key] = NaflexExec(self, store, prefix, int(patch_dim))
    return self._execs[key]

  def apply(self, variables, image, *, train=False, rngs=None, collect=True, **kw):
    del rngs, train, kw
    pd = int(image[0].shape[-1])
    params = variables["params"]
    if isinstance(params, ParamTree) and params.store is not None:
      store, prefix = params.store, params.prefix
    else:
      dev = torch.device("cuda", torch.cuda.current_device())
      store = adhoc_store(self._execs, ("naflex", pd, dev.index), params,
                          lambda: ParamStore(self.entries("", pd), dev, scan_prefixes=self.scan_prefixes()))
      prefix = ""
    store.refresh_shadow()
    x, out, _ = self.executor(store, prefix, pd).fwd(image, save=False, collect=collect)
    return x, out


def Model(num_classes=None, *, variant=None, **kw):  # pylint: disable=invalid-name
  """Factory function (naflex_vit.py:288-290)."""
  return _Model(num_classes, **{**vit.decode_variant(variant), **kw})


load = vit.load
