"""Text tower on gfx950 kernels behind `models.proj.image_text.text_transformer`.

Mirrors big_vision/models/proj/image_text/text_transformer.py: `Model(num_classes,
*, variant=None, **kw)` (:102-104), Flax-named params (`Embed_0/embedding`,
`pos_embedding`, `Encoder_0/...`, `head/...`), forward per `_Model.__call__`
(:55-99): embed -> +posemb -> bidirectional vit.Encoder -> pooling -> head.
`vocab_logits` (:80) is dead code for the loss and is only produced on request.
"""
from __future__ import annotations

import math

import torch

from big_vision_amd import engine as E
from big_vision_amd import ops
from big_vision_amd import utils
from big_vision_amd.models import common
from big_vision_amd.models import vit
from big_vision_amd.params import Entry, ParamStore, ParamTree, adhoc_store

BF16, F32 = torch.bfloat16, torch.float32


class TextExec:
  def __init__(self, m: "_Model", store: ParamStore, prefix: str, seq_len: int):
    self.m, self.store, self.seq_len = m, store, seq_len
    D, H, M = m.width, m.num_heads, m.mlp_dim
    self.table = E._W(store, f"{prefix}Embed_0/embedding")
    self.pos = E._W(store, f"{prefix}pos_embedding", (seq_len, D))
    self.enc = E.Encoder(store, f"{prefix}Encoder_0", m.depth, D, H, M, scan=getattr(m, "scan", False))
    self.map = E.MAPHead(store, f"{prefix}MAPHead_0", D, H, M) if m.pool_type == "map" else None
    self.head = None
    if m.num_classes:
      self.head = (E._W(store, f"{prefix}head/kernel"), E._W(store, f"{prefix}head/bias"))

  def fwd(self, text, save=False, collect=False, drop=None):
    """drop (engine.Dropout): train mode with dropout > 0 - the encoder blocks' sites (text_transformer.py:72-75)."""
    m = self.m
    D = m.width
    out = {}
    ids = text.to(torch.int32).contiguous()
    n, L = ids.shape
    assert L == self.seq_len, f"text length {L} != initialised length {self.seq_len}"
    x = ops.embed_fwd(ids, self.table.f32, self.pos.f32, n, L)
    enc_out = {} if collect else None
    xL, saved = self.enc.fwd(x, n, L, save, enc_out, drop=drop)
    if collect:
      out.update(enc_out)
    ctx = dict(n=n, L=L, ids=ids, enc=saved, xL=xL)
    T = n * L
    if collect:
      _, yf, _, _ = self.enc.norm.fwd(xL, T, D, want_bf16=False, want_f32=True)
      out["transformed"] = yf.view(n, L, D)
      # text_transformer.py:64 / :80: the token embedding before the position embedding, and the
      # weight-tied vocabulary logits embedding.attend(x) = x . table^T (fp32, diagnostics only)
      out["embedded"] = ops.embed_fwd(ids, self.table.f32, torch.zeros_like(self.pos.f32), n, L).view(n, L, D)
      V = self.table.f32.shape[0]
      vl = torch.empty((T, V), device=yf.device, dtype=F32)
      ops.sgemm(yf, D, 1, self.table.f32, 1, D, vl, T, V, D)
      out["vocab_logits"] = vl.view(n, L, V)
    if m.pool_type in ("last", "first"):
      off = L - 1 if m.pool_type == "last" else 0
      zb, z, mean, rstd = self.enc.norm.fwd(xL, n, D, row_stride=L, row_offset=off, want_f32=True)
      ctx.update(norm=(mean, rstd), off=off)
    elif m.pool_type in ("mean", "gap"):
      _, yf, mean, rstd = self.enc.norm.fwd(xL, T, D, want_bf16=False, want_f32=True)
      z = ops.pool_gap_fwd(yf, n, L, D)
      zb = ops.cast_bf16(z)
      ctx.update(norm=(mean, rstd))
    elif m.pool_type in ("max", "gmp"):       # text_transformer.py:89-90
      _, yf, mean, rstd = self.enc.norm.fwd(xL, T, D, want_bf16=False, want_f32=True)
      z, arg = ops.pool_max_fwd(yf, n, L, D)
      zb = ops.cast_bf16(z)
      ctx.update(norm=(mean, rstd), argmax=arg)
    elif m.pool_type == "map":
      y, _, mean, rstd = self.enc.norm.fwd(xL, T, D)
      z, msaved = self.map.fwd(y, n, L)
      zb = ops.cast_bf16(z)
      ctx.update(norm=(mean, rstd), map=msaved)
    else:
      raise NotImplementedError(f"Cannot do pooling '{m.pool_type}'")
    out["pre_logits"] = z
    x = z
    if self.head is not None:
      x = E.linear_fwd(zb, self.head[0], self.head[1], out_dtype=F32)
      out["logits"] = x
      ctx["head_in"] = zb
    return x, out, (ctx if save else None)

  def bwd(self, ctx, dx, on_block=None):
    m = self.m
    D = m.width
    n, L = ctx["n"], ctx["L"]
    T = n * L
    dz = dx.contiguous()
    if self.head is not None:
      dzb = ops.cast_bf16(dz)
      E.linear_bwd_w(ctx["head_in"], dzb, self.head[0], self.head[1], dy_for_bias=dz)
      dz = E.linear_bwd_x(dzb, self.head[0], out_dtype=F32)
    mean, rstd = ctx["norm"]
    xL = ctx["xL"]
    dxL_bf = torch.empty((T, D), device=xL.device, dtype=BF16)
    if m.pool_type in ("last", "first"):
      dxL = torch.zeros((T, D), device=xL.device, dtype=F32)
      dxL_bf.zero_()
      self.enc.norm.bwd(dz, xL, mean, rstd, n, D, dx=dxL, dx_bf16=dxL_bf, row_stride=L, row_offset=ctx["off"],
                        dx_colsum=self.enc.last_b2_grad(ctx["enc"]))
    elif m.pool_type in ("mean", "gap"):
      dyf = ops.pool_gap_bwd(dz, n, L, D)
      dxL = self.enc.norm.bwd(dyf, xL, mean, rstd, T, D, dx_bf16=dxL_bf, dx_colsum=self.enc.last_b2_grad(ctx["enc"]))
    elif m.pool_type in ("max", "gmp"):
      dyf = ops.pool_max_bwd(dz, ctx["argmax"], n, L, D)
      dxL = self.enc.norm.bwd(dyf, xL, mean, rstd, T, D, dx_bf16=dxL_bf, dx_colsum=self.enc.last_b2_grad(ctx["enc"]))
    else:
      dy = self.map.bwd(ctx["map"], dz, n, L)
      dxL = self.enc.norm.bwd(dy, xL, mean, rstd, T, D, dx_bf16=dxL_bf, dx_colsum=self.enc.last_b2_grad(ctx["enc"]))
    dx0, _ = self.enc.bwd(ctx["enc"], dxL, dxL_bf, n, L, b2_done=not self.enc.dropped(ctx["enc"]), on_block=on_block)
    if self.table.grad is not None:
      ops.embed_bwd(ctx["ids"].view(-1), dx0, self.table.grad)
    if self.pos.grad is not None:
      ops.batchsum(dx0, self.pos.grad, n, L, D)


class _Model:
  """Text transformer similar to CLIP (config holder + Flax-like init/apply)."""

  def __init__(self, num_classes, width=512, depth=12, mlp_dim=2048, num_heads=8, dropout=0.0,
               vocab_size=32_000, pool_type="last", scan=False, remat_policy="nothing_saveable",
               name=None):
    if not 0.0 <= float(dropout) < 1.0:
      raise ValueError(f"dropout must be in [0, 1), got {dropout}")
    self.dropout = float(dropout)
    if pool_type not in ("last", "first", "mean", "gap", "max", "gmp", "map"):
      raise NotImplementedError(f"Cannot do pooling '{pool_type}'")
    if width % num_heads or (width // num_heads) % 8 or width // num_heads > 128:
      raise NotImplementedError(f"attention kernels need a head_dim that is a multiple of 8 and <= 128 (64 is the "
                                f"fast path), got {width}/{num_heads}")
    self.num_classes, self.width, self.depth, self.mlp_dim = num_classes, width, depth, mlp_dim
    self.num_heads, self.vocab_size, self.pool_type, self.scan = num_heads, vocab_size, pool_type, scan
    self.name = name
    self._execs = {}

  def entries(self, prefix, seq_len):
    D, H, M = self.width, self.num_heads, self.mlp_dim
    # nn.Embed default init: variance_scaling(1.0, "fan_in", "normal", out_axis=0)
    ents = [Entry(f"{prefix}Embed_0/embedding", (self.vocab_size, D), E.init_normal(1 / math.sqrt(D))),
            Entry(f"{prefix}pos_embedding", (1, seq_len, D), E.init_normal(1 / math.sqrt(D)))]
    ents += E.encoder_entries(f"{prefix}Encoder_0", self.depth, D, H, M)
    if self.pool_type == "map":
      ents += E.map_entries(f"{prefix}MAPHead_0", D, H, M)
    if self.num_classes:
      ents += [Entry(f"{prefix}head/kernel", (D, self.num_classes), E.init_lecun_normal(D)),
               Entry(f"{prefix}head/bias", (self.num_classes,), E.init_zeros)]
    return ents

  def scan_prefixes(self, prefix=""):
    """Encoders presented with stacked blocks (scan=True)."""
    return (f"{prefix}Encoder_0",) if self.scan else ()

  def init(self, rng, text, **kw):
    del kw
    dev = text.device if torch.is_tensor(text) and text.is_cuda else torch.device("cuda", torch.cuda.current_device())
    store = ParamStore(self.entries("", text.shape[1]), dev, scan_prefixes=self.scan_prefixes())
    store.init_random(vit._seed_of(rng))
    store.refresh_shadow()
    return {"params": store.tree()}

  def executor(self, store, prefix, seq_len):
    key = (id(store), prefix, seq_len, getattr(store, "want_grads", False))
    if key not in self._execs:
      self._execs[key] = TextExec(self, store, prefix, seq_len)
    return self._execs[key]

  def apply(self, variables, text, *, train=False, rngs=None, collect=True, **kw):
    del kw
    params = variables["params"]
    if isinstance(params, ParamTree) and params.store is not None:
      store, prefix = params.store, params.prefix
    else:
      dev = torch.device("cuda", torch.cuda.current_device())
      store = adhoc_store(self._execs, ("txt", int(text.shape[1]), dev.index), params,
                          lambda: ParamStore(self.entries("", text.shape[1]), dev,
                                             scan_prefixes=self.scan_prefixes()))
      prefix = ""
    store.refresh_shadow()
    x, out, _ = self.executor(store, prefix, text.shape[1]).fwd(text, save=False, collect=collect,
                                                                drop=vit.dropout_for(self.dropout, train, rngs))
    return x, out


def Model(num_classes, *, variant=None, **kw):  # pylint: disable=invalid-name
  """Factory function (reference text_transformer.py:102-104)."""
  return _Model(num_classes, **{**vit.decode_variant(variant), **kw})


def load(init_params, init_file, model_cfg, dont_load=()):  # pylint: disable=invalid-name
  """Load init from checkpoint (text_transformer.py:107-119)."""
  params = utils.load_params(init_file)
  params = utils.tree_map(lambda x: x, params)
  extra_posemb = params["Encoder_0"].pop("pos_embedding", 0)
  params["pos_embedding"] = params["pos_embedding"] + extra_posemb
  if init_params:
    want_scan = "encoderblock" in init_params.get("Encoder_0", {})
  else:
    want_scan = bool((model_cfg or {}).get("scan", False))
  have_scan = "encoderblock" in params["Encoder_0"]
  if have_scan and not want_scan:
    params["Encoder_0"] = vit.scan_to_pyloop({"Transformer": params["Encoder_0"]})["Transformer"]
  elif want_scan and not have_scan:
    params["Encoder_0"] = vit.pyloop_to_scan({"Transformer": params["Encoder_0"]})["Transformer"]
  return common.merge_params(params, init_params, dont_load)
