"""BERT text tower on gfx950 kernels behind `models.proj.flaxformer.bert`.

Mirrors big_vision/models/proj/flaxformer/bert.py: `Model(config, num_classes=None,
head_zeroinit=True)` (:33-38) returning `(x, out)` with `out["transformed"]`, `out["pre_logits"]`
(the CLS token, :59) and `out["logits"]` (:61-63); `load(params, path, model_cfg, dont_load)` (:67-94).
This is the text tower of the literal LiT config (configs/proj/image_text/siglip_lit_coco.py:78,84-87:
`text_model='proj.flaxformer.bert'`, `text=dict(config='base')`).

The encoder itself is flaxformer's `BertEncoder` - an un-vendored, un-pinned dependency of the
reference (requirements.txt) - i.e. the original BERT encoder: token + position + segment embeddings ->
LayerNorm(eps 1e-12) -> POST-LayerNorm blocks.  bert.py:50-57 fixes its inputs: position_ids = arange,
segment_ids = 0, input_mask = (text != 0).  PARITY UNPINNED against the reference's own arithmetic and
parameter NAMES (both live in flaxformer); the arithmetic is pinned to HuggingFace `BertModel(gelu_new)`
through the CPU restatement of the test infrastructure (`bert_forward`, tests/test_oracle.py), the tree layout below follows
flaxformer's `bert.py` / `bert_checkpoint_converter.py` as published: BertEncoder_0/{embedder/
embedders_{token,position,segment}_ids/embedding, layer_norm, encoder_block_<i>/{attention_block/
{attention_layer/{query,key,value,out}, layer_norm}, mlp_block/{mlp/{wi,wo}, layer_norm}}}.  `load`
goes through `common.merge_params`, which refuses a tree that does not match, naming the differences.

Every block runs on the kernels of the pre-LN ViT blocks (engine.py), re-ordered for post-LN:
  forward   x -> QKV GEMM -> attention (key-padding length from input_mask) -> out-proj GEMM (+bias +x:
            BV_EPI_RESIDUAL) -> LayerNorm (bf16 copy for the GEMMs + fp32 for the residual) -> wi GEMM
            (GELU epilogue) -> wo GEMM (+bias +residual) -> LayerNorm
  backward  LayerNorm bwd (fp32 cotangent in, fp32 + bf16 out, bias column sums fused) -> MLP backward whose
            last dX GEMM adds the residual branch in its epilogue (fp32 out) -> LayerNorm bwd -> out-proj
            dW / dX -> attention backward -> QKV dW / dX (+ residual branch)
Padded positions: the reference also masks padded QUERIES (make_attention_mask(input_mask, input_mask));
their rows never reach the CLS output or any gradient, the kernels leave them attending to the valid
keys - `out["transformed"]` differs from the reference at padded positions only.
"""
from __future__ import annotations

import torch

from big_vision_amd import engine as E
from big_vision_amd import ops
from big_vision_amd import utils
from big_vision_amd.models import common
from big_vision_amd.models import vit
from big_vision_amd.params import Entry, ParamStore, ParamTree, adhoc_store

BF16, F32 = torch.bfloat16, torch.float32
LN_EPS = 1e-12   # flaxformer/architectures/bert: layer norm epsilon of BERT

# flaxformer/architectures/bert/configs.py: BertBaseConfig / BertLargeConfig
CONFIGS = {
    "base": dict(hidden_size=768, intermediate_dim=3072, num_hidden_layers=12, num_attention_heads=12,
                 vocab_size=30522, max_length=512, num_segments=2),
    "large": dict(hidden_size=1024, intermediate_dim=4096, num_hidden_layers=24, num_attention_heads=16,
                  vocab_size=30522, max_length=512, num_segments=2),
}


class _PostLNBlock:
  """One flaxformer BERT encoder block (AttentionBlock + MlpBlock, post-LayerNorm)."""

  def __init__(self, store, P, D, H, M):
    A = f"{P}/attention_block/attention_layer"
    self.D, self.H = D, H
    self.wqkv = E._W(store, f"{A}/qkv/kernel", (D, 3 * D))
    self.bqkv = E._W(store, f"{A}/qkv/bias", (3 * D,))
    self.wo = E._W(store, f"{A}/out/kernel", (D, D))
    self.bo = E._W(store, f"{A}/out/bias")
    self.ln1 = E.LN(store, f"{P}/attention_block/layer_norm")
    self.ln2 = E.LN(store, f"{P}/mlp_block/layer_norm")
    self.mlp = E.MLP.__new__(E.MLP)
    self.mlp.w1 = E._W(store, f"{P}/mlp_block/mlp/wi/kernel"); self.mlp.b1 = E._W(store, f"{P}/mlp_block/mlp/wi/bias")
    self.mlp.w2 = E._W(store, f"{P}/mlp_block/mlp/wo/kernel"); self.mlp.b2 = E._W(store, f"{P}/mlp_block/mlp/wo/bias")
    self.mlp.M = M

  def fwd(self, xb, xf, n, L, lens):
    """xb / xf: the block input as bf16 GEMM operand and as fp32 residual (both come out of the previous
    LayerNorm kernel).  Returns (x2b, x2f, saved)."""
    T, D, H = n * L, self.D, self.H
    qkv = E.linear_fwd(xb, self.wqkv, self.bqkv, out_dtype=BF16)
    o, lse = ops.attn_fwd(qkv, n, L, H, kv_len=lens)
    a = E.linear_fwd(o, self.wo, self.bo, out_dtype=F32, epilogue=ops.EPI_RESIDUAL, aux=xf)
    x1b, x1f, m1, r1 = self.ln1.fwd(a, T, D, want_f32=True, eps=LN_EPS)
    m, hd, g = self.mlp.fwd(x1b, x1f)
    x2b, x2f, m2, r2 = self.ln2.fwd(m, T, D, want_f32=True, eps=LN_EPS)
    return x2b, x2f, (xb, qkv, o, lse, a, m1, r1, x1b, hd, g, m, m2, r2)

  def bwd(self, saved, dx2, n, L, lens):
    """dx2: fp32 cotangent of this block's output (GEMM-operand and residual uses summed).  Returns the
    fp32 cotangent of the block input."""
    xb, qkv, o, lse, a, m1, r1, x1b, hd, g, m, m2, r2 = saved
    T, D, H = n * L, self.D, self.H
    dev = dx2.device
    dm_bf = torch.empty((T, D), device=dev, dtype=BF16)
    dm = self.ln2.bwd(dx2, m, m2, r2, T, D, dx_bf16=dm_bf, dx_colsum=self.mlp.b2.grad)   # wo bias grad = colsum(dm)
    # gradient w.r.t. x1 = MLP branch (the last dX GEMM) + residual branch dm, summed in that GEMM's epilogue
    dx1 = self.mlp.bwd(dm, dm_bf, x1b, hd, g, bias2_done=True,
                       dx_kw=dict(out_dtype=F32, epilogue=ops.EPI_RESIDUAL, aux=dm))
    da_bf = torch.empty((T, D), device=dev, dtype=BF16)
    da = self.ln1.bwd(dx1, a, m1, r1, T, D, dx_bf16=da_bf, dx_colsum=self.bo.grad)      # out-proj bias grad
    E.linear_bwd_w(o, da_bf, self.wo, None)
    d_o = E.linear_bwd_x(da_bf, self.wo)
    dqkv = ops.attn_bwd(qkv, o, d_o, lse, n, L, H, dbias=self.bqkv.grad, kv_len=lens)
    E.linear_bwd_w(xb, dqkv, self.wqkv, None)
    return E.linear_bwd_x(dqkv, self.wqkv, out_dtype=F32, epilogue=ops.EPI_RESIDUAL, aux=da)


class BertExec:
  """Forward / backward of the BERT tower bound to a ParamStore at `prefix`."""

  def __init__(self, m: "_Model", store: ParamStore, prefix: str, seq_len: int):
    self.m, self.store, self.seq_len = m, store, seq_len
    c = m.cfg
    D, H, M = c["hidden_size"], c["num_attention_heads"], c["intermediate_dim"]
    B = f"{prefix}BertEncoder_0"
    self.tok = E._W(store, f"{B}/embedder/embedders_token_ids/embedding")
    self.pos = E._W(store, f"{B}/embedder/embedders_position_ids/embedding")
    self.seg = E._W(store, f"{B}/embedder/embedders_segment_ids/embedding")
    self.ln = E.LN(store, f"{B}/layer_norm")
    self.blocks = [_PostLNBlock(store, f"{B}/encoder_block_{i}", D, H, M) for i in range(c["num_hidden_layers"])]
    self.head = None
    if m.num_classes:
      self.head = (E._W(store, f"{prefix}head/kernel"), E._W(store, f"{prefix}head/bias"))
    self._mask_checked = False

  def _lengths(self, ids):
    """input_mask = (text != 0) (bert.py:55) as a key-padding LENGTH per sample: the attention kernels mask a
    suffix, so the non-pad tokens must be a prefix (what tokenize-then-pad produces).  EVERY batch is checked
    (advisor r3: a later batch with an inner pad id or an all-pad row must not slip through), without stalling
    the launch queue: the verdict of batch k is computed on the device, copied to pinned host memory
    asynchronously and read by `check_pending()` - which the callers that own a step run BEFORE anything is
    committed: the SigLIP / contrastive trainers right before `opt.step()` (TwoTowersExec.check_inputs: by then the
    copy of this step's first kernels has long completed, and a refused batch leaves the weights untouched),
    `two_towers.Model.apply` and `Model.apply` of this module before they return (the evaluators' predict_fn) - and,
    as a backstop, when the next batch arrives.  lens is clamped to >= 1 so that a forward that is later refused
    never runs an attention row over zero keys."""
    self.check_pending()
    valid = ids != 0
    lens = valid.sum(dim=1).to(torch.int32)
    L = ids.shape[1]
    prefix = torch.arange(L, device=ids.device)[None, :] < lens[:, None]
    bad = torch.stack([(valid != prefix).any(), (lens < 1).any()]).to(torch.uint8)
    if ids.is_cuda:
      host = torch.empty(2, dtype=torch.uint8, pin_memory=True)
      host.copy_(bad, non_blocking=True)
      ev = torch.cuda.Event()
      ev.record()
      self._pending = (host, ev)
    else:
      self._pending = (bad.clone(), None)
    if not self._mask_checked:   # the very first batch is checked synchronously (nothing is queued yet)
      self.check_pending()
      self._mask_checked = True
    return lens.clamp_(min=1).contiguous()

  def check_pending(self):
    """Raises if the last batch handed to fwd() had an input_mask the kernels cannot express."""
    pend, self._pending = getattr(self, "_pending", None), None
    if pend is None:
      return
    host, ev = pend
    if ev is not None:
      ev.synchronize()
    if int(host[0]):
      raise NotImplementedError("BERT input_mask must mark a prefix of the sequence (pad id 0 at the end only)")
    if int(host[1]):
      raise ValueError("an example without any token (all ids are 0)")

  def fwd(self, text, save=False, collect=False):
    m = self.m
    D = m.cfg["hidden_size"]
    ids = text.to(torch.int32).contiguous()
    n, L = ids.shape
    assert L <= self.pos.f32.shape[0], f"text length {L} > max_length {self.pos.f32.shape[0]}"
    T = n * L
    lens = self._lengths(ids)
    # position_ids = arange(L), segment_ids = 0 (bert.py:52-54): one [L, D] additive table
    pos_eff = (self.pos.f32[:L] + self.seg.f32[0]).contiguous()
    e = ops.embed_fwd(ids, self.tok.f32, pos_eff, n, L)
    xb, xf, mean0, rstd0 = self.ln.fwd(e, T, D, want_f32=True, eps=LN_EPS)
    saved = []
    for blk in self.blocks:
      xb, xf, s = blk.fwd(xb, xf, n, L, lens)
      if save:
        saved.append(s)
    out = {}
    z = xf.view(n, L, D)[:, 0].contiguous()          # CLS token (bert.py:59)
    if collect:
      out["transformed"] = xf.view(n, L, D)
    out["pre_logits"] = z
    x = z
    ctx = dict(n=n, L=L, ids=ids, lens=lens, emb=(e, mean0, rstd0), blocks=saved)
    if self.head is not None:
      zb = xb.view(n, L, D)[:, 0].contiguous()
      x = E.linear_fwd(zb, self.head[0], self.head[1], out_dtype=F32)
      out["logits"] = x
      ctx["head_in"] = zb
    return x, out, (ctx if save else None)

  def bwd(self, ctx, dx, on_block=None):
    D = self.m.cfg["hidden_size"]
    n, L, lens = ctx["n"], ctx["L"], ctx["lens"]
    T = n * L
    dz = dx.contiguous()
    if self.head is not None:
      dzb = ops.cast_bf16(dz)
      E.linear_bwd_w(ctx["head_in"], dzb, self.head[0], self.head[1], dy_for_bias=dz)
      dz = E.linear_bwd_x(dzb, self.head[0], out_dtype=F32)
    dxl = torch.zeros((n, L, D), device=dz.device, dtype=F32)   # only the CLS rows carry a cotangent
    dxl[:, 0] = dz
    dxl = dxl.view(T, D)
    for i in range(len(self.blocks) - 1, -1, -1):
      dxl = self.blocks[i].bwd(ctx["blocks"][i], dxl, n, L, lens)
      if on_block is not None:
        on_block(i)
    e, mean0, rstd0 = ctx["emb"]
    de = self.ln.bwd(dxl, e, mean0, rstd0, T, D)
    if self.tok.grad is not None:
      ops.embed_bwd(ctx["ids"].view(-1), de, self.tok.grad)
    if self.pos.grad is not None or self.seg.grad is not None:
      dpos = torch.zeros((L, D), device=de.device, dtype=F32)
      ops.batchsum(de, dpos, n, L, D)
      if self.pos.grad is not None:
        self.pos.grad[:L] += dpos
      if self.seg.grad is not None:
        self.seg.grad[0] += dpos.sum(0)


class _Model:
  """BERT encoder with linear projection on the last layer's CLS token (bert.py:33-64)."""

  def __init__(self, config, num_classes=None, head_zeroinit=True, name=None):
    self.cfg = dict(CONFIGS[config]) if isinstance(config, str) else dict(config)   # a dict: test-sized encoders
    self.config, self.num_classes, self.head_zeroinit, self.name = config, num_classes, head_zeroinit, name
    D, H = self.cfg["hidden_size"], self.cfg["num_attention_heads"]
    if D % H or (D // H) % 8 or D // H > 128:
      raise NotImplementedError(f"attention kernels need a head_dim that is a multiple of 8 and <= 128, got {D}/{H}")
    self.depth = 0          # (no `Encoder_0/encoderblock_<i>` ranges: two_towers syncs this tower's gradients in one piece)
    self._execs = {}

  def entries(self, prefix, seq_len):
    del seq_len   # the position table always has max_length rows (bert.py:78-88 crops a checkpoint's to it)
    c = self.cfg
    D, H, M = c["hidden_size"], c["num_attention_heads"], c["intermediate_dim"]
    B = f"{prefix}BertEncoder_0"
    init = E.init_normal(0.02)
    ents = [Entry(f"{B}/embedder/embedders_token_ids/embedding", (c["vocab_size"], D), init),
            Entry(f"{B}/embedder/embedders_position_ids/embedding", (c["max_length"], D), init),
            Entry(f"{B}/embedder/embedders_segment_ids/embedding", (c["num_segments"], D), init)]
    ents += E.ln_entries(f"{B}/layer_norm")(D)
    for i in range(c["num_hidden_layers"]):
      P = f"{B}/encoder_block_{i}"
      ents += E.mha_entries(f"{P}/attention_block/attention_layer", D, H, "qkv")
      ents += E.ln_entries(f"{P}/attention_block/layer_norm")(D)
      ents += [Entry(f"{P}/mlp_block/mlp/wi/kernel", (D, M), init), Entry(f"{P}/mlp_block/mlp/wi/bias", (M,), E.init_zeros),
               Entry(f"{P}/mlp_block/mlp/wo/kernel", (M, D), init), Entry(f"{P}/mlp_block/mlp/wo/bias", (D,), E.init_zeros)]
      ents += E.ln_entries(f"{P}/mlp_block/layer_norm")(D)
    if self.num_classes:
      kinit = E.init_zeros if self.head_zeroinit else E.init_lecun_normal(D)
      ents += [Entry(f"{prefix}head/kernel", (D, self.num_classes), kinit),
               Entry(f"{prefix}head/bias", (self.num_classes,), E.init_zeros)]
    return ents

  def scan_prefixes(self, prefix=""):
    return ()

  def init(self, rng, text, **kw):
    del kw
    dev = text.device if torch.is_tensor(text) and text.is_cuda else torch.device("cuda", torch.cuda.current_device())
    store = ParamStore(self.entries("", text.shape[1]), dev)
    store.init_random(vit._seed_of(rng))
    store.refresh_shadow()
    return {"params": store.tree()}

  def executor(self, store, prefix, seq_len):
    key = (id(store), prefix, seq_len, getattr(store, "want_grads", False))
    if key not in self._execs:
      self._execs[key] = BertExec(self, store, prefix, seq_len)
    return self._execs[key]

  def apply(self, variables, text, *, train=False, rngs=None, collect=True, **kw):
    del rngs, train, kw
    params = variables["params"]
    if isinstance(params, ParamTree) and params.store is not None:
      store, prefix = params.store, params.prefix
    else:
      dev = torch.device("cuda", torch.cuda.current_device())
      store = adhoc_store(self._execs, ("bert", int(text.shape[1]), dev.index), params,
                          lambda: ParamStore(self.entries("", text.shape[1]), dev))
      prefix = ""
    store.refresh_shadow()
    ex = self.executor(store, prefix, text.shape[1])
    x, out, _ = ex.fwd(text, save=False, collect=collect)
    ex.check_pending()        # the verdict on THIS batch's input_mask, before its outputs are handed out
    return x, out


def Model(config, num_classes=None, head_zeroinit=True, **kw):  # pylint: disable=invalid-name
  """bert.py:33-38 (a flax dataclass there: same field names)."""
  return _Model(config, num_classes=num_classes, head_zeroinit=head_zeroinit, **kw)


def load(params, path, model_cfg=None, dont_load=()):
  """Returns `params` with BERT weights replaced from the checkpoint at `path` (bert.py:67-94).

  The reference first looks for an ORIGINAL TensorFlow BERT checkpoint (`<path>/bert_model.ckpt.index`) and
  converts it with flaxformer's bert_checkpoint_converter; neither TensorFlow nor flaxformer exists here,
  so that branch raises.  Otherwise `path` is a big_vision checkpoint (npz[:subtree]) in the layout above."""
  del model_cfg
  import os
  if isinstance(path, str) and os.path.exists(f"{path}/bert_model.ckpt.index"):
    raise NotImplementedError(
        f"'{path}/bert_model.ckpt' is an original TensorFlow BERT checkpoint: converting it needs tensorflow and "
        "flaxformer.architectures.bert.bert_checkpoint_converter (bert.py:72-88); convert it with the reference "
        "once and point model_init at the resulting big_vision checkpoint")
  restored = utils.load_params(path)
  if params and "BertEncoder_0" in restored:
    # bert.py:81-88 crops the position table of a longer checkpoint to the model's max_length
    want = params["BertEncoder_0"]["embedder"]["embedders_position_ids"]["embedding"].shape[0]
    tab = restored["BertEncoder_0"]["embedder"]["embedders_position_ids"]["embedding"]
    if tab.shape[0] > want:
      restored["BertEncoder_0"]["embedder"]["embedders_position_ids"]["embedding"] = tab[:want]
  return common.merge_params(restored, params, dont_load)
