"""Tensor-level wrappers over the libbvhip C ABI (include/bvhip.h).

PyTorch is used only for device memory and the HIP stream; every function here
enqueues hand-written HIP kernels on torch's current stream.  There is no
eager/PyTorch fallback: non-GPU tensors raise.
"""
import numpy as np
import torch

from big_vision_amd import _lib
from big_vision_amd._lib import (EPI_NONE, EPI_RESIDUAL, EPI_POS, EPI_GELU, EPI_GELU_BWD,
                                 EPI_ATOMIC, EPI_GELU_BWD_EMIT, EPI_GELU_GD, EPI_MUL, EPI_GELU_G)

BF16 = torch.bfloat16
F32 = torch.float32


def _stream():
  return torch.cuda.current_stream().cuda_stream


def _p(t):
  return None if t is None else t.data_ptr()


class ShiftedBase:
  """A device buffer that holds elements [shift, shift + numel) of a larger index space: `data_ptr()` is the address
  element 0 of that space WOULD have.  For kernels that address operands through a table of absolute offsets
  (bv_adafactor_step) when a rank holds its own run of the buffer only; the caller's table must stay inside the run."""

  def __init__(self, t, shift):
    self.t, self.shift, self.dtype = t, int(shift), t.dtype

  def data_ptr(self):
    return self.t.data_ptr() - self.shift * self.t.element_size()


def _chk(t, dtype, name):
  if not t.is_cuda:
    raise RuntimeError(f"{name}: expected a GPU tensor (libbvhip has no CPU path)")
  if t.dtype != dtype:
    raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
  return t


def _rowmajor2d(t, name):
  if t.dim() != 2 or t.stride(1) != 1:
    raise ValueError(f"{name}: expected a 2-D tensor with contiguous rows, got {tuple(t.shape)} strides {t.stride()}")
  return t.stride(0)


WORKSPACE_BYTES = 64 << 20   # 256 split-K partial tiles of 256 KiB (one per CU)


class Context:
  """The caller's `bv_ctx` (include/bvhip.h "Context"): kernel-variant options, the split-K workspace of the
  weight-gradient GEMMs and the launch counters.  libbvhip keeps no process-global state; the Python host keeps
  ONE context per (device, stream) it launches GEMMs on (`ctx()`), so two streams never share a split-K slab."""

  def __init__(self):
    self.ptr = _lib.load().bv_ctx_create()
    if not self.ptr:
      raise RuntimeError("bv_ctx_create failed")
    self._ws = None
    import os
    if os.environ.get("BV_CTX_OPTS"):        # A/B runs of whole programs: "gemm_nt=1,gemm_group_n=4" applied to every new context
      for kv in os.environ["BV_CTX_OPTS"].split(","):
        k, v = kv.split("=")
        self.set(k.strip(), int(v))
    self.use_workspace = True   # False: weight-gradient GEMMs combine their split-K partials with fp32 atomics (A/B)

  def __del__(self):
    try:
      if self.ptr:
        _lib.load().bv_ctx_destroy(self.ptr)
    except Exception:   # interpreter shutdown
      pass

  def set(self, name, value):
    """Sets option `name` (_lib.OPTS), returns the previous value."""
    old = _lib.load().bv_ctx_set(self.ptr, _lib.OPTS[name], int(value))
    if old < 0:
      raise RuntimeError(f"bv_ctx_set({name}): {_lib.load().bv_last_error().decode()}")
    return old

  def get(self, name):
    return _lib.load().bv_ctx_get(self.ptr, _lib.OPTS[name])

  def copy_options_from(self, other):
    """Every option of `other` (the settable ids of _lib.OPTS, i.e. below 100: the rest are counters) and its
    workspace policy: a side stream's context runs the configuration of the stream it was forked from."""
    if other is self:
      return
    for name, opt_id in _lib.OPTS.items():
      if opt_id < 100:
        v = other.get(name)
        if v != self.get(name):
          self.set(name, v)
    self.use_workspace = other.use_workspace

  def ensure_workspace(self, device):
    if not self.use_workspace:
      if self._ws is not None:
        self._ws = None
        _lib.call("bv_ctx_set_workspace", self.ptr, None, 0)
      return
    if self._ws is None or self._ws.device != device:
      self._ws = torch.empty(WORKSPACE_BYTES, device=device, dtype=torch.uint8)
      _lib.call("bv_ctx_set_workspace", self.ptr, self._ws.data_ptr(), WORKSPACE_BYTES)


_contexts = {}


def ctx() -> Context:
  """The context of the current (device, stream)."""
  key = (torch.cuda.current_device() if torch.cuda.is_available() else -1, _stream())
  c = _contexts.get(key)
  if c is None:
    c = _contexts[key] = Context()
  return c


def ctx_set(name, value):
  """A/B tools and tests: option `name` of the current stream's context; returns the previous value."""
  return ctx().set(name, value)


def ctx_get(name):
  return ctx().get(name)


class option:
  """`with ops.option("gemm_roll", 0): ...` - an option of the current stream's context for the duration of a block."""

  def __init__(self, name, value):
    self.name, self.value = name, value

  def __enter__(self):
    self.c = ctx()
    self.old = self.c.set(self.name, self.value)
    return self

  def __exit__(self, *exc):
    self.c.set(self.name, self.old)


# ------------------------------------------------------------------- GEMMs --
def gemm(a, b, *, a_kmajor=True, b_kmajor=False, out=None, out_dtype=BF16, M=None, N=None, K=None,
         epilogue=EPI_NONE, bias=None, aux=None, aux_rows=0, out2=None, alpha=1.0, split_k=0,
         colsum=None):
  """C[M,N] = A(MxK) B(KxN) with bf16 inputs; see bv_gemm_bf16 in include/bvhip.h.
  colsum (fp32 [N], GELU_BWD / MUL epilogues): += column sums of C (bv_gemm_bf16_colsum).

  a: [M,K] if a_kmajor else [K,M];  b: [N,K] if b_kmajor else [K,N].
  """
  _chk(a, BF16, "gemm.a"); _chk(b, BF16, "gemm.b")
  lda = _rowmajor2d(a, "gemm.a"); ldb = _rowmajor2d(b, "gemm.b")
  m, k = (a.shape if a_kmajor else (a.shape[1], a.shape[0]))
  n, k2 = (b.shape if b_kmajor else (b.shape[1], b.shape[0]))
  if k != k2:
    raise ValueError(f"gemm: reduction dims differ: {k} vs {k2}")
  M = m if M is None else M; N = n if N is None else N; K = k if K is None else K
  if out is None:
    out = torch.empty((M, N), device=a.device, dtype=out_dtype)
  ldc = _rowmajor2d(out, "gemm.out")
  if out.dtype not in (BF16, F32):
    raise TypeError("gemm.out must be bf16 or fp32")
  ldaux = 0
  if aux is not None:
    ldaux = _rowmajor2d(aux, "gemm.aux")
  if bias is not None:
    _chk(bias, F32, "gemm.bias")
  c = ctx()
  if epilogue == EPI_ATOMIC:
    c.ensure_workspace(a.device)
  if colsum is not None:
    _chk(colsum, F32, "gemm.colsum")
    _lib.call("bv_gemm_bf16_colsum", int(a_kmajor), int(b_kmajor), _p(a), lda, _p(b), ldb, _p(out), ldc,
              int(out.dtype == F32), M, N, K, epilogue, _p(bias), _p(aux), ldaux, aux_rows, _p(out2),
              float(alpha), split_k, _p(colsum), _stream(), c.ptr)
    return out
  _lib.call("bv_gemm_bf16", int(a_kmajor), int(b_kmajor), _p(a), lda, _p(b), ldb, _p(out), ldc,
            int(out.dtype == F32), M, N, K, epilogue, _p(bias), _p(aux), ldaux, aux_rows, _p(out2),
            float(alpha), split_k, _stream(), c.ptr)
  return out


def sgemm(a, sam, sak, b, sbk, sbn, out, M, N, K, alpha=1.0, beta=0.0, log_alpha=None):
  """fp32 strided GEMM (bv_sgemm_strided); alpha is multiplied by exp(log_alpha[0]) (device)."""
  _chk(a, F32, "sgemm.a"); _chk(b, F32, "sgemm.b"); _chk(out, F32, "sgemm.out")
  _lib.call("bv_sgemm_strided", _p(a), sam, sak, _p(b), sbk, sbn, _p(out), out.stride(0), M, N, K,
            float(alpha), float(beta), _p(log_alpha), _stream(), ctx().ptr)
  return out


# --------------------------------------------------------------- LayerNorm --
def layernorm_fwd(x, scale, bias, *, rows, D, row_stride=1, row_offset=0, want_bf16=True,
                  want_f32=False, eps=1e-6):
  """x fp32, or bf16 on a bf16 residual stream (bv_layernorm_fwd_bf16x)."""
  _chk(x, x.dtype if x.dtype in (F32, BF16) else F32, "layernorm.x")
  _chk(scale, F32, "layernorm.scale"); _chk(bias, F32, "layernorm.bias")
  dev = x.device
  y_bf = torch.empty((rows, D), device=dev, dtype=BF16) if want_bf16 else None
  y_f = torch.empty((rows, D), device=dev, dtype=F32) if want_f32 else None
  mean = torch.empty((rows,), device=dev, dtype=F32)
  rstd = torch.empty((rows,), device=dev, dtype=F32)
  _lib.call("bv_layernorm_fwd_bf16x" if x.dtype == BF16 else "bv_layernorm_fwd", _p(x), _p(scale), _p(bias),
            _p(y_bf), _p(y_f), _p(mean), _p(rstd), rows, D, row_stride, row_offset, float(eps), _stream())
  return y_bf, y_f, mean, rstd


def layernorm_bwd(dy, x, scale, mean, rstd, *, rows, D, dres=None, dx=None, dx_bf16=None,
                  dscale=None, dbias=None, dx_colsum=None, row_stride=1, row_offset=0, bias=None, y_out=None):
  """dx = dres + LN_bwd(dy); dscale/dbias (and dx_colsum += column sums of dx) accumulated in place.
  y_out (bf16 [rows, D], fp32 x only, needs the LayerNorm `bias`): also re-emits the forward's output
  (bv_layernorm_bwd_y)."""
  if dy.dtype not in (BF16, F32):
    raise TypeError("layernorm_bwd.dy must be bf16 or fp32")
  if x.dtype == BF16:
    # bf16 residual stream: dres and dx are bf16 and dx IS the bf16 copy the next GEMM reads; `dx`
    # (the fp32 stream of the other mode) is not written.  Strided calls start from zeros.
    _chk(x, BF16, "layernorm_bwd.x")
    if y_out is not None:
      # only the fp32-stream kernel re-emits the forward output (bv_layernorm_bwd_y); silently returning an
      # uninitialised y_out buffer here was advisor finding r3 #5
      raise ValueError("layernorm_bwd: y_out is not supported on the bf16 residual stream (re-normalise with layernorm_fwd)")
    if dres is not None:
      _chk(dres, BF16, "layernorm_bwd.dres")
    if dx_bf16 is None:
      dx_bf16 = torch.empty_like(x) if row_stride == 1 else torch.zeros_like(x)
    _lib.call("bv_layernorm_bwd_bf16x", _p(dy), int(dy.dtype == F32), _p(x), _p(scale), _p(mean), _p(rstd),
              _p(dres), _p(dx_bf16), _p(dscale), _p(dbias), _p(dx_colsum), rows, D, row_stride, row_offset,
              _stream())
    return dx_bf16
  _chk(x, F32, "layernorm_bwd.x")
  if dx is None:
    dx = torch.empty_like(x) if row_stride == 1 else torch.zeros_like(x)
  if y_out is not None:
    _chk(y_out, BF16, "layernorm_bwd.y_out"); _chk(bias, F32, "layernorm_bwd.bias")
    assert row_stride == 1 and y_out.is_contiguous()
    _lib.call("bv_layernorm_bwd_y", _p(dy), int(dy.dtype == F32), _p(x), _p(scale), _p(mean), _p(rstd),
              _p(dres), _p(dx), _p(dx_bf16), _p(dscale), _p(dbias), _p(dx_colsum), rows, D, row_stride,
              row_offset, _p(bias), _p(y_out), _stream())
    return dx
  _lib.call("bv_layernorm_bwd", _p(dy), int(dy.dtype == F32), _p(x), _p(scale), _p(mean), _p(rstd),
            _p(dres), _p(dx), _p(dx_bf16), _p(dscale), _p(dbias), _p(dx_colsum), rows, D, row_stride,
            row_offset,
            _stream())
  return dx


# --------------------------------------------------------------- Attention --
def _head_dim(numel, rows, parts, H, what):
  Dh, rem = divmod(numel, rows * parts * H)
  assert rem == 0 and Dh % 8 == 0 and 8 <= Dh <= 128, f"{what}: head dim {numel}/({rows}*{parts}*{H}) unsupported"
  return Dh


def attn_fwd(qkv, n, L, H, kv_len=None):
  """qkv [n*L, 3*H*Dh] (Dh inferred; 64 takes the LDS-resident kernels, other multiples of 8 up to 128
  and sequences longer than 576 the general ones).  kv_len (int32 [n], optional): valid keys per sample
  (key-padding mask of the NaFlex tower)."""
  _chk(qkv, BF16, "attn.qkv")
  assert qkv.is_contiguous()
  Dh = _head_dim(qkv.numel(), n * L, 3, H, "attn_fwd")
  o = torch.empty((n * L, H * Dh), device=qkv.device, dtype=BF16)
  lse = torch.empty((n, H, L), device=qkv.device, dtype=F32)
  if kv_len is not None:
    _chk(kv_len, torch.int32, "attn.kv_len")
    assert kv_len.is_contiguous() and kv_len.numel() == n
  if Dh != 64 or L > 576:
    _lib.call("bv_attn_fwd_dh", _p(qkv), _p(o), _p(lse), _p(kv_len), n, L, H, Dh, _stream())
  elif kv_len is not None:
    _lib.call("bv_attn_fwd_masked", _p(qkv), _p(o), _p(lse), _p(kv_len), n, L, H, _stream(), ctx().ptr)
  else:
    _lib.call("bv_attn_fwd", _p(qkv), _p(o), _p(lse), n, L, H, _stream(), ctx().ptr)
  return o, lse


def attn_bwd(qkv, o, d_o, lse, n, L, H, dqkv=None, dbias=None, kv_len=None):
  """dbias (fp32, 3*H*Dh elements): += column sums of dqkv (the q/k/v bias gradients)."""
  _chk(qkv, BF16, "attn.qkv"); _chk(o, BF16, "attn.o"); _chk(d_o, BF16, "attn.do")
  assert d_o.is_contiguous() and o.is_contiguous()
  Dh = _head_dim(qkv.numel(), n * L, 3, H, "attn_bwd")
  if dqkv is None:
    dqkv = torch.empty_like(qkv)
  if dbias is not None:
    _chk(dbias, F32, "attn.dbias")
    assert dbias.is_contiguous() and dbias.numel() == 3 * H * Dh
  delta = torch.empty((n, H, L), device=qkv.device, dtype=F32)
  rows = torch.empty((n, 3 * H * Dh), device=qkv.device, dtype=F32) if dbias is not None else None
  if kv_len is not None:
    _chk(kv_len, torch.int32, "attn.kv_len")
  if Dh != 64 or L > 576:
    _lib.call("bv_attn_bwd_dh", _p(qkv), _p(d_o), _p(lse), _p(kv_len), _p(delta), _p(dqkv), _p(rows),
              n, L, H, Dh, _stream())
  elif kv_len is not None:
    _lib.call("bv_attn_bwd_masked", _p(qkv), _p(d_o), _p(lse), _p(kv_len), _p(delta), _p(dqkv), _p(rows),
              n, L, H, _stream(), ctx().ptr)
  else:
    _lib.call("bv_attn_bwd", _p(qkv), _p(o), _p(d_o), _p(lse), _p(delta), _p(dqkv), _p(rows), n, L, H,
              _stream(), ctx().ptr)
  if dbias is not None:
    colsum(rows, dbias)   # per-sample sums (written by the kernels) -> bias gradient
  return dqkv


def map_attn_fwd(q, kv, n, L, H, kv_len=None):
  """q [n, H*Dh], kv [n*L, 2*H*Dh].  kv_len (int32 [n], optional): valid keys per sample (NaFlex pool mask)."""
  _chk(q, BF16, "map_attn.q"); _chk(kv, BF16, "map_attn.kv")
  assert q.is_contiguous() and kv.is_contiguous()
  Dh = _head_dim(q.numel(), n, 1, H, "map_attn_fwd")
  o = torch.empty((n, H * Dh), device=q.device, dtype=BF16)
  p = torch.empty((n, H, L), device=q.device, dtype=F32)
  if kv_len is not None:
    _chk(kv_len, torch.int32, "map_attn.kv_len")
  if Dh != 64 or L > 2048:
    _lib.call("bv_map_attn_fwd_dh", _p(q), _p(kv), _p(o), _p(p), _p(kv_len), n, L, H, Dh, _stream())
  elif kv_len is not None:
    _lib.call("bv_map_attn_fwd_masked", _p(q), _p(kv), _p(o), _p(p), _p(kv_len), n, L, H, _stream())
  else:
    _lib.call("bv_map_attn_fwd", _p(q), _p(kv), _p(o), _p(p), n, L, H, _stream())
  return o, p


def map_attn_bwd(q, kv, p, d_o, n, L, H):
  _chk(d_o, BF16, "map_attn.do")
  assert d_o.is_contiguous()
  Dh = _head_dim(q.numel(), n, 1, H, "map_attn_bwd")
  dq = torch.empty_like(q)
  dkv = torch.empty_like(kv)
  if Dh != 64 or L > 2048:
    _lib.call("bv_map_attn_bwd_dh", _p(q), _p(kv), _p(p), _p(d_o), _p(dq), _p(dkv), n, L, H, Dh, _stream())
  else:
    _lib.call("bv_map_attn_bwd", _p(q), _p(kv), _p(p), _p(d_o), _p(dq), _p(dkv), n, L, H, _stream())
  return dq, dkv


# ------------------------------------------------------------ data movers ---
def patchify(image, P):
  _chk(image, F32, "patchify.image")
  assert image.is_contiguous() and image.dim() == 4 and image.shape[3] == 3, image.shape
  n, Hi, Wi, _ = image.shape
  h, w = Hi // P, Wi // P
  K = P * P * 3
  Kp = (K + 7) // 8 * 8    # rows padded with zero columns when K is not a multiple of 8 (14 x 14 x 3 = 588 -> 592)
  out = torch.empty((n * h * w, Kp), device=image.device, dtype=BF16)
  _lib.call("bv_patchify_ld", _p(image), _p(out), n, Hi, Wi, P, Kp, _stream())
  return out, (h, w)


def embed_fwd(ids, table, pos, n, L):
  _chk(ids, torch.int32, "embed.ids"); _chk(table, F32, "embed.table"); _chk(pos, F32, "embed.pos")
  vocab, D = table.shape
  x = torch.empty((n * L, D), device=table.device, dtype=F32)
  _lib.call("bv_embed_fwd", _p(ids), _p(table), _p(pos), _p(x), n, L, D, vocab, _stream())
  return x


def embed_bwd(ids, dx, dtable):
  _chk(ids, torch.int32, "embed.ids"); _chk(dx, F32, "embed.dx"); _chk(dtable, F32, "embed.dtable")
  vocab, D = dtable.shape
  _lib.call("bv_embed_bwd", _p(ids), _p(dx), _p(dtable), dx.shape[0], D, vocab, _stream())


def colsum(x, out, rows=None, cols=None):
  """out[c] += sum_r x[r][c]."""
  ldx = _rowmajor2d(x, "colsum.x")
  _chk(out, F32, "colsum.out")
  rows = x.shape[0] if rows is None else rows
  cols = x.shape[1] if cols is None else cols
  _lib.call("bv_colsum", _p(x), int(x.dtype == F32), ldx, _p(out), rows, cols, _stream())


def batchsum(x, out, n, L, D):
  _chk(x, F32, "batchsum.x"); _chk(out, F32, "batchsum.out")
  _lib.call("bv_batchsum", _p(x), _p(out), n, L, D, _stream())


def cast_bf16(x, out=None):
  _chk(x, F32, "cast_bf16.x")
  assert x.is_contiguous()
  if out is None:
    out = torch.empty(x.shape, device=x.device, dtype=BF16)
  _lib.call("bv_cast_bf16", _p(x), _p(out), x.numel(), _stream())
  return out


def cast_f32(x, out=None):
  _chk(x, BF16, "cast_f32.x")
  assert x.is_contiguous()
  if out is None:
    out = torch.empty(x.shape, device=x.device, dtype=F32)
  _lib.call("bv_cast_f32", _p(x), _p(out), x.numel(), _stream())
  return out


def transpose_bf16(x, out=None):
  """out[c][r] = x[r][c] (bf16, 2-D, contiguous rows)."""
  _chk(x, BF16, "transpose.x")
  lds = _rowmajor2d(x, "transpose.x")
  rows, cols = x.shape
  if out is None:
    out = torch.empty((cols, rows), device=x.device, dtype=BF16)
  ldd = _rowmajor2d(out, "transpose.out")
  _lib.call("bv_transpose_bf16", _p(x), _p(out), rows, cols, lds, ldd, _stream())
  return out


TR_LEAF = np.dtype([("src", np.uint64), ("dst", np.uint64), ("lds", np.int64), ("ldd", np.int64), ("rows", np.int32),
                    ("cols", np.int32), ("tile0", np.int32), ("tiles_x", np.int32)])   # struct bv_tr_leaf (include/bvhip.h)


def transpose_table(pairs, device):
  """Device table for transpose_bf16_batched from [(src [rows, cols], dst [cols, >= rows])] (bf16, row-major).
  Returns (table tensor, number of entries, total 64 x 64 tiles); the table holds raw addresses: the caller
  keeps the tensors alive and rebuilds it when one of them is re-allocated."""
  tab = np.zeros(len(pairs), TR_LEAF)
  tiles = 0
  for i, (src, dst) in enumerate(pairs):
    _chk(src, BF16, "transpose.x")
    _chk(dst, BF16, "transpose.out")
    lds, ldd = _rowmajor2d(src, "transpose.x"), _rowmajor2d(dst, "transpose.out")
    rows, cols = src.shape
    if dst.shape[0] != cols or ldd < rows:
      raise ValueError(f"transpose_table: dst {tuple(dst.shape)} does not hold the transpose of {tuple(src.shape)}")
    tx = (cols + 63) // 64
    tab[i] = (src.data_ptr(), dst.data_ptr(), lds, ldd, rows, cols, tiles, tx)
    tiles += tx * ((rows + 63) // 64)
  return torch.from_numpy(tab.view(np.uint8).copy()).to(device), len(pairs), tiles


def transpose_bf16_batched(table, nleaves, tiles):
  _lib.call("bv_transpose_bf16_batched", _p(table), nleaves, tiles, _stream())


def concat_cls(cls, x, n, L, D):
  y = torch.empty((n * (L + 1), D), device=x.device, dtype=F32)
  _lib.call("bv_concat_cls", _p(cls), _p(x), _p(y), n, L, D, _stream())
  return y


def pool_gap_fwd(x, n, L, D, lens=None):
  """lens (int32 [n], optional): average over the first lens[b] tokens only (NaFlex)."""
  y = torch.empty((n, D), device=x.device, dtype=F32)
  if lens is not None:
    _lib.call("bv_pool_gap_masked_fwd", _p(x), _p(y), _p(lens), n, L, D, _stream())
  else:
    _lib.call("bv_pool_gap_fwd", _p(x), _p(y), n, L, D, _stream())
  return y


def pool_gap_bwd(dy, n, L, D, lens=None):
  dx = torch.empty((n * L, D), device=dy.device, dtype=F32)
  if lens is not None:
    _lib.call("bv_pool_gap_masked_bwd", _p(dy), _p(dx), _p(lens), n, L, D, _stream())
  else:
    _lib.call("bv_pool_gap_bwd", _p(dy), _p(dx), n, L, D, _stream())
  return dx


def pool_max_fwd(x, n, L, D, lens=None):
  """max over the L positions (text pool_type "max" / "gmp"); returns (pooled [n, D], argmax int32 [n, D]).
  lens (int32 [n], optional): over the first lens[b] positions only (NaFlex pool_type "max")."""
  _chk(x, F32, "pool_max.x")
  y = torch.empty((n, D), device=x.device, dtype=F32)
  arg = torch.empty((n, D), device=x.device, dtype=torch.int32)
  if lens is not None:
    _lib.call("bv_pool_max_masked_fwd", _p(x), _p(y), _p(arg), _p(lens), n, L, D, _stream())
  else:
    _lib.call("bv_pool_max_fwd", _p(x), _p(y), _p(arg), n, L, D, _stream())
  return y, arg


def pool_max_bwd(dy, arg, n, L, D):
  _chk(dy, F32, "pool_max.dy")
  dx = torch.empty((n * L, D), device=dy.device, dtype=F32)
  _lib.call("bv_pool_max_bwd", _p(dy), _p(arg), _p(dx), n, L, D, _stream())
  return dx


def naflex_posemb_weights(yabs, xabs, P):
  """[n*N, P*P] bf16 resize-and-gather weights of the NaFlex position embedding (bv_naflex_posemb_weights)."""
  _chk(yabs, torch.int32, "naflex.yabs"); _chk(xabs, torch.int32, "naflex.xabs")
  assert yabs.is_contiguous() and xabs.is_contiguous() and yabs.shape == xabs.shape
  n, N = yabs.shape
  W = torch.empty((n * N, P * P), device=yabs.device, dtype=BF16)
  _lib.call("bv_naflex_posemb_weights", _p(yabs), _p(xabs), _p(W), n, N, P, _stream())
  return W


def l2norm_fwd(z, eps=1e-8):
  _chk(z, F32, "l2norm.z")
  assert z.is_contiguous()
  zn = torch.empty_like(z)
  norm = torch.empty((z.shape[0],), device=z.device, dtype=F32)
  _lib.call("bv_l2norm_fwd", _p(z), _p(zn), _p(norm), z.shape[0], z.shape[1], float(eps), _stream())
  return zn, norm


def l2norm_bwd(z, norm, dzn, eps=1e-8):
  _chk(dzn, F32, "l2norm.dzn")
  assert dzn.is_contiguous()
  dz = torch.empty_like(z)
  _lib.call("bv_l2norm_bwd", _p(z), _p(norm), _p(dzn), _p(dz), z.shape[0], z.shape[1], float(eps), _stream())
  return dz


# -------------------------------------------------------------- loss / opt --
def siglip_loss_(raw, t_param, b_param, stats, row_offset, B_global):
  """In place: raw [n,B] (zimg.ztxt_all^T) -> G = dL/dS; stats (f64[3]) accumulated."""
  _chk(raw, F32, "siglip_loss.raw")
  assert raw.is_contiguous() and stats.dtype == torch.float64
  n, B = raw.shape
  _lib.call("bv_siglip_loss", _p(raw), _p(t_param), _p(b_param), _p(stats), n, B, row_offset,
            B_global, _stream())


LOGIT_STATS_NAMES = ("pos_min_logit", "pos_max_logit", "pos_avg_logit", "local_neg_min_logit",
                     "local_neg_max_logit", "local_neg_avg_logit", "neg_min_logit", "neg_max_logit",
                     "neg_avg_logit")


def logit_stats(raw, t_param, b_param, row_offset):
  """fp32[9] logit statistics of logits = exp(t') raw + b (order LOGIT_STATS_NAMES); raw [n, B] as
  handed to siglip_loss_ (call this BEFORE it: the loss kernel overwrites raw)."""
  _chk(raw, F32, "logit_stats.raw")
  assert raw.is_contiguous()
  n, B = raw.shape
  part = torch.empty((512 * 9,), device=raw.device, dtype=F32)
  out = torch.empty((9,), device=raw.device, dtype=F32)
  _lib.call("bv_logit_stats", _p(raw), _p(t_param), _p(b_param), _p(part), _p(out), n, B, row_offset, _stream())
  return out


def dot_(a, b, out):
  """out (f64[1]) += sum(a * b)."""
  _chk(a, F32, "dot.a"); _chk(b, F32, "dot.b")
  assert a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel() and out.dtype == torch.float64
  _lib.call("bv_dot_f32", _p(a), _p(b), a.numel(), _p(out), _stream())
  return out


def softmax_xent(logits, labels, loss_sum, want_grad=True, n_global=None, name="bv_softmax_xent"):
  """loss_sum (f64[1]) += mean-over-n_global softmax cross-entropy; returns dlogits (or None)."""
  _chk(logits, F32, "xent.logits"); _chk(labels, F32, "xent.labels")
  assert logits.is_contiguous() and labels.is_contiguous() and loss_sum.dtype == torch.float64
  n, C = logits.shape
  dl = torch.empty_like(logits) if want_grad else None
  _lib.call(name, _p(logits), _p(labels), _p(loss_sum), _p(dl), n, C, int(n_global or n), _stream())
  return dl


def sigmoid_xent(logits, labels, loss_sum, want_grad=True, n_global=None):
  return softmax_xent(logits, labels, loss_sum, want_grad, n_global, name="bv_sigmoid_xent")


def tanh_fwd(x):
  _chk(x, F32, "tanh.x")
  assert x.is_contiguous()
  y = torch.empty_like(x)
  _lib.call("bv_tanh_fwd", _p(x), _p(y), x.numel(), _stream())
  return y


def tanh_bwd(y, dy):
  _chk(y, F32, "tanh.y"); _chk(dy, F32, "tanh.dy")
  assert y.is_contiguous() and dy.is_contiguous()
  dx = torch.empty_like(y)
  _lib.call("bv_tanh_bwd", _p(y), _p(dy), _p(dx), y.numel(), _stream())
  return dx


def mixup(x, a):
  """a x + (1 - a) roll(x, 1, axis 0) (utils.py:1146-1154)."""
  _chk(x, F32, "mixup.x")
  assert x.is_contiguous()
  out = torch.empty_like(x)
  _lib.call("bv_mixup", _p(x), _p(out), float(a), x.shape[0], x.numel() // x.shape[0], _stream())
  return out


def dropout_f32(x, key, rate, addend=None, out=None, out_bf16=None):
  """addend (or 0) + keep x / (1 - rate) on fp32 `x` (models/vit.py:100,109,228); the keep bits are a function of
  (key, element index) only - see bv_dropout_f32.  Writes `out` (fp32; may be x or addend) and / or `out_bf16`;
  with neither given a new fp32 tensor is returned."""
  _chk(x, F32, "dropout.x")
  assert x.is_contiguous()
  if addend is not None:
    _chk(addend, F32, "dropout.addend")
    assert addend.is_contiguous() and addend.numel() == x.numel()
  if out is None and out_bf16 is None:
    out = torch.empty_like(x)
  for o, dt in ((out, F32), (out_bf16, BF16)):
    if o is not None:
      _chk(o, dt, "dropout.out")
      assert o.is_contiguous() and o.numel() == x.numel()
  _lib.call("bv_dropout_f32", _p(x), _p(addend), _p(out), _p(out_bf16), x.numel(), int(key) & (2 ** 64 - 1), float(rate), _stream())
  return out if out is not None else out_bf16


def dropout_bf16_(a, key, rate, b=None):
  """a (and b) *= keep / (1 - rate) in place, bf16, the same bits on both (models/vit.py:76: the dropout behind the
  GELU, applied to gelu(h) and to the stored gelu'(h) alike so that the backward's product carries the mask)."""
  _chk(a, BF16, "dropout.a")
  assert a.is_contiguous()
  if b is not None:
    _chk(b, BF16, "dropout.b")
    assert b.is_contiguous() and b.numel() == a.numel()
  _lib.call("bv_dropout_bf16", _p(a), _p(b), a.numel(), int(key) & (2 ** 64 - 1), float(rate), _stream())
  return a


def dropout_mask(shape, key, rate, device):
  """The keep bits (bool tensor of `shape`) the dropout kernels derive from `key` (tests / diagnostics)."""
  out = torch.empty(shape, device=device, dtype=torch.uint8)
  _lib.call("bv_dropout_mask", _p(out), out.numel(), int(key) & (2 ** 64 - 1), float(rate), _stream())
  return out.bool()


def sqnorm_(x, out):
  """out (f64[1]) += sum(x^2)."""
  _chk(x, F32, "sqnorm.x")
  assert x.is_contiguous() and out.dtype == torch.float64
  _lib.call("bv_sqnorm", _p(x), x.numel(), _p(out), _stream())


def adam_step_(params, grads, mu, nu, shadow, segs, chunk_seg, count, sched, gsq, clip_norm, b1, b2,
               eps, bc1, bc2, stats):
  """segs: device int32/float32 table of bv_adam_seg; sched: python list of floats (<= 8)."""
  import ctypes
  arr = (ctypes.c_float * len(sched))(*[float(v) for v in sched])
  _lib.call("bv_adam_step", _p(params), _p(grads), _p(mu), int(mu.dtype == BF16), _p(nu), _p(shadow),
            _p(segs), _p(chunk_seg), count, ctypes.cast(arr, ctypes.c_void_p), len(sched), _p(gsq),
            float(clip_norm or 0.0), float(b1), float(b2), float(eps), float(bc1), float(bc2),
            _p(stats), _stream())


def adafactor_leaf_(params, grads, momentum, shadow, view, state, factored, gsq, clip_norm, decay, eps, mom,
                    lr_eff, wd, sched, stats):
  """One leaf of the fused Adafactor step (bv_adafactor_leaf); view: ctypes array of 9 longs (host)."""
  import ctypes
  _lib.call("bv_adafactor_leaf", _p(params), _p(grads), _p(momentum),
            int(momentum is not None and momentum.dtype == BF16), _p(shadow),
            ctypes.cast(view, ctypes.c_void_p), _p(state), int(factored), _p(gsq), float(clip_norm or 0.0),
            float(decay), float(eps), float(mom), float(lr_eff), float(wd), float(sched), _p(stats), _stream())


def adafactor_step_(params, grads, momentum, shadow, leaves, nleaves, max_rows, max_cols, max_b, max_total, state, gsq,
                    clip_norm, decay, eps, mom, sched, stats, block_rms_clip=0.0, block_usq=None):
  """The whole Adafactor step in four launches (bv_adafactor_step); leaves: device uint8 tensor holding the
  bv_af_leaf table; sched: python list of this step's schedule values (<= 8).  block_rms_clip > 0 (with block_usq, a
  float64 scratch of nleaves elements): optax.clip_by_block_rms per leaf (one more launch)."""
  import ctypes
  arr = (ctypes.c_float * len(sched))(*[float(v) for v in sched])
  _lib.call("bv_adafactor_step", _p(params), _p(grads), _p(momentum),
            int(momentum is not None and momentum.dtype == BF16), _p(shadow), _p(leaves), int(nleaves), int(max_rows),
            int(max_cols), int(max_b), int(max_total), _p(state), _p(gsq), float(clip_norm or 0.0), float(decay),
            float(eps), float(mom), ctypes.cast(arr, ctypes.c_void_p), len(sched), _p(stats),
            float(block_rms_clip or 0.0), _p(block_usq), _stream())
