"""Two-tower image/text model behind `models.proj.image_text.two_towers`.

Mirrors big_vision/models/proj/image_text/two_towers.py: `Model(**config.model)`
with fields image/text/image_model/text_model/out_dim/temperature_init/bias_init
(:28-37); towers are resolved by module path below `big_vision_amd.models`
(:51-53, :64-66); embeddings are L2-normalised with eps 1e-8 (:60-61, :73-74);
`t` (log-temperature) and `b` are (1,)-shaped params (:76-85).  `apply` returns
`(zimg, ztxt, out)` and accepts `image=None` or `text=None` (:43).
"""
from __future__ import annotations

import importlib
import math
import os

import torch

from big_vision_amd import engine as E
from big_vision_amd import ops
from big_vision_amd import utils
from big_vision_amd.params import Entry, ParamStore, ParamTree, adhoc_store

F32 = torch.float32
MODELS_PKG = "big_vision_amd.models"


_SIDE_STREAMS = {}


class TwoTowersExec:
  def __init__(self, m: "Model", store: ParamStore, prefix: str, hw, seq_len):
    self.m, self.store = m, store
    self.img = m.image_tower.executor(store, f"{prefix}img/", hw) if hw is not None else None
    self.txt = m.text_tower.executor(store, f"{prefix}txt/", seq_len) if seq_len is not None else None
    self.prefix, self._ranges = prefix, {}
    self.t = E._W(store, f"{prefix}t")
    self.b = E._W(store, f"{prefix}b") if m.bias_init is not None else None
    # Trainer option config.tower_streams (default 2 since round 6; executors built outside a trainer start at 1): the text tower on a side stream beside the image
    # tower.  The towers share nothing until the loss (two_towers.py:56-75); their persistent GEMMs each fill the
    # grid, so what the second stream buys is the other tower's workgroups in the ragged last round of a launch
    # (text N = 768 GEMMs at 512 pairs: 1.5 rounds).  Same kernels on the same inputs: identical results.
    self.streams = 1
    self._side = None
    self._will_fork = None

  def _fork(self):
    """(main, side) with every transposed weight image current and the side stream behind everything enqueued so far.

    The side stream's `bv_ctx` is brought to the main context's state first - EVERY option (kernel variants, the CU
    reservation of an overlapped gradient sync, attention configuration, A/B switches set through `ops.option` /
    `ops.ctx_set`) and `use_workspace` - so that both towers always run one configuration.

    Stream-ownership invariant the allocator bookkeeping below relies on: everything the side stream ever does
    sits between a `side.wait_stream(main)` here and a `main.wait_stream(side)` / `current.wait_stream(side)` join
    in `fwd` / `bwd`.  Tensors produced on the side stream and consumed on the main one (ztxt, its norm, the text
    contexts when the backward runs on one stream) are `record_stream(main)`-ed where they cross."""
    main = torch.cuda.current_stream()
    if self._side is None:
      # BV_SIDE_STREAM_PRIORITY (A/B knob, tools/stream_priority_ab.py): HIP priority of the text tower's stream
      # (-1 = high, 0 = the default); within the noise: 512 pairs +-0.2 %, headline -0.4 +- 0.4 % (profiles/r06_stream_priority_ab.txt)
      # (one side stream per device and process, shared by every executor: see dp._collective_stream for why streams are not
      #  drawn from torch's pool again and again)
      prio = int(os.environ.get("BV_SIDE_STREAM_PRIORITY", "0"))
      key = (torch.device(self.store.device).index, prio)
      if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=self.store.device, priority=prio)
      self._side = _SIDE_STREAMS[key]
    E.refresh_twins(self.store)
    mc = ops.ctx()
    with torch.cuda.stream(self._side):
      ops.ctx().copy_options_from(mc)
    self._side.wait_stream(main)
    return main, self._side

  def _two_streams(self, image, text, collect=False):
    # (collect=True hands nested dicts of intermediate activations to the caller: they stay on one stream)
    return self.streams == 2 and not collect and image is not None and text is not None and text.is_cuda

  def _tower_trainable(self, tower):
    """False when every parameter of `tower` ("img/" | "txt/") is frozen (config.schedule None: LiT's image tower).  Such a
    tower's forward keeps NO context whatever `save` asks for - nothing is ever differentiated through it (the trainers
    hand `bwd` None for its cotangent, inputs take no gradient) - so its MLP writes gelu(h) only (BV_EPI_GELU_G) and its
    activations are freed as the forward proceeds."""
    if not hasattr(self, "_trainable"):
      self._trainable = {}
    if tower not in self._trainable:     # (the frozen set of a store is fixed at construction)
      frozen = self.store.frozen
      self._trainable[tower] = any(n.startswith(f"{self.prefix}{tower}") and n not in frozen for n in self.store.entries)
    return self._trainable[tower]

  def _backward_will_fork(self):
    """False when one of the towers takes no gradient (LiT / a frozen tower: `bwd` then runs the other tower alone on
    the main stream)."""
    if self._will_fork is None:     # (the frozen set of a store is fixed at construction)
      frozen = self.store.frozen
      self._will_fork = all(any(n.startswith(f"{self.prefix}{tower}") and n not in frozen for n in self.store.entries)
                            for tower in ("img/", "txt/"))
    return self._will_fork

  def _drop_kw(self, tower, name, drop_key):
    """{"drop": engine.Dropout} for a tower whose config asks for dropout > 0 when the pass is a training one
    (`drop_key`: the pass's 64-bit dropout key, None = deterministic; two_towers.py:56,69 hand **kw - train - to both
    towers, flax gives each module path its own stream of the "dropout" rng)."""
    rate = float(getattr(tower, "dropout", 0.0) or 0.0)
    if drop_key is None or rate <= 0.0:
      return {}
    return {"drop": E.Dropout(rate, drop_key).fold(name)}

  def fwd(self, image, text, save=False, collect=False, drop_key=None):
    out, ctx = {}, {}
    zimg = ztxt = None
    save_img = save if self._tower_trainable("img/") else False
    save_txt = save if self._tower_trainable("txt/") else False
    tkw = self._drop_kw(self.m.text_tower, "txt", drop_key)
    ikw = self._drop_kw(self.m.image_tower, "img", drop_key)
    # (a saving forward forks even when its backward will not - LiT / a frozen tower: the FORWARD overlap alone is worth
    #  1.1-1.9 % of the LiT steps, profiles/r06_lit_forward_fork_ab.txt; the one-stream backward then record_stream()s the
    #  text contexts it consumes, see bwd.  BV_FORK_FWD_TRAINABLE_ONLY=1 restores the former rule for an A/B.)
    if self._two_streams(image, text, collect) and (not save or self._backward_will_fork()
                                                    or os.environ.get("BV_FORK_FWD_TRAINABLE_ONLY") != "1"):
      main, side = self._fork()
      with torch.cuda.stream(side):
        z, o, c = self.txt.fwd(text, save_txt, collect, **tkw)
        ztxt, norm = ops.l2norm_fwd(z)
      out.update({f"txt/{k}": v for k, v in o.items()})
      out["txt/norm"] = norm.view(-1, 1)
      out["txt/normalized"] = ztxt
      ctx["txt"] = (c, z, norm)
      for t_ in [ztxt, norm, z] + [v for v in o.values() if torch.is_tensor(v)]:
        t_.record_stream(main)          # produced on the side stream, read by the loss / the caller on the main one
      ctx["txt_on_side"] = True
      text = None                        # (done; the image tower below runs on the main stream meanwhile)
      join = side
    else:
      join = None
    if text is not None:
      z, o, c = self.txt.fwd(text, save_txt, collect, **tkw)
      out.update({f"txt/{k}": v for k, v in o.items()})
      ztxt, norm = ops.l2norm_fwd(z)
      out["txt/norm"] = norm.view(-1, 1)
      out["txt/normalized"] = ztxt
      ctx["txt"] = (c, z, norm)
    if image is not None:
      z, o, c = self.img.fwd(image, save_img, collect, **ikw)
      out.update({f"img/{k}": v for k, v in o.items()})
      zimg, norm = ops.l2norm_fwd(z)
      out["img/norm"] = norm.view(-1, 1)
      out["img/normalized"] = zimg
      ctx["img"] = (c, z, norm)
    if join is not None:
      torch.cuda.current_stream().wait_stream(join)
    out["t"] = torch.exp(self.t.f32)
    out["t/parameter"] = self.t.f32
    if self.b is not None:
      out["b"] = self.b.f32
    return zimg, ztxt, out, (ctx if save else None)

  def check_inputs(self):
    """Raises what the towers' deferred input validation found for the batch(es) handed to fwd() since the last call
    (the BERT tower validates input_mask on the device and reads the verdict late, bert.BertExec._lengths).  The
    trainers call it before `opt.step()`, `Model.apply` before it returns: a refused batch is never committed."""
    for tw in (self.txt, self.img):
      chk = getattr(tw, "check_pending", None)
      if chk is not None:
        chk()

  def _block_ranges(self, tower, enc):
    """{block index: [lo, hi) of its gradients in the flat buffer} for `tower` ('img/' | 'txt/')."""
    key = (tower, enc)
    if key not in self._ranges:
      tw = self.m.image_tower if tower.endswith("img/") else self.m.text_tower
      out = {}
      for i in range(getattr(tw, "depth", 0)):
        r = self.store.grad_range(lambda n, p=f"{tower}{enc}/encoderblock_{i}/": n.startswith(p))
        if r is not None:
          out[i] = r
      self._ranges[key] = out
    return self._ranges[key]

  def bwd(self, ctx, dzimg, dztxt, sync=None):
    """dzimg / dztxt: gradients w.r.t. the NORMALISED embeddings (None = tower skipped).
    sync (dp.GradSync, last backward of a step on N > 1 ranks): gradient ranges are handed to
    the all-reduce as soon as they are final - every encoder block right after its backward is
    enqueued (28 MB per B/16 block: 24 + 24 messages that RCCL runs next to the remaining
    GEMMs), the rest of the text tower (embedding table, final norm, head) when the text backward
    is done, the image tower's final norm / MAP head together with its last block; what is left
    (stem, position embedding, t, b) is reduced by sync.finish()."""
    if dzimg is not None and "img" in ctx and ctx["img"][0] is None:   # a frozen tower kept no context: nothing to differentiate
      dzimg = None
    if dztxt is not None and "txt" in ctx and ctx["txt"][0] is None:
      dztxt = None
    two = self.streams == 2 and dztxt is not None and dzimg is not None and "txt" in ctx and "img" in ctx and dztxt.is_cuda
    if two:
      main, side = self._fork()
      dztxt.record_stream(side)
      with torch.cuda.stream(side):
        self._bwd_txt(ctx, dztxt, sync)
      self._bwd_img(ctx, dzimg, sync)
      main.wait_stream(side)
      return
    if dztxt is not None and "txt" in ctx:
      if ctx.get("txt_on_side"):   # saved by a forked forward, consumed (and freed) here on the main stream
        E.record_stream_tree(ctx["txt"], torch.cuda.current_stream())
      self._bwd_txt(ctx, dztxt, sync)
    if dzimg is not None and "img" in ctx:
      self._bwd_img(ctx, dzimg, sync)

  def _bwd_txt(self, ctx, dztxt, sync):
    store = self.store
    c, z, norm = ctx["txt"]
    on_block = None
    if sync is not None:
      rt = self._block_ranges(f"{self.prefix}txt/", "Encoder_0")
      on_block = lambda i, rt=rt: sync.launch(*rt[i]) if i in rt else None
    self.txt.bwd(c, ops.l2norm_bwd(z, norm, dztxt), on_block=on_block)
    if sync is not None:
      r = store.grad_range(lambda n: n.startswith(f"{self.prefix}txt/"))
      if r is not None:
        sync.launch_gaps(*r)

  def _bwd_img(self, ctx, dzimg, sync):
    store = self.store
    c, z, norm = ctx["img"]
    on_block = None
    if sync is not None:
      ri = self._block_ranges(f"{self.prefix}img/", "Transformer")
      stem = tuple(f"{self.prefix}img/{k}" for k in ("embedding", "pos_embedding", "cls", "patchln_pre", "patchln_post"))
      tail = store.grad_range(lambda n: n.startswith(f"{self.prefix}img/") and "/encoderblock_" not in n
                              and not n.startswith(stem))
      last = max(ri) if ri else None

      def on_block(i, ri=ri, tail=tail, last=last):
        if i == last and tail is not None:
          sync.launch(*tail)          # final norm + pooling head: final before the first block's backward
        if i in ri:
          sync.launch(*ri[i])
    self.img.bwd(c, ops.l2norm_bwd(z, norm, dzimg), on_block=on_block)


class Model:
  """Two towers transformer."""

  def __init__(self, image=None, text=None, text_model="proj.image_text.text_transformer",
               image_model="vit", out_dim=128, temperature_init=1.0, bias_init=None, name=None):
    self.image, self.text = dict(image or {}), dict(text or {})
    self.text_model, self.image_model = text_model, image_model
    self.out_dim, self.temperature_init, self.bias_init = out_dim, temperature_init, bias_init
    out_dims = (out_dim, out_dim) if isinstance(out_dim, int) else tuple(out_dim)
    self.text_tower = importlib.import_module(f"{MODELS_PKG}.{text_model}").Model(
        **{"num_classes": out_dims[1], **self.text}, name="txt")
    self.image_tower = importlib.import_module(f"{MODELS_PKG}.{image_model}").Model(
        **{"num_classes": out_dims[0], **self.image}, name="img")
    self._execs = {}

  def entries(self, prefix, hw, seq_len):
    ents = self.image_tower.entries(f"{prefix}img/", hw) + self.text_tower.entries(f"{prefix}txt/", seq_len)
    ents.append(Entry(f"{prefix}t", (1,), E.init_const(math.log(self.temperature_init))))
    if self.bias_init is not None:
      ents.append(Entry(f"{prefix}b", (1,), E.init_const(self.bias_init)))
    return ents

  def scan_prefixes(self, prefix=""):
    sp = getattr(self.image_tower, "scan_prefixes", lambda p: ())(f"{prefix}img/")
    return tuple(sp) + tuple(getattr(self.text_tower, "scan_prefixes", lambda p: ())(f"{prefix}txt/"))

  def leaf_names(self, image_shape, text_shape):
    """Leaf names as presented (stacked `encoderblock` for towers built with scan=True)."""
    from big_vision_amd.params import external_leaf_names
    hw = self.image_tower.grid(tuple(image_shape))
    return external_leaf_names([l for e in self.entries("", hw, text_shape[1]) for l, _ in e.flax_leaves()],
                               self.scan_prefixes())

  def make_store(self, image_shape, text_shape, device=None, frozen_leaves=()):
    """Allocates the flat parameter store for both towers (+ t, b).

    `frozen_leaves`: exact Flax leaf names that get no gradient / optimizer
    state (config.schedule None entries); they are laid out last."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    hw = self.image_tower.grid(tuple(image_shape))
    ents = self.entries("", hw, text_shape[1])
    from big_vision_amd.params import scan_name
    frozen_leaves = set(frozen_leaves)
    sp = self.scan_prefixes()
    frozen = set()
    for e in ents:
      hits = [scan_name(leaf, sp)[0] in frozen_leaves for leaf, _ in e.flax_leaves()]
      if any(hits) and not all(hits):
        raise NotImplementedError(f"fused tensor {e.name} is only partially frozen")
      if all(hits):
        frozen.add(e.name)
    return ParamStore(ents, device, frozen=frozen, scan_prefixes=sp)

  def init(self, rng, image, text=None, **kw):
    del kw
    from big_vision_amd.models.vit import _seed_of
    ishape = tuple(image[0].shape) if isinstance(image, (tuple, list)) else tuple(image.shape)
    store = self.make_store(ishape, tuple(text.shape))
    store.init_random(_seed_of(rng))
    store.refresh_shadow()
    return {"params": store.tree()}

  def executor(self, store, prefix, image_shape, text_shape):
    hw = self.image_tower.grid(tuple(image_shape)) if image_shape is not None else None
    sl = text_shape[1] if text_shape is not None else None
    key = (id(store), prefix, hw, sl, getattr(store, "want_grads", False))
    if key not in self._execs:
      self._execs[key] = TwoTowersExec(self, store, prefix, hw, sl)
    return self._execs[key]

  def _image_shape_of(self, params):
    """(1, H, W, 3) of the images an ad-hoc parameter tree was built for, from its learned position embedding (a
    square grid of patches; sincos2d towers carry no such leaf: pass an image, or a tree bound to a store)."""
    pe = params.get("img", {}).get("pos_embedding")
    if pe is None:
      raise ValueError("cannot size the image tower from this parameter tree (no img/pos_embedding): pass an image")
    tokens = int(pe.shape[1])
    side = int(round(math.sqrt(tokens)))
    if side * side != tokens:
      raise ValueError(f"img/pos_embedding has {tokens} positions, not a square grid: pass an image")
    ph, pw = self.image_tower.patch_size
    return (1, side * ph, side * pw, 3)

  def apply(self, variables, image, text=None, *, train=False, rngs=None, collect=True, **kw):
    """train=True: the towers' dropout (if their configs ask for any) is drawn from rngs["dropout"] (siglip.py:288-290)."""
    del kw
    drop_key = None
    if train and any(float(getattr(t, "dropout", 0.0) or 0.0) > 0.0 for t in (self.image_tower, self.text_tower)):
      if not rngs or "dropout" not in rngs:
        raise ValueError("train=True with dropout > 0 needs rngs={'dropout': key}")
      from big_vision_amd.models import vit as _vit
      drop_key = _vit._seed_of(rngs["dropout"])
    params = variables["params"]
    if isinstance(params, ParamTree) and params.store is not None:
      store, prefix = params.store, params.prefix
    else:
      # two_towers.py:43: either input may be None.  The store holds BOTH towers' parameters, so the geometry of the
      # absent input is read off the tree itself (position-embedding lengths)
      ishape = (tuple(image[0].shape) if isinstance(image, (tuple, list)) else tuple(image.shape)) if image is not None \
          else self._image_shape_of(params)
      tshape = tuple(text.shape) if text is not None else (1, int(params["txt"]["pos_embedding"].shape[1]))
      store = adhoc_store(self._execs, ("two_towers", ishape[1:], tshape[1:], torch.cuda.current_device()),
                          params, lambda: self.make_store(ishape, tshape))
      prefix = ""
    store.refresh_shadow()
    ishape = None if image is None else (tuple(image[0].shape) if isinstance(image, (tuple, list)) else tuple(image.shape))
    ex = self.executor(store, prefix, ishape, None if text is None else tuple(text.shape))
    zimg, ztxt, out, _ = ex.fwd(image, text, save=False, collect=collect, drop_key=drop_key)
    ex.check_inputs()
    return zimg, ztxt, out


# init-file keys of `config.model_init` -> (leaf of the param tree, default tower module or None for a plain array)
_PARTS = (
    (("image", "img"), "img", "image_model", "vit"),
    (("text", "txt"), "txt", "text_model", "proj.image_text.text_transformer"),
    (("temperature", "t"), "t", None, None),
    (("bias", "b"), "b", None, None),
)


def load(init_params, init_files, model_cfg, img_load_kw={}, txt_load_kw={}):  # pylint: disable=dangerous-default-value
  """Restores the two towers and the loss scalars (contract of two_towers.py:93-137).

  `init_files`: a vanity name, ONE checkpoint path (its `img` / `txt` / `t` (/ `b` when the model has a bias)
  sub-trees are taken), or a dict {"img"|"image", "txt"|"text", "t"|"temperature", "b"|"bias"} -> path.  Towers are
  restored by their own module's `load`; parts without an init file keep `init_params`; a key nobody consumed is
  refused with the reference's message (a typo in a config must not pass silently)."""
  if isinstance(init_files, str):
    init_files = VANITY_NAMES.get(init_files, init_files)
  if isinstance(init_files, str):
    parts = ["img", "txt", "t"] + (["b"] if "bias_init" in model_cfg.keys() else [])
    todo = {k: f"{init_files}:{k}" for k in parts}
  else:
    todo = dict(init_files)
  restored_params = dict(init_params) if init_params else {"img": None, "txt": None}
  tower_kw = {"img": img_load_kw, "txt": txt_load_kw}
  for aliases, leaf, module_key, default_module in _PARTS:
    found = [todo.pop(a) for a in aliases if a in todo]
    source = found[0] if found else None
    if not source:
      continue
    if module_key is None:
      restored_params[leaf] = utils.load_params(source)
    else:
      tower = importlib.import_module(f"{MODELS_PKG}.{model_cfg.get(module_key, default_module)}")
      restored_params[leaf] = tower.load(restored_params.get(leaf), source, model_cfg.get(aliases[0]), **tower_kw[leaf])
  init_files = todo
  assert not init_files, (
      f"There's something unused left in `config.model_init`. You probably got "
      f"a typo. Here it is: {init_files}")
  return restored_params


# Shortcut names for some canonical paper checkpoints (gs:// paths; kept for
# config compatibility — unreachable offline).
VANITY_NAMES = {
    "SigLIP B/16 224": "gs://big_vision/siglip/webli_en_b16_224_63724782.npz",
    "SigLIP B/16 256": "gs://big_vision/siglip/webli_en_b16_256_60500360.npz",
    "SigLIP L/16 256": "gs://big_vision/siglip/webli_en_l16_256_60552751.npz",
}
