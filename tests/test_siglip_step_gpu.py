"""End-to-end parity: the HIP SigLIP training step vs the CPU oracle.

Same random-init weights and synthetic batch go through (a) the libbvhip path
(`big_vision_amd.trainers.proj.image_text.siglip.update_fn`) and (b) the fp64
oracle restatement of the reference step (oracle/bv_oracle.py).  Tolerances
(bf16 MFMA operands / fp32 accumulate vs fp64):
  embeddings zimg/ztxt (unit norm)   max-abs  <= 2e-2
  logits  S = t z.z + b  (t = 10)    max-abs  <= 0.25
  loss                               rel      <= 1e-2
  parameter gradients                per tensor: cosine >= 0.999 and rel-L2 <= 3e-2; the
                                     error the prescribed bf16-operand arithmetic itself shows on
                                     each tensor (oracle with bf16-rounded contraction operands vs
                                     fp64: _parity.bf16_floor) is REPORTED next to it, it is not a
                                     bound; a tensor outside the bounds is a per-tensor exception
                                     named in the test body with its own bound;
                                     tensors below 1e-3 of the global grad norm: abs err <= 2e-3 of it
  params after 1 Adam step           compared to the oracle chain fed OUR grads
                                     (isolates the optimizer): rtol 1e-5
"""
import math
import os

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu


LIT_SCHEDULE = [("img/.*", None), (".*", dict(decay_type="cosine", warmup_steps=2))]
_ORACLE_CACHE = {}   # see _run_case


def _cfg(total_steps=10, **kw):
  from big_vision_amd.compat.ml_collections import ConfigDict
  c = ConfigDict()
  c.lr = 1e-3
  c.wd = 1e-2
  c.schedule = dict(decay_type="cosine", warmup_steps=2)
  c.optax_name = "scale_by_adam"
  c.grad_clip_norm = 1.0
  c.total_steps = total_steps
  for k, v in kw.items():
    c[k] = v
  return c


def _run_case(dev, image_cfg, text_cfg, E, n, res, seq, vocab, bias_init=-10.0, config=None,
              tol_z=2e-2, tol_logit=0.25, frozen=(), floor=False, dirty_step=False, case=None, rel_max=None,
              exceptions=None, text_model=None, pad_id=None, cos_min=None, optimizer_check=True):
  """frozen: leaf-name prefixes config.schedule freezes (LiT).  floor: also measure the bf16-operand
  noise floor of the oracle for this case (tests/_parity.py; reported, not a bound).  exceptions:
  {leaf name: (rel_max, cos_min)} for tensors held to their own stated bound.
  dirty_step: the weights are edited in place (as store.load_tree does) and update_fn runs FIRST,
  with the bf16 shadow still dirty - the step itself has to refresh it; the forward-only parity
  then runs on the restored pre-step weights."""
  import bv_oracle as O
  import _parity
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  from big_vision_amd import utils as u

  case = case or f"siglip {image_cfg.get('variant', image_cfg.get('width'))} n={n} res={res} seq={seq}"
  # text_model (two_towers.py:51-53): another text tower by module path, e.g. "proj.flaxformer.bert" whose
  # config carries its own vocabulary; pad_id: the sticky-EOS padding of the synthetic batch (id 1) is
  # re-labelled (BERT's input_mask is text != 0)
  tcfg = dict(text_cfg) if text_model else {**text_cfg, "vocab_size": vocab}
  mkw = dict(text_model=text_model) if text_model else {}
  model = two_towers.Model(image=image_cfg, text=tcfg, out_dim=(None, E), temperature_init=10.0, bias_init=bias_init,
                           **mkw)
  config = config or _cfg()
  image, text = O.synthetic_batch(1, n, res, seq, vocab)
  if pad_id is not None:
    text = torch.where(text == 1, torch.full_like(text, pad_id), text)
  image_d, text_d = image.to(dev), text.to(dev)
  train_state, sched_fns = siglip.make_train_state(model, config, tuple(image.shape), tuple(text.shape),
                                                   rng=0, total_steps=config.total_steps)
  store = train_state["params"].store
  is_frozen = lambda k: any(k.startswith(p) for p in frozen)
  assert {e for e in store.frozen} == {e for e in store.entries if is_frozen(e)}
  # break the symmetric inits (zero biases, unit scales) so every gradient path is exercised
  g = torch.Generator().manual_seed(7)
  for name in store.leaf_names():
    if name.endswith(("bias", "scale", "cls")):
      leaf = store.leaf(name)
      leaf.add_((0.05 * torch.randn(leaf.shape, generator=g)).to(dev))
  store.mark_dirty()
  if not dirty_step:
    store.refresh_shadow()

  params64 = O.recover_tree([(k, v.detach().cpu().double().clone().requires_grad_(not is_frozen(k)))
                             for k, v in u.tree_flatten_with_names(train_state["params"])[0]])
  okw = dict(image_cfg=image_cfg, text_cfg=tcfg, out_dim=(None, E), **mkw)
  p_before = {k: v.detach().cpu().double().clone() for k, v in u.tree_flatten_with_names(train_state["params"])[0]}
  # The fp64 oracle pass (forward + backward, and the bf16-operand floor) depends on the model, the weights and
  # the batch only - not on the trainer options a case varies (residual stream, micro-batching, context kind).
  # Cases that share all three reuse it (the B/16 n = 32 oracle alone is ~45 s of host time); the key holds a
  # fingerprint of the weights, so a case with other weights can never pick up a stale reference.
  fp = float(sum(v.sum().item() * (i + 1) for i, v in enumerate(p_before.values())))
  ckey = (repr(sorted(image_cfg.items(), key=str)), repr(sorted(tcfg.items(), key=str)), E, n, res, seq, vocab, bias_init,
          tuple(frozen), text_model, pad_id, fp)
  ref = _ORACLE_CACHE.get(ckey)
  if ref is None:
    loss_ref, (zi_ref, zt_ref, logits_ref, _) = O.siglip_step_loss(params64, image.double(), text, **okw)
    loss_ref.backward()
    ref = dict(loss=loss_ref.detach(), zi=zi_ref.detach(), zt=zt_ref.detach(), logits=logits_ref.detach(),
               gref={k: v.grad for k, v in u.tree_flatten_with_names(params64)[0] if v.grad is not None}, floor=None)
    _ORACLE_CACHE[ckey] = ref
  loss_ref, zi_ref, zt_ref, logits_ref = ref["loss"], ref["zi"], ref["zt"], ref["logits"]

  def forward_parity():
    from big_vision_amd import engine as E
    old_stream = E.set_residual_stream(config.get("residual_stream", "float32"))   # the stream the step will use
    try:
      zimg, ztxt, out = model.apply({"params": train_state["params"]}, image_d, text_d, collect=False)
      assert (zimg.cpu().double() - zi_ref).abs().max() <= tol_z
      assert (ztxt.cpu().double() - zt_ref).abs().max() <= tol_z
      t = math.exp(store.leaf("t").item()); b = store.leaf("b").item()
      logits = (zimg.double() @ ztxt.double().T * t + b).cpu()
      assert (logits - logits_ref).abs().max() <= tol_logit
      loss_fwd = siglip.loss_fn(model, train_state["params"], image_d, text_d)
      assert abs(loss_fwd.item() - loss_ref.item()) <= 1e-2 * abs(loss_ref.item())
    finally:
      E.set_residual_stream(old_stream)

  if not dirty_step:
    forward_parity()

  # ---- one training step --------------------------------------------------------
  update_fn = siglip.make_update_fn(model, config)
  train_state, meas = update_fn(train_state, None, {"image": image_d, "labels": text_d})
  assert abs(meas["training_loss"].item() - loss_ref.item()) <= 1e-2 * abs(loss_ref.item())
  gref = ref["gref"]
  gours = {k: v.detach().cpu().double() for k, v in u.tree_flatten_with_names(store.tree("grad"))[0]}
  fl = None
  if floor:
    if ref["floor"] is None:
      for k, v in u.tree_flatten_with_names(params64)[0]:   # bf16_floor compares against the fp64 grads in .grad
        v.grad = gref.get(k)
      ref["floor"] = _parity.bf16_floor(lambda p: O.siglip_step_loss(p, image.double(), text, **okw)[0], params64)
    fl = ref["floor"]
  kw_tol = {} if rel_max is None else {"rel_max": rel_max}
  if cos_min is not None:
    kw_tol["cos_min"] = cos_min
  gnorm, rows = _parity.compare_grads(case, gref, gours, frozen=frozen, floor=fl, exceptions=exceptions, **kw_tol)
  # l2_grads / clip norm cover the trainable leaves only (optax.py:105, siglip.py:316)
  assert abs(meas["l2_grads"].item() - gnorm) <= 2e-2 * gnorm, (meas["l2_grads"].item(), gnorm)
  if not optimizer_check:
    # (the fp64 optimizer oracle walks every parameter in Python: ~60 s of host time at 652 M parameters; the cases
    # that skip it have a sibling of the same widths that runs it)
    return rows
  # ---- optimizer: oracle chain on OUR grads must reproduce OUR new params -------
  opt = train_state["opt"]
  assert opt.mu.numel() == store.trainable_count == opt.nu.numel() == store.grad.numel()
  orc = O.OptaxOracle(config.to_dict(), O.recover_tree(list(p_before.items())),
                      sched_kw=dict(total_steps=config.total_steps, batch_size=n))
  g_all = {k: gours.get(k, torch.zeros_like(v)) for k, v in p_before.items()}
  upd = orc.update(O.recover_tree(list(g_all.items())), O.recover_tree(list(p_before.items())))
  upd = dict(O.tree_flatten_with_names(upd))
  p_after = {k: v.detach().cpu().double() for k, v in u.tree_flatten_with_names(train_state["params"])[0]}
  for k, v in p_after.items():
    ref = p_before[k] + upd[k]
    err = (v - ref).abs().max().item()
    assert err <= 1e-5 * max(1.0, ref.abs().max().item()), f"{k}: param err {err:.3e}"
    if is_frozen(k):
      assert torch.equal(v, p_before[k]), f"frozen leaf {k} changed"
  l2p = math.sqrt(sum((v ** 2).sum().item() for v in p_after.values()))          # incl. frozen (siglip.py:318)
  l2u = math.sqrt(sum((upd[k] ** 2).sum().item() for k in p_after))
  assert abs(meas["l2_params"].item() - l2p) <= 1e-4 * l2p
  assert abs(meas["l2_updates"].item() - l2u) <= 1e-3 * l2u
  if dirty_step:   # forward-only parity on the restored pre-step weights (load_tree -> dirty -> apply refreshes)
    store.load_tree(O.recover_tree(list(p_before.items())))
    forward_parity()
  return rows


def test_tiny_two_towers_step(dev):
  image_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="map")
  text_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2)
  _run_case(dev, image_cfg, text_cfg, E=128, n=8, res=64, seq=16, vocab=100, floor=True)


def test_tiny_tok_pooling(dev):
  """pool_type='tok' image tower (cls token) with LiT's bias_init; nothing frozen here - the
  frozen-tower step is test_lit_frozen_image_tower_step below.  rel_max 4e-2: the only tensor above
  the 3e-2 default in ANY case is the scalar temperature gradient of this 6-pair batch (measured
  0.0331 twice, bf16-operand floor 0.0156: a sum of 36 signed terms; 0.005-0.009 at n >= 8)."""
  image_cfg = dict(width=128, depth=1, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="tok")
  text_cfg = dict(width=128, depth=1, mlp_dim=256, num_heads=2)
  _run_case(dev, image_cfg, text_cfg, E=128, n=6, res=48, seq=8, vocab=64, bias_init=-2.71, floor=True, rel_max=4e-2)


@pytest.mark.parametrize("pool", ["max", "gmp", "mean", "first"])
def test_tiny_text_pooling_variants(dev, pool):
  """The text tower's other pool_type values (text_transformer.py:83-90; the BASELINE configs use "last" and "map"):
  "max" / "gmp" = max over the sequence (bv_pool_max_fwd / _bwd: the cotangent goes to the arg-max position), "mean",
  "first" - against the fp64 oracle, the bounds of every other case."""
  image_cfg = dict(width=128, depth=1, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="gap")
  text_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, pool_type=pool)
  # max pooling is not continuous: where two positions of a feature are closer than the bf16 rounding of the GEMM
  # operands, the arg-max - and with it the whole cotangent of that feature - moves to another token.  The oracle
  # ITSELF shows it: its gradients with bf16-rounded operands differ from its fp64 gradients by rel-L2 0.06-0.13
  # (cosine 0.992-0.997) on this toy text tower (the "bf16 floor" printed next to each tensor; measured 0.124 / ours
  # 0.107 on the worst tensor).  Forward, loss and the optimizer keep the default bounds; the gradient bound of the two
  # max cases is the floor's order.  The kernels themselves are exact (test_kernels_gpu.py::test_pool_max_*).
  kw = dict(rel_max=0.2, cos_min=0.98) if pool in ("max", "gmp") else {}
  _run_case(dev, image_cfg, text_cfg, E=128, n=8, res=48, seq=16, vocab=100, floor=True, **kw)


@pytest.mark.parametrize("which", ["tiny", "tiny_tok_lit", "b16", "b16_n32", "lit_b16"])
def test_bf16_residual_stream_step(dev, which):
  """config.residual_stream = "bfloat16": the activations between the blocks and their gradients are
  bf16 (LayerNorm inputs, +residual GEMM epilogues, saved block inputs); everything else as before.
  Same bounds as the fp32-stream cases; the measured bf16-operand floor of each case is reported next
  to it (tools/bf16_residual_budget.py predicts +10-25 % on the worst tensors)."""
  if which == "tiny":
    image_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="map")
    text_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2)
    _run_case(dev, image_cfg, text_cfg, E=128, n=8, res=64, seq=16, vocab=100, floor=True,
              config=_cfg(residual_stream="bfloat16"), case="bf16 stream: siglip tiny n=8")
  elif which == "tiny_tok_lit":
    image_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="tok",
                     head_zeroinit=False)
    text_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2)
    _run_case(dev, image_cfg, text_cfg, E=128, n=8, res=64, seq=16, vocab=100, bias_init=-2.71,
              config=_cfg(schedule=LIT_SCHEDULE, residual_stream="bfloat16"), frozen=("img/",), floor=True,
              case="bf16 stream: LiT tiny frozen img")
  elif which == "b16":
    image_cfg = dict(variant="B/16", pool_type="map")
    text_cfg = dict(variant="B")
    _run_case(dev, image_cfg, text_cfg, E=768, n=8, res=224, seq=64, vocab=32_000, optimizer_check=False,   # (the fused Adam chain on this model is checked by the fp32-stream siblings)
              config=_cfg(residual_stream="bfloat16", microbatch=4, microbatch_keep="all", microbatch_light=True),
              case="bf16 stream: siglip B/16 n=8 microbatch=4 light")
  elif which == "b16_n32":   # what bench.py runs: B/16 + text-B, two-pass micro-batches with light contexts
    image_cfg = dict(variant="B/16", pool_type="map")
    text_cfg = dict(variant="B")
    # (no floor measurement here: two more fp64 oracle passes at n = 32; measured 0.0127 at a floor of 0.0108,
    #  profiles/r02_parity_report.jsonl - the default bounds hold with a wide margin)
    _run_case(dev, image_cfg, text_cfg, E=768, n=32, res=224, seq=64, vocab=32_000, optimizer_check=False,   # (the fused Adam chain on this model is checked by the fp32-stream siblings)
              config=_cfg(residual_stream="bfloat16", microbatch=8, microbatch_keep="all", microbatch_light=True),
              case="bf16 stream: siglip B/16 n=32 microbatch=8 light")
  else:                      # BASELINE configs[4] shapes
    image_cfg = dict(variant="B/16", pool_type="tok", head_zeroinit=False)
    text_cfg = dict(variant="B")
    # The ONE tensor of the whole suite outside SURVEY 8c's rel-L2 <= 3e-2, listed by name: the query kernel
    # of text block 5 on this 8-pair, 16-token batch measures 0.0338 at cosine 0.99943 on the bf16 stream
    # (0.0228 on the fp32 stream = its bf16-operand floor 0.0229; 0.1 % of the global gradient norm).  The
    # bf16 stream is an opt-in mode (not the reference's arithmetic, not what bench.py's `value` runs).
    _run_case(dev, image_cfg, text_cfg, E=768, n=8, res=224, seq=16, vocab=32_000, bias_init=-2.71, optimizer_check=False,   # (the fused Adam chain on this model is checked by the fp32-stream siblings)
              config=_cfg(schedule=LIT_SCHEDULE, residual_stream="bfloat16"), frozen=("img/",), floor=True,
              exceptions={"txt/Encoder_0/encoderblock_5/MultiHeadDotProductAttention_0/query/kernel": (4e-2, 0.999)},
              case="bf16 stream: LiT B/16 frozen img n=8")


def test_mu_variant_step(dev):
  """`mu/16` (models/vit.py:297-300: width 32, depth 1, mlp 128, 2 heads -> head dim 16), the variant the
  reference's own tests and the SURVEY's golden-vector plan use: the general attention kernels
  (attention_dh.hip) inside the full step."""
  image_cfg = dict(variant="mu/16", pool_type="map")
  text_cfg = dict(variant="mu")
  _run_case(dev, image_cfg, text_cfg, E=32, n=8, res=64, seq=16, vocab=100, floor=True, case="siglip mu/16 n=8")


def test_so400m_shapes_step(dev):
  """So400m/14 (width 1152, 16 heads -> head dim 72, mlp 4304, 14 x 14 patches: K = 588 padded to 592,
  256 tokens at 224 px) + text So400m, the flagship SigLIP size, cut to 2 blocks per tower for the CPU
  oracle: head dim 72 self-attention and MAP head, ragged GEMM shapes (N = 4304, 1152 = 4.5 x 256),
  the zero-padded stem and its gradient."""
  image_cfg = dict(variant="So400m/14", pool_type="map", depth=2)
  text_cfg = dict(variant="So400m", depth=2)
  _run_case(dev, image_cfg, text_cfg, E=1152, n=4, res=224, seq=16, vocab=32_000, floor=True,
            case="siglip So400m/14 depth2 n=4")




def test_lit_frozen_image_tower_step_tiny(dev):
  """LiT (configs/proj/image_text/siglip_lit_coco.py:79-104) on a toy width: the image tower is
  frozen by `schedule=[("img/.*", None), ...]` - no image-tower gradient, no Adam state, clip norm
  / l2_grads over the text tower + t + b only, l2_params including the frozen weights; the step
  runs FIRST on a dirty bf16 shadow (advisor r1: update_fn must refresh it itself)."""
  image_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="tok",
                   head_zeroinit=False)
  text_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2)
  _run_case(dev, image_cfg, text_cfg, E=128, n=8, res=64, seq=16, vocab=100, bias_init=-2.71,
            config=_cfg(schedule=LIT_SCHEDULE), frozen=("img/",), dirty_step=True, floor=True,
            case="LiT tiny frozen img")


def test_lit_frozen_image_tower_step_b16(dev):
  """BASELINE configs[4] shapes: frozen ViT-B/16 (pool 'tok': 197 tokens) + trainable text-B at
  16 tokens, bias_init -2.71, out_dim (None, 768), n = 8 (siglip_lit_coco.py:33,46,79-104 with the
  in-repo text transformer standing in for BERT)."""
  image_cfg = dict(variant="B/16", pool_type="tok", head_zeroinit=False)
  text_cfg = dict(variant="B")
  _run_case(dev, image_cfg, text_cfg, E=768, n=8, res=224, seq=16, vocab=32_000, bias_init=-2.71,
            config=_cfg(schedule=LIT_SCHEDULE), frozen=("img/",), dirty_step=True, floor=True,
            case="LiT B/16 frozen img n=8")


def test_b16_siglip_step_small_batch(dev):
  """The real ViT-B/16 + text-B SigLIP model (BASELINE config 3 shapes) at n=4."""
  image_cfg = dict(variant="B/16", pool_type="map")
  text_cfg = dict(variant="B")
  _run_case(dev, image_cfg, text_cfg, E=768, n=4, res=224, seq=64, vocab=32_000, floor=True)


def test_b16_siglip_step_n32_through_microbatches(dev):
  """The real B/16 + text-B model at n = 32 through the two-pass micro-batch path (4 micro-batches of
  8, light contexts) against the fp64 oracle on the whole batch - the production N=1 code path
  checked against the oracle rather than against itself."""
  image_cfg = dict(variant="B/16", pool_type="map")
  text_cfg = dict(variant="B")
  _run_case(dev, image_cfg, text_cfg, E=768, n=32, res=224, seq=64, vocab=32_000,
            config=_cfg(microbatch=8, microbatch_keep="all", microbatch_light=True),
            case="siglip B/16 n=32 microbatch=8 light")   # (no floor measurement: two more fp64 passes at n = 32; r02: 0.0108)


@pytest.mark.parametrize("light", [True, False])
def test_b16_depth2_n64_image_tower_on_gemm256(dev, light):
  """VERDICT r3 weak #1b: an end-to-end case whose IMAGE tower runs on the 256x256 kernel.  B/16 shapes,
  depth 2, n = 64: image token rows M = 64 x 196 = 12 544 = 49 x 256 (every smaller case has M % 256 != 0
  and takes the general 128x128 kernel), text rows 64 x 64 = 4096; fp32 residual stream; once with light
  contexts (the bench's: LayerNorm outputs and gelu(h) re-emitted by the backward kernels, BV_EPI_GELU /
  BV_EPI_GELU_BWD_EMIT) and once with full contexts (BV_EPI_GELU_GD / BV_EPI_MUL), both as ONE micro-batch of
  64 so every fused epilogue walks its multi-tile persistent schedule, against the fp64 oracle."""
  from big_vision_amd import ops
  image_cfg = dict(variant="B/16", pool_type="map", depth=2)
  text_cfg = dict(variant="B", depth=2)
  calls0 = ops.ctx_get("gemm256_fused")   # multi-tile walks with a fused epilogue through this stream's context so far
  _run_case(dev, image_cfg, text_cfg, E=768, n=64, res=224, seq=64, vocab=32_000,
            config=_cfg(microbatch=64, microbatch_keep="all", microbatch_light=light),
            case=f"siglip B/16 depth2 n=64 (image tower on gemm256), {'light' if light else 'full'} contexts")
  # forward + backward of 2 image blocks: fc1 GELU(_GD), out-proj / fc2 +residual, fc2 dX GELU'(MUL) + column sums
  assert ops.ctx_get("gemm256_fused") >= calls0 + 8, "the image tower did not run on the 256x256 kernel"


@pytest.mark.parametrize("stream", ["float32", "bfloat16"])
def test_b16_n32_bench_mode_gelu_free_contexts(dev, stream):
  """The context kind bench.py's N = 1 step can end up in when `microbatch_light="auto"` finds that full
  contexts do not fit: "g" = gelu(h) is not kept, the fc2 dX GEMM re-emits it (BV_EPI_GELU_BWD_EMIT) while
  the LayerNorm outputs stay.  B/16 + text-B, n = 32 as 4 micro-batches of 8, both residual streams,
  against the fp64 oracle on the whole batch (VERDICT r2 weak #2b: this mode had only a CPU dry run)."""
  image_cfg = dict(variant="B/16", pool_type="map")
  text_cfg = dict(variant="B")
  _run_case(dev, image_cfg, text_cfg, E=768, n=32, res=224, seq=64, vocab=32_000,
            optimizer_check=False,   # (checked on this model and batch by test_b16_siglip_step_n32_through_microbatches)
            config=_cfg(residual_stream=stream, microbatch=8, microbatch_keep="all", microbatch_light="g"),
            case=f"siglip B/16 n=32 microbatch=8 gelu(h)-free contexts, {stream} stream")


def test_l16_336_siglip_step_small_batch(dev):
  """BASELINE configs[3] shapes: ViT-L/16 at 336 px (441 tokens, width 1024, 16 heads) + text-L,
  E = 1024, at n = 2: the long-sequence attention kernels (L > 224), the width-1024 LayerNorm
  instantiation and ragged GEMM shapes (882 tokens) against the fp64 oracle.  Depth is cut from
  24 to 4 blocks per tower to keep the CPU oracle at ~20 s (the full-depth model passed with the
  same tolerances in 134 s; every block has the same shapes)."""
  image_cfg = dict(variant="L/16", pool_type="map", depth=4)
  text_cfg = dict(variant="L", depth=4)
  # The text key-projection gradient (cancellation-heavy: every row of dS sums to zero) showed rel-L2
  # 0.107 at cosine 0.994 in round 1 - caused by delta = rowsum(dO o O) with the bf16-rounded O; with the
  # exact fp32 delta of attention3.hip it measures 0.0145 (bf16-operand floor 0.0148): default bounds.
  _run_case(dev, image_cfg, text_cfg, E=1024, n=2, res=336, seq=64, vocab=32_000, floor=True)


def test_l16_336_siglip_step_full_depth(dev):
  """BASELINE configs[3] at its REAL depth: ViT-L/16@336 (24 blocks, 441 tokens, width 1024) + text-L (24 blocks),
  n = 2, against the fp64 oracle - the case the depth-2/4 tests above stand in for (the full depth was
  only claimed in a docstring; r3: un-gated, ~3 min of fp64 oracle inside the default GPU suite).  Default bounds."""
  image_cfg = dict(variant="L/16", pool_type="map")
  text_cfg = dict(variant="L")
  # optimizer_check=False: loss, logits and every gradient of the 48 blocks are what this case is for; the optimizer
  # chain on these widths is checked by the depth-4 case above (profiles/r05_slowest_test_profile.txt: the Python
  # optimizer oracle over 652 M fp64 parameters was 56 of this test's 146 s)
  _run_case(dev, image_cfg, text_cfg, E=1024, n=2, res=336, seq=64, vocab=32_000, optimizer_check=False,
            case="siglip L/16@336 FULL depth (24 + 24 blocks) n=2")


def test_l16_336_siglip_step_n16(dev):
  """Same shapes at n = 16 (depth 2): the evidence the round-1 review asked for - the worst tensors of
  the two-sample case re-measured on a larger batch, each next to its bf16-operand floor."""
  image_cfg = dict(variant="L/16", pool_type="map", depth=2)
  text_cfg = dict(variant="L", depth=2)
  _run_case(dev, image_cfg, text_cfg, E=1024, n=16, res=336, seq=64, vocab=32_000, floor=True,
            case="siglip L/16@336 depth2 n=16")


@pytest.mark.parametrize("micro,frozen", [(0, None), (4, None), (0, "img"), (4, "img"), (0, "txt")])
def test_towers_on_two_streams_change_nothing(dev, micro, frozen):
  """config.tower_streams = 2 (the default since round 6; 1 = both towers on the caller's stream): the text tower runs on a side stream beside the image tower, forward and
  backward, with and without micro-batches.  Same kernels on the same inputs: loss, gradient norm and every gradient
  agree with the one-stream step (LayerNorm scale / bias gradients are fp32 atomics: summation order only).
  frozen = "img" (LiT) / "txt": the forward still forks, the backward of the one trainable tower runs on the caller's stream
  and consumes text contexts that were allocated on the side stream."""
  import bv_oracle as O
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  image_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="map")
  text_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, vocab_size=100)
  image, text = O.synthetic_batch(1, 8, 64, 16, 100)
  batch = {"image": image.to(dev), "labels": text.to(dev)}
  res = {}
  for streams in (1, 2):
    model = two_towers.Model(image=image_cfg, text=text_cfg, out_dim=(None, 128), temperature_init=10.0, bias_init=-10.0)
    config = _cfg(tower_streams=streams, microbatch=micro, microbatch_keep="all")
    if frozen:
      config.schedule = [(f"{frozen}/.*", None), (".*", dict(decay_type="cosine", warmup_steps=2))]
    state, _ = siglip.make_train_state(model, config, tuple(image.shape), tuple(text.shape), rng=0, total_steps=config.total_steps)
    fn = siglip.make_update_fn(model, config)
    if frozen:
      assert state["params"].store.frozen and all(n.startswith(f"{frozen}/") for n in state["params"].store.frozen)
    for _ in range(2):     # two steps: the second one re-transposes the weight images before the fork
      state, meas = fn(state, None, batch)
    torch.cuda.synchronize()
    res[streams] = (meas["training_loss"].item(), meas["l2_grads"].item(), state["params"].store.grad.clone(),
                    state["params"].store.master.clone())
  (l1, g1, v1, p1), (l2, g2, v2, p2) = res[1], res[2]
  assert abs(l1 - l2) <= 1e-6 * abs(l1) and abs(g1 - g2) <= 1e-5 * g1, (l1, l2, g1, g2)
  assert (v1 - v2).norm().item() <= 1e-5 * v1.norm().item()
  assert (p1 - p2).abs().max().item() <= 2.1e-3      # (two Adam steps; a ~0 gradient may flip a sign-like first update)
  assert ((p1 - p2).abs() > 1e-6).double().mean().item() <= 0.01


@pytest.mark.parametrize("keep,light", [(0, False), (1, False), ("all", False), ("auto", "auto"),
                                        ("all", True), (2, True)])
def test_microbatched_step_equals_full_batch(dev, keep, light):
  """Two-pass micro-batching (config.microbatch / microbatch_keep / microbatch_light) is a
  pure memory/compute trade: loss and gradients must match the single-pass step on the same
  batch; "light" contexts (LayerNorm outputs and gelu(h) re-derived by the backward) feed the
  backward the same bits as full contexts, so the gradients agree to fp32 accumulation-order
  noise (LayerNorm scale/bias and bias gradients are summed with fp32 atomics)."""
  import bv_oracle as O
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  from big_vision_amd import utils as u
  image_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="map")
  text_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, vocab_size=100)
  image, text = O.synthetic_batch(1, 8, 64, 16, 100)
  batch = {"image": image.to(dev), "labels": text.to(dev)}

  def run(**kw):
    model = two_towers.Model(image=image_cfg, text=text_cfg, out_dim=(None, 128), temperature_init=10.0,
                             bias_init=-10.0)
    config = _cfg(**kw)
    state, _ = siglip.make_train_state(model, config, tuple(image.shape), tuple(text.shape), rng=0,
                                       total_steps=config.total_steps)
    state, meas = siglip.make_update_fn(model, config)(state, None, batch)
    grads = {k: v.detach().cpu().double().clone()
             for k, v in u.tree_flatten_with_names(state["params"].store.tree("grad"))[0]}
    return meas["training_loss"].item(), grads

  loss_full, g_full = run()
  loss_mb, g_mb = run(microbatch=2, microbatch_keep=keep, microbatch_light=light)
  assert abs(loss_full - loss_mb) <= 1e-5 * abs(loss_full)
  gn = math.sqrt(sum((v ** 2).sum().item() for v in g_full.values()))
  for k, v in g_full.items():
    assert (v - g_mb[k]).norm().item() <= 2e-3 * max(v.norm().item(), 1e-3 * gn), k
  if light is True:
    loss_ref, g_ref = run(microbatch=2, microbatch_keep=keep, microbatch_light=False)
    assert loss_ref == loss_mb
    for k, v in g_ref.items():
      assert (v - g_mb[k]).norm().item() <= 1e-5 * max(v.norm().item(), 1e-3 * gn), \
          f"light context changed {k}"


def test_scan_layout_model_steps_like_the_pyloop_model(dev):
  """scan=True only changes how parameters are PRESENTED (stacked `encoderblock` leaves, strided
  views of the same storage): same seed -> same loss, and the stacked gradient leaves are the
  per-block gradients stacked."""
  import bv_oracle as O
  from big_vision_amd.models import vit
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  from big_vision_amd import utils as u
  image_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="map")
  text_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, vocab_size=100)
  image, text = O.synthetic_batch(1, 8, 64, 16, 100)
  batch = {"image": image.to(dev), "labels": text.to(dev)}

  def run(scan):
    model = two_towers.Model(image=dict(image_cfg, scan=scan), text=dict(text_cfg, scan=scan), out_dim=(None, 128),
                             temperature_init=10.0, bias_init=-10.0)
    config = _cfg()
    state, _ = siglip.make_train_state(model, config, tuple(image.shape), tuple(text.shape), rng=0,
                                       total_steps=config.total_steps)
    state, meas = siglip.make_update_fn(model, config)(state, None, batch)
    g = u.tree_map(lambda v: v.detach().cpu().clone().numpy(), dict(state["params"].store.tree("grad")))
    p = u.tree_map(lambda v: v.detach().cpu().clone().numpy(), dict(state["params"]))
    return meas["training_loss"].item(), g, p

  loss_a, g_a, p_a = run(False)
  loss_b, g_b, p_b = run(True)
  assert abs(loss_a - loss_b) <= 1e-6 * abs(loss_a)
  for tree_a, tree_b in ((g_a, g_b), (p_a, p_b)):
    want = dict(tree_a)
    want["img"] = vit.pyloop_to_scan(tree_a["img"])
    want["txt"] = dict(tree_a["txt"])
    want["txt"]["Encoder_0"] = vit.pyloop_to_scan({"Transformer": tree_a["txt"]["Encoder_0"]})["Transformer"]
    fa, fb = dict(u.tree_flatten_with_names(want)[0]), dict(u.tree_flatten_with_names(tree_b)[0])
    assert fa.keys() == fb.keys()
    gn = math.sqrt(sum(float((v ** 2).sum()) for v in fa.values()))
    for k in fa:
      assert fa[k].shape == fb[k].shape, k
      assert float(np.linalg.norm(fa[k] - fb[k])) <= 1e-5 * max(float(np.linalg.norm(fa[k])), 1e-3 * gn), k



def test_frozen_tower_keeps_no_context(dev):
  """LiT: the frozen image tower's forward runs context-free inside a SAVING step (single-output GELU epilogue, activations
  freed as it goes): its context slot is empty, its embeddings are the bits of a saving forward of the same tower, and a
  cotangent handed to `bwd` for it is ignored instead of differentiating through nothing."""
  import bv_oracle as O
  from big_vision_amd import ops
  from big_vision_amd.models.proj.image_text import two_towers
  from big_vision_amd.trainers.proj.image_text import siglip
  image_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="map")
  text_cfg = dict(width=128, depth=2, mlp_dim=256, num_heads=2, vocab_size=100)
  image, text = O.synthetic_batch(1, 8, 64, 16, 100)
  image, text = image.to(dev), text.to(dev)
  for name, schedule in (("lit", LIT_SCHEDULE), ("all", dict(decay_type="cosine", warmup_steps=2))):
    model = two_towers.Model(image=image_cfg, text=text_cfg, out_dim=(None, 128), temperature_init=10.0, bias_init=-10.0)
    config = _cfg(schedule=schedule)
    state, _ = siglip.make_train_state(model, config, tuple(image.shape), tuple(text.shape), rng=0, total_steps=config.total_steps)
    ex = model.executor(state["params"].store, "", tuple(image.shape), tuple(text.shape))
    zimg, ztxt, _, ctx = ex.fwd(image, text, save=True)
    assert (ctx["img"][0] is None) == (name == "lit") and ctx["txt"][0] is not None
    z, _, c = ex.img.fwd(image, True, False)        # the same tower, saving (two-output GELU epilogue)
    assert c is not None and torch.equal(ops.l2norm_fwd(z)[0], zimg), name
    if name == "lit":
      ex.bwd(ctx, torch.ones_like(zimg), torch.ones_like(ztxt))     # the image cotangent is dropped, the text tower differentiates
    torch.cuda.synchronize()
