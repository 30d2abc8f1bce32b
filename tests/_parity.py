"""Shared gradient-parity bookkeeping of the end-to-end GPU tests.

Every e2e case compares each parameter gradient of the HIP step with the fp64 oracle's and
records (cosine, rel-L2) per tensor.  The worst tensors of every case are printed (pytest -s /
failure output) and appended to `gpurun_out/parity_report.jsonl` when that directory can be
written, so the tolerances stated in the tests are the MEASURED ones plus margin rather than
guesses ("report the worst per-tensor (cosine, rel-L2) for every e2e case and
tighten to what is measured").

Tolerances (bf16 MFMA operands, fp32 accumulate / residual stream / LayerNorm / softmax / loss vs
the fp64 oracle; SURVEY.md §8c proposed cosine >= 0.999 and rel-L2 <= 3e-2):
  COS_MIN, REL_MAX below are the defaults every case uses; a case may pass a different bound only
  with a measured reason in its docstring.
Measured on MI355X with attention3.hip (exact fp32 softmax-backward delta), round 2: worst rel-L2 per
case 0.0075-0.0223 (every case within ~1x its bf16-operand floor; B/16 n=4 0.0138, L/16@336 n=2 0.0145,
LiT B/16 n=8 0.0223), worst cosine 0.99976 - all inside the SURVEY bounds; round 1 (delta from the
bf16 O) had 0.05-0.107 on the same cases.
Tensors whose reference gradient is below SMALL x the global gradient norm (e.g. the key bias,
whose gradient is exactly zero by the shift invariance of softmax) are held to an absolute error
of ABS_SMALL x the global norm instead: their relative error is noise over noise.  Round 6: every
leaf that takes this branch is listed in the report (name, share of the global norm, error in
units of it); over the 35 end-to-end cases these are 1-78 leaves per case - key / query biases and
kernels, a few LayerNorm_1 scales - and the worst error measured is 3.5e-5 of the global norm
(profiles/r06_parity_report.jsonl), so ABS_SMALL went from 2e-3 to 1e-4: a leaf right at the
threshold is held to 10 % of its own norm, and no leaf may contribute more than 1e-4 of the global
gradient norm in error.
"""
import json
import math
import os

COS_MIN = 0.999
REL_MAX = 3e-2
SMALL = 1e-3
ABS_SMALL = 1e-4
FLOOR_AUTO_MAX_PARAMS = 20_000_000   # bf16_floor: larger models measure the floor only under BV_PARITY_FLOOR=1

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# Round 3: the measured bf16-operand floor is REPORTED next to every tensor but no
# longer widens a bound.  A tensor above REL_MAX / below COS_MIN fails unless the case lists it by
# name in `exceptions` with its own stated bound (a visible, per-tensor exception in the test body).


def bf16_floor(loss_closure, params64):
  """Noise floor of the PRESCRIBED arithmetic (bf16 MFMA operands, fp32+ accumulate): gradients of
  the same oracle with every contraction's operands / cotangents rounded to bfloat16
  (bv_oracle.bf16_operands) vs the fp64 gradients already sitting in params64[..].grad.
  Returns {name: (rel-L2, cosine)}."""
  import bv_oracle as O
  from big_vision_amd import utils as u
  # The floor is REPORTED next to each tensor, never a bound, and costs one more oracle forward + backward (with rounding
  # hooks on every contraction: slower than the fp64 pass itself).  For the real-size models (B/16, L/16@336, So400m,
  # BERT-base: 30-80 s of host time per case on the GPU box) it is therefore measured only on request - BV_PARITY_FLOOR=1,
  # which tools/gpu_final.sh sets for the runs whose report is committed under profiles/ - so that the plain
  # `pytest -m gpu` the driver runs stays ~2 minutes shorter.  Toy-width cases always measure it.
  if (sum(v.numel() for _, v in u.tree_flatten_with_names(params64)[0]) > FLOOR_AUTO_MAX_PARAMS
      and os.environ.get("BV_PARITY_FLOOR") != "1"):
    return {}
  ref = {k: v.grad.clone() for k, v in u.tree_flatten_with_names(params64)[0] if v.grad is not None}
  p2 = O.recover_tree([(k, v.detach().clone().requires_grad_(k in ref)) for k, v in u.tree_flatten_with_names(params64)[0]])
  with O.bf16_operands():
    loss_closure(p2).backward()
  out = {}
  for k, v in u.tree_flatten_with_names(p2)[0]:
    if k not in ref:
      continue
    gr, g = ref[k], (v.grad if v.grad is not None else 0 * ref[k])
    nr = gr.norm().item()
    if nr == 0:
      continue
    out[k] = ((g - gr).norm().item() / nr, (g * gr).sum().item() / (g.norm().item() * nr + 1e-30))
  return out


def compare_grads(case, gref, gours, *, cos_min=COS_MIN, rel_max=REL_MAX, frozen=(), floor=None, exceptions=None):
  """gref / gours: {leaf name: fp64 cpu tensor}.  `frozen`: name prefixes that must be ABSENT from
  gours (no gradient is ever produced for frozen leaves).  `floor` ({name: (rel, cos)} from
  bf16_floor): reported next to each tensor, never used as a bound.  `exceptions` ({leaf name:
  (rel_max, cos_min)}): tensors held to their own stated bound instead of the case's.
  Returns (global norm over the trainable leaves, sorted worst list); raises AssertionError naming
  every offending tensor."""
  is_frozen = lambda k: any(k.startswith(p) for p in frozen)
  leaked = [k for k in gours if is_frozen(k)]
  assert not leaked, f"{case}: frozen leaves have gradients: {leaked[:4]}"
  live = {k: v for k, v in gref.items() if not is_frozen(k)}
  missing = [k for k in live if k not in gours]
  assert not missing, f"{case}: no gradient for {missing[:4]}"
  gnorm = math.sqrt(sum((v ** 2).sum().item() for v in live.values()))
  rows, bad, small = [], [], []
  for k, gr in live.items():
    go = gours[k]
    nr = gr.norm().item()
    err = (go - gr).norm().item()
    if nr < SMALL * gnorm:
      # the loose branch ("the report does not list which leaves took it"): recorded by name with
      # the tensor's share of the global norm and its error in units of that norm
      small.append((err / gnorm, k, nr / gnorm))
      if err > ABS_SMALL * gnorm:
        bad.append(f"{k}: small tensor, abs err {err / gnorm:.2e} of the global norm")
      continue
    cos = (go * gr).sum().item() / (go.norm().item() * nr + 1e-30)
    rel = err / nr
    frel, fcos = (floor or {}).get(k, (0.0, 1.0))
    rows.append((rel, cos, k, nr / gnorm, frel, fcos))
    rmax, cmin = (exceptions or {}).get(k, (rel_max, cos_min))
    if not (cos >= cmin and rel <= rmax):
      bad.append(f"{k}: cosine {cos:.5f} (>= {cmin:.5f}) rel-L2 {rel:.4f} (<= {rmax:.4f}) "
                 f"share of global norm {nr / gnorm:.3f}, bf16 floor rel {frel:.4f}")
  rows.sort(reverse=True)
  small.sort(reverse=True)
  report(case, rows, gnorm, cos_min, rel_max, small)
  if os.environ.get("BV_PARITY_REPORT_ONLY"):   # diagnostics runs: collect the table for every case, assert nothing
    if bad:
      print(f"[parity] {case}: WOULD FAIL:\n  " + "\n  ".join(bad))
    return gnorm, rows
  assert not bad, f"{case}: gradient parity failed (cos >= {cos_min}, rel-L2 <= {rel_max}):\n  " + "\n  ".join(bad)
  return gnorm, rows


def report(case, rows, gnorm, cos_min, rel_max, small=()):
  worst = rows[:5]
  print(f"\n[parity] {case}: {len(rows)} tensors, global grad norm {gnorm:.4e}; worst rel-L2 / cosine:")
  for rel, cos, k, share, frel, fcos in worst:
    print(f"[parity]   {k:70s} rel-L2 {rel:.4f} cos {cos:.6f} share {share:.3f}"
          + (f"  [bf16 floor rel {frel:.4f} cos {fcos:.6f}]" if frel else ""))
  if small:
    print(f"[parity]   {len(small)} tensors below {SMALL:g} of the global norm (absolute bound {ABS_SMALL:g} of it); "
          f"largest error {small[0][0]:.2e} of the global norm: {small[0][1]} (its share {small[0][2]:.2e})")
  rec = {"case": case, "tensors": len(rows), "bounds": {"cos_min": cos_min, "rel_max": rel_max},
         "small": {"count": len(small), "threshold_share": SMALL, "abs_bound_of_global_norm": ABS_SMALL,
                   "worst_abs_err_of_global_norm": small[0][0] if small else 0.0,
                   "leaves": [{"name": k, "share": sh, "abs_err_of_global_norm": e} for e, k, sh in small]},
         "worst_rel": max((r[0] for r in rows), default=0.0), "worst_cos": min((r[1] for r in rows), default=1.0),
         "worst": [{"name": k, "rel_l2": rel, "cos": cos, "share": share, "floor_rel": frel, "floor_cos": fcos}
                   for rel, cos, k, share, frel, fcos in worst]}
  out = os.path.join(ROOT, "gpurun_out")
  try:
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_report.jsonl"), "a") as f:
      f.write(json.dumps(rec) + "\n")
  except OSError:
    pass
