"""Executes the reference's own optimizer factory and writes what it produces as golden fixtures.

TEST INFRASTRUCTURE (oracle/): run by hand or by tests/test_reference_optax_cpu.py, never by the product.

  python oracle/run_reference_optax.py [out_dir]          (default tests/golden/)

`/root/reference/big_vision/optax.py` (`make` :75-149, `scale_by_adafactor` :187-216, `get_count` :30-41) and
`big_vision/utils.py` (`create_learning_rate_schedule` :1070-1143, `make_mask_trees` :1195-1212,
`tree_flatten_with_names` :642-668) are imported UNMODIFIED, from where they lie, over the stand-ins of `oracle/refshim/`
(optax / jax are not installed; `refshim/optax/__init__.py` says what that does and does not pin: the reference's
WIRING of the chain - order, masks, frozen parameters, multipliers, schedules, the BigVision Adafactor's arguments - over
restated optax stages).  For every case: `tx, sched_fns = bv_optax.make(config, params, sched_kw=...)`, `tx.init`, then
STEPS updates on seeded gradients with `optax.apply_updates` in between.

`refoptax_<case>.npz`: `param/<name>` (initial), `grad/<step>/<name>`, `update/<step>/<name>`,
`sched/<i>` (schedule function i at steps 0 .. total_steps), `state/<name>` = the optimizer state after the last step
under the names `u.tree_flatten_with_names` gives it (the checkpoint contract), `meta` (JSON: config, sched_kw, the
ordered state names and shapes, `get_count`).  `refoptax_schedules.npz`: `create_learning_rate_schedule` alone on a grid
of configurations.  `refoptax_state_names.json`: the state names / shapes `make` produces for the parameter trees of two
model fixtures (tests/golden/refwiring_two_map_last_bias.npz, refwiring_two_scan.npz).  float64 throughout (the
schedule's final cast to float32 is a dtype detail the stand-ins do not model)."""
import json
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REFERENCE = os.environ.get("BV_REFERENCE_ROOT", "/root/reference")
STEPS = 3


def _isolate_imports():
  """`big_vision` must resolve to the REFERENCE, not to this repository's alias package of the same name."""
  drop = {REPO, ""}
  sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != REPO and p not in drop]
  sys.path.insert(0, REFERENCE)
  sys.path.insert(0, os.path.join(HERE, "refshim"))
  for m in list(sys.modules):
    if m == "big_vision" or m.startswith("big_vision.") or m in ("optax", "jax", "flax") or m.startswith(("jax.", "flax.", "optax.")):
      del sys.modules[m]


# a parameter tree with every kind of leaf the chain treats differently: factored 2-D / 3-D kernels (Adafactor: the two
# largest axes, both >= min_dim_size_to_factor), an unfactored 4-D stem kernel and small kernel, biases, an embedding
# table (no `kernel` in its name: no weight decay by default), the scalars t / b
SHAPES = {
    "img/embedding/kernel": (2, 2, 3, 32), "img/embedding/bias": (32,),
    "img/Transformer/encoderblock_0/MultiHeadDotProductAttention_0/query/kernel": (32, 2, 34),
    "img/Transformer/encoderblock_0/MultiHeadDotProductAttention_0/query/bias": (2, 34),
    "img/Transformer/encoderblock_0/MlpBlock_0/Dense_0/kernel": (32, 48),
    "img/Transformer/encoderblock_0/MlpBlock_0/Dense_0/bias": (48,),
    "img/head/kernel": (8, 16),
    "txt/Embed_0/embedding": (12, 32), "txt/pos_embedding": (1, 8, 32),
    "txt/head/kernel": (32, 33), "txt/head/bias": (33,),
    "t": (1,), "b": (1,),
}
SCHED_KW = dict(total_steps=10, batch_size=8, data_size=100)

CASES = {
    # optax.py:99-149 with scale_by_adam: clip -> adam -> lr -> wd (default mask: */kernel) -> schedule -> -1
    "adam_clip_wd": dict(lr=1e-3, wd=1e-2, grad_clip_norm=1.0, optax_name="scale_by_adam",
                         optax=dict(b2=0.95, mu_dtype="bfloat16"), schedule=dict(decay_type="cosine", warmup_steps=2)),
    # frozen tower (schedule None), two live schedules (one with `mult`), lr_mults and wd_mults with first-match-wins masks
    "adam_frozen_mults": dict(lr=3e-3, wd=1e-3, optax_name="scale_by_adam", optax=dict(eps=1e-6),
                              schedule=[("img/embedding/.*", None), ("txt/.*", dict(decay_type="linear", warmup_percent=0.2, linear_end=0.1, mult=0.5)),
                                        (".*", dict(decay_type="rsqrt", timescale=5, warmup_steps=1, cooldown_steps=3))],
                              lr_mults=[("txt/head/.*", 3.0), (".*/bias", 0.5), (".*", 1.0)],
                              wd_mults=[(".*/kernel$", 1.0), (".*embedding$", 0.25)]),
    # the BigVision Adafactor with its defaults (factored second moment, bf16 momentum 0.9, no block clip)
    "adafactor_default": dict(lr=1e-2, wd=1e-3, grad_clip_norm=2.0, optax_name="big_vision.scale_by_adafactor",
                              schedule=dict(decay_type="rsqrt", timescale=4, warmup_steps=2)),
    # every argument moved: block clip, no momentum, a small factoring threshold (the [8, 16] kernel factors), offset, cap
    "adafactor_clip_nomom": dict(lr=1e-2, optax_name="big_vision.scale_by_adafactor",
                                 optax=dict(clipping_threshold=0.7, momentum=None, min_dim_size_to_factor=8, decay_offset=-2,
                                            beta2_cap=0.9, decay_rate=0.6, eps=1e-20),
                                 schedule=dict(decay_type="stair", steps=[1, 2], mults=[0.5, 0.25])),
    "adafactor_f32mom_frozen": dict(lr=5e-3, wd=1e-2, optax_name="big_vision.scale_by_adafactor",
                                    optax=dict(dtype_momentum="float32", momentum=0.8),
                                    schedule=[("txt/.*", None), (".*", dict(decay_type="cosine", warmup_steps=1))]),
    # plain SGD spelled the reference's way (optax.py:227: big_vision.sgd = identity)
    "sgd": dict(lr=0.1, wd=1e-2, optax_name="big_vision.sgd", schedule=dict(decay_type="polynomial", power=2, end=0.01)),
}

SCHEDULES = {
    "cosine_warm": dict(decay_type="cosine", warmup_steps=3),
    "cosine_pct_cool": dict(decay_type="cosine", warmup_percent=0.1, cooldown_percent=0.2),
    "linear_end": dict(decay_type="linear", warmup_steps=2, linear_end=0.25),
    "poly2": dict(decay_type="polynomial", power=2, end=0.1, warmup_examples=16),
    "rsqrt": dict(decay_type="rsqrt", timescale=7, warmup_steps=4, cooldown_steps=5),
    "rsqrt_shift_epochs": dict(decay_type="rsqrt", timescale_epochs=0.5, shift=3, warmup_epochs=0.25),
    "stair": dict(decay_type="stair", steps=[5, 12], mults=[0.1, 0.01], warmup_steps=2),
    "batchscaled": dict(decay_type="cosine", scale_with_batchsize=True, base=0.3),
}
SCHEDULE_KW = dict(total_steps=20, batch_size=8, data_size=100)


def _nest(flat):
  tree = {}
  for k, v in flat.items():
    node = tree
    *parents, last = k.split("/")
    for p in parents:
      node = node.setdefault(p, {})
    node[last] = v
  return tree


def _config(d):
  from ml_collections import ConfigDict
  return ConfigDict(d)


def run_case(name, out_dir):
  import numpy as np
  import optax
  import big_vision.optax as bv_optax
  import big_vision.utils as u
  cfg = CASES[name]
  g = np.random.default_rng([23, zlib.crc32(name.encode())])
  params = _nest({k: g.normal(0.0, 0.3, s) for k, s in SHAPES.items()})
  arrays = {f"param/{k}": v for k, v in u.tree_flatten_with_names(params)[0]}
  tx, sched_fns = bv_optax.make(_config(cfg), params, sched_kw=dict(SCHED_KW))
  state = tx.init(params)
  for step in range(STEPS):
    # gradient scales differ by leaf and step so that the clip triggers on some steps and not on others
    grads = _nest({k: g.normal(0.0, 0.004 * (1 + 3 * step) * (1 + (i % 3)), s) for i, (k, s) in enumerate(SHAPES.items())})
    updates, state = tx.update(grads, state, params)
    params = optax.apply_updates(params, updates)
    arrays.update({f"grad/{step}/{k}": v for k, v in u.tree_flatten_with_names(grads)[0]})
    arrays.update({f"update/{step}/{k}": v for k, v in u.tree_flatten_with_names(updates)[0]})
  for i, fn in enumerate(sched_fns):
    arrays[f"sched/{i}"] = np.array([float(fn(s)) for s in range(SCHED_KW["total_steps"] + 1)])
  flat_state = u.tree_flatten_with_names(state)[0]
  arrays.update({f"state/{k}": np.asarray(v) for k, v in flat_state})
  meta = dict(case=name, config=cfg, sched_kw=SCHED_KW, steps=STEPS, count=int(bv_optax.get_count(state)),
              state_names=[k for k, _ in flat_state], state_shapes={k: list(np.shape(v)) for k, v in flat_state},
              n_schedules=len(sched_fns))
  arrays["meta"] = np.frombuffer(json.dumps(meta, sort_keys=True).encode(), np.uint8)
  np.savez_compressed(os.path.join(out_dir, f"refoptax_{name}.npz"), **arrays)
  return meta


def run_schedules(out_dir):
  import numpy as np
  import big_vision.utils as u
  arrays = {}
  for name, kw in SCHEDULES.items():
    fn = u.create_learning_rate_schedule(**SCHEDULE_KW, **kw)
    arrays[name] = np.array([float(fn(s)) for s in range(SCHEDULE_KW["total_steps"] + 1)])
  arrays["meta"] = np.frombuffer(json.dumps(dict(schedules=SCHEDULES, kw=SCHEDULE_KW), sort_keys=True).encode(), np.uint8)
  np.savez_compressed(os.path.join(out_dir, "refoptax_schedules.npz"), **arrays)


MODEL_STATE_CASES = {
    # (model fixture whose parameter tree is used, optimizer config): names and shapes of the state `make` builds
    "two_map_last_bias/adam": ("refwiring_two_map_last_bias", dict(lr=1e-3, wd=1e-4, optax_name="scale_by_adam", optax=dict(mu_dtype="bfloat16"),
                                                                    grad_clip_norm=1.0, schedule=dict(decay_type="cosine", warmup_steps=1))),
    "two_map_last_bias/adam_frozen_img": ("refwiring_two_map_last_bias", dict(lr=1e-3, optax_name="scale_by_adam",
                                                                              schedule=[("img/.*", None), (".*", dict(decay_type="cosine"))])),
    "two_map_last_bias/adafactor": ("refwiring_two_map_last_bias", dict(lr=1e-3, optax_name="big_vision.scale_by_adafactor",
                                                                         schedule=dict(decay_type="cosine", warmup_steps=1))),
    "two_scan/adafactor": ("refwiring_two_scan", dict(lr=1e-3, optax_name="big_vision.scale_by_adafactor", optax=dict(momentum=None),
                                                      schedule=dict(decay_type="cosine"))),
}


def run_model_state_names(out_dir):
  import numpy as np
  import big_vision.optax as bv_optax
  import big_vision.utils as u
  out = {}
  for name, (fixture, cfg) in MODEL_STATE_CASES.items():
    z = np.load(os.path.join(REPO, "tests", "golden", f"{fixture}.npz"))
    params = _nest({k[len("param/"):]: np.asarray(z[k], np.float64) for k in z.files if k.startswith("param/")})
    tx, _ = bv_optax.make(_config(cfg), params, sched_kw=dict(SCHED_KW))
    flat = u.tree_flatten_with_names(tx.init(params))[0]
    out[name] = dict(fixture=fixture, config=cfg, state=[[k, list(np.shape(v))] for k, v in flat])
  # replace_frozen (optax.py:44-51) on an integer-leaved tree with the schedules of the cases above
  tree = _nest({k: i + 1 for i, k in enumerate(SHAPES)})
  out["__replace_frozen__"] = {}
  for cname in ("adam_frozen_mults", "adafactor_f32mom_frozen", "adam_clip_wd"):
    sched = CASES[cname]["schedule"]
    res = bv_optax.replace_frozen(sched, tree, 0)
    out["__replace_frozen__"][cname] = {"schedule": sched, "result": [[k, int(v)] for k, v in u.tree_flatten_with_names(res)[0]]}
  with open(os.path.join(out_dir, "refoptax_state_names.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)


def main():
  out_dir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "tests", "golden")
  os.makedirs(out_dir, exist_ok=True)
  _isolate_imports()
  for name in CASES:
    meta = run_case(name, out_dir)
    print(name, meta["count"], len(meta["state_names"]), "state leaves")
  run_schedules(out_dir)
  run_model_state_names(out_dir)


if __name__ == "__main__":
  main()
