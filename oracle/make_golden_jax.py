"""Golden vectors from the REFERENCE ITSELF (JAX): pins oracle/bv_oracle.py to big_vision.

This build container has no jax / flax / optax, so the oracle is pinned to HuggingFace ports and to
the reference's numpy-only pieces ("parity unpinned against the JAX reference").  This
script closes that gap on ANY host that has `jax flax optax ml_collections numpy` (CPU is enough)
and a checkout of google-research/big_vision:

    python oracle/make_golden_jax.py --reference /path/to/big_vision_repo [--b16]

It imports the reference's own modules - `big_vision.models.proj.image_text.two_towers`
(models/vit.py, text_transformer.py underneath), `big_vision.optax`, `big_vision.utils` - runs
  * the forward of the two-tower model on seeded synthetic inputs (zimg, ztxt, out["t"], out["b"]),
  * the sigmoid loss exactly as trainers/proj/image_text/siglip.py:291-306 states it (the closure
    there is not importable; its eight lines are repeated below with jnp),
  * jax.value_and_grad of that loss w.r.t. every parameter,
  * one `tx.update` + `optax.apply_updates` of `bv_optax.make(config, ...)` (Adam + clip + wd + cosine),
and writes tests/golden/siglip_jax_<tag>.npz with the inputs, the flattened parameters (names as
`big_vision.utils.tree_flatten_with_names` gives them), and all results in float32/float64.
`tests/test_oracle.py::test_oracle_matches_jax_reference_when_present` consumes the file when it
exists (and says so when it does not).  Commit the .npz together with the jax/flax/optax versions
it prints.  TEST INFRASTRUCTURE - nothing in the product imports this.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CONFIGS = {
    "tiny": dict(image=dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="map"),
                 text=dict(width=128, depth=2, mlp_dim=256, num_heads=2, vocab_size=100),
                 out_dim=(None, 128), res=32, seq=16, n=4, grads="all"),
    "tiny_tok": dict(image=dict(width=128, depth=1, mlp_dim=256, num_heads=2, patch_size=(16, 16), pool_type="tok"),
                     text=dict(width=128, depth=1, mlp_dim=256, num_heads=2, vocab_size=64),
                     out_dim=(None, 128), res=48, seq=8, n=6, bias_init=-2.71, grads="all"),
    "b16": dict(image=dict(variant="B/16", pool_type="map"), text=dict(variant="B", vocab_size=32_000),
                out_dim=(None, 768), res=224, seq=64, n=2, grads="some"),
}


def synthetic_batch(seed, n, res, seq, vocab):
  """Same generator as bv_oracle.synthetic_batch would need torch; plain numpy here, stored in the file."""
  rng = np.random.RandomState(seed)
  image = rng.uniform(-1, 1, (n, res, res, 3)).astype(np.float32)
  text = rng.randint(2, vocab, (n, seq)).astype(np.int32)
  lens = rng.randint(4, seq, (n,))
  text = np.where(np.arange(seq)[None, :] >= lens[:, None], 1, text).astype(np.int32)
  return image, text


def run(tag, cfg, reference):
  sys.path.insert(0, reference)
  import jax
  import jax.numpy as jnp
  import flax
  import optax
  import ml_collections
  from big_vision import optax as bv_optax
  from big_vision import utils as u
  from big_vision.models.proj.image_text import two_towers

  jax.config.update("jax_enable_x64", False)
  model = two_towers.Model(image=cfg["image"], text=cfg["text"], out_dim=cfg["out_dim"],
                           temperature_init=10.0, bias_init=cfg.get("bias_init", -10.0))
  image, text = synthetic_batch(1, cfg["n"], cfg["res"], cfg["seq"], cfg["text"]["vocab_size"])
  params = model.init(jax.random.PRNGKey(0), jnp.asarray(image), jnp.asarray(text))["params"]
  params = flax.core.unfreeze(params) if hasattr(flax.core, "unfreeze") else params
  # break the symmetric initialisations (zero biases, unit scales) so every gradient path carries signal
  rng = np.random.RandomState(123)
  flat, tdef = u.tree_flatten_with_names(params)
  flat = [(n, (np.asarray(v) + 0.05 * rng.randn(*v.shape).astype(np.float32))
           if n.endswith(("bias", "scale", "cls")) else np.asarray(v)) for n, v in flat]
  params = tdef.unflatten([jnp.asarray(v) for _, v in flat])     # utils.tree_flatten_with_names returns a PyTreeDef

  def loss_fn(params):
    zimg, ztxt, extras = model.apply({"params": params}, jnp.asarray(image), jnp.asarray(text), train=True)
    # trainers/proj/image_text/siglip.py:291-306
    logits = jnp.dot(zimg, ztxt.T)
    logits = logits * extras["t"] + extras["b"]
    eye = jnp.eye(zimg.shape[0])
    m1_diag1 = -jnp.ones_like(logits) + 2 * eye
    loglik = jax.nn.log_sigmoid(m1_diag1 * logits)
    nll = -jnp.sum(loglik, axis=-1)
    return jnp.mean(nll), (zimg, ztxt, logits)

  (loss, (zimg, ztxt, logits)), grads = jax.value_and_grad(loss_fn, has_aux=True)(params)

  config = ml_collections.ConfigDict()
  config.lr, config.wd = 1e-3, 1e-2
  config.schedule = dict(decay_type="cosine", warmup_steps=2)
  config.optax_name = "scale_by_adam"
  config.grad_clip_norm = 1.0
  tx, _ = bv_optax.make(config, params, sched_kw=dict(total_steps=10, batch_size=cfg["n"], data_size=None))
  opt = tx.init(params)
  updates, opt = tx.update(grads, opt, params)
  new_params = optax.apply_updates(params, updates)

  out = {"image": image, "text": text, "loss": np.asarray(loss, np.float64), "zimg": np.asarray(zimg),
         "ztxt": np.asarray(ztxt), "logits": np.asarray(logits),
         "versions": np.asarray(f"jax {jax.__version__} flax {flax.__version__} optax {optax.__version__}"),
         "cfg_json": np.asarray(__import__("json").dumps({k: v for k, v in cfg.items()}))}
  gflat = dict(u.tree_flatten_with_names(grads)[0])
  nflat = dict(u.tree_flatten_with_names(new_params)[0])
  keep = list(gflat) if cfg["grads"] == "all" else \
      [n for n in gflat if n in ("t", "b") or "encoderblock_0/" in n or "encoderblock_11/MlpBlock_0/Dense_1" in n
       or n.endswith(("head/kernel", "probe", "pos_embedding", "embedding/bias"))]
  for n, v in u.tree_flatten_with_names(params)[0]:
    out["param:" + n] = np.asarray(v, np.float32)
  for n in keep:
    out["grad:" + n] = np.asarray(gflat[n], np.float32)
    out["new:" + n] = np.asarray(nflat[n], np.float32)
  dst = os.path.join(ROOT, "tests", "golden", f"siglip_jax_{tag}.npz")
  np.savez_compressed(dst, **out)
  print(f"wrote {dst}: loss {float(loss):.6f}, {len(keep)} gradients, {out['versions']}")


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--reference", required=True, help="root of a google-research/big_vision checkout")
  ap.add_argument("--b16", action="store_true", help="also the real ViT-B/16 + text-B shapes (n = 2; ~3 GB of RAM)")
  args = ap.parse_args()
  for tag in ("tiny", "tiny_tok") + (("b16",) if args.b16 else ()):
    run(tag, CONFIGS[tag], args.reference)


if __name__ == "__main__":
  main()
