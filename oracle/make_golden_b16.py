"""tests/golden/siglip_hf_b16.npz: HuggingFace SiglipModel at the REAL ViT-B/16 + text-B shapes
(BASELINE configs[2]: 224 px, 196 patches, 12 + 12 layers, width 768, 64 tokens, vocab 32 000),
forward only, two pairs.  Extends the tiny-shape pin of oracle/make_golden.py ("extend
the HF pin to the real B/16 shapes").  Only the OUTPUTS are stored (embeddings, logits, loss: a few
KB); weights and inputs are regenerated from the seeds below by the test, which runs
oracle/bv_oracle.py on them in fp32 and compares.  TEST INFRASTRUCTURE; needs `transformers`:

    python oracle/make_golden_b16.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bv_oracle as O  # noqa: E402
from make_golden import flax_to_hf_state  # noqa: E402

CFG = dict(width=768, depth=12, mlp_dim=3072, num_heads=12, patch=16, res=224, vocab=32_000, seq=64, n=2,
           temperature_init=10.0, bias_init=-10.0, seed=0, perturb_seed=123, batch_seed=1)
IMAGE_CFG = dict(variant="B/16", pool_type="map")
TEXT_CFG = dict(variant="B", vocab_size=32_000)
# second fixture: the So400m/14 shapes (width 1152, 16 heads -> head dim 72, mlp 4304, 14 x 14 patches, 256
# tokens), cut to 2 blocks per tower: pins the oracle where the general attention kernels and the padded
# stem are checked against it (tests/test_siglip_step_gpu.py::test_so400m_shapes_step)
CFG_SO = dict(width=1152, depth=2, mlp_dim=4304, num_heads=16, patch=14, res=224, vocab=32_000, seq=16, n=2,
              temperature_init=10.0, bias_init=-10.0, seed=3, perturb_seed=321, batch_seed=2)
IMAGE_CFG_SO = dict(variant="So400m/14", pool_type="map", depth=2)
TEXT_CFG_SO = dict(variant="So400m", depth=2, vocab_size=32_000)
# third fixture: the L/16@336 shapes of BASELINE configs[3] (width 1024, 16 heads, mlp 4096, 441 tokens) at 2 blocks
CFG_L = dict(width=1024, depth=2, mlp_dim=4096, num_heads=16, patch=16, res=336, vocab=32_000, seq=64, n=2,
             temperature_init=10.0, bias_init=-10.0, seed=5, perturb_seed=55, batch_seed=3)
IMAGE_CFG_L = dict(variant="L/16", pool_type="map", depth=2)
TEXT_CFG_L = dict(variant="L", depth=2, vocab_size=32_000)
FIXTURES = {"b16": (CFG, IMAGE_CFG, TEXT_CFG), "so400m_d2": (CFG_SO, IMAGE_CFG_SO, TEXT_CFG_SO),
            "l16_336_d2": (CFG_L, IMAGE_CFG_L, TEXT_CFG_L)}


def make_inputs(dtype=torch.float32, tag="b16"):
  """Seeded weights (Flax layout, biases / scales perturbed) and batch - shared with the test."""
  c, IMAGE_CFG, TEXT_CFG = FIXTURES[tag]
  params = O.init_two_towers(c["seed"], (c["res"], c["res"]), c["seq"], image_cfg=IMAGE_CFG, text_cfg=TEXT_CFG,
                             out_dim=(None, c["width"]), temperature_init=c["temperature_init"],
                             bias_init=c["bias_init"], dtype=dtype)
  gen = torch.Generator().manual_seed(c["perturb_seed"])
  flat = [(n, v + 0.05 * torch.randn(v.shape, generator=gen, dtype=dtype) if n.endswith(("bias", "scale")) else v)
          for n, v in O.tree_flatten_with_names(params)]
  image, text = O.synthetic_batch(c["batch_seed"], c["n"], c["res"], c["seq"], c["vocab"], dtype=dtype)
  return O.recover_tree(flat), image, text


def grad_selection(g, c):
  """HF parameter gradients (dict name -> tensor) mapped back to the Flax leaf names / layouts: one leaf of every
  layer kind per tower, from the first, a middle and the last block (the same mapping as oracle/make_golden.py)."""
  D, H, depth = c["width"], c["num_heads"], c["depth"]
  Li, Lm, Ll = 0, depth // 2, depth - 1
  V = lambda i: f"vision_model.encoder.layers.{i}"
  T = lambda i: f"text_model.encoder.layers.{i}"
  IB = lambda i: f"img/Transformer/encoderblock_{i}"
  TB = lambda i: f"txt/Encoder_0/encoderblock_{i}"
  A = "MultiHeadDotProductAttention_0"
  return {
      "img/embedding/kernel": g["vision_model.embeddings.patch_embedding.weight"].permute(2, 3, 1, 0),
      "img/pos_embedding": g["vision_model.embeddings.position_embedding.weight"][None],
      f"{IB(Li)}/LayerNorm_0/scale": g[V(Li) + ".layer_norm1.weight"],
      f"{IB(Li)}/{A}/query/kernel": g[V(Li) + ".self_attn.q_proj.weight"].T.reshape(D, H, D // H),
      f"{IB(Li)}/{A}/key/bias": g[V(Li) + ".self_attn.k_proj.bias"].reshape(H, D // H),
      f"{IB(Lm)}/{A}/key/kernel": g[V(Lm) + ".self_attn.k_proj.weight"].T.reshape(D, H, D // H),
      f"{IB(Lm)}/{A}/out/kernel": g[V(Lm) + ".self_attn.out_proj.weight"].T.reshape(H, D // H, D),
      f"{IB(Lm)}/MlpBlock_0/Dense_0/kernel": g[V(Lm) + ".mlp.fc1.weight"].T,
      f"{IB(Ll)}/MlpBlock_0/Dense_1/kernel": g[V(Ll) + ".mlp.fc2.weight"].T,
      f"{IB(Ll)}/MlpBlock_0/Dense_1/bias": g[V(Ll) + ".mlp.fc2.bias"],
      f"{IB(Ll)}/LayerNorm_1/bias": g[V(Ll) + ".layer_norm2.bias"],
      "img/Transformer/encoder_norm/scale": g["vision_model.post_layernorm.weight"],
      "img/MAPHead_0/probe": g["vision_model.head.probe"],
      "img/MAPHead_0/MultiHeadDotProductAttention_0/value/kernel":
          g["vision_model.head.attention.in_proj_weight"][2 * D:].T.reshape(D, H, D // H),
      "img/MAPHead_0/MlpBlock_0/Dense_0/kernel": g["vision_model.head.mlp.fc1.weight"].T,
      "txt/Embed_0/embedding": g["text_model.embeddings.token_embedding.weight"],
      "txt/pos_embedding": g["text_model.embeddings.position_embedding.weight"][None],
      f"{TB(Li)}/{A}/value/kernel": g[T(Li) + ".self_attn.v_proj.weight"].T.reshape(D, H, D // H),
      f"{TB(Lm)}/{A}/query/kernel": g[T(Lm) + ".self_attn.q_proj.weight"].T.reshape(D, H, D // H),
      f"{TB(Lm)}/MlpBlock_0/Dense_0/bias": g[T(Lm) + ".mlp.fc1.bias"],
      f"{TB(Ll)}/MlpBlock_0/Dense_1/kernel": g[T(Ll) + ".mlp.fc2.weight"].T,
      f"{TB(Ll)}/LayerNorm_0/scale": g[T(Ll) + ".layer_norm1.weight"],
      "txt/Encoder_0/encoder_norm/bias": g["text_model.final_layer_norm.bias"],
      "txt/head/kernel": g["text_model.head.weight"].T,
      "t": g["logit_scale"],
      "b": g["logit_bias"],
  }


NPROBE = 48


def fingerprint(name, v):
  """A gradient tensor as a few numbers (the tensors themselves are up to 19 MB in fp64): its sum, its L2 norm and
  NPROBE entries at flat indices drawn from a generator seeded by the leaf name."""
  v = v.detach().double().contiguous().view(-1)
  gen = torch.Generator().manual_seed(sum(name.encode()) * 7919 + v.numel())
  idx = torch.randint(0, v.numel(), (NPROBE,), generator=gen)
  return torch.cat([v.sum()[None], v.norm()[None], v[idx]]).numpy()


def main(tag="b16"):
  from transformers import SiglipConfig, SiglipModel
  c, IMAGE_CFG, TEXT_CFG = FIXTURES[tag]
  params, image, text = make_inputs(tag=tag)
  hf_cfg = SiglipConfig(
      text_config=dict(hidden_size=c["width"], intermediate_size=c["mlp_dim"], num_hidden_layers=c["depth"],
                       num_attention_heads=c["num_heads"], vocab_size=c["vocab"], max_position_embeddings=c["seq"],
                       projection_size=c["width"], layer_norm_eps=1e-6, hidden_act="gelu_pytorch_tanh",
                       bos_token_id=None, eos_token_id=None, pad_token_id=1),
      vision_config=dict(hidden_size=c["width"], intermediate_size=c["mlp_dim"], num_hidden_layers=c["depth"],
                         num_attention_heads=c["num_heads"], image_size=c["res"], patch_size=c["patch"],
                         layer_norm_eps=1e-6, hidden_act="gelu_pytorch_tanh"))
  hf = SiglipModel(hf_cfg).to(torch.float64).eval()
  p64 = O.tree_map(lambda v: v.double(), params)
  missing, unexpected = hf.load_state_dict(flax_to_hf_state(p64, c), strict=False)
  missing = [m for m in missing if "position_ids" not in m]
  assert not missing and not unexpected, (missing, unexpected)
  with torch.no_grad():
    res = hf(input_ids=text.long(), pixel_values=image.double().permute(0, 3, 1, 2).contiguous(), return_loss=True)
    loss, (zimg, ztxt, logits, _) = O.siglip_step_loss(p64, image.double(), text, image_cfg=IMAGE_CFG,
                                                       text_cfg=TEXT_CFG, out_dim=(None, c["width"]))
  for name, a, b in (("zimg", zimg, res.image_embeds), ("ztxt", ztxt, res.text_embeds),
                     ("logits", logits, res.logits_per_image), ("loss", loss, res.loss)):
    err = (a - b).abs().max().item()
    print(f"oracle(fp64) vs HF(fp64) {name}: max abs err {err:.3e}")
    assert err < 1e-6, name
  # the BACKWARD at these shapes: HF's autograd gradients of its own loss, fp64, as fingerprints
  hf.zero_grad()
  res_g = hf(input_ids=text.long(), pixel_values=image.double().permute(0, 3, 1, 2).contiguous(), return_loss=True)
  res_g.loss.backward()
  sel = grad_selection({n: p_.grad for n, p_ in hf.named_parameters()}, c)
  fps = {"hfgradfp:" + n: fingerprint(n, v) for n, v in sel.items()}
  print(f"{len(fps)} gradient fingerprints, largest |grad| norm {max(float(f[1]) for f in fps.values()):.3e}")
  dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", f"siglip_hf_{tag}.npz")
  np.savez_compressed(dst, **fps, hf_zimg=res.image_embeds.numpy(), hf_ztxt=res.text_embeds.numpy(),
                      hf_logits=res.logits_per_image.numpy(), hf_loss=res.loss.numpy(),
                      param_checksum=np.asarray(sum(float(v.double().sum()) for _, v in O.tree_flatten_with_names(params))),
                      **{"cfg_" + k: np.asarray(v) for k, v in c.items()})
  print("wrote", os.path.normpath(dst), os.path.getsize(dst), "bytes")


if __name__ == "__main__":
  for t in (sys.argv[1:] or list(FIXTURES)):
    main(t)
