"""Generate tests/golden/siglip_hf_tiny.npz — the fixture that pins the oracle.

TEST INFRASTRUCTURE.  Run in the build container (needs `transformers`):

    python oracle/make_golden.py

The JAX reference cannot run here (no jax/flax); the independent second
opinion is HuggingFace `SiglipModel`, a PyTorch port of the same big_vision
model (models/vit.py + text_transformer.py + two_towers.py + the sigmoid loss
of trainers/proj/image_text/siglip.py:287-306).  We draw random Flax-layout
parameters, copy them into the HF module, run HF in fp64 and store HF's
outputs.  `tests/test_oracle.py` then checks `oracle/bv_oracle.py` against the
stored HF outputs (no transformers import at test time).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bv_oracle as O  # noqa: E402

CFG = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch=16, res=32,
           vocab=100, seq=16, n=4, temperature_init=10.0, bias_init=-10.0)


def flax_to_hf_state(p, cfg):
  D, H = cfg["width"], cfg["num_heads"]
  sd = {}

  def put_ln(dst, src):
    sd[dst + ".weight"] = src["scale"]
    sd[dst + ".bias"] = src["bias"]

  def put_dense(dst, src):
    sd[dst + ".weight"] = src["kernel"].T.contiguous()
    sd[dst + ".bias"] = src["bias"]

  def put_encoder(prefix, enc):
    for i in range(cfg["depth"]):
      b = enc[f"encoderblock_{i}"]
      L = f"{prefix}.encoder.layers.{i}"
      put_ln(L + ".layer_norm1", b["LayerNorm_0"])
      put_ln(L + ".layer_norm2", b["LayerNorm_1"])
      a = b["MultiHeadDotProductAttention_0"]
      for hf, fx in (("q_proj", "query"), ("k_proj", "key"), ("v_proj", "value")):
        sd[f"{L}.self_attn.{hf}.weight"] = a[fx]["kernel"].reshape(D, D).T.contiguous()
        sd[f"{L}.self_attn.{hf}.bias"] = a[fx]["bias"].reshape(D)
      sd[f"{L}.self_attn.out_proj.weight"] = a["out"]["kernel"].reshape(D, D).T.contiguous()
      sd[f"{L}.self_attn.out_proj.bias"] = a["out"]["bias"]
      put_dense(L + ".mlp.fc1", b["MlpBlock_0"]["Dense_0"])
      put_dense(L + ".mlp.fc2", b["MlpBlock_0"]["Dense_1"])

  img, txt = p["img"], p["txt"]
  sd["vision_model.embeddings.patch_embedding.weight"] = img["embedding"]["kernel"].permute(3, 2, 0, 1).contiguous()
  sd["vision_model.embeddings.patch_embedding.bias"] = img["embedding"]["bias"]
  sd["vision_model.embeddings.position_embedding.weight"] = img["pos_embedding"][0]
  put_encoder("vision_model", img["Transformer"])
  put_ln("vision_model.post_layernorm", img["Transformer"]["encoder_norm"])
  m = img["MAPHead_0"]
  sd["vision_model.head.probe"] = m["probe"]
  a = m["MultiHeadDotProductAttention_0"]
  sd["vision_model.head.attention.in_proj_weight"] = torch.cat(
      [a[k]["kernel"].reshape(D, D).T for k in ("query", "key", "value")], 0).contiguous()
  sd["vision_model.head.attention.in_proj_bias"] = torch.cat(
      [a[k]["bias"].reshape(D) for k in ("query", "key", "value")], 0)
  sd["vision_model.head.attention.out_proj.weight"] = a["out"]["kernel"].reshape(D, D).T.contiguous()
  sd["vision_model.head.attention.out_proj.bias"] = a["out"]["bias"]
  put_ln("vision_model.head.layernorm", m["LayerNorm_0"])
  put_dense("vision_model.head.mlp.fc1", m["MlpBlock_0"]["Dense_0"])
  put_dense("vision_model.head.mlp.fc2", m["MlpBlock_0"]["Dense_1"])

  sd["text_model.embeddings.token_embedding.weight"] = txt["Embed_0"]["embedding"]
  sd["text_model.embeddings.position_embedding.weight"] = txt["pos_embedding"][0]
  put_encoder("text_model", txt["Encoder_0"])
  put_ln("text_model.final_layer_norm", txt["Encoder_0"]["encoder_norm"])
  put_dense("text_model.head", txt["head"])
  sd["logit_scale"] = p["t"]
  sd["logit_bias"] = p["b"]
  return sd


def main():
  from transformers import SiglipConfig, SiglipModel
  c = CFG
  image_cfg = dict(width=c["width"], depth=c["depth"], mlp_dim=c["mlp_dim"],
                   num_heads=c["num_heads"], patch_size=(c["patch"], c["patch"]),
                   pool_type="map")
  text_cfg = dict(width=c["width"], depth=c["depth"], mlp_dim=c["mlp_dim"],
                  num_heads=c["num_heads"], vocab_size=c["vocab"])
  dt = torch.float64
  params = O.init_two_towers(0, (c["res"], c["res"]), c["seq"], image_cfg=image_cfg,
                             text_cfg=text_cfg, out_dim=(None, c["width"]),
                             temperature_init=c["temperature_init"],
                             bias_init=c["bias_init"], dtype=dt)
  # Perturb the all-zero / all-one initialisations so every term is exercised.
  gen = torch.Generator().manual_seed(123)
  flat = O.tree_flatten_with_names(params)
  flat = [(n, v + 0.05 * torch.randn(v.shape, generator=gen, dtype=dt)
           if n.endswith(("bias", "scale")) else v) for n, v in flat]
  params = O.recover_tree(flat)
  image, text = O.synthetic_batch(1, c["n"], c["res"], c["seq"], c["vocab"], dtype=dt)

  hf_cfg = SiglipConfig(
      text_config=dict(hidden_size=c["width"], intermediate_size=c["mlp_dim"],
                       num_hidden_layers=c["depth"], num_attention_heads=c["num_heads"],
                       vocab_size=c["vocab"], max_position_embeddings=c["seq"],
                       projection_size=c["width"], layer_norm_eps=1e-6,
                       hidden_act="gelu_pytorch_tanh", bos_token_id=None, eos_token_id=None,
                       pad_token_id=1),
      vision_config=dict(hidden_size=c["width"], intermediate_size=c["mlp_dim"],
                         num_hidden_layers=c["depth"], num_attention_heads=c["num_heads"],
                         image_size=c["res"], patch_size=c["patch"], layer_norm_eps=1e-6,
                         hidden_act="gelu_pytorch_tanh"))
  hf = SiglipModel(hf_cfg).to(dt).eval()
  sd = flax_to_hf_state(params, c)
  missing, unexpected = hf.load_state_dict(sd, strict=False)
  missing = [m for m in missing if "position_ids" not in m]
  assert not missing and not unexpected, (missing, unexpected)
  with torch.no_grad():
    res = hf(input_ids=text.long(), pixel_values=image.permute(0, 3, 1, 2).contiguous(),
             return_loss=True)
  hf_out = dict(zimg=res.image_embeds, ztxt=res.text_embeds,
                logits=res.logits_per_image, loss=res.loss)

  with torch.no_grad():
    loss, (zimg, ztxt, logits, _) = O.siglip_step_loss(
        params, image, text, image_cfg=image_cfg, text_cfg=text_cfg,
        out_dim=(None, c["width"]))
  for name, a, b in (("zimg", zimg, hf_out["zimg"]), ("ztxt", ztxt, hf_out["ztxt"]),
                     ("logits", logits, hf_out["logits"]), ("loss", loss, hf_out["loss"])):
    err = (a - b).abs().max().item()
    print(f"oracle vs HF {name}: max abs err {err:.3e}")
    assert err < 1e-6, name

  out = {"cfg_" + k: np.asarray(v) for k, v in c.items()}
  for n, v in O.tree_flatten_with_names(params):
    out["param:" + n] = v.numpy().astype(np.float32)
  out["image"] = image.numpy().astype(np.float32)
  out["text"] = text.numpy().astype(np.int32)
  # HF outputs recomputed from the fp32-rounded params/inputs so the stored
  # inputs reproduce the stored outputs exactly.
  params32 = O.recover_tree([(n, torch.from_numpy(out["param:" + n]).to(dt)) for n, _ in flat])
  hf.load_state_dict(flax_to_hf_state(params32, c), strict=False)
  with torch.no_grad():
    res = hf(input_ids=text.long(),
             pixel_values=torch.from_numpy(out["image"]).to(dt).permute(0, 3, 1, 2).contiguous(),
             return_loss=True)
  out["hf_zimg"] = res.image_embeds.numpy()
  out["hf_ztxt"] = res.text_embeds.numpy()
  out["hf_logits"] = res.logits_per_image.numpy()
  out["hf_loss"] = res.loss.numpy()
  # HF's autograd gradients of its loss w.r.t. a selection of parameters, mapped back to the Flax
  # names / layouts: pins the BACKWARD of the oracle (and through it the hand-written HIP
  # backward) to an independent implementation.
  hf.zero_grad()
  res = hf(input_ids=text.long(),
           pixel_values=torch.from_numpy(out["image"]).to(dt).permute(0, 3, 1, 2).contiguous(),
           return_loss=True)
  res.loss.backward()
  g = {n: p_.grad for n, p_ in hf.named_parameters()}
  D, H = c["width"], c["num_heads"]
  L0 = "vision_model.encoder.layers.0"
  T1 = "text_model.encoder.layers.1"
  sel = {
      "img/embedding/kernel": g["vision_model.embeddings.patch_embedding.weight"].permute(2, 3, 1, 0),
      "img/embedding/bias": g["vision_model.embeddings.patch_embedding.bias"],
      "img/pos_embedding": g["vision_model.embeddings.position_embedding.weight"][None],
      "img/Transformer/encoderblock_0/LayerNorm_0/scale": g[L0 + ".layer_norm1.weight"],
      "img/Transformer/encoderblock_0/MultiHeadDotProductAttention_0/query/kernel":
          g[L0 + ".self_attn.q_proj.weight"].T.reshape(D, H, D // H),
      "img/Transformer/encoderblock_0/MultiHeadDotProductAttention_0/key/bias":
          g[L0 + ".self_attn.k_proj.bias"].reshape(H, D // H),
      "img/Transformer/encoderblock_0/MultiHeadDotProductAttention_0/out/kernel":
          g[L0 + ".self_attn.out_proj.weight"].T.reshape(H, D // H, D),
      "img/Transformer/encoderblock_0/MlpBlock_0/Dense_0/kernel": g[L0 + ".mlp.fc1.weight"].T,
      "img/Transformer/encoderblock_0/MlpBlock_0/Dense_1/bias": g[L0 + ".mlp.fc2.bias"],
      "img/Transformer/encoder_norm/bias": g["vision_model.post_layernorm.bias"],
      "img/MAPHead_0/probe": g["vision_model.head.probe"],
      "img/MAPHead_0/MultiHeadDotProductAttention_0/value/kernel":
          g["vision_model.head.attention.in_proj_weight"][2 * D:].T.reshape(D, H, D // H),
      "img/MAPHead_0/MlpBlock_0/Dense_1/kernel": g["vision_model.head.mlp.fc2.weight"].T,
      "txt/Embed_0/embedding": g["text_model.embeddings.token_embedding.weight"],
      "txt/pos_embedding": g["text_model.embeddings.position_embedding.weight"][None],
      "txt/Encoder_0/encoderblock_1/MultiHeadDotProductAttention_0/value/kernel":
          g[T1 + ".self_attn.v_proj.weight"].T.reshape(D, H, D // H),
      "txt/Encoder_0/encoderblock_1/MlpBlock_0/Dense_0/bias": g[T1 + ".mlp.fc1.bias"],
      "txt/Encoder_0/encoder_norm/scale": g["text_model.final_layer_norm.weight"],
      "txt/head/kernel": g["text_model.head.weight"].T,
      "txt/head/bias": g["text_model.head.bias"],
      "t": g["logit_scale"],
      "b": g["logit_bias"],
  }
  for n, v in sel.items():
    out["hfgrad:" + n] = v.detach().contiguous().numpy()
  dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden",
                     "siglip_hf_tiny.npz")
  np.savez_compressed(dst, **out)
  print("wrote", os.path.normpath(dst), os.path.getsize(dst) // 1024, "KiB")


if __name__ == "__main__":
  main()
