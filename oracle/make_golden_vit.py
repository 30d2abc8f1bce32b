"""Generate tests/golden/vit_hf_tiny.npz - pins the oracle's SINGLE-TOWER CLASSIFICATION path.

TEST INFRASTRUCTURE.  Run in the build container (needs `transformers`):

    python oracle/make_golden_vit.py

Independent second opinion for models/vit.py (`pool_type="tok"`, learned posemb, classification
head) + `utils.softmax_xent` (train.py:295-300): HuggingFace `ViTForImageClassification`, the
PyTorch port of the same ViT.  Random Flax-layout parameters are copied into the HF module
(NHWC/HWIO -> NCHW/OIHW, (in,out) -> (out,in), per-head projections flattened; big_vision adds
the position embedding BEFORE concatenating the cls token, so HF's cls position row is zero),
HF runs in fp64 with integer labels (= one-hot softmax cross-entropy, mean over the batch), and
its logits, loss and autograd gradients are stored.  tests/test_oracle.py checks
oracle/bv_oracle.py against them without importing transformers.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bv_oracle as O  # noqa: E402

CFG = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch=16, res=48, classes=10, n=5)


def flax_to_hf(p, c):
  D = c["width"]
  sd = {}
  sd["vit.embeddings.cls_token"] = p["cls"]
  L = p["pos_embedding"].shape[1]
  sd["vit.embeddings.position_embeddings"] = torch.cat([torch.zeros(1, 1, D, dtype=p["cls"].dtype), p["pos_embedding"]], 1)
  assert sd["vit.embeddings.position_embeddings"].shape[1] == L + 1
  sd["vit.embeddings.patch_embeddings.projection.weight"] = p["embedding"]["kernel"].permute(3, 2, 0, 1).contiguous()
  sd["vit.embeddings.patch_embeddings.projection.bias"] = p["embedding"]["bias"]
  for i in range(c["depth"]):
    b = p["Transformer"][f"encoderblock_{i}"]
    a = b["MultiHeadDotProductAttention_0"]
    P = f"vit.layers.{i}"
    for hf, fx in (("q_proj", "query"), ("k_proj", "key"), ("v_proj", "value")):
      sd[f"{P}.attention.{hf}.weight"] = a[fx]["kernel"].reshape(D, D).T.contiguous()
      sd[f"{P}.attention.{hf}.bias"] = a[fx]["bias"].reshape(D)
    sd[f"{P}.attention.o_proj.weight"] = a["out"]["kernel"].reshape(D, D).T.contiguous()
    sd[f"{P}.attention.o_proj.bias"] = a["out"]["bias"]
    sd[f"{P}.layernorm_before.weight"] = b["LayerNorm_0"]["scale"]
    sd[f"{P}.layernorm_before.bias"] = b["LayerNorm_0"]["bias"]
    sd[f"{P}.layernorm_after.weight"] = b["LayerNorm_1"]["scale"]
    sd[f"{P}.layernorm_after.bias"] = b["LayerNorm_1"]["bias"]
    sd[f"{P}.mlp.fc1.weight"] = b["MlpBlock_0"]["Dense_0"]["kernel"].T.contiguous()
    sd[f"{P}.mlp.fc1.bias"] = b["MlpBlock_0"]["Dense_0"]["bias"]
    sd[f"{P}.mlp.fc2.weight"] = b["MlpBlock_0"]["Dense_1"]["kernel"].T.contiguous()
    sd[f"{P}.mlp.fc2.bias"] = b["MlpBlock_0"]["Dense_1"]["bias"]
  sd["vit.layernorm.weight"] = p["Transformer"]["encoder_norm"]["scale"]
  sd["vit.layernorm.bias"] = p["Transformer"]["encoder_norm"]["bias"]
  sd["classifier.weight"] = p["head"]["kernel"].T.contiguous()
  sd["classifier.bias"] = p["head"]["bias"]
  return sd


def main():
  from transformers import ViTConfig, ViTForImageClassification
  c = CFG
  dt = torch.float64
  mcfg = dict(width=c["width"], depth=c["depth"], mlp_dim=c["mlp_dim"], num_heads=c["num_heads"],
              patch_size=(c["patch"], c["patch"]), pool_type="tok", head_zeroinit=False)
  gen = torch.Generator().manual_seed(11)
  params = O.init_vit(gen, (c["res"], c["res"]), num_classes=c["classes"], dtype=dt, **mcfg)
  flat = [(n, (v + 0.05 * torch.randn(v.shape, generator=gen, dtype=dt)) if n.endswith(("bias", "scale", "cls")) else v)
          for n, v in O.tree_flatten_with_names(params)]
  # store fp32-exact values so the fixture's inputs reproduce its outputs
  flat = [(n, v.float().double()) for n, v in flat]
  params = O.recover_tree(flat)
  image = (torch.rand((c["n"], c["res"], c["res"], 3), generator=gen) * 2 - 1).double()
  labels_int = torch.randint(0, c["classes"], (c["n"],), generator=gen)

  hf = ViTForImageClassification(ViTConfig(
      hidden_size=c["width"], num_hidden_layers=c["depth"], num_attention_heads=c["num_heads"],
      intermediate_size=c["mlp_dim"], hidden_act="gelu_pytorch_tanh", image_size=c["res"], patch_size=c["patch"],
      num_labels=c["classes"], layer_norm_eps=1e-6, qkv_bias=True, hidden_dropout_prob=0.0,
      attention_probs_dropout_prob=0.0)).to(dt).eval()
  missing, unexpected = hf.load_state_dict(flax_to_hf(params, c), strict=False)
  assert not missing and not unexpected, (missing, unexpected)
  res = hf(pixel_values=image.permute(0, 3, 1, 2).contiguous(), labels=labels_int)
  res.loss.backward()
  g = {n: p_.grad for n, p_ in hf.named_parameters()}

  onehot = torch.nn.functional.one_hot(labels_int, c["classes"]).double()
  p_req = O.tree_map(lambda v: v.clone().requires_grad_(True), params)
  loss, logits = O.classification_step_loss(p_req, image, onehot, model_cfg=mcfg, num_classes=c["classes"],
                                            loss="softmax_xent")
  loss.backward()
  print(f"oracle vs HF logits: {(logits - res.logits).abs().max().item():.3e}  loss: {abs(loss.item() - res.loss.item()):.3e}")
  assert (logits - res.logits).abs().max() < 1e-8 and abs(loss.item() - res.loss.item()) < 1e-9

  D, H = c["width"], c["num_heads"]
  sel = {
      "embedding/kernel": g["vit.embeddings.patch_embeddings.projection.weight"].permute(2, 3, 1, 0),
      "pos_embedding": g["vit.embeddings.position_embeddings"][:, 1:],
      "cls": g["vit.embeddings.cls_token"],
      "Transformer/encoderblock_0/MultiHeadDotProductAttention_0/key/kernel":
          g["vit.layers.0.attention.k_proj.weight"].T.reshape(D, H, D // H),
      "Transformer/encoderblock_1/MultiHeadDotProductAttention_0/out/bias": g["vit.layers.1.attention.o_proj.bias"],
      "Transformer/encoderblock_1/LayerNorm_1/scale": g["vit.layers.1.layernorm_after.weight"],
      "Transformer/encoderblock_0/MlpBlock_0/Dense_1/kernel": g["vit.layers.0.mlp.fc2.weight"].T,
      "Transformer/encoder_norm/scale": g["vit.layernorm.weight"],
      "head/kernel": g["classifier.weight"].T,
      "head/bias": g["classifier.bias"],
  }
  out = {"cfg_" + k: np.asarray(v) for k, v in c.items()}
  for n, v in flat:
    out["param:" + n] = v.numpy().astype(np.float32)
  out["image"] = image.numpy().astype(np.float32)
  out["labels"] = labels_int.numpy().astype(np.int32)
  out["hf_logits"] = res.logits.detach().numpy()
  out["hf_loss"] = res.loss.detach().numpy()
  for n, v in sel.items():
    out["hfgrad:" + n] = v.detach().contiguous().numpy()
  dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "vit_hf_tiny.npz")
  np.savez_compressed(dst, **out)
  print("wrote", os.path.normpath(dst), os.path.getsize(dst) // 1024, "KiB")


if __name__ == "__main__":
  main()
