"""Generate tests/golden/naflex_hf_tiny.npz - pins the oracle's NaFlex tower (bv_oracle.naflex_vit_forward).

TEST INFRASTRUCTURE.  Run in the build container (needs `transformers`):

    python oracle/make_golden_naflex.py

models/proj/image_text/naflex_vit.py:38-293 (flattened patches + per-example resized 2-D position embedding
+ padding masks + masked MAP pooling) has no test or golden vector in the reference and needs jax to run.
The independent second opinion is HuggingFace `Siglip2VisionModel`, the PyTorch port of the same NaFlex
tower: random parameters in the Flax layout are copied into the HF module, HF runs in fp64 on RAGGED inputs
(three examples with 3x4, 2x2 and 4x4 patch grids inside 16 patch slots; `pixel_attention_mask`,
`spatial_shapes`) and the encoded tokens of the valid patches, the MAP-pooled output, a scalar loss and its
autograd gradients w.r.t. parameters of every kind are stored.  (HF interpolates the position embedding in
fp32 on CPU - its own upcast - so the fixture agrees with the fp64 oracle to ~1e-6, not 1e-12.)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bv_oracle as O  # noqa: E402

CFG = dict(width=128, depth=2, mlp_dim=256, num_heads=2, patch=4, nposemb=8, slots=16)
SEED = 21
GRIDS = [(3, 4), (2, 2), (4, 4)]
GRADS = ["embedding/kernel", "embedding/bias", "pos_embedding", "Transformer/encoderblock_0/LayerNorm_0/scale",
         "Transformer/encoderblock_0/MultiHeadDotProductAttention_0/key/kernel",
         "Transformer/encoderblock_1/MultiHeadDotProductAttention_0/out/kernel",
         "Transformer/encoderblock_1/MlpBlock_0/Dense_0/kernel", "Transformer/encoder_norm/bias",
         "MAPHead_0/probe", "MAPHead_0/MultiHeadDotProductAttention_0/query/kernel", "MAPHead_0/MlpBlock_0/Dense_1/kernel"]


def make_params(dtype=torch.float64):
  c = CFG
  gen = torch.Generator().manual_seed(SEED)
  D, pd, P = c["width"], c["patch"] * c["patch"] * 3, c["nposemb"]
  full = O.init_vit(gen, (c["patch"] * 4, c["patch"] * 4), patch_size=(c["patch"], c["patch"]), width=D, depth=c["depth"],
                    mlp_dim=c["mlp_dim"], num_heads=c["num_heads"], pool_type="map", dtype=dtype)
  p = {"embedding": {"kernel": torch.randn((pd, D), generator=gen, dtype=dtype) / pd ** 0.5,
                     "bias": torch.zeros(D, dtype=dtype)},
       "pos_embedding": torch.randn((P, P, D), generator=gen, dtype=dtype) / D ** 0.5,
       "Transformer": full["Transformer"], "MAPHead_0": full["MAPHead_0"]}
  flat = [(n, v + 0.05 * torch.randn(v.shape, generator=gen, dtype=dtype) if n.endswith(("bias", "scale")) else v)
          for n, v in O.tree_flatten_with_names(p)]
  return O.recover_tree(flat)


def make_inputs(dtype=torch.float64):
  c = CFG
  gen = torch.Generator().manual_seed(SEED + 1)
  n, N, pd = len(GRIDS), c["slots"], c["patch"] * c["patch"] * 3
  patches = torch.randn((n, N, pd), generator=gen, dtype=dtype)
  ptype = torch.zeros((n, N), dtype=torch.int32)
  yabs = torch.zeros((n, N), dtype=torch.int32)
  xabs = torch.zeros((n, N), dtype=torch.int32)
  for e, (h, w) in enumerate(GRIDS):
    k = h * w
    ptype[e, :k] = 1
    yabs[e, :k] = torch.arange(k, dtype=torch.int32) // w
    xabs[e, :k] = torch.arange(k, dtype=torch.int32) % w
    patches[e, k:] = 0
  return patches, ptype, yabs, xabs


def main():
  from transformers import Siglip2VisionConfig, Siglip2VisionModel
  c = CFG
  p = make_params()
  patches, ptype, yabs, xabs = make_inputs()
  D = c["width"]
  sd = {}

  def put_ln(dst, src):
    sd[dst + ".weight"], sd[dst + ".bias"] = src["scale"], src["bias"]

  def put_dense(dst, src):
    sd[dst + ".weight"], sd[dst + ".bias"] = src["kernel"].T.contiguous(), src["bias"]

  put_dense("embeddings.patch_embedding", p["embedding"])
  sd["embeddings.position_embedding.weight"] = p["pos_embedding"].reshape(c["nposemb"] ** 2, D)
  for i in range(c["depth"]):
    b = p["Transformer"][f"encoderblock_{i}"]
    L = f"encoder.layers.{i}"
    put_ln(L + ".layer_norm1", b["LayerNorm_0"]); put_ln(L + ".layer_norm2", b["LayerNorm_1"])
    a = b["MultiHeadDotProductAttention_0"]
    for hf, fx in (("q_proj", "query"), ("k_proj", "key"), ("v_proj", "value")):
      sd[f"{L}.self_attn.{hf}.weight"] = a[fx]["kernel"].reshape(D, D).T.contiguous()
      sd[f"{L}.self_attn.{hf}.bias"] = a[fx]["bias"].reshape(D)
    sd[f"{L}.self_attn.out_proj.weight"] = a["out"]["kernel"].reshape(D, D).T.contiguous()
    sd[f"{L}.self_attn.out_proj.bias"] = a["out"]["bias"]
    put_dense(L + ".mlp.fc1", b["MlpBlock_0"]["Dense_0"]); put_dense(L + ".mlp.fc2", b["MlpBlock_0"]["Dense_1"])
  put_ln("post_layernorm", p["Transformer"]["encoder_norm"])
  m = p["MAPHead_0"]
  a = m["MultiHeadDotProductAttention_0"]
  sd["head.probe"] = m["probe"]
  sd["head.attention.in_proj_weight"] = torch.cat([a[k]["kernel"].reshape(D, D).T for k in ("query", "key", "value")], 0).contiguous()
  sd["head.attention.in_proj_bias"] = torch.cat([a[k]["bias"].reshape(D) for k in ("query", "key", "value")], 0)
  sd["head.attention.out_proj.weight"] = a["out"]["kernel"].reshape(D, D).T.contiguous()
  sd["head.attention.out_proj.bias"] = a["out"]["bias"]
  put_ln("head.layernorm", m["LayerNorm_0"])
  put_dense("head.mlp.fc1", m["MlpBlock_0"]["Dense_0"]); put_dense("head.mlp.fc2", m["MlpBlock_0"]["Dense_1"])

  hc = Siglip2VisionConfig(hidden_size=D, intermediate_size=c["mlp_dim"], num_hidden_layers=c["depth"],
                           num_attention_heads=c["num_heads"], num_channels=3, num_patches=c["nposemb"] ** 2,
                           patch_size=c["patch"], hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6, attention_dropout=0.0)
  hc._attn_implementation = "eager"
  hf = Siglip2VisionModel(hc).double().train(False)
  missing, unexpected = hf.load_state_dict(sd, strict=False)
  assert not missing and not unexpected, (missing, unexpected)
  out = hf(pixel_values=patches, pixel_attention_mask=ptype.long(), spatial_shapes=torch.tensor(GRIDS))
  pooled, hidden = out.pooler_output, out.last_hidden_state
  gen = torch.Generator().manual_seed(SEED + 2)
  cot = torch.randn(pooled.shape, generator=gen, dtype=torch.float64)
  loss = (pooled * cot).sum()
  loss.backward()
  hp = dict(hf.named_parameters())
  H, Dh = c["num_heads"], D // c["num_heads"]
  inv = {"embedding/kernel": hp["embeddings.patch_embedding.weight"].grad.T,
         "embedding/bias": hp["embeddings.patch_embedding.bias"].grad,
         "pos_embedding": hp["embeddings.position_embedding.weight"].grad.reshape(c["nposemb"], c["nposemb"], D),
         "Transformer/encoderblock_0/LayerNorm_0/scale": hp["encoder.layers.0.layer_norm1.weight"].grad,
         "Transformer/encoderblock_0/MultiHeadDotProductAttention_0/key/kernel":
             hp["encoder.layers.0.self_attn.k_proj.weight"].grad.T.reshape(D, H, Dh),
         "Transformer/encoderblock_1/MultiHeadDotProductAttention_0/out/kernel":
             hp["encoder.layers.1.self_attn.out_proj.weight"].grad.T.reshape(H, Dh, D),
         "Transformer/encoderblock_1/MlpBlock_0/Dense_0/kernel": hp["encoder.layers.1.mlp.fc1.weight"].grad.T,
         "Transformer/encoder_norm/bias": hp["post_layernorm.bias"].grad,
         "MAPHead_0/probe": hp["head.probe"].grad,
         "MAPHead_0/MultiHeadDotProductAttention_0/query/kernel": hp["head.attention.in_proj_weight"].grad[:D].T.reshape(D, H, Dh),
         "MAPHead_0/MlpBlock_0/Dense_1/kernel": hp["head.mlp.fc2.weight"].grad.T}
  res = {"seed": np.int64(SEED), "cot": cot.numpy(), "hf_pooled": pooled.detach().numpy(), "hf_hidden": hidden.detach().numpy(),
         "hf_loss": np.float64(loss.item()), "grids": np.asarray(GRIDS)}
  res.update({f"cfg_{k}": np.int64(v) for k, v in c.items()})
  for n in GRADS:
    res["hfgrad:" + n] = inv[n].detach().contiguous().numpy()
  path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "naflex_hf_tiny.npz")
  np.savez_compressed(path, **res)
  print("wrote", os.path.normpath(path), os.path.getsize(path), "bytes; loss", loss.item())
  # sanity: the oracle on the same inputs
  z, o = O.naflex_vit_forward(p, (patches, ptype, yabs, xabs), num_classes=None, width=D, depth=c["depth"], num_heads=H,
                              pool_type="map", posemb=f"learn_2d({c['slots']})")
  print("oracle vs HF pooled max-abs", (z - pooled.detach()).abs().max().item())


if __name__ == "__main__":
  main()
