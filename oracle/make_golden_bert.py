"""Generate tests/golden/bert_hf_tiny.npz - pins the oracle's BERT text tower (bv_oracle.bert_forward).

TEST INFRASTRUCTURE.  Run in the build container (needs `transformers`):

    python oracle/make_golden_bert.py

models/proj/flaxformer/bert.py:33-64 wraps flaxformer's BertEncoder - un-vendored, un-pinned (so this part
of the oracle is "parity unpinned" against the reference's own arithmetic).  The independent second opinion
for the published network (original BERT: post-LN blocks, LayerNorm eps 1e-12, tanh-approximated gelu) is
HuggingFace `BertModel(hidden_act="gelu_new", add_pooling_layer=False)`.  Random parameters in the
flaxformer tree layout (init_bert, regenerated from the stored seed by the test) are copied into the HF
module ((in,out) -> (out,in), per-head projections flattened), HF runs in fp64 on padded token ids
(attention_mask = ids != 0, token_type_ids = 0) and the CLS hidden state, the head output, a scalar loss
and its autograd gradients w.r.t. a handful of parameters of every kind are stored.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bv_oracle as O  # noqa: E402

CFG = dict(hidden_size=64, intermediate_dim=128, num_hidden_layers=2, num_attention_heads=2, vocab_size=50,
           max_length=16, num_segments=2)
SEED, CLASSES = 11, 32
GRADS = ["BertEncoder_0/embedder/embedders_token_ids/embedding", "BertEncoder_0/embedder/embedders_position_ids/embedding",
         "BertEncoder_0/embedder/embedders_segment_ids/embedding", "BertEncoder_0/layer_norm/scale",
         "BertEncoder_0/encoder_block_0/attention_block/attention_layer/query/kernel",
         "BertEncoder_0/encoder_block_0/attention_block/attention_layer/key/bias",
         "BertEncoder_0/encoder_block_0/attention_block/attention_layer/out/kernel",
         "BertEncoder_0/encoder_block_0/attention_block/layer_norm/bias",
         "BertEncoder_0/encoder_block_1/mlp_block/mlp/wi/kernel", "BertEncoder_0/encoder_block_1/mlp_block/mlp/wo/bias",
         "BertEncoder_0/encoder_block_1/mlp_block/layer_norm/scale", "head/kernel"]


def make_params(dtype=torch.float64):
  gen = torch.Generator().manual_seed(SEED)
  p = O.init_bert(gen, config=CFG, num_classes=CLASSES, head_zeroinit=False, dtype=dtype)
  # break the symmetric inits (zero biases, unit scales) so every gradient path is exercised
  return O.tree_map(lambda t: t + 0.05 * torch.randn(t.shape, generator=gen, dtype=dtype), p)


def make_text():
  return torch.tensor([[5, 6, 7, 0, 0, 0, 0, 0], [9, 3, 4, 8, 2, 1, 1, 0], [2, 0, 0, 0, 0, 0, 0, 0], [4, 4, 9, 9, 3, 3, 7, 7]])


def flax_to_hf(p):
  D = CFG["hidden_size"]
  e = p["BertEncoder_0"]
  sd = {"embeddings.word_embeddings.weight": e["embedder"]["embedders_token_ids"]["embedding"],
        "embeddings.position_embeddings.weight": e["embedder"]["embedders_position_ids"]["embedding"],
        "embeddings.token_type_embeddings.weight": e["embedder"]["embedders_segment_ids"]["embedding"],
        "embeddings.LayerNorm.weight": e["layer_norm"]["scale"], "embeddings.LayerNorm.bias": e["layer_norm"]["bias"]}
  for i in range(CFG["num_hidden_layers"]):
    b = e[f"encoder_block_{i}"]
    at = b["attention_block"]["attention_layer"]
    P = f"encoder.layer.{i}."
    for n in ("query", "key", "value"):
      sd[P + f"attention.self.{n}.weight"] = at[n]["kernel"].reshape(D, D).T
      sd[P + f"attention.self.{n}.bias"] = at[n]["bias"].reshape(D)
    sd[P + "attention.output.dense.weight"] = at["out"]["kernel"].reshape(D, D).T
    sd[P + "attention.output.dense.bias"] = at["out"]["bias"]
    sd[P + "attention.output.LayerNorm.weight"] = b["attention_block"]["layer_norm"]["scale"]
    sd[P + "attention.output.LayerNorm.bias"] = b["attention_block"]["layer_norm"]["bias"]
    sd[P + "intermediate.dense.weight"] = b["mlp_block"]["mlp"]["wi"]["kernel"].T
    sd[P + "intermediate.dense.bias"] = b["mlp_block"]["mlp"]["wi"]["bias"]
    sd[P + "output.dense.weight"] = b["mlp_block"]["mlp"]["wo"]["kernel"].T
    sd[P + "output.dense.bias"] = b["mlp_block"]["mlp"]["wo"]["bias"]
    sd[P + "output.LayerNorm.weight"] = b["mlp_block"]["layer_norm"]["scale"]
    sd[P + "output.LayerNorm.bias"] = b["mlp_block"]["layer_norm"]["bias"]
  return sd


def main():
  from transformers import BertConfig, BertModel
  p = make_params()
  text = make_text()
  hc = BertConfig(vocab_size=CFG["vocab_size"], hidden_size=CFG["hidden_size"], num_hidden_layers=CFG["num_hidden_layers"],
                  num_attention_heads=CFG["num_attention_heads"], intermediate_size=CFG["intermediate_dim"],
                  hidden_act="gelu_new", max_position_embeddings=CFG["max_length"], type_vocab_size=CFG["num_segments"],
                  layer_norm_eps=1e-12, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
  m = BertModel(hc, add_pooling_layer=False).double().train(False)
  hf_leaf = {k: v.detach().clone().contiguous().requires_grad_(True) for k, v in flax_to_hf(p).items()}
  missing, unexpected = m.load_state_dict({k: v.detach() for k, v in hf_leaf.items()}, strict=False)
  assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
  hs = m(input_ids=text, attention_mask=(text != 0).long(), token_type_ids=torch.zeros_like(text)).last_hidden_state
  cls = hs[:, 0]
  head_k = p["head"]["kernel"].clone().requires_grad_(True)
  logits = cls @ head_k + p["head"]["bias"]
  gen = torch.Generator().manual_seed(SEED + 1)
  cot = torch.randn(logits.shape, generator=gen, dtype=torch.float64)
  loss = (logits * cot).sum()
  loss.backward()
  hfp = dict(m.named_parameters())
  inv = {"BertEncoder_0/embedder/embedders_token_ids/embedding": lambda: hfp["embeddings.word_embeddings.weight"].grad,
         "BertEncoder_0/embedder/embedders_position_ids/embedding": lambda: hfp["embeddings.position_embeddings.weight"].grad,
         "BertEncoder_0/embedder/embedders_segment_ids/embedding": lambda: hfp["embeddings.token_type_embeddings.weight"].grad,
         "BertEncoder_0/layer_norm/scale": lambda: hfp["embeddings.LayerNorm.weight"].grad,
         "BertEncoder_0/encoder_block_0/attention_block/attention_layer/query/kernel":
             lambda: hfp["encoder.layer.0.attention.self.query.weight"].grad.T.reshape(64, 2, 32),
         "BertEncoder_0/encoder_block_0/attention_block/attention_layer/key/bias":
             lambda: hfp["encoder.layer.0.attention.self.key.bias"].grad.reshape(2, 32),
         "BertEncoder_0/encoder_block_0/attention_block/attention_layer/out/kernel":
             lambda: hfp["encoder.layer.0.attention.output.dense.weight"].grad.T.reshape(2, 32, 64),
         "BertEncoder_0/encoder_block_0/attention_block/layer_norm/bias":
             lambda: hfp["encoder.layer.0.attention.output.LayerNorm.bias"].grad,
         "BertEncoder_0/encoder_block_1/mlp_block/mlp/wi/kernel": lambda: hfp["encoder.layer.1.intermediate.dense.weight"].grad.T,
         "BertEncoder_0/encoder_block_1/mlp_block/mlp/wo/bias": lambda: hfp["encoder.layer.1.output.dense.bias"].grad,
         "BertEncoder_0/encoder_block_1/mlp_block/layer_norm/scale": lambda: hfp["encoder.layer.1.output.LayerNorm.weight"].grad,
         "head/kernel": lambda: head_k.grad}
  out = {"seed": np.int64(SEED), "classes": np.int64(CLASSES), "text": text.numpy(), "cot": cot.numpy(),
         "hf_cls": cls.detach().numpy(), "hf_hidden": hs.detach().numpy(), "hf_logits": logits.detach().numpy(),
         "hf_loss": np.float64(loss.item())}
  out.update({f"cfg_{k}": np.int64(v) for k, v in CFG.items()})
  for n in GRADS:
    out["hfgrad:" + n] = inv[n]().detach().contiguous().numpy()
  path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "bert_hf_tiny.npz")
  np.savez_compressed(path, **out)
  print("wrote", os.path.normpath(path), os.path.getsize(path), "bytes; loss", loss.item())


if __name__ == "__main__":
  main()
