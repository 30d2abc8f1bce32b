"""Discriminative zero-shot classification evaluator on top of the accelerated `predict_fn`.

Mirrors big_vision/evaluators/proj/image_text/discriminative_classifier.py:145-440:
  1. every class name is expanded with every prompt template, tokenised and embedded with
     `_, ztxt, _ = predict_fn(train_state, {"labels": texts})` (:300-303);
  2. the embeddings are averaged per class and L2-normalised with eps 1e-8
     (`_average_embeddings` :145-166; a class without any embedding is an error);
  3. every image batch is embedded with `zimg, _, _ = predict_fn(train_state, {"image": image})`,
     `best_txt = argmax(zimg . ztxt^T)` and an example is correct when the best class equals ANY of
     its labels (labels may be [n] or multi-label [n, k]); padded examples are masked out (:305-322);
  4. `evaluate` -> {"accuracy", "correct", "count"} (+ embeddings on request, :430-438), `run` ->
     [("<dataset>_accuracy", value)] (:442-446).
The reference builds its datasets from TFDS with prompt_engineering / pp strings (input pipeline:
out of the hot-path scope); here a dataset is given as arrays: images [N, H, W, 3]
fp32 with integer labels [N] or [N, k], and the already tokenised prompts [P, Lt] int32 with the
class index of every prompt (P = classes x templates when built with `expand_prompts`).  The
similarity / argmax / correct count runs on the GPU (fp32 strided GEMM of libbvhip for the
[batch, classes] logits); batches are padded to one shape and masked like the reference's
`mask` feature.
"""
import numpy as np
import torch

from big_vision_amd import ops


def expand_prompts(class_names, prompt_templates, first_class_name_only=True):
  """[(class index, text)] for every class x template (prepare_datasets :85-106): a class name may
  list aliases separated by ','; a template contains exactly one '{}'."""
  assert prompt_templates, "Must specify prompt templates (e.g. simply ['{}'])"
  out = []
  for idx, name in enumerate(class_names):
    aliases = name.split(",")
    if first_class_name_only:
      aliases = aliases[:1]
    for alias in aliases:
      for tpl in prompt_templates:
        parts = tpl.split("{}")
        assert len(parts) == 2, tpl
        out.append((idx, parts[0] + alias + parts[1]))
  return out


def _average_embeddings(embeddings, *, labels, num_classes, normalize):
  """Per-class averages of `embeddings` (:145-166)."""
  embeddings = np.asarray(embeddings)
  labels = np.asarray(labels)
  assert embeddings.ndim == 2, f"Expected {embeddings.ndim}==2"
  assert labels.ndim == 1, f"Expected {labels.ndim}==1"
  assert len(labels) == len(embeddings), f"Expected {len(labels)}=={len(embeddings)}"
  sums = np.zeros((num_classes, embeddings.shape[1]), np.float64)
  counts = np.zeros(num_classes, np.int64)
  np.add.at(sums, labels, embeddings.astype(np.float64))
  np.add.at(counts, labels, 1)
  missing = set(np.nonzero(counts == 0)[0].tolist())
  assert not missing, f"Classes without embeddings: {missing}"
  avg = (sums / counts[:, None]).astype(embeddings.dtype)
  if normalize:
    avg = avg / (1e-8 + np.linalg.norm(avg, axis=1, keepdims=True))
  return avg


class Evaluator:
  """Zero-shot classification evaluator."""

  def __init__(self, predict_fn, *, datasets, batch_size=64, device=None):
    """datasets: {name: dict(images=[N,H,W,3], labels=[N] or [N,k], prompts=[P,Lt] int32,
    prompt_labels=[P], num_classes=int (optional))}."""
    self.predict_fn = predict_fn
    self.datasets = {}
    for name, d in datasets.items():
      pl = np.asarray(d["prompt_labels"])
      if len(pl) != len(d["prompts"]):
        raise ValueError(f"{name}: {len(d['prompts'])} prompts but {len(pl)} prompt labels")
      if len(d["labels"]) != len(d["images"]):
        raise ValueError(f"{name}: {len(d['images'])} images but {len(d['labels'])} labels")
      self.datasets[name] = dict(d, prompt_labels=pl, num_classes=int(d.get("num_classes", pl.max() + 1)))
    self.batch_size = int(batch_size)
    self.device = device or torch.device("cuda", torch.cuda.current_device())

  def _batches(self, data):
    bs, n = self.batch_size, len(data)
    for s in range(0, n, bs):
      chunk = torch.as_tensor(np.asarray(data[s:s + bs]) if not torch.is_tensor(data) else data[s:s + bs])
      valid = chunk.shape[0]
      if valid < bs:   # pad with copies of the last element, masked out below
        chunk = torch.cat([chunk, chunk[-1:].expand(bs - valid, *chunk.shape[1:])])
      yield s, valid, chunk.to(self.device)

  def _embed_texts(self, train_state, name):
    d = self.datasets[name]
    embs = []
    for _, valid, texts in self._batches(d["prompts"]):
      _, ztxt, _ = self.predict_fn(train_state, {"labels": texts})
      embs.append(ztxt[:valid].detach().float().cpu().numpy())
    emb = np.concatenate(embs)
    return {"embedding": emb, "label": d["prompt_labels"],
            "average_embedding": _average_embeddings(emb, labels=d["prompt_labels"],
                                                     num_classes=d["num_classes"], normalize=True)}

  def evaluate(self, train_state, dataset_name, *, return_embeddings=False):
    """Returns evaluation results (:365-438)."""
    d = self.datasets[dataset_name]
    texts = self._embed_texts(train_state, dataset_name)
    ztxt = torch.from_numpy(np.ascontiguousarray(texts["average_embedding"], np.float32)).to(self.device)
    C, E = ztxt.shape
    labels_all = torch.as_tensor(np.asarray(d["labels"]))
    if labels_all.dim() == 1:
      labels_all = labels_all[:, None]
    correct, count, embs = 0, 0, []
    logits = torch.empty((self.batch_size, C), device=self.device, dtype=torch.float32)
    for s, valid, image in self._batches(d["images"]):
      zimg, _, _ = self.predict_fn(train_state, {"image": image})
      zimg = zimg.float().contiguous()
      ops.sgemm(zimg, E, 1, ztxt, 1, E, logits, self.batch_size, C, E)          # zimg . ztxt^T
      best = logits[:valid].argmax(dim=1)                                       # (:308)
      lab = labels_all[s:s + valid].to(self.device)
      matching = (best[:, None] == lab).sum(dim=1)
      correct += int((matching > 0).sum().item())
      count += valid
      if return_embeddings:
        embs.append(zimg[:valid].cpu().numpy())
    ret = {"accuracy": correct / count, "correct": correct, "count": count}
    if return_embeddings:
      ret["images"] = {"embedding": np.concatenate(embs), "label": np.asarray(d["labels"])}
      ret["texts"] = texts
    return ret

  def run(self, train_state):
    """Returns metrics (:442-446)."""
    return [(f"{name}_accuracy", self.evaluate(train_state, name)["accuracy"]) for name in self.datasets]
