"""Zero-shot image/text retrieval evaluator on top of the accelerated `predict_fn`.

Mirrors big_vision/evaluators/proj/image_text/retrieval.py:150-306: `Evaluator(predict_fn,
...)` embeds all images and all captions with `zimg, ztxt, out = predict_fn(train_state,
batch)`, forms `similarities = zimg . ztxt^T` on the host and reports
`img2txt_recall@{1,5,10}` / `txt2img_recall@{1,5,10}` (`run`) or the full dictionary with
the embeddings (`evaluate`).  The reference reads a TFDS dataset through its input pipeline
(out of scope here); this evaluator takes the already pre-processed arrays:
images [N, H, W, 3] fp32, tokenised captions [M, Lt] int32 and the image id every caption
belongs to.  Batches are padded to the fixed `batch_size` and masked, like the reference's
`mask` feature, so the kernels always see one shape.
"""
import numpy as np
import torch

from big_vision_amd.evaluators.proj.image_text import image_text_retrieval


class Evaluator:
  """Image/text retrieval evaluator."""

  def __init__(self, predict_fn, *, images, texts, image_ids=None, text_image_ids=None, batch_size=64,
               device=None):
    self.predict_fn = predict_fn
    self.images, self.texts = images, texts
    n_img, n_txt = len(images), len(texts)
    self.image_ids = np.arange(n_img) if image_ids is None else np.asarray(image_ids)
    if text_image_ids is None:
      if n_txt % n_img:
        raise ValueError("text_image_ids is required unless every image has the same number of captions")
      text_image_ids = np.repeat(self.image_ids, n_txt // n_img)
    self.text_image_ids = np.asarray(text_image_ids)
    if len(self.text_image_ids) != n_txt:
      raise ValueError(f"{n_txt} captions but {len(self.text_image_ids)} caption->image ids")
    self.batch_size = int(batch_size)
    self.device = device or torch.device("cuda", torch.cuda.current_device())

  def _embed(self, name, train_state, data):
    """Embeds `data` in fixed-size batches; the padded tail of the last batch is masked out."""
    out, n, bs = [], len(data), self.batch_size
    for s in range(0, n, bs):
      chunk = torch.as_tensor(np.asarray(data[s:s + bs]) if not torch.is_tensor(data) else data[s:s + bs])
      valid = chunk.shape[0]
      if valid < bs:
        pad = chunk[-1:].expand(bs - valid, *chunk.shape[1:])
        chunk = torch.cat([chunk, pad])
      zimg, ztxt, _ = self.predict_fn(train_state, {name: chunk.to(self.device)})
      z = zimg if name == "image" else ztxt
      out.append(z[:valid].detach().float().cpu().numpy())
    return np.concatenate(out)

  def evaluate(self, train_state):
    """Returns evaluation results (retrieval.py:264-291)."""
    images = {"embeddings": self._embed("image", train_state, self.images), "id": self.image_ids}
    texts = {"embeddings": self._embed("labels", train_state, self.texts), "id": self.text_image_ids}
    similarities = np.dot(images["embeddings"], texts["embeddings"].T)
    id2img = {id_: i for i, id_ in enumerate(images["id"].tolist())}
    corr = [id2img[id_] for id_ in texts["id"].tolist()]
    return dict(
        images=images, texts=texts, similarities=similarities,
        img2txt=image_text_retrieval.image_to_text_retrieval_eval(-similarities, corr),
        txt2img=image_text_retrieval.text_to_image_retrieval_eval(-similarities, corr))

  def run(self, train_state):
    """Yields (metric name, value) pairs (retrieval.py:293-300)."""
    results = self.evaluate(train_state)
    return [(f"{direction}_{k.lower()}", v)
            for direction in ("img2txt", "txt2img")
            for k, v in results[direction].items()]
