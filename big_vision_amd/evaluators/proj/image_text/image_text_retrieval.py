"""Recall@k of image-text retrieval from a distance matrix.

Same functions, argument meaning and result keys as
big_vision/evaluators/proj/image_text/image_text_retrieval.py:24-85 (`dist_matrix` is
[n_images, n_texts]; `text_image_correspondence[j]` = row of the image that text j belongs
to; several texts may share an image).  Host-side numpy: it consumes the embeddings the
accelerated `predict_fn` produced.  A hit at k means the correct item
is among the k smallest distances; ties are ordered by numpy's default argsort, exactly as
in the reference (checked against it on 2000 random matrices with and without ties, and
pinned by tests/golden/retrieval_recall.npz + the reference's own known answers).
"""
from typing import List, Mapping

import numpy as np

RECALL_THRESHOLDS = (1, 5, 10)


def _recalls(hit_rank: np.ndarray) -> Mapping[str, float]:
  """hit_rank[q] = 0-based rank of the best-ranked correct answer of query q."""
  return {f"Recall@{k}": float(np.mean(hit_rank < k)) for k in RECALL_THRESHOLDS}


def text_to_image_retrieval_eval(dist_matrix: np.ndarray,
                                 text_image_correspondence: List[int]) -> Mapping[str, float]:
  """For every text (column): is its image among the k nearest images?"""
  dist = np.asarray(dist_matrix)
  corr = np.asarray(text_image_correspondence)
  order = np.argsort(dist, axis=0)            # [n_images, n_texts]: image ids by rank
  rank_of_image = np.empty_like(order)
  np.put_along_axis(rank_of_image, order, np.arange(dist.shape[0])[:, None], axis=0)
  return _recalls(rank_of_image[corr, np.arange(dist.shape[1])])


def image_to_text_retrieval_eval(dist_matrix: np.ndarray,
                                 text_image_correspondence: List[int]) -> Mapping[str, float]:
  """For every image (row): is one of ITS texts among the k nearest texts?"""
  dist = np.asarray(dist_matrix)
  corr = np.asarray(text_image_correspondence)
  order = np.argsort(dist, axis=1)            # [n_images, n_texts]: text ids by rank
  owner_by_rank = corr[order]                                # image each ranked text belongs to
  mine = owner_by_rank == np.arange(dist.shape[0])[:, None]
  first = np.where(mine.any(axis=1), mine.argmax(axis=1), np.iinfo(np.int64).max)   # images without texts never hit
  return _recalls(first)
