"""Recall@k of the retrieval evaluator (host numpy) against (1) the reference's own known
answers, big_vision/evaluators/proj/image_text/image_text_retrieval_test.py:26-80, restated with
the same numbers, and (2) tests/golden/retrieval_recall.npz = outputs of the reference module
itself on seeded random matrices (oracle/make_retrieval_golden.py)."""
import os

import numpy as np
import pytest

from big_vision_amd.evaluators.proj.image_text import image_text_retrieval as itr

CORR = [0, 0, 1, 1, 2, 2, 3, 3]
PERFECT = np.array([[0.0, 0.0, 0.1, 0.5, 0.1, 0.2, 0.5, 0.1],
                    [0.5, 0.4, 0.0, 0.0, 0.4, 0.2, 0.6, 0.4],
                    [0.5, 0.4, 0.1, 0.5, 0.0, 0.0, 0.8, 0.3],
                    [0.5, 0.4, 0.1, 0.5, 0.3, 0.2, 0.0, 0.0]])
I2T = np.array([[0.8, 0.8, 0.1, 0.5, 0.1, 0.2, 0.5, 0.1],
                [0.5, 0.4, 0.0, 0.0, 0.4, 0.2, 0.6, 0.4],
                [0.5, 0.4, 0.1, 0.5, 0.0, 0.8, 0.8, 0.3],
                [0.5, 0.4, 0.1, 0.5, 0.4, 0.2, 0.3, 0.3]])
T2I = np.array([[0.8, 0.8, 0.1, 0.5, 0.1, 0.2, 0.1, 0.1],
                [0.5, 0.4, 0.0, 0.0, 0.4, 0.2, 0.6, 0.4],
                [0.5, 0.4, 0.1, 0.5, 0.0, 0.8, 0.8, 0.3],
                [0.5, 0.4, 0.1, 0.5, 0.4, 0.2, 0.3, 0.3]])


@pytest.mark.parametrize("dist,expected", [
    (PERFECT, {"Recall@1": 1.0, "Recall@5": 1.0, "Recall@10": 1.0}),
    (I2T, {"Recall@1": 0.5, "Recall@5": 0.75, "Recall@10": 1.0})])
def test_image_to_text_known_answers(dist, expected):
  assert itr.image_to_text_retrieval_eval(dist, CORR) == expected


@pytest.mark.parametrize("dist,expected", [
    (PERFECT, {"Recall@1": 1.0, "Recall@5": 1.0, "Recall@10": 1.0}),
    (T2I, {"Recall@1": 0.375, "Recall@5": 1.0, "Recall@10": 1.0})])
def test_text_to_image_known_answers(dist, expected):
  assert itr.text_to_image_retrieval_eval(dist, CORR) == expected


def test_golden_from_the_reference_module():
  z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "retrieval_recall.npz"))
  for i in range(4):
    d, c = z[f"d{i}"], z[f"c{i}"].tolist()
    t2i = itr.text_to_image_retrieval_eval(d, c)
    i2t = itr.image_to_text_retrieval_eval(d, c)
    np.testing.assert_array_equal([t2i[f"Recall@{k}"] for k in (1, 5, 10)], z[f"t2i{i}"])
    np.testing.assert_array_equal([i2t[f"Recall@{k}"] for k in (1, 5, 10)], z[f"i2t{i}"])


def test_edge_cases():
  one = np.zeros((1, 1))
  assert itr.text_to_image_retrieval_eval(one, [0]) == {"Recall@1": 1.0, "Recall@5": 1.0, "Recall@10": 1.0}
  # an image without any caption can never be retrieved image->text, even with fewer than k texts
  d = np.array([[0.1, 0.9], [0.8, 0.2], [0.5, 0.5]])
  r = itr.image_to_text_retrieval_eval(d, [0, 1])
  assert r["Recall@1"] == pytest.approx(2 / 3) and r["Recall@10"] == pytest.approx(2 / 3)


def test_zeroshot_class_average_matches_the_reference_function():
  """tests/golden/zeroshot_average.npz holds outputs of the reference's own `_average_embeddings`
  (discriminative_classifier.py:145-166, executed by oracle/make_zeroshot_golden.py)."""
  import os
  from big_vision_amd.evaluators.proj.image_text import discriminative_classifier as dc
  g = np.load(os.path.join(os.path.dirname(__file__), "golden", "zeroshot_average.npz"))
  got = dc._average_embeddings(g["emb"], labels=g["labels"], num_classes=7, normalize=True)
  np.testing.assert_allclose(got, g["avg_norm"], rtol=1e-6, atol=1e-7)
  got = dc._average_embeddings(g["emb"], labels=g["labels"], num_classes=7, normalize=False)
  np.testing.assert_allclose(got, g["avg_raw"], rtol=1e-6, atol=1e-7)
  with pytest.raises(AssertionError, match="Classes without embeddings"):
    dc._average_embeddings(g["emb"], labels=g["labels"], num_classes=8, normalize=True)


def test_zeroshot_prompt_expansion():
  from big_vision_amd.evaluators.proj.image_text import discriminative_classifier as dc
  got = dc.expand_prompts(["cat,kitty", "dog"], ["a photo of a {}.", "{}"])
  assert got == [(0, "a photo of a cat."), (0, "cat"), (1, "a photo of a dog."), (1, "dog")]
  got = dc.expand_prompts(["cat,kitty"], ["{}!"], first_class_name_only=False)
  assert got == [(0, "cat!"), (0, "kitty!")]
