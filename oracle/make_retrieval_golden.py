"""Generates tests/golden/retrieval_recall.npz by running the REFERENCE's numpy-only
big_vision/evaluators/proj/image_text/image_text_retrieval.py (importable in the build
container: no jax/tf dependency) on seeded random distance matrices.  Test infrastructure
only; run from the repo root with /root/reference mounted:

  python oracle/make_retrieval_golden.py
"""
import importlib.util
import numpy as np

spec = importlib.util.spec_from_file_location(
    "ref_itr", "/root/reference/big_vision/evaluators/proj/image_text/image_text_retrieval.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)
g = np.random.default_rng(7)
gold = {}
for i, (ni, nt) in enumerate([(4, 8), (7, 30), (12, 12), (25, 100)]):
  d = g.normal(size=(ni, nt)).astype(np.float64)
  corr = g.integers(0, ni, size=nt)
  gold[f"d{i}"] = d
  gold[f"c{i}"] = corr
  gold[f"t2i{i}"] = np.array([ref.text_to_image_retrieval_eval(d, list(corr))[f"Recall@{k}"] for k in (1, 5, 10)])
  gold[f"i2t{i}"] = np.array([ref.image_to_text_retrieval_eval(d, list(corr))[f"Recall@{k}"] for k in (1, 5, 10)])
np.savez("tests/golden/retrieval_recall.npz", **gold)
print("wrote tests/golden/retrieval_recall.npz")
