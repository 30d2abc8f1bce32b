"""Golden vectors for the zero-shot classification evaluator from the REFERENCE's own code.

big_vision/evaluators/proj/image_text/discriminative_classifier.py cannot be imported here (it pulls
in jax / tensorflow / tfds at module level), but its per-class averaging `_average_embeddings`
(:145-166) is plain numpy: this script extracts that one function from the reference checkout by
AST, executes it unchanged on seeded inputs and writes inputs + outputs to
tests/golden/zeroshot_average.npz.  Also records the reference's decision rule on a small logits
matrix (argmax over classes, correct if it matches ANY label of the example, masked: :305-318),
restated in numpy next to it.  Run in the build container:  python oracle/make_zeroshot_golden.py
"""
import ast
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/big_vision/evaluators/proj/image_text/discriminative_classifier.py"


def reference_fn(name):
  src = open(REF).read()
  node = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == name][0]
  ns = {"np": np}
  exec(compile(ast.Module([node], []), REF, "exec"), ns)   # the reference's function, unmodified
  return ns[name]


def main():
  avg = reference_fn("_average_embeddings")
  rng = np.random.RandomState(0)
  emb = rng.randn(37, 16).astype(np.float32)
  labels = np.concatenate([np.arange(7), rng.randint(0, 7, 30)]).astype(np.int64)
  out_n = avg(emb.copy(), labels=labels, num_classes=7, normalize=True)
  out_u = avg(emb.copy(), labels=labels, num_classes=7, normalize=False)
  # decision rule (:305-318), numpy restatement
  zimg = rng.randn(9, 16).astype(np.float32)
  best = (zimg @ out_n.T).argmax(axis=1)
  lab2 = np.stack([rng.randint(0, 7, 9), rng.randint(0, 7, 9)], 1)
  mask = np.array([1, 1, 1, 0, 1, 1, 0, 1, 1], bool)
  matching = (best[:, None] == lab2).sum(axis=1)
  correct = int(np.where(mask, (matching > 0).astype(np.int32), 0).sum())
  np.savez(os.path.join(ROOT, "tests", "golden", "zeroshot_average.npz"), emb=emb, labels=labels, avg_norm=out_n,
           avg_raw=out_u, zimg=zimg, best=best, labels2=lab2, mask=mask, correct=correct)
  print("wrote tests/golden/zeroshot_average.npz", out_n.shape, correct)


if __name__ == "__main__":
  main()
