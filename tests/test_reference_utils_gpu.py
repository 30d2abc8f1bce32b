"""The PRODUCT's classification-loss and mixup kernels (bv_sigmoid_xent, bv_softmax_xent, bv_mixup through the C ABI) on the
fixture the reference's own utils.py produced (tests/golden/refutils.npz, oracle/run_reference_utils.py): losses to 1e-6
relative (fp32 kernels, float64 reference), the gradient against the finite difference of the reference's loss values is left
to tests/test_train_step_gpu.py; mixup to 1 ulp of fp32."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "refutils.npz")


def test_loss_kernels_equal_the_executed_reference(dev):
  from big_vision_amd import ops
  z = np.load(GOLDEN)
  logits = torch.from_numpy(z["in/logits"].astype(np.float32)).to(dev)
  for kernel, cases in ((ops.sigmoid_xent, [("sigmoid_xent", l) for l in ("hard", "soft", "multi")]),
                        (ops.softmax_xent, [("softmax_xent", l) for l in ("hard", "soft")])):
    for fam, lab in cases:
      labels = torch.from_numpy(z[f"in/{lab}"].astype(np.float32)).to(dev)
      acc = torch.zeros(1, device=dev, dtype=torch.float64)
      d = kernel(logits.contiguous(), labels.contiguous(), acc, want_grad=True)
      torch.cuda.synchronize()
      want = float(z[f"{fam}/{lab}/mean"])
      assert abs(acc.item() - want) <= 2e-6 * abs(want), (fam, lab, acc.item(), want)
      # d(mean loss)/d(logits) in closed form from the fixture's inputs: (sigmoid(x) - y) / n resp. (softmax(x) sum(y) - y) / n
      x, y = z["in/logits"], z[f"in/{lab}"]
      n = x.shape[0]
      if fam == "sigmoid_xent":
        ref = (1.0 / (1.0 + np.exp(-x)) - y) / n
      else:
        e = np.exp(x - x.max(-1, keepdims=True))
        ref = (e / e.sum(-1, keepdims=True) * y.sum(-1, keepdims=True) - y) / n
      assert np.max(np.abs(d.cpu().double().numpy() - ref)) <= 2e-6 * np.max(np.abs(ref)), (fam, lab)


def test_mixup_kernel_equals_the_executed_reference(dev):
  from big_vision_amd import ops
  z = np.load(GOLDEN)
  a = float(z["mixup/a"])
  for k in ("images", "labels"):
    x = torch.from_numpy(z[f"in/{k}"].astype(np.float32)).to(dev).contiguous()
    got = ops.mixup(x, a).cpu().double().numpy()
    assert np.max(np.abs(got - z[f"mixup/{k}"])) <= 3e-7 * max(1.0, float(np.max(np.abs(z[f"mixup/{k}"])))), k


def test_get_mixup_has_the_reference_call_shape(dev):
  """`rng, (images, labels), _ = u.get_mixup(rng, p)(images, labels)` (train.py:285-289) and the legacy keyword spelling
  (utils.py:1158-1159) unpack as they do on the reference; the mixing itself is the kernel checked above."""
  from big_vision_amd import ops, utils as u
  z = np.load(GOLDEN)
  images = torch.from_numpy(z["in/images"].astype(np.float32)).to(dev)
  labels = torch.from_numpy(z["in/labels"].astype(np.float32)).to(dev)
  fn = u.get_mixup(7, 0.2)
  rng, (mi, ml), more = fn(images, labels)
  assert rng == 7 and more == {} and 0.5 <= fn.a <= 1.0
  assert torch.equal(mi, ops.mixup(images.contiguous(), fn.a)) and torch.equal(ml, ops.mixup(labels.contiguous(), fn.a))
  rng2, things, kw = u.mixup(7, images, p=0.2, labels=labels)
  assert len(things) == 1 and set(kw) == {"labels"} and torch.equal(things[0], mi) and torch.equal(kw["labels"], ml)
