"""jax.image.scale_and_translate, RESTATED (jax/_src/image/scale.py `_scale_and_translate` / `compute_weight_mat`, bilinear
= triangle kernel): the NaFlex tower resizes its learned position grid with it (naflex_vit.py:61-68).  Output sample o of a
spatial axis sits at input coordinate (o + 0.5) / scale - translation / scale - 0.5; with antialias the kernel widens by
1 / scale when downsampling; weights are normalised per output sample and zeroed where the sample lies outside the input."""
import numpy as np


def _weight_mat(input_size, output_size, scale, translation, antialias):
  inv_scale = 1.0 / scale
  kernel_scale = max(inv_scale, 1.0) if antialias else 1.0
  sample_f = (np.arange(output_size) + 0.5) * inv_scale - translation * inv_scale - 0.5
  x = np.abs(sample_f[None, :] - np.arange(input_size)[:, None]) / kernel_scale
  weights = np.maximum(0.0, 1.0 - np.abs(x))
  total = np.sum(weights, axis=0, keepdims=True)
  weights = np.where(np.abs(total) > 1000.0 * float(np.finfo(np.float32).eps), weights / np.where(total != 0, total, 1.0), 0.0)
  inside = np.logical_and(sample_f >= -0.5, sample_f <= input_size - 0.5)
  return np.where(inside[None, :], weights, 0.0)


def scale_and_translate(image, shape, spatial_dims, scale, translation, method, antialias=True, precision=None):
  del precision
  assert method in ("bilinear", "linear", "triangle"), method
  out = np.asarray(image, np.float64)
  for i, d in enumerate(spatial_dims):
    w = _weight_mat(out.shape[d], shape[d], float(scale[i]), float(translation[i]), antialias)      # [in, out]
    out = np.moveaxis(np.tensordot(out, w, axes=([d], [0])), -1, d)
  assert tuple(out.shape) == tuple(shape), (out.shape, shape)
  return out
