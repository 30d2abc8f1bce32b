"""ctypes binding of libbvhip.so (the C ABI declared in include/bvhip.h).

The product path has NO fallback: if the shared library is missing or a call
fails, a RuntimeError is raised.  Build it with `python big_vision_amd/build.py`
(or `__graft_entry__.build()`).
"""
import ctypes
import os

# torch bundles its own libamdhip64; it must be loaded BEFORE libbvhip.so so both
# share ONE HIP runtime (otherwise libbvhip binds /opt/rocm's copy and launches
# fail with "no ROCm-capable device").
import torch  # noqa: F401  pylint: disable=unused-import

from ctypes import c_int, c_long, c_float, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbvhip.so")

P = c_void_p

# name -> argtypes (return type is always int, except bv_last_error)
PROTOTYPES = {
    "bv_version": [],
    # the caller's context (options, split-K workspace, launch counters): the library keeps no global state
    "bv_ctx_create": [], "bv_ctx_destroy": [P], "bv_ctx_set": [P, c_int, c_long], "bv_ctx_get": [P, c_int],
    "bv_ctx_set_workspace": [P, P, c_long],
    "bv_gemm_bf16": [c_int, c_int, P, c_long, P, c_long, P, c_long, c_int, c_int, c_int, c_int,
                     c_int, P, P, c_long, c_int, P, c_float, c_int, P, P],
    "bv_gemm_bf16_colsum": [c_int, c_int, P, c_long, P, c_long, P, c_long, c_int, c_int, c_int, c_int,
                            c_int, P, P, c_long, c_int, P, c_float, c_int, P, P, P],
    "bv_gemm_workspace_bytes": [c_int, c_int, c_int],
    "bv_sgemm_strided": [P, c_long, c_long, P, c_long, c_long, P, c_long, c_int, c_int, c_int,
                         c_float, c_float, P, P, P],
    "bv_layernorm_fwd": [P, P, P, P, P, P, P, c_int, c_int, c_long, c_long, c_float, P],
    "bv_layernorm_fwd_bf16x": [P, P, P, P, P, P, P, c_int, c_int, c_long, c_long, c_float, P],
    "bv_layernorm_bwd": [P, c_int, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_long, c_long, P],
    "bv_layernorm_bwd_y": [P, c_int, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_long, c_long, P, P, P],
    "bv_layernorm_bwd_bf16x": [P, c_int, P, P, P, P, P, P, P, P, P, c_int, c_int, c_long, c_long, P],
    "bv_attn_fwd": [P, P, P, c_int, c_int, c_int, P, P],
    "bv_attn_bwd": [P, P, P, P, P, P, P, c_int, c_int, c_int, P, P],
    "bv_attn_fwd_masked": [P, P, P, P, c_int, c_int, c_int, P, P],
    "bv_attn_bwd_masked": [P, P, P, P, P, P, P, c_int, c_int, c_int, P, P],
    "bv_map_attn_fwd": [P, P, P, P, c_int, c_int, c_int, P],
    "bv_map_attn_bwd": [P, P, P, P, P, P, c_int, c_int, c_int, P],
    "bv_map_attn_fwd_masked": [P, P, P, P, P, c_int, c_int, c_int, P],
    "bv_attn_fwd_dh": [P, P, P, P, c_int, c_int, c_int, c_int, P],
    "bv_attn_bwd_dh": [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "bv_map_attn_fwd_dh": [P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "bv_map_attn_bwd_dh": [P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "bv_pool_gap_masked_fwd": [P, P, P, c_int, c_int, c_int, P],
    "bv_pool_gap_masked_bwd": [P, P, P, c_int, c_int, c_int, P],
    "bv_naflex_posemb_weights": [P, P, P, c_int, c_int, c_int, P],
    "bv_patchify": [P, P, c_int, c_int, c_int, c_int, P],
    "bv_patchify_ld": [P, P, c_int, c_int, c_int, c_int, c_int, P],
    "bv_embed_fwd": [P, P, P, P, c_int, c_int, c_int, c_int, P],
    "bv_embed_bwd": [P, P, P, c_int, c_int, c_int, P],
    "bv_colsum": [P, c_int, c_long, P, c_int, c_int, P],
    "bv_batchsum": [P, P, c_int, c_int, c_int, P],
    "bv_cast_bf16": [P, P, c_long, P],
    "bv_cast_f32": [P, P, c_long, P],
    "bv_transpose_bf16": [P, P, c_int, c_int, c_long, c_long, P],
    "bv_transpose_bf16_batched": [P, c_int, c_int, P],
    "bv_concat_cls": [P, P, P, c_int, c_int, c_int, P],
    "bv_pool_gap_fwd": [P, P, c_int, c_int, c_int, P],
    "bv_pool_gap_bwd": [P, P, c_int, c_int, c_int, P],
    "bv_pool_max_fwd": [P, P, P, c_int, c_int, c_int, P],
    "bv_pool_max_masked_fwd": [P, P, P, P, c_int, c_int, c_int, P],
    "bv_pool_max_bwd": [P, P, P, c_int, c_int, c_int, P],
    "bv_l2norm_fwd": [P, P, P, c_int, c_int, c_float, P],
    "bv_l2norm_bwd": [P, P, P, P, c_int, c_int, c_float, P],
    "bv_siglip_loss": [P, P, P, P, c_int, c_int, c_int, c_int, P],
    "bv_logit_stats": [P, P, P, P, P, c_int, c_int, c_int, P],
    "bv_dot_f32": [P, P, c_long, P, P],
    "bv_softmax_xent": [P, P, P, P, c_int, c_int, c_int, P],
    "bv_sigmoid_xent": [P, P, P, P, c_int, c_int, c_int, P],
    "bv_tanh_fwd": [P, P, c_long, P],
    "bv_tanh_bwd": [P, P, P, c_long, P],
    "bv_mixup": [P, P, c_float, c_int, c_long, P],
    "bv_dropout_f32": [P, P, P, P, c_long, ctypes.c_ulonglong, c_float, P],
    "bv_dropout_bf16": [P, P, c_long, ctypes.c_ulonglong, c_float, P],
    "bv_dropout_mask": [P, c_long, ctypes.c_ulonglong, c_float, P],
    "bv_sqnorm": [P, c_long, P, P],
    "bv_adam_step": [P, P, P, c_int, P, P, P, P, c_long, P, c_int, P, c_float, c_float, c_float,
                     c_float, c_float, c_float, P, P],
}

PROTOTYPES["bv_adafactor_leaf"] = [P, P, P, c_int, P, P, P, c_int, P, c_float, c_float, c_float, c_float,
                                   c_float, c_float, c_float, P, P]

PROTOTYPES["bv_adafactor_step"] = [P, P, P, c_int, P, P, c_int, c_long, c_long, c_long, c_long, P, P, c_float, c_float,
                                   c_float, c_float, P, c_int, P, c_float, P, P]

# collectives for non-Python hosts (csrc/comm.cpp; the Python host uses torch.distributed, dp.py)
PROTOTYPES.update({
    "bv_comm_version": [P], "bv_comm_unique_id": [P], "bv_comm_init": [P, c_int, c_int, P], "bv_comm_destroy": [P],
    "bv_comm_all_gather": [P, P, P, c_long, c_int, P], "bv_comm_reduce_scatter": [P, P, P, c_long, c_int, P],
    "bv_comm_all_reduce_bucket": [P, P, c_long, c_long, c_int, P],
})

# everything else returns an int status
RESTYPES = {"bv_gemm_workspace_bytes": c_long, "bv_ctx_create": P, "bv_ctx_destroy": None, "bv_ctx_set": c_long,
            "bv_ctx_get": c_long}

# bv_ctx options / statistics (include/bvhip.h)
OPTS = {"fast_path": 0, "gemm_nt": 1, "gemm_skew_mode": 2, "gemm_skew_pct": 3, "gemm_pre_issue": 4, "gemm_roll": 5,
        "gemm_group_n": 6, "gemm_reserve_cus": 7, "attn_cfg": 8, "sgemm_mfma": 9,
        "gemm256_calls": 100, "gemm256_multi": 101, "gemm256_fused": 102}

EPI_NONE, EPI_RESIDUAL, EPI_POS, EPI_GELU, EPI_GELU_BWD, EPI_ATOMIC, EPI_GELU_BWD_EMIT, EPI_GELU_GD, EPI_MUL, EPI_GELU_G = range(10)

_lib = None


def load():
  """Loads libbvhip.so (once).  Raises if it is not built — no fallback."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise RuntimeError(
        f"{LIB_PATH} is missing: the HIP extension is not built. Run "
        "`python big_vision_amd/build.py`. There is no CPU/eager fallback.")
  lib = ctypes.CDLL(LIB_PATH)
  lib.bv_last_error.restype = ctypes.c_char_p
  lib.bv_last_error.argtypes = []
  for name, argtypes in PROTOTYPES.items():
    fn = getattr(lib, name)  # AttributeError if the ABI drifted
    fn.restype = RESTYPES.get(name, c_int)
    fn.argtypes = argtypes
  if lib.bv_version() != 2:
    raise RuntimeError("libbvhip.so ABI version mismatch")
  _lib = lib
  return lib


# Optional launch observer (bench.py installs one to bracket selected kernels
# with HIP events on the launch stream); None on the normal path.
observer = None


def call(name, *args):
  lib = load()
  obs = observer
  tok = obs.begin(name, args) if obs is not None else None
  rc = getattr(lib, name)(*args)
  if tok is not None:
    obs.end(tok)
  if name in RESTYPES:
    return rc
  if rc != 0:
    raise RuntimeError(f"{name} failed (rc={rc}): {lib.bv_last_error().decode()}")
