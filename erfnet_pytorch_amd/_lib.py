"""ctypes binding of libermi355.so (C ABI declared in include/erfnet_mi355.h).

The library is the product: there is no CPU or PyTorch fallback.  Importing this module
without the built shared object raises, and every call checks the library's status code.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_double, c_float, c_int, c_longlong, c_size_t, c_ulonglong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libermi355.so")

F32, BF16 = 0, 1
AXIS_H, AXIS_W = 0, 1

_ERRORS = {-1: "invalid argument", -2: "unsupported shape or dtype", -3: "kernel launch failed",
           -4: "size beyond the fixed limit of the no-allocation entry point (C <= 128)"}


class ErfError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C erfnet_pytorch_amd/csrc` (hipcc, --offload-arch=gfx950). There is no fallback path.")
    return ctypes.CDLL(LIB_PATH)


_lib = _load()

P = c_void_p
I = c_int
LL = c_longlong
F = c_float
D = c_double
ULL = c_ulonglong

# name -> (restype, argtypes); every int-returning entry is status-checked by `call`
SIGNATURES = {
    "erf_version": (I, []),
    "erf_conv1d_packed_bytes": (c_size_t, [I, I]),
    "erf_conv1d_pack": (I, [P, P, I, I, I, P]),
    "erf_conv1d_factorized": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "erf_maxpool2_avgpool_fwd": (I, [P, I, P, I, I, I, I, P]),
    "erf_maxpool2_avgpool_bwd": (I, [P, P, P, I, I, I, I, I, P]),
    "erf_linear_fwd": (I, [P, P, P, P, I, I, I, P]),
    "erf_linear_bwd": (I, [P, P, P, P, P, P, I, I, I, P]),
    "erf_conv1d_factorized_bn": (I, [P, P, P, P, I, I, I, I, I, I, I, P, P, P, P, ctypes.c_float, ctypes.c_float, P, P, P, P, P, P]),
    "erf_conv1d_dgrad_bias": (I, [P, P, P, P, P, I, I, I, I, I, I, I, P, P, P]),
    "erf_output_conv_ce_workspace_bytes": (c_size_t, [I, I, I]),
    "erf_output_conv_ce_fwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "erf_output_conv_ce_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "erf_wgrad_reduce_desc_bytes": (c_size_t, []),
    "erf_conv1d_wgrad_partial": (I, [P, P, P, I, I, I, I, I, I, I, P, c_size_t, P, P]),
    "erf_wgrad_reduce_many": (I, [P, I, P]),
    "erf_wgrad_reduce_record_begin": (I, [P, I]),
    "erf_wgrad_reduce_record_end": (I, []),
    "erf_conv1d_wgrad_partial_many": (I, [P, I, I, I, I, I, I, c_size_t, P, P]),
    "erf_conv1d_dgrad_bias_partial": (I, [P, P, P, P, P, I, I, I, I, I, I, I, P, P]),
    "erf_channel_sum_finalize_many": (I, [P, I, P]),
    "erf_conv1d_dgrad_bn_backward": (I, [P, P, P, P, P, I, I, I, I, I, I, I, P, P, P, P, P, I, P, P, P, P, P]),
    "erf_conv1d_dgrad_bn_backward_bits": (I, [P, P, P, P, P, I, I, I, I, I, I, I, P, P, P, P, P, I, P, P, P, P, P]),
    "erf_conv1d_pack_many": (I, [P, I, I, I, P]),
    "erf_conv1d_pair": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "erf_nb1d_block_infer": (I, [P, P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "erf_downsampler_packed_bytes": (c_size_t, [I, I, I]),
    "erf_downsampler_pack": (I, [P, P, I, I, I, P]),
    "erf_downsampler": (I, [P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "erf_stem_downsampler": (I, [P, I, P, P, P, P, I, I, I, I, I, P]),
    "erf_upsampler_packed_bytes": (c_size_t, [I, I, I]),
    "erf_upsampler_pack": (I, [P, P, I, I, I, P]),
    "erf_upsampler_convT3x3s2": (I, [P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "erf_output_conv_packed_bytes": (c_size_t, [I, I]),
    "erf_output_conv_pack": (I, [P, P, I, I, P]),
    "erf_output_convT2x2s2": (I, [P, P, P, P, I, P, I, I, I, I, I, I, P]),
    "erf_pointwise_packed_bytes": (c_size_t, [I, I, I]),
    "erf_pointwise_pack": (I, [P, P, I, I, I, P]),
    "erf_pointwise_conv": (I, [P, P, P, P, I, I, I, I, I, I, I, P]),
    "erf_nchw_to_nhwc": (I, [P, I, P, I, I, I, I, I, P]),
    "erf_nhwc_to_nchw": (I, [P, I, P, I, I, I, I, I, P]),
    "erf_argmax_c": (I, [P, I, P, I, I, I, LL, P]),
    "erf_iou_hist_accumulate": (I, [P, I, P, I, LL, I, P, P]),
    "erf_iou_hist_accumulate_relabel": (I, [P, I, P, I, LL, I, I, I, P, P]),
    "erf_labels_from_u8": (I, [P, P, LL, I, I, P]),
    "erf_iou_planes_workspace_bytes": (c_size_t, [I]),
    "erf_iou_planes_accumulate": (I, [P, I, P, I, I, I, LL, I, P, P, P]),
    # ---- training
    "erf_bn_workspace_bytes": (c_size_t, [I]),
    "erf_bn_stats_finalize": (I, [P, I, LL, I, P, P, P, P, F, F, P, P, P, P, P, P]),
    "erf_bn_apply": (I, [P, P, P, P, P, P, P, I, I, LL, I, I, P]),
    "erf_bn_apply_gates": (I, [P, P, P, P, P, P, P, I, I, LL, I, I, P, P]),
    "erf_bn_backward": (I, [P, P, P, P, P, P, P, I, P, P, P, P, I, I, LL, I, P, P]),
    "erf_relu_backward": (I, [P, P, P, I, LL, P]),
    "erf_channel_sum": (I, [P, I, LL, I, P, P, P]),
    "erf_dropout2d_mask": (I, [P, I, F, ULL, P]),
    "erf_dropout2d_mask_many": (I, [P, I, P, P, P, P]),
    "erf_maxpool2x2_backward": (I, [P, P, P, I, I, I, I, I, I, I, I, P]),
    "erf_ce_workspace_bytes": (c_size_t, []),
    "erf_weighted_ce_fwd": (I, [P, I, P, P, I, I, LL, P, P, P]),
    "erf_weighted_ce_bwd": (I, [P, I, P, P, I, I, LL, P, P, P]),
    "erf_adam_fused_arena": (I, [P, P, P, P, LL, D, D, D, D, D, I, D, P]),
    "erf_adam_advance": (I, [P, P, P, D, D, P]),
    "erf_adam_fused_arena_hyper": (I, [P, P, P, P, LL, P, D, D, D, D, D, P]),
    "erf_dropout2d_mask_many_counter": (I, [P, I, P, P, P, P]),
    "erf_sgd_fused_arena": (I, [P, P, P, LL, D, D, D, I, D, P]),
    "erf_sgd_fused_arena_hyper": (I, [P, P, P, LL, P, D, I, D, P]),
    "erf_conv1d_dgrad": (I, [P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "erf_downsampler_dgrad_packed_bytes": (c_size_t, [I, I, I]),
    "erf_downsampler_dgrad_pack": (I, [P, P, I, I, I, P]),
    "erf_downsampler_dgrad": (I, [P, P, P, I, I, I, I, I, I, P]),
    "erf_upsampler_dgrad_packed_bytes": (c_size_t, [I, I, I]),
    "erf_upsampler_dgrad_pack": (I, [P, P, I, I, I, P]),
    "erf_upsampler_dgrad": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "erf_output_conv_patch_channels": (I, [I]),
    "erf_output_conv_patches": (I, [P, I, P, I, I, I, I, I, P]),
    "erf_output_conv_dgrad_packed_bytes": (c_size_t, [I, I]),
    "erf_output_conv_dgrad_pack": (I, [P, P, I, I, P]),
    "erf_output_conv_dgrad": (I, [P, P, P, P, I, I, I, I, I, P]),
    "erf_wgrad_workspace_bytes": (c_size_t, [I, I, I, LL]),
    "erf_conv1d_wgrad": (I, [P, P, P, I, I, I, I, I, I, I, P, c_size_t, P]),
    "erf_downsampler_wgrad": (I, [P, P, P, I, I, I, I, I, I, P, c_size_t, P]),
    "erf_image_to_nhwc_padded": (I, [P, I, P, I, I, I, I, P]),
    "erf_stem_wgrad": (I, [P, P, P, I, I, I, I, P, c_size_t, P]),
    "erf_upsampler_wgrad": (I, [P, P, P, I, I, I, I, I, I, P, c_size_t, P]),
    "erf_output_conv_wgrad": (I, [P, P, P, I, I, I, I, I, P, c_size_t, P]),
    "erf_pack_desc_bytes": (c_size_t, []),
    "erf_pack_record_begin": (I, [P, I]),
    "erf_pack_record_end": (I, [P, P]),
    "erf_pack_replay": (I, [P, I, ctypes.c_longlong, I, P]),
    "erf_conv3x3_packed_bytes": (c_size_t, [I, I, I]),
    "erf_conv3x3_pack": (I, [P, P, I, I, I, I, P]),
    "erf_conv3x3": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "erf_maxpool2x2_affine": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "erf_convT2x2s2_packed_bytes": (c_size_t, [I, I, I]),
    "erf_convT2x2s2_pack": (I, [P, P, I, I, I, P]),
    "erf_convT2x2s2": (I, [P, P, P, P, I, I, I, I, I, I, I, P]),
    "erf_spatial_mean": (I, [P, I, P, I, I, I, I, P]),
    "erf_concat_channels": (I, [P, I, P, I, P, I, I, I, I, P]),
    "erf_image_to_nhwc_channels": (I, [P, I, P, I, I, I, I, I, P]),
    "erf_pointwise_grad_channels": (I, [I]),
    "erf_pointwise_grad_nhwc": (I, [P, I, P, I, I, I, I, I, P]),
    "erf_pointwise_dgrad_packed_bytes": (c_size_t, [I, I, I]),
    "erf_pointwise_dgrad_pack": (I, [P, P, I, I, I, P]),
    "erf_pointwise_dgrad": (I, [P, P, P, I, I, I, I, I, I, P]),
    "erf_pointwise_wgrad": (I, [P, P, P, I, I, I, I, I, I, P, c_size_t, P]),
    "erf_probe_mfma_flop": (LL, [I, I, I]),
    "erf_probe_mfma": (I, [I, I, I, P, P]),
    "erf_probe_copy": (I, [P, P, LL, I, P]),
    "erf_static_scratch_max_channels": (I, []),
    "erf_bn_stats_finalize_static": (I, [P, I, LL, I, P, P, P, P, F, F, P, P, P, P, P]),
}
# entry points whose int return value is data, not a status
_PLAIN_INT = {"erf_version", "erf_output_conv_patch_channels", "erf_pointwise_grad_channels",
              "erf_static_scratch_max_channels"}

_fn = {}
for _name, (_res, _args) in SIGNATURES.items():
    _f = getattr(_lib, _name)          # AttributeError here = header and library out of sync
    _f.restype = _res
    _f.argtypes = _args
    _fn[_name] = _f


def call(name: str, *args):
    """Invoke a status-returning entry point; raise ErfError on a non-zero status."""
    rc = _fn[name](*args)
    if rc != 0:
        raise ErfError(f"{name} failed: {_ERRORS.get(rc, rc)}")


def query(name: str, *args):
    """Invoke a size/version query."""
    return _fn[name](*args)


_SIZES = {}


def size_query(name: str, *args):
    """query() for the pure size functions (workspace / packed-buffer bytes, padded channel counts): the answer depends on
    the arguments only, so it is remembered -- a training step asks several hundred of them."""
    key = (name,) + args
    v = _SIZES.get(key)
    if v is None:
        v = _SIZES[key] = _fn[name](*args)
    return v
