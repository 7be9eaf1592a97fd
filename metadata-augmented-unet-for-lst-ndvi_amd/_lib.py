"""ctypes binding of ``libmau_hip.so`` (C ABI declared in ``include/mau_hip.h``).

The library is the product: there is NO fallback.  If the shared object is
missing or does not export every symbol of the header, importing this module
raises; if a call returns non-zero, ``MauError`` carries ``mau_last_error()``.
"""
from __future__ import annotations

import ctypes as C
import os

# torch must load ITS HIP runtime (libamdhip64) first: libmau_hip.so then binds to that same runtime
# instance by SONAME.  Loading libmau_hip.so first would bring in a second runtime instance that owns
# no device ("no ROCm-capable device is detected" on the first launch).
import torch  # noqa: F401  (import order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MAU_LIB") or os.path.join(_HERE, "libmau_hip.so")     # MAU_LIB: A/B builds of the same ABI

MAU_F32 = 0
MAU_BF16 = 1
MAU_F16 = 2

_p, _i, _i64, _f, _d, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes); mirrors include/mau_hip.h one to one
PROTOTYPES = {
    "mau_abi_version": (_i, []),
    "mau_last_error": (C.c_char_p, []),
    "mau_device_check": (_i, []),
    "mau_nchw_to_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mau_pack_tile_onehot": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "mau_flip_rows": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "mau_nhwc_to_nchw": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mau_conv3x3_kc": (_i, [_i]),
    "mau_conv3x3_packed_elems": (_sz, [_i, _i, _i]),
    "mau_conv3x3_pack_weights": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "mau_conv3x3_pack_desc_bytes": (_sz, []),
    "mau_conv3x3_pack_desc_fill": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _p]),
    "mau_conv3x3_pack_weights_multi": (_i, [_p, _i, _i, _i, _p]),
    "mau_conv3x3_num_pixel_tiles": (_i, [_i, _i, _i, _i, _i]),
    "mau_conv3x3_variant": (_i, [_i, _i, _i, _i, _i, _p, _p, _p]),
    "mau_conv3x3_first_max_channels": (_i, []),
    "mau_conv3x3_first_rows": (_i, [_i, _i, _i]),
    "mau_conv3x3_first_wgrad_ws_elems": (_sz, [_i, _i, _i, _i]),
    "mau_conv3x3_first_wgrad": (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mau_conv3x3_first_fwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _i, _i, _p, _p, _i, _i, _i, _i, _p]),
    "mau_conv3x3_fwd": (_i, [_p, _i, _i, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _p, _i, _i, _i, _i, _p]),
    "mau_conv3x3_fwd_pool": (_i, [_p, _i, _i, _p, _p, _p, _p, _p, _i, _i, _p, _i, _i, _i, _i, _i, _p]),
    "mau_conv3x3_fwd2": (_i, [_p, _i, _i, _p, _i, _i, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _p, _i, _i, _i, _i, _p]),
    "mau_conv3x3_wgrad_splits": (_i, [_i, _i, _i, _i, _i, _i]),
    "mau_conv3x3_wgrad_acc_elems": (_sz, [_i, _i, _i, _i, _i, _i]),
    "mau_conv3x3_wgrad": (_i, [_p, _i, _i, _p, _p, _i, _p, _i, _i, _p, _i, _i, _i, _i, _p]),
    "mau_conv3x3_wgrad2": (_i, [_p, _i, _i, _p, _i, _i, _p, _p, _i, _p, _i, _i, _p, _i, _i, _i, _i, _p]),
    "mau_conv3x3_unpack_wgrad": (_i, [_p, _i, _p, _i, _i, _p]),
    "mau_reduce_rows_ws_elems": (_sz, [_i, _i]),
    "mau_reduce_tickets_elems": (_i, []),
    "mau_reduce_rows_f64": (_i, [_p, _i, _i, _i, _p, _p, _p, _p]),
    "mau_reduce_rows_f64_f32": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _d, _p]),
    "mau_reduce_rows_f32": (_i, [_p, _i, _i, _i, _p, _p, _p, _p]),
    "mau_bn_stats_sums_f64": (_i, [_p, _i, _i, _p, _p, _p, _d, _p]),
    "mau_bn_finalize_train": (_i, [_p, _d, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _i, _p]),
    "mau_bn_stats_ws_elems": (_sz, [_i, _i]),
    "mau_bn_stats_finalize_train": (_i, [_p, _i, _d, _p, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _i, _p]),
    "mau_bn_coeffs_eval": (_i, [_p, _p, _p, _p, _f, _p, _p, _p, _p, _i, _p]),
    "mau_bn_relu_apply": (_i, [_p, _i, _p, _p, _p, _i, _i, _i64, _i, _p]),
    "mau_bn_relu_apply_pool": (_i, [_p, _i, _p, _p, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p]),
    "mau_bn_relu_bwd_reduce": (_i, [_p, _i, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i64, _i, _p]),
    "mau_bn_relu_bwd_apply": (_i, [_p, _i, _p, _i, _p, _p, _p, _p, _p, _d, _p, _i, _i, _i64, _i, _p]),
    "mau_bn_bwd_rows": (_i, [_i64]),
    "mau_pool_bn_bwd_reduce": (_i, [_p, _i, _p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mau_pool_bn_bwd_apply": (_i, [_p, _i, _p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _d, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mau_head_bn_max_channels": (_i, []),
    "mau_head_bn_fwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mau_head_bn_bwd_reduce": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mau_head_bn_bwd_apply": (_i, [_p, _i, _p, _p, _p, _p, _p, _d, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "mau_resize_bilinear_bn_fwd": (_i, [_p, _i, _i, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "mau_maxpool2x2_fwd": (_i, [_p, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mau_maxpool2x2_bwd": (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mau_maxpool2x2_bwd_add": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mau_resize_bilinear_fwd": (_i, [_p, _i, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "mau_resize_bilinear_bwd": (_i, [_p, _i, _i, _i, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mau_copy_channels": (_i, [_p, _i, _p, _i, _i, _i, _i, _i64, _i, _p]),
    "mau_bcast_fill": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "mau_bcast_bwd_ws_elems": (_sz, [_i, _i, _i]),
    "mau_bcast_bwd": (_i, [_p, _i, _i, _p, _p, _i, _i, _i, _i, _p]),
    "mau_head_fwd": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mau_head_bwd": (_i, [_p, _i, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    "mau_head_bwd_rows": (_i, [_i, _i]),
    "mau_head_bwd_rowlen": (_i, [_i, _i]),
    "mau_meta_mlp_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "mau_meta_mlp_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "mau_lstm_max_hidden": (_i, []),
    "mau_lstm_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "mau_lstm_bwd_ws_elems": (_sz, [_i, _i, _i]),
    "mau_lstm_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "mau_linear_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "mau_linear_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "mau_adamw_pack_desc_bytes": (_sz, []),
    "mau_adamw_pack_desc_fill": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "mau_adamw_pack_step": (_i, [_p, _i, _i, _i, _p, _f, _f, _f, _f, _f, _p]),
    "mau_sum_tensors_max": (_i, []),
    "mau_sum_tensors": (_i, [_p, _p, _i, _p, _i, _i, _i64, _i, _p]),
    "mau_emb_fold_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "mau_emb_fold_ws_elems": (_sz, [_i, _i, _i]),
    "mau_emb_fold_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "mau_mse_blocks": (_i, [_i64]),
    "mau_l1_gradient_blocks": (_i, [_i64]),
    "mau_l1_gradient_loss": (_i, [_p, _p, _p, _p, _p, _f, _f, _i, _i, _i, _i, _p]),
    "mau_ssim_ws_elems": (_sz, [_i, _i, _i, _i]),
    "mau_ssim_loss": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "mau_mse_fwd_bwd": (_i, [_p, _p, _p, _p, _p, _i64, _p]),
}


class MauError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the MI355X-native HIP library has not been built "
            f"(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C {os.path.join(_HERE, 'csrc')}`). "
            "There is no CPU or PyTorch fallback for this path.")
    lib = C.CDLL(LIB_PATH)
    missing = [n for n in PROTOTYPES if not hasattr(lib, n)]
    if missing:
        raise ImportError(f"{LIB_PATH} does not export {missing}; rebuild it")
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(status: int, what: str = ""):
    if status != 0:
        msg = lib.mau_last_error()
        raise MauError(f"libmau_hip {what} failed (status {status}): {msg.decode() if msg else '?'}")


def conv3x3_variant(dtype: int, N: int, H: int, W: int, Cout: int, Cin: int = 0):
    """(tile rows, waves per workgroup, output channels per workgroup, K groups per workgroup) of the convolution variant that runs
    such a layer; ``Cin = 0``: input channels unknown (K groups reported as 1)."""
    th, nw, bn, kg = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    check(lib.mau_conv3x3_variant(dtype, N, H, W, Cin, Cout, C.byref(th), C.byref(nw), C.byref(bn), C.byref(kg)), "mau_conv3x3_variant")
    return th.value, nw.value, bn.value, kg.value


def call(name: str, *args):
    """Call an int-status entry point and raise on error."""
    check(getattr(lib, name)(*args), name)
