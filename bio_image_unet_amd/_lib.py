"""ctypes binding of ``libbiu_hip.so`` (the C ABI declared in ``include/biu.h``).

The library is built in-tree by ``__graft_entry__.build()`` (``hipcc --offload-arch=gfx950``).  There is no
fallback: if the shared object is missing, importing this module raises, and every op of the package fails.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- FIRST: the library must bind to the HIP runtime torch ships (loaded before torch, /opt/rocm's copy comes in as a
#                 second runtime and every call of ours then fails with 'no ROCm-capable device is detected')

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BIU_LIB_PATH") or os.path.join(_HERE, "libbiu_hip.so")     # override: A/B builds of tools/build_variant.sh

BIU_F32, BIU_BF16 = 0, 1
BN_MAX_PARTIALS = 1024


class BiuError(RuntimeError):
    pass


class biu_act(C.Structure):
    _fields_ = [("p", C.c_void_p), ("n", C.c_int32), ("d", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("c", C.c_int32), ("pitch", C.c_int32)]


class biu_pack_job(C.Structure):
    _fields_ = [("w", C.c_void_p), ("packed", C.c_void_p), ("transposed", C.c_int32), ("kind", C.c_int32), ("cin", C.c_int32),
                ("cout", C.c_int32), ("kd", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32), ("reserved", C.c_int32)]


class biu_xform(C.Structure):
    _fields_ = [("scale", C.c_void_p), ("shift", C.c_void_p), ("slope", C.c_void_p)]


_P = C.c_void_p
_A = C.POINTER(biu_act)
_X = C.POINTER(biu_xform)
_I = C.c_int
_F = C.c_float
_D = C.c_double
_Z = C.c_size_t

# name -> (restype, argtypes); mirrors include/biu.h one to one
SIGNATURES = {
    "biu_last_error": (C.c_char_p, []),
    "biu_version": (_I, []),
    "biu_set_fp32_products": (_I, [_I]),
    "biu_conv_packed_bytes": (_Z, [_I, _I, _I, _I, _I, _I, _I, _I]),
    "biu_conv_pack": (_I, [_I, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "biu_pack_batch": (_I, [_P, _I, _I, _P]),
    "biu_conv_split_workspace": (_Z, [_I, _A, _A, _I, _I, _I, _I, _I]),
    "biu_conv_fwd": (_I, [_A, _X, _P, _P, _P, _I, _I, _I, _I, _A, _P, _Z, _I, _P]),
    "biu_conv_fwd_stats_floats": (_Z, [_A, _I]),
    "biu_conv_fwd_stats": (_I, [_A, _X, _P, _P, _P, _I, _I, _I, _I, _A, _P, _Z, C.POINTER(C.c_int), _P, _Z, _I, _P]),
    "biu_conv_bwd_data": (_I, [_A, _P, _P, _I, _I, _I, _I, _A, _I, _P, _Z, _I, _P]),
    "biu_bwd_data_bnred_floats": (_Z, [_A, _I, _I]),
    "biu_conv_bwd_data_bnred": (_I, [_A, _P, _P, _I, _I, _I, _I, _A, _A, _P, _P, _P, _P, _P, _P, _Z, C.POINTER(C.c_int), _P, _Z, _I, _P]),
    "biu_convt_bwd_data_bnred": (_I, [_A, _P, _P, _I, _A, _A, _P, _P, _P, _P, _P, _P, _Z, C.POINTER(C.c_int), _I, _P]),
    "biu_conv_bwd_weight_workspace": (_Z, [_I, _I, _I, _I, _I, _I]),
    "biu_conv_bwd_weight": (_I, [_A, _X, _A, _I, _I, _I, _I, _P, _P, _P, _Z, _I, _P]),
    "biu_conv_bwd_weight_bn": (_I, [_A, _X, _A, _A, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _Z, _I, _P]),
    "biu_bn_stats": (_I, [_A, _P, C.POINTER(C.c_int), _I, _P]),
    "biu_bn_finalize": (_I, [_P, _I, _I, _D, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P]),
    "biu_bn_eval_affine": (_I, [_I, _P, _P, _P, _P, _F, _P, _P, _P]),
    "biu_xform_apply": (_I, [_A, _X, _A, _I, _P]),
    "biu_bn_bwd_reduce": (_I, [_A, _A, _P, _P, _P, _P, _P, _P, C.POINTER(C.c_int), _I, _P]),
    "biu_bn_bwd_finalize": (_I, [_P, _I, _I, _D, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "biu_bn_bwd_finalize_eval": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "biu_bn_bwd_apply": (_I, [_A, _A, _P, _P, _P, _P, _P, _P, _A, _I, _P]),
    "biu_maxpool_fwd": (_I, [_A, _X, _A, _I, _P]),
    "biu_maxpool_bwd": (_I, [_A, _X, _A, _A, _I, _I, _P]),
    "biu_maxpool_bwd_bnred": (_I, [_A, _X, _A, _A, _I, _P, _P, _P, _Z, C.POINTER(C.c_int), _I, _P]),
    "biu_nearest_down_fwd": (_I, [_A, _X, _A, _I, _P]),
    "biu_nearest_down_bwd": (_I, [_A, _A, _I, _I, _P]),
    "biu_nearest_up_fwd": (_I, [_A, _X, _A, _I, _P]),
    "biu_upconv_ok": (_I, [_A, _A, _I]),
    "biu_upconv_packed_bytes": (_Z, [_I, _I, _I, _I]),
    "biu_upconv_pack": (_I, [_I, _P, _I, _I, _I, _P, _P]),
    "biu_upconv_bwd_data": (_I, [_A, _P, _A, _I, _I, _P]),
    "biu_upconv_bwd_weight_workspace": (_Z, [_I, _I, _I]),
    "biu_upconv_bwd_weight_bn": (_I, [_A, _X, _A, _A, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _I, _P]),
    "biu_upconv_fwd_stats_floats": (_Z, [_A, _A]),
    "biu_upconv_fwd": (_I, [_A, _X, _P, _P, _A, _P, _Z, C.POINTER(C.c_int), _I, _P]),
    "biu_nearest_up_bwd": (_I, [_A, _A, _I, _I, _P]),
    "biu_bce_dice_blocks": (_I, [C.c_longlong]),
    "biu_bce_dice_fwd": (_I, [_P, _P, _I, C.c_longlong, _P, _P]),
    "biu_bce_dice_bwd": (_I, [_P, _P, _I, C.c_longlong, _P, _P, _I, _P]),
    "biu_pair_smooth_l1_blocks": (_I, [C.c_longlong]),
    "biu_pair_smooth_l1_fwd": (_I, [_P, _I, C.c_longlong, _P, _P]),
    "biu_pair_smooth_l1_bwd": (_I, [_P, _I, C.c_longlong, _P, _P, _I, _P]),
    "biu_seg_loss_finish": (_I, [_P, _I, _I, C.c_longlong, _P, _I, _F, _F, _F, _I, _F, _F, _F, _I, _F, _P, _P]),
    "biu_seg_loss_coef": (_I, [_P, _P, _I, C.c_longlong, _F, _F, _F, _I, _F, _F, _F, _I, _F, _P, _P, _P]),
    "biu_head_dlogits": (_I, [_P, _P, _P, _I, _I, _I, C.c_longlong, _P, _I, _I, _P]),
    "biu_trilinear_up_fwd": (_I, [_A, _X, _A, _I, _P]),
    "biu_trilinear_up_bwd": (_I, [_A, _A, _I, _I, _P]),
    "biu_xcorr_fwd": (_I, [_A, _X, _A, _X, _A, _I, _P]),
    "biu_xcorr_bwd": (_I, [_A, _X, _A, _X, _A, _A, _A, _I, _I, _P]),
    "biu_conv_cat_ok": (_I, [_A, _A, _A, _I, _I, _I, _I, _I]),
    "biu_conv_fwd_cat": (_I, [_A, _X, _A, _X, _P, _P, _P, _I, _I, _I, _I, _A, _P, _Z, C.POINTER(C.c_int), _P, _Z, _I, _P]),
    "biu_conv_bwd_data_cat": (_I, [_A, _P, _P, _I, _I, _I, _I, _A, _I, _A, _I, _P, _Z, _I, _P]),
    "biu_conv_bwd_weight_cat": (_I, [_A, _X, _A, _X, _A, _A, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _Z, _I, _P]),
    "biu_convt_packed_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "biu_convt_pack": (_I, [_I, _P, _I, _I, _I, _I, _P, _P]),
    "biu_foldt_ok": (_I, [_A, _A, _A, _I]),
    "biu_foldt_packed_bytes": (_Z, [_I, _I, _I, _I]),
    "biu_foldt_pack": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "biu_foldt_fwd_stats_floats": (_Z, [_A, _A]),
    "biu_foldt_fwd_form": (_I, [_A, _A, _A, _I]),
    "biu_foldt_fwd": (_I, [_A, _X, _A, _X, _P, _A, _P, _Z, C.POINTER(C.c_int), _I, _P]),
    "biu_foldt_bwd_data_bnred_floats": (_Z, [_A]),
    "biu_foldt_bwd_data": (_I, [_A, _P, _A, _I, _A, _I, _A, _P, _P, _P, _P, _P, _P, _Z, C.POINTER(C.c_int), _P, _Z, _I, _P]),
    "biu_foldt_bwd_weight_workspace": (_Z, [_I, _I, _I, _I]),
    "biu_foldt_bwd_weight_bn": (_I, [_A, _X, _A, _X, _A, _A, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _Z, _I, _P]),
    "biu_foldt_bwd_weight_bn_phase": (_I, [_A, _X, _A, _X, _A, _A, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _Z, _I, _I, _P]),
    "biu_convt_fwd": (_I, [_A, _X, _P, _P, _P, _I, _A, _I, _P]),
    "biu_convt_bwd_data": (_I, [_A, _P, _P, _I, _A, _I, _I, _P]),
    "biu_convt_bwd_weight_workspace": (_Z, [_I, _I, _I, _I]),
    "biu_convt_bwd_weight": (_I, [_A, _X, _A, _I, _P, _P, _P, _Z, _I, _P]),
    "biu_head_fwd": (_I, [_A, _X, _P, _P, _I, _I, _P, _P, _I, _P]),
    "biu_head_bwd_workspace": (_Z, [_I]),
    "biu_head_bwd": (_I, [_A, _X, _P, _I, _P, _A, _P, _P, _P, _Z, _I, _P]),
    "biu_head_bwd_bnred": (_I, [_A, _X, _P, _I, _P, _A, _P, _P, _P, _Z, _P, _P, _P, _Z, C.POINTER(C.c_int), _I, _P]),
    "biu_max_join_fwd": (_I, [_A, _X, _A, _X, _A, _I, _P]),
    "biu_max_join_bwd": (_I, [_A, _X, _A, _X, _A, _A, _A, _I, _I, _P]),
    "biu_act_add": (_I, [_A, _A, _I, _I, _P]),
    "biu_add_relu_fwd": (_I, [_A, _X, _A, _X, _A, _I, _P]),
    "biu_add_relu_bwd": (_I, [_A, _A, _A, _A, _I, _I, _P]),
    "biu_gate_fwd": (_I, [_A, _X, _A, _X, _A, _I, _P]),
    "biu_gate_bwd": (_I, [_A, _X, _A, _X, _A, _A, _I, _A, _I, _P]),
    "biu_from_nchw": (_I, [_P, _A, _I, _P]),
    "biu_from_nchw_u8": (_I, [_P, _F, _A, _I, _P]),
    "biu_u8_to_f32": (_I, [_P, _F, _P, C.c_longlong, _P]),
    "biu_quantize_u8": (_I, [_P, _F, _P, C.c_longlong, _P]),
    "biu_stitch_add": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "biu_stitch_finish": (_I, [_P, _P, _I, _I, C.c_longlong, _P, _I, _P]),
    "biu_to_nchw": (_I, [_A, _X, _P, _I, _P]),
    "biu_adam_step": (_I, [_I, _P, _P, _P, _P, _P, _F, _F, _F, _F, _I, _F, _P]),
    "biu_adam_set_hyper": (_I, [_P, _F, _F, _F, _F, _I, _F, _P]),
    "biu_adam_step_hyper": (_I, [_I, _P, _P, _P, _P, _P, _P, _P]),
    "biu_grad_clip_scratch_floats": (_Z, [_I]),
    "biu_grad_clip": (_I, [_I, _P, _P, _F, _P, _Z, _P, _P]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP kernels are not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` from the repository root (needs hipcc). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here == header / library out of sync
        fn.restype = res
        fn.argtypes = args
    return lib


class _Lib:
    """Thin proxy over the CDLL: plain attribute access in normal operation, optional per-call HIP-event timing.

    ``prof``  : list -> every call appends (api name, current label, start event, end event)
    ``watch`` : (api name, label) -> only that call is timed, into ``watched`` (used inside bench.py's timed region)
    Events are recorded on torch's current stream, which is the stream every kernel is launched on.
    """

    def __init__(self, cdll):
        object.__setattr__(self, "_c", cdll)
        object.__setattr__(self, "prof", None)
        object.__setattr__(self, "watch", None)
        object.__setattr__(self, "watched", [])
        object.__setattr__(self, "label", "")

    def __setattr__(self, k, v):
        object.__setattr__(self, k, v)

    def __getattr__(self, name):
        fn = getattr(self._c, name)
        if self.prof is None and self.watch is None:
            return fn
        if self.prof is None and self.watch != (name, self.label):
            return fn
        import torch

        def timed(*args):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args)
            e1.record()
            (self.prof if self.prof is not None else self.watched).append((name, self.label, e0, e1))
            return rc

        return timed


lib = _Lib(_load())


def check(status: int, what: str = ""):
    if status != 0:
        msg = lib.biu_last_error()
        raise BiuError(f"{what}: biu status {status}: {msg.decode() if msg else ''}")
