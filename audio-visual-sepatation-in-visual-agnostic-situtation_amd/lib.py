"""ctypes binding of libavsep_gfx950.so (the C ABI declared in include/avsep.h).

There is deliberately no fallback: if the shared library is missing or a call
returns an error, this raises.  PyTorch is used only for device memory and
streams; every pointer handed over is ``tensor.data_ptr()``.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libavsep_gfx950.so")

ACT_NONE, ACT_RELU, ACT_LRELU02, ACT_SIGMOID, ACT_TANH, ACT_SOFTMAX2 = range(6)
ACT_BY_NAME = {"no": ACT_NONE, "relu": ACT_RELU, "sigmoid": ACT_SIGMOID, "tanh": ACT_TANH,
               "softmax": ACT_SOFTMAX2}
LOSS_BY_NAME = {"bce": 0, "l1": 1, "l2": 2}
# avsep_conv_desc.algo: kernel families a call must not use (include/avsep.h AVSEP_ALGO_NO_*)
ALGO_NO = {"winograd": 1, "winograd_wgrad": 2, "flat": 4, "misc_patch": 8, "bf16_kernels": 16, "smallci_wgrad": 32, "winograd4": 64}


class AvsepError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                "N Cin H W Cout Ho Wo KH KW stride pad dil C0 act0 act1 up2x prec plan_n xfmt yfmt dyfmt dxfmt algo tune".split()] + \
               [(n, C.c_void_p) for n in "x0 x1 scale0 shift0 scale1 shift1".split()]


class ActBwd(C.Structure):
    """avsep_act_bwd (include/avsep.h): the operands of avsep_affine_act_bwd for avsep_conv2d_dgrad_act."""
    _fields_ = [(n, C.c_void_p) for n in "y scale shift residual res_scale res_shift dz2 add mean invstd bstats".split()] + \
               [("act", C.c_int32)]


class CatDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in "N C0 C1 H W bcast0 bcast1".split()] + \
               [(n, C.c_void_p) for n in "x0 x1 scale0 shift0 scale1 shift1".split()]


_P, _I, _F, _D, _Z = C.c_void_p, C.c_int32, C.c_float, C.c_double, C.c_size_t
_CD, _KD = C.POINTER(ConvDesc), C.POINTER(CatDesc)

# name -> (restype, argtypes); one entry per prototype in include/avsep.h
SIGNATURES = {
    "avsep_version": (C.c_int, []),
    "avsep_arch": (C.c_char_p, []),
    "avsep_strerror": (C.c_char_p, [C.c_int]),
    "avsep_conv_packed_floats": (_Z, [_CD, C.c_int]),
    "avsep_conv_pack_weights": (C.c_int, [_CD, _P, _P, C.c_int, _P]),
    "avsep_conv_io_formats": (C.c_int, [_CD, _I, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "avsep_conv2d_fwd_workspace_bytes": (_Z, [_CD]),
    "avsep_conv2d_fwd": (C.c_int, [_CD, _P, _P, _P, _P, _P, _Z, _P]),
    "avsep_conv2d_dgrad_workspace_bytes": (_Z, [_CD]),
    "avsep_conv2d_dgrad": (C.c_int, [_CD, _P, _P, _P, _P, _Z, _P]),
    "avsep_conv2d_dgrad_act_fused": (C.c_int32, [_CD]),
    "avsep_conv2d_dgrad_act": (C.c_int, [_CD, _P, _P, C.POINTER(ActBwd), _P, _P, _Z, _P]),
    "avsep_conv2d_head_applicable": (C.c_int32, [_CD]),
    "avsep_conv_kernel_name": (C.c_char_p, [_CD, _I, _I]),
    "avsep_conv_kernel_variant": (C.c_int, [_CD, _I, _I, C.c_char_p, _Z]),
    "avsep_space_to_depth2": (C.c_int, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "avsep_maxpool_bn_relu_bwd_stats": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "avsep_maxpool_bn_relu_bwd_apply": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "avsep_attmodel_infer_fwd": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "avsep_attmodel_infer_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "avsep_conv2d_dgrad_up2x_workspace_bytes": (_Z, [_CD]),
    "avsep_conv2d_dgrad_up2x": (C.c_int, [_CD, _P, _P, _P, _P, _P, _P, _P, _I, _P, _Z, _P]),
    "avsep_conv2d_wgrad_workspace_bytes": (_Z, [_CD]),
    "avsep_conv2d_wgrad": (C.c_int, [_CD, _P, _P, _P, _P, _Z, _P]),
    "avsep_channel_stats": (C.c_int, [_P, _I, _I, _I, _P, _P]),
    "avsep_bn_finalize": (C.c_int, [_P, _D, _P, _P, _P, _P, _F, _F, _I, _I, _P, _P, _P, _P, _P, _I, _P]),
    "avsep_bn_bwd_coeffs": (C.c_int, [_P, _D, _P, _P, _P, _I, _P, _P, _P, _P]),
    "avsep_bn_bwd_apply": (C.c_int, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "avsep_affine_act": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "avsep_affine_act_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "avsep_relu_up2x_fwd": (C.c_int, [_KD, _P, _P]),
    "avsep_relu_up2x_bwd": (C.c_int, [_KD, _P, _P, _P, _P, _P, _P, _I, _P]),
    "avsep_prepare": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "avsep_warp": (C.c_int, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "avsep_fusion_av_fwd": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "avsep_fusion_av_bwd": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _F,
                                      _P, _P, _P, _P]),
    "avsep_fusion_ao_fwd": (C.c_int, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "avsep_fusion_ao_bwd": (C.c_int, [_P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "avsep_fusion_n_av_fwd": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "avsep_fusion_n_av_bwd": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P]),
    "avsep_fusion_n_ao_fwd": (C.c_int, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "avsep_fusion_n_ao_bwd": (C.c_int, [_P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "avsep_mask_loss_fwd": (C.c_int, [_P, _P, _P, C.c_int64, _I, _I, _I, _I, _I, _P, _P, _P]),
    "avsep_mask_loss_bwd": (C.c_int, [_P, _P, _P, C.c_int64, _P, _I, _I, _I, _I, _I, _P, _P]),
    "avsep_stft_basis_floats": (_Z, [_I, _I]),
    "avsep_stft_basis": (C.c_int, [_I, _P, _P, _P]),
    "avsep_stft_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "avsep_stft_mag": (C.c_int, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _Z, _P]),
    "avsep_istft_workspace_bytes": (_Z, [_I, _I, _I]),
    "avsep_istft": (C.c_int, [_P, _P, _I, _I, _I, _I, _P, _P, _I, _P, _Z, _P]),
    "avsep_maxpool3x3s2_fwd": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "avsep_maxpool3x3s2_bwd": (C.c_int, [_P, _P, _I, _I, _I, _P, _P]),
    "avsep_temporal_mean_fwd": (C.c_int, [_P, _I, _I, _I, _P, _P]),
    "avsep_temporal_mean_bwd": (C.c_int, [_P, _I, _I, _I, _P, _P]),
    "avsep_sgd_momentum": (C.c_int, [_P, _P, _P, _Z, _F, _F, _F, _F, _I, _P]),
    "avsep_innerprod_fwd": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "avsep_innerprod_nosum": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "avsep_innerprod_pixelwise": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "avsep_innerprod_bwd": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "avsep_f32_to_b16": (C.c_int, [_P, _I, _I, _I, _P, _P]),
    "avsep_b16_to_f32": (C.c_int, [_P, _I, _I, _I, _P, _P]),
    "avsep_b16_affine_act": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "avsep_b16_affine_act_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    "avsep_b16_grid_pack": (C.c_int, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P]),
    "avsep_bss_corr": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "avsep_bss_solve_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "avsep_bss_solve": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P, _Z, _P, _P, _P]),
    "avsep_bss_project": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "avsep_grid_unpack": (C.c_int, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P]),
    "avsep_bn_bwd_apply_to_b16": (C.c_int, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "avsep_b16_bn_bwd_apply": (C.c_int, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "avsep_b16_relu_up2x_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "avsep_b16_relu_up2x_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P]),
    "avsep_b16_maxpool3x3s2_fwd": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "avsep_b16_maxpool_bn_relu_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _I, _P]),
    "avsep_b16_space_to_depth2": (C.c_int, [_P, _I, _I, _I, _I, _P, _P]),
    "avsep_sdr_sums": (C.c_int, [_P, _P, _I, _I, C.c_int64, C.c_int64, _P, _P]),
}

_lib = None


def load():
    """Load the shared library (raises AvsepError when it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AvsepError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU / PyTorch fallback for the HIP path.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def ptr(t):
    if t is None:
        return None
    assert t.is_contiguous(), "avsep kernels take dense tensors"
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    """Invoke an int-returning entry point on the current stream; raise on a non-zero status."""
    lib = load()
    rc = getattr(lib, name)(*args, stream())
    if rc != 0:
        raise AvsepError(f"{name} failed: {lib.avsep_strerror(rc).decode()} ({rc})")


def require_gpu(t):
    if not t.is_cuda:
        raise AvsepError("the HIP path needs tensors on an MI355X (cuda) device; there is no CPU fallback")
