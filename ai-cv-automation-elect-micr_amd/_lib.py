"""ctypes binding of libemdenoise.so (the C ABI declared in include/emdenoise.h).

There is no CPU fallback: if the HIP library is missing, or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EMD_LIB_PATH") or os.path.join(PKG_DIR, "libemdenoise.so")  # override: A/B kernel builds

EMD_OK = 0
EMD_K_SYMMETRIC = 1

_c_float_p = C.c_void_p  # device pointers travel as integers

# name -> (restype, argtypes): must list every symbol include/emdenoise.h declares
SIGNATURES = {
    "emd_version": (C.c_int, []),
    "emd_last_error": (C.c_char_p, []),
    "emd_crc32c": (C.c_uint32, [C.c_void_p, C.c_size_t, C.c_uint32]),
    "emd_kernel_params_count": (C.c_size_t, [C.c_int, C.c_int]),
    "emd_kernel_denoise_f32": (C.c_int, [_c_float_p, _c_float_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         _c_float_p, C.c_uint, C.c_void_p]),
    "emd_packed_weight_elems": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "emd_pack_weights_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    # x ldx whi wlo scale1 shift1 scale2 shift2 res ldres y ldy B H W Cin Cout stride act precision stream
    "emd_conv1x1_f32": (C.c_int, [_c_float_p, C.c_int, C.c_void_p, C.c_void_p, _c_float_p, _c_float_p, _c_float_p,
                                  _c_float_p, _c_float_p, C.c_int, _c_float_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "emd_sep3x3_fused_supported": (C.c_int, [C.c_int] * 6),
    # x ldx dw whi wlo scale1 shift1 scale2 shift2 res ldres y ldy B H W Cin Cout act precision stream
    "emd_sep3x3_fused_f32": (C.c_int, [_c_float_p, C.c_int, _c_float_p, C.c_void_p, C.c_void_p, _c_float_p, _c_float_p,
                                       _c_float_p, _c_float_p, _c_float_p, C.c_int, _c_float_p, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    # x ldx whi wlo scale1 shift1 scale2 shift2 res ldres y ldy B H W Cin Cout stride rate act precision stream
    "emd_conv3x3_f32": (C.c_int, [_c_float_p, C.c_int, C.c_void_p, C.c_void_p, _c_float_p, _c_float_p, _c_float_p,
                                  _c_float_p, _c_float_p, C.c_int, _c_float_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    # x ldx y ldy B H W C stream
    "emd_avgpool2x2_f32": (C.c_int, [_c_float_p, C.c_int, _c_float_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p]),
    "emd_deconv_phase_taps": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    # x ldx whi[4] wlo[4] scale1 shift1 y ldy B H W Cin Cout act precision stream
    "emd_deconv3x3s2_f32": (C.c_int, [_c_float_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _c_float_p,
                                      _c_float_p, _c_float_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_void_p]),
    # x ldx w y ldy B H W C stride rate stream
    "emd_dw3x3_f32": (C.c_int, [_c_float_p, C.c_int, _c_float_p, _c_float_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_int, C.c_void_p]),
    # x w9 a shift y ldy B H W Cout stride act stream
    "emd_cin1_f32": (C.c_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, _c_float_p, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    # x ldx w scale shift y B H W Cin act stream
    "emd_conv3x3_cout1_f32": (C.c_int, [_c_float_p, C.c_int, _c_float_p, C.c_float, C.c_float, _c_float_p, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    # x ldx scale shift res ldres y ldy npix C act stream
    "emd_affine_act_f32": (C.c_int, [_c_float_p, C.c_int, _c_float_p, _c_float_p, _c_float_p, C.c_int, _c_float_p,
                                     C.c_int, C.c_long, C.c_int, C.c_int, C.c_void_p]),
    "emd_bn_stats_workspace_bytes": (C.c_size_t, [C.c_long, C.c_int]),
    "emd_bn_stats_f32": (C.c_int, [_c_float_p, C.c_int, C.c_long, C.c_int, _c_float_p, _c_float_p, C.c_void_p,
                                   C.c_void_p]),
    "emd_bn_fold_f32": (C.c_int, [_c_float_p, _c_float_p, _c_float_p, _c_float_p, C.c_float, _c_float_p, _c_float_p,
                                  C.c_int, C.c_void_p]),
    # x ldx y ldy B Hi Wi Ho Wo C stream
    "emd_resize_bilinear_f32": (C.c_int, [_c_float_p, C.c_int, _c_float_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_void_p]),
    # x ldx scale shift y ldy npix C act stream
    "emd_affine_relu6_f32": (C.c_int, [_c_float_p, C.c_int, _c_float_p, _c_float_p, _c_float_p, C.c_int, C.c_long,
                                       C.c_int, C.c_int, C.c_void_p]),
}

_lib = None


class EmdError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EmdError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python ai-cv-automation-elect-micr_amd/build.py, or __graft_entry__.build()). "
            "There is no CPU fallback for this path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != EMD_OK:
        msg = load().emd_last_error()
        raise EmdError(f"{what or 'libemdenoise'} failed (code {rc}): {msg.decode() if msg else ''}")


def stream_ptr(stream=None):
    """hipStream_t of a torch stream (default: torch's current stream) as an integer."""
    import torch

    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def ptr(t):
    """Device pointer of a torch CUDA tensor (must be float32 and dense in its last dim)."""
    return C.c_void_p(t.data_ptr())
