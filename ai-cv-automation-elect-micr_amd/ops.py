"""Thin torch-tensor wrappers over the graph-D entry points of libemdenoise.so.

An activation is an ``Act``: a view [B,H,W,C] into a torch CUDA buffer whose pixel stride ``ld`` may be
wider than C (a channel slice of a concat buffer, denoiser.py:203/:353/:365).  Every function here
only checks shapes and forwards pointers; the arithmetic is in the HIP library.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

PREC_BF16 = 1
PREC_BF16X3 = 3
ACT_NONE, ACT_RELU6, ACT_RELU, ACT_RELU6_CLIP01, ACT_LEAKY = 0, 1, 2, 3, 4


def _act(a):
    """bool -> relu6 / none (graph D); int EMD_ACT_* passes through."""
    return int(a) if not isinstance(a, bool) else (ACT_RELU6 if a else ACT_NONE)


class Act:
    """[B,H,W,C] float32 view: channels [c0, c0+C) of ``buf`` [B,H,W,ld]."""

    __slots__ = ("buf", "B", "H", "W", "C", "ld", "c0")

    def __init__(self, buf, C_=None, c0=0):
        assert buf.dim() == 4 and buf.is_contiguous() and str(buf.dtype) == "torch.float32" and buf.is_cuda
        self.buf = buf
        self.B, self.H, self.W, self.ld = buf.shape
        self.C = self.ld - c0 if C_ is None else C_
        self.c0 = c0
        assert 0 <= c0 and c0 + self.C <= self.ld

    @classmethod
    def empty(cls, B, H, W, Cc, device):
        import torch

        return cls(torch.empty((B, H, W, Cc), dtype=torch.float32, device=device))

    def slice(self, c0, Cc):
        return Act(self.buf, Cc, self.c0 + c0)

    def images(self, b0, b1):
        """Images [b0, b1) of the batch (a view of the same memory)."""
        return Act(self.buf[b0:b1], self.C, self.c0)

    @property
    def ptr(self):
        return C.c_void_p(self.buf.data_ptr() + 4 * self.c0)

    def torch(self):
        return self.buf[..., self.c0: self.c0 + self.C]


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class PackedWeights:
    """bf16 hi/lo planes of one conv's weights on the device (emd_pack_weights_bf16 layout)."""

    def __init__(self, w_tf: np.ndarray, cout_major: bool, device):
        import torch

        lib = _lib.load()
        w = np.ascontiguousarray(w_tf, dtype=np.float32)
        if cout_major:
            taps, cout, cin = w.shape
        else:
            taps, cin, cout = w.shape
        n = lib.emd_packed_weight_elems(taps, cin, cout)
        hi = np.empty(n, np.uint16)
        lo = np.empty(n, np.uint16)
        _lib.check(lib.emd_pack_weights_bf16(w.ctypes.data, taps, cin, cout, 1 if cout_major else 0,
                                             hi.ctypes.data, lo.ctypes.data), "emd_pack_weights_bf16")
        self.taps, self.cin, self.cout = taps, cin, cout
        # uint16 planes travel as int16 torch tensors (same bits); one allocation, lo right behind hi (128-byte aligned):
        # the persistent split32 GEMM addresses both planes from one base
        npad = -(-n // 64) * 64
        both = torch.empty(2 * npad, dtype=torch.int16, device=device)
        both[:n].copy_(torch.from_numpy(hi.view(np.int16)))
        both[npad:npad + n].copy_(torch.from_numpy(lo.view(np.int16)))
        self._planes = both
        self.hi = both[:n]
        self.lo = both[npad:npad + n]


def conv1x1(x: Act, w: PackedWeights, scale1, shift1, out: Act, stride=1, act=True, scale2=None, shift2=None,
            res: Act | None = None, precision=PREC_BF16X3, stream=None):
    lib = _lib.load()
    Ho, Wo = -(-x.H // stride), -(-x.W // stride)
    assert (out.B, out.H, out.W, out.C) == (x.B, Ho, Wo, w.cout) and w.cin == x.C and w.taps == 1
    if res is not None:
        assert (res.B, res.H, res.W, res.C) == (out.B, out.H, out.W, out.C)
    rc = lib.emd_conv1x1_f32(x.ptr, x.ld, _p(w.hi), _p(w.lo), _p(scale1), _p(shift1), _p(scale2), _p(shift2),
                             res.ptr if res is not None else C.c_void_p(0), res.ld if res is not None else 0,
                             out.ptr, out.ld, x.B, x.H, x.W, x.C, w.cout, stride, _act(act), precision,
                             _lib.stream_ptr(stream))
    _lib.check(rc, "emd_conv1x1_f32")
    return out


def conv3x3(x: Act, w: PackedWeights, scale1, shift1, out: Act, stride=1, rate=1, act=True, res: Act | None = None,
            precision=PREC_BF16X3, stream=None, scale2=None, shift2=None):
    """Dense 3x3 conv (9-tap implicit GEMM), TF SAME, optional dilation."""
    lib = _lib.load()
    Ho, Wo = -(-x.H // stride), -(-x.W // stride)
    assert (out.B, out.H, out.W, out.C) == (x.B, Ho, Wo, w.cout) and w.cin == x.C and w.taps == 9
    rc = lib.emd_conv3x3_f32(x.ptr, x.ld, _p(w.hi), _p(w.lo), _p(scale1), _p(shift1), _p(scale2), _p(shift2),
                             res.ptr if res is not None else C.c_void_p(0), res.ld if res is not None else 0,
                             out.ptr, out.ld, x.B, x.H, x.W, x.C, w.cout, stride, rate, _act(act), precision,
                             _lib.stream_ptr(stream))
    _lib.check(rc, "emd_conv3x3_f32")
    return out


def conv_stats_supported(x: Act, stride=1, images=False):
    """Can conv1x1_stats / conv3x3_stats deliver the statistics of this convolution's output?  Per-image statistics need whole
    128-row tiles per image."""
    Ho, Wo = -(-x.H // stride), -(-x.W // stride)
    return (not images) or (Ho * Wo) % 128 == 0


def conv_stats(x: Act, w: PackedWeights, ones, zeros, out: Act, stride=1, rate=1, images=False, precision=PREC_BF16X3, stream=None, fold=None):
    """out = conv(x) (1x1, stride 1 / 2; or dense 3x3 with dilation `rate`: by w.taps), no affine, no activation, and the batch
    statistics of out from the GEMM's epilogue -> (mean, var): [Cout], or [B][Cout] flattened with images=True."""
    import torch

    lib = _lib.load()
    Ho, Wo = -(-x.H // stride), -(-x.W // stride)
    assert (out.B, out.H, out.W, out.C) == (x.B, Ho, Wo, w.cout) and w.cin == x.C and w.taps in (1, 9)
    n = x.B * w.cout if images else w.cout
    mean = torch.empty(n, dtype=torch.float32, device=x.buf.device)
    var = torch.empty_like(mean)
    ws = torch.empty(max(lib.emd_conv_stats_workspace_bytes(x.B * Ho * Wo, w.cout) // 8, 1), dtype=torch.float64, device=x.buf.device)
    # fold: a train_ops.FoldRequest -- the training-mode fold of the norm behind the conv runs in the statistics' final kernel too
    if w.taps == 1:
        args = (x.ptr, x.ld, _p(w.hi), _p(w.lo), _p(ones), _p(zeros), out.ptr, out.ld, x.B, x.H, x.W, x.C, w.cout, stride, precision,
                1 if images else 0, _p(mean), _p(var), _p(ws))
        if fold is not None:
            _lib.check(lib.emd_conv1x1_stats_fold_f32(*args, C.byref(fold.struct), _lib.stream_ptr(stream)), "emd_conv1x1_stats_fold_f32")
        else:
            _lib.check(lib.emd_conv1x1_stats_f32(*args, _lib.stream_ptr(stream)), "emd_conv1x1_stats_f32")
    else:
        assert stride == 1
        args = (x.ptr, x.ld, _p(w.hi), _p(w.lo), _p(ones), _p(zeros), out.ptr, out.ld, x.B, x.H, x.W, x.C, w.cout, rate, precision,
                1 if images else 0, _p(mean), _p(var), _p(ws))
        if fold is not None:
            _lib.check(lib.emd_conv3x3_stats_fold_f32(*args, C.byref(fold.struct), _lib.stream_ptr(stream)), "emd_conv3x3_stats_fold_f32")
        else:
            _lib.check(lib.emd_conv3x3_stats_f32(*args, _lib.stream_ptr(stream)), "emd_conv3x3_stats_f32")
    return mean, var


def deconv_stats(x: Act, w_phases, ones, zeros, out: Act, images=False, precision=PREC_BF16X3, stream=None, fold=None):
    """out = deconv3x3s2(x), no affine, no activation, and the batch statistics of out from the four phase GEMMs' epilogues
    (emd_deconv3x3s2_stats_f32) -> (mean, var): [Cout], or [B][Cout] flattened with images=True (needs x.H * x.W % 128 == 0)."""
    import torch

    lib = _lib.load()
    cout = w_phases[0].cout
    assert len(w_phases) == 4 and (out.B, out.H, out.W, out.C) == (x.B, 2 * x.H, 2 * x.W, cout)
    hi = (C.c_void_p * 4)(*[w.hi.data_ptr() for w in w_phases])
    lo = (C.c_void_p * 4)(*[w.lo.data_ptr() for w in w_phases])
    n = x.B * cout if images else cout
    mean = torch.empty(n, dtype=torch.float32, device=x.buf.device)
    var = torch.empty_like(mean)
    ws = torch.empty(max(lib.emd_conv_stats_workspace_bytes(4 * x.B * x.H * x.W, cout) // 8, 1), dtype=torch.float64, device=x.buf.device)
    args = (x.ptr, x.ld, hi, lo, _p(ones), _p(zeros), out.ptr, out.ld, x.B, x.H, x.W, x.C, cout, precision, 1 if images else 0, _p(mean), _p(var),
            _p(ws))
    if fold is not None:
        _lib.check(lib.emd_deconv3x3s2_stats_fold_f32(*args, C.byref(fold.struct), _lib.stream_ptr(stream)), "emd_deconv3x3s2_stats_fold_f32")
    else:
        _lib.check(lib.emd_deconv3x3s2_stats_f32(*args, _lib.stream_ptr(stream)), "emd_deconv3x3s2_stats_f32")
    return mean, var


def avgpool2x2(x: Act, out: Act, stream=None):
    lib = _lib.load()
    assert (out.B, out.H, out.W, out.C) == (x.B, -(-x.H // 2), -(-x.W // 2), x.C)
    rc = lib.emd_avgpool2x2_f32(x.ptr, x.ld, out.ptr, out.ld, x.B, x.H, x.W, x.C, _lib.stream_ptr(stream))
    _lib.check(rc, "emd_avgpool2x2_f32")
    return out


def sep_fused_supported(x: Act, cout: int, stride: int, rate: int) -> bool:
    return bool(_lib.load().emd_sep3x3_fused_supported(x.H, x.W, x.C, cout, stride, rate))


def sep_fused(x: Act, dw_dev, w: PackedWeights, scale1, shift1, out: Act, act=True, scale2=None, shift2=None,
              res: Act | None = None, precision=PREC_BF16X3, stream=None, reflect=False, stride=1):
    """Depthwise 3x3 + pointwise + epilogue in one launch (emd_sep3x3_fused_f32; reflect=True: the depthwise stage reads the
    REFLECT-padded border, emd_sep3x3_fused_reflect_f32; stride=2: emd_sep3x3_fused_s2_f32, split-bf16 only)."""
    lib = _lib.load()
    assert (out.B, out.H, out.W, out.C) == (x.B, -(-x.H // stride), -(-x.W // stride), w.cout) and w.cin == x.C and w.taps == 1
    if res is not None:
        assert (res.B, res.H, res.W, res.C) == (out.B, out.H, out.W, out.C)
    if stride == 2:
        assert precision == PREC_BF16X3 and not isinstance(out, SplitAct)
        rc = (lib.emd_sep3x3_fused_s2_reflect_f32 if reflect else lib.emd_sep3x3_fused_s2_f32)(x.ptr, x.ld, _p(dw_dev), _p(w.hi), _p(w.lo), _p(scale1), _p(shift1), _p(scale2), _p(shift2),
                                         res.ptr if res is not None else C.c_void_p(0), res.ld if res is not None else 0,
                                         out.ptr, out.ld, x.B, x.H, x.W, x.C, w.cout, _act(act), _lib.stream_ptr(stream))
        _lib.check(rc, "emd_sep3x3_fused_s2_f32")
        return out
    assert stride == 1
    if isinstance(out, SplitAct):   # split32 output for a following split32 GEMM (emd_sep3x3_fused_out_f32)
        assert not reflect and precision == PREC_BF16X3
        rc = lib.emd_sep3x3_fused_out_f32(x.ptr, x.ld, _p(dw_dev), _p(w.hi), _p(w.lo), _p(scale1), _p(shift1), _p(scale2), _p(shift2),
                                          res.ptr if res is not None else C.c_void_p(0), res.ld if res is not None else 0,
                                          out.ptr, out.ld, x.B, x.H, x.W, x.C, w.cout, _act(act), _lib.stream_ptr(stream))
        _lib.check(rc, "emd_sep3x3_fused_out_f32")
        return out
    fn = lib.emd_sep3x3_fused_reflect_f32 if reflect else lib.emd_sep3x3_fused_f32
    rc = fn(x.ptr, x.ld, _p(dw_dev), _p(w.hi), _p(w.lo), _p(scale1), _p(shift1), _p(scale2),
                                  _p(shift2), res.ptr if res is not None else C.c_void_p(0),
                                  res.ld if res is not None else 0, out.ptr, out.ld, x.B, x.H, x.W, x.C, w.cout,
                                  _act(act), precision, _lib.stream_ptr(stream))
    _lib.check(rc, "emd_sep3x3_fused_f32")
    return out


def sep_gemm_supported(x: Act, cout: int, stride: int = 1, rate: int = 1) -> bool:
    return stride == 1 and rate == 1 and bool(_lib.load().emd_sep3x3_gemm_supported(x.H, x.W, x.C, cout))


def sep_gemm(x: Act, dw_dev, w: PackedWeights, scale1, shift1, out: Act, act=True, scale2=None, shift2=None, res: Act | None = None,
             stream=None):
    """Separable conv with the depthwise stage computed inside the pointwise GEMM (emd_sep3x3_gemm_f32): the 728-channel flow."""
    lib = _lib.load()
    assert (out.B, out.H, out.W, out.C) == (x.B, x.H, x.W, w.cout) and w.cin == x.C and w.taps == 1
    if res is not None:
        assert (res.B, res.H, res.W, res.C) == (out.B, out.H, out.W, out.C)
    rc = lib.emd_sep3x3_gemm_f32(x.ptr, x.ld, _p(dw_dev), _p(w.hi), _p(w.lo), _p(scale1), _p(shift1), _p(scale2), _p(shift2),
                                 res.ptr if res is not None else C.c_void_p(0), res.ld if res is not None else 0, out.ptr, out.ld,
                                 x.B, x.H, x.W, x.C, w.cout, _act(act), _lib.stream_ptr(stream))
    _lib.check(rc, "emd_sep3x3_gemm_f32")
    return out


def sep_dual_supported(x: Act, cout: int, cout2: int) -> bool:
    return bool(_lib.load().emd_sep3x3_dual_supported(x.H, x.W, x.C, cout, cout2))


def sep_dual_preferred(x: Act, cout: int, cout2: int) -> bool:
    """True where the one-launch form is also the faster route (emd_sep3x3_dual_preferred)."""
    return bool(_lib.load().emd_sep3x3_dual_preferred(x.H, x.W, x.C, cout, cout2))


def sep_dual(x: Act, dw_dev, w: PackedWeights, w2: PackedWeights, scale1, shift1, out: Act, scale_b, shift_b, out2: Act, act=True,
             stream=None):
    """One launch for a decoder pair (emd_sep3x3_dual_f32): out = act(pw(dw3x3(x)) * scale1 + shift1) and
    out2 = relu6((x . w2) * scale_b + shift_b), the 1x1 residual projection of the same input."""
    lib = _lib.load()
    assert (out.B, out.H, out.W, out.C) == (x.B, x.H, x.W, w.cout) and w.cin == x.C and w.taps == 1
    assert (out2.B, out2.H, out2.W, out2.C) == (x.B, x.H, x.W, w2.cout) and w2.cin == x.C and w2.taps == 1
    rc = lib.emd_sep3x3_dual_f32(x.ptr, x.ld, _p(dw_dev), _p(w.hi), _p(w.lo), _p(scale1), _p(shift1), out.ptr, out.ld, _p(w2.hi), _p(w2.lo),
                                 _p(scale_b), _p(shift_b), out2.ptr, out2.ld, x.B, x.H, x.W, x.C, w.cout, w2.cout, _act(act),
                                 _lib.stream_ptr(stream))
    _lib.check(rc, "emd_sep3x3_dual_f32")
    return out, out2


def sep_fused_gen(d: Act, gen_a, gen_t, dw_dev, w: PackedWeights, scale1, shift1, out: Act, gen_act=True, act=True,
                  scale2=None, shift2=None, res: Act | None = None, precision=PREC_BF16X3, stream=None, reflect=False):
    """sep_fused on the generated input act(d[..., 0] * gen_a + gen_t) (emd_sep3x3_fused_gen_f32): d holds one value per
    pixel in its channel 0, the w.cin-channel tensor is rebuilt in registers."""
    lib = _lib.load()
    assert (out.B, out.H, out.W, out.C) == (d.B, d.H, d.W, w.cout) and w.taps == 1
    assert gen_a.numel() == w.cin and gen_t.numel() == w.cin
    if res is not None:
        assert (res.B, res.H, res.W, res.C) == (out.B, out.H, out.W, out.C)
    rc = lib.emd_sep3x3_fused_gen_f32(d.ptr, d.ld, _p(gen_a), _p(gen_t), _act(gen_act), _p(dw_dev), _p(w.hi), _p(w.lo),
                                      _p(scale1), _p(shift1), _p(scale2), _p(shift2),
                                      res.ptr if res is not None else C.c_void_p(0), res.ld if res is not None else 0,
                                      out.ptr, out.ld, d.B, d.H, d.W, w.cin, w.cout, _act(act), precision,
                                      1 if reflect else 0, _lib.stream_ptr(stream))
    _lib.check(rc, "emd_sep3x3_fused_gen_f32")
    return out


def deconv3x3s2(x: Act, w_phases, scale1, shift1, out: Act, act=True, precision=PREC_BF16X3, stream=None):
    lib = _lib.load()
    assert len(w_phases) == 4 and (out.B, out.H, out.W) == (x.B, 2 * x.H, 2 * x.W) and out.C == w_phases[0].cout
    hi = (C.c_void_p * 4)(*[w.hi.data_ptr() for w in w_phases])
    lo = (C.c_void_p * 4)(*[w.lo.data_ptr() for w in w_phases])
    rc = lib.emd_deconv3x3s2_f32(x.ptr, x.ld, hi, lo, _p(scale1), _p(shift1), out.ptr, out.ld, x.B, x.H, x.W, x.C,
                                 out.C, _act(act), precision, _lib.stream_ptr(stream))
    _lib.check(rc, "emd_deconv3x3s2_f32")
    return out


def deconv_phase_taps(phase):
    lib = _lib.load()
    ky = (C.c_int * 4)()
    kx = (C.c_int * 4)()
    n = lib.emd_deconv_phase_taps(phase, ky, kx)
    return [(ky[i], kx[i]) for i in range(n)]


def pack_deconv(w_tf: np.ndarray, device):
    """w_tf [3,3,Cout,Cin] (slim.conv2d_transpose) -> the four per-phase PackedWeights."""
    out = []
    for ph in range(4):
        taps = deconv_phase_taps(ph)
        sub = np.stack([w_tf[ky, kx] for (ky, kx) in taps])  # [taps,Cout,Cin]
        out.append(PackedWeights(sub, True, device))
    return out


def dw3x3(x: Act, w_dev, out: Act, stride=1, rate=1, stream=None, pre=None):
    """pre = (scale, shift): the depthwise conv runs on relu(x*scale + shift) (emd_dw3x3_pre_f32)."""
    lib = _lib.load()
    assert out.C == x.C and out.B == x.B and (out.H, out.W) == (-(-x.H // stride), -(-x.W // stride))
    if pre is not None:
        rc = lib.emd_dw3x3_pre_f32(x.ptr, x.ld, _p(pre[0]), _p(pre[1]), _p(w_dev), out.ptr, out.ld, x.B, x.H, x.W, x.C, stride,
                                   rate, _lib.stream_ptr(stream))
    else:
        rc = lib.emd_dw3x3_f32(x.ptr, x.ld, _p(w_dev), out.ptr, out.ld, x.B, x.H, x.W, x.C, stride, rate,
                               _lib.stream_ptr(stream))
    _lib.check(rc, "emd_dw3x3_f32")
    return out


class PreAct:
    """An activation that was never written: act(r * scale + shift) of the tensor ``r`` (an Act), applied by its consumers while they
    load r (emd_dw3x3_pre_act_f32, emd_dw3x3_wgrad_pre_f32).  scale / shift: device float[C], or [B][C] with images=True (per-image
    statistics).  Shape attributes and ``buf`` / ``c0`` are r's, so that it keys gradient tables like the Act it stands for."""

    __slots__ = ("r", "scale", "shift", "images", "act", "buf", "B", "H", "W", "C", "c0", "ld")

    def __init__(self, r, scale, shift, images=False, act=ACT_RELU6):
        assert act in (ACT_RELU6, ACT_RELU)
        assert scale.numel() == (r.B * r.C if images else r.C) and shift.numel() == scale.numel()
        self.r, self.scale, self.shift, self.images, self.act = r, scale, shift, bool(images), act
        self.buf, self.B, self.H, self.W, self.C, self.c0, self.ld = r.buf, r.B, r.H, r.W, r.C, r.c0, r.ld


def dw3x3_pre_act(x: PreAct, w_dev, out: Act, stride=1, rate=1, stream=None):
    """Depthwise 3x3 of the never-written activation x (emd_dw3x3_pre_act_f32): the bits of affine_act[_images] followed by dw3x3."""
    lib = _lib.load()
    r = x.r
    assert out.C == r.C and out.B == r.B and (out.H, out.W) == (-(-r.H // stride), -(-r.W // stride))
    rc = lib.emd_dw3x3_pre_act_f32(r.ptr, r.ld, _p(x.scale), _p(x.shift), 1 if x.images else 0, _act(x.act), _p(w_dev), out.ptr, out.ld,
                                   r.B, r.H, r.W, r.C, stride, rate, _lib.stream_ptr(stream))
    _lib.check(rc, "emd_dw3x3_pre_act_f32")
    return out


class SplitAct:
    """A split32 activation tensor [B,H,W,C] (include/emdenoise.h: every value as bf16 hi + bf16 lo, 32-channel groups of
    128 bytes); ``buf`` is a float32 torch tensor [B,H,W,ld] of the same bytes, ld = C rounded up to 32."""

    __slots__ = ("buf", "B", "H", "W", "C", "ld")

    def __init__(self, B, H, W, Cc, device):
        import torch

        self.B, self.H, self.W, self.C = B, H, W, Cc
        self.ld = _lib.load().emd_split32_ld(Cc)
        self.buf = torch.empty((B, H, W, self.ld), dtype=torch.float32, device=device)
        assert self.buf.data_ptr() % 128 == 0

    @property
    def ptr(self):
        return C.c_void_p(self.buf.data_ptr())

    def to_float(self):
        """hi + lo as float32 [B,H,W,C] (test helper)."""
        import torch

        g = self.buf.view(torch.bfloat16).view(self.B, self.H, self.W, self.ld // 32, 2, 32).float()
        return (g[..., 0, :] + g[..., 1, :]).reshape(self.B, self.H, self.W, self.ld)[..., : self.C]


def to_split32(x: Act, out: SplitAct | None = None, stream=None):
    lib = _lib.load()
    if out is None:
        out = SplitAct(x.B, x.H, x.W, x.C, x.buf.device)
    assert (out.B, out.H, out.W, out.C) == (x.B, x.H, x.W, x.C)
    _lib.check(lib.emd_to_split32_f32(x.ptr, x.ld, out.ptr, out.ld, C.c_long(x.B * x.H * x.W), x.C,
                                      _lib.stream_ptr(stream)), "emd_to_split32_f32")
    return out


def dw3x3_split32(x: Act, w_dev, out: SplitAct, stride=1, rate=1, stream=None, pre=None):
    lib = _lib.load()
    assert out.C == x.C and out.B == x.B and (out.H, out.W) == (-(-x.H // stride), -(-x.W // stride))
    if pre is not None:
        rc = lib.emd_dw3x3_pre_split32_f32(x.ptr, x.ld, _p(pre[0]), _p(pre[1]), _p(w_dev), out.ptr, out.ld, x.B, x.H, x.W, x.C,
                                           stride, rate, _lib.stream_ptr(stream))
    else:
        rc = lib.emd_dw3x3_split32_f32(x.ptr, x.ld, _p(w_dev), out.ptr, out.ld, x.B, x.H, x.W, x.C, stride, rate,
                                       _lib.stream_ptr(stream))
    _lib.check(rc, "emd_dw3x3_split32_f32")
    return out


def dw3x3_reflect_split32(x: Act, w_dev, out: SplitAct, stride=1, stream=None):
    lib = _lib.load()
    assert out.C == x.C and out.B == x.B and (out.H, out.W) == ((x.H - 1) // stride + 1, (x.W - 1) // stride + 1)
    _lib.check(lib.emd_dw3x3_reflect_split32_f32(x.ptr, x.ld, _p(w_dev), out.ptr, out.ld, x.B, x.H, x.W, x.C, stride,
                                                 _lib.stream_ptr(stream)), "emd_dw3x3_reflect_split32_f32")
    return out


def sep_split32(x: Act, dw_dev, w: PackedWeights, scale1, shift1, out: Act, stride=1, rate=1, act=True, scale2=None,
                shift2=None, res: Act | None = None, reflect=False, stream=None, pre=None, stats=False, fold=None):
    """Separable conv as depthwise (split32 output) -> LDS-DMA pointwise GEMM; the intermediate exists only in split form.
    stats=True: (out, mean, var) with the batch statistics of the output from the GEMM epilogue."""
    d = SplitAct(out.B, out.H, out.W, x.C, x.buf.device)
    if reflect:
        assert pre is None
        dw3x3_reflect_split32(x, dw_dev, d, stride=stride, stream=stream)
    else:
        dw3x3_split32(x, dw_dev, d, stride=stride, rate=rate, stream=stream, pre=pre)
    return conv1x1_split32(d, w, scale1, shift1, out, act=act, scale2=scale2, shift2=shift2, res=res, stream=stream, stats=stats,
                           fold=fold)


def conv3x3_split32(x: SplitAct, w: PackedWeights, scale1, shift1, out, stride=1, rate=1, act=True, scale2=None, shift2=None,
                    res: Act | None = None, stream=None):
    """Dense 3x3 conv on a split32 input; ``out`` an Act (fp32) or a SplitAct (split32 output for a following split32 conv)."""
    lib = _lib.load()
    Ho, Wo = -(-x.H // stride), -(-x.W // stride)
    assert (out.B, out.H, out.W, out.C) == (x.B, Ho, Wo, w.cout) and w.cin == x.C and w.taps == 9
    rc = lib.emd_conv3x3_split32_f32(x.ptr, x.ld, _p(w.hi), _p(w.lo), _p(scale1), _p(shift1), _p(scale2), _p(shift2),
                                     res.ptr if res is not None else C.c_void_p(0), res.ld if res is not None else 0,
                                     out.ptr, out.ld, x.B, x.H, x.W, x.C, w.cout, stride, rate, _act(act),
                                     1 if isinstance(out, SplitAct) else 0, _lib.stream_ptr(stream))
    _lib.check(rc, "emd_conv3x3_split32_f32")
    return out


def deconv3x3s2_split32(x: SplitAct, w_phases, scale1, shift1, out, act=True, stream=None):
    lib = _lib.load()
    assert len(w_phases) == 4 and (out.B, out.H, out.W) == (x.B, 2 * x.H, 2 * x.W) and out.C == w_phases[0].cout
    hi = (C.c_void_p * 4)(*[w.hi.data_ptr() for w in w_phases])
    lo = (C.c_void_p * 4)(*[w.lo.data_ptr() for w in w_phases])
    rc = lib.emd_deconv3x3s2_split32_f32(x.ptr, x.ld, hi, lo, _p(scale1), _p(shift1), out.ptr, out.ld, x.B, x.H, x.W, x.C,
                                         out.C, _act(act), 1 if isinstance(out, SplitAct) else 0, _lib.stream_ptr(stream))
    _lib.check(rc, "emd_deconv3x3s2_split32_f32")
    return out


def deconv3x3s2_fused(x: SplitAct, w_phases, scale1, shift1, out, act=True, stream=None):
    """The transposed conv as one launch (emd_deconv3x3s2_fused_split32_f32): every workgroup runs the four output phases of its
    input pixels -- the input comes from HBM once.  Where H % 8 == 0 and W % 32 == 0 the patch-resident kernel (csrc/deconv_pipe.hip;
    chunk-major sums: last-bit differences to the GEMM forms), otherwise the GEMM form, bit-identical to deconv3x3s2_split32."""
    lib = _lib.load()
    assert len(w_phases) == 4 and (out.B, out.H, out.W) == (x.B, 2 * x.H, 2 * x.W) and out.C == w_phases[0].cout
    hi = (C.c_void_p * 4)(*[w.hi.data_ptr() for w in w_phases])
    lo = (C.c_void_p * 4)(*[w.lo.data_ptr() for w in w_phases])
    rc = lib.emd_deconv3x3s2_fused_split32_f32(x.ptr, x.ld, hi, lo, _p(scale1), _p(shift1), out.ptr, out.ld, x.B, x.H, x.W, x.C,
                                               out.C, _act(act), 1 if isinstance(out, SplitAct) else 0, _lib.stream_ptr(stream))
    _lib.check(rc, "emd_deconv3x3s2_fused_split32_f32")
    return out


def conv1x1_split32_supported(npix: int, cin: int, cout: int) -> bool:
    return bool(_lib.load().emd_conv1x1_split32_supported(C.c_long(npix), cin, cout))


def conv1x1_split32(x: SplitAct, w: PackedWeights, scale1, shift1, out: Act, act=True, scale2=None, shift2=None,
                    res: Act | None = None, stream=None, stats=False, fold=None):
    """Pointwise conv on a split32 input (split-bf16 precision; bit-identical to conv1x1 on the fp32 twin).
    stats=True: returns (out, mean, var), the batch statistics of the output gathered in the GEMM epilogue; with
    fold=(gamma or None, beta or None, eps) also the folded norm: (out, mean, var, scale, shift), same launch count."""
    lib = _lib.load()
    assert (out.B, out.H, out.W, out.C) == (x.B, x.H, x.W, w.cout) and w.cin == x.C and w.taps == 1
    if stats:
        import torch

        assert scale2 is None and res is None
        M = x.B * x.H * x.W
        mean = torch.empty(w.cout, dtype=torch.float32, device=x.buf.device)
        var = torch.empty_like(mean)
        ws = torch.empty(lib.emd_conv1x1_split32_stats_workspace_bytes(C.c_long(M), w.cout) // 8, dtype=torch.float64,
                         device=x.buf.device)
        if fold is not None:
            scale, shift = torch.empty_like(mean), torch.empty_like(mean)
            rc = lib.emd_conv1x1_split32_stats_fold_f32(x.ptr, x.ld, _p(w.hi), _p(w.lo), _p(scale1), _p(shift1), out.ptr, out.ld,
                                                        C.c_long(M), x.C, w.cout, _act(act), _p(mean), _p(var), _p(ws),
                                                        _p(fold[0]), _p(fold[1]), C.c_float(fold[2]), _p(scale), _p(shift),
                                                        _lib.stream_ptr(stream))
            _lib.check(rc, "emd_conv1x1_split32_stats_fold_f32")
            return out, mean, var, scale, shift
        rc = lib.emd_conv1x1_split32_stats_f32(x.ptr, x.ld, _p(w.hi), _p(w.lo), _p(scale1), _p(shift1), out.ptr, out.ld,
                                               C.c_long(M), x.C, w.cout, _act(act), _p(mean), _p(var), _p(ws),
                                               _lib.stream_ptr(stream))
        _lib.check(rc, "emd_conv1x1_split32_stats_f32")
        return out, mean, var
    if res is not None:
        assert (res.B, res.H, res.W, res.C) == (out.B, out.H, out.W, out.C)
    if isinstance(out, SplitAct):   # split32 output for a following split32 GEMM (emd_conv1x1_split32_out_f32)
        rc = lib.emd_conv1x1_split32_out_f32(x.ptr, x.ld, _p(w.hi), _p(w.lo), _p(scale1), _p(shift1), _p(scale2), _p(shift2),
                                             res.ptr if res is not None else C.c_void_p(0), res.ld if res is not None else 0,
                                             out.ptr, out.ld, C.c_long(x.B * x.H * x.W), x.C, w.cout, _act(act),
                                             _lib.stream_ptr(stream))
        _lib.check(rc, "emd_conv1x1_split32_out_f32")
        return out
    rc = lib.emd_conv1x1_split32_f32(x.ptr, x.ld, _p(w.hi), _p(w.lo), _p(scale1), _p(shift1), _p(scale2), _p(shift2),
                                     res.ptr if res is not None else C.c_void_p(0), res.ld if res is not None else 0,
                                     out.ptr, out.ld, C.c_long(x.B * x.H * x.W), x.C, w.cout, _act(act),
                                     _lib.stream_ptr(stream))
    _lib.check(rc, "emd_conv1x1_split32_f32")
    return out


def conv3x3_cin1(x_img, w_dev, scale, shift, out, stride=1, act=True, stream=None):
    """Dense 3x3 conv of a 1-channel image (emd_conv3x3_cin1_f32).  x_img: torch CUDA float32 [B,H,W] or [B,H,W,1] contiguous; w_dev
    [9, Cout]; out an Act (fp32) or a SplitAct (split32)."""
    lib = _lib.load()
    B, H, W = x_img.shape[0], x_img.shape[1], x_img.shape[2]
    assert x_img.is_contiguous() and (out.B, out.H, out.W) == (B, -(-H // stride), -(-W // stride)) and w_dev.shape == (9, out.C)
    rc = lib.emd_conv3x3_cin1_f32(_p(x_img), _p(w_dev), _p(scale), _p(shift), out.ptr, out.ld, B, H, W, out.C, stride, _act(act),
                                  1 if isinstance(out, SplitAct) else 0, _lib.stream_ptr(stream))
    _lib.check(rc, "emd_conv3x3_cin1_f32")
    return out


def cin1(x_img, w9_dev, a_dev, shift_dev, out: Act, stride=1, act=True, stream=None):
    """x_img: torch CUDA float32 [B,H,W] or [B,H,W,1] contiguous."""
    lib = _lib.load()
    B, H, W = x_img.shape[0], x_img.shape[1], x_img.shape[2]
    assert x_img.is_contiguous() and (out.B, out.H, out.W) == (B, -(-H // stride), -(-W // stride))
    rc = lib.emd_cin1_f32(_p(x_img), _p(w9_dev), _p(a_dev), _p(shift_dev), out.ptr, out.ld, B, H, W, out.C, stride,
                          1 if act else 0, _lib.stream_ptr(stream))
    _lib.check(rc, "emd_cin1_f32")
    return out


def conv3x3_cout1(x: Act, w_dev, scale: float, shift: float, out_img, act=True, pre_bias=0.0, pre_relu=False,
                  stream=None):
    """act: False/0 none, True/1 relu6, 2 relu6 then clip to [0,1]; pre_relu: relu(conv + pre_bias) before the affine."""
    lib = _lib.load()
    assert out_img.is_contiguous() and out_img.numel() == x.B * x.H * x.W
    rc = lib.emd_conv3x3_cout1_f32(x.ptr, x.ld, _p(w_dev), C.c_float(scale), C.c_float(shift), _p(out_img), x.B, x.H,
                                   x.W, x.C, int(act), C.c_float(pre_bias), 1 if pre_relu else 0, _lib.stream_ptr(stream))
    _lib.check(rc, "emd_conv3x3_cout1_f32")
    return out_img


def resize_bilinear(x: Act, out: Act, stream=None):
    lib = _lib.load()
    assert out.C == x.C and out.B == x.B
    rc = lib.emd_resize_bilinear_f32(x.ptr, x.ld, out.ptr, out.ld, x.B, x.H, x.W, out.H, out.W, x.C,
                                     _lib.stream_ptr(stream))
    _lib.check(rc, "emd_resize_bilinear_f32")
    return out


def affine_relu6(x: Act, scale_dev, shift_dev, out: Act, act=True, stream=None):
    lib = _lib.load()
    assert (out.B, out.H, out.W, out.C) == (x.B, x.H, x.W, x.C)
    rc = lib.emd_affine_relu6_f32(x.ptr, x.ld, _p(scale_dev), _p(shift_dev), out.ptr, out.ld,
                                  C.c_long(x.B * x.H * x.W), x.C, 1 if act else 0, _lib.stream_ptr(stream))
    _lib.check(rc, "emd_affine_relu6_f32")
    return out


def affine_act(x: Act, scale_dev, shift_dev, out: Act, act=ACT_RELU, res: Act | None = None, stream=None):
    """out = act(x*scale + shift) [+ res]; out may be x."""
    lib = _lib.load()
    assert (out.B, out.H, out.W, out.C) == (x.B, x.H, x.W, x.C)
    rc = lib.emd_affine_act_f32(x.ptr, x.ld, _p(scale_dev), _p(shift_dev), res.ptr if res is not None else C.c_void_p(0),
                                res.ld if res is not None else 0, out.ptr, out.ld, C.c_long(x.B * x.H * x.W), x.C,
                                _act(act), _lib.stream_ptr(stream))
    _lib.check(rc, "emd_affine_act_f32")
    return out


def bn_batch_stats(x: Act, stream=None):
    """Per-channel batch mean and biased variance of x over (B,H,W): two float32 CUDA vectors."""
    import torch

    lib = _lib.load()
    npix = x.B * x.H * x.W
    mean = torch.empty(x.C, dtype=torch.float32, device=x.buf.device)
    var = torch.empty_like(mean)
    ws = torch.empty(lib.emd_bn_stats_workspace_bytes(npix, x.C) // 8, dtype=torch.float64, device=x.buf.device)
    _lib.check(lib.emd_bn_stats_f32(x.ptr, x.ld, C.c_long(npix), x.C, _p(mean), _p(var), _p(ws),
                                    _lib.stream_ptr(stream)), "emd_bn_stats_f32")
    return mean, var


def bn_batch_stats_images(x: Act, stream=None):
    """Per-image, per-channel mean and biased variance of x [B,H,W,C]: two float32 CUDA vectors of B*C entries ([B][C])."""
    import torch

    lib = _lib.load()
    npix = x.H * x.W
    mean = torch.empty(x.B * x.C, dtype=torch.float32, device=x.buf.device)
    var = torch.empty_like(mean)
    ws = torch.empty(x.B * (lib.emd_bn_stats_workspace_bytes(npix, x.C) // 8), dtype=torch.float64, device=x.buf.device)
    _lib.check(lib.emd_bn_stats_images_f32(x.ptr, x.ld, x.B, C.c_long(npix), x.C, _p(mean), _p(var), _p(ws),
                                           _lib.stream_ptr(stream)), "emd_bn_stats_images_f32")
    return mean, var


def affine_act_images(x: Act, scale_dev, shift_dev, out: Act, act=ACT_RELU, res: Act | None = None, stream=None):
    """out = act(x*scale[b] + shift[b]) [+ res] with per-image scale / shift ([B][C]); out may be x."""
    lib = _lib.load()
    assert (out.B, out.H, out.W, out.C) == (x.B, x.H, x.W, x.C) and scale_dev.numel() == x.B * x.C
    rc = lib.emd_affine_act_images_f32(x.ptr, x.ld, _p(scale_dev), _p(shift_dev), res.ptr if res is not None else C.c_void_p(0),
                                       res.ld if res is not None else 0, out.ptr, out.ld, x.B, C.c_long(x.H * x.W), x.C,
                                       _act(act), _lib.stream_ptr(stream))
    _lib.check(rc, "emd_affine_act_images_f32")
    return out


def affine_act_res_pre(x: Act, scale_dev, shift_dev, out: Act, res: "PreAct", act=ACT_RELU6, stream=None):
    """out = act(x*scale + shift) + res_act(res.r*res.scale + res.shift): the residual operand a never-written PreAct
    (emd_affine_act_res_affine_f32); scale / shift per image exactly when res's are."""
    lib = _lib.load()
    r = res.r
    assert (out.B, out.H, out.W, out.C) == (x.B, x.H, x.W, x.C) == (r.B, r.H, r.W, r.C)
    assert scale_dev.numel() == (x.B * x.C if res.images else x.C)
    rc = lib.emd_affine_act_res_affine_f32(x.ptr, x.ld, _p(scale_dev), _p(shift_dev), r.ptr, r.ld, _p(res.scale), _p(res.shift), _act(res.act),
                                           out.ptr, out.ld, x.B if res.images else 0, C.c_long(x.H * x.W if res.images else x.B * x.H * x.W),
                                           x.C, _act(act), _lib.stream_ptr(stream))
    _lib.check(rc, "emd_affine_act_res_affine_f32")
    return out


def bn_fold(mean, var, gamma, beta, eps=1e-3, stream=None):
    """(mean, var, gamma|None, beta|None) -> (scale, shift) device vectors of the equivalent affine."""
    import torch

    lib = _lib.load()
    scale = torch.empty_like(mean)
    shift = torch.empty_like(mean)
    _lib.check(lib.emd_bn_fold_f32(_p(mean), _p(var), _p(gamma), _p(beta), C.c_float(eps), _p(scale), _p(shift),
                                   mean.numel(), _lib.stream_ptr(stream)), "emd_bn_fold_f32")
    return scale, shift


# ---- graph G (the in-filling generator, misc_py/gan-infilling-100.py:133-374)
def dw3x3_reflect(x: Act, w_dev, out: Act, stride=1, stream=None):
    """Depthwise 3x3 over the reflect-padded (1 px) input, VALID."""
    lib = _lib.load()
    assert out.C == x.C and out.B == x.B and (out.H, out.W) == ((x.H - 1) // stride + 1, (x.W - 1) // stride + 1)
    _lib.check(lib.emd_dw3x3_reflect_f32(x.ptr, x.ld, _p(w_dev), out.ptr, out.ld, x.B, x.H, x.W, x.C, stride,
                                         _lib.stream_ptr(stream)), "emd_dw3x3_reflect_f32")
    return out


def cin1_k7_reflect(x_img, w49_dev, a_dev, shift_dev, out: Act, act=True, stream=None):
    lib = _lib.load()
    B, H, W = x_img.shape[0], x_img.shape[1], x_img.shape[2]
    assert x_img.is_contiguous() and (out.B, out.H, out.W) == (B, H, W) and w49_dev.numel() == 49
    _lib.check(lib.emd_cin1_k7_reflect_f32(_p(x_img), _p(w49_dev), _p(a_dev), _p(shift_dev), out.ptr, out.ld, B, H, W, out.C,
                                           1 if act else 0, _lib.stream_ptr(stream)), "emd_cin1_k7_reflect_f32")
    return out


def dw3x3_reflect_gen(d: Act, gen_a, gen_t, w_dev, out: Act, stride=1, leaky=True, stream=None):
    """dw3x3_reflect over the generated input act(d[..., 0] * gen_a + gen_t) (emd_dw3x3_reflect_gen_f32)."""
    lib = _lib.load()
    assert (out.B, out.H, out.W) == (d.B, (d.H - 1) // stride + 1, (d.W - 1) // stride + 1) and gen_a.numel() == out.C
    _lib.check(lib.emd_dw3x3_reflect_gen_f32(d.ptr, d.ld, _p(gen_a), _p(gen_t), 1 if leaky else 0, _p(w_dev), out.ptr, out.ld,
                                             d.B, d.H, d.W, out.C, stride, _lib.stream_ptr(stream)), "emd_dw3x3_reflect_gen_f32")
    return out


def conv3x3_cout1_reflect(x: Act, w_dev, bias: float, out_img, stream=None):
    lib = _lib.load()
    assert out_img.is_contiguous() and out_img.numel() == x.B * x.H * x.W
    _lib.check(lib.emd_conv3x3_cout1_reflect_f32(x.ptr, x.ld, _p(w_dev), C.c_float(bias), _p(out_img), x.B, x.H, x.W, x.C,
                                                 _lib.stream_ptr(stream)), "emd_conv3x3_cout1_reflect_f32")
    return out_img


def instnorm_tanh(x_img, out_img, eps=1e-3, stream=None):
    """tanh(instance_norm(x)) of a 1-channel batch [B,H,W,1]; the per-image statistics are reduced on the device."""
    import torch

    lib = _lib.load()
    B = x_img.shape[0]
    npix = x_img.numel() // B
    mean = torch.empty(B, dtype=torch.float32, device=x_img.device)
    var = torch.empty_like(mean)
    # the statistics of all B images in one pair of launches (each image reduced exactly as it would be alone)
    ws = torch.empty(B * max(lib.emd_bn_stats_workspace_bytes(npix, 1) // 8, 1), dtype=torch.float64, device=x_img.device)
    _lib.check(lib.emd_bn_stats_images_f32(_p(x_img), 1, B, C.c_long(npix), 1, _p(mean), _p(var), _p(ws),
                                           _lib.stream_ptr(stream)), "emd_bn_stats_images_f32")
    _lib.check(lib.emd_instnorm_tanh_f32(_p(x_img), _p(mean), _p(var), _p(out_img), B, C.c_long(npix), C.c_float(eps),
                                         _lib.stream_ptr(stream)), "emd_instnorm_tanh_f32")
    return out_img
