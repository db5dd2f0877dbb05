"""Torch-tensor wrappers over the training-path entry points of libemdenoise.so (include/emdenoise.h, second
half): weight gradients, transposed-weight packing on the device, training-mode batch norm, the backward of the
HBM-bound layers, the loss and the Nesterov step of misc_py/denoiser-multi-gpu.py:752-782 / :1011-1077.
Shape checks and pointer plumbing only; all arithmetic is in the HIP library.
"""
from __future__ import annotations

import ctypes as C

from . import _lib
from .ops import PREC_BF16X3, Act, _act, _p

MASK_NONE, MASK_RELU6, MASK_RELU6_CLIP, MASK_LEAKY = 0, 1, 2, 3
BN_EPS = 1e-3
BN_DECAY = 0.999  # tf.contrib.layers.batch_norm default


def _ints(v):
    return (C.c_int * len(v))(*[int(a) for a in v])


def same_pad_before(n, stride, rate):
    o = -(-n // stride)
    return max((o - 1) * stride + 2 * rate + 1 - n, 0) // 2


class DevPackedWeights:
    """bf16 hi/lo planes (emd_pack_weights_bf16 layout) refreshed on the device from an fp32 weight tensor.
    Quacks like ops.PackedWeights (taps, cin, cout, hi, lo) for the forward entry points."""

    def __init__(self, taps, cin, cout, device):
        import torch

        n = _lib.load().emd_packed_weight_elems(taps, cin, cout)
        self.taps, self.cin, self.cout = taps, cin, cout
        self.hi = torch.empty(n, dtype=torch.int16, device=device)
        self.lo = torch.empty(n, dtype=torch.int16, device=device)

    def pack(self, w_dev, src_taps, cout_major, tap_sel=None, stream=None):
        """w_dev: fp32 device tensor [src_taps][cin][cout] (cout_major False) or [src_taps][cout][cin] (True), where
        cin/cout are THIS pack's GEMM K and N."""
        assert w_dev.is_contiguous() and w_dev.numel() == src_taps * self.cin * self.cout
        sel = _ints(tap_sel) if tap_sel is not None else None
        rc = _lib.load().emd_pack_weights_dev(_p(w_dev), src_taps, self.taps, sel, self.cin, self.cout,
                                              1 if cout_major else 0, _p(self.hi), _p(self.lo), _lib.stream_ptr(stream))
        _lib.check(rc, "emd_pack_weights_dev")
        return self


class PackBatch:
    """Every DevPackedWeights.pack call of a model as ONE launch (emd_pack_weights_batch_dev): add() the calls once -- the
    pointers must stay where they are -- then run() after every optimizer step."""

    def __init__(self, device):
        self.device, self.jobs, self.table, self.n_blocks, self._keep = device, [], None, 0, []

    def add(self, dst: DevPackedWeights, w_dev, src_taps, cout_major, tap_sel=None):
        assert self.table is None and w_dev.is_contiguous() and w_dev.numel() == src_taps * dst.cin * dst.cout
        job = _lib.PackJob()
        rc = _lib.load().emd_pack_job_fill(C.byref(job), _p(w_dev), src_taps, dst.taps, _ints(tap_sel) if tap_sel is not None else None,
                                           dst.cin, dst.cout, 1 if cout_major else 0, _p(dst.hi), _p(dst.lo))
        _lib.check(rc, "emd_pack_job_fill")
        job.first_block = self.n_blocks
        self.n_blocks += job.n_blocks
        self.jobs.append(job)
        self._keep.append((dst, w_dev))

    def run(self, stream=None):
        import numpy as np
        import torch

        if not self.jobs:
            return
        if self.table is None:
            arr = (_lib.PackJob * len(self.jobs))(*self.jobs)
            host = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
            self.table = torch.from_numpy(host).to(self.device)
        _lib.check(_lib.load().emd_pack_weights_batch_dev(_p(self.table), len(self.jobs), self.n_blocks, _lib.stream_ptr(stream)),
                   "emd_pack_weights_batch_dev")


def conv_wgrad(a: Act, dy: Act, dw_dev, taps_dy=None, taps_dx=None, sa=1, stream=None):
    """dw_dev [ntaps][a.C][dy.C] += sum over dy's grid of a[src_t] (x) dy."""
    ntaps = 1 if taps_dy is None else len(taps_dy)
    assert dw_dev.is_contiguous() and dw_dev.numel() == ntaps * a.C * dy.C and a.B == dy.B
    rc = _lib.load().emd_conv_wgrad_f32(a.ptr, a.ld, dy.ptr, dy.ld, _p(dw_dev), dy.B, dy.H, dy.W, a.H, a.W, a.C, dy.C,
                                        ntaps, _ints(taps_dy) if taps_dy is not None else None,
                                        _ints(taps_dx) if taps_dx is not None else None, sa, _lib.stream_ptr(stream))
    _lib.check(rc, "emd_conv_wgrad_f32")


def conv_taps(H, W, stride, rate):
    """(tap_dy, tap_dx) of a 3x3 TF-SAME conv on an HxW input."""
    pt, pl = same_pad_before(H, stride, rate), same_pad_before(W, stride, rate)
    return [ky * rate - pt for ky in range(3) for _ in range(3)], [kx * rate - pl for _ in range(3) for kx in range(3)]


def conv1x1_s2_bwd_data(dy: Act, w: DevPackedWeights, ones, zeros, dx: Act, accumulate, precision=PREC_BF16X3, stream=None):
    """dx[b,2i,2j,:] (+)= dy[b,i,j,:] W^T; w packed transposed (K = Cout, N = Cin)."""
    assert (dy.H, dy.W) == (-(-dx.H // 2), -(-dx.W // 2)) and w.cin == dy.C and w.cout == dx.C and w.taps == 1
    rc = _lib.load().emd_conv1x1_s2_bwd_data_f32(dy.ptr, dy.ld, _p(w.hi), _p(w.lo), _p(ones), _p(zeros),
                                                 dx.ptr if accumulate else C.c_void_p(0), dx.ld if accumulate else 0,
                                                 dx.ptr, dx.ld, dx.B, dx.H, dx.W, dy.C, dx.C, precision,
                                                 _lib.stream_ptr(stream))
    _lib.check(rc, "emd_conv1x1_s2_bwd_data_f32")


def _reduce_ws(npix, Cc, device):
    import torch

    return torch.empty(max(_lib.load().emd_chan_reduce_workspace_bytes(npix, Cc) // 8, 1), dtype=torch.float64, device=device)


def bn_train_fold(mean, var, gamma2, beta2, npix, gamma1=None, beta1=None, bias=None, moving=None, eps=BN_EPS,
                  decay=BN_DECAY, stream=None, images=0):
    """-> dict(scale, shift, rstd1, rstd2|None).  moving: None, (mm2, mv2) for a single BN, or (mm1, mv1, mm2, mv2).
    images = B > 0: per-image statistics mean / var [B][C] (ops.bn_batch_stats_images), npix pixels per image; the result vectors
    are [B][C] too and the dict carries "B" (the per-image forms of affine_act / bn_backward follow from it)."""
    import torch

    if images:
        Cc = mean.numel() // images
        scale, shift, rstd1 = torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(mean)
        rstd2 = torch.empty_like(mean) if gamma1 is not None else None
        mm1 = mv1 = mm2 = mv2 = None
        if moving is not None:
            if gamma1 is not None:
                mm1, mv1, mm2, mv2 = moving
            else:
                mm2, mv2 = moving
        rc = _lib.load().emd_bn_train_fold_images_f32(_p(mean), _p(var), _p(gamma1), _p(beta1), _p(gamma2), _p(beta2), _p(bias),
                                                      C.c_float(eps), C.c_long(npix), images, Cc, _p(scale), _p(shift), _p(rstd1),
                                                      _p(rstd2), _p(mm1), _p(mv1), _p(mm2), _p(mv2), C.c_double(decay),
                                                      _lib.stream_ptr(stream))
        _lib.check(rc, "emd_bn_train_fold_images_f32")
        return {"scale": scale, "shift": shift, "rstd1": rstd1, "rstd2": rstd2, "mean": mean, "B": images}
    Cc = mean.numel()
    scale, shift, rstd1 = torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(mean)
    rstd2 = torch.empty_like(mean) if gamma1 is not None else None
    mm1 = mv1 = mm2 = mv2 = None
    if moving is not None:
        if gamma1 is not None:
            mm1, mv1, mm2, mv2 = moving
        else:
            mm2, mv2 = moving
    rc = _lib.load().emd_bn_train_fold_f32(_p(mean), _p(var), _p(gamma1), _p(beta1), _p(gamma2), _p(beta2), _p(bias),
                                           C.c_float(eps), C.c_long(npix), Cc, _p(scale), _p(shift), _p(rstd1), _p(rstd2),
                                           _p(mm1), _p(mv1), _p(mm2), _p(mv2), C.c_double(decay), _lib.stream_ptr(stream))
    _lib.check(rc, "emd_bn_train_fold_f32")
    return {"scale": scale, "shift": shift, "rstd1": rstd1, "rstd2": rstd2, "mean": mean}


class FoldRequest:
    """The argument block (emd_bn_train_fold_t) for the convolutions that run bn_train_fold's per-channel step in their statistics' final
    kernel (ops.conv_stats / ops.deconv_stats with fold=...): outputs allocated here, ``result(mean)`` = the dict bn_train_fold returns."""

    def __init__(self, device, images, Cc, gamma2, beta2, gamma1=None, beta1=None, bias=None, moving=None, eps=BN_EPS, decay=BN_DECAY):
        import torch

        n = (images or 1) * Cc
        self.images = images
        self.scale, self.shift, self.rstd1 = (torch.empty(n, dtype=torch.float32, device=device) for _ in range(3))
        self.rstd2 = torch.empty(n, dtype=torch.float32, device=device) if gamma1 is not None else None
        mm1 = mv1 = mm2 = mv2 = None
        if moving is not None:
            if gamma1 is not None:
                mm1, mv1, mm2, mv2 = moving
            else:
                mm2, mv2 = moving
        self._keep = (gamma1, beta1, gamma2, beta2, bias, mm1, mv1, mm2, mv2)
        q = lambda t: t.data_ptr() if t is not None else None
        self.struct = _lib.BnTrainFold(q(gamma1), q(beta1), q(gamma2), q(beta2), q(bias), eps, 0.0, q(self.scale), q(self.shift), q(self.rstd1),
                                       q(self.rstd2), q(mm1), q(mv1), q(mm2), q(mv2), decay)

    def result(self, mean):
        d = {"scale": self.scale, "shift": self.shift, "rstd1": self.rstd1, "rstd2": self.rstd2, "mean": mean}
        if self.images:
            d["B"] = self.images
        return d


FUSE_PREP = __import__("os").environ.get("EMD_T_FUSE_PREP", "1") == "1"   # bn_bwd_prep's per-channel step inside the reduction's final kernel


def _prep_struct(fold, gamma1, gamma2, eps, K, m1, m2, dgamma1, dgamma2, dbeta2):
    q = lambda t: t.data_ptr() if t is not None else None
    return _lib.BnBwdPrep(q(gamma1), q(gamma2), q(fold["rstd1"]), q(fold["rstd2"]), eps, 0.0, q(K), q(m1), q(m2), q(dgamma1), q(dgamma2), q(dbeta2))


def bn_small_supported(r: Act):
    """The one-launch training batch norm (emd_bn_train_fwd_small_f32 / _bwd_small_f32) takes per-image maps of up to 4096 pixels."""
    return bool(_lib.load().emd_bn_train_small_supported(C.c_long(r.H * r.W), r.C)) and r.ld % 4 == 0


def bn_train_fwd_small(r: Act, gamma2, beta2, out: Act, act, gamma1=None, beta1=None, bias=None, moving=None, res: Act | None = None,
                       eps=BN_EPS, decay=BN_DECAY, stream=None):
    """Per-image batch statistics of r, the fold of the layer's BN chain, out = act(r * scale + shift) [+ res] and the moving-average
    update (from image 0) in ONE launch -> the fold dict bn_train_fold returns (per-image form)."""
    import torch

    n = r.B * r.C
    dev = r.buf.device
    scale, shift, rstd1, mean = (torch.empty(n, dtype=torch.float32, device=dev) for _ in range(4))
    rstd2 = torch.empty(n, dtype=torch.float32, device=dev) if gamma1 is not None else None
    mm1 = mv1 = mm2 = mv2 = None
    if moving is not None:
        if gamma1 is not None:
            mm1, mv1, mm2, mv2 = moving
        else:
            mm2, mv2 = moving
    rc = _lib.load().emd_bn_train_fwd_small_f32(r.ptr, r.ld, r.B, C.c_long(r.H * r.W), r.C, _p(gamma1), _p(beta1), _p(gamma2), _p(beta2),
                                                _p(bias), C.c_float(eps), _p(scale), _p(shift), _p(rstd1), _p(rstd2), _p(mean), _p(mm1),
                                                _p(mv1), _p(mm2), _p(mv2), C.c_double(decay), res.ptr if res is not None else C.c_void_p(0),
                                                res.ld if res is not None else 0, out.ptr, out.ld, act, _lib.stream_ptr(stream))
    _lib.check(rc, "emd_bn_train_fwd_small_f32")
    return {"scale": scale, "shift": shift, "rstd1": rstd1, "rstd2": rstd2, "mean": mean, "B": r.B, "small": True}


def bn_backward_small(dy: Act, r: Act, fold, gamma2, dgamma2, dbeta2, dr: Act, mask=MASK_RELU6, gamma1=None, dgamma1=None, eps=BN_EPS,
                      stream=None):
    """bn_backward (per-image form) as one launch."""
    assert fold.get("B") == dy.B
    ms, mh = (fold["scale"], fold["shift"]) if mask else (None, None)
    rc = _lib.load().emd_bn_train_bwd_small_f32(dy.ptr, dy.ld, r.ptr, r.ld, dy.B, C.c_long(dy.H * dy.W), dy.C, _p(fold["mean"]),
                                                _p(fold["rstd1"]), _p(fold["rstd2"]), _p(ms), _p(mh), mask, _p(gamma1), _p(gamma2),
                                                C.c_float(eps), _p(dgamma1), _p(dgamma2), _p(dbeta2), dr.ptr, dr.ld,
                                                _lib.stream_ptr(stream))
    _lib.check(rc, "emd_bn_train_bwd_small_f32")
    return dr


def chan_reduce(dy: Act, s1, x: Act | None = None, mean=None, rstd=None, s2=None, mscale=None, mshift=None,
                mask=MASK_NONE, accumulate_s1=False, stream=None):
    npix = dy.B * dy.H * dy.W
    ws = _reduce_ws(npix, dy.C, dy.buf.device)
    rc = _lib.load().emd_bn_bwd_reduce_f32(dy.ptr, dy.ld, x.ptr if x is not None else C.c_void_p(0),
                                           x.ld if x is not None else 0, _p(mean), _p(rstd), _p(mscale), _p(mshift), mask,
                                           C.c_long(npix), dy.C, _p(s1), _p(s2), 1 if accumulate_s1 else 0, _p(ws),
                                           _lib.stream_ptr(stream))
    _lib.check(rc, "emd_bn_bwd_reduce_f32")


def bn_backward(dy: Act, r: Act, fold, gamma2, dgamma2, dbeta2, dr: Act, mask=MASK_RELU6, gamma1=None, dgamma1=None,
                eps=BN_EPS, stream=None):
    """Backward of  r -> [BN1] -> BN2 -> mask  given the forward's `fold` dict: dr (may be dy) = d loss / d r; the
    batch-norm parameter gradients are added into dgamma1 / dgamma2 / dbeta2."""
    import torch

    lib = _lib.load()
    Cc = dy.C
    assert (r.B, r.H, r.W, r.C) == (dy.B, dy.H, dy.W, Cc) and (dr.B, dr.H, dr.W, dr.C) == (dy.B, dy.H, dy.W, Cc)
    dev = r.buf.device
    images = int(fold.get("B") or 0)   # per-image statistics: B one-image towers as one batched pass
    assert images in (0, dy.B)
    n = (images or 1) * Cc
    npix = dy.H * dy.W if images else dy.B * dy.H * dy.W
    s1, t = torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev)
    ms, mh = (fold["scale"], fold["shift"]) if mask else (None, None)
    K, m1, m2 = torch.empty_like(s1), torch.empty_like(s1), torch.empty_like(s1)
    ws = torch.empty(max((images or 1) * (lib.emd_chan_reduce_workspace_bytes(npix, Cc) // 8), 1), dtype=torch.float64, device=dev)
    if isinstance(dy, Cout1Grad):
        prep = _prep_struct(fold, gamma1, gamma2, eps, K, m1, m2, dgamma1, dgamma2, dbeta2)
        _lib.check(lib.emd_bn_bwd_reduce_prep_cout1_f32(_p(dy.g1), _p(dy.w9), dy.B, dy.H, dy.W, r.ptr, r.ld, _p(fold["mean"]), _p(fold["rstd1"]),
                                                        _p(ms), _p(mh), mask, 1 if images else 0, Cc, _p(s1), _p(t), _p(ws), C.byref(prep),
                                                        _lib.stream_ptr(stream)), "emd_bn_bwd_reduce_prep_cout1_f32")
        _lib.check(lib.emd_bn_bwd_apply_cout1_f32(_p(dy.g1), _p(dy.w9), dy.B, dy.H, dy.W, r.ptr, r.ld, _p(K), _p(m1), _p(fold["mean"]), _p(m2),
                                                  _p(ms), _p(mh), mask, 1 if images else 0, dr.ptr, dr.ld, Cc, _lib.stream_ptr(stream)),
                   "emd_bn_bwd_apply_cout1_f32")
        return dr
    if FUSE_PREP:   # the per-channel step inside the reduction's final kernel
        prep = _prep_struct(fold, gamma1, gamma2, eps, K, m1, m2, dgamma1, dgamma2, dbeta2)
        _lib.check(lib.emd_bn_bwd_reduce_prep_f32(dy.ptr, dy.ld, r.ptr, r.ld, _p(fold["mean"]), _p(fold["rstd1"]), _p(ms), _p(mh), mask, images,
                                                  C.c_long(npix), Cc, _p(s1), _p(t), _p(ws), C.byref(prep), _lib.stream_ptr(stream)),
                   "emd_bn_bwd_reduce_prep_f32")
    elif images:
        _lib.check(lib.emd_bn_bwd_reduce_images_f32(dy.ptr, dy.ld, r.ptr, r.ld, _p(fold["mean"]), _p(fold["rstd1"]), _p(ms), _p(mh), mask,
                                                    images, C.c_long(npix), Cc, _p(s1), _p(t), _p(ws), _lib.stream_ptr(stream)),
                   "emd_bn_bwd_reduce_images_f32")
        _lib.check(lib.emd_bn_bwd_prep_images_f32(_p(s1), _p(t), _p(gamma1), _p(gamma2), _p(fold["rstd1"]), _p(fold["rstd2"]),
                                                  C.c_float(eps), C.c_long(npix), images, Cc, _p(K), _p(m1), _p(m2), _p(dgamma1), _p(dgamma2),
                                                  _p(dbeta2), _lib.stream_ptr(stream)), "emd_bn_bwd_prep_images_f32")
    else:
        chan_reduce(dy, s1, r, fold["mean"], fold["rstd1"], t, ms, mh, mask, stream=stream)
        _lib.check(lib.emd_bn_bwd_prep_f32(_p(s1), _p(t), _p(gamma1), _p(gamma2), _p(fold["rstd1"]), _p(fold["rstd2"]),
                                           C.c_float(eps), C.c_long(npix), Cc, _p(K), _p(m1), _p(m2), _p(dgamma1), _p(dgamma2),
                                           _p(dbeta2), _lib.stream_ptr(stream)), "emd_bn_bwd_prep_f32")
    if images:
        _lib.check(lib.emd_bn_bwd_apply_images_f32(dy.ptr, dy.ld, r.ptr, r.ld, _p(K), _p(m1), _p(fold["mean"]), _p(m2), _p(ms), _p(mh),
                                                   mask, dr.ptr, dr.ld, images, C.c_long(npix), Cc, _lib.stream_ptr(stream)),
                   "emd_bn_bwd_apply_images_f32")
    else:
        _lib.check(lib.emd_bn_bwd_apply_f32(dy.ptr, dy.ld, r.ptr, r.ld, _p(K), _p(m1), _p(fold["mean"]), _p(m2), _p(ms), _p(mh),
                                            mask, dr.ptr, dr.ld, C.c_long(npix), Cc, _lib.stream_ptr(stream)),
                   "emd_bn_bwd_apply_f32")
    return dr


class Cout1Grad:
    """A gradient that was never written: the data gradient of the 3x3 conv to one output channel (the final conv), dy[p][c] = sum_taps
    g1[p + (1-ky, 1-kx)] * w9[tap][c]; bn_backward forms it from the 1-channel image g1 [B,H,W,1] in both of its passes."""

    __slots__ = ("g1", "w9", "B", "H", "W", "C")

    def __init__(self, g1, w9):
        assert g1.dim() == 4 and g1.shape[3] == 1 and g1.is_contiguous() and w9.is_contiguous() and w9.shape[0] == 9
        self.g1, self.w9 = g1, w9
        self.B, self.H, self.W, self.C = g1.shape[0], g1.shape[1], g1.shape[2], w9.shape[1]


class DwGrad:
    """A gradient that was never written: dy = the data gradient of a separable conv's depthwise stage (dw3x3(dd, w_flipped) at stride 1;
    dw3x3_bwd_data's gather at stride 2 / with dilation), the ONLY contribution to the gradient of its input.  bn_backward forms it on
    the fly (emd_dw3x3_bn_bwd_*_f32)."""

    __slots__ = ("dd", "w", "gdw", "stride", "rate", "B", "H", "W", "C")

    def __init__(self, dd: Act, w_flipped, gdw=None, stride=1, rate=1, hw=None):
        """gdw: the consumer's depthwise weight-gradient slice [9][C] when bn_backward_dw is to add that gradient too (its reduction pass
        reads exactly the operands: dd and the r behind the consumer's never-written input) -- the consumer then skips its own launch."""
        assert w_flipped.is_contiguous() and w_flipped.numel() == 9 * dd.C
        assert gdw is None or (gdw.is_contiguous() and gdw.numel() == 9 * dd.C)
        self.dd, self.w, self.gdw, self.stride, self.rate = dd, w_flipped, gdw, stride, rate
        # shape of the gradient = the consumer's INPUT grid (hw: needed with stride 2, where dd is the smaller OUTPUT grid)
        self.B, self.C = dd.B, dd.C
        self.H, self.W = hw if hw is not None else (dd.H, dd.W)
        assert (dd.H, dd.W) == (-(-self.H // stride), -(-self.W // stride))


def bn_backward_dw(dy: DwGrad, r: Act, fold, gamma2, dgamma2, dbeta2, dr: Act, mask=MASK_RELU6, gamma1=None, dgamma1=None, eps=BN_EPS,
                   stream=None):
    """bn_backward for a gradient given as a DwGrad: reduction and apply each recompute dy from dd (5 passes over the tensor instead of
    7, two launches instead of three); the per-channel step between them is bn_backward's."""
    import torch

    lib = _lib.load()
    dd, Cc = dy.dd, dy.C
    assert (r.B, r.H, r.W, r.C) == (dy.B, dy.H, dy.W, Cc) and (dr.B, dr.H, dr.W, dr.C) == (dy.B, dy.H, dy.W, Cc)
    dev = r.buf.device
    images = int(fold.get("B") or 0)
    assert images in (0, dy.B) and (dy.gdw is None or mask == MASK_RELU6)
    n = (images or 1) * Cc
    npix = dy.H * dy.W if images else dy.B * dy.H * dy.W
    s1, t = torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev)
    ms, mh = (fold["scale"], fold["shift"]) if mask else (None, None)
    ws = torch.empty(max(lib.emd_dw3x3_bn_bwd_workspace_bytes(dy.B, dy.H, dy.W, Cc) // 8, 1), dtype=torch.float64, device=dev)
    K, m1, m2 = torch.empty_like(s1), torch.empty_like(s1), torch.empty_like(s1)
    prep = _prep_struct(fold, gamma1, gamma2, eps, K, m1, m2, dgamma1, dgamma2, dbeta2) if FUSE_PREP else None
    _lib.check(lib.emd_dw3x3_bn_bwd_reduce_f32(dd.ptr, dd.ld, _p(dy.w), r.ptr, r.ld, _p(fold["mean"]), _p(fold["rstd1"]), _p(ms), _p(mh), mask,
                                               1 if images else 0, dy.B, dy.H, dy.W, Cc, dy.stride, dy.rate, _p(s1), _p(t), _p(dy.gdw), _p(ws),
                                               C.byref(prep) if prep is not None else None, _lib.stream_ptr(stream)),
               "emd_dw3x3_bn_bwd_reduce_f32")
    if prep is not None:
        pass
    elif images:
        _lib.check(lib.emd_bn_bwd_prep_images_f32(_p(s1), _p(t), _p(gamma1), _p(gamma2), _p(fold["rstd1"]), _p(fold["rstd2"]),
                                                  C.c_float(eps), C.c_long(npix), images, Cc, _p(K), _p(m1), _p(m2), _p(dgamma1), _p(dgamma2),
                                                  _p(dbeta2), _lib.stream_ptr(stream)), "emd_bn_bwd_prep_images_f32")
    else:
        _lib.check(lib.emd_bn_bwd_prep_f32(_p(s1), _p(t), _p(gamma1), _p(gamma2), _p(fold["rstd1"]), _p(fold["rstd2"]),
                                           C.c_float(eps), C.c_long(npix), Cc, _p(K), _p(m1), _p(m2), _p(dgamma1), _p(dgamma2),
                                           _p(dbeta2), _lib.stream_ptr(stream)), "emd_bn_bwd_prep_f32")
    _lib.check(lib.emd_dw3x3_bn_bwd_apply_f32(dd.ptr, dd.ld, _p(dy.w), r.ptr, r.ld, _p(K), _p(m1), _p(fold["mean"]), _p(m2), _p(ms), _p(mh),
                                              mask, 1 if images else 0, dr.ptr, dr.ld, dy.B, dy.H, dy.W, Cc, dy.stride, dy.rate,
                                              _lib.stream_ptr(stream)),
               "emd_dw3x3_bn_bwd_apply_f32")
    return dr


def dw3x3_wgrad(x: Act, dy: Act, dw_dev, stride=1, rate=1, stream=None):
    assert dw_dev.is_contiguous() and dw_dev.numel() == 9 * x.C and dy.C == x.C
    assert (dy.B, dy.H, dy.W) == (x.B, -(-x.H // stride), -(-x.W // stride))
    _lib.check(_lib.load().emd_dw3x3_wgrad_f32(x.ptr, x.ld, dy.ptr, dy.ld, _p(dw_dev), x.B, x.H, x.W, x.C, stride, rate,
                                               _lib.stream_ptr(stream)), "emd_dw3x3_wgrad_f32")


def dw3x3_bwd_both(dd: Act, w_flipped, x: Act, dx: Act, dw_dev, stream=None):
    """Both gradients of a stride-1 depthwise 3x3 in one pass (emd_dw3x3_bwd_both_f32): dx = dw3x3(dd, w_flipped), dw_dev += wgrad(x, dd)."""
    assert dw_dev.is_contiguous() and dw_dev.numel() == 9 * x.C and w_flipped.is_contiguous() and w_flipped.numel() == 9 * x.C
    assert (dd.B, dd.H, dd.W, dd.C) == (x.B, x.H, x.W, x.C) == (dx.B, dx.H, dx.W, dx.C)
    _lib.check(_lib.load().emd_dw3x3_bwd_both_f32(dd.ptr, dd.ld, _p(w_flipped), x.ptr, x.ld, dx.ptr, dx.ld, _p(dw_dev), x.B, x.H, x.W, x.C,
                                                  _lib.stream_ptr(stream)), "emd_dw3x3_bwd_both_f32")
    return dx


def dw3x3_wgrad_pre(x, dy: Act, dw_dev, stride=1, rate=1, stream=None):
    """dw3x3_wgrad with the layer's input given as an ops.PreAct (never written; rebuilt from its r in the loads)."""
    r = x.r
    assert dw_dev.is_contiguous() and dw_dev.numel() == 9 * r.C and dy.C == r.C
    assert (dy.B, dy.H, dy.W) == (r.B, -(-r.H // stride), -(-r.W // stride))
    _lib.check(_lib.load().emd_dw3x3_wgrad_pre_f32(r.ptr, r.ld, _p(x.scale), _p(x.shift), 1 if x.images else 0, _act(x.act), dy.ptr,
                                                   dy.ld, _p(dw_dev), r.B, r.H, r.W, r.C, stride, rate, _lib.stream_ptr(stream)),
               "emd_dw3x3_wgrad_pre_f32")


def dw3x3_bwd_data(dy: Act, w_dev, dx: Act, stride=1, rate=1, stream=None):
    assert dy.C == dx.C and (dy.B, dy.H, dy.W) == (dx.B, -(-dx.H // stride), -(-dx.W // stride))
    _lib.check(_lib.load().emd_dw3x3_bwd_data_f32(dy.ptr, dy.ld, _p(w_dev), dx.ptr, dx.ld, dx.B, dx.H, dx.W, dx.C, stride,
                                                  rate, _lib.stream_ptr(stream)), "emd_dw3x3_bwd_data_f32")
    return dx


def conv3x3_cout1_wgrad(x: Act, dy_img, dw_dev, stream=None):
    assert dy_img.is_contiguous() and dy_img.numel() == x.B * x.H * x.W and dw_dev.numel() == 9 * x.C
    _lib.check(_lib.load().emd_conv3x3_cout1_wgrad_f32(x.ptr, x.ld, _p(dy_img), _p(dw_dev), x.B, x.H, x.W, x.C,
                                                       _lib.stream_ptr(stream)), "emd_conv3x3_cout1_wgrad_f32")


def conv3x3_cout1_bwd_data(dy_img, w_dev, dx: Act, stream=None):
    assert dy_img.is_contiguous() and dy_img.numel() == dx.B * dx.H * dx.W and w_dev.numel() == 9 * dx.C
    _lib.check(_lib.load().emd_conv3x3_cout1_bwd_data_f32(_p(dy_img), _p(w_dev), dx.ptr, dx.ld, dx.B, dx.H, dx.W, dx.C,
                                                          _lib.stream_ptr(stream)), "emd_conv3x3_cout1_bwd_data_f32")
    return dx


def resize_bilinear_bwd(dy: Act, dx: Act, stream=None):
    assert dy.C == dx.C and dy.B == dx.B
    _lib.check(_lib.load().emd_resize_bilinear_bwd_f32(dy.ptr, dy.ld, dx.ptr, dx.ld, dx.B, dx.H, dx.W, dy.H, dy.W, dx.C,
                                                       _lib.stream_ptr(stream)), "emd_resize_bilinear_bwd_f32")
    return dx


def avgpool2x2_bwd(dy: Act, dx: Act, stream=None):
    assert dy.C == dx.C and (dy.B, dy.H, dy.W) == (dx.B, -(-dx.H // 2), -(-dx.W // 2))
    _lib.check(_lib.load().emd_avgpool2x2_bwd_f32(dy.ptr, dy.ld, dx.ptr, dx.ld, dx.B, dx.H, dx.W, dx.C,
                                                  _lib.stream_ptr(stream)), "emd_avgpool2x2_bwd_f32")
    return dx


def axpy(x: Act, y: Act, alpha=1.0, stream=None):
    assert (x.B, x.H, x.W, x.C) == (y.B, y.H, y.W, y.C)
    _lib.check(_lib.load().emd_axpy_f32(x.ptr, x.ld, y.ptr, y.ld, C.c_long(x.B * x.H * x.W), x.C, C.c_float(alpha),
                                        _lib.stream_ptr(stream)), "emd_axpy_f32")
    return y


def denoise_loss(out, truth, dout=None, grad_scale=1.0, stream=None):
    """-> device tensor [3] = (mse, loss, dloss/dout factor); fills dout (same shape as out) if given."""
    import torch

    lib = _lib.load()
    assert out.is_contiguous() and truth.is_contiguous() and out.numel() == truth.numel()
    res = torch.empty(3, dtype=torch.float32, device=out.device)
    ws = torch.empty(lib.emd_denoise_loss_workspace_bytes() // 8, dtype=torch.float64, device=out.device)
    _lib.check(lib.emd_denoise_loss_f32(_p(out), _p(truth), C.c_long(out.numel()), C.c_float(grad_scale), _p(res), _p(dout),
                                        _p(ws), _lib.stream_ptr(stream)), "emd_denoise_loss_f32")
    return res


def nesterov_step(param, grad, accum, lr, momentum=0.9, grad_scale=1.0, stream=None):
    assert param.is_contiguous() and grad.is_contiguous() and accum.is_contiguous()
    assert param.numel() == grad.numel() == accum.numel()
    _lib.check(_lib.load().emd_nesterov_step_f32(_p(param), _p(grad), _p(accum), C.c_long(param.numel()), C.c_float(lr),
                                                 C.c_float(momentum), C.c_float(grad_scale), _lib.stream_ptr(stream)),
               "emd_nesterov_step_f32")


# ---- graph G training (misc_py/gan-infilling-100.py:982-1088, :1378-1379, :1429-1431)
def gan_head(logit3, label, mode, grad_scale=1.0, stream=None):
    """-> (result2 = [out, loss], dlogit3) device tensors.  mode 0: discriminator loss, 1: generator loss."""
    import torch

    res = torch.empty(2, dtype=torch.float32, device=logit3.device)
    dl = torch.empty(3, dtype=torch.float32, device=logit3.device)
    _lib.check(_lib.load().emd_gan_head_f32(_p(logit3), C.c_float(label), mode, C.c_float(grad_scale), _p(res), _p(dl),
                                            _lib.stream_ptr(stream)), "emd_gan_head_f32")
    return res, dl


def fc_row_bwd(x, w, dlogit, dw, db, stream=None):
    import torch

    dx = torch.empty_like(x)
    _lib.check(_lib.load().emd_fc_row_bwd_f32(_p(x), _p(w), _p(dlogit), _p(dw), _p(db), _p(dx), x.numel(),
                                              _lib.stream_ptr(stream)), "emd_fc_row_bwd_f32")
    return dx


def bcast_rows(v, out: Act, alpha, stream=None):
    _lib.check(_lib.load().emd_bcast_rows_f32(_p(v), out.ptr, out.ld, C.c_long(out.B * out.H * out.W), out.C, C.c_float(alpha),
                                              _lib.stream_ptr(stream)), "emd_bcast_rows_f32")
    return out


def sumsq(x, scale=1.0, stream=None):
    import torch

    lib = _lib.load()
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    ws = torch.empty(lib.emd_sumsq_workspace_bytes() // 8, dtype=torch.float64, device=x.device)
    _lib.check(lib.emd_sumsq_f32(_p(x), C.c_long(x.numel()), C.c_float(scale), _p(out), _p(ws), _lib.stream_ptr(stream)),
               "emd_sumsq_f32")
    return out


def adam_lr_t(lr, step, beta1=0.5, beta2=0.999):
    return lr * (1.0 - beta2 ** step) ** 0.5 / (1.0 - beta1 ** step)


def adam_step(param, grad, m, v, step, lr, beta1=0.5, beta2=0.999, eps=1e-8, grad_scale=1.0, gnorm_sq=None, clip_norm=0.0,
              stream=None, lr_t_dev=None):
    """``step`` = the 1-based step count t (bias correction lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t), as TF computes it)."""
    if lr_t_dev is not None:   # lr_t lives on the device (a replayed hipGraph): the caller refreshes it every step
        _lib.check(_lib.load().emd_adam_step_dev_f32(_p(param), _p(grad), _p(m), _p(v), C.c_long(param.numel()), _p(lr_t_dev),
                                                     C.c_float(beta1), C.c_float(beta2), C.c_float(eps), C.c_float(grad_scale),
                                                     _p(gnorm_sq), C.c_float(clip_norm), _lib.stream_ptr(stream)),
                   "emd_adam_step_dev_f32")
        return
    lr_t = adam_lr_t(lr, step, beta1, beta2)
    _lib.check(_lib.load().emd_adam_step_f32(_p(param), _p(grad), _p(m), _p(v), C.c_long(param.numel()), C.c_float(lr_t),
                                             C.c_float(beta1), C.c_float(beta2), C.c_float(eps), C.c_float(grad_scale),
                                             _p(gnorm_sq), C.c_float(clip_norm), _lib.stream_ptr(stream)), "emd_adam_step_f32")


def dw3x3_reflect_wgrad(x: Act, dy: Act, dw_dev, stride=1, stream=None):
    _lib.check(_lib.load().emd_dw3x3_reflect_wgrad_f32(x.ptr, x.ld, dy.ptr, dy.ld, _p(dw_dev), x.B, x.H, x.W, x.C, stride,
                                                       _lib.stream_ptr(stream)), "emd_dw3x3_reflect_wgrad_f32")


def dw3x3_reflect_bwd_data(dy: Act, w_dev, dx: Act, stride=1, stream=None):
    _lib.check(_lib.load().emd_dw3x3_reflect_bwd_data_f32(dy.ptr, dy.ld, _p(w_dev), dx.ptr, dx.ld, dx.B, dx.H, dx.W, dx.C, stride,
                                                          _lib.stream_ptr(stream)), "emd_dw3x3_reflect_bwd_data_f32")
    return dx


def conv3x3_cout1_reflect_wgrad(x: Act, dy_img, dw_dev, stream=None):
    _lib.check(_lib.load().emd_conv3x3_cout1_reflect_wgrad_f32(x.ptr, x.ld, _p(dy_img), _p(dw_dev), x.B, x.H, x.W, x.C,
                                                               _lib.stream_ptr(stream)), "emd_conv3x3_cout1_reflect_wgrad_f32")


def conv3x3_cout1_reflect_bwd_data(dy_img, w_dev, dx: Act, stream=None):
    _lib.check(_lib.load().emd_conv3x3_cout1_reflect_bwd_data_f32(_p(dy_img), _p(w_dev), dx.ptr, dx.ld, dx.B, dx.H, dx.W, dx.C,
                                                                  _lib.stream_ptr(stream)), "emd_conv3x3_cout1_reflect_bwd_data_f32")
    return dx


def dw7_c1_reflect(x_img, w49, d4: Act, stream=None):
    _lib.check(_lib.load().emd_dw7_c1_reflect_f32(_p(x_img), _p(w49), d4.ptr, d4.B, d4.H, d4.W, _lib.stream_ptr(stream)),
               "emd_dw7_c1_reflect_f32")
    return d4


def dw7_c1_reflect_wgrad(x_img, dd4: Act, dw49, stream=None):
    _lib.check(_lib.load().emd_dw7_c1_reflect_wgrad_f32(_p(x_img), dd4.ptr, _p(dw49), dd4.B, dd4.H, dd4.W,
                                                        _lib.stream_ptr(stream)), "emd_dw7_c1_reflect_wgrad_f32")


def tanh_bwd(dy, y, stream=None):
    import torch

    g = torch.empty_like(dy)
    _lib.check(_lib.load().emd_tanh_bwd_f32(_p(dy), _p(y), _p(g), C.c_long(dy.numel()), _lib.stream_ptr(stream)), "emd_tanh_bwd_f32")
    return g


def l1_feature(a, b, weight, dy, accumulate, loss_acc, stream=None):
    """a, b, dy: contiguous torch tensors of one feature map; loss_acc: device float[1] accumulator."""
    assert a.is_contiguous() and b.is_contiguous() and dy.is_contiguous() and a.numel() == b.numel() == dy.numel()
    _lib.check(_lib.load().emd_l1_feature_f32(_p(a), _p(b), C.c_long(a.numel()), C.c_float(weight), _p(dy), 1 if accumulate else 0,
                                              _p(loss_acc), _lib.stream_ptr(stream)), "emd_l1_feature_f32")


def crop_scatter(dcrop, ldc, dimg, y0, x0, n, S, stream=None, yx_dev=None):
    """yx_dev: int32 device tensor [2] = (y0, x0) read by the kernel (then y0, x0 are ignored)."""
    if yx_dev is not None:
        _lib.check(_lib.load().emd_crop_scatter_dev_f32(_p(dcrop), ldc, _p(dimg), _p(yx_dev), int(n), int(S),
                                                        _lib.stream_ptr(stream)), "emd_crop_scatter_dev_f32")
        return
    _lib.check(_lib.load().emd_crop_scatter_f32(_p(dcrop), ldc, _p(dimg), int(y0), int(x0), int(n), int(S), _lib.stream_ptr(stream)),
               "emd_crop_scatter_f32")


def bn_infer_fold2(g1, b1, m1, v1, g2, b2, m2, v2, eps, stream=None, out=None):
    """-> dict(scale, shift, mprime, rprime, rstd1, a2, mean1) for an inference-mode double batch norm.  ``out``: a dict
    from a previous call, rewritten IN PLACE (a captured hipGraph keeps pointing at the same vectors)."""
    import torch

    if out is None:
        out = {k: torch.empty_like(g1) for k in ("scale", "shift", "mprime", "rprime", "rstd1", "a2")}
    _lib.check(_lib.load().emd_bn_infer_fold2_f32(_p(g1), _p(b1), _p(m1), _p(v1), _p(g2), _p(b2), _p(m2), _p(v2), C.c_float(eps),
                                                  g1.numel(), _p(out["scale"]), _p(out["shift"]), _p(out["mprime"]), _p(out["rprime"]),
                                                  _p(out["rstd1"]), _p(out["a2"]), _lib.stream_ptr(stream)), "emd_bn_infer_fold2_f32")
    out["mean1"] = m1
    return out


def bn_infer_grads(s1, t1, t2, a2, dg1, db1, dg2, db2, stream=None):
    _lib.check(_lib.load().emd_bn_infer_grads_f32(_p(s1), _p(t1), _p(t2), _p(a2), s1.numel(), _p(dg1), _p(db1), _p(dg2), _p(db2),
                                                  _lib.stream_ptr(stream)), "emd_bn_infer_grads_f32")
