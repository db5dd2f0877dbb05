"""TensorFlow-1.x op semantics restated on the CPU (oracle; test infrastructure only).

The reference's hot path is a graph of stock ``tf.*`` / ``slim.*`` calls
(machine_learning/denoiser.py:71-229); TensorFlow itself is a third-party
dependency that is neither vendored under /root/reference nor pinned there
(no requirements file; API usage bounds it to about TF 1.8-1.12), so each op's
published semantics is restated here.  PARITY UNPINNED (see oracle/__init__.py).

Every op exists twice:

* ``*_t``  : PyTorch-CPU (``torch.nn.functional``) with TF padding made
             explicit; runs in float32 or float64; this is the oracle proper.
* ``*_np`` : plain numpy from the index formulas, no torch; slow, used by the
             tests to cross-check ``*_t`` at small shapes.

All tensors are NHWC like the reference (data_format='NHWC', denoiser.py:120).
Weight layouts are TensorFlow's:
  depthwise [kh,kw,C,1], pointwise [1,1,Cin,Cout], conv [kh,kw,Cin,Cout],
  conv2d_transpose [kh,kw,Cout,Cin].
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3  # tf.contrib.layers.batch_norm default epsilon (denoiser.py:71-79 uses defaults)


# ----------------------------------------------------------------------------------------------
# padding arithmetic
# ----------------------------------------------------------------------------------------------
def same_pads(n: int, k: int, s: int = 1, r: int = 1):
    """TF 'SAME': out=ceil(n/s); pad_total=max((out-1)*s+(k-1)*r+1-n,0); before=total//2."""
    out = -(-n // s)
    eff = (k - 1) * r + 1
    total = max((out - 1) * s + eff - n, 0)
    return out, total // 2, total - total // 2


def _nchw(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 3, 1, 2)


def _nhwc(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 2, 3, 1).contiguous()


def _as_t(a, dtype):
    if isinstance(a, torch.Tensor):
        return a.to(dtype)
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


# ----------------------------------------------------------------------------------------------
# torch forms
# ----------------------------------------------------------------------------------------------
def depthwise_conv2d_t(x, w, stride=1, rate=1):
    """tf.nn.depthwise_conv2d(SAME) as used inside slim.separable_convolution2d
    (denoiser.py:113-131).  x [B,H,W,C]; w [kh,kw,C,1]; cross-correlation."""
    B, H, W, C = x.shape
    kh, kw = w.shape[0], w.shape[1]
    _, pt, pb = same_pads(H, kh, stride, rate)
    _, pl, pr = same_pads(W, kw, stride, rate)
    xn = F.pad(_nchw(x), (pl, pr, pt, pb))
    wt = w.permute(2, 3, 0, 1).contiguous()  # [C,1,kh,kw]
    y = F.conv2d(xn, wt, None, stride=stride, dilation=rate, groups=C)
    return _nhwc(y)


def conv2d_t(x, w, bias=None, stride=1, rate=1):
    """slim.conv2d(padding='SAME') (denoiser.py:91-96, :159, :208, :220).
    x [B,H,W,Cin]; w [kh,kw,Cin,Cout]; cross-correlation; optional bias [Cout]."""
    B, H, W, C = x.shape
    kh, kw = w.shape[0], w.shape[1]
    _, pt, pb = same_pads(H, kh, stride, rate)
    _, pl, pr = same_pads(W, kw, stride, rate)
    xn = F.pad(_nchw(x), (pl, pr, pt, pb))
    wt = w.permute(3, 2, 0, 1).contiguous()  # [Cout,Cin,kh,kw] (contiguous: torch's CPU conv backward requires it)
    y = F.conv2d(xn, wt, bias, stride=stride, dilation=rate)
    return _nhwc(y)


def conv2d_transpose_s2_t(x, w, bias=None):
    """slim.conv2d_transpose(kernel_size=3, stride=2, padding='same') (denoiser.py:141-147).

    It is the gradient of the SAME stride-2 3x3 convolution from a 2N input to an N output,
    whose padding is 0 before / 1 after, hence  y[j] = sum_{i,k: 2i+k=j} x[i] w[k]  for j in
    [0,2N): the full (2N+1)-long transposed convolution cropped at the END.
    x [B,H,W,Cin]; w [3,3,Cout,Cin]; output [B,2H,2W,Cout]."""
    B, H, W, C = x.shape
    wt = w.permute(3, 2, 0, 1).contiguous()  # conv_transpose2d wants [Cin,Cout,kh,kw]
    y = F.conv_transpose2d(_nchw(x), wt, bias, stride=2, padding=0)
    return _nhwc(y[:, :, : 2 * H, : 2 * W])


def batch_norm_inference_t(x, gamma, beta, mean, var, eps=BN_EPS):
    """tf.contrib.layers.batch_norm(is_training=False) (denoiser.py:71-79)."""
    return (x - mean) * (gamma / torch.sqrt(var + eps)) + beta


def relu6_t(x):
    return torch.clamp(x, 0.0, 6.0)


def resize_bilinear_legacy_t(x, oh, ow):
    """tf.image.resize_images(x,[oh,ow]) = bilinear, align_corners=False, no half-pixel
    offset (denoiser.py:199, :350).  src = dst*(in/out) in float32; lo=floor; hi=min(lo+1,in-1);
    top/bottom lerped along W first, then along H (TF resize_bilinear kernel order)."""
    B, H, W, C = x.shape

    def axis(n_in, n_out):
        scale = np.float32(n_in) / np.float32(n_out)
        src = np.arange(n_out, dtype=np.float32) * scale
        lo = np.floor(src).astype(np.int64)
        hi = np.minimum(lo + 1, n_in - 1)
        return torch.from_numpy(lo), torch.from_numpy(hi), torch.from_numpy(src - lo.astype(np.float32)).to(x.dtype)

    ylo, yhi, yl = axis(H, oh)
    xlo, xhi, xl = axis(W, ow)
    xl = xl.view(1, 1, ow, 1)
    yl = yl.view(1, oh, 1, 1)
    top_rows = x[:, ylo]
    bot_rows = x[:, yhi]
    top = top_rows[:, :, xlo] + (top_rows[:, :, xhi] - top_rows[:, :, xlo]) * xl
    bot = bot_rows[:, :, xlo] + (bot_rows[:, :, xhi] - bot_rows[:, :, xlo]) * xl
    return top + (bot - top) * yl


def avg_pool2x2_same_t(x):
    """tf.nn.pool(window (2,2), 'AVG', 'SAME', strides (2,2)) (denoiser-multi-gpu.py:331-335): mean over the
    in-image samples of each window (padding does not count)."""
    xn = _nchw(x)
    H, W = xn.shape[2], xn.shape[3]
    return _nhwc(F.avg_pool2d(xn, 2, 2, padding=0, ceil_mode=True, count_include_pad=False) if (H % 2 or W % 2)
                 else F.avg_pool2d(xn, 2, 2))


def reflect_pad_t(x, p):
    """tf.pad(mode='REFLECT') on H and W (noise-removal-kernels.py:99-105): mirror without
    repeating the border sample."""
    if p == 0:
        return x
    return _nhwc(F.pad(_nchw(x), (p, p, p, p), mode="reflect"))


# ----------------------------------------------------------------------------------------------
# numpy forms (independent of torch; written from the index formulas)
# ----------------------------------------------------------------------------------------------
def _pad_hw_np(x, pt, pb, pl, pr):
    return np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))


def depthwise_conv2d_np(x, w, stride=1, rate=1):
    B, H, W, C = x.shape
    kh, kw = w.shape[:2]
    oh, pt, pb = same_pads(H, kh, stride, rate)
    ow, pl, pr = same_pads(W, kw, stride, rate)
    xp = _pad_hw_np(x, pt, pb, pl, pr)
    y = np.zeros((B, oh, ow, C), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            ys = i * rate
            xs = j * rate
            patch = xp[:, ys: ys + (oh - 1) * stride + 1: stride, xs: xs + (ow - 1) * stride + 1: stride, :]
            y += patch * w[i, j, :, 0]
    return y


def conv2d_np(x, w, bias=None, stride=1, rate=1):
    B, H, W, C = x.shape
    kh, kw, _, co = w.shape
    oh, pt, pb = same_pads(H, kh, stride, rate)
    ow, pl, pr = same_pads(W, kw, stride, rate)
    xp = _pad_hw_np(x, pt, pb, pl, pr)
    y = np.zeros((B, oh, ow, co), dtype=x.dtype)
    for i in range(kh):
        for j in range(kw):
            ys = i * rate
            xs = j * rate
            patch = xp[:, ys: ys + (oh - 1) * stride + 1: stride, xs: xs + (ow - 1) * stride + 1: stride, :]
            y += patch @ w[i, j]
    if bias is not None:
        y += bias
    return y


def conv2d_transpose_s2_np(x, w, bias=None):
    """Scatter form: y[2i+k] += x[i] * w[k], cropped to [0, 2N)."""
    B, H, W, C = x.shape
    co = w.shape[2]
    full = np.zeros((B, 2 * H + 1, 2 * W + 1, co), dtype=x.dtype)
    for i in range(3):
        for j in range(3):
            full[:, i: i + 2 * H: 2, j: j + 2 * W: 2, :] += x @ w[i, j].T  # w[i,j] is [Cout,Cin]
    y = full[:, : 2 * H, : 2 * W, :]
    if bias is not None:
        y = y + bias
    return y


def batch_norm_inference_np(x, gamma, beta, mean, var, eps=BN_EPS):
    return (x - mean) * (gamma / np.sqrt(var + np.asarray(eps, dtype=x.dtype))) + beta


def relu6_np(x):
    return np.minimum(np.maximum(x, 0), 6)


def resize_bilinear_legacy_np(x, oh, ow):
    B, H, W, C = x.shape
    y = np.empty((B, oh, ow, C), dtype=x.dtype)
    sy = np.float32(H) / np.float32(oh)
    sx = np.float32(W) / np.float32(ow)
    for i in range(oh):
        fy = np.float32(i) * sy
        y0 = int(np.floor(fy))
        y1 = min(y0 + 1, H - 1)
        ly = x.dtype.type(fy - np.float32(y0))
        for j in range(ow):
            fx = np.float32(j) * sx
            x0 = int(np.floor(fx))
            x1 = min(x0 + 1, W - 1)
            lx = x.dtype.type(fx - np.float32(x0))
            top = x[:, y0, x0] + (x[:, y0, x1] - x[:, y0, x0]) * lx
            bot = x[:, y1, x0] + (x[:, y1, x1] - x[:, y1, x0]) * lx
            y[:, i, j] = top + (bot - top) * ly
    return y


def avg_pool2x2_same_np(x):
    B, H, W, C = x.shape
    oh, ow = -(-H // 2), -(-W // 2)
    y = np.zeros((B, oh, ow, C), dtype=x.dtype)
    for i in range(oh):
        for j in range(ow):
            y[:, i, j] = x[:, 2 * i: min(2 * i + 2, H), 2 * j: min(2 * j + 2, W)].mean(axis=(1, 2))
    return y


def reflect_pad_np(x, p):
    if p == 0:
        return x
    return np.pad(x, ((0, 0), (p, p), (p, p), (0, 0)), mode="reflect")


def reflect_index(i: int, n: int) -> int:
    """Index into [0,n) that REFLECT padding reads for position i in [-(n-1), 2n-2]."""
    if i < 0:
        return -i
    if i >= n:
        return 2 * n - 2 - i
    return i
