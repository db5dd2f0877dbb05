"""Graph K host side: the learned symmetric-kernel ("dedicated kernel") denoiser.

Mirrors the apply-side class of the reference, ``Micrograph_Autoencoder``
(misc_py/apply_kernels+MLPs.py:568-703): same constructor arguments, ``preprocess``,
``denoise_crop`` and ``denoise``; the per-pixel ``sess.run`` loop (:669-698) is replaced by one
launch of ``emd_kernel_denoise_f32`` over the whole image (or batch).  Parameters follow the
reference's make_layer scheme (misc_py/noise-removal-kernels.py:107-358): each w x w map is
built from (o+1)(o+2)/2 scalars shared across D4-symmetric positions.

All arithmetic runs in the HIP library; there is no CPU path here.
"""
from __future__ import annotations

import numpy as np

from . import _lib


def sym_pairs(width: int):
    """(x, y) index pairs in the reference's variable-creation order (``var_x-{x}_y-{y}``)."""
    o = width // 2
    return [(x, y) for x in range(o + 1) for y in range(x + 1)]


def expand_symmetric(vals, width: int) -> np.ndarray:
    """[nsym] scalars -> full [w,w] D4-symmetric float32 map (make_layer, :107-358)."""
    o = width // 2
    lut = {p: k for k, p in enumerate(sym_pairs(width))}
    vals = np.asarray(vals, dtype=np.float32)
    if vals.shape != (len(lut),):
        raise ValueError(f"expected {len(lut)} scalars for width {width}, got shape {vals.shape}")
    full = np.empty((width, width), dtype=np.float32)
    for i in range(width):
        for j in range(width):
            a, b = abs(i - o), abs(j - o)
            full[i, j] = vals[lut[(max(a, b), min(a, b))]]
    return full


def is_d4_symmetric(m: np.ndarray) -> bool:
    return bool(np.array_equal(m, m.T) and np.array_equal(m, m[::-1, :]) and np.array_equal(m, m[:, ::-1]))


class KernelParams:
    """Full weight / bias maps of one (depth, width) filter, as the HIP kernel consumes them."""

    def __init__(self, wmaps: np.ndarray, bmaps: np.ndarray, s: np.ndarray):
        wmaps = np.ascontiguousarray(wmaps, dtype=np.float32)
        bmaps = np.ascontiguousarray(bmaps, dtype=np.float32)
        s = np.ascontiguousarray(s, dtype=np.float32)
        if wmaps.ndim != 3 or wmaps.shape[1] != wmaps.shape[2] or wmaps.shape != bmaps.shape or s.shape != (wmaps.shape[0],):
            raise ValueError("wmaps/bmaps must be [depth,w,w] and s [depth]")
        self.depth, self.width = int(wmaps.shape[0]), int(wmaps.shape[1])
        self.wmaps, self.bmaps, self.s = wmaps, bmaps, s
        self.symmetric = all(is_d4_symmetric(wmaps[i]) for i in range(self.depth)) and all(
            is_d4_symmetric(bmaps[i]) for i in range(1, self.depth))

    @classmethod
    def initial(cls, depth: int, width: int):
        """The reference's initial values: weights 1/w^2, biases 0 (noise-removal-kernels.py:109-112)."""
        w = np.full((depth, width, width), 1.0 / (width * width), dtype=np.float32)
        return cls(w, np.zeros_like(w), np.ones(depth, dtype=np.float32))

    @classmethod
    def from_symmetric(cls, w_scalars, b_scalars, s, width: int):
        """w_scalars / b_scalars: per layer, the (o+1)(o+2)/2 scalars in creation order."""
        depth = len(w_scalars)
        wm = np.stack([expand_symmetric(w_scalars[i], width) for i in range(depth)])
        bm = np.stack([expand_symmetric(b_scalars[i], width) for i in range(depth)])
        return cls(wm, bm, np.asarray(s, dtype=np.float32))

    def packed(self) -> np.ndarray:
        """The flat parameter block of emd_kernel_denoise_f32: wmaps, bmaps, s."""
        return np.concatenate([self.wmaps.ravel(), self.bmaps.ravel(), self.s.ravel()]).astype(np.float32)


def kernel_denoise(x, params_dev, width: int, depth: int, symmetric: bool, out=None, stream=None):
    """x: torch CUDA float32 [B,H,W,1] or [B,H,W] (contiguous) -> same shape.  One launch."""
    import torch

    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()):
        raise ValueError("kernel_denoise needs a contiguous float32 CUDA tensor")
    if x.dim() == 4:
        if x.shape[3] != 1:
            raise ValueError("channel dimension must be 1")
        B, H, W = x.shape[0], x.shape[1], x.shape[2]
    elif x.dim() == 3:
        B, H, W = x.shape
    else:
        raise ValueError("expected [B,H,W,1] or [B,H,W]")
    if out is None:
        out = torch.empty_like(x)
    lib = _lib.load()
    n = lib.emd_kernel_params_count(width, depth)
    if params_dev.numel() != n or params_dev.dtype != torch.float32 or not params_dev.is_cuda:
        raise ValueError(f"params block must be a float32 CUDA tensor of {n} elements")
    rc = lib.emd_kernel_denoise_f32(_lib.ptr(x), _lib.ptr(out), B, H, W, width, depth, _lib.ptr(params_dev),
                                    _lib.EMD_K_SYMMETRIC if symmetric else 0, _lib.stream_ptr(stream))
    _lib.check(rc, "emd_kernel_denoise_f32")
    return out


class Micrograph_Autoencoder(object):
    """Drop-in for the reference class of the same name (apply_kernels+MLPs.py:568-703)."""

    def __init__(self, ckpt_loc=None, visible_cuda=None, depth=1, width=3, params: KernelParams | None = None):
        import torch

        if width < 1 or width % 2 == 0:
            raise ValueError("width must be odd")
        self.cropsize = width
        self.depth, self.width = depth, width
        if params is None:
            if ckpt_loc is not None:
                params = load_kernel_params(ckpt_loc, depth, width)
            else:
                params = KernelParams.initial(depth, width)
        if params.depth != depth or params.width != width:
            raise ValueError("params do not match depth/width")
        self.params = params
        # reference: os.environ["CUDA_VISIBLE_DEVICES"] = visible_cuda (:585-586); here: device index
        idx = int(str(visible_cuda).split(",")[0]) if visible_cuda not in (None, "") else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        _lib.load()
        self._params_dev = torch.from_numpy(params.packed()).to(self.device)

    # ---- apply_kernels+MLPs.py:611-622
    def preprocess(self, img, pad_width=0):
        img = np.array(img, dtype=np.float32, copy=True)
        img[np.isnan(img)] = 0.0
        img[np.isinf(img)] = 0.0
        img = np.pad(img, pad_width=pad_width, mode="reflect").reshape(
            img.shape[0] + 2 * pad_width, img.shape[1] + 2 * pad_width, 1)
        return img.astype(np.float32)

    def _run(self, batch_dev):
        return kernel_denoise(batch_dev, self._params_dev, self.width, self.depth, self.params.symmetric)

    # ---- apply_kernels+MLPs.py:624-636: a (cropsize,cropsize) crop -> the one pixel at its centre
    def denoise_crop(self, crop):
        import torch

        c = self.preprocess(np.asarray(crop).reshape(np.asarray(crop).shape[0], -1))
        h, w = c.shape[0], c.shape[1]
        o = self.width // 2
        if h < self.width or w < self.width:
            raise ValueError("crop smaller than the kernel")
        x = torch.from_numpy(c.reshape(1, h, w, 1)).to(self.device)
        y = self._run(x)[0, :, :, 0]
        # VALID region: outputs whose whole window lies inside the crop (no padding on this path, :420)
        return y[o: h - o, o: w - o].cpu().numpy()

    def denoise_batch(self, lq_batch):
        """lq_batch [B,H,W,1] float32 (numpy or torch) -> hq_batch, same container and shape.
        REFLECT-padded SAME filtering (the trainer's convention, noise-removal-kernels.py:405)."""
        import torch

        is_np = isinstance(lq_batch, np.ndarray)
        x = torch.from_numpy(np.ascontiguousarray(lq_batch, dtype=np.float32)) if is_np else lq_batch
        dev_in = x.is_cuda
        xd = x.to(self.device, dtype=torch.float32).contiguous()
        y = self._run(xd)
        if is_np:
            return y.cpu().numpy()
        return y if dev_in else y.cpu()

    # ---- apply_kernels+MLPs.py:638-703
    def denoise(self, img, preprocess=True, postprocess=True, used_overlap=1):
        """Whole-image filtering.  The reference reflect-pads by w//2, rescales by the PADDED image's
        (min, mean-min), evaluates the filter at every pixel (one sess.run each) and undoes the
        scaling; that is one SAME/REFLECT launch here."""
        import torch

        img = np.asarray(img, dtype=np.float32)
        scale = offset = None
        if preprocess:
            p = self.width // 2
            padded = self.preprocess(img, pad_width=p)
            offset = float(np.min(padded))
            if float(np.max(padded)) == offset:
                padded.fill(1.0)
            else:
                scale = float(np.mean(padded)) - offset
                padded = (padded - offset) / scale
            core = padded[p: padded.shape[0] - p, p: padded.shape[1] - p, 0]
        else:
            core = img.reshape(img.shape[0], img.shape[1])
        x = torch.from_numpy(np.ascontiguousarray(core)[None, :, :, None]).to(self.device)
        den = self._run(x)[0, :, :, 0].cpu().numpy().astype(np.float64)
        if postprocess and preprocess:
            den = den * scale + offset if scale else den * offset
        return den


def load_kernel_params(ckpt_loc, depth: int, width: int) -> KernelParams:
    """Load filter scalars saved as ``<ckpt_loc>/kernel_params_depth-{d}_size-{w}.npz`` with the
    TF variable names as keys (``depth-{d}_size-{w}/w0/var_x-{x}_y-{y}/v`` ...,
    ``depth-{d}_size-{w}/fully_connected[_k]/weights``).  Reading TensorFlow checkpoint bundles
    directly is a later step."""
    import os

    path = os.path.join(ckpt_loc, f"kernel_params_depth-{depth}_size-{width}.npz")
    z = np.load(path, allow_pickle=False)
    scope = f"depth-{depth}_size-{width}"
    pairs = sym_pairs(width)

    def layer(kind, i):
        return [float(np.asarray(z[f"{scope}/{kind}{i}/var_x-{x}_y-{y}/v"]).reshape(-1)[0]) for (x, y) in pairs]

    ws = [layer("w", i) for i in range(depth)]
    bs = [[0.0] * len(pairs)] + [layer("b", i) for i in range(1, depth)]
    s = [1.0]
    for i in range(1, depth):
        key = f"{scope}/fully_connected/weights" if i == 1 else f"{scope}/fully_connected_{i - 1}/weights"
        s.append(float(np.asarray(z[key]).reshape(-1)[0]))
    return KernelParams.from_symmetric(ws, bs, s, width)
