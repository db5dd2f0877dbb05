"""Graph D oracle: the inference denoiser (oracle; test infrastructure only).

Restates ``architecture()`` of machine_learning/denoiser.py:58-398 with the TF op semantics of
oracle/tf_ops.py, on PyTorch-CPU tensors (float32, or float64 to measure the oracle's own
rounding).  PARITY UNPINNED (oracle/__init__.py).

Variables are fetched by their TensorFlow names, generated in graph-construction order under
``tf.variable_scope('nn')`` (denoiser.py:514): slim layers take the default scopes
``SeparableConv2d``/``Conv``/``Conv2d_transpose`` and ``tf.contrib.layers.batch_norm`` takes
``BatchNorm``, each uniquified with ``_1, _2, ...`` inside its parent scope.

``cropsize`` is 512 in the reference (denoiser.py:54) with ``aspp_size = 32 = cropsize/16``
(:45); the oracle keeps that ratio so that the same graph can be checked at small crops.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np
import torch

from . import tf_ops as T

# denoiser.py:38-52
features0, features1, features2, features3, features4 = 64, 128, 256, 728, 728
aspp_filters = features4
aspp_output = 256
aspp_rateSmall, aspp_rateMedium, aspp_rateLarge = 6, 12, 18
num_extra_blocks = 11
channels = 1
BN_DECAY = 0.999  # tf.contrib.layers.batch_norm default


class _Names:
    """tf.variable_scope default-name uniquifier (one counter per parent scope)."""

    def __init__(self, prefix):
        self.prefix = prefix
        self.counts = {}

    def unique(self, base):
        n = self.counts.get(base, 0)
        self.counts[base] = n + 1
        return f"{self.prefix}/{base}" if n == 0 else f"{self.prefix}/{base}_{n}"


class _Graph:
    """variant "D": machine_learning/denoiser.py:58-398.  variant "Dprime": its training twin,
    misc_py/denoiser-multi-gpu.py:200-540, run with phase=False (inference batch norm): tf.layers convs
    (scopes conv2d_k / conv2d_transpose_k, variables kernel/bias; named ASPP convs), dense dilated 3x3 ASPP
    branches (:306-328), a real image-level branch (:331-345) and an in-graph clip to [0,1] (:534-538)."""

    def __init__(self, get_var, dtype, variant="D"):
        assert variant in ("D", "Dprime")
        self.twin = variant == "Dprime"
        self.get = get_var
        self.names = _Names("nn")
        self.dtype = dtype
        self.trace = None  # optional list of tensors (post-activation) for per-layer statistics
        self.calibrate = None  # optional dict: batch statistics of every BN input are written here AND used
        self.training = False  # phase=True (denoiser-multi-gpu.py:214): batch statistics, differentiable
        self.moving_updates = None  # training: dict scope/moving_* -> updated value (decay 0.999, unbiased variance)
        self.saved = None  # optional dict scope -> {"d": depthwise output, "r": conv / pointwise / transposed-conv output (bias included)}

    # ---- denoiser.py:71-84
    def _batch_norm_fn(self, x, scope=None):
        scope = scope or self.names.unique("BatchNorm")
        C = x.shape[-1]
        beta = self.get(scope + "/beta", (C,))
        gamma = self.get(scope + "/gamma", (C,))
        if self.training:
            # tf.contrib.layers.batch_norm(is_training=True, fused=True): normalise with the batch mean and BIASED
            # variance; the moving statistics move by (1-decay) towards the batch mean / UNBIASED variance
            mean = x.mean(dim=(0, 1, 2))
            var = x.var(dim=(0, 1, 2), unbiased=False)
            if self.moving_updates is not None:
                n = x.shape[0] * x.shape[1] * x.shape[2]
                mm = self.get(scope + "/moving_mean", (C,)).detach()
                mv = self.get(scope + "/moving_variance", (C,)).detach()
                self.moving_updates[scope + "/moving_mean"] = mm - (mm - mean.detach()) * (1.0 - BN_DECAY)
                self.moving_updates[scope + "/moving_variance"] = mv - (mv - var.detach() * (n / max(n - 1, 1))) * (1.0 - BN_DECAY)
            return (x - mean) / torch.sqrt(var + T.BN_EPS) * gamma + beta
        if self.calibrate is not None:
            # data-dependent initialisation of the moving statistics (used only to SYNTHESISE weights:
            # tests/golden/make_synth_bn.py); biased variance, like TF's fused batch norm normalisation
            mean = x.mean(dim=(0, 1, 2))
            var = x.var(dim=(0, 1, 2), unbiased=False)
            self.calibrate[scope + "/moving_mean"] = mean.to(torch.float32).numpy().copy()
            self.calibrate[scope + "/moving_variance"] = var.to(torch.float32).numpy().copy()
            mean, var = mean.to(torch.float32).to(self.dtype), var.to(torch.float32).to(self.dtype)
        else:
            mean = self.get(scope + "/moving_mean", (C,))
            var = self.get(scope + "/moving_variance", (C,))
        return T.batch_norm_inference_t(x, gamma, beta, mean, var)

    def batch_then_activ(self, x):
        y = T.relu6_t(self._batch_norm_fn(x))
        if self.trace is not None:
            self.trace.append(y)
        return y

    # ---- denoiser.py:86-99
    def _conv(self, x, filters, kernel_size, stride=1, rate=1, name=None):
        """slim.conv2d (D, :91-96) / tf.layers.conv2d (twin, denoiser-multi-gpu.py:231-235): conv + bias."""
        if self.twin:
            scope = self.names.prefix + "/" + name if name else self.names.unique("conv2d")
            wn, bn = "/kernel", "/bias"
        else:
            scope = self.names.unique("Conv")
            wn, bn = "/weights", "/biases"
        w = self.get(scope + wn, (kernel_size, kernel_size, x.shape[-1], filters))
        b = self.get(scope + bn, (filters,))
        y = T.conv2d_t(x, w, b, stride=stride, rate=rate)
        if self.saved is not None:
            self.saved[scope] = {"r": y.detach()}
        return y

    def conv_block_not_sep(self, x, filters, kernel_size=3, stride=1):
        return self.batch_then_activ(self._conv(x, filters, kernel_size, stride))

    # ---- denoiser.py:110-136 (slim.separable_convolution2d: stride and rate act on the
    #      depthwise stage; no bias because normalizer_fn is set; normalizer BN then outer BN+relu6)
    def strided_conv_block(self, x, filters, stride, rate=1):
        scope = self.names.unique("SeparableConv2d")
        cin = x.shape[-1]
        dw = self.get(scope + "/depthwise_weights", (3, 3, cin, 1))
        pw = self.get(scope + "/pointwise_weights", (1, 1, cin, filters))
        d = T.depthwise_conv2d_t(x, dw, stride=stride, rate=rate)
        y = T.conv2d_t(d, pw, None)
        if self.saved is not None:
            self.saved[scope] = {"d": d.detach(), "r": y.detach()}
        y = self._batch_norm_fn(y, scope + "/BatchNorm")
        return self.batch_then_activ(y)

    def conv_block(self, x, filters):
        return self.strided_conv_block(x, filters, 1, 1)

    # ---- denoiser.py:138-150
    def deconv_block(self, x, filters):
        scope = self.names.unique("conv2d_transpose" if self.twin else "Conv2d_transpose")
        w = self.get(scope + ("/kernel" if self.twin else "/weights"), (3, 3, filters, x.shape[-1]))
        b = self.get(scope + ("/bias" if self.twin else "/biases"), (filters,))
        y = T.conv2d_transpose_s2_t(x, w, b)
        if self.saved is not None:
            self.saved[scope] = {"r": y.detach()}
        return self.batch_then_activ(y)

    # ---- denoiser.py:152-216
    def aspp_block_twin(self, x, aspp_size):
        """denoiser-multi-gpu.py:291-361."""
        conv1x1 = self.batch_then_activ(self._conv(x, aspp_filters, 1, name="1x1"))
        small = self.batch_then_activ(self._conv(x, aspp_filters, 3, rate=aspp_rateSmall, name="lowRate"))
        medium = self.batch_then_activ(self._conv(x, aspp_filters, 3, rate=aspp_rateMedium, name="mediumRate"))
        large = self.batch_then_activ(self._conv(x, aspp_filters, 3, rate=aspp_rateLarge, name="highRate"))
        pooling = T.avg_pool2x2_same_t(x)
        pooling = self._conv(pooling, aspp_filters, 1, name="imageLevel")
        pooling = T.resize_bilinear_legacy_t(pooling, aspp_size, aspp_size)
        pooling = self.batch_then_activ(pooling)
        cat = torch.cat([conv1x1, small, medium, large, pooling], dim=3)
        return self.batch_then_activ(self._conv(cat, aspp_output, 1, name="pellet"))

    def aspp_block(self, x, aspp_size):
        if self.twin:
            return self.aspp_block_twin(x, aspp_size)
        conv1x1 = self.conv_block_not_sep(x, aspp_filters, 1)
        small = self.batch_then_activ(self.strided_conv_block(x, aspp_filters, 1, aspp_rateSmall))
        medium = self.batch_then_activ(self.strided_conv_block(x, aspp_filters, 1, aspp_rateMedium))
        large = self.batch_then_activ(self.strided_conv_block(x, aspp_filters, 1, aspp_rateLarge))
        # :185-189 tf.nn.pool result is discarded; :199 replaces it by a resize of the INPUT
        pooling = T.resize_bilinear_legacy_t(x, aspp_size, aspp_size)
        pooling = self.batch_then_activ(pooling)
        cat = torch.cat([conv1x1, small, medium, large, pooling], dim=3)
        return self.conv_block_not_sep(cat, aspp_output, 1)

    # ---- denoiser.py:218-229
    def residual_conv(self, x, filters):
        return self.conv_block_not_sep(x, filters, 1, stride=2)

    # ---- denoiser.py:231-246
    def xception_middle_block(self, x, features):
        m = self.strided_conv_block(x, features, 1)
        m = self.strided_conv_block(m, features, 1)
        m = self.strided_conv_block(m, features, 1)
        return m + x

    # ---- denoiser.py:248-398
    def build(self, inputs, cropsize):
        aspp_size = cropsize // 16
        x = inputs.reshape(-1, cropsize, cropsize, channels)

        cnn0 = self.conv_block(x, features0)
        cnn0_last = self.conv_block(cnn0, features0)
        cnn0_strided = self.strided_conv_block(cnn0_last, features1, 2)
        cnn0_strided = cnn0_strided + self.residual_conv(x, features1)

        cnn1 = self.conv_block(cnn0_strided, features1)
        cnn1_last = self.conv_block(cnn1, features1)
        cnn1_strided = self.strided_conv_block(cnn1_last, features1, 2)
        cnn1_strided = cnn1_strided + self.residual_conv(cnn0_strided, features1)

        cnn2 = self.conv_block(cnn1_strided, features2)
        cnn2_last = self.conv_block(cnn2, features2)
        cnn2_strided = self.strided_conv_block(cnn2_last, features2, 2)
        cnn2_strided = cnn2_strided + self.residual_conv(cnn1_strided, features2)

        cnn3 = self.conv_block(cnn2_strided, features3)
        cnn3_last = self.conv_block(cnn3, features3)
        cnn3_strided = self.strided_conv_block(cnn3_last, features3, 2)
        cnn3_strided = cnn3_strided + self.residual_conv(cnn2_strided, features3)

        cnn4 = self.conv_block(cnn3_strided, features4)
        cnn4 = self.conv_block(cnn4, features4)
        cnn4_last = self.conv_block(cnn4, features4)
        cnn4_last = cnn4_last + cnn3_strided

        for _ in range(num_extra_blocks):
            cnn4_last = self.xception_middle_block(cnn4_last, features4)

        aspp = self.aspp_block(cnn4_last, aspp_size)

        deconv3 = T.resize_bilinear_legacy_t(aspp, aspp_size * 4, aspp_size * 4)

        concat2 = torch.cat([deconv3, cnn1_strided], dim=3)
        deconv2 = self.conv_block(concat2, features2)
        deconv2 = self.conv_block(deconv2, features2)
        deconv2 = deconv2 + self.conv_block_not_sep(concat2, features2, 1)

        deconv2to1 = self.deconv_block(deconv2, features2)

        concat1 = torch.cat([deconv2to1, cnn0_strided], dim=3)
        deconv1 = self.conv_block(concat1, features1)
        deconv1 = self.conv_block(deconv1, features1)
        deconv1 = deconv1 + self.conv_block_not_sep(concat1, features1, 1)

        deconv1to0 = self.deconv_block(deconv1, features1)

        deconv0 = self.conv_block(deconv1to0, features0)
        deconv0 = self.conv_block(deconv0, features0)
        deconv0 = deconv0 + self.conv_block_not_sep(deconv1to0, features0, 1)

        # :387 "1x1" in the comment, but kernel_size defaults to 3
        out = self.conv_block_not_sep(deconv0, 1)
        if self.twin:  # denoiser-multi-gpu.py:534-538
            out = torch.clamp(out, 0.0, 1.0)
        return out


def variable_specs(cropsize=32, variant="D") -> "OrderedDict[str, tuple]":
    """Names and shapes of every variable, in creation order."""
    specs = OrderedDict()

    def rec(name, shape):
        specs[name] = tuple(int(s) for s in shape)
        return torch.zeros(shape, dtype=torch.float32)

    g = _Graph(rec, torch.float32, variant)
    with torch.no_grad():
        g.build(torch.zeros(1, cropsize, cropsize, 1), cropsize)
    return specs


def architecture(inputs, weights, cropsize=512, dtype=torch.float32, trace=None, calibrate=None, variant="D"):
    """inputs [B,cropsize,cropsize,1] (numpy or torch) -> torch tensor [B,cropsize,cropsize,1].
    ``weights``: dict TF-name -> numpy array.  No output clip (denoiser.py:396)."""
    cache = {}

    def get(name, shape):
        if name not in cache:
            w = weights[name]
            assert tuple(w.shape) == tuple(shape), (name, w.shape, shape)
            cache[name] = torch.from_numpy(np.ascontiguousarray(w)).to(dtype)
        return cache[name]

    g = _Graph(get, dtype, variant)
    g.trace = trace
    g.calibrate = calibrate
    x = inputs if isinstance(inputs, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(inputs))
    with torch.no_grad():
        return g.build(x.to(dtype), cropsize)


def tower_gradients(inputs, truth, weights, cropsize, dtype=torch.float64, variant="Dprime", trace=None, saved=None, _native=False):
    """One tower of the training twin (misc_py/denoiser-multi-gpu.py:752-782): architecture(phase=True) on
    ``inputs``, mse = mean((out-truth)^2), loss = 1000*mse if mse < 1e-3 else sqrt(1000*mse) (+ weight_decay * sum of
    l2 losses with weight_decay = 0, :117), tf.gradients(loss, trainable variables) via PyTorch autograd.
    -> dict(out, mse, loss, grads {name: numpy}, moving {name: numpy updated moving statistics}).
    float64 by default; the float32 form runs on ATen's native convolution kernels, not oneDNN (see below; ORACLE_MKLDNN=1 puts
    oneDNN back for diagnosis)."""
    leaves = {}

    def get(name, shape):
        if name not in leaves:
            w = weights[name]
            assert tuple(w.shape) == tuple(shape), (name, w.shape, shape)
            t = torch.from_numpy(np.ascontiguousarray(w)).to(dtype)
            if not name.endswith(("/moving_mean", "/moving_variance")):
                t.requires_grad_(True)
            leaves[name] = t
        return leaves[name]

    if dtype == torch.float32 and not _native and os.environ.get("ORACLE_MKLDNN", "0") != "1":
        # PyTorch-CPU's float32 convolutions go to oneDNN, float64 to ATen's own kernels.  On the GPU boxes' host (EPYC 9575F) the oneDNN
        # float32 backward of this tower aborts with glibc heap corruption -- deterministically, in a process that has loaded neither
        # libemdenoise.so nor the HIP device runtime (gpurun_out/r3i) -- and runs clean on ATen's kernels.  The oracle is a
        # checker: it takes the path that works on every host.
        with torch.backends.mkldnn.flags(enabled=False):
            return tower_gradients(inputs, truth, weights, cropsize, dtype=torch.float32, variant=variant, trace=trace, saved=saved, _native=True)
    g = _Graph(get, dtype, variant)
    g.training = True
    g.moving_updates = {}
    g.trace = trace  # optional list: every relu6 output (detached by the caller if kept)
    g.saved = saved  # optional dict: per conv scope, the tensors a hand-written backward pass saves (see _Graph.saved)
    x = inputs if isinstance(inputs, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(inputs))
    t = truth if isinstance(truth, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(truth))
    out = g.build(x.to(dtype), cropsize)
    mse = ((out - t.to(dtype).reshape(out.shape)) ** 2).mean()
    loss = 1000.0 * mse if float(mse.detach()) < 0.001 else torch.sqrt(1000.0 * mse)
    names = [n for n, v in leaves.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)
    return {
        "out": out.detach(), "mse": float(mse.detach()), "loss": float(loss.detach()),
        "grads": {n: (gr.numpy() if gr is not None else np.zeros(tuple(leaves[n].shape))) for n, gr in zip(names, grads)},
        "moving": {n: v.numpy() for n, v in g.moving_updates.items()},
    }


def nesterov_step(params, grads, accums, lr, momentum=0.9):
    """tf.train.MomentumOptimizer(lr, momentum, use_nesterov=True).apply_gradients (denoiser-multi-gpu.py:1064-1071),
    i.e. ApplyMomentum: accum = momentum*accum + g; var -= lr*g + lr*momentum*accum.  Dicts of numpy arrays; returns
    (new params, new accums)."""
    new_p, new_a = {}, {}
    for n, g in grads.items():
        a = momentum * accums[n] + g
        new_a[n] = a
        new_p[n] = params[n] - (g * lr + a * momentum * lr)
    return new_p, new_a
