"""Graph G oracle: the in-filling GAN's GENERATOR (oracle; test infrastructure only).

Restates ``generator_architecture`` of misc_py/gan-infilling-100.py:133-374 with the TF op semantics of
oracle/tf_ops.py on PyTorch-CPU tensors.  PARITY UNPINNED (oracle/__init__.py).  Inference form: the batch norms
(``train_batch_norm`` placeholder False, :164-174, :1509) use their moving statistics, epsilon 0.01.

Things the reference does that are reproduced on purpose:
  * separable convs are reflect-padded by ``pad_size`` and run VALID (:209-216); with stride 2 the window of output i
    therefore starts at input row 2i-1 (reflected at -1), not at 2i as TF's SAME would;
  * ``deconv_block`` (:259-266) calls ``conv_block(deconv, filters, pad_size)``: the third positional parameter of
    conv_block is ``phase``, so ``pad_size`` is swallowed and those four separable convs run with SAME zero padding;
  * ``_instance_norm`` (:140-148) creates two NON-trainable tf.Variables (shift 0, scale 1) per call: they are part
    of the variable list (names GAN/Gen/Variable, Variable_1) but never change;
  * the first layer is a 7x7 separable conv on the 1-channel image (:343-347); the output is
    tanh(instance_norm(conv3x3 + bias)) (:362-372);
  * ``xception_encoding_block`` (:268-285) references an undefined ``cnn1`` -- it is never called by the generator.
Variables live under ``tf.variable_scope("GAN/Gen")`` with an inner ``reg`` scope (:353): default layer scopes are
uniquified per parent scope, so the numbering restarts inside ``reg``.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import tf_ops as T

# gan-infilling-100.py:40-62
gen_features0, gen_features1, gen_features2, gen_features3 = 32, 64, 64, 32
nin_features1, nin_features2, nin_features3 = 128, 256, 768
nin_features_out1, nin_features_out2, nin_features_out3 = 256, 128, 64
num_global_enhancer_blocks, num_local_enhancer_blocks = 8, 3
BN_EPS_GEN = 0.01      # :167
IN_EPS = 1e-3          # :146
LEAKY = 0.2            # tf.nn.leaky_relu default


class _Scope:
    """tf default-name uniquifier with nested variable scopes (one counter per parent scope)."""

    def __init__(self, root):
        self.stack, self.counts = [root], {}

    def push(self, name):
        self.stack.append(name)

    def pop(self):
        self.stack.pop()

    def unique(self, base):
        parent = "/".join(self.stack)
        k = self.counts.get((parent, base), 0)
        self.counts[(parent, base)] = k + 1
        return f"{parent}/{base}" if k == 0 else f"{parent}/{base}_{k}"


def reflect_pad_t(x, p):
    """tf.pad(mode="REFLECT") of H and W by p (:152-158): the border sample is not repeated."""
    return F.pad(x.permute(0, 3, 1, 2), (p, p, p, p), mode="reflect").permute(0, 2, 3, 1)


def depthwise_valid_t(x, w, stride):
    """Depthwise conv, VALID padding (after an explicit reflect pad).  x [B,H,W,C]; w [k,k,C,1]."""
    C = x.shape[-1]
    wt = w.permute(2, 3, 0, 1).contiguous()
    return F.conv2d(x.permute(0, 3, 1, 2), wt, None, stride=stride, groups=C).permute(0, 2, 3, 1)


class _Gen:
    def __init__(self, get, dtype):
        self.get, self.dtype, self.sc = get, dtype, _Scope("GAN/Gen")
        self.trace = None
        self.calibrate = None  # optional dict: batch statistics of every BN input are written here AND used
        self.moving_updates = None   # dict: train_batch_norm = True (:170) -- batch statistics are used and the moving averages updated

    def _bn(self, x, scope):
        C = x.shape[-1]
        beta, gamma = self.get(scope + "/beta", (C,)), self.get(scope + "/gamma", (C,))
        mean, var = self.get(scope + "/moving_mean", (C,)), self.get(scope + "/moving_variance", (C,))
        if self.calibrate is not None:  # used only to SYNTHESISE weights (tests/golden/make_synth_bn.py)
            mean, var = x.mean(dim=(0, 1, 2)), x.var(dim=(0, 1, 2), unbiased=False)
            self.calibrate[scope + "/moving_mean"] = mean.to(torch.float32).numpy().copy()
            self.calibrate[scope + "/moving_variance"] = var.to(torch.float32).numpy().copy()
            mean, var = mean.to(torch.float32).to(self.dtype), var.to(torch.float32).to(self.dtype)
        if self.moving_updates is not None:   # is_training=True: tf.contrib.layers.batch_norm, fused, decay batch_decay_gen = 0.9997 (:112)
            bm, bv = x.mean(dim=(0, 1, 2)), x.var(dim=(0, 1, 2), unbiased=False)
            n = x.shape[0] * x.shape[1] * x.shape[2]
            self.moving_updates[scope + "/moving_mean"] = mean - (mean - bm) * (1.0 - 0.9997)
            self.moving_updates[scope + "/moving_variance"] = var - (var - bv * (n / max(n - 1, 1))) * (1.0 - 0.9997)   # Bessel-corrected
            mean, var = bm, bv
        return (x - mean) * (gamma / torch.sqrt(var + BN_EPS_GEN)) + beta

    def batch_then_activ(self, x):
        y = F.leaky_relu(self._bn(x, self.sc.unique("BatchNorm")), LEAKY)
        if self.trace is not None:
            self.trace.append(y)
        return y

    def instance_norm(self, x):
        C = x.shape[-1]
        shift = self.get(self.sc.unique("Variable"), (C,))   # tf.Variable(tf.zeros), trainable=False
        scale = self.get(self.sc.unique("Variable"), (C,))   # tf.Variable(tf.ones), trainable=False
        mu = x.mean(dim=(1, 2), keepdim=True)
        var = x.var(dim=(1, 2), unbiased=False, keepdim=True)
        return scale * ((x - mu) / torch.sqrt(var + IN_EPS)) + shift

    # :205-243
    def sep(self, x, filters, stride=1, k=3, pad=None):
        scope = self.sc.unique("SeparableConv2d")
        cin = x.shape[-1]
        dw = self.get(scope + "/depthwise_weights", (k, k, cin, 1))
        pw = self.get(scope + "/pointwise_weights", (1, 1, cin, filters))
        if pad:
            y = depthwise_valid_t(reflect_pad_t(x, pad), dw, stride)
        else:
            y = T.depthwise_conv2d_t(x, dw, stride=stride)
        y = T.conv2d_t(y, pw, None)
        y = self._bn(y, scope + "/BatchNorm")
        return self.batch_then_activ(y)

    # :259-266 -- pad_size lands in conv_block's `phase` parameter: SAME padding
    def deconv_block(self, x, filters, size):
        return self.sep(T.resize_bilinear_legacy_t(x, size, size), filters)

    # :287-307
    def middle_block(self, x, features):
        m = self.sep(x, features, pad=1)
        m = self.sep(m, features, pad=1)
        m = self.sep(m, features, pad=1)
        return m + x

    # :326-339 (sizes are hard-coded for a 512-px input; kept proportional for smaller test crops)
    def network_in_network(self, x, S):
        n = self.sep(x, nin_features1, 2, pad=1)
        n = self.sep(n, nin_features2, 2, pad=1)
        n = self.sep(n, nin_features3, 2, pad=1)
        for _ in range(num_global_enhancer_blocks):
            n = self.middle_block(n, nin_features3)
        n = self.deconv_block(n, nin_features_out1, S // 8)
        n = self.deconv_block(n, nin_features_out2, S // 4)
        return self.deconv_block(n, nin_features_out3, S // 2)

    def build(self, inputs, S):
        x = inputs.reshape(-1, S, S, 1)
        enc = self.sep(x, gen_features0, 1, k=7, pad=3)
        enc = self.sep(enc, gen_features1, 2, pad=1)
        self.sc.push("reg")
        enc = enc + self.network_in_network(enc, S)
        for _ in range(num_local_enhancer_blocks):
            enc = self.middle_block(enc, gen_features2)
        enc = self.deconv_block(enc, gen_features3, S)
        enc = self.sep(enc, gen_features3, 1, pad=1)
        self.sc.pop()
        scope = self.sc.unique("Conv")
        w = self.get(scope + "/weights", (3, 3, gen_features3, 1))
        b = self.get(scope + "/biases", (1,))
        wt = w.permute(3, 2, 0, 1).contiguous()
        enc = F.conv2d(reflect_pad_t(enc, 1).permute(0, 3, 1, 2), wt, b).permute(0, 2, 3, 1)
        return torch.tanh(self.instance_norm(enc))


def generator_moving_update(inputs, weights, cropsize=512, dtype=torch.float64):
    """The generator train op's update ops while train_batch_norm_on (gan-infilling-100.py:1644, :1708-1712, :866-871, :1384): one
    forward pass with is_training = True on every batch norm (:164-174) -> (output, {moving statistic name: updated value})."""
    cache = {}

    def get(name, shape):
        if name not in cache:
            w = weights[name]
            assert tuple(w.shape) == tuple(shape), (name, w.shape, shape)
            cache[name] = torch.from_numpy(np.ascontiguousarray(w)).to(dtype)
        return cache[name]

    g = _Gen(get, dtype)
    g.moving_updates = {}
    x = inputs if isinstance(inputs, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(inputs))
    out = g.build(x.to(dtype), cropsize)
    return out, {k: v.numpy() for k, v in g.moving_updates.items()}


def variable_specs(cropsize=64) -> "OrderedDict[str, tuple]":
    """Names and shapes of every generator variable, in creation order."""
    specs = OrderedDict()

    def rec(name, shape):
        specs[name] = tuple(int(s) for s in shape)
        return torch.ones(shape, dtype=torch.float32)

    with torch.no_grad():
        _Gen(rec, torch.float32).build(torch.zeros(1, cropsize, cropsize, 1), cropsize)
    return specs


def generator(inputs, weights, cropsize=512, dtype=torch.float32, trace=None, calibrate=None):
    """inputs [B,S,S,1] (-1 = missing pixel) -> torch [B,S,S,1] in (-1,1).  S a multiple of 16."""
    cache = {}

    def get(name, shape):
        if name not in cache:
            w = weights[name]
            assert tuple(w.shape) == tuple(shape), (name, w.shape, shape)
            cache[name] = torch.from_numpy(np.ascontiguousarray(w)).to(dtype)
        return cache[name]

    g = _Gen(get, dtype)
    g.trace = trace
    g.calibrate = calibrate
    x = inputs if isinstance(inputs, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(inputs))
    with torch.no_grad():
        return g.build(x.to(dtype), cropsize)


# ================================================================================================
# Discriminator (misc_py/gan-infilling-100.py:376-710), inference form (phase=False: the separable convs' batch norms
# use their moving statistics, epsilon 1e-3; "batch_then_activ" is an INSTANCE norm + leaky_relu, :413-416).
# ================================================================================================
features1, features2, features3, features4, features5 = 32, 64, 128, 256, 512
BN_EPS_DISCR = 1e-3   # :401


def reflect_indices(n, pad):
    idx = np.arange(-pad, n + pad)
    idx = np.abs(idx)
    return np.where(idx >= n, 2 * n - 2 - idx, idx)


def multiscale_crops(img, offsets):
    """get_multiscale_crops (:957-980) with the tf.random_crop offsets given: img [B,S,S,C] is reflect-padded by 3S/4,
    then small = S/4 crop, medium = S/2 crop, large = 3S/4 crop resized (legacy bilinear) to S/4.
    offsets: ((y,x) small, (y,x) medium, (y,x) large) in the PADDED image.  The medium crop is returned at S/2: the
    discriminator average-pools it (:585-588)."""
    x = img if isinstance(img, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img))
    B, S = x.shape[0], x.shape[1]
    pad = (3 * S) // 4
    ridx = torch.from_numpy(reflect_indices(S, pad))
    xp = x[:, ridx][:, :, ridx]
    (ys, xs), (ym, xm), (yl, xl) = offsets
    small = xp[:, ys:ys + S // 4, xs:xs + S // 4]
    medium = xp[:, ym:ym + S // 2, xm:xm + S // 2]
    large = xp[:, yl:yl + pad, xl:xl + pad]
    return small, medium, T.resize_bilinear_legacy_t(large, S // 4, S // 4)


class _Discr:
    def __init__(self, get, dtype, training=False):
        self.get, self.dtype, self.sc = get, dtype, _Scope("GAN/Discr")
        self.training = training      # phase=True: the separable convs' batch norms use batch statistics (:399-409)
        self.moving_updates = {} if training else None   # decay 0.9997 (batch_decay_discr, :110)

    def instance_norm(self, x):
        C = x.shape[-1]
        shift = self.get(self.sc.unique("Variable"), (C,))
        scale = self.get(self.sc.unique("Variable"), (C,))
        mu = x.mean(dim=(1, 2), keepdim=True)
        var = x.var(dim=(1, 2), unbiased=False, keepdim=True)
        return scale * ((x - mu) / torch.sqrt(var + IN_EPS)) + shift

    # :440-463
    def strided_conv_block(self, x, filters, stride):
        scope = self.sc.unique("SeparableConv2d")
        cin = x.shape[-1]
        dw = self.get(scope + "/depthwise_weights", (3, 3, cin, 1))
        pw = self.get(scope + "/pointwise_weights", (1, 1, cin, filters))
        y = T.conv2d_t(T.depthwise_conv2d_t(x, dw, stride=stride), pw, None)
        b = scope + "/BatchNorm"
        beta, gamma = self.get(b + "/beta", (filters,)), self.get(b + "/gamma", (filters,))
        mean, var = self.get(b + "/moving_mean", (filters,)), self.get(b + "/moving_variance", (filters,))
        if self.training:
            bm, bv = y.mean(dim=(0, 1, 2)), y.var(dim=(0, 1, 2), unbiased=False)
            n = y.shape[0] * y.shape[1] * y.shape[2]
            self.moving_updates[b + "/moving_mean"] = mean.detach() - (mean.detach() - bm.detach()) * (1.0 - 0.9997)
            self.moving_updates[b + "/moving_variance"] = var.detach() - (var.detach() - bv.detach() * (n / max(n - 1, 1))) * (1.0 - 0.9997)
            mean, var = bm, bv
        y = (y - mean) * (gamma / torch.sqrt(var + BN_EPS_DISCR)) + beta
        return F.leaky_relu(self.instance_norm(y), LEAKY)

    def branch(self, name, x, layers):
        self.sc.push(name)
        if name == "medium":
            x = T.avg_pool2x2_same_t(x)
        for f in (features1, features2, features3, features4, features5):
            x = self.strided_conv_block(x, f, 2)
            layers.append(x)
        x = x.mean(dim=(1, 2))
        scope = self.sc.unique("fully_connected")
        w = self.get(scope + "/weights", (features5, 1))
        b = self.get(scope + "/biases", (1,))
        self.sc.pop()
        return x @ w + b

    def build(self, inputs):
        layers = []
        logits = [self.branch(n, x, layers) for n, x in zip(("small", "medium", "large"), inputs)]
        out = torch.sigmoid(torch.cat(logits, dim=1).max(dim=1).values)
        return [out] + layers


def discriminator_variable_specs(cropsize=64):
    specs = OrderedDict()

    def rec(name, shape):
        specs[name] = tuple(int(s) for s in shape)
        return torch.ones(shape, dtype=torch.float32)

    S = cropsize
    with torch.no_grad():
        _Discr(rec, torch.float32).build([torch.zeros(1, S // 4, S // 4, 1), torch.zeros(1, S // 2, S // 2, 1),
                                          torch.zeros(1, S // 4, S // 4, 1)])
    return specs


def discriminator(inputs, weights, dtype=torch.float32):
    """inputs: [small [B,S/4,S/4,1], medium [B,S/2,S/2,1], large [B,S/4,S/4,1]] -> [output [B]] + the 15 feature maps
    (the list the generator's feature-matching loss walks, :1029-1035)."""
    cache = {}

    def get(name, shape):
        if name not in cache:
            w = weights[name]
            assert tuple(w.shape) == tuple(shape), (name, w.shape, shape)
            cache[name] = torch.from_numpy(np.ascontiguousarray(w)).to(dtype)
        return cache[name]

    xs = [(x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))).to(dtype) for x in inputs]
    with torch.no_grad():
        return _Discr(get, dtype).build(xs)


# ================================================================================================
# Training towers (misc_py/gan-infilling-100.py:1048-1088 discriminator, :982-1046 generator) and the optimizer
# (:1378-1379, :1429-1431: Adam beta1 0.5 wrapped in clip_gradients_by_norm), via PyTorch autograd.
# ================================================================================================
def _leaves(weights, dtype, trainable):
    leaves = {}

    def get(name, shape):
        if name not in leaves:
            w = weights[name]
            assert tuple(w.shape) == tuple(shape), (name, w.shape, shape)
            t = torch.from_numpy(np.ascontiguousarray(w)).to(dtype)
            if trainable(name):
                t.requires_grad_(True)
            leaves[name] = t
        return leaves[name]

    return leaves, get


def _is_trainable(name):
    leaf = name.rsplit("/", 1)[1]
    return not (leaf.startswith("moving_") or leaf.startswith("Variable"))


def discriminator_tower(image, label, weights, offsets, adapt=1.0, dtype=torch.float64):
    """_discriminator_tower_fn (:1048-1088) for ONE image [1,S,S,1] (batch_size = 1, :74): crops, discriminator with
    phase=True, tower_loss = -log(clip(1 - |label - D|, 1e-8, 1 - 1e-8)) + 5e-5 * sum l2_loss(v) over the trainable
    discriminator variables; gradients of adapt * tower_loss.  -> dict(output, loss, grads, moving)."""
    leaves, get = _leaves(weights, dtype, _is_trainable)
    x = image if isinstance(image, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(image))
    crops = multiscale_crops(x.to(dtype), offsets)
    d = _Discr(get, dtype, training=True)
    out = d.build(list(crops))[0]
    names = [n for n, v in leaves.items() if v.requires_grad]
    l2 = sum(0.5 * (leaves[n] ** 2).sum() for n in names)
    loss = -torch.log(torch.clamp(1.0 - torch.abs(float(label) - out), 1e-8, 1.0 - 1e-8)).sum() + 5e-5 * l2
    grads = torch.autograd.grad(float(adapt) * loss, [leaves[n] for n in names], allow_unused=True)
    return {"output": out.detach().numpy(), "loss": float(loss.detach()),
            "grads": {n: (g.numpy() if g is not None else np.zeros(tuple(leaves[n].shape))) for n, g in zip(names, grads)},
            "moving": {n: v.numpy() for n, v in d.moving_updates.items()}}


def clip_by_global_norm(grads, clip_norm):
    """tf.clip_by_global_norm as clip_gradients_by_norm applies it: g * clip_norm / max(global_norm, clip_norm)."""
    gn = float(np.sqrt(sum(float((np.asarray(g, np.float64) ** 2).sum()) for g in grads.values())))
    scale = clip_norm / max(gn, clip_norm)
    return {n: g * scale for n, g in grads.items()}, gn


def adam_step(params, grads, m, v, t, lr, beta1=0.5, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer(lr, beta1): lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t); m, v moments; var -= lr_t*m/(sqrt(v)+eps).
    t = the 1-based step count.  Dicts of numpy arrays; returns (params, m, v)."""
    lr_t = lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    new_p, new_m, new_v = {}, {}, {}
    for n, g in grads.items():
        new_m[n] = beta1 * m[n] + (1.0 - beta1) * g
        new_v[n] = beta2 * v[n] + (1.0 - beta2) * g * g
        new_p[n] = params[n] - lr_t * new_m[n] / (np.sqrt(new_v[n]) + eps)
    return new_p, new_m, new_v


def generator_tower(lq, truth, gen_weights, discr_weights, offsets, dtype=torch.float64):
    """_generator_tower_fn (:982-1046) for ONE image (batch_size = 1), as the training loop runs it: the generator with
    its batch norms on MOVING statistics (batch_norm_on_ph is fed False while the tower gradients are evaluated, :1667,
    so decay = 0 and there is no l2 term, :1039-1042); concat(output, truth) is cropped ONCE (the same offsets for both,
    :1008-1015); the discriminator runs with phase=True on the generated and on the natural crops;
        loss = -log(clip(D(fake), 1e-8, 1)) + 12 * sum_l mean|f_l(fake) - f_l(natural)|   (:1027-1037)
    gradients w.r.t. the trainable GENERATOR variables only.  -> dict(output, d_fake, loss, grads)."""
    leaves, get = _leaves(gen_weights, dtype, _is_trainable)
    g = _Gen(get, dtype)
    x = lq if isinstance(lq, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(lq))
    t = truth if isinstance(truth, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(truth))
    S = x.shape[1]
    out = g.build(x.to(dtype), S)
    both = multiscale_crops(torch.cat([out, t.to(dtype).reshape(out.shape)], dim=3), offsets)
    fake = [c[..., 0:1] for c in both]
    natural = [c[..., 1:2].detach() for c in both]
    dleaves, dget = _leaves(discr_weights, dtype, lambda n: False)
    dfake = _Discr(dget, dtype, training=True).build(fake)
    dnat = _Discr(dget, dtype, training=True).build(natural)
    stat = sum((a - b).abs().mean() for a, b in zip(dfake[1:], dnat[1:]))
    loss = -torch.log(torch.clamp(dfake[0], 1e-8, 1.0)).sum() + 12.0 * stat
    names = [n for n, v in leaves.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)
    return {"output": out.detach().numpy(), "d_fake": dfake[0].detach().numpy(), "loss": float(loss.detach()),
            "stat_loss": float(stat.detach()),
            "grads": {n: (gr.numpy() if gr is not None else np.zeros(tuple(leaves[n].shape))) for n, gr in zip(names, grads)}}
