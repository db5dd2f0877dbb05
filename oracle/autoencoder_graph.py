"""ORACLE (test infrastructure only; never imported by the product path): CPU restatement of the small separable
autoencoder of misc_py/apply_autoencoders.py ("graph S"), PyTorch-CPU, float32 or float64.

Follows, as text, /root/reference/misc_py/apply_autoencoders.py:
  :91-187   architecture(input, encoding_features): 4 x strided_conv_block (slim.separable_convolution2d 3x3, SAME,
            strides 2,2,2,1, channels 64/128/256/encoding_features; normalizer batch_norm, then batch_then_activ =
            a second batch_norm + relu), 3 x deconv_block (slim.conv2d_transpose k3 s2 SAME + bias; batch norm + relu on
            the first two, relu alone on the third), slim.conv2d 3x3 64 -> 1 without bias or activation.
  :105-116  every batch_norm runs with is_training=True (batch statistics, biased variance, eps 1e-3, center and scale)
            -- at inference too; the class feeds ONE 160x160 crop per sess.run (:331), so the statistics are per image.
PARITY UNPINNED by the reference (it has no tests, vectors or checkpoints for this path); pinned as the other graphs are:
TF-op restatements with known-answer tests (tests/test_oracle_ops.py), a float64 run as the arbiter, a committed golden
vector (tests/golden/s_graph_160.npz).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from . import tf_ops as T

BN_EPS = 1e-3
CROPSIZE = 160
ENC_CHANNELS = (64, 128, 256)
DEC_CHANNELS = (256, 128, 64)


def variable_specs(encoding_features: int = 16):
    """TF variable name -> shape in creation order.  No outer variable_scope in this file (:190-196): slim's default
    scopes SeparableConv2d[_k] (with the normalizer's BatchNorm inside), BatchNorm[_k] for batch_then_activ,
    Conv2d_transpose[_k], Conv."""
    v = OrderedDict()

    def bn(scope, c):
        for leaf in ("beta", "gamma", "moving_mean", "moving_variance"):
            v[f"{scope}/{leaf}"] = (c,)

    cin, nbn = 1, 0
    for k, cout in enumerate(ENC_CHANNELS + (encoding_features,)):
        s = "SeparableConv2d" if k == 0 else f"SeparableConv2d_{k}"
        v[s + "/depthwise_weights"] = (3, 3, cin, 1)
        v[s + "/pointwise_weights"] = (1, 1, cin, cout)
        bn(s + "/BatchNorm", cout)
        bn("BatchNorm" if nbn == 0 else f"BatchNorm_{nbn}", cout)
        nbn += 1
        cin = cout
    for k, cout in enumerate(DEC_CHANNELS):
        s = "Conv2d_transpose" if k == 0 else f"Conv2d_transpose_{k}"
        v[s + "/weights"] = (3, 3, cout, cin)
        v[s + "/biases"] = (cout,)
        if k < 2:
            bn(f"BatchNorm_{nbn}", cout)
            nbn += 1
        cin = cout
    v["Conv/weights"] = (3, 3, cin, 1)
    return v


def _bn_batch(x, gamma, beta):
    """tf.contrib.layers.batch_norm(is_training=True): statistics over (N,H,W) of THIS call, biased variance."""
    mean = x.mean(dim=(0, 1, 2), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(0, 1, 2), keepdim=True)
    return (x - mean) / torch.sqrt(var + BN_EPS) * gamma + beta


def architecture(x, w, encoding_features: int = 16, dtype=torch.float64, trace=None):
    """x [B,S,S,1] (numpy) -> [B,S,S,1] torch tensor; every image is run on its own (the reference's batch of one)."""
    g = lambda name: torch.from_numpy(np.asarray(w[name])).to(dtype)
    outs = []
    for b in range(x.shape[0]):
        a = torch.from_numpy(np.asarray(x[b:b + 1])).to(dtype)
        nbn = 0
        for k, stride in enumerate((2, 2, 2, 1)):
            s = "SeparableConv2d" if k == 0 else f"SeparableConv2d_{k}"
            a = T.depthwise_conv2d_t(a, g(s + "/depthwise_weights"), stride=stride)
            a = T.conv2d_t(a, g(s + "/pointwise_weights"), None)
            a = _bn_batch(a, g(s + "/BatchNorm/gamma"), g(s + "/BatchNorm/beta"))
            o = "BatchNorm" if nbn == 0 else f"BatchNorm_{nbn}"
            a = torch.relu(_bn_batch(a, g(o + "/gamma"), g(o + "/beta")))
            nbn += 1
            if trace is not None and b == 0:
                trace.append(a.numpy().copy())
        for k in range(3):
            s = "Conv2d_transpose" if k == 0 else f"Conv2d_transpose_{k}"
            a = T.conv2d_transpose_s2_t(a, g(s + "/weights"), g(s + "/biases"))
            if k < 2:
                o = f"BatchNorm_{nbn}"
                a = _bn_batch(a, g(o + "/gamma"), g(o + "/beta"))
                nbn += 1
            a = torch.relu(a)
            if trace is not None and b == 0:
                trace.append(a.numpy().copy())
        a = T.conv2d_t(a, g("Conv/weights"), None)
        outs.append(a)
    return torch.cat(outs, 0)
