"""Graph D host side: the depthwise-separable encoder-decoder denoiser on MI355X.

Mirrors machine_learning/denoiser.py of the reference:
  * ``architecture()`` (:58-398)  -> ``DenoiserEngine`` (declares the layers in the reference's
    graph-construction order so that every variable keeps its TensorFlow name, folds the inference
    batch norms into per-channel affines, packs the weights once, and runs the graph as a sequence of
    libemdenoise.so launches);
  * ``class Denoiser`` (:584-682) -> ``Denoiser`` with the same constructor arguments and
    ``preprocess`` / ``denoise_crop`` / ``denoise`` methods, plus the batched
    ``denoise(lq_batch[B,512,512,1]) -> hq_batch`` form;
  * ``scale0to1`` (:684-695).
Python here is plumbing only (buffers, pointers, launch order); there is no CPU compute path.

``cropsize`` is 512 in the reference (:54) with ``aspp_size = 32 = cropsize/16`` (:45); the engine keeps
that ratio so the same graph can be run (and checked against the oracle) at smaller crops.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np

from . import _lib, ops, streams

# denoiser.py:38-52
features0, features1, features2, features3, features4 = 64, 128, 256, 728, 728
aspp_filters = features4
aspp_output = 256
aspp_rateSmall, aspp_rateMedium, aspp_rateLarge = 6, 12, 18
num_extra_blocks = 11
cropsize = 512
channels = 1
BN_EPS = 1e-3  # tf.contrib.layers.batch_norm default

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
SYNTH_SEED = 1234


# ------------------------------------------------------------------------------------------------
# layer declarations in the reference's creation order (=> TensorFlow variable names)
# ------------------------------------------------------------------------------------------------
class _Scope:
    """tf.variable_scope default-name uniquifier inside scope 'nn' (denoiser.py:514)."""

    def __init__(self):
        self.n = {}

    def __call__(self, base):
        k = self.n.get(base, 0)
        self.n[base] = k + 1
        return f"nn/{base}" if k == 0 else f"nn/{base}_{k}"


class Layer:
    def __init__(self, kind, cin, cout, **kw):
        self.kind, self.cin, self.cout = kind, cin, cout
        self.stride = kw.get("stride", 1)
        self.rate = kw.get("rate", 1)
        self.k = kw.get("k", 1)
        self.scope = kw.get("scope")      # conv / separable-conv / conv2d_transpose scope
        self.bn = kw.get("bn", [])        # batch-norm scopes applied after it, in order
        self.extra_bn = kw.get("extra_bn")  # a second batch_then_activ (ASPP rate branches)
        self.style = kw.get("style", "slim")  # "slim": weights/biases; "layers" (tf.layers): kernel/bias

    @property
    def wname(self):
        return "weights" if self.style == "slim" else "kernel"

    @property
    def bname(self):
        return "biases" if self.style == "slim" else "bias"

    def variables(self):
        v = OrderedDict()
        if self.kind == "sep":
            v[self.scope + "/depthwise_weights"] = (3, 3, self.cin, 1)
            v[self.scope + "/pointwise_weights"] = (1, 1, self.cin, self.cout)
        elif self.kind == "conv":
            v[self.scope + "/" + self.wname] = (self.k, self.k, self.cin, self.cout)
            v[self.scope + "/" + self.bname] = (self.cout,)
        elif self.kind == "deconv":
            v[self.scope + "/" + self.wname] = (3, 3, self.cout, self.cin)
            v[self.scope + "/" + self.bname] = (self.cout,)
        c = self.cout
        for s in self.bn + ([self.extra_bn] if self.extra_bn else []):
            # creation order inside tf.contrib.layers.batch_norm: beta, gamma, moving_mean, moving_variance
            for leaf in ("beta", "gamma", "moving_mean", "moving_variance"):
                v[f"{s}/{leaf}"] = (c,)
        return v


def declare_layers(variant="D"):
    """Every parameterised layer of architecture() keyed by the reference's Python variable name, in creation
    order.  variant "D": machine_learning/denoiser.py:248-398 (slim.conv2d / slim.conv2d_transpose);
    variant "Dprime": the training twin misc_py/denoiser-multi-gpu.py:200-540 run with phase=False
    (tf.layers.conv2d / conv2d_transpose => scopes conv2d_k / conv2d_transpose_k with kernel/bias, named ASPP
    convs, dense dilated 3x3 ASPP branches, a real image-level branch)."""
    if variant not in ("D", "Dprime"):
        raise ValueError("variant must be 'D' or 'Dprime'")
    twin = variant == "Dprime"
    sc = _Scope()
    L = OrderedDict()

    def sep(key, cin, cout, stride=1, rate=1, extra=False):
        scope = sc("SeparableConv2d")
        inner = scope + "/BatchNorm"          # normalizer_fn runs inside the layer's scope (:123)
        outer = sc("BatchNorm")               # batch_then_activ (:134)
        L[key] = Layer("sep", cin, cout, stride=stride, rate=rate, scope=scope, bn=[inner, outer],
                       extra_bn=sc("BatchNorm") if extra else None)

    def conv(key, cin, cout, k=1, stride=1, rate=1, name=None, bn=True):
        scope = ("nn/" + name if name else sc("conv2d")) if twin else sc("Conv")
        L[key] = Layer("conv", cin, cout, k=k, stride=stride, rate=rate, scope=scope,
                       bn=[sc("BatchNorm")] if bn else [], style="layers" if twin else "slim")

    def deconv(key, cin, cout):
        scope = sc("conv2d_transpose") if twin else sc("Conv2d_transpose")
        L[key] = Layer("deconv", cin, cout, scope=scope, bn=[sc("BatchNorm")], style="layers" if twin else "slim")

    f0, f1, f2, f3, f4 = features0, features1, features2, features3, features4
    sep("cnn0", channels, f0); sep("cnn0_last", f0, f0); sep("cnn0_strided", f0, f1, stride=2)
    conv("residual0", channels, f1, stride=2)
    sep("cnn1", f1, f1); sep("cnn1_last", f1, f1); sep("cnn1_strided", f1, f1, stride=2)
    conv("residual1", f1, f1, stride=2)
    sep("cnn2", f1, f2); sep("cnn2_last", f2, f2); sep("cnn2_strided", f2, f2, stride=2)
    conv("residual2", f1, f2, stride=2)
    sep("cnn3", f2, f3); sep("cnn3_last", f3, f3); sep("cnn3_strided", f3, f3, stride=2)
    conv("residual3", f2, f3, stride=2)
    sep("cnn4_a", f3, f4); sep("cnn4_b", f4, f4); sep("cnn4_last", f4, f4)
    for i in range(num_extra_blocks):
        for j in range(3):
            sep(f"middle{i}_{j}", f4, f4)
    if not twin:
        conv("aspp_conv1x1", f4, aspp_filters)
        sep("aspp_small", f4, aspp_filters, rate=aspp_rateSmall, extra=True)
        sep("aspp_medium", f4, aspp_filters, rate=aspp_rateMedium, extra=True)
        sep("aspp_large", f4, aspp_filters, rate=aspp_rateLarge, extra=True)
        L["aspp_pooling_bn"] = Layer("bn", f4, f4, bn=[sc("BatchNorm")])   # :199-200
        conv("aspp_reduce", 5 * aspp_filters, aspp_output)
    else:  # denoiser-multi-gpu.py:291-361
        conv("aspp_conv1x1", f4, aspp_filters, name="1x1")
        conv("aspp_small", f4, aspp_filters, k=3, rate=aspp_rateSmall, name="lowRate")
        conv("aspp_medium", f4, aspp_filters, k=3, rate=aspp_rateMedium, name="mediumRate")
        conv("aspp_large", f4, aspp_filters, k=3, rate=aspp_rateLarge, name="highRate")
        conv("aspp_image_conv", f4, aspp_filters, name="imageLevel", bn=False)   # conv -> resize -> BN -> relu6
        L["aspp_pooling_bn"] = Layer("bn", aspp_filters, aspp_filters, bn=[sc("BatchNorm")])
        conv("aspp_reduce", 5 * aspp_filters, aspp_output, name="pellet")
    sep("deconv2_a", aspp_output + f1, f2); sep("deconv2_b", f2, f2)
    conv("residual2_d", aspp_output + f1, f2)
    deconv("deconv2to1", f2, f2)
    sep("deconv1_a", f2 + f1, f1); sep("deconv1_b", f1, f1)
    conv("residual1_d", f2 + f1, f1)
    deconv("deconv1to0", f1, f1)
    sep("deconv0_a", f1, f0); sep("deconv0_b", f0, f0)
    conv("residual0_d", f1, f0)
    conv("deconv_final", f0, 1, k=3)      # :387 -- kernel_size defaults to 3
    return L


def variable_specs(variant="D"):
    """TF variable name -> shape, in creation order."""
    out = OrderedDict()
    for layer in declare_layers(variant).values():
        out.update(layer.variables())
    return out


# ------------------------------------------------------------------------------------------------
# synthetic weights (no checkpoint ships with the reference: its paths are network shares, :588)
# ------------------------------------------------------------------------------------------------
def synthetic_weights(seed: int = SYNTH_SEED, bn: str = "calibrated", variant: str = "D"):
    """Seeded weights: Xavier-uniform kernels as the reference initialises them (denoiser.py:125),
    small random biases and batch-norm gamma/beta; moving statistics either TF's initial values
    (bn='tf_init': mean 0, variance 1) or the calibrated set shipped in data/ for the default seed
    (bn='calibrated'), which keeps every layer's pre-activation near zero mean / unit variance so that
    relu6 is exercised on both sides through all ~60 layers."""
    rng = np.random.default_rng(seed)
    w = OrderedDict()
    for name, shape in variable_specs(variant).items():
        leaf = name.rsplit("/", 1)[1]
        if leaf in ("depthwise_weights", "pointwise_weights", "weights", "kernel"):
            rf = shape[0] * shape[1]
            lim = np.sqrt(6.0 / (rf * shape[2] + rf * shape[3]))
            w[name] = rng.uniform(-lim, lim, shape).astype(np.float32)
        elif leaf in ("biases", "bias"):
            w[name] = rng.uniform(-0.1, 0.1, shape).astype(np.float32)
        elif leaf == "gamma":
            w[name] = rng.uniform(0.8, 2.0, shape).astype(np.float32)
        elif leaf == "beta":
            w[name] = rng.uniform(-0.5, 1.0, shape).astype(np.float32)
        elif leaf == "moving_mean":
            w[name] = np.zeros(shape, np.float32)
        elif leaf == "moving_variance":
            w[name] = np.ones(shape, np.float32)
        else:
            raise AssertionError(name)
    if bn == "calibrated":
        path = os.path.join(DATA_DIR, f"synth_bn_seed{seed}.npz" if variant == "D" else f"synth_bn_{variant}_seed{seed}.npz")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: calibrated batch-norm statistics exist only for the shipped seed; "
                                    "use bn='tf_init' for other seeds")
        z = np.load(path, allow_pickle=False)
        for name in z.files:
            assert name in w and w[name].shape == z[name].shape, name
            w[name] = z[name].astype(np.float32)
    elif bn != "tf_init":
        raise ValueError("bn must be 'calibrated' or 'tf_init'")
    return w


def load_weights(checkpoint_loc, variant="D"):
    """Weights for ``checkpoint_loc`` (the constructor argument of the reference's Denoiser, denoiser.py:587-589):
      * a directory holding a TensorFlow checkpoint (``checkpoint`` + ``<prefix>.index`` + ``<prefix>.data-*``, what
        tf.train.Saver wrote and ``tf.train.latest_checkpoint`` resolves, :621-626) or the prefix itself -- read
        directly by emdenoise.tf_checkpoint, no TensorFlow needed; optimizer slots and global_step are ignored;
      * a ``.npz`` file (or a directory holding ``denoiser_weights.npz``) keyed by TF variable name."""
    from . import tf_checkpoint as ckpt

    specs = variable_specs(variant)
    prefix = None
    if os.path.isdir(checkpoint_loc):
        prefix = ckpt.latest_checkpoint(checkpoint_loc)
    elif os.path.exists(checkpoint_loc + ".index"):
        prefix = checkpoint_loc
    if prefix is not None:
        z = ckpt.read_checkpoint(prefix, names=list(specs))
        src = prefix + ".index"
    else:
        src = checkpoint_loc if checkpoint_loc.endswith(".npz") else os.path.join(checkpoint_loc, "denoiser_weights.npz")
        npz = np.load(src, allow_pickle=False)
        missing = [n for n in specs if n not in npz.files]
        if missing:
            raise KeyError(f"{src}: missing variable {missing[0]}")
        z = {n: npz[n] for n in specs}
    w = OrderedDict()
    for name, shape in specs.items():
        a = z[name]
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{src}: {name}: shape {a.shape} != {shape}")
        w[name] = a.astype(np.float32)
    return w


# ------------------------------------------------------------------------------------------------
def _bn_affine(w, scope):
    """Inference batch norm as y = x*g + h (float64)."""
    g = w[scope + "/gamma"].astype(np.float64) / np.sqrt(w[scope + "/moving_variance"].astype(np.float64) + BN_EPS)
    h = w[scope + "/beta"].astype(np.float64) - w[scope + "/moving_mean"].astype(np.float64) * g
    return g, h


def _fold(w, layer, bias=None):
    """bias + the layer's consecutive batch norms -> one affine (scale, shift), float32."""
    c = layer.cout
    s = np.ones(c)
    t = np.zeros(c) if bias is None else bias.astype(np.float64)
    for scope in layer.bn:
        g, h = _bn_affine(w, scope)
        s, t = s * g, t * g + h
    return s.astype(np.float32), t.astype(np.float32)


class DenoiserEngine:
    """Weights resident on one GPU + the launch sequence of architecture() (denoiser.py:248-398)."""

    def __init__(self, weights, device, precision="bf16x3", fuse_sep=True, variant="D"):
        import torch

        _lib.load()
        self.device = device
        self.fuse_sep = fuse_sep
        self.variant = variant
        self.precision = {"bf16x3": ops.PREC_BF16X3, "bf16": ops.PREC_BF16}[precision]
        self.layers = declare_layers(variant)
        self.two_streams = os.environ.get("EMD_D_TWO_STREAMS", "1") != "0"   # see _middle_flow
        # staggered two-half pipeline over the whole graph (see forward): opt-in -- measured 27.5 ms against 26.3 ms for the
        # single pass with the two-stream middle flow (profiles/r02_experiments.txt): the halves' HBM-bound stages collide more
        # than their matrix-core stages overlap
        self.pipeline = os.environ.get("EMD_D_PIPELINE", "0") == "1"
        self._pipe_streams = None
        self._halves = streams.TwoHalves(device)
        self.P = {}
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
        self._unit4, self._zero4 = d(np.array([1, 0, 0, 0])), d(np.zeros(4))
        for key, L in self.layers.items():
            p = {}
            if L.kind == "sep":
                dw = weights[L.scope + "/depthwise_weights"][..., 0]           # [3,3,Cin]
                pw = weights[L.scope + "/pointwise_weights"][0]                # [1,Cin,Cout]
                s, t = _fold(weights, L)
                if L.cin == 1:   # cnn0: depthwise on the 1-channel image, then an outer product
                    p["w9"] = d(dw.reshape(9))
                    p["a"] = d(pw.reshape(L.cout).astype(np.float64) * s)
                    p["shift"] = d(t)
                else:
                    p["dw"] = d(dw.reshape(9, L.cin))
                    p["pw"] = ops.PackedWeights(pw, False, device)
                    p["scale"], p["shift"] = d(s), d(t)
                if L.extra_bn:
                    g, h = _bn_affine(weights, L.extra_bn)
                    p["scale2"], p["shift2"] = d(g), d(h)
            elif L.kind == "conv":
                wt = weights[L.scope + "/" + L.wname]
                s, t = _fold(weights, L, weights[L.scope + "/" + L.bname])
                if L.cin == 1:   # residual0
                    p["a"] = d(wt.reshape(L.cout).astype(np.float64) * s)
                    p["shift"] = d(t)
                elif L.cout == 1:  # deconv_final
                    p["w"] = d(wt[..., 0].reshape(9, L.cin))
                    p["scale_f"], p["shift_f"] = float(s[0]), float(t[0])
                else:
                    p["pw"] = ops.PackedWeights(wt.reshape(L.k * L.k, L.cin, L.cout), False, device)
                    p["scale"], p["shift"] = d(s), d(t)
            elif L.kind == "deconv":
                s, t = _fold(weights, L, weights[L.scope + "/" + L.bname])
                p["phases"] = ops.pack_deconv(weights[L.scope + "/" + L.wname], device)
                p["scale"], p["shift"] = d(s), d(t)
            elif L.kind == "bn":
                g, h = _bn_affine(weights, L.bn[0])
                p["scale"], p["shift"] = d(g), d(h)
            self.P[key] = p

    # ---- building blocks
    def _sep(self, key, x, out=None, res=None):
        """strided_conv_block (denoiser.py:110-136): depthwise 3x3 -> 1x1 on the matrix cores with
        BN x2 (+ the optional extra BN) + relu6 (+ residual) fused into its epilogue."""
        L, p = self.layers[key], self.P[key]
        Ho, Wo = -(-x.H // L.stride), -(-x.W // L.stride)
        if (self.fuse_sep and ops.sep_fused_supported(x, L.cout, L.stride, L.rate)
                and (L.stride == 1 or (self.precision == ops.PREC_BF16X3 and not isinstance(out, ops.SplitAct)
                                       and os.environ.get("EMD_D_SEP_S2", "1") != "0"))):
            # one launch, the depthwise result stays in LDS (the HBM-bound single-N-tile layers)
            if out is None:
                out = ops.Act.empty(x.B, Ho, Wo, L.cout, self.device)
            return ops.sep_fused(x, p["dw"], p["pw"], p["scale"], p["shift"], out, scale2=p.get("scale2"),
                                 shift2=p.get("shift2"), res=res, precision=self.precision, stride=L.stride)
        if out is None:
            out = ops.Act.empty(x.B, Ho, Wo, L.cout, self.device)
        if self._sep_gemm_ok(x, L) and not isinstance(out, ops.SplitAct):
            # the 728-channel flow: the depthwise stage is computed per K step inside the pointwise GEMM (csrc/sep_gemm.hip)
            return ops.sep_gemm(x, p["dw"], p["pw"], p["scale"], p["shift"], out, scale2=p.get("scale2"), shift2=p.get("shift2"), res=res)
        if self.precision == ops.PREC_BF16X3 and ops.conv1x1_split32_supported(x.B * Ho * Wo, L.cin, L.cout):
            # matrix-core bound layers (the 728-channel flow): the depthwise kernel writes its result pre-split into
            # bf16 hi/lo, the pointwise GEMM gets both operands by LDS-DMA (csrc/gemm_split.hip); same arithmetic
            return ops.sep_split32(x, p["dw"], p["pw"], p["scale"], p["shift"], out, stride=L.stride, rate=L.rate,
                                   scale2=p.get("scale2"), shift2=p.get("shift2"), res=res)
        assert not isinstance(out, ops.SplitAct)
        tmp = ops.Act.empty(x.B, Ho, Wo, L.cin, self.device)
        ops.dw3x3(x, p["dw"], tmp, stride=L.stride, rate=L.rate)
        ops.conv1x1(tmp, p["pw"], p["scale"], p["shift"], out, scale2=p.get("scale2"), shift2=p.get("shift2"),
                    res=res, precision=self.precision)
        return out

    def _sep_gemm_ok(self, x, L):
        return (self.fuse_sep and self.precision == ops.PREC_BF16X3 and os.environ.get("EMD_D_SEPGEMM", "0") == "1"   # opt-in: measured slower than the two-kernel route
                and ops.sep_gemm_supported(x, L.cout, L.stride, L.rate))

    def _sep_and_projection(self, sep_key, conv_key, x):
        """A decoder pair that reads the same tensor (denoiser.py:356-359, :368-371, :380-383): the separable conv `sep_key`
        and the 1x1 residual projection `conv_key` -> (sep output, projection), one launch where emd_sep3x3_dual_f32 covers the
        shape (the input is then read from HBM once), two otherwise."""
        Ls, ps, Lc, pc = self.layers[sep_key], self.P[sep_key], self.layers[conv_key], self.P[conv_key]
        if (self.fuse_sep and self.precision == ops.PREC_BF16X3 and os.environ.get("EMD_D_DUAL", "1") != "0" and "scale2" not in ps
                # the one-launch form where it is the faster route: 64 | 64 columns (deconv0_a + residual0_d at 512^2: 2.14 ms against
                # 1.43 + 1.63 for the pair) and, on the LDS-DMA pipelined kernel, 128 | 128 (deconv1_a + residual1_d: 2.07 against 1.21 + 0.99)
                and Ls.stride == 1 and Ls.rate == 1 and Lc.stride == 1 and ops.sep_dual_preferred(x, Ls.cout, Lc.cout)):
            out = ops.Act.empty(x.B, x.H, x.W, Ls.cout, self.device)
            out2 = ops.Act.empty(x.B, x.H, x.W, Lc.cout, self.device)
            return ops.sep_dual(x, ps["dw"], ps["pw"], pc["pw"], ps["scale"], ps["shift"], out, pc["scale"], pc["shift"], out2)
        proj = self._conv1x1(conv_key, x)
        return self._sep(sep_key, x), proj

    def _middle_chain(self, x, out):
        """Encoder 4 and the middle flow (:312-325) on one batch (or part of one): 9 residual blocks of 3 separable convs."""
        t = self._sep("cnn4_a", x)
        t = self._sep("cnn4_b", t)
        cur = self._sep("cnn4_last", t, res=x, out=out if num_extra_blocks == 0 else None)
        for i in range(num_extra_blocks):
            t = self._sep(f"middle{i}_0", cur)
            t = self._sep(f"middle{i}_1", t)
            cur = self._sep(f"middle{i}_2", t, res=cur, out=out if i == num_extra_blocks - 1 else None)
            yield   # one block issued: the caller alternates between the halves of the batch
        return

    def _middle_flow(self, x):
        """The 27 separable convs at 1/16 resolution; with an even batch as two halves on two streams (streams.TwoHalves)."""
        out = ops.Act.empty(x.B, x.H, x.W, self.layers["cnn4_last"].cout, self.device)
        half = x.B // 2
        # with the depthwise stage inside the GEMM there is no bandwidth-bound kernel left for the other half's GEMM to overlap with
        fused = self._sep_gemm_ok(x, self.layers["cnn4_a"]) and os.environ.get("EMD_D_TWO_STREAMS_FUSED", "0") != "1"
        # ... and only while a half still fills the chip: its GEMMs run on 128-row tiles below 192 tiles of 256 rows, and two half-filled
        # launches side by side lose to one (B = 4 at 512^2: 3.84 ms as halves, 3.71 ms whole; B = 8: 6.65 vs 6.72; profiles/r03_experiments.txt 12)
        if (self.two_streams and not fused and x.B % 2 == 0 and half * x.H * x.W >= 4096 and self.precision == ops.PREC_BF16X3
                and ops.conv1x1_split32_supported(half * x.H * x.W, x.C, out.C)):
            # experiment (EMD_D_PARTS = 4 / 8): the batch in that many parts, two at a time on the two streams, each part through ALL 27 layers
            # before the next pair starts -- a part's 24 / 12 MB tensors stay in the 256 MiB Infinity Cache from layer to layer
            parts = int(os.environ.get("EMD_D_PARTS", "2"))
            if parts > 2 and x.B % parts == 0 and ops.conv1x1_split32_supported((x.B // parts) * x.H * x.W, x.C, out.C):
                per = 2 * (x.B // parts)
                for a in range(0, x.B, per):
                    self._halves.run(x.images(a, a + per), out.images(a, a + per), self._middle_chain)
                return out
            return self._halves.run(x, out, self._middle_chain)
        for _ in self._middle_chain(x, out):
            pass
        return out

    def _split_gemm_ok(self, npix, cout, ktot):
        """The LDS-DMA split32 GEMMs (csrc/gemm_split.hip) pay where the GEMM is matrix-core bound and 256 x 128 tiles fill the chip."""
        return (self.precision == ops.PREC_BF16X3 and cout >= 128 and ktot >= 512
                and (-(-npix // 256)) * (-(-cout // 128)) >= 192)

    def _pw_split_ok(self, cin, cout):
        """The pointwise split32 GEMM (16x16x32 MFMAs since round 3) serves a layer at EVERY batch size or at none: its K-step sum is
        ordered differently from the register-staged kernel's, and image b of a batch must equal the image alone."""
        return self.precision == ops.PREC_BF16X3 and ops.conv1x1_split32_supported(1 << 20, cin, cout)

    def _conv1x1(self, key, x, out=None, res=None, xs=None):
        """xs: x already converted to split32 (shared by several consumers, e.g. the ASPP branches)."""
        L, p = self.layers[key], self.P[key]
        Ho, Wo = -(-x.H // L.stride), -(-x.W // L.stride)
        if out is None:
            out = ops.Act.empty(x.B, Ho, Wo, L.cout, self.device)
        if xs is not None and L.stride == 1 and self._pw_split_ok(L.cin, L.cout):
            return ops.conv1x1_split32(xs, p["pw"], p["scale"], p["shift"], out, act=bool(L.bn), res=res)
        ops.conv1x1(x, p["pw"], p["scale"], p["shift"], out, stride=L.stride, act=bool(L.bn), res=res,
                    precision=self.precision)
        return out

    def _conv3x3(self, key, x, out=None, xs=None):
        """Dense (dilated) 3x3 conv + bias + BN + relu6: the twin's ASPP rate branches (denoiser-multi-gpu.py:306-328)."""
        L, p = self.layers[key], self.P[key]
        if out is None:
            out = ops.Act.empty(x.B, x.H, x.W, L.cout, self.device)
        if L.stride == 1 and self._split_gemm_ok(x.B * x.H * x.W, L.cout, 9 * L.cin):
            xs = xs if xs is not None else ops.to_split32(x)
            return ops.conv3x3_split32(xs, p["pw"], p["scale"], p["shift"], out, rate=L.rate)
        return ops.conv3x3(x, p["pw"], p["scale"], p["shift"], out, stride=L.stride, rate=L.rate, precision=self.precision)

    def _deconv_fused_ok(self, B, H, W, cin, cout):
        """The one-launch transposed conv for this input ([B,H,W,cin]): emd_deconv3x3s2_fused_preferred (independent of B where the
        patch-resident kernel applies: it sums in another order than the GEMM forms)."""
        return (self.precision == ops.PREC_BF16X3 and os.environ.get("EMD_D_DECONV_FUSED", "1") != "0"
                and bool(_lib.load().emd_deconv3x3s2_fused_preferred(B, H, W, cin, cout)))

    def _split_out(self, B, H, W, Cc):
        """Output tensor of the layer in front of a transposed conv: split32 when the fused transposed conv will read it (the
        producer's epilogue splits; no fp32 tensor, no converter pass), fp32 otherwise."""
        if (self.fuse_sep and self._deconv_fused_ok(B, H, W, Cc, Cc) and Cc % 32 == 0 and H % 8 == 0 and W % 16 == 0
                and os.environ.get("EMD_D_SPLIT_OUT", "1") != "0"):
            return ops.SplitAct(B, H, W, Cc, self.device)
        return None

    def _deconv(self, key, x, out):
        L, p = self.layers[key], self.P[key]
        # measured (tools/conv_split_bench.py): converting the input (fp32 -> split32, one pass) + the LDS-DMA GEMM beats the
        # register-staged GEMM where K = taps x Cin >= 1024 per output phase (deconv2to1: 2.72 -> 0.18 + 2.17 ms)
        if self._deconv_fused_ok(x.B, x.H, x.W, L.cin, L.cout):
            # one launch, the four output phases per workgroup: the input is read from HBM once instead of four times
            xs = x if isinstance(x, ops.SplitAct) else ops.to_split32(x)
            return ops.deconv3x3s2_fused(xs, p["phases"], p["scale"], p["shift"], out)
        assert not isinstance(x, ops.SplitAct)
        if L.cin >= 256 and self._split_gemm_ok(x.B * x.H * x.W, L.cout, 4 * L.cin):
            return ops.deconv3x3s2_split32(ops.to_split32(x), p["phases"], p["scale"], p["shift"], out)
        return ops.deconv3x3s2(x, p["phases"], p["scale"], p["shift"], out, precision=self.precision)

    # ---- the graph
    def forward(self, x):
        """x: torch CUDA float32 [B,S,S,1] contiguous, S a multiple of 16 -> [B,S,S,1].
        No output clip (denoiser.py:396; the clip is applied by Denoiser.denoise_crop, :649).

        EMD_D_PIPELINE=1 (opt-in, see __init__): an even batch of >= 8 images runs as two staggered passes on two HIP
        streams: half B starts its encoder when half A has finished its own, and its 1/16-resolution flow when A has
        finished that, so that the matrix-core bound middle of one half shares the chip with the HBM-bound encoder / decoder
        of the other.  Images are independent and every kernel treats them so: same bits as the single pass."""
        import torch

        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4 and x.shape[3] == 1
        B = x.shape[0]
        if not (self.pipeline and B % 2 == 0 and B >= 8):
            return self._forward_pass(x)
        out = torch.empty_like(x)
        half = B // 2
        main = torch.cuda.current_stream(self.device)
        if self._pipe_streams is None:
            self._pipe_streams = [torch.cuda.Stream(device=self.device) for _ in range(2)]
        sa, sb = self._pipe_streams
        ev = {"enc": torch.cuda.Event(), "mid": torch.cuda.Event()}
        two = self.two_streams
        self.two_streams = False          # the halves already are the two streams
        try:
            sa.wait_stream(main)
            sb.wait_stream(main)
            with torch.cuda.stream(sa):
                self._forward_pass(x[:half], out[:half], stage=lambda name: ev[name].record(sa))
            with torch.cuda.stream(sb):
                self._forward_pass(x[half:], out[half:], stage=lambda name: sb.wait_event(ev[name]), stage_first=True)
            main.wait_stream(sa)
            main.wait_stream(sb)
        finally:
            self.two_streams = two
        return out

    def _forward_pass(self, x, out=None, stage=None, stage_first=False):
        """One pass over a batch (or half of one).  stage(name) is called at the encoder / middle-flow boundaries: after the
        stage's launches (stage_first=False: the leading half records an event) or before the NEXT stage's launches of the
        trailing half (stage_first=True: it waits for the leading half's event of the stage it is about to enter... the
        encoder of the trailing half waits for "enc", its middle flow for "mid")."""
        import torch

        B, S = x.shape[0], x.shape[1]
        assert x.shape[2] == S and S % 16 == 0 and S >= 16, "square crops with side a multiple of 16"
        dev = self.device
        if stage is not None and stage_first:
            stage("enc")
        E = lambda H, Cc: ops.Act.empty(B, H, H, Cc, dev)
        P = self.P
        S2, S4, S8, S16 = S // 2, S // 4, S // 8, S // 16
        f0, f1, f2, f3 = features0, features1, features2, features3

        # encoder 0 (:252-264).  cnn0_strided lives in the channel slice of concat1 that :365 reads it from.
        if (self.fuse_sep and os.environ.get("EMD_D_GEN", "1") != "0"
                and bool(_lib.load().emd_sep3x3_fused_supported(S, S, f0, f0, 1, 1))):
            # cnn0 = relu6(d * a + t) is an outer product of the 1-channel depthwise result d: cnn0_last's patch loader
            # rebuilds it from d (4-channel scratch, d in channel 0) instead of reading a [B,S,S,64] tensor
            pc, pl = P["cnn0"], P["cnn0_last"]
            d4 = ops.cin1(x, pc["w9"], self._unit4, self._zero4, E(S, 4), act=False)
            cnn0 = None
            cnn0_last = ops.sep_fused_gen(d4, pc["a"], pc["shift"], pl["dw"], pl["pw"], pl["scale"], pl["shift"], E(S, f0),
                                          scale2=pl.get("scale2"), shift2=pl.get("shift2"), precision=self.precision)
        else:
            cnn0 = ops.cin1(x, P["cnn0"]["w9"], P["cnn0"]["a"], P["cnn0"]["shift"], E(S, f0))
            cnn0_last = self._sep("cnn0_last", cnn0)
        residual0 = ops.cin1(x, None, P["residual0"]["a"], P["residual0"]["shift"], E(S2, f1), stride=2)
        concat1 = E(S2, f2 + f1)
        cnn0_strided = self._sep("cnn0_strided", cnn0_last, out=concat1.slice(f2, f1), res=residual0)
        del cnn0, cnn0_last, residual0
        # encoder 1 (:267-279); cnn1_strided lives in concat2 (:353)
        residual1 = self._conv1x1("residual1", cnn0_strided)
        cnn1 = self._sep("cnn1", cnn0_strided)
        cnn1_last = self._sep("cnn1_last", cnn1)
        concat2 = E(S4, aspp_output + f1)
        cnn1_strided = self._sep("cnn1_strided", cnn1_last, out=concat2.slice(aspp_output, f1), res=residual1)
        del cnn1, cnn1_last, residual1
        # encoder 2 (:282-294)
        residual2 = self._conv1x1("residual2", cnn1_strided)
        cnn2 = self._sep("cnn2", cnn1_strided)
        cnn2_last = self._sep("cnn2_last", cnn2)
        cnn2_strided = self._sep("cnn2_strided", cnn2_last, res=residual2)
        del cnn2, cnn2_last, residual2
        # encoder 3 (:297-309)
        residual3 = self._conv1x1("residual3", cnn2_strided)
        cnn3 = self._sep("cnn3", cnn2_strided)
        cnn3_last = self._sep("cnn3_last", cnn3)
        cnn3_strided = self._sep("cnn3_strided", cnn3_last, res=residual3)
        del cnn2_strided, cnn3, cnn3_last, residual3
        if stage is not None:
            stage("mid" if stage_first else "enc")
        # encoder 4 (:312-322) and the middle flow (:324-325)
        cur = self._middle_flow(cnn3_strided)
        del cnn3_strided
        t = None
        # ASPP (:152-216): the five branches write straight into their slices of the 3640-channel concat
        af = aspp_filters
        cat = E(S16, 5 * af)
        curs = (ops.to_split32(cur) if self._pw_split_ok(cur.C, af) or (self.variant != "D" and self._split_gemm_ok(B * S16 * S16, af, 9 * cur.C))
                else None)   # shared by the GEMM branches
        self._conv1x1("aspp_conv1x1", cur, out=cat.slice(0, af), xs=curs)
        if self.variant == "D":
            self._sep("aspp_small", cur, out=cat.slice(af, af))
            self._sep("aspp_medium", cur, out=cat.slice(2 * af, af))
            self._sep("aspp_large", cur, out=cat.slice(3 * af, af))
            # :185-189 the pooled tensor is discarded; :199 "pooling" = resize of the INPUT to [aspp,aspp]
            # (an identity resize here) followed by BN + relu6 (:200)
            ops.affine_relu6(cur, P["aspp_pooling_bn"]["scale"], P["aspp_pooling_bn"]["shift"], cat.slice(4 * af, af))
        else:
            # the training twin (denoiser-multi-gpu.py:306-345): dense dilated 3x3 branches, and a real image-level
            # branch: avg-pool 2x2 -> 1x1 conv + bias -> bilinear back to [aspp,aspp] -> BN -> relu6
            self._conv3x3("aspp_small", cur, out=cat.slice(af, af), xs=curs)
            self._conv3x3("aspp_medium", cur, out=cat.slice(2 * af, af), xs=curs)
            self._conv3x3("aspp_large", cur, out=cat.slice(3 * af, af), xs=curs)
            pooled = ops.avgpool2x2(cur, ops.Act.empty(B, -(-S16 // 2), -(-S16 // 2), af, dev))
            img_lvl = self._conv1x1("aspp_image_conv", pooled)
            up = ops.resize_bilinear(img_lvl, E(S16, af))
            ops.affine_relu6(up, P["aspp_pooling_bn"]["scale"], P["aspp_pooling_bn"]["shift"], cat.slice(4 * af, af))
            del pooled, img_lvl, up
        aspp = self._conv1x1("aspp_reduce", cat)
        del cur, cat, t, curs
        if stage is not None and not stage_first:
            stage("mid")
        # decoder (:350-384)
        ops.resize_bilinear(aspp, concat2.slice(0, aspp_output))            # deconv3 (:350)
        residual2_d = self._conv1x1("residual2_d", concat2)
        t = self._sep("deconv2_a", concat2)
        deconv2 = self._sep("deconv2_b", t, res=residual2_d, out=self._split_out(B, S4, S4, f2))
        del aspp, concat2, cnn1_strided, residual2_d, t
        self._deconv("deconv2to1", deconv2, concat1.slice(0, f2))
        t, residual1_d = self._sep_and_projection("deconv1_a", "residual1_d", concat1)
        deconv1 = self._sep("deconv1_b", t, res=residual1_d, out=self._split_out(B, S2, S2, f1))
        del deconv2, concat1, cnn0_strided, residual1_d, t
        deconv1to0 = self._deconv("deconv1to0", deconv1, E(S, f1))
        del deconv1
        t, residual0_d = self._sep_and_projection("deconv0_a", "residual0_d", deconv1to0)
        deconv0 = self._sep("deconv0_b", t, res=residual0_d)
        del deconv1to0, residual0_d, t
        if out is None:
            out = torch.empty((B, S, S, 1), dtype=torch.float32, device=dev)
        pf = P["deconv_final"]
        # the twin clips in-graph (denoiser-multi-gpu.py:534-538); D does not (denoiser.py:396)
        ops.conv3x3_cout1(deconv0, pf["w"], pf["scale_f"], pf["shift_f"], out, act=2 if self.variant == "Dprime" else 1)
        return out


# ------------------------------------------------------------------------------------------------
def scale0to1(img):
    """Rescale image between 0 and 1 (denoiser.py:684-695)."""
    img = np.asarray(img)
    lo, hi = np.min(img), np.max(img)
    if lo == hi:
        img = np.full(img.shape, 0.5)
    else:
        img = (img - lo) / (hi - lo)
    return img.astype(np.float32)


def _resize_bilinear_host(img, size):
    """cv2.resize(img, size) with its default INTER_LINEAR (half-pixel centres, edge clamp), in numpy:
    the reference's preprocessing step (denoiser.py:634), host side, identity for 512x512 inputs."""
    H, W = img.shape
    oh, ow = size[1], size[0]
    if (H, W) == (oh, ow):
        return img.astype(np.float32)

    def axis(n_in, n_out):
        src = (np.arange(n_out) + 0.5) * (n_in / n_out) - 0.5
        lo = np.floor(src).astype(np.int64)
        frac = src - lo
        return np.clip(lo, 0, n_in - 1), np.clip(lo + 1, 0, n_in - 1), frac

    y0, y1, fy = axis(H, oh)
    x0, x1, fx = axis(W, ow)
    img = img.astype(np.float64)
    top = img[y0][:, x0] * (1 - fx) + img[y0][:, x1] * fx
    bot = img[y1][:, x0] * (1 - fx) + img[y1][:, x1] * fx
    return (top * (1 - fy)[:, None] + bot * fy[:, None]).astype(np.float32)


class Denoiser(object):
    """Drop-in for the reference's ``Denoiser`` (machine_learning/denoiser.py:584-682)."""

    def __init__(self, checkpoint_loc=None, visible_cuda=None, precision="bf16x3", weights=None, seed=SYNTH_SEED):
        import torch

        # reference: os.environ["CUDA_VISIBLE_DEVICES"] = visible_cuda (:591); here: a device index
        idx = int(str(visible_cuda).split(",")[0]) if visible_cuda not in (None, "") else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        if weights is None:
            weights = load_weights(checkpoint_loc) if checkpoint_loc else synthetic_weights(seed)
        self.engine = DenoiserEngine(weights, self.device, precision)

    # ---- :632-643
    def preprocess(self, img):
        img = _resize_bilinear_host(np.asarray(img, dtype=np.float32), (cropsize, cropsize))
        img = scale0to1(img)
        img[np.isnan(img)] = 0.5
        img[np.isinf(img)] = 0.5
        return scale0to1(img).reshape(1, cropsize, cropsize, 1)

    def _forward_np(self, batch):
        import torch

        x = torch.from_numpy(np.ascontiguousarray(batch, dtype=np.float32)).to(self.device)
        return self.engine.forward(x).cpu().numpy()

    # ---- :645-651
    def denoise_crop(self, img, preprocess=True, postprocess=True):
        x = self.preprocess(img) if preprocess else np.asarray(img, np.float32).reshape(1, cropsize, cropsize, 1)
        pred = self._forward_np(x)
        if postprocess:
            return pred.clip(0.0, 1.0).reshape(cropsize, cropsize)
        return pred

    def denoise_batch(self, lq_batch, postprocess=False):
        """lq_batch [B,S,S,1] float32, numpy or torch (host or device) -> hq_batch, same container."""
        import torch

        is_np = isinstance(lq_batch, np.ndarray)
        x = torch.from_numpy(np.ascontiguousarray(lq_batch, dtype=np.float32)) if is_np else lq_batch
        on_dev = x.is_cuda
        y = self.engine.forward(x.to(self.device, dtype=torch.float32).contiguous())
        if postprocess:
            y = y.clamp_(0.0, 1.0)
        if is_np:
            return y.cpu().numpy()
        return y if on_dev else y.cpu()

    # ---- :653-682 (the reference's body is not executable as written: no `self`, float slice indices,
    #      `=` instead of `+=`); this implements its stated intent, with all tiles in one batch
    def denoise(self, img, preprocess=True, postprocess=True, overlap=80, max_batch=32):
        if not isinstance(img, np.ndarray) or img.ndim == 4:
            return self.denoise_batch(img, postprocess=postprocess)
        img = np.asarray(img, dtype=np.float32)
        if preprocess:
            img = self.preprocess(img).reshape(cropsize, cropsize)
        H, W = img.shape
        if H < cropsize or W < cropsize:
            raise ValueError("denoise(preprocess=False) needs an image of at least 512x512")
        num0 = (H - cropsize + (cropsize - overlap) - 1) // (cropsize - overlap) + 1 if H > cropsize else 1
        num1 = (W - cropsize + (cropsize - overlap) - 1) // (cropsize - overlap) + 1 if W > cropsize else 1
        ys = [int(round(i * (H - cropsize) / max(num0 - 1, 1))) for i in range(num0)]
        xs = [int(round(j * (W - cropsize) / max(num1 - 1, 1))) for j in range(num1)]
        tiles = [(y, x) for y in ys for x in xs]
        denoised = np.zeros((H, W), np.float64)
        contributions = np.zeros((H, W), np.float64)
        for k in range(0, len(tiles), max_batch):
            chunk = tiles[k:k + max_batch]
            batch = np.stack([img[y:y + cropsize, x:x + cropsize] for (y, x) in chunk])[..., None]
            pred = self._forward_np(batch)[..., 0]
            for (y, x), p in zip(chunk, pred):
                denoised[y:y + cropsize, x:x + cropsize] += p
                contributions[y:y + cropsize, x:x + cropsize] += 1
        denoised /= contributions
        return denoised.clip(0.0, 1.0) if postprocess else denoised


def architecture(inputs, ground_truth=None, phase=False, params=None, engine=None, trainer=None):
    """Signature of the reference's graph builder (machine_learning/denoiser.py:58-61, misc_py/denoiser-multi-gpu.py:200-203):
    ``architecture(inputs, ground_truth, phase, params) -> output``.  The graph itself is a launch sequence of an engine, so the
    object that owns the device weights is passed too (or in ``params``: {'engine': ...} / {'trainer': ...}):

    * ``phase=False`` -- inference batch norms (moving statistics, folded): ``engine`` = DenoiserEngine, -> engine.forward(inputs);
    * ``phase=True``  -- the tower's training-mode graph (batch statistics in every norm, in-graph clip; graph D',
      misc_py/denoiser-multi-gpu.py:200-540): ``trainer`` = trainer.DenoiserTrainer, -> DenoiserTrainer.forward_train(inputs).
      Like the reference's builder it only builds/evaluates the forward graph: the moving-statistics updates are the train op's
      UPDATE_OPS (get_model_fn / train_step), not a side effect of this call.

    ``ground_truth`` is accepted and unused, as in the reference (the loss lives in _tower_fn)."""
    params = params or {}
    if isinstance(params, dict):
        engine = engine if engine is not None else params.get("engine")
        trainer = trainer if trainer is not None else params.get("trainer")
    if phase:
        if trainer is None:
            raise ValueError("phase=True evaluates the training-mode graph: pass trainer=DenoiserTrainer(...)")
        return trainer.forward_train(inputs)
    if engine is None:
        raise ValueError("pass engine=DenoiserEngine(...)")
    return engine.forward(inputs)
