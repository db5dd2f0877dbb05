"""Graph X host side: the Xception autoencoder of misc_py/modified_Xception.py:194-654 (inference) on MI355X.

Same kernels as graph D plus what X alone needs:
  * its separable convs run ``tf.contrib.layers.batch_norm`` with the contrib defaults (is_training=True,
    scale=False; modified_Xception.py:312-314), i.e. BATCH statistics even at inference: the pointwise GEMM
    writes the raw product, ``emd_bn_stats_f32`` reduces per-channel mean / biased variance (double
    accumulation), ``emd_bn_fold_f32`` turns them into an affine on the device and ``emd_affine_act_f32``
    applies it with relu (and the residual add) in place -- no host round trip;
  * ``conv_block`` (:215-229) = dense 3x3 conv + bias -> relu -> batch norm (moving stats) -> relu, one launch of
    the 9-tap implicit GEMM with a two-stage epilogue;
  * transposed convs (:551-605) through the four-phase GEMM; plain relu everywhere (EMD_ACT_RELU).
Batch statistics are taken over the batch handed to ``forward`` (the reference computes them per tower), so a
sharded batch gives each rank its own statistics, as in the reference.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np

from . import _lib, ops
from .denoiser import BN_EPS, DATA_DIR, SYNTH_SEED

# modified_Xception.py:38-70
filters00, filters01, filters1, filters2, filters4 = 32, 64, 128, 256, 728
filters5, filters6, filters7 = 1024, 1536, 2048
numMiddleXception = 16
aspp_filters, aspp_output = 256, 32
aspp_rateSmall, aspp_rateMedium, aspp_rateLarge = 3, 6, 9
decode_channels = [728, 728, 512, 384, 256, 192, 128, 64]


class _Scope:
    """tf.variable_scope default-name uniquifier inside scope 'pellet' (modified_Xception.py:794)."""

    def __init__(self):
        self.n = {}

    def __call__(self, base):
        k = self.n.get(base, 0)
        self.n[base] = k + 1
        return f"pellet/{base}" if k == 0 else f"pellet/{base}_{k}"


class XLayer:
    def __init__(self, kind, cin, cout, scope=None, bn=None, k=1, stride=1, rate=1):
        self.kind, self.cin, self.cout, self.scope, self.bn = kind, cin, cout, scope, bn
        self.k, self.stride, self.rate = k, stride, rate

    def variables(self):
        v = OrderedDict()
        if self.kind == "conv":
            v[self.scope + "/kernel"] = (self.k, self.k, self.cin, self.cout)
            v[self.scope + "/bias"] = (self.cout,)
        elif self.kind == "deconv":
            v[self.scope + "/kernel"] = (3, 3, self.cout, self.cin)
            v[self.scope + "/bias"] = (self.cout,)
        elif self.kind == "sep":
            v[self.scope + "/depthwise_weights"] = (3, 3, self.cin, 1)
            v[self.scope + "/pointwise_weights"] = (1, 1, self.cin, self.cout)
            for leaf in ("beta", "moving_mean", "moving_variance"):   # scale=False: no gamma
                v[f"{self.scope}/BatchNorm/{leaf}"] = (self.cout,)
        if self.bn:
            for leaf in ("beta", "gamma", "moving_mean", "moving_variance"):
                v[f"{self.bn}/{leaf}"] = (self.cout,)
        return v


def declare_layers():
    """Layers of architecture() in graph-construction order (a list: middle blocks repeat the same shapes)."""
    sc = _Scope()
    L = []

    def conv(cin, cout, k=1, stride=1, rate=1, name=None, bn=True):
        scope = "pellet/" + name if name else sc("conv2d")
        L.append(XLayer("conv", cin, cout, scope, sc("BatchNorm") if bn else None, k, stride, rate))

    def sep(cin, cout, stride=1):
        L.append(XLayer("sep", cin, cout, sc("SeparableConv2d"), None, 3, stride))

    def deconv(c):
        L.append(XLayer("deconv", c, c, sc("conv2d_transpose"), sc("BatchNorm")))

    # entry flow (:356-473)
    conv(1, filters00, 3, stride=2)
    conv(filters00, filters01, 3)
    c = filters01
    for f in (filters1, filters2, filters4):
        conv(c, f, 1, stride=2)
        sep(c, f); sep(f, f); sep(f, f, stride=2)
        c = f
    for _ in range(numMiddleXception):                      # :475-491, :629-630
        sep(c, c); sep(c, c); sep(c, c)
    conv(c, filters5, 1, stride=2)                          # exit flow (:493-535)
    sep(c, filters4); sep(filters4, filters5); sep(filters5, filters5, stride=2)
    sep(filters5, filters6); sep(filters6, filters6, stride=2); sep(filters6, filters7)
    c = filters7
    conv(c, aspp_filters, 1, name="1x1")                    # ASPP (:231-299)
    conv(c, aspp_filters, 3, rate=aspp_rateSmall, name="lowRate")
    conv(c, aspp_filters, 3, rate=aspp_rateMedium, name="mediumRate")
    conv(c, aspp_filters, 3, rate=aspp_rateLarge, name="highRate")
    conv(c, aspp_filters, 1, name="imageLevel", bn=False)   # created, its output is discarded (:268-285)
    L.append(XLayer("bn", aspp_filters, aspp_filters, None, sc("BatchNorm")))
    conv(5 * aspp_filters, aspp_output, 1)
    conv(aspp_output, decode_channels[0], 1)                # decoder (:538-621)
    c = decode_channels[0]
    for _ in range(3):
        conv(c, decode_channels[1], 3)
        c = decode_channels[1]
    for ch, nblocks in zip(decode_channels[2:], (3, 3, 3, 2, 2, 2)):
        deconv(c)
        for _ in range(nblocks):
            conv(c, ch, 3)
            c = ch
    conv(c, 1, 3)
    return L


def variable_specs():
    out = OrderedDict()
    for layer in declare_layers():
        out.update(layer.variables())
    return out


def synthetic_weights(seed: int = SYNTH_SEED, bn: str = "calibrated"):
    """Seeded Xavier-uniform kernels, small biases, random gamma/beta; moving statistics of the
    batch_then_activ norms from the shipped calibration (bn='calibrated') or TF's initial values."""
    rng = np.random.default_rng(seed)
    w = OrderedDict()
    for name, shape in variable_specs().items():
        leaf = name.rsplit("/", 1)[1]
        if leaf in ("depthwise_weights", "pointwise_weights", "kernel"):
            rf = shape[0] * shape[1]
            lim = np.sqrt(6.0 / (rf * shape[2] + rf * shape[3]))
            w[name] = rng.uniform(-lim, lim, shape).astype(np.float32)
        elif leaf == "bias":
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
    # The single output channel: with gamma ~ U(0.8, 2), beta ~ U(-0.5, 1) the last norm + relu + clip (:639-641) turns the image
    # into a nearly binary one (57 % zeros, 38 % ones at 512 px), whose relative L2 against any other implementation counts
    # threshold crossings -- the oracle's own float32 run is 4.9e-3 from its float64 run there.  A gentle last affine
    # (0.15 z + 0.5 on the calibrated z) keeps the output inside (0, 1) so that parity measures the arithmetic of the network.
    # The decoder's norms (every batch_then_activ from the 1x1 that opens the decoder to the last 64-channel block, :538-621) get
    # beta = gamma: their relu then sits at +1 sigma of the calibrated activation and clips ~16 % of the units instead of ~50 %.
    # Together with the bias centring of the calibration (oracle/xception_graph.py, conv_block) this keeps the residual-free
    # decoder from compounding rounding noise 1.2-1.4x per block (see there); the encoder keeps U(-0.5, 1).
    Ls = declare_layers()
    first = next(i for i, L in enumerate(Ls) if L.kind == "conv" and L.cin == aspp_output)
    for L in Ls[first:-1]:
        if L.bn:
            w[L.bn + "/beta"] = w[L.bn + "/gamma"].copy()
    last = [n for n, sh in variable_specs().items() if n.endswith("/gamma") and tuple(sh) == (1,)]
    assert len(last) == 1
    w[last[0]] = np.full((1,), 0.15, np.float32)
    w[last[0][:-5] + "beta"] = np.full((1,), 0.5, np.float32)
    if bn == "calibrated":
        path = os.path.join(DATA_DIR, f"synth_bn_X_seed{seed}.npz")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: calibrated statistics exist only for the shipped seed; use bn='tf_init'")
        z = np.load(path, allow_pickle=False)
        for name in z.files:
            assert name in w and w[name].shape == z[name].shape, name
            w[name] = z[name].astype(np.float32)
    elif bn != "tf_init":
        raise ValueError("bn must be 'calibrated' or 'tf_init'")
    return w


def _affine(w, scope):
    g = w[scope + "/gamma"].astype(np.float64) / np.sqrt(w[scope + "/moving_variance"].astype(np.float64) + BN_EPS)
    return g, w[scope + "/beta"].astype(np.float64) - w[scope + "/moving_mean"].astype(np.float64) * g


class XceptionEngine:
    """Weights resident on one GPU + the launch sequence of modified_Xception.architecture() (inference)."""

    def __init__(self, weights, device, precision="bf16x3"):
        import torch

        _lib.load()
        self.device = device
        self.precision = {"bf16x3": ops.PREC_BF16X3, "bf16": ops.PREC_BF16}[precision]
        self.layers = declare_layers()
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
        self.P = []
        for L in self.layers:
            p = {}
            if L.kind == "conv":
                wt, bias = weights[L.scope + "/kernel"], weights[L.scope + "/bias"].astype(np.float64)
                if L.cout == 1:                                  # final conv_block(.., 1)
                    g, h = _affine(weights, L.bn)
                    p.update(w=d(wt[..., 0].reshape(9, L.cin)), pre_bias=float(bias[0]), scale_f=float(g[0]), shift_f=float(h[0]))
                else:
                    cin = L.cin
                    if cin == 1:                                 # entry conv: the 1-channel image is zero-padded to 4 channels
                        p["w9"] = d(wt[:, :, 0, :].reshape(9, L.cout))          # ... or goes through the 1-channel kernel (forward)
                        wt = np.concatenate([wt, np.zeros((L.k, L.k, 3, L.cout), np.float32)], axis=2)
                        cin = 4
                    p["pw"] = ops.PackedWeights(wt.reshape(L.k * L.k, cin, L.cout), False, device)
                    p["bias"] = d(bias)
                    p["one"] = d(np.ones(L.cout))
                    if L.bn:
                        g, h = _affine(weights, L.bn)
                        p["g"], p["h"] = d(g), d(h)                         # BN as an affine (second epilogue stage)
                        p["gs"], p["hs"] = d(g), d(bias * g + h)            # conv+bias+BN folded (no relu in between)
            elif L.kind == "deconv":
                g, h = _affine(weights, L.bn)
                p["phases"] = ops.pack_deconv(weights[L.scope + "/kernel"], device)
                p["scale"], p["shift"] = d(g), d(weights[L.scope + "/bias"].astype(np.float64) * g + h)
            elif L.kind == "sep":
                p["dw"] = d(weights[L.scope + "/depthwise_weights"][..., 0].reshape(9, L.cin))
                p["pw"] = ops.PackedWeights(weights[L.scope + "/pointwise_weights"][0], False, device)
                p["beta"] = d(weights[L.scope + "/BatchNorm/beta"])
                p["one"], p["zero"] = d(np.ones(L.cout)), d(np.zeros(L.cout))
            elif L.kind == "bn":
                g, h = _affine(weights, L.bn)
                p["scale"], p["shift"] = d(g), d(h)
            self.P.append(p)

    def forward(self, x, trace=None, teacher=None):
        """x: torch CUDA float32 [B,S,S,1], S a multiple of 64 -> [B,S,S,1] in [0,1].
        trace: a list that receives every block output as a host array (the oracle's trace order).  teacher (with trace): a list
        of reference tensors in the same order; after block i has been traced its output is REPLACED by teacher[i], so every
        block is measured on the reference's input and an error cannot hide behind, or be blamed on, its propagation."""
        import torch

        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4 and x.shape[3] == 1
        B, S = x.shape[0], x.shape[1]
        assert x.shape[2] == S and S % 64 == 0, "square crops with side a multiple of 64"
        dev, prec, RELU = self.device, self.precision, ops.ACT_RELU
        seq = list(zip(self.layers, self.P))
        pos = [0]

        class _It:   # next(it) with look-ahead: the decoder decides per layer whether its consumer takes split32 input
            def __next__(self_inner):
                if pos[0] >= len(seq):
                    raise StopIteration
                pos[0] += 1
                return seq[pos[0] - 1]
        it = _It()

        def split_ok(L, npix_in):
            """This conv / deconv layer runs on the LDS-DMA split32 GEMM (ops.conv3x3_split32 / deconv3x3s2_split32)."""
            if prec != ops.PREC_BF16X3 or L.kind not in ("conv", "deconv") or (L.kind == "conv" and L.k != 3):
                return False
            m = npix_in if L.kind == "deconv" or L.stride == 1 else npix_in // (L.stride * L.stride)
            bn = 64 if L.cout <= 64 else 128
            return L.cout >= 32 and L.cin >= 32 and (-(-m // 256)) * (-(-L.cout // bn)) >= 192

        def next_takes_split(npix_out):
            return pos[0] < len(seq) and split_ok(seq[pos[0]][0], npix_out)

        def as_split(a):
            return a if isinstance(a, ops.SplitAct) else ops.to_split32(a)

        def host(a):
            return (a.to_float() if isinstance(a, ops.SplitAct) else a.torch()).cpu().numpy()
        E = lambda H, W, Cc: ops.Act.empty(B, H, W, Cc, dev)
        assert teacher is None or trace is not None

        def forced(r):
            """Teacher forcing: the block output just traced is replaced by the reference's tensor (same storage where it is a view)."""
            if teacher is None:
                return r
            t = torch.from_numpy(np.ascontiguousarray(teacher[len(trace) - 1], dtype=np.float32)).to(dev)
            if isinstance(r, ops.SplitAct):
                return ops.Act(t.contiguous())
            r.torch().copy_(t)
            return r

        def conv_bn_relu(a, out=None):
            """tf.layers.conv2d (+bias) -> batch_then_activ: one GEMM launch.  The two entry convs (:356-372) take their own routes in
            the split-bf16 mode: 1 -> 32 (k 3, stride 2) is nine fp32 FMAs per output on emd_conv3x3_cin1_f32, writing split32; 32 -> 64
            (k 3) then runs on the patch-resident split32 kernel (csrc/conv3_pipe.hip) -- 0.57 + 0.65 ms on the 9-tap GEMM before."""
            L, p = next(it)
            if L.cin == 1 and L.k == 3 and prec == ops.PREC_BF16X3 and out is None and L.cout % 32 == 0:
                Ho, Wo = -(-S // L.stride), -(-S // L.stride)
                r = ops.conv3x3_cin1(x, p["w9"], p["gs"], p["hs"], ops.SplitAct(B, Ho, Wo, L.cout, dev), stride=L.stride, act=RELU)
                if trace is not None:
                    trace.append(host(r))
                    r = forced(r)
                return r
            Ho, Wo = -(-a.H // L.stride), -(-a.W // L.stride)
            if (L.k == 3 and L.stride == 1 and L.rate == 1 and out is None and prec == ops.PREC_BF16X3 and L.cin % 32 == 0
                    and L.cout <= 256 and Ho % 8 == 0 and Wo % 32 == 0 and (isinstance(a, ops.SplitAct) or teacher is not None)):
                r = ops.conv3x3_split32(as_split(a), p["pw"], p["gs"], p["hs"], E(Ho, Wo, L.cout), act=RELU)
                if trace is not None:
                    trace.append(r.torch().cpu().numpy())
                    r = forced(r)
                return r
            if isinstance(a, ops.SplitAct):
                a = ops.Act(a.to_float().contiguous())
            out = out or E(Ho, Wo, L.cout)
            if L.k == 1:
                r = ops.conv1x1(a, p["pw"], p["gs"], p["hs"], out, stride=L.stride, act=RELU, precision=prec)
            else:
                r = ops.conv3x3(a, p["pw"], p["gs"], p["hs"], out, stride=L.stride, rate=L.rate, act=RELU, precision=prec)
            if trace is not None:
                trace.append(r.torch().cpu().numpy())
                r = forced(r)
            return r

        def conv_block(a):
            """conv3x3 + bias -> relu -> BN -> relu (:215-229): two-stage epilogue.  Where the GEMM is matrix-core bound it
            runs from a split32 input, and writes split32 when the next layer does too (no fp32 activation in between)."""
            L, p = next(it)
            if split_ok(L, B * a.H * a.W):
                out = (ops.SplitAct(B, a.H, a.W, L.cout, dev) if next_takes_split(B * a.H * a.W) else E(a.H, a.W, L.cout))
                r = ops.conv3x3_split32(as_split(a), p["pw"], p["one"], p["bias"], out, act=RELU, scale2=p["g"], shift2=p["h"])
            else:
                r = ops.conv3x3(a, p["pw"], p["one"], p["bias"], E(a.H, a.W, L.cout), act=RELU, precision=prec,
                                scale2=p["g"], shift2=p["h"])
            if trace is not None:
                trace.append(host(r))
                r = forced(r)
            return r

        def sep(a, res=None, defer=False):
            """depthwise -> pointwise (raw) -> batch-statistics BN (beta only) -> relu [+ res] (:302-323).  defer=True (the
            output feeds nothing but the next block's depthwise conv): the norm + relu are not applied here -- the raw
            pointwise output travels with its (scale, shift) and the next depthwise kernel applies them while loading."""
            L, p = next(it)
            pre = None
            if isinstance(a, tuple):
                a, pre = a[0], (a[1], a[2])
            Ho, Wo = -(-a.H // L.stride), -(-a.W // L.stride)
            if prec == ops.PREC_BF16X3 and ops.conv1x1_split32_supported(a.B * Ho * Wo, L.cin, L.cout):
                # the batch statistics of the output come out of the GEMM epilogue: no second pass over y
                # ... and the norm is folded in the statistics' final-reduction launch (dev knob EMD_X_FOLD=0: a launch of its own)
                if os.environ.get("EMD_X_FOLD", "1") != "0":
                    y, mean, var, scale, shift = ops.sep_split32(a, p["dw"], p["pw"], p["one"], p["zero"], E(Ho, Wo, L.cout),
                                                                 stride=L.stride, act=ops.ACT_NONE, pre=pre, stats=True,
                                                                 fold=(None, p["beta"], BN_EPS))
                else:
                    y, mean, var = ops.sep_split32(a, p["dw"], p["pw"], p["one"], p["zero"], E(Ho, Wo, L.cout), stride=L.stride,
                                                   act=ops.ACT_NONE, pre=pre, stats=True)
                    scale, shift = ops.bn_fold(mean, var, None, p["beta"], BN_EPS)
            else:
                # (the narrow entry-flow layers at 256^2: no split32 form) -- the statistics come out of the fp32 GEMM's epilogue all the same
                # (emd_conv1x1_stats_f32, round 4: built for the training step; a pass over y less)
                tmp = ops.dw3x3(a, p["dw"], E(Ho, Wo, L.cin), stride=L.stride, pre=pre)
                y = E(Ho, Wo, L.cout)
                if os.environ.get("EMD_X_CONV_STATS", "1") != "0":
                    mean, var = ops.conv_stats(tmp, p["pw"], p["one"], p["zero"], y, precision=prec)
                else:   # (dev: the two-launch form; the native executor has the epilogue form only, so its bit-identity test needs the default)
                    ops.conv1x1(tmp, p["pw"], p["one"], p["zero"], y, act=ops.ACT_NONE, precision=prec)
                    mean, var = ops.bn_batch_stats(y)
                scale, shift = ops.bn_fold(mean, var, None, p["beta"], BN_EPS)
            if trace is not None:   # the oracle traces the SEP output before the residual add
                tr_out = ops.affine_act(y, scale, shift, E(Ho, Wo, L.cout), act=RELU)
                trace.append(tr_out.torch().cpu().numpy())
                if teacher is not None:   # the reference's block output (>= 0: the relu is a no-op on it) [+ res], same kernel
                    return ops.affine_act(forced(tr_out), p["one"], p["zero"], y, act=RELU, res=res)
            if defer and res is None and trace is None:
                return (y, scale, shift)
            return ops.affine_act(y, scale, shift, y, act=RELU, res=res)

        def deconv(a):
            L, p = next(it)
            if split_ok(L, B * a.H * a.W):
                out = (ops.SplitAct(B, 2 * a.H, 2 * a.W, L.cout, dev) if next_takes_split(4 * B * a.H * a.W)
                       else E(2 * a.H, 2 * a.W, L.cout))
                fn = ops.deconv3x3s2_fused if os.environ.get("EMD_D_DECONV_FUSED", "1") != "0" else ops.deconv3x3s2_split32
                r = fn(as_split(a), p["phases"], p["scale"], p["shift"], out, act=RELU)
            else:
                r = ops.deconv3x3s2(a, p["phases"], p["scale"], p["shift"], E(2 * a.H, 2 * a.W, L.cout), act=RELU, precision=prec)
            if trace is not None:
                trace.append(host(r))
                r = forced(r)
            return r

        # entry flow: the 1-channel image as a 4-channel tensor (3 zero channels) feeds the 9-tap GEMM
        if prec == ops.PREC_BF16X3:
            e = conv_bn_relu(None)                              # reads x itself (emd_conv3x3_cin1_f32)
        else:
            x4 = torch.zeros((B, S, S, 4), dtype=torch.float32, device=dev)
            x4[..., 0] = x[..., 0]
            e = conv_bn_relu(ops.Act(x4))
        e = conv_bn_relu(e)
        for _ in range(3):
            res = conv_bn_relu(e)
            m = sep(e, defer=True)
            m = sep(m, defer=True)
            e = sep(m, res=res)
        for _ in range(numMiddleXception):
            m = sep(e, defer=True)
            m = sep(m, defer=True)
            e = sep(m, res=e)
        res = conv_bn_relu(e)                                   # exit flow
        m = sep(e, defer=True)
        m = sep(m, defer=True)
        m = sep(m, res=res)
        m = sep(m, defer=True)
        m = sep(m, defer=True)
        m = sep(m)
        # ASPP: branches write into their slices of the 1280-channel concat
        af = aspp_filters
        cat = E(m.H, m.W, 5 * af)
        conv_bn_relu(m, out=cat.slice(0, af))
        conv_bn_relu(m, out=cat.slice(af, af))
        conv_bn_relu(m, out=cat.slice(2 * af, af))
        large = conv_bn_relu(m, out=cat.slice(3 * af, af))
        next(it)                                                # 'imageLevel': variables exist, output discarded (:268-285)
        L, p = next(it)                                         # pooling = batch_then_activ(conv3x3_rateLarge)
        pool = ops.affine_act(large, p["scale"], p["shift"], cat.slice(4 * af, af), act=RELU)
        if trace is not None:
            trace.append(pool.torch().cpu().numpy())
            pool = forced(pool)
        d_ = conv_bn_relu(cat)
        d_ = conv_bn_relu(d_)                                   # decoder
        for _ in range(3):
            d_ = conv_block(d_)
        for nblocks in (3, 3, 3, 2, 2, 2):
            d_ = deconv(d_)
            for _ in range(nblocks):
                d_ = conv_block(d_)
        L, p = next(it)                                         # conv_block(decoding, 1) then clip to [0,1] (:639-641)
        out = torch.empty((B, S, S, 1), dtype=torch.float32, device=dev)
        ops.conv3x3_cout1(d_, p["w"], p["scale_f"], p["shift_f"], out, act=2, pre_bias=p["pre_bias"], pre_relu=True)
        assert next(it, None) is None
        return out
