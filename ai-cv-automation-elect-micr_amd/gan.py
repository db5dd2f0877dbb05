"""Graph G host side: the in-filling GAN's generator on MI355X (inference).

Mirrors ``generator_architecture(inputs, phase, params, train_batch_norm)`` of misc_py/gan-infilling-100.py:133-374
with ``train_batch_norm`` False (moving statistics, epsilon 0.01): a 512x512 micrograph of which only 1/64 of the
pixels were measured (the rest set to -1, :1173-1182) in, the in-filled image in (-1,1) out.  Layers are declared in
the reference's graph-construction order so that every variable keeps its TensorFlow name (scopes ``GAN/Gen`` and
``GAN/Gen/reg``); both batch norms of a separable conv fold into the pointwise GEMM's epilogue together with
leaky_relu(0.2) and the residual add.  The discriminator and the adversarial training loop (:376-710, :957-1088) are
not built yet.  Python here is plumbing only; there is no CPU compute path.

Reference behaviours kept on purpose (see oracle/gan_graph.py): reflect-pad + VALID separable convs (stride 2 samples
rows 2i-1..2i+1), ``deconv_block`` passing ``pad_size`` into ``conv_block``'s ``phase`` slot (=> SAME zero padding
there), the instance norm's two non-trainable variables.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np

from . import _lib, ops, streams

# gan-infilling-100.py:40-62
gen_features0, gen_features1, gen_features2, gen_features3 = 32, 64, 64, 32
nin_features1, nin_features2, nin_features3 = 128, 256, 768
nin_features_out1, nin_features_out2, nin_features_out3 = 256, 128, 64
num_global_enhancer_blocks, num_local_enhancer_blocks = 8, 3
cropsize = 512
BN_EPS_GEN = 0.01   # :167
IN_EPS = 1e-3       # :146

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
SYNTH_SEED = 1234


class _Scope:
    def __init__(self, root):
        self.stack, self.counts = [root], {}

    def unique(self, base):
        parent = "/".join(self.stack)
        k = self.counts.get((parent, base), 0)
        self.counts[(parent, base)] = k + 1
        return f"{parent}/{base}" if k == 0 else f"{parent}/{base}_{k}"


class SepLayer:
    def __init__(self, scope, outer_bn, cin, cout, k=3, stride=1, reflect=True):
        self.scope, self.outer_bn, self.cin, self.cout = scope, outer_bn, cin, cout
        self.k, self.stride, self.reflect = k, stride, reflect

    def variables(self):
        v = OrderedDict()
        v[self.scope + "/depthwise_weights"] = (self.k, self.k, self.cin, 1)
        v[self.scope + "/pointwise_weights"] = (1, 1, self.cin, self.cout)
        for s in (self.scope + "/BatchNorm", self.outer_bn):
            for leaf in ("beta", "gamma", "moving_mean", "moving_variance"):
                v[f"{s}/{leaf}"] = (self.cout,)
        return v


def declare_layers():
    """The generator's parameterised layers keyed by role, in creation order (:341-372)."""
    sc = _Scope("GAN/Gen")
    L = OrderedDict()

    def sep(key, cin, cout, k=3, stride=1, reflect=True):
        scope = sc.unique("SeparableConv2d")
        L[key] = SepLayer(scope, sc.unique("BatchNorm"), cin, cout, k, stride, reflect)

    def middle(prefix, f):
        for j in range(3):
            sep(f"{prefix}_{j}", f, f)

    sep("enc0", 1, gen_features0, k=7)
    sep("enc1", gen_features0, gen_features1, stride=2)
    sc.stack.append("reg")
    sep("nin_down0", gen_features1, nin_features1, stride=2)
    sep("nin_down1", nin_features1, nin_features2, stride=2)
    sep("nin_down2", nin_features2, nin_features3, stride=2)
    for i in range(num_global_enhancer_blocks):
        middle(f"nin_mid{i}", nin_features3)
    sep("nin_up0", nin_features3, nin_features_out1, reflect=False)   # deconv_block: SAME (pad_size swallowed)
    sep("nin_up1", nin_features_out1, nin_features_out2, reflect=False)
    sep("nin_up2", nin_features_out2, nin_features_out3, reflect=False)
    for i in range(num_local_enhancer_blocks):
        middle(f"local{i}", gen_features2)
    sep("up", gen_features2, gen_features3, reflect=False)
    sep("last_sep", gen_features3, gen_features3)
    sc.stack.pop()
    return L, sc.unique("Conv"), [sc.unique("Variable"), sc.unique("Variable")]


def variable_specs():
    """TF variable name -> shape, in creation order."""
    L, conv_scope, in_vars = declare_layers()
    out = OrderedDict()
    for layer in L.values():
        out.update(layer.variables())
    out[conv_scope + "/weights"] = (3, 3, gen_features3, 1)
    out[conv_scope + "/biases"] = (1,)
    for v in in_vars:
        out[v] = (1,)
    return out


def synthetic_weights(seed: int = SYNTH_SEED, bn: str = "calibrated"):
    """Seeded weights (no checkpoint ships with the reference, :68 is a network share): Xavier-uniform kernels as
    :225, random batch-norm gamma/beta, moving statistics either TF's initial values (bn='tf_init') or the calibrated
    set shipped in data/ for the default seed (tests/golden/make_synth_bn.py G)."""
    rng = np.random.default_rng(seed)
    w = OrderedDict()
    specs = variable_specs()
    names = list(specs)
    for name, shape in specs.items():
        leaf = name.rsplit("/", 1)[1]
        if leaf in ("depthwise_weights", "pointwise_weights", "weights"):
            rf = shape[0] * shape[1]
            lim = np.sqrt(6.0 / (rf * shape[2] + rf * shape[3]))
            w[name] = rng.uniform(-lim, lim, shape).astype(np.float32)
        elif leaf == "biases":
            w[name] = rng.uniform(-0.1, 0.1, shape).astype(np.float32)
        elif leaf == "gamma":
            w[name] = rng.uniform(0.8, 1.6, shape).astype(np.float32)
        elif leaf == "beta":
            w[name] = rng.uniform(-0.3, 0.3, shape).astype(np.float32)
        elif leaf == "moving_mean":
            w[name] = np.zeros(shape, np.float32)
        elif leaf == "moving_variance":
            w[name] = np.ones(shape, np.float32)
        elif leaf.startswith("Variable"):   # _instance_norm: shift = zeros, scale = ones (in that order), frozen
            w[name] = np.zeros(shape, np.float32) if name == names[-2] else np.ones(shape, np.float32)
        else:
            raise AssertionError(name)
    if bn == "calibrated":
        path = os.path.join(DATA_DIR, f"synth_bn_G_seed{seed}.npz")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: calibrated statistics exist only for the shipped seed; use bn='tf_init'")
        z = np.load(path, allow_pickle=False)
        for name in z.files:
            assert name in w and w[name].shape == z[name].shape, name
            w[name] = z[name].astype(np.float32)
    elif bn != "tf_init":
        raise ValueError("bn must be 'calibrated' or 'tf_init'")
    return w


def _affine(w, layer):
    """Both batch norms of a separable conv as one (scale, shift), float64."""
    s, t = np.ones(layer.cout), np.zeros(layer.cout)
    for scope in (layer.scope + "/BatchNorm", layer.outer_bn):
        g = w[scope + "/gamma"].astype(np.float64) / np.sqrt(w[scope + "/moving_variance"].astype(np.float64) + BN_EPS_GEN)
        h = w[scope + "/beta"].astype(np.float64) - w[scope + "/moving_mean"].astype(np.float64) * g
        s, t = s * g, t * g + h
    return s, t


class GeneratorEngine:
    """Weights resident on one GPU + the launch sequence of generator_architecture (:341-372)."""

    def __init__(self, weights, device, precision="bf16x3"):
        import torch

        _lib.load()
        self.device = device
        self.precision = {"bf16x3": ops.PREC_BF16X3, "bf16": ops.PREC_BF16}[precision]
        self.layers, conv_scope, in_vars = declare_layers()
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
        self.P = {}
        for key, L in self.layers.items():
            s, t = _affine(weights, L)
            dw = weights[L.scope + "/depthwise_weights"][..., 0]
            pw = weights[L.scope + "/pointwise_weights"][0]
            if L.cin == 1:
                self.P[key] = {"w49": d(dw.reshape(49)), "a": d(pw.reshape(L.cout).astype(np.float64) * s), "shift": d(t)}
            else:
                self.P[key] = {"dw": d(dw.reshape(9, L.cin)), "pw": ops.PackedWeights(pw, False, device), "scale": d(s),
                               "shift": d(t)}
        self._unit4, self._zero4 = d(np.array([1, 0, 0, 0])), d(np.zeros(4))
        self.two_streams = os.environ.get("EMD_D_TWO_STREAMS", "1") != "0"   # streams.TwoHalves
        self._halves = streams.TwoHalves(device)
        self.w_last = d(weights[conv_scope + "/weights"][..., 0].reshape(9, gen_features3))
        self.b_last = float(weights[conv_scope + "/biases"][0])
        shift, scale = (float(weights[v][0]) for v in in_vars)
        if shift != 0.0 or scale != 1.0:
            raise ValueError("the instance norm's shift/scale variables are frozen at 0/1 in the reference (:144-145)")

    def _sep(self, key, x, res=None):
        L, p = self.layers[key], self.P[key]
        Ho, Wo = (x.H - 1) // L.stride + 1, (x.W - 1) // L.stride + 1
        if L.stride == 1 and ops.sep_fused_supported(x, L.cout, 1, 1) and x.H >= 2 and x.W >= 2:
            # one launch, the depthwise result stays in LDS (the HBM-bound <= 128-channel layers)
            out = ops.Act.empty(x.B, Ho, Wo, L.cout, self.device)
            return ops.sep_fused(x, p["dw"], p["pw"], p["scale"], p["shift"], out, act=ops.ACT_LEAKY, res=res,
                                 precision=self.precision, reflect=L.reflect)
        if (L.stride == 2 and L.reflect and self.precision == ops.PREC_BF16X3 and x.H % 2 == 0 and x.W % 2 == 0
                and os.environ.get("EMD_G_SEP_S2", "1") != "0" and ops.sep_fused_supported(x, L.cout, 2, 1)):
            # the down-sampling layers (:345-352) in one launch too (round 3: sep_pipe's stride-2 form on the REFLECT-padded image)
            out = ops.Act.empty(x.B, Ho, Wo, L.cout, self.device)
            return ops.sep_fused(x, p["dw"], p["pw"], p["scale"], p["shift"], out, act=ops.ACT_LEAKY, res=res, reflect=True, stride=2)
        if self.precision == ops.PREC_BF16X3 and ops.conv1x1_split32_supported(x.B * Ho * Wo, L.cin, L.cout):
            out = ops.Act.empty(x.B, Ho, Wo, L.cout, self.device)
            return ops.sep_split32(x, p["dw"], p["pw"], p["scale"], p["shift"], out, stride=L.stride, act=ops.ACT_LEAKY,
                                   res=res, reflect=L.reflect)
        d = ops.Act.empty(x.B, Ho, Wo, L.cin, self.device)
        if L.reflect:
            ops.dw3x3_reflect(x, p["dw"], d, stride=L.stride)
        else:
            ops.dw3x3(x, p["dw"], d, stride=L.stride)
        out = ops.Act.empty(x.B, Ho, Wo, L.cout, self.device)
        return ops.conv1x1(d, p["pw"], p["scale"], p["shift"], out, act=ops.ACT_LEAKY, res=res, precision=self.precision)

    def _middle(self, prefix, x):
        t = self._sep(prefix + "_0", x)
        t = self._sep(prefix + "_1", t)
        return self._sep(prefix + "_2", t, res=x)

    def _enhancer_chain(self, x, out):
        """The global enhancer's residual blocks (:351-352) on one batch or part of one; yields after every block."""
        n = x
        for i in range(num_global_enhancer_blocks):
            t = self._sep(f"nin_mid{i}_0", n)
            t = self._sep(f"nin_mid{i}_1", t)
            last = self._sep(f"nin_mid{i}_2", t, res=n)
            if i == num_global_enhancer_blocks - 1:
                out.torch().copy_(last.torch())
            n = last
            yield

    def _global_enhancer(self, x):
        half = x.B // 2
        L = self.layers[f"nin_mid{num_global_enhancer_blocks - 1}_2"]
        if (self.two_streams and x.B % 2 == 0 and half >= 1 and self.precision == ops.PREC_BF16X3
                and ops.conv1x1_split32_supported(half * x.H * x.W, L.cin, L.cout)):
            out = ops.Act.empty(x.B, x.H, x.W, L.cout, self.device)
            return self._halves.run(x, out, self._enhancer_chain)
        n = x
        for i in range(num_global_enhancer_blocks):
            n = self._middle(f"nin_mid{i}", n)
        return n

    def _up(self, key, x, size, res=None):
        up = ops.resize_bilinear(x, ops.Act.empty(x.B, size, size, x.C, self.device))
        return self._sep(key, up, res=res)

    def forward(self, x):
        """x: torch CUDA float32 [B,S,S,1] contiguous (missing pixels = -1), S a multiple of 16 -> [B,S,S,1] in (-1,1)."""
        import torch

        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4 and x.shape[3] == 1
        B, S = x.shape[0], x.shape[1]
        assert x.shape[2] == S and S % 16 == 0 and S >= 32
        p0 = self.P["enc0"]
        L1, p1 = self.layers["enc1"], self.P["enc1"]
        So = (S - 1) // L1.stride + 1
        if L1.reflect and L1.stride == 2 and not (self.precision == ops.PREC_BF16X3
                                                  and ops.conv1x1_split32_supported(B * So * So, L1.cin, L1.cout)):
            # enc0 = leaky(d7 * a + t) is an outer product of the 7x7 stencil of the 1-channel image: enc1's depthwise conv rebuilds
            # it from d7 (a 4-channel scratch tensor, d7 in channel 0) instead of a [B,S,S,32] tensor going out and coming back
            d4 = ops.cin1_k7_reflect(x, p0["w49"], self._unit4, self._zero4, ops.Act.empty(B, S, S, 4, self.device), act=False)
            dd = ops.dw3x3_reflect_gen(d4, p0["a"], p0["shift"], p1["dw"], ops.Act.empty(B, So, So, L1.cin, self.device),
                                       stride=L1.stride)
            enc = ops.conv1x1(dd, p1["pw"], p1["scale"], p1["shift"], ops.Act.empty(B, So, So, L1.cout, self.device),
                              act=ops.ACT_LEAKY, precision=self.precision)
        else:
            enc = ops.cin1_k7_reflect(x, p0["w49"], p0["a"], p0["shift"], ops.Act.empty(B, S, S, gen_features0, self.device))
            enc = self._sep("enc1", enc)
        n = self._sep("nin_down0", enc)
        n = self._sep("nin_down1", n)
        n = self._sep("nin_down2", n)
        n = self._global_enhancer(n)
        n = self._up("nin_up0", n, S // 8)
        n = self._up("nin_up1", n, S // 4)
        enc = self._up("nin_up2", n, S // 2, res=enc)          # enc += network_in_network(enc)  (:355)
        for i in range(num_local_enhancer_blocks):
            enc = self._middle(f"local{i}", enc)
        enc = self._up("up", enc, S)
        enc = self._sep("last_sep", enc)
        raw = torch.empty((B, S, S, 1), dtype=torch.float32, device=self.device)
        ops.conv3x3_cout1_reflect(enc, self.w_last, self.b_last, raw)
        return ops.instnorm_tanh(raw, torch.empty_like(raw), eps=IN_EPS)


def algorithmic_flops(S=cropsize):
    """Pointwise-GEMM flops of one generator forward pass on an S x S image (2*M*Cin*Cout per separable conv)."""
    L, _, _ = declare_layers()
    size = {"enc0": S, "enc1": S // 2, "nin_down0": S // 4, "nin_down1": S // 8, "nin_down2": S // 16, "nin_up0": S // 8,
            "nin_up1": S // 4, "nin_up2": S // 2, "up": S, "last_sep": S}
    fl = 0
    for key, layer in L.items():
        out = size.get(key, S // 16 if key.startswith("nin_mid") else S // 2)
        fl += 2 * out * out * layer.cin * layer.cout
    return fl


def gen_lq(img, select=None, seed=1, frac=1.0 / 64):
    """The reference's input synthesis (:1172-1182): a fixed Bernoulli(1/64) pixel mask (np.random.seed(1)) keeps the
    measured pixels, every other pixel is -1.  img [..., H, W] in [-1,1] (norm_img, :1147-1160)."""
    img = np.asarray(img, np.float32)
    if select is None:
        select = np.random.RandomState(seed).random_sample(img.shape[-2:]) < frac
    lq = -np.ones(img.shape, np.float32)
    lq[..., select] = img[..., select]
    return lq


def spiral_mask(size, frac=1.0 / 64):
    """NOT in the reference (its mask is the fixed Bernoulli(1/64) field of gen_lq, :1172-1175): a spiral-scan mask of our own, for
    BASELINE configs[4]'s wording "512x512 spiral-scan masks".  An Archimedean spiral r = p theta / (2 pi) from the centre out to the
    corners, one pixel wide, sampled more finely than a pixel so that consecutive visited pixels touch (a beam path, not a point
    cloud); the pitch p is bisected until the number of visited pixels is frac * size**2 to within one turn's granularity.
    Deterministic.  Returns a bool [size, size] array: True = measured.  Use as gen_lq(img, select=spiral_mask(S))."""
    target = frac * size * size
    c = (size - 1) / 2.0
    rmax = c * 2.0 ** 0.5

    def build(pitch):
        turns = rmax / pitch
        n = int(turns * 2.0 * np.pi * rmax * 2.0) + 16          # at least two samples per pixel of arc on the outermost turn
        th = np.sqrt(np.linspace(0.0, 1.0, n)) * turns * 2.0 * np.pi   # sqrt spacing: uniform arc-length steps
        r = pitch * th / (2.0 * np.pi)
        y = np.rint(c + r * np.sin(th)).astype(np.int64)
        x = np.rint(c + r * np.cos(th)).astype(np.int64)
        ok = (y >= 0) & (y < size) & (x >= 0) & (x < size)
        m = np.zeros((size, size), bool)
        m[y[ok], x[ok]] = True
        return m

    lo, hi = 1.0, float(size)          # pitch in pixels: small pitch = dense mask
    for _ in range(40):
        mid = 0.5 * (lo + hi)
        if build(mid).sum() > target:
            lo = mid
        else:
            hi = mid
    return build(0.5 * (lo + hi))


def generator_architecture(inputs, phase=False, params=None, train_batch_norm=None, engine=None, trainer=None):
    """Signature of the reference's graph builder (misc_py/gan-infilling-100.py:133).

    * ``train_batch_norm`` falsy -- every batch norm on its MOVING statistics (``batch_norm_on_ph: False``, what the reference feeds
      whenever it evaluates outputs, losses or gradients, :1668): ``engine`` = GeneratorEngine, -> engine.forward(inputs);
    * ``train_batch_norm`` true  -- ``is_training=True`` in every norm (:164-174): the forward pass on BATCH statistics, and the norms'
      update ops (moving <- moving - (moving - batch) * (1 - 0.9997), :866-871) applied, which is what the reference's generator train op
      runs while counter < 250 000 (:1644, :1708-1712): ``trainer`` = gan_trainer.GeneratorTrainer, ->
      GeneratorTrainer.update_moving_statistics(inputs) (its output; the inference folds are rebuilt).  The reference's batch is one
      image (:74), and so is this call's."""
    params = params or {}
    if isinstance(params, dict):
        engine = engine if engine is not None else params.get("engine")
        trainer = trainer if trainer is not None else params.get("trainer")
    if train_batch_norm:
        if trainer is None:
            raise ValueError("train_batch_norm=True runs the batch-statistics phase: pass trainer=GeneratorTrainer(...)")
        if inputs.shape[0] != 1:
            raise ValueError("the batch-statistics phase takes ONE image (batch_size = 1, gan-infilling-100.py:74)")
        return trainer.update_moving_statistics(inputs)
    if engine is None:
        raise ValueError("pass engine=GeneratorEngine(...)")
    return engine.forward(inputs)


# ================================================================================================
# Discriminator (misc_py/gan-infilling-100.py:376-710), inference: three branches (a 128-px crop; a 256-px crop
# average-pooled to 128; a 384-px crop bilinear-resized to 128, :957-980) of five stride-2 separable convs
# [BN (moving statistics) -> INSTANCE norm -> leaky_relu] at 32..512 channels, global mean, FC -> 1;
# output = sigmoid(max of the three logits) (:708).  Returns the 15 feature maps too: the generator's
# feature-matching loss walks them (:1029-1035).
# ================================================================================================
features1, features2, features3, features4, features5 = 32, 64, 128, 256, 512
BN_EPS_DISCR = 1e-3
DISCR_FEATURES = (features1, features2, features3, features4, features5)
BRANCHES = ("small", "medium", "large")


def discriminator_variable_specs():
    """TF variable name -> shape, in creation order (scopes GAN/Discr/{small,medium,large})."""
    out = OrderedDict()
    for br in BRANCHES:
        sc = _Scope("GAN/Discr/" + br)
        cin = 1
        for f in DISCR_FEATURES:
            scope = sc.unique("SeparableConv2d")
            out[scope + "/depthwise_weights"] = (3, 3, cin, 1)
            out[scope + "/pointwise_weights"] = (1, 1, cin, f)
            for leaf in ("beta", "gamma", "moving_mean", "moving_variance"):
                out[f"{scope}/BatchNorm/{leaf}"] = (f,)
            out[sc.unique("Variable")] = (f,)   # _instance_norm shift (zeros), frozen
            out[sc.unique("Variable")] = (f,)   # _instance_norm scale (ones), frozen
            cin = f
        scope = sc.unique("fully_connected")
        out[scope + "/weights"] = (features5, 1)
        out[scope + "/biases"] = (1,)
    return out


def discriminator_synthetic_weights(seed: int = SYNTH_SEED + 1):
    """Seeded discriminator weights.  Its activations are instance-normalised after every layer, so TF-initial moving
    statistics would do; mildly random ones are used so that the batch-norm fold is exercised."""
    rng = np.random.default_rng(seed)
    w = OrderedDict()
    shift_next = True
    for name, shape in discriminator_variable_specs().items():
        leaf = name.rsplit("/", 1)[1]
        if leaf in ("depthwise_weights", "pointwise_weights"):
            rf = shape[0] * shape[1]
            lim = np.sqrt(6.0 / (rf * shape[2] + rf * shape[3]))
            w[name] = rng.uniform(-lim, lim, shape).astype(np.float32)
        elif leaf == "weights":
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            w[name] = rng.uniform(-lim, lim, shape).astype(np.float32)
        elif leaf == "biases":
            w[name] = rng.uniform(-0.1, 0.1, shape).astype(np.float32)
        elif leaf == "gamma":
            w[name] = rng.uniform(0.8, 1.6, shape).astype(np.float32)
        elif leaf == "beta":
            w[name] = rng.uniform(-0.3, 0.3, shape).astype(np.float32)
        elif leaf == "moving_mean":
            w[name] = rng.uniform(-0.2, 0.2, shape).astype(np.float32)
        elif leaf == "moving_variance":
            w[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif leaf.startswith("Variable"):
            w[name] = np.zeros(shape, np.float32) if shift_next else np.ones(shape, np.float32)
            shift_next = not shift_next
        else:
            raise AssertionError(name)
    return w


def reflect_indices(n, pad):
    idx = np.abs(np.arange(-pad, n + pad))
    return np.where(idx >= n, 2 * n - 2 - idx, idx)


_RIDX_CACHE = {}


def multiscale_crops(img, offsets):
    """get_multiscale_crops (:957-980) on the device, with the tf.random_crop offsets given (parity needs them as
    inputs): img torch [B,S,S,C] -> (small [B,S/4,S/4,C], medium [B,S/2,S/2,C], large crop [B,3S/4,3S/4,C]) taken from
    the image reflect-padded by 3S/4.  Index gathers only; the large crop's resize to S/4 is done by the engine.
    offsets: ((y,x),(y,x),(y,x)) host integers, or an integer DEVICE tensor [3,2] (no host round trip: the gather
    indices are computed on the device, so a captured hipGraph can be replayed with new crops)."""
    import torch

    B, S = img.shape[0], img.shape[1]
    pad = (3 * S) // 4
    key = (S, str(img.device))
    if key not in _RIDX_CACHE:   # uploaded once (a host-to-device copy is not allowed inside stream capture)
        _RIDX_CACHE[key] = torch.from_numpy(reflect_indices(S, pad)).to(img.device)
    ridx = _RIDX_CACHE[key]
    on_device = isinstance(offsets, torch.Tensor)
    off = offsets.to(torch.int64) if on_device else None

    def crop(k, n):
        if on_device:
            ar = torch.arange(n, device=img.device)
            rows, cols = ridx[off[k, 0] + ar], ridx[off[k, 1] + ar]
        else:
            y0, x0 = offsets[k]
            rows, cols = ridx[y0:y0 + n], ridx[x0:x0 + n]
        return img[:, rows][:, :, cols].contiguous()

    return crop(0, S // 4), crop(1, S // 2), crop(2, pad)


class DiscriminatorEngine:
    """Weights resident on one GPU + the launch sequence of discriminator_architecture (:569-630, :708)."""

    def __init__(self, weights, device, precision="bf16x3"):
        import torch

        _lib.load()
        self.device = device
        self.precision = {"bf16x3": ops.PREC_BF16X3, "bf16": ops.PREC_BF16}[precision]
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
        self.P = {}
        for br in BRANCHES:
            sc = _Scope("GAN/Discr/" + br)
            cin, layers = 1, []
            for f in DISCR_FEATURES:
                scope = sc.unique("SeparableConv2d")
                shift_v, scale_v = weights[sc.unique("Variable")], weights[sc.unique("Variable")]
                if np.any(shift_v != 0) or np.any(scale_v != 1):
                    raise ValueError("the instance norm's shift/scale variables are frozen at 0/1 in the reference (:388-389)")
                dw = weights[scope + "/depthwise_weights"][..., 0]
                pw = weights[scope + "/pointwise_weights"][0]
                b = scope + "/BatchNorm"
                g = weights[b + "/gamma"].astype(np.float64) / np.sqrt(weights[b + "/moving_variance"].astype(np.float64) + BN_EPS_DISCR)
                h = weights[b + "/beta"].astype(np.float64) - weights[b + "/moving_mean"].astype(np.float64) * g
                cp = max(cin, 4)   # the 1-channel crop is processed as 4 channels (3 of zeros)
                dwp, pwp = np.zeros((9, cp), np.float32), np.zeros((1, cp, f), np.float32)
                dwp[:, :cin], pwp[:, :cin] = dw.reshape(9, cin), pw
                layers.append({"dw": d(dwp), "pw": ops.PackedWeights(pwp, False, device), "scale": d(g), "shift": d(h), "cout": f})
                cin = f
            scope = sc.unique("fully_connected")
            self.P[br] = {"layers": layers, "fc_w": d(weights[scope + "/weights"].reshape(features5)),
                          "fc_b": float(weights[scope + "/biases"][0])}

    def _instance_leaky(self, r):
        """_instance_norm (per image, per channel, eps 1e-3, unit affine) + leaky_relu(0.2), in place."""
        import torch

        for b in range(r.B):
            img = ops.Act(r.buf[b:b + 1], r.C, r.c0)
            mean, var = ops.bn_batch_stats(img)
            scale, shift = ops.bn_fold(mean, var, None, None, eps=IN_EPS)
            ops.affine_act(img, scale, shift, img, act=ops.ACT_LEAKY)
        return r

    def _branch(self, br, x4, layers_out):
        import torch

        p = self.P[br]
        x = ops.Act(x4)
        if br == "medium":
            x = ops.avgpool2x2(x, ops.Act.empty(x.B, x.H // 2, x.W // 2, 4, self.device))
        for lp in p["layers"]:
            Ho, Wo = -(-x.H // 2), -(-x.W // 2)
            dd = ops.dw3x3(x, lp["dw"], ops.Act.empty(x.B, Ho, Wo, x.C, self.device), stride=2)
            r = ops.conv1x1(dd, lp["pw"], lp["scale"], lp["shift"], ops.Act.empty(x.B, Ho, Wo, lp["cout"], self.device),
                            act=False, precision=self.precision)
            x = self._instance_leaky(r)
            layers_out.append(x.buf)
        B = x.B
        means = torch.empty((B, features5), dtype=torch.float32, device=self.device)
        for b in range(B):
            m, _ = ops.bn_batch_stats(ops.Act(x.buf[b:b + 1]))
            means[b].copy_(m)
        logit = torch.empty(B, dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().emd_fc_rows_f32(ops._p(means), features5, ops._p(p["fc_w"]), ops.C.c_float(p["fc_b"]),
                                               ops._p(logit), B, features5, _lib.stream_ptr()), "emd_fc_rows_f32")
        return logit

    def forward(self, small, medium, large_crop):
        """small [B,S/4,S/4,1], medium [B,S/2,S/2,1], large_crop [B,3S/4,3S/4,1] (torch CUDA float32, as
        multiscale_crops returns them) -> (output [B], [15 feature maps [B,h,w,c]])."""
        import torch

        def pad4(t):
            out = torch.zeros(t.shape[:3] + (4,), dtype=torch.float32, device=self.device)
            out[..., 0:1].copy_(t)
            return out

        S4 = small.shape[1]
        lg = ops.resize_bilinear(ops.Act(pad4(large_crop)), ops.Act.empty(small.shape[0], S4, S4, 4, self.device)).buf
        layers = []
        logits = [self._branch("small", pad4(small), layers), self._branch("medium", pad4(medium), layers),
                  self._branch("large", lg, layers)]
        out = torch.empty_like(logits[0])
        _lib.check(_lib.load().emd_max3_sigmoid_f32(ops._p(logits[0]), ops._p(logits[1]), ops._p(logits[2]), ops._p(out),
                                                    out.numel(), _lib.stream_ptr()), "emd_max3_sigmoid_f32")
        return out, layers
