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

from . import _lib, ops

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
        self.w_last = d(weights[conv_scope + "/weights"][..., 0].reshape(9, gen_features3))
        self.b_last = float(weights[conv_scope + "/biases"][0])
        shift, scale = (float(weights[v][0]) for v in in_vars)
        if shift != 0.0 or scale != 1.0:
            raise ValueError("the instance norm's shift/scale variables are frozen at 0/1 in the reference (:144-145)")

    def _sep(self, key, x, res=None):
        L, p = self.layers[key], self.P[key]
        Ho, Wo = (x.H - 1) // L.stride + 1, (x.W - 1) // L.stride + 1
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
        enc = ops.cin1_k7_reflect(x, p0["w49"], p0["a"], p0["shift"], ops.Act.empty(B, S, S, gen_features0, self.device))
        enc = self._sep("enc1", enc)
        n = self._sep("nin_down0", enc)
        n = self._sep("nin_down1", n)
        n = self._sep("nin_down2", n)
        for i in range(num_global_enhancer_blocks):
            n = self._middle(f"nin_mid{i}", n)
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


def generator_architecture(inputs, phase=False, params=None, train_batch_norm=None, engine=None):
    """Signature of the reference's graph builder (:133); inference only."""
    if train_batch_norm:
        raise NotImplementedError("training-mode batch norm of graph G is not built yet")
    if engine is None:
        raise ValueError("pass engine=GeneratorEngine(...)")
    return engine.forward(inputs)
