"""Graph S: the small separable autoencoder of the reference's ``misc_py/apply_autoencoders.py`` on MI355X, behind the
reference's own class surface.

Replaces (reference file:line):
  * ``architecture(input, encoding_features)`` (:91-187): 4 x ``strided_conv_block`` (slim.separable_convolution2d 3x3 SAME,
    strides 2/2/2/1, 64/128/256/encoding_features channels, normalizer batch norm + ``batch_then_activ`` = a second batch norm
    + relu), 3 x ``deconv_block`` (slim.conv2d_transpose k3 s2 + bias; batch norm + relu, the last one relu only), a bias-free
    3x3 conv to one channel                                    -> ``AutoencoderEngine.forward``
  * ``class Micrograph_Autoencoder`` (:312-534): ``preprocess`` (:346-358), ``denoise_crop`` (:360-383), ``denoise``
    (:385-534)                                                 -> ``Micrograph_Autoencoder`` with the same method names.

Every batch norm of this graph runs on BATCH statistics at inference too (``is_training=True`` is hard-coded, :105-116) and
the reference feeds one 160x160 crop per ``sess.run`` (:331), so the statistics are PER IMAGE: a batch of crops here gets
per-image statistics (``emd_bn_stats_images_f32``: every image reduced exactly as it would be alone), the two norms of a
separable block collapse into one per-channel affine on the device (``emd_bn_train_fold_f32``, the double-norm algebra of csrc/bn_train.hip), applied with the relu by
``emd_affine_act_images_f32``.  The convolutions are the graph-D kernels: depthwise 3x3 stride 1/2, pointwise and transposed
convolutions on the matrix cores (split-bf16), the final 3x3 -> 1 conv.  The one-channel input travels as a 4-channel tensor
(3 zero channels) and ``encoding_features`` < 4 is zero-padded to 4 channels -- zeros in, zeros out, nothing else changes.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np

from . import _lib, ops, train_ops

CROPSIZE = 160          # apply_autoencoders.py:75
BN_EPS = 1e-3           # :109
ENC_CHANNELS = (64, 128, 256)
DEC_CHANNELS = (256, 128, 64)
SYNTH_SEED = 4321


def variable_specs(encoding_features: int = 16):
    """TF variable name -> shape in creation order (no outer scope in this file, :190-196): SeparableConv2d[_k] with the
    normalizer's BatchNorm inside, BatchNorm[_k] for batch_then_activ, Conv2d_transpose[_k], Conv."""
    names = OrderedDict()
    count = {}

    def scope(base):
        k = count.get(base, 0)
        count[base] = k + 1
        return base if k == 0 else f"{base}_{k}"

    def add_bn(s, c):
        for leaf in ("beta", "gamma", "moving_mean", "moving_variance"):
            names[f"{s}/{leaf}"] = (c,)

    cin = 1
    for cout in ENC_CHANNELS + (encoding_features,):
        s = scope("SeparableConv2d")
        names[f"{s}/depthwise_weights"] = (3, 3, cin, 1)
        names[f"{s}/pointwise_weights"] = (1, 1, cin, cout)
        add_bn(f"{s}/BatchNorm", cout)
        add_bn(scope("BatchNorm"), cout)
        cin = cout
    for i, cout in enumerate(DEC_CHANNELS):
        s = scope("Conv2d_transpose")
        names[f"{s}/weights"] = (3, 3, cout, cin)
        names[f"{s}/biases"] = (cout,)
        if i < 2:
            add_bn(scope("BatchNorm"), cout)
        cin = cout
    names["Conv/weights"] = (3, 3, cin, 1)
    return names


def synthetic_weights(encoding_features: int = 16, seed: int = SYNTH_SEED):
    """Seeded weights (no checkpoint ships with the reference; its paths are local drives, :542): Xavier-uniform kernels
    (:137), small biases, gamma / beta around 1 / 0.  The moving statistics exist as variables but are never read."""
    rng = np.random.default_rng(seed)
    w = OrderedDict()
    for name, shape in variable_specs(encoding_features).items():
        leaf = name.rsplit("/", 1)[1]
        if leaf in ("depthwise_weights", "pointwise_weights", "weights"):
            rf = shape[0] * shape[1]
            lim = np.sqrt(6.0 / (rf * shape[2] + rf * shape[3]))
            w[name] = rng.uniform(-lim, lim, shape).astype(np.float32)
        elif leaf == "biases":
            w[name] = rng.uniform(-0.1, 0.1, shape).astype(np.float32)
        elif leaf == "gamma":
            w[name] = rng.uniform(0.7, 1.4, shape).astype(np.float32)
        elif leaf == "beta":
            w[name] = rng.uniform(-0.3, 0.5, shape).astype(np.float32)
        elif leaf == "moving_mean":
            w[name] = np.zeros(shape, np.float32)
        else:
            w[name] = np.ones(shape, np.float32)
    return w


def load_weights(checkpoint_loc, encoding_features: int = 16):
    """``checkpoint_loc`` as the reference's constructor takes it (:316, resolved with tf.train.latest_checkpoint, :340): a
    directory with a TensorFlow checkpoint, a checkpoint prefix, or an ``.npz`` keyed by TF variable name."""
    from . import tf_checkpoint as ckpt

    specs = variable_specs(encoding_features)
    prefix = None
    if os.path.isdir(checkpoint_loc):
        prefix = ckpt.latest_checkpoint(checkpoint_loc)
    elif os.path.exists(checkpoint_loc + ".index"):
        prefix = checkpoint_loc
    if prefix is not None:
        z = ckpt.read_checkpoint(prefix, names=list(specs))
    else:
        src = checkpoint_loc if checkpoint_loc.endswith(".npz") else os.path.join(checkpoint_loc, "autoencoder_weights.npz")
        npz = np.load(src, allow_pickle=False)
        missing = [n for n in specs if n not in npz.files]
        if missing:
            raise KeyError(f"{src}: missing variable {missing[0]}")
        z = {n: npz[n] for n in specs}
    w = OrderedDict()
    for name, shape in specs.items():
        if tuple(z[name].shape) != tuple(shape):
            raise ValueError(f"{name}: shape {z[name].shape} != {shape}")
        w[name] = np.asarray(z[name], np.float32)
    return w


def _pad_last(a, n):
    """Zero-pad the last axis to n entries."""
    if a.shape[-1] == n:
        return a
    out = np.zeros(a.shape[:-1] + (n,), a.dtype)
    out[..., : a.shape[-1]] = a
    return out


class AutoencoderEngine:
    """Weights resident on one GPU + the launch sequence of architecture() (:91-187)."""

    def __init__(self, weights, device, encoding_features: int = 16):
        import torch

        _lib.load()
        self.device = device
        self.enc = encoding_features
        d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
        p4 = lambda c: -(-c // 4) * 4
        self.sep, self.dec = [], []
        names = iter(variable_specs(encoding_features))
        specs = variable_specs(encoding_features)
        assert set(specs) == set(weights), "weights do not match variable_specs(encoding_features)"
        cin, nbn = 1, 0
        for k, cout in enumerate(ENC_CHANNELS + (encoding_features,)):
            s = "SeparableConv2d" if k == 0 else f"SeparableConv2d_{k}"
            o = "BatchNorm" if nbn == 0 else f"BatchNorm_{nbn}"
            nbn += 1
            ci, co = p4(cin), p4(cout)
            dw = np.zeros((9, ci), np.float32)
            dw[:, :cin] = weights[s + "/depthwise_weights"][..., 0].reshape(9, cin)
            pw = np.zeros((1, ci, co), np.float32)
            pw[0, :cin, :cout] = weights[s + "/pointwise_weights"][0]
            self.sep.append({
                "stride": 2 if k < 3 else 1, "cin": ci, "cout": co, "dw": d(dw), "pw": ops.PackedWeights(pw, False, device),
                "g1": d(_pad_last(weights[s + "/BatchNorm/gamma"], co)), "b1": d(_pad_last(weights[s + "/BatchNorm/beta"], co)),
                "g2": d(_pad_last(weights[o + "/gamma"], co)), "b2": d(_pad_last(weights[o + "/beta"], co)),
                "one": d(np.ones(co)), "zero": d(np.zeros(co))})
            cin = cout
        for k, cout in enumerate(DEC_CHANNELS):
            s = "Conv2d_transpose" if k == 0 else f"Conv2d_transpose_{k}"
            ci = p4(cin)
            wt = np.zeros((3, 3, cout, ci), np.float32)
            wt[..., :cin] = weights[s + "/weights"]
            e = {"cin": ci, "cout": cout, "phases": ops.pack_deconv(wt, device), "one": d(np.ones(cout)),
                 "bias": d(weights[s + "/biases"])}
            if k < 2:
                o = f"BatchNorm_{nbn}"
                nbn += 1
                e["g"], e["b"] = d(weights[o + "/gamma"]), d(weights[o + "/beta"])
            self.dec.append(e)
            cin = cout
        self.w_final = d(weights["Conv/weights"][..., 0].reshape(9, cin))
        del names

    def _norm_relu(self, r: ops.Act, gamma2, beta2, gamma1=None, beta1=None):
        """Per-image batch-statistics norm(s) + relu, in place: the statistics of every image in one pair of launches
        ([B][C]), one fold over the B*C (image, channel) pairs with the norm parameters repeated per image, one affine + relu."""
        rep = lambda v: None if v is None else v.repeat(r.B)
        mean, var = ops.bn_batch_stats_images(r)
        f = train_ops.bn_train_fold(mean, var, rep(gamma2), rep(beta2), r.H * r.W, gamma1=rep(gamma1), beta1=rep(beta1), eps=BN_EPS)
        return ops.affine_act_images(r, f["scale"], f["shift"], r, act=ops.ACT_RELU)

    def forward(self, x, trace=None):
        """x: torch CUDA float32 [B,S,S,1] contiguous, S a multiple of 8 -> [B,S,S,1] (no activation on the output, :176-184)."""
        import torch

        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4 and x.shape[3] == 1
        B, S = x.shape[0], x.shape[1]
        assert x.shape[2] == S and S % 8 == 0 and S >= 8, "square crops with side a multiple of 8"
        dev = self.device
        x4 = torch.zeros((B, S, S, 4), dtype=torch.float32, device=dev)
        x4[..., 0] = x[..., 0]
        a = ops.Act(x4)
        for L in self.sep:
            Ho = -(-a.H // L["stride"])
            dwo = ops.dw3x3(a, L["dw"], ops.Act.empty(B, Ho, Ho, L["cin"], dev), stride=L["stride"])
            r = ops.conv1x1(dwo, L["pw"], L["one"], L["zero"], ops.Act.empty(B, Ho, Ho, L["cout"], dev), act=ops.ACT_NONE)
            a = self._norm_relu(r, L["g2"], L["b2"], gamma1=L["g1"], beta1=L["b1"])
            if trace is not None:
                trace.append(a.torch()[:1].cpu().numpy())
        for k, L in enumerate(self.dec):
            out = ops.Act.empty(B, 2 * a.H, 2 * a.W, L["cout"], dev)
            if k < 2:
                r = ops.deconv3x3s2(a, L["phases"], L["one"], L["bias"], out, act=ops.ACT_NONE)
                a = self._norm_relu(r, L["g"], L["b"])
            else:
                a = ops.deconv3x3s2(a, L["phases"], L["one"], L["bias"], out, act=ops.ACT_RELU)
            if trace is not None:
                trace.append(a.torch()[:1].cpu().numpy())
        y = torch.empty((B, S, S, 1), dtype=torch.float32, device=dev)
        ops.conv3x3_cout1(a, self.w_final, 1.0, 0.0, y, act=0)
        return y


def scale0to1(img):
    """Rescale image between 0 and 1 (:236-247)."""
    img = np.asarray(img, dtype=np.float32)
    lo, hi = np.min(img), np.max(img)
    if lo == hi:
        return np.full_like(img, 0.5)
    return ((img - lo) / (hi - lo)).astype(np.float32)


class Micrograph_Autoencoder(object):
    """Drop-in for the reference class of the same name (apply_autoencoders.py:312-534)."""

    def __init__(self, checkpoint_loc=None, visible_cuda=None, encoding_features=16, weights=None, seed=SYNTH_SEED):
        import torch

        self.cropsize = CROPSIZE
        self.encoding_features = encoding_features
        if weights is None:
            weights = (load_weights(checkpoint_loc, encoding_features) if checkpoint_loc is not None
                       else synthetic_weights(encoding_features, seed))
        # reference: os.environ["CUDA_VISIBLE_DEVICES"] = visible_cuda (:320-321); here: device index
        idx = int(str(visible_cuda).split(",")[0]) if visible_cuda not in (None, "") else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self.engine = AutoencoderEngine(weights, self.device, encoding_features)

    # ---- :346-358
    def preprocess(self, img, pad_width=0):
        img = np.array(img, dtype=np.float32, copy=True)
        img[np.isnan(img)] = 0.0
        img[np.isinf(img)] = 0.0
        img = scale0to1(img)
        img = img / np.mean(img)
        img = np.pad(img, pad_width=pad_width, mode="reflect").reshape(
            img.shape[0] + 2 * pad_width, img.shape[1] + 2 * pad_width, 1)
        return img.astype(np.float32)

    def _run(self, crops):
        """crops [N,S,S] float32 numpy -> [N,S,S]."""
        import torch

        x = torch.from_numpy(np.ascontiguousarray(crops, dtype=np.float32)[..., None]).to(self.device)
        return self.engine.forward(x)[..., 0].cpu().numpy()

    # ---- :360-383
    def denoise_crop(self, crop, preprocess=True, scaling=True, postprocess=True):
        crop = np.array(crop, dtype=np.float32, copy=True)
        scale = offset = None
        if scaling:
            offset = float(np.min(crop))
            scale = float(np.mean(crop)) - offset
            if scale:
                crop = (crop - offset) / scale
            else:
                crop.fill(1.0)
        inp = self.preprocess(crop.reshape(crop.shape[0], crop.shape[1])) if preprocess else crop
        pred = self._run(np.asarray(inp).reshape(1, inp.shape[0], inp.shape[1]))[0]
        if scaling:
            pred = scale * pred + offset if scale else pred * offset / np.mean(pred)
        return pred.reshape(self.cropsize, self.cropsize) if postprocess else pred.reshape(1, *pred.shape, 1)

    # ---- :385-534
    def denoise(self, img, preprocess=True, postprocess=True, overlap=25, used_overlap=1, max_batch=64):
        """Whole-image denoising.  The reference reflect-pads by ``overlap``, walks 160-px crops at stride
        160 - 2*overlap (the last one of a row / column aligned with the end), rescales each crop to minimum 0 /
        mean 1, denoises it, maps it back, keeps its centre (dropping ``overlap - used_overlap`` pixels at each side)
        and averages where kept regions meet.  Its four copies of the crop code disagree on the rescaling
        (``scale = 1/(mean - offset)`` then ``(crop - offset)/scale``, :412-418) -- the convention of ``denoise_crop``
        (:364-381) is used for every crop here, and all crops of an image go through the GPU as batches."""
        del postprocess
        cs = self.cropsize
        img = np.asarray(img, dtype=np.float32)
        if img.ndim != 2 or min(img.shape) + 2 * overlap < cs:
            raise ValueError("denoise expects a 2-D image of at least cropsize - 2*overlap pixels per side")
        overlap = max(overlap, used_overlap)
        padded = self.preprocess(img, pad_width=overlap)[..., 0] if preprocess else np.pad(img, overlap, mode="reflect")
        H, W = padded.shape
        step = cs - 2 * overlap

        def starts(n):
            s = list(range(0, max(n - cs, 0) + 1, step))
            if s[-1] != n - cs:
                s.append(n - cs)
            return s

        pos = [(y, x) for y in starts(H) for x in starts(W)]
        crops = np.stack([padded[y:y + cs, x:x + cs] for (y, x) in pos])
        offs = crops.min(axis=(1, 2))
        scales = crops.mean(axis=(1, 2)) - offs
        flat = scales == 0
        norm = (crops - offs[:, None, None]) / np.where(flat, 1.0, scales)[:, None, None]
        norm[flat] = 1.0
        preds = np.concatenate([self._run(norm[i:i + max_batch]) for i in range(0, len(pos), max_batch)])
        preds = preds * np.where(flat, 0.0, scales)[:, None, None] + offs[:, None, None]
        acc = np.zeros((H, W), np.float64)
        cnt = np.zeros((H, W), np.float64)
        m = overlap - used_overlap
        for (y, x), pr in zip(pos, preds):
            acc[y + m:y + cs - m, x + m:x + cs - m] += pr[m:cs - m, m:cs - m]
            cnt[y + m:y + cs - m, x + m:x + cs - m] += 1
        core = (slice(overlap, H - overlap), slice(overlap, W - overlap))
        return (acc[core] / cnt[core]).astype(np.float32)
