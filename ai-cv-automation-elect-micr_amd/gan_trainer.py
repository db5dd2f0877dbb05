"""Graph G training on MI355X, discriminator side: ``_discriminator_tower_fn`` (misc_py/gan-infilling-100.py:1048-1088)
and ``_discriminator_train_op`` (:1390-1440).

One tower = ONE 512x512 image (batch_size = 1, :74): multi-scale crops (offsets are inputs, :957-980), the
discriminator with phase=True -- every separable conv is depthwise(stride 2) -> pointwise -> batch norm on batch
statistics (decay 0.9997) -> INSTANCE norm -> leaky_relu -- the head sigmoid(max(FC(mean))), the loss
``-log(clip(1 - |label - D|, 1e-8, 1 - 1e-8)) + 5e-5 * sum l2_loss(v)`` scaled by the host-computed ``adapt`` rate (:1084),
and its gradients by a hand-written reverse pass.  With one image per tower the instance norm that follows the batch
norm normalises over the same pixels, so the pair is the double-norm chain of csrc/bn_train.hip (second norm with unit
affine, same epsilon 1e-3) and is folded / differentiated by the same kernels as graph D'.
``step`` averages the towers' gradient sets (:1411-1415), clips by global norm 15 and applies Adam(beta1 = 0.5)
(:1429-1431) on flat device vectors; in a multi-GPU job the gradient vector is all-reduced as in trainer.py.
The generator-side tower (:982-1046: feature-matching loss through the discriminator, generator backward) is not built
yet.  Python here is plumbing only; there is no CPU compute path.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from . import _lib, ops
from . import train_ops as TO
from .gan import BRANCHES, DISCR_FEATURES, _Scope, discriminator_variable_specs, features5, multiscale_crops
from .trainer import sync_gradients

BN_EPS_DISCR = 1e-3
BN_DECAY_DISCR = 0.9997       # batch_decay_discr (:110)
L2_DISCR = 5e-5               # :1082
CLIP_DISCR = 15.0             # :1430-1431
ADAM_BETA1 = 0.5              # :1429


def _frozen(name):
    leaf = name.rsplit("/", 1)[1]
    return leaf.startswith("moving_") or leaf.startswith("Variable")


class DiscriminatorTrainer:
    """Discriminator parameters, gradients and Adam moments on one GPU + the tower's forward / reverse pass."""

    def __init__(self, weights, device, precision="bf16x3", learning_rate=0.0001):
        import torch

        _lib.load()
        self.device = device
        self.precision = {"bf16x3": ops.PREC_BF16X3, "bf16": ops.PREC_BF16}[precision]
        self.lr = learning_rate
        specs = discriminator_variable_specs()
        self.trainable = OrderedDict((n, s) for n, s in specs.items() if not _frozen(n))
        self.frozen = OrderedDict((n, s) for n, s in specs.items() if _frozen(n))

        def flat(names, fill=None):
            offs, n = {}, 0
            for name, shape in names.items():
                offs[name] = n
                n += -(-int(np.prod(shape)) // 4) * 4
            buf = torch.zeros(n, dtype=torch.float32, device=device)
            views = OrderedDict((name, buf[offs[name]: offs[name] + int(np.prod(shape))].view(shape)) for name, shape in names.items())
            if fill is not None:
                host = np.zeros(n, np.float32)
                for name, shape in names.items():
                    host[offs[name]: offs[name] + int(np.prod(shape))] = np.asarray(fill[name], np.float32).reshape(-1)
                buf.copy_(torch.from_numpy(host))
            return buf, views

        self.params, self.v = flat(self.trainable, weights)
        self.grads, self.g = flat(self.trainable)
        self.adam_m, _ = flat(self.trainable)
        self.adam_v, _ = flat(self.trainable)
        self.moving, self.m = flat(self.frozen, weights)
        self.t = 0
        self._pad_dirty = False
        self.ones = torch.ones(features5, dtype=torch.float32, device=device)
        self.zeros = torch.zeros(features5, dtype=torch.float32, device=device)
        self.scratch = torch.zeros(4 * features5, dtype=torch.float32, device=device)   # unit-affine "gradients", IN moving stats
        # per (branch, layer): TF scopes, padded first-layer weights, packed GEMM weights (forward, transposed)
        self.L = {}
        for br in BRANCHES:
            sc = _Scope("GAN/Discr/" + br)
            cin, layers = 1, []
            for f in DISCR_FEATURES:
                scope = sc.unique("SeparableConv2d")
                sc.unique("Variable"), sc.unique("Variable")
                cp = max(cin, 4)
                lay = {"scope": scope, "cin": cin, "cp": cp, "cout": f,
                       "pk_f": TO.DevPackedWeights(1, cp, f, device), "pk_b": TO.DevPackedWeights(1, f, cp, device)}
                if cin == 1:   # the 1-channel crop runs as 4 channels (3 of zeros): padded copies, channel 0 is real
                    lay["dw4"], lay["pw4"] = torch.zeros(9, 4, device=device), torch.zeros(1, 4, f, device=device)
                    lay["gdw4"], lay["gpw4"] = torch.zeros(9, 4, device=device), torch.zeros(1, 4, f, device=device)
                layers.append(lay)
                cin = f
            self.L[br] = {"layers": layers, "fc": sc.unique("fully_connected")}
        self.repack()

    # ---- parameters
    def _dw(self, lay):
        return lay["dw4"] if lay["cin"] == 1 else self.v[lay["scope"] + "/depthwise_weights"].view(9, lay["cin"])

    def _pw(self, lay):
        return lay["pw4"] if lay["cin"] == 1 else self.v[lay["scope"] + "/pointwise_weights"].view(1, lay["cin"], lay["cout"])

    def _gdw(self, lay):
        return lay["gdw4"] if lay["cin"] == 1 else self.g[lay["scope"] + "/depthwise_weights"].view(9, lay["cin"])

    def _gpw(self, lay):
        return lay["gpw4"] if lay["cin"] == 1 else self.g[lay["scope"] + "/pointwise_weights"].view(1, lay["cin"], lay["cout"])

    def repack(self):
        for br in BRANCHES:
            for lay in self.L[br]["layers"]:
                if lay["cin"] == 1:
                    lay["dw4"][:, 0].copy_(self.v[lay["scope"] + "/depthwise_weights"].view(9))
                    lay["pw4"][0, 0].copy_(self.v[lay["scope"] + "/pointwise_weights"].view(lay["cout"]))
                w = self._pw(lay)
                lay["pk_f"].pack(w, 1, cout_major=False)
                lay["pk_b"].pack(w.contiguous(), 1, cout_major=True)

    def zero_grad(self):
        self._pad_dirty = False
        self.grads.zero_()
        for br in BRANCHES:
            lay = self.L[br]["layers"][0]
            lay["gdw4"].zero_()
            lay["gpw4"].zero_()

    def _unpad_grads(self):
        if not self._pad_dirty:   # already folded in (and possibly regularised since): do not overwrite
            return
        self._pad_dirty = False
        for br in BRANCHES:
            lay = self.L[br]["layers"][0]
            self.g[lay["scope"] + "/depthwise_weights"].view(9).copy_(lay["gdw4"][:, 0])
            self.g[lay["scope"] + "/pointwise_weights"].view(lay["cout"]).copy_(lay["gpw4"][0, 0])

    def gradients(self):
        self._unpad_grads()
        return OrderedDict((n, t.detach().cpu().numpy().copy()) for n, t in self.g.items())

    def state_dict(self):
        out = OrderedDict()
        for name in discriminator_variable_specs():
            out[name] = (self.m if _frozen(name) else self.v)[name].detach().cpu().numpy().copy()
        return out

    # ---- one tower
    def _E(self, B, H, W, Cc):
        return ops.Act.empty(B, H, W, Cc, self.device)

    def prepare_inputs(self, small, medium, large):
        """The three crops ([1,n,n,1] torch tensors; the large one at 3S/4) -> 4-channel branch inputs at S/4 (channel 0
        real): the medium crop is average-pooled (:585-588), the large one bilinearly resized (:975)."""
        import torch

        def pad4(t):
            out = torch.zeros(t.shape[:3] + (4,), dtype=torch.float32, device=self.device)
            out[..., 0:1].copy_(t)
            return out

        S4 = small.shape[1]
        self._raw_inputs = {"medium": ops.Act(pad4(medium)), "large": ops.Act(pad4(large))}
        return {"small": ops.Act(pad4(small)),
                "medium": ops.avgpool2x2(self._raw_inputs["medium"], self._E(1, S4, S4, 4)),
                "large": ops.resize_bilinear(self._raw_inputs["large"], self._E(1, S4, S4, 4))}

    def forward(self, inputs, update_moving=False):
        """discriminator_architecture(phase=True) on the prepared inputs -> (logits [3] device tensor, saved contexts).
        saved[br] = (per-layer dicts x, d, r, fold, y; pooled mean)."""
        import torch

        saved, logits = {}, []
        for br in BRANCHES:
            x, ctxs = inputs[br], []
            for lay in self.L[br]["layers"]:
                Ho, Wo = -(-x.H // 2), -(-x.W // 2)
                d = ops.dw3x3(x, self._dw(lay), self._E(1, Ho, Wo, x.C), stride=2)
                # (round 4: the batch statistics from the GEMM's epilogue, emd_conv1x1_stats_f32 -- no pass over r of their own)
                r = self._E(1, Ho, Wo, lay["cout"])
                mean, var = ops.conv_stats(d, lay["pk_f"], self.ones, self.zeros, r, precision=self.precision)
                b = lay["scope"] + "/BatchNorm"
                f = lay["cout"]
                mv = (self.m[b + "/moving_mean"], self.m[b + "/moving_variance"], self.scratch[2 * features5: 2 * features5 + f],
                      self.scratch[3 * features5: 3 * features5 + f]) if update_moving else None
                fold = TO.bn_train_fold(mean, var, self.ones[:f], self.zeros[:f], Ho * Wo, gamma1=self.v[b + "/gamma"],
                                        beta1=self.v[b + "/beta"], moving=mv, eps=BN_EPS_DISCR, decay=BN_DECAY_DISCR)
                y = ops.affine_act(r, fold["scale"], fold["shift"], self._E(1, Ho, Wo, f), act=ops.ACT_LEAKY)
                ctxs.append({"x": x, "d": d, "r": r, "fold": fold, "y": y})
                x = y
            mean, _ = ops.bn_batch_stats(x)         # tf.reduce_mean(., [1,2]) (:578)
            fc = self.L[br]["fc"]
            logit = torch.empty(1, dtype=torch.float32, device=self.device)
            _lib.check(_lib.load().emd_fc_rows_f32(ops._p(mean), features5, ops._p(self.v[fc + "/weights"]),
                                                   ops.C.c_float(0.0), ops._p(logit), 1, features5, _lib.stream_ptr()),
                       "emd_fc_rows_f32")
            logits.append(logit + self.v[fc + "/biases"])
            saved[br] = (ctxs, mean)
        return torch.cat(logits), saved

    def backward(self, saved, dlogit, weight_grads=True, feature_targets=None, feature_weight=0.0, loss_acc=None,
                 input_grads=False):
        """Reverse pass of ``forward``.  weight_grads: accumulate parameter gradients (the discriminator's own step);
        feature_targets[br][li]: the NATURAL image's feature map of that layer -- the generator's feature-matching term
        feature_weight * mean|f - f_natural| (:1027-1035) is added to *loss_acc and its gradient to the layer's output
        gradient; input_grads: also return d loss / d (4-channel branch input) per branch."""
        out = {}
        for k, br in enumerate(BRANCHES):
            ctxs, mean = saved[br]
            last = ctxs[-1]["y"]
            fc = self.L[br]["fc"]
            if weight_grads:
                dmean = TO.fc_row_bwd(mean, self.v[fc + "/weights"].view(features5), dlogit[k:k + 1],
                                      self.g[fc + "/weights"].view(features5), self.g[fc + "/biases"])
            else:
                dmean = TO.fc_row_bwd(mean, self.v[fc + "/weights"].view(features5), dlogit[k:k + 1],
                                      self.scratch[:features5], self.scratch[features5: features5 + 1])
            dy = TO.bcast_rows(dmean, self._E(1, last.H, last.W, features5), 1.0 / (last.H * last.W))
            for li in reversed(range(len(ctxs))):
                lay, c = self.L[br]["layers"][li], ctxs[li]
                if feature_targets is not None:
                    TO.l1_feature(c["y"].buf, feature_targets[br][li], feature_weight, dy.buf, True, loss_acc)
                b = lay["scope"] + "/BatchNorm"
                f = lay["cout"]
                dg1 = self.g[b + "/gamma"] if weight_grads else self.scratch[2 * features5: 2 * features5 + f]
                dr = TO.bn_backward(dy, c["r"], c["fold"], self.ones[:f], self.scratch[:f], self.scratch[features5: features5 + f],
                                    c["r"], mask=TO.MASK_LEAKY, gamma1=self.v[b + "/gamma"], dgamma1=dg1, eps=BN_EPS_DISCR)
                if weight_grads:
                    TO.conv_wgrad(c["d"], dr, self._gpw(lay))
                dd = ops.conv1x1(dr, lay["pk_b"], self.ones, self.zeros, c["d"], act=False, precision=self.precision)
                if weight_grads:
                    TO.dw3x3_wgrad(c["x"], dd, self._gdw(lay), stride=2)
                if li > 0 or input_grads:
                    dy = TO.dw3x3_bwd_data(dd, self._dw(lay), self._E(1, c["x"].H, c["x"].W, c["x"].C), stride=2)
            if input_grads:
                out[br] = dy
        return out

    def tower(self, image, label, offsets, adapt=1.0, update_moving=True):
        """image: torch CUDA float32 [1,S,S,1]; label, adapt: host floats (the reference feeds them through placeholders,
        :1554-1556, :1729-1736).  Adds adapt * d(data loss)/d(parameters) into self.grads (the l2 term is added by
        ``step``).  -> device tensor [out, data loss]."""
        assert image.shape[0] == 1 and image.shape[3] == 1, "one image per tower (batch_size = 1 in the reference)"
        self._pad_dirty = True
        logits, saved = self.forward(self.prepare_inputs(*multiscale_crops(image, offsets)), update_moving)
        result, dlogit = TO.gan_head(logits, float(label), 0, grad_scale=float(adapt))
        self.backward(saved, dlogit)
        return result

    # ---- one optimizer step
    def step(self, images, labels, offsets, adapts=None, learning_rate=None, group=None, streams=1, lr_t_dev=None):
        """_discriminator_train_op (:1390-1440) on this rank's towers: images torch [T,S,S,1] (generated and real ones
        with their labels, :1720-1775); the towers' gradients are averaged, clipped to global norm 15 and applied by
        Adam(beta1 = 0.5).  offsets: one ((y,x),(y,x),(y,x)) per tower, or an integer device tensor [T,3,2].
        streams > 1: the towers are issued round-robin on that many HIP streams (they meet only in atomically
        accumulated gradients).  lr_t_dev: Adam's bias-corrected rate on the device (for a replayed hipGraph; the caller
        refreshes it and counts the steps).  -> device tensor [T, 2] of (out, data loss)."""
        import torch

        T = images.shape[0]
        adapts = [1.0] * T if adapts is None else list(adapts)
        self.zero_grad()
        main = torch.cuda.current_stream()
        side = _side_streams(self, min(streams, T)) if streams > 1 else []
        for s_ in side:
            s_.wait_stream(main)
        res = []
        for k in range(T):
            if side:
                with torch.cuda.stream(side[k % len(side)]):
                    r = self.tower(images[k:k + 1].contiguous(), labels[k], offsets[k], adapt=adapts[k], update_moving=(k == 0))
                    r.record_stream(main)
            else:
                r = self.tower(images[k:k + 1].contiguous(), labels[k], offsets[k], adapt=adapts[k], update_moving=(k == 0))
            res.append(r)
        for s_ in side:
            main.wait_stream(s_)
        self._unpad_grads()
        # + adapt * 5e-5 * d(sum l2_loss)/dv = adapt * 5e-5 * v, summed over the towers
        n4 = self.params.numel() // 4
        TO.axpy(ops.Act(self.params.view(1, 1, n4, 4)), ops.Act(self.grads.view(1, 1, n4, 4)), alpha=L2_DISCR * float(sum(adapts)))
        world = sync_gradients(self.grads, self.moving, group)
        scale = 1.0 / (T * world)
        gn2 = TO.sumsq(self.grads, scale=scale)
        if lr_t_dev is None:
            self.t += 1
        TO.adam_step(self.params, self.grads, self.adam_m, self.adam_v, self.t, self.lr if learning_rate is None else learning_rate,
                     beta1=ADAM_BETA1, grad_scale=scale, gnorm_sq=gn2, clip_norm=CLIP_DISCR, lr_t_dev=lr_t_dev)
        self.repack()
        return torch.stack(res)


def _side_streams(obj, n):
    import torch

    if not hasattr(obj, "_streams"):
        obj._streams = []
    while len(obj._streams) < n:
        obj._streams.append(torch.cuda.Stream(device=obj.device))
    return obj._streams[:n]


# ================================================================================================
# Generator side: _generator_tower_fn (:982-1046) and _train_op (:1330-1388).
# ================================================================================================
BN_EPS_GEN = 0.01
BN_DECAY_GEN = 0.9997        # batch_decay_gen (:112)
CLIP_GEN = 50.0                # :1379
WEIGHT_NATURAL_STATS = 12.0    # :1035


class GeneratorTrainer:
    """Generator parameters, gradients and Adam moments on one GPU + the generator tower's forward / reverse pass.

    As the reference's loop evaluates it (:1660-1680): the generator's batch norms stay on their MOVING statistics
    (``batch_norm_on_ph: False`` is fed whenever the tower gradients are evaluated, :1668 -- in BOTH phases of a run), so each
    separable conv is depthwise -> pointwise -> fixed affine -> leaky_relu and
    there is no l2 term (decay = 0, :1039); concat(output, truth) is cropped once; the discriminator (phase=True, its
    parameters frozen here) scores the generated crops and supplies 15 feature maps of the generated and of the natural
    crops; loss = -log(clip(D(fake), 1e-8, 1)) + 12 * sum_l mean|f_l(fake) - f_l(natural)|.  The reverse pass goes
    through the discriminator (data gradients only), the crops, and the generator (weight + batch-norm gamma/beta
    gradients)."""

    def __init__(self, weights, discriminator: DiscriminatorTrainer, device, precision="bf16x3", learning_rate=0.0002):
        import torch

        from . import gan as GN

        _lib.load()
        self.device, self.D = device, discriminator
        self.precision = {"bf16x3": ops.PREC_BF16X3, "bf16": ops.PREC_BF16}[precision]
        self.lr = learning_rate
        self.layers, self.conv_scope, self.in_vars = GN.declare_layers()
        specs = GN.variable_specs()
        self.trainable = OrderedDict((n, s) for n, s in specs.items() if not _frozen(n))
        self.frozen = OrderedDict((n, s) for n, s in specs.items() if _frozen(n))

        def flat(names, fill=None):
            offs, n = {}, 0
            for name, shape in names.items():
                offs[name] = n
                n += -(-int(np.prod(shape)) // 4) * 4
            buf = torch.zeros(n, dtype=torch.float32, device=device)
            views = OrderedDict((name, buf[offs[name]: offs[name] + int(np.prod(shape))].view(shape)) for name, shape in names.items())
            if fill is not None:
                host = np.zeros(n, np.float32)
                for name, shape in names.items():
                    host[offs[name]: offs[name] + int(np.prod(shape))] = np.asarray(fill[name], np.float32).reshape(-1)
                buf.copy_(torch.from_numpy(host))
            return buf, views

        self.params, self.v = flat(self.trainable, weights)
        self.grads, self.g = flat(self.trainable)
        self.adam_m, _ = flat(self.trainable)
        self.adam_v, _ = flat(self.trainable)
        self.moving, self.m = flat(self.frozen, weights)
        self.t = 0
        cmax = max(L.cout for L in self.layers.values())
        self.ones = torch.ones(cmax, dtype=torch.float32, device=device)
        self.zeros = torch.zeros(cmax, dtype=torch.float32, device=device)
        self.pw4 = torch.zeros(1, 4, GN.gen_features0, device=device)      # first layer's pointwise weights, K padded to 4
        self.gpw4 = torch.zeros_like(self.pw4)
        self.pk_f, self.pk_b, self.fold = {}, {}, {}
        for key, L in self.layers.items():
            cin = max(L.cin, 4)
            self.pk_f[key] = TO.DevPackedWeights(1, cin, L.cout, device)
            self.pk_b[key] = TO.DevPackedWeights(1, L.cout, cin, device)
        self._pad_dirty = False
        self.repack()

    # ---- parameters
    def _pw(self, key):
        L = self.layers[key]
        return self.pw4 if L.cin == 1 else self.v[L.scope + "/pointwise_weights"].view(1, L.cin, L.cout)

    def _gpw(self, key):
        L = self.layers[key]
        return self.gpw4 if L.cin == 1 else self.g[L.scope + "/pointwise_weights"].view(1, L.cin, L.cout)

    def _dw(self, key):
        L = self.layers[key]
        return self.v[L.scope + "/depthwise_weights"].view(L.k * L.k, L.cin)

    def _gdw(self, key):
        L = self.layers[key]
        return self.g[L.scope + "/depthwise_weights"].view(L.k * L.k, L.cin)

    def repack(self):
        """Parameters -> packed bf16 weights (both orientations), flipped depthwise taps, and the per-layer folds of the
        two inference-mode batch norms (all on the device; runs after every optimizer step)."""
        L0 = self.layers["enc0"]
        self.pw4[0, 0].copy_(self.v[L0.scope + "/pointwise_weights"].view(L0.cout))
        if not hasattr(self, "dw_flip"):
            self.dw_flip = {}
        for key, L in self.layers.items():
            w = self._pw(key)
            self.pk_f[key].pack(w, 1, cout_major=False)
            self.pk_b[key].pack(w.contiguous(), 1, cout_major=True)
            b1, b2 = L.scope + "/BatchNorm", L.outer_bn
            self.fold[key] = TO.bn_infer_fold2(self.v[b1 + "/gamma"], self.v[b1 + "/beta"], self.m[b1 + "/moving_mean"],
                                               self.m[b1 + "/moving_variance"], self.v[b2 + "/gamma"], self.v[b2 + "/beta"],
                                               self.m[b2 + "/moving_mean"], self.m[b2 + "/moving_variance"], BN_EPS_GEN,
                                               out=self.fold.get(key))
            if not L.reflect and L.stride == 1:
                flipped = self._dw(key).flip(0)
                if key not in self.dw_flip:
                    self.dw_flip[key] = flipped.contiguous()
                else:
                    self.dw_flip[key].copy_(flipped)

    def zero_grad(self):
        self._pad_dirty = False
        self.grads.zero_()
        self.gpw4.zero_()

    def _unpad_grads(self):
        if not self._pad_dirty:
            return
        self._pad_dirty = False
        L0 = self.layers["enc0"]
        self.g[L0.scope + "/pointwise_weights"].view(L0.cout).copy_(self.gpw4[0, 0])

    def gradients(self):
        self._unpad_grads()
        return OrderedDict((n, t.detach().cpu().numpy().copy()) for n, t in self.g.items())

    def state_dict(self):
        from . import gan as GN

        out = OrderedDict()
        for name in GN.variable_specs():
            out[name] = (self.m if _frozen(name) else self.v)[name].detach().cpu().numpy().copy()
        return out

    # ---- forward blocks
    def _E(self, B, H, W, Cc):
        return ops.Act.empty(B, H, W, Cc, self.device)

    def _sep_fwd(self, key, x, res=None, x_img=None):
        L, f = self.layers[key], self.fold[key]
        if L.cin == 1:     # 7x7 depthwise on the image -> channel 0 of a 4-channel tensor -> K = 4 pointwise GEMM
            d = TO.dw7_c1_reflect(x_img, self._dw(key).view(49), self._E(x_img.shape[0], x_img.shape[1], x_img.shape[2], 4))
            Ho, Wo = d.H, d.W
        else:
            Ho, Wo = (x.H - 1) // L.stride + 1, (x.W - 1) // L.stride + 1
            d = self._E(x.B, Ho, Wo, L.cin)
            if L.reflect:
                ops.dw3x3_reflect(x, self._dw(key), d, stride=L.stride)
            else:
                ops.dw3x3(x, self._dw(key), d, stride=L.stride)
        if self._batch_stats:
            # the train op's update ops (update_moving_statistics): both norms of the block on the BATCH statistics of this tower's
            # image (the second norm's follow analytically from the first's, as in graph D'), moving averages assigned in the same launch
            b1, b2 = L.scope + "/BatchNorm", L.outer_bn
            r = self._E(d.B, Ho, Wo, L.cout)
            mean, var = ops.conv_stats(d, self.pk_f[key], self.ones, self.zeros, r, precision=self.precision)   # statistics from the GEMM's epilogue
            f = TO.bn_train_fold(mean, var, self.v[b2 + "/gamma"], self.v[b2 + "/beta"], d.B * Ho * Wo, gamma1=self.v[b1 + "/gamma"],
                                 beta1=self.v[b1 + "/beta"], eps=BN_EPS_GEN, decay=BN_DECAY_GEN,
                                 moving=(self.m[b1 + "/moving_mean"], self.m[b1 + "/moving_variance"], self.m[b2 + "/moving_mean"],
                                         self.m[b2 + "/moving_variance"]))
        else:
            r = ops.conv1x1(d, self.pk_f[key], self.ones, self.zeros, self._E(d.B, Ho, Wo, L.cout), act=False, precision=self.precision)
        y = ops.affine_act(r, f["scale"], f["shift"], self._E(d.B, Ho, Wo, L.cout), act=ops.ACT_LEAKY, res=res)
        return y, {"x": x, "d": d, "r": r, "x_img": x_img}

    def _sep_bwd(self, key, dy, ctx, need_dx=True):
        """dy = d loss / d (layer output, before the residual add's other branch) -> d loss / d (layer input)."""
        import torch

        L, f = self.layers[key], self.fold[key]
        r, d, x = ctx["r"], ctx["d"], ctx["x"]
        Cc = L.cout
        dev = self.device
        s1, t1, t2 = (torch.empty(Cc, dtype=torch.float32, device=dev) for _ in range(3))
        TO.chan_reduce(dy, s1, r, f["mprime"], f["rprime"], t2, f["scale"], f["shift"], TO.MASK_LEAKY)
        TO.chan_reduce(dy, s1, r, f["mean1"], f["rstd1"], t1, f["scale"], f["shift"], TO.MASK_LEAKY)
        b1, b2 = L.scope + "/BatchNorm", L.outer_bn
        TO.bn_infer_grads(s1, t1, t2, f["a2"], self.g[b1 + "/gamma"], self.g[b1 + "/beta"], self.g[b2 + "/gamma"], self.g[b2 + "/beta"])
        z = self.zeros[:Cc]
        _lib.check(_lib.load().emd_bn_bwd_apply_f32(dy.ptr, dy.ld, r.ptr, r.ld, ops._p(f["scale"]), ops._p(z), ops._p(z), ops._p(z),
                                                    ops._p(f["scale"]), ops._p(f["shift"]), TO.MASK_LEAKY, r.ptr, r.ld,
                                                    ops.C.c_long(r.B * r.H * r.W), Cc, _lib.stream_ptr()), "emd_bn_bwd_apply_f32")
        dr = r
        TO.conv_wgrad(d, dr, self._gpw(key))
        dd = ops.conv1x1(dr, self.pk_b[key], self.ones, self.zeros, d, act=False, precision=self.precision)
        if L.cin == 1:
            TO.dw7_c1_reflect_wgrad(ctx["x_img"], dd, self._gdw(key).view(49))
            return None
        if L.reflect:
            TO.dw3x3_reflect_wgrad(x, dd, self._gdw(key), stride=L.stride)
        else:
            TO.dw3x3_wgrad(x, dd, self._gdw(key), stride=L.stride)
        if not need_dx:
            return None
        dx = self._E(x.B, x.H, x.W, x.C)
        if L.reflect:
            TO.dw3x3_reflect_bwd_data(dd, self._dw(key), dx, stride=L.stride)
        elif L.stride == 1:
            ops.dw3x3(dd, self.dw_flip[key], dx)
        else:
            TO.dw3x3_bwd_data(dd, self._dw(key), dx, stride=L.stride)
        return dx

    _batch_stats = False   # True only inside update_moving_statistics

    def _generator_forward(self, lq):
        """generator_architecture (:133-374) on one image [1,S,S,1] with the trainer's parameters -> saved contexts + output."""
        import torch

        from . import gan as GN

        S = lq.shape[1]
        dev = self.device
        C = {}
        # ---------------- generator forward (:341-372)
        enc0, C["enc0"] = self._sep_fwd("enc0", None, x_img=lq)
        enc1, C["enc1"] = self._sep_fwd("enc1", enc0)
        n, C["nin_down0"] = self._sep_fwd("nin_down0", enc1)
        n, C["nin_down1"] = self._sep_fwd("nin_down1", n)
        n, C["nin_down2"] = self._sep_fwd("nin_down2", n)

        def middle_fwd(prefix, x):
            t, C[prefix + "_0"] = self._sep_fwd(prefix + "_0", x)
            t, C[prefix + "_1"] = self._sep_fwd(prefix + "_1", t)
            y, C[prefix + "_2"] = self._sep_fwd(prefix + "_2", t, res=x)
            return y

        for i in range(GN.num_global_enhancer_blocks):
            n = middle_fwd(f"nin_mid{i}", n)

        def up_fwd(key, x, size, res=None):
            up = ops.resize_bilinear(x, self._E(x.B, size, size, x.C))
            y, C[key] = self._sep_fwd(key, up, res=res)
            C[key]["pre_resize"] = x
            return y

        n = up_fwd("nin_up0", n, S // 8)
        n = up_fwd("nin_up1", n, S // 4)
        enc = up_fwd("nin_up2", n, S // 2, res=enc1)                     # enc += network_in_network(enc)  (:355)
        for i in range(GN.num_local_enhancer_blocks):
            enc = middle_fwd(f"local{i}", enc)
        enc = up_fwd("up", enc, S)
        last, C["last_sep"] = self._sep_fwd("last_sep", enc)
        w_last = self.v[self.conv_scope + "/weights"].view(9, GN.gen_features3)
        raw = torch.empty((1, S, S, 1), dtype=torch.float32, device=dev)
        # the conv bias sits in front of an instance norm: it cancels in the output and its gradient is zero
        ops.conv3x3_cout1_reflect(last, w_last, 0.0, raw)
        rawa = ops.Act(raw)
        mean, var = ops.bn_batch_stats(rawa)
        fold_in = TO.bn_train_fold(mean, var, self.ones[:1], self.zeros[:1], S * S, eps=GN.IN_EPS)
        out = torch.empty_like(raw)
        _lib.check(_lib.load().emd_instnorm_tanh_f32(ops._p(raw), ops._p(mean), ops._p(var), ops._p(out), 1, ops.C.c_long(S * S),
                                                     ops.C.c_float(GN.IN_EPS), _lib.stream_ptr()), "emd_instnorm_tanh_f32")
        return C, out, raw, rawa, mean, var, fold_in, last

    def update_moving_statistics(self, lq):
        """What the reference's generator train op does besides applying the gradients while ``train_batch_norm_on`` (counter <
        250 000, :1644, :1708-1712): the update ops of tower 0's batch norms (:866-871, :1384) run with batch_norm_on_ph = True -- one
        forward pass of the generator on tower 0's image with every norm on BATCH statistics (each layer fed by the batch-normalised
        output of the one before), assigning moving <- moving - (moving - batch) * (1 - 0.9997), the variance Bessel-corrected as the
        fused batch norm does.  No gradient passes through it: the tower gradients themselves are evaluated with batch_norm_on_ph
        False in both phases (:1668).  Afterwards the inference folds are rebuilt (repack)."""
        self._batch_stats = True
        try:
            out = self._generator_forward(lq[0:1].contiguous())[1]
        finally:
            self._batch_stats = False
        self.repack()
        return out

    # ---- one tower
    def tower(self, lq, truth, offsets):
        """lq, truth: torch CUDA float32 [1,S,S,1] (missing pixels of lq = -1).  Adds the generator gradients into
        self.grads.  -> (output [1,S,S,1], device tensor [D(fake), adversarial loss], device tensor [stat loss * 12])."""
        import torch

        from . import gan as GN

        assert lq.shape[0] == 1 and lq.shape[3] == 1 and truth.shape == lq.shape
        S = lq.shape[1]
        dev = self.device
        self._pad_dirty = True
        C, out, raw, rawa, mean, var, fold_in, last = self._generator_forward(lq)
        w_last = self.v[self.conv_scope + "/weights"].view(9, GN.gen_features3)
        # ---------------- discriminator on the generated and on the natural crops (same offsets, :1008-1015)
        D = self.D
        crops_f = multiscale_crops(out, offsets)
        crops_n = multiscale_crops(truth, offsets)
        in_f = D.prepare_inputs(*crops_f)
        raw_f = D._raw_inputs
        logits_f, saved_f = D.forward(in_f)
        _, saved_n = D.forward(D.prepare_inputs(*crops_n))
        targets = {br: [c["y"].buf for c in saved_n[br][0]] for br in BRANCHES}
        result, dlogit = TO.gan_head(logits_f, 0.0, 1)
        stat = torch.zeros(1, dtype=torch.float32, device=dev)
        din = D.backward(saved_f, dlogit, weight_grads=False, feature_targets=targets, feature_weight=WEIGHT_NATURAL_STATS,
                         loss_acc=stat, input_grads=True)
        # ---------------- crops backward -> d loss / d output
        dout = torch.zeros((1, S, S, 1), dtype=torch.float32, device=dev)
        S4, S2, S34 = S // 4, S // 2, (3 * S) // 4
        if isinstance(offsets, torch.Tensor):   # int32 device tensor [3,2]: the kernel reads the offsets itself
            off32 = offsets.to(torch.int32).contiguous()
            where = [dict(y0=0, x0=0, yx_dev=off32[k]) for k in range(3)]
        else:
            where = [dict(y0=y, x0=x) for (y, x) in offsets]
        TO.crop_scatter(din["small"].buf, 4, dout, n=S4, S=S, **where[0])
        dmed = TO.avgpool2x2_bwd(din["medium"], self._E(1, S2, S2, 4))
        TO.crop_scatter(dmed.buf, 4, dout, n=S2, S=S, **where[1])
        dlarge = TO.resize_bilinear_bwd(din["large"], self._E(1, S34, S34, 4))
        TO.crop_scatter(dlarge.buf, 4, dout, n=S34, S=S, **where[2])
        del raw_f
        # ---------------- generator backward
        g = TO.tanh_bwd(dout, out)
        scr = torch.zeros(2, dtype=torch.float32, device=dev)
        draw = TO.bn_backward(ops.Act(g), rawa, fold_in, self.ones[:1], scr[0:1], scr[1:2], ops.Act(g), mask=TO.MASK_NONE,
                              eps=GN.IN_EPS)
        TO.conv3x3_cout1_reflect_wgrad(last, draw.buf, self.g[self.conv_scope + "/weights"].view(9, GN.gen_features3))
        TO.chan_reduce(draw, self.g[self.conv_scope + "/biases"], accumulate_s1=True)
        dy = TO.conv3x3_cout1_reflect_bwd_data(draw.buf, w_last, self._E(1, S, S, GN.gen_features3))
        dy = self._sep_bwd("last_sep", dy, C["last_sep"])

        def up_bwd(key, dy):
            dup = self._sep_bwd(key, dy, C[key])
            x = C[key]["pre_resize"]
            return TO.resize_bilinear_bwd(dup, self._E(x.B, x.H, x.W, x.C))

        def middle_bwd(prefix, dy):      # y = sep2(sep1(sep0(x))) + x
            t = self._sep_bwd(prefix + "_2", dy, C[prefix + "_2"])
            t = self._sep_bwd(prefix + "_1", t, C[prefix + "_1"])
            t = self._sep_bwd(prefix + "_0", t, C[prefix + "_0"])
            return TO.axpy(t, dy)        # + the identity branch

        dy = up_bwd("up", dy)
        for i in reversed(range(GN.num_local_enhancer_blocks)):
            dy = middle_bwd(f"local{i}", dy)
        d_enc1 = dy                                        # enc1 + nin(enc1): the identity branch
        dn = up_bwd("nin_up2", dy)
        dn = up_bwd("nin_up1", dn)
        dn = up_bwd("nin_up0", dn)
        for i in reversed(range(GN.num_global_enhancer_blocks)):
            dn = middle_bwd(f"nin_mid{i}", dn)
        dn = self._sep_bwd("nin_down2", dn, C["nin_down2"])
        dn = self._sep_bwd("nin_down1", dn, C["nin_down1"])
        dn = self._sep_bwd("nin_down0", dn, C["nin_down0"])
        d_enc1 = TO.axpy(dn, d_enc1)
        d_enc0 = self._sep_bwd("enc1", d_enc1, C["enc1"])
        self._sep_bwd("enc0", d_enc0, C["enc0"], need_dx=False)
        return out, result, stat

    # ---- one optimizer step
    def step(self, lq, truth, offsets, learning_rate=None, group=None):
        """_train_op (:1330-1388) on this rank's towers: lq, truth torch [T,S,S,1]; the towers' gradient sets are averaged,
        clipped to global norm 50 and applied by Adam(beta1 = 0.5).  -> device tensor [T, 3] (D(fake), -log D, 12*stat)."""
        import torch

        T = lq.shape[0]
        self.zero_grad()
        res = []
        for k in range(T):
            _, r, st = self.tower(lq[k:k + 1].contiguous(), truth[k:k + 1].contiguous(), offsets[k])
            res.append(torch.cat([r, st]))
        self._unpad_grads()
        world = sync_gradients(self.grads, self.moving, group)
        scale = 1.0 / (T * world)
        gn2 = TO.sumsq(self.grads, scale=scale)
        self.t += 1
        TO.adam_step(self.params, self.grads, self.adam_m, self.adam_v, self.t, self.lr if learning_rate is None else learning_rate,
                     beta1=ADAM_BETA1, grad_scale=scale, gnorm_sq=gn2, clip_norm=CLIP_GEN)
        self.repack()
        return torch.stack(res)


def gan_iteration(G: GeneratorTrainer, D: DiscriminatorTrainer, lq, truth, offsets, lr_gen=0.0002, label_real=1.0, label_fake=0.0,
                  adapts=None, group=None, streams=1, lr_t_dev=None, labels=None, train="both", batch_norm_on=False):
    """One iteration of the reference's training loop (:1650-1790), deterministic parts: (1) the generator towers on
    this rank's [T,S,S,1] batch and the generator's Adam step (:1660-1700); (2) the discriminator trained on the T
    generated images (label_fake) and the T natural ones (label_real) with learning rate lr_gen/2 (:1645) -- 2T towers,
    one Adam step (:1720-1790).  The reference's random label flips and its ``adapt`` heuristics are host-side choices:
    pass ``adapts`` / labels to reproduce them.  offsets: per-tower host tuples or an integer device tensor [T,3,2].
    streams: HIP streams the towers of each phase are spread over.  lr_t_dev: device tensor [2] with the two
    bias-corrected Adam rates (generator, discriminator) when the iteration runs inside a replayed hipGraph (GanLoop).
    labels: 2T per-image labels (generated images first) instead of label_fake / label_real -- what emdenoise.gan_policy.GanPolicy
    draws (:1733-1737, :1772-1776).  train: "both" (default), "gen" (:1700-1703: the generator's train op only; the discriminator
    is not trained this iteration and the second result is None) or "discr" (:1704-1806: the generator towers still run -- they
    produce the images the discriminator is shown -- but its optimizer step is skipped): the reference trains ONE of the two
    per iteration, chosen by GanPolicy.observe.  batch_norm_on (GanPolicy.batch_norm_on(counter): True for the first 250 000
    iterations, :1644): the generator's train op also refreshes its moving statistics from tower 0's batch statistics
    (GeneratorTrainer.update_moving_statistics) -- the gradients are on moving statistics either way, as in the reference.
    -> (generator results [T,3], discriminator results [2T,2] or None)."""
    import torch

    T = lq.shape[0]
    G.zero_grad()
    main = torch.cuda.current_stream()
    side = _side_streams(G, min(streams, T)) if streams > 1 else []
    for s_ in side:
        s_.wait_stream(main)
    outs, res_g = [], []
    for k in range(T):
        if side:
            with torch.cuda.stream(side[k % len(side)]):
                out, r, st = G.tower(lq[k:k + 1].contiguous(), truth[k:k + 1].contiguous(), offsets[k])
                rr = torch.cat([r, st])
                out.record_stream(main)
                rr.record_stream(main)
        else:
            out, r, st = G.tower(lq[k:k + 1].contiguous(), truth[k:k + 1].contiguous(), offsets[k])
            rr = torch.cat([r, st])
        outs.append(out)
        res_g.append(rr)
    for s_ in side:
        main.wait_stream(s_)
    assert train in ("both", "gen", "discr")
    if train != "discr":
        G._unpad_grads()
        world = sync_gradients(G.grads, G.moving, group)
        scale = 1.0 / (T * world)
        gn2 = TO.sumsq(G.grads, scale=scale)
        if lr_t_dev is None:
            G.t += 1
        if batch_norm_on:   # the update ops read the variables the towers were evaluated with (tf.group leaves the order open; this one is defined)
            G.update_moving_statistics(lq)
        TO.adam_step(G.params, G.grads, G.adam_m, G.adam_v, G.t, lr_gen, beta1=ADAM_BETA1, grad_scale=scale, gnorm_sq=gn2, clip_norm=CLIP_GEN,
                     lr_t_dev=None if lr_t_dev is None else lr_t_dev[0:1])
        G.repack()
    if train == "gen":
        return torch.stack(res_g), None
    images = torch.cat(outs + [truth[k:k + 1] for k in range(T)])
    if labels is None:
        labels = [label_fake] * T + [label_real] * T
    assert len(labels) == 2 * T
    offs2 = torch.cat([offsets, offsets]) if isinstance(offsets, torch.Tensor) else list(offsets) + list(offsets)
    res_d = D.step(images, labels, offs2, adapts=adapts, learning_rate=lr_gen / 2, group=group, streams=streams,
                   lr_t_dev=None if lr_t_dev is None else lr_t_dev[1:2])
    return torch.stack(res_g), res_d


class GanLoop:
    """gan_iteration captured once into a hipGraph and replayed: inputs, crop offsets and the two Adam rates live in
    static device buffers that are refreshed before every replay (new random crops and the bias correction do not need a
    re-capture); the towers of each phase run on ``streams`` HIP streams inside the graph.  Labels (label_fake /
    label_real) and adapt = 1 are constants of the captured graph.  Single-process use (a captured graph cannot hold
    the gradient all-reduce)."""

    def __init__(self, G: GeneratorTrainer, D: DiscriminatorTrainer, streams=4, label_real=1.0, label_fake=0.0):
        self.G, self.D, self.streams = G, D, streams
        self.labels = (label_real, label_fake)
        self.graph = None

    def iteration(self, lq, truth, offsets, lr_gen=0.0002):
        """lq, truth: torch CUDA [T,S,S,1]; offsets: per-tower ((y,x),(y,x),(y,x)) host integers."""
        import torch

        G, D = self.G, self.D
        dev = G.device
        off = torch.tensor(np.asarray(offsets, np.int32).reshape(lq.shape[0], 3, 2), dtype=torch.int32)
        lr = torch.tensor([TO.adam_lr_t(lr_gen, G.t + 1, ADAM_BETA1), TO.adam_lr_t(lr_gen / 2, D.t + 1, ADAM_BETA1)], dtype=torch.float32)
        if self.graph is None:
            self.s_lq, self.s_truth = torch.empty_like(lq), torch.empty_like(truth)
            self.s_off, self.s_lr = off.to(dev), lr.to(dev)
            # one eager pass outside capture: every kernel family is loaded (state is restored afterwards)
            keep = [t.clone() for t in (G.params, G.adam_m, G.adam_v, G.moving, D.params, D.adam_m, D.adam_v, D.moving)]
            self.s_lq.copy_(lq)
            self.s_truth.copy_(truth)
            gan_iteration(G, D, self.s_lq, self.s_truth, self.s_off, lr_gen, *self.labels, streams=self.streams, lr_t_dev=self.s_lr)
            for t, k in zip((G.params, G.adam_m, G.adam_v, G.moving, D.params, D.adam_m, D.adam_v, D.moving), keep):
                t.copy_(k)
            G.repack()
            D.repack()
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.res = gan_iteration(G, D, self.s_lq, self.s_truth, self.s_off, lr_gen, *self.labels, streams=self.streams,
                                         lr_t_dev=self.s_lr)
        self.s_lq.copy_(lq)
        self.s_truth.copy_(truth)
        self.s_off.copy_(off)
        self.s_lr.copy_(lr)
        self.graph.replay()
        G.t += 1
        D.t += 1
        return self.res[0].clone(), self.res[1].clone()
