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

    def tower(self, image, label, offsets, adapt=1.0, update_moving=True):
        """image: torch CUDA float32 [1,S,S,1]; label, adapt: host floats (the reference feeds them through placeholders,
        :1554-1556, :1729-1736).  Adds adapt * d(data loss)/d(parameters) into self.grads (the l2 term is added by
        ``step``).  -> device tensor [out, data loss]."""
        import torch

        assert image.shape[0] == 1 and image.shape[3] == 1, "one image per tower (batch_size = 1 in the reference)"
        dev = self.device

        def pad4(t):
            out = torch.zeros(t.shape[:3] + (4,), dtype=torch.float32, device=dev)
            out[..., 0:1].copy_(t)
            return out

        self._pad_dirty = True
        small, medium, large = multiscale_crops(image, offsets)
        S4 = small.shape[1]
        inputs = {"small": ops.Act(pad4(small)),
                  "medium": ops.avgpool2x2(ops.Act(pad4(medium)), self._E(1, S4, S4, 4)),
                  "large": ops.resize_bilinear(ops.Act(pad4(large)), self._E(1, S4, S4, 4))}
        saved, logits, means = {}, [], {}
        for br in BRANCHES:
            x, ctxs = inputs[br], []
            for lay in self.L[br]["layers"]:
                Ho, Wo = -(-x.H // 2), -(-x.W // 2)
                d = ops.dw3x3(x, self._dw(lay), self._E(1, Ho, Wo, x.C), stride=2)
                r = ops.conv1x1(d, lay["pk_f"], self.ones, self.zeros, self._E(1, Ho, Wo, lay["cout"]), act=False,
                                precision=self.precision)
                mean, var = ops.bn_batch_stats(r)
                b = lay["scope"] + "/BatchNorm"
                f = lay["cout"]
                mv = (self.m[b + "/moving_mean"], self.m[b + "/moving_variance"], self.scratch[2 * features5: 2 * features5 + f],
                      self.scratch[3 * features5: 3 * features5 + f]) if update_moving else None
                fold = TO.bn_train_fold(mean, var, self.ones[:f], self.zeros[:f], Ho * Wo, gamma1=self.v[b + "/gamma"],
                                        beta1=self.v[b + "/beta"], moving=mv, eps=BN_EPS_DISCR, decay=BN_DECAY_DISCR)
                y = ops.affine_act(r, fold["scale"], fold["shift"], self._E(1, Ho, Wo, f), act=ops.ACT_LEAKY)
                ctxs.append({"x": x, "d": d, "r": r, "fold": fold})
                x = y
            mean, _ = ops.bn_batch_stats(x)         # tf.reduce_mean(., [1,2]) (:578)
            fc = self.L[br]["fc"]
            logit = torch.empty(1, dtype=torch.float32, device=dev)
            _lib.check(_lib.load().emd_fc_rows_f32(ops._p(mean), features5, ops._p(self.v[fc + "/weights"]),
                                                   ops.C.c_float(0.0), ops._p(logit), 1, features5, _lib.stream_ptr()),
                       "emd_fc_rows_f32")
            logits.append(logit + self.v[fc + "/biases"])
            saved[br], means[br] = (ctxs, x), mean
        result, dlogit = TO.gan_head(torch.cat(logits), float(label), 0, grad_scale=float(adapt))
        # ---- reverse pass, branch by branch (only the arg-max branch has a non-zero dlogit)
        for k, br in enumerate(BRANCHES):
            ctxs, last = saved[br]
            fc = self.L[br]["fc"]
            dmean = TO.fc_row_bwd(means[br], self.v[fc + "/weights"].view(features5), dlogit[k:k + 1],
                                  self.g[fc + "/weights"].view(features5), self.g[fc + "/biases"])
            dy = TO.bcast_rows(dmean, self._E(1, last.H, last.W, features5), 1.0 / (last.H * last.W))
            for li in reversed(range(len(ctxs))):
                lay, c = self.L[br]["layers"][li], ctxs[li]
                b = lay["scope"] + "/BatchNorm"
                f = lay["cout"]
                dr = TO.bn_backward(dy, c["r"], c["fold"], self.ones[:f], self.scratch[:f], self.scratch[features5: features5 + f],
                                    c["r"], mask=TO.MASK_LEAKY, gamma1=self.v[b + "/gamma"], dgamma1=self.g[b + "/gamma"],
                                    eps=BN_EPS_DISCR)
                TO.conv_wgrad(c["d"], dr, self._gpw(lay))
                dd = ops.conv1x1(dr, lay["pk_b"], self.ones, self.zeros, c["d"], act=False, precision=self.precision)
                TO.dw3x3_wgrad(c["x"], dd, self._gdw(lay), stride=2)
                if li > 0:
                    dy = TO.dw3x3_bwd_data(dd, self._dw(lay), self._E(1, c["x"].H, c["x"].W, c["x"].C), stride=2)
        return result

    # ---- one optimizer step
    def step(self, images, labels, offsets, adapts=None, learning_rate=None, group=None):
        """_discriminator_train_op (:1390-1440) on this rank's towers: images torch [T,S,S,1] (generated and real ones
        with their labels, :1720-1775); the towers' gradients are averaged, clipped to global norm 15 and applied by
        Adam(beta1 = 0.5).  offsets: one ((y,x),(y,x),(y,x)) per tower.  -> device tensor [T, 2] of (out, data loss)."""
        import torch

        T = images.shape[0]
        adapts = [1.0] * T if adapts is None else list(adapts)
        self.zero_grad()
        res = [self.tower(images[k:k + 1].contiguous(), labels[k], offsets[k], adapt=adapts[k], update_moving=(k == 0))
               for k in range(T)]
        self._unpad_grads()
        # + adapt * 5e-5 * d(sum l2_loss)/dv = adapt * 5e-5 * v, summed over the towers
        n4 = self.params.numel() // 4
        TO.axpy(ops.Act(self.params.view(1, 1, n4, 4)), ops.Act(self.grads.view(1, 1, n4, 4)), alpha=L2_DISCR * float(sum(adapts)))
        world = sync_gradients(self.grads, self.moving, group)
        scale = 1.0 / (T * world)
        gn2 = TO.sumsq(self.grads, scale=scale)
        self.t += 1
        TO.adam_step(self.params, self.grads, self.adam_m, self.adam_v, self.t, self.lr if learning_rate is None else learning_rate,
                     beta1=ADAM_BETA1, grad_scale=scale, gnorm_sq=gn2, clip_norm=CLIP_DISCR)
        self.repack()
        return torch.stack(res)
