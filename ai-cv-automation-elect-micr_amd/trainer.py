"""Graph D' training on MI355X: the data-parallel trainer of misc_py/denoiser-multi-gpu.py.

Mirrors, for one process per GPU:
  * ``_tower_fn`` (:752-782)  -> ``DenoiserTrainer.tower(lq, truth)``: architecture(phase=True) -- every batch norm
    on BATCH statistics -- the capped-MSE loss, and tf.gradients as a hand-written reverse pass over saved
    activations (no autograd engine: each layer's backward is a fixed sequence of libemdenoise.so launches);
  * ``_train_op`` (:1011-1077) + the loop (:1169-1206) -> ``train_step``: gradient sets from all towers / micro
    batches are summed in ONE flat fp32 buffer (parameter gradients accumulate in place), all-reduced over RCCL,
    averaged (`add_n * 1/len`, :1040) and applied by tf.train.MomentumOptimizer(use_nesterov=True) (:1064-1066);
    batch-norm moving statistics are updated by the first tower only (:701-707).
The reference moves 10 gradient sets of 154 MB through host memory per step (:1177-1196); here parameters,
gradients and momentum are three flat device vectors and the only exchange is one all-reduce of the gradient vector.

A tower normalises over the images it is given: ``tower_batch=1`` is the reference's behaviour (one image per
tower, :763), larger tower batches are ordinary batch-statistics training.
Python here is plumbing (buffers, views, launch order); there is no CPU compute path.
"""
from __future__ import annotations

from collections import OrderedDict

import os
import types

import numpy as np

from . import _lib, ops
from . import train_ops as TO
from .denoiser import (aspp_filters, aspp_output, aspp_rateLarge, aspp_rateMedium, aspp_rateSmall, declare_layers,
                       features0, features1, features2, features3, features4, num_extra_blocks, variable_specs)

INITIAL_LEARNING_RATE = 0.001   # denoiser-multi-gpu.py:118
MOMENTUM = 0.9                  # :1065


def _is_moving(name):
    return name.endswith(("/moving_mean", "/moving_variance"))


def sync_gradients(grads, moving, group=None):
    """The step's only exchange (replaces the host-side gradient shuffle of denoiser-multi-gpu.py:1177-1196): SUM the
    flat gradient vector over all ranks (RCCL all-reduce over xGMI; gloo in the CPU tests) and let the moving
    statistics follow rank 0's first tower (:701-707).  Returns the world size (1 if torch.distributed is not
    initialised); the caller divides by the total number of gradient sets (:1040)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return 1
    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(grads, group=group)
        dist.broadcast(moving, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return world


class DenoiserTrainer:
    """Parameters, gradients and momentum of graph D' resident on one GPU + the forward/backward launch sequence."""

    def __init__(self, weights, device, precision="bf16x3", learning_rate=INITIAL_LEARNING_RATE, momentum=MOMENTUM):
        import torch

        _lib.load()
        self.device = device
        self.precision = {"bf16x3": ops.PREC_BF16X3, "bf16": ops.PREC_BF16}[precision]
        self.lr, self.momentum = learning_rate, momentum
        self.layers = declare_layers("Dprime")
        specs = variable_specs("Dprime")
        self.trainable = OrderedDict((n, s) for n, s in specs.items() if not _is_moving(n))
        self.moving_names = OrderedDict((n, s) for n, s in specs.items() if _is_moving(n))

        def flat(names, fill=None):
            # every view starts on a 16-byte boundary (the kernels read weights as float4)
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
        self.accum, _ = flat(self.trainable)
        self.moving, self.m = flat(self.moving_names, weights)
        cmax = 5 * aspp_filters
        self.ones = torch.ones(cmax, dtype=torch.float32, device=device)
        self.zeros = torch.zeros(cmax, dtype=torch.float32, device=device)
        # the 1-channel image is processed as 4 channels (3 of zeros): padded copies of the three Cin = 1 weights
        # and of their gradients (row / channel 0 is the real one)
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=device)
        self.pad_w = {"cnn0_dw": z(9, 4), "cnn0_pw": z(1, 4, features0), "residual0": z(1, 4, features1)}
        self.pad_g = {k: torch.zeros_like(t) for k, t in self.pad_w.items()}
        # packed bf16 weights, forward and transposed (data gradient) orientation
        self.pk_f, self.pk_b = {}, {}
        for key, L in self.layers.items():
            cin = max(L.cin, 4)
            if L.kind == "sep":
                self.pk_f[key] = TO.DevPackedWeights(1, cin, L.cout, device)
                if L.cin > 1:
                    self.pk_b[key] = TO.DevPackedWeights(1, L.cout, cin, device)
            elif L.kind == "conv" and L.cout > 1:
                t = L.k * L.k
                self.pk_f[key] = TO.DevPackedWeights(t, cin, L.cout, device)
                if L.cin > 1:
                    self.pk_b[key] = TO.DevPackedWeights(t, L.cout, cin, device)
            elif L.kind == "deconv":
                self.pk_f[key] = [TO.DevPackedWeights(len(ops.deconv_phase_taps(ph)), L.cin, L.cout, device) for ph in range(4)]
                self.pk_b[key] = TO.DevPackedWeights(9, L.cout, L.cin, device)
        self.pk_b["cnn0"] = TO.DevPackedWeights(1, features0, 4, device)   # d loss / d (depthwise output), 4 padded channels
        self.dw_flip = {}
        self._streams, self._graphs = [], {}
        self._wg_side, self._wg_keep = None, []   # side stream of the weight-gradient launches of a batched pass (see _wg)
        self._pack_batch = None
        self._flip_idx = self._flip_flat = None
        self._per_image = False
        # the norms of small per-image maps as one launch per direction (train_ops.bn_train_fwd_small / bn_backward_small): opt-in -- measured
        # 47.3-47.7 ms per step against 46.2 with the four-launch forms (profiles/r04_experiments.txt 6)
        self.fuse_bn_small = os.environ.get("EMD_T_BN_SMALL", "0") == "1"
        # affine + relu6 of a separable conv that feeds only the next one applied in that one's loads (ops.PreAct; round 4)
        self.lazy_affine = os.environ.get("EMD_T_LAZY_AFFINE", "1") == "1"
        self.lazy_res = os.environ.get("EMD_T_LAZY_RES", "1") == "1"   # ... and of a 1x1 residual projection in the pass that adds it (with lazy_affine)
        # ... and the gradient of such a never-written activation is never written either: the producer's BN backward forms it from the
        # consumer's depthwise data gradient on the fly (TO.DwGrad / bn_backward_dw; round 4)
        self.fuse_dw_bn_bwd = os.environ.get("EMD_T_DW_BN_BWD", "1") == "1"
        # ... also behind a stride-2 (or dilated) consumer (gather form): correct and tested, but measured SLOWER (44.76 / 44.83 ms per step
        # against 44.43 / 44.41: the gather's per-pixel index arithmetic costs more than the two passes it saves on four layers); opt-in
        self.fuse_dw_bn_bwd_s2 = os.environ.get("EMD_T_DW_BN_BWD_S2", "0") == "1"
        self.fuse_dw_both = os.environ.get("EMD_T_DW_BOTH", "1") == "1"     # a written input: the depthwise stage's two gradients in one pass over dd
        self.fuse_dw_wgrad = os.environ.get("EMD_T_DW_WGRAD", "1") == "1"   # ... whose reduction pass also adds the consumer's depthwise weight gradient
        # the final conv's data gradient formed on the fly in its two consumers' BN backward (TO.Cout1Grad): correct, tested, and measured
        # SLOWER (44.4-44.5 ms per step against 43.0-43.2: nine conditional scalar loads per pixel, four times, cost more than the five
        # passes over a 64-channel 512^2 tensor they replace); opt-in
        self.fuse_cout1_grad = os.environ.get("EMD_T_COUT1_GRAD", "0") == "1"
        self.fuse_fold = os.environ.get("EMD_T_FUSE_FOLD", "1") == "1"    # the BN fold inside the statistics' final kernel (with fuse_stats)
        self.fuse_stats = os.environ.get("EMD_T_FUSE_STATS", "1") == "1"   # batch statistics from the producing GEMM's epilogue (ops.conv_stats)
        self.teacher = None   # test hook: scope -> {"d": ..., "r": ...} reference tensors that REPLACE the forward's conv outputs (see _force)
        self.repack()
        self.last = None

    # ---- parameters ------------------------------------------------------------------------------------
    def _w(self, key):
        """fp32 weight tensor of a layer as [taps][Cin][Cout] (conv / pointwise) or [9][Cout][Cin] (transposed conv)."""
        L = self.layers[key]
        if L.kind == "sep":
            return self.pad_w["cnn0_pw"] if L.cin == 1 else self.v[L.scope + "/pointwise_weights"].view(1, L.cin, L.cout)
        if L.kind == "conv":
            return self.pad_w[key] if L.cin == 1 else self.v[L.scope + "/" + L.wname].view(L.k * L.k, L.cin, L.cout)
        return self.v[L.scope + "/" + L.wname].view(9, L.cout, L.cin)

    def _dw(self, key):
        L = self.layers[key]
        return self.pad_w["cnn0_dw"] if L.cin == 1 else self.v[L.scope + "/depthwise_weights"].view(9, L.cin)

    def repack(self):
        """fp32 parameters -> padded Cin = 1 copies, bf16 hi/lo planes (both orientations), flipped depthwise taps.
        Runs after every optimizer step; everything stays on the device."""
        L0, R0 = self.layers["cnn0"], self.layers["residual0"]
        self.pad_w["cnn0_dw"][:, 0].copy_(self.v[L0.scope + "/depthwise_weights"].view(9))
        self.pad_w["cnn0_pw"][0, 0].copy_(self.v[L0.scope + "/pointwise_weights"].view(L0.cout))
        self.pad_w["residual0"][0, 0].copy_(self.v[R0.scope + "/" + R0.wname].view(R0.cout))
        if self._pack_batch is None:
            # every pack of the model as one launch (emd_pack_weights_batch_dev): the job table is built once, the parameter
            # views and the packed planes it points at never move
            pb = TO.PackBatch(self.device)
            for key, L in self.layers.items():
                if L.kind == "sep" or (L.kind == "conv" and L.cout > 1):
                    w = self._w(key)
                    taps = w.shape[0]
                    pb.add(self.pk_f[key], w, taps, cout_major=False)
                    if key in self.pk_b:  # K = Cout, N = Cin: the same array read "cout_major"; taps reversed
                        pb.add(self.pk_b[key], w, taps, cout_major=True, tap_sel=list(range(taps))[::-1])
                elif L.kind == "deconv":
                    w = self._w(key)
                    for ph in range(4):
                        sel = [ky * 3 + kx for (ky, kx) in ops.deconv_phase_taps(ph)]
                        pb.add(self.pk_f[key][ph], w, 9, cout_major=True, tap_sel=sel)
                    pb.add(self.pk_b[key], w, 9, cout_major=False)
            self._pack_batch = pb
        self._pack_batch.run()
        # taps reversed for the depthwise data gradients (stride 1: the forward kernel on them; every stride: TO.DwGrad): one gather from the flat parameter vector into one buffer that the per-layer
        # [9][C] views point into; updated IN PLACE (a captured hipGraph keeps the pointers)
        import torch

        if self._flip_idx is None:
            keys = [k for k, L in self.layers.items() if L.kind == "sep" and L.cin > 1]
            idx, off = [], 0
            for k in keys:
                w = self._dw(k)
                base = (w.data_ptr() - self.params.data_ptr()) // 4
                assert w.is_contiguous() and 0 <= base and base + w.numel() <= self.params.numel()
                c = w.shape[1]
                idx.append(torch.arange(base, base + 9 * c, dtype=torch.int64).view(9, c).flip(0).reshape(-1))
                off += 9 * c
            self._flip_idx = torch.cat(idx).to(self.device)
            self._flip_flat = torch.empty(off, dtype=torch.float32, device=self.device)
            off = 0
            for k in keys:
                c = self._dw(k).shape[1]
                self.dw_flip[k] = self._flip_flat[off:off + 9 * c].view(9, c)
                off += 9 * c
        torch.index_select(self.params, 0, self._flip_idx, out=self._flip_flat)

    def state_dict(self):
        """TF variable name -> numpy array (parameters and moving statistics)."""
        out = OrderedDict()
        for name in variable_specs("Dprime"):
            out[name] = (self.m if _is_moving(name) else self.v)[name].detach().cpu().numpy().copy()
        return out

    def save_checkpoint(self, directory, global_step=0, name="model"):
        """Write parameters, moving statistics, the optimizer's ``<variable>/Momentum`` slots and ``global_step`` as a
        TensorFlow checkpoint bundle (what the reference's ``saver.save(sess, save_path=model_dir+"model/", ...)`` leaves,
        misc_py/denoiser-multi-gpu.py:1226), readable by tf.train.Saver and by emdenoise.Denoiser(checkpoint_loc=dir)."""
        import os

        from . import tf_checkpoint as ckpt

        tensors = dict(self.state_dict())
        acc = self.accum.detach().cpu().numpy()
        off = 0
        for n, shape in self.trainable.items():
            size = int(np.prod(shape))
            tensors[n + "/Momentum"] = acc[off: off + size].reshape(shape).copy()
            off += -(-size // 4) * 4
        tensors["global_step"] = np.array(int(global_step), np.int64)
        prefix = os.path.join(directory, f"{name}-{int(global_step)}")
        ckpt.write_checkpoint(prefix, tensors)
        return prefix

    def gradients(self):
        """TF variable name -> numpy gradient accumulated since zero_grad() (sum over towers)."""
        self._unpad_grads()
        return OrderedDict((n, t.detach().cpu().numpy().copy()) for n, t in self.g.items())

    def zero_grad(self):
        self.grads.zero_()
        for t in self.pad_g.values():
            t.zero_()

    def _unpad_grads(self):
        L0, R0 = self.layers["cnn0"], self.layers["residual0"]
        self.g[L0.scope + "/depthwise_weights"].view(9).copy_(self.pad_g["cnn0_dw"][:, 0])
        self.g[L0.scope + "/pointwise_weights"].view(L0.cout).copy_(self.pad_g["cnn0_pw"][0, 0])
        self.g[R0.scope + "/" + R0.wname].view(R0.cout).copy_(self.pad_g["residual0"][0, 0])

    def _gw(self, key):
        L = self.layers[key]
        if L.kind == "sep":
            return self.pad_g["cnn0_pw"] if L.cin == 1 else self.g[L.scope + "/pointwise_weights"].view(1, L.cin, L.cout)
        if L.kind == "conv":
            return self.pad_g[key] if L.cin == 1 else self.g[L.scope + "/" + L.wname].view(L.k * L.k, L.cin, L.cout)
        return self.g[L.scope + "/" + L.wname].view(9, L.cout, L.cin)

    def _gdw(self, key):
        L = self.layers[key]
        return self.pad_g["cnn0_dw"] if L.cin == 1 else self.g[L.scope + "/depthwise_weights"].view(9, L.cin)

    # ---- forward building blocks (training mode); each returns (y, ctx) -----------------------------------
    def _E(self, B, H, W, Cc):
        return ops.Act.empty(B, H, W, Cc, self.device)

    def _bn(self, key, r, bias_name=None, stats=None):
        """Batch statistics of r -> fold dict of the layer's BN chain (+ moving-average updates on the first tower).  stats: (mean,
        var) when the producing convolution has delivered them already (ops.conv_stats: the GEMM's epilogue)."""
        L = self.layers[key]
        img = r.B if self._per_image else 0    # per-image statistics: B one-image towers as one batched pass
        if stats is not None and len(stats) == 3:   # (mean, var, fold): the conv's final statistics kernel ran the fold too (_fold_req)
            return stats[2]
        if stats is not None:
            mean, var = stats
            npix = r.H * r.W if img else r.B * r.H * r.W
        elif img:
            mean, var = ops.bn_batch_stats_images(r)
            npix = r.H * r.W
        else:
            mean, var = ops.bn_batch_stats(r)
            npix = r.B * r.H * r.W
        upd = self._update_moving
        if len(L.bn) == 2:
            b1, b2 = L.bn
            mv = (self.m[b1 + "/moving_mean"], self.m[b1 + "/moving_variance"], self.m[b2 + "/moving_mean"],
                  self.m[b2 + "/moving_variance"]) if upd else None
            return TO.bn_train_fold(mean, var, self.v[b2 + "/gamma"], self.v[b2 + "/beta"], npix, gamma1=self.v[b1 + "/gamma"],
                                    beta1=self.v[b1 + "/beta"], moving=mv, images=img)
        (b2,) = L.bn
        mv = (self.m[b2 + "/moving_mean"], self.m[b2 + "/moving_variance"]) if upd else None
        return TO.bn_train_fold(mean, var, self.v[b2 + "/gamma"], self.v[b2 + "/beta"], npix,
                                bias=self.v[bias_name] if bias_name else None, moving=mv, images=img)

    def _fold_req(self, key, B, bias_name=None):
        """The fold of layer `key`'s BN chain as a request to the convolution that delivers its statistics (TO.FoldRequest; round 4: the
        per-channel step runs in the statistics' final kernel -- one launch less per layer), or None (EMD_T_FUSE_FOLD=0)."""
        if not self.fuse_fold:
            return None
        L = self.layers[key]
        upd = self._update_moving
        img = B if self._per_image else 0
        if len(L.bn) == 2:
            b1, b2 = L.bn
            mv = (self.m[b1 + "/moving_mean"], self.m[b1 + "/moving_variance"], self.m[b2 + "/moving_mean"],
                  self.m[b2 + "/moving_variance"]) if upd else None
            return TO.FoldRequest(self.device, img, L.cout, self.v[b2 + "/gamma"], self.v[b2 + "/beta"], gamma1=self.v[b1 + "/gamma"],
                                  beta1=self.v[b1 + "/beta"], moving=mv)
        (b2,) = L.bn
        mv = (self.m[b2 + "/moving_mean"], self.m[b2 + "/moving_variance"]) if upd else None
        return TO.FoldRequest(self.device, img, L.cout, self.v[b2 + "/gamma"], self.v[b2 + "/beta"],
                              bias=self.v[bias_name] if bias_name else None, moving=mv)

    def _bn_small_shape(self, npix_img, cout, B):
        """The one-launch forms of the norm (train_ops.bn_train_fwd_small / bn_backward_small) for the small per-image maps (32 x 32,
        64 x 64) of a batched pass of one-image towers -- and of a tower of ONE image, whose batch statistics are per-image statistics:
        a tower alone and the same image inside a batched pass then run the same arithmetic (tests/test_train_gpu.py holds them to
        equality).  Not under teacher forcing (its hooks sit between the launches)."""
        return (self.fuse_bn_small and (self._per_image or B == 1) and self.teacher is None and npix_img <= 4096 and cout % 4 == 0)

    def _bn_small(self, r):
        return self._bn_small_shape(r.H * r.W, r.C, r.B) and TO.bn_small_supported(r)

    def _bn_apply(self, key, r, out, act, res=None, bias_name=None, stats=None):
        """Batch norm chain of layer `key` on the conv output r + activation (+ residual) -> (out, fold): one launch for small per-image
        maps, else statistics (unless the conv delivered them) + fold + affine."""
        if self._bn_small(r):
            L = self.layers[key]
            upd = self._update_moving
            if len(L.bn) == 2:
                b1, b2 = L.bn
                mv = (self.m[b1 + "/moving_mean"], self.m[b1 + "/moving_variance"], self.m[b2 + "/moving_mean"],
                      self.m[b2 + "/moving_variance"]) if upd else None
                fold = TO.bn_train_fwd_small(r, self.v[b2 + "/gamma"], self.v[b2 + "/beta"], out, act, gamma1=self.v[b1 + "/gamma"],
                                             beta1=self.v[b1 + "/beta"], moving=mv, res=res)
            else:
                (b2,) = L.bn
                mv = (self.m[b2 + "/moving_mean"], self.m[b2 + "/moving_variance"]) if upd else None
                fold = TO.bn_train_fwd_small(r, self.v[b2 + "/gamma"], self.v[b2 + "/beta"], out, act,
                                             bias=self.v[bias_name] if bias_name else None, moving=mv, res=res)
            return out, fold
        fold = self._bn(key, r, bias_name, stats=stats)
        self._affine(r, fold, out, act, res)
        return out, fold

    @staticmethod
    def _affine(r, fold, out, act, res=None):
        """out = act(r * scale + shift) [+ res] with the fold of _bn: per-channel, or per (image, channel) for per-image statistics."""
        if isinstance(res, ops.PreAct):   # the residual projection's own norm + relu6 applied here (its output was never written)
            assert bool(fold.get("B")) == res.images
            return ops.affine_act_res_pre(r, fold["scale"], fold["shift"], out, res, act=act)
        if fold.get("B"):
            return ops.affine_act_images(r, fold["scale"], fold["shift"], out, act=act, res=res)
        return ops.affine_act(r, fold["scale"], fold["shift"], out, act=act, res=res)

    def _force(self, act, scope, which):
        """Teacher forcing of the forward pass for the mask-forced gradient experiment (tests/test_train_gpu.py): the tensor just
        computed is replaced by the reference's, so that everything downstream -- batch statistics, activations, and the relu6 /
        clip MASKS the backward pass derives from them -- is the reference's to float32 rounding."""
        if self.teacher is None or scope not in self.teacher or which not in self.teacher[scope]:
            return
        import torch

        v = torch.from_numpy(np.ascontiguousarray(self.teacher[scope][which], dtype=np.float32)).to(self.device)
        act.torch()[..., : v.shape[-1]].copy_(v)

    def _fuse_stats(self, x, stride=1):
        """Statistics in the producing GEMM's epilogue (ops.conv_stats) -- not under teacher forcing (the statistics must then be those of
        the FORCED tensor) and only where per-image statistics see whole tiles."""
        return self.fuse_stats and self.teacher is None and ops.conv_stats_supported(x, stride, images=self._per_image)

    def _sep_fwd(self, key, x, out=None, res=None, lazy=False):
        """lazy=True (the caller's promise: the output feeds ONE separable conv and nothing else): the affine + relu6 is left to that
        consumer's loads -- the result is an ops.PreAct (r, scale, shift), never written (round 4: two passes over the tensor and
        one launch less; same bits).  x may be such a PreAct."""
        L = self.layers[key]
        Ho, Wo = -(-x.H // L.stride), -(-x.W // L.stride)
        if isinstance(x, ops.PreAct):
            d = ops.dw3x3_pre_act(x, self._dw(key), self._E(x.B, Ho, Wo, x.C), stride=L.stride, rate=L.rate)
        else:
            d = ops.dw3x3(x, self._dw(key), self._E(x.B, Ho, Wo, x.C), stride=L.stride, rate=L.rate)
        self._force(d, L.scope, "d")
        stats = None
        small = self._bn_small_shape(Ho * Wo, L.cout, x.B)   # the one-launch norm takes its own statistics
        if not small and self._fuse_stats(d):     # the batch statistics of r from the pointwise GEMM's epilogue: no second pass over r
            r = self._E(x.B, Ho, Wo, L.cout)
            fq = self._fold_req(key, x.B)
            stats = ops.conv_stats(d, self.pk_f[key], self.ones, self.zeros, r, images=self._per_image, precision=self.precision, fold=fq)
            if fq is not None:
                stats = (stats[0], stats[1], fq.result(stats[0]))
        else:
            r = ops.conv1x1(d, self.pk_f[key], self.ones, self.zeros, self._E(x.B, Ho, Wo, L.cout), act=False,
                            precision=self.precision)
        self._force(r, L.scope, "r")
        # (not with the weight gradients on a side stream: the consumer's depthwise weight gradient re-reads r, which this layer's own
        # BN backward overwrites on the main stream; not under teacher forcing, whose hooks replace whole tensors)
        if lazy and self.lazy_affine and not small and out is None and res is None and self._wg_side is None and self.teacher is None:
            fold = self._bn(key, r, stats=stats)
            return ops.PreAct(r, fold["scale"], fold["shift"], images=bool(fold.get("B")), act=ops.ACT_RELU6), {"x": x, "d": d, "r": r, "fold": fold}
        if out is None:
            out = self._E(x.B, Ho, Wo, L.cout)
        out, fold = self._bn_apply(key, r, out, ops.ACT_RELU6, res, stats=stats)
        return out, {"x": x, "d": d, "r": r, "fold": fold}

    def _conv_fwd(self, key, x, out=None, act=True, lazy=False):
        """conv + bias -> BN -> relu6 (bias folds into the batch mean), or conv + bias alone (the image-level conv).
        lazy=True (the caller's promise: the output is ONLY the residual operand of one separable conv's affine): the norm + relu6 is
        left to that affine pass (ops.PreAct as ``res``, emd_affine_act_res_affine_f32): one read and one write of the tensor less."""
        L = self.layers[key]
        Ho, Wo = -(-x.H // L.stride), -(-x.W // L.stride)
        bias = self.v[L.scope + "/" + L.bname]
        has_bn = bool(L.bn)
        tgt = self._E(x.B, Ho, Wo, L.cout) if (has_bn or out is None) else out
        shift = self.zeros if has_bn else bias
        stats = None
        small = has_bn and self._bn_small_shape(Ho * Wo, L.cout, x.B)
        if has_bn and not small and (L.k == 1 or L.stride == 1) and self._fuse_stats(x, L.stride):
            fq = self._fold_req(key, x.B, L.scope + "/" + L.bname)
            stats = ops.conv_stats(x, self.pk_f[key], self.ones, self.zeros, tgt, stride=L.stride, rate=L.rate, images=self._per_image,
                                   precision=self.precision, fold=fq)
            if fq is not None:
                stats = (stats[0], stats[1], fq.result(stats[0]))
        elif L.k == 1:
            ops.conv1x1(x, self.pk_f[key], self.ones, shift, tgt, stride=L.stride, act=False, precision=self.precision)
        else:
            ops.conv3x3(x, self.pk_f[key], self.ones, shift, tgt, stride=L.stride, rate=L.rate, act=False,
                        precision=self.precision)
        self._force(tgt, L.scope, "r")
        if not has_bn:
            return tgt, {"x": x}
        if lazy and self.lazy_affine and self.lazy_res and act and not small and not self.fuse_bn_small and out is None and self.teacher is None:
            fold = self._bn(key, tgt, L.scope + "/" + L.bname, stats=stats)
            return (ops.PreAct(tgt, fold["scale"], fold["shift"], images=bool(fold.get("B")), act=ops.ACT_RELU6), {"x": x, "r": tgt, "fold": fold})
        if out is None:
            out = self._E(x.B, Ho, Wo, L.cout)
        out, fold = self._bn_apply(key, tgt, out, ops.ACT_RELU6 if act else ops.ACT_NONE, None, L.scope + "/" + L.bname, stats=stats)
        return out, {"x": x, "r": tgt, "fold": fold}

    def _deconv_fwd(self, key, x, out):
        L = self.layers[key]
        stats = None
        if self._fuse_stats(x):    # the batch statistics of r from the four phase GEMMs' epilogues (per image: whole 128-row tiles of the INPUT grid)
            r = self._E(x.B, 2 * x.H, 2 * x.W, L.cout)
            fq = self._fold_req(key, x.B, L.scope + "/" + L.bname)
            stats = ops.deconv_stats(x, self.pk_f[key], self.ones, self.zeros, r, images=self._per_image, precision=self.precision, fold=fq)
            if fq is not None:
                stats = (stats[0], stats[1], fq.result(stats[0]))
        else:
            r = ops.deconv3x3s2(x, self.pk_f[key], self.ones, self.zeros, self._E(x.B, 2 * x.H, 2 * x.W, L.cout), act=False,
                                precision=self.precision)
        self._force(r, L.scope, "r")
        fold = self._bn(key, r, L.scope + "/" + L.bname, stats=stats)
        self._affine(r, fold, out, ops.ACT_RELU6)
        return out, {"x": x, "r": r, "fold": fold}

    # ---- backward building blocks ---------------------------------------------------------------------------
    def _put(self, gslot, x, write, gemm_accumulate=None):
        """Deliver a gradient for tensor x.  gslot: dict id -> Act of gradients that already exist.  write(dst):
        overwrite dst with the gradient; gemm_accumulate(dst): add it into dst (if the producer can)."""
        k = id(x.buf), x.c0, x.C
        if k not in gslot:
            parent = self._gparent.get((id(x.buf)))
            dst = parent.slice(x.c0, x.C) if parent is not None else self._E(x.B, x.H, x.W, x.C)
            write(dst)
            gslot[k] = dst
        elif gemm_accumulate is not None:
            gemm_accumulate(gslot[k])
        else:
            tmp = self._E(x.B, x.H, x.W, x.C)
            write(tmp)
            TO.axpy(tmp, gslot[k])
        return gslot[k]

    def _wg(self, fn, *keep):
        """A weight-gradient launch.  Nothing downstream of it runs before the optimizer step (it only adds into its own slice of
        the gradient vector), so a batched pass CAN issue them on a side stream, behind everything the main stream has issued so
        far (opt-in, EMD_T_WGRAD_STREAM=1: measured 52.4-52.5 ms against 51.9-52.1 ms inline -- the chip is not idle enough beside
        the data-gradient chain for the overlap to pay).  ``keep``: tensors the launch reads that nothing else references (kept until the join in tower)."""
        side = self._wg_side
        if side is None:
            fn()
            return
        import torch

        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        self._wg_keep.extend(keep)

    def _bn_bwd(self, key, dy, ctx, mask=TO.MASK_RELU6):
        """dy -> d loss / d r, written over r (no longer needed); BN parameter gradients accumulate."""
        L = self.layers[key]
        r = ctx["r"]
        if isinstance(dy, TO.DwGrad):     # the gradient exists only as the consumer's depthwise data gradient (see _sep_bwd): formed on the fly
            bwd = TO.bn_backward_dw
        else:
            bwd = TO.bn_backward_small if (ctx["fold"].get("small") and self._bn_small(r)) else TO.bn_backward
        if len(L.bn) == 2:
            b1, b2 = L.bn
            return bwd(dy, r, ctx["fold"], self.v[b2 + "/gamma"], self.g[b2 + "/gamma"], self.g[b2 + "/beta"], r,
                       mask=mask, gamma1=self.v[b1 + "/gamma"], dgamma1=self.g[b1 + "/gamma"])
        (b2,) = L.bn
        return bwd(dy, r, ctx["fold"], self.v[b2 + "/gamma"], self.g[b2 + "/gamma"], self.g[b2 + "/beta"], r, mask=mask)

    def _sep_bwd(self, key, dy, ctx, gslot, need_dx=True):
        L = self.layers[key]
        x, d = ctx["x"], ctx["d"]
        dr = self._bn_bwd(key, dy, ctx)
        self._wg(lambda: TO.conv_wgrad(d, dr, self._gw(key)))
        # the data gradient lands on d -- unless the weight gradient above may still be reading d on its side stream
        dd_buf = d if self._wg_side is None else self._E(d.B, d.H, d.W, d.C)
        dd = ops.conv1x1(dr, self.pk_b[key], self.ones, self.zeros, dd_buf, act=False, precision=self.precision)
        deferred = (isinstance(x, ops.PreAct) and need_dx and self.fuse_dw_bn_bwd and self._wg_side is None
                    and (self.fuse_dw_bn_bwd_s2 or (L.stride == 1 and L.rate == 1))
                    and self._gkey(x) not in gslot and id(x.buf) not in self._gparent and x.C % 4 == 0)
        if deferred:
            # x was never written and this layer is its only consumer: its gradient is this depthwise data gradient and nothing else,
            # so it is not written either -- the producer's BN backward recomputes it from dd in both of its passes (TO.bn_backward_dw),
            # and its reduction pass, which streams dd and the r behind x, adds THIS layer's depthwise weight gradient on the way
            g = TO.DwGrad(dd, self.dw_flip[key], self._gdw(key) if (self.fuse_dw_wgrad and x.act == ops.ACT_RELU6) else None,
                          stride=L.stride, rate=L.rate, hw=(x.H, x.W))
            gslot[self._gkey(x)] = g
            if g.gdw is None:
                TO.dw3x3_wgrad_pre(x, dd, self._gdw(key), stride=L.stride, rate=L.rate)
            return
        if (self.fuse_dw_both and not isinstance(x, ops.PreAct) and need_dx and L.stride == 1 and L.rate == 1 and self._wg_side is None
                and x.C % 4 == 0 and L.cin > 1):
            # a written input (a block's first separable conv): data gradient and weight gradient in one pass over dd
            self._put(gslot, x, lambda dst: TO.dw3x3_bwd_both(dd, self.dw_flip[key], x, dst, self._gdw(key)))
            return
        if isinstance(x, ops.PreAct):    # the input was never written: rebuilt from the previous layer's r in the loads
            self._wg(lambda: TO.dw3x3_wgrad_pre(x, dd, self._gdw(key), stride=L.stride, rate=L.rate), dd)
        else:
            self._wg(lambda: TO.dw3x3_wgrad(x, dd, self._gdw(key), stride=L.stride, rate=L.rate), dd)
        if not need_dx:
            return
        if L.stride == 1:
            # SAME padding of a stride-1 dilated 3x3 is symmetric (rate, rate): the data gradient is the forward kernel
            # with the taps reversed at the same dilation
            self._put(gslot, x, lambda dst: ops.dw3x3(dd, self.dw_flip[key], dst, rate=L.rate))
        else:
            self._put(gslot, x, lambda dst: TO.dw3x3_bwd_data(dd, self._dw(key), dst, stride=L.stride, rate=L.rate))

    def _conv_bwd(self, key, dy, ctx, gslot, need_dx=True):
        L = self.layers[key]
        x = ctx["x"]
        if L.bn:
            dr = self._bn_bwd(key, dy, ctx)     # the bias before a training-mode batch norm has zero gradient
        else:
            dr = dy
            TO.chan_reduce(dy, self.g[L.scope + "/" + L.bname], accumulate_s1=True)
        if L.k == 1:
            self._wg(lambda: TO.conv_wgrad(x, dr, self._gw(key), [0], [0], sa=L.stride))
        else:
            tdy, tdx = TO.conv_taps(x.H, x.W, L.stride, L.rate)
            self._wg(lambda: TO.conv_wgrad(x, dr, self._gw(key), tdy, tdx, sa=L.stride))
        if not need_dx:
            return
        pk = self.pk_b[key]
        P = self.precision
        if L.stride == 2:
            def first(dst):
                dst.torch().zero_()
                TO.conv1x1_s2_bwd_data(dr, pk, self.ones, self.zeros, dst, accumulate=False, precision=P)
            self._put(gslot, x, first, lambda dst: TO.conv1x1_s2_bwd_data(dr, pk, self.ones, self.zeros, dst, accumulate=True, precision=P))
        elif L.k == 1:
            self._put(gslot, x, lambda dst: ops.conv1x1(dr, pk, self.ones, self.zeros, dst, act=False, precision=P),
                      lambda dst: ops.conv1x1(dr, pk, self.ones, self.zeros, dst, act=False, res=dst, precision=P))
        else:
            self._put(gslot, x, lambda dst: ops.conv3x3(dr, pk, self.ones, self.zeros, dst, rate=L.rate, act=False, precision=P),
                      lambda dst: ops.conv3x3(dr, pk, self.ones, self.zeros, dst, rate=L.rate, act=False, res=dst, precision=P))

    def _deconv_bwd(self, key, dy, ctx, gslot):
        L = self.layers[key]
        x = ctx["x"]
        dr = self._bn_bwd(key, dy, ctx)
        tdy, tdx = TO.conv_taps(dr.H, dr.W, 2, 1)
        self._wg(lambda: TO.conv_wgrad(dr, x, self._gw(key), tdy, tdx, sa=2))
        pk = self.pk_b[key]
        self._put(gslot, x, lambda dst: ops.conv3x3(dr, pk, self.ones, self.zeros, dst, stride=2, act=False, precision=self.precision),
                  lambda dst: ops.conv3x3(dr, pk, self.ones, self.zeros, dst, stride=2, act=False, res=dst, precision=self.precision))

    # ---- one tower ---------------------------------------------------------------------------------------------
    def tower(self, lq, truth, update_moving=True, grad_scale=1.0, per_image=False, wgrad_stream=False):
        """Forward (phase=True) + loss + backward for the images of one tower; parameter gradients are ADDED into
        self.grads.  lq, truth: CUDA float32 [B,S,S,1] contiguous, S a multiple of 32.  Returns (out, result3) with
        result3 a device tensor (mse, loss, dloss/dout factor) -- no host synchronisation.
        per_image=True: the B images are B TOWERS OF ONE IMAGE (the reference's towers, :763) run as one batched pass: every batch
        norm takes per-image statistics (and its backward per-image reductions), every image has its own loss; convolutions,
        depthwise convs, resampling and every parameter gradient are per-pixel or sums over pixels, so they need no change.
        Same arithmetic per image as B separate towers (the moving statistics follow image 0), B times the GEMM M, B times
        fewer launches.  result3 is then [B, 3].
        wgrad_stream=True: the weight-gradient launches of the backward pass go to a side stream (see _wg); same arithmetic.

        The pass is six segments (_enc_fwd, _mid_fwd, _dec_fwd, _dec_bwd, _mid_bwd, _enc_bwd) so that the
        1/16-resolution part can be driven on its own (round 3 ran it for several groups as ONE pass: slower, removed in round 4); here
        they simply follow each other."""
        import torch

        self._update_moving = update_moving
        self._per_image = bool(per_image)
        self._wg_side = self._side_streams(1)[0] if wgrad_stream else None
        st = self._enc_fwd(lq)
        ms = self._mid_fwd(st.cnn3_strided)
        self._dec_fwd(st, ms.aspp, truth, grad_scale)
        daspp = self._E(ms.aspp.B, ms.aspp.H, ms.aspp.W, ms.aspp.C)
        self._dec_bwd(st, ms.aspp, daspp)
        dx = self._mid_bwd(ms, daspp)
        self._enc_bwd(st, dx)
        if self._wg_side is not None:   # join: the gradient vector is complete, the tensors the side stream read may go
            torch.cuda.current_stream().wait_stream(self._wg_side)
            self._wg_side, self._wg_keep = None, []
        self.last = {"out": st.out, "result": st.result}
        return st.out, st.result

    def forward_train(self, lq, truth=None, update_moving=False, per_image=False):
        """``architecture(inputs, ground_truth, phase=True)`` (misc_py/denoiser-multi-gpu.py:200-540): the training-mode forward
        graph alone -- batch statistics in every norm, relu6, the in-graph clip -- on [B,S,S,1]; no loss gradient, no parameter
        gradient, and (default) no moving-statistics update, which in the reference is an UPDATE_OP of the train op, not of the
        builder.  Returns the output tensor [B,S,S,1]."""
        import torch

        self._update_moving = update_moving
        self._per_image = bool(per_image)
        self._wg_side = None
        st = self._enc_fwd(lq)
        ms = self._mid_fwd(st.cnn3_strided)
        self._dec_fwd(st, ms.aspp, torch.zeros_like(lq) if truth is None else truth)
        return st.out

    @staticmethod
    def _gkey(a):
        return id(a.buf), a.c0, a.C

    def _enc_fwd(self, lq, mid_out=None):
        """Encoder down to the input of the 1/16-resolution flow (cnn3_strided; written into ``mid_out`` when given)."""
        import torch

        assert lq.is_cuda and lq.dtype == torch.float32 and lq.is_contiguous() and lq.dim() == 4 and lq.shape[3] == 1
        B, S = lq.shape[0], lq.shape[1]
        assert lq.shape[2] == S and S % 32 == 0, "square crops with side a multiple of 32"
        st = types.SimpleNamespace(B=B, S=S, C={})
        E = lambda H, Cc: self._E(B, H, H, Cc)
        S2, S4 = S // 2, S // 4
        f1, f2 = features1, features2
        C = st.C
        x4 = torch.zeros((B, S, S, 4), dtype=torch.float32, device=self.device)
        x4[..., 0:1].copy_(lq)
        x = ops.Act(x4)
        cnn0, C["cnn0"] = self._sep_fwd("cnn0", x, lazy=True)
        cnn0_last, C["cnn0_last"] = self._sep_fwd("cnn0_last", cnn0, lazy=True)
        residual0, C["residual0"] = self._conv_fwd("residual0", x, lazy=True)
        st.concat1 = E(S2, f2 + f1)
        st.cnn0_strided, C["cnn0_strided"] = self._sep_fwd("cnn0_strided", cnn0_last, out=st.concat1.slice(f2, f1), res=residual0)
        cnn1, C["cnn1"] = self._sep_fwd("cnn1", st.cnn0_strided, lazy=True)
        cnn1_last, C["cnn1_last"] = self._sep_fwd("cnn1_last", cnn1, lazy=True)
        residual1, C["residual1"] = self._conv_fwd("residual1", st.cnn0_strided, lazy=True)
        st.concat2 = E(S4, aspp_output + f1)
        st.cnn1_strided, C["cnn1_strided"] = self._sep_fwd("cnn1_strided", cnn1_last, out=st.concat2.slice(aspp_output, f1), res=residual1)
        cnn2, C["cnn2"] = self._sep_fwd("cnn2", st.cnn1_strided, lazy=True)
        cnn2_last, C["cnn2_last"] = self._sep_fwd("cnn2_last", cnn2, lazy=True)
        residual2, C["residual2"] = self._conv_fwd("residual2", st.cnn1_strided, lazy=True)
        st.cnn2_strided, C["cnn2_strided"] = self._sep_fwd("cnn2_strided", cnn2_last, res=residual2)
        cnn3, C["cnn3"] = self._sep_fwd("cnn3", st.cnn2_strided, lazy=True)
        cnn3_last, C["cnn3_last"] = self._sep_fwd("cnn3_last", cnn3, lazy=True)
        residual3, C["residual3"] = self._conv_fwd("residual3", st.cnn2_strided, lazy=True)
        st.cnn3_strided, C["cnn3_strided"] = self._sep_fwd("cnn3_strided", cnn3_last, out=mid_out, res=residual3)
        return st

    def _mid_fwd(self, x):
        """The 1/16-resolution part: encoder 4, the middle flow and ASPP -> the 256-channel ASPP output (x: [B,S/16,S/16,728])."""
        B, S16 = x.B, x.H
        ms = types.SimpleNamespace(B=B, C={}, x=x)
        E = lambda H, Cc: self._E(B, H, H, Cc)
        af = aspp_filters
        C = ms.C
        t, C["cnn4_a"] = self._sep_fwd("cnn4_a", x, lazy=True)
        t, C["cnn4_b"] = self._sep_fwd("cnn4_b", t, lazy=True)
        cur, C["cnn4_last"] = self._sep_fwd("cnn4_last", t, res=x)
        for i in range(num_extra_blocks):
            t, C[f"middle{i}_0"] = self._sep_fwd(f"middle{i}_0", cur, lazy=True)
            t, C[f"middle{i}_1"] = self._sep_fwd(f"middle{i}_1", t, lazy=True)
            cur, C[f"middle{i}_2"] = self._sep_fwd(f"middle{i}_2", t, res=cur)
        ms.cur = cur
        ms.cat = cat = E(S16, 5 * af)
        _, C["aspp_conv1x1"] = self._conv_fwd("aspp_conv1x1", cur, out=cat.slice(0, af))
        _, C["aspp_small"] = self._conv_fwd("aspp_small", cur, out=cat.slice(af, af))
        _, C["aspp_medium"] = self._conv_fwd("aspp_medium", cur, out=cat.slice(2 * af, af))
        _, C["aspp_large"] = self._conv_fwd("aspp_large", cur, out=cat.slice(3 * af, af))
        ms.pooled = ops.avgpool2x2(cur, self._E(B, S16 // 2, S16 // 2, af))
        ms.img_lvl, C["aspp_image_conv"] = self._conv_fwd("aspp_image_conv", ms.pooled)
        ms.up = ops.resize_bilinear(ms.img_lvl, E(S16, af))
        ms.fold_p = self._bn("aspp_pooling_bn", ms.up)
        self._affine(ms.up, ms.fold_p, cat.slice(4 * af, af), ops.ACT_RELU6)
        C["aspp_pooling_bn"] = {"r": ms.up, "fold": ms.fold_p}
        ms.aspp, C["aspp_reduce"] = self._conv_fwd("aspp_reduce", cat)
        return ms

    def _dec_fwd(self, st, aspp, truth, grad_scale=1.0):
        """Decoder from the ASPP output (``aspp``: this tower's images) to the output, and the loss."""
        import torch

        B, S, C = st.B, st.S, st.C
        assert truth.shape == (B, S, S, 1) and truth.is_contiguous() and truth.dtype == torch.float32
        E = lambda H, Cc: self._E(B, H, H, Cc)
        f0, f1, f2 = features0, features1, features2
        ops.resize_bilinear(aspp, st.concat2.slice(0, aspp_output))
        t, C["deconv2_a"] = self._sep_fwd("deconv2_a", st.concat2, lazy=True)
        residual2_d, C["residual2_d"] = self._conv_fwd("residual2_d", st.concat2, lazy=True)
        st.deconv2, C["deconv2_b"] = self._sep_fwd("deconv2_b", t, res=residual2_d)
        _, C["deconv2to1"] = self._deconv_fwd("deconv2to1", st.deconv2, st.concat1.slice(0, f2))
        t, C["deconv1_a"] = self._sep_fwd("deconv1_a", st.concat1, lazy=True)
        residual1_d, C["residual1_d"] = self._conv_fwd("residual1_d", st.concat1, lazy=True)
        st.deconv1, C["deconv1_b"] = self._sep_fwd("deconv1_b", t, res=residual1_d)
        st.deconv1to0, C["deconv1to0"] = self._deconv_fwd("deconv1to0", st.deconv1, E(S, f1))
        t, C["deconv0_a"] = self._sep_fwd("deconv0_a", st.deconv1to0, lazy=True)
        residual0_d, C["residual0_d"] = self._conv_fwd("residual0_d", st.deconv1to0, lazy=True)
        st.deconv0, C["deconv0_b"] = self._sep_fwd("deconv0_b", t, res=residual0_d)
        # final 3x3 conv to one channel (+ bias) -> BN -> relu6 -> clip [0,1]  (:528-538)
        Lf = self.layers["deconv_final"]
        st.wf = self.v[Lf.scope + "/" + Lf.wname].view(9, f0)
        rf = torch.empty((B, S, S, 1), dtype=torch.float32, device=self.device)
        ops.conv3x3_cout1(st.deconv0, st.wf, 1.0, 0.0, rf, act=0)
        st.rfa = ops.Act(rf)
        self._force(st.rfa, Lf.scope, "r")
        st.fold_f = fold_f = self._bn("deconv_final", st.rfa, Lf.scope + "/" + Lf.bname)
        out = torch.empty_like(rf)
        # one channel: run the per-channel affine over a [.., 4] view with the scalar replicated
        if self._per_image:
            sc4 = fold_f["scale"].view(B, 1).expand(B, 4).contiguous().view(-1)
            sh4 = fold_f["shift"].view(B, 1).expand(B, 4).contiguous().view(-1)
            ops.affine_act_images(ops.Act(rf.view(B, S, S // 4, 4)), sc4, sh4, ops.Act(out.view(B, S, S // 4, 4)), act=ops.ACT_RELU6_CLIP01)
        else:
            sc4, sh4 = fold_f["scale"].expand(4).contiguous(), fold_f["shift"].expand(4).contiguous()
            ops.affine_act(ops.Act(rf.view(B, S, S // 4, 4)), sc4, sh4, ops.Act(out.view(B, S, S // 4, 4)), act=ops.ACT_RELU6_CLIP01)
        # ---------------- loss (:768-775)
        st.dout = dout = torch.empty_like(out)
        if self._per_image:   # every image is its own tower: its own mse, loss and dloss/dout (:763, :768-775)
            result = torch.stack([TO.denoise_loss(out[b:b + 1], truth[b:b + 1], dout[b:b + 1], grad_scale=grad_scale) for b in range(B)])
        else:
            result = TO.denoise_loss(out, truth, dout, grad_scale=grad_scale)
        st.out, st.result = out, result

    def _dec_bwd(self, st, aspp, daspp):
        """Backward through the decoder; the gradient w.r.t. the ASPP output lands in ``daspp`` (this tower's images)."""
        B, S, C = st.B, st.S, st.C
        E = lambda H, Cc: self._E(B, H, H, Cc)
        S2, S4 = S // 2, S // 4
        f0, f1, f2 = features0, features1, features2
        st.G = G = {}
        st.gconcat1, st.gconcat2 = E(S2, f2 + f1), E(S4, aspp_output + f1)
        self._gparent = {id(st.concat1.buf): st.gconcat1, id(st.concat2.buf): st.gconcat2, id(aspp.buf): daspp}
        grad = lambda a: G[self._gkey(a)]
        Lf = self.layers["deconv_final"]
        drf = TO.bn_backward(ops.Act(st.dout), st.rfa, st.fold_f, self.v[Lf.bn[0] + "/gamma"], self.g[Lf.bn[0] + "/gamma"],
                             self.g[Lf.bn[0] + "/beta"], st.rfa, mask=TO.MASK_RELU6_CLIP)
        deconv0 = st.deconv0
        self._wg(lambda: TO.conv3x3_cout1_wgrad(deconv0, drf.buf, self.g[Lf.scope + "/" + Lf.wname].view(9, f0)))
        if self.fuse_cout1_grad and self.teacher is None and self._wg_side is None and not self.fuse_bn_small:
            # the final conv's data gradient is never written: deconv0_b's and residual0_d's BN backward form it from drf (TO.Cout1Grad)
            G[self._gkey(deconv0)] = TO.Cout1Grad(drf.buf, st.wf)
        else:
            self._put(G, deconv0, lambda dst: TO.conv3x3_cout1_bwd_data(drf.buf, st.wf, dst))
        # decoder 0: deconv0 = sep_b(sep_a(deconv1to0)) + residual0_d(deconv1to0)
        g = grad(deconv0)
        self._sep_bwd("deconv0_b", g, C["deconv0_b"], G)
        self._sep_bwd("deconv0_a", grad(C["deconv0_b"]["x"]), C["deconv0_a"], G)
        self._conv_bwd("residual0_d", g, C["residual0_d"], G)   # after the separable branch: a GEMM can ADD its data gradient, a depthwise kernel cannot
        self._deconv_bwd("deconv1to0", grad(st.deconv1to0), C["deconv1to0"], G)
        # decoder 1
        g = grad(st.deconv1)
        self._sep_bwd("deconv1_b", g, C["deconv1_b"], G)
        self._sep_bwd("deconv1_a", grad(C["deconv1_b"]["x"]), C["deconv1_a"], G)
        self._conv_bwd("residual1_d", g, C["residual1_d"], G)   # after the separable branch: a GEMM can ADD its data gradient, a depthwise kernel cannot
        self._deconv_bwd("deconv2to1", st.gconcat1.slice(0, f2), C["deconv2to1"], G)
        # decoder 2
        g = grad(st.deconv2)
        self._sep_bwd("deconv2_b", g, C["deconv2_b"], G)
        self._sep_bwd("deconv2_a", grad(C["deconv2_b"]["x"]), C["deconv2_a"], G)
        self._conv_bwd("residual2_d", g, C["residual2_d"], G)   # after the separable branch: a GEMM can ADD its data gradient, a depthwise kernel cannot
        self._put(G, aspp, lambda dst: TO.resize_bilinear_bwd(st.gconcat2.slice(0, aspp_output), dst))
        self._gparent = {}

    def _mid_bwd(self, ms, daspp):
        """Backward through ASPP, the middle flow and encoder 4: ``daspp`` = gradient w.r.t. the ASPP output (all images of the
        pass) -> the gradient w.r.t. the part's input."""
        B, C = ms.B, ms.C
        S16 = ms.x.H
        af = aspp_filters
        G = {self._gkey(ms.aspp): daspp}
        gcat = self._E(B, S16, S16, 5 * af)
        self._gparent = {id(ms.cat.buf): gcat}
        grad = lambda a: G[self._gkey(a)]
        cur = ms.cur
        # ASPP
        self._conv_bwd("aspp_reduce", grad(ms.aspp), C["aspp_reduce"], G)          # writes gcat (all five slices)
        Lp = self.layers["aspp_pooling_bn"]
        dup = TO.bn_backward(gcat.slice(4 * af, af), ms.up, ms.fold_p, self.v[Lp.bn[0] + "/gamma"], self.g[Lp.bn[0] + "/gamma"],
                             self.g[Lp.bn[0] + "/beta"], ms.up, mask=TO.MASK_RELU6)
        self._put(G, ms.img_lvl, lambda dst: TO.resize_bilinear_bwd(dup, dst))
        self._conv_bwd("aspp_image_conv", grad(ms.img_lvl), C["aspp_image_conv"], G)
        self._put(G, cur, lambda dst: TO.avgpool2x2_bwd(grad(ms.pooled), dst))
        self._conv_bwd("aspp_large", gcat.slice(3 * af, af), C["aspp_large"], G)
        self._conv_bwd("aspp_medium", gcat.slice(2 * af, af), C["aspp_medium"], G)
        self._conv_bwd("aspp_small", gcat.slice(af, af), C["aspp_small"], G)
        self._conv_bwd("aspp_conv1x1", gcat.slice(0, af), C["aspp_conv1x1"], G)
        # middle flow and encoder 4: y = sep2(sep1(sep0(x))) + x.  The gradient of y becomes the gradient of x once
        # sep2's backward has read it (the identity branch), and sep0's data gradient is then added into it.
        def residual_block(keys, y):
            gy = grad(y)
            self._sep_bwd(keys[2], gy, C[keys[2]], G)
            self._sep_bwd(keys[1], grad(C[keys[2]]["x"]), C[keys[1]], G)
            xin = C[keys[0]]["x"]
            G[self._gkey(xin)] = gy
            self._sep_bwd(keys[0], grad(C[keys[1]]["x"]), C[keys[0]], G)

        y = cur
        for i in reversed(range(num_extra_blocks)):
            residual_block([f"middle{i}_{j}" for j in range(3)], y)
            y = C[f"middle{i}_0"]["x"]
        residual_block(["cnn4_a", "cnn4_b", "cnn4_last"], y)
        self._gparent = {}
        return grad(ms.x)

    def _enc_bwd(self, st, dx):
        """Backward through encoders 3..0: y = sep_strided(sep_last(sep(x))) + residual_conv(x); ``dx`` = gradient w.r.t.
        cnn3_strided (this tower's images)."""
        C, G = st.C, st.G
        f1 = features1
        self._gparent = {id(st.concat1.buf): st.gconcat1, id(st.concat2.buf): st.gconcat2}
        grad = lambda a: G[self._gkey(a)]

        def encoder(keys, res_key, y, need_dx=True):
            gy = grad(y)
            self._sep_bwd(keys[2], gy, C[keys[2]], G)
            self._sep_bwd(keys[1], grad(C[keys[2]]["x"]), C[keys[1]], G)
            self._sep_bwd(keys[0], grad(C[keys[1]]["x"]), C[keys[0]], G, need_dx=need_dx)
            self._conv_bwd(res_key, gy, C[res_key], G, need_dx=need_dx)   # last: its data gradient is added by the GEMM (no axpy pass)

        # cnn1_strided / cnn0_strided live in concat2 / concat1: their gradients started in the decoder (the slices of
        # gconcat2 / gconcat1 written above); the encoder-side consumers below add into them
        G[self._gkey(st.cnn3_strided)] = dx
        G[self._gkey(st.cnn1_strided)] = st.gconcat2.slice(aspp_output, f1)
        G[self._gkey(st.cnn0_strided)] = st.gconcat1.slice(features2, f1)
        encoder(["cnn3", "cnn3_last", "cnn3_strided"], "residual3", st.cnn3_strided)
        encoder(["cnn2", "cnn2_last", "cnn2_strided"], "residual2", st.cnn2_strided)
        encoder(["cnn1", "cnn1_last", "cnn1_strided"], "residual1", st.cnn1_strided)
        encoder(["cnn0", "cnn0_last", "cnn0_strided"], "residual0", st.cnn0_strided, need_dx=False)
        self._gparent = {}

    def _side_streams(self, n):
        import torch

        while len(self._streams) < n:
            self._streams.append(torch.cuda.Stream(device=self.device))
        return self._streams[:n]

    @staticmethod
    def batched_groups(B):
        """Number of concurrent batched passes a rank's B one-image towers are issued as (EMD_T_GROUPS overrides; 1 = one pass)."""
        env = os.environ.get("EMD_T_GROUPS")
        if env is not None:
            g = int(env)
            return g if g >= 1 and B % g == 0 else 1
        if B >= 8 and B % 4 == 0:
            return 4
        return 2 if B >= 4 and B % 2 == 0 else 1

    def local_gradients(self, lq, truth, tower_batch=1, streams=1, batched=False):
        """zero_grad + every tower of this rank's images (forward, loss, backward) -> device tensor [n_towers, 3] of
        (mse, loss, factor); the gradient sets are summed into self.grads.  streams > 1: the towers are independent
        (they only meet in the atomically accumulated parameter gradients), so they are issued round-robin on that
        many HIP streams and overlap on the chip -- a single 512x512 tower leaves most of the 256 CUs idle in its
        32x32 layers."""
        import torch

        B = lq.shape[0]
        assert B % tower_batch == 0
        n_local = B // tower_batch
        self.zero_grad()
        if batched and tower_batch == 1 and B > 1:
            # the B one-image towers as ONE batched pass with per-image batch-norm statistics (see tower): same arithmetic per image
            groups = self.batched_groups(B)
            # (the groups' 1/16-resolution parts merged into ONE pass of all B images was measured slower -- 51.0 ms against 47.5 ms for the
            # groups side by side all the way, profiles/r03_experiments.txt 15 -- and removed in round 4; the six-segment tower stays)
            if groups > 1:
                # ... as `groups` such passes on as many streams: at 8 images the 32 x 32 and 64 x 64 levels' kernels leave most CUs
                # idle, two-image passes side by side fill them (8 pairs of 512^2: 50.9 ms as one pass, 48.5 as two, 47.1-47.8 as
                # four, 55.8 as eight one-image towers)
                main = torch.cuda.current_stream()
                side = self._side_streams(groups)
                per = B // groups
                outs = []
                for s in side:
                    s.wait_stream(main)
                for k, s in enumerate(side):
                    with torch.cuda.stream(s):
                        _, r = self.tower(lq[k * per:(k + 1) * per].contiguous(), truth[k * per:(k + 1) * per].contiguous(),
                                          update_moving=(k == 0), per_image=True)
                        r.record_stream(main)
                    outs.append(r)
                for s in side:
                    main.wait_stream(s)
                self._unpad_grads()
                return torch.cat(outs)
            _, res = self.tower(lq, truth, update_moving=True, per_image=True, wgrad_stream=os.environ.get("EMD_T_WGRAD_STREAM", "0") == "1")
            self._unpad_grads()
            return res
        results = []
        main = torch.cuda.current_stream()
        side = self._side_streams(min(streams, n_local)) if streams > 1 else []
        for s in side:
            s.wait_stream(main)
        for k in range(n_local):
            sl = slice(k * tower_batch, (k + 1) * tower_batch)
            if side:
                with torch.cuda.stream(side[k % len(side)]):
                    _, res = self.tower(lq[sl].contiguous(), truth[sl].contiguous(), update_moving=(k == 0))
                    res.record_stream(main)
            else:
                _, res = self.tower(lq[sl].contiguous(), truth[sl].contiguous(), update_moving=(k == 0))
            results.append(res)
        for s in side:
            main.wait_stream(s)
        self._unpad_grads()
        return torch.stack(results)

    def apply_gradients(self, n_sets, learning_rate=None):
        """Nesterov step on the summed gradient vector (n_sets = towers x ranks, :1040) + re-pack of the weights."""
        TO.nesterov_step(self.params, self.grads, self.accum, self.lr if learning_rate is None else learning_rate,
                         self.momentum, grad_scale=1.0 / n_sets)
        self.repack()

    def _warm(self):
        """Launch every kernel family once outside of stream capture (code objects load on first use)."""
        import torch

        z = torch.zeros((1, 32, 32, 1), dtype=torch.float32, device=self.device)
        keep = self.grads.clone(), {k: t.clone() for k, t in self.pad_g.items()}
        self.tower(z, z, update_moving=False)
        self.grads.copy_(keep[0])
        for k, t in keep[1].items():
            self.pad_g[k].copy_(t)
        torch.cuda.synchronize()

    def _capture(self, lq, truth, tower_batch, streams, group, batched=False):
        """Capture local_gradients for this input shape into a hipGraph.  Returns (graph, static lq, static truth,
        static results), or None if capture failed on ANY rank (all ranks then run eagerly: a rank that skipped the
        graph while the others replay it would still meet them at the all-reduce, but the decision must be common
        so that every rank's step has the same structure)."""
        import torch
        import torch.distributed as dist

        entry, err = None, None
        try:
            self._warm()
            slq, str_ = torch.empty_like(lq), torch.empty_like(truth)
            g = torch.cuda.CUDAGraph()
            # thread_local: other threads (e.g. the RCCL watchdog polling its events) may touch the runtime meanwhile
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                sres = self.local_gradients(slq, str_, tower_batch, streams, batched)
            entry = (g, slq, str_, sres)
        except Exception as e:  # noqa: BLE001 -- fall back to eager launches
            err = e
            torch.cuda.synchronize()
        ok = torch.tensor([1 if entry is not None else 0], dtype=torch.int32, device=self.device)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            import warnings

            warnings.warn(f"hipGraph capture of the training step failed ({err}); launching eagerly")
            return None
        return entry

    def train_step(self, lq, truth, tower_batch=1, learning_rate=None, group=None, streams=1, graph=False, batched=False):
        """One optimizer step on this rank's images (misc_py/denoiser-multi-gpu.py:1169-1206): every ``tower_batch``
        images form a tower (gradient set); all sets of all ranks are averaged (:1040) and applied with Nesterov
        momentum (:1064-1066).  Returns the device tensor [n_towers_local, 3] of (mse, loss, factor).
        graph=True: the whole local part (all towers on their streams) is captured once per input shape into a
        hipGraph and replayed -- ~9000 launches per step become one."""
        import torch

        n_local = lq.shape[0] // tower_batch
        key = (tuple(lq.shape), tower_batch, streams, bool(batched))
        if graph and key not in self._graphs:
            self._graphs[key] = self._capture(lq, truth, tower_batch, streams, group, batched)
        if not graph or self._graphs[key] is None:
            results = self.local_gradients(lq, truth, tower_batch, streams, batched)
        else:
            g, slq, str_, sres = self._graphs[key]
            slq.copy_(lq)
            str_.copy_(truth)
            g.replay()
            results = sres.clone()
        world = sync_gradients(self.grads, self.moving, group)
        self.apply_gradients(n_local * world, learning_rate)
        return results


def get_model_fn(num_gpus, variable_strategy="GPU", num_workers=1, trainer=None):
    """The reference's model-function factory (misc_py/denoiser-multi-gpu.py:634-717; machine_learning/denoiser.py:463-539):
    ``get_model_fn(num_gpus, variable_strategy, num_workers)`` -> ``_model_fn(features, labels, mode, params)`` ->
    ``[tower_losses, tower_preds, tower_mses, update_ops] + tower_grads``.

    Here one process drives ONE GPU (the towers of a rank; the reference's in-graph towers across GPUs are ranks under
    torch.distributed), so ``num_gpus`` is the number of towers this call evaluates -- ``len(features)`` must equal
    ``max(num_gpus, 1)`` -- and ``variable_strategy`` / ``num_workers`` (where TensorFlow placed the variables) are accepted and have
    nothing to decide: the variables live on the trainer's device.  ``trainer`` (or ``params['trainer']`` / ``params.trainer``) is the
    DenoiserTrainer that owns weights, gradient vector and moving statistics.

    _model_fn follows the reference line by line:
    * tower i = ``_tower_fn(is_training, features[i], labels[i])`` (:752-782), which uses ONLY image 0 of the tower's shard
      (``feature[0]``, :763): forward in training mode, mse, the Huberised loss (:772-773), ``tf.gradients`` w.r.t. every trainable;
    * ``tower_grads[i]`` = list of that tower's gradients in ``tf.trainable_variables()`` order (= DenoiserTrainer.trainable's);
      they are SEPARATE sets (copies of the trainer's gradient vector, which is zeroed between towers);
    * ``update_ops`` = the moving-statistics assignments of tower 0 only (:701-707): here a dict name -> new value, already applied to
      the trainer (the reference applies them when the train op runs, :1074);
    * ``_tower_preds = tf.stack(preds)`` stacks the LAST tower's prediction only (:711 -- the loop variable, not the list); reproduced.
    ``mode`` falsy (is_training False) would differentiate the inference-mode graph, which the reference never runs and which is not
    built: ValueError."""

    def _model_fn(features, labels=None, mode=None, params=None):
        import torch

        tr = trainer
        if tr is None and params is not None:
            tr = params.get("trainer") if isinstance(params, dict) else getattr(params, "trainer", None)
        if tr is None:
            raise ValueError("pass trainer=DenoiserTrainer(...) to get_model_fn, or params['trainer']")
        if not mode:
            raise ValueError("mode must be true (training): the reference's train loop never evaluates the towers with is_training False")
        num_devices = max(int(num_gpus), 1)
        if len(features) != num_devices or labels is None or len(labels) != num_devices:
            raise ValueError(f"features / labels must be lists of {num_devices} per-tower tensors (input_fn's shard lists)")
        tower_losses, tower_grads, tower_mses, preds, update_ops = [], [], [], None, {}
        for i in range(num_devices):
            f0 = features[i][0:1].contiguous()        # feature[0] (:763)
            t0 = labels[i][0:1].contiguous()
            tr.zero_grad()
            before = tr.moving.clone() if i == 0 else None
            preds, res = tr.tower(f0, t0, update_moving=(i == 0))
            tr._unpad_grads()
            tower_mses.append(res[0])
            tower_losses.append(res[1])
            tower_grads.append([g.clone() for g in tr.g.values()])
            if i == 0:
                changed = (tr.moving != before)
                update_ops = {n: t.clone() for n, t in tr.m.items()}
                update_ops["__changed__"] = bool(changed.any().item())
        return [tuple(tower_losses), torch.stack([preds]), tuple(tower_mses), update_ops] + tower_grads

    return _model_fn
