"""GPU parity tests for the training-path kernels (D', misc_py/denoiser-multi-gpu.py:752-782, :1011-1077), op by op:
each backward entry point of libemdenoise.so (through the C ABI via emdenoise.train_ops) against PyTorch-CPU
autograd (float64) of the oracle's restatement of the forward op (oracle/tf_ops.py) on the same seeded inputs.
Tolerances (relative L2):
  fp32 VALU kernels with fp32 block sums + float atomics (weight gradients) ... 2e-5
  fp32 elementwise / gather kernels ............................................. 2e-6
  split-bf16 matrix-core data gradients ......................................... 2e-5
"""
import numpy as np
import pytest
import torch

from tests.test_ops_gpu import dev, out_act, rel_l2, rnd, t64, to_act

pytestmark = pytest.mark.gpu

TOL_WGRAD = 2e-5
TOL_F32 = 2e-6
TOL_X3 = 2e-5


def d32(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())


def leaf(a):
    return t64(a).requires_grad_(True)


# ------------------------------------------------------------------------------------------------ conv weight gradients
@pytest.mark.parametrize("B,H,W,ci,co,k,stride,rate", [
    (2, 16, 16, 64, 64, 1, 1, 1),
    (1, 9, 13, 128, 256, 1, 1, 1),      # ragged M
    (2, 8, 8, 728, 728, 1, 1, 1),       # K, N tails
    (1, 7, 9, 256, 728, 1, 2, 1),       # strided residual projection, odd sizes
    (2, 16, 16, 4, 64, 1, 1, 1),        # the zero-padded 1-channel image (cnn0 / residual0)
    (1, 12, 12, 64, 128, 3, 1, 1),
    (1, 16, 16, 128, 64, 3, 1, 6),      # dilated ASPP branch
    (2, 10, 6, 64, 64, 3, 2, 1),
    (2, 64, 64, 64, 128, 1, 1, 1),      # several M splits
    # the layer shapes of ONE 512 x 512 tower (BASELINE configs[3]): where the M split sized to whole rounds of resident workgroups,
    # the transposed staging and the atomics run at the sizes training uses (the gradient of the whole tower cannot be held to a
    # tight bar there -- relu6 mask flips, test_train_gpu.py -- so each backward kernel is, on its own)
    (1, 512, 512, 64, 64, 1, 1, 1),     # deconv0_b pointwise: M = 262144
    (1, 512, 512, 128, 64, 1, 1, 1),    # deconv0_a pointwise / residual0_d
    (1, 256, 256, 384, 128, 1, 1, 1),   # deconv1_a pointwise / residual1_d
    (1, 128, 128, 384, 256, 1, 1, 1),   # deconv2_a
    (1, 512, 512, 4, 128, 1, 2, 1),     # residual0: the zero-padded 1-channel image, stride 2 (conv_wgrad_k4_kernel, round 4)
    (1, 512, 512, 4, 64, 1, 1, 1),      # cnn0's pointwise conv: the same, plain
    (2, 9, 13, 4, 24, 1, 1, 1),         # ... ragged: a slab tail, a column tail
    (1, 32, 32, 728, 728, 3, 1, 18),    # ASPP rate-18 branch at the tower's 32 x 32: most taps fall into the padding
])
def test_conv_wgrad(B, H, W, ci, co, k, stride, rate):
    from emdenoise import train_ops as TO
    from oracle import tf_ops as T

    x = rnd((B, H, W, ci), 1)
    w = leaf(rnd((k, k, ci, co), 2, 0.1))
    Ho, Wo = -(-H // stride), -(-W // stride)
    dy = rnd((B, Ho, Wo, co), 3)
    y = T.conv2d_t(t64(x), w, None, stride=stride, rate=rate)
    (ref,) = torch.autograd.grad(y, w, t64(dy))
    dw = torch.zeros((k * k, ci, co), dtype=torch.float32, device=dev())
    if k == 1:
        TO.conv_wgrad(to_act(x, ld=ci + 8, c0=4), to_act(dy), dw, [0], [0], sa=stride)
    else:
        tdy, tdx = TO.conv_taps(H, W, stride, rate)
        TO.conv_wgrad(to_act(x), to_act(dy, ld=co + 4, c0=0), dw, tdy, tdx, sa=stride)
    torch.cuda.synchronize()
    assert rel_l2(dw.cpu().numpy().reshape(k, k, ci, co), ref.numpy()) < TOL_WGRAD


def test_conv_wgrad_accumulates():
    from emdenoise import train_ops as TO

    x, dy = rnd((1, 8, 8, 64), 4), rnd((1, 8, 8, 64), 5)
    dw = torch.zeros((1, 64, 64), dtype=torch.float32, device=dev())
    TO.conv_wgrad(to_act(x), to_act(dy), dw)
    once = dw.cpu().numpy().copy()
    TO.conv_wgrad(to_act(x), to_act(dy), dw)
    torch.cuda.synchronize()
    assert rel_l2(dw.cpu().numpy(), 2 * once) < 1e-6


@pytest.mark.parametrize("B,H,W,ci,co", [(2, 5, 7, 64, 64), (1, 8, 8, 256, 128)])
def test_deconv_wgrad_and_dgrad(B, H, W, ci, co):
    """conv2d_transpose [3,3,Cout,Cin]: dW through emd_conv_wgrad_f32 (a := dL/dy, sa = 2), dx through the forward
    stride-2 conv with the same kernel read as [kh,kw,in=Cout,out=Cin]."""
    from emdenoise import ops, train_ops as TO
    from oracle import tf_ops as T

    x = leaf(rnd((B, H, W, ci), 6))
    w = leaf(rnd((3, 3, co, ci), 7, 0.1))
    dy = rnd((B, 2 * H, 2 * W, co), 8)
    y = T.conv2d_transpose_s2_t(x, w, None)
    gx, gw = torch.autograd.grad(y, (x, w), t64(dy))
    dw = torch.zeros((9, co, ci), dtype=torch.float32, device=dev())
    tdy, tdx = TO.conv_taps(2 * H, 2 * W, 2, 1)
    TO.conv_wgrad(to_act(dy), to_act(x.detach().numpy().astype(np.float32)), dw, tdy, tdx, sa=2)
    wdev = d32(w.detach().numpy().reshape(9, co, ci))
    pk = TO.DevPackedWeights(9, co, ci, dev()).pack(wdev, 9, cout_major=False)
    dx = out_act(B, H, W, ci)
    ones, zeros = torch.ones(ci, device=dev()), torch.zeros(ci, device=dev())
    ops.conv3x3(to_act(dy), pk, ones, zeros, dx, stride=2, act=False)
    torch.cuda.synchronize()
    assert rel_l2(dw.cpu().numpy().reshape(3, 3, co, ci), gw.numpy()) < TOL_WGRAD
    assert rel_l2(dx.torch().cpu().numpy(), gx.numpy()) < TOL_X3


# ------------------------------------------------------------------------------------------------ device packing + data gradients
def test_pack_weights_dev_matches_host_pack():
    from emdenoise import ops, train_ops as TO

    w = rnd((9, 100, 72), 9)
    host = ops.PackedWeights(w, False, dev())
    devp = TO.DevPackedWeights(9, 100, 72, dev()).pack(d32(w), 9, cout_major=False)
    torch.cuda.synchronize()
    assert torch.equal(host.hi, devp.hi) and torch.equal(host.lo, devp.lo)
    wt = np.ascontiguousarray(w.transpose(0, 2, 1))
    host = ops.PackedWeights(wt, True, dev())
    devp = TO.DevPackedWeights(9, 100, 72, dev()).pack(d32(wt), 9, cout_major=True)
    torch.cuda.synchronize()
    assert torch.equal(host.hi, devp.hi) and torch.equal(host.lo, devp.lo)
    # the tap subsets of the transposed conv
    wd = rnd((3, 3, 40, 24), 10)
    for ph, hp in enumerate(ops.pack_deconv(wd, dev())):
        sel = [ky * 3 + kx for (ky, kx) in ops.deconv_phase_taps(ph)]
        dp = TO.DevPackedWeights(len(sel), 24, 40, dev()).pack(d32(wd.reshape(9, 40, 24)), 9, cout_major=True, tap_sel=sel)
        torch.cuda.synchronize()
        assert torch.equal(hp.hi, dp.hi) and torch.equal(hp.lo, dp.lo)


def test_pack_weights_batch_equals_the_single_packs():
    """emd_pack_weights_batch_dev (one launch for a whole job table) writes what the same packs write one by one: both
    orientations, reversed taps, the tap subsets of a transposed conv, a 4-channel pad, sizes that end inside a block."""
    from emdenoise import ops, train_ops as TO

    specs = [(1, 100, 72, False, None), (1, 72, 100, True, [0]), (9, 64, 36, False, None), (9, 36, 64, True, list(range(9))[::-1]),
             (1, 4, 64, False, None), (1, 728, 728, False, None)]
    ws, single, batched = [], [], []
    pb = TO.PackBatch(dev())
    for i, (taps, ci, co, cm, sel) in enumerate(specs):
        shape = (taps, co, ci) if cm else (taps, ci, co)
        w = d32(rnd(shape, 30 + i))
        ws.append(w)
        single.append(TO.DevPackedWeights(taps, ci, co, dev()).pack(w, taps, cout_major=cm, tap_sel=sel))
        b = TO.DevPackedWeights(taps, ci, co, dev())
        b.hi.fill_(-1); b.lo.fill_(-1)
        pb.add(b, w, taps, cout_major=cm, tap_sel=sel)
        batched.append(b)
    wd = d32(rnd((9, 40, 24), 40))
    for ph in range(4):
        sel = [ky * 3 + kx for (ky, kx) in ops.deconv_phase_taps(ph)]
        single.append(TO.DevPackedWeights(len(sel), 24, 40, dev()).pack(wd, 9, cout_major=True, tap_sel=sel))
        b = TO.DevPackedWeights(len(sel), 24, 40, dev())
        pb.add(b, wd, 9, cout_major=True, tap_sel=sel)
        batched.append(b)
    pb.run()
    pb.run()   # the table is built once and reused
    torch.cuda.synchronize()
    for a, b in zip(single, batched):
        assert torch.equal(a.hi, b.hi) and torch.equal(a.lo, b.lo)


@pytest.mark.parametrize("B,H,W,ci,co,k,stride,rate", [
    (2, 16, 16, 64, 128, 1, 1, 1), (1, 8, 8, 728, 728, 1, 1, 1), (1, 16, 16, 128, 64, 3, 1, 6), (1, 9, 11, 64, 64, 3, 1, 1),
    (2, 16, 16, 128, 256, 1, 2, 1), (1, 7, 9, 256, 728, 1, 2, 1),
])
def test_conv_data_gradient(B, H, W, ci, co, k, stride, rate):
    """dx of the 1x1 / 3x3 convs = the forward implicit GEMM on dL/dy with W packed transposed (+ taps reversed)."""
    from emdenoise import ops, train_ops as TO
    from oracle import tf_ops as T

    x = leaf(rnd((B, H, W, ci), 11))
    w = rnd((k, k, ci, co), 12, 0.1)
    Ho, Wo = -(-H // stride), -(-W // stride)
    dy = rnd((B, Ho, Wo, co), 13)
    y = T.conv2d_t(x, t64(w), None, stride=stride, rate=rate)
    (ref,) = torch.autograd.grad(y, x, t64(dy))
    taps = k * k
    # GEMM K = co, N = ci: the TF array [taps][ci][co] read "cout_major" with the roles swapped
    pk = TO.DevPackedWeights(taps, co, ci, dev()).pack(d32(w.reshape(taps, ci, co)), taps, cout_major=True,
                                                      tap_sel=list(range(taps))[::-1])
    ones, zeros = torch.ones(ci, device=dev()), torch.zeros(ci, device=dev())
    if stride == 1:
        dx = out_act(B, H, W, ci)
        (ops.conv1x1 if k == 1 else ops.conv3x3)(to_act(dy), pk, ones, zeros, dx, act=False, **({"rate": rate} if k == 3 else {}))
        got = dx.torch().cpu().numpy()
    else:
        base = rnd((B, H, W, ci), 14)
        dx = to_act(base.copy())
        TO.conv1x1_s2_bwd_data(to_act(dy), pk, ones, zeros, dx, accumulate=True)
        got = dx.torch().cpu().numpy() - base
    torch.cuda.synchronize()
    assert rel_l2(got, ref.numpy()) < TOL_X3


# ------------------------------------------------------------------------------------------------ depthwise
@pytest.mark.parametrize("B,H,W,Cc,stride", [(2, 16, 16, 64, 1), (1, 13, 9, 128, 1), (1, 8, 8, 728, 1), (2, 16, 16, 64, 2),
                                             (1, 9, 7, 256, 2), (2, 12, 12, 4, 1), (2, 64, 64, 64, 1), (1, 70, 80, 128, 1), (3, 64, 72, 4, 1),
                                             # one 512 x 512 tower's layers (rolling weight-gradient kernel at full height, stride-2 blocks)
                                             (1, 512, 512, 64, 1), (1, 256, 256, 128, 1), (1, 512, 512, 64, 2), (1, 128, 128, 384, 1)])
def test_dw3x3_backward(B, H, W, Cc, stride):
    from emdenoise import ops, train_ops as TO
    from oracle import tf_ops as T

    x, w = leaf(rnd((B, H, W, Cc), 15)), leaf(rnd((3, 3, Cc, 1), 16, 0.4))
    Ho, Wo = -(-H // stride), -(-W // stride)
    dy = rnd((B, Ho, Wo, Cc), 17)
    gx, gw = torch.autograd.grad(T.depthwise_conv2d_t(x, w, stride, 1), (x, w), t64(dy))
    xa, dya = to_act(x.detach().numpy().astype(np.float32), ld=Cc + 8, c0=4), to_act(dy)
    dw = torch.zeros((9, Cc), dtype=torch.float32, device=dev())
    TO.dw3x3_wgrad(xa, dya, dw, stride=stride)
    wdev = d32(w.detach().numpy().reshape(9, Cc))
    dx = TO.dw3x3_bwd_data(dya, wdev, out_act(B, H, W, Cc), stride=stride)
    torch.cuda.synchronize()
    assert rel_l2(dw.cpu().numpy().reshape(3, 3, Cc, 1), gw.numpy()) < TOL_WGRAD
    assert rel_l2(dx.torch().cpu().numpy(), gx.numpy()) < TOL_F32
    if stride == 1 and Cc >= 8:  # the fast path the engine uses: the forward kernel with the taps reversed
        dx2 = ops.dw3x3(dya, wdev.flip(0).contiguous(), out_act(B, H, W, Cc))
        torch.cuda.synchronize()
        assert rel_l2(dx2.torch().cpu().numpy(), gx.numpy()) < TOL_F32


@pytest.mark.parametrize("B,H,W,ci", [(2, 12, 16, 64), (1, 5, 7, 128), (2, 64, 64, 64), (1, 72, 88, 24)])
def test_conv3x3_cout1_backward(B, H, W, ci):
    from emdenoise import train_ops as TO
    from oracle import tf_ops as T

    x, w = leaf(rnd((B, H, W, ci), 18)), leaf(rnd((3, 3, ci, 1), 19, 0.1))
    dy = rnd((B, H, W, 1), 20)
    # (sum over channels of the depthwise conv == the conv to one channel; torch's CPU conv backward rejects Cout = 1 here)
    gx, gw = torch.autograd.grad(T.depthwise_conv2d_t(x, w).sum(-1, keepdim=True), (x, w), t64(dy))
    dw = torch.zeros((9, ci), dtype=torch.float32, device=dev())
    TO.conv3x3_cout1_wgrad(to_act(x.detach().numpy().astype(np.float32)), d32(dy), dw)
    dx = TO.conv3x3_cout1_bwd_data(d32(dy), d32(w.detach().numpy().reshape(9, ci)), out_act(B, H, W, ci))
    torch.cuda.synchronize()
    assert rel_l2(dw.cpu().numpy().reshape(3, 3, ci, 1), gw.numpy()) < TOL_WGRAD
    assert rel_l2(dx.torch().cpu().numpy(), gx.numpy()) < TOL_F32


# ------------------------------------------------------------------------------------------------ resampling
@pytest.mark.parametrize("Hi,Wi,Ho,Wo,Cc", [(4, 4, 16, 16, 256), (8, 8, 8, 8, 64), (3, 5, 12, 20, 64), (2, 2, 4, 4, 728),
                                            (1, 1, 2, 2, 64), (5, 5, 9, 9, 8)])
def test_resize_bilinear_backward(Hi, Wi, Ho, Wo, Cc):
    from emdenoise import train_ops as TO
    from oracle import tf_ops as T

    x = leaf(rnd((2, Hi, Wi, Cc), 21))
    dy = rnd((2, Ho, Wo, Cc), 22)
    (ref,) = torch.autograd.grad(T.resize_bilinear_legacy_t(x, Ho, Wo), x, t64(dy))
    dx = TO.resize_bilinear_bwd(to_act(dy, ld=Cc + 4, c0=4), out_act(2, Hi, Wi, Cc))
    torch.cuda.synchronize()
    assert rel_l2(dx.torch().cpu().numpy(), ref.numpy()) < TOL_F32


@pytest.mark.parametrize("H,W,Cc", [(8, 8, 64), (7, 5, 128), (2, 2, 728), (1, 1, 64)])
def test_avgpool2x2_backward(H, W, Cc):
    from emdenoise import train_ops as TO
    from oracle import tf_ops as T

    x = leaf(rnd((2, H, W, Cc), 23))
    dy = rnd((2, -(-H // 2), -(-W // 2), Cc), 24)
    (ref,) = torch.autograd.grad(T.avg_pool2x2_same_t(x), x, t64(dy))
    dx = TO.avgpool2x2_bwd(to_act(dy), out_act(2, H, W, Cc))
    torch.cuda.synchronize()
    assert rel_l2(dx.torch().cpu().numpy(), ref.numpy()) < TOL_F32


def test_axpy():
    from emdenoise import train_ops as TO

    x, y = rnd((2, 5, 7, 64), 25), rnd((2, 5, 7, 64), 26)
    ya = to_act(y, ld=96, c0=32)
    TO.axpy(to_act(x), ya, alpha=0.5)
    torch.cuda.synchronize()
    assert rel_l2(ya.torch().cpu().numpy(), y + 0.5 * x) < 1e-7


# ------------------------------------------------------------------------------------------------ batch norm (training mode)
def _bn_train_t(r, gamma, beta, eps=1e-3):
    mean, var = r.mean(dim=(0, 1, 2)), r.var(dim=(0, 1, 2), unbiased=False)
    return (r - mean) / torch.sqrt(var + eps) * gamma + beta


@pytest.mark.parametrize("double", [True, False])
@pytest.mark.parametrize("B,H,W,Cc,mask", [(2, 8, 8, 64, 1), (1, 16, 16, 728, 1), (2, 32, 32, 1, 2), (3, 5, 7, 128, 0),
                                           (1, 512, 512, 64, 1), (1, 256, 256, 128, 1)])   # one 512 x 512 tower's largest layers
def test_bn_train_forward_and_backward(double, B, H, W, Cc, mask):
    """r -> [BN1] -> BN2 -> relu6 [-> clip]: forward affine, moving-average updates and the backward, against
    autograd through two explicit batch-statistic normalisations (the outer one computes ITS statistics from the
    inner one's output numerically, as TensorFlow does)."""
    from emdenoise import ops, train_ops as TO

    if double and Cc == 1:
        pytest.skip("the 1-channel layer has a single batch norm")
    r = leaf(rnd((B, H, W, Cc), 27, 1.5) + 0.7)
    g1, b1 = leaf(rnd((Cc,), 28, 0.3) + 1.0), leaf(rnd((Cc,), 29, 0.3))
    g2, b2 = leaf(rnd((Cc,), 30, 0.3) + 1.2), leaf(rnd((Cc,), 31, 0.5) + (0.4 if mask == 2 else 1.0))
    dy = rnd((B, H, W, Cc), 32)
    z1 = _bn_train_t(r, g1, b1) if double else r
    z = _bn_train_t(z1, g2, b2)
    y = torch.clamp(z, 0.0, 6.0) if mask else z
    if mask == 2:
        y = torch.clamp(y, 0.0, 1.0)
    grads = torch.autograd.grad(y, (r, g1, b1, g2, b2) if double else (r, g2, b2), t64(dy), allow_unused=True)

    f32 = lambda t: d32(t.detach().numpy())
    ra = to_act(r.detach().numpy().astype(np.float32))
    mean, var = ops.bn_batch_stats(ra)
    npix = B * H * W
    mm = [torch.zeros(Cc, device=dev()), torch.ones(Cc, device=dev()), torch.zeros(Cc, device=dev()), torch.ones(Cc, device=dev())]
    fold = TO.bn_train_fold(mean, var, f32(g2), f32(b2), npix, gamma1=f32(g1) if double else None,
                            beta1=f32(b1) if double else None, moving=mm if double else mm[2:])
    # forward
    if Cc % 4 == 0:
        ya = ops.affine_act(ra, fold["scale"], fold["shift"], out_act(B, H, W, Cc), act=ops.ACT_RELU6 if mask else ops.ACT_NONE)
        torch.cuda.synchronize()
        assert rel_l2(ya.torch().cpu().numpy(), y.detach().numpy()) < 3e-6
    # moving statistics: decay 0.999 from (0, 1), unbiased batch variance
    bessel = npix / (npix - 1)
    rd = r.detach()
    if double:
        v1 = rd.var(dim=(0, 1, 2), unbiased=False)
        assert rel_l2(mm[0].cpu().numpy(), 0.001 * rd.mean(dim=(0, 1, 2)).numpy()) < 1e-5
        assert rel_l2(mm[1].cpu().numpy(), 0.999 + 0.001 * bessel * v1.numpy()) < 1e-6
        z1d = z1.detach()
        assert rel_l2(mm[2].cpu().numpy(), 0.001 * z1d.mean(dim=(0, 1, 2)).numpy()) < 1e-5
        assert rel_l2(mm[3].cpu().numpy(), 0.999 + 0.001 * bessel * z1d.var(dim=(0, 1, 2), unbiased=False).numpy()) < 1e-6
    else:
        assert rel_l2(mm[2].cpu().numpy(), 0.001 * rd.mean(dim=(0, 1, 2)).numpy()) < 1e-5
        assert rel_l2(mm[3].cpu().numpy(), 0.999 + 0.001 * bessel * rd.var(dim=(0, 1, 2), unbiased=False).numpy()) < 1e-6
    # backward
    dg1, dg2, db2 = (torch.zeros(Cc, device=dev()) for _ in range(3))
    dya = to_act(dy)
    dr = TO.bn_backward(dya, ra, fold, f32(g2), dg2, db2, dya, mask=mask, gamma1=f32(g1) if double else None,
                        dgamma1=dg1 if double else None)
    torch.cuda.synchronize()
    if double:
        gr, gg1, gb1, gg2, gb2 = grads
        scale = float(gg2.abs().max())
        assert np.abs(dg1.cpu().numpy() - gg1.numpy()).max() < 2e-5 * scale    # d gamma1 ~ eps-suppressed
        assert gb1 is None or float(gb1.abs().max()) < 1e-9 * scale            # d beta1 = 0 analytically
    else:
        gr, gg2, gb2 = grads
    assert rel_l2(dr.torch().cpu().numpy(), gr.numpy()) < 1e-5
    assert rel_l2(dg2.cpu().numpy(), gg2.numpy()) < 1e-5
    assert rel_l2(db2.cpu().numpy(), gb2.numpy()) < 1e-5


@pytest.mark.parametrize("B,H,W,Cc", [(2, 256, 256, 128), (3, 512, 512, 64)])
def test_bn_backward_per_image_at_tower_sizes(B, H, W, Cc):
    """The batched-towers path (emd_bn_stats_images / _train_fold_images / _bn_bwd_{reduce,prep,apply}_images_f32) at the pixel
    counts of a 512 x 512 tower: image b of the batch gives the bits of the one-image kernels run on it alone (the promise behind
    DenoiserTrainer.tower(per_image=True)), and image 0 matches float64 autograd through the double batch norm + relu6."""
    from emdenoise import ops, train_ops as TO

    r = rnd((B, H, W, Cc), 127, 1.5) + 0.7
    dy = rnd((B, H, W, Cc), 132)
    g1, b1 = rnd((Cc,), 128, 0.3) + 1.0, rnd((Cc,), 129, 0.3)
    g2, b2 = rnd((Cc,), 130, 0.3) + 1.2, rnd((Cc,), 131, 0.5) + 1.0
    ra, dya = to_act(r), to_act(dy)
    npix = H * W
    mean, var = ops.bn_batch_stats_images(ra)
    fold = TO.bn_train_fold(mean, var, d32(g2), d32(b2), npix, gamma1=d32(g1), beta1=d32(b1), images=B)
    dg1, dg2, db2 = (torch.zeros(Cc, device=dev()) for _ in range(3))
    dr = TO.bn_backward(dya, ra, fold, d32(g2), dg2, db2, out_act(B, H, W, Cc), mask=1, gamma1=d32(g1), dgamma1=dg1)
    torch.cuda.synchronize()
    acc1, acc2, accb = (torch.zeros(Cc, device=dev()) for _ in range(3))
    for b in range(B):
        rb, dyb = to_act(r[b:b + 1]), to_act(dy[b:b + 1])
        m1, v1 = ops.bn_batch_stats(rb)
        f1 = TO.bn_train_fold(m1, v1, d32(g2), d32(b2), npix, gamma1=d32(g1), beta1=d32(b1))
        e1, e2, eb = (torch.zeros(Cc, device=dev()) for _ in range(3))
        drb = TO.bn_backward(dyb, rb, f1, d32(g2), e2, eb, out_act(1, H, W, Cc), mask=1, gamma1=d32(g1), dgamma1=e1)
        torch.cuda.synchronize()
        assert torch.equal(drb.torch()[0], dr.torch()[b]), f"image {b}: the batched per-image kernels must give the one-image kernels' bits"
        acc1 += e1; acc2 += e2; accb += eb
    # parameter gradients: the batched form adds every image's contribution with float atomics / in another order: 1e-6
    for got, want in ((dg1, acc1), (dg2, acc2), (db2, accb)):
        assert float((got - want).norm() / want.norm().clamp_min(1e-20)) < 2e-6
    # image 0 against float64 autograd
    rl, l1, lb1, l2, lb2 = leaf(r[0:1]), leaf(g1), leaf(b1), leaf(g2), leaf(b2)
    z = _bn_train_t(_bn_train_t(rl, l1, lb1), l2, lb2)
    y = torch.clamp(z, 0.0, 6.0)
    (gr,) = torch.autograd.grad(y, rl, t64(dy[0:1]))
    # of 10^7 units one or two sit within float32 rounding of a relu6 kink, where the float32 forward and the float64 oracle pick
    # different masks (a unit-sized error each): compared away from the kinks
    away = ((z.detach().abs() > 1e-4) & ((z.detach() - 6.0).abs() > 1e-4)).numpy()
    got0 = dr.torch()[0:1].cpu().numpy()
    assert away.mean() > 0.999
    assert rel_l2(got0[away], gr.numpy()[away]) < 1e-5


def test_bias_gradient_reduction():
    from emdenoise import train_ops as TO

    dy = rnd((2, 9, 9, 728), 33)
    s1 = torch.full((728,), 1.0, device=dev())
    TO.chan_reduce(to_act(dy), s1, accumulate_s1=True)
    torch.cuda.synchronize()
    assert rel_l2(s1.cpu().numpy(), 1.0 + dy.astype(np.float64).sum(axis=(0, 1, 2))) < 1e-6


# ------------------------------------------------------------------------------------------------ loss and optimizer
@pytest.mark.parametrize("noise", [0.01, 0.2])   # both branches of the capped loss
def test_denoise_loss(noise):
    from emdenoise import train_ops as TO

    truth = rnd((2, 64, 64, 1), 34, 0.2, positive=True)
    out = leaf(truth + rnd((2, 64, 64, 1), 35, noise))
    mse = ((out - t64(truth)) ** 2).mean()
    loss = 1000.0 * mse if float(mse.detach()) < 0.001 else torch.sqrt(1000.0 * mse)
    (ref,) = torch.autograd.grad(loss, out)
    dout = torch.empty((2, 64, 64, 1), dtype=torch.float32, device=dev())
    res = TO.denoise_loss(d32(out.detach().numpy()), d32(truth), dout).cpu().numpy()
    assert abs(res[0] - float(mse)) < 1e-6 * float(mse) + 1e-12 and abs(res[1] - float(loss)) < 2e-6 * float(loss)
    assert rel_l2(dout.cpu().numpy(), ref.numpy()) < 2e-6
    assert (float(mse) < 0.001) == (noise == 0.01)


def test_nesterov_step_matches_apply_momentum():
    from emdenoise import train_ops as TO

    p, g, a = rnd((1000,), 36).astype(np.float64), rnd((1000,), 37).astype(np.float64), rnd((1000,), 38).astype(np.float64)
    lr, mom, gs = 0.001, 0.9, 0.1
    pd, gd, ad = d32(p), d32(g), d32(a)
    TO.nesterov_step(pd, gd, ad, lr, mom, gs)
    a2 = mom * a + g * gs                 # ApplyMomentum, use_nesterov=true
    p2 = p - (g * gs * lr + a2 * mom * lr)
    torch.cuda.synchronize()
    assert rel_l2(ad.cpu().numpy(), a2) < 1e-6 and rel_l2(pd.cpu().numpy(), p2) < 1e-6


@pytest.mark.parametrize("B,H,W,ci,co,k,stride,rate,images", [
    (2, 32, 32, 728, 728, 1, 1, 1, True),      # the 1/16-resolution flow of two one-image towers (128 x 64 tiles: < 192 tiles of 128)
    (3, 16, 24, 64, 36, 1, 1, 1, True),        # 384 pixels per image = three tiles; channel tail in a 64-column tile
    (2, 64, 64, 128, 256, 1, 2, 1, True),      # strided 1x1 (the encoder's residual convs): the row map, 1024 output pixels per image
    (2, 32, 32, 96, 160, 3, 1, 6, True),       # dense dilated 3x3 (ASPP of graph D'), per-image
    (2, 20, 12, 64, 64, 1, 1, 1, False),       # batch statistics over a ragged M (480 rows: the last tile is partial)
    (4, 64, 64, 64, 128, 1, 1, 1, False),
])
def test_conv_stats_equals_conv_then_statistics(B, H, W, ci, co, k, stride, rate, images):
    """emd_conv1x1_stats_f32 / emd_conv3x3_stats_f32 (round 4: the training forward's batch statistics from the GEMM's epilogue,
    misc_py/denoiser-multi-gpu.py:200-540 phase = True): y bit for bit the plain convolution's, mean / var equal to
    emd_bn_stats_f32 / emd_bn_stats_images_f32 of y to float32 rounding (both sum in double; the partials are cut differently), and
    to the float64 oracle; image b of the batch gets the statistics it gets alone."""
    from emdenoise import ops

    x = rnd((B, H, W, ci), 700, positive=True)
    w = rnd((k * k, ci, co), 701, scale=(2.0 / (k * k * ci + co)) ** 0.5)
    pk = ops.PackedWeights(w, False, dev())
    ones, zeros = torch.ones(co, device=dev()), torch.zeros(co, device=dev())
    xa = to_act(x, ld=ci + 8, c0=4)
    Ho, Wo = -(-H // stride), -(-W // stride)
    y1, y2 = out_act(B, Ho, Wo, co, ld=co + 4, c0=0), out_act(B, Ho, Wo, co, ld=co + 4, c0=0)
    assert ops.conv_stats_supported(xa, stride, images)
    mean, var = ops.conv_stats(xa, pk, ones, zeros, y1, stride=stride, rate=rate, images=images)
    if k == 1:
        ops.conv1x1(xa, pk, ones, zeros, y2, stride=stride, act=False)
    else:
        ops.conv3x3(xa, pk, ones, zeros, y2, rate=rate, act=False)
    m2, v2 = ops.bn_batch_stats_images(y2) if images else ops.bn_batch_stats(y2)
    torch.cuda.synchronize()
    assert torch.equal(y1.torch(), y2.torch())
    yy = y2.torch().double().cpu().numpy()
    ax = (1, 2) if images else (0, 1, 2)
    assert rel_l2(mean.cpu().numpy(), yy.mean(ax).ravel()) < 2e-7 and rel_l2(var.cpu().numpy(), yy.var(ax).ravel()) < 2e-6
    assert rel_l2(mean.cpu().numpy(), m2.cpu().numpy()) < 2e-7 and rel_l2(var.cpu().numpy(), v2.cpu().numpy()) < 2e-6
    if images:    # the statistics of image 1 do not depend on the batch it came in
        xb = to_act(x[1:2], ld=ci + 8, c0=4)
        mo, vo = ops.conv_stats(xb, pk, ones, zeros, out_act(1, Ho, Wo, co), stride=stride, rate=rate, images=True)
        torch.cuda.synchronize()
        assert torch.equal(mo, mean[co:2 * co]) and torch.equal(vo, var[co:2 * co])


def test_conv_stats_rejects_images_that_share_a_tile():
    from emdenoise import _lib, ops

    x = to_act(rnd((2, 10, 10, 64), 710))
    assert not ops.conv_stats_supported(x, 1, images=True)
    pk = ops.PackedWeights(rnd((1, 64, 64), 711, 0.1), False, dev())
    ones, zeros = torch.ones(64, device=dev()), torch.zeros(64, device=dev())
    with pytest.raises(RuntimeError, match="straddle"):
        ops.conv_stats(x, pk, ones, zeros, out_act(2, 10, 10, 64), images=True)


@pytest.mark.parametrize("B,H,W,Cc,double,mask,res", [
    (2, 32, 32, 728, True, 1, True),      # the middle flow of a pair of one-image towers: BN1 -> BN2 -> relu6 (+ residual), 12 channel blocks
    (3, 64, 64, 256, True, 1, False),     # 4096 pixels per image: the largest map the one-launch form takes
    (2, 16, 16, 36, False, 1, False),     # a single norm behind conv + bias, channel tail in a 64-channel block
    (1, 32, 32, 64, False, 2, False),     # relu6 then clip to [0, 1]
    (2, 8, 8, 128, True, 0, False),       # no activation
])
def test_bn_small_one_launch_forms_equal_the_slab_forms(B, H, W, Cc, double, mask, res):
    """emd_bn_train_fwd_small_f32 / emd_bn_train_bwd_small_f32 (round 4) against the launches they replace (emd_bn_stats_images_f32 +
    emd_bn_train_fold_images_f32 + emd_affine_act_images_f32; emd_bn_bwd_reduce / _prep / _apply_images_f32 -- themselves checked against
    float64 autograd above and in test_bn_backward_per_image_at_tower_sizes): same formulas, sums cut differently -> agreement to
    rounding; plus the moving-average update from image 0 and "dx may be written over r"."""
    from emdenoise import ops, train_ops as TO

    r = rnd((B, H, W, Cc), 800, 1.5) + 0.7
    g1, b1 = d32(rnd((Cc,), 801, 0.3) + 1.0), d32(rnd((Cc,), 802, 0.3))
    g2, b2 = d32(rnd((Cc,), 803, 0.3) + 1.2), d32(rnd((Cc,), 804, 0.5) + (0.4 if mask == 2 else 1.0))
    bias = None if double else d32(rnd((Cc,), 805, 0.2))
    dy = rnd((B, H, W, Cc), 806)
    rr = to_act(rnd((B, H, W, Cc), 807, positive=True)) if res else None
    act = {0: ops.ACT_NONE, 1: ops.ACT_RELU6, 2: ops.ACT_RELU6_CLIP01}[mask]
    ra = to_act(r, ld=Cc + 8, c0=4)
    assert TO.bn_small_supported(ra)
    mk = lambda: [torch.zeros(Cc, device=dev()), torch.ones(Cc, device=dev()), torch.zeros(Cc, device=dev()), torch.ones(Cc, device=dev())]
    # the slab forms
    mean, var = ops.bn_batch_stats_images(ra)
    mm_ref = mk()
    fold_ref = TO.bn_train_fold(mean, var, g2, b2, H * W, gamma1=g1 if double else None, beta1=b1 if double else None, bias=bias,
                                moving=mm_ref if double else mm_ref[2:], images=B)
    y_ref = ops.affine_act_images(ra, fold_ref["scale"], fold_ref["shift"], out_act(B, H, W, Cc), act=act, res=rr)
    # one launch
    mm = mk()
    y = out_act(B, H, W, Cc, ld=Cc + 4, c0=0)
    fold = TO.bn_train_fwd_small(ra, g2, b2, y, act, gamma1=g1 if double else None, beta1=b1 if double else None, bias=bias,
                                 moving=mm if double else mm[2:], res=rr)
    torch.cuda.synchronize()
    for k in ("scale", "shift", "rstd1", "mean") + (("rstd2",) if double else ()):
        assert rel_l2(fold[k].cpu().numpy(), fold_ref[k].cpu().numpy()) < 1e-6, k
    assert rel_l2(y.torch().cpu().numpy(), y_ref.torch().cpu().numpy()) < 1e-6
    for a, b in zip(mm, mm_ref):
        assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-6
    assert not np.isnan(y.torch().cpu().numpy()).any() and np.isnan(y.buf.cpu().numpy()[..., Cc:]).all()
    # backward: dx written over a copy of r (as the trainer does)
    bm = {0: TO.MASK_NONE, 1: TO.MASK_RELU6, 2: TO.MASK_RELU6_CLIP}[mask]
    dya = to_act(dy)
    dgr = [torch.zeros(Cc, device=dev()) for _ in range(3)]
    dr_ref = TO.bn_backward(dya, ra, fold_ref, g2, dgr[1], dgr[2], out_act(B, H, W, Cc), mask=bm, gamma1=g1 if double else None,
                            dgamma1=dgr[0] if double else None)
    dg = [torch.zeros(Cc, device=dev()) for _ in range(3)]
    r2 = to_act(r, ld=Cc + 8, c0=4)
    dr = TO.bn_backward_small(dya, r2, fold, g2, dg[1], dg[2], r2, mask=bm, gamma1=g1 if double else None, dgamma1=dg[0] if double else None)
    torch.cuda.synchronize()
    assert rel_l2(dr.torch().cpu().numpy(), dr_ref.torch().cpu().numpy()) < 2e-6
    scale = float(dgr[1].abs().max())
    for a, b in zip(dg, dgr):
        assert np.abs(a.cpu().numpy() - b.cpu().numpy()).max() < 2e-6 * scale + 1e-12
    assert np.isnan(r2.buf.cpu().numpy()[..., :4]).all()        # nothing outside the slice


def test_bn_small_refuses_large_maps():
    from emdenoise import ops, train_ops as TO

    x = to_act(rnd((1, 128, 64, 64), 810))
    assert not TO.bn_small_supported(x)
    g = torch.ones(64, device=dev())
    with pytest.raises(RuntimeError, match="4096"):
        TO.bn_train_fwd_small(x, g, g, out_act(1, 128, 64, 64), ops.ACT_RELU6)


@pytest.mark.parametrize("B,H,W,Cc,stride,rate,images,act", [
    (2, 32, 32, 728, 1, 1, True, 1),      # the 1/16-resolution flow of a batched pass of one-image towers
    (3, 20, 24, 64, 2, 1, True, 1),       # a strided consumer (cnn*_strided), ragged tiles
    (2, 16, 16, 128, 1, 2, False, 1),     # dilated, batch statistics
    (1, 72, 80, 64, 1, 1, False, 2),      # 16-row strips, relu
])
def test_dw3x3_pre_act_equals_affine_then_depthwise(B, H, W, Cc, stride, rate, images, act):
    """emd_dw3x3_pre_act_f32 (round 4: the training step's affine + relu6 in the consumer's loads) == emd_affine_act[_images]_f32 written
    out, then emd_dw3x3_f32 -- bit for bit (padding is applied to the ACTIVATED tensor: a padded tap contributes 0, not act(shift))."""
    from emdenoise import ops

    g = torch.Generator(device=dev()).manual_seed(5)
    r = ops.Act(torch.randn(B, H, W, Cc, device=dev(), generator=g) * 3)
    n = B * Cc if images else Cc
    sc, sh = torch.rand(n, device=dev(), generator=g) + 0.5, torch.randn(n, device=dev(), generator=g)
    wd = torch.randn(9, Cc, device=dev(), generator=g) * 0.3
    y = ops.Act.empty(B, H, W, Cc, dev())
    (ops.affine_act_images if images else ops.affine_act)(r, sc, sh, y, act=act)
    Ho, Wo = -(-H // stride), -(-W // stride)
    want = ops.dw3x3(y, wd, ops.Act.empty(B, Ho, Wo, Cc, dev()), stride=stride, rate=rate)
    out = ops.Act.empty(B, Ho, Wo, Cc, dev())
    out.buf.fill_(float("nan"))
    got = ops.dw3x3_pre_act(ops.PreAct(r, sc, sh, images=images, act=act), wd, out, stride=stride, rate=rate)
    torch.cuda.synchronize()
    assert torch.equal(got.buf, want.buf)


@pytest.mark.parametrize("B,H,W,Cc,stride,rate,images,act", [
    (2, 32, 32, 728, 1, 1, True, 1),
    (2, 64, 80, 64, 1, 1, True, 1),       # the rolling-window form (H, W >= 64)
    (3, 20, 24, 64, 2, 1, False, 1),
    (2, 16, 16, 128, 1, 2, False, 2),
])
def test_dw3x3_wgrad_pre_equals_affine_then_weight_gradient(B, H, W, Cc, stride, rate, images, act):
    """emd_dw3x3_wgrad_pre_f32 == emd_affine_act[_images]_f32 written out, then emd_dw3x3_wgrad_f32 (to the spread of the float atomics
    both end in), and both against a float64 sum of the same products on the host."""
    from emdenoise import ops, train_ops as TO

    g = torch.Generator(device=dev()).manual_seed(6)
    r = ops.Act(torch.randn(B, H, W, Cc, device=dev(), generator=g) * 3)
    n = B * Cc if images else Cc
    sc, sh = torch.rand(n, device=dev(), generator=g) + 0.5, torch.randn(n, device=dev(), generator=g)
    Ho, Wo = -(-H // stride), -(-W // stride)
    dy = ops.Act(torch.randn(B, Ho, Wo, Cc, device=dev(), generator=g))
    y = ops.Act.empty(B, H, W, Cc, dev())
    (ops.affine_act_images if images else ops.affine_act)(r, sc, sh, y, act=act)
    want = torch.zeros(9, Cc, device=dev())
    TO.dw3x3_wgrad(y, dy, want, stride=stride, rate=rate)
    got = torch.zeros(9, Cc, device=dev())
    TO.dw3x3_wgrad_pre(ops.PreAct(r, sc, sh, images=images, act=act), dy, got, stride=stride, rate=rate)
    torch.cuda.synchronize()
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() < 2e-5 * scale
    # float64 on the host: dw[t][c] = sum x[b, oy*s + ky*r - pt, ox*s + kx*r - pl, c] * dy[b, oy, ox, c]  (TF SAME)
    xa, da = y.buf.double().cpu(), dy.buf.double().cpu()
    def pad_before(nn):
        o = -(-nn // stride)
        return max((o - 1) * stride + 2 * rate + 1 - nn, 0) // 2
    pt, pl = pad_before(H), pad_before(W)
    xp = torch.zeros(B, H + 4 * rate + 2, W + 4 * rate + 2, Cc, dtype=torch.float64)
    off = 2 * rate
    xp[:, off:off + H, off:off + W] = xa
    ref = torch.zeros(9, Cc, dtype=torch.float64)
    for ky in range(3):
        for kx in range(3):
            ys = off - pt + ky * rate
            xs = off - pl + kx * rate
            win = xp[:, ys:ys + (Ho - 1) * stride + 1:stride, xs:xs + (Wo - 1) * stride + 1:stride]
            ref[ky * 3 + kx] = (win * da).sum(dim=(0, 1, 2))
    assert (got.double().cpu() - ref).abs().max().item() < 1e-4 * ref.abs().max().item()


@pytest.mark.parametrize("B,H,W,Cc,images,double_bn,mask,stride,rate", [
    (2, 32, 32, 728, True, True, 1, 1, 1),      # the 1/16-resolution flow, a pair of one-image towers
    (2, 72, 80, 64, True, False, 1, 1, 1),      # 16-row strips, a ragged last strip
    (3, 20, 24, 24, False, True, 1, 1, 1),      # batch statistics; W, C not multiples of the workgroup's 16 columns / 64 channels
    (1, 16, 16, 128, False, False, 0, 1, 1),    # no activation mask
    (2, 64, 64, 64, True, True, 1, 2, 1),       # a stride-2 consumer (cnn0_last -> cnn0_strided): the gather form
    (3, 22, 18, 24, False, False, 1, 2, 1),     # ... ragged, batch statistics
    (2, 16, 16, 128, True, True, 1, 1, 2),      # a dilated consumer
])
def test_bn_backward_of_a_never_written_gradient(B, H, W, Cc, images, double_bn, mask, stride, rate):
    """TO.bn_backward_dw (emd_dw3x3_bn_bwd_reduce_f32 / _apply_f32: dy = dw3x3(dd, flipped taps) formed on the fly in both passes) ==
    ops.dw3x3 written out, then TO.bn_backward: dr and the parameter gradients to the rounding of the re-cut double sums."""
    from emdenoise import ops, train_ops as TO

    g = torch.Generator(device=dev()).manual_seed(7)
    rn = lambda *sh: torch.randn(*sh, device=dev(), generator=g)
    r0 = rn(B, H, W, Cc) * 2
    Ho, Wo = -(-H // stride), -(-W // stride)
    dd = ops.Act(rn(B, Ho, Wo, Cc))
    wf = rn(9, Cc) * 0.3
    w0 = wf.flip(0).contiguous()      # the consumer's own taps (wf = reversed)
    gamma2, beta2 = torch.rand(Cc, device=dev(), generator=g) + 0.5, rn(Cc)
    gamma1, beta1 = (torch.rand(Cc, device=dev(), generator=g) + 0.5, rn(Cc)) if double_bn else (None, None)
    rA = ops.Act(r0.clone())
    mean, var = (ops.bn_batch_stats_images if images else ops.bn_batch_stats)(rA)
    npix = H * W if images else B * H * W
    fold = TO.bn_train_fold(mean, var, gamma2, beta2, npix, gamma1=gamma1, beta1=beta1, images=B if images else 0)
    outs = {}
    wg = mask == 1     # the reduction pass also adds the CONSUMER's depthwise weight gradient (x = relu6(r * scale + shift) behind the mask)
    for fused in (False, True):
        r = ops.Act(r0.clone())
        dg2, db2 = torch.zeros(Cc, device=dev()), torch.zeros(Cc, device=dev())
        dg1 = torch.zeros(Cc, device=dev()) if double_bn else None
        gdw = torch.zeros(9, Cc, device=dev())
        if fused:
            TO.bn_backward_dw(TO.DwGrad(dd, wf, gdw if wg else None, stride=stride, rate=rate, hw=(H, W)), r, fold, gamma2, dg2, db2, r, mask=mask,
                              gamma1=gamma1, dgamma1=dg1)
        else:
            if wg:
                TO.dw3x3_wgrad_pre(ops.PreAct(r, fold["scale"], fold["shift"], images=images, act=ops.ACT_RELU6), dd, gdw, stride=stride, rate=rate)
            if stride == 1:
                dy = ops.dw3x3(dd, wf, ops.Act.empty(B, H, W, Cc, dev()), rate=rate)
            else:
                dy = TO.dw3x3_bwd_data(dd, w0, ops.Act.empty(B, H, W, Cc, dev()), stride=stride, rate=rate)
            TO.bn_backward(dy, r, fold, gamma2, dg2, db2, r, mask=mask, gamma1=gamma1, dgamma1=dg1)
        torch.cuda.synchronize()
        outs[fused] = (r.buf.clone(), dg2, db2, dg1, gdw)
    a, b = outs[True], outs[False]
    assert not torch.isnan(a[0]).any()
    assert (a[0] - b[0]).abs().max().item() < 2e-5 * b[0].abs().max().item()
    for u, v in zip(a[1:], b[1:]):
        if u is not None:
            assert (u - v).abs().max().item() < 2e-5 * max(v.abs().max().item(), 1e-3)
    if wg:
        assert b[4].abs().max().item() > 0


@pytest.mark.parametrize("B,H,W,Cc,images", [(2, 16, 24, 128, True), (3, 8, 8, 64, False)])
def test_affine_with_the_residual_before_its_own_affine(B, H, W, Cc, images):
    """emd_affine_act_res_affine_f32 (round 4: the residual projection's norm + relu6 applied where the block adds it) == the residual's
    affine pass written out, then emd_affine_act[_images]_f32 with it as ``res`` -- bit for bit."""
    from emdenoise import ops

    g = torch.Generator(device=dev()).manual_seed(9)
    rn = lambda *sh: torch.randn(*sh, device=dev(), generator=g)
    r, r2 = ops.Act(rn(B, H, W, Cc) * 3), ops.Act(rn(B, H, W, Cc) * 3)
    n = B * Cc if images else Cc
    sc, sh, sc2, sh2 = torch.rand(n, device=dev(), generator=g) + 0.5, rn(n), torch.rand(n, device=dev(), generator=g) + 0.5, rn(n)
    aff = ops.affine_act_images if images else ops.affine_act
    res = aff(r2, sc2, sh2, ops.Act.empty(B, H, W, Cc, dev()), act=ops.ACT_RELU6)
    want = aff(r, sc, sh, ops.Act.empty(B, H, W, Cc, dev()), act=ops.ACT_RELU6, res=res)
    out = ops.Act.empty(B, H, W, Cc, dev())
    out.buf.fill_(float("nan"))
    got = ops.affine_act_res_pre(r, sc, sh, out, ops.PreAct(r2, sc2, sh2, images=images, act=ops.ACT_RELU6), act=ops.ACT_RELU6)
    torch.cuda.synchronize()
    assert torch.equal(got.buf, want.buf)


@pytest.mark.parametrize("B,H,W,ci,co,images", [(2, 16, 16, 64, 64, True), (3, 8, 12, 128, 32, False), (2, 16, 8, 256, 128, True)])
def test_deconv_stats_equals_deconv_then_statistics(B, H, W, ci, co, images):
    """emd_deconv3x3s2_stats_f32 (round 4: the transposed conv of the training forward with the batch statistics of its output from the
    four phase GEMMs' epilogues): y bit for bit emd_deconv3x3s2_f32's, mean / var equal to emd_bn_stats[_images]_f32 of y to float32
    rounding (double sums, other slabs); image 1's statistics do not depend on the batch it came in."""
    from emdenoise import ops

    x = rnd((B, H, W, ci), 720, positive=True)
    w = rnd((3, 3, co, ci), 721, scale=(2.0 / (9 * ci + co)) ** 0.5)
    ph = ops.pack_deconv(w, dev())
    ones, zeros = torch.ones(co, device=dev()), torch.zeros(co, device=dev())
    xa = to_act(x, ld=ci + 8, c0=4)
    y1, y2 = out_act(B, 2 * H, 2 * W, co, ld=co + 4, c0=0), out_act(B, 2 * H, 2 * W, co, ld=co + 4, c0=0)
    mean, var = ops.deconv_stats(xa, ph, ones, zeros, y1, images=images)
    ops.deconv3x3s2(xa, ph, ones, zeros, y2, act=False)
    m2, v2 = ops.bn_batch_stats_images(y2) if images else ops.bn_batch_stats(y2)
    torch.cuda.synchronize()
    assert torch.equal(y1.torch(), y2.torch())
    assert rel_l2(mean.cpu().numpy(), m2.cpu().numpy()) < 2e-7 and rel_l2(var.cpu().numpy(), v2.cpu().numpy()) < 2e-6
    if images:
        xb = to_act(x[1:2], ld=ci + 8, c0=4)
        mo, vo = ops.deconv_stats(xb, ph, ones, zeros, out_act(1, 2 * H, 2 * W, co), images=True)
        torch.cuda.synchronize()
        assert torch.equal(mo, mean[co:2 * co]) and torch.equal(vo, var[co:2 * co])


@pytest.mark.parametrize("B,H,W,Cc", [(2, 32, 32, 728), (1, 72, 80, 64), (3, 20, 24, 24)])
def test_both_depthwise_gradients_in_one_pass(B, H, W, Cc):
    """emd_dw3x3_bwd_both_f32: dx bit for bit emd_dw3x3_f32 on the reversed taps, dw == emd_dw3x3_wgrad_f32 to the spread of its float
    atomics; x is padded into a wider buffer (pitches)."""
    from emdenoise import ops, train_ops as TO

    g = torch.Generator(device=dev()).manual_seed(11)
    rn = lambda *sh: torch.randn(*sh, device=dev(), generator=g)
    xbuf = torch.full((B, H, W, Cc + 8), float("nan"), device=dev())
    xbuf[..., 4:4 + Cc] = rn(B, H, W, Cc)
    x = ops.Act(xbuf, Cc, 4)
    dd = ops.Act(rn(B, H, W, Cc))
    wf = rn(9, Cc) * 0.3
    want_dx = ops.dw3x3(dd, wf, ops.Act.empty(B, H, W, Cc, dev()))
    want_dw = torch.zeros(9, Cc, device=dev())
    TO.dw3x3_wgrad(x, dd, want_dw)
    dx = ops.Act.empty(B, H, W, Cc, dev())
    dx.buf.fill_(float("nan"))
    got_dw = torch.zeros(9, Cc, device=dev())
    TO.dw3x3_bwd_both(dd, wf, x, dx, got_dw)
    torch.cuda.synchronize()
    assert torch.equal(dx.buf, want_dx.buf)
    assert (got_dw - want_dw).abs().max().item() < 2e-5 * want_dw.abs().max().item()


@pytest.mark.parametrize("B,H,W,Cc,images,double_bn", [(2, 24, 40, 64, True, True), (3, 16, 16, 24, False, False)])
def test_bn_backward_of_the_final_convs_data_gradient(B, H, W, Cc, images, double_bn):
    """TO.bn_backward on a TO.Cout1Grad (emd_bn_bwd_reduce_prep_cout1_f32 / emd_bn_bwd_apply_cout1_f32: the 3x3-to-one-channel conv's data
    gradient formed from the 1-channel image in both passes) == emd_conv3x3_cout1_bwd_data_f32 written out, then TO.bn_backward: dr bit for
    bit where the sums agree (same slabs, same order: they do), parameter gradients to the atomics' spread."""
    from emdenoise import ops, train_ops as TO

    g = torch.Generator(device=dev()).manual_seed(13)
    rn = lambda *sh: torch.randn(*sh, device=dev(), generator=g)
    r0 = rn(B, H, W, Cc) * 2
    g1 = rn(B, H, W, 1).contiguous()
    w9 = (rn(9, Cc) * 0.3).contiguous()
    gamma2, beta2 = torch.rand(Cc, device=dev(), generator=g) + 0.5, rn(Cc)
    gamma1, beta1 = (torch.rand(Cc, device=dev(), generator=g) + 0.5, rn(Cc)) if double_bn else (None, None)
    mean, var = (ops.bn_batch_stats_images if images else ops.bn_batch_stats)(ops.Act(r0.clone()))
    fold = TO.bn_train_fold(mean, var, gamma2, beta2, H * W if images else B * H * W, gamma1=gamma1, beta1=beta1, images=B if images else 0)
    outs = {}
    for fused in (False, True):
        r = ops.Act(r0.clone())
        dg2, db2 = torch.zeros(Cc, device=dev()), torch.zeros(Cc, device=dev())
        dg1 = torch.zeros(Cc, device=dev()) if double_bn else None
        if fused:
            dy = TO.Cout1Grad(g1, w9)
        else:
            dy = TO.conv3x3_cout1_bwd_data(g1, w9, ops.Act.empty(B, H, W, Cc, dev()))
        TO.bn_backward(dy, r, fold, gamma2, dg2, db2, r, mask=TO.MASK_RELU6, gamma1=gamma1, dgamma1=dg1)
        torch.cuda.synchronize()
        outs[fused] = (r.buf.clone(), dg2, db2, dg1)
    a, b = outs[True], outs[False]
    assert not torch.isnan(a[0]).any()
    assert torch.equal(a[0], b[0])
    for u, v in zip(a[1:], b[1:]):
        if u is not None:
            assert (u - v).abs().max().item() < 1e-5 * max(v.abs().max().item(), 1e-3)
