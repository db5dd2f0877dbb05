"""GPU parity tests, op by op: every graph-D entry point of libemdenoise.so (called through the C ABI
via emdenoise.ops) against the oracle's TF-op restatement (oracle/tf_ops.py, float64) on the same
seeded inputs.  Tolerances are relative L2:
  split-bf16 matrix-core ops (EMD_PREC_BF16X3) ........ 2e-5   (north_star bar for the network: 1e-3)
  one-pass bf16 matrix-core ops (EMD_PREC_BF16) ....... 6e-3   (fast mode, ~2^-9 per product)
  fp32 VALU ops (depthwise, resize, ...) .............. 2e-6
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL_X3 = 2e-5
TOL_X1 = 6e-3
TOL_F32 = 2e-6


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def rnd(shape, seed, scale=1.0, positive=False):
    r = np.random.default_rng(seed)
    a = r.standard_normal(shape) * scale
    if positive:
        a = np.abs(a)
    return a.astype(np.float32)


def dev():
    return torch.device("cuda", 0)


def to_act(x_np, ld=None, c0=0):
    """Upload [B,H,W,C]; with ld, place it as channels [c0,c0+C) of a wider NaN-filled buffer."""
    from emdenoise import ops

    B, H, W, Cc = x_np.shape
    if ld is None:
        return ops.Act(torch.from_numpy(x_np).to(dev()))
    buf = torch.full((B, H, W, ld), float("nan"), dtype=torch.float32, device=dev())
    buf[..., c0:c0 + Cc] = torch.from_numpy(x_np).to(dev())
    return ops.Act(buf, Cc, c0)


def out_act(B, H, W, Cc, ld=None, c0=0):
    from emdenoise import ops

    buf = torch.full((B, H, W, ld or Cc), float("nan"), dtype=torch.float32, device=dev())
    return ops.Act(buf, Cc, c0)


def t64(a):
    return torch.from_numpy(np.asarray(a, np.float64))


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,W,ci,co,stride", [
    (2, 16, 16, 64, 64, 1),      # N <= 64 tile
    (1, 9, 13, 128, 256, 1),     # ragged M
    (2, 8, 8, 728, 728, 1),      # K, N not multiples of 32 / 128
    (1, 4, 4, 3640, 256, 1),     # ASPP reduce
    (2, 16, 16, 128, 128, 2),    # residual projection, stride 2
    (1, 7, 9, 256, 728, 2),      # stride 2 on odd sizes
    (1, 32, 32, 384, 128, 1),
])
@pytest.mark.parametrize("prec", [3, 1])
def test_conv1x1(B, H, W, ci, co, stride, prec):
    from emdenoise import ops
    from oracle import tf_ops as T

    x = rnd((B, H, W, ci), 1, positive=True)
    w = rnd((1, 1, ci, co), 2, scale=(2.0 / (ci + co)) ** 0.5)
    s1, t1 = rnd((co,), 3, 0.3) + 1.0, rnd((co,), 4, 0.5)
    ref = T.relu6_t(T.conv2d_t(t64(x), t64(w), None, stride=stride) * t64(s1) + t64(t1)).numpy()
    pw = ops.PackedWeights(w[0], False, dev())
    Ho, Wo = -(-H // stride), -(-W // stride)
    out = out_act(B, Ho, Wo, co)
    ops.conv1x1(to_act(x), pw, torch.from_numpy(s1).to(dev()), torch.from_numpy(t1).to(dev()), out, stride=stride,
                precision=prec)
    torch.cuda.synchronize()
    assert rel_l2(out.torch().cpu().numpy(), ref) < (TOL_X3 if prec == 3 else TOL_X1)


def test_conv1x1_epilogue_slices_and_residual():
    """Second affine+relu6, residual add, input and output as channel slices of wider buffers; bytes
    outside the output slice must stay untouched."""
    from emdenoise import ops
    from oracle import tf_ops as T

    B, H, W, ci, co = 2, 10, 12, 128, 256
    x = rnd((B, H, W, ci), 5, positive=True)
    w = rnd((1, 1, ci, co), 6, scale=0.08)
    s1, t1, s2, t2 = rnd((co,), 7, 0.2) + 1, rnd((co,), 8, 0.5), rnd((co,), 9, 0.2) + 1, rnd((co,), 10, 0.5)
    r = rnd((B, H, W, co), 11, positive=True)
    y = T.relu6_t(T.conv2d_t(t64(x), t64(w)) * t64(s1) + t64(t1))
    y = T.relu6_t(y * t64(s2) + t64(t2)) + t64(r)
    pw = ops.PackedWeights(w[0], False, dev())
    out = out_act(B, H, W, co, ld=co + 128, c0=64)
    d = lambda a: torch.from_numpy(a).to(dev())
    ops.conv1x1(to_act(x, ld=ci + 32, c0=16), pw, d(s1), d(t1), out, scale2=d(s2), shift2=d(t2),
                res=to_act(r, ld=co + 4, c0=4))
    torch.cuda.synchronize()
    assert rel_l2(out.torch().cpu().numpy(), y.numpy()) < TOL_X3
    full = out.buf.cpu().numpy()
    assert np.isnan(full[..., :64]).all() and np.isnan(full[..., 64 + co:]).all()


@pytest.mark.parametrize("B,H,W,ci,co", [(2, 5, 7, 64, 32), (1, 8, 8, 128, 128), (1, 3, 3, 256, 256), (1, 1, 1, 64, 64)])
@pytest.mark.parametrize("prec", [3, 1])
def test_deconv3x3s2(B, H, W, ci, co, prec):
    from emdenoise import ops
    from oracle import tf_ops as T

    x = rnd((B, H, W, ci), 12, positive=True)
    w = rnd((3, 3, co, ci), 13, scale=(2.0 / (9 * ci)) ** 0.5)
    bias = rnd((co,), 14, 0.2)
    s1, t1 = rnd((co,), 15, 0.2) + 1, rnd((co,), 16, 0.4)
    ref = T.relu6_t(T.conv2d_transpose_s2_t(t64(x), t64(w), t64(bias)) * t64(s1) + t64(t1)).numpy()
    packs = ops.pack_deconv(w, dev())
    shift = (bias.astype(np.float64) * s1 + t1).astype(np.float32)  # bias folded into the shift
    out = out_act(B, 2 * H, 2 * W, co)
    ops.deconv3x3s2(to_act(x), packs, torch.from_numpy(s1).to(dev()), torch.from_numpy(shift).to(dev()), out,
                    precision=prec)
    torch.cuda.synchronize()
    got = out.torch().cpu().numpy()
    assert not np.isnan(got).any()
    assert rel_l2(got, ref) < (TOL_X3 if prec == 3 else TOL_X1)


@pytest.mark.parametrize("B,H,W,Cc,stride,rate", [
    (2, 16, 16, 64, 1, 1), (1, 13, 9, 128, 1, 1), (1, 8, 8, 728, 1, 1), (2, 16, 16, 64, 2, 1), (1, 9, 7, 256, 2, 1),
    (1, 32, 32, 128, 1, 6), (1, 32, 32, 64, 1, 12), (1, 32, 32, 64, 1, 18), (1, 3, 3, 384, 1, 1),
])
def test_dw3x3(B, H, W, Cc, stride, rate):
    from emdenoise import ops
    from oracle import tf_ops as T

    x = rnd((B, H, W, Cc), 17)
    w = rnd((3, 3, Cc, 1), 18, 0.4)
    ref = T.depthwise_conv2d_t(t64(x), t64(w), stride, rate).numpy()
    Ho, Wo = -(-H // stride), -(-W // stride)
    out = out_act(B, Ho, Wo, Cc)
    ops.dw3x3(to_act(x, ld=Cc + 8, c0=4), torch.from_numpy(np.ascontiguousarray(w[..., 0])).to(dev()), out,
              stride=stride, rate=rate)
    torch.cuda.synchronize()
    assert rel_l2(out.torch().cpu().numpy(), ref) < TOL_F32


def test_cin1_both_forms():
    from emdenoise import ops
    from oracle import tf_ops as T

    B, H, W = 2, 17, 24
    x = rnd((B, H, W, 1), 19, positive=True)
    # cnn0: depthwise 3x3 on one channel, then 1 -> 64 pointwise, affine, relu6
    dw = rnd((3, 3, 1, 1), 20, 0.5)
    pw = rnd((1, 1, 1, 64), 21, 0.7)
    s, t = rnd((64,), 22, 0.2) + 1, rnd((64,), 23, 0.3)
    ref = T.relu6_t(T.conv2d_t(T.depthwise_conv2d_t(t64(x), t64(dw)), t64(pw)) * t64(s) + t64(t)).numpy()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())
    out = out_act(B, H, W, 64)
    ops.cin1(d(x), d(dw.reshape(9)), d(pw.reshape(64) * s), d(t), out)
    torch.cuda.synchronize()
    assert rel_l2(out.torch().cpu().numpy(), ref) < TOL_F32
    # residual0: 1x1 stride-2 conv 1 -> 128 with bias
    w = rnd((1, 1, 1, 128), 24, 0.7)
    bias = rnd((128,), 25, 0.2)
    s, t = rnd((128,), 26, 0.2) + 1, rnd((128,), 27, 0.3)
    ref = T.relu6_t(T.conv2d_t(t64(x), t64(w), t64(bias), stride=2) * t64(s) + t64(t)).numpy()
    Ho, Wo = -(-H // 2), -(-W // 2)
    out = out_act(B, Ho, Wo, 128, ld=384, c0=256)
    ops.cin1(d(x), None, d(w.reshape(128) * s), d(bias * s + t), out, stride=2)
    torch.cuda.synchronize()
    assert rel_l2(out.torch().cpu().numpy(), ref) < TOL_F32


@pytest.mark.parametrize("B,H,W,ci", [(2, 12, 16, 64), (1, 5, 7, 128), (1, 9, 9, 16), (1, 13, 8, 64), (2, 3, 64, 256)])
def test_conv3x3_cout1(B, H, W, ci):
    from emdenoise import ops
    from oracle import tf_ops as T

    x = rnd((B, H, W, ci), 28, positive=True)
    w = rnd((3, 3, ci, 1), 29, 0.1)
    bias, s, t = 0.13, 1.7, -0.2
    ref = T.relu6_t((T.conv2d_t(t64(x), t64(w)) + bias) * s + t).numpy()
    out = torch.full((B, H, W, 1), float("nan"), dtype=torch.float32, device=dev())
    ops.conv3x3_cout1(to_act(x), torch.from_numpy(np.ascontiguousarray(w[..., 0])).to(dev()), s, bias * s + t, out)
    torch.cuda.synchronize()
    assert rel_l2(out.cpu().numpy(), ref) < TOL_F32


@pytest.mark.parametrize("Hi,Wi,Ho,Wo,Cc", [(4, 4, 16, 16, 256), (8, 8, 8, 8, 728), (3, 5, 12, 20, 64), (32, 32, 128, 128, 8),
                                            (16, 12, 32, 24, 64), (1, 1, 2, 2, 8), (5, 7, 10, 14, 728)])   # exact 2x: the block form
def test_resize_bilinear(Hi, Wi, Ho, Wo, Cc):
    from emdenoise import ops
    from oracle import tf_ops as T

    x = rnd((2, Hi, Wi, Cc), 30)
    ref = T.resize_bilinear_legacy_t(t64(x), Ho, Wo).numpy()
    out = out_act(2, Ho, Wo, Cc, ld=Cc + 128, c0=0)
    ops.resize_bilinear(to_act(x), out)
    torch.cuda.synchronize()
    assert rel_l2(out.torch().cpu().numpy(), ref) < TOL_F32


def test_affine_relu6():
    from emdenoise import ops

    x = rnd((2, 6, 5, 728), 31, 3.0)
    s, t = rnd((728,), 32, 0.3) + 1, rnd((728,), 33, 1.0)
    ref = np.clip(x.astype(np.float64) * s + t, 0, 6)
    out = out_act(2, 6, 5, 728, ld=3640, c0=2912)
    ops.affine_relu6(to_act(x), torch.from_numpy(s).to(dev()), torch.from_numpy(t).to(dev()), out)
    torch.cuda.synchronize()
    assert rel_l2(out.torch().cpu().numpy(), ref) < TOL_F32


def test_error_paths_report_through_last_error():
    from emdenoise import _lib, ops

    x = to_act(rnd((1, 4, 4, 6), 34))  # C = 6: not a multiple of 4
    out = out_act(1, 4, 4, 6)
    with pytest.raises(_lib.EmdError, match="multiples of 4"):
        ops.dw3x3(x, torch.zeros(9 * 6, device=dev()), out)


@pytest.mark.parametrize("B,H,W,ci,co,res,extra", [
    (2, 16, 32, 64, 64, False, False),      # BN = 64 tile
    (1, 8, 16, 128, 128, True, False),      # single tile: every patch border is zero padding
    (2, 24, 48, 384, 128, False, True),     # concat-slice input, second affine
    (1, 32, 32, 32, 8, True, True),
])
@pytest.mark.parametrize("prec", [3, 1])
def test_sep_fused(B, H, W, ci, co, res, extra, prec):
    """emd_sep3x3_fused_f32 == depthwise 3x3 (SAME) -> pointwise -> affine -> relu6 [-> affine -> relu6] [+ res]."""
    from emdenoise import ops
    from oracle import tf_ops as T

    x = rnd((B, H, W, ci), 40, positive=True)
    dw = rnd((3, 3, ci, 1), 41, 0.35)
    pw = rnd((1, 1, ci, co), 42, scale=(2.0 / (ci + co)) ** 0.5)
    s1, t1 = rnd((co,), 43, 0.2) + 1, rnd((co,), 44, 0.5)
    s2, t2 = rnd((co,), 45, 0.2) + 1, rnd((co,), 46, 0.5)
    r = rnd((B, H, W, co), 47, positive=True)
    y = T.relu6_t(T.conv2d_t(T.depthwise_conv2d_t(t64(x), t64(dw)), t64(pw)) * t64(s1) + t64(t1))
    if extra:
        y = T.relu6_t(y * t64(s2) + t64(t2))
    if res:
        y = y + t64(r)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())
    xa = to_act(x, ld=ci + 64, c0=32)
    assert ops.sep_fused_supported(xa, co, 1, 1)
    out = out_act(B, H, W, co, ld=co + 8, c0=4)
    ops.sep_fused(xa, d(dw[..., 0]), ops.PackedWeights(pw[0], False, dev()), d(s1), d(t1), out,
                  scale2=d(s2) if extra else None, shift2=d(t2) if extra else None,
                  res=to_act(r, ld=co + 12, c0=8) if res else None, precision=prec)
    torch.cuda.synchronize()
    got = out.torch().cpu().numpy()
    assert not np.isnan(got).any()
    assert rel_l2(got, y.numpy()) < (TOL_X3 if prec == 3 else TOL_X1)
    full = out.buf.cpu().numpy()
    assert np.isnan(full[..., :4]).all() and np.isnan(full[..., 4 + co:]).all()   # nothing written outside the slice


@pytest.mark.parametrize("B,H,W,ci,co,res,extra,split", [
    (2, 16, 32, 128, 256, False, False, False),   # cnn2
    (1, 24, 16, 256, 256, True, False, True),     # deconv2_b: residual, split32 output for the transposed conv
    (2, 8, 48, 64, 160, True, True, False),       # N tail inside the 256-column tile, second affine
])
def test_sep_fused_wide(B, H, W, ci, co, res, extra, split):
    """128 < Cout <= 256 (Cin <= 256): the one-tile-of-256-columns form on 4 x 16 pixel tiles against the oracle and, for the split32
    output, against emd_to_split32_f32 of its own fp32 output."""
    from emdenoise import ops
    from oracle import tf_ops as T

    x = rnd((B, H, W, ci), 240, positive=True)
    dw = rnd((3, 3, ci, 1), 241, 0.35)
    pw = rnd((1, 1, ci, co), 242, scale=(2.0 / (ci + co)) ** 0.5)
    s1, t1 = rnd((co,), 243, 0.2) + 1, rnd((co,), 244, 0.5)
    s2, t2 = rnd((co,), 245, 0.2) + 1, rnd((co,), 246, 0.5)
    r = rnd((B, H, W, co), 247, positive=True)
    y = T.relu6_t(T.conv2d_t(T.depthwise_conv2d_t(t64(x), t64(dw)), t64(pw)) * t64(s1) + t64(t1))
    if extra:
        y = T.relu6_t(y * t64(s2) + t64(t2))
    if res:
        y = y + t64(r)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())
    xa = to_act(x, ld=ci + 64, c0=32)
    assert ops.sep_fused_supported(xa, co, 1, 1)
    kw = dict(scale2=d(s2) if extra else None, shift2=d(t2) if extra else None, res=to_act(r, ld=co + 12, c0=8) if res else None)
    pk = ops.PackedWeights(pw[0], False, dev())
    out = out_act(B, H, W, co, ld=co + 8, c0=4)
    ops.sep_fused(xa, d(dw[..., 0]), pk, d(s1), d(t1), out, **kw)
    torch.cuda.synchronize()
    got = out.torch().cpu().numpy()
    assert rel_l2(got, y.numpy()) < TOL_X3
    full = out.buf.cpu().numpy()
    assert np.isnan(full[..., :4]).all() and np.isnan(full[..., 4 + co:]).all()
    if split:
        sp = ops.SplitAct(B, H, W, co, dev())
        sp.buf.fill_(float("nan"))
        ops.sep_fused(xa, d(dw[..., 0]), pk, d(s1), d(t1), sp, **kw)
        want = ops.to_split32(ops.Act(out.torch().contiguous()))
        torch.cuda.synchronize()
        assert torch.equal(sp.buf.view(torch.int32), want.buf.view(torch.int32))


@pytest.mark.parametrize("B,H,W,ci,co,gen_act,reflect,extra", [
    (2, 32, 48, 64, 64, 1, False, False),     # graph D's cnn0 -> cnn0_last
    (1, 64, 64, 64, 64, 1, False, True),      # several tiles per workgroup
    (2, 16, 32, 32, 128, 4, True, False),     # leaky relu, reflect border
    (1, 8, 16, 128, 24, 0, False, False),
])
def test_sep_fused_generated_input(B, H, W, ci, co, gen_act, reflect, extra):
    """emd_sep3x3_fused_gen_f32 == emd_cin1_f32 (written out) followed by emd_sep3x3_fused[_reflect]_f32, bit for bit."""
    from emdenoise import ops

    img = rnd((B, H, W, 1), 140)
    w9, a, t = rnd((9,), 141, 0.4), rnd((ci,), 142, 0.8), rnd((ci,), 143, 0.5) + 0.5
    dw = rnd((3, 3, ci), 144, 0.35)
    pw = ops.PackedWeights(rnd((1, ci, co), 145, scale=(2.0 / (ci + co)) ** 0.5), False, dev())
    d = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(dev())
    s1, t1, s2, t2 = d(rnd((co,), 146, 0.2) + 1), d(rnd((co,), 147, 0.5)), d(rnd((co,), 148, 0.2) + 1), d(rnd((co,), 149, 0.5))
    x = d(img)
    # the written-out route; cin1 only knows relu6 / none, so the other activations are applied by torch on its raw output
    full = ops.cin1(x, d(w9), d(a), d(t), out_act(B, H, W, ci, ld=ci, c0=0), act=(gen_act == 1))
    if gen_act == 4:
        v = full.buf
        full.buf.copy_(torch.minimum(torch.maximum(v, 0.2 * v), torch.full_like(v, float("inf"))))
    want = ops.sep_fused(full, d(dw), pw, s1, t1, out_act(B, H, W, co, ld=co, c0=0), scale2=s2 if extra else None,
                         shift2=t2 if extra else None, reflect=reflect)
    d4 = ops.cin1(x, d(w9), d(np.array([1, 0, 0, 0])), d(np.zeros(4)), out_act(B, H, W, 4, ld=4, c0=0), act=False)
    got = ops.sep_fused_gen(d4, d(a), d(t), d(dw), pw, s1, t1, out_act(B, H, W, co, ld=co, c0=0), gen_act=gen_act,
                            scale2=s2 if extra else None, shift2=t2 if extra else None, reflect=reflect)
    torch.cuda.synchronize()
    assert torch.equal(got.torch(), want.torch())
    assert not torch.isnan(got.torch()).any()


@pytest.mark.parametrize("B,H,W,ci,co,gen_act,reflect,extra,tpw", [
    (2, 32, 48, 64, 64, 1, False, False, 0),      # graph D's cnn0 -> cnn0_last; three tile columns: left edge, interior, right edge
    (1, 16, 16, 64, 64, 1, False, False, 0),      # one tile column: both image edges in the same patch
    (2, 24, 64, 64, 64, 1, False, True, 2),       # two tiles per workgroup (d of the second tile requested a tile ahead), second affine
    (1, 16, 128, 64, 40, 2, False, False, 4),     # four, then a workgroup that starts in the interior; relu; a channel tail
    (1, 16, 128, 64, 64, 4, True, False, 8),      # one workgroup per tile row, reflect border, leaky relu
    (2, 8, 32, 32, 24, 0, True, False, 2),        # one chunk per tile: a tile change at every step
])
def test_sep_generated_input_on_the_pipelined_kernel(B, H, W, ci, co, gen_act, reflect, extra, tpw):
    """The generated-input fused separable conv on sep_pipe.hip's 4-wave instance (round 4; dev knob sep_gen_pipe = 1) ==
    sep_fused.hip's register-staged kernel (sep_gen_pipe = 0, the default: the test above holds it to the written-out route), bit for bit."""
    from emdenoise import _lib, ops

    img = rnd((B, H, W, 1), 150)
    w9, a, t = rnd((9,), 151, 0.4), rnd((ci,), 152, 0.8), rnd((ci,), 153, 0.5) + 0.5
    dw = rnd((3, 3, ci), 154, 0.35)
    pw = ops.PackedWeights(rnd((1, ci, co), 155, scale=(2.0 / (ci + co)) ** 0.5), False, dev())
    d = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(dev())
    s1, t1, s2, t2 = d(rnd((co,), 156, 0.2) + 1), d(rnd((co,), 157, 0.5)), d(rnd((co,), 158, 0.2) + 1), d(rnd((co,), 159, 0.5))
    d4 = ops.cin1(d(img), d(w9), d(np.array([1, 0, 0, 0])), d(np.zeros(4)), out_act(B, H, W, 4, ld=4, c0=0), act=False)
    outs = {}
    try:
        for k in (1, 0):
            _lib.knob("sep_gen_pipe", k)
            _lib.knob("sep_tpw", tpw if k else 0)
            o = out_act(B, H, W, co, ld=co + 4, c0=0)
            o.buf.fill_(float("nan"))
            outs[k] = ops.sep_fused_gen(d4, d(a), d(t), d(dw), pw, s1, t1, o, gen_act=gen_act, scale2=s2 if extra else None,
                                        shift2=t2 if extra else None, reflect=reflect)
    finally:
        _lib.knob("sep_gen_pipe", 0)
        _lib.knob("sep_tpw", 0)
    torch.cuda.synchronize()
    assert not torch.isnan(outs[1].torch()).any()
    assert torch.equal(outs[1].torch(), outs[0].torch())
    assert torch.isnan(outs[1].buf.view(B, H, W, co + 4)[..., co:]).all()      # nothing written past the layer's channels


@pytest.mark.parametrize("B,H,W,ci,co,co2", [
    (2, 16, 32, 128, 64, 64),       # deconv0_a + residual0_d: 64 | 64 columns on 8 x 16 tiles
    (1, 8, 16, 384, 128, 128),      # deconv1_a + residual1_d: 128 | 128 columns on 4 x 16 tiles, concat-slice input
    (2, 12, 48, 96, 128, 32),       # H % 4 == 0 only (wide form), unequal widths
    (1, 24, 16, 64, 36, 64),        # a channel tail in the separable output
    (2, 64, 64, 32, 8, 4),          # several tiles per workgroup, one K chunk
])
def test_sep_dual(B, H, W, ci, co, co2):
    """emd_sep3x3_dual_f32 (denoiser.py:356-359 / :368-371 / :380-383 in one launch): output 1 == the separable conv block,
    output 2 == slim.conv2d(kernel 1) + bias + BN + relu6 of the same input, both against oracle/tf_ops.py (float64); and each
    against the kernel it replaces (emd_sep3x3_fused_f32 / emd_conv1x1_f32)."""
    from emdenoise import ops
    from oracle import tf_ops as T

    x = rnd((B, H, W, ci), 240, positive=True)
    dw = rnd((3, 3, ci, 1), 241, 0.35)
    pw = rnd((1, 1, ci, co), 242, scale=(2.0 / (ci + co)) ** 0.5)
    w2 = rnd((1, 1, ci, co2), 243, scale=(2.0 / (ci + co2)) ** 0.5)
    bias2 = rnd((co2,), 244, 0.2)
    s1, t1 = rnd((co,), 245, 0.2) + 1, rnd((co,), 246, 0.5)
    sb, tb = rnd((co2,), 247, 0.2) + 1, rnd((co2,), 248, 0.5)
    y1 = T.relu6_t(T.conv2d_t(T.depthwise_conv2d_t(t64(x), t64(dw)), t64(pw)) * t64(s1) + t64(t1)).numpy()
    y2 = T.relu6_t(T.conv2d_t(t64(x), t64(w2), t64(bias2)) * t64(sb) + t64(tb)).numpy()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())
    xa = to_act(x, ld=ci + 64, c0=32)
    assert ops.sep_dual_supported(xa, co, co2)
    out, out2 = out_act(B, H, W, co, ld=co + 8, c0=4), out_act(B, H, W, co2, ld=co2 + 12, c0=8)
    p1, p2 = ops.PackedWeights(pw[0], False, dev()), ops.PackedWeights(w2[0], False, dev())
    shift_b = (bias2.astype(np.float64) * sb + tb).astype(np.float32)
    ops.sep_dual(xa, d(dw[..., 0]), p1, p2, d(s1), d(t1), out, d(sb), d(shift_b), out2)
    torch.cuda.synchronize()
    g1, g2 = out.torch().cpu().numpy(), out2.torch().cpu().numpy()
    assert not np.isnan(g1).any() and not np.isnan(g2).any()
    assert rel_l2(g1, y1) < TOL_X3 and rel_l2(g2, y2) < TOL_X3
    for o, c in ((out, co), (out2, co2)):   # nothing written outside the slices
        full = o.buf.cpu().numpy()
        assert np.isnan(full[..., :o.c0]).all() and np.isnan(full[..., o.c0 + c:]).all()
    # the kernels it replaces: same products, same order along K
    want1 = ops.sep_fused(xa, d(dw[..., 0]), p1, d(s1), d(t1), out_act(B, H, W, co)) if ops.sep_fused_supported(xa, co, 1, 1) else None
    want2 = ops.conv1x1(xa, p2, d(sb), d(shift_b), out_act(B, H, W, co2))
    torch.cuda.synchronize()
    if want1 is not None:
        assert rel_l2(g1, want1.torch().cpu().numpy()) < 1e-6
    assert rel_l2(g2, want2.torch().cpu().numpy()) < 1e-6


@pytest.mark.parametrize("B,H,W,ci,co,res,extra", [
    (2, 32, 32, 728, 728, True, False),     # the middle flow's third block conv (residual add), graph D at 512 px
    (1, 8, 32, 256, 728, False, False),     # cnn3-like: K = 8 steps
    (1, 4, 64, 728, 728, True, True),       # two tiles side by side (the patch's left / right columns are real pixels), extra BN
    (1, 12, 32, 260, 388, False, False),    # channel tail (260 = 8 x 32 + 4) and an N half with 4 real columns
])
def test_sep_gemm(B, H, W, ci, co, res, extra):
    """emd_sep3x3_gemm_f32 (depthwise stage inside the pointwise GEMM) against oracle/tf_ops.py, float64: depthwise 3x3 (SAME) ->
    pointwise -> affine -> relu6 [-> affine -> relu6] [+ res]; and against the two-kernel route it replaces."""
    from emdenoise import ops
    from oracle import tf_ops as T

    x = rnd((B, H, W, ci), 340, positive=True)
    dw = rnd((3, 3, ci, 1), 341, 0.35)
    pw = rnd((1, 1, ci, co), 342, scale=(2.0 / (ci + co)) ** 0.5)
    s1, t1 = rnd((co,), 343, 0.2) + 1, rnd((co,), 344, 0.5)
    s2, t2 = rnd((co,), 345, 0.2) + 1, rnd((co,), 346, 0.5)
    r = rnd((B, H, W, co), 347, positive=True)
    y = T.relu6_t(T.conv2d_t(T.depthwise_conv2d_t(t64(x), t64(dw)), t64(pw)) * t64(s1) + t64(t1))
    if extra:
        y = T.relu6_t(y * t64(s2) + t64(t2))
    if res:
        y = y + t64(r)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())
    xa = to_act(x, ld=ci + 64, c0=32)
    assert ops.sep_gemm_supported(xa, co)
    out = out_act(B, H, W, co, ld=co + 8, c0=4)
    pk = ops.PackedWeights(pw[0], False, dev())
    kw = dict(scale2=d(s2) if extra else None, shift2=d(t2) if extra else None, res=to_act(r, ld=co + 12, c0=8) if res else None)
    ops.sep_gemm(xa, d(dw[..., 0]), pk, d(s1), d(t1), out, **kw)
    torch.cuda.synchronize()
    got = out.torch().cpu().numpy()
    assert not np.isnan(got).any()
    assert rel_l2(got, y.numpy()) < TOL_X3
    full = out.buf.cpu().numpy()
    assert np.isnan(full[..., :4]).all() and np.isnan(full[..., 4 + co:]).all()   # nothing written outside the slice
    if ops.conv1x1_split32_supported(B * H * W, ci, co):
        want = ops.sep_split32(xa, d(dw[..., 0]), pk, d(s1), d(t1), out_act(B, H, W, co), **kw)
        torch.cuda.synchronize()
        assert rel_l2(got, want.torch().cpu().numpy()) < 2e-6


def test_sep_fused_falls_back_cleanly():
    from emdenoise import _lib, ops

    x = to_act(rnd((1, 12, 16, 64), 48))       # H % 8 != 0
    assert not ops.sep_fused_supported(x, 64, 1, 1)
    assert not ops.sep_fused_supported(to_act(rnd((1, 8, 16, 64), 49)), 260, 1, 1)   # more than one 256-column N tile
    assert not ops.sep_fused_supported(to_act(rnd((1, 8, 16, 384), 49)), 256, 1, 1)  # the 256-column form stops at Cin = 256
    with pytest.raises(_lib.EmdError, match="emd_dw3x3_f32"):
        ops.sep_fused(x, torch.zeros(9 * 64, device=dev()), ops.PackedWeights(rnd((1, 64, 64), 50), False, dev()),
                      torch.ones(64, device=dev()), torch.zeros(64, device=dev()), out_act(1, 12, 16, 64))


@pytest.mark.parametrize("B,H,W,ci,co,stride,rate", [
    (1, 12, 12, 64, 64, 1, 1), (2, 32, 32, 128, 256, 1, 6), (1, 32, 32, 728, 728, 1, 18), (1, 9, 11, 32, 64, 2, 1),
    (1, 16, 16, 64, 128, 2, 1), (1, 8, 8, 256, 64, 1, 12),
])
def test_conv3x3_dense(B, H, W, ci, co, stride, rate):
    """9-tap implicit GEMM: tf.layers.conv2d(kernel_size=3, dilation_rate, 'same') + bias + affine + relu6."""
    from emdenoise import ops
    from oracle import tf_ops as T

    x = rnd((B, H, W, ci), 60, positive=True)
    w = rnd((3, 3, ci, co), 61, scale=(2.0 / (9 * ci)) ** 0.5)
    bias = rnd((co,), 62, 0.2)
    s1, t1 = rnd((co,), 63, 0.2) + 1, rnd((co,), 64, 0.4)
    ref = T.relu6_t(T.conv2d_t(t64(x), t64(w), t64(bias), stride=stride, rate=rate) * t64(s1) + t64(t1)).numpy()
    pw = ops.PackedWeights(w.reshape(9, ci, co), False, dev())
    Ho, Wo = -(-H // stride), -(-W // stride)
    out = out_act(B, Ho, Wo, co, ld=co + 4, c0=4)
    shift = (bias.astype(np.float64) * s1 + t1).astype(np.float32)
    ops.conv3x3(to_act(x), pw, torch.from_numpy(s1).to(dev()), torch.from_numpy(shift).to(dev()), out, stride=stride,
                rate=rate)
    torch.cuda.synchronize()
    assert rel_l2(out.torch().cpu().numpy(), ref) < TOL_X3


@pytest.mark.parametrize("H,W,Cc", [(32, 32, 728), (5, 7, 64), (2, 2, 8)])
def test_avgpool2x2(H, W, Cc):
    from emdenoise import ops
    from oracle import tf_ops as T

    x = rnd((2, H, W, Cc), 65)
    ref = T.avg_pool2x2_same_t(t64(x)).numpy()
    out = out_act(2, -(-H // 2), -(-W // 2), Cc)
    ops.avgpool2x2(to_act(x, ld=Cc + 4, c0=0), out)
    torch.cuda.synchronize()
    assert rel_l2(out.torch().cpu().numpy(), ref) < TOL_F32


def test_generated_input_entry_points_validate_and_accept_empty_batches():
    """emd_sep3x3_fused_gen_f32 / emd_dw3x3_reflect_gen_f32: B = 0 is a no-op; misaligned (a, t) vectors, reflect padding on a
    one-row image and an unsupported shape come back as negative status codes with a message (nothing is launched)."""
    import ctypes as C

    from emdenoise import _lib, ops

    lib = _lib.load()
    d4 = out_act(1, 8, 16, 4, ld=4, c0=0)
    a = torch.zeros(68, device=dev())
    pw = ops.PackedWeights(rnd((1, 64, 64), 150), False, dev())
    dw, s1 = torch.zeros(9 * 64, device=dev()), torch.ones(64, device=dev())
    y = out_act(1, 8, 16, 64, ld=64, c0=0)
    p = lambda t: C.c_void_p(t.data_ptr())
    args = lambda ga, B=1, H=8: (d4.ptr, 4, ga, p(a), 1, p(dw), p(pw.hi), p(pw.lo), p(s1), p(s1), None, None, None, 0, y.ptr, 64,
                                B, H, 16, 64, 64, 1, ops.PREC_BF16X3, 0, None)
    assert lib.emd_sep3x3_fused_gen_f32(*args(p(a), B=0)) == 0                                   # empty batch
    assert lib.emd_sep3x3_fused_gen_f32(*args(C.c_void_p(a.data_ptr() + 4))) < 0                 # gen_a not 16-byte aligned
    assert b"gen_a" in lib.emd_last_error()
    assert lib.emd_sep3x3_fused_gen_f32(*args(p(a), H=12)) < 0                                   # H % 8 != 0: not a fused shape
    y2 = out_act(1, 4, 8, 64, ld=64, c0=0)
    g = lambda B, H, ga: lib.emd_dw3x3_reflect_gen_f32(d4.ptr, 4, ga, p(a), 1, p(dw), y2.ptr, 64, B, H, 16, 64, 2, None)
    assert g(0, 8, p(a)) == 0
    assert g(1, 1, p(a)) < 0 and b"reflect" in lib.emd_last_error()
    assert g(1, 8, C.c_void_p(a.data_ptr() + 4)) < 0
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,H,W,co,stride", [(2, 64, 64, 32, 2), (1, 33, 47, 32, 2), (2, 24, 40, 64, 1), (1, 16, 16, 40, 2)])
@pytest.mark.parametrize("split", [False, True])
def test_conv3x3_cin1(B, H, W, co, stride, split):
    """emd_conv3x3_cin1_f32: graph X's entry conv (tf.layers.conv2d(1 -> 32, k 3, stride 2) + bias -> BN -> relu,
    misc_py/modified_Xception.py:356-364) in fp32 FMAs, against the oracle's TF-SAME conv (float64); fp32 output into a NaN-filled concat
    slice, split32 output == emd_to_split32_f32 of the fp32 one (padding channels zero)."""
    from emdenoise import ops
    from oracle import tf_ops as T

    x = rnd((B, H, W, 1), 501)
    w = rnd((3, 3, 1, co), 502, 0.4)
    bias, g, h = rnd((co,), 503, 0.2), rnd((co,), 504, 0.3) + 1.0, rnd((co,), 505, 0.4)
    ref = torch.relu((T.conv2d_t(t64(x), t64(w), t64(bias), stride=stride)) * t64(g) + t64(h)).numpy()
    Ho, Wo = -(-H // stride), -(-W // stride)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())
    xd, wd = d(x), d(w[:, :, 0, :].reshape(9, co))
    gs, hs = d(g), d(bias.astype(np.float64) * g + h)
    out = out_act(B, Ho, Wo, co, ld=co + 8, c0=4)
    ops.conv3x3_cin1(xd, wd, gs, hs, out, stride=stride, act=ops.ACT_RELU)
    torch.cuda.synchronize()
    got = out.torch().cpu().numpy()
    assert rel_l2(got, ref) < TOL_F32
    full = out.buf.cpu().numpy()
    assert np.isnan(full[..., :4]).all() and np.isnan(full[..., 4 + co:]).all()
    if split:
        sp = ops.SplitAct(B, Ho, Wo, co, dev())
        sp.buf.fill_(float("nan"))
        ops.conv3x3_cin1(xd, wd, gs, hs, sp, stride=stride, act=ops.ACT_RELU)
        want = ops.to_split32(ops.Act(out.torch().contiguous()))
        torch.cuda.synchronize()
        assert torch.equal(sp.buf.view(torch.int32), want.buf.view(torch.int32))
