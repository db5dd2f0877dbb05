"""GPU parity tests of the split32 path (csrc/gemm_split.hip): the depthwise kernels writing pre-split bf16 hi/lo
activations, the converter, and the LDS-DMA pointwise GEMM that consumes them.  Checked three ways:
  * against the oracle's TF-op restatement (oracle/tf_ops.py, float64): 2e-5 relative L2 (split-bf16 bar);
  * bit for bit against emd_conv1x1_f32 (the register-staged kernel computes the same products in the same order);
  * the split32 tensor itself: hi + lo reproduces the fp32 value to 2^-16 relative, padding channels are zero.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL_X3 = 2e-5
DECONV_DEFAULT = 3   # dev knob deconv_direct (csrc/emd_common.hpp)


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


def t64(a):
    return torch.from_numpy(np.asarray(a, np.float64))


def up(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


@pytest.mark.parametrize("C", [32, 40, 728, 100])
def test_to_split32_layout(C):
    from emdenoise import ops

    x = rnd((2, 5, 7, C), 11, 3.0)
    xa = ops.Act(up(x))
    sp = ops.to_split32(xa)
    torch.cuda.synchronize()
    assert sp.ld == -(-C // 32) * 32
    back = sp.to_float().cpu().numpy()
    assert np.max(np.abs(back - x) / np.maximum(np.abs(x), 1e-30)) < 2.0 ** -15.9
    raw = sp.buf.view(torch.bfloat16).view(2, 5, 7, sp.ld // 32, 2, 32).float().cpu().numpy()
    hi = raw[..., 0, :].reshape(2, 5, 7, sp.ld)
    lo = raw[..., 1, :].reshape(2, 5, 7, sp.ld)
    # hi is the round-to-nearest bf16 of x; the padding channels are exactly zero in both planes
    ref_hi = torch.from_numpy(x).to(torch.bfloat16).float().numpy()
    assert np.array_equal(hi[..., :C], ref_hi)
    assert not hi[..., C:].any() and not lo[..., C:].any()


@pytest.mark.parametrize("B,H,W,C,stride,rate", [(2, 16, 16, 728, 1, 1), (1, 64, 64, 40, 1, 1), (2, 17, 13, 64, 2, 1),
                                                  (1, 32, 32, 96, 1, 6), (1, 70, 33, 256, 1, 1)])
def test_dw3x3_split32_equals_dw3x3_then_split(B, H, W, C, stride, rate):
    from emdenoise import ops

    x = rnd((B, H, W, C), 21)
    w = rnd((9, C), 22, 0.3)
    xa, wd = ops.Act(up(x)), up(w)
    Ho, Wo = -(-H // stride), -(-W // stride)
    plain = ops.dw3x3(xa, wd, ops.Act.empty(B, Ho, Wo, C, dev()), stride=stride, rate=rate)
    want = ops.to_split32(plain)
    got = ops.SplitAct(B, Ho, Wo, C, dev())
    got.buf.fill_(float("nan"))
    ops.dw3x3_split32(xa, wd, got, stride=stride, rate=rate)
    torch.cuda.synchronize()
    assert torch.equal(got.buf.view(torch.int32), want.buf.view(torch.int32))


@pytest.mark.parametrize("B,H,W,C,stride", [(1, 32, 32, 768, 1), (2, 17, 13, 40, 2)])
def test_dw3x3_reflect_split32_equals_reflect_then_split(B, H, W, C, stride):
    from emdenoise import ops

    x = rnd((B, H, W, C), 31)
    w = rnd((9, C), 32, 0.3)
    xa, wd = ops.Act(up(x)), up(w)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    want = ops.to_split32(ops.dw3x3_reflect(xa, wd, ops.Act.empty(B, Ho, Wo, C, dev()), stride=stride))
    got = ops.SplitAct(B, Ho, Wo, C, dev())
    got.buf.fill_(float("nan"))
    ops.dw3x3_reflect_split32(xa, wd, got, stride=stride)
    torch.cuda.synchronize()
    assert torch.equal(got.buf.view(torch.int32), want.buf.view(torch.int32))


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7])
def test_conv1x1_split32_kernel_variants_agree(variant):
    """Every tile / stage / loop variant of the GEMM on 32x32x16 MFMAs (dev knob) gives variant 3's bits (4 = the persistent form);
    variant 5 -- the default since round 3 -- runs on 16x16x32 MFMAs, whose K-step sum is ordered differently: same error class
    (2e-7), not the same bits."""
    from emdenoise import _lib, ops

    lib = _lib.load()
    x = rnd((4, 32, 32, 512), 41, positive=True)          # M = 4096 = 16 tiles of 256 rows
    w = rnd((1, 512, 384), 42, scale=0.05)
    r = rnd((4, 32, 32, 384), 43)
    pw = ops.PackedWeights(w, False, dev())
    s1, t1 = up(rnd((384,), 44, 0.3) + 1.0), up(rnd((384,), 45, 0.5))
    xs = ops.to_split32(ops.Act(up(x)))
    lib.emd_debug_split_variant(3)
    ref = ops.conv1x1_split32(xs, pw, s1, t1, ops.Act.empty(4, 32, 32, 384, dev()), res=ops.Act(up(r)))
    try:
        lib.emd_debug_split_variant(variant)
        got = ops.conv1x1_split32(xs, pw, s1, t1, ops.Act.empty(4, 32, 32, 384, dev()), res=ops.Act(up(r)))
        torch.cuda.synchronize()
    finally:
        lib.emd_debug_split_variant(-1)
    if variant == 5:
        assert float((got.buf - ref.buf).norm() / ref.buf.norm()) < 1e-6
    else:
        assert torch.equal(got.buf, ref.buf)


@pytest.mark.parametrize("B,H,W,ci,co,res,extra", [
    (2, 32, 32, 728, 728, True, False),    # middle flow (M tail: 2048 rows = 8 tiles exactly)
    (1, 24, 20, 728, 728, False, True),    # M = 480: one full tile + a ragged one; ASPP's extra BN
    (1, 16, 16, 256, 728, False, False),   # cnn3
    (1, 40, 40, 384, 256, True, False),    # deconv2_a
    (1, 8, 8, 160, 132, False, False),     # N tail inside the second N tile, K = 5 steps
    (1, 8, 8, 48, 128, False, False),      # K tail with an all-padding half step
])
def test_conv1x1_split32(B, H, W, ci, co, res, extra):
    from emdenoise import ops
    from oracle import tf_ops as T

    x = rnd((B, H, W, ci), 1, positive=True)
    w = rnd((1, 1, ci, co), 2, scale=(2.0 / (ci + co)) ** 0.5)
    s1, t1 = rnd((co,), 3, 0.3) + 1.0, rnd((co,), 4, 0.5)
    s2, t2 = rnd((co,), 5, 0.2) + 1.0, rnd((co,), 6, 0.3)
    r = rnd((B, H, W, co), 7)
    ref = T.relu6_t(T.conv2d_t(t64(x), t64(w), None) * t64(s1) + t64(t1))
    if extra:
        ref = T.relu6_t(ref * t64(s2) + t64(t2))
    if res:
        ref = ref + t64(r)
    pw = ops.PackedWeights(w[0], False, dev())
    xa = ops.Act(up(x))
    kw = dict(scale2=up(s2) if extra else None, shift2=up(t2) if extra else None, res=ops.Act(up(r)) if res else None)
    # the split32 output sits in a channel slice of a wider NaN-filled buffer (tf.concat targets, denoiser.py:203)
    wide = torch.full((B, H, W, co + 8), float("nan"), dtype=torch.float32, device=dev())
    from emdenoise import _lib

    xs = ops.to_split32(xa)
    got = ops.conv1x1_split32(xs, pw, up(s1), up(t1), ops.Act(wide, co, 4), **kw)
    old = ops.conv1x1(xa, pw, up(s1), up(t1), ops.Act.empty(B, H, W, co, dev()), precision=ops.PREC_BF16X3, **kw)
    try:   # the 32x32x16 form of the split32 GEMM sums exactly as the register-staged kernel does
        _lib.load().emd_debug_split_variant(3)
        v3 = ops.conv1x1_split32(xs, pw, up(s1), up(t1), ops.Act.empty(B, H, W, co, dev()), **kw)
        torch.cuda.synchronize()
    finally:
        _lib.load().emd_debug_split_variant(-1)
    assert rel_l2(got.torch().cpu().numpy(), ref.numpy()) < TOL_X3
    assert torch.equal(v3.torch(), old.torch()), "the 32x32x16 split32 GEMM must reproduce emd_conv1x1_f32 bit for bit"
    assert float((got.torch() - old.torch()).norm() / old.torch().norm()) < 1e-6   # the default (16x16x32 MFMAs): another summation order
    assert torch.isnan(wide[..., :4]).all() and torch.isnan(wide[..., 4 + co:]).all()


def test_conv1x1_split32_full_size_identity():
    """BASELINE size (B=32, 32x32x728 -> 728): the 32x32x16 form is bit-identical to the register-staged kernel over all 32768 rows; the
    default (16x16x32 MFMAs) agrees to 1e-6 relative, and its 128-row form (what a batch of 4 images runs on) gives the bits of the
    256-row form -- image b of a batch == the image alone, whatever kernel shape the batch size selects."""
    from emdenoise import ops

    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.rand((32, 32, 32, 728), generator=g).mul_(3.0).to(dev())
    w = (np.random.default_rng(6).standard_normal((1, 728, 728)) * 0.04).astype(np.float32)
    pw = ops.PackedWeights(w, False, dev())
    s, t = torch.ones(728, device=dev()), torch.zeros(728, device=dev())
    from emdenoise import _lib

    xa = ops.Act(x)
    xs = ops.to_split32(xa)
    b = ops.conv1x1(xa, pw, s, t, ops.Act.empty(32, 32, 32, 728, dev()))
    d = ops.conv1x1_split32(xs, pw, s, t, ops.Act.empty(32, 32, 32, 728, dev()))     # the default kernel: 16x16x32 MFMAs, 256-row tiles
    d4 = ops.conv1x1_split32(ops.to_split32(ops.Act(x[8:12].contiguous())), pw, s, t, ops.Act.empty(4, 32, 32, 728, dev()))   # 128 x 64 tiles
    d8 = ops.conv1x1_split32(ops.to_split32(ops.Act(x[16:24].contiguous())), pw, s, t, ops.Act.empty(8, 32, 32, 728, dev()))  # 128 x 128 tiles
    d1 = ops.conv1x1_split32(ops.to_split32(ops.Act(x[31:32].contiguous())), pw, s, t, ops.Act.empty(1, 32, 32, 728, dev()))
    try:
        _lib.load().emd_debug_split_variant(3)                                        # the 32x32x16 form
        a = ops.conv1x1_split32(xs, pw, s, t, ops.Act.empty(32, 32, 32, 728, dev()))
        torch.cuda.synchronize()
    finally:
        _lib.load().emd_debug_split_variant(-1)
    assert torch.equal(a.buf, b.buf)
    assert float((d.buf - b.buf).norm() / b.buf.norm()) < 1e-6
    assert torch.equal(d4.buf, d.buf[8:12]) and torch.equal(d8.buf, d.buf[16:24]) and torch.equal(d1.buf, d.buf[31:32])


@pytest.mark.parametrize("B,H,W,ci,co,res,extra", [
    (16, 32, 32, 728, 728, True, False),     # 64 x 4 tiles of 256 x 192: the smallest grid that takes the wide form; residual
    (33, 23, 23, 728, 728, False, True),     # M = 17457: a ragged last row tile; second affine
    (32, 32, 32, 256, 384, True, True),      # two column tiles, K = 8 steps
    (20, 32, 32, 96, 192, False, False),     # one column tile, K = 3 steps (fewer than the A ring has stages)
])
@pytest.mark.parametrize("form", [1, 2])
def test_conv1x1_split32_wide_tiles(B, H, W, ci, co, res, extra, form):
    """gemm_split16_wide_kernel (256 x 192 tiles; dev knob split_wide: 1 = 8 waves of 64 x 96, 2 = 4 waves of 128 x 96) gives the bits of
    gemm_split16_kernel (knob 0): same products, same order along K per output element -- so the tile the host picks by M never shows in
    a result -- and writes nothing outside its channel slice."""
    from emdenoise import _lib, ops

    x = rnd((B, H, W, ci), 71, positive=True)
    w = rnd((1, ci, co), 72, scale=0.04)
    r = rnd((B, H, W, co), 73)
    pw = ops.PackedWeights(w, False, dev())
    s1, t1 = up(rnd((co,), 74, 0.3) + 1.0), up(rnd((co,), 75, 0.5))
    s2, t2 = up(rnd((co,), 76, 0.2) + 1.0), up(rnd((co,), 77, 0.3))
    kw = dict(scale2=s2 if extra else None, shift2=t2 if extra else None, res=ops.Act(up(r)) if res else None)
    xs = ops.to_split32(ops.Act(up(x)))
    try:
        _lib.knob("split_wide", 0)
        ref = ops.conv1x1_split32(xs, pw, s1, t1, ops.Act.empty(B, H, W, co, dev()), **kw)
        _lib.knob("split_wide", form)
        wide = torch.full((B, H, W, co + 8), float("nan"), dtype=torch.float32, device=dev())
        got = ops.conv1x1_split32(xs, pw, s1, t1, ops.Act(wide, co, 4), **kw)
        torch.cuda.synchronize()
    finally:
        _lib.knob("split_wide", 0)
    assert torch.equal(got.torch(), ref.torch())
    assert torch.isnan(wide[..., :4]).all() and torch.isnan(wide[..., 4 + co:]).all()


@pytest.mark.parametrize("B,H,W,ci,co,res", [
    (16, 32, 32, 728, 728, True),      # 256 x 128 tiles, 23 K steps, residual
    (4, 32, 32, 728, 728, False),      # 128 x 64 tiles (the small-batch form)
    (2, 32, 32, 96, 192, False),       # 128 x 128 tiles, K = 3 steps: as many as the ring has stages
    (1, 16, 16, 32, 64, True),         # one K step: every younger DMA group is a surplus re-read of it
    (3, 23, 23, 64, 40, False),        # two K steps, ragged M, channel tail
])
def test_conv1x1_split32_two_steps_ahead(B, H, W, ci, co, res):
    """gemm_split16_kernel with the DMA of tile kt + 3 issued in step kt into the stage of tile kt (dev knob split_lead = 2, the default)
    gives the bits of the one-step-ahead schedule (1) at every tile form and K depth, and writes nothing outside its channel slice."""
    from emdenoise import _lib, ops

    x = rnd((B, H, W, ci), 81, positive=True)
    w = rnd((1, ci, co), 82, scale=0.04)
    r = rnd((B, H, W, co), 83)
    pw = ops.PackedWeights(w, False, dev())
    s1, t1 = up(rnd((co,), 84, 0.3) + 1.0), up(rnd((co,), 85, 0.5))
    kw = dict(res=ops.Act(up(r)) if res else None)
    xs = ops.to_split32(ops.Act(up(x)))
    try:
        _lib.knob("split_lead", 1)
        ref = ops.conv1x1_split32(xs, pw, s1, t1, ops.Act.empty(B, H, W, co, dev()), **kw)
        _lib.knob("split_lead", 2)
        wide = torch.full((B, H, W, co + 8), float("nan"), dtype=torch.float32, device=dev())
        got = ops.conv1x1_split32(xs, pw, s1, t1, ops.Act(wide, co, 4), **kw)
        torch.cuda.synchronize()
    finally:
        _lib.knob("split_lead", 2)
    assert not torch.isnan(got.torch()).any()
    assert torch.equal(got.torch(), ref.torch())
    assert torch.isnan(wide[..., :4]).all() and torch.isnan(wide[..., 4 + co:]).all()


def test_split32_argument_checks():
    from emdenoise import _lib, ops

    lib = _lib.load()
    x = ops.Act.empty(1, 4, 4, 64, dev())
    sp = ops.SplitAct(1, 4, 4, 64, dev())
    # ldy not a multiple of 32
    assert lib.emd_to_split32_f32(x.ptr, x.ld, sp.ptr, 68, 16, 64, None) != 0
    assert b"multiple of 32" in lib.emd_last_error()
    assert lib.emd_conv1x1_split32_supported(32768, 728, 728) == 1
    assert lib.emd_conv1x1_split32_supported(524288, 384, 256) == 1
    assert lib.emd_conv1x1_split32_supported(524288, 256, 256) == 1
    assert lib.emd_conv1x1_split32_supported(2097152, 384, 128) == 0
    assert lib.emd_conv1x1_split32_supported(32768, 64, 728) == 0
    assert lib.emd_conv1x1_split32_supported(1024, 728, 728) == 1   # round 3: at every M (the 128-row form below 192 tiles of 256 rows)


@pytest.mark.parametrize("B,H,W,ci,co,stride,rate,two_stage", [
    (2, 16, 16, 64, 128, 1, 1, True),     # X conv_block: conv + bias -> relu -> BN -> relu
    (1, 33, 21, 96, 132, 1, 1, False),    # ragged M, N tail, K tail (96 = 3 x 32)
    (1, 32, 32, 728, 728, 1, 6, False),   # D' ASPP rate branch
    (2, 18, 14, 40, 128, 2, 1, False),    # stride 2 (TF SAME: pad 0 before, 1 after), K padding inside a 32-group
    (2, 24, 24, 64, 64, 1, 1, True),      # 64-wide N tile (4-stage kernel): X's 512^2 conv_blocks
    (1, 17, 19, 32, 36, 1, 1, False),     # 64-wide N tile with an N tail, a single K step per tap
])
@pytest.mark.parametrize("out_split", [False, True])
def test_conv3x3_split32_equals_conv3x3(B, H, W, ci, co, stride, rate, two_stage, out_split):
    from emdenoise import ops

    x = rnd((B, H, W, ci), 51)
    w = rnd((9, ci, co), 52, scale=(2.0 / (9 * ci + co)) ** 0.5)
    s1, t1 = up(rnd((co,), 53, 0.3) + 1.0), up(rnd((co,), 54, 0.5))
    s2 = up(rnd((co,), 55, 0.2) + 1.0) if two_stage else None
    t2 = up(rnd((co,), 56, 0.3)) if two_stage else None
    pw = ops.PackedWeights(w, False, dev())
    xa = ops.Act(up(x))
    Ho, Wo = -(-H // stride), -(-W // stride)
    act = ops.ACT_RELU if two_stage else ops.ACT_RELU6
    want = ops.conv3x3(xa, pw, s1, t1, ops.Act.empty(B, Ho, Wo, co, dev()), stride=stride, rate=rate, act=act, scale2=s2, shift2=t2)
    xs = ops.to_split32(xa)
    if out_split:
        got = ops.SplitAct(B, Ho, Wo, co, dev())
        got.buf.fill_(float("nan"))
        ops.conv3x3_split32(xs, pw, s1, t1, got, stride=stride, rate=rate, act=act, scale2=s2, shift2=t2)
        torch.cuda.synchronize()
        assert torch.equal(got.buf.view(torch.int32), ops.to_split32(want).buf.view(torch.int32))
    else:
        got = ops.conv3x3_split32(xs, pw, s1, t1, ops.Act.empty(B, Ho, Wo, co, dev()), stride=stride, rate=rate, act=act,
                                  scale2=s2, shift2=t2)
        torch.cuda.synchronize()
        assert torch.equal(got.buf, want.buf)


@pytest.mark.parametrize("B,H,W,ci,co", [(2, 8, 8, 256, 256), (1, 13, 9, 64, 132), (2, 16, 16, 128, 64)])
@pytest.mark.parametrize("out_split", [False, True])
def test_deconv3x3s2_split32_equals_deconv(B, H, W, ci, co, out_split):
    from emdenoise import ops

    x = rnd((B, H, W, ci), 61)
    w = rnd((3, 3, co, ci), 62, scale=(2.0 / (9 * ci + co)) ** 0.5)   # slim.conv2d_transpose layout [kh,kw,Cout,Cin]
    s1, t1 = up(rnd((co,), 63, 0.3) + 1.0), up(rnd((co,), 64, 0.5))
    phases = ops.pack_deconv(w, dev())
    xa = ops.Act(up(x))
    want = ops.deconv3x3s2(xa, phases, s1, t1, ops.Act.empty(B, 2 * H, 2 * W, co, dev()))
    xs = ops.to_split32(xa)
    if out_split:
        got = ops.SplitAct(B, 2 * H, 2 * W, co, dev())
        got.buf.fill_(float("nan"))
        ops.deconv3x3s2_split32(xs, phases, s1, t1, got)
        torch.cuda.synchronize()
        assert torch.equal(got.buf.view(torch.int32), ops.to_split32(want).buf.view(torch.int32))
    else:
        got = ops.deconv3x3s2_split32(xs, phases, s1, t1, ops.Act.empty(B, 2 * H, 2 * W, co, dev()))
        torch.cuda.synchronize()
        assert torch.equal(got.buf, want.buf)
    # the one-launch form (four phases per workgroup): the same bits, into a NaN-filled buffer (every output pixel is written)
    if out_split:
        one = ops.SplitAct(B, 2 * H, 2 * W, co, dev())
        one.buf.fill_(float("nan"))
        ops.deconv3x3s2_fused(xs, phases, s1, t1, one)
        torch.cuda.synchronize()
        assert torch.equal(one.buf.view(torch.int32), got.buf.view(torch.int32))
    else:
        from emdenoise import _lib

        # every form of the one-launch kernel (dev knob deconv_direct: 0 = LDS-staged epilogue, 1 = epilogue from the registers on
        # 256-row tiles, 2 = the same on 128-row tiles at two workgroups per CU)
        try:
            for form in (0, 1, 2):
                _lib.knob("deconv_direct", form)
                wide = torch.full((B, 2 * H, 2 * W, co + 8), float("nan"), dtype=torch.float32, device=dev())
                one = ops.deconv3x3s2_fused(xs, phases, s1, t1, ops.Act(wide, co, 4))
                torch.cuda.synchronize()
                assert torch.equal(one.torch(), got.torch()), form
                assert torch.isnan(wide[..., :4]).all() and torch.isnan(wide[..., 4 + co:]).all()
        finally:
            _lib.knob("deconv_direct", DECONV_DEFAULT)


@pytest.mark.parametrize("B,H,W,ci,co,tpw", [
    (2, 8, 32, 128, 128, 0),      # one tile per image (the top / left halo is all padding), two column tiles (D's deconv1to0 widths)
    (1, 16, 96, 64, 64, 0),       # left, interior and right tiles, two tile rows
    (1, 8, 128, 256, 132, 2),     # three column tiles (the last: 4 live columns), 8 chunks, two tiles per workgroup
    (2, 24, 64, 96, 36, 0),       # channel tail inside the only column tile, three chunks
    (1, 8, 256, 32, 64, 4),       # one chunk, four tiles per workgroup (the pointer-increment path)
])
@pytest.mark.parametrize("out_split", [False, True])
@pytest.mark.parametrize("epi", [0, 4])
def test_deconv_pipe_patch_resident_kernel(B, H, W, ci, co, tpw, out_split, epi):
    """csrc/deconv_pipe.hip (dev knob deconv_direct = 3; slim.conv2d_transpose k 3 s 2, denoiser.py:138-150) through
    emd_deconv3x3s2_fused_split32_f32: against the register-staged four-phase GEMM (which tests/test_ops_gpu.py holds to the oracle) at
    1e-6 -- another summation order, not bits -- into a NaN-filled concat slice, and as a split32 tensor."""
    from emdenoise import _lib, ops
    from oracle import tf_ops as T

    x = rnd((B, H, W, ci), 261)
    w = rnd((3, 3, co, ci), 262, scale=(2.0 / (9 * ci + co)) ** 0.5)
    s1, t1 = up(rnd((co,), 263, 0.3) + 1.0), up(rnd((co,), 264, 0.5))
    phases = ops.pack_deconv(w, dev())
    xa = ops.Act(up(x))
    want = ops.deconv3x3s2(xa, phases, s1, t1, ops.Act.empty(B, 2 * H, 2 * W, co, dev()))
    ref = T.relu6_t(T.conv2d_transpose_s2_t(t64(x), t64(w)) * t64(s1.cpu().numpy()) + t64(t1.cpu().numpy())).numpy()
    xs = ops.to_split32(xa)
    try:
        _lib.knob("deconv_direct", 3)
        _lib.knob("epi_width", epi)      # dwords a lane stores at a time (dev knob)
        _lib.knob("sep_tpw", tpw)
        wide = torch.full((B, 2 * H, 2 * W, co + 8), float("nan"), dtype=torch.float32, device=dev())
        got = ops.deconv3x3s2_fused(xs, phases, s1, t1, ops.Act(wide, co, 4))
        if out_split:
            sp = ops.SplitAct(B, 2 * H, 2 * W, co, dev())
            sp.buf.fill_(float("nan"))
            ops.deconv3x3s2_fused(xs, phases, s1, t1, sp)
        torch.cuda.synchronize()
    finally:
        _lib.knob("deconv_direct", DECONV_DEFAULT)
        _lib.knob("epi_width", 0)
        _lib.knob("sep_tpw", 0)
    g_np = got.torch().cpu().numpy()
    assert not np.isnan(g_np).any()
    assert rel_l2(g_np, ref) < TOL_X3
    assert float((got.torch() - want.torch()).norm() / want.torch().norm()) < 1e-6
    assert torch.isnan(wide[..., :4]).all() and torch.isnan(wide[..., 4 + co:]).all()
    if out_split:
        assert torch.equal(sp.buf.view(torch.int32), ops.to_split32(ops.Act(got.torch().contiguous())).buf.view(torch.int32))


def test_deconv_pipe_full_size_properties_and_route():
    """Graph D's deconv1to0 at [8,256,256,128] -> [8,512,512,128] on the patch-resident kernel: image b of the batch == the image alone,
    bit for bit; the pre-activation is exactly linear under powers of two; and the host-side route query is independent of the batch size
    wherever that kernel applies (it sums in another order than the GEMM forms: an M-dependent choice would show in the bits)."""
    from emdenoise import _lib, ops

    lib = _lib.load()
    B, S, c = 8, 256, 128
    g = torch.Generator(device=dev()).manual_seed(9)
    x = torch.rand(B, S, S, c, device=dev(), generator=g)
    phases = ops.pack_deconv(rnd((3, 3, c, c), 271, 0.03), dev())
    s1, t0 = torch.rand(c, device=dev(), generator=g) + 0.5, torch.zeros(c, device=dev())
    f = lambda xx: ops.deconv3x3s2_fused(ops.to_split32(ops.Act(xx)), phases, s1, t0, ops.Act.empty(xx.shape[0], 2 * S, 2 * S, c, dev()), act=False).torch()
    y, y1, y2 = f(x), f(x[5:6].contiguous()), f(2 * x)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    assert torch.equal(y[5:6], y1)
    assert torch.equal(y2, 2 * y)
    for (H, W, ci) in ((256, 256, 128), (128, 128, 256), (8, 32, 32)):
        assert {lib.emd_deconv3x3s2_fused_preferred(b, H, W, ci, ci) for b in (1, 2, 4, 32)} == {1}
    assert lib.emd_deconv3x3s2_fused_preferred(1, 20, 20, 128, 128) == 0 and lib.emd_deconv3x3s2_fused_preferred(256, 20, 20, 128, 128) == 1


@pytest.mark.parametrize("B,H,W,ci,co,res", [(2, 16, 16, 256, 256, True), (1, 24, 20, 728, 132, False), (1, 8, 8, 64, 36, True)])
def test_conv1x1_split32_with_split32_output(B, H, W, ci, co, res):
    """emd_conv1x1_split32_out_f32: the split32 tensor it writes == emd_to_split32_f32 of what emd_conv1x1_split32_f32 writes
    (padding channels zero), bit for bit."""
    from emdenoise import ops

    x = rnd((B, H, W, ci), 91, positive=True)
    pw = ops.PackedWeights(rnd((1, ci, co), 92, scale=(2.0 / (ci + co)) ** 0.5), False, dev())
    s1, t1 = up(rnd((co,), 93, 0.3) + 1.0), up(rnd((co,), 94, 0.5))
    r = ops.Act(up(rnd((B, H, W, co), 95))) if res else None
    xs = ops.to_split32(ops.Act(up(x)))
    want = ops.conv1x1_split32(xs, pw, s1, t1, ops.Act.empty(B, H, W, co, dev()), res=r)
    got = ops.SplitAct(B, H, W, co, dev())
    got.buf.fill_(float("nan"))
    ops.conv1x1_split32(xs, pw, s1, t1, got, res=r)
    torch.cuda.synchronize()
    assert torch.equal(got.buf.view(torch.int32), ops.to_split32(want).buf.view(torch.int32))


@pytest.mark.parametrize("B,H,W,ci,co,res,extra,split", [
    (2, 32, 32, 728, 728, True, False, False), (1, 24, 20, 728, 728, False, True, False), (1, 8, 8, 160, 132, True, False, False),
    (1, 8, 8, 48, 128, False, False, False), (2, 16, 16, 256, 256, True, False, True), (1, 24, 20, 728, 132, False, False, True),
    (1, 8, 8, 64, 36, True, True, True)])
def test_conv1x1_split32_direct_epilogue(B, H, W, ci, co, res, extra, split):
    """Dev variant 7 (operands swapped, epilogue straight from the accumulator registers) writes the default kernel's bits: fp32 into a
    channel slice of a wider buffer, or split32 (padding channels zero); ragged M, N tails, residual, second affine stage."""
    from emdenoise import _lib, ops

    lib = _lib.load()
    x = rnd((B, H, W, ci), 61, positive=True)
    pw = ops.PackedWeights(rnd((1, ci, co), 62, scale=(2.0 / (ci + co)) ** 0.5), False, dev())
    s1, t1 = up(rnd((co,), 63, 0.3) + 1.0), up(rnd((co,), 64, 0.5))
    kw = dict(scale2=up(rnd((co,), 65, 0.2) + 1.0) if extra else None, shift2=up(rnd((co,), 66, 0.3)) if extra else None,
              res=ops.Act(up(rnd((B, H, W, co), 67))) if res else None)
    xs = ops.to_split32(ops.Act(up(x)))

    def run():
        if split:
            out = ops.SplitAct(B, H, W, co, dev())
            out.buf.fill_(float("nan"))
            ops.conv1x1_split32(xs, pw, s1, t1, out, **kw)
            return out.buf.view(torch.int32).clone()
        wide = torch.full((B, H, W, co + 8), float("nan"), dtype=torch.float32, device=dev())
        ops.conv1x1_split32(xs, pw, s1, t1, ops.Act(wide, co, 4), **kw)
        return wide.view(torch.int32).clone()

    try:
        lib.emd_debug_split_variant(3)
        want = run()
        lib.emd_debug_split_variant(7)
        got = run()
        torch.cuda.synchronize()
    finally:
        lib.emd_debug_split_variant(-1)
    assert torch.equal(got, want)


@pytest.mark.parametrize("B,H,W,ci,co,res", [(2, 16, 32, 128, 128, True), (1, 8, 16, 64, 64, False), (1, 24, 48, 384, 96, True)])
def test_sep_fused_with_split32_output(B, H, W, ci, co, res):
    """emd_sep3x3_fused_out_f32 == emd_to_split32_f32(emd_sep3x3_fused_f32), bit for bit."""
    from emdenoise import ops

    x = ops.Act(up(rnd((B, H, W, ci), 96, positive=True)))
    dw = up(rnd((9, ci), 97, 0.35))
    pw = ops.PackedWeights(rnd((1, ci, co), 98, scale=(2.0 / (ci + co)) ** 0.5), False, dev())
    s1, t1 = up(rnd((co,), 99, 0.3) + 1.0), up(rnd((co,), 100, 0.5))
    r = ops.Act(up(rnd((B, H, W, co), 101))) if res else None
    want = ops.sep_fused(x, dw, pw, s1, t1, ops.Act.empty(B, H, W, co, dev()), res=r)
    got = ops.SplitAct(B, H, W, co, dev())
    got.buf.fill_(float("nan"))
    ops.sep_fused(x, dw, pw, s1, t1, got, res=r)
    torch.cuda.synchronize()
    assert torch.equal(got.buf.view(torch.int32), ops.to_split32(want).buf.view(torch.int32))


@pytest.mark.parametrize("B,H,W,ci,co,tpw", [
    (2, 8, 32, 64, 64, 0),        # one tile per image: every patch border is padding
    (1, 16, 96, 128, 64, 0),      # left-edge, interior and right-edge tiles, two tile rows (X decoder's 128 -> 64)
    (1, 8, 256, 32, 64, 4),       # one chunk; four tiles per workgroup: the pointer-increment path between interior tiles
    (2, 24, 64, 96, 36, 2),       # channel tail (36 of 64 columns), three chunks, two tiles per workgroup
    (1, 8, 64, 40, 64, 0),        # Cin not a multiple of 32: the split32 tensor's zero padding channels are read as the last chunk
    (1, 16, 64, 192, 128, 0),     # two column tiles of 64 (X decoder's 192 -> 128)
    (1, 8, 32, 64, 132, 0),       # three column tiles, the last with 4 live columns
    (2, 8, 64, 96, 192, 2),       # three full column tiles, two pixel tiles per workgroup
])
@pytest.mark.parametrize("out_split", [False, True])
@pytest.mark.parametrize("epi", [0, 4])
def test_conv3_pipe_patch_resident_kernel(B, H, W, ci, co, tpw, out_split, epi):
    """csrc/conv3_pipe.hip (dense 3x3, stride 1, <= 64 output channels, the patch resident in LDS across the nine taps; reached through
    emd_conv3x3_split32_f32): X's conv_block against the oracle (float64), against the tap-major GEMM it replaces (dev knob
    conv3_pipe = 0; another summation order: 1e-6, not bits), fp32 output into a NaN-filled concat slice and split32 output equal to
    emd_to_split32_f32 of the fp32 one; with the per-channel dword epilogue (epi = 0: the kernel's rule) and the transposed 16-byte one
    (dev knob epi_width = 4)."""
    from emdenoise import _lib, ops
    from oracle import tf_ops as T

    x = rnd((B, H, W, ci), 171)
    w = rnd((3, 3, ci, co), 172, scale=(2.0 / (9 * ci + co)) ** 0.5)
    bias, g, h = rnd((co,), 173, 0.2), rnd((co,), 174, 0.3) + 1.0, rnd((co,), 175, 0.4)
    ref = torch.relu(torch.relu(T.conv2d_t(t64(x), t64(w), t64(bias))) * t64(g) + t64(h)).numpy()
    pw = ops.PackedWeights(w.reshape(9, ci, co), False, dev())
    xs = ops.to_split32(ops.Act(up(x)))
    one = up(np.ones(co, np.float32))
    kw = dict(act=ops.ACT_RELU, scale2=up(g), shift2=up(h))
    try:
        _lib.knob("sep_tpw", tpw)
        _lib.knob("epi_width", epi)
        wide = torch.full((B, H, W, co + 8), float("nan"), dtype=torch.float32, device=dev())
        got = ops.conv3x3_split32(xs, pw, one, up(bias), ops.Act(wide, co, 4), **kw)
        if out_split:
            sp = ops.SplitAct(B, H, W, co, dev())
            sp.buf.fill_(float("nan"))
            ops.conv3x3_split32(xs, pw, one, up(bias), sp, **kw)
        _lib.knob("conv3_pipe", 0)
        old = ops.conv3x3_split32(xs, pw, one, up(bias), ops.Act.empty(B, H, W, co, dev()), **kw)
        torch.cuda.synchronize()
    finally:
        _lib.knob("conv3_pipe", 1)
        _lib.knob("sep_tpw", 0)
        _lib.knob("epi_width", 0)
    g_np = got.torch().cpu().numpy()
    assert not np.isnan(g_np).any()
    assert rel_l2(g_np, ref) < TOL_X3
    assert float((got.torch() - old.torch()).norm() / old.torch().norm()) < 1e-6
    assert torch.isnan(wide[..., :4]).all() and torch.isnan(wide[..., 4 + co:]).all()
    if out_split:
        want = ops.to_split32(ops.Act(got.torch().contiguous()))
        assert torch.equal(sp.buf.view(torch.int32), want.buf.view(torch.int32))


def test_conv3_pipe_full_size_properties():
    """Graph X's 64 -> 64 decoder layer at full size, [8,512,512,64]: image b of the batch == the image alone, bit for bit, and the
    pre-activation is exactly linear in the input under powers of two."""
    from emdenoise import ops

    B, S, ci, co = 8, 512, 64, 64
    g = torch.Generator(device=dev()).manual_seed(7)
    x = torch.rand(B, S, S, ci, device=dev(), generator=g)
    pw = ops.PackedWeights(rnd((9, ci, co), 181, 0.05), False, dev())
    s1, t0 = torch.rand(co, device=dev(), generator=g) + 0.5, torch.zeros(co, device=dev())
    f = lambda xx: ops.conv3x3_split32(ops.to_split32(ops.Act(xx)), pw, s1, t0, ops.Act.empty(xx.shape[0], S, S, co, dev()), act=ops.ACT_NONE).torch()
    y, y1, y2 = f(x), f(x[5:6].contiguous()), f(2 * x)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    assert torch.equal(y[5:6], y1)
    assert torch.equal(y2, 2 * y)


@pytest.mark.parametrize("B,H,W,ci,co,stride,rate", [
    (2, 16, 16, 64, 128, 1, 1), (1, 33, 21, 96, 132, 1, 1), (1, 32, 32, 728, 728, 1, 6), (2, 18, 14, 40, 128, 2, 1),
    (2, 24, 24, 64, 64, 1, 1), (1, 17, 19, 32, 36, 1, 3)])
def test_conv3x3_split32_against_the_oracle(B, H, W, ci, co, stride, rate):
    """emd_conv3x3_split32_f32 directly against oracle/tf_ops.py (float64 tf.layers.conv2d, TF SAME, dilation): the
    bit-identity tests above compare it with the register-staged kernel only, i.e. with a sibling."""
    from emdenoise import ops
    from oracle import tf_ops as T

    x = rnd((B, H, W, ci), 71)
    w = rnd((3, 3, ci, co), 72, scale=(2.0 / (9 * ci + co)) ** 0.5)
    bias, g, h = rnd((co,), 73, 0.2), rnd((co,), 74, 0.3) + 1.0, rnd((co,), 75, 0.4)
    # X's conv_block (modified_Xception.py:215-229): conv + bias -> relu -> affine (BN on moving statistics) -> relu
    ref = torch.relu(torch.relu(T.conv2d_t(t64(x), t64(w), t64(bias), stride=stride, rate=rate)) * t64(g) + t64(h)).numpy()
    pw = ops.PackedWeights(w.reshape(9, ci, co), False, dev())
    Ho, Wo = -(-H // stride), -(-W // stride)
    xs = ops.to_split32(ops.Act(up(x)))
    one = up(np.ones(co, np.float32))
    got = ops.conv3x3_split32(xs, pw, one, up(bias), ops.Act.empty(B, Ho, Wo, co, dev()), stride=stride, rate=rate, act=ops.ACT_RELU,
                              scale2=up(g), shift2=up(h))
    sp = ops.conv3x3_split32(xs, pw, one, up(bias), ops.SplitAct(B, Ho, Wo, co, dev()), stride=stride, rate=rate, act=ops.ACT_RELU,
                             scale2=up(g), shift2=up(h))
    torch.cuda.synchronize()
    assert rel_l2(got.torch().cpu().numpy(), ref) < TOL_X3
    assert rel_l2(sp.to_float().cpu().numpy(), ref) < TOL_X3 + 2.0 ** -16       # + the split32 output's own rounding


@pytest.mark.parametrize("B,H,W,ci,co", [(2, 8, 8, 256, 256), (1, 13, 9, 64, 132), (2, 16, 16, 128, 64), (1, 32, 32, 128, 128)])
def test_deconv3x3s2_split32_against_the_oracle(B, H, W, ci, co):
    """emd_deconv3x3s2_split32_f32 directly against oracle/tf_ops.py conv2d_transpose_s2_t (float64: the gradient of the SAME
    stride-2 conv, cropped at the end; denoiser.py:138-150) + bias + folded BN + relu6."""
    from emdenoise import ops
    from oracle import tf_ops as T

    x = rnd((B, H, W, ci), 81)
    w = rnd((3, 3, co, ci), 82, scale=(2.0 / (9 * ci + co)) ** 0.5)
    bias, g, h = rnd((co,), 83, 0.2), rnd((co,), 84, 0.3) + 1.0, rnd((co,), 85, 0.4)
    ref = T.relu6_t(T.conv2d_transpose_s2_t(t64(x), t64(w), t64(bias)) * t64(g) + t64(h)).numpy()
    phases = ops.pack_deconv(w, dev())
    shift = (bias.astype(np.float64) * g + h).astype(np.float32)
    xs = ops.to_split32(ops.Act(up(x)))
    got = ops.deconv3x3s2_split32(xs, phases, up(g), up(shift), ops.Act.empty(B, 2 * H, 2 * W, co, dev()))
    one = ops.deconv3x3s2_fused(xs, phases, up(g), up(shift), ops.Act.empty(B, 2 * H, 2 * W, co, dev()))
    torch.cuda.synchronize()
    assert rel_l2(got.torch().cpu().numpy(), ref) < TOL_X3
    assert rel_l2(one.torch().cpu().numpy(), ref) < TOL_X3


def test_split32_convs_random_shapes_match_register_staged_kernels():
    """Seeded sweep over ragged shapes (M, N and K tails, 1..3 M tiles, strides, dilations): every split32 GEMM form gives
    the bits of its register-staged twin -- the DMA source addressing (per-tap rows, zero line, swizzle) has no shape it gets wrong."""
    from emdenoise import _lib, ops

    rng = np.random.default_rng(2024)
    for case in range(24):
        B = int(rng.integers(1, 3))
        H, W = int(rng.integers(5, 29)), int(rng.integers(5, 29))
        ci = int(rng.choice([4, 8, 20, 32, 36, 64, 96, 100, 160]))
        co = int(rng.choice([4, 36, 64, 68, 128, 132, 200]))
        kind = case % 3
        x = rnd((B, H, W, ci), 100 + case)
        xa = ops.Act(up(x))
        xs = ops.to_split32(xa)
        s1, t1 = up(rnd((co,), 200 + case, 0.3) + 1.0), up(rnd((co,), 300 + case, 0.5))
        if kind == 0:      # dense 3x3, stride 1 (optionally dilated) or 2
            stride = int(rng.integers(1, 3))
            rate = int(rng.integers(1, 4)) if stride == 1 else 1
            pw = ops.PackedWeights(rnd((9, ci, co), 400 + case, 0.05), False, dev())
            Ho, Wo = -(-H // stride), -(-W // stride)
            want = ops.conv3x3(xa, pw, s1, t1, ops.Act.empty(B, Ho, Wo, co, dev()), stride=stride, rate=rate)
            got = ops.conv3x3_split32(xs, pw, s1, t1, ops.Act.empty(B, Ho, Wo, co, dev()), stride=stride, rate=rate)
        elif kind == 1:    # transposed 3x3 stride 2
            ph = ops.pack_deconv(rnd((3, 3, co, ci), 400 + case, 0.05), dev())
            want = ops.deconv3x3s2(xa, ph, s1, t1, ops.Act.empty(B, 2 * H, 2 * W, co, dev()))
            got = ops.deconv3x3s2_split32(xs, ph, s1, t1, ops.Act.empty(B, 2 * H, 2 * W, co, dev()))
        else:              # pointwise
            pw = ops.PackedWeights(rnd((1, ci, co), 400 + case, 0.05), False, dev())
            want = ops.conv1x1(xa, pw, s1, t1, ops.Act.empty(B, H, W, co, dev()))
            got = ops.conv1x1_split32(xs, pw, s1, t1, ops.Act.empty(B, H, W, co, dev()))       # default: 16x16x32 MFMAs, 128-row tiles here
            _lib.load().emd_debug_split_variant(3)
            got3 = ops.conv1x1_split32(xs, pw, s1, t1, ops.Act.empty(B, H, W, co, dev()))      # the 32x32x16 form: the twin's bits
            _lib.load().emd_debug_split_variant(-1)
            torch.cuda.synchronize()
            assert float((got.buf - want.buf).norm() / want.buf.norm()) < 1e-6, (case, kind, B, H, W, ci, co)
            got = got3
        torch.cuda.synchronize()
        assert torch.equal(got.buf, want.buf), (case, kind, B, H, W, ci, co)


@pytest.mark.parametrize("B,H,W,C,stride,rate,split", [(2, 16, 16, 728, 1, 1, True), (1, 70, 33, 64, 1, 1, False),
                                                        (2, 17, 13, 40, 2, 1, True), (1, 32, 32, 96, 1, 6, False)])
def test_dw3x3_pre_equals_affine_then_dw3x3(B, H, W, C, stride, rate, split):
    """The depthwise kernels with the previous block's norm + relu applied on the fly (emd_dw3x3_pre*_f32) give the bits of
    emd_affine_act_f32 followed by the plain kernel (padding is applied after the activation)."""
    from emdenoise import ops

    x = rnd((B, H, W, C), 71)
    w = up(rnd((9, C), 72, 0.3))
    sc, sh = up(rnd((C,), 73, 0.5) + 1.0), up(rnd((C,), 74, 0.5))
    xa = ops.Act(up(x))
    act = ops.affine_act(xa, sc, sh, ops.Act.empty(B, H, W, C, dev()), act=ops.ACT_RELU)
    Ho, Wo = -(-H // stride), -(-W // stride)
    if split:
        want = ops.dw3x3_split32(act, w, ops.SplitAct(B, Ho, Wo, C, dev()), stride=stride, rate=rate)
        got = ops.dw3x3_split32(xa, w, ops.SplitAct(B, Ho, Wo, C, dev()), stride=stride, rate=rate, pre=(sc, sh))
    else:
        want = ops.dw3x3(act, w, ops.Act.empty(B, Ho, Wo, C, dev()), stride=stride, rate=rate)
        got = ops.dw3x3(xa, w, ops.Act.empty(B, Ho, Wo, C, dev()), stride=stride, rate=rate, pre=(sc, sh))
    torch.cuda.synchronize()
    assert torch.equal(got.buf.view(torch.int32), want.buf.view(torch.int32))


@pytest.mark.parametrize("B,H,W,ci,co", [(32, 32, 32, 728, 728), (2, 19, 23, 96, 132), (1, 16, 16, 64, 36)])
def test_conv1x1_split32_stats_epilogue(B, H, W, ci, co):
    """The statistics epilogue (emd_conv1x1_split32_stats_f32): same output bits as the plain GEMM, and mean / biased variance
    of that output equal to a second pass over it (emd_bn_stats_f32) to double-rounding noise; ragged M and N tails."""
    from emdenoise import ops

    x = rnd((B, H, W, ci), 81)
    w = rnd((1, ci, co), 82, scale=0.1)
    pw = ops.PackedWeights(w, False, dev())
    s1, t1 = up(rnd((co,), 83, 0.3) + 1.0), up(rnd((co,), 84, 0.5))
    xs = ops.to_split32(ops.Act(up(x)))
    plain = ops.conv1x1_split32(xs, pw, s1, t1, ops.Act.empty(B, H, W, co, dev()), act=ops.ACT_NONE)
    y, mean, var = ops.conv1x1_split32(xs, pw, s1, t1, ops.Act.empty(B, H, W, co, dev()), act=ops.ACT_NONE, stats=True)
    m2, v2 = ops.bn_batch_stats(plain)
    torch.cuda.synchronize()
    assert torch.equal(y.buf, plain.buf)
    ref = plain.buf.double().reshape(-1, co)
    assert torch.allclose(mean.double(), ref.mean(0), rtol=1e-6, atol=1e-7)
    assert torch.allclose(var.double(), ref.var(0, unbiased=False), rtol=1e-5, atol=1e-9)
    assert torch.allclose(mean, m2, rtol=1e-6, atol=1e-7) and torch.allclose(var, v2, rtol=1e-5, atol=1e-9)
    # deterministic: a second run gives the same bits
    _, mean_b, var_b = ops.conv1x1_split32(xs, pw, s1, t1, ops.Act.empty(B, H, W, co, dev()), act=ops.ACT_NONE, stats=True)
    assert torch.equal(mean, mean_b) and torch.equal(var, var_b)
    # the norm folded in the same final-reduction launch == emd_bn_fold_f32 on those statistics, bit for bit (with and without gamma)
    beta, gamma = up(rnd((co,), 85, 0.4)), up(rnd((co,), 86, 0.2) + 1.0)
    for g in (None, gamma):
        y3, m3, v3, sc, sh = ops.conv1x1_split32(xs, pw, s1, t1, ops.Act.empty(B, H, W, co, dev()), act=ops.ACT_NONE, stats=True,
                                                 fold=(g, beta, 1e-3))
        sc2, sh2 = ops.bn_fold(mean, var, g, beta, 1e-3)
        torch.cuda.synchronize()
        assert torch.equal(y3.buf, plain.buf) and torch.equal(m3, mean) and torch.equal(v3, var)
        assert torch.equal(sc, sc2) and torch.equal(sh, sh2)
