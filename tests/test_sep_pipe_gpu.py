"""GPU parity tests of the LDS-DMA pipelined fused separable convs (csrc/sep_pipe.hip and, round 4, csrc/sep_pipe2.hip), reached through the same C-ABI entry points
as the register-staged kernel it replaces for W % 32 == 0 (emd_sep3x3_fused_f32 / _out_f32 / _reflect_f32 / emd_sep3x3_dual_f32):

  * against the oracle's TF-op restatement (oracle/tf_ops.py, float64): machine_learning/denoiser.py:110-136 (the separable block),
    :356-359 / :368-371 / :380-383 (block + 1x1 projection of the same input), relative L2 < 2e-5 (split-bf16);
  * bit for bit against csrc/sep_fused.hip (dev knob sep_pipe = 0), both issue schedules (knob sep_mode) and both workgroup shapes (knob sep_nw), because the two kernels
    promise the same products in the same order -- which is what keeps "image b of a batch == the image alone" exact when a
    shape falls to one kernel or the other;
  * edge cases: one tile (every border is padding), tiles on the left / right image edge and between, several tiles per workgroup,
    one-chunk inputs, channel tails (Cout < tile columns), concat-slice inputs and outputs (NaN-filled surroundings stay NaN),
    second affine, residual, REFLECT borders (graph G), leaky relu, split32 output.
"""
import numpy as np
import pytest
import torch

from tests.test_ops_gpu import TOL_X3, dev, out_act, rel_l2, rnd, t64, to_act

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[(0, 2), (4, 2), (0, 0)], ids=["epi_rule", "epi_16B", "lockstep"])
def _restore_knobs(request):
    """Every test runs (a) on the software-pipelined kernel (csrc/sep_pipe2.hip, round 4) wherever it has an instance -- dev knob sep_pipe2
    = 2; it is off by default (0: measured no faster, profiles/r04_experiments.txt) -- with the epilogue its rule picks (per-channel dword
    stores, mostly), (b) the same with the transposed 16-byte epilogue forced (dev knob epi_width = 4): same values, other lanes, and
    (c) with sep_pipe2 = 0: the lockstep kernel (csrc/sep_pipe.hip) alone, as in round 3.  sep_nw = 4 always means sep_pipe.hip's
    4-wave form."""
    from emdenoise import _lib

    _lib.knob("epi_width", request.param[0])
    _lib.knob("sep_pipe2", request.param[1])
    yield
    for k, v in (("sep_pipe", 1), ("sep_pipe2", 0), ("sep_mode", -1), ("sep_tpw", 0), ("sep_nw", 0), ("epi_width", 0)):
        _lib.knob(k, v)


def _oracle(x, dw, pw, s1, t1, s2t2, r, act, reflect):
    from oracle import tf_ops as T

    if reflect:   # tf.pad(REFLECT, 1) + VALID depthwise == the interior of the SAME depthwise conv of the padded image
        xp = torch.from_numpy(np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)), mode="reflect"))
        d = T.depthwise_conv2d_t(xp, t64(dw))[:, 1:-1, 1:-1, :]
    else:
        d = T.depthwise_conv2d_t(t64(x), t64(dw))
    y = T.conv2d_t(d, t64(pw)) * t64(s1) + t64(t1)
    y = T.relu6_t(y) if act == 1 else (torch.where(y > 0, y, 0.2 * y) if act == 4 else y)
    if s2t2 is not None:
        y = T.relu6_t(y * t64(s2t2[0]) + t64(s2t2[1]))
    if r is not None:
        y = y + t64(r)
    return y.numpy()


CASES = [
    # B, H, W, Cin, Cout, res, extra, act, reflect, tpw
    (2, 8, 32, 64, 64, False, False, 1, False, 0),       # one tile per image: every patch border is padding
    (1, 16, 96, 128, 64, True, False, 1, False, 0),      # left-edge, interior and right-edge tiles; residual
    (2, 24, 64, 32, 64, False, True, 1, False, 2),       # one chunk per tile, two tiles per workgroup, second affine
    (1, 8, 256, 96, 128, True, True, 1, False, 8),       # eight tiles per workgroup: the pointer-increment path between interior tiles
    (2, 16, 32, 384, 128, False, False, 1, False, 0),    # 12 chunks (deconv1_a)
    (1, 16, 64, 128, 256, True, False, 1, False, 0),     # 256 columns, one weight tile in LDS (cnn2 / deconv2_b)
    (1, 8, 32, 64, 36, False, False, 0, False, 0),       # channel tail in a 64-column tile, no activation
    (1, 8, 64, 64, 160, True, False, 1, False, 0),       # channel tail in a 256-column tile
    (1, 16, 48, 64, 64, True, False, 1, False, 0),       # W % 32 != 0: only the 4-wave form (8 x 16 tiles) covers it
    (2, 8, 80, 96, 128, True, True, 1, False, 0),        # the same with 128 columns (five tiles of 16, three chunks)
    (2, 16, 32, 64, 32, False, False, 4, True, 0),       # graph G: REFLECT border, leaky relu
    (1, 8, 64, 32, 128, True, True, 4, True, 2),
]


@pytest.mark.parametrize("B,H,W,ci,co,res,extra,act,reflect,tpw", CASES)
@pytest.mark.parametrize("lead,nw", [(0, 8), (1, 8), (0, 4), (1, 4)])
def test_sep_pipe_vs_oracle_and_register_staged_kernel(B, H, W, ci, co, res, extra, act, reflect, tpw, lead, nw):
    from emdenoise import _lib, ops

    if nw == 4 and co > 128:
        pytest.skip("the 4-wave form (8 x 16 tiles, two workgroups per CU) covers up to 128 output channels")

    x = rnd((B, H, W, ci), 340, positive=not reflect)
    dw = rnd((3, 3, ci, 1), 341, 0.35)
    pw = rnd((1, 1, ci, co), 342, scale=(2.0 / (ci + co)) ** 0.5)
    s1, t1 = rnd((co,), 343, 0.2) + 1, rnd((co,), 344, 0.5)
    s2, t2 = rnd((co,), 345, 0.2) + 1, rnd((co,), 346, 0.5)
    r = rnd((B, H, W, co), 347, positive=True) if res else None
    want = _oracle(x, dw, pw, s1, t1, (s2, t2) if extra else None, r, act, reflect)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())
    xa = to_act(x, ld=ci + 64, c0=32)
    pk = ops.PackedWeights(pw[0], False, dev())
    kw = dict(scale2=d(s2) if extra else None, shift2=d(t2) if extra else None, res=to_act(r, ld=co + 12, c0=8) if res else None,
              act=act, reflect=reflect)

    def run(pipe):
        _lib.knob("sep_pipe", pipe)
        _lib.knob("sep_mode", lead)
        _lib.knob("sep_nw", nw)
        _lib.knob("sep_tpw", tpw)
        out = out_act(B, H, W, co, ld=co + 8, c0=4)
        ops.sep_fused(xa, d(dw[..., 0]), pk, d(s1), d(t1), out, **kw)
        torch.cuda.synchronize()
        return out

    new, old = run(1), run(0)
    got = new.torch().cpu().numpy()
    assert not np.isnan(got).any()
    assert rel_l2(got, want) < TOL_X3
    full = new.buf.cpu().numpy()
    assert np.isnan(full[..., :4]).all() and np.isnan(full[..., 4 + co:]).all()   # nothing written outside the slice
    assert torch.equal(new.torch(), old.torch()), "sep_pipe and sep_fused promise the same bits"


@pytest.mark.parametrize("B,H,W,ci,co,res", [(1, 16, 64, 128, 128, True), (2, 8, 32, 256, 256, True), (1, 8, 96, 64, 224, False),
                                             (1, 8, 48, 64, 96, True)])      # W % 32 != 0: 4-wave form only
@pytest.mark.parametrize("lead", [0, 1])
@pytest.mark.parametrize("nw", [8, 4])
def test_sep_pipe_split32_output(B, H, W, ci, co, res, lead, nw):
    """emd_sep3x3_fused_out_f32 (the producer of a split32 convolution's input: deconv1_b / deconv2_b, denoiser.py:357, :369) ==
    emd_to_split32_f32 of the fp32 output, padding channels zero."""
    from emdenoise import _lib, ops

    x = rnd((B, H, W, ci), 350, positive=True)
    dw = rnd((9, ci), 351, 0.35)
    pw = rnd((1, ci, co), 352, scale=(2.0 / (ci + co)) ** 0.5)
    s1, t1 = rnd((co,), 353, 0.2) + 1, rnd((co,), 354, 0.5)
    r = rnd((B, H, W, co), 355, positive=True)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())
    xa, pk = to_act(x), ops.PackedWeights(pw, False, dev())
    kw = dict(res=to_act(r) if res else None)
    if nw == 4 and co > 128:
        pytest.skip("the 4-wave form covers up to 128 output channels")
    _lib.knob("sep_mode", lead)
    _lib.knob("sep_nw", nw)
    out = out_act(B, H, W, co)
    ops.sep_fused(xa, d(dw), pk, d(s1), d(t1), out, **kw)
    sp = ops.SplitAct(B, H, W, co, dev())
    sp.buf.fill_(float("nan"))
    ops.sep_fused(xa, d(dw), pk, d(s1), d(t1), sp, **kw)
    want = ops.to_split32(ops.Act(out.torch().contiguous()))
    torch.cuda.synchronize()
    assert torch.equal(sp.buf.view(torch.int32), want.buf.view(torch.int32))


@pytest.mark.parametrize("B,H,W,ci,co,co2,tpw", [
    (2, 16, 32, 128, 64, 64, 0),        # deconv0_a + residual0_d
    (1, 8, 96, 384, 128, 128, 0),       # deconv1_a + residual1_d: 128 | 128 columns on the same 8 x 32 tile
    (1, 24, 64, 96, 128, 32, 2),        # unequal widths, two tiles per workgroup
    (2, 8, 64, 32, 36, 64, 0),          # channel tail in the separable output, one chunk
    (1, 16, 48, 64, 64, 64, 0),         # W % 32 != 0: the 4-wave two-output form on 8 x 16 tiles
])
@pytest.mark.parametrize("nw", [8, 4])
def test_sep_pipe_dual(B, H, W, ci, co, co2, tpw, nw):
    """emd_sep3x3_dual_f32 through sep_pipe: output 1 == the separable block, output 2 == conv 1x1 + bias + BN + relu6 of the same
    input (denoiser.py:356-359 / :368-371 / :380-383), against the oracle and against the two kernels it replaces."""
    from emdenoise import _lib, ops
    from oracle import tf_ops as T

    x = rnd((B, H, W, ci), 360, positive=True)
    dw = rnd((3, 3, ci, 1), 361, 0.35)
    pw = rnd((1, 1, ci, co), 362, scale=(2.0 / (ci + co)) ** 0.5)
    w2 = rnd((1, 1, ci, co2), 363, scale=(2.0 / (ci + co2)) ** 0.5)
    bias2 = rnd((co2,), 364, 0.2)
    s1, t1 = rnd((co,), 365, 0.2) + 1, rnd((co,), 366, 0.5)
    sb, tb = rnd((co2,), 367, 0.2) + 1, rnd((co2,), 368, 0.5)
    y1 = T.relu6_t(T.conv2d_t(T.depthwise_conv2d_t(t64(x), t64(dw)), t64(pw)) * t64(s1) + t64(t1)).numpy()
    y2 = T.relu6_t(T.conv2d_t(t64(x), t64(w2), t64(bias2)) * t64(sb) + t64(tb)).numpy()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())
    xa = to_act(x, ld=ci + 64, c0=32)
    assert ops.sep_dual_supported(xa, co, co2)
    out, out2 = out_act(B, H, W, co, ld=co + 8, c0=4), out_act(B, H, W, co2, ld=co2 + 12, c0=8)
    p1, p2 = ops.PackedWeights(pw[0], False, dev()), ops.PackedWeights(w2[0], False, dev())
    shift_b = (bias2.astype(np.float64) * sb + tb).astype(np.float32)
    if nw == 4 and (co > 64 or co2 > 64):
        pytest.skip("the 4-wave two-output form: 64 | 64 columns")
    _lib.knob("sep_tpw", tpw)
    _lib.knob("sep_nw", nw)       # 4: 8 x 16 tiles, two workgroups per CU
    ops.sep_dual(xa, d(dw[..., 0]), p1, p2, d(s1), d(t1), out, d(sb), d(shift_b), out2)
    torch.cuda.synchronize()
    g1, g2 = out.torch().cpu().numpy(), out2.torch().cpu().numpy()
    assert not np.isnan(g1).any() and not np.isnan(g2).any()
    assert rel_l2(g1, y1) < TOL_X3 and rel_l2(g2, y2) < TOL_X3
    for o, c in ((out, co), (out2, co2)):
        full = o.buf.cpu().numpy()
        assert np.isnan(full[..., :o.c0]).all() and np.isnan(full[..., o.c0 + c:]).all()
    _lib.knob("sep_pipe", 0)   # the kernels it replaces: same products, same order along K
    want1 = ops.sep_fused(xa, d(dw[..., 0]), p1, d(s1), d(t1), out_act(B, H, W, co))
    want2 = ops.conv1x1(xa, p2, d(sb), d(shift_b), out_act(B, H, W, co2))
    torch.cuda.synchronize()
    assert torch.equal(out.torch(), want1.torch())
    assert rel_l2(g2, want2.torch().cpu().numpy()) < 1e-6


S2_CASES = [
    # B, H, W, Cin, Cout, res, extra, tpw
    (2, 8, 32, 64, 128, False, False, 0),     # one tile per image (4 x 16 output pixels): bottom / right padding in every patch
    (1, 16, 96, 64, 128, True, False, 0),     # left, interior and right-edge tiles, two tile rows; residual (cnn0_strided's widths)
    (2, 24, 64, 32, 128, False, True, 2),     # one chunk, two tiles per workgroup, second affine
    (1, 8, 256, 128, 256, True, False, 4),    # 256 columns (cnn1_strided); the pointer-increment path between interior tiles
    (1, 16, 64, 256, 256, False, False, 0),   # 8 chunks
    (1, 8, 64, 96, 160, True, True, 0),       # channel tail in the 256-column tile
    (1, 8, 32, 64, 36, False, False, 0),      # channel tail in the 128-column tile
]


@pytest.mark.parametrize("B,H,W,ci,co,res,extra,tpw", S2_CASES)
@pytest.mark.parametrize("lead", [0, 1])
def test_sep_pipe_stride2(B, H, W, ci, co, res, extra, tpw, lead):
    """emd_sep3x3_fused_s2_f32 (strided_conv_block(stride=2), machine_learning/denoiser.py:258, :273, :288) against the oracle's TF
    restatement (SAME on even sizes: one pixel of padding after, none before) and bit for bit against the two kernels it replaces
    (emd_dw3x3_f32 stride 2 -> emd_conv1x1_f32: the same depthwise sums, the same products in the same order along K)."""
    from emdenoise import _lib, ops
    from oracle import tf_ops as T

    x = rnd((B, H, W, ci), 380, positive=True)
    dw = rnd((3, 3, ci, 1), 381, 0.35)
    pw = rnd((1, 1, ci, co), 382, scale=(2.0 / (ci + co)) ** 0.5)
    s1, t1 = rnd((co,), 383, 0.2) + 1, rnd((co,), 384, 0.5)
    s2, t2 = rnd((co,), 385, 0.2) + 1, rnd((co,), 386, 0.5)
    Ho, Wo = H // 2, W // 2
    r = rnd((B, Ho, Wo, co), 387, positive=True) if res else None
    y = T.relu6_t(T.conv2d_t(T.depthwise_conv2d_t(t64(x), t64(dw), stride=2), t64(pw)) * t64(s1) + t64(t1))
    if extra:
        y = T.relu6_t(y * t64(s2) + t64(t2))
    if res:
        y = y + t64(r)
    want = y.numpy()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())
    xa = to_act(x, ld=ci + 64, c0=32)
    assert ops.sep_fused_supported(xa, co, 2, 1)
    pk = ops.PackedWeights(pw[0], False, dev())
    kw = dict(scale2=d(s2) if extra else None, shift2=d(t2) if extra else None, res=to_act(r, ld=co + 12, c0=8) if res else None)
    _lib.knob("sep_mode", lead)
    _lib.knob("sep_tpw", tpw)
    out = out_act(B, Ho, Wo, co, ld=co + 8, c0=4)
    ops.sep_fused(xa, d(dw[..., 0]), pk, d(s1), d(t1), out, stride=2, **kw)
    torch.cuda.synchronize()
    got = out.torch().cpu().numpy()
    assert not np.isnan(got).any()
    assert rel_l2(got, want) < TOL_X3
    full = out.buf.cpu().numpy()
    assert np.isnan(full[..., :4]).all() and np.isnan(full[..., 4 + co:]).all()
    tmp = ops.dw3x3(xa, d(dw[..., 0]), out_act(B, Ho, Wo, ci), stride=2)
    two = ops.conv1x1(tmp, pk, d(s1), d(t1), out_act(B, Ho, Wo, co), **kw)
    torch.cuda.synchronize()
    assert torch.equal(out.torch(), two.torch()), "the one-launch form and the two kernels promise the same bits"


@pytest.mark.parametrize("B,H,W,ci,co,tpw", [(2, 8, 32, 32, 64, 0), (1, 16, 96, 64, 128, 0), (1, 8, 256, 128, 256, 4), (1, 24, 64, 96, 36, 2)])
@pytest.mark.parametrize("lead", [0, 1])
def test_sep_pipe_stride2_reflect(B, H, W, ci, co, tpw, lead):
    """emd_sep3x3_fused_s2_reflect_f32: graph G's down-sampling strided_conv_block(stride 2, pad_size = (1, 1))
    (misc_py/gan-infilling-100.py:205-243, :345-352: tf.pad(REFLECT, 1), depthwise VALID stride 2, pointwise, BN x2, leaky relu) against
    the oracle and bit for bit against emd_dw3x3_reflect_f32(stride 2) -> emd_conv1x1_f32."""
    from emdenoise import _lib, ops
    from oracle import tf_ops as T

    x = rnd((B, H, W, ci), 480)
    dw = rnd((3, 3, ci, 1), 481, 0.35)
    pw = rnd((1, 1, ci, co), 482, scale=(2.0 / (ci + co)) ** 0.5)
    s1, t1 = rnd((co,), 483, 0.2) + 1, rnd((co,), 484, 0.5)
    xp = torch.from_numpy(np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)), mode="reflect"))
    d64 = T.depthwise_conv2d_t(xp, t64(dw))[:, 1:-1, 1:-1, :][:, ::2, ::2, :]       # VALID stride 2 on the padded image
    y = T.conv2d_t(d64, t64(pw)) * t64(s1) + t64(t1)
    want = torch.where(y > 0, y, 0.2 * y).numpy()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())
    xa = to_act(x, ld=ci + 64, c0=32)
    pk = ops.PackedWeights(pw[0], False, dev())
    _lib.knob("sep_mode", lead)
    _lib.knob("sep_tpw", tpw)
    out = out_act(B, H // 2, W // 2, co, ld=co + 8, c0=4)
    ops.sep_fused(xa, d(dw[..., 0]), pk, d(s1), d(t1), out, act=ops.ACT_LEAKY, reflect=True, stride=2)
    torch.cuda.synchronize()
    got = out.torch().cpu().numpy()
    assert not np.isnan(got).any()
    assert rel_l2(got, want) < TOL_X3
    full = out.buf.cpu().numpy()
    assert np.isnan(full[..., :4]).all() and np.isnan(full[..., 4 + co:]).all()
    tmp = ops.dw3x3_reflect(xa, d(dw[..., 0]), out_act(B, H // 2, W // 2, ci), stride=2)
    two = ops.conv1x1(tmp, pk, d(s1), d(t1), out_act(B, H // 2, W // 2, co), act=ops.ACT_LEAKY)
    torch.cuda.synchronize()
    assert torch.equal(out.torch(), two.torch())


def test_sep_pipe_stride2_argument_checks():
    from emdenoise import _lib, ops

    d = dev()
    pk = ops.PackedWeights(rnd((1, 64, 128), 390, 0.1), False, d)
    one, dwv = torch.ones(128, device=d), torch.zeros(9, 64, device=d)
    x = ops.Act(torch.zeros(1, 12, 32, 64, device=d))   # H % 8 != 0
    assert not ops.sep_fused_supported(x, 128, 2, 1)
    with pytest.raises(_lib.EmdError):
        ops.sep_fused(x, dwv, pk, one, one, ops.Act.empty(1, 6, 16, 128, d), stride=2)
    x = ops.Act(torch.zeros(1, 8, 48, 64, device=d))    # W % 32 != 0
    assert not ops.sep_fused_supported(x, 128, 2, 1)
    assert not ops.sep_fused_supported(ops.Act(torch.zeros(1, 8, 32, 64, device=d)), 128, 2, 2)   # no dilated stride-2 form


def test_sep_pipe_stride2_full_size_properties():
    """cnn0_strided at BASELINE configs[2]'s batch, [8,512,512,64] -> [8,256,256,128]: image b of the batch == the image alone, bit
    for bit, and exact linearity of the pre-activation in the input."""
    from emdenoise import ops

    B, S, ci, co = 8, 512, 64, 128
    g = torch.Generator(device=dev()).manual_seed(6)
    x = torch.rand(B, S, S, ci, device=dev(), generator=g)
    dw = (torch.rand(9, ci, device=dev(), generator=g) - 0.5)
    pk = ops.PackedWeights(rnd((1, ci, co), 391, 0.1), False, dev())
    s1, t0 = torch.rand(co, device=dev(), generator=g) + 0.5, torch.zeros(co, device=dev())
    f = lambda xx: ops.sep_fused(ops.Act(xx), dw, pk, s1, t0, ops.Act.empty(xx.shape[0], S // 2, S // 2, co, dev()), act=False, stride=2).torch()
    y, y1, y2 = f(x), f(x[5:6].contiguous()), f(2 * x)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    assert torch.equal(y[5:6], y1)
    assert torch.equal(y2, 2 * y)


def test_sep_pipe_full_size_properties():
    """BASELINE configs[2]'s largest fused layer, [8,512,512,128] -> 64 (+ the 64-channel projection): size-independent properties --
    image b of the batch == the image alone, bit for bit (no tile or workgroup boundary depends on the batch), and linearity of the
    pre-activation in the input (act none): f(2x) == 2 f(x) exactly (powers of two commute with every rounding on the path)."""
    from emdenoise import ops

    B, S, ci, co = 8, 512, 128, 64
    g = torch.Generator(device=dev()).manual_seed(5)
    x = torch.rand(B, S, S, ci, device=dev(), generator=g)
    dw = (torch.rand(9, ci, device=dev(), generator=g) - 0.5)
    pk = ops.PackedWeights(rnd((1, ci, co), 371, 0.1), False, dev())
    s1, t0 = torch.rand(co, device=dev(), generator=g) + 0.5, torch.zeros(co, device=dev())
    y = ops.sep_fused(ops.Act(x), dw, pk, s1, t0, ops.Act.empty(B, S, S, co, dev()), act=False).torch()
    y1 = ops.sep_fused(ops.Act(x[5:6].contiguous()), dw, pk, s1, t0, ops.Act.empty(1, S, S, co, dev()), act=False).torch()
    y2 = ops.sep_fused(ops.Act(2 * x), dw, pk, s1, t0, ops.Act.empty(B, S, S, co, dev()), act=False).torch()
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    assert torch.equal(y[5:6], y1)
    assert torch.equal(y2, 2 * y)
