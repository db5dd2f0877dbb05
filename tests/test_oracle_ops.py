"""CPU tests that pin the oracle's TF-op restatement (oracle/tf_ops.py).

The reference holds no golden vectors for this path (parity unpinned), so the pins
are: (1) the analytic known-answer tests SURVEY.md 8c lists, (2) agreement of the two independent
implementations of every op (PyTorch-CPU vs plain numpy index formulas), (3) identities that
define the TF semantics (conv2d_transpose == gradient of the SAME stride-2 conv, computed by
autograd), (4) the committed golden vectors under tests/golden/.
"""
import numpy as np
import pytest
import torch

from oracle import tf_ops as T


def rnd(shape, seed, dtype=np.float64):
    return np.random.default_rng(seed).standard_normal(shape).astype(dtype)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


# ---------------------------------------------------------------- SAME padding arithmetic
@pytest.mark.parametrize("n,k,s,r,expect", [
    (512, 3, 1, 1, (512, 1, 1)),
    (512, 3, 2, 1, (256, 0, 1)),   # stride 2 on an even size: 0 before, 1 after
    (32, 3, 1, 6, (32, 6, 6)),
    (32, 3, 1, 18, (32, 18, 18)),
    (512, 1, 2, 1, (256, 0, 0)),   # 1x1 stride 2 samples x[0::2]
    (7, 3, 2, 1, (4, 1, 1)),       # odd size: symmetric
    (5, 3, 1, 1, (5, 1, 1)),
])
def test_same_pads(n, k, s, r, expect):
    assert T.same_pads(n, k, s, r) == expect


# ---------------------------------------------------------------- torch vs numpy, per op
@pytest.mark.parametrize("H,W,C,stride,rate", [
    (8, 8, 3, 1, 1), (8, 10, 5, 2, 1), (7, 9, 4, 2, 1), (12, 12, 2, 1, 6), (6, 6, 2, 1, 18), (9, 7, 1, 1, 1),
])
def test_depthwise_torch_vs_numpy(H, W, C, stride, rate):
    x = rnd((2, H, W, C), 1)
    w = rnd((3, 3, C, 1), 2)
    a = T.depthwise_conv2d_t(t(x), t(w), stride, rate).numpy()
    b = T.depthwise_conv2d_np(x, w, stride, rate)
    assert a.shape == b.shape
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("H,W,ci,co,k,stride", [
    (8, 8, 3, 5, 1, 1), (8, 8, 3, 5, 1, 2), (9, 7, 4, 2, 3, 1), (8, 6, 2, 3, 3, 2), (5, 5, 6, 1, 3, 1),
])
def test_conv2d_torch_vs_numpy(H, W, ci, co, k, stride):
    x = rnd((2, H, W, ci), 3)
    w = rnd((k, k, ci, co), 4)
    b = rnd((co,), 5)
    a = T.conv2d_t(t(x), t(w), t(b), stride).numpy()
    n = T.conv2d_np(x, w, b, stride)
    assert a.shape == n.shape
    np.testing.assert_allclose(a, n, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("H,W,ci,co", [(4, 4, 3, 2), (5, 7, 2, 4), (1, 1, 1, 1)])
def test_conv2d_transpose_torch_vs_numpy(H, W, ci, co):
    x = rnd((2, H, W, ci), 6)
    w = rnd((3, 3, co, ci), 7)
    b = rnd((co,), 8)
    a = T.conv2d_transpose_s2_t(t(x), t(w), t(b)).numpy()
    n = T.conv2d_transpose_s2_np(x, w, b)
    assert a.shape == (2, 2 * H, 2 * W, co) == n.shape
    np.testing.assert_allclose(a, n, rtol=1e-12, atol=1e-12)


def test_conv2d_transpose_is_gradient_of_same_stride2_conv():
    """slim.conv2d_transpose(k3,s2,'same') is by definition the input-gradient of the SAME
    stride-2 3x3 convolution (2N -> N).  Check the restatement against autograd."""
    N, ci, co = 5, 3, 4
    w_t = rnd((3, 3, co, ci), 9)            # conv2d_transpose layout [kh,kw,Cout,Cin]
    g = rnd((2, N, N, ci), 10)              # "x" of the transposed conv = dL/d(conv output)
    inp = torch.zeros(2, 2 * N, 2 * N, co, dtype=torch.float64, requires_grad=True)
    # forward conv maps co channels -> ci channels with weights [kh,kw,in=co,out=ci] == w_t
    out = T.conv2d_t(inp, t(w_t), None, stride=2)
    assert out.shape == (2, N, N, ci)
    out.backward(t(g))
    ours = T.conv2d_transpose_s2_t(t(g), t(w_t)).numpy()
    np.testing.assert_allclose(ours, inp.grad.numpy(), rtol=1e-12, atol=1e-12)
    # PyTorch's habitual padding=1,output_padding=1 is a DIFFERENT operator
    wt = t(w_t).permute(3, 2, 0, 1)
    habit = torch.nn.functional.conv_transpose2d(t(g).permute(0, 3, 1, 2), wt, stride=2, padding=1,
                                                 output_padding=1).permute(0, 2, 3, 1).numpy()
    assert np.abs(habit - ours).max() > 1e-3


@pytest.mark.parametrize("H,W,oh,ow", [(4, 4, 16, 16), (4, 4, 4, 4), (3, 5, 12, 20), (8, 8, 5, 3)])
def test_resize_torch_vs_numpy(H, W, oh, ow):
    x = rnd((2, H, W, 3), 11)
    a = T.resize_bilinear_legacy_t(t(x), oh, ow).numpy()
    n = T.resize_bilinear_legacy_np(x, oh, ow)
    np.testing.assert_allclose(a, n, rtol=1e-12, atol=1e-12)


def test_resize_known_answers():
    # same size is the identity (denoiser.py:199 resizes the 32x32 input to [32,32])
    x = rnd((1, 4, 4, 2), 12)
    np.testing.assert_array_equal(T.resize_bilinear_legacy_t(t(x), 4, 4).numpy(), x)
    # x4 legacy sampling of a ramp: src = dst/4, clamped at the last sample (no half-pixel shift)
    ramp = np.arange(4, dtype=np.float64).reshape(1, 1, 4, 1)
    y = T.resize_bilinear_legacy_t(t(ramp), 1, 16).numpy().reshape(-1)
    expect = np.minimum(np.arange(16) / 4.0, 3.0)
    np.testing.assert_allclose(y, expect, rtol=0, atol=1e-12)


def test_reflect_pad():
    x = np.arange(12, dtype=np.float64).reshape(1, 3, 4, 1)
    a = T.reflect_pad_t(t(x), 1).numpy()
    n = T.reflect_pad_np(x, 1)
    np.testing.assert_array_equal(a, n)
    assert a[0, 0, 0, 0] == x[0, 1, 1, 0]          # mirror WITHOUT repeating the border sample
    assert a[0, -1, -1, 0] == x[0, 1, 2, 0]
    for i, n_, e in [(-1, 5, 1), (5, 5, 3), (2, 5, 2), (-2, 5, 2), (6, 5, 2)]:
        assert T.reflect_index(i, n_) == e


def test_batch_norm_known_answer():
    """KAT #2: at TF initial values (gamma 1, beta 0, mean 0, var 1) BN is x/sqrt(1.001)."""
    x = rnd((2, 3, 3, 4), 13)
    one, zero = np.ones(4), np.zeros(4)
    y = T.batch_norm_inference_t(t(x), t(one), t(zero), t(zero), t(one)).numpy()
    np.testing.assert_allclose(y, x / np.sqrt(1.001), rtol=1e-14)
    g, b, m, v = rnd((4,), 14), rnd((4,), 15), rnd((4,), 16), np.abs(rnd((4,), 17)) + 0.1
    np.testing.assert_allclose(T.batch_norm_inference_t(t(x), t(g), t(b), t(m), t(v)).numpy(),
                               T.batch_norm_inference_np(x, g, b, m, v), rtol=1e-12, atol=1e-12)


def test_relu6():
    x = np.array([-1.0, 0.0, 3.0, 6.0, 7.5])
    np.testing.assert_array_equal(T.relu6_t(t(x)).numpy(), [0, 0, 3, 6, 6])
    np.testing.assert_array_equal(T.relu6_np(x), [0, 0, 3, 6, 6])


def test_depthwise_stride2_samples_from_index_zero():
    """TF SAME with stride 2 on an even size pads 0 before / 1 after: output (i,j) is centred on
    input (2i+1, 2j+1)... i.e. its window starts at input row 2i (not 2i-1)."""
    x = np.zeros((1, 6, 6, 1))
    x[0, 0, 0, 0] = 1.0
    w = np.zeros((3, 3, 1, 1))
    w[0, 0, 0, 0] = 1.0   # top-left tap
    y = T.depthwise_conv2d_t(t(x), t(w), stride=2).numpy()
    assert y[0, 0, 0, 0] == 1.0  # window of output (0,0) starts at input (0,0)


@pytest.mark.parametrize("H,W,rate,stride", [(12, 12, 6, 1), (8, 10, 18, 1), (9, 9, 2, 1), (8, 8, 1, 2)])
def test_dense_dilated_conv_torch_vs_numpy(H, W, rate, stride):
    """tf.layers.conv2d(kernel_size=3, dilation_rate=r, 'same') -- the training twin's ASPP branches
    (misc_py/denoiser-multi-gpu.py:306-328)."""
    x = rnd((2, H, W, 3), 40)
    w = rnd((3, 3, 3, 4), 41)
    b = rnd((4,), 42)
    a = T.conv2d_t(t(x), t(w), t(b), stride=stride, rate=rate).numpy()
    n = T.conv2d_np(x, w, b, stride=stride, rate=rate)
    np.testing.assert_allclose(a, n, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("H,W", [(4, 4), (6, 8), (5, 7), (1, 1)])
def test_avg_pool_same(H, W):
    x = rnd((2, H, W, 3), 43)
    a = T.avg_pool2x2_same_t(t(x)).numpy()
    n = T.avg_pool2x2_same_np(x)
    assert a.shape == (2, -(-H // 2), -(-W // 2), 3)
    np.testing.assert_allclose(a, n, rtol=1e-13, atol=1e-13)
    # known answer: the window that hangs over the edge averages only the samples inside the image
    if H == 5:
        np.testing.assert_allclose(a[:, 2, 0], x[:, 4, 0:2].mean(axis=1))
