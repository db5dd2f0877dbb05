"""CPU tests of the graph-D oracle and of the host logic around it (no GPU, no compute in the library)."""
import os

import numpy as np
import torch

import emdenoise
from emdenoise import denoiser as D
from oracle import denoiser_graph as G
from oracle import tf_ops as T
from tests.synth_inputs import synthetic_lq

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_product_and_oracle_agree_on_tf_variable_names():
    """Two independent walks of architecture() (denoiser.py:248-398) must produce the same TensorFlow
    variable names and shapes, in the same creation order."""
    a, b = D.variable_specs(), G.variable_specs()
    assert list(a.items()) == list(b.items())
    assert len(a) == 658
    nconv = sum(int(np.prod(s)) for n, s in a.items() if n.rsplit("/", 1)[1] in ("weights", "depthwise_weights", "pointwise_weights"))
    assert nconv == 25_267_105                      # 25.27 M conv parameters
    assert "nn/SeparableConv2d_56/pointwise_weights" in a and "nn/SeparableConv2d_57/pointwise_weights" not in a  # 57 SEP
    assert a["nn/Conv_9/weights"] == (3, 3, 64, 1)  # the final layer is 3x3, not 1x1 (denoiser.py:387)
    assert a["nn/Conv_5/weights"] == (1, 1, 3640, 256)  # ASPP concat is 5 x 728 wide
    assert a["nn/Conv2d_transpose_1/weights"] == (3, 3, 128, 128)


def test_shipped_calibration_covers_every_moving_statistic():
    w = emdenoise.synthetic_weights()
    w0 = emdenoise.synthetic_weights(bn="tf_init")
    moved = [n for n in w if n.endswith(("moving_mean", "moving_variance"))]
    assert len(moved) == 260
    assert all(not np.array_equal(w[n], w0[n]) for n in moved)
    assert all(np.array_equal(w[n], w0[n]) for n in w if n not in moved)      # everything else is seed-only
    assert all((w[n] > 0).all() for n in moved if n.endswith("variance"))


def test_oracle_reproduces_committed_golden():
    z = np.load(os.path.join(GOLDEN, "d_graph_64.npz"), allow_pickle=False)
    w = emdenoise.synthetic_weights()
    y32 = G.architecture(z["x"], w, 64, dtype=torch.float32).numpy()
    rel = np.linalg.norm(y32.astype(np.float64) - z["y"]) / np.linalg.norm(z["y"])
    assert rel < 5e-5                                # oracle float32 vs the stored float64 run
    assert z["y"].min() >= 0 and z["y"].max() <= 6   # the last op is a relu6 (batch_then_activ, :387)


def test_bn_folding_matches_sequential_batch_norms():
    """Host logic: two consecutive inference BNs (+ bias) folded to one affine (denoiser.py:123,:134)."""
    w = emdenoise.synthetic_weights()
    L = D.declare_layers()["cnn1"]
    s, t = D._fold(w, L)
    x = np.random.default_rng(0).standard_normal((5, L.cout))
    y = torch.from_numpy(x)
    for scope in L.bn:
        y = T.batch_norm_inference_t(y, *[torch.from_numpy(w[f"{scope}/{k}"].astype(np.float64)) for k in ("gamma", "beta", "moving_mean", "moving_variance")])
    np.testing.assert_allclose(x * s + t, y.numpy(), rtol=2e-6, atol=2e-6)


def test_scale0to1_known_answer():
    """KAT #3: a constant image maps to 0.5 (denoiser.py:690-691)."""
    assert (D.scale0to1(np.full((4, 4), 3.0)) == 0.5).all()
    np.testing.assert_allclose(D.scale0to1(np.array([[1.0, 3.0], [2.0, 5.0]])), [[0, 0.5], [0.25, 1.0]])


def test_host_resize_is_identity_at_512_and_bilinear_otherwise():
    x = synthetic_lq(1, 512, 512, seed=1)[0, :, :, 0]
    np.testing.assert_array_equal(D._resize_bilinear_host(x, (512, 512)), x)
    ramp = np.tile(np.arange(4, dtype=np.float32), (4, 1))
    up = D._resize_bilinear_host(ramp, (8, 8))
    np.testing.assert_allclose(up[0], [0, 0.25, 0.75, 1.25, 1.75, 2.25, 2.75, 3.0], atol=1e-6)  # half-pixel centres


def test_training_twin_names_and_sizes():
    """misc_py/denoiser-multi-gpu.py:200-540 (phase=False): tf.layers scopes, named ASPP convs, 38.50 M conv
    parameters."""
    a, b = D.variable_specs("Dprime"), G.variable_specs(variant="Dprime")
    assert list(a.items()) == list(b.items())
    nconv = sum(int(np.prod(s)) for n, s in a.items() if n.rsplit("/", 1)[1] in ("kernel", "depthwise_weights", "pointwise_weights"))
    assert nconv == 38_497_049
    assert a["nn/lowRate/kernel"] == (3, 3, 728, 728) and a["nn/pellet/kernel"] == (1, 1, 3640, 256)
    assert a["nn/conv2d/kernel"] == (1, 1, 1, 128)              # residual0 is the first unnamed tf.layers conv
    assert a["nn/conv2d_transpose_1/kernel"] == (3, 3, 128, 128)
    assert "nn/imageLevel/bias" in a and "nn/Conv/weights" not in a
    w = emdenoise.synthetic_weights(variant="Dprime")
    x = synthetic_lq(1, 32, 32, seed=5)
    y = G.architecture(x, w, 32, variant="Dprime").numpy()
    assert y.min() >= 0.0 and y.max() <= 1.0                     # in-graph clip (:534-538)
