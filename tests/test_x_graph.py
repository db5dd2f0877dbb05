"""Graph X (misc_py/modified_Xception.py:194-654, inference): CPU checks of the two independent graph walks and
GPU parity of emdenoise.xception.XceptionEngine against the oracle (float64)."""
import numpy as np
import pytest
import torch

from tests.synth_inputs import synthetic_lq


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def test_product_and_oracle_agree_on_names_and_sizes():
    from emdenoise import xception as X
    from oracle import xception_graph as XG

    a, b = X.variable_specs(), XG.variable_specs()
    assert list(a.items()) == list(b.items())
    nconv = sum(int(np.prod(s)) for n, s in a.items() if n.rsplit("/", 1)[1] in ("kernel", "depthwise_weights", "pointwise_weights"))
    total = sum(int(np.prod(s)) for s in a.values())
    assert abs(total - 92.6e6) < 0.3e6, total                     # 92.6 M parameters
    assert nconv < total
    assert a["pellet/conv2d/kernel"] == (3, 3, 1, 32)               # entry conv 3x3 stride 2
    assert "pellet/SeparableConv2d_62/pointwise_weights" in a and "pellet/SeparableConv2d_63/pointwise_weights" not in a
    assert "pellet/SeparableConv2d/BatchNorm/gamma" not in a        # scale=False: the SEP norm has beta only
    assert a["pellet/imageLevel/kernel"] == (1, 1, 2048, 256)       # created although its output is discarded
    assert a["pellet/conv2d_6/kernel"] == (1, 1, 1280, 32)          # ASPP concat is 5 x 256
    assert a["pellet/conv2d_transpose_5/kernel"] == (3, 3, 128, 128)


def test_oracle_batch_statistics_make_output_batch_dependent():
    """The SEP norms use BATCH statistics (contrib defaults, :312-314): an image's output depends on its batch."""
    from emdenoise import xception as X
    from oracle import xception_graph as XG

    w = X.synthetic_weights()
    x = synthetic_lq(2, 64, 64, seed=8)
    both = XG.architecture(x, w, 64).numpy()
    alone = XG.architecture(x[:1], w, 64).numpy()
    assert both.min() >= 0.0 and both.max() <= 1.0
    assert rel_l2(alone[0], both[0]) > 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("B,S", [(2, 128), (1, 256)])
def test_engine_matches_oracle(B, S):
    """With random weights, batch-statistics norms and unbounded relu the synthetic X graph amplifies rounding
    noise: the oracle's OWN float32 run differs from its float64 run by 1.6e-4 (128 px) to 4e-3 (256 px) at the
    output, so no float32-class implementation can be held to 1e-3 end to end there.  Checked instead:
      * layer by layer (free running), the encoder + ASPP (first 73 traced tensors) within 3e-4 of float64;
      * end to end within max(1e-3, 25 x the oracle's own float32-vs-float64 error on this very input)
        (split-bf16 carries ~10x the rounding noise of float32)."""
    from emdenoise import xception as X
    from oracle import xception_graph as XG

    w = X.synthetic_weights()
    eng = X.XceptionEngine(w, torch.device("cuda", 0), "bf16x3")
    x = synthetic_lq(B, S, S, seed=400 + S)
    t64, tgpu = [], []
    ref = XG.architecture(x, w, S, dtype=torch.float64, trace=t64).numpy()
    ref32 = XG.architecture(x, w, S, dtype=torch.float32).numpy()
    got = eng.forward(torch.from_numpy(x).cuda(), trace=tgpu).cpu().numpy()
    assert got.shape == ref.shape and got.min() >= 0.0 and got.max() <= 1.0
    assert len(tgpu) == len(t64) - 1            # the oracle also traces the final conv_block
    layer_err = [rel_l2(b, a.numpy()) for a, b in zip(t64, tgpu)]
    noise32, r = rel_l2(ref32, ref), rel_l2(got, ref)
    print(f"X graph B={B} S={S}: encoder+ASPP max layer rel L2 {max(layer_err[:73]):.2e}; end to end {r:.2e} "
          f"(oracle float32 vs float64 on the same input: {noise32:.2e})")
    assert max(layer_err[:73]) < 3e-4
    assert r < max(1e-3, 25 * noise32)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W", [(3, 64, 48), (2, 512, 512), (1, 7, 5)])
def test_single_channel_image_statistics(B, H, W):
    """The dense one-channel reduction (emd_bn_stats_images_f32 with C == 1, ldx == 1: the generator's output instance norm) and
    the batch form over the same buffer, against numpy float64."""
    from emdenoise import ops

    rng = np.random.default_rng(H * W + B)
    x = (rng.standard_normal((B, H, W, 1)) * np.array([0.5, 2.0, 1.0])[:B].reshape(B, 1, 1, 1) + 3.0).astype(np.float32)
    a = ops.Act(torch.from_numpy(x).to(torch.device("cuda", 0)))
    mean, var = ops.bn_batch_stats_images(a)
    x64 = x.astype(np.float64).reshape(B, -1)
    assert np.allclose(mean.cpu().numpy(), x64.mean(1), rtol=1e-6, atol=1e-6)
    assert np.allclose(var.cpu().numpy(), x64.var(1), rtol=2e-6, atol=1e-7)
    m1, v1 = ops.bn_batch_stats(a)
    assert np.allclose(m1.cpu().numpy(), x64.mean(), rtol=1e-6) and np.allclose(v1.cpu().numpy(), x64.var(), rtol=2e-6)
    m2, v2 = ops.bn_batch_stats_images(a)
    assert torch.equal(mean, m2) and torch.equal(var, v2)          # reproducible run to run


@pytest.mark.gpu
@pytest.mark.parametrize("npix_shape,Cc", [((2, 33, 17), 728), ((1, 64, 64), 64), ((3, 5, 7), 2048), ((1, 1, 1), 8)])
def test_bn_batch_stats_and_fold(npix_shape, Cc):
    from emdenoise import ops

    B, H, W = npix_shape
    rng = np.random.default_rng(B * H * W + Cc)
    x = (rng.standard_normal((B, H, W, Cc)) * rng.uniform(0.1, 3, Cc) + rng.uniform(-5, 5, Cc)).astype(np.float32)
    dev = torch.device("cuda", 0)
    buf = torch.full((B, H, W, Cc + 8), float("nan"), dtype=torch.float32, device=dev)
    buf[..., 4:4 + Cc] = torch.from_numpy(x).to(dev)
    a = ops.Act(buf, Cc, 4)
    mean, var = ops.bn_batch_stats(a)
    x64 = x.astype(np.float64).reshape(-1, Cc)
    assert np.allclose(mean.cpu().numpy(), x64.mean(0), rtol=1e-6, atol=1e-6)
    assert np.allclose(var.cpu().numpy(), x64.var(0), rtol=2e-6, atol=1e-7)
    beta = torch.from_numpy(rng.standard_normal(Cc).astype(np.float32)).to(dev)
    scale, shift = ops.bn_fold(mean, var, None, beta, 1e-3)
    out = ops.Act.empty(B, H, W, Cc, dev)
    res = ops.Act(torch.from_numpy(rng.random((B, H, W, Cc)).astype(np.float32)).to(dev))
    ops.affine_act(a, scale, shift, out, act=ops.ACT_RELU, res=res)
    ref = np.maximum((x64 - x64.mean(0)) / np.sqrt(x64.var(0) + 1e-3) + beta.cpu().numpy(), 0).reshape(B, H, W, Cc) + res.torch().cpu().numpy()
    assert rel_l2(out.torch().cpu().numpy(), ref) < 1e-5   # float32 mean, var, rsqrt and fma on the device


@pytest.mark.gpu
def test_deferred_norm_relu_is_the_same_arithmetic():
    """Without a trace the engine leaves a separable block's norm + relu to the next block's depthwise kernel wherever that
    is the only consumer (emd_dw3x3_pre*_f32); with a trace every block output is materialised.  Same bits either way."""
    from emdenoise import xception as X

    eng = X.XceptionEngine(X.synthetic_weights(), torch.device("cuda", 0), "bf16x3")
    x = torch.from_numpy(synthetic_lq(2, 128, 128, seed=77)).cuda()
    fused = eng.forward(x).cpu().numpy()
    plain = eng.forward(x, trace=[]).cpu().numpy()
    assert np.array_equal(fused, plain)
