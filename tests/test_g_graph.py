"""Graph G (the in-filling GAN's generator, misc_py/gan-infilling-100.py:133-374), inference.
CPU: variable names / shapes agree between the host module and the oracle, the reference's fixed 1/64 pixel mask,
and the oracle's reflect-pad + VALID convention against an index-level numpy restatement.
GPU: the G-only kernels op by op and the generator end to end against the oracle (float64).  PARITY UNPINNED by the
reference (no tests, vectors or checkpoints; TensorFlow 1.x not installable): the oracle is the restatement.
"""
import numpy as np
import pytest
import torch

from tests.synth_inputs import synthetic_lq


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def lq_batch(B, S, seed=7):
    from emdenoise import gan as GN

    return GN.gen_lq(2.0 * synthetic_lq(B, S, S, seed=seed)[..., 0] - 1.0)[..., None]


# ------------------------------------------------------------------------------------------------ CPU
def test_variable_names_and_counts():
    from emdenoise import gan as GN
    from oracle import gan_graph as GG

    a, b = GN.variable_specs(), GG.variable_specs()
    assert list(a.items()) == list(b.items())
    names = list(a)
    assert names[0] == "GAN/Gen/SeparableConv2d/depthwise_weights" and a[names[0]] == (7, 7, 1, 1)
    assert "GAN/Gen/reg/SeparableConv2d/pointwise_weights" in a          # numbering restarts inside scope "reg" (:353)
    assert a["GAN/Gen/reg/SeparableConv2d_2/pointwise_weights"] == (1, 1, 256, 768)
    assert names[-4:] == ["GAN/Gen/Conv/weights", "GAN/Gen/Conv/biases", "GAN/Gen/Variable", "GAN/Gen/Variable_1"]
    n_sep = sum(1 for n in names if n.endswith("depthwise_weights"))
    assert n_sep == 2 + 3 + 24 + 3 + 9 + 2   # enc0-1, nin down, 8 middle blocks, nin up, 3 local blocks, up + last


def test_gen_lq_mask_is_the_references():
    """np.random.seed(1); select = np.random.random((512,512)) < 1/64  (:1172-1174)."""
    from emdenoise import gan as GN

    img = np.linspace(-1, 1, 512 * 512, dtype=np.float32).reshape(512, 512)
    lq = GN.gen_lq(img)
    state = np.random.RandomState(1)
    select = state.random_sample((512, 512)) < 1.0 / 64
    assert (lq[select] == img[select]).all() and (lq[~select] == -1).all()
    assert abs(select.mean() - 1 / 64) < 2e-3


def test_oracle_reflect_valid_convention():
    """tf.pad(REFLECT,1) + VALID depthwise, stride 2: output i reads rows 2i-1..2i+1 with -1 -> 1 and H -> H-2."""
    from oracle import gan_graph as GG

    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 6, 8, 3))
    w = rng.standard_normal((3, 3, 3, 1))
    got = GG.depthwise_valid_t(GG.reflect_pad_t(torch.from_numpy(x), 1), torch.from_numpy(w), 2).numpy()
    refl = lambda i, n: (-i if i < 0 else (2 * n - 2 - i if i >= n else i))
    ref = np.zeros((1, 3, 4, 3))
    for oy in range(3):
        for ox in range(4):
            for i in range(3):
                for j in range(3):
                    ref[0, oy, ox] += x[0, refl(2 * oy - 1 + i, 6), refl(2 * ox - 1 + j, 8)] * w[i, j, :, 0]
    assert got.shape == ref.shape and np.allclose(got, ref, atol=1e-12)


def test_oracle_generator_runs_and_is_bounded():
    from emdenoise import gan as GN
    from oracle import gan_graph as GG

    y = GG.generator(lq_batch(1, 64), GN.synthetic_weights(), 64, dtype=torch.float32).numpy()
    assert y.shape == (1, 64, 64, 1) and np.isfinite(y).all() and -1.0 <= y.min() and y.max() <= 1.0
    assert y.std() > 0.1  # the tanh is not saturated everywhere


def test_oracle_reproduces_committed_golden():
    import os

    from emdenoise import gan as GN
    from oracle import gan_graph as GG

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g_graph_64.npz"), allow_pickle=False)
    y = GG.generator(z["x"], GN.synthetic_weights(), 64, dtype=torch.float64).numpy()
    assert rel_l2(y, z["y"]) < 1e-6


# ------------------------------------------------------------------------------------------------ GPU
gpu = pytest.mark.gpu


@gpu
def test_generator_matches_committed_golden():
    import os

    from emdenoise import gan as GN

    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g_graph_64.npz"), allow_pickle=False)
    got = GN.GeneratorEngine(GN.synthetic_weights(), dev()).forward(torch.from_numpy(z["x"]).to(dev())).cpu().numpy()
    assert rel_l2(got, z["y"]) < 3e-4


def dev():
    return torch.device("cuda", 0)


@gpu
@pytest.mark.parametrize("B,H,W,Cc,stride", [(2, 16, 16, 64, 1), (1, 9, 13, 128, 1), (2, 16, 16, 64, 2), (1, 7, 10, 768, 2),
                                             (1, 2, 2, 768, 1)])
def test_dw3x3_reflect(B, H, W, Cc, stride):
    from emdenoise import ops
    from oracle import gan_graph as GG
    from tests.test_ops_gpu import out_act, rnd, t64, to_act

    x, w = rnd((B, H, W, Cc), 1), rnd((3, 3, Cc, 1), 2, 0.4)
    ref = GG.depthwise_valid_t(GG.reflect_pad_t(t64(x), 1), t64(w), stride).numpy()
    out = out_act(B, (H - 1) // stride + 1, (W - 1) // stride + 1, Cc)
    ops.dw3x3_reflect(to_act(x, ld=Cc + 8, c0=4), torch.from_numpy(np.ascontiguousarray(w[..., 0])).to(dev()), out, stride=stride)
    torch.cuda.synchronize()
    assert out.torch().shape[1:3] == ref.shape[1:3] and rel_l2(out.torch().cpu().numpy(), ref) < 2e-6


@gpu
def test_first_and_last_layer_kernels():
    from emdenoise import ops
    from oracle import gan_graph as GG
    from oracle import tf_ops as T
    from tests.test_ops_gpu import out_act, rnd, t64, to_act

    B, H, W = 2, 20, 24
    x = rnd((B, H, W, 1), 3)
    dw, pw = rnd((7, 7, 1, 1), 4, 0.2), rnd((1, 1, 1, 32), 5, 0.7)
    s, t = rnd((32,), 6, 0.2) + 1, rnd((32,), 7, 0.3)
    d = GG.depthwise_valid_t(GG.reflect_pad_t(t64(x), 3), t64(dw), 1)
    ref = torch.nn.functional.leaky_relu(T.conv2d_t(d, t64(pw)) * t64(s) + t64(t), 0.2).numpy()
    dd = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())
    out = out_act(B, H, W, 32)
    ops.cin1_k7_reflect(dd(x), dd(dw.reshape(49)), dd(pw.reshape(32) * s), dd(t), out)
    torch.cuda.synchronize()
    assert rel_l2(out.torch().cpu().numpy(), ref) < 2e-6
    # last conv: reflect pad 1 + 3x3 VALID to one channel + bias, then instance norm + tanh
    xin, w = rnd((B, H, W, 32), 8), rnd((3, 3, 32, 1), 9, 0.2)
    wt = t64(w).permute(3, 2, 0, 1).contiguous()
    raw = torch.nn.functional.conv2d(GG.reflect_pad_t(t64(xin), 1).permute(0, 3, 1, 2), wt, t64(np.array([0.3]))).permute(0, 2, 3, 1)
    mu, var = raw.mean(dim=(1, 2), keepdim=True), raw.var(dim=(1, 2), unbiased=False, keepdim=True)
    ref2 = torch.tanh((raw - mu) / torch.sqrt(var + 1e-3)).numpy()
    got_raw = torch.empty((B, H, W, 1), dtype=torch.float32, device=dev())
    ops.conv3x3_cout1_reflect(to_act(xin), dd(w[..., 0].reshape(9, 32)), 0.3, got_raw)
    got = ops.instnorm_tanh(got_raw, torch.empty_like(got_raw))
    torch.cuda.synchronize()
    assert rel_l2(got_raw.cpu().numpy(), raw.numpy()) < 2e-6 and rel_l2(got.cpu().numpy(), ref2) < 5e-6


@gpu
@pytest.mark.parametrize("B,H,W,ci", [(1, 16, 64, 32), (2, 9, 20, 32), (1, 8, 8, 64), (1, 33, 16, 16)])
def test_conv3x3_cout1_reflect_shapes(B, H, W, ci):
    """Both forms of emd_conv3x3_cout1_reflect_f32 (rolling strips where a wave stays inside an image row, per-pixel otherwise),
    ragged strip heights included, against the float64 reflect-pad + VALID conv."""
    from emdenoise import ops
    from oracle import gan_graph as GG
    from tests.test_ops_gpu import rnd, t64, to_act

    xin, w = rnd((B, H, W, ci), 80), rnd((3, 3, ci, 1), 81, 0.2)
    wt = t64(w).permute(3, 2, 0, 1).contiguous()
    raw = torch.nn.functional.conv2d(GG.reflect_pad_t(t64(xin), 1).permute(0, 3, 1, 2), wt, t64(np.array([-0.2]))).permute(0, 2, 3, 1)
    got = torch.empty((B, H, W, 1), dtype=torch.float32, device=dev())
    ops.conv3x3_cout1_reflect(to_act(xin, ld=ci + 8, c0=4), torch.from_numpy(np.ascontiguousarray(w[..., 0].reshape(9, ci))).to(dev()), -0.2, got)
    torch.cuda.synchronize()
    assert rel_l2(got.cpu().numpy(), raw.numpy()) < 2e-6


@gpu
@pytest.mark.parametrize("B,H,W,co,stride", [(2, 20, 24, 32, 2), (1, 16, 16, 64, 1), (1, 9, 7, 32, 2), (1, 12, 40, 16, 2), (1, 6, 6, 128, 2)])
def test_dw_reflect_on_generated_input(B, H, W, co, stride):
    """emd_dw3x3_reflect_gen_f32 == emd_cin1_k7_reflect_f32 (written out) followed by emd_dw3x3_reflect_f32 (bit for bit at stride 2)."""
    from emdenoise import ops
    from tests.test_ops_gpu import out_act, rnd

    dd = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())
    x, w49, a, t, dw = dd(rnd((B, H, W, 1), 90)), dd(rnd((49,), 91, 0.2)), dd(rnd((co,), 92, 0.8)), dd(rnd((co,), 93, 0.3)), dd(rnd((9, co), 94, 0.3))
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    full = ops.cin1_k7_reflect(x, w49, a, t, out_act(B, H, W, co, ld=co, c0=0))
    want = ops.dw3x3_reflect(full, dw, out_act(B, Ho, Wo, co, ld=co, c0=0), stride=stride)
    d4 = ops.cin1_k7_reflect(x, w49, dd(np.array([1, 0, 0, 0])), dd(np.zeros(4)), out_act(B, H, W, 4, ld=4, c0=0), act=False)
    got = ops.dw3x3_reflect_gen(d4, a, t, dw, out_act(B, Ho, Wo, co, ld=co + 4, c0=4), stride=stride)
    torch.cuda.synchronize()
    if stride == 2:
        assert torch.equal(got.torch(), want.torch())
    else:   # the stride-1 reference runs on the rolling kernel, which sums the nine taps row by row
        assert rel_l2(got.torch().cpu().numpy(), want.torch().cpu().numpy()) < 1e-6
    assert not torch.isnan(got.torch()).any()


@gpu
def test_leaky_relu_epilogue():
    from emdenoise import ops
    from oracle import tf_ops as T
    from tests.test_ops_gpu import out_act, rnd, t64, to_act

    x, w = rnd((2, 8, 8, 64), 10), rnd((1, 1, 64, 128), 11, 0.15)
    s, t, r = rnd((128,), 12, 0.2) + 1, rnd((128,), 13, 0.5), rnd((2, 8, 8, 128), 14)
    ref = (torch.nn.functional.leaky_relu(T.conv2d_t(t64(x), t64(w)) * t64(s) + t64(t), 0.2) + t64(r)).numpy()
    out = out_act(2, 8, 8, 128)
    dd = lambda a: torch.from_numpy(a).to(dev())
    ops.conv1x1(to_act(x), ops.PackedWeights(w[0], False, dev()), dd(s), dd(t), out, act=ops.ACT_LEAKY, res=to_act(r))
    torch.cuda.synchronize()
    assert rel_l2(out.torch().cpu().numpy(), ref) < 2e-5
    assert (ref < 0).mean() > 0.1  # the negative side is exercised


@gpu
@pytest.mark.parametrize("S,B", [(64, 2), (128, 1), (256, 1)])
def test_generator_end_to_end(S, B):
    """North-star bar: relative L2 <= 1e-3 against the fp32 semantics of the reference (oracle in float64)."""
    from emdenoise import gan as GN
    from oracle import gan_graph as GG

    w = GN.synthetic_weights()
    x = lq_batch(B, S, seed=20 + S)
    ref = GG.generator(x, w, S, dtype=torch.float64).numpy()
    eng = GN.GeneratorEngine(w, dev())
    got = eng.forward(torch.from_numpy(x).to(dev())).cpu().numpy()
    e = rel_l2(got, ref)
    print(f"G {S}px: rel L2 {e:.2e}")
    assert got.shape == ref.shape and np.isfinite(got).all() and e < 3e-4


# ------------------------------------------------------------------------------------------------ discriminator
def test_discriminator_variable_names():
    from emdenoise import gan as GN
    from oracle import gan_graph as GG

    a, b = GN.discriminator_variable_specs(), GG.discriminator_variable_specs()
    assert list(a.items()) == list(b.items()) and len(a) == 3 * (5 * 8 + 2)
    assert a["GAN/Discr/medium/SeparableConv2d_4/pointwise_weights"] == (1, 1, 256, 512)
    assert a["GAN/Discr/large/fully_connected/weights"] == (512, 1)


def test_multiscale_crops_reflect_and_sizes():
    """get_multiscale_crops (:957-980): reflect pad by 3S/4, crops of S/4, S/2 and 3S/4 (the last resized to S/4)."""
    from oracle import gan_graph as GG

    S = 16
    img = np.arange(S * S, dtype=np.float64).reshape(1, S, S, 1)
    small, medium, large = GG.multiscale_crops(img, ((0, 0), (12, 12), (5, 7)))
    assert small.shape == (1, 4, 4, 1) and medium.shape == (1, 8, 8, 1) and large.shape == (1, 4, 4, 1)
    assert float(small[0, 0, 0, 0]) == img[0, 12, 12, 0]            # padded (0,0) mirrors to (12,12) for pad 12
    assert float(medium[0, 0, 0, 0]) == img[0, 0, 0, 0]             # padded (12,12) is the image origin
    assert np.array_equal(GG.reflect_indices(4, 3), [3, 2, 1, 0, 1, 2, 3, 2, 1, 0])


@gpu
@pytest.mark.parametrize("S,B", [(256, 2), (512, 1)])
def test_discriminator_forward(S, B):
    from emdenoise import gan as GN
    from oracle import gan_graph as GG

    w = GN.discriminator_synthetic_weights()
    img = (2.0 * synthetic_lq(B, S, S, seed=3 + S) - 1.0).astype(np.float32)
    offsets = ((S // 3, S // 5), (S // 7, S // 2), (S // 4 + 3, S // 9))
    ref = GG.discriminator(list(GG.multiscale_crops(img, offsets)), w, dtype=torch.float64)
    x = torch.from_numpy(img).to(dev())
    small, medium, large = GN.multiscale_crops(x, offsets)
    out, layers = GN.DiscriminatorEngine(w, dev()).forward(small, medium, large)
    torch.cuda.synchronize()
    assert len(layers) == 15 == len(ref) - 1
    worst = max(rel_l2(l.cpu().numpy(), r.numpy()) for l, r in zip(layers, ref[1:]) if r.shape[1] > 1)
    print(f"discriminator {S}px: output {out.cpu().numpy()} vs {ref[0].numpy()}, worst feature map rel L2 {worst:.2e}")
    assert worst < 3e-4
    assert np.allclose(out.cpu().numpy(), ref[0].numpy(), rtol=1e-3, atol=1e-4)


@gpu
@pytest.mark.parametrize("B,H,W,ci,co", [(2, 16, 32, 64, 64), (1, 8, 16, 32, 32), (1, 24, 16, 128, 64)])
def test_sep_fused_reflect_leaky(B, H, W, ci, co):
    """One launch: reflect-padded depthwise 3x3 -> pointwise -> affine -> leaky_relu (+ residual)."""
    from emdenoise import ops
    from oracle import gan_graph as GG
    from oracle import tf_ops as T
    from tests.test_ops_gpu import out_act, rnd, t64, to_act

    x, dw, pw = rnd((B, H, W, ci), 30), rnd((3, 3, ci, 1), 31, 0.4), rnd((1, 1, ci, co), 32, 0.15)
    s, t, r = rnd((co,), 33, 0.2) + 1, rnd((co,), 34, 0.3), rnd((B, H, W, co), 35)
    d = GG.depthwise_valid_t(GG.reflect_pad_t(t64(x), 1), t64(dw), 1)
    ref = (torch.nn.functional.leaky_relu(T.conv2d_t(d, t64(pw)) * t64(s) + t64(t), 0.2) + t64(r)).numpy()
    dd = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev())
    assert ops.sep_fused_supported(to_act(x), co, 1, 1)
    out = out_act(B, H, W, co)
    ops.sep_fused(to_act(x), dd(dw[..., 0].reshape(9, ci)), ops.PackedWeights(pw[0], False, dev()), dd(s), dd(t), out,
                  act=ops.ACT_LEAKY, res=to_act(r), reflect=True)
    torch.cuda.synchronize()
    assert rel_l2(out.torch().cpu().numpy(), ref) < 2e-5
