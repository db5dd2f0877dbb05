"""Graph S: the small separable autoencoder of misc_py/apply_autoencoders.py.
CPU: the oracle against the committed golden vector, variable names (product and oracle generate them independently),
the class's host-side steps.  GPU: the HIP launch sequence against the oracle (float64), layer by layer and end to end, for
encoding_features 1 / 4 / 16, the golden vector, whole-image tiling.  Tolerance: relative L2 1e-3 (north_star); the graph
has six batch-statistics norms, measured 1-3e-5 per layer.
"""
import os

import numpy as np
import pytest
import torch

from tests.synth_inputs import synthetic_lq

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "s_graph_160.npz")


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("enc", [1, 4, 16])
def test_variable_names_agree(enc):
    from emdenoise import autoencoder as AE
    from oracle import autoencoder_graph as AG

    a, b = AE.variable_specs(enc), AG.variable_specs(enc)
    assert list(a.items()) == list(b.items())
    assert list(a)[:3] == ["SeparableConv2d/depthwise_weights", "SeparableConv2d/pointwise_weights", "SeparableConv2d/BatchNorm/beta"]
    assert "BatchNorm_5/gamma" in a and "BatchNorm_6/gamma" not in a and list(a)[-1] == "Conv/weights"
    assert a["SeparableConv2d_3/pointwise_weights"] == (1, 1, 256, enc) and a["Conv2d_transpose/weights"] == (3, 3, 256, enc)


def test_oracle_reproduces_committed_golden():
    from emdenoise import autoencoder as AE
    from oracle import autoencoder_graph as AG

    z = np.load(GOLD, allow_pickle=False)
    y = AG.architecture(z["x"], AE.synthetic_weights(16), 16, dtype=torch.float64).numpy()
    assert rel_l2(y, z["y"]) < 1e-6
    # per-image statistics: an image's output does not depend on its batch mates
    y1 = AG.architecture(z["x"][1:], AE.synthetic_weights(16), 16, dtype=torch.float64).numpy()
    assert np.array_equal(y1[0], y[1])


def test_batch_norm_at_init_is_standardisation():
    """KAT: with gamma 1 / beta 0 the norm of a constant-variance map is (x - mean)/sqrt(var + 1e-3) per image and channel."""
    from oracle import autoencoder_graph as AG

    x = torch.from_numpy(np.random.default_rng(0).standard_normal((1, 6, 5, 3)))
    y = AG._bn_batch(x, torch.ones(3, dtype=torch.float64), torch.zeros(3, dtype=torch.float64)).numpy()
    xn = x.numpy()
    want = (xn - xn.mean(axis=(0, 1, 2))) / np.sqrt(xn.var(axis=(0, 1, 2)) + 1e-3)
    assert np.allclose(y, want, atol=1e-12)


def test_preprocess_matches_reference_steps():
    from emdenoise import autoencoder as AE

    img = np.arange(12, dtype=np.float32).reshape(3, 4)
    img[0, 0] = np.nan
    img[1, 1] = np.inf
    cls = AE.Micrograph_Autoencoder.__new__(AE.Micrograph_Autoencoder)   # host-side method only: no GPU
    out = cls.preprocess(img, pad_width=2)
    ref = np.array(img, copy=True)
    ref[np.isnan(ref)] = 0.0
    ref[np.isinf(ref)] = 0.0
    ref = (ref - ref.min()) / (ref.max() - ref.min())
    ref = ref / ref.mean()
    ref = np.pad(ref, 2, mode="reflect")
    assert out.shape == (7, 8, 1) and np.allclose(out[..., 0], ref, atol=1e-6)
    assert np.all(AE.scale0to1(np.full((3, 3), 7.0)) == 0.5)


@pytest.mark.gpu
@pytest.mark.parametrize("enc,S,B", [(16, 160, 2), (4, 64, 3), (1, 32, 2)])
def test_engine_matches_oracle(enc, S, B):
    from emdenoise import autoencoder as AE
    from oracle import autoencoder_graph as AG

    w = AE.synthetic_weights(enc)
    x = synthetic_lq(B, S, S, seed=900 + S)
    x = (x / x.mean(axis=(1, 2, 3), keepdims=True)).astype(np.float32)
    eng = AE.AutoencoderEngine(w, torch.device("cuda", 0), enc)
    t64, tgpu = [], []
    ref = AG.architecture(x, w, enc, dtype=torch.float64, trace=t64).numpy()
    got = eng.forward(torch.from_numpy(x).cuda(), trace=tgpu).cpu().numpy()
    assert len(t64) == len(tgpu) == 7
    for k, (a, b) in enumerate(zip(tgpu, t64)):
        assert rel_l2(a[..., : b.shape[-1]], b) < 3e-4, f"layer {k}"
        assert not a[..., b.shape[-1]:].any(), "zero-padded channels must stay zero"
    assert rel_l2(got, ref) < 1e-3
    # per-image statistics on the device too: image 1 alone gives the same bits as image 1 in the batch
    alone = eng.forward(torch.from_numpy(x[1:2]).cuda()).cpu().numpy()
    assert np.array_equal(alone[0], got[1])


@pytest.mark.gpu
def test_engine_matches_committed_golden():
    from emdenoise import autoencoder as AE

    z = np.load(GOLD, allow_pickle=False)
    eng = AE.AutoencoderEngine(AE.synthetic_weights(16), torch.device("cuda", 0), 16)
    got = eng.forward(torch.from_numpy(z["x"]).cuda()).cpu().numpy()
    assert rel_l2(got, z["y"]) < 1e-3


@pytest.mark.gpu
def test_class_surface_crop_and_whole_image():
    from emdenoise import autoencoder as AE

    nn = AE.Micrograph_Autoencoder(checkpoint_loc=None, visible_cuda=None, encoding_features=16)
    rng = np.random.default_rng(3)
    crop = (rng.random((160, 160)) * 50 + 10).astype(np.float32)
    out = nn.denoise_crop(crop)
    assert out.shape == (160, 160) and np.isfinite(out).all()
    # the scaling of denoise_crop (:364-381) is an affine map: denoise_crop(a*x + b) = a*denoise_crop(x) + b up to the
    # preprocess step's min-max (invariant under it), so the result follows the input's offset and scale
    out2 = nn.denoise_crop(3.0 * crop + 7.0)
    assert rel_l2(out2, 3.0 * out + 7.0) < 1e-4
    img = (rng.random((230, 301)) + 0.5).astype(np.float32)
    den = nn.denoise(img, overlap=25, used_overlap=1)
    assert den.shape == img.shape and np.isfinite(den).all()
    # a 110 x 110 image (= cropsize - 2*overlap) is exactly one crop: the whole-image path equals the crop path on the padded image
    small = img[:110, :110]
    one = nn.denoise(small, overlap=25, used_overlap=25)
    padded = nn.preprocess(small, pad_width=25)[..., 0]
    ref = nn.denoise_crop(padded, preprocess=False)[25:135, 25:135]
    assert rel_l2(one, ref) < 1e-5
