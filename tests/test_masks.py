"""CPU checks of gan.spiral_mask -- a mask generator of OUR OWN for BASELINE configs[4]'s "spiral-scan masks" (the reference's mask is the
fixed Bernoulli(1/64) field of gen_lq, misc_py/gan-infilling-100.py:1172-1175, which stays the default of gan.gen_lq)."""
import numpy as np
import pytest


@pytest.mark.parametrize("S", [128, 256, 512])
def test_spiral_mask_density_shape_and_path(S):
    from emdenoise import gan as GN

    m = GN.spiral_mask(S)
    assert m.shape == (S, S) and m.dtype == np.bool_
    assert abs(m.sum() / (S * S / 64.0) - 1.0) < 0.01          # 1/64 of the pixels, as the reference's frac
    assert np.array_equal(m, GN.spiral_mask(S))                 # deterministic
    # a scan path, not a point cloud: (almost) every measured pixel touches another one
    p = np.pad(m, 1)
    nb = sum(p[1 + dy:1 + dy + S, 1 + dx:1 + dx + S] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dy, dx) != (0, 0))
    assert int((m & (nb == 0)).sum()) <= 2
    # it winds: the measured radii are spread from the centre to the corners
    yy, xx = np.nonzero(m)
    r = np.hypot(yy - (S - 1) / 2.0, xx - (S - 1) / 2.0)
    assert r.min() < 1.0 and r.max() > 0.55 * S   # (the corners are at 0.707 S; the last turn leaves the image before them)
    hist, _ = np.histogram(r, bins=8, range=(0, S / 2.0))
    assert (hist > 0).all()


def test_gen_lq_with_a_spiral_mask():
    from emdenoise import gan as GN

    img = np.random.default_rng(0).uniform(-1, 1, (2, 64, 64)).astype(np.float32)
    m = GN.spiral_mask(64)
    lq = GN.gen_lq(img, select=m)
    assert np.array_equal(lq[:, m], img[:, m]) and (lq[:, ~m] == -1).all()
