"""GPU parity tests of the device input functions (csrc/input_ops.hip) through the C ABI:
emd_philox4x32_u32 / emd_get_scale_f32 / emd_d4_choices_i32 / emd_flip_rotate_f32 / emd_minmax_images_f32 /
emd_scale0to1_images_f32 / emd_gen_lq_f32 against oracle/input_ops_ref.py and numpy (misc_py/denoiser-multi-gpu.py:783-870).
Bars: bit-exact for the generator, the D4 kernel, min-max and scale0to1, and for lq given the counts; 1e-6 relative for the
float32 scale (device log vs libm) and for truth (the kernel reduces the means in float64, numpy pairwise in float32);
distribution tests (moments per intensity bin, two-sample KS against numpy.random.Generator.poisson) for the Poisson draws."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import input_ops_ref as R  # noqa: E402


@pytest.fixture(scope="module")
def rp():
    from emdenoise import input_pipeline as ip

    assert torch.cuda.is_available()
    return ip.DeviceRecordParser(torch.device("cuda", 0), seed=0xC0FFEE1234, first_image=0)


def test_philox_raw_bit_exact(rp):
    import ctypes as C

    for n4, seed, c0 in ((1000, 0, 0), (4097, 0xFFFFFFFFFFFFFFFF, 2 ** 32 - 5), (256, 0x299F31D0A4093822, 12345678901234)):
        out = torch.empty(n4 * 4, dtype=torch.int32, device=rp.device)
        rp._lib.check(rp.lib.emd_philox4x32_u32(C.c_void_p(out.data_ptr()), n4, seed, c0, rp._lib.stream_ptr()))
        got = out.cpu().numpy().view(np.uint32).reshape(n4, 4)
        np.testing.assert_array_equal(got, R.raw(n4, seed, c0))


def test_scale_and_choices(rp):
    for first in (0, 1000003):
        rp.next_image = first
        s = rp.get_scale(513).cpu().numpy()
        np.testing.assert_allclose(s, R.get_scale(513, rp.seed, first), rtol=1e-6)
        _, ch = rp.flip_rotate(torch.zeros(513, 8, 8, device=rp.device))
        np.testing.assert_array_equal(ch.cpu().numpy(), R.d4_choices(513, rp.seed, first))
    rp.next_image = 0


@pytest.mark.parametrize("S", [8, 64, 100, 512])
def test_flip_rotate_bit_exact(rp, S):
    rng = np.random.default_rng(S)
    x = rng.standard_normal((16, S, S)).astype(np.float32)
    ch = np.arange(16, dtype=np.int32) % 8
    y, _ = rp.flip_rotate(torch.from_numpy(x).to(rp.device), torch.from_numpy(ch).to(rp.device))
    got = y.cpu().numpy()
    for b in range(16):
        np.testing.assert_array_equal(got[b], R.flip_rotate(x[b], ch[b]))
    # preprocess()'s NaN / Inf -> 0.5 (:855-856), applied to the source pixels
    x[0, 1, 2], x[1, 3, 0], x[2, S - 1, S - 1] = np.nan, np.inf, -np.inf
    y, _ = rp.flip_rotate(torch.from_numpy(x).to(rp.device), torch.from_numpy(ch).to(rp.device), fix_nonfinite=True)
    got = y.cpu().numpy()
    for b in range(16):
        ref = x[b].copy()
        ref[~np.isfinite(ref)] = 0.5
        np.testing.assert_array_equal(got[b], R.flip_rotate(ref, ch[b]))


def test_scale0to1_bit_exact_and_constant_image(rp):
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((5, 200, 200, 1)) * 7 + 3).astype(np.float32)
    x[4] = 2.5                                                     # constant image -> 0.5 (:823-824)
    got = rp.scale0to1(torch.from_numpy(x).to(rp.device)).cpu().numpy()
    for b in range(5):
        np.testing.assert_array_equal(got[b], R.scale0to1(x[b]))
    assert (got[4] == 0.5).all()


def test_gen_lq_tail_is_exact_given_the_counts(rp):
    rng = np.random.default_rng(9)
    img = np.stack([R.scale0to1(rng.random((128, 128)).astype(np.float32) ** 2) for _ in range(4)])[..., None]
    img[3] = 0.0                                                   # an all-zero image: every count 0 -> lq 0 (int64 fill(0.5) stores 0)
    scale = np.array([30.0, 110.0, 700.0, 50.0], np.float32)
    lq, truth, counts = rp.gen_lq(torch.from_numpy(img).to(rp.device), torch.from_numpy(scale).to(rp.device), want_counts=True)
    lq, truth, counts = lq.cpu().numpy(), truth.cpu().numpy(), counts.cpu().numpy()
    assert counts.min() >= 0
    for b in range(3):
        ref_lq, ref_truth = R.lq_and_truth_from_counts(counts[b].astype(np.int64), img[b])
        np.testing.assert_array_equal(lq[b], ref_lq)               # float64 division rounded to float32, as numpy does it
        np.testing.assert_allclose(truth[b], ref_truth, rtol=2e-6)
        assert np.isclose(truth[b].mean(), lq[b].mean(), rtol=1e-5)   # truth carries the LQ mean (:868)
    assert (counts[3] == 0).all() and (lq[3] == 0.0).all()
    assert np.isnan(truth[3]).all()                                # (0 / 0) * img, as the reference computes it for an all-zero image


def test_poisson_moments_per_intensity_bin(rp):
    """Mean and variance of the counts per rate, over rates on both sides of the sampler switch at 10 and up to the
    reference's largest plausible rate (scale = 25 + Exp(75) reaches several hundred)."""
    rates = np.array([0.0, 0.05, 0.7, 3.0, 9.99, 10.0, 10.01, 31.0, 100.0, 457.0, 1800.0])
    n = 1 << 18
    img = np.repeat((rates / rates.max()).astype(np.float32)[:, None], n, axis=1)
    scale = np.full(len(rates), rates.max(), np.float32)
    _, _, counts = rp.gen_lq(torch.from_numpy(img).to(rp.device), torch.from_numpy(scale).to(rp.device), want_counts=True)
    c = counts.cpu().numpy().astype(np.float64)
    for k, lam in enumerate((img[:, 0].astype(np.float64) * float(scale[0]))):
        m, v = c[k].mean(), c[k].var()
        se_m = np.sqrt(max(lam, 1e-12) / n)
        assert abs(m - lam) < 5 * se_m + 1e-12, (lam, m)
        se_v = np.sqrt((lam + 2 * lam * lam) / n) if lam > 0 else 1e-12      # Var(sample variance) ~ (mu4 - sigma^4)/n
        assert abs(v - lam) < 6 * se_v + 1e-12, (lam, v)
    assert (c[0] == 0).all()


def test_poisson_ks_against_numpy(rp):
    from scipy import stats

    n = 200000
    for lam in (2.5, 25.0, 250.0):
        img = np.full((1, n), 0.5, np.float32)
        _, _, counts = rp.gen_lq(torch.from_numpy(img).to(rp.device), torch.tensor([2 * lam], dtype=torch.float32, device=rp.device),
                                 want_counts=True)
        a = counts.cpu().numpy().ravel()
        b = np.random.default_rng(int(lam * 10)).poisson(lam, n)
        ks = stats.ks_2samp(a, b)
        assert ks.statistic < 0.01, (lam, ks)
        # and against the exact pmf: chi-square over the bulk of the support
        lo, hi = int(stats.poisson.ppf(1e-4, lam)), int(stats.poisson.ppf(1 - 1e-4, lam))
        obs = np.bincount(np.clip(a, lo, hi) - lo, minlength=hi - lo + 1).astype(np.float64)
        pmf = stats.poisson.pmf(np.arange(lo, hi + 1), lam)
        pmf[0] += stats.poisson.cdf(lo - 1, lam)
        pmf[-1] += stats.poisson.sf(hi, lam)
        chi2 = ((obs - n * pmf) ** 2 / (n * pmf)).sum()
        assert chi2 < stats.chi2.ppf(1 - 1e-6, hi - lo), (lam, chi2)


def test_draws_depend_on_image_and_pixel_not_on_batching(rp):
    rng = np.random.default_rng(1)
    img = rng.random((6, 64 * 64)).astype(np.float32)
    scale = torch.full((6,), 80.0, device=rp.device)
    x = torch.from_numpy(img).to(rp.device)
    rp.next_image = 100
    _, _, c_all = rp.gen_lq(x, scale, want_counts=True)
    rp.next_image = 103
    _, _, c_tail = rp.gen_lq(x[3:].contiguous(), scale[3:].contiguous(), want_counts=True)
    rp.next_image = 0
    assert torch.equal(c_all[3:], c_tail)
    assert not torch.equal(c_all[0], c_all[1])
    # the seed is the Philox KEY and the image index sits in the COUNTER: (seed, image) pairs with equal XOR -- seed 0 / image 1 and
    # seed 1 / image 0, what a "seed = base + rank" scheme produces -- draw different noise fields
    from emdenoise import input_pipeline as ip

    a = ip.DeviceRecordParser(rp.device, seed=0, first_image=1).gen_lq(x[:1].contiguous(), scale[:1].contiguous(), want_counts=True)[2]
    b = ip.DeviceRecordParser(rp.device, seed=1, first_image=0).gen_lq(x[:1].contiguous(), scale[:1].contiguous(), want_counts=True)[2]
    assert not torch.equal(a, b)


def test_record_parser_end_to_end(rp):
    """DeviceRecordParser.__call__ = record_parser (:861-870) over a batch: ranges, the truth's mean, D4 membership."""
    rng = np.random.default_rng(2)
    hq = rng.random((8, 128, 128, 1)).astype(np.float32)
    hq[0, 5, 5, 0] = np.nan
    rp.next_image = 40
    lq, truth = rp(torch.from_numpy(hq).to(rp.device))
    assert rp.next_image == 48
    lq, truth = lq.cpu().numpy(), truth.cpu().numpy()
    ch = R.d4_choices(8, rp.seed, 40)
    sc = R.get_scale(8, rp.seed, 40)
    for b in range(8):
        assert lq[b].min() == 0.0 and lq[b].max() == 1.0 and np.isfinite(truth[b]).all()
        src = hq[b, :, :, 0].copy()
        src[~np.isfinite(src)] = 0.5
        img = R.scale0to1(R.flip_rotate(src, ch[b]))
        ratio = truth[b, :, :, 0].astype(np.float64).sum() / img.astype(np.float64).sum()
        np.testing.assert_allclose(truth[b, :, :, 0], (np.float32(ratio) * img), rtol=3e-6)       # truth is a multiple of the D4 image
        assert np.isclose(truth[b].mean(), lq[b].mean(), rtol=1e-5)
        # the counts behind lq have the rate img * scale: corr(lq, img) is high at these count levels
        assert np.corrcoef(lq[b].ravel(), img.ravel())[0, 1] > 0.8, sc[b]
    rp.next_image = 0
