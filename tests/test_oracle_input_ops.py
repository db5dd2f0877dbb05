"""CPU tests of the oracle for the device input functions (oracle/input_ops_ref.py): the Philox generator against the
known-answer vectors published with Random123, the derived draws against their distributions, and the deterministic tail of
gen_lq / record_parser against the product's host numpy functions (emdenoise.input_pipeline, misc_py/denoiser-multi-gpu.py:783-870)."""
import numpy as np

from emdenoise import input_pipeline as ip
from oracle import input_ops_ref as R

# Random123 kat_vectors, philox4x32 with 10 rounds: (counter, key, expected)
KAT = [
    ((0x00000000,) * 4, (0x00000000,) * 2, (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


def test_philox_known_answers():
    for ctr, key, want in KAT:
        got = R.philox4x32_10([np.uint64(v) for v in ctr], key)
        assert tuple(int(v) for v in got) == want


def test_raw_stream_is_counter_indexed():
    a = R.raw(64, seed=0x1234_5678_9ABC_DEF0)
    b = R.raw(16, seed=0x1234_5678_9ABC_DEF0, counter0=48)
    assert (a[48:] == b).all() and a.dtype == np.uint32
    assert len({tuple(r) for r in a}) == 64


def test_uniform_is_open_interval_and_flat():
    r = R.raw(50000, seed=7)
    u = R.u01(r[:, 0], r[:, 1])
    assert u.min() > 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 5e-3 and abs(u.var() - 1 / 12) < 2e-3
    assert R.u01(np.array([0], np.uint32), np.array([0], np.uint32))[0] > 0.0
    assert R.u01(np.array([0xFFFFFFFF], np.uint32), np.array([0xFFFFFFFF], np.uint32))[0] < 1.0


def test_scale_and_choice_distributions():
    s = R.get_scale(40000, seed=11).astype(np.float64)
    assert s.min() >= 25.0 and abs(s.mean() - 100.0) < 2.0 and abs((s - 25.0).std() - 75.0) < 2.5     # 25 + Exp(mean 75), :783-784
    c = R.d4_choices(80000, seed=11)
    assert c.min() == 0 and c.max() == 7
    assert np.abs(np.bincount(c, minlength=8) / c.size - 0.125).max() < 6e-3                          # int(8 * rand), :833
    assert (R.get_scale(8, 3, first_image=4)[:4] == R.get_scale(8, 3)[4:]).all()                      # indexed by image, not by batch


def test_flip_rotate_is_the_host_function():
    img = np.random.default_rng(0).random((12, 12)).astype(np.float32)
    for ch in range(8):
        np.testing.assert_array_equal(R.flip_rotate(img, ch), ip.flip_rotate(img, ch))


def test_lq_and_truth_from_counts_match_the_host_gen_lq():
    rng = np.random.default_rng(5)
    img = ip.scale0to1(rng.random((32, 32)).astype(np.float32))

    class Fixed:   # a generator that returns known counts, to compare the deterministic tail only
        def __init__(self, c):
            self.c = c

        def poisson(self, lam):
            return self.c

    counts = rng.poisson(img.astype(np.float64) * 60.0)
    lq = ip.gen_lq(img, 60.0, Fixed(counts))
    lq2, truth2 = R.lq_and_truth_from_counts(counts, img)
    np.testing.assert_array_equal(lq, lq2)
    np.testing.assert_array_equal(truth2, ((np.mean(lq) / np.mean(img)) * img).astype(np.float32))
    flat = np.full((8, 8), 3, np.int64)
    # constant counts: the reference's scale0to1 fills an INT64 array with 0.5, which stores 0 (denoiser-multi-gpu.py:797, :824)
    lq0, truth0 = R.lq_and_truth_from_counts(flat, np.ones((8, 8), np.float32))
    assert (lq0 == 0.0).all() and (truth0 == 0.0).all()
    assert (ip.gen_lq(np.ones((8, 8), np.float32), 3.0, Fixed(flat)) == 0.0).all()
    assert (ip.scale0to1(np.full((4, 4), 2.5, np.float32)) == 0.5).all()                               # a constant FLOAT image: 0.5 (:823-824)
