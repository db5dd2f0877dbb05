"""TEST INFRASTRUCTURE ONLY (never imported by the product): numpy restatement of the device input functions of
csrc/input_ops.hip, which replace misc_py/denoiser-multi-gpu.py:783-870.

* Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11) -- pinned by the
  known-answer vectors published with the Random123 library (tests/test_oracle_input_ops.py);
* the uniform / get_scale / D4-choice formulas of the kernels on top of it (bit-exact twins);
* gen_lq's deterministic tail: scale0to1 of integer counts and the truth rescale, written with numpy exactly as the
  reference writes them (:797-799, :817-828, :868).
The Poisson sampler itself has no bit-exact twin here: the kernel and numpy.random.Generator.poisson use the same two
algorithms (inversion below 10, Hoermann's PTRS above) on different uniform streams; tests compare distributions."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
TAG_RAW, TAG_SCALE, TAG_CHOICE, TAG_POISSON = 0, 1, 2, 3
MASK = 0xFFFFFFFF


def philox4x32_10(counter, key):
    """counter: 4 uint32 (or arrays of equal shape), key: 2 uint32 -> 4 uint32 arrays."""
    c = [np.asarray(v, dtype=np.uint64) for v in counter]
    k0, k1 = int(key[0]) & MASK, int(key[1]) & MASK
    for _ in range(10):
        p0 = np.uint64(M0) * c[0]
        p1 = np.uint64(M1) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)) & np.uint64(MASK), p1 & np.uint64(MASK),
             ((p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)) & np.uint64(MASK), p0 & np.uint64(MASK)]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return [v.astype(np.uint32) for v in c]


def raw(n4, seed, counter0=0):
    """What emd_philox4x32_u32 writes: [n4, 4] uint32."""
    idx = np.arange(n4, dtype=np.uint64) + np.uint64(counter0)
    r = philox4x32_10([idx & np.uint64(MASK), idx >> np.uint64(32), np.zeros(n4, np.uint64), np.full(n4, TAG_RAW, np.uint64)],
                      [seed & MASK, (seed >> 32) & MASK])
    return np.stack(r, axis=1)


def u01(hi, lo):
    m = ((hi.astype(np.uint64) << np.uint64(32)) | lo.astype(np.uint64)) >> np.uint64(12)   # 52 bits: m + 1/2 is exact
    return (m.astype(np.float64) + 0.5) * (1.0 / 4503599627370496.0)


def _per_image(B, seed, first_image, tag):
    idx = np.arange(B, dtype=np.uint64) + np.uint64(first_image)
    return philox4x32_10([idx & np.uint64(MASK), idx >> np.uint64(32), np.zeros(B, np.uint64), np.full(B, tag, np.uint64)],
                         [seed & MASK, (seed >> 32) & MASK])


def get_scale(B, seed, first_image=0):
    """25 + Exp(mean 75) (:783-784) from the kernel's uniform: 25 - 75 ln(u), rounded to float32."""
    r = _per_image(B, seed, first_image, TAG_SCALE)
    return (25.0 - 75.0 * np.log(u01(r[0], r[1]))).astype(np.float32)


def d4_choices(B, seed, first_image=0):
    """int(8 * u) (:833) with u = word / 2^32."""
    return (_per_image(B, seed, first_image, TAG_CHOICE)[0] >> np.uint32(29)).astype(np.int32)


def flip_rotate(img, choice):
    """:830-851, verbatim semantics (numpy's own rot90 / flip)."""
    return [lambda a: a, lambda a: np.rot90(a, 1), lambda a: np.rot90(a, 2), lambda a: np.rot90(a, 3), lambda a: np.flip(a, 0),
            lambda a: np.flip(a, 1), lambda a: np.flip(np.rot90(a, 1), 0), lambda a: np.flip(np.rot90(a, 1), 1)][int(choice)](img)


def scale0to1(img):
    """:817-828."""
    lo, hi = np.min(img), np.max(img)
    if lo == hi:
        # img.fill(0.5): 0.5 for a float image, but ZERO for the int64 Poisson counts gen_lq passes in (:797 -- fill casts to the dtype)
        return np.full(img.shape, 0.0 if np.issubdtype(np.asarray(img).dtype, np.integer) else 0.5, np.float32)
    return ((img - lo) / (hi - lo)).astype(np.float32)


def lq_and_truth_from_counts(counts, img):
    """:797-799 and :868 given the Poisson counts (int64, as numpy's poisson returns them) and the float32 image."""
    lq = scale0to1(counts.astype(np.int64))
    return lq, ((np.mean(lq) / np.mean(img)) * img).astype(np.float32)
