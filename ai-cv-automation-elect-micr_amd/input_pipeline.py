"""Input-side host logic: batch sharding across GPUs and the reference's numpy feeding conventions.

Mirrors misc_py/denoiser-multi-gpu.py:783-913 (get_scale, gen_lq, scale0to1, flip_rotate, preprocess,
record_parser, input_fn's round-robin sharding) and small_scans/convert_to_numpy.py:11-21 (.npy stacks of
[N,H,W,1] float32).  Everything here is host-side numpy; the GPU path starts at Denoiser.denoise.

Multi-GPU model: one process per GPU; inference shards WHOLE IMAGES across ranks and needs no
collective -- weights are replicated, results stay on the rank that produced them (or are gathered by the
caller).  The reference instead builds in-graph towers and deals images round-robin (`i % num_shards`,
denoiser-multi-gpu.py:905-909); ``shard_round_robin`` reproduces that order, ``shard_contiguous`` is the
per-process form used here.
"""
from __future__ import annotations

import numpy as np


# ---- sharding ---------------------------------------------------------------------------------------------
def shard_contiguous(n: int, world: int, rank: int):
    """Index range [lo, hi) of the `rank`-th of `world` near-equal contiguous shards of n items."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_round_robin(batch, num_shards: int):
    """The reference's input_fn sharding (denoiser-multi-gpu.py:898-913): image i goes to shard i % num_shards;
    returns the list of per-GPU arrays (`feature_shards`)."""
    if num_shards <= 1:
        return [batch]
    return [batch[i::num_shards] for i in range(num_shards)]


def denoise_sharded(denoiser, lq_batch, rank: int, world: int):
    """Run this rank's contiguous shard of a host batch; returns (lo, hi, hq_shard).  No collective."""
    lo, hi = shard_contiguous(len(lq_batch), world, rank)
    if hi == lo:
        return lo, hi, lq_batch[:0]
    return lo, hi, denoiser.denoise_batch(lq_batch[lo:hi])


def max_over_ranks(value: float, dist=None, device=None) -> float:
    """Step time of a multi-rank job = the slowest rank's (bench.py contract)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch

    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---- the reference's host-side image functions ------------------------------------------------------------
def scale0to1(img):
    """denoiser-multi-gpu.py:817-828."""
    img = np.asarray(img)
    lo, hi = np.min(img), np.max(img)
    if lo == hi:
        # the reference's img.fill(0.5): 0.5 for a float image -- and 0 for the int64 Poisson counts gen_lq hands over (:797; fill casts)
        return np.full(img.shape, 0.0 if np.issubdtype(np.asarray(img).dtype, np.integer) else 0.5, np.float32)
    return ((img - lo) / (hi - lo)).astype(np.float32)


def get_scale(rng):
    """denoiser-multi-gpu.py:783-784: 25 + Exp(mean 75)."""
    return 25.0 + rng.exponential(75.0)


def gen_lq(img, scale, rng, img_type=np.float32):
    """denoiser-multi-gpu.py:787-799: Poisson counts at `scale`, min-max rescaled."""
    return scale0to1(rng.poisson(np.asarray(img, np.float64) * scale)).astype(img_type)


def flip_rotate(img, choice: int):
    """denoiser-multi-gpu.py:830-851: the 8 elements of D4, selected by `choice` in 0..7."""
    if choice == 0:
        return img
    if choice in (1, 2, 3):
        return np.rot90(img, choice)
    if choice == 4:
        return np.flip(img, 0)
    if choice == 5:
        return np.flip(img, 1)
    if choice == 6:
        return np.flip(np.rot90(img, 1), 0)
    if choice == 7:
        return np.flip(np.rot90(img, 1), 1)
    raise ValueError("choice must be 0..7")


def preprocess(img, rng):
    """denoiser-multi-gpu.py:853-858: NaN/Inf -> 0.5, random D4 element, min-max."""
    img = np.array(img, dtype=np.float32, copy=True)
    img[np.isnan(img)] = 0.5
    img[np.isinf(img)] = 0.5
    return scale0to1(flip_rotate(img, int(8 * rng.random())))


def record_parser(img, rng):
    """denoiser-multi-gpu.py:861-870: (lq, truth rescaled to the lq mean)."""
    img = preprocess(img, rng)
    lq = gen_lq(img, get_scale(rng), rng)
    return lq, ((np.mean(lq) / np.mean(img)) * img).astype(np.float32)


def load_npy_stack(path):
    """small_scans/convert_to_numpy.py:11-21 layout: float32 [N,H,W,1] (mmap, nothing is unpickled)."""
    a = np.load(path, mmap_mode="r", allow_pickle=False)
    if a.ndim == 3:
        a = a[..., None]
    if a.ndim != 4 or a.shape[3] != 1:
        raise ValueError(f"{path}: expected [N,H,W,1], got {a.shape}")
    return a


def input_fn(stack, batch_size: int, num_shards: int, seed: int = 0, epochs: int = 1):
    """Iterator of (feature_shards, truth_shards) lists of per-GPU [n,H,W,1] arrays, like the reference's
    input_fn (denoiser-multi-gpu.py:878-913), from an in-memory / mmapped [N,H,W,1] stack of HQ images."""
    rng = np.random.default_rng(seed)
    n = len(stack)
    for _ in range(epochs):
        order = rng.permutation(n)
        for k in range(0, n - batch_size + 1, batch_size):
            pairs = [record_parser(np.asarray(stack[i])[..., 0], rng) for i in order[k:k + batch_size]]
            lq = np.stack([p[0] for p in pairs])[..., None]
            hq = np.stack([p[1] for p in pairs])[..., None]
            yield shard_round_robin(lq, num_shards), shard_round_robin(hq, num_shards)


# ---- TFRecord files (misc_py/TFRecord_creator.py:57-85) -----------------------------------------------------
# One tf.train.Example per image with a single bytes feature 'image' = the raw float32 pixels.  TensorFlow is
# not needed to read them: the container and the protobuf wire format are public.  (Format restated from the
# TFRecord / protobuf specifications; no TensorFlow-written file ships with the reference to pin it against.)
def _masked_crc(data) -> int:
    from . import _lib

    buf = np.frombuffer(data, dtype=np.uint8)
    crc = _lib.load().emd_crc32c(buf.ctypes.data if len(buf) else None, len(buf), 0)
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf):
    """Yield (field number, wire type, value) of one protobuf message; LEN values are memoryviews."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = bytes(buf[pos:pos + 4])
            pos += 4
        elif wt == 1:
            v = bytes(buf[pos:pos + 8])
            pos += 8
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, v


def parse_example(record) -> dict:
    """tf.train.Example -> {feature name: list of bytes objects} (bytes_list features only)."""
    out = {}
    for fno, _, feats in _fields(memoryview(record)):
        if fno != 1:
            continue
        for f2, _, entry in _fields(feats):          # Features.feature: map<string, Feature>
            if f2 != 1:
                continue
            key, value = None, None
            for f3, _, v in _fields(entry):
                if f3 == 1:
                    key = bytes(v).decode()
                elif f3 == 2:
                    value = v
            vals = []
            if value is not None:
                for f4, _, lst in _fields(value):        # Feature.bytes_list = 1
                    if f4 == 1:
                        vals.extend(bytes(b) for f5, _, b in _fields(lst) if f5 == 1)
            out[key] = vals
    return out


def read_tfrecord(path, verify=True):
    """Yield the raw record payloads of a TFRecord file (memory-mapped, CRCs checked when verify)."""
    import mmap
    import struct

    with open(path, "rb") as fh:
        mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
        try:
            pos, n = 0, len(mm)
            while pos < n:
                if pos + 12 > n:
                    raise ValueError(f"{path}: truncated record header at byte {pos}")
                (length,) = struct.unpack_from("<Q", mm, pos)
                (lcrc,) = struct.unpack_from("<I", mm, pos + 8)
                if verify and _masked_crc(mm[pos:pos + 8]) != lcrc:
                    raise ValueError(f"{path}: corrupt length field at byte {pos}")
                start, end = pos + 12, pos + 12 + length
                if end + 4 > n:
                    raise ValueError(f"{path}: truncated record at byte {pos}")
                data = mm[start:end]
                (dcrc,) = struct.unpack_from("<I", mm, end)
                if verify and _masked_crc(data) != dcrc:
                    raise ValueError(f"{path}: corrupt record at byte {pos}")
                yield data
                pos = end + 4
        finally:
            mm.close()


def tfrecord_images(path, shape=None, verify=True):
    """Yield float32 images from the 'image' feature (TFRecord_creator.py:75); `shape` = (H, W) if known,
    else square images are assumed (the reference stores 2048 x 2048)."""
    for rec in read_tfrecord(path, verify):
        raw = parse_example(rec)["image"][0]
        a = np.frombuffer(raw, dtype=np.float32)
        if shape is None:
            side = int(round(len(a) ** 0.5))
            if side * side != len(a):
                raise ValueError("non-square image: pass shape=(H, W)")
            shp = (side, side)
        else:
            shp = tuple(shape)
        yield a.reshape(shp)


def write_tfrecord(path, images):
    """Write float32 images exactly as TFRecord_creator.py:57-85 does (one Example, bytes feature 'image')."""
    import struct

    def ln(b):  # protobuf varint
        out = bytearray()
        while True:
            out.append((b & 0x7F) | (0x80 if b > 0x7F else 0))
            b >>= 7
            if not b:
                return bytes(out)

    def field(no, payload):
        return ln((no << 3) | 2) + ln(len(payload)) + payload

    with open(path, "wb") as fh:
        for img in images:
            raw = np.ascontiguousarray(img, dtype=np.float32).tobytes()
            feature = field(1, field(1, raw))                       # Feature{bytes_list{value}}
            entry = field(1, b"image") + field(2, feature)          # map entry
            example = field(1, field(1, entry))                     # Example{features{feature}}
            head = struct.pack("<Q", len(example))
            fh.write(head + struct.pack("<I", _masked_crc(head)) + example + struct.pack("<I", _masked_crc(example)))


# ---- the same functions on the device (csrc/input_ops.hip) ---------------------------------------------------
class DeviceRecordParser:
    """record_parser (denoiser-multi-gpu.py:861-870) for a whole batch on the GPU: preprocess (NaN/Inf -> 0.5, a random element
    of D4, min-max), get_scale, gen_lq (Poisson counts, min-max) and the truth rescale, as libemdenoise.so launches on torch's
    current stream.  HQ images in, (lq, truth) out, all float32 CUDA tensors [B,S,S,1]; nothing returns to the host, so the
    trainer is fed at device speed (the reference runs these in 8 tf.py_func threads, :78).

    Randomness is counter-based (Philox4x32-10 keyed by `seed`, indexed by the image's position in the stream of images this
    parser has served): image k of the stream gets the same scale, D4 element and Poisson draws however the stream is cut into
    batches or spread over ranks (give each rank a disjoint `first_image` range)."""

    def __init__(self, device, seed: int = 0, first_image: int = 0):
        from . import _lib

        self._lib = _lib
        self.lib = _lib.load()
        self.device = device
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.next_image = int(first_image)

    def _ws(self, B, npix):
        import torch

        n = self.lib.emd_input_workspace_bytes(B, npix)
        return torch.empty((n + 15) // 16 * 4, dtype=torch.float32, device=self.device)

    def flip_rotate(self, x, choices=None, fix_nonfinite=False):
        """x [B,S,S(,1)] -> the D4 element `choices[b]` (int32 CUDA tensor; None = draw them) of every image."""
        import ctypes as C

        import torch

        B, S = x.shape[0], x.shape[1]
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.shape[2] == S
        st = self._lib.stream_ptr()
        if choices is None:
            choices = torch.empty(B, dtype=torch.int32, device=self.device)
            self._lib.check(self.lib.emd_d4_choices_i32(C.c_void_p(choices.data_ptr()), B, self.seed, self.next_image, st), "emd_d4_choices_i32")
        assert choices.dtype == torch.int32 and choices.numel() == B and choices.is_cuda
        y = torch.empty_like(x)
        self._lib.check(self.lib.emd_flip_rotate_f32(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), B, S, S,
                                                     C.c_void_p(choices.data_ptr()), 1 if fix_nonfinite else 0, st), "emd_flip_rotate_f32")
        return y, choices

    def scale0to1(self, x, out=None):
        import ctypes as C

        import torch

        B = x.shape[0]
        npix = x.numel() // B
        st = self._lib.stream_ptr()
        mn = torch.empty(B, dtype=torch.float32, device=self.device)
        mx = torch.empty_like(mn)
        ws = self._ws(B, npix)
        out = torch.empty_like(x) if out is None else out
        p = lambda t: C.c_void_p(t.data_ptr())
        self._lib.check(self.lib.emd_minmax_images_f32(p(x), B, npix, p(mn), p(mx), p(ws), st), "emd_minmax_images_f32")
        self._lib.check(self.lib.emd_scale0to1_images_f32(p(x), p(out), B, npix, p(mn), p(mx), st), "emd_scale0to1_images_f32")
        return out

    def get_scale(self, B):
        import ctypes as C

        import torch

        s = torch.empty(B, dtype=torch.float32, device=self.device)
        self._lib.check(self.lib.emd_get_scale_f32(C.c_void_p(s.data_ptr()), B, self.seed, self.next_image, self._lib.stream_ptr()),
                        "emd_get_scale_f32")
        return s

    def gen_lq(self, img, scale, want_counts=False):
        """img [B,...] in [0,1], scale [B] -> (lq, truth[, counts int32]) (:787-799, :868)."""
        import ctypes as C

        import torch

        B = img.shape[0]
        npix = img.numel() // B
        assert img.is_cuda and img.dtype == torch.float32 and img.is_contiguous() and scale.numel() == B and scale.dtype == torch.float32
        lq, truth = torch.empty_like(img), torch.empty_like(img)
        counts = torch.empty(img.shape, dtype=torch.int32, device=self.device) if want_counts else None
        ws = self._ws(B, npix)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
        self._lib.check(self.lib.emd_gen_lq_f32(p(img), p(scale), p(lq), p(truth), p(counts), B, npix, self.seed, self.next_image, p(ws),
                                                self._lib.stream_ptr()), "emd_gen_lq_f32")
        return (lq, truth, counts) if want_counts else (lq, truth)

    def __call__(self, hq):
        """hq [B,S,S,1] float32 CUDA (raw HQ images) -> (lq, truth): record_parser over the batch; advances the image counter."""
        B = hq.shape[0]
        img, _ = self.flip_rotate(hq, fix_nonfinite=True)       # preprocess (:853-858) ...
        img = self.scale0to1(img, out=img)                      # ... min-max in place
        lq, truth = self.gen_lq(img, self.get_scale(B))         # :865-868
        self.next_image += B
        return lq, truth
