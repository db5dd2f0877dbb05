"""TensorFlow checkpoint bundle reader/writer (emdenoise.tf_checkpoint).  CPU only.
PARITY UNPINNED: no checkpoint ships with the reference and TensorFlow is not installable here; the pins are the
format's published constants (table magic, CRC-32C check value and mask, protobuf field numbers), structural checks
of the bytes this writer produces, and round trips."""
import os
import struct

import numpy as np
import pytest

from emdenoise import denoiser as D
from emdenoise import tf_checkpoint as ck


def test_crc32c_check_value_and_mask():
    assert ck._crc32c(b"123456789") == 0xE3069283                      # CRC-32C (Castagnoli) check value
    assert ck._mask(0) == 0xA282EAD8 and ck._mask(0xFFFFFFFF) == (0xFFFFFFFF + 0xA282EAD8) & 0xFFFFFFFF
    assert ck._mask(1 << 15) == (1 + 0xA282EAD8)                        # rotate right by 15, then add the delta


def test_varint_and_entry_proto_round_trip():
    for v in (0, 1, 127, 128, 300, 2 ** 31, 2 ** 40 + 5):
        b = ck._put_varint(v)
        assert ck._get_varint(b, 0) == (v, len(b))
    e = ck._parse_entry(ck._entry_proto(1, (3, 3, 728, 1), 123456, 26208, 0xDEADBEEF))
    assert e["dtype"] == 1 and e["shape"] == (3, 3, 728, 1) and e["offset"] == 123456 and e["size"] == 26208
    assert e["crc32c"] == 0xDEADBEEF and e["shard_id"] == 0
    # hand-assembled BundleEntryProto: dtype DT_FLOAT, shape [2,5], offset 8, size 40, crc
    raw = bytes([0x08, 1, 0x12, 8, 0x12, 2, 0x08, 2, 0x12, 2, 0x08, 5, 0x20, 8, 0x28, 40, 0x35]) + struct.pack("<I", 7)
    e = ck._parse_entry(raw)
    assert (e["dtype"], e["shape"], e["offset"], e["size"], e["crc32c"]) == (1, (2, 5), 8, 40, 7)


def test_round_trip_of_graph_d_and_denoiser_loader(tmp_path):
    """All 658 variables of graph D (+ an optimizer slot and global_step, as tf.train.Saver writes them) through
    write -> index with many blocks -> read; then the reference's constructor path: a DIRECTORY with a ``checkpoint``
    state file, resolved as tf.train.latest_checkpoint does."""
    w = D.synthetic_weights(bn="tf_init")
    extra = {"global_step": np.array(12345, np.int64), "nn/Conv/weights/Momentum": np.zeros((1, 1, 1, 128), np.float32)}
    prefix = str(tmp_path / "model" / "model.ckpt-12345")
    ck.write_checkpoint(prefix, {**w, **extra})
    assert os.path.getsize(prefix + ".index") > 3 * ck.BLOCK_SIZE      # several data blocks + an index block
    header, entries = ck.read_index(prefix + ".index")
    assert header[1] == 1 and list(entries) == sorted(entries) and len(entries) == len(w) + 2
    assert entries["nn/SeparableConv2d_7/depthwise_weights"]["shape"] == w["nn/SeparableConv2d_7/depthwise_weights"].shape
    back = ck.read_checkpoint(prefix)
    assert back["global_step"].shape == () and int(back["global_step"]) == 12345
    for k, v in w.items():
        assert back[k].dtype == np.float32 and np.array_equal(back[k], v), k
    assert ck.latest_checkpoint(str(tmp_path / "model")) == prefix
    loaded = D.load_weights(str(tmp_path / "model"))                 # Denoiser(checkpoint_loc=<dir>) path
    assert list(loaded) == list(w) and all(np.array_equal(loaded[k], w[k]) for k in w)
    loaded = D.load_weights(prefix)                                   # the prefix itself
    assert np.array_equal(loaded["nn/BatchNorm_3/gamma"], w["nn/BatchNorm_3/gamma"])


def test_index_bytes_follow_the_table_format(tmp_path):
    prefix = str(tmp_path / "m")
    ck.write_checkpoint(prefix, {"nn/a/weights": np.arange(6, dtype=np.float32).reshape(2, 3), "nn/a/biases": np.ones(3, np.float32)})
    idx = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", idx[-8:])[0] == 0xDB4775248B80FB57 and len(idx) >= 48
    # first data block starts at 0: entry 0 is the header (empty key), entry 1 "nn/a/biases", entry 2 shares "nn/a/"
    shared, p = ck._get_varint(idx, 0)
    nonshared, p = ck._get_varint(idx, p)
    assert (shared, nonshared) == (0, 0)
    blk = bytes(ck._read_block(memoryview(idx), 0, _first_block_len(idx), True))
    assert [k for k, _ in ck._block_entries(blk)] == [b"", b"nn/a/biases", b"nn/a/weights"]
    assert b"nn/a/biases" in blk and b"nn/a/weights" not in blk and b"weights" in blk   # prefix compression: "nn/a/" shared
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    assert np.array_equal(np.frombuffer(data, np.float32), [1, 1, 1, 0, 1, 2, 3, 4, 5])   # keys sorted: biases first


def _first_block_len(idx):
    """Size of the first data block, from the index block the footer points at."""
    footer = idx[-48:]
    p = 0
    _, p = ck._get_varint(footer, p)
    _, p = ck._get_varint(footer, p)
    ioff, p = ck._get_varint(footer, p)
    isize, p = ck._get_varint(footer, p)
    handle = next(ck._block_entries(ck._read_block(memoryview(idx), ioff, isize, True)))[1]
    off, q = ck._get_varint(handle, 0)
    size, _ = ck._get_varint(handle, q)
    assert off == 0
    return size


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / "m")
    ck.write_checkpoint(prefix, {"v": np.arange(100, dtype=np.float32)})
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    data[17] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    with pytest.raises(ValueError, match="tensor checksum"):
        ck.read_checkpoint(prefix)
    assert ck.read_checkpoint(prefix, verify=False)["v"].shape == (100,)
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[3] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError, match="block checksum"):
        ck.read_index(prefix + ".index")
    with pytest.raises(ValueError, match="table magic"):
        open(prefix + ".index", "wb").write(b"\0" * 64)
        ck.read_index(prefix + ".index")
    with pytest.raises(KeyError):
        ck.write_checkpoint(prefix, {"v": np.zeros(2, np.float32)})
        ck.read_checkpoint(prefix, names=["w"])
