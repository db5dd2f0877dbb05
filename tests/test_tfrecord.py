"""CPU tests of the TFRecord reader (emdenoise.input_pipeline), the container misc_py/TFRecord_creator.py:57-85
writes.  No TensorFlow-written file ships with the reference, so the pins are the published format constants:
the CRC-32C check value, the TFRecord CRC mask, and a hand-assembled tf.train.Example."""
import struct

import numpy as np
import pytest

from emdenoise import _lib
from emdenoise import input_pipeline as ip


def test_crc32c_known_answers():
    lib = _lib.load()
    data = np.frombuffer(b"123456789", dtype=np.uint8)
    assert lib.emd_crc32c(data.ctypes.data, 9, 0) == 0xE3069283            # CRC-32C check value (RFC 3720)
    zeros = np.zeros(32, np.uint8)
    assert lib.emd_crc32c(zeros.ctypes.data, 32, 0) == 0x8A9136AA          # RFC 3720 B.4: 32 bytes of zeros
    a = lib.emd_crc32c(data.ctypes.data, 4, 0)
    assert lib.emd_crc32c(data.ctypes.data + 4, 5, a) == 0xE3069283        # incremental == one shot
    assert ip._masked_crc(b"123456789") == ((((0xE3069283 >> 15) | (0xE3069283 << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def test_hand_assembled_example_parses():
    raw = np.arange(4, dtype=np.float32).tobytes()                           # 16 bytes
    bytes_list = b"\x0a\x10" + raw                                           # field 1, LEN 16
    feature = b"\x0a" + bytes([len(bytes_list)]) + bytes_list                # Feature.bytes_list
    entry = b"\x0a\x05image" + b"\x12" + bytes([len(feature)]) + feature     # map entry key=1, value=2
    features = b"\x0a" + bytes([len(entry)]) + entry
    example = b"\x0a" + bytes([len(features)]) + features
    assert ip.parse_example(example) == {"image": [raw]}


def test_round_trip_and_corruption(tmp_path):
    rng = np.random.default_rng(0)
    imgs = [rng.random((12, 12)).astype(np.float32), rng.random((12, 12)).astype(np.float32) * 5 - 1]
    path = str(tmp_path / "train.tfrecords")
    ip.write_tfrecord(path, imgs)
    got = list(ip.tfrecord_images(path))
    assert len(got) == 2 and all(np.array_equal(a, b) for a, b in zip(got, imgs))
    rect = [rng.random((3, 5)).astype(np.float32)]
    ip.write_tfrecord(path, rect)
    assert np.array_equal(next(ip.tfrecord_images(path, shape=(3, 5))), rect[0])
    with pytest.raises(ValueError, match="non-square"):
        next(ip.tfrecord_images(path))
    blob = bytearray(open(path, "rb").read())
    blob[40] ^= 0xFF                                                          # flip a payload byte
    open(path, "wb").write(bytes(blob))
    with pytest.raises(ValueError, match="corrupt record"):
        list(ip.read_tfrecord(path))
    assert len(list(ip.read_tfrecord(path, verify=False))) == 1              # unchecked read still frames it
    open(path, "wb").write(bytes(blob[:-3]))
    with pytest.raises(ValueError, match="truncated"):
        list(ip.read_tfrecord(path, verify=False))
    length = struct.unpack_from("<Q", blob, 0)[0]
    assert length == len(blob) - 16
