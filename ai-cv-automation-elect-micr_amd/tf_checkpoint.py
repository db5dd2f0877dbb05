"""TensorFlow checkpoint ("tensor bundle", the V2 format tf.train.Saver writes) reader and writer in pure Python.

The reference restores its weights with ``tf.train.Saver().restore(sess, tf.train.latest_checkpoint(dir))``
(machine_learning/denoiser.py:621-626, misc_py/apply_kernels+MLPs.py:622-627) from directories such as
``.../models/denoiser-multi-gpu-13/model/`` (:588) holding ``checkpoint``, ``model.index`` and
``model.data-00000-of-00001``.  This module lets ``Denoiser(checkpoint_loc=<that dir>)`` read those files directly,
without TensorFlow, and lets the trainer write weights back in the same format.

Format (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/table -- the LevelDB table format):
  <prefix>.index   an SSTable: data blocks of prefix-compressed (key, value) entries + restart array, each block
                   followed by a 1-byte compression type and a 4-byte masked CRC-32C; a metaindex block; an index
                   block (last key of each data block -> BlockHandle); a 48-byte footer ending in the magic
                   0xdb4775248b80fb57.  Key "" -> BundleHeaderProto, every other key = a variable name ->
                   BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}.
  <prefix>.data-XXXXX-of-YYYYY   the raw little-endian tensor bytes.
  checkpoint       text: ``model_checkpoint_path: "<prefix basename>"``.
PARITY UNPINNED: no checkpoint ships with the reference and TensorFlow cannot be installed here, so the pins are the
published format constants, CRC-32C check values and round trips through this module's own writer (tests/test_tf_checkpoint.py).
Index blocks are expected uncompressed (BundleWriter sets table::kNoCompression); a Snappy block raises.
"""
from __future__ import annotations

import os
import re
import struct
from collections import OrderedDict

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
CRC_MASK_DELTA = 0xA282EAD8
BLOCK_SIZE = 4096          # table::Options::block_size default
RESTART_INTERVAL = 16      # table::Options::block_restart_interval default

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DTYPE_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


def _crc32c(data: bytes) -> int:
    from . import _lib

    buf = np.frombuffer(data, np.uint8)
    return int(_lib.load().emd_crc32c(buf.ctypes.data if len(buf) else None, len(buf), 0))


def _mask(crc: int) -> int:
    return (((crc >> 15) | (crc << 17)) + CRC_MASK_DELTA) & 0xFFFFFFFF


# ---- protobuf / varint primitives --------------------------------------------------------------------------
def _get_varint(b, pos):
    shift = result = 0
    while True:
        c = b[pos]
        pos += 1
        result |= (c & 0x7F) << shift
        if not c & 0x80:
            return result, pos
        shift += 7


def _put_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        c = v & 0x7F
        v >>= 7
        if v:
            out.append(c | 0x80)
        else:
            out.append(c)
            return bytes(out)


def _parse_fields(b):
    """protobuf wire format -> list of (field number, wire type, value)."""
    pos, out = 0, []
    while pos < len(b):
        tag, pos = _get_varint(b, pos)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(b, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", b, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(b, pos)
            v = bytes(b[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", b, pos)[0]
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        out.append((f, wt, v))
    return out


def _signed64(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def _parse_entry(value: bytes):
    """BundleEntryProto -> dict(dtype, shape, shard_id, offset, size, crc32c, sliced)."""
    e = {"dtype": 0, "shape": (), "shard_id": 0, "offset": 0, "size": 0, "crc32c": None, "sliced": False}
    for f, _, v in _parse_fields(value):
        if f == 1:
            e["dtype"] = v
        elif f == 2:   # TensorShapeProto: repeated Dim dim = 2 { int64 size = 1 }
            dims = []
            for f2, _, v2 in _parse_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _parse_fields(v2):
                        if f3 == 1:
                            size = _signed64(v3)
                    dims.append(size)
            e["shape"] = tuple(dims)
        elif f == 3:
            e["shard_id"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc32c"] = v
        elif f == 7:
            e["sliced"] = True
    return e


def _entry_proto(dtype_code, shape, offset, size, crc) -> bytes:
    dims = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(int(s)) for s in shape))
    out = b"\x08" + _put_varint(dtype_code) + b"\x12" + _put_varint(len(dims)) + dims
    if offset:
        out += b"\x20" + _put_varint(offset)
    out += b"\x28" + _put_varint(size) + b"\x35" + struct.pack("<I", crc)
    return out


# ---- table reader ---------------------------------------------------------------------------------------------
def _read_block(buf, offset, size, verify):
    contents = buf[offset:offset + size]
    ctype = buf[offset + size]
    if verify:
        stored = struct.unpack_from("<I", buf, offset + size + 1)[0]
        if stored != _mask(_crc32c(bytes(buf[offset:offset + size + 1]))):
            raise ValueError("checkpoint index: block checksum mismatch")
    if ctype == 1:
        raise ValueError("checkpoint index: Snappy-compressed block (BundleWriter writes them uncompressed); not supported")
    if ctype != 0:
        raise ValueError(f"checkpoint index: unknown block compression {ctype}")
    return contents


def _block_entries(block):
    n_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_index(path, verify=True):
    """<prefix>.index -> (header fields, OrderedDict name -> entry dict)."""
    buf = memoryview(open(path, "rb").read())
    if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != TABLE_MAGIC:
        raise ValueError(f"{path}: not a TensorFlow checkpoint index (bad table magic)")
    footer = buf[len(buf) - 48:]
    pos = 0
    _, pos = _get_varint(footer, pos)      # metaindex handle
    _, pos = _get_varint(footer, pos)
    ioff, pos = _get_varint(footer, pos)   # index handle
    isize, pos = _get_varint(footer, pos)
    entries, header = OrderedDict(), None
    for _, handle in _block_entries(_read_block(buf, ioff, isize, verify)):
        boff, p = _get_varint(handle, 0)
        bsize, _ = _get_varint(handle, p)
        for key, value in _block_entries(_read_block(buf, boff, bsize, verify)):
            if key == b"":
                header = {f: v for f, _, v in _parse_fields(value)}
            else:
                entries[key.decode()] = _parse_entry(value)
    if header is None:
        raise ValueError(f"{path}: no bundle header entry")
    if header.get(2, 0) != 0:
        raise ValueError(f"{path}: big-endian bundle")
    return header, entries


def latest_checkpoint(directory):
    """tf.train.latest_checkpoint: the prefix named by <directory>/checkpoint (model_checkpoint_path), or None."""
    state = os.path.join(directory, "checkpoint")
    if not os.path.exists(state):
        return None
    for line in open(state):
        m = re.match(r'\s*model_checkpoint_path:\s*"(.*)"', line)
        if m:
            p = m.group(1)
            return p if os.path.isabs(p) else os.path.join(directory, p)
    return None


def read_checkpoint(prefix, names=None, verify=True):
    """Tensors of the bundle <prefix>(.index, .data-*) as an OrderedDict name -> numpy array.  ``names``: only these
    (missing ones raise KeyError).  Optimizer slots, global_step etc. come along unless ``names`` filters them."""
    header, entries = read_index(prefix + ".index", verify)
    nshards = header.get(1, 1)
    want = list(entries) if names is None else list(names)
    shards = {}
    out = OrderedDict()
    for name in want:
        if name not in entries:
            raise KeyError(f"{prefix}: no tensor named {name!r}")
        e = entries[name]
        if e["sliced"]:
            raise ValueError(f"{name}: partitioned (sliced) variables are not supported")
        if e["dtype"] not in _DTYPES:
            raise ValueError(f"{name}: unsupported dtype code {e['dtype']}")
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = np.memmap(f"{prefix}.data-{sid:05d}-of-{nshards:05d}", dtype=np.uint8, mode="r")
        raw = bytes(shards[sid][e["offset"]:e["offset"] + e["size"]])
        if len(raw) != e["size"]:
            raise ValueError(f"{name}: data shard is truncated")
        if verify and e["crc32c"] is not None and e["crc32c"] != _mask(_crc32c(raw)):
            raise ValueError(f"{name}: tensor checksum mismatch")
        a = np.frombuffer(raw, dtype=_DTYPES[e["dtype"]])
        n = int(np.prod(e["shape"])) if e["shape"] else 1
        if a.size != n:
            raise ValueError(f"{name}: {a.size} elements for shape {e['shape']}")
        out[name] = a.reshape(e["shape"]).copy()
    return out


# ---- writer ------------------------------------------------------------------------------------------------------
class _BlockBuilder:
    def __init__(self):
        self.buf, self.restarts, self.count, self.last = bytearray(), [0], 0, b""

    def add(self, key: bytes, value: bytes):
        shared = 0
        if self.count % RESTART_INTERVAL == 0 and self.count:
            self.restarts.append(len(self.buf))
        elif self.count:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value)) + key[shared:] + value
        self.last, self.count = key, self.count + 1

    def finish(self) -> bytes:
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _emit_block(out: bytearray, contents: bytes):
    handle = _put_varint(len(out)) + _put_varint(len(contents))
    out += contents + b"\x00" + struct.pack("<I", _mask(_crc32c(contents + b"\x00")))
    return handle


def write_checkpoint(prefix, tensors):
    """Write ``tensors`` (name -> array) as a one-shard bundle <prefix>.index / <prefix>.data-00000-of-00001 plus the
    ``checkpoint`` state file next to it, so that tf.train.Saver.restore (and read_checkpoint) can load it."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items = sorted((k.encode(), np.asarray(v, order="C")) for k, v in tensors.items())  # (ascontiguousarray would make 0-d arrays 1-d)
    records, offset = [], 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for key, a in items:
            if a.dtype not in _DTYPE_CODES:
                raise ValueError(f"{key.decode()}: dtype {a.dtype} not supported")
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes()
            f.write(raw)
            records.append((key, _entry_proto(_DTYPE_CODES[a.dtype], a.shape, offset, len(raw), _mask(_crc32c(raw)))))
            offset += len(raw)
    header = b"\x08\x01" + b"\x1a\x02\x08\x01"   # num_shards = 1; endianness LITTLE (default, omitted); version.producer = 1
    out = bytearray()
    index = _BlockBuilder()
    block = _BlockBuilder()
    for key, value in [(b"", header)] + records:
        block.add(key, value)
        if len(block.buf) >= BLOCK_SIZE:
            index.add(block.last, _emit_block(out, block.finish()))
            block = _BlockBuilder()
    if block.count:
        index.add(block.last, _emit_block(out, block.finish()))
    meta_handle = _emit_block(out, _BlockBuilder().finish())
    index_handle = _emit_block(out, index.finish())
    footer = meta_handle + index_handle
    out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
    with open(os.path.join(os.path.dirname(os.path.abspath(prefix)), "checkpoint"), "w") as f:
        base = os.path.basename(prefix)
        f.write(f'model_checkpoint_path: "{base}"\nall_model_checkpoint_paths: "{base}"\n')
