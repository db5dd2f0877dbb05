"""Dev tool (run by tools/asan_build.sh --run under AddressSanitizer + UBSan): the host-only code paths of libemdenoise.so that need no GPU --
weight packing in every orientation and shape class, the native graph executor's layer tables / batch-norm folding / packing up to the
point where it would upload (no device: creation fails cleanly), the TFRecord reader / writer with CRC-32C, argument validation of every
entry point with a stream.  Anything the sanitizers find aborts the process."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import emdenoise
from emdenoise import _lib, input_pipeline as IP

lib = _lib.load()
print("library:", _lib.LIB_PATH)
rng = np.random.default_rng(0)
# 1. emd_pack_weights_bf16: taps 1 / 9, both orientations, channel counts around the 32 / 64 / 128 padding units
for taps in (1, 9):
    for cin, cout in ((1, 64), (4, 4), (31, 33), (64, 64), (728, 728), (3640, 256), (130, 260)):
        for cm in (0, 1):
            w = rng.standard_normal((taps, cout, cin) if cm else (taps, cin, cout)).astype(np.float32)
            n = lib.emd_packed_weight_elems(taps, cin, cout)
            hi, lo = np.empty(n, np.uint16), np.empty(n, np.uint16)
            rc = lib.emd_pack_weights_bf16(w.ctypes.data, taps, cin, cout, cm, hi.ctypes.data, lo.ctypes.data)
            assert rc == 0, (taps, cin, cout, cm, lib.emd_last_error())
print("pack_weights ok")
# 2. native graph executor: creation walks the 658-name table, folds and packs on the host, then tries to upload
for variant, code in (("D", 0), ("Dprime", 1)):
    from emdenoise import denoiser as DN
    weights = DN.synthetic_weights(variant=variant)
    names = list(weights)
    arrays = [np.ascontiguousarray(weights[n], dtype=np.float32) for n in names]
    n = len(names)
    c_names = (C.c_char_p * n)(*[s_.encode() for s_ in names])
    c_data = (C.c_void_p * n)(*[a_.ctypes.data for a_ in arrays])
    c_counts = (C.c_long * n)(*[a_.size for a_ in arrays])
    handle = C.c_void_p()
    rc = lib.emd_graph_create(C.byref(handle), code, n, c_names, c_data, c_counts)
    print(f"graph {variant}: emd_graph_create -> {rc} ({lib.emd_last_error().decode()[:90]})")
    if rc == 0:
        print("   workspace bytes at B=2, S=64:", lib.emd_graph_workspace_bytes(handle, 2, 64))
        lib.emd_graph_destroy(handle)
    # a wrong table (one array short) must be refused, not read past
    rc = lib.emd_graph_create(C.byref(handle), code, n - 1, c_names, c_data, c_counts)
    print(f"graph {variant} with a missing variable -> {rc}")
# 3. TFRecord writer / reader (CRC-32C in host_utils.cpp)
import tempfile
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "t.tfrecord")
    imgs = [rng.random((17, 23)).astype(np.float32), rng.random((64, 64)).astype(np.float32), np.zeros((1, 1), np.float32)]
    IP.write_tfrecord(path, imgs)
    back = list(IP.tfrecord_images(path, shape=None)) if False else [np.frombuffer(IP.parse_example(r)["image"][0], np.float32) for r in IP.read_tfrecord(path)]
    assert len(back) == 3 and all(np.array_equal(a.reshape(-1), b) for a, b in zip(imgs, back))
print("tfrecord ok")
# 4. argument validation of the fused separable entry points (host code before any launch)
one, two = C.c_void_p(64), C.c_void_p(128)
rc = lib.emd_sep3x3_fused_f32(one, 64, one, one, one, one, one, None, None, None, 0, two, 64, 1, 8, 30, 64, 64, 1, 3, None)
assert rc != 0
print("validation ok; last error:", lib.emd_last_error().decode()[:80])
