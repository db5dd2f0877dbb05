"""CPU tests of the drop-in boundary: libemdenoise.so loads, exports every symbol that
include/emdenoise.h declares, the ctypes table covers them all, and argument validation (which
runs before any launch) reports errors through the documented channel.  No compute here."""
import ctypes
import os
import re

import pytest

import emdenoise
from emdenoise import _lib


def declared_symbols(root, header="emdenoise.h"):
    text = open(os.path.join(root, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(emd_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound(repo_root):
    syms = declared_symbols(repo_root)
    assert "emd_kernel_denoise_f32" in syms and "emd_last_error" in syms
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in emdenoise.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in _lib.py"
    for s in _lib.SIGNATURES:
        assert s in syms, f"{s} bound in _lib.py but not declared in emdenoise.h"


def test_every_exported_emd_symbol_is_declared(repo_root):
    """export -> header: the library exports no C symbol with the emd_ prefix that neither header declares (the product ABI is
    include/emdenoise.h; the development hooks with process-global state are fenced off in include/emdenoise_dev.h)."""
    import subprocess

    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("emd_")})
    product = set(declared_symbols(repo_root))
    dev = set(declared_symbols(repo_root, "emdenoise_dev.h"))
    assert dev == set(_lib.DEV_SIGNATURES) and not (dev & product)
    assert all(s.startswith("emd_debug_") for s in dev)
    stray = [s for s in exported if s not in product and s not in dev]
    assert not stray, f"exported but declared in no header: {stray}"
    missing = [s for s in sorted(product | dev) if s not in exported]
    assert not missing, f"declared but not exported: {missing}"


def test_version_and_params_count():
    lib = _lib.load()
    assert lib.emd_version() == 100
    assert lib.emd_kernel_params_count(3, 2) == 2 * 2 * 9 + 2
    assert lib.emd_kernel_params_count(0, 2) == 0


def test_argument_validation_needs_no_gpu():
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    one = ctypes.c_void_p(16)
    two = ctypes.c_void_p(32)
    rc = lib.emd_kernel_denoise_f32(null, one, 1, 8, 8, 3, 1, one, 0, null)
    assert rc == -1 and b"null" in lib.emd_last_error()
    rc = lib.emd_kernel_denoise_f32(one, two, 1, 8, 8, 4, 1, one, 0, null)       # even width
    assert rc == -1 and b"width" in lib.emd_last_error()
    rc = lib.emd_kernel_denoise_f32(one, two, 1, 8, 8, 3, 9, one, 0, null)       # depth too large
    assert rc == -1
    rc = lib.emd_kernel_denoise_f32(one, two, 1, 1, 8, 3, 1, one, 0, null)       # pad >= H
    assert rc == -1 and b"REFLECT" in lib.emd_last_error()
    rc = lib.emd_kernel_denoise_f32(one, one, 1, 8, 8, 3, 1, one, 0, null)       # aliasing
    assert rc == -1
    rc = lib.emd_kernel_denoise_f32(one, two, 0, 8, 8, 3, 1, one, 0, null)       # empty batch: no-op
    assert rc == 0
    with pytest.raises(_lib.EmdError):
        _lib.check(-1, "x")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.EmdError, match="no CPU fallback"):
        _lib.load()


def test_host_param_packing():
    import numpy as np

    p = emdenoise.KernelParams.from_symmetric([[1, 2, 3], [4, 5, 6]], [[0, 0, 0], [7, 8, 9]], [1.0, 0.5], 3)
    assert p.symmetric and p.depth == 2 and p.width == 3
    blk = p.packed()
    assert blk.shape == (38,)
    np.testing.assert_array_equal(blk[:9].reshape(3, 3), [[3, 2, 3], [2, 1, 2], [3, 2, 3]])
    np.testing.assert_array_equal(blk[27:36].reshape(3, 3), [[9, 8, 9], [8, 7, 8], [9, 8, 9]])
    assert blk[36] == 1.0 and blk[37] == 0.5
    w = np.arange(9, dtype=np.float32).reshape(1, 3, 3)
    assert not emdenoise.KernelParams(w, np.zeros_like(w), np.ones(1, np.float32)).symmetric


def test_pack_job_struct_layout_matches_the_header():
    """emd_pack_job_fill is host-only: the ctypes mirror of emd_pack_job_t (emdenoise._lib.PackJob) must land every field where the
    C struct has it (a mismatch would put garbage pointers into the job table of emd_pack_weights_batch_dev)."""
    import ctypes as C

    from emdenoise import _lib

    lib = _lib.load()
    job = _lib.PackJob()
    sel = (C.c_int * 2)(1, 0)
    rc = lib.emd_pack_job_fill(C.byref(job), C.c_void_p(0x1000), 3, 2, sel, 100, 72, 1, C.c_void_p(0x2000), C.c_void_p(0x3000))
    assert rc == 0
    assert (job.w, job.hi, job.lo) == (0x1000, 0x2000, 0x3000)
    assert job.tap_sel == (1 | (0 << 4)) and (job.ntaps, job.cin, job.cout, job.cout_major) == (2, 100, 72, 1)
    # (cout_major: a thread packs four consecutive elements -- round 4)
    assert job.cpad == 128 and job.total == 128 * 2 * 128 and job.n_blocks == (job.total // 4 + 255) // 256 and job.first_block == 0
    # the transposing orientation: 16 (n) x 64 (c) tiles per tap
    rc = lib.emd_pack_job_fill(C.byref(job), C.c_void_p(0x1000), 3, 2, sel, 100, 72, 0, C.c_void_p(0x2000), C.c_void_p(0x3000))
    assert rc == 0 and job.cout_major == 0 and job.n_blocks == 2 * ((job.total // (2 * job.cpad) + 15) // 16) * ((job.cpad + 63) // 64)
    rc = lib.emd_pack_job_fill(C.byref(job), C.c_void_p(0x1000), 3, 2, sel, 100, 72, 1, C.c_void_p(0x2000), C.c_void_p(0x3000))
    assert rc == 0
    assert C.sizeof(_lib.PackJob) == 80
    # a tap subset needs tap_sel; out-of-range selections are refused
    assert lib.emd_pack_job_fill(C.byref(job), C.c_void_p(0x1000), 3, 2, None, 100, 72, 1, C.c_void_p(0x2000), C.c_void_p(0x3000)) != 0
    bad = (C.c_int * 2)(3, 0)
    assert lib.emd_pack_job_fill(C.byref(job), C.c_void_p(0x1000), 3, 2, bad, 100, 72, 1, C.c_void_p(0x2000), C.c_void_p(0x3000)) != 0


def test_batched_groups_rule():
    from emdenoise.trainer import DenoiserTrainer as T

    assert [T.batched_groups(b) for b in (1, 2, 4, 6, 8, 12, 16)] == [1, 1, 2, 2, 4, 4, 4]
