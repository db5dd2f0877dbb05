import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:  # the oracle runs on PyTorch-CPU: more threads than the box's CPU share only oversubscribe it
        import torch

        torch.set_num_threads(min(16, os.cpu_count() or 1))
    except Exception:
        pass


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


@pytest.fixture(scope="session")
def k_oracle_lib():
    """The oracle's C restatement of graph K, built on demand with gcc."""
    import ctypes
    import subprocess

    so = os.path.join(ROOT, "oracle", "_build", "libk_oracle.so")
    src = os.path.join(ROOT, "oracle", "k_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(so)
    lib.k_oracle_f32.restype = ctypes.c_int
    lib.k_oracle_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                 ctypes.c_int]
    return lib
