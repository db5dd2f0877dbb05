"""Graph D through the library's NATIVE executor (csrc/graph_exec.hip: emd_graph_create / _run / _destroy).

``NativeGraph`` is the thin host side a non-Python host would write in its own language: hand the weights over once (host float32
arrays keyed by TensorFlow variable name), then call ``run`` with device buffers.  Layer table, batch-norm folding, weight packing,
kernel selection and the launch sequence all live in the library; this file only moves pointers.  ``DenoiserEngine`` (denoiser.py)
is the Python twin of the same sequence and produces the same bits (tests/test_graph_exec_gpu.py)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class NativeGraph:
    """architecture() behind the C ABI: variant "D" = machine_learning/denoiser.py:58-398, "Dprime" = the training twin
    misc_py/denoiser-multi-gpu.py:200-540 run with phase=False (the graph a trained checkpoint of that script serves), "X" = the
    Xception autoencoder misc_py/modified_Xception.py:194-654 (csrc/graph_exec_x.hip; side a multiple of 64, output in [0,1]), "G" = the
    in-filling generator misc_py/gan-infilling-100.py:133-374 (csrc/graph_exec_g.hip; side a multiple of 16, >= 32, output in (-1,1))."""

    def __init__(self, weights, device, variant="D"):
        import torch

        code = {"D": 0, "Dprime": 1, "X": 2, "G": 3}[variant]

        self.lib = _lib.load()
        self.device = device
        names = list(weights)
        arrays = [np.ascontiguousarray(weights[n], dtype=np.float32) for n in names]   # kept alive during the call
        n = len(names)
        c_names = (C.c_char_p * n)(*[s.encode() for s in names])
        c_data = (C.c_void_p * n)(*[a.ctypes.data for a in arrays])
        c_counts = (C.c_long * n)(*[a.size for a in arrays])
        handle = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(self.lib.emd_graph_create(C.byref(handle), code, n, c_names, c_data, c_counts), "emd_graph_create")
        self._h = handle
        self._ws = None

    def set_two_streams(self, on):
        """The 1/16-resolution flow of an even batch as two halves on two internal streams (same bits; default off)."""
        _lib.check(self.lib.emd_graph_set_two_streams(self._h, 1 if on else 0), "emd_graph_set_two_streams")

    def workspace_bytes(self, B, S):
        return int(self.lib.emd_graph_workspace_bytes(self._h, B, S))

    def forward(self, x):
        """x: torch CUDA float32 [B,S,S,1] contiguous -> [B,S,S,1] (D: no output clip, denoiser.py:396; D': clipped to [0,1])."""
        import torch

        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4 and x.shape[3] == 1 and x.shape[1] == x.shape[2]
        B, S = x.shape[0], x.shape[1]
        need = self.workspace_bytes(B, S)
        if need == 0:
            raise ValueError("square crops with side a multiple of 16 (graph X: 64)")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        y = torch.empty_like(x)
        _lib.check(self.lib.emd_graph_run(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), B, S, C.c_void_p(self._ws.data_ptr()),
                                          C.c_size_t(self._ws.numel()), _lib.stream_ptr()), "emd_graph_run")
        return y

    def close(self):
        if self._h:
            self.lib.emd_graph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
