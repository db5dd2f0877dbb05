"""The native graph executor (csrc/graph_exec.hip: emd_graph_create / _workspace_bytes / _run / _destroy) against the
Python engine and the oracle.  The executor re-derives everything from the TensorFlow-named weights on its own (layer table, folded
batch norms, packed weights, kernel selection, launch order, workspace planning), so bit-identity with DenoiserEngine checks all of it."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.synth_inputs import synthetic_lq


def test_create_reports_a_missing_or_misshapen_variable():
    """Argument validation runs before any device work for the first missing name (no GPU needed for this branch)."""
    import emdenoise
    from emdenoise import _lib

    lib = _lib.load()
    w = emdenoise.synthetic_weights()
    names = [n for n in w if not n.endswith("/gamma")][:5]
    arrays = [np.ascontiguousarray(w[n], np.float32) for n in names]
    h = C.c_void_p()
    rc = lib.emd_graph_create(C.byref(h), 4, len(names), (C.c_char_p * 5)(*[n.encode() for n in names]),
                              (C.c_void_p * 5)(*[a.ctypes.data for a in arrays]), (C.c_long * 5)(*[a.size for a in arrays]))
    assert rc == -2 and b"variant" in lib.emd_last_error()          # 0 = graph D, 1 = graph D', 2 = graph X, 3 = graph G's generator
    assert lib.emd_graph_workspace_bytes(None, 1, 64) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("B,S", [(2, 64), (1, 48), (4, 512), (16, 512)])   # 16 x 512^2: set_two_streams runs the 1/16-resolution flow as two halves
def test_native_graph_equals_the_python_engine(B, S):
    import emdenoise
    from emdenoise.graph_exec import NativeGraph

    dev = torch.device("cuda", 0)
    w = emdenoise.synthetic_weights()
    eng = emdenoise.DenoiserEngine(w, dev, "bf16x3")
    nat = NativeGraph(w, dev)
    x = torch.from_numpy(synthetic_lq(B, S, S, seed=70 + S)).to(dev)
    want = eng.forward(x)
    got = nat.forward(x)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    again = nat.forward(x)                                           # the workspace is reused; nothing stale in it matters
    torch.cuda.synchronize()
    assert torch.equal(again, want)
    nat.set_two_streams(True)                                        # launch-order option: same bits (taken at 16 x 512^2, a no-op below)
    two = nat.forward(x)
    torch.cuda.synchronize()
    assert torch.equal(two, want)
    assert nat.workspace_bytes(B, S) > 0
    nat.close()


@pytest.mark.gpu
def test_native_graph_matches_the_oracle_and_reports_errors():
    import emdenoise
    from emdenoise import _lib
    from emdenoise.graph_exec import NativeGraph
    from oracle import denoiser_graph as G

    dev = torch.device("cuda", 0)
    w = emdenoise.synthetic_weights()
    nat = NativeGraph(w, dev)
    x = synthetic_lq(2, 64, 64, seed=5)
    ref = G.architecture(x, w, 64, dtype=torch.float64).numpy()
    got = nat.forward(torch.from_numpy(x).to(dev)).cpu().numpy()
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 3e-4
    # a workspace that is too small is reported, not overrun
    xs = torch.from_numpy(x).to(dev)
    small = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    rc = nat.lib.emd_graph_run(nat._h, C.c_void_p(xs.data_ptr()), C.c_void_p(torch.empty_like(xs).data_ptr()), 2, 64, C.c_void_p(small.data_ptr()),
                               C.c_size_t(small.numel()), _lib.stream_ptr())
    assert rc != 0 and b"workspace" in nat.lib.emd_last_error()
    torch.cuda.synchronize()
    bad = dict(w)
    del bad["nn/Conv_3/biases"]
    with pytest.raises(_lib.EmdError, match="missing variable nn/Conv_3/biases"):
        NativeGraph(bad, dev)
    nat.close()


@pytest.mark.gpu
@pytest.mark.parametrize("B,S", [(2, 64), (8, 512)])   # 8 x 512^2: the dense ASPP branches take the split32 implicit GEMM
def test_native_graph_dprime_equals_the_python_engine(B, S):
    """variant 1: graph D' (misc_py/denoiser-multi-gpu.py:200-540, phase=False) -- tf.layers variable names, dense dilated 3x3 ASPP
    branches, the image-level branch, the in-graph clip -- bit for bit against DenoiserEngine(variant="Dprime")."""
    import emdenoise
    from emdenoise.graph_exec import NativeGraph

    dev = torch.device("cuda", 0)
    w = emdenoise.synthetic_weights(variant="Dprime")
    eng = emdenoise.DenoiserEngine(w, dev, "bf16x3", variant="Dprime")
    nat = NativeGraph(w, dev, variant="Dprime")
    x = torch.from_numpy(synthetic_lq(B, S, S, seed=90 + S)).to(dev)
    want = eng.forward(x)
    got = nat.forward(x)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert float(got.min()) >= 0.0 and float(got.max()) <= 1.0
    nat.close()


@pytest.mark.gpu
@pytest.mark.parametrize("B,S", [(2, 128), (1, 64), (4, 512)])
def test_native_graph_x_equals_the_python_engine(B, S):
    """emd_graph_create(variant 2): graph X (misc_py/modified_Xception.py:194-654) from C -- layer table under scope "pellet", folded
    moving-statistics norms, packed weights, the per-layer choice between the fp32 / split32 / patch-resident kernels with its
    look-ahead, batch-statistics norms folded on the device, workspace planning -- bit-identical to XceptionEngine (split-bf16 mode),
    and a workspace that is too small is reported."""
    from emdenoise import _lib, xception
    from emdenoise.graph_exec import NativeGraph

    dev = torch.device("cuda", 0)
    w = xception.synthetic_weights()
    eng = xception.XceptionEngine(w, dev, "bf16x3")
    nat = NativeGraph(w, dev, variant="X")
    x = torch.from_numpy(synthetic_lq(B, S, S, seed=90 + S)).to(dev)
    want = eng.forward(x)
    got = nat.forward(x)
    torch.cuda.synchronize()
    assert torch.isfinite(got).all() and float(got.min()) >= 0.0 and float(got.max()) <= 1.0
    assert torch.equal(got, want)
    again = nat.forward(x)
    torch.cuda.synchronize()
    assert torch.equal(again, want)
    need = nat.workspace_bytes(B, S)
    assert need > 0 and nat.workspace_bytes(B, 48) == 0            # graph X: side a multiple of 64
    small = torch.empty(need // 2, dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    rc = nat.lib.emd_graph_run(nat._h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), B, S, C.c_void_p(small.data_ptr()),
                               C.c_size_t(small.numel()), None)
    assert rc != 0 and b"workspace" in _lib.load().emd_last_error()
    nat.close()


@pytest.mark.gpu
@pytest.mark.parametrize("B,S", [(2, 64), (1, 96), (4, 512)])
def test_native_graph_g_equals_the_python_engine(B, S):
    """emd_graph_create(variant 3): the in-filling generator (misc_py/gan-infilling-100.py:133-374) from C -- layer table under
    "GAN/Gen" / "GAN/Gen/reg", both norms of every separable conv folded, the four routes of a separable block (fused, fused stride 2 on
    the REFLECT-padded image, split32 pair, register-staged pair), the generated-input first layers, resizes, the reflect-padded last
    conv and the instance norm -- bit-identical to GeneratorEngine (split-bf16 mode; its two-halves streams give the same bits)."""
    from emdenoise import gan
    from emdenoise.graph_exec import NativeGraph

    dev = torch.device("cuda", 0)
    w = gan.synthetic_weights()
    eng = gan.GeneratorEngine(w, dev, "bf16x3")
    nat = NativeGraph(w, dev, variant="G")
    hq = 2.0 * synthetic_lq(B, S, S, seed=95 + S)[..., 0] - 1.0
    x = torch.from_numpy(gan.gen_lq(hq, select=gan.spiral_mask(S))[..., None]).to(dev)
    want = eng.forward(x)
    got = nat.forward(x)
    torch.cuda.synchronize()
    assert torch.isfinite(got).all() and float(got.abs().max()) <= 1.0
    assert torch.equal(got, want)
    assert torch.equal(nat.forward(x), want)
    nat.close()
