#!/usr/bin/env python3
"""bench.py -- megapixels/s restored on synthetic 512x512x1 micrograph batches (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload K|D|X|T|both|all] [--batch B]

A "step" is one pass of the hot path over one batch that is already resident in HBM.
Workloads:
  K  BASELINE configs[1]: the 3-layer 3x3 kernel denoiser (misc_py/noise-removal-kernels.py, depth 2,
     width 3) on [32,512,512,1].  This is the configuration the metric line is quoted on.
  D  BASELINE configs[2]: the modified-Xception encoder-decoder of machine_learning/denoiser.py on
     [32,512,512,1] (matrix cores in split-bf16 parity mode unless --precision bf16).
  X  the other graph BASELINE configs[2] can mean: misc_py/modified_Xception.py at 512x512.
  G  BASELINE configs[4], the part that is built: the in-filling GAN's generator forward pass
     (misc_py/gan-infilling-100.py:133-374) on [32,512,512,1]; rides along as "workload_G".
  A  BASELINE configs[4]: one iteration of the in-filling GAN's adversarial training loop (generator towers through
     the discriminator + Adam, discriminator towers + Adam) on --gan-batch images per GPU; rides along as "workload_A".
  T  BASELINE configs[3]: graph D' TRAINING (misc_py/denoiser-multi-gpu.py): --train-batch LQ/HQ pairs per GPU per
     step (default 8 = bs 64 over 8 GPUs), towers of --tower-batch images (default 1, the reference), one RCCL
     all-reduce of the flat gradient vector per step, Nesterov step.  Rides along as "workload_T"; --workload T
     makes it the primary line (metric "megapixels/sec trained").
Default ("all"): the JSON line's metric/value/roofline/cpu_baseline are workload K's; workload D's and X's
figures ride along under "workload_D" / "workload_X".  --workload D makes D the primary line.
For N > 1 the driver launches one rank per GPU (torch.distributed.run); inference shards whole images
across ranks with no data-path collective (weak scaling: --batch images PER GPU).  Rank 0 prints ONE
JSON line.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CPU_THREADS = min(16, os.cpu_count() or 1)  # a 1-GPU box owns a 16-CPU share; more threads only oversubscribe it
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA
D_GMAC_MATRIX_B32_512 = 2218.3  # pointwise 1304.5 + dense 1x1 295.3 - final 4.8 + conv-T 618.5 ... per B=32 batch


def pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC run (tools/collect_traffic.sh: FETCH_SIZE and
    WRITE_SIZE in separate passes, gfx950 correction applied).  bench.py cannot profile itself, so `traffic`
    is the figure of that run for the same kernels and shapes; None if the file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        return json.load(open(path))["kernels"]
    except Exception:
        return None


def synthetic_lq(B, H, W, seed=1234):
    """Synthetic low-quality crops of the reference's shape and statistics: smooth field ->
    Poisson counts (scale = 25 + Exp(75), denoiser-multi-gpu.py:783-799) -> min-max to [0,1]."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    base = []
    for i in range(min(B, 4)):
        hq = np.zeros((H, W), np.float32)
        for _ in range(8):
            cy, cx, s = rng.uniform(0, H), rng.uniform(0, W), rng.uniform(8, 64)
            hq += np.float32(rng.uniform(0.2, 1.0)) * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / np.float32(2 * s * s))
        hq = (hq - hq.min()) / max(float(hq.max() - hq.min()), 1e-9)
        lq = rng.poisson(hq * (25.0 + rng.exponential(75.0))).astype(np.float32)
        base.append((lq - lq.min()) / max(float(lq.max() - lq.min()), 1e-9))
    out = np.stack([base[i % len(base)] for i in range(B)])[..., None].astype(np.float32)
    out += rng.random(out.shape, dtype=np.float32) * np.float32(1e-3)  # no two images identical
    return np.clip(out, 0.0, 1.0)


# ------------------------------------------------------------------------------------------------
# CPU baselines: the ORACLE timed on the host cores (reported next to the GPU number; never shipped)
# ------------------------------------------------------------------------------------------------
def cpu_baseline_K(x_host, W, Bm, s, budget_s=10.0):
    """oracle/k_oracle.c (plain-C port of graph K, OpenMP over rows)."""
    import subprocess

    so = os.path.join(ROOT, "oracle", "_build", "libk_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    lib.k_oracle_f32.restype = ctypes.c_int
    lib.k_oracle_f32.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3 + [ctypes.c_int]
    lib.k_oracle_max_threads.restype = ctypes.c_int
    cores = min(lib.k_oracle_max_threads(), CPU_THREADS)
    x = np.ascontiguousarray(x_host[..., 0])
    y = np.empty_like(x)
    B, H, Wd = x.shape

    def run():
        assert lib.k_oracle_f32(x.ctypes.data, y.ctypes.data, B, H, Wd, W.shape[1], W.shape[0], W.ctypes.data,
                                Bm.ctypes.data, s.ctypes.data, cores) == 0

    run()
    reps, t0 = 0, time.perf_counter()
    while True:
        run()
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= 2000:
            break
    return {"value": round(B * H * Wd / 1e6 * reps / el, 2), "unit": "MPx/s", "cores": cores, "kind": "port",
            "sample": f"{reps} passes over the same [{B},{H},{Wd},1] batch, oracle/k_oracle.c (gcc -O3 -fopenmp), {el:.1f} s"}, y


def cpu_baseline_D(x_host, weights):
    """oracle/denoiser_graph.py (PyTorch-CPU float32 restatement of the TF graph) on ONE 512x512 image,
    which mirrors the reference's own batch shape (denoiser.py:613)."""
    import torch

    from oracle import denoiser_graph as G

    cores = CPU_THREADS
    torch.set_num_threads(cores)
    S = x_host.shape[1]
    t0 = time.perf_counter()
    y = G.architecture(x_host[:1], weights, S, dtype=torch.float32).numpy()   # also the parity reference for image 0
    t1 = time.perf_counter() - t0
    n = int(max(1, min(len(x_host), round(12.0 / max(t1, 1e-3)))))             # ~12 s of CPU work
    t0 = time.perf_counter()
    G.architecture(x_host[:n], weights, S, dtype=torch.float32)
    el = time.perf_counter() - t0
    return {"value": round(n * S * S / 1e6 / el, 4), "unit": "MPx/s", "cores": cores, "kind": "port",
            "sample": f"1 pass over the first {n} images of the batch ([{n},{S},{S},1]), oracle/denoiser_graph.py "
                      f"(PyTorch-CPU float32, {torch.get_num_threads()} threads), {el:.1f} s"}, y


# ------------------------------------------------------------------------------------------------
class Timer:
    """Barrier + synchronize on both sides, exactly `steps` steps, max over ranks."""

    def __init__(self, torch, dist, dev):
        self.torch, self.dist, self.dev = torch, dist, dev

    def sync(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def run(self, step, steps, warmup):
        for _ in range(warmup):
            step()
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.sync()
        wall = time.perf_counter() - t0
        if self.dist is not None:
            tt = self.torch.tensor([wall], dtype=self.torch.float64, device=self.dev)
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
            wall = float(tt.item())
        return wall * 1e3 / steps


def bench_K(a, torch, emdenoise, dev, timer, rank, world, want_cpu):
    B, H, W = a.batch, a.size, a.size
    steps = a.steps if a.steps is not None else 200
    warmup = a.warmup if a.warmup is not None else 20
    x_host = synthetic_lq(B, H, W, seed=1234 + rank)
    x_host = (x_host / np.maximum(x_host.mean(axis=(1, 2, 3), keepdims=True), 1e-9)).astype(np.float32)  # noise-removal-kernels.py:525-527
    rng = np.random.default_rng(7)
    pairs = emdenoise.kernel_denoiser.sym_pairs(3)
    wsc = [(rng.standard_normal(len(pairs)) * 0.15 + 1.0 / 9).astype(np.float32) for _ in range(2)]
    bsc = [np.zeros(len(pairs), np.float32), (rng.standard_normal(len(pairs)) * 0.5).astype(np.float32)]
    params = emdenoise.KernelParams.from_symmetric(wsc, bsc, [1.0, 1.3], 3)
    pd = torch.from_numpy(params.packed()).to(dev)
    x = torch.from_numpy(x_host).to(dev)
    y = torch.empty_like(x)

    def step():
        emdenoise.kernel_denoise(x, pd, 3, 2, params.symmetric, out=y)

    ms = timer.run(step, steps, warmup)
    # the dominant (only) kernel, timed live with HIP events on the launch stream: torch's current stream IS
    # the stream handed to the C ABI.  One event pair around a back-to-back burst cancels the per-event cost.
    n_burst = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n_burst):
            step()
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    launch_us = e0.elapsed_time(e1) * 1e3 / n_burst
    alg_bytes = 8.0 * B * H * W  # 8 B per pixel (read 4 + write 4)
    achieved = alg_bytes / (launch_us * 1e-6) / 1e9
    tr = pmc_traffic()
    traffic = None
    if tr and "K:k3_roll<8, 2>" in tr and (B, H, W) == (32, 512, 512):
        traffic = round(tr["K:k3_roll<8, 2>"]["hbm_bytes_per_launch_corrected"])
    out = {
        "value": B * H * W / 1e6 * world / (ms / 1e3), "ms_per_step": ms, "steps": steps, "warmup": warmup, "dtype": "f32",
        "config": {"workload": f"K: kernel denoiser depth 2 width 3 (noise-removal-kernels.py), [{B},{H},{W},1] fp32 per GPU",
                   "global_batch": B * world, "image": f"{H}x{W}x1", "sharding": f"{world} x {B} whole images, no collective"},
        "roofline": {"bound": "hbm", "kernel": "k3_roll<8,2>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                     "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, 2x FETCH correction)",
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "avg_launch_us": round(launch_us, 3),
                     "how": f"HIP events around a hipGraph of {n_burst} back-to-back launches (includes the ~1.5 us kernel boundary)"},
    }
    # the same kernel on a batch that is not launch-bound and does not fit the 256 MiB Infinity Cache (SURVEY.md 8d:
    # "the kernel is launch/latency-limited at this size, so also report B=256"): 8x the images, one launch
    if (B, H, W) == (32, 512, 512) and rank == 0:
        try:
            xb = x.repeat(8, 1, 1, 1).contiguous()
            yb = torch.empty_like(xb)
            for _ in range(3):
                emdenoise.kernel_denoise(xb, pd, 3, 2, params.symmetric, out=yb)
            eb0, eb1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            eb0.record()
            for _ in range(20):
                emdenoise.kernel_denoise(xb, pd, 3, 2, params.symmetric, out=yb)
            eb1.record()
            torch.cuda.synchronize()
            us256 = eb0.elapsed_time(eb1) * 1e3 / 20
            out["roofline"]["batch256"] = {"avg_launch_us": round(us256, 2), "achieved": round(8 * alg_bytes / (us256 * 1e-6) / 1e9, 1),
                                           "frac": round(8 * alg_bytes / (us256 * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                                           "note": "[256,512,512,1]: 537 MB per launch, HIP events around 20 launches"}
            del xb, yb
        except Exception as e:  # out of memory on a shared card: the primary figure stands on its own
            out["roofline"]["batch256"] = {"error": f"{type(e).__name__}: {e}"}
    if want_cpu:
        cb, y_cpu = cpu_baseline_K(x_host, params.wmaps, params.bmaps, params.s)
        out["cpu_baseline"] = cb
        y_gpu = y.cpu().numpy()[..., 0].astype(np.float64)
        out["rel_l2_vs_oracle"] = float(f"{np.linalg.norm(y_gpu - y_cpu) / np.linalg.norm(y_cpu):.3e}")
    return out


from importlib import import_module as _imp


class _LazyGraphed:
    def __call__(self, eng):
        return _imp('emdenoise.graphed').GraphedForward(eng)


GraphedForward = _LazyGraphed()


def bench_D(a, torch, emdenoise, dev, timer, rank, world, want_cpu):
    from emdenoise import ops

    B, H, W = a.batch, a.size, a.size
    steps = a.steps if (a.steps is not None and a.workload == "D") else 10
    warmup = a.warmup if (a.warmup is not None and a.workload == "D") else 2
    x_host = synthetic_lq(B, H, W, seed=1234 + rank)
    weights = emdenoise.synthetic_weights()
    eng = emdenoise.DenoiserEngine(weights, dev, a.precision)
    x = torch.from_numpy(x_host).to(dev)
    box = [None]

    def step():
        box[0] = eng.forward(x)

    ms = timer.run(step, steps, warmup)
    # per-kernel-family device time of ONE more step, HIP events around every launch of the family
    fam = {}
    orig = {}
    dw_bytes_box = [0.0]
    pw_flops_box = [0.0]

    def wrap(name):
        f = getattr(ops, name)
        orig[name] = f

        def g(*args, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = f(*args, **kw)
            e1.record()
            fam.setdefault(name, []).append((e0, e1))
            if name in ("dw3x3", "dw3x3_split32"):  # algorithmic bytes of THIS launch: fp32 in + out;
                xin, out = args[0], args[2]          # a split32 output has the bytes of its fp32 twin
                dw_bytes_box[0] += 4.0 * xin.C * (xin.B * xin.H * xin.W + out.B * out.H * out.W)
            if name == "conv1x1_split32":            # issued flops of THIS pointwise launch (3 bf16 MFMA passes)
                xin, wgt = args[0], args[1]
                pw_flops_box[0] += 6.0 * xin.B * xin.H * xin.W * wgt.cin * wgt.cout
            return r

        setattr(ops, name, g)

    for name in ("conv1x1", "conv3x3", "deconv3x3s2", "sep_fused", "dw3x3", "cin1", "conv3x3_cout1", "resize_bilinear",
                 "affine_relu6", "affine_act", "bn_batch_stats", "avgpool2x2", "conv1x1_split32", "conv3x3_split32",
                 "deconv3x3s2_split32", "dw3x3_split32", "to_split32"):
        wrap(name)
    two = eng.two_streams
    eng.two_streams = False   # attribution pass on the single-stream launch sequence: with the two half batches on two streams
    try:                      # (streams.TwoHalves) a launch's event pair would also span the other half's kernels
        step()
        torch.cuda.synchronize()
    finally:
        eng.two_streams = two
        for name, f in orig.items():
            setattr(ops, name, f)
    fam_ms = {k: sum(e0.elapsed_time(e1) for e0, e1 in v) for k, v in fam.items()}
    # matrix-core time: the implicit-GEMM launches plus the fused separable convs (whose pointwise halves carry
    # part of the algorithmic flops)
    gemm_ms = sum(fam_ms.get(k, 0.0) for k in ("conv1x1", "conv3x3", "deconv3x3s2", "sep_fused", "conv1x1_split32",
                                               "conv3x3_split32", "deconv3x3s2_split32"))
    dw_ms = fam_ms.get("dw3x3", 0.0) + fam_ms.get("dw3x3_split32", 0.0)
    scale = (B / 32.0) * (H * W) / (512.0 * 512.0)
    alg_flops = 2.0 * D_GMAC_MATRIX_B32_512 * 1e9 * scale
    achieved = alg_flops / (gemm_ms * 1e-3) / 1e12
    passes = 3 if a.precision == "bf16x3" else 1
    dw_bytes = dw_bytes_box[0]  # only the STANDALONE depthwise launches (the fused layers never write the depthwise result)
    tr = pmc_traffic()
    traffic = None
    if tr and (B, H, W) == (32, 512, 512):  # HBM bytes of every matrix-core launch of one step
        traffic = round(sum(v["hbm_bytes_per_launch_corrected"] * v["launches_sampled"] / 4.0
                            for k, v in tr.items() if k.startswith(("D:gemm_conv_kernel", "D:gemm_split", "D:sep_fused"))))
    out = {
        "value": B * H * W / 1e6 * world / (ms / 1e3), "ms_per_step": ms, "steps": steps, "warmup": warmup,
        "dtype": "bf16x3 (split-bf16 MFMA inputs, fp32 accumulate and activations)" if passes == 3 else "bf16",
        "config": {"workload": f"D: modified-Xception encoder-decoder (machine_learning/denoiser.py), [{B},{H},{W},1] fp32 per GPU",
                   "global_batch": B * world, "image": f"{H}x{W}x1", "precision": a.precision,
                   "sharding": f"{world} x {B} whole images, no collective"},
        "roofline": {"bound": "mfma", "kernel": "gemm_conv_kernel + gemm_split*_kernel + sep_fused_kernel (every matrix-core launch of one step)",
                     "achieved": round(achieved, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic,
                     "traffic_note": "HBM bytes per step over the family (PMC run of 4 forwards, profiles/r01_pmc_traffic.json)",
                     "algorithmic_flops_per_step": alg_flops, "mfma_passes": passes,
                     "issued_tflops": round(achieved * passes, 1),
                     "kernel_ms_per_step": round(gemm_ms, 3),
                     "how": "HIP events around every launch of the family in one extra step (single-stream launch sequence)"},
        "depthwise": {"bound": "hbm", "kernel": "dw3x3_s1_roll / dw3x3_generic, fp32 or split32 output (standalone launches only)",
                      "algorithmic_bytes_per_step": dw_bytes, "ms_per_step": round(dw_ms, 3),
                      "achieved_GBps": round(dw_bytes / (max(dw_ms, 1e-9) * 1e-3) / 1e9, 1),
                      "frac_of_8TBps": round(dw_bytes / (max(dw_ms, 1e-9) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
        "pointwise": {"bound": "mfma", "kernel": "gemm_split_kernel<256,3,true> (the 1x1 halves of the 728-channel separable convs, LDS-DMA from split32)",
                      "launches": len(fam.get("conv1x1_split32", [])), "issued_flops_per_step": pw_flops_box[0],
                      "ms_per_step": round(fam_ms.get("conv1x1_split32", 0.0), 3),
                      "issued_tflops": round(pw_flops_box[0] / (max(fam_ms.get("conv1x1_split32", 0.0), 1e-9) * 1e-3) / 1e12, 1),
                      "frac_of_2500": round(pw_flops_box[0] / (max(fam_ms.get("conv1x1_split32", 0.0), 1e-9) * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)},
        "kernel_family_ms": {k: round(v, 3) for k, v in sorted(fam_ms.items(), key=lambda kv: -kv[1])},
    }
    # the headline pointwise GEMM on its own (32768 x 728 x 728 at B = 32: 40 of D's layers), interleaved rounds in this
    # process: the default kernel, and the same pipeline on 16x16x32 MFMAs (opt-in: it sums a K step in another order)
    if (B, H, W) == (32, 512, 512) and rank == 0 and a.precision == "bf16x3":
        try:
            from emdenoise import _lib as _L
            lib = _L.load()
            xa = ops.Act(torch.rand(B, 32, 32, 728, device=dev) * 2)
            xs = ops.to_split32(xa)
            wpk = ops.PackedWeights((np.random.default_rng(0).standard_normal((1, 728, 728)) * 0.05).astype(np.float32), False, dev)
            one, zero = torch.ones(728, device=dev), torch.zeros(728, device=dev)
            o = ops.Act.empty(B, 32, 32, 728, dev)
            tms = {-1: [], 5: []}
            for _ in range(3):
                for v in (-1, 5):
                    lib.emd_debug_split_variant(v)
                    ops.conv1x1_split32(xs, wpk, one, zero, o)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(20):
                        ops.conv1x1_split32(xs, wpk, one, zero, o)
                    e1.record()
                    torch.cuda.synchronize()
                    tms[v].append(e0.elapsed_time(e1) * 1e3 / 20)
            lib.emd_debug_split_variant(-1)
            fl = 6.0 * B * 1024 * 728 * 728
            iso = {}
            for v, nm in ((-1, "default_32x32x16"), (5, "variant_16x16x32")):
                us = float(np.median(tms[v]))
                iso[nm] = {"us": round(us, 1), "issued_tflops": round(fl / us / 1e6, 1), "frac_of_2500": round(fl / us / 1e6 / MFMA_BF16_PEAK_TFLOPS, 4)}
            out["pointwise"]["isolated_32768x728x728"] = iso
        except Exception as e:
            out["pointwise"]["isolated_32768x728x728"] = {"error": f"{type(e).__name__}: {e}"}
    if want_cpu:
        cb, y_cpu = cpu_baseline_D(x_host, weights)
        out["cpu_baseline"] = cb
        y_gpu = box[0][:1].cpu().numpy().astype(np.float64)
        out["rel_l2_vs_oracle"] = float(f"{np.linalg.norm(y_gpu - y_cpu) / np.linalg.norm(y_cpu):.3e}")
    return out


def bench_X(a, torch, emdenoise, dev, timer, rank, world, want_cpu):
    """BASELINE configs[2], second graph of that name: misc_py/modified_Xception.py at 512x512."""
    from emdenoise import xception as X

    B, H, W = a.batch, a.size, a.size
    steps, warmup = 3, 1
    x_host = synthetic_lq(B, H, W, seed=1234 + rank)
    weights = X.synthetic_weights()
    eng = X.XceptionEngine(weights, dev, a.precision)
    x = torch.from_numpy(x_host).to(dev)
    box = [None]

    def step():
        box[0] = eng.forward(x)

    ms = timer.run(step, steps, warmup)
    scale = (B / 32.0) * (H * W) / (512.0 * 512.0)
    out = {"value": round(B * H * W / 1e6 * world / (ms / 1e3), 1), "unit": "MPx/s", "ms_per_step": round(ms, 3),
           "steps": steps, "warmup": warmup,
           "config": {"workload": f"X: Xception autoencoder (misc_py/modified_Xception.py), [{B},{H},{W},1] fp32 per GPU",
                      "precision": a.precision, "algorithmic_tflop_per_step": round(9.01 * scale, 3)},
           "tflops_algorithmic": round(9.01 * scale / (ms / 1e3), 1)}
    if want_cpu:
        from oracle import xception_graph as XG

        torch.set_num_threads(CPU_THREADS)
        t0 = time.perf_counter()
        XG.architecture(x_host[:1], weights, H, dtype=torch.float32)
        t1 = time.perf_counter() - t0
        n = int(max(1, min(B, round(12.0 / max(t1, 1e-3)))))
        t0 = time.perf_counter()
        XG.architecture(x_host[:n], weights, H, dtype=torch.float32)
        el = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(n * H * W / 1e6 / el, 4), "unit": "MPx/s", "cores": CPU_THREADS, "kind": "port",
                               "sample": f"1 pass over the first {n} images ([{n},{H},{W},1]), oracle/xception_graph.py "
                                         f"(PyTorch-CPU float32, {CPU_THREADS} threads), {el:.1f} s"}
    return out


def bench_G(a, torch, emdenoise, dev, timer, rank, world, want_cpu):
    """BASELINE configs[4], the part that is built: the in-filling GAN's GENERATOR forward pass
    (misc_py/gan-infilling-100.py:133-374) on 1/64-sampled 512x512 images (the discriminator and the adversarial
    training step are not built yet)."""
    from emdenoise import gan as GN

    B, S = a.batch, a.size
    steps, warmup = 5, 1
    x_host = GN.gen_lq(2.0 * synthetic_lq(B, S, S, seed=77 + rank)[..., 0] - 1.0)[..., None]
    weights = GN.synthetic_weights()
    eng = GN.GeneratorEngine(weights, dev, a.precision)
    x = torch.from_numpy(x_host).to(dev)
    box = [None]

    def step():
        box[0] = eng.forward(x)

    ms = timer.run(step, steps, warmup)
    tflop = GN.algorithmic_flops(S) * B / 1e12
    out = {"value": round(B * S * S / 1e6 * world / (ms / 1e3), 1), "unit": "MPx/s in-filled", "ms_per_step": round(ms, 3),
           "steps": steps, "warmup": warmup,
           "config": {"workload": f"G: in-filling generator forward (misc_py/gan-infilling-100.py), [{B},{S},{S},1] fp32 per GPU, "
                                  "1/64 of the pixels given", "precision": a.precision,
                      "algorithmic_tflop_per_step": round(tflop, 3)},
           "tflops_algorithmic": round(tflop / (ms / 1e3), 1)}
    if want_cpu:
        from oracle import gan_graph as GG

        torch.set_num_threads(CPU_THREADS)
        t0 = time.perf_counter()
        ref = GG.generator(x_host[:1], weights, S, dtype=torch.float32).numpy()
        el = time.perf_counter() - t0
        got = box[0][:1].cpu().numpy()
        out["cpu_baseline"] = {"value": round(S * S / 1e6 / el, 4), "unit": "MPx/s in-filled", "cores": CPU_THREADS, "kind": "port",
                               "sample": f"1 image ([1,{S},{S},1]), oracle/gan_graph.py (PyTorch-CPU float32, {CPU_THREADS} threads), {el:.1f} s"}
        out["rel_l2_vs_oracle"] = float(f"{np.linalg.norm(got - ref) / np.linalg.norm(ref):.3e}")
    return out


def bench_S(a, torch, emdenoise, dev, timer, rank, world, want_cpu):
    """the small separable autoencoder of misc_py/apply_autoencoders.py (:91-187), the reference's
    own size: 160x160 crops, encoding_features 16, a batch of `--batch` crops per GPU with per-image batch statistics."""
    from emdenoise import autoencoder as AE

    B, S = a.batch, 160
    steps, warmup = 10, 2
    x_host = synthetic_lq(B, S, S, seed=160 + rank)
    x_host = (x_host / x_host.mean(axis=(1, 2, 3), keepdims=True)).astype(np.float32)
    weights = AE.synthetic_weights(16)
    eng = AE.AutoencoderEngine(weights, dev, 16)
    x = torch.from_numpy(x_host).to(dev)
    box = [None]

    # 45 launches of a few microseconds each: captured once into a hipGraph and replayed (--no-graph: eager launches).  The big
    # graphs (D, X, G) gain nothing from a replay -- their launch queue never drains -- and stay eager.
    fwd = eng.forward if a.no_graph else GraphedForward(eng)

    def step():
        box[0] = fwd(x)

    ms = timer.run(step, steps, warmup)
    out = {"value": round(B * S * S / 1e6 * world / (ms / 1e3), 1), "unit": "MPx/s", "ms_per_step": round(ms, 3),
           "steps": steps, "warmup": warmup,
           "config": {"workload": f"S: separable autoencoder (misc_py/apply_autoencoders.py), [{B},{S},{S},1] fp32 per GPU, "
                                  "encoding_features 16, per-image batch-statistics norms", "precision": "bf16x3", "hip_graph": not a.no_graph}}
    if want_cpu:
        from oracle import autoencoder_graph as AG

        torch.set_num_threads(CPU_THREADS)
        t0 = time.perf_counter()
        ref = AG.architecture(x_host, weights, 16, dtype=torch.float32).numpy()
        el = time.perf_counter() - t0
        got = box[0].cpu().numpy()
        out["cpu_baseline"] = {"value": round(B * S * S / 1e6 / el, 4), "unit": "MPx/s", "cores": CPU_THREADS, "kind": "port",
                               "sample": f"the same [{B},{S},{S},1] batch, oracle/autoencoder_graph.py (PyTorch-CPU float32, {CPU_THREADS} threads), {el:.1f} s"}
        out["rel_l2_vs_oracle"] = float(f"{np.linalg.norm(got - ref) / np.linalg.norm(ref):.3e}")
    return out


def bench_A(a, torch, emdenoise, dev, timer, rank, world, want_cpu):
    """BASELINE configs[4]: one iteration of the in-filling GAN's training loop (misc_py/gan-infilling-100.py:1650-1790)
    on `--gan-batch` 512x512 images per GPU: generator towers through the discriminator (feature matching), generator
    Adam step, then the discriminator trained on the generated and the natural images (2T towers) and its Adam step."""
    from emdenoise import gan as GN, gan_trainer as GT

    T, S = a.gan_batch, a.size
    steps = a.steps if (a.steps is not None and a.workload == "A") else 3
    warmup = a.warmup if (a.warmup is not None and a.workload == "A") else 1
    hq = (2.0 * synthetic_lq(T, S, S, seed=177 + rank) - 1.0).astype(np.float32)
    lq = GN.gen_lq(hq[..., 0])[..., None]
    D = GT.DiscriminatorTrainer(GN.discriminator_synthetic_weights(), dev, a.precision)
    G = GT.GeneratorTrainer(GN.synthetic_weights(), D, dev, a.precision)
    rng = np.random.default_rng(5 + rank)
    pad = (3 * S) // 4
    offsets = [tuple((int(rng.integers(0, S + 2 * pad - n + 1)), int(rng.integers(0, S + 2 * pad - n + 1))) for n in (S // 4, S // 2, pad))
               for _ in range(T)]
    x, t = torch.from_numpy(lq).to(dev), torch.from_numpy(hq).to(dev)
    box = [None]

    loop = GT.GanLoop(G, D, streams=a.train_streams) if (world == 1 and not a.no_graph) else None

    def step():
        # single GPU: the iteration is replayed from a hipGraph (towers on --train-streams streams; crop offsets and Adam
        # rates live on the device); multi-GPU: eager, because the gradient all-reduces sit between the phases
        box[0] = loop.iteration(x, t, offsets) if loop is not None else GT.gan_iteration(G, D, x, t, offsets, streams=a.train_streams)

    ms = timer.run(step, steps, warmup)
    rg, rd = box[0]
    out = {"value": round(T * S * S / 1e6 * world / (ms / 1e3), 2), "unit": "MPx/s trained (GAN)", "ms_per_step": round(ms, 3),
           "steps": steps, "warmup": warmup, "dtype": "bf16x3 GEMMs (split-bf16 MFMA inputs, fp32 accumulate), fp32 elsewhere",
           "config": {"workload": f"A: in-filling GAN training iteration (misc_py/gan-infilling-100.py), [{T},{S},{S},1] fp32 per GPU: "
                                  f"{T} generator towers + Adam, {2 * T} discriminator towers + Adam",
                      "global_batch": T * world, "precision": a.precision, "parallelism": f"dp{world}",
                      "streams": a.train_streams, "hip_graph": loop is not None},
           "d_fake_first": float(rg[0, 0].item()), "d_out_first": float(rd[0, 0].item())}
    out["roofline"] = {"bound": "hbm", "kernel": "whole iteration (<= 128-channel separable convs at 256-512 px dominate)",
                       "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None, "traffic": None}
    if want_cpu:
        from oracle import gan_graph as GG

        torch.set_num_threads(CPU_THREADS)
        t0 = time.perf_counter()
        GG.generator_tower(lq[:1], hq[:1], GN.synthetic_weights(), GN.discriminator_synthetic_weights(), offsets[0])
        el = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(S * S / 1e6 / el, 4), "unit": "MPx/s trained (GAN)", "cores": CPU_THREADS, "kind": "port",
                               "sample": f"ONE generator tower ([1,{S},{S},1]: generator + 2 discriminator passes, forward + backward), "
                                         f"oracle/gan_graph.py generator_tower (PyTorch-CPU autograd, float64, {CPU_THREADS} threads), "
                                         f"{el:.1f} s; discriminator towers and optimizer steps not included"}
    return out


def bench_T(a, torch, emdenoise, dev, timer, rank, world, want_cpu, dist):
    """BASELINE configs[3]: graph D' TRAINING (misc_py/denoiser-multi-gpu.py): data-parallel steps of `--train-batch`
    512x512 LQ/HQ pairs per GPU (bs=64 over 8 GPUs => 8 per GPU), towers of `--tower-batch` images (1 = the
    reference, :763), gradients averaged over all towers and ranks (one RCCL all-reduce of the flat gradient vector),
    Nesterov momentum step, weights re-packed on the device."""
    from emdenoise import denoiser as D, trainer as TR

    B, S, tb = a.train_batch, a.size, a.tower_batch
    steps = a.steps if (a.steps is not None and a.workload == "T") else 3
    warmup = a.warmup if (a.warmup is not None and a.workload == "T") else 1
    rng = np.random.default_rng(4321 + rank)
    hq = synthetic_lq(B, S, S, seed=99 + rank)          # smooth synthetic micrographs as the clean images
    lq = np.clip(hq + rng.normal(0.0, 0.1, hq.shape).astype(np.float32), 0.0, 1.0)
    weights = D.synthetic_weights(variant="Dprime")
    tr = TR.DenoiserTrainer(weights, dev, a.precision)
    x, t = torch.from_numpy(lq).to(dev), torch.from_numpy(hq).to(dev)
    box = [None]

    def step():
        box[0] = tr.train_step(x, t, tower_batch=tb, streams=a.train_streams, graph=not a.no_graph)

    ms = timer.run(step, steps, warmup)
    tflop = 3 * 5.38 / 32.0 * B * (S * S) / (512.0 * 512.0)   # forward + data gradient + weight gradient
    out = {"value": round(B * S * S / 1e6 * world / (ms / 1e3), 2), "unit": "MPx/s trained", "ms_per_step": round(ms, 3),
           "steps": steps, "warmup": warmup, "dtype": "bf16x3 GEMMs (split-bf16 MFMA inputs, fp32 accumulate), fp32 elsewhere",
           "config": {"workload": f"T: graph D' training step (misc_py/denoiser-multi-gpu.py), [{B},{S},{S},1] fp32 LQ/HQ pairs per GPU, "
                                  f"towers of {tb}, Nesterov momentum 0.9, lr 1e-3",
                      "global_batch": B * world, "tower_batch": tb, "streams": a.train_streams, "hip_graph": not a.no_graph,
                      "precision": a.precision, "parallelism": f"dp{world}",
                      "algorithmic_tflop_per_step_per_gpu": round(tflop, 3)},
           "tflops_algorithmic": round(tflop / (ms / 1e3), 1),
           "loss_first_tower": float(box[0][0, 1].item())}
    if world > 1:  # the step's only exchange, timed on its own (SURVEY.md 8d cfg 4): all-reduce of the flat gradient
        from emdenoise.trainer import sync_gradients

        for _ in range(2):
            sync_gradients(tr.grads, tr.moving)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            sync_gradients(tr.grads, tr.moving)
        e1.record()
        torch.cuda.synchronize()
        ar_ms = e0.elapsed_time(e1) / 5
        nbytes = tr.grads.numel() * 4
        out["allreduce"] = {"ms": round(ar_ms, 3), "bytes": nbytes,
                            "bus_GBps": round(2.0 * (world - 1) / world * nbytes / (ar_ms / 1e3) / 1e9, 1),
                            "note": "RCCL all-reduce (sum) of the fp32 gradient vector + broadcast of the moving statistics"}
    out["roofline"] = {"bound": "mfma", "kernel": "whole step (gemm_conv + conv_wgrad dominate)", "achieved": out["tflops_algorithmic"],
                       "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": round(out["tflops_algorithmic"] / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None}
    if want_cpu:
        from oracle import denoiser_graph as G

        torch.set_num_threads(CPU_THREADS)
        t0 = time.perf_counter()
        G.tower_gradients(lq[:1], hq[:1], weights, S, dtype=torch.float64)
        el = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(S * S / 1e6 / el, 4), "unit": "MPx/s trained", "cores": CPU_THREADS, "kind": "port",
                               "sample": f"forward + backward of ONE tower of 1 image ([1,{S},{S},1]), oracle/denoiser_graph.py "
                                         f"tower_gradients (PyTorch-CPU autograd, float64, {CPU_THREADS} threads), {el:.1f} s; "
                                         "optimizer step not included"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["K", "D", "X", "T", "G", "A", "S", "both", "all"], default="all",
                    help="K and D: see the module docstring; X: misc_py/modified_Xception.py; all (default) = K primary, D and X alongside")
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--train-batch", type=int, default=8, help="workload T: LQ/HQ pairs per GPU per step (bs=64 over 8 GPUs)")
    ap.add_argument("--tower-batch", type=int, default=1, help="workload T: images per tower (batch-norm statistics are per tower)")
    ap.add_argument("--gan-batch", type=int, default=4, help="workload A: images per GPU per GAN iteration")
    ap.add_argument("--train-streams", type=int, default=8, help="workload T: HIP streams the towers are issued on")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying captured hipGraphs (training steps T / A, forward pass of S)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["bf16x3", "bf16"], default="bf16x3",
                    help="workload D matrix-core mode: bf16x3 = split-bf16 parity mode (default), bf16 = fast mode")
    a = ap.parse_args()

    import torch

    import emdenoise

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    timer = Timer(torch, dist, dev)
    want_cpu = rank == 0 and world == 1 and not a.no_cpu_baseline

    primary_is_D = a.workload == "D"
    res_K = res_D = None
    res_X = None
    # multi-GPU runs of the default line keep the riders to D and the training step T (the one workload with a
    # collective): every extra rider is one more place where a rank-local failure would leave the others in a barrier
    multi = world > 1
    if a.workload in ("K", "both", "all"):
        res_K = bench_K(a, torch, emdenoise, dev, timer, rank, world, want_cpu)
    if a.workload == "X" or (a.workload == "all" and not multi):
        try:
            res_X = bench_X(a, torch, emdenoise, dev, timer, rank, world, want_cpu)
        except Exception as e:
            res_X = {"error": f"{type(e).__name__}: {e}"}
    if a.workload in ("D", "both", "all"):
        try:
            res_D = bench_D(a, torch, emdenoise, dev, timer, rank, world, want_cpu)
        except Exception as e:  # the primary (K) line must survive a failure of the rider
            if primary_is_D:
                raise
            res_D = {"error": f"{type(e).__name__}: {e}"}

    res_T = None
    if a.workload in ("T", "all"):
        try:
            res_T = bench_T(a, torch, emdenoise, dev, timer, rank, world, want_cpu, dist)
        except Exception as e:
            if a.workload == "T":
                raise
            res_T = {"error": f"{type(e).__name__}: {e}"}

    res_G = None
    if a.workload == "G" or (a.workload == "all" and not multi):
        try:
            res_G = bench_G(a, torch, emdenoise, dev, timer, rank, world, want_cpu)
        except Exception as e:
            if a.workload == "G":
                raise
            res_G = {"error": f"{type(e).__name__}: {e}"}

    res_S = None
    if a.workload == "S" or (a.workload == "all" and not multi):
        try:
            res_S = bench_S(a, torch, emdenoise, dev, timer, rank, world, want_cpu)
        except Exception as e:
            if a.workload == "S":
                raise
            res_S = {"error": f"{type(e).__name__}: {e}"}

    res_A = None
    if a.workload == "A" or (a.workload == "all" and not multi):
        try:
            res_A = bench_A(a, torch, emdenoise, dev, timer, rank, world, want_cpu and a.workload == "A")
        except Exception as e:
            if a.workload == "A":
                raise
            res_A = {"error": f"{type(e).__name__}: {e}"}

    prim = res_D if primary_is_D else res_K
    if prim is None and a.workload == "A":
        prim, res_A = res_A, None
    if prim is None and a.workload == "G":
        prim, res_G = dict(res_G), None
        prim["dtype"] = "bf16x3 (split-bf16 MFMA inputs, fp32 accumulate and activations)"
        prim["roofline"] = {"bound": "mfma", "kernel": "gemm_conv_kernel", "achieved": prim["tflops_algorithmic"],
                            "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(prim["tflops_algorithmic"] / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None}
    if prim is None and a.workload == "S":
        prim, res_S = dict(res_S), None
        prim["dtype"] = "bf16x3 (split-bf16 MFMA inputs, fp32 accumulate and activations)"
        prim["roofline"] = {"bound": "hbm", "kernel": "whole forward (launch-bound at 160 px: per-image statistics launches)",
                            "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None, "traffic": None}
    if prim is None and a.workload == "T":
        prim, res_T = res_T, None
    if prim is None:  # --workload X alone
        prim = dict(res_X)
        prim.setdefault("dtype", "bf16x3 (split-bf16 MFMA inputs, fp32 accumulate and activations)")
        prim.setdefault("roofline", {"bound": "mfma", "kernel": "gemm_conv_kernel", "achieved": prim.get("tflops_algorithmic"),
                                     "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": round((prim.get("tflops_algorithmic") or 0.0) / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None})
        res_X = None
    out = {
        "metric": {"T": "megapixels/sec trained (512x512x1 LQ/HQ pairs)", "G": "megapixels/sec in-filled (512x512x1 bs=32)", "S": "megapixels/sec restored (160x160x1 crops, bs=32)",
                   "A": "megapixels/sec trained (in-filling GAN, 512x512x1)"}.get(
            a.workload, "megapixels/sec restored (512x512x1 bs=32)"),
        "value": round(prim["value"], 1),
        "unit": prim.get("unit", "MPx/s"),
        "n_gpus": world,
        "steps": prim["steps"],
        "warmup": prim["warmup"],
        "ms_per_step": round(prim["ms_per_step"], 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": prim["dtype"],
        "data": "synthetic",
        "config": prim["config"],
        "roofline": prim["roofline"],
    }
    for k in ("cpu_baseline", "rel_l2_vs_oracle", "depthwise", "pointwise", "kernel_family_ms", "tflops_algorithmic", "loss_first_tower", "d_fake_first", "d_out_first"):
        if k in prim:
            out[k] = prim[k]
    if not primary_is_D and res_D is not None:
        if "value" in res_D:
            res_D["value"] = round(res_D["value"], 1)
            res_D["ms_per_step"] = round(res_D["ms_per_step"], 4)
            res_D["unit"] = "MPx/s"
        out["workload_D"] = res_D
    if res_X is not None:
        out["workload_X"] = res_X
    if res_T is not None:
        out["workload_T"] = res_T
    if res_G is not None:
        out["workload_G"] = res_G
    if res_A is not None:
        out["workload_A"] = res_A
    if res_S is not None:
        out["workload_S"] = res_S
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
