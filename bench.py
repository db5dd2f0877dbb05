#!/usr/bin/env python3
"""bench.py -- megapixels/s restored on synthetic 512x512x1 micrograph batches (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload K|D] [--batch B]

A "step" is one pass of the hot path over one batch that is already resident in HBM.
Workloads:
  K  BASELINE configs[1]: the 3-layer 3x3 kernel denoiser (misc_py/noise-removal-kernels.py,
     depth 2, width 3) on [32,512,512,1]                                   -- default
  D  BASELINE configs[2]: the modified-Xception encoder-decoder of machine_learning/denoiser.py
     on [32,512,512,1]
For N > 1 the driver launches one rank per GPU (torch.distributed.run); inference shards whole
images across ranks with no data-path collective (weak scaling: B images PER GPU).
Rank 0 prints ONE JSON line.
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

HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA


def synthetic_lq(B, H, W, seed=1234):
    """Synthetic low-quality crops of the reference's shape and statistics:
    smooth field -> Poisson counts (scale = 25 + Exp(75), denoiser-multi-gpu.py:783-799) ->
    min-max to [0,1]; the K path additionally divides by the mean (noise-removal-kernels.py:525-527)."""
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
        lq = (lq - lq.min()) / max(float(lq.max() - lq.min()), 1e-9)
        base.append(lq)
    out = np.stack([base[i % len(base)] for i in range(B)])[..., None].astype(np.float32)
    # decorrelate the replicas with a little extra shot noise so no two images are identical
    out += rng.random(out.shape, dtype=np.float32) * np.float32(1e-3)
    return out


# ------------------------------------------------------------------------------------------------
# CPU baseline (the oracle, timed; reported next to the GPU number, never the thing shipped)
# ------------------------------------------------------------------------------------------------
def cpu_baseline_K(x_host, W, Bm, s, budget_s=12.0):
    """Times oracle/k_oracle.c (plain-C port of graph K, OpenMP over rows) on the host cores."""
    import subprocess

    so = os.path.join(ROOT, "oracle", "_build", "libk_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    lib.k_oracle_f32.restype = ctypes.c_int
    lib.k_oracle_f32.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 5 + [ctypes.c_void_p] * 3 + [ctypes.c_int]
    lib.k_oracle_max_threads.restype = ctypes.c_int
    cores = min(lib.k_oracle_max_threads(), os.cpu_count() or 1)
    x = np.ascontiguousarray(x_host[..., 0])
    y = np.empty_like(x)
    B, H, Wd = x.shape

    def run():
        rc = lib.k_oracle_f32(x.ctypes.data, y.ctypes.data, B, H, Wd, W.shape[1], W.shape[0], W.ctypes.data,
                              Bm.ctypes.data, s.ctypes.data, cores)
        assert rc == 0

    run()  # warm-up
    reps, t0 = 0, time.perf_counter()
    while True:
        run()
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= 200:
            break
    mpx = B * H * Wd / 1e6 * reps / el
    return {"value": round(mpx, 2), "unit": "MPx/s", "cores": cores, "kind": "port",
            "sample": f"{reps} passes over the same [{B},{H},{Wd},1] batch, oracle/k_oracle.c (gcc -O3 -fopenmp), {el:.1f} s"}, y


# ------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["K", "D"], default="K")
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch

    import emdenoise

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
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

    B, H, W = a.batch, a.size, a.size
    x_host = synthetic_lq(B, H, W, seed=1234 + rank)

    if a.workload == "K":
        steps = a.steps if a.steps is not None else 200
        warmup = a.warmup if a.warmup is not None else 20
        x_host = x_host / np.maximum(x_host.mean(axis=(1, 2, 3), keepdims=True), 1e-9)
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

        workload = f"K: kernel denoiser depth 2 width 3 (noise-removal-kernels.py), [{B},{H},{W},1] fp32 per GPU"
        dtype = "f32"
        alg_bytes_per_launch = 8.0 * B * H * W  # 8 B/pixel (read 4 + write 4)
        dominant = "k3_rows<MODE_SYM>"
        launches_per_step = 1
    else:
        raise SystemExit("workload D is not wired into bench.py yet")

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    sync_all()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for s_ev, e_ev in evs:
        s_ev.record()
        step()
        e_ev.record()
    sync_all()
    t1 = time.perf_counter()
    wall = t1 - t0
    if dist is not None:
        tt = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())
    ms_per_step = wall * 1e3 / steps
    # HIP events on the launch stream (torch's current stream IS the stream the C ABI launches on)
    ev_ms = np.array([s_ev.elapsed_time(e_ev) for s_ev, e_ev in evs])
    kern_ms = float(np.mean(ev_ms)) / launches_per_step
    total_ms_events = evs[0][0].elapsed_time(evs[-1][1])

    mpx_per_step = B * H * W / 1e6 * world
    value = mpx_per_step / (ms_per_step / 1e3)
    achieved = alg_bytes_per_launch / (kern_ms * 1e-3) / 1e9

    out = {
        "metric": "megapixels/sec restored (512x512x1 bs=32)",
        "value": round(value, 1),
        "unit": "MPx/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": dtype,
        "data": "synthetic",
        "config": {"workload": workload, "global_batch": B * world, "image": f"{H}x{W}x1",
                   "sharding": f"{world} x {B} whole images, no collective"},
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                     "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                     "avg_launch_us_hip_events": round(kern_ms * 1e3, 3),
                     "avg_step_us_back_to_back": round(total_ms_events * 1e3 / steps, 3)},
    }
    if rank == 0 and not a.no_cpu_baseline:
        W_, Bm_, s_ = params.wmaps, params.bmaps, params.s
        cb, y_cpu = cpu_baseline_K(x_host, W_, Bm_, s_)
        out["cpu_baseline"] = cb
        y_gpu = y.cpu().numpy()[..., 0]
        rel = float(np.linalg.norm(y_gpu.astype(np.float64) - y_cpu) / np.linalg.norm(y_cpu))
        out["rel_l2_vs_oracle"] = float(f"{rel:.3e}")
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
