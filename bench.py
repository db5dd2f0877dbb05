#!/usr/bin/env python3
"""bench.py -- megapixels/s restored on synthetic 512x512x1 micrograph batches (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload D|K|X|T|G|A|S|all] [--batch B]
                    [--scaling weak|strong] [--dry-run]

A "step" is one pass of the hot path over one batch that is already resident in HBM.  Workloads:
  D  BASELINE configs[2], THE PRIMARY LINE: the modified-Xception encoder-decoder of machine_learning/denoiser.py on
     [32,512,512,1] -- the conv stack north_star sets its depthwise / pointwise targets on (matrix cores in split-bf16
     parity mode unless --precision bf16).
  K  BASELINE configs[1]: the 3-layer 3x3 kernel denoiser (misc_py/noise-removal-kernels.py, depth 2, width 3).
  X  the other graph BASELINE configs[2] can mean: misc_py/modified_Xception.py at 512x512.
  T  BASELINE configs[3]: graph D' TRAINING (misc_py/denoiser-multi-gpu.py): --train-batch LQ/HQ pairs per GPU per step,
     one RCCL all-reduce of the flat gradient vector per step, Nesterov step.
  G  BASELINE configs[4], forward part: the in-filling GAN's generator (misc_py/gan-infilling-100.py:133-374).
  A  BASELINE configs[4]: one adversarial training iteration of the in-filling GAN.
  S  SURVEY.md 8f rank 4: the small separable autoencoder (misc_py/apply_autoencoders.py) on 160-px crops.
Default ("all"): the JSON line's metric/value/roofline/cpu_baseline are workload D's, measured with --steps/--warmup
(default 20 / 5); with one rank K, X, T, G, S, A ride along under "workload_<letter>" with short fixed step counts, and a rider that
throws is reported in the line AND makes the exit code non-zero.  With more than one rank only the primary workload runs (the
training step's scaling: `--workload T --gpus N`); any exception there re-raises at once, so that the launcher tears every rank
down instead of leaving the others in a collective.

Multi-GPU: `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the ranks itself -- a parent
that has touched neither torch nor the GPU runs `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child
process and relays rank 0's JSON line and the exit code.  Under an external launcher (the driver's torch.distributed.run)
the ranks are used as given.  Inference shards whole images across ranks with no data-path collective:
  --scaling weak   (default) --batch images PER GPU;
  --scaling strong --batch images in total, cut contiguously (input_pipeline.shard_contiguous).
--dry-run runs the whole launch / rendezvous / sharding / timing / reporting path over gloo on the CPU with a host stand-in
for the step (no GPU, no kernels): what tests/test_bench_launch.py exercises with world_size 2.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CPU_THREADS = min(16, os.cpu_count() or 1)  # a 1-GPU box owns a 16-CPU share; more threads only oversubscribe it
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA
D_GMAC_MATRIX_B32_512 = 2218.3  # pointwise 1304.5 + dense 1x1 295.3 - final 4.8 + conv-T 618.5 ... per B=32 batch
PMC_FILES = tuple(f"r{n:02d}_pmc_traffic.json" for n in range(9, 0, -1))   # newest first (a round that has not refreshed it yet falls back to the last one, stale if csrc changed)


# ================================================================================================
# launcher (no torch, no GPU): python bench.py --gpus N  ->  N ranks under torch.distributed.run
# ================================================================================================
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["K", "D", "X", "T", "G", "A", "S", "both", "all"], default="all",
                    help="all (default) = D primary with K, X, T, G, S, A alongside; a letter = that workload alone as the primary line")
    ap.add_argument("--batch", type=int, default=32, help="images per GPU (--scaling weak) or in total (--scaling strong)")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --batch images per GPU; strong: --batch images in total, cut contiguously across the ranks")
    ap.add_argument("--dry-run", action="store_true", help="gloo on the CPU, a host stand-in for the step: launch / sharding / timing path only")
    ap.add_argument("--train-batch", type=int, default=8, help="workload T: LQ/HQ pairs per GPU per step (bs=64 over 8 GPUs)")
    ap.add_argument("--tower-batch", type=int, default=1, help="workload T: images per tower (batch-norm statistics are per tower)")
    ap.add_argument("--gan-batch", type=int, default=4, help="workload A: images per GPU per GAN iteration")
    ap.add_argument("--train-streams", type=int, default=8, help="workload T: HIP streams the towers are issued on")
    ap.add_argument("--tower-mode", choices=["streams", "batched"], default="batched",
                    help="workload T, towers of 1: the towers on --train-streams HIP streams, or as ONE batched pass with per-image batch-norm statistics")
    ap.add_argument("--mask", choices=["spiral", "bernoulli"], default="spiral",
                    help="workloads G / A: which 1/64 of the pixels is given -- a spiral scan path (ours) or the reference's fixed Bernoulli field")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying captured hipGraphs (T / A / S)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-child", default=None, metavar="DIR",
                    help="internal: run the CPU-baseline legs of --cpu-legs in THIS process (no GPU: HIP_VISIBLE_DEVICES is empty) and "
                         "write <leg>.json / <leg>_ref.npy into DIR")
    ap.add_argument("--cpu-legs", default="", help="internal: comma-separated workloads for --cpu-child")
    ap.add_argument("--no-riders", action="store_true", help="with --workload all: the primary workload only")
    ap.add_argument("--profile-clean", action="store_true",
                    help="for rocprofv3: the primary workload's timed steps and NOTHING else in the process (no per-family attribution pass, no "
                         "isolated-GEMM / peak micro-benchmarks, no native executor, no pinned-host pass, no CPU legs): per-kernel averages of "
                         "the trace then mean one thing.  SINGLE stream by default (full-batch launches back to back: avg us x launches of a "
                         "kernel family reproduces depthwise_frac / pointwise_frac by hand); --profile-streams two = the timed step's own form")
    ap.add_argument("--profile-streams", choices=["single", "two"], default="single",
                    help="with --profile-clean: single = DenoiserEngine.two_streams / .pipeline off (every launch is a full batch and runs alone); "
                         "two = the two-half-batch form the headline step runs (a kernel's average then means 'beside the other half's kernels')")
    ap.add_argument("--precision", choices=["bf16x3", "bf16"], default="bf16x3",
                    help="matrix-core mode: bf16x3 = split-bf16 parity mode (default), bf16 = fast mode (never reported as parity)")
    return ap.parse_args(argv)


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(a, argv):
    """Parent of a self-launched multi-rank run.  It has imported neither torch nor the package and never touches the GPU
    (a process that has initialised HIP must not exec or re-launch itself on this pool); it starts the ranks as a CHILD
    process tree, relays the child's stdout (rank 0's JSON line) and returns its exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, CPU_THREADS // max(a.gpus, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(proc.stdout)
    sys.stdout.flush()
    return proc.returncode


# ================================================================================================
# helpers shared by the workloads
# ================================================================================================
def csrc_sha16():
    """Hash of the kernel sources (csrc/*.hip, *.hpp, *.cpp): what a committed PMC traffic file was measured on.  (The GPU box has no
    .git, and committing the file itself moves HEAD: the kernels' text is the thing that must not have changed.)"""
    import hashlib

    d = os.path.join(ROOT, "ai-cv-automation-elect-micr_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC run (tools/collect_traffic.sh: FETCH_SIZE and WRITE_SIZE in
    separate passes, gfx950 correction applied).  bench.py cannot profile itself, so `traffic` is the figure of that run for
    the same kernels and shapes -- and ONLY while the kernel sources are the ones it was measured on (csrc_sha16 in the file):
    (None, why) otherwise."""
    for name in PMC_FILES:
        path = os.path.join(ROOT, "profiles", name)
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("csrc_sha16") != csrc_sha16():
            return None, f"profiles/{name} was measured on other kernel sources (csrc_sha16 {d.get('csrc_sha16')} != {csrc_sha16()}): stale, not reported"
        return d["kernels"], "profiles/" + name
    return None, None


def whole_step_traffic_roofline(w, ms, kernel):
    """Graphs S and A have no closed-form algorithmic byte count here (S: 45 launches with per-image statistics; A: ~3000 launches of a
    whole GAN iteration): their roofline is the MEASURED HBM traffic of one step (committed PMC run, "<w>:*step total*") over the step
    time -- how busy the memory system is, an upper bound of the algorithmic fraction."""
    tr, src = pmc_traffic()
    rec = (tr or {}).get(f"{w}:*step total*")
    if not rec:
        return {"bound": "hbm", "kernel": kernel, "achieved": None, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": None, "traffic": None, "traffic_source": src}
    by = rec["hbm_bytes_per_step_corrected"]
    ach = by / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
            "traffic": round(by), "traffic_source": src,
            "how": "achieved = PMC HBM bytes of one step (FETCH_SIZE, WRITE_SIZE in separate rocprofv3 passes, gfx950 correction) / step time: "
                   "traffic-based, not algorithmic"}


def step_traffic(w):
    """HBM bytes of ONE step of workload w from the committed PMC run ("<w>:*step total*"), or (None, why)."""
    tr, src = pmc_traffic()
    rec = (tr or {}).get(f"{w}:*step total*")
    if not rec:
        return None, src
    return round(rec["hbm_bytes_per_step_corrected"]), src


def synthetic_lq(B, H, W, seed=1234):
    """Synthetic low-quality crops of the reference's shape and statistics: smooth field ->
    Poisson counts (scale = 25 + Exp(75), denoiser-multi-gpu.py:783-799) -> min-max to [0,1]."""
    import numpy as np

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


def d_graph_algorithmic_bytes(B, S):
    """fp32 activation bytes a FULLY FUSED graph D has to move per step (SURVEY.md 8d / 8a' stage table): every layer reads its
    input once and writes its output once, a residual add reads the residual, a separable conv's depthwise result never touches
    HBM, concats are free (producers write into their slice).  B=32, S=512: 80.0 GB = SURVEY's 118 GB of unfused Sigma(in + out)
    - 2 x 22.5 GB of depthwise intermediates + 6.5 GB of residual reads (which the survey's sum leaves out)."""
    e = 0.0   # elements per image
    P = lambda h: float(h) * float(h)
    s1, s2, s4, s8, s16 = S, S // 2, S // 4, S // 8, S // 16
    lay = lambda hin, cin, hout, cout, res=False: cin * P(hin) + cout * P(hout) * (2 if res else 1)
    e += lay(s1, 1, s1, 64) + lay(s1, 64, s1, 64) + lay(s1, 64, s2, 128, True) + lay(s1, 1, s2, 128)
    e += 2 * lay(s2, 128, s2, 128) + lay(s2, 128, s4, 128, True) + lay(s2, 128, s4, 128)
    e += lay(s4, 128, s4, 256) + lay(s4, 256, s4, 256) + lay(s4, 256, s8, 256, True) + lay(s4, 128, s8, 256)
    e += lay(s8, 256, s8, 728) + lay(s8, 728, s8, 728) + lay(s8, 728, s16, 728, True) + lay(s8, 256, s16, 728)
    e += 24 * lay(s16, 728, s16, 728) + 12 * lay(s16, 728, s16, 728, True)
    e += 5 * lay(s16, 728, s16, 728) + lay(s16, 3640, s16, 256) + lay(s16, 256, s4, 256)
    e += lay(s4, 384, s4, 256) + lay(s4, 256, s4, 256, True) + lay(s4, 384, s4, 256) + lay(s4, 256, s2, 256)
    e += lay(s2, 384, s2, 128) + lay(s2, 128, s2, 128, True) + lay(s2, 384, s2, 128) + lay(s2, 128, s1, 128)
    e += lay(s1, 128, s1, 64) + lay(s1, 64, s1, 64, True) + lay(s1, 128, s1, 64) + lay(s1, 64, s1, 1)
    return 4.0 * e * B


def psnr_rel(y_gpu, y_ref):
    import numpy as np

    d = y_gpu.astype(np.float64) - y_ref.astype(np.float64)
    mse = float(np.mean(d * d))
    rel = float(np.linalg.norm(d) / max(np.linalg.norm(y_ref.astype(np.float64)), 1e-30))
    return (float("inf") if mse == 0 else round(10.0 * math.log10(1.0 / mse), 2)), float(f"{rel:.3e}")


class Timer:
    """Barrier + synchronize on both sides, exactly `steps` steps, max over ranks (the bench contract); one HIP-event pair per
    step on the launch stream gives the median device-side step time beside it."""

    def __init__(self, torch, dist, dev):
        self.torch, self.dist, self.dev = torch, dist, dev
        self.gpu = dev is not None and dev.type == "cuda"

    def sync(self):
        if self.gpu:
            self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            if self.gpu:
                self.torch.cuda.synchronize()

    def run(self, step, steps, warmup):
        for _ in range(warmup):
            step()
        self.sync()
        # per-step events only where a step is long against an event pair (~3 us each): graph K's 15 us steps are timed bare
        per_step = False
        if self.gpu:
            t0 = time.perf_counter()
            step()
            self.torch.cuda.synchronize()
            per_step = (time.perf_counter() - t0) > 5e-4
            if self.dist is not None:
                self.dist.barrier()
        evs = []
        t0 = time.perf_counter()
        for _ in range(steps):
            if per_step:
                e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
                e0.record()
                step()
                e1.record()
                evs.append((e0, e1))
            else:
                step()
        self.sync()
        wall = time.perf_counter() - t0
        if self.dist is not None:
            tt = self.torch.tensor([wall], dtype=self.torch.float64, device=self.dev if self.gpu else "cpu")
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
            wall = float(tt.item())
        self.median_event_ms = None
        if evs:
            ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
            self.median_event_ms = round(ts[len(ts) // 2], 4)
        return wall * 1e3 / steps


class FamilyTimer:
    """Device time per kernel family of ONE extra step: HIP events around every launch that goes through a function of
    emdenoise.ops, with the algorithmic flops / bytes of the launch taken from its arguments."""

    MATRIX = ("conv1x1", "conv3x3", "deconv3x3s2", "sep_fused", "sep_fused_gen", "sep_dual", "conv1x1_split32", "conv3x3_split32",
              "deconv3x3s2_split32", "deconv3x3s2_fused")
    NAMES = MATRIX + ("dw3x3", "dw3x3_split32", "dw3x3_reflect", "dw3x3_reflect_split32", "dw3x3_reflect_gen", "cin1", "cin1_k7_reflect",
                      "conv3x3_cout1", "conv3x3_cout1_reflect", "resize_bilinear", "affine_relu6", "affine_act", "affine_act_images",
                      "bn_batch_stats", "bn_batch_stats_images", "avgpool2x2", "to_split32", "instnorm_tanh")

    def __init__(self, torch, ops):
        self.torch, self.ops = torch, ops
        self.ev, self.flops, self.bytes_, self.orig = {}, {}, {}, {}
        self.pw_shapes = []

    def _account(self, name, args, kw):
        ops = self.ops
        try:
            if name in ("conv1x1", "conv1x1_split32"):
                x, w, out = args[0], args[1], (args[4] if len(args) > 4 else kw["out"])
                self.flops[name] = self.flops.get(name, 0.0) + 2.0 * out.B * out.H * out.W * w.cin * w.cout
                if name == "conv1x1_split32":   # the 728-channel flow on its own (north_star's pointwise path): remember which launches
                    self.pw_shapes.append((w.cin, w.cout, out.B * out.H * out.W, 2.0 * out.B * out.H * out.W * w.cin * w.cout))
            elif name in ("conv3x3", "conv3x3_split32"):
                w, out = args[1], (args[4] if len(args) > 4 else kw["out"])
                self.flops[name] = self.flops.get(name, 0.0) + 2.0 * out.B * out.H * out.W * 9 * w.cin * w.cout
            elif name in ("deconv3x3s2", "deconv3x3s2_split32", "deconv3x3s2_fused"):
                x, wp = args[0], args[1]
                self.flops[name] = self.flops.get(name, 0.0) + 2.0 * x.B * x.H * x.W * 9 * wp[0].cin * wp[0].cout
            elif name == "sep_fused":   # algorithmic bytes of a fused separable launch: in + out (+ residual), fp32
                x, w = args[0], args[2]
                self.flops[name] = self.flops.get(name, 0.0) + 2.0 * x.B * x.H * x.W * w.cin * w.cout
                self.bytes_[name] = self.bytes_.get(name, 0.0) + 4.0 * x.B * x.H * x.W * (w.cin + w.cout * (2 if kw.get("res") is not None else 1))
            elif name == "sep_fused_gen":   # the generated input is one value per pixel
                d, w = args[0], args[4]
                self.flops[name] = self.flops.get(name, 0.0) + 2.0 * d.B * d.H * d.W * w.cin * w.cout
                self.bytes_[name] = self.bytes_.get(name, 0.0) + 4.0 * d.B * d.H * d.W * (1 + w.cout * (2 if kw.get("res") is not None else 1))
            elif name == "sep_dual":
                x, w, w2 = args[0], args[2], args[3]
                self.flops[name] = self.flops.get(name, 0.0) + 2.0 * x.B * x.H * x.W * w.cin * (w.cout + w2.cout)
                self.bytes_[name] = self.bytes_.get(name, 0.0) + 4.0 * x.B * x.H * x.W * (w.cin + w.cout + w2.cout)
            elif name in ("dw3x3", "dw3x3_split32"):   # algorithmic bytes: fp32 in + out (a split32 output has the bytes of its fp32 twin)
                xin, out = args[0], args[2]
                self.bytes_[name] = self.bytes_.get(name, 0.0) + 4.0 * xin.C * (xin.B * xin.H * xin.W + out.B * out.H * out.W)
        except Exception:   # accounting never breaks the measured step
            pass

    def __enter__(self):
        torch = self.torch
        for name in self.NAMES:
            f = getattr(self.ops, name, None)
            if f is None:
                continue
            self.orig[name] = f

            def g(*args, _f=f, _n=name, **kw):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = _f(*args, **kw)
                e1.record()
                self.ev.setdefault(_n, []).append((e0, e1))
                self._account(_n, args, kw)
                return r

            setattr(self.ops, name, g)
        return self

    def __exit__(self, *exc):
        for name, f in self.orig.items():
            setattr(self.ops, name, f)
        self.torch.cuda.synchronize()
        self.ms = {k: sum(e0.elapsed_time(e1) for e0, e1 in v) for k, v in self.ev.items()}
        self.launches = {k: len(v) for k, v in self.ev.items()}
        return False

    def matrix_ms(self):
        return sum(self.ms.get(k, 0.0) for k in self.MATRIX)

    def matrix_flops(self):
        return sum(self.flops.get(k, 0.0) for k in self.MATRIX)

    def table(self):
        return {k: round(v, 3) for k, v in sorted(self.ms.items(), key=lambda kv: -kv[1])}


def measured_peaks(torch, dev):
    """On-box peaks: a read + write stream copy of 1 GiB (far beyond the 256 MiB Infinity Cache) and a
    back-to-back bf16 MFMA loop on random operands, one wave per SIMD on every CU; HIP events, median of 5 bursts."""
    from emdenoise import _lib

    lib = _lib.load()
    out = {}
    try:
        n = 1 << 28   # floats: 1 GiB each way
        a = torch.rand(n, device=dev)
        b = torch.empty_like(a)
        st = _lib.stream_ptr()
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        ts = []
        for r in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                _lib.check(lib.emd_debug_stream_copy_f32(p(a), p(b), n, st))
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts.append(e0.elapsed_time(e1) / 4)
        ms = sorted(ts)[len(ts) // 2]
        out["stream_copy_GBps"] = round(2.0 * 4.0 * n / (ms * 1e-3) / 1e9, 1)
        out["stream_copy_note"] = "emd_debug_stream_copy_f32, 1 GiB read + 1 GiB write per launch, one float4 per lane and no loop (a grid-stride loop over the same bytes reaches 3.5-5.0 TB/s), HIP events"
        del a, b
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        opsb = (torch.randn(2048, device=dev)).to(torch.bfloat16).contiguous()
        sink = torch.empty(cus * 256, device=dev)
        iters = 1500
        ts = []
        for r in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(lib.emd_debug_mfma_peak_bf16(p(opsb), p(sink), cus, iters, st))
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts.append(e0.elapsed_time(e1))
        ms = sorted(ts)[len(ts) // 2]
        fl = float(cus) * 4 * iters * 32 * 32768.0
        out["mfma_bf16_TFLOPs"] = round(fl / (ms * 1e-3) / 1e12, 1)
        out["mfma_note"] = (f"emd_debug_mfma_peak_bf16: {cus} workgroups x 4 waves x {iters * 32} back-to-back v_mfma_f32_32x32x16_bf16 "
                            "on random register operands (the clock the chip holds under this load is part of the figure)")
    except Exception as e:
        out["error"] = f"{type(e).__name__}: {e}"
    return out


# ------------------------------------------------------------------------------------------------
# host-side inputs of every workload (numpy only): ONE definition, used by the GPU leg and by the CPU-baseline child
# ------------------------------------------------------------------------------------------------
def inputs_K(a, B, first):
    import numpy as np

    import emdenoise

    x_host = synthetic_lq(max(B, 1), a.size, a.size, seed=1234 + first)[:max(B, 1)]
    x_host = (x_host / np.maximum(x_host.mean(axis=(1, 2, 3), keepdims=True), 1e-9)).astype(np.float32)  # noise-removal-kernels.py:525-527
    rng = np.random.default_rng(7)
    pairs = emdenoise.kernel_denoiser.sym_pairs(3)
    wsc = [(rng.standard_normal(len(pairs)) * 0.15 + 1.0 / 9).astype(np.float32) for _ in range(2)]
    bsc = [np.zeros(len(pairs), np.float32), (rng.standard_normal(len(pairs)) * 0.5).astype(np.float32)]
    return x_host, emdenoise.KernelParams.from_symmetric(wsc, bsc, [1.0, 1.3], 3)


def inputs_G(a, B, first):
    from emdenoise import gan as GN

    return GN.gen_lq(2.0 * synthetic_lq(max(B, 1), a.size, a.size, seed=77 + first)[..., 0] - 1.0, select=g_mask(a))[..., None]


def g_mask(a):
    """--mask spiral (default; BASELINE configs[4]'s wording, a generator of our own: gan.spiral_mask) or bernoulli (the reference's
    fixed field, gan-infilling-100.py:1172-1175).  The mask is input DATA: no kernel's work depends on it."""
    from emdenoise import gan as GN

    return GN.spiral_mask(a.size) if a.mask == "spiral" else None


def inputs_S(a, B, first):
    import numpy as np

    x_host = synthetic_lq(max(B, 1), 160, 160, seed=160 + first)
    return (x_host / x_host.mean(axis=(1, 2, 3), keepdims=True)).astype(np.float32)


def inputs_A(a, rank):
    import numpy as np

    from emdenoise import gan as GN

    T, S = a.gan_batch, a.size
    hq = (2.0 * synthetic_lq(T, S, S, seed=177 + rank) - 1.0).astype(np.float32)
    lq = GN.gen_lq(hq[..., 0], select=g_mask(a))[..., None]
    rng = np.random.default_rng(5 + rank)
    pad = (3 * S) // 4
    offsets = [tuple((int(rng.integers(0, S + 2 * pad - n + 1)), int(rng.integers(0, S + 2 * pad - n + 1))) for n in (S // 4, S // 2, pad))
               for _ in range(T)]
    return hq, lq, offsets


def inputs_T(a, rank):
    import numpy as np

    B, S = a.train_batch, a.size
    rng = np.random.default_rng(4321 + rank)
    hq = synthetic_lq(B, S, S, seed=99 + rank)          # smooth synthetic micrographs as the clean images
    lq = np.clip(hq + rng.normal(0.0, 0.1, hq.shape).astype(np.float32), 0.0, 1.0)
    return hq, lq


# ------------------------------------------------------------------------------------------------
# CPU baselines: the ORACLE timed on the host cores (reported next to the GPU number; never shipped).
# They run in a CHILD process that is started before this process has touched the GPU and that never sees one
# (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES empty): oneDNN's threads do not share a process with the HIP
# runtime (two aborts on record came from exactly that combination) and do not compete with the launch thread.
# ------------------------------------------------------------------------------------------------
def cpu_leg(a, w):
    """One CPU-baseline leg in the current (GPU-free) process -> (cpu_baseline dict, reference array or None, extra dict)."""
    import numpy as np
    import torch

    import emdenoise

    torch.set_num_threads(CPU_THREADS)
    B, first, _ = local_batch(a, 0, 1)
    S = a.size
    if w == "K":
        x_host, params = inputs_K(a, B, first)
        cb, y = cpu_baseline_K(x_host[:B], params.wmaps, params.bmaps, params.s)
        return cb, y, {}
    if w == "D":
        cb, y = cpu_baseline_D(synthetic_lq(max(B, 1), S, S, seed=1234 + first)[:B], emdenoise.synthetic_weights())
        return cb, y, {}
    if w == "X":
        from emdenoise import xception as X
        from oracle import xception_graph as XG

        x_host = synthetic_lq(max(B, 1), S, S, seed=1234 + first)
        n = min(B, 2)   # X normalises with the statistics of the batch it is given: parity is defined on the SAME sub-batch
        t0 = time.perf_counter()
        ref = XG.architecture(x_host[:n], X.synthetic_weights(), S, dtype=torch.float32).numpy()
        el = time.perf_counter() - t0
        return ({"value": round(n * S * S / 1e6 / el, 4), "unit": "MPx/s", "cores": CPU_THREADS, "kind": "port",
                 "sample": f"1 pass over the first {n} images ([{n},{S},{S},1]), oracle/xception_graph.py "
                           f"(PyTorch-CPU float32, {CPU_THREADS} threads), {el:.1f} s"}, ref, {"n": n})
    if w == "G":
        from emdenoise import gan as GN
        from oracle import gan_graph as GG

        x_host = inputs_G(a, B, first)
        t0 = time.perf_counter()
        ref = GG.generator(x_host[:1], GN.synthetic_weights(), S, dtype=torch.float32).numpy()
        el = time.perf_counter() - t0
        return ({"value": round(S * S / 1e6 / el, 4), "unit": "MPx/s in-filled", "cores": CPU_THREADS, "kind": "port",
                 "sample": f"1 image ([1,{S},{S},1]), oracle/gan_graph.py (PyTorch-CPU float32, {CPU_THREADS} threads), {el:.1f} s"}, ref, {})
    if w == "S":
        from emdenoise import autoencoder as AE
        from oracle import autoencoder_graph as AG

        x_host = inputs_S(a, B, first)
        t0 = time.perf_counter()
        ref = AG.architecture(x_host[:B], AE.synthetic_weights(16), 16, dtype=torch.float32).numpy()
        el = time.perf_counter() - t0
        return ({"value": round(B * 160 * 160 / 1e6 / el, 4), "unit": "MPx/s", "cores": CPU_THREADS, "kind": "port",
                 "sample": f"the same [{B},160,160,1] batch, oracle/autoencoder_graph.py (PyTorch-CPU float32, {CPU_THREADS} threads), {el:.1f} s"},
                ref, {})
    if w == "A":
        from emdenoise import gan as GN
        from oracle import gan_graph as GG

        hq, lq, offsets = inputs_A(a, 0)
        t0 = time.perf_counter()
        GG.generator_tower(lq[:1], hq[:1], GN.synthetic_weights(), GN.discriminator_synthetic_weights(), offsets[0])
        el = time.perf_counter() - t0
        return ({"value": round(S * S / 1e6 / el, 4), "unit": "MPx/s trained (GAN)", "cores": CPU_THREADS, "kind": "port",
                 "sample": f"ONE generator tower ([1,{S},{S},1]: generator + 2 discriminator passes, forward + backward), "
                           f"oracle/gan_graph.py generator_tower (PyTorch-CPU autograd, float64, {CPU_THREADS} threads), "
                           f"{el:.1f} s; discriminator towers and optimizer steps not included"}, None, {})
    if w == "T":
        from emdenoise import denoiser as D
        from oracle import denoiser_graph as G

        hq, lq = inputs_T(a, 0)
        t0 = time.perf_counter()
        G.tower_gradients(lq[:1], hq[:1], D.synthetic_weights(variant="Dprime"), S, dtype=torch.float32)
        el = time.perf_counter() - t0
        return ({"value": round(S * S / 1e6 / el, 4), "unit": "MPx/s trained", "cores": CPU_THREADS, "kind": "port",
                 "sample": f"forward + backward of ONE tower of 1 image ([1,{S},{S},1]), oracle/denoiser_graph.py "
                           f"tower_gradients (PyTorch-CPU autograd, float32, {CPU_THREADS} threads), {el:.1f} s; "
                           "optimizer step not included"}, None, {})
    raise ValueError(w)


def cpu_child(a):
    """--cpu-child DIR: every requested leg, each isolated from the others' failures; results on disk for the parent."""
    import numpy as np

    assert not any(os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")), "the CPU child must not see a GPU"
    rc = 0
    for w in filter(None, a.cpu_legs.split(",")):
        print(f"[bench cpu-child] leg {w} ...", file=sys.stderr, flush=True)
        try:
            cb, ref, extra = cpu_leg(a, w)
            if ref is not None:
                np.save(os.path.join(a.cpu_child, f"{w}_ref.npy"), ref)
            rec = {"cpu_baseline": cb, "extra": extra}
        except Exception as e:   # reported in the line; the other legs still run
            import traceback

            traceback.print_exc()
            rec = {"error": f"{type(e).__name__}: {e}"}
            rc = 1
        with open(os.path.join(a.cpu_child, f"{w}.json"), "w") as f:
            json.dump(rec, f)
    return rc


def run_cpu_child(a, argv, legs):
    """Start the CPU-baseline child (this process has not imported torch and has not touched the GPU), wait for it, collect
    {leg: {"cpu_baseline": ..., "ref": array | None, "extra": {...}} | {"error": ...}}."""
    import tempfile

    import numpy as np

    out = {}
    with tempfile.TemporaryDirectory(prefix="emd_cpu_") as d:
        env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        cmd = [sys.executable, os.path.abspath(__file__), *[x for x in argv if x not in ("--no-riders",)], "--cpu-child", d, "--cpu-legs", ",".join(legs)]
        t0 = time.perf_counter()
        proc = subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL)
        print(f"[bench] CPU-baseline child ({','.join(legs)}): exit code {proc.returncode}, {time.perf_counter() - t0:.0f} s", file=sys.stderr, flush=True)
        for w in legs:
            try:
                rec = json.load(open(os.path.join(d, f"{w}.json")))
            except OSError:
                rec = {"error": f"the CPU-baseline child wrote no result for {w} (exit code {proc.returncode})"}
            rp = os.path.join(d, f"{w}_ref.npy")
            rec["ref"] = np.load(rp) if os.path.exists(rp) else None
            out[w] = rec
    return out


def attach_cpu(out, cpu, w, got_fn=None):
    """Put leg w's CPU baseline (and, with got_fn, the parity figures against its reference output) into a workload's result."""
    rec = (cpu or {}).get(w)
    if not rec:
        return
    if "error" in rec:
        out["cpu_baseline"] = {"error": rec["error"]}
        return
    out["cpu_baseline"] = rec["cpu_baseline"]
    if got_fn is not None and rec.get("ref") is not None:
        out["psnr_vs_oracle_db"], out["rel_l2_vs_oracle"] = psnr_rel(got_fn(rec), rec["ref"])


def cpu_baseline_K(x_host, W, Bm, s, budget_s=10.0):
    """oracle/k_oracle.c (plain-C port of graph K, OpenMP over rows)."""
    import numpy as np

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
                      f"(PyTorch-CPU float32, {torch.get_num_threads()} threads), {el:.1f} s",
            # the figure is noisy (one pass, a shared host): the one-image pass before it is the second sample of this run, and the
            # committed lines of rounds 2-4 are the run-to-run range
            "spread": {"one_image_pass_before": round(S * S / 1e6 / t1, 4), "batch_pass": round(n * S * S / 1e6 / el, 4),
                       "committed_lines_r02_r04": [0.146, 0.195]}}, y


def local_batch(a, rank, world):
    """(images on this rank, first global index, images in the whole job) under --scaling."""
    if a.scaling == "strong":
        from emdenoise import input_pipeline as ip

        lo, hi = ip.shard_contiguous(a.batch, world, rank)
        return hi - lo, lo, a.batch
    return a.batch, rank * a.batch, a.batch * world


def shard_note(a, world, Bl, total):
    if a.scaling == "strong":
        return f"strong scaling: {total} images in total cut contiguously over {world} ranks ({Bl} on rank 0), no collective"
    return f"{world} x {Bl} whole images, no collective"


# ================================================================================================
# workloads
# ================================================================================================
def bench_K(a, torch, emdenoise, dev, timer, rank, world, cpu, primary):
    import numpy as np

    (B, first, total), H, W = local_batch(a, rank, world), a.size, a.size
    steps = a.steps if (primary and a.steps is not None) else 200
    warmup = a.warmup if (primary and a.warmup is not None) else 20
    x_host, params = inputs_K(a, B, first)
    pd = torch.from_numpy(params.packed()).to(dev)
    x = torch.from_numpy(x_host[:B]).to(dev)
    y = torch.empty_like(x)

    def step():
        if B:
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
    tr, src = pmc_traffic()
    traffic = None
    if tr and "K:k3_roll<8, 2>" in tr and (B, H, W) == (32, 512, 512):
        traffic = round(tr["K:k3_roll<8, 2>"]["hbm_bytes_per_launch_corrected"])
    out = {
        "value": total * H * W / 1e6 / (ms / 1e3), "unit": "MPx/s", "ms_per_step": ms, "median_hipevent_ms": timer.median_event_ms,
        "steps": steps, "warmup": warmup, "dtype": "f32",
        "config": {"workload": f"K: kernel denoiser depth 2 width 3 (noise-removal-kernels.py), [{B},{H},{W},1] fp32 per GPU",
                   "global_batch": total, "image": f"{H}x{W}x1", "sharding": shard_note(a, world, B, total)},
        "roofline": {"bound": "hbm", "kernel": "k3_roll<8,2>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                     "traffic_source": f"{src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, 2x FETCH correction)",
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "avg_launch_us": round(launch_us, 3),
                     "how": f"HIP events around a hipGraph of {n_burst} back-to-back launches (includes the ~1.5 us kernel boundary); "
                            "67 MB replayed back to back lives in the 256 MiB Infinity Cache -- batch256 below is the figure beyond it"},
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
    if B:
        attach_cpu(out, cpu, "K", lambda rec: y.cpu().numpy()[..., 0])
    return out


def bench_D(a, torch, emdenoise, dev, timer, rank, world, cpu, primary):
    import numpy as np

    from emdenoise import ops

    (B, first, total), H, W = local_batch(a, rank, world), a.size, a.size
    steps = a.steps if (primary and a.steps is not None) else (20 if primary else 10)
    warmup = a.warmup if (primary and a.warmup is not None) else (5 if primary else 2)
    x_host = synthetic_lq(max(B, 1), H, W, seed=1234 + first)
    weights = emdenoise.synthetic_weights()
    eng = emdenoise.DenoiserEngine(weights, dev, a.precision)
    x = torch.from_numpy(x_host[:B]).to(dev)
    box = [None]

    def step():
        if B:
            box[0] = eng.forward(x)

    if a.profile_clean and a.profile_streams == "single":
        eng.two_streams = eng.pipeline = False
    ms = timer.run(step, steps, warmup)
    med = timer.median_event_ms
    if a.profile_clean:
        return {"value": total * H * W / 1e6 / (ms / 1e3), "unit": "MPx/s", "ms_per_step": ms, "median_hipevent_ms": med, "steps": steps, "warmup": warmup,
                "dtype": "bf16x3 (split-bf16 MFMA inputs, fp32 accumulate and activations)",
                "config": {"workload": f"D: modified-Xception encoder-decoder (machine_learning/denoiser.py), [{B},{H},{W},1] fp32 per GPU", "profile_clean": True},
                "roofline": {"bound": "hbm", "achieved": round(d_graph_algorithmic_bytes(B, H) / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": round(d_graph_algorithmic_bytes(B, H) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None}}
    # per-kernel-family device time of ONE more step on the single-stream launch sequence (with the two half batches staggered on
    # two streams -- DenoiserEngine.forward -- a launch's event pair would also span the other half's kernels)
    two, pipe = eng.two_streams, eng.pipeline
    eng.two_streams = eng.pipeline = False
    try:
        with FamilyTimer(torch, ops) as fam:
            step()
    finally:
        eng.two_streams, eng.pipeline = two, pipe
    gemm_ms = fam.matrix_ms()
    dw_ms = fam.ms.get("dw3x3", 0.0) + fam.ms.get("dw3x3_split32", 0.0)
    dw_bytes = fam.bytes_.get("dw3x3", 0.0) + fam.bytes_.get("dw3x3_split32", 0.0)
    pw_ms, pw_flops = fam.ms.get("conv1x1_split32", 0.0), 3.0 * fam.flops.get("conv1x1_split32", 0.0)
    # ... and the 728-channel launches among them (36 middle-flow blocks + block 4 + ASPP + cnn3*: K or N = 728), which are the
    # matrix-core bound ones; the rest of the family (K, N <= 384 at 128^2) is HBM-bound
    pw728_ms = sum(e0.elapsed_time(e1) for (e0, e1), sh in zip(fam.ev.get("conv1x1_split32", []), fam.pw_shapes) if 728 in sh[:2])
    pw728_flops = 3.0 * sum(sh[3] for sh in fam.pw_shapes if 728 in sh[:2])
    pw728_n = sum(1 for sh in fam.pw_shapes if 728 in sh[:2])
    # what the matrix cores execute for those launches: K zero-padded to the MFMA's 16 and N to the 64-column wave tile (728 -> 736 and 768;
    # every tile variant of gemm_split.hip pads at least this much).  Reported beside pointwise_frac, never in its place.
    pw728_exec = 3.0 * sum(2.0 * sh[2] * (-(-sh[0] // 16) * 16) * (-(-sh[1] // 64) * 64) for sh in fam.pw_shapes if 728 in sh[:2])
    # north_star's "depthwise path": EVERY launch that contains a depthwise stage -- the standalone depthwise kernels and the fused
    # separable convs (depthwise -> pointwise in one kernel, one or two outputs) -- algorithmic bytes over their summed device time
    DWP = ("dw3x3", "dw3x3_split32", "sep_fused", "sep_fused_gen", "sep_dual")
    dwp_ms = sum(fam.ms.get(k, 0.0) for k in DWP)
    dwp_bytes = sum(fam.bytes_.get(k, 0.0) for k in DWP)
    scale = (B / 32.0) * (H * W) / (512.0 * 512.0)
    alg_flops = 2.0 * D_GMAC_MATRIX_B32_512 * 1e9 * scale
    passes = 3 if a.precision == "bf16x3" else 1
    alg_bytes = d_graph_algorithmic_bytes(B, H)
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    tr, src = pmc_traffic()
    traffic = None
    if tr and (B, H, W) == (32, 512, 512):   # HBM bytes of every launch of one step (PMC run of `steps_sampled` forwards)
        dk = {k: v for k, v in tr.items() if k.startswith("D:")}
        if dk:
            n = float(next(iter(dk.values())).get("steps_sampled", 4))
            traffic = round(sum(v["hbm_bytes_per_launch_corrected"] * v["launches_sampled"] / n for v in dk.values()))
    out = {
        "value": total * H * W / 1e6 / (ms / 1e3), "unit": "MPx/s", "ms_per_step": ms, "median_hipevent_ms": med, "steps": steps, "warmup": warmup,
        "dtype": "bf16x3 (split-bf16 MFMA inputs, fp32 accumulate and activations)" if passes == 3 else "bf16",
        "config": {"workload": f"D: modified-Xception encoder-decoder (machine_learning/denoiser.py), [{B},{H},{W},1] fp32 per GPU",
                   "global_batch": total, "image": f"{H}x{W}x1", "precision": a.precision, "sharding": shard_note(a, world, B, total)},
        "roofline": {"bound": "hbm", "kernel": "whole step (every launch of one forward pass; the graph is HBM-bound end to end, SURVEY.md 8d)",
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                     "algorithmic_bytes_per_step": alg_bytes,
                     "algorithmic_note": "fp32 activation bytes of a fully fused graph: in + out (+ residual) per layer, no depthwise intermediates "
                                         "(bench.d_graph_algorithmic_bytes; 80.0 GB at B=32, 512^2: SURVEY 8d's 118 GB unfused - 45 GB depthwise intermediates + 6.5 GB residual reads)",
                     "depthwise_frac": round(dw_bytes / (max(dw_ms, 1e-9) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                     "depthwise_path_frac": round(dwp_bytes / (max(dwp_ms, 1e-9) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                     "pointwise_frac": round(pw728_flops / (max(pw728_ms, 1e-9) * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                     "targets_note": "north_star's two targets, where the driver keeps them: depthwise_frac = standalone depthwise launches, "
                                     "depthwise_path_frac = every launch with a depthwise stage (standalone + fused separable convs), both algorithmic "
                                     "bytes / device time / 8 TB/s (target 0.70); pointwise_frac = issued bf16 flops of the 728-channel pointwise "
                                     "GEMMs / device time / 2.5 PFLOP/s (target 0.40); details under depthwise / depthwise_path / pointwise",
                     "traffic": traffic, "traffic_source": src,
                     "hbm_busy_frac": round(traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None,
                     "hbm_busy_note": "PMC bytes per step / step time / 8 TB/s (traffic above the algorithmic bytes = re-reads and unfused intermediates)",
                     "how": "step time = wall clock over the timed steps (barrier + synchronize on both sides); median of per-step HIP events beside it"},
        "matrix_cores": {"bound": "mfma", "kernel": "gemm_conv + gemm_split* + sep_fused / sep_dual + deconv (every matrix-core launch of one step)",
                         "algorithmic_flops_per_step": alg_flops, "mfma_passes": passes, "kernel_ms_per_step": round(gemm_ms, 3),
                         "achieved_tflops": round(alg_flops / (max(gemm_ms, 1e-9) * 1e-3) / 1e12, 1),
                         "issued_tflops": round(passes * alg_flops / (max(gemm_ms, 1e-9) * 1e-3) / 1e12, 1),
                         "frac_of_2500_algorithmic": round(alg_flops / (max(gemm_ms, 1e-9) * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                         "how": "HIP events around every launch of the family in one extra step (single-stream launch sequence)"},
        "depthwise": {"bound": "hbm", "kernel": "dw3x3_s1_roll / dw3x3_generic, fp32 or split32 output (standalone launches only)",
                      "launches": fam.launches.get("dw3x3", 0) + fam.launches.get("dw3x3_split32", 0),
                      "algorithmic_bytes_per_step": dw_bytes, "ms_per_step": round(dw_ms, 3),
                      "achieved_GBps": round(dw_bytes / (max(dw_ms, 1e-9) * 1e-3) / 1e9, 1),
                      "frac_of_8TBps": round(dw_bytes / (max(dw_ms, 1e-9) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
        "depthwise_path": {"bound": "hbm", "kernel": "every launch with a depthwise stage: dw3x3* (standalone) + sep_fused / sep_fused_gen / sep_dual (fused)",
                           "launches": sum(fam.launches.get(k, 0) for k in DWP), "algorithmic_bytes_per_step": dwp_bytes, "ms_per_step": round(dwp_ms, 3),
                           "achieved_GBps": round(dwp_bytes / (max(dwp_ms, 1e-9) * 1e-3) / 1e9, 1),
                           "frac_of_8TBps": round(dwp_bytes / (max(dwp_ms, 1e-9) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                           "by_family": {k: {"ms": round(fam.ms.get(k, 0.0), 3), "GB": round(fam.bytes_.get(k, 0.0) / 1e9, 2)} for k in DWP if k in fam.ms}},
        "pointwise": {"bound": "mfma", "kernel": "gemm_split16_kernel (the 1x1 halves of the 728-channel separable convs, LDS-DMA from split32)",
                      "launches": fam.launches.get("conv1x1_split32", 0), "issued_flops_per_step": pw_flops, "ms_per_step": round(pw_ms, 3),
                      "issued_tflops": round(pw_flops / (max(pw_ms, 1e-9) * 1e-3) / 1e12, 1),
                      "frac_of_2500": round(pw_flops / (max(pw_ms, 1e-9) * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                      "channels_728": {"launches": pw728_n, "issued_flops_per_step": pw728_flops, "ms_per_step": round(pw728_ms, 3),
                                       "issued_tflops": round(pw728_flops / (max(pw728_ms, 1e-9) * 1e-3) / 1e12, 1),
                                       "frac_of_2500": round(pw728_flops / (max(pw728_ms, 1e-9) * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                                       "mfma_executed_frac_of_2500": round(pw728_exec / (max(pw728_ms, 1e-9) * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                                       "mfma_executed_note": "the same launches with the zero padding the tiles carry (K 728 -> 736, N 728 -> 768): what the "
                                                             "matrix pipe is busy with; pointwise_frac counts the unpadded 3 x 2MKN only",
                                       "note": "the launches with K or N = 728 (the matrix-core bound ones); the other launches of the family are HBM-bound"}},
        "kernel_family_ms": fam.table(),
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
            tms = {-1: [], 3: [], 6: []}
            for _ in range(3):
                for v in (-1, 3, 6):
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
            for v, nm in ((-1, "default_16x16x32"), (3, "variant_32x32x16"), (6, "variant_w_through_registers_32x32x16")):
                us = float(np.median(tms[v]))
                iso[nm] = {"us": round(us, 1), "issued_tflops": round(fl / us / 1e6, 1), "frac_of_2500": round(fl / us / 1e6 / MFMA_BF16_PEAK_TFLOPS, 4)}
            out["pointwise"]["isolated_32768x728x728"] = iso
        except Exception as e:
            out["pointwise"]["isolated_32768x728x728"] = {"error": f"{type(e).__name__}: {e}"}
    if primary and rank == 0 and B and a.precision == "bf16x3":
        # the same graph through the library's native executor (emd_graph_create / _run: the launch sequence in C++, single stream)
        try:
            from emdenoise.graph_exec import NativeGraph

            nat = NativeGraph(weights, dev)
            yn = nat.forward(x)
            torch.cuda.synchronize()
            same = bool(torch.equal(yn, box[0]))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                nat.forward(x)
            e1.record()
            torch.cuda.synchronize()
            out["native_executor"] = {"ms_per_step": round(e0.elapsed_time(e1) / 5, 3), "bit_identical_to_python_engine": same,
                                      "workspace_GiB": round(nat.workspace_bytes(B, H) / 2 ** 30, 2),
                                      "note": "emd_graph_run (csrc/graph_exec.hip): layer table, BN folding, packing, kernel selection and launch order "
                                              "in the library; single stream (emd_graph_set_two_streams: the Python engine's two-halves form, measured slower from C)"}
            nat.close()
            del nat, yn
        except Exception as e:
            out["native_executor"] = {"error": f"{type(e).__name__}: {e}"}
    if primary and rank == 0 and B >= 8 and B % 8 == 0:
        # strong scaling of a global batch of B over 8 GPUs is B / 8 images per GPU: what ONE GPU makes of that sub-batch bounds the
        # 8-GPU figure from above (inference has no collective): projected_strong_scaling_8 = T(B) / T(B / 8).  Eager and as one
        # replayed hipGraph (a small batch's launch queue can drain).
        try:
            from emdenoise.graphed import GraphedForward

            xs8 = x[:B // 8].contiguous()

            def timed(fn, n=20):
                for _ in range(3):
                    fn(xs8)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    fn(xs8)
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / n

            t_eager = timed(eng.forward)
            t_graph = timed(GraphedForward(eng))
            t_small = min(t_eager, t_graph)
            out["small_batch"] = {"batch": B // 8, "ms_eager": round(t_eager, 3), "ms_hipgraph": round(t_graph, 3),
                                  "projected_strong_scaling_8": round(ms / t_small, 2),
                                  "note": f"T({B}) / T({B // 8}) on this GPU: the ceiling of the 8-GPU strong-scaling figure for a global batch of {B} "
                                          "(north_star target >= 6.5; inference shards whole images, no collective)"}
        except Exception as e:
            out["small_batch"] = {"error": f"{type(e).__name__}: {e}"}
    if primary and rank == 0 and B:
        # end to end from pinned host memory: H2D copy + forward + D2H copy of the same batch
        try:
            xp = torch.from_numpy(x_host[:B]).pin_memory()
            yp = torch.empty((B, H, W, 1), dtype=torch.float32).pin_memory()
            ts = []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                xd = xp.to(dev, non_blocking=True)
                yp.copy_(eng.forward(xd), non_blocking=True)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            out["from_pinned_host"] = {"ms_per_step": round(min(ts), 3), "MPx_per_s": round(B * H * W / 1e6 / (min(ts) / 1e3), 1),
                                       "note": "pinned host -> HBM, forward, HBM -> pinned host, one batch, best of 3 (never the headline value)"}
        except Exception as e:
            out["from_pinned_host"] = {"error": f"{type(e).__name__}: {e}"}
    if B:
        attach_cpu(out, cpu, "D", lambda rec: box[0][:1].cpu().numpy())
    return out


def bench_X(a, torch, emdenoise, dev, timer, rank, world, cpu, primary):
    """BASELINE configs[2], second graph of that name: misc_py/modified_Xception.py at 512x512."""
    import numpy as np

    from emdenoise import ops, xception as X

    (B, first, total), H, W = local_batch(a, rank, world), a.size, a.size
    steps = a.steps if (primary and a.steps is not None) else 3
    warmup = a.warmup if (primary and a.warmup is not None) else 1
    x_host = synthetic_lq(max(B, 1), H, W, seed=1234 + first)
    weights = X.synthetic_weights()
    eng = X.XceptionEngine(weights, dev, a.precision)
    x = torch.from_numpy(x_host[:B]).to(dev)
    box = [None]

    def step():
        if B:
            box[0] = eng.forward(x)

    ms = timer.run(step, steps, warmup)
    if a.profile_clean:   # for rocprofv3: the timed steps and nothing else (graph X is single stream by construction)
        return {"value": round(total * H * W / 1e6 / (ms / 1e3), 1), "unit": "MPx/s", "ms_per_step": round(ms, 3), "steps": steps, "warmup": warmup,
                "dtype": "bf16x3", "config": {"workload": f"X: [{B},{H},{W},1] fp32 per GPU", "profile_clean": True}, "roofline": None}
    with FamilyTimer(torch, ops) as fam:
        step()
    scale = (B / 32.0) * (H * W) / (512.0 * 512.0)
    dec = ("conv3x3", "conv3x3_split32", "deconv3x3s2", "deconv3x3s2_split32", "deconv3x3s2_fused")
    dec_ms = sum(fam.ms.get(k, 0.0) for k in dec)
    dec_fl = sum(fam.flops.get(k, 0.0) for k in dec)
    out = {"value": round(total * H * W / 1e6 / (ms / 1e3), 1), "unit": "MPx/s", "ms_per_step": round(ms, 3), "median_hipevent_ms": timer.median_event_ms,
           "steps": steps, "warmup": warmup, "dtype": "bf16x3 (split-bf16 MFMA inputs, fp32 accumulate and activations)",
           "config": {"workload": f"X: Xception autoencoder (misc_py/modified_Xception.py), [{B},{H},{W},1] fp32 per GPU",
                      "precision": a.precision, "algorithmic_tflop_per_step": round(9.01 * scale, 3), "sharding": shard_note(a, world, B, total)},
           "tflops_algorithmic": round(9.01 * scale / (ms / 1e3), 1),
           "roofline": {"bound": "mfma", "kernel": "decoder family: dense 3x3 convs + transposed convs (gemm_split_conv / gemm_conv), 60 % of X's flops",
                        "achieved": round(dec_fl / (max(dec_ms, 1e-9) * 1e-3) / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(dec_fl / (max(dec_ms, 1e-9) * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                        "traffic": step_traffic("X")[0] if (B, H, W) == (32, 512, 512) else None, "traffic_source": step_traffic("X")[1],
                        "traffic_note": "HBM bytes of one WHOLE step (PMC), not of the decoder family alone",
                        "issued_frac": round(3.0 * dec_fl / (max(dec_ms, 1e-9) * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                        "algorithmic_flops_per_step": dec_fl, "kernel_ms_per_step": round(dec_ms, 3),
                        "how": "HIP events around every launch of the family in one extra step; flops from the launch arguments"},
           "kernel_family_ms": fam.table()}
    if B and rank == 0 and H == W and H % 64 == 0 and a.precision == "bf16x3":
        # the same graph through the library's native executor (csrc/graph_exec_x.hip, emd_graph_create variant 2)
        try:
            from emdenoise.graph_exec import NativeGraph

            nat = NativeGraph(weights, dev, variant="X")
            yn = nat.forward(x)
            torch.cuda.synchronize()
            same = bool(torch.equal(yn, box[0]))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                nat.forward(x)
            e1.record()
            torch.cuda.synchronize()
            out["native_executor"] = {"ms_per_step": round(e0.elapsed_time(e1) / 3, 3), "bit_identical_to_python_engine": same,
                                      "workspace_GiB": round(nat.workspace_bytes(B, H) / 2 ** 30, 2)}
            nat.close()
            del nat, yn
        except Exception as e:
            out["native_executor"] = {"error": f"{type(e).__name__}: {e}"}
    if B and cpu and "X" in cpu and "error" not in cpu["X"]:
        # X normalises with the statistics of the batch it is given, so parity is defined per batch: the GPU runs the SAME
        # sub-batch the CPU leg ran (2 images: ~25 s of CPU work), and that pair is what rel_l2_vs_oracle compares
        n = cpu["X"]["extra"]["n"]
        attach_cpu(out, cpu, "X", lambda rec: eng.forward(x[:n].contiguous()).cpu().numpy())
        out["parity_note"] = f"GPU and oracle both on the sub-batch [{n},{H},{W},1] (batch-statistics norms: the output depends on the batch)"
    else:
        attach_cpu(out, cpu, "X")
    return out


def bench_G(a, torch, emdenoise, dev, timer, rank, world, cpu, primary):
    """BASELINE configs[4], forward part: the in-filling GAN's GENERATOR (misc_py/gan-infilling-100.py:133-374) on
    1/64-sampled 512x512 images."""
    from emdenoise import gan as GN, ops

    (B, first, total), S = local_batch(a, rank, world), a.size
    steps = a.steps if (primary and a.steps is not None) else 5
    warmup = a.warmup if (primary and a.warmup is not None) else 1
    x_host = inputs_G(a, B, first)
    weights = GN.synthetic_weights()
    eng = GN.GeneratorEngine(weights, dev, a.precision)
    x = torch.from_numpy(x_host[:B]).to(dev)
    box = [None]

    def step():
        if B:
            box[0] = eng.forward(x)

    ms = timer.run(step, steps, warmup)
    if a.profile_clean:   # for rocprofv3 (kernel statistics, PMC step totals): the timed steps and nothing else
        return {"value": round(total * S * S / 1e6 / (ms / 1e3), 1), "unit": "MPx/s in-filled", "ms_per_step": round(ms, 3), "steps": steps, "warmup": warmup,
                "dtype": "bf16x3", "config": {"workload": f"G: [{B},{S},{S},1] fp32 per GPU", "profile_clean": True}, "roofline": None}
    two = getattr(eng, "two_streams", None)
    if two is not None:
        eng.two_streams = False
    try:
        with FamilyTimer(torch, ops) as fam:
            step()
    finally:
        if two is not None:
            eng.two_streams = two
    tflop = GN.algorithmic_flops(S) * B / 1e12
    mm, mf = fam.matrix_ms(), fam.matrix_flops()
    out = {"value": round(total * S * S / 1e6 / (ms / 1e3), 1), "unit": "MPx/s in-filled", "ms_per_step": round(ms, 3),
           "median_hipevent_ms": timer.median_event_ms, "steps": steps, "warmup": warmup,
           "dtype": "bf16x3 (split-bf16 MFMA inputs, fp32 accumulate and activations)",
           "config": {"workload": f"G: in-filling generator forward (misc_py/gan-infilling-100.py), [{B},{S},{S},1] fp32 per GPU, "
                                  f"1/64 of the pixels given ({a.mask} mask)", "precision": a.precision,
                      "algorithmic_tflop_per_step": round(tflop, 3), "sharding": shard_note(a, world, B, total)},
           "tflops_algorithmic": round(tflop / (ms / 1e3), 1),
           "roofline": {"bound": "mfma", "kernel": "matrix-core family (sep_fused + gemm_split + gemm_conv launches of one step)",
                        "achieved": round(mf / (max(mm, 1e-9) * 1e-3) / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(mf / (max(mm, 1e-9) * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                        "traffic": step_traffic("G")[0] if (B, S) == (32, 512) else None, "traffic_source": step_traffic("G")[1],
                        "hbm_busy_frac": round(step_traffic("G")[0] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if (step_traffic("G")[0] and (B, S) == (32, 512)) else None,
                        "algorithmic_flops_per_step": mf, "kernel_ms_per_step": round(mm, 3),
                        "note": "<= 128 channels at 256-512 px: the graph is HBM-bound, the matrix-core fraction is low by construction"},
           "kernel_family_ms": fam.table()}
    if B:
        attach_cpu(out, cpu, "G", lambda rec: box[0][:1].cpu().numpy())
    return out


def bench_S(a, torch, emdenoise, dev, timer, rank, world, cpu, primary):
    """the small separable autoencoder of misc_py/apply_autoencoders.py (:91-187), the reference's
    own size: 160x160 crops, encoding_features 16, a batch of `--batch` crops per GPU with per-image batch statistics."""
    import numpy as np

    from emdenoise import autoencoder as AE, graphed

    (B, first, total), S = local_batch(a, rank, world), 160
    steps = a.steps if (primary and a.steps is not None) else 10
    warmup = a.warmup if (primary and a.warmup is not None) else 2
    x_host = inputs_S(a, B, first)
    weights = AE.synthetic_weights(16)
    eng = AE.AutoencoderEngine(weights, dev, 16)
    x = torch.from_numpy(x_host[:B]).to(dev)
    box = [None]
    # 45 launches of a few microseconds each: captured once into a hipGraph and replayed (--no-graph: eager launches).  The big
    # graphs (D, X, G) gain nothing from a replay -- their launch queue never drains -- and stay eager.
    fwd = eng.forward if a.no_graph else graphed.GraphedForward(eng)

    def step():
        if B:
            box[0] = fwd(x)

    ms = timer.run(step, steps, warmup)
    out = {"value": round(total * S * S / 1e6 / (ms / 1e3), 1), "unit": "MPx/s", "ms_per_step": round(ms, 3), "median_hipevent_ms": timer.median_event_ms,
           "steps": steps, "warmup": warmup, "dtype": "bf16x3 (split-bf16 MFMA inputs, fp32 accumulate and activations)",
           "config": {"workload": f"S: separable autoencoder (misc_py/apply_autoencoders.py), [{B},{S},{S},1] fp32 per GPU, "
                                  "encoding_features 16, per-image batch-statistics norms", "precision": "bf16x3", "hip_graph": not a.no_graph},
           "roofline": whole_step_traffic_roofline("S", ms, "whole forward (45 short launches at 160 px: launch / latency bound)")}
    if B:
        attach_cpu(out, cpu, "S", lambda rec: box[0].cpu().numpy())
    return out


def bench_A(a, torch, emdenoise, dev, timer, rank, world, cpu, primary):
    """BASELINE configs[4]: one iteration of the in-filling GAN's training loop (misc_py/gan-infilling-100.py:1650-1790)
    on `--gan-batch` 512x512 images per GPU: generator towers through the discriminator (feature matching), generator
    Adam step, then the discriminator trained on the generated and the natural images (2T towers) and its Adam step."""
    import numpy as np

    from emdenoise import gan as GN, gan_trainer as GT

    T, S = a.gan_batch, a.size
    steps = a.steps if (primary and a.steps is not None) else 3
    warmup = a.warmup if (primary and a.warmup is not None) else 1
    hq, lq, offsets = inputs_A(a, rank)
    D = GT.DiscriminatorTrainer(GN.discriminator_synthetic_weights(), dev, a.precision)
    G = GT.GeneratorTrainer(GN.synthetic_weights(), D, dev, a.precision)
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
           "median_hipevent_ms": timer.median_event_ms,
           "steps": steps, "warmup": warmup, "dtype": "bf16x3 GEMMs (split-bf16 MFMA inputs, fp32 accumulate), fp32 elsewhere",
           "config": {"workload": f"A: in-filling GAN training iteration (misc_py/gan-infilling-100.py), [{T},{S},{S},1] fp32 per GPU: "
                                  f"{T} generator towers + Adam, {2 * T} discriminator towers + Adam",
                      "global_batch": T * world, "precision": a.precision, "parallelism": f"dp{world}",
                      "streams": a.train_streams, "hip_graph": loop is not None},
           "d_fake_first": float(rg[0, 0].item()), "d_out_first": float(rd[0, 0].item())}
    out["roofline"] = whole_step_traffic_roofline("A", ms, "whole iteration (<= 128-channel separable convs at 256-512 px dominate)")
    attach_cpu(out, cpu, "A")
    return out


def bench_T(a, torch, emdenoise, dev, timer, rank, world, cpu, primary):
    """BASELINE configs[3]: graph D' TRAINING (misc_py/denoiser-multi-gpu.py): data-parallel steps of `--train-batch`
    512x512 LQ/HQ pairs per GPU (bs=64 over 8 GPUs => 8 per GPU), towers of `--tower-batch` images (1 = the
    reference, :763), gradients averaged over all towers and ranks (one RCCL all-reduce of the flat gradient vector),
    Nesterov momentum step, weights re-packed on the device."""
    import numpy as np

    from emdenoise import denoiser as D, trainer as TR

    B, S, tb = a.train_batch, a.size, a.tower_batch
    steps = a.steps if (primary and a.steps is not None) else 3
    warmup = a.warmup if (primary and a.warmup is not None) else 1
    hq, lq = inputs_T(a, rank)
    weights = D.synthetic_weights(variant="Dprime")
    tr = TR.DenoiserTrainer(weights, dev, a.precision)
    x, t = torch.from_numpy(lq).to(dev), torch.from_numpy(hq).to(dev)
    box = [None]

    def step():
        box[0] = tr.train_step(x, t, tower_batch=tb, streams=a.train_streams, graph=not a.no_graph, batched=a.tower_mode == "batched")

    ms = timer.run(step, steps, warmup)
    if a.profile_clean:   # for rocprofv3: run with --no-graph so that the trace names every kernel of the step
        return {"value": round(B * S * S / 1e6 * world / (ms / 1e3), 2), "unit": "MPx/s trained", "ms_per_step": round(ms, 3), "steps": steps, "warmup": warmup,
                "dtype": "bf16x3", "config": {"workload": f"T: [{B},{S},{S},1] pairs per GPU, towers of {tb}, {a.tower_mode}", "profile_clean": True,
                                              "hip_graph": not a.no_graph}, "roofline": None}
    tflop = 3 * 5.38 / 32.0 * B * (S * S) / (512.0 * 512.0)   # forward + data gradient + weight gradient
    out = {"value": round(B * S * S / 1e6 * world / (ms / 1e3), 2), "unit": "MPx/s trained", "ms_per_step": round(ms, 3),
           "median_hipevent_ms": timer.median_event_ms,
           "steps": steps, "warmup": warmup, "dtype": "bf16x3 GEMMs (split-bf16 MFMA inputs, fp32 accumulate), fp32 elsewhere",
           "config": {"workload": f"T: graph D' training step (misc_py/denoiser-multi-gpu.py), [{B},{S},{S},1] fp32 LQ/HQ pairs per GPU, "
                                  f"towers of {tb}, Nesterov momentum 0.9, lr 1e-3",
                      "global_batch": B * world, "tower_batch": tb, "tower_mode": a.tower_mode, "batched_groups": (TR.DenoiserTrainer.batched_groups(B) if a.tower_mode == "batched" and tb == 1 else None), "streams": a.train_streams, "hip_graph": not a.no_graph,
                      "precision": a.precision, "parallelism": f"dp{world}",
                      "algorithmic_tflop_per_step_per_gpu": round(tflop, 3)},
           "tflops_algorithmic": round(tflop / (ms / 1e3), 1),
           "loss_first_tower": float(box[0][0, 1].item())}
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        # the step's only exchange, timed on its own (SURVEY.md 8d cfg 4): all-reduce of the flat gradient + broadcast of
        # the moving statistics (with one rank the process group exists but there is nothing to exchange: RCCL init only)
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
        out["allreduce"] = {"ms": round(ar_ms, 3), "bytes": nbytes, "ranks": world,
                            "bus_GBps": round(2.0 * (world - 1) / world * nbytes / (max(ar_ms, 1e-9) / 1e3) / 1e9, 1),
                            "note": "RCCL all-reduce (sum) of the fp32 gradient vector + broadcast of the moving statistics, HIP events over 5 rounds"}
    out["roofline"] = {"bound": "mfma", "kernel": "whole step (forward + data-gradient + weight-gradient GEMMs dominate)", "achieved": out["tflops_algorithmic"],
                       "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": round(out["tflops_algorithmic"] / MFMA_BF16_PEAK_TFLOPS, 4),
                       "traffic": step_traffic("T")[0] if (B, S, tb, a.tower_mode) == (8, 512, 1, "batched") else None,
                       "traffic_source": step_traffic("T")[1]}
    if out["roofline"]["traffic"]:
        out["roofline"]["hbm_busy_frac"] = round(out["roofline"]["traffic"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
    attach_cpu(out, cpu, "T")
    return out


BENCHES = {"K": bench_K, "D": bench_D, "X": bench_X, "T": bench_T, "G": bench_G, "A": bench_A, "S": bench_S}
METRIC = {"T": "megapixels/sec trained (512x512x1 LQ/HQ pairs)", "G": "megapixels/sec in-filled (512x512x1 bs=32)",
          "S": "megapixels/sec restored (160x160x1 crops, bs=32)", "A": "megapixels/sec trained (in-filling GAN, 512x512x1)"}


# ================================================================================================
def dry_run(a, rank, world, real_stdout=None):
    """The launch / rendezvous / sharding / timing / reporting path with a host stand-in for the step: no GPU, no kernels.
    Exercised by tests/test_bench_launch.py with two ranks over gloo."""
    import numpy as np
    import torch
    import torch.distributed as dist

    if "WORLD_SIZE" in os.environ:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    d = dist if dist.is_initialized() else None
    timer = Timer(torch, d, torch.device("cpu"))
    Bl, first, total = local_batch(a, rank, world)
    S = min(a.size, 64)
    x = synthetic_lq(max(Bl, 1), S, S, seed=1234 + first)[:Bl]
    acc = [0.0]

    def step():
        acc[0] += float(np.square(x).sum())   # stand-in for one pass over this rank's shard

    steps, warmup = a.steps or 3, a.warmup if a.warmup is not None else 1
    ms = timer.run(step, steps, warmup)
    shards = [None] * world
    if d is not None:
        dist.all_gather_object(shards, (first, first + Bl))
        dist.barrier()
        dist.destroy_process_group()
    else:
        shards = [(first, first + Bl)]
    if rank == 0:
        print(json.dumps({"metric": "megapixels/sec restored (512x512x1 bs=32)", "value": round(total * S * S / 1e6 / (ms / 1e3), 3), "unit": "MPx/s",
                          "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 5), "higher_is_better": True,
                          "scaling": a.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry_run": True,
                          "config": {"workload": f"dry run: host stand-in for the step on [{Bl},{S},{S},1] per rank", "global_batch": total,
                                     "shards": shards, "sharding": shard_note(a, world, Bl, total)},
                          "roofline": None, "cpu_baseline": None}), file=real_stdout or sys.stdout, flush=True)
    return 0


def worker(a):
    # stdout carries ONE JSON line and nothing else: libraries that print to fd 1 (RCCL's version banner at communicator
    # creation does) are sent to stderr for the life of the process, the line goes to a duplicate of the real stdout
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    try:
        return _worker(a, real_stdout)
    finally:
        real_stdout.flush()


def _worker(a, real_stdout):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.dry_run:
        return dry_run(a, rank, world, real_stdout)

    multi = world > 1
    if a.workload in ("all", "both"):
        primary = "D"
        # more than one rank: the primary workload only.  The training step is the one workload with a collective, and a failure in
        # a rider would take the inference scaling figure down with it (a rider's exception tears every rank down): its scaling
        # is measured on its own with `--workload T --gpus N`
        riders = [] if (a.no_riders or multi) else (["K"] if a.workload == "both" else ["K", "X", "T", "G", "S", "A"])
    else:
        primary, riders = a.workload, []
    # the CPU baselines FIRST, in a child that never sees a GPU, while this process has not yet loaded torch or the HIP runtime
    cpu = None
    if a.profile_clean:
        riders = []
    if rank == 0 and not multi and not a.no_cpu_baseline and not a.profile_clean:
        assert "torch" not in sys.modules, "the CPU-baseline child must be started before this process loads torch / HIP"
        cpu = run_cpu_child(a, a.argv, [primary] + [w for w in riders if w != "A"])

    import torch

    import emdenoise

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if "WORLD_SIZE" in os.environ:   # under a launcher, also with one rank: the RCCL process group is part of what is exercised
        import torch.distributed as dist_mod

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist_mod.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        dist = dist_mod
    timer = Timer(torch, dist, dev)

    def note(msg):   # progress on stderr: a run that dies mid-way says where (stdout carries the one JSON line only)
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    note(f"workload {primary} (primary) ...")
    res = {primary: BENCHES[primary](a, torch, emdenoise, dev, timer, rank, world, cpu, True)}
    note(f"workload {primary}: {res[primary]['ms_per_step']:.3f} ms per step")
    failed = []
    for w in riders:
        try:
            note(f"rider {w} ...")
            res[w] = BENCHES[w](a, torch, emdenoise, dev, timer, rank, world, cpu, False)
            note(f"rider {w}: {res[w]['ms_per_step']:.3f} ms per step")
        except Exception as e:
            if multi:   # the other ranks sit in this workload's collectives: fail the job, the launcher tears every rank down
                raise
            import traceback

            traceback.print_exc()
            res[w] = {"error": f"{type(e).__name__}: {e}"}
            failed.append(w)

    prim = res[primary]
    out = {
        "metric": METRIC.get(primary, "megapixels/sec restored (512x512x1 bs=32)"),
        "value": round(prim["value"], 1),
        "unit": prim.get("unit", "MPx/s"),
        "n_gpus": world,
        "steps": prim["steps"],
        "warmup": prim["warmup"],
        "ms_per_step": round(prim["ms_per_step"], 5),
        "higher_is_better": True,
        "scaling": a.scaling if primary in ("D", "K", "X", "G", "S") else "weak",
        "vs_baseline": None,
        "dtype": prim["dtype"],
        "data": "synthetic",
        "config": prim["config"],
        "roofline": prim["roofline"],
    }
    for k in ("median_hipevent_ms", "cpu_baseline", "rel_l2_vs_oracle", "psnr_vs_oracle_db", "parity_note", "matrix_cores", "depthwise", "pointwise",
              "depthwise_path", "small_batch", "kernel_family_ms", "from_pinned_host", "native_executor", "tflops_algorithmic", "loss_first_tower", "allreduce", "d_fake_first", "d_out_first"):
        if k in prim:
            out[k] = prim[k]
    if rank == 0 and primary == "D" and not multi and not a.profile_clean:
        note("on-box peaks ...")
        pk = out["measured_peaks"] = measured_peaks(torch, dev)
        # the same fractions against what THIS device delivers (nominal and measured peaks, both stated)
        if "stream_copy_GBps" in pk:
            out["roofline"]["frac_of_measured_stream_copy"] = round(out["roofline"]["achieved"] / pk["stream_copy_GBps"], 4)
            if out["roofline"].get("traffic"):
                out["roofline"]["hbm_busy_frac_of_measured_stream_copy"] = round(
                    out["roofline"]["traffic"] / (prim["ms_per_step"] * 1e-3) / 1e9 / pk["stream_copy_GBps"], 4)
            out["depthwise"]["frac_of_measured_stream_copy"] = round(out["depthwise"]["achieved_GBps"] / pk["stream_copy_GBps"], 4)
        if "mfma_bf16_TFLOPs" in pk:
            out["pointwise"]["frac_of_measured_mfma_peak"] = round(out["pointwise"]["issued_tflops"] / pk["mfma_bf16_TFLOPs"], 4)
            out["pointwise"]["channels_728"]["frac_of_measured_mfma_peak"] = round(
                out["pointwise"]["channels_728"]["issued_tflops"] / pk["mfma_bf16_TFLOPs"], 4)
    for w in riders:
        r = res[w]
        if "value" in r:
            r["value"] = round(r["value"], 2)
            r["ms_per_step"] = round(r["ms_per_step"], 5)
        out["workload_" + w] = r
    if failed:
        out["failed_riders"] = failed
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), file=real_stdout, flush=True)
    return 1 if failed else 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse_args(argv)
    a.argv = list(argv)
    if a.cpu_child:
        return cpu_child(a)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a, argv)
    return worker(a)


if __name__ == "__main__":
    sys.exit(main())
