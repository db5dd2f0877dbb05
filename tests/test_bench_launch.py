"""CPU tests of bench.py's multi-rank path: `python bench.py --gpus 2 --dry-run` starts its own ranks (a parent that has touched
neither torch nor the GPU runs torch.distributed.run as a child), the ranks rendezvous over gloo on 127.0.0.1, cut the batch
(weak: --batch per rank; strong: --batch in total, contiguous shards), time the stand-in step with the barrier / max-over-ranks
protocol and rank 0 prints ONE JSON line.  No GPU, no kernels."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, timeout=240):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env["OMP_NUM_THREADS"] = "2"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


def test_bench_docstring_and_parent_do_not_import_torch():
    """The self-launching parent must not initialise the GPU: it may not even import torch (or the package) before it forks."""
    code = ("import sys, bench; a = bench.parse_args(['--gpus', '2', '--dry-run']); "
            "assert 'torch' not in sys.modules and 'emdenoise' not in sys.modules and 'numpy' not in sys.modules, sorted(m for m in sys.modules if 'torch' in m)")
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT, timeout=60)


@pytest.mark.parametrize("scaling,batch,expect", [("weak", 3, [[0, 3], [3, 6]]), ("strong", 5, [[0, 3], [3, 5]])])
def test_self_launched_two_ranks_over_gloo(scaling, batch, expect):
    p, lines = run_bench("--gpus", "2", "--dry-run", "--scaling", scaling, "--batch", str(batch), "--steps", "4", "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout            # rank 0 only, one line
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 4 and r["warmup"] == 1 and r["scaling"] == scaling and r["dry_run"] is True
    assert [list(s) for s in r["config"]["shards"]] == expect
    assert r["config"]["global_batch"] == (6 if scaling == "weak" else 5)
    assert r["value"] > 0 and r["ms_per_step"] > 0 and r["higher_is_better"] is True


def test_single_rank_dry_run_needs_no_launcher():
    p, lines = run_bench("--dry-run", "--batch", "2")
    assert p.returncode == 0 and len(lines) == 1, p.stderr[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["config"]["shards"] == [[0, 2]]


def test_gpus_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=60)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_algorithmic_bytes_of_graph_d():
    sys.path.insert(0, ROOT)
    import bench

    gb = bench.d_graph_algorithmic_bytes(32, 512) / 1e9
    assert 78.0 < gb < 82.0, gb       # 118 GB unfused - 2 x 22.5 GB of depthwise intermediates + 6.5 GB of residual reads
    assert abs(bench.d_graph_algorithmic_bytes(1, 512) * 32 - bench.d_graph_algorithmic_bytes(32, 512)) < 1.0
