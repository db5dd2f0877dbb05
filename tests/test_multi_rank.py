"""CPU tests of the N>1 path: sharding host logic, and a world_size-2 run over gloo (no GPU) that checks
that the ranks' contiguous shards partition the batch and that step time is reduced as MAX over ranks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from emdenoise import input_pipeline as ip


def test_shard_contiguous_partitions():
    for n in (0, 1, 5, 32, 33):
        for world in (1, 2, 3, 8):
            parts = [ip.shard_contiguous(n, world, r) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        ip.shard_contiguous(4, 2, 2)


def test_shard_round_robin_matches_reference_order():
    """denoiser-multi-gpu.py:905-909: image i goes to tower i % num_shards."""
    batch = np.arange(10)
    shards = ip.shard_round_robin(batch, 4)
    assert [s.tolist() for s in shards] == [[0, 4, 8], [1, 5, 9], [2, 6], [3, 7]]
    assert ip.shard_round_robin(batch, 1)[0] is batch


def test_reference_host_functions():
    rng = np.random.default_rng(0)
    assert (ip.scale0to1(np.full((3, 3), 7.0)) == 0.5).all()              # constant image -> 0.5
    img = rng.random((16, 16)).astype(np.float32)
    for c in range(8):
        out = ip.flip_rotate(img, c)
        assert out.shape == (16, 16) and np.isclose(out.sum(), img.sum())
    assert len({ip.flip_rotate(img, c).tobytes() for c in range(8)}) == 8  # the 8 elements of D4 are distinct
    bad = img.copy()
    bad[0, 0], bad[1, 1] = np.nan, np.inf
    pre = ip.preprocess(bad, rng)
    assert np.isfinite(pre).all() and pre.min() == 0.0 and pre.max() == 1.0
    lq, hq = ip.record_parser(img, rng)
    assert lq.shape == hq.shape == (16, 16) and lq.min() == 0.0 and lq.max() == 1.0
    assert np.isclose(hq.mean(), lq.mean(), rtol=1e-5)                     # truth rescaled to the LQ mean (:868)
    assert ip.get_scale(rng) >= 25.0


def test_input_fn_shapes(tmp_path):
    stack = np.random.default_rng(1).random((6, 8, 8, 1)).astype(np.float32)
    np.save(tmp_path / "s.npy", stack)
    mm = ip.load_npy_stack(str(tmp_path / "s.npy"))
    batches = list(ip.input_fn(mm, batch_size=4, num_shards=2, seed=3))
    assert len(batches) == 1
    feats, truths = batches[0]
    assert len(feats) == len(truths) == 2 and feats[0].shape == (2, 8, 8, 1) and truths[1].shape == (2, 8, 8, 1)


class _FakeDenoiser:
    def denoise_batch(self, x):
        return x * 2.0


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    batch = np.arange(5 * 4, dtype=np.float32).reshape(5, 2, 2, 1)
    lo, hi, out = ip.denoise_sharded(_FakeDenoiser(), batch, rank, world)
    step = ip.max_over_ranks(1.0 + rank, dist)           # slowest rank defines the step time
    idx = [None] * world
    dist.all_gather_object(idx, (lo, hi, float(out.sum())))
    if rank == 0:
        q.put((idx, step))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_over_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    idx, step = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [(lo, hi) for lo, hi, _ in idx] == [(0, 3), (3, 5)]           # disjoint, covers the batch
    batch = np.arange(20, dtype=np.float32).reshape(5, 2, 2, 1)
    assert np.isclose(sum(s_ for _, _, s_ in idx), 2.0 * batch.sum())    # every image processed exactly once
    assert step == 2.0


def _train_sync_worker(rank, world, port, q):
    """The collective step of DenoiserTrainer.train_step on CPU tensors over gloo: gradient vectors are summed over
    ranks, moving statistics follow rank 0, and averaging over all gradient sets reproduces the single-process mean."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emdenoise import trainer as TR

    n_local = 3                                        # towers per rank
    sets = [torch.arange(10, dtype=torch.float32) * (1 + rank * n_local + k) for k in range(n_local)]
    grads = torch.stack(sets).sum(0)                   # towers accumulate into one flat vector
    moving = torch.full((4,), float(rank + 1))
    w = TR.sync_gradients(grads, moving)
    mean = grads / (n_local * w)
    out = [None] * world
    dist.all_gather_object(out, (w, mean.tolist(), moving.tolist()))
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_training_gradient_exchange_over_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = (np.arange(10) * np.mean(np.arange(1, 7))).tolist()   # mean over the 6 gradient sets of both ranks
    for w, mean, moving in out:
        assert w == 2 and np.allclose(mean, expect) and moving == [1.0] * 4


def test_sync_gradients_without_process_group_is_identity():
    from emdenoise import trainer as TR

    g, m = torch.ones(5), torch.zeros(3)
    assert TR.sync_gradients(g, m) == 1 and g.tolist() == [1.0] * 5
