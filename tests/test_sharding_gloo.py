"""CPU, world_size 2 over gloo: the N>1 path -- contiguous sharding, independent per-rank work, optional
gather -- reproduces the unsharded result.  The per-rank 'engine' here is the oracle (no GPU in this container);
on the GPU box bench.py runs the same sharding with the HIP engine over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))

from iv_interpolation_amd import sharding, synth   # noqa: E402


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 1000, 1_000_001):
        for w in (1, 2, 3, 8):
            b = [sharding.shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_ragged_bounds_balance_bytes():
    d = synth.numpy_ragged_batch(2000, 16, 8, 128, seed=1)
    for w in (2, 4, 8):
        b = sharding.ragged_shard_bounds(d["k_off"], w)
        assert b[0][0] == 0 and b[-1][1] == 2000 and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        loads = [int(d["k_off"][hi] - d["k_off"][lo]) for lo, hi in b]
        assert max(loads) - min(loads) <= 2 * 128


def test_strong_scaling_shards_of_one_global_batch():
    """bench.py --scaling strong: every rank draws the SAME strike counts for the global ragged batch and takes the
    block ragged_shard_bounds gives it; the blocks tile the batch and carry equal strike totals (uniform batches:
    shard_bounds, sizes within one surface of each other)."""
    B = 100_003
    counts = synth.ragged_counts(B, 8, 128, seed=synth.BASE_SEED + 1000)
    assert np.array_equal(counts, synth.ragged_counts(B, 8, 128, seed=synth.BASE_SEED + 1000))      # deterministic
    assert counts.min() >= 8 and counts.max() <= 128
    k_off = np.concatenate([[0], np.cumsum(counts)])
    for world in (2, 4, 8):
        b = sharding.ragged_shard_bounds(k_off, world)
        assert b[0][0] == 0 and b[-1][1] == B and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        loads = [int(k_off[hi] - k_off[lo]) for lo, hi in b]
        assert max(loads) - min(loads) <= 2 * 128 and sum(loads) == int(k_off[-1])
        u = [sharding.shard_bounds(B, r, world) for r in range(world)]
        assert sum(hi - lo for lo, hi in u) == B


def _worker(rank, world, port, q):
    import ivs_oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = 37
    d = synth.numpy_batch(B, 16, 8, seed=synth.BASE_SEED)       # every rank knows the recipe, builds only its shard
    Kq, Tq = synth.query_grids(16, 8, nT=8)
    lo, hi = sharding.shard_bounds(B, rank, world)
    out, st = O.surface_batch(d["K"][lo:hi], d["T"], d["sigma"][lo:hi], Kq, Tq, O.CUBIC)
    full = sharding.gather_outputs(torch.from_numpy(out))
    dist.barrier()
    if rank == 0:
        ref, _ = O.surface_batch(d["K"], d["T"], d["sigma"], Kq, Tq, O.CUBIC)
        q.put(bool(np.array_equal(full.numpy(), ref)))
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True
