"""Clip sharding over ranks (moditalker_amd/parallel.py) on the gloo backend, world_size 2, CPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from moditalker_amd.parallel import sample_clips_sharded, shard_indices


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_sample(i):
    # stands in for DDPM.sample of clip i: deterministic per clip, independent of the rank that runs it
    g = torch.Generator().manual_seed(1000 + i)
    return torch.randn(1, 4, 128, generator=g)


def _worker(rank, world, port, n_clips, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []

        def fn(i):
            calls.append(i)
            return _fake_sample(i)

        out = sample_clips_sharded(fn, n_clips)
        ok = len(out) == n_clips and all(torch.equal(out[i], _fake_sample(i)) for i in range(n_clips))
        q.put((rank, ok, calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_clips", [2, 5])
def test_sharded_sampling_gloo_world2(n_clips):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == shard_indices(n_clips, 0, 2) and res[1][2] == shard_indices(n_clips, 1, 2)
    assert sorted(res[0][2] + res[1][2]) == list(range(n_clips))      # every clip sampled exactly once


def test_without_process_group_is_a_plain_loop():
    out = sample_clips_sharded(_fake_sample, 3)
    assert all(torch.equal(out[i], _fake_sample(i)) for i in range(3))


def test_shard_indices():
    assert shard_indices(8, 3, 8) == [3]
    assert shard_indices(10, 1, 4) == [1, 5, 9]
    with pytest.raises(ValueError):
        shard_indices(4, 4, 4)


def _fake_batched(idx):
    return torch.cat([_fake_sample(i) for i in idx], dim=0)


def _worker_batched(rank, world, port, n_clips, k, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        calls = []

        def fn(idx):
            calls.append(list(idx))
            return _fake_batched(idx)

        out = sample_clips_sharded(fn, n_clips, clips_per_call=k)
        ok = len(out) == n_clips and all(torch.equal(out[i], _fake_sample(i)) for i in range(n_clips))
        q.put((rank, ok, calls))
    finally:
        dist.destroy_process_group()


def test_sharded_sampling_with_clips_batched_per_call_gloo_world2():
    """B > 1 clips per GPU call: same results, each rank's clips grouped k at a time."""
    n_clips, k = 7, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_batched, args=(r, 2, port, n_clips, k, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == [[0, 2, 4], [6]] and res[1][2] == [[1, 3, 5]]


def test_batched_per_call_without_process_group_and_shape_check():
    out = sample_clips_sharded(_fake_batched, 5, clips_per_call=2)
    assert len(out) == 5 and all(torch.equal(out[i], _fake_sample(i)) for i in range(5))
    with pytest.raises(ValueError):
        sample_clips_sharded(lambda idx: _fake_batched(idx)[:1], 4, clips_per_call=2)
    with pytest.raises(ValueError):
        sample_clips_sharded(_fake_batched, 4, clips_per_call=0)
