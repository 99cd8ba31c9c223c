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


@pytest.mark.parametrize("n_clips", [1, 2, 5])   # 1: fewer clips than ranks -> rank 1 idles, still gathers
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


def test_sharded_sampling_gloo_world8_eight_clips_mirrors_configs2():
    """BASELINE configs[2] at its own width: 8 clips over 8 ranks, one clip each, ONE all_gather (sample.py:267,305 iterate the
    same clips sequentially on one GPU).  gloo on CPU stands in for RCCL; the sharding / gather code is the same."""
    world = n_clips = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
    assert [r for r, _, _ in res] == list(range(world))
    assert all(ok for _, ok, _ in res)                                  # every rank holds all 8 latents, in clip order, bit-equal
    assert all(calls == [r] for r, _, calls in res)                     # clip i ran on rank i and nowhere else


def test_bench_py_eight_rank_dry_run_under_gloo():
    """bench.py --gpus 8 with MTV_BENCH_DRYRUN=1: eight processes under torch.distributed.run, gloo instead of RCCL, the sampler call
    replaced by a stub (no GPU here) -- everything else is bench.py's own N > 1 path: rendezvous from the env, barrier-bracketed timed
    region, per-rank step times gathered, the final all_gather of the latents, MAX over ranks, ONE JSON line on rank 0."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MTV_BENCH_DRYRUN="1", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "6", "--warmup", "2"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 8 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak"
    di = d["distributed"]
    assert di["backend"] == "gloo" and di["world_size_reported"] == 8 and di["latents_gathered"] == 8
    pr = di["per_rank_ms_per_step"]
    assert 0 < pr["min"] <= pr["median"] <= pr["max"] <= d["ms_per_step"] * 1.001
    assert abs(d["value"] - 8 * 6 / (d["ms_per_step"] * 6e-3)) <= 1e-2 * d["value"]      # whole-job aggregate: N K / max-over-ranks time


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


# ----------------------------------------------------------------------------------------------
# the real thing on the GPU box: RCCL (backend "nccl") + real DDPM.sample through the sharding helper
# ----------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_rccl_world1_real_sampler_matches_unsharded():
    """BASELINE configs[2] path, as far as one GPU can exercise it: an `nccl` (= RCCL) process group of world size 1
    on cuda:0, real DDPM.sample calls routed through sample_clips_sharded (clip per call and clips_per_call=2),
    compared with the un-sharded per-clip results.  The 8-GPU run only adds ranks: there is no other collective."""
    from conftest import NARROW_CFG
    from moditalker_amd import DDPM, DiffusionWrapper, UNetModel, filler
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    net = DiffusionWrapper(UNetModel(**NARROW_CFG, frames=16, max_batch=2)).eval()
    filler.fill_module_(net, seed=11, skip_prefixes=("output_bg_",))
    net = net.to(dev)
    S, L, n_clips = 6, 2048, 3
    dm = DDPM(net, channels=4, image_size=32, sampling_timesteps=S, w=0.0).to(dev)
    clips = []
    for i in range(n_clips):
        x, cond, ic = filler.synthetic_inputs(1, 32, 16, seed=40 + i, tag="rccl")
        noise = filler.noise_list(S, (1, 4, L), seed=40 + i, tag="rccl.noise")
        clips.append((cond.to(dev), ic.to(dev), [n.to(dev) for n in noise]))

    def one(i):
        cond, ic, noise = clips[i]
        return dm.sample(batch_size=1, cond=cond, image_cond=ic, noise=noise)

    def many(idx):
        cond = torch.cat([clips[i][0] for i in idx])
        ic = torch.cat([clips[i][1] for i in idx])
        noise = [torch.cat([clips[i][2][k] for i in idx]) for k in range(S)]
        return dm.sample(batch_size=len(idx), cond=cond, image_cond=ic, noise=noise)

    plain = [one(i) for i in range(n_clips)]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        sharded = sample_clips_sharded(one, n_clips)
        assert len(sharded) == n_clips and all(torch.equal(a, b) for a, b in zip(sharded, plain))
        batched = sample_clips_sharded(many, n_clips, clips_per_call=2)
        # a different batch size may pick other split-K tilings: fp32 summation-order noise only, over 6 steps
        assert all(float((a - b).abs().max()) <= 1e-4 for a, b in zip(batched, plain))
        # the gathered tensors really went through the collective: a second all_gather of a marker tensor works
        mark = torch.full((4,), 7.0, device=dev)
        got = [torch.empty_like(mark)]
        dist.all_gather(got, mark)
        assert torch.equal(got[0], mark)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_bench_py_distributed_path_single_rank():
    """bench.py's own N > 1 code path (RCCL init, per-rank timing gather, final all_gather of the latents, NUMA binding),
    exercised as far as one GPU allows: MTV_BENCH_FORCE_DIST=1 runs it with a world of one rank.  The driver's 8-GPU
    scaling run then only adds ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MTV_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "12", "--warmup", "2", "--ramp-steps", "10",
                        "--no-cpu-baseline", "--batched-clips", "0", "--no-autoencoder", "--profile-iters", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "weak" and d["steps"] == 12
    di = d["distributed"]
    assert di["backend"] == "nccl" and di["world_size_reported"] == 1 and di["latents_gathered"] == 1
    assert 0 < di["per_rank_ms_per_step"]["min"] <= di["per_rank_ms_per_step"]["max"] <= d["ms_per_step"] * 1.001
    assert di["gather_and_barrier_ms"] >= 0 and "cpu_affinity" in d and "steps_sampled" in d
