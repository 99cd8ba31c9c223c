#!/usr/bin/env python3
"""bench.py -- denoise-steps/sec of the MToV DDIM loop on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one B=1 UNet forward over a 16-frame 256x256 clip's tri-plane latent [1,4,2048]
+ the eta=1 DDIM update (BASELINE.json configs[1]: second-stage base UNet, 250 DDIM steps).
Each rank denoises its own clip (clip-sharded, weak scaling); the only collective is the final
all_gather of the 32 KiB latents, inside the timed region.  Weights are random-init including the
reference's zero-initialised tensors, inputs/noise synthetic, already resident
in HBM when the timed region starts.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, exact f32 matrix peak


def synth_weights_(module, device, seed):
    """Random-init EVERY tensor (incl. zero_module'd convs and GN affine) on the device RNG:
    >=2-D: U(-1,1)*sqrt(3/fan_in); GN gamma 1+0.2U; biases 0.2U (same recipe as filler.fill_tensor)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    with torch.no_grad():
        for k, v in module.state_dict().items():
            if not torch.is_floating_point(v):
                continue
            u = torch.rand(v.shape, generator=g, device=device) * 2 - 1
            if v.dim() >= 2:
                fan_in = v[0].numel()
                v.copy_(u * (3.0 / fan_in) ** 0.5)
            elif k.endswith("weight"):
                v.copy_(1.0 + 0.2 * u)
            else:
                v.copy_(0.2 * u)


def cycled_steps(dm, K):
    """K consecutive entries of the 250-step DDIM schedule (wrapping around if K > 250)."""
    from moditalker_amd.ddpm import ddim_step_table
    pairs = dm._time_pairs()
    seq = [pairs[i % len(pairs)] for i in range(K)]
    return ddim_step_table(dm.alphas_cumprod, dm.sqrt_recip_alphas_cumprod, dm.sqrt_recipm1_alphas_cumprod,
                           seq, dm.ddim_sampling_eta)


def bind_to_gpu_numa_node(dev):
    """Pin this process to the CPUs of the NUMA node its GPU hangs off (8 ranks each replaying ~430 graphs/s from the host
    is where clip-sharded scaling can be lost to cross-socket launches).  Best effort: returns what was done, never raises."""
    try:
        pr = torch.cuda.get_device_properties(dev)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return dict(numa_node=node, bound=False, why="no NUMA affinity reported for the device")
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        before = os.sched_getaffinity(0)
        cpus &= before
        if not cpus:
            return dict(numa_node=node, bound=False, why="node's CPUs are outside this process's cpuset")
        os.sched_setaffinity(0, cpus)
        return dict(numa_node=node, bound=True, cpus=len(cpus), pci=bdf, _before=sorted(before))
    except (OSError, ValueError, AttributeError) as e:
        return dict(bound=False, why=f"{type(e).__name__}: {e}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=250)
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=10, help="timed oracle steps for the cpu_baseline leg")
    ap.add_argument("--cpu-physical-probe", action="store_true",
                    help="also time the oracle at torch.set_num_threads(<physical cores>) (one warm-up step, then a 10 s cap): a side note in "
                         "cpu_baseline.physical_core_probe, never `value`")
    ap.add_argument("--profile-iters", type=int, default=5)
    ap.add_argument("--batched-clips", type=int, default=8,
                    help="extra, informational: steps/s with this many clips batched on ONE GPU (0 = skip); never `value`")
    ap.add_argument("--ramp-steps", type=int, default=200,
                    help="untimed denoising steps run BEFORE the warm-up so that plan build, tile tuning, graph capture and the "
                         "GPU's clock ramp (DVFS: a cold MI355X runs its first ~100 ms well below its sustained clock) all "
                         "happen outside the timed region; set-up, not steps")
    ap.add_argument("--no-autoencoder", action="store_true",
                    help="skip the informational autoencoder timings (decode_from_sample / extract of one 16-frame 256x256 clip)")
    ap.add_argument("--no-res64", action="store_true",
                    help="skip the informational configs[3] (R = 64) timing that rides in the default line as res64_info")
    ap.add_argument("--res", type=int, default=32, choices=(32, 64),
                    help="latent resolution R: 32 = BASELINE configs[1] (the metric's workload), 64 = configs[3] (512x512 clip)")
    args = ap.parse_args()

    import ctypes as C
    import torch.distributed as dist
    # MTV_BENCH_DRYRUN=1 (tests/test_distributed.py, no GPU): the N > 1 scaffolding of this file -- rendezvous, barrier-bracketed timed
    # region, per-rank times, final all_gather, MAX over ranks, one JSON line -- on gloo / CPU with the sampler call replaced by a stub.
    # The line says `"dry_run": true` and carries no roofline: it is a plumbing check, never a measurement.
    dry = os.environ.get("MTV_BENCH_DRYRUN") == "1"
    if not dry:
        from moditalker_amd import BASE_UNET_CONFIG, DDPM, DiffusionWrapper, UNetModel, _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    dev = torch.device("cpu") if dry else torch.device("cuda", local_rank)
    if not dry:
        torch.cuda.set_device(dev)
    affinity = bind_to_gpu_numa_node(dev) if os.environ.get("MTV_BENCH_NO_BIND") != "1" and not dry else dict(bound=False, why="MTV_BENCH_NO_BIND=1 or dry run")

    def sync():
        if not dry:
            torch.cuda.synchronize(dev)
    # (MTV_BENCH_FORCE_DIST=1: go through the RCCL path with a single rank too -- a self-test of the N>1 code)
    use_dist = world > 1 or os.environ.get("MTV_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    R, T, S = args.res, 16, 250
    L = R * R + 2 * T * R
    K, W = args.steps, args.warmup
    if not dry:
        BASE_UNET_CONFIG = dict(BASE_UNET_CONFIG, image_size=R)
        net = DiffusionWrapper(UNetModel(**BASE_UNET_CONFIG, frames=T, max_batch=1)).eval().to(dev)
        synth_weights_(net, dev, seed=1234)          # same weights on every rank (replica per GPU)
        dm = DDPM(net, channels=4, image_size=R, sampling_timesteps=S, w=0.0).to(dev)
        um = net.diffusion_model
    g = torch.Generator(device=dev)
    g.manual_seed(100 + rank)                    # each rank owns its clip and its noise stream
    cond = torch.rand(1, 8, L, generator=g, device=dev) * 2 - 1
    image_cond = torch.rand(1, 4, R * R, generator=g, device=dev) * 2 - 1
    x = torch.randn(1, 4, L, generator=g, device=dev)
    n_noise = max(K, W, args.ramp_steps, 1)
    noise = torch.randn(n_noise, 1, 4, L, generator=g, device=dev)
    if not dry:
        if os.environ.get("MTV_EAGER") == "1":
            um.set_eager(True)                       # plain launches instead of hipGraph replay
        ctx = um.hip_context(dev, 1)
        lib = _lib.load()
        stream = torch.cuda.current_stream(dev)

    def run(nsteps, xbuf):
        if dry:                                       # stub: stands in for the sampler call (about the product's pace, rank-dependent)
            time.sleep(nsteps * (0.002 + 0.0002 * rank))
            xbuf.mul_(0.5)
            return
        steps, n_draws = cycled_steps(dm, nsteps)
        _lib.check(lib.mtv_ddim_sample(ctx, xbuf.data_ptr(), cond.data_ptr(), image_cond.data_ptr(), R * R,
                                       noise.data_ptr(), n_noise, steps, nsteps, 1, C.c_void_p(stream.cuda_stream)),
                   "mtv_ddim_sample")

    def barrier():
        if use_dist:
            dist.barrier()
        sync()

    xw = x.clone()
    gpu_sections = {}                             # wall time of the sections in which the GPU works (host-synchronised on both sides)
    t_sec = time.perf_counter()
    run(max(args.ramp_steps, 1) if not dry else 1, xw)   # set-up: builds the plan, loads/tunes tiles, captures the graphs, ramps clocks
    sync()
    gpu_sections["setup_and_ramp"] = time.perf_counter() - t_sec
    t_sec = time.perf_counter()
    if W > 0:
        run(W, xw)                                # the W untimed warm-up steps of the contract
    barrier()
    gpu_sections["warmup"] = time.perf_counter() - t_sec
    xt = x.clone()
    t0 = time.perf_counter()
    run(K, xt)
    t_own = None
    if use_dist:                                  # final gather of the finished latents (32 KiB each)
        sync()                                    # (splits the timed region into this rank's K steps | the gather; costs one host sync)
        t_own = time.perf_counter() - t0
        out = [torch.empty_like(xt) for _ in range(world)]
        dist.all_gather(out, xt)
    barrier()
    dt = time.perf_counter() - t0
    if t_own is None:
        t_own = dt
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist_info = None
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # what every rank measured: its own K steps, and the whole region incl. the gather + closing barrier
        mine = torch.tensor([t_own, dt], device=dev, dtype=torch.float64)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        own = sorted(1e3 * float(v[0]) / K for v in allr)
        dist_info = dict(backend=dist.get_backend(), world_size_reported=dist.get_world_size(), visible_gpus=0 if dry else torch.cuda.device_count(),
                         per_rank_ms_per_step=dict(min=round(own[0], 4), median=round(own[len(own) // 2], 4), max=round(own[-1], 4)),
                         gather_and_barrier_ms=round(1e3 * max(float(v[1]) - float(v[0]) for v in allr), 3),
                         latents_gathered=[bool(torch.isfinite(o).all()) for o in out].count(True),
                         note="per_rank_ms_per_step = each rank's own K steps (host-synchronised before the gather); "
                              "`value` uses the max over ranks of the whole region, gather and closing barrier included")
    dt = float(tmax.item())
    gpu_sections["timed_steps"] = dt
    assert torch.isfinite(xt).all()

    result = None
    if rank == 0 and dry:
        result = {"metric": "DRY RUN of bench.py's distributed scaffolding (gloo, CPU, stub sampler) -- not a measurement", "dry_run": True,
                  "value": round(world * K / dt, 3), "unit": "stub-steps/s", "n_gpus": world, "steps": K, "warmup": W,
                  "ms_per_step": round(1e3 * dt / K, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                  "data": "none (stub)", "distributed": dist_info, "config": {"workload": "dry run", "clips": world}}
    elif rank == 0:
        work = um.work(dev)
        # ---- roofline of the dominant kernel family, measured live with hipEvents around every launch
        t_sec = time.perf_counter()
        prof = um.profile_forward(1, args.profile_iters, dev, step=True)   # the launches of one SAMPLER step
        gpu_sections["per_launch_profile"] = time.perf_counter() - t_sec
        fam = {}
        for p in prof:
            key = p["name"].split(":")[0]
            if key == "attn" and "+proj" in p["name"]:
                key = "attnproj"                   # k_deep_attn: attention core + proj_out of a deep level in one launch (csrc/deep.hip)
            if key == "attn" and " blk " in p["name"]:
                key = "attnblock"                  # k_deep_block: GroupNorm -> qkv -> attention -> proj_out of a deep level in one launch (csrc/block.hip)
            f = fam.setdefault(key, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
            f["ms"] += p["ms"]
            f["flops"] += p["flops"]
            f["bytes"] += p["bytes"]
            f["launches"] += 1
        conv = dict(ms=fam.get("conv3", {}).get("ms", 0) + fam.get("conv1", {}).get("ms", 0),
                    flops=fam.get("conv3", {}).get("flops", 0) + fam.get("conv1", {}).get("flops", 0),
                    bytes=fam.get("conv3", {}).get("bytes", 0) + fam.get("conv1", {}).get("bytes", 0),
                    launches=fam.get("conv3", {}).get("launches", 0) + fam.get("conv1", {}).get("launches", 0))
        attn = fam.get("attn", dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
        # (the conv family = k_conv, the window-staged k_conv_win, the pointwise k_conv_pw and the K-sliced k_deep_conv of the <= 128-token
        # levels: op names "conv3:" / "conv1:" whichever kernel the tile table names)
        deep = dict(ms=0.0, flops=0.0, bytes=0.0, launches=0)
        for p in prof:
            if p["name"].startswith("conv") and " d" in p["name"].split("[")[-1]:
                deep["ms"] += p["ms"]; deep["flops"] += p["flops"]; deep["bytes"] += p["bytes"]; deep["launches"] += 1
        step_ms_events = sum(f["ms"] for f in fam.values())
        # An event record between two launches costs a few us that rocprofv3 does not see.  Calibrate it
        # from this run: (sum of the event-bracketed launches) - (timed step) spread over the launches, and
        # take it off every launch so `avg_launch_us` is the kernel duration rocprofv3 reports
        # (profiles/r01_step_summary.txt: the two agree to ~1 %).  The raw figures are kept beside it.
        n_prof = max(1, len(prof))
        ev_ms = max(0.0, (step_ms_events - 1e3 * dt / K) / n_prof)
        for f in list(fam.values()) + [conv, attn, deep]:
            if "ms_raw" in f:                     # `attn` is fam["attn"] itself
                continue
            f["ms_raw"] = f["ms"]
            f["ms"] = max(1e-6, f["ms"] - ev_ms * f["launches"])
        dom_name, dom = ("k_conv", conv) if conv["ms"] >= attn["ms"] else ("k_attention", attn)
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
        # HBM bytes per launch from the PMC passes (cannot be collected inside this process): the committed
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summaries of this same command, gfx950 correction applied
        # (a constant of the committed profile, NOT a measurement of this run: `traffic_source.kind` says so)
        # The file is keyed by workload ("R32" = configs[1], "R64" = configs[3]): a workload without a committed pass prints
        # traffic / counters as null, never another workload's constants.
        traffic, traffic_src, pmc, hbm_counter_GBs = None, None, {}, None
        try:
            pmc_all = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")))
            pmc = dict(pmc_all["workloads"][f"R{R}"], source=pmc_all["source"])
            e = pmc[dom_name]
            traffic = round((2.0 * e["fetch_raw_MB_per_step"] + e["write_raw_MB_per_step"]) * 1e6 / e["launches_per_step"])
            traffic_src = dict(kind="committed", collected=pmc.get("collected"), commit=pmc.get("commit"), detail=pmc["source"])
            # counter bytes of the conv family (committed PMC passes) over THIS run's family time
            ec = pmc["k_conv"]
            hbm_counter_GBs = round((2.0 * ec["fetch_raw_MB_per_step"] + ec["write_raw_MB_per_step"]) * 1e6 / (conv["ms"] * 1e-3) / 1e9, 1)
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            pass
        roofline = dict(bound="mfma", kernel=dom_name if dom_name != "k_conv" else "k_conv / k_conv_win / k_conv_pw + k_deep_conv (every conv launch)", achieved=round(achieved, 3), peak=MFMA_F32_PEAK_TF,
                        unit="TFLOP/s", frac=round(achieved / MFMA_F32_PEAK_TF, 4), traffic=traffic, traffic_unit="bytes/launch",
                        traffic_source=traffic_src,
                        launches_per_step=dom["launches"], avg_launch_us=round(1e3 * dom["ms"] / max(1, dom["launches"]), 3),
                        avg_launch_us_with_event_overhead=round(1e3 * dom["ms_raw"] / max(1, dom["launches"]), 3),
                        event_overhead_us_per_launch=round(1e3 * ev_ms, 3),
                        algorithmic_bytes_per_launch=round(dom["bytes"] / max(1, dom["launches"])),
                        flops_per_step=dom["flops"],
                        hbm_counter_GBs=hbm_counter_GBs,            # conv family: (2 x FETCH_SIZE + WRITE_SIZE, committed PMC passes) / this run's family time
                        mfma_util_counter=pmc.get("k_conv", {}).get("mfma_util"),   # SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles (committed PMC pass, tools/pmc_util.py)
                        counter_kind="committed" if traffic is not None else None)
        # the HBM side of the same family, with BOTH byte definitions: SURVEY section 8(d)'s "fused conv/GN path"
        # (3x3 conv weights + fused 1x1 skip weights + their activations: 477 MB at configs[1]) and this build's wider
        # one (every k_conv launch incl. qkv/proj: weights once + activations in/out once)
        conv3 = fam.get("conv3", dict(ms=0.0, bytes=0.0))
        b8d = work["bytes_weights_conv"] + work["bytes_act_conv_path"]
        roofline["hbm_view"] = dict(
            peak_GBs=HBM_PEAK_GBS,
            survey_8d=dict(what="conv3x3 (+fused 1x1 skip) launches: weights once + activations in/out once",
                           algorithmic_bytes_per_step=b8d, ms_per_step=round(conv3["ms"], 4),
                           achieved_GBs=round(b8d / (conv3["ms"] * 1e-3) / 1e9, 1) if conv3["ms"] > 0 else 0.0,
                           frac=round(b8d / (conv3["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if conv3["ms"] > 0 else 0.0),
            all_k_conv=dict(what="every k_conv launch (3x3, 1x1 skip, qkv, proj_out)",
                            algorithmic_bytes_per_step=dom["bytes"] if dom_name == "k_conv" else conv["bytes"],
                            achieved_GBs=round(conv["bytes"] / (conv["ms"] * 1e-3) / 1e9, 1) if conv["ms"] > 0 else 0.0,
                            frac=round(conv["bytes"] / (conv["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if conv["ms"] > 0 else 0.0))
        if deep["launches"] and deep["ms"] > 0:
            # the only HBM-shaped part of the path at one clip per GPU: the weight-streaming 3x3 convs of the <= 128-token levels
            # (unet.py:178-207 at levels 2 / 3): algorithmic bytes (weights once + activations in / out once) over their launch time
            roofline["hbm_view"]["deep_levels"] = dict(
                what="k_deep_conv launches (3x3 convs of the <= 128-token levels): weights once + activations in/out once",
                launches=deep["launches"], algorithmic_bytes_per_step=deep["bytes"], ms_per_step=round(deep["ms"], 4),
                achieved_GBs=round(deep["bytes"] / (deep["ms"] * 1e-3) / 1e9, 1),
                frac=round(deep["bytes"] / (deep["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
            # ... and the family's own counter row (committed PMC passes of the k_deep_conv launches alone, tools/pmc_util.py) over THIS run's family time
            ed = pmc.get("k_deep_conv_only")
            if ed and "fetch_raw_MB_per_step" in ed:
                cb = (2.0 * ed["fetch_raw_MB_per_step"] + ed["write_raw_MB_per_step"]) * 1e6
                roofline["hbm_view"]["deep_levels"].update(
                    counter_bytes_per_step=round(cb), counter_GBs=round(cb / (deep["ms"] * 1e-3) / 1e9, 1),
                    counter_over_algorithmic=round(cb / deep["bytes"], 3) if deep["bytes"] else None, counter_kind="committed")
            roofline["hbm_view"]["headline"] = "deep_levels"
        families = {k: dict(ms_per_step=round(v["ms"], 4), launches=v["launches"],
                            tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 and v["flops"] else None)
                    for k, v in fam.items()}
        for tagk, label in ((",80,", "conv3_window_staged(k_conv_win)"), (",96,", "conv1_pointwise(k_conv_pw)")):
            sel = [p for p in prof if p["name"].startswith("conv") and tagk in p["name"].split("[")[-1]]
            if sel:
                ms = max(1e-6, sum(p["ms"] for p in sel) - ev_ms * len(sel))
                families[label] = dict(ms_per_step=round(ms, 4), launches=len(sel), tflops=round(sum(p["flops"] for p in sel) / (ms * 1e-3) / 1e12, 2))
        if deep["launches"]:
            # the weight-streaming convs of the deep levels on their own: algorithmic bytes (weights once + activations) per second
            families["conv_deep_levels(k_deep_conv)"] = dict(ms_per_step=round(deep["ms"], 4), launches=deep["launches"],
                                                             tflops=round(deep["flops"] / (deep["ms"] * 1e-3) / 1e12, 2),
                                                             algorithmic_GBs=round(deep["bytes"] / (deep["ms"] * 1e-3) / 1e9, 1))
        # ---- CPU baseline: the oracle (PyTorch CPU restatement of the reference) on this box's cores
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import ref_ddpm, ref_unet
            # (round 6: the oracle STAYS on the CPUs of the GPU's NUMA node, where bind_to_gpu_numa_node put this process: 8 threads wandering over
            # both sockets of a shared 256-CPU host gave 1.45-2.2 steps/s from run to run on one box; MTV_BENCH_CPU_WHOLE_BOX=1 restores the old rule)
            if affinity.get("bound") and os.environ.get("MTV_BENCH_CPU_WHOLE_BOX") == "1":
                os.sched_setaffinity(0, affinity["_before"])
            ncores = max(1, (os.cpu_count() or 2) // 2)
            sd = {k: v.detach().cpu() for k, v in net.state_dict().items() if "output_bg_" not in k}
            cfg = dict(BASE_UNET_CONFIG)
            cc, ic, xc = cond.cpu(), image_cond.cpu(), x.cpu()
            pairs = ref_ddpm.ddim_time_pairs(1000, S)
            nz = noise[:, :, :, :].cpu()
            buf = ref_ddpm.schedule_buffers()

            def cpu_steps(n, img, budget_s=None):
                # (budget_s: stop after the first step that ends past the budget -- a box whose 128 oracle threads crawl must not
                # turn the default run into many minutes; returns (sample, steps done))
                t_start, done = time.perf_counter(), 0
                for i in range(n):
                    t_step = time.perf_counter()
                    time_, time_next = pairs[i]
                    tt = torch.full((1,), time_, dtype=torch.long)
                    eps = ref_unet.unet_forward(sd, cfg, img, cc, ic, tt, R, T)
                    x0 = (buf["sqrt_recip_alphas_cumprod"][time_] * img - buf["sqrt_recipm1_alphas_cumprod"][time_] * eps).clamp_(-1, 1)
                    a, an = buf["alphas_cumprod"][time_], buf["alphas_cumprod"][time_next]
                    sigma = ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
                    c = (1 - an - sigma ** 2).sqrt()
                    img = x0 * an.sqrt() + c * eps + sigma * nz[i]
                    done += 1
                    step_times.append(time.perf_counter() - t_step)
                    if budget_s is not None and time.perf_counter() - t_start > budget_s:
                        break
                return img, done

            step_times = []                        # wall time of every oracle step, in call order (cpu_steps appends)

            # ONE figure, measured the way BASELINE.md section 3 asks for config 2 -- >= 10 timed steps after 2 warm-up steps -- at the thread
            # count this B=1 workload actually scales to.  `torch.set_num_threads(<physical cores>)` on the 128-core GPU hosts does not
            # give a usable number: rounds 1-5 recorded 2.01 / 0.287 / 0.0217 / 2.03 / 0.0115 steps/s for it on five boxes (oneDNN / bmm at
            # 2048 tokens stop scaling at 8-32 threads; beyond that the OpenMP team mostly spins), one un-warmed step taking up to 87 s.
            # So: `value` is the 10-step figure at 8 threads, `cores` = the threads actually used; a short probe (1 warm + 1 timed step each) over
            # 8 / 16 / 32 threads rides along as information.  The physical-core figure is opt-in (--cpu-physical-probe), warm, capped.
            model = ""
            try:
                model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
            except (OSError, IndexError):
                pass
            n_timed = max(10, args.cpu_steps)
            t_cpu0 = time.perf_counter()
            cand = sorted({t for t in (8, 16, 32) if t <= max(8, ncores)})
            probe = {}
            for tcount in cand:
                torch.set_num_threads(tcount)
                cpu_steps(1, xc)
                t1 = time.perf_counter()
                cpu_steps(1, xc)
                probe[tcount] = time.perf_counter() - t1
            # `value` is timed at a FIXED count -- 8 threads, where this workload stops scaling on most hosts measured -- so that two boxes report the
            # same experiment (round 6: the best-of-probe rule picked 8 threads on two boxes, 1.90 / 1.95 steps/s, and 16 on a third, 2.44: the
            # probe's winner is box-dependent, the 8-thread figure is not, 1.90-2.04); the probe stays in the line as information
            probe_best = min(probe, key=probe.get)
            best_t = 8 if 8 in probe else probe_best
            torch.set_num_threads(best_t)
            cpu_steps(2, xc)                      # the 2 warm-up steps
            tc = time.perf_counter()
            del step_times[:]
            _, cpu_done = cpu_steps(n_timed, xc, budget_s=40.0)
            tc = time.perf_counter() - tc
            timed = sorted(step_times)
            med = timed[len(timed) // 2] if len(timed) % 2 else 0.5 * (timed[len(timed) // 2 - 1] + timed[len(timed) // 2])
            phys = None
            if args.cpu_physical_probe:           # side note only: <= 10 s per leg, one warm-up step first; never `value`
                torch.set_num_threads(ncores)
                tw = time.perf_counter()
                cpu_steps(1, xc)
                tw = time.perf_counter() - tw
                if tw < 10.0:
                    tp = time.perf_counter()
                    _, pdone = cpu_steps(n_timed, xc, budget_s=10.0)
                    tp = time.perf_counter() - tp
                    phys = dict(threads=ncores, value=round(pdone / tp, 4), steps=pdone, sample="after one warm-up step, 10 s cap")
                else:
                    phys = dict(threads=ncores, value=round(1.0 / tw, 4), steps=1, sample=f"the warm-up step alone took {tw:.0f} s: not repeated")
                torch.set_num_threads(best_t)
            try:
                aff_n = len(os.sched_getaffinity(0))
            except (OSError, AttributeError):
                aff_n = None
            # `value` = 1 / MEDIAN step time of the timed steps: the GPU hosts are shared machines, and a burst of somebody else's load during two of
            # the ten steps moved the mean by 25 % between two runs on one box (round 6: 1.45 and 1.90 steps/s at the same 0.46-0.50 s probe step);
            # the mean is reported beside it
            cpu = dict(value=round(1.0 / med, 4), mean_value=round(cpu_done / tc, 4), step_s=dict(min=round(timed[0], 3), median=round(med, 3), max=round(timed[-1], 3)),
                       unit="denoise-steps/s", cores=best_t, kind="port",
                       sample=f"1 / median step time of {cpu_done} timed steps after 2 warm-up steps (BASELINE.md section 3, config 2) at torch.set_num_threads({best_t}) (fixed: the count at which "
                              f"this B=1 workload stops scaling on most hosts; a 1-step probe over {cand} threads is reported beside it, fastest here: {probe_best}); the first steps of the 250-step DDIM schedule of the same clip; oracle/ref_unet.py "
                              f"op-for-op PyTorch {torch.__version__} CPU restatement, fp32; {os.cpu_count()} logical CPUs ({ncores} physical), {model}",
                       thread_probe_s_per_step={str(k): round(v, 3) for k, v in probe.items()},
                       affinity_cpus=aff_n, omp_proc_bind=os.environ.get("OMP_PROC_BIND"), omp_places=os.environ.get("OMP_PLACES"),
                       omp_num_threads_env=os.environ.get("OMP_NUM_THREADS"), physical_cores=ncores, physical_core_probe=phys,
                       leg_wall_s=round(time.perf_counter() - t_cpu0, 1),
                       note="`cores` = threads used for `value`; the all-physical-core setting is not timed by default (it crawls on the 128-core "
                            "hosts: 0.01-2.0 steps/s across five boxes, rounds 1-5) -- --cpu-physical-probe adds it as a capped side note")
        # ---- informational only: the same loop with several clips batched on this GPU (amortises the
        # per-launch floor and the weight stream; NOT the BASELINE workload, never `value`)
        batched = None
        if world == 1 and args.batched_clips > 1:
            Bc = args.batched_clips
            t_sec = time.perf_counter()
            netb = DiffusionWrapper(UNetModel(**BASE_UNET_CONFIG, frames=T, max_batch=Bc)).eval().to(dev)
            netb.load_state_dict(net.state_dict())
            dmb = DDPM(netb, channels=4, image_size=R, sampling_timesteps=S, w=0.0).to(dev)
            ctxb = netb.diffusion_model.hip_context(dev, Bc)
            cb = cond.expand(Bc, -1, -1).contiguous()
            ib = image_cond.expand(Bc, -1, -1).contiguous()
            nsb = 20
            nb = torch.randn(nsb, Bc, 4, L, generator=g, device=dev)

            def runb(nsteps):
                xb = torch.randn(Bc, 4, L, generator=g, device=dev)
                stp, _ = cycled_steps(dmb, nsteps)
                _lib.check(lib.mtv_ddim_sample(ctxb, xb.data_ptr(), cb.data_ptr(), ib.data_ptr(), R * R, nb.data_ptr(), nsb,
                                               stp, nsteps, Bc, C.c_void_p(stream.cuda_stream)), "mtv_ddim_sample")

            runb(3)
            torch.cuda.synchronize(dev)
            tb = time.perf_counter()
            runb(nsb)
            torch.cuda.synchronize(dev)
            tb = time.perf_counter() - tb
            pb = netb.diffusion_model.profile_forward(Bc, 3, dev, step=True)
            cms = sum(p["ms"] for p in pb if p["name"].startswith("conv"))
            cfl = sum(p["flops"] for p in pb if p["name"].startswith("conv"))
            ams = sum(p["ms"] for p in pb if p["name"].startswith("attn"))
            afl = sum(p["flops"] for p in pb if p["name"].startswith("attn"))
            batched = dict(clips_per_gpu=Bc, steps=nsb, clip_steps_per_s=round(Bc * nsb / tb, 2), ms_per_batched_step=round(1e3 * tb / nsb, 3),
                           k_conv_tflops=round(cfl / cms / 1e9, 2), k_conv_frac_of_f32_mfma_peak=round(cfl / cms / 1e9 / MFMA_F32_PEAK_TF, 3),
                           k_attention_tflops=round(afl / ams / 1e9, 2),
                           k_attention_frac_of_f32_mfma_peak=round(afl / ams / 1e9 / MFMA_F32_PEAK_TF, 3))
            del netb, dmb
            gpu_sections["batched_info"] = time.perf_counter() - t_sec
        # ---- informational only: BASELINE configs[3] (16-frame 512x512 clip: R = 64, L = 6144 tokens, B = 1) in the same line, so that the
        # driver's default run carries a number for it (`python bench.py --res 64` makes it the line's own workload); never `value`
        res64 = None
        if world == 1 and R == 32 and not args.no_res64:
            t_sec = time.perf_counter()
            R6 = 64
            L6 = R6 * R6 + 2 * T * R6
            net6 = DiffusionWrapper(UNetModel(**dict(BASE_UNET_CONFIG, image_size=R6), frames=T, max_batch=1)).eval().to(dev)
            net6.load_state_dict(net.state_dict())          # (no parameter of the UNet depends on the latent resolution)
            dm6 = DDPM(net6, channels=4, image_size=R6, sampling_timesteps=S, w=0.0).to(dev)
            ctx6 = net6.diffusion_model.hip_context(dev, 1)
            c6 = torch.rand(1, 8, L6, generator=g, device=dev) * 2 - 1
            i6 = torch.rand(1, 4, R6 * R6, generator=g, device=dev) * 2 - 1
            ns6 = 40
            n6 = torch.randn(ns6, 1, 4, L6, generator=g, device=dev)

            def run6(nsteps):
                x6 = torch.randn(1, 4, L6, generator=g, device=dev)
                stp, _ = cycled_steps(dm6, nsteps)
                _lib.check(lib.mtv_ddim_sample(ctx6, x6.data_ptr(), c6.data_ptr(), i6.data_ptr(), R6 * R6, n6.data_ptr(), ns6,
                                               stp, nsteps, 1, C.c_void_p(stream.cuda_stream)), "mtv_ddim_sample")

            run6(ns6)                                        # plan, graphs, clocks
            torch.cuda.synchronize(dev)
            t6 = time.perf_counter()
            run6(ns6)
            torch.cuda.synchronize(dev)
            t6 = time.perf_counter() - t6
            p6 = net6.diffusion_model.profile_forward(1, 3, dev, step=True)
            cms = sum(p["ms"] for p in p6 if p["name"].startswith("conv"))
            cfl = sum(p["flops"] for p in p6 if p["name"].startswith("conv"))
            ams = sum(p["ms"] for p in p6 if p["name"].startswith("attn"))
            afl = sum(p["flops"] for p in p6 if p["name"].startswith("attn"))
            ev6 = max(0.0, (sum(p["ms"] for p in p6) - 1e3 * t6 / ns6) / max(1, len(p6)))      # event overhead per launch, as above
            cms = max(1e-6, cms - ev6 * sum(1 for p in p6 if p["name"].startswith("conv")))
            ams = max(1e-6, ams - ev6 * sum(1 for p in p6 if p["name"].startswith("attn")))
            res64 = dict(workload=f"configs[3]: 16-frame 512x512 clip = tri-plane latent [1,4,{L6}] (R=64,T=16), same UNet, B=1",
                         steps=ns6, steps_per_s=round(ns6 / t6, 2), ms_per_step=round(1e3 * t6 / ns6, 3), launches_per_step=len(p6),
                         conv_tflops=round(cfl / cms / 1e9, 2), conv_frac_of_f32_mfma_peak=round(cfl / cms / 1e9 / MFMA_F32_PEAK_TF, 3),
                         attention_tflops=round(afl / ams / 1e9, 2), attention_frac_of_f32_mfma_peak=round(afl / ams / 1e9 / MFMA_F32_PEAK_TF, 3),
                         note=f"{ns6} timed steps after {ns6} untimed; fractions from hipEvents around every launch of one step; "
                              "counters for this workload: `python bench.py --res 64` (roofline.traffic there)")
            del net6, dm6
            gpu_sections["res64_info"] = time.perf_counter() - t_sec
        # ---- informational only: the steps either side of the loop (BASELINE configs[4]'s decode tail, section 8 f-1/f-2),
        # one 16-frame 256x256 clip through the HIP autoencoder (recipe-filled weights; never part of `value`)
        ae_info = None
        if world == 1 and not args.no_autoencoder and R == 32:
            from moditalker_amd import BASE_AE_DDCONFIG, ViTAutoencoder
            from moditalker_amd import filler
            t_sec = time.perf_counter()
            ae = ViTAutoencoder(4, BASE_AE_DDCONFIG).eval()
            filler.fill_autoencoder_(ae, seed=77)       # arithmetic recipe; output layers kept out of saturation
            ae = ae.to(dev)
            zlat = xt.clone()
            vid = torch.rand(1, 3, 16, 256, 256, generator=g, device=dev) * 2 - 1

            def timed(fn, n=5):
                fn()
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize(dev)
                return (time.perf_counter() - t1) / n

            t_dec = timed(lambda: ae.decode_from_sample(zlat))
            t_ext = timed(lambda: ae.extract(vid))
            pd = ae.profile(1, False, 2)
            gfl = sum(p["flops"] for p in pd if p["name"].startswith("gemm"))
            gms = sum(p["ms"] for p in pd if p["name"].startswith("gemm"))
            ae_info = dict(decode_from_sample_ms=round(1e3 * t_dec, 3), extract_ms=round(1e3 * t_ext, 3), launches_decode=len(pd),
                           decode_tflop=round(sum(p["flops"] for p in pd) / 1e12, 3),
                           gemm_tflops=round(gfl / gms / 1e9, 1) if gms else None,
                           gemm_frac_of_f32_mfma_peak=round(gfl / gms / 1e9 / MFMA_F32_PEAK_TF, 3) if gms else None,
                           clip_end_to_end_ms_at_250_steps=round(250 * 1e3 * dt / K + 1e3 * (t_dec + 4 * t_ext), 1),
                           note="4 extracts + 250 steps + decode per 16-frame clip (sample.py:328-386); weights random-init")
            del ae
            gpu_sections["autoencoder_info"] = time.perf_counter() - t_sec
        attn_roof = dict(bound="mfma", kernel="k_attention", achieved=round(attn["flops"] / (attn["ms"] * 1e-3) / 1e12, 3) if attn["ms"] else 0.0,
                         peak=MFMA_F32_PEAK_TF, unit="TFLOP/s", launches_per_step=attn["launches"], flops_per_step=attn["flops"])
        attn_roof["frac"] = round(attn_roof["achieved"] / MFMA_F32_PEAK_TF, 4)
        attn_roof["mfma_util_counter"] = pmc.get("k_attention", {}).get("mfma_util")   # matrix-pipe busy fraction from SQ counters (committed pass)
        attn_roof["counter_kind"] = "committed" if attn_roof["mfma_util_counter"] is not None else None
        result = {
            "metric": f"denoise-steps/sec (16-frame {8 * R}^2 clip, 250 DDIM steps)",
            "value": round(world * K / dt, 3),
            "unit": "denoise-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ramp_steps_untimed": args.ramp_steps,
            "steps_sampled": f"the first {K} steps of the 250-step DDIM schedule (the network's work does not depend on the step index)"
                             if K <= 250 else f"{K} steps cycling through the 250-step DDIM schedule",
            "ms_per_step": round(1e3 * dt / K, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            # fp32 in, fp32 out, fp32-class products everywhere: v_mfma_f32_16x16x4_f32 (exact) except the QK^T of the 12 level-0 attention
            # launches, taken as six bf16 partial products of a three-term split of q and k (24 mantissa bits, f32 accumulation) --
            # same parity bars as the exact path (tests/test_gpu_parity.py: eps <= 2e-4, the 250-step sample <= 1e-3 vs the reference golden)
            "precision_note": "fp32-class throughout; QK^T of the 2048-token attentions = 3-term bf16 split, 6 products (~2^-24 relative)",
            "data": "synthetic (random-init weights incl. zero-init tensors, U(-1,1) cond latents, N(0,1) noise)",
            "distributed": dist_info,
            "cpu_affinity": {k: v for k, v in affinity.items() if not k.startswith("_")},
            "config": {"workload": f"configs[{1 if R == 32 else 3}]: 16-frame {8 * R}x{8 * R} clip = tri-plane latent [1,4,{L}] (R={R},T=16), "
                                   "base second-stage UNet (132.2M live params), DDIM eta=1, 250-step schedule, B=1 per GPU",
                       "clips": world, "parallelism": f"clip-sharded x{world}, all_gather of latents at the end"},
            "roofline": roofline,
            # wall seconds of the sections of this process in which the GPU works (a utilisation sampler with a 1 Hz cadence sees little of
            # them next to the CPU-baseline leg, which keeps the GPU idle for tens of seconds)
            "gpu_active_s": {"total": round(sum(gpu_sections.values()), 3), **{k: round(v, 3) for k, v in gpu_sections.items()}},
            "cpu_baseline": cpu,
            "step_ms_sum_of_launches": round(step_ms_events, 4),
            "launches_per_step": work["n_launches_step"],
            "flops_per_step": {k: work[k] for k in ("flops_conv3x3", "flops_1x1", "flops_attn_core", "flops_linear")},
            "families": families,
            "roofline_attention": attn_roof,
            "batched_info": batched,
            "res64_info": res64,
            "autoencoder_info": ae_info,
        }
    barrier()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: RCCL writes its version banner through C stdio, buffered until exit
        try:
            C.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
