#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Runs only in the build container (needs /root/reference; never on the GPU box).  It imports
MToV/models/ddpm/unet.py and MToV/losses/ddpm.py unmodified, with the two harness-side shims
SURVEY.md section 8c documents:
  1. unet.py:1024 hard-codes `.to("cuda")`            -> map "cuda*" strings to "cpu"
  2. ddpm.py:19,25-27,31 import torchvision / cv2      -> empty stub modules (unused on this path)
Weights, inputs and noise come from the arithmetic recipe in moditalker_amd/filler.py (every
tensor, including the zero-initialised ones), noise is injected by patching torch.randn /
torch.randn_like while DDPM.sample runs (draw order: initial, then one per non-final step).

It also cross-checks oracle/ref_unet.py + oracle/ref_ddpm.py against the reference on every
case and prints the max-abs differences (the oracle's pin).

Usage:  python tests/golden/make_golden.py [--quick]
Outputs (fp32, compressed .npz): tests/golden/*.npz  -- data only, no reference source.
"""
import argparse
import contextlib
import importlib.machinery
import io
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference/MToV"

from moditalker_amd import filler  # noqa: E402
from oracle import ref_ddpm, ref_unet  # noqa: E402


def import_reference():
    sys.path.insert(0, REF)

    def stub(name, **a):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__path__ = []
        m.__dict__.update(a)
        sys.modules[name] = m
        return m

    tv = stub("torchvision")
    tv.utils = stub("torchvision.utils", make_grid=None)
    tv.transforms = stub("torchvision.transforms", ToTensor=object, ToPILImage=object)
    stub("cv2")
    _to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: _to(
        self, *[("cpu" if isinstance(x, str) and x.startswith("cuda") else x) for x in a], **k)
    from models.ddpm.unet import UNetModel, DiffusionWrapper
    import losses.ddpm as D
    D.tqdm = lambda it, **k: it
    return UNetModel, DiffusionWrapper, D


@contextlib.contextmanager
def injected_noise(noise):
    """torch.randn / randn_like pop from `noise` in call order."""
    q = list(noise)
    r0, r1, ms = torch.randn, torch.randn_like, torch.manual_seed
    seeds = []

    def _randn(*shape, **k):
        t = q.pop(0)
        shp = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        assert tuple(t.shape) == shp, (t.shape, shp)
        return t.clone()

    def _randn_like(x, **k):
        t = q.pop(0)
        assert t.shape == x.shape
        return t.clone()

    def _seed(s):
        seeds.append(s)
        return ms(s)

    torch.randn, torch.randn_like, torch.manual_seed = _randn, _randn_like, _seed
    try:
        yield q, seeds
    finally:
        torch.randn, torch.randn_like, torch.manual_seed = r0, r1, ms


NARROW_CFG = dict(image_size=32, in_channels=4, out_channels=4, model_channels=32,
                  attention_resolutions=[1, 2], num_res_blocks=1, channel_mult=[1, 2],
                  num_heads=2, use_scale_shift_norm=True, resblock_updown=True, cond_model=False)
# 3-level UNet of BASELINE config 1 (R=8,T=4 needs <=3 levels); run by the reference at (32,16)
SHALLOW_CFG = dict(image_size=32, in_channels=4, out_channels=4, model_channels=32,
                   attention_resolutions=[4, 2, 1], num_res_blocks=2, channel_mult=[1, 2, 4],
                   num_heads=8, use_scale_shift_norm=True, resblock_updown=True, cond_model=False)


def build(UNetModel, DiffusionWrapper, cfg, seed):
    with contextlib.redirect_stdout(io.StringIO()):
        net = DiffusionWrapper(UNetModel(**cfg)).eval()
    filler.fill_module_(net, seed=seed)
    return net


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="skip the long base-config sampling runs")
    ap.add_argument("--add-base-run", default=None, metavar="S[:ratio]",
                    help="only run one more base-config sampler case (e.g. 250) and ADD it to the existing base.npz")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    UNetModel, DiffusionWrapper, D = import_reference()
    R, T = 32, 16
    L = R * R + 2 * T * R
    report = []

    def check(name, a, b, tol):
        d = float((a - b).abs().max())
        report.append((name, d))
        print(f"  oracle vs reference  {name:38s} max-abs {d:.3e}")
        assert d <= tol, (name, d)

    # ---------------------------------------------------------------- schedule / time pairs
    with contextlib.redirect_stdout(io.StringIO()):
        dm0 = D.DDPM(torch.nn.Identity(), channels=4, image_size=32, sampling_timesteps=50, w=0.0)
    buf = ref_ddpm.schedule_buffers()
    sched = {}
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
              "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod"):
        ref = getattr(dm0, k)
        assert torch.equal(ref, buf[k]), k
        sched[k] = ref.numpy()
    for S in (4, 50, 100, 250):
        times = torch.linspace(-1, 999, steps=S + 1)
        times = list(reversed(times.int().tolist()))
        pairs = list(zip(times[:-1], times[1:]))
        assert pairs == ref_ddpm.ddim_time_pairs(1000, S)
        sched[f"times_S{S}"] = np.array(times, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "schedule.npz"), **sched)
    print("schedule.npz written (oracle schedule bit-equal to reference)")

    # ---------------------------------------------------------------- narrow + shallow models: forward, taps, sampling
    for tag, cfg, seed in (() if args.add_base_run else (("narrow", NARROW_CFG, 11), ("shallow", SHALLOW_CFG, 12))):
        net = build(UNetModel, DiffusionWrapper, cfg, seed)
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        B = 2
        x, cond, ic = filler.synthetic_inputs(B, R, T, seed=seed, tag=tag)
        t = torch.tensor([977, 13], dtype=torch.long)
        # reference taps via forward hooks on the cross-plane attention blocks
        taps_ref = {}
        um = net.diffusion_model
        hooks = []
        for i, m in enumerate(um.input_attns):
            hooks.append(m.register_forward_hook(lambda mod, inp, out, i=i: taps_ref.__setitem__(f"in{i}", out.detach().clone())))
        hooks.append(um.mid_attn.register_forward_hook(lambda mod, inp, out: taps_ref.__setitem__("mid", out.detach().clone())))
        for i, m in enumerate(um.output_attns):
            hooks.append(m.register_forward_hook(lambda mod, inp, out, i=i: taps_ref.__setitem__(f"out{i}", out.detach().clone())))
        with torch.no_grad():
            eps = net(x, cond, ic, t)
        for h in hooks:
            h.remove()
        taps = {}
        eps_o = ref_unet.unet_forward(sd, cfg, x, cond, ic, t, R, T, taps=taps)
        check(f"{tag} forward eps", eps_o, eps, 2e-5)
        for k in taps_ref:
            check(f"{tag} tap {k}", taps[k], taps_ref[k], 2e-5)
        out = dict(eps=eps.numpy(), t=t.numpy(), seed=np.int64(seed), batch=np.int64(B))
        for k, v in taps_ref.items():
            out["tap_" + k] = v[..., ::7].contiguous().numpy()
        # sampling: plain DDIM S=8 and noised-start S=20 ratio .25 (5 steps), B=2
        for S, ratio, fix in ((8, None, False), (20, 0.25, False), (20, 0.25, True)):
            with contextlib.redirect_stdout(io.StringIO()):
                dm = D.DDPM(net, channels=4, image_size=32, sampling_timesteps=S, w=0.0)
            n = ref_ddpm.num_noise_draws(S, ratio)
            noise = filler.noise_list(n, (B, 4, L), seed=seed, tag=f"{tag}.S{S}")
            ns = filler.uniform_pm1(f"{tag}.noised_start", (B, 4, L), seed) if ratio else None
            with injected_noise(noise) as (q, seeds), contextlib.redirect_stdout(io.StringIO()):
                z = dm.sample(batch_size=B, cond=cond, image_cond=ic, noised_start=ns, ratio_=ratio, fix_noise=fix)
            assert len(q) == 0, "noise draw count mismatch"
            assert seeds == ([1004] if fix else []), seeds
            zo = ref_ddpm.ddim_sample(lambda a, b, c, d: ref_unet.unet_forward(sd, cfg, a, b, c, d, R, T),
                                      cond, ic, noise, S, noised_start=ns, ratio_=ratio)
            nm = f"sample_S{S}" + (f"_r{ratio}" if ratio else "") + ("_fix" if fix else "")
            check(f"{tag} {nm}", zo, z, 1e-4)
            out[nm] = z.numpy()
        np.savez_compressed(os.path.join(HERE, f"{tag}.npz"), **out)
        print(f"{tag}.npz written")

    # ---------------------------------------------------------------- base config
    cfg = ref_unet.BASE_CFG
    seed = 7
    t0 = time.time()
    net = build(UNetModel, DiffusionWrapper, cfg, seed)
    sd = {k: v for k, v in net.state_dict().items()}
    assert len(sd) == 804, len(sd)
    print(f"base model built+filled in {time.time() - t0:.1f}s ({sum(v.numel() for v in sd.values()) / 1e6:.1f} M params)")
    x, cond, ic = filler.synthetic_inputs(1, R, T, seed=seed, tag="base")
    out = dict(seed=np.int64(seed))
    for tv in (999, 500, 0):
        t = torch.tensor([tv], dtype=torch.long)
        with torch.no_grad():
            eps = net(x, cond, ic, t)
        eps_o = ref_unet.unet_forward(sd, cfg, x, cond, ic, t, R, T)
        check(f"base forward t={tv}", eps_o, eps, 2e-5)
        out[f"eps_t{tv}"] = eps.numpy()
    # key/shape manifest of the 804-key checkpoint layout (names + shapes are data, not code)
    out["keys"] = np.array(list(sd.keys()))
    out["shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
    # S=250 is the schedule BASELINE.json's metric is quoted on (ddpm.py:371-375 with sampling_timesteps=250)
    runs = [(4, None, False)] if args.quick else [(4, None, False), (50, None, False), (100, 0.25, False), (250, None, False)]
    if args.add_base_run:
        old = dict(np.load(os.path.join(HERE, "base.npz")))
        for k in ("eps_t999", "eps_t500", "eps_t0"):
            assert np.array_equal(old[k], out[k]), f"existing base.npz disagrees on {k}: regenerate everything"
        out = old
        sr = args.add_base_run.split(":")
        runs = [(int(sr[0]), float(sr[1]) if len(sr) > 1 else None, False)]
    for S, ratio, fix in runs:
        with contextlib.redirect_stdout(io.StringIO()):
            dm = D.DDPM(net, channels=4, image_size=32, sampling_timesteps=S, w=0.0)
        n = ref_ddpm.num_noise_draws(S, ratio)
        noise = filler.noise_list(n, (1, 4, L), seed=seed, tag=f"base.S{S}")
        ns = filler.uniform_pm1("base.noised_start", (1, 4, L), seed) if ratio else None
        t0 = time.time()
        with injected_noise(noise) as (q, seeds), contextlib.redirect_stdout(io.StringIO()):
            z = dm.sample(batch_size=1, cond=cond, image_cond=ic, noised_start=ns, ratio_=ratio, fix_noise=fix)
        dt = time.time() - t0
        assert len(q) == 0
        nm = f"sample_S{S}" + (f"_r{ratio}" if ratio else "")
        print(f"  reference {nm}: {dt:.1f}s ({len(noise)} UNet steps, {len(noise) / dt:.2f} steps/s, {torch.get_num_threads()} threads)")
        out[nm] = z.numpy()
        if S == 4:
            zo = ref_ddpm.ddim_sample(lambda a, b, c, d: ref_unet.unet_forward(sd, cfg, a, b, c, d, R, T),
                                      cond, ic, noise, S)
            check(f"base {nm}", zo, z, 1e-4)
    np.savez_compressed(os.path.join(HERE, "base.npz"), **out)
    print("base.npz written")
    with open(os.path.join(HERE, "PIN_REPORT.txt"), "a" if args.add_base_run else "w") as f:
        if args.add_base_run:
            f.write(f"# --add-base-run {args.add_base_run}: reference sample added to base.npz; forwards re-checked bit-equal\n")
            for k, d in report:
                f.write(f"{k:44s} {d:.3e}\n")
            return
        f.write("oracle (oracle/ref_unet.py, oracle/ref_ddpm.py) vs imported reference, max-abs, fp32 CPU\n")
        f.write(f"torch {torch.__version__}, {torch.get_num_threads()} threads\n")
        for k, d in report:
            f.write(f"{k:44s} {d:.3e}\n")


if __name__ == "__main__":
    main()
