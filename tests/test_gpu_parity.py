"""GPU parity: the HIP path (through the C ABI, via the reference-shaped Python classes) against
(a) the golden vectors captured from the reference itself and (b) the CPU oracle on the same seeded
inputs.  Tolerances: fp32, 1e-3 max-abs is the north-star bar; the per-forward checks use a much
tighter 2e-4 so a real defect cannot hide under the sampler's tolerance."""
import os

import numpy as np
import pytest
import torch

import copy
import functools

from conftest import BASE_CFG, GOLDEN, NARROW_CFG, SHALLOW_CFG, report
from moditalker_amd import DDPM, DiffusionWrapper, UNetModel, filler

pytestmark = pytest.mark.gpu

FWD_TOL = 2e-4
SAMPLE_TOL = 1e-3     # BASELINE.json north_star: <= 1e-3 max-abs in fp32


def _dev():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch.device("cuda:0")


# The recipe-filled parameter values of a (config, seed) are computed ONCE per session (the arithmetic filler walks 132 M scalars for
# the base net); every test still gets its OWN module and its own library context -- the mtv_debug_force_* knobs act on plans built
# after the call, so a context must never be shared across tests.
_FILLED = {}


def _build(cfg, seed, frames=16, max_batch=2, **kw):
    net = DiffusionWrapper(UNetModel(**cfg, frames=frames, max_batch=max_batch, **kw)).eval()
    key = (repr(sorted(cfg.items())), seed, frames, repr(sorted(kw.items())))
    if key not in _FILLED:
        filler.fill_module_(net, seed=seed, skip_prefixes=("output_bg_",))
        _FILLED[key] = {k: v.detach().clone() for k, v in net.state_dict().items()}
    else:
        net.load_state_dict(_FILLED[key])
    return net.to(_dev())


def _maxabs(a, b):
    return float((a.detach().cpu().float() - torch.as_tensor(b).float()).abs().max())


@functools.lru_cache(maxsize=None)
def _golden(tag):
    return dict(np.load(os.path.join(GOLDEN, f"{tag}.npz")))


RAGGED_CFG = dict(BASE_CFG, image_size=24)            # 24x24 | 8x24 | 8x24 planes: ragged tiles at every level, segments of 576 / 192 keys ... down to 9 / 3


@functools.lru_cache(maxsize=None)
def _ragged_case(B):
    """The ragged geometry every forced-kernel test also runs: base UNet at (R, T) = (24, 8), B clips, against the CPU oracle -- inputs and the
    oracle's output computed once per session (the oracle forward is seconds of CPU time; the tests differ in the KERNELS, not the inputs)."""
    from oracle import ref_unet
    net = _build(RAGGED_CFG, 21, frames=8, max_batch=B)
    x, cond, ic = filler.synthetic_inputs(B, 24, 8, seed=5, tag="ragged")
    t = torch.tensor([700, 3, 999, 250][:B])
    ref = ref_unet.unet_forward({k: v.cpu() for k, v in net.state_dict().items()}, RAGGED_CFG, x, cond, ic, t, 24, 8)
    return x, cond, ic, t, ref


def _check_ragged(B, what):
    x, cond, ic, t, ref = _ragged_case(B)
    dev = _dev()
    net2 = _build(RAGGED_CFG, 21, frames=8, max_batch=B)
    e = report(f"{what}: ragged (24,8) B={B} eps vs oracle", _maxabs(net2(x.to(dev), cond.to(dev), ic.to(dev), t.to(dev)), ref), FWD_TOL)
    assert e <= FWD_TOL, (what, e)


def _check_base_eps(net, what, tvs=(999, 0)):
    g = _golden("base")
    dev = _dev()
    x, cond, ic = filler.synthetic_inputs(1, 32, 16, seed=7, tag="base")
    for tv in tvs:
        e = report(f"{what}: base eps t={tv} vs reference golden", _maxabs(net(x.to(dev), cond.to(dev), ic.to(dev), torch.tensor([tv], device=dev)), g[f"eps_t{tv}"]), FWD_TOL)
        assert e <= FWD_TOL, (what, tv, e)
    return x, cond, ic


def _check_base_sample(net, what, S, inputs=None):
    g = _golden("base")
    dev = _dev()
    x, cond, ic = inputs or filler.synthetic_inputs(1, 32, 16, seed=7, tag="base")
    noise = [z.to(dev) for z in filler.noise_list(S, (1, 4, 2048), seed=7, tag=f"base.S{S}")]
    dm = DDPM(net, channels=4, image_size=32, sampling_timesteps=S, w=0.0).to(dev)
    e = report(f"{what}: base {S}-step sample vs reference golden", _maxabs(dm.sample(batch_size=1, cond=cond.to(dev), image_cond=ic.to(dev), noise=noise), g[f"sample_S{S}"]), SAMPLE_TOL)
    assert e <= SAMPLE_TOL, (what, S, e)


@pytest.mark.parametrize("tag,cfg,seed", [("narrow", NARROW_CFG, 11), ("shallow", SHALLOW_CFG, 12)])
def test_forward_and_taps_vs_reference_golden(tag, cfg, seed):
    g = np.load(os.path.join(GOLDEN, f"{tag}.npz"))
    net = _build(cfg, seed)
    B = int(g["batch"])
    x, cond, ic = filler.synthetic_inputs(B, 32, 16, seed=seed, tag=tag)
    t = torch.from_numpy(g["t"])
    dev = _dev()
    eps = net(x.to(dev), cond.to(dev), ic.to(dev), t.to(dev))
    um = net.diffusion_model
    worst = {}
    for k in g.files:
        if k.startswith("tap_"):
            tap = um.debug_tap(k[4:], B)
            worst[k] = _maxabs(tap[..., ::7], g[k])
    bad = {k: v for k, v in worst.items() if not v <= FWD_TOL}
    report(f"{tag}: worst of {len(worst)} taps vs reference golden", max(worst.values()), FWD_TOL)
    assert not bad, f"taps off: {bad} (all: {worst})"
    assert report(f"{tag}: eps vs reference golden", _maxabs(eps, g["eps"]), FWD_TOL) <= FWD_TOL


@pytest.mark.parametrize("tag,cfg,seed", [("narrow", NARROW_CFG, 11), ("shallow", SHALLOW_CFG, 12)])
@pytest.mark.parametrize("S,ratio,fix", [(8, None, False), (20, 0.25, False), (20, 0.25, True)])
def test_sampler_vs_reference_golden(tag, cfg, seed, S, ratio, fix):
    g = np.load(os.path.join(GOLDEN, f"{tag}.npz"))
    net = _build(cfg, seed)
    dev = _dev()
    B, L = 2, 2048
    x, cond, ic = filler.synthetic_inputs(B, 32, 16, seed=seed, tag=tag)
    dm = DDPM(net, channels=4, image_size=32, sampling_timesteps=S, w=0.0).to(dev)
    n = S if ratio is None else S - int(S * (1 - ratio))
    noise = [z.to(dev) for z in filler.noise_list(n, (B, 4, L), seed=seed, tag=f"{tag}.S{S}")]
    ns = filler.uniform_pm1(f"{tag}.noised_start", (B, 4, L), seed).to(dev) if ratio else None
    z = dm.sample(batch_size=B, cond=cond.to(dev), image_cond=ic.to(dev), noised_start=ns, ratio_=ratio,
                  fix_noise=fix, noise=noise)
    nm = f"sample_S{S}" + (f"_r{ratio}" if ratio else "") + ("_fix" if fix else "")
    assert z.shape == (B, 4, L) and z.device.type == "cuda"
    assert float(z.abs().max()) <= 1.0          # clamped x0 of the last step (ddpm.py:346-351,386-388)
    assert report(f"{tag}: {nm} vs reference golden", _maxabs(z, g[nm]), SAMPLE_TOL) <= SAMPLE_TOL


def test_base_forward_vs_reference_golden():
    g = np.load(os.path.join(GOLDEN, "base.npz"))
    net = _build(BASE_CFG, 7, max_batch=1)
    dev = _dev()
    x, cond, ic = filler.synthetic_inputs(1, 32, 16, seed=7, tag="base")
    for tv in (999, 500, 0):
        eps = net(x.to(dev), cond.to(dev), ic.to(dev), torch.tensor([tv], device=dev))
        assert report(f"default plan: base eps t={tv} vs reference golden", _maxabs(eps, g[f"eps_t{tv}"]), FWD_TOL) <= FWD_TOL, tv


# S=250 is the schedule BASELINE.json's metric is quoted on (ddpm.py:371-375 with sampling_timesteps=250)
@pytest.mark.parametrize("S,ratio", [(4, None), (50, None), (100, 0.25), (250, None)])
def test_base_sampler_vs_reference_golden(S, ratio):
    g = np.load(os.path.join(GOLDEN, "base.npz"))
    net = _build(BASE_CFG, 7, max_batch=1)
    dev = _dev()
    L = 2048
    x, cond, ic = filler.synthetic_inputs(1, 32, 16, seed=7, tag="base")
    dm = DDPM(net, channels=4, image_size=32, sampling_timesteps=S, w=0.0).to(dev)
    n = S if ratio is None else S - int(S * (1 - ratio))
    noise = [z.to(dev) for z in filler.noise_list(n, (1, 4, L), seed=7, tag=f"base.S{S}")]
    ns = filler.uniform_pm1("base.noised_start", (1, 4, L), 7).to(dev) if ratio else None
    z = dm.sample(batch_size=1, cond=cond.to(dev), image_cond=ic.to(dev), noised_start=ns, ratio_=ratio, noise=noise)
    nm = f"sample_S{S}" + (f"_r{ratio}" if ratio else "")
    assert report(f"default plan: base {nm} vs reference golden", _maxabs(z, g[nm]), SAMPLE_TOL) <= SAMPLE_TOL


def test_eager_equals_graph_and_is_deterministic():
    net = _build(NARROW_CFG, 11)
    dev = _dev()
    x, cond, ic = filler.synthetic_inputs(2, 32, 16, seed=3, tag="det")
    t = torch.tensor([400, 20], device=dev)
    a = net(x.to(dev), cond.to(dev), ic.to(dev), t)
    b = net(x.to(dev), cond.to(dev), ic.to(dev), t)
    net.diffusion_model.set_eager(True)
    c = net(x.to(dev), cond.to(dev), ic.to(dev), t)
    assert _maxabs(a, b.cpu()) <= 1e-6
    assert _maxabs(a, c.cpu()) <= 1e-6


def test_config3_512px_clip_vs_oracle():
    """BASELINE configs[3]: 16-frame 512x512 clip = (R,T)=(64,16), base UNet, L=6144 tokens.  One forward at a
    low timestep and a 2-step DDIM (t = 999, 499) against the oracle (the reference hard-wires R=32; an oracle forward at this size is
    ~7 s of host time)."""
    from oracle import ref_ddpm, ref_unet
    R, T, S = 64, 16, 2
    cfg = dict(BASE_CFG, image_size=R)
    net = _build(cfg, 9, frames=T, max_batch=1)
    dev = _dev()
    L = R * R + 2 * T * R
    x, cond, ic = filler.synthetic_inputs(1, R, T, seed=9, tag="c3")
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    for tv in (3,):       # (t = 999 is the first step of the sampler run below)
        t = torch.tensor([tv])
        ref = ref_unet.unet_forward(sd, cfg, x, cond, ic, t, R, T)
        eps = net(x.to(dev), cond.to(dev), ic.to(dev), t.to(dev))
        assert report(f"configs[3] R=64: eps t={tv} vs oracle", _maxabs(eps, ref), FWD_TOL) <= FWD_TOL, tv
    noise = filler.noise_list(S, (1, 4, L), seed=9, tag="c3.noise")
    dm = DDPM(net, channels=4, image_size=R, sampling_timesteps=S, w=0.0).to(dev)
    z = dm.sample(batch_size=1, cond=cond.to(dev), image_cond=ic.to(dev), noise=[n.to(dev) for n in noise])
    zr = ref_ddpm.ddim_sample(lambda a, b, c, d: ref_unet.unet_forward(sd, cfg, a, b, c, d, R, T), cond, ic, noise, S)
    assert report("configs[3] R=64: 2-step sample vs oracle", _maxabs(z, zr), SAMPLE_TOL) <= 1e-3


# ----------------------------------------------------------------------------------------------
# geometries the reference cannot execute (hard-wired 32/16): pinned by the oracle (validated
# against the reference at (32,16) by tests/golden/make_golden.py)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("R,T,cfg,B", [
    (8, 4, SHALLOW_CFG, 3),                                   # BASELINE config 1 geometry (4-frame 64x64)
    (16, 8, NARROW_CFG, 1),
    (8, 8, dict(SHALLOW_CFG, use_scale_shift_norm=False), 2), # the h + emb_out branch (unet.py:204-206)
    (24, 8, SHALLOW_CFG, 2),                                  # not powers of two: 60-token deepest level, ragged tiles
    # level 0 itself is a deep level (128 tokens, B <= 2): the last stage ends on a K-slice slab tensor that the head conv's
    # GroupNorm + conv must see as ONE plain tensor (head.in is materialised)
    (8, 4, NARROW_CFG, 1),
    (8, 4, dict(NARROW_CFG, model_channels=128, num_heads=8), 2),
    # 96 tokens at level 1 (8x8 | 2x8 | 2x8): k_deep_block's row groups must be a power of two -- two groups of 48 rows, not three of 32
    (16, 4, SHALLOW_CFG, 1),
    (16, 4, SHALLOW_CFG, 2),
])
def test_other_geometries_vs_oracle(R, T, cfg, B):
    from oracle import ref_unet
    cfg = dict(cfg, image_size=R)
    net = _build(cfg, 21, frames=T, max_batch=B)
    dev = _dev()
    x, cond, ic = filler.synthetic_inputs(B, R, T, seed=5, tag="geo")
    t = torch.tensor([999, 0, 37][:B])
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    ref = ref_unet.unet_forward(sd, cfg, x, cond, ic, t, R, T)
    eps = net(x.to(dev), cond.to(dev), ic.to(dev), t.to(dev))
    assert report(f"geometry (R,T)=({R},{T}) B={B} mc={cfg['model_channels']}: eps vs oracle", _maxabs(eps, ref), FWD_TOL) <= FWD_TOL


def test_config1_cpu_plumbing_case_matches_oracle_sampler():
    """BASELINE config 1: DDIM 50 steps, 4-frame 64x64 clip (R=8,T=4), 3-level UNet; oracle on CPU vs HIP.

    With random weights this tiny geometry (GroupNorm over 2x2 / 1x2 planes at the deepest level) is
    chaotic: the oracle against ITSELF with a different CPU thread count moves the 50-step result by
    5e-2 (measured, /tmp probe recorded in DESIGN.md), so an end-to-end comparison only measures fp32
    summation order.  Parity is therefore checked one step ahead along the oracle's trajectory: every
    one of the 50 steps (UNet forward at that t + DDIM update with that step's noise) must reproduce
    the oracle's next state to 2e-4."""
    from oracle import ref_ddpm, ref_unet
    R, T, S = 8, 4, 50
    cfg = dict(SHALLOW_CFG, image_size=R)
    net = _build(cfg, 31, frames=T, max_batch=1)
    dev = _dev()
    L = R * R + 2 * T * R
    x, cond, ic = filler.synthetic_inputs(1, R, T, seed=9, tag="cfg1")
    noise = filler.noise_list(S, (1, 4, L), seed=9, tag="cfg1.noise")
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    traj = []
    ref = ref_ddpm.ddim_sample(lambda a, b, c, d: ref_unet.unet_forward(sd, cfg, a, b, c, d, R, T), cond, ic, noise, S,
                               trajectory=traj)
    assert len(traj) == S + 1
    dm = DDPM(net, channels=4, image_size=R, sampling_timesteps=S, w=0.0).to(dev)
    pairs = dm._time_pairs()
    worst = 0.0
    for i in range(S):
        nz = [noise[i + 1].to(dev)] if i + 1 < S else []
        nxt = dm._run_ddim(traj[i].to(dev), cond.to(dev), ic.to(dev), pairs[i:i + 1], nz)
        worst = max(worst, _maxabs(nxt, traj[i + 1]))
    assert report("configs[0] (8,4): worst one-step-ahead of 50 vs oracle", worst, FWD_TOL) <= FWD_TOL, worst
    # and the full 50-step run executes end to end, stays clamped and lands near the oracle's sample
    z = dm.sample(batch_size=1, cond=cond.to(dev), image_cond=ic.to(dev), noise=[n.to(dev) for n in noise])
    assert z.shape == (1, 4, L) and float(z.abs().max()) <= 1.0
    assert float((z.cpu() - ref).abs().mean()) <= 0.05


def test_batch_elements_are_independent():
    """Clip sharding relies on it: a clip's result does not depend on what else is in the batch."""
    net = _build(NARROW_CFG, 11, max_batch=3)
    dev = _dev()
    x, cond, ic = filler.synthetic_inputs(3, 32, 16, seed=4, tag="ind")
    t = torch.tensor([10, 500, 900], device=dev)
    full = net(x.to(dev), cond.to(dev), ic.to(dev), t)
    one = net(x[1:2].to(dev), cond[1:2].to(dev), ic[1:2].to(dev), t[1:2])
    # different batch sizes may pick different split-K tilings -> fp32 summation-order noise only
    assert _maxabs(full[1:2], one.cpu()) <= 2e-5
    again = net(x[1:2].to(dev), cond[1:2].to(dev), ic[1:2].to(dev), t[1:2])
    assert _maxabs(again, one.cpu()) == 0.0


def test_repeated_forwards_are_bit_identical():
    """The cross-workgroup split-K completion (tickets + write-through slab) must never read a stale
    partial: 30 forwards of the base model on the same inputs are bit-equal (any staleness shows up
    as a run-to-run difference), and so are two DDIM runs."""
    net = _build(BASE_CFG, 7, max_batch=1)
    dev = _dev()
    x, cond, ic = filler.synthetic_inputs(1, 32, 16, seed=7, tag="base")
    t = torch.tensor([321], device=dev)
    ref = net(x.to(dev), cond.to(dev), ic.to(dev), t).clone()
    for _ in range(30):
        out = net(x.to(dev), cond.to(dev), ic.to(dev), t)
        assert torch.equal(out, ref)
    dm = DDPM(net, channels=4, image_size=32, sampling_timesteps=10, w=0.0).to(dev)
    noise = [z.to(dev) for z in filler.noise_list(10, (1, 4, 2048), seed=7, tag="rep")]
    za = dm.sample(batch_size=1, cond=cond.to(dev), image_cond=ic.to(dev), noise=noise)
    zb = dm.sample(batch_size=1, cond=cond.to(dev), image_cond=ic.to(dev), noise=noise)
    assert torch.equal(za, zb)


def test_image_cond_tail_is_ignored():
    """Only the first R*R tokens of image_cond are read (unet.py:1022-1025)."""
    net = _build(NARROW_CFG, 11)
    dev = _dev()
    x, cond, ic = filler.synthetic_inputs(1, 32, 16, seed=4, tag="tail")
    t = torch.tensor([123], device=dev)
    a = net(x.to(dev), cond.to(dev), ic.to(dev), t)
    ic_long = torch.cat([ic, torch.full((1, 4, 1024), 7.0)], dim=2)
    b = net(x.to(dev), cond.to(dev), ic_long.to(dev), t)
    assert _maxabs(a, b.cpu()) == 0.0


def test_errors_are_loud():
    from moditalker_amd import MtvError
    net = DiffusionWrapper(UNetModel(**NARROW_CFG)).eval()        # left on CPU
    x, cond, ic = filler.synthetic_inputs(1, 32, 16, seed=1, tag="err")
    with pytest.raises(MtvError):
        net(x, cond, ic, torch.tensor([1]))
    net = net.to(_dev())
    with pytest.raises(ValueError):
        net(x[:, :, :100].to(_dev()), cond.to(_dev()), ic.to(_dev()), torch.tensor([1], device=_dev()))


def test_sampler_shape_errors_are_loud():
    """DDPM.sample hands raw pointers to the C ABI: every shape the reference would reject in torch.cat / broadcasting
    must be rejected here before the call (_run_ddim had no checks)."""
    dev = _dev()
    net = _build(NARROW_CFG, 11)
    dm = DDPM(net, channels=4, image_size=32, sampling_timesteps=4, w=0.0).to(dev)
    x, cond, ic = [t.to(dev) for t in filler.synthetic_inputs(2, 32, 16, seed=1, tag="err2")]
    noise = [z.to(dev) for z in filler.noise_list(4, (2, 4, 2048), seed=1, tag="err2.n")]
    with pytest.raises(ValueError):
        dm.sample(batch_size=2, cond=cond[:1], image_cond=ic, noise=noise)             # cond batch != batch_size
    with pytest.raises(ValueError):
        dm.sample(batch_size=2, cond=cond, image_cond=ic[:, :, :100], noise=noise)     # image_cond shorter than R*R
    with pytest.raises((ValueError, RuntimeError)):    # q_sample's broadcast fails first, exactly as in the reference (ddpm.py:486-491)
        dm.sample(batch_size=2, cond=cond, image_cond=ic, noised_start=x[:, :, :1000], ratio_=0.5, noise=noise)
    with pytest.raises(ValueError):
        dm.sample(batch_size=2, cond=cond, image_cond=ic, noise=[noise[0]] + [n[:1] for n in noise[1:]])   # draw batch dim
    with pytest.raises(ValueError):
        dm.sample(batch_size=2, cond=cond, image_cond=ic, noise=noise[:2])             # too few draws
    z = dm.sample(batch_size=2, cond=cond, image_cond=ic, noise=noise)                 # and the valid call still runs
    assert z.shape == (2, 4, 2048)


def test_module_copies_and_data_writes():
    """copy.deepcopy / pickle after a forward (the reference deep-copies the model for its EMA twin, sample.py:226),
    and the documented invalidate_weights() for writes through .data (which do not bump the version counter)."""
    import copy
    import pickle
    dev = _dev()
    net = _build(NARROW_CFG, 11)
    x, cond, ic = [t.to(dev) for t in filler.synthetic_inputs(1, 32, 16, seed=2, tag="cp")]
    t = torch.tensor([77], device=dev)
    a = net(x, cond, ic, t)
    twin = copy.deepcopy(net)
    assert twin.diffusion_model._ctx is None
    # (a copy builds its own context; shapes the committed tile table does not list are re-tuned there and may
    # pick another split-K: fp32 summation-order noise only)
    assert _maxabs(twin(x, cond, ic, t), a.cpu()) <= 2e-5
    blob = pickle.dumps(net)
    assert _maxabs(pickle.loads(blob).to(dev)(x, cond, ic, t), a.cpu()) <= 2e-5
    um = net.diffusion_model
    um.out[2].weight.data.mul_(0.5)              # .data write: invisible to the fingerprint
    um.out[2].bias.data.mul_(0.5)
    um.invalidate_weights()
    b = net(x, cond, ic, t)
    assert _maxabs(b, (a * 0.5).cpu()) <= 1e-6   # the head conv is linear in (weight, bias)
    net.load_state_dict(twin.state_dict())       # load_state_dict invalidates by itself
    assert torch.equal(net(x, cond, ic, t), a)   # (same context, same plan: bit-equal)


def _forced(setter, args, off_args, what, expect, tvs=(999, 0), sample_S=None, ragged_B=2):
    """One forced-kernel case: plans built while the knob is set run every eligible launch on that kernel / tile.  eps of the base UNet vs
    the reference golden, the plan really names the kernel (`expect(names)`), optionally a sampler run vs the golden, and the ragged
    two-clip geometry vs the oracle."""
    from moditalker_amd import _lib
    lib = _lib.load()
    fn = getattr(lib, setter)
    _lib.check(fn(*args), setter)
    try:
        net = _build(BASE_CFG, 7, max_batch=1)
        inputs = _check_base_eps(net, what, tvs)
        names = [p["name"] for p in net.diffusion_model.profile_forward(1, 1, _dev())]
        expect(names)
        if sample_S:
            _check_base_sample(net, what, sample_S, inputs)
        del net
        _check_ragged(ragged_B, what)
    finally:
        fn(*off_args)


@pytest.mark.parametrize("wm,wn", [(2, 4), (4, 8)])
def test_lds_tiled_conv_kernel_vs_reference_golden(wm, wn):
    """k_conv_lds (the large-token-count kernel: operands staged in LDS, 2x2 waves per workgroup) forced onto every
    eligible conv of the base UNet: eps vs the reference golden, and a ragged 2-clip geometry vs the oracle.  (Unforced it is also what
    the 8-clip batched plan and the autoencoder's out-projections select: test_batched_eight_clip_plan / test_gpu_autoencoder.)"""
    def expect(names):
        assert sum(",32,1]" in n for n in names) > 100, "the LDS-tiled kernel was not selected"
    _forced("mtv_debug_force_lds", (wm, wn), (0, 0), f"k_conv_lds<{wm},{wn}>", expect)


@pytest.mark.parametrize("mt,nt,nwv", [(1, 4, 1), (2, 2, 4)])
def test_lean_1x1_kernel_vs_reference_golden(mt, nt, nwv):
    """k_lin (csrc/lin.hip: whole K per wave, weight in the checkpoint's [N][K] layout, epilogue from the accumulators; the tile table
    selects it for six proj_out shapes of the 8-clip batch and of R = 64) forced onto every eligible qkv / proj_out conv: base UNet eps vs
    the reference golden at t = 999 / 0 and a ragged 2-clip geometry (row tiles straddling plane boundaries, partial tiles) vs the oracle."""
    def expect(names):
        assert sum(f"t{mt},{nt},64,{nwv}]" in n for n in names) >= 60, "the lean kernel was not selected"
    _forced("mtv_debug_force_lin", (mt, nt, nwv), (0, 0, 0), f"k_lin<{mt},{nt},{nwv}>", expect)


def test_bf16_pipe_qk_attention_vs_reference_golden():
    """k_attention<..., QB = 1> (csrc/kernels.hip: QK^T on v_mfma_f32_16x16x32_bf16 through a three-term split of q and k --
    d = 16: two 16-wide products per instruction; PV unchanged on the f32 instruction) switched on for EVERY 8-wave attention
    shape of the base UNet (the default has it on for d = 16 / 32 only -- and that default is what test_base_sampler runs to 250 steps):
    eps vs the reference golden at t = 999 / 500 / 0, the 4-step sample, and a ragged 2-clip geometry (partial key blocks, partial
    query tiles) vs the oracle."""
    _forced("mtv_debug_attention_qb", (1,), (-1,), "k_attention QB=1 everywhere", lambda names: None, tvs=(999, 500, 0), sample_S=4)


def test_exact_f32_attention_vs_reference_golden():
    """... and switched OFF everywhere (QK^T on the exact f32 instruction, round 2's core): the same checks, so that both arithmetic
    forms of the default kernel stay pinned."""
    _forced("mtv_debug_attention_qb", (0,), (-1,), "k_attention QB=0 everywhere", lambda names: None, tvs=(999, 0), sample_S=4)


def test_batched_eight_clip_plan_vs_reference_golden():
    """The B = 8 plan bench.py's `batched_info` times (its own tiles: k_conv_lds on the large convs, small key blocks in the
    attention), untouched by any forcing: clips 0, 2, 4, 6 carry the reference golden's inputs and noise and must reproduce
    the reference's 4-step sample and eps; clips 1, 3, 5, 7 carry a different clip (other inputs, other noise), checked
    against the oracle -- so a clip read or written at the wrong batch offset cannot pass."""
    from oracle import ref_ddpm, ref_unet
    g = np.load(os.path.join(GOLDEN, "base.npz"))
    B, S, L = 8, 4, 2048
    net = _build(BASE_CFG, 7, max_batch=B)
    dev = _dev()
    sd = {k: v.cpu() for k, v in net.state_dict().items() if "output_bg_" not in k}
    xa, ca, ia = filler.synthetic_inputs(1, 32, 16, seed=7, tag="base")
    xb, cb, ib = filler.synthetic_inputs(1, 32, 16, seed=19, tag="b8.other")
    pick = lambda a, b: torch.cat([a if k % 2 == 0 else b for k in range(B)], dim=0)
    x, cond, ic = pick(xa, xb), pick(ca, cb), pick(ia, ib)
    t = torch.tensor([999 if k % 2 == 0 else 321 for k in range(B)])
    eps = net(x.to(dev), cond.to(dev), ic.to(dev), t.to(dev)).cpu()
    eps_b = ref_unet.unet_forward(sd, BASE_CFG, xb, cb, ib, torch.tensor([321]), 32, 16)
    for k in range(B):
        assert _maxabs(eps[k:k + 1], g["eps_t999"] if k % 2 == 0 else eps_b) <= FWD_TOL, k
    names = [p["name"] for p in net.diffusion_model.profile_forward(B, 1, dev)]
    assert any(",32,1]" in n for n in names), "the batched plan is expected to use the LDS-tiled conv kernel somewhere"
    na = filler.noise_list(S, (1, 4, L), seed=7, tag=f"base.S{S}")
    nb = filler.noise_list(S, (1, 4, L), seed=19, tag="b8.other.noise")
    noise = [pick(a, b).to(dev) for a, b in zip(na, nb)]
    dm = DDPM(net, channels=4, image_size=32, sampling_timesteps=S, w=0.0).to(dev)
    z = dm.sample(batch_size=B, cond=cond.to(dev), image_cond=ic.to(dev), noise=noise).cpu()
    zb = ref_ddpm.ddim_sample(lambda a, b, c, d: ref_unet.unet_forward(sd, BASE_CFG, a, b, c, d, 32, 16), cb, ib, nb, S)
    for k in range(B):
        assert _maxabs(z[k:k + 1], g[f"sample_S{S}"] if k % 2 == 0 else zb) <= SAMPLE_TOL, k


@pytest.mark.parametrize("mt,nt,ks", [(4, 2, 1), (8, 2, 1), (4, 4, 1), (2, 2, 4), (2, 1, 8)])
def test_split_bf16_lds_conv_kernel_vs_reference_golden(mt, nt, ks):
    """k_x3_prep + k_conv_x3 (csrc/conv_x3.hip: GroupNorm / SiLU / three-term bf16 split in one elementwise pass, then a
    gathering GEMM on v_mfma_f32_16x16x32_bf16 with both operands by LDS-DMA, six partial products, f32 accumulation) forced onto
    every eligible conv of the base UNet: eps vs the reference golden at t = 999 / 0, the 4-step sample, and a ragged 4-clip
    geometry (partial tiles, rows straddling planes, column tiles wider than N) vs the oracle -- the same bars as the exact-f32
    kernels.  (The remaining tiles run unforced wherever the committed table selects them: the 8-clip batch, the autoencoder's GEMMs.)"""
    def expect(names):
        assert sum(f"t{mt},{nt},48," in n for n in names) >= 25, "the split-bf16 kernel was not selected"
        assert ks == 1 or sum(f"t{mt},{nt},48,{ks}]" in n for n in names) >= 10, "no K slices"      # (the level-0 convs: rows >= 2048)
    _forced("mtv_debug_force_b3", (mt, nt, ks), (0, 0, 1), f"k_conv_x3<{mt},{nt}> ks{ks}", expect, sample_S=4, ragged_B=4)


# ----------------------------------------------------------------------------------------------
# deep levels (csrc/deep.hip): K-sliced convs whose consumers add the partial slabs, fused attention + proj_out
# ----------------------------------------------------------------------------------------------
def _base_eps_and_sample(lib, names_out=None, what="deep levels"):
    g = _golden("base")
    net = _build(BASE_CFG, 7, max_batch=1)
    dev = _dev()
    x, cond, ic = filler.synthetic_inputs(1, 32, 16, seed=7, tag="base")
    worst = 0.0
    for tv in (999, 500, 0):
        eps = net(x.to(dev), cond.to(dev), ic.to(dev), torch.tensor([tv], device=dev))
        worst = max(worst, _maxabs(eps, g[f"eps_t{tv}"]))
    if names_out is not None:
        names_out.extend(p["name"] for p in net.diffusion_model.profile_forward(1, 1, dev, step=True))
    S = 4
    dm = DDPM(net, channels=4, image_size=32, sampling_timesteps=S, w=0.0).to(dev)
    noise = [z.to(dev) for z in filler.noise_list(S, (1, 4, 2048), seed=7, tag=f"base.S{S}")]
    z = dm.sample(batch_size=1, cond=cond.to(dev), image_cond=ic.to(dev), noise=noise)
    report(f"{what}: base eps (worst of t=999/500/0) vs reference golden", worst, FWD_TOL)
    return worst, report(f"{what}: base 4-step sample vs reference golden", _maxabs(z, g[f"sample_S{S}"]), SAMPLE_TOL)


def test_deep_levels_are_on_by_default_and_off_keeps_the_k_conv_path_green():
    """Default plan of the base UNet at one clip: the convs of levels 2 / 3 (<= 128 tokens) run on k_deep_conv; of their 18 attention
    blocks the 8 where it measures faster (32 tokens, and [128 x 256]) are ONE launch of k_deep_block (csrc/block.hip: no finalize
    pass, no qkv launch), the other 10 keep round 4's finalize + qkv + fused k_deep_attn; MTV_DEEP_OPT_NO_BLOCK selects the three
    launches everywhere, MTV_BLOCK_MAX_L=128 the one-launch block everywhere; mtv_debug_deep(0) puts every conv back on k_conv.
    All against the reference golden."""
    from moditalker_amd import _lib
    lib = _lib.load()
    names = []
    e1, s1 = _base_eps_and_sample(lib, names)
    assert e1 <= FWD_TOL and s1 <= SAMPLE_TOL, (e1, s1)
    deep = [n for n in names if n.startswith("conv") and " d" in n.split("[")[-1]]
    blocks = [n for n in names if n.startswith("attn") and " blk " in n]
    fused = [n for n in names if n.startswith("attn") and "+proj" in n]
    assert len(deep) >= 28 and len(blocks) == 8 and len(fused) == 10 and len(names) <= 170, (len(deep), len(blocks), len(fused), len(names))
    _lib.check(lib.mtv_debug_deep_options(16), "mtv_debug_deep_options")
    try:
        names = []
        e2, s2 = _base_eps_and_sample(lib, names)
        assert e2 <= FWD_TOL and s2 <= SAMPLE_TOL, (e2, s2)
        assert len([n for n in names if n.startswith("attn") and "+proj" in n]) == 18 and not [n for n in names if " blk " in n]
    finally:
        lib.mtv_debug_deep_options(-1)
    _lib.check(lib.mtv_debug_deep(0), "mtv_debug_deep")
    try:
        names = []
        e0, s0 = _base_eps_and_sample(lib, names)
        assert e0 <= FWD_TOL and s0 <= SAMPLE_TOL, (e0, s0)
        assert not [n for n in names if "+proj" in n or " blk " in n or n.startswith("fin")], "deep kernels in a plan built with them switched off"
    finally:
        lib.mtv_debug_deep(-1)


@pytest.mark.parametrize("mask", [1, 2, 4, 8, 32, 64, 80])      # (16 alone: test_deep_levels_are_on_by_default...; each bit once + round 4's whole plan)
def test_deep_level_dataflow_variants_vs_reference_golden(mask):
    """include/mtv_hip.h MTV_DEEP_OPT_*: in-launch completion (slab + ticket) instead of finalize passes, K-sliced / un-sliced qkv on
    k_deep_conv, k_attention + proj conv instead of the fused kernel, the three-launch attention block everywhere (16), the one-launch
    k_deep_block everywhere (32: also the [128 x 512] blocks the default leaves on three launches), k_deep_finalize passes instead of the
    default completion by tagged granules inside the producing kernel (64; 80 = round 4's plan: 64 + 16) -- eps at three timesteps and a
    4-step sample vs the reference golden, and repeated forwards bit-equal (the arrival order of K slices / cluster workgroups must not
    matter, and no hand-off may ever deliver a stale granule)."""
    from moditalker_amd import _lib
    lib = _lib.load()
    _lib.check(lib.mtv_debug_deep_options(mask), "mtv_debug_deep_options")
    try:
        e, s = _base_eps_and_sample(lib, what=f"MTV_DEEP_OPT mask {mask}")
        assert e <= FWD_TOL and s <= SAMPLE_TOL, (mask, e, s)
        net = _build(SHALLOW_CFG, 12)
        dev = _dev()
        x, cond, ic = filler.synthetic_inputs(2, 32, 16, seed=12, tag="shallow")
        t = torch.tensor([321, 9], device=dev)
        a = net(x.to(dev), cond.to(dev), ic.to(dev), t)
        for _ in range(5):
            assert torch.equal(net(x.to(dev), cond.to(dev), ic.to(dev), t), a)
    finally:
        lib.mtv_debug_deep_options(-1)


def test_deep_kernels_against_cpu_conv_and_attention():
    """tools/ubench/deep_bench (built by __graft_entry__.build): k_deep_conv on 19 shapes (3x3 / 1x1, concatenated sources, fused skip
    conv, residuals in slabs, upsampled sources, both row groupings, ragged planes, two clips, the base model's full-size shapes)
    and k_deep_attn on 9 (1-D / per-plane, head dims 16 / 32 / 64, ragged) against plain CPU restatements in double; k_deep_block
    (csrc/block.hip, the whole attention block in one launch) on 13 shapes -- 1 / 2 clips, 1-8 input slabs, head dims 16 / 32 / 64,
    GroupNorm groups of 1 - 16 channels, ragged planes -- against a double-precision GroupNorm -> qkv -> attention -> proj_out."""
    import subprocess
    exe = os.path.join(os.path.dirname(GOLDEN), "..", "tools", "ubench", "deep_bench")
    # (the only kernel-level check of the in-launch hand-off code must not drop out silently -- on a box where the library
    # built, a missing checker binary is a failure, not a skip)
    assert os.path.exists(exe), "tools/ubench/deep_bench not built (python -c 'import __graft_entry__ as g; g.build()')"
    for mode, ok in (("check", "CHECK OK"), ("attn", "ATTN CHECK OK"), ("block", "BLOCK CHECK OK")):
        out = subprocess.run([exe, mode], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and ok in out.stdout, out.stdout[-2000:] + out.stderr[-500:]


@pytest.mark.parametrize("mt,nt", [(1, 4), (1, 2), (2, 2), (2, 4)])
def test_window_staged_conv_kernel_vs_reference_golden(mt, nt):
    """k_conv_win (csrc/deep.hip: the transformed input window of a row tile staged in LDS once, all nine taps read from it) forced
    onto every eligible 3x3 conv: eps of the base UNet vs the reference golden and a ragged two-clip geometry (row tiles that straddle
    planes, windows clipped by plane borders) vs the oracle."""
    def expect(names):
        assert sum(",80,1]" in n for n in names) >= 20, "the window-staged kernel was not selected"
    _forced("mtv_debug_force_win", (mt, nt), (0, 0), f"k_conv_win<{mt},{nt}>", expect)


@pytest.mark.parametrize("mt,nt,ks", [(2, 4, 4), (2, 2, 2), (1, 4, 2)])
def test_window_staged_conv_kernel_with_k_slices_vs_reference_golden(mt, nt, ks):
    """k_conv_win with 2 / 4 K slices per tile (round 6: slice z stages its 1 / ks of the input channels, partial tiles meet in the plan's
    split-K slab inside the launch, the last slice sums them in slice order and runs the epilogue) forced onto every 3x3 conv that can take
    them (no fused skip conv); same references as the unsliced kernel.  The sum order is fixed: repeats are bit-equal."""
    def expect(names):
        assert sum(f",80,{ks}]" in n or f",80,{ks}x]" in n for n in names) >= 10, "the K-sliced window-staged kernel was not selected"
    _forced("mtv_debug_force_win_ks", (mt, nt, ks), (0, 0, 1), f"k_conv_win<{mt},{nt}> ks{ks}", expect)


@pytest.mark.parametrize("mt,ntw,waves", [(1, 1, 8), (1, 2, 8), (2, 2, 8), (2, 1, 6), (1, 1, 4), (2, 1, 4), (1, 1, 2)])
def test_pointwise_conv_kernel_vs_reference_golden(mt, ntw, waves):
    """k_conv_pw<MT, NTW, NWA> (csrc/deep.hip: the rows of a tile normalised once into LDS, NWA of the 8 waves side by side along N -- column
    tile 16 NTW NWA --, weights in [N][K]) forced onto every eligible 1x1 conv (qkv with its GroupNorm, proj_out with residual + statistics):
    eps of the base UNet vs the reference golden, and a ragged two-clip geometry (partial row tiles, tiles that straddle planes, partial
    column tiles) vs the oracle.  The committed table selects the 6- / 4- / 2-wave forms for the B = 1 step (round 6)."""
    def expect(names):
        assert sum(f",96,{1 if waves == 8 else waves}]" in n for n in names) >= 20, "the pointwise kernel was not selected"
    _forced("mtv_debug_force_pw_waves", (mt, ntw, waves), (0, 0, 0), f"k_conv_pw<{mt},{ntw},{waves}>", expect)


# ----------------------------------------------------------------------------------------------
# fault path and residency of the in-launch hand-offs (include/mtv_hip.h mtv_check_fault)
# ----------------------------------------------------------------------------------------------
def test_hand_off_fault_is_reported_by_the_same_call_and_is_sticky():
    """A poll that times out inside a launch raises a host-mapped fault word and lets the launch fall through with garbage.  The reference
    raises synchronously (unet.py:995-1117 is plain torch); here `strict=True` drains the stream and checks before returning, so the SAME
    call raises; without it `UNetModel.check_fault()` after the caller's own sync does; afterwards every entry point of that context
    refuses to run.  The time-out itself is simulated by mtv_debug_arm_fault (the same system-scope store the kernels' time-out branch does)."""
    from moditalker_amd import MtvError, _lib
    lib = _lib.load()
    dev = _dev()
    net = _build(NARROW_CFG, 11)
    um = net.diffusion_model
    x, cond, ic = [t.to(dev) for t in filler.synthetic_inputs(1, 32, 16, seed=2, tag="fault")]
    t = torch.tensor([77], device=dev)
    a = net(x, cond, ic, t, ) if False else um(x, cond, ic, t, strict=True)       # a clean strict call passes
    um.check_fault()
    _lib.check(lib.mtv_debug_arm_fault(um._ctx), "mtv_debug_arm_fault")
    with pytest.raises(MtvError, match="hand-off timed out"):
        um(x, cond, ic, t, strict=True)                                           # the faulting call itself raises
    with pytest.raises(MtvError, match="hand-off timed out"):
        um(x, cond, ic, t)                                                        # sticky: the context refuses every later call
    with pytest.raises(MtvError):
        um.check_fault()
    # the sampler entry point, non-strict: the call returns, the fault is there once the caller has synchronised
    net2 = _build(NARROW_CFG, 11)
    dm = DDPM(net2, channels=4, image_size=32, sampling_timesteps=4, w=0.0).to(dev)
    noise = [z.to(dev) for z in filler.noise_list(4, (1, 4, 2048), seed=2, tag="fault.n")]
    z0 = dm.sample(batch_size=1, cond=cond, image_cond=ic, noise=noise, strict=True)
    _lib.check(lib.mtv_debug_arm_fault(net2.diffusion_model._ctx), "mtv_debug_arm_fault")
    z1 = dm.sample(batch_size=1, cond=cond, image_cond=ic, noise=noise)           # returns (asynchronous)
    torch.cuda.synchronize(dev)
    with pytest.raises(MtvError, match="hand-off timed out"):
        net2.diffusion_model.check_fault()
    with pytest.raises(MtvError):
        dm.sample(batch_size=1, cond=cond, image_cond=ic, noise=noise, strict=True)
    assert z0.shape == z1.shape and torch.equal(a, a)
    # a fresh context of the same module is clean again
    net2.diffusion_model._release()
    # (a new context re-tunes the shapes the committed tile table does not list and may pick another split-K: fp32 summation-order noise only)
    assert _maxabs(dm.sample(batch_size=1, cond=cond, image_cond=ic, noise=noise, strict=True), z0.cpu()) <= 1e-4


def _cu_masked_stream(n_cus):
    """A HIP stream whose queue may use only the first n_cus CUs (hipExtStreamCreateWithCUMask), wrapped for torch."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    words = (n_cus + 31) // 32
    mask = (C.c_uint32 * words)(*[(0xFFFFFFFF if n_cus >= 32 * (i + 1) else (1 << (n_cus - 32 * i)) - 1) for i in range(words)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(words), mask)
    if rc != 0 or not st.value:
        return None
    return torch.cuda.ExternalStream(st.value, device=_dev())


def test_fewer_resident_cus_keep_the_plan_correct_and_are_enforced():
    """The in-launch hand-offs wait for workgroups of their own launch: they are planned within the CUs the context may count on
    (mtv_resident_cus: the device's count, MTV_RESIDENT_CUS / mtv_debug_resident_cus below it).  With 48 CUs the base plan has no
    one-launch attention block wider than 24 workgroups and no tagged completion wider than 48 -- the three-launch / finalize forms
    take over -- and still meets the reference golden, also when it really runs on a 48-CU queue; a queue with fewer CUs than planned
    for is refused loudly instead of depending on dispatch order."""
    from moditalker_amd import MtvError, _lib
    lib = _lib.load()
    dev = _dev()
    full = _build(BASE_CFG, 7, max_batch=1)
    _check_base_eps(full, "default residency", tvs=(999,))
    ncu = full.diffusion_model.resident_cus
    assert ncu >= 64, ncu
    names_full = [p["name"] for p in full.diffusion_model.profile_forward(1, 1, dev, step=True)]
    _lib.check(lib.mtv_debug_resident_cus(48), "mtv_debug_resident_cus")
    try:
        net = _build(BASE_CFG, 7, max_batch=1)
        inputs = _check_base_eps(net, "48 resident CUs", tvs=(999, 0))
        assert net.diffusion_model.resident_cus == 48
        names = [p["name"] for p in net.diffusion_model.profile_forward(1, 1, dev, step=True)]
        blocks = [n for n in names if " blk " in n]
        assert len(blocks) < len([n for n in names_full if " blk " in n]), "48 CUs cannot hold 128-workgroup attention blocks"
        assert len([n for n in names if n.startswith("fin")]) > len([n for n in names_full if n.startswith("fin")]), "finalize passes expected"
        _check_base_sample(net, "48 resident CUs", 4, inputs)
        st = _cu_masked_stream(48)
        if st is not None:
            x, cond, ic = inputs
            g = _golden("base")
            with torch.cuda.stream(st):
                eps = net.diffusion_model(x.to(dev), cond.to(dev), ic.to(dev), torch.tensor([500], device=dev), strict=True)
            st.synchronize()
            assert report("48 resident CUs, on a 48-CU queue: base eps t=500 vs reference golden", _maxabs(eps, g["eps_t500"]), FWD_TOL) <= FWD_TOL
            narrow = _cu_masked_stream(16)
            if narrow is not None:
                with torch.cuda.stream(narrow):
                    with pytest.raises(MtvError, match="CU mask"):
                        net(x.to(dev), cond.to(dev), ic.to(dev), torch.tensor([500], device=dev))
    finally:
        lib.mtv_debug_resident_cus(0)
