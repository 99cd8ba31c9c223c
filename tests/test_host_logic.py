"""Host-side logic of the product (no GPU): checkpoint key layout, schedule buffers, DDIM step table,
the C-ABI library's exports, loud failures off-GPU, synthetic-data recipe."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import BASE_CFG, GOLDEN, NARROW_CFG, ROOT
from moditalker_amd import DDPM, DiffusionWrapper, MtvError, UNetModel, _lib, filler
from moditalker_amd.ddpm import ddim_step_table, ddim_time_pairs


def test_state_dict_layout_matches_reference_manifest():
    """804 keys, same names, order and shapes as the reference DiffusionWrapper.state_dict()."""
    g = np.load(os.path.join(GOLDEN, "base.npz"))
    with torch.device("meta"):
        m = DiffusionWrapper(UNetModel(**BASE_CFG))
    sd = m.state_dict()
    assert len(sd) == 804
    assert list(sd.keys()) == list(g["keys"])
    assert [",".join(map(str, v.shape)) for v in sd.values()] == list(g["shapes"])
    assert sum(1 for k in sd if ".output_bg_" in k) == 246


def test_zero_init_sites_match_reference():
    """Fresh modules zero the same convs the reference's zero_module does (unet.py:159,242,289,974)."""
    m = UNetModel(**NARROW_CFG)
    sd = m.state_dict()
    zero = [k for k, v in sd.items() if v.dim() >= 2 and not v.any()]
    assert zero and all(re.search(r"(out_layers\.3|proj_out|^out\.2)\.weight$", k) for k in zero)


def test_schedule_buffers_bit_equal_reference():
    g = np.load(os.path.join(GOLDEN, "schedule.npz"))
    dm = DDPM(torch.nn.Identity(), channels=4, image_size=32, sampling_timesteps=50, w=0.0)
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
              "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod"):
        assert np.array_equal(getattr(dm, k).numpy(), g[k]), k
    assert dm.is_ddim_sampling and dm.num_timesteps == 1000 and dm.ddim_sampling_eta == 1.0
    for S in (4, 50, 100, 250):
        times = g[f"times_S{S}"].tolist()
        assert ddim_time_pairs(1000, S) == list(zip(times[:-1], times[1:]))
    dm100 = DDPM(torch.nn.Identity(), sampling_timesteps=100)
    pairs = dm100._time_pairs(0.25)                      # ddpm.py:430: S=100, ratio .25 -> 25 steps from t=249
    assert len(pairs) == 25 and pairs[0][0] == 249 and pairs[-1] == (9, -1)


def test_all_beta_schedules_bit_equal_reference():
    """DDPM(beta_schedule=...) accepts the reference's four names (ddpm.py:78-99); every 50th beta of each schedule at
    (1000, .0015, .0195) was captured from the reference's make_beta_schedule (tests/golden/make_golden_schedules.py)."""
    from moditalker_amd.ddpm import make_beta_schedule
    g = np.load(os.path.join(GOLDEN, "beta_schedules.npz"))
    for name in ("linear", "cosine", "sqrt_linear", "sqrt"):
        b = make_beta_schedule(name, 1000, 0.0015, 0.0195)
        assert b.dtype == np.float64 and b.shape == (1000,)
        assert np.array_equal(b[::50], g[name]), name
    with pytest.raises(ValueError):
        make_beta_schedule("quadratic", 1000)
    dm = DDPM(torch.nn.Identity(), beta_schedule="cosine", sampling_timesteps=50)
    assert float(dm.betas.max()) == np.float32(0.999) and dm.betas.dtype == torch.float32


def test_ddim_step_table_matches_reference_arithmetic():
    from oracle import ref_ddpm
    buf = ref_ddpm.schedule_buffers()
    pairs = ddim_time_pairs(1000, 50)
    steps, n = ddim_step_table(buf["alphas_cumprod"], buf["sqrt_recip_alphas_cumprod"], buf["sqrt_recipm1_alphas_cumprod"], pairs, 1.0)
    assert n == 49 and len(steps) == 50
    ac = buf["alphas_cumprod"]
    for i, (t, tn) in enumerate(pairs):
        s = steps[i]
        assert s.t == t
        assert s.sqrt_recip_ac == float(buf["sqrt_recip_alphas_cumprod"][t])
        if tn < 0:
            assert s.last == 1 and s.noise_index == -1 and i == 49
            continue
        a, an = ac[t], ac[tn]
        sigma = 1.0 * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()      # ddpm.py:393
        c = (1 - an - sigma ** 2).sqrt()                               # ddpm.py:394
        assert (s.sigma, s.c, s.sqrt_ac_next) == (float(sigma), float(c), float(an.sqrt()))
        assert s.noise_index == i and s.last == 0
    assert ctypes.sizeof(_lib.MtvDdimStep) == 32


def test_library_loads_and_exports_every_declared_symbol():
    """Every function include/mtv_hip.h declares is exported (no compute call: no GPU here)."""
    hdr = open(os.path.join(ROOT, "include", "mtv_hip.h")).read()
    declared = set(re.findall(r"\b(mtv_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"mtv_config", "mtv_ctx", "mtv_ddim_step", "mtv_work", "mtv_op_time"}
    lib = _lib.load()
    bound = {name for name, _, _ in _lib.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.mtv_version() >= 1
    assert ctypes.sizeof(_lib.MtvConfig) == 4 * (4 + 8 + 1 + 8 + 5)


def test_no_cpu_fallback():
    net = DiffusionWrapper(UNetModel(**NARROW_CFG)).eval()
    x, cond, ic = filler.synthetic_inputs(1, 32, 16, seed=1, tag="cpu")
    with pytest.raises(MtvError):
        net(x, cond, ic, torch.tensor([5]))
    dm = DDPM(net, channels=4, image_size=32, sampling_timesteps=4, w=0.0)
    with pytest.raises(MtvError):
        dm.sample(batch_size=1, cond=cond, image_cond=ic)
    with pytest.raises(TypeError):
        DDPM(torch.nn.Identity(), sampling_timesteps=4).sample(batch_size=1, cond=cond, image_cond=ic)
    with pytest.raises(NotImplementedError):
        DDPM(net, sampling_timesteps=1000).sample(batch_size=1, cond=cond, image_cond=ic)   # ancestral loop not built
    if not torch.cuda.is_available():
        cfg = _lib.MtvConfig()
        ctx = ctypes.c_void_p()
        rc = _lib.load().mtv_create(ctypes.byref(cfg), ctypes.byref(ctx))
        assert rc < 0 and _lib.load().mtv_last_error()


def test_unsupported_options_raise():
    for kw in (dict(use_spatial_transformer=True, context_dim=512), dict(resblock_updown=False), dict(use_fp16=True),
               dict(num_classes=10), dict(dims=3)):
        with pytest.raises(NotImplementedError):
            UNetModel(**dict(NARROW_CFG, **kw))


def test_product_does_not_import_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "moditalker_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src, f


def test_filler_is_deterministic_and_well_scaled():
    a = filler.fill_tensor("diffusion_model.x.weight", (64, 32, 3, 3), seed=5)
    b = filler.fill_tensor("diffusion_model.x.weight", (64, 32, 3, 3), seed=5)
    assert torch.equal(a, b)
    assert not torch.equal(a, filler.fill_tensor("diffusion_model.y.weight", (64, 32, 3, 3), seed=5))
    assert abs(float(a.std()) - (1.0 / (32 * 9)) ** 0.5) < 0.1 * (1.0 / (32 * 9)) ** 0.5
    n = filler.normal("n", (200000,), seed=1)
    assert abs(float(n.mean())) < 0.01 and abs(float(n.std()) - 1.0) < 0.01


@pytest.mark.parametrize("R,T,levels", [(32, 16, 4), (64, 16, 4), (8, 4, 3), (24, 8, 3), (12, 6, 2)])
def test_arithmetic_gather_matches_torch_unfold_semantics(R, T, levels):
    """The conv kernels compute their im2col source indices arithmetically (csrc/mtv_internal.h: geo_source).
    Pin that arithmetic, through the C ABI and without a device, to what torch's own ops do to an index image:
    F.pad(-1) + F.unfold(3x3) = the zero-padded conv2d of unet.py:131-167 on each plane; F.interpolate(x2,
    nearest) in front of it = Upsample (unet.py:531-561) inside a ResBlock(up=True)."""
    import torch.nn.functional as F
    from moditalker_amd import _lib
    lib = _lib.load()
    assert lib.mtv_selftest_geometry(R, T, levels) == 0
    for lvl in range(levels):
        r, t = R >> lvl, T >> lvl
        planes = [(r, r, 0), (t, r, r * r), (t, r, r * r + t * r)]           # (h, w, first token) of xy | yt | xt
        for up in (0, 1):
            if up and lvl + 1 >= levels:
                continue
            rs, ts = (r >> 1, t >> 1) if up else (r, t)
            src_planes = [(rs, rs, 0), (ts, rs, rs * rs), (ts, rs, rs * rs + ts * rs)]
            for p, ((h, w, off), (hs, ws, offs)) in enumerate(zip(planes, src_planes)):
                img = (torch.arange(hs * ws, dtype=torch.float32) + offs).reshape(1, 1, hs, ws)
                if up:
                    img = F.interpolate(img, scale_factor=2, mode="nearest")
                assert img.shape[-2:] == (h, w)
                cols = F.unfold(F.pad(img, (1, 1, 1, 1), value=-1.0), kernel_size=3)[0].to(torch.int64)   # [9, h*w]
                # sample the plane densely when small, on a stride when large (borders always included)
                toks = range(h * w) if h * w <= 1024 else sorted(set(list(range(0, h * w, 7)) + list(range(w)) + list(range(h * w - w, h * w))
                                                                     + [y * w for y in range(h)] + [y * w + w - 1 for y in range(h)]))
                for local in toks:
                    for tap in range(9):
                        got = lib.mtv_debug_gather_index(r, t, off + local, tap // 3, tap % 3, up)
                        want = int(cols[tap, local])
                        if want < 0:
                            assert got == -1, (lvl, up, p, local, tap)
                        else:
                            assert got >= 0 and (got & 0x0FFFFFFF) == want and (got >> 28) == p, (lvl, up, p, local, tap, got, want)
    assert lib.mtv_debug_gather_index(R, T, R * R + 2 * T * R, 1, 1, 0) == -2      # out of range -> error, not "padding"


def test_autoencoder_state_dict_layout_matches_reference_manifest():
    """415 keys, same names, order and shapes as the reference ViTAutoencoder.state_dict() (fixture written by
    tests/golden/make_golden_ae.py from the imported reference at resolution 64)."""
    from moditalker_amd import BASE_AE_DDCONFIG, ViTAutoencoder
    g = np.load(os.path.join(GOLDEN, "ae.npz"))
    with torch.device("meta"):
        m = ViTAutoencoder(4, dict(BASE_AE_DDCONFIG, resolution=64))
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    assert [",".join(map(str, v.shape)) for v in sd.values()] == [str(x) for x in g["shapes"]]
    with pytest.raises(NotImplementedError):
        ViTAutoencoder(4, BASE_AE_DDCONFIG).forward(torch.zeros(1))
    with pytest.raises(MtvError):                                   # off-GPU: loud, no CPU fallback
        ViTAutoencoder(4, dict(BASE_AE_DDCONFIG, resolution=64)).decode_from_sample(torch.zeros(1, 4, 8 * 8 + 2 * 16 * 8))


def test_committed_tile_table_is_well_formed():
    """moditalker_amd/csrc/tune_gfx950.txt (conv shape -> measured best tile, read at plan build): unique keys, five integers per
    entry, every tile one that a launcher exists for -- and the one-clip R = 32 plan (the metric's workload) keeps the exact-f32
    kernels: none of its shapes is on the split-bf16 pair (NW = 48)."""
    import re
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "moditalker_amd", "csrc", "tune_gfx950.txt")
    x3_tiles = {(4, 2), (8, 2), (4, 4), (2, 2), (4, 1), (2, 1), (8, 1)}           # csrc/conv_x3.hip X3_TILES
    seen, n48, twins = set(), 0, []
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        key, _, val = line.rstrip("\n").partition("|")
        assert key not in seen, key
        seen.add(key)
        mt, nt, nw, ks, xm = (int(v) for v in val.split())
        m = re.match(r"B(\d+) L(\d+)/(\d+)/(\d+) N(\d+) t(\d+) C(\d+)\+(\d+) ", key)
        assert m, key
        if key.endswith(" x"):            # the entry of a 1x1 conv k_lin cannot run, beside the one (same shape) it can
            twins.append(key[:-2])
            assert nw != 64, key
        B, L, N, taps, Cm, Cs = int(m.group(1)), int(m.group(2)), int(m.group(5)), int(m.group(6)), int(m.group(7)), int(m.group(8))
        if nw == 48:                      # k_x3_prep + k_conv_x3<MT, NT>, KS K slices of >= 6 chunks of 32 channels
            n48 += 1
            assert (mt, nt) in x3_tiles and ks in (1, 2, 4, 8) and xm == 0, line
            assert 6 * ks <= taps * (Cm // 32) + Cs // 32 and Cm % 32 == 0 and Cs % 32 == 0 and N % 4 == 0 and N >= 64, line
            assert B * L >= 1024 and not (B == 1 and L <= 2048), line
        elif nw == 64:                    # k_lin<MT, NT, NWV>: 1x1 only
            assert taps == 1 and mt in (1, 2) and nt in (1, 2, 4) and ks in (1, 2, 4), line
        elif nw == 80:                    # k_conv_win<MT, NT> (csrc/deep.hip): 3x3 only; KS = 2 | 4 K slices (round 6):
            # whole 16-channel chunks per slice, at most one round of workgroups.  XM = the XCD-aware block order, set by rule: exactly where
            # the weights are the larger operand
            assert taps == 9 and (mt, nt) in ((1, 4), (1, 2), (2, 2), (2, 4)) and ks in (1, 2, 4) and N % (16 * nt) == 0, line
            if ks > 1:
                assert (Cs // 16) % ks == 0 and (Cm // 16) % ks == 0 and N % 4 == 0 and B * -(-L // (16 * mt)) * (N // (16 * nt)) * ks <= 256, line
            Ls, Lk = int(m.group(3)), int(m.group(4))
            assert xm == (1 if (taps * Cm + Cs) * N >= B * (Ls * Cm + Lk * Cs + L * N) else 0), line
        elif nw == 96:                    # k_conv_pw<MT, NTW, NWA> (csrc/deep.hip): 1x1 on identity rows, whole K per wave; KS = multiplying waves (1 = all 8)
            assert taps == 1 and mt in (1, 2) and nt in (1, 2) and N % (16 * nt) == 0 and Cs == 0 and 64 <= Cm <= 512, line
            assert ks == 1 or (nt == 1 and ks in (2, 4, 6) and N % (16 * ks) == 0), line       # narrower column tiles only where they divide N
            # XM (round 6): a column group's weights through ONE XCD's L2 -- only where the column groups tile the 8 XCDs and the weights
            # dwarf the rows (measured: it pays at 128 tokens x 1536 columns, costs 0.3-0.8 us per launch at 512 tokens)
            groups = N // (16 * nt * (8 if ks == 1 else ks))
            assert xm == (1 if groups % 8 == 0 and N >= 8 * B * L else 0), line
        elif nw == 32:                    # k_conv_lds<WM, WN>
            assert mt in (2, 4) and nt in (2, 4, 8) and ks == 1, line
        else:                             # k_conv<MT, NT, NW>
            assert mt in (1, 2, 4) and nt in (1, 2, 4) and nw in (1, 2, 4, 8, 16) and ks in (1, 2, 4, 8, 16) and xm in (0, 1), line
    assert len(seen) >= 150 and n48 >= 20 and all(k in seen for k in twins)


@pytest.mark.parametrize("R,T,levels", [(32, 16, 4), (64, 16, 4), (8, 4, 3), (16, 8, 4), (24, 8, 3), (16, 8, 2), (4, 2, 1)])
def test_deep_level_row_tables_match_im2col_tables(R, T, levels):
    """csrc/deep.hip: the convs of the levels of <= 128 tokens stage a row group of a channel slice in LDS and address it through
    a host-built table of LDS rows (deep_rowtab).  Every entry -- both row groupings, 3x3 and 1x1, same-level and
    nearest-upsampled source -- against the explicitly constructed im2col tables (host only, through the C ABI)."""
    from moditalker_amd import _lib
    lib = _lib.load()
    assert lib.mtv_selftest_deep(R, T, levels) == 0
    assert lib.mtv_selftest_deep(0, 4, 2) < 0            # bad arguments are an error, not a pass


@pytest.mark.parametrize("tokens,channels,heads,batch", [(32, 512, 8, 1), (32, 512, 8, 2), (128, 512, 8, 1), (128, 512, 8, 2), (128, 256, 8, 1),
                                                         (128, 128, 8, 2), (72, 256, 8, 1), (60, 128, 8, 2), (15, 128, 2, 1), (128, 64, 2, 1),
                                                         (32, 32, 2, 2), (96, 512, 8, 1), (18, 512, 8, 1),
                                                         # 81..96 tokens: LP / 32 = 3 row groups is not a launchable cluster -> two groups of 48 rows
                                                         (96, 256, 8, 2), (91, 256, 8, 2), (96, 128, 8, 1), (80, 512, 8, 1), (112, 256, 8, 1)])
def test_one_launch_attention_block_work_split(tokens, channels, heads, batch):
    """csrc/block.hip, k_deep_block (an attention block of a deep level in one launch): the cluster configuration for the base model's shapes,
    the test models' and ragged token counts -- grid within the residency bound, LDS within 160 KB, every row pair of stage 2 and every
    (query tile, column part) of stage 3 dealt to exactly one workgroup, every granule offset inside its scratch buffer (host only)."""
    from moditalker_amd import _lib
    lib = _lib.load()
    assert lib.mtv_selftest_block(tokens, channels, heads, batch) == 0
    assert lib.mtv_selftest_block(0, 512, 8, 1) < 0
    assert lib.mtv_selftest_block(32, 512, 16, 1) == 1          # 16 heads: more output slabs than a deep tensor holds -- falls back, legitimately


@pytest.mark.parametrize("R,T,levels", [(32, 16, 4), (64, 16, 4), (8, 4, 3), (16, 8, 4), (24, 8, 3), (16, 8, 2), (48, 12, 3)])
def test_window_staged_conv_window_covers_every_tap(R, T, levels):
    """csrc/deep.hip, k_conv_win: the contiguous source-token window a row tile stages in LDS (conv_win_window, shared by the host's
    LDS sizing and the kernel) contains the source of every 3x3 tap of every row of the tile, at every level, for 16- and 32-row tiles,
    same-level and nearest-upsampled sources, and stays inside the source tensor (host only, through the C ABI)."""
    from moditalker_amd import _lib
    lib = _lib.load()
    assert lib.mtv_selftest_win(R, T, levels) == 0
    assert lib.mtv_selftest_win(0, 4, 2) < 0
