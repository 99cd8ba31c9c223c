"""The oracle (oracle/ref_unet.py, oracle/ref_ddpm.py) against the golden vectors the REFERENCE
produced (tests/golden/*.npz, written by tests/golden/make_golden.py in the build container).
This is the oracle's pin; it runs on CPU everywhere."""
import os

import numpy as np
import pytest
import torch

from conftest import BASE_CFG, GOLDEN, NARROW_CFG, SHALLOW_CFG
from moditalker_amd import filler
from oracle import ref_ddpm, ref_unet

TOL = 2e-5   # same ATen ops as the reference: differences are thread-count reorder noise only


def _sd(cfg, seed):
    """Recipe-filled state_dict of the keys the forward reads (prefix diffusion_model. as in the wrapper)."""
    from moditalker_amd import UNetModel
    shapes = {}
    with torch.device("meta"):
        m = UNetModel(**cfg)
    for k, v in m.state_dict().items():
        shapes[k] = tuple(v.shape)
    used = set(ref_unet.used_keys(cfg))
    return {k: filler.fill_tensor("diffusion_model." + k, shapes[k], seed) for k in used}


@pytest.mark.parametrize("tag,cfg,seed", [("narrow", NARROW_CFG, 11), ("shallow", SHALLOW_CFG, 12)])
def test_small_models_forward_taps_and_sampling(tag, cfg, seed):
    g = np.load(os.path.join(GOLDEN, f"{tag}.npz"))
    sd = _sd(cfg, seed)
    B = int(g["batch"])
    x, cond, ic = filler.synthetic_inputs(B, 32, 16, seed=seed, tag=tag)
    t = torch.from_numpy(g["t"])
    taps = {}
    eps = ref_unet.unet_forward(sd, cfg, x, cond, ic, t, 32, 16, taps=taps)
    assert float((eps - torch.from_numpy(g["eps"])).abs().max()) <= TOL
    for k in g.files:
        if k.startswith("tap_"):
            assert float((taps[k[4:]][..., ::7] - torch.from_numpy(g[k])).abs().max()) <= TOL, k
    model = lambda a, b, c, d: ref_unet.unet_forward(sd, cfg, a, b, c, d, 32, 16)
    for S, ratio in ((8, None), (20, 0.25)):
        n = ref_ddpm.num_noise_draws(S, ratio)
        noise = filler.noise_list(n, (B, 4, 2048), seed=seed, tag=f"{tag}.S{S}")
        ns = filler.uniform_pm1(f"{tag}.noised_start", (B, 4, 2048), seed) if ratio else None
        z = ref_ddpm.ddim_sample(model, cond, ic, noise, S, noised_start=ns, ratio_=ratio)
        nm = f"sample_S{S}" + (f"_r{ratio}" if ratio else "")
        assert float((z - torch.from_numpy(g[nm])).abs().max()) <= 1e-4, nm
        # fix_noise only changes WHERE the first draw comes from (ddpm.py:424-427); with injected noise it is identical
        if ratio:
            assert np.array_equal(g[nm], g[nm + "_fix"])


def test_base_forward():
    g = np.load(os.path.join(GOLDEN, "base.npz"))
    sd = _sd(BASE_CFG, 7)
    x, cond, ic = filler.synthetic_inputs(1, 32, 16, seed=7, tag="base")
    eps = ref_unet.unet_forward(sd, BASE_CFG, x, cond, ic, torch.tensor([500]), 32, 16)
    assert float((eps - torch.from_numpy(g["eps_t500"])).abs().max()) <= TOL


def test_schedule_and_time_pairs():
    g = np.load(os.path.join(GOLDEN, "schedule.npz"))
    buf = ref_ddpm.schedule_buffers()
    for k, v in buf.items():
        assert np.array_equal(v.numpy(), g[k]), k
    for S in (4, 50, 100, 250):
        times = g[f"times_S{S}"].tolist()
        assert ref_ddpm.ddim_time_pairs(1000, S) == list(zip(times[:-1], times[1:]))
    # S=50: 999, 979, ..., 19, -1 ; S=250: 999, 995, ..., 3, -1  
    assert g["times_S50"][:3].tolist() == [999, 979, 959] and g["times_S50"][-2:].tolist() == [19, -1]
    assert g["times_S250"][:3].tolist() == [999, 995, 991] and g["times_S250"][-2:].tolist() == [3, -1]


def test_block_structure_matches_survey_counts():
    st = ref_unet.block_structure(BASE_CFG)
    nres = sum(1 for s in st["inputs"] + [st["middle"]] + st["outputs"] for l in s if l[0] == "res")
    nattn2 = sum(1 for s in st["inputs"] + [st["middle"]] + st["outputs"] for l in s if l[0] == "attn")
    nattn1 = sum(1 for c in st["in_attn"] if c) + 1 + len(st["out_attn"])
    assert (nres, nattn2, nattn1) == (28, 16, 24)
    assert len(ref_unet.used_keys(BASE_CFG)) == 804 - 246  # 246 dead output_bg_* keys


# ----------------------------------------------------------------------------------------------
# autoencoder steps either side of the loop: oracle/ref_ae.py vs the reference's own outputs
# ----------------------------------------------------------------------------------------------
def _ae_state_dict(g, seed):
    """Recipe-filled ViTAutoencoder state_dict from the key/shape manifest in the fixture; the rotary buffers are
    recomputed exactly as the reference's constructors do (vit_modules.py:22-27, 51-55; autoencoder_vit.py:124-125)."""
    import math
    sd = {}
    for k, shp in zip(g["keys"], g["shapes"]):
        shape = tuple(int(x) for x in str(shp).split(",")) if str(shp) else ()
        k = str(k)
        if k.endswith("inv_freqs"):
            sd[k] = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))
        elif k.endswith("scales"):
            sd[k] = torch.logspace(0.0, math.log(10 / 2) / math.log(2), 64 // 4, base=2)
        elif k == "coords":
            sd[k] = torch.linspace(-1, 1, steps=shape[0]).unsqueeze(-1)
        else:
            sd[k] = filler.fill_tensor(k, shape, seed) * filler.AE_KEY_GAINS.get(k, 1.0)
    return sd


@pytest.mark.parametrize("tag,res,B,sub", [("small", 64, 2, 2), ("full", 256, 1, 5)])
def test_autoencoder_oracle_vs_reference_golden(tag, res, B, sub):
    from oracle import ref_ae
    g = np.load(os.path.join(GOLDEN, "ae.npz"))
    seed = int(g[f"{tag}_seed"])
    sd = _ae_state_dict(g, seed)
    if tag == "full":       # same architecture, other resolution: only the two position-embedding tables change shape
        for k in ("xt_pos_embedding", "yt_pos_embedding"):
            sd[k] = filler.fill_tensor(k, (1, res // 8 + 1, 384), seed)
    r = res // 8
    L = r * r + 2 * 16 * r
    lat = filler.uniform_pm1(f"ae.{tag}.latent", (B, 4, L), seed)
    frames = ref_ae.decode_from_sample(sd, lat, res, 16)
    assert frames.shape == (B * 16, 3, res, res)
    assert float((frames[:, :, ::sub, ::sub] - torch.from_numpy(g[f"{tag}_frames_sub{sub}"])).abs().max()) <= TOL
    assert float((frames.mean(dim=(1, 2, 3)) - torch.from_numpy(g[f"{tag}_frames_mean_per_frame"])).abs().max()) <= TOL
    assert abs(float(frames.double().abs().sum()) - float(g[f"{tag}_frames_abs_sum"])) <= 1e-6 * float(g[f"{tag}_frames_abs_sum"])
    vid = filler.uniform_pm1(f"ae.{tag}.video", (B, 3, 16, res, res), seed)
    z = ref_ae.extract(sd, vid)
    assert z.shape == (B, 4, L)
    assert float((z - torch.from_numpy(g[f"{tag}_extract"])).abs().max()) <= TOL


def test_composed_s250_oracle_decode_of_reference_sample_vs_reference_golden():
    """The composed configs[4] fixture (make_golden_ae.py --from-base-s250): the oracle's decode of the reference's own 250-step
    sample reproduces the reference autoencoder's frames -- the CPU-side pin of what the GPU test composes."""
    from oracle import ref_ae
    g = np.load(os.path.join(GOLDEN, "composed_s250.npz"))
    ga = np.load(os.path.join(GOLDEN, "ae.npz"))
    seed = int(g["ae_seed"])
    sd = _ae_state_dict(ga, seed)
    for k in ("xt_pos_embedding", "yt_pos_embedding"):
        sd[k] = filler.fill_tensor(k, (1, 256 // 8 + 1, 384), seed)
    z = torch.from_numpy(np.load(os.path.join(GOLDEN, "base.npz"))["sample_S250"])
    fake = ref_ae.decode_from_sample(sd, z, 256, 16).clamp(-1, 1)
    assert float((fake[:, :, ::5, ::5] - torch.from_numpy(g["frames_sub5"])).abs().max()) <= TOL
    u8 = ((1 + fake.permute(0, 2, 3, 1)) * 127.5).to(torch.uint8).numpy()
    assert np.abs(u8[:, ::5, ::5].astype(np.int32) - g["u8_sub5"].astype(np.int32)).max() <= 1


def test_cross_attention_oracle_vs_reference_golden():
    from oracle import ref_xattn
    g = np.load(os.path.join(GOLDEN, "xattn.npz"))
    for tag, qd, cd, H, d, B, N, M in (("cross", 256, 128, 8, 32, 2, 96, 77), ("self", 128, None, 4, 64, 1, 160, None)):
        inner = H * d
        shapes = {"to_q.weight": (inner, qd), "to_k.weight": (inner, cd or qd), "to_v.weight": (inner, cd or qd),
                  "to_out.0.weight": (qd, inner), "to_out.0.bias": (qd,)}
        sd = {k: filler.fill_tensor(k, s, 51) for k, s in shapes.items()}
        x = filler.uniform_pm1(f"xattn.{tag}.x", (B, N, qd), 51)
        ctx = filler.uniform_pm1(f"xattn.{tag}.ctx", (B, M, cd), 51) if cd else None
        assert float((ref_xattn.cross_attention(sd, x, ctx, None, H) - torch.from_numpy(g[f"{tag}_out"])).abs().max()) <= 2e-6
        if cd:
            mask = torch.from_numpy(g[f"{tag}_mask"])
            assert float((ref_xattn.cross_attention(sd, x, ctx, mask, H) - torch.from_numpy(g[f"{tag}_out_masked"])).abs().max()) <= 2e-6


def test_split_bf16_attention_emulation():
    """The split-bf16 attention core (csrc/attn_b3.hip: q, k, v as three bf16 terms, p as two, six / five partial products)
    keeps fp32-class accuracy: against an fp64 evaluation its error stays within 4x of plain fp32's (a few 1e-6) on head dims 16 / 32 / 64."""
    from oracle import ref_attn_b3
    torch.manual_seed(0)
    for L, d in ((1024, 16), (512, 32), (128, 64)):
        q, k, v = torch.randn(4, L, d) * 1.5, torch.randn(4, L, d) * 1.5, torch.randn(4, L, d)
        sc = d ** -0.25
        w64 = torch.softmax((q.double() * sc) @ (k.double() * sc).transpose(-1, -2), -1) @ v.double()
        w32 = torch.softmax((q * sc) @ (k * sc).transpose(-1, -2), -1) @ v
        e32 = float((w32.double() - w64).abs().max())
        eb3 = float((ref_attn_b3.attention_split_bf16(q, k, v).double() - w64).abs().max())
        assert eb3 <= 4.0 * e32 + 1e-6, (L, d, e32, eb3)
        # a ONE-term probability (plain bf16 P) would not do: three orders of magnitude worse
        p1 = ref_attn_b3.split_terms(torch.softmax((q * sc) @ (k * sc).transpose(-1, -2), -1), 1)[0] @ v
        assert float((p1.double() - w64).abs().max()) > 50 * e32


def test_split_bf16_conv_and_qk_emulation():
    """The arithmetic of the two split-bf16 forms that ship -- k_conv_x3's GEMM (activations and weights as three bf16 terms, six
    products) and k_attention<QB = 1>'s QK^T -- keeps fp32-class accuracy against fp64: within 4x of plain fp32's error, where two
    terms per operand (three products) are an order of magnitude off (random operands: the errors average over K)."""
    from oracle import ref_attn_b3
    torch.manual_seed(1)
    for M, K, N in ((256, 1152, 128), (128, 4608, 256), (512, 384, 384)):
        a, w = torch.randn(M, K), torch.randn(K, N) * K ** -0.5
        w64 = a.double() @ w.double()
        e32 = float(((a @ w).double() - w64).abs().max())
        e3 = float((ref_attn_b3.gemm_split_bf16(a, w).double() - w64).abs().max())
        assert e3 <= 4.0 * e32 + 1e-6, (M, K, N, e32, e3)
        at, wt = ref_attn_b3.split_terms(a, 2), ref_attn_b3.split_terms(w, 2)
        e2 = float(((at[0] @ wt[0] + at[0] @ wt[1] + at[1] @ wt[0]).double() - w64).abs().max())
        assert e2 > 5 * e32, (e2, e32)
    for L, d in ((1024, 16), (512, 32), (256, 64)):
        q, k, v = torch.randn(4, L, d) * 1.5, torch.randn(4, L, d) * 1.5, torch.randn(4, L, d)
        sc = d ** -0.25
        w64 = torch.softmax((q.double() * sc) @ (k.double() * sc).transpose(-1, -2), -1) @ v.double()
        e32 = float(((torch.softmax((q * sc) @ (k * sc).transpose(-1, -2), -1) @ v).double() - w64).abs().max())
        eqb = float((ref_attn_b3.attention_qk_split_bf16(q, k, v).double() - w64).abs().max())
        assert eqb <= 4.0 * e32 + 1e-6, (L, d, e32, eqb)
