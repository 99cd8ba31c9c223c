"""GPU parity of the autoencoder steps either side of the denoising loop: the HIP path
(C ABI mtv_ae_decode / mtv_ae_extract through moditalker_amd.ViTAutoencoder) against the golden vectors captured from the
reference's own ViTAutoencoder (tests/golden/ae.npz) and against the CPU oracle, plus the end-to-end chain of BASELINE
configs[4]: HIP sampler -> HIP decode, frame-MSE against oracle sampler -> oracle decode."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, SHALLOW_CFG, report
from moditalker_amd import BASE_AE_DDCONFIG, DDPM, DiffusionWrapper, UNetModel, ViTAutoencoder, filler

pytestmark = pytest.mark.gpu

TOL = 1e-3      # BASELINE.json north_star: <= 1e-3 max-abs in fp32 (frames live in (-1, 1))


def _dev():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    return torch.device("cuda:0")


def _ae(res, seed, max_batch=1):
    ae = ViTAutoencoder(4, dict(BASE_AE_DDCONFIG, resolution=res), max_batch=max_batch).eval()
    filler.fill_autoencoder_(ae, seed=seed)
    return ae.to(_dev())


@pytest.mark.parametrize("tag,res,B,sub", [("small", 64, 2, 2), ("full", 256, 1, 5)])
def test_decode_from_sample_vs_reference_golden(tag, res, B, sub):
    g = np.load(os.path.join(GOLDEN, "ae.npz"))
    seed = int(g[f"{tag}_seed"])
    ae = _ae(res, seed, B)
    r = res // 8
    lat = filler.uniform_pm1(f"ae.{tag}.latent", (B, 4, r * r + 2 * 16 * r), seed)
    frames = ae.decode_from_sample(lat.to(_dev())).cpu()
    assert frames.shape == (B * 16, 3, res, res)
    assert float(frames.abs().max()) < 1.0
    d = report(f"autoencoder decode_from_sample {res}^2 vs reference golden", float((frames[:, :, ::sub, ::sub] - torch.from_numpy(g[f"{tag}_frames_sub{sub}"])).abs().max()), TOL)
    assert d <= TOL, d
    assert float((frames.mean(dim=(1, 2, 3)) - torch.from_numpy(g[f"{tag}_frames_mean_per_frame"])).abs().max()) <= 1e-4
    assert abs(float(frames.double().abs().sum()) - float(g[f"{tag}_frames_abs_sum"])) <= 1e-4 * float(g[f"{tag}_frames_abs_sum"])


@pytest.mark.parametrize("tag,res,B", [("small", 64, 2), ("full", 256, 1)])
def test_extract_vs_reference_golden(tag, res, B):
    g = np.load(os.path.join(GOLDEN, "ae.npz"))
    seed = int(g[f"{tag}_seed"])
    ae = _ae(res, seed, B)
    vid = filler.uniform_pm1(f"ae.{tag}.video", (B, 3, 16, res, res), seed)
    z = ae.extract(vid.to(_dev())).cpu()
    r = res // 8
    assert z.shape == (B, 4, r * r + 2 * 16 * r) and float(z.abs().max()) <= 1.0
    d = report(f"autoencoder extract {res}^2 vs reference golden", float((z - torch.from_numpy(g[f"{tag}_extract"])).abs().max()), TOL)
    assert d <= TOL, d


def test_decode_is_deterministic_and_batch_independent():
    ae = _ae(64, 5, 3)
    dev = _dev()
    lat = filler.uniform_pm1("ae.det", (3, 4, 8 * 8 + 2 * 16 * 8), 5).to(dev)
    a = ae.decode_from_sample(lat)
    b = ae.decode_from_sample(lat)
    assert torch.equal(a, b)
    one = ae.decode_from_sample(lat[1:2])
    assert float((a[16:32] - one).abs().max()) <= 2e-5      # another batch size may pick other tiles: summation order only


def test_end_to_end_sampler_then_decode_frame_mse():
    """BASELINE configs[4] at test size: DDIM sampler -> decode_from_sample -> clamp -> uint8 frames, the sequence of
    sample.py:369-386, HIP against the CPU oracle on identical weights / noise; frame MSE and max-abs checked."""
    from oracle import ref_ae, ref_ddpm, ref_unet
    dev = _dev()
    R, T, S = 8, 16, 6                         # 64x64 frames: latent side 8, 16 frames
    cfg = dict(SHALLOW_CFG, image_size=R)
    net = DiffusionWrapper(UNetModel(**cfg, frames=T, max_batch=1)).eval()
    filler.fill_module_(net, seed=33, skip_prefixes=("output_bg_",))
    sd_u = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.to(dev)
    ae = _ae(64, 34)
    sd_a = {k: v.detach().cpu() for k, v in ae.state_dict().items()}
    L = R * R + 2 * T * R
    x, cond, ic = filler.synthetic_inputs(1, R, T, seed=33, tag="e2e")
    noise = filler.noise_list(S, (1, 4, L), seed=33, tag="e2e.noise")
    dm = DDPM(net, channels=4, image_size=R, sampling_timesteps=S, w=0.0).to(dev)
    z = dm.sample(batch_size=1, cond=cond.to(dev), image_cond=ic.to(dev), noise=[n.to(dev) for n in noise])
    frames = ae.decode_from_sample(z).clamp(-1, 1)
    zr = ref_ddpm.ddim_sample(lambda a, b, c, d: ref_unet.unet_forward(sd_u, cfg, a, b, c, d, R, T), cond, ic, noise, S)
    fr = ref_ae.decode_from_sample(sd_a, zr, 64, T).clamp(-1, 1)
    assert float((z.cpu() - zr).abs().max()) <= TOL
    mse = float(((frames.cpu() - fr) ** 2).mean())
    assert mse <= 1e-7, mse
    assert float((frames.cpu() - fr).abs().max()) <= TOL
    u8, u8r = ((1 + frames.cpu()) * 127.5).to(torch.uint8), ((1 + fr) * 127.5).to(torch.uint8)     # sample.py:386-387
    assert float((u8.int() - u8r.int()).abs().max()) <= 1


def test_autoencoder_errors_are_loud():
    ae = _ae(64, 5)
    with pytest.raises(ValueError):
        ae.decode_from_sample(torch.zeros(1, 4, 100, device=_dev()))
    with pytest.raises(ValueError):
        ae.extract(torch.zeros(1, 3, 8, 64, 64, device=_dev()))


def test_two_chunk_chained_run_vs_oracle(tmp_path):
    """--use_last_as_reference chaining (sample.py:344-362,388-398) through the library loop
    (moditalker_amd.pipeline.MToVSampler) on the HIP sampler + HIP autoencoder, against the same sequence composed from the
    CPU oracle's pieces: conditioning assembly (4 extracts, cat), sample, decode, 8-bit last frame -> next chunk's image_cond."""
    from moditalker_amd import pipeline as P
    from oracle import ref_ae, ref_ddpm, ref_unet
    dev = _dev()
    R, T, S, res = 8, 16, 4, 64
    cfg = dict(SHALLOW_CFG, image_size=R)
    net = DiffusionWrapper(UNetModel(**cfg, frames=T, max_batch=1)).eval()
    filler.fill_module_(net, seed=41, skip_prefixes=("output_bg_",))
    sd_u = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.to(dev)
    ae = _ae(res, 42)
    sd_a = {k: v.detach().cpu() for k, v in ae.state_dict().items()}
    L = R * R + 2 * T * R
    dm = DDPM(net, channels=4, image_size=R, sampling_timesteps=S, w=0.0).to(dev)
    sampler = P.MToVSampler(dm, ae, latent_res=R)
    lm = np.stack([np.stack([np.array([20 + 3 * (i % 8) + f, 20 + 4 * (i // 8)]) for i in range(68)]) for f in range(T)])
    chunks, noises = [], []
    for it in range(2):
        vid = (filler.uniform_pm1(f"chain.vid{it}", (1, T, 3, res, res), 41) + 1) * 127.5
        ref = vid[:, :1].expand(-1, T, -1, -1, -1).contiguous()
        x_l = torch.from_numpy(P.landmarks_to_images(lm * 4, WH=256)).float().permute(0, 3, 1, 2)[None, :, :, ::4, ::4].contiguous()
        masked = vid.clone()
        masked[:, :, :, res // 2:] = 0
        chunks.append((ref, vid, x_l, masked))
        noises.append(filler.noise_list(S, (1, 4, L), seed=41 + it, tag="chain.noise"))
    out = sampler.run_identity([tuple(t.to(dev) for t in c) for c in chunks], use_last_as_reference=True, out_dir=str(tmp_path),
                               noise_per_chunk=[[n.to(dev) for n in nz] for nz in noises])
    # ---- the same sequence on the oracle
    model = lambda a, b, c, d: ref_unet.unet_forward(sd_u, cfg, a, b, c, d, R, T)
    image_cond, ref_out = None, []
    for it, (ref, vid, x_l, masked) in enumerate(chunks):
        xr, xv, xl, xm = (P.to_model_range(t) for t in (ref, vid, x_l, masked))
        ic_ = ref_ae.extract(sd_a, xr)
        c = torch.cat([ref_ae.extract(sd_a, xl), ref_ae.extract(sd_a, xm)], dim=1)
        ic = image_cond if image_cond is not None else ic_[:, :, :R * R]
        z = ref_ddpm.ddim_sample(model, c, ic, noises[it], S)
        fake = (1 + ref_ae.decode_from_sample(sd_a, z, res, T).clamp(-1, 1).reshape(1, T, 3, res, res).permute(0, 1, 3, 4, 2)) * 127.5
        last = np.rint(fake[:, -1].numpy()).clip(0, 255).astype(np.uint8)
        image_cond = ref_ae.extract(sd_a, P.reference_from_uint8(last, T))[:, :, :R * R]
        ref_out.append(fake.to(torch.uint8).numpy())
    for a, b in zip(out, ref_out):
        d = np.abs(a.astype(np.int32) - b.astype(np.int32))
        assert d.max() <= 2 and d.mean() <= 0.05, (int(d.max()), float(d.mean()))
    assert len(os.listdir(tmp_path / "frames")) == 2 * T and (tmp_path / "references" / "16" / "0.png").exists()


def test_config4_composed_at_250_steps_vs_reference_golden():
    """BASELINE configs[4] composed at the metric's own schedule against the REFERENCE: HIP sampler (S = 250, the base golden's
    weights / inputs / noise) -> HIP decode_from_sample (256x256) -> clamp -> 8-bit frames, against tests/golden/composed_s250.npz =
    the reference's own 250-step sample decoded by the reference's own autoencoder (make_golden_ae.py --from-base-s250;
    sample.py:377-387,402).  Frames max-abs <= 1e-3, stored 8-bit frames within 1 count."""
    from conftest import BASE_CFG
    from moditalker_amd.pipeline import frames_to_uint8
    g = np.load(os.path.join(GOLDEN, "composed_s250.npz"))
    dev = _dev()
    R, T, S = 32, 16, 250
    L = R * R + 2 * T * R
    from test_gpu_parity import _build          # (recipe-filled weights of a (config, seed) are computed once per session)
    net = _build(BASE_CFG, int(g["unet_seed"]), frames=T, max_batch=1)
    ae = _ae(256, int(g["ae_seed"]))
    x, cond, ic = filler.synthetic_inputs(1, R, T, seed=7, tag="base")
    noise = [z.to(dev) for z in filler.noise_list(S, (1, 4, L), seed=7, tag=f"base.S{S}")]
    dm = DDPM(net, channels=4, image_size=R, sampling_timesteps=S, w=0.0).to(dev)
    z = dm.sample(batch_size=1, cond=cond.to(dev), image_cond=ic.to(dev), noise=noise)
    fake = ae.decode_from_sample(z).clamp(-1, 1).cpu()
    assert fake.shape == (16, 3, 256, 256)
    d = report("configs[4] composed: HIP 250-step sampler -> HIP decode, frames vs reference golden", float((fake[:, :, ::5, ::5] - torch.from_numpy(g["frames_sub5"])).abs().max()), TOL)
    assert d <= TOL, d
    assert float((fake.mean(dim=(1, 2, 3)) - torch.from_numpy(g["frames_mean_per_frame"])).abs().max()) <= 1e-4
    u8 = frames_to_uint8((1 + fake.permute(0, 2, 3, 1)[None]) * 127.5)[0]          # [T, H, W, 3], as sample.py:387,402 stores them
    du = np.abs(u8[:, ::5, ::5].astype(np.int32) - g["u8_sub5"].astype(np.int32))
    assert du.max() <= 1, int(du.max())
    assert abs(int(u8.astype(np.int64).sum()) - int(g["u8_sum"])) <= 1e-4 * int(g["u8_sum"])


def test_config4_full_size_through_conditioning_vs_oracle(tmp_path):
    """BASELINE configs[4] end to end at its own geometry, starting from the files the pipeline starts from: aligned landmark
    .npy files -> landmarks_to_images (cv2.circle restated) -> the four 256x256 extracts of sample.py:328-331 (RGB autoencoder
    for x, x_ref, masked_x; landmark autoencoder for x_l) -> cat -> base second-stage UNet sampler -> decode_from_sample ->
    8-bit frames, through moditalker_amd.pipeline.MToVSampler on the HIP kernels, against the same composition built from
    the CPU oracle's pieces (and the landmark images against oracle/ref_circle.py).  2 DDIM steps keep the oracle side under a
    minute on the GPU box's host (its four 256^2 extracts + decode are most of it); the 250-step schedule is pinned by
    test_base_sampler_vs_reference_golden and, composed with the decode, by test_config4_composed_at_250_steps_vs_reference_golden.
    (Round 6: this test subsumes the former sampler -> decode-only variant at the same geometry.)"""
    from conftest import BASE_CFG
    from moditalker_amd import pipeline as P
    from oracle import ref_ae, ref_circle, ref_ddpm, ref_unet
    dev = _dev()
    R, T, S, res = 32, 16, 2, 256
    from test_gpu_parity import _build
    net = _build(BASE_CFG, 7, frames=T, max_batch=1)
    sd_u = {k: v.detach().cpu().clone() for k, v in net.state_dict().items() if "output_bg_" not in k}
    ae, ae_l = _ae(res, 22), _ae(res, 23)                    # two checkpoints of one architecture (sample.py:206-218)
    sd_a = {k: v.detach().cpu() for k, v in ae.state_dict().items()}
    sd_l = {k: v.detach().cpu() for k, v in ae_l.state_dict().items()}
    L = R * R + 2 * T * R
    # ---- the on-disk inputs: aligned_npy/<id>/NNNNN.npy, one [68, 2] array per frame (align_face_recon.py:347)
    rng = np.random.default_rng(11)
    base = np.stack([[60 + 2.1 * i, 70 + 1.7 * ((i * 7) % 68)] for i in range(68)])
    for f in range(5, 5 + T):
        np.save(tmp_path / f"{str(f).zfill(5)}.npy", base + rng.normal(0, 1.5, size=(68, 2)) + f)
    lm = P.load_aligned_landmarks(str(tmp_path), 5, T)
    img_l = P.landmarks_to_images(lm, WH=256)
    assert np.array_equal(img_l, ref_circle.landmarks_to_images(lm, WH=256)) and img_l.any()
    vid = ((filler.uniform_pm1("cfg4c.vid", (1, T, 3, res, res), 7) + 1) * 127.5).round()
    ref = vid[:, :1].expand(-1, T, -1, -1, -1).contiguous()
    x_l = torch.from_numpy(img_l).float().permute(0, 3, 1, 2)[None].contiguous()
    masked = torch.from_numpy(np.stack([P.crop_lower_half(vid[0, f].numpy(), lm[f]) for f in range(T)])).float()[None]
    noise = filler.noise_list(S, (1, 4, L), seed=7, tag="cfg4c.noise")
    dm = DDPM(net, channels=4, image_size=R, sampling_timesteps=S, w=0.0).to(dev)
    sampler = P.MToVSampler(dm, ae, ae_l, latent_res=R)
    cond = sampler.conditioning(*(t.to(dev) for t in (ref, vid, x_l, masked)))
    z, fake = sampler.sample_chunk(cond, noise=[n.to(dev) for n in noise])
    u8 = P.frames_to_uint8(fake)
    # ---- the oracle composition
    xr, xv, xl, xm = (P.to_model_range(t) for t in (ref, vid, x_l, masked))
    ic_r = ref_ae.extract(sd_a, xr)[:, :, :R * R]
    c_r = torch.cat([ref_ae.extract(sd_l, xl), ref_ae.extract(sd_a, xm)], dim=1)
    assert float((cond["image_cond"].cpu() - ic_r).abs().max()) <= TOL and float((cond["c"].cpu() - c_r).abs().max()) <= TOL
    zr = ref_ddpm.ddim_sample(lambda a, b, c, d: ref_unet.unet_forward(sd_u, BASE_CFG, a, b, c, d, R, T), c_r, ic_r, noise, S)
    fr = (1 + ref_ae.decode_from_sample(sd_a, zr, res, T).clamp(-1, 1).reshape(1, T, 3, res, res).permute(0, 1, 3, 4, 2)) * 127.5
    assert report("configs[4] through conditioning: latents vs oracle", float((z.cpu() - zr).abs().max()), TOL) <= TOL
    assert report("configs[4] through conditioning: frames vs oracle", float(((fake - fr) / 127.5).abs().max()), TOL) <= TOL
    d = np.abs(u8.astype(np.int32) - fr.to(torch.uint8).numpy().astype(np.int32))
    assert u8.shape == (1, T, res, res, 3) and d.max() <= 1 and d.mean() <= 0.01, (int(d.max()), float(d.mean()))


def test_autoencoder_on_split_bf16_gemms_vs_reference_golden():
    """The autoencoder's 16384-token GEMMs forced onto k_x3_prep + k_conv_x3 (split-bf16 kernels, csrc/conv_x3.hip): decode_from_sample
    and extract at the shipped 256x256 geometry against the reference's golden outputs, the same 1e-3 bar as the f32 kernels."""
    from moditalker_amd import _lib
    lib = _lib.load()
    _lib.check(lib.mtv_debug_force_b3(4, 2, 1), "mtv_debug_force_b3")
    try:
        g = np.load(os.path.join(GOLDEN, "ae.npz"))
        seed, res = int(g["full_seed"]), 256
        ae = _ae(res, seed, 1)
        r = res // 8
        lat = filler.uniform_pm1("ae.full.latent", (1, 4, r * r + 2 * 16 * r), seed)
        frames = ae.decode_from_sample(lat.to(_dev())).cpu()
        assert float((frames[:, :, ::5, ::5] - torch.from_numpy(g["full_frames_sub5"])).abs().max()) <= TOL
        vid = filler.uniform_pm1("ae.full.video", (1, 3, 16, res, res), seed)
        z = ae.extract(vid.to(_dev())).cpu()
        assert float((z - torch.from_numpy(g["full_extract"])).abs().max()) <= TOL
        names = [p["name"] for p in ae.profile(1, False, 1)]
        assert any("gemm" in n for n in names)
    finally:
        lib.mtv_debug_force_b3(0, 0, 1)


def test_autoencoder_on_bf16_pipe_qk_attention_vs_reference_golden():
    """The autoencoder's d = 64 attentions with QK^T on the bf16 matrix pipe (k_attention<64, ., ., ., QB = 1>): decode_from_sample
    and extract at the shipped 256x256 geometry against the reference's golden outputs."""
    from moditalker_amd import _lib
    lib = _lib.load()
    _lib.check(lib.mtv_debug_attention_qb(1), "mtv_debug_attention_qb")
    try:
        g = np.load(os.path.join(GOLDEN, "ae.npz"))
        seed, res = int(g["full_seed"]), 256
        ae = _ae(res, seed, 1)
        r = res // 8
        lat = filler.uniform_pm1("ae.full.latent", (1, 4, r * r + 2 * 16 * r), seed)
        frames = ae.decode_from_sample(lat.to(_dev())).cpu()
        assert float((frames[:, :, ::5, ::5] - torch.from_numpy(g["full_frames_sub5"])).abs().max()) <= TOL
        vid = filler.uniform_pm1("ae.full.video", (1, 3, 16, res, res), seed)
        z = ae.extract(vid.to(_dev())).cpu()
        assert float((z - torch.from_numpy(g["full_extract"])).abs().max()) <= TOL
    finally:
        lib.mtv_debug_attention_qb(-1)
