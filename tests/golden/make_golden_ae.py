#!/usr/bin/env python3
"""Golden vectors for the autoencoder steps either side of the denoising loop, produced by
running the REFERENCE's ViTAutoencoder itself (build container only; never on the GPU box).

  MToV/models/autoencoder/autoencoder_vit.py  ViTAutoencoder.decode_from_sample / .extract   (imported unmodified)

Weights by the arithmetic recipe of moditalker_amd/filler.py (the rotary-frequency buffers keep the values the reference
computes), inputs by the same recipe.  Two geometries: the shipped one (resolution 256, 16 frames: 16384 tokens) and a
small one (resolution 64, 16 frames: 1024 tokens) that the CPU suite can afford.  Also cross-checks oracle/ref_ae.py.
Outputs: tests/golden/ae.npz (frames sub-sampled to keep the fixture small; full-frame checksums beside them).
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/MToV")

from moditalker_amd import filler  # noqa: E402
from oracle import ref_ae  # noqa: E402


def ddconfig(res):
    return dict(double_z=False, channels=384, resolution=res, timesteps=16, skip=1, in_channels=3, out_ch=3,
                num_res_blocks=2, attn_resolutions=[], splits=1)


def composed_s250():
    """--from-base-s250: BASELINE configs[4] composed at the metric's own schedule.  The reference's OWN 250-step sample
    (tests/golden/base.npz `sample_S250`, written by make_golden.py from the imported reference DDPM + UNet) is decoded by the
    reference's own ViTAutoencoder.decode_from_sample (256x256, recipe weights seed 22) exactly as sample.py:377-387 does
    (clamp, (1 + x) * 127.5, uint8) -> tests/golden/composed_s250.npz.  The GPU test runs HIP sampler (S = 250) -> HIP decode."""
    torch.set_num_threads(os.cpu_count() or 8)
    from einops import rearrange
    from models.autoencoder.autoencoder_vit import ViTAutoencoder
    base = np.load(os.path.join(HERE, "base.npz"))
    z = torch.from_numpy(base["sample_S250"])
    ae = ViTAutoencoder(4, ddconfig(256)).eval()
    filler.fill_autoencoder_(ae, seed=22)
    sd = {k: v.clone() for k, v in ae.state_dict().items()}
    with torch.no_grad():
        fake = ae.decode_from_sample(z).clamp(-1, 1).cpu()                       # sample.py:386
    d = float((ref_ae.decode_from_sample(sd, z, 256, 16).clamp(-1, 1) - fake).abs().max())
    print(f"composed S=250: frames std {float(fake.std()):.3f}, |frames| max {float(fake.abs().max()):.3f}; oracle decode vs reference {d:.3e}")
    assert d <= 2e-5, d
    f255 = (1 + rearrange(fake, "(b t) c h w -> b t h w c", b=1)) * 127.5        # sample.py:387
    u8 = f255.type(torch.uint8)                                                  # sample.py:402 (as stored)
    sub = 5
    np.savez_compressed(os.path.join(HERE, "composed_s250.npz"), ae_seed=np.int64(22), unet_seed=np.int64(int(base["seed"])),
                        frames_sub5=fake[:, :, ::sub, ::sub].contiguous().numpy(), frames_mean_per_frame=fake.mean(dim=(1, 2, 3)).numpy(),
                        frames_abs_sum=np.float64(fake.double().abs().sum()), u8_sub5=u8[0, :, ::sub, ::sub].contiguous().numpy(),
                        u8_sum=np.int64(int(u8.to(torch.int64).sum())))
    print("composed_s250.npz written")
    with open(os.path.join(HERE, "PIN_REPORT.txt"), "a") as f:
        f.write("# composed configs[4] at S = 250: reference sample_S250 -> reference decode_from_sample (make_golden_ae.py --from-base-s250)\n")
        f.write(f"{'oracle decode of the reference S=250 sample':44s} {d:.3e}\n")


def main():
    if "--from-base-s250" in sys.argv[1:]:
        return composed_s250()
    torch.set_num_threads(os.cpu_count() or 8)
    from models.autoencoder.autoencoder_vit import ViTAutoencoder
    out, report = {}, []
    for tag, res, B, sub in (("small", 64, 2, 2), ("full", 256, 1, 5)):
        seed = 21 if tag == "small" else 22
        ae = ViTAutoencoder(4, ddconfig(res)).eval()
        filler.fill_autoencoder_(ae, seed=seed)          # recipe + the two output-layer gains (filler.AE_KEY_GAINS)
        sd = {k: v.clone() for k, v in ae.state_dict().items()}
        r = res // 8
        L = r * r + 2 * 16 * r
        lat = filler.uniform_pm1(f"ae.{tag}.latent", (B, 4, L), seed)
        vid = filler.uniform_pm1(f"ae.{tag}.video", (B, 3, 16, res, res), seed)
        t0 = time.time()
        with torch.no_grad():
            frames = ae.decode_from_sample(lat)
            t1 = time.time()
            z = ae.extract(vid)
        t2 = time.time()
        print(f"reference {tag}: decode_from_sample {t1 - t0:.1f}s, extract {t2 - t1:.1f}s; frames std {float(frames.std()):.3f}, "
              f"|frames|max {float(frames.abs().max()):.3f}, z std {float(z.std()):.3f}")
        fo = ref_ae.decode_from_sample(sd, lat, res, 16)
        zo = ref_ae.extract(sd, vid)
        for nm, a, b in ((f"ae {tag} decode_from_sample", fo, frames), (f"ae {tag} extract", zo, z)):
            d = float((a - b).abs().max())
            report.append((nm, d))
            print(f"  oracle vs reference  {nm:34s} max-abs {d:.3e}")
            assert d <= 2e-5, (nm, d)
        out[f"{tag}_seed"] = np.int64(seed)
        out[f"{tag}_frames_sub{sub}"] = frames[:, :, ::sub, ::sub].contiguous().numpy()
        out[f"{tag}_frames_mean_per_frame"] = frames.mean(dim=(1, 2, 3)).numpy()
        out[f"{tag}_frames_abs_sum"] = np.float64(frames.double().abs().sum())
        out[f"{tag}_extract"] = z.numpy()
        if tag == "small":
            out["keys"] = np.array(list(sd.keys()))
            out["shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
    np.savez_compressed(os.path.join(HERE, "ae.npz"), **out)
    print("ae.npz written")
    with open(os.path.join(HERE, "PIN_REPORT.txt"), "a") as f:
        f.write("# oracle/ref_ae.py vs imported reference ViTAutoencoder (make_golden_ae.py)\n")
        for k, d in report:
            f.write(f"{k:44s} {d:.3e}\n")


if __name__ == "__main__":
    main()
