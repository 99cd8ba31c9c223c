"""Host logic of the callers / data formats either side of the loop (moditalker_amd/pipeline.py):
conditioning assembly, 8-bit chaining round trip, frame naming, chunk ordering -- CPU only, the models are stubs."""
import os

import numpy as np
import pytest
import torch

from moditalker_amd import pipeline as P


def test_disc_rows_are_opencvs_radius3_filled_circle():
    rows = dict(P._disc_rows(3))
    assert rows == {0: 3, 1: 2, -1: 2, 2: 2, -2: 2, 3: 0, -3: 0}       # 7 + 5*4 + 1*2 = 29 pixels
    assert sum(2 * hw + 1 for hw in rows.values()) == 29
    rows1 = dict(P._disc_rows(1))
    assert rows1 == {0: 1, 1: 0, -1: 0}                                  # the plus sign cv2 draws for radius 1


def test_landmarks_to_images_scaling_clipping_flip():
    lm = np.zeros((2, 68, 2), dtype=np.int64)
    lm[0, :] = (100, 50)
    lm[1, :] = (0, 255)                       # a corner: the disc is clipped at the border
    img = P.landmarks_to_images(lm, WH=256, flip=False)
    assert img.shape == (2, 256, 256, 3) and img.dtype == np.uint8
    assert img[0].sum() == 29 * 3 * 255 and img[0, 50, 100, 0] == 255 and img[0, 50, 97, 0] == 255 and img[0, 53, 100, 0] == 255
    assert img[0, 53, 101, 0] == 0 and img[0, 51, 103, 0] == 0
    assert img[1, 255, 0, 0] == 255 and img[1].sum() == (4 + 3 + 3 + 1) * 3 * 255     # rows dy = 0,-1,-2,-3, right halves only
    big = P.landmarks_to_images(lm * 2, WH=512)                         # dataloader_sample.py:170: x / WH * 256
    assert np.array_equal(big, img)
    assert np.array_equal(P.landmarks_to_images(lm, flip=True), img[:, ::-1])
    lm3 = np.zeros((1, 68, 3)); lm3[..., 0] = -0.5; lm3[..., 1] = 0.25   # normalised 3-D form: (lm * WH/2 + WH/2)
    assert P.landmarks_to_images(lm3)[0, 160, 64, 0] == 255


def test_crop_lower_half_and_model_range():
    img = np.full((3, 8, 8), 200.0)
    lmk = np.zeros((68, 2)); lmk[33] = (4, 5.9)
    out = P.crop_lower_half(img, lmk)
    assert out.dtype == np.uint8 and (out[:, :5] == 200).all() and (out[:, 5:] == 0).all()
    x = torch.full((2, 16, 3, 4, 4), 255.0)
    y = P.to_model_range(x)
    assert y.shape == (2, 3, 16, 4, 4) and float(y.max()) == 1.0 and float(P.to_model_range(x * 0).min()) == -1.0


def test_uint8_round_trip_and_frame_files(tmp_path):
    fake = torch.rand(2, 16, 8, 8, 3) * 255
    u8 = P.frames_to_uint8(fake)
    assert u8.dtype == np.uint8 and np.array_equal(u8, np.floor(fake.numpy()).astype(np.uint8))       # .type(uint8) truncates
    last = P.last_frame_to_uint8(fake)
    assert np.array_equal(last, np.rint(fake[:, -1].numpy()).clip(0, 255).astype(np.uint8))           # the chain rounds
    ref = P.reference_from_uint8(last, frames=16)
    assert ref.shape == (2, 3, 16, 8, 8)
    assert torch.equal(ref[:, :, 0], ref[:, :, 15]) and float(ref.max()) <= 1.0 and float(ref.min()) >= -1.0
    assert torch.allclose(ref[:, :, 0].permute(0, 2, 3, 1), torch.from_numpy(last).float() / 255 * 2 - 1)
    names = P.save_frames(32, u8[0], str(tmp_path / "frames"))
    assert [os.path.basename(n) for n in names[:2]] == ["0032.png", "0033.png"] and len(names) == 16
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(names[3])), u8[0, 3])                                  # lossless


def test_aligned_landmark_files(tmp_path):
    for i in range(3, 7):
        np.save(tmp_path / f"{str(i).zfill(5)}.npy", np.full((68, 2), i))
    lm = P.load_aligned_landmarks(str(tmp_path), 4, 2)
    assert lm.shape == (2, 68, 2) and lm[0, 0, 0] == 4 and lm[1, 0, 0] == 5


class _StubAE(torch.nn.Module):
    """extract: mean colour per clip spread over the latent; decode: latent mean -> constant frames.  Records calls."""

    def __init__(self):
        super().__init__()
        self.p = torch.nn.Parameter(torch.zeros(1))
        self.s = 16
        self.calls = []

    def extract(self, x):
        self.calls.append(("extract", tuple(x.shape), float(x.mean())))
        return x.mean(dim=(1, 2, 3, 4), keepdim=False)[:, None, None].expand(-1, 4, 2048).clone()

    def decode_from_sample(self, z):
        self.calls.append(("decode", tuple(z.shape)))
        return z.mean(dim=(1, 2))[:, None, None, None, None].expand(-1, 16, 3, 8, 8).reshape(-1, 3, 8, 8).clone()


class _StubDM:
    def __init__(self):
        self.calls = []

    def sample(self, batch_size, cond, image_cond, noised_start, ratio_, fix_noise, noise):
        self.calls.append(dict(batch_size=batch_size, cond=tuple(cond.shape), image_cond=image_cond.clone(), noised_start=noised_start,
                               ratio_=ratio_, fix_noise=fix_noise))
        return image_cond.mean() * torch.ones(batch_size, 4, 2048) * 0.5 + 0.1 * len(self.calls)


def test_chunk_loop_ordering_and_chaining(tmp_path):
    ae, dm = _StubAE(), _StubDM()
    s = P.MToVSampler(dm, ae)
    mk = lambda v: torch.full((1, 16, 3, 8, 8), float(v))
    chunks = [(mk(255), mk(10), mk(20), mk(30)), (mk(255), mk(40), mk(50), mk(60))]
    out = s.run_identity(chunks, use_last_as_reference=True, out_dir=str(tmp_path), ratio_=0.3, fix_noise=True, x_noisy_start=True)
    assert len(out) == 2 and out[0].shape == (1, 16, 8, 8, 3) and out[0].dtype == np.uint8
    # per chunk: extract x, x_ref, x_l, masked_x (sample.py:328-331), sample, decode, then the chained extract
    kinds = [c[0] for c in ae.calls]
    assert kinds == ["extract"] * 4 + ["decode", "extract"] + ["extract"] * 4 + ["decode", "extract"]
    assert ae.calls[0][1] == (1, 3, 16, 8, 8)
    first, second = dm.calls
    assert first["cond"] == (1, 8, 2048) and first["image_cond"].shape == (1, 4, 1024) and first["ratio_"] == 0.3 and first["fix_noise"]
    assert first["noised_start"] is not None and tuple(first["noised_start"].shape) == (1, 4, 2048)      # --x_noisy_start: extract(x_ref)
    assert abs(float(first["image_cond"].mean()) - 1.0) < 1e-6                                          # x_ref = 255 -> +1
    # chunk 2's image_cond comes from chunk 1's last frame through the 8-bit file, not from its own x_ref
    png = tmp_path / "references" / "16" / "0.png"
    assert png.exists() and (tmp_path / "references" / "32" / "0.png").exists()
    from PIL import Image
    v = float(np.asarray(Image.open(png)).mean()) / 255 * 2 - 1
    assert abs(float(second["image_cond"].mean()) - v) < 1e-6 and abs(v - 1.0) > 1e-3
    assert sorted(os.listdir(tmp_path / "frames"))[0] == "0000.png" and len(os.listdir(tmp_path / "frames")) == 32
    # unchained: every chunk uses its own reference
    ae2, dm2 = _StubAE(), _StubDM()
    P.MToVSampler(dm2, ae2).run_identity(chunks, use_last_as_reference=False)
    assert all(abs(float(c["image_cond"].mean()) - 1.0) < 1e-6 for c in dm2.calls) and all(c["noised_start"] is None for c in dm2.calls)
