"""Host logic of the callers / data formats either side of the loop (moditalker_amd/pipeline.py):
conditioning assembly, 8-bit chaining round trip, frame naming, chunk ordering -- CPU only, the models are stubs."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
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


def test_overlapping_chunks_of_the_cross_identity_script(tmp_path):
    """sample_crossID.py:185,343-353 (`--overlap`, `--num_frames`): chunks every 8 frames; chunk `it` is chained to the chunk that ENDED at
    its first frame (it - 2: `references/<ldmk_srt>`), the first two keep their own x_ref; frame files of the overlapped half are
    overwritten; the loop stops before the first chunk that starts past num_frames."""
    ae, dm = _StubAE(), _StubDM()
    s = P.MToVSampler(dm, ae)
    mk = lambda v: torch.full((1, 16, 3, 8, 8), float(v))
    chunks = [(mk(255), mk(10 + i), mk(20), mk(30)) for i in range(5)]
    out = s.run_identity(chunks, use_last_as_reference=True, out_dir=str(tmp_path), overlap=True, num_frames=24)
    assert len(out) == 4 and len(dm.calls) == 4                 # chunks start at 0, 8, 16, 24; the fifth would start at 32 > 24
    assert sorted(os.listdir(tmp_path / "references")) == ["16", "24", "32", "40"]
    assert len(os.listdir(tmp_path / "frames")) == 24 + 16      # frames 0 .. 39
    from PIL import Image
    own = 1.0                                                   # x_ref = 255 -> +1
    v16 = float(np.asarray(Image.open(tmp_path / "references" / "16" / "0.png")).mean()) / 255 * 2 - 1
    v24 = float(np.asarray(Image.open(tmp_path / "references" / "24" / "0.png")).mean()) / 255 * 2 - 1
    got = [float(c["image_cond"].mean()) for c in dm.calls]
    assert abs(got[0] - own) < 1e-6 and abs(got[1] - own) < 1e-6          # nothing ended at frame 0 or 8
    assert abs(got[2] - v16) < 1e-6 and abs(got[3] - v24) < 1e-6          # chunk 2 <- chunk 0's last frame, chunk 3 <- chunk 1's
    # without overlap the same call chains every chunk to its predecessor (stride = T)
    ae2, dm2 = _StubAE(), _StubDM()
    P.MToVSampler(dm2, ae2).run_identity(chunks[:3], use_last_as_reference=True, out_dir=str(tmp_path / "plain"))
    assert sorted(os.listdir(tmp_path / "plain" / "references"), key=int) == ["16", "32", "48"]


def test_batched_identity_writes_the_reference_grid_and_ignores_stale_pngs(tmp_path):
    """B = 2 clips per call: frames/NNNN.png holds both clips side by side ([H, 2W, 3], the reference's grid_size=(k, 1),
    sample.py:79-104), and the chained image_cond is read back from exactly the two files just written -- a stale
    `5.png` of an earlier, larger batch in the same folder does not join the batch."""
    ae, dm = _StubAE(), _StubDM()
    s = P.MToVSampler(dm, ae)
    mk = lambda a, b: torch.cat([torch.full((1, 16, 3, 8, 8), float(a)), torch.full((1, 16, 3, 8, 8), float(b))], dim=0)
    chunks = [(mk(255, 0), mk(10, 90), mk(20, 80), mk(30, 70)), (mk(255, 0), mk(40, 60), mk(50, 50), mk(60, 40))]
    from PIL import Image
    stale = tmp_path / "references" / "16"
    os.makedirs(stale)
    Image.fromarray(np.full((8, 8, 3), 77, np.uint8), "RGB").save(stale / "5.png")
    out = s.run_identity(chunks, use_last_as_reference=True, out_dir=str(tmp_path))
    assert out[0].shape == (2, 16, 8, 8, 3)
    grid = np.asarray(Image.open(tmp_path / "frames" / "0000.png"))
    assert grid.shape == (8, 16, 3)
    assert np.array_equal(grid[:, :8], out[0][0, 0]) and np.array_equal(grid[:, 8:], out[0][1, 0])
    assert dm.calls[1]["image_cond"].shape[0] == 2            # not 3: the stale file stayed out


def _read_bitmaps(name):
    out, cur, title = [], [], None
    for line in open(os.path.join(GOLDEN, name)).read().splitlines():
        if line.startswith("#"):
            if title is not None:
                out.append((title, np.array(cur, dtype=np.uint8)))
            title, cur = line[2:], []
        else:
            cur.append([1 if ch == "X" else 0 for ch in line])
    out.append((title, np.array(cur, dtype=np.uint8)))
    return out


def test_disc_rasteriser_matches_opencv_circle_golden_bitmaps():
    """cv2.circle(radius=3, thickness=-1) (dataloader_sample.py:169; radius 6 in the line above it): the product's row
    table against the committed bitmaps of drawing.cpp Circle() -- whole discs at r = 3 and r = 6 and r = 3 discs cut by
    every border and corner, or lying outside (tests/golden/make_golden_circle.py; hand-checked rows in oracle/ref_circle.py)."""
    import re
    for name in ("circle_r3.txt", "circle_r6.txt", "circle_clipped.txt"):
        for title, want in _read_bitmaps(name):
            r, cx, cy = (int(v) for v in re.match(r"radius (\d+), centre \((-?\d+),(-?\d+)\)", title).groups())
            img = np.zeros(want.shape, np.uint8)
            P._draw_disc(img, cx, cy, P._disc_rows(r), value=1)
            assert np.array_equal(img, want), title
    assert int(_read_bitmaps("circle_r3.txt")[0][1].sum()) == 29 and int(_read_bitmaps("circle_r6.txt")[0][1].sum()) == 113


def test_landmark_images_match_circle_restatement_everywhere():
    """landmarks_to_images against oracle.ref_circle (Circle() restated with its `inside` and clipped branches): random
    landmark sets incl. points on and beyond every border, both landmark formats, with and without the flip; radii 1..9."""
    from oracle import ref_circle
    rng = np.random.default_rng(5)
    lm = rng.integers(-6, 262, size=(3, 68, 2))
    lm[0, :8] = [[0, 0], [255, 255], [0, 255], [255, 0], [3, 3], [252, 252], [-3, 100], [100, 258]]
    for flip in (False, True):
        assert np.array_equal(P.landmarks_to_images(lm, flip=flip), ref_circle.landmarks_to_images(lm, flip=flip))
    lm3 = rng.uniform(-1.05, 1.05, size=(2, 68, 3))
    assert np.array_equal(P.landmarks_to_images(lm3, flip=True), ref_circle.landmarks_to_images(lm3, flip=True))
    for r in range(1, 10):
        for cx, cy in [(12, 12), (0, 3), (24, 24), (-r, 5), (5, 24 + r), (30, 30)]:
            a, b = np.zeros((25, 25), np.uint8), np.zeros((25, 25), np.uint8)
            P._draw_disc(a, cx, cy, P._disc_rows(r), value=255)
            ref_circle.circle_filled(b, (cx, cy), r)
            assert np.array_equal(a, b), (r, cx, cy)


def test_gif_output_and_video_mux_is_loud_without_its_tools(tmp_path):
    """sample.py:55-76 / 107-116: GIF through PIL (always available); the mp4 mux needs imageio + ffmpeg like the reference and
    says so when they are absent instead of writing nothing."""
    from PIL import Image
    fr = (np.arange(4 * 8 * 8 * 3) % 251).astype(np.uint8).reshape(4, 8, 8, 3)
    name = P.save_gif(fr, str(tmp_path / "generated_gif.gif"))
    im = Image.open(name)
    assert im.n_frames == 4 and im.size == (8, 8)
    try:
        import imageio  # noqa: F401
    except ImportError:
        with pytest.raises(RuntimeError, match="imageio"):
            P.make_video(fr, None, str(tmp_path / "out.mp4"))
