"""The callers and data formats either side of the denoising loop: the reference's
per-identity sampling loop as a library, driving the HIP sampler (moditalker_amd.DDPM) and the HIP autoencoder
(moditalker_amd.ViTAutoencoder).

Restates, step for step,
  MToV/sample.py:305-400                      the 16-frame chunk loop: conditioning assembly -> DDPM.sample -> decode -> frames
  MToV/sample.py:344-362,388-398              --use_last_as_reference chaining (last decoded frame, through an 8-bit image,
                                              becomes the next chunk's image_cond)
  MToV/sample.py:79-104                       frames/NNNN.png output (save_image_at_folder)
  MToV/tools/dataloader_sample.py:130-139     masked_x: rows from landmark 33's y downwards zeroed
  MToV/tools/dataloader_sample.py:153-179     x_l: 68 landmarks drawn as filled radius-3 discs on black 256x256
  data/data_utils/motion_align/align_face_recon.py:347   aligned_npy/<id>/NNNNN.npy: one [68, 2] landmark array per frame

This is host logic (numpy / PIL for the image formats, torch for device tensors); every model call goes to the HIP library.
`sample.py` itself is not importable (module-level argparse, hard-wired paths, cv2 / torchvision / omegaconf) and the disc
rasteriser it uses, cv2.circle, is not installed here: `_disc_rows` produces the row spans of OpenCV's filled-circle scan
conversion (modules/imgproc/src/drawing.cpp, Circle(), OpenCV 4.x).  It is checked bit for bit against golden bitmaps and
against the test suite's statement-for-statement restatement of Circle() incl. its clipping path
(tests/test_pipeline_host.py); cv2 itself never ran here, see DESIGN.md section 4.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch


# ------------------------------------------------------------------------------------------------------------------
# conditioning assembly (dataloader_sample.py)
# ------------------------------------------------------------------------------------------------------------------
def _disc_rows(radius: int) -> List[Tuple[int, int]]:
    """(dy, half_width) rows of cv2.circle(..., radius, thickness=-1): OpenCV's midpoint scan conversion.
    radius 3 -> half widths 3,2,2,0 at |dy| = 0,1,2,3."""
    rows = {}
    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1
    while dx >= dy:
        rows[dy] = max(rows.get(dy, -1), dx)       # rows cy +- dy span cx +- dx
        rows[dx] = max(rows.get(dx, -1), dy)       # rows cy +- dx span cx +- dy
        dy += 1
        err += plus
        plus += 2
        mask = -1 if err > 0 else 0                # (err <= 0) - 1
        err -= minus & mask
        dx += mask
        minus -= mask & 2
    out = []
    for d, hw in rows.items():
        out.append((d, hw))
        if d:
            out.append((-d, hw))
    return out


def _draw_disc(img: np.ndarray, cx: int, cy: int, rows: Sequence[Tuple[int, int]], value: int = 255) -> None:
    """One filled disc, clipped to the image ([H, W] or [H, W, C], in place): row cy + dy spans cx - hw .. cx + hw."""
    h, w = img.shape[:2]
    for dy, hw in rows:
        yy = cy + dy
        if 0 <= yy < h:
            x0, x1 = max(cx - hw, 0), min(cx + hw, w - 1)
            if x0 <= x1:
                img[yy, x0:x1 + 1] = value


def landmarks_to_images(lm: np.ndarray, WH: int = 256, flip: bool = False) -> np.ndarray:
    """dataloader_sample.py:153-179 `_change_np_img_size`: lm [T, 68, 2] (image-sized ints) or [T, 68, 3] (normalised
    3-D) -> uint8 [T, 256, 256, 3], white filled radius-3 discs on black; optional vertical flip."""
    lm = np.asarray(lm)
    T = lm.shape[0]
    if lm.shape[-1] == 3:
        lm2d = (lm * WH / 2 + WH / 2).astype(int)[:, :, :2]
    else:
        lm2d = lm.astype(int)
    img = np.zeros([T, 256, 256, 3], dtype=np.uint8)
    rows = _disc_rows(3)
    for b in range(T):
        for x, y in lm2d[b]:
            _draw_disc(img[b], int(x / WH * 256.0), int(y / WH * 256.0), rows)
    if flip:
        img = img[:, ::-1].copy()
    return img


def crop_lower_half(img: np.ndarray, landmarks: np.ndarray) -> np.ndarray:
    """dataloader_sample.py:130-139: zero every row from landmark 33's y downwards. img [C, H, W] (0..255)."""
    mask = np.ones(img.shape[-2:])
    mask[int(landmarks[33][1]):, :] = 0.0
    return (img * mask[None]).astype(np.uint8)


def load_aligned_landmarks(folder: str, start: int, count: int) -> np.ndarray:
    """aligned_npy/<id>/NNNNN.npy (align_face_recon.py:347): frames start .. start+count-1 -> [count, 68, 2]."""
    return np.stack([np.load(os.path.join(folder, f"{str(i).zfill(5)}.npy")) for i in range(start, start + count)], axis=0)


def to_model_range(x: torch.Tensor) -> torch.Tensor:
    """sample.py:321-325: [B, T, C, H, W] in 0..255 -> [B, C, T, H, W] in [-1, 1]."""
    return (x / 127.5 - 1).permute(0, 2, 1, 3, 4).contiguous()


# ------------------------------------------------------------------------------------------------------------------
# on-disk formats (sample.py)
# ------------------------------------------------------------------------------------------------------------------
def frames_to_uint8(fake: torch.Tensor) -> np.ndarray:
    """sample.py:402 / :97-99: frames in 0..255 float -> uint8 the way the reference stores them (`fake.type(torch.uint8)`:
    truncation, then the no-op rint/clip of save_image_at_folder)."""
    return fake.to(torch.uint8).cpu().numpy()


def save_frames(start_iter: int, frames_u8: np.ndarray, folder: str) -> List[str]:
    """sample.py:79-104 save_image_at_folder for one identity column: frames_u8 [T, H, W, 3] -> <folder>/NNNN.png."""
    from PIL import Image
    os.makedirs(folder, exist_ok=True)
    names = []
    for i in range(frames_u8.shape[0]):
        name = os.path.join(folder, f"{start_iter + i}".zfill(4) + ".png")
        Image.fromarray(frames_u8[i], "RGB").save(name)
        names.append(name)
    return names


def save_gif(frames_u8: np.ndarray, fname: str, duration_ms: int = 100) -> str:
    """sample.py:55-76 save_image_grid's output format: frames_u8 [T, H, W, 3] -> an animated GIF (PIL: `save_all`, 100 ms per
    frame, loop forever, the reference's arguments)."""
    from PIL import Image
    imgs = [Image.fromarray(f, "RGB") for f in frames_u8]
    imgs[0].save(fname, quality=95, save_all=True, append_images=imgs[1:], duration=duration_ms, loop=0)
    return fname


def make_video(result_frames, audio_path: Optional[str], save_path: str, fps: int = 25) -> str:
    """sample.py:107-116: frames -> <save_path minus .mp4>no_audio.mp4 (imageio / ffmpeg writer at `fps`), then ffmpeg muxes the
    driving audio in (`-map 0:v -map 1:a -c:v copy -shortest`) and the silent file is removed.  Host-only and after the path;
    imageio and the ffmpeg binary are the reference's own dependencies and are not part of this image: without them this raises
    a RuntimeError that says so (no silent fallback), `audio_path=None` skips the mux."""
    import shutil
    import subprocess
    try:
        import imageio
    except ImportError as e:
        raise RuntimeError("make_video needs imageio (+ imageio-ffmpeg), as MToV/sample.py does; PNG frames and GIFs need only PIL") from e
    if audio_path and not save_path.endswith(".mp4"):
        # (the reference forms the silent name with .replace('.mp4', ...): without that suffix ffmpeg would read and write the SAME
        # file and the final os.remove would delete the result)
        raise ValueError("make_video: save_path must end in .mp4 when an audio track is muxed in (sample.py:107-116)")
    silent = os.path.splitext(save_path)[0] + "no_audio.mp4" if audio_path else save_path
    imageio.mimwrite(silent, list(result_frames), fps=fps, output_params=["-vf", f"fps={fps}"])
    if audio_path:
        if shutil.which("ffmpeg") is None:
            raise RuntimeError("make_video: the ffmpeg binary is needed to mux the audio track (sample.py:110-113)")
        subprocess.run(["ffmpeg", "-y", "-i", silent, "-i", audio_path, "-map", "0:v", "-map", "1:a", "-c:v", "copy", "-shortest", save_path], check=True)
        os.remove(silent)
    return save_path


def last_frame_to_uint8(fake: torch.Tensor) -> np.ndarray:
    """sample.py:391-398: the last frame of every clip as stored in references/<n>/<idx>.png: rint, clip, uint8
    (the BGR<->RGB swaps of cvtColor + imwrite cancel).  fake [B, T, H, W, 3] in 0..255 -> [B, H, W, 3] uint8."""
    return np.rint(np.asarray(fake[:, -1].cpu(), dtype=np.float32)).clip(0, 255).astype(np.uint8)


def reference_from_uint8(last_u8: np.ndarray, frames: int = 16) -> torch.Tensor:
    """sample.py:349-360: an 8-bit RGB frame per clip -> ToTensor (/255) -> *2-1 -> repeated over `frames` frames:
    [B, 3, frames, H, W] in [-1, 1], the input of extract() for the chained image_cond."""
    t = torch.from_numpy(last_u8).float().div(255.0).permute(0, 3, 1, 2) * 2.0 - 1.0      # [B, 3, H, W]
    return t.unsqueeze(2).expand(-1, -1, frames, -1, -1).contiguous()


# ------------------------------------------------------------------------------------------------------------------
# the chunk loop (sample.py:305-400)
# ------------------------------------------------------------------------------------------------------------------
class MToVSampler:
    """diffusion_model: moditalker_amd.DDPM; first_stage_model / first_stage_model_ldmk: moditalker_amd.ViTAutoencoder
    (the reference loads two checkpoints of the same architecture, sample.py:206-218)."""

    def __init__(self, diffusion_model, first_stage_model, first_stage_model_ldmk=None, latent_res: int = 32):
        self.dm = diffusion_model
        self.ae = first_stage_model
        self.ae_ldmk = first_stage_model_ldmk if first_stage_model_ldmk is not None else first_stage_model
        self.n_xy = latent_res * latent_res                 # the reference hard-wires 32 * 32 (sample.py:332)

    @torch.no_grad()
    def conditioning(self, x_ref, x, x_l, masked_x):
        """sample.py:321-332,366: inputs [B, T, C, H, W] in 0..255 on the device -> dict of latents."""
        x_ref, x, x_l, masked_x = (to_model_range(t) for t in (x_ref, x, x_l, masked_x))
        z_ = self.ae.extract(x)
        image_cond_ = self.ae.extract(x_ref)
        z_l = self.ae_ldmk.extract(x_l)
        masked_z = self.ae.extract(masked_x)
        return dict(z_=z_, image_cond_=image_cond_, image_cond=image_cond_[:, :, 0:self.n_xy],
                    c=torch.cat([z_l, masked_z], dim=1))

    @torch.no_grad()
    def sample_chunk(self, cond: dict, image_cond: Optional[torch.Tensor] = None, x_noisy_start: bool = False,
                     refvid_noisy_start: bool = False, ratio_: Optional[float] = None, fix_noise: bool = False, noise=None):
        """sample.py:366-387 -> (z [B,4,L], fake [B,T,H,W,3] float in 0..255 on the CPU like the reference)."""
        k = cond["c"].shape[0]
        noised_start = None
        if x_noisy_start:
            noised_start = cond["image_cond_"].float()
        elif refvid_noisy_start:
            noised_start = cond["z_"].float()
        z = self.dm.sample(batch_size=k, cond=cond["c"].float(),
                           image_cond=(image_cond if image_cond is not None else cond["image_cond"]).float(),
                           noised_start=noised_start, ratio_=ratio_, fix_noise=fix_noise, noise=noise)
        fake = self.ae.decode_from_sample(z).clamp(-1, 1).cpu()
        T = fake.shape[0] // k
        fake = (1 + fake.reshape(k, T, *fake.shape[1:]).permute(0, 1, 3, 4, 2)) * 127.5
        return z, fake

    @torch.no_grad()
    def chained_image_cond(self, fake: torch.Tensor, out_dir: Optional[str] = None, ldmk_end: int = 0) -> torch.Tensor:
        """sample.py:344-362,388-398: the next chunk's image_cond from this chunk's last frame (8-bit round trip; written
        to <out_dir>/references/<ldmk_end>/<idx>.png and read back when out_dir is given, exactly as the reference does)."""
        u8 = last_frame_to_uint8(fake)
        if out_dir is not None:
            from PIL import Image
            folder = os.path.join(out_dir, "references", str(ldmk_end))
            os.makedirs(folder, exist_ok=True)
            for idx in range(u8.shape[0]):
                Image.fromarray(u8[idx], "RGB").save(os.path.join(folder, f"{idx}.png"))
            # read back exactly what was just written, in clip order.  (sample.py:347-348 lists and sorts the folder: stale
            # PNGs of an earlier, larger batch would join the batch there, and "10.png" sorts before "2.png".)
            u8 = np.stack([np.asarray(Image.open(os.path.join(folder, f"{idx}.png")).convert("RGB")) for idx in range(u8.shape[0])], axis=0)
        dev = next(self.ae.parameters()).device
        ref = reference_from_uint8(u8, frames=self.ae.s).to(dev)
        return self.ae.extract(ref)[:, :, 0:self.n_xy]

    @torch.no_grad()
    def run_identity(self, chunks: Iterable[Sequence[torch.Tensor]], use_last_as_reference: bool = False,
                     out_dir: Optional[str] = None, noise_per_chunk: Optional[Sequence] = None, overlap: bool = False,
                     num_frames: Optional[int] = None, **sample_kw):
        """The per-identity loop (sample.py:305-432): chunks yield (x_ref, x, x_l, masked_x) in 0..255, [B,T,C,H,W].
        Chunks of one identity are sequential when chained (chunk k+1 needs chunk k's last frame): shard by identity.
        Returns the list of uint8 frame arrays [B, T, H, W, 3]; when out_dir is given writes frames/NNNN.png with the B clips
        of the batch side by side ([H, B*W, 3] per frame), the reference's grid_size=(k, 1) layout (sample.py:79-104).

        `overlap` (sample_crossID.py:185,343-348, the cross-identity script's `--overlap`): chunks start every T/2 frames instead of
        every T -- chunk `it` covers frames [it T/2, it T/2 + T), its frame files overwrite the second half of the chunk before, and
        with `use_last_as_reference` its reference is the last frame of the chunk that ENDED where it starts (the folder
        `references/<ldmk_srt>` there: chunk it - 2; the first two chunks find none and keep their own x_ref, sample_crossID.py:394-396).
        `num_frames` (sample_crossID.py:352-353): stop before the first chunk that starts past it."""
        results = []
        T = self.ae.s
        stride = T // 2 if overlap else T
        chained = {}                                  # frame index a chunk ended at -> image_cond made from its last frame
        for it, (x_ref, x, x_l, masked_x) in enumerate(chunks):
            ldmk_srt = it * stride
            if num_frames is not None and num_frames < ldmk_srt:
                break
            cond = self.conditioning(x_ref, x, x_l, masked_x)
            nz = noise_per_chunk[it] if noise_per_chunk is not None else None
            z, fake = self.sample_chunk(cond, image_cond=chained.get(ldmk_srt) if use_last_as_reference else None, noise=nz, **sample_kw)
            if use_last_as_reference:
                chained[ldmk_srt + T] = self.chained_image_cond(fake, out_dir, ldmk_srt + T)
                chained.pop(ldmk_srt - stride, None)  # (no longer reachable: keeps at most two latents alive)
            u8 = frames_to_uint8(fake)
            results.append(u8)
            if out_dir is not None:
                save_frames(ldmk_srt, np.concatenate(list(u8), axis=2), os.path.join(out_dir, "frames"))
        return results
