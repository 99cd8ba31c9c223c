"""Drop-in look-alike of the reference's sampler class, driving the HIP denoising loop.

Mirrors MToV/losses/ddpm.py:
  :121-193  DDPM.__init__  (same kwargs; called as DDPM(ema_model, channels=4, image_size=32,
                            sampling_timesteps=100, w=0.0).to(device), sample.py:239-245)
  :195-264  register_schedule (same buffer names, float64 math -> float32 buffers)
  :362-454  ddim_sample / ddim_sample_noised_start
  :456-484  sample(batch_size, cond, image_cond, context, return_intermediates, noised_start,
                   first_stage_model, ratio_, fix_noise)           [+ optional noise=, appended]
  :486-491  q_sample
The schedule and the per-step DDIM coefficients are host logic (this file, fp32 exactly as the
reference computes them); the loop body -- UNet forward, x0 prediction + clamp, x_{t_next} -- runs
in libmtv_hip.so (`mtv_ddim_sample`).  Training (`forward`/`p_losses`) and the ancestral sampler
(unreachable when sampling_timesteps < timesteps, and broken in the reference: ddpm.py:299) are
out of scope and raise.
"""
from __future__ import annotations

import ctypes as C
from functools import partial
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .unet import DiffusionWrapper, UNetModel


def make_beta_schedule(schedule: str, n_timestep: int, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3) -> np.ndarray:
    """The four schedule names `DDPM(beta_schedule=...)` accepts in the reference (ddpm.py:78-99), as fp64 arrays.
    The sampling path itself only ever builds "linear" (DDPM's default; sample.py:239-245 and the shipped configs never
    override it); the others are host-only arithmetic kept for constructor parity."""
    ramp = partial(torch.linspace, steps=n_timestep, dtype=torch.float64)
    if schedule == "linear":            # linear in sqrt(beta)
        betas = ramp(linear_start ** 0.5, linear_end ** 0.5).square()
    elif schedule == "sqrt_linear":     # (the reference's name for: linear in beta)
        betas = ramp(linear_start, linear_end)
    elif schedule == "sqrt":
        betas = ramp(linear_start, linear_end).sqrt()
    elif schedule == "cosine":          # alpha_bar(t) = cos^2((t/T + s) / (1 + s) * pi/2), normalised to alpha_bar(0) = 1
        grid = torch.arange(n_timestep + 1, dtype=torch.float64) / n_timestep
        abar = torch.cos((grid + cosine_s) / (1 + cosine_s) * (np.pi / 2)).square()
        abar = abar / abar[0]
        betas = (1 - abar[1:] / abar[:-1]).clamp(0, 0.999)
    else:
        raise ValueError(f"schedule '{schedule}' unknown.")
    return betas.numpy()


def ddim_time_pairs(total_timesteps: int, sampling_timesteps: int) -> List[Tuple[int, int]]:
    # ddpm.py:371-375
    times = torch.linspace(-1, total_timesteps - 1, steps=sampling_timesteps + 1)
    times = list(reversed(times.int().tolist()))
    return list(zip(times[:-1], times[1:]))


def ddim_step_table(alphas_cumprod: torch.Tensor, sqrt_recip: torch.Tensor, sqrt_recipm1: torch.Tensor,
                    pairs: Sequence[Tuple[int, int]], eta: float):
    """Per-step scalars exactly as ddpm.py:390-394 evaluates them (0-dim fp32 tensor arithmetic).
    Returns a ctypes array of MtvDdimStep; noise_index i means noise[i] of the in-loop draws."""
    ac = alphas_cumprod.detach().float().cpu()
    sr = sqrt_recip.detach().float().cpu()
    srm1 = sqrt_recipm1.detach().float().cpu()
    steps = (_lib.MtvDdimStep * len(pairs))()
    k = 0
    for i, (time, time_next) in enumerate(pairs):
        s = steps[i]
        s.t = int(time)
        s.sqrt_recip_ac = float(sr[time])
        s.sqrt_recipm1_ac = float(srm1[time])
        if time_next < 0:
            s.last, s.sqrt_ac_next, s.c, s.sigma, s.noise_index = 1, 0.0, 0.0, 0.0, -1
            continue
        alpha, alpha_next = ac[time], ac[time_next]
        sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
        c = (1 - alpha_next - sigma ** 2).sqrt()
        s.last = 0
        s.sqrt_ac_next = float(alpha_next.sqrt())
        s.c = float(c)
        s.sigma = float(sigma)
        s.noise_index = k
        k += 1
    return steps, k


class DDPM(nn.Module):
    def __init__(self, model, timesteps=1000, beta_schedule="linear", loss_type="l2", ckpt_path=None, ignore_keys=[],
                 load_only_unet=False, monitor="val/loss", use_ema=True, first_stage_key="image", image_size=256,
                 channels=3, log_every_t=200, clip_denoised=True, linear_start=0.0015, linear_end=0.0195,
                 cosine_s=8e-3, given_betas=None, original_elbo_weight=0.0, v_posterior=0.0, l_simple_weight=1.0,
                 conditioning_key=None, parameterization="eps", use_positional_encodings=False, learn_logvar=False,
                 logvar_init=0.0, sampling_timesteps=1000, ddim_sampling_eta=1.0, w=1.0, first_stage_model=None):
        super().__init__()
        assert parameterization in ["eps", "x0"], 'currently only supporting "eps" and "x0"'
        if parameterization != "eps":
            raise NotImplementedError("the shipped MToV model is eps-parameterised; x0 is not built")
        self.parameterization = parameterization
        self.clip_denoised = clip_denoised
        self.log_every_t = log_every_t
        self.first_stage_key = first_stage_key
        self.channels = channels
        self.model = model
        self.use_ema = use_ema
        self.v_posterior = v_posterior
        self.original_elbo_weight = original_elbo_weight
        self.l_simple_weight = l_simple_weight
        # the reference hard-wires 2048 = 32*32 + 2*16*32 (ddpm.py:162); here it follows the model geometry
        um = self._unet()
        self.image_size = (um.image_size ** 2 + 2 * um.frames * um.image_size) if um is not None else 2048
        self.register_schedule(given_betas=given_betas, beta_schedule=beta_schedule, timesteps=timesteps,
                               linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        self.loss_type = loss_type
        self.learn_logvar = learn_logvar
        self.logvar = torch.full(fill_value=logvar_init, size=(self.num_timesteps,))
        self.sampling_timesteps = sampling_timesteps if sampling_timesteps is not None else timesteps
        assert self.sampling_timesteps <= timesteps
        self.is_ddim_sampling = self.sampling_timesteps < timesteps
        self.ddim_sampling_eta = ddim_sampling_eta
        self.w = w
        self.first_stage_model = first_stage_model

    def _unet(self) -> Optional[UNetModel]:
        m = self.model
        if isinstance(m, DiffusionWrapper):
            m = m.diffusion_model
        return m if isinstance(m, UNetModel) else None

    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        betas = given_betas if given_betas is not None else make_beta_schedule(
            beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        alphas = 1.0 - betas
        alphas_cumprod = np.cumprod(alphas, axis=0)
        alphas_cumprod_prev = np.append(1.0, alphas_cumprod[:-1])
        (timesteps,) = betas.shape
        self.num_timesteps = int(timesteps)
        self.linear_start, self.linear_end = linear_start, linear_end
        f32 = partial(torch.tensor, dtype=torch.float32)
        self.register_buffer("betas", f32(betas))
        self.register_buffer("alphas_cumprod", f32(alphas_cumprod))
        self.register_buffer("alphas_cumprod_prev", f32(alphas_cumprod_prev))
        self.register_buffer("sqrt_alphas_cumprod", f32(np.sqrt(alphas_cumprod)))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", f32(np.sqrt(1.0 - alphas_cumprod)))
        self.register_buffer("log_one_minus_alphas_cumprod", f32(np.log(1.0 - alphas_cumprod)))
        self.register_buffer("sqrt_recip_alphas_cumprod", f32(np.sqrt(1.0 / alphas_cumprod)))
        self.register_buffer("sqrt_recipm1_alphas_cumprod", f32(np.sqrt(1.0 / alphas_cumprod - 1)))
        pv = (1 - self.v_posterior) * betas * (1.0 - alphas_cumprod_prev) / (1.0 - alphas_cumprod) + self.v_posterior * betas
        self.register_buffer("posterior_variance", f32(pv))
        self.register_buffer("posterior_log_variance_clipped", f32(np.log(np.maximum(pv, 1e-20))))
        self.register_buffer("posterior_mean_coef1", f32(betas * np.sqrt(alphas_cumprod_prev) / (1.0 - alphas_cumprod)))
        self.register_buffer("posterior_mean_coef2", f32((1.0 - alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - alphas_cumprod)))
        lvlb = self.betas ** 2 / (2 * self.posterior_variance * f32(alphas) * (1 - self.alphas_cumprod))
        lvlb[0] = lvlb[1]
        self.register_buffer("lvlb_weights", lvlb, persistent=False)

    # ------------------------------------------------------------------ host-side pieces
    def q_sample(self, x_start, t, noise=None):
        # ddpm.py:486-491 (one-time host-side op before the loop)
        noise = torch.randn_like(x_start) if noise is None else noise
        sh = (t.shape[0],) + (1,) * (x_start.dim() - 1)
        return (self.sqrt_alphas_cumprod.gather(-1, t).reshape(sh) * x_start
                + self.sqrt_one_minus_alphas_cumprod.gather(-1, t).reshape(sh) * noise)

    def predict_start_from_noise(self, x_t, t, noise):
        sh = (t.shape[0],) + (1,) * (x_t.dim() - 1)
        return (self.sqrt_recip_alphas_cumprod.gather(-1, t).reshape(sh) * x_t
                - self.sqrt_recipm1_alphas_cumprod.gather(-1, t).reshape(sh) * noise)

    def _time_pairs(self, ratio_: Optional[float] = None):
        pairs = ddim_time_pairs(self.num_timesteps, self.sampling_timesteps)
        if ratio_ is not None:
            pairs = pairs[int(len(pairs) * (1 - ratio_)):]          # ddpm.py:430
        return pairs

    # ------------------------------------------------------------------ the hot loop
    @torch.no_grad()
    def _run_ddim(self, x_init, cond, image_cond, pairs, noise, strict=None):
        um = self._unet()
        if um is None:
            raise TypeError("DDPM.sample drives the HIP denoiser: `model` must be moditalker_amd's "
                            "DiffusionWrapper(UNetModel) (no PyTorch fallback loop)")
        dev = self.betas.device
        # everything below goes down as raw pointers: validate what the reference's torch.cat / broadcasting would
        B = um.check_inputs(x_init, cond, image_cond)
        if self.channels != 4 or um.out_channels != 4:
            raise ValueError("the DDIM loop needs channels == out_channels == 4 (eps has the shape of x)")
        ctx = um.hip_context(dev, B)
        steps, n_draws = ddim_step_table(self.alphas_cumprod, self.sqrt_recip_alphas_cumprod,
                                         self.sqrt_recipm1_alphas_cumprod, pairs, self.ddim_sampling_eta)
        if noise is None:
            # the reference draws one randn_like per non-final step from the global generator (ddpm.py:396,447)
            noise = [torch.randn(x_init.shape, device=dev) for _ in range(n_draws)]
        if isinstance(noise, (list, tuple)):
            noise = torch.stack(list(noise)[:n_draws]) if n_draws else torch.empty(0, *x_init.shape, device=dev)
        noise = noise.to(device=dev, dtype=torch.float32).contiguous()
        if noise.shape[0] < n_draws:
            raise ValueError(f"need {n_draws} in-loop noise draws, got {noise.shape[0]}")
        if n_draws and tuple(noise.shape[1:]) != tuple(x_init.shape):
            raise ValueError(f"every noise draw must have the shape of x {tuple(x_init.shape)}; got {tuple(noise.shape[1:])}")
        x = x_init.to(device=dev, dtype=torch.float32).contiguous().clone()
        cf = cond.to(device=dev, dtype=torch.float32).contiguous()
        icf = image_cond.to(device=dev, dtype=torch.float32).contiguous()
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(_lib.load().mtv_ddim_sample(ctx, x.data_ptr(), cf.data_ptr(), icf.data_ptr(), icf.shape[2],
                                                   noise.data_ptr() if n_draws else None, int(noise.shape[0]),
                                                   steps, len(pairs), B, C.c_void_p(stream)), "mtv_ddim_sample")
            um.after_call(dev, strict)
        return x

    @torch.no_grad()
    def ddim_sample(self, shape, cond, image_cond, context=None, clip_denoised=True, noise=None, strict=None):
        # ddpm.py:362-404; noise[0] = x_T, noise[1:] = in-loop draws
        dev = self.betas.device
        if noise is None:
            x_T = torch.randn(shape, device=dev)
            rest = None
        else:
            x_T, rest = noise[0], noise[1:]
        assert clip_denoised, "the reference always clamps x0 (ddpm.py:384)"
        return self._run_ddim(x_T, cond, image_cond, self._time_pairs(), rest, strict=strict)

    @torch.no_grad()
    def ddim_sample_noised_start(self, shape, x_start, cond, image_cond, context=None, clip_denoised=True,
                                 ratio_=None, fixed_noise=False, noise=None, strict=None):
        # ddpm.py:407-454
        t = torch.tensor([int(self.num_timesteps * ratio_)], device=x_start.device).long()
        if noise is not None:
            q_noise, rest = noise[0], noise[1:]
        else:
            if fixed_noise:
                torch.manual_seed(1004)
            q_noise, rest = torch.randn_like(x_start).contiguous(), None
        x_noisy = self.q_sample(x_start=x_start, t=t, noise=q_noise.to(x_start.device))
        return self._run_ddim(x_noisy, cond, image_cond, self._time_pairs(ratio_), rest, strict=strict)

    @torch.no_grad()
    def sample(self, batch_size=16, cond=None, image_cond=None, context=None, return_intermediates=False,
               noised_start=None, first_stage_model=None, ratio_=None, fix_noise=False, noise=None, strict=None):
        """ddpm.py:456-484.  Returns [batch_size, channels, L] fp32 on the module's device.
        `noise` (appended kwarg): explicit list/tensor of N(0,1) draws in the reference's draw order
        (initial x_T or q_sample noise first, then one per non-final step).
        `strict` (appended kwarg; default: the MTV_STRICT environment variable, else off): synchronise the stream and raise MtvError if an
        in-launch hand-off of this call timed out (include/mtv_hip.h mtv_check_fault) -- the reference's torch ops raise synchronously
        (unet.py:995-1117); without it the fault surfaces at the caller's next call into the library or at `UNetModel.check_fault()`.
        `return_intermediates` / `first_stage_model`: accepted and unused, exactly like the reference's DDIM branch
        (ddpm.py:473-482 forwards return_intermediates only to the unreachable p_sample_loop)."""
        shape = (batch_size, self.channels, self.image_size)
        if not self.is_ddim_sampling:
            raise NotImplementedError("ancestral p_sample_loop is unreachable in the reference's sampling scripts "
                                      "(sampling_timesteps < timesteps) and is not built")
        if noised_start is not None:
            return self.ddim_sample_noised_start(shape, noised_start, cond, image_cond, context, ratio_=ratio_,
                                                 fixed_noise=fix_noise, noise=noise, strict=strict)
        return self.ddim_sample(shape, cond, image_cond, context, noise=noise, strict=strict)

    def forward(self, *a, **k):
        raise NotImplementedError("training (p_losses) is outside the MI355X hot path (SURVEY.md section 8)")
