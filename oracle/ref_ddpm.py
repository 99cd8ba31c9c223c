"""CPU oracle for the MToV DDPM schedule and DDIM sampling loop -- TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this.
Restates, in numpy (schedule, float64 like the reference) and PyTorch CPU fp32 (loop):
  MToV/losses/ddpm.py:79-81     make_beta_schedule("linear")
  MToV/losses/ddpm.py:195-264   DDPM.register_schedule
  MToV/losses/ddpm.py:278-282   predict_start_from_noise
  MToV/losses/ddpm.py:338-360   model_predictions (eps parameterisation, clamp, eps NOT re-derived)
  MToV/losses/ddpm.py:362-404   ddim_sample
  MToV/losses/ddpm.py:407-454   ddim_sample_noised_start
  MToV/losses/ddpm.py:486-491   q_sample
The reference draws noise from torch's global generator inside the loop; here noise is an
explicit list (draw order: initial x_T / q_sample noise first, then one draw per non-final step),
which is how the golden vectors were captured (tests/golden/make_golden.py).
Pinned against the imported reference by that script; see tests/test_oracle_golden.py.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch


def linear_beta_schedule(n_timestep: int = 1000, linear_start: float = 0.0015,
                         linear_end: float = 0.0195) -> np.ndarray:
    # ddpm.py:79-81 -- torch.linspace(sqrt(a), sqrt(b), n, float64) ** 2
    return torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64).numpy() ** 2


def schedule_buffers(n_timestep: int = 1000, linear_start: float = 0.0015,
                     linear_end: float = 0.0195) -> dict:
    """float32 buffers exactly as DDPM.register_schedule registers them (ddpm.py:228-239)."""
    betas = linear_beta_schedule(n_timestep, linear_start, linear_end)
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    f32 = lambda a: torch.tensor(a, dtype=torch.float32)
    return dict(
        betas=f32(betas),
        alphas_cumprod=f32(ac),
        alphas_cumprod_prev=f32(np.append(1.0, ac[:-1])),
        sqrt_alphas_cumprod=f32(np.sqrt(ac)),
        sqrt_one_minus_alphas_cumprod=f32(np.sqrt(1.0 - ac)),
        sqrt_recip_alphas_cumprod=f32(np.sqrt(1.0 / ac)),
        sqrt_recipm1_alphas_cumprod=f32(np.sqrt(1.0 / ac - 1)),
    )


def ddim_time_pairs(total_timesteps: int, sampling_timesteps: int) -> List[Tuple[int, int]]:
    # ddpm.py:371-375
    times = torch.linspace(-1, total_timesteps - 1, steps=sampling_timesteps + 1)
    times = list(reversed(times.int().tolist()))
    return list(zip(times[:-1], times[1:]))


ModelFn = Callable[[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor], torch.Tensor]


@torch.no_grad()
def ddim_sample(model: ModelFn, cond: torch.Tensor, image_cond: torch.Tensor, noise: Sequence[torch.Tensor],
                sampling_timesteps: int, total_timesteps: int = 1000, eta: float = 1.0,
                noised_start: Optional[torch.Tensor] = None, ratio_: Optional[float] = None,
                buffers: Optional[dict] = None, trajectory: Optional[list] = None) -> torch.Tensor:
    """model(x, cond, image_cond, t[B] int64) -> eps.  noise[0] is x_T (or the q_sample noise when
    `noised_start` is given); noise[1+i] is the draw of the i-th non-final step."""
    buf = buffers or schedule_buffers(total_timesteps)
    ac = buf["alphas_cumprod"]
    pairs = ddim_time_pairs(total_timesteps, sampling_timesteps)
    if noised_start is not None:                       # ddpm.py:422-430
        t0 = int(total_timesteps * ratio_)
        img = buf["sqrt_alphas_cumprod"][t0] * noised_start + buf["sqrt_one_minus_alphas_cumprod"][t0] * noise[0]
        pairs = pairs[int(len(pairs) * (1 - ratio_)):]
    else:
        img = noise[0].clone()                         # ddpm.py:378
    batch = img.shape[0]
    k = 1
    out = None
    for time, time_next in pairs:
        if trajectory is not None:
            trajectory.append(img.clone())     # x_t fed to the model at this step
        t = torch.full((batch,), time, dtype=torch.long)
        eps = model(img, cond, image_cond, t)
        # ddpm.py:278-282, 346-351
        x_start = buf["sqrt_recip_alphas_cumprod"][time] * img - buf["sqrt_recipm1_alphas_cumprod"][time] * eps
        x_start.clamp_(-1.0, 1.0)
        if time_next < 0:                              # ddpm.py:386-388
            out = x_start
            continue
        alpha, alpha_next = ac[time], ac[time_next]
        sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
        c = (1 - alpha_next - sigma ** 2).sqrt()
        img = x_start * alpha_next.sqrt() + c * eps + sigma * noise[k]      # ddpm.py:398
        k += 1
    if trajectory is not None:
        trajectory.append(out.clone())
    return out


def num_noise_draws(sampling_timesteps: int, ratio_: Optional[float] = None) -> int:
    n = sampling_timesteps
    if ratio_ is not None:
        n = n - int(n * (1 - ratio_))
    return n  # one initial draw + (n-1) in-loop draws
