"""moditalker_amd -- MI355X-native denoising loop of MoDiTalker's MToV stage (tri-plane UNet forward +
DDIM update) behind the reference's own Python signatures.  Host code is Python on PyTorch-ROCm
(device memory, streams, RCCL); the step itself is hand-written gfx950 HIP in csrc/ behind the
C ABI of include/mtv_hip.h.  See DESIGN.md."""
from ._lib import MtvError  # noqa: F401
from .ddpm import DDPM, ddim_time_pairs, make_beta_schedule  # noqa: F401
from .unet import DiffusionWrapper, UNetModel  # noqa: F401
from .autoencoder import ViTAutoencoder  # noqa: F401
from .cross_attention import CrossAttention  # noqa: F401

BASE_AE_DDCONFIG = dict(  # MToV/configs/autoencoder/base.yaml:12-24 (embed_dim: 4)
    double_z=False, channels=384, resolution=256, timesteps=16, skip=1, in_channels=3, out_ch=3, num_res_blocks=2,
    attn_resolutions=[], splits=1,
)

BASE_UNET_CONFIG = dict(  # MToV/configs/latent-diffusion/base.yaml:27-38
    image_size=32, in_channels=4, out_channels=4, model_channels=128, attention_resolutions=[4, 2, 1],
    num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_scale_shift_norm=True,
    resblock_updown=True, cond_model=False,
)
