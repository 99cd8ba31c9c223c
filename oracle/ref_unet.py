"""CPU oracle for the MToV tri-plane UNet forward -- TEST INFRASTRUCTURE ONLY.

This file is the parity checker for the HIP path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; nothing under
`moditalker_amd/` does.  It is never the thing measured or shipped.

It restates, op for op in plain PyTorch CPU fp32, what the reference computes in
  MToV/models/ddpm/unet.py:995-1117   UNetModel.forward          (the per-step graph)
  MToV/models/ddpm/unet.py:178-207    ResBlock._forward
  MToV/models/ddpm/unet.py:248-254    AttentionBlock._forward      (2-D, per plane)
  MToV/models/ddpm/unet.py:295-300    AttentionBlock1D._forward    (whole tri-plane)
  MToV/models/ddpm/unet.py:312-326    QKVAttentionLegacy.forward
  MToV/models/ddpm/diffusionmodules.py:108-128  timestep_embedding
  MToV/models/ddpm/diffusionmodules.py:156-173  GroupNorm32 (32 groups, eps 1e-5)
as a *function of a state_dict* (the reference's 804-key `DiffusionWrapper.state_dict()`
layout, prefix `diffusion_model.` optional) and of the geometry (R, T), which the reference
hard-wires to (32, 16) (unet.py:1027-1029).  Same ATen ops as the reference (conv2d, conv1d,
group_norm, linear, einsum->bmm with a materialised LxL score matrix, softmax, avg_pool2d,
nearest interpolate, silu), three sequential plane passes.

Pinning: validated against the imported reference at (R,T)=(32,16) by
`tests/golden/make_golden.py` (run in the build container, where /root/reference exists); the
golden vectors it wrote are re-checked by `tests/test_oracle_golden.py` on every run.
Geometries the reference cannot execute (R=8,T=4; R=64,T=16) are pinned only by this
restatement.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

BASE_CFG = dict(  # MToV/configs/latent-diffusion/base.yaml:27-38
    image_size=32, in_channels=4, out_channels=4, model_channels=128,
    attention_resolutions=[4, 2, 1], num_res_blocks=2, channel_mult=[1, 2, 4, 4],
    num_heads=8, use_scale_shift_norm=True, resblock_updown=True, cond_model=False,
)


# --------------------------------------------------------------------------------------
# block structure, derived the way UNetModel.__init__ builds it (unet.py:710-975)
# --------------------------------------------------------------------------------------
def block_structure(cfg: dict) -> dict:
    """Returns the module layout as nested lists of ('res', cin, cout, updown) / ('attn', c)."""
    mc = cfg["model_channels"]
    mult = list(cfg["channel_mult"])
    nrb = cfg["num_res_blocks"]
    att = set(cfg["attention_resolutions"])
    assert cfg.get("resblock_updown", False), "oracle restates the resblock_updown=True path only"
    inputs: List[list] = [[("conv", 16, mc)]]       # unet.py:714 (hard-wired 16 in-channels)
    in_attn: List[Optional[int]] = [None]           # unet.py:719 nn.Identity
    chans = [mc]
    ch, ds = mc, 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            layers = [("res", ch, m * mc, None)]
            ch = m * mc
            if ds in att:
                layers.append(("attn", ch))
            inputs.append(layers)
            in_attn.append(ch)
            chans.append(ch)
        if level != len(mult) - 1:
            inputs.append([("res", ch, ch, "down")])
            in_attn.append(ch)
            chans.append(ch)
            ds *= 2
    middle = [("res", ch, ch, None), ("attn", ch), ("res", ch, ch, None)]
    outputs: List[list] = []
    out_attn: List[int] = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, mc * m, None)]
            ch = mc * m
            if ds in att:
                layers.append(("attn", ch))
            if level and i == nrb:
                layers.append(("res", ch, ch, "up"))
                ds //= 2
            outputs.append(layers)
            out_attn.append(ch)
    return dict(inputs=inputs, in_attn=in_attn, middle=middle, mid_attn=ch,
                outputs=outputs, out_attn=out_attn, final=ch)


def _strip(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    p = "diffusion_model."
    if any(k.startswith(p) for k in sd):
        return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
    return dict(sd)


# --------------------------------------------------------------------------------------
# leaf ops
# --------------------------------------------------------------------------------------
def timestep_embedding(t: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    # diffusionmodules.py:108-128 -- cos first, then sin
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _gn(x, sd, name):
    return F.group_norm(x.float(), 32, sd[name + ".weight"], sd[name + ".bias"], eps=1e-5)


def _resblock(x, emb, sd, pre, cin, cout, updown, scale_shift):
    # unet.py:178-207
    h = F.silu(_gn(x, sd, pre + "in_layers.0"))
    if updown == "down":                              # unet.py:179-184, Downsample.op :594
        h = F.avg_pool2d(h, 2, 2)
        x = F.avg_pool2d(x, 2, 2)
    elif updown == "up":                              # Upsample.forward :554
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    h = F.conv2d(h, sd[pre + "in_layers.2.weight"], sd[pre + "in_layers.2.bias"], padding=1)
    emb_out = F.linear(F.silu(emb), sd[pre + "emb_layers.1.weight"], sd[pre + "emb_layers.1.bias"])
    emb_out = emb_out[:, :, None, None]
    if scale_shift:
        scale, shift = torch.chunk(emb_out, 2, dim=1)
        h = _gn(h, sd, pre + "out_layers.0") * (1 + scale) + shift
        h = F.silu(h)
    else:
        h = h + emb_out
        h = F.silu(_gn(h, sd, pre + "out_layers.0"))
    h = F.conv2d(h, sd[pre + "out_layers.3.weight"], sd[pre + "out_layers.3.bias"], padding=1)
    if cin == cout:
        skip = x                                      # unet.py:162-163
    else:
        skip = F.conv2d(x, sd[pre + "skip_connection.weight"], sd[pre + "skip_connection.bias"])
    return skip + h


def qkv_attention_legacy(qkv: torch.Tensor, n_heads: int) -> torch.Tensor:
    # unet.py:312-326 -- head-major, then q|k|v inside each head; d^-1/4 on q and on k
    bs, width, length = qkv.shape
    ch = width // (3 * n_heads)
    q, k, v = qkv.reshape(bs * n_heads, ch * 3, length).split(ch, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v)
    return a.reshape(bs, -1, length)


def _attn(x3, sd, pre, heads):
    # unet.py:248-254 / 295-300 on a [B, C, L] tensor
    qkv = F.conv1d(_gn(x3, sd, pre + "norm"), sd[pre + "qkv.weight"], sd[pre + "qkv.bias"])
    h = qkv_attention_legacy(qkv, heads)
    h = F.conv1d(h, sd[pre + "proj_out.weight"], sd[pre + "proj_out.bias"])
    return x3 + h


def _run_layers(layers, x, emb, sd, pre, heads, scale_shift):
    for j, ly in enumerate(layers):
        p = f"{pre}{j}."
        if ly[0] == "conv":
            x = F.conv2d(x, sd[p + "weight"], sd[p + "bias"], padding=1)
        elif ly[0] == "res":
            x = _resblock(x, emb, sd, p, ly[1], ly[2], ly[3], scale_shift)
        else:
            b, c, hh, ww = x.shape
            x = _attn(x.reshape(b, c, -1), sd, p, heads).reshape(b, c, hh, ww)
    return x


# --------------------------------------------------------------------------------------
# the per-step graph
# --------------------------------------------------------------------------------------
@torch.no_grad()
def unet_forward(sd: Dict[str, torch.Tensor], cfg: dict, x: torch.Tensor, cond: torch.Tensor,
                 image_cond: torch.Tensor, timesteps: torch.Tensor, res: int = 32, frames: int = 16,
                 taps: Optional[dict] = None) -> torch.Tensor:
    """x [B,4,L], cond [B,8,L], image_cond [B,4,>=R*R], timesteps [B] int64 -> eps [B,4,L].

    L = R*R + 2*T*R.  `taps`, if given, receives intermediate [B,C,L'] tensors keyed
    'in{i}', 'mid', 'out{i}' (post cross-plane attention) for bisecting.
    """
    sd = _strip(sd)
    st = block_structure(cfg)
    heads = cfg["num_heads"]
    ss = cfg.get("use_scale_shift_norm", False)
    R, T = res, frames
    L = R * R + 2 * T * R
    assert x.shape[-1] == L and cond.shape[-1] == L, (x.shape, cond.shape, L)

    emb = timestep_embedding(timesteps, cfg["model_channels"])                 # unet.py:1011
    emb = F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])    # unet.py:1012
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])

    h = x.float()
    # unet.py:1022-1025: only the xy plane of image_cond is kept, yt/xt are zeros
    pad = torch.zeros(x.shape[0], image_cond.shape[1], L - R * R)
    ic = torch.cat([image_cond[:, :, : R * R].float(), pad], dim=2)
    h = torch.cat([h, cond.float(), ic], dim=1)

    def split(hh, r, t):
        b, c = hh.shape[:2]
        return (hh[:, :, : r * r].reshape(b, c, r, r),
                hh[:, :, r * r: r * (r + t)].reshape(b, c, t, r),
                hh[:, :, r * (r + t): r * (r + 2 * t)].reshape(b, c, t, r))

    def join(planes):
        return torch.cat([p.reshape(p.shape[0], p.shape[1], -1) for p in planes], dim=-1)

    planes = split(h, R, T)
    skips = []
    for i, layers in enumerate(st["inputs"]):                                   # unet.py:1031-1053
        planes = [_run_layers(layers, p, emb, sd, f"input_blocks.{i}.", heads, ss) for p in planes]
        r, t = planes[0].shape[-2], planes[2].shape[-2]
        hcat = join(planes)
        if st["in_attn"][i] is not None:
            hcat = _attn(hcat, sd, f"input_attns.{i}.", heads)
        if taps is not None:
            taps[f"in{i}"] = hcat.clone()
        planes = split(hcat, r, t)
        skips.append(planes)

    planes = [_run_layers(st["middle"], p, emb, sd, "middle_block.", heads, ss) for p in planes]
    r, t = planes[0].shape[-2], planes[2].shape[-2]                             # unet.py:1055-1071
    hcat = _attn(join(planes), sd, "mid_attn.", heads)
    if taps is not None:
        taps["mid"] = hcat.clone()
    planes = split(hcat, r, t)

    for i, layers in enumerate(st["outputs"]):                                  # unet.py:1075-1101
        sk = skips.pop()
        planes = [torch.cat([p, s], dim=1) for p, s in zip(planes, sk)]
        planes = [_run_layers(layers, p, emb, sd, f"output_blocks.{i}.", heads, ss) for p in planes]
        r, t = planes[0].shape[-2], planes[2].shape[-2]
        hcat = _attn(join(planes), sd, f"output_attns.{i}.", heads)
        if taps is not None:
            taps[f"out{i}"] = hcat.clone()
        planes = split(hcat, r, t)

    outs = []
    for p in planes:                                                            # unet.py:1103-1112
        p = F.silu(_gn(p, sd, "out.0"))
        outs.append(F.conv2d(p, sd["out.2.weight"], sd["out.2.bias"], padding=1))
    return join(outs).type(x.dtype)


def used_keys(cfg: dict) -> List[str]:
    """State-dict keys (without prefix) the forward reads -- everything except output_bg_*."""
    st = block_structure(cfg)
    keys: List[str] = []
    for nm in ("time_embed.0", "time_embed.2"):
        keys += [nm + ".weight", nm + ".bias"]

    def layer_keys(layers, pre):
        for j, ly in enumerate(layers):
            p = f"{pre}{j}."
            if ly[0] == "conv":
                yield from (p + "weight", p + "bias")
            elif ly[0] == "res":
                for nm in ("in_layers.0", "in_layers.2", "emb_layers.1", "out_layers.0", "out_layers.3"):
                    yield from (p + nm + ".weight", p + nm + ".bias")
                if ly[1] != ly[2]:
                    yield from (p + "skip_connection.weight", p + "skip_connection.bias")
            else:
                for nm in ("norm", "qkv", "proj_out"):
                    yield from (p + nm + ".weight", p + nm + ".bias")

    def attn_keys(pre):
        for nm in ("norm", "qkv", "proj_out"):
            yield from (pre + nm + ".weight", pre + nm + ".bias")

    for i, layers in enumerate(st["inputs"]):
        keys += list(layer_keys(layers, f"input_blocks.{i}."))
        if st["in_attn"][i] is not None:
            keys += list(attn_keys(f"input_attns.{i}."))
    keys += list(layer_keys(st["middle"], "middle_block."))
    keys += list(attn_keys("mid_attn."))
    for i, layers in enumerate(st["outputs"]):
        keys += list(layer_keys(layers, f"output_blocks.{i}."))
        keys += list(attn_keys(f"output_attns.{i}."))
    keys += ["out.0.weight", "out.0.bias", "out.2.weight", "out.2.bias"]
    return keys
