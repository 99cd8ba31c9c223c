"""CPU oracle for the RGB / landmark autoencoder steps either side of the denoising loop.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg may import it; the product
never does).  A PyTorch CPU restatement, op for op, of

  MToV/models/autoencoder/autoencoder_vit.py:257-275   ViTAutoencoder.decode_from_sample
  MToV/models/autoencoder/autoencoder_vit.py:212-255   ViTAutoencoder.extract
  MToV/models/autoencoder/vit_modules.py:7-64          rotate_every_two / apply_rot_emb / AxialRotaryEmbedding / RotaryEmbedding
  MToV/models/autoencoder/vit_modules.py:88-146        GEGLU / FeedForward / attn / Attention
  MToV/models/autoencoder/vit_modules.py:150-303       TimeSformerEncoder / TimeSformerDecoder
  MToV/models/autoencoder/autoencoder_vit.py:15-84     PreNorm / FeedForward / Attention / Transformer (the *_quant_attn stacks)

as functions of a state_dict (the reference module's key names).  Pinned: tests/golden/make_golden_ae.py imports the
reference ViTAutoencoder, fills it with the arithmetic recipe of moditalker_amd/filler.py and stores its outputs in
tests/golden/ae.npz; tests/test_oracle_golden.py re-checks this file against them on every CPU run.

Geometry is the shipped one (configs/autoencoder/base.yaml): resolution 256, 16 frames, patch 8 -> 32x32 sites per
frame, channels 384, 8 heads x 64, depth 8, embed_dim 4; the functions take (res, frames) so the tests can also use
smaller clips.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

HEADS, DIM_HEAD = 8, 64          # vit_modules.py:162-163,243-244 defaults, never overridden (autoencoder_vit.py:110-116)
DEPTH, PATCH = 8, 8              # autoencoder_vit.py:105-116


def rotate_every_two(x):
    # vit_modules.py:7-11: (x0, x1, x2, x3, ...) -> (-x1, x0, -x3, x2, ...)
    x = x.reshape(*x.shape[:-1], x.shape[-1] // 2, 2)
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).reshape(*x.shape[:-2], -1)


def frame_rot_emb(inv_freqs: torch.Tensor, n: int):
    # RotaryEmbedding.forward (vit_modules.py:57-62): sin/cos [1, n, 64]
    seq = torch.arange(n)
    freqs = torch.einsum("i, j -> i j", seq, inv_freqs)
    freqs = torch.cat((freqs, freqs), dim=-1)[None]
    return freqs.sin(), freqs.cos()


def image_rot_emb(scales: torch.Tensor, h: int, w: int):
    # AxialRotaryEmbedding.forward (vit_modules.py:29-49): sin/cos [1, h*w, 64]
    scales = scales[None]
    h_seq = torch.linspace(-1.0, 1.0, steps=h).unsqueeze(-1) * scales * math.pi
    w_seq = torch.linspace(-1.0, 1.0, steps=w).unsqueeze(-1) * scales * math.pi
    x_sinu = h_seq[:, None, :].expand(h, w, -1)
    y_sinu = w_seq[None, :, :].expand(h, w, -1)
    sin = torch.cat((x_sinu.sin(), y_sinu.sin()), dim=-1).reshape(h * w, -1)
    cos = torch.cat((x_sinu.cos(), y_sinu.cos()), dim=-1).reshape(h * w, -1)
    sin = sin.repeat_interleave(2, dim=-1)[None]
    cos = cos.repeat_interleave(2, dim=-1)[None]
    return sin, cos


def _vit_attention(sd, pre, x, mode, f, n, rot):
    """Attention.forward (vit_modules.py:119-146) for einops_to = '(b n) f d' (mode 'time') or '(b f) n d' ('space')."""
    b = x.shape[0]
    h = HEADS
    q, k, v = F.linear(x, sd[pre + "to_qkv.weight"]).chunk(3, dim=-1)

    def heads(t):                                   # 'b n (h d) -> (b h) n d'
        return t.reshape(b, f * n, h, DIM_HEAD).permute(0, 2, 1, 3).reshape(b * h, f * n, DIM_HEAD)

    q, k, v = heads(q), heads(k), heads(v)
    q = q * DIM_HEAD ** -0.5

    def fold(t):                                    # 'b (f n) d -> (b n) f d' or '(b f) n d'
        t = t.reshape(b * h, f, n, DIM_HEAD)
        return t.permute(0, 2, 1, 3).reshape(b * h * n, f, DIM_HEAD) if mode == "time" else t.reshape(b * h * f, n, DIM_HEAD)

    q, k, v = fold(q), fold(k), fold(v)
    sin, cos = rot
    q = q * cos + rotate_every_two(q) * sin         # apply_rot_emb (vit_modules.py:13-19): rot_dim == DIM_HEAD
    k = k * cos + rotate_every_two(k) * sin
    sim = torch.einsum("b i d, b j d -> b i j", q, k)
    out = torch.einsum("b i j, b j d -> b i d", sim.softmax(dim=-1), v)
    if mode == "time":
        out = out.reshape(b * h, n, f, DIM_HEAD).permute(0, 2, 1, 3)
    out = out.reshape(b, h, f * n, DIM_HEAD).permute(0, 2, 1, 3).reshape(b, f * n, h * DIM_HEAD)
    return F.linear(out, sd[pre + "to_out.0.weight"], sd[pre + "to_out.0.bias"])


def timesformer_layers(sd: Dict[str, torch.Tensor], pre: str, x: torch.Tensor, f: int, hp: int, wp: int) -> torch.Tensor:
    """The layer stack shared by TimeSformerEncoder/Decoder.forward (vit_modules.py:225-234, 294-303). x [b, f*hp*wp, dim]."""
    n = hp * wp
    frame_rot = frame_rot_emb(sd[pre + "frame_rot_emb.inv_freqs"], f)
    image_rot = image_rot_emb(sd[pre + "image_rot_emb.scales"], hp, wp)
    dim = x.shape[-1]

    def ln(t, p):
        return F.layer_norm(t, (dim,), sd[p + "norm.weight"], sd[p + "norm.bias"])

    for i in range(DEPTH):
        p = f"{pre}layers.{i}."
        x = _vit_attention(sd, p + "0.fn.", ln(x, p + "0."), "time", f, n, frame_rot) + x
        x = _vit_attention(sd, p + "1.fn.", ln(x, p + "1."), "space", f, n, image_rot) + x
        hdn = F.linear(ln(x, p + "2."), sd[p + "2.fn.net.0.weight"], sd[p + "2.fn.net.0.bias"])
        a, gates = hdn.chunk(2, dim=-1)             # GEGLU (vit_modules.py:88-91)
        x = F.linear(a * F.gelu(gates), sd[p + "2.fn.net.3.weight"], sd[p + "2.fn.net.3.bias"]) + x
    return x


def decode_from_sample(sd: Dict[str, torch.Tensor], h: torch.Tensor, res: int = 256, frames: int = 16) -> torch.Tensor:
    """autoencoder_vit.py:257-275.  h [B, 4, r*r + 2*frames*r] (r = res/8) -> frames [B*frames, 3, res, res] in (-1, 1).
    (The reference hard-wires 16 frames at :260-261; `frames` generalises it for small test clips.)"""
    r = res // PATCH
    B, E = h.shape[0], h.shape[1]
    h_xy = h[:, :, 0:r * r].reshape(B, E, r, r)
    h_yt = h[:, :, r * r:r * (r + frames)].reshape(B, E, frames, r)
    h_xt = h[:, :, r * (r + frames):r * (r + 2 * frames)].reshape(B, E, frames, r)
    h_xy = F.conv2d(h_xy, sd["post_xy.weight"], sd["post_xy.bias"])
    h_yt = F.conv2d(h_yt, sd["post_yt.weight"], sd["post_yt.bias"])
    h_xt = F.conv2d(h_xt, sd["post_xt.weight"], sd["post_xt.bias"])
    z = h_xy.unsqueeze(-3) + h_yt.unsqueeze(-2) + h_xt.unsqueeze(-1)            # [B, C, frames(t), r(h), r(w)]
    x = z.permute(0, 2, 3, 4, 1).reshape(B, frames * r * r, -1)                # 'b c f h w -> b (f h w) c'
    dec = timesformer_layers(sd, "decoder.", x, frames, r, r)
    dec = dec.reshape(B * frames, r, r, -1).permute(0, 3, 1, 2)                # 'b (t h w) c -> (b t) c h w'
    pix = F.conv_transpose2d(dec, sd["to_pixel.1.weight"], sd["to_pixel.1.bias"], stride=PATCH)
    return 2 * torch.sigmoid(pix) - 1


def _quant_transformer(sd, pre, x):
    """autoencoder_vit.py:66-84 Transformer(dim, depth 4, heads 4, dim_head dim/8, mlp 512) incl. its Attention (:35-63)."""
    dim = x.shape[-1]
    heads, depth = 4, 4
    dh = dim // 8
    for i in range(depth):
        p = f"{pre}layers.{i}."
        y = F.layer_norm(x, (dim,), sd[p + "0.norm.weight"], sd[p + "0.norm.bias"])
        q, k, v = F.linear(y, sd[p + "0.fn.to_qkv.weight"]).chunk(3, dim=-1)
        b, n = y.shape[0], y.shape[1]
        q, k, v = (t.reshape(b, n, heads, dh).permute(0, 2, 1, 3) for t in (q, k, v))
        dots = torch.matmul(q, k.transpose(-1, -2)) * dh ** -0.5
        out = torch.matmul(dots.softmax(dim=-1), v).permute(0, 2, 1, 3).reshape(b, n, heads * dh)
        x = F.linear(out, sd[p + "0.fn.to_out.0.weight"], sd[p + "0.fn.to_out.0.bias"]) + x
        y = F.layer_norm(x, (dim,), sd[p + "1.norm.weight"], sd[p + "1.norm.bias"])
        y = F.linear(F.gelu(F.linear(y, sd[p + "1.fn.net.0.weight"], sd[p + "1.fn.net.0.bias"])),
                     sd[p + "1.fn.net.3.weight"], sd[p + "1.fn.net.3.bias"])
        x = y + x
    return x


def extract(sd: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    """autoencoder_vit.py:212-255.  x [B, 3, T, res, res] in [-1, 1] -> latents [B, 4, r*r + 2*T*r] (tanh outputs)."""
    B, _, T, H, W = x.shape
    p = PATCH
    hp, wp = H // p, W // p
    v = x.permute(0, 2, 1, 3, 4)                                               # 'b c t h w -> b t c h w'
    v = v.reshape(B, T, 3, hp, p, wp, p).permute(0, 1, 3, 5, 4, 6, 2).reshape(B, T * hp * wp, p * p * 3)   # (p1 p2 c)
    tok = F.linear(v, sd["encoder.to_patch_embedding.weight"], sd["encoder.to_patch_embedding.bias"])
    hfeat = timesformer_layers(sd, "encoder.", tok, T, hp, wp)                 # [B, (t h w), C]
    C = hfeat.shape[-1]
    hf = hfeat.reshape(B, T, hp, wp, C)

    def plane(seq, token, pos, pre):
        n = seq.shape[1]
        seq = torch.cat([seq, token.expand(seq.shape[0], 1, C)], dim=1) + pos[:, :n + 1]
        return _quant_transformer(sd, pre, seq)[:, 0]                           # NB: position 0 (a data token), as the reference does

    h_xy = plane(hf.permute(0, 2, 3, 1, 4).reshape(B * hp * wp, T, C), sd["xy_token"], sd["xy_pos_embedding"], "xy_quant_attn.")
    h_xy = h_xy.reshape(B, hp, wp, C).permute(0, 3, 1, 2)                       # '(b h w) c -> b c h w'
    h_yt = plane(hf.permute(0, 1, 3, 2, 4).reshape(B * T * wp, hp, C), sd["yt_token"], sd["yt_pos_embedding"], "yt_quant_attn.")
    h_yt = h_yt.reshape(B, T, wp, C).permute(0, 3, 1, 2)                        # '(b t w) c -> b c t w'
    h_xt = plane(hf.reshape(B * T * hp, wp, C), sd["xt_token"], sd["xt_pos_embedding"], "xt_quant_attn.")
    h_xt = h_xt.reshape(B, T, hp, C).permute(0, 3, 1, 2)                        # '(b t h) c -> b c t h'
    h_xy = torch.tanh(F.conv2d(h_xy, sd["pre_xy.weight"], sd["pre_xy.bias"]))
    h_yt = torch.tanh(F.conv2d(h_yt, sd["pre_yt.weight"], sd["pre_yt.bias"]))
    h_xt = torch.tanh(F.conv2d(h_xt, sd["pre_xt.weight"], sd["pre_xt.bias"]))
    return torch.cat([h_xy.flatten(2), h_yt.flatten(2), h_xt.flatten(2)], dim=-1)
