"""Drop-in look-alike of the reference's first-stage autoencoder for the two calls the sampling pipeline makes:

  MToV/models/autoencoder/autoencoder_vit.py:89-155   ViTAutoencoder.__init__  (same ctor arguments; sample.py:206-218 builds
                                                       it as ViTAutoencoder(embed_dim, ddconfig) and load_state_dict()s it)
  :257-275  decode_from_sample(h)   latents [B,4,L] -> frames [B*16,3,res,res] in (-1,1)   (sample.py:369,386)
  :212-255  extract(x)              video [B,3,T,res,res] in [-1,1] -> latents [B,4,L]     (sample.py:328-331)

The module only HOLDS parameters under the reference's 415 state_dict keys (TimeSformer encoder/decoder stacks,
quant transformers, tokens, position embeddings, pre/post 1x1 convs, to_pixel, rotary buffers); the two methods hand
raw device pointers to libmtv_hip.so (`mtv_ae_decode` / `mtv_ae_extract`).  There is no PyTorch/CPU fallback.
Training-side entry points (`encode`, `decode`, `forward`) are outside the MI355X hot path and raise.
"""
from __future__ import annotations

import ctypes as C
from math import log, pi
from typing import Optional

import torch
import torch.nn as nn

from . import _lib

HEADS, DIM_HEAD, DEPTH = 8, 64, 8        # vit_modules.py:162-163,243-244; autoencoder_vit.py:110-116


class _Fn(nn.Module):
    pass


class _PreNorm(nn.Module):
    """Key layout of PreNorm(dim, fn): `norm.*`, `fn.*` (vit_modules.py:70-79, autoencoder_vit.py:15-23)."""

    def __init__(self, dim: int, fn: nn.Module, norm_first: bool = False):
        super().__init__()
        if norm_first:                      # autoencoder_vit.py's PreNorm registers norm before fn (state_dict order)
            self.norm = nn.LayerNorm(dim)
            self.fn = fn
        else:
            self.fn = fn
            self.norm = nn.LayerNorm(dim)


def _attention(dim: int, heads: int, dim_head: int) -> nn.Module:
    m = _Fn()
    m.to_qkv = nn.Linear(dim, heads * dim_head * 3, bias=False)
    m.to_out = nn.Sequential(nn.Linear(heads * dim_head, dim), nn.Dropout(0.0))
    return m


def _feedforward(dim: int, hidden: int, geglu: bool) -> nn.Module:
    m = _Fn()
    # vit_modules.py:93-104: [Linear(dim, 8 dim), GEGLU, Dropout, Linear(4 dim, dim)];
    # autoencoder_vit.py:26-32: [Linear, GELU, Dropout, Linear, Dropout] -- parameters sit at indices 0 and 3 in both
    m.net = nn.Sequential(nn.Linear(dim, hidden * 2 if geglu else hidden), nn.Identity(), nn.Dropout(0.0), nn.Linear(hidden, dim))
    return m


class _RotBuf(nn.Module):
    def __init__(self, name: str, value: torch.Tensor):
        super().__init__()
        self.register_buffer(name, value)


class _TimeSformer(nn.Module):
    """Parameter holder with the key layout of TimeSformerEncoder / TimeSformerDecoder (vit_modules.py:150-303)."""

    def __init__(self, dim: int, patch_dim: Optional[int]):
        super().__init__()
        if patch_dim is not None:
            self.to_patch_embedding = nn.Linear(patch_dim, dim)
        self.frame_rot_emb = _RotBuf("inv_freqs", 1.0 / (10000 ** (torch.arange(0, DIM_HEAD, 2).float() / DIM_HEAD)))
        self.image_rot_emb = _RotBuf("scales", torch.logspace(0.0, log(10 / 2) / log(2), DIM_HEAD // 4, base=2))
        self.layers = nn.ModuleList([
            nn.ModuleList([_PreNorm(dim, _attention(dim, HEADS, DIM_HEAD)), _PreNorm(dim, _attention(dim, HEADS, DIM_HEAD)),
                           _PreNorm(dim, _feedforward(dim, dim * 4, True))]) for _ in range(DEPTH)])


class _QuantTransformer(nn.Module):
    """autoencoder_vit.py:66-84 Transformer(dim, depth, heads, dim_head, mlp_dim)."""

    def __init__(self, dim: int, depth: int, heads: int, dim_head: int, mlp_dim: int):
        super().__init__()
        self.layers = nn.ModuleList([
            nn.ModuleList([_PreNorm(dim, _attention(dim, heads, dim_head), True), _PreNorm(dim, _feedforward(dim, mlp_dim, False), True)])
            for _ in range(depth)])


class ViTAutoencoder(nn.Module):
    def __init__(self, embed_dim, ddconfig, ckpt_path=None, ignore_keys=[], image_key="image", colorize_nlabels=None,
                 monitor=None, max_batch: int = 1):
        super().__init__()
        self.splits = ddconfig["splits"]
        self.s = ddconfig["timesteps"] // self.splits
        self.res = ddconfig["resolution"]
        self.embed_dim = embed_dim
        self.image_key = image_key
        self.channels = ddconfig["channels"]
        self.patch_size = 4 if self.res == 128 else 8          # autoencoder_vit.py:105-107
        self.down = 3
        self.max_batch = max_batch
        if self.splits != 1:
            raise NotImplementedError("splits != 1 only matters for ViTAutoencoder.forward (training); sampling uses splits = 1")
        ch, p = self.channels, self.patch_size
        self.encoder = _TimeSformer(ch, 3 * p * p)
        self.decoder = _TimeSformer(ch, None)
        self.to_pixel = nn.Sequential(nn.Identity(), nn.ConvTranspose2d(ch, 3, kernel_size=(p, p), stride=p))
        self.register_buffer("coords", torch.linspace(-1, 1, steps=self.s).unsqueeze(-1))
        self.xy_token = nn.Parameter(torch.randn(1, 1, ch))
        self.xt_token = nn.Parameter(torch.randn(1, 1, ch))
        self.yt_token = nn.Parameter(torch.randn(1, 1, ch))
        lat = self.res // (2 ** self.down)
        self.xy_pos_embedding = nn.Parameter(torch.randn(1, self.s + 1, ch))
        self.xt_pos_embedding = nn.Parameter(torch.randn(1, lat + 1, ch))
        self.yt_pos_embedding = nn.Parameter(torch.randn(1, lat + 1, ch))
        self.xy_quant_attn = _QuantTransformer(ch, 4, 4, ch // 8, 512)
        self.yt_quant_attn = _QuantTransformer(ch, 4, 4, ch // 8, 512)
        self.xt_quant_attn = _QuantTransformer(ch, 4, 4, ch // 8, 512)
        self.pre_xy = nn.Conv2d(ch, embed_dim, 1)
        self.pre_xt = nn.Conv2d(ch, embed_dim, 1)
        self.pre_yt = nn.Conv2d(ch, embed_dim, 1)
        self.post_xy = nn.Conv2d(embed_dim, ch, 1)
        self.post_xt = nn.Conv2d(embed_dim, ch, 1)
        self.post_yt = nn.Conv2d(embed_dim, ch, 1)
        if self.res // p != lat:
            raise NotImplementedError("only patch 8 / down 3 geometries (latent side = res / 8)")
        self._ctx: Optional[C.c_void_p] = None
        self._ctx_device: Optional[torch.device] = None
        self._ctx_batch = 0
        self._fingerprint = None

    # ------------------------------------------------------------------ host-side tables (exactly the reference's arithmetic)
    def _rotary_tables(self):
        """[frames][2][64] and [r*r][2][64]: (sin, cos) of RotaryEmbedding / AxialRotaryEmbedding.forward
        (vit_modules.py:29-49,57-62), evaluated with torch on the host like the reference does."""
        for a, b in ((self.encoder.frame_rot_emb.inv_freqs, self.decoder.frame_rot_emb.inv_freqs),
                     (self.encoder.image_rot_emb.scales, self.decoder.image_rot_emb.scales)):
            if not torch.equal(a.detach().cpu(), b.detach().cpu()):
                raise NotImplementedError("encoder and decoder rotary buffers differ (the reference derives both from the same constants)")
        inv = self.decoder.frame_rot_emb.inv_freqs.detach().float().cpu()
        seq = torch.arange(self.s)
        fr = torch.einsum("i, j -> i j", seq, inv)
        fr = torch.cat((fr, fr), dim=-1)
        time_tab = torch.stack((fr.sin(), fr.cos()), dim=1).contiguous()
        scales = self.decoder.image_rot_emb.scales.detach().float().cpu()[None]
        r = self.res // self.patch_size
        h_seq = torch.linspace(-1.0, 1.0, steps=r).unsqueeze(-1) * scales * pi
        w_seq = torch.linspace(-1.0, 1.0, steps=r).unsqueeze(-1) * scales * pi
        x_sinu = h_seq[:, None, :].expand(r, r, -1)
        y_sinu = w_seq[None, :, :].expand(r, r, -1)
        sin = torch.cat((x_sinu.sin(), y_sinu.sin()), dim=-1).reshape(r * r, -1).repeat_interleave(2, dim=-1)
        cos = torch.cat((x_sinu.cos(), y_sinu.cos()), dim=-1).reshape(r * r, -1).repeat_interleave(2, dim=-1)
        space_tab = torch.stack((sin, cos), dim=1).contiguous()
        return time_tab, space_tab

    # ------------------------------------------------------------------ HIP context management
    def _release(self):
        if getattr(self, "_ctx", None) is not None:
            _lib.load().mtv_ae_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_ctx"], st["_ctx_device"], st["_ctx_batch"], st["_fingerprint"] = None, None, 0, None
        return st

    def invalidate_weights(self):
        self._fingerprint = None

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._fingerprint = None
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._fingerprint = None
        return r

    def hip_context(self, device: torch.device, batch: int) -> C.c_void_p:
        if device.type != "cuda":
            raise _lib.MtvError("ViTAutoencoder runs only on a HIP device (tensor is on %s); there is no CPU fallback" % device)
        lib = _lib.load()
        if self._ctx is None or self._ctx_device != device or batch > self._ctx_batch:
            self._release()
            cfg = _lib.MtvAeConfig(self.channels, self.res, self.s, self.patch_size, self.embed_dim, DEPTH, HEADS, DIM_HEAD,
                                   max(batch, self.max_batch))
            with torch.cuda.device(device):
                ctx = C.c_void_p()
                _lib.check(lib.mtv_ae_create(C.byref(cfg), C.byref(ctx)), "mtv_ae_create")
            self._ctx, self._ctx_device, self._ctx_batch = ctx, device, max(batch, self.max_batch)
            self._fingerprint = None
        fp = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if fp != self._fingerprint:
            sd = self.state_dict()
            key = C.create_string_buffer(256)
            ndim = C.c_int()
            shape = (C.c_int64 * 4)()
            torch.cuda.synchronize(device)
            with torch.cuda.device(device):
                for i in range(lib.mtv_num_weights(self._ctx)):
                    _lib.check(lib.mtv_weight_info(self._ctx, i, key, 256, C.byref(ndim), shape), "mtv_weight_info")
                    k = key.value.decode()
                    if k not in sd:
                        raise _lib.MtvError(f"library expects weight '{k}' which this module does not hold")
                    t = sd[k].detach().to(device=device, dtype=torch.float32).contiguous()
                    if t.data_ptr() != sd[k].data_ptr():
                        # a converted / copied temporary is produced on torch's CURRENT stream, while mtv_load_weight copies and
                        # repacks on the NULL stream: finish producing it first (it stays referenced until the call returns,
                        # and the call returns only after its own copy + repack have run)
                        torch.cuda.current_stream(device).synchronize()
                    shp = (C.c_int64 * t.dim())(*t.shape)
                    _lib.check(lib.mtv_load_weight(self._ctx, k.encode(), C.c_void_p(t.data_ptr()), t.dim(), shp), f"mtv_load_weight({k})")
                tt, st = self._rotary_tables()
                _lib.check(lib.mtv_ae_set_rotary(self._ctx, C.c_void_p(tt.data_ptr()), C.c_void_p(st.data_ptr())), "mtv_ae_set_rotary")
            if lib.mtv_weights_missing(self._ctx):
                raise _lib.MtvError(f"{lib.mtv_weights_missing(self._ctx)} weights missing after upload")
            self._fingerprint = fp
        return self._ctx

    # ------------------------------------------------------------------ reference-shaped API
    @torch.no_grad()
    def decode_from_sample(self, h):
        """autoencoder_vit.py:257-275: h [B, embed_dim, r*r + 2*16*r] -> [B*16, 3, res, res] in (-1, 1)."""
        r = self.res // self.patch_size
        L = r * r + 2 * self.s * r
        if h.dim() != 3 or h.shape[1] != self.embed_dim or h.shape[2] != L:
            raise ValueError(f"h must be [B,{self.embed_dim},{L}]; got {tuple(h.shape)}")
        dev, B = h.device, h.shape[0]
        ctx = self.hip_context(dev, B)
        hf = h.to(torch.float32).contiguous()
        out = torch.empty(B * self.s, 3, self.res, self.res, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().mtv_ae_decode(ctx, hf.data_ptr(), out.data_ptr(), B,
                                                 C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "mtv_ae_decode")
        return out.type(h.dtype)

    @torch.no_grad()
    def extract(self, x):
        """autoencoder_vit.py:212-255: x [B, 3, T, res, res] in [-1, 1] -> latents [B, embed_dim, r*r + 2*T*r]."""
        if x.dim() != 5 or tuple(x.shape[1:]) != (3, self.s, self.res, self.res):
            raise ValueError(f"x must be [B,3,{self.s},{self.res},{self.res}]; got {tuple(x.shape)}")
        dev, B = x.device, x.shape[0]
        ctx = self.hip_context(dev, B)
        r = self.res // self.patch_size
        xf = x.to(torch.float32).contiguous()
        out = torch.empty(B, self.embed_dim, r * r + 2 * self.s * r, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().mtv_ae_extract(ctx, xf.data_ptr(), out.data_ptr(), B,
                                                  C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "mtv_ae_extract")
        return out.type(x.dtype)

    def profile(self, batch: int = 1, extract: bool = False, iters: int = 3, device=None):
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        ctx = self.hip_context(dev, batch)
        lib = _lib.load()
        n = C.c_int()
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            _lib.check(lib.mtv_ae_profile(ctx, batch, int(extract), iters, None, 0, C.byref(n), stream), "mtv_ae_profile")
            table = (_lib.MtvOpTime * n.value)()
            _lib.check(lib.mtv_ae_profile(ctx, batch, int(extract), iters, table, n.value, C.byref(n), stream), "mtv_ae_profile")
        return [dict(name=t.name.decode(), ms=float(t.ms), flops=float(t.flops), bytes=float(t.bytes)) for t in table]

    def encode(self, x):
        raise NotImplementedError("ViTAutoencoder.encode is the training-time path (autoencoder_vit.py:157-205); sampling uses extract()")

    def decode(self, z):
        raise NotImplementedError("ViTAutoencoder.decode takes the expanded training-time tensor; sampling uses decode_from_sample()")

    def forward(self, input):
        raise NotImplementedError("autoencoder training forward is outside the MI355X hot path (SURVEY.md section 8)")
