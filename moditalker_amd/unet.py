"""Drop-in look-alikes of the reference's denoiser classes, backed by the HIP library.

Mirrors the *interface* of
  MToV/models/ddpm/unet.py:601-1117  UNetModel   (ctor kwargs, .forward signature, attributes read by
                                                  callers: .in_channels, .image_size, .cond_model)
  MToV/models/ddpm/unet.py:34-61     DiffusionWrapper
and the 804-key `state_dict()` layout (incl. the 246 dead `output_bg_*` keys a strict
`load_state_dict` needs).  The modules below only HOLD parameters under the
reference's names; `forward` hands raw device pointers to libmtv_hip.so (`mtv_forward`).  There is
no PyTorch/CPU fallback: calling forward off-GPU, or without the built library, raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib


def _zero(m: nn.Module) -> nn.Module:
    for p in m.parameters():
        p.detach().zero_()
    return m


class _Res(nn.Module):
    """Parameter holder with the key layout of ResBlock (unet.py:109-167)."""

    def __init__(self, cin: int, emb: int, cout: int, scale_shift: bool, updown: Optional[str]):
        super().__init__()
        self.channels, self.out_channels, self.updown = cin, cout, updown
        self.in_layers = nn.Sequential(nn.GroupNorm(32, cin), nn.SiLU(), nn.Conv2d(cin, cout, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb, 2 * cout if scale_shift else cout))
        self.out_layers = nn.Sequential(nn.GroupNorm(32, cout), nn.SiLU(), nn.Dropout(p=0.0),
                                        _zero(nn.Conv2d(cout, cout, 3, padding=1)))
        self.skip_connection = nn.Identity() if cin == cout else nn.Conv2d(cin, cout, 1)


class _Attn(nn.Module):
    """Parameter holder with the key layout of AttentionBlock / AttentionBlock1D (unet.py:217-242)."""

    def __init__(self, ch: int):
        super().__init__()
        self.channels = ch
        self.norm = nn.GroupNorm(32, ch)
        self.qkv = nn.Conv1d(ch, ch * 3, 1)
        self.proj_out = _zero(nn.Conv1d(ch, ch, 1))


class UNetModel(nn.Module):
    """Tri-plane UNet denoiser; constructor kwargs exactly as the reference YAML passes them
    (configs/latent-diffusion/base.yaml:27-38 -> unet.py:631-659), plus the geometry the reference
    hard-wires: `frames` (T, default 16; R is `image_size`)."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks,
                 attention_resolutions, dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2,
                 num_classes=None, use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1,
                 num_heads_upsample=-1, use_scale_shift_norm=False, resblock_updown=False,
                 use_new_attention_order=False, use_spatial_transformer=False, transformer_depth=1,
                 context_dim=None, n_embed=None, legacy=True, cond_model=False, frames=16, max_batch=1):
        super().__init__()
        # options the hot path never exercises in the reference
        if use_spatial_transformer or context_dim is not None:
            raise NotImplementedError("SpatialTransformer cannot be constructed in the reference either (its BasicTransformerBlock is "
                                      "commented out, unet.py:470-489,513); the usable piece, CrossAttention, is moditalker_amd.CrossAttention")
        if dims != 2 or num_classes is not None or use_fp16 or use_new_attention_order or n_embed is not None:
            raise NotImplementedError("only dims=2, fp32, legacy attention order, no class/codebook heads")
        if not resblock_updown:
            raise NotImplementedError("only resblock_updown=True (the shipped configuration) is built")
        if dropout:
            raise NotImplementedError("dropout must be 0 (inference path)")
        if num_heads == -1 or num_head_channels != -1 or num_heads_upsample not in (-1, num_heads):
            raise NotImplementedError("set num_heads (num_head_channels / num_heads_upsample are unused by the reference config)")
        if len(channel_mult) > _lib.MTV_MAX_LEVELS:
            raise ValueError("too many levels")
        self.image_size = image_size
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.num_res_blocks = num_res_blocks
        self.attention_resolutions = list(attention_resolutions)
        self.dropout = dropout
        self.channel_mult = list(channel_mult)
        self.conv_resample = conv_resample
        self.num_classes = num_classes
        self.use_checkpoint = use_checkpoint
        self.dtype = torch.float32
        self.num_heads = num_heads
        self.num_head_channels = num_head_channels
        self.num_heads_upsample = num_heads
        self.use_scale_shift_norm = bool(use_scale_shift_norm)
        self.predict_codebook_ids = False
        self.cond_model = cond_model
        self.frames = frames
        self.max_batch = max_batch
        if cond_model:
            self.register_buffer("zeros", torch.zeros(1, self.in_channels, 2048))   # unet.py:697-698

        mc, ss = model_channels, self.use_scale_shift_norm
        emb = mc * 4
        self.time_embed = nn.Sequential(nn.Linear(mc, emb), nn.SiLU(), nn.Linear(emb, emb))
        Seq = nn.Sequential
        self.input_blocks = nn.ModuleList([Seq(nn.Conv2d(16, mc, 3, padding=1))])      # unet.py:714
        self.input_attns = nn.ModuleList([nn.Identity()])
        chans = [mc]
        ch, ds = mc, 1
        att = set(self.attention_resolutions)
        for level, mult in enumerate(self.channel_mult):
            for _ in range(num_res_blocks):
                layers: List[nn.Module] = [_Res(ch, emb, mult * mc, ss, None)]
                ch = mult * mc
                if ds in att:
                    layers.append(_Attn(ch))
                self.input_blocks.append(Seq(*layers))
                chans.append(ch)
                self.input_attns.append(_Attn(ch))
            if level != len(self.channel_mult) - 1:
                self.input_blocks.append(Seq(_Res(ch, emb, ch, ss, "down")))
                chans.append(ch)
                ds *= 2
                self.input_attns.append(_Attn(ch))
        self.middle_block = Seq(_Res(ch, emb, ch, ss, None), _Attn(ch), _Res(ch, emb, ch, ss, None))
        self.mid_attn = _Attn(ch)
        self.output_blocks = nn.ModuleList([])
        self.output_bg_blocks = nn.ModuleList([])    # constructed, never called (unet.py:859-968)
        self.output_attns = nn.ModuleList([])
        self.output_bg_attns = nn.ModuleList([])
        for level, mult in list(enumerate(self.channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [_Res(ch + ich, emb, mc * mult, ss, None)]
                layers_bg: List[nn.Module] = [_Res(ch + ich, emb, mc * mult, ss, None)]
                ch = mc * mult
                if ds in att:
                    layers.append(_Attn(ch))
                if level and i == num_res_blocks:
                    layers.append(_Res(ch, emb, ch, ss, "up"))
                    layers_bg.append(_Res(ch, emb, ch, ss, "up"))
                    ds //= 2
                self.output_blocks.append(Seq(*layers))
                self.output_bg_blocks.append(Seq(*layers_bg))
                self.output_attns.append(_Attn(ch))
                self.output_bg_attns.append(_Attn(ch))
        self.out = Seq(nn.GroupNorm(32, ch), nn.SiLU(), _zero(nn.Conv2d(mc, out_channels, 3, padding=1)))

        self._ctx: Optional[C.c_void_p] = None
        self._ctx_device: Optional[torch.device] = None
        self._ctx_batch = 0
        self._fingerprint = None
        self._eager = False

    # ------------------------------------------------------------------ HIP context management
    def _config(self, max_batch: int) -> _lib.MtvConfig:
        cfg = _lib.MtvConfig()
        cfg.model_channels = self.model_channels
        cfg.num_res_blocks = self.num_res_blocks
        cfg.num_heads = self.num_heads
        cfg.n_levels = len(self.channel_mult)
        for i, m in enumerate(self.channel_mult):
            cfg.channel_mult[i] = int(m)
        cfg.n_attention_resolutions = len(self.attention_resolutions)
        for i, a in enumerate(self.attention_resolutions):
            cfg.attention_resolutions[i] = int(a)
        cfg.use_scale_shift_norm = int(self.use_scale_shift_norm)
        cfg.out_channels = self.out_channels
        cfg.res = self.image_size
        cfg.frames = self.frames
        cfg.max_batch = max_batch
        return cfg

    def _release(self):
        if getattr(self, "_ctx", None) is not None:
            _lib.load().mtv_destroy(self._ctx)       # (binds the context's own device before freeing)
            self._ctx = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _weights_fingerprint(self):
        """(storage, version) of every parameter.  In-place writes through `.data` (p.data.copy_/lerp_: the usual
        EMA / weight-surgery idiom) do NOT bump `_version`: call `invalidate_weights()` after such writes.
        `load_state_dict` and `.to()/_apply` invalidate by themselves."""
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def invalidate_weights(self):
        """Force a re-upload of all weights to the HIP context before the next forward / sample."""
        self._fingerprint = None

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate_weights()
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._fingerprint = None
        return r

    # the library handle is process- and device-local: copies / pickles start without one and create their own
    # lazily (the reference workflow deep-copies the model for its EMA twin, sample.py:226)
    def __getstate__(self):
        st = dict(self.__dict__)
        st["_ctx"], st["_ctx_device"], st["_ctx_batch"], st["_fingerprint"] = None, None, 0, None
        return st

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__getstate__().items():
            new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def hip_context(self, device: torch.device, batch: int) -> C.c_void_p:
        """Create (or reuse) the library context for `device`, sized for `batch`, with current weights."""
        if device.type != "cuda":
            raise _lib.MtvError("UNetModel.forward runs only on a HIP device (tensor is on %s); there is no CPU fallback" % device)
        lib = _lib.load()
        if self._ctx is None or self._ctx_device != device or batch > self._ctx_batch:
            self._release()
            with torch.cuda.device(device):
                ctx = C.c_void_p()
                cfg = self._config(max(batch, self.max_batch))
                _lib.check(lib.mtv_create(C.byref(cfg), C.byref(ctx)), "mtv_create")
            self._ctx, self._ctx_device, self._ctx_batch = ctx, device, max(batch, self.max_batch)
            self._fingerprint = None
            lib.mtv_set_eager(self._ctx, int(self._eager))
        fp = self._weights_fingerprint()
        if fp != self._fingerprint:
            self._upload_weights(device)
            self._fingerprint = fp
        return self._ctx

    def _upload_weights(self, device: torch.device):
        lib = _lib.load()
        sd = self.state_dict()
        n = lib.mtv_num_weights(self._ctx)
        key = C.create_string_buffer(256)
        ndim = C.c_int()
        shape = (C.c_int64 * 4)()
        torch.cuda.synchronize(device)
        for i in range(n):
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
        missing = lib.mtv_weights_missing(self._ctx)
        if missing:
            raise _lib.MtvError(f"{missing} weights missing after upload")

    def check_fault(self):
        """Raise MtvError if an in-launch hand-off of an earlier call on this module's context timed out (include/mtv_hip.h
        mtv_check_fault).  Meaningful once the stream those calls ran on has drained: call it after your own synchronisation
        point (`.cpu()`, `torch.cuda.synchronize()`); it does not synchronise.  No context yet: nothing to check."""
        if self._ctx is not None:
            _lib.check(_lib.load().mtv_check_fault(self._ctx), "mtv_check_fault")

    def after_call(self, dev, strict=None):
        """strict (None: MTV_STRICT=1 in the environment): drain the current stream and check the fault word, so that a call whose
        result is invalid raises ITSELF, as the reference's synchronous torch ops would."""
        if strict is None:
            strict = os.environ.get("MTV_STRICT") == "1"
        if strict:
            torch.cuda.current_stream(dev).synchronize()
            self.check_fault()

    @property
    def resident_cus(self) -> int:
        """CUs the context plans its in-launch hand-offs for (0: no context yet)."""
        return int(_lib.load().mtv_resident_cus(self._ctx)) if self._ctx is not None else 0

    def set_eager(self, eager: bool):
        """True: plain kernel launches instead of hipGraph replay (profiling / debugging)."""
        self._eager = bool(eager)
        if self._ctx is not None:
            _lib.load().mtv_set_eager(self._ctx, int(self._eager))

    def work(self, device=None) -> Dict[str, float]:
        """Algorithmic FLOPs / bytes of one batch-1 forward (for bench.py's roofline line)."""
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        ctx = self.hip_context(dev, 1)
        w = _lib.MtvWork()
        _lib.check(_lib.load().mtv_get_work(ctx, C.byref(w)), "mtv_get_work")
        return {f: getattr(w, f) for f, _ in w._fields_}

    def profile_forward(self, batch: int = 1, iters: int = 5, device=None, step: bool = False):
        """Per-launch hipEvent timings (plain launches) of one forward -- or, with step=True, of one SAMPLER STEP
        (the launch sequence DDPM.sample replays: UNet launches only, DDIM update inside the head conv): list of
        dicts {name, ms, flops, bytes}.  Inputs are whatever the staging buffers currently hold."""
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        ctx = self.hip_context(dev, batch)
        lib = _lib.load()
        fn, what = (lib.mtv_profile_step, "mtv_profile_step") if step else (lib.mtv_profile_forward, "mtv_profile_forward")
        n = C.c_int()
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            _lib.check(fn(ctx, batch, iters, None, 0, C.byref(n), stream), what)
            table = (_lib.MtvOpTime * n.value)()
            _lib.check(fn(ctx, batch, iters, table, n.value, C.byref(n), stream), what)
        return [dict(name=t.name.decode(), ms=float(t.ms), flops=float(t.flops), bytes=float(t.bytes)) for t in table]

    # ------------------------------------------------------------------ reference-shaped API
    def check_inputs(self, x, cond, image_cond) -> int:
        """Shape contract of the hot path (what `torch.cat` at unet.py:1022-1025 enforces in the reference): the C
        side only sees raw pointers, so everything that reaches it is validated here.  Returns B."""
        R, T = self.image_size, self.frames
        L = R * R + 2 * T * R
        if x is None or x.dim() != 3 or x.shape[1] != 4 or x.shape[2] != L:
            raise ValueError(f"x must be [B,4,{L}] for (R,T)=({R},{T}); got {None if x is None else tuple(x.shape)}")
        B = x.shape[0]
        if cond is None or tuple(cond.shape) != (B, 8, L):
            raise ValueError(f"cond must be [{B},8,{L}]; got {None if cond is None else tuple(cond.shape)}")
        if image_cond is None or image_cond.dim() != 3 or tuple(image_cond.shape[:2]) != (B, 4) or image_cond.shape[2] < R * R:
            raise ValueError(f"image_cond must be [{B},4,>={R * R}]; got {None if image_cond is None else tuple(image_cond.shape)}")
        return B

    @torch.no_grad()
    def forward(self, x, cond=None, image_cond=None, timesteps=None, context=None, y=None, **kwargs):
        """x [B,4,L], cond [B,8,L], image_cond [B,4,>=R*R], timesteps [B] -> eps [B,out_channels,L]
        (unet.py:995).  `context` is ignored exactly as the reference ignores it (always None there)."""
        assert (y is not None) == (self.num_classes is not None), "must specify y if and only if the model is class-conditional"
        strict = kwargs.pop("strict", None)      # (appended option, see after_call; every other kwarg is ignored as in the reference)
        R, T = self.image_size, self.frames
        L = R * R + 2 * T * R
        B = self.check_inputs(x, cond, image_cond)
        dev = x.device
        ctx = self.hip_context(dev, B)
        xf = x.to(torch.float32).contiguous()
        cf = cond.to(device=dev, dtype=torch.float32).contiguous()
        icf = image_cond.to(device=dev, dtype=torch.float32).contiguous()
        tt = timesteps.to(device=dev, dtype=torch.int64).contiguous()
        if tt.numel() != B:
            raise ValueError("timesteps must have one entry per batch element")
        out = torch.empty(B, self.out_channels, L, device=dev, dtype=torch.float32)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(_lib.load().mtv_forward(ctx, xf.data_ptr(), cf.data_ptr(), icf.data_ptr(), icf.shape[2],
                                               tt.data_ptr(), out.data_ptr(), B, C.c_void_p(stream)), "mtv_forward")
            self.after_call(dev, strict)
        return out.type(x.dtype)

    def debug_tap(self, name: str, batch: int) -> torch.Tensor:
        """Intermediate activation after the cross-plane attention of stage `name` as [B, C, L']
        (the reference's layout), for parity bisecting."""
        lib = _lib.load()
        tok, ch = C.c_int(), C.c_int()
        _lib.check(lib.mtv_debug_tap(self._ctx, name.encode(), None, 0, C.byref(tok), C.byref(ch)), "mtv_debug_tap")
        buf = torch.empty(batch, tok.value, ch.value, device=self._ctx_device, dtype=torch.float32)
        _lib.check(lib.mtv_debug_tap(self._ctx, name.encode(), C.c_void_p(buf.data_ptr()), buf.numel(), None, None), "mtv_debug_tap")
        return buf.permute(0, 2, 1).contiguous()


class DiffusionWrapper(nn.Module):
    """unet.py:34-61.  Only conditioning_key=None is live in the reference."""

    def __init__(self, model, conditioning_key=None):
        super().__init__()
        self.diffusion_model = model
        self.conditioning_key = conditioning_key
        assert self.conditioning_key in [None, "concat", "crossattn", "hybrid", "adm"]

    def forward(self, x, cond, image_cond, t, kpt_coord=None, c_concat: list = None, c_crossattn: list = None):
        if self.conditioning_key is None:
            return self.diffusion_model(x, cond, image_cond, t, context=c_crossattn)
        raise NotImplementedError("only conditioning_key=None is reachable in the reference sampling path")
