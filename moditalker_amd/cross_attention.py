"""Drop-in look-alike of the reference's CrossAttention module (MToV/models/ddpm/unet.py:429-467) on the HIP library.

This is the one usable piece of the reference's dormant cross-attention conditioning path: `SpatialTransformer`
(unet.py:492-528) needs `BasicTransformerBlock`, which is commented out in the reference (unet.py:470-489), so neither the
reference nor this build can construct a UNet with `use_spatial_transformer=True`; `UNetModel` here raises for it.
Same constructor, same state_dict keys (to_q / to_k / to_v / to_out.0), same forward(x, context=None, mask=None);
the arithmetic runs in libmtv_hip.so (`mtv_xattn_forward`): three k_conv GEMMs + k_attention in its cross mode.
No CPU fallback."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.nn as nn

from . import _lib


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner_dim = dim_head * heads
        context_dim = query_dim if context_dim is None else context_dim
        if dropout:
            raise NotImplementedError("dropout must be 0 (inference path)")
        self.query_dim, self.context_dim = query_dim, context_dim
        self.scale = dim_head ** -0.5
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Linear(query_dim, inner_dim, bias=False)
        self.to_k = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_v = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, query_dim), nn.Dropout(dropout))
        self._ctx: Optional[C.c_void_p] = None
        self._key = None
        self._fingerprint = None

    def _release(self):
        if getattr(self, "_ctx", None) is not None:
            _lib.load().mtv_xattn_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_ctx"], st["_key"], st["_fingerprint"] = None, None, None
        return st

    def invalidate_weights(self):
        self._fingerprint = None

    def _context(self, device, B, N, M):
        if device.type != "cuda":
            raise _lib.MtvError("CrossAttention runs only on a HIP device (tensor is on %s); there is no CPU fallback" % device)
        lib = _lib.load()
        key = (device, B, N, M)
        if self._ctx is None or self._key is None or self._key[0] != device or B > self._key[1] or N > self._key[2] or M > self._key[3]:
            self._release()
            cfg = _lib.MtvXattnConfig(self.query_dim, self.context_dim, self.heads, self.dim_head, B, N, M)
            with torch.cuda.device(device):
                ctx = C.c_void_p()
                _lib.check(lib.mtv_xattn_create(C.byref(cfg), C.byref(ctx)), "mtv_xattn_create")
            self._ctx, self._key, self._fingerprint = ctx, key, None
        fp = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if fp != self._fingerprint:
            torch.cuda.synchronize(device)
            with torch.cuda.device(device):
                for k, v in self.state_dict().items():
                    t = v.detach().to(device=device, dtype=torch.float32).contiguous()
                    if t.data_ptr() != v.data_ptr():
                        # a converted / copied temporary is produced on torch's CURRENT stream, while mtv_load_weight copies and
                        # repacks on the NULL stream: finish producing it first (it stays referenced until the call returns,
                        # and the call returns only after its own copy + repack have run)
                        torch.cuda.current_stream(device).synchronize()
                    shp = (C.c_int64 * t.dim())(*t.shape)
                    _lib.check(lib.mtv_load_weight(self._ctx, k.encode(), C.c_void_p(t.data_ptr()), t.dim(), shp), f"mtv_load_weight({k})")
            self._fingerprint = fp
        return self._ctx

    @torch.no_grad()
    def forward(self, x, context=None, mask=None):
        """x [B, N, query_dim], context [B, M, context_dim] (None: self-attention), mask [B, ...] bool over the context
        tokens (True = attend) -> [B, N, query_dim]."""
        if x.dim() != 3 or x.shape[2] != self.query_dim:
            raise ValueError(f"x must be [B,N,{self.query_dim}]; got {tuple(x.shape)}")
        B, N = x.shape[0], x.shape[1]
        if context is not None and (context.dim() != 3 or context.shape[0] != B or context.shape[2] != self.context_dim):
            raise ValueError(f"context must be [{B},M,{self.context_dim}]; got {tuple(context.shape)}")
        if context is None and self.context_dim != self.query_dim:
            raise ValueError("context=None needs context_dim == query_dim")
        M = N if context is None else context.shape[1]
        dev = x.device
        ctx = self._context(dev, B, N, M)
        xf = x.to(torch.float32).contiguous()
        cf = None if context is None else context.to(device=dev, dtype=torch.float32).contiguous()
        mk = None
        if mask is not None:
            mk = mask.reshape(B, -1).to(device=dev, dtype=torch.uint8).contiguous()
            if mk.shape[1] != M:
                raise ValueError(f"mask must cover the {M} context tokens")
            if not bool(mk.any(dim=1).all()):
                raise ValueError("every batch element needs at least one unmasked context token")
        out = torch.empty(B, N, self.query_dim, device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().mtv_xattn_forward(ctx, xf.data_ptr(), None if cf is None else cf.data_ptr(),
                                                     None if mk is None else mk.data_ptr(), out.data_ptr(), B, N, M,
                                                     C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "mtv_xattn_forward")
        return out.type(x.dtype)
