"""CPU oracle for the stand-alone cross-attention operator.  TEST INFRASTRUCTURE ONLY.

Restates MToV/models/ddpm/unet.py:429-467 CrossAttention.forward, op for op, as a function of a state_dict
("to_q.weight", "to_k.weight", "to_v.weight", "to_out.0.weight", "to_out.0.bias").  Pinned: tests/golden/make_golden_xattn.py
imports the reference class, fills it with the recipe of moditalker_amd/filler.py and stores its outputs in
tests/golden/xattn.npz (bit-equal at generation time: PIN_REPORT.txt)."""
import torch
import torch.nn.functional as F


def cross_attention(sd, x, context=None, mask=None, heads=8):
    h = heads
    q = F.linear(x, sd["to_q.weight"])
    context = x if context is None else context                    # default(context, x), unet.py:445
    k = F.linear(context, sd["to_k.weight"])
    v = F.linear(context, sd["to_v.weight"])
    b, n, inner = q.shape
    d = inner // h

    def split(t):                                                   # 'b n (h d) -> (b h) n d'
        return t.reshape(b, t.shape[1], h, d).permute(0, 2, 1, 3).reshape(b * h, t.shape[1], d)

    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum("b i d, b j d -> b i j", q, k) * d ** -0.5
    if mask is not None:                                            # unet.py:451-456
        m = mask.reshape(b, -1)
        m = m[:, None, None, :].expand(b, h, 1, m.shape[1]).reshape(b * h, 1, m.shape[1])
        sim = sim.masked_fill(~m, -torch.finfo(sim.dtype).max)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("b i j, b j d -> b i d", attn, v)
    out = out.reshape(b, h, n, d).permute(0, 2, 1, 3).reshape(b, n, h * d)
    return F.linear(out, sd["to_out.0.weight"], sd["to_out.0.bias"])
