"""TEST INFRASTRUCTURE ONLY: CPU emulation of the arithmetic of k_attention_b3 (moditalker_amd/csrc/attn_b3.hip) -- the
QKVAttentionLegacy core (MToV/models/ddpm/unet.py:312-326) with q, k, v split into three bf16 terms (round to nearest
even), the softmax probabilities into two, and every product taken as the partial products the kernel issues.  It is not
the reference's arithmetic (that is oracle/ref_unet.py's fp32 attention, pinned to the reference); it exists to show on
the CPU that the split keeps fp32-class accuracy (tests/test_oracle_golden.py::test_split_bf16_attention_emulation)."""
import math

import torch


def split_terms(x: torch.Tensor, n: int):
    out, r = [], x.clone()
    for _ in range(n):
        h = r.to(torch.bfloat16).to(torch.float32)
        out.append(h)
        r = r - h
    return out


QK_PAIRS = [(2, 0), (0, 2), (1, 1), (0, 1), (1, 0), (0, 0)]          # (k term, q term): six products, small first
PV_PAIRS = [(0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]                  # (p term, v term): p has two terms


def attention_split_bf16(q, k, v):
    """q, k, v [H, L, d] fp32 -> [H, L, d]; scale d^-1/4 on q and k, softmax over keys in the log2 domain."""
    d = q.shape[-1]
    sc = d ** -0.25
    qt = split_terms(q * (sc * math.log2(math.e)), 3)
    kt = split_terms(k * sc, 3)
    s = None
    for ik, iq in QK_PAIRS:
        p = qt[iq] @ kt[ik].transpose(-1, -2)
        s = p if s is None else s + p
    m = s.max(-1, keepdim=True).values
    p = torch.exp2(s - m)
    l = p.sum(-1, keepdim=True)
    pt, vt = split_terms(p, 2), split_terms(v, 3)
    o = None
    for ip, iv in PV_PAIRS:
        t = pt[ip] @ vt[iv]
        o = t if o is None else o + t
    return o / l


def attention_qk_split_bf16(q, k, v):
    """The shipped hybrid (k_attention<..., QB = 1>, csrc/kernels.hip): QK^T from three bf16 terms of q and k (six products, small
    first), softmax and PV in fp32."""
    d = q.shape[-1]
    sc = d ** -0.25
    qt = split_terms(q * (sc * math.log2(math.e)), 3)
    kt = split_terms(k * sc, 3)
    s = None
    for ik, iq in QK_PAIRS:
        p = qt[iq] @ kt[ik].transpose(-1, -2)
        s = p if s is None else s + p
    m = s.max(-1, keepdim=True).values
    p = torch.exp2(s - m)
    return (p @ v) / p.sum(-1, keepdim=True)


CONV_PAIRS = [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)]          # (activation term, weight term): k_conv_x3's order


def gemm_split_bf16(a, w):
    """k_conv_x3's arithmetic (csrc/conv_x3.hip): a [M, K], w [K, N] fp32, both as three bf16 terms, six partial products with fp32
    accumulation (the MFMA's own summation order inside a product is not modelled: torch.matmul's stands in)."""
    at, wt = split_terms(a, 3), split_terms(w, 3)
    o = None
    for ia, iw in CONV_PAIRS:
        t = at[ia] @ wt[iw]
        o = t if o is None else o + t
    return o
