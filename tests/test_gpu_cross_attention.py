"""GPU parity of the cross-attention operator: moditalker_amd.CrossAttention (C ABI
mtv_xattn_forward) vs the golden vectors of the reference's CrossAttention class (tests/golden/xattn.npz)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from moditalker_amd import CrossAttention, filler

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,qd,cd,H,d,B,N,M", [("cross", 256, 128, 8, 32, 2, 96, 77), ("self", 128, None, 4, 64, 1, 160, None)])
def test_cross_attention_vs_reference_golden(tag, qd, cd, H, d, B, N, M):
    g = np.load(os.path.join(GOLDEN, "xattn.npz"))
    dev = torch.device("cuda:0")
    m = CrossAttention(qd, cd, heads=H, dim_head=d).eval()
    filler.fill_module_(m, seed=51)
    m = m.to(dev)
    x = filler.uniform_pm1(f"xattn.{tag}.x", (B, N, qd), 51).to(dev)
    ctx = filler.uniform_pm1(f"xattn.{tag}.ctx", (B, M, cd), 51).to(dev) if cd else None
    y = m(x, context=ctx).cpu()
    assert float((y - torch.from_numpy(g[f"{tag}_out"])).abs().max()) <= 2e-5
    if cd:
        mask = torch.from_numpy(g[f"{tag}_mask"])
        ym = m(x, context=ctx, mask=mask.to(dev)).cpu()
        assert float((ym - torch.from_numpy(g[f"{tag}_out_masked"])).abs().max()) <= 2e-5
        assert float((ym - y).abs().max()) > 1e-3           # the mask does something
        with pytest.raises(ValueError):
            m(x, context=ctx, mask=torch.zeros_like(mask).to(dev))
        with pytest.raises(ValueError):
            m(x, context=ctx[:, :, :64])
