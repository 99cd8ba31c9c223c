#!/usr/bin/env python3
"""Golden vectors for the cross-attention operator from the REFERENCE's CrossAttention class
(MToV/models/ddpm/unet.py:429-467, imported unmodified with make_golden.py's harness shims; build container only).
Cases: cross (context 77 x 256, the usual text/landmark-token shape) with and without a key mask, and self-attention
(context=None).  Output: tests/golden/xattn.npz; cross-checks oracle/ref_xattn.py."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True
from make_golden import import_reference  # noqa: E402
from moditalker_amd import filler  # noqa: E402
from oracle import ref_xattn  # noqa: E402

CASES = [  # tag, query_dim, context_dim, heads, dim_head, B, N, M
    ("cross", 256, 128, 8, 32, 2, 96, 77),
    ("self", 128, None, 4, 64, 1, 160, None),
]


def main():
    import_reference()
    from models.ddpm.unet import CrossAttention
    out, report = {}, []
    for tag, qd, cd, H, d, B, N, M in CASES:
        m = CrossAttention(qd, cd, heads=H, dim_head=d).eval()
        filler.fill_module_(m, seed=51)
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        x = filler.uniform_pm1(f"xattn.{tag}.x", (B, N, qd), 51)
        ctx = filler.uniform_pm1(f"xattn.{tag}.ctx", (B, M, cd), 51) if cd else None
        with torch.no_grad():
            y = m(x, context=ctx)
        yo = ref_xattn.cross_attention(sd, x, ctx, None, H)
        report.append((f"xattn {tag}", float((yo - y).abs().max())))
        out[f"{tag}_out"] = y.numpy()
        if cd:
            mask = torch.from_numpy(filler.uniform01(f"xattn.{tag}.mask", B * M, 51).reshape(B, M) > 0.3)
            mask[:, 0] = True
            with torch.no_grad():
                ym = m(x, context=ctx, mask=mask)
            yom = ref_xattn.cross_attention(sd, x, ctx, mask, H)
            report.append((f"xattn {tag} masked", float((yom - ym).abs().max())))
            out[f"{tag}_mask"] = mask.numpy()
            out[f"{tag}_out_masked"] = ym.numpy()
        if tag == "cross":
            out["keys"] = np.array(list(sd.keys()))
            out["shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
    for k, dd in report:
        print(f"  oracle vs reference  {k:30s} max-abs {dd:.3e}")
        assert dd <= 2e-6, (k, dd)
    np.savez_compressed(os.path.join(HERE, "xattn.npz"), **out)
    with open(os.path.join(HERE, "PIN_REPORT.txt"), "a") as f:
        f.write("# oracle/ref_xattn.py vs imported reference CrossAttention (make_golden_xattn.py)\n")
        for k, dd in report:
            f.write(f"{k:44s} {dd:.3e}\n")
    print("xattn.npz written")


if __name__ == "__main__":
    main()
