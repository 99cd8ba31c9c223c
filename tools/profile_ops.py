#!/usr/bin/env python3
"""Per-launch table of one base-config sampler step (or, --forward, one UNet forward); hipEvent timing around plain launches:
name [shape, tile], us, TFLOP/s, algorithmic GB/s.  Usage: python tools/profile_ops.py [--batch B] [--iters N]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_weights_  # noqa: E402
from moditalker_amd import BASE_UNET_CONFIG, DiffusionWrapper, UNetModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--res", type=int, default=32)
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--forward", action="store_true", help="profile UNetModel.forward's launches instead of one sampler step's")
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = dict(BASE_UNET_CONFIG, image_size=args.res)
net = DiffusionWrapper(UNetModel(**cfg, frames=args.frames, max_batch=args.batch)).eval().to(dev)
synth_weights_(net, dev, 1234)
B, R, T = args.batch, args.res, args.frames
L = R * R + 2 * T * R
x = torch.randn(B, 4, L, device=dev)
cond = torch.rand(B, 8, L, device=dev) * 2 - 1
ic = torch.rand(B, 4, R * R, device=dev) * 2 - 1
net(x, cond, ic, torch.full((B,), 500, device=dev))
um = net.diffusion_model
um.profile_forward(B, 2, step=not args.forward)
prof = um.profile_forward(B, args.iters, step=not args.forward)
tot = sum(p["ms"] for p in prof)
print(f"# {len(prof)} launches, sum {tot:.3f} ms")
for p in prof:
    us = p["ms"] * 1e3
    tf = p["flops"] / (p["ms"] * 1e-3) / 1e12 if p["flops"] and p["ms"] > 0 else 0
    gb = p["bytes"] / (p["ms"] * 1e-3) / 1e9 if p["bytes"] and p["ms"] > 0 else 0
    print(f"{us:9.2f} us  {tf:7.2f} TF/s  {gb:8.1f} GB/s  {p['name']}")
