#!/usr/bin/env python3
"""Per-launch table and wall time of the HIP autoencoder at the shipped geometry (256x256, 16 frames):
python tools/ae_profile.py [--extract] [--batch B]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moditalker_amd import BASE_AE_DDCONFIG, ViTAutoencoder, filler  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--extract", action="store_true")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--res", type=int, default=256)
args = ap.parse_args()
dev = torch.device("cuda:0")
ae = ViTAutoencoder(4, dict(BASE_AE_DDCONFIG, resolution=args.res), max_batch=args.batch).eval()
filler.fill_autoencoder_(ae, seed=1)
ae = ae.to(dev)
r = args.res // 8
B = args.batch
if args.extract:
    x = torch.rand(B, 3, 16, args.res, args.res, device=dev) * 2 - 1
    fn = lambda: ae.extract(x)
else:
    x = torch.rand(B, 4, r * r + 2 * 16 * r, device=dev) * 2 - 1
    fn = lambda: ae.decode_from_sample(x)
fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    fn()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
prof = ae.profile(B, args.extract, 3)
tot = sum(p["ms"] for p in prof)
fl = sum(p["flops"] for p in prof)
print(f"# {'extract' if args.extract else 'decode_from_sample'} B={B} res={args.res}: {dt * 1e3:.2f} ms per call (graph), {len(prof)} launches, "
      f"sum of launches {tot:.2f} ms, {fl / 1e12:.3f} TFLOP -> {fl / dt / 1e12:.1f} TFLOP/s")
fam = {}
for p in prof:
    k = p["name"].split(":")[0]
    f = fam.setdefault(k, [0, 0.0, 0.0])
    f[0] += 1; f[1] += p["ms"]; f[2] += p["flops"]
for k, (n, ms, f) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"#   {k:16s} n={n:4d} {ms:8.3f} ms" + (f"  {f / ms / 1e9:7.1f} TF/s" if f else ""))
for p in prof[:40]:
    print(f"{p['ms'] * 1e3:9.1f} us  {p['flops'] / max(p['ms'], 1e-9) / 1e9:7.1f} TF/s  {p['name']}")
