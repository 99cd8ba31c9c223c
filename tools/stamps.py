#!/usr/bin/env python3
"""In-kernel phase anatomy of every conv of one sampler step (diagnostic build):
   MTV_BUILD_STAMP=1 bash moditalker_amd/csrc/build.sh
   MTV_LIB=moditalker_amd/csrc/libmtv_hip_stamp.so MTV_STAMPS=1 python tools/stamps.py [--batch B] > stamps.txt
Prints, per conv, the phase durations (us, assuming a 100 MHz s_memtime reference unless --mhz is given) of the
workgroup that finished last among the sampled ones, and the spread of workgroup start times."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_weights_  # noqa: E402
from moditalker_amd import BASE_UNET_CONFIG, DiffusionWrapper, UNetModel, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--res", type=int, default=32)
ap.add_argument("--mhz", type=float, default=100.0, help="s_memtime tick rate")
ap.add_argument("--raw", default="/tmp/stamps_raw.txt")
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = dict(BASE_UNET_CONFIG, image_size=args.res)
net = DiffusionWrapper(UNetModel(**cfg, frames=16, max_batch=args.batch)).eval().to(dev)
synth_weights_(net, dev, 1234)
um = net.diffusion_model
ctx = um.hip_context(dev, args.batch)
lib = _lib.load()
_lib.check(lib.mtv_debug_stamps(ctx, args.batch, args.raw.encode(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "mtv_debug_stamps")
names = ["args+decode", "segs+ring issue", "prologue(GN)", "K loop", "reduce", "splitK+epilogue", "stats"]
order = [7, 0, 1, 2, 3, 4, 5, 6]
print("# us per phase of the last-finishing sampled workgroup: " + " | ".join(names) + " || total, start spread over sampled blocks")
tot = [0.0] * 9
for line in open(args.raw):
    if line.startswith("#"):
        continue
    parts = line.split()
    k = len(parts) - 32
    name = " ".join(parts[:k])
    v = [int(x) for x in parts[k:]]
    blocks = [v[i * 8:(i + 1) * 8] for i in range(4)]
    blocks = [b for b in blocks if b[7] > 0]
    if not blocks:
        continue
    def end(b):
        return max(b)
    last = max(blocks, key=end)
    seq = [last[i] for i in order]
    ph = []
    prev = seq[0]
    for x in seq[1:]:
        if x == 0:
            ph.append(0.0)
            continue
        ph.append((x - prev) / args.mhz)
        prev = x
    total = (end(last) - min(b[7] for b in blocks)) / args.mhz
    spread = (max(b[7] for b in blocks) - min(b[7] for b in blocks)) / args.mhz
    for i, x in enumerate(ph):
        tot[i] += x
    tot[8] += total
    print(f"{name:52s} " + " ".join(f"{x:6.2f}" for x in ph) + f" || {total:6.2f} {spread:5.2f}")
print(f"{'# SUM':52s} " + " ".join(f"{x:6.1f}" for x in tot[:7]) + f" || {tot[8]:6.1f}")
