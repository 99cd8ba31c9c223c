#!/usr/bin/env python3
"""In-kernel phase anatomy of every conv of one sampler step (diagnostic build):
   MTV_BUILD_STAMP=1 bash moditalker_amd/csrc/build.sh
   MTV_LIB=moditalker_amd/csrc/libmtv_hip_stamp.so MTV_STAMPS=1 python tools/stamps.py [--batch B] > stamps.txt
Prints, per conv, the phase durations (us at --mhz, default the ~2.1 GHz shader clock) of the sampled workgroup with the
longest entry-to-exit span."""
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
ap.add_argument("--mhz", type=float, default=2100.0, help="s_memtime tick rate (the shader clock, ~2.1 GHz under this load)")
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
names = ["args+decode", "tables", "barrier", "1st chunk", "ring issue", "prologue(GN)", "K loop", "reduce", "splitK+epilogue", "stats"]
order = [7, 0, 8, 9, 10, 1, 2, 3, 4, 5, 6]
# (s_memtime counters of different XCDs are not synchronised: only deltas INSIDE one workgroup mean anything)
print("# us per phase, of the sampled workgroup whose own entry->exit span is longest: " + " | ".join(names) + " || span")
tot = [0.0] * 11
for line in open(args.raw):
    if line.startswith("#"):
        continue
    parts = line.split()
    k = len(parts) - 64
    name = " ".join(parts[:k])
    v = [int(x) for x in parts[k:]]
    blocks = [v[i * 16:(i + 1) * 16] for i in range(4)]
    blocks = [b for b in blocks if b[7] > 0]
    if not blocks:
        continue
    last = max(blocks, key=lambda b: max(b) - b[7])
    seq = [last[i] for i in order]
    ph, prev = [], seq[0]
    for x in seq[1:]:
        if x == 0:
            ph.append(0.0)
            continue
        ph.append((x - prev) / args.mhz)
        prev = x
    span = (max(last) - last[7]) / args.mhz
    for i, x in enumerate(ph):
        tot[i] += x
    tot[10] += span
    print(f"{name:52s} " + " ".join(f"{x:6.2f}" for x in ph) + f" || {span:6.2f}")
print(f"{'# SUM':52s} " + " ".join(f"{x:6.1f}" for x in tot[:10]) + f" || {tot[10]:6.1f}")
