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
attn = []
for line in open(args.raw):
    if line.startswith("#"):
        continue
    parts = line.split()
    k = len(parts) - 64
    name = " ".join(parts[:k])
    v = [int(x) for x in parts[k:]]
    blocks = [v[i * 16:(i + 1) * 16] for i in range(4)]
    if name.startswith("attn:"):
        attn.append((name, blocks))
        continue
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

# attention launches (kernels.hip -DMTV_ATT_STAMP): wave 0 of the sampled workgroup with the longest span
if attn:
    print("# attention, us: args+decode | first tile (loads -> LDS -> barrier) | key loop = block math + next-tile store (incl. load wait) "
          "+ barrier | merge | store || span")
    at = [0.0] * 9
    for name, blocks in attn:
        blocks = [b for b in blocks if b[0] > 0 and b[5] > 0]
        if not blocks:
            continue
        b = max(blocks, key=lambda x: x[5] - x[0])
        us = lambda x: x / args.mhz
        row = [us(b[1] - b[0]), us(b[2] - b[1]), us(b[3] - b[2]), us(b[8]), us(b[9]), us(b[10]), us(b[4] - b[3]), us(b[5] - b[4]), us(b[5] - b[0])]
        for i, x in enumerate(row):
            at[i] += x
        print(f"{name:40s} {row[0]:6.2f} {row[1]:6.2f} {row[2]:6.2f} = {row[3]:6.2f} + {row[4]:6.2f} + {row[5]:6.2f} | {row[6]:6.2f} {row[7]:6.2f} || {row[8]:6.2f}")
    print(f"{'# SUM attention':40s} {at[0]:6.1f} {at[1]:6.1f} {at[2]:6.1f} = {at[3]:6.1f} + {at[4]:6.1f} + {at[5]:6.1f} | {at[6]:6.1f} {at[7]:6.1f} || {at[8]:6.1f}")
