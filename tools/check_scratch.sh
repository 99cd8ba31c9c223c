#!/bin/bash
# Lists every kernel of csrc/*.hip that uses scratch memory or spills registers (cross-compiles each file to gfx950 assembly; no GPU needed).
# A scratch access is a vector memory operation: it waits like one (`s_waitcnt vmcnt`) -- inside a prologue that has weight requests in
# flight that is a wait for all of them -- so new entries in this list deserve a look at the source.
cd "$(dirname "$0")/../moditalker_amd/csrc"
for f in kernels conv conv_x3 lin deep block ae xattn; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -S --cuda-device-only $f.hip -o /tmp/chk_$f.s 2>/dev/null || { echo "$f.hip: compile failed"; continue; }
    python3 - /tmp/chk_$f.s $f <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
n = 0
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)(?=\n  - \.agpr_count|\Z)", txt, re.S):
    name, blk = m.group(1), m.group(2)
    def g(k):
        r = re.search(r"\." + k + r":\s+(\d+)", blk)
        return int(r.group(1)) if r else 0
    ps, vs, ss, vg = g("private_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("vgpr_count")
    n += 1
    if ps or vs:
        print(f"{sys.argv[2]}.hip  {name}: scratch {ps} B, vgpr spills {vs}, sgpr spills {ss}, vgprs {vg}")
print(f"{sys.argv[2]}.hip: {n} kernels checked")
PY
done
