#!/bin/bash
# Dump the gfx950 ISA of conv.hip and summarise one k_conv<MT,NT,NW> instantiation: loads, waits, barriers.
# Usage: tools/isa_conv.sh MT NT NW [first_line last_line]
set -e
D=/root/repo/moditalker_amd/csrc
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -S --cuda-device-only $D/conv.hip -o /tmp/isa/conv.s 2>&1 | grep -E "error" || true
K="_ZN3mtv6k_convILi$1ELi$2ELi$3EEEvNS_8ConvArgsE"
awk -v k="$K" '$0 ~ "^"k":"{p=1} p{print} $0 ~ "amdhsa_kernel "k{exit}' /tmp/isa/conv.s > /tmp/isa/k.s
grep -n "private_seg_size, [1-9]" /tmp/isa/conv.s || true
grep -n "s_waitcnt vmcnt\|s_barrier\|global_load\|global_store\|global_atomic\|ds_bpermute\|v_mfma\|v_div_scale_f64" /tmp/isa/k.s | awk -F: '{print $1": "$2}' | awk '{k=$2" "$3; if (k==last) {n++} else { if (n>0) print "   x"n+1; print; n=0}; last=k}' | sed -n "${4:-1},${5:-80}p"
