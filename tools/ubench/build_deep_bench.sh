#!/bin/bash
# builds tools/ubench/deep_bench (needs moditalker_amd/csrc/{conv_x3,lin}.o from build.sh); extra flags pass through, e.g.
#   OUT=deep_bench_stamp tools/ubench/build_deep_bench.sh -DMTV_DEEP_STAMP      (in-kernel phase timestamps)
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OBJ=$(mktemp --suffix=.o)            # (a fixed /tmp name would let concurrent builds collide)
trap 'rm -f "$OBJ"' EXIT
for o in conv_x3 lin; do
    [ -f ../../moditalker_amd/csrc/$o.o ] || { echo "build_deep_bench: moditalker_amd/csrc/$o.o missing -- run moditalker_amd/csrc/build.sh first" >&2; exit 1; }
done
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics "$@" -c deep_bench.hip -o "$OBJ"
$HIPCC --offload-arch=gfx950 "$OBJ" ../../moditalker_amd/csrc/conv_x3.o ../../moditalker_amd/csrc/lin.o -o ${OUT:-deep_bench}
