#!/bin/bash
# builds tools/ubench/deep_bench (needs moditalker_amd/csrc/{conv_x3,lin}.o from build.sh); extra flags pass through, e.g.
#   OUT=deep_bench_stamp tools/ubench/build_deep_bench.sh -DMTV_DEEP_STAMP      (in-kernel phase timestamps)
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics "$@" -c deep_bench.hip -o /tmp/${OUT:-deep_bench}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/${OUT:-deep_bench}.o ../../moditalker_amd/csrc/conv_x3.o ../../moditalker_amd/csrc/lin.o -o ${OUT:-deep_bench}
