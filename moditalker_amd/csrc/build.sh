#!/bin/bash
# Build libmtv_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function"
pids=()
for f in kernels conv conv_x3 lin deep block plan ae xattn; do
    $HIPCC $FLAGS -c $f.hip -o $f.o "$@" &
    pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done      # (a failed compile fails the build: `wait` alone would link a stale object)
$HIPCC --offload-arch=gfx950 -shared -fPIC kernels.o conv.o conv_x3.o lin.o deep.o block.o plan.o ae.o xattn.o -o libmtv_hip.so
echo "built $(pwd)/libmtv_hip.so"
if [ -n "$MTV_BUILD_STAMP" ]; then   # diagnostic twin with in-kernel phase timestamps (tools/stamps.py)
    $HIPCC $FLAGS -DMTV_ABLATE=64 -c conv.hip -o conv_stamp.o & p1=$!
    $HIPCC $FLAGS -DMTV_ATT_STAMP -c kernels.hip -o kernels_stamp.o & p2=$!
    wait $p1; wait $p2
    $HIPCC --offload-arch=gfx950 -shared -fPIC kernels_stamp.o conv_stamp.o conv_x3.o lin.o deep.o block.o plan.o ae.o xattn.o -o libmtv_hip_stamp.so
    echo "built $(pwd)/libmtv_hip_stamp.so"
fi
