#!/bin/bash
# Build libmtv_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function"
$HIPCC $FLAGS -c kernels.hip -o kernels.o "$@" &
$HIPCC $FLAGS -c conv.hip -o conv.o "$@" &
$HIPCC $FLAGS -c plan.hip -o plan.o "$@" &
$HIPCC $FLAGS -c ae.hip -o ae.o "$@" &
$HIPCC $FLAGS -c xattn.hip -o xattn.o "$@" &
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC kernels.o conv.o plan.o ae.o xattn.o -o libmtv_hip.so
echo "built $(pwd)/libmtv_hip.so"
if [ -n "$MTV_BUILD_STAMP" ]; then   # diagnostic twin with in-kernel phase timestamps (tools/stamps.py)
    $HIPCC $FLAGS -DMTV_ABLATE=64 -c conv.hip -o conv_stamp.o &
    $HIPCC $FLAGS -DMTV_ATT_STAMP -c kernels.hip -o kernels_stamp.o &
    wait
    $HIPCC --offload-arch=gfx950 -shared -fPIC kernels_stamp.o conv_stamp.o plan.o ae.o xattn.o -o libmtv_hip_stamp.so
    echo "built $(pwd)/libmtv_hip_stamp.so"
fi
