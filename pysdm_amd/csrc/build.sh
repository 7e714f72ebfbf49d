#!/bin/bash
# Builds libsdm_hip.so (gfx950) next to the Python package.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libsdm_hip.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
objs=()
for f in ctx index collisions fused; do
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ common.h -nt $f.o ] || [ physics.h -nt $f.o ] || [ index.h -nt $f.o ] || [ ../../include/sdm_hip.h -nt $f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o $f.o &
  fi
  objs+=($f.o)
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT "${objs[@]}"
echo "built $(realpath $OUT)"
