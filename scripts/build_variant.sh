#!/bin/bash
# A/B measurement builds of the library: scripts/build_variant.sh NAME "-DFLAG ..." writes
# build_variants/libsdm_NAME.so (git-ignored; travels to the GPU box); use with SDM_HIP_LIB=...
# ONLY=index: the flags go to that translation unit alone (-DBIN_PROFILE: its device symbol
# cannot be shared with fused.hip, which uses the same tile sort).  A failed compile fails the build.
set -euo pipefail
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build_variants/libsdm_$name.so
tmp=$(mktemp -d)
mkdir -p "$root/build_variants"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
pids=()
for f in ctx index collisions fused displacement calib comm; do
  extra=("$@")
  if [ -n "${ONLY:-}" ] && [ "$f" != "$ONLY" ]; then extra=(); fi
  /opt/rocm/bin/hipcc $FLAGS "${extra[@]}" -c "$root/pysdm_amd/csrc/$f.hip" -o "$tmp/$f.o" 2>"$tmp/$f.log" &
  pids+=($!)
done
for pid in "${pids[@]}"; do wait "$pid" || { grep -h -A3 "error" "$tmp"/*.log | head -20; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o "$out" "$tmp"/*.o -ldl
rm -rf "$tmp"
echo "built $out"
