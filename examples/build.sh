#!/bin/bash
# builds the C-ABI example next to its source (links libsdm_hip.so of the package, found at run time
# through an $ORIGIN-relative rpath)
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -O2 -std=c++17 --offload-arch=gfx950 shima_box_c_abi.cpp -o shima_box_c_abi \
  -L../pysdm_amd -lsdm_hip -Wl,-rpath,'$ORIGIN/../pysdm_amd'
echo "built $(realpath shima_box_c_abi)"
