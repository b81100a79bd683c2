#!/bin/bash
# Prints VGPR / SGPR / scratch / occupancy of every kernel in rt_backend.hip.
cd "$(dirname "$0")/../rust-raytracer_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
  -Rpass-analysis=kernel-resource-usage -c rt_backend.hip -o /tmp/rt_backend_res.o 2>&1 |
  grep "remark:" | sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//; s/.*remark: *//' |
  awk -F': ' '/Function Name/ {name=$2} /^VGPRs:/ {v=$2} /TotalSGPRs/ {s=$2} /ScratchSize/ {sc=$2} /Occupancy/ {o=$2} /LDS Size/ {print name, "vgpr="v, "sgpr="s, "scratch="sc, "waves/simd="o}' |
  sed 's/_ZN12_GLOBAL__N_1//'
