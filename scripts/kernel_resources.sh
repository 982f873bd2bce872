#!/usr/bin/env bash
# Registers / occupancy / LDS of every kernel, from the compiler's resource-usage remarks.
# usage: scripts/kernel_resources.sh [source.hip]
src=${1:-mmsbm_amd/csrc/unity.hip}   # all translation units as one (the product compiles them side by side)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared \
  -Rpass-analysis=kernel-resource-usage -o /dev/null "$src" 2>&1 |
awk '/Function Name:/ {name=$(NF-1)} / VGPRs:/ {v=$(NF-1)} /TotalSGPRs:/ {s=$(NF-1)} /ScratchSize/ {sc=$(NF-1)}
     /Occupancy/ {o=$(NF-1)} /LDS Size/ {printf "%-110s sgpr %3s vgpr %3s scratch %s occ %s lds %s\n", name, s, v, sc, o, $(NF-1)}' |
sed 's/_ZN12_GLOBAL__N_1[0-9]*//' | c++filt 2>/dev/null | sort -u
