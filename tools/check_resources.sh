#!/bin/bash
# Register / occupancy report of the headline kernels (hipcc remarks; no GPU needed).  The VJP tile kernel must stay at
# <= 128 VGPRs (3 waves/SIMD) and the mean-field kernel at 7 waves/SIMD: an inlined helper that raises them costs ~1 us.
cd "$(dirname "$0")/../advancedvi.jl_amd/csrc"
for f in kernels_fullrank.hip kernels_meanfield.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/_res.o 2>&1 |
    grep -E "Function Name|VGPRs:|AGPRs:|Occupancy|LDS Size" |
    awk '/Function Name/ {name=$5} / VGPRs:/ {v=$4} /AGPRs:/ {a=$4} /Occupancy/ {o=$5} /LDS Size/ {printf "%-70s VGPR %4s AGPR %4s occupancy %s LDS %s\n", name, v, a, o, $6}' |
    grep -E "k_fr_tile_mfmaILi[012]ELi[48]ELb1|k_mf_mainIf|k_mf_sgd_loopIf"
done
