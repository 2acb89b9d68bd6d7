#!/bin/bash
# developer, on the GPU box: same-box A/B of two builds (advancedvi.jl_amd/libmivi_<name>.so, names as arguments) with tools/dbg/chains.py
cp advancedvi.jl_amd/libmivi.so /tmp/libmivi_keep.so
for r in 1 2; do
  for v in "$@"; do
    cp advancedvi.jl_amd/libmivi_$v.so advancedvi.jl_amd/libmivi.so
    echo "== $v"; python tools/dbg/chains.py 2>&1 | grep "chunk\|isolated"
  done
done
cp /tmp/libmivi_keep.so advancedvi.jl_amd/libmivi.so
