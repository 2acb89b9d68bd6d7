#!/bin/bash
# developer: A/B of the lane-batched kernels (k_fr_prod32q / k_fr_vjp32s) with tools/dbg/chains.py (long batches back to back, isolated 20-estimate calls)
run() { echo "== $*"; env "$@" python tools/dbg/chains.py 2>&1 | grep "chunk\|isolated"; }
for r in 1 2; do
run MIVI_DUMMY=1
run MIVI_STRIP_ROWS=1
run MIVI_VJP_STRIP=4
run MIVI_VJP_STRIP=2
done
