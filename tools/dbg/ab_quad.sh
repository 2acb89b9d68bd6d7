#!/bin/bash
# developer: A/B of the lane-batched kernels (k_fr_prod32q / k_fr_vjp32s) with tools/dbg/chains.py (long batches back to back, isolated 20-estimate calls)
run() { echo "== $*"; env "$@" python tools/dbg/chains.py 2>&1 | grep "chunk\|isolated"; }
for r in 1 2; do
run MIVI_PROD_QUAD=1 MIVI_VJP_STRIP=4
run MIVI_PROD_QUAD=0 MIVI_VJP_STRIP=0
run MIVI_PROD_QUAD=1 MIVI_VJP_STRIP=0
run MIVI_PROD_QUAD=1 MIVI_VJP_STRIP=3
run MIVI_PROD_QUAD=0 MIVI_VJP_STRIP=3
done
