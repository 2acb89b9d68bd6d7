#!/bin/bash
# developer (DEV build): steady-state us per estimate with parts of the lane-batched kernels knocked out (MIVI_KNOCK bits: 2 no split / MFMA,
# 4 no operand loads, 16 no epilogue, 64 no eps riders, 128 no mirrored zero tile, 256 no gradient store)
for k in 0 64 2 4 6 16 70 86 384; do echo -n "KNOCK=$k  "; MIVI_KNOCK=$k python tools/dbg/chains.py 2>&1 | grep "chunk 100" | awk '{print $4}'; done
