#!/bin/bash
# developer (DEV build: make DEV=1): durations of the four-lane kernels in an isolated 20-estimate call with parts knocked out
# (MIVI_KNOCK bits: 2 no split / MFMA, 4 no operand loads, 16 no epilogue, 32 no epilogue arithmetic / stores, 64 no eps riders, 128 no mirrored zero tile,
#  256 no gradient store, 1024 one tile per strip, 2048 strips return at once); KS = the list of values
for k in ${KS:-0 64 66 68 80 70}; do echo "KNOCK=$k"; MIVI_KNOCK=$k tools/dbg/iso20_prof.sh 12 | grep -E "prod|vjp" | tail -6 | awk '{print $1, $5}' | tr '\n' ' '; echo; done
