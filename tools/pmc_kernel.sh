#!/bin/bash
# On the GPU box: counters of ONE kernel under a command, in separate rocprofv3 passes (kernel-trace + pmc only).
# usage: tools/pmc_kernel.sh <kernel-name-substring> <command ...>
set -u
KSUB=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/summ; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for ctr in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM"; do
  i=$((i+1)); rm -rf /tmp/pmc_k_$i
  ( cd $REPO && rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_k_$i -o run -- "$@" > /tmp/pmc_k_$i.log 2>&1 )
  db=$(find /tmp/pmc_k_$i -name '*.db' | head -1)
  if [ -n "$db" ]; then python $REPO/tools/rocpd_pmc.py $db | grep -E "$KSUB|^\| kernel|^\|---"; else echo "(no database: $(tail -2 /tmp/pmc_k_$i.log))"; fi
done | tee $OUT/pmc_kernel.md
