#!/bin/bash
# On the GPU box: counters of the row-separable full-rank loop (k_fr_rows_loop) at one shape: VALU / LDS instruction counts, LDS bank
# conflicts and wait cycles, in separate rocprofv3 passes (kernel-trace + pmc only).  Usage: tools/pmc_rows.sh "1,1024,16,0"
set -u
SHAPE=${1:-1,1024,16,0}
REPO=$(pwd); OUT=$REPO/gpurun_out/summ; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for ctr in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM"; do
  i=$((i+1)); rm -rf /tmp/pmc_rows_$i
  rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_rows_$i -o run -- python $REPO/tools/small_loop_bench.py $SHAPE > /tmp/pmc_rows_$i.log 2>&1
  db=$(find /tmp/pmc_rows_$i -name '*.db' | head -1)
  if [ -n "$db" ]; then python $REPO/tools/rocpd_pmc.py $db | grep -E "k_fr_rows_loop|^\| kernel|^\|---"; else echo "(no database: $(tail -2 /tmp/pmc_rows_$i.log))"; fi
done | tee $OUT/pmc_rows.md
