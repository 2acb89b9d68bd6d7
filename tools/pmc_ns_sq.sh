#!/bin/bash
# On the GPU box: SQ counters of the north-star kernels (own rocprofv3 passes, kernel-trace + pmc only) -> gpurun_out/summ/<tag>_ns_pmc_sq.md
set -u
TAG=${1:-r0x}
REPO=$(pwd); OUT=$REPO/gpurun_out/summ; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
{ echo "# $TAG: rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --steps 200 --warmup 100 --no-cpu-baseline --no-also --concurrent 1   (one pass per counter group)"; echo; } > $OUT/${TAG}_ns_pmc_sq.md
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1)); rm -rf /tmp/pmc_sq$i
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_sq$i -o run -- python $REPO/bench.py --steps 200 --warmup 100 --no-cpu-baseline --no-also --concurrent 1 > /tmp/pmc_sq$i.log 2>&1
  db=$(find /tmp/pmc_sq$i -name '*.db' | head -1)
  if [ -n "$db" ]; then python $REPO/tools/rocpd_pmc.py $db | grep -E "k_fb_|k_fr_prod32|k_fr_vjp32|^\| kernel|^\|---" >> $OUT/${TAG}_ns_pmc_sq.md; else echo "(pass $i: no database: $(tail -2 /tmp/pmc_sq$i.log))" >> $OUT/${TAG}_ns_pmc_sq.md; fi
  echo >> $OUT/${TAG}_ns_pmc_sq.md
done
cat $OUT/${TAG}_ns_pmc_sq.md
