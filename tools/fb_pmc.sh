#!/bin/bash
# On the GPU box: SQ / TCC counters of the batch-engine kernels (own rocprofv3 passes, kernel-trace + pmc only) -> gpurun_out/fb_<tag>_pmc.md
set -u
TAG=${1:-x}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
{ echo "# $TAG: rocprofv3 --kernel-trace --pmc <counters> -- python tools/fb_check.py time   (one pass per counter group; 20- and 100-estimate batches)"; echo; } > $OUT/fb_${TAG}_pmc.md
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pmc_fb$i
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_fb$i -o run -- python $REPO/tools/fb_check.py time100 > /tmp/pmc_fb$i.log 2>&1
  db=$(find /tmp/pmc_fb$i -name '*.db' | head -1)
  if [ -n "$db" ]; then python $REPO/tools/rocpd_pmc.py $db | grep -E "k_fb_|^\| kernel|^\|---" >> $OUT/fb_${TAG}_pmc.md; else echo "(pass $i: no database: $(tail -2 /tmp/pmc_fb$i.log))" >> $OUT/fb_${TAG}_pmc.md; fi
  echo >> $OUT/fb_${TAG}_pmc.md
done
cat $OUT/fb_${TAG}_pmc.md
