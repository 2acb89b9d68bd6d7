#!/bin/bash
# On the GPU box: counters of the batch engine's kernels at a given lane count (tools/fb_lane_curve.py = mivi_profile_batch: back-to-back
# launches of each kernel alone), one rocprofv3 pass per counter group (kernel-trace + pmc only) -> gpurun_out/summ/<tag>_fb_pmc_L<lanes>.md
set -u
TAG=${1:-r0x}; L=${2:-50}
REPO=$(pwd); OUT=$REPO/gpurun_out/summ; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
F=$OUT/${TAG}_fb_pmc_L$L.md
{ echo "# $TAG: rocprofv3 --kernel-trace --pmc <counters> -- python tools/fb_lane_curve.py $L   (north-star shape, $L lanes per launch, one pass per counter group)"; echo; } > $F
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/fbp$i
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/fbp$i -o run -- python $REPO/tools/fb_lane_curve.py $L > /tmp/fbp$i.log 2>&1
  db=$(find /tmp/fbp$i -name '*.db' | head -1)
  if [ -n "$db" ]; then python $REPO/tools/rocpd_pmc.py $db | grep -E "k_fb_|^\| kernel|^\|---" >> $F; else echo "(pass $i: no database: $(tail -2 /tmp/fbp$i.log))" >> $F; fi
  echo >> $F
done
cat $F
