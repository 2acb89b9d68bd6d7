#!/bin/bash
# LDS bank conflicts / LDS-array cycles of the VJP with the draws in both orientations (e0) and once (e1): separate counter passes
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for v in e0 e1; do
  cp $REPO/tools/bin/libmivi_$v.so $REPO/advancedvi.jl_amd/libmivi.so
  for c in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES; do
    rm -rf /tmp/p_$v_$c
    rocprofv3 --kernel-trace --pmc $c -d /tmp/p_${v}_$c -o run -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --concurrent 1 > /tmp/p.log 2>&1
    db=$(find /tmp/p_${v}_$c -name '*.db' | head -1)
    echo "== $v $c"; python $REPO/tools/rocpd_pmc.py $db | grep -E "k_fb_(vjp|prod|eps)" | cut -c1-160
  done
done
