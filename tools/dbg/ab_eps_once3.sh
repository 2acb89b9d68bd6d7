#!/bin/bash
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
for v in e0 e1 e2; do
  cp $REPO/tools/bin/libmivi_$v.so $REPO/advancedvi.jl_amd/libmivi.so
  rm -rf /tmp/t_$v
  rocprofv3 --kernel-trace --stats -d /tmp/t_$v -o run -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --concurrent 1 > /tmp/t.log 2>&1
  echo "== $v"; python $REPO/tools/rocpd_stats.py $(find /tmp/t_$v -name '*.db' | head -1) 2>/dev/null | grep -E "k_fb_" | cut -c1-140
  c=SQ_LDS_BANK_CONFLICT
  rocprofv3 --kernel-trace --pmc $c -d /tmp/p_${v}_$c -o run -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --concurrent 1 > /tmp/p.log 2>&1
  python $REPO/tools/rocpd_pmc.py $(find /tmp/p_${v}_$c -name '*.db' | head -1) | grep -E "k_fb_vjp" | cut -c1-160
done
