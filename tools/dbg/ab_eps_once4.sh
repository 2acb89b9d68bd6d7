#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
line() { python bench.py --no-cpu-baseline --no-also "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])"; }
for r in 1 2 3; do
  for v in e0 e1; do
    cp tools/bin/libmivi_$v.so advancedvi.jl_amd/libmivi.so
    echo "$v round $r: driver $(line --steps 20 --warmup 5) | ns $(line) | dense $(line --workload ns_dense) | stl $(line --workload ns_stl)"
  done
done
cd /tmp && export TMPDIR=/tmp
for v in e0 e1; do
  cp $REPO/tools/bin/libmivi_$v.so $REPO/advancedvi.jl_amd/libmivi.so
  rm -rf /tmp/t_$v
  rocprofv3 --kernel-trace --stats -d /tmp/t_$v -o run -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also --concurrent 1 > /tmp/t.log 2>&1
  echo "== $v"; python $REPO/tools/rocpd_stats.py $(find /tmp/t_$v -name '*.db' | head -1) 2>/dev/null | grep -E "k_fb_.*(740x1|128x21|320x1)" | cut -c1-140
done
cd $REPO; cp tools/bin/libmivi_e1.so advancedvi.jl_amd/libmivi.so
timeout 900 python -m pytest tests/test_gpu_each.py tests/test_gpu_batches.py tests/test_gpu_engine_fuzz.py -m gpu -x -q 2>&1 | tail -2
