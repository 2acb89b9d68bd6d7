# A/B over the LogReg kernel variants: per-kernel avg time from rocprofv3 kernel-trace (runs on the GPU box)
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
for v in "$@"; do
  rm -rf /tmp/c3v; env $v rocprofv3 --kernel-trace --stats -d /tmp/c3v -o run -- python $REPO/bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline --concurrent 1 > /tmp/c3v.log 2>&1
  echo "== $v: $(tail -1 /tmp/c3v.log | python -c 'import json,sys; print(round(json.loads(sys.stdin.read())["ms_per_step"],3))') ms/step"
  python $REPO/tools/rocpd_stats.py $(find /tmp/c3v -name '*.db' | head -1) | grep -E "k_lr_(logits|xtr)" | awk -F'|' '{print $2, $5}' | cut -c1-90
done
