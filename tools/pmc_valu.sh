#!/bin/bash
# On the GPU box: vector-ALU counters of the launch-free mean-field loops (C2: k_mf_sgd_loop, C5: k_mf_funnel_loop) -- one rocprofv3 pass
# (kernel-trace + pmc only) per workload -> gpurun_out/summ/<tag>_<w>_pmc_valu.md and gpurun_out/summ/pmc_valu.json (copied to profiles/:
# bench.py's VALU roofline reads the wave-level instruction count per launch from it).
set -u
TAG=${1:-r0x}
REPO=$(pwd); OUT=$REPO/gpurun_out/summ; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for w in c2 c5; do
  rm -rf /tmp/pmc_valu_$w
  { echo "# $TAG: rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -- python bench.py --workload $w --steps 2000 --warmup 200 --no-cpu-baseline --no-also --concurrent 1"; echo; } > $OUT/${TAG}_${w}_pmc_valu.md
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d /tmp/pmc_valu_$w -o run -- python $REPO/bench.py --workload $w --steps 2000 --warmup 200 --no-cpu-baseline --no-also --concurrent 1 > /tmp/pmc_valu_$w.log 2>&1
  db=$(find /tmp/pmc_valu_$w -name '*.db' | head -1)
  if [ -n "$db" ]; then python $REPO/tools/rocpd_pmc.py $db | grep -E "k_mf_|^\| kernel|^\|---" >> $OUT/${TAG}_${w}_pmc_valu.md; else echo "(no database: $(tail -2 /tmp/pmc_valu_$w.log))" >> $OUT/${TAG}_${w}_pmc_valu.md; fi
done
python - "$OUT" "$TAG" <<'PY'
import json, re, sys
out, tag = sys.argv[1], sys.argv[2]
tab = {}
for w in ("c2", "c5"):
    try:
        for line in open(f"{out}/{tag}_{w}_pmc_valu.md"):
            c = [x.strip() for x in line.strip().strip("|").split("|")]
            if len(c) >= 6 and "k_mf_" in c[0]:
                k = c[0].strip("`")
                tab.setdefault(k, {})[c[1]] = float(c[2])
                tab[k]["avg_ns"] = float(c[5])
                tab[k]["dispatches"] = int(c[4])
    except OSError:
        pass
json.dump(dict(source=f"tools/pmc_valu.sh {tag} (rocprofv3 --pmc, sum over hardware instances per dispatch, averaged over dispatches)", kernels=tab),
          open(f"{out}/pmc_valu.json", "w"), indent=1)
print(json.dumps(tab, indent=1))
PY
