# developer, on the GPU box: same-box A/B of libmivi_head.so against libmivi.so on the mean-field bench workloads
cd advancedvi.jl_amd; cp libmivi.so libmivi_new.so; cd ..
for r in 1 2; do for v in head new; do cp advancedvi.jl_amd/libmivi_$v.so advancedvi.jl_amd/libmivi.so; for w in c2 c5; do
  python3 bench.py --no-cpu-baseline --workload $w 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$v $w', round(j['value']), round(j['ms_per_step']*1e3,3))"; done; done; done
cp advancedvi.jl_amd/libmivi_new.so advancedvi.jl_amd/libmivi.so
