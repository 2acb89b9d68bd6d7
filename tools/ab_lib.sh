# same-box A/B of two libraries: advancedvi.jl_amd/libmivi_head.so vs the current libmivi.so (runs on the GPU box)
cd advancedvi.jl_amd; cp libmivi.so libmivi_new.so
B="python ../bench.py --no-cpu-baseline --concurrent 1 $*"
P='import json,sys; j=json.loads(sys.stdin.readline()); print(round(j["value"],1), round(j["ms_per_step"]*1e3,2), j["stage_us"])'
for r in 1 2; do
  cp libmivi_head.so libmivi.so; echo -n "head: "; $B 2>/dev/null | tail -1 | python -c "$P"
  cp libmivi_new.so libmivi.so;  echo -n "new : "; $B 2>/dev/null | tail -1 | python -c "$P"
done
