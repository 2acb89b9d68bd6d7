# developer, on the GPU box: same-box A/B of advancedvi.jl_amd/libmivi_head.so against the current libmivi.so with tools/dbg/chains.py + stage times
cd advancedvi.jl_amd; cp libmivi.so libmivi_new.so; cd ..
for r in 1 2; do
  for v in head new; do
    cp advancedvi.jl_amd/libmivi_$v.so advancedvi.jl_amd/libmivi.so
    echo "== $v"; python tools/stage_times.py ${1:-ns} 2>&1 | tail -1; python tools/dbg/chains.py 2>&1 | grep -v amdgpu.ids | grep "chunk\|isolated"
  done
done
cp advancedvi.jl_amd/libmivi_new.so advancedvi.jl_amd/libmivi.so
