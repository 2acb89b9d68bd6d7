cd advancedvi.jl_amd; cp libmivi.so libmivi_new.so
for r in 1 2; do
  cp libmivi_head.so libmivi.so; echo "head:"; PYTHONPATH=.. python ../tools/objective_bench.py 2>/dev/null | grep "full-rank"
  cp libmivi_new.so libmivi.so;  echo "new :"; PYTHONPATH=.. python ../tools/objective_bench.py 2>/dev/null | grep "full-rank"
done
cp libmivi_new.so libmivi.so
