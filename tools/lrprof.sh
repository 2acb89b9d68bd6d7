REPO=$(pwd); export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/lrp
cat > /tmp/lr.py <<'PY'
import sys, warnings, numpy as np
sys.path.insert(0, sys.argv[1])
import advancedvi_jl_amd as avi
rng = np.random.default_rng(0)
n, p = int(sys.argv[2]), int(sys.argv[3])
X = (rng.normal(size=(n, p)) / np.sqrt(p)).astype(np.float32); y = (rng.uniform(size=n) < 0.5).astype(np.uint8)
alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), n_samples=16, optimizer=avi.Adam(1e-2), operator=avi.ClipScale(), averager=avi.NoAveraging())
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    avi.optimize(avi.PhiloxRNG(1), alg, 256, avi.LogRegProblem(X, y), avi.MeanFieldGaussian(np.zeros(p + 1, np.float32), np.ones(p + 1, np.float32)))
PY
for cfg in "20000 128" "1000 32"; do
rm -rf /tmp/lrp; rocprofv3 --kernel-trace --stats -d /tmp/lrp -o run -- python /tmp/lr.py $REPO $cfg > /tmp/lrp.log 2>&1
echo "== n p = $cfg"; python $REPO/tools/rocpd_stats.py $(find /tmp/lrp -name '*.db' | head -1) | cut -c1-130 | sed -n 3,12p
done
