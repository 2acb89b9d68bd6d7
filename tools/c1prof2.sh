REPO=$(pwd); export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/c1p
cat > /tmp/c1.py <<'PY'
import sys, warnings, numpy as np
sys.path.insert(0, sys.argv[1])
import advancedvi_jl_amd as avi
rng = np.random.default_rng(0)
X = rng.normal(size=(1000, 32)); y = (rng.uniform(size=1000) < 0.5).astype(np.uint8)
alg = avi.KLMinRepGradDescent(avi.AutoMIVI(), n_samples=16, optimizer=avi.DoWG(), operator=avi.ClipScale())
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    avi.optimize(avi.PhiloxRNG(1), alg, 512, avi.LogRegProblem(X, y), avi.MeanFieldGaussian(np.zeros(33), np.ones(33)))
PY
rocprofv3 --kernel-trace --stats -d /tmp/c1p -o run -- python /tmp/c1.py $REPO > /tmp/c1p.log 2>&1
python $REPO/tools/rocpd_stats.py $(find /tmp/c1p -name '*.db' | head -1) | cut -c1-140 | sed -n 1,14p
