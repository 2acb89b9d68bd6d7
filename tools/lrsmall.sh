REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
cat > /tmp/lr2.py <<'PY'
import sys, time, warnings, numpy as np, torch
sys.path.insert(0, sys.argv[1])
import advancedvi_jl_amd as avi
rng = np.random.default_rng(0)
for n, p, M in [(1000, 32, 16), (1000, 32, 128), (20000, 128, 16), (20000, 128, 128), (100000, 64, 32), (5000, 511, 64)]:
    X = (rng.normal(size=(n, p)) / np.sqrt(p)).astype(np.float32); y = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    d = p + 1
    q = avi.MeanFieldGaussian(np.zeros(d, np.float32), np.ones(d, np.float32))
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, 0, d, M, 0, 1)
    ctx.set_problem(avi.LogRegProblem(X, y))
    pd = ctx.to_device(params); v, g = ctx.empty(1), ctx.empty(ctx.params_len)
    for i in range(5): ctx.estimate_gradient(pd, i, v, g)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(200): ctx.estimate_gradient(pd, 10 + i, v, g)
    torch.cuda.synchronize()
    print(f"n={n} p={p} M={M}: {(time.perf_counter() - t0) / 200 * 1e6:8.1f} us/estimate", flush=True)
    ctx.close()
PY
echo "== default (MFMA route)"; python /tmp/lr2.py $REPO 2>&1 | grep us/est
echo "== MIVI_LOGREG_GENERIC=1"; MIVI_LOGREG_GENERIC=1 python /tmp/lr2.py $REPO 2>&1 | grep us/est
