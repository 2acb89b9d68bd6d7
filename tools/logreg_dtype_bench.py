"""ms per estimate of the LogReg target, f32 vs f64: tools/logreg_dtype_bench.py [n] [p] [M]"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import advancedvi_jl_amd as avi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 511
M = int(sys.argv[3]) if len(sys.argv) > 3 else 128
d = p + 1
rng = np.random.default_rng(0)
X = (rng.normal(size=(n, p)) / np.sqrt(p)).astype(np.float32)
y = (rng.uniform(size=n) < 0.5).astype(np.uint8)
for dt in (np.float32, np.float64):
    q = avi.FullRankGaussian(np.zeros(d, dt), 0.6 * np.eye(d, dtype=dt))
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(dt, avi.FULLRANK, d, M, 0, 1)
    ctx.set_problem(avi.LogRegProblem(X.astype(dt), y))
    pd = ctx.to_device(params)
    v, g = ctx.empty(1), ctx.empty(ctx.params_len)
    for i in range(2):
        ctx.estimate_gradient(pd, i, v, g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 5
    for i in range(K):
        ctx.estimate_gradient(pd, 10 + i, v, g)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    fl = 4.0 * n * p * M
    print(f"{np.dtype(dt).name}: {ms:8.3f} ms / estimate  ({fl / ms / 1e9:7.1f} TFLOP/s on the two X contractions)", flush=True)
    ctx.close()
