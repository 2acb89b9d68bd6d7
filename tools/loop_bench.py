"""Device-resident SGD loop throughput (mivi_optimize_steps): steps/s for C2-shaped and bench/benchmarks.jl-shaped problems."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import advancedvi_jl_amd as avi
for name, d, M, fam in (("C2 mean-field d=1024 M=256", 1024, 256, 0), ("reference bench shape d=10 M=1 mean-field", 10, 1, 0),
                        ("C5 shard: funnel d=2048 M=64 mean-field STL", 2048, 64, 0), ("NS full-rank d=1024 M=256", 1024, 256, 1)):
    q = avi.MeanFieldGaussian(np.zeros(d, np.float32), np.ones(d, np.float32)) if fam == 0 else avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
    p0, _ = avi.destructure(q)
    funnel = "funnel" in name
    ctx = avi.MiviContext(np.float32, fam, d, M, 3 if funnel else 0, 1)
    ctx.set_problem(avi.FunnelProblem(d, 3.0) if funnel else avi.DiagNormalProblem(np.full(d, 5.0, np.float32), np.ones(d, np.float32)))
    p = ctx.to_device(p0).clone()
    st = ctx.empty(2 * p.numel()).zero_()
    T = 1000
    ctx.optimize_steps(p, st, 0, 0, T, 1, 1e-3, 1e-5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for r in range(reps):
        ctx.optimize_steps(p, st, (r + 1) * T, (r + 1) * T, T, 1, 1e-3, 1e-5)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {reps*T/dt:,.0f} Adam steps/s ({dt/(reps*T)*1e6:.2f} us/step), mu[0] -> {float(p[0]):.3f}")
    ctx.close()
