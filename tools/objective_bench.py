"""estimate_objective at monitoring sample counts (callers use n up to 1e5 and beyond): wall time per call, samples per second."""
import time, numpy as np, torch, advancedvi_jl_amd as avi
SEED = 0x38BEF07CF9CC549D
for fam, d, name in ((avi.FULLRANK, 1024, "full-rank d=1024"), (avi.MEANFIELD, 1024, "mean-field d=1024")):
    q = avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32)) if fam == avi.FULLRANK else avi.MeanFieldGaussian(np.zeros(d, np.float32), np.ones(d, np.float32))
    p0, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, fam, d, 256, 1, SEED)
    ctx.set_problem(avi.DiagNormalProblem(np.full(d, 5.0, np.float32), np.ones(d, np.float32)))
    p = ctx.to_device(p0)
    for n in (10_000, 100_000, 1_000_000):
        v = ctx.estimate_objective(p, 1, n_samples=n); ctx.synchronize()
        t0 = time.perf_counter(); R = 5
        for r in range(R): v = ctx.estimate_objective(p, 2 + r, n_samples=n)
        ctx.synchronize(); dt = (time.perf_counter() - t0) / R
        print(f"{name}: n_samples={n}: {dt*1e3:.3f} ms per call, {n/dt/1e6:.1f} M samples/s, value {float(v.item()):.4f}")
    ctx.close()
