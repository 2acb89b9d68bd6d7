#!/usr/bin/env python
"""Developer: the device-resident Adam loop at the reference's own benchmark shapes (bench/benchmarks.jl:43-94: normal target d = 10, one sample
per step, mean-field and full-rank, ClosedFormEntropy and STL) and a few small neighbours: steps/s of mivi_optimize_steps."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import advancedvi_jl_amd as avi
SHAPES = [tuple(int(x) for x in s.split(",")) for s in sys.argv[1:]] or [(0, 10, 1, 0), (1, 10, 1, 0), (1, 10, 1, 3), (1, 10, 8, 0), (1, 32, 16, 0), (1, 64, 32, 0)]
for fam, d, M, ent in SHAPES:
    q = avi.MeanFieldGaussian(np.zeros(d, np.float32), np.ones(d, np.float32)) if fam == 0 else avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
    p0, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, fam, d, M, ent, 1)
    ctx.set_problem(avi.DiagNormalProblem(np.full(d, 5.0, np.float32), np.ones(d, np.float32)))
    p = ctx.to_device(p0).clone()
    st = ctx.empty(2 * p.numel()).zero_()
    T = 1000
    ctx.optimize_steps(p, st, 0, 0, T, 1, 1e-3, 1e-5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(5):
        ctx.optimize_steps(p, st, (r + 1) * T, (r + 1) * T, T, 1, 1e-3, 1e-5)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"family {fam} d={d} n_mc={M} entropy {ent}: {5*T/dt:,.0f} Adam steps/s ({dt/(5*T)*1e6:.2f} us/step), mu[0] -> {float(p[0]):.3f}", flush=True)
    ctx.close()
