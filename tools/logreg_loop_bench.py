#!/usr/bin/env python
"""Developer: the device-resident loop on the reference README's logistic regression (BASELINE configs[0]: n = 1000, d = 32, mean-field,
n_mc = 16) and neighbours: steps/s of mivi_optimize_loop, Adam + ClipScale and the reference's default DoWG + PolynomialAveraging + ClipScale."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import advancedvi_jl_amd as avi
SHAPES = [tuple(int(x) for x in s.split(",")) for s in sys.argv[1:]] or [(0, 1000, 32, 16), (0, 1000, 32, 1), (1, 1000, 32, 16), (0, 10000, 64, 16)]
rng = np.random.default_rng(0)
for fam, n, d, M in SHAPES:
    X = rng.normal(size=(n, d - 1)).astype(np.float32)
    y = (rng.uniform(size=n) < 0.5).astype(np.float32)
    prob = avi.LogRegProblem(X, y)
    dd = prob.dimension()
    q = avi.MeanFieldGaussian(np.zeros(dd, np.float32), np.full(dd, 0.6, np.float32)) if fam == 0 else avi.FullRankGaussian(np.zeros(dd, np.float32), 0.6 * np.eye(dd, dtype=np.float32))
    p0, _ = avi.destructure(q)
    for name, rule, op, avg in (("Adam+ClipScale", 1, 1, 0), ("DoWG+ClipScale+PolynomialAveraging", 3, 1, 1)):
        ctx = avi.MiviContext(np.float32, fam, dd, M, 0, 1)
        ctx.set_problem(prob)
        p = ctx.to_device(p0).clone()
        if rule == 1:
            st = ctx.empty(2 * p.numel()).zero_()
        else:
            st = ctx.dog_state(); ctx.dog_init(p, st, 1e-6)
        avgp = p.clone() if avg else None
        T = 300
        kw = dict(rule=rule, op=op, averager=avg, eta=1e-3, clip_epsilon=1e-5, opt_state=st, avg_params=avgp)
        try:
            ctx.optimize_loop(p, T, 0, 0, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for r in range(3):
                ctx.optimize_loop(p, T, (r + 1) * T, (r + 1) * T, **kw)
            torch.cuda.synchronize()
        except Exception as e:   # noqa: BLE001
            print(f"family {fam} n={n} d={dd} n_mc={M} {name}: {e}", flush=True); ctx.close(); continue
        dt = time.perf_counter() - t0
        print(f"family {fam} n={n} d={dd} n_mc={M} {name}: {3*T/dt:,.0f} steps/s ({dt/(3*T)*1e6:.2f} us/step)", flush=True)
        ctx.close()
