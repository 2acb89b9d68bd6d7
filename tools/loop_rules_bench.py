#!/usr/bin/env python
"""Developer: the device-resident loop with the reference's DEFAULT algorithm settings (KLMinRepGradDescent: DoWG + PolynomialAveraging +
ClipScale, src/algorithms/constructors.jl:44-120) beside Adam + ClipScale, diagonal-Gaussian target: steps/s of mivi_optimize_loop.
  python tools/loop_rules_bench.py fam,d,M ...   (fam 0 mean-field, 1 full-rank)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import advancedvi_jl_amd as avi
SHAPES = [tuple(int(x) for x in s.split(",")) for s in sys.argv[1:]] or [(1, 1024, 1), (1, 1024, 8), (1, 1024, 256), (0, 1024, 1), (0, 1024, 256), (1, 10, 1)]
for fam, d, M in SHAPES:
    q = avi.MeanFieldGaussian(np.zeros(d, np.float32), np.ones(d, np.float32)) if fam == 0 else avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
    p0, _ = avi.destructure(q)
    for name, rule, op, avg in (("Adam+ClipScale", 1, 1, 0), ("DoWG+ClipScale+PolynomialAveraging", 3, 1, 1), ("DoWG+Prox+PolynomialAveraging", 3, 2, 1), ("Descent+ClipScale", 0, 1, 0)):
        ctx = avi.MiviContext(np.float32, fam, d, M, 0 if op != 2 else 1, 1)
        ctx.set_problem(avi.DiagNormalProblem(np.full(d, 5.0, np.float32), np.ones(d, np.float32)))
        p = ctx.to_device(p0).clone()
        if rule == 1:
            st = ctx.empty(2 * p.numel()).zero_()
        elif rule >= 2:
            st = ctx.dog_state()
            ctx.dog_init(p, st, 1e-6)
        else:
            st = None
        avgp = p.clone() if avg else None
        T = 500
        kw = dict(rule=rule, op=op, averager=avg, eta=1e-3, clip_epsilon=1e-5, opt_state=st, avg_params=avgp)
        try:
            ctx.optimize_loop(p, T, 0, 0, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for r in range(3):
                ctx.optimize_loop(p, T, (r + 1) * T, (r + 1) * T, **kw)
            torch.cuda.synchronize()
        except Exception as e:   # noqa: BLE001 -- a diverged run still timed what it ran
            print(f"family {fam} d={d} n_mc={M} {name}: {e}", flush=True)
            ctx.close()
            continue
        dt = time.perf_counter() - t0
        print(f"family {fam} d={d} n_mc={M} {name}: {3*T/dt:,.0f} steps/s ({dt/(3*T)*1e6:.2f} us/step), mu[0] -> {float(p[0]):.3f}", flush=True)
        ctx.close()
