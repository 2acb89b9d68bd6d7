"""us per estimate for f32 vs f64 contexts (single calls, hipEvent-timed): tools/dtype_bench.py [d] [M]"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import advancedvi_jl_amd as avi
d = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
M = int(sys.argv[2]) if len(sys.argv) > 2 else 256
for dt in (np.float32, np.float64):
    for fam, name in ((avi.MEANFIELD, "meanfield"), (avi.FULLRANK, "fullrank")):
        for ent in (0, 3):
            if fam == avi.MEANFIELD:
                q = avi.MeanFieldGaussian(np.zeros(d, dt), np.ones(d, dt))
            else:
                q = avi.FullRankGaussian(np.zeros(d, dt), np.eye(d, dtype=dt))
            params, _ = avi.destructure(q)
            ctx = avi.MiviContext(dt, fam, d, M, ent, 1234)
            ctx.set_problem(avi.DiagNormalProblem(np.full(d, 5, dt), np.ones(d, dt)))
            p = ctx.to_device(params)
            row = [f"{np.dtype(dt).name:8s} {name:9s} ent={ent}", f"estimate {ctx.profile_kernel(0, p, 50) * 1e3:9.1f} us"]
            if fam == avi.FULLRANK:
                for w, nm in ((1, "eps"), (2, "sample"), (3, "vjp")):
                    row.append(f"{nm} {ctx.profile_kernel(w, p, 50) * 1e3:8.1f}")
            print("  ".join(row), flush=True)
            ctx.close()
