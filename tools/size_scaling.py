"""Stage times and f32-MFMA rate of the full-rank estimate as (d, M) grow: where the north-star shape sits on the curve.
Per stage: algorithmic flops d^2 M (lower triangle only), hipEvent-timed launches (mivi_profile_kernel)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import advancedvi_jl_amd as avi

for d, M in [(512, 128), (1024, 256), (2048, 256), (2048, 1024), (4096, 1024), (4096, 4096), (8192, 2048)]:
    q = avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
    prob = avi.DiagNormalProblem(np.full(d, 5, np.float32), np.ones(d, np.float32))
    p_h, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, 1)
    ctx.set_problem(prob)
    p = ctx.to_device(p_h)
    t = {}
    for name, which in (("sample", 2), ("vjp", 3), ("estimate", 0)):
        ctx.profile_kernel(which, p, 5)
        t[name] = min(ctx.profile_kernel(which, p, 30) for _ in range(2)) * 1e3   # us
    fl = float(d) * d * M
    print(f"d={d:5d} M={M:5d}: sample {t['sample']:9.1f} us ({fl / t['sample'] / 1e6:6.1f} TF)  vjp {t['vjp']:9.1f} us "
          f"({fl / t['vjp'] / 1e6:6.1f} TF)  whole estimate {t['estimate']:9.1f} us ({2 * fl / t['estimate'] / 1e6:6.1f} TF = "
          f"{2 * fl / t['estimate'] / 1e6 / 157.3 * 100:4.1f} % of the f32-MFMA peak)", flush=True)
    ctx.close()
