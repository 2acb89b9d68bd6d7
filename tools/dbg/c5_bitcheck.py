# developer: mivi_estimate_gradient_n (launch-free funnel loop) against single calls, which elements differ and by how much
import numpy as np, sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import advancedvi_jl_amd as avi
from tests.helpers import SEED
for (d, M, ent) in ((300, 200, 4), (300, 200, 3), (2048, 64, 3), (64, 1024, 2)):
    q = avi.MeanFieldGaussian((0.1 * np.arange(d) / d).astype(np.float32), np.full(d, 0.8, np.float32))
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(np.float32, avi.MEANFIELD, d, M, ent, SEED)
    ctx.set_problem(avi.FunnelProblem(d, 1.5))
    p = ctx.to_device(params)
    v, g = ctx.empty(1), ctx.empty(ctx.params_len)
    bad = 0
    for idx in range(30, 30 + 40 * 9, 9):
        ctx.estimate_gradient_n(p, idx, 9, v, g); ctx.synchronize()
        v1, g1 = ctx.estimate_gradient(p, idx + 8)
        a, b = g.cpu().numpy(), g1.cpu().numpy()
        if not np.array_equal(a, b) or float(v.item()) != float(v1.item()):
            bad += 1
            w = np.nonzero(a != b)[0]
            if bad <= 2: print("  idx", idx, "value equal", float(v.item()) == float(v1.item()), "n diff", len(w), "where", w[:6], "rel", np.max(np.abs(a[w] - b[w]) / np.abs(b[w])) if len(w) else 0)
    print((d, M, ent), "mismatching batches:", bad, "of 40")
