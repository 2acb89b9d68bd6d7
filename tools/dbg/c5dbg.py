import numpy as np, sys
sys.path.insert(0, '/root/repo')
import advancedvi_jl_amd as avi
from tests.helpers import SEED
for dtype, d, M, ent in ((np.float64, 300, 200, 4), (np.float64, 300, 200, 3), (np.float64, 300, 64, 4), (np.float64, 304, 200, 4), (np.float32, 300, 200, 4)):
    q = avi.MeanFieldGaussian((0.1 * np.arange(d) / d).astype(dtype), np.full(d, 0.8, dtype))
    params, _ = avi.destructure(q)
    ctx = avi.MiviContext(dtype, avi.MEANFIELD, d, M, ent, SEED); ctx.set_problem(avi.FunnelProblem(d, 1.5))
    p = ctx.to_device(params); v, g = ctx.empty(1), ctx.empty(ctx.params_len).fill_(float("nan"))
    ctx.estimate_gradient_n(p, 30, 9, v, g); ctx.synchronize()
    v1, g1 = ctx.estimate_gradient(p, 38)
    a, b = g.cpu().numpy(), g1.cpu().numpy()
    bad = np.flatnonzero(a != b)
    print(dtype.__name__, d, M, ent, "nbad", bad.size, bad[:10], (a[bad[:4]] - b[bad[:4]]) / np.abs(b[bad[:4]]))
