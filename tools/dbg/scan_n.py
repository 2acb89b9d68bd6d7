# developer: which batch lengths break the bitwise batch == single-call property (entropy estimator argv[1], default 2)
import numpy as np, sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import advancedvi_jl_amd as avi
from tests.helpers import SEED, make_family, make_problem
ent = int(sys.argv[1]) if len(sys.argv) > 1 else 2
d, M = 128, 128
rng = np.random.default_rng(21)
q, _ = make_family(rng, d, avi.FULLRANK, np.float32)
prob, _ = make_problem(rng, "diag", d, np.float32)
params, _ = avi.destructure(q)
ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED); ctx.set_problem(prob)
ref = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED); ref.set_problem(prob)
p, pr = ctx.to_device(params), ref.to_device(params)
v, g = ctx.empty(1), ctx.empty(ctx.params_len)
idx = 3
bad = []
for rep in range(2):
    for n in list(range(1, 41)) + [49, 50, 51, 52, 64]:
        ctx.estimate_gradient_n(p, idx, n, v, g); ctx.synchronize()
        v1, g1 = ref.estimate_gradient(pr, idx + n - 1)
        if float(v.item()) != float(v1.item()) or not np.array_equal(g.cpu().numpy(), g1.cpu().numpy()):
            bad.append((n, idx, float(v.item()) - float(v1.item()), bool(np.array_equal(g.cpu().numpy(), g1.cpu().numpy()))))
        idx += n + 2
print("bad:", bad)
