# developer: reproduce the 5-chain recapture crash (count 20 -> 24)
import numpy as np, torch, sys, time, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import advancedvi_jl_amd as avi
from tests.helpers import SEED
d, M = 1024, 256
q = avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
prob = avi.DiagNormalProblem(np.full(d, 5.0, np.float32), np.ones(d, np.float32))
params, _ = avi.destructure(q)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    ctx = avi.MiviContext(np.float32, 1, d, M, 0, SEED); ctx.set_problem(prob)
    p = ctx.to_device(params); v, g = ctx.empty(1), ctx.empty(ctx.params_len)
    for count in [int(x) for x in sys.argv[1:]]:
        print("count", count, flush=True)
        ctx.estimate_gradient_n(p, 0, count, v, g); st.synchronize()
        print("  ok", float(v), flush=True)
