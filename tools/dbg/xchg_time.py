import numpy as np, torch, sys, time
sys.path.insert(0, '/root/repo')
import advancedvi_jl_amd as avi
from advancedvi_jl_amd.distributed import p2p_geometry
from tests.helpers import SEED
d, M = 1024, 256
q = avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
prob = avi.DiagNormalProblem(np.full(d, 5.0, np.float32), np.ones(d, np.float32))
params, _ = avi.destructure(q)
ctx = avi.MiviContext(np.float32, 1, d, M, 0, SEED)
ctx.set_problem(prob)
ctx.p2p_attach([ctx.p2p_export(0, 1)])
L = ctx.partials_len; n, cn, G, vs = p2p_geometry(L, 1)
print("L n cn G", L, n, cn, G)
p = ctx.to_device(params)
P = ctx.empty(n).zero_(); ctx.estimate_partials(p, 5, P[:L])
v, g = ctx.empty(1), ctx.empty(ctx.params_len)
torch.cuda.synchronize()
def timeit(phases_list, reps=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        for ph in phases_list: ctx.p2p_exchange(p, P, v, g, ph)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        for ph in phases_list: ctx.p2p_exchange(p, P, v, g, ph)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
# phases alone need the full sequence for epoch bookkeeping: time cumulative sequences
print("p1+p2+p3 fused   us", timeit([7]))
print("p1;p2;p3 separate us", timeit([1, 2, 4]))
print("p1 then (2|4)     us", timeit([1, 6]))
print("(1|2) then 4      us", timeit([3, 4]))
