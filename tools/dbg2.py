import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import advancedvi_jl_amd as avi
d, M = int(sys.argv[1]), int(sys.argv[2])
what = sys.argv[3]
q = avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
params, _ = avi.destructure(q)
prob = avi.DiagNormalProblem(np.full(d, 5, np.float32), np.ones(d, np.float32))
ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, 1234)
ctx.set_problem(prob)
p = ctx.to_device(params)
value, grad = ctx.empty(1), ctx.empty(ctx.params_len)
if what == "single":
    for i in range(5):
        ctx.estimate_gradient(p, i, value, grad)
    torch.cuda.synchronize(); print("single ok", value.item())
elif what == "graph":
    ctx.estimate_gradient_n(p, 0, 10, value, grad)
    torch.cuda.synchronize(); print("graph ok", value.item())
elif what == "prof":
    for w in range(0, 4):
        print(w, ctx.profile_kernel(w, p, 10))
elif what == "part":
    part = ctx.estimate_partials(p, 0)
    torch.cuda.synchronize(); print("part ok")
    v, g = ctx.finalize(p, part)
    torch.cuda.synchronize(); print("fin ok", v.item())
