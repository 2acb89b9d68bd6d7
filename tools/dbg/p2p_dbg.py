import numpy as np, torch, sys
sys.path.insert(0, '/root/repo')
import advancedvi_jl_amd as avi
from advancedvi_jl_amd.distributed import ShardPlan, p2p_geometry
from tests.helpers import SEED, make_family, make_problem
dtype, family, d, M, R, ent = np.float32, 1, 256, 256, 2, 0
rng = np.random.default_rng(5)
q, _ = make_family(rng, d, family, dtype)
prob, _ = make_problem(rng, "diag", d, dtype)
params, _ = avi.destructure(q)
full = avi.MiviContext(dtype, family, d, M, ent, SEED); full.set_problem(prob)
plan = ShardPlan(M, R)
ctxs = []
for r in range(R):
    c = avi.MiviContext(dtype, family, d, plan.count(r), ent, SEED, m_offset=plan.offset(r), m_total=M); c.set_problem(prob); ctxs.append(c)
hs = [c.p2p_export(r, R) for r, c in enumerate(ctxs)]
for c in ctxs: c.p2p_attach(hs); c.comm_set_route("p2p")
L = ctxs[0].partials_len
n, cn, G, vs = p2p_geometry(L, R)
print("L n cn G vs", L, n, cn, G, vs)
for idx in (17, 18, 19):
    v_ref, g_ref = full.estimate_gradient(params, idx)
    parts, outs = [], []
    for r, c in enumerate(ctxs):
        P = c.empty(n * R).zero_(); c.estimate_partials(params, idx, P[:L]); parts.append(P)
        outs.append((c.empty(1), c.empty(c.params_len).fill_(float('nan'))))
    torch.cuda.synchronize()
    tot = sum(P.double() for P in parts)
    v0, g0 = ctxs[0].finalize(params, tot.float()[:L].contiguous())
    for ph in (1, 2, 4):
        for r, c in enumerate(ctxs): c.p2p_exchange(c.to_device(params), parts[r], outs[r][0], outs[r][1], ph)
        torch.cuda.synchronize()
    print(idx, "ref", float(v_ref), "finalize", float(v0), "p2p", [float(o[0]) for o in outs],
          "gerr fin", float((g0 - g_ref).norm() / g_ref.norm()), "gerr p2p", float((outs[0][1] - g_ref).norm() / g_ref.norm()),
          "scalars", [P[L-2:L].tolist() for P in parts])
