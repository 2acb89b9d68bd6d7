import numpy as np, torch, sys
sys.path.insert(0, '/root/repo')
import advancedvi_jl_amd as avi
from advancedvi_jl_amd.distributed import ShardPlan, p2p_geometry
from tests.helpers import SEED, make_family, make_problem
def run(dtype, family, d, M, R, ent):
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
    print("case", dtype.__name__, family, d, M, R, "L n cn G vs", L, n, cn, G, vs)
    for idx in (17, 18):
        v_ref, g_ref = full.estimate_gradient(params, idx)
        parts, outs = [], []
        for r, c in enumerate(ctxs):
            P = c.empty(n * R).zero_(); c.estimate_partials(params, idx, P[:L]); parts.append(P)
            outs.append((c.empty(1), c.empty(c.params_len).fill_(float('nan'))))
        torch.cuda.synchronize()
        for ph in (1, 2, 4):
            for r, c in enumerate(ctxs): c.p2p_exchange(c.to_device(params), parts[r], outs[r][0], outs[r][1], ph)
            torch.cuda.synchronize()
        errs = []
        for c in ctxs:
            try: c.synchronize()
            except Exception as e: errs.append(str(e)[:60])
        g = outs[0][1].cpu().numpy().astype(np.float64); gr = g_ref.cpu().numpy().astype(np.float64)
        bad = np.flatnonzero(~(np.abs(g - gr) <= 1e-5 * (1 + np.abs(gr))))
        print(" idx", idx, "v", float(outs[0][0]), float(v_ref), "nbad", bad.size, "of", g.size, "first bad", bad[:12], g[bad[:6]], gr[bad[:6]], errs[:1])
    for c in ctxs + [full]: c.close()
run(np.float32, 0, 64, 48, 4, 0)
run(np.float32, 1, 40, 30, 1, 0)
run(np.float32, 1, 40, 30, 3, 0)
run(np.float64, 0, 64, 48, 2, 0)
