"""Odd and large shapes through the C ABI vs the oracle (tools-only; the regular suite covers the small ones)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import advancedvi_jl_amd as avi
from oracle import oracle as O
from tests.helpers import make_family, make_problem, rel_err
SEED = 77
cases = [(avi.FULLRANK, 2048, 64, "diag", 0, np.float32), (avi.FULLRANK, 2048, 64, "diag", 3, np.float32),
         (avi.FULLRANK, 4096, 32, "diag", 2, np.float32), (avi.FULLRANK, 4096, 32, "diag", 3, np.float32),
         (avi.FULLRANK, 1000, 100, "dense", 0, np.float32), (avi.FULLRANK, 1537, 33, "diag", 4, np.float32),
         (avi.FULLRANK, 1537, 33, "diag", 4, np.float64), (avi.FULLRANK, 2048, 64, "diag", 3, np.float64),
         (avi.MEANFIELD, 1 << 20, 8, "diag", 0, np.float32), (avi.MEANFIELD, 100003, 7, "diag", 3, np.float32),
         (avi.FULLRANK, 1, 1, "diag", 0, np.float32), (avi.FULLRANK, 3, 1000, "diag", 2, np.float32),
         (avi.MEANFIELD, 1, 1, "diag", 2, np.float64), (avi.FULLRANK, 33, 4096, "dense", 3, np.float32)]
for fam, d, M, kind, ent, dt in cases:
    rng = np.random.default_rng(d + M)
    q, q_o = make_family(rng, d, fam, dt)
    prob, tgt = make_problem(rng, kind, d, dt)
    params, _ = avi.destructure(q)
    t0 = time.time()
    try:
        ctx = avi.MiviContext(dt, fam, d, M, ent, SEED)
        ctx.set_problem(prob)
        Z, eps = ctx.sample(params, 3)
        v, g = ctx.estimate_gradient(params, 3)
        ctx.synchronize()
        v = float(v.item()); g = g.cpu().numpy().astype(np.float64)
        ref = O.estimate_gradient(O.destructure(q_o), d, fam, tgt, eps.cpu().numpy().astype(np.float64), ent)
        ev = abs(v - ref["value"]) / max(abs(ref["value"]), 1e-300)
        eg = np.linalg.norm(g - ref["grad"]) / max(np.linalg.norm(ref["grad"]), 1.0)
        print(f"fam={fam} d={d} M={M} {kind} ent={ent} {np.dtype(dt).name}: value rel {ev:.2e} grad rel {eg:.2e}  ({time.time()-t0:.1f}s)", flush=True)
        ctx.close()
    except Exception as e:   # noqa: BLE001
        print(f"fam={fam} d={d} M={M} {kind} ent={ent} {np.dtype(dt).name}: EXC {type(e).__name__}: {e}", flush=True)
