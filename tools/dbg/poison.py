import numpy as np, torch, sys
sys.path.insert(0, '/root/repo')
import advancedvi_jl_amd as avi
from tests.helpers import SEED, make_family, make_problem
def poison(val):
    xs = [torch.full((64 << 20,), val, dtype=torch.float32, device="cuda") for _ in range(8)]   # 2 GiB
    torch.cuda.synchronize(); del xs; torch.cuda.empty_cache()
def run(dtype, family, d, M, mloc, moff, ent, tag):
    rng = np.random.default_rng(5)
    q, _ = make_family(rng, d, family, dtype)
    prob, _ = make_problem(rng, "diag", d, dtype)
    params, _ = avi.destructure(q)
    full = avi.MiviContext(dtype, family, d, M, ent, SEED); full.set_problem(prob)
    c = avi.MiviContext(dtype, family, d, mloc, ent, SEED, m_offset=moff, m_total=M); c.set_problem(prob)
    c2 = avi.MiviContext(dtype, family, d, M - mloc, ent, SEED, m_offset=(moff + mloc) % M, m_total=M); c2.set_problem(prob)
    for idx in (17, 18):
        fp = full.estimate_partials(params, idx).double()
        pa = c.estimate_partials(params, idx).double(); pb = c2.estimate_partials(params, idx).double()
        tot = pa + pb
        L = fp.numel()
        def e(a, b): return float((a - b).norm() / (b.norm() + 1e-30))
        print(tag, idx, "mu", e(tot[:d], fp[:d]), "tri", e(tot[d:L-2], fp[d:L-2]), "scal", tot[L-2:].tolist(), fp[L-2:].tolist(), "nan", int(torch.isnan(tot).sum()))
    for x in (full, c, c2): x.close()
run(np.float32, 1, 256, 256, 128, 0, 0, "fresh")
poison(float("nan"))
run(np.float32, 1, 256, 256, 128, 0, 0, "nan-poison")
poison(1e3)
run(np.float32, 1, 256, 256, 128, 0, 0, "1e3-poison")
run(np.float64, 1, 256, 256, 128, 0, 0, "f64 after")
