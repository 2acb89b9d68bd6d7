import numpy as np, torch, sys, time
sys.path.insert(0, '/root/repo')
import advancedvi_jl_amd as avi
from tests.helpers import SEED, make_family, make_problem
d, M = 1024, 256
q = avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32))
prob = avi.DiagNormalProblem(np.full(d, 5.0, np.float32), np.ones(d, np.float32))
params, _ = avi.destructure(q)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    ctx = avi.MiviContext(np.float32, 1, d, M, 0, SEED)
    ctx.set_problem(prob)
    ctx.p2p_attach([ctx.p2p_export(0, 1)])
    ctx.p2p_set_spin_budget(1 << 15)
    p = ctx.to_device(params)
    v, g = ctx.empty(1), ctx.empty(ctx.params_len)
    v1, g1 = ctx.estimate_gradient_dist(p, 7); ctx.synchronize()
    print("single ok", float(v1))
    import ctypes as C
    def words():
        buf = (C.c_uint32 * 128)(); ctx.lib.mivi_p2p_debug_words.argtypes = [C.c_void_p, C.c_void_p]; ctx.lib.mivi_p2p_debug_words(ctx.h, buf)
        return dict(lane0=list(buf[0:2]), lane1=list(buf[16:18]), ready=buf[64], freed=list(buf[80:84]))
    print(words())
    for count in (1, 2):
        t0 = time.perf_counter()
        try:
            ctx.estimate_gradient_dist_n(p, 100, count, v, g); ctx.synchronize(); err = None
        except Exception as e: err = str(e)[:70]
        dt = time.perf_counter() - t0
        vr, gr = ctx.estimate_gradient(p, 100 + count - 1)
        print(words())
        print("count", count, "%.3f ms" % (dt * 1e3), "err", err, "v", float(v), float(vr), "gerr", float((g - gr).norm() / gr.norm()))
        t0 = time.perf_counter()
        try:
            ctx.estimate_gradient_dist_n(p, 200, count, v, g); ctx.synchronize(); err = None
        except Exception as e: err = str(e)[:70]
        print("   replay %.3f ms" % ((time.perf_counter() - t0) * 1e3), err)
