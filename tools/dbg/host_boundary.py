# developer: mivi_estimate_gradient_host per call, before and after the interleaved contexts exist
import os, sys, time, ctypes as C, numpy as np, torch
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
    ph = np.ascontiguousarray(params, dtype=np.float32)
    v_h, g_h = np.zeros(1, np.float32), np.zeros(ctx.params_len, np.float32)
    def host_call(i):
        s = ctx.lib.mivi_estimate_gradient_host(ctx.h, ph.ctypes.data_as(C.c_void_p), i, v_h.ctypes.data_as(C.c_void_p), g_h.ctypes.data_as(C.c_void_p))
        assert s == 0
    def bench(tag):
        for i in range(5): host_call(i)
        t0 = time.perf_counter()
        for i in range(30): host_call(100 + i)
        print(tag, "%.0f us per call" % ((time.perf_counter() - t0) / 30 * 1e6))
    bench("fresh context      ")
    p = ctx.to_device(params); v, g = ctx.empty(1), ctx.empty(ctx.params_len)
    ctx.estimate_gradient_n(p, 0, 20, v, g); st.synchronize()
    bench("after a 20-batch   ")
    ctx.estimate_gradient_n(p, 0, 100, v, g); st.synchronize()
    bench("after a 100-batch  ")
