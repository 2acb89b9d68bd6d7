# developer: cost of the synchronisation bracket around an isolated 20-estimate call (stream sync / device sync / both)
import numpy as np, torch, sys, time, os, ctypes
if os.environ.get('SPIN'):
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
    print('hipSetDeviceFlags ->', hip.hipSetDeviceFlags(ctypes.c_uint(int(os.environ['SPIN']))))
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
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2: ctx.estimate_gradient_n(p, 5, 20, v, g)
    st.synchronize()
    modes = {"stream": lambda: st.synchronize(), "device": lambda: torch.cuda.synchronize(), "stream+device": lambda: (st.synchronize(), torch.cuda.synchronize()),
             "ctx.synchronize": lambda: ctx.synchronize(), "ctx+device": lambda: (ctx.synchronize(), torch.cuda.synchronize())}
    for name, sync in modes.items():
        ts = []
        for r in range(40):
            sync()
            t0 = time.perf_counter(); ctx.estimate_gradient_n(p, 25 + 20 * r, 20, v, g); sync(); ts.append((time.perf_counter() - t0) / 20 * 1e6)
        ts.sort()
        print("%-16s us/est median %.2f min %.2f max %.2f" % (name, ts[20], ts[0], ts[-1]))
