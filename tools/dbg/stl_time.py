# developer: time the STL term alone (profile stage 8) on the ns_stl workload; run under rocprofv3 for the kernel split
import os, sys, numpy as np
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import advancedvi_jl_amd as avi, bench
w = bench.WORKLOADS["ns_stl"]
q, prob = bench.make_problem(avi, w)
p_h, _ = avi.destructure(q)
ctx = avi.MiviContext(np.float32, w["family"], w["d"], w["n_mc"], w["entropy"], bench.SEED); ctx.set_problem(prob)
p = ctx.to_device(p_h)
ctx.profile_kernel(8, p, 50)
print("stl_term us:", [round(ctx.profile_kernel(8, p, 300) * 1e3, 2) for _ in range(4)])
