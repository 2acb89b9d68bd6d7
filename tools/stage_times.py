import sys, numpy as np
sys.path.insert(0, ".")
import advancedvi_jl_amd as avi, bench
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "ns"]
q, prob = bench.make_problem(avi, w)
p_h, _ = avi.destructure(q)
ctx = avi.MiviContext(np.float32, w["family"], w["d"], w["n_mc"], w["entropy"], bench.SEED); ctx.set_problem(prob)
p = ctx.to_device(p_h)
out = {}
for name, which in (("eps", 1), ("sample", 2), ("vjp", 3)):
    ctx.profile_kernel(which, p, 50)
    out[name] = round(min(ctx.profile_kernel(which, p, 300) for _ in range(3)) * 1e3, 2)
print(out)
