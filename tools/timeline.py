"""Per-workgroup timeline of one launch of a pipeline stage (developer tool; wall_clock64 @ 100 MHz)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import advancedvi_jl_amd as avi
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "ns"
which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w = bench.WORKLOADS[wl]
q, prob = bench.make_problem(avi, w)
params_h, _ = avi.destructure(q)
ctx = avi.MiviContext(np.float32, w["family"], w["d"], w["n_mc"], w["entropy"], bench.SEED)
ctx.set_problem(prob)
params = ctx.to_device(params_h)
ctx.profile_kernel(which, params, 20)
nb = 4096
buf = torch.zeros(nb * 8, dtype=torch.int64, device="cuda")
ctx.lib.mivi_debug_timeline(ctx.h, buf.data_ptr())
ms = ctx.profile_kernel(which, params, 1)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nb, 8).astype(np.float64)
used = t[:, 0] > 0
t = t[used]
t0 = t[:, 0].min()
print(f"workload {wl} stage {which}: {used.sum()} blocks, launch ms {ms*1e3:.2f} us (incl. warm estimate)")
ns = 10.0  # ns per tick
for k in range(5):
    col = t[:, k]
    ok = col > 0
    if ok.sum() == 0:
        continue
    rel = (col[ok] - t0) * ns / 1e3
    print(f"  stamp {k}: n={ok.sum():5d} min {rel.min():7.2f} us  median {np.median(rel):7.2f}  p90 {np.percentile(rel,90):7.2f}  max {rel.max():7.2f}")
for k in range(1, 5):
    ok = (t[:, k] > 0) & (t[:, k - 1] > 0)
    if ok.sum():
        dd = (t[ok, k] - t[ok, k - 1]) * ns / 1e3
        print(f"  phase {k-1}->{k}: median {np.median(dd):6.2f} us  p90 {np.percentile(dd,90):6.2f}  max {dd.max():6.2f}")
