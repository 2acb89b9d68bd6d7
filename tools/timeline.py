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
nb = 3 * 4096
buf = torch.zeros(nb * 8, dtype=torch.int64, device="cuda")
ctx.lib.mivi_debug_timeline(ctx.h, buf.data_ptr())
ctx.profile_kernel(which, params, 1)
torch.cuda.synchronize()
buf.zero_()
torch.cuda.synchronize()
ms = ctx.profile_kernel(which, params, 1)   # warm estimate (all kernels stamp) + the stage alone (latest stamps win)
torch.cuda.synchronize()
kind = {1: 0, 2: 0, 3: 1, 4: 2}[which]
t = buf.cpu().numpy().reshape(3, 4096, 8)[kind].astype(np.float64)
t = t[t[:, 0] > 0]
print(f"workload {wl} stage {which}: {len(t)} blocks stamped")
t0 = t[:, 0].min()
ns = 10.0  # ns per tick
for k in range(4):
    col = t[:, k]
    ok = col > 0
    if ok.sum() == 0:
        continue
    rel = (col[ok] - t0) * ns / 1e3
    print(f"  stamp {k}: n={ok.sum():5d} min {rel.min():7.2f} us  median {np.median(rel):7.2f}  p90 {np.percentile(rel,90):7.2f}  max {rel.max():7.2f}")
for k in range(1, 4):
    ok = (t[:, k] > 0) & (t[:, k - 1] > 0)
    if ok.sum():
        dd = (t[ok, k] - t[ok, k - 1]) * ns / 1e3
        print(f"  phase {k-1}->{k}: median {np.median(dd):6.2f} us  p90 {np.percentile(dd,90):6.2f}  max {dd.max():6.2f}")

ok = (t[:, 6] > 0) & (t[:, 7] > 0) & (t[:, 2] > 0)
if ok.sum():
    cyc = t[ok, 7] - t[ok, 6]
    wall_us = (t[ok, 2] - t[ok, 0]) * ns / 1e3
    print(f"  shader clock during main loop: median {np.median(cyc / wall_us) / 1e3:.2f} GHz  (cycles {np.median(cyc):.0f} over {np.median(wall_us):.2f} us)")

ok = (t[:, 5] > 0) | True
end = (t[:, 3] - t0) * ns / 1e3
tile = t[:, 5].astype(np.int64)
ib, jb = tile >> 16, tile & 0xFFFF
order = np.argsort(-end)[:12]
print("  slowest blocks (end us, ib, jb, start us, main, reduce, epi):")
for o in order:
    print(f"    {end[o]:6.2f}  ib={ib[o]:2d} jb={jb[o]:2d}  start {(t[o,0]-t0)*ns/1e3:5.2f}  main {(t[o,1]-t[o,0])*ns/1e3:5.2f}  red {(t[o,2]-t[o,1])*ns/1e3:5.2f}  epi {(t[o,3]-t[o,2])*ns/1e3:5.2f}")
if which == 3:
    diag = ib == jb
    print(f"  diag tiles: end median {np.median(end[diag]):.2f} max {end[diag].max():.2f}; off-diag: median {np.median(end[~diag]):.2f} max {end[~diag].max():.2f}")
