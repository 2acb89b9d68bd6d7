"""Per-workgroup timeline of k_fr_prod32 WITH its riders (developer tool; wall_clock64 @ 100 MHz): one STL estimate at the north-star
shape, stamps 0 entry, 6 diagonal-block inverse done (its hosts), 1 first operands, 2 main loop, 3 accumulators in LDS, 4 tile done,
5 rider done.   python tools/rider_timeline.py [d] [entropy: 0 closed form | 3 STL]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import advancedvi_jl_amd as avi

d = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ent = int(sys.argv[2]) if len(sys.argv) > 2 else 3
M = 256
rng = np.random.default_rng(1)
Cm = np.tril(rng.normal(size=(d, d)) * (0.3 / np.sqrt(d))).astype(np.float32)
Cm[np.diag_indices(d)] = 1.0
q = avi.FullRankGaussian(rng.normal(size=d).astype(np.float32), Cm)
prob = avi.DiagNormalProblem(np.full(d, 5, np.float32), np.ones(d, np.float32))
p_h, _ = avi.destructure(q)
ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, 1)
ctx.set_problem(prob)
p = ctx.to_device(p_h)
buf = torch.zeros(4 * 4096 * 8, dtype=torch.int64, device="cuda")
NS = 10.0
for i in range(3):
    ctx.estimate_gradient(p, i)
ctx.lib.mivi_debug_timeline(ctx.h, buf.data_ptr())
best = None
for rep in range(8):
    buf.zero_()
    torch.cuda.synchronize()
    ctx.estimate_gradient(p, 4 + rep)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(4, 4096, 8).astype(np.float64)[0]
    t = t[t[:, 0] > 0]
    span = (t[:, :7].max() - t[:, 0].min()) * NS / 1e3
    if best is None or span < best[0]:
        best = (span, t.copy())
ctx.lib.mivi_debug_timeline(ctx.h, None)
span, t = best
t0 = t[:, 0].min()
print(f"k_fr_prod32 with riders, d={d} entropy={ent}: {len(t)} workgroups, first entry -> last stamp {span:.2f} us")
for k, nm in ((0, "entry"), (6, "inverse done"), (1, "first operands"), (2, "main loop"), (3, "acc in LDS"), (4, "tile done"), (5, "rider done")):
    col = t[:, k]
    ok = col > 0
    if ok.any():
        rel = (col[ok] - t0) * NS / 1e3
        print(f"   {nm:15s} n={int(ok.sum()):4d}  min {rel.min():6.2f}  median {np.median(rel):6.2f}  p90 {np.percentile(rel, 90):6.2f}  max {rel.max():6.2f} us")
end = np.maximum(t[:, 4], t[:, 5])
late = np.argsort(-end)[:20]
print("   last workgroups: block, entry, inverse, first operands, tile done, rider done [us]")
for b in late:
    f = lambda k: (t[b, k] - t0) * NS / 1e3 if t[b, k] > 0 else float("nan")
    print(f"     block {b:4d}  {f(0):6.2f} {f(6):6.2f} {f(1):6.2f} {f(4):6.2f} {f(5):6.2f}")
ctx.close()
