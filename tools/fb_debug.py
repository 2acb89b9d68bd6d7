"""Developer: where does a batch-engine estimate differ from the single call?  Per part (mu / C), per 128-row block, per 128-column block."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
import advancedvi_jl_amd as avi
from tests.helpers import SEED, make_family, make_problem
d, M, ent, kind = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
rng = np.random.default_rng(5 + d + M)
q, _ = make_family(rng, d, avi.FULLRANK, np.float32)
prob, tgt = make_problem(rng, kind, d, np.float32)
params, _ = avi.destructure(q)
ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED); ctx.set_problem(prob)
ref = avi.MiviContext(np.float32, avi.FULLRANK, d, M, ent, SEED); ref.set_problem(prob)
p, pr = ctx.to_device(params), ref.to_device(params)
n = 3
vals, grads = ctx.estimate_gradient_each(p, 40, n)
ctx.synchronize()
vals, grads = vals.cpu().numpy(), grads.cpu().numpy()
for i in range(n):
    v1, g1 = ref.estimate_gradient(pr, 40 + i)
    g1 = g1.cpu().numpy().astype(np.float64); g = grads[i].astype(np.float64)
    print(f"est {i}: value {vals[i]} vs {float(v1.item())}  rel {abs(vals[i]-float(v1.item()))/abs(float(v1.item())):.2e}")
    print("  mu part rel l2:", np.linalg.norm(g[:d]-g1[:d])/np.linalg.norm(g1[:d]), " C part:", np.linalg.norm(g[d:]-g1[d:])/np.linalg.norm(g1[d:]))
    G, G1 = g[d:].reshape(d, d).T, g1[d:].reshape(d, d).T   # [row][col]
    nb = d // 128
    E = np.zeros((nb, nb))
    for a in range(nb):
        for b in range(a + 1):
            blk, blk1 = G[128*a:128*a+128, 128*b:128*b+128], G1[128*a:128*a+128, 128*b:128*b+128]
            E[a, b] = np.linalg.norm(blk - blk1) / max(np.linalg.norm(blk1), 1e-30)
    np.set_printoptions(precision=1, linewidth=200)
    print("  C tiles rel err:\n", E)
    em = np.abs(g[:d]-g1[:d]).reshape(-1, 32).max(axis=1)
    print("  mu abs err per 32-row block:", em)
