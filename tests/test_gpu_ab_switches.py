"""Every environment switch libmivi still reads selects an in-library reference route (switch table).  Each is
read once per process, so every case runs in a child process: the route under the switch must agree with the oracle (estimates) or
with the step-by-step entries (loops) exactly like the default route does."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ESTIMATE = """
import numpy as np, advancedvi_jl_amd as avi
from oracle import oracle as O
from tests.helpers import SEED, make_family, make_problem
fam, d, M, kind, ent, dt = {fam}, {d}, {M}, {kind!r}, {ent}, np.{dt}
rng = np.random.default_rng(d + M)
q, q_o = make_family(rng, d, fam, dt)
prob, tgt = make_problem(rng, kind, d, dt)
params, _ = avi.destructure(q)
ctx = avi.MiviContext(dt, fam, d, M, ent, SEED)
ctx.set_problem(prob)
{pre}
_, eps = ctx.sample(params, 3)
v, g = ctx.estimate_gradient(params, 3)
ref = O.estimate_gradient(O.destructure(q_o), d, fam, tgt, eps.cpu().numpy().astype(np.float64), ent)
vt, gt = (1e-5, 2e-5) if dt == np.float32 else (1e-12, 1e-11)
assert abs(float(v.item()) - ref['value']) <= vt * max(abs(ref['value']), 1.0), (float(v.item()), ref['value'])
assert np.linalg.norm(g.cpu().numpy() - ref['grad']) <= gt * max(np.linalg.norm(ref['grad']), 1.0)
print('ok')
"""

LOOP = """
import numpy as np, advancedvi_jl_amd as avi
from tests.helpers import SEED
fam, d, M, T = {fam}, {d}, {M}, 7
tm, ts = np.full(d, 2.0, np.float32), np.full(d, 0.7, np.float32)
q0 = (avi.MeanFieldGaussian(np.zeros(d, np.float32), np.ones(d, np.float32)) if fam == avi.MEANFIELD
      else avi.FullRankGaussian(np.zeros(d, np.float32), np.eye(d, dtype=np.float32)))
p0, _ = avi.destructure(q0)
ctx = avi.MiviContext(np.float32, fam, d, M, 0, SEED)
ctx.set_problem(avi.DiagNormalProblem(tm, ts))
pa = ctx.to_device(p0).clone(); st = ctx.empty(2 * pa.numel()).zero_()
for t in range(T):
    v, g = ctx.estimate_gradient(pa, 50 + t)
    ctx.adam_update(pa, g, st, t + 1, 1e-2)
    ctx.clip_scale(pa, 1e-5)
pb = ctx.to_device(p0).clone(); st2 = ctx.empty(2 * pb.numel()).zero_()
ctx.optimize_steps(pb, st2, 50, 0, T, 1, 1e-2, 1e-5, None)
ctx.synchronize()
assert np.array_equal(pa.cpu().numpy(), pb.cpu().numpy())
v, g = ctx.empty(1), ctx.empty(ctx.params_len)
ctx.estimate_gradient_n(pb, 90, 4, v, g)
v1, g1 = ctx.estimate_gradient(pb, 93)
from tests.helpers import assert_batch_matches_single
assert_batch_matches_single(v.item(), v1.item(), g.cpu().numpy(), g1.cpu().numpy(), ctx.batch_takes_engine(pb))   # (the engine: to rounding; every other route: bitwise)
print('ok')
"""

STEIN = """
import numpy as np, advancedvi_jl_amd as avi
from oracle import oracle as O
from tests.helpers import SEED, make_family, make_problem
d, n = 256, 256
rng = np.random.default_rng(5)
q, q_o = make_family(rng, d, avi.FULLRANK, np.float32)
prob, tgt = make_problem(rng, 'dense', d, np.float32)
params, _ = avi.destructure(q)
ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, n, 0, SEED)
ctx.set_problem(prob)
_, eps = ctx.sample(params, 4)
lp, g, H = ctx.gauss_expected_grad_hess(params, 4)
ref = O.gaussian_expectation_gradient_and_hessian(q_o, tgt, eps.cpu().numpy().astype(np.float64))
assert np.linalg.norm(H.cpu().numpy() - ref[2]) <= 5e-5 * np.linalg.norm(ref[2])
assert np.linalg.norm(g.cpu().numpy() - ref[1]) <= 2e-5 * max(1.0, np.linalg.norm(ref[1]))
print('ok')
"""

CHAINS = """
import numpy as np, advancedvi_jl_amd as avi
from tests.helpers import SEED, assert_batch_matches_single, make_family, make_problem
d, M, n = 256, 128, 25
rng = np.random.default_rng(9)
q, _ = make_family(rng, d, avi.FULLRANK, np.float32)
prob, _ = make_problem(rng, {kind!r}, d, np.float32)
params, _ = avi.destructure(q)
ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
ctx.set_problem(prob)
p = ctx.to_device(params)
v, g = ctx.empty(1), ctx.empty(ctx.params_len)
ULPS = 0
for rep in range(2):
    ctx.estimate_gradient_n(p, 10 + 40 * rep, n, v, g)
    ctx.synchronize()
    v1, g1 = ctx.estimate_gradient(p, 10 + 40 * rep + n - 1)
    assert_batch_matches_single(v.item(), v1.item(), g.cpu().numpy(), g1.cpu().numpy(), ctx.batch_takes_engine(p), ulps=ULPS)   # (the engine: to rounding; every other route: bitwise)
print('ok')
"""

FUNNEL = """
import numpy as np, advancedvi_jl_amd as avi
from oracle import oracle as O
from tests.helpers import SEED
d, M, n = 96, 64, 9
q = avi.MeanFieldGaussian((0.1 * np.arange(d) / d).astype(np.float32), np.full(d, 0.8, np.float32))
params, _ = avi.destructure(q)
ctx = avi.MiviContext(np.float32, avi.MEANFIELD, d, M, 3, SEED)
ctx.set_problem(avi.FunnelProblem(d, 1.5))
p = ctx.to_device(params)
v, g = ctx.empty(1), ctx.empty(ctx.params_len)
ctx.estimate_gradient_n(p, 30, n, v, g)
ctx.synchronize()
_, eps = ctx.sample(params, 30 + n - 1)
ref = O.estimate_gradient(params.astype(np.float64), d, avi.MEANFIELD, O.FunnelStackedTarget(d, 1.5), eps.cpu().numpy().astype(np.float64), 3)
assert abs(float(v.item()) - ref["value"]) <= 2e-5 * abs(ref["value"])
assert np.linalg.norm(g.cpu().numpy() - ref["grad"]) <= 2e-5 * max(1.0, np.linalg.norm(ref["grad"]))
print('ok')
"""

CHAINS_NS = CHAINS.replace("d, M, n = 256, 128, 25", "d, M, n = 1024, 256, 23")   # the shape whose lane-batched launches are k_fr_prod32q / k_fr_vjp32s

# the sticking-the-landing estimator at a shape whose batches take the engine by default (values: the Monte Carlo entropy's one-ulp note of tests/test_gpu_batches.py)
CHAINS_STL = CHAINS_NS.replace("d, M, 0, SEED", "d, M, 3, SEED").replace("ULPS = 0", "ULPS = 1")
assert CHAINS_STL != CHAINS_NS and "M, 3, SEED" in CHAINS_STL

# the sharded estimate at world 1 on the peer-to-peer route (single estimates and a pipelined batch): with MIVI_P2P_DIRECT=1 the partial vector
# goes straight into the staging areas instead of through the ring slots and the push pass (the default)
P2P = """
import numpy as np, advancedvi_jl_amd as avi
from tests.helpers import SEED, make_family, make_problem
d, M = 256, 128
rng = np.random.default_rng(6)
q, _ = make_family(rng, d, avi.FULLRANK, np.float32)
prob, _ = make_problem(rng, 'diag', d, np.float32)
params, _ = avi.destructure(q)
ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
ctx.set_problem(prob)
ref = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
ref.set_problem(prob)
ctx.p2p_attach([ctx.p2p_export(0, 1)])
assert ctx.comm_route() == 'p2p'
p = ctx.to_device(params)
for idx in (5, 6, 7):
    v1, g1 = ctx.estimate_gradient_dist(p, idx)
    ctx.synchronize()
    v0, g0 = ref.estimate_gradient(params, idx)
    assert abs(float(v1.item()) - float(v0.item())) <= 2e-6 * abs(float(v0.item()))
    assert np.linalg.norm(g1.cpu().numpy() - g0.cpu().numpy()) <= 5e-6 * max(1.0, np.linalg.norm(g0.cpu().numpy()))
v, g = ctx.empty(1), ctx.empty(ctx.params_len)
for rep in range(2):
    ctx.estimate_gradient_dist_n(p, 40 + 20 * rep, 11, v, g)
    ctx.synchronize()
    v0, g0 = ref.estimate_gradient(params, 40 + 20 * rep + 10)
    assert abs(float(v.item()) - float(v0.item())) <= 2e-6 * abs(float(v0.item()))
    assert np.linalg.norm(g.cpu().numpy() - g0.cpu().numpy()) <= 5e-6 * max(1.0, np.linalg.norm(g0.cpu().numpy()))
print('ok')
"""

LOGREG_BIG = """
import numpy as np, advancedvi_jl_amd as avi
from oracle import oracle as O
from tests.helpers import SEED, make_family
d, M, n = 128, 128, 2048   # n (d - 1) >= 1e5: the planes of X exist (kernels_targets.hip, logreg_prepare_f32)
rng = np.random.default_rng(77)
q, q_o = make_family(rng, d, avi.FULLRANK, np.float32)
X = (rng.normal(size=(n, d - 1)) / np.sqrt(d)).astype(np.float32)
y = (rng.uniform(size=n) < 0.5).astype(np.uint8)
prob, tgt = avi.LogRegProblem(X, y, "logsigma_normal", 1.7), O.LogRegTarget(X, y, "logsigma_normal", 1.7)
params, _ = avi.destructure(q)
ctx = avi.MiviContext(np.float32, avi.FULLRANK, d, M, 0, SEED)
ctx.set_problem(prob)
ctx.set_logreg_route(1)
_, eps = ctx.sample(params, 3)
v, g = ctx.estimate_gradient(params, 3)
ref = O.estimate_gradient(O.destructure(q_o), d, avi.FULLRANK, tgt, eps.cpu().numpy().astype(np.float64), 0)
assert abs(float(v.item()) - ref['value']) <= 1e-5 * max(abs(ref['value']), 1.0), (float(v.item()), ref['value'])
assert np.linalg.norm(g.cpu().numpy() - ref['grad']) <= 2e-5 * max(np.linalg.norm(ref['grad']), 1.0)
print('ok')
"""

F, MF = 1, 0
CASES = [
    # switch, script, parameters
    ("MIVI_FR_GEN1=1", ESTIMATE, dict(fam=F, d=512, M=256, kind="diag", ent=0, dt="float32")),            # first-generation tile kernels
    ("MIVI_FR_F32MFMA=1", ESTIMATE, dict(fam=F, d=512, M=256, kind="dense", ent=0, dt="float32")),         # f32-MFMA chains instead of bf16x3
    ("MIVI_FR_F32MFMA=1", ESTIMATE, dict(fam=F, d=2048, M=512, kind="diag", ent=0, dt="float32")),         # ... on the 64 x 64 kernels
    ("MIVI_VJP_TILE=64", ESTIMATE, dict(fam=F, d=1024, M=256, kind="diag", ent=0, dt="float32")),          # 64 x 64 VJP tiles at the north star
    ("MIVI_VJP_TILE=32", ESTIMATE, dict(fam=F, d=2048, M=1024, kind="diag", ent=0, dt="float32")),         # 32 x 32 VJP tiles at a large shape
    ("MIVI_STL_GEN1=1", ESTIMATE, dict(fam=F, d=512, M=128, kind="diag", ent=3, dt="float32")),            # look-ahead solve instead of k_stl_solve64
    ("MIVI_STL_VALU=1", ESTIMATE, dict(fam=F, d=256, M=64, kind="diag", ent=3, dt="float32")),             # VALU back substitution
    ("MIVI_STL_VALU=1", ESTIMATE, dict(fam=F, d=128, M=32, kind="diag", ent=4, dt="float64")),
    ("MIVI_F64_VALU=1", ESTIMATE, dict(fam=F, d=160, M=48, kind="dense", ent=3, dt="float64")),            # f64 tiles on the vector ALU
    ("MIVI_LR_F32_LOGITS=1", ESTIMATE, dict(fam=F, d=64, M=128, kind="logreg0", ent=0, dt="float32", pre="ctx.set_logreg_route(1)")),
    ("MIVI_LR_F32_XTR=1", ESTIMATE, dict(fam=F, d=64, M=128, kind="logreg0", ent=0, dt="float32", pre="ctx.set_logreg_route(1)")),
    ("MIVI_LR_NO_PLANES=1", LOGREG_BIG, dict()),                                                          # logits with X split in the tile (k_lr_logits_f16x2) instead of the prebuilt planes
    ("MIVI_LR_NO_XPLANES=1", LOGREG_BIG, dict()),                                                         # X^T R with the splits made in the tile (k_lr_xtr_f16x2); logits still on planes
    ("MIVI_DUMMY_DEFAULT=1", LOGREG_BIG, dict()),                                                         # (no switch: both contractions on planes)
    ("MIVI_LOGREG_GENERIC=1", ESTIMATE, dict(fam=MF, d=64, M=128, kind="logreg0", ent=0, dt="float32")),
    ("MIVI_LOGREG_MFMA=1", ESTIMATE, dict(fam=MF, d=64, M=128, kind="logreg0", ent=0, dt="float32")),
    ("MIVI_NO_FUSED_LOOP=1", LOOP, dict(fam=MF, d=64, M=32)),                                             # graph loop instead of the launch-free kernel
    ("MIVI_NO_FUSED_LOOP=1", LOOP, dict(fam=F, d=64, M=32)),                                              # ... instead of the row-separable full-rank loop (few samples per step)
    ("MIVI_NO_FUSED_LOOP=1", LOOP, dict(fam=F, d=256, M=8)),
    ("MIVI_DUMMY_DEFAULT=1", FUNNEL, dict()),                                                             # (no switch: the table)
    ("MIVI_NO_FUSED_UPDATE=1", LOOP, dict(fam=F, d=128, M=128)),                                          # separate update kernel in the graph loop
    ("MIVI_GRAPH_MIN=1", LOOP, dict(fam=F, d=128, M=128)),                                                # graph replay even for the shortest batches
    ("MIVI_GRAPH_MIN=100", LOOP, dict(fam=F, d=128, M=128)),                                              # eager chain for every batch
    ("MIVI_STEIN_GEN1=1", STEIN, dict()),
    ("MIVI_BATCH_GEN3=0,MIVI_CHAINS=1", CHAINS, dict(kind="diag")),                                                         # one chain instead of interleaved ones
    ("MIVI_BATCH_GEN3=0,MIVI_CHAINS=4", CHAINS, dict(kind="dense")),                                                        # four contexts
    ("MIVI_BATCH_GEN3=0,MIVI_LANE_BATCH=0", CHAINS, dict(kind="diag")),                                                     # every context on a graph branch of its own (no lane-batched launches)
    ("MIVI_BATCH_GEN3=0,MIVI_LANE_BATCH=2", CHAINS, dict(kind="dense")),                                                    # two contexts per lane-batched launch
    ("MIVI_BATCH_GEN3=0,MIVI_CHAINS=8", CHAINS, dict(kind="diag")),                                                         # eight contexts: two branches of four lanes
    ("MIVI_BATCH_GEN3=0,MIVI_PROD_QUAD=0", CHAINS_NS, dict(kind="diag")),                                                   # four lanes' products on k_fr_prod32's tiles (k_fr_prod32m)
    ("MIVI_BATCH_GEN3=0,MIVI_VJP_STRIP=0", CHAINS_NS, dict(kind="diag")),                                                   # one VJP tile per workgroup (k_fr_vjp32m)
    ("MIVI_BATCH_GEN3=0,MIVI_VJP_STRIP=5", CHAINS_NS, dict(kind="dense")),
    ("MIVI_BATCH_GEN3=0", CHAINS_NS, dict(kind="diag")),                                                  # the lane-batched second-generation kernels (k_fr_prod32q + k_fr_vjp32s) where the batch engine would run
    ("MIVI_FB_LANES=7", CHAINS, dict(kind="diag")),                                                       # batch engine: seven estimates per step (25 estimates: four steps, the last one shorter)
    ("MIVI_DUMMY_DEFAULT=1", CHAINS_NS, dict(kind="diag")),                                               # (no switch: the batch engine, one step)
    ("MIVI_BATCH_GEN3=0", CHAINS_NS, dict(kind="dense")),                                                 # ... with the dense target (second product on k_fr_prod32m)
    ("MIVI_P2P_DIRECT=1", P2P, dict()),                                                                   # sharded estimates: direct staging instead of ring slots + push
    ("MIVI_DUMMY_DEFAULT=1", P2P, dict()),                                                                # (no switch: ring slots + push)
    ("MIVI_FB_STL=0", CHAINS_STL, dict(kind="diag")),                                                     # sticking-the-landing batches: the lanes' solves instead of the engine's C^-T product -- bitwise the single calls'
    ("MIVI_DUMMY_DEFAULT=1", CHAINS, dict(kind="dense")),                                                 # (no switch: the default interleaving)                                                                 # first-generation accumulation kernel
]


@pytest.mark.parametrize("switch,script,par", CASES, ids=[f"{c[0]}-{i}" for i, c in enumerate(CASES)])
def test_route_under_switch_agrees(switch, script, par):
    par = dict(par)
    par.setdefault("pre", "")
    code = script.format(**par)
    env = {k: v for k, v in os.environ.items() if not k.startswith("MIVI_")}
    for kv in switch.split(","):   # (the second-generation batch switches only apply once the batch engine is switched off: MIVI_BATCH_GEN3=0)
        k, v = kv.split("=")
        env[k] = v
    env["PYTHONPATH"] = ROOT
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
