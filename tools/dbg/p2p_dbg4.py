import numpy as np, torch, sys, struct, ctypes as C
sys.path.insert(0, '/root/repo')
import advancedvi_jl_amd as avi
from advancedvi_jl_amd.distributed import ShardPlan, p2p_geometry
from tests.helpers import SEED, make_family, make_problem
hip = C.CDLL("libamdhip64.so")
def readback(ptr, nbytes):
    buf = (C.c_char * nbytes)(); hip.hipMemcpy(buf, C.c_void_p(ptr), C.c_size_t(nbytes), C.c_int(2)); return bytes(buf)
dtype, family, d, M, R, ent = np.float32, 0, 64, 48, 1, 0
rng = np.random.default_rng(5)
q, _ = make_family(rng, d, family, dtype); prob, _ = make_problem(rng, "diag", d, dtype); params, _ = avi.destructure(q)
c = avi.MiviContext(dtype, family, d, M, ent, SEED); c.set_problem(prob)
h = c.p2p_export(0, 1); ptr = struct.unpack_from("<Q", h, 64)[0]; nb = struct.unpack_from("<Q", h, 48)[0]
c.p2p_attach([h])
L = c.partials_len; n, cn, G, vs = p2p_geometry(L, 1)
print("L n cn G vs bytes", L, n, cn, G, vs, nb)
P = c.empty(n).zero_(); c.estimate_partials(params, 17, P[:L]); v, g = c.empty(1), c.empty(c.params_len).fill_(float('nan'))
torch.cuda.synchronize()
p_dev = c.to_device(params)
c.p2p_exchange(p_dev, P, v, g, 1); torch.cuda.synchronize()
raw = np.frombuffer(readback(ptr, nb), dtype=np.uint32)
stage = raw[: 2 * 1 * n * 2].reshape(2, n, 2)
Pw = P.cpu().numpy().view(np.uint32)
print("stage parity1 words == P:", np.array_equal(stage[1, :, 0], Pw), "flags", np.unique(stage[1, :, 1]), "first mism", np.flatnonzero(stage[1, :, 0] != Pw)[:10])
c.p2p_exchange(p_dev, P, v, g, 2); torch.cuda.synchronize()
raw = np.frombuffer(readback(ptr, nb), dtype=np.uint32)
off = (nb // 2) // 4
fin = raw[off: off + 2 * n * 2].reshape(2, n, 2)
ref = c.finalize(params, P[:L].contiguous())[1].cpu().numpy()
fw = fin[1, :2*d, 0].view(np.float32)
print("fin flags", np.unique(fin[1, :, 1]), "fin vs finalize mism", np.flatnonzero(~np.isclose(fw, ref, rtol=1e-6))[:10], fw[:6], ref[:6])
c.p2p_exchange(p_dev, P, v, g, 4); torch.cuda.synchronize()
gg = g.cpu().numpy()
print("grad mism", np.flatnonzero(~np.isclose(gg, ref, rtol=1e-6))[:10], gg[:6])
