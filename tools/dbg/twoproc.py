import os, sys, socket
sys.path.insert(0, '/root/repo')
import numpy as np, torch, torch.multiprocessing as mp
def worker(rank, world, port, family, d, M):
    import torch.distributed as dist, ctypes as C
    import advancedvi_jl_amd as avi
    from advancedvi_jl_amd.distributed import ShardPlan
    from tests.helpers import SEED, make_family, make_problem
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    rng = np.random.default_rng(7)
    q, _ = make_family(rng, d, family, np.float32); prob, _ = make_problem(rng, "diag", d, np.float32); params, _ = avi.destructure(q)
    full = avi.MiviContext(np.float32, family, d, M, 0, SEED); full.set_problem(prob)
    plan = ShardPlan(M, world)
    ctx = avi.MiviContext(np.float32, family, d, plan.count(rank), 0, SEED, m_offset=plan.offset(rank), m_total=M); ctx.set_problem(prob)
    mine = torch.frombuffer(bytearray(ctx.p2p_export(rank, world)), dtype=torch.uint8)
    blobs = [torch.zeros(256, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(blobs, mine)
    ctx.p2p_attach([bytes(b.numpy().tobytes()) for b in blobs])
    ctx.p2p_set_spin_budget(1 << 18)
    p = ctx.to_device(params)
    def words():
        buf = (C.c_uint32 * 128)(); ctx.lib.mivi_p2p_debug_words.argtypes = [C.c_void_p, C.c_void_p]; ctx.lib.mivi_p2p_debug_words(ctx.h, buf)
        return dict(l0=list(buf[0:2]), l1=list(buf[16:18]), ready=buf[64], freed=list(buf[80:84]))
    for idx in range(30, 33):
        dist.barrier()
        try:
            v, g = ctx.estimate_gradient_dist(p, idx); ctx.synchronize(); err = None
        except Exception as e: err = str(e)[:40]
        v_ref, g_ref = full.estimate_gradient(params, idx)
        print(rank, "single", idx, err, float(v), float(v_ref), words(), flush=True)
    v, g = ctx.empty(1), ctx.empty(ctx.params_len)
    for rep, cnt in enumerate((1, 2, 3, 9)):
        dist.barrier()
        try:
            ctx.estimate_gradient_dist_n(p, 60 + 10 * rep, cnt, v, g); ctx.synchronize(); err = None
        except Exception as e: err = str(e)[:40]
        v_ref, g_ref = full.estimate_gradient(params, 60 + 10 * rep + cnt - 1)
        print(rank, "batch", cnt, err, float(v), float(v_ref), float((g - g_ref).norm() / g_ref.norm()), words(), flush=True)
    dist.barrier(); dist.destroy_process_group()
if __name__ == "__main__":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mpc = mp.get_context("spawn")
    ps = [mpc.Process(target=worker, args=(r, 2, port, 1, 256, 256)) for r in range(2)]
    [p.start() for p in ps]; [p.join(200) for p in ps]
