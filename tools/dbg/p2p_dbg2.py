import numpy as np, torch, sys, struct, ctypes as C
sys.path.insert(0, '/root/repo')
import advancedvi_jl_amd as avi
from advancedvi_jl_amd.distributed import ShardPlan, p2p_geometry
from tests.helpers import SEED, make_family, make_problem
hip = C.CDLL("libamdhip64.so")
def readback(ptr, nbytes):
    buf = (C.c_char * nbytes)()
    hip.hipMemcpy(buf, C.c_void_p(ptr), C.c_size_t(nbytes), C.c_int(2))
    return bytes(buf)
def run_noexp(dtype, family, d, M, R, ent, ctxs, full, params):
    for idx in (17, 18):
        fp = full.estimate_partials(params, idx).double()
        tot = sum(c.estimate_partials(params, idx).double() for c in ctxs)
        print("noexp", family, d, M, R, idx, "perr", float((tot - fp).norm() / fp.norm()))
    for c in ctxs + [full]: c.close()

def run(dtype, family, d, M, R, ent, verbose):
    rng = np.random.default_rng(5)
    q, q_o = make_family(rng, d, family, dtype)
    prob, tgt_o = make_problem(rng, "diag", d, dtype)
    params, _ = avi.destructure(q)
    full = avi.MiviContext(dtype, family, d, M, ent, SEED); full.set_problem(prob)
    plan = ShardPlan(M, R)
    ctxs = []
    for r in range(R):
        c = avi.MiviContext(dtype, family, d, plan.count(r), ent, SEED, m_offset=plan.offset(r), m_total=M); c.set_problem(prob); ctxs.append(c)
    import os
    if os.environ.get("NOEXPORT") == "1":
        return run_noexp(dtype, family, d, M, R, ent, ctxs, full, params)
    hs = [c.p2p_export(r, R) for r, c in enumerate(ctxs)]
    ptrs = [struct.unpack_from("<Q", h, 64)[0] for h in hs]
    nbytes = [struct.unpack_from("<Q", h, 48)[0] for h in hs]
    import os
    NOP2P = os.environ.get("NOP2P") == "1"
    for c in ctxs: c.p2p_attach(hs); c.comm_set_route("p2p")
    L = ctxs[0].partials_len
    n, cn, G, vs = p2p_geometry(L, R)
    if verbose: print("L n cn G vs", L, n, cn, G, vs, [hex(p) for p in ptrs], nbytes)
    for idx in (17, 18):
        v_ref, g_ref = full.estimate_gradient(params, idx)
        parts, outs = [], []
        for r, c in enumerate(ctxs):
            P = c.empty(n * R).zero_(); c.estimate_partials(params, idx, P[:L]); parts.append(P)
            outs.append((c.empty(1), c.empty(c.params_len).fill_(float('nan'))))
        torch.cuda.synchronize()
        if verbose:
            from oracle import oracle as O
            _, tgt = make_problem(np.random.default_rng(5) if False else rng2, "diag", d, dtype) if False else (None, None)
        tot = sum(P.double() for P in parts)
        tdt = torch.float32 if dtype == np.float32 else torch.float64
        v0, g0 = ctxs[0].finalize(params, tot.to(tdt)[:L].contiguous())
        fp = full.estimate_partials(params, idx).double()
        if verbose:
            from oracle import oracle as O
            def e(a, b): return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))
            for r, c in enumerate(ctxs + [full]):
                _, eps = c.sample(params, idx)
                ref = O.estimate_gradient(O.destructure(q_o), d, family, tgt_o, eps.cpu().numpy().astype(np.float64), ent)["partials"]
                got = (parts[r][:L] if r < R else fp).cpu().numpy().astype(np.float64)
                print("   rank", r, "mu", e(got[:d], ref[:d]), "tri", e(got[d:L-2], ref[d:L-2]), "scal", got[L-2:], ref[L-2:])
        perr = float((tot[:L] - fp).norm() / fp.norm())
        torch.cuda.synchronize()
        for ph in (() if NOP2P else (1, 2, 4)):
            for r, c in enumerate(ctxs): c.p2p_exchange(c.to_device(params), parts[r], outs[r][0], outs[r][1], ph)
            torch.cuda.synchronize()
            if verbose and ph == 1:
                ep = idx - 16; p = ep & 1
                es = 4 if dtype == np.float32 else 8
                for s in range(R):
                    raw = np.frombuffer(readback(ptrs[s], 2 * R * n * es), dtype=np.float32 if dtype == np.float32 else np.float64).reshape(2, R, n)
                    for src in range(R):
                        exp = parts[src].cpu().numpy()[s * n:(s + 1) * n]
                        bad = np.flatnonzero(raw[p, src] != exp)
                        if bad.size: print(" epoch", ep, "stage owner", s, "src", src, "mismatch", bad.size, "first", bad[:5], raw[p, src][bad[:3]], exp[bad[:3]])
        errs = []
        for c in ctxs:
            try: c.synchronize()
            except Exception as e: errs.append(str(e)[:60])
        print(family, d, M, R, idx, "perr", perr, "fin", float(v0), float((g0 - g_ref).norm() / g_ref.norm()), "ref", float(v_ref), "p2p", [float(o[0]) for o in outs][:3], "gerr", float((outs[0][1] - g_ref).norm() / g_ref.norm()), errs[:1])
    for c in ctxs + [full]: c.close()
CASES=[(0, 64, 48, 4), (0, 3, 8, 8), (1, 96, 64, 2), (1, 40, 30, 3), (1, 128, 256, 8), (0, 1024, 256, 8), (1, 256, 256, 2), (1, 5, 7, 5)]
for (fam, d, M, R) in CASES:
    for dt in (np.float32, np.float64):
        run(dt, fam, d, M, R, 0, (fam, d, M, R) == (1, 256, 256, 2) and dt == np.float32)
