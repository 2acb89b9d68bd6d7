"""Lanes (estimates) of a batch-engine dispatch from its grid -- derived, never assumed (round 4's pmc_traffic.py assumed 100 lanes per
launch while the launches carried 50).  Grids (kernels_fullrank_batch.hip host side): k_fb_prod: L (d/128)(M/128) workgroups;
k_fb_vjp: L (nrb (nrb + 1) / 2 + 1) workgroups (the lanes' value blocks ride in the launch); k_fb_eps: grid.y = L (+ the tril(C) plane riders
of a call's first draw: ceil(d/32 / gx) rows, gx = d/64 * M/32)."""


def disp_cols(con, disp):
    cols = [r[1] for r in con.execute(f"pragma table_info({disp})")]
    gx = "grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else None)
    gy = "grid_size_y" if "grid_size_y" in cols else ("grid_y" if "grid_y" in cols else None)
    wx = "workgroup_size_x" if "workgroup_size_x" in cols else ("workgroup_x" if "workgroup_x" in cols else None)
    wy = "workgroup_size_y" if "workgroup_size_y" in cols else ("workgroup_y" if "workgroup_y" in cols else None)
    return gx, gy, wx, wy


def lanes_of(kernel, ngx, ngy, d=1024, M=256):
    """ngx, ngy = workgroups along x / y.  None when the kernel is not a batch-engine kernel or the grid does not divide."""
    nrb, ncb = d // 128, M // 128
    if "k_fb_prod" in kernel:
        return ngx // (nrb * ncb) if ngx % (nrb * ncb) == 0 else None
    if "k_fb_vjp" in kernel:
        per = nrb * (nrb + 1) // 2 + 1
        return ngx // per if ngx % per == 0 else None
    if "k_fb_eps" in kernel:
        return ngy     # (riders included: the caller subtracts them when a product dispatch of ngy - riders lanes exists)
    return None


def eps_riders(d=1024, M=256):
    gx = (d // 64) * (M // 32)
    return -(-(4 * (d // 32)) // gx)   # (round 5: four rider workgroups per 32-row block of tril(C))
