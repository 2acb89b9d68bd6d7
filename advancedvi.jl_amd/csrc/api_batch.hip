// libmivi C ABI, part 5: estimates at FIXED parameters -- hipGraph chains, the lane-batched second-generation kernels and the
// third-generation batch engine (kernels_fullrank_batch.hip): mivi_estimate_gradient_n / _each.
#include "api_common.h"

// ---------------------------------------------------------------------------------------------
// hipGraph-batched estimates and the device-resident optimisation loop
// ---------------------------------------------------------------------------------------------
// The null stream cannot be captured: record on an internal stream, replay on the context's stream.
mivi_status_t begin_capture(mivi_ctx *c, hipStream_t *saved) {
  if (!c->cap_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking));

  HIPCHK(c, hipStreamSynchronize(c->stream));   // pending memsets / uploads on the launch stream
  *saved = c->stream;
  c->stream = c->cap_stream;
  hipError_t e = hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) { c->stream = *saved; c->err = std::string("hipStreamBeginCapture: ") + hipGetErrorString(e); return MIVI_ERR_HIP; }
  return MIVI_OK;
}
hipError_t end_capture(mivi_ctx *c, hipStream_t saved, hipGraph_t *graph) {
  hipError_t e = hipStreamEndCapture(c->cap_stream, graph);
  c->stream = saved;
  return e;
}

bool graph_capturable(const mivi_ctx *c) {   // every device-resident target (the host callback is not)
  return c->target == TGT_DIAG_GAUSS || c->target == TGT_DENSE_GAUSS || c->target == TGT_FUNNEL || c->target == TGT_LOGREG;
}
// allocations are not allowed inside a capture: size whatever the target's launchers would otherwise grow lazily
mivi_status_t reserve_target(mivi_ctx *c, int M) {
  if (c->target == TGT_LOGREG && !logreg_reserve(c, M)) return fail(c, MIVI_ERR_HIP, "logistic regression: scratch allocation failed");
  return MIVI_OK;
}

static mivi_status_t estimate_gradient_chain(mivi_ctx *c, const void *params, uint64_t idx0, int32_t count, void *value, void *grad) {
  if (!c || !params || !value || !grad || count <= 0) return MIVI_ERR_BAD_ARG;
  const uint64_t st = (uint64_t)c->idx_stride;   // estimates idx0, idx0 + st, ... (st = 1 unless this is one of several interleaved chains)
  if (!graph_capturable(c)) return fail(c, MIVI_ERR_UNSUPPORTED, "graph batching needs a device-resident built-in target");
  if (c->idx_src) return fail(c, MIVI_ERR_UNSUPPORTED, "an index source is set (mivi_set_index_source): graph-batched calls keep their own device counter");
  (void)hipSetDevice(c->cfg.device);
  mivi_status_t s = ensure_work(c, c->cfg.n_mc);
  if (s) return s;
  prepare_tables(c, c->cfg.n_mc);   // host->device uploads are not allowed inside the capture
  if ((s = reserve_target(c, c->cfg.n_mc))) return s;
  static const bool no_fused_loop_n = getenv("MIVI_NO_FUSED_LOOP") != nullptr;
  if (c->cfg.family == MIVI_MEANFIELD && c->target == TGT_DIAG_GAUSS && !c->bij_on && c->cfg.n_mc <= 4096 && !c->idx_src && !no_fused_loop_n) {
    // rows are independent for this family / target pair: all `count` estimates run inside ONE launch (every workgroup
    // keeps its four rows and walks the estimate indices), the value partials are reduced by a second launch
    const size_t hist_doubles = (size_t)count * 4 * (size_t)((c->cfg.d + 3) / 4);
    const size_t lane_bytes = (size_t)mf_loop_lanes(c, count) * 2 * (size_t)c->cfg.d * c->esize;   // the estimate lanes' gradient scratch
    if ((s = ensure(c, c->X, ((size_t)count + hist_doubles + 8) * sizeof(double) + lane_bytes, false))) return s;
    double *rec = (double *)c->X.p;
    launch_mf_sgd_loop(c, const_cast<void *>(params), nullptr, idx0, 0, count, -1, 0.0, (double)NAN, rec + count, rec, grad,
                       (void *)(rec + count + hist_doubles + 8), value);   // (the value kernel also leaves the last estimate's objective value: no launch of its own)
    HIPCHK(c, hipGetLastError());
    return MIVI_OK;
  }
  if (c->cfg.family == MIVI_MEANFIELD && c->target == TGT_FUNNEL && !c->funnel_constrained && !c->bij_on && c->cfg.n_mc <= 256 && !c->idx_src &&
      !no_fused_loop_n) {
    // fused funnel target (BASELINE config 5): the cross-row sums enter row 0 and ell linearly, so the batch is launch-free too --
    // per-estimate partials to a history buffer, one finishing workgroup per estimate (k_mf_funnel_loop / _value)
    const size_t d4 = (size_t)((c->cfg.d + 3) / 4);
    const size_t nd = (size_t)count * 6 * d4 + (size_t)count + 8;
    const size_t sc_bytes = ((size_t)count * ((size_t)c->cfg.d + 2) * c->esize + 63) & ~(size_t)63;
    const size_t lane_bytes = (size_t)mf_loop_lanes(c, count) * 2 * (size_t)c->cfg.d * c->esize;   // the estimate lanes' gradient scratch
    const size_t lane_al = (lane_bytes + 63) & ~(size_t)63;
    const size_t e0_bytes = (size_t)count * (size_t)c->cfg.n_mc * c->esize;   // eps[0, m] of every estimate, shared by the row quads
    if ((s = ensure(c, c->X, nd * sizeof(double) + sc_bytes + lane_al + e0_bytes + 64, false))) return s;
    double *hist = (double *)c->X.p, *elbo = hist + (size_t)count * 6 * d4;
    void *scratch = (void *)(elbo + count + 8);
    launch_mf_funnel_loop(c, params, idx0, count, hist, elbo, scratch, value, grad, (void *)((char *)scratch + sc_bytes),
                          (void *)((char *)scratch + sc_bytes + lane_al));
    HIPCHK(c, hipGetLastError());
    return MIVI_OK;
  }
  // The shortest batches run as an eager chain of the same launches: a graph replay carries ~25 us of fixed host cost (and its first
  // use a capture + instantiation), an eager chain ~17 us but ~0.7 us more per estimate (north star, n = 1 / 5 / 10 / 20 estimates
  // done after 31 / 93 / 168 / 314 us eagerly against 39 / 97 / 166 / 301 us replayed).  MIVI_GRAPH_MIN pins the
  // smallest batch that is captured.
  static const int graph_min = getenv("MIVI_GRAPH_MIN") ? atoi(getenv("MIVI_GRAPH_MIN")) : 6;
  if (count < graph_min && c->cfg.family == MIVI_FULLRANK) {
    Chain chn;
    chn.on = true;
    chn.estimates_only = true;
    for (int i = 0; i < count && s == MIVI_OK; ++i) {
      c->cur = i & 1;
      chn.has_next = (i + 1 < count);
      chn.next_rng = rng_of(c, idx0 + ((uint64_t)i + 1) * st);
      s = run_estimate(c, params, rng_of(c, idx0 + (uint64_t)i * st), c->cfg.n_mc, 1, final_out(c, value, grad), &chn);
    }
    if (s == MIVI_OK) flush_chain(c, params, &chn);
    c->cur = 0;
    c->pre_valid = false;
    if (s) return s;
    HIPCHK(c, hipGetLastError());
    return MIVI_OK;
  }
  GraphCache &g = c->graph;
  if (!(g.exec && g.kind == 1 && g.count == count && g.params == params && g.value == value && g.grad == grad)) {
    invalidate_graph(c);
    hipGraph_t graph = nullptr;
    hipStream_t saved;
    if ((s = begin_capture(c, &saved))) return s;
    Chain chn;
    chn.on = true;
    chn.estimates_only = true;
    for (int i = 0; i < count && s == MIVI_OK; ++i) {
      RngArgs r = rng_of(c, (uint64_t)i * st);
      r.idx_ptr = (const uint64_t *)c->d_idx.p;
      c->cur = i & 1;
      chn.has_next = (i + 1 < count);
      chn.next_rng = rng_of(c, ((uint64_t)i + 1) * st);
      chn.next_rng.idx_ptr = r.idx_ptr;
      s = run_estimate(c, params, r, c->cfg.n_mc, 1, final_out(c, value, grad), &chn);
    }
    if (s == MIVI_OK) flush_chain(c, params, &chn);
    // the graph leaves the device-side estimate counter at idx0 + count * st: a caller that walks the indices in order (an SGD-style
    // driver does) needs no counter-setting launch in front of the next replay
    if (s == MIVI_OK) hipLaunchKernelGGL(k_bump_u64, dim3(1), dim3(1), 0, c->stream, (uint64_t *)c->d_idx.p, (uint64_t)count * st);
    c->cur = 0;
    hipError_t e = end_capture(c, saved, &graph);
    if (s) { if (graph) (void)hipGraphDestroy(graph); return s; }
    HIPCHK(c, e);
    HIPCHK(c, hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    g.kind = 1; g.count = count; g.params = params; g.value = value; g.grad = grad;
  }
  if (!(c->d_idx_valid && c->d_idx_expect == idx0))
    hipLaunchKernelGGL(k_set_u64x2, dim3(1), dim3(1), 0, c->stream, (uint64_t *)c->d_idx.p, idx0, 0ull, 1);
  HIPCHK(c, hipGraphLaunch(g.exec, c->stream));
  c->d_idx_valid = true;
  c->d_idx_expect = idx0 + (uint64_t)count * st;
  return MIVI_OK;
}

// ---- interleaved chains ------------------------------------------------------------------------------------------------------------
static int chain_lanes() {   // developer override (A/B): MIVI_CHAINS = 1 .. 4
  static const int v = getenv("MIVI_CHAINS") ? atoi(getenv("MIVI_CHAINS")) : 0;
  return v;
}
mivi_status_t sync_kid(mivi_ctx *c, mivi_ctx *k, int lanes) {
  if (k->kid_gen == c->target_gen && k->idx_stride == lanes) return MIVI_OK;
  invalidate_graph(k);
  k->target = c->target;
  k->t_const = c->t_const;
  k->t_mean = c->t_mean; k->t_istd = c->t_istd; k->t_prec = c->t_prec;   // borrowed (is_child: never freed there)
  k->M_total = c->M_total;
  if (k->target == TGT_DENSE_GAUSS && !k->RT.p) k->cap_M = 0;              // (allocates its own transposed-sample buffer)
  k->idx_stride = lanes;
  k->kid_gen = c->target_gen;
  return MIVI_OK;
}

// children of an interleaved / lane-batched batch: the same configuration, their own stream and work buffers, the target borrowed
mivi_status_t ensure_kids(mivi_ctx *c, int lanes) {
  mivi_status_t s;
  while (c->n_kids < lanes - 1) {
    mivi_config_t cfg = c->cfg;
    cfg.stream = nullptr;
    cfg.own_stream = 1;
    mivi_ctx *k = nullptr;
    if ((s = mivi_create(&cfg, &k))) return fail(c, s, "interleaved chains: child context creation failed");
    k->is_child = true;
    const int j = c->n_kids;
    (void)hipFree(k->status.p);                                    // the child's sticky flags: word j + 1 of the parent's status buffer
    k->status.p = (char *)c->status.p + sizeof(int) * (j + 1);    // (borrowed: is_child contexts never free it)
    if ((s = ensure(c, c->kid_out[j], 16 + (size_t)mivi_params_len(c) * c->esize, false))) { (void)mivi_destroy(k); return s; }
    if (hipEventCreateWithFlags(&c->ev_join[j], hipEventDisableTiming) != hipSuccess ||
        (!c->ev_fork && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess)) {
      (void)hipGetLastError();
      (void)mivi_destroy(k);   // (not yet registered with the parent: nobody else would)
      return fail(c, MIVI_ERR_HIP, "interleaved chains: event creation failed");
    }
    c->kids[c->n_kids++] = k;
  }
  return MIVI_OK;
}

// ---- third-generation batch engine (kernels_fullrank_batch.hip) -------------------------------------------------------------------------
// `count` estimates at the same parameters as steps of up to fb_lanes_max() LANES: a step is four launches (eps, product + target, VJP,
// values) that cover all of its lanes.  No child contexts, no forked graph: a lane's buffers are base + lane * stride.
static int fb_lanes_max() {
  // MIVI_FB_LANES: estimates per step (A/B).  Per estimate (eps + product + VJP, tools/fb_lane_curve.py, north-star shape): 8 lanes 8.4 us,
  // 20: 4.6, 32: 4.05, 48: 3.95, 80: 3.90, 100: 4.05, 128: 4.11 -- the triangular product is paced by its heaviest tile up to ~30 lanes, and
  // beyond ~90 the step's planes (4.5 MB per lane) outgrow the 256 MB memory-side cache (the draws' writes slow down first).  So long batches
  // are cut into equal steps of at most 80 lanes (100 estimates: two steps of 50).
  static const int v = getenv("MIVI_FB_LANES") ? atoi(getenv("MIVI_FB_LANES")) : 80;
  return v < 1 ? 1 : (v > 256 ? 256 : v);
}
static bool fb_stl_on() {   // MIVI_FB_STL=0: the sticking-the-landing estimators keep the lane-batched second-generation kernels + solves (A/B)
  static const bool off = getenv("MIVI_FB_STL") && atoi(getenv("MIVI_FB_STL")) == 0;
  return !off;
}
static bool fb_route(const mivi_ctx *c, const void *params, const void *grad_last, const void *grads_all) {
  const bool stl = c->cfg.entropy == MIVI_ENT_STL || c->cfg.entropy == MIVI_ENT_STL_ZERO_GRAD;
  return !c->is_child && c->cfg.family == MIVI_FULLRANK && c->cfg.dtype == MIVI_F32 && !c->bij_on && !c->idx_src && !c->dbg &&
         (c->target == TGT_DIAG_GAUSS || c->target == TGT_DENSE_GAUSS) && (!stl || (fb_stl_on() && stl2_shape_ok(c, c->cfg.d))) &&
         fb_shape_ok(c, c->cfg.n_mc) && ((!stl && c->target == TGT_DIAG_GAUSS) || fb_whole_tiles(c, c->cfg.n_mc)) &&   // (padded geometry: the diagonal target, no STL term)
         ((uintptr_t)params & 15) == 0 && ((uintptr_t)grad_last & 15) == 0 && ((uintptr_t)grads_all & 15) == 0;
}
// value_last / grad_last: the batch's LAST estimate (mivi_estimate_gradient_n's contract), or nullptr; values_all T[count] / grads_all
// T[count * params_len]: every estimate's (mivi_estimate_gradient_each), or nullptr (lane scratch)
// obj_ent >= 0: objective mode (fb_objective): the `count` lanes are consecutive blocks of n_mc samples of estimate idx0, values only, the
// value's entropy estimator obj_ent
// dist: a SHARDED batch (mivi_estimate_gradient_dist_n on an engine shape): this context draws its n_mc columns of every estimate's n_mc x world
// samples; the VJP launch leaves the lanes' partial vectors, ONE all-reduce per step sums them over the ranks (dist_allreduce_f32: nothing
// to do on one rank), k_fb_finalize_parts turns the sums into the lanes' values and gradients.
static mivi_status_t fb_batch(mivi_ctx *c, const void *params, uint64_t idx0, int count, void *value_last, void *grad_last, void *values_all,
                              void *grads_all, int obj_ent = -1, bool dist = false) {
  mivi_status_t s;
  const int M = c->cfg.n_mc, d = c->cfg.d;
  if ((s = ensure_work(c, M))) return s;
  // sharded batches with a communicator: steps of at most 24 lanes, so that a 100-estimate call is five steps whose all-reduces (on
  // comm_stream, the partial vectors double-buffered) run UNDER the next steps' kernels -- across ranks the exchange, not the kernels, paces
  // the batch
  const bool overlap = dist && c->comm != nullptr;
  const int Lmax = (overlap && c->comm_world > 1) ? (fb_lanes_max() < 24 ? fb_lanes_max() : 24) : fb_lanes_max();   // (one rank: nothing to hide, the widest steps)
  const int steps = (count + Lmax - 1) / Lmax, L = (count + steps - 1) / steps, Llast = count - (steps - 1) * L;
  if (overlap && (s = dist_comm_stream(c))) return s;
  const size_t plen = (size_t)mivi_params_len(c);
  const int dG = (d + 127) / 128 * 128, MG = (M + 127) / 128 * 128;   // the engine's geometry: whole 128 x 128 tiles (kernels_fullrank_batch.hip fb_pad)
  FbTables &t = c->fb;
  if (t.cap_L < L || t.cap_M != M) {
    invalidate_graph(c);
    const size_t pw = fb_plane_words(c, M) * 4;
    if ((s = ensure(c, t.CA, fb_cplane_words(c) * 4, false)) || (s = ensure(c, t.epsP, (size_t)L * pw, false)) ||
        (s = ensure(c, t.WV, (size_t)L * pw, false)) ||
        (s = ensure(c, t.ell, (size_t)L * (dG / 32) * (MG / 32) * sizeof(double), false)) ||
        (s = ensure(c, t.he, (size_t)L * (dG / 64) * (MG / 32) * sizeof(double), false)) ||
        (s = ensure(c, t.ld, 2 * (size_t)(dG / 32) * sizeof(double) + 64, false)) || (s = ensure(c, t.values, (size_t)L * 4 + 64, false)) ||
        (s = ensure(c, t.cscale, 2 * (size_t)dG * 4, false)) || (s = ensure(c, t.winv, (size_t)L * (MG / 128) * dG * 4, false)))
      return s;
    t.grads.bytes = 0;   // (re-zeroed: the lanes' scratch gradients rely on exact zeros above the diagonal that no kernel writes)
    if ((s = ensure(c, t.grads, (size_t)L * plen * 4, true))) return s;
    t.cap_L = L;
    t.cap_M = M;
    t.cap_LR = 0;
  }
  const size_t part_len = dist ? fb_part_len(c) : 0;
  if (dist && t.cap_LP < L) {
    invalidate_graph(c);
    if ((s = ensure(c, t.parts, 2 * (size_t)L * part_len * 4, false))) return s;   // (two sets: the all-reduce of step s under the kernels of step s + 1)
    t.cap_LP = L;
  }
  const bool dense = c->target == TGT_DENSE_GAUSS;
  if (dense) {   // R = Z - m planes per lane; the planes of P once per target
    if (t.cap_LR < L) {
      invalidate_graph(c);
      if ((s = ensure(c, t.RP, (size_t)L * fb_plane_words(c, M) * 4, false)) || (s = ensure(c, t.PA, fb_cplane_words(c) * 4, false)) ||
          (s = ensure(c, t.pscale, 2 * (size_t)d * 4, false)) || (s = ensure(c, t.rinv, (size_t)L * (d / 128) * M * 4, false)))
        return s;
      t.cap_LR = L;
      t.PA_valid = false;
    }
    if (!t.PA_valid) {
      fb_launch_pplanes(c, c->stream);
      t.PA_valid = true;
    }
  }
  const bool values_only = obj_ent >= 0 || (!grads_all && !grad_last);
  const bool stl = !values_only && (c->cfg.entropy == MIVI_ENT_STL || c->cfg.entropy == MIVI_ENT_STL_ZERO_GRAD);   // (a value needs no gradient term)
  if (stl) {
    // W += C^-T eps: inside a call the parameters are fixed, so C^-T is formed ONCE -- the solve kernels (kernels_stl.hip) on the identity's d
    // columns -- and the term is one more triangular product per lane (k_fb_prod<FB_STL_U>)
    const size_t es = 4;
    if (!t.Eye.p) {
      std::vector<float> eye((size_t)d * d, 0.f);
      for (int i = 0; i < d; ++i) eye[(size_t)i * d + i] = 1.f;
      if ((s = ensure(c, t.Eye, (size_t)d * d * es, false))) return s;
      HIPCHK(c, hipMemcpy(t.Eye.p, eye.data(), (size_t)d * d * es, hipMemcpyHostToDevice));
    }
    const bool grow = !t.Tinv.p || c->stl_X.bytes < ((size_t)d * d + (size_t)(d / 2) * (d / 2)) * es + 4096 || !c->stl_F.p;
    if (grow) {
      invalidate_graph(c);
      if ((s = ensure(c, t.Tinv, (size_t)d * d * es, false)) || (s = ensure(c, t.TA, fb_cplane_words(c) * 4, false)) || (s = ensure(c, t.tscale, 2 * (size_t)d * 4, false)) ||
          (s = ensure(c, c->stl_X, ((size_t)d * d + (size_t)(d / 2) * (d / 2)) * es + 4096, false)) ||
          (s = ensure(c, c->stl_F, mivi::stl_pack_units(d) * 4, false)))
        return s;
    }
    launch_stl2(c, params, d, false, t.Eye.p, t.Tinv.p, true);
    fb_launch_tplanes(c, c->stream);
  }
  const FbTab *tabF = fb_prepare(c, M, L), *tabL = Llast != L ? fb_prepare(c, M, Llast) : tabF;
  if (Llast != L) tabF = fb_prepare(c, M, L);   // (re-resolve: four table slots, round robin)
  if (!tabF || !tabL) return fail(c, MIVI_ERR_HIP, "batch engine: work table allocation failed");
  auto make_step = [&](int st) {
    FbStep fs{};
    fs.params = params;
    fs.M = M;
    fs.L = st == steps - 1 ? Llast : L;
    fs.tab = st == steps - 1 ? tabL : tabF;
    fs.rng = rng_of(c, obj_ent >= 0 ? idx0 : idx0 + (uint64_t)st * L);
    fs.obj = obj_ent >= 0 ? 1 : 0;
    fs.ent_kind = obj_ent;
    fs.values_only = values_only ? 1 : 0;
    if (obj_ent >= 0) fs.rng.m_offset += st * L * M;
    if (grads_all) { fs.grads = (char *)grads_all + (size_t)st * L * plen * 4; fs.grad_stride = (long long)plen; fs.write_upper = 1; }
    else { fs.grads = t.grads.p; fs.grad_stride = (long long)plen; fs.write_upper = 0; }
    if (values_all) { fs.values = (char *)values_all + (size_t)st * L * 4; fs.value_stride = 1; }
    else { fs.values = t.values.p; fs.value_stride = 1; }
    fs.lane_last = -1;
    fs.dense = dense ? 1 : 0;
    fs.stl = stl ? 1 : 0;
    if (st == steps - 1 && (value_last || grad_last)) { fs.lane_last = Llast - 1; fs.grad_last = grad_last; fs.value_last = value_last; }
    if (dist) { fs.parts = (char *)t.parts.p + (size_t)(overlap ? (st & 1) : 0) * (size_t)t.cap_LP * part_len * 4; fs.part_stride = (long long)part_len; }
    return fs;
  };
  // One stream, no graph: per step {draws (+ tril(C)'s planes as riders of the first) -> product -> VJP + values} = three launches for up to
  // fb_lanes_max() estimates; the host is far ahead of the device.  (Measured and dropped: the batch as ONE hipGraph -- 4.45 against 4.26 us
  // per estimate in 100-estimate batches -- and the draws of step s + 1 on a second graph branch beside the products of step s: the draws
  // are bound by their 3 MB of plane writes per estimate and by the vector ALU, beside them the products ran 25 % longer: 4.58 us.)
  for (int st = 0; st < steps; ++st) {
    const FbStep fs = make_step(st);
    const int b = st & 1;
    if (overlap && st >= 2) HIPCHK(c, hipStreamWaitEvent(c->stream, c->fb_ev_comm[b], 0));   // the finalisation of step st - 2 has read this set of partial vectors
    fb_launch_eps(c, fs, st == 0, c->stream);
    fb_launch_compute(c, fs, c->stream);
    if (dist) {
      hipStream_t xs = c->stream;
      if (overlap) {
        HIPCHK(c, hipEventRecord(c->fb_ev_part[b], c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->fb_comm_stream, c->fb_ev_part[b], 0));
        xs = c->fb_comm_stream;
      }
      if ((s = dist_allreduce_f32(c, fs.parts, (size_t)fs.L * part_len, xs))) return s;
      fb_launch_finalize_parts(c, fs, xs);
      if (overlap) HIPCHK(c, hipEventRecord(c->fb_ev_comm[b], c->fb_comm_stream));
    }
  }
  if (overlap) {   // join: the batch is complete when its last two exchanges + finalisations are
    if (steps >= 2) HIPCHK(c, hipStreamWaitEvent(c->stream, c->fb_ev_comm[(steps - 2) & 1], 0));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->fb_ev_comm[(steps - 1) & 1], 0));
  }
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}

mivi_status_t mivi::fb_objective(mivi_ctx *c, const void *params, uint64_t idx, int lanes, int entropy, void *values) {
  if (!fb_route(c, params, nullptr, nullptr) || c->M_total != c->cfg.n_mc) return MIVI_ERR_UNSUPPORTED;
  return fb_batch(c, params, idx, lanes, nullptr, nullptr, values, nullptr, entropy);
}

// Lanes per step of a `count`-estimate batch on the batch engine (equal steps of at most MIVI_FB_LANES = 80): what a roofline leg must profile.
int32_t mivi_batch_lanes(const mivi_ctx_t *c, int32_t count) {
  if (!c || count <= 0) return 0;
  const int Lmax = fb_lanes_max(), steps = (count + Lmax - 1) / Lmax;
  return (count + steps - 1) / steps;
}

int32_t mivi_batch_info(const mivi_ctx_t *c, const void *params, int32_t what) {
  if (!c) return 0;
  if (what == 1) return 3;   // fr_planes.h kSplitProducts
  if (what == 2) return 4;   // two f16 planes
  if (what == 3) return c->exchange_lost ? 1 : 0;
  return (what == 0 && params && fb_route(c, params, nullptr, nullptr)) ? 1 : 0;
}

// Roofline leg of the batch engine: `reps` launches of each of a step's kernels for `lanes` estimates, hipEvents on the context's
// stream.  us_out[0..4] = average launch duration (us) of the draws, the product (+ fused diagonal target), the VJP (+ values), the dense
// target's product (0 with the diagonal target), the sticking-the-landing product (0 with the other estimators).
mivi_status_t mivi_profile_batch(mivi_ctx_t *c, const void *params, int32_t lanes, int32_t reps, double *us_out) {
  if (!c || !params || lanes <= 0 || reps <= 0 || !us_out) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  if (!fb_route(c, params, nullptr, nullptr)) return fail(c, MIVI_ERR_UNSUPPORTED, "mivi_profile_batch: this configuration does not take the batch engine");
  if (lanes > fb_lanes_max()) lanes = fb_lanes_max();
  char *o = (char *)c->tmp_out.p;
  mivi_status_t s = fb_batch(c, params, 1, lanes, o, o + 16, nullptr, nullptr);   // buffers, tables, operand planes of every lane
  if (s) return s;
  const FbTab *tab = fb_prepare(c, c->cfg.n_mc, lanes);
  if (!tab) return fail(c, MIVI_ERR_HIP, "batch engine: work table allocation failed");
  FbStep fs{};
  fs.params = params; fs.M = c->cfg.n_mc; fs.L = lanes; fs.tab = tab;
  fs.rng = rng_of(c, 1);
  fs.grads = c->fb.grads.p; fs.grad_stride = (long long)mivi_params_len(c); fs.values = c->fb.values.p; fs.value_stride = 1; fs.lane_last = -1;
  fs.dense = c->target == TGT_DENSE_GAUSS ? 1 : 0;
  fs.stl = (c->cfg.entropy == MIVI_ENT_STL || c->cfg.entropy == MIVI_ENT_STL_ZERO_GRAD) ? 1 : 0;
  us_out[3] = us_out[4] = 0.0;
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0));
  HIPCHK(c, hipEventCreate(&e1));
  for (int which = 0; which < 5; ++which) {
    if ((which == 3 && !fs.dense) || (which == 4 && !fs.stl)) continue;
    for (int r = -2; r < reps; ++r) {
      if (r == 0) HIPCHK(c, hipEventRecord(e0, c->stream));
      if (which == 0) fb_launch_eps(c, fs, true, c->stream);
      else fb_launch_compute(c, fs, c->stream, which == 1 ? 1 : (which == 2 ? 2 : (which == 3 ? 4 : 8)));
    }
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    us_out[which] = (double)ms * 1e3 / reps;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}

// mivi_estimate_gradient_dist_n on a batch-engine shape (api_dist.hip decides): every rank runs the engine on ITS sample columns, the lanes'
// partial vectors cross the ranks in one all-reduce per step
bool fb_dist_route(const mivi_ctx *c, const void *params, const void *grad) { return fb_route(c, params, grad, nullptr) && fb_whole_tiles(c, c->cfg.n_mc); }
mivi_status_t fb_batch_dist(mivi_ctx *c, const void *params, uint64_t idx0, int count, void *value, void *grad) {
  return fb_batch(c, params, idx0, count, value, grad, nullptr, nullptr, -1, true);
}

mivi_status_t mivi_estimate_gradient_each(mivi_ctx_t *c, const void *params, uint64_t idx0, int32_t count, void *values, void *grads) {
  if (!c || !params || !values || count <= 0) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  if (c->target == TGT_NONE) return fail(c, MIVI_ERR_NO_TARGET, "no target set");
  if (fb_route(c, params, nullptr, grads)) return fb_batch(c, params, idx0, count, nullptr, nullptr, values, grads);
  // every other configuration: the single calls, one after the other (results are those of mivi_estimate_gradient by definition)
  const size_t plen = (size_t)mivi_params_len(c);
  mivi_status_t s = MIVI_OK;
  for (int i = 0; i < count && s == MIVI_OK; ++i)
    s = run_estimate(c, params, rng_of(c, idx0 + (uint64_t)i), c->cfg.n_mc, 1,
                     final_out(c, (char *)values + (size_t)i * c->esize, grads ? (void *)((char *)grads + (size_t)i * plen * c->esize) : c->tmp_out.p));
  return s;
}

mivi_status_t mivi_estimate_gradient_n(mivi_ctx_t *c, const void *params, uint64_t idx0, int32_t count, void *value, void *grad) {
  if (!c || !params || !value || !grad || count <= 0) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  if (count >= 2 && fb_route(c, params, grad, nullptr)) return fb_batch(c, params, idx0, count, value, grad, nullptr, nullptr);
  // Several interleaved chains pay when an estimate is a short chain of latency-bound launches (the second-generation full-rank
  // kernels at the BASELINE sizes: two launches of 6-8 us that leave most CUs idle half of the time).  One chain otherwise.
  int lanes = 1;
  if (!c->is_child && c->cfg.family == MIVI_FULLRANK && c->cfg.dtype == MIVI_F32 && !c->bij_on && !c->idx_src && !c->dbg &&
      (c->target == TGT_DIAG_GAUSS || c->target == TGT_DENSE_GAUSS) && lds_path_shape_ok(c, c->cfg.n_mc) &&
      (long long)c->cfg.d * c->cfg.n_mc <= 2048LL * 512)
    lanes = 4;   // (measured at the north star, us per estimate with 1 / 2 / 3 / 4 chains: isolated 20-estimate calls 14.4 / 13.9 / 13.3 / 10.8,
                 //  100-estimate calls back to back 13.4 / 9.7 / 8.9 / 8.1 -- the product kernel's 64 KiB of LDS lets two of them, or one and
                 //  a VJP workgroup, share a CU)
  // Without a sticking-the-landing solve between the two kernels the contexts are LANE-BATCHED: E = 4 contexts per graph branch whose
  // product kernels are ONE launch (blockIdx.y = lane) and so are their VJP kernels -- half the launches per estimate, no fork / join for a
  // short batch (4 contexts, one branch: isolated 20-estimate calls 10.6 -> 9.7 us per estimate), two branches of four from twelve estimates on
  // (since the lane-batched launches are kernels of their own -- k_fr_prod32q, k_fr_vjp32s: one product + one VJP workgroup fit a CU -- the second
  // branch pays for 20-estimate calls too: 9.0 -> 8.1 us; 100-estimate calls back to back 7.0 us; 8, 12 and 16 contexts agree there).
  // MIVI_LANE_BATCH=0 keeps every context on a branch of its own (A/B reference; the STL estimators always do); MIVI_CHAINS = contexts.
  static const int lane_env = getenv("MIVI_LANE_BATCH") ? atoi(getenv("MIVI_LANE_BATCH")) : -1;
  const bool stl_ent = c->cfg.entropy == MIVI_ENT_STL || c->cfg.entropy == MIVI_ENT_STL_ZERO_GRAD;
  int lane_e = 0;   // contexts per branch (0: one each)
  if (lanes > 1 && (!stl_ent || stl2_shape_ok(c, c->cfg.n_mc)) && lds_use_prod32(c, c->cfg.n_mc) && lds_bf16x3() && ((uintptr_t)params & 15) == 0 &&
      ((uintptr_t)grad & 15) == 0 && lane_env != 0) {
    lane_e = lane_env > 0 ? (lane_env > 4 ? 4 : lane_env) : 4;
    lanes = count < 12 ? 4 : 8;   // (isolated batches at the north star, 4 vs 8 contexts, us per estimate: 8: 10.8 / 10.4, 10: 11.0 / 11.5, 12: 9.9 / 9.4, 20: 9.1 / 8.1, 48: 8.3 / 7.2)
  }
  if (chain_lanes() > 0) lanes = c->is_child ? 1 : (chain_lanes() > mivi_ctx::kMaxKids + 1 ? mivi_ctx::kMaxKids + 1 : chain_lanes());
  if (lane_e > 0 && (lanes % lane_e != 0 || count < lanes)) lane_e = 0;
  if (lane_e <= 0 && lanes > 4) lanes = 4;          // (as graph BRANCHES: at most four -- see kMaxKids)
  if (lane_e > 0 && lanes / lane_e > 4) lanes = 4 * lane_e;
  while (lanes > 1 && count < 4 * lanes && lane_e <= 0) --lanes;   // (short batches: not worth the fork / join)
  if (lanes <= 1) {
    if (!c->is_child && c->idx_stride != 1) { invalidate_graph(c); c->idx_stride = 1; }
    return estimate_gradient_chain(c, params, idx0, count, value, grad);
  }
  mivi_status_t s;
  if ((s = ensure_kids(c, lanes))) return s;
  if (c->idx_stride != lanes) { invalidate_graph(c); c->idx_stride = lanes; }
  for (int j = 0; j < lanes - 1; ++j)
    if ((s = sync_kid(c, c->kids[j], lanes))) return s;
  // chain q serves estimates idx0 + q, idx0 + q + lanes, ...; the chain that holds the LAST estimate writes the caller's buffers.
  // ONE hipGraph for the whole batch: the children's streams join the capture behind one fork event, so the batch is one graph
  // launch with `lanes` parallel branches and one join (two graph launches + events per call cost a 20-estimate batch what the
  // overlap gained: 14.7 us per estimate against 14.4 with one chain, 9.7 in steady state).
  const int q_last = (count - 1) % lanes;
  if ((s = ensure_work(c, c->cfg.n_mc))) return s;
  prepare_tables(c, c->cfg.n_mc);
  for (int j = 0; j < lanes - 1; ++j) {
    mivi_ctx *k = c->kids[j];
    if ((s = ensure_work(k, k->cfg.n_mc))) { c->err = k->err; return s; }
    prepare_tables(k, k->cfg.n_mc);
    if (!lds_prepare(k, k->cfg.n_mc)) return fail(c, MIVI_ERR_HIP, "full-rank work lists: allocation failed");
  }
  if (!lds_prepare(c, c->cfg.n_mc)) return fail(c, MIVI_ERR_HIP, "full-rank work lists: allocation failed");
  GraphCache &g = c->graph;
  // LANE-BATCHED contexts: the launchers of the two kernels record their arguments into a sink instead of launching
  // (kernels_fullrank_lds.hip: launch_lanes_*), the driver issues one launch per kernel and branch.
  if (lane_e > 0) {
    const int lane_mode = lane_e;
    const int E = lane_mode, B = lanes / E;
    if (!(g.exec && g.kind == 3 && g.count == count && g.params == params && g.value == value && g.grad == grad && g.p0 == (double)(lanes * 16 + E))) {
      invalidate_graph(c);
      c->idx_stride = lanes;
      for (int j = 0; j < lanes - 1; ++j) {
        c->kids[j]->kid_gen = c->target_gen;
        HIPCHK(c, hipStreamSynchronize(c->kids[j]->stream));
      }
      hipGraph_t graph = nullptr;
      hipStream_t saved;
      if ((s = begin_capture(c, &saved))) return s;
      hipError_t he = hipEventRecord(c->ev_fork, c->stream);
      mivi_ctx *ctxs[1 + mivi_ctx::kMaxKids];
      ctxs[0] = c;
      for (int l = 1; l < lanes; ++l) ctxs[l] = c->kids[l - 1];
      const bool dense = c->target == TGT_DENSE_GAUSS;
      // branch b: contexts b E .. b E + E - 1 (global lane g serves estimates g, g + lanes, ...), their launches on the stream of the branch's
      // first context; its product kernels are ONE launch (blockIdx.y = lane) and so are its VJP kernels
      auto branch = [&](int b) -> mivi_status_t {
        mivi_status_t st = MIVI_OK;
        LaneSink *sink = lane_sinks_alloc(E);
        StlSink *ssink = stl_ent ? stl_sinks_alloc(E) : nullptr;
        EpsSink *esink = eps_sink_alloc();
        Chain chn[4];
        hipStream_t bs = ctxs[b * E]->stream, kept[4];
        mivi_ctx *lead = ctxs[b * E];
        for (int l = 0; l < E; ++l) {
          mivi_ctx *k = ctxs[b * E + l];
          kept[l] = k->stream;
          k->stream = bs;   // (the other lanes' few stand-alone launches -- the first eps, the last value -- go to the branch's stream too)
          chn[l].on = true; chn[l].estimates_only = true;
          k->lane_sink = sink; k->lane_id = l;
          k->stl_sink = ssink;
          k->eps_sink = esink;
        }
        const int steps = (count + lanes - 1) / lanes;
        for (int i = 0; i < steps && st == MIVI_OK; ++i) {
          int L = 0;
          eps_sink_reset(esink);
          for (int l = 0; l < E && st == MIVI_OK; ++l) {
            const int gl = b * E + l;
            const int cnt = (count - gl + lanes - 1) / lanes;   // estimates of global lane gl: gl, gl + lanes, ...
            if (i >= cnt) break;
            mivi_ctx *k = ctxs[gl];
            lane_sink_reset(sink, l);
            if (ssink) stl_sink_reset(ssink, l);
            RngArgs r = rng_of(k, (uint64_t)gl + (uint64_t)i * lanes);
            r.idx_ptr = (const uint64_t *)c->d_idx.p;   // ONE device counter (the parent's) for all lanes
            k->cur = i & 1;
            chn[l].has_next = (i + 1 < cnt);
            chn[l].next_rng = rng_of(k, (uint64_t)gl + ((uint64_t)i + 1) * lanes);
            chn[l].next_rng.idx_ptr = r.idx_ptr;
            char *ko = gl ? (char *)c->kid_out[gl - 1].p : (char *)c->tmp_out.p;
            st = run_estimate(k, params, r, k->cfg.n_mc, 1, final_out(k, gl == q_last ? value : (void *)ko, gl == q_last ? grad : (void *)(ko + 16)), &chn[l]);
            if (st) { c->err = k->err; break; }
            if (lane_sink_counts(sink, l) != (dense ? 2 : 1) * 16 + 1 || (ssink && stl_sink_count(ssink, l) != 1))
              st = fail(c, MIVI_ERR_HIP, "lane-batched estimates: an estimate did not take the expected kernel route");
            ++L;
          }
          if (st == MIVI_OK && L > 0) launch_lanes_eps(lead, esink, L);   // (the lanes' first draws, if this is their first estimate: one launch)
          if (st == MIVI_OK && L > 0 && !(launch_lanes_prod(lead, sink, L, 0) && (!dense || launch_lanes_prod(lead, sink, L, 1)) &&
                                          (!ssink || launch_lanes_stl(lead, ssink, L, i == 0)) && launch_lanes_vjp(lead, sink, L)))
            st = fail(c, MIVI_ERR_HIP, "lane-batched estimates: the lanes' launches do not match");
        }
        ValueSink *vsink = value_sink_alloc();   // the lanes' closing value kernels (the last estimate of every chain): one launch
        for (int l = 0; l < E; ++l) {
          mivi_ctx *k = ctxs[b * E + l];
          k->lane_sink = nullptr;
          k->stl_sink = nullptr;
          k->eps_sink = nullptr;
          k->value_sink = vsink;
          if (st == MIVI_OK) flush_chain(k, params, &chn[l]);
          k->value_sink = nullptr;
          k->cur = 0;
          k->pre_valid = false;
        }
        if (st == MIVI_OK) launch_lanes_value(lead, params, vsink);
        value_sink_free(vsink);
        for (int l = 0; l < E; ++l) ctxs[b * E + l]->stream = kept[l];
        lane_sinks_free(sink);
        if (ssink) stl_sinks_free(ssink);
        eps_sink_free(esink);
        return st;
      };
      for (int b = 1; b < B && s == MIVI_OK && he == hipSuccess; ++b) {
        mivi_ctx *k = ctxs[b * E];
        he = hipStreamWaitEvent(k->stream, c->ev_fork, 0);   // the branch's stream joins the capture
        if (he != hipSuccess) break;
        s = branch(b);
        if (s == MIVI_OK) he = hipEventRecord(c->ev_join[b * E - 1], k->stream);
      }
      if (s == MIVI_OK && he == hipSuccess) s = branch(0);
      for (int b = 1; b < B && s == MIVI_OK && he == hipSuccess; ++b) he = hipStreamWaitEvent(c->stream, c->ev_join[b * E - 1], 0);
      if (s == MIVI_OK && he == hipSuccess) hipLaunchKernelGGL(k_bump_u64, dim3(1), dim3(1), 0, c->stream, (uint64_t *)c->d_idx.p, (uint64_t)count);
      hipError_t e = end_capture(c, saved, &graph);
      if (s) { if (graph) (void)hipGraphDestroy(graph); return s; }
      HIPCHK(c, he);
      HIPCHK(c, e);
      HIPCHK(c, hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0));
      (void)hipGraphDestroy(graph);
      g.kind = 3; g.count = count; g.params = params; g.value = value; g.grad = grad; g.p0 = (double)(lanes * 16 + E);
    }
    if (!(c->d_idx_valid && c->d_idx_expect == idx0))
      hipLaunchKernelGGL(k_set_u64x2, dim3(1), dim3(1), 0, c->stream, (uint64_t *)c->d_idx.p, idx0, 0ull, 1);
    HIPCHK(c, hipGraphLaunch(g.exec, c->stream));
    c->d_idx_valid = true;
    c->d_idx_expect = idx0 + (uint64_t)count;
    return MIVI_OK;
  }
  if (!(g.exec && g.kind == 2 && g.count == count && g.params == params && g.value == value && g.grad == grad && g.p0 == (double)lanes)) {
    invalidate_graph(c);
    c->idx_stride = lanes;   // (invalidate_graph leaves it; the children were synced above: re-stamp their generation)
    for (int j = 0; j < lanes - 1; ++j) {
      c->kids[j]->kid_gen = c->target_gen;
      HIPCHK(c, hipStreamSynchronize(c->kids[j]->stream));   // (their table uploads, before the capture -- not on every replay)
    }
    hipGraph_t graph = nullptr;
    hipStream_t saved;
    if ((s = begin_capture(c, &saved))) return s;
    hipError_t he = hipEventRecord(c->ev_fork, c->stream);
    auto chain_body = [&](mivi_ctx *k, int q, int cnt, void *v, void *gr) -> mivi_status_t {
      Chain chn;
      chn.on = true;
      chn.estimates_only = true;
      mivi_status_t st = MIVI_OK;
      for (int i = 0; i < cnt && st == MIVI_OK; ++i) {
        RngArgs r = rng_of(k, (uint64_t)q + (uint64_t)i * lanes);
        r.idx_ptr = (const uint64_t *)c->d_idx.p;   // ONE device counter (the parent's) for all chains
        k->cur = i & 1;
        chn.has_next = (i + 1 < cnt);
        chn.next_rng = rng_of(k, (uint64_t)q + ((uint64_t)i + 1) * lanes);
        chn.next_rng.idx_ptr = r.idx_ptr;
        st = run_estimate(k, params, r, k->cfg.n_mc, 1, final_out(k, v, gr), &chn);
      }
      if (st == MIVI_OK) flush_chain(k, params, &chn);
      k->cur = 0;
      k->pre_valid = false;
      return st;
    };
    for (int q = 1; q < lanes && s == MIVI_OK && he == hipSuccess; ++q) {
      mivi_ctx *k = c->kids[q - 1];
      char *ko = (char *)c->kid_out[q - 1].p;
      he = hipStreamWaitEvent(k->stream, c->ev_fork, 0);   // the child's stream joins the capture
      if (he != hipSuccess) break;
      s = chain_body(k, q, (count - q + lanes - 1) / lanes, q == q_last ? value : (void *)ko, q == q_last ? grad : (void *)(ko + 16));
      if (s) c->err = k->err;
      if (s == MIVI_OK) he = hipEventRecord(c->ev_join[q - 1], k->stream);
    }
    if (s == MIVI_OK && he == hipSuccess) {
      char *ko = (char *)c->tmp_out.p;
      s = chain_body(c, 0, (count + lanes - 1) / lanes, q_last == 0 ? value : (void *)ko, q_last == 0 ? grad : (void *)(ko + 16));
    }
    for (int q = 1; q < lanes && s == MIVI_OK && he == hipSuccess; ++q) he = hipStreamWaitEvent(c->stream, c->ev_join[q - 1], 0);
    if (s == MIVI_OK && he == hipSuccess) hipLaunchKernelGGL(k_bump_u64, dim3(1), dim3(1), 0, c->stream, (uint64_t *)c->d_idx.p, (uint64_t)count);
    hipError_t e = end_capture(c, saved, &graph);
    if (s) { if (graph) (void)hipGraphDestroy(graph); return s; }
    HIPCHK(c, he);
    HIPCHK(c, e);
    HIPCHK(c, hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    g.kind = 2; g.count = count; g.params = params; g.value = value; g.grad = grad; g.p0 = (double)lanes;
  }
  if (!(c->d_idx_valid && c->d_idx_expect == idx0))
    hipLaunchKernelGGL(k_set_u64x2, dim3(1), dim3(1), 0, c->stream, (uint64_t *)c->d_idx.p, idx0, 0ull, 1);
  HIPCHK(c, hipGraphLaunch(g.exec, c->stream));
  c->d_idx_valid = true;
  c->d_idx_expect = idx0 + (uint64_t)count;
  // (the children's sticky status flags are folded in by mivi_synchronize / read_status)
  return MIVI_OK;
}

