// libmivi C ABI, part 6: update rules on the device and the device-resident optimisation loop (src/optimize.jl:64-77).
#include "api_common.h"

mivi_status_t mivi_clip_scale(mivi_ctx_t *c, void *params, double epsilon) {
  if (!c || !params) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  launch_clip(c, params, epsilon);
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}
mivi_status_t mivi_prox_scale_entropy(mivi_ctx_t *c, void *params, double stepsize, const void *dog_state, int32_t dog_kind) {
  if (!c || !params || (dog_state && dog_kind != 0 && dog_kind != 1)) return MIVI_ERR_BAD_ARG;
  if (!dog_state && !(stepsize >= 0.0)) return fail(c, MIVI_ERR_BAD_ARG, "proximal step size must be non-negative");
  (void)hipSetDevice(c->cfg.device);
  launch_prox(c, params, stepsize, dog_state, dog_kind);
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}
mivi_status_t mivi_descent_update(mivi_ctx_t *c, void *params, const void *grad, double eta) {
  if (!c || !params || !grad) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  launch_descent(c, params, grad, eta);
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}
mivi_status_t mivi_adam_update(mivi_ctx_t *c, void *params, const void *grad, void *state, int64_t t, double eta,
                               double b1, double b2, double eps) {
  if (!c || !params || !grad || !state || t < 1) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  launch_adam(c, params, grad, state, nullptr, t, eta, b1, b2, eps);
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}

mivi_status_t mivi_cocob_update(mivi_ctx_t *c, void *params, const void *grad, void *state, double alpha) {
  if (!c || !params || !grad || !state || !(alpha > 0.0)) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  launch_cocob(c, params, grad, state, alpha);
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}

mivi_status_t mivi_optimize_steps(mivi_ctx_t *c, void *params, void *opt_state, uint64_t idx0, int64_t t0, int32_t n_steps,
                                  int32_t rule, double eta, double clip_eps, void *elbo) {
  if (rule != 0 && rule != 1) return MIVI_ERR_BAD_ARG;
  mivi_loop_t l{};
  l.rule = rule;
  l.op = clip_eps > 0.0 ? 1 : 0;
  l.averager = 0;
  l.n_steps = n_steps;
  l.eta = eta;
  l.beta1 = 0.9;
  l.beta2 = 0.999;
  l.adam_eps = 1e-8;
  l.clip_epsilon = clip_eps;
  l.opt_state_dev = opt_state;
  l.estimate_idx0 = idx0;
  l.t0 = t0;
  l.elbo_dev = elbo;
  return mivi_optimize_loop(c, params, &l);
}

// elbo record (double, device) -> caller's T[n_steps], on the device (no host round trip inside a "launch-free" call)
__global__ void k_elbo_to_f32(int n, const double *__restrict__ rec, float *__restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (float)rec[i];
}
static mivi_status_t deliver_elbo(mivi_ctx *c, const double *rec, int n_steps, void *elbo) {
  if (!elbo) return MIVI_OK;
  if (c->cfg.dtype == MIVI_F64) {
    HIPCHK(c, hipMemcpyAsync(elbo, rec, (size_t)n_steps * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  } else {
    hipLaunchKernelGGL(k_elbo_to_f32, dim3((n_steps + 255) / 256), dim3(256), 0, c->stream, n_steps, rec, (float *)elbo);
    HIPCHK(c, hipGetLastError());
  }
  return MIVI_OK;
}

static bool same_loop(const mivi_loop_t &a, const mivi_loop_t &b) {   // everything baked into a captured loop
  return a.rule == b.rule && a.op == b.op && a.averager == b.averager && a.n_steps == b.n_steps && a.eta == b.eta &&
         a.beta1 == b.beta1 && a.beta2 == b.beta2 && a.adam_eps == b.adam_eps && a.clip_epsilon == b.clip_epsilon &&
         a.avg_eta == b.avg_eta && a.opt_state_dev == b.opt_state_dev && a.avg_params_dev == b.avg_params_dev;
}

// One call's steps on the best route.  allow_exchange = false: the launch-free loops whose workgroups exchange partials grid-wide every step are
// skipped (their graph-of-launches equivalents run instead); *used_exchange: one of them was launched.
static mivi_status_t optimize_loop_run(mivi_ctx_t *c, void *params, const mivi_loop_t *lp, bool allow_exchange, bool *used_exchange) {
  const mivi_loop_t &l = *lp;
  const int n_steps = l.n_steps, rule = l.rule;
  if (n_steps <= 0 || rule < 0 || rule > 4 || l.op < 0 || l.op > 2 || l.averager < 0 || l.averager > 1) return MIVI_ERR_BAD_ARG;
  if (rule != 0 && !l.opt_state_dev) return fail(c, MIVI_ERR_BAD_ARG, "this optimisation rule needs opt_state_dev");
  if (l.averager == 1 && !l.avg_params_dev) return fail(c, MIVI_ERR_BAD_ARG, "PolynomialAveraging needs avg_params_dev");
  if (l.op == 2 && (rule == 1 || rule == 4)) return fail(c, MIVI_ERR_BAD_ARG, "ProximalLocationScaleEntropy does not support Adam / COCOB (Descent, DoG, DoWG: proximal_location_scale_entropy.jl:26-42)");
  const bool dog = rule == 2 || rule == 3;        // the rules with two global norms per step
  const bool loops_know = rule <= 3;              // the launch-free loops implement rules 0..3; COCOB (4) runs on the graph of launches
  if (!graph_capturable(c)) return fail(c, MIVI_ERR_UNSUPPORTED, "device-resident loop needs a built-in target");
  if (c->idx_src) return fail(c, MIVI_ERR_UNSUPPORTED, "an index source is set (mivi_set_index_source): the device-resident loop keeps its own counter");
  (void)hipSetDevice(c->cfg.device);
  mivi_status_t s = ensure_work(c, c->cfg.n_mc);
  if (s) return s;
  prepare_tables(c, c->cfg.n_mc);
  if ((s = reserve_target(c, c->cfg.n_mc))) return s;
  const size_t plen = (size_t)mivi_params_len(c), es = c->esize;
  const double eta = l.eta, clip_eps = (l.op == 1) ? l.clip_epsilon : (double)NAN;   // NaN = no ClipScale
  void *opt_state = l.opt_state_dev;
  // internal value/grad/elbo-record buffers
  size_t hist_doubles = (size_t)n_steps * 4 * (size_t)((c->cfg.d + 3) / 4);
  {   // (the funnel loop: six partials per row quad and step, every step's set at addresses of its own, 128-byte aligned)
    const size_t fh = (size_t)n_steps * ((6 * (size_t)((c->cfg.d + 3) / 4) + 15) / 16 * 16) + 16 + (size_t)((c->cfg.d + 3) / 4) / 2 + 8;   // (+ the arrival flags, the published row)
    if (hist_doubles < fh) hist_doubles = fh;
  }
  if ((s = ensure(c, c->X, (plen + 8) * es + ((size_t)n_steps + hist_doubles + 8) * sizeof(double), false))) return s;
  char *vbuf = (char *)c->X.p;
  char *gbuf = vbuf + 8 * es;
  double *rec = (double *)(((uintptr_t)(gbuf + plen * es) + 7) & ~(uintptr_t)7);
  static const bool no_fused_loop = getenv("MIVI_NO_FUSED_LOOP") != nullptr;
  const bool simple = rule <= 1 && l.op <= 1 && l.averager == 0;   // what the fused paths implement
  const bool default_adam = l.beta1 == 0.9 && l.beta2 == 0.999 && l.adam_eps == 1e-8;
  if (simple && (rule == 0 || default_adam) && c->cfg.family == MIVI_MEANFIELD && c->target == TGT_DIAG_GAUSS && !c->bij_on &&
      c->cfg.n_mc <= 4096 && !no_fused_loop) {   // (the launch-free kernel has no Stacked-bijector handling: explicit-sample route)
    // launch-free loop: every workgroup owns four rows of (mu, sigma); no graph, two launches for all n_steps
    HIPCHK(c, hipMemsetAsync(c->status.p, 0, sizeof(int) * (1 + mivi_ctx::kMaxKids), c->stream));   // (every word read_status folds in: a stale flag of an
    launch_mf_sgd_loop(c, params, opt_state, l.estimate_idx0, (long long)l.t0, n_steps, rule, eta, clip_eps, rec + n_steps, rec);   //  earlier batch's child contexts is not this run's)
    HIPCHK(c, hipGetLastError());
    if ((s = deliver_elbo(c, rec, n_steps, l.elbo_dev))) return s;
    return read_status(c);
  }
  if (loops_know && lr_small_loop_ok(c) && !no_fused_loop && allow_exchange) {
    // small hierarchical logistic regressions (the reference README's own example, BASELINE configs[0]): the whole loop in ONE workgroup, every
    // rule x operator x averager (k_lr_small_loop); larger ones: up to 64 workgroups that exchange partial sums every step
    if (lr_small_part_bytes(c, n_steps) && (s = ensure(c, c->gen_scratch, lr_small_part_bytes(c, n_steps) + 256, false))) return s;
    HIPCHK(c, hipMemsetAsync(c->status.p, 0, sizeof(int) * (1 + mivi_ctx::kMaxKids), c->stream));
    if (launch_lr_small_loop(c, params, l, rec, vbuf, (double *)c->gen_scratch.p)) {
      *used_exchange = true;
      HIPCHK(c, hipGetLastError());
      if ((s = deliver_elbo(c, rec, n_steps, l.elbo_dev))) return s;
      return read_status(c);
    }
  }
  const bool general = !simple || (rule == 1 && !default_adam);   // (Adam with other betas than the fused paths' defaults: the general loops take them from the call)
  if (general && loops_know && mf_gen_loop_ok(c, rule) && !no_fused_loop && (rule < 2 || allow_exchange)) {
    // every other rule x operator x averager of the reference's algorithms (DoG / DoWG, ProximalLocationScaleEntropy, PolynomialAveraging -- its
    // defaults), mean-field + diagonal-Gaussian target: launch-free as well (k_mf_gen_loop; DoG / DoWG: one grid-wide exchange of two norms per step)
    if ((s = ensure(c, c->gen_scratch, mf_gen_loop_scratch_bytes(c, n_steps), false))) return s;
    HIPCHK(c, hipMemsetAsync(c->status.p, 0, sizeof(int) * (1 + mivi_ctx::kMaxKids), c->stream));
    if (launch_mf_gen_loop(c, params, l, rec + n_steps, rec, (char *)c->gen_scratch.p)) {
      if (rule >= 2) *used_exchange = true;
      HIPCHK(c, hipGetLastError());
      if ((s = deliver_elbo(c, rec, n_steps, l.elbo_dev))) return s;
      return read_status(c);
    }
  }
  if (general && loops_know && fr_small_loop_ok(c) && !no_fused_loop) {
    // ... and small full-rank problems in one workgroup (k_fr_small_loop: the two norms of DoG / DoWG are block sums there)
    HIPCHK(c, hipMemsetAsync(c->status.p, 0, sizeof(int) * (1 + mivi_ctx::kMaxKids), c->stream));
    launch_fr_small_loop(c, params, opt_state, l.estimate_idx0, (long long)l.t0, n_steps, rule, eta, clip_eps, rec, vbuf, &l);
    HIPCHK(c, hipGetLastError());
    if ((s = deliver_elbo(c, rec, n_steps, l.elbo_dev))) return s;
    return read_status(c);
  }
  if (general && loops_know && fr_rows_loop_ok(c) && (rule < 2 || allow_exchange) && !no_fused_loop && fr_rows_eps_bytes(c, n_steps) <= ((size_t)1 << 31)) {
    // ... and on the full-rank family with few samples per step (the reference's default n_samples = 1): the row-owning workgroups of
    // k_fr_rows_loop, DoG / DoWG with the same per-step exchange of two norm partials (every workgroup resident: checked by the launcher)
    if ((s = ensure(c, c->rows_eps, fr_rows_eps_bytes(c, n_steps), false))) return s;
    if (rule >= 2 && (s = ensure(c, c->gen_scratch, fr_rows_part_bytes(c, n_steps) + 256, false))) return s;
    HIPCHK(c, hipMemsetAsync(c->status.p, 0, sizeof(int) * (1 + mivi_ctx::kMaxKids), c->stream));
    if (launch_fr_rows_loop(c, params, opt_state, l.estimate_idx0, (long long)l.t0, n_steps, rule, eta, clip_eps, (float *)c->rows_eps.p, rec + n_steps, rec, vbuf, &l,
                            (double *)c->gen_scratch.p)) {
      if (rule >= 2) *used_exchange = true;
      HIPCHK(c, hipGetLastError());
      if ((s = deliver_elbo(c, rec, n_steps, l.elbo_dev))) return s;
      return read_status(c);
    }
  }
  if (simple && (rule == 0 || default_adam) && fr_small_loop_ok(c) && !no_fused_loop) {
    // small full-rank problems (the reference's own benchmark grid: d = 10, one sample per step): the whole loop in ONE workgroup
    HIPCHK(c, hipMemsetAsync(c->status.p, 0, sizeof(int) * (1 + mivi_ctx::kMaxKids), c->stream));
    launch_fr_small_loop(c, params, opt_state, l.estimate_idx0, (long long)l.t0, n_steps, rule, eta, clip_eps, rec, vbuf);
    HIPCHK(c, hipGetLastError());
    if ((s = deliver_elbo(c, rec, n_steps, l.elbo_dev))) return s;
    return read_status(c);
  }
  if (simple && (rule == 0 || default_adam) && fr_rows_loop_ok(c) && !no_fused_loop && fr_rows_eps_bytes(c, n_steps) <= ((size_t)1 << 31)) {
    // full-rank family, few samples per step, elementwise target: the rows are independent -- every workgroup keeps its rows' parameters and
    // optimiser state in registers for all n_steps (k_fr_rows_loop); eps of the whole call is drawn up front
    if ((s = ensure(c, c->rows_eps, fr_rows_eps_bytes(c, n_steps), false))) return s;
    HIPCHK(c, hipMemsetAsync(c->status.p, 0, sizeof(int) * (1 + mivi_ctx::kMaxKids), c->stream));
    if (launch_fr_rows_loop(c, params, opt_state, l.estimate_idx0, (long long)l.t0, n_steps, rule, eta, clip_eps, (float *)c->rows_eps.p, rec + n_steps, rec, vbuf)) {
      HIPCHK(c, hipGetLastError());
      if ((s = deliver_elbo(c, rec, n_steps, l.elbo_dev))) return s;
      return read_status(c);
    }
  }
  if (simple && (rule == 0 || default_adam) && c->cfg.family == MIVI_MEANFIELD && c->target == TGT_FUNNEL && !c->funnel_constrained && !c->bij_on &&
      c->cfg.n_mc <= 256 && c->cfg.d <= 16384 && !no_fused_loop && allow_exchange) {
    // launch-free loop for the fused funnel target: the row quads and the row-0 workgroup of ONE kernel exchange two scalars per workgroup and
    // row 0's parameters per step (k_mf_funnel_sgd_loop) instead of three launches per step
    double *hist = (double *)(((uintptr_t)(rec + n_steps) + 127) & ~(uintptr_t)127);
    const size_t nq = (size_t)((c->cfg.d + 3) / 4);
    void *pub = (void *)(rec + n_steps + hist_doubles - (nq / 2 + 8));
    unsigned *sync = (unsigned *)((double *)pub + 4);
    HIPCHK(c, hipMemsetAsync(c->status.p, 0, sizeof(int) * (1 + mivi_ctx::kMaxKids), c->stream));
    if (launch_mf_funnel_sgd_loop(c, params, opt_state, l.estimate_idx0, (long long)l.t0, n_steps, rule, eta, clip_eps, hist, sync, pub, gbuf, rec, vbuf)) {
      *used_exchange = true;
      HIPCHK(c, hipGetLastError());
      if ((s = deliver_elbo(c, rec, n_steps, l.elbo_dev))) return s;
      return read_status(c);
    }
  }
  GraphCache &g = c->graph;
  if (!(g.exec && g.kind == 9 && g.params == params && g.value == (void *)vbuf && same_loop(g.loop, l))) {
    invalidate_graph(c);
    hipGraph_t graph = nullptr;
    hipStream_t saved;
    if ((s = begin_capture(c, &saved))) return s;
    Chain chn;
    chn.on = true;
    const long long *t_ptr = (const long long *)c->d_idx.p + 1;   // iterations done before this call
    for (int i = 0; i < n_steps && s == MIVI_OK; ++i) {
      RngArgs r = rng_of(c, (uint64_t)i);
      r.idx_ptr = (const uint64_t *)c->d_idx.p;
      OutArgs o = final_out(c, vbuf, gbuf);
      o.elbo_rec = rec;
      o.rec_slot = i;
      c->cur = i & 1;
      chn.has_next = (i + 1 < n_steps);
      chn.next_rng = rng_of(c, (uint64_t)i + 1);
      chn.next_rng.idx_ptr = r.idx_ptr;
      // full-rank f32 MFMA path: the optimiser step (and ClipScale) rides in the VJP epilogue -- no update kernel
      const bool fuse_upd = simple && c->cfg.family == MIVI_FULLRANK && c->cfg.dtype == MIVI_F32 && hetero_ok(c, 1) &&
                            !no_fused_update();
      FusedUpdate fu;
      if (fuse_upd) {
        fu.rule = rule;
        fu.params = params;
        fu.state = opt_state;
        fu.t_ptr = t_ptr;
        fu.t_base = (long long)i + 1;
        fu.eta = eta;
        fu.b1 = l.beta1;
        fu.b2 = l.beta2;
        fu.eps = l.adam_eps;
        fu.clip_eps = l.clip_epsilon;
        fu.do_clip = (l.op == 1);
      }
      s = run_estimate(c, params, r, c->cfg.n_mc, 1, o, &chn, fuse_upd ? &fu : nullptr);
      if (s) break;
      if (fuse_upd) continue;
      // Optimisers.update! (common.jl:92); ClipScale rides in the Descent / Adam kernels
      if (rule == 0) launch_descent(c, params, gbuf, eta, clip_eps);
      else if (rule == 1) launch_adam(c, params, gbuf, opt_state, (const int64_t *)t_ptr, (int64_t)i + 1, eta, l.beta1, l.beta2, l.adam_eps, clip_eps);
      else if (rule == 4) launch_cocob(c, params, gbuf, opt_state, eta /* = alpha */, clip_eps);   // rules.jl:78-96; ClipScale rides in the kernel
      else if (l.op <= 1 && launch_dog_update_fused(c, params, gbuf, opt_state, rule - 2, clip_eps,
                                                     l.averager == 1 ? l.avg_params_dev : nullptr, l.avg_eta, t_ptr, (long long)i + 1))
        continue;   // DoG / DoWG + ClipScale + averaging in one apply pass (large parameter vectors)
      else launch_dog_update(c, params, gbuf, opt_state, rule - 2);
      // operator (common.jl:93-95)
      if (l.op == 1 && dog) launch_clip(c, params, clip_eps);
      if (l.op == 2) launch_prox(c, params, eta, dog ? opt_state : nullptr, rule - 2);
      // averager (common.jl:96)
      if (l.averager == 1) launch_poly_average(c, l.avg_params_dev, params, l.avg_eta, t_ptr, (long long)i + 1);
    }
    if (s == MIVI_OK) flush_chain(c, params, &chn);
    c->cur = 0;
    hipError_t e = end_capture(c, saved, &graph);
    if (s) { if (graph) (void)hipGraphDestroy(graph); return s; }
    HIPCHK(c, e);
    HIPCHK(c, hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    g.kind = 9; g.count = n_steps; g.params = params; g.value = vbuf;
    g.loop = l;
  }
  c->d_idx_valid = false;
  hipLaunchKernelGGL(k_set_u64x2, dim3(1), dim3(1), 0, c->stream, (uint64_t *)c->d_idx.p, l.estimate_idx0, (uint64_t)l.t0, 2);
  HIPCHK(c, hipMemsetAsync(c->status.p, 0, sizeof(int) * (1 + mivi_ctx::kMaxKids), c->stream));   // a stale flag of earlier estimates (this context's word or a child context's: read_status folds them all in) is not this run's
  HIPCHK(c, hipGraphLaunch(g.exec, c->stream));
  if ((s = deliver_elbo(c, rec, n_steps, l.elbo_dev))) return s;
  return read_status(c);
}

// `optimize`'s inner loop (src/optimize.jl:64-77) on the device.  The launch-free loops whose workgroups exchange partials every step need their
// whole grid resident: the launchers check that against the device (hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs), and should an exchange
// still be lost at run time (status bit 8: another context's work took the CUs), the call restores the state it was given and runs the same
// steps on the graph of launches; the context then stays on that route.
mivi_status_t mivi_optimize_loop(mivi_ctx_t *c, void *params, const mivi_loop_t *lp) {
  if (!c || !params || !lp) return MIVI_ERR_BAD_ARG;
  const mivi_loop_t &l = *lp;
  (void)hipSetDevice(c->cfg.device);
  const size_t plen = (size_t)mivi_params_len(c), es = c->esize;
  const size_t st_bytes = !l.opt_state_dev ? 0 : (l.rule == 1 ? 2 * plen * es : (l.rule == 4 ? 5 * plen * es : (l.rule >= 2 ? (size_t)mivi_dog_state_bytes(c) : 0)));
  const size_t avg_bytes = (l.averager == 1 && l.avg_params_dev) ? plen * es : 0;
  const bool may_exchange = !c->exchange_lost && (l.rule == 2 || l.rule == 3 || c->target == TGT_FUNNEL || c->target == TGT_LOGREG);
  if (may_exchange && l.n_steps > 0 && l.rule >= 0 && l.rule <= 4) {
    mivi_status_t s = ensure(c, c->snap, plen * es + st_bytes + avg_bytes + 64, false);
    if (s) return s;
    char *sp = (char *)c->snap.p;
    HIPCHK(c, hipMemcpyAsync(sp, params, plen * es, hipMemcpyDeviceToDevice, c->stream));
    if (st_bytes) HIPCHK(c, hipMemcpyAsync(sp + plen * es, l.opt_state_dev, st_bytes, hipMemcpyDeviceToDevice, c->stream));
    if (avg_bytes) HIPCHK(c, hipMemcpyAsync(sp + plen * es + st_bytes, l.avg_params_dev, avg_bytes, hipMemcpyDeviceToDevice, c->stream));
  }
  bool used = false;
  c->last_status_bits = 0;   // only a bit-8 ("exchange lost") flag read by THIS call may trigger the retry below: a stale one would mask a real HIP error
  mivi_status_t s = optimize_loop_run(c, params, lp, may_exchange, &used);
  static const bool force_lost = getenv("MIVI_FORCE_EXCHANGE_LOST") != nullptr;   // test hook (tests/test_gpu_loop_oracle.py): treat the first exchanging call as lost
  if (force_lost && used && s == MIVI_OK) { s = MIVI_ERR_HIP; c->last_status_bits |= 8; }
  if (s == MIVI_ERR_HIP && used && (c->last_status_bits & 8)) {
    const char *sp = (const char *)c->snap.p;
    HIPCHK(c, hipMemcpyAsync(params, sp, plen * es, hipMemcpyDeviceToDevice, c->stream));
    if (st_bytes) HIPCHK(c, hipMemcpyAsync(l.opt_state_dev, sp + plen * es, st_bytes, hipMemcpyDeviceToDevice, c->stream));
    if (avg_bytes) HIPCHK(c, hipMemcpyAsync(l.avg_params_dev, sp + plen * es + st_bytes, avg_bytes, hipMemcpyDeviceToDevice, c->stream));
    c->exchange_lost = true;
    c->err.clear();
    used = false;
    s = optimize_loop_run(c, params, lp, false, &used);
  }
  return s;
}
