// libmivi C ABI, part 7: per-kernel timing entries (bench.py's roofline block, tools/).
#include "api_common.h"

mivi_status_t mivi_profile_kernel(mivi_ctx_t *c, int32_t which, const void *params, int32_t reps, double *ms_out) {
  if (!c || !params || reps <= 0 || !ms_out || which < 0 || which > 11) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  const int M = c->cfg.n_mc;
  char *o = (char *)c->tmp_out.p;
  OutArgs out = final_out(c, o, o + 16);
  RngArgs rng = rng_of(c, 0);
  c->pre_valid = false;
  mivi_status_t s = run_estimate(c, params, rng, M, 1, out);   // warm + populate eps / W / partial buffers
  if (s) return s;
  if (which != 0) c->pre_valid = false;                        // the stage launches below work on parity 0
  out.M_local = M;
  ValueIn vin{};
  vin.ell_const = c->t_const;
  const bool fr = c->cfg.family == MIVI_FULLRANK;
  c->cur = 0;
  const bool lds = fr && lds_route(c, params, M, 1, out);   // second-generation kernels: stages 2 / 4 include their reduce
  if (which == 6 || which == 7) return fail(c, MIVI_ERR_UNSUPPORTED, "which = 6 / 7: the split-K product / reduce stages were removed (round 3)");
  // which = 10 / 11: the product / VJP launch of FOUR lane-batched estimates, as mivi_estimate_gradient_n issues them.  A batch of eight
  // estimates first (it creates and fills the four contexts), then the four contexts' launches are recorded once and the ONE launch
  // that serves them is replayed.
  LaneSink *psink = nullptr;
  struct SinkGuard {   // (the stage code below returns early on errors)
    LaneSink *&p;
    ~SinkGuard() { if (p) lane_sinks_free(p); }
  } sink_guard{psink};
  if (which == 10 || which == 11) {
    if (!(lds && lds_use_prod32(c, M)) || c->is_child || c->target != TGT_DIAG_GAUSS) return fail(c, MIVI_ERR_UNSUPPORTED, "which = 10 / 11: full-rank second-generation kernels, diagonal-Gaussian target");
    if ((s = mivi_estimate_gradient_n(c, params, 1, 8, o, o + 16))) return s;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!(c->graph.exec && c->graph.kind == 3)) return fail(c, MIVI_ERR_UNSUPPORTED, "which = 10 / 11: this configuration does not take the lane-batched route");
    psink = lane_sinks_alloc(4);
    for (int l = 0; l < 4 && s == MIVI_OK; ++l) {
      mivi_ctx *k = l ? c->kids[l - 1] : c;
      char *ko = l ? (char *)c->kid_out[l - 1].p : o;
      hipStream_t kept = k->stream;
      k->stream = c->stream;
      k->lane_sink = psink; k->lane_id = l;
      lane_sink_reset(psink, l);
      k->cur = 0;
      EpsJob nx{rng_of(k, (uint64_t)(100 + l)), 1};
      launch_lds_prod32(k, params, M, false, R_DIAG, nullptr, &nx, true, false);
      launch_lds_vjp(k, params, M, final_out(k, ko, ko + 16), nullptr, nullptr);
      k->lane_sink = nullptr;
      k->stream = kept;
    }
  }
  if (which == 8 && !(fr && (c->cfg.entropy == MIVI_ENT_STL || c->cfg.entropy == MIVI_ENT_STL_ZERO_GRAD)))
    return fail(c, MIVI_ERR_UNSUPPORTED, "which = 8: full-rank family with a sticking-the-landing estimator");
  if (which == 5) {   // the launch-free loop of 100 estimates (mean-field + diagonal target): one launch per rep
    const bool fn5 = !fr && c->target == TGT_FUNNEL && !c->funnel_constrained;
    if (fr || (c->target != TGT_DIAG_GAUSS && !fn5) || c->bij_on || M > 4096) return fail(c, MIVI_ERR_UNSUPPORTED, "which = 5: mean-field + diagonal-Gaussian / fused funnel target, no bijector");
    if ((s = ensure(c, c->X, ((size_t)100 + 600 * (size_t)((c->cfg.d + 3) / 4) + 16) * sizeof(double) + 100 * ((size_t)c->cfg.d + 2) * c->esize + 64 +
                              32 * 2 * (size_t)c->cfg.d * c->esize + 64 + 100 * (size_t)M * c->esize + 64, false))) return s;   // (+ the estimate lanes' gradient scratch, + the funnel loop's eps[0, m] table)
  } else if (which != 0 && which != 8 && which != 9) {
    if (!fr && which != 2) return fail(c, MIVI_ERR_UNSUPPORTED, "mean-field has a single fused kernel (which = 2)");
    if (which == 4 && c->target != TGT_DENSE_GAUSS) return fail(c, MIVI_ERR_UNSUPPORTED, "no dense target set");
    if (which == 2 && c->target != TGT_DIAG_GAUSS && c->target != TGT_DENSE_GAUSS)
      return fail(c, MIVI_ERR_UNSUPPORTED, "stage timing needs a fused built-in target");
  }
  // Stage launches are captured into ONE graph and replayed: eager back-to-back launches of a 2-5 us kernel are bound by the
  // host's launch rate (3-7 us per launch with these argument blocks), not by the kernel.  which = 0 stays eager (it is what
  // a host-driven loop sees); the graph-batched whole estimate is mivi_estimate_gradient_n.
  auto one = [&](int r) -> mivi_status_t {
    mivi_status_t st = MIVI_OK;
    switch (which) {
      case 0: st = run_estimate(c, params, rng_of(c, (uint64_t)r + 1), M, 1, out); break;
      case 1: launch_eps(c, rng, M); break;
      case 2:
        if (lds && lds_use_prod32(c, M)) {
          launch_lds_prod32(c, params, M, false, c->target == TGT_DENSE_GAUSS ? R_DENSE_R : R_DIAG, nullptr, nullptr, true);
        } else if (lds) {
          launch_lds_prod64(c, params, M, false, c->target == TGT_DENSE_GAUSS ? R_DENSE_R : R_DIAG, nullptr, nullptr, true);
        } else if (fr) launch_fr_sample(c, params, M, c->target, c->target == TGT_DENSE_GAUSS ? c->Z.p : nullptr);
        else launch_mf_main(c, params, rng, M, 1, nullptr, vin, out);
        break;
      case 3:
        if (lds) launch_lds_vjp(c, params, M, out, nullptr, nullptr);
        else launch_fr_vjp(c, params, M, out);
        break;
      case 10: if (!launch_lanes_prod(c, psink, 4, 0)) st = fail(c, MIVI_ERR_HIP, "lane-batched product: the lanes' launches do not match"); break;
      case 11: if (!launch_lanes_vjp(c, psink, 4)) st = fail(c, MIVI_ERR_HIP, "lane-batched VJP: the lanes' launches do not match"); break;
      case 9: {   // two EMPTY dependent launches with the grids / blocks / LDS of the product and VJP kernels: what the launch structure costs
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_empty), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
        const int d32 = (c->cfg.d + 31) / 32, m32 = (M + 31) / 32;
        hipLaunchKernelGGL(k_empty, dim3(d32 * m32), dim3(512), 131 * 1024, c->stream, (int *)nullptr);
        hipLaunchKernelGGL(k_empty, dim3(d32 * (d32 + 1) / 2 + 1), dim3(256), 52 * 1024, c->stream, (int *)nullptr);
        break;
      }
      case 8:   // the STL term W += C^-T eps alone (the parameter-only preparation was left by the warm estimate)
        if (stl2_shape_ok(c, M)) launch_stl2(c, params, M, lds && lds_use_prod32(c, M));
        else launch_fr_stl(c, params, M);
        break;
      case 5:
        if (c->target == TGT_FUNNEL) {
          const size_t d4 = (size_t)((c->cfg.d + 3) / 4);
          double *hist = (double *)c->X.p, *elbo = hist + 600 * d4;
          char *sc = (char *)(elbo + 108);
          char *ls = sc + ((100 * ((size_t)c->cfg.d + 2) * c->esize + 63) & ~(size_t)63);
          launch_mf_funnel_loop(c, params, (uint64_t)r * 100, 100, hist, elbo, (void *)sc, o, o + 16, (void *)ls,
                                (void *)(ls + ((32 * 2 * (size_t)c->cfg.d * c->esize + 63) & ~(size_t)63)));
          break;
        }
        launch_mf_sgd_loop(c, const_cast<void *>(params), nullptr, (uint64_t)r * 100, 0, 100, -1, 0.0, (double)NAN, (double *)c->X.p + 100,
                           (double *)c->X.p, o + 16, (void *)((double *)c->X.p + 100 + 400 * (size_t)((c->cfg.d + 3) / 4) + 8));
        break;
      default:
        if (lds && lds_use_prod32(c, M)) {
          launch_lds_prod32(c, params, M, true, R_DENSE_G, nullptr, nullptr, false);
        } else if (lds) {
          launch_lds_prod64(c, params, M, true, R_DENSE_G, nullptr, nullptr, false);
        } else launch_fr_dense_target(c, M, 1);
        break;
    }
    return st;
  };
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0));
  HIPCHK(c, hipEventCreate(&e1));
  float ms = 0.f;
  const bool graphed = which != 0 && which != 5 && !c->dbg;
  if (graphed) {
    invalidate_graph(c);
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t saved;
    if ((s = begin_capture(c, &saved))) return s;
    for (int r = 0; r < reps && s == MIVI_OK; ++r) s = one(r);
    hipError_t e = end_capture(c, saved, &graph);
    if (s) { if (graph) (void)hipGraphDestroy(graph); return s; }
    HIPCHK(c, e);
    HIPCHK(c, hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    HIPCHK(c, hipGraphLaunch(exec, c->stream));   // warm replay
    HIPCHK(c, hipEventRecord(e0, c->stream));
    HIPCHK(c, hipGraphLaunch(exec, c->stream));
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipEventSynchronize(e1));
    HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    (void)hipGraphExecDestroy(exec);
  } else {
    HIPCHK(c, hipEventRecord(e0, c->stream));
    for (int r = 0; r < reps && s == MIVI_OK; ++r) s = one(r);
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipEventSynchronize(e1));
    HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (which != 0) c->pre_valid = false;
  if (s) return s;
  *ms_out = (double)ms / reps;
  return MIVI_OK;
}

