// libmivi C ABI, part 2: the estimate driver -- one RepGradELBO estimate as a sequence of kernel launches
// (estimate_gradient!, src/algorithms/repgradelbo.jl:151-177), mivi_sample, partials / finalisation.
#include "api_common.h"

// ---------------------------------------------------------------------------------------------
// estimate driver
// ---------------------------------------------------------------------------------------------
static mivi_status_t eval_generic_target(mivi_ctx *c, int M, int want_grad) {
  switch (c->target) {
    case TGT_DIAG_GAUSS:
    case TGT_FUNNEL:
      launch_col_target(c, M, want_grad);
      return MIVI_OK;
    case TGT_LOGREG:
      if (!launch_logreg_target(c, M, want_grad)) return fail(c, MIVI_ERR_HIP, "logistic regression: scratch allocation failed");
      return MIVI_OK;
    case TGT_CALLBACK: {
      const size_t es = c->esize, d = c->cfg.d;
      c->h_Z.resize(d * M * es);
      c->h_G.resize(d * M * es);
      c->h_ell.resize((size_t)M * es);
      HIPCHK(c, hipMemcpyAsync(c->h_Z.data(), c->Z.p, d * M * es, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      int rc;
      if (!want_grad && c->cb_value)
        rc = c->cb_value(c->cb_user, c->h_Z.data(), (int)d, M, c->h_ell.data());
      else
        rc = c->cb_grad(c->cb_user, c->h_Z.data(), (int)d, M, c->h_ell.data(), c->h_G.data());
      if (rc != 0) return fail(c, MIVI_ERR_BAD_ARG, "target callback returned non-zero");
      HIPCHK(c, hipMemcpyAsync(c->ell.p, c->h_ell.data(), (size_t)M * es, hipMemcpyHostToDevice, c->stream));
      if (want_grad) HIPCHK(c, hipMemcpyAsync(c->W.p, c->h_G.data(), d * M * es, hipMemcpyHostToDevice, c->stream));
      return MIVI_OK;
    }
    default:
      return fail(c, MIVI_ERR_NO_TARGET, "no target set");
  }
}

bool no_fused_update() {   // MIVI_NO_FUSED_UPDATE=1: separate update kernel in the graph loop (A/B reference)
  static const bool v = getenv("MIVI_NO_FUSED_UPDATE") != nullptr;
  return v;
}

bool hetero_ok(const mivi_ctx *c, int want_grad, const Chain *ch) {
  if (!want_grad || c->bij_on) return false;   // (a Stacked bijector runs on the explicit-sample route)
  // (the fused funnel target's value workgroup also finishes two gradient entries, which an optimiser step right after the
  //  estimate must already see: chained only when nothing reads the gradient between the estimates)
  if (c->cfg.family == MIVI_MEANFIELD)
    return c->target == TGT_DIAG_GAUSS || (c->target == TGT_FUNNEL && !c->funnel_constrained && ch && ch->estimates_only);
  if (c->cfg.dtype != MIVI_F32 && f64_valu()) return false;
  return c->target == TGT_DIAG_GAUSS || c->target == TGT_DENSE_GAUSS;
}

// Second-generation full-rank route (kernels_fullrank_lds.hip): f32, d and M multiples of 64, fused Gaussian targets,
// 16-byte aligned parameter / gradient vectors.  Everything else (and MIVI_FR_GEN1=1) takes the first-generation kernels.
bool lds_route(const mivi_ctx *c, const void *params, int M, int want_grad, const OutArgs &out) {
  if (!lds_path_shape_ok(c, M) || c->bij_on) return false;
  if (c->target != TGT_DIAG_GAUSS && c->target != TGT_DENSE_GAUSS) return false;
  if ((uintptr_t)params & 15) return false;
  if (want_grad && !out.partials_mode && ((uintptr_t)out.grad & 15)) return false;
  return true;
}

// One estimate on the second-generation route:
//   [k_eps unless the previous estimate's product kernel already drew this eps]
//   k_fr_prod32 / k_fr_prod64 <SAMPLE> (z, fused target, ell / log-det partials, riders: eps of the next estimate, STL operands)
//   [dense target: the same kernel <DENSE>]  [STL: back substitution]  -> k_fr_vjp32 / k_fr_vjp64 (+ this estimate's value)
static mivi_status_t run_estimate_lds(mivi_ctx *c, const void *params, const RngArgs &rng, int M, int want_grad, OutArgs out,
                                      Chain *ch, const FusedUpdate *upd, bool stop_after_target) {
  if (!lds_prepare(c, M)) return fail(c, MIVI_ERR_HIP, "full-rank work lists: allocation failed");
  ValueIn vin{};
  vin.ell_const = c->t_const;
  const bool grad_stage = want_grad && !stop_after_target;
  const bool chained = ch && ch->on && grad_stage && !out.partials_mode;
  const bool spec = !chained && want_grad;   // (also the Stein estimator's calls: stop_after_target)
  bool hit = false;
  int capturing = 0;
  unsigned long long cap_id = 0;
  if (spec) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamGetCaptureInfo(c->stream, &cs, &cap_id) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
    capturing = cs == hipStreamCaptureStatusActive;
    if (!capturing) cap_id = 0;
    hit = c->pre_valid && c->pre_M == M && c->pre_rng.seed == rng.seed && c->pre_rng.idx_base == rng.idx_base &&
          c->pre_rng.idx_ptr == rng.idx_ptr && c->pre_rng.m_offset == rng.m_offset && c->pre_capturing == capturing &&
          c->pre_capture_id == cap_id;
  }
  c->pre_valid = false;
  if (!chained) c->cur = hit ? c->pre_parity : 0;
  const int p = c->cur;
  if (chained ? ch->first : !hit) {
    c->he_n[p] = launch_eps(c, rng, M);
  }
  vin.he_part = (const double *)c->he_part[p].p;
  vin.n_he_part = c->he_n[p];
  EpsJob nx{};
  const EpsJob *next = nullptr;
  if (chained && ch->has_next) {
    nx.rng = ch->next_rng;
    nx.parity = p ^ 1;
    next = &nx;
  } else if (spec) {   // speculate that the caller asks for estimate idx + 1 next (an SGD loop does)
    nx.rng = rng;
    nx.rng.idx_base = rng.idx_base + 1ull;   // (a single call: the NEXT index, whatever stride an earlier batched call left on this context)
    nx.parity = p ^ 1;
    next = &nx;
  }
  const bool dense = c->target == TGT_DENSE_GAUSS;
  const bool p32 = lds_use_prod32(c, M);
  bool dinv_done = false;
  if (p32) {   // unsplit 32 x 32 tiles with the target fused into the epilogue: one kernel from eps to W
    const bool stl_here = (grad_stage && (out.ent_kind == MIVI_ENT_STL || out.ent_kind == MIVI_ENT_STL_ZERO_GRAD) && stl2_shape_ok(c, M)) ||
                          (c->want_stl_pack && c->stl_F.p);   // (the Stein estimator's solve: the riders prepare its operands too)
    // (a chain of estimates with no optimiser step in between reads the SAME parameters: the solve's parameter-only preparation of
    //  the chain's first estimate stays valid, the later ones carry no STL riders)
    const bool reuse_pack = stl_here && chained && ch->estimates_only && !ch->first;
    launch_lds_prod32(c, params, M, false, dense ? R_DENSE_R : R_DIAG, nullptr, next, grad_stage, stl_here && !reuse_pack);
    dinv_done = stl_here;
    c->stl_pack_done = stl_here;
    if (next) c->he_n[p ^ 1] = lds_prod32_eps_blocks(c, M);
    if (dense) launch_lds_prod32(c, params, M, true, R_DENSE_G, nullptr, nullptr, false);
    vin.ell_part = (const double *)c->ell_part[p].p;
    vin.n_ell_part = lds_prod32_tiles(c, M);
  } else {   // large shapes: unsplit 64 x 64 tiles, the target fused into the epilogue
    launch_lds_prod64(c, params, M, false, dense ? R_DENSE_R : R_DIAG, nullptr, next, grad_stage);
    if (next) c->he_n[p ^ 1] = lds_eps_blocks(c, M);
    if (dense) launch_lds_prod64(c, params, M, true, R_DENSE_G, nullptr, nullptr, false);
    vin.ell_part = (const double *)c->ell_part[p].p;
    vin.n_ell_part = lds_prod64_tiles(c, M);
  }
  if (ch) { ch->have_prev = false; ch->first = !chained; }
  if (grad_stage) {
    if (out.ent_kind == MIVI_ENT_STL || out.ent_kind == MIVI_ENT_STL_ZERO_GRAD) {
      const size_t sh = (8 * (size_t)c->dP + 32 * 33) * c->esize;
      if (sh > 160 * 1024 && !c->stl_CT.p) return fail(c, MIVI_ERR_UNSUPPORTED, "full-rank STL: d too large for the LDS-resident solve");
      if (stl2_shape_ok(c, M)) launch_stl2(c, params, M, dinv_done);
      else launch_fr_stl(c, params, M);
    }
    vin.ld_part = (const double *)c->ld_part[p].p;   // left by the reduce kernel (the VJP kernel may already be updating C)
    vin.n_ld_part = p32 ? fr_ld_blocks(c) : lds_ld_blocks(c);
    ValueJob self{vin, out};
    launch_lds_vjp(c, params, M, out, &self, chained ? upd : nullptr);
    if (spec) {
      c->pre_valid = true;
      c->pre_rng = nx.rng;
      c->pre_M = M;
      c->pre_parity = p ^ 1;
      c->pre_capturing = capturing;
      c->pre_capture_id = cap_id;
    }
    HIPCHK(c, hipGetLastError());
    return MIVI_OK;
  }
  if (spec) {
    c->pre_valid = true;
    c->pre_rng = nx.rng;
    c->pre_M = M;
    c->pre_parity = p ^ 1;
    c->pre_capturing = capturing;
    c->pre_capture_id = cap_id;
  }
  if (c->defer_value) {   // (Stein estimator: its accumulation kernel assembles the value partials in one of its own workgroups)
    *c->defer_value = ValueJob{vin, out};
    c->value_deferred = true;
  } else {
    launch_value_only(c, params, vin, out);
  }
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}

// One estimate over M local samples. out.partials_mode selects final vs shard partials.
mivi_status_t run_estimate(mivi_ctx *c, const void *params, const RngArgs &rng, int M, int want_grad,
                                  OutArgs out, Chain *ch, const FusedUpdate *upd,
                                  bool stop_after_target) {
  if (c->target == TGT_NONE) return fail(c, MIVI_ERR_NO_TARGET, "no target set");
  mivi_status_t s = ensure_work(c, M);
  if (s) return s;
  out.M_local = M;
  if (!out.status) out.status = (int *)c->status.p;
  if (c->cfg.family == MIVI_FULLRANK && lds_route(c, params, M, want_grad, out))
    return run_estimate_lds(c, params, rng, M, want_grad, out, ch, upd, stop_after_target);
  ValueIn vin{};
  vin.ell_const = c->t_const;
  const int d = c->cfg.d, d4 = (d + 3) / 4;
  const bool chained = ch && ch->on && hetero_ok(c, want_grad, ch) && !out.partials_mode;
  // single calls on the MFMA full-rank path: did the previous call's VJP kernel already generate this estimate's eps?
  const bool spec = !chained && c->cfg.family == MIVI_FULLRANK && hetero_ok(c, want_grad) && !stop_after_target;
  bool hit = false;
  int capturing = 0;
  unsigned long long cap_id = 0;
  if (spec) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamGetCaptureInfo(c->stream, &cs, &cap_id) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
    capturing = cs == hipStreamCaptureStatusActive;
    if (!capturing) cap_id = 0;
    hit = c->pre_valid && c->pre_M == M && c->pre_rng.seed == rng.seed && c->pre_rng.idx_base == rng.idx_base &&
          c->pre_rng.idx_ptr == rng.idx_ptr && c->pre_rng.m_offset == rng.m_offset && c->pre_capturing == capturing &&
          c->pre_capture_id == cap_id;
  }
  c->pre_valid = false;
  if (!chained) c->cur = hit ? c->pre_parity : 0;
  const int p = c->cur;
  const ValueJob *prev = (chained && ch->have_prev) ? &ch->prev : nullptr;

  if (c->cfg.family == MIVI_MEANFIELD) {
    const bool bij = c->bij_on;
    if (!bij && (c->target == TGT_DIAG_GAUSS || (c->target == TGT_FUNNEL && want_grad && !c->funnel_constrained))) {
      launch_mf_main(c, params, rng, M, want_grad, nullptr, vin, out, prev);
      if (c->target == TGT_FUNNEL) {   // row 0 and ell are finished by whoever assembles the value (FunnelFin)
        vin.fn.ab = (const double *)c->sc_part[p].p + 4 * (size_t)c->mf_nblk;
        vin.fn.n_part = c->mf_nblk;
        vin.fn.params = params;
        vin.fn.rng = rng;
        vin.fn.d4 = d4;
        vin.fn.M = M;
        vin.fn.sigma_v = c->funnel_sigma_v;
      }
      vin.ell_part2 = (const double *)c->sc_part[p].p;
      vin.n_ell_part2 = c->mf_nblk;
      vin.he_part = (const double *)c->sc_part[p].p + c->mf_nblk;
      vin.n_he_part = c->mf_nblk;
      vin.ld_part = (const double *)c->sc_part[p].p + 2 * (size_t)c->mf_nblk;
      vin.n_ld_part = c->mf_nblk;
    } else {
      launch_sample_mf(c, params, rng, M, c->Z.p, nullptr, 0, want_grad ? nullptr : (double *)c->he_part[p].p);
      if (bij) launch_bij_forward(c, M);   // the target sees binv(z)
      if (c->target == TGT_DENSE_GAUSS) {
        launch_rt_from_z(c, M);
        launch_fr_dense_target(c, M, want_grad);
        vin.ell_part = (const double *)c->ell_part[p].p;
        vin.n_ell_part = fr_dense_blocks(c, M);
        if (bij) { vin.ell = c->bij_ld.p; vin.n_ell = M; }   // + logabsdetjac per sample
      } else {
        if (logreg_uses_mfma(c, M)) launch_rt_from_z(c, M);   // Z^T for the MFMA route
        if ((s = eval_generic_target(c, M, want_grad))) return s;
        vin.ell = c->ell.p;
        vin.n_ell = M;
      }
      if (bij) launch_bij_backward(c, M, want_grad, c->target != TGT_DENSE_GAUSS);
      if (want_grad) {
        launch_mf_main(c, params, rng, M, 1, c->W.p, vin, out);
        vin.ell_part2 = (const double *)c->sc_part[p].p;   // zeros for the non-fused target; he / logdet live here
        vin.n_ell_part2 = c->mf_nblk;
        vin.he_part = (const double *)c->sc_part[p].p + c->mf_nblk;
        vin.n_he_part = c->mf_nblk;
        vin.ld_part = (const double *)c->sc_part[p].p + 2 * (size_t)c->mf_nblk;
        vin.n_ld_part = c->mf_nblk;
      } else {
        vin.he_part = (const double *)c->he_part[p].p;
        vin.n_he_part = ((d4 + 255) / 256) * M;
      }
    }
  } else {
    if (chained ? ch->first : !hit) launch_eps(c, rng, M);   // otherwise generated inside the previous VJP kernel
    vin.he_part = (const double *)c->he_part[p].p;
    vin.n_he_part = eps_blocks(c, M);
    if (c->bij_on) {   // Stacked bijector: explicit samples, transformed in place around whatever target is set
      launch_fr_sample(c, params, M, TGT_NONE, c->Z.p);
      launch_bij_forward(c, M);
      if (c->target == TGT_DENSE_GAUSS) {
        launch_rt_from_z(c, M);
        launch_fr_dense_target(c, M, want_grad);
        vin.ell_part = (const double *)c->ell_part[p].p;
        vin.n_ell_part = fr_dense_blocks(c, M);
        vin.ell = c->bij_ld.p;
        vin.n_ell = M;
      } else {
        if (logreg_uses_mfma(c, M)) launch_rt_from_z(c, M);
        if ((s = eval_generic_target(c, M, want_grad))) return s;
        vin.ell = c->ell.p;
        vin.n_ell = M;
      }
      launch_bij_backward(c, M, want_grad, c->target != TGT_DENSE_GAUSS);
    } else if (c->target == TGT_DIAG_GAUSS) {
      launch_fr_sample(c, params, M, TGT_DIAG_GAUSS, nullptr, prev);
      vin.ell_part = (const double *)c->ell_part[p].p;
      vin.n_ell_part = fr_sample_blocks(c, M);
    } else if (c->target == TGT_DENSE_GAUSS) {
      launch_fr_sample(c, params, M, TGT_DENSE_GAUSS, c->Z.p, prev);
      launch_fr_dense_target(c, M, want_grad);
      vin.ell_part = (const double *)c->ell_part[p].p;
      vin.n_ell_part = fr_dense_blocks(c, M);
    } else {
      const bool lr32 = c->target == TGT_LOGREG && c->cfg.dtype == MIVI_F32;
      launch_fr_sample(c, params, M, lr32 ? TGT_LOGREG : TGT_NONE, c->Z.p);   // LogReg: also leaves Z^T in RT
      if ((s = eval_generic_target(c, M, want_grad))) return s;
      vin.ell = c->ell.p;
      vin.n_ell = M;
    }
    if (want_grad && !stop_after_target) {   // (the Stein estimator stops here: eps, W = grad log pi and the ell sums are ready)
      if (out.ent_kind == MIVI_ENT_STL || out.ent_kind == MIVI_ENT_STL_ZERO_GRAD) {
        const size_t sh = (8 * (size_t)c->dP + 32 * 33) * c->esize;
        if (sh > 160 * 1024 && !c->stl_CT.p) return fail(c, MIVI_ERR_UNSUPPORTED, "full-rank STL: d too large for the LDS-resident solve");
        if (stl2_shape_ok(c, M)) launch_stl2(c, params, M);
        else launch_fr_stl(c, params, M);
      }
      EpsJob nx{};
      const EpsJob *next = nullptr;
      if (chained && ch->has_next) {
        nx.rng = ch->next_rng;
        nx.parity = p ^ 1;
        next = &nx;
      } else if (spec) {   // speculate that the caller asks for estimate idx + 1 next (an SGD loop does)
        nx.rng = rng;
        nx.rng.idx_base = rng.idx_base + 1ull;   // (a single call: the NEXT index, whatever stride an earlier batched call left on this context)
        nx.parity = p ^ 1;
        next = &nx;
      }
      if (spec) {          // this estimate's value rides in the same kernel: no separate value launch
        ValueJob self{vin, out};
        launch_fr_vjp(c, params, M, out, next, &self);
        c->pre_valid = true;
        c->pre_rng = nx.rng;
        c->pre_M = M;
        c->pre_parity = p ^ 1;
        c->pre_capturing = capturing;
        c->pre_capture_id = cap_id;
        if (ch) { ch->have_prev = false; ch->first = true; }
        HIPCHK(c, hipGetLastError());
        return MIVI_OK;
      }
      launch_fr_vjp(c, params, M, out, next, nullptr, chained ? upd : nullptr);
      vin.ld_part = (const double *)c->ld_part[p].p;   // emitted by the VJP kernel's diagonal tiles
      vin.n_ld_part = fr_ld_blocks(c);
    }
  }
  // ---- objective value (or the two scalar partials) -------------------------------------------------
  if (chained) {
    ch->prev.vin = vin;
    ch->prev.out = out;
    ch->have_prev = true;
    ch->first = false;
  } else {
    if (ch) { ch->have_prev = false; ch->first = true; }
    launch_value_only(c, params, vin, out);
  }
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}

void flush_chain(mivi_ctx *c, const void *params, Chain *ch) {
  if (ch->have_prev) launch_value_only(c, params, ch->prev.vin, ch->prev.out);
  ch->have_prev = false;
}

OutArgs final_out(mivi_ctx *c, void *value, void *grad) {
  OutArgs o{};
  o.grad = grad;
  o.value = value;
  o.partials = nullptr;
  o.partials_mode = 0;
  o.ent_kind = c->cfg.entropy;
  o.M_total = c->M_total;
  o.status = (int *)c->status.p;
  return o;
}

RngArgs rng_of(mivi_ctx *c, uint64_t idx) {
  RngArgs r;
  r.seed = c->cfg.seed;
  r.idx_base = idx;
  r.idx_ptr = c->idx_src;
  r.m_offset = c->cfg.m_offset;
  return r;
}

mivi_status_t mivi_sample(mivi_ctx_t *c, const void *params, uint64_t idx, void *Z, void *eps) {
  if (!c || !params || !Z) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  const int M = c->cfg.n_mc, d = c->cfg.d;
  mivi_status_t s = ensure_work(c, M);
  if (s) return s;
  if (c->cfg.family == MIVI_MEANFIELD) {
    launch_sample_mf(c, params, rng_of(c, idx), M, Z, eps, d, nullptr);
  } else {
    c->cur = 0;
    c->pre_valid = false;
    launch_eps(c, rng_of(c, idx), M);
    launch_fr_sample(c, params, M, TGT_NONE, Z);
    if (eps)
      HIPCHK(c, hipMemcpy2DAsync(eps, (size_t)d * c->esize, c->eps[0].p, (size_t)c->dP * c->esize, (size_t)d * c->esize, M,
                                 hipMemcpyDeviceToDevice, c->stream));
  }
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}

mivi_status_t mivi_estimate_gradient(mivi_ctx_t *c, const void *params, uint64_t idx, void *value, void *grad) {
  if (!c || !params || !value || !grad) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  return run_estimate(c, params, rng_of(c, idx), c->cfg.n_mc, 1, final_out(c, value, grad));
}

mivi_status_t read_status(mivi_ctx *c) {
  // this context's sticky flags (word 0) and, with interleaved chains, the children's (words 1 .. n_kids of the same buffer: their kernels are
  // joined into this stream by the batch's graph) -- one copy, one wait
  int sk[1 + mivi_ctx::kMaxKids] = {};
  const int nw = 1 + c->n_kids;
  HIPCHK(c, hipMemcpyAsync(sk, c->status.p, sizeof(int) * nw, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int st = 0;
  for (int j = 0; j < nw; ++j) st |= sk[j];
  c->last_status_bits = st;
  if (st) HIPCHK(c, hipMemsetAsync(c->status.p, 0, sizeof(int) * nw, c->stream));
  if (st & 8) return fail(c, MIVI_ERR_HIP, "a device-side wait expired (peer-to-peer exchange: a peer did not arrive within the spin budget -- lost rank or unmapped buffer; a launch-free loop with a per-step exchange -- funnel target, DoG / DoWG, tile ownership: its workgroups did not run side by side)");
  if (st & 2) return fail(c, MIVI_ERR_NONPOSITIVE_SCALE, "scale diagonal is not positive (use ClipScale)");
  if (st & 1) return fail(c, MIVI_ERR_NONFINITE, "the objective value is not finite: the optimization run diverged");
  return MIVI_OK;
}

mivi_status_t mivi_estimate_gradient_host(mivi_ctx_t *c, const void *params_h, uint64_t idx, void *value_h, void *grad_h) {
  if (!c || !params_h || !value_h || !grad_h) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  const size_t plen = (size_t)mivi_params_len(c), es = c->esize;
  HIPCHK(c, hipMemcpyAsync(c->tmp_params.p, params_h, plen * es, hipMemcpyHostToDevice, c->stream));
  char *o = (char *)c->tmp_out.p;
  mivi_status_t s = run_estimate(c, c->tmp_params.p, rng_of(c, idx), c->cfg.n_mc, 1, final_out(c, o, o + 16));
  if (s) return s;
  HIPCHK(c, hipMemcpyAsync(value_h, o, es, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(grad_h, o + 16, plen * es, hipMemcpyDeviceToHost, c->stream));
  return read_status(c);
}

mivi_status_t mivi_estimate_partials(mivi_ctx_t *c, const void *params, uint64_t idx, void *partials) {
  if (!c || !params || !partials) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  OutArgs o = final_out(c, nullptr, nullptr);
  o.partials = partials;
  o.partials_mode = 1;
  o.scalars_off = mivi_partials_len(c) - 2;
  return run_estimate(c, params, rng_of(c, idx), c->cfg.n_mc, 1, o);
}

mivi_status_t mivi_finalize(mivi_ctx_t *c, const void *params, const void *partials, void *value, void *grad) {
  if (!c || !params || !partials || !value || !grad) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  launch_finalize(c, params, partials, value, grad);
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}


