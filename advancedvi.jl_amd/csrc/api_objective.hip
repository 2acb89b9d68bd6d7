// libmivi C ABI, part 4: estimate_objective (src/algorithms/repgradelbo.jl:125-140) in chunks of samples, and
// gaussian_expectation_gradient_and_hessian! (src/algorithms/gauss_expected_grad_hess.jl).
#include "api_common.h"

mivi_status_t mivi_estimate_objective(mivi_ctx_t *c, const void *params, uint64_t idx, int32_t n_samples, int32_t entropy,
                                      void *value) {
  if (!c || !params || !value) return MIVI_ERR_BAD_ARG;
  if (n_samples <= 0) n_samples = c->cfg.n_mc;
  if (entropy < 0) entropy = c->cfg.entropy;
  if (entropy > MIVI_ENT_STL_ZERO_GRAD) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  const int CH = 16384;
  // all estimators share their *value* within {closed-form} / {MC, STL, STL-zero-grad}
  const bool engine_sized = c->cfg.family == MIVI_FULLRANK && c->cfg.dtype == MIVI_F32 && n_samples / c->cfg.n_mc >= 8 && c->cfg.n_mc >= 128;   // (tried below; falls back to the chunks)
  if (n_samples <= CH && !engine_sized) {
    OutArgs o = final_out(c, value, nullptr);
    o.ent_kind = entropy;
    o.M_total = n_samples;
    return run_estimate(c, params, rng_of(c, idx), n_samples, 0, o);
  }
  // chunked: the objective is a mean over samples plus parameter-only terms, so the weighted mean of the
  // chunk objectives is the full objective
  char *tmpv = (char *)c->tmp_out.p;
  int off0 = 0, first0 = 1;
  // Full-rank f32 contexts of an engine shape: whole blocks of n_mc samples as LANES of the batch engine (draws -> product + target -> value
  // workgroups; no VJP), the same (estimate index, global sample column) stream as the chunks below: 73 M -> 204 M samples/s at the
  // north-star shape.  Their values are averaged in a fixed order; what is left over goes through the chunk loop.
  {
    const int M = c->cfg.n_mc, lanes = n_samples / M;
    if (c->cfg.dtype == MIVI_F32 && lanes >= 8) {
      if (ensure(c, c->obj_vals, (size_t)lanes * 4 + 64, false) == MIVI_OK) {
        const mivi_status_t sb = mivi::fb_objective(c, params, idx, lanes, entropy, c->obj_vals.p);
        if (sb == MIVI_OK) {
          hipLaunchKernelGGL(k_mean_values_f32, dim3(1), dim3(256), 0, c->stream, (double *)c->acc.p, (const float *)c->obj_vals.p, lanes,
                             (double)M / (double)n_samples);
          off0 = lanes * M;
          first0 = 0;
        } else if (sb != MIVI_ERR_UNSUPPORTED) {
          return sb;
        }
      }
    }
  }
  for (int off = off0, first = first0; off < n_samples; off += CH, first = 0) {
    const int Mc = n_samples - off < CH ? n_samples - off : CH;
    OutArgs o = final_out(c, tmpv, nullptr);
    o.ent_kind = entropy;
    o.M_total = Mc;
    RngArgs r = rng_of(c, idx);
    r.m_offset += off;
    mivi_status_t s = run_estimate(c, params, r, Mc, 0, o);
    if (s) return s;
    const double w = (double)Mc / (double)n_samples;
    if (c->cfg.dtype == MIVI_F32)
      hipLaunchKernelGGL(k_acc_value_f32, dim3(1), dim3(1), 0, c->stream, (double *)c->acc.p, (const float *)tmpv, w, first);
    else
      hipLaunchKernelGGL(k_acc_value_f64, dim3(1), dim3(1), 0, c->stream, (double *)c->acc.p, (const double *)tmpv, w, first);
  }
  if (c->cfg.dtype == MIVI_F32)
    hipLaunchKernelGGL(k_store_value_f32, dim3(1), dim3(1), 0, c->stream, (float *)value, (const double *)c->acc.p);
  else
    hipLaunchKernelGGL(k_store_value_f64, dim3(1), dim3(1), 0, c->stream, (double *)value, (const double *)c->acc.p);
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}

mivi_status_t mivi_estimate_objective_host(mivi_ctx_t *c, const void *params_h, uint64_t idx, int32_t n_samples,
                                           int32_t entropy, void *value_h) {
  if (!c || !params_h || !value_h) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  const size_t plen = (size_t)mivi_params_len(c), es = c->esize;
  HIPCHK(c, hipMemcpyAsync(c->tmp_params.p, params_h, plen * es, hipMemcpyHostToDevice, c->stream));
  char *o = (char *)c->tmp_out.p + 16;
  mivi_status_t s = mivi_estimate_objective(c, c->tmp_params.p, idx, n_samples, entropy, o);
  if (s) return s;
  HIPCHK(c, hipMemcpyAsync(value_h, o, es, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int st = 0;
  HIPCHK(c, hipMemcpy(&st, c->status.p, sizeof(int), hipMemcpyDeviceToHost));
  if (st) HIPCHK(c, hipMemset(c->status.p, 0, sizeof(int)));
  if (st & 2) return fail(c, MIVI_ERR_NONPOSITIVE_SCALE, "scale diagonal is not positive (use ClipScale)");
  return MIVI_OK;  // a non-finite value is returned as-is, like the reference's estimate_objective
}

// gaussian_expectation_gradient_and_hessian!, first-order branch (src/algorithms/gauss_expected_grad_hess.jl:32-60):
//   u ~ N(0, I) (d x n), z = C u + m, per sample (logpi, g) from the target;  logpi_avg = mean logpi, grad = mean g,
//   hess = C' \ mean(u g').  Same eps stream and sample/target kernels as the ELBO path; the extra work is the full
//   eps G^T product (k_stein_outer) and one back substitution with d right-hand sides (the STL solve kernels).
mivi_status_t mivi_gauss_expected_grad_hess(mivi_ctx_t *c, const void *params, uint64_t idx, int32_t n_samples,
                                            void *logpi_avg, void *grad, void *hess) {
  if (!c || !params || !logpi_avg || !grad || !hess) return MIVI_ERR_BAD_ARG;
  if (c->cfg.family != MIVI_FULLRANK)
    return fail(c, MIVI_ERR_UNSUPPORTED, "gauss_expected_grad_hess takes a triangular scale (full-rank family)");
  if (n_samples <= 0) n_samples = c->cfg.n_mc;
  (void)hipSetDevice(c->cfg.device);
  const int d = c->cfg.d, dP = round_up(d, 64);
  const size_t es = c->esize;
  if ((8 * (size_t)dP + 32 * 33) * es > 160 * 1024 && ((size_t)dP * 16 + 8 * 8 * 64) * es > 160 * 1024)
    return fail(c, MIVI_ERR_UNSUPPORTED, "gauss_expected_grad_hess: d too large for the LDS-resident solve");
  mivi_status_t s;
  if ((s = ensure(c, c->stein_A, (size_t)dP * dP * es, true)) || (s = ensure(c, c->stein_g, (size_t)(d + 8) * sizeof(double), true)) ||
      (s = ensure(c, c->stl_CT, (size_t)dP * dP * es, true)) || (s = ensure(c, c->stl_Dinv, (size_t)((d + 31) / 32) * 1024 * es, false)))
    return s;
  const bool stl2 = stl2_shape_ok(c, d);   // second-generation solve with the d columns of the product as right-hand sides
  if (stl2 && ((s = ensure(c, c->stl_X, ((size_t)d * d + (size_t)(d / 2) * (d / 2)) * es + 4096, false)) || (s = ensure(c, c->stl_F, mivi::stl_pack_units(d) * 4, false)))) return s;
  const int CH = 16384;
  const bool single_chunk = n_samples <= CH;
  bool pack_done = false, tail_done = false;
  char *part = (char *)c->tmp_out.p;   // [sum ell, sum 0.5 eps^2] of a chunk
  for (int off = 0, first = 1; off < n_samples; off += CH, first = 0) {
    const int Mc = n_samples - off < CH ? n_samples - off : CH;
    OutArgs o = final_out(c, nullptr, nullptr);
    o.partials = part;
    o.partials_mode = 1;
    o.scalars_off = 0;
    o.ent_kind = MIVI_ENT_CLOSED_FORM;
    o.M_total = Mc;
    RngArgs r = rng_of(c, idx);
    r.m_offset += off;
    c->want_stl_pack = stl2 && first;   // the solve's parameter-only preparation rides in the first chunk's sampling kernel
    c->stl_pack_done = false;
    // second-generation accumulation kernel (f32, d and chunk multiples of 64 / 128): with ONE chunk it also assembles the chunk's
    // value partials (no k_value_only launch) and writes grad / logpi_avg itself (no finishing launch)
    const bool st2 = c->cfg.dtype == MIVI_F32 && lds_stein_ok(c, Mc);
    ValueJob vj{};
    c->value_deferred = false;
    c->defer_value = (st2 && single_chunk) ? &vj : nullptr;
    s = run_estimate(c, params, r, Mc, 1, o, nullptr, nullptr, true);
    c->defer_value = nullptr;
    c->want_stl_pack = false;
    if (s) return s;
    const bool fused_tail = c->value_deferred;
    c->value_deferred = false;
    if (first) pack_done = c->stl_pack_done;
    if (single_chunk) {
      // (its partial is read by the finishing kernel directly: no accumulation launch)
    } else if (c->cfg.dtype == MIVI_F32)
      hipLaunchKernelGGL(k_acc_value_f32, dim3(1), dim3(1), 0, c->stream, (double *)c->acc.p, (const float *)part, 1.0, first);
    else
      hipLaunchKernelGGL(k_acc_value_f64, dim3(1), dim3(1), 0, c->stream, (double *)c->acc.p, (const double *)part, 1.0, first);
    const bool last = off + CH >= n_samples;
    const double scale = last ? 1.0 / (double)n_samples : 1.0;
    if (st2) {
      launch_lds_stein_outer(c, Mc, c->stein_A.p, (double *)c->stein_g.p, first, scale, (double)n_samples, fused_tail ? grad : nullptr,
                             fused_tail ? logpi_avg : nullptr, fused_tail ? &vj : nullptr);
      tail_done = fused_tail;
    } else {
      launch_stein_outer(c, Mc, c->stein_A.p, (double *)c->stein_g.p, first, scale);
    }
  }
  if (!tail_done)
    launch_stein_finish(c, (double)n_samples, (const double *)c->stein_g.p, (const double *)c->acc.p, single_chunk ? part : nullptr, grad, logpi_avg);
  if (stl2) {
    launch_stl2(c, params, d, pack_done, c->stein_A.p, hess, true);   // hess = C^-T (eps G^T / n), written (not added)
  } else {
    HIPCHK(c, hipMemsetAsync(hess, 0, (size_t)d * d * es, c->stream));
    launch_fr_stl(c, params, d, c->stein_A.p, hess);
  }
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}

mivi_status_t mivi_set_target_hess_callback(mivi_ctx_t *c, mivi_logdensity_gradient_and_hessian_fn fn, void *user) {
  if (!c) return MIVI_ERR_BAD_ARG;
  c->cb_hess = fn;   // (nullptr removes it)
  c->cb_hess_user = user;
  return MIVI_OK;
}

// gaussian_expectation_gradient_and_hessian!, second-order branch (src/algorithms/gauss_expected_grad_hess.jl:61-83): z = rand(rng, q, n)
// (the same eps stream as the first-order branch), per sample (logpi, g, H) from the target; the three sample averages.  Built-in
// Gaussian targets: sampling + fused target kernels as the first-order branch, column sums of G, the constant Hessian written exactly.
// Plugin with a Hessian callback: Z to the host in chunks, the callback returns ell, G and the chunk's Hessian SUM; accumulated in f64.
mivi_status_t mivi_gauss_expected_grad_hess2(mivi_ctx_t *c, const void *params, uint64_t idx, int32_t n_samples,
                                             void *logpi_avg, void *grad, void *hess) {
  if (!c || !params || !logpi_avg || !grad || !hess) return MIVI_ERR_BAD_ARG;
  if (c->cfg.family != MIVI_FULLRANK)
    return fail(c, MIVI_ERR_UNSUPPORTED, "gauss_expected_grad_hess takes a triangular scale (full-rank family)");
  const bool sampled = target_has_hess2(c);   // logistic regression / funnel: the Hessian depends on z -- linear in per-sample statistics (kernels_hess2.hip)
  const bool builtin = ((c->target == TGT_DIAG_GAUSS || c->target == TGT_DENSE_GAUSS) && !c->bij_on) || sampled;
  const bool plugin = c->target == TGT_CALLBACK && c->cb_hess && !c->bij_on;
  if (!builtin && !plugin)
    return fail(c, MIVI_ERR_UNSUPPORTED, "second-order branch: the target has no Hessian here (built-in targets without a Stacked bijector, or a "
                                         "plugin with mivi_set_target_hess_callback); use mivi_gauss_expected_grad_hess (Stein identity)");
  if (n_samples <= 0) n_samples = c->cfg.n_mc;
  (void)hipSetDevice(c->cfg.device);
  const int d = c->cfg.d;
  const size_t es = c->esize;
  const int CH = 16384;
  mivi_status_t s;
  if (plugin) {
    std::vector<double> gs((size_t)d, 0.0), Hs((size_t)d * d, 0.0);
    double ls = 0.0;
    std::vector<char> hH((size_t)d * d * es);
    for (int off = 0; off < n_samples; off += CH) {
      const int Mc = n_samples - off < CH ? n_samples - off : CH;
      if ((s = ensure_work(c, Mc))) return s;
      RngArgs r = rng_of(c, idx);
      r.m_offset += off;
      c->cur = 0;
      c->pre_valid = false;
      launch_eps(c, r, Mc);
      launch_fr_sample(c, params, Mc, TGT_NONE, c->Z.p);
      c->h_Z.resize((size_t)d * Mc * es);
      c->h_G.resize((size_t)d * Mc * es);
      c->h_ell.resize((size_t)Mc * es);
      HIPCHK(c, hipMemcpyAsync(c->h_Z.data(), c->Z.p, (size_t)d * Mc * es, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (c->cb_hess(c->cb_hess_user, c->h_Z.data(), d, Mc, c->h_ell.data(), c->h_G.data(), hH.data()) != 0)
        return fail(c, MIVI_ERR_BAD_ARG, "target Hessian callback returned non-zero");
      for (int m = 0; m < Mc; ++m) {
        ls += host_get(c->h_ell.data(), c->cfg.dtype, (size_t)m);
        for (int i = 0; i < d; ++i) gs[i] += host_get(c->h_G.data(), c->cfg.dtype, (size_t)m * d + i);
      }
      for (size_t e = 0; e < (size_t)d * d; ++e) Hs[e] += host_get(hH.data(), c->cfg.dtype, e);
    }
    const double inv = 1.0 / (double)n_samples;
    std::vector<char> out((1 + (size_t)d + (size_t)d * d) * es);
    auto put = [&](size_t k, double v) {
      if (c->cfg.dtype == MIVI_F32) ((float *)out.data())[k] = (float)v;
      else ((double *)out.data())[k] = v;
    };
    put(0, ls * inv);
    for (int i = 0; i < d; ++i) put(1 + (size_t)i, gs[i] * inv);
    for (size_t e = 0; e < (size_t)d * d; ++e) put(1 + (size_t)d + e, Hs[e] * inv);
    HIPCHK(c, hipMemcpyAsync(logpi_avg, out.data(), es, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(grad, out.data() + es, (size_t)d * es, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(hess, out.data() + (1 + (size_t)d) * es, (size_t)d * d * es, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // (`out` lives on this frame)
    return MIVI_OK;
  }
  if ((s = ensure(c, c->stein_g, (size_t)(d + 8) * sizeof(double), true))) return s;
  if (sampled) {
    if ((s = ensure(c, c->h2_acc, target_hess2_bytes(c), false))) return s;
    if (!target_hess2_begin(c)) return fail(c, MIVI_ERR_HIP, "second-order branch: clearing the accumulators failed");
  }
  const bool single_chunk = n_samples <= CH;
  char *part = (char *)c->tmp_out.p;   // [sum ell, sum 0.5 eps^2] of a chunk
  for (int off = 0, first = 1; off < n_samples; off += CH, first = 0) {
    const int Mc = n_samples - off < CH ? n_samples - off : CH;
    OutArgs o = final_out(c, nullptr, nullptr);
    o.partials = part;
    o.partials_mode = 1;
    o.scalars_off = 0;
    o.ent_kind = MIVI_ENT_CLOSED_FORM;
    o.M_total = Mc;
    RngArgs r = rng_of(c, idx);
    r.m_offset += off;
    if ((s = run_estimate(c, params, r, Mc, 1, o, nullptr, nullptr, true))) return s;   // sampling + target: W = grad logpi(z), the chunk's value partials
    if (!single_chunk) {
      if (c->cfg.dtype == MIVI_F32)
        hipLaunchKernelGGL(k_acc_value_f32, dim3(1), dim3(1), 0, c->stream, (double *)c->acc.p, (const float *)part, 1.0, first);
      else
        hipLaunchKernelGGL(k_acc_value_f64, dim3(1), dim3(1), 0, c->stream, (double *)c->acc.p, (const double *)part, 1.0, first);
    }
    launch_stein_gsum(c, Mc, (double *)c->stein_g.p, first);
    if (sampled) target_hess2_accumulate(c, Mc);   // (the chunk's samples are still in c->Z)
  }
  launch_stein_finish(c, (double)n_samples, (const double *)c->stein_g.p, (const double *)c->acc.p, single_chunk ? part : nullptr, grad, logpi_avg);
  if (sampled) target_hess2_finish(c, n_samples, hess);
  else launch_const_hess(c, hess);
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}

mivi_status_t mivi_gauss_expected_grad_hess_host(mivi_ctx_t *c, const void *params_h, uint64_t idx, int32_t n_samples,
                                                 void *logpi_avg_h, void *grad_h, void *hess_h) {
  if (!c || !params_h || !logpi_avg_h || !grad_h || !hess_h) return MIVI_ERR_BAD_ARG;
  if (c->cfg.family != MIVI_FULLRANK)
    return fail(c, MIVI_ERR_UNSUPPORTED, "gauss_expected_grad_hess takes a triangular scale (full-rank family)");
  (void)hipSetDevice(c->cfg.device);
  const size_t plen = (size_t)mivi_params_len(c), es = c->esize, d = (size_t)c->cfg.d;
  HIPCHK(c, hipMemcpyAsync(c->tmp_params.p, params_h, plen * es, hipMemcpyHostToDevice, c->stream));
  // the chunk partials use tmp_out[0..1]; results go behind them: [.., logpi (slot 2), grad (d), hess (d*d)] <= params_len + 16
  char *o = (char *)c->tmp_out.p + 2 * 8;
  mivi_status_t s = mivi_gauss_expected_grad_hess(c, c->tmp_params.p, idx, n_samples, o, o + 8, o + 8 + d * es);
  if (s) return s;
  HIPCHK(c, hipMemcpyAsync(logpi_avg_h, o, es, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(grad_h, o + 8, d * es, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(hess_h, o + 8 + d * es, d * d * es, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MIVI_OK;
}

mivi_status_t mivi_gauss_expected_grad_hess2_host(mivi_ctx_t *c, const void *params_h, uint64_t idx, int32_t n_samples,
                                                  void *logpi_avg_h, void *grad_h, void *hess_h) {
  if (!c || !params_h || !logpi_avg_h || !grad_h || !hess_h) return MIVI_ERR_BAD_ARG;
  if (c->cfg.family != MIVI_FULLRANK)
    return fail(c, MIVI_ERR_UNSUPPORTED, "gauss_expected_grad_hess takes a triangular scale (full-rank family)");
  (void)hipSetDevice(c->cfg.device);
  const size_t plen = (size_t)mivi_params_len(c), es = c->esize, d = (size_t)c->cfg.d;
  HIPCHK(c, hipMemcpyAsync(c->tmp_params.p, params_h, plen * es, hipMemcpyHostToDevice, c->stream));
  char *o = (char *)c->tmp_out.p + 2 * 8;   // (behind the chunk partials, as the first-order _host entry)
  mivi_status_t s = mivi_gauss_expected_grad_hess2(c, c->tmp_params.p, idx, n_samples, o, o + 8, o + 8 + d * es);
  if (s) return s;
  HIPCHK(c, hipMemcpyAsync(logpi_avg_h, o, es, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(grad_h, o + 8, d * es, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(hess_h, o + 8 + d * es, d * d * es, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return MIVI_OK;
}

