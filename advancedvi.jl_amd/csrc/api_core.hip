// libmivi C ABI (include/mivi.h), part 1: context management, target / bijector set-up, small setters, the host-side RNG restatement.
// Reference call stack being replaced: SURVEY.md section 3.2 (estimate_gradient! -> _value_and_gradient! -> AD of
// estimate_repgradelbo_ad_forward).  The other parts: api_estimate.hip (one estimate), api_batch.hip (estimates at fixed
// parameters), api_objective.hip (estimate_objective, the Gaussian-expectation gradient / Hessian), api_dist.hip (sharded
// estimates), api_optimize.hip (update rules, the device-resident loop), api_profile.hip (per-kernel timing entries).
#include "api_common.h"

// ---- one-thread kernels shared by the api_*.hip units (declared in api_common.h) ----
// weighted accumulation of chunk objective values
__global__ void k_acc_value_f32(double *acc, const float *v, double w, int first) { acc[0] = (first ? 0.0 : acc[0]) + w * (double)v[0]; }
__global__ void k_acc_value_f64(double *acc, const double *v, double w, int first) { acc[0] = (first ? 0.0 : acc[0]) + w * v[0]; }
__global__ void k_store_value_f32(float *out, const double *acc) { out[0] = (float)acc[0]; }
// acc = w * sum of n values (objective mode of the batch engine: the lanes' values), a fixed order: thread t sums entries t, t + 256, .. in
// f64, the 256 partials are added in index order by thread 0
__global__ __launch_bounds__(256) void k_mean_values_f32(double *acc, const float *v, int n, double w) {
  __shared__ double part[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)v[i];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 256; ++i) t += part[i];
    acc[0] = w * t;
  }
}
__global__ void k_store_value_f64(double *out, const double *acc) { out[0] = acc[0]; }
// device counters of the graph-batched calls, set BY VALUE (an async copy from a stack local may outlive the caller's frame)
__global__ void k_set_u64x2(uint64_t *dst, uint64_t a, uint64_t b, int n) { dst[0] = a; if (n > 1) dst[1] = b; }
__global__ void k_bump_u64(uint64_t *dst, uint64_t by) { dst[0] += by; }
// latency floor (mivi_profile_kernel which = 9): a kernel that does nothing, launched with the grid / block / LDS footprint of a real one
__global__ void k_empty(int *sink) {
  extern __shared__ int lds_dyn[];
  if (sink && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) sink[0] = lds_dyn[0];
}


mivi_status_t fail(mivi_ctx *c, mivi_status_t s, const char *msg) {
  if (c) c->err = msg;
  return s;
}

namespace mivi {
bool grid_resident(const mivi_ctx *c, const void *kernel, int block, size_t dyn_lds, long long grid) {
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, dyn_lds) != hipSuccess) { (void)hipGetLastError(); return false; }
  return c->n_cu > 0 && (long long)nb * c->n_cu >= grid;
}
}  // namespace mivi

mivi_status_t ensure(mivi_ctx *c, DevBuf &b, size_t bytes, bool zero) {
  if (b.bytes >= bytes && b.p) return MIVI_OK;
  if (b.p) HIPCHK(c, hipFree(b.p));
  b.p = nullptr;
  b.bytes = 0;
  if (bytes == 0) bytes = 16;
  HIPCHK(c, hipMalloc(&b.p, bytes));
  b.bytes = bytes;
  if (zero) HIPCHK(c, hipMemsetAsync(b.p, 0, bytes, c->stream));
  return MIVI_OK;
}


// A captured graph bakes buffer pointers, leading dimensions and work-list contents; the eps speculation remembers a buffer
// parity.  Anything that reallocates or rewrites those calls this.
namespace mivi {
void invalidate_graph(mivi_ctx *c) {
  c->pre_valid = false;
  ++c->target_gen;
  if (c->graph.exec) { (void)hipGraphExecDestroy(c->graph.exec); c->graph = GraphCache{}; }
}
}  // namespace mivi

// work buffers for up to M samples per launch
mivi_status_t ensure_work(mivi_ctx *c, int M) {
  const int d = c->cfg.d;
  const size_t es = c->esize;
  mivi_status_t s;
  if (M > c->cap_M) {
    invalidate_graph(c);   // the work buffers below are reallocated and MP / dP change under any cached graph
    // a capacity change re-zeros the padded RNG buffers (their padding must stay 0 / finite)
    const int capM = round_up(M, 64);
    c->dP = round_up(d, 64);
    c->MP = capM;
    const int d4 = (d + 3) / 4;
    // per-workgroup ell partials: the XCD-interleaved work tables have up to 8*ceil(nb/8)*ncb slots
    size_t n_part = 8 * (size_t)(((d + 31) / 32 + 7) / 8) * (size_t)((capM + 31) / 32) + 64;
    size_t n_he = (size_t)((d + 15) / 16) * (size_t)(capM / 64 + 1);
    const size_t n_he_mf = (size_t)((d4 + 255) / 256) * (size_t)capM;
    if (n_he_mf > n_he) n_he = n_he_mf;
    const size_t ncc = (size_t)(capM / 256 + 2);
    for (int b = 0; b < 2; ++b) {
      if (c->cfg.family == MIVI_FULLRANK) {
        c->eps[b].bytes = 0; c->epsT[b].bytes = 0;
        if ((s = ensure(c, c->eps[b], (size_t)c->dP * c->MP * es, true))) return s;
        if ((s = ensure(c, c->epsT[b], (size_t)c->dP * c->MP * es, true))) return s;
      }
      if ((s = ensure(c, c->ell_part[b], n_part * sizeof(double), false))) return s;
      if ((s = ensure(c, c->he_part[b], (n_he + 64) * sizeof(double), false))) return s;
      if ((s = ensure(c, c->sc_part[b], 6 * ncc * d4 * sizeof(double) + 64, false))) return s;
      if ((s = ensure(c, c->ld_part[b], 2 * (size_t)((d + 31) / 32) * sizeof(double) + 64, false))) return s;
    }
    if (c->target == TGT_DENSE_GAUSS || (c->target == TGT_LOGREG && c->cfg.dtype == MIVI_F32)) {
      c->RT.bytes = 0;
      if ((s = ensure(c, c->RT, (size_t)c->dP * c->MP * es, true))) return s;
    }
    if (c->cfg.family == MIVI_FULLRANK && (c->cfg.entropy == MIVI_ENT_STL || c->cfg.entropy == MIVI_ENT_STL_ZERO_GRAD)) {
      if ((s = ensure(c, c->stl_CT, (size_t)c->dP * c->dP * es, true))) return s;
      if ((s = ensure(c, c->stl_Dinv, (size_t)((d + 63) / 64) * 4096 * es, false))) return s;   // 32x32 or 64x64 diagonal inverses
      if ((s = ensure(c, c->stl_X, ((size_t)d * capM + (size_t)(d / 2) * (d / 2)) * es + 4096, false))) return s;   // X2, Y1, F (kernels_stl.hip)
      if (d % 128 == 0 && (s = ensure(c, c->stl_F, mivi::stl_pack_units(d) * 4, false))) return s;   // + developer stamp page (MIVI_STL_STAMPS)
    }
    if ((s = ensure(c, c->Z, (size_t)d * capM * es, false))) return s;
    if ((s = ensure(c, c->W, (size_t)d * capM * es, false))) return s;
    if ((s = ensure(c, c->ell, (size_t)capM * es, false))) return s;
    if (c->bij_on && (s = ensure(c, c->bij_ld, (size_t)capM * es, false))) return s;
    if ((s = ensure(c, c->row_part, ncc * d4 * 8 * sizeof(double), false))) return s;
    c->cap_M = capM;
  }
  return MIVI_OK;
}


int32_t mivi_version(void) { return MIVI_VERSION_MAJOR * 1000 + MIVI_VERSION_MINOR; }

const char *mivi_last_error(const mivi_ctx_t *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int64_t mivi_params_len(const mivi_ctx_t *c) {
  const int64_t d = c->cfg.d;
  return c->cfg.family == MIVI_MEANFIELD ? 2 * d : d + d * d;
}
int64_t mivi_partials_len(const mivi_ctx_t *c) {
  const int64_t d = c->cfg.d;
  return (c->cfg.family == MIVI_MEANFIELD ? 2 * d : d + d * (d + 1) / 2) + 2;
}

mivi_status_t mivi_create(const mivi_config_t *cfg, mivi_ctx_t **out) {
  if (!cfg || !out) return MIVI_ERR_BAD_ARG;
  if (cfg->d <= 0 || cfg->n_mc <= 0) return MIVI_ERR_BAD_ARG;
  if (cfg->dtype != MIVI_F32 && cfg->dtype != MIVI_F64) return MIVI_ERR_BAD_ARG;
  if (cfg->family != MIVI_MEANFIELD && cfg->family != MIVI_FULLRANK) return MIVI_ERR_BAD_ARG;
  if (cfg->entropy < 0 || cfg->entropy > MIVI_ENT_STL_ZERO_GRAD) return MIVI_ERR_BAD_ARG;
  if (cfg->m_offset < 0) return MIVI_ERR_BAD_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device < 0 || cfg->device >= ndev) {
    fprintf(stderr, "libmivi: no usable HIP device %d (found %d): this library has no CPU fallback\n", cfg->device, ndev);
    return MIVI_ERR_HIP;
  }
  mivi_ctx *c = new mivi_ctx();
  c->cfg = *cfg;
  c->esize = cfg->dtype == MIVI_F32 ? 4 : 8;
  {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && v > 0) c->n_cu = v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, cfg->device) == hipSuccess && v > 0) c->lds_max = (size_t)v;
    (void)hipGetLastError();
  }
  c->M_total = cfg->m_total > 0 ? cfg->m_total : cfg->n_mc;
  if (hipSetDevice(cfg->device) != hipSuccess) { delete c; return MIVI_ERR_HIP; }
  if (!cfg->own_stream) {
    c->stream = (hipStream_t)cfg->stream;   // NULL = the null stream
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return MIVI_ERR_HIP; }
    c->own_stream = true;
  }
  mivi_status_t s;
  if ((s = ensure(c, c->status, 64, true)) ||
      (s = ensure(c, c->d_idx, 64, true)) || (s = ensure(c, c->acc, 64, true)) ||
      (s = ensure(c, c->dog_part, (2 * 512 + 8) * sizeof(double), true)) ||
      (s = ensure(c, c->tmp_params, (size_t)mivi_params_len(c) * c->esize, false)) ||
      (s = ensure(c, c->tmp_out, ((size_t)mivi_params_len(c) + 16) * c->esize, false))) {
    delete c;
    return s;
  }
  *out = c;
  return MIVI_OK;
}

mivi_status_t mivi_destroy(mivi_ctx_t *c) {
  if (!c) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  (void)hipStreamSynchronize(c->stream);
  for (int j = 0; j < c->n_kids; ++j) {
    (void)mivi_destroy(c->kids[j]);
    if (c->ev_join[j]) (void)hipEventDestroy(c->ev_join[j]);
    if (c->kid_out[j].p) (void)hipFree(c->kid_out[j].p);
  }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->is_child) c->t_mean = c->t_istd = c->t_prec = c->status = mivi::DevBuf{};   // borrowed from the parent
  (void)mivi_comm_destroy(c);
  if (c->graph.exec) (void)hipGraphExecDestroy(c->graph.exec);
  DevBuf *bufs[] = {&c->t_mean, &c->t_istd, &c->t_prec, &c->lr_X_own, &c->lr_y_own, &c->lr_scratch, &c->lr_part, &c->lr_Xrm,
                    &c->eps[0], &c->eps[1], &c->epsT[0], &c->epsT[1], &c->Z, &c->W, &c->RT, &c->ell, &c->X,
                    &c->ell_part[0], &c->ell_part[1], &c->he_part[0], &c->he_part[1], &c->row_part,
                    &c->sc_part[0], &c->sc_part[1], &c->ld_part[0], &c->ld_part[1], &c->tabA, &c->tabB, &c->tabD, &c->stl_CT, &c->stl_Dinv, &c->stl_X, &c->stl_F, &c->lr_xmax, &c->lr_XA, &c->lr_XB, &c->lr_ZP, &c->lr_Xsub, &c->lr_ysub, &c->lr_Xrm_sub, &c->lr_idx, &c->dog_part, &c->stein_A, &c->stein_g, &c->h2_acc, &c->bij_mask, &c->bij_ld, &c->dist_P, &c->dist_P2, &c->dist_ring[0], &c->dist_ring[1], &c->dist_ring[2], &c->dist_ring[3], &c->dist_ring[4], &c->dist_ring[5], &c->p2p_scratch, &c->dist_S, &c->dist_F, &c->p2p_tab, &c->p2p_ctr, &c->lds_tabV, &c->lds_tabV64, &c->lds_tabS, &c->lds_tabSt, &c->status, &c->d_idx, &c->acc, &c->tmp_params, &c->tmp_out, &c->obj_vals};
  for (DevBuf *b : bufs)
    if (b->p) (void)hipFree(b->p);
  {
    DevBuf *fbb[] = {&c->fb.CA, &c->fb.epsP, &c->fb.WV, &c->fb.ell, &c->fb.he, &c->fb.ld, &c->fb.grads, &c->fb.values, &c->fb.PA, &c->fb.RP, &c->fb.Tinv, &c->fb.TA, &c->fb.Eye, &c->snap, &c->fb.cscale, &c->fb.pscale, &c->fb.tscale, &c->fb.winv, &c->fb.rinv, &c->p2p_direct, &c->rows_eps, &c->gen_scratch};
    for (DevBuf *b : fbb)
      if (b->p) (void)hipFree(b->p);
    for (auto &tb : c->fb.tab) {
      if (tb.prod.p) (void)hipFree(tb.prod.p);
      if (tb.vjp.p) (void)hipFree(tb.vjp.p);
      if (tb.prod2.p) (void)hipFree(tb.prod2.p);
      if (tb.prod3.p) (void)hipFree(tb.prod3.p);
    }
  }
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  if (c->cap_stream) (void)hipStreamDestroy(c->cap_stream);
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  if (c->comm_stream2) (void)hipStreamDestroy(c->comm_stream2);
  if (c->fb_comm_stream) (void)hipStreamDestroy(c->fb_comm_stream);
  for (int k = 0; k < 2; ++k) {
    if (c->fb_ev_part[k]) (void)hipEventDestroy(c->fb_ev_part[k]);
    if (c->fb_ev_comm[k]) (void)hipEventDestroy(c->fb_ev_comm[k]);
  }
  for (int k = 0; k < 2; ++k) {
    if (c->ev_part[k]) (void)hipEventDestroy(c->ev_part[k]);
    if (c->ev_comm[k]) (void)hipEventDestroy(c->ev_comm[k]);
  }
  delete c;
  return MIVI_OK;
}

mivi_status_t mivi_set_stream(mivi_ctx_t *c, void *s) {
  if (!c) return MIVI_ERR_BAD_ARG;
  if (c->own_stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); c->own_stream = false; }
  c->stream = (hipStream_t)s;   // NULL = the null stream
  c->pre_valid = false;
  if (c->graph.exec) { (void)hipGraphExecDestroy(c->graph.exec); c->graph = GraphCache{}; }
  return MIVI_OK;
}


// waits for the context's stream and reports (then clears) the sticky device flags of the estimates since the last read
mivi_status_t mivi_synchronize(mivi_ctx_t *c) {
  if (!c) return MIVI_ERR_BAD_ARG;
  return read_status(c);
}

// ---------------------------------------------------------------------------------------------
// targets
// ---------------------------------------------------------------------------------------------
double host_get(const void *p, int dtype, size_t i) {
  return dtype == MIVI_F32 ? (double)((const float *)p)[i] : ((const double *)p)[i];
}
static mivi_status_t upload_vec(mivi_ctx *c, DevBuf &b, const std::vector<double> &v) {
  mivi_status_t s = ensure(c, b, v.size() * c->esize, false);
  if (s) return s;
  if (c->cfg.dtype == MIVI_F32) {
    std::vector<float> f(v.begin(), v.end());
    HIPCHK(c, hipMemcpy(b.p, f.data(), f.size() * 4, hipMemcpyHostToDevice));
  } else {
    HIPCHK(c, hipMemcpy(b.p, v.data(), v.size() * 8, hipMemcpyHostToDevice));
  }
  return MIVI_OK;
}

mivi_status_t mivi_set_target_diag_gauss(mivi_ctx_t *c, const void *mean, const void *stdv) {
  if (!c || !mean || !stdv) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  const int d = c->cfg.d;
  std::vector<double> m(d), is(d);
  double cst = -0.5 * d * kLog2Pi;
  for (int i = 0; i < d; ++i) {
    const double s = host_get(stdv, c->cfg.dtype, i);
    if (!(s > 0.0)) return fail(c, MIVI_ERR_BAD_ARG, "diag_gauss: std must be positive");
    m[i] = host_get(mean, c->cfg.dtype, i);
    is[i] = 1.0 / s;
    cst -= log(s);
  }
  mivi_status_t s;
  if ((s = upload_vec(c, c->t_mean, m)) || (s = upload_vec(c, c->t_istd, is))) return s;
  c->t_const = cst;
  c->target = TGT_DIAG_GAUSS;
  invalidate_graph(c);
  return MIVI_OK;
}

mivi_status_t mivi_set_target_dense_gauss(mivi_ctx_t *c, const void *mean, const void *Lh) {
  if (!c || !mean || !Lh) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  const int d = c->cfg.d;
  const int dP = round_up(d, 64);
  std::vector<double> L((size_t)d * d, 0.0), Li((size_t)d * d, 0.0), m(d);
  double logdet = 0.0;
  for (int j = 0; j < d; ++j)
    for (int i = j; i < d; ++i) L[(size_t)j * d + i] = host_get(Lh, c->cfg.dtype, (size_t)j * d + i);
  for (int i = 0; i < d; ++i) {
    m[i] = host_get(mean, c->cfg.dtype, i);
    const double lii = L[(size_t)i * d + i];
    if (!(lii > 0.0)) return fail(c, MIVI_ERR_BAD_ARG, "dense_gauss: Cholesky diagonal must be positive");
    logdet += 2.0 * log(lii);
  }
  // Li = L^-1 (lower), column by column (forward substitution), fp64 on the host
  for (int j = 0; j < d; ++j) {
    Li[(size_t)j * d + j] = 1.0 / L[(size_t)j * d + j];
    for (int i = j + 1; i < d; ++i) {
      double s = 0.0;
      for (int k = j; k < i; ++k) s += L[(size_t)k * d + i] * Li[(size_t)j * d + k];
      Li[(size_t)j * d + i] = -s / L[(size_t)i * d + i];
    }
  }
  // P = Li^T Li, padded to dP x dP (zeros)
  std::vector<double> P((size_t)dP * dP, 0.0);
  for (int i = 0; i < d; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0.0;
      for (int k = i; k < d; ++k) s += Li[(size_t)i * d + k] * Li[(size_t)j * d + k];
      P[(size_t)j * dP + i] = s;
      P[(size_t)i * dP + j] = s;
    }
  mivi_status_t s;
  if ((s = upload_vec(c, c->t_mean, m)) || (s = upload_vec(c, c->t_prec, P))) return s;
  c->t_const = -0.5 * logdet - 0.5 * d * kLog2Pi;
  c->target = TGT_DENSE_GAUSS;
  c->fb.PA_valid = false;   // (the batch engine's planes of P)
  c->cap_M = 0;  // force (re)allocation of RT
  invalidate_graph(c);
  return MIVI_OK;
}

mivi_status_t mivi_set_target_logreg(mivi_ctx_t *c, const void *X, const uint8_t *y, int64_t n, int32_t variant,
                                     double likeadj, int32_t on_device) {
  if (!c || !X || !y || n <= 0 || (variant != 0 && variant != 1) || c->cfg.d < 2) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  const int p = c->cfg.d - 1;
  if (on_device) {
    c->lr_X = X;
    c->lr_y = y;
  } else {
    mivi_status_t s;
    if ((s = ensure(c, c->lr_X_own, (size_t)n * p * c->esize, false)) || (s = ensure(c, c->lr_y_own, (size_t)n, false)))
      return s;
    HIPCHK(c, hipMemcpy(c->lr_X_own.p, X, (size_t)n * p * c->esize, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->lr_y_own.p, y, (size_t)n, hipMemcpyHostToDevice));
    c->lr_X = c->lr_X_own.p;
    c->lr_y = (const uint8_t *)c->lr_y_own.p;
  }
  c->lr_n = n;
  c->lr_variant = variant;
  c->lr_likeadj = likeadj;
  c->lr_X_full = c->lr_X;
  c->lr_y_full = c->lr_y;
  c->lr_n_full = n;
  c->lr_likeadj_full = likeadj;
  c->t_const = 0.0;
  c->target = TGT_LOGREG;
  c->cap_M = 0;   // (re)allocate the transposed-sample buffer
  if (c->cfg.dtype == MIVI_F32 && !logreg_prepare_f32(c)) return fail(c, MIVI_ERR_HIP, "logistic regression: row-major copy allocation failed");
  c->lr_Xrm_act = c->cfg.dtype == MIVI_F32 ? c->lr_Xrm.p : nullptr;
  invalidate_graph(c);
  return MIVI_OK;
}

mivi_status_t mivi_logreg_select_rows(mivi_ctx_t *c, const int64_t *idx, int64_t b, double likeadj) {
  if (!c || b < 0 || (b > 0 && !idx)) return MIVI_ERR_BAD_ARG;
  if (c->target != TGT_LOGREG || !c->lr_X_full) return fail(c, MIVI_ERR_NO_TARGET, "no logistic-regression target set");
  (void)hipSetDevice(c->cfg.device);
  invalidate_graph(c);
  if (b == 0) {   // back to the full data set
    c->lr_X = c->lr_X_full;
    c->lr_y = c->lr_y_full;
    c->lr_n = c->lr_n_full;
    c->lr_likeadj = c->lr_likeadj_full;
    c->lr_Xrm_act = c->cfg.dtype == MIVI_F32 ? c->lr_Xrm.p : nullptr;
    return MIVI_OK;
  }
  if (!(likeadj > 0.0)) return fail(c, MIVI_ERR_BAD_ARG, "likelihood adjustment must be positive");
  for (int64_t j = 0; j < b; ++j)
    if (idx[j] < 0 || idx[j] >= c->lr_n_full) return fail(c, MIVI_ERR_BAD_ARG, "batch row index out of range");
  const int p = c->cfg.d - 1;
  mivi_status_t s;
  if ((s = ensure(c, c->lr_idx, (size_t)b * sizeof(int64_t), false)) ||
      (s = ensure(c, c->lr_Xsub, (size_t)b * p * c->esize, false)) || (s = ensure(c, c->lr_ysub, (size_t)b, false)))
    return s;
  if (c->cfg.dtype == MIVI_F32 && (s = ensure(c, c->lr_Xrm_sub, (size_t)((b + 15) / 16 * 16) * ((p + 31) / 32 * 32) * sizeof(float), false))) return s;
  // the previous estimate may still be reading the batch buffers: order the upload behind it
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(c->lr_idx.p, idx, (size_t)b * sizeof(int64_t), hipMemcpyHostToDevice));
  launch_logreg_gather(c, b);
  HIPCHK(c, hipGetLastError());
  c->lr_X = c->lr_Xsub.p;
  c->lr_y = (const uint8_t *)c->lr_ysub.p;
  c->lr_n = b;
  c->lr_likeadj = likeadj;
  c->lr_Xrm_act = c->cfg.dtype == MIVI_F32 ? c->lr_Xrm_sub.p : nullptr;
  return MIVI_OK;
}

mivi_status_t mivi_set_target_funnel_constrained(mivi_ctx_t *c, double sigma_v) {
  mivi_status_t s = mivi_set_target_funnel(c, sigma_v);
  if (s == MIVI_OK) c->funnel_constrained = 1;
  return s;
}

mivi_status_t mivi_set_bijector_stacked(mivi_ctx_t *c, int32_t n_blocks, const int32_t *ranges, const int32_t *kinds) {
  if (!c || n_blocks < 0 || (n_blocks > 0 && (!ranges || !kinds))) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  const int d = c->cfg.d;
  std::vector<uint8_t> mask(d, 0), seen(d, 0);
  bool any = false;
  for (int b = 0; b < n_blocks; ++b) {
    const int lo = ranges[2 * b], hi = ranges[2 * b + 1];
    if (lo < 0 || hi > d || lo > hi) return fail(c, MIVI_ERR_BAD_ARG, "stacked bijector: block range outside [0, d)");
    if (kinds[b] != 0 && kinds[b] != 1) return fail(c, MIVI_ERR_BAD_ARG, "stacked bijector: kind must be 0 (identity) or 1 (exp)");
    for (int i = lo; i < hi; ++i) {
      if (seen[i]) return fail(c, MIVI_ERR_BAD_ARG, "stacked bijector: blocks overlap");
      seen[i] = 1;
      mask[i] = (uint8_t)kinds[b];
      any = any || kinds[b] == 1;
    }
  }
  invalidate_graph(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));   // an estimate in flight may still read the old mask
  if (!any) {   // all-identity (or no) bijector: nothing to apply
    c->bij_on = false;
    return MIVI_OK;
  }
  mivi_status_t s = ensure(c, c->bij_mask, (size_t)d, false);
  if (s) return s;
  HIPCHK(c, hipMemcpy(c->bij_mask.p, mask.data(), (size_t)d, hipMemcpyHostToDevice));
  c->bij_on = true;
  c->cap_M = 0;   // (re)allocate the work buffers incl. the per-column log-Jacobian sums
  return MIVI_OK;
}

mivi_status_t mivi_set_target_funnel(mivi_ctx_t *c, double sigma_v) {
  if (!c || !(sigma_v > 0.0) || c->cfg.d < 2) return MIVI_ERR_BAD_ARG;
  c->funnel_constrained = 0;
  c->funnel_sigma_v = sigma_v;
  c->t_const = -log(sigma_v) - 0.5 * kLog2Pi - 0.5 * (c->cfg.d - 1) * kLog2Pi;
  c->target = TGT_FUNNEL;
  invalidate_graph(c);
  return MIVI_OK;
}

mivi_status_t mivi_set_target_callback(mivi_ctx_t *c, mivi_logdensity_and_gradient_fn fg, mivi_logdensity_fn fv,
                                       void *user) {
  if (!c || !fg) return MIVI_ERR_BAD_ARG;
  c->cb_grad = fg;
  c->cb_value = fv;
  c->cb_user = user;
  c->t_const = 0.0;
  c->target = TGT_CALLBACK;
  invalidate_graph(c);
  return MIVI_OK;
}

int32_t mivi_fullrank_route(const mivi_ctx_t *c, int32_t n_samples) {
  if (!c || c->cfg.family != MIVI_FULLRANK) return 0;
  if (n_samples <= 0) n_samples = c->cfg.n_mc;
  if (!lds_path_shape_ok(c, n_samples) || (c->target != TGT_DIAG_GAUSS && c->target != TGT_DENSE_GAUSS)) return 0;
  return (lds_use_prod32(c, n_samples) ? 1 : 3) | (lds_bf16x3() ? 16 : 0);
}

int32_t mivi_logreg_kernels(const mivi_ctx_t *c, int32_t n_samples) {
  if (!c) return 0;
  return logreg_kernel_bits(c, n_samples > 0 ? n_samples : c->cfg.n_mc);
}

mivi_status_t mivi_set_logreg_route(mivi_ctx_t *c, int32_t route) {
  if (!c || route < 0 || route > 2) return MIVI_ERR_BAD_ARG;
  c->lr_route = route;
  invalidate_graph(c);
  return MIVI_OK;
}

mivi_status_t mivi_set_index_source(mivi_ctx_t *c, const uint64_t *idx_dev) {
  if (!c) return MIVI_ERR_BAD_ARG;
  c->idx_src = idx_dev;
  invalidate_graph(c);
  return MIVI_OK;
}

mivi_status_t mivi_debug_timeline(mivi_ctx_t *c, void *buf) {
  if (!c) return MIVI_ERR_BAD_ARG;
#ifdef MIVI_DEV
  c->dbg = (long long *)buf;
  invalidate_graph(c);
  return MIVI_OK;
#else
  if (!buf) return MIVI_OK;
  return fail(c, MIVI_ERR_UNSUPPORTED, "timeline stamps are compiled out of the release library (build with `make DEV=1`)");
#endif
}

// ---------------------------------------------------------------------------------------------
// host-side RNG restatement
// ---------------------------------------------------------------------------------------------
void mivi_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  u32x4 c{ctr[0], ctr[1], ctr[2], ctr[3]};
  const u32x4 r = philox4x32_10(c, key[0], key[1]);
  out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

void mivi_eps_bits_host(uint64_t seed, uint64_t idx, int32_t d, int64_t m, int32_t i0, int32_t count, uint32_t *out) {
  const uint64_t d4 = (uint64_t)((d + 3) / 4);
  for (int32_t t = 0; t < count; ++t) {
    const int32_t i = i0 + t;
    const u32x4 b = eps_block_bits(seed, idx, (uint64_t)m * d4 + (uint64_t)(i / 4));
    const uint32_t w[4] = {b.x, b.y, b.z, b.w};
    out[t] = w[i & 3];
  }
}

void mivi_eps_host(uint64_t seed, uint64_t idx, int32_t d, int64_t m, int32_t i0, int32_t count, int32_t dtype,
                   double *out) {
  const uint64_t d4 = (uint64_t)((d + 3) / 4);
  for (int32_t t = 0; t < count; ++t) {
    const int32_t i = i0 + t;
    const uint64_t q = (uint64_t)m * d4 + (uint64_t)(i / 4);
    if (dtype == MIVI_F32) {
      float e[4];
      eps_block<float>(seed, idx, q, e);
      out[t] = (double)e[i & 3];
    } else {
      double e[4];
      eps_block<double>(seed, idx, q, e);
      out[t] = e[i & 3];
    }
  }
}

