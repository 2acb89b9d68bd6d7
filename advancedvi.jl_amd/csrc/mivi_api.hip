// libmivi C ABI (include/mivi.h): context management, target set-up and the estimate drivers that
// sequence the HIP kernels.  Reference call stack being replaced: SURVEY.md section 3.2
// (estimate_gradient! -> _value_and_gradient! -> AD of estimate_repgradelbo_ad_forward).
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <dlfcn.h>
#include <unistd.h>
#include <rccl/rccl.h>   // types only: the library is opened at run time (mivi_comm_init), libmivi has no link-time RCCL dependency

#include "mivi_internal.h"
#include "stl_dinv.h"

using namespace mivi;

#define HIPCHK(c, call)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      (c)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
      return MIVI_ERR_HIP;                                                                \
    }                                                                                     \
  } while (0)

static mivi_status_t fail(mivi_ctx *c, mivi_status_t s, const char *msg) {
  if (c) c->err = msg;
  return s;
}

static mivi_status_t ensure(mivi_ctx *c, DevBuf &b, size_t bytes, bool zero) {
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

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

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
static mivi_status_t ensure_work(mivi_ctx *c, int M) {
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

extern "C" {

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
                    &c->sc_part[0], &c->sc_part[1], &c->ld_part[0], &c->ld_part[1], &c->tabA, &c->tabB, &c->tabD, &c->stl_CT, &c->stl_Dinv, &c->stl_X, &c->stl_F, &c->lr_Xsub, &c->lr_ysub, &c->lr_Xrm_sub, &c->lr_idx, &c->dog_part, &c->stein_A, &c->stein_g, &c->bij_mask, &c->bij_ld, &c->dist_P, &c->dist_P2, &c->dist_ring[0], &c->dist_ring[1], &c->dist_ring[2], &c->dist_ring[3], &c->dist_ring[4], &c->dist_ring[5], &c->p2p_scratch, &c->dist_S, &c->dist_F, &c->p2p_tab, &c->p2p_ctr, &c->lds_tabV, &c->lds_tabV64, &c->lds_tabS, &c->lds_tabSt, &c->status, &c->d_idx, &c->acc, &c->tmp_params, &c->tmp_out};
  for (DevBuf *b : bufs)
    if (b->p) (void)hipFree(b->p);
  {
    DevBuf *fbb[] = {&c->fb.CA, &c->fb.epsP, &c->fb.epsV, &c->fb.WV, &c->fb.ell, &c->fb.he, &c->fb.ld, &c->fb.grads, &c->fb.values};
    for (DevBuf *b : fbb)
      if (b->p) (void)hipFree(b->p);
    for (auto &tb : c->fb.tab) {
      if (tb.prod.p) (void)hipFree(tb.prod.p);
      if (tb.vjp.p) (void)hipFree(tb.vjp.p);
    }

  }
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  if (c->cap_stream) (void)hipStreamDestroy(c->cap_stream);
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  if (c->comm_stream2) (void)hipStreamDestroy(c->comm_stream2);
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

static mivi_status_t read_status(mivi_ctx *c);

// waits for the context's stream and reports (then clears) the sticky device flags of the estimates since the last read
mivi_status_t mivi_synchronize(mivi_ctx_t *c) {
  if (!c) return MIVI_ERR_BAD_ARG;
  return read_status(c);
}

// ---------------------------------------------------------------------------------------------
// targets
// ---------------------------------------------------------------------------------------------
static double host_get(const void *p, int dtype, size_t i) {
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

// Chained mode (graph-batched estimates, device-resident optimisation loop): everything stays on ONE stream and
// the two pieces of an estimate that do not sit on its critical path ride along as heterogeneous workgroups:
//   * eps of estimate t+1 is generated by extra workgroups of estimate t's VJP kernel (VALU under the MFMA waves),
//   * the objective value of estimate t is assembled by one extra workgroup of estimate t+1's first kernel.
// Buffers those jobs touch are double-buffered by the parity c->cur.  The chain is closed by flush_chain().
struct Chain {
  bool on = false;
  bool first = true;        // eps of this estimate has not been generated yet
  bool has_next = false;    // another estimate follows: prefetch its eps
  RngArgs next_rng{};
  bool have_prev = false;   // a value job is pending
  bool estimates_only = false;   // no optimiser step between the estimates: a gradient entry may be finished one launch late
  ValueJob prev{};
};

static bool no_fused_update() {   // MIVI_NO_FUSED_UPDATE=1: separate update kernel in the graph loop (A/B reference)
  static const bool v = getenv("MIVI_NO_FUSED_UPDATE") != nullptr;
  return v;
}

static bool hetero_ok(const mivi_ctx *c, int want_grad, const Chain *ch = nullptr) {
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
static bool lds_route(const mivi_ctx *c, const void *params, int M, int want_grad, const OutArgs &out) {
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
static mivi_status_t run_estimate(mivi_ctx *c, const void *params, const RngArgs &rng, int M, int want_grad,
                                  OutArgs out, Chain *ch = nullptr, const FusedUpdate *upd = nullptr,
                                  bool stop_after_target = false) {
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

static void flush_chain(mivi_ctx *c, const void *params, Chain *ch) {
  if (ch->have_prev) launch_value_only(c, params, ch->prev.vin, ch->prev.out);
  ch->have_prev = false;
}

static OutArgs final_out(mivi_ctx *c, void *value, void *grad) {
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

static RngArgs rng_of(mivi_ctx *c, uint64_t idx) {
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

static mivi_status_t read_status(mivi_ctx *c) {
  // this context's sticky flags (word 0) and, with interleaved chains, the children's (words 1 .. n_kids of the same buffer: their kernels are
  // joined into this stream by the batch's graph) -- one copy, one wait
  int sk[1 + mivi_ctx::kMaxKids] = {};
  const int nw = 1 + c->n_kids;
  HIPCHK(c, hipMemcpyAsync(sk, c->status.p, sizeof(int) * nw, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  int st = 0;
  for (int j = 0; j < nw; ++j) st |= sk[j];
  if (st) HIPCHK(c, hipMemsetAsync(c->status.p, 0, sizeof(int) * nw, c->stream));
  if (st & 8) return fail(c, MIVI_ERR_HIP, "peer-to-peer exchange: a peer did not arrive within the spin budget (lost rank or unmapped buffer)");
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


// ---------------------------------------------------------------------------------------------
// Sharded finalisation and the collective behind the C ABI: reduce-scatter -> slice finalise -> all-gather ->
// unpack.  RCCL is opened with dlopen: a host without it (or a CPU-only symbol check) still loads libmivi.
// ---------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
void load_rccl(RcclApi &api) {
  const char *env = getenv("MIVI_RCCL_LIB");
  const char *cands[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (int pass = 0; pass < 2 && !api.lib; ++pass)       // pass 0: a copy the process already loaded (e.g. the host framework's)
    for (const char *n : cands)
      if (n && !api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
  if (!api.lib) return;
  api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
  api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
  api.ReduceScatter = (decltype(api.ReduceScatter))dlsym(api.lib, "ncclReduceScatter");
  api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
  api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
  api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.ReduceScatter || !api.AllGather) api.lib = nullptr;
}
// the fully built table behind a C++11 magic static: contexts initialised from different host threads see it complete or not at all
struct RcclOnce { RcclApi api; RcclOnce() { load_rccl(api); } };
RcclApi *rccl() {
  static RcclOnce once;
  return once.api.lib ? &once.api : nullptr;
}
long long slice_len_of(const mivi_ctx *c, int world) {
  const long long L = mivi_partials_len(c);
  return (L + world - 1) / world;
}
}  // namespace

int64_t mivi_slice_len(const mivi_ctx_t *c, int32_t world) { return (c && world > 0) ? slice_len_of(c, world) : 0; }

mivi_status_t mivi_finalize_slice(mivi_ctx_t *c, const void *params, const void *slice_sum, int32_t rank, int32_t world, void *final_slice) {
  if (!c || !params || !slice_sum || !final_slice || world <= 0 || rank < 0 || rank >= world) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  const long long n = slice_len_of(c, world);
  if (world > 1 && n < world + 2) return fail(c, MIVI_ERR_UNSUPPORTED, "parameter vector too short to shard over this many ranks");
  launch_finalize_slice(c, params, slice_sum, (long long)rank * n, n, final_slice);
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}

mivi_status_t mivi_unpack_final(mivi_ctx_t *c, const void *packed_final, void *value, void *grad) {
  if (!c || !packed_final || !value || !grad) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  launch_unpack_final(c, packed_final, value, grad);
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}

mivi_status_t mivi_comm_unique_id(void *id_host) {
  if (!id_host) return MIVI_ERR_BAD_ARG;
  RcclApi *r = rccl();
  if (!r) return MIVI_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (r->GetUniqueId(&id) != ncclSuccess) return MIVI_ERR_HIP;
  memcpy(id_host, &id, sizeof(id));
  return MIVI_OK;
}

// ---- peer-to-peer exchange buffers (kernels_p2p.hip) ---------------------------------------------------------------------------------
namespace {
struct P2PHandle {   // what travels between the ranks (MIVI_P2P_HANDLE_BYTES = 256 per rank)
  uint32_t magic, version;
  int32_t rank, world;
  int64_t L, n, cn;
  int32_t esize, G;
  uint64_t bytes, pid, local_ptr;
  int32_t device, pad;
  hipIpcMemHandle_t ipc;
};
static_assert(sizeof(P2PHandle) <= MIVI_P2P_HANDLE_BYTES, "handle blob");
constexpr uint32_t kP2PMagic = 0x4D495650u;   // "MIVP"
constexpr int kLanes = 1, kRing = 8, kGroup = 4;   // (kernels_p2p.hip: kP2PLanes, kP2PRing, kP2PGroup)
struct P2PTableHost { char *stage[kLanes][8]; char *fin[kLanes][8]; unsigned *arr[kLanes][8]; unsigned *farr[kLanes][8]; };   // == P2PTable (kernels_p2p.hip)

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// geometry of the exchange for (L, world): identical on every rank
void p2p_geometry(long long L, int R, long long &n, long long &cn, int &G, int &vs) {
  n = ((L + R - 1) / R + 3) & ~3LL;
  while ((L - 1) % n == 0) n += 4;        // the two scalars (L - 2, L - 1) must lie in ONE slice
  vs = (int)((L - 2) / n);
  // Small chunks, but at most 255 (+ the value workgroup = one workgroup per CU): the exchange kernel is persistent and spins beside the
  // compute chain.  More would be faster for an exchange on its own (system-scope accesses are limited per CU) but 513 spinning
  // workgroups held every CU's registers and the compute kernels could not be scheduled beside them at all (found on the GPU: the
  // hand-over timed out); measured in the pipelined batch (groups of four estimates, one lane): 127 -> 18.9, 191 -> 16.4, 255 -> 16.3,
  // 383 -> 17.6 us per estimate.
  long long g = (n + 511) / 512;
  G = (int)(g < 1 ? 1 : (g > 255 ? 255 : g));
  cn = ((n + G - 1) / G + 3) & ~3LL;
}
}  // namespace

// host-only: the geometry the exchange uses for a partial vector of length L over `world` ranks: out = {slice length n, chunk length cn,
// chunk workgroups G, value-owner rank}.  Identical on every rank by construction (tests/test_abi_and_host.py checks its invariants).
void mivi_p2p_geometry(int64_t L, int32_t world, int64_t *out4) {
  long long n, cn;
  int G, vs;
  p2p_geometry(L, world, n, cn, G, vs);
  out4[0] = n; out4[1] = cn; out4[2] = G; out4[3] = vs;
}

mivi_status_t mivi_p2p_detach(mivi_ctx_t *c) {
  if (!c) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  (void)hipStreamSynchronize(c->stream);
  if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
  if (c->comm_stream2) (void)hipStreamSynchronize(c->comm_stream2);
  invalidate_graph(c);
  for (int r = 0; r < 8; ++r) {
    if (c->p2p_opened[r] && c->p2p_peer[r]) (void)hipIpcCloseMemHandle(c->p2p_peer[r]);
    c->p2p_opened[r] = false;
    c->p2p_peer[r] = nullptr;
  }
  if (c->p2p_buf) (void)hipFree(c->p2p_buf);
  c->p2p_buf = nullptr;
  c->p2p_bytes = 0;
  c->p2p_on = false;
  return MIVI_OK;
}

mivi_status_t mivi_p2p_export(mivi_ctx_t *c, int32_t rank, int32_t world, void *handle_out) {
  if (!c || !handle_out || world < 1 || world > 8 || rank < 0 || rank >= world) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  (void)mivi_p2p_detach(c);
  const long long L = mivi_partials_len(c);
  long long n, cn;
  int G, vs;
  p2p_geometry(L, world, n, cn, G, vs);
  const size_t es = c->esize;
  // per lane (double-buffered by epoch parity; an epoch carries a group of kGroup estimates): staging [2][V][R][n] T, final [2][V][R n] T,
  // arrival flags [2][R][G], final flags [2][R][G + 1]
  const size_t b_stage = align256((size_t)2 * kGroup * world * n * es), b_fin = align256((size_t)2 * kGroup * world * n * es);
  const size_t b_arr = align256((size_t)2 * world * G * 4), b_farr = align256((size_t)2 * world * (G + 1) * 4);
  const size_t lane_bytes = b_stage + b_fin + b_arr + b_farr;
  const size_t bytes = lane_bytes * kLanes;
  // FINE-GRAINED device memory: peers write it over xGMI, system-scope releases / acquires and the consumers' system-scope loads
  // (kernels_p2p.hip ld_sys) keep it coherent.  NOT hipDeviceMallocUncached: on this stack (ROCm 7.0 / gfx950) running the exchange on an
  // uncached allocation corrupted UNRELATED buffers of later contexts once the area had been freed and its pages re-used (found on one
  // GPU: the estimates of contexts created after a p2p context were off by 1e-2 until their buffers had been rewritten a few times;
  // fine-grained and plain allocations never showed it).
  void *buf = nullptr;
  hipError_t e = hipExtMallocWithFlags(&buf, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) { (void)hipGetLastError(); return fail(c, MIVI_ERR_HIP, "peer-to-peer exchange buffer: fine-grained allocation failed"); }
  HIPCHK(c, hipMemset(buf, 0, bytes));
  HIPCHK(c, hipDeviceSynchronize());
  c->p2p_buf = buf;
  c->p2p_bytes = bytes;
  c->p2p_rank = rank; c->p2p_world = world; c->p2p_n = n; c->p2p_cn = cn; c->p2p_G = G; c->p2p_vs = vs;
  c->p2p_lane_bytes = lane_bytes; c->p2p_off_fin = b_stage; c->p2p_off_arr = b_stage + b_fin; c->p2p_off_farr = b_stage + b_fin + b_arr;
  P2PHandle h{};
  h.magic = kP2PMagic; h.version = 3; h.rank = rank; h.world = world; h.L = L; h.n = n; h.cn = cn; h.esize = (int32_t)es; h.G = G;
  h.bytes = bytes; h.pid = (uint64_t)getpid(); h.local_ptr = (uint64_t)(uintptr_t)buf; h.device = c->cfg.device;
  if (hipIpcGetMemHandle(&h.ipc, buf) != hipSuccess) {   // single-process use (tests, world = 1) still works through local_ptr
    (void)hipGetLastError();
    memset(&h.ipc, 0, sizeof(h.ipc));
    h.pad = 1;   // no IPC handle: other processes cannot attach
  }
  memset(handle_out, 0, MIVI_P2P_HANDLE_BYTES);
  memcpy(handle_out, &h, sizeof(h));
  return MIVI_OK;
}

mivi_status_t mivi_p2p_attach(mivi_ctx_t *c, const void *handles) {
  if (!c || !handles) return MIVI_ERR_BAD_ARG;
  if (!c->p2p_buf) return fail(c, MIVI_ERR_BAD_ARG, "mivi_p2p_export has not been called");
  (void)hipSetDevice(c->cfg.device);
  const int R = c->p2p_world;
  const uint64_t me = (uint64_t)getpid();
  P2PTableHost tab{};
  for (int r = 0; r < R; ++r) {
    P2PHandle h;
    memcpy(&h, (const char *)handles + (size_t)r * MIVI_P2P_HANDLE_BYTES, sizeof(h));
    if (h.magic != kP2PMagic || h.version != 3 || h.rank != r || h.world != R || h.L != mivi_partials_len(c) || h.n != c->p2p_n ||
        h.cn != c->p2p_cn || h.G != c->p2p_G || h.esize != (int32_t)c->esize || h.bytes != c->p2p_bytes)
      return fail(c, MIVI_ERR_BAD_ARG, "peer-to-peer handle does not match this context (rank order, family, d, dtype or world differ)");
    void *base = nullptr;
    if (r == c->p2p_rank) {
      base = c->p2p_buf;
    } else if (h.pid == me) {   // another context of this process: its pointer is valid here (peer access for another device)
      base = (void *)(uintptr_t)h.local_ptr;
      if (h.device != c->cfg.device) {
        const hipError_t e = hipDeviceEnablePeerAccess(h.device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return fail(c, MIVI_ERR_HIP, "hipDeviceEnablePeerAccess failed"); }
        (void)hipGetLastError();
      }
    } else {
      if (h.pad) return fail(c, MIVI_ERR_HIP, "peer exported no IPC handle (hipIpcGetMemHandle failed there)");
      if (hipIpcOpenMemHandle(&base, h.ipc, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        (void)hipGetLastError();
        return fail(c, MIVI_ERR_HIP, "hipIpcOpenMemHandle failed (is HSA_ENABLE_IPC_MODE_LEGACY=0 exported?)");
      }
      c->p2p_opened[r] = true;
    }
    c->p2p_peer[r] = base;
    for (int ln = 0; ln < kLanes; ++ln) {
      char *lb = (char *)base + (size_t)ln * c->p2p_lane_bytes;
      tab.stage[ln][r] = lb;
      tab.fin[ln][r] = lb + c->p2p_off_fin;
      tab.arr[ln][r] = (unsigned *)(lb + c->p2p_off_arr);
      tab.farr[ln][r] = (unsigned *)(lb + c->p2p_off_farr);
    }
  }
  mivi_status_t s;
  if ((s = ensure(c, c->p2p_tab, sizeof(tab), false)) || (s = ensure(c, c->p2p_ctr, 512, false)) ||
      (s = ensure(c, c->p2p_scratch, ((size_t)mivi_params_len(c) + 4) * kGroup * kLanes * c->esize, false)))
    return s;
  HIPCHK(c, hipMemcpy(c->p2p_tab.p, &tab, sizeof(tab), hipMemcpyHostToDevice));
  // (stream-ordered on the context's stream and waited for: a null-stream memset is NOT ordered against a non-blocking stream and
  //  would zero the epoch counter after the first exchange has advanced it)
  HIPCHK(c, hipMemsetAsync(c->p2p_ctr.p, 0, 512, c->stream));   // [lane] {epoch, ticket} at 64-byte spacing, ready at byte 256, freed[ring] at byte 320
  HIPCHK(c, hipStreamSynchronize(c->stream));
  invalidate_graph(c);
  c->p2p_on = true;
  return MIVI_OK;
}

mivi_status_t mivi_p2p_debug_words(mivi_ctx_t *c, uint32_t *out128) {   // developer: the exchange's device words (lane epochs, ready, freed)
  if (!c || !out128 || !c->p2p_ctr.p) return MIVI_ERR_BAD_ARG;
  HIPCHK(c, hipMemcpy(out128, c->p2p_ctr.p, 512, hipMemcpyDeviceToHost));
  return MIVI_OK;
}

mivi_status_t mivi_p2p_set_pipeline(mivi_ctx_t *c, int32_t on) {
  if (!c) return MIVI_ERR_BAD_ARG;
  // (a second persistent exchange kernel serving every other group was an option until the groups: two of them are 512 resident
  //  workgroups, which starve the compute chain of registers, and one measured better wherever both ran)
  if (on < 0 || on > 1) return fail(c, MIVI_ERR_BAD_ARG, "mivi_p2p_set_pipeline: 0 = off (serial steps), 1 = the persistent exchange kernel beside the compute chain");
  c->p2p_pipe_state = on ? 1 : -1;
  invalidate_graph(c);
  return MIVI_OK;
}

mivi_status_t mivi_p2p_set_spin_budget(mivi_ctx_t *c, int32_t polls) {
  if (!c || polls < 16) return MIVI_ERR_BAD_ARG;
  c->p2p_spin = polls;
  invalidate_graph(c);
  return MIVI_OK;
}

mivi_status_t mivi_comm_set_route(mivi_ctx_t *c, int32_t route) {
  if (!c || route < 0 || route > 3) return MIVI_ERR_BAD_ARG;
  if (route == 3 && !c->p2p_on) return fail(c, MIVI_ERR_UNSUPPORTED, "peer-to-peer route: no exchange buffers attached (mivi_p2p_export / mivi_p2p_attach)");
  c->dist_route = route;
  invalidate_graph(c);
  return MIVI_OK;
}

int32_t mivi_comm_route(const mivi_ctx_t *c) {   // the route the next sharded estimate takes: 1 all-reduce, 2 reduce-scatter/all-gather, 3 peer-to-peer, 0 none (one rank, no communicator)
  if (!c) return 0;
  int r = c->dist_route;
  if (r == 0) r = c->p2p_on ? 3 : (((size_t)mivi_partials_len(c) * c->esize >= ((size_t)16 << 20)) ? 2 : 1);
  if (r == 3 && !c->p2p_on) r = 1;
  if ((r == 1 || r == 2) && !c->comm) return c->comm_world > 1 ? r : 0;
  return r;
}

mivi_status_t mivi_comm_destroy(mivi_ctx_t *c) {
  if (!c) return MIVI_ERR_BAD_ARG;
  (void)mivi_p2p_detach(c);
  if (c->comm) {
    RcclApi *r = rccl();
    (void)hipStreamSynchronize(c->stream);
    if (r) (void)r->CommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr;
  }
  c->comm_rank = 0;
  c->comm_world = 1;
  return MIVI_OK;
}

mivi_status_t mivi_comm_init(mivi_ctx_t *c, const void *id_host, int32_t rank, int32_t world) {
  if (!c || world <= 0 || rank < 0 || rank >= world || (world > 1 && !id_host)) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  (void)mivi_comm_destroy(c);
  invalidate_graph(c);
  if (id_host) {   // also for world == 1: exercises the collective path on one GPU
    RcclApi *r = rccl();
    if (!r) return fail(c, MIVI_ERR_UNSUPPORTED, "librccl could not be opened (set MIVI_RCCL_LIB)");
    ncclUniqueId id;
    memcpy(&id, id_host, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t e = r->CommInitRank(&comm, world, id, rank);
    if (e != ncclSuccess) {
      c->err = std::string("ncclCommInitRank: ") + (r->GetErrorString ? r->GetErrorString(e) : "error");
      return MIVI_ERR_HIP;
    }
    c->comm = comm;
  }
  c->comm_rank = rank;
  c->comm_world = world;
  return MIVI_OK;
}

// Exchange the peer-to-peer handles through the RCCL communicator itself (hosts without another channel: julia/MIVI.jl) and attach.
// A failure leaves the context on the RCCL routes (the reason is in mivi_last_error).
mivi_status_t mivi_comm_enable_p2p(mivi_ctx_t *c) {
  if (!c) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  const int R = c->comm_world;
  if (R > 8) return fail(c, MIVI_ERR_UNSUPPORTED, "peer-to-peer exchange: at most 8 ranks (one xGMI node)");
  if (R > 1 && !c->comm) return fail(c, MIVI_ERR_BAD_ARG, "mivi_comm_init has not been called");
  std::vector<char> all((size_t)R * MIVI_P2P_HANDLE_BYTES);
  mivi_status_t s = mivi_p2p_export(c, c->comm_rank, R, all.data() + (size_t)c->comm_rank * MIVI_P2P_HANDLE_BYTES);
  if (s) return s;
  if (R > 1) {
    RcclApi *r = rccl();
    DevBuf tmp;
    if ((s = ensure(c, tmp, all.size(), false))) return s;
    HIPCHK(c, hipMemcpy((char *)tmp.p + (size_t)c->comm_rank * MIVI_P2P_HANDLE_BYTES, all.data() + (size_t)c->comm_rank * MIVI_P2P_HANDLE_BYTES,
                        MIVI_P2P_HANDLE_BYTES, hipMemcpyHostToDevice));
    const ncclResult_t e = r->AllGather((char *)tmp.p + (size_t)c->comm_rank * MIVI_P2P_HANDLE_BYTES, tmp.p, MIVI_P2P_HANDLE_BYTES, ncclChar,
                                        (ncclComm_t)c->comm, c->stream);
    if (e != ncclSuccess) { (void)hipFree(tmp.p); (void)mivi_p2p_detach(c); return fail(c, MIVI_ERR_HIP, "ncclAllGather of the peer-to-peer handles failed"); }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(all.data(), tmp.p, all.size(), hipMemcpyDeviceToHost));
    (void)hipFree(tmp.p);
  }
  s = mivi_p2p_attach(c, all.data());
  if (s) { const std::string why = c->err; (void)mivi_p2p_detach(c); c->err = why; }
  return s;
}

// buffers of the sharded estimate: padded partial vectors (two: the pipelined batch double-buffers them), slice sum, packed final
static mivi_status_t ensure_dist(mivi_ctx *c) {
  const int R = c->comm_world > c->p2p_world ? c->comm_world : c->p2p_world;
  const long long n = slice_len_of(c, c->comm_world), Lp = n * c->comm_world;
  long long need = Lp;
  if (c->p2p_on && c->p2p_n * c->p2p_world > need) need = c->p2p_n * c->p2p_world;
  (void)R;
  const size_t es = c->esize;
  const size_t need34 = c->p2p_on ? (size_t)need * es : 0;
  bool ring_short = false;
  for (int k = 0; k < 6; ++k) ring_short = ring_short || c->dist_ring[k].bytes < need34;
  if (c->dist_P.bytes < (size_t)need * es || c->dist_P2.bytes < (size_t)need * es || ring_short ||
      c->dist_S.bytes < (size_t)n * es || c->dist_F.bytes < (size_t)Lp * es) {
    invalidate_graph(c);
    mivi_status_t s;
    c->dist_P.bytes = 0; c->dist_P2.bytes = 0;   // (re-zero: the padding behind the partial vector must be 0)
    for (int k = 0; k < 6; ++k) {
      c->dist_ring[k].bytes = 0;
      if (need34 && (s = ensure(c, c->dist_ring[k], need34, true))) return s;
    }
    if ((s = ensure(c, c->dist_P, (size_t)need * es, true)) || (s = ensure(c, c->dist_P2, (size_t)need * es, true)) ||
        (s = ensure(c, c->dist_S, (size_t)n * es, true)) || (s = ensure(c, c->dist_F, (size_t)Lp * es, true)))
      return s;
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return MIVI_OK;
}

// The exchange + finalisation of ONE estimate whose partial vector sits in P, on c->stream: value / gradient on every rank.
static mivi_status_t dist_collective(mivi_ctx *c, const void *params, void *P, void *value, void *grad) {
  const int R = c->comm_world, rank = c->comm_rank;
  const size_t es = c->esize;
  const int route = mivi_comm_route(c);
  if (route == 3) {
    const void *Ps[1] = {P};
    launch_p2p_exchange(c, params, Ps, 1, value, grad, 7, 0, 1, 1, nullptr, nullptr);
    HIPCHK(c, hipGetLastError());
    return MIVI_OK;
  }
  const long long n = slice_len_of(c, R);
  // Route: two collectives cost one more launch + rendezvous than one; below 16 MB of partials the step is latency
  // bound and ONE all-reduce + the whole finalisation on every rank is the faster form (mivi_comm_set_route pins it).
  bool rsag = route == 2 || !c->comm;   // (one rank without a communicator: the slice kernels without the collectives)
  if (c->comm && !rsag && !rccl()->AllReduce) rsag = true;   // (a librccl without ncclAllReduce: the two-collective route needs only the required symbols)
  // the slice route gives every rank n >= world + 2 elements (the two trailing scalars must lie in the last slice); short parameter
  // vectors are exactly the ones the single all-reduce serves, so fall back to it instead of refusing
  if (rsag && R > 1 && n < R + 2) {
    if (c->comm && rccl()->AllReduce) rsag = false;
    else return fail(c, MIVI_ERR_UNSUPPORTED, "parameter vector too short to shard over this many ranks");
  }
  if (c->comm && !rsag) {
    RcclApi *r = rccl();
    const ncclDataType_t dt = c->cfg.dtype == MIVI_F32 ? ncclFloat : ncclDouble;
    if (r->AllReduce(P, P, (size_t)mivi_partials_len(c), dt, ncclSum, (ncclComm_t)c->comm, c->stream) != ncclSuccess)
      return fail(c, MIVI_ERR_HIP, "ncclAllReduce failed");
    launch_finalize(c, params, P, value, grad);
    HIPCHK(c, hipGetLastError());
    return MIVI_OK;
  }
  const void *sum = (const char *)P + (size_t)rank * n * es;   // one rank: its "slice" is the whole vector
  if (c->comm) {
    RcclApi *r = rccl();
    const ncclDataType_t dt = c->cfg.dtype == MIVI_F32 ? ncclFloat : ncclDouble;
    if (r->ReduceScatter(P, c->dist_S.p, (size_t)n, dt, ncclSum, (ncclComm_t)c->comm, c->stream) != ncclSuccess)
      return fail(c, MIVI_ERR_HIP, "ncclReduceScatter failed");
    sum = c->dist_S.p;
  }
  char *fin_slice = (char *)c->dist_F.p + (size_t)rank * n * es;
  launch_finalize_slice(c, params, sum, (long long)rank * n, n, fin_slice);
  if (c->comm) {
    RcclApi *r = rccl();
    const ncclDataType_t dt = c->cfg.dtype == MIVI_F32 ? ncclFloat : ncclDouble;
    if (r->AllGather(fin_slice, c->dist_F.p, (size_t)n, dt, (ncclComm_t)c->comm, c->stream) != ncclSuccess)
      return fail(c, MIVI_ERR_HIP, "ncclAllGather failed");
  }
  launch_unpack_final(c, c->dist_F.p, value, grad);
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}

static mivi_status_t dist_check(mivi_ctx *c) {
  if (c->comm_world > 1 && !c->comm && !c->p2p_on) return fail(c, MIVI_ERR_BAD_ARG, "mivi_comm_init has not been called");
  if (c->p2p_on && (c->p2p_world != c->comm_world || c->p2p_rank != c->comm_rank) && (c->comm || c->comm_world > 1))
    return fail(c, MIVI_ERR_BAD_ARG, "peer-to-peer buffers were exported for another rank / world than the communicator's");
  return MIVI_OK;
}

// estimate_gradient! of ONE estimate whose n_mc * world samples are sharded over the ranks (this context draws columns
// [m_offset, m_offset + n_mc) of m_total): partials -> exchange -> finalisation, all on the context's stream.
mivi_status_t mivi_estimate_gradient_dist(mivi_ctx_t *c, const void *params, uint64_t idx, void *value, void *grad) {
  if (!c || !params || !value || !grad) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  if (c->p2p_on && !c->comm) { c->comm_world = c->p2p_world; c->comm_rank = c->p2p_rank; }   // (peer-to-peer without RCCL)
  mivi_status_t s;
  if ((s = dist_check(c)) || (s = ensure_dist(c))) return s;
  if ((s = mivi_estimate_partials(c, params, idx, c->dist_P.p))) return s;
  return dist_collective(c, params, c->dist_P.p, value, grad);
}

// weighted accumulation of chunk objective values
__global__ void k_acc_value_f32(double *acc, const float *v, double w, int first) { acc[0] = (first ? 0.0 : acc[0]) + w * (double)v[0]; }
__global__ void k_acc_value_f64(double *acc, const double *v, double w, int first) { acc[0] = (first ? 0.0 : acc[0]) + w * v[0]; }
__global__ void k_neg_value_f32(float *out, const double *elbo) { out[0] = (float)(-elbo[0]); }
__global__ void k_neg_value_f64(double *out, const double *elbo) { out[0] = -elbo[0]; }
__global__ void k_store_value_f32(float *out, const double *acc) { out[0] = (float)acc[0]; }
__global__ void k_store_value_f64(double *out, const double *acc) { out[0] = acc[0]; }
// device counters of the graph-batched calls, set BY VALUE (an async copy from a stack local may outlive the caller's frame)
__global__ void k_set_u64x2(uint64_t *dst, uint64_t a, uint64_t b, int n) { dst[0] = a; if (n > 1) dst[1] = b; }
__global__ void k_bump_u64(uint64_t *dst, uint64_t by) { dst[0] += by; }
// latency floor (mivi_profile_kernel which = 9): a kernel that does nothing, launched with the grid / block / LDS footprint of a real one
__global__ void k_empty(int *sink) {
  extern __shared__ int lds_dyn[];
  if (sink && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) sink[0] = lds_dyn[0];
}

mivi_status_t mivi_estimate_objective(mivi_ctx_t *c, const void *params, uint64_t idx, int32_t n_samples, int32_t entropy,
                                      void *value) {
  if (!c || !params || !value) return MIVI_ERR_BAD_ARG;
  if (n_samples <= 0) n_samples = c->cfg.n_mc;
  if (entropy < 0) entropy = c->cfg.entropy;
  if (entropy > MIVI_ENT_STL_ZERO_GRAD) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  const int CH = 16384;
  // all estimators share their *value* within {closed-form} / {MC, STL, STL-zero-grad}
  if (n_samples <= CH) {
    OutArgs o = final_out(c, value, nullptr);
    o.ent_kind = entropy;
    o.M_total = n_samples;
    return run_estimate(c, params, rng_of(c, idx), n_samples, 0, o);
  }
  // chunked: the objective is a mean over samples plus parameter-only terms, so the weighted mean of the
  // chunk objectives is the full objective
  char *tmpv = (char *)c->tmp_out.p;
  for (int off = 0, first = 1; off < n_samples; off += CH, first = 0) {
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
  const bool builtin = (c->target == TGT_DIAG_GAUSS || c->target == TGT_DENSE_GAUSS) && !c->bij_on;
  const bool plugin = c->target == TGT_CALLBACK && c->cb_hess && !c->bij_on;
  if (!builtin && !plugin)
    return fail(c, MIVI_ERR_UNSUPPORTED, "second-order branch: the target has no Hessian here (built-in Gaussian targets, or a plugin with "
                                         "mivi_set_target_hess_callback, no bijector); use mivi_gauss_expected_grad_hess (Stein identity)");
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
  }
  launch_stein_finish(c, (double)n_samples, (const double *)c->stein_g.p, (const double *)c->acc.p, single_chunk ? part : nullptr, grad, logpi_avg);
  launch_const_hess(c, hess);
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

// ---------------------------------------------------------------------------------------------
// hipGraph-batched estimates and the device-resident optimisation loop
// ---------------------------------------------------------------------------------------------
// The null stream cannot be captured: record on an internal stream, replay on the context's stream.
static mivi_status_t begin_capture(mivi_ctx *c, hipStream_t *saved) {
  if (!c->cap_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking));

  HIPCHK(c, hipStreamSynchronize(c->stream));   // pending memsets / uploads on the launch stream
  *saved = c->stream;
  c->stream = c->cap_stream;
  hipError_t e = hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) { c->stream = *saved; c->err = std::string("hipStreamBeginCapture: ") + hipGetErrorString(e); return MIVI_ERR_HIP; }
  return MIVI_OK;
}
static hipError_t end_capture(mivi_ctx *c, hipStream_t saved, hipGraph_t *graph) {
  hipError_t e = hipStreamEndCapture(c->cap_stream, graph);
  c->stream = saved;
  return e;
}

static bool graph_capturable(const mivi_ctx *c) {   // every device-resident target (the host callback is not)
  return c->target == TGT_DIAG_GAUSS || c->target == TGT_DENSE_GAUSS || c->target == TGT_FUNNEL || c->target == TGT_LOGREG;
}
// allocations are not allowed inside a capture: size whatever the target's launchers would otherwise grow lazily
static mivi_status_t reserve_target(mivi_ctx *c, int M) {
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
                       (void *)(rec + count + hist_doubles + 8));
    if (c->cfg.dtype == MIVI_F32) hipLaunchKernelGGL(k_neg_value_f32, dim3(1), dim3(1), 0, c->stream, (float *)value, rec + count - 1);
    else hipLaunchKernelGGL(k_neg_value_f64, dim3(1), dim3(1), 0, c->stream, (double *)value, rec + count - 1);
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
    static const bool no_e0 = getenv("MIVI_FUNNEL_NO_E0TAB") != nullptr;   // (A/B: every thread re-derives eps[0, m])
    launch_mf_funnel_loop(c, params, idx0, count, hist, elbo, scratch, value, grad, (void *)((char *)scratch + sc_bytes),
                          no_e0 ? nullptr : (void *)((char *)scratch + sc_bytes + lane_al));
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
static mivi_status_t sync_kid(mivi_ctx *c, mivi_ctx *k, int lanes) {
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
static mivi_status_t ensure_kids(mivi_ctx *c, int lanes) {
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
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_join[j], hipEventDisableTiming));
    if (!c->ev_fork) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    c->kids[c->n_kids++] = k;
  }
  return MIVI_OK;
}

// ---- third-generation batch engine (kernels_fullrank_batch.hip) -------------------------------------------------------------------------
// `count` estimates at the same parameters as steps of up to fb_lanes_max() LANES: a step is four launches (eps, product + target, VJP,
// values) that cover all of its lanes.  No child contexts, no forked graph: a lane's buffers are base + lane * stride.
static int fb_lanes_max() {   // MIVI_FB_LANES: estimates per step (A/B; default 128: the triangular product is paced by its heaviest tile, 36 us for ANY lane count up to ~40, so long batches take few, wide steps; 100-estimate batches: 4.26 us per estimate with 25-lane steps, 3.92 with one step)
  static const int v = getenv("MIVI_FB_LANES") ? atoi(getenv("MIVI_FB_LANES")) : 128;
  return v < 1 ? 1 : (v > 256 ? 256 : v);
}
static bool fb_route(const mivi_ctx *c, const void *params, const void *grad_last, const void *grads_all) {
  return !c->is_child && c->cfg.family == MIVI_FULLRANK && c->cfg.dtype == MIVI_F32 && !c->bij_on && !c->idx_src && !c->dbg &&
         c->target == TGT_DIAG_GAUSS && c->cfg.entropy != MIVI_ENT_STL && c->cfg.entropy != MIVI_ENT_STL_ZERO_GRAD &&
         fb_shape_ok(c, c->cfg.n_mc) && ((uintptr_t)params & 15) == 0 && ((uintptr_t)grad_last & 15) == 0 && ((uintptr_t)grads_all & 15) == 0;
}
// value_last / grad_last: the batch's LAST estimate (mivi_estimate_gradient_n's contract), or nullptr; values_all T[count] / grads_all
// T[count * params_len]: every estimate's (mivi_estimate_gradient_each), or nullptr (lane scratch)
static mivi_status_t fb_batch(mivi_ctx *c, const void *params, uint64_t idx0, int count, void *value_last, void *grad_last, void *values_all,
                              void *grads_all) {
  mivi_status_t s;
  const int M = c->cfg.n_mc, d = c->cfg.d;
  if ((s = ensure_work(c, M))) return s;
  const int Lmax = fb_lanes_max();
  const int steps = (count + Lmax - 1) / Lmax, L = (count + steps - 1) / steps, Llast = count - (steps - 1) * L;
  const size_t plen = (size_t)mivi_params_len(c);
  FbTables &t = c->fb;
  if (t.cap_L < L || t.cap_M != M) {
    invalidate_graph(c);
    const size_t pw = fb_plane_words(c, M) * 4;
    if ((s = ensure(c, t.CA, fb_cplane_words(c) * 4, false)) || (s = ensure(c, t.epsP, (size_t)L * pw, false)) ||
        (s = ensure(c, t.epsV, (size_t)L * pw, false)) || (s = ensure(c, t.WV, (size_t)L * pw, false)) ||
        (s = ensure(c, t.ell, (size_t)L * (d / 32) * (M / 32) * sizeof(double), false)) ||
        (s = ensure(c, t.he, (size_t)L * (d / 64) * (M / 32) * sizeof(double), false)) ||
        (s = ensure(c, t.ld, 2 * (size_t)(d / 32) * sizeof(double) + 64, false)) || (s = ensure(c, t.values, (size_t)L * 4 + 64, false)))
      return s;
    t.grads.bytes = 0;   // (re-zeroed: the lanes' scratch gradients rely on exact zeros above the diagonal that no kernel writes)
    if ((s = ensure(c, t.grads, (size_t)L * plen * 4, true))) return s;
    t.cap_L = L;
    t.cap_M = M;
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
    fs.rng = rng_of(c, idx0 + (uint64_t)st * L);
    if (grads_all) { fs.grads = (char *)grads_all + (size_t)st * L * plen * 4; fs.grad_stride = (long long)plen; fs.write_upper = 1; }
    else { fs.grads = t.grads.p; fs.grad_stride = (long long)plen; fs.write_upper = 0; }
    if (values_all) { fs.values = (char *)values_all + (size_t)st * L * 4; fs.value_stride = 1; }
    else { fs.values = t.values.p; fs.value_stride = 1; }
    fs.lane_last = -1;
    if (st == steps - 1 && (value_last || grad_last)) { fs.lane_last = Llast - 1; fs.grad_last = grad_last; fs.value_last = value_last; }
    return fs;
  };
  // One stream, no graph: per step {draws (+ tril(C)'s planes as riders of the first) -> product -> VJP + values} = three launches for up to
  // fb_lanes_max() estimates; the host is far ahead of the device.  (Measured and dropped: the batch as ONE hipGraph -- 4.45 against 4.26 us
  // per estimate in 100-estimate batches -- and the draws of step s + 1 on a second graph branch beside the products of step s: the draws
  // are bound by their 3 MB of plane writes per estimate and by the vector ALU, beside them the products ran 25 % longer: 4.58 us.)
  for (int st = 0; st < steps; ++st) {
    const FbStep fs = make_step(st);
    fb_launch_eps(c, fs, st == 0, c->stream);
    fb_launch_compute(c, fs, c->stream);
  }
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}

// Roofline leg of the batch engine: `reps` launches of each of a step's three kernels for `lanes` estimates, hipEvents on the context's
// stream.  us_out[0..2] = average launch duration (us) of the draws, the product + target, the VJP (+ values).
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
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0));
  HIPCHK(c, hipEventCreate(&e1));
  for (int which = 0; which < 3; ++which) {
    for (int r = -2; r < reps; ++r) {
      if (r == 0) HIPCHK(c, hipEventRecord(e0, c->stream));
      if (which == 0) fb_launch_eps(c, fs, true, c->stream);
      else fb_launch_compute(c, fs, c->stream, which == 1 ? 1 : 2);
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

// ---------------------------------------------------------------------------------------------
// Sharded estimates in batches: the exchange of estimate t overlapped with the kernels of estimate t + 1
// ---------------------------------------------------------------------------------------------
// Estimates at fixed parameters are independent (what mivi_estimate_gradient_n serves on one GPU), so the exchange + finalisation of
// estimate t (comm_stream) runs UNDER the partial kernels of estimate t + 1 (the context's stream): partial vectors are double
// buffered, estimate t + 2 waits for the exchange of t to release its buffer.  One hipGraph with two branches per estimate; if the
// capture is refused (a collective that cannot be captured) the same sequence is issued eagerly with events.
//   mode 0 pipelined | 1 serial {partials -> exchange} on one stream | 2 partials only | 3 exchange only (on the last partial vector)
// mode 4 with LANE-BATCHED compute (second-generation full-rank kernels, no STL solve): four contexts compute four consecutive estimates
// with ONE product launch and ONE VJP launch (blockIdx.y = lane, each lane's packed partials into its ring slot), ONE hand-over per four
// estimates -- the exchange kernel serves them as one group (kernels_p2p.hip).
static mivi_status_t dist_sequence_lanes(mivi_ctx *c, const void *params, bool counter_idx, uint64_t idx0, int count) {
  constexpr int E = kGroup;
  mivi_status_t s = MIVI_OK;
  void *ringP[kRing] = {c->dist_P.p, c->dist_P2.p, c->dist_ring[0].p, c->dist_ring[1].p, c->dist_ring[2].p, c->dist_ring[3].p, c->dist_ring[4].p, c->dist_ring[5].p};
  mivi_ctx *ctxs[E];
  hipStream_t kept[E];
  ctxs[0] = c;
  for (int l = 1; l < E; ++l) ctxs[l] = c->kids[l - 1];
  LaneSink *sink = lane_sinks_alloc(E);
  EpsSink *esink = eps_sink_alloc();
  const bool dense = c->target == TGT_DENSE_GAUSS;
  for (int l = 0; l < E; ++l) { kept[l] = ctxs[l]->stream; ctxs[l]->stream = c->stream; ctxs[l]->lane_sink = sink; ctxs[l]->lane_id = l; ctxs[l]->eps_sink = esink; }
  unsigned *w = (unsigned *)c->p2p_ctr.p;
  for (int s0 = 0; s0 < count && s == MIVI_OK; s0 += E) {
    const int L = count - s0 < E ? count - s0 : E;
    eps_sink_reset(esink);
    for (int l = 0; l < L && s == MIVI_OK; ++l) {
      const int i = s0 + l;
      mivi_ctx *k = ctxs[l];
      lane_sink_reset(sink, l);
      RngArgs r = rng_of(k, counter_idx ? (uint64_t)i : idx0 + (uint64_t)i);
      if (counter_idx) r.idx_ptr = (const uint64_t *)c->d_idx.p;
      OutArgs o = final_out(k, nullptr, nullptr);
      o.partials = ringP[i % kRing];
      o.partials_mode = 1;
      o.scalars_off = mivi_partials_len(c) - 2;
      if ((s = run_estimate(k, params, r, k->cfg.n_mc, 1, o))) { c->err = k->err; break; }
      if (lane_sink_counts(sink, l) != (dense ? 2 : 1) * 16 + 1) s = fail(c, MIVI_ERR_HIP, "lane-batched sharded estimates: an estimate did not take the two-kernel route");
    }
    if (s == MIVI_OK) launch_lanes_eps(c, esink, L);
    if (s == MIVI_OK && !(launch_lanes_prod(c, sink, L, 0) && (!dense || launch_lanes_prod(c, sink, L, 1)) && launch_lanes_vjp(c, sink, L)))
      s = fail(c, MIVI_ERR_HIP, "lane-batched sharded estimates: the lanes' launches do not match");
    if (s) break;
    // announce the group's partial vectors; hold the chain until the exchange has read the ring slots the NEXT group overwrites
    const unsigned *fr[E];
    unsigned fmin[E];
    int nf = 0;
    for (int l = 0; l < E; ++l) {
      const int nx = s0 + E + l;
      if (nx >= count) break;
      const int prev_users = nx / kRing;
      if (prev_users >= 1) { fr[nf] = w + 80 + nx % kRing; fmin[nf] = (unsigned)prev_users * (unsigned)c->p2p_G; ++nf; }
    }
    launch_p2p_handover4(c, w + 64, (unsigned)(s0 + L), fr, fmin, nf);
  }
  for (int l = 0; l < E; ++l) {
    ctxs[l]->lane_sink = nullptr;
    ctxs[l]->eps_sink = nullptr;
    ctxs[l]->stream = kept[l];
    ctxs[l]->cur = 0;
    ctxs[l]->pre_valid = false;
  }
  lane_sinks_free(sink);
  eps_sink_free(esink);
  return s;
}

static mivi_status_t dist_sequence(mivi_ctx *c, const void *params, bool counter_idx, uint64_t idx0, int count, void *value, void *grad, int mode) {
  if (mode == 4 && c->dist_lane4) return dist_sequence_lanes(c, params, counter_idx, idx0, count);
  mivi_status_t s = MIVI_OK;
  hipStream_t main = c->stream;
  for (int i = 0; i < count && s == MIVI_OK; ++i) {
    const int par = i & 1;
    void *ringP[kRing] = {c->dist_P.p, c->dist_P2.p, c->dist_ring[0].p, c->dist_ring[1].p, c->dist_ring[2].p, c->dist_ring[3].p, c->dist_ring[4].p, c->dist_ring[5].p};
    void *P = mode == 4 ? ringP[i % kRing] : (par ? c->dist_P2.p : c->dist_P.p);
    if (mode == 0 && i >= 2) HIPCHK(c, hipStreamWaitEvent(main, c->ev_comm[par], 0));   // the exchange of i - 2 has released this partial buffer
    if (mode != 3) {
      RngArgs r = rng_of(c, counter_idx ? (uint64_t)i : idx0 + (uint64_t)i);
      if (counter_idx) r.idx_ptr = (const uint64_t *)c->d_idx.p;
      OutArgs o = final_out(c, nullptr, nullptr);
      o.partials = P;
      o.partials_mode = 1;
      o.scalars_off = mivi_partials_len(c) - 2;
      if ((s = run_estimate(c, params, r, c->cfg.n_mc, 1, o))) break;
    }
    if (mode == 2) continue;
    if (mode == 4) {   // peer-to-peer pipeline, compute chain: announce partial vector i, then wait until the exchange has read the ring slot estimate i + 1 overwrites
      unsigned *w = (unsigned *)c->p2p_ctr.p;
      const int slot = (i + 1) % kRing, prev_users = (i + 1) / kRing;
      // (folding this one-thread launch into the next estimate's product kernel as an extra workgroup was tried: the 8 us it takes from
      //  dispatch to completion beside the persistent exchange kernels moved into that kernel -- 9 + 8 -> 21.5 us --, the step stayed at 31 us)
      launch_p2p_handover(c, w + 64, (unsigned)i + 1u, prev_users >= 1 ? w + 80 + slot : nullptr, (unsigned)prev_users * (unsigned)c->p2p_G);
      continue;
    }
    if (mode == 0) {
      HIPCHK(c, hipEventRecord(c->ev_part[par], main));
      HIPCHK(c, hipStreamWaitEvent(c->comm_stream, c->ev_part[par], 0));
      c->stream = c->comm_stream;
      s = dist_collective(c, params, P, value, grad);
      c->stream = main;
      if (s) break;
      HIPCHK(c, hipEventRecord(c->ev_comm[par], c->comm_stream));
    } else {
      s = dist_collective(c, params, mode == 3 ? c->dist_P.p : P, value, grad);
    }
  }
  if (s == MIVI_OK && mode == 0) {   // join: the batch is complete when its last two exchanges are
    if (count >= 2) HIPCHK(c, hipStreamWaitEvent(main, c->ev_comm[(count - 2) & 1], 0));
    HIPCHK(c, hipStreamWaitEvent(main, c->ev_comm[(count - 1) & 1], 0));
  }
  return s;
}

static mivi_status_t dist_batch(mivi_ctx *c, const void *params, uint64_t idx0, int count, void *value, void *grad, int mode) {
  if (!graph_capturable(c)) return fail(c, MIVI_ERR_UNSUPPORTED, "batched sharded estimates need a device-resident built-in target");
  if (c->idx_src) return fail(c, MIVI_ERR_UNSUPPORTED, "an index source is set (mivi_set_index_source): batched calls keep their own device counter");
  if (c->p2p_on && !c->comm) { c->comm_world = c->p2p_world; c->comm_rank = c->p2p_rank; }
  mivi_status_t s;
  if ((s = dist_check(c)) || (s = ensure_work(c, c->cfg.n_mc))) return s;
  prepare_tables(c, c->cfg.n_mc);
  if ((s = reserve_target(c, c->cfg.n_mc)) || (s = ensure_dist(c))) return s;
  if (c->cfg.family == MIVI_FULLRANK && lds_path_shape_ok(c, c->cfg.n_mc) && !lds_prepare(c, c->cfg.n_mc)) return fail(c, MIVI_ERR_HIP, "full-rank work lists: allocation failed");
  if (!c->comm_stream) {
    // (plain non-blocking streams: a HIGH-priority stream starved the compute chain it was supposed to run beside -- its spinning
    //  kernel was scheduled first and the normal-priority graph never progressed; found on the GPU)
    // The persistent exchange kernels run BESIDE the compute chain on these streams.  Non-blocking: a blocking stream (what
    // hipExtStreamCreateWithCUMask creates) synchronises with the null stream, so a context living on the null stream deadlocked
    // against its own exchange kernel; a HIGH-priority stream starved the compute chain (both found on the GPU).
    // They need hardware queues of their own (HIP maps streams onto a small pool, GPU_MAX_HW_QUEUES, and two streams on one queue
    // serialise: the bounded hand-over waits then expire): a stream created with a CU mask carries the mask in its queue and gets one
    // -- all CUs enabled = no restriction.  Only for contexts on a real stream (see above).
    uint32_t mask[16];
    for (int k = 0; k < 16; ++k) mask[k] = 0xFFFFFFFFu;
    hipStream_t *cs[2] = {&c->comm_stream, &c->comm_stream2};
    for (int k = 0; k < 2; ++k) {
      if (c->stream == nullptr || hipExtStreamCreateWithCUMask(cs[k], 16, mask) != hipSuccess) {
        (void)hipGetLastError();
        HIPCHK(c, hipStreamCreateWithFlags(cs[k], hipStreamNonBlocking));
      }
    }
    for (int k = 0; k < 2; ++k) {
      HIPCHK(c, hipEventCreateWithFlags(&c->ev_part[k], hipEventDisableTiming));
      HIPCHK(c, hipEventCreateWithFlags(&c->ev_comm[k], hipEventDisableTiming));
    }
  }
  GraphCache &g = c->graph;
  const int route = mivi_comm_route(c);
  // Peer-to-peer route, pipelined: the exchange is ONE persistent kernel on comm_stream for the whole batch (kernels_p2p.hip), the compute
  // chain is a single-stream graph of {partial kernels, hand-over} per estimate; the two talk through two device words.
  bool p2p_pipe = mode == 0 && route == 3;
  if (p2p_pipe && c->p2p_pipe_state < 0) { p2p_pipe = false; mode = 1; }   // (its kernels did not run beside the compute chain on this context: serial steps)
  if (p2p_pipe) mode = 4;
  {   // lane-batched compute chain for the pipelined batches (see dist_sequence_lanes)
    static const bool no_lanes = getenv("MIVI_LANE_BATCH") && atoi(getenv("MIVI_LANE_BATCH")) == 0;
    const bool stl_ent = c->cfg.entropy == MIVI_ENT_STL || c->cfg.entropy == MIVI_ENT_STL_ZERO_GRAD;
    OutArgs on = final_out(c, nullptr, nullptr);
    on.partials = c->dist_P.p;
    on.partials_mode = 1;
    const bool lane4 = mode == 4 && !no_lanes && !stl_ent && !c->dbg && c->cfg.family == MIVI_FULLRANK && lds_route(c, params, c->cfg.n_mc, 1, on) &&
                       lds_use_prod32(c, c->cfg.n_mc) && lds_bf16x3() && count >= kGroup;
    const int stride = lane4 ? kGroup : 1;
    if (c->dist_lane4 != lane4 || c->idx_stride != stride) { invalidate_graph(c); c->dist_lane4 = lane4; c->idx_stride = stride; }
    if (lane4) {
      if ((s = ensure_kids(c, kGroup))) return s;
      for (int j = 0; j < kGroup - 1; ++j) {
        mivi_ctx *k = c->kids[j];
        if ((s = sync_kid(c, k, kGroup)) || (s = ensure_work(k, k->cfg.n_mc))) { c->err = k->err; return s; }
        prepare_tables(k, k->cfg.n_mc);
        if (!lds_prepare(k, k->cfg.n_mc)) return fail(c, MIVI_ERR_HIP, "full-rank work lists: allocation failed");
      }
    }
  }
  const int kind = 20 + mode;
  bool &capture_refused = c->dist_capture_refused;   // (a collective library that cannot be captured: do not retry on every call of THIS context)
  if (!(g.exec && g.kind == kind && g.count == count && g.params == params && g.value == value && g.grad == grad && g.p0 == (double)route) && !capture_refused) {
    invalidate_graph(c);
    if (c->dist_lane4) {
      for (int j = 0; j < kGroup - 1; ++j) {
        c->kids[j]->kid_gen = c->target_gen;
        HIPCHK(c, hipStreamSynchronize(c->kids[j]->stream));   // (their table uploads, before the capture)
      }
    }
    hipGraph_t graph = nullptr;
    hipStream_t saved;
    if ((s = begin_capture(c, &saved))) return s;
    s = dist_sequence(c, params, true, 0, count, value, grad, mode);
    if (s == MIVI_OK && mode != 3) hipLaunchKernelGGL(k_bump_u64, dim3(1), dim3(1), 0, c->stream, (uint64_t *)c->d_idx.p, (uint64_t)count);
    c->cur = 0;
    c->pre_valid = false;
    hipError_t e = end_capture(c, saved, &graph);
    if (s == MIVI_OK && e == hipSuccess && graph) e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
    if (graph) (void)hipGraphDestroy(graph);
    if (s != MIVI_OK || e != hipSuccess) {
      (void)hipGetLastError();
      g = GraphCache{};
      capture_refused = true;
      (void)hipStreamSynchronize(c->comm_stream);
    } else {
      g.kind = kind; g.count = count; g.params = params; g.value = value; g.grad = grad; g.p0 = (double)route;
    }
  }
  auto p2p_front = [&]() -> mivi_status_t {   // hand-over words reset, then the persistent exchange kernels (one per lane) on their own streams
    unsigned *w = (unsigned *)c->p2p_ctr.p;
    HIPCHK(c, hipMemsetAsync(w + 64, 0, 128, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_part[0], c->stream));
    hipStream_t main = c->stream;
    const void *ringP[kRing] = {c->dist_P.p, c->dist_P2.p, c->dist_ring[0].p, c->dist_ring[1].p, c->dist_ring[2].p, c->dist_ring[3].p, c->dist_ring[4].p, c->dist_ring[5].p};
    // ONE persistent exchange kernel (measured on one GPU: 21 us per estimate against 31 with two of them serving alternate estimates -- a
    // second resident kernel costs the compute chain more than its overlap wins)
    const int lanes = 1;
    for (int ln = 0; ln < lanes; ++ln) {
      hipStream_t cs = ln ? c->comm_stream2 : c->comm_stream;
      HIPCHK(c, hipStreamWaitEvent(cs, c->ev_part[0], 0));
      c->stream = cs;
      launch_p2p_exchange(c, params, ringP, kRing, value, grad, 7, ln, lanes, count, w + 64, w + 80);
      c->stream = main;
      HIPCHK(c, hipEventRecord(c->ev_comm[ln], cs));
    }
    return MIVI_OK;
  };
  auto p2p_back = [&]() -> mivi_status_t {
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_comm[0], 0));
    return MIVI_OK;
  };
  if (g.exec && g.kind == kind && g.count == count && g.params == params && g.value == value && g.grad == grad && g.p0 == (double)route) {
    if (!(c->d_idx_valid && c->d_idx_expect == idx0))
      hipLaunchKernelGGL(k_set_u64x2, dim3(1), dim3(1), 0, c->stream, (uint64_t *)c->d_idx.p, idx0, 0ull, 1);
    if (p2p_pipe && (s = p2p_front())) return s;
    HIPCHK(c, hipGraphLaunch(g.exec, c->stream));
    if (p2p_pipe && (s = p2p_back())) return s;
    c->d_idx_valid = mode != 3;
    c->d_idx_expect = idx0 + (uint64_t)count;
    return MIVI_OK;
  }
  // eager: the same sequence with by-value indices
  c->pre_valid = false;
  if (p2p_pipe && (s = p2p_front())) return s;
  s = dist_sequence(c, params, false, idx0, count, value, grad, mode);
  if (p2p_pipe && s == MIVI_OK) s = p2p_back();
  c->cur = 0;
  c->pre_valid = false;
  return s;
}

mivi_status_t mivi_estimate_gradient_dist_n(mivi_ctx_t *c, const void *params, uint64_t idx0, int32_t count, void *value, void *grad) {
  if (!c || !params || !value || !grad || count <= 0) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  return dist_batch(c, params, idx0, count, value, grad, 0);
}

// tests: the phases of the peer-to-peer exchange one launch at a time (several ranks of ONE process driven from one host thread)
mivi_status_t mivi_p2p_exchange(mivi_ctx_t *c, const void *params, const void *partials, void *value, void *grad, int32_t phases) {
  if (!c || !params || !partials || !value || !grad || phases < 1 || phases > 7) return MIVI_ERR_BAD_ARG;
  if (!c->p2p_on) return fail(c, MIVI_ERR_BAD_ARG, "no peer-to-peer exchange buffers attached");
  (void)hipSetDevice(c->cfg.device);
  const void *Ps[1] = {partials};
  launch_p2p_exchange(c, params, Ps, 1, value, grad, phases, 0, 1, 1, nullptr, nullptr);
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}

// us per estimate of the sharded step and of its pieces, every rank calling collectively: out[0] partial kernels, out[1] exchange +
// finalisation, out[2] serial step (one stream), out[3] pipelined step (mivi_estimate_gradient_dist_n).  hipEvents on the context's
// stream around ONE graph replay of `reps` estimates each (after a warm replay).
mivi_status_t mivi_profile_dist(mivi_ctx_t *c, const void *params, int32_t reps, double *us_out) {
  if (!c || !params || reps <= 0 || !us_out) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  char *o = (char *)c->tmp_out.p;
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0));
  HIPCHK(c, hipEventCreate(&e1));
  mivi_status_t s = MIVI_OK;
  const int order[4] = {2, 3, 1, 0};   // partials first: the exchange-only leg works on the partial vector they leave
  for (int k = 0; k < 4 && s == MIVI_OK; ++k) {
    const int mode = order[k];
    if ((s = dist_batch(c, params, 1000, reps, o, o + 16, mode))) break;   // warm (captures)
    HIPCHK(c, hipEventRecord(e0, c->stream));
    if ((s = dist_batch(c, params, 1000 + reps, reps, o, o + 16, mode))) break;
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    us_out[mode == 2 ? 0 : (mode == 3 ? 1 : (mode == 1 ? 2 : 3))] = (double)ms * 1e3 / reps;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  invalidate_graph(c);
  return s;
}

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

// elbo record (double, device) -> caller's T[n_steps]
static mivi_status_t deliver_elbo(mivi_ctx *c, const double *rec, int n_steps, void *elbo) {
  if (!elbo) return MIVI_OK;
  if (c->cfg.dtype == MIVI_F64) {
    HIPCHK(c, hipMemcpyAsync(elbo, rec, (size_t)n_steps * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  } else {
    std::vector<double> h(n_steps);
    HIPCHK(c, hipMemcpyAsync(h.data(), rec, (size_t)n_steps * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<float> f(h.begin(), h.end());
    HIPCHK(c, hipMemcpy(elbo, f.data(), (size_t)n_steps * sizeof(float), hipMemcpyHostToDevice));
  }
  return MIVI_OK;
}

static bool same_loop(const mivi_loop_t &a, const mivi_loop_t &b) {   // everything baked into a captured loop
  return a.rule == b.rule && a.op == b.op && a.averager == b.averager && a.n_steps == b.n_steps && a.eta == b.eta &&
         a.beta1 == b.beta1 && a.beta2 == b.beta2 && a.adam_eps == b.adam_eps && a.clip_epsilon == b.clip_epsilon &&
         a.avg_eta == b.avg_eta && a.opt_state_dev == b.opt_state_dev && a.avg_params_dev == b.avg_params_dev;
}

mivi_status_t mivi_optimize_loop(mivi_ctx_t *c, void *params, const mivi_loop_t *lp) {
  if (!c || !params || !lp) return MIVI_ERR_BAD_ARG;
  const mivi_loop_t &l = *lp;
  const int n_steps = l.n_steps, rule = l.rule;
  if (n_steps <= 0 || rule < 0 || rule > 3 || l.op < 0 || l.op > 2 || l.averager < 0 || l.averager > 1) return MIVI_ERR_BAD_ARG;
  if (rule != 0 && !l.opt_state_dev) return fail(c, MIVI_ERR_BAD_ARG, "this optimisation rule needs opt_state_dev");
  if (l.averager == 1 && !l.avg_params_dev) return fail(c, MIVI_ERR_BAD_ARG, "PolynomialAveraging needs avg_params_dev");
  if (l.op == 2 && rule == 1) return fail(c, MIVI_ERR_BAD_ARG, "ProximalLocationScaleEntropy does not support Adam (Descent, DoG, DoWG)");
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
  const size_t hist_doubles = (size_t)n_steps * 4 * (size_t)((c->cfg.d + 3) / 4);
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
      else if (l.op <= 1 && launch_dog_update_fused(c, params, gbuf, opt_state, rule - 2, clip_eps,
                                                     l.averager == 1 ? l.avg_params_dev : nullptr, l.avg_eta, t_ptr, (long long)i + 1))
        continue;   // DoG / DoWG + ClipScale + averaging in one apply pass (large parameter vectors)
      else launch_dog_update(c, params, gbuf, opt_state, rule - 2);
      // operator (common.jl:93-95)
      if (l.op == 1 && rule >= 2) launch_clip(c, params, clip_eps);
      if (l.op == 2) launch_prox(c, params, eta, rule >= 2 ? opt_state : nullptr, rule - 2);
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

int32_t mivi_fullrank_route(const mivi_ctx_t *c, int32_t n_samples) {
  if (!c || c->cfg.family != MIVI_FULLRANK) return 0;
  if (n_samples <= 0) n_samples = c->cfg.n_mc;
  if (!lds_path_shape_ok(c, n_samples) || (c->target != TGT_DIAG_GAUSS && c->target != TGT_DENSE_GAUSS)) return 0;
  return (lds_use_prod32(c, n_samples) ? 1 : 3) | (lds_bf16x3() ? 16 : 0);
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

}  // extern "C"
