// Small elementwise kernels next to the hot path:
//   finalize   partials (after the RCCL all-reduce) -> value, gradient        
//   value-only objective assembly for estimate_objective                      src/algorithms/repgradelbo.jl:112-118
//   ClipScale  scale[diagind] = max(scale[diagind], eps)                      src/optimization/clip_scale.jl:18-29
//   Descent / Adam parameter updates (Optimisers.update!)                     src/algorithms/common.jl:92
#include "device_common.h"
#include "optim_rules.h"

namespace mivi {

template <typename T>
struct FinArgs {
  int d, family;
  const T *params;
  const T *partials;
  T *grad;
  T *value;
  int ent_kind, M_total;
  int *status;
};

template <typename T>
__global__ __launch_bounds__(256) void k_finalize(FinArgs<T> a) {
  __shared__ double red[4];
  const int d = a.d;
  const int64_t plen = a.family == MIVI_MEANFIELD ? 2 * (int64_t)d : (int64_t)d + (int64_t)d * d;
  const double invM = 1.0 / (double)a.M_total;
  const double direct = direct_entropy_coeff(a.ent_kind);
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < plen; t += (int64_t)gridDim.x * 256) {
    double g;
    if (a.family == MIVI_MEANFIELD || t < d) {
      g = -(double)a.partials[t] * invM;
      if (t >= d) g -= direct / (double)a.params[t];
    } else {   // unpack the column-packed lower triangle; the strict upper triangle of the gradient is exact zeros
      const int64_t e = t - d;
      const int64_t j = e / d, i = e - j * d;
      if (j > i) {
        g = 0.0;
      } else {
        g = -(double)a.partials[d + j * d - (j * (j - 1)) / 2 + (i - j)] * invM;
        if (i == j) g -= direct / (double)a.params[t];
      }
    }
    a.grad[t] = (T)g;
  }
  if (blockIdx.x == 0) {
    double s_ld = 0.0, bad = 0.0;
    for (int i = threadIdx.x; i < d; i += 256) {
      const double c = (double)(a.family == MIVI_MEANFIELD ? a.params[d + i] : a.params[d + (size_t)i * d + i]);
      if (!(c > 0.0)) bad = 1.0;
      s_ld += log(c);
    }
    s_ld = block_sum<double, 256>(s_ld, red);
    bad = block_sum<double, 256>(bad, red);
    if (threadIdx.x == 0) {
      const int64_t so = a.family == MIVI_MEANFIELD ? plen : (int64_t)d + ((int64_t)d * (d + 1)) / 2;
      const double sum_ell = (double)a.partials[so], s_he = (double)a.partials[so + 1];
      const double Mt = (double)a.M_total;
      const double ent = (ent_is_closed(a.ent_kind) ? 0.5 * d * (1.0 + kLog2Pi) : s_he / Mt + 0.5 * d * kLog2Pi) + s_ld;
      const double value = -(sum_ell / Mt + ent);
      *a.value = (T)value;
      int st = 0;
      if (!isfinite(value)) st |= 1;
      if (bad > 0.0) st |= 2;
      if (st && a.status) atomicOr(a.status, st);
    }
  }
}

template <typename T>
static void finalize_impl(mivi_ctx *c, const void *params, const void *partials, void *value, void *grad) {
  FinArgs<T> a;
  a.d = c->cfg.d;
  a.family = c->cfg.family;
  a.params = (const T *)params;
  a.partials = (const T *)partials;
  a.grad = (T *)grad;
  a.value = (T *)value;
  a.ent_kind = c->cfg.entropy;
  a.M_total = c->M_total;
  a.status = (int *)c->status.p;
  const int64_t plen = mivi_params_len(c);
  int nb = (int)((plen + 255) / 256);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(k_finalize<T>, dim3(nb), dim3(256), 0, c->stream, a);
}
void launch_finalize(mivi_ctx *c, const void *params, const void *partials, void *value, void *grad) {
  if (c->cfg.dtype == MIVI_F32) finalize_impl<float>(c, params, partials, value, grad);
  else finalize_impl<double>(c, params, partials, value, grad);
}

// ---------------------------------------------------------------------------------------------
// Sharded finalisation (multi-GPU): reduce-scatter -> every rank finalises ITS 1/R slice of the partial vector in the packed
// domain -> all-gather -> every rank unpacks.  The packed "final" vector F has the layout of the partial vector:
//   [d/dmu (d); mean-field: d/dsigma (d) | full-rank: column-packed lower triangle of d/dC (d(d+1)/2); value; status bits]
// so that the collective moves the packed triangle both ways and no rank normalises more than its slice.
// ---------------------------------------------------------------------------------------------
template <typename T>
struct SliceArgs {
  int d, family, ent_kind, M_total;
  long long L, g0, n;      // partial length, first global index of this slice, slice length
  const T *params;
  const T *sum;            // summed partials of this slice [n]
  T *fin;                  // packed final values of this slice [n]
  double ell_const;
  int M_units;             // number of sample columns behind sum(ell): ell_const is added once per column
};

template <typename T>
__global__ __launch_bounds__(256) void k_finalize_slice(SliceArgs<T> a) {
  __shared__ double red[4];
  const int d = a.d;
  const double invM = 1.0 / (double)a.M_total;
  const double direct = direct_entropy_coeff(a.ent_kind);
  const long long tri_end = a.L - 2;
  if (blockIdx.x + 1 < gridDim.x) {
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < a.n; t += (long long)(gridDim.x - 1) * 256) {
      const long long g = a.g0 + t;
      if (g >= tri_end) continue;          // the two scalars (and the padding) belong to the value block
      double v = -(double)a.sum[t] * invM;
      if (g >= d) {
        if (a.family == MIVI_MEANFIELD) {
          v -= direct / (double)a.params[g];
        } else {                           // packed entry e = j d - j (j - 1) / 2 + (i - j): find its column j
          const long long e = g - d;
          const double b = 2.0 * d + 1.0;
          long long j = (long long)((b - sqrt(b * b - 8.0 * (double)e)) * 0.5);
          if (j < 0) j = 0;
          if (j > d - 1) j = d - 1;
          while (j > 0 && j * d - (j * (j - 1)) / 2 > e) --j;
          while (j + 1 < d && (j + 1) * d - ((j + 1) * j) / 2 <= e) ++j;
          const long long i = j + (e - (j * d - (j * (j - 1)) / 2));
          if (i == j) v -= direct / (double)a.params[d + (size_t)j * d + j];
        }
      }
      a.fin[t] = (T)v;
    }
    return;
  }
  // value block: only on the rank whose slice holds the scalars
  if (a.g0 > tri_end || a.g0 + a.n < a.L) {
    for (long long t = threadIdx.x; t < a.n; t += 256)
      if (a.g0 + t >= tri_end) a.fin[t] = T(0);      // padding beyond L
    return;
  }
  double s_ld = 0.0, bad = 0.0;
  for (int i = threadIdx.x; i < d; i += 256) {
    const double c = (double)(a.family == MIVI_MEANFIELD ? a.params[d + i] : a.params[d + (size_t)i * d + i]);
    if (!(c > 0.0)) bad = 1.0;
    s_ld += log(c);
  }
  s_ld = block_sum<double, 256>(s_ld, red);
  bad = block_sum<double, 256>(bad, red);
  for (long long t = threadIdx.x; t < a.n; t += 256)
    if (a.g0 + t >= a.L) a.fin[t] = T(0);
  if (threadIdx.x == 0) {
    const long long so = tri_end - a.g0;
    const double sum_ell = (double)a.sum[so], s_he = (double)a.sum[so + 1];
    const double Mt = (double)a.M_total;
    const double ent = (ent_is_closed(a.ent_kind) ? 0.5 * d * (1.0 + kLog2Pi) : s_he / Mt + 0.5 * d * kLog2Pi) + s_ld;
    const double value = -(sum_ell / Mt + ent);
    int st = 0;
    if (!isfinite(value)) st |= 1;
    if (bad > 0.0) st |= 2;
    a.fin[so] = (T)value;
    a.fin[so + 1] = (T)st;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_unpack_final(int d, int family, long long L, const T *fin, T *value, T *grad, int *status) {
  const int64_t plen = family == MIVI_MEANFIELD ? 2 * (int64_t)d : (int64_t)d + (int64_t)d * d;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < plen; t += (int64_t)gridDim.x * 256) {
    T g;
    if (family == MIVI_MEANFIELD || t < d) {
      g = fin[t];
    } else {
      const int64_t e = t - d;
      const int64_t j = e / d, i = e - j * d;
      g = (j > i) ? T(0) : fin[d + j * d - (j * (j - 1)) / 2 + (i - j)];
    }
    grad[t] = g;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *value = fin[L - 2];
    const int st = (int)fin[L - 1];
    if (st && status) atomicOr(status, st);
  }
}

void launch_finalize_slice(mivi_ctx *c, const void *params, const void *sum, long long g0, long long n, void *fin) {
  const long long L = mivi_partials_len(c);
  int nb = (int)((n + 255) / 256);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  if (c->cfg.dtype == MIVI_F32) {
    SliceArgs<float> a{c->cfg.d, c->cfg.family, c->cfg.entropy, c->M_total, L, g0, n, (const float *)params, (const float *)sum, (float *)fin, 0.0, 0};
    hipLaunchKernelGGL(k_finalize_slice<float>, dim3(nb + 1), dim3(256), 0, c->stream, a);
  } else {
    SliceArgs<double> a{c->cfg.d, c->cfg.family, c->cfg.entropy, c->M_total, L, g0, n, (const double *)params, (const double *)sum, (double *)fin, 0.0, 0};
    hipLaunchKernelGGL(k_finalize_slice<double>, dim3(nb + 1), dim3(256), 0, c->stream, a);
  }
}
void launch_unpack_final(mivi_ctx *c, const void *fin, void *value, void *grad) {
  const long long L = mivi_partials_len(c);
  const int64_t plen = mivi_params_len(c);
  int nb = (int)((plen + 255) / 256);
  if (nb > 2048) nb = 2048;
  if (c->cfg.dtype == MIVI_F32)
    hipLaunchKernelGGL(k_unpack_final<float>, dim3(nb), dim3(256), 0, c->stream, c->cfg.d, c->cfg.family, L, (const float *)fin, (float *)value,
                       (float *)grad, (int *)c->status.p);
  else
    hipLaunchKernelGGL(k_unpack_final<double>, dim3(nb), dim3(256), 0, c->stream, c->cfg.d, c->cfg.family, L, (const double *)fin, (double *)value,
                       (double *)grad, (int *)c->status.p);
}

template <typename T>
__global__ __launch_bounds__(256) void k_value_only(int d, int family, const T *params, ValueIn vin, OutArgs out) {
  __shared__ double red[4 * 4];
  const int64_t plen = family == MIVI_MEANFIELD ? 2 * (int64_t)d : (int64_t)d + (int64_t)d * d;
  const int fam = family;
  finalize_value_block<T, 256, false, false>(d, vin, out, plen,
      [params, d, fam](int i) { return fam == MIVI_MEANFIELD ? params[d + i] : params[d + (size_t)i * d + i]; }, red);
}
// the fused funnel's finisher (funnel_finish) is its own kernel: inlined into k_value_only it would sit in every caller's
// register budget.  (Neither 1024 threads (10.0 us) nor 32 loads in flight per lane (8.1 us) beat this 8.1 us: a dependent
// single-workgroup launch costs 4.5 us before it does anything.)
template <typename T>
__global__ __launch_bounds__(256) void k_value_funnel(int d, const T *params, ValueIn vin, OutArgs out) {
  __shared__ double red[6 * 4];
  finalize_value_block<T, 256, false, true>(d, vin, out, 2 * (int64_t)d, [params, d](int i) { return params[d + i]; }, red);
}
// lane-batched contexts (api_batch.hip): the closing value kernels of up to four contexts as ONE launch (blockIdx.x = lane)
struct ValueMulti { ValueIn vin[4]; OutArgs out[4]; };
__global__ __launch_bounds__(256) void k_value_only_m(int d, int family, const float *params, ValueMulti m) {
  __shared__ double red[4 * 4];
  const int64_t plen = family == MIVI_MEANFIELD ? 2 * (int64_t)d : (int64_t)d + (int64_t)d * d;
  const int fam = family;
  finalize_value_block<float, 256, false, false>(d, m.vin[blockIdx.x], m.out[blockIdx.x], plen,
      [params, d, fam](int i) { return fam == MIVI_MEANFIELD ? params[d + i] : params[d + (size_t)i * d + i]; }, red);
}
struct ValueSink { ValueMulti m; int n; };
ValueSink *value_sink_alloc() { return new ValueSink(); }
void value_sink_free(ValueSink *s) { delete s; }
void launch_lanes_value(mivi_ctx *c, const void *params, ValueSink *s) {
  if (s->n > 0) hipLaunchKernelGGL(k_value_only_m, dim3(s->n), dim3(256), 0, c->stream, c->cfg.d, c->cfg.family, (const float *)params, s->m);
  s->n = 0;
}

void launch_value_only(mivi_ctx *c, const void *params, const ValueIn &vin, const OutArgs &out) {
  if (c->value_sink && c->cfg.dtype == MIVI_F32 && !(vin.fn.ab && c->cfg.family == MIVI_MEANFIELD) && ((ValueSink *)c->value_sink)->n < 4) {
    ValueSink *sk = (ValueSink *)c->value_sink;   // record: the driver issues the lanes' value kernels as one launch
    sk->m.vin[sk->n] = vin;
    sk->m.out[sk->n] = out;
    ++sk->n;
    return;
  }
  if (vin.fn.ab && c->cfg.family == MIVI_MEANFIELD) {
    if (c->cfg.dtype == MIVI_F32)
      hipLaunchKernelGGL(k_value_funnel<float>, dim3(1), dim3(256), 0, c->stream, c->cfg.d, (const float *)params, vin, out);
    else
      hipLaunchKernelGGL(k_value_funnel<double>, dim3(1), dim3(256), 0, c->stream, c->cfg.d, (const double *)params, vin, out);
    return;
  }
  if (c->cfg.dtype == MIVI_F32)
    hipLaunchKernelGGL(k_value_only<float>, dim3(1), dim3(256), 0, c->stream, c->cfg.d, c->cfg.family,
                       (const float *)params, vin, out);
  else
    hipLaunchKernelGGL(k_value_only<double>, dim3(1), dim3(256), 0, c->stream, c->cfg.d, c->cfg.family,
                       (const double *)params, vin, out);
}

template <typename T>
__global__ void k_clip(int d, int family, T *params, T epsilon) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < d) {
    const size_t o = family == MIVI_MEANFIELD ? (size_t)d + i : (size_t)d + (size_t)i * d + i;
    params[o] = clip_step(params[o], epsilon);
  }
}
void launch_clip(mivi_ctx *c, void *params, double epsilon) {
  const int d = c->cfg.d;
  if (c->cfg.dtype == MIVI_F32)
    hipLaunchKernelGGL(k_clip<float>, dim3((d + 255) / 256), dim3(256), 0, c->stream, d, c->cfg.family, (float *)params,
                       (float)epsilon);
  else
    hipLaunchKernelGGL(k_clip<double>, dim3((d + 255) / 256), dim3(256), 0, c->stream, d, c->cfg.family,
                       (double *)params, epsilon);
}

// ProximalLocationScaleEntropy (src/optimization/proximal_location_scale_entropy.jl:44-61).  The step size comes from the
// optimiser: Descent -> eta (by value); DoG -> r / sqrt(v); DoWG -> r^2 / sqrt(v) read from the device-resident (v, r) (:26-42).
template <typename T>
__global__ void k_prox(int d, int family, T *params, double stepsize, const double *dog_sc, int dog_kind) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < d) {
    double g = stepsize;
    if (dog_sc) {
      const double v = dog_sc[0], r = dog_sc[1];
      g = (dog_kind == 1 ? r * r : r) / sqrt(v);
    }
    const size_t o = family == MIVI_MEANFIELD ? (size_t)d + i : (size_t)d + (size_t)i * d + i;
    params[o] = prox_entropy_step(params[o], (T)g);
  }
}
void launch_prox(mivi_ctx *c, void *params, double stepsize, const void *dog_state, int dog_kind) {
  const int d = c->cfg.d;
  const double *sc = dog_state ? (const double *)((const char *)dog_state + mivi_dog_state_bytes(c) - 16) : nullptr;
  if (c->cfg.dtype == MIVI_F32)
    hipLaunchKernelGGL(k_prox<float>, dim3((d + 255) / 256), dim3(256), 0, c->stream, d, c->cfg.family, (float *)params,
                       stepsize, sc, dog_kind);
  else
    hipLaunchKernelGGL(k_prox<double>, dim3((d + 255) / 256), dim3(256), 0, c->stream, d, c->cfg.family, (double *)params,
                       stepsize, sc, dog_kind);
}

// is flat index i a scale-diagonal entry? (ClipScale fused into the update; clip_eps = NaN means "no ClipScale": any real epsilon, also <= 0, is a legal one)
__device__ __forceinline__ bool is_scale_diag(int64_t i, int d, int family) {
  if (i < d) return false;
  if (family == MIVI_MEANFIELD) return true;
  const int64_t e = i - d;
  return (e / d) == (e % d);
}

template <typename T>
__global__ void k_descent(int64_t n, T *params, const T *grad, T eta, int d, int family, T clip_eps) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    T x = descent_step(params[i], grad[i], eta);
    if (clip_eps == clip_eps && is_scale_diag(i, d, family)) x = clip_step(x, clip_eps);
    params[i] = x;
  }
}
void launch_descent(mivi_ctx *c, void *params, const void *grad, double eta, double clip_eps) {
  const int64_t n = mivi_params_len(c);
  int nb = (int)((n + 255) / 256);
  if (nb > 2048) nb = 2048;
  if (c->cfg.dtype == MIVI_F32)
    hipLaunchKernelGGL(k_descent<float>, dim3(nb), dim3(256), 0, c->stream, n, (float *)params, (const float *)grad,
                       (float)eta, c->cfg.d, c->cfg.family, (float)clip_eps);
  else
    hipLaunchKernelGGL(k_descent<double>, dim3(nb), dim3(256), 0, c->stream, n, (double *)params, (const double *)grad,
                       eta, c->cfg.d, c->cfg.family, clip_eps);
}

// Optimisers.Adam (optim_rules.h); bias corrections once per workgroup
template <typename T>
__global__ void k_adam(int64_t n, T *params, const T *grad, T *state, const int64_t *t_ptr, int64_t t_base, double eta,
                       double b1, double b2, double eps, int d, int family, T clip_eps) {
  __shared__ T cc[2];
  if (threadIdx.x == 0) adam_bias<T>(t_base + (t_ptr ? *t_ptr : 0), b1, b2, cc[0], cc[1]);
  __syncthreads();
  const T c1 = cc[0], c2 = cc[1];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    T m = state[i], v = state[n + i];
    T x = adam_step<T>(params[i], grad[i], m, v, c1, c2, (T)eta, (T)b1, (T)b2, (T)eps);
    if (clip_eps == clip_eps && is_scale_diag(i, d, family)) x = clip_step(x, clip_eps);
    params[i] = x;
    state[i] = m;
    state[n + i] = v;
  }
}
void launch_adam(mivi_ctx *c, void *params, const void *grad, void *state, const int64_t *t_ptr, int64_t t_base,
                 double eta, double b1, double b2, double eps, double clip_eps) {
  const int64_t n = mivi_params_len(c);
  int nb = (int)((n + 255) / 256);
  if (nb > 2048) nb = 2048;
  if (c->cfg.dtype == MIVI_F32)
    hipLaunchKernelGGL(k_adam<float>, dim3(nb), dim3(256), 0, c->stream, n, (float *)params, (const float *)grad,
                       (float *)state, t_ptr, t_base, eta, b1, b2, eps, c->cfg.d, c->cfg.family, (float)clip_eps);
  else
    hipLaunchKernelGGL(k_adam<double>, dim3(nb), dim3(256), 0, c->stream, n, (double *)params, (const double *)grad,
                       (double *)state, t_ptr, t_base, eta, b1, b2, eps, c->cfg.d, c->cfg.family, clip_eps);
}

// COCOB (optim_rules.h): state = [L; G; R; theta; x1], n elements each
template <typename T>
__global__ void k_cocob(int64_t n, T *params, const T *grad, T *state, T alpha, int d, int family, T clip_eps) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    T L = state[i], G = state[n + i], R = state[2 * n + i], th = state[3 * n + i];
    T x = cocob_step<T>(params[i], grad[i], L, G, R, th, state[4 * n + i], alpha);
    if (clip_eps == clip_eps && is_scale_diag(i, d, family)) x = clip_step(x, clip_eps);
    params[i] = x;
    state[i] = L;
    state[n + i] = G;
    state[2 * n + i] = R;
    state[3 * n + i] = th;
  }
}
void launch_cocob(mivi_ctx *c, void *params, const void *grad, void *state, double alpha, double clip_eps) {
  const int64_t n = mivi_params_len(c);
  int nb = (int)((n + 255) / 256);
  if (nb > 2048) nb = 2048;
  if (c->cfg.dtype == MIVI_F32)
    hipLaunchKernelGGL(k_cocob<float>, dim3(nb), dim3(256), 0, c->stream, n, (float *)params, (const float *)grad, (float *)state,
                       (float)alpha, c->cfg.d, c->cfg.family, (float)clip_eps);
  else
    hipLaunchKernelGGL(k_cocob<double>, dim3(nb), dim3(256), 0, c->stream, n, (double *)params, (const double *)grad, (double *)state,
                       alpha, c->cfg.d, c->cfg.family, clip_eps);
}

__global__ void k_bump(uint64_t *ctr, uint64_t by) { *ctr += by; }
void launch_bump(mivi_ctx *c, uint64_t *ctr, uint64_t by) {
  hipLaunchKernelGGL(k_bump, dim3(1), dim3(1), 0, c->stream, ctr, by);
}

}  // namespace mivi

// ---------------------------------------------------------------------------------------------
// axpby (PolynomialAveraging, src/optimization/averaging.jl:40-47) and DoG / DoWG
// (src/optimization/rules.jl:17-64).  DoG/DoWG need two global norms; one 1024-thread workgroup
// reduces and applies (O(params) work, off the estimator's critical path).
// ---------------------------------------------------------------------------------------------
namespace mivi {

// PolynomialAveraging inside a captured loop: the weight depends on the step, which lives in a device counter
// (same double arithmetic as the host computes for mivi_axpby: bitwise identical averages)
template <typename T>
__global__ void k_poly_average(int64_t n, T *y, const T *x, double avg_eta, const long long *t_ptr, long long t_base) {
  const double t = (double)(t_base + (t_ptr ? *t_ptr : 0));
  const double w = (avg_eta + 1.0) / (t + avg_eta);
  const double a = w, b = 1.0 - w;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = poly_avg_step<T>(x[i], y[i], a, b);
}
void launch_poly_average(mivi_ctx *c, void *avg, const void *params, double avg_eta, const long long *t_ptr, long long t_base) {
  const int64_t n = mivi_params_len(c);
  int nb = (int)((n + 255) / 256);
  if (nb > 2048) nb = 2048;
  if (c->cfg.dtype == MIVI_F32)
    hipLaunchKernelGGL(k_poly_average<float>, dim3(nb), dim3(256), 0, c->stream, n, (float *)avg, (const float *)params, avg_eta, t_ptr, t_base);
  else
    hipLaunchKernelGGL(k_poly_average<double>, dim3(nb), dim3(256), 0, c->stream, n, (double *)avg, (const double *)params, avg_eta, t_ptr, t_base);
}

template <typename T>
__global__ void k_axpby(int64_t n, T *y, double a, const T *x, double b) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = poly_avg_step<T>(x[i], y[i], a, b);
}

template <typename T>
__global__ __launch_bounds__(1024) void k_dog_init(int64_t n, const T *params, T *x0, double *sc, double alpha) {
  __shared__ double red[16];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const T v = params[i];
    x0[i] = v;
    s += (double)v * (double)v;
  }
  s = block_sum<double, 1024>(s, red);
  if (threadIdx.x == 0) {
    sc[0] = 0.0;                          // v
    sc[1] = alpha * (1.0 + sqrt(s));      // r
  }
}

// FUSE: ClipScale and PolynomialAveraging folded into the apply loop (device-resident loop), same per-element arithmetic
template <typename T, bool FUSE = false>
__global__ __launch_bounds__(1024) void k_dog_update(int64_t n, T *params, const T *grad, const T *x0, double *sc, int kind,
                                                     int d = 0, int family = 0, T clip_eps = T(NAN), T *avg = nullptr,
                                                     double avg_eta = 0.0, const long long *t_ptr = nullptr, long long t_base = 0) {
  __shared__ double red[16];
  double dist2 = 0.0, g2 = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const double dx = (double)params[i] - (double)x0[i];
    const double g = (double)grad[i];
    dist2 += dx * dx;
    g2 += g * g;
  }
  dist2 = block_sum<double, 1024>(dist2, red);
  g2 = block_sum<double, 1024>(g2, red);
  double r = sc[1], v = sc[0];
  r = fmax(sqrt(dist2), r);
  double eta;
  if (kind == 1) {  // DoWG
    const double r2 = r * r;
    v = v + r2 * g2;
    eta = r2 / sqrt(v);
  } else {          // DoG
    v = v + g2;
    eta = r / sqrt(v);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    sc[0] = v;
    sc[1] = r;
  }
  if (!FUSE) {
    for (int64_t i = threadIdx.x; i < n; i += 1024) params[i] = (T)((double)params[i] - eta * (double)grad[i]);
  } else {
    double wa = 0.0, wb = 0.0;
    if (avg) {
      const double t = (double)(t_base + (t_ptr ? *t_ptr : 0));
      wa = (avg_eta + 1.0) / (t + avg_eta);
      wb = 1.0 - wa;
    }
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
      T x = (T)((double)params[i] - eta * (double)grad[i]);
      if (clip_eps == clip_eps && is_scale_diag(i, d, family)) x = clip_step(x, clip_eps);
      params[i] = x;
      if (avg) avg[i] = poly_avg_step<T>(x, avg[i], wa, wb);
    }
  }
}

void launch_axpby(mivi_ctx *c, void *y, double a, const void *x, double b, int64_t n) {
  int nb = (int)((n + 255) / 256);
  if (nb > 2048) nb = 2048;
  if (c->cfg.dtype == MIVI_F32)
    hipLaunchKernelGGL(k_axpby<float>, dim3(nb), dim3(256), 0, c->stream, n, (float *)y, a, (const float *)x, b);
  else
    hipLaunchKernelGGL(k_axpby<double>, dim3(nb), dim3(256), 0, c->stream, n, (double *)y, a, (const double *)x, b);
}
void launch_dog_init(mivi_ctx *c, const void *params, void *state, double alpha) {
  const int64_t n = mivi_params_len(c);
  double *sc = (double *)((char *)state + mivi_dog_state_bytes(c) - 16);
  if (c->cfg.dtype == MIVI_F32)
    hipLaunchKernelGGL(k_dog_init<float>, dim3(1), dim3(1024), 0, c->stream, n, (const float *)params, (float *)state, sc, alpha);
  else
    hipLaunchKernelGGL(k_dog_init<double>, dim3(1), dim3(1024), 0, c->stream, n, (const double *)params, (double *)state, sc, alpha);
}
// Large parameter vectors (full-rank: d + d^2): the single-workgroup kernel above walks 10^6 elements in 600 us.
// Three launches instead: per-workgroup partials of the two norms -> one thread folds them (fixed order), advances
// (v, r) and leaves the step size -> every workgroup applies it.  DoG / DoWG are the reference's DEFAULT rules.
template <typename T>
__global__ __launch_bounds__(256) void k_dog_norms(int64_t n, const T *params, const T *grad, const T *x0, double *part) {
  __shared__ double red[4];
  double dist2 = 0.0, g2 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double dx = (double)params[i] - (double)x0[i];
    const double g = (double)grad[i];
    dist2 += dx * dx;
    g2 += g * g;
  }
  dist2 = block_sum<double, 256>(dist2, red);
  g2 = block_sum<double, 256>(g2, red);
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = dist2;
    part[2 * blockIdx.x + 1] = g2;
  }
}
__global__ __launch_bounds__(256) void k_dog_eta(int nb, const double *part, double *sc, double *eta_out, int kind) {
  __shared__ double red[4];
  double dist2 = 0.0, g2 = 0.0;
  for (int b = threadIdx.x; b < nb; b += 256) {   // (a single thread walking 512 dependent loads took ~50 us)
    dist2 += part[2 * b];
    g2 += part[2 * b + 1];
  }
  dist2 = block_sum<double, 256>(dist2, red);
  g2 = block_sum<double, 256>(g2, red);
  if (threadIdx.x != 0) return;
  double r = sc[1], v = sc[0];
  r = fmax(sqrt(dist2), r);
  double eta;
  if (kind == 1) {  // DoWG
    const double r2 = r * r;
    v = v + r2 * g2;
    eta = r2 / sqrt(v);
  } else {          // DoG
    v = v + g2;
    eta = r / sqrt(v);
  }
  sc[0] = v;
  sc[1] = r;
  *eta_out = eta;
}
template <typename T>
__global__ void k_dog_apply(int64_t n, T *params, const T *grad, const double *eta_ptr) {
  const double eta = *eta_ptr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    params[i] = (T)((double)params[i] - eta * (double)grad[i]);
}
// the same update with the operator (ClipScale) and the averager (PolynomialAveraging) of the iteration folded in --
// one pass over the parameters instead of three launches; element for element the arithmetic of the separate kernels
template <typename T>
__global__ void k_dog_apply_fused(int64_t n, T *params, const T *grad, const double *eta_ptr, int d, int family, T clip_eps,
                                  T *avg, double avg_eta, const long long *t_ptr, long long t_base) {
  const double eta = *eta_ptr;
  double wa = 0.0, wb = 0.0;
  if (avg) {
    const double t = (double)(t_base + (t_ptr ? *t_ptr : 0));
    wa = (avg_eta + 1.0) / (t + avg_eta);
    wb = 1.0 - wa;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    T x = (T)((double)params[i] - eta * (double)grad[i]);
    if (clip_eps == clip_eps && is_scale_diag(i, d, family)) x = clip_step(x, clip_eps);
    params[i] = x;
    if (avg) avg[i] = poly_avg_step<T>(x, avg[i], wa, wb);
  }
}

// DoG / DoWG step with ClipScale and PolynomialAveraging folded into the apply pass (device-resident loop, large vectors).
// Always handles the step (returns true); kept as a bool for callers that fall back to separate launches.
bool launch_dog_update_fused(mivi_ctx *c, void *params, const void *grad, void *state, int kind, double clip_eps, void *avg,
                             double avg_eta, const long long *t_ptr, long long t_base) {
  const int64_t n = mivi_params_len(c);
  double *sc = (double *)((char *)state + mivi_dog_state_bytes(c) - 16);
  if (!(n > 16384 && c->dog_part.p)) {   // small vectors: the single-workgroup kernel, with the same fused tail
    if (c->cfg.dtype == MIVI_F32)
      hipLaunchKernelGGL((k_dog_update<float, true>), dim3(1), dim3(1024), 0, c->stream, n, (float *)params, (const float *)grad,
                         (const float *)state, sc, kind, c->cfg.d, c->cfg.family, (float)clip_eps, (float *)avg, avg_eta, t_ptr, t_base);
    else
      hipLaunchKernelGGL((k_dog_update<double, true>), dim3(1), dim3(1024), 0, c->stream, n, (double *)params, (const double *)grad,
                         (const double *)state, sc, kind, c->cfg.d, c->cfg.family, clip_eps, (double *)avg, avg_eta, t_ptr, t_base);
    return true;
  }
  const int nb = 512;
  double *part = (double *)c->dog_part.p, *eta = part + 2 * nb;
  if (c->cfg.dtype == MIVI_F32) {
    hipLaunchKernelGGL(k_dog_norms<float>, dim3(nb), dim3(256), 0, c->stream, n, (const float *)params, (const float *)grad,
                       (const float *)state, part);
    hipLaunchKernelGGL(k_dog_eta, dim3(1), dim3(256), 0, c->stream, nb, part, sc, eta, kind);
    hipLaunchKernelGGL(k_dog_apply_fused<float>, dim3(2048), dim3(256), 0, c->stream, n, (float *)params, (const float *)grad,
                       eta, c->cfg.d, c->cfg.family, (float)clip_eps, (float *)avg, avg_eta, t_ptr, t_base);
  } else {
    hipLaunchKernelGGL(k_dog_norms<double>, dim3(nb), dim3(256), 0, c->stream, n, (const double *)params, (const double *)grad,
                       (const double *)state, part);
    hipLaunchKernelGGL(k_dog_eta, dim3(1), dim3(256), 0, c->stream, nb, part, sc, eta, kind);
    hipLaunchKernelGGL(k_dog_apply_fused<double>, dim3(2048), dim3(256), 0, c->stream, n, (double *)params,
                       (const double *)grad, eta, c->cfg.d, c->cfg.family, clip_eps, (double *)avg, avg_eta, t_ptr, t_base);
  }
  return true;
}

void launch_dog_update(mivi_ctx *c, void *params, const void *grad, void *state, int kind) {
  const int64_t n = mivi_params_len(c);
  double *sc = (double *)((char *)state + mivi_dog_state_bytes(c) - 16);
  if (n > 16384 && c->dog_part.p) {
    const int nb = 512;
    double *part = (double *)c->dog_part.p, *eta = part + 2 * nb;
    if (c->cfg.dtype == MIVI_F32) {
      hipLaunchKernelGGL(k_dog_norms<float>, dim3(nb), dim3(256), 0, c->stream, n, (const float *)params, (const float *)grad,
                         (const float *)state, part);
      hipLaunchKernelGGL(k_dog_eta, dim3(1), dim3(256), 0, c->stream, nb, part, sc, eta, kind);
      hipLaunchKernelGGL(k_dog_apply<float>, dim3(2048), dim3(256), 0, c->stream, n, (float *)params, (const float *)grad, eta);
    } else {
      hipLaunchKernelGGL(k_dog_norms<double>, dim3(nb), dim3(256), 0, c->stream, n, (const double *)params,
                         (const double *)grad, (const double *)state, part);
      hipLaunchKernelGGL(k_dog_eta, dim3(1), dim3(256), 0, c->stream, nb, part, sc, eta, kind);
      hipLaunchKernelGGL(k_dog_apply<double>, dim3(2048), dim3(256), 0, c->stream, n, (double *)params, (const double *)grad, eta);
    }
    return;
  }
  if (c->cfg.dtype == MIVI_F32)
    hipLaunchKernelGGL(k_dog_update<float>, dim3(1), dim3(1024), 0, c->stream, n, (float *)params, (const float *)grad,
                       (const float *)state, sc, kind);
  else
    hipLaunchKernelGGL(k_dog_update<double>, dim3(1), dim3(1024), 0, c->stream, n, (double *)params, (const double *)grad,
                       (const double *)state, sc, kind);
}

}  // namespace mivi

extern "C" {
int64_t mivi_dog_state_bytes(const mivi_ctx_t *c) {
  const int64_t b = mivi_params_len(c) * (int64_t)c->esize;
  return (b + 7) / 8 * 8 + 16;
}
mivi_status_t mivi_axpby(mivi_ctx_t *c, void *y, double a, const void *x, double b, int64_t n) {
  if (!c || !y || !x || n < 0) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  mivi::launch_axpby(c, y, a, x, b, n);
  return hipGetLastError() == hipSuccess ? MIVI_OK : MIVI_ERR_HIP;
}
mivi_status_t mivi_dog_init(mivi_ctx_t *c, const void *params, void *state, double alpha) {
  if (!c || !params || !state) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  mivi::launch_dog_init(c, params, state, alpha);
  return hipGetLastError() == hipSuccess ? MIVI_OK : MIVI_ERR_HIP;
}
mivi_status_t mivi_dog_update(mivi_ctx_t *c, void *params, const void *grad, void *state, int32_t kind) {
  if (!c || !params || !grad || !state || (kind != 0 && kind != 1)) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  mivi::launch_dog_update(c, params, grad, state, kind);
  return hipGetLastError() == hipSuccess ? MIVI_OK : MIVI_ERR_HIP;
}
}
