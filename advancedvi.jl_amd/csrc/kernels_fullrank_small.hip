// Launch-free optimisation loop for SMALL full-rank problems (d <= 32, n_mc <= 64, d x n_mc <= 768; see fr_small_loop_ok): the reference's own benchmark grid
// (bench/benchmarks.jl:43-94: `optimize(alg, 10^4, normal(n_dims = 10), q)` with one sample per step, mean-field and full-rank families,
// ClosedFormEntropy and StickingTheLandingEntropy, Adam(1e-3), ClipScale) and its README-sized neighbours.
//
// At these sizes a step of the general route is two or three launches of tile kernels that are all launch latency (8.8 us per step at
// d = 10, 17 us with the sticking-the-landing solve); the whole problem -- tril(C), eps, W, the optimiser state -- fits one workgroup.  So
// n_steps iterations of {estimate_gradient! (src/algorithms/repgradelbo.jl:151-177), Optimisers.update!, ClipScale}
// (src/algorithms/common.jl:69-104) run inside ONE kernel, ONE workgroup: parameters and optimiser state in registers (a thread owns up to
// three entries of [mu; packed tril(C)]), C / eps / W in LDS, three barriers per step, no memory traffic but the ELBO record.
//   z = mu + tril(C) eps                      src/families/location_scale.jl:71-77
//   W = grad log pi(z) (+ C^-T eps for the sticking-the-landing estimators: one back substitution per sample column, entropy.jl:57-65)
//   d/dmu = -(1/M) W 1,  d/dC = -(1/M) tril(W eps') - direct diag(1 / C_ii)        
// Same eps stream (Philox counter = global column, estimate index), same update rules (optim_rules.h) and the same closed-form gradient as
// every other route; the sums are plain sequential f32 fused multiply-adds instead of MFMA tiles, so a trajectory equals the launch-per-step
// one to rounding, not to the bit (tests/test_gpu_optimize.py::test_small_fullrank_loop states the tolerance; MIVI_NO_FUSED_LOOP=1 keeps
// the graph of launches).
#include <cstdlib>

#include "device_common.h"
#include "optim_rules.h"

namespace mivi {

template <typename T>
struct FrSmallLoopArgs {
  int d, M, n_steps, rule, ent_kind, m_offset, M_total;
  T *params, *opt_state;          // [mu; vec C column-major]; Adam: [m (d + d^2); v (d + d^2)]
  const T *t_mean, *t_istd;
  uint64_t seed, idx0;
  long long t0;
  double eta, clip_eps, b1, b2, adam_eps, ell_const;
  double *elbo;                   // [n_steps]
  T *value;                       // the last step's objective value
  int *status;
  // beyond Descent / Adam + ClipScale (the general loop, mivi_optimize_loop): rule 2 DoG / 3 DoWG (the two norms are block sums: one workgroup),
  // op 2 ProximalLocationScaleEntropy, PolynomialAveraging -- the reference's defaults are DoWG + averaging
  int op, averager;
  double avg_eta;
  T *avg;
  const T *x0;
  double *dog_sc;
};

constexpr int kSmallD = 32, kSmallM = 64, kSmallNE = 3;   // (32 + 528 entries over 256 threads)

template <typename T, int RULE>
__global__ __launch_bounds__(256) void k_fr_small_loop(FrSmallLoopArgs<T> a) {
  constexpr int NT = 256;
  __shared__ T Cs[kSmallD * kSmallD];   // C[k d + i]: column k, row i; zero above the diagonal
  __shared__ T mus[kSmallD], tms[kSmallD], tiss[kSmallD];
  __shared__ T E[kSmallM * kSmallD];    // eps[m d + i]
  __shared__ T Wl[kSmallM * kSmallD];   // W[m d + i]
  __shared__ T U[kSmallM * kSmallD];    // C^-T eps (sticking-the-landing estimators)
  __shared__ double red[4 * (NT / 64)];
  __shared__ T cc_tab[NT][2];
  const int tid = threadIdx.x;
  const int d = a.d, M = a.M, d4 = (d + 3) >> 2, nl = d * (d + 1) / 2, ne = d + nl;
  const bool stl = ent_is_stl(a.ent_kind);
  const double direct = direct_entropy_coeff(a.ent_kind);
  const double invM = 1.0 / (double)a.M_total;
  const T eta = (T)a.eta, b1 = (T)a.b1, b2 = (T)a.b2, aeps = (T)a.adam_eps, ceps = (T)a.clip_eps;
  const bool clip = a.op == 1 && a.clip_eps == a.clip_eps, prox = a.op == 2, averaging = a.averager == 1;   // (NaN clip_eps = no ClipScale)
  __shared__ double nred[2 * (NT / 64)];
  const size_t plen = (size_t)d + (size_t)d * d;

  // this thread's entries: e < d: mu_e; else the packed lower entry e - d = j d - j (j - 1) / 2 + (i - j) (column j, row i >= j)
  int ej[kSmallNE], ei[kSmallNE];
  size_t ep[kSmallNE];               // index inside the dense parameter vector
  bool eok[kSmallNE];
  T px[kSmallNE], pm[kSmallNE], pv[kSmallNE];
#pragma unroll
  for (int u = 0; u < kSmallNE; ++u) {
    const int e = tid + u * NT;
    eok[u] = e < ne;
    ej[u] = -1; ei[u] = eok[u] ? e : 0;
    if (eok[u] && e >= d) {
      int j = 0, r = e - d;
      while (r >= d - j) { r -= d - j; ++j; }
      ej[u] = j; ei[u] = j + r;
    }
    ep[u] = ej[u] < 0 ? (size_t)ei[u] : (size_t)d + (size_t)ej[u] * d + ei[u];
    px[u] = eok[u] ? a.params[ep[u]] : T(0);
    pm[u] = (RULE == 1 && eok[u]) ? a.opt_state[ep[u]] : ((RULE >= 2 && eok[u]) ? a.x0[ep[u]] : T(0));   // (DoG / DoWG: x0)
    pv[u] = (RULE == 1 && eok[u]) ? a.opt_state[plen + ep[u]] : ((averaging && RULE != 1 && eok[u]) ? a.avg[ep[u]] : T(0));   // (... the running average)
  }
  T pa[kSmallNE];   // Adam + PolynomialAveraging: the running average beside the two moments
#pragma unroll
  for (int u = 0; u < kSmallNE; ++u) pa[u] = (RULE == 1 && averaging && eok[u]) ? a.avg[ep[u]] : T(0);
  double dog_v = 0.0, dog_r = 0.0;
  if (RULE >= 2) { dog_v = a.dog_sc[0]; dog_r = a.dog_sc[1]; }
  for (int i = tid; i < d * d; i += NT) Cs[i] = T(0);
  if (tid < d) { tms[tid] = a.t_mean[tid]; tiss[tid] = a.t_istd[tid]; }
  __syncthreads();

  for (int t = 0; t < a.n_steps; ++t) {
    if (RULE == 1 && (t & (NT - 1)) == 0) adam_bias<T>(a.t0 + t + tid + 1, a.b1, a.b2, cc_tab[tid][0], cc_tab[tid][1]);   // (read behind the barriers below)
    // parameters of this step -> LDS
#pragma unroll
    for (int u = 0; u < kSmallNE; ++u) {
      if (!eok[u]) continue;
      if (ej[u] < 0) mus[ei[u]] = px[u];
      else Cs[ej[u] * d + ei[u]] = px[u];
    }
    // the draws: one Philox block = rows 4 q .. 4 q + 3 of column m
    for (int b = tid; b < d4 * M; b += NT) {
      const int m = b / d4, q = b - m * d4;
      T e[4];
      eps_block<T>(a.seed, a.idx0 + (uint64_t)t, (uint64_t)(a.m_offset + m) * (uint64_t)d4 + (uint64_t)q, e);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * q + r < d) E[m * d + 4 * q + r] = e[r];
    }
    __syncthreads();
    // z = mu + tril(C) eps, the fused diagonal-Gaussian target
    T ell = 0, he = 0;
    for (int o = tid; o < d * M; o += NT) {
      const int m = o / d, i = o - m * d;
      T z = mus[i];
#pragma unroll 8
      for (int k = 0; k <= i; ++k) z = fma(Cs[k * d + i], E[m * d + k], z);   // (unrolled: the LDS reads of eight terms in flight, one chain of fmas)
      const T uu = (z - tms[i]) * tiss[i];
      ell = fma(T(-0.5) * uu, uu, ell);
      const T er = E[m * d + i];
      he = fma(T(0.5) * er, er, he);
      Wl[o] = -uu * tiss[i];
    }
    if (stl) {
      // U = C^-T eps, i.e. C^T u = eps (upper triangular), by column-oriented back substitution: 32 lanes per sample column, lane k owns the
      // running right-hand side s_k; row i = d - 1 .. 0: u_i = s_i / C_ii (broadcast from lane i), every lane k < i takes C_ik u_i off its
      // s_k.  One shuffle + one division + one fused multiply-add per row on the dependency chain (a thread per column walked d^2 / 2
      // dependent LDS round trips: 3 us at d = 10).
      const int grp = tid >> 5, k = tid & 31;
      for (int m = grp; m < M; m += NT / 32) {
        T sk = k < d ? E[m * d + k] : T(0);
        for (int i = d - 1; i >= 0; --i) {
          const T ui = __shfl(sk, i, 32) / Cs[i * d + i];
          if (k == i) U[m * d + i] = ui;
          if (k < i) sk = fma(-Cs[k * d + i], ui, sk);
        }
      }
    }
    // the step's four scalars in ONE exchange: sum ell, sum 0.5 eps^2, log|det C| and the positivity check from this step's diagonal
    double s_ell, s_he, s_ld, s_bad;
    {
      T lgv = 0, badv = 0;
      if (tid < d) {
        const T c = Cs[tid * d + tid];
        lgv = log(c);
        badv = (c > T(0)) ? T(0) : T(1);
      }
      double v4[4];
      v4[0] = wave_sum_fast(ell); v4[1] = wave_sum_fast(he); v4[2] = wave_sum_fast(lgv); v4[3] = wave_sum_fast(badv);
      const int lane = tid & 63, wv = tid >> 6;
      if (lane == 0) {   // (`red` was last read before the previous step's closing barrier)
#pragma unroll
        for (int k = 0; k < 4; ++k) red[k * (NT / 64) + wv] = v4[k];
      }
      __syncthreads();   // (also publishes Wl / U)
      s_ell = (red[0] + red[1]) + (red[2] + red[3]);
      s_he = (red[4] + red[5]) + (red[6] + red[7]);
      s_ld = (red[8] + red[9]) + (red[10] + red[11]);
      s_bad = (red[12] + red[13]) + (red[14] + red[15]);
    }
    if (tid == 0) {
      const double Mt = (double)a.M_total;
      const double ent = (ent_is_closed(a.ent_kind) ? 0.5 * d * (1.0 + kLog2Pi) : s_he / Mt + 0.5 * d * kLog2Pi) + s_ld;
      const double value = -((s_ell + (double)M * a.ell_const) / Mt + ent);
      a.elbo[t] = -value;
      if (t == a.n_steps - 1) *a.value = (T)value;
      int st = 0;
      if (!isfinite(value)) st |= 1;
      if (s_bad > 0.0) st |= 2;
      if (st && a.status) atomicOr(a.status, st);
    }
    // gradient entries of this thread + Optimisers.update! + operator + averager
    T gE[kSmallNE];
#pragma unroll
    for (int u = 0; u < kSmallNE; ++u) {
      gE[u] = T(0);
      if (!eok[u]) continue;
      const int i = ei[u], j = ej[u];
      T v = 0;
      if (j < 0) {
#pragma unroll 8
        for (int m = 0; m < M; ++m) v += Wl[m * d + i] + (stl ? U[m * d + i] : T(0));
      } else {
#pragma unroll 8
        for (int m = 0; m < M; ++m) v = fma(Wl[m * d + i] + (stl ? U[m * d + i] : T(0)), E[m * d + j], v);
      }
      double gx = -(double)v * invM;
      if (j >= 0 && i == j) gx -= direct / (double)Cs[j * d + j];
      gE[u] = (T)gx;
    }
    double e_t = 0.0, gamma = a.eta;
    if (RULE >= 2) {   // DoG / DoWG (src/optimization/rules.jl:26-42, :48-64): ||x - x0||^2, ||g||^2 over all parameters, (v, r), the step size
      double nn[2] = {0.0, 0.0};
#pragma unroll
      for (int u = 0; u < kSmallNE; ++u) {
        const double dx = eok[u] ? (double)px[u] - (double)pm[u] : 0.0, gg = (double)gE[u];
        nn[0] += dx * dx;
        nn[1] += gg * gg;
      }
      block_sum_n<double, NT, 2>(nn, nred);
      dog_r = fmax(sqrt(nn[0]), dog_r);
      if (RULE == 3) {
        const double r2 = dog_r * dog_r;
        dog_v = dog_v + r2 * nn[1];
        e_t = r2 / sqrt(dog_v);
      } else {
        dog_v = dog_v + nn[1];
        e_t = dog_r / sqrt(dog_v);
      }
      gamma = e_t;
    }
    const double tt = (double)(a.t0 + t + 1);
    const double wa = (a.avg_eta + 1.0) / (tt + a.avg_eta), wb = 1.0 - wa;
#pragma unroll
    for (int u = 0; u < kSmallNE; ++u) {
      if (!eok[u]) continue;
      const int i = ei[u], j = ej[u];
      const bool diag = j >= 0 && i == j;
      if (RULE == 0) px[u] = descent_step(px[u], gE[u], eta);
      else if (RULE == 1) px[u] = adam_step<T>(px[u], gE[u], pm[u], pv[u], cc_tab[t & (NT - 1)][0], cc_tab[t & (NT - 1)][1], eta, b1, b2, aeps);
      else px[u] = (T)((double)px[u] - e_t * (double)gE[u]);
      if (clip && diag) px[u] = clip_step(px[u], ceps);
      if (prox && diag) px[u] = prox_entropy_step(px[u], (T)gamma);
      if (averaging) {
        if (RULE == 1) pa[u] = poly_avg_step<T>(px[u], pa[u], wa, wb);
        else pv[u] = poly_avg_step<T>(px[u], pv[u], wa, wb);
      }
    }
    __syncthreads();   // (every thread is done with this step's LDS images)
  }
#pragma unroll
  for (int u = 0; u < kSmallNE; ++u) {
    if (!eok[u]) continue;
    a.params[ep[u]] = px[u];
    if (RULE == 1) {
      a.opt_state[ep[u]] = pm[u];
      a.opt_state[plen + ep[u]] = pv[u];
    }
    if (averaging) a.avg[ep[u]] = RULE == 1 ? pa[u] : pv[u];
  }
  if (RULE >= 2 && tid == 0) { a.dog_sc[0] = dog_v; a.dog_sc[1] = dog_r; }
}

// Where ONE workgroup beats the graph of launches (tools/small_loop_bench.py, us per step, this loop / the graph): d x n_mc = 10 x 1: 2.3 / 8.8
// (STL 3.9 / 17.2), 16 x 32: 4.9 / 9.0 (STL 15.5 / 18.5), 32 x 16: 6.3 / 9.7 (STL 17.2 / 19.0), 10 x 64: 7.2 / 9.1 (STL 20.4 / 19.1),
// 32 x 32: 10.0 / 9.6, 32 x 64: 17.8 / 9.9 -- the single workgroup's time grows with d x n_mc, a launch's hardly does.
bool fr_small_loop_ok(const mivi_ctx *c) {
  const bool stl = c->cfg.entropy == MIVI_ENT_STL || c->cfg.entropy == MIVI_ENT_STL_ZERO_GRAD;
  const long long dm = (long long)c->cfg.d * c->cfg.n_mc;
  return c->cfg.family == MIVI_FULLRANK && c->target == TGT_DIAG_GAUSS && !c->bij_on && c->cfg.d <= kSmallD && c->cfg.n_mc <= kSmallM &&
         dm <= (stl ? 512 : 768) && c->cfg.m_offset == 0 && c->M_total == c->cfg.n_mc;
}

template <typename T>
static void fr_small_loop_impl(mivi_ctx *c, void *params, void *opt_state, uint64_t idx0, long long t0, int n_steps, int rule, double eta,
                               double clip_eps, double *elbo, void *value, const mivi_loop_t *gen) {
  FrSmallLoopArgs<T> a;
  a.d = c->cfg.d; a.M = c->cfg.n_mc; a.n_steps = n_steps; a.rule = rule; a.ent_kind = c->cfg.entropy;
  a.m_offset = c->cfg.m_offset; a.M_total = c->M_total;
  a.params = (T *)params; a.opt_state = (T *)opt_state;
  a.t_mean = (const T *)c->t_mean.p; a.t_istd = (const T *)c->t_istd.p;
  a.seed = c->cfg.seed; a.idx0 = idx0; a.t0 = t0;
  a.eta = eta; a.clip_eps = clip_eps; a.b1 = 0.9; a.b2 = 0.999; a.adam_eps = 1e-8; a.ell_const = c->t_const;
  a.elbo = elbo; a.value = (T *)value; a.status = (int *)c->status.p;
  a.op = (clip_eps == clip_eps) ? 1 : 0; a.averager = 0; a.avg_eta = 0.0; a.avg = nullptr; a.x0 = nullptr; a.dog_sc = nullptr;
  if (gen) {
    a.op = gen->op; a.averager = gen->averager; a.avg_eta = gen->avg_eta; a.avg = (T *)gen->avg_params_dev;
    a.b1 = gen->beta1; a.b2 = gen->beta2; a.adam_eps = gen->adam_eps;
    if (rule >= 2) {
      a.x0 = (const T *)gen->opt_state_dev;
      a.dog_sc = (double *)((char *)gen->opt_state_dev + mivi_dog_state_bytes(c) - 16);
    }
  }
  if (rule == 0) hipLaunchKernelGGL((k_fr_small_loop<T, 0>), dim3(1), dim3(256), 0, c->stream, a);
  else if (rule == 1) hipLaunchKernelGGL((k_fr_small_loop<T, 1>), dim3(1), dim3(256), 0, c->stream, a);
  else if (rule == 2) hipLaunchKernelGGL((k_fr_small_loop<T, 2>), dim3(1), dim3(256), 0, c->stream, a);
  else hipLaunchKernelGGL((k_fr_small_loop<T, 3>), dim3(1), dim3(256), 0, c->stream, a);
}
// rule 0 Descent / 1 Adam (default betas); elbo: n_steps doubles; value: one element of T
void launch_fr_small_loop(mivi_ctx *c, void *params, void *opt_state, uint64_t idx0, long long t0, int n_steps, int rule, double eta,
                          double clip_eps, double *elbo, void *value, const mivi_loop_t *gen) {
  if (c->cfg.dtype == MIVI_F32) fr_small_loop_impl<float>(c, params, opt_state, idx0, t0, n_steps, rule, eta, clip_eps, elbo, value, gen);
  else fr_small_loop_impl<double>(c, params, opt_state, idx0, t0, n_steps, rule, eta, clip_eps, elbo, value, gen);
}

}  // namespace mivi
