// Launch-free optimisation loop for SMALL hierarchical logistic regressions with few samples per step: the reference README's own example
// (README.md:42-119: the sonar data set, n = 208 rows, 60 features, theta = [beta; sigma] behind the exp bijector, full-rank or mean-field q,
// KLMinRepGradProxDescent, one sample per step -- the reference's default) in ONE workgroup; larger data sets (n (d - 1) n_mc <= 2^20) on up to
// 64 workgroups that split the rows of X and exchange their partial sums once per step.  (d - 1) n_mc <= 256: beyond that the graph of launches
// wins (lr_small_loop_ok; BASELINE configs[0], n = 1000, d = 32, 16 samples, stays there).
//
// At these sizes a step of the general route is eight to ten launches (draws, z, logits, X^T r, finish, the reduction, two or three
// optimiser launches, operator, averager) that are all launch latency: 30-36 us per step whatever the rule.  The whole problem fits ONE
// workgroup: n_steps iterations of `step` (src/algorithms/common.jl:69-104) -- estimate_gradient! (src/algorithms/repgradelbo.jl:151-177),
// Optimisers.update!, operator, averager -- run inside one kernel with the parameters and the optimiser state in registers (a thread owns up
// to nine entries of [mu; packed tril(C)] or [mu; sigma]), C / eps / z / W in LDS, X resident in LDS where it fits (streamed through the L2
// otherwise).  Per step:
//   z = mu + tril(C) eps  (mean-field: mu + sigma .* eps)                                  src/families/location_scale.jl:71-87
//   the target, a chunk of rows at a time: logit = X beta, r = y - sigmoid(logit), ll += y logit - softplus(logit);  g_beta = likeadj X' r
//     - beta / sigma^2, the sigma entry and the priors of the two variants exactly as k_lr_finish has them (kernels_targets.hip; variant 0:
//     docs/src/tutorials/subsampling.md:26-38, variant 1: README.md:42-66 inside the TransformedLogDensityProblem of README.md:91-106)
//   d/dmu = -(1/M) W 1,  d/dC = -(1/M) tril(W eps') - direct diag(1 / C_ii)  (mean-field: d/dsigma_i = -(1/M) sum_m W_im eps_im - direct / sigma_i)
//   every rule x operator x averager of the reference's algorithms (Descent / Adam / DoG / DoWG x Identity / ClipScale /
//     ProximalLocationScaleEntropy x No / PolynomialAveraging: optim_rules.h); DoG / DoWG's two norms are block sums -- one workgroup.
// Same eps stream, same update rules, same closed-form gradient as every other route; the sums are sequential multiply-adds in a fixed order
// (deterministic), not the tile kernels' chains: a trajectory equals the launch-per-step one to rounding
// (tests/test_gpu_optimize.py::test_logreg_small_loop).  Not for the sticking-the-landing estimators, a generic Stacked bijector, sharded
// contexts or minibatch views; MIVI_NO_FUSED_LOOP=1 keeps the graph of launches.
#include <cstdlib>

#include "device_common.h"
#include "optim_rules.h"

namespace mivi {

constexpr int kLrSmallD = 64, kLrSmallM = 64, kLrSmallNE = 9, kLrSmallGPT = 4, kLrSmallNT = 256;

template <typename T>
struct LrSmallLoopArgs {
  int family, d, M, n_steps, rule, ent_kind, m_offset, M_total, variant, x_resident, ldx, rc;
  long long n;
  T *params, *opt_state;
  const T *X;                      // n x p column-major
  const uint8_t *y;
  uint64_t seed, idx0;
  long long t0;
  double eta, clip_eps, b1, b2, adam_eps, likeadj;
  double *elbo;
  T *value;
  int *status;
  int op, averager;
  double avg_eta;
  T *avg;
  const T *x0;
  double *dog_sc;
  double *part;                    // G > 1: [n_steps][G][pstride] partial sums, NaN until delivered
  int pstride, spin;
};

template <typename T>
__device__ __forceinline__ T lrs_softplus(T x) { return x > T(0) ? x + log1p(exp(-x)) : log1p(exp(x)); }

template <typename T, int RULE, bool XRES>   // XRES: X lives in LDS (no generic pointers: a pointer that may be LDS or global makes every load a flat load)
__global__ __launch_bounds__(kLrSmallNT) void k_lr_small_loop(LrSmallLoopArgs<T> a) {
  constexpr int NT = kLrSmallNT, NE = kLrSmallNE, GPT = kLrSmallGPT;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int d = a.d, p = d - 1, M = a.M, d4 = (d + 3) >> 2, RC = a.rc;
  const bool fr = a.family == MIVI_FULLRANK;
  const int nl = fr ? d * (d + 1) / 2 : d, ne = d + nl;
  // workgroup wb of G works on rows [r_lo, r_hi) of X (G > 1: the partial log-likelihoods and X' r sums are exchanged once per step, below)
  const long long nstr = a.n;
  const int G = gridDim.x, wb = blockIdx.x;
  const long long r_lo = nstr * wb / G, r_hi = nstr * (wb + 1) / G, n = r_hi - r_lo;
  const T *Xg = a.X + r_lo;
  const uint8_t *yg = a.y + r_lo;
  // LDS carve-up
  double *red = reinterpret_cast<double *>(lds_raw);          // [16] block sums, [16 ..) per-sample scalars
  double *llw = red + 16;                                      // [M][4] wave partials of the log-likelihood of a chunk
  double *ellm = llw + (size_t)kLrSmallM * 4;                  // [M] ell of the samples; [M ..) bb
  T *Cs = reinterpret_cast<T *>(ellm + 2 * kLrSmallM);         // full-rank: C[k d + i]; mean-field: sigma[i]
  T *mus = Cs + (fr ? d * d : d);
  T *E = mus + d;                                              // eps[m d + i]
  T *Zl = E + (size_t)M * d;                                   // z[m d + i]
  T *Wl = Zl + (size_t)M * d;                                  // W[m d + i]
  T *Rc = Wl + (size_t)M * d;                                  // resid[m RC + r] of the current row chunk
  T(*cc_tab)[2] = reinterpret_cast<T(*)[2]>(Rc + (size_t)M * RC);   // [256][2]
  T *Xs = reinterpret_cast<T *>(cc_tab + NT);                  // X resident: column k at Xs[k ldx ..], ldx odd
  uint8_t *ys = reinterpret_cast<uint8_t *>(Xs + (XRES ? (size_t)p * a.ldx : 0));   // y resident (n bytes): no global load inside the loop
  const double direct = direct_entropy_coeff(a.ent_kind);
  const double invM = 1.0 / (double)a.M_total;
  const T eta = (T)a.eta, b1 = (T)a.b1, b2 = (T)a.b2, aeps = (T)a.adam_eps, ceps = (T)a.clip_eps;
  const bool clip = a.op == 1 && a.clip_eps == a.clip_eps, prox = a.op == 2, averaging = a.averager == 1;
  const size_t plen = fr ? (size_t)d + (size_t)d * d : 2 * (size_t)d;
  int SPL = 1;   // threads per output of X' r (a power of two, at most 8; groups never straddle a wave)
  while (SPL < 8 && 2 * SPL * p * M <= NT) SPL *= 2;

  // this thread's entries: e < d: mu_e; full-rank: packed lower entry e - d = j d - j (j - 1) / 2 + (i - j); mean-field: sigma_{e - d}
  int ej[NE], ei[NE];
  size_t ep[NE];
  bool eok[NE];
  T px[NE], pm[NE], pv[NE], pa[NE];
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    const int e = tid + u * NT;
    eok[u] = e < ne;
    ej[u] = -1; ei[u] = eok[u] ? e : 0;
    if (eok[u] && e >= d) {
      if (fr) {
        int j = 0, r = e - d;
        while (r >= d - j) { r -= d - j; ++j; }
        ej[u] = j; ei[u] = j + r;
      } else {
        ej[u] = e - d; ei[u] = e - d;
      }
    }
    ep[u] = ej[u] < 0 ? (size_t)ei[u] : (fr ? (size_t)d + (size_t)ej[u] * d + ei[u] : (size_t)d + ei[u]);
    px[u] = eok[u] ? a.params[ep[u]] : T(0);
    pm[u] = (RULE == 1 && eok[u]) ? a.opt_state[ep[u]] : ((RULE >= 2 && eok[u]) ? a.x0[ep[u]] : T(0));
    pv[u] = (RULE == 1 && eok[u]) ? a.opt_state[plen + ep[u]] : T(0);
    pa[u] = (averaging && eok[u]) ? a.avg[ep[u]] : T(0);
  }
  double dog_v = 0.0, dog_r = 0.0;
  if (RULE >= 2) { dog_v = a.dog_sc[0]; dog_r = a.dog_sc[1]; }
  if (fr)
    for (int i = tid; i < d * d; i += NT) Cs[i] = T(0);
  for (long long i = tid; i < n; i += NT) ys[i] = yg[i];
  if (XRES)
    for (long long i = tid; i < n * p; i += NT) {
      const long long k = i / n, r = i - k * n;
      Xs[(size_t)k * a.ldx + r] = Xg[(size_t)k * nstr + r];
    }
  __syncthreads();

  for (int t = 0; t < a.n_steps; ++t) {
    if (RULE == 1 && (t & (NT - 1)) == 0) adam_bias<T>(a.t0 + t + tid + 1, a.b1, a.b2, cc_tab[tid][0], cc_tab[tid][1]);
    // parameters of this step -> LDS; the draws
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      if (!eok[u]) continue;
      if (ej[u] < 0) mus[ei[u]] = px[u];
      else if (fr) Cs[ej[u] * d + ei[u]] = px[u];
      else Cs[ei[u]] = px[u];
    }
    for (int b = tid; b < d4 * M; b += NT) {
      const int m = b / d4, q = b - m * d4;
      T e[4];
      eps_block<T>(a.seed, a.idx0 + (uint64_t)t, (uint64_t)(a.m_offset + m) * (uint64_t)d4 + (uint64_t)q, e);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * q + r < d) E[m * d + 4 * q + r] = e[r];
    }
    __syncthreads();
    // z = mu + tril(C) eps (mean-field: mu + sigma eps)
    T he = 0;
    for (int o = tid; o < d * M; o += NT) {
      const int m = o / d, i = o - m * d;
      T z = mus[i];
      if (fr) {
#pragma unroll 8
        for (int k = 0; k <= i; ++k) z = fma(Cs[k * d + i], E[m * d + k], z);
      } else {
        z = fma(Cs[i], E[o], z);
      }
      Zl[o] = z;
      const T er = E[o];
      he = fma(T(0.5) * er, er, he);
    }
    // this thread's outputs of X' r: (k, m) = (o % p, o / p), o = tid + u NT
    T gacc[GPT];
#pragma unroll
    for (int u = 0; u < GPT; ++u) gacc[u] = T(0);
    double ll_own = 0.0;   // threads tid < M: the log-likelihood of sample tid
    __syncthreads();
    // ---- the target, RC rows at a time ----------------------------------------------------------------------------------------------------
    for (long long r0 = 0; r0 < n; r0 += RC) {
      const long long r = r0 + tid;
      const bool rok = tid < RC && r < n;
      const T yv = rok ? (T)ys[r] : T(0);
      for (int mb = 0; mb < M; mb += 8) {
        T acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = T(0);
        if (rok) {
          // (three bodies: a per-element test of mb + j < M inside the k loop is eight scalar branches per column -- 480 per row at p = 60)
          const int jn = M - mb < 8 ? M - mb : 8;
          if (jn == 8) {
#pragma unroll 2
            for (int k = 0; k < p; ++k) {
              const T x = XRES ? Xs[(size_t)k * a.ldx + r] : Xg[(size_t)k * nstr + r];
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[j] = fma(x, Zl[(mb + j) * d + k], acc[j]);
            }
          } else if (jn == 1) {
            // one sample: four interleaved partial sums (columns k mod 4), so that four columns' LDS reads are in flight instead of one chain
            T a4[4] = {T(0), T(0), T(0), T(0)};
            int k = 0;
            for (; k + 4 <= p; k += 4) {
#pragma unroll
              for (int u = 0; u < 4; ++u)
                a4[u] = fma(XRES ? Xs[(size_t)(k + u) * a.ldx + r] : Xg[(size_t)(k + u) * nstr + r], Zl[mb * d + k + u], a4[u]);
            }
            for (; k < p; ++k) a4[0] = fma(XRES ? Xs[(size_t)k * a.ldx + r] : Xg[(size_t)k * nstr + r], Zl[mb * d + k], a4[0]);
            acc[0] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
          } else {
            for (int k = 0; k < p; ++k) {
              const T x = XRES ? Xs[(size_t)k * a.ldx + r] : Xg[(size_t)k * nstr + r];
#pragma unroll
              for (int j = 0; j < 8; ++j)
                if (j < jn) acc[j] = fma(x, Zl[(mb + j) * d + k], acc[j]);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (mb + j >= M) break;   // (uniform)
          T llv = T(0);
          if (rok) {
            const T lg = acc[j];
            llv = yv * lg - lrs_softplus(lg);
            Rc[(mb + j) * RC + tid] = yv - T(1) / (T(1) + exp(-lg));
          } else if (tid < RC) {
            Rc[(mb + j) * RC + tid] = T(0);
          }
          const double s = wave_sum_fast(llv);
          if (lane == 0) llw[(mb + j) * 4 + wv] = s;
        }
      }
      __syncthreads();
      if (tid < M) ll_own += (llw[tid * 4 + 0] + llw[tid * 4 + 1]) + (llw[tid * 4 + 2] + llw[tid * 4 + 3]);
      const int rcn = (int)((n - r0) < RC ? (n - r0) : RC);
      if (SPL > 1) {   // few outputs: SPL threads share one output's rows (interleaved), their partial sums folded by a fixed xor tree
        const int o = tid / SPL, part = tid - o * SPL;
        T g = T(0);
        if (o < p * M) {
          const int m = o / p, k = o - m * p;
          const T *rr = Rc + m * RC;
          T g1 = T(0);   // (two interleaved partial sums per thread: two rows' reads in flight)
          if (XRES) {
            const T *xc = Xs + (size_t)k * a.ldx + r0;
            int q = part;
            for (; q + SPL < rcn; q += 2 * SPL) { g = fma(xc[q], rr[q], g); g1 = fma(xc[q + SPL], rr[q + SPL], g1); }
            if (q < rcn) g = fma(xc[q], rr[q], g);
          } else {
            const T *xc = Xg + (size_t)k * nstr + r0;
            int q = part;
            for (; q + SPL < rcn; q += 2 * SPL) { g = fma(xc[q], rr[q], g); g1 = fma(xc[q + SPL], rr[q + SPL], g1); }
            if (q < rcn) g = fma(xc[q], rr[q], g);
          }
          g += g1;
        }
        for (int w2 = SPL >> 1; w2 > 0; w2 >>= 1) g += __shfl_xor(g, w2, 64);
        gacc[0] += g;   // (every thread of the group holds the output's sum)
      } else {
#pragma unroll
        for (int u = 0; u < GPT; ++u) {
          const int o = tid + u * NT;
          if (o >= p * M) break;
          const int m = o / p, k = o - m * p;
          const T *rr = Rc + m * RC;
          T g = gacc[u];
          if (XRES) {
            const T *xc = Xs + (size_t)k * a.ldx + r0;
#pragma unroll 8
            for (int q = 0; q < rcn; ++q) g = fma(xc[q], rr[q], g);
          } else {
            const T *xc = Xg + (size_t)k * nstr + r0;
#pragma unroll 8
            for (int q = 0; q < rcn; ++q) g = fma(xc[q], rr[q], g);
          }
          gacc[u] = g;
        }
      }
      __syncthreads();
    }
    if (G > 1) {
      // every workgroup's partial sums -> slots of this step's own, NaN until stored (the data are their own flags; a NaN sum travels as +Inf);
      // then every workgroup adds all of them in workgroup order: the same bits everywhere, so the replicated parameters stay identical
      double *mine = a.part + ((size_t)t * G + wb) * a.pstride;
      const bool writer = SPL > 1 ? (tid % SPL == 0 && tid / SPL < p * M) : true;
#pragma unroll
      for (int u = 0; u < GPT; ++u) {
        const int o = SPL > 1 ? tid / SPL : tid + u * NT;
        if ((SPL > 1 && u > 0) || o >= p * M || !writer) break;
        const double v = (double)gacc[u];
        __hip_atomic_store(mine + o, v == v ? v : (double)INFINITY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (tid < M) __hip_atomic_store(mine + p * M + tid, ll_own == ll_own ? ll_own : (double)INFINITY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool ok = true;
      auto gather = [&](int o) {   // eight workgroups' slots in flight at a time (one load after the other is a memory round trip each)
        double sum = 0.0;
        for (int w0 = 0; w0 < G; w0 += 8) {
          double q[8];
#pragma unroll
          for (int j = 0; j < 8; ++j)
            q[j] = w0 + j < G ? __hip_atomic_load(a.part + ((size_t)t * G + w0 + j) * a.pstride + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (w0 + j < G && !(q[j] == q[j])) {
              const double *pp = a.part + ((size_t)t * G + w0 + j) * a.pstride + o;
              int budget = a.spin;
              while (true) {
                q[j] = __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (q[j] == q[j]) break;
                if (--budget <= 0) { ok = false; q[j] = 0.0; break; }
                __builtin_amdgcn_s_sleep(1);
              }
            }
            sum += q[j];
          }
        }
        return sum;
      };
#pragma unroll
      for (int u = 0; u < GPT; ++u) {
        const int o = SPL > 1 ? tid / SPL : tid + u * NT;
        if ((SPL > 1 && u > 0) || o >= p * M) break;
        gacc[u] = (T)gather(o);
      }
      if (tid < M) ll_own = gather(p * M + tid);
      if (__syncthreads_or(ok ? 0 : 1)) {
        if (tid == 0) atomicOr(a.status, 8);
        break;
      }
    }
    // ---- priors, the sigma entry, ell per sample (k_lr_finish's arithmetic) -----------------------------------------------------------------
    if (tid < M) {
      const int m = tid;
      double bb = 0.0;
#pragma unroll 8
      for (int k = 0; k < p; ++k) { const double bk = (double)Zl[m * d + k]; bb += bk * bk; }
      const double sv = (double)Zl[m * d + p];
      const double sigma = exp(sv), inv_s2 = exp(-2.0 * sv);
      const double logprior_beta = -0.5 * p * kLog2Pi - p * sv - 0.5 * bb * inv_s2;
      double gs = -(double)p + bb * inv_s2;
      double logprior_sigma, jac;
      if (a.variant == 0) {
        logprior_sigma = -0.5 * log(2.0 * 3.14159265358979323846 * 9.0) - sigma * sigma / 18.0;
        gs += -(sigma * sigma) / 9.0;
        jac = 0.0;
      } else {
        logprior_sigma = -sv - log(3.0) - 0.5 * kLog2Pi - sv * sv / 18.0;
        gs += -1.0 - sv / 9.0 + 1.0;
        jac = sv;
      }
      ellm[m] = a.likeadj * ll_own + logprior_beta + logprior_sigma + jac;
      ellm[kLrSmallM + m] = inv_s2;
      Wl[m * d + p] = (T)gs;
    }
    __syncthreads();
    if (SPL > 1) {
      const int o = tid / SPL;
      if (o < p * M && tid - o * SPL == 0) {
        const int m = o / p, k = o - m * p;
        Wl[m * d + k] = (T)(a.likeadj * (double)gacc[0] - (double)Zl[m * d + k] * ellm[kLrSmallM + m]);
      }
    } else {
#pragma unroll
      for (int u = 0; u < GPT; ++u) {
        const int o = tid + u * NT;
        if (o >= p * M) break;
        const int m = o / p, k = o - m * p;
        Wl[m * d + k] = (T)(a.likeadj * (double)gacc[u] - (double)Zl[m * d + k] * ellm[kLrSmallM + m]);
      }
    }
    // the step's scalars: sum ell, sum 0.5 eps^2, log|det C| and the positivity check from this step's scale diagonal
    double s_ell, s_he, s_ld, s_bad;
    {
      double v4[4] = {tid < M ? ellm[tid] : 0.0, (double)he, 0.0, 0.0};
      if (tid < d) {
        const T c = fr ? Cs[tid * d + tid] : Cs[tid];
        v4[2] = (double)log(c);
        v4[3] = (c > T(0)) ? 0.0 : 1.0;
      }
      block_sum_n<double, NT, 4>(v4, red);   // (its barriers also publish W)
      s_ell = v4[0]; s_he = v4[1]; s_ld = v4[2]; s_bad = v4[3];
    }
    if (tid == 0 && wb == 0) {
      const double Mt = (double)a.M_total;
      const double ent = (ent_is_closed(a.ent_kind) ? 0.5 * d * (1.0 + kLog2Pi) : s_he / Mt + 0.5 * d * kLog2Pi) + s_ld;
      const double value = -(s_ell / Mt + ent);
      a.elbo[t] = -value;
      if (t == a.n_steps - 1) *a.value = (T)value;
      int st = 0;
      if (!isfinite(value)) st |= 1;
      if (s_bad > 0.0) st |= 2;
      if (st && a.status) atomicOr(a.status, st);
    }
    // ---- gradient entries of this thread, Optimisers.update!, operator, averager ---------------------------------------------------------------
    T gE[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      gE[u] = T(0);
      if (!eok[u]) continue;
      const int i = ei[u], j = ej[u];
      T v = 0;
      if (j < 0) {
#pragma unroll 8
        for (int m = 0; m < M; ++m) v += Wl[m * d + i];
      } else {
#pragma unroll 8
        for (int m = 0; m < M; ++m) v = fma(Wl[m * d + i], E[m * d + j], v);
      }
      double gx = -(double)v * invM;
      if (j >= 0 && i == j) gx -= direct / (double)(fr ? Cs[j * d + j] : Cs[j]);
      gE[u] = (T)gx;
    }
    double e_t = 0.0, gamma = a.eta;
    if (RULE >= 2) {
      double nn[2] = {0.0, 0.0};
#pragma unroll
      for (int u = 0; u < NE; ++u) {
        const double dx = eok[u] ? (double)px[u] - (double)pm[u] : 0.0, gg = (double)gE[u];
        nn[0] += dx * dx;
        nn[1] += gg * gg;
      }
      block_sum_n<double, NT, 2>(nn, red);
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
    for (int u = 0; u < NE; ++u) {
      if (!eok[u]) continue;
      const bool diag = ej[u] >= 0 && ei[u] == ej[u];
      if (RULE == 0) px[u] = descent_step(px[u], gE[u], eta);
      else if (RULE == 1) px[u] = adam_step<T>(px[u], gE[u], pm[u], pv[u], cc_tab[t & (NT - 1)][0], cc_tab[t & (NT - 1)][1], eta, b1, b2, aeps);
      else px[u] = (T)((double)px[u] - e_t * (double)gE[u]);
      if (clip && diag) px[u] = clip_step(px[u], ceps);
      if (prox && diag) px[u] = prox_entropy_step(px[u], (T)gamma);
      if (averaging) pa[u] = poly_avg_step<T>(px[u], pa[u], wa, wb);
    }
    __syncthreads();   // (every thread is done with this step's LDS images)
  }
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    if (!eok[u] || wb != 0) continue;
    a.params[ep[u]] = px[u];
    if (RULE == 1) { a.opt_state[ep[u]] = pm[u]; a.opt_state[plen + ep[u]] = pv[u]; }
    if (averaging) a.avg[ep[u]] = pa[u];
  }
  if (RULE >= 2 && tid == 0 && wb == 0) { a.dog_sc[0] = dog_v; a.dog_sc[1] = dog_r; }
}

// rows of X per chunk: the residuals of a chunk (n_mc x RC) stay small
static int lr_small_rc(int M) { return M <= 16 ? 256 : (M <= 32 ? 128 : 64); }
// workgroups: about 2^14 multiply-adds of the target per workgroup and step, at least 16 rows each, at most 64 (they exchange their partial sums
// every step: all of them resident)
static int lr_small_groups(const mivi_ctx *c) {
  const long long work = (long long)c->lr_n * (c->cfg.d - 1) * c->cfg.n_mc;
  long long g = (work + (1 << 14) - 1) >> 14;
  if (g > 64) g = 64;
  if (g > c->lr_n / 16) g = c->lr_n / 16;
  return g < 1 ? 1 : (int)g;
}
static size_t lr_small_lds(const mivi_ctx *c, int G, bool resident, int *ldx_out) {
  const int d = c->cfg.d, M = c->cfg.n_mc, p = d - 1;
  const size_t es = c->esize;
  const bool fr = c->cfg.family == MIVI_FULLRANK;
  const long long nloc = (c->lr_n + G - 1) / G + 1;   // rows of the largest workgroup
  const int ldx = (int)(nloc | 1);                    // odd: the columns of X start on different banks
  if (ldx_out) *ldx_out = ldx;
  size_t b = (16 + (size_t)kLrSmallM * 4 + 2 * kLrSmallM) * sizeof(double);
  b += ((fr ? (size_t)d * d : (size_t)d) + d + 3 * (size_t)M * d + (size_t)M * lr_small_rc(M) + 2 * kLrSmallNT) * es;
  if (resident) b += (size_t)p * ldx * es;
  b += ((size_t)nloc + 15) & ~(size_t)15;   // y
  return b;
}
size_t lr_small_part_bytes(const mivi_ctx *c, int n_steps) {
  const int G = lr_small_groups(c);
  if (G <= 1) return 0;
  const size_t ps = (((size_t)(c->cfg.d - 1) * c->cfg.n_mc + c->cfg.n_mc) + 15) & ~(size_t)15;
  return (size_t)n_steps * G * ps * sizeof(double);
}
bool lr_small_loop_ok(const mivi_ctx *c) {
  const int d = c->cfg.d, M = c->cfg.n_mc;
  const bool stl = c->cfg.entropy == MIVI_ENT_STL || c->cfg.entropy == MIVI_ENT_STL_ZERO_GRAD;
  if (!(c->target == TGT_LOGREG && !c->bij_on && !stl && d >= 2 && d <= kLrSmallD && M >= 1 && M <= kLrSmallM && c->cfg.m_offset == 0 && c->M_total == M &&
        c->lr_X && c->lr_y && c->lr_n >= 1 && c->lr_route == 0))   // (mivi_set_logreg_route pins the general route's kernels: the graph of launches)
    return false;
  const bool fr = c->cfg.family == MIVI_FULLRANK;
  const long long ne = d + (fr ? (long long)d * (d + 1) / 2 : d);
  if (ne > (long long)kLrSmallNE * kLrSmallNT || (long long)(d - 1) * M > (long long)kLrSmallGPT * kLrSmallNT) return false;
  // One workgroup per ~2^14 multiply-adds of the target (its time grows with them: four waves, every sum a latency chain), up to 64 workgroups
  // that split the rows of X and exchange their partial sums once per step.  Measured (tools/logreg_loop_bench.py, n x (d - 1) x n_mc, us per
  // step, this loop / the graph of launches): 208 x 60 x 1 (one workgroup): 9.4 (full-rank 14.3) / 30-36; 1000 x 32 x 1 (two): 12.5 / 30;
  // 4000 x 16 x 4 (sixteen): 17.5 / 33; 208 x 60 x 8 (seven): 28.9 / 36; 1000 x 32 x 16 (32; BASELINE configs[0]): 35.4 / 31 -- a step's fixed
  // work and the exchanged partials grow with (d - 1) n_mc.  Taken where it wins: (d - 1) n_mc <= 256.
  return (long long)(d - 1) * M <= 256 && (long long)c->lr_n * (d - 1) * M <= (1ll << 20) &&
         lr_small_lds(c, lr_small_groups(c), false, nullptr) <= c->lds_max;
}

template <typename T>
static bool lr_small_loop_impl(mivi_ctx *c, void *params, const mivi_loop_t &l, double *elbo, void *value, double *part) {
  LrSmallLoopArgs<T> a;
  a.family = c->cfg.family; a.d = c->cfg.d; a.M = c->cfg.n_mc; a.n_steps = l.n_steps; a.rule = l.rule; a.ent_kind = c->cfg.entropy;
  a.m_offset = c->cfg.m_offset; a.M_total = c->M_total; a.variant = c->lr_variant; a.n = c->lr_n;
  a.params = (T *)params; a.opt_state = l.rule == 1 ? (T *)l.opt_state_dev : nullptr;
  a.X = (const T *)c->lr_X; a.y = c->lr_y;
  a.seed = c->cfg.seed; a.idx0 = l.estimate_idx0; a.t0 = (long long)l.t0;
  a.eta = l.eta; a.clip_eps = l.clip_epsilon; a.b1 = l.beta1; a.b2 = l.beta2; a.adam_eps = l.adam_eps; a.likeadj = c->lr_likeadj;
  a.elbo = elbo; a.value = (T *)value; a.status = (int *)c->status.p;
  a.op = l.op; a.averager = l.averager; a.avg_eta = l.avg_eta; a.avg = (T *)l.avg_params_dev;
  a.x0 = l.rule >= 2 ? (const T *)l.opt_state_dev : nullptr;
  a.dog_sc = l.rule >= 2 ? (double *)((char *)l.opt_state_dev + mivi_dog_state_bytes(c) - 16) : nullptr;
  a.rc = lr_small_rc(a.M);
  const int G = lr_small_groups(c);
  a.part = part;
  a.pstride = (int)((((size_t)(a.d - 1) * a.M + a.M) + 15) & ~(size_t)15);
  a.spin = 1 << 20;
  if (G > 1) (void)hipMemsetAsync(part, 0xFF, lr_small_part_bytes(c, l.n_steps), c->stream);   // (NaN: not delivered yet)
  int ldx = 0;
  const size_t with_x = lr_small_lds(c, G, true, &ldx);
  a.x_resident = with_x <= c->lds_max ? 1 : 0;
  a.ldx = ldx;
  const size_t lds = a.x_resident ? with_x : lr_small_lds(c, G, false, nullptr);
  bool launched = true;
  auto go = [&](auto kern) {
    // (the LDS must be granted; G > 1 workgroups exchange their partial sums every step by spin-wait: all of them resident -- checked)
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        (G > 1 && !grid_resident(c, reinterpret_cast<const void *>(kern), kLrSmallNT, lds, G))) {
      (void)hipGetLastError();
      launched = false;
      return;
    }
    hipLaunchKernelGGL(kern, dim3(G), dim3(kLrSmallNT), lds, c->stream, a);
  };
  if (a.x_resident) {
    switch (l.rule) {
      case 0: go(k_lr_small_loop<T, 0, true>); break;
      case 1: go(k_lr_small_loop<T, 1, true>); break;
      case 2: go(k_lr_small_loop<T, 2, true>); break;
      default: go(k_lr_small_loop<T, 3, true>); break;
    }
  } else {
    switch (l.rule) {
      case 0: go(k_lr_small_loop<T, 0, false>); break;
      case 1: go(k_lr_small_loop<T, 1, false>); break;
      case 2: go(k_lr_small_loop<T, 2, false>); break;
      default: go(k_lr_small_loop<T, 3, false>); break;
    }
  }
  return launched;
}
// elbo: n_steps doubles; value: one element of T (the last step's objective value)
bool launch_lr_small_loop(mivi_ctx *c, void *params, const mivi_loop_t &l, double *elbo, void *value, double *part) {
  if (c->cfg.dtype == MIVI_F32) return lr_small_loop_impl<float>(c, params, l, elbo, value, part);
  return lr_small_loop_impl<double>(c, params, l, elbo, value, part);
}

}  // namespace mivi
