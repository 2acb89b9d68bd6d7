// Second-order capability of the built-in logistic-regression and funnel targets (gfx950): the sample average of the Hessians that
// `gaussian_expectation_gradient_and_hessian!` takes for a target whose capabilities exceed LogDensityOrder{1}
// (src/algorithms/gauss_expected_grad_hess.jl:61-83), beside the constant Hessians of the Gaussian targets (kernels_fullrank.hip k_const_hess).
// Both Hessians are LINEAR in per-sample statistics, so the average over the n samples never forms a per-sample matrix:
//   logistic regression, theta = [beta (p); s], sigma = e^s (the targets of README.md:42-66 / docs/src/tutorials/subsampling.md:26-38):
//     d2/dbeta dbeta' = -likeadj X' diag(pi_i (1 - pi_i)) X - I / sigma^2        pi_i = logistic(x_i' beta)
//       => mean over samples = -likeadj X' diag(wbar) X - mean(sigma^-2) I,      wbar_i = mean_m pi_im (1 - pi_im):  ONE weighted Gram matrix
//     d2/dbeta ds = 2 beta / sigma^2,   d2/ds2 = -2 beta'beta / sigma^2 + {logsigma_normal: -2 sigma^2 / 9 | lognormal_exp_bijector: -1 / 9}
//   funnel, eta = [v; x] (v = log s under Stacked([log, identity])): log pi = -v^2 / (2 sv^2) - (d - 1) v - e^{-2v} |x|^2 / 2 + c
//     d2/dv2 = -1 / sv^2 - 2 e^{-2v} |x|^2,   d2/dv dx_i = 2 e^{-2v} x_i,   d2/dx_i2 = -e^{-2v}          (an arrow matrix)
//   funnel on the constrained scale, theta = [s; x]: log pi = -log s - (log s)^2 / (2 sv^2) - (d - 1) log s - |x|^2 / (2 s^2) + c
//     d2/ds2 = d / s^2 - (1 - log s) / (sv^2 s^2) - 3 |x|^2 / s^4,   d2/ds dx_i = 2 x_i / s^3,   d2/dx_i2 = -1 / s^2
// Sums are accumulated in f64 (atomics on doubles: order-dependent in the last bits, like every cross-workgroup f64 sum here that is
// not part of a bitwise-reproducible path); chunks of samples add up, the finisher divides by n once.
#include "device_common.h"
#include "mivi_internal.h"

namespace mivi {

namespace {

template <typename T>
__device__ __forceinline__ T logistic_t(T x) {
  return x >= T(0) ? T(1) / (T(1) + exp(-x)) : exp(x) / (T(1) + exp(x));
}

// wsum[r] += sum over the samples of this column block of pi (1 - pi), pi = logistic(x_r' beta_m).  64 rows x 64 samples per workgroup
// (the tiling of k_lr_logits: X column-major n x p, Z column-major d x M).
template <typename T>
__global__ __launch_bounds__(256) void k_h2_lr_wsum(int d, int p, int M, int64_t n, const T *__restrict__ X, const T *__restrict__ Z, double *__restrict__ wsum) {
  __shared__ T As[16][65];
  __shared__ T Bs[16][65];
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  const int m0 = blockIdx.y * 64;
  const int tr = tid & 15, tc = tid >> 4;
  T acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0;
  for (int k0 = 0; k0 < p; k0 += 16) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u;
      const int rl = e & 63, kl = e >> 6;
      const int64_t r = r0 + rl;
      const int k = k0 + kl;
      As[kl][rl] = (r < n && k < p) ? X[(size_t)k * n + r] : T(0);
      const int kl2 = e & 15, ml = e >> 4;
      const int k2 = k0 + kl2, m = m0 + ml;
      Bs[kl2][ml] = (k2 < p && m < M) ? Z[(size_t)m * d + k2] : T(0);
    }
    __syncthreads();
#pragma unroll
    for (int kl = 0; kl < 16; ++kl) {
      T av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = As[kl][tr + 16 * i];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = Bs[kl][tc + 16 * j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
    }
    __syncthreads();
  }
  // per row: this thread's four samples, then the 16 threads (tc) that share the row through LDS
  __shared__ double part[16][64];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + tc + 16 * j;
      if (m < M) {
        const T pi = logistic_t<T>(acc[i][j]);
        s += (double)(pi * (T(1) - pi));
      }
    }
    part[tc][tr + 16 * i] = s;
  }
  __syncthreads();
  if (tid < 64 && r0 + tid < n) {
    double s = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += part[q][tid];
    atomicAdd(wsum + r0 + tid, s);
  }
}

// Hacc[j + k p] += sum over the rows of this split of X_rj wsum_r X_rk: 32 x 32 tile per workgroup, rows in chunks of 64 through LDS
template <typename T>
__global__ __launch_bounds__(256) void k_h2_lr_gram(int p, int64_t n, int64_t rows_per_split, const T *__restrict__ X, const double *__restrict__ wsum, double *__restrict__ Hacc) {
  __shared__ T Aj[64][33];
  __shared__ T Bk[64][33];
  const int tid = threadIdx.x, j0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  if (k0 > j0) return;   // lower triangle of tiles; the finisher mirrors
  const int64_t lo = (int64_t)blockIdx.z * rows_per_split, hi = lo + rows_per_split < n ? lo + rows_per_split : n;
  const int tj = tid & 31, tk = tid >> 5;   // outputs (tj, tk + 8 q), q < 4
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t r0 = lo; r0 < hi; r0 += 64) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = tid + 256 * u;   // 0 .. 2047
      const int rl = e & 63, cl = e >> 6;
      const int64_t r = r0 + rl;
      const bool in = r < hi;
      Aj[rl][cl] = (in && j0 + cl < p) ? X[(size_t)(j0 + cl) * n + r] : T(0);
      Bk[rl][cl] = (in && k0 + cl < p) ? (T)((double)X[(size_t)(k0 + cl) * n + r] * wsum[r]) : T(0);
    }
    __syncthreads();
    T part[4] = {0, 0, 0, 0};
#pragma unroll 8
    for (int rl = 0; rl < 64; ++rl) {
      const T a = Aj[rl][tj];
#pragma unroll
      for (int q = 0; q < 4; ++q) part[q] += a * Bk[rl][tk + 8 * q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] += (double)part[q];
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (j0 + tj < p && k0 + tk + 8 * q < p) atomicAdd(Hacc + (size_t)(j0 + tj) + (size_t)(k0 + tk + 8 * q) * p, acc[q]);
}

// Per-sample statistics of both targets, one WAVE per sample (row 0 / row d - 1 carries the scale coordinate):
//   kind 0 logistic regression (s = Z[p, m]): stats[0] += sigma^-2, stats[1] += -2 beta'beta sigma^-2 + hyper'', rows[j] += 2 beta_j sigma^-2
//   kind 1 funnel, unconstrained (v = Z[0, m]): stats[0] += e^{-2v}, stats[1] += -1 / sv^2 - 2 e^{-2v} |x|^2, rows[i] += 2 e^{-2v} x_i (i >= 1)
//   kind 2 funnel, constrained (s = Z[0, m]): stats[0] += s^-2, stats[1] += d / s^2 - (1 - log s) / (sv^2 s^2) - 3 |x|^2 / s^4, rows[i] += 2 x_i / s^3
template <typename T>
__global__ __launch_bounds__(256) void k_h2_stats(int kind, int d, int M, const T *__restrict__ Z, int variant, double sv, double *__restrict__ stats, double *__restrict__ rows) {
  const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const T *z = Z + (size_t)m * d;
  const int sc = kind == 0 ? d - 1 : 0, lo = kind == 0 ? 0 : 1, hi = kind == 0 ? d - 1 : d;
  const double t = (double)z[sc];
  double f, sq = 0.0;   // f: the factor of x_i in the cross term
  if (kind == 0) f = 2.0 * exp(-2.0 * t);
  else if (kind == 1) f = 2.0 * exp(-2.0 * t);
  else f = 2.0 / (t * t * t);
  for (int i = lo + lane; i < hi; i += 64) {
    const double x = (double)z[i];
    sq += x * x;
    atomicAdd(rows + i, f * x);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  if (lane == 0) {
    double a, b;
    if (kind == 0) {
      const double is2 = exp(-2.0 * t);
      a = is2;
      b = -2.0 * sq * is2 + (variant == 0 ? -2.0 * exp(2.0 * t) / 9.0 : -1.0 / 9.0);
    } else if (kind == 1) {
      const double e2 = exp(-2.0 * t);
      a = e2;
      b = -1.0 / (sv * sv) - 2.0 * e2 * sq;
    } else {
      const double s2 = t * t;
      a = 1.0 / s2;
      b = (double)d / s2 - (1.0 - log(t)) / (sv * sv * s2) - 3.0 * sq / (s2 * s2);
    }
    atomicAdd(stats, a);
    atomicAdd(stats + 1, b);
  }
}

// hess (d x d, column-major, T) from the accumulated sums
template <typename T>
__global__ void k_h2_finish(int kind, int d, double inv_n, double likeadj, const double *__restrict__ Hacc, const double *__restrict__ stats,
                            const double *__restrict__ rows, T *__restrict__ hess) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)d * d) return;
  const int i = (int)(e % d), j = (int)(e / d);
  const int sc = kind == 0 ? d - 1 : 0;
  double v;
  if (i == sc && j == sc) v = stats[1] * inv_n;
  else if (i == sc) v = rows[j] * inv_n;
  else if (j == sc) v = rows[i] * inv_n;
  else if (kind == 0) {
    const int p = d - 1, a = i > j ? i : j, b = i > j ? j : i;   // (the Gram kernel fills the lower triangle of tiles: row index >= column index by tile)
    const double g = (a / 32 == b / 32) ? Hacc[(size_t)i + (size_t)j * p] : Hacc[(size_t)a + (size_t)b * p];
    v = -likeadj * g * inv_n - (i == j ? stats[0] * inv_n : 0.0);
  } else {
    v = i == j ? -stats[0] * inv_n : 0.0;
  }
  hess[e] = (T)v;
}

}  // namespace

bool target_has_hess2(const mivi_ctx *c) { return (c->target == TGT_LOGREG || c->target == TGT_FUNNEL) && !c->bij_on; }

// scratch (c->h2_acc, allocated by the caller): [stats (8) | rows (d) | wsum (n) | Hacc (p p)] doubles, zeroed before the first chunk
size_t target_hess2_bytes(const mivi_ctx *c) {
  const int d = c->cfg.d;
  const size_t n = c->target == TGT_LOGREG ? (size_t)c->lr_n : 0, pp = c->target == TGT_LOGREG ? (size_t)(d - 1) * (d - 1) : 0;
  return (8 + (size_t)d + n + pp) * sizeof(double);
}
bool target_hess2_begin(mivi_ctx *c) { return hipMemsetAsync(c->h2_acc.p, 0, target_hess2_bytes(c), c->stream) == hipSuccess; }

// one chunk of Mc samples whose Z (d x Mc, column-major) sits in c->Z
void target_hess2_accumulate(mivi_ctx *c, int Mc) {
  const int d = c->cfg.d;
  double *stats = (double *)c->h2_acc.p, *rows = stats + 8, *wsum = rows + d;
  const bool f32 = c->cfg.dtype == MIVI_F32;
  const int kind = c->target == TGT_LOGREG ? 0 : (c->funnel_constrained ? 2 : 1);
  if (f32) hipLaunchKernelGGL(k_h2_stats<float>, dim3((Mc + 3) / 4), dim3(256), 0, c->stream, kind, d, Mc, (const float *)c->Z.p, c->lr_variant, c->funnel_sigma_v, stats, rows);
  else hipLaunchKernelGGL(k_h2_stats<double>, dim3((Mc + 3) / 4), dim3(256), 0, c->stream, kind, d, Mc, (const double *)c->Z.p, c->lr_variant, c->funnel_sigma_v, stats, rows);
  if (kind == 0) {
    const int64_t n = c->lr_n;
    const dim3 grid((unsigned)((n + 63) / 64), (unsigned)((Mc + 63) / 64));
    if (f32) hipLaunchKernelGGL(k_h2_lr_wsum<float>, grid, dim3(256), 0, c->stream, d, d - 1, Mc, n, (const float *)c->lr_X, (const float *)c->Z.p, wsum);
    else hipLaunchKernelGGL(k_h2_lr_wsum<double>, grid, dim3(256), 0, c->stream, d, d - 1, Mc, n, (const double *)c->lr_X, (const double *)c->Z.p, wsum);
  }
}

void target_hess2_finish(mivi_ctx *c, int n_samples, void *hess) {
  const int d = c->cfg.d;
  double *stats = (double *)c->h2_acc.p, *rows = stats + 8, *wsum = rows + d;
  const bool f32 = c->cfg.dtype == MIVI_F32;
  const int kind = c->target == TGT_LOGREG ? 0 : (c->funnel_constrained ? 2 : 1);
  double *Hacc = nullptr;
  if (kind == 0) {
    const int p = d - 1;
    const int64_t n = c->lr_n;
    Hacc = wsum + n;
    const int nt = (p + 31) / 32;
    int S = (int)((n + 8191) / 8192);   // row splits: enough workgroups for a small p, bounded atomics
    const int cap = (4096 + nt * nt - 1) / (nt * nt);
    if (S > cap) S = cap;
    if (S < 1) S = 1;
    const int64_t rps = ((n + S - 1) / S + 63) / 64 * 64;
    S = (int)((n + rps - 1) / rps);
    const dim3 grid(nt, nt, S);
    if (f32) hipLaunchKernelGGL(k_h2_lr_gram<float>, grid, dim3(256), 0, c->stream, p, n, rps, (const float *)c->lr_X, wsum, Hacc);
    else hipLaunchKernelGGL(k_h2_lr_gram<double>, grid, dim3(256), 0, c->stream, p, n, rps, (const double *)c->lr_X, wsum, Hacc);
  }
  const size_t nn = (size_t)d * d;
  const dim3 g2((unsigned)((nn + 255) / 256));
  if (f32) hipLaunchKernelGGL(k_h2_finish<float>, g2, dim3(256), 0, c->stream, kind, d, 1.0 / (double)n_samples, c->lr_likeadj, Hacc, stats, rows, (float *)hess);
  else hipLaunchKernelGGL(k_h2_finish<double>, g2, dim3(256), 0, c->stream, kind, d, 1.0 / (double)n_samples, c->lr_likeadj, Hacc, stats, rows, (double *)hess);
}

}  // namespace mivi
