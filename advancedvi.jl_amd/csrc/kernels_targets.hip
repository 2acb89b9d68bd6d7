// Built-in target log-densities evaluated as standalone kernels: Z (d x M) -> ell (M), G (d x M).
// These sit where the reference calls the LogDensityProblems plugin once per column
// (`mean(logdensity(prob, z_m))`, src/algorithms/repgradelbo.jl:84-86; gradient through the rrule
// seam src/mixedad_logdensity.jl:23-34).
//
//   diag Gaussian  MvNormal(mean, Diagonal(std^2))          test/models/normal.jl:56-75
//   funnel         Neal's funnel + Stacked([log, identity])  SURVEY.md 8d (README.md:76-82,102-106 wrapper)
//   logreg         hierarchical logistic regression           docs/src/tutorials/subsampling.md:26-38 (variant 0)
//                                                              README.md:42-66,91-106                (variant 1)
#include <cstdlib>

#include "device_common.h"
#include "fr_planes.h"

namespace mivi {

// one workgroup per sample column; rows strided over threads (coalesced)
template <typename T>
__global__ __launch_bounds__(256) void k_col_target(ColTargetArgs<T> a) {
  __shared__ double red[4];
  const int m = blockIdx.x, tid = threadIdx.x, d = a.d;
  const T *z = a.Z + (size_t)m * d;
  T *g = a.G + (size_t)m * d;
  if (a.kind == TGT_DIAG_GAUSS) {
    T acc = 0;
    for (int i = tid; i < d; i += 256) {
      const T u = (z[i] - a.t_mean[i]) * a.t_istd[i];
      acc += T(-0.5) * u * u;
      if (a.want_grad) g[i] = -u * a.t_istd[i];
    }
    const double s = block_sum<double, 256>((double)acc, red);
    if (tid == 0) a.ell[m] = (T)s;
  } else {  // TGT_FUNNEL
    // unconstrained form: z[0] = eta_1 = log s (exp bijector + log-Jacobian built in); constrained form: z[0] = s itself
    const double e1 = a.constrained ? log((double)z[0]) : (double)z[0];
    const double inv_s2 = exp(-2.0 * e1);
    T acc = 0;
    for (int i = 1 + tid; i < d; i += 256) {
      const T x = z[i];
      acc += x * x;
      if (a.want_grad) g[i] = (T)(-(double)x * inv_s2);
    }
    const double sx2 = block_sum<double, 256>((double)acc, red);
    if (tid == 0) {
      const double n = (double)(d - 1), sv2 = a.sigma_v * a.sigma_v;
      // log LogNormal(e^{e1}; 0, sv) + sum_i log N(x_i; 0, e^{e1}) + log|det J| (= e1); constants in ell_const
      if (a.constrained) {   // d/ds of log LogNormal(s) + sum_i log N(x_i; 0, s), no Jacobian term
        const double s_ = (double)z[0];
        a.ell[m] = (T)((-e1 - e1 * e1 / (2.0 * sv2)) + (-n * e1 - 0.5 * sx2 * inv_s2));
        if (a.want_grad) g[0] = (T)(((-1.0 - e1 / sv2) + (-n + sx2 * inv_s2)) / s_);
      } else {
        a.ell[m] = (T)((-e1 - e1 * e1 / (2.0 * sv2)) + (-n * e1 - 0.5 * sx2 * inv_s2) + e1);
        if (a.want_grad) g[0] = (T)((-1.0 - e1 / sv2) + (-n + sx2 * inv_s2) + 1.0);
      }
    }
  }
}

template <typename T>
static void col_target_impl(mivi_ctx *c, int M, int want_grad) {
  ColTargetArgs<T> a;
  a.d = c->cfg.d;
  a.M = M;
  a.kind = c->target;
  a.Z = (const T *)c->Z.p;
  a.G = (T *)c->W.p;
  a.ell = (T *)c->ell.p;
  a.t_mean = (const T *)c->t_mean.p;
  a.t_istd = (const T *)c->t_istd.p;
  a.sigma_v = c->funnel_sigma_v;
  a.want_grad = want_grad;
  a.constrained = c->funnel_constrained;
  hipLaunchKernelGGL(k_col_target<T>, dim3(M), dim3(256), 0, c->stream, a);
}

void launch_col_target(mivi_ctx *c, int M, int want_grad) {
  if (c->cfg.dtype == MIVI_F32) col_target_impl<float>(c, M, want_grad); else col_target_impl<double>(c, M, want_grad);
}

// ---------------------------------------------------------------------------------------------
// Stacked bijector around any target (README.md:76-82,91-119): one workgroup per sample column.
//   forward : x = binv(eta) in place (exp rows), ld[m] = sum over exp rows of eta   (= logabsdetjac(binv, eta))
//   backward: G <- J' G + d logabsdetjac / d eta = x .* G + 1 on exp rows;  ell[m] += ld[m]
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_bij_fwd(int d, const uint8_t *mask, T *Z, T *ld) {
  __shared__ double red[4];
  const int m = blockIdx.x, tid = threadIdx.x;
  T *z = Z + (size_t)m * d;
  double acc = 0.0;
  for (int i = tid; i < d; i += 256)
    if (mask[i]) {
      const T eta = z[i];
      acc += (double)eta;
      z[i] = exp(eta);
    }
  const double s = block_sum<double, 256>(acc, red);
  if (tid == 0) ld[m] = (T)s;
}
template <typename T>
__global__ __launch_bounds__(256) void k_bij_bwd(int d, const uint8_t *mask, const T *X, T *G, T *ell, const T *ld, int want_grad) {
  const int m = blockIdx.x, tid = threadIdx.x;
  if (want_grad) {
    const T *x = X + (size_t)m * d;
    T *g = G + (size_t)m * d;
    for (int i = tid; i < d; i += 256)
      if (mask[i]) g[i] = fma(x[i], g[i], T(1));
  }
  if (ell && tid == 0) ell[m] += ld[m];
}
void launch_bij_forward(mivi_ctx *c, int M) {
  if (c->cfg.dtype == MIVI_F32)
    hipLaunchKernelGGL(k_bij_fwd<float>, dim3(M), dim3(256), 0, c->stream, c->cfg.d, (const uint8_t *)c->bij_mask.p, (float *)c->Z.p, (float *)c->bij_ld.p);
  else
    hipLaunchKernelGGL(k_bij_fwd<double>, dim3(M), dim3(256), 0, c->stream, c->cfg.d, (const uint8_t *)c->bij_mask.p, (double *)c->Z.p, (double *)c->bij_ld.p);
}
void launch_bij_backward(mivi_ctx *c, int M, int want_grad, bool add_to_ell) {
  if (c->cfg.dtype == MIVI_F32)
    hipLaunchKernelGGL(k_bij_bwd<float>, dim3(M), dim3(256), 0, c->stream, c->cfg.d, (const uint8_t *)c->bij_mask.p, (const float *)c->Z.p,
                       (float *)c->W.p, add_to_ell ? (float *)c->ell.p : nullptr, (const float *)c->bij_ld.p, want_grad);
  else
    hipLaunchKernelGGL(k_bij_bwd<double>, dim3(M), dim3(256), 0, c->stream, c->cfg.d, (const uint8_t *)c->bij_mask.p, (const double *)c->Z.p,
                       (double *)c->W.p, add_to_ell ? (double *)c->ell.p : nullptr, (const double *)c->bij_ld.p, want_grad);
}

// ---------------------------------------------------------------------------------------------
// Hierarchical logistic regression, generic route (any T): three kernels
//   L1: logits = X B (n x p by p x M), resid = y - sigmoid(logit), per-(rowblock, m) loglik partials
//   L2: split-K  X^T resid  partials
//   L3: per column: reduce, add priors, write ell_m and G[:, m]
// X is n x p COLUMN-major (X[r + k*n], Julia's native Matrix layout), theta = [beta (p); s].
// ---------------------------------------------------------------------------------------------
template <typename T>
struct LrArgs {
  int d, p, M;
  int64_t n;
  const T *X;
  const uint8_t *y;
  const T *Z;      // d x M
  T *R;            // n x M resid (col-major)
  double *ll_part; // [nrb][M]
  T *g_part;       // [S][p*M]
  int S, nrb;
  int64_t rows_per_split;
  T *G;
  T *ell;
  int variant;
  double likeadj;
  int want_grad;
};

template <typename T>
__device__ __forceinline__ T softplus_t(T x) {
  // log(1 + e^x), stable
  return x > T(0) ? x + log1p(exp(-x)) : log1p(exp(x));
}

template <typename T>
__global__ __launch_bounds__(256) void k_lr_logits(LrArgs<T> a) {
  __shared__ T As[16][65];  // As[k][r]
  __shared__ T Bs[16][65];  // Bs[k][m]
  __shared__ double colsum[4][64];
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  const int m0 = blockIdx.y * 64;
  const int tr = tid & 15, tc = tid >> 4;  // thread computes rows tr + 16*i, cols tc + 16*j
  T acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0;
  for (int k0 = 0; k0 < a.p; k0 += 16) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u;     // 0..1023
      const int rl = e & 63, kl = e >> 6;
      const int64_t r = r0 + rl;
      const int k = k0 + kl;
      As[kl][rl] = (r < a.n && k < a.p) ? a.X[(size_t)k * a.n + r] : T(0);
      const int kl2 = e & 15, ml = e >> 4;
      const int k2 = k0 + kl2, m = m0 + ml;
      Bs[kl2][ml] = (k2 < a.p && m < a.M) ? a.Z[(size_t)m * a.d + k2] : T(0);
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
  // epilogue: residuals + log-likelihood column partials
  double ll[4] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t r = r0 + tr + 16 * i;
    if (r < a.n) {
      const T yv = (T)a.y[r];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = m0 + tc + 16 * j;
        if (m < a.M) {
          const T lg = acc[i][j];
          ll[j] += (double)(yv * lg - softplus_t(lg));
          if (a.want_grad) a.R[(size_t)m * a.n + r] = yv - T(1) / (T(1) + exp(-lg));
        }
      }
    }
  }
  // reduce over the 16 threads (tr) sharing a column set: lanes tid&15 are adjacent
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double v = ll[j];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
    if (tr == 0) colsum[j][tc] = v;
  }
  __syncthreads();
  if (tid < 64) {
    const int j = tid >> 4, tcc = tid & 15;
    const int m = m0 + tcc + 16 * j;
    if (m < a.M) a.ll_part[(size_t)blockIdx.x * a.M + m] = colsum[j][tcc];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_lr_xtr(LrArgs<T> a) {
  __shared__ T As[16][65];  // As[r][k]
  __shared__ T Bs[16][65];  // Bs[r][m]
  const int tid = threadIdx.x;
  const int k0 = blockIdx.x * 64, m0 = blockIdx.y * 64, s = blockIdx.z;
  const int64_t rbeg = (int64_t)s * a.rows_per_split;
  const int64_t rend = min(a.n, rbeg + a.rows_per_split);
  const int tr = tid & 15, tc = tid >> 4;
  T acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0;
  for (int64_t r0 = rbeg; r0 < rend; r0 += 16) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = tid + 256 * u;
      const int rl = e & 15, cl = e >> 4;   // 16 consecutive rows per column: 64-byte segments
      const int64_t r = r0 + rl;
      const int k = k0 + cl, m = m0 + cl;
      As[rl][cl] = (r < rend && k < a.p) ? a.X[(size_t)k * a.n + r] : T(0);
      Bs[rl][cl] = (r < rend && m < a.M) ? a.R[(size_t)m * a.n + r] : T(0);
    }
    __syncthreads();
#pragma unroll
    for (int rl = 0; rl < 16; ++rl) {
      T av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = As[rl][tr + 16 * i];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = Bs[rl][tc + 16 * j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + tr + 16 * i, m = m0 + tc + 16 * j;
      if (k < a.p && m < a.M) a.g_part[((size_t)s * a.M + m) * a.p + k] = acc[i][j];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_lr_finish(LrArgs<T> a) {
  __shared__ double red[4];
  const int m = blockIdx.x, tid = threadIdx.x;
  const int p = a.p, d = a.d;
  const T *z = a.Z + (size_t)m * d;
  const double sv = (double)z[p];
  const double sigma = exp(sv), inv_s2 = exp(-2.0 * sv);
  double ll = 0.0;
  for (int b = tid; b < a.nrb; b += 8 * 256) {   // eight loads in flight (a thread's entries are M doubles apart: one line each), same summation order
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = b + 256 * u < a.nrb ? a.ll_part[(size_t)(b + 256 * u) * a.M + m] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) ll += v[u];
  }
  ll = block_sum<double, 256>(ll, red);
  double bb = 0.0;
  for (int k = tid; k < p; k += 256) {
    const double bk = (double)z[k];
    bb += bk * bk;
    if (a.want_grad) {
      double g = 0.0;
      int s = 0;
      for (; s + 8 <= a.S; s += 8) {   // eight loads in flight, same summation order
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = a.g_part[((size_t)(s + u) * a.M + m) * p + k];
#pragma unroll
        for (int u = 0; u < 8; ++u) g += (double)v[u];
      }
      for (; s < a.S; ++s) g += (double)a.g_part[((size_t)s * a.M + m) * p + k];
      a.G[(size_t)m * d + k] = (T)(a.likeadj * g - bk * inv_s2);
    }
  }
  bb = block_sum<double, 256>(bb, red);
  if (tid == 0) {
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
    a.ell[m] = (T)(a.likeadj * ll + logprior_beta + logprior_sigma + jac);
    if (a.want_grad) a.G[(size_t)m * d + p] = (T)gs;
  }
}

// ---------------------------------------------------------------------------------------------
// f32 MFMA route for the two LogReg contractions (v_mfma_f32_32x32x2_f32, 64x64 register blocking per wave,
// operands straight from L2/HBM in operand layout, double-buffered 16-k stages):
//   logits  L[r, m] = sum_k X[r + k n] * ZT[m + k ldz]          -> resid stored sample-contiguous R[m + r ldr]
//   X^T r   G[k, m] = sum_r Xrm[k + r ldx] * R[m + r ldr]        (row range split over gridDim.x, partials)
// Xrm is a row-major, 32-padded copy of X built once by mivi_set_target_logreg; ZT is the transposed sample
// matrix the sampling kernel (or k_rt_from_z) leaves in c->RT.  Every load and store is a contiguous 128-byte
// segment per half-wave.
// ---------------------------------------------------------------------------------------------
typedef float lr_f32x16 __attribute__((ext_vector_type(16)));

struct LrMfmaArgs {
  int d, p, M;
  long long n;
  const float *X;      // n x p column-major
  const float *Xrm;    // n x ldx row-major (zero padded columns)
  int ldx;
  const uint8_t *y;
  const float *ZT;     // ZT[m + k*ldz]
  int ldz;
  const float *Zcm;    // the sample matrix itself, d x M column-major (k contiguous per sample): the split-operand logits
  const unsigned *xmax;   // bits of max |X| (k_lr_make_xrm): the power-of-two scale of X's f16 splits
  const unsigned *XA;     // X as A-operand planes (k_lr_xplanes): fragment (rb32, kg) at (rb32 (ldx / 16) + kg) kFrag
  unsigned *ZP;           // the samples as B-operand planes (k_lr_zplanes): fragment (mb32, kg)
  const unsigned *XB;     // X as B-operand planes of X^T R (k_lr_xbplanes): fragment (fb32, rg = 16-row group) at (fb32 nrg + rg) kFrag
  unsigned *RP;           // the residuals as A-operand planes (2^13 r): fragment (mb32, rg) at (mb32 nrg + rg) kFrag; nullptr: R in f32
  long long nrg;          // 16-row groups (rows padded to whole 128-row tiles)
  int gps;                // row groups per split of k_lr_xtr_planes
  float *R;            // R[m + r*ldr]
  int ldr;
  double *ll_part;     // [gridDim.x][M]
  float *g_part;       // [S][p*M]
  long long rows_per_split;
  int want_grad;
};

__device__ __forceinline__ float lr_softplus(float x) { return x > 0.f ? x + log1pf(expf(-x)) : log1pf(expf(x)); }

typedef float lr_f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// f32-MFMA kernels with operands staged through LDS.  What was measured on their register-operand first generation (MFMA pipe busy
// 48 % / 63 %, PMC; removed in round 3) and on register-pipelined variants of it:
//   * a branch around the prefetch loads makes the compiler merge the s_waitcnt of both paths to the conservative one
//     (it then waits for the loads it has just issued) -> prefetch must be unconditional on a clamped stage index;
//   * without a scheduling barrier the machine scheduler sinks every prefetch load down to its first use
//     (load -> s_waitcnt vmcnt(0) -> mfma), silently removing the software pipeline;
//   * a wave can have at most 63 loads outstanding (vmcnt is 6 bits): dword operand loads cap the prefetch depth;
//   * with the pipeline pinned, the deeper the per-wave prefetch the SLOWER X^T R ran (1.5 -> 2.3-2.7 ms): the waves of
//     a workgroup share operand rows only through the 32 KB L1, and with several stages in flight per wave the shared
//     lines are evicted before the sibling waves ask for them.
// Staging makes the sharing explicit: every operand byte is fetched once per workgroup with 16-byte coalesced loads
// (3 per thread per stage instead of 32 dword loads per wave), a two-slot LDS ring feeds the MFMAs through ds_read,
// global loads run two stages ahead in registers, one barrier per stage.  X^T R 1.52 -> 1.13 ms (74 % of the f32 MFMA
// peak), logits 1.90 -> 1.33 ms (63 %) at n = 1e6, p = 511, M = 128.
//   stage = 16 data rows: Rs[16][128] samples, Xs[16][256] features  (24 KB per slot)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_lr_xtr_mfma_lds(LrMfmaArgs a) {
  __shared__ __attribute__((aligned(16))) float Rs[2][16 * 128];
  __shared__ __attribute__((aligned(16))) float Xs[2][16 * 256];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wk = w & 3;
  const int mbase = blockIdx.z * 128, kbase = blockIdx.y * 256;
  const int m0 = mbase + wm * 64, k0 = kbase + wk * 64;
  const long long rbeg = (long long)blockIdx.x * a.rows_per_split;
  const long long rend = min(a.n, rbeg + a.rows_per_split);
  const int ldr = a.ldr, ldx = a.ldx;
  const int nst = (int)((rend - rbeg + 15) / 16);
  // cooperative loads: R one float4 per thread (row t>>5, 4 samples), X two (rows t>>6 and 8 + t>>6, 4 features)
  const int rrow = tid >> 5, rcol = min(mbase + 4 * (tid & 31), ldr - 4);
  const int xrow = tid >> 6, xcol = min(kbase + 4 * (tid & 63), ldx - 4);
  struct G { lr_f32x4 r, x0, x1; };
  auto gload = [&](int st, G &g) {
    st = min(st, nst - 1);
    const long long rb = rbeg + 16LL * st;
    const long long r_r = rb + rrow, r_x0 = rb + xrow, r_x1 = rb + 8 + xrow;
    const bool okr = r_r < rend;
    g.r = *(const lr_f32x4 *)(a.R + (size_t)(okr ? r_r : rend - 1) * ldr + rcol);
    g.x0 = *(const lr_f32x4 *)(a.Xrm + (size_t)min(r_x0, rend - 1) * ldx + xcol);
    g.x1 = *(const lr_f32x4 *)(a.Xrm + (size_t)min(r_x1, rend - 1) * ldx + xcol);
    if (!okr) g.r = lr_f32x4{0.f, 0.f, 0.f, 0.f};   // rows past the split contribute nothing (X stays finite)
  };
  auto lstore = [&](int slot, const G &g) {
    *(lr_f32x4 *)&Rs[slot][rrow * 128 + 4 * (tid & 31)] = g.r;
    *(lr_f32x4 *)&Xs[slot][xrow * 256 + 4 * (tid & 63)] = g.x0;
    *(lr_f32x4 *)&Xs[slot][(8 + xrow) * 256 + 4 * (tid & 63)] = g.x1;
  };
  lr_f32x16 c00, c01, c10, c11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { c00[r] = 0.f; c01[r] = 0.f; c10[r] = 0.f; c11[r] = 0.f; }
  const int ao = h * 128 + wm * 64 + l31, bo = h * 256 + wk * 64 + l31;
  auto compute = [&](int slot) {
    const float *rs = &Rs[slot][ao], *xs = &Xs[slot][bo];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float a0 = rs[2 * u * 128], a1 = rs[2 * u * 128 + 32];
      const float b0 = xs[2 * u * 256], b1 = xs[2 * u * 256 + 32];
      c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c00, 0, 0, 0);
      c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c01, 0, 0, 0);
      c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c10, 0, 0, 0);
      c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c11, 0, 0, 0);
    }
  };
  if (nst > 0) {
    G ga, gb;
    gload(0, ga);
    lstore(0, ga);
    gload(1, ga);
    gload(2, gb);
    lds_barrier();   // LDS-only: __syncthreads() would also drain the prefetch loads (vmcnt)
    int st = 0;
    while (true) {
      lstore((st + 1) & 1, ga);      // stage st+1 (loaded two iterations ago)
      gload(st + 3, ga);
      compute(st & 1);
      lds_barrier();   // LDS-only: __syncthreads() would also drain the prefetch loads (vmcnt)
      if (++st >= nst) break;
      lstore((st + 1) & 1, gb);
      gload(st + 3, gb);
      compute(st & 1);
      lds_barrier();   // LDS-only: __syncthreads() would also drain the prefetch loads (vmcnt)
      if (++st >= nst) break;
    }
  }
  auto epi = [&](const lr_f32x16 &c, int mb, int kb) {
    const int k = k0 + kb * 32 + l31;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = m0 + mb * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
      if (k < a.p && m < a.M) a.g_part[((size_t)blockIdx.x * a.M + m) * a.p + k] = c[q];
    }
  };
  epi(c00, 0, 0);
  epi(c01, 0, 1);
  epi(c10, 1, 0);
  epi(c11, 1, 1);
}

// logits through LDS: 256 rows x 128 samples per workgroup (8 waves as 4 x 2, each 64 x 64), stage = 16 k:
//   Xs[256][20]  (row-major slab of Xrm, row stride 20 words: 16-byte aligned and conflict-free for the b128 operand reads)
//   Zs[16][128]  (ZT rows)
// Lane half h feeds k = 8g + 4h + i to MFMA i of group g on both operands (the order of k inside a dot product is free),
// so the A operand is one ds_read_b128 per 4 MFMAs.
// PART: some 32-sample tiles of the 128-sample workgroup tile lie entirely beyond M (small n_samples): their MFMAs are skipped
// (a separate instantiation, so the full-tile kernel keeps its branch-free stage body).
template <bool PART>
__global__ __launch_bounds__(512) void k_lr_logits_mfma_lds(LrMfmaArgs a) {
  __shared__ __attribute__((aligned(16))) float Xs[2][256 * 20];
  __shared__ __attribute__((aligned(16))) float Zs[2][16 * 128];
  __shared__ float ll_lds[128];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wm = w & 1;
  const long long rb0 = (long long)blockIdx.x * 256;
  const long long r0 = rb0 + wr * 64;
  const int mbase = blockIdx.y * 128, m0 = mbase + wm * 64;
  const int ldx = a.ldx, ldz = a.ldz;
  const int nst = ldx / 16;
  if (tid < 128) ll_lds[tid] = 0.f;
  const int xrow = tid >> 2, xc = 4 * (tid & 3);
  const float *Xg0 = a.Xrm + (size_t)min(rb0 + xrow, a.n - 1) * ldx + xc;
  const float *Xg1 = a.Xrm + (size_t)min(rb0 + 128 + xrow, a.n - 1) * ldx + xc;
  const int zk = tid >> 5, zc = min(mbase + 4 * (tid & 31), ldz - 4);
  struct G { lr_f32x4 x0, x1, z; };
  auto gload = [&](int st, G &g) {
    st = min(st, nst - 1);
    g.x0 = *(const lr_f32x4 *)(Xg0 + 16 * st);
    g.x1 = *(const lr_f32x4 *)(Xg1 + 16 * st);
    g.z = *(const lr_f32x4 *)(a.ZT + (size_t)(16 * st + zk) * ldz + zc);
  };
  auto lstore = [&](int slot, const G &g) {
    *(lr_f32x4 *)&Xs[slot][xrow * 20 + xc] = g.x0;
    *(lr_f32x4 *)&Xs[slot][(128 + xrow) * 20 + xc] = g.x1;
    *(lr_f32x4 *)&Zs[slot][zk * 128 + 4 * (tid & 31)] = g.z;
  };
  lr_f32x16 c00, c01, c10, c11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { c00[r] = 0.f; c01[r] = 0.f; c10[r] = 0.f; c11[r] = 0.f; }
  const int ao = (wr * 64 + l31) * 20 + 4 * h, bo = 4 * h * 128 + wm * 64 + l31;
  const bool has0 = m0 < a.M, has1 = m0 + 32 < a.M;   // wave-uniform
  auto compute = [&](int slot) {
    const float *xs = &Xs[slot][ao], *zs = &Zs[slot][bo];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const lr_f32x4 a0 = *(const lr_f32x4 *)(xs + 8 * g), a1 = *(const lr_f32x4 *)(xs + 32 * 20 + 8 * g);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float b0 = zs[(8 * g + i) * 128], b1 = zs[(8 * g + i) * 128 + 32];
        if (!PART || has0) {
          c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0, c00, 0, 0, 0);
          c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b0, c10, 0, 0, 0);
        }
        if (!PART || has1) {
          c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b1, c01, 0, 0, 0);
          c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1, c11, 0, 0, 0);
        }
      }
    }
  };
  {
    G ga, gb;
    gload(0, ga);
    lstore(0, ga);
    gload(1, ga);
    gload(2, gb);
    lds_barrier();   // LDS-only: __syncthreads() would also drain the prefetch loads (vmcnt)
    int st = 0;
    while (true) {
      lstore((st + 1) & 1, ga);
      gload(st + 3, ga);
      compute(st & 1);
      lds_barrier();   // LDS-only: __syncthreads() would also drain the prefetch loads (vmcnt)
      if (++st >= nst) break;
      lstore((st + 1) & 1, gb);
      gload(st + 3, gb);
      compute(st & 1);
      lds_barrier();   // LDS-only: __syncthreads() would also drain the prefetch loads (vmcnt)
      if (++st >= nst) break;
    }
  }
  float ll0 = 0.f, ll1 = 0.f;
  auto epi = [&](const lr_f32x16 &ca, const lr_f32x16 &cb, int rb) {
    const int ma = m0 + l31, mb = m0 + 32 + l31;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const long long r = r0 + rb * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
      if (r < a.n) {
        const float yv = (float)a.y[r];
        {
          const float lg = ca[q], e = __expf(-fabsf(lg)), inv = __frcp_rn(1.f + e);
          if (ma < a.M) {
            ll0 += yv * lg - (fmaxf(lg, 0.f) + __logf(1.f + e));
            if (a.want_grad) a.R[(size_t)r * a.ldr + ma] = yv - (lg >= 0.f ? inv : e * inv);
          }
        }
        {
          const float lg = cb[q], e = __expf(-fabsf(lg)), inv = __frcp_rn(1.f + e);
          if (mb < a.M) {
            ll1 += yv * lg - (fmaxf(lg, 0.f) + __logf(1.f + e));
            if (a.want_grad) a.R[(size_t)r * a.ldr + mb] = yv - (lg >= 0.f ? inv : e * inv);
          }
        }
      }
    }
  };
  epi(c00, c01, 0);
  epi(c10, c11, 1);
  ll0 += __shfl_xor(ll0, 32, 64);
  ll1 += __shfl_xor(ll1, 32, 64);
  if (h == 0) {
    atomicAdd(&ll_lds[wm * 64 + l31], ll0);
    atomicAdd(&ll_lds[wm * 64 + 32 + l31], ll1);
  }
  __syncthreads();
  if (tid < 128) {
    const int m = mbase + tid;
    if (m < a.M) a.ll_part[(size_t)blockIdx.x * a.M + m] = (double)ll_lds[tid];
  }
}

// ---------------------------------------------------------------------------------------------
// logits on the 16-bit matrix cores with f32 accuracy (round 5: two-way f16 split; round 3-4: three bf16 pieces, six products): every f32
// operand is split into hi = f16(s x), lo = f16(s x - hi) when it is staged into LDS (s: a power of two that puts X's largest magnitude
// into [2^13, 2^14) so that its lo parts stay normal; samples and residuals are O(1) and take s = 1 -- f16 subnormals are honoured by the
// pipe), and a product block is the three MFMAs lo*hi, hi*lo, hi*hi (the dropped lo*lo term is below 2^-22 of the product) accumulated in
// f32 by v_mfma_f32_32x32x16_f16.  Half the matrix-pipe work and less than half of the split arithmetic of the bf16 scheme; X is still
// read from HBM once per contraction, as f32.  Same tile shape, ring and epilogue as
// k_lr_logits_mfma_lds; the B operand comes from the column-major sample matrix (k contiguous per sample), which is
// what a 16-byte operand of 8 consecutive k needs (requires d % 4 == 0).
// ---------------------------------------------------------------------------------------------
typedef _Float16 lr_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 lr_f16x4 __attribute__((ext_vector_type(4)));

// two-way f16 split of s * v (s a power of two: exact): hi = f16(s v), lo = f16(s v - hi)
__device__ __forceinline__ void lr_split2(const lr_f32x4 &v, float s, lr_f16x4 &hi, lr_f16x4 &lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x = v[i] * s;
    const _Float16 h = (_Float16)x;
    hi[i] = h;
    lo[i] = (_Float16)(x - (float)h);
  }
}
// the power-of-two scale of X (largest magnitude of the data matrix into [2^13, 2^14): its f16 lo parts stay normal) and its inverse
__device__ __forceinline__ void lr_xscale(const unsigned *xmax_bits, float &s, float &inv) {
  unsigned eb = (*xmax_bits >> 23) & 0xffu;
  eb = eb < 25u ? 25u : (eb > 254u ? 254u : eb);
  s = __builtin_bit_cast(float, (267u - eb) << 23);
  inv = __builtin_bit_cast(float, (eb - 13u) << 23);
}

template <bool PART>
__device__ __forceinline__ void lr_logits_f16x2_body(const LrMfmaArgs &a) {
  __shared__ __attribute__((aligned(16))) _Float16 Xs[2][2][256 * 16];   // [slot][piece][row][16 k]
  __shared__ __attribute__((aligned(16))) _Float16 Zs[2][2][128 * 16];   // [slot][piece][sample][16 k]
  __shared__ float ll_lds[128];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w >> 1, wm = w & 1;
  const long long rb0 = (long long)blockIdx.x * 256;
  const long long r0 = rb0 + wr * 64;
  const int mbase = blockIdx.y * 128, m0 = mbase + wm * 64;
  const int ldx = a.ldx, d = a.d;
  const int nst = ldx / 16;
  if (tid < 128) ll_lds[tid] = 0.f;
  const int xrow = tid >> 2, xc = 4 * (tid & 3);
  const float *Xg0 = a.Xrm + (size_t)min(rb0 + xrow, a.n - 1) * ldx + xc;
  const float *Xg1 = a.Xrm + (size_t)min(rb0 + 128 + xrow, a.n - 1) * ldx + xc;
  const float *Zg = a.Zcm + (size_t)min(mbase + xrow, a.M - 1) * d;   // sample xrow of this block (128 samples x 4 chunks)
  struct G { lr_f32x4 x0, x1, z; };
  auto gload = [&](int st, G &g) {
    st = min(st, nst - 1);
    g.x0 = *(const lr_f32x4 *)(Xg0 + 16 * st);
    g.x1 = *(const lr_f32x4 *)(Xg1 + 16 * st);
    // k >= p multiplies the zero padding of Xrm: any finite in-bounds value will do there
    g.z = *(const lr_f32x4 *)(Zg + min(16 * st + xc, d - 4));
  };
  float xs, xinv;
  lr_xscale(a.xmax, xs, xinv);
  auto lstore = [&](int slot, const G &g) {
    lr_f16x4 p0, p1;
    lr_split2(g.x0, xs, p0, p1);
    *(lr_f16x4 *)&Xs[slot][0][xrow * 16 + xc] = p0;
    *(lr_f16x4 *)&Xs[slot][1][xrow * 16 + xc] = p1;
    lr_split2(g.x1, xs, p0, p1);
    *(lr_f16x4 *)&Xs[slot][0][(128 + xrow) * 16 + xc] = p0;
    *(lr_f16x4 *)&Xs[slot][1][(128 + xrow) * 16 + xc] = p1;
    lr_split2(g.z, 1.f, p0, p1);   // (samples: O(1) magnitudes, f16 subnormals are honoured by the matrix pipe)
    *(lr_f16x4 *)&Zs[slot][0][xrow * 16 + xc] = p0;
    *(lr_f16x4 *)&Zs[slot][1][xrow * 16 + xc] = p1;
  };
  lr_f32x16 c00, c01, c10, c11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { c00[r] = 0.f; c01[r] = 0.f; c10[r] = 0.f; c11[r] = 0.f; }
  const int ao = (wr * 64 + l31) * 16 + 8 * h, bo = (wm * 64 + l31) * 16 + 8 * h;
  const bool has0 = m0 < a.M, has1 = m0 + 32 < a.M;   // wave-uniform
  auto compute = [&](int slot) {
    lr_f16x8 A0[2], A1[2], B0[2], B1[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      A0[s] = *(const lr_f16x8 *)&Xs[slot][s][ao];
      A1[s] = *(const lr_f16x8 *)&Xs[slot][s][ao + 32 * 16];
      B0[s] = *(const lr_f16x8 *)&Zs[slot][s][bo];
      B1[s] = *(const lr_f16x8 *)&Zs[slot][s][bo + 32 * 16];
    }
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};   // lo.hi, hi.lo, hi.hi: small terms first
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (!PART || has0) {
        c00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0[PA[t]], B0[PB[t]], c00, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1[PA[t]], B0[PB[t]], c10, 0, 0, 0);
      }
      if (!PART || has1) {
        c01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0[PA[t]], B1[PB[t]], c01, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1[PA[t]], B1[PB[t]], c11, 0, 0, 0);
      }
    }
  };
  {
    G ga, gb;
    gload(0, ga);
    lstore(0, ga);
    gload(1, ga);
    gload(2, gb);
    lds_barrier();
    int st = 0;
    while (true) {
      lstore((st + 1) & 1, ga);
      gload(st + 3, ga);
      compute(st & 1);
      lds_barrier();
      if (++st >= nst) break;
      lstore((st + 1) & 1, gb);
      gload(st + 3, gb);
      compute(st & 1);
      lds_barrier();
      if (++st >= nst) break;
    }
  }
  float ll0 = 0.f, ll1 = 0.f;
  auto epi = [&](const lr_f32x16 &ca, const lr_f32x16 &cb, int rb) {
    const int ma = m0 + l31, mb = m0 + 32 + l31;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const long long r = r0 + rb * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
      if (r < a.n) {
        const float yv = (float)a.y[r];
        {
          const float lg = ca[q] * xinv, e = __expf(-fabsf(lg)), inv = __frcp_rn(1.f + e);
          if (ma < a.M) {
            ll0 += yv * lg - (fmaxf(lg, 0.f) + __logf(1.f + e));
            if (a.want_grad) a.R[(size_t)r * a.ldr + ma] = yv - (lg >= 0.f ? inv : e * inv);
          }
        }
        {
          const float lg = cb[q] * xinv, e = __expf(-fabsf(lg)), inv = __frcp_rn(1.f + e);
          if (mb < a.M) {
            ll1 += yv * lg - (fmaxf(lg, 0.f) + __logf(1.f + e));
            if (a.want_grad) a.R[(size_t)r * a.ldr + mb] = yv - (lg >= 0.f ? inv : e * inv);
          }
        }
      }
    }
  };
  epi(c00, c01, 0);
  epi(c10, c11, 1);
  ll0 += __shfl_xor(ll0, 32, 64);
  ll1 += __shfl_xor(ll1, 32, 64);
  if (h == 0) {
    atomicAdd(&ll_lds[wm * 64 + l31], ll0);
    atomicAdd(&ll_lds[wm * 64 + 32 + l31], ll1);
  }
  __syncthreads();
  if (tid < 128) {
    const int m = mbase + tid;
    if (m < a.M) a.ll_part[(size_t)blockIdx.x * a.M + m] = (double)ll_lds[tid];
  }
}

// ---------------------------------------------------------------------------------------------
// logits on OPERAND PLANES (round 5; fr_planes.h): X is constant across estimates, so its two-way f16 split is made ONCE per data set
// (k_lr_xplanes: 4 bytes per element in MFMA-fragment order, the power-of-two scale of max|X| applied) and the samples' once per estimate
// (k_lr_zplanes: 128 x 512 elements).  The contraction is then the batch engine's loop: 128-row x 128-sample tiles, 16-k stages of 16 KiB
// by LDS-DMA into a three-slot ring, ds_read_b128 -> three v_mfma_f32_32x32x16_f16 per fragment pair, no vector arithmetic, three
// workgroups per CU.  The epilogue works in the accumulators' own layout (lane = sample, registers = rows: residual stores are 128-byte
// segments per half-wave), writes R in the f32 layout k_lr_xtr_f16x2 reads, and the per-sample log-likelihood partials.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lr_xplanes(long long n, long long nrb32, int ldx, const float *Xrm, const unsigned *xmax, unsigned *XA) {
  const long long f = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5, ng = ldx >> 4;
  const long long rb = f / ng;
  const int kg = (int)(f % ng);
  if (rb >= nrb32) return;
  float s, inv;
  lr_xscale(xmax, s, inv);
  const long long row = 32 * rb + l31;
  float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (row < n) {
    const float *src = Xrm + (size_t)row * ldx + 16 * kg + 4 * h;
    const lr_f32x4 p = *(const lr_f32x4 *)src, q = *(const lr_f32x4 *)(src + 8);
#pragma unroll
    for (int c = 0; c < 4; ++c) { x[c] = p[c] * s; x[4 + c] = q[c] * s; }
  }
  fb_store_frag(XA + (size_t)f * kFrag + 4 * lane, x);
}
__global__ __launch_bounds__(256) void k_lr_zplanes(int M, int d, int ldx, const float *Zcm, unsigned *ZP) {
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5, ng = ldx >> 4;
  const int mb = f / ng, kg = f % ng;
  if (mb >= (M >> 5)) return;
  const float *src = Zcm + (size_t)(32 * mb + l31) * d;
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 16 * kg + 8 * (e >> 2) + 4 * h + (e & 3);
    x[e] = k < d ? src[k] : 0.f;   // (k >= p meets the zero padding of X)
  }
  fb_store_frag(ZP + (size_t)f * kFrag + 4 * lane, x);
}
__global__ __launch_bounds__(512, 6) void k_lr_logits_planes(LrMfmaArgs a) {
  constexpr int NR = 3, kPW = 2;
  __shared__ __attribute__((aligned(16))) unsigned lds[NR * kStageW];
  __shared__ float ll_lds[128];
  __shared__ float y_lds[128];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;
  const long long tile = blockIdx.x, row0 = tile * 128;
  const int ng = a.ldx >> 4, G = ng;
  const int mb0 = blockIdx.y * 4;
  // wave w stages fragment w of every stage: w < 4: X's 32-row block 4 tile + w; else the samples' block mb0 + w - 4
  const unsigned *sp = (w < 4 ? a.XA + (size_t)(4 * tile + w) * ng * kFrag : a.ZP + (size_t)(mb0 + w - 4) * ng * kFrag) + 4 * lane;
  auto issue = [&](int slot) {
    unsigned *dst = lds + slot * kStageW + w * 512;
    FB_GLDS16(sp, dst, 0);
    FB_GLDS16(sp, dst, 1024);
    sp += kFrag;
  };
  f32x16 acc[2][1];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
  issue(0); issue(1);
  if (tid < 128) {
    ll_lds[tid] = 0.f;
    const long long r = row0 + tid;
    y_lds[tid] = r < a.n ? (float)a.y[r] : 0.f;
  }
  int slot = 0;
  for (int g = 0; g < G; ++g) {
    if (g + 1 < G) fb_wait_vm<kPW>();
    else fb_wait_vm<0>();
    fb_barrier();
    if (g + 2 < G) issue(slot == 0 ? 2 : slot - 1);
    FbFrags<1> F;
    fb_read_frags<1>(lds, slot, wm, wn, lane, F);
    fb_group<1>(F, acc);
    slot = slot == 2 ? 0 : slot + 1;
  }
  float xs, xinv;
  lr_xscale(a.xmax, xs, xinv);
  const int m = 128 * blockIdx.y + 32 * wn + l31;
  float ll = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float res[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lr = 64 * wm + 32 * i + 8 * (r >> 2) + 4 * h + (r & 3);
      const long long row = row0 + lr;
      res[r] = 0.f;
      if (row < a.n) {
        const float yv = y_lds[lr];
        const float lg = acc[i][0][r] * xinv, e = __expf(-fabsf(lg)), inv = __frcp_rn(1.f + e);
        ll += yv * lg - (fmaxf(lg, 0.f) + __logf(1.f + e));
        res[r] = yv - (lg >= 0.f ? inv : e * inv);
        if (a.want_grad && !a.RP) a.R[(size_t)row * a.ldr + m] = res[r];
      }
    }
    if (a.want_grad && a.RP) {
      // the residuals as the A operand of X^T R (rows = samples, k = data rows): fragment (sample block, 16-row group g2 of this 32-row
      // block) = this lane's own registers 4 (2 g2 + e / 4) + e % 4 -- no transposition; 2^13 r (|r| < 1), rows beyond n are zeros
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = res[4 * (2 * g2 + (e >> 2)) + (e & 3)] * 8192.f;
        const long long rg = (row0 + 64 * wm + 32 * i) / 16 + g2;
        fb_store_frag(a.RP + ((size_t)(4 * blockIdx.y + wn) * a.nrg + rg) * kFrag + 4 * lane, x);
      }
    }
  }
  ll += __shfl_xor(ll, 32, 64);
  if (h == 0) atomicAdd(&ll_lds[32 * wn + l31], ll);
  __syncthreads();
  if (tid < 128) a.ll_part[(size_t)tile * a.M + 128 * blockIdx.y + tid] = (double)ll_lds[tid];
}

// X^T R on operand planes: G[m, k] = sum_r R[r, m] X[r, k].  A = the residuals' planes (rows = samples; left by k_lr_logits_planes straight from
// its accumulators), B = X's planes in the second orientation (rows = features, k = data rows; k_lr_xbplanes, once per data set); the same
// three-slot LDS-DMA ring; the data rows are split over gridDim.y workgroups per (128 samples x 128 features) tile, partial sums to
// g_part[split][m][k] (k_lr_greduce adds them in a fixed order).  blockIdx.x = feature group: the groups of one row range run side by side
// and share its residual planes in the memory-side cache.
__global__ __launch_bounds__(256) void k_lr_xbplanes(long long n, long long nrg, int ldx, const float *Xrm, const unsigned *xmax, unsigned *XB) {
  const long long f = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  const long long fb = f / nrg, rg = f % nrg;
  const long long nfb32 = (long long)((ldx + 127) / 128) * 4;   // k_lr_xtr_planes stages whole 128-feature groups: the blocks past ldx / 32 are zeros
  if (fb >= nfb32) return;
  float s, inv;
  lr_xscale(xmax, s, inv);
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const long long row = 16 * rg + 8 * (e >> 2) + 4 * h + (e & 3);
    x[e] = (row < n && fb < (ldx >> 5)) ? Xrm[(size_t)row * ldx + 32 * fb + l31] * s : 0.f;
  }
  fb_store_frag(XB + (size_t)f * kFrag + 4 * lane, x);
}
__global__ __launch_bounds__(512, 6) void k_lr_xtr_planes(LrMfmaArgs a) {
  constexpr int NR = 3, kPW = 2;
  __shared__ __attribute__((aligned(16))) unsigned lds[NR * kStageW];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;
  // Workgroup -> (feature group, split): the feature groups of ONE split read the same residual planes, so they go to the same XCD side by
  // side (dispatch order b lands on XCD b % 8): FETCH_SIZE had the residuals fetched once per feature group (2 GB at C3 instead of 0.5);
  // 557 -> 476 us at C3.
  int fg = blockIdx.x, sp_i = blockIdx.y;
  const int mg = blockIdx.z;
  if ((gridDim.y & 7) == 0) {
    const int b = (int)(blockIdx.x + gridDim.x * blockIdx.y), j = b >> 3;
    fg = j % (int)gridDim.x;
    sp_i = (j / (int)gridDim.x) * 8 + (b & 7);
  }
  const long long gbeg = (long long)sp_i * a.gps;
  const long long gend = gbeg + a.gps < a.nrg ? gbeg + a.gps : a.nrg;
  const int G = (int)(gend - gbeg);
  const unsigned *sp = (w < 4 ? a.RP + ((size_t)(4 * mg + w) * a.nrg + gbeg) * kFrag : a.XB + ((size_t)(4 * fg + w - 4) * a.nrg + gbeg) * kFrag) + 4 * lane;
  auto issue = [&](int slot) {
    unsigned *dst = lds + slot * kStageW + w * 512;
    FB_GLDS16(sp, dst, 0);
    FB_GLDS16(sp, dst, 1024);
    sp += kFrag;
  };
  f32x16 acc[2][1];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
  issue(0);
  if (G > 1) issue(1);
  int slot = 0;
  for (int g = 0; g < G; ++g) {
    if (g + 1 < G) fb_wait_vm<kPW>();
    else fb_wait_vm<0>();
    fb_barrier();
    if (g + 2 < G) issue(slot == 0 ? 2 : slot - 1);
    FbFrags<1> F;
    fb_read_frags<1>(lds, slot, wm, wn, lane, F);
    fb_group<1>(F, acc);
    slot = slot == 2 ? 0 : slot + 1;
  }
  float xs, xinv;
  lr_xscale(a.xmax, xs, xinv);
  const float f = xinv * (1.f / 8192.f);
  const int k = 128 * fg + 32 * wn + l31;
  if (k < a.p) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 128 * mg + 64 * wm + 32 * i + 8 * (r >> 2) + 4 * h + (r & 3);
        a.g_part[((size_t)sp_i * a.M + m) * a.p + k] = acc[i][0][r] * f;
      }
  }
}

// the full-tile kernel is pinned to 128 VGPRs (4 waves per SIMD, two workgroups per CU); the partial-tile variant's extra
// control flow does not fit that budget without scratch (228 B/lane, 3x slower), so it runs unconstrained
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_lr_logits_f16x2(LrMfmaArgs a) {
  lr_logits_f16x2_body<false>(a);
}
__global__ __launch_bounds__(512) void k_lr_logits_f16x2_part(LrMfmaArgs a) { lr_logits_f16x2_body<true>(a); }

// X^T R on the 16-bit matrix cores, same two-way f16 scheme.  The contraction runs over data rows, so both operands need 8
// consecutive ROWS per lane while memory has rows outermost: a thread loads a 4-row x 1-column strip (lanes along the
// contiguous axis: coalesced dword loads), splits it, and writes the 4 row-consecutive f16 of each piece as one 8-byte
// LDS word into [column][16 rows] tiles -- the transposition costs nothing.  Column stride 40 bytes: conflict-free for
// the 8-byte writes and the 8-byte operand reads (a 16-byte operand = two reads).
__device__ __forceinline__ void lr_split2s(const float (&v)[4], float s, lr_f16x4 &hi, lr_f16x4 &lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x = v[i] * s;
    const _Float16 h = (_Float16)x;
    hi[i] = h;
    lo[i] = (_Float16)(x - (float)h);
  }
}

// NWK = waves along the feature axis: 4 -> 256-feature tile, 512 threads, one workgroup per CU (91 KB of LDS);
// 2 -> 128-feature tile, 256 threads, 61 KB: two workgroups per CU that are not in barrier lock-step with each other (R is
// then read by four feature groups instead of two).  Measured at C3: 880 us against 790 us for NWK = 4, the only one instantiated.
template <int NWK, bool PART>
__global__ __launch_bounds__(128 * NWK) void k_lr_xtr_f16x2(LrMfmaArgs a) {
  constexpr int NT = 128 * NWK, KT = 64 * NWK, RN = 4 / NWK;   // threads, features per tile, R strips per thread
  constexpr int CS = 20;   // column stride in f16 (40 bytes)
  __shared__ __attribute__((aligned(16))) _Float16 Rs[2][2][128 * CS];   // [slot][piece][sample][16 rows]
  __shared__ __attribute__((aligned(16))) _Float16 Xs[2][2][KT * CS];   // [slot][piece][feature][16 rows]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / NWK, wk = w % NWK;
  const int mbase = blockIdx.z * 128, kbase = blockIdx.y * KT;
  const int m0 = mbase + wm * 64, k0 = kbase + wk * 64;
  // R and Xrm carry zero rows up to the next multiple of 16 (logreg_mfma / logreg_prepare_f32), so every 16-row stage
  // is whole: no per-row clamps or masks, and an address is one uniform stage base (SALU) + a per-lane offset that never
  // changes (the 64-bit per-load address arithmetic this replaces was 100 of the 266 VALU instructions of a stage)
  const long long rbeg = (long long)blockIdx.x * a.rows_per_split;
  const long long rend = min((a.n + 15) / 16 * 16, rbeg + a.rows_per_split);
  const int ldr = a.ldr, ldx = a.ldx;
  const int nst = (int)((rend - rbeg) / 16);
  // strips (4 rows x 1 column per load group): R  sample tid & 127, row groups tid >> 7 (+ NT/128 per extra strip);
  // X  feature tid % KT, row groups q = tid / KT and q + 2
  const int rm = tid & 127, rrg = tid >> 7;
  const int xf = tid % KT, xrg = tid / KT;
  const int rcol = min(mbase + rm, ldr - 1), xcol = min(kbase + xf, ldx - 1);
  const float *Rb = a.R + (size_t)rbeg * ldr, *Xb = a.Xrm + (size_t)rbeg * ldx;
  const int roff = 4 * rrg * ldr + rcol, xoff = 4 * xrg * ldx + xcol;
  struct G { float r[RN][4], x0[4], x1[4]; };
  auto gload = [&](int st, G &g) {
    st = min(st, nst - 1);
    const float *rs = Rb + (size_t)st * 16 * ldr, *xs = Xb + (size_t)st * 16 * ldx;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int u = 0; u < RN; ++u) g.r[u][i] = (rs + (4 * u * (NT / 128) + i) * ldr)[roff];
      g.x0[i] = (xs + i * ldx)[xoff];
      g.x1[i] = (xs + (8 + i) * ldx)[xoff];
    }
  };
  float xs, xinv;
  lr_xscale(a.xmax, xs, xinv);
  auto lstore = [&](int slot, const G &g) {
    lr_f16x4 p0, p1;
#pragma unroll
    for (int u = 0; u < RN; ++u) {
      lr_split2s(g.r[u], 1.f, p0, p1);   // (residuals y - sigmoid: in (-1, 1))
      *(lr_f16x4 *)&Rs[slot][0][rm * CS + 4 * (rrg + u * (NT / 128))] = p0;
      *(lr_f16x4 *)&Rs[slot][1][rm * CS + 4 * (rrg + u * (NT / 128))] = p1;
    }
    lr_split2s(g.x0, xs, p0, p1);
    *(lr_f16x4 *)&Xs[slot][0][xf * CS + 4 * xrg] = p0;
    *(lr_f16x4 *)&Xs[slot][1][xf * CS + 4 * xrg] = p1;
    lr_split2s(g.x1, xs, p0, p1);
    *(lr_f16x4 *)&Xs[slot][0][xf * CS + 4 * xrg + 8] = p0;
    *(lr_f16x4 *)&Xs[slot][1][xf * CS + 4 * xrg + 8] = p1;
  };
  lr_f32x16 c00, c01, c10, c11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { c00[r] = 0.f; c01[r] = 0.f; c10[r] = 0.f; c11[r] = 0.f; }
  const int ao = (wm * 64 + l31) * CS + 8 * h, bo = (wk * 64 + l31) * CS + 8 * h;
  const bool has0 = m0 < a.M, has1 = m0 + 32 < a.M;   // wave-uniform (PART: sample tiles beyond M are skipped)
  auto ld8 = [&](const _Float16 *p) {   // 8 consecutive rows of one column = two 8-byte words
    const lr_f16x4 lo4 = *(const lr_f16x4 *)p, hi4 = *(const lr_f16x4 *)(p + 4);
    lr_f16x8 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = lo4[i]; v[4 + i] = hi4[i]; }
    return v;
  };
  auto compute = [&](int slot) {
    lr_f16x8 A0[2], A1[2], B0[2], B1[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      A0[s] = ld8(&Rs[slot][s][ao]);
      A1[s] = ld8(&Rs[slot][s][ao + 32 * CS]);
      B0[s] = ld8(&Xs[slot][s][bo]);
      B1[s] = ld8(&Xs[slot][s][bo + 32 * CS]);
    }
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (!PART || has0) {   // A = residuals of samples m0 .. m0+31
        c00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0[PA[t]], B0[PB[t]], c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0[PA[t]], B1[PB[t]], c01, 0, 0, 0);
      }
      if (!PART || has1) {   // samples m0+32 .. m0+63
        c10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1[PA[t]], B0[PB[t]], c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1[PA[t]], B1[PB[t]], c11, 0, 0, 0);
      }
    }
  };
  if (nst > 0) {
    G ga, gb;
    gload(0, ga);
    lstore(0, ga);
    gload(1, ga);
    gload(2, gb);
    lds_barrier();
    int st = 0;
    while (true) {
      lstore((st + 1) & 1, ga);
      gload(st + 3, ga);
      compute(st & 1);
      lds_barrier();
      if (++st >= nst) break;
      lstore((st + 1) & 1, gb);
      gload(st + 3, gb);
      compute(st & 1);
      lds_barrier();
      if (++st >= nst) break;
    }
  }
  auto epi = [&](const lr_f32x16 &c, int mb, int kb) {
    const int k = k0 + kb * 32 + l31;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = m0 + mb * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
      if (k < a.p && m < a.M) a.g_part[((size_t)blockIdx.x * a.M + m) * a.p + k] = c[q] * xinv;
    }
  };
  epi(c00, 0, 0);
  epi(c01, 0, 1);
  epi(c10, 1, 0);
  epi(c11, 1, 1);
}

// one-time: row-major zero-padded copy of X (n x p column-major -> n x ldx row-major), 64x64 LDS transpose
__global__ __launch_bounds__(256) void k_lr_make_xrm(long long n, int p, int ldx, const float *X, float *Xrm, unsigned *xmax) {
  __shared__ float tile[64][65];
  float amax = 0.f;
  const long long r0 = (long long)blockIdx.x * 64;
  const int k0 = blockIdx.y * 64, tid = threadIdx.x;
  const int a = tid & 63, b = tid >> 6;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int kc = b + 4 * j;
    const long long r = r0 + a;
    const int k = k0 + kc;
    const float v = (r < n && k < p) ? X[(size_t)k * n + r] : 0.f;
    tile[kc][a] = v;
    amax = fmaxf(amax, fabsf(v));
  }
  amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  if ((tid & 63) == 0 && amax > 0.f) atomicMax(xmax, __builtin_bit_cast(unsigned, amax));   // (non-negative floats order like their bits)
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int rr = b + 4 * j;
    const long long r = r0 + rr;
    const int k = k0 + a;
    if (r < n && k < ldx) Xrm[(size_t)r * ldx + k] = tile[a][rr];
  }
}

// g_part[0][e] = sum_s g_part[s][e] in a fixed order (coalesced over e); k_lr_finish then reads one slab
__global__ __launch_bounds__(256) void k_lr_greduce(int S, size_t len, float *g_part) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= len) return;
  float s = g_part[e];
  int i = 1;
  for (; i + 16 <= S; i += 16) {   // sixteen loads in flight, summed in the same fixed order (one load per add was a
    float v[16];                   // latency chain: 29 us for S = 128 at n = 20 000)
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = g_part[(size_t)(i + u) * len + e];
#pragma unroll
    for (int u = 0; u < 16; ++u) s += v[u];
  }
  for (; i < S; ++i) s += g_part[(size_t)i * len + e];
  g_part[e] = s;
}

// grow-only device buffer; false on allocation failure (the buffer is then empty)
static bool grow(DevBuf &b, size_t bytes) {
  if (b.bytes >= bytes) return true;
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.bytes = 0;
  if (hipMalloc(&b.p, bytes) != hipSuccess) { (void)hipGetLastError(); b.p = nullptr; return false; }
  b.bytes = bytes;
  return true;
}

bool logreg_prepare_f32(mivi_ctx *c) {
  // builds Xrm in c->lr_Xrm (called by mivi_set_target_logreg for MIVI_F32)
  const int p = c->cfg.d - 1;
  const int ldx = (p + 31) / 32 * 32;
  const size_t n16 = ((size_t)c->lr_n + 15) / 16 * 16;   // zero rows up to a whole 16-row stage (k_lr_xtr_f16x2)
  const size_t bytes = n16 * ldx * sizeof(float);
  if (!grow(c->lr_Xrm, bytes)) return false;
  if (n16 > (size_t)c->lr_n &&
      hipMemsetAsync((float *)c->lr_Xrm.p + (size_t)c->lr_n * ldx, 0, (n16 - (size_t)c->lr_n) * ldx * sizeof(float), c->stream) != hipSuccess)
    return false;
  if (!grow(c->lr_xmax, 64) || hipMemsetAsync(c->lr_xmax.p, 0, 64, c->stream) != hipSuccess) return false;
  dim3 grid((unsigned)((c->lr_n + 63) / 64), (ldx + 63) / 64);
  hipLaunchKernelGGL(k_lr_make_xrm, grid, dim3(256), 0, c->stream, (long long)c->lr_n, p, ldx, (const float *)c->lr_X,
                     (float *)c->lr_Xrm.p, (unsigned *)c->lr_xmax.p);
  // X as A-operand planes for k_lr_logits_planes (as many bytes again as X; problems large enough for the matrix-core route only)
  if ((double)c->lr_n * p >= 1.0e5) {
    const long long nrb32 = (c->lr_n + 127) / 128 * 4;
    const size_t nfr = (size_t)nrb32 * (ldx / 16);
    if (!grow(c->lr_XA, nfr * kFrag * 4)) return false;
    hipLaunchKernelGGL(k_lr_xplanes, dim3((unsigned)((nfr + 3) / 4)), dim3(256), 0, c->stream, (long long)c->lr_n, nrb32, ldx, (const float *)c->lr_Xrm.p,
                       (const unsigned *)c->lr_xmax.p, (unsigned *)c->lr_XA.p);
    const long long nrg = nrb32 / 4 * 8;
    const size_t nfb = (size_t)((ldx + 127) / 128 * 4) * nrg;   // (whole 128-feature groups: k_lr_xtr_planes stages four 32-feature blocks per group)
    if (!grow(c->lr_XB, nfb * kFrag * 4)) return false;
    hipLaunchKernelGGL(k_lr_xbplanes, dim3((unsigned)((nfb + 3) / 4)), dim3(256), 0, c->stream, (long long)c->lr_n, nrg, ldx, (const float *)c->lr_Xrm.p,
                       (const unsigned *)c->lr_xmax.p, (unsigned *)c->lr_XB.p);
  } else {
    if (c->lr_XA.p) (void)hipFree(c->lr_XA.p);
    if (c->lr_XB.p) (void)hipFree(c->lr_XB.p);
    c->lr_XA = DevBuf{};
    c->lr_XB = DevBuf{};
  }
  return true;
}

// minibatch gather: column-major subset (generic / f64 route), labels, and -- f32 -- whole rows of the row-major copy
template <typename T>
__global__ void k_lr_gather_cm(long long n, long long b, int p, const long long *idx, const T *X, const uint8_t *y, T *Xs,
                               uint8_t *ys) {
  const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
  if (j >= b) return;
  const long long r = idx[j];
  const int k = blockIdx.y;
  Xs[(size_t)k * b + j] = X[(size_t)k * n + r];
  if (k == 0) ys[j] = y[r];
}
__global__ void k_lr_gather_rm(long long b, int ldx, const long long *idx, const float *Xrm, float *Xs) {
  const long long j = blockIdx.x;   // rows b .. gridDim.x-1: the zero padding up to a whole 16-row stage
  const float4 *src = (const float4 *)(Xrm + (size_t)(j < b ? idx[j] : 0) * ldx);
  float4 *dst = (float4 *)(Xs + (size_t)j * ldx);
  for (int t = threadIdx.x; t < ldx / 4; t += 128) dst[t] = j < b ? src[t] : make_float4(0.f, 0.f, 0.f, 0.f);
}
void launch_logreg_gather(mivi_ctx *c, int64_t b) {
  const int p = c->cfg.d - 1;
  const dim3 g((unsigned)((b + 255) / 256), p);
  const long long *idx = (const long long *)c->lr_idx.p;
  if (c->cfg.dtype == MIVI_F32) {
    hipLaunchKernelGGL(k_lr_gather_cm<float>, g, dim3(256), 0, c->stream, (long long)c->lr_n_full, (long long)b, p, idx,
                       (const float *)c->lr_X_full, c->lr_y_full, (float *)c->lr_Xsub.p, (uint8_t *)c->lr_ysub.p);
    const int ldx = (p + 31) / 32 * 32;
    hipLaunchKernelGGL(k_lr_gather_rm, dim3((unsigned)((b + 15) / 16 * 16)), dim3(128), 0, c->stream, (long long)b, ldx, idx,
                       (const float *)c->lr_Xrm.p, (float *)c->lr_Xrm_sub.p);
  } else {
    hipLaunchKernelGGL(k_lr_gather_cm<double>, g, dim3(256), 0, c->stream, (long long)c->lr_n_full, (long long)b, p, idx,
                       (const double *)c->lr_X_full, c->lr_y_full, (double *)c->lr_Xsub.p, (uint8_t *)c->lr_ysub.p);
  }
}

// Scratch geometry of the two routes.  logreg_reserve() sizes the buffers ahead of time (and is what makes the target
// graph-capturable: no allocation at launch); the launchers call it again as a no-op / safety net.
struct LrGeom {
  bool mfma, planes, xplanes;   // planes: k_lr_logits_planes (128-row tiles on the prebuilt planes of X); xplanes: also k_lr_xtr_planes
  long long nrg;
  int gps;
  int nrb, S, ldr;
  long long rps;
  size_t need_R, need_g, need_ll;
};
static LrGeom lr_geom(const mivi_ctx *c, int M) {
  static const bool force_generic = getenv("MIVI_LOGREG_GENERIC") != nullptr;
  LrGeom g;
  g.planes = false;
  g.xplanes = false;
  g.nrg = 0;
  g.gps = 0;
  const long long n = c->lr_n;
  const int p = c->cfg.d - 1;
  // small problems take the VALU route: the matrix-core kernels carry fixed 256-row x 128-sample tiles and three more
  // launches (measured crossover around n p M = 2e7: n = 1000, p = 32: 35 vs 47 us at 16 samples, 50 vs 57 us at 128)
  static const bool force_mfma = getenv("MIVI_LOGREG_MFMA") != nullptr;
  g.mfma = c->cfg.dtype == MIVI_F32 && c->lr_Xrm_act && !force_generic && c->lr_route != 2 &&
           (force_mfma || c->lr_route == 1 || (double)n * (double)p * (double)M >= 1.6e7);
  if (g.mfma) {
    g.ldr = (M + 63) / 64 * 64;
    static const bool no_planes = getenv("MIVI_LR_NO_PLANES") != nullptr;   // A/B: the logits kernel that splits X in the tile (k_lr_logits_f16x2)
    g.planes = M % 128 == 0 && c->lr_XA.p && c->lr_Xrm_act == c->lr_Xrm.p && !no_planes && !getenv("MIVI_LR_F32_LOGITS");
    g.nrb = g.planes ? (int)((n + 127) / 128) : (int)((n + 255) / 256);
    // row splits of X^T R: 128 x 2 feature groups = one workgroup per CU at C3 (n = 1e6, p = 511); small data sets still
    // get a split per 128 rows (one workgroup walking n = 20 000 rows alone is a 270 us chain of 16-row stages)
    int S = (int)((n + 127) / 128);
    if (S > 128) S = 128;   // (256 / 384 splits -- two workgroups per CU now that a tile's LDS is 60 KB -- measured 713 / 701 against 737 estimates/s at C3)
    if (S < 1) S = 1;
    long long rps = (n + S - 1) / S;
    rps = (rps + 15) / 16 * 16;
    g.S = (int)((n + rps - 1) / rps);
    g.rps = rps;
    g.need_R = ((size_t)((n + 15) / 16 * 16) * g.ldr * sizeof(float) + 255) / 256 * 256;   // + zero rows to a whole stage
    static const bool no_xplanes = getenv("MIVI_LR_NO_XPLANES") != nullptr;   // A/B: X^T R with the splits made in the tile (k_lr_xtr_f16x2)
    g.nrg = (n + 127) / 128 * 8;
    g.xplanes = g.planes && c->lr_XB.p && !no_xplanes && !getenv("MIVI_LR_F32_XTR") && g.nrg >= 16;
    if (g.xplanes) {
      // row splits: three workgroups per CU over the feature groups (at C3: 4 feature groups x 192 splits = 768 workgroups), at least 8 row groups each
      int S2 = (int)(768 / ((p + 127) / 128) / (M / 128));
      if (S2 > g.nrg / 8) S2 = (int)(g.nrg / 8);
      if (S2 < 1) S2 = 1;
      g.gps = (int)((g.nrg + S2 - 1) / S2);
      g.S = (int)((g.nrg + g.gps - 1) / g.gps);
      g.need_R = (size_t)(M / 32) * g.nrg * kFrag * 4;
    }
    g.need_g = (size_t)g.S * p * M * sizeof(float);
    g.need_ll = (size_t)g.nrb * M * sizeof(double);
  } else {
    g.ldr = M;
    g.nrb = (int)((n + 63) / 64);
    // row splits of the X^T r pass: at least 64 rows each, up to 128 of them (one 64 x 64 output tile per workgroup
    // otherwise leaves a README-sized problem, n = 1000, on a single workgroup: 125 us)
    int S = (int)((n + 63) / 64);
    if (S > 128) S = 128;
    if (S < 1) S = 1;
    long long rps = (n + S - 1) / S;
    rps = (rps + 15) / 16 * 16;
    g.S = (int)((n + rps - 1) / rps);
    g.rps = rps;
    g.need_R = (size_t)n * M * c->esize;
    g.need_g = (size_t)g.S * p * M * c->esize;
    g.need_ll = (size_t)g.nrb * M * sizeof(double);
  }
  return g;
}
bool logreg_uses_mfma(const mivi_ctx *c, int M) { return c->target == TGT_LOGREG && lr_geom(c, M).mfma; }
int logreg_kernel_bits(const mivi_ctx *c, int M) {
  if (c->target != TGT_LOGREG) return 0;
  const LrGeom g = lr_geom(c, M);
  return (g.mfma ? 1 : 0) | (g.planes ? 2 : 0) | (g.xplanes ? 4 : 0);
}
bool logreg_reserve(mivi_ctx *c, int M) {
  const LrGeom g = lr_geom(c, M);
  if (g.planes && !grow(c->lr_ZP, (size_t)(M / 32) * (((c->cfg.d - 1 + 31) / 32 * 32) / 16) * kFrag * 4)) return false;
  return grow(c->lr_scratch, g.need_R + g.need_g) && grow(c->lr_part, g.need_ll);
}

static bool logreg_mfma(mivi_ctx *c, int M, int want_grad) {
  if (!logreg_reserve(c, M)) return false;
  const LrGeom geo = lr_geom(c, M);
  LrMfmaArgs a;
  a.d = c->cfg.d;
  a.p = a.d - 1;
  a.M = M;
  a.n = c->lr_n;
  a.X = (const float *)c->lr_X;
  a.Xrm = (const float *)c->lr_Xrm_act;
  a.ldx = (a.p + 31) / 32 * 32;
  a.y = c->lr_y;
  a.ZT = (const float *)c->RT.p;
  a.ldz = c->MP;
  a.ldr = (M + 63) / 64 * 64;
  const int nrb = geo.nrb, S = geo.S;
  const long long rps = geo.rps;
  a.rows_per_split = rps;
  a.want_grad = want_grad;
  const size_t need_R = geo.need_R;
  a.R = (float *)c->lr_scratch.p;
  a.g_part = (float *)((char *)c->lr_scratch.p + need_R);
  a.ll_part = (double *)c->lr_part.p;
  static const bool no_split = getenv("MIVI_LR_F32_LOGITS") != nullptr;   // A/B: f32 MFMA logits
  a.Zcm = (const float *)c->Z.p;
  a.xmax = (const unsigned *)c->lr_xmax.p;
  const bool part = M % 128 != 0 && M % 128 <= 96;   // whole 32-sample tiles of the last 128-sample group are empty
  if (geo.planes) {
    const int ng = a.ldx / 16, nfz = (M / 32) * ng;
    if (!grow(c->lr_ZP, (size_t)nfz * kFrag * 4)) return false;
    a.XA = (const unsigned *)c->lr_XA.p;
    a.ZP = (unsigned *)c->lr_ZP.p;
    a.XB = (const unsigned *)c->lr_XB.p;
    a.RP = geo.xplanes ? (unsigned *)c->lr_scratch.p : nullptr;   // (the residuals' planes take the place of the f32 R)
    a.nrg = geo.nrg;
    a.gps = geo.gps;
    hipLaunchKernelGGL(k_lr_zplanes, dim3((nfz + 3) / 4), dim3(256), 0, c->stream, M, a.d, a.ldx, a.Zcm, a.ZP);
    hipLaunchKernelGGL(k_lr_logits_planes, dim3(nrb, M / 128), dim3(512), 0, c->stream, a);
  } else if (a.d % 4 == 0 && a.d >= 4 && !no_split) {
    if (part) hipLaunchKernelGGL(k_lr_logits_f16x2_part, dim3(nrb, (M + 127) / 128), dim3(512), 0, c->stream, a);
    else hipLaunchKernelGGL(k_lr_logits_f16x2, dim3(nrb, (M + 127) / 128), dim3(512), 0, c->stream, a);
  } else {
    if (part) hipLaunchKernelGGL(k_lr_logits_mfma_lds<true>, dim3(nrb, (M + 127) / 128), dim3(512), 0, c->stream, a);
    else hipLaunchKernelGGL(k_lr_logits_mfma_lds<false>, dim3(nrb, (M + 127) / 128), dim3(512), 0, c->stream, a);
  }
  if (want_grad && a.n % 16 != 0 && !geo.xplanes) {
    // the zero residual rows k_lr_xtr_f16x2's last stage reads (the logits kernels stop at n).  Nobody else writes them, but other
    // routes reuse lr_scratch (the residuals' planes, the generic route's layout), so they are zeroed on every call: at most 15 rows
    // of ldr floats, in stream order (a captured graph carries its own)
    (void)hipMemsetAsync(a.R + (size_t)a.n * a.ldr, 0, (size_t)(16 - a.n % 16) * a.ldr * sizeof(float), c->stream);
  }
  if (want_grad && geo.xplanes) {
    hipLaunchKernelGGL(k_lr_xtr_planes, dim3((a.p + 127) / 128, S, M / 128), dim3(512), 0, c->stream, a);
  } else if (want_grad) {
    const dim3 gx(S, (a.p + 255) / 256, (M + 127) / 128);
    static const bool xtr_f32 = getenv("MIVI_LR_F32_XTR") != nullptr;   // A/B: f32 MFMA X^T R
    if (!xtr_f32) {
      if (part) hipLaunchKernelGGL((k_lr_xtr_f16x2<4, true>), gx, dim3(512), 0, c->stream, a);
      else hipLaunchKernelGGL((k_lr_xtr_f16x2<4, false>), gx, dim3(512), 0, c->stream, a);
    }
    else hipLaunchKernelGGL(k_lr_xtr_mfma_lds, gx, dim3(512), 0, c->stream, a);
  }
  if (want_grad && S > 1) {
    const size_t len = (size_t)a.p * M;
    hipLaunchKernelGGL(k_lr_greduce, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, c->stream, S, len, a.g_part);
  }
  // finish (shared with the generic route)
  LrArgs<float> f;
  f.d = a.d; f.p = a.p; f.M = M; f.n = a.n;
  f.X = a.X; f.y = a.y; f.Z = (const float *)c->Z.p;
  f.R = nullptr; f.ll_part = a.ll_part; f.g_part = a.g_part;
  f.S = 1; f.nrb = nrb; f.rows_per_split = rps;
  f.G = (float *)c->W.p; f.ell = (float *)c->ell.p;
  f.variant = c->lr_variant; f.likeadj = c->lr_likeadj; f.want_grad = want_grad;
  hipLaunchKernelGGL(k_lr_finish<float>, dim3(M), dim3(256), 0, c->stream, f);
  return true;
}

template <typename T>
static bool logreg_impl(mivi_ctx *c, int M, int want_grad) {
  if (!logreg_reserve(c, M)) return false;
  const LrGeom geo = lr_geom(c, M);
  LrArgs<T> a;
  a.d = c->cfg.d;
  a.p = a.d - 1;
  a.M = M;
  a.n = c->lr_n;
  a.X = (const T *)c->lr_X;
  a.y = c->lr_y;
  a.Z = (const T *)c->Z.p;
  a.nrb = geo.nrb;
  const int S = geo.S;
  a.S = S;
  a.rows_per_split = geo.rps;
  const size_t need_R = geo.need_R;
  // scratch layout inside lr_scratch: [R | g_part], lr_part: ll_part
  a.R = (T *)c->lr_scratch.p;
  a.g_part = (T *)((char *)c->lr_scratch.p + need_R);
  a.ll_part = (double *)c->lr_part.p;
  a.G = (T *)c->W.p;
  a.ell = (T *)c->ell.p;
  a.variant = c->lr_variant;
  a.likeadj = c->lr_likeadj;
  a.want_grad = want_grad;
  hipLaunchKernelGGL(k_lr_logits<T>, dim3(a.nrb, (M + 63) / 64), dim3(256), 0, c->stream, a);
  if (want_grad)
    hipLaunchKernelGGL(k_lr_xtr<T>, dim3((a.p + 63) / 64, (M + 63) / 64, S), dim3(256), 0, c->stream, a);
  hipLaunchKernelGGL(k_lr_finish<T>, dim3(M), dim3(256), 0, c->stream, a);
  return true;
}

// false: device allocation of the residual / partial buffers failed
bool launch_logreg_target(mivi_ctx *c, int M, int want_grad) {
  if (lr_geom(c, M).mfma) return logreg_mfma(c, M, want_grad);
  if (c->cfg.dtype == MIVI_F32) return logreg_impl<float>(c, M, want_grad);
  return logreg_impl<double>(c, M, want_grad);
}

}  // namespace mivi
