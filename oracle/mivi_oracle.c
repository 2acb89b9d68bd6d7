/*
 * CPU ORACLE, C leg (TEST INFRASTRUCTURE / CPU BASELINE ONLY -- never linked into libmivi).
 *
 * Plain-C restatement of the AdvancedVI.jl v0.7.0 RepGradELBO estimate over a location-scale
 * Gaussian family with the reference's benchmark target MvNormal(mean, Diagonal(std^2))
 * (bench/benchmarks.jl:43-47, test/models/normal.jl:56-75).  It follows, per sample column
 * (src/utils.jl:6):
 *     z   = scale * eps + mu                 src/families/location_scale.jl:71-87
 *     ell = logdensity(prob, z)              src/algorithms/repgradelbo.jl:84-86
 *     H   = estimate_entropy(...)            src/algorithms/entropy.jl:13-90 (location_scale.jl:52-63)
 *     f   = -(mean ell + H)                  src/algorithms/repgradelbo.jl:142-149
 * and the gradient of f that the reference obtains by AD, written in closed form.
 * This is strictly cheaper than the reference's AD-taped path, so GPU/CPU ratios quoted from it
 * are conservative ("kind": "port" in bench.py).  Parity unpinned against real Julia output (no
 * Julia toolchain in the image); pinned against the numpy oracle in tests/test_oracle_c.py.
 *
 * Built twice by oracle/Makefile: -DREAL=double (mo64_*) and -DREAL=float (mo32_*).
 * OpenMP; `mo*_set_threads` selects the thread count.  The two full-rank contractions (tril(C) eps and tril(W eps')) run as
 * packed, register-blocked panels (2 SIMD vectors of rows x 6 columns per micro-tile, operands packed so that every inner-loop
 * load is contiguous), one row panel per OpenMP task, heaviest panels first: what a BLAS trmm / syrk-shaped kernel does, so the
 * threads scale and the single-thread rate is a sane fraction of the core's FMA peak (bench.py prints GFLOP/s beside est/s).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef REAL
#define REAL double
#endif
#ifndef PFX
#define PFX mo64_
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(PFX, name)

#define LOG2PI 1.8378770664093454835606594728112

/* ---- Philox4x32-10 + Box-Muller: the same eps stream as advancedvi.jl_amd/csrc/philox.h ---- */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

static void box_muller(uint32_t wa, uint32_t wb, REAL *n0, REAL *n1) {
  double ua, ub;
  if (sizeof(REAL) == 4) {
    ua = ((double)(wa >> 9) + 0.5) * 1.1920928955078125e-07;
    ub = ((double)(wb >> 9) + 0.5) * 1.1920928955078125e-07;
  } else {
    ua = ((double)wa + 0.5) * 2.3283064365386962890625e-10;
    ub = ((double)wb + 0.5) * 2.3283064365386962890625e-10;
  }
  const double r = sqrt(-2.0 * log(ua)), ang = 6.283185307179586476925286766559 * ub;
  *n0 = (REAL)(r * cos(ang));
  *n1 = (REAL)(r * sin(ang));
}

/* eps (d x M column-major) of estimate `idx`, global columns m_offset .. m_offset+M-1 */
void FN(fill_eps)(uint64_t seed, uint64_t idx, int d, int M, int m_offset, REAL *eps) {
  const int d4 = (d + 3) / 4;
#pragma omp parallel for schedule(static)
  for (int m = 0; m < M; ++m) {
    for (int b = 0; b < d4; ++b) {
      const uint64_t q = (uint64_t)(m_offset + m) * (uint64_t)d4 + (uint64_t)b;
      uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)idx, (uint32_t)(idx >> 32)};
      philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
      REAL e[4];
      box_muller(c[0], c[1], &e[0], &e[1]);
      box_muller(c[2], c[3], &e[2], &e[3]);
      for (int r = 0; r < 4 && 4 * b + r < d; ++r) eps[(size_t)m * d + 4 * b + r] = e[r];
    }
  }
}

/* ---- the register-blocked micro-tile of the two full-rank contractions ----
 * out[n][v] = sum_k A[k][v] * B[k][n],  v < VL rows (two SIMD vectors), n < NB columns; A and B packed, 64-byte aligned. */
typedef REAL vreal __attribute__((vector_size(32)));
#define VW ((int)(32 / sizeof(REAL)))
#define VL (2 * VW)
#define NB 6
static void micro_tile(int K, const REAL *A, const REAL *B, REAL *out) {
  vreal c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0}, c20 = {0}, c21 = {0}, c30 = {0}, c31 = {0}, c40 = {0}, c41 = {0}, c50 = {0}, c51 = {0};
  for (int k = 0; k < K; ++k) {
    const vreal a0 = *(const vreal *)(A + (size_t)k * VL), a1 = *(const vreal *)(A + (size_t)k * VL + VW);
    const REAL *b = B + (size_t)k * NB;
    c00 += a0 * b[0]; c01 += a1 * b[0];
    c10 += a0 * b[1]; c11 += a1 * b[1];
    c20 += a0 * b[2]; c21 += a1 * b[2];
    c30 += a0 * b[3]; c31 += a1 * b[3];
    c40 += a0 * b[4]; c41 += a1 * b[4];
    c50 += a0 * b[5]; c51 += a1 * b[5];
  }
  vreal *o = (vreal *)out;
  o[0] = c00; o[1] = c01; o[2] = c10; o[3] = c11; o[4] = c20; o[5] = c21;
  o[6] = c30; o[7] = c31; o[8] = c40; o[9] = c41; o[10] = c50; o[11] = c51;
}

void FN(set_threads)(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
int FN(max_threads)(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static double direct_coeff(int ent) {
  switch (ent) {
    case 0: return 1.0;   /* ClosedFormEntropy */
    case 1: return 0.0;   /* ClosedFormEntropyZeroGradient */
    case 2: return 1.0;   /* MonteCarloEntropy */
    case 3: return 0.0;   /* StickingTheLandingEntropy */
    default: return -1.0; /* StickingTheLandingEntropyZeroGradient */
  }
}

/*
 * One RepGradELBO estimate.  family 0: params = [mu; sigma], 1: [mu; vec(C) column-major].
 * eps: d x M column-major.  Target: MvNormal(t_mean, Diagonal(t_std^2)).
 * work: caller-provided scratch of at least 2*d*M REALs (Z/W and U = C^-T eps).
 * Returns the objective value (-elbo); grad fully overwritten (upper triangle zero).
 */
double FN(estimate_gradient)(int family, int d, int M, const REAL *params, const REAL *eps, const REAL *t_mean,
                             const REAL *t_std, int ent_kind, REAL *grad, REAL *work) {
  const REAL *mu = params;
  const REAL *C = params + d;  /* sigma (d) or C (d*d) */
  REAL *W = work;              /* Z, then W, d x M */
  REAL *U = work + (size_t)d * M;
  const int stl = (ent_kind == 3 || ent_kind == 4);
  double sum_ell = 0.0, sum_he = 0.0;

  /* z = scale*eps + mu   (location_scale.jl:71-87).  Full-rank: Z = tril(C) eps by row panels of VL rows; the panel of C is
   * packed [k][VL] with the entries above the diagonal zeroed, eps was packed [column block][k][NB] above. */
  REAL *epsQ = NULL, *epsP = NULL;
  const int nmb = (M + NB - 1) / NB, njb = (d + NB - 1) / NB, npan = (d + VL - 1) / VL;
  if (family == 1) {
    epsQ = (REAL *)aligned_alloc(64, (((size_t)nmb * d * NB + (size_t)njb * M * NB) * sizeof(REAL) + 63) / 64 * 64);
    epsP = epsQ + (size_t)nmb * d * NB;
#pragma omp parallel
    {
#pragma omp for schedule(static) nowait
      for (int mb = 0; mb < nmb; ++mb)          /* for the product: B[k][n] = eps[k, mb*NB + n] */
        for (int k = 0; k < d; ++k)
          for (int n = 0; n < NB; ++n) {
            const int m = mb * NB + n;
            epsQ[((size_t)mb * d + k) * NB + n] = m < M ? eps[(size_t)m * d + k] : (REAL)0;
          }
#pragma omp for schedule(static)
      for (int jb = 0; jb < njb; ++jb)          /* for the VJP: B[m][n] = eps[jb*NB + n, m] */
        for (int m = 0; m < M; ++m)
          for (int n = 0; n < NB; ++n) {
            const int j = jb * NB + n;
            epsP[((size_t)jb * M + m) * NB + n] = j < d ? eps[(size_t)m * d + j] : (REAL)0;
          }
      REAL *Ap = (REAL *)aligned_alloc(64, ((size_t)d * VL * sizeof(REAL) + 63) / 64 * 64);
      REAL out[NB * VL] __attribute__((aligned(64)));
#pragma omp for schedule(dynamic, 1)
      for (int p = npan - 1; p >= 0; --p) {
        const int i0 = p * VL, K = (i0 + VL < d) ? i0 + VL : d;
        for (int k = 0; k < K; ++k)
          for (int v = 0; v < VL; ++v) {
            const int i = i0 + v;
            Ap[(size_t)k * VL + v] = (i < d && k <= i) ? C[(size_t)k * d + i] : (REAL)0;
          }
        for (int mb = 0; mb < nmb; ++mb) {
          micro_tile(K, Ap, epsQ + (size_t)mb * d * NB, out);
          for (int n = 0; n < NB && mb * NB + n < M; ++n) {
            REAL *z = W + (size_t)(mb * NB + n) * d;
            for (int v = 0; v < VL && i0 + v < d; ++v) z[i0 + v] = mu[i0 + v] + out[n * VL + v];
          }
        }
      }
      free(Ap);
    }
  }
  /* ell; W = grad log pi(z)   (repgradelbo.jl:84-86) */
#pragma omp parallel for schedule(static) reduction(+ : sum_ell, sum_he)
  for (int m = 0; m < M; ++m) {
    const REAL *e = eps + (size_t)m * d;
    REAL *z = W + (size_t)m * d;
    if (family == 0) {
      for (int i = 0; i < d; ++i) z[i] = mu[i] + C[i] * e[i];
    }
    double ell = 0.0, he = 0.0;
    for (int i = 0; i < d; ++i) {
      const REAL u = (z[i] - t_mean[i]) / t_std[i];
      ell += -0.5 * (double)u * (double)u;
      he += 0.5 * (double)e[i] * (double)e[i];
      z[i] = -u / t_std[i];                    /* z now holds grad log pi */
    }
    sum_ell += ell;
    sum_he += he;
    if (stl) {                                 /* U = C^-T eps  (entropy.jl:59-65: -grad_z log q_stop) */
      REAL *u = U + (size_t)m * d;
      if (family == 0) {
        for (int i = 0; i < d; ++i) u[i] = e[i] / C[i];
      } else {
        for (int i = d - 1; i >= 0; --i) {     /* back substitution with C^T */
          REAL s = e[i];
          const REAL *ci = C + (size_t)i * d;
          for (int k = i + 1; k < d; ++k) s -= ci[k] * u[k];
          u[i] = s / ci[i];
        }
      }
      for (int i = 0; i < d; ++i) z[i] += u[i];
    }
  }

  double logdet = 0.0, tconst = -0.5 * d * LOG2PI;
  for (int i = 0; i < d; ++i) {
    logdet += log((double)(family == 0 ? C[i] : C[(size_t)i * d + i]));
    tconst -= log((double)t_std[i]);
  }
  const double ent = (ent_kind <= 1 ? 0.5 * d * (1.0 + LOG2PI) : sum_he / M + 0.5 * d * LOG2PI) + logdet;
  const double value = -(sum_ell / M + tconst + ent);
  const double direct = direct_coeff(ent_kind), invM = 1.0 / M;

  /* d/dmu = -(1/M) W 1: chunks of rows, the sample loop outermost inside a chunk (contiguous reads) */
#pragma omp parallel for schedule(static)
  for (int c0 = 0; c0 < d; c0 += 64) {
    double s[64];
    const int nc = (c0 + 64 < d) ? 64 : d - c0;
    for (int i = 0; i < nc; ++i) s[i] = 0.0;
    for (int m = 0; m < M; ++m) {
      const REAL *w = W + (size_t)m * d + c0;
      for (int i = 0; i < nc; ++i) s[i] += (double)w[i];
    }
    for (int i = 0; i < nc; ++i) grad[c0 + i] = (REAL)(-s[i] * invM);
  }
  if (family == 0) {
#pragma omp parallel for schedule(static)
    for (int c0 = 0; c0 < d; c0 += 64) {
      double s[64];
      const int nc = (c0 + 64 < d) ? 64 : d - c0;
      for (int i = 0; i < nc; ++i) s[i] = 0.0;
      for (int m = 0; m < M; ++m) {
        const REAL *w = W + (size_t)m * d + c0, *e = eps + (size_t)m * d + c0;
        for (int i = 0; i < nc; ++i) s[i] += (double)w[i] * (double)e[i];
      }
      for (int i = 0; i < nc; ++i) grad[d + c0 + i] = (REAL)(-s[i] * invM - direct / (double)C[c0 + i]);
    }
  } else {
    /* d/dC = -(1/M) tril(W eps') - direct diag(1/C_ii): row panels of VL rows of W packed [m][VL], column blocks of eps from the
     * [column block][m][NB] pack; only the micro-tiles that touch the lower triangle are computed, the rest of the panel's rows
     * is written as zeros */
#pragma omp parallel
    {
      REAL *Wp = (REAL *)aligned_alloc(64, ((size_t)M * VL * sizeof(REAL) + 63) / 64 * 64);
      REAL out[NB * VL] __attribute__((aligned(64)));
#pragma omp for schedule(dynamic, 1)
      for (int p = npan - 1; p >= 0; --p) {
        const int i0 = p * VL, i1 = (i0 + VL < d) ? i0 + VL : d;
        for (int m = 0; m < M; ++m)
          for (int v = 0; v < VL; ++v) Wp[(size_t)m * VL + v] = (i0 + v < d) ? W[(size_t)m * d + i0 + v] : (REAL)0;
        for (int jb = 0; jb < njb; ++jb) {
          const int j0 = jb * NB;
          if (j0 >= i1) {                                  /* wholly above the diagonal */
            for (int n = 0; n < NB && j0 + n < d; ++n)
              for (int i = i0; i < i1; ++i) grad[d + (size_t)(j0 + n) * d + i] = (REAL)0;
            continue;
          }
          micro_tile(M, Wp, epsP + (size_t)jb * M * NB, out);
          for (int n = 0; n < NB && j0 + n < d; ++n) {
            const int j = j0 + n;
            REAL *gj = grad + d + (size_t)j * d;
            for (int i = i0; i < i1; ++i) {
              if (i < j) { gj[i] = (REAL)0; continue; }
              double x = -(double)out[n * VL + (i - i0)] * invM;
              if (i == j) x -= direct / (double)C[(size_t)j * d + j];
              gj[i] = (REAL)x;
            }
          }
        }
      }
      free(Wp);
    }
  }
  free(epsQ);
  return value;
}
