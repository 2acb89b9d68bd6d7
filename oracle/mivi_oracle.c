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
 * OpenMP over samples / output columns; `mo*_set_threads` selects the thread count.
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

  /* z = scale*eps + mu   (location_scale.jl:71-87).  Full-rank: cache-blocked lower-triangular product, one
   * 64-row block of C per task reused across every sample column (what a BLAS trmm would do). */
  if (family == 1) {
    const int RB = 64;
#pragma omp parallel for schedule(dynamic, 1)
    for (int ib = (d + RB - 1) / RB - 1; ib >= 0; --ib) {
      const int i0 = ib * RB, i1 = (i0 + RB < d) ? i0 + RB : d;
      for (int m = 0; m < M; ++m) {
        REAL *z = W + (size_t)m * d;
        for (int i = i0; i < i1; ++i) z[i] = mu[i];
      }
      for (int k = 0; k < i1; ++k) {
        const REAL *ck = C + (size_t)k * d;
        const int is = k > i0 ? k : i0;
        for (int m = 0; m < M; ++m) {
          const REAL ek = eps[(size_t)m * d + k];
          REAL *z = W + (size_t)m * d;
          for (int i = is; i < i1; ++i) z[i] += ck[i] * ek;
        }
      }
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

  /* d/dmu = -(1/M) W 1 */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < d; ++i) {
    double s = 0.0;
    for (int m = 0; m < M; ++m) s += (double)W[(size_t)m * d + i];
    grad[i] = (REAL)(-s * invM);
  }
  if (family == 0) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < d; ++i) {
      double s = 0.0;
      for (int m = 0; m < M; ++m) s += (double)W[(size_t)m * d + i] * (double)eps[(size_t)m * d + i];
      grad[d + i] = (REAL)(-s * invM - direct / (double)C[i]);
    }
  } else {
    /* d/dC = -(1/M) tril(W eps') - direct diag(1/C_ii): column j accumulates rank-1 pieces */
#pragma omp parallel for schedule(dynamic, 4)
    for (int j = 0; j < d; ++j) {
      REAL *gj = grad + d + (size_t)j * d;
      for (int i = 0; i < d; ++i) gj[i] = 0;
      for (int m = 0; m < M; ++m) {
        const REAL ej = eps[(size_t)m * d + j];
        const REAL *w = W + (size_t)m * d;
        for (int i = j; i < d; ++i) gj[i] += w[i] * ej;
      }
      for (int i = j; i < d; ++i) gj[i] = (REAL)(-(double)gj[i] * invM);
      gj[j] = (REAL)((double)gj[j] - direct / (double)C[(size_t)j * d + j]);
    }
  }
  return value;
}
