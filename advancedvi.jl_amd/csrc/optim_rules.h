// Elementwise optimiser rules shared by the per-launch kernels (kernels_update.hip) and the launch-free loop
// (kernels_meanfield.hip) so both paths produce bit-identical parameters.  Arithmetic is in the parameter type T,
// as Optimisers.jl does (`apply!(o::Adam, state, x::AbstractArray{T}, dx)`); explicit fma() pins the rounding.
//   Descent: x <- x - eta g                                   (Optimisers.Descent)
//   Adam   : m <- b1 m + (1-b1) g; v <- b2 v + (1-b2) g^2; x <- x - eta (m/(1-b1^t)) / (sqrt(v/(1-b2^t)) + eps)
//   ClipScale: sigma <- max(sigma, eps)                       src/optimization/clip_scale.jl:18-29
#pragma once
#include <hip/hip_runtime.h>

namespace mivi {

template <typename T>
__device__ __forceinline__ T descent_step(T x, T g, T eta) {
  return fma(-eta, g, x);
}

// beta^t by squaring (double): a cheap, deterministic function of (beta, t) shared by both code paths
__device__ __forceinline__ double ipow(double b, long long t) {
  double r = 1.0;
  while (t > 0) {
    if (t & 1) r *= b;
    b *= b;
    t >>= 1;
  }
  return r;
}

// bias corrections 1 - beta^t, evaluated once per workgroup
template <typename T>
__device__ __forceinline__ void adam_bias(long long t, double b1, double b2, T &c1, T &c2) {
  c1 = (T)(1.0 - ipow(b1, t));
  c2 = (T)(1.0 - ipow(b2, t));
}

template <typename T>
__device__ __forceinline__ T adam_step(T x, T g, T &m, T &v, T c1, T c2, T eta, T b1, T b2, T eps) {
  m = fma(b1, m, (T(1) - b1) * g);
  v = fma(b2, v, ((T(1) - b2) * g) * g);
  const T mh = m / c1, vh = v / c2;
  const T step = (eta * mh) / (sqrt(vh) + eps);
  return x - step;
}

// COCOB-Backprop (src/optimization/rules.jl:78-96): per-coordinate coin betting.  State (L, G, R, theta, x1); returns the new x.
//   L = max(L, |g|); G += |g|; R = max(R + (x - x1)(-g), 0); theta -= g;  x' = x1 + theta / (L max(G + L, alpha L)) (L + R)
// A coordinate that has never seen a non-zero gradient (L = 0: the ignored entries above the diagonal of a full-rank scale) stays
// where it is -- the reference's expression is 0/0 there.
template <typename T>
__device__ __forceinline__ T cocob_step(T x, T g, T &L, T &G, T &R, T &th, T x1, T alpha) {
  const T ag = g < T(0) ? -g : g;
  L = L > ag ? L : ag;
  G = G + ag;
  const T r = R + (x - x1) * -g;
  R = r > T(0) ? r : T(0);
  th = th + -g;
  if (!(L > T(0))) return x;
  const T den = G + L > alpha * L ? G + L : alpha * L;
  const T dx = -(x1 - x) - (th / (L * den) * (L + R));
  return x - dx;
}

template <typename T>
__device__ __forceinline__ T clip_step(T v, T eps) {
  if (v != v) return v;           // NaN propagates, like Julia's max
  return v > eps ? v : eps;
}

// ProximalLocationScaleEntropy on one scale-diagonal entry: argmin_c' -log c' + (c' - c)^2 / (2 gamma)
// = c + (sqrt(c^2 + 4 gamma) - c) / 2     (src/optimization/proximal_location_scale_entropy.jl:56)
// For c < 0 (an un-clipped DoG / DoWG step overshot: the proximal operator is what brings the entry back) the reference's expression cancels:
// in Float32 it returns exactly 0 once 4 gamma < eps c^2 -- log|det| = -Inf, the run "diverges" although the exact value gamma / |c| is a
// perfectly good positive number.  There the same quantity is evaluated without the cancellation, 2 gamma / (sqrt(c^2 + 4 gamma) - c): equal
// in exact arithmetic, positive in every precision (tests/test_gpu_optimize.py::test_prox_keeps_a_negative_diagonal_positive).
template <typename T>
__device__ __forceinline__ T prox_entropy_step(T c, T gamma) {
#pragma clang fp contract(off)   // (k_prox and the launch-free loops must round it identically)
  const T cc = c * c, g4 = T(4) * gamma;
  const T rt = sqrt(cc + g4);
  if (c < T(0)) return (T(2) * gamma) / (rt - c);
  return c + (rt - c) / T(2);
}

// PolynomialAveraging on one element (src/optimization/averaging.jl:40-47): avg <- w x + (1 - w) avg in f64, rounded to T.  Contraction off:
// the expression sits in several kernels (k_poly_average, the fused DoG apply passes, the launch-free loops) that must round it identically.
template <typename T>
__device__ __forceinline__ T poly_avg_step(T x, T avg, double wa, double wb) {
#pragma clang fp contract(off)
  const double p = wa * (double)x, q = wb * (double)avg;
  return (T)(p + q);
}

}  // namespace mivi
