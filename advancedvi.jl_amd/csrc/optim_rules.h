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

template <typename T>
__device__ __forceinline__ T clip_step(T v, T eps) {
  if (v != v) return v;           // NaN propagates, like Julia's max
  return v > eps ? v : eps;
}

// ProximalLocationScaleEntropy on one scale-diagonal entry: argmin_c' -log c' + (c' - c)^2 / (2 gamma)
// = c + (sqrt(c^2 + 4 gamma) - c) / 2     (src/optimization/proximal_location_scale_entropy.jl:56)
template <typename T>
__device__ __forceinline__ T prox_entropy_step(T c, T gamma) {
  return c + (sqrt(c * c + T(4) * gamma) - c) / T(2);
}

}  // namespace mivi
