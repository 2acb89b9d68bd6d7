// Per-element arithmetic of the full-rank epilogues, written ONCE for every kernel generation that must agree bit for bit
// (kernels_fullrank_lds.hip: one estimate per launch; kernels_fullrank_batch.hip: a batch of estimates per launch).
// Floating-point contraction is off inside these helpers and the fused multiply-adds are explicit: what a kernel computes for an
// element must not depend on what the compiler would fuse in that kernel's surroundings.
//
// Reference semantics (AdvancedVI.jl v0.7.0):
//   diagonal-Gaussian target  ell = -1/2 sum ((z - m) / s)^2 (+ const),  grad = -((z - m) / s) / s
//   d/dC = -(1/M) tril(W eps') - direct * diag(1 / C_ii)              src/algorithms/repgradelbo.jl:142-149, src/families/location_scale.jl:52-57
#pragma once
#include "device_common.h"

namespace mivi {

// one entry of the fused diagonal-Gaussian target: z -> (ell += -u^2 / 2, w = d log pi / dz), u = (z - m) / s
__device__ __forceinline__ float diag_target_elem(float z, float tm, float tis, float &ell) {
#pragma clang fp contract(off)
  const float u = (z - tm) * tis;
  ell = __builtin_fmaf(-0.5f * u, u, ell);
  return -u * tis;
}

// one entry of the dense-Gaussian target's second product: g = -(P r), ell += r g / 2
__device__ __forceinline__ float dense_target_elem(float pr, float r, float &ell) {
#pragma clang fp contract(off)
  const float g = -pr;
  ell = __builtin_fmaf(0.5f * r, g, ell);
  return g;
}

// one entry (gi, gj) of the dense gradient d f / d C from the raw sum v = sum_m W[gi, m] eps[gj, m]:
// exact zero above the diagonal, -v / M below it (an f32 product when M is a power of two: exact, i.e. the f64 route's result
// without its conversions), the entropy estimator's direct term on the diagonal
__device__ __forceinline__ float vjp_elem(float v, int gi, int gj, bool pow2M, float invMf, double invM, double direct, float cjj) {
#pragma clang fp contract(off)
  if (gj > gi) return 0.f;
  if (pow2M && gj != gi) return -v * invMf;
  double x = -(double)v * invM;
  if (gj == gi) {
    const double q = direct / (double)cjj;
    x = x - q;
  }
  return (float)x;
}

// d f / d mu entry from the f64 row sum of W
__device__ __forceinline__ float dmu_elem(double sm, double invM) {
#pragma clang fp contract(off)
  return (float)(-sm * invM);
}

// log|det C| partial of one 32-row block: lanes 0..31 of a wave hold C_ii of the block's rows (the other 32 lanes hold 1):
// the xor tree of the 32-lane half, the same in every kernel that leaves ld_part.  Returns (sum log C_ii, #non-positive C_ii) in lane 0.
__device__ __forceinline__ void logdet_block32(float cii, float &lg, float &bad) {
  lg = logf(cii);
  bad = (cii > 0.f) ? 0.f : 1.f;
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) {
    lg += __shfl_xor(lg, o, 64);
    bad += __shfl_xor(bad, o, 64);
  }
}

}  // namespace mivi
