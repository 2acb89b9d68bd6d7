// Inverse of one 64 x 64 diagonal block of the scale matrix by recursive doubling inside LDS (kernels_stl.hip); shared between
// the stand-alone kernel and the workgroups that ride in the sampling kernel (the inverse depends only on the parameters).
#pragma once
#include <hip/hip_runtime.h>

namespace mivi {

// DinvT[J][i * 64 + k] = (C_JJ^{-1})[k, i].  `sm` holds 3 * 64 * 65 floats.  NT threads, all of them call this.
template <int NT>
__device__ __forceinline__ void stl_dinv64_block(int d, const float *C, float *DinvT, int J, float *sm) {
  float(*L)[65] = reinterpret_cast<float(*)[65]>(sm);
  float(*Li)[65] = reinterpret_cast<float(*)[65]>(sm + 64 * 65);
  float(*T)[65] = reinterpret_cast<float(*)[65]>(sm + 2 * 64 * 65);
  const int tid = threadIdx.x;
  const float *src = C + (size_t)(64 * J) * d + 64 * J;
  for (int e = tid; e < 4096; e += NT) {
    const int r = e & 63, c = e >> 6;                    // lanes along rows: 256-byte runs of a column of C
    const float v = (r >= c) ? src[(size_t)c * d + r] : 0.f;
    L[r][c] = v;
    Li[r][c] = (r == c) ? 1.f / v : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int b = 1; b < 64; b <<= 1) {
    // [A 0; C B]^{-1} = [A^{-1} 0; -B^{-1} (C A^{-1}) B^{-1}] for every pair of b x b diagonal sub-blocks
    for (int e = tid; e < 32 * b; e += NT) {
      const int pr = e / (b * b), rem = e % (b * b), i = rem / b, j = rem % b;
      const int r0 = 2 * b * pr;
      float s = 0.f;
#pragma unroll 8
      for (int k = 0; k < b; ++k) s += L[r0 + b + i][r0 + k] * Li[r0 + k][r0 + j];   // (A^{-1} lower triangular: k < j terms are zeros)
      T[r0 + b + i][r0 + j] = s;
    }
    __syncthreads();
    for (int e = tid; e < 32 * b; e += NT) {
      const int pr = e / (b * b), rem = e % (b * b), i = rem / b, j = rem % b;
      const int r0 = 2 * b * pr;
      float s = 0.f;
#pragma unroll 8
      for (int k = 0; k < b; ++k) s += Li[r0 + b + i][r0 + b + k] * T[r0 + b + k][r0 + j];   // (B^{-1} lower triangular: k > i terms are zeros)
      Li[r0 + b + i][r0 + j] = -s;
    }
    __syncthreads();
  }
  float *dst = DinvT + (size_t)J * 4096;
  for (int e = tid; e < 4096; e += NT) {
    const int k = e & 63, i = e >> 6;
    dst[i * 64 + k] = Li[k][i];     // DinvT[i][k] = Dinv[k][i]
  }
}

}  // namespace mivi
