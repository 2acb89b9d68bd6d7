// Parameter-only preparation of the full-rank STL solve (kernels_stl.hip): the inverses of the 64 x 64 diagonal blocks of the
// scale matrix and the off-diagonal blocks of the two half-size triangles, re-laid in the FRAGMENT ORDER of k_stl_solve64 so that
// every operand load of the chain kernel is one lane-linear 16-byte access of a contiguous 1 KiB piece (coalesced from global
// memory, conflict-free from LDS, LDS-DMA-able), and everything a chain wave multiplies with is already split into bf16 pieces.
// Shared between the stand-alone kernel and the workgroups that ride in the sampling kernel (off the critical path: nothing
// here depends on eps).
#pragma once
#include <hip/hip_runtime.h>

#include "device_common.h"

namespace mivi {

// ---- the packed operand buffer (32-bit units) -----------------------------------------------------------------------
//   [ DinvP : d/64 blocks x 6144 ]  [ solve 0 (r0 = 0) ]  [ solve 1 (r0 = d/2) ],   solve = [ crit : (NB-1) x 6144 ][ bulk : S x 4096 ]
// NB = d/128 blocks per half, S = (NB-1)(NB-2)/2.
//   plane tile (6144 per block = 4 tiles q x 1536): [q][m 2][plane 3 (hi, mid, lo)][lane 64][4 x u32 = 8 bf16]; lane (g = lane/16,
//     i = lane%16) holds the A-operand slots k = 32 m + 4 g + r (r < 4) then k = 32 m + 16 + 4 g + r of output row 16 q + i.
//     DinvP block J:  A(k, row) = (C_JJ^{-1})[k, row]                    (X_J = C_JJ^{-T} R_J)
//     crit block J:   A(k, row) = Cs[64 (J+1) + k, 64 J + row]           (the update of block J by X_{J+1}: the next pivot)
//   f32 tile (4096 per block pair = 4 tiles q x 1024): [q][u 4][lane 64][4 floats] = Cs[64 K + 4 g + 16 u + r, 64 I + 16 q + i];
//     bulk entry s <-> (K, I): K = NB-1 .. 2, I = K-2 .. 0 (the order the bulk waves consume them).
constexpr int STL_PLANE_BLOCK = 6144, STL_F32_BLOCK = 4096;
constexpr int stl_seq_first(int NB, int K) { int s = 0; for (int k = NB - 1; k > K; --k) s += k - 1; return s; }
constexpr int stl_seq_K(int NB, int s) { int K = NB - 1; while (K >= 2 && s >= K - 1) { s -= K - 1; --K; } return K; }
constexpr int stl_seq_I(int NB, int s) { int K = NB - 1; while (K >= 2 && s >= K - 1) { s -= K - 1; --K; } return K - 2 - s; }
constexpr int stl_seq_len(int NB) { return (NB - 1) * (NB - 2) / 2; }
constexpr size_t stl_solve_units(int NB) { return (size_t)(NB - 1) * STL_PLANE_BLOCK + (size_t)stl_seq_len(NB) * STL_F32_BLOCK; }
constexpr size_t stl_pack_units(int d) { return (size_t)(d / 64) * STL_PLANE_BLOCK + 2 * stl_solve_units(d / 128); }
constexpr int stl_pack_riders(int d) { return 2 * ((d / 128 - 1) + stl_seq_len(d / 128)); }   // one workgroup per off-diagonal block

// exact three-way bf16 split by truncation: v = hi + mid + lo up to 2^-24 |v| (the same pieces split3x4 makes in kernels_stl.hip)
__device__ __forceinline__ void stl_split3(float v, unsigned &hi, unsigned &mid, unsigned &lo) {
  const unsigned b0 = __builtin_bit_cast(unsigned, v);
  const float r1 = v - __builtin_bit_cast(float, b0 & 0xFFFF0000u);
  const unsigned b1 = __builtin_bit_cast(unsigned, r1);
  const float r2 = r1 - __builtin_bit_cast(float, b1 & 0xFFFF0000u);
  hi = b0 >> 16; mid = b1 >> 16; lo = __builtin_bit_cast(unsigned, r2) >> 16;
}
// one lane's 8 slots (v[0..7]) -> its 16 bytes in each of the three planes of a plane tile
__device__ __forceinline__ void stl_store_planes(unsigned *tile, int m, int lane, const float (&v)[8]) {
  unsigned h[8], mm[8], l[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) stl_split3(v[i], h[i], mm[i], l[i]);
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const u4 ph = {h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
  const u4 pm = {mm[0] | (mm[1] << 16), mm[2] | (mm[3] << 16), mm[4] | (mm[5] << 16), mm[6] | (mm[7] << 16)};
  const u4 pl = {l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16)};
  store16_wt(tile + ((m * 3 + 0) * 64 + lane) * 4, ph);   // (written through: see store16_wt -- these are rider outputs)
  store16_wt(tile + ((m * 3 + 1) * 64 + lane) * 4, pm);
  store16_wt(tile + ((m * 3 + 2) * 64 + lane) * 4, pl);
}

// Inverse of the 64 x 64 diagonal block J by recursive doubling inside LDS, written as plane tiles.  `sm` holds 3 * 64 * 65
// floats.  NT threads (a multiple of 64, <= 512), all of them call this.
template <int NT>
__device__ __forceinline__ void stl_dinv64_block(int d, const float *C, unsigned *DinvP, int J, float *sm) {
  float(*L)[65] = reinterpret_cast<float(*)[65]>(sm);
  float(*Li)[65] = reinterpret_cast<float(*)[65]>(sm + 64 * 65);
  float(*T)[65] = reinterpret_cast<float(*)[65]>(sm + 2 * 64 * 65);
  const int tid = threadIdx.x;
  const float *src = C + (size_t)(64 * J) * d + 64 * J;
  for (int e = tid; e < 1024; e += NT) {                // 16-byte loads along the rows of a column (256-byte runs): one round trip
    const int r4 = 4 * (e & 15), c = e >> 4;
    const float4 v4 = *(const float4 *)(src + (size_t)c * d + r4);
    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = r4 + t;
      const float x = (r >= c) ? v[t] : 0.f;
      L[r][c] = x;
      Li[r][c] = (r == c) ? 1.f / x : 0.f;
    }
  }
  __syncthreads();
  // [A 0; C B]^{-1} = [A^{-1} 0; -B^{-1} (C A^{-1}) B^{-1}] for every pair of b x b diagonal sub-blocks.  A thread owns a 2 x 2
  // block of outputs (half the LDS reads per FMA: the b = 32 level was bound by LDS bandwidth).
  for (int e = tid; e < 32; e += NT) {                  // b = 1
    const int r0 = 2 * e;
    const float t = L[r0 + 1][r0] * Li[r0][r0];
    Li[r0 + 1][r0] = -(Li[r0 + 1][r0 + 1] * t);
  }
  __syncthreads();
#pragma unroll
  for (int b = 2; b < 64; b <<= 1) {
    const int hb = b >> 1, nblk = hb * hb;
    for (int e = tid; e < 8 * b; e += NT) {
      const int pr = e / nblk, rem = e % nblk, i = 2 * (rem / hb), j = 2 * (rem % hb);
      const int r0 = 2 * b * pr;
      float s00 = 0.f, s01 = 0.f, s10 = 0.f, s11 = 0.f;
#pragma unroll 8
      for (int k = 0; k < b; ++k) {   // (full range, fixed trip count: the reads pipeline; A^{-1}[k][j] = 0 for k < j adds exact zeros)
        const float a0 = L[r0 + b + i][r0 + k], a1 = L[r0 + b + i + 1][r0 + k];
        const float b0 = Li[r0 + k][r0 + j], b1 = Li[r0 + k][r0 + j + 1];
        s00 += a0 * b0; s01 += a0 * b1; s10 += a1 * b0; s11 += a1 * b1;
      }
      T[r0 + b + i][r0 + j] = s00; T[r0 + b + i][r0 + j + 1] = s01;
      T[r0 + b + i + 1][r0 + j] = s10; T[r0 + b + i + 1][r0 + j + 1] = s11;
    }
    __syncthreads();
    for (int e = tid; e < 8 * b; e += NT) {
      const int pr = e / nblk, rem = e % nblk, i = 2 * (rem / hb), j = 2 * (rem % hb);
      const int r0 = 2 * b * pr;
      float s00 = 0.f, s01 = 0.f, s10 = 0.f, s11 = 0.f;
#pragma unroll 8
      for (int k = 0; k < b; ++k) {
        const float a0 = Li[r0 + b + i][r0 + b + k], a1 = Li[r0 + b + i + 1][r0 + b + k];
        const float b0 = T[r0 + b + k][r0 + j], b1 = T[r0 + b + k][r0 + j + 1];
        s00 += a0 * b0; s01 += a0 * b1; s10 += a1 * b0; s11 += a1 * b1;
      }
      Li[r0 + b + i][r0 + j] = -s00; Li[r0 + b + i][r0 + j + 1] = -s01;
      Li[r0 + b + i + 1][r0 + j] = -s10; Li[r0 + b + i + 1][r0 + j + 1] = -s11;
    }
    __syncthreads();
  }
  unsigned *dst = DinvP + (size_t)J * STL_PLANE_BLOCK;
  for (int e = tid; e < 512; e += NT) {                  // item = (q, m, lane)
    const int ln = e & 63, m = (e >> 6) & 1, q = e >> 7, g = ln >> 4, i = 16 * q + (ln & 15);
    float v[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[r] = Li[32 * m + 4 * g + r][i]; v[4 + r] = Li[32 * m + 16 + 4 * g + r][i]; }
    stl_store_planes(dst + q * 1536, m, ln, v);
  }
}

// Rider `id` < stl_pack_riders(d): one off-diagonal 64 x 64 block of one half-size triangle, re-laid (and split, for the blocks
// the chain waves consume).  NT threads, all of them call this.
template <int NT>
__device__ __forceinline__ void stl_pack_block(int d, const float *C, unsigned *pack, int id) {
  const int NB = d / 128, n = d / 2, per = (NB - 1) + stl_seq_len(NB);
  const int sv = id / per, e0 = id % per;                // solve, entry: crit blocks first, then the bulk sequence
  const float *Cs = C + (size_t)(sv * n) * d + sv * n;
  unsigned *base = pack + (size_t)(d / 64) * STL_PLANE_BLOCK + (size_t)sv * stl_solve_units(NB);
  const int tid = threadIdx.x;
  if (e0 < NB - 1) {
    const int J = e0;                                    // A(k, row) = Cs[64 (J+1) + k, 64 J + row]
    unsigned *dst = base + (size_t)J * STL_PLANE_BLOCK;
    const float *blk = Cs + (size_t)(64 * J) * d + 64 * (J + 1);
    for (int e = tid; e < 512; e += NT) {
      const int ln = e & 63, m = (e >> 6) & 1, q = e >> 7, g = ln >> 4, i = 16 * q + (ln & 15);
      const float4 a0 = *(const float4 *)(blk + (size_t)i * d + 32 * m + 4 * g), a1 = *(const float4 *)(blk + (size_t)i * d + 32 * m + 16 + 4 * g);
      const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      stl_store_planes(dst + q * 1536, m, ln, v);
    }
  } else {
    const int s = e0 - (NB - 1), K = stl_seq_K(NB, s), I = stl_seq_I(NB, s);
    float *dst = reinterpret_cast<float *>(base + (size_t)(NB - 1) * STL_PLANE_BLOCK + (size_t)s * STL_F32_BLOCK);
    const float *blk = Cs + (size_t)(64 * I) * d + 64 * K;
    for (int e = tid; e < 1024; e += NT) {               // consecutive threads along a column of C (256-byte runs)
      const int ch = e & 15, c = e >> 4, g = ch & 3, u = ch >> 2, q = c >> 4, i = c & 15;
      const float4 v = *(const float4 *)(blk + (size_t)c * d + 4 * ch);
      store16_wt(dst + ((q * 4 + u) * 64 + 16 * g + i) * 4, v);
    }
  }
}

}  // namespace mivi
