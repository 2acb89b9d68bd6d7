// Operand PLANES of the full-rank contractions (gfx950): every f32 operand element as its exact three-way bf16 split (hi / mid / lo planes,
// 6 bytes per element) in MFMA-FRAGMENT ORDER.  A fragment = 32 rows x 16 k of one operand = 3 planes x 64 lanes x 16 bytes (kFrag words):
// lane (row = lane % 32, h = lane / 32) holds the eight k slots k = 16 g + 8 (e / 4) + 4 h + e % 4 -- the slot assignment of the
// second-generation kernels' mfma_bf16x3, so a product on planes is the same numbers.  Whoever PRODUCES an operand splits it, once.
// Users: kernels_fullrank_batch.hip (batches of estimates at fixed parameters), kernels_fullrank_lds.hip (k_fr_prod32p / k_fr_vjp32p: the
// device-resident optimisation loop's two launches per step).
#pragma once
#include "device_common.h"

namespace mivi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

// LDS-DMA of 16 bytes per lane; the instruction's immediate offset moves BOTH addresses (global: vaddr + off, LDS: M0 + off + 16 lane):
// the three planes of a fragment are 1 KiB apart in memory and in the ring, so a stage's three pieces share one pointer and one M0
#define FB_GLDS16(gptr, lptr, off)                                                                         \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                 \
                                   (__attribute__((address_space(3))) void *)(lptr), 16, off, 0)

// Waits as BUILTINS (the compiler's wait-count bookkeeping sees them: behind an inline-asm wait it does not know that the fragments read
// one iteration ago have arrived and puts its own lgkmcnt(0) -- which also waits for the reads just issued for the NEXT group -- in front
// of the MFMAs), the barrier itself as asm with a memory clobber (nothing moves across it).
template <int N>
__device__ __forceinline__ void fb_wait_vm() {
  __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void fb_barrier() {
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
  asm volatile("s_barrier" ::: "memory");
}

// exact three-way bf16 split of eight f32 values (kernels_fullrank_lds.hip split3_bf16: the same pieces)
__device__ __forceinline__ void fb_split3(const float *x, u32x4v &uh, u32x4v &um, u32x4v &ul) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = x[2 * p], b = x[2 * p + 1];
    const unsigned ab = __builtin_bit_cast(unsigned, a), bb = __builtin_bit_cast(unsigned, b);
    const float ra = a - __builtin_bit_cast(float, ab & 0xFFFF0000u), rb = b - __builtin_bit_cast(float, bb & 0xFFFF0000u);
    const unsigned rab = __builtin_bit_cast(unsigned, ra), rbb = __builtin_bit_cast(unsigned, rb);
    const float sa = ra - __builtin_bit_cast(float, rab & 0xFFFF0000u), sb = rb - __builtin_bit_cast(float, rbb & 0xFFFF0000u);
    uh[p] = __builtin_amdgcn_perm(bb, ab, 0x07060302u);
    um[p] = __builtin_amdgcn_perm(rbb, rab, 0x07060302u);
    ul[p] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, sb), __builtin_bit_cast(unsigned, sa), 0x07060302u);
  }
}
// the f32 value behind slot e of a lane's three plane vectors: hi + mid + lo, exact (the pieces do not overlap)
__device__ __forceinline__ float fb_unsplit(const u32x4v &uh, const u32x4v &um, const u32x4v &ul, int e) {
  const int p = e >> 1;
  const unsigned hh = (e & 1) ? (uh[p] & 0xFFFF0000u) : (uh[p] << 16), mm = (e & 1) ? (um[p] & 0xFFFF0000u) : (um[p] << 16),
                 ll = (e & 1) ? (ul[p] & 0xFFFF0000u) : (ul[p] << 16);
  return (__builtin_bit_cast(float, hh) + __builtin_bit_cast(float, mm)) + __builtin_bit_cast(float, ll);
}

// the six products of one 32 x 32 x 16 block, smallest terms first (mfma_bf16x3's order); planes as three 16-byte vectors
__device__ __forceinline__ void fb_mfma6(const u32x4v *a, const u32x4v *b, f32x16 &c) {   // a[0..2] = hi, mid, lo
  const bf16x8 ah = __builtin_bit_cast(bf16x8, a[0]), am = __builtin_bit_cast(bf16x8, a[1]), al = __builtin_bit_cast(bf16x8, a[2]);
  const bf16x8 bh = __builtin_bit_cast(bf16x8, b[0]), bm = __builtin_bit_cast(bf16x8, b[1]), bl = __builtin_bit_cast(bf16x8, b[2]);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
}

constexpr int kFrag = 768;   // a fragment blob in 4-byte words: 3 planes x 64 lanes x 16 bytes

// One 64-row x 32-column block of eps drawn (or, SRC != nullptr, read back from the f32 matrix eps[i + m ld]) by a 512-thread workgroup and
// written as operand planes in BOTH orientations: four product fragments (B operand of tril(C) eps: column = sample, k = rows) and four VJP
// fragments (B operand of W eps': row j, k = samples).  Draws: one Philox block per thread (rows 4 q .. 4 q + 3 of one column): the stream and
// the he_part partial of k_eps_m / the riders of k_fr_prod32.  E: 32 * 65 floats of LDS, red: 8 doubles.
struct PlaneEps {
  int d, M;
  uint64_t seed, idx;
  int m_offset;
  unsigned *epsP, *epsV;   // this estimate's plane sets
  double *he_part;         // [d / 64 * M / 32] or nullptr
  const float *src;        // nullptr: draw
  int ld_src;
};
__device__ __forceinline__ void plane_eps_block(const PlaneEps &a, int eb, float *E, double *red) {
  const int tid = threadIdx.x, d = a.d, nrb6 = d >> 6;
  const int R64 = eb % nrb6, c32 = eb / nrb6;
  const int q = tid & 15, c = tid >> 4;
  const int ri = R64 * 64 + 4 * q, rm = c32 * 32 + c;
  float e[4];
  if (a.src) {
    const f32x4 v = *(const f32x4 *)(a.src + (size_t)rm * a.ld_src + ri);
    e[0] = v[0]; e[1] = v[1]; e[2] = v[2]; e[3] = v[3];
  } else {
    eps_block<float>(a.seed, a.idx, (uint64_t)(a.m_offset + rm) * (uint64_t)(d >> 2) + (uint64_t)(ri >> 2), e);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) E[c * 65 + 4 * q + r] = e[r];
  const float he = 0.5f * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3]);
  const double sh = block_sum_nodrain_f32<512>(he, red);   // (its barriers also publish the tile)
  if (tid == 0 && a.he_part) a.he_part[eb] = sh;
  const int lane = tid & 63, l31 = lane & 31, h = lane >> 5, f = (tid >> 6) & 3;
  float x[8];
  unsigned *dst;
  if (tid < 256) {   // product fragment (mb32 = c32, kg = 4 R64 + f): lane = column l31, slots = rows 16 f + ..
#pragma unroll
    for (int s = 0; s < 8; ++s) x[s] = E[l31 * 65 + 16 * f + 8 * (s >> 2) + 4 * h + (s & 3)];
    dst = a.epsP + ((size_t)c32 * (d >> 4) + 4 * R64 + f) * kFrag;
  } else {           // VJP fragment (jb32 = 2 R64 + f / 2, mg = 2 c32 + f % 2): lane = row l31, slots = samples 16 (f % 2) + ..
    const int jb = f >> 1, mg = f & 1;
#pragma unroll
    for (int s = 0; s < 8; ++s) x[s] = E[(16 * mg + 8 * (s >> 2) + 4 * h + (s & 3)) * 65 + 32 * jb + l31];
    dst = a.epsV + ((size_t)(2 * R64 + jb) * (a.M >> 4) + 2 * c32 + mg) * kFrag;
  }
  u32x4v uh, um, ul;
  fb_split3(x, uh, um, ul);
  dst += 4 * lane;
  store16_wt(dst, uh);
  store16_wt(dst + 256, um);
  store16_wt(dst + 512, ul);
}

}  // namespace mivi
