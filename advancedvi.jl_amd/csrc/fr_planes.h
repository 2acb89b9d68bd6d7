// Operand PLANES of the batch engine's contractions (gfx950), round 5: every f32 operand element x is kept as the two-way f16 split of
// s * x -- hi = f16(s x), lo = f16(s x - hi): two planes, 4 bytes per element, what the f32 element itself weighs -- with a POWER-OF-TWO
// scale s that is constant along the contraction index of the run that consumes it, in MFMA-FRAGMENT ORDER.
//
// Why f16 and not bf16 (round 4: three bf16 planes, six products per block).  f16 carries 11 significant bits, so hi + lo holds 22 of the
// f32 element's 24 and THREE products -- lo.hi, hi.lo, hi.hi -- reproduce the f32 product to 2^-22 (the dropped lo.lo term); measured on the
// device (tools/ubench_f16split.hip, K = 1024, N(0,1) operands): 3.5e-7 relative l2 against the f64 product, where a plain f32 FMA chain
// has 5.9e-7.  Half the matrix-pipe work and two thirds of the LDS / HBM bytes of the bf16 scheme.  What f16 lacks is RANGE (2^-14 .. 2^15
// normal, subnormals honoured by v_mfma_f32_32x32x16_f16 -- probed): every operand is therefore scaled, exactly, by a power of two chosen
// from the data it multiplies:
//   * eps: |eps| < 6.7 (Box-Muller of a 32-bit uniform), fixed s = 2^11;
//   * a parameter-only operand (tril(C), the dense target's P, C^-T): one scale per ROW (the contraction runs along the row), from the
//     row's largest magnitude, computed by the workgroup that lays the row block out;
//   * an operand a kernel PRODUCES (W = grad log pi(Z), R = Z - m): one scale per (row, 128-sample block) resp. (sample, 128-row block) =
//     per tile of the producing workgroup, which knows the tile's maxima; the consuming product folds its chain accumulator into the
//     total at the same 128-wide boundaries, multiplying by the block's inverse scale (exact: powers of two).
// A scale puts the largest magnitude of its group into [2^13, 2^14): elements down to 2^-15 of that keep the full 22 bits, smaller ones
// lose relative -- never absolute -- precision (error <= 2^-23 of the group's maximum for every element).
//
// A fragment = 32 rows x 16 k of one operand = 2 planes x 64 lanes x 16 bytes (kFrag words): lane (row = lane % 32, h = lane / 32) holds
// the eight k slots k = 16 g + 8 (e / 4) + 4 h + e % 4.  Whoever PRODUCES an operand scales and splits it, once.
// Users: kernels_fullrank_batch.hip, kernels_targets.hip (the logistic regression's logits on planes of X).
#pragma once
#include "device_common.h"

namespace mivi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

constexpr int kFrag = 512;              // a fragment blob in 4-byte words: 2 planes x 64 lanes x 16 bytes
constexpr int kSplitProducts = 3;       // MFMAs per 32 x 32 x 16 product block (lo.hi, hi.lo, hi.hi)
constexpr float kEpsScale = 2048.f;     // 2^11: eps planes hold f16 splits of 2^11 eps
constexpr float kEpsInv = 1.f / 2048.f;

// LDS-DMA of 16 bytes per lane; the instruction's immediate offset moves BOTH addresses (global: vaddr + off, LDS: M0 + off + 16 lane):
// the two planes of a fragment are 1 KiB apart in memory and in the ring, so a stage's two pieces share one pointer and one M0
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

// The power-of-two scale of a group whose largest magnitude is amax: s amax in [2^13, 2^14); inv = 1 / s.  Magnitudes below 2^-102 (or
// zero) take the scale of 2^-102, non-finite ones the scale of 2^127 (the products then carry the non-finite value on, as f32 would).
__device__ __forceinline__ void fb_scale_of(float amax, float &s, float &inv) {
  unsigned eb = (__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu;
  eb = eb < 25u ? 25u : (eb > 254u ? 254u : eb);
  s = __builtin_bit_cast(float, (267u - eb) << 23);
  inv = __builtin_bit_cast(float, (eb - 13u) << 23);
}

// two-way f16 split of eight (already scaled) f32 values: hi = f16(x) (round to nearest even), lo = f16(x - hi) (x - hi is exact)
__device__ __forceinline__ void fb_split2(const float *x, u32x4v &uh, u32x4v &ul) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = x[2 * p], b = x[2 * p + 1];
    const _Float16 ah = (_Float16)a, bh = (_Float16)b;
    const float ra = a - (float)ah, rb = b - (float)bh;
    const f16x2 hh = {ah, bh}, ll = {(_Float16)ra, (_Float16)rb};
    uh[p] = __builtin_bit_cast(unsigned, hh);
    ul[p] = __builtin_bit_cast(unsigned, ll);
  }
}
// the (scaled) f32 value behind slot e of a lane's two plane vectors: hi + lo
__device__ __forceinline__ float fb_unsplit2(const u32x4v &uh, const u32x4v &ul, int e) {
  const unsigned wh = uh[e >> 1], wl = ul[e >> 1];   // (by value first: __builtin_bit_cast of a vector ELEMENT expression reads the vector's first word -- found on the GPU)
  const f16x2 hh = __builtin_bit_cast(f16x2, wh), ll = __builtin_bit_cast(f16x2, wl);
  return (float)hh[e & 1] + (float)ll[e & 1];
}
__device__ __forceinline__ float fb_unsplit2_word(unsigned wh, unsigned wl, int odd) {
  const f16x2 hh = __builtin_bit_cast(f16x2, wh), ll = __builtin_bit_cast(f16x2, wl);
  return (float)hh[odd] + (float)ll[odd];
}
__device__ __forceinline__ void fb_store_frag(unsigned *dst, const float *x) {   // dst: the lane's 16 bytes of plane 0
  u32x4v uh, ul;
  fb_split2(x, uh, ul);
  store16_wt(dst, uh);
  store16_wt(dst + 256, ul);
}

// ---- shared by the kernels that stage 128 x 128 tiles through an LDS ring of 16-k stages (kernels_fullrank_batch.hip, kernels_targets.hip) ----
constexpr int kStageW = 16 * 256;   // words per stage: four A + four B fragments, two planes each (16 KiB)

// A wave's operands of one 16-k group: two A fragments (its two 32-row blocks) and WJ B fragments (its 32-column blocks), two planes each
template <int WJ>
struct FbFrags {
  u32x4v A[2][2], B[WJ][2];
};
__device__ __forceinline__ f32x16 fb_mma(const u32x4v &a, const u32x4v &b, const f32x16 &c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// the 6 WJ MFMAs of a group, the 2 WJ accumulators' chains interleaved; per accumulator smallest terms first: lo.hi, hi.lo, hi.hi
template <int WJ>
__device__ __forceinline__ void fb_group(const FbFrags<WJ> &F, f32x16 (&acc)[2][WJ]) {
  constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};   // planes: 0 hi, 1 lo
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < WJ; ++j) acc[i][j] = fb_mma(F.A[i][PA[p]], F.B[j][PB[p]], acc[i][j]);
}

// One 16-k group of a 64 x 32 wave tile (WJ = 1) as ONE assembly block: the six MFMAs on the fragments Fc (already in registers) with
// the six ds_read_b128 of the NEXT group's fragments Fn issued between them -- pinned.  Left to the compiler the reads end up BEHIND the
// last MFMA (it lets Fn reuse Fc's registers), and the lgkmcnt(0) of the barrier that follows waits out the whole LDS latency with both
// waves of the SIMD in step (the ~200 cycles of "barrier + wait" per group of round 5's knock-out runs).  Early-clobber outputs keep Fn in
// registers of its own.  ab / bb: this lane's byte addresses of its first A / B fragment in ring slot 0; OFF: the slot's byte offset
// (< 64 KiB: slots 4..7 go through a second pair of base registers).  The block ENDS with lgkmcnt(0) -- the last read is two MFMAs old by
// then -- so Fn is valid for whatever the compiler does with it behind the block (it copies loop-carried registers at control-flow joins
// and cannot know of the reads: found as 3e-6 errors in one estimate of 52 when the VJP's prefetch variant used this block without the
// wait); MFMA chains alternate between the two accumulators (SrcC = vDst back to back needs no nops).
template <int OFF>
__device__ __forceinline__ void fb_group_read_asm(f32x16 &acc0, f32x16 &acc1, const FbFrags<1> &Fc, FbFrags<1> &Fn, unsigned ab, unsigned bb) {
  static_assert(OFF >= 0 && OFF + 3072 < 65536, "ds_read_b128 offset field");
  asm volatile(
      "v_mfma_f32_32x32x16_f16 %[c0], %[a0l], %[bh], %[c0]\n\t"
      "ds_read_b128 %[n0h], %[ab] offset:%[o0]\n\t"
      "ds_read_b128 %[n0l], %[ab] offset:%[o1]\n\t"
      "v_mfma_f32_32x32x16_f16 %[c1], %[a1l], %[bh], %[c1]\n\t"
      "ds_read_b128 %[n1h], %[ab] offset:%[o2]\n\t"
      "ds_read_b128 %[n1l], %[ab] offset:%[o3]\n\t"
      "v_mfma_f32_32x32x16_f16 %[c0], %[a0h], %[bl], %[c0]\n\t"
      "ds_read_b128 %[nbh], %[bb] offset:%[o0]\n\t"
      "v_mfma_f32_32x32x16_f16 %[c1], %[a1h], %[bl], %[c1]\n\t"
      "ds_read_b128 %[nbl], %[bb] offset:%[o1]\n\t"
      "v_mfma_f32_32x32x16_f16 %[c0], %[a0h], %[bh], %[c0]\n\t"
      "v_mfma_f32_32x32x16_f16 %[c1], %[a1h], %[bh], %[c1]\n\t"
      "s_waitcnt lgkmcnt(0)"   // Fn is VALID when the block ends: the compiler may move or copy it (loop-carried registers) without knowing of the reads
      : [c0] "+v"(acc0), [c1] "+v"(acc1), [n0h] "=&v"(Fn.A[0][0]), [n0l] "=&v"(Fn.A[0][1]), [n1h] "=&v"(Fn.A[1][0]), [n1l] "=&v"(Fn.A[1][1]),
        [nbh] "=&v"(Fn.B[0][0]), [nbl] "=&v"(Fn.B[0][1])
      : [a0h] "v"(Fc.A[0][0]), [a0l] "v"(Fc.A[0][1]), [a1h] "v"(Fc.A[1][0]), [a1l] "v"(Fc.A[1][1]), [bh] "v"(Fc.B[0][0]), [bl] "v"(Fc.B[0][1]),
        [ab] "v"(ab), [bb] "v"(bb), [o0] "n"(OFF), [o1] "n"(OFF + 1024), [o2] "n"(OFF + 2048), [o3] "n"(OFF + 3072)
      : "memory");
}

template <int WJ>
__device__ __forceinline__ void fb_read_frags(const unsigned *lds, int slot, int wm, int wn, int lane, FbFrags<WJ> &F) {
  const unsigned *cur = lds + slot * kStageW + 4 * lane;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int p = 0; p < 2; ++p) F.A[i][p] = *(const u32x4v *)(cur + ((2 * wm + i) * 2 + p) * 256);
#pragma unroll
  for (int j = 0; j < WJ; ++j)
#pragma unroll
    for (int p = 0; p < 2; ++p) F.B[j][p] = *(const u32x4v *)(cur + (8 + (WJ * wn + j) * 2 + p) * 256);
}

}  // namespace mivi
