// Full-rank RepGradELBO contractions, second generation (gfx950): LDS-staged tiles, exact f32 products on the bf16 matrix cores.
//
// Reference semantics (AdvancedVI.jl v0.7.0), unchanged from kernels_fullrank.hip:
//   sampling   Z = scale * eps .+ mu                                  src/families/location_scale.jl:71-77
//   gradient   d/dC = -(1/M) tril(W eps') - direct * diag(1/C_ii),  d/dmu = -(1/M) W 1   
//   dense-Gaussian target  G = -P (Z - m)                             (bench target of the north star)
//
// Why a second generation: the first one gives every 32x32 output tile its own workgroup and feeds the MFMAs one dword
// per lane straight from L2 -- each operand element is fetched for exactly one MFMA (34.6 MB of L2->L1 traffic per
// contraction at d=1024, M=256 against 3.1 MB of input).  What runs now, by shape (launch_* at the end of the file):
//   * every wave stages its OWN k range of both operands through a private LDS buffer with direct-to-LDS loads
//     (global_load_lds_dwordx4) and waits on its own vmcnt: no workgroup barrier before the epilogue;
//   * products are v_mfma_f32_32x32x16_bf16 x 6 on the exact three-way bf16 split of the f32 operands (mfma_bf16x3);
//     MIVI_FR_F32MFMA=1 keeps the v_mfma_f32_32x32x2_f32 chains as a reference;
//   * k_fr_prod32 / k_fr_vjp32: 32x32 tiles, one per workgroup, for d * n_mc <= 1024 * 512 (the north star: 256 product tiles = 256
//     CUs, 528 VJP tiles, three resident per CU); the target, the ell / log-det partials and (riders) eps of the next estimate
//     and the STL solve's parameter-only preparation live in the product kernel, the optimiser step in the VJP epilogue;
//   * k_fr_prod64 / k_fr_vjp64: 64x64 tiles beyond that (four accumulators per wave, every operand element split once for two MFMA
//     tiles): 186 / 151 TF f32-equivalent at 8192 x 2048; k_fr_vjp64<STEIN> is the Stein estimator's accumulation stage;
//   * eps lives in ONE layout, eps[i + m*dP]: the sampling product reads it k-major (LDS image [n][k] with XOR-swizzled 16-byte
//     chunks, b128 fragment reads), the VJP reads the same buffer row-major.  The MFMA k-slots of both operands are permuted
//     identically (k = 8s + 4(lane>>5) + j), which leaves the sum unchanged.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "device_common.h"
#include "fr_elem.h"
#include "optim_rules.h"
#include "stl_dinv.h"
#include "fr_lds.h"

namespace mivi {

// -----------------------------------------------------------------------------------------------------------------
// Epilogue of the Stein mode (E_q[hess log pi] estimator, accumulation stage; kernels_fullrank.hip k_stein_outer is the generic
// version): tile (row0, col0) of eps G^T summed over the KW partial images, A = ((first ? 0 : A) + sum) * scale, rows of eps
// contiguous in memory.  cs_lds[KW][BN]: partial column sums of G (tiles of the first row block only): gsum (+)= and, with one
// chunk, grad = gsum / n.
// -----------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int KW, int NT>
__device__ __forceinline__ void stein_epilogue(const GemmArgs &a, const float *Cs, const float *cs_lds, bool col_tile, int row0, int col0) {
  constexpr int LDC = BM + 4;
  constexpr int NE = BM * BN / 4 / NT;
  const int tid = threadIdx.x;
#pragma unroll
  for (int u = 0; u < NE; ++u) {
    const int e = tid + u * NT, i4 = 4 * (e % (BM / 4)), n = e / (BM / 4);
    float *dst = a.st_A + (size_t)(col0 + n) * a.st_ld + row0 + i4;
    f32x4 old = {0.f, 0.f, 0.f, 0.f};
    if (!a.st_first) old = *(const f32x4 *)dst;
    f32x4 v = *(const f32x4 *)(Cs + n * LDC + i4);
#pragma unroll
    for (int k2 = 1; k2 < KW; ++k2) v += *(const f32x4 *)(Cs + (k2 * BN + n) * LDC + i4);
    const f32x4 o = (old + v) * a.st_scale;
    store16_wt(dst, o);
  }
  if (col_tile && tid < BN) {
    double sm = 0.0;
#pragma unroll
    for (int g = 0; g < KW; ++g) sm += (double)cs_lds[g * BN + tid];
    if (!a.st_first) sm += a.st_gsum[col0 + tid];
    a.st_gsum[col0 + tid] = sm;
    if (a.st_grad) a.st_grad[col0 + tid] = (float)(sm / a.st_n);
  }
}

// direct-to-LDS load of 16 bytes per lane (LDS-DMA): lane-linear 1 KiB pieces, no staging registers, no ds_write pass
#define MIVI_GLDS16(gptr, lptr)                                                                            \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                 \
                                   (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// -----------------------------------------------------------------------------------------------------------------
// f32 products on the bf16 matrix cores ("bf16x3"): every f32 operand is split EXACTLY into three bf16 pieces
// (x = hi + mid + lo: 8 + 8 + 8 significand bits, by truncation, all of one sign) and a 32 x 32 x 16 product block is the six
// v_mfma_f32_32x32x16_bf16  lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi  accumulated in f32 (smallest terms first).  The three
// dropped terms are <= 3 * 2^-24 of |x y|: f32-roundoff class (measured against the fp64 oracle: the same 1e-7 as the
// f32-MFMA chain).  Six 8-pass MFMAs per 16 k replace eight 16-pass f32 MFMAs: 2.7x less matrix-pipe time; the split costs
// ~5.5 VALU per operand element, which runs beside the other wave's MFMAs.  MIVI_FR_F32MFMA=1 selects the f32-MFMA chain.
// -----------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3_bf16(const float *x, bf16x8 &hi, bf16x8 &mid, bf16x8 &lo) {   // x[0..7]
  u32x4v uh, um, ul;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = x[2 * p], b = x[2 * p + 1];
    const unsigned ab = __builtin_bit_cast(unsigned, a), bb = __builtin_bit_cast(unsigned, b);
    const float ra = a - __builtin_bit_cast(float, ab & 0xFFFF0000u), rb = b - __builtin_bit_cast(float, bb & 0xFFFF0000u);
    const unsigned rab = __builtin_bit_cast(unsigned, ra), rbb = __builtin_bit_cast(unsigned, rb);
    const float sa = ra - __builtin_bit_cast(float, rab & 0xFFFF0000u), sb = rb - __builtin_bit_cast(float, rbb & 0xFFFF0000u);
    uh[p] = __builtin_amdgcn_perm(bb, ab, 0x07060302u);    // {b.hi16, a.hi16}
    um[p] = __builtin_amdgcn_perm(rbb, rab, 0x07060302u);
    ul[p] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, sb), __builtin_bit_cast(unsigned, sa), 0x07060302u);
  }
  hi = __builtin_bit_cast(bf16x8, uh);
  mid = __builtin_bit_cast(bf16x8, um);
  lo = __builtin_bit_cast(bf16x8, ul);
}

// acc += A(32 x 16) B(16 x 32) from the eight f32 fragment values av[0..7], bv[0..7] of this lane (same k slots in both)
__device__ __forceinline__ void mfma_bf16x3(const float *av, const float *bv, f32x16 &acc) {
  bf16x8 ah, am, al, bh, bm, bl;
  split3_bf16(av, ah, am, al);
  split3_bf16(bv, bh, bm, bl);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

// -----------------------------------------------------------------------------------------------------------------
// k_fr_vjp32: tril(W eps^T), one 32 x 32 tile per workgroup, the four waves split K = n_mc into contiguous quarters.
//   With a 32 x 32 tile no operand element is shared between waves (they differ only in k), so every wave stages its OWN
//   k range: LDS-DMA into a private 8 KiB buffer (32 k of W rows and of eps rows), wait on its own vmcnt, fragments to
//   registers, the buffer re-requested at once, MFMA chain -- no workgroup barrier until the epilogue, the waves of
//   a SIMD (2-3 workgroups are resident per CU: 48 KiB of LDS each) drift apart and fill each other's load latency.
//   All 528 tiles of the north star are resident at once: no second dispatch round for the tiles beyond 2 x 256.
// -----------------------------------------------------------------------------------------------------------------
template <bool FUSED, bool BF3>
__device__ __forceinline__ void fr_vjp32_body(const GemmArgs &a) {
  constexpr int BM = 32, BN = 32, KW = 4, NT = 256, SUB = 32, RING = 1, NV = SUB / 2;   // (SUB 16, RING 3 measured slower: 7.4 vs 6.6 us)
  constexpr int LDC = BM + 4;
  constexpr int WAVE_F = RING * 2 * SUB * 32;               // floats per wave: RING x {As[16 k][32] + Bs[16 k][32]}
  constexpr int EPI = KW * BN * LDC + (NT / BM) * BM;
  // LDS is padded to ~52 KiB so that at most THREE workgroups share a CU: with the 32 KiB the kernel needs the dispatcher packs
  // up to five onto some CUs and leaves others with one, and the packed CUs finish last
  constexpr int MAIN = 13 * 1024;
  static_assert(MAIN >= KW * WAVE_F && MAIN >= EPI, "LDS budget");
  __shared__ __attribute__((aligned(16))) float lds[MAIN + 2 * KW + 4];
  float *adam_cc = lds + MAIN + 2 * KW;
  {
    const unsigned long long pA = (unsigned long long)a.A, pB = (unsigned long long)a.B, pW = (unsigned long long)a.work,
                             pP = (unsigned long long)a.params, pD = (unsigned long long)a.dbg, pG = (unsigned long long)a.out.grad,
                             pQ = (unsigned long long)a.out.partials;
    asm volatile("" ::"s"(pA), "s"(pB), "s"(pW), "s"(pP), "s"(pD), "s"(pG), "s"(pQ), "s"(a.d), "s"(a.M), "s"(a.lda), "s"(a.ldb),
                 "s"(a.n_work), "s"(a.knock), "s"(a.out.partials_mode), "s"(a.out.ent_kind), "s"(a.out.M_total));
  }
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int d = a.d;
  if ((int)blockIdx.x == a.n_work) {   // objective value of THIS estimate (everything it sums is older)
    const float *pp = a.params;
    finalize_value_block<float, NT, false>(d, a.self_vin, a.self_out, (int64_t)d + (int64_t)d * d,
                                           [pp, d](int i) { return pp[d + (size_t)i * d + i]; }, reinterpret_cast<double *>(lds));   // (scratch: the idle staging area)
    return;
  }
  MIVI_STAMP_K(a.dbg, G_VJP, 0);
  const __attribute__((address_space(4))) int *wp = (const __attribute__((address_space(4))) int *)a.work + 4 * blockIdx.x;
  const int4 wk = make_int4(wp[0], wp[1], wp[2], wp[3]);
  asm volatile("" ::"s"(wk.x), "s"(wk.y), "s"(wk.z), "s"(wk.w));
  const int rb = wk.x & 0xffff, cb = wk.x >> 16;
  const int row0 = rb * BM, col0 = cb * BN;
  const bool mu_tile = (wk.w & 2);
  const int Kq = a.M / KW, nsub = Kq / SUB;   // this wave's k range: [w * Kq, (w + 1) * Kq), in sub-stages of 16
  MIVI_DEV_ONLY(if (a.dbg && tid == 0 && blockIdx.x < 4096) a.dbg[((size_t)G_VJP * 4096 + blockIdx.x) * 8 + 7] = wk.x;)

  if (FUSED && a.upd.rule == 1 && tid == 0)
    adam_bias<float>(a.upd.t_base + (a.upd.t_ptr ? *a.upd.t_ptr : 0), a.upd.b1, a.upd.b2, adam_cc[0], adam_cc[1]);

  // a 1 KiB piece = 8 k of 32 rows; lane -> (k = lane / 8, rows 4 (lane % 8) ..)
  // ring of RING sub-stages of 16 k per wave: {As[16 k][32 rows], Bs[16 k][32 cols]} = 4 KiB each, 4 LDS-DMA pieces
  float *buf = lds + w * WAVE_F;
  const float *Ag = a.A + row0 + 4 * (lane & 7) + (size_t)(w * Kq + (lane >> 3)) * a.lda;
  const float *Bg = a.B + col0 + 4 * (lane & 7) + (size_t)(w * Kq + (lane >> 3)) * a.ldb;
  auto issue = [&](int t) {
    const float *pa = Ag + (size_t)(t * SUB) * a.lda, *pb = Bg + (size_t)(t * SUB) * a.ldb;
    float *dst = buf + (t % RING) * (2 * SUB * 32);
#pragma unroll
    for (int p = 0; p < SUB / 8; ++p) {
      MIVI_GLDS16(pa + (size_t)(8 * p) * a.lda, dst + p * 256);
      MIVI_GLDS16(pb + (size_t)(8 * p) * a.ldb, dst + SUB * 32 + p * 256);
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float rsum = 0.f;
  // a wave that is requesting operands outranks the waves already inside their MFMA chains (which only need an issue slot
  // now and then): without this the youngest workgroup of a CU gets its first data after the older ones are done
  __builtin_amdgcn_s_setprio(3);
  if (!MIVI_KNOCKED(a, 4))
    for (int t = 0; t < RING && t < nsub; ++t) issue(t);
  // workgroups beyond two per CU (dispatch order = block order) arrive last and are the youngest waves of their SIMD: age
  // arbitration would give them the matrix pipe only after the older two are done -- let them go first instead
  if (blockIdx.x >= 512) __builtin_amdgcn_s_setprio(2);
  else __builtin_amdgcn_s_setprio(0);
  for (int t = 0; t < nsub; ++t) {
    const int behind = nsub - 1 - t < RING - 1 ? nsub - 1 - t : RING - 1;   // sub-stages that may stay in flight
    if (behind >= 2) wait_vmcnt<2 * (SUB / 4)>();
    else if (behind == 1) wait_vmcnt<SUB / 4>();
    else wait_vmcnt<0>();
    const float *cur = buf + (t % RING) * (2 * SUB * 32);
    float av[NV], bv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {   // the lane's k slots of this sub-stage: k = 8 (i / 4) + 4 h + (i % 4), both operands
      const int k = 8 * (i >> 2) + 4 * h + (i & 3);
      av[i] = cur[k * 32 + l31];
      bv[i] = cur[SUB * 32 + k * 32 + l31];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this ring slot is free again before it is requested anew
    if (t == 0) MIVI_STAMP_K(a.dbg, G_VJP, 1);
    if (t + RING < nsub && !MIVI_KNOCKED(a, 4)) issue(t + RING);
    if (mu_tile) {
#pragma unroll
      for (int i = 0; i < NV; ++i) rsum += av[i];
    }
    if (!MIVI_KNOCKED(a, 2)) {
      if (BF3) {
#pragma unroll
        for (int g = 0; g < NV / 8; ++g) mfma_bf16x3(av + 8 * g, bv + 8 * g, acc);
      } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc, 0, 0, 0);
      }
    }
  }
  __builtin_amdgcn_s_barrier();   // every wave is done with its buffer: LDS becomes the epilogue image
  MIVI_STAMP_K(a.dbg, G_VJP, 2);
  if (MIVI_KNOCKED(a, 16)) return;
  float *Cs = lds;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
    *(f32x4 *)(Cs + (w * BN + l31) * LDC + 8 * q + 4 * h) = v;
  }
  float *rs_lds = lds + KW * BN * LDC;   // [NT/BM = 8][BM]: wave w, half h -> slot 2 w + h
  if (mu_tile) rs_lds[(2 * w + h) * BM + l31] = rsum;
  lds_barrier();
  MIVI_STAMP_K(a.dbg, G_VJP, 3);
  if (MIVI_KNOCKED(a, 32)) return;
  vjp_epilogue<BM, BN, KW, NT, FUSED>(a, Cs, rs_lds, adam_cc, wk, row0, col0);
}

struct GemmMulti { GemmArgs lane[4]; };   // (kMaxLanes, see k_fr_prod32m)
template <bool FUSED, bool BF3>
__global__ __launch_bounds__(256) void k_fr_vjp32(GemmArgs a) { fr_vjp32_body<FUSED, BF3>(a); }
template <bool FUSED, bool BF3>
__global__ __launch_bounds__(256) void k_fr_vjp32m(GemmMulti m) { fr_vjp32_body<FUSED, BF3>(m.lane[blockIdx.y]); }

// -----------------------------------------------------------------------------------------------------------------
// k_fr_vjp32s: the VJP of lane-batched estimates on STRIPS -- a workgroup takes up to NS consecutive 32 x 32 tiles of ONE block row
// (same W rows, eps rows cb0, cb0 + 1, ..) instead of one tile: the W fragments of a wave's K quarter are loaded once, split into
// their bf16 pieces once and stay in registers (48 VGPRs) for the whole strip; per tile a wave stages only its eps fragments (the next
// tile's are requested before this tile's MFMA chain), so the launch boundary, the first-operand latency and half of the split VALU
// are paid once per strip.  n_mc = 256 only (two sub-stages per wave).  BIT-IDENTICAL to k_fr_vjp32: same K quarters, same MFMA
// chains, same epilogue (vjp_epilogue on the same four partial images).  The wait before a tile's fragments are read leaves exactly
// the previous epilogue's two stores in flight (VMEM operations of a wave complete in issue order on gfx9).
// blockIdx.y = lane, blockIdx.x -> strips[]: {rb | cb0 << 16, tiles, 0, 0}.
// -----------------------------------------------------------------------------------------------------------------
struct StripMulti {
  GemmArgs lane[4];   // (kMaxLanes)
  const int4 *strips;
};
__global__ __launch_bounds__(256) void k_fr_vjp32s(StripMulti m) {
  constexpr int BM = 32, BN = 32, KW = 4, NT = 256, SUB = 32, LDC = BM + 4;
  constexpr int WAVE_F = 2 * SUB * 32;                       // the wave's K quarter of one operand: two sub-stages [32 k][32]
  constexpr int STAGE = KW * WAVE_F;
  constexpr int EPI = KW * BN * LDC + (NT / BM) * BM;
  constexpr int MAIN = 13 * 1024;                            // (k_fr_vjp32's footprint: at most three workgroups per CU)
  static_assert(MAIN >= STAGE + EPI, "LDS budget");
  __shared__ __attribute__((aligned(16))) float lds[MAIN + 4];
  const GemmArgs &a = m.lane[blockIdx.y];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((int)blockIdx.x == a.n_work) {   // objective value of THIS estimate (everything it sums is older), as in k_fr_vjp32
    const float *pp = a.params;
    const int dd = a.d;
    finalize_value_block<float, NT, false>(dd, a.self_vin, a.self_out, (int64_t)dd + (int64_t)dd * dd,
                                           [pp, dd](int i) { return pp[dd + (size_t)i * dd + i]; }, reinterpret_cast<double *>(lds));
    return;
  }
  const __attribute__((address_space(4))) int *wp = (const __attribute__((address_space(4))) int *)m.strips + 4 * blockIdx.x;
  const int sx = wp[0], nc = MIVI_KNOCKED(a, 1024) ? (wp[1] ? 1 : 0) : wp[1];
  if (nc == 0 || MIVI_KNOCKED(a, 2048)) return;
  const int rb = sx & 0xffff, cb0 = sx >> 16;
  const int row0 = rb * BM;
  const int Kq = a.M / KW;   // = 2 SUB
  float *buf = lds + w * WAVE_F;
  float *Cs = lds + STAGE;
  float *rs_lds = Cs + KW * BN * LDC;
  auto issue = [&](const float *base, int ld, int c0) {   // rows/columns c0..c0+31, k in the wave's quarter: 8 pieces of 8 k x 32
    const float *p0 = base + c0 + 4 * (lane & 7) + (size_t)(w * Kq + (lane >> 3)) * ld;
    if (MIVI_KNOCKED(a, 4)) return;
#pragma unroll
    for (int p = 0; p < 8; ++p) MIVI_GLDS16(p0 + (size_t)(8 * p) * ld, buf + p * 256);
  };
  __builtin_amdgcn_s_setprio(3);
  issue(a.A, a.lda, row0);
  __builtin_amdgcn_s_setprio(0);
  wait_vmcnt<0>();
  float av[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) av[i] = buf[((i >> 4) * SUB + 8 * ((i & 15) >> 2) + 4 * h + (i & 3)) * 32 + l31];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  issue(a.B, a.ldb, cb0 * BN);
  float rsum = 0.f;
  if (cb0 + nc - 1 == rb) {   // the strip ends on the diagonal tile: d/dmu partial row sums, k_fr_vjp32's order
#pragma unroll
    for (int i = 0; i < 32; ++i) rsum += av[i];
  }
  bf16x8 Ah[4], Am[4], Al[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) split3_bf16(av + 8 * g, Ah[g], Am[g], Al[g]);
  for (int jt = 0; jt < nc; ++jt) {
    const int cb = cb0 + jt;
    if (jt == 0 || a.out.partials_mode) wait_vmcnt<0>();
    else wait_vmcnt<2>();   // the tile's fragments are in; the previous epilogue's two stores may still be on their way
    float bv[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) bv[i] = buf[((i >> 4) * SUB + 8 * ((i & 15) >> 2) + 4 * h + (i & 3)) * 32 + l31];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (jt + 1 < nc) issue(a.B, a.ldb, (cb + 1) * BN);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (!MIVI_KNOCKED(a, 2))
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x8 bh, bm, bl;
      split3_bf16(bv + 8 * g, bh, bm, bl);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al[g], bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[g], bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am[g], bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am[g], bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[g], bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[g], bh, acc, 0, 0, 0);
    }
    if (MIVI_KNOCKED(a, 16)) { if (acc[0] == 123.f) Cs[tid] = acc[3]; continue; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
      *(f32x4 *)(Cs + (w * BN + l31) * LDC + 8 * q + 4 * h) = v;
    }
    const bool diag = cb == rb;
    if (diag) rs_lds[(2 * w + h) * BM + l31] = rsum;
    lds_barrier();
    if (!MIVI_KNOCKED(a, 32)) vjp_epilogue<BM, BN, KW, NT, false>(a, Cs, rs_lds, nullptr, make_int4(rb | (cb << 16), 0, 0, diag ? 3 : 0), row0, cb * BN);
    lds_barrier();   // the images are free for the next tile
  }
}

// -----------------------------------------------------------------------------------------------------------------
// k_fr_vjp64: tril(W eps^T) on 64 x 64 tiles, eight waves split K = n_mc into contiguous runs of 32-k sub-stages; every wave
// stages its own 64 rows of W and 64 rows of eps (16 KiB, wave-private, LDS-DMA) and holds all four 32 x 32 accumulators of
// the tile, so an operand element feeds two MFMA tiles instead of one and is split into bf16 pieces once for both.
//   Used for the large shapes (n_mc >= 1024: every wave has four or more sub-stages, so its one-deep staging overlaps with
//   its own MFMA chains): 4096 x 1024 VJP 108 -> 118 TF f32-equivalent.  At the north star the 32 x 32 kernel stays faster
//   (6.0 vs 7.4 us) although its most loaded CUs carry 192 KiB against 128 KiB here -- see launch_lds_vjp.
// Epilogue: vjp_epilogue<64, 64, 8, 512> (the eight partial tiles summed in wave order through LDS).
// -----------------------------------------------------------------------------------------------------------------
template <bool FUSED, bool BF3, bool STEIN = false>
__global__ __launch_bounds__(512) void k_fr_vjp64(GemmArgs a) {
  constexpr int BM = 64, BN = 64, KW = 8, NT = 512, SUB = 32;
  constexpr int LDC = BM + 4;
  constexpr int WAVE_F = 2 * SUB * 64;                      // floats per wave: As[32 k][64 rows] + Bs[32 k][64 rows]
  constexpr int EPI = KW * BN * LDC + (NT / BM) * BM;
  constexpr int MAIN = EPI > KW * WAVE_F ? EPI : KW * WAVE_F;
  __shared__ __attribute__((aligned(16))) float lds[MAIN + 2 * KW + 4];
  float *adam_cc = lds + MAIN + 2 * KW;
  {
    const unsigned long long pA = (unsigned long long)a.A, pB = (unsigned long long)a.B, pW = (unsigned long long)a.work,
                             pP = (unsigned long long)a.params, pD = (unsigned long long)a.dbg, pG = (unsigned long long)a.out.grad,
                             pQ = (unsigned long long)a.out.partials;
    asm volatile("" ::"s"(pA), "s"(pB), "s"(pW), "s"(pP), "s"(pD), "s"(pG), "s"(pQ), "s"(a.d), "s"(a.M), "s"(a.lda), "s"(a.ldb),
                 "s"(a.n_work), "s"(a.knock), "s"(a.out.partials_mode), "s"(a.out.ent_kind), "s"(a.out.M_total));
  }
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int d = a.d;
  if ((int)blockIdx.x == a.n_work) {   // objective value of THIS estimate (everything it sums is older)
    const float *pp = a.params;
    finalize_value_block<float, NT, false>(d, a.self_vin, a.self_out, (int64_t)d + (int64_t)d * d,
                                           [pp, d](int i) { return pp[d + (size_t)i * d + i]; }, reinterpret_cast<double *>(lds));   // (scratch: the idle staging area)
    if (STEIN && a.st_logpi && tid == 0)   // (partials mode: thread 0 just wrote sum_m ell_m)
      *a.st_logpi = (float)((double)((const float *)a.self_out.partials)[a.self_out.scalars_off] / a.st_n);
    return;
  }
  MIVI_STAMP_K(a.dbg, G_VJP, 0);
  const __attribute__((address_space(4))) int *wp = (const __attribute__((address_space(4))) int *)a.work + 4 * blockIdx.x;
  const int4 wk = make_int4(wp[0], wp[1], wp[2], wp[3]);
  asm volatile("" ::"s"(wk.x), "s"(wk.y), "s"(wk.z), "s"(wk.w));
  const int rb = wk.x & 0xffff, cb = wk.x >> 16;
  const int row0 = rb * BM, col0 = cb * BN;
  const bool mu_tile = !STEIN && (wk.w & 2);
  const bool col_tile = STEIN && rb == 0;   // column sums of the B operand (G)
  const int nsub = a.M / SUB;
  const int t_beg = (w * nsub) / KW, t_end = ((w + 1) * nsub) / KW;   // this wave's run of sub-stages
  MIVI_DEV_ONLY(if (a.dbg && tid == 0 && blockIdx.x < 4096) a.dbg[((size_t)G_VJP * 4096 + blockIdx.x) * 8 + 7] = wk.x;)

  if (FUSED && a.upd.rule == 1 && tid == 0)
    adam_bias<float>(a.upd.t_base + (a.upd.t_ptr ? *a.upd.t_ptr : 0), a.upd.b1, a.upd.b2, adam_cc[0], adam_cc[1]);

  // a 1 KiB piece = 8 k of 32 rows (lane -> k = lane / 8, rows 4 (lane % 8) ..); image [k][64 rows]: piece (kq, rh) at
  // k = 8 kq .., rows 32 rh .. -> float offset (8 kq + lane / 8) * 64 + 32 rh + 4 (lane % 8): NOT lane-linear inside a k row of
  // 64, so the two row halves of a k group are kept as two separate [8 k][32] blocks: offset ((kq * 2 + rh) * 8 + k) * 32 + row
  float *buf = lds + w * WAVE_F;
  const float *Ag = a.A + row0 + 4 * (lane & 7) + (size_t)(lane >> 3) * a.lda;
  const float *Bg = a.B + col0 + 4 * (lane & 7) + (size_t)(lane >> 3) * a.ldb;
  auto issue = [&](int t) {
    const float *pa = Ag + (size_t)(t * SUB) * a.lda, *pb = Bg + (size_t)(t * SUB) * a.ldb;
#pragma unroll
    for (int kq = 0; kq < 4; ++kq)
#pragma unroll
      for (int rh = 0; rh < 2; ++rh) {
        MIVI_GLDS16(pa + (size_t)(8 * kq) * a.lda + 32 * rh, buf + (kq * 2 + rh) * 256);
        MIVI_GLDS16(pb + (size_t)(8 * kq) * a.ldb + 32 * rh, buf + SUB * 64 + (kq * 2 + rh) * 256);
      }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float rsum[2] = {0.f, 0.f};   // row sums of A (d/dmu tiles) / column sums of B (Stein mode)
  __builtin_amdgcn_s_setprio(3);
  if (t_beg < t_end && !MIVI_KNOCKED(a, 4)) issue(t_beg);
  __builtin_amdgcn_s_setprio(0);
  for (int t = t_beg; t < t_end; ++t) {
    wait_vmcnt<0>();
    float av[2][16], bv[2][16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {   // the lane's k slots of this sub-stage: k = 8 (i / 4) + 4 h + (i % 4), both operands
      const int kq = i >> 2, kk = 4 * h + (i & 3);
#pragma unroll
      for (int rh = 0; rh < 2; ++rh) {
        av[rh][i] = buf[((kq * 2 + rh) * 8 + kk) * 32 + l31];
        bv[rh][i] = buf[SUB * 64 + ((kq * 2 + rh) * 8 + kk) * 32 + l31];
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the buffer is free again before it is requested anew
    if (t == t_beg) MIVI_STAMP_K(a.dbg, G_VJP, 1);
    if (t + 1 < t_end && !MIVI_KNOCKED(a, 4)) issue(t + 1);
    if (mu_tile) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { rsum[0] += av[0][i]; rsum[1] += av[1][i]; }
    }
    if (col_tile) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { rsum[0] += bv[0][i]; rsum[1] += bv[1][i]; }
    }
    if (!MIVI_KNOCKED(a, 2)) {
      if (BF3) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {   // two K = 16 groups per sub-stage; every operand half is split ONCE for its two tiles
          bf16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
          for (int x = 0; x < 2; ++x) {
            split3_bf16(av[x] + 8 * g, ah[x], am[x], al[x]);
            split3_bf16(bv[x] + 8 * g, bh[x], bm[x], bl[x]);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              f32x16 c = acc[i][j];
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm[j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[j], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], c, 0, 0, 0);
              acc[i][j] = c;
            }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
      }
    }
  }
  __builtin_amdgcn_s_barrier();   // every wave is done with its buffer: LDS becomes the epilogue image
  MIVI_STAMP_K(a.dbg, G_VJP, 2);
  if (MIVI_KNOCKED(a, 16)) return;
  float *Cs = lds;   // Cs[kw][n (64 eps rows = tile columns)][LDC]: rows of the tile contiguous
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        *(f32x4 *)(Cs + (w * BN + 32 * j + l31) * LDC + 32 * i + 8 * q + 4 * h) = v;
      }
  float *rs_lds = lds + KW * BN * LDC;   // [NT/BM = 8 waves][BM]
  if (mu_tile || col_tile) {
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const float o = rsum[x] + __shfl_xor(rsum[x], 32, 64);   // the two k halves of the wave
      if (h == 0) rs_lds[w * BM + 32 * x + l31] = o;
    }
  }
  lds_barrier();
  MIVI_STAMP_K(a.dbg, G_VJP, 3);
  if (MIVI_KNOCKED(a, 32)) return;
  if (STEIN) stein_epilogue<BM, BN, KW, NT>(a, Cs, rs_lds, col_tile, row0, col0);
  else vjp_epilogue<BM, BN, KW, NT, FUSED>(a, Cs, rs_lds, adam_cc, wk, row0, col0);
}

// -----------------------------------------------------------------------------------------------------------------
// k_fr_prod32: Z = mu + tril(C) eps (G_SAMPLE) or G = -P (Z - m) (G_DENSE) WITHOUT split-K: one 32 x 32 output tile per
// workgroup, eight waves split the tile's k range into contiguous runs of 32-k sub-stages and stage their OWN operands
// (private 8 KiB LDS buffer per wave, LDS-DMA, no workgroup barrier before the epilogue -- the structure of k_fr_vjp32).
// The epilogue sums the eight partial tiles in wave order through LDS and applies mu / the fused target right there:
//   R_DIAG: W = grad log pi(z), ell partial      R_DENSE_R: R = z - m (k-major operand of the dense product)
//   R_DENSE_G: W = g, ell += r g / 2              R_PLAIN: Z = z
// so no slab, no reduce kernel and no second kernel boundary sits between the draw and the VJP.  The heaviest tile (K = d)
// bounds the kernel (d/32 sub-stages on one CU); shapes where that is too long take the 64 x 64 kernel (k_fr_prod64).
// Trailing workgroups draw eps of the next estimate; the first column block's workgroups leave the log-det partials.
// -----------------------------------------------------------------------------------------------------------------

template <int MODE, bool BF3>
__device__ __forceinline__ void fr_prod32_body(const Prod32Args &a) {
  constexpr int NW = 8, NT = 512, SUB = 32, LDC = 36;
  // ONE sub-stage buffer per wave (64 KiB per workgroup): with two (128 KiB, the wave's next sub-stage in flight under its MFMAs) a product
  // workgroup owned its CU; with one, two of them -- or one and a VJP workgroup of another chain -- share it, and the other seven waves cover a
  // wave's load latency anyway: alone 7.2 -> 6.85 us, isolated 20-estimate batches 13.2 -> 12.4 (-> 10.8 with four chains), steady 9.9 -> 8.1 us
  constexpr int RING = 1;
  constexpr int WAVE_F = RING * 2 * SUB * 32;   // RING sub-stage buffers per wave: {As[32 k][32], Bs[32 cols][32 k]} x RING
  constexpr int EPI = NW * 32 * LDC;
  constexpr int MAIN = (NW * WAVE_F > EPI) ? NW * WAVE_F : EPI;
  __shared__ __attribute__((aligned(16))) float lds[MAIN + 2 * NW];
  double *red = reinterpret_cast<double *>(lds + MAIN);
  {
    const unsigned long long p0 = (unsigned long long)a.A, p1 = (unsigned long long)a.B, p2 = (unsigned long long)a.params,
                             p3 = (unsigned long long)a.t_mean, p4 = (unsigned long long)a.t_istd, p5 = (unsigned long long)a.Z,
                             p6 = (unsigned long long)a.W, p7 = (unsigned long long)a.R, p8 = (unsigned long long)a.ell_part,
                             p9 = (unsigned long long)a.ld_part, p10 = (unsigned long long)a.dbg;
    asm volatile("" ::"s"(p0), "s"(p1), "s"(p2), "s"(p3), "s"(p4), "s"(p5), "s"(p6), "s"(p7), "s"(p8), "s"(p9), "s"(p10), "s"(a.d),
                 "s"(a.dP), "s"(a.mode), "s"(a.lda), "s"(a.n_tiles), "s"(a.ncb), "s"(a.knock));
  }
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int d = a.d;
  // Block -> tile.  Tiles are listed heaviest first (t = 0 ..); the n_dinv lightest ones (the END of that list) are dispatched FIRST
  // and invert one 64 x 64 diagonal block of C each before their own tile (a 6.6 us latency chain + a 1-2 us tile against the
  // 8 us of the heaviest tile): with workgroups of their own the inversions took 16 CUs for most of the kernel, 16 tiles had
  // to wait for a CU, and the riders hitched to those tiles finished last (+2.4 us).
  const bool trailing = (int)blockIdx.x >= a.n_tiles;
  const int bid = trailing ? (int)blockIdx.x : ((int)blockIdx.x < a.n_dinv ? a.n_tiles - 1 - (int)blockIdx.x : (int)blockIdx.x - a.n_dinv);
  if (!trailing && (int)blockIdx.x < a.n_dinv) {
    stl_dinv64_block<NT>(d, a.A, a.stl_pack, (int)blockIdx.x, lds);
    lds_barrier();   // (LDS reuse only: the inverse's write-through stores drain behind the tile)
    MIVI_STAMP_K(a.dbg, MODE, 6);
  }
  // Riders (work that is off this kernel's critical path): the re-lay of the STL operands (n_pack blocks), then eps(t+1)
  // (n_eps blocks).  Rider r hitches onto the (n_dinv + r)-th lightest tile's workgroup AFTER that tile's epilogue: the light
  // tiles are done after 1-3 us of a ~8 us kernel.  Riders beyond the number of tiles get trailing workgroups.
  auto rider = [&](int r) {
    if (r < a.n_pack) {
      stl_pack_block<NT>(d, a.A, a.stl_pack, r);
      return;
    }
    // eps(t+1): one Philox block per thread (rows gi..gi+3 of column gm), the k_eps stream
    const SampleArgs<float> &n = a.next_eps;
    const int eb = r - a.n_pack, nrb6 = d >> 6;
    const int gi = (eb % nrb6) * 64 + 4 * (tid & 15), gm = (eb / nrb6) * 32 + (tid >> 4);
    float e[4];
    eps_block<float>(n.rng.seed, rng_index(n.rng), (uint64_t)(n.rng.m_offset + gm) * (uint64_t)(d >> 2) + (uint64_t)(gi >> 2), e);
    const f32x4 ev = {e[0], e[1], e[2], e[3]};
    store16_wt(n.eps + (size_t)gm * n.ld_eps + gi, ev);   // (written through: see store16_wt)
    const float he = 0.5f * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3]);
    const double sh = block_sum_nodrain_f32<NT>(he, red);
    if (tid == 0) n.he_part[eb] = sh;
  };
  if (trailing) {   // riders beyond the tile count (a lane-batched launch is as wide as its widest lane: nothing to do beyond this lane's own)
    if (bid - a.n_dinv < a.n_pack + a.n_eps) rider(bid - a.n_dinv);
    return;
  }
  MIVI_STAMP_K(a.dbg, MODE, 0);
  // heaviest row blocks first (they bound the kernel)
  const int nrb = d >> 5;
  // tile <- block: workgroup b runs on XCD b % 8 (each XCD has its own L2).  XCD x = (xr, xc) = (x & 3, x >> 2) owns the row
  // blocks of class rb % 4 == xr and the column blocks of class cb % 2 == xc: an A row panel comes over the fabric twice and a
  // B column panel four times (8 MB at the north star) instead of A eight times (17 MB with cb = b % 8).
  int rb, cb;
  if (!MIVI_KNOCKED(a, 8) && (nrb & 3) == 0 && (a.ncb & 1) == 0) {
    const int x = bid & 7, j = bid >> 3, hc = a.ncb >> 1;
    rb = nrb - 1 - (4 * (j / hc) + (x & 3));
    cb = 2 * (j % hc) + (x >> 2);
  } else {
    rb = nrb - 1 - bid / a.ncb;
    cb = bid % a.ncb;
  }
  const int row0 = rb * 32, col0 = cb * 32;
  const int nst = (MODE == G_SAMPLE) ? rb + 1 : nrb;            // 32-k sub-stages of this tile
  // this wave's run: chunks of c = ceil(nst / 8) sub-stages (the longest run is what the even split had; the last waves may stay idle).
  // The boundaries are multiples of c, and c is the same for the row blocks 2 k and 2 k + 1: the batch kernels
  // (kernels_fullrank_batch.hip), whose waves walk the runs of two row blocks one after the other, fold both at the same places.
  const int rc = (nst + NW - 1) / NW;
  const int t_beg = w * rc < nst ? w * rc : nst, t_end = (w + 1) * rc < nst ? (w + 1) * rc : nst;

  // epilogue operands that do not depend on the product: requested now, consumed after the MFMA chain
  const int ei4 = 4 * (tid & 7), en = (tid >> 3) & 31;   // threads 0..255: rows ei4..+3 of column en
  const int gi = row0 + ei4, gm = col0 + en;
  f32x4 mu = {0.f, 0.f, 0.f, 0.f}, tm = mu, tis = mu, rr = mu;
  float cii = 1.f;
  const bool ld_blk = a.ld_part && cb == 0 && tid < 32;
  if (tid < 256) {
    if (a.mode == R_DENSE_G) {
      rr = *(const f32x4 *)(a.B + (size_t)gm * a.dP + gi);
    } else {
      mu = *(const f32x4 *)(a.params + gi);
      if (a.mode == R_DIAG || a.mode == R_DENSE_R) tm = *(const f32x4 *)(a.t_mean + gi);
      if (a.mode == R_DIAG) tis = *(const f32x4 *)(a.t_istd + gi);
    }
    if (ld_blk) cii = a.params[d + (size_t)(row0 + tid) * d + row0 + tid];
  }

  // staging: a 1 KiB piece of A = 8 k x 32 rows (lane -> k = lane / 8, rows 4 (lane % 8)); a piece of B = 8 columns x 32 k,
  // lane -> column lane / 8, 16-byte chunk (lane % 8) ^ swizzle(column) of that column's 128 contiguous bytes
  float *buf = lds + w * WAVE_F;
  const float *Ag = a.A + row0 + 4 * (lane & 7) + (size_t)(lane >> 3) * a.lda;
  const float *Bg[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int n = 8 * p + (lane >> 3);
    Bg[p] = a.B + (size_t)(col0 + n) * a.dP + 4 * ((lane & 7) ^ ((n >> 1) & 7));
  }
  auto issue = [&](int t, int slot) {
    const float *pa = Ag + (size_t)(t * SUB) * a.lda;
    float *dst = buf + slot * (2 * SUB * 32);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      MIVI_GLDS16(pa + (size_t)(8 * p) * a.lda, dst + p * 256);
      MIVI_GLDS16(Bg[p] + t * SUB, dst + SUB * 32 + p * 256);
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int b_off = SUB * 32 + l31 * 32;            // this lane's column of the B image
  const int b_swz = h ^ ((l31 >> 1) & 7);
  __builtin_amdgcn_s_setprio(3);
  if (t_beg < t_end) issue(t_beg, 0);
  if (RING == 2 && t_beg + 1 < t_end) issue(t_beg + 1, 1);
  __builtin_amdgcn_s_setprio(0);
  for (int t = t_beg; t < t_end; ++t) {
    const int slot = RING == 2 ? (t - t_beg) & 1 : 0;
    if (RING == 2 && t + 1 < t_end) wait_vmcnt<8>();   // the sub-stage behind this one (8 pieces) may stay in flight
    else wait_vmcnt<0>();
    const float *cur = buf + slot * (2 * SUB * 32);
    float av[16];
    f32x4 bq[4];
#pragma unroll
    for (int s8 = 0; s8 < 4; ++s8) {   // MFMA step (s8, j) uses k = 8 s8 + 4 h + j on both operands
      bq[s8] = *(const f32x4 *)(cur + b_off + 4 * ((2 * s8) ^ b_swz));
#pragma unroll
      for (int j = 0; j < 4; ++j) av[4 * s8 + j] = cur[(8 * s8 + 4 * h + j) * 32 + l31];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this buffer is free again before it is requested anew
    if (t == t_beg) MIVI_STAMP_K(a.dbg, MODE, 1);
    if (t + RING < t_end) issue(t + RING, slot);
    if (MODE == G_SAMPLE && t == rb) {   // the diagonal block of tril(C): keep k <= i
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (8 * (i >> 2) + 4 * h + (i & 3) > l31) av[i] = 0.f;
    }
    if (!MIVI_KNOCKED(a, 2)) {
      if (BF3) {
        float bv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) bv[i] = bq[i >> 2][i & 3];
        mfma_bf16x3(av, bv, acc);
        mfma_bf16x3(av + 8, bv + 8, acc);
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bq[i >> 2][i & 3], acc, 0, 0, 0);
      }
    }
  }
  __builtin_amdgcn_s_barrier();   // every wave is done with its buffer: LDS becomes the epilogue image
  MIVI_STAMP_K(a.dbg, MODE, 2);
  float *Cs = lds;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
    *(f32x4 *)(Cs + (w * 32 + l31) * LDC + 8 * q + 4 * h) = v;
  }
  lds_barrier();
  MIVI_STAMP_K(a.dbg, MODE, 3);
  float ell = 0.f;
  if (tid < 256) {
    f32x4 v = *(const f32x4 *)(Cs + en * LDC + ei4);
#pragma unroll
    for (int k2 = 1; k2 < NW; ++k2) v += *(const f32x4 *)(Cs + (k2 * 32 + en) * LDC + ei4);   // fixed order
    if (a.mode == R_DENSE_G) {
      f32x4 g;
#pragma unroll
      for (int c = 0; c < 4; ++c) g[c] = dense_target_elem(v[c], rr[c], ell);
      store16_wt(a.W + (size_t)gm * d + gi, g);
    } else {
      const f32x4 z = mu + v;
      if (a.Z) store16_wt(a.Z + (size_t)gm * d + gi, z);
      if (a.mode == R_DIAG) {
        f32x4 wv;
#pragma unroll
        for (int c = 0; c < 4; ++c) wv[c] = diag_target_elem(z[c], tm[c], tis[c], ell);   // (fr_elem.h: shared with the batch kernels)
        store16_wt(a.W + (size_t)gm * d + gi, wv);
      } else if (a.mode == R_DENSE_R) {
        const f32x4 rz = z - tm;
        store16_wt(a.R + (size_t)gm * a.dP + gi, rz);
      }
    }
  }
  if (a.mode == R_DIAG || a.mode == R_DENSE_G) {
    const double sl = block_sum_nodrain_f32<NT>(ell, red);
    if (tid == 0) a.ell_part[bid] = sl;
  }
  if (ld_blk) {   // log|det C| partial of this 32-row block (lanes 0..31 of wave 0)
    float lg = logf(cii), bad = (cii > 0.f) ? 0.f : 1.f;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      lg += __shfl_xor(lg, o, 64);
      bad += __shfl_xor(bad, o, 64);
    }
    if (tid == 0) {
      a.ld_part[rb] = (double)lg;
      a.ld_part[nrb + rb] = (double)bad;
    }
  }
  MIVI_STAMP_K(a.dbg, MODE, 4);
  {
    const int r = a.n_tiles - 1 - bid - a.n_dinv;   // this tile's rider, if any (the n_dinv lightest tiles carried an inversion)
    if (r >= 0 && r < a.n_pack + a.n_eps) {
      lds_barrier();                     // (the riders reuse `red` / LDS; no drain of this tile's stores: see block_sum_nodrain)
      rider(r);
      MIVI_STAMP_K(a.dbg, MODE, 5);
    }
  }
}

// one estimate per launch, and LANES: the same tiles for up to four independent estimates at once (blockIdx.y = lane; every lane its own
// operands and outputs: the contexts of mivi_estimate_gradient_n's interleaved estimates, api_batch.hip)
constexpr int kMaxLanes = 4;
struct Prod32Multi { Prod32Args lane[kMaxLanes]; };
template <int MODE, bool BF3>
__global__ __launch_bounds__(512) void k_fr_prod32(Prod32Args a) { fr_prod32_body<MODE, BF3>(a); }
template <int MODE, bool BF3>
__global__ __launch_bounds__(512) void k_fr_prod32m(Prod32Multi m) { fr_prod32_body<MODE, BF3>(m.lane[blockIdx.y]); }

// -----------------------------------------------------------------------------------------------------------------
// k_fr_prod32q: the sampling product of FOUR lanes (estimates at the same parameters: one tril(C), four eps) as 2 x 2 work per
// workgroup -- the tile PAIR {(nrb-1-p, cb), (p, cb)} (K = 32 (nrb + 1) between them: every workgroup carries the same work, where
// k_fr_prod32's heaviest tile carries 32 x its lightest) for the lane PAIR {2 lp, 2 lp + 1} (every fragment of C is staged once and
// split into its bf16 pieces once for two estimates).  (d/64) (M/32) 2 workgroups -- 256 at the north star, one per CU, one round --
// instead of four lanes x 256 tiles whose fixed costs (boundary, first operands, epilogue) were paid 1024 times.
// BIT-IDENTICAL to k_fr_prod32: a tile's k range is cut into the same eight runs (wave w: run w of both tiles), every run is the same
// MFMA chain, the epilogue adds the eight partial tiles in the same order and the ell partial lands in the same slot of ell_part.
// L2 -> LDS bytes per four estimates 135 -> 101 MB, split VALU -25 %.  The eps(t+1) riders (two per workgroup) follow the epilogues.
// -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_fr_prod32q(Prod32Multi m) {
  constexpr int NW = 8, NT = 512, SUB = 32, LDC = 36;
  constexpr int WAVE_F = 3 * SUB * 32;        // per wave: As[32 k][32 rows], Bs of lane 0 [32 cols][32 k], Bs of lane 1
  constexpr int EPI = 2 * NW * 32 * LDC;      // sixteen partial tiles (two tiles x eight runs) of ONE lane
  constexpr int MAIN = (NW * WAVE_F > EPI) ? NW * WAVE_F : EPI;
  __shared__ __attribute__((aligned(16))) float lds[MAIN + 2 * NW + 192];
  double *red = reinterpret_cast<double *>(lds + MAIN);
  float *lds_e = lds + MAIN + 2 * NW;   // the epilogue's row vectors {mu, target mean, target 1/std} x {tile 0, tile 1} x 32 rows
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int d = m.lane[0].d, dP = m.lane[0].dP, lda = m.lane[0].lda, ncb = m.lane[0].ncb, mode = m.lane[0].mode;
  const float *A = m.lane[0].A, *params = m.lane[0].params;
  const int nrb = d >> 5, nq4 = ncb >> 1;     // (2 ncb (column block, lane pair) combinations, four classes)
  // workgroup -> (pair, column block, lane pair): XCD x = b % 8 = (x & 1, x >> 1) owns the tile pairs of class p % 2 and the combinations of
  // class q % 4: a row panel of C crosses the fabric four times, an eps column panel twice
  const int b = blockIdx.x, x = b & 7, j = b >> 3;
  const int pp = 2 * (j / nq4) + (x & 1), q = 4 * (j % nq4) + (x >> 1);
  const int cb = q % ncb, lp = q / ncb;
  const Prod32Args &a0 = m.lane[2 * lp], &a1 = m.lane[2 * lp + 1];
  const int rbT[2] = {nrb - 1 - pp, pp};
  const int col0 = cb * 32;
  int t_beg[2], t_end[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {   // (k_fr_prod32's runs: chunks of ceil(nst / 8) sub-stages)
    const int nst = rbT[t] + 1, rc = (nst + NW - 1) / NW;
    t_beg[t] = w * rc < nst ? w * rc : nst;
    t_end[t] = (w + 1) * rc < nst ? (w + 1) * rc : nst;
  }
  // epilogue operands that do not depend on the product (threads 0..255: tile 0, 256..511: tile 1; rows ei4..+3 of column en)
  const int half = tid >> 8, t2 = tid & 255;
  const int ei4 = 4 * (t2 & 7), en = (t2 >> 3) & 31;
  const int rowE = 32 * (half ? rbT[1] : rbT[0]);
  const int gi = rowE + ei4, gm = col0 + en;
  // (staged through LDS by three waves' LDS-DMA: held in registers across the main loop they cost every thread twelve VGPRs, and two waves
  //  of this kernel + one of the VJP kernel -- or of the spinning exchange kernel -- have to fit a SIMD's 512)
  if (w < 3 && (w == 0 || mode == R_DIAG || (w == 1 && mode == R_DENSE_R))) {
    const float *src = w == 0 ? params : (w == 1 ? m.lane[0].t_mean : m.lane[0].t_istd);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 32 * (lane < 32 ? rbT[0] : rbT[1]) + l31),
                                     (__attribute__((address_space(3))) void *)(lds_e + w * 64), 4, 0, 0);
  }
  const bool ld_blk = a0.ld_part && cb == 0 && t2 < 32;
  float cii = 1.f;
  if (ld_blk) cii = params[d + (size_t)(rowE + t2) * d + rowE + t2];

  float *buf = lds + w * WAVE_F;
  const float *Ag = A + 4 * (lane & 7) + (size_t)(lane >> 3) * lda;
  // eps fragments: piece p holds columns n = 8 p + lane / 8, 16-byte chunk (lane % 8) ^ ((n >> 1) & 7) of the column's 128 bytes: the swizzle
  // repeats every two pieces, so two per-lane bases (even / odd pieces) + uniform offsets address all of them, for either lane
  const float *Bg0[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int n = 8 * p + (lane >> 3);
    Bg0[p] = a0.B + (size_t)(col0 + n) * dP + 4 * ((lane & 7) ^ ((n >> 1) & 7));
  }
  const long long dB = a1.B - a0.B;
  auto issue = [&](int tile, int t) {   // sub-stage t (32 k) of tile `tile`: the C fragment and both lanes' eps fragments
    const float *pa = Ag + 32 * rbT[tile] + (size_t)(t * SUB) * lda;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      MIVI_GLDS16(pa + (size_t)(8 * p) * lda, buf + p * 256);
      const float *pb = Bg0[p & 1] + (size_t)(16 * (p >> 1)) * dP + t * SUB;
      MIVI_GLDS16(pb, buf + SUB * 32 + p * 256);
      MIVI_GLDS16(pb + dB, buf + 2 * SUB * 32 + p * 256);
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int l = 0; l < 2; ++l)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][l][r] = 0.f;
  const int b_off = SUB * 32 + l31 * 32;
  const int b_swz = h ^ ((l31 >> 1) & 7);
  __builtin_amdgcn_s_setprio(3);
  if (MIVI_KNOCKED(m.lane[0], 4)) {
  } else if (t_beg[0] < t_end[0]) issue(0, t_beg[0]);
  else if (t_beg[1] < t_end[1]) issue(1, t_beg[1]);
  __builtin_amdgcn_s_setprio(0);
#pragma unroll
  for (int tile = 0; tile < 2; ++tile) {
    for (int t = t_beg[tile]; t < t_end[tile]; ++t) {
      wait_vmcnt<0>();
      float av[16];
      f32x4 bq[2][4];
#pragma unroll
      for (int s8 = 0; s8 < 4; ++s8) {
        bq[0][s8] = *(const f32x4 *)(buf + b_off + 4 * ((2 * s8) ^ b_swz));
        bq[1][s8] = *(const f32x4 *)(buf + SUB * 32 + b_off + 4 * ((2 * s8) ^ b_swz));
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) av[4 * s8 + jj] = buf[(8 * s8 + 4 * h + jj) * 32 + l31];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the buffer is free again before it is requested anew
      if (MIVI_KNOCKED(m.lane[0], 4)) {
      } else if (t + 1 < t_end[tile]) issue(tile, t + 1);
      else if (tile == 0 && t_beg[1] < t_end[1]) issue(1, t_beg[1]);
      if (t == rbT[tile]) {   // the diagonal block of tril(C): keep k <= i
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (8 * (i >> 2) + 4 * h + (i & 3) > l31) av[i] = 0.f;
      }
      if (MIVI_KNOCKED(m.lane[0], 2)) continue;
      bf16x8 ah[2], am[2], al[2];
      split3_bf16(av, ah[0], am[0], al[0]);
      split3_bf16(av + 8, ah[1], am[1], al[1]);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        bf16x8 bh[2], bm[2], bl[2];
#pragma unroll
        for (int l = 0; l < 2; ++l) {
          float bv[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) bv[i] = bq[l][2 * g + (i >> 2)][i & 3];
          split3_bf16(bv, bh[l], bm[l], bl[l]);
        }
        // (the two lanes' chains interleaved: a dependent MFMA issues behind an independent one)
        f32x16 c0 = acc[tile][0], c1 = acc[tile][1];
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[g], bh[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[g], bh[1], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g], bl[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g], bl[1], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[g], bm[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[g], bm[1], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[g], bh[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[g], bh[1], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g], bm[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g], bm[1], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g], bh[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[g], bh[1], c1, 0, 0, 0);
        acc[tile][0] = c0;
        acc[tile][1] = c1;
      }
    }
  }
  wait_vmcnt<0>();                // (the row vectors' DMA, had this wave no sub-stage to wait for)
  __builtin_amdgcn_s_barrier();   // every wave is done with its buffer: LDS becomes the epilogue image
  if (MIVI_KNOCKED(m.lane[0], 16)) return;
  // ell_part slot of a tile = its workgroup index in k_fr_prod32's own (XCD-aware) order
  const int rE = nrb - 1 - (half ? rbT[1] : rbT[0]);
  const int bidE = ((rE & 3) + 4 * (cb & 1)) + 8 * ((rE >> 2) * (ncb >> 1) + (cb >> 1));
  float *Cs = lds;
  f32x4 mu = *(const f32x4 *)(lds_e + 32 * half + ei4), tm = {0.f, 0.f, 0.f, 0.f}, tis = tm;
  if (mode == R_DIAG || mode == R_DENSE_R) tm = *(const f32x4 *)(lds_e + 64 + 32 * half + ei4);
  if (mode == R_DIAG) tis = *(const f32x4 *)(lds_e + 128 + 32 * half + ei4);
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const Prod32Args &a = l ? a1 : a0;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        f32x4 v = {acc[t][l][4 * qq], acc[t][l][4 * qq + 1], acc[t][l][4 * qq + 2], acc[t][l][4 * qq + 3]};
        *(f32x4 *)(Cs + ((t * NW + w) * 32 + l31) * LDC + 8 * qq + 4 * h) = v;
      }
    lds_barrier();
    float ell = 0.f;
    {
      const float *Ct = Cs + half * NW * 32 * LDC;
      f32x4 v = *(const f32x4 *)(Ct + en * LDC + ei4);
#pragma unroll
      for (int k2 = 1; k2 < NW; ++k2) v += *(const f32x4 *)(Ct + (k2 * 32 + en) * LDC + ei4);   // fixed order
      const f32x4 z = mu + v;
      if (a.Z) store16_wt(a.Z + (size_t)gm * d + gi, z);
      if (mode == R_DIAG) {
        f32x4 wv;
#pragma unroll
        for (int c = 0; c < 4; ++c) wv[c] = diag_target_elem(z[c], tm[c], tis[c], ell);
        store16_wt(a.W + (size_t)gm * d + gi, wv);
      } else if (mode == R_DENSE_R) {
        const f32x4 rz = z - tm;
        store16_wt(a.R + (size_t)gm * dP + gi, rz);
      }
    }
    if (mode == R_DIAG) {   // the two tiles' ell partials: waves 0..3 / 4..7, each the tree of k_fr_prod32's block sum
      const double sv = (double)wave_sum_f32(ell);
      lds_barrier();
      if (lane == 0) red[w] = sv;
      lds_barrier();
      if (t2 == 0) {
        double s = red[4 * half];
#pragma unroll
        for (int i = 1; i < 4; ++i) s += red[4 * half + i];
#pragma unroll
        for (int i = 0; i < 4; ++i) s += 0.0;   // (k_fr_prod32 adds its four idle waves' zeros: -0.0 would become +0.0 there)
        a.ell_part[bidE] = s;
      }
    } else {
      lds_barrier();
    }
  }
  if (ld_blk) {   // log|det C| partial of this 32-row block (lanes 0..31 of waves 0 and 4), the same for both lanes
    float lg = logf(cii), bad = (cii > 0.f) ? 0.f : 1.f;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      lg += __shfl_xor(lg, o, 64);
      bad += __shfl_xor(bad, o, 64);
    }
    if (t2 == 0) {
      const int rb = half ? rbT[1] : rbT[0];
      a0.ld_part[rb] = (double)lg;
      a0.ld_part[nrb + rb] = (double)bad;
      if (a1.ld_part) {
        a1.ld_part[rb] = (double)lg;
        a1.ld_part[nrb + rb] = (double)bad;
      }
    }
  }
  // riders: eps(t+1) of both lanes, block eb = this workgroup's index among its lane pair's (d/64) (M/32) workgroups
  const int eb = pp * ncb + cb;
#pragma unroll
  for (int l = 0; l < 2; ++l) {
    const Prod32Args &a = l ? a1 : a0;
    if (eb >= a.n_eps || MIVI_KNOCKED(m.lane[0], 64)) continue;
    const SampleArgs<float> &n = a.next_eps;
    const int nrb6 = d >> 6;
    const int ri = (eb % nrb6) * 64 + 4 * (tid & 15), rm = (eb / nrb6) * 32 + (tid >> 4);
    float e[4];
    eps_block<float>(n.rng.seed, rng_index(n.rng), (uint64_t)(n.rng.m_offset + rm) * (uint64_t)(d >> 2) + (uint64_t)(ri >> 2), e);
    const f32x4 ev = {e[0], e[1], e[2], e[3]};
    store16_wt(n.eps + (size_t)rm * n.ld_eps + ri, ev);
    const float he = 0.5f * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3]);
    const double sh = block_sum_nodrain_f32<NT>(he, red);
    if (tid == 0) n.he_part[eb] = sh;
  }
}

// -----------------------------------------------------------------------------------------------------------------
// k_fr_prod64: the unsplit product on 64 x 64 tiles for the LARGE shapes (more tiles than CUs): Z = mu + tril(C) eps (G_SAMPLE) or
// G = -P (Z - m) (G_DENSE) with the target fused into the epilogue, the structure of k_fr_vjp64 -- eight waves split the tile's k
// range into contiguous runs of 32-k sub-stages, every wave stages its OWN 64 rows of A ([k][64 rows], the VJP kernel's image) and
// 64 columns of B (k-major source: [column][32 k] with the 16-byte chunks XOR-swizzled, k_fr_prod32's image), 16 KiB per wave, holds
// the tile's four 32 x 32 accumulators and splits every operand element into bf16 pieces once for two MFMA tiles; no workgroup
// barrier before the epilogue, no partial slabs, no reduce kernel (the split-K route of round 2 reached 87 TF f32-equivalent at
// 8192 x 2048 against the 152 TF of the VJP kernel with this structure).  Tiles heaviest first (K = 64 (rb + 1) for the triangular product).
// Trailing workgroups draw eps of the next estimate.
// -----------------------------------------------------------------------------------------------------------------
template <int MODE, bool BF3>
__global__ __launch_bounds__(512) void k_fr_prod64(Prod32Args a) {
  constexpr int BM = 64, BN = 64, KW = 8, NT = 512, SUB = 32;
  constexpr int LDC = BM + 4;
  constexpr int WAVE_F = 2 * SUB * 64;                      // floats per wave: As[32 k][64 rows] + Bs[64 cols][32 k]
  constexpr int EPI = KW * BN * LDC;
  constexpr int MAIN = EPI > KW * WAVE_F ? EPI : KW * WAVE_F;
  __shared__ __attribute__((aligned(16))) float lds[MAIN + 2 * KW];
  double *red = reinterpret_cast<double *>(lds + MAIN);
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int d = a.d;
  if ((int)blockIdx.x >= a.n_tiles) {   // eps(t+1): one Philox block per thread (rows gi..gi+3 of column gm), the k_eps stream
    const SampleArgs<float> &n = a.next_eps;
    const int eb = (int)blockIdx.x - a.n_tiles, nrb6 = d >> 6;
    const int gi = (eb % nrb6) * 64 + 4 * (tid & 15), gm = (eb / nrb6) * 32 + (tid >> 4);
    float e[4];
    eps_block<float>(n.rng.seed, rng_index(n.rng), (uint64_t)(n.rng.m_offset + gm) * (uint64_t)(d >> 2) + (uint64_t)(gi >> 2), e);
    const f32x4 ev = {e[0], e[1], e[2], e[3]};
    store16_wt(n.eps + (size_t)gm * n.ld_eps + gi, ev);
    const float he = 0.5f * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3]);
    const double sh = block_sum_nodrain_f32<NT>(he, red);
    if (tid == 0) n.he_part[eb] = sh;
    return;
  }
  const int bid = blockIdx.x;
  const int nrb = d >> 6;
  int rb, cb;   // block -> tile: heavy row blocks first; XCD x = b % 8 = (x & 3, x >> 2) owns row class rb % 4 and column class cb % 2
  if ((nrb & 3) == 0 && (a.ncb & 1) == 0) {
    const int x = bid & 7, j = bid >> 3, hc = a.ncb >> 1;
    rb = nrb - 1 - (4 * (j / hc) + (x & 3));
    cb = 2 * (j % hc) + (x >> 2);
  } else {
    rb = nrb - 1 - bid / a.ncb;
    cb = bid % a.ncb;
  }
  const int row0 = rb * BM, col0 = cb * BN;
  const int nst = (MODE == G_SAMPLE) ? 2 * (rb + 1) : (d >> 5);     // 32-k sub-stages of this tile
  const int t_beg = (w * nst) / KW, t_end = ((w + 1) * nst) / KW;   // this wave's run

  // A piece = 8 k of 32 rows (lane -> k = lane / 8, rows 4 (lane % 8) ..), kept as [8 k][32] blocks (kq, rh) like k_fr_vjp64;
  // B piece = 8 columns x 32 k (lane -> column lane / 8, 16-byte chunk (lane % 8) ^ swizzle(column) of its 128 contiguous bytes)
  float *buf = lds + w * WAVE_F;
  const float *Ag = a.A + row0 + 4 * (lane & 7) + (size_t)(lane >> 3) * a.lda;
  const float *Bg[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int n = 8 * p + (lane >> 3);
    Bg[p] = a.B + (size_t)(col0 + n) * a.dP + 4 * ((lane & 7) ^ ((n >> 1) & 7));
  }
  auto issue = [&](int t) {
    const float *pa = Ag + (size_t)(t * SUB) * a.lda;
#pragma unroll
    for (int kq = 0; kq < 4; ++kq)
#pragma unroll
      for (int rh = 0; rh < 2; ++rh) MIVI_GLDS16(pa + (size_t)(8 * kq) * a.lda + 32 * rh, buf + (kq * 2 + rh) * 256);
#pragma unroll
    for (int p = 0; p < 8; ++p) MIVI_GLDS16(Bg[p] + t * SUB, buf + SUB * 64 + p * 256);
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // epilogue operands that do not depend on the product: requested now
  const int ei4 = 4 * (tid & 15), en = tid >> 4;   // rows ei4..+3 of columns en and en + 32
  const int gi = row0 + ei4;
  f32x4 mu = {0.f, 0.f, 0.f, 0.f}, tm = mu, tis = mu, rr[2] = {mu, mu};
  float cii = 1.f;
  const bool ld_blk = a.ld_part && cb == 0 && tid < 64;
  if (a.mode == R_DENSE_G) {
#pragma unroll
    for (int u = 0; u < 2; ++u) rr[u] = *(const f32x4 *)(a.B + (size_t)(col0 + en + 32 * u) * a.dP + gi);
  } else {
    mu = *(const f32x4 *)(a.params + gi);
    if (a.mode == R_DIAG || a.mode == R_DENSE_R) tm = *(const f32x4 *)(a.t_mean + gi);
    if (a.mode == R_DIAG) tis = *(const f32x4 *)(a.t_istd + gi);
  }
  if (ld_blk) cii = a.params[d + (size_t)(row0 + tid) * d + row0 + tid];
  __builtin_amdgcn_s_setprio(3);
  if (t_beg < t_end) issue(t_beg);
  __builtin_amdgcn_s_setprio(0);
  for (int t = t_beg; t < t_end; ++t) {
    wait_vmcnt<0>();
    float av[2][16], bv[2][16];
#pragma unroll
    for (int s8 = 0; s8 < 4; ++s8) {   // MFMA step (s8, jj) uses k = 8 s8 + 4 h + jj on both operands
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int n = 32 * x + l31;
        const f32x4 bq = *(const f32x4 *)(buf + SUB * 64 + n * 32 + 4 * ((2 * s8 + h) ^ ((n >> 1) & 7)));
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          bv[x][4 * s8 + jj] = bq[jj];
          av[x][4 * s8 + jj] = buf[((s8 * 2 + x) * 8 + 4 * h + jj) * 32 + l31];
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the buffer is free again before it is requested anew
    if (t + 1 < t_end) issue(t + 1);
    if (MODE == G_SAMPLE && t >= 2 * rb) {   // the diagonal block of tril(C): keep k <= i
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (32 * t + 8 * (i >> 2) + 4 * h + (i & 3) > row0 + 32 * x + l31) av[x][i] = 0.f;
    }
    if (BF3) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {   // two K = 16 groups per sub-stage; every operand half is split ONCE for its two tiles
        bf16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          split3_bf16(av[x] + 8 * g, ah[x], am[x], al[x]);
          split3_bf16(bv[x] + 8 * g, bh[x], bm[x], bl[x]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            f32x16 c = acc[i][j];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm[j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh[j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm[j], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], c, 0, 0, 0);
            acc[i][j] = c;
          }
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
    }
  }
  __builtin_amdgcn_s_barrier();   // every wave is done with its buffer: LDS becomes the epilogue image
  float *Cs = lds;   // Cs[kw][n (64 columns)][LDC]: rows of the tile contiguous
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        *(f32x4 *)(Cs + (w * BN + 32 * j + l31) * LDC + 32 * i + 8 * q + 4 * h) = v;
      }
  lds_barrier();
  float ell = 0.f;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int n = en + 32 * u, gm = col0 + n;
    f32x4 v = *(const f32x4 *)(Cs + n * LDC + ei4);
#pragma unroll
    for (int k2 = 1; k2 < KW; ++k2) v += *(const f32x4 *)(Cs + (k2 * BN + n) * LDC + ei4);   // fixed order
    if (a.mode == R_DENSE_G) {
      f32x4 g;
#pragma unroll
      for (int c = 0; c < 4; ++c) g[c] = dense_target_elem(v[c], rr[u][c], ell);
      store16_wt(a.W + (size_t)gm * d + gi, g);
    } else {
      const f32x4 z = mu + v;
      if (a.Z) store16_wt(a.Z + (size_t)gm * d + gi, z);
      if (a.mode == R_DIAG) {
        f32x4 wv;
#pragma unroll
        for (int c = 0; c < 4; ++c) wv[c] = diag_target_elem(z[c], tm[c], tis[c], ell);
        store16_wt(a.W + (size_t)gm * d + gi, wv);
      } else if (a.mode == R_DENSE_R) {
        const f32x4 rz = z - tm;
        store16_wt(a.R + (size_t)gm * a.dP + gi, rz);
      }
    }
  }
  if (a.mode == R_DIAG || a.mode == R_DENSE_G) {
    const double sl = block_sum_nodrain_f32<NT>(ell, red);
    if (tid == 0) a.ell_part[bid] = sl;
  }
  if (ld_blk) {   // log|det C| partial of this 64-row block (wave 0)
    float lg = logf(cii), bad = (cii > 0.f) ? 0.f : 1.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lg += __shfl_xor(lg, o, 64);
      bad += __shfl_xor(bad, o, 64);
    }
    if (tid == 0) {
      a.ld_part[rb] = (double)lg;
      a.ld_part[nrb + rb] = (double)bad;
    }
  }
}

// -----------------------------------------------------------------------------------------------------------------
// Host side: work lists
// -----------------------------------------------------------------------------------------------------------------
namespace {

constexpr int kVBK = 64;   // vjp work item: k range in units of 64 (informational: the kernels derive their runs from M)

struct Item {
  int rb, cb, s0, s1, slab, flags, cost;
};

// interleave 8 per-XCD lists so that workgroup b (observed to run on XCD b % 8) takes list b % 8's next item
std::vector<Item> interleave8(std::vector<std::vector<Item>> &lists) {
  while (true) {   // even the lists out (a wrong XCD guess costs speed only)
    int mx = 0, mn = 0;
    for (int x = 1; x < 8; ++x) {
      if (lists[x].size() > lists[mx].size()) mx = x;
      if (lists[x].size() < lists[mn].size()) mn = x;
    }
    if (lists[mx].size() <= lists[mn].size() + 1) break;
    lists[mn].push_back(lists[mx].back());
    lists[mx].pop_back();
  }
  std::vector<Item> out;
  size_t L = 0;
  for (auto &l : lists) L = std::max(L, l.size());
  for (size_t s2 = 0; s2 < L; ++s2)
    for (int x = 0; x < 8; ++x)
      if (s2 < lists[x].size()) out.push_back(lists[x][s2]);
  return out;
}

void upload(mivi_ctx *c, DevBuf &b, const void *src, size_t bytes) {
  if (b.bytes < bytes || !b.p) {
    if (b.p) (void)hipFree(b.p);
    (void)hipMalloc(&b.p, bytes);
    b.bytes = bytes;
  }
  (void)hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice);
}

std::vector<int4> pack(const std::vector<Item> &v) {
  std::vector<int4> t(v.size());
  for (size_t i = 0; i < v.size(); ++i) t[i] = make_int4(v[i].rb | (v[i].cb << 16), v[i].s0 | (v[i].s1 << 16), v[i].slab, v[i].flags);
  return t;
}

void build_vjp(mivi_ctx *c, int d, int M, int tile, DevBuf &tab, int &n_items) {
  const int kVBM = tile, kVBN = tile;   // (shadows the 32 x 32 constants: the same table for either tile size)
  const int nrb = d / kVBM;
  const int SR = 256 / kVBM, SC = 256 / kVBN;   // super-blocks of 256 x 256 elements: one XCD's L2 holds their operands
  std::vector<std::vector<Item>> sbs;
  for (int sr = 0; sr * SR < nrb; ++sr)
    for (int sc = 0; sc <= sr; ++sc) {
      std::vector<Item> t;
      for (int rb = sr * SR; rb < (sr + 1) * SR && rb < nrb; ++rb)
        for (int cb = sc * SC; cb < (sc + 1) * SC && cb * kVBN <= rb * kVBM + kVBM - 1; ++cb) {
          const bool diag = cb * kVBN + kVBN - 1 >= rb * kVBM;          // touches the diagonal
          const bool mu = cb * kVBN == rb * kVBM;
          const bool half = cb * kVBN > rb * kVBM;                      // only the lower 32x32 sub-tile has work
          t.push_back(Item{rb, cb, 0, M / kVBK, 0, (diag ? 1 : 0) | (mu ? 2 : 0), half ? 1 : 2});
        }
      if (!t.empty()) sbs.push_back(t);
    }
  std::sort(sbs.begin(), sbs.end(), [](const std::vector<Item> &x, const std::vector<Item> &y) { return x.size() > y.size(); });
  std::vector<std::vector<Item>> lists(8);
  for (auto &sb : sbs) {
    int best = 0;
    for (int x = 1; x < 8; ++x)
      if (lists[x].size() < lists[best].size()) best = x;
    lists[best].insert(lists[best].end(), sb.begin(), sb.end());
  }
  for (auto &l : lists) std::stable_sort(l.begin(), l.end(), [](const Item &x, const Item &y) { return x.cost > y.cost; });
  auto items = interleave8(lists);
  n_items = (int)items.size();
  auto packed = pack(items);
  upload(c, tab, packed.data(), packed.size() * sizeof(int4));
}

// strips of k_fr_vjp32s: up to NS consecutive tiles of one block row per workgroup.  XCD x (= workgroup index % 8) takes whole 8 x 8-tile
// super-blocks of the lower triangle (its L2 then holds 8 W panels + 8 eps panels per super-block: 5 MB over the fabric per estimate at the
// north star, against 9 MB when an XCD owns whole block rows and so pulls every eps panel), super-blocks dealt heaviest first onto the
// lightest XCD; with fewer than eight super-blocks (d < 1024) the block rows {x, 15 - x, 16 + x, 31 - x} instead.
void build_strips(mivi_ctx *c, int d, int NS, DevBuf &tab, int &n_items) {
  const int nrb = d / 32, nsr = (nrb + 7) / 8;
  std::vector<std::vector<int4>> lists(8);
  if (nsr * (nsr + 1) / 2 >= 8) {
    std::vector<std::vector<int4>> sbs;
    std::vector<int> tiles;
    for (int sr = 0; sr < nsr; ++sr)
      for (int sc = 0; sc <= sr; ++sc) {
        std::vector<int4> t;
        int nt = 0;
        for (int rb = sr * 8; rb < sr * 8 + 8 && rb < nrb; ++rb) {
          const int c_hi = (sc * 8 + 8 < rb + 1) ? sc * 8 + 8 : rb + 1;
          for (int cb0 = sc * 8; cb0 < c_hi; cb0 += NS) {
            const int nc = (c_hi - cb0 < NS) ? c_hi - cb0 : NS;
            t.push_back(make_int4(rb | (cb0 << 16), nc, 0, 0));
            nt += nc;
          }
        }
        if (!t.empty()) { sbs.push_back(t); tiles.push_back(nt); }
      }
    std::vector<int> order(sbs.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int p, int q) { return tiles[p] > tiles[q]; });
    int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i : order) {
      int best = 0;
      for (int x = 1; x < 8; ++x)
        if (load[x] < load[best]) best = x;
      lists[best].insert(lists[best].end(), sbs[i].begin(), sbs[i].end());
      load[best] += tiles[i];
    }
  } else {
    for (int rb = 0; rb < nrb; ++rb) {
      const int r = rb & 15, x = (r < 8) ? r : 15 - r;
      for (int cb0 = 0; cb0 <= rb; cb0 += NS) {
        const int nc = (rb + 1 - cb0 < NS) ? rb + 1 - cb0 : NS;
        lists[x].push_back(make_int4(rb | (cb0 << 16), nc, 0, 0));
      }
    }
  }
  size_t L = 0;
  for (auto &l : lists) {
    std::stable_sort(l.begin(), l.end(), [](const int4 &p, const int4 &q) { return p.y > q.y; });
    L = l.size() > L ? l.size() : L;
  }
  std::vector<int4> out;
  for (size_t i = 0; i < L; ++i)
    for (int x = 0; x < 8; ++x) out.push_back(i < lists[x].size() ? lists[x][i] : make_int4(0, 0, 0, 0));
  n_items = (int)out.size();
  upload(c, tab, out.data(), out.size() * sizeof(int4));
}

}  // namespace

bool lds_path_shape_ok(const mivi_ctx *c, int M) {
  static const bool off = getenv("MIVI_FR_GEN1") != nullptr;   // A/B: first-generation tile kernels
  return !off && c->cfg.family == MIVI_FULLRANK && c->cfg.dtype == MIVI_F32 && c->cfg.d % 64 == 0 && M % 128 == 0 &&
         c->cfg.d <= 65535 * 32 && M > 0;
}

static bool f32_mfma();
static int knock_flags();
static int vjp_strip_len() {   // MIVI_VJP_STRIP: tiles per workgroup of the lane-batched VJP (k_fr_vjp32s); 0 = one tile per workgroup (k_fr_vjp32m)
  static const int v = getenv("MIVI_VJP_STRIP") ? atoi(getenv("MIVI_VJP_STRIP")) : 3;   // (748 workgroups at the north star: one round of three per CU)
  return v < 0 ? 0 : (v > 32 ? 32 : v);
}

// (re)build the VJP work lists for M samples per launch (the product kernels derive their tile from blockIdx)
bool lds_prepare(mivi_ctx *c, int M) {
  if (c->lds_M == M && c->lds_tabV.p) return true;
  invalidate_graph(c);   // a captured graph bakes the list contents and sizes
  const int d = c->cfg.d;
  build_vjp(c, d, M, 32, c->lds_tabV, c->lds_nV);
  build_vjp(c, d, M, 64, c->lds_tabV64, c->lds_nV64);
  build_strips(c, d, vjp_strip_len() ? vjp_strip_len() : 1, c->lds_tabS, c->lds_nS);
  if (!c->lds_tabV.p || !c->lds_tabV64.p || !c->lds_tabS.p) return false;
  c->lds_M = M;
  return true;
}

// Stein accumulation stage on the second-generation kernels: A (+)= eps G^T (all of it), gsum (+)= G 1, one 64 x 64 tile per workgroup
// (k_fr_vjp64<.., STEIN>); self: the chunk's value partials are assembled by one more workgroup of the same launch.
bool lds_stein_ok(const mivi_ctx *c, int M) {
  static const bool off = getenv("MIVI_STEIN_GEN1") != nullptr;   // A/B: the first-generation kernel
  return !off && lds_path_shape_ok(c, M) && !f32_mfma();   // (d is a multiple of 64: the padded leading dimension dP equals d)
}
void launch_lds_stein_outer(mivi_ctx *c, int M, void *A, double *gsum, int first, double scale, double n, void *grad, void *logpi,
                            const ValueJob *self) {
  const int d = c->cfg.d, nb = d / 64;
  if (c->lds_st_d != d) {   // full square of 64 x 64 tiles; 4 x 4 super-blocks (256 x 256 elements: one XCD's L2 holds their operands)
    std::vector<std::vector<Item>> lists(8);
    int sbi = 0;
    for (int sr = 0; sr * 4 < nb; ++sr)
      for (int sc = 0; sc * 4 < nb; ++sc, ++sbi)
        for (int rb = sr * 4; rb < sr * 4 + 4 && rb < nb; ++rb)
          for (int cb = sc * 4; cb < sc * 4 + 4 && cb < nb; ++cb) lists[sbi % 8].push_back(Item{rb, cb, 0, 0, 0, 0, 1});
    auto items = interleave8(lists);
    auto packed = pack(items);
    upload(c, c->lds_tabSt, packed.data(), packed.size() * sizeof(int4));
    c->lds_nSt = (int)items.size();
    c->lds_st_d = d;
  }
  GemmArgs a{};
  a.d = d; a.M = M; a.dP = c->dP;
  a.A = (const float *)c->eps[c->cur].p; a.lda = c->dP;   // rows of the product: eps coordinates
  a.B = (const float *)c->W.p; a.ldb = d;                  // columns: the target's gradient coordinates
  a.work = (const int4 *)c->lds_tabSt.p;
  a.n_work = 0x7fffffff;
  a.dbg = c->dbg;
  a.knock = knock_flags();
  a.st_A = (float *)A; a.st_ld = c->dP;
  a.st_gsum = gsum;
  a.st_grad = (float *)grad;
  a.st_logpi = (float *)logpi;
  a.st_first = first;
  a.st_scale = (float)scale;
  a.st_n = n;
  int grid = c->lds_nSt;
  if (self) {
    a.n_work = grid;
    a.self_vin = self->vin;
    a.self_out = self->out;
    grid += 1;
  }
  hipLaunchKernelGGL((k_fr_vjp64<false, true, true>), dim3(grid), dim3(512), 0, c->stream, a);
}

int lds_ld_blocks(const mivi_ctx *c) { return c->cfg.d / 64; }

static bool f32_mfma() {   // MIVI_FR_F32MFMA=1: v_mfma_f32_32x32x2_f32 chains instead of bf16x3 (A/B reference)
  static const bool v = getenv("MIVI_FR_F32MFMA") != nullptr;
  return v;
}
static int knock_flags() {   // developer knock-outs (operand loads / MFMAs / epilogue / stores): -DMIVI_DEV builds only (make DEV=1)
#ifdef MIVI_DEV
  static const int v = getenv("MIVI_KNOCK") ? atoi(getenv("MIVI_KNOCK")) : 0;
  return v;
#else
  return 0;
#endif
}

// eps(t+1) rider blocks of the 64 x 64 product kernel: 512 threads = 64 rows x 32 columns each
struct LaneSink {   // one lane's recorded launches of an estimate (see launch_lanes_prod / launch_lanes_vjp)
  Prod32Args prod[2];
  int prod_grid[2], prod_dense[2], n_prod;
  GemmArgs vjp;
  int vjp_grid, n_vjp;
};
int lds_eps_blocks(const mivi_ctx *c, int M) { return (c->cfg.d / 64) * (M / 32); }

// unsplit 32 x 32-tile product + fused epilogue (k_fr_prod32).  dense = false: Z = mu + tril(C) eps with `mode` in
// {R_DIAG, R_DENSE_R, R_PLAIN}; dense = true: G = -P (Z - m) (mode R_DENSE_G, R = Z - m in c->RT).
void launch_lds_prod32(mivi_ctx *c, const void *params, int M, bool dense, int mode, void *Z, const EpsJob *next, bool want_ld,
                       bool with_dinv) {
  Prod32Args a{};
  a.d = c->cfg.d; a.M = M; a.dP = c->dP; a.mode = mode;
  if (dense) { a.A = (const float *)c->t_prec.p; a.lda = c->dP; a.B = (const float *)c->RT.p; }
  else { a.A = (const float *)params + c->cfg.d; a.lda = c->cfg.d; a.B = (const float *)c->eps[c->cur].p; }
  a.params = (const float *)params;
  a.t_mean = (const float *)c->t_mean.p;
  a.t_istd = (const float *)c->t_istd.p;
  a.Z = (float *)Z;
  a.W = (float *)c->W.p;
  a.R = (float *)c->RT.p;
  a.ell_part = (double *)c->ell_part[c->cur].p;
  a.ld_part = want_ld ? (double *)c->ld_part[c->cur].p : nullptr;
  a.ncb = M / 32;
  a.n_tiles = (c->cfg.d / 32) * a.ncb;
  a.dbg = c->dbg;
  a.knock = knock_flags();
  int grid = a.n_tiles;
  if (next) {
    a.next_eps.d = c->cfg.d;
    a.next_eps.M = M;
    a.next_eps.rng = next->rng;
    a.next_eps.eps = (float *)c->eps[next->parity].p;
    a.next_eps.ld_eps = c->dP;
    a.next_eps.he_part = (double *)c->he_part[next->parity].p;
    a.n_eps = (c->cfg.d / 64) * (M / 32);
  }
  if (with_dinv && !dense) {
    a.stl_pack = (unsigned *)c->stl_F.p;
    a.n_dinv = c->cfg.d / 64;
    a.n_pack = stl_pack_riders(c->cfg.d);
  }
  // riders hitch onto the tile workgroups; only the ones beyond the tile count get workgroups of their own
  grid = a.n_tiles > a.n_dinv + a.n_pack + a.n_eps ? a.n_tiles : a.n_dinv + a.n_pack + a.n_eps;
  if (c->lane_sink) {   // lane-batched estimates (api_batch.hip): record the launch, the driver issues it for all lanes at once
    LaneSink &sk = ((LaneSink *)c->lane_sink)[c->lane_id];
    if (sk.n_prod < 2) { sk.prod[sk.n_prod] = a; sk.prod_grid[sk.n_prod] = grid; sk.prod_dense[sk.n_prod] = dense ? 1 : 0; }
    ++sk.n_prod;
    return;
  }
  if (dense && f32_mfma()) hipLaunchKernelGGL((k_fr_prod32<G_DENSE, false>), dim3(grid), dim3(512), 0, c->stream, a);
  else if (dense) hipLaunchKernelGGL((k_fr_prod32<G_DENSE, true>), dim3(grid), dim3(512), 0, c->stream, a);
  else if (f32_mfma()) hipLaunchKernelGGL((k_fr_prod32<G_SAMPLE, false>), dim3(grid), dim3(512), 0, c->stream, a);
  else hipLaunchKernelGGL((k_fr_prod32<G_SAMPLE, true>), dim3(grid), dim3(512), 0, c->stream, a);
}

// ---- lane-batched launches: up to kMaxLanes contexts' product / VJP kernels as ONE launch each (blockIdx.y = lane) ---------------------
LaneSink *lane_sinks_alloc(int n) { return new LaneSink[n](); }
void lane_sinks_free(LaneSink *s) { delete[] s; }
void lane_sink_reset(LaneSink *s, int lane) { s[lane].n_prod = 0; s[lane].n_vjp = 0; }
int lane_sink_counts(const LaneSink *s, int lane) { return s[lane].n_prod * 16 + s[lane].n_vjp; }
static bool prod_quad_on() {   // MIVI_PROD_QUAD=0: four lanes' sampling products as 4 x k_fr_prod32's tiles again (A/B)
  static const bool v = !(getenv("MIVI_PROD_QUAD") && atoi(getenv("MIVI_PROD_QUAD")) == 0);
  return v;
}
// which: 0 = the sampling product (with the fused diagonal target, or R = Z - m of the dense one), 1 = the dense target's product
bool launch_lanes_prod(mivi_ctx *c, LaneSink *s, int lanes, int which) {
  if (lanes < 1 || lanes > kMaxLanes || f32_mfma()) return false;
  Prod32Multi m;
  for (int l = 0; l < lanes; ++l) {
    if (s[l].n_prod <= which || s[l].n_prod > 2 || s[l].prod_dense[which] != which || s[l].prod[which].n_tiles != s[0].prod[which].n_tiles) return false;
    m.lane[l] = s[l].prod[which];
  }
  int gx = 0;   // (the lane that carries the STL riders may need trailing workgroups the others do not: d = 2048)
  for (int l = 0; l < lanes; ++l) gx = s[l].prod_grid[which] > gx ? s[l].prod_grid[which] : gx;
  if (which == 0 && lanes == 4 && prod_quad_on()) {   // two tiles x two lanes per workgroup (k_fr_prod32q) where it applies
    const Prod32Args &a = m.lane[0];
    const int nrb = a.d >> 5;
    bool ok = a.n_dinv == 0 && a.n_pack == 0 && (nrb & 3) == 0 && (a.ncb & 1) == 0 && a.n_tiles >= 256 &&
              (a.mode == R_DIAG || a.mode == R_DENSE_R || a.mode == R_PLAIN);
    for (int l = 1; l < lanes && ok; ++l) {
      const Prod32Args &o = m.lane[l];
      ok = o.A == a.A && o.params == a.params && o.lda == a.lda && o.d == a.d && o.M == a.M && o.dP == a.dP && o.mode == a.mode &&
           o.t_mean == a.t_mean && o.t_istd == a.t_istd && o.ncb == a.ncb && o.n_dinv == 0 && o.n_pack == 0 && o.n_eps == a.n_eps &&
           (o.ld_part != nullptr) == (a.ld_part != nullptr);
    }
    if (ok) {
      hipLaunchKernelGGL(k_fr_prod32q, dim3((nrb >> 1) * a.ncb * 2), dim3(512), 0, c->stream, m);
      return true;
    }
  }
  const dim3 grid(gx, lanes);
  if (which) hipLaunchKernelGGL((k_fr_prod32m<G_DENSE, true>), grid, dim3(512), 0, c->stream, m);
  else hipLaunchKernelGGL((k_fr_prod32m<G_SAMPLE, true>), grid, dim3(512), 0, c->stream, m);
  return true;
}
bool launch_lanes_vjp(mivi_ctx *c, LaneSink *s, int lanes) {
  if (lanes < 1 || lanes > kMaxLanes || f32_mfma()) return false;
  GemmMulti m;
  for (int l = 0; l < lanes; ++l) {
    if (s[l].n_vjp != 1 || s[l].vjp_grid != s[0].vjp_grid) return false;
    m.lane[l] = s[l].vjp;
  }
  if (vjp_strip_len() > 0 && lanes >= 2 && m.lane[0].M == 256 && c->lds_tabS.p) {   // strips of tiles per workgroup where they apply
    bool ok = true;
    const bool self = m.lane[0].n_work != 0x7fffffff;   // (one more workgroup per lane assembles the estimate's value)
    for (int l = 0; l < lanes; ++l) ok = ok && (m.lane[l].n_work != 0x7fffffff) == self && m.lane[l].M == 256 && m.lane[l].d == m.lane[0].d;
    if (ok) {
      StripMulti sm;
      for (int l = 0; l < lanes; ++l) {
        sm.lane[l] = m.lane[l];
        if (self) sm.lane[l].n_work = c->lds_nS;
      }
      sm.strips = (const int4 *)c->lds_tabS.p;
      hipLaunchKernelGGL(k_fr_vjp32s, dim3(c->lds_nS + (self ? 1 : 0), lanes), dim3(256), 0, c->stream, sm);
      return true;
    }
  }
  hipLaunchKernelGGL((k_fr_vjp32m<false, true>), dim3(s[0].vjp_grid, lanes), dim3(256), 0, c->stream, m);
  return true;
}
// unsplit 64 x 64-tile product + fused epilogue for the large shapes (k_fr_prod64); arguments as launch_lds_prod32
void launch_lds_prod64(mivi_ctx *c, const void *params, int M, bool dense, int mode, void *Z, const EpsJob *next, bool want_ld) {
  Prod32Args a{};
  a.d = c->cfg.d; a.M = M; a.dP = c->dP; a.mode = mode;
  if (dense) { a.A = (const float *)c->t_prec.p; a.lda = c->dP; a.B = (const float *)c->RT.p; }
  else { a.A = (const float *)params + c->cfg.d; a.lda = c->cfg.d; a.B = (const float *)c->eps[c->cur].p; }
  a.params = (const float *)params;
  a.t_mean = (const float *)c->t_mean.p;
  a.t_istd = (const float *)c->t_istd.p;
  a.Z = (float *)Z;
  a.W = (float *)c->W.p;
  a.R = (float *)c->RT.p;
  a.ell_part = (double *)c->ell_part[c->cur].p;
  a.ld_part = want_ld ? (double *)c->ld_part[c->cur].p : nullptr;
  a.ncb = M / 64;
  a.n_tiles = (c->cfg.d / 64) * a.ncb;
  a.dbg = c->dbg;
  int grid = a.n_tiles;
  if (next) {
    a.next_eps.d = c->cfg.d;
    a.next_eps.M = M;
    a.next_eps.rng = next->rng;
    a.next_eps.eps = (float *)c->eps[next->parity].p;
    a.next_eps.ld_eps = c->dP;
    a.next_eps.he_part = (double *)c->he_part[next->parity].p;
    a.n_eps = (c->cfg.d / 64) * (M / 32);
    grid += a.n_eps;
  }
  if (dense && f32_mfma()) hipLaunchKernelGGL((k_fr_prod64<G_DENSE, false>), dim3(grid), dim3(512), 0, c->stream, a);
  else if (dense) hipLaunchKernelGGL((k_fr_prod64<G_DENSE, true>), dim3(grid), dim3(512), 0, c->stream, a);
  else if (f32_mfma()) hipLaunchKernelGGL((k_fr_prod64<G_SAMPLE, false>), dim3(grid), dim3(512), 0, c->stream, a);
  else hipLaunchKernelGGL((k_fr_prod64<G_SAMPLE, true>), dim3(grid), dim3(512), 0, c->stream, a);
}
int lds_prod64_tiles(const mivi_ctx *c, int M) { return (c->cfg.d / 64) * (M / 64); }
// beyond the 32 x 32 kernel's range (d n_mc > 1024 x 512: at least 128 tiles of 64 x 64) the 64 x 64 kernel; at and below the boundary
// the 32 x 32 kernel wins (1024 x 512: 11.8 vs 14.7 us).  (The split-K slabs + reduce route these two replaced -- 1536 x 512 sample stage
// 25.4 -> 20.0 us, 4096 x 1024 206 -> 119 us -- was removed in round 3: no default path reached it.)
bool lds_use_prod64(const mivi_ctx *c, int M) { return !lds_use_prod32(c, M); }
int lds_prod32_tiles(const mivi_ctx *c, int M) { return (c->cfg.d / 32) * (M / 32); }
int lds_prod32_eps_blocks(const mivi_ctx *c, int M) { return (c->cfg.d / 64) * (M / 32); }
// the unsplit 32 x 32 product is bounded by its heaviest tile (d/32 sub-stages on one CU); beyond this the 64 x 64 kernel takes over
bool lds_bf16x3() { return !f32_mfma(); }
bool lds_use_prod32(const mivi_ctx *c, int M) { return (long long)c->cfg.d * M <= 1024LL * 512; }

// tril(W eps^T) (+ d/dmu); self != nullptr: one trailing workgroup assembles this estimate's objective value
void launch_lds_vjp(mivi_ctx *c, const void *params, int M, const OutArgs &out, const ValueJob *self, const FusedUpdate *upd) {
  GemmArgs a{};
  a.d = c->cfg.d; a.M = M; a.dP = c->dP;
  a.A = (const float *)c->W.p; a.lda = c->cfg.d;
  a.B = (const float *)c->eps[c->cur].p; a.ldb = c->dP;
  a.work = (const int4 *)c->lds_tabV.p;
  a.n_work = 0x7fffffff;
  a.params = (const float *)params;
  a.out = out;
  a.dbg = c->dbg;
  a.knock = knock_flags();
  int grid = c->lds_nV;
  if (self) {
    a.n_work = c->lds_nV;
    a.self_vin = self->vin;
    a.self_out = self->out;
    grid += 1;
  }
  if (upd) a.upd = *upd;
  // tile size: 64 x 64 (half the operand bytes per flop) pays once a wave has several sub-stages to overlap its one-deep
  // staging with -- measured: 4096 x 1024 VJP 159 -> 145 us (118 TF), but 1024 x 256 6.0 -> 7.4 us and 2048 x 256 16.9 -> 20.4 us
  // (a wave's single sub-stage there is load, then compute, then a four times larger epilogue, nothing overlapped, against
  // two or three 32 x 32 workgroups per CU covering for each other).  MIVI_VJP_TILE=32 / 64 pins it.
  static const int pin = getenv("MIVI_VJP_TILE") ? atoi(getenv("MIVI_VJP_TILE")) : 0;
  const bool t64 = pin ? pin == 64 : ((M >= 1024 && c->cfg.d >= 2048) || (M >= 512 && c->cfg.d >= 4096));   // (4096 x 512: 90.6 -> 83.7 us)
  if (t64 && c->lane_sink) { ((LaneSink *)c->lane_sink)[c->lane_id].n_vjp += 16; return; }   // (not a lane-batched route: the driver reports it)
  if (t64) {
    a.work = (const int4 *)c->lds_tabV64.p;
    grid = c->lds_nV64;
    if (self) { a.n_work = c->lds_nV64; grid += 1; }
    if (upd && f32_mfma()) hipLaunchKernelGGL((k_fr_vjp64<true, false>), dim3(grid), dim3(512), 0, c->stream, a);
    else if (upd) hipLaunchKernelGGL((k_fr_vjp64<true, true>), dim3(grid), dim3(512), 0, c->stream, a);
    else if (f32_mfma()) hipLaunchKernelGGL((k_fr_vjp64<false, false>), dim3(grid), dim3(512), 0, c->stream, a);
    else hipLaunchKernelGGL((k_fr_vjp64<false, true>), dim3(grid), dim3(512), 0, c->stream, a);
    return;
  }
  if (c->lane_sink) {   // lane-batched estimates: record (the driver only enters this mode where this route is taken)
    LaneSink &sk = ((LaneSink *)c->lane_sink)[c->lane_id];
    if (sk.n_vjp < 1 && !upd) { sk.vjp = a; sk.vjp_grid = grid; }
    sk.n_vjp += upd ? 16 : 1;
    return;
  }
  if (upd && f32_mfma()) hipLaunchKernelGGL((k_fr_vjp32<true, false>), dim3(grid), dim3(256), 0, c->stream, a);
  else if (upd) hipLaunchKernelGGL((k_fr_vjp32<true, true>), dim3(grid), dim3(256), 0, c->stream, a);
  else if (f32_mfma()) hipLaunchKernelGGL((k_fr_vjp32<false, false>), dim3(grid), dim3(256), 0, c->stream, a);
  else hipLaunchKernelGGL((k_fr_vjp32<false, true>), dim3(grid), dim3(256), 0, c->stream, a);
}

}  // namespace mivi
