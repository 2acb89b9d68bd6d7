// Full-rank RepGradELBO contractions for a BATCH of estimates at the same parameters (gfx950): third generation.
//
// Reference semantics (AdvancedVI.jl v0.7.0), per estimate unchanged from kernels_fullrank_lds.hip:
//   sampling   Z = scale * eps .+ mu                                  src/families/location_scale.jl:71-77
//   energy     mean_m logdensity(prob, z_m)                           src/algorithms/repgradelbo.jl:84-86
//   gradient   d/dC = -(1/M) tril(W eps') - direct * diag(1/C_ii),  d/dmu = -(1/M) W 1   (repgradelbo.jl:142-149)
// The reference evaluates ONE estimate per `estimate_gradient!` call; estimates at fixed parameters (monitoring with many samples,
// averaged gradients, the bench's step) are independent, and L of them are ONE matrix product each way:
//   product   [Z_1 .. Z_L] = mu + tril(C) [eps_1 .. eps_L]            1024 x (256 L) x 1024 (triangular) at the north star
//   VJP       dC_l = tril(W_l eps_l'),  l = 1 .. L                    L products 1024 x 1024 (lower) x 256
// The second generation gives every 32 x 32 tile of ONE estimate a workgroup whose waves split K (latency-bound launches, every operand
// element split into its bf16 pieces by every tile that uses it: 20 vector instructions per MFMA).  Here the launch is shaped like the
// large product it is:
//   * a workgroup owns a (64 WGM) x (64 WGN) output tile (128 x 128), a wave a 64 x 64 part of it (2 x 2 MFMA tiles: every operand
//     fragment is split into bf16 pieces once for two tiles, 7 vector instructions per MFMA) over the WHOLE K range;
//   * operands are staged through LDS once per workgroup (LDS-DMA, 1 KiB pieces, two stages: the DMA of sub-stage t + 1 runs under the
//     MFMAs of t), one barrier per 32-k sub-stage;
//   * no cross-wave reduction: the epilogue works on a wave's own accumulators (transposed through a wave-private LDS image so that
//     stores are whole 128-byte lines);
//   * the launch covers every lane (estimate) of the step: per-lane buffers are base + lane * stride, the work table names (lane, tile).
// BIT-IDENTICAL to the one-estimate kernels (k_fr_prod32 / k_fr_vjp32): those cut a tile's K range into runs (one per wave: eight for the
// product, four for the VJP), every run an MFMA chain from zero, the runs summed in wave order.  A wave here walks the same runs one after
// the other -- chain accumulator `acc`, folded into `tot` at every run boundary (tot = tot + acc: the same f32 additions in the same
// order) -- with the same k-slot assignment inside every MFMA and the same per-element epilogue arithmetic (fr_elem.h), the same wave
// sums behind every ell partial and the same slots for them.  So "a batch's estimates are bitwise the single calls'" holds by construction
// (tests/test_gpu_batches.py).
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "device_common.h"
#include "fr_elem.h"

namespace mivi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

#define FB_GLDS16(gptr, lptr)                                                                              \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                 \
                                   (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

__device__ __forceinline__ void fb_wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// exact three-way bf16 split of eight f32 values (kernels_fullrank_lds.hip split3_bf16: the same pieces)
__device__ __forceinline__ void fb_split3(const float *x, bf16x8 &hi, bf16x8 &mid, bf16x8 &lo) {
  u32x4v uh, um, ul;
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
  hi = __builtin_bit_cast(bf16x8, uh);
  mid = __builtin_bit_cast(bf16x8, um);
  lo = __builtin_bit_cast(bf16x8, ul);
}

// the six products of one 32 x 32 x 16 block, smallest terms first (mfma_bf16x3's order)
__device__ __forceinline__ void fb_mfma6(const bf16x8 &ah, const bf16x8 &am, const bf16x8 &al, const bf16x8 &bh, const bf16x8 &bm,
                                         const bf16x8 &bl, f32x16 &c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
}

struct FbArgs {
  int d, M, dP, L;
  const float *params;            // [mu; vec C]
  const float *t_mean, *t_istd;   // diagonal-Gaussian target
  float *eps;                     // lane l: eps + l * eps_stride, eps[i + m * dP]
  long long eps_stride;
  float *W;                       // lane l: W + l * W_stride, W[i + m * d]
  long long W_stride;
  double *ell_part;               // lane l: ell_part + l * ell_stride; slots = k_fr_prod32's workgroup indices
  long long ell_stride;
  double *he_part;                // lane l: he_part + l * he_stride
  long long he_stride;
  double *ld_part;                // [2][d / 32] (parameters only: written once per launch, shared by the lanes)
  const int4 *work;               // {lane, rb | cb << 16, flags, 0}
  int n_work;
  // VJP / value outputs
  float *grads;                   // lane l (but the one that writes the caller's buffers): grads + l * grad_stride
  long long grad_stride;
  float *values;                  // lane l: values + l * value_stride
  long long value_stride;
  float *grad_last, *value_last;  // lane L_last writes these instead (nullptr: every lane writes grads / values)
  int lane_last;
  int write_upper;                // 1: every lane writes the exact zeros above the diagonal; 0: only lane_last does (the others' buffers hold them already)
  int ent_kind, M_total;
  int *status;
  double ell_const;
  // eps draws
  RngArgs rng;                    // lane l draws estimate rng_index(rng) + l
};

// -----------------------------------------------------------------------------------------------------------------
// k_fb_eps: eps of L estimates, the blocks of the product kernels' riders (64 rows x 32 columns, one Philox block per thread) -- the
// same stream, the same layout and the same he_part partials as k_eps_m / the riders of k_fr_prod32.  blockIdx.y = lane.
// -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_fb_eps(FbArgs a) {
  __shared__ double red[8];
  const int tid = threadIdx.x, eb = blockIdx.x, l = blockIdx.y, d = a.d, nrb6 = d >> 6;
  const int ri = (eb % nrb6) * 64 + 4 * (tid & 15), rm = (eb / nrb6) * 32 + (tid >> 4);
  float e[4];
  eps_block<float>(a.rng.seed, rng_index(a.rng) + (uint64_t)l, (uint64_t)(a.rng.m_offset + rm) * (uint64_t)(d >> 2) + (uint64_t)(ri >> 2), e);
  const f32x4 ev = {e[0], e[1], e[2], e[3]};
  store16_wt(a.eps + (size_t)l * a.eps_stride + (size_t)rm * a.dP + ri, ev);
  const float he = 0.5f * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3]);
  const double sh = block_sum_nodrain_f32<512>(he, red);
  if (tid == 0) a.he_part[(size_t)l * a.he_stride + eb] = sh;
}

// -----------------------------------------------------------------------------------------------------------------
// k_fb_prod: W_l = grad log pi(mu + tril(C) eps_l) + ell partials, for every lane l of the step.
// Work item = (lane, rb, cb): rows [BM rb, BM rb + BM) of columns [BN cb, BN cb + BN) of lane l; K = 32 (last 32-row block index + 1).
// LDS stage (32 k): A panel as BM / 32 blocks [32 k][32 rows] (k_fr_prod32's image), B panel [BN columns][32 k] with XOR-swizzled
// 16-byte chunks (k_fr_prod32's image).  Sub-stage t of a 32-row block r32: active for t <= r32, diagonal mask at t == r32.
// Run boundaries of row block r32 (nst = r32 + 1 sub-stages): t_beg(w) = (w nst) >> 3, w = 1 .. 7 -- k_fr_prod32's eight runs.
// -----------------------------------------------------------------------------------------------------------------
template <int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void k_fb_prod(FbArgs a) {
  constexpr int NW = WGM * WGN, BM = 64 * WGM, BN = 64 * WGN, LDC = 36;
  constexpr int A_F = BM * 32, B_F = BN * 32, STAGE_F = A_F + B_F;
  constexpr int PPA = (BM / 8) / NW, PPB = (BN / 8) / NW;   // 1 KiB pieces per wave and stage
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "pieces per wave");
  constexpr int EPI_F = NW * 32 * LDC;
  constexpr int MAIN_F = 2 * STAGE_F > EPI_F ? 2 * STAGE_F : EPI_F;
  __shared__ __attribute__((aligned(16))) float lds[MAIN_F + 3 * BM];
  float *vec = lds + MAIN_F;   // mu, target mean, target 1 / std of the tile's rows
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / WGN, wn = w % WGN;
  const __attribute__((address_space(4))) int *wp = (const __attribute__((address_space(4))) int *)a.work + 4 * blockIdx.x;
  const int ln = wp[0], rc = wp[1], flags = wp[2];
  const int rb = rc & 0xffff, cb = rc >> 16;
  const int d = a.d, dP = a.dP;
  const int row0 = rb * BM, col0 = cb * BN;
  const int R0 = row0 >> 5;                       // first 32-row block of the tile
  const int T = R0 + BM / 32;                     // sub-stages of the workgroup (the last row block's K)
  const float *A = a.params + d;                  // tril(C), A[row + k d]
  const float *B = a.eps + (size_t)ln * a.eps_stride;
  // row vectors of the epilogue
  for (int i = tid; i < BM; i += 64 * NW) {
    vec[i] = a.params[row0 + i];
    vec[BM + i] = a.t_mean[row0 + i];
    vec[2 * BM + i] = a.t_istd[row0 + i];
  }
  // staging: this wave's pieces
  const float *Ag[PPA];
  int a_blk[PPA];
#pragma unroll
  for (int i = 0; i < PPA; ++i) {
    const int pa = w * PPA + i, ab = pa >> 2, kq = pa & 3;
    a_blk[i] = ab;
    Ag[i] = A + row0 + 32 * ab + 4 * (lane & 7) + (size_t)(8 * kq + (lane >> 3)) * d;
  }
  const float *Bg[PPB];
#pragma unroll
  for (int i = 0; i < PPB; ++i) {
    const int n = 8 * (w * PPB + i) + (lane >> 3);
    Bg[i] = B + (size_t)(col0 + n) * dP + 4 * ((lane & 7) ^ ((n >> 1) & 7));
  }
  auto issue = [&](int t, int s) {
    float *dst = lds + s * STAGE_F;
#pragma unroll
    for (int i = 0; i < PPA; ++i) {
      const int pa = w * PPA + i;
      if (t <= R0 + a_blk[i]) FB_GLDS16(Ag[i] + (size_t)(32 * t) * d, dst + pa * 256);   // (blocks above the diagonal are never read)
    }
#pragma unroll
    for (int i = 0; i < PPB; ++i) FB_GLDS16(Bg[i] + 32 * t, dst + A_F + (w * PPB + i) * 256);
  };
  f32x16 acc[2][2], tot[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
  const int r32[2] = {R0 + 2 * wm, R0 + 2 * wm + 1};
  int wnext[2] = {1, 1};   // next run boundary of each row half
#pragma unroll
  for (int i = 0; i < 2; ++i)
    while (wnext[i] < 8 && ((wnext[i] * (r32[i] + 1)) >> 3) == 0) ++wnext[i];
  const int b_swz = (l31 >> 1) & 7;
  issue(0, 0);
  for (int t = 0; t < T; ++t) {
    fb_wait_vm0();
    lds_barrier();   // sub-stage t has landed for every wave; every wave is done with the other stage
    if (t + 1 < T) issue(t + 1, (t + 1) & 1);
    const float *cur = lds + (t & 1) * STAGE_F;
    bool act[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      act[i] = t <= r32[i];
      bool fold = false;
      while (wnext[i] < 8 && ((wnext[i] * (r32[i] + 1)) >> 3) == t) { fold = true; ++wnext[i]; }
      if (fold && act[i]) {   // a run of this row block ended before sub-stage t
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          tot[i][j] += acc[i][j];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
      }
    }
    if (!act[1]) continue;   // (act[0] implies act[1]: r32[0] < r32[1])
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      bf16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {   // B fragments: column 64 wn + 32 j + l31, k slots 8 (2 g + q) + 4 h + {0..3}
        const float *bc = cur + A_F + (64 * wn + 32 * j + l31) * 32;
        const f32x4 q0 = *(const f32x4 *)(bc + 4 * ((4 * g + h) ^ b_swz));
        const f32x4 q1 = *(const f32x4 *)(bc + 4 * ((4 * g + 2 + h) ^ b_swz));
        const float bv[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
        fb_split3(bv, bh[j], bm[j], bl[j]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (!act[i]) continue;
        const float *ac = cur + (2 * wm + i) * 1024 + l31;
        float av[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] = ac[(8 * (2 * g + (e >> 2)) + 4 * h + (e & 3)) * 32];
        if (t == r32[i]) {   // the diagonal block of tril(C): keep k <= row
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (8 * (2 * g + (e >> 2)) + 4 * h + (e & 3) > l31) av[e] = 0.f;
        }
        fb_split3(av, ah[i], am[i], al[i]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (!act[i]) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) fb_mfma6(ah[i], am[i], al[i], bh[j], bm[j], bl[j], acc[i][j]);
      }
    }
  }
  // the last run of each row block
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) tot[i][j] += acc[i][j];
  lds_barrier();   // every wave is done with the stages: LDS becomes the waves' private epilogue images
  float *Cs = lds + w * (32 * LDC);
  float *Wl = a.W + (size_t)ln * a.W_stride;
  double *ellp = a.ell_part + (size_t)ln * a.ell_stride;
  const int nrb = d >> 5, ncb = a.M >> 5;
  const bool xcd_slots = (nrb & 3) == 0 && (ncb & 1) == 0;   // (k_fr_prod32's block -> tile map)
  const int ei4 = 4 * (lane & 7);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lr = 64 * wm + 32 * i;   // row offset inside the tile
    const f32x4 mu = *(const f32x4 *)(vec + lr + ei4), tm = *(const f32x4 *)(vec + BM + lr + ei4), tis = *(const f32x4 *)(vec + 2 * BM + lr + ei4);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = {tot[i][j][4 * q], tot[i][j][4 * q + 1], tot[i][j][4 * q + 2], tot[i][j][4 * q + 3]};
        *(f32x4 *)(Cs + l31 * LDC + 8 * q + 4 * h) = v;
      }
      double s = 0.0;
#pragma unroll
      for (int p = 0; p < 4; ++p) {   // pass p = wave p of k_fr_prod32's epilogue: columns 8 p .. 8 p + 7, rows ei4 .. ei4 + 3 per lane
        const int en = 8 * p + (lane >> 3);
        const f32x4 v = *(const f32x4 *)(Cs + en * LDC + ei4);
        const f32x4 z = mu + v;
        float ell = 0.f;
        f32x4 wv;
#pragma unroll
        for (int c = 0; c < 4; ++c) wv[c] = diag_target_elem(z[c], tm[c], tis[c], ell);
        const int gi = row0 + lr + ei4, gm = col0 + 64 * wn + 32 * j + en;
        store16_wt(Wl + (size_t)gm * d + gi, wv);
        const double sv = (double)wave_sum_f32(ell);
        s = p ? s + sv : sv;
      }
      s += 0.0;   // (k_fr_prod32 adds its four idle waves' zeros: -0.0 becomes +0.0 there)
      if (lane == 0) {
        const int rbE = r32[i], cbE = (col0 + 64 * wn + 32 * j) >> 5, rE = nrb - 1 - rbE;
        const int slot = xcd_slots ? ((rE & 3) + 4 * (cbE & 1)) + 8 * ((rE >> 2) * (ncb >> 1) + (cbE >> 1)) : rE * ncb + cbE;
        ellp[slot] = s;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the image is read before the next tile overwrites it
    }
  }
  if ((flags & 1) && wn == 0 && lane < 32) {   // log|det C| partials of this wave's two row blocks (one workgroup per row block class carries the flag)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 32 * r32[i] + lane;
      float lg, bad;
      logdet_block32(a.params[d + (size_t)r * d + r], lg, bad);
      if (lane == 0) {
        a.ld_part[r32[i]] = (double)lg;
        a.ld_part[nrb + r32[i]] = (double)bad;
      }
    }
  }
}

// -----------------------------------------------------------------------------------------------------------------
// k_fb_vjp: dC_l = -(1/M) tril(W_l eps_l') - direct diag(1 / C_ii), dmu_l = -(1/M) W_l 1, for every lane l of the step.
// Work item = (lane, rb, cb), cb <= rb: the BM x BN tile of the lower triangle; both operands MN-major ([32 k][32 rows] blocks).
// k_fr_vjp32's four runs = the K quarters (M / 4 each); 32 x 32 sub-tiles strictly above the diagonal are skipped, the exact zeros
// of the upper triangle are written as the mirror images of the strictly lower ones (lanes with the write_upper duty).
// -----------------------------------------------------------------------------------------------------------------
template <int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void k_fb_vjp(FbArgs a) {
  constexpr int NW = WGM * WGN, BM = 64 * WGM, BN = 64 * WGN, LDC = 36;
  constexpr int A_F = BM * 32, B_F = BN * 32, STAGE_F = A_F + B_F;
  constexpr int PPA = (BM / 8) / NW, PPB = (BN / 8) / NW;
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "pieces per wave");
  constexpr int EPI_F = NW * 32 * LDC;
  constexpr int MAIN_F = 2 * STAGE_F > EPI_F ? 2 * STAGE_F : EPI_F;
  __shared__ __attribute__((aligned(16))) float lds[MAIN_F];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / WGN, wn = w % WGN;
  const __attribute__((address_space(4))) int *wp = (const __attribute__((address_space(4))) int *)a.work + 4 * blockIdx.x;
  const int ln = wp[0], rc = wp[1];
  const int rb = rc & 0xffff, cb = rc >> 16;
  const int d = a.d, dP = a.dP, M = a.M;
  const int row0 = rb * BM, col0 = cb * BN;
  const float *A = a.W + (size_t)ln * a.W_stride;      // W[row + m d]
  const float *B = a.eps + (size_t)ln * a.eps_stride;  // eps[row + m dP]
  const bool last = ln == a.lane_last && a.grad_last;
  float *grad = last ? a.grad_last : a.grads + (size_t)ln * a.grad_stride;
  const bool upper = a.write_upper || last;
  const int T = M >> 5, nsub = T >> 2;   // sub-stages; per K quarter
  const float *Ag[PPA], *Bg[PPB];
#pragma unroll
  for (int i = 0; i < PPA; ++i) {
    const int pa = w * PPA + i, ab = pa >> 2, kq = pa & 3;
    Ag[i] = A + row0 + 32 * ab + 4 * (lane & 7) + (size_t)(8 * kq + (lane >> 3)) * d;
  }
#pragma unroll
  for (int i = 0; i < PPB; ++i) {
    const int pb = w * PPB + i, bb = pb >> 2, kq = pb & 3;
    Bg[i] = B + col0 + 32 * bb + 4 * (lane & 7) + (size_t)(8 * kq + (lane >> 3)) * dP;
  }
  auto issue = [&](int t, int s) {
    float *dst = lds + s * STAGE_F;
#pragma unroll
    for (int i = 0; i < PPA; ++i) FB_GLDS16(Ag[i] + (size_t)(32 * t) * d, dst + (w * PPA + i) * 256);
#pragma unroll
    for (int i = 0; i < PPB; ++i) FB_GLDS16(Bg[i] + (size_t)(32 * t) * dP, dst + A_F + (w * PPB + i) * 256);
  };
  // this wave's 32 x 32 sub-tiles: (ri, cj) = global 32-blocks; active iff cj <= ri
  const int ri[2] = {(row0 >> 5) + 2 * wm, (row0 >> 5) + 2 * wm + 1}, cj[2] = {(col0 >> 5) + 2 * wn, (col0 >> 5) + 2 * wn + 1};
  bool on[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) on[i][j] = cj[j] <= ri[i];
  const bool rowon[2] = {on[0][0] || on[0][1], on[1][0] || on[1][1]}, colon[2] = {on[0][0] || on[1][0], on[0][1] || on[1][1]};
  const bool dg[2] = {(on[0][0] && ri[0] == cj[0]) || (on[0][1] && ri[0] == cj[1]), (on[1][0] && ri[1] == cj[0]) || (on[1][1] && ri[1] == cj[1])};   // row half i meets the diagonal here: d/dmu
  f32x16 acc[2][2], tot[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
  float rs[2][4];      // d/dmu: this lane's partial row sums of W, per row half and K quarter (k_fr_vjp32's rsum of wave q, half h)
  float rcur[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) rs[i][q] = 0.f;
  issue(0, 0);
  for (int t = 0; t < T; ++t) {
    fb_wait_vm0();
    lds_barrier();
    if (t + 1 < T) issue(t + 1, (t + 1) & 1);
    const float *cur = lds + (t & 1) * STAGE_F;
    if (t > 0 && t % nsub == 0) {   // a K quarter ended
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (!on[i][j]) continue;
          tot[i][j] += acc[i][j];
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
      const int q = t / nsub - 1;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq)
          if (qq == q) rs[i][qq] = rcur[i];
        rcur[i] = 0.f;
      }
    }
    if (!rowon[0] && !rowon[1]) continue;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      bf16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (!rowon[i]) continue;
        const float *ac = cur + (2 * wm + i) * 1024 + l31;
        float av[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) av[e] = ac[(8 * (2 * g + (e >> 2)) + 4 * h + (e & 3)) * 32];
        if (dg[i]) {
#pragma unroll
          for (int e = 0; e < 8; ++e) rcur[i] += av[e];
        }
        fb_split3(av, ah[i], am[i], al[i]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (!colon[j]) continue;
        const float *bc = cur + A_F + (2 * wn + j) * 1024 + l31;
        float bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = bc[(8 * (2 * g + (e >> 2)) + 4 * h + (e & 3)) * 32];
        fb_split3(bv, bh[j], bm[j], bl[j]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (on[i][j]) fb_mfma6(ah[i], am[i], al[i], bh[j], bm[j], bl[j], acc[i][j]);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (on[i][j]) tot[i][j] += acc[i][j];
    rs[i][3] = rcur[i];
  }
  lds_barrier();
  float *Cs = lds + w * (32 * LDC);
  const double invM = 1.0 / (double)a.M_total;
  const bool pow2M = (a.M_total & (a.M_total - 1)) == 0;
  const float invMf = (float)invM;
  const double direct = direct_entropy_coeff(a.ent_kind);
  const int i4 = 4 * (lane & 7);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (!on[i][j]) continue;
      const bool diag = ri[i] == cj[j];
      const int rbase = 32 * ri[i], cbase = 32 * cj[j];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = {tot[i][j][4 * q], tot[i][j][4 * q + 1], tot[i][j][4 * q + 2], tot[i][j][4 * q + 3]};
        *(f32x4 *)(Cs + l31 * LDC + 8 * q + 4 * h) = v;
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {   // k_fr_vjp32's epilogue thread (i4, n) = lane of pass p
        const int n = 8 * p + (lane >> 3);
        const int gi = rbase + i4, gj = cbase + n;
        float cjj = 1.f;
        if (diag && gj >= gi && gj < gi + 4) cjj = a.params[d + (size_t)gj * d + gj];
        const f32x4 v = *(const f32x4 *)(Cs + n * LDC + i4);
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = vjp_elem(v[c], gi + c, gj, pow2M, invMf, invM, direct, cjj);
        store16_wt(grad + d + (size_t)gj * d + gi, o);
      }
      if (!diag && upper) {   // the mirrored, strictly upper 32 x 32 block is structurally zero
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int ii = 8 * p + (lane >> 3);
          store16_wt(grad + d + (size_t)(rbase + ii) * d + cbase + i4, z4);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (dg[i]) {   // d/dmu rows of this row block: k_fr_vjp32 sums the eight (wave q, half h) partials in the order 2 q + h, in f64
      double sm = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float mine = rs[i][q], other = __shfl_xor(mine, 32, 64);
        const float r0 = h ? other : mine, r1 = h ? mine : other;
        sm += (double)r0;
        sm += (double)r1;
      }
      if (lane < 32) grad[32 * ri[i] + lane] = dmu_elem(sm, invM);
    }
  }
}

// -----------------------------------------------------------------------------------------------------------------
// k_fb_value: the objective values of the step's lanes (one workgroup each: finalize_value_block, the single calls' assembly)
// -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fb_value(FbArgs a) {
  __shared__ double red[4 * 4];
  const int l = blockIdx.x, d = a.d;
  ValueIn vin{};
  vin.ell_const = a.ell_const;
  vin.ell_part = a.ell_part + (size_t)l * a.ell_stride;
  vin.n_ell_part = (d >> 5) * (a.M >> 5);
  vin.he_part = a.he_part + (size_t)l * a.he_stride;
  vin.n_he_part = (d >> 6) * (a.M >> 5);
  vin.ld_part = a.ld_part;
  vin.n_ld_part = d >> 5;
  OutArgs out{};
  const bool last = l == a.lane_last && a.value_last;
  out.value = last ? a.value_last : a.values + (size_t)l * a.value_stride;
  out.ent_kind = a.ent_kind;
  out.M_total = a.M_total;
  out.M_local = a.M;
  out.status = a.status;
  const float *pp = a.params;
  finalize_value_block<float, 256, false>(d, vin, out, (int64_t)d + (int64_t)d * d, [pp, d](int i) { return pp[d + (size_t)i * d + i]; }, red);
}

// -----------------------------------------------------------------------------------------------------------------
// Host side
// -----------------------------------------------------------------------------------------------------------------
namespace {
constexpr int kBM = 128, kBN = 128;

void fb_upload(DevBuf &b, const void *src, size_t bytes) {
  if (b.bytes < bytes || !b.p) {
    if (b.p) (void)hipFree(b.p);
    (void)hipMalloc(&b.p, bytes);
    b.bytes = bytes;
  }
  (void)hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice);
}
}  // namespace

bool fb_shape_ok(const mivi_ctx *c, int M) {
  static const bool off = getenv("MIVI_BATCH_GEN3") && atoi(getenv("MIVI_BATCH_GEN3")) == 0;   // A/B: the lane-batched second-generation kernels
  return !off && c->cfg.family == MIVI_FULLRANK && c->cfg.dtype == MIVI_F32 && c->cfg.d % kBM == 0 && M % kBN == 0 && M % 128 == 0 &&
         c->cfg.d >= kBM && c->cfg.d <= 32768;
}

// work tables for L lanes: product tiles heaviest first, (lane, column block) panels dealt round-robin onto the XCDs (workgroup b runs on
// XCD b % 8: a panel's eps columns stay in one L2); VJP tiles in lane order, cut into eight equal runs
const FbTab *fb_prepare(mivi_ctx *c, int M, int L) {
  FbTables &ft = c->fb;
  for (FbTab &t : ft.tab)
    if (t.L == L && t.M == M && t.prod.p && t.vjp.p) return &t;
  FbTab &t = ft.tab[ft.next_tab];
  ft.next_tab = (ft.next_tab + 1) & 3;
  invalidate_graph(c);   // a captured graph bakes the table contents
  const int d = c->cfg.d, nrb = d / kBM, ncb = M / kBN;
  std::vector<std::vector<int4>> lists(8);
  int panel = 0;
  for (int l = 0; l < L; ++l)
    for (int cb = 0; cb < ncb; ++cb, ++panel)
      for (int rb = nrb - 1; rb >= 0; --rb) lists[panel % 8].push_back(make_int4(l, rb | (cb << 16), (l == 0 && cb == 0) ? 1 : 0, 0));
  while (true) {   // even the lists out: every workgroup index is real work
    int a = 0, b = 0;
    for (int x = 1; x < 8; ++x) {
      if (lists[x].size() > lists[a].size()) a = x;
      if (lists[x].size() < lists[b].size()) b = x;
    }
    if (lists[a].size() <= lists[b].size() + 1) break;
    lists[b].push_back(lists[a].back());
    lists[a].pop_back();
  }
  for (auto &li : lists) std::stable_sort(li.begin(), li.end(), [](const int4 &p, const int4 &q) { return (p.y & 0xffff) > (q.y & 0xffff); });
  std::vector<int4> prod;
  size_t mx = 0;
  for (auto &li : lists) mx = std::max(mx, li.size());
  for (size_t i = 0; i < mx; ++i)
    for (int x = 0; x < 8; ++x)
      if (i < lists[x].size()) prod.push_back(lists[x][i]);
  std::vector<int4> flat;
  for (int l = 0; l < L; ++l)
    for (int rb = 0; rb < nrb; ++rb)
      for (int cb = 0; cb <= rb; ++cb) flat.push_back(make_int4(l, rb | (cb << 16), 0, 0));
  std::vector<int4> vjp;
  {
    const size_t n = flat.size();
    size_t pos = 0;
    std::vector<std::vector<int4>> lx(8);
    for (int x = 0; x < 8; ++x) {
      const size_t e = (n * (x + 1)) / 8;
      for (; pos < e; ++pos) lx[x].push_back(flat[pos]);
    }
    size_t m2 = 0;
    for (auto &li : lx) m2 = std::max(m2, li.size());
    for (size_t i = 0; i < m2; ++i)
      for (int x = 0; x < 8; ++x)
        if (i < lx[x].size()) vjp.push_back(lx[x][i]);
  }
  fb_upload(t.prod, prod.data(), prod.size() * sizeof(int4));
  fb_upload(t.vjp, vjp.data(), vjp.size() * sizeof(int4));
  if (!t.prod.p || !t.vjp.p) return nullptr;
  t.n_prod = (int)prod.size();
  t.n_vjp = (int)vjp.size();
  t.L = L;
  t.M = M;
  return &t;
}

// one step of L estimates: eps -> product + target -> VJP -> values, on c->stream
void fb_launch_step(mivi_ctx *c, const FbStep &s) {
  FbTables &t = c->fb;
  const FbTab &tb = *s.tab;
  const int d = c->cfg.d, M = s.M, L = s.L;
  FbArgs a{};
  a.d = d; a.M = M; a.dP = c->dP; a.L = L;
  a.params = (const float *)s.params;
  a.t_mean = (const float *)c->t_mean.p;
  a.t_istd = (const float *)c->t_istd.p;
  a.eps = (float *)t.eps.p; a.eps_stride = (long long)c->dP * M;
  a.W = (float *)t.W.p; a.W_stride = (long long)d * M;
  a.ell_part = (double *)t.ell.p; a.ell_stride = (long long)(d / 32) * (M / 32);
  a.he_part = (double *)t.he.p; a.he_stride = (long long)(d / 64) * (M / 32);
  a.ld_part = (double *)t.ld.p;
  a.grads = (float *)s.grads; a.grad_stride = s.grad_stride;
  a.values = (float *)s.values; a.value_stride = s.value_stride;
  a.grad_last = (float *)s.grad_last; a.value_last = (float *)s.value_last; a.lane_last = s.lane_last;
  a.write_upper = s.write_upper;
  a.ent_kind = c->cfg.entropy; a.M_total = c->M_total;
  a.status = (int *)c->status.p;
  a.ell_const = c->t_const;
  a.rng = s.rng;
  hipLaunchKernelGGL(k_fb_eps, dim3((d / 64) * (M / 32), L), dim3(512), 0, c->stream, a);
  a.work = (const int4 *)tb.prod.p; a.n_work = tb.n_prod;
  hipLaunchKernelGGL((k_fb_prod<2, 2>), dim3(tb.n_prod), dim3(256), 0, c->stream, a);
  a.work = (const int4 *)tb.vjp.p; a.n_work = tb.n_vjp;
  hipLaunchKernelGGL((k_fb_vjp<2, 2>), dim3(tb.n_vjp), dim3(256), 0, c->stream, a);
  hipLaunchKernelGGL(k_fb_value, dim3(L), dim3(256), 0, c->stream, a);
}

}  // namespace mivi
