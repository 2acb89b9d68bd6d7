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
// The second generation gives every 32 x 32 tile of ONE estimate a workgroup whose waves split K: latency-bound launches, and every
// operand element is split into its three bf16 pieces by every tile that uses it (20 vector instructions per MFMA: the vector ALU is the
// busiest unit).  Here the launches are shaped like the large products they are, and the split is done ONCE, by whoever produces an operand:
//   * OPERAND PLANES.  Every operand lives in memory as its exact three-way bf16 split (hi / mid / lo planes, 6 bytes per element) in
//     MFMA-FRAGMENT ORDER: a fragment = 32 rows x 16 k of one operand = 3 planes x 64 lanes x 16 bytes, lane (row = lane % 32,
//     h = lane / 32) holding the eight k slots k = 16 g + 8 (e / 4) + 4 h + e % 4 -- the slot assignment of the second-generation kernels.
//     tril(C) is laid out once per call (k_fb_cplanes, diagonal blocks already masked), eps by its generator in both orientations
//     (k_fb_eps: rows as k for the product, samples as k for the VJP), W by the product's epilogue.  A main loop is then
//     LDS-DMA (1 KiB pieces) -> ds_read_b128 -> six MFMAs per fragment pair: no vector arithmetic at all.
//   * a workgroup owns a 128 x 128 output tile, a wave a 64 x 64 part of it (2 x 2 MFMA tiles) over the WHOLE K range; operands are
//     staged once per workgroup in a three-slot LDS ring of 16-k stages (the DMA of stage g + 2 is in flight under the MFMAs of g),
//     one barrier per stage; no cross-wave reduction: the epilogue works on a wave's own accumulators (transposed through a
//     wave-private LDS image so that stores are whole 128-byte lines);
//   * the launch covers every lane (estimate) of the step: per-lane buffers are base + lane * stride, the work table names (lane, tile).
// BIT-IDENTICAL to the one-estimate kernels (k_fr_prod32 / k_fr_vjp32): those cut a tile's K range into runs (one per wave: eight for the
// product, four for the VJP), every run an MFMA chain from zero, the runs summed in wave order.  A wave here walks the same runs one after
// the other -- chain accumulator `acc`, folded into `tot` at every run boundary (tot = tot + acc: the same f32 additions in the same
// order) -- on the same bf16 pieces (the same split arithmetic, applied by the producer instead of the consumer) in the same k slots,
// with the same per-element epilogue arithmetic (fr_elem.h), the same wave sums behind every ell partial and the same slots for them.
// So "a batch's estimates are bitwise the single calls'" holds by construction (tests/test_gpu_batches.py, tests/test_gpu_each.py).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "device_common.h"
#include "fr_elem.h"
#include "fr_planes.h"

namespace mivi {

// A wave's operands of one 16-k group: two A fragments (its two 32-row blocks) and WJ B fragments (its 32-column blocks), three planes each
template <int WJ>
struct FbFrags {
  u32x4v A[2][3], B[WJ][3];
};
__device__ __forceinline__ f32x16 fb_mma(const u32x4v &a, const u32x4v &b, const f32x16 &c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the 12 WJ MFMAs of a group, the 2 WJ accumulators' chains interleaved (a dependent MFMA issues 2 WJ slots behind its predecessor); per
// accumulator the order is mfma_bf16x3's: lo.hi, hi.lo, mid.mid, mid.hi, hi.mid, hi.hi  (planes: 0 hi, 1 mid, 2 lo)
template <int WJ>
__device__ __forceinline__ void fb_group(const FbFrags<WJ> &F, f32x16 (&acc)[2][WJ]) {
  constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
  for (int p = 0; p < 6; ++p)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < WJ; ++j) acc[i][j] = fb_mma(F.A[i][PA[p]], F.B[j][PB[p]], acc[i][j]);
}


struct FbArgs {
  int d, M, L;
  const float *params;            // [mu; vec C]
  const float *t_mean, *t_istd;   // diagonal-Gaussian target
  unsigned *CA;                   // planes of tril(C): fragment (rb32, kg), kg <= 2 rb32 + 1, at (rb32 (d / 16) + kg) kFrag
  unsigned *epsP;                 // lane l: epsP + l * plane_stride; fragment (mb32, kg = row group) at (mb32 (d / 16) + kg) kFrag
  unsigned *epsV;                 // lane l: epsV + l * plane_stride; fragment (jb32, mg = sample group) at (jb32 (M / 16) + mg) kFrag
  unsigned *WV;                   // lane l: WV + l * plane_stride; fragment (rb32, mg) at (rb32 (M / 16) + mg) kFrag
  // dense-Gaussian target: g = -P (z - m) is a second product per lane (k_fb_prod<MODE 2>) between the draw's product and the VJP
  const float *t_prec;            // P, leading dimension dP
  int dP;
  unsigned *PA;                   // planes of P: fragment (rb32, kg), every kg < d / 16, at (rb32 (d / 16) + kg) kFrag (k_fb_pplanes)
  unsigned *RP;                   // lane l: RP + l * plane_stride: R = Z - m as the second product's B operand, eps' product layout
  // sticking-the-landing estimators: W += C^-T eps with the INVERSE of the scale formed once per call (the parameters are fixed inside it)
  const float *Tinv;              // C^-T, d x d, (row i, column k) at [i + k d], upper triangular (the solve kernels on the identity)
  unsigned *TA;                   // its planes: fragment (rb32, kg), every kg, entries k < row zeroed (k_fb_tplanes)
  long long plane_stride;         // words per lane = d M / 512 * kFrag
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
  float *grad_last, *value_last;  // lane_last writes these instead (nullptr: every lane writes grads / values)
  int lane_last;
  int write_upper;                // 1: every lane writes the exact zeros above the diagonal; 0: only lane_last does (the others' buffers hold them already)
  int ent_kind, M_total;
  int *status;
  double ell_const;
  RngArgs rng;                    // lane l draws estimate rng_index(rng) + l
  int knock;                      // developer knock-outs (-DMIVI_DEV builds only: tools/ubench_fb.hip)
  long long *dbg;                 // developer timeline (-DMIVI_DEV): per workgroup {hw id | xcc << 32, start, main loop done, end}
};
#ifdef MIVI_DEV
#define FB_STAMP(a, slot) do { if ((a).dbg && threadIdx.x == 0 && blockIdx.x == 0 && ((slot) == 1 || (slot) == 2)) (a).dbg[8 * 4096 + (slot)] = (long long)clock64(); \
  if ((a).dbg && threadIdx.x == 0) (a).dbg[(size_t)blockIdx.x * 8 + (slot)] = (slot) ? (long long)wall_clock64() : \
  (long long)(__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32)); } while (0)
#else
#define FB_STAMP(a, slot) do { } while (0)
#endif

// -----------------------------------------------------------------------------------------------------------------
// k_fb_cplanes: tril(C) as operand planes, one wave per fragment (rb32, kg): lane (row, h) reads its eight k slots (coalesced over the
// 32 rows), zeroes the entries above the diagonal, splits, stores 3 x 16 bytes.  Once per call: the parameters are fixed inside it.
// -----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fb_cplanes_frag(const FbArgs &a, int f, int lane) {
  const int l31 = lane & 31, h = lane >> 5;
  const int d = a.d, ng = d >> 4;
  const int rb = f / ng, kg = f % ng;
  if (rb >= (d >> 5) || kg > 2 * rb + 3) return;   // (the two groups behind the diagonal block: zero fragments -- a wave of k_fb_prod walks the
  const int row = 32 * rb + l31;                   //  K range of its SECOND row block with both, the first one's chain then adds exact zeros)
  const float *C = a.params + d;
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 16 * kg + 8 * (e >> 2) + 4 * h + (e & 3);
    const float v = C[(size_t)k * d + row];
    x[e] = k > row ? 0.f : v;
  }
  u32x4v uh, um, ul;
  fb_split3(x, uh, um, ul);
  unsigned *dst = a.CA + (size_t)f * kFrag + 4 * lane;
  store16_wt(dst, uh);
  store16_wt(dst + 256, um);
  store16_wt(dst + 512, ul);
}
__global__ __launch_bounds__(256) void k_fb_cplanes(FbArgs a) {   // (stand-alone form: tools/ubench_fb.hip)
  fb_cplanes_frag(a, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

// k_fb_pplanes: the dense-Gaussian target's precision matrix P as operand planes (every k group: P is full), one wave per fragment.
// Once per target (the planes are kept until the target changes).
__global__ __launch_bounds__(256) void k_fb_pplanes(FbArgs a) {
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  const int d = a.d, ng = d >> 4;
  const int rb = f / ng, kg = f % ng;
  if (rb >= (d >> 5)) return;
  const int row = 32 * rb + l31;
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = a.t_prec[(size_t)(16 * kg + 8 * (e >> 2) + 4 * h + (e & 3)) * a.dP + row];   // (k_fr_prod32<G_DENSE>: A[row + k lda])
  u32x4v uh, um, ul;
  fb_split3(x, uh, um, ul);
  unsigned *dst = a.PA + (size_t)f * kFrag + 4 * lane;
  store16_wt(dst, uh);
  store16_wt(dst + 256, um);
  store16_wt(dst + 512, ul);
}

// k_fb_tplanes: C^-T (upper triangular) as operand planes, one wave per fragment; the entries below the diagonal are exact zeros whatever
// the solve left there.  Once per call.
__global__ __launch_bounds__(256) void k_fb_tplanes(FbArgs a) {
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  const int d = a.d, ng = d >> 4;
  const int rb = f / ng, kg = f % ng;
  if (rb >= (d >> 5)) return;
  const int row = 32 * rb + l31;
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 16 * kg + 8 * (e >> 2) + 4 * h + (e & 3);
    x[e] = (kg >= 2 * rb && k >= row) ? a.Tinv[(size_t)k * d + row] : 0.f;
  }
  u32x4v uh, um, ul;
  fb_split3(x, uh, um, ul);
  unsigned *dst = a.TA + (size_t)f * kFrag + 4 * lane;
  store16_wt(dst, uh);
  store16_wt(dst + 256, um);
  store16_wt(dst + 512, ul);
}

// -----------------------------------------------------------------------------------------------------------------
// k_fb_eps: eps of L estimates as operand planes in both orientations.  Draws: the blocks of the product kernels' riders (64 rows x 32
// columns, one Philox block per thread: the same stream and the same he_part partials as k_eps_m / the riders of k_fr_prod32); the block's
// 64 x 32 values go through an LDS tile, threads 0..255 then assemble the four product fragments (column = row of the B operand, k = rows
// 16 ig ..), threads 256..511 the four VJP fragments (row j, k = samples 16 mg ..).  blockIdx.y = lane.
// -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_fb_eps(FbArgs a) {
  __shared__ double red[8];
  __shared__ float E[32 * 65];   // E[m][i], leading dimension 65
  const int tid = threadIdx.x, eb = blockIdx.x, l = blockIdx.y, d = a.d;
  if (l >= a.L) {   // riders of a call's first draw: tril(C) as operand planes (parameters only), eight fragments per workgroup
    fb_cplanes_frag(a, (((int)blockIdx.y - a.L) * (int)gridDim.x + eb) * 8 + (tid >> 6), tid & 63);
    return;
  }
  PlaneEps pe{};
  pe.d = d; pe.M = a.M;
  pe.seed = a.rng.seed; pe.idx = rng_index(a.rng) + (uint64_t)l; pe.m_offset = a.rng.m_offset;
  pe.epsP = a.epsP + (size_t)l * a.plane_stride;
  pe.epsV = a.epsV + (size_t)l * a.plane_stride;
  pe.he_part = a.he_part + (size_t)l * a.he_stride;
  plane_eps_block(pe, eb, E, red);
}

// -----------------------------------------------------------------------------------------------------------------
// The two products share a staging scheme: a 128 x 128 tile, four waves (2 x 2), 16-k stages of 24 KiB (A: four fragments, B: four
// fragments, three planes each) in a three-slot LDS ring.  Piece pc of a stage (1 KiB): pc < 12: A fragment pc / 3, plane pc % 3; else B.
// Wave w issues pieces 6 w .. 6 w + 5: always six requests per wave and stage, so the vmcnt accounting is a compile-time constant.
// -----------------------------------------------------------------------------------------------------------------
constexpr int kStageW = 24 * 256;   // words per stage
constexpr int kRing = 4;            // LDS ring slots (96 KiB: ONE workgroup per CU): three stages in flight behind the one being read;
                                    // the main loops are unrolled by kRing, so every slot address is a compile-time constant
// Wave layout of a 128 x 128 tile, WJ = 32-column blocks per wave: 8 / WJ waves = 2 (row halves of 64) x 4 / WJ (column parts of 32 WJ).
//   WJ = 2: four waves (one per SIMD, up to 512 registers each), a wave owns 64 x 64: 48 KiB of LDS reads per group and workgroup
//   WJ = 1: eight waves (two per SIMD), a wave owns 64 x 32: 72 KiB of LDS reads per group -- the LDS read port then paces the loop
template <int WJ>
__device__ __forceinline__ void fb_read_frags(const unsigned *lds, int slot, int wm, int wn, int lane, FbFrags<WJ> &F) {
  const unsigned *cur = lds + slot * kStageW + 4 * lane;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int p = 0; p < 3; ++p) F.A[i][p] = *(const u32x4v *)(cur + ((2 * wm + i) * 3 + p) * 256);
#pragma unroll
  for (int j = 0; j < WJ; ++j)
#pragma unroll
    for (int p = 0; p < 3; ++p) F.B[j][p] = *(const u32x4v *)(cur + (12 + (WJ * wn + j) * 3 + p) * 256);
}
// The issue order inside one iteration's straight-line block {3 WJ LDS-DMA requests, 9 WJ.. fragment reads of the next group, 12 WJ MFMAs}:
// every wave of the workgroup leaves the barrier at the same moment, and with the reads first (where the scheduler puts loads) all of
// them queue on the LDS port before the first MFMA of anybody issues -- the matrix pipe idles for the length of that burst.  One memory
// operation behind every MFMA instead: the pipe starts at once, the reads trickle in under it.
template <int WJ>
__device__ __forceinline__ void fb_sched_interleave() {
  constexpr int NM = 12 * WJ, ND = 6 + 3 * WJ, NV = 3 * WJ;
#pragma unroll
  for (int k = 0; k < NM; ++k) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                  // one MFMA
    if (k < ND) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      // one LDS read
    else if (k < ND + NV) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);            // one LDS-DMA request
  }
}
using FbI0 = std::integral_constant<int, 0>;
using FbI1 = std::integral_constant<int, 1>;
using FbI2 = std::integral_constant<int, 2>;
using FbI3 = std::integral_constant<int, 3>;
using FbT = std::true_type;
using FbN = std::false_type;

// -----------------------------------------------------------------------------------------------------------------
// k_fb_prod: W_l = grad log pi(mu + tril(C) eps_l) as VJP operand planes + ell partials, for every lane l of the step.
// Work item = (lane, rb, cb): rows [128 rb, +128) of columns [128 cb, +128) of lane l.  A 32-row block r32 has 2 (r32 + 1) groups;
// k_fr_prod32's runs of a row block with nst sub-stages are chunks of ceil(nst / 8) sub-stages.
// Software pipeline: iteration g computes on the fragments of stage g (already in registers) while the fragments of stage g + 1 are read
// from LDS and the DMA of stage g + kRing is issued.  Per iteration: wait for the own pieces of stage g + 1, barrier (stage g + 1 has landed
// for every wave; every wave has read stage g, whose slot stage g + kRing takes), issue, read, 12 WJ MFMAs.
// -----------------------------------------------------------------------------------------------------------------
// PF = 1: fragments of the next group prefetched into registers (ONE workgroup per CU, four ring slots); PF = 0: no register prefetch, three
// ring slots, at most 128 registers: TWO workgroups per CU cover each other's barriers, read latencies, prologues and epilogues.
// MODE (k_fr_prod32's epilogue modes): FB_DIAG: the fused diagonal-Gaussian target, W planes + ell partials (R_DIAG);
//   FB_DENSE_R: R = (mu + tril(C) eps) - m as the B-operand planes of the dense target's product (R_DENSE_R);
//   FB_DENSE_G: the dense target's product itself, G = -P R: A = the planes of P over the WHOLE K range (every row block d / 32 sub-stages:
//   runs of ceil(d / 256) for all of them), B = R's planes, epilogue g = -(P r), ell += r g / 2 (R_DENSE_G), W planes + ell partials.
//   FB_STL_U: the sticking-the-landing term, W += C^-T eps: A = the planes of C^-T (upper triangular: a tile's K range starts at its first
//   row and runs to the end), B = eps' planes, epilogue: the lane's W planes read back, + U, split and stored again.  No counterpart among
//   the one-estimate kernels (they SOLVE C^T X = eps, kernels_stl.hip): one chain over the tile's K range, results equal to the solve's to
//   rounding (tests/test_gpu_each.py states the tolerance).
enum { FB_DIAG = 0, FB_DENSE_R = 1, FB_DENSE_G = 2, FB_STL_U = 3 };
template <int WJ, int PF, int MODE>
__global__ __launch_bounds__(512 / WJ, PF ? 2 / WJ : 4 / WJ) void k_fb_prod(FbArgs a) {
  constexpr bool kDG = MODE == FB_DENSE_G, kSU = MODE == FB_STL_U;
  constexpr int LDC = 36, NF = WJ, kPW = 3 * WJ;   // fragments / pieces this wave stages per group
  constexpr int NR = PF ? kRing : 3;
  __shared__ __attribute__((aligned(16))) unsigned lds[NR * kStageW + 3 * 128];
  float *vec = reinterpret_cast<float *>(lds + NR * kStageW);   // mu, target mean, target 1 / std of the tile's rows
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / (4 / WJ), wn = w % (4 / WJ);
  const __attribute__((address_space(4))) int *wp = (const __attribute__((address_space(4))) int *)a.work + 4 * blockIdx.x;
  const int ln = wp[0], rc = wp[1], flags = wp[2];
  const int rb = rc & 0xffff, cb = rc >> 16;
  const int d = a.d, ng = d >> 4;
  const int row0 = rb * 128, col0 = cb * 128;
  const int R0 = row0 >> 5;               // first 32-row block of the tile
  const int g0 = kSU ? 2 * R0 : 0;        // first group of the K range
  const int G = kDG ? ng : (kSU ? ng - g0 : 2 * (R0 + 4));  // groups of the workgroup (the last row block's K; the dense product: all of K)
  if (tid < 128 && !kDG && !kSU) {
    vec[tid] = a.params[row0 + tid];
    vec[128 + tid] = a.t_mean[row0 + tid];
    if (MODE == FB_DIAG) vec[256 + tid] = a.t_istd[row0 + tid];
  }
  // this wave's NF fragments of a stage (fragment f of the stage: f < 4: A fragment f; else B fragment f - 4; three 1 KiB pieces each)
  const unsigned *sp[NF];   // the stage the next issue takes
  int gmax[NF];             // last group of the fragment that is ever read (tril(C): the diagonal block + two zero groups: clamped beyond)
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int fs = NF * w + f, fr = fs & 3;
    if (fs < 4) {
      sp[f] = (kDG ? a.PA : (kSU ? a.TA : a.CA)) + ((size_t)(R0 + fr) * ng + g0) * kFrag + 4 * lane;
      gmax[f] = (!kDG && !kSU && 2 * (R0 + fr) + 3 < G - 1) ? 2 * (R0 + fr) + 3 : G - 1;
    } else {
      sp[f] = (kDG ? a.RP : a.epsP) + (size_t)ln * a.plane_stride + ((size_t)((col0 >> 5) + fr) * ng + g0) * kFrag + 4 * lane;
      gmax[f] = G - 1;
    }
  }
  int gd = 0;               // the group the pointers stand at
  auto issue = [&](int slot) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      unsigned *dst = lds + slot * kStageW + (NF * w + f) * 768;
      FB_GLDS16(sp[f], dst, 0);
      FB_GLDS16(sp[f], dst, 1024);
      FB_GLDS16(sp[f], dst, 2048);
      sp[f] += gd < gmax[f] ? kFrag : 0;
    }
    ++gd;
  };
  f32x16 acc[2][WJ], tot[2][WJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
  const int r32[2] = {R0 + 2 * wm, R0 + 2 * wm + 1};
  // k_fr_prod32's runs are chunks of ceil(nst / 8) sub-stages -- the same chunk for this wave's two row blocks (nst = r32[1] and r32[1] + 1,
  // the first odd): ONE fold schedule, every 2 rc groups
  const int rc2 = kSU ? (1 << 30) : (kDG ? 2 * (((d >> 5) + 7) >> 3) : 2 * ((r32[1] + 1 + 7) >> 3));
  FB_STAMP(a, 0);
  FB_STAMP(a, 1);
  // A wave computes groups 0 .. Gw - 1 (its second row block's K range: a multiple of four groups) with BOTH row blocks, unconditionally:
  // one straight MFMA block per group (a choice between a full and a half group per iteration made the compiler copy the accumulators
  // behind every group, i.e. wait for the matrix pipe to drain).  The first row block ends two groups earlier: k_fb_cplanes laid two zero
  // fragments behind its diagonal block, so its chain adds exact zeros there.
  const int Gw = (kDG || kSU) ? G : 2 * r32[1] + 2;
  int gfold = rc2;   // the next run boundary (even: checked on even groups only)
  auto compute = [&](auto S, int g, const FbFrags<WJ> &F) {   // S = g mod kRing
    if (!MIVI_KNOCKED(a, 2)) fb_group<WJ>(F, acc);
    if constexpr (PF) fb_sched_interleave<WJ>();
    if constexpr ((decltype(S)::value & 1) == 1) {   // (behind the MFMAs: the block in front of them stays one basic block; S = -1: every group)
      if (__builtin_expect(g + 1 == gfold, 0)) {     // a run ended with this group
        gfold += rc2;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < WJ; ++j) {
            tot[i][j] += acc[i][j];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
          }
        asm volatile("" ::: "memory");
      }
    }
  };
  // iteration g (S = g mod kRing): wait for the own pieces of stage g + 1, barrier, issue stage g + kRing into the slot of stage g, read the
  // fragments of stage g + 1, compute stage g.  VM = stages that may stay in flight across the wait; CMP: this wave still has work.
  auto step = [&](auto S, auto VM, auto ISS, auto CMP, int g, const FbFrags<WJ> &Fc, FbFrags<WJ> &Fn) {
    constexpr int sl = decltype(S)::value;
    fb_wait_vm<kPW * decltype(VM)::value>();
    if (!MIVI_KNOCKED(a, 64)) fb_barrier();
    if constexpr (decltype(ISS)::value) {
      if (!MIVI_KNOCKED(a, 1)) issue(sl);
    }
    if constexpr (decltype(CMP)::value) {
      if (!MIVI_KNOCKED(a, 32)) fb_read_frags<WJ>(lds, MIVI_KNOCKED(a, 4) ? 0 : (sl + 1) % kRing, wm, wn, lane, Fn);
      compute(S, g, Fc);
    }
  };
  if constexpr (PF) {
  static_assert(kRing == 4, "the unrolled loops below are written for four slots");
  issue(0); issue(1); issue(2); issue(3);   // (G >= 8)
  fb_wait_vm<kPW * 3>();
  fb_barrier();
  FbFrags<WJ> F0, F1;
  fb_read_frags<WJ>(lds, 0, wm, wn, lane, F0);
  int g = 0;
  for (; g + 4 < G; g += 4) {   // (G is a multiple of eight; Gw = G for the waves of the tile's lower half, G - 4 for the upper half's)
    step(FbI0{}, FbI2{}, FbT{}, FbT{}, g, F0, F1);
    step(FbI1{}, FbI2{}, FbT{}, FbT{}, g + 1, F1, F0);
    step(FbI2{}, FbI2{}, FbT{}, FbT{}, g + 2, F0, F1);
    step(FbI3{}, FbI2{}, FbT{}, FbT{}, g + 3, F1, F0);
  }
  if (g < Gw) {   // the last four groups: nothing left to request
    step(FbI0{}, FbI2{}, FbN{}, FbT{}, g, F0, F1);
    step(FbI1{}, FbI1{}, FbN{}, FbT{}, g + 1, F1, F0);
    step(FbI2{}, FbI0{}, FbN{}, FbT{}, g + 2, F0, F1);
    compute(FbI3{}, g + 3, F1);
  } else {        // (the upper half's K range has ended: its waves only keep the barriers)
    step(FbI0{}, FbI2{}, FbN{}, FbN{}, g, F0, F1);
    step(FbI1{}, FbI1{}, FbN{}, FbN{}, g + 1, F1, F0);
    step(FbI2{}, FbI0{}, FbN{}, FbN{}, g + 2, F0, F1);
  }
  } else {
    // plain loop: wait for stage g, barrier, request stage g + 2 into the slot of stage g - 1, read, compute -- the other workgroup of the CU
    // runs its MFMAs under this one's waits
    issue(0); issue(1);
    int slot = 0;
    for (int g = 0; g < G; ++g) {
      if (g + 1 < G) fb_wait_vm<kPW>();
      else fb_wait_vm<0>();
      if (!MIVI_KNOCKED(a, 64)) fb_barrier();
      if (g + 2 < G && !MIVI_KNOCKED(a, 1)) issue(slot == 0 ? 2 : slot - 1);
      if (g < Gw) {
        FbFrags<WJ> F;
        fb_read_frags<WJ>(lds, slot, wm, wn, lane, F);
        compute(std::integral_constant<int, -1>{}, g, F);
      }
      slot = slot == 2 ? 0 : slot + 1;
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WJ; ++j) tot[i][j] += acc[i][j];   // the last run
  fb_barrier();   // every wave is done with the ring: LDS becomes the waves' private epilogue images
  FB_STAMP(a, 2);
  if (MIVI_KNOCKED(a, 16)) { if (tot[0][0][0] == 123.f) a.ld_part[0] = tot[1][WJ - 1][3] + tot[0][0][2]; return; }
  float *Cs = reinterpret_cast<float *>(lds) + w * (32 * LDC);
  unsigned *WVl = a.WV + (size_t)ln * a.plane_stride;
  unsigned *RPl = a.RP + (size_t)ln * a.plane_stride;
  double *ellp = a.ell_part + (size_t)ln * a.ell_stride;
  const int nrb = d >> 5, ncb = a.M >> 5, nmg = a.M >> 4;
  const bool xcd_slots = (nrb & 3) == 0 && (ncb & 1) == 0;   // (k_fr_prod32's block -> tile map)
  const int ei4 = 4 * (lane & 7);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lr = 64 * wm + 32 * i;   // row offset inside the tile
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, tm = mu, tis = mu;
    if constexpr (kSU) {
      // U's 32 x 32 tile (image [sample][row]) added to this wave's two W fragments in place
#pragma unroll
      for (int j = 0; j < WJ; ++j) {
        const int cb32 = (col0 >> 5) + WJ * wn + j;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {tot[i][j][4 * q], tot[i][j][4 * q + 1], tot[i][j][4 * q + 2], tot[i][j][4 * q + 3]};
          *(f32x4 *)(Cs + l31 * LDC + 8 * q + 4 * h) = v;
        }
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          unsigned *dst = WVl + ((size_t)r32[i] * nmg + 2 * cb32 + g2) * kFrag + 4 * lane;
          const u32x4v wh = *(const u32x4v *)dst, wmid = *(const u32x4v *)(dst + 256), wl = *(const u32x4v *)(dst + 512);
          float x[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = fb_unsplit(wh, wmid, wl, e) + Cs[(16 * g2 + 8 * (e >> 2) + 4 * h + (e & 3)) * LDC + l31];
          u32x4v uh, um, ul;
          fb_split3(x, uh, um, ul);
          store16_wt(dst, uh);
          store16_wt(dst + 256, um);
          store16_wt(dst + 512, ul);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      continue;
    }
    if constexpr (!kDG) {
      mu = *(const f32x4 *)(vec + lr + ei4);
      tm = *(const f32x4 *)(vec + 128 + lr + ei4);
      if constexpr (MODE == FB_DIAG) tis = *(const f32x4 *)(vec + 256 + lr + ei4);
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int cb32 = (col0 >> 5) + WJ * wn + j;
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
        float ell = 0.f;
        f32x4 wv;
        if constexpr (MODE == FB_DIAG) {
          const f32x4 z = mu + v;
#pragma unroll
          for (int c = 0; c < 4; ++c) wv[c] = diag_target_elem(z[c], tm[c], tis[c], ell);
        } else if constexpr (MODE == FB_DENSE_R) {
          const f32x4 z = mu + v;
          wv = z - tm;
        } else {
          // r = (z - m)[rows ei4 .. + 3, column en] back from R's planes, exactly (hi + mid + lo; the pieces do not overlap): fragment
          // (cb32, kg = 2 r32 + ei4 / 16), lane (en, h' = ei4 / 4 % 2), slots 4 (ei4 / 8 % 2) + c = the two words 2 (ei4 / 8 % 2) + {0, 1}
          const unsigned *fr = RPl + ((size_t)cb32 * ng + 2 * r32[i] + (ei4 >> 4)) * kFrag + 4 * (en + 32 * ((ei4 >> 2) & 1)) + 2 * ((ei4 >> 3) & 1);
          const uint2 qh = *(const uint2 *)fr, qm = *(const uint2 *)(fr + 256), ql = *(const uint2 *)(fr + 512);
          const unsigned uh[2] = {qh.x, qh.y}, um[2] = {qm.x, qm.y}, ul[2] = {ql.x, ql.y};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const unsigned hh = (c & 1) ? (uh[c >> 1] & 0xFFFF0000u) : (uh[c >> 1] << 16), mm = (c & 1) ? (um[c >> 1] & 0xFFFF0000u) : (um[c >> 1] << 16),
                           ll = (c & 1) ? (ul[c >> 1] & 0xFFFF0000u) : (ul[c >> 1] << 16);
            const float r = (__builtin_bit_cast(float, hh) + __builtin_bit_cast(float, mm)) + __builtin_bit_cast(float, ll);
            wv[c] = dense_target_elem(v[c], r, ell);
          }
        }
        *(f32x4 *)(Cs + en * LDC + ei4) = wv;   // the image becomes W[m][i] (FB_DENSE_R: R[m][i])
        if constexpr (MODE != FB_DENSE_R) {
          const double sv = (double)wave_sum_f32(ell);
          s = p ? s + sv : sv;
        }
      }
      if constexpr (MODE != FB_DENSE_R) {
        s += 0.0;   // (k_fr_prod32 adds its four idle waves' zeros: -0.0 becomes +0.0 there)
        if (lane == 0) {
          const int rE = nrb - 1 - r32[i];
          const int slot = xcd_slots ? ((rE & 3) + 4 * (cb32 & 1)) + 8 * ((rE >> 2) * (ncb >> 1) + (cb32 >> 1)) : rE * ncb + cb32;
          ellp[slot] = s;
        }
      }
      if constexpr (MODE == FB_DENSE_R) {
        // R as the dense product's B fragments (mb32 = cb32, kg = 2 r32 + g2): lane (column l31, h), slots = rows 16 g2 + 8 (e / 4) + 4 h + e % 4
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          const f32x4 x0 = *(const f32x4 *)(Cs + l31 * LDC + 16 * g2 + 4 * h), x1 = *(const f32x4 *)(Cs + l31 * LDC + 16 * g2 + 8 + 4 * h);
          const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
          u32x4v uh, um, ul;
          fb_split3(x, uh, um, ul);
          unsigned *dst = RPl + ((size_t)cb32 * ng + 2 * r32[i] + g2) * kFrag + 4 * lane;
          store16_wt(dst, uh);
          store16_wt(dst + 256, um);
          store16_wt(dst + 512, ul);
        }
      } else {
        // W as the VJP's A fragments (rb32 = r32[i], mg = 2 cb32 + g2): lane (row l31, h), slots = samples 16 g2 + 8 (e / 4) + 4 h + e % 4
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          float x[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = Cs[(16 * g2 + 8 * (e >> 2) + 4 * h + (e & 3)) * LDC + l31];
          u32x4v uh, um, ul;
          fb_split3(x, uh, um, ul);
          unsigned *dst = WVl + ((size_t)r32[i] * nmg + 2 * cb32 + g2) * kFrag + 4 * lane;
          store16_wt(dst, uh);
          store16_wt(dst + 256, um);
          store16_wt(dst + 512, ul);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the image is read before the next tile overwrites it
    }
  }
  FB_STAMP(a, 3);
  if (!kDG && !kSU && (flags & 1) && wn == 0 && lane < 32) {   // log|det C| partials of this wave's two row blocks (lane 0's first column block carries the flag)
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
// The objective value of lane l, by 256 threads (finalize_value_block: the single calls' assembly)
// -----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fb_value_block(const FbArgs &a, int l, double *red) {
  const int d = a.d;
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
__global__ __launch_bounds__(256) void k_fb_value(FbArgs a) {   // (stand-alone form: tools/ubench_fb.hip)
  __shared__ double red[4 * 4];
  fb_value_block(a, blockIdx.x, red);
}

// -----------------------------------------------------------------------------------------------------------------
// k_fb_vjp: dC_l = -(1/M) tril(W_l eps_l') - direct diag(1 / C_ii), dmu_l = -(1/M) W_l 1, for every lane l of the step.
// Work item = (lane, rb, cb), cb <= rb: the 128 x 128 tile of the lower triangle; K = M samples = M / 16 groups.
// k_fr_vjp32's four runs = the K quarters; a wave whose sub-tiles all lie strictly above the diagonal only carries its share of the
// staging, the exact zeros of the upper triangle are written as the mirror images of the strictly lower 32 x 32 blocks (lanes with the
// write_upper duty).
// -----------------------------------------------------------------------------------------------------------------
template <int WJ, int PF>
__global__ __launch_bounds__(512 / WJ, PF ? 2 / WJ : 4 / WJ) void k_fb_vjp(FbArgs a) {
  constexpr int LDC = 36, NF = WJ, kPW = 3 * WJ;
  constexpr int NR = PF ? kRing : 3;
  __shared__ __attribute__((aligned(16))) unsigned lds[NR * kStageW];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / (4 / WJ), wn = w % (4 / WJ);
  if ((int)blockIdx.x >= a.n_work) {   // the objective values of the step's lanes: everything they sum is older than this launch
    if (tid < 256) fb_value_block(a, (int)blockIdx.x - a.n_work, reinterpret_cast<double *>(lds));
    return;
  }
  const __attribute__((address_space(4))) int *wp = (const __attribute__((address_space(4))) int *)a.work + 4 * blockIdx.x;
  const int ln = wp[0], rc = wp[1];
  const int rb = rc & 0xffff, cb = rc >> 16;
  const int d = a.d, M = a.M, nmg = M >> 4;
  const int row0 = rb * 128, col0 = cb * 128;
  const bool last = ln == a.lane_last && a.grad_last;
  float *grad = last ? a.grad_last : a.grads + (size_t)ln * a.grad_stride;
  const bool upper = a.write_upper || last;
  const int G = nmg, gq = G >> 2;   // groups; per K quarter
  const unsigned *sp[NF];           // the stage the next issue takes
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int fs = NF * w + f, fr = fs & 3;
    sp[f] = (fs < 4 ? a.WV + (size_t)ln * a.plane_stride + ((size_t)((row0 >> 5) + fr) * nmg) * kFrag
                    : a.epsV + (size_t)ln * a.plane_stride + ((size_t)((col0 >> 5) + fr) * nmg) * kFrag) + 4 * lane;
  }
  auto issue = [&](int slot) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      unsigned *dst = lds + slot * kStageW + (NF * w + f) * 768;
      FB_GLDS16(sp[f], dst, 0);
      FB_GLDS16(sp[f], dst, 1024);
      FB_GLDS16(sp[f], dst, 2048);
      sp[f] += kFrag;
    }
  };
  // this wave's 32 x 32 sub-tiles: (ri[i], cj[j]) = global 32-blocks; stored iff cj <= ri
  const int ri[2] = {(row0 >> 5) + 2 * wm, (row0 >> 5) + 2 * wm + 1};
  int cj[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) cj[j] = (col0 >> 5) + WJ * wn + j;
  const bool work = cj[0] <= ri[1];   // any sub-tile in the lower triangle
  bool dg[2];                         // row block i meets the diagonal in this wave: d/dmu
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    dg[i] = false;
#pragma unroll
    for (int j = 0; j < WJ; ++j) dg[i] = dg[i] || ri[i] == cj[j];
  }
  f32x16 acc[2][WJ], tot[2][WJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
  float rs[2][4];      // d/dmu: this lane's partial row sums of W, per row block and K quarter (k_fr_vjp32's rsum of wave q, half h)
  float rcur[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) rs[i][q] = 0.f;
  FB_STAMP(a, 0);
  FB_STAMP(a, 1);
  int gfold = gq;   // the next K-quarter boundary (gq is even: checked on even groups only)
  // A wave with work computes ALL its sub-tiles in every group (one straight MFMA block: see k_fb_prod); a sub-tile strictly above the
  // diagonal (diagonal tiles only) is simply not stored.
  auto compute = [&](auto S, int g, const FbFrags<WJ> &F) {
    if (!MIVI_KNOCKED(a, 2)) fb_group<WJ>(F, acc);
    if constexpr (PF) fb_sched_interleave<WJ>();
    if (__builtin_expect(dg[0] || dg[1], 0)) {   // (two waves of a diagonal tile; behind the MFMAs: the block in front of them stays one basic block)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if (dg[i]) {
#pragma unroll
          for (int e = 0; e < 8; ++e) rcur[i] += fb_unsplit(F.A[i][0], F.A[i][1], F.A[i][2], e);
        }
    }
    if constexpr ((decltype(S)::value & 1) == 1) {
      if (__builtin_expect(g + 1 == gfold && g + 1 < G, 0)) {   // a K quarter (but the last) ended with this group
        gfold += gq;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int j = 0; j < WJ; ++j) {
            tot[i][j] += acc[i][j];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
          }
          rs[i][3] = rs[i][2]; rs[i][2] = rs[i][1]; rs[i][1] = rs[i][0]; rs[i][0] = rcur[i];   // (newest first: re-ordered at the end)
          rcur[i] = 0.f;
        }
        asm volatile("" ::: "memory");
      }
    }
  };
  auto step = [&](auto S, auto VM, auto ISS, auto CMP, int g, const FbFrags<WJ> &Fc, FbFrags<WJ> &Fn) {   // (as in k_fb_prod)
    constexpr int sl = decltype(S)::value;
    fb_wait_vm<kPW * decltype(VM)::value>();
    if (!MIVI_KNOCKED(a, 64)) fb_barrier();
    if constexpr (decltype(ISS)::value) {
      if (!MIVI_KNOCKED(a, 1)) issue(sl);
    }
    if constexpr (decltype(CMP)::value) {
      if (!MIVI_KNOCKED(a, 32)) fb_read_frags<WJ>(lds, MIVI_KNOCKED(a, 4) ? 0 : (sl + 1) % kRing, wm, wn, lane, Fn);
      compute(S, g, Fc);
    }
  };
  if constexpr (PF) {
  static_assert(kRing == 4, "the unrolled loops below are written for four slots");
  issue(0); issue(1); issue(2); issue(3);   // (G >= 8: M >= 128)
  fb_wait_vm<kPW * 3>();
  fb_barrier();
  FbFrags<WJ> F0, F1;
  fb_read_frags<WJ>(lds, 0, wm, wn, lane, F0);
  if (work) {
    int g = 0;
    for (; g + 4 < G; g += 4) {   // (G is a multiple of 8)
      step(FbI0{}, FbI2{}, FbT{}, FbT{}, g, F0, F1);
      step(FbI1{}, FbI2{}, FbT{}, FbT{}, g + 1, F1, F0);
      step(FbI2{}, FbI2{}, FbT{}, FbT{}, g + 2, F0, F1);
      step(FbI3{}, FbI2{}, FbT{}, FbT{}, g + 3, F1, F0);
    }
    step(FbI0{}, FbI2{}, FbN{}, FbT{}, g, F0, F1);
    step(FbI1{}, FbI1{}, FbN{}, FbT{}, g + 1, F1, F0);
    step(FbI2{}, FbI0{}, FbN{}, FbT{}, g + 2, F0, F1);
    compute(FbI3{}, g + 3, F1);
  } else {      // a wave above the diagonal: it only carries its share of the staging
    int g = 0;
    for (; g + 4 < G; g += 4) {
      step(FbI0{}, FbI2{}, FbT{}, FbN{}, g, F0, F1);
      step(FbI1{}, FbI2{}, FbT{}, FbN{}, g + 1, F1, F0);
      step(FbI2{}, FbI2{}, FbT{}, FbN{}, g + 2, F0, F1);
      step(FbI3{}, FbI2{}, FbT{}, FbN{}, g + 3, F1, F0);
    }
    step(FbI0{}, FbI2{}, FbN{}, FbN{}, g, F0, F1);
    step(FbI1{}, FbI1{}, FbN{}, FbN{}, g + 1, F1, F0);
    step(FbI2{}, FbI0{}, FbN{}, FbN{}, g + 2, F0, F1);
  }
  } else {
    issue(0); issue(1);   // (the plain loop of k_fb_prod<.., 0>)
    int slot = 0;
    for (int g = 0; g < G; ++g) {
      if (g + 1 < G) fb_wait_vm<kPW>();
      else fb_wait_vm<0>();
      if (!MIVI_KNOCKED(a, 64)) fb_barrier();
      if (g + 2 < G && !MIVI_KNOCKED(a, 1)) issue(slot == 0 ? 2 : slot - 1);
      if (work) {
        FbFrags<WJ> F;
        fb_read_frags<WJ>(lds, slot, wm, wn, lane, F);
        compute(std::integral_constant<int, -1>{}, g, F);
      }
      slot = slot == 2 ? 0 : slot + 1;
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < WJ; ++j) tot[i][j] += acc[i][j];
    const float r3 = rcur[i], r2 = rs[i][0], r1 = rs[i][1], r0 = rs[i][2];   // quarters 3, 2, 1, 0
    rs[i][0] = r0; rs[i][1] = r1; rs[i][2] = r2; rs[i][3] = r3;
  }
  fb_barrier();
  FB_STAMP(a, 2);
  if (MIVI_KNOCKED(a, 16)) { if (tot[0][0][0] == 123.f) grad[0] = tot[1][WJ - 1][3] + tot[0][0][2]; return; }
  float *Cs = reinterpret_cast<float *>(lds) + w * (32 * LDC);
  const double invM = 1.0 / (double)a.M_total;
  const bool pow2M = (a.M_total & (a.M_total - 1)) == 0;
  const float invMf = (float)invM;
  const double direct = direct_entropy_coeff(a.ent_kind);
  const int i4 = 4 * (lane & 7);
  FB_STAMP(a, 4);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (i == 1) FB_STAMP(a, 5);
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      if (cj[j] > ri[i]) continue;
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
        if (!diag && pow2M) {   // strictly below the diagonal, power-of-two sample count: vjp_elem's f32 branch for all four (no per-element branches)
          o = -v * invMf;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] = vjp_elem(v[c], gi + c, gj, pow2M, invMf, invM, direct, cjj);
        }
        if (!MIVI_KNOCKED(a, 8)) store16_wt(grad + d + (size_t)gj * d + gi, o);
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
  FB_STAMP(a, 3);
}

// -----------------------------------------------------------------------------------------------------------------
// Host side
// -----------------------------------------------------------------------------------------------------------------
namespace {
constexpr int kBM = 128, kBN = 128;
constexpr int kWJ = 1;
constexpr int kPFprod = 1, kPFvjp = 0;   // register prefetch + one workgroup per CU (product: its heaviest tile must own a CU) / two plain workgroups per CU (VJP: equal tiles)   // 32-column blocks per wave (k_fb_prod / k_fb_vjp): 2 = four waves of 64 x 64 per tile, 1 = eight waves of 64 x 32

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
  return !off && c->cfg.family == MIVI_FULLRANK && c->cfg.dtype == MIVI_F32 && c->cfg.d % kBM == 0 && M % kBN == 0 && c->cfg.d >= kBM &&
         c->cfg.d <= 2048 && M >= 128;   // (the run-boundary masks of k_fb_prod hold 64 sub-stages)
}
size_t fb_plane_words(const mivi_ctx *c, int M) { return (size_t)c->cfg.d * M / 512 * kFrag; }       // one lane's eps / W planes
size_t fb_cplane_words(const mivi_ctx *c) { return (size_t)(c->cfg.d / 32) * (c->cfg.d / 16) * kFrag; }

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
  // (an XCD's tiles heaviest first ACROSS its panels; panel by panel -- a panel's eight tiles side by side, sharing its eps fragments in the
  //  L2 -- measured slower: 50 against 36 us for 20 lanes, 221 against 183 us for 100)
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
  // the dense target's second product: every tile walks the whole K range (equal tiles); workgroup b runs on XCD b % 8 -- consecutive
  // workgroups take different row panels of P, so that with eight (or a multiple of eight) row panels an XCD's L2 keeps ONE of them
  std::vector<int4> prod2;
  for (int l = 0; l < L; ++l)
    for (int cb = 0; cb < ncb; ++cb)
      for (int rb = 0; rb < nrb; ++rb) prod2.push_back(make_int4(l, rb | (cb << 16), 0, 0));
  // the sticking-the-landing product: the draw's product's tiles, the first row blocks (the longest K ranges there) first
  std::vector<int4> prod3;
  for (auto &li : lists) std::stable_sort(li.begin(), li.end(), [](const int4 &p, const int4 &q) { return (p.y & 0xffff) < (q.y & 0xffff); });
  for (size_t i = 0; i < mx; ++i)
    for (int x = 0; x < 8; ++x)
      if (i < lists[x].size()) prod3.push_back(lists[x][i]);
  fb_upload(t.prod, prod.data(), prod.size() * sizeof(int4));
  fb_upload(t.vjp, vjp.data(), vjp.size() * sizeof(int4));
  fb_upload(t.prod2, prod2.data(), prod2.size() * sizeof(int4));
  fb_upload(t.prod3, prod3.data(), prod3.size() * sizeof(int4));
  if (!t.prod.p || !t.vjp.p || !t.prod2.p || !t.prod3.p) return nullptr;
  t.n_prod2 = (int)prod2.size();
  t.n_prod = (int)prod.size();
  t.n_vjp = (int)vjp.size();
  t.L = L;
  t.M = M;
  return &t;
}

static FbArgs fb_args(mivi_ctx *c, const void *params, int M) {
  FbTables &t = c->fb;
  const int d = c->cfg.d;
  FbArgs a{};
  a.d = d; a.M = M;
  a.params = (const float *)params;
  a.t_mean = (const float *)c->t_mean.p;
  a.t_istd = (const float *)c->t_istd.p;
  a.CA = (unsigned *)t.CA.p;
  a.epsP = (unsigned *)t.epsP.p;
  a.epsV = (unsigned *)t.epsV.p;
  a.WV = (unsigned *)t.WV.p;
  a.t_prec = (const float *)c->t_prec.p; a.dP = c->dP;
  a.PA = (unsigned *)t.PA.p;
  a.RP = (unsigned *)t.RP.p;
  a.Tinv = (const float *)t.Tinv.p;
  a.TA = (unsigned *)t.TA.p;
  a.plane_stride = (long long)fb_plane_words(c, M);
  a.ell_part = (double *)t.ell.p; a.ell_stride = (long long)(d / 32) * (M / 32);
  a.he_part = (double *)t.he.p; a.he_stride = (long long)(d / 64) * (M / 32);
  a.ld_part = (double *)t.ld.p;
  a.ent_kind = c->cfg.entropy; a.M_total = c->M_total;
  a.status = (int *)c->status.p;
  a.ell_const = c->t_const;
  return a;
}

// the draws of a step (+ tril(C)'s planes as riders of a call's first draw) on `stream`
void fb_launch_eps(mivi_ctx *c, const FbStep &s, bool with_cplanes, hipStream_t stream) {
  const int d = c->cfg.d, M = s.M, L = s.L;
  FbArgs a = fb_args(c, s.params, M);
  a.L = L;
  a.rng = s.rng;
  const int gx = (d / 64) * (M / 32), nf = (d / 32) * (d / 16);
  const int ycp = with_cplanes ? (nf + 8 * gx - 1) / (8 * gx) : 0;
  hipLaunchKernelGGL(k_fb_eps, dim3(gx, L + ycp), dim3(512), 0, stream, a);
}
// the dense-Gaussian target's precision matrix as operand planes (once per target: FbTables::PA_valid)
void fb_launch_pplanes(mivi_ctx *c, hipStream_t stream) {
  FbArgs a = fb_args(c, nullptr, c->cfg.n_mc);
  const int nf = (c->cfg.d / 32) * (c->cfg.d / 16);
  hipLaunchKernelGGL(k_fb_pplanes, dim3((nf + 3) / 4), dim3(256), 0, stream, a);
}
// C^-T (t.Tinv, left there by the solve kernels on the identity) as operand planes: once per call
void fb_launch_tplanes(mivi_ctx *c, hipStream_t stream) {
  FbArgs a = fb_args(c, nullptr, c->cfg.n_mc);
  const int nf = (c->cfg.d / 32) * (c->cfg.d / 16);
  hipLaunchKernelGGL(k_fb_tplanes, dim3((nf + 3) / 4), dim3(256), 0, stream, a);
}
// product + target (dense target: product -> R, the target's product) -> VJP (+ the lanes' values as extra workgroups of the VJP launch) on `stream`
void fb_launch_compute(mivi_ctx *c, const FbStep &s, hipStream_t stream, int which) {   // which (profiling): bit 0 the draw's product, bit 1 the VJP, bit 2 the dense target's product, bit 3 the sticking-the-landing product; 15 = all (default)
  const FbTab &tb = *s.tab;
  FbArgs a = fb_args(c, s.params, s.M);
  a.L = s.L;
  a.grads = (float *)s.grads; a.grad_stride = s.grad_stride;
  a.values = (float *)s.values; a.value_stride = s.value_stride;
  a.grad_last = (float *)s.grad_last; a.value_last = (float *)s.value_last; a.lane_last = s.lane_last;
  a.write_upper = s.write_upper;
  a.work = (const int4 *)tb.prod.p; a.n_work = tb.n_prod;
  if (s.dense) {
    if (which & 1) hipLaunchKernelGGL((k_fb_prod<kWJ, kPFprod, FB_DENSE_R>), dim3(tb.n_prod), dim3(512 / kWJ), 0, stream, a);
    a.work = (const int4 *)tb.prod2.p; a.n_work = tb.n_prod2;
    if (which & 4) hipLaunchKernelGGL((k_fb_prod<kWJ, kPFprod, FB_DENSE_G>), dim3(tb.n_prod2), dim3(512 / kWJ), 0, stream, a);
  } else if (which & 1) {
    hipLaunchKernelGGL((k_fb_prod<kWJ, kPFprod, FB_DIAG>), dim3(tb.n_prod), dim3(512 / kWJ), 0, stream, a);
  }
  if (s.stl && (which & 8)) {
    a.work = (const int4 *)tb.prod3.p; a.n_work = tb.n_prod;
    hipLaunchKernelGGL((k_fb_prod<kWJ, kPFprod, FB_STL_U>), dim3(tb.n_prod), dim3(512 / kWJ), 0, stream, a);
  }
  a.work = (const int4 *)tb.vjp.p; a.n_work = tb.n_vjp;
  if (which & 2) hipLaunchKernelGGL((k_fb_vjp<kWJ, kPFvjp>), dim3(tb.n_vjp + s.L), dim3(512 / kWJ), 0, stream, a);
}

}  // namespace mivi
