// Full-rank RepGradELBO contractions for a BATCH of estimates at the same parameters (gfx950): the batch engine.
//
// Reference semantics (AdvancedVI.jl v0.7.0), per estimate:
//   sampling   Z = scale * eps .+ mu                                  src/families/location_scale.jl:71-77
//   energy     mean_m logdensity(prob, z_m)                           src/algorithms/repgradelbo.jl:84-86
//   gradient   d/dC = -(1/M) tril(W eps') - direct * diag(1/C_ii),  d/dmu = -(1/M) W 1   (repgradelbo.jl:142-149)
// The reference evaluates ONE estimate per `estimate_gradient!` call; estimates at fixed parameters (monitoring with many samples,
// averaged gradients, the bench's step) are independent, and L of them are ONE matrix product each way:
//   product   [Z_1 .. Z_L] = mu + tril(C) [eps_1 .. eps_L]            1024 x (256 L) x 1024 (triangular) at the north star
//   VJP       dC_l = tril(W_l eps_l'),  l = 1 .. L                    L products 1024 x 1024 (lower) x 256
//   * OPERAND PLANES (fr_planes.h).  Every operand lives in memory as the two-way f16 split of its power-of-two-scaled f32 elements (hi / lo
//     planes, 4 bytes per element) in MFMA-FRAGMENT ORDER: a fragment = 32 rows x 16 k of one operand = 2 planes x 64 lanes x 16 bytes.
//     tril(C) is laid out once per call (rider workgroups of the first draw: row maxima -> row scales -> fragments, diagonal blocks masked),
//     eps by its generator ONCE, in the product's orientation (k_fb_eps: rows as k; round 6 -- the VJP, whose k are the samples, gathers its
//     B pieces from the same planes and transposes them on the way out of LDS with ds_read_b64_tr_b16), W by the product's
//     epilogue with one scale per (row, 128-sample tile).  A main loop is then LDS-DMA (1 KiB pieces) -> ds_read_b128 -> three
//     v_mfma_f32_32x32x16_f16 per fragment pair (lo.hi, hi.lo, hi.hi): no vector arithmetic at all.
//   * a workgroup owns a 128 x 128 output tile, a wave a 64 x 32 WJ part of it over the WHOLE K range; operands are staged once per
//     workgroup in an LDS ring of 16-k stages (16 KiB each), one barrier per stage; no cross-wave reduction of products; the epilogue
//     works through wave-private LDS images (stores are whole 128-byte lines), the tile's scale maxima cross the waves once.
//   * the launch covers every lane (estimate) of the step: per-lane buffers are base + lane * stride, the work table names (lane, tile).
// Accuracy: an f32-accurate product (2^-22 per term; measured 3.5e-7 relative l2 at K = 1024) -- the single calls (kernels_fullrank_lds.hip:
// exact three-way bf16 split) agree with a batch's estimates to rounding, not bitwise (tests/test_gpu_each.py states the tolerances).
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <cstdio>
#include <vector>

#include "device_common.h"
#include "fr_elem.h"
#include "fr_planes.h"

#ifndef FB_VK
#define FB_VK 0   // developer: compile-time work-skipping knock-outs of k_fb_vjp (16 no DMA, 32 no MFMA, 64 no LDS fragment reads, 128 no epilogue): tools/dbg/knock_vjp.sh
#endif
#ifndef FB_WJ
#define FB_WJ 1
#endif

namespace mivi {

struct FbArgs {
  int d, M, L;                    // d, M: the GEOMETRY -- rows / samples padded to whole 128 x 128 tiles (planes, tables, work items)
  int dT, MT;                     // the family's dimension and the samples per estimate (multiples of 32): the parameter / gradient layout (leading
                                  // dimension dT), the eps stream's index arithmetic, the normalisation; whole 32-blocks beyond them are padding: eps and W
                                  // planes hold zeros there, nothing of them is summed or stored
  const float *params;            // [mu; vec C]
  const float *t_mean, *t_istd;   // diagonal-Gaussian target
  unsigned *CA;                   // planes of tril(C): fragment (rb32, kg), kg <= 2 rb32 + 3, at (rb32 (d / 16) + kg) kFrag
  float *cscale;                  // [2][d]: row scales of tril(C), their inverses
  unsigned *epsP;                 // lane l: epsP + l * plane_stride; fragment (mb32, kg = row group) at (mb32 (d / 16) + kg) kFrag; 2^11 eps
  unsigned *WV;                   // lane l: WV + l * plane_stride; fragment (rb32, mg) at (rb32 (M / 16) + mg) kFrag
  float *winv;                    // lane l: winv + l * (M / 128) d: inverse scale of W's (row, 128-sample block): [M / 128][d]
  // dense-Gaussian target: g = -P (z - m) is a second product per lane (k_fb_prod<FB_DENSE_G>) between the draw's product and the VJP
  const float *t_prec;            // P, leading dimension dP
  int dP;
  unsigned *PA;                   // planes of P: fragment (rb32, kg), every kg < d / 16 (k_fb_pplanes)
  float *pscale;                  // [2][d]: row scales of P, their inverses
  unsigned *RP;                   // lane l: RP + l * plane_stride: R = Z - m as the second product's B operand, eps' product layout
  float *rinv;                    // lane l: rinv + l * (d / 128) M: inverse scale of R's (sample, 128-row block): [d / 128][M]
  // sticking-the-landing estimators: W += C^-T eps with the INVERSE of the scale formed once per call (the parameters are fixed inside it)
  const float *Tinv;              // C^-T, d x d, (row i, column k) at [i + k d], upper triangular (the solve kernels on the identity)
  unsigned *TA;                   // its planes: fragment (rb32, kg), kg >= 2 (rb32 & ~3), entries k < row zeroed (k_fb_tplanes)
  float *tscale;                  // [2][d]
  long long plane_stride;         // words per lane = d M / 512 * kFrag
  double *ell_part;               // lane l: ell_part + l * ell_stride; slot per (32-row block, 32-column block)
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
  int n_riders;                   // k_fb_eps: grid rows in front of the lanes' that lay out tril(C)
  int obj;                        // k_fb_eps: 1 = lane l draws samples [l M, (l + 1) M) of ONE estimate index (objective mode); 0 = estimate index + l
  // sharded batches: the VJP launch leaves lane l's shard-additive, UNNORMALISED partial vector at parts + l part_stride:
  //   [sum_m W_im (d) | the lower triangle of sum_m W (x) eps as its 128 x 128 tiles, tile (rb, cb <= rb) at d + (rb (rb + 1) / 2 + cb) 128^2,
  //    column-major inside, exact zeros above the diagonal of the diagonal tiles | sum_m ell_m | sum_m |eps_m|^2 / 2]   (fb_part_len floats)
  // -- every store a whole 16-byte vector (the column-packed triangle of the C ABI's partial vector is not 16-byte aligned per column);
  // the ranks' vectors are summed by ONE all-reduce per step, k_fb_finalize_parts turns the sums into values and gradients.
  float *parts;
  long long part_stride;
  int knock;                      // developer knock-outs (-DMIVI_DEV, MIVI_FB_KNOCK): 1 no DMA, 2 no MFMA, 4 no LDS reads, 8 no barriers
  long long *dbg;                 // developer timeline (-DMIVI_DEV builds, MIVI_FB_DBG=1): per workgroup {entry, first stage landed, main loop done, end} (100 MHz ticks), groups
};
#ifdef MIVI_DEV
#define FB_STAMP(a, slot) do { if ((a).dbg && threadIdx.x == 0) (a).dbg[(size_t)blockIdx.x * 8 + (slot)] = (long long)wall_clock64(); } while (0)
#define FB_NOTE(a, slot, v) do { if ((a).dbg && threadIdx.x == 0) (a).dbg[(size_t)blockIdx.x * 8 + (slot)] = (long long)(v); } while (0)
#define FB_KNOCKED(a, b) ((a).knock & (b))
#else
#define FB_KNOCKED(a, b) false
#define FB_STAMP(a, slot) do { } while (0)
#define FB_NOTE(a, slot, v) do { } while (0)
#endif

// -----------------------------------------------------------------------------------------------------------------
// A parameter-only A operand as operand planes: ONE workgroup (512 threads) per 32-row block rb -- the rows' largest magnitudes over the
// k range the products read (fragments kg_lo .. kg_hi) -> one power-of-two scale per row (scale[row], scale[d + row] = its inverse) ->
// the block's fragments, one wave per fragment.  ld(k, row) returns the (masked) f32 element.  red: 16 * 32 + 32 floats of LDS.
// -----------------------------------------------------------------------------------------------------------------
template <class Ld4, class Ld>
__device__ __forceinline__ void fb_rowblock_planes(int d, int rb, int kg_lo, int kg_hi, int sub, int nsub, Ld4 ld4, Ld ld, unsigned *planes, float *scale, float *red) {
  // row maxima: a thread takes 4 rows x one k per 16-byte load (rows are the contiguous axis), 64 k parts, eight loads in flight
  const int tid = threadIdx.x, r4 = tid & 7, part = tid >> 3, ng = d >> 4;
  f32x4 m = {0.f, 0.f, 0.f, 0.f};
  {
    int k = 16 * kg_lo + part;
    for (; k + 448 <= 16 * kg_hi + 15; k += 512) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ld4(k + 64 * u, 32 * rb + 4 * r4);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) m[c] = fmaxf(m[c], fabsf(v[u][c]));
    }
    for (; k <= 16 * kg_hi + 15; k += 64) {
      const f32x4 v = ld4(k, 32 * rb + 4 * r4);
#pragma unroll
      for (int c = 0; c < 4; ++c) m[c] = fmaxf(m[c], fabsf(v[c]));
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {   // lanes 8 apart hold the same rows: the wave's eight parts, then the eight waves through LDS
    m[c] = fmaxf(m[c], __shfl_xor(m[c], 8, 64));
    m[c] = fmaxf(m[c], __shfl_xor(m[c], 16, 64));
    m[c] = fmaxf(m[c], __shfl_xor(m[c], 32, 64));
  }
  if ((tid & 63) < 8) *(f32x4 *)(red + (tid >> 6) * 32 + 4 * r4) = m;
  lds_barrier();
  if (tid < 32) {
    float mm = red[tid];
#pragma unroll
    for (int p = 1; p < 8; ++p) mm = fmaxf(mm, red[p * 32 + tid]);
    float s, inv;
    fb_scale_of(mm, s, inv);
    red[512 + tid] = s;
    if (sub == 0) {
      scale[32 * rb + tid] = s;
      scale[d + 32 * rb + tid] = inv;
    }
  }
  lds_barrier();
  const int w = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const float s = red[512 + l31];
  for (int kg = kg_lo + w + 8 * sub; kg <= kg_hi; kg += 8 * nsub) {
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = ld(16 * kg + 8 * (e >> 2) + 4 * h + (e & 3), 32 * rb + l31) * s;
    fb_store_frag(planes + ((size_t)rb * ng + kg) * kFrag + 4 * lane, x);
  }
}
// tril(C): the fragments up to the diagonal block + the two zero groups behind it (a wave of k_fb_prod walks the K range of its SECOND row
// block with both of its blocks: the first one's chain adds exact zeros there)
__device__ __forceinline__ void fb_cplanes_block(const FbArgs &a, int rb, int sub, int nsub, float *red) {
  const int d = a.d, ng = d >> 4, dT = a.dT;
  const float *C = a.params + dT;   // (leading dimension dT; rows / columns beyond dT are padding: zeros)
  const int hi = 2 * rb + 3 < ng - 1 ? 2 * rb + 3 : ng - 1;
  if (dT == d) {   // whole tiles (the common case, and every benchmark shape): no padding masks in the row-maximum loops
    fb_rowblock_planes(d, rb, 0, hi, sub, nsub,
                       [C, d](int k, int row) { f32x4 v = *(const f32x4 *)(C + (size_t)k * d + row);
#pragma unroll
                                                for (int c = 0; c < 4; ++c) v[c] = k > row + c ? 0.f : v[c];
                                                return v; },
                       [C, d](int k, int row) { const float v = C[(size_t)k * d + row]; return k > row ? 0.f : v; }, a.CA, a.cscale, red);
    return;
  }
  fb_rowblock_planes(d, rb, 0, hi, sub, nsub,
                     [C, dT](int k, int row) { const bool in = row < dT && k < dT;   // (dT % 4 == 0: a vector is inside or outside; outside: any valid address, masked)
                                               f32x4 v = *(const f32x4 *)(C + (size_t)(in ? k : 0) * dT + (in ? row : 0));
#pragma unroll
                                               for (int c = 0; c < 4; ++c) v[c] = (!in || k > row + c) ? 0.f : v[c];
                                               return v; },
                     [C, dT](int k, int row) { if (row >= dT || k > row) return 0.f; return C[(size_t)k * dT + row]; }, a.CA, a.cscale, red);
}
// k_fb_pplanes: the dense-Gaussian target's precision matrix P (every k group: P is full).  Once per target.
__global__ __launch_bounds__(512) void k_fb_pplanes(FbArgs a) {
  __shared__ float red[16 * 32 + 32];
  const float *P = a.t_prec;
  const int dP = a.dP;
  fb_rowblock_planes(a.d, (int)blockIdx.x, 0, (a.d >> 4) - 1, (int)blockIdx.y, (int)gridDim.y,
                     [P, dP](int k, int row) { return *(const f32x4 *)(P + (size_t)k * dP + row); },
                     [P, dP](int k, int row) { return P[(size_t)k * dP + row]; }, a.PA, a.pscale, red);
}
// k_fb_tplanes: C^-T (upper triangular) -- entries below the diagonal are exact zeros whatever the solve left there; a 128-row tile of
// k_fb_prod<FB_STL_U> starts its K range at its FIRST row block's diagonal, so a row block's fragments start there.  Once per call.
__global__ __launch_bounds__(512) void k_fb_tplanes(FbArgs a) {
  __shared__ float red[16 * 32 + 32];
  const float *T = a.Tinv;
  const int d = a.d, rb = (int)blockIdx.x;
  fb_rowblock_planes(d, rb, 2 * (rb & ~3), (d >> 4) - 1, (int)blockIdx.y, (int)gridDim.y,
                     [T, d](int k, int row) { f32x4 v = *(const f32x4 *)(T + (size_t)k * d + row);
#pragma unroll
                                              for (int c = 0; c < 4; ++c) v[c] = k >= row + c ? v[c] : 0.f;
                                              return v; },
                     [T, d](int k, int row) { const float v = T[(size_t)k * d + row]; return k >= row ? v : 0.f; }, a.TA, a.tscale, red);
}

// -----------------------------------------------------------------------------------------------------------------
// k_fb_eps: eps of L estimates as operand planes (of 2^11 eps).  Draws: blocks of 64 rows x 32 columns, one Philox block
// per thread (rows 4 q .. 4 q + 3 of one column: the stream and the he_part partials of the single calls' k_eps); the block's values go
// through an LDS tile, threads 0..255 then assemble the four product fragments (column = row of the B operand, k = rows 16 ig ..).
// The kernel is bound by the generator's vector arithmetic (Philox's 32-bit multiplies + Box-Muller), not by its plane writes: halving the
// writes (round 6) took 7 % off it.  blockIdx.y - n_riders = lane; the first n_riders grid rows
// (a call's first draw only) lay out tril(C): one workgroup per 32-row block, the heaviest first.
// -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_fb_eps(FbArgs a) {
  __shared__ double red[8];
  __shared__ float E[32 * 65];   // E[m][i], leading dimension 65 (the riders' reduction area: 544 floats)
  const int tid = threadIdx.x, eb = blockIdx.x, d = a.d;
  if ((int)blockIdx.y < a.n_riders) {   // four workgroups per 32-row block of tril(C) (each finds the rows' scales, takes every fourth share of the fragments)
    const int r = (int)blockIdx.y * (int)gridDim.x + eb;
    if (r < 4 * (d >> 5)) fb_cplanes_block(a, (d >> 5) - 1 - (r >> 2), r & 3, 4, E);
    return;
  }
  const int l = (int)blockIdx.y - a.n_riders;
  const uint64_t idx = rng_index(a.rng) + (a.obj ? 0ull : (uint64_t)l);
  const int moff = a.rng.m_offset + (a.obj ? l * a.MT : 0);
  unsigned *epsP = a.epsP + (size_t)l * a.plane_stride;
  const int nrb6 = d >> 6;
  const int R64 = eb % nrb6, c32 = eb / nrb6;
  const int q = tid & 15, c = tid >> 4;
  const int ri = R64 * 64 + 4 * q, rm = c32 * 32 + c;
  float e[4];   // (the stream's index arithmetic is the family's: dT / 4 Philox blocks per sample; a padding row / sample holds zeros -- drawn and
  eps_block<float>(a.rng.seed, idx, (uint64_t)(moff + rm) * (uint64_t)(a.dT >> 2) + (uint64_t)(ri >> 2), e);   // discarded: a branch around the draw cost 18 registers)
  if (!(ri < a.dT && rm < a.MT)) { e[0] = 0.f; e[1] = 0.f; e[2] = 0.f; e[3] = 0.f; }
#pragma unroll
  for (int r = 0; r < 4; ++r) E[c * 65 + 4 * q + r] = e[r] * kEpsScale;
  const float he = 0.5f * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3]);
  const double sh = block_sum_nodrain_f32<512>(he, red);   // (its barriers also publish the tile)
  if (tid == 0) a.he_part[(size_t)l * a.he_stride + eb] = sh;
  // ONE orientation (round 6): the product's fragments (mb32 = c32, kg = 4 R64 + f): lane = column l31, slots = rows 16 f + ..; k_fb_vjp takes its B
  // operand from these same planes and transposes it on the way out of LDS.  (Rounds 4-5 wrote a second, VJP-oriented set: 2 MB per lane.)
  if (tid >= 256) return;
  const int lane = tid & 63, l31 = lane & 31, h = lane >> 5, f = tid >> 6;
  float x[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) x[s] = E[l31 * 65 + 16 * f + 8 * (s >> 2) + 4 * h + (s & 3)];
  fb_store_frag(epsP + ((size_t)c32 * (d >> 4) + 4 * R64 + f) * kFrag + 4 * lane, x);
}

// -----------------------------------------------------------------------------------------------------------------
// The two products share a staging scheme: a 128 x 128 tile, 8 / WJ waves, 16-k stages of 16 KiB (A: four fragments, B: four fragments, two
// planes each) in an LDS ring.  Fragment f of a stage (2 KiB): f < 4: A fragment f; else B fragment f - 4.  Wave w issues fragments
// WJ w .. WJ w + WJ - 1: always 2 WJ requests per wave and stage, so the vmcnt accounting is a compile-time constant.
// -----------------------------------------------------------------------------------------------------------------
constexpr int kImgW = 32 * 36;      // a wave-private 32 x 32 epilogue image (leading dimension 36), words
// Wave layout of a 128 x 128 tile, WJ = 32-column blocks per wave: 8 / WJ waves = 2 (row halves of 64) x 4 / WJ (column parts of 32 WJ).
//   WJ = 2: four waves (one per SIMD), a wave owns 64 x 64: 32 KiB of LDS reads per group and workgroup
//   WJ = 1: eight waves (two per SIMD), a wave owns 64 x 32: 48 KiB of LDS reads per group
// The issue order inside one iteration's straight-line block {2 WJ LDS-DMA requests, 4 + 2 WJ fragment reads of the next group, 6 WJ MFMAs}:
// every wave of the workgroup leaves the barrier at the same moment, and with the reads first (where the scheduler puts loads) all of
// them queue on the LDS port before the first MFMA of anybody issues -- the matrix pipe idles for the length of that burst.  Memory
// operations behind the MFMAs instead: the pipe starts at once, the reads trickle in under it.
template <int WJ>
__device__ __forceinline__ void fb_sched_interleave() {
  constexpr int NM = 6 * WJ, ND = 4 + 2 * WJ, NV = 2 * WJ;
  static_for<0, NM>([&](auto K) {
    constexpr int k = decltype(K)::value;
    constexpr int nrd = (k + 1) * ND / NM - k * ND / NM;                                       // this MFMA's share of the LDS reads
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                        // one MFMA
    if constexpr (nrd > 0) __builtin_amdgcn_sched_group_barrier(0x100, nrd, 0);
    if constexpr (k < NV) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                  // one LDS-DMA request
  });
}
using FbI0 = std::integral_constant<int, 0>;
using FbI1 = std::integral_constant<int, 1>;
using FbI2 = std::integral_constant<int, 2>;
using FbI3 = std::integral_constant<int, 3>;
using FbT = std::true_type;
using FbN = std::false_type;

// max of v over the lanes that differ in the given lane-index bits (xor butterfly)
template <int MASK>
__device__ __forceinline__ float fb_lane_max(float v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1)
    if (MASK & o) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// -----------------------------------------------------------------------------------------------------------------
// k_fb_prod: one 128 x 128 tile of a product whose B operand is a lane's planes, for every lane l of the step.
// Work item = (lane, rb, cb): rows [128 rb, +128) of columns [128 cb, +128) of lane l.  A 32-row block r32 of tril(C) has 2 (r32 + 1) groups.
// Software pipeline: iteration g computes on the fragments of stage g (already in registers) while the fragments of stage g + 1 are read
// from LDS and the DMA of stage g + kRing is issued.  Per iteration: wait for the own pieces of stage g + 1, barrier (stage g + 1 has landed
// for every wave; every wave has read stage g, whose slot stage g + kRing takes), issue, read, 6 WJ MFMAs.
// NRP = LDS ring slots (NRP - 1 stages in flight behind the one being read; the loops are unrolled by NRP): 4 (78 KiB of LDS: two workgroups
// share a CU and cover each other's barriers and epilogues) or 8 (135 KiB: the workgroup owns its CU) -- the host picks per step width.
// MODE: FB_DIAG: A = tril(C), B = eps; epilogue = the fused diagonal-Gaussian target: W planes (the VJP's A operand) + ell partials;
//   FB_DENSE_R: the same product, epilogue R = (mu + tril(C) eps) - m as the B-operand planes of the dense target's product;
//   FB_DENSE_G: the dense target's product itself, G = -P R: A = the planes of P over the WHOLE K range, B = R's planes -- scaled per
//   (sample, 128-row block): the chain accumulator is folded into the total every eight groups with the block's inverse scales --,
//   epilogue g = -(P r), ell += r g / 2, W planes + ell partials;
//   FB_STL_U: the sticking-the-landing term, W += C^-T eps: A = the planes of C^-T (upper triangular: a tile's K range starts at its first
//   row and runs to the end), B = eps' planes, epilogue: the lane's W planes read back, + U, re-scaled, split and stored again.  No
//   counterpart among the one-estimate kernels (they SOLVE C^T X = eps, kernels_stl.hip).
// Epilogues that produce an operand: the values go to wave-private LDS images, the tile's maxima (per row over its 128 samples for W, per
// sample over its 128 rows for R) cross the waves through LDS once, then the images are scaled, split and stored as fragments.
// -----------------------------------------------------------------------------------------------------------------
enum { FB_DIAG = 0, FB_DENSE_R = 1, FB_DENSE_G = 2, FB_STL_U = 3 };
template <int WJ, int MODE, int NRP>
__global__ __launch_bounds__(512 / WJ, (MODE == 2 && NRP == 4 && WJ == 1) ? 4 : 2 / WJ) void k_fb_prod(FbArgs a) {   // (the dense target's product at two workgroups per CU: 128 registers)
  constexpr bool kDG = MODE == FB_DENSE_G, kSU = MODE == FB_STL_U, kDR = MODE == FB_DENSE_R;
  constexpr int LDC = 36, NF = WJ, kPW = 2 * WJ;   // fragments / pieces this wave stages per group
  constexpr int NW = 8 / WJ, NWN = 4 / WJ;         // waves; waves along the columns
  constexpr int kBody = (NRP * kStageW > 16 * kImgW) ? NRP * kStageW : 16 * kImgW;   // the ring, later the waves' epilogue images
  // DENSE_G: R's planes are scaled per (sample, 128-row block); the chain accumulator stays ONE accumulator, re-based at every block boundary
  // by the exact power-of-two ratio of the neighbouring scales of its column (as k_fb_vjp does for W: no second accumulator set, 139 -> <= 128
  // registers).  `tab` [d / 128][128]: the boundary ratios of the tile's 128 samples, the last row the closing factor.  With four ring slots
  // it sits in the part of the body that only the epilogue images use (16 images = 72 KiB, the ring 64 KiB), so the kernel takes 78 KiB like
  // the other modes and TWO workgroups share a CU; `rsv` [128]: R's inverse scales of the tile's OWN row block (the epilogue reads r back).
  constexpr bool kTabInBody = kDG && NRP * kStageW + 2048 <= kBody;
  constexpr int kTab = kDG ? (kTabInBody ? 128 : 2048 + 128) : 0;
  __shared__ __attribute__((aligned(16))) unsigned lds[kBody + 4 * 128 + NW * 64 + kTab];
  float *vec = reinterpret_cast<float *>(lds + kBody);          // [4][128]: mu, target mean, target 1 / std, the rows' output factor
  float *red = vec + 4 * 128;                                   // [NW][64]: the waves' maxima
  float *rsv = red + NW * 64;
  float *tab = kTabInBody ? reinterpret_cast<float *>(lds + NRP * kStageW) : rsv + 128;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / NWN, wn = w % NWN;
  const __attribute__((address_space(4))) int *wp = (const __attribute__((address_space(4))) int *)a.work + 4 * blockIdx.x;
  const int ln = wp[0], rc = wp[1], flags = wp[2];
  const int rb = rc & 0xffff, cb = rc >> 16;
  const int d = a.d, ng = d >> 4;
  const int row0 = rb * 128, col0 = cb * 128;
  const int R0 = row0 >> 5;               // first 32-row block of the tile
  const int g0 = kSU ? 2 * R0 : 0;        // first group of the K range
  const int G = kDG ? ng : (kSU ? ng - g0 : 2 * (R0 + 4));  // groups of the workgroup (the last row block's K; the dense product: all of K)
  // this wave's NF fragments of a stage
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
  // The tile's row vectors (+ DENSE_G: R's inverse scales) in two halves.  load_tables: every global load of them, issued IN FRONT of the
  // prologue's DMA requests and not waited for -- the loads and the first stages travel together; stage_tables: behind the requests, the
  // arithmetic and the LDS writes.  (Round 6: the inverse scales used to be loaded inside the serial re-basing loop, behind the requests --
  // d / 128 dependent round trips, each behind a vmcnt(0), before the tile's first group: 6.5 us of a dense tile's lifetime at d = 1024.)
  constexpr int kMaxNB = 16;   // d <= 2048
  float tv[4] = {0.f, 0.f, 0.f, 0.f}, tri[kDG ? kMaxNB : 1], trs = 0.f;
  auto load_tables = [&]() {
    if (tid < 128) {
      if constexpr (!kDG && !kSU) {
        const bool in = row0 + tid < a.dT;   // (padding rows: mu = m = 0, 1 / std = 0)
        tv[0] = in ? a.params[row0 + tid] : 0.f;
        tv[1] = in ? a.t_mean[row0 + tid] : 0.f;
        if (MODE == FB_DIAG) tv[2] = in ? a.t_istd[row0 + tid] : 0.f;
      }
      const float *sc = kDG ? a.pscale : (kSU ? a.tscale : a.cscale);
      tv[3] = sc[d + row0 + tid];
      if constexpr (kDG) {
        const float *ri = a.rinv + (size_t)ln * (size_t)(d >> 7) * a.M + col0 + tid;
        const int NB = d >> 7;
#pragma unroll
        for (int b = 0; b < kMaxNB; ++b) tri[b] = ri[(size_t)(b < NB ? b : NB - 1) * a.M];   // (clamped, not predicated: straight-line loads -- a branch per load and the compiler sinks the lot behind the requests)
        trs = ri[(size_t)rb * a.M];
      }
    }
    asm volatile("" ::: "memory");   // (the loads stay in front of the requests: the compiler otherwise sinks them into stage_tables' branch)
  };
  auto stage_tables = [&]() {
    if (tid < 128) {
      if constexpr (!kDG && !kSU) {
        vec[tid] = tv[0];
        vec[128 + tid] = tv[1];
        if (MODE == FB_DIAG) vec[256 + tid] = tv[2];
      }
      vec[384 + tid] = tv[3] * ((kDG) ? 1.f : kEpsInv);   // what a row's raw sums are multiplied by
      if constexpr (kDG) {
        const int NB = d >> 7;
        float u = tri[0], grow = 1.f;   // u: the inverse of the unit the accumulator is in; grow: how far the re-basing has scaled it up
#pragma unroll
        for (int b = 1; b < kMaxNB; ++b) {
          if (b < NB) {
            float ratio = u / tri[b];
            if (!(ratio <= 1099511627776.f)) ratio = 1099511627776.f;          // 2^40 per boundary (also for a non-finite quotient) ...
            if (ratio > 1.f && grow * ratio > 1.2089258e24f) ratio = fmaxf(1.2089258e24f / grow, 1.f);   // ... and 2^80 in all: the accumulator cannot overflow
            if (ratio > 1.f) grow *= ratio;
            tab[(b - 1) * 128 + tid] = ratio;
            u = u / ratio;
          }
        }
        tab[(NB - 1) * 128 + tid] = u;
        rsv[tid] = trs;
      }
    }
  };
  auto issue = [&](int slot) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      unsigned *dst = lds + slot * kStageW + (NF * w + f) * 512;
      FB_GLDS16(sp[f], dst, 0);
      FB_GLDS16(sp[f], dst, 1024);
      sp[f] += gd < gmax[f] ? kFrag : 0;
    }
    ++gd;
  };
  f32x16 acc[2][WJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int r32[2] = {R0 + 2 * wm, R0 + 2 * wm + 1};
  // A wave computes groups 0 .. Gw - 1 (its second row block's K range: a multiple of four groups) with BOTH row blocks, unconditionally:
  // one straight MFMA block per group.  The first row block ends two groups earlier: its planes carry two zero fragments behind its
  // diagonal block, so its chain adds exact zeros there.
  const int Gw = (kDG || kSU) ? G : 2 * r32[1] + 2;
  auto after_group = [&](auto S, int g) {   // S = g mod kRing
    if constexpr (kDG && (decltype(S)::value & 3) == 3) {   // (behind the MFMAs: the block in front of them stays one basic block)
      if (((g + 1) & 7) == 0) {                        // a 128-row block of R ended with this group: into the next block's unit (after the last: out of the scaled units)
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
          const float f = tab[((g >> 3) << 7) + 32 * (WJ * wn + j) + l31];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] *= f;
          }
        }
        asm volatile("" ::: "memory");
      }
    }
  };
  auto compute = [&](auto S, int g, const FbFrags<WJ> &F) {
    if (!FB_KNOCKED(a, 2)) fb_group<WJ>(F, acc);
    fb_sched_interleave<WJ>();
    after_group(S, g);
  };
  // (the assembly group body's LDS byte addresses: this lane's first A / B fragment in ring slot 0, and in slot 4)
  const unsigned lds0 = (unsigned)(uintptr_t)lds + 16u * lane;
  const unsigned ab_lo = lds0 + (2 * wm) * 2 * 1024, bb_lo = lds0 + (8 + WJ * wn * 2) * 1024;
  const unsigned ab_hi = ab_lo + 4 * kStageW * 4, bb_hi = bb_lo + 4 * kStageW * 4;
  (void)ab_hi; (void)bb_hi;
  // iteration g (S = g mod kRing): wait for the own pieces of stage g + 1, barrier, issue stage g + kRing into the slot of stage g, read the
  // fragments of stage g + 1, compute stage g.  VM = stages that may stay in flight across the wait; CMP: this wave still has work.
  auto step = [&](auto S, auto VM, auto ISS, int g, const FbFrags<WJ> &Fc, FbFrags<WJ> &Fn) {
    constexpr int sl = decltype(S)::value;
    fb_wait_vm<kPW * decltype(VM)::value>();
    if (!FB_KNOCKED(a, 8)) fb_barrier();
#ifdef MIVI_SB
    __builtin_amdgcn_sched_barrier(0);   // nothing of this group moves up in front of its barrier
#endif
    // the group's MFMAs with the next group's fragment reads between them: one pinned assembly block (fr_planes.h) for the shipped wave
    // tile; the compiler's own schedule (reads behind the last MFMA) for WJ = 2 and the developer knock-outs
    auto group_and_read = [&]() {
#ifndef MIVI_DEV
      if constexpr (WJ == 1) {
        constexpr int nx = (sl + 1) % NRP, off = (nx & 3) * kStageW * 4;
        fb_group_read_asm<off>(acc[0][0], acc[1][0], Fc, Fn, nx < 4 ? ab_lo : ab_hi, nx < 4 ? bb_lo : bb_hi);
        if constexpr (kDG && (sl & 3) == 3) {
          // MFMA results -> vector ALU: the wait states the compiler would have counted (tied to the accumulators: nothing that reads them moves above)
          if (((g + 1) & 7) == 0) asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0][0]), "+v"(acc[1][0])::"memory");
        }
        after_group(S, g);
      } else
#endif
      {
        if (!FB_KNOCKED(a, 4)) fb_read_frags<WJ>(lds, (sl + 1) % NRP, wm, wn, lane, Fn);
        compute(S, g, Fc);
      }
    };
    if constexpr (decltype(ISS)::value) {
      if (!FB_KNOCKED(a, 1)) issue(sl);
      group_and_read();
    } else if (g < Gw) {   // (the tail: a wave-uniform branch)
      group_and_read();
    }
  };
  FB_STAMP(a, 0);
  FB_NOTE(a, 4, G);
  load_tables();
  static_for<0, NRP>([&](auto S) { issue(decltype(S)::value); });   // (G >= 8 >= NRP)
  fb_wait_vm<kPW * NRP>();   // the tables' loads are older than every request: they have landed, the stages need not have
  stage_tables();
  fb_wait_vm<kPW * (NRP - 1)>();
  fb_barrier();
  FB_STAMP(a, 1);
  {
    FbFrags<WJ> F0, F1;
    fb_read_frags<WJ>(lds, 0, wm, wn, lane, F0);
    int g = 0;
    for (; g + NRP < G; g += NRP) {   // (G is a multiple of eight; Gw = G for the waves of the tile's lower half, G - 4 for the upper half's)
      static_for<0, NRP>([&](auto S) {
        constexpr int sv = decltype(S)::value;
        if constexpr (sv % 2 == 0) step(S, std::integral_constant<int, NRP - 2>{}, FbT{}, g + sv, F0, F1);
        else step(S, std::integral_constant<int, NRP - 2>{}, FbT{}, g + sv, F1, F0);
      });
    }
    // the last NRP groups: nothing left to request; the upper half's K range ends four groups early (its waves then only keep the barriers)
    static_for<0, NRP - 1>([&](auto S) {
      constexpr int sv = decltype(S)::value;
      if constexpr (sv % 2 == 0) step(S, std::integral_constant<int, NRP - 2 - sv>{}, FbN{}, g + sv, F0, F1);
      else step(S, std::integral_constant<int, NRP - 2 - sv>{}, FbN{}, g + sv, F1, F0);
    });
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the last fragments were read inside an assembly block the compiler cannot see into
    if (g + NRP - 1 < Gw) compute(std::integral_constant<int, NRP - 1>{}, g + NRP - 1, F1);
  }
  f32x16 (&tot)[2][WJ] = acc;
  fb_barrier();   // every wave is done with the ring: LDS becomes the waves' private epilogue images
  FB_STAMP(a, 2);
  float *img = reinterpret_cast<float *>(lds) + w * (2 * WJ * kImgW);   // image (i, j) at img + (i WJ + j) kImgW: [column][row], then W[m][i]
  unsigned *WVl = a.WV + (size_t)ln * a.plane_stride;
  unsigned *RPl = a.RP + (size_t)ln * a.plane_stride;
  double *ellp = a.ell_part + (size_t)ln * a.ell_stride;
  const int ncb = a.M >> 5, nmg = a.M >> 4;
  const int ei4 = 4 * (lane & 7);
  float amax[2][4];        // W-producing modes: rows ei4 .. ei4 + 3 of block i over this lane's columns; STL: amax[i][0] = row l31 of block i
  float cmax[WJ][4];       // DENSE_R: column 8 p + lane / 8 of block j over this lane's rows
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) amax[i][c] = 0.f;
#pragma unroll
  for (int j = 0; j < WJ; ++j)
#pragma unroll
    for (int p = 0; p < 4; ++p) cmax[j][p] = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lr = 64 * wm + 32 * i;   // row offset inside the tile
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, tm = mu, tis = mu;
    const f32x4 zf = *(const f32x4 *)(vec + 384 + lr + ei4);
    if constexpr (!kDG && !kSU) {
      mu = *(const f32x4 *)(vec + lr + ei4);
      tm = *(const f32x4 *)(vec + 128 + lr + ei4);
      if constexpr (MODE == FB_DIAG) tis = *(const f32x4 *)(vec + 256 + lr + ei4);
    }
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      float *I = img + (i * WJ + j) * kImgW;
      const int cb32 = (col0 >> 5) + WJ * wn + j;
      const bool pad = 32 * r32[i] >= a.dT || 32 * cb32 >= a.MT;   // (whole 32-blocks: dT and MT are multiples of 32)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = {tot[i][j][4 * q], tot[i][j][4 * q + 1], tot[i][j][4 * q + 2], tot[i][j][4 * q + 3]};
        *(f32x4 *)(I + l31 * LDC + 8 * q + 4 * h) = v;
      }
      if constexpr (kSU) {
        // U's 32 x 32 tile (image [sample][row]) added to this wave's two W fragments: the new W stays in the image
        const float uf = vec[384 + lr + l31];
        const float wo = a.winv[(size_t)ln * (size_t)(a.M >> 7) * d + (size_t)cb * d + row0 + lr + l31];
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          const unsigned *src = WVl + ((size_t)r32[i] * nmg + 2 * cb32 + g2) * kFrag + 4 * lane;
          const u32x4v wh = *(const u32x4v *)src, wl = *(const u32x4v *)(src + 256);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float *p = I + (16 * g2 + 8 * (e >> 2) + 4 * h + (e & 3)) * LDC + l31;
            const float x = __builtin_fmaf(*p, uf, fb_unsplit2(wh, wl, e) * wo);
            *p = x;
            amax[i][0] = fmaxf(amax[i][0], fabsf(x));
          }
        }
        continue;
      }
      double s = 0.0;
#pragma unroll
      for (int p = 0; p < 4; ++p) {   // pass p: columns 8 p .. 8 p + 7, rows ei4 .. ei4 + 3 per lane
        const int en = 8 * p + (lane >> 3);
        const f32x4 v = *(const f32x4 *)(I + en * LDC + ei4) * zf;
        float ell = 0.f;
        f32x4 wv;
        if constexpr (MODE == FB_DIAG) {
          const f32x4 z = mu + v;
#pragma unroll
          for (int c = 0; c < 4; ++c) wv[c] = diag_target_elem(z[c], tm[c], tis[c], ell);
        } else if constexpr (kDR) {
          const f32x4 z = mu + v;
          wv = z - tm;
        } else {
          // r = (z - m)[rows ei4 .. + 3, column en] back from R's planes: fragment (cb32, kg = 2 r32 + ei4 / 16), lane (en, h' = ei4 / 4 % 2),
          // slots 4 (ei4 / 8 % 2) + c = the two words 2 (ei4 / 8 % 2) + {0, 1}; times the inverse scale of (sample en, this 128-row block)
          const unsigned *fr = RPl + ((size_t)cb32 * ng + 2 * r32[i] + (ei4 >> 4)) * kFrag + 4 * (en + 32 * ((ei4 >> 2) & 1)) + 2 * ((ei4 >> 3) & 1);
          const uint2 qh = *(const uint2 *)fr, ql = *(const uint2 *)(fr + 256);
          const float rs = rsv[32 * (WJ * wn + j) + en];
          const unsigned uh[2] = {qh.x, qh.y}, ul[2] = {ql.x, ql.y};
#pragma unroll
          for (int c = 0; c < 4; ++c) wv[c] = dense_target_elem(v[c], fb_unsplit2_word(uh[c >> 1], ul[c >> 1], c & 1) * rs, ell);
        }
        if (pad) { wv = f32x4{0.f, 0.f, 0.f, 0.f}; ell = 0.f; }   // a padding row block / sample block: zero planes, nothing summed
        *(f32x4 *)(I + en * LDC + ei4) = wv;   // the image becomes W[m][i] (FB_DENSE_R: R[m][i])
        if constexpr (kDR) {
          cmax[j][p] = fmaxf(cmax[j][p], fmaxf(fmaxf(fabsf(wv[0]), fabsf(wv[1])), fmaxf(fabsf(wv[2]), fabsf(wv[3]))));
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) amax[i][c] = fmaxf(amax[i][c], fabsf(wv[c]));
          const double sv = (double)wave_sum_f32(ell);
          s = p ? s + sv : sv;
        }
      }
      if constexpr (!kDR) {
        if (lane == 0) ellp[(size_t)r32[i] * ncb + cb32] = s;
      }
    }
  }
  // the tile's maxima: across the lanes of a wave, then across the waves that share the rows (W) / the columns (R)
  if constexpr (kDR) {
#pragma unroll
    for (int j = 0; j < WJ; ++j)
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float m = fb_lane_max<7>(cmax[j][p]);
        if ((lane & 7) == 0) red[w * 64 + 32 * j + 8 * p + (lane >> 3)] = m;
      }
  } else if constexpr (kSU) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float m = fb_lane_max<32>(amax[i][0]);
      if (lane < 32) red[w * 64 + 32 * i + lane] = m;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float m = fb_lane_max<56>(amax[i][c]);
        if (lane < 8) red[w * 64 + 32 * i + ei4 + c] = m;
      }
  }
  fb_barrier();
  if constexpr (kDR) {
    // R as the dense product's B fragments (mb32 = cb32, kg = 2 r32 + g2): lane (column l31, h), slots = rows 16 g2 + 8 (e / 4) + 4 h + e % 4
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int cb32 = (col0 >> 5) + WJ * wn + j;
      float s, inv;
      fb_scale_of(fmaxf(red[wn * 64 + 32 * j + l31], red[(NWN + wn) * 64 + 32 * j + l31]), s, inv);
      if (wm == 0 && lane < 32) a.rinv[(size_t)ln * (size_t)(d >> 7) * a.M + (size_t)rb * a.M + col0 + 32 * (WJ * wn + j) + lane] = inv;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float *I = img + (i * WJ + j) * kImgW;
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          const f32x4 x0 = *(const f32x4 *)(I + l31 * LDC + 16 * g2 + 4 * h) * s, x1 = *(const f32x4 *)(I + l31 * LDC + 16 * g2 + 8 + 4 * h) * s;
          const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
          fb_store_frag(RPl + ((size_t)cb32 * ng + 2 * r32[i] + g2) * kFrag + 4 * lane, x);
        }
      }
    }
  } else {
    // W as the VJP's A fragments (rb32 = r32[i], mg = 2 cb32 + g2): lane (row l31, h), slots = samples 16 g2 + 8 (e / 4) + 4 h + e % 4
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float mx = red[(wm * NWN) * 64 + 32 * i + l31];
#pragma unroll
      for (int x = 1; x < NWN; ++x) mx = fmaxf(mx, red[(wm * NWN + x) * 64 + 32 * i + l31]);
      float s, inv;
      fb_scale_of(mx, s, inv);
      if (wn == 0 && lane < 32) a.winv[(size_t)ln * (size_t)(a.M >> 7) * d + (size_t)cb * d + row0 + 64 * wm + 32 * i + lane] = inv;
#pragma unroll
      for (int j = 0; j < WJ; ++j) {
        const float *I = img + (i * WJ + j) * kImgW;
        const int cb32 = (col0 >> 5) + WJ * wn + j;
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          float x[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = I[(16 * g2 + 8 * (e >> 2) + 4 * h + (e & 3)) * LDC + l31] * s;
          fb_store_frag(WVl + ((size_t)r32[i] * nmg + 2 * cb32 + g2) * kFrag + 4 * lane, x);
        }
      }
    }
  }
  FB_STAMP(a, 3);
  if (!kDG && !kSU && (flags & 1) && wn == 0 && lane < 32) {   // log|det C| partials of this wave's two row blocks (lane 0's first column block carries the flag)
    const int nrb = d >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 32 * r32[i] + lane;
      float lg, bad;
      logdet_block32(r < a.dT ? a.params[a.dT + (size_t)r * a.dT + r] : 1.f, lg, bad);   // (a padding block: log 1 = 0)
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
  if (a.parts) {   // sharded batches: this shard's two scalars of the lane's partial vector
    out.partials_mode = 1;
    out.partials = a.parts + (size_t)l * a.part_stride;
    out.scalars_off = (int64_t)d + (int64_t)(d >> 7) * ((d >> 7) + 1) / 2 * 16384;   // (sharded batches run on unpadded shapes: d == dT)
  }
  out.ent_kind = a.ent_kind;
  out.M_total = a.M_total;
  out.M_local = a.MT;
  out.status = a.status;
  const float *pp = a.params;
  const int dT = a.dT;   // (the partial arrays are laid out by the padded geometry, their padding entries are zeros; the entropy constants are the family's)
  finalize_value_block<float, 256, false>(dT, vin, out, (int64_t)dT + (int64_t)dT * dT, [pp, dT](int i) { return pp[dT + (size_t)i * dT + i]; }, red);
}

// -----------------------------------------------------------------------------------------------------------------
// k_fb_vjp: dC_l = -(1/M) tril(W_l eps_l') - direct diag(1 / C_ii), dmu_l = -(1/M) W_l 1, for every lane l of the step.
// Work item = (lane, rb, cb), cb <= rb: the 128 x 128 tile of the lower triangle; K = M samples = M / 16 groups.  W's planes are scaled per
// (row, 128-sample block): every eight groups the chain accumulator is folded into the total with the block's inverse row scales (staged
// in LDS once per workgroup).  A wave whose sub-tiles all lie strictly above the diagonal only carries its share of the staging; the
// exact zeros of the upper triangle are written as the mirror images of the strictly lower 32 x 32 blocks (lanes with the write_upper
// duty).  d/dmu: the waves of the diagonal tiles sum W's fragments (hi + lo, times the block's inverse scale).  The step's objective
// values ride as extra workgroups.
// -----------------------------------------------------------------------------------------------------------------
// NRV = ring slots of the plain loop, TABW = words of the scale table (M <= TABW), WPE = waves per SIMD the register budget allows:
// (3, 256, 6) = 50 KiB of LDS and <= 80 registers: THREE workgroups per CU for n_mc <= 256; (3, 2048, 4): two per CU otherwise.
template <int WJ, int NRV, int TABW, int WPE, bool PART = false>
__global__ __launch_bounds__(512 / WJ, WPE / WJ) void k_fb_vjp(FbArgs a) {
  static_assert(WJ == 1, "the transposing B reads are written for the 64 x 32 wave tiles");
  constexpr int LDC = 36, NF = WJ, kPW = 2 * WJ;
  constexpr int NR = NRV;
  constexpr int kBody = (NR * kStageW > 16 * kImgW / 2) ? NR * kStageW : 16 * kImgW / 2;   // the ring, later one image per wave
  __shared__ __attribute__((aligned(16))) unsigned lds[kBody + TABW];
  // W's planes are scaled per (row, 128-sample block).  The chain accumulator stays ONE accumulator: at a block boundary its rows are
  // multiplied by s_next / s_current (powers of two: exact), at the end by the last block's inverse scale.  wf[r][row], r < n_blocks - 1: the
  // boundary ratios, clamped to 2^40 (beyond that the next block's contribution is below 2^-40 of what the row already holds: entering it
  // in a too small unit over-weights it by the clamped excess, still below 2^-40 of the row's largest block); the ratios above one are also
  // capped at 2^80 in all, so the accumulator cannot overflow; wf[n_blocks - 1][row]: the closing factor.
  float *wf = reinterpret_cast<float *>(lds + kBody);
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w / (4 / WJ), wn = w % (4 / WJ);
  if ((int)blockIdx.x < a.L) {   // the objective values of the step's lanes: everything they sum is older than this launch.  In FRONT of the
    if (tid < 256) fb_value_block(a, (int)blockIdx.x, reinterpret_cast<double *>(lds));   // tiles: a one-workgroup latency chain each, hidden under them
    return;
  }
  const __attribute__((address_space(4))) int *wp = (const __attribute__((address_space(4))) int *)a.work + 4 * ((int)blockIdx.x - a.L);
  const int ln = wp[0], rc = wp[1];
  const int rb = rc & 0xffff, cb = rc >> 16;
  const int d = a.d, M = a.M, nmg = M >> 4, dT = a.dT;   // (d, M: the padded geometry; the gradient has leading dimension dT)
  const int row0 = rb * 128, col0 = cb * 128;
  const bool last = ln == a.lane_last && a.grad_last;
  float *grad = last ? a.grad_last : a.grads + (size_t)ln * a.grad_stride;
  const bool upper = a.write_upper || last;
  const int G = nmg;
  const int NB = M >> 7;   // 128-sample blocks
  const unsigned *sp[NF];           // the stage the next issue takes
  // EPS ONCE (round 6): B = eps' comes from the draws' PRODUCT-orientation planes (fragment (mb32, kg): lane (sample, h) holds the eight
  // dims 16 kg + 8 (e / 4) + 4 h + e % 4).  A B piece pair (1 KiB hi + 1 KiB lo) of a stage = the 16 samples of group mg x the 32 dims of
  // block jb = two product fragments' sample halves: DMA lane p fetches the 16-byte chunk (kgpar, hp, s) with
  //   p = 8 (4 (s / 8) + 2 hp + kgpar) + ((s % 8) ^ 4 hp)
  // -- eight consecutive lanes = one 128-byte line (its two 64-byte halves swapped where hp = 1) -- and the wave's B fragment comes out of
  // LDS through ds_read_b64_tr_b16 (lane c of a 16-lane group receives element c % 4 of the 8-byte chunks whose addresses lanes
  // 4 j + c / 4 supply): lane (dim n = 16 (G % 2) + c, h' = G / 2) gets samples 4 h' + j (second read: + 8) = the fragment slots e = j,
  // 4 + j.  The position map makes the 16 chunks of a 32-lane read group (kgpar, hp, s % 4) tile 256 bytes: no bank conflict.
  unsigned btr;                     // this lane's byte offset inside a B piece for the transposing reads
  {
    const int G4 = lane >> 4, c16 = lane & 15;
    const int kgpar = G4 & 1, hq = G4 >> 1, hp = c16 & 1, half = (c16 >> 1) & 1, sl = 4 * hq + (c16 >> 2);
    btr = 16u * (unsigned)(8 * (2 * hp + kgpar) + ((sl & 7) ^ (4 * hp))) + 8u * (unsigned)half;
  }
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int fs = NF * w + f, fr = fs & 3;
    if (fs < 4) {
      sp[f] = a.WV + (size_t)ln * a.plane_stride + ((size_t)((row0 >> 5) + fr) * nmg) * kFrag + 4 * lane;
    } else {
      const int m8 = lane >> 3, s8 = m8 >> 2, hp = (m8 >> 1) & 1, kgpar = m8 & 1, sl = 8 * s8 + ((lane & 7) ^ (4 * hp));
      sp[f] = a.epsP + (size_t)ln * a.plane_stride + ((size_t)((col0 >> 4) + 2 * fr + kgpar)) * kFrag + 4 * (sl + 32 * hp);
    }
  }
  int gi = 0;                       // groups requested so far (EPS ONCE: the B pointers alternate between a fragment's sample halves)
  const int bstep_odd = (d >> 4) * kFrag - 64;   // from the second sample half of (mb32, kg) to the first of (mb32 + 1, kg), words
  // W's inverse scales: loaded IN FRONT of the prologue's DMA requests and not waited for (load_table); the divisions and the LDS writes
  // behind the requests (build_table).  (Round 6: they were loaded inside the serial loop, behind the requests: n_mc / 128 dependent round
  // trips, each behind a vmcnt(0), in front of a main loop of only n_mc / 16 groups.)
  constexpr int kMaxNBv = 16;   // n_mc <= 2048
  float twi[kMaxNBv];
  auto load_table = [&]() {
    if (tid < 128) {
      const float *wi = a.winv + (size_t)ln * (size_t)NB * d + row0 + tid;
#pragma unroll
      for (int r = 0; r < kMaxNBv; ++r) twi[r] = wi[(size_t)(r < NB ? r : NB - 1) * d];   // (clamped, not predicated: straight-line loads)
    }
    asm volatile("" ::: "memory");   // (the loads stay in front of the requests)
  };
  auto build_table = [&]() {
    if (tid < 128) {
      float u = twi[0], grow = 1.f;   // u: the inverse of the unit the accumulator is in; grow: how far the re-basing has scaled it up
#pragma unroll
      for (int r = 1; r < kMaxNBv; ++r) {
        if (r < NB) {
          float ratio = u / twi[r];
          if (!(ratio <= 1099511627776.f)) ratio = 1099511627776.f;   // 2^40 per boundary (also for a non-finite quotient) ...
          if (ratio > 1.f && grow * ratio > 1.2089258e24f) ratio = fmaxf(1.2089258e24f / grow, 1.f);   // ... and 2^80 over all boundaries (up to 15 of them at n_mc = 2048)
          if (ratio > 1.f) grow *= ratio;
          wf[(r - 1) * 128 + tid] = ratio;
          u = u / ratio;
        }
      }
      wf[(NB - 1) * 128 + tid] = u;
    }
  };
  auto issue = [&](int slot) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      unsigned *dst = lds + slot * kStageW + (NF * w + f) * 512;
      if (!(FB_VK & 16)) {
        FB_GLDS16(sp[f], dst, 0);
        FB_GLDS16(sp[f], dst, 1024);
      }
      if (NF * w + f >= 4) sp[f] += (gi & 1) ? bstep_odd : 64;
      else sp[f] += kFrag;
    }
    ++gi;
  };
  auto read_frags = [&](int slot, FbFrags<WJ> &F) {
    if ((FB_VK & 64)) return;
    const unsigned *cur = lds + slot * kStageW + 4 * lane;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 2; ++p) F.A[i][p] = *(const u32x4v *)(cur + ((2 * wm + i) * 2 + p) * 256);
    // (assembly, not __builtin_amdgcn_ds_read_tr16_b64: behind the builtin the compiler waits vmcnt(0) -- for the LDS-DMA requests just issued --
    //  in every group; the block ends with lgkmcnt(0), so the values are valid for whatever the compiler does with them)
    const unsigned bb = (unsigned)(uintptr_t)(lds + slot * kStageW) + (8 + 2 * wn) * 1024 + btr;
    uint2 t0, t1, t2, t3;
    asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:512\n\tds_read_b64_tr_b16 %2, %4 offset:1024\n\t"
                 "ds_read_b64_tr_b16 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(bb) : "memory");
    F.B[0][0] = u32x4v{t0.x, t0.y, t1.x, t1.y};
    F.B[0][1] = u32x4v{t2.x, t2.y, t3.x, t3.y};
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
  f32x16 acc[2][WJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < WJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  double rsd[2] = {0.0, 0.0};   // d/dmu: this lane's share (row l31, its half of the k slots) of the row sums of W, in the accumulator's unit
  float rcur[2] = {0.f, 0.f};
  // A wave with work computes ALL its sub-tiles in every group (one straight MFMA block: see k_fb_prod); a sub-tile strictly above the
  // diagonal (diagonal tiles only) is simply not stored.
  auto after_group = [&](auto S, int g, const FbFrags<WJ> &F) {
    if (__builtin_expect(dg[0] || dg[1], 0)) {   // (two waves of a diagonal tile; behind the MFMAs: the block in front of them stays one basic block)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        if (dg[i]) {
#pragma unroll
          for (int e = 0; e < 8; ++e) rcur[i] += fb_unsplit2(F.A[i][0], F.A[i][1], e);
        }
    }
    if constexpr (decltype(S)::value == 3 || decltype(S)::value == -1) {
      if (((g + 1) & 7) == 0 && g + 1 < G) {   // a 128-sample block (but the last) ended with this group: into the next block's unit
        const float *wr = wf + ((g >> 3) << 7) + 64 * wm;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 f4 = *(const f32x4 *)(wr + 32 * i + 8 * q + 4 * h);
#pragma unroll
            for (int j = 0; j < WJ; ++j)
#pragma unroll
              for (int c = 0; c < 4; ++c) acc[i][j][4 * q + c] *= f4[c];
          }
          rsd[i] = (rsd[i] + (double)rcur[i]) * (double)wr[32 * i + l31];
          rcur[i] = 0.f;
        }
        asm volatile("" ::: "memory");
      }
    }
  };
  auto compute = [&](auto S, int g, const FbFrags<WJ> &F) {
    if (!(FB_VK & 32)) fb_group<WJ>(F, acc);
    after_group(S, g, F);
  };
  {
    // plain loop, NR-slot ring: wait for stage g, barrier, request stage g + NR - 1 into the slot stage g - 1 was read from, read, compute -- the
    // other workgroup of the CU runs its MFMAs under this one's waits
    FB_STAMP(a, 0);
    FB_NOTE(a, 4, G);
    load_table();
#pragma unroll
    for (int s0 = 0; s0 < NR - 1; ++s0) issue(s0);   // (G >= 8)
    fb_wait_vm<kPW * (NR - 1)>();   // (the table's loads are older than the requests)
    build_table();
    int slot = 0;
    for (int g = 0; g < G; ++g) {
      if (g == 1) FB_STAMP(a, 1);
      if (g + NR - 2 < G) fb_wait_vm<kPW * (NR - 2)>();
      else if (NR > 3 && g + 1 < G) fb_wait_vm<kPW>();
      else fb_wait_vm<0>();
      fb_barrier();
      if (g + NR - 1 < G) issue(slot == 0 ? NR - 1 : slot - 1);
      if (work) {
        FbFrags<WJ> F;
        read_frags(slot, F);
        compute(std::integral_constant<int, -1>{}, g, F);
      }
      slot = slot == NR - 1 ? 0 : slot + 1;
    }
  }
  fb_barrier();
  FB_STAMP(a, 2);
  if ((FB_VK & 128)) return;   // (developer knock-out: no epilogue)
  float *Cs = reinterpret_cast<float *>(lds) + w * kImgW;
  const float *wfin = wf + (NB - 1) * 128 + 64 * wm;   // the closing factors of this wave's rows
  const double invM = 1.0 / (double)a.M_total;
  const bool pow2M = (a.M_total & (a.M_total - 1)) == 0;
  const float invMe = (float)invM * kEpsInv;   // (eps' planes hold 2^11 eps)
  const double direct = direct_entropy_coeff(a.ent_kind);
  const int i4 = 4 * (lane & 7);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      if (cj[j] > ri[i] || 32 * ri[i] >= dT) continue;   // above the diagonal / a padding row block (then cj <= ri: the columns are inside too)
      const bool diag = ri[i] == cj[j];
      const int rbase = 32 * ri[i], cbase = 32 * cj[j];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        *(f32x4 *)(Cs + l31 * LDC + 8 * q + 4 * h) = v;
      }
      const f32x4 ff = *(const f32x4 *)(wfin + 32 * i + i4);
#pragma unroll
      for (int p = 0; p < 4; ++p) {   // lane = (rows i4 .. i4 + 3, column n) of pass p
        const int n = 8 * p + (lane >> 3);
        const int gi = rbase + i4, gj = cbase + n;
        float cjj = 1.f;
        if (diag && gj >= gi && gj < gi + 4) cjj = a.params[dT + (size_t)gj * dT + gj];
        const f32x4 v = *(const f32x4 *)(Cs + n * LDC + i4) * ff;
        f32x4 o;
        if constexpr (PART) {   // the raw sums into the lane's partial vector, tile-packed (no normalisation, no entropy term: k_fb_finalize_parts)
          o = v * kEpsInv;
          if (diag) {
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = gj > gi + c ? 0.f : o[c];
          }
          store16_wt(a.parts + (size_t)ln * a.part_stride + d + ((size_t)(rb * (rb + 1) / 2 + cb) << 14) + (size_t)(gj - col0) * 128 + (gi - row0), o);
          continue;
        }
        if (!diag && pow2M) {   // strictly below the diagonal, power-of-two sample count: exact scaling, no per-element branches
          o = -v * invMe;
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] = vjp_elem(v[c] * kEpsInv, gi + c, gj, pow2M, (float)invM, invM, direct, cjj);
        }
        store16_wt(grad + dT + (size_t)gj * dT + gi, o);
      }
      if (!PART && !diag && upper) {   // the mirrored, strictly upper 32 x 32 block is structurally zero
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int ii = 8 * p + (lane >> 3);
          store16_wt(grad + dT + (size_t)(rbase + ii) * dT + cbase + i4, z4);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the image is read before the next sub-tile overwrites it
    }
    if (dg[i] && 32 * ri[i] < dT) {   // d/dmu rows of this row block: the two halves' shares
      const double mine = (rsd[i] + (double)rcur[i]) * (double)wfin[32 * i + l31];
      const double sm = mine + __shfl_xor(mine, 32, 64);
      if constexpr (PART) {
        if (lane < 32) a.parts[(size_t)ln * a.part_stride + 32 * ri[i] + lane] = (float)sm;
      } else {
        if (lane < 32) grad[32 * ri[i] + lane] = dmu_elem(sm, invM);
      }
    }
  }
  FB_STAMP(a, 3);
}

// -----------------------------------------------------------------------------------------------------------------
// k_fb_finalize_parts: the lanes' (all-reduced) partial vectors -> values and dense gradients, what k_finalize does for the C ABI's packed
// vector.  blockIdx.y = lane, blockIdx.x = column j of d/dC (the last grid column: d/dmu + the value).  Per element vjp_elem / dmu_elem, so a
// one-rank "sharded" batch reproduces mivi_estimate_gradient_n's numbers.
// -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fb_finalize_parts(FbArgs a) {
  __shared__ double red[4 * 4];
  const int d = a.d, l = blockIdx.y, j = blockIdx.x, tid = threadIdx.x;
  const float *P = a.parts + (size_t)l * a.part_stride;
  const bool last = l == a.lane_last && a.grad_last;
  float *grad = last ? a.grad_last : a.grads + (size_t)l * a.grad_stride;
  const bool upper = a.write_upper || last;
  const double invM = 1.0 / (double)a.M_total;
  const bool pow2M = (a.M_total & (a.M_total - 1)) == 0;
  const double direct = direct_entropy_coeff(a.ent_kind);
  if (j == d) {   // d/dmu and the objective value
    for (int i = tid; i < d; i += 256) grad[i] = dmu_elem((double)P[i], invM);
    double s_ld = 0.0, bad = 0.0;
    for (int i = tid; i < d; i += 256) {
      const double c = (double)a.params[d + (size_t)i * d + i];
      if (!(c > 0.0)) bad = 1.0;
      s_ld += log(c);
    }
    double v[2] = {s_ld, bad};
    block_sum_n<double, 256, 2>(v, red);
    if (tid == 0) {
      const int64_t so = (int64_t)d + (int64_t)(d >> 7) * ((d >> 7) + 1) / 2 * 16384;
      const double sum_ell = (double)P[so], s_he = (double)P[so + 1], Mt = (double)a.M_total;
      const double ent = (ent_is_closed(a.ent_kind) ? 0.5 * d * (1.0 + kLog2Pi) : s_he / Mt + 0.5 * d * kLog2Pi) + v[0];
      const double value = -(sum_ell / Mt + ent);
      float *vo = (l == a.lane_last && a.value_last) ? a.value_last : a.values + (size_t)l * a.value_stride;
      *vo = (float)value;
      int st = 0;
      if (!isfinite(value)) st |= 1;
      if (v[1] > 0.0) st |= 2;
      if (st && a.status) atomicOr(a.status, st);
    }
    return;
  }
  const int cb = j >> 7, lj = j & 127;
  const float cjj = a.params[d + (size_t)j * d + j];
  for (int i4 = 4 * tid; i4 < d; i4 += 1024) {   // rows i4 .. i4 + 3 of column j
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    if (i4 + 3 >= j) {
      const int rb = i4 >> 7;
      const f32x4 v = *(const f32x4 *)(P + d + ((size_t)(rb * (rb + 1) / 2 + cb) << 14) + (size_t)lj * 128 + (i4 & 127));
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] = vjp_elem(v[c], i4 + c, j, pow2M, (float)invM, invM, direct, cjj);
    } else if (!upper) {
      continue;   // (a scratch lane's buffer holds the zeros above the diagonal already)
    }
    *(f32x4 *)(grad + d + (size_t)j * d + i4) = o;
  }
}

// -----------------------------------------------------------------------------------------------------------------
// Host side
// -----------------------------------------------------------------------------------------------------------------
namespace {
constexpr int kBM = 128, kBN = 128;
constexpr int kWJ = FB_WJ;      // 32-column blocks per wave (k_fb_prod / k_fb_vjp): 2 = four waves of 64 x 64 per tile, 1 = eight waves of 64 x 32
constexpr int kOwnCuTilesPerCU10 = 17;   // (x 0.1) tiles per CU up to which the triangular products keep one workgroup per CU at d = 1024 (fb_launch_compute: scaled by the heaviest tile's chain)
// (the product: register prefetch + one workgroup per CU at few lanes; the VJP: plain loops, three workgroups per CU)

void fb_upload(DevBuf &b, const void *src, size_t bytes) {
  if (b.bytes < bytes || !b.p) {
    if (b.p) (void)hipFree(b.p);
    (void)hipMalloc(&b.p, bytes);
    b.bytes = bytes;
  }
  (void)hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice);
}
}  // namespace

// Round 6: d and n_mc are multiples of 32 (round 5: of 128); the geometry is padded to whole 128 x 128 tiles (fb_pad), whole 32-blocks of
// padding carry zero planes and are neither summed nor stored.  The dense target, the sticking-the-landing estimators and the sharded
// batches keep whole tiles (fb_whole_tiles: their parameter-only operands P / C^-T and the tile-packed partial vector are laid out by d).
static int fb_pad(int x) { return (x + 127) / 128 * 128; }
bool fb_shape_ok(const mivi_ctx *c, int M) {
  static const bool off = getenv("MIVI_BATCH_GEN3") && atoi(getenv("MIVI_BATCH_GEN3")) == 0;   // A/B: the lane-batched second-generation kernels
  return !off && c->cfg.family == MIVI_FULLRANK && c->cfg.dtype == MIVI_F32 && c->cfg.d % 32 == 0 && M % 32 == 0 && c->cfg.d >= kBM &&
         c->cfg.d <= 2048 && M >= 128 && M <= 2048;   // (the inverse-scale tables of k_fb_prod<FB_DENSE_G> / k_fb_vjp hold 2048 entries)
}
bool fb_whole_tiles(const mivi_ctx *c, int M) { return c->cfg.d % kBM == 0 && M % kBN == 0; }
size_t fb_plane_words(const mivi_ctx *c, int M) { return (size_t)fb_pad(c->cfg.d) * fb_pad(M) / 512 * kFrag; }       // one lane's eps / W planes
size_t fb_cplane_words(const mivi_ctx *c) { return (size_t)(fb_pad(c->cfg.d) / 32) * (fb_pad(c->cfg.d) / 16) * kFrag; }
size_t fb_part_len(const mivi_ctx *c) {   // [sum W (d) | lower-triangle tiles | sum ell, sum eps^2 / 2 | pad to four floats]
  const size_t d = (size_t)c->cfg.d, T = d / 128;
  return (d + T * (T + 1) / 2 * 16384 + 2 + 3) / 4 * 4;
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
  // the slot's previous tables may still be read by launches in flight on the context's stream (the uploads below are synchronous copies on
  // the null stream, which a non-blocking stream does not order against); and a failed upload must not leave the slot looking valid
  (void)hipStreamSynchronize(c->stream);
  t.L = t.M = 0;
  const int d = fb_pad(c->cfg.d), nrb = d / kBM, ncb = fb_pad(M) / kBN;   // (the padded geometry)
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
  // Round 5 (FETCH_SIZE: 9.4 MB per lane against 2 MB of R + W): with one row panel per XCD a lane's R panel (0.5 MB of planes) is fetched by
  // all eight XCDs.  While two row panels of P fit an L2 beside the streams (<= 1 MiB of planes: d <= 1024), an XCD keeps a PAIR of row
  // panels and takes every second column panel: R is fetched by four XCDs instead of eight, P still once per XCD
  // (ns_dense, 50 lanes: 202 k -> 210 k estimates/s; one, two or four row groups measured the same).
  std::vector<int4> prod2;
  const size_t panel_bytes = (size_t)kBM * d * 4;
  constexpr int NG = 4, NC = 8 / NG;
  if (nrb % NG == 0 && (size_t)(nrb / NG) * panel_bytes <= (1u << 20)) {
    std::vector<std::vector<int4>> lp(8);
    int pn = 0;
    for (int l = 0; l < L; ++l)
      for (int cb = 0; cb < ncb; ++cb, ++pn)
        for (int rg = 0; rg < NG; ++rg)
          for (int r = 0; r < nrb / NG; ++r) lp[rg + NG * (pn % NC)].push_back(make_int4(l, (rg * (nrb / NG) + r) | (cb << 16), 0, 0));
    size_t m3 = 0;
    for (auto &li : lp) m3 = std::max(m3, li.size());
    for (size_t i = 0; i < m3; ++i)
      for (int x = 0; x < 8; ++x)
        if (i < lp[x].size()) prod2.push_back(lp[x][i]);
  } else {
    for (int l = 0; l < L; ++l)
      for (int cb = 0; cb < ncb; ++cb)
        for (int rb = 0; rb < nrb; ++rb) prod2.push_back(make_int4(l, rb | (cb << 16), 0, 0));
  }
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

static FbArgs fb_args(mivi_ctx *c, const void *params, int M_true) {
  FbTables &t = c->fb;
  const int d = fb_pad(c->cfg.d), M = fb_pad(M_true);   // the geometry; dT / MT: the family's dimension, the samples per estimate
  FbArgs a{};
  a.d = d; a.M = M;
  a.dT = c->cfg.d; a.MT = M_true;
  a.params = (const float *)params;
  a.t_mean = (const float *)c->t_mean.p;
  a.t_istd = (const float *)c->t_istd.p;
  a.CA = (unsigned *)t.CA.p;
  a.cscale = (float *)t.cscale.p;
  a.pscale = (float *)t.pscale.p;
  a.tscale = (float *)t.tscale.p;
  a.winv = (float *)t.winv.p;
  a.rinv = (float *)t.rinv.p;
  a.epsP = (unsigned *)t.epsP.p;
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
  const int d = fb_pad(c->cfg.d), M = fb_pad(s.M), L = s.L;
  FbArgs a = fb_args(c, s.params, s.M);
  a.L = L;
  a.rng = s.rng;
  a.obj = s.obj;
  const int gx = (d / 64) * (M / 32);
  a.n_riders = with_cplanes ? (4 * (d / 32) + gx - 1) / gx : 0;   // tril(C)'s planes: four workgroups per 32-row block, in FRONT of the lanes' draws
  hipLaunchKernelGGL(k_fb_eps, dim3(gx, L + a.n_riders), dim3(512), 0, stream, a);
}
// the dense-Gaussian target's precision matrix as operand planes (once per target: FbTables::PA_valid)
void fb_launch_pplanes(mivi_ctx *c, hipStream_t stream) {
  FbArgs a = fb_args(c, nullptr, c->cfg.n_mc);
  hipLaunchKernelGGL(k_fb_pplanes, dim3(c->cfg.d / 32, 4), dim3(512), 0, stream, a);
}
// C^-T (t.Tinv, left there by the solve kernels on the identity) as operand planes: once per call
void fb_launch_tplanes(mivi_ctx *c, hipStream_t stream) {
  FbArgs a = fb_args(c, nullptr, c->cfg.n_mc);
  hipLaunchKernelGGL(k_fb_tplanes, dim3(c->cfg.d / 32, 4), dim3(512), 0, stream, a);
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
  a.parts = (float *)s.parts; a.part_stride = s.part_stride;
  if (s.obj) { a.ent_kind = s.ent_kind; a.M_total = s.M; }
#ifdef MIVI_DEV
  static long long *dbg_buf = nullptr;
  static const bool dbg_on = getenv("MIVI_FB_DBG") != nullptr;
  if (dbg_on && !dbg_buf) (void)hipMalloc(&dbg_buf, 2 * 8192 * 8 * sizeof(long long));
  auto dump = [&](const char *name, int n) {
    if (!dbg_on) return;
    (void)hipStreamSynchronize(stream);
    std::vector<long long> h((size_t)n * 8);
    (void)hipMemcpy(h.data(), dbg_buf, h.size() * 8, hipMemcpyDeviceToHost);
    long long t0 = h[0], t3 = 0;
    for (int i = 0; i < n; ++i) { t0 = std::min(t0, h[(size_t)i * 8]); t3 = std::max(t3, h[(size_t)i * 8 + 3]); }
    double pro = 0, loop = 0, epi = 0, pg = 0; long long gs = 0;
    for (int i = 0; i < n; ++i) { const long long *r = &h[(size_t)i * 8]; pro += r[1] - r[0]; loop += r[2] - r[1]; epi += r[3] - r[2]; gs += r[4]; }
    pg = loop / (double)gs;
    fprintf(stderr, "[fbdbg] %s: %d workgroups, span %.2f us | mean prologue %.2f us, main loop %.2f us (%.3f us per group), epilogue %.2f us | last start %.2f us\n", name, n, (t3 - t0) * 0.01, pro / n * 0.01, loop / n * 0.01, pg * 0.01, epi / n * 0.01, 0.0);
    // start-time histogram: when do workgroups begin, relative to the first
    std::vector<long long> st; for (int i = 0; i < n; ++i) st.push_back(h[(size_t)i * 8] - t0);
    std::sort(st.begin(), st.end());
    fprintf(stderr, "[fbdbg]   starts (us): p25 %.2f p50 %.2f p75 %.2f p90 %.2f max %.2f\n", st[n / 4] * 0.01, st[n / 2] * 0.01, st[3 * n / 4] * 0.01, st[9 * n / 10] * 0.01, st[n - 1] * 0.01);
  };
  a.dbg = dbg_on ? dbg_buf : nullptr;
  a.knock = getenv("MIVI_FB_KNOCK") ? atoi(getenv("MIVI_FB_KNOCK")) : 0;
#endif
  a.work = (const int4 *)tb.prod.p; a.n_work = tb.n_prod;
  // One workgroup per CU (eight ring slots) where a step's heaviest tiles pace the launch and a second workgroup on their CU only slows them:
  // measured at the north star (us, 4 / 8 slots): 16 lanes 23.1 / 23.4, 20: 27.6 / 23.8, 24: 28.9 / 27.7, 28: 30.7 / 30.9, 40: 37.9 / 44.5.
  // Round 6: the rule is on the LAUNCH'S SHAPE, not on the lane count (round 5 keyed it on 17 <= L <= 26, a window around the north star's
  // 20-lane step that is wrong 4x off at d = 512 or n_mc = 512): the eight-slot ring pays where the triangular product's tiles number between
  // one and kOwnCuTilesPerCU10 / 10 per CU -- fewer: every tile has a CU to itself either way (measured equal); more: throughput, i.e. two
  // workgroups per CU covering each other's barriers and epilogues, decides.  The dense target's second product has EQUAL tiles (all of K):
  // it never takes the one-per-CU ring (20 lanes: 320 tiles would run as two rounds on 256 CUs -- 46.0 against 54.7 us measured).
  const int n_cu = c->n_cu > 0 ? c->n_cu : 256;
  auto own_cu_for = [&](int n, bool equal_tiles) {
#ifdef MIVI_DEV
    static const int force_ring = getenv("MIVI_FB_RING") ? atoi(getenv("MIVI_FB_RING")) : 0;   // developer builds: 4 / 8 pins the ring (tools/fb_lane_curve.py)
    if (force_ring == 4) return false;
    if (force_ring == 8) return true;
#endif
    // tiles per CU (x 10) up to which the one-per-CU ring wins, by the heaviest tile's chain (gmax = d / 16 groups): measured at
    // (512, 128) and (512, 512) -- 32 groups: never --, (1024, 256) -- 64 groups: up to 1.7 --, (2048, 128) -- 128 groups: up to 2.2 (32 lanes:
    // 58.5 against 65.5 us; 48 lanes: 88.4 against 84.5); interpolated between (tools/experiments/README_r06.md)
    const int gmax = fb_pad(c->cfg.d) / 16;
    const int thr10 = gmax <= 32 ? 10 : (gmax <= 64 ? 10 + (gmax - 32) * 7 / 32 : kOwnCuTilesPerCU10 + (gmax - 64 < 64 ? gmax - 64 : 64) * 5 / 64);
    return !equal_tiles && n > n_cu && 10 * (long long)n <= (long long)thr10 * n_cu;
  };
  auto prod = [&](auto MODE, int n) {
    constexpr int md = decltype(MODE)::value;
    if (own_cu_for(n, md == FB_DENSE_G)) hipLaunchKernelGGL((k_fb_prod<kWJ, md, 8>), dim3(n), dim3(512 / kWJ), 0, stream, a);
    else hipLaunchKernelGGL((k_fb_prod<kWJ, md, 4>), dim3(n), dim3(512 / kWJ), 0, stream, a);
  };
  if (s.dense) {
    if (which & 1) prod(std::integral_constant<int, FB_DENSE_R>{}, tb.n_prod);
    a.work = (const int4 *)tb.prod2.p; a.n_work = tb.n_prod2;
    if (which & 4) prod(std::integral_constant<int, FB_DENSE_G>{}, tb.n_prod2);
  } else if (which & 1) {
    prod(std::integral_constant<int, FB_DIAG>{}, tb.n_prod);
#ifdef MIVI_DEV
    dump("k_fb_prod<DIAG>", tb.n_prod);
#endif
  }
  if (s.stl && (which & 8)) {
    a.work = (const int4 *)tb.prod3.p; a.n_work = tb.n_prod;
    prod(std::integral_constant<int, FB_STL_U>{}, tb.n_prod);
  }
  a.work = (const int4 *)tb.vjp.p; a.n_work = tb.n_vjp;
  if (s.values_only) {   // values only (objective mode; the each entry without gradients): the lanes' value workgroups alone, no VJP tile
    if (which & 2) hipLaunchKernelGGL((k_fb_vjp<kWJ, 3, 256, 6>), dim3(s.L), dim3(512 / kWJ), 0, stream, a);
    return;
  }
  if (which & 2) {
    // measured at the north star (us per 20 / 50 / 80 lanes): ring 3 + three workgroups per CU 26.5 / 63.3 / 105-115; ring 3, two per CU 26.4 /
    // 69.8 / 109; ring 4, two per CU 36.7 / 81.5 / 126; ring 2, three per CU 26.4 / 72.6 / 113 (round 4's kernel with its second accumulator: 29 / 70 / 115)
    if (s.parts) {   // sharded batches: the lanes' partial vectors instead of gradients
      if (a.M <= 256) hipLaunchKernelGGL((k_fb_vjp<kWJ, 3, 256, 6, true>), dim3(tb.n_vjp + s.L), dim3(512 / kWJ), 0, stream, a);
      else hipLaunchKernelGGL((k_fb_vjp<kWJ, 3, 2048, 4, true>), dim3(tb.n_vjp + s.L), dim3(512 / kWJ), 0, stream, a);
    }
    else if (a.M <= 256) hipLaunchKernelGGL((k_fb_vjp<kWJ, 3, 256, 6>), dim3(tb.n_vjp + s.L), dim3(512 / kWJ), 0, stream, a);
    else hipLaunchKernelGGL((k_fb_vjp<kWJ, 3, 2048, 4>), dim3(tb.n_vjp + s.L), dim3(512 / kWJ), 0, stream, a);
#ifdef MIVI_DEV
    dump("k_fb_vjp", tb.n_vjp);
#endif
  }
}

void fb_launch_finalize_parts(mivi_ctx *c, const FbStep &s, hipStream_t stream) {
  FbArgs a = fb_args(c, s.params, s.M);
  a.L = s.L;
  a.grads = (float *)s.grads; a.grad_stride = s.grad_stride;
  a.values = (float *)s.values; a.value_stride = s.value_stride;
  a.grad_last = (float *)s.grad_last; a.value_last = (float *)s.value_last; a.lane_last = s.lane_last;
  a.write_upper = s.write_upper;
  a.parts = (float *)s.parts; a.part_stride = s.part_stride;
  hipLaunchKernelGGL(k_fb_finalize_parts, dim3(c->cfg.d + 1, s.L), dim3(256), 0, stream, a);
}

}  // namespace mivi
