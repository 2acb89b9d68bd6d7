// Sticking-the-landing term of the full-rank family, second generation:  W += X,  C^T X = eps.
//
// Reference: `logpdf(q_stop, z)` inside StickingTheLandingEntropy (src/algorithms/entropy.jl:57-65, 80-90) differentiates
// through C^{-1}(z - mu) (src/families/location_scale.jl:59-63); in the closed-form VJP that is the extra term C^{-T} eps in W.
//
// The first generation (kernels_fullrank.hip, k_stl_solve_la16) gives every 16 sample columns one workgroup that walks all
// d/32 block rows: 16 CUs busy, 95 us at d = 1024.  A triangular solve is sequential in its block rows, and a step of the
// chain is only cheap while it stays inside one CU (LDS + s_barrier: a few hundred cycles; across CUs: a kernel boundary).  So:
//   * the flops move out of the chain: one level of recursion,  [C11 0; C21 C22]^T [X1; X2] = [E1; E2]  =>
//         X2 = C22^{-T} E2                                   (half-size solve)
//         X1 = C11^{-T} (E1 - C21^T X2) = Y1 - F^T X2,       Y1 = C11^{-T} E1,   F^T = C11^{-T} C21^T
//     Y1 does not depend on X2 and F depends on the parameters only, so ALL THREE half-size solves -- X2 and Y1 with the M sample
//     columns, F^T with the d/2 columns of C21^T as right-hand sides -- run side by side in ONE launch (M/16 + M/16 + d/32
//     workgroups), followed by one plain (d/2 x d/2) x (d/2 x M) product on the whole chip (k_stl_update32).  (Round 2 ran
//     X2, then R1 = E1 - C21^T X2, then X1 = C11^{-T} R1: two half-size solves back to back on 16 CUs, 13.5 + 4.9 + 13.5 us at
//     d = 1024; this order is 13.5 + 4.9 us.)  Half of the work is a GEMM, each solve streams a quarter of C;
//   * the chain itself (k_stl_solve64) runs on 64-row blocks with PRE-INVERTED diagonal blocks, 16 columns per workgroup, four
//     waves on the dependency chain and four on the updates that are not urgent, the pivot block exchanged through LDS already
//     split into bf16 pieces in fragment order; products on v_mfma_f32_16x16x32_bf16 with the exact three-way split
//     (kernels_fullrank_lds.hip);
//   * everything that depends only on the parameters -- the inverses of the diagonal blocks, the re-laying of the off-diagonal
//     blocks into fragment order, the bf16 split of what the chain waves consume -- is done by workgroups that RIDE in the
//     sampling kernel (stl_dinv.h), off the critical path.
// d in {256, 512, 1024, 2048}, M % 32 == 0; other shapes keep the first-generation kernels.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "device_common.h"
#include "stl_dinv.h"

namespace mivi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));

#define MIVI_GLDS16(gptr, lptr)                                                                            \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),                 \
                                   (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

// exact three-way bf16 split (truncation) of 4 / 8 f32 values; pieces packed in element order
__device__ __forceinline__ void split3x4(const f32x4 &x, u32x2v &hi, u32x2v &mid, u32x2v &lo) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float a = x[2 * p], b = x[2 * p + 1];
    const unsigned ab = __builtin_bit_cast(unsigned, a), bb = __builtin_bit_cast(unsigned, b);
    const float ra = a - __builtin_bit_cast(float, ab & 0xFFFF0000u), rb = b - __builtin_bit_cast(float, bb & 0xFFFF0000u);
    const unsigned rab = __builtin_bit_cast(unsigned, ra), rbb = __builtin_bit_cast(unsigned, rb);
    const float sa = ra - __builtin_bit_cast(float, rab & 0xFFFF0000u), sb = rb - __builtin_bit_cast(float, rbb & 0xFFFF0000u);
    hi[p] = __builtin_amdgcn_perm(bb, ab, 0x07060302u);
    mid[p] = __builtin_amdgcn_perm(rbb, rab, 0x07060302u);
    lo[p] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, sb), __builtin_bit_cast(unsigned, sa), 0x07060302u);
  }
}
__device__ __forceinline__ void split3x8(const f32x4 &x0, const f32x4 &x1, bf16x8 &hi, bf16x8 &mid, bf16x8 &lo) {
  u32x2v h0, m0, l0, h1, m1, l1;
  split3x4(x0, h0, m0, l0);
  split3x4(x1, h1, m1, l1);
  const u32x4v uh = {h0[0], h0[1], h1[0], h1[1]}, um = {m0[0], m0[1], m1[0], m1[1]}, ul = {l0[0], l0[1], l1[0], l1[1]};
  hi = __builtin_bit_cast(bf16x8, uh);
  mid = __builtin_bit_cast(bf16x8, um);
  lo = __builtin_bit_cast(bf16x8, ul);
}
// acc(16x16) += A(16 x 32) B(32 x 16), A given as this lane's 8 f32 values, B already split
__device__ __forceinline__ void mfma16_bf16x3(const f32x4 &a0, const f32x4 &a1, const bf16x8 &bh, const bf16x8 &bm, const bf16x8 &bl,
                                              f32x4 &acc) {
  bf16x8 ah, am, al;
  split3x8(a0, a1, ah, am, al);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
}

// -----------------------------------------------------------------------------------------------------------------
// k_stl_pack: the parameter-only preparation as a kernel of its own (routes where the sampling kernel carries no riders):
// blocks [0, d/64) invert the diagonal blocks, the rest re-lay the off-diagonal blocks (stl_dinv.h).
// Recursive doubling inside LDS: with inverses of the b x b diagonal sub-blocks in place,
//     [A 0; C B]^{-1} = [A^{-1} 0; -B^{-1} (C A^{-1}) B^{-1}]
// gives the 2b x 2b ones from two b x b x b products (all pairs and all outputs in parallel over the threads):
// 87 k MACs per block instead of a 64-step substitution chain per column.
// -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stl_pack(int d, const float *C, unsigned *pack) {
  __shared__ float sm[3 * 64 * 65];
  if ((int)blockIdx.x < (d >> 6)) stl_dinv64_block<256>(d, C, pack, blockIdx.x, sm);
  else stl_pack_block<256>(d, C, pack, (int)blockIdx.x - (d >> 6));
}

// -----------------------------------------------------------------------------------------------------------------
// k_stl_solve64: T^T X = R for the n x n diagonal sub-block T = C[r0 : r0 + n, r0 : r0 + n], 16 right-hand-side columns per
// workgroup, NB = n / 64 block steps bottom up.  The eight waves have FIXED ROLES and the step loop is fully unrolled, so every
// address, every prefetch distance and every s_waitcnt count is a compile-time constant (a first version with data-dependent
// roles compiled to vmcnt(0) at every branch merge: 3 us per step, all of it exposed load latency):
//   * chain waves 0-3 (wave q owns the 16-row tile q of every block) walk the dependency chain, step J:
//       (a) u  = T[J+1, J]^T-tile . X_{J+1}                 12 MFMAs, X block from LDS, operand planes requested PD steps ahead
//       (b) r  = E_J - P_J - u                               P_J: partial tile left in LDS by bulk wave q (updates from K >= J + 2)
//       (c) r split to LDS (B-fragment order)                | barrier A_J
//       (e) x  = DinvT_J (rows of the tile) . R_J            12 MFMAs; X split to LDS, x stored / added into W   | barrier B_J
//     Everything they multiply with arrives ALREADY SPLIT into bf16 planes in fragment order (stl_dinv.h): no VALU work on the
//     operands, six coalesced 1 KiB loads per product.
//   * bulk waves 4-7 (wave 4 + q owns tile q of the accumulators of blocks 0 .. NB - 3) apply X_K to the blocks I <= K - 2, the
//     most urgent tile (I = K - 2) first, and hand P_{K-2} to the chain through LDS (lane-to-lane, 16 bytes per lane).  Their C
//     tiles are f32 in fragment order, DEPTH tiles in flight in registers, split in the wave (88 VALU + 12 MFMAs per tile).
// What bounds it (tools/stl_stamps.py prints the per-step shader-clock stamps, tools/ubench_cu_stream.hip the per-CU streaming
// rates): a workgroup pulls its whole triangle (n = 512: 336 KiB of bulk tiles + 384 KiB of planes) through ONE CU, ~30 B/clk;
// the early windows are bulk-bound (6, 5, 4 tiles of ~650 cycles each), the late ones chain-bound (~1400 cycles per step).
// k slots: MFMA m (K = 32) takes rows 32 m .. 32 m + 31 of the block; lane group g = lane / 16 supplies rows
// {32 m + 4 g + r} and {32 m + 16 + 4 g + r}, r < 4 -- exactly the rows an accumulator lane of tiles 2 m and 2 m + 1 holds.
// -----------------------------------------------------------------------------------------------------------------
constexpr int kStlMaxJobs = 12;   // three per context, four lane-batched contexts (api_batch.hip)
struct StlJob {
  int r0, nwg;           // the n x n system T = C[r0 : r0 + n, r0 : r0 + n]; workgroups of this job (16 right-hand-side columns each)
  const unsigned *pack;  // the packed operands of stl_dinv.h (pivot inverses, chain blocks, bulk blocks of both halves) of THIS job's scale matrix
  const float *rhs;      // R(i, m) = rhs[i * rs_i + m * ld_rhs]   (i < n: the pointer is at row 0 of this system)
  long rs_i, ld_rhs;
  float *X;              // optional: X(i, m) -> X[i * xs_i + m * ld_x]
  long xs_i, ld_x;
  float *W;              // optional: W[i + m * ld_w] += X(i, m)   (pointer at row 0 of this system)
  int ld_w;
  int w_set;             // W = X instead of W += X
};
struct StlSolveArgs {
  int d, n, njobs;
  StlJob job[kStlMaxJobs];   // independent solves side by side in one launch: blockIdx.x walks job 0's workgroups, then job 1's, ...
  unsigned *stamps;      // developer (-DMIVI_DEV, MIVI_STL_STAMPS): shader-clock stamps of workgroup 0, [role 2][step 16][4]
};

// acc(16x16) += A(16 x 32) B(32 x 16), both already split
__device__ __forceinline__ void mfma16_pre(const bf16x8 &ah, const bf16x8 &am, const bf16x8 &al, const bf16x8 &bh, const bf16x8 &bm,
                                           const bf16x8 &bl, f32x4 &acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
}

template <int NB>
__global__ __launch_bounds__(512) void k_stl_solve64(StlSolveArgs a) {
  constexpr int PD = 2;                                   // chain operands are requested PD steps ahead
  constexpr int IMG = 3 * 2 * 64 * 4;                     // one pivot-block image: [plane 3][m 2][lane 64] x 16 bytes
  constexpr int P_OFF = 2 * IMG, STAMP_OFF = P_OFF + 2 * 4 * 256;
  __shared__ __attribute__((aligned(16))) float lds[STAMP_OFF + 128];
  unsigned *Rimg = reinterpret_cast<unsigned *>(lds), *Ximg = Rimg + IMG;
  float *Pb = lds + P_OFF;                                // [slot 2][q 4][lane 64][4]
  const int tid = threadIdx.x, lane = tid & 63, n16 = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = w & 3;
  const int d = a.d;
  int wg = blockIdx.x, ji = 0;                            // which job this workgroup belongs to (uniform)
  while (ji + 1 < a.njobs && wg >= a.job[ji].nwg) { wg -= a.job[ji].nwg; ++ji; }
  const StlJob &jb = a.job[ji];
  const int r0 = jb.r0;
  const int col = wg * 16 + n16;
  const unsigned *Dp = jb.pack + (size_t)(r0 >> 6) * STL_PLANE_BLOCK;
  const unsigned *Cp = jb.pack + (size_t)(d >> 6) * STL_PLANE_BLOCK + (r0 ? stl_solve_units(NB) : 0);   // crit blocks, then the bulk sequence
#ifdef MIVI_DEV
  unsigned *stp = reinterpret_cast<unsigned *>(lds + STAMP_OFF);   // developer: [role 2][step 8][4]
  const bool stamping = a.stamps != nullptr && blockIdx.x == 0 && q == 0 && lane == 0;
  auto stamp = [&](int r, int J, int k) {
    if (stamping && J < 8) stp[(r * 8 + J) * 4 + k] = (unsigned)__builtin_readcyclecounter();
  };
#else
  constexpr bool stamping = false;
  unsigned *stp = nullptr;
  auto stamp = [](int, int, int) {};
#endif

  if (w < 4) {
    // ------------------------------------------------ chain ------------------------------------------------
    // Two copies of the whole chain, chosen once per workgroup: unit row stride (sample columns: one 16-byte load / store per
    // step) and strided (the rows of C21 in, F transposed out: four scalar accesses).  A chain wave's memory instructions queue
    // behind the bulk waves' streaming loads, so every extra one is paid in full (four scalar loads for everybody: 13.5 -> 17 us);
    // a branch per access instead of two copies would cost the compile-time s_waitcnt counts.
    auto chain = [&](auto trc) {
    constexpr bool TR = decltype(trc)::value;
    __builtin_amdgcn_s_setprio(3);
    const unsigned *Dg = Dp + q * 1536 + lane * 4;               // + J * 6144 + (m * 3 + plane) * 256
    const unsigned *Cg = Cp + q * 1536 + lane * 4;
    // right-hand side at the job's row stride (1 for sample columns; d for the rows of C21 that are the columns of C21^T)
    const long rs = jb.rs_i;
    const float *Eg = jb.rhs + (size_t)col * jb.ld_rhs + (size_t)(16 * q + 4 * g) * rs;          // + (64 J + t) rs
    float *const Wb = jb.W;
    const int w_set = jb.w_set, ld_w = jb.ld_w;
    const float *Wg = Wb ? Wb + (size_t)col * ld_w + 16 * q + 4 * g : nullptr;
    float *const Xo = jb.X;
    const long xs = jb.xs_i, ld_x = jb.ld_x;
    f32x4 ef[NB], wf[NB];
    bf16x8 df[NB][6], cf[NB][6];
    auto request = [&](int J) {
      if constexpr (TR) {
#pragma unroll
        for (int t = 0; t < 4; ++t) ef[J][t] = Eg[(size_t)(64 * J + t) * rs];
      } else ef[J] = *(const f32x4 *)(Eg + 64 * J);
      if (Wg && !w_set) wf[J] = *(const f32x4 *)(Wg + 64 * J);
#pragma unroll
      for (int u = 0; u < 6; ++u) df[J][u] = *(const bf16x8 *)(Dg + (size_t)J * STL_PLANE_BLOCK + u * 256);
      if (J < NB - 1) {
#pragma unroll
        for (int u = 0; u < 6; ++u) cf[J][u] = *(const bf16x8 *)(Cg + (size_t)J * STL_PLANE_BLOCK + u * 256);
      }
    };
#pragma unroll
    for (int J = NB - 1; J > NB - 1 - PD && J >= 0; --J) request(J);   // (compile-time trip count: plain indices)
    const int slot = ((q >> 1) * 64 + lane) * 4 + (q & 1) * 2;   // [m = q / 2][lane], half q & 1 (8 bytes)
    static_for<0, NB>([&](auto jc) {
      constexpr int J = NB - 1 - decltype(jc)::value;
      stamp(0, J, 0);
      if constexpr (J - PD >= 0) request(J - PD);
      f32x4 r = ef[J];
      if constexpr (J < NB - 1) {
        f32x4 u0 = {0.f, 0.f, 0.f, 0.f}, u1 = {0.f, 0.f, 0.f, 0.f};
        {
          const bf16x8 xh = *(const bf16x8 *)(Ximg + 0 * 512 + lane * 4);
          const bf16x8 xm = *(const bf16x8 *)(Ximg + 1 * 512 + lane * 4);
          const bf16x8 xl = *(const bf16x8 *)(Ximg + 2 * 512 + lane * 4);
          mfma16_pre(cf[J][0], cf[J][1], cf[J][2], xh, xm, xl, u0);
        }
        {
          const bf16x8 xh = *(const bf16x8 *)(Ximg + 0 * 512 + (64 + lane) * 4);
          const bf16x8 xm = *(const bf16x8 *)(Ximg + 1 * 512 + (64 + lane) * 4);
          const bf16x8 xl = *(const bf16x8 *)(Ximg + 2 * 512 + (64 + lane) * 4);
          mfma16_pre(cf[J][3], cf[J][4], cf[J][5], xh, xm, xl, u1);
        }
        r -= u0 + u1;
      }
      if constexpr (J < NB - 2) r -= *(const f32x4 *)(Pb + ((J & 1) * 4 + q) * 256 + lane * 4);
      {
        u32x2v h2, m2, l2;
        split3x4(r, h2, m2, l2);
        *(u32x2v *)(Rimg + 0 * 512 + slot) = h2;
        *(u32x2v *)(Rimg + 1 * 512 + slot) = m2;
        *(u32x2v *)(Rimg + 2 * 512 + slot) = l2;
      }
      stamp(0, J, 1);
      lds_barrier();   // A_J
      stamp(0, J, 2);
      f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = {0.f, 0.f, 0.f, 0.f};
      {
        const bf16x8 bh = *(const bf16x8 *)(Rimg + 0 * 512 + lane * 4);
        const bf16x8 bm = *(const bf16x8 *)(Rimg + 1 * 512 + lane * 4);
        const bf16x8 bl = *(const bf16x8 *)(Rimg + 2 * 512 + lane * 4);
        mfma16_pre(df[J][0], df[J][1], df[J][2], bh, bm, bl, x0);
      }
      {
        const bf16x8 bh = *(const bf16x8 *)(Rimg + 0 * 512 + (64 + lane) * 4);
        const bf16x8 bm = *(const bf16x8 *)(Rimg + 1 * 512 + (64 + lane) * 4);
        const bf16x8 bl = *(const bf16x8 *)(Rimg + 2 * 512 + (64 + lane) * 4);
        mfma16_pre(df[J][3], df[J][4], df[J][5], bh, bm, bl, x1);
      }
      const f32x4 x = x0 + x1;
      if constexpr (J > 0) {
        u32x2v h2, m2, l2;
        split3x4(x, h2, m2, l2);
        *(u32x2v *)(Ximg + 0 * 512 + slot) = h2;
        *(u32x2v *)(Ximg + 1 * 512 + slot) = m2;
        *(u32x2v *)(Ximg + 2 * 512 + slot) = l2;
      }
      const int row = 64 * J + 16 * q + 4 * g;
      if (Xo) {                                                    // (written through: see store16_wt)
        if constexpr (TR) {
          float *xp = Xo + (size_t)col * ld_x + (size_t)row * xs;
#pragma unroll
          for (int t = 0; t < 4; ++t) store4_wt(xp + (size_t)t * xs, x[t]);
        } else store16_wt(Xo + (size_t)col * ld_x + row, x);
      }
      if (Wb) {
        const f32x4 wo = w_set ? x : wf[J] + x;
        store16_wt(Wb + (size_t)col * ld_w + row, wo);
      }
      stamp(0, J, 3);
      if constexpr (J > 0) lds_barrier();   // B_J
    });
    };
    if (jb.rs_i == 1 && jb.xs_i == 1) chain(std::false_type{});
    else chain(std::true_type{});
    if (stamping)
      for (int i = 0; i < 32; ++i) a.stamps[i] = stp[i];
    (void)stp;
    return;
  }
  // -------------------------------------------------- bulk --------------------------------------------------
  // update sequence s: K = NB - 1 .. 2, I = K - 2 .. 0 -- the order the packed buffer stores the blocks in: wave q streams the
  // 4 KiB tile [s][q] with plain 16-byte loads, DEPTH tiles (16 registers each) in flight.  (LDS-DMA would save the registers, but
  // one wave gets only ~1 KiB per 250 cycles through it -- tools/ubench_cu_stream.hip: 36 GB/s per CU with four waves against
  // 77 GB/s with sixteen plain loads in flight per wave -- and this kernel is bound by what one CU can pull.)
  constexpr int S = stl_seq_len(NB), DEPTH = 6;
  const float *Bg = reinterpret_cast<const float *>(Cp + (size_t)(NB - 1) * STL_PLANE_BLOCK) + q * 1024 + lane * 4;
  f32x4 fb[S > 0 ? S : 1][4];
  auto issue = [&](auto sc_) {   // request tile s of the sequence
    constexpr int s = decltype(sc_)::value;
    if constexpr (s < S) {
#pragma unroll
      for (int p = 0; p < 4; ++p) fb[s][p] = *(const f32x4 *)(Bg + (size_t)s * STL_F32_BLOCK + p * 256);
    }
  };
  static_for<0, DEPTH>(issue);
  constexpr int NACC = NB > 2 ? NB - 2 : 1;
  f32x4 acc[NACC];
#pragma unroll
  for (int I = 0; I < NACC; ++I) acc[I] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 xb[2][3];
  auto update = [&](auto sc_) {   // tile s of the sequence: acc[I] += T[K, I]^T-tile . X_K
    constexpr int s = decltype(sc_)::value;
    constexpr int I = stl_seq_I(NB, s);
    mfma16_bf16x3(fb[s][0], fb[s][1], xb[0][0], xb[0][1], xb[0][2], acc[I]);
    mfma16_bf16x3(fb[s][2], fb[s][3], xb[1][0], xb[1][1], xb[1][2], acc[I]);
    issue(std::integral_constant<int, s + DEPTH>{});
  };
  static_for<0, NB>([&](auto jc) {
    constexpr int J = NB - 1 - decltype(jc)::value;
    stamp(1, J, 0);
    lds_barrier();   // A_J
    stamp(1, J, 1);
    if constexpr (J + 1 <= NB - 1 && J + 1 >= 2) {   // the rest of the X_{J+1} updates
      constexpr int K = J + 1, nf = K / 2, s0 = stl_seq_first(NB, K);
      static_for<s0 + nf, s0 + K - 1>(update);
    }
    stamp(1, J, 2);
    if constexpr (J > 0) lds_barrier();   // B_J
    stamp(1, J, 3);
    if constexpr (J >= 2) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) xb[m][pl] = *(const bf16x8 *)(Ximg + pl * 512 + (m * 64 + lane) * 4);
      constexpr int nf = J / 2, s0 = stl_seq_first(NB, J);   // the first nf of the J - 1 tiles, I = J - 2 first
      update(std::integral_constant<int, s0>{});
      *(f32x4 *)(Pb + (((J - 2) & 1) * 4 + q) * 256 + lane * 4) = acc[J - 2];
      static_for<s0 + 1, s0 + nf>(update);
    }
  });
  if (stamping)
    for (int i = 32; i < 64; ++i) a.stamps[i] = stp[i];
}

// -----------------------------------------------------------------------------------------------------------------
// k_stl_update32: R(i, m) = E(i, m) - sum_k A(i, k) X(k, m) [+ R(i, m)]  for i < n_i, k < n_k, with A(i, k) = A[k + i * lda] -- the
// combination X1 = Y1 - F^T X2 of the two halves (A = F as k_stl_solve64's third job stores it, E = Y1, R = the rows of W).
// One 32 x 32 tile per workgroup, eight waves split K into 32-k sub-stages and stage their own operands through a private LDS
// buffer -- both operands are K-MAJOR here (a column of F, a column of X), so both images are [row][32 k] with the 16-byte chunks
// XOR-swizzled and both fragments are b128 reads (k_fr_prod32's B side).
// -----------------------------------------------------------------------------------------------------------------
struct StlUpdArgs {
  int n_i, n_k;
  const float *A; long lda;
  const float *X; int ld_x;
  const float *E; int ld_e;
  float *R; int ld_r;      // R[i + m * ld_r]
  int accumulate;          // R += (E - A X) instead of R = E - A X
  int ncb;
};

__device__ __forceinline__ void stl_update32_body(const StlUpdArgs &a) {
  constexpr int NW = 8, SUB = 32, LDC = 36;
  constexpr int WAVE_F = 2 * SUB * 32;
  constexpr int EPI = NW * 32 * LDC;
  constexpr int MAIN = (NW * WAVE_F > EPI) ? NW * WAVE_F : EPI;
  __shared__ __attribute__((aligned(16))) float lds[MAIN];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rb = (int)blockIdx.x / a.ncb, cb = (int)blockIdx.x % a.ncb;
  const int row0 = rb * 32, col0 = cb * 32;
  const int nst = a.n_k / SUB;
  const int t_beg = (w * nst) / NW, t_end = ((w + 1) * nst) / NW;
  float *buf = lds + w * WAVE_F;
  const float *Ag[4], *Bg[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int n = 8 * p + (lane >> 3);
    const int ch = 4 * ((lane & 7) ^ ((n >> 1) & 7));
    Ag[p] = a.A + (size_t)(row0 + n) * a.lda + ch;
    Bg[p] = a.X + (size_t)(col0 + n) * a.ld_x + ch;
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int f_off = l31 * 32, f_swz = h ^ ((l31 >> 1) & 7);
  for (int t = t_beg; t < t_end; ++t) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      MIVI_GLDS16(Ag[p] + t * SUB, buf + p * 256);
      MIVI_GLDS16(Bg[p] + t * SUB, buf + SUB * 32 + p * 256);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4 aq[4], bq[4];
#pragma unroll
    for (int s8 = 0; s8 < 4; ++s8) {
      aq[s8] = *(const f32x4 *)(buf + f_off + 4 * ((2 * s8) ^ f_swz));
      bq[s8] = *(const f32x4 *)(buf + SUB * 32 + f_off + 4 * ((2 * s8) ^ f_swz));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      bf16x8 ah, am, al, bh, bm, bl;
      split3x8(aq[2 * gq], aq[2 * gq + 1], ah, am, al);
      split3x8(bq[2 * gq], bq[2 * gq + 1], bh, bm, bl);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
  }
  __builtin_amdgcn_s_barrier();
  float *Cs = lds;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
    *(f32x4 *)(Cs + (w * 32 + l31) * LDC + 8 * q + 4 * h) = v;
  }
  lds_barrier();
  if (tid < 256) {
    const int ei4 = 4 * (tid & 7), en = tid >> 3;
    f32x4 v = *(const f32x4 *)(Cs + en * LDC + ei4);
#pragma unroll
    for (int k2 = 1; k2 < NW; ++k2) v += *(const f32x4 *)(Cs + (k2 * 32 + en) * LDC + ei4);
    const f32x4 e = *(const f32x4 *)(a.E + (size_t)(col0 + en) * a.ld_e + row0 + ei4);
    float *rp = a.R + (size_t)(col0 + en) * a.ld_r + row0 + ei4;
    f32x4 ro = e - v;
    if (a.accumulate) ro += *(const f32x4 *)rp;
    store16_wt(rp, ro);
  }
}

struct StlUpdMulti { StlUpdArgs lane[4]; };   // lane-batched contexts: blockIdx.y = lane
__global__ __launch_bounds__(512) void k_stl_update32(StlUpdArgs a) { stl_update32_body(a); }
__global__ __launch_bounds__(512) void k_stl_update32m(StlUpdMulti m) { stl_update32_body(m.lane[blockIdx.y]); }

// one context's recorded STL term (lane-batched estimates: launch_stl2 records, launch_lanes_stl issues the lanes together)
struct StlSink {
  StlSolveArgs solve;
  StlUpdArgs upd;
  int upd_grid, n;
};
StlSink *stl_sinks_alloc(int n) { return new StlSink[n](); }
void stl_sinks_free(StlSink *s) { delete[] s; }
void stl_sink_reset(StlSink *s, int lane) { s[lane].n = 0; }
int stl_sink_count(const StlSink *s, int lane) { return s[lane].n; }

// -----------------------------------------------------------------------------------------------------------------
bool stl2_shape_ok(const mivi_ctx *c, int M) {
  static const bool off = getenv("MIVI_STL_GEN1") != nullptr;
  const int d = c->cfg.d;
  return !off && c->cfg.dtype == MIVI_F32 && c->cfg.family == MIVI_FULLRANK && (d == 256 || d == 512 || d == 1024 || d == 2048) &&
         M % 32 == 0 && M > 0;
}

static void launch_solve(mivi_ctx *c, const StlSolveArgs &a) {
  const int nb = a.n / 64;
  int nwg = 0;
  for (int j = 0; j < a.njobs; ++j) nwg += a.job[j].nwg;
  const dim3 grid(nwg), block(512);
  if (nb == 2) hipLaunchKernelGGL(k_stl_solve64<2>, grid, block, 0, c->stream, a);
  else if (nb == 4) hipLaunchKernelGGL(k_stl_solve64<4>, grid, block, 0, c->stream, a);
  else if (nb == 8) hipLaunchKernelGGL(k_stl_solve64<8>, grid, block, 0, c->stream, a);
  else hipLaunchKernelGGL(k_stl_solve64<16>, grid, block, 0, c->stream, a);
}

// W += C^{-T} eps for the current estimate (W: d x M, ld d; eps: ld dP).  Needs c->stl_F (the packed operands) and c->stl_X
// ((d M + d^2/4) floats: X2, then Y1, then F).
void launch_stl2(mivi_ctx *c, const void *params, int M, bool dinv_done, const void *rhs, void *out, bool overwrite) {
  const int d = c->cfg.d, n = d / 2;
  const float *C = (const float *)params + d;
  unsigned *pack = (unsigned *)c->stl_F.p;
  float *Xb = (float *)c->stl_X.p, *Y1 = Xb + (size_t)n * M, *F = Y1 + (size_t)n * M;
  const float *eps = rhs ? (const float *)rhs : (const float *)c->eps[c->cur].p;   // right-hand sides, ld dP
  float *Wout = out ? (float *)out : (float *)c->W.p;                               // X is ADDED here, ld d
  if (!dinv_done) hipLaunchKernelGGL(k_stl_pack, dim3(d / 64 + stl_pack_riders(d)), dim3(256), 0, c->stream, d, C, pack);
  StlSolveArgs s{};
  s.d = d; s.n = n; s.njobs = 3;
  for (int j = 0; j < 3; ++j) s.job[j].pack = pack;
  // job 0 -- lower half: C22^T X2 = E2, X2 also straight into the rows n .. d of W
  StlJob &j0 = s.job[0];
  j0.r0 = n; j0.nwg = M / 16; j0.rhs = eps + n; j0.rs_i = 1; j0.ld_rhs = c->dP; j0.X = Xb; j0.xs_i = 1; j0.ld_x = n;
  j0.W = Wout + n; j0.ld_w = d; j0.w_set = overwrite ? 1 : 0;
  // job 1 -- upper half without the coupling: C11^T Y1 = E1
  StlJob &j1 = s.job[1];
  j1.r0 = 0; j1.nwg = M / 16; j1.rhs = eps; j1.rs_i = 1; j1.ld_rhs = c->dP; j1.X = Y1; j1.xs_i = 1; j1.ld_x = n;
  // job 2 -- the coupling, parameters only: C11^T F^T = C21^T.  Column m of C21^T is row n + m of C (stride d along it); the
  // solution is stored transposed, F[k + i n] = F^T(i, k): k-major rows, the layout k_stl_update32 stages
  StlJob &j2 = s.job[2];
  j2.r0 = 0; j2.nwg = n / 16; j2.rhs = C + n; j2.rs_i = d; j2.ld_rhs = 1; j2.X = F; j2.xs_i = n; j2.ld_x = 1;
  StlUpdArgs u{};   // rows 0 .. n of W:  X1 = Y1 - F^T X2
  u.n_i = n; u.n_k = n; u.A = F; u.lda = n; u.X = Xb; u.ld_x = n; u.E = Y1; u.ld_e = n;
  u.R = Wout; u.ld_r = d; u.accumulate = overwrite ? 0 : 1; u.ncb = M / 32;
  if (c->stl_sink) {   // lane-batched estimates (api_batch.hip): record; the driver issues the lanes' solves and products as one launch each
    StlSink &sk = ((StlSink *)c->stl_sink)[c->lane_id];
    if (sk.n == 0) { sk.solve = s; sk.upd = u; sk.upd_grid = (n / 32) * (M / 32); }
    ++sk.n;
    return;
  }
#ifdef MIVI_DEV
  static const bool stamps = getenv("MIVI_STL_STAMPS") != nullptr;
  if (stamps) s.stamps = (unsigned *)((char *)c->stl_X.p + c->stl_X.bytes - 4096);
#endif
  launch_solve(c, s);
  hipLaunchKernelGGL(k_stl_update32, dim3((n / 32) * (M / 32)), dim3(512), 0, c->stream, u);
#ifdef MIVI_DEV
  if (stamps) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(c->stream, &cs);
    if (cs == hipStreamCaptureStatusNone) {
      unsigned h[64];
      (void)hipStreamSynchronize(c->stream);
      (void)hipMemcpy(h, s.stamps, sizeof h, hipMemcpyDeviceToHost);
      for (int role = 0; role < 2; ++role) {
        fprintf(stderr, "[stl stamps] job 0 %s:", role ? "bulk " : "chain");
        const unsigned t0 = h[(0 * 8 + 7) * 4];
        for (int J = 7; J >= 0; --J)
          for (int i = 0; i < 4; ++i) fprintf(stderr, "%s%u", i ? " " : " | ", h[(role * 8 + J) * 4 + i] - t0);
        fprintf(stderr, "\n");
      }
    }
  }
#endif
}

// the recorded STL terms of `lanes` contexts: ONE solve launch with all their jobs, ONE combining product (blockIdx.y = lane).
// The lanes are estimates at the SAME parameters, so the parameter-only coupling solve (job 2: F) runs for lane 0 only and every lane's
// product reads lane 0's F; with_F = false (a later step of the same batch): nobody solves for it again.
bool launch_lanes_stl(mivi_ctx *c, StlSink *sk, int lanes, bool with_F) {
  if (lanes < 1 || lanes > 4) return false;
  StlSolveArgs s = sk[0].solve;
  StlUpdMulti m;
  s.njobs = 0;
  for (int l = 0; l < lanes; ++l) {
    if (sk[l].n != 1 || sk[l].solve.njobs != 3 || sk[l].solve.d != s.d || sk[l].solve.n != s.n || sk[l].upd_grid != sk[0].upd_grid) return false;
    for (int j = 0; j < 3; ++j)
      if (j < 2 || (l == 0 && with_F)) s.job[s.njobs++] = sk[l].solve.job[j];
    m.lane[l] = sk[l].upd;
    m.lane[l].A = sk[0].upd.A;
  }
  launch_solve(c, s);
  hipLaunchKernelGGL(k_stl_update32m, dim3(sk[0].upd_grid, lanes), dim3(512), 0, c->stream, m);
  return true;
}

}  // namespace mivi
