// Sticking-the-landing term of the full-rank family, second generation:  W += X,  C^T X = eps.
//
// Reference: `logpdf(q_stop, z)` inside StickingTheLandingEntropy (src/algorithms/entropy.jl:57-65, 80-90) differentiates
// through C^{-1}(z - mu) (src/families/location_scale.jl:59-63); in the closed-form VJP that is the extra term C^{-T} eps in W.
//
// The first generation (kernels_fullrank.hip, k_stl_solve_la16) gives every 16 sample columns one workgroup that walks all
// d/32 block rows: 16 CUs busy, 95 us at d = 1024.  A triangular solve is sequential in its block rows, and a step of the
// chain is only cheap while it stays inside one CU (LDS + s_barrier: a few hundred cycles; across CUs: a kernel boundary).  So:
//   * the flops move out of the chain: one level of recursion,  [C11 0; C21 C22]^T [X1; X2] = [E1; E2]  =>
//         X2 = C22^{-T} E2          (half-size solve)
//         R1 = E1 - C21^T X2        (a plain (d/2 x d/2) x (d/2 x M) product on the whole chip: k_stl_update32)
//         X1 = C11^{-T} R1          (half-size solve)
//     half of the work becomes a GEMM, each solve streams a quarter of C;
//   * the chain itself (k_stl_solve64) runs on 64-row blocks with PRE-INVERTED diagonal blocks (k_stl_dinv64), 16 columns per
//     workgroup, the residual tiles resident in MFMA accumulators (a wave owns the 16-row tiles t = w, w + 8, ...), the pivot
//     block exchanged through LDS already split into bf16 pieces in fragment order; products on v_mfma_f32_16x16x32_bf16 with the
//     exact three-way split (kernels_fullrank_lds.hip); the C fragments of the updates come straight from L2 with 16-byte
//     loads issued one chain step ahead (every C element is used by exactly one wave: nothing to share through LDS).
// d in {256, 512, 1024, 2048}, M % 32 == 0; other shapes keep the first-generation kernels.
#include <cstdlib>

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
// k_stl_dinv64: DinvT[J][i * 64 + k] = (C_JJ^{-1})[k, i] for every 64 x 64 diagonal block J, one workgroup per block.
// Recursive doubling inside LDS: with inverses of the b x b diagonal sub-blocks in place,
//     [A 0; C B]^{-1} = [A^{-1} 0; -B^{-1} (C A^{-1}) B^{-1}]
// gives the 2b x 2b ones from two b x b x b products (all pairs and all outputs in parallel over the 256 threads):
// 87 k MACs per block instead of a 64-step substitution chain per column.
// -----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stl_dinv64(int d, const float *C, float *DinvT) {
  __shared__ float sm[3 * 64 * 65];
  stl_dinv64_block<256>(d, C, DinvT, blockIdx.x, sm);
}

// -----------------------------------------------------------------------------------------------------------------
// k_stl_solve64: T^T X = R for the n x n diagonal sub-block T = C[r0 : r0 + n, r0 : r0 + n], 16 right-hand-side columns per
// workgroup, 8 waves, wave w owns the 16-row tiles t = w + 8 j (j < TPW = n / 128).  Block J = rows 64 J .. 64 J + 63 = tiles
// 4 J .. 4 J + 3, owned by waves 0-3 (J even) / 4-7 (J odd).  Back substitution over the blocks, bottom up:
//   (1) owners of block J: residual tile E - acc, split, to LDS in B-fragment order         | barrier
//   (2) owners: X tile = DinvT_J (rows of the tile) . R_J  (12 MFMAs), W += X / X stored, X split to LDS   | barrier
//   (3) every wave: acc_t += C[J, t]^T X_J for its tiles t above block J (12 MFMAs per tile; the C fragments were requested
//       during the previous step)
// k slots: MFMA m (K = 32) takes rows 32 m .. 32 m + 31 of the block; lane group g = lane / 16 supplies rows
// {32 m + 4 g + r} and {32 m + 16 + 4 g + r}, r < 4 -- exactly the rows an accumulator lane of tiles 2 m and 2 m + 1 holds, and
// two 16-byte runs of a column of C / a row of DinvT.
// -----------------------------------------------------------------------------------------------------------------
struct StlSolveArgs {
  int d, n, r0;
  const float *C;        // params + d, column-major, ld = d
  const float *DinvT;    // [d / 64][64 * 64]
  const float *rhs;      // R(i, m) = rhs[(rhs_r0 + i) + m * ld_rhs]
  int rhs_r0, ld_rhs;
  float *X;              // optional: X(i, m) -> X[i + m * ld_x] (rows of this system only)
  int ld_x;
  float *W;              // optional: W[(r0 + i) + m * ld_w] += X(i, m)
  int ld_w;
  int knock;             // developer knock-outs (MIVI_STL_KNOCK): 1 no C-fragment loads, 2 no update MFMAs, 4 no pivot MFMAs
};

template <int TPW>
__global__ __launch_bounds__(512) void k_stl_solve64(StlSolveArgs a) {
  // LDS: pivot block in B-fragment order, [plane 3][m 2][lane 64] x 16 bytes, one image for R and one for X
  __shared__ __attribute__((aligned(16))) unsigned img[2][3 * 2 * 64 * 4];
  const int tid = threadIdx.x, lane = tid & 63, n16 = lane & 15, g = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int d = a.d, nb = a.n >> 6;
  const int col = blockIdx.x * 16 + n16;
  const float *Cs = a.C + (size_t)a.r0 * d + a.r0;    // the sub-block: Cs[i + j * d]

  f32x4 E[TPW], acc[TPW], Wt[TPW];   // right-hand side, accumulated updates, and the W tile X is added to (off the chain)
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int t = w + 8 * j;
    E[j] = *(const f32x4 *)(a.rhs + (size_t)col * a.ld_rhs + a.rhs_r0 + 16 * t + 4 * g);
    if (a.W) Wt[j] = *(const f32x4 *)(a.W + (size_t)col * a.ld_w + a.r0 + 16 * t + 4 * g);
    acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // C fragments of the update by block J for tile j: column (16 t + i) of Cs, rows 64 J + {32 m + 4 g, 32 m + 16 + 4 g} (+ r)
  f32x4 cf[TPW][4];
  auto load_cf = [&](int J) {
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int t = w + 8 * j;
      if ((t >> 2) < J && !(a.knock & 1)) {
        const float *p = Cs + (size_t)(16 * t + n16) * d + 64 * J + 4 * g;
        cf[j][0] = *(const f32x4 *)(p);
        cf[j][1] = *(const f32x4 *)(p + 16);
        cf[j][2] = *(const f32x4 *)(p + 32);
        cf[j][3] = *(const f32x4 *)(p + 48);
      }
    }
  };
  f32x4 df[4];   // DinvT fragments of the owner's tile of the current block
  auto load_df = [&](int J) {
    const int q = w & 3;
    const float *p = a.DinvT + (size_t)((a.r0 >> 6) + J) * 4096 + (size_t)(16 * q + n16) * 64 + 4 * g;
    df[0] = *(const f32x4 *)(p);
    df[1] = *(const f32x4 *)(p + 16);
    df[2] = *(const f32x4 *)(p + 32);
    df[3] = *(const f32x4 *)(p + 48);
  };
  if (((nb - 1) & 1) == (w >> 2)) load_df(nb - 1);
  load_cf(nb - 1);

  for (int J = nb - 1; J >= 0; --J) {
    const bool owner = (J & 1) == (w >> 2);
    const int q = w & 3;                    // owner: its tile inside the block (tile 4 J + q, local index j = (4 J + q) / 8)
    const int jo = (4 * J + q) >> 3;
    unsigned *Rimg = img[0], *Ximg = img[1];
    const int slot = ((q >> 1) * 64 + lane) * 4 + (q & 1) * 2;   // [m = q / 2][lane], half q & 1 (8 bytes)
    if (owner) {   // (1) residual tile -> LDS, split
      f32x4 r;
#pragma unroll
      for (int j = 0; j < TPW; ++j)
        if (j == jo) r = E[j] - acc[j];
      u32x2v h2, m2, l2;
      split3x4(r, h2, m2, l2);
      *(u32x2v *)(Rimg + 0 * 512 + slot) = h2;
      *(u32x2v *)(Rimg + 1 * 512 + slot) = m2;
      *(u32x2v *)(Rimg + 2 * 512 + slot) = l2;
    }
    lds_barrier();
    if (owner) {   // (2) X tile = DinvT rows . R_J
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const bf16x8 bh = *(const bf16x8 *)(Rimg + 0 * 512 + (m * 64 + lane) * 4);
        const bf16x8 bm = *(const bf16x8 *)(Rimg + 1 * 512 + (m * 64 + lane) * 4);
        const bf16x8 bl = *(const bf16x8 *)(Rimg + 2 * 512 + (m * 64 + lane) * 4);
        if (!(a.knock & 4)) mfma16_bf16x3(df[2 * m], df[2 * m + 1], bh, bm, bl, x);
      }
      u32x2v h2, m2, l2;
      split3x4(x, h2, m2, l2);
      *(u32x2v *)(Ximg + 0 * 512 + slot) = h2;
      *(u32x2v *)(Ximg + 1 * 512 + slot) = m2;
      *(u32x2v *)(Ximg + 2 * 512 + slot) = l2;
      const int row = 64 * J + 16 * q + 4 * g;
      if (a.X) *(f32x4 *)(a.X + (size_t)col * a.ld_x + row) = x;
      if (a.W) {
        f32x4 wv;
#pragma unroll
        for (int j = 0; j < TPW; ++j)
          if (j == jo) wv = Wt[j] + x;
        *(f32x4 *)(a.W + (size_t)col * a.ld_w + a.r0 + row) = wv;
      }
    }
    if (J == 0) break;
    if (((J - 1) & 1) == (w >> 2)) load_df(J - 1);   // next pivot's inverse rows: in flight across the barrier
    lds_barrier();
    // (3) updates with X_J; the fragments of the NEXT step's updates are requested before this step's MFMAs run
    bf16x8 xb[2][3];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) xb[m][pl] = *(const bf16x8 *)(Ximg + pl * 512 + (m * 64 + lane) * 4);
    f32x4 cur[TPW][4];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u) cur[j][u] = cf[j][u];
    load_cf(J - 1);
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
      const int t = w + 8 * j;
      if ((t >> 2) < J && !(a.knock & 2)) {
        mfma16_bf16x3(cur[j][0], cur[j][1], xb[0][0], xb[0][1], xb[0][2], acc[j]);
        mfma16_bf16x3(cur[j][2], cur[j][3], xb[1][0], xb[1][1], xb[1][2], acc[j]);
      }
    }
  }
}

// -----------------------------------------------------------------------------------------------------------------
// k_stl_update32: R(i, m) = E(i, m) - sum_k C[k0 + k, i0 + i] X(k, m)  for i < n_i, k < n_k (the off-diagonal block C21^T
// applied to the half already solved).  One 32 x 32 tile per workgroup, eight waves split K into 32-k sub-stages and stage
// their own operands through a private LDS buffer -- both operands are K-MAJOR here (a column of C, a column of X), so both
// images are [row][32 k] with the 16-byte chunks XOR-swizzled and both fragments are b128 reads (k_fr_prod32's B side).
// -----------------------------------------------------------------------------------------------------------------
struct StlUpdArgs {
  int d, n_i, n_k, i0, k0;
  const float *C;
  const float *X; int ld_x;
  const float *E; int e_r0, ld_e;
  float *R; int ld_r;      // R[i + m * ld_r]
  int ncb;
};

__global__ __launch_bounds__(512) void k_stl_update32(StlUpdArgs a) {
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
    Ag[p] = a.C + (size_t)(a.i0 + row0 + n) * a.d + a.k0 + ch;
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
    const f32x4 e = *(const f32x4 *)(a.E + (size_t)(col0 + en) * a.ld_e + a.e_r0 + row0 + ei4);
    *(f32x4 *)(a.R + (size_t)(col0 + en) * a.ld_r + row0 + ei4) = e - v;
  }
}

// -----------------------------------------------------------------------------------------------------------------
bool stl2_shape_ok(const mivi_ctx *c, int M) {
  static const bool off = getenv("MIVI_STL_GEN1") != nullptr;
  const int d = c->cfg.d;
  return !off && c->cfg.dtype == MIVI_F32 && c->cfg.family == MIVI_FULLRANK && (d == 256 || d == 512 || d == 1024 || d == 2048) &&
         M % 32 == 0 && M > 0;
}

static void launch_solve(mivi_ctx *c, const StlSolveArgs &a, int M) {
  const int tpw = a.n / 128;
  const dim3 grid(M / 16), block(512);
  if (tpw == 1) hipLaunchKernelGGL(k_stl_solve64<1>, grid, block, 0, c->stream, a);
  else if (tpw == 2) hipLaunchKernelGGL(k_stl_solve64<2>, grid, block, 0, c->stream, a);
  else if (tpw == 4) hipLaunchKernelGGL(k_stl_solve64<4>, grid, block, 0, c->stream, a);
  else hipLaunchKernelGGL(k_stl_solve64<8>, grid, block, 0, c->stream, a);
}

// W += C^{-T} eps for the current estimate (W: d x M, ld d; eps: ld dP).  Needs c->stl_Dinv (d/64 * 4096 floats) and
// c->stl_X (d x M floats: X of the lower half, then the updated right-hand side of the upper half).
void launch_stl2(mivi_ctx *c, const void *params, int M, bool dinv_done) {
  const int d = c->cfg.d, n = d / 2;
  const float *C = (const float *)params + d;
  float *Dinv = (float *)c->stl_Dinv.p;
  float *Xb = (float *)c->stl_X.p, *Rt = Xb + (size_t)n * M;
  const float *eps = (const float *)c->eps[c->cur].p;
  if (!dinv_done) hipLaunchKernelGGL(k_stl_dinv64, dim3(d / 64), dim3(256), 0, c->stream, d, C, Dinv);
  StlSolveArgs s{};
  static const int knock = getenv("MIVI_STL_KNOCK") ? atoi(getenv("MIVI_STL_KNOCK")) : 0;
  s.knock = knock;
  s.d = d; s.n = n; s.C = C; s.DinvT = Dinv;
  // lower half: C22^T X2 = E2
  s.r0 = n; s.rhs = eps; s.rhs_r0 = n; s.ld_rhs = c->dP; s.X = Xb; s.ld_x = n; s.W = (float *)c->W.p; s.ld_w = d;
  launch_solve(c, s, M);
  if (getenv("MIVI_STL_TWICE")) { s.W = nullptr; launch_solve(c, s, M); s.W = (float *)c->W.p; }   // developer: does a repeat hit a warm L2?
  // R1 = E1 - C21^T X2
  StlUpdArgs u{};
  u.d = d; u.n_i = n; u.n_k = n; u.i0 = 0; u.k0 = n; u.C = C; u.X = Xb; u.ld_x = n; u.E = eps; u.e_r0 = 0; u.ld_e = c->dP;
  u.R = Rt; u.ld_r = n; u.ncb = M / 32;
  hipLaunchKernelGGL(k_stl_update32, dim3((n / 32) * (M / 32)), dim3(512), 0, c->stream, u);
  // upper half: C11^T X1 = R1
  s.r0 = 0; s.rhs = Rt; s.rhs_r0 = 0; s.ld_rhs = n; s.X = nullptr; s.ld_x = 0;
  launch_solve(c, s, M);
}

}  // namespace mivi
