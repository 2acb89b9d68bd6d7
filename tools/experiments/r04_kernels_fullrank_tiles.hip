// Launch-free optimisation loop for the full-rank family at the north-star shape class (d <= 1024, n_mc = 128 or 256, f32) and the
// diagonal-Gaussian target: ONE persistent kernel whose workgroups OWN tiles of tril(C) for all steps of the call.
//
// Reference semantics per iteration (src/algorithms/common.jl:69-104 -- what `optimize`, src/optimize.jl:64-77, runs): estimate_gradient!
// (src/algorithms/repgradelbo.jl:151-177) with z = mu + tril(C) eps (src/families/location_scale.jl:71-77), W = grad log pi(z),
// d/dC = -(1/M) tril(W eps') - direct diag(1 / C_ii), d/dmu = -(1/M) W 1, Optimisers.update! (Descent / Adam), ClipScale.
//
// Why: a dependent step of the general route is two launches (k_fr_prod32, k_fr_vjp32<FUSED>) whose critical paths are memory round trips,
// not arithmetic -- the heaviest product tile pulls its operands in four sequential stages, the VJP's epilogue reads and writes 12.6 MB of
// parameters and moments -- plus two kernel boundaries: 15.3 us per step at d = 1024, n_mc = 256.  Here nothing that belongs to
// a tile ever leaves its CU: the parameters and Adam moments of a workgroup's tiles live in registers, their operand image in LDS.
//
// Decomposition -- chosen so that every sum is the SAME chain of the same instructions as in the launch-per-step kernels (a trajectory is
// bitwise theirs: tests/test_gpu_optimize.py::test_device_resident_loop_matches_host_loop, test_fullrank_tiles_loop):
//   * k_fr_prod32 cuts the k range of row block rb (32 (rb + 1) columns of tril(C)) into runs of c = ceil((rb + 1) / 8) sub-stages of 32 k, one
//     run per wave, and adds the eight partial tiles in wave order.  Workgroup (rb, g) of this kernel owns RUN g of row block rb: the 32 x 32 c
//     block of tril(C) (up to four 32 x 32 tiles), for all n_mc samples.  205 workgroups at d = 1024, at most eight per row block.
//   * a step, for the workgroups of one row block (no other workgroup is ever waited for):
//       1. product: the run's MFMA chain (mfma_bf16x3 twice per sub-stage, ascending) for every 32-sample block -> partial tiles, stored
//          write-through (sc1) into the row block's exchange area, flag;
//       2. reduce: sample block cb is summed by workgroup cb mod (number of runs): partial tiles added in run order (+ zeros for the idle
//          waves of k_fr_prod32), z = mu + v, W and the value's ell partial (fr_elem.h: the launch-per-step arithmetic), W tile stored, flag;
//       3. VJP: every workgroup reads the row block's W (32 x n_mc), and for each of its tiles runs k_fr_vjp32's four chains (one wave per
//          quarter of the samples), adds them in wave order, applies vjp_elem + the optimiser step + ClipScale to the registers that hold the
//          tile, and refreshes the tile's LDS image for the next step's product.  The workgroup with the diagonal tile also owns mu.
//     Two hand-offs per step (store + acknowledge 0.65 us, flag 0.7 us, 32 KiB read 0.7 us each: tools/ubench_handoff.hip); every access to the
//     exchange areas is agent scope (sc1 stores write through, sc1 loads never hit a line the XCD's L2 kept from two steps ago), areas
//     double-buffered by step parity (a workgroup reaches step t + 2 only after every peer of its row block has finished reading step t).
//   * eps of ALL steps is drawn up front by one launch (k_eps_steps_fr: the same Philox stream as every other route, eps[i + m d] per step) and
//     streamed through two 32 KiB LDS slots per workgroup (one sub-stage = 32 rows x n_mc, 16-byte chunks XOR-swizzled as in k_fr_prod32: the
//     product reads it k-major, the VJP row-major -- one image serves both).
//   * every spin is bounded: a workgroup that never sees its peers (the row block's workgroups must be resident together: at most 256
//     workgroups of one per CU) sets status bit 8 and the kernel leaves; the host reports it.
// Not for the sticking-the-landing estimators, other targets, f64, sharded contexts, d > 1024 or n_mc outside {128, 256}: those keep the
// hipGraph of launches (MIVI_NO_FUSED_LOOP=1 forces it).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "device_common.h"
#include "fr_elem.h"
#include "fr_lds.h"
#include "optim_rules.h"

namespace mivi {

namespace {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4t __attribute__((ext_vector_type(4)));

// (kernels_fullrank_lds.hip split3_bf16 / mfma_bf16x3: the same instructions in the same order)
__device__ __forceinline__ void ft_split3(const float *x, bf16x8_t &hi, bf16x8_t &mid, bf16x8_t &lo) {
  u32x4t uh, um, ul;
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
  hi = __builtin_bit_cast(bf16x8_t, uh);
  mid = __builtin_bit_cast(bf16x8_t, um);
  lo = __builtin_bit_cast(bf16x8_t, ul);
}
__device__ __forceinline__ void ft_mfma_bf16x3(const float *av, const float *bv, f32x16 &acc) {
  bf16x8_t ah, am, al, bh, bm, bl;
  ft_split3(av, ah, am, al);
  ft_split3(bv, bh, bm, bl);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

#define FT_GLDS16(gptr, lptr, aux)                                                         \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr), \
                                   (__attribute__((address_space(3))) void *)(lptr), 16, 0, aux)

__device__ __forceinline__ void ft_store16_sc1(void *p, f32x4 v) {
  const u32x4t r = __builtin_bit_cast(u32x4t, v);
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ void ft_store4_sc1(float *p, float v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
// 16 bytes at agent scope (sc1 loads: never a line this XCD's L2 kept from two steps ago); four dword loads the compiler keeps its own
// wait counts for (an inline-asm dwordx4 load would leave its destination registers unguarded until a later, separate wait)
__device__ __forceinline__ f32x4 ft_load16_agent(const float *p) {
  f32x4 r;
  r.x = __hip_atomic_load(p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  r.y = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  r.z = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  r.w = __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return r;
}
__device__ __forceinline__ unsigned ft_poll(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ft_flag(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ft_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

}  // namespace

constexpr int kTilesMaxSub = 4;     // sub-stages (32 x 32 tiles of tril(C)) per workgroup: ceil(32 / 8) at d = 1024
constexpr int kTilesNT = 256;

struct FrTilesArgs {
  int d, M, ncb, nrb, n_steps, rule, ent_kind, M_total, spin;
  float *params, *opt_state;
  const float *t_mean, *t_istd;
  const float *eps_all;       // [n_steps][M][d]
  const int4 *work;           // per workgroup: .x = rb | g << 16, .y = first sub-stage | end sub-stage << 16, .z = runs of the row block, .w unused
  float *P;                   // [2][nrb][8][ncb][1024] partial tiles (column n: 32 rows contiguous)
  float *Wx;                  // [2][nrb][M][32] W of a row block (sample m: 32 rows contiguous)
  float *MU;                  // [2][nrb][32]
  unsigned *flagP, *flagW;    // [nrb][8] steps whose partial tiles run g has delivered; [nrb][8] steps whose W tile of sample block cb is in place
  double *hist_ell;           // [n_steps][nrb * 8] ell partials per (row block, sample block)
  double *hist_ld;            // [n_steps][2][nrb] sum log C_ii, count of non-positive C_ii per row block
  long long t0;
  double eta, clip_eps, b1, b2, adam_eps;
  int *status;
  long long *dbg;             // developer builds (make DEV=1, MIVI_TILES_DBG=1): [workgroup][step][10] stamps of the 100 MHz wall clock
};

#ifdef MIVI_DEV
#define FT_STAMP(k) do { if (a.dbg && tid == 0) a.dbg[((size_t)blockIdx.x * a.n_steps + t) * 10 + (k)] = (long long)wall_clock64(); } while (0)
#else
#define FT_STAMP(k) do { } while (0)
#endif

// eps of n_steps estimates, out[(t M + m) d + i]: one Philox block (rows 4 q .. 4 q + 3 of column m of estimate idx0 + t) per thread; the
// block's sum of 0.5 eps^2 -> he[t][blockIdx.x]
__global__ __launch_bounds__(256) void k_eps_steps_fr(uint64_t seed, uint64_t idx0, int d, int M, int m_offset, float *out, double *he) {
  __shared__ double red[4];
  const int d4 = d >> 2, t = blockIdx.y;
  const long long n = (long long)d4 * M, i = (long long)blockIdx.x * 256 + threadIdx.x;
  float hv = 0.f;
  if (i < n) {
    const int m = (int)(i / d4), q = (int)(i - (long long)m * d4);
    float e[4];
    eps_block<float>(seed, idx0 + (uint64_t)t, (uint64_t)(m_offset + m) * (uint64_t)d4 + (uint64_t)q, e);
    const f32x4 ev = {e[0], e[1], e[2], e[3]};
    *(f32x4 *)(out + ((size_t)t * M + m) * d + 4 * q) = ev;
    hv = 0.5f * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3]);
  }
  const double s = block_sum<double, 256>((double)hv, red);
  if (threadIdx.x == 0) he[(size_t)t * gridDim.x + blockIdx.x] = s;
}

// partial tiles of sample blocks cb = g, g + NR, ... (CNT of them at most) from all NR runs: every load in flight before the first add
template <int NR>
__device__ __forceinline__ void ft_reduce_load(const float *Pb, int ncb, int g, int off, f32x4 (&v)[8]) {
  constexpr int CNT = (8 + NR - 1) / NR;
  f32x4 tmp[CNT][NR];
#pragma unroll
  for (int ci = 0; ci < CNT; ++ci) {
    const int cb = g + ci * NR;
#pragma unroll
    for (int gg = 0; gg < NR; ++gg)
      tmp[ci][gg] = cb < ncb ? ft_load16_agent(Pb + ((size_t)gg * ncb + cb) * 1024 + off) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int ci = 0; ci < CNT; ++ci) {
    f32x4 s = tmp[ci][0];
#pragma unroll
    for (int gg = 1; gg < NR; ++gg) s += tmp[ci][gg];   // (run order: k_fr_prod32's wave order)
#pragma unroll
    for (int gg = NR; gg < 8; ++gg) s += 0.f;            // (its idle waves' zero tiles)
    v[ci] = s;
  }
}

template <int RULE, int CBW>   // CBW: sample blocks per wave in the product (n_mc = 128 CBW)
__global__ __launch_bounds__(kTilesNT) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_fr_tiles_loop(FrTilesArgs a) {
  constexpr int NT = kTilesNT, LDC = 36, ESLOT = 32 * 32 * 4 * CBW;   // floats per eps slot: n_mc x 32
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *E = lds;                               // [2][n_mc][32] eps sub-stages, 16-byte chunks swizzled by (m >> 1) & 7
  float *Wimg = E + 2 * ESLOT;                  // [n_mc][32] W of the row block; the VJP tiles' partial images Cs[4][32][LDC] afterwards
  constexpr int WSZ = ESLOT > 4 * 32 * LDC ? ESLOT : 4 * 32 * LDC;
  float *pimg = Wimg + WSZ;                     // [kTilesMaxSub][32 k][LDC]: this workgroup's tiles of tril(C), column k: 32 rows
  float *rs_lds = pimg + kTilesMaxSub * 32 * LDC;   // [8][32] partial row sums of W
  float(*cc_tab)[2] = reinterpret_cast<float(*)[2]>(rs_lds + 256);   // [256][2] Adam bias corrections
  double *red = reinterpret_cast<double *>(rs_lds + 256 + 512);      // [4]
  int *s_ok = reinterpret_cast<int *>(red + 4);
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int d = a.d, M = a.M, ncb = a.ncb, nrb = a.nrb;
  const __attribute__((address_space(4))) int *wp = (const __attribute__((address_space(4))) int *)a.work + 4 * blockIdx.x;
  const int rb = wp[0] & 0xffff, g = wp[0] >> 16, s0 = wp[1] & 0xffff, s1 = wp[1] >> 16, nruns = wp[2];
  const int nsub = s1 - s0, row0 = rb * 32;
  const bool has_diag = s1 == rb + 1;          // this run ends with the diagonal tile: the workgroup owns mu of the row block
  const int i4 = 4 * (tid & 7), en = tid >> 3;  // epilogue / reduce element of this thread: rows i4 .. i4 + 3 of column en of a 32 x 32 tile
  const size_t plen = (size_t)d + (size_t)d * d;
  const double invM = 1.0 / (double)a.M_total;
  const bool pow2M = (a.M_total & (a.M_total - 1)) == 0;
  const float invMf = (float)invM;
  const double direct = direct_entropy_coeff(a.ent_kind);
  const bool do_clip = a.clip_eps == a.clip_eps;   // NaN = no ClipScale

  // ---- this workgroup's tiles: parameters and moments -> registers, parameters -> LDS image ---------------------------------------------
  f32x4 px[kTilesMaxSub], pm[kTilesMaxSub], pv[kTilesMaxSub];
#pragma unroll
  for (int u = 0; u < kTilesMaxSub; ++u) {
    px[u] = pm[u] = pv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (u < nsub) {
      const size_t pi = (size_t)d + (size_t)((s0 + u) * 32 + en) * d + row0 + i4;
      px[u] = *(const f32x4 *)(a.params + pi);
      if (RULE == 1) { pm[u] = *(const f32x4 *)(a.opt_state + pi); pv[u] = *(const f32x4 *)(a.opt_state + plen + pi); }
      *(f32x4 *)(pimg + (u * 32 + en) * LDC + i4) = px[u];
    }
  }
  float mu_x = 0.f, mu_m = 0.f, mu_v = 0.f;    // mu of row row0 + tid (threads 0 .. 31 of the workgroup with the diagonal tile)
  if (has_diag && tid < 32) {
    mu_x = a.params[row0 + tid];
    if (RULE == 1) { mu_m = a.opt_state[row0 + tid]; mu_v = a.opt_state[plen + row0 + tid]; }
  }
  const f32x4 tm4 = *(const f32x4 *)(a.t_mean + row0 + i4), tis4 = *(const f32x4 *)(a.t_istd + row0 + i4);

  // eps sub-stage (step t, sub-stage s of this workgroup) -> slot: n_mc / 8 pieces of 1 KiB (8 samples x 8 chunks), wave w pieces w, w + 4, ...
  auto eps_in = [&](int t, int s, int slot) {
    const float *G = a.eps_all + (size_t)t * M * d + (size_t)(s0 + s) * 32;
    float *dst = E + slot * ESLOT;
    const int c = lane & 7;
#pragma unroll
    for (int p = 0; p < 4 * CBW; ++p) {
      const int m = 8 * (w + 4 * p) + (lane >> 3);
      FT_GLDS16(G + (size_t)m * d + 4 * (c ^ ((m >> 1) & 7)), dst + (w + 4 * p) * 256, 0);
    }
  };
  auto lost = [&]() {
    if (tid == 0) atomicOr(a.status, 8);
  };
  // thread k < count polls flags[k] until it has reached `want` (bounded); uniform result
  auto wait_flags = [&](const unsigned *flags, int count, unsigned want) -> bool {
    if (tid == 0) *s_ok = 1;
    lds_barrier();
    if (tid < count) {
      int budget = a.spin;
      while ((int)(ft_poll(flags + tid) - want) < 0) {
        if (--budget <= 0) { atomicAnd(s_ok, 0); break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    lds_barrier();
    return *s_ok != 0;
  };

  eps_in(0, 0, 0);
  if (nsub > 1) eps_in(0, 1, 1);

  for (int t = 0; t < a.n_steps; ++t) {
    const int par = t & 1;
    if (RULE == 1 && (t & 255) == 0) adam_bias<float>(a.t0 + t + tid + 1, a.b1, a.b2, cc_tab[tid][0], cc_tab[tid][1]);
    // ---- 1. product: this run's chain for every sample block ---------------------------------------------------------------------------
    f32x16 acc[CBW];
#pragma unroll
    for (int j = 0; j < CBW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    ft_vm0();
    lds_barrier();   // (eps sub-stages 0 and 1 of this step, the refreshed parameter images and the bias table are in place)
    FT_STAMP(0);
#pragma unroll 1
    for (int u = 0; u < nsub; ++u) {
      if (u == 2) {   // sub-stages 2 and 3 take the slots of 0 and 1
        lds_barrier();
        eps_in(t, 2, 0);
        if (nsub > 3) eps_in(t, 3, 1);
        ft_vm0();
        lds_barrier();
      }
      const float *Es = E + (u & 1) * ESLOT, *Ps = pimg + u * 32 * LDC;
      float av[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) av[i] = Ps[(8 * (i >> 2) + 4 * h + (i & 3)) * LDC + l31];
      if (s0 + u == rb) {   // the diagonal block of tril(C): keep k <= i
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (8 * (i >> 2) + 4 * h + (i & 3) > l31) av[i] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < CBW; ++j) {
        const int m = (w + 4 * j) * 32 + l31;
        const int swz = h ^ ((m >> 1) & 7);
        float bv[16];
#pragma unroll
        for (int s8 = 0; s8 < 4; ++s8) {
          const f32x4 q4 = *(const f32x4 *)(Es + m * 32 + 4 * ((2 * s8) ^ swz));
          bv[4 * s8 + 0] = q4.x; bv[4 * s8 + 1] = q4.y; bv[4 * s8 + 2] = q4.z; bv[4 * s8 + 3] = q4.w;
        }
        ft_mfma_bf16x3(av, bv, acc[j]);
        ft_mfma_bf16x3(av + 8, bv + 8, acc[j]);
      }
    }
    // partial tiles -> the row block's exchange area (column l31: rows 8 q + 4 h .. + 3), mu of the step, flag
    FT_STAMP(1);
    {
      float *Pt = a.P + ((((size_t)par * nrb + rb) * 8 + g) * ncb) * 1024;
#pragma unroll
      for (int j = 0; j < CBW; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]};
          ft_store16_sc1(Pt + (size_t)(w + 4 * j) * 1024 + l31 * 32 + 8 * q + 4 * h, v);
        }
      if (has_diag && tid < 32) ft_store4_sc1(a.MU + ((size_t)par * nrb + rb) * 32 + tid, mu_x);
      if (has_diag && tid < 32) {   // log|det C| partial of the row block, from the diagonal tile's registers' image
        const float cii = pimg[((nsub - 1) * 32 + tid) * LDC + tid];
        float lg, bad;
        logdet_block32(cii, lg, bad);
        if (tid == 0) { a.hist_ld[((size_t)t * 2 + 0) * nrb + rb] = (double)lg; a.hist_ld[((size_t)t * 2 + 1) * nrb + rb] = (double)bad; }
      }
    }
    ft_vm0();
    lds_barrier();   // (every thread's stores are acknowledged; every wave is done with the eps slots of the product)
    if (tid == 0) ft_flag(a.flagP + rb * 8 + g, (unsigned)(t + 1));
    FT_STAMP(2);
    if (nsub > 2) {   // the VJP starts over at sub-stage 0: on its way under the exchange
      eps_in(t, 0, 0);
      eps_in(t, 1, 1);
    }
    // ---- 2. reduce: sample blocks cb = g, g + nruns, ... of this row block ---------------------------------------------------------------
    if (!wait_flags(a.flagP + rb * 8, nruns, (unsigned)(t + 1))) { lost(); return; }
    FT_STAMP(3);
    {
      const float *Pb = a.P + (((size_t)par * nrb + rb) * 8) * ncb * 1024;
      const int off = en * 32 + i4;
      f32x4 v[8];
      const f32x4 mu4 = ft_load16_agent(a.MU + ((size_t)par * nrb + rb) * 32 + i4);
      switch (nruns) {
        case 1: ft_reduce_load<1>(Pb, ncb, g, off, v); break;
        case 2: ft_reduce_load<2>(Pb, ncb, g, off, v); break;
        case 3: ft_reduce_load<3>(Pb, ncb, g, off, v); break;
        case 4: ft_reduce_load<4>(Pb, ncb, g, off, v); break;
        case 5: ft_reduce_load<5>(Pb, ncb, g, off, v); break;
        case 6: ft_reduce_load<6>(Pb, ncb, g, off, v); break;
        case 7: ft_reduce_load<7>(Pb, ncb, g, off, v); break;
        default: ft_reduce_load<8>(Pb, ncb, g, off, v); break;
      }
      float *Wb = a.Wx + ((size_t)par * nrb + rb) * M * 32;
#pragma unroll
      for (int ci = 0; ci < 8; ++ci) {
        const int cb = g + ci * nruns;
        if (cb >= ncb) break;
        const f32x4 z = mu4 + v[ci];
        float ell = 0.f;
        f32x4 wv;
#pragma unroll
        for (int c = 0; c < 4; ++c) wv[c] = diag_target_elem(z[c], tm4[c], tis4[c], ell);
        ft_store16_sc1(Wb + (size_t)(cb * 32 + en) * 32 + i4, wv);
        const double sl = block_sum_nodrain_f32<NT>(ell, red);
        if (tid == 0) a.hist_ell[(size_t)t * nrb * 8 + rb * 8 + cb] = sl;
        lds_barrier();   // (`red` is reused)
      }
      ft_vm0();
      lds_barrier();
      if (tid == 0)
        for (int cb = g; cb < ncb; cb += nruns) ft_flag(a.flagW + rb * 8 + cb, (unsigned)(t + 1));
    }
    FT_STAMP(4);
    // ---- 3. VJP of this workgroup's tiles + update ------------------------------------------------------------------------------------------
    if (!wait_flags(a.flagW + rb * 8, ncb, (unsigned)(t + 1))) { lost(); return; }
    FT_STAMP(5);
    {
      const float *Wb = a.Wx + ((size_t)par * nrb + rb) * M * 32;
#pragma unroll
      for (int p = 0; p < 4 * CBW; ++p) FT_GLDS16(Wb + (size_t)(w + 4 * p) * 256 + 4 * lane, Wimg + (w + 4 * p) * 256, 16);   // (sc1)
    }
    ft_vm0();
    lds_barrier();   // (W of the row block; eps sub-stages 0 and 1 again where they had been replaced)
    FT_STAMP(6);
    constexpr int NSM = CBW;             // 32-sample sub-stages of a wave's quarter of the samples: (n_mc / 4) / 32
    float avW[NSM][16];
    float rsum = 0.f;
#pragma unroll
    for (int tq = 0; tq < NSM; ++tq)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = w * (32 * NSM) + 32 * tq + 8 * (i >> 2) + 4 * h + (i & 3);
        avW[tq][i] = Wimg[m * 32 + l31];
      }
    if (has_diag) {
#pragma unroll
      for (int tq = 0; tq < NSM; ++tq)
#pragma unroll
        for (int i = 0; i < 16; ++i) rsum += avW[tq][i];
    }
    lds_barrier();   // (every wave has its W fragments: the W image becomes the tiles' partial images)
    float *Cs = Wimg;
    if (has_diag) rs_lds[(2 * w + h) * 32 + l31] = rsum;
    const float c1 = RULE == 1 ? cc_tab[t & 255][0] : 0.f, c2 = RULE == 1 ? cc_tab[t & 255][1] : 0.f;
#pragma unroll
    for (int u = 0; u < kTilesMaxSub; ++u) {
      if (u >= nsub) break;
      if (u == 2) {   // sub-stages 2 and 3 take the slots of 0 and 1 (every wave is past its reads of them: the barrier behind tile 1's image)
        eps_in(t, 2, 0);
        if (nsub > 3) eps_in(t, 3, 1);
        ft_vm0();
        lds_barrier();
      }
      const float *Es = E + (u & 1) * ESLOT;
      f32x16 ac;
#pragma unroll
      for (int r = 0; r < 16; ++r) ac[r] = 0.f;
#pragma unroll
      for (int tq = 0; tq < NSM; ++tq) {
        float bv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int m = w * (32 * NSM) + 32 * tq + 8 * (i >> 2) + 4 * h + (i & 3);
          bv[i] = Es[m * 32 + 4 * ((l31 >> 2) ^ ((m >> 1) & 7)) + (l31 & 3)];
        }
        ft_mfma_bf16x3(avW[tq], bv, ac);
        ft_mfma_bf16x3(avW[tq] + 8, bv + 8, ac);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = {ac[4 * q], ac[4 * q + 1], ac[4 * q + 2], ac[4 * q + 3]};
        *(f32x4 *)(Cs + (w * 32 + l31) * LDC + 8 * q + 4 * h) = v;
      }
      lds_barrier();
      {   // vjp_epilogue's fused-update branch on the registers that hold the tile (fr_lds.h)
        const int gi = row0 + i4, gj = (s0 + u) * 32 + en;
        f32x4 v = *(const f32x4 *)(Cs + en * LDC + i4);
#pragma unroll
        for (int k2 = 1; k2 < 4; ++k2) v += *(const f32x4 *)(Cs + (k2 * 32 + en) * LDC + i4);
        float cjj = 1.f;
        const bool diag_tile = s0 + u == rb;
        if (diag_tile && gj >= gi && gj < gi + 4) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (gj == gi + c) cjj = px[u][c];
        }
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = vjp_elem(v[c], gi + c, gj, pow2M, invMf, invM, direct, cjj);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (gj > gi + c) continue;   // zero gradients above the diagonal move nothing
          float x;
          if (RULE == 0) {
            x = descent_step(px[u][c], o[c], (float)a.eta);
          } else {
            float m = pm[u][c], vv = pv[u][c];
            x = adam_step<float>(px[u][c], o[c], m, vv, c1, c2, (float)a.eta, (float)a.b1, (float)a.b2, (float)a.adam_eps);
            pm[u][c] = m;
            pv[u][c] = vv;
          }
          if (gj == gi + c && do_clip) x = clip_step(x, (float)a.clip_eps);
          px[u][c] = x;
        }
        *(f32x4 *)(pimg + (u * 32 + en) * LDC + i4) = px[u];
      }
      lds_barrier();   // (the partial images are free again; rs_lds is in place behind the first of these)
    }
    FT_STAMP(7);
    if (has_diag && tid < 32) {   // d/dmu of the row block and its update (vjp_epilogue's mu_tile branch)
      double sm = 0.0;
#pragma unroll
      for (int gq = 0; gq < 8; ++gq) sm += (double)rs_lds[gq * 32 + tid];
      const float gm = dmu_elem(sm, invM);
      if (RULE == 0) mu_x = descent_step(mu_x, gm, (float)a.eta);
      else mu_x = adam_step<float>(mu_x, gm, mu_m, mu_v, c1, c2, (float)a.eta, (float)a.b1, (float)a.b2, (float)a.adam_eps);
    }
    // eps of the next step on its way (every wave is past this step's last read of the slots: the barrier above)
    if (t + 1 < a.n_steps) {
      eps_in(t + 1, 0, 0);
      if (nsub > 1) eps_in(t + 1, 1, 1);
    }
  }
  // ---- the tiles go back to memory ---------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int u = 0; u < kTilesMaxSub; ++u) {
    if (u >= nsub) break;
    const size_t pi = (size_t)d + (size_t)((s0 + u) * 32 + en) * d + row0 + i4;
    const bool diag_tile = s0 + u == rb;
    if (!diag_tile) {
      *(f32x4 *)(a.params + pi) = px[u];
      if (RULE == 1) { *(f32x4 *)(a.opt_state + pi) = pm[u]; *(f32x4 *)(a.opt_state + plen + pi) = pv[u]; }
    } else {   // nothing above the diagonal is touched
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if ((s0 + u) * 32 + en <= row0 + i4 + c) {
          a.params[pi + c] = px[u][c];
          if (RULE == 1) { a.opt_state[pi + c] = pm[u][c]; a.opt_state[plen + pi + c] = pv[u][c]; }
        }
    }
  }
  if (has_diag && tid < 32) {
    a.params[row0 + tid] = mu_x;
    if (RULE == 1) { a.opt_state[row0 + tid] = mu_m; a.opt_state[plen + row0 + tid] = mu_v; }
  }
}

// elbo[t] (and the status word) from the per-step partials; one workgroup per step
__global__ __launch_bounds__(256) void k_fr_tiles_value(int d, int n_ell, int n_he, int nrb, int M_local, int M_total, int ent_kind, double ell_const,
                                                        const double *hist_ell, const double *he, const double *hist_ld, double *elbo, float *value_last,
                                                        int n_steps, int *status) {
  __shared__ double red[4 * 4];
  const int t = blockIdx.x, tid = threadIdx.x;
  double v[4] = {0, 0, 0, 0};
  for (int i = tid; i < n_ell; i += 256) v[0] += hist_ell[(size_t)t * n_ell + i];
  for (int i = tid; i < n_he; i += 256) v[1] += he[(size_t)t * n_he + i];
  for (int i = tid; i < nrb; i += 256) {
    v[2] += hist_ld[((size_t)t * 2 + 0) * nrb + i];
    v[3] += hist_ld[((size_t)t * 2 + 1) * nrb + i];
  }
  block_sum_n<double, 256, 4>(v, red);
  if (tid == 0) {
    const double Mt = (double)M_total;
    const double ent = (ent_is_closed(ent_kind) ? 0.5 * d * (1.0 + kLog2Pi) : v[1] / Mt + 0.5 * d * kLog2Pi) + v[2];
    const double value = -((v[0] + (double)M_local * ell_const) / Mt + ent);
    elbo[t] = -value;
    if (t == n_steps - 1 && value_last) *value_last = (float)value;
    int st = 0;
    if (!isfinite(value)) st |= 1;
    if (v[3] > 0.0) st |= 2;
    if (st && status) atomicOr(status, st);
  }
}

bool fr_tiles_loop_ok(const mivi_ctx *c) {
  // OPT-IN (MIVI_TILES_LOOP=1, read per call): built, bitwise the launch-per-step trajectory, and measured SLOWER than the hipGraph of launches
  // (25 against 15.3 us per step at d = 1024, n_mc = 256: the two hand-offs inside a row block cost 7.7 us, and one wave per SIMD leaves the
  // operand split beside no other wave's MFMAs -- DESIGN.md 9)
  const bool off = getenv("MIVI_TILES_LOOP") == nullptr;
  const int d = c->cfg.d, M = c->cfg.n_mc;
  const bool stl = c->cfg.entropy == MIVI_ENT_STL || c->cfg.entropy == MIVI_ENT_STL_ZERO_GRAD;
  return !off && c->cfg.family == MIVI_FULLRANK && c->cfg.dtype == MIVI_F32 && c->target == TGT_DIAG_GAUSS && !c->bij_on && !stl && d % 64 == 0 && d >= 64 &&
         d <= 1024 && (M == 128 || M == 256) && c->cfg.m_offset == 0 && c->M_total == M;
}

// bytes of the loop's buffers behind each other: eps of all steps, the half-squared-norm partials, exchange areas, flags, work table, value partials
struct FrTilesLayout {
  size_t eps, he, P, Wx, MU, flags, work, hist_ell, hist_ld, total;
  int n_he, n_wg;
};
static FrTilesLayout fr_tiles_layout(const mivi_ctx *c, int n_steps) {
  const int d = c->cfg.d, M = c->cfg.n_mc, nrb = d / 32, ncb = M / 32;
  FrTilesLayout L;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  L.n_he = (int)(((long long)(d / 4) * M + 255) / 256);
  int n_wg = 0;
  for (int rb = 0; rb < nrb; ++rb) {
    const int rc = (rb + 1 + 7) / 8;
    n_wg += (rb + 1 + rc - 1) / rc;
  }
  L.n_wg = n_wg;
  size_t o = 0;
  L.eps = o; o += al((size_t)n_steps * M * d * sizeof(float));
  L.he = o; o += al((size_t)n_steps * L.n_he * sizeof(double));
  L.P = o; o += al((size_t)2 * nrb * 8 * ncb * 1024 * sizeof(float));
  L.Wx = o; o += al((size_t)2 * nrb * M * 32 * sizeof(float));
  L.MU = o; o += al((size_t)2 * nrb * 32 * sizeof(float));
  L.flags = o; o += al((size_t)2 * nrb * 8 * sizeof(unsigned));
  L.work = o; o += al((size_t)n_wg * sizeof(int4));
  L.hist_ell = o; o += al((size_t)n_steps * nrb * 8 * sizeof(double));
  L.hist_ld = o; o += al((size_t)n_steps * 2 * nrb * sizeof(double));
  L.total = o;
  return L;
}
size_t fr_tiles_bytes(const mivi_ctx *c, int n_steps) { return fr_tiles_layout(c, n_steps).total; }

// buf: fr_tiles_bytes; elbo: n_steps doubles; value: one float (the last step's objective value)
void launch_fr_tiles_loop(mivi_ctx *c, void *params, void *opt_state, uint64_t idx0, long long t0, int n_steps, int rule, double eta, double clip_eps,
                          char *buf, double *elbo, void *value) {
  const int d = c->cfg.d, M = c->cfg.n_mc, nrb = d / 32, ncb = M / 32;
  const FrTilesLayout L = fr_tiles_layout(c, n_steps);
  // work table: one workgroup per run of every row block; the row blocks' groups dealt heaviest first onto the lightest XCD (workgroup b runs
  // on XCD b % 8: a row block's exchange then stays in one L2's neighbourhood -- a speed hint only, every exchange access is agent scope)
  std::vector<std::vector<int4>> lists(8);
  {
    std::vector<int> order(nrb);
    for (int i = 0; i < nrb; ++i) order[i] = nrb - 1 - i;
    int load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int rb : order) {
      const int rc = (rb + 1 + 7) / 8, nruns = (rb + 1 + rc - 1) / rc;
      int best = 0;
      for (int x = 1; x < 8; ++x)
        if (load[x] < load[best]) best = x;
      for (int g = 0; g < nruns; ++g) {
        const int sb = g * rc, se = std::min((g + 1) * rc, rb + 1);
        lists[best].push_back(make_int4(rb | (g << 16), sb | (se << 16), nruns, 0));
      }
      load[best] += nruns;
    }
  }
  std::vector<int4> work;
  {
    size_t Lmax = 0;
    for (auto &l : lists) Lmax = std::max(Lmax, l.size());
    // (every slot must hold a real workgroup: lists of unequal length are interleaved as far as they go, the rest appended)
    for (size_t i = 0; i < Lmax; ++i)
      for (int x = 0; x < 8; ++x)
        if (i < lists[x].size()) work.push_back(lists[x][i]);
  }
  (void)hipMemcpyAsync(buf + L.work, work.data(), work.size() * sizeof(int4), hipMemcpyHostToDevice, c->stream);
  (void)hipStreamSynchronize(c->stream);   // (the table is a host temporary)
  (void)hipMemsetAsync(buf + L.flags, 0, (size_t)2 * nrb * 8 * sizeof(unsigned), c->stream);
  hipLaunchKernelGGL(k_eps_steps_fr, dim3(L.n_he, n_steps), dim3(256), 0, c->stream, c->cfg.seed, idx0, d, M, c->cfg.m_offset, (float *)(buf + L.eps),
                     (double *)(buf + L.he));
  FrTilesArgs a;
  a.d = d; a.M = M; a.ncb = ncb; a.nrb = nrb; a.n_steps = n_steps; a.rule = rule; a.ent_kind = c->cfg.entropy; a.M_total = c->M_total;
  a.spin = 1 << 20;
  a.params = (float *)params; a.opt_state = (float *)opt_state;
  a.t_mean = (const float *)c->t_mean.p; a.t_istd = (const float *)c->t_istd.p;
  a.eps_all = (const float *)(buf + L.eps);
  a.work = (const int4 *)(buf + L.work);
  a.P = (float *)(buf + L.P); a.Wx = (float *)(buf + L.Wx); a.MU = (float *)(buf + L.MU);
  a.flagP = (unsigned *)(buf + L.flags); a.flagW = a.flagP + (size_t)nrb * 8;
  a.hist_ell = (double *)(buf + L.hist_ell); a.hist_ld = (double *)(buf + L.hist_ld);
  a.t0 = t0; a.eta = eta; a.clip_eps = clip_eps; a.b1 = 0.9; a.b2 = 0.999; a.adam_eps = 1e-8;
  a.status = (int *)c->status.p;
  a.dbg = nullptr;
#ifdef MIVI_DEV
  static long long *dbg_dev = nullptr;
  static size_t dbg_n = 0;
  if (getenv("MIVI_TILES_DBG")) {
    const size_t need = work.size() * (size_t)n_steps * 10;
    if (dbg_n < need) { if (dbg_dev) (void)hipFree(dbg_dev); (void)hipMalloc(&dbg_dev, need * sizeof(long long)); dbg_n = need; }
    (void)hipMemsetAsync(dbg_dev, 0, need * sizeof(long long), c->stream);
    a.dbg = dbg_dev;
  }
#endif
  (void)hipMemsetAsync(buf + L.hist_ell, 0, (size_t)n_steps * nrb * 8 * sizeof(double), c->stream);
  const int CBW = M / 128;
  const size_t eslot = (size_t)32 * 32 * 4 * CBW;
  const size_t wsz = eslot > 4 * 32 * 36 ? eslot : 4 * 32 * 36;
  const size_t lds = (2 * eslot + wsz + kTilesMaxSub * 32 * 36 + 256 + 512 + 8 + 8) * sizeof(float);
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)work.size()), dim3(kTilesNT), lds, c->stream, a);
  };
  if (rule == 0) { if (CBW == 1) go(k_fr_tiles_loop<0, 1>); else go(k_fr_tiles_loop<0, 2>); }
  else { if (CBW == 1) go(k_fr_tiles_loop<1, 1>); else go(k_fr_tiles_loop<1, 2>); }
#ifdef MIVI_DEV
  if (a.dbg) {   // per phase and class of workgroup (sub-stages owned): mean over the workgroups and the steps of the second half of the call
    (void)hipStreamSynchronize(c->stream);
    std::vector<long long> hst(work.size() * (size_t)n_steps * 10);
    (void)hipMemcpy(hst.data(), a.dbg, hst.size() * sizeof(long long), hipMemcpyDeviceToHost);
    static const char *nm[9] = {"product", "store+ack", "wait P", "reduce", "wait W", "W load", "vjp", "mu/tail", "next eps wait"};
    for (int cls = 1; cls <= 4; ++cls) {
      double acc[9] = {0}, cnt = 0;
      for (size_t b = 0; b < work.size(); ++b) {
        if ((work[b].y >> 16) - (work[b].y & 0xffff) != cls) continue;
        for (int t = n_steps / 2; t + 1 < n_steps; ++t) {
          const long long *p = &hst[(b * n_steps + t) * 10], *nx = &hst[(b * n_steps + t + 1) * 10];
          for (int k = 0; k < 7; ++k) acc[k] += (double)(p[k + 1] - p[k]);
          acc[7] += 0.0;
          acc[8] += (double)(nx[0] - p[7]);
          cnt += 1;
        }
      }
      if (cnt == 0) continue;
      fprintf(stderr, "[tiles dbg] workgroups with %d sub-stages:", cls);
      for (int k = 0; k < 9; ++k) if (k != 7) fprintf(stderr, " %s %.2f us |", nm[k], acc[k] / cnt * 0.01);
      fprintf(stderr, "\n");
    }
  }
#endif
  hipLaunchKernelGGL(k_fr_tiles_value, dim3(n_steps), dim3(256), 0, c->stream, d, nrb * 8, L.n_he, nrb, M, c->M_total, c->cfg.entropy, c->t_const,
                     (const double *)(buf + L.hist_ell), (const double *)(buf + L.he), (const double *)(buf + L.hist_ld), elbo, (float *)value, n_steps,
                     (int *)c->status.p);
}

}  // namespace mivi
