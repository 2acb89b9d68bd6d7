// Mean-field (Diagonal scale) RepGradELBO kernels for gfx950.
//
// Reference semantics (AdvancedVI.jl v0.7.0):
//   sampling   z = diag .* eps .+ mu                      src/families/location_scale.jl:80-87
//   entropy    five estimators                            src/algorithms/entropy.jl:13-90
//   objective  -(mean_m logpi(z_m) + entropy)             src/algorithms/repgradelbo.jl:142-149
//   gradient   what AD of that forward yields (closed form):
//              d/dmu = -(1/M) sum_m W_m,  d/dsigma = -(1/M) sum_m W_m .* eps_m - direct/sigma,
//              W = grad logpi(z) (+ eps/sigma for the sticking-the-landing estimators)
//
// Elementwise work + row reductions: eps is generated in registers (one Philox block = rows 4b..4b+3 of one column),
// lanes run along the sample axis so every row sum is a wave64 reduction, and the whole estimate is ONE launch:
// workgroup (b, 0) owns rows 4b..4b+3 for all columns and writes their gradient entries directly; per-workgroup scalar
// partials are assembled into the objective value by one extra workgroup of the NEXT estimate's launch (graph-chained
// mode) or by k_value_only (single calls) in a fixed summation order => bitwise reproducible.
#include <cstdio>
#include <cstdlib>
#include "device_common.h"
#include "optim_rules.h"

namespace mivi {

// DPP reduction to the 16-lane row level: afterwards every lane holds the sum of its row of 16 lanes.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}
__device__ __forceinline__ double row16_sum(double v) {   // f64 contexts: plain shuffles within the row
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Wave total, valid in LANE 63 (f32: the row sums above, then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2-3 --
// six DPP adds, nothing through LDS; f64: shuffles, valid everywhere).  One fixed tree, shared by the one-launch kernel and
// the device loop, so both produce the same bits.
__device__ __forceinline__ float wave_total63(float v) {
  v = row16_sum(v);
  // (the two cross-row steps as ONE DPP add each: from update_dpp + add the compiler makes v_mov_dpp, v_mov, v_add for a partial row mask --
  //  forty vector instructions per estimate of the launch-free loops for nothing; same values, same order; s_nop 1: a VALU result read by
  //  a DPP operand needs two wait states)
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v));
  return v;
}
__device__ __forceinline__ double wave_total63(double v) { return wave_sum(v); }

// One sample column of the fused funnel target for rows 4 rq .. 4 rq + 3 (Neal's funnel + Stacked([log, identity]), see FunnelFin):
// rows >= 1 need only e1 = z[0, m], re-derived from the eps stream; their sum of squares enters row 0 and ell only through
// x^2 exp(-2 e1) (and that times eps_0).  ONE body for k_mf_main<T, true> and k_mf_funnel_loop: the two must produce the same bits.
template <typename T>
__device__ __forceinline__ void funnel_column(int rq, int d, const T (&mu)[4], const T (&sg)[4], const T (&e)[4], T e0, T mu0, T sg0, T (&g)[4],
                                              T &s_ell, T &sA, T &sB) {
  // Floating-point contraction off, the fused multiply-adds explicit: this body is instantiated in three kernels (k_mf_main<T, true>,
  // k_mf_funnel_loop, k_mf_funnel_sgd_loop) that must agree to the bit, and what the optimiser fuses depends on the code around it (the A / B
  // partials of the third came out one ulp apart from the first's once in a few steps).
#pragma clang fp contract(off)
  const T e1 = fma(sg0, e0, mu0);   // (funnel_finish re-derives exactly this value)
  const T inv_s2 = exp(T(-2) * e1);
  T x2 = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * rq + r;
    const T z = fma(sg[r], e[r], mu[r]);
    if (i >= 1 && i < d) {
      g[r] = -z * inv_s2;
      x2 = fma(z, z, x2);
    }
  }
  const T xi = x2 * inv_s2;
  s_ell = fma(T(-0.5), xi, s_ell);
  sA += xi;
  sB = fma(xi, e0, sB);
}
// The row sums of one column: W = g (+ eps / sigma for the sticking-the-landing estimators), sum W, sum W eps, sum 0.5 eps^2.
template <typename T>
__device__ __forceinline__ void mf_accumulate(int rq, int d, bool want_grad, bool stl, bool skip_row0, const T (&sg)[4], const T (&e)[4], const T (&g)[4],
                                              T (&sW)[4], T (&sWe)[4], T &s_he) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool ok = (4 * rq + r) < d;
    const T er = ok ? e[r] : T(0);
    s_he += T(0.5) * er * er;
    if (want_grad) {
      const T isg = T(1) / sg[r];
      const bool mine = ok && !(skip_row0 && 4 * rq + r == 0);   // funnel row 0: value workgroup
      const T w = mine ? (g[r] + (stl ? er * isg : T(0))) : T(0);
      sW[r] += w;
      sWe[r] += w * er;
    }
  }
}

// Scalar partial layout written by k_mf_main and consumed by k_mf_value: sc[k*nblk + blk],
// k = 0 sum ell (variable part), 1 sum 0.5 eps^2, 2 sum log sigma_i (this block's rows), 3 #non-positive sigma.
// FN: the fused funnel target is a separate instantiation (its extra Philox block / exp / partial store would otherwise
// sit in the register budget of the plain kernel).
template <typename T, bool FN = false>
__global__ __launch_bounds__(256) void k_mf_main(MfArgs<T> a) {
  constexpr int NV = FN ? 12 : 10;   // reduced values: sW[4], sWe[4], ell, 0.5 eps^2 (+ the funnel's A, B)
  __shared__ T xw[NV][4];         // [value][wave] partial sums
  __shared__ double tot[14];      // 0-7 rows, 8 ell, 9 he, 10 log sigma, 11 bad, 12 A, 13 B
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int rq = (int)blockIdx.x - a.has_prev, cc = blockIdx.y;
  const int d = a.d, d4 = (d + 3) >> 2;
  if (rq < 0) {   // heterogeneous workgroup: objective value of the PREVIOUS estimate.  Block 0: a one-workgroup latency chain
    if (cc == 0) {   // (partials from memory, block sums) as long as the rest of the kernel, so it has to start first
      __shared__ double red[6 * 4];
      const T *sig = a.params + d;
      finalize_value_block<T, 256, false, FN>(d, a.prev_vin, a.prev_out, 2 * (int64_t)d, [sig](int i) { return sig[i]; }, red);
    }
    return;
  }
  MIVI_STAMP(a.dbg, 0);
  const uint64_t idx = rng_index(a.rng);
  const bool stl = ent_is_stl(a.out.ent_kind);

  T mu[4], sg[4], tm[4], tis[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = min(4 * rq + r, d - 1);
    mu[r] = a.params[i];
    sg[r] = a.params[d + i];
    tm[r] = (a.target == TGT_DIAG_GAUSS) ? a.t_mean[i] : T(0);
    tis[r] = (a.target == TGT_DIAG_GAUSS) ? a.t_istd[i] : T(0);
  }

  T sW[4] = {0, 0, 0, 0}, sWe[4] = {0, 0, 0, 0};
  T s_ell = 0, s_he = 0, sA = 0, sB = 0;
  const int c_end = min(a.M, (cc + 1) * a.cols_per_cc);
  for (int m = cc * a.cols_per_cc + tid; m < c_end; m += 256) {
    T e[4];
    eps_block<T>(a.rng.seed, idx, (uint64_t)(a.rng.m_offset + m) * (uint64_t)d4 + (uint64_t)rq, e);
    T g[4] = {0, 0, 0, 0};
    if (a.target == TGT_DIAG_GAUSS) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const T z = mu[r] + sg[r] * e[r];
        const T u = (z - tm[r]) * tis[r];
        if (4 * rq + r < d) s_ell += T(-0.5) * u * u;
        g[r] = -u * tis[r];
      }
    } else if (FN && a.target == TGT_FUNNEL) {
      // Neal's funnel + Stacked([log, identity]) fused (see FunnelFin): rows >= 1 need only e1[m] = z[0, m], re-derived
      // from the eps stream; their sum of squares enters row 0 and ell only through x^2 exp(-2 e1) (and that times eps_0)
      T e0q[4];
      if (rq == 0) { e0q[0] = e[0]; }
      else eps_block<T>(a.rng.seed, idx, (uint64_t)(a.rng.m_offset + m) * (uint64_t)d4, e0q);
      funnel_column<T>(rq, d, mu, sg, e, e0q[0], a.params[0], a.params[d], g, s_ell, sA, sB);
    } else if (a.want_grad) {
#pragma unroll
      for (int r = 0; r < 4; ++r) g[r] = a.G[(size_t)m * d + min(4 * rq + r, d - 1)];
    }
    mf_accumulate<T>(rq, d, a.want_grad != 0, stl, FN && a.target == TGT_FUNNEL, sg, e, g, sW, sWe, s_he);
  }
  MIVI_STAMP(a.dbg, 1);

  // ---- reductions: wave totals on the DPP crossbar, one LDS exchange, NV threads finish in fp64 ------------
  {
    T v[NV];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = wave_total63(sW[r]);
      v[4 + r] = wave_total63(sWe[r]);
    }
    v[8] = wave_total63(s_ell);
    v[9] = wave_total63(s_he);
    if (FN) {
      v[NV - 2] = wave_total63(sA);
      v[NV - 1] = wave_total63(sB);
    }
    if (lane == 63) {
#pragma unroll
      for (int k = 0; k < NV; ++k) xw[k][wv] = v[k];
    }
  }
  __syncthreads();
  if (tid < NV) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += (double)xw[tid][j];
    tot[tid < 10 ? tid : tid + 2] = s;
  } else if (tid >= 16 && tid < 20) {   // log-determinant / positivity partial of this block's rows
    const int r = tid - 16, i = 4 * rq + r;
    double lg = 0.0, bad = 0.0;
    if (i < d && cc == 0) {
      const T sv = a.params[d + i];
      lg = (double)log(sv);
      bad = (sv > T(0)) ? 0.0 : 1.0;
    }
    lg += __shfl_xor(lg, 1, 64);
    lg += __shfl_xor(lg, 2, 64);
    bad += __shfl_xor(bad, 1, 64);
    bad += __shfl_xor(bad, 2, 64);
    if (r == 0) {
      tot[10] = lg;
      tot[11] = bad;
    }
  }
  __syncthreads();
  MIVI_STAMP(a.dbg, 2);

  const int nblk = d4 * gridDim.y;
  const int blk = cc * d4 + rq;
  if (a.want_grad && tid < 8) {
    if (a.n_cc == 1) {
      const int r = tid & 3, i = 4 * rq + r;
      if (i < d && !(FN && a.target == TGT_FUNNEL && i == 0)) {
        if (a.out.partials_mode) {
          ((T *)a.out.partials)[(tid < 4 ? 0 : d) + i] = (T)tot[tid];
        } else {
          T *gr = (T *)a.out.grad;
          const double invM = 1.0 / (double)a.out.M_total;
          gr[(tid < 4 ? 0 : d) + i] = mf_grad_entry<T>(tot[tid], invM, tid >= 4, direct_entropy_coeff(a.out.ent_kind), (double)a.params[d + i]);
        }
      }
    } else {
      a.row_part[((size_t)cc * d4 + rq) * 8 + tid] = tot[tid];
    }
  }
  if (tid >= 8 && tid < (FN ? 14 : 12)) a.sc_part[(size_t)(tid - 8) * nblk + blk] = tot[tid];
  MIVI_STAMP(a.dbg, 3);
}

// Row-sum second pass when the columns were split over gridDim.y > 1 workgroups.
template <typename T>
__global__ __launch_bounds__(256) void k_mf_colreduce(MfArgs<T> a) {
  const int d = a.d, d4 = (d + 3) >> 2;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (a.want_grad && t < d4 * 8) {
    const int rq = t >> 3, k = t & 7;
    double s = 0.0;
    for (int cc = 0; cc < a.n_cc; ++cc) s += a.row_part[((size_t)cc * d4 + rq) * 8 + k];
    const int i = 4 * rq + (k & 3);
    if (i < d && !(a.target == TGT_FUNNEL && i == 0)) {   // funnel row 0 belongs to the value workgroup
      if (a.out.partials_mode) {
        ((T *)a.out.partials)[(k < 4 ? 0 : d) + i] = (T)s;
      } else {
        const double invM = 1.0 / (double)a.out.M_total;
        T *gr = (T *)a.out.grad;
        gr[(k < 4 ? 0 : d) + i] = mf_grad_entry<T>(s, invM, k >= 4, direct_entropy_coeff(a.out.ent_kind), (double)a.params[d + i]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Launch-free SGD loop for the mean-field family with the fused diagonal-Gaussian target:
// workgroup b owns rows 4b..4b+3 of (mu, sigma) -- the gradient of those rows needs nothing from any other
// workgroup -- so `n_steps` iterations of {estimate_gradient!, Optimisers.update!, ClipScale}
// (src/algorithms/common.jl:69-104) run inside ONE kernel with the parameters in registers.  The per-step
// scalar partials (sum ell, sum 0.5 eps^2, sum log sigma, #non-positive sigma) go to hist[t][k][block] and are
// assembled into elbo[t] afterwards by k_mf_loop_value.  Arithmetic (and therefore every bit of the result) is
// identical to the launch-per-step path.
// ---------------------------------------------------------------------------------------------
template <typename T>
struct MfLoopArgs {
  int d, M, n_steps, rule;      // rule 0 Descent, 1 Adam
  T *params;                    // [mu; sigma], updated in place
  T *opt_state;                 // Adam: [m (2d); v (2d)]
  const T *t_mean, *t_istd;
  uint64_t seed, idx0;
  int m_offset, M_total, ent_kind;
  long long t0;                 // Adam step count before this call
  double eta, clip_eps, b1, b2, adam_eps;
  double *hist;                 // [n_steps][4][nblk]
  T *grad_out;                  // rule < 0 (estimates at fixed parameters): every estimate writes its gradient here
  T *lane_scratch;              // rule < 0 with estimate lanes (gridDim.y > 1): [lanes][2 d], the gradients of every estimate but the last
};

// RULE is a template parameter: the update rules' scalars (eta, betas, clip, Adam state) would otherwise all be live across the
// Philox block of every variant -- 68 spilled SGPRs in the estimates-only loop.
template <typename T, int RULE>
__global__ __launch_bounds__(256) void k_mf_sgd_loop(MfLoopArgs<T> a) {
  // ONE barrier per iteration: the wave totals of the ten sums go to xw[t & 1] (double-buffered, so the next iteration may
  // write while a slow wave still reads), and after the barrier the eight lanes that own a parameter row read the four
  // per-wave partials of their row directly.  The Adam bias corrections of 256 steps at a time are tabulated by all threads.
  __shared__ T xw[2][10][4];
  __shared__ T cc_tab[256][2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int rq = blockIdx.x, d = a.d, d4 = (d + 3) >> 2, nblk = gridDim.x;
  const int M = a.M, n_steps = a.n_steps;
  constexpr int rule = RULE;
  const bool stl = ent_is_stl(a.ent_kind);
  const double direct = direct_entropy_coeff(a.ent_kind);
  const double invM = 1.0 / (double)a.M_total;
  const T eta = (T)a.eta, b1 = (T)a.b1, b2 = (T)a.b2, aeps = (T)a.adam_eps, ceps = (T)a.clip_eps;
  const bool clip = a.clip_eps == a.clip_eps;   // NaN = no ClipScale (a real epsilon <= 0 is still a ClipScale)
  // every lane holds the four (mu, sigma) rows of this workgroup; lane j < 8 additionally owns the optimiser state
  // of row j (j < 4: mu_j, j >= 4: sigma_{j-4}) and performs that row's update, which is then broadcast
  T mu[4], sg[4], tm[4], tis[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = min(4 * rq + r, d - 1);
    mu[r] = a.params[i];
    sg[r] = a.params[d + i];
    tm[r] = a.t_mean[i];
    tis[r] = a.t_istd[i];
  }
  const int myrow = lane & 7;
  const int myi = min(4 * rq + (myrow & 3), d - 1);
  const bool row_ok = 4 * rq + (myrow & 3) < d;
  double lg = 0.0, bad = 0.0;
  T st_m = 0, st_v = 0;
  if (rule == 1) {
    st_m = a.opt_state[(myrow < 4 ? 0 : d) + myi];
    st_v = a.opt_state[2 * d + (myrow < 4 ? 0 : d) + myi];
  }
  // Estimates at fixed parameters (RULE < 0) are independent: lane blockIdx.y of gridDim.y walks estimates le, le + E, ... -- a row-quad's
  // workgroup is ONE wave per SIMD, and an iteration is a latency chain (Philox, Box-Muller, reductions, one barrier); E of them per CU
  // cover each other (C2: 1.16 -> 0.73 us per estimate with 16 lanes, C5: 1.62 -> 0.58 with 32).  Every estimate runs the
  // same code on the same indices: bitwise unchanged.
  const int le = RULE < 0 ? (int)blockIdx.y : 0, E = RULE < 0 ? (int)gridDim.y : 1;
  int it = 0;
  for (int t = le; t < n_steps; t += E, ++it) {
    if (rule == 1 && (t & 255) == 0) {
      __syncthreads();
      adam_bias<T>(a.t0 + t + tid + 1, a.b1, a.b2, cc_tab[tid][0], cc_tab[tid][1]);
      __syncthreads();
    }
    T sW[4] = {0, 0, 0, 0}, sWe[4] = {0, 0, 0, 0};
    T s_ell = 0, s_he = 0;
    T isg[4] = {0, 0, 0, 0};
    if (stl) {
#pragma unroll
      for (int r = 0; r < 4; ++r) isg[r] = T(1) / sg[r];
    }
    for (int m = tid; m < M; m += 256) {
      T e[4];
      eps_block<T>(a.seed, a.idx0 + (uint64_t)t, (uint64_t)(a.m_offset + m) * (uint64_t)d4 + (uint64_t)rq, e);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = (4 * rq + r) < d;
        const T er = ok ? e[r] : T(0);
        const T z = mu[r] + sg[r] * e[r];
        const T u = (z - tm[r]) * tis[r];
        if (ok) s_ell += T(-0.5) * u * u;
        const T w = ok ? (-u * tis[r] + (stl ? er * isg[r] : T(0))) : T(0);
        s_he += T(0.5) * er * er;
        sW[r] += w;
        sWe[r] += w * er;
      }
    }
    T(*xb)[4] = xw[it & 1];
    {
      T v[10];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = wave_total63(sW[r]);
        v[4 + r] = wave_total63(sWe[r]);
      }
      v[8] = wave_total63(s_ell);
      v[9] = wave_total63(s_he);
      if (lane == 63) {
#pragma unroll
        for (int k = 0; k < 10; ++k) xb[k][wv] = v[k];
      }
    }
    // log-determinant / positivity partial of this workgroup's rows, the same association as k_mf_main: (l0 + l1) + (l2 + l3)
    // (estimates at fixed parameters: the parameters do not move, computed once)
    if (rule >= 0 || it == 0) {
      double lgs[4], bads[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = 4 * rq + r < d;
        lgs[r] = ok ? (double)log(sg[r]) : 0.0;
        bads[r] = (ok && !(sg[r] > T(0))) ? 1.0 : 0.0;
      }
      lg = (lgs[0] + lgs[1]) + (lgs[2] + lgs[3]);
      bad = (bads[0] + bads[1]) + (bads[2] + bads[3]);
    }
    lds_barrier();
    if (tid >= 8 && tid < 12) {   // ell and he: wave partials in index order
      double hv = tid == 10 ? lg : bad;
      if (tid < 10) {
        hv = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) hv += (double)xb[tid][j];
      }
      a.hist[((size_t)t * 4 + (tid - 8)) * nblk + rq] = hv;
    }
    if (rule < 0) {   // estimates at fixed parameters: every estimate's gradient is written (the last one stays)
      if (tid < 8) {
        double trow = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) trow += (double)xb[myrow][j];
        const T sgv = (myrow & 3) == 0 ? sg[0] : (myrow & 3) == 1 ? sg[1] : (myrow & 3) == 2 ? sg[2] : sg[3];
        const T g = mf_grad_entry<T>(trow, invM, myrow >= 4, direct, (double)sgv);
        T *go = (E > 1 && t != n_steps - 1) ? a.lane_scratch + (size_t)le * 2 * d : a.grad_out;   // (the batch's last estimate: the caller's buffer)
        if (row_ok) go[(myrow < 4 ? 0 : d) + myi] = g;
      }
      continue;
    }
    // every lane: total of its row (value lane & 7) in fp64, wave partials in index order; lane j < 8 of every group of eight:
    // gradient of row j exactly as k_mf_main writes it (rounded to T), update, clip
    double trow = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) trow += (double)xb[myrow][j];
    T mine = (myrow < 4) ? mu[0] : sg[0];
#pragma unroll
    for (int r = 1; r < 4; ++r) {
      if ((myrow & 3) == r) mine = (myrow < 4) ? mu[r] : sg[r];
    }
    {
      const T sgv = (myrow & 3) == 0 ? sg[0] : (myrow & 3) == 1 ? sg[1] : (myrow & 3) == 2 ? sg[2] : sg[3];
      const T g = mf_grad_entry<T>(trow, invM, myrow >= 4, direct, (double)sgv);
      if (rule == 0) mine = descent_step(mine, g, eta);
      else mine = adam_step<T>(mine, g, st_m, st_v, cc_tab[t & 255][0], cc_tab[t & 255][1], eta, b1, b2, aeps);
      if (clip && myrow >= 4) mine = clip_step(mine, ceps);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      mu[r] = __shfl(mine, r, 8);
      sg[r] = __shfl(mine, 4 + r, 8);
    }
  }
  if (tid < 8 && rule >= 0) {
    const int i = 4 * rq + (tid & 3);
    if (i < d) {
      T val = (tid < 4) ? mu[0] : sg[0];   // (select chain, not mu[tid & 3]: a dynamic index would put both arrays in scratch
#pragma unroll                            //  memory for the whole kernel -- a memory round trip per loop iteration)
      for (int r = 1; r < 4; ++r)
        if ((tid & 3) == r) val = (tid < 4) ? mu[r] : sg[r];
      a.params[(tid < 4 ? 0 : d) + i] = val;
      if (rule == 1) {
        a.opt_state[(tid < 4 ? 0 : d) + i] = st_m;
        a.opt_state[2 * d + (tid < 4 ? 0 : d) + i] = st_v;
      }
    }
  }
}

// elbo[t] (and the status word) from the per-step partials of k_mf_sgd_loop; one workgroup per step
template <typename T>
__global__ __launch_bounds__(256) void k_mf_loop_value(int d, int nblk, int M_local, int M_total, int ent_kind,
                                                       double ell_const, const double *hist, double *elbo, int *status, T *value_last = nullptr) {
  __shared__ double red[4];
  const int t = blockIdx.x, tid = threadIdx.x;
  double s[4] = {0, 0, 0, 0};
  for (int k = 0; k < 4; ++k)
    for (int i = tid; i < nblk; i += 256) s[k] += hist[((size_t)t * 4 + k) * nblk + i];
  for (int k = 0; k < 4; ++k) s[k] = block_sum<double, 256>(s[k], red);
  if (tid == 0) {
    const double Mt = (double)M_total;
    const double ent = (ent_is_closed(ent_kind) ? 0.5 * d * (1.0 + kLog2Pi) : s[1] / Mt + 0.5 * d * kLog2Pi) + s[2];
    const double value = -((s[0] + (double)M_local * ell_const) / Mt + ent);
    elbo[t] = -value;
    if (value_last && t == (int)gridDim.x - 1) value_last[0] = (T)value;   // (the batch's contract: the LAST estimate's objective value -- no launch of its own)
    int st = 0;
    if (!isfinite(value)) st |= 1;
    if (s[3] > 0.0) st |= 2;
    if (st) atomicOr(status, st);
  }
}

// Estimate lanes of the launch-free batches at fixed parameters: enough workgroups per row-quad for about sixteen waves per SIMD's worth of
// work in flight (a row-quad's workgroup is 4 waves -- or ONE for the funnel shard with n_mc <= 64 -- on a chip of 1024 SIMDs), at most 32,
// at most n_steps.  Measured (C2 / C5, estimates per second): 1 lane 0.86 M / 0.62 M, 4 / 8 lanes 1.22 M / 1.62 M, 8 / 16: 1.34 M / 1.69 M,
// 16 / 32: 1.38 M / 1.71 M.
int mf_loop_lanes(const mivi_ctx *c, int n_steps) {
  const int d4 = (c->cfg.d + 3) / 4;
  const int waves = (c->target == TGT_FUNNEL && c->cfg.n_mc <= 64) ? 1 : 4;
  int e = 16384 / (d4 * waves > 0 ? d4 * waves : 1);
  if (e > 32) e = 32;
  if (e > n_steps) e = n_steps;
  return e < 1 ? 1 : e;
}

template <typename T>
static void mf_sgd_loop_impl(mivi_ctx *c, void *params, void *opt_state, uint64_t idx0, long long t0, int n_steps, int rule,
                             double eta, double clip_eps, double *hist, double *elbo, void *grad_out, void *lane_scratch, void *value_last) {
  MfLoopArgs<T> a;
  a.d = c->cfg.d;
  a.M = c->cfg.n_mc;
  a.n_steps = n_steps;
  a.rule = rule;
  a.params = (T *)params;
  a.opt_state = (T *)opt_state;
  a.t_mean = (const T *)c->t_mean.p;
  a.t_istd = (const T *)c->t_istd.p;
  a.seed = c->cfg.seed;
  a.idx0 = idx0;
  a.m_offset = c->cfg.m_offset;
  a.M_total = c->M_total;
  a.ent_kind = c->cfg.entropy;
  a.t0 = t0;
  a.eta = eta;
  a.clip_eps = clip_eps;
  a.b1 = 0.9;
  a.b2 = 0.999;
  a.adam_eps = 1e-8;
  a.hist = hist;
  a.grad_out = (T *)grad_out;
  a.lane_scratch = (T *)lane_scratch;
  const int d4 = (a.d + 3) / 4;
  if (rule < 0) hipLaunchKernelGGL((k_mf_sgd_loop<T, -1>), dim3(d4, lane_scratch ? mf_loop_lanes(c, n_steps) : 1), dim3(256), 0, c->stream, a);
  else if (rule == 0) hipLaunchKernelGGL((k_mf_sgd_loop<T, 0>), dim3(d4), dim3(256), 0, c->stream, a);
  else hipLaunchKernelGGL((k_mf_sgd_loop<T, 1>), dim3(d4), dim3(256), 0, c->stream, a);
  hipLaunchKernelGGL(k_mf_loop_value<T>, dim3(n_steps), dim3(256), 0, c->stream, a.d, d4, a.M, a.M_total, a.ent_kind,
                     c->t_const, (const double *)hist, elbo, (int *)c->status.p, (T *)value_last);
}

// ---------------------------------------------------------------------------------------------
// The launch-free loop for EVERY rule x operator x averager the reference's ParamSpaceSGD algorithms combine (constructors.jl:44-157;
// DoWG + PolynomialAveraging + ClipScale / ProximalLocationScaleEntropy are the reference's DEFAULTS), mean-field family, diagonal-Gaussian
// target.  k_mf_sgd_loop's structure (workgroup b owns rows 4 b .. 4 b + 3 of (mu, sigma), parameters and optimiser state in registers),
// plus:
//   * DoG / DoWG (src/optimization/rules.jl:17-64) need ||x - x0||^2 and ||g||^2 over ALL parameters before the step: every workgroup leaves
//     its two partials at addresses of this step's own (NaN until then: the data are their own flags); every workgroup then waits for all of
//     them and adds them in index order (the same order everywhere: (v, r) and the step size are the same bits in every workgroup).  One grid-wide
//     exchange per step -- the workgroups must be resident together (d <= 4096 in f32, 2048 in f64); every spin is bounded (status bit 8);
//   * ProximalLocationScaleEntropy (proximal_location_scale_entropy.jl:44-61) and ClipScale on the sigma rows, PolynomialAveraging
//     (averaging.jl:40-47) with the running average in registers: the per-element arithmetic of kernels_update.hip (optim_rules.h).
// The two norms are summed in another order than k_dog_norms / k_dog_update sum them: a DoG / DoWG trajectory equals the launch-per-step
// one to the rounding of those f64 sums (tests/test_gpu_optimize.py::test_meanfield_general_loop), the other rules' bit for bit.
// ---------------------------------------------------------------------------------------------
template <typename T>
struct MfGenLoopArgs {
  int d, M, n_steps, rule, op, averager;   // rule 0 Descent, 1 Adam, 2 DoG, 3 DoWG; op 0 identity, 1 ClipScale, 2 proximal; averager 0 / 1
  T *params, *opt_state;                   // Adam: [m (2d); v (2d)]
  const T *x0;                             // DoG / DoWG: the initial parameters (mivi_dog_init)
  double *dog_sc;                          // DoG / DoWG: (v, r), in / out
  T *avg;                                  // PolynomialAveraging: running average, in / out
  const T *t_mean, *t_istd;
  uint64_t seed, idx0;
  int m_offset, M_total, ent_kind;
  long long t0;
  double eta, clip_eps, b1, b2, adam_eps, avg_eta;
  double *hist;                            // [n_steps][4][nblk]
  double *part;                            // DoG / DoWG: [n_steps][nblk][2] partial norms
  int *status;
  int spin;
};

template <typename T, int RULE>
__global__ __launch_bounds__(256) void k_mf_gen_loop(MfGenLoopArgs<T> a) {
  __shared__ T xw[2][10][4];
  __shared__ T cc_tab[256][2];
  __shared__ double red[2 * 4];
  __shared__ int ok_s;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int rq = blockIdx.x, d = a.d, d4 = (d + 3) >> 2, nblk = gridDim.x;
  const int M = a.M, n_steps = a.n_steps;
  const bool stl = ent_is_stl(a.ent_kind);
  const double direct = direct_entropy_coeff(a.ent_kind);
  const double invM = 1.0 / (double)a.M_total;
  const T eta = (T)a.eta, b1 = (T)a.b1, b2 = (T)a.b2, aeps = (T)a.adam_eps, ceps = (T)a.clip_eps;
  const bool clip = a.op == 1, prox = a.op == 2, averaging = a.averager == 1;
  T mu[4], sg[4], tm[4], tis[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = min(4 * rq + r, d - 1);
    mu[r] = a.params[i];
    sg[r] = a.params[d + i];
    tm[r] = a.t_mean[i];
    tis[r] = a.t_istd[i];
  }
  const int myrow = lane & 7;
  const int myi = min(4 * rq + (myrow & 3), d - 1);
  const bool row_ok = 4 * rq + (myrow & 3) < d;
  const size_t myp = (size_t)(myrow < 4 ? 0 : d) + myi;   // this lane's parameter
  T st_m = 0, st_v = 0, x0v = 0, avgv = 0;
  if (RULE == 1) { st_m = a.opt_state[myp]; st_v = a.opt_state[2 * (size_t)d + myp]; }
  if (RULE >= 2) x0v = a.x0[myp];
  if (averaging) avgv = a.avg[myp];
  double dog_v = 0.0, dog_r = 0.0;
  if (RULE >= 2) { dog_v = a.dog_sc[0]; dog_r = a.dog_sc[1]; }
  double lg = 0.0, bad = 0.0;
  bool lost = false;
  T e_next[4] = {0, 0, 0, 0};   // DoG / DoWG: this thread's first draw of the NEXT step, made while the norms are on their way (the draws do not
  bool have_next = false;       // depend on the parameters)
  for (int t = 0; t < n_steps && !lost; ++t) {
    if (RULE == 1 && (t & 255) == 0) {
      __syncthreads();
      adam_bias<T>(a.t0 + t + tid + 1, a.b1, a.b2, cc_tab[tid][0], cc_tab[tid][1]);
      __syncthreads();
    }
    T sW[4] = {0, 0, 0, 0}, sWe[4] = {0, 0, 0, 0};
    T s_ell = 0, s_he = 0;
    T isg[4] = {0, 0, 0, 0};
    if (stl) {
#pragma unroll
      for (int r = 0; r < 4; ++r) isg[r] = T(1) / sg[r];
    }
    for (int m = tid; m < M; m += 256) {   // (k_mf_sgd_loop's column work: the same values in the same order)
      T e[4];
      if (RULE >= 2 && m == tid && have_next) {   // (drawn under the previous step's exchange)
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = e_next[r];
      } else {
        eps_block<T>(a.seed, a.idx0 + (uint64_t)t, (uint64_t)(a.m_offset + m) * (uint64_t)d4 + (uint64_t)rq, e);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = (4 * rq + r) < d;
        const T er = ok ? e[r] : T(0);
        const T z = mu[r] + sg[r] * e[r];
        const T u = (z - tm[r]) * tis[r];
        if (ok) s_ell += T(-0.5) * u * u;
        const T w = ok ? (-u * tis[r] + (stl ? er * isg[r] : T(0))) : T(0);
        s_he += T(0.5) * er * er;
        sW[r] += w;
        sWe[r] += w * er;
      }
    }
    T(*xb)[4] = xw[t & 1];
    {
      T v[10];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = wave_total63(sW[r]);
        v[4 + r] = wave_total63(sWe[r]);
      }
      v[8] = wave_total63(s_ell);
      v[9] = wave_total63(s_he);
      if (lane == 63) {
#pragma unroll
        for (int k = 0; k < 10; ++k) xb[k][wv] = v[k];
      }
    }
    {
      double lgs[4], bads[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = 4 * rq + r < d;
        lgs[r] = ok ? (double)log(sg[r]) : 0.0;
        bads[r] = (ok && !(sg[r] > T(0))) ? 1.0 : 0.0;
      }
      lg = (lgs[0] + lgs[1]) + (lgs[2] + lgs[3]);
      bad = (bads[0] + bads[1]) + (bads[2] + bads[3]);
    }
    lds_barrier();
    if (tid >= 8 && tid < 12) {
      double hv = tid == 10 ? lg : bad;
      if (tid < 10) {
        hv = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) hv += (double)xb[tid][j];
      }
      a.hist[((size_t)t * 4 + (tid - 8)) * nblk + rq] = hv;
    }
    double trow = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) trow += (double)xb[myrow][j];
    T mine = (myrow < 4) ? mu[0] : sg[0];
#pragma unroll
    for (int r = 1; r < 4; ++r) {
      if ((myrow & 3) == r) mine = (myrow < 4) ? mu[r] : sg[r];
    }
    const T sgv = (myrow & 3) == 0 ? sg[0] : (myrow & 3) == 1 ? sg[1] : (myrow & 3) == 2 ? sg[2] : sg[3];
    const T g = mf_grad_entry<T>(trow, invM, myrow >= 4, direct, (double)sgv);
    double step_gamma = a.eta;   // the proximal operator's step size (Descent: eta; DoG / DoWG: the step just taken)
    if (RULE == 0) {
      mine = descent_step(mine, g, eta);
    } else if (RULE == 1) {
      mine = adam_step<T>(mine, g, st_m, st_v, cc_tab[t & 255][0], cc_tab[t & 255][1], eta, b1, b2, aeps);
    } else {
      // this workgroup's share of the two norms: its eight rows (lanes 0 .. 7 of every group of eight hold them), a fixed xor tree
      double dx = row_ok ? (double)mine - (double)x0v : 0.0, gg = row_ok ? (double)g : 0.0;
      double p0 = dx * dx, p1 = gg * gg;
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) {
        p0 += __shfl_xor(p0, o, 8);
        p1 += __shfl_xor(p1, o, 8);
      }
      // the partials ARE the flags: this step's slots hold NaN until their workgroup has stored them (8-byte stores do not tear), so a
      // reader spins on the data itself -- no acknowledge-then-flag round trip on the producer's side
      if (tid == 0) {
        double *pp = a.part + ((size_t)t * nblk + rq) * 2;
        // (a NaN norm -- diverged parameters -- travels as +Inf: NaN means "not delivered"; the step size and the value come out non-finite either way)
        __hip_atomic_store(pp, p0 == p0 ? p0 : (double)INFINITY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + 1, p1 == p1 ? p1 : (double)INFINITY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok_s = 1;
      }
      __syncthreads();
      have_next = t + 1 < n_steps && tid < M;
      if (have_next) eps_block<T>(a.seed, a.idx0 + (uint64_t)(t + 1), (uint64_t)(a.m_offset + tid) * (uint64_t)d4 + (uint64_t)rq, e_next);
      double sums[2] = {0.0, 0.0};
      for (int k = tid; k < nblk; k += 256) {
        const double *pp = a.part + ((size_t)t * nblk + k) * 2;
        int budget = a.spin;
        double q0, q1;
        while (true) {
          q0 = __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          q1 = __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (q0 == q0 && q1 == q1) break;
          if (--budget <= 0) { atomicAnd(&ok_s, 0); q0 = q1 = 0.0; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        sums[0] += q0;
        sums[1] += q1;
      }
      __syncthreads();
      if (!ok_s) {
        if (tid == 0) atomicOr(a.status, 8);
        lost = true;
        continue;
      }
      block_sum_n<double, 256, 2>(sums, red);
      // (v, r) and the step size: rules.jl:26-42 / :48-64, the arithmetic of k_dog_eta
      dog_r = fmax(sqrt(sums[0]), dog_r);
      double e_t;
      if (RULE == 3) {
        const double r2 = dog_r * dog_r;
        dog_v = dog_v + r2 * sums[1];
        e_t = r2 / sqrt(dog_v);
      } else {
        dog_v = dog_v + sums[1];
        e_t = dog_r / sqrt(dog_v);
      }
      mine = (T)((double)mine - e_t * (double)g);
      step_gamma = (RULE == 3 ? dog_r * dog_r : dog_r) / sqrt(dog_v);
    }
    if (myrow >= 4) {
      if (clip) mine = clip_step(mine, ceps);
      if (prox) mine = prox_entropy_step(mine, (T)step_gamma);
    }
    if (averaging) {
      const double tt = (double)(a.t0 + t + 1);
      const double wa = (a.avg_eta + 1.0) / (tt + a.avg_eta), wb = 1.0 - wa;
      avgv = poly_avg_step<T>(mine, avgv, wa, wb);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      mu[r] = __shfl(mine, r, 8);
      sg[r] = __shfl(mine, 4 + r, 8);
    }
  }
  if (tid < 8) {
    const int i = 4 * rq + (tid & 3);
    if (i < d) {
      T val = (tid < 4) ? mu[0] : sg[0];
#pragma unroll
      for (int r = 1; r < 4; ++r)
        if ((tid & 3) == r) val = (tid < 4) ? mu[r] : sg[r];
      a.params[myp] = val;
      if (RULE == 1) { a.opt_state[myp] = st_m; a.opt_state[2 * (size_t)d + myp] = st_v; }
      if (averaging) a.avg[myp] = avgv;
    }
  }
  if (RULE >= 2 && rq == 0 && tid == 0) { a.dog_sc[0] = dog_v; a.dog_sc[1] = dog_r; }
}

bool mf_gen_loop_ok(const mivi_ctx *c, int rule) {
  if (!(c->cfg.family == MIVI_MEANFIELD && c->target == TGT_DIAG_GAUSS && !c->bij_on && c->cfg.n_mc <= 4096)) return false;
  // DoG / DoWG: a grid-wide exchange per step -- every workgroup resident (f32: six one-wave-per-SIMD workgroups fit a CU, f64: three)
  return rule <= 1 || c->cfg.d <= (c->cfg.dtype == MIVI_F32 ? 4096 : 2048);
}
size_t mf_gen_loop_scratch_bytes(const mivi_ctx *c, int n_steps) {   // partial norms of every step
  const size_t nblk = (size_t)(c->cfg.d + 3) / 4;
  return (size_t)n_steps * nblk * 2 * sizeof(double) + 256;
}

template <typename T>
static bool mf_gen_loop_impl(mivi_ctx *c, void *params, const mivi_loop_t &l, double *hist, double *elbo, char *scratch) {
  MfGenLoopArgs<T> a;
  a.d = c->cfg.d; a.M = c->cfg.n_mc; a.n_steps = l.n_steps; a.rule = l.rule; a.op = l.op; a.averager = l.averager;
  a.params = (T *)params;
  a.opt_state = l.rule == 1 ? (T *)l.opt_state_dev : nullptr;
  a.x0 = l.rule >= 2 ? (const T *)l.opt_state_dev : nullptr;
  a.dog_sc = l.rule >= 2 ? (double *)((char *)l.opt_state_dev + mivi_dog_state_bytes(c) - 16) : nullptr;
  a.avg = l.averager == 1 ? (T *)l.avg_params_dev : nullptr;
  a.t_mean = (const T *)c->t_mean.p; a.t_istd = (const T *)c->t_istd.p;
  a.seed = c->cfg.seed; a.idx0 = l.estimate_idx0; a.m_offset = c->cfg.m_offset; a.M_total = c->M_total; a.ent_kind = c->cfg.entropy;
  a.t0 = (long long)l.t0;
  a.eta = l.eta; a.clip_eps = l.clip_epsilon; a.b1 = l.beta1; a.b2 = l.beta2; a.adam_eps = l.adam_eps; a.avg_eta = l.avg_eta;
  a.hist = hist;
  const int d4 = (a.d + 3) / 4;
  a.part = (double *)scratch;
  a.status = (int *)c->status.p;
  a.spin = 1 << 20;
  // DoG / DoWG: a grid-wide exchange of two norm partials per step by spin-wait -- every workgroup must be resident: checked against the device
  if (l.rule >= 2 && !grid_resident(c, l.rule == 2 ? reinterpret_cast<const void *>(k_mf_gen_loop<T, 2>) : reinterpret_cast<const void *>(k_mf_gen_loop<T, 3>), 256, 0, d4))
    return false;
  if (l.rule >= 2) (void)hipMemsetAsync(a.part, 0xFF, (size_t)l.n_steps * d4 * 2 * sizeof(double), c->stream);   // (NaN: not delivered yet)
  switch (l.rule) {
    case 0: hipLaunchKernelGGL((k_mf_gen_loop<T, 0>), dim3(d4), dim3(256), 0, c->stream, a); break;
    case 1: hipLaunchKernelGGL((k_mf_gen_loop<T, 1>), dim3(d4), dim3(256), 0, c->stream, a); break;
    case 2: hipLaunchKernelGGL((k_mf_gen_loop<T, 2>), dim3(d4), dim3(256), 0, c->stream, a); break;
    default: hipLaunchKernelGGL((k_mf_gen_loop<T, 3>), dim3(d4), dim3(256), 0, c->stream, a); break;
  }
  hipLaunchKernelGGL(k_mf_loop_value<T>, dim3(l.n_steps), dim3(256), 0, c->stream, a.d, d4, a.M, a.M_total, a.ent_kind, c->t_const, (const double *)hist, elbo,
                     (int *)c->status.p);
  return true;
}
// hist: n_steps * 4 * ceil(d / 4) doubles; elbo: n_steps doubles; scratch: mf_gen_loop_scratch_bytes.  false: not launched (the device cannot hold the
// exchanging grid at once): the caller takes the graph of launches
bool launch_mf_gen_loop(mivi_ctx *c, void *params, const mivi_loop_t &l, double *hist, double *elbo, char *scratch) {
  if (c->cfg.dtype == MIVI_F32) return mf_gen_loop_impl<float>(c, params, l, hist, elbo, scratch);
  return mf_gen_loop_impl<double>(c, params, l, hist, elbo, scratch);
}

// rule 0 Descent / 1 Adam: n_steps SGD iterations;  rule -1: n_steps estimates at fixed parameters, the last one's gradient
// into grad_out (what mivi_estimate_gradient_n returns).  elbo[t] of every step / estimate either way.
// lane_scratch (rule < 0): mf_loop_lanes(c, n_steps) * 2 d elements of T, or nullptr = one lane.
void launch_mf_sgd_loop(mivi_ctx *c, void *params, void *opt_state, uint64_t idx0, long long t0, int n_steps, int rule,
                        double eta, double clip_eps, double *hist, double *elbo, void *grad_out, void *lane_scratch, void *value_last) {
  if (c->cfg.dtype == MIVI_F32) mf_sgd_loop_impl<float>(c, params, opt_state, idx0, t0, n_steps, rule, eta, clip_eps, hist, elbo, grad_out, lane_scratch, value_last);
  else mf_sgd_loop_impl<double>(c, params, opt_state, idx0, t0, n_steps, rule, eta, clip_eps, hist, elbo, grad_out, lane_scratch, value_last);
}

// ---------------------------------------------------------------------------------------------
// Launch-free batch of estimates for the mean-field family with the fused FUNNEL target (BASELINE config 5: Neal's funnel +
// Stacked([log, identity]), d = 2048, 64 samples per GPU): n estimates at fixed parameters inside ONE kernel, EVERY estimate writes
// its gradient.  The funnel's only cross-row coupling enters row 0 and ell linearly (FunnelFin), so workgroup b (rows 4b..4b+3)
// needs nothing from the others: per estimate it leaves its eight gradient entries and six scalar partials
// hist[t][{ell, 0.5 eps^2, log sigma, #bad, A, B}][b]; k_mf_funnel_loop_value (one workgroup per estimate, the code of the
// single-call value workgroup: finalize_value_block) adds the O(M) per-column terms, finishes row 0 and the objective.
// Arithmetic and association are those of k_mf_main<T, true>: the results are bitwise those of n single calls.
// NW waves per workgroup: 1 for n_mc <= 64 (config 5's shard: no LDS exchange, no barrier in the loop), 4 beyond.
// ---------------------------------------------------------------------------------------------
template <typename T>
struct MfFunnelLoopArgs {
  int d, M, n_steps;
  const T *params;
  uint64_t seed, idx0;
  int m_offset, M_total, ent_kind;
  double *hist;        // [n_steps][6][nblk]
  T *grad_out;         // rows >= 1 of every estimate (row 0: the value kernel)
  T *lane_scratch;     // estimate lanes (gridDim.y > 1): [lanes][2 d], the gradient rows of every estimate but the last
  const T *e0;         // [n_steps][M]: eps[0, m] of every estimate (k_funnel_e0) -- every row quad needs z[0, m]; nullptr: re-derived per thread
};

// eps[0, m] of estimates idx0 .. idx0 + n_steps - 1: one table for all the row-quad workgroups of k_mf_funnel_loop, each of which would
// otherwise run a second Philox block + Box-Muller pair per column to re-derive it (a quarter of the loop's vector instructions)
template <typename T>
__global__ __launch_bounds__(256) void k_funnel_e0(uint64_t seed, uint64_t idx0, int n_steps, int m_offset, int M, int d4, T *out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_steps * M) return;
  const int t = i / M, m = i - t * M;
  T e[4];
  eps_block<T>(seed, idx0 + (uint64_t)t, (uint64_t)(m_offset + m) * (uint64_t)d4, e);
  out[i] = e[0];
}

template <typename T, int NW>
__global__ __launch_bounds__(64 * NW) void k_mf_funnel_loop(MfFunnelLoopArgs<T> a) {
  constexpr int NT = 64 * NW;
  __shared__ T xw[2][12][4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int rq = blockIdx.x, d = a.d, d4 = (d + 3) >> 2, nblk = gridDim.x;
  const bool stl = ent_is_stl(a.ent_kind);
  const double direct = direct_entropy_coeff(a.ent_kind);
  const double invM = 1.0 / (double)a.M_total;
  T mu[4], sg[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = min(4 * rq + r, d - 1);
    mu[r] = a.params[i];
    sg[r] = a.params[d + i];
  }
  const T mu0 = a.params[0], sg0 = a.params[d];
  double lg, bad;
  {
    double lgs[4], bads[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = 4 * rq + r < d;
      lgs[r] = ok ? (double)log(sg[r]) : 0.0;
      bads[r] = (ok && !(sg[r] > T(0))) ? 1.0 : 0.0;
    }
    lg = (lgs[0] + lgs[1]) + (lgs[2] + lgs[3]);
    bad = (bads[0] + bads[1]) + (bads[2] + bads[3]);
  }
  const int myrow = lane & 7;
  const int myi = 4 * rq + (myrow & 3);
  const bool row_ok = myi < d && myi != 0;   // (row 0: the value kernel)
  const T sgv = (myrow & 3) == 0 ? sg[0] : (myrow & 3) == 1 ? sg[1] : (myrow & 3) == 2 ? sg[2] : sg[3];
  const int le = blockIdx.y, E = gridDim.y;   // estimate lanes: see k_mf_sgd_loop
  int it = 0;
  for (int t = le; t < a.n_steps; t += E, ++it) {
    T sW[4] = {0, 0, 0, 0}, sWe[4] = {0, 0, 0, 0};
    T s_ell = 0, s_he = 0, sA = 0, sB = 0;
    for (int m = tid; m < a.M; m += NT) {
      T e[4], e0q[4];
      eps_block<T>(a.seed, a.idx0 + (uint64_t)t, (uint64_t)(a.m_offset + m) * (uint64_t)d4 + (uint64_t)rq, e);
      if (rq == 0) e0q[0] = e[0];
      else if (a.e0) e0q[0] = a.e0[(size_t)t * a.M + m];
      else eps_block<T>(a.seed, a.idx0 + (uint64_t)t, (uint64_t)(a.m_offset + m) * (uint64_t)d4, e0q);
      T g[4] = {0, 0, 0, 0};
      funnel_column<T>(rq, d, mu, sg, e, e0q[0], mu0, sg0, g, s_ell, sA, sB);
      mf_accumulate<T>(rq, d, true, stl, true, sg, e, g, sW, sWe, s_he);
    }
    T v[12];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = wave_total63(sW[r]);
      v[4 + r] = wave_total63(sWe[r]);
    }
    v[8] = wave_total63(s_ell);
    v[9] = wave_total63(s_he);
    v[10] = wave_total63(sA);
    v[11] = wave_total63(sB);
    T(*xb)[4] = xw[it & 1];
    if (NW > 1) {
      if (lane == 63) {
#pragma unroll
        for (int k = 0; k < 12; ++k) xb[k][wv] = v[k];
      }
      lds_barrier();
    } else {   // one wave: its totals ARE the workgroup's (lane 63 holds them); hand them to the lanes that finish
      if (lane == 63) {
#pragma unroll
        for (int k = 0; k < 12; ++k) xb[k][0] = v[k];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (tid < 8) {   // gradient entries of this workgroup's rows, exactly as k_mf_main writes them
      double trow = 0.0;
#pragma unroll
      for (int j = 0; j < NW; ++j) trow += (double)xb[myrow][j];
      const T g = mf_grad_entry<T>(trow, invM, myrow >= 4, direct, (double)sgv);
      T *go = (E > 1 && t != a.n_steps - 1) ? a.lane_scratch + (size_t)le * 2 * d : a.grad_out;
      if (row_ok) go[(myrow < 4 ? 0 : d) + myi] = g;
    } else if (tid >= 8 && tid < 14) {
      const int k = tid - 8;   // 0 ell, 1 he, 2 log sigma, 3 bad, 4 A, 5 B
      double hv;
      if (k == 2) hv = lg;
      else if (k == 3) hv = bad;
      else {
        const int src = k < 2 ? 8 + k : 6 + k;
        hv = 0.0;
#pragma unroll
        for (int j = 0; j < NW; ++j) hv += (double)xb[src][j];
      }
      a.hist[((size_t)t * 6 + k) * nblk + rq] = hv;
    }
  }
}

template <typename T>
struct MfFunnelValueArgs {
  int d, nblk, n_steps, M, M_total, ent_kind, m_offset;
  const T *params;
  uint64_t seed, idx0;
  double sigma_v, ell_const;
  const double *hist;
  T *value, *grad;       // the LAST estimate's results
  T *scratch;            // [n_steps][d + 2]: value (slot d + 1) and the two row-0 gradient entries (slots 0, d) of the earlier ones
  double *elbo;          // [n_steps]
  int *status;
};

template <typename T>
__global__ __launch_bounds__(256) void k_mf_funnel_loop_value(MfFunnelValueArgs<T> a) {
  __shared__ double red[6 * 4];
  const int t = blockIdx.x, d = a.d, nblk = a.nblk;
  const double *h = a.hist + (size_t)t * 6 * nblk;
  ValueIn vin{};
  vin.fn.ab = h + 4 * (size_t)nblk;
  vin.fn.n_part = nblk;
  vin.fn.params = a.params;
  vin.fn.rng.seed = a.seed;
  vin.fn.rng.idx_base = a.idx0 + (uint64_t)t;
  vin.fn.rng.idx_ptr = nullptr;
  vin.fn.rng.m_offset = a.m_offset;
  vin.fn.d4 = (d + 3) >> 2;
  vin.fn.M = a.M;
  vin.fn.sigma_v = a.sigma_v;
  vin.ell_part2 = h;
  vin.n_ell_part2 = nblk;
  vin.he_part = h + nblk;
  vin.n_he_part = nblk;
  vin.ld_part = h + 2 * (size_t)nblk;
  vin.n_ld_part = nblk;
  vin.ell_const = a.ell_const;
  OutArgs out{};
  const bool last = t == a.n_steps - 1;
  T *sc = a.scratch + (size_t)t * (d + 2);
  out.grad = last ? a.grad : sc;
  out.value = last ? a.value : sc + d + 1;
  out.ent_kind = a.ent_kind;
  out.M_total = a.M_total;
  out.M_local = a.M;
  out.status = a.status;
  out.elbo_rec = a.elbo;
  out.rec_slot = t;
  const T *sig = a.params + d;
  finalize_value_block<T, 256, false, true>(d, vin, out, 2 * (int64_t)d, [sig](int i) { return sig[i]; }, red);
}

template <typename T>
static void mf_funnel_loop_impl(mivi_ctx *c, const void *params, uint64_t idx0, int n_steps, double *hist, double *elbo, void *scratch,
                                void *value, void *grad, void *lane_scratch, void *e0_tab) {
  MfFunnelLoopArgs<T> a;
  a.d = c->cfg.d; a.M = c->cfg.n_mc; a.n_steps = n_steps;
  a.params = (const T *)params;
  a.seed = c->cfg.seed; a.idx0 = idx0;
  a.m_offset = c->cfg.m_offset; a.M_total = c->M_total; a.ent_kind = c->cfg.entropy;
  a.hist = hist;
  a.grad_out = (T *)grad;
  a.lane_scratch = (T *)lane_scratch;
  const int d4 = (a.d + 3) / 4, lanes = lane_scratch ? mf_loop_lanes(c, n_steps) : 1;
  a.e0 = (const T *)e0_tab;
  if (e0_tab)
    hipLaunchKernelGGL(k_funnel_e0<T>, dim3((n_steps * a.M + 255) / 256), dim3(256), 0, c->stream, a.seed, idx0, n_steps, a.m_offset, a.M, d4, (T *)e0_tab);
  if (a.M <= 64) hipLaunchKernelGGL((k_mf_funnel_loop<T, 1>), dim3(d4, lanes), dim3(64), 0, c->stream, a);
  else hipLaunchKernelGGL((k_mf_funnel_loop<T, 4>), dim3(d4, lanes), dim3(256), 0, c->stream, a);
  MfFunnelValueArgs<T> v;
  v.d = a.d; v.nblk = d4; v.n_steps = n_steps; v.M = a.M; v.M_total = a.M_total; v.ent_kind = a.ent_kind; v.m_offset = a.m_offset;
  v.params = a.params; v.seed = a.seed; v.idx0 = idx0;
  v.sigma_v = c->funnel_sigma_v; v.ell_const = c->t_const;
  v.hist = hist; v.value = (T *)value; v.grad = (T *)grad; v.scratch = (T *)scratch; v.elbo = elbo; v.status = (int *)c->status.p;
  hipLaunchKernelGGL(k_mf_funnel_loop_value<T>, dim3(n_steps), dim3(256), 0, c->stream, v);
}

// n_steps estimates of the fused funnel target at fixed parameters in one launch + one finishing launch (see k_mf_funnel_loop).
// hist: n_steps * 6 * ceil(d/4) doubles, elbo: n_steps doubles, scratch: n_steps * (d + 2) elements of T.
// e0_tab: n_steps * n_mc elements of T (eps[0, m] of every estimate, filled here by k_funnel_e0), or nullptr.
void launch_mf_funnel_loop(mivi_ctx *c, const void *params, uint64_t idx0, int n_steps, double *hist, double *elbo, void *scratch,
                           void *value, void *grad, void *lane_scratch, void *e0_tab) {
  if (c->cfg.dtype == MIVI_F32) mf_funnel_loop_impl<float>(c, params, idx0, n_steps, hist, elbo, scratch, value, grad, lane_scratch, e0_tab);
  else mf_funnel_loop_impl<double>(c, params, idx0, n_steps, hist, elbo, scratch, value, grad, lane_scratch, e0_tab);
}

// ---------------------------------------------------------------------------------------------
// Launch-free OPTIMISATION loop for the mean-field family with the fused funnel target (BASELINE config 5's `optimize`): n_steps iterations
// of {estimate_gradient!, Optimisers.update!, ClipScale} (src/algorithms/common.jl:69-104) inside ONE kernel.
// Unlike the diagonal-Gaussian loop (k_mf_sgd_loop) the rows are not independent: row 0 of (mu, sigma) enters every other row's sample
// (z_0 scales the funnel), and every row quad contributes two scalars (A, B: FunnelFin) to row 0's gradient.  So a step is a grid-wide
// exchange: workgroup b (rows 4b .. 4b+3, parameters and optimiser state in registers) computes its rows from the row-0 parameters of this
// step, updates them, and leaves its six scalar partials; the ROW-0 WORKGROUP (the last block) waits for all of them, assembles the
// objective value and row 0's gradient (finalize_value_block: the single calls' code), updates (mu_0, sigma_0) and publishes them for
// the next step.  Two dependent hand-offs across the chip per step (tools/ubench_handoff.hip: 1-2 us each) instead of three launches; the
// draws of step t (Philox + Box-Muller: most of a row quad's instructions) do not depend on the parameters and are made BEFORE the wait.
// Every word that crosses workgroups is an agent-scope atomic (the XCDs' L2s are not coherent); every wait is bounded (status bit 8).
// Arithmetic and association: k_mf_main<T, true> + k_value_funnel + the update kernels, bit for bit (tests/test_gpu_optimize.py).
// All n_blk + 1 workgroups must be resident at once (d <= 16 384: 4 097 workgroups of at most four waves).
// ---------------------------------------------------------------------------------------------
template <typename T>
struct MfFunnelSgdArgs {
  int d, M, n_steps, rule;      // rule 0 Descent, 1 Adam
  T *params, *opt_state;
  uint64_t seed, idx0;
  int m_offset, M_total, ent_kind;
  long long t0;
  double eta, clip_eps, b1, b2, adam_eps, sigma_v, ell_const;
  double *hist;                 // [n_steps][hstride]: the row quads' partials [6][n_blk] of every step at addresses of their own, a multiple of
  long long hstride;            // 128 bytes apart -- read ONCE, after they are complete, so plain (pipelined) loads cannot see a stale line
  unsigned *sync;               // [1] steps whose row-0 parameters are published, [2 + b] steps whose partials row quad b has delivered
  T *pub;                       // [2][2]: (mu_0, sigma_0) at the start of a step, by step parity
  T *gtmp;                      // [d + 1]: row 0's two gradient entries (the row-0 workgroup's scratch)
  double *elbo;                 // [n_steps]
  T *value;
  int *status;
  int spin;
};

template <typename T>
__device__ __forceinline__ void fn_store(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ T fn_load(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one thread waits until *p >= want (bounded); false: gave up
__device__ __forceinline__ bool fn_wait(const unsigned *p, unsigned want, int budget) {
  while ((int)(fn_load(p) - want) < 0) {
    if (--budget <= 0) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  return true;
}

template <typename T, int NW, int RULE>
__global__ __launch_bounds__(256) void k_mf_funnel_sgd_loop(MfFunnelSgdArgs<T> a) {
  constexpr int NT = 64 * NW;
  __shared__ T xw[2][12][4];
  __shared__ T cc_tab[64][2];
  __shared__ T pub_s[2];
  __shared__ int ok_s;
  __shared__ double red[6 * 4];
  constexpr int kFnMirror = 6 * 1024;      // the row-0 workgroup's LDS image of a step's partials (d <= 4096; beyond: read from memory)
  __shared__ double hmir[kFnMirror];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int d = a.d, d4 = (d + 3) >> 2, nblk = d4, n_steps = a.n_steps;
  const bool stl = ent_is_stl(a.ent_kind);
  const double direct = direct_entropy_coeff(a.ent_kind);
  const double invM = 1.0 / (double)a.M_total;
  const T eta = (T)a.eta, b1 = (T)a.b1, b2 = (T)a.b2, aeps = (T)a.adam_eps, ceps = (T)a.clip_eps;
  const bool clip = a.clip_eps == a.clip_eps;   // NaN = no ClipScale

  if ((int)blockIdx.x == nblk) {
    // ---- the row-0 workgroup: objective value, row 0's gradient and update, publication ------------------------------------------------
    T mu0 = a.params[0], sg0 = a.params[d];
    T m_mu = 0, v_mu = 0, m_sg = 0, v_sg = 0;
    if (RULE == 1 && tid == 0) {
      m_mu = a.opt_state[0]; m_sg = a.opt_state[d];
      v_mu = a.opt_state[2 * d]; v_sg = a.opt_state[2 * d + d];
    }
    for (int t = 0; t < n_steps; ++t) {
      if (RULE == 1 && (t & 63) == 0) {
        __syncthreads();
        if (tid < 64) adam_bias<T>(a.t0 + t + tid + 1, a.b1, a.b2, cc_tab[tid][0], cc_tab[tid][1]);
      }
      const double *hg = a.hist + (size_t)t * a.hstride;
      const bool mir = 6 * nblk <= kFnMirror;
      const double *h = mir ? hmir : hg;
      ValueIn vin{};
      vin.fn.wait_word = a.sync + 2;       // (waited for inside, behind the per-column work)
      vin.fn.wait_val = (unsigned)(t + 1);
      vin.fn.wait_n = nblk;
      vin.fn.wait_budget = a.spin;
      if (mir) { vin.fn.mirror = hmir; vin.fn.mirror_src = hg; vin.fn.mirror_n = 6 * nblk; }
      vin.fn.ab = h + 4 * (size_t)nblk;
      vin.fn.n_part = nblk;
      vin.fn.params = a.params;
      vin.fn.row0 = t ? a.pub + 2 * (t & 1) : nullptr;   // row 0's parameters as of the start of this step (step 0: the parameter vector's)
      vin.fn.rng.seed = a.seed;
      vin.fn.rng.idx_base = a.idx0 + (uint64_t)t;
      vin.fn.rng.idx_ptr = nullptr;
      vin.fn.rng.m_offset = a.m_offset;
      vin.fn.d4 = d4;
      vin.fn.M = a.M;
      vin.fn.sigma_v = a.sigma_v;
      vin.ell_part2 = h;
      vin.n_ell_part2 = nblk;
      vin.he_part = h + nblk;
      vin.n_he_part = nblk;
      vin.ld_part = h + 2 * (size_t)nblk;
      vin.n_ld_part = nblk;
      vin.ell_const = a.ell_const;
      OutArgs out{};
      out.grad = a.gtmp;
      out.value = a.value;
      out.ent_kind = a.ent_kind;
      out.M_total = a.M_total;
      out.M_local = a.M;
      out.status = a.status;
      out.elbo_rec = a.elbo;
      out.rec_slot = t;
      const T *sig = a.params + d;
      finalize_value_block<T, 256, false, true>(d, vin, out, 2 * (int64_t)d, [sig](int i) { return sig[i]; }, red);
      __syncthreads();
      if (tid == 0) {
        const T g0 = a.gtmp[0], gs = a.gtmp[d];
        if (RULE == 0) {
          mu0 = descent_step(mu0, g0, eta);
          sg0 = descent_step(sg0, gs, eta);
        } else {
          const T c1 = cc_tab[t & 63][0], c2 = cc_tab[t & 63][1];
          mu0 = adam_step<T>(mu0, g0, m_mu, v_mu, c1, c2, eta, b1, b2, aeps);
          sg0 = adam_step<T>(sg0, gs, m_sg, v_sg, c1, c2, eta, b1, b2, aeps);
        }
        if (clip) sg0 = clip_step(sg0, ceps);
        T *pb = a.pub + 2 * ((t + 1) & 1);
        fn_store(pb, mu0);
        fn_store(pb + 1, sg0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        fn_store(a.sync + 1, (unsigned)(t + 1));
      }
      __syncthreads();
    }
    if (tid == 0) {
      a.params[0] = mu0;
      a.params[d] = sg0;
      if (RULE == 1) {
        a.opt_state[0] = m_mu; a.opt_state[d] = m_sg;
        a.opt_state[2 * d] = v_mu; a.opt_state[2 * d + d] = v_sg;
      }
    }
    return;
  }

  // ---- a row-quad workgroup -----------------------------------------------------------------------------------------------------------
  if (tid >= NT) return;
  const int rq = blockIdx.x;
  T mu[4], sg[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = min(4 * rq + r, d - 1);
    mu[r] = a.params[i];
    sg[r] = a.params[d + i];
  }
  const int myrow = lane & 7;
  const int myi = min(4 * rq + (myrow & 3), d - 1);
  const bool row_ok = 4 * rq + (myrow & 3) < d && !(rq == 0 && (myrow & 3) == 0);   // (row 0: the row-0 workgroup)
  T st_m = 0, st_v = 0;
  if (RULE == 1) {
    st_m = a.opt_state[(myrow < 4 ? 0 : d) + myi];
    st_v = a.opt_state[2 * d + (myrow < 4 ? 0 : d) + myi];
  }
  T mu0 = a.params[0], sg0 = a.params[d];
  bool lost = false;
  for (int t = 0; t < n_steps; ++t) {
    if (RULE == 1 && (t & 63) == 0) {
      if (NW > 1) __syncthreads();
      if (tid < 64) adam_bias<T>(a.t0 + t + tid + 1, a.b1, a.b2, cc_tab[tid][0], cc_tab[tid][1]);
      if (NW > 1) __syncthreads();
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // the draws of this step: independent of the parameters, made before the wait
    const int m = tid;
    const bool has = m < a.M;
    T e[4] = {0, 0, 0, 0}, e0q[4] = {0, 0, 0, 0};
    if (has) {
      eps_block<T>(a.seed, a.idx0 + (uint64_t)t, (uint64_t)(a.m_offset + m) * (uint64_t)d4 + (uint64_t)rq, e);
      if (rq == 0) e0q[0] = e[0];
      else eps_block<T>(a.seed, a.idx0 + (uint64_t)t, (uint64_t)(a.m_offset + m) * (uint64_t)d4, e0q);
    }
    if (t > 0) {   // row 0's parameters as of the start of this step
      if (NW > 1) {
        if (tid == 0) {
          ok_s = fn_wait(a.sync + 1, (unsigned)t, lost ? 64 : a.spin) ? 1 : 0;
          pub_s[0] = fn_load(a.pub + 2 * (t & 1));
          pub_s[1] = fn_load(a.pub + 2 * (t & 1) + 1);
        }
        __syncthreads();
        if (!ok_s) lost = true;
        mu0 = pub_s[0];
        sg0 = pub_s[1];
      } else {
        int okv = 1;
        T p0 = 0, p1 = 0;
        if (lane == 0) {
          okv = fn_wait(a.sync + 1, (unsigned)t, lost ? 64 : a.spin) ? 1 : 0;
          p0 = fn_load(a.pub + 2 * (t & 1));
          p1 = fn_load(a.pub + 2 * (t & 1) + 1);
        }
        okv = __shfl(okv, 0, 64);
        mu0 = __shfl(p0, 0, 64);
        sg0 = __shfl(p1, 0, 64);
        if (!okv) lost = true;
      }
      if (rq == 0) { mu[0] = mu0; sg[0] = sg0; }
    }
    T sW[4] = {0, 0, 0, 0}, sWe[4] = {0, 0, 0, 0};
    T s_ell = 0, s_he = 0, sA = 0, sB = 0;
    if (has) {
      T g[4] = {0, 0, 0, 0};
      funnel_column<T>(rq, d, mu, sg, e, e0q[0], mu0, sg0, g, s_ell, sA, sB);
      mf_accumulate<T>(rq, d, true, stl, true, sg, e, g, sW, sWe, s_he);
    }
    T v[12];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = wave_total63(sW[r]);
      v[4 + r] = wave_total63(sWe[r]);
    }
    v[8] = wave_total63(s_ell);
    v[9] = wave_total63(s_he);
    v[10] = wave_total63(sA);
    v[11] = wave_total63(sB);
    T(*xb)[4] = xw[t & 1];
    if (lane == 63) {
#pragma unroll
      for (int k = 0; k < 12; ++k) xb[k][NW > 1 ? wv : 0] = v[k];
    }
    double lg, bad;
    {
      double lgs[4], bads[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = 4 * rq + r < d;
        lgs[r] = ok ? (double)log(sg[r]) : 0.0;
        bads[r] = (ok && !(sg[r] > T(0))) ? 1.0 : 0.0;
      }
      lg = (lgs[0] + lgs[1]) + (lgs[2] + lgs[3]);
      bad = (bads[0] + bads[1]) + (bads[2] + bads[3]);
    }
    if (NW > 1) lds_barrier();
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (tid >= 8 && tid < 14) {   // this step's partials: 0 ell, 1 he, 2 log sigma, 3 bad, 4 A, 5 B
      const int k = tid - 8;
      double hv;
      if (k == 2) hv = lg;
      else if (k == 3) hv = bad;
      else {
        const int src = k < 2 ? 8 + k : 6 + k;
        hv = 0.0;
#pragma unroll
        for (int j = 0; j < NW; ++j) hv += (double)xb[src][j];
      }
      fn_store(a.hist + (size_t)t * a.hstride + (size_t)k * nblk + rq, hv);
    }
    if (wv == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the six partials have been acknowledged ...
      if (tid == 0) fn_store(a.sync + 2 + rq, (unsigned)(t + 1));   // ... before they are announced (a flag of this workgroup's own)
    }
    // every lane: the total of its row in fp64 (wave partials in index order), the gradient entry exactly as k_mf_main writes it, the update
    double trow = 0.0;
#pragma unroll
    for (int j = 0; j < NW; ++j) trow += (double)xb[myrow][j];
    T mine = (myrow < 4) ? mu[0] : sg[0];
#pragma unroll
    for (int r = 1; r < 4; ++r) {
      if ((myrow & 3) == r) mine = (myrow < 4) ? mu[r] : sg[r];
    }
    if (row_ok) {
      const T sgv = (myrow & 3) == 0 ? sg[0] : (myrow & 3) == 1 ? sg[1] : (myrow & 3) == 2 ? sg[2] : sg[3];
      const T g = mf_grad_entry<T>(trow, invM, myrow >= 4, direct, (double)sgv);
      if (RULE == 0) mine = descent_step(mine, g, eta);
      else mine = adam_step<T>(mine, g, st_m, st_v, cc_tab[t & 63][0], cc_tab[t & 63][1], eta, b1, b2, aeps);
      if (clip && myrow >= 4) mine = clip_step(mine, ceps);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      mu[r] = __shfl(mine, r, 8);
      sg[r] = __shfl(mine, 4 + r, 8);
    }
  }
  if (tid < 8) {
    const int i = 4 * rq + (tid & 3);
    if (i < d && !(rq == 0 && (tid & 3) == 0)) {
      T val = (tid < 4) ? mu[0] : sg[0];
#pragma unroll
      for (int r = 1; r < 4; ++r)
        if ((tid & 3) == r) val = (tid < 4) ? mu[r] : sg[r];
      a.params[(tid < 4 ? 0 : d) + i] = val;
      if (RULE == 1) {
        a.opt_state[(tid < 4 ? 0 : d) + i] = st_m;
        a.opt_state[2 * d + (tid < 4 ? 0 : d) + i] = st_v;
      }
    }
  }
  if (lost && tid == 0 && a.status) atomicOr(a.status, 8);
}

template <typename T>
static bool mf_funnel_sgd_loop_impl(mivi_ctx *c, void *params, void *opt_state, uint64_t idx0, long long t0, int n_steps, int rule, double eta,
                                    double clip_eps, double *hist, unsigned *sync, void *pub, void *gtmp, double *elbo, void *value) {
  MfFunnelSgdArgs<T> a;
  a.d = c->cfg.d; a.M = c->cfg.n_mc; a.n_steps = n_steps; a.rule = rule;
  a.params = (T *)params; a.opt_state = (T *)opt_state;
  a.seed = c->cfg.seed; a.idx0 = idx0;
  a.m_offset = c->cfg.m_offset; a.M_total = c->M_total; a.ent_kind = c->cfg.entropy;
  a.t0 = t0; a.eta = eta; a.clip_eps = clip_eps; a.b1 = 0.9; a.b2 = 0.999; a.adam_eps = 1e-8;
  a.sigma_v = c->funnel_sigma_v; a.ell_const = c->t_const;
  a.hist = hist; a.hstride = (long long)((6 * ((a.d + 3) / 4) + 15) / 16 * 16); a.sync = sync; a.pub = (T *)pub; a.gtmp = (T *)gtmp; a.elbo = elbo; a.value = (T *)value;
  a.status = (int *)c->status.p;
  a.spin = 1 << 22;
  const int d4 = (a.d + 3) / 4;
  const dim3 grid(d4 + 1), block(256);
  const void *kern = a.M <= 64 ? (rule == 0 ? reinterpret_cast<const void *>(k_mf_funnel_sgd_loop<T, 1, 0>) : reinterpret_cast<const void *>(k_mf_funnel_sgd_loop<T, 1, 1>))
                               : (rule == 0 ? reinterpret_cast<const void *>(k_mf_funnel_sgd_loop<T, 4, 0>) : reinterpret_cast<const void *>(k_mf_funnel_sgd_loop<T, 4, 1>));
  if (!grid_resident(c, kern, 256, 0, d4 + 1)) return false;   // (a grid-wide exchange per step: every workgroup resident -- checked, not assumed)
  (void)hipMemsetAsync(sync, 0, (2 + (size_t)d4) * sizeof(unsigned), c->stream);
  if (a.M <= 64) {
    if (rule == 0) hipLaunchKernelGGL((k_mf_funnel_sgd_loop<T, 1, 0>), grid, block, 0, c->stream, a);
    else hipLaunchKernelGGL((k_mf_funnel_sgd_loop<T, 1, 1>), grid, block, 0, c->stream, a);
  } else {
    if (rule == 0) hipLaunchKernelGGL((k_mf_funnel_sgd_loop<T, 4, 0>), grid, block, 0, c->stream, a);
    else hipLaunchKernelGGL((k_mf_funnel_sgd_loop<T, 4, 1>), grid, block, 0, c->stream, a);
  }
  return true;
}
// n_steps optimisation steps of the fused funnel target in ONE launch (n_mc <= 256).  hist: 128-byte aligned, n_steps * roundup(6 * ceil(d/4), 16) doubles; sync: 2 + ceil(d/4) words;
// pub: 4 elements of T; gtmp: d + 1 elements of T; elbo: n_steps doubles; value: one element of T (the last step's objective value).
bool launch_mf_funnel_sgd_loop(mivi_ctx *c, void *params, void *opt_state, uint64_t idx0, long long t0, int n_steps, int rule, double eta,
                               double clip_eps, double *hist, unsigned *sync, void *pub, void *gtmp, double *elbo, void *value) {
  if (c->cfg.dtype == MIVI_F32) return mf_funnel_sgd_loop_impl<float>(c, params, opt_state, idx0, t0, n_steps, rule, eta, clip_eps, hist, sync, pub, gtmp, elbo, value);
  return mf_funnel_sgd_loop_impl<double>(c, params, opt_state, idx0, t0, n_steps, rule, eta, clip_eps, hist, sync, pub, gtmp, elbo, value);
}

// rand(rng, q::MvLocationScale{<:Diagonal}, M): Z = mu + sigma .* eps  (location_scale.jl:80-87)
// lanes run along rows => Z / eps stores are fully coalesced.
template <typename T>
__global__ __launch_bounds__(256) void k_mf_sample(SampleArgs<T> a) {
  __shared__ double red[4];
  const int d = a.d, d4 = (d + 3) >> 2;
  const int rq = blockIdx.x * 256 + threadIdx.x;
  const int m = blockIdx.y;
  const uint64_t idx = rng_index(a.rng);
  T he = 0;
  if (rq < d4) {
    T e[4];
    eps_block<T>(a.rng.seed, idx, (uint64_t)(a.rng.m_offset + m) * (uint64_t)d4 + (uint64_t)rq, e);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * rq + r;
      if (i < d) {
        a.Z[(size_t)m * d + i] = a.params[i] + a.params[d + i] * e[r];
        if (a.eps) a.eps[(size_t)m * a.ld_eps + i] = e[r];
        he += T(0.5) * e[r] * e[r];
      }
    }
  }
  if (a.he_part) {
    const double s = block_sum<double, 256>((double)he, red);
    if (threadIdx.x == 0) a.he_part[blockIdx.y * gridDim.x + blockIdx.x] = s;
  }
}

template <typename T>
static void mf_main_impl(mivi_ctx *c, const void *params, const RngArgs &rng, int M, int want_grad, const void *G,
                         const ValueIn &vin, const OutArgs &out, const ValueJob *prev) {
  MfArgs<T> a;
  a.d = c->cfg.d;
  a.M = M;
  const int d4 = (a.d + 3) / 4;
  int n_cc = 1;
  if (M > 256 && d4 < 512) {
    n_cc = (M + 255) / 256;
    const int cap = (1024 + d4 - 1) / d4;
    if (n_cc > cap) n_cc = cap;
    if (n_cc < 1) n_cc = 1;
  }
  int cols = (M + n_cc - 1) / n_cc;
  cols = (cols + 255) / 256 * 256;
  n_cc = (M + cols - 1) / cols;
  a.n_cc = n_cc;
  a.cols_per_cc = cols;
  a.params = (const T *)params;
  a.rng = rng;
  a.target = (G == nullptr && (c->target == TGT_DIAG_GAUSS || c->target == TGT_FUNNEL)) ? c->target : TGT_NONE;
  a.t_mean = (const T *)c->t_mean.p;
  a.t_istd = (const T *)c->t_istd.p;
  a.G = (const T *)G;
  a.want_grad = want_grad;
  a.row_part = (double *)c->row_part.p;
  a.sc_part = (double *)c->sc_part[c->cur].p;
  a.vin = vin;
  a.out = out;
  a.dbg = c->dbg;
  a.has_prev = prev ? 1 : 0;
  if (prev) {
    a.prev_vin = prev->vin;
    a.prev_out = prev->out;
  } else {
    a.prev_vin = ValueIn{};
    a.prev_out = OutArgs{};
  }
  dim3 grid(d4 + (prev ? 1 : 0), n_cc);
  if (a.target == TGT_FUNNEL) hipLaunchKernelGGL((k_mf_main<T, true>), grid, dim3(256), 0, c->stream, a);
  else hipLaunchKernelGGL((k_mf_main<T, false>), grid, dim3(256), 0, c->stream, a);
  if (n_cc > 1 && want_grad) {
    const int nb = (d4 * 8 + 255) / 256;
    hipLaunchKernelGGL(k_mf_colreduce<T>, dim3(nb), dim3(256), 0, c->stream, a);
  }
  c->mf_nblk = d4 * n_cc;
}

void launch_mf_main(mivi_ctx *c, const void *params, const RngArgs &rng, int M, int want_grad, const void *G,
                    const ValueIn &vin, const OutArgs &out, const ValueJob *prev) {
  if (c->cfg.dtype == MIVI_F32)
    mf_main_impl<float>(c, params, rng, M, want_grad, G, vin, out, prev);
  else
    mf_main_impl<double>(c, params, rng, M, want_grad, G, vin, out, prev);
}

template <typename T>
static void sample_mf_impl(mivi_ctx *c, const void *params, const RngArgs &rng, int M, void *Z, void *eps, int ld_eps,
                           double *he_part) {
  SampleArgs<T> a;
  a.d = c->cfg.d;
  a.M = M;
  a.params = (const T *)params;
  a.rng = rng;
  a.Z = (T *)Z;
  a.eps = (T *)eps;
  a.ld_eps = ld_eps;
  a.epsT = nullptr;
  a.ld_epsT = 0;
  a.he_part = he_part;
  const int d4 = (a.d + 3) / 4;
  dim3 grid((d4 + 255) / 256, M);
  hipLaunchKernelGGL(k_mf_sample<T>, grid, dim3(256), 0, c->stream, a);
}

void launch_sample_mf(mivi_ctx *c, const void *params, const RngArgs &rng, int M, void *Z, void *eps, int ld_eps,
                      double *he_part) {
  if (c->cfg.dtype == MIVI_F32)
    sample_mf_impl<float>(c, params, rng, M, Z, eps, ld_eps, he_part);
  else
    sample_mf_impl<double>(c, params, rng, M, Z, eps, ld_eps, he_part);
}

}  // namespace mivi
