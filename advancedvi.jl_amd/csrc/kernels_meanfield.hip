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
// HBM-bound elementwise work + row reductions: eps is generated in registers (one Philox block =
// rows 4b..4b+3 of one column), lanes run along the sample axis so every row sum is a wave64
// reduction, and the whole estimate is ONE launch: workgroup (b, 0) owns rows 4b..4b+3 for all
// columns, writes their gradient entries directly, and the last workgroup to draw a ticket
// assembles the scalar objective from per-workgroup partials (agent-scope atomics both sides;
// fixed summation order => bitwise reproducible).
#include "device_common.h"

namespace mivi {

template <typename T>
__global__ __launch_bounds__(256) void k_mf_main(MfArgs<T> a) {
  __shared__ double red[4];
  __shared__ double xw[4][12];   // per-wave partials: 0-3 sum W, 4-7 sum W*eps, 8 ell, 9 0.5 eps^2
  __shared__ double tot[12];
  __shared__ int s_last;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int rq = blockIdx.x, cc = blockIdx.y;
  const int d = a.d, d4 = (d + 3) >> 2;
  const uint64_t idx = rng_index(a.rng);
  const bool stl = ent_is_stl(a.out.ent_kind);
  MIVI_STAMP(a.dbg, 0);

  T mu[4], sg[4], isg[4], tm[4], tis[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * rq + r;
    const bool ok = i < d;
    mu[r] = ok ? a.params[i] : T(0);
    sg[r] = ok ? a.params[d + i] : T(1);
    isg[r] = T(1) / sg[r];
    tm[r] = (ok && a.target == TGT_DIAG_GAUSS) ? a.t_mean[i] : T(0);
    tis[r] = (ok && a.target == TGT_DIAG_GAUSS) ? a.t_istd[i] : T(0);
  }

  T sW[4] = {0, 0, 0, 0}, sWe[4] = {0, 0, 0, 0};
  T s_ell = 0, s_he = 0;
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
        s_ell += T(-0.5) * u * u;
        g[r] = -u * tis[r];
      }
    } else if (a.want_grad) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * rq + r;
        g[r] = a.G[(size_t)m * d + min(i, d - 1)];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = (4 * rq + r) < d;
      const T er = ok ? e[r] : T(0);
      s_he += T(0.5) * er * er;
      if (a.want_grad) {
        const T w = ok ? (g[r] + (stl ? er * isg[r] : T(0))) : T(0);
        sW[r] += w;
        sWe[r] += w * er;
      }
    }
  }

  MIVI_STAMP(a.dbg, 1);
  // ---- one wave-level pass (f32 DPP for T = float), one LDS exchange ---------------------------
  {
    double v[10];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = a.want_grad ? wave_sum_fast(sW[r]) : 0.0;
      v[4 + r] = a.want_grad ? wave_sum_fast(sWe[r]) : 0.0;
    }
    v[8] = wave_sum_fast(s_ell);
    v[9] = wave_sum_fast(s_he);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 10; ++k) xw[wv][k] = v[k];
    }
  }
  __syncthreads();
  if (tid < 10) tot[tid] = xw[0][tid] + xw[1][tid] + xw[2][tid] + xw[3][tid];
  // log-determinant / positivity partials of this workgroup's four rows (cc == 0 only counts once)
  if (tid >= 16 && tid < 20) {
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

  if (a.want_grad) {
    if (a.n_cc == 1) {
      if (tid < 4) {
        const int i = 4 * rq + tid;
        if (i < d) {
          if (a.out.partials_mode) {
            T *p = (T *)a.out.partials;
            p[i] = (T)tot[tid];
            p[d + i] = (T)tot[4 + tid];
          } else {
            T *gr = (T *)a.out.grad;
            const double invM = 1.0 / (double)a.out.M_total;
            const double sgi = (double)a.params[d + i];
            gr[i] = (T)(-tot[tid] * invM);
            gr[d + i] = (T)(-tot[4 + tid] * invM - direct_entropy_coeff(a.out.ent_kind) / sgi);
          }
        }
      }
    } else if (tid < 8) {
      a.row_part[((size_t)cc * d4 + rq) * 8 + tid] = tot[tid];
    }
  }

  MIVI_STAMP(a.dbg, 2);
  // ---- scalar partials: [ell | he | logdet | bad] x nblk ----------------------------------------
  const int nblk = gridDim.x * gridDim.y;
  const int blk = cc * gridDim.x + rq;
  if (a.n_cc == 1) {
    if (tid < 4) __hip_atomic_store(a.sc_part + (size_t)tid * nblk + blk, tot[8 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (t == (unsigned)(nblk - 1));
      if (s_last) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    MIVI_STAMP(a.dbg, 3);
    if (s_last) {
      ValueIn vin = a.vin;
      vin.ell_part2 = a.sc_part;
      vin.n_ell_part2 = nblk;
      vin.he_part = a.sc_part + nblk;
      vin.n_he_part = nblk;
      vin.ld_part = a.sc_part + 2 * (size_t)nblk;
      vin.n_ld_part = nblk;
      const T *sig = a.params + d;
      finalize_value_block<T, 256, true>(d, vin, a.out, 2 * (int64_t)d, [sig](int i) { return sig[i]; }, red);
      MIVI_STAMP(a.dbg, 4);
    }
  } else if (tid < 4) {
    a.sc_part[(size_t)tid * nblk + blk] = tot[8 + tid];
  }
}

// second pass when the columns were split over gridDim.y > 1 workgroups
template <typename T>
__global__ __launch_bounds__(256) void k_mf_colreduce(MfArgs<T> a, int nblk_main) {
  __shared__ double red[4];
  const int d = a.d, d4 = (d + 3) >> 2;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (a.want_grad && t < d4 * 8) {
    const int rq = t >> 3, k = t & 7;
    double s = 0.0;
    for (int cc = 0; cc < a.n_cc; ++cc) s += a.row_part[((size_t)cc * d4 + rq) * 8 + k];
    const int i = 4 * rq + (k & 3);
    if (i < d) {
      if (a.out.partials_mode) {
        ((T *)a.out.partials)[(k < 4 ? 0 : d) + i] = (T)s;
      } else {
        const double invM = 1.0 / (double)a.out.M_total;
        T *gr = (T *)a.out.grad;
        if (k < 4)
          gr[i] = (T)(-s * invM);
        else
          gr[d + i] = (T)(-s * invM - direct_entropy_coeff(a.out.ent_kind) / (double)a.params[d + i]);
      }
    }
  }
  if (blockIdx.x == 0) {
    ValueIn vin = a.vin;
    vin.ell_part2 = a.sc_part;
    vin.n_ell_part2 = nblk_main;
    vin.he_part = a.sc_part + nblk_main;
    vin.n_he_part = nblk_main;
    vin.ld_part = a.sc_part + 2 * (size_t)nblk_main;
    vin.n_ld_part = nblk_main;
    const T *sig = a.params + d;
    finalize_value_block<T, 256, false>(d, vin, a.out, 2 * (int64_t)d, [sig](int i) { return sig[i]; }, red);
  }
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
                         const ValueIn &vin, const OutArgs &out) {
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
  a.target = (G == nullptr && c->target == TGT_DIAG_GAUSS) ? TGT_DIAG_GAUSS : TGT_NONE;
  a.t_mean = (const T *)c->t_mean.p;
  a.t_istd = (const T *)c->t_istd.p;
  a.G = (const T *)G;
  a.want_grad = want_grad;
  a.row_part = (double *)c->row_part.p;
  a.sc_part = (double *)c->sc_part.p;
  a.ticket = (unsigned int *)c->ticket.p;
  a.vin = vin;
  a.out = out;
  a.dbg = c->dbg;
  dim3 grid(d4, n_cc);
  hipLaunchKernelGGL(k_mf_main<T>, grid, dim3(256), 0, c->stream, a);
  if (n_cc > 1) {
    const int nb = (d4 * 8 + 255) / 256;
    hipLaunchKernelGGL(k_mf_colreduce<T>, dim3(nb), dim3(256), 0, c->stream, a, d4 * n_cc);
  }
}

void launch_mf_main(mivi_ctx *c, const void *params, const RngArgs &rng, int M, int want_grad, const void *G,
                    const ValueIn &vin, const OutArgs &out) {
  if (c->cfg.dtype == MIVI_F32)
    mf_main_impl<float>(c, params, rng, M, want_grad, G, vin, out);
  else
    mf_main_impl<double>(c, params, rng, M, want_grad, G, vin, out);
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
