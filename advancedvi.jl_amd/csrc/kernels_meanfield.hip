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

// Scalar partial layout written by k_mf_main and consumed by k_mf_value: sc[k*nblk + blk],
// k = 0 sum ell (variable part), 1 sum 0.5 eps^2, 2 sum log sigma_i (this block's rows), 3 #non-positive sigma.
template <typename T>
__global__ __launch_bounds__(256) void k_mf_main(MfArgs<T> a) {
  __shared__ T xw[10][16];        // [value][wave*4 + row16] partial sums
  __shared__ double tot[12];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int rq = blockIdx.x, cc = blockIdx.y;
  const int d = a.d, d4 = (d + 3) >> 2;
  if (rq >= d4) {   // heterogeneous workgroup: objective value of the PREVIOUS estimate
    if (a.has_prev && cc == 0) {
      __shared__ double red[4];
      const T *sig = a.params + d;
      finalize_value_block<T, 256, false>(d, a.prev_vin, a.prev_out, 2 * (int64_t)d, [sig](int i) { return sig[i]; }, red);
    }
    return;
  }
  MIVI_STAMP(a.dbg, 0);
  const uint64_t idx = rng_index(a.rng);
  const bool stl = ent_is_stl(a.out.ent_kind);

  T mu[4], sg[4], isg[4], tm[4], tis[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = min(4 * rq + r, d - 1);
    mu[r] = a.params[i];
    sg[r] = a.params[d + i];
    tm[r] = (a.target == TGT_DIAG_GAUSS) ? a.t_mean[i] : T(0);
    tis[r] = (a.target == TGT_DIAG_GAUSS) ? a.t_istd[i] : T(0);
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
        if (4 * rq + r < d) s_ell += T(-0.5) * u * u;
        g[r] = -u * tis[r];
      }
    } else if (a.want_grad) {
#pragma unroll
      for (int r = 0; r < 4; ++r) g[r] = a.G[(size_t)m * d + min(4 * rq + r, d - 1)];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = (4 * rq + r) < d;
      const T er = ok ? e[r] : T(0);
      s_he += T(0.5) * er * er;
      if (a.want_grad) {
        isg[r] = T(1) / sg[r];
        const T w = ok ? (g[r] + (stl ? er * isg[r] : T(0))) : T(0);
        sW[r] += w;
        sWe[r] += w * er;
      }
    }
  }
  MIVI_STAMP(a.dbg, 1);

  // ---- reductions: DPP to 16-lane rows, one LDS exchange, 12 threads finish in fp64 ------------
  {
    T v[10];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = row16_sum(sW[r]);
      v[4 + r] = row16_sum(sWe[r]);
    }
    v[8] = row16_sum(s_ell);
    v[9] = row16_sum(s_he);
    if ((lane & 15) == 0) {
      const int slot = wv * 4 + (lane >> 4);
#pragma unroll
      for (int k = 0; k < 10; ++k) xw[k][slot] = v[k];
    }
  }
  __syncthreads();
  if (tid < 10) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += (double)xw[tid][j];
    tot[tid] = s;
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
      if (i < d) {
        if (a.out.partials_mode) {
          ((T *)a.out.partials)[(tid < 4 ? 0 : d) + i] = (T)tot[tid];
        } else {
          T *gr = (T *)a.out.grad;
          const double invM = 1.0 / (double)a.out.M_total;
          if (tid < 4) gr[i] = (T)(-tot[tid] * invM);
          else gr[d + i] = (T)(-tot[tid] * invM - direct_entropy_coeff(a.out.ent_kind) / (double)a.params[d + i]);
        }
      }
    } else {
      a.row_part[((size_t)cc * d4 + rq) * 8 + tid] = tot[tid];
    }
  }
  if (tid >= 8 && tid < 12) a.sc_part[(size_t)(tid - 8) * nblk + blk] = tot[tid];
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
  a.target = (G == nullptr && c->target == TGT_DIAG_GAUSS) ? TGT_DIAG_GAUSS : TGT_NONE;
  a.t_mean = (const T *)c->t_mean.p;
  a.t_istd = (const T *)c->t_istd.p;
  a.G = (const T *)G;
  a.want_grad = want_grad;
  a.row_part = (double *)c->row_part.p;
  a.sc_part = (double *)c->sc_part[c->cur].p;
  a.ticket = nullptr;
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
  hipLaunchKernelGGL(k_mf_main<T>, grid, dim3(256), 0, c->stream, a);
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
