// Full-rank (lower-triangular Cholesky scale) RepGradELBO kernels for gfx950.
//
// Reference semantics (AdvancedVI.jl v0.7.0):
//   sampling   Z = scale * eps .+ mu   (d x d  by  d x M)            src/families/location_scale.jl:71-77
//   gradient   d/dC = -(1/M) tril(W eps') - direct * diag(1/C_ii),  d/dmu = -(1/M) W 1   
//   STL term   W += C^-T eps                                         src/algorithms/entropy.jl:59-65,80-90
//
// The two contractions (C eps and W eps') are the only MFMA work in the library. Both are
// "sum_k A[:,k] (x) B[:,k]" with 32-float contiguous operand columns, so one kernel template serves
// both (and the dense-Gaussian target's P (Z-m) product):
//   * one workgroup per 32x32 output tile, NW waves split the K range, v_mfma_f32_32x32x2_f32
//     (exact f32, k-ordered fma chain), operands straight from L2 (one dword per lane per MFMA),
//     partial accumulators reduced through LDS (stride-65 rows => <=2-way bank conflicts), then a
//     fused epilogue (mu add, target, tril mask, -1/M scaling, row sums, objective value).
//   * only lower-triangular tiles are computed for C eps (K <= i) and tril(W eps').
// eps is generated once per estimate by k_eps in both layouts the contractions need:
//   eps [i + m*dP] (column-major, the reference's layout)  and  epsT[m + k*MP].
#include <algorithm>
#include <cstdlib>

#include "device_common.h"
#include "optim_rules.h"

namespace mivi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { MODE_SAMPLE = 0, MODE_VJP = 1, MODE_DENSE = 2 };

// ---------------------------------------------------------------------------------------------
// eps generation (Philox4x32-10 + Box-Muller), both layouts, + sum 0.5 eps^2 partials.
// One 256-thread workgroup = 64 columns x 16 rows (one Philox block per thread).  Runs either as the
// standalone kernel k_eps or as extra workgroups inside the VJP kernel of the PREVIOUS estimate
// (VALU work under the MFMA waves; eps depends only on the estimate index).
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ void eps_tile_block(const SampleArgs<T> &a, int tile, T (*lds)[17], double *red) {
  const int d = a.d, d4 = (d + 3) >> 2;
  const int tid = threadIdx.x;
  const int nrt = (d + 15) >> 4;
  const int i0 = (tile % nrt) * 16, m0 = (tile / nrt) * 64;
  const int ml = tid & 63, rql = tid >> 6;
  const int m = m0 + ml, rq = (i0 >> 2) + rql;
  const uint64_t idx = rng_index(a.rng);
  T e[4] = {0, 0, 0, 0};
  if (m < a.M && rq < d4)
    eps_block<T>(a.rng.seed, idx, (uint64_t)(a.rng.m_offset + m) * (uint64_t)d4 + (uint64_t)rq, e);
  T he = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * rq + r;
    const bool ok = (m < a.M) && (i < d);
    const T v = ok ? e[r] : T(0);
    he += T(0.5) * v * v;
    lds[ml][rql * 4 + r] = v;
    if (a.epsT && ok) a.epsT[(size_t)i * a.ld_epsT + m] = v;   // lanes along m: coalesced
  }
  __syncthreads();
  if (a.eps) {
    const int row = tid & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = (tid >> 4) + 16 * j;
      const int i = i0 + row, mm = m0 + col;
      if (i < d && mm < a.M) a.eps[(size_t)mm * a.ld_eps + i] = lds[col][row];  // 64-byte row segments
    }
  }
  if (a.he_part) {
    const double s = block_sum<double, 256>((double)he, red);
    if (tid == 0) a.he_part[tile] = s;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_eps(SampleArgs<T> a) {
  __shared__ T lds[64][17];
  __shared__ double red[4];
  eps_tile_block<T>(a, blockIdx.x, lds, red);
}
// lane-batched contexts (api_batch.hip): the first draws of up to four contexts as ONE launch (blockIdx.y = lane)
struct EpsMulti { SampleArgs<float> lane[4]; };
// -- in the BLOCKS of the product kernels' eps(t+1) riders (64 rows x 32 columns, one Philox block per thread, 256 contiguous bytes per
// column, the same eight wave sums behind he_part): a lane's first draw is laid down exactly like all its later ones
__global__ __launch_bounds__(512) void k_eps_m(EpsMulti m) {
  __shared__ double red[8];
  const SampleArgs<float> &n = m.lane[blockIdx.y];
  const int tid = threadIdx.x, eb = blockIdx.x, d = n.d, nrb6 = d >> 6;
  const int ri = (eb % nrb6) * 64 + 4 * (tid & 15), rm = (eb / nrb6) * 32 + (tid >> 4);
  float e[4];
  eps_block<float>(n.rng.seed, rng_index(n.rng), (uint64_t)(n.rng.m_offset + rm) * (uint64_t)(d >> 2) + (uint64_t)(ri >> 2), e);
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  const f32x4_t ev = {e[0], e[1], e[2], e[3]};
  store16_wt(n.eps + (size_t)rm * n.ld_eps + ri, ev);
  const float he = 0.5f * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2] + e[3] * e[3]);
  const double sh = block_sum_nodrain_f32<512>(he, red);
  if (tid == 0) n.he_part[eb] = sh;
}

// ---------------------------------------------------------------------------------------------
// Tile decomposition helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void tri_tile(int t, int &ib, int &jb) {
  // t = ib*(ib+1)/2 + jb, 0 <= jb <= ib
  int r = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while (r * (r + 1) / 2 > t) --r;
  while ((r + 1) * (r + 2) / 2 <= t) ++r;
  ib = r;
  jb = t - r * (r + 1) / 2;
}

template <int MODE>
__device__ __forceinline__ void tile_coords(const int d, const int M, int &ib, int &nb_col, int &Ktile, int bidx) {
  if (MODE == MODE_VJP) {
    tri_tile(bidx, ib, nb_col);
    Ktile = M;
  } else {
    const int ncb = (M + 31) >> 5, nrb = (d + 31) >> 5;
    ib = nrb - 1 - bidx / ncb;   // heavy (large K) tiles dispatch first
    nb_col = bidx % ncb;
    Ktile = (MODE == MODE_SAMPLE) ? min(d, 32 * (ib + 1)) : d;
  }
}
template <int MODE>
__device__ __forceinline__ void tile_coords(const int d, const int M, int &ib, int &nb_col, int &Ktile) {
  tile_coords<MODE>(d, M, ib, nb_col, Ktile, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// Epilogue shared by the MFMA and the generic tile kernels.  `get(row, col)` returns the reduced
// accumulator of tile element (row, col); rs_lds[NT] holds per-thread partial row sums of A
// (only meaningful for diagonal VJP tiles).  NT threads, all participate.
// ---------------------------------------------------------------------------------------------
template <typename T, int MODE, int NT, bool FUSED = false, typename Get>
__device__ double tile_epilogue(const FrArgs<T> &a, int ib, int cb, Get get, T *rs_lds, double *red,
                                const T *cii_lds = nullptr) {
  const int tid = threadIdx.x;
  const int d = a.d, M = a.M;
  const int i0 = ib * 32, n0 = cb * 32;
  const int row = tid & 31, cg = tid >> 5;
  constexpr int CG = NT / 32;
  T ell = 0;

  if (MODE == MODE_SAMPLE) {
    const int gi = i0 + row;
    const T mu = gi < d ? a.params[gi] : T(0);
    const T tm = (gi < d && (a.fused_target == TGT_DIAG_GAUSS || a.fused_target == TGT_DENSE_GAUSS)) ? a.t_mean[gi] : T(0);
    const T tis = (gi < d && a.fused_target == TGT_DIAG_GAUSS) ? a.t_istd[gi] : T(0);
#pragma unroll
    for (int col = cg; col < 32; col += CG) {
      const int gm = n0 + col;
      if (gi < d && gm < M) {
        const T z = mu + get(row, col);
        if (a.Z) a.Z[(size_t)gm * d + gi] = z;
        if (a.fused_target == TGT_DIAG_GAUSS) {
          const T u = (z - tm) * tis;
          ell += T(-0.5) * u * u;
          a.W[(size_t)gm * d + gi] = -u * tis;
        }
      }
    }
    if (a.fused_target == TGT_DENSE_GAUSS || a.fused_target == TGT_LOGREG) {
      // second pass, lanes along the sample axis: RT[m + i*MP] = z - t_mean (coalesced); LogReg wants plain z^T
      const int col = tid & 31;
      for (int r2 = cg; r2 < 32; r2 += CG) {
        const int gi2 = i0 + r2, gm = n0 + col;
        if (gi2 < d && gm < M)
          a.RT[(size_t)gi2 * a.MP + gm] =
              a.params[gi2] + get(r2, col) - (a.fused_target == TGT_DENSE_GAUSS ? a.t_mean[gi2] : T(0));
      }
    }
  } else if (MODE == MODE_DENSE) {
    // G = -P (Z - m);  ell_m = 0.5 * sum_i (z-m)_i G_im + const
    const int gi = i0 + row;
    const T tm = gi < d ? a.t_mean[gi] : T(0);
#pragma unroll
    for (int col = cg; col < 32; col += CG) {
      const int gm = n0 + col;
      if (gi < d && gm < M) {
        const T g = -get(row, col);
        const T r = a.Z[(size_t)gm * d + gi] - tm;
        ell += T(0.5) * r * g;
        a.W[(size_t)gm * d + gi] = g;
      }
    }
  } else {  // MODE_VJP: tile (ib, jb = cb) of tril(W eps^T)
    const int jb = cb;
    const int gi = i0 + row;
    const double invM = 1.0 / (double)a.out.M_total;
    const double direct = direct_entropy_coeff(a.out.ent_kind);
    T *dst = a.out.partials_mode ? (T *)a.out.partials : (T *)a.out.grad;
    // optimiser step applied in place (FusedUpdate); a separate instantiation so that the plain VJP kernel carries none of it
    const bool fused = FUSED && !a.out.partials_mode && a.upd.rule >= 0;
    const size_t plen = (size_t)d + (size_t)d * d;
    auto apply_update = [&](size_t pi, T g, bool is_diag) {
      T *pp = (T *)a.upd.params;
      T x;
      if (a.upd.rule == 0) {
        x = descent_step(pp[pi], g, (T)a.upd.eta);
      } else {
        T *st = (T *)a.upd.state;
        T m = st[pi], vv = st[plen + pi];
        x = adam_step<T>(pp[pi], g, m, vv, a.adam_cc[0], a.adam_cc[1], (T)a.upd.eta, (T)a.upd.b1, (T)a.upd.b2, (T)a.upd.eps);
        st[pi] = m;
        st[plen + pi] = vv;
      }
      if (is_diag && a.upd.do_clip) x = clip_step(x, (T)a.upd.clip_eps);
      pp[pi] = x;
    };
    // fused optimiser step: fetch this thread's parameters (and Adam moments) up front -- one memory round trip for
    // all of its elements instead of one per element (the stores below would otherwise order the loads behind them)
    // fused optimiser step: fetch this thread's parameters (and Adam moments) in one batch at the top of the epilogue --
    // one memory round trip for all of its elements instead of one per element (the stores below would otherwise order
    // the loads behind them).  (Fetching them before the main loop was tried: the registers held across the loop cost
    // more than the hidden round trip gained.)
    constexpr int NE = (32 + CG - 1) / CG;
    T px[NE], pm[NE], pv[NE];
    if (fused) {
      const T *pp = (const T *)a.upd.params;
      const T *st = (const T *)a.upd.state;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int col = cg + e * CG, gj = n0 + col;
        const bool mine = col < 32 && gi < d && gj <= gi;
        const size_t pi = mine ? d + (size_t)gj * d + gi : 0;
        px[e] = pp[pi];
        pm[e] = (a.upd.rule == 1) ? st[pi] : T(0);
        pv[e] = (a.upd.rule == 1) ? st[plen + pi] : T(0);
      }
    }
    auto apply_loaded = [&](int e, size_t pi, T g, bool is_diag) {
      T x;
      if (a.upd.rule == 0) {
        x = descent_step(px[e], g, (T)a.upd.eta);
      } else {
        T m = pm[e], vv = pv[e];
        x = adam_step<T>(px[e], g, m, vv, a.adam_cc[0], a.adam_cc[1], (T)a.upd.eta, (T)a.upd.b1, (T)a.upd.b2, (T)a.upd.eps);
        ((T *)a.upd.state)[pi] = m;
        ((T *)a.upd.state)[plen + pi] = vv;
      }
      if (is_diag && a.upd.do_clip) x = clip_step(x, (T)a.upd.clip_eps);
      ((T *)a.upd.params)[pi] = x;
    };
#pragma unroll
    for (int col = cg; col < 32; col += CG) {
      const int gj = n0 + col;
      if (a.out.partials_mode) {   // shard partials: packed lower triangle, nothing above the diagonal
        if (gi < d && gj <= gi) dst[d + (size_t)gj * d - ((size_t)gj * (gj - 1)) / 2 + (gi - gj)] = get(row, col);
        continue;
      }
      if (gi < d && gj < d) {
        T v = get(row, col);
        T o;
        if (gj > gi) {
          o = T(0);
        } else {
          double x = -(double)v * invM;
          if (gi == gj) x -= direct / (double)(cii_lds ? cii_lds[row] : a.params[d + (size_t)gi * d + gi]);
          o = (T)x;
        }
        if (!fused) dst[d + (size_t)gj * d + gi] = o;
        else if (gj <= gi) apply_loaded((col - cg) / CG, d + (size_t)gj * d + gi, o, gi == gj);   // zero gradients above the diagonal move nothing
      }
      if (fused) continue;
      // mirrored strictly-upper tile is structurally zero
      if (jb != ib) {
        const int ui = n0 + row, uj = i0 + col;   // element (ui, uj) with ui < uj
        if (ui < d && uj < d) dst[d + (size_t)uj * d + ui] = T(0);
      }
    }
    if (jb == ib) {  // row sums of W over all samples -> d/dmu; log-determinant partial of this diagonal block
      if (tid < 32) {
        double s = 0.0;
#pragma unroll
        for (int g = 0; g < CG; ++g) s += (double)rs_lds[tid + 32 * g];
        const int gr = i0 + tid;
        if (gr < d) {
          if (fused) apply_update((size_t)gr, (T)(-s * invM), false);
          else dst[gr] = a.out.partials_mode ? (T)s : (T)(-s * invM);
        }
        T lg = 0, bad = 0;
        if (gr < d) {
          const T cii = cii_lds ? cii_lds[tid] : a.params[d + (size_t)gr * d + gr];
          lg = log(cii);
          bad = (cii > T(0)) ? T(0) : T(1);
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {   // lanes 0..31 of wave 0
          lg += __shfl_xor(lg, o, 64);
          bad += __shfl_xor(bad, o, 64);
        }
        if (tid == 0 && a.ld_part) {
          const int nbk = (d + 31) >> 5;
          a.ld_part[ib] = (double)lg;
          a.ld_part[nbk + ib] = (double)bad;
        }
      }
    }
  }
  return (double)ell;
}

// ---------------------------------------------------------------------------------------------
// MFMA tile kernel (float only): v_mfma_f32_32x32x2_f32, NW waves split the K range of the workgroup's
// tile(s) in 32-k blocks; operands come straight from L2 through a double-buffered 32-k stage.
//   A operand lane l: A[i = l&31][k slot = l>>5],  B operand: B[k slot = l>>5][n = l&31]
//   D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
// Work descriptors come from a host-built table (one int2 per workgroup) so that the block -> tile map
// can be XCD-aware (observed dispatch: block b -> XCD b % 8; a wrong guess costs speed only):
//   MODE_SAMPLE : .x = pair index p, .y = column block; the workgroup owns row-blocks p and nb-1-p
//                 (total K = 32(nb+1) for every pair => balanced), XCD x owns pairs p = x (mod 8)
//   MODE_VJP    : .x = ib, .y = jb; XCDs own 8x8 super-blocks of the lower triangle
//   MODE_DENSE  : .x = ib, .y = cb
// Heterogeneous workgroups: blockIdx < a.n_pre run `eps(t+1)` tiles (VJP kernel); the workgroup after the
// last tile assembles the objective value of the PREVIOUS estimate (sample kernel).
// ---------------------------------------------------------------------------------------------
// One wave accumulates 32-k blocks [kb_beg, kb_end) of tile (ib, cb) into `acc`.
// ALIGNED (d and M multiples of 32): every operand read is in bounds, so a stage is 32 loads whose
// addresses are (scalar stage base) + (per-lane 32-bit offsets computed once) -- no per-load VALU -- and the
// triangular mask is applied only in the diagonal k-block of the sampling product.
template <int MODE, bool ALIGNED>
__device__ __forceinline__ void run_kblocks(const FrArgs<float> &a, int ib, int cb, int kb_beg, int kb_end, f32x16 &acc,
                                            float &rs, bool want_rs) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  const int d = a.d, M = a.M;
  const int gi = ib * 32 + l31, n0 = cb * 32;
  const float *Abase;
  int lda, kmaxA;
  const float *Bbase;
  int ldb;
  if (MODE == MODE_SAMPLE) {
    Abase = a.params + d;  lda = d;    kmaxA = d - 1;    Bbase = a.epsT;  ldb = a.MP;
  } else if (MODE == MODE_VJP) {
    Abase = a.W;           lda = d;    kmaxA = M - 1;    Bbase = a.eps;   ldb = a.dP;
  } else {
    Abase = a.t_prec;      lda = a.dP; kmaxA = a.dP - 1; Bbase = a.RT;    ldb = a.MP;
  }
  const bool row_ok = gi < d;
  float a0[16], b0[16], a1[16], b1[16];

  if (ALIGNED) {
    // address = (uniform pointer, SALU arithmetic) + (one per-lane unsigned 32-bit offset)
    const unsigned voffA = (unsigned)(gi + h * lda), voffB = (unsigned)(n0 + l31 + h * ldb);
    auto load_stage = [&](int k, float (&av)[16], float (&bv)[16]) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float *pa = Abase + (size_t)(k + 2 * u) * (size_t)lda;   // uniform
        const float *pb = Bbase + (size_t)(k + 2 * u) * (size_t)ldb;
        av[u] = pa[voffA];
        bv[u] = pb[voffB];
      }
    };
    auto mma_stage = [&](int k, const float (&av)[16], const float (&bv)[16]) {
      const bool masked = (MODE == MODE_SAMPLE) && (k == ib * 32);   // uniform: diagonal block of tril(C)
      if (masked) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const float av_m = (k + 2 * u + h <= gi) ? av[u] : 0.f;
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av_m, bv[u], acc, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          if (MODE == MODE_VJP && want_rs) rs += av[u];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        }
      }
    };
    if (kb_beg < kb_end && MODE == MODE_DENSE) {
      // Long K per wave (dense target: d/8): two stages requested up front (64 loads: the 6-bit vmcnt lets a wave have
      // 63 outstanding), every retired stage requests the one two ahead, and the exits are separate arms.  A branch
      // AROUND the prefetch (the loop below) makes the compiler merge the s_waitcnt of both paths to the conservative
      // one -- the MFMAs of a stage then wait for the loads of the next.  Measured: dense target 13.4 -> 10.9 us.
      int k = kb_beg * 32;
      const int kend = kb_end * 32;
      load_stage(k, a0, b0);
      if (k + 32 >= kend) {
        mma_stage(k, a0, b0);
      } else {
        load_stage(k + 32, a1, b1);
        while (true) {   // invariant: stages k and k+32 exist and are loaded / in flight
          mma_stage(k, a0, b0);
          if (k + 64 >= kend) { mma_stage(k + 32, a1, b1); break; }
          load_stage(k + 64, a0, b0);
          mma_stage(k + 32, a1, b1);
          if (k + 96 >= kend) { mma_stage(k + 64, a0, b0); break; }
          load_stage(k + 96, a1, b1);
          k += 64;
        }
      }
    } else if (kb_beg < kb_end) {
      // Short K per wave (VJP: two stages; the sampling product has its own loop): here the "conservative" form is the
      // faster one -- every wave fires all its loads and then runs its MFMAs as one burst while the other wave of the
      // SIMD is waiting on memory; the exact-wait pipeline above measured 9.2 -> 9.9 us on the VJP kernel (and
      // 11.1 -> 12.9 us on the sampling product).
      int k = kb_beg * 32;
      const int kend = kb_end * 32;
      load_stage(k, a0, b0);
      while (true) {
        const bool more1 = (k + 32) < kend;
        if (more1) load_stage(k + 32, a1, b1);
        mma_stage(k, a0, b0);
        if (!more1) break;
        k += 32;
        const bool more0 = (k + 32) < kend;
        if (more0) load_stage(k + 32, a0, b0);
        mma_stage(k, a1, b1);
        if (!more0) break;
        k += 32;
      }
    }
    return;
  }

  // ---- general shapes: unconditional loads on clamped addresses, operands zeroed at use ----------
  const float *Arow = Abase + (MODE == MODE_DENSE ? gi : min(gi, d - 1));
  const float *Bcol = Bbase + n0 + l31;
  auto load_stage = [&](int k, float (&av)[16], float (&bv)[16]) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int kk = k + 2 * u + h;
      av[u] = Arow[(size_t)min(kk, kmaxA) * lda];
      bv[u] = Bcol[(size_t)kk * ldb];
    }
  };
  auto mma_stage = [&](int k, const float (&av)[16], const float (&bv)[16]) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int kk = k + 2 * u + h;
      bool ok;
      if (MODE == MODE_SAMPLE) ok = row_ok && (kk <= gi);
      else if (MODE == MODE_VJP) ok = row_ok && (kk < M);
      else ok = true;
      const float av_m = ok ? av[u] : 0.f;
      if (MODE == MODE_VJP) rs += av_m;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av_m, bv[u], acc, 0, 0, 0);
    }
  };
  if (kb_beg < kb_end) {
    int k = kb_beg * 32;
    const int kend = kb_end * 32;
    load_stage(k, a0, b0);
    while (true) {
      const bool more1 = (k + 32) < kend;
      if (more1) load_stage(k + 32, a1, b1);
      mma_stage(k, a0, b0);
      if (!more1) break;
      k += 32;
      const bool more0 = (k + 32) < kend;
      if (more0) load_stage(k + 32, a0, b0);
      mma_stage(k, a1, b1);
      if (!more0) break;
      k += 32;
    }
  }
}

// Sampling product, ALIGNED shapes: one continuous double-buffered pipeline over the wave's 32-k blocks of BOTH
// paired tiles (row-blocks ib0 then ib1), so the wave that straddles the junction does not drain and refill.
// Blocks [b0, e0) of tile ib0 are followed by blocks [b1, e1) of tile ib1; the accumulator is picked by a
// wave-uniform branch (no dynamic register indexing).
__device__ __forceinline__ void run_pair_sample(const FrArgs<float> &a, int ib0, int ib1, int cb, int b0, int e0, int b1,
                                                int e1, f32x16 &acc0, f32x16 &acc1) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, h = lane >> 5;
  const int d = a.d;
  const float *Abase = a.params + d;
  const float *Bbase = a.epsT;
  const int lda = d, ldb = a.MP;
  const unsigned voffB = (unsigned)(cb * 32 + l31 + h * ldb);
  const int n0 = e0 - b0, n1 = e1 - b1, n = n0 + n1;
  if (n <= 0) return;
  float a0[16], bb0[16], a1[16], bb1[16], a2[16], bb2[16], a3[16], bb3[16];   // 4-deep stage ring
  auto load_stage = [&](int sidx, float (&av)[16], float (&bv)[16]) {
    const bool second = sidx >= n0;                       // uniform
    const int k = (second ? b1 + (sidx - n0) : b0 + sidx) * 32;
    const unsigned voffA = (unsigned)((second ? ib1 : ib0) * 32 + l31 + h * lda);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const float *pa = Abase + (size_t)(k + 2 * u) * (size_t)lda;   // uniform
      const float *pb = Bbase + (size_t)(k + 2 * u) * (size_t)ldb;
      av[u] = pa[voffA];
      bv[u] = pb[voffB];
    }
  };
  auto mma_stage = [&](int sidx, const float (&av)[16], const float (&bv)[16]) {
    const bool second = sidx >= n0;
    const int ib = second ? ib1 : ib0;
    const int k = (second ? b1 + (sidx - n0) : b0 + sidx) * 32;
    const bool masked = (k == ib * 32);                   // diagonal block of tril(C)
    const int gi = ib * 32 + l31;
    if (!second) {
      if (masked) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32((k + 2 * u + h <= gi) ? av[u] : 0.f, bv[u], acc0, 0, 0, 0);
      } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc0, 0, 0, 0);
      }
    } else {
      if (masked) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32((k + 2 * u + h <= gi) ? av[u] : 0.f, bv[u], acc1, 0, 0, 0);
      } else {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc1, 0, 0, 0);
      }
    }
  };
  int sidx = 0;
  load_stage(0, a0, bb0);
  if (1 < n) load_stage(1, a1, bb1);
  if (2 < n) load_stage(2, a2, bb2);
  while (true) {   // three stages of loads stay in flight behind the MFMAs of the current one
    if (sidx + 3 < n) load_stage(sidx + 3, a3, bb3);
    mma_stage(sidx, a0, bb0);
    if (++sidx >= n) break;
    if (sidx + 3 < n) load_stage(sidx + 3, a0, bb0);
    mma_stage(sidx, a1, bb1);
    if (++sidx >= n) break;
    if (sidx + 3 < n) load_stage(sidx + 3, a1, bb1);
    mma_stage(sidx, a2, bb2);
    if (++sidx >= n) break;
    if (sidx + 3 < n) load_stage(sidx + 3, a2, bb2);
    mma_stage(sidx, a3, bb3);
    if (++sidx >= n) break;
  }
}

template <int MODE, int NW, bool ALIGNED, bool FUSED = false>
__global__ __launch_bounds__(NW * 64) void k_fr_tile_mfma(FrArgs<float> a) {
  constexpr int NT = NW * 64;
  constexpr int NSEG = (MODE == MODE_SAMPLE) ? 2 : 1;
  __shared__ float red_acc[NSEG][NW][16 * 65];
  __shared__ float rs_lds[NT];
  __shared__ float cii_lds[32];
  __shared__ float adam_cc[2];
  __shared__ double red[4 * NW];   // (finalize_value_block sums four values in one pass)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index, provably uniform
  const int d = a.d, M = a.M;
  if (FUSED && a.upd.rule == 1 && tid == 0)                 // visible to everyone after the reduction barrier below
    adam_bias<float>(a.upd.t_base + (a.upd.t_ptr ? *a.upd.t_ptr : 0), a.upd.b1, a.upd.b2, adam_cc[0], adam_cc[1]);
  a.adam_cc = adam_cc;

  // ---- heterogeneous workgroups -----------------------------------------------------------------
  if (MODE == MODE_VJP && (int)blockIdx.x < a.n_pre) {   // eps(t+1) tile
    float(*lds)[17] = reinterpret_cast<float(*)[17]>(&red_acc[0][0][0]);
    eps_tile_block<float>(a.next_eps, blockIdx.x, lds, red);
    return;
  }
  const int widx = (int)blockIdx.x - (MODE == MODE_VJP ? a.n_pre : 0);
  // objective value: of the previous estimate (sample kernel, chained mode) or of THIS estimate (VJP kernel,
  // single calls: everything it sums was written by earlier kernels; log|det C| is taken from the parameters)
  if ((MODE == MODE_SAMPLE || MODE == MODE_VJP) && widx == a.n_work) {
    const float *pp = a.params;
    finalize_value_block<float, NT, false>(d, a.prev_vin, a.prev_out, (int64_t)d + (int64_t)d * d,
                                           [pp, d](int i) { return pp[d + (size_t)i * d + i]; }, red);
    return;
  }
  MIVI_STAMP_K(a.dbg, MODE, 0);
  MIVI_DEV_ONLY(if (a.dbg && threadIdx.x == 0 && blockIdx.x < 4096) a.dbg[((size_t)MODE * 4096 + blockIdx.x) * 8 + 6] = clock64();)
  const int2 wk = a.work_tab[widx];
  if (wk.x < 0) {
    if (tid == 0 && MODE != MODE_VJP) a.ell_part[widx] = 0.0;
    return;
  }
  MIVI_DEV_ONLY(if (a.dbg && threadIdx.x == 0 && blockIdx.x < 4096) a.dbg[((size_t)MODE * 4096 + blockIdx.x) * 8 + 5] = ((long long)wk.x << 16) | wk.y;)

  // ---- segments and the K split -----------------------------------------------------------------
  const int nb = (d + 31) >> 5;
  int seg_ib[2], seg_kb[2];
  int cb = wk.y;
  if (MODE == MODE_SAMPLE) {
    seg_ib[0] = wk.x;
    seg_ib[1] = nb - 1 - wk.x;
    seg_kb[0] = min(nb, seg_ib[0] + 1);                    // K = min(d, 32(ib+1)) in 32-blocks
    seg_kb[1] = (seg_ib[1] > seg_ib[0]) ? min(nb, seg_ib[1] + 1) : 0;
  } else {
    seg_ib[0] = wk.x;
    seg_ib[1] = -1;
    seg_kb[0] = (MODE == MODE_VJP) ? ((M + 31) >> 5) : nb;
    seg_kb[1] = 0;
  }
  const int U = seg_kb[0] + seg_kb[1];
  const int u0 = (w * U) / NW, u1 = ((w + 1) * U) / NW;

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  float rs = 0.f, rs_dummy = 0.f;
  const bool want_rs = (MODE == MODE_VJP) && (seg_ib[0] == cb);   // row sums only on diagonal tiles
  if (MODE == MODE_VJP && want_rs && tid < 32) {                    // C_ii of this diagonal block: issue the loads now
    const int gr = seg_ib[0] * 32 + tid;
    cii_lds[tid] = gr < d ? a.params[d + (size_t)gr * d + gr] : 1.f;
  }
  if (MODE == MODE_SAMPLE && ALIGNED) {
    run_pair_sample(a, seg_ib[0], seg_ib[1], cb, min(u0, seg_kb[0]), min(u1, seg_kb[0]), max(u0, seg_kb[0]) - seg_kb[0],
                    max(u1, seg_kb[0]) - seg_kb[0], acc0, acc1);
  } else {
    run_kblocks<MODE, ALIGNED>(a, seg_ib[0], cb, min(u0, seg_kb[0]), min(u1, seg_kb[0]), acc0, rs, want_rs);
    if (NSEG == 2 && seg_kb[1] > 0)
      run_kblocks<MODE, ALIGNED>(a, seg_ib[1], cb, max(u0, seg_kb[0]) - seg_kb[0], max(u1, seg_kb[0]) - seg_kb[0], acc1,
                                 rs_dummy, false);
  }
  MIVI_STAMP_K(a.dbg, MODE, 1);

#pragma unroll
  for (int r = 0; r < 16; ++r) red_acc[0][w][r * 65 + lane] = acc0[r];
  if (NSEG == 2) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red_acc[NSEG - 1][w][r * 65 + lane] = acc1[r];
  }
  rs_lds[tid] = rs;
  __syncthreads();
  MIVI_STAMP_K(a.dbg, MODE, 2);
  MIVI_DEV_ONLY(if (a.dbg && threadIdx.x == 0 && blockIdx.x < 4096) a.dbg[((size_t)MODE * 4096 + blockIdx.x) * 8 + 7] = clock64();)

  double ell_acc = 0.0;
#pragma unroll
  for (int sg = 0; sg < NSEG; ++sg) {
    if (sg == 1 && seg_kb[1] == 0) break;
    auto get = [&](int row, int col) -> float {
      const int r = (row & 3) + 4 * (row >> 3);
      const int hh = (row >> 2) & 1;
      const int off = r * 65 + col + 32 * hh;
      float s = red_acc[sg][0][off];
#pragma unroll
      for (int ww = 1; ww < NW; ++ww) s += red_acc[sg][ww][off];
      return s;
    };
    ell_acc += tile_epilogue<float, MODE, NT, FUSED>(a, seg_ib[sg], cb, get, rs_lds, red, cii_lds);
  }
  if (MODE != MODE_VJP && (MODE == MODE_DENSE || a.fused_target == TGT_DIAG_GAUSS)) {
    const double s = block_sum<double, NT>(ell_acc, red);
    if (tid == 0) a.ell_part[widx] = s;
  }
  MIVI_STAMP_K(a.dbg, MODE, 3);
}

// ---------------------------------------------------------------------------------------------
// Generic tile kernel (any T, VALU): same tiles and epilogue, 256 threads, LDS-staged 32x32x32.
// Used for MIVI_F64 and as the in-library cross-check of the MFMA kernel.
// ---------------------------------------------------------------------------------------------
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k_fr_tile_generic(FrArgs<T> a) {
  constexpr int NT = 256;
  __shared__ T As[32][33];   // As[k][i]
  __shared__ T Bs[32][33];   // Bs[k][n]
  __shared__ T outt[32][33]; // outt[col][row]
  __shared__ T rs_lds[NT];
  __shared__ double red[4];
  const int tid = threadIdx.x;
  const int d = a.d, M = a.M;
  int ib, cb, Ktile;
  tile_coords<MODE>(d, M, ib, cb, Ktile);
  const int i0 = ib * 32, n0 = cb * 32;
  const int row = tid & 31, cg = tid >> 5;
  T acc[4] = {0, 0, 0, 0};
  T rs = 0;
  for (int k0 = 0; k0 < Ktile; k0 += 32) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kl = cg + 8 * u, kk = k0 + kl;
      const int gi = i0 + row, gn = n0 + row;
      T av = 0, bv = 0;
      if (MODE == MODE_SAMPLE) {
        if (gi < d && kk <= gi && kk < d) av = a.params[d + (size_t)kk * d + gi];
        if (kk < d) bv = a.epsT[(size_t)kk * a.MP + gn];
      } else if (MODE == MODE_VJP) {
        if (gi < d && kk < M) av = a.W[(size_t)kk * d + gi];
        if (kk < M) bv = a.eps[(size_t)kk * a.dP + gn];
      } else {
        if (kk < d) av = a.t_prec[(size_t)kk * a.dP + gi];
        if (kk < d) bv = a.RT[(size_t)kk * a.MP + gn];
      }
      As[kl][row] = av;
      Bs[kl][row] = bv;
      if (MODE == MODE_VJP) rs += av;
    }
    __syncthreads();
#pragma unroll 8
    for (int kl = 0; kl < 32; ++kl) {
      const T av = As[kl][row];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += av * Bs[kl][cg + 8 * j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) outt[cg + 8 * j][row] = acc[j];
  rs_lds[tid] = rs;
  __syncthreads();
  auto get = [&](int r, int c) -> T { return outt[c][r]; };
  const double ell = tile_epilogue<T, MODE, NT>(a, ib, cb, get, rs_lds, red);
  if (MODE != MODE_VJP && (MODE == MODE_DENSE || a.fused_target == TGT_DIAG_GAUSS)) {
    const double s2 = block_sum<double, NT>(ell, red);
    if (tid == 0) a.ell_part[blockIdx.x] = s2;
  }
}

// ---------------------------------------------------------------------------------------------
// MIVI_F64 on the matrix cores: v_mfma_f64_16x16x4_f64 (78.6 TF dense peak).  One 32x32 tile per workgroup
// (the generic tile map), 8 waves split K in 32-k blocks, each wave holds the tile as 2x2 MFMA blocks.
//   A operand lane l: A[row = l&15][k = l>>4],  B: B[k = l>>4][col = l&15],  D: col = l&15, row = (l>>4) + 4*reg
// Operands come straight from L2 (every layout keeps 32 consecutive rows/cols of one k contiguous); loads are
// unconditional on clamped addresses, masks applied at use.  Same epilogue as the other tile kernels.
// ---------------------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void k_fr_tile_mfma64(FrArgs<double> a) {
  constexpr int NT = NW * 64, NS = NW / 2;
  __shared__ double part[NS][32 * 33];   // part[slot][col * 33 + row]
  __shared__ double rs_lds[NT];
  __shared__ double red[4 * NW];   // (finalize_value_block sums four values in one pass)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int d = a.d, M = a.M;
  // heterogeneous workgroups, as in the f32 kernel: leading eps(t+1) tiles (VJP), one value workgroup
  int bidx = (int)blockIdx.x;
  if (MODE == MODE_VJP) {
    if (bidx < a.n_pre) {
      if (tid < 256) eps_tile_block<double>(a.next_eps, bidx, reinterpret_cast<double(*)[17]>(&part[0][0]), red);
      return;
    }
    bidx -= a.n_pre;
  }
  if (MODE != MODE_DENSE && bidx == a.n_work) {
    const double *pp = a.params;
    finalize_value_block<double, NT, false>(d, a.prev_vin, a.prev_out, (int64_t)d + (int64_t)d * d,
                                            [pp, d](int i) { return pp[d + (size_t)i * d + i]; }, red);
    return;
  }
  int ib, cb, Ktile;
  tile_coords<MODE>(d, M, ib, cb, Ktile, bidx);
  const int i0 = ib * 32, n0 = cb * 32;
  const int r16 = lane & 15, ks = lane >> 4;

  const double *A, *B;
  size_t lda, ldb;
  int rowsA;   // rows of A that exist in memory
  if (MODE == MODE_SAMPLE) {
    A = a.params + d; lda = (size_t)d; rowsA = d;
    B = a.epsT; ldb = (size_t)a.MP;
  } else if (MODE == MODE_VJP) {
    A = a.W; lda = (size_t)d; rowsA = d;
    B = a.eps; ldb = (size_t)a.dP;
  } else {
    A = a.t_prec; lda = (size_t)a.dP; rowsA = a.dP;
    B = a.RT; ldb = (size_t)a.MP;
  }
  const int klim = (MODE == MODE_VJP) ? M : Ktile;
  const int gi0 = i0 + r16, gi1 = gi0 + 16;
  const int gn0 = n0 + r16, gn1 = gn0 + 16;

  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

  const int nkb = (klim + 31) >> 5;
  for (int kb = w; kb < nkb; kb += NW) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int k = kb * 32 + 4 * s + ks;
      const bool kok = k < klim;
      const bool ok0 = kok && gi0 < rowsA && (MODE != MODE_SAMPLE || k <= gi0);
      const bool ok1 = kok && gi1 < rowsA && (MODE != MODE_SAMPLE || k <= gi1);
      const size_t ka = kok ? (size_t)k : 0;
      double a0 = A[ka * lda + (gi0 < rowsA ? gi0 : 0)];
      double a1 = A[ka * lda + (gi1 < rowsA ? gi1 : 0)];
      double b0 = B[ka * ldb + gn0];
      double b1 = B[ka * ldb + gn1];
      a0 = ok0 ? a0 : 0.0;
      a1 = ok1 ? a1 : 0.0;
      b0 = kok ? b0 : 0.0;
      b1 = kok ? b1 : 0.0;
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
  }

  // ---- cross-wave reduction: waves 4..7 park their tiles, waves 0..3 add their own on top ----------------
  if (w >= NS) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[w - NS][(16 * j + r16) * 33 + 16 * i + ks + 4 * r] = acc[i][j][r];
  }
  // row sums of W (d/dmu) for diagonal VJP tiles: thread (row = tid & 31, group = tid >> 5) strides the samples
  double rs = 0.0;
  if (MODE == MODE_VJP && ib == cb) {
    const int gi = i0 + (tid & 31);
    if (gi < d)
      for (int k = tid >> 5; k < M; k += NT / 32) rs += a.W[(size_t)k * d + gi];
  }
  rs_lds[tid] = rs;
  __syncthreads();
  if (w < NS) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[w][(16 * j + r16) * 33 + 16 * i + ks + 4 * r] += acc[i][j][r];
  }
  __syncthreads();
  auto get = [&](int r, int c) -> double {
    const int o = c * 33 + r;
    double v = part[0][o];
#pragma unroll
    for (int q = 1; q < NS; ++q) v += part[q][o];
    return v;
  };
  const double ell = tile_epilogue<double, MODE, NT>(a, ib, cb, get, rs_lds, red);
  if (MODE != MODE_VJP && (MODE == MODE_DENSE || a.fused_target == TGT_DIAG_GAUSS)) {
    const double s2 = block_sum<double, NT>(ell, red);
    if (tid == 0) a.ell_part[bidx] = s2;
  }
}

// ---------------------------------------------------------------------------------------------
// STL term for the full-rank family: W += C^-T eps  (back substitution with C^T, 8 columns per
// workgroup, blocked by 32 rows).  O(d^2 M); the reference's own docs call this the expensive
// estimator (docs/src/klminrepgraddescent.md:93-95).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_fr_stl(FrArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *x = (T *)smem_raw;                 // x[col][dP]   (in place: rhs -> solution)
  const int d = a.d, M = a.M, dP = a.dP;
  T *Cd = x + 8 * (size_t)dP;           // Cd[s][r] = C[i0+s, i0+r] (33 stride)
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * 8;
  const T *C = a.params + d;
  const int nb = (d + 31) >> 5;
  for (int t = tid; t < 8 * dP; t += 256) {
    const int col = t / dP, i = t - col * dP;
    const int m = m0 + col;
    x[t] = (i < d && m < M) ? a.eps[(size_t)m * dP + i] : T(0);
  }
  __syncthreads();
  for (int b = nb - 1; b >= 0; --b) {
    const int i0 = b * 32;
    for (int t = tid; t < 32 * 32; t += 256) {
      const int s = t >> 5, r = t & 31;   // want C[i0+s, i0+r], s >= r
      const int gs = i0 + s, gr = i0 + r;
      T v = 0;
      if (gs < d && gr < d && gs >= gr) v = C[(size_t)gr * d + gs];
      if (gs >= d && s == r) v = 1;       // padding rows: identity
      Cd[s * 33 + r] = v;
    }
    __syncthreads();
    {  // solve the 32x32 block: 8 columns x 32 lanes; lane r holds row i0+r
      const int col = tid >> 5, r = tid & 31;
      T v = x[(size_t)col * dP + i0 + r];
      for (int s = 31; s >= 0; --s) {
        const T xs = __shfl(v, s, 32) / Cd[s * 33 + s];
        if (r == s) v = xs;
        if (r < s) v -= Cd[s * 33 + r] * xs;   // C[i0+s, i0+r] * x_s
      }
      x[(size_t)col * dP + i0 + r] = v;
    }
    __syncthreads();
    // right-looking update of rows i < i0: rhs_i -= sum_k C[i0+k, i] * x_{i0+k}
    for (int i = tid; i < i0; i += 256) {
      T accv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const T *cc = C + (size_t)i * d + i0;
      const int kmax = min(32, d - i0);
      for (int k = 0; k < kmax; ++k) {
        const T c = cc[k];
#pragma unroll
        for (int col = 0; col < 8; ++col) accv[col] += c * x[(size_t)col * dP + i0 + k];
      }
#pragma unroll
      for (int col = 0; col < 8; ++col) x[(size_t)col * dP + i] -= accv[col];
    }
    __syncthreads();
  }
  for (int t = tid; t < 8 * dP; t += 256) {
    const int col = t / dP, i = t - col * dP;
    const int m = m0 + col;
    if (i < d && m < M) a.W[(size_t)m * d + i] += x[t];
  }
}

// ---------------------------------------------------------------------------------------------
// STL term, f32 MFMA route:  W += X,  C^T X = eps  (blocked back substitution).
//   pre-kernel  k_stl_prep : CT[i + k*dP] = C[k, i]  (the lower triangle transposed, so that the update's A operand
//               has its lanes along the output row) and DinvT[b][i + 32 k] = (C_bb^{-1})[k, i] for every 32x32
//               diagonal block b (forward substitution, one workgroup per block).
//   main kernel k_stl_solve: one workgroup per 32 sample columns; X (dP x 32) lives in LDS as x[k][m].
//               for b = nb-1 .. 0:  S = sum_{j>b} CT(b, j) X_j   (32-k units split over the waves, MFMA, LDS reduce)
//                                   X_b = DinvT_b (eps_b - S)     (16 MFMAs)
//               The column groups are independent, the row blocks are inherently sequential: the reference's
//               docs call this estimator O(d^3) per step for the same reason (docs/src/klminrepgraddescent.md:93-95).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_stl_prep(int d, int dP, const T *C, T *CT, T *DinvT) {
  __shared__ T tile[32][33];
  const int nb = (d + 31) >> 5;
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < nb) {          // ---- inverse of diagonal block b ----
    const int b = blockIdx.x, i0 = b * 32;
    for (int t = tid; t < 1024; t += 256) {
      const int r = t & 31, c = t >> 5;         // D[r][c], lower triangular (identity on padding)
      const int gr = i0 + r, gc = i0 + c;
      T v = 0;
      if (gr < d && gc < d && gr >= gc) v = C[(size_t)gc * d + gr];
      if (gr >= d && r == c) v = 1;
      tile[r][c] = v;
    }
    __syncthreads();
    if (tid < 32) {                     // lane c: column c of D^{-1} by forward substitution
      const int c = tid;
      T y[32];
#pragma unroll
      for (int r = 0; r < 32; ++r) y[r] = 0;
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        T sacc = (r == c) ? T(1) : T(0);
#pragma unroll
        for (int k = 0; k < r; ++k) sacc -= tile[r][k] * y[k];
        y[r] = sacc / tile[r][r];
      }
#pragma unroll
      for (int r = 0; r < 32; ++r) DinvT[(size_t)b * 1024 + c + 32 * r] = y[r];   // DinvT[i=c][k=r] = Dinv[r][c]
    }
    return;
  }
  // ---- transpose tile (ib, jb), ib >= jb: CT[(jb*32 + c) + (ib*32 + r)*dP] = C[ib*32 + r, jb*32 + c] ----
  int ib, jb;
  tri_tile((int)blockIdx.x - nb, ib, jb);
  for (int t = tid; t < 1024; t += 256) {
    const int r = t & 31, c = t >> 5;
    const int gr = ib * 32 + r, gc = jb * 32 + c;
    tile[c][r] = (gr < d && gc < d && gr >= gc) ? C[(size_t)gc * d + gr] : T(0);
  }
  __syncthreads();
  for (int t = tid; t < 1024; t += 256) {
    const int c = t & 31, r = t >> 5;
    CT[(size_t)(ib * 32 + r) * dP + jb * 32 + c] = tile[c][r];
  }
}

// Blocked back substitution with look-ahead: every block row is  S_b = sum_{j>b} CT(b,j) X_j -> X_b = Dinv_b (eps_b - S_b), and only
// the LAST term of S_b (j = b+1) depends on the block solved just before.  Wave 0 is the sequential chain (last term of row b,
// R_b = eps_b - bulk_b - last, X_b = Dinv_b R_b), waves 1..NW-1 already accumulate the bulk of row b-1 next to it: a block row costs
// max(chain, bulk / 7 waves) instead of their sum.  (The strictly sequential and the 32-column variants this replaced were removed
// in round 3: no default path reached them.)
// 16 sample columns per workgroup on v_mfma_f32_16x16x4_f32: the look-ahead solve is bound by the MFMA throughput of
// the CUs it runs on, and M / 32 workgroups are only 8 CUs at M = 256.  Half the columns per workgroup = twice the CUs at
// half the MFMA time per 32x32 unit (16 MFMAs of 8 passes instead of 16 of 16 passes).
//   A operand lane l: A[row = l&15][k = l>>4],  B: B[k = l>>4][col = l&15],  D reg r: row = 4*(l>>4) + r, col = l&15
// The same kernel serves MIVI_F64 through v_mfma_f64_16x16x4_f64: identical A / B operand shapes, only the accumulator
// rows differ (f32: row = 4*(l>>4) + r;  f64: row = (l>>4) + 4*r).
template <typename T>
struct Mfma16;
template <>
struct Mfma16<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int kq, int r) { return 4 * kq + r; }
};
template <>
struct Mfma16<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int kq, int r) { return kq + 4 * r; }
};

template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) void k_stl_solve_la16(FrArgs<T> a, const T *CT, const T *DinvT) {
  typedef typename Mfma16<T>::acc_t f32x4a;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *x = (T *)smem_raw;                                   // x[k*16 + m], k < dP
  const int d = a.d, M = a.M, dP = a.dP;
  T *part = x + (size_t)dP * 16;                      // part[w][8*64]; part[0] = reduced bulk of the current row
  const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, kq = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * 16;
  const int nb = (d + 31) >> 5;
  for (int t = tid; t < dP * 16; t += NW * 64) {
    const int k = t % dP, m = t / dP;                     // lanes along k: coalesced global reads
    x[k * 16 + m] = (m0 + m < M) ? a.eps[(size_t)(m0 + m) * dP + k] : T(0);
  }
  for (int t = tid; t < 8 * 64; t += NW * 64) part[t] = T(0);
  // one 32x32 unit of A = CT(brow, j): av[2g + hb] = A[row = 16*hb + c16][k = 4g + kq]
  // up to four units of the NEXT block row are requested at the end of the current one (their L2 latency, ~0.7 us,
  // then hides under the barriers and the chain); a fifth unit (only the last few block rows have one) is fetched in place
  T av0[16], av1[16], av2[16], av3[16];
  auto loadA = [&](int brow, int j, T (&av)[16]) {
    const T *Arow = CT + brow * 32 + c16 + (size_t)(j * 32 + kq) * dP;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      av[2 * g] = Arow[(size_t)(4 * g) * dP];
      av[2 * g + 1] = Arow[(size_t)(4 * g) * dP + 16];
    }
  };
  __syncthreads();
  T dv[16];
  auto loadD = [&](int b) {   // inverted diagonal block, same operand shape: A[i][k] = DinvT[i + 32 k]
    const T *Di = DinvT + (size_t)b * 1024 + c16 + 32 * kq;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      dv[2 * g] = Di[32 * 4 * g];
      dv[2 * g + 1] = Di[32 * 4 * g + 16];
    }
  };
  if (w == 0) {
    loadD(nb - 1);
    __builtin_amdgcn_s_setprio(3);   // the sequential chain shares its SIMD with a bulk wave: let it issue first
  }
  for (int b = nb - 1; b >= 0; --b) {
    f32x4a acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};   // rows 0..15 / 16..31 of the block
    auto mma = [&](int j, const T (&av)[16]) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const T bv = x[(j * 32 + 4 * g + kq) * 16 + c16];
        acc0 = Mfma16<T>::mma(av[2 * g], bv, acc0);
        acc1 = Mfma16<T>::mma(av[2 * g + 1], bv, acc1);
      }
    };
    if (w == 0) {
      // ---- the sequential chain ------------------------------------------------------------------
      if (b + 1 < nb) mma(b + 1, av0);                    // av0: CT(b, b+1), fetched during the previous block row
      if (b > 0) loadA(b - 1, b, av0);
      // R_b = eps_b - bulk_b - last term (accumulator layout)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        T *x0 = &x[(b * 32 + Mfma16<T>::row(kq, r)) * 16 + c16], *x1 = x0 + 16 * 16;
        *x0 = *x0 - part[r * 64 + lane] - acc0[r];
        *x1 = *x1 - part[(4 + r) * 64 + lane] - acc1[r];
      }
      f32x4a xa0 = {0, 0, 0, 0}, xa1 = {0, 0, 0, 0};
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const T bv = x[(b * 32 + 4 * g + kq) * 16 + c16];
        xa0 = Mfma16<T>::mma(dv[2 * g], bv, xa0);
        xa1 = Mfma16<T>::mma(dv[2 * g + 1], bv, xa1);
      }
      if (b > 0) loadD(b - 1);                            // next block row's inverse: in flight across the barriers
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        x[(b * 32 + Mfma16<T>::row(kq, r)) * 16 + c16] = xa0[r];
        x[(b * 32 + 16 + Mfma16<T>::row(kq, r)) * 16 + c16] = xa1[r];
      }
    } else if (b > 0) {
      // ---- bulk of block row b-1: units j = b+1 .. nb-1 dealt to waves 1 .. NW-1 ------------------------
      const int j = b + w, S = NW - 1;                     // units j, j+S, j+2S, ... < nb (at most five at nb = 32)
      if (j + 4 * S < nb) {                                // rare fifth unit: request it before the burst, use it last
        T av4[16];
        loadA(b - 1, j + 4 * S, av4);
        mma(j, av0); mma(j + S, av1); mma(j + 2 * S, av2); mma(j + 3 * S, av3);
        for (int jj = j + 4 * S; jj < nb; jj += S) {       // (and any beyond, for nb > 32, one by one)
          if (jj != j + 4 * S) loadA(b - 1, jj, av4);
          mma(jj, av4);
        }
      } else {
        if (j < nb) mma(j, av0);
        if (j + S < nb) mma(j + S, av1);
        if (j + 2 * S < nb) mma(j + 2 * S, av2);
        if (j + 3 * S < nb) mma(j + 3 * S, av3);
      }
      if (b > 1) {                                         // next block row (b-2): units b-1+w, +S, +2S, +3S
        const int jn = b - 1 + w;
        if (jn < nb) loadA(b - 2, jn, av0);
        if (jn + S < nb) loadA(b - 2, jn + S, av1);
        if (jn + 2 * S < nb) loadA(b - 2, jn + 2 * S, av2);
        if (jn + 3 * S < nb) loadA(b - 2, jn + 3 * S, av3);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        part[w * 512 + r * 64 + lane] = acc0[r];
        part[w * 512 + (4 + r) * 64 + lane] = acc1[r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) part[w * 512 + r * 64 + lane] = T(0);
    }
    __syncthreads();
    if (tid < 512) {
      T sacc = part[512 + tid];
#pragma unroll
      for (int ww = 2; ww < NW; ++ww) sacc += part[ww * 512 + tid];
      part[tid] = sacc;
    }
    __syncthreads();
  }
  // W[i + m*d] += X[i, m]
  for (int t = tid; t < dP * 16; t += NW * 64) {
    const int i = t % dP, m = t / dP;
    if (i < d && m0 + m < M) a.W[(size_t)(m0 + m) * d + i] += x[i * 16 + m];
  }
}

// RT[m + i*MP] = Z[i + m*d] - t_mean[i]: feeds the dense-Gaussian target product when Z was not
// produced by the full-rank sample kernel (mean-field family).  64x64 LDS-tiled transpose.
template <typename T>
__global__ __launch_bounds__(256) void k_rt_from_z(int d, int M, int MP, const T *Z, const T *t_mean, T *RT) {
  __shared__ T tile[64][65];
  const int i0 = blockIdx.x * 64, m0 = blockIdx.y * 64, tid = threadIdx.x;
  const int a = tid & 63, b = tid >> 6;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int col = b + 4 * j, i = i0 + a, m = m0 + col;
    tile[col][a] = (i < d && m < M) ? Z[(size_t)m * d + i] - (t_mean ? t_mean[i] : T(0)) : T(0);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int row = b + 4 * j, i = i0 + row, m = m0 + a;
    if (i < d && m < M) RT[(size_t)i * MP + m] = tile[a][row];
  }
}

// ---------------------------------------------------------------------------------------------
// Host side: work tables (XCD-aware block -> tile maps) and launchers
// ---------------------------------------------------------------------------------------------
static void upload_tab(mivi_ctx *c, DevBuf &b, const std::vector<int2> &t) {
  const size_t bytes = t.size() * sizeof(int2);
  if (b.bytes < bytes) {
    if (b.p) (void)hipFree(b.p);
    (void)hipMalloc(&b.p, bytes);
    b.bytes = bytes;
  }
  (void)hipMemcpy(b.p, t.data(), bytes, hipMemcpyHostToDevice);
}

// move work from the fullest to the emptiest XCD list until they differ by at most one item
static void rebalance(std::vector<std::vector<int2>> &lists) {
  while (true) {
    int mx = 0, mn = 0;
    for (int x = 1; x < 8; ++x) {
      if (lists[x].size() > lists[mx].size()) mx = x;
      if (lists[x].size() < lists[mn].size()) mn = x;
    }
    if (lists[mx].size() <= lists[mn].size() + 1) break;
    lists[mn].push_back(lists[mx].back());
    lists[mx].pop_back();
  }
}

static std::vector<int2> interleave_xcd(const std::vector<std::vector<int2>> &lists) {
  size_t L = 0;
  for (auto &l : lists) L = l.size() > L ? l.size() : L;
  std::vector<int2> tab(8 * L, make_int2(-1, -1));
  for (int x = 0; x < 8; ++x)
    for (size_t s2 = 0; s2 < lists[x].size(); ++s2) tab[s2 * 8 + x] = lists[x][s2];
  return tab;
}

// (re)build the tables for M samples per launch
static void ensure_tabs(mivi_ctx *c, int M) {
  if (c->tab_M == M && c->tabA.p) return;
  const int d = c->cfg.d, nb = (d + 31) / 32, ncb = (M + 31) / 32;
  {  // sample: pairs (p, nb-1-p), XCD x owns pairs p = x (mod 8), all column blocks
    const int npairs = (nb + 1) / 2;
    std::vector<std::vector<int2>> lists(8);
    for (int pq = 0; pq < npairs; ++pq)
      for (int cb = 0; cb < ncb; ++cb) lists[pq % 8].push_back(make_int2(pq, cb));
    auto tab = interleave_xcd(lists);
    c->nA = (int)tab.size();
    upload_tab(c, c->tabA, tab);
  }
  {  // dense target: XCD x owns row-blocks ib = x (mod 8)
    std::vector<std::vector<int2>> lists(8);
    for (int ib = 0; ib < nb; ++ib)
      for (int cb = 0; cb < ncb; ++cb) lists[ib % 8].push_back(make_int2(ib, cb));
    auto tab = interleave_xcd(lists);
    c->nD = (int)tab.size();
    upload_tab(c, c->tabD, tab);
  }
  {  // vjp: 8x8 super-blocks of the lower triangle, longest-processing-time assignment to the 8 XCDs
    std::vector<std::vector<int2>> lists(8);
    if (nb >= 16) {
      const int S = 8, ns = (nb + S - 1) / S;
      std::vector<std::vector<int2>> sbs;
      for (int sr = 0; sr < ns; ++sr)
        for (int sc = 0; sc <= sr; ++sc) {
          std::vector<int2> t;
          for (int ib = sr * S; ib < (sr + 1) * S && ib < nb; ++ib)
            for (int jb = sc * S; jb < (sc + 1) * S && jb <= ib; ++jb) t.push_back(make_int2(ib, jb));
          if (!t.empty()) sbs.push_back(t);
        }
      std::sort(sbs.begin(), sbs.end(), [](const std::vector<int2> &u, const std::vector<int2> &v) { return u.size() > v.size(); });
      for (auto &sb : sbs) {
        int best = 0;
        for (int x = 1; x < 8; ++x)
          if (lists[x].size() < lists[best].size()) best = x;
        lists[best].insert(lists[best].end(), sb.begin(), sb.end());
      }
      rebalance(lists);
    } else {
      int t = 0;
      for (int ib = 0; ib < nb; ++ib)
        for (int jb = 0; jb <= ib; ++jb) lists[(t++) % 8].push_back(make_int2(ib, jb));
    }
    auto tab = interleave_xcd(lists);
    c->nB = (int)tab.size();
    upload_tab(c, c->tabB, tab);
  }
  c->tab_M = M;
}

void prepare_tables(mivi_ctx *c, int M) {
  if (lds_path_shape_ok(c, M) && (c->target == TGT_DIAG_GAUSS || c->target == TGT_DENSE_GAUSS)) (void)lds_prepare(c, M);
  if (c->cfg.family == MIVI_FULLRANK && c->cfg.dtype == MIVI_F32) ensure_tabs(c, M);
  if (c->cfg.family == MIVI_MEANFIELD && c->cfg.dtype == MIVI_F32 && c->target == TGT_DENSE_GAUSS) ensure_tabs(c, M);
}

int eps_blocks(const mivi_ctx *c, int M) { return ((c->cfg.d + 15) / 16) * ((M + 63) / 64); }
int fr_sample_blocks(const mivi_ctx *c, int M) {
  if (c->cfg.dtype == MIVI_F32) { ensure_tabs(const_cast<mivi_ctx *>(c), M); return c->nA; }
  return ((c->cfg.d + 31) / 32) * ((M + 31) / 32);
}
int fr_dense_blocks(const mivi_ctx *c, int M) {
  if (c->cfg.dtype == MIVI_F32) { ensure_tabs(const_cast<mivi_ctx *>(c), M); return c->nD; }
  return ((c->cfg.d + 31) / 32) * ((M + 31) / 32);
}
int fr_ld_blocks(const mivi_ctx *c) { return (c->cfg.d + 31) / 32; }

template <typename T>
static SampleArgs<T> eps_args(mivi_ctx *c, const RngArgs &rng, int M, int parity) {
  SampleArgs<T> a;
  a.d = c->cfg.d;
  a.M = M;
  a.params = nullptr;
  a.rng = rng;
  a.Z = nullptr;
  a.eps = (T *)c->eps[parity].p;
  a.ld_eps = c->dP;
  a.epsT = (T *)c->epsT[parity].p;
  a.ld_epsT = c->MP;
  a.he_part = (double *)c->he_part[parity].p;
  return a;
}

struct EpsSink { SampleArgs<float> a[4]; int grid[4], n[4]; };
EpsSink *eps_sink_alloc() { return new EpsSink(); }
void eps_sink_free(EpsSink *s) { delete s; }
void eps_sink_reset(EpsSink *s) { for (int l = 0; l < 4; ++l) s->n[l] = 0; }
// the recorded first draws of the lanes that made one: ONE launch
void launch_lanes_eps(mivi_ctx *c, EpsSink *s, int lanes) {
  EpsMulti m;
  int L = 0, grid = 0;
  for (int l = 0; l < lanes && l < 4; ++l)
    if (s->n[l] > 0) { m.lane[L++] = s->a[l]; grid = s->grid[l]; }
  if (L > 0) hipLaunchKernelGGL(k_eps_m, dim3(grid, L), dim3(512), 0, c->stream, m);
}

// returns the number of he_part entries the draw leaves
int launch_eps(mivi_ctx *c, const RngArgs &rng, int M) {
  const int nblk = eps_blocks(c, M);
  if (c->eps_sink && c->cfg.dtype == MIVI_F32) {   // lane-batched contexts: record (every lane has the same shape, so the same grid)
    EpsSink *sk = (EpsSink *)c->eps_sink;        // (k_eps_m works in the riders' blocks: d % 64 == 0, M % 32 == 0 on this route)
    const int nrid = (c->cfg.d / 64) * (M / 32);
    sk->a[c->lane_id] = eps_args<float>(c, rng, M, c->cur);
    sk->grid[c->lane_id] = nrid;
    ++sk->n[c->lane_id];
    return nrid;
  }
  if (c->cfg.dtype == MIVI_F32)
    hipLaunchKernelGGL(k_eps<float>, dim3(nblk), dim3(256), 0, c->stream, eps_args<float>(c, rng, M, c->cur));
  else
    hipLaunchKernelGGL(k_eps<double>, dim3(nblk), dim3(256), 0, c->stream, eps_args<double>(c, rng, M, c->cur));
  return nblk;
}

template <typename T>
static FrArgs<T> fr_args(mivi_ctx *c, const void *params, int M) {
  FrArgs<T> a;
  a.d = c->cfg.d;
  a.M = M;
  a.dP = c->dP;
  a.MP = c->MP;
  a.params = (const T *)params;
  a.eps = (const T *)c->eps[c->cur].p;
  a.epsT = (const T *)c->epsT[c->cur].p;
  a.Z = (T *)c->Z.p;
  a.W = (T *)c->W.p;
  a.RT = (T *)c->RT.p;
  a.fused_target = TGT_NONE;
  a.t_mean = (const T *)c->t_mean.p;
  a.t_istd = (const T *)c->t_istd.p;
  a.t_prec = (const T *)c->t_prec.p;
  a.ell_part = (double *)c->ell_part[c->cur].p;
  a.ld_part = (double *)c->ld_part[c->cur].p;
  a.vin = ValueIn{};
  a.out = OutArgs{};
  a.dbg = c->dbg;
  a.work_tab = nullptr;
  a.n_work = 0;
  a.n_pre = 0;
  a.prev_vin = ValueIn{};
  a.prev_out = OutArgs{};
  a.next_eps = SampleArgs<T>{};
  a.upd = FusedUpdate{};
  a.adam_cc = nullptr;
  return a;
}

// MIVI_F64_VALU=1 keeps the f64 tiles on the vector ALU (the in-library cross-check of the f64 MFMA kernel)
bool f64_valu() {
  static const bool v = getenv("MIVI_F64_VALU") != nullptr;
  return v;
}

// Z = mu + C eps (+ fused target).  prev != nullptr: one extra workgroup assembles the PREVIOUS estimate's value.
void launch_fr_sample(mivi_ctx *c, const void *params, int M, int fused_target, void *Z, const ValueJob *prev) {
  if (c->cfg.dtype == MIVI_F32) {
    ensure_tabs(c, M);
    FrArgs<float> a = fr_args<float>(c, params, M);
    a.fused_target = fused_target;
    a.Z = (float *)Z;
    a.work_tab = (const int2 *)c->tabA.p;
    a.n_work = c->nA;
    int grid = c->nA;
    if (prev) {
      a.prev_vin = prev->vin;
      a.prev_out = prev->out;
      grid += 1;
    } else {
      a.n_work = 0x7fffffff;   // no value workgroup
    }
    if (c->cfg.d % 32 == 0 && M % 32 == 0)   // (eight waves per tile; sixteen measured slower in round 2: the switch and its instantiation are gone)
      hipLaunchKernelGGL((k_fr_tile_mfma<MODE_SAMPLE, 8, true>), dim3(grid), dim3(512), 0, c->stream, a);
    else
      hipLaunchKernelGGL((k_fr_tile_mfma<MODE_SAMPLE, 8, false>), dim3(grid), dim3(512), 0, c->stream, a);
  } else {
    const int nblk = ((c->cfg.d + 31) / 32) * ((M + 31) / 32);
    FrArgs<double> a = fr_args<double>(c, params, M);
    a.fused_target = fused_target;
    a.Z = (double *)Z;
    a.n_work = 0x7fffffff;
    if (f64_valu()) {
      hipLaunchKernelGGL((k_fr_tile_generic<double, MODE_SAMPLE>), dim3(nblk), dim3(256), 0, c->stream, a);
    } else {
      int grid = nblk;
      if (prev) {   // one trailing workgroup assembles the previous estimate's value
        a.n_work = nblk;
        a.prev_vin = prev->vin;
        a.prev_out = prev->out;
        grid += 1;
      }
      hipLaunchKernelGGL((k_fr_tile_mfma64<MODE_SAMPLE, 8>), dim3(grid), dim3(512), 0, c->stream, a);
    }
  }
}

void launch_fr_dense_target(mivi_ctx *c, int M, int want_grad) {
  (void)want_grad;
  if (c->cfg.dtype == MIVI_F32) {
    ensure_tabs(c, M);
    FrArgs<float> a = fr_args<float>(c, nullptr, M);
    a.work_tab = (const int2 *)c->tabD.p;
    a.n_work = 0x7fffffff;
    if (c->cfg.d % 32 == 0 && M % 32 == 0)
      hipLaunchKernelGGL((k_fr_tile_mfma<MODE_DENSE, 8, true>), dim3(c->nD), dim3(512), 0, c->stream, a);
    else
      hipLaunchKernelGGL((k_fr_tile_mfma<MODE_DENSE, 8, false>), dim3(c->nD), dim3(512), 0, c->stream, a);
  } else {
    const int nblk = ((c->cfg.d + 31) / 32) * ((M + 31) / 32);
    FrArgs<double> a = fr_args<double>(c, nullptr, M);
    a.n_work = 0x7fffffff;
    if (f64_valu())
      hipLaunchKernelGGL((k_fr_tile_generic<double, MODE_DENSE>), dim3(nblk), dim3(256), 0, c->stream, a);
    else
      hipLaunchKernelGGL((k_fr_tile_mfma64<MODE_DENSE, 8>), dim3(nblk), dim3(512), 0, c->stream, a);
  }
}

// tril(W eps^T) (+ d/dmu, log-det partials).  next != nullptr: extra leading workgroups generate eps of the NEXT estimate.
void launch_fr_vjp(mivi_ctx *c, const void *params, int M, const OutArgs &out, const EpsJob *next, const ValueJob *self,
                   const FusedUpdate *upd) {
  if (c->cfg.dtype == MIVI_F32) {
    ensure_tabs(c, M);
    FrArgs<float> a = fr_args<float>(c, params, M);
    a.out = out;
    if (upd) a.upd = *upd;
    a.work_tab = (const int2 *)c->tabB.p;
    a.n_work = 0x7fffffff;
    int grid = c->nB;
    if (self) {   // one trailing workgroup assembles this estimate's value (or its two scalar partials)
      a.n_work = c->nB;
      a.prev_vin = self->vin;
      a.prev_out = self->out;
      grid += 1;
    }
    if (next) {
      a.n_pre = (eps_blocks(c, M) + 7) / 8 * 8;     // keep (blockIdx - n_pre) % 8 == blockIdx % 8
      a.next_eps = eps_args<float>(c, next->rng, M, next->parity);
      grid += a.n_pre;
    }
    if (c->cfg.d % 32 == 0 && M % 32 == 0 && upd)   // (four waves per VJP tile: the two- and eight-wave variants and their switch are gone)
      hipLaunchKernelGGL((k_fr_tile_mfma<MODE_VJP, 4, true, true>), dim3(grid), dim3(256), 0, c->stream, a);
    else if (upd)
      hipLaunchKernelGGL((k_fr_tile_mfma<MODE_VJP, 4, false, true>), dim3(grid), dim3(256), 0, c->stream, a);
    else if (c->cfg.d % 32 == 0 && M % 32 == 0)
      hipLaunchKernelGGL((k_fr_tile_mfma<MODE_VJP, 4, true>), dim3(grid), dim3(256), 0, c->stream, a);
    else
      hipLaunchKernelGGL((k_fr_tile_mfma<MODE_VJP, 4, false>), dim3(grid), dim3(256), 0, c->stream, a);
  } else {
    const int nb = (c->cfg.d + 31) / 32;
    FrArgs<double> a = fr_args<double>(c, params, M);
    a.out = out;
    const int ntile = nb * (nb + 1) / 2;
    a.n_work = 0x7fffffff;
    if (f64_valu()) {
      hipLaunchKernelGGL((k_fr_tile_generic<double, MODE_VJP>), dim3(ntile), dim3(256), 0, c->stream, a);
    } else {
      int grid = ntile;
      if (self) {
        a.n_work = ntile;
        a.prev_vin = self->vin;
        a.prev_out = self->out;
        grid += 1;
      }
      if (next) {
        a.n_pre = eps_blocks(c, M);
        a.next_eps = eps_args<double>(c, next->rng, M, next->parity);
        grid += a.n_pre;
      }
      hipLaunchKernelGGL((k_fr_tile_mfma64<MODE_VJP, 8>), dim3(grid), dim3(512), 0, c->stream, a);
    }
  }
}

void launch_rt_from_z(mivi_ctx *c, int M) {
  dim3 grid((c->cfg.d + 63) / 64, (M + 63) / 64);
  const void *tm = c->target == TGT_DENSE_GAUSS ? c->t_mean.p : nullptr;   // LogReg: plain transpose
  if (c->cfg.dtype == MIVI_F32)
    hipLaunchKernelGGL(k_rt_from_z<float>, grid, dim3(256), 0, c->stream, c->cfg.d, M, c->MP, (const float *)c->Z.p,
                       (const float *)tm, (float *)c->RT.p);
  else
    hipLaunchKernelGGL(k_rt_from_z<double>, grid, dim3(256), 0, c->stream, c->cfg.d, M, c->MP, (const double *)c->Z.p,
                       (const double *)tm, (double *)c->RT.p);
}

// ---------------------------------------------------------------------------------------------
// Stein / Price estimator of E_q[hess log pi] (src/algorithms/gauss_expected_grad_hess.jl:32-60), accumulation stage:
//   A (+)= eps G^T   -- the whole d x d product, not only the lower triangle the ELBO gradient needs
//   gsum (+)= G 1
// One workgroup per 32x32 tile of A, the sample axis dealt to its four waves (16x16x4 MFMA in T, so f32 and f64 share
// the kernel); `scale` = 1/n on the last chunk.  The solve  hess = C^-T A  reuses the STL back-substitution kernels.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_stein_outer(int d, int dP, int M, const T *eps, const T *G, T *A, double *gsum,
                                                     int first, T scale) {
  typedef typename Mfma16<T>::acc_t acc_t;
  __shared__ T red[4][32 * 33];
  __shared__ double gred[8][32];
  const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, kq = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
  acc_t acc00 = {0, 0, 0, 0}, acc01 = {0, 0, 0, 0}, acc10 = {0, 0, 0, 0}, acc11 = {0, 0, 0, 0};
  const bool okA = j0 + c16 < d, okB = j0 + 16 + c16 < d;
  const T *e0 = eps + i0 + c16;                           // rows < dP: eps is zero padded
  const T *g0 = G + (okA ? j0 + c16 : 0), *g1 = G + (okB ? j0 + 16 + c16 : 0);
  int k0 = 4 * w;
  for (; k0 + 48 + 4 <= M; k0 += 64) {                    // four 4-sample steps of this wave, all in range: loads first
    T a0[4], a1[4], b0[4], b1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t kk = (size_t)(k0 + 16 * u + kq);
      a0[u] = e0[kk * dP];
      a1[u] = e0[kk * dP + 16];
      b0[u] = okA ? g0[kk * d] : T(0);
      b1[u] = okB ? g1[kk * d] : T(0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc00 = Mfma16<T>::mma(a0[u], b0[u], acc00);
      acc01 = Mfma16<T>::mma(a0[u], b1[u], acc01);
      acc10 = Mfma16<T>::mma(a1[u], b0[u], acc10);
      acc11 = Mfma16<T>::mma(a1[u], b1[u], acc11);
    }
  }
  for (; k0 < M; k0 += 16) {
    const int k = k0 + kq;
    const bool ok = k < M;
    const size_t kk = ok ? (size_t)k : 0;
    const T a0 = ok ? e0[kk * dP] : T(0), a1 = ok ? e0[kk * dP + 16] : T(0);
    const T b0 = (ok && okA) ? g0[kk * d] : T(0), b1 = (ok && okB) ? g1[kk * d] : T(0);
    acc00 = Mfma16<T>::mma(a0, b0, acc00);
    acc01 = Mfma16<T>::mma(a0, b1, acc01);
    acc10 = Mfma16<T>::mma(a1, b0, acc10);
    acc11 = Mfma16<T>::mma(a1, b1, acc11);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = Mfma16<T>::row(kq, r);
    red[w][row * 33 + c16] = acc00[r];
    red[w][row * 33 + 16 + c16] = acc01[r];
    red[w][(16 + row) * 33 + c16] = acc10[r];
    red[w][(16 + row) * 33 + 16 + c16] = acc11[r];
  }
  if (blockIdx.x == 0) {                                   // column sums of G for this block of coordinates
    const int jj = tid & 31, part = tid >> 5;
    double sacc = 0.0;
    if (j0 + jj < d)
      for (int b = part; b < M; b += 8) sacc += (double)G[(size_t)b * d + j0 + jj];
    gred[part][jj] = sacc;
  }
  __syncthreads();
  for (int t = tid; t < 1024; t += 256) {
    const int row = t & 31, col = t >> 5;
    if (j0 + col >= d) continue;
    const T sum = (red[0][row * 33 + col] + red[1][row * 33 + col]) + (red[2][row * 33 + col] + red[3][row * 33 + col]);
    T *dst = A + (size_t)(j0 + col) * dP + i0 + row;
    *dst = ((first ? T(0) : *dst) + sum) * scale;
  }
  if (blockIdx.x == 0 && tid < 32 && j0 + tid < d) {
    double sacc = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) sacc += gred[q][tid];
    gsum[j0 + tid] = (first ? 0.0 : gsum[j0 + tid]) + sacc;
  }
}

template <typename T>
__global__ void k_stein_finish(int d, double n, const double *gsum, const double *ell_sum, const T *ell_single, T *grad, T *logpi_avg) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < d) grad[i] = (T)(gsum[i] / n);
  if (i == 0) *logpi_avg = (T)((ell_single ? (double)ell_single[0] : ell_sum[0]) / n);   // one chunk: straight from its partial
}

// second-order branch (src/algorithms/gauss_expected_grad_hess.jl:61-83): only the column sums of G are needed -- gsum (+)= G 1,
// one workgroup per 32 coordinates, the sample axis dealt to eight 32-lane groups (f64 sums, fixed order)
template <typename T>
__global__ __launch_bounds__(256) void k_stein_gsum(int d, int M, const T *G, double *gsum, int first) {
  __shared__ double gred[8][32];
  const int tid = threadIdx.x, jj = tid & 31, part = tid >> 5, j0 = blockIdx.x * 32;
  double sacc = 0.0;
  if (j0 + jj < d)
    for (int b = part; b < M; b += 8) sacc += (double)G[(size_t)b * d + j0 + jj];
  gred[part][jj] = sacc;
  __syncthreads();
  if (tid < 32 && j0 + tid < d) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += gred[q][tid];
    gsum[j0 + tid] = (first ? 0.0 : gsum[j0 + tid]) + t;
  }
}
void launch_stein_gsum(mivi_ctx *c, int M, double *gsum, int first) {
  const dim3 grid((c->cfg.d + 31) / 32);
  if (c->cfg.dtype == MIVI_F32) hipLaunchKernelGGL(k_stein_gsum<float>, grid, dim3(256), 0, c->stream, c->cfg.d, M, (const float *)c->W.p, gsum, first);
  else hipLaunchKernelGGL(k_stein_gsum<double>, grid, dim3(256), 0, c->stream, c->cfg.d, M, (const double *)c->W.p, gsum, first);
}
// the built-in Gaussian targets' Hessian, a constant: diagonal N(m, diag(sigma^2)): -diag(1 / sigma^2); dense N(m, L L'): -P, P = (L L')^-1
template <typename T>
__global__ void k_const_hess(int d, int ldp, const T *istd, const T *prec, T *hess) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)d * d) return;
  const int i = (int)(e % d), j = (int)(e / d);
  hess[e] = prec ? -prec[(size_t)j * ldp + i] : (i == j ? -(istd[i] * istd[i]) : T(0));
}
void launch_const_hess(mivi_ctx *c, void *hess) {
  const size_t n = (size_t)c->cfg.d * c->cfg.d;
  const dim3 grid((unsigned)((n + 255) / 256));
  const bool dense = c->target == TGT_DENSE_GAUSS;
  if (c->cfg.dtype == MIVI_F32)
    hipLaunchKernelGGL(k_const_hess<float>, grid, dim3(256), 0, c->stream, c->cfg.d, c->dP, (const float *)c->t_istd.p, dense ? (const float *)c->t_prec.p : nullptr, (float *)hess);
  else
    hipLaunchKernelGGL(k_const_hess<double>, grid, dim3(256), 0, c->stream, c->cfg.d, c->dP, (const double *)c->t_istd.p, dense ? (const double *)c->t_prec.p : nullptr, (double *)hess);
}

void launch_stein_outer(mivi_ctx *c, int M, void *A, double *gsum, int first, double scale) {
  dim3 grid(c->dP / 32, (c->cfg.d + 31) / 32);
  if (c->cfg.dtype == MIVI_F32)
    hipLaunchKernelGGL(k_stein_outer<float>, grid, dim3(256), 0, c->stream, c->cfg.d, c->dP, M, (const float *)c->eps[c->cur].p,
                       (const float *)c->W.p, (float *)A, gsum, first, (float)scale);
  else
    hipLaunchKernelGGL(k_stein_outer<double>, grid, dim3(256), 0, c->stream, c->cfg.d, c->dP, M, (const double *)c->eps[c->cur].p,
                       (const double *)c->W.p, (double *)A, gsum, first, scale);
}

void launch_stein_finish(mivi_ctx *c, double n, const double *gsum, const double *ell_sum, const void *ell_single, void *grad, void *logpi_avg) {
  const int nb = (c->cfg.d + 255) / 256;
  if (c->cfg.dtype == MIVI_F32)
    hipLaunchKernelGGL(k_stein_finish<float>, dim3(nb), dim3(256), 0, c->stream, c->cfg.d, n, gsum, ell_sum, (const float *)ell_single, (float *)grad, (float *)logpi_avg);
  else
    hipLaunchKernelGGL(k_stein_finish<double>, dim3(nb), dim3(256), 0, c->stream, c->cfg.d, n, gsum, ell_sum, (const double *)ell_single, (double *)grad, (double *)logpi_avg);
}

// W += C^-T eps for the M sample columns -- or, with `rhs` / `out` given, out += C^-T rhs for any M-column right-hand
// side laid out like eps (leading dimension dP, zero padded) and any output laid out like W (leading dimension d)
void launch_fr_stl(mivi_ctx *c, const void *params, int M, const void *rhs, void *out) {
  const int nblk = (M + 7) / 8;
  static const bool old_stl = getenv("MIVI_STL_VALU") != nullptr;
  const size_t sh_mfma16 = ((size_t)c->dP * 16 + 8 * 8 * 64) * sizeof(float);   // 16-column solve: d up to 2304
  const size_t sh16_f64 = ((size_t)c->dP * 16 + 8 * 8 * 64) * sizeof(double);   // f64: d up to 1024
  if (c->cfg.dtype == MIVI_F64 && c->stl_CT.p && sh16_f64 <= 160 * 1024 && !old_stl && !f64_valu()) {
    FrArgs<double> a = fr_args<double>(c, params, M);
    if (rhs) a.eps = (const double *)rhs;
    if (out) a.W = (double *)out;
    const int nb = (c->cfg.d + 31) / 32;
    hipLaunchKernelGGL(k_stl_prep<double>, dim3(nb + nb * (nb + 1) / 2), dim3(256), 0, c->stream, c->cfg.d, c->dP,
                       (const double *)params + c->cfg.d, (double *)c->stl_CT.p, (double *)c->stl_Dinv.p);
    static size_t attr64 = 0;
    if (attr64 < sh16_f64) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stl_solve_la16<double, 8>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh16_f64);
      attr64 = sh16_f64;
    }
    hipLaunchKernelGGL((k_stl_solve_la16<double, 8>), dim3((M + 15) / 16), dim3(512), sh16_f64, c->stream, a,
                       (const double *)c->stl_CT.p, (const double *)c->stl_Dinv.p);
    return;
  }
  if (c->cfg.dtype == MIVI_F32 && c->stl_CT.p && sh_mfma16 <= 160 * 1024 && !old_stl) {
    FrArgs<float> a = fr_args<float>(c, params, M);
    if (rhs) a.eps = (const float *)rhs;
    if (out) a.W = (float *)out;
    const int nb = (c->cfg.d + 31) / 32;
    hipLaunchKernelGGL(k_stl_prep<float>, dim3(nb + nb * (nb + 1) / 2), dim3(256), 0, c->stream, c->cfg.d, c->dP,
                       (const float *)params + c->cfg.d, (float *)c->stl_CT.p, (float *)c->stl_Dinv.p);
    static size_t attr16 = 0;   // raise the dynamic-LDS cap once per size (the call is slow)
    if (attr16 < sh_mfma16) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stl_solve_la16<float, 8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)sh_mfma16);
      attr16 = sh_mfma16;
    }
    hipLaunchKernelGGL((k_stl_solve_la16<float, 8>), dim3((M + 15) / 16), dim3(512), sh_mfma16, c->stream, a, (const float *)c->stl_CT.p,
                       (const float *)c->stl_Dinv.p);
    return;
  }
  if (c->cfg.dtype == MIVI_F32) {
    FrArgs<float> a = fr_args<float>(c, params, M);
    if (rhs) a.eps = (const float *)rhs;
    if (out) a.W = (float *)out;
    const size_t sh = (8 * (size_t)c->dP + 32 * 33) * sizeof(float);
    static size_t attr_f = 0;
    if (attr_f < sh) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fr_stl<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      attr_f = sh;
    }
    hipLaunchKernelGGL(k_fr_stl<float>, dim3(nblk), dim3(256), sh, c->stream, a);
  } else {
    FrArgs<double> a = fr_args<double>(c, params, M);
    if (rhs) a.eps = (const double *)rhs;
    if (out) a.W = (double *)out;
    const size_t sh = (8 * (size_t)c->dP + 32 * 33) * sizeof(double);
    static size_t attr_d = 0;
    if (attr_d < sh) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fr_stl<double>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
      attr_d = sh;
    }
    hipLaunchKernelGGL(k_fr_stl<double>, dim3(nblk), dim3(256), sh, c->stream, a);
  }
}

}  // namespace mivi
