// Shared by the second-generation full-rank kernels (kernels_fullrank_lds.hip) and their operand-plane variants
// (kernels_fullrank_planes.hip): argument blocks and the VJP epilogue (final / shard / fused-update modes).
#pragma once
#include "device_common.h"
#include "fr_elem.h"
#include "optim_rules.h"

namespace mivi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { G_SAMPLE = 0, G_VJP = 1, G_DENSE = 2 };

struct GemmArgs {
  int d, M, dP;
  const float *A;   // MN-major operand A[row + k*lda]: tril(C) (sample), W (vjp), P (dense)
  int lda;
  const float *B;   // sample / dense: k-major B[k + n*ldb] (eps, Z - m); vjp: B[n + k*ldb] (eps)
  int ldb;
  const int4 *work;  // .x = rb | cb << 16, .y = first stage | end stage << 16, .z unused, .w = flags
  int n_work;        // the workgroup after the last item assembles the objective value (vjp, optional)
  // vjp epilogue
  const float *params;
  OutArgs out;
  FusedUpdate upd;
  ValueIn self_vin;
  OutArgs self_out;
  long long *dbg;    // optional timeline (tools/timeline2.py)
  int knock;         // developer knock-outs (MIVI_KNOCK): 1 no loads after the prologue, 2 no MFMAs, 4 no loads at all
  // Stein mode of k_fr_vjp64 (the whole product eps G^T, A = eps, B = G): see stein_epilogue
  float *st_A;       // d x d accumulation target, element (i, j) at st_A[j * st_ld + i]
  int st_ld;
  double *st_gsum;   // [d] column sums of G (accumulated over chunks)
  float *st_grad;    // single chunk: grad = gsum / n written directly (else nullptr)
  float *st_logpi;   // single chunk: mean log-density, by the value workgroup (else nullptr)
  int st_first;      // first chunk: overwrite instead of accumulate
  float st_scale;    // 1 / n on the last chunk
  double st_n;
};

// -----------------------------------------------------------------------------------------------------------------
// Epilogue of the VJP kernels: tile (row0, col0) of tril(W eps^T), KW partial images Cs[kw][n][LDC] in LDS (rows contiguous),
// rs_lds[NT/BM][BM] partial row sums of W.  Final mode: -1/M scaling, entropy diagonal term, exact zeros above the diagonal
// (and in the mirrored tile), d/dmu on the tiles that hold a diagonal block's first columns; shard mode: packed triangle;
// FUSED: Descent / Adam (+ ClipScale) applied in place instead of writing the gradient.
// -----------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int KW, int NT, bool FUSED>
__device__ __forceinline__ void vjp_epilogue(const GemmArgs &a, const float *Cs, const float *rs_lds, const float *adam_cc, int4 wk,
                                             int row0, int col0, float *pimg = nullptr) {   // pimg (fused update only): LDS image [k][row], leading dimension LDC, of the tile's UPDATED parameters (kernels_fullrank_planes.hip)
  constexpr int LDC = BM + 4;
  constexpr int NE = BM * BN / 4 / NT;
  const int tid = threadIdx.x;
  const int d = a.d;
  const bool mu_tile = (wk.w & 2);
  // ---- vjp epilogue: tile (rb, cb) of tril(W eps^T) -------------------------------------------------------------------
  const double invM = 1.0 / (double)a.out.M_total;
  const bool pow2M = (a.out.M_total & (a.out.M_total - 1)) == 0;
  const float invMf = (float)invM;
  const double direct = direct_entropy_coeff(a.out.ent_kind);
  const bool diag_tile = (wk.w & 1);
  if (a.out.partials_mode) {   // shard partials: raw sums, packed lower triangle
    const PartialDst pd = partial_dst(a.out);
    float *dst = (float *)a.out.partials;
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int e = tid + u * NT, i4 = 4 * (e % (BM / 4)), n = e / (BM / 4);
      f32x4 v = *(const f32x4 *)(Cs + n * LDC + i4);
#pragma unroll
      for (int k2 = 1; k2 < KW; ++k2) v += *(const f32x4 *)(Cs + (k2 * BN + n) * LDC + i4);
      const int gj = col0 + n;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int gi = row0 + i4 + c;
        if (gj <= gi) {
          const size_t pi = d + (size_t)gj * d - ((size_t)gj * (gj - 1)) / 2 + (gi - gj);
          if (pd.tab) partial_store(pd, (long long)pi, v[c]);   // (peer-to-peer route, direct: into the owner's staging area)
          else dst[pi] = v[c];
        }
      }
    }
  } else {
    const bool fused = FUSED && a.upd.rule >= 0;
    float *dst = (float *)a.out.grad;
    const size_t plen = (size_t)d + (size_t)d * d;
#pragma unroll
    for (int u = 0; u < NE; ++u) {
      const int e = tid + u * NT, i4 = 4 * (e % (BM / 4)), n = e / (BM / 4);
      const int gi = row0 + i4, gj = col0 + n;
      const size_t pi = d + (size_t)gj * d + gi;
      f32x4 px, pm, pv;
      if (fused) {   // parameters (and Adam moments) of this thread's elements: one round trip, issued before the LDS reads
        px = *(const f32x4 *)((const float *)a.upd.params + pi);
        if (a.upd.rule == 1) {
          pm = *(const f32x4 *)((const float *)a.upd.state + pi);
          pv = *(const f32x4 *)((const float *)a.upd.state + plen + pi);
        }
      }
      float cjj = 1.f;
      if (diag_tile && gj >= gi && gj < gi + 4) cjj = a.params[d + (size_t)gj * d + gj];
      f32x4 v = *(const f32x4 *)(Cs + n * LDC + i4);
#pragma unroll
      for (int k2 = 1; k2 < KW; ++k2) v += *(const f32x4 *)(Cs + (k2 * BN + n) * LDC + i4);
      f32x4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] = vjp_elem(v[c], gi + c, gj, pow2M, invMf, invM, direct, cjj);   // (fr_elem.h: shared with the batch kernels)
      if (!fused) {
        if (!MIVI_KNOCKED(a, 256)) store16_wt(dst + pi, o);
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (gj > gi + c) continue;   // zero gradients above the diagonal move nothing
          float x;
          if (a.upd.rule == 0) {
            x = descent_step(px[c], o[c], (float)a.upd.eta);
          } else {
            float m = pm[c], vv = pv[c];
            x = adam_step<float>(px[c], o[c], m, vv, adam_cc[0], adam_cc[1], (float)a.upd.eta, (float)a.upd.b1,
                                 (float)a.upd.b2, (float)a.upd.eps);
            pm[c] = m;
            pv[c] = vv;
          }
          if (gj == gi + c && a.upd.do_clip) x = clip_step(x, (float)a.upd.clip_eps);
          px[c] = x;
        }
        store16_wt((float *)a.upd.params + pi, px);   // (written through: see store16_wt)
        if (pimg) *(f32x4 *)(pimg + n * LDC + i4) = px;
        if (a.upd.rule == 1) {
          store16_wt((float *)a.upd.state + pi, pm);
          store16_wt((float *)a.upd.state + plen + pi, pv);
        }
      }
    }
    if (!fused && !diag_tile && !MIVI_KNOCKED(a, 128)) {   // the mirrored, strictly upper tile is structurally zero
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < NE; ++u) {
        const int e = tid + u * NT, j4 = 4 * (e % (BN / 4)), ii = e / (BN / 4);
        store16_wt(dst + d + (size_t)(row0 + ii) * d + col0 + j4, z4);
      }
    }
  }
  if (mu_tile && tid < BM) {   // d/dmu rows of this row block: sum of W over all samples
    double sm = 0.0;
#pragma unroll
    for (int g = 0; g < NT / BM; ++g) sm += (double)rs_lds[g * BM + tid];
    const int gr = row0 + tid;
    if (a.out.partials_mode) {
      if (a.out.p2p_direct) partial_store(partial_dst(a.out), gr, (float)sm);
      else ((float *)a.out.partials)[gr] = (float)sm;
    } else {
      const float g = dmu_elem(sm, invM);
      if (FUSED && a.upd.rule >= 0) {
        float *pp = (float *)a.upd.params;
        float x;
        if (a.upd.rule == 0) {
          x = descent_step(pp[gr], g, (float)a.upd.eta);
        } else {
          float *st = (float *)a.upd.state;
          const size_t plen = (size_t)d + (size_t)d * d;
          float m = st[gr], vv = st[plen + gr];
          x = adam_step<float>(pp[gr], g, m, vv, adam_cc[0], adam_cc[1], (float)a.upd.eta, (float)a.upd.b1, (float)a.upd.b2,
                               (float)a.upd.eps);
          st[gr] = m;
          st[plen + gr] = vv;
        }
        pp[gr] = x;
      } else {
        ((float *)a.out.grad)[gr] = g;
      }
    }
  }
  MIVI_STAMP_K(a.dbg, G_VJP, 4);
}

// arguments of k_fr_prod32 (and, with the operand planes beside them, of k_fr_prod32p)
struct Prod32Args {
  int d, M, dP, mode;
  const float *A;    // tril(C) (lda = d) or P (lda = dP), row-major operand A[row + k*lda]
  int lda;
  const float *B;    // eps or Z - m, k-major B[k + n*dP]
  const float *params;
  const float *t_mean, *t_istd;
  float *Z, *W, *R;
  double *ell_part;  // one per tile workgroup
  double *ld_part;   // [2][d/32] or nullptr
  int n_tiles;       // blocks in [n_tiles, n_tiles + n_eps) draw eps of the next estimate
  int n_eps;
  int n_dinv;        // STL riders (parameters only, off the critical path; stl_dinv.h): the FIRST n_dinv blocks invert the 64x64
  int n_pack;        // diagonal blocks of C (a long latency chain: started first), the LAST n_pack re-lay its off-diagonal blocks
  unsigned *stl_pack;
  int ncb;
  SampleArgs<float> next_eps;
  long long *dbg;
  int knock;
};

}  // namespace mivi
