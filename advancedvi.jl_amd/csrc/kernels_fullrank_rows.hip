// Launch-free optimisation loop for the full-rank family with FEW samples per step (n_mc <= 32; the reference's default is n_samples = 1,
// src/algorithms/klminrepgraddescent.jl; d <= 1024) and the diagonal-Gaussian target, f32.
//
// Reference semantics per iteration (src/algorithms/common.jl:69-104): estimate_gradient! (src/algorithms/repgradelbo.jl:151-177) with
//   z = mu + tril(C) eps (src/families/location_scale.jl:71-77),  W = grad log pi(z),
//   d/dmu = -(1/M) W 1,  d/dC = -(1/M) tril(W eps') - direct diag(1 / C_ii)            
// then Optimisers.update! (Descent / Adam) and ClipScale.
//
// With few samples a step of the general route is two launches of tile kernels built for n_mc >= 128 (first generation below that): 13-20 us
// per step at d = 256 .. 1024, almost all of it launch latency and parameter traffic.  But for a target whose gradient is elementwise in z
// (the diagonal Gaussian) ROW i of the problem -- C[i, 0..i], mu_i, their optimiser state -- needs nothing from any other row: z_i is a dot
// product of row i with eps, W_i a function of z_i, and d/dC[i, k] = -(1/M) sum_m W_im eps_km.  The only thing the rows share is eps.  So, as
// in the mean-field loop (k_mf_sgd_loop), a workgroup can OWN rows for all n_steps with their parameters and Adam moments in registers:
//   * eps of ALL steps of the call is drawn first by one launch (k_eps_steps: [step][k][m], the same Philox stream as every other route);
//   * workgroup b owns the row pairs (2b, d-1-2b) and (2b+1, d-2-2b): every pair holds d + 1 entries, so all workgroups carry the same work
//     (256 of them at d = 1024, one per CU); 128 threads per pair, the first ceil((p+1)/9) of them nine consecutive entries of row p each,
//     the others nine of row q: a thread works for ONE row (one set of per-sample accumulators);
//   * a step: the step's eps slab (fresh addresses every step: plain loads, L2-shared by the workgroups of an XCD) is copied into LDS; partial
//     dot products of the thread's entries for every sample, wave sums + one LDS exchange per pair -> z, W, ell; the gradient entries from W
//     and the slab; update + ClipScale in registers.  Three barriers per step, no grid-wide synchronisation at all.  The slab's row stride is
//     padded to 4 x odd words (n_mc 8 -> 12, 16 -> 20, 32 -> 36) so that the 16-byte LDS reads of lanes nine rows apart fall on distinct
//     banks (unpadded: 26 us per step at n_mc = 16, 87 us at 32, against 20 us for the launches);
//   * per-step partials {sum ell, sum 0.5 eps^2, sum log C_ii, #non-positive C_ii} per workgroup; elbo[t] assembled afterwards
//     (k_fr_rows_value).
// Sums are sequential f32 fused multiply-adds in a fixed order (deterministic); a trajectory equals the launch-per-step one to rounding
// (tests/test_gpu_optimize.py::test_fullrank_rows_loop).  Not for the sticking-the-landing estimators (C^-T eps couples the rows) and not
// for other targets; MIVI_NO_FUSED_LOOP=1 keeps the graph of launches.
#include <cstdlib>
#include <type_traits>

#include "device_common.h"
#include "optim_rules.h"

namespace mivi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct FrRowsArgs {
  int d, M, Mp, nch, n_steps, rule, ent_kind, m_offset, M_total;
  float *params, *opt_state;       // [mu; vec C column-major]; Adam: [m (d + d^2); v (d + d^2)]
  const float *t_mean, *t_istd;
  const float *eps_all;            // [n_steps][d][Mp], columns M .. Mp - 1 zero
  long long t0;
  double eta, clip_eps, b1, b2, adam_eps;
  double *hist;                    // [n_steps][4][n_wg]
  // the rules / operators / averager beyond Descent / Adam + ClipScale (k_mf_gen_loop's set: the reference's defaults are DoWG + averaging)
  int op, averager;                // op 0 identity, 1 ClipScale, 2 ProximalLocationScaleEntropy; averager 1: PolynomialAveraging
  double avg_eta;
  float *avg;                      // running average [mu; vec C], in / out
  const float *x0;                 // DoG / DoWG: the initial parameters
  double *dog_sc;                  // DoG / DoWG: (v, r), in / out
  double *part;                    // DoG / DoWG: [n_steps][n_wg][2] partial norms, NaN until delivered
  int *status;
  int spin;
};

constexpr int kRowsEPT = 9;        // entries of a row per thread (128 threads per pair of rows: d + 1 <= 9 * 126)
constexpr int kRowsNT = 256;       // two pairs of rows per workgroup, two waves each (three waves -- one for row p, two for row q, no wave
                                   // working for both rows -- measured slower: 8.4 against 7.7 us per step at d = 1024, n_mc = 16)
constexpr int kRowsMaxM = 32;

// the slab's sample columns: 1, 2, 4 as they are; above that a multiple of 8 (the kernel's sample chunk) + 4 = 4 x odd
static int rows_mm(int M) { return M <= 2 ? M : (M <= 4 ? 4 : ((M + 7) & ~7)); }
static int rows_mp(int MM) { return MM <= 4 ? MM : MM + 4; }

// eps of n_steps estimates, [t][k][Mp]: one Philox block (rows 4 q .. 4 q + 3 of column m of estimate idx0 + t) per thread; zero pads
__global__ __launch_bounds__(256) void k_eps_steps(uint64_t seed, uint64_t idx0, int n_steps, int d, int M, int Mp, int m_offset, float *out) {
  const int d4 = (d + 3) >> 2;
  const long long n = (long long)n_steps * d4 * Mp, i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int t = (int)(i / ((long long)d4 * Mp)), r = (int)(i - (long long)t * d4 * Mp);
  const int q = r / Mp, m = r - q * Mp;   // (consecutive threads: consecutive samples of one row quad -> contiguous stores per row)
  float e[4] = {0.f, 0.f, 0.f, 0.f};
  if (m < M) eps_block<float>(seed, idx0 + (uint64_t)t, (uint64_t)(m_offset + m) * (uint64_t)d4 + (uint64_t)q, e);
  float *o = out + ((size_t)t * d + 4 * q) * Mp + m;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr)
    if (4 * q + rr < d) o[(size_t)rr * Mp] = e[rr];
}

#define ROWS_GLDS16(gptr, lptr)                                                            \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr), \
                                   (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

// RULE: 0 Descent, 1 Adam, 2 DoG, 3 DoWG (the last two: one grid-wide exchange of two norm partials per step, as in k_mf_gen_loop).  MC: samples per chunk (1, 2, 4: the whole batch; 8: a.nch chunks).  DB: two slabs fit the LDS -- the next
// step's slab arrives (LDS-DMA, no registers) while this step computes.
template <int RULE, int MC, bool DB, bool GEN>   // GEN: the proximal operator / PolynomialAveraging code and registers exist
__global__ __launch_bounds__(kRowsNT) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_fr_rows_loop(FrRowsArgs a) {
  constexpr int NT = kRowsNT, EPT = kRowsEPT, ZS = kRowsMaxM;
  extern __shared__ __attribute__((aligned(16))) float S[];   // eps slab(s) [k][Mp], then the small exchange areas
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int d = a.d, M = a.M, Mp = a.Mp, nch = MC == 8 ? a.nch : 1, nwg = gridDim.x, b = blockIdx.x;
  const int slabN = d * Mp;                   // (a multiple of 4)
  float *zred = S + (DB ? 2 : 1) * slabN;     // [4 waves][2 rows][ZS]: wave partials of the dot products
  float *sc = zred + 4 * 2 * ZS;              // [step parity][pair][row]{mu, C_rr}; [16 ..]: [pair][row] scalar partials {ell, he, lg, bad}
  float(*cc_tab)[2] = reinterpret_cast<float(*)[2]>(sc + 16 + 16);   // [256][2] Adam bias corrections
  double *gred = reinterpret_cast<double *>(sc + 16 + 16 + 512);     // [8] block sums of the two norm partials
  int *ok_s = reinterpret_cast<int *>(gred + 8);
  const int pr = tid >> 7, t7 = tid & 127;    // the pair this thread works for, its index inside the pair
  const int p = 2 * b + pr, q = d - 1 - p;    // rows p and q (one row when they meet, none beyond)
  const bool pair_ok = p <= q, two = p < q;
  const int nA = pair_ok ? (p + EPT) / EPT : 0;   // threads of row p: ceil((p + 1) / EPT)
  const bool isA = t7 < nA;
  const int row = isA ? p : q, k0 = (isA ? t7 : t7 - nA) * EPT;
  const bool row_ok = pair_ok && (isA || two);
  const int rs = isA ? 0 : 1;                 // the row's slot inside the pair
  const bool mixed = !__all(isA) && !__all(!isA);   // (wave-uniform: this wave works for both rows)
  const double direct = direct_entropy_coeff(a.ent_kind);
  const float invMf = 1.f / (float)a.M_total;
  const double invM = 1.0 / (double)a.M_total;
  const float eta = (float)a.eta, b1 = (float)a.b1, b2 = (float)a.b2, aeps = (float)a.adam_eps, ceps = (float)a.clip_eps;
  const bool clip = a.op == 1 && a.clip_eps == a.clip_eps, prox = GEN && a.op == 2, averaging = GEN && a.averager == 1;
  const size_t plen = (size_t)d + (size_t)d * d;

  // this thread's entries C[row, k0 .. k0 + 8]
  bool eok[EPT];
  int eo[EPT];                                // slab offsets of their eps rows (an entry beyond the row has px = 0 and reads row d - 1)
  float px[EPT], pm[EPT], pv[EPT];           // (DoG / DoWG: pm holds x0; pv, with PolynomialAveraging, the running average)
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    eok[e] = row_ok && k0 + e <= row;
    eo[e] = min(k0 + e, d - 1) * Mp;
    const size_t at = (size_t)d + (size_t)(eok[e] ? k0 + e : 0) * d + (eok[e] ? row : 0);
    px[e] = eok[e] ? a.params[at] : 0.f;
    pm[e] = (RULE == 1 && eok[e]) ? a.opt_state[at] : ((RULE >= 2 && eok[e]) ? a.x0[at] : 0.f);
    pv[e] = (RULE == 1 && eok[e]) ? a.opt_state[plen + at] : 0.f;
  }
  float pa[EPT];                              // PolynomialAveraging: the running average of the thread's entries
#pragma unroll
  for (int e = 0; e < EPT; ++e) pa[e] = (averaging && eok[e]) ? a.avg[(size_t)d + (size_t)(k0 + e) * d + row] : 0.f;
  const bool mu_own = row_ok && k0 == 0;      // the row's first thread also owns mu_row
  float mx = mu_own ? a.params[row] : 0.f, mm1 = (RULE == 1 && mu_own) ? a.opt_state[row] : ((RULE >= 2 && mu_own) ? a.x0[row] : 0.f),
        mv1 = (RULE == 1 && mu_own) ? a.opt_state[plen + row] : 0.f, ma = (averaging && mu_own) ? a.avg[row] : 0.f;
  double dog_v = 0.0, dog_r = 0.0;
  if (RULE >= 2) { dog_v = a.dog_sc[0]; dog_r = a.dog_sc[1]; }
  bool lost = false;
  const float tm = row_ok ? a.t_mean[row] : 0.f, ti = row_ok ? a.t_istd[row] : 0.f;
  const int diag_e = row_ok ? row - k0 : -1;  // the entry that is C[row, row], if this thread holds it (0 .. 8)

  // 1 KiB per wave and instruction, straight into the LDS
  auto slab_in = [&](int t, float *buf) {
    const float *G = a.eps_all + (size_t)t * slabN;
    const int nv = slabN >> 2;
    for (int i0 = wv * 64; i0 < nv; i0 += NT)
      if (i0 + lane < nv) ROWS_GLDS16(G + 4 * (i0 + lane), buf + 4 * i0);
  };
  auto publish = [&](int par) {               // this step's mu and scale diagonal of the workgroup's rows
    float *dst = sc + 8 * par + 2 * (2 * pr + rs);
    if (mu_own) dst[0] = mx;
#pragma unroll
    for (int e = 0; e < EPT; ++e)
      if (e == diag_e) dst[1] = px[e];
  };
  if (tid < 16) sc[16 + tid] = 0.f;           // (a pair beyond the middle never writes its partials)
  slab_in(0, S);
  publish(0);
  __builtin_amdgcn_s_waitcnt(0x0f70);         // vmcnt(0)
  __syncthreads();

  for (int t = 0; t < a.n_steps && !lost; ++t) {
    const float *cur = S + (DB ? (t & 1) * slabN : 0);
    if (RULE == 1 && (t & 255) == 0 && tid < 256) adam_bias<float>(a.t0 + t + tid + 1, a.b1, a.b2, cc_tab[tid][0], cc_tab[tid][1]);
    if (DB && t + 1 < a.n_steps) slab_in(t + 1, S + ((t + 1) & 1) * slabN);
    // partial dot products of this thread's entries, per sample, chunk by chunk; wave sums -> zred
#pragma unroll 1
    for (int ch = 0; ch < nch; ++ch) {
      float acc[MC];
#pragma unroll
      for (int m = 0; m < MC; ++m) acc[m] = 0.f;
      if (MC >= 4) {
        f32x4 ev[EPT][MC >= 4 ? MC / 4 : 1];               // (all of the chunk's LDS reads in flight before the first multiply-add)
#pragma unroll
        for (int e = 0; e < EPT; ++e)
#pragma unroll
          for (int j = 0; j < MC / 4; ++j) ev[e][j] = *(const f32x4 *)(cur + eo[e] + ch * 8 + 4 * j);
        f32x2 a2[MC >= 2 ? MC / 2 : 1];                    // two samples per v_pk_fma_f32; every sample's sum is the same sequential chain
#pragma unroll
        for (int j = 0; j < MC / 2; ++j) a2[j] = f32x2{0.f, 0.f};
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          const f32x2 p2 = {px[e], px[e]};
#pragma unroll
          for (int j = 0; j < MC / 4; ++j) {
            a2[2 * j + 0] = __builtin_elementwise_fma(p2, f32x2{ev[e][j].x, ev[e][j].y}, a2[2 * j + 0]);
            a2[2 * j + 1] = __builtin_elementwise_fma(p2, f32x2{ev[e][j].z, ev[e][j].w}, a2[2 * j + 1]);
          }
        }
#pragma unroll
        for (int j = 0; j < MC / 2; ++j) { acc[2 * j] = a2[j].x; acc[2 * j + 1] = a2[j].y; }
      } else {
        float ev[EPT][MC];
#pragma unroll
        for (int e = 0; e < EPT; ++e)
#pragma unroll
          for (int m = 0; m < MC; ++m) ev[e][m] = cur[eo[e] + m];
#pragma unroll
        for (int e = 0; e < EPT; ++e)
#pragma unroll
          for (int m = 0; m < MC; ++m) acc[m] = fmaf(px[e], ev[e][m], acc[m]);
      }
      // wave sums per row (independent chains: they pipeline); a wave working for both rows sums twice
      float sb[MC];
      if (mixed) {
#pragma unroll
        for (int m = 0; m < MC; ++m) sb[m] = wave_sum_f32(isA ? 0.f : acc[m]);
#pragma unroll
        for (int m = 0; m < MC; ++m) acc[m] = wave_sum_f32(isA ? acc[m] : 0.f);
      } else {
#pragma unroll
        for (int m = 0; m < MC; ++m) {
          const float sm = wave_sum_f32(acc[m]);
          acc[m] = isA ? sm : 0.f;
          sb[m] = isA ? 0.f : sm;
        }
      }
      if (lane == 0) {
#pragma unroll
        for (int m = 0; m < MC; ++m) { zred[(wv * 2 + 0) * ZS + ch * 8 + m] = acc[m]; zred[(wv * 2 + 1) * ZS + ch * 8 + m] = sb[m]; }
      }
    }
    lds_barrier();
    // z, W of the thread's row (every thread of the row: it needs W for its gradient entries), chunk by chunk, and the gradient sums
    f32x2 gv2[EPT];                          // gradient sums: even / odd samples of a pair (one v_pk_fma_f32 per pair), added at the end
#pragma unroll
    for (int e = 0; e < EPT; ++e) gv2[e] = f32x2{0.f, 0.f};
    float ell = 0.f, he = 0.f, wsum = 0.f;
    const float mu = sc[8 * (t & 1) + 2 * (2 * pr + rs)];
    const float *z0 = zred + ((2 * pr) * 2 + rs) * ZS, *z1 = zred + ((2 * pr + 1) * 2 + rs) * ZS;   // the pair's two waves
    const float *erow = cur + (row_ok ? row : 0) * Mp;
    const bool first = t7 == 0 || t7 == nA;   // the row's first thread: mu, the value's partials
#pragma unroll 1
    for (int ch = 0; ch < nch; ++ch) {
      float w[MC];
#pragma unroll
      for (int m = 0; m < MC; ++m) {
        const float z = mu + (z0[ch * 8 + m] + z1[ch * 8 + m]);
        const float u = (z - tm) * ti;
        const bool on = row_ok && ch * 8 + m < M;
        w[m] = on ? -u * ti : 0.f;
      }
      if (first) {   // (only the waves that hold a row's first thread run this)
#pragma unroll
        for (int m = 0; m < MC; ++m) {
          const float z = mu + (z0[ch * 8 + m] + z1[ch * 8 + m]);
          const float u = (z - tm) * ti;
          const bool on = row_ok && ch * 8 + m < M;
          ell = on ? fmaf(-0.5f * u, u, ell) : ell;
          wsum += w[m];
          const float ev = row_ok ? erow[ch * 8 + m] : 0.f;   // (pad columns are zero)
          he = fmaf(0.5f * ev, ev, he);
        }
      }
      if (MC >= 4) {
        f32x4 ev[EPT][MC >= 4 ? MC / 4 : 1];
#pragma unroll
        for (int e = 0; e < EPT; ++e)
#pragma unroll
          for (int j = 0; j < MC / 4; ++j) ev[e][j] = *(const f32x4 *)(cur + eo[e] + ch * 8 + 4 * j);
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          f32x2 v = gv2[e];
#pragma unroll
          for (int j = 0; j < MC / 4; ++j) {
            v = __builtin_elementwise_fma(f32x2{w[4 * j + 0], w[4 * j + 1]}, f32x2{ev[e][j].x, ev[e][j].y}, v);
            v = __builtin_elementwise_fma(f32x2{w[4 * j + 2], w[4 * j + 3]}, f32x2{ev[e][j].z, ev[e][j].w}, v);
          }
          gv2[e] = v;
        }
      } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          float v = gv2[e].x;
#pragma unroll
          for (int m = 0; m < MC; ++m) v = fmaf(w[m], cur[eo[e] + m], v);
          gv2[e].x = v;
        }
      }
    }
    float gv[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) gv[e] = gv2[e].x + gv2[e].y;
    if (t7 == 0 || t7 == nA) {   // the row's scalar partials; workgroup threads 0 .. 3 write the history entry below
      float lg = 0.f, bad = 0.f;
      if (row_ok) {
        const float cr = sc[8 * (t & 1) + 2 * (2 * pr + rs) + 1];
        lg = logf(cr);
        bad = (cr > 0.f) ? 0.f : 1.f;
      } else {
        ell = 0.f; he = 0.f;
      }
      float *ps = sc + 16 + 4 * (2 * pr + rs);
      ps[0] = ell; ps[1] = he; ps[2] = lg; ps[3] = bad;
    }
    // gradient entries + Optimisers.update! + ClipScale, in registers
    // Adam with the two bias corrections as reciprocals taken once per step (one division and one square root per entry instead of three
    // divisions: within an ulp of optim_rules.h adam_step, inside this loop's stated rounding-level agreement)
    const float rc1 = RULE == 1 ? 1.f / cc_tab[t & 255][0] : 0.f, rc2 = RULE == 1 ? 1.f / cc_tab[t & 255][1] : 0.f;
    auto adam = [&](float x, float g, float &m, float &v) {
      m = fmaf(b1, m, (1.f - b1) * g);
      v = fmaf(b2, v, ((1.f - b2) * g) * g);
      return x - (eta * (m * rc1)) / (sqrtf(v * rc2) + aeps);
    };
    float gE[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const bool diag = e == diag_e;
      if (diag) gE[e] = (float)(-(double)gv[e] * invM - direct / (double)px[e]);
      else gE[e] = -gv[e] * invMf;
      if (!eok[e]) gE[e] = 0.f;
    }
    const float gmu = mu_own ? (float)(-(double)wsum * invM) : 0.f;
    double e_t = 0.0, gamma = a.eta;   // DoG / DoWG: the step size of this step; gamma: the proximal operator's step size
    if (RULE >= 2) {
      // ||x - x0||^2 and ||g||^2 over ALL parameters (src/optimization/rules.jl:26-42, :48-64): this workgroup's share, then every
      // workgroup's -- slots of this step's own that hold NaN until their workgroup has stored them (the data are their own flags)
      double nn[2] = {0.0, 0.0};
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const double dx = eok[e] ? (double)px[e] - (double)pm[e] : 0.0, gg = (double)gE[e];
        nn[0] += dx * dx;
        nn[1] += gg * gg;
      }
      if (mu_own) {
        const double dx = (double)mx - (double)mm1, gg = (double)gmu;
        nn[0] += dx * dx;
        nn[1] += gg * gg;
      }
      block_sum_n<double, NT, 2>(nn, gred);
      if (tid == 0) {
        double *pp = a.part + ((size_t)t * nwg + b) * 2;
        // (a NaN norm -- diverged parameters -- travels as +Inf: NaN means "not delivered"; the step size and the value come out non-finite either way)
        __hip_atomic_store(pp, nn[0] == nn[0] ? nn[0] : (double)INFINITY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + 1, nn[1] == nn[1] ? nn[1] : (double)INFINITY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *ok_s = 1;
      }
      lds_barrier();
      double sums[2] = {0.0, 0.0};
      for (int k = tid; k < nwg; k += NT) {
        const double *pp = a.part + ((size_t)t * nwg + k) * 2;
        int budget = a.spin;
        double q0, q1;
        while (true) {
          q0 = __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          q1 = __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (q0 == q0 && q1 == q1) break;
          if (--budget <= 0) { atomicAnd(ok_s, 0); q0 = q1 = 0.0; break; }
          __builtin_amdgcn_s_sleep(1);
        }
        sums[0] += q0;
        sums[1] += q1;
      }
      lds_barrier();
      if (!*ok_s) {
        if (tid == 0) atomicOr(a.status, 8);
        lost = true;
        continue;
      }
      block_sum_n<double, NT, 2>(sums, gred);
      dog_r = fmax(sqrt(sums[0]), dog_r);
      if (RULE == 3) {
        const double r2 = dog_r * dog_r;
        dog_v = dog_v + r2 * sums[1];
        e_t = r2 / sqrt(dog_v);
      } else {
        dog_v = dog_v + sums[1];
        e_t = dog_r / sqrt(dog_v);
      }
      gamma = e_t;
    }
    const double tt = (double)(a.t0 + t + 1);
    const double wa = (a.avg_eta + 1.0) / (tt + a.avg_eta), wb = 1.0 - wa;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      if (!eok[e]) continue;
      const bool diag = e == diag_e;
      if (RULE == 0) px[e] = descent_step(px[e], gE[e], eta);
      else if (RULE == 1) px[e] = adam(px[e], gE[e], pm[e], pv[e]);
      else px[e] = (float)((double)px[e] - e_t * (double)gE[e]);
      if (clip && diag) px[e] = clip_step(px[e], ceps);
      if (prox && diag) px[e] = prox_entropy_step(px[e], (float)gamma);
      if (averaging) pa[e] = poly_avg_step<float>(px[e], pa[e], wa, wb);
    }
    if (mu_own) {
      if (RULE == 0) mx = descent_step(mx, gmu, eta);
      else if (RULE == 1) mx = adam(mx, gmu, mm1, mv1);
      else mx = (float)((double)mx - e_t * (double)gmu);
      if (averaging) ma = poly_avg_step<float>(mx, ma, wa, wb);
    }
    publish((t + 1) & 1);
    if (DB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // vmcnt(0): the next slab is in
    lds_barrier();   // (every thread is done with this step's slab and the exchange areas; the rows' scalar partials are in place)
    if (tid < 4) a.hist[((size_t)t * 4 + tid) * nwg + b] = ((double)sc[16 + tid] + (double)sc[20 + tid]) + ((double)sc[24 + tid] + (double)sc[28 + tid]);
    if (!DB && t + 1 < a.n_steps) {
      slab_in(t + 1, S);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
    }
  }
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    if (!eok[e]) continue;
    const size_t at = (size_t)d + (size_t)(k0 + e) * d + row;
    a.params[at] = px[e];
    if (RULE == 1) { a.opt_state[at] = pm[e]; a.opt_state[plen + at] = pv[e]; }
    if (averaging) a.avg[at] = pa[e];
  }
  if (mu_own) {
    a.params[row] = mx;
    if (RULE == 1) { a.opt_state[row] = mm1; a.opt_state[plen + row] = mv1; }
    if (averaging) a.avg[row] = ma;
  }
  if (RULE >= 2 && b == 0 && tid == 0) { a.dog_sc[0] = dog_v; a.dog_sc[1] = dog_r; }
}

// elbo[t] (and the status word) from the per-step partials of k_fr_rows_loop; one workgroup per step
__global__ __launch_bounds__(256) void k_fr_rows_value(int d, int nwg, int M_local, int M_total, int ent_kind, double ell_const, const double *hist,
                                                       double *elbo, float *value_last, int n_steps, int *status) {
  __shared__ double red[4];
  const int t = blockIdx.x, tid = threadIdx.x;
  double s[4] = {0, 0, 0, 0};
  for (int k = 0; k < 4; ++k)
    for (int i = tid; i < nwg; i += 256) s[k] += hist[((size_t)t * 4 + k) * nwg + i];
  for (int k = 0; k < 4; ++k) s[k] = block_sum<double, 256>(s[k], red);
  if (tid == 0) {
    const double Mt = (double)M_total;
    const double ent = (ent_is_closed(ent_kind) ? 0.5 * d * (1.0 + kLog2Pi) : s[1] / Mt + 0.5 * d * kLog2Pi) + s[2];
    const double value = -((s[0] + (double)M_local * ell_const) / Mt + ent);
    elbo[t] = -value;
    if (t == n_steps - 1 && value_last) *value_last = (float)value;
    int st = 0;
    if (!isfinite(value)) st |= 1;
    if (s[3] > 0.0) st |= 2;
    if (st && status) atomicOr(status, st);
  }
}

bool fr_rows_loop_ok(const mivi_ctx *c) {
  const int d = c->cfg.d, M = c->cfg.n_mc;
  const bool stl = c->cfg.entropy == MIVI_ENT_STL || c->cfg.entropy == MIVI_ENT_STL_ZERO_GRAD;
  if (!(c->cfg.family == MIVI_FULLRANK && c->cfg.dtype == MIVI_F32 && c->target == TGT_DIAG_GAUSS && !c->bij_on && !stl && M >= 1 && M <= kRowsMaxM &&
        c->cfg.m_offset == 0 && c->M_total == M))
    return false;
  const int Mp = rows_mp(rows_mm(M));
  // two rows' threads fit 128 (d <= 1126); the slab fits the LDS beside the exchange areas; 16-byte slab copies
  return d >= 8 && d <= 126 * kRowsEPT - 8 && ((long long)d * Mp) % 4 == 0 && (size_t)d * Mp * 4 + (8 * kRowsMaxM + 32 + 2 * 256 + 32) * 4 <= c->lds_max;
}
size_t fr_rows_eps_bytes(const mivi_ctx *c, int n_steps) { return (size_t)n_steps * c->cfg.d * rows_mp(rows_mm(c->cfg.n_mc)) * sizeof(float); }
size_t fr_rows_hist_doubles(const mivi_ctx *c, int n_steps) { return (size_t)n_steps * 4 * (size_t)((c->cfg.d + 3) / 4); }

// eps_all: fr_rows_eps_bytes; hist: fr_rows_hist_doubles; elbo: n_steps doubles; value: one float (the last step's objective value)
size_t fr_rows_part_bytes(const mivi_ctx *c, int n_steps) { return (size_t)n_steps * (size_t)((c->cfg.d + 3) / 4) * 2 * sizeof(double); }

// gen: the general loop's description (rules / operators / averager beyond Descent / Adam + ClipScale; part: fr_rows_part_bytes) or nullptr
bool launch_fr_rows_loop(mivi_ctx *c, void *params, void *opt_state, uint64_t idx0, long long t0, int n_steps, int rule, double eta, double clip_eps,
                         float *eps_all, double *hist, double *elbo, void *value, const mivi_loop_t *gen, double *part) {
  const int d = c->cfg.d, M = c->cfg.n_mc, d4 = (d + 3) / 4, MM = rows_mm(M), Mp = rows_mp(MM);
  {
    const long long n = (long long)n_steps * d4 * Mp;
    hipLaunchKernelGGL(k_eps_steps, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->cfg.seed, idx0, n_steps, d, M, Mp, c->cfg.m_offset, eps_all);
  }
  FrRowsArgs a;
  a.d = d; a.M = M; a.Mp = Mp; a.n_steps = n_steps; a.rule = rule; a.ent_kind = c->cfg.entropy; a.m_offset = c->cfg.m_offset; a.M_total = c->M_total;
  a.params = (float *)params; a.opt_state = (float *)opt_state;
  a.t_mean = (const float *)c->t_mean.p; a.t_istd = (const float *)c->t_istd.p;
  a.eps_all = eps_all; a.t0 = t0; a.eta = eta; a.clip_eps = clip_eps; a.b1 = 0.9; a.b2 = 0.999; a.adam_eps = 1e-8;
  a.hist = hist;
  a.op = (clip_eps == clip_eps) ? 1 : 0; a.averager = 0; a.avg_eta = 0.0; a.avg = nullptr; a.x0 = nullptr; a.dog_sc = nullptr; a.part = part;
  a.status = (int *)c->status.p; a.spin = 1 << 20;
  if (gen) {
    a.op = gen->op; a.averager = gen->averager; a.avg_eta = gen->avg_eta; a.avg = (float *)gen->avg_params_dev;
    a.b1 = gen->beta1; a.b2 = gen->beta2; a.adam_eps = gen->adam_eps;
    if (rule >= 2) {
      a.x0 = (const float *)gen->opt_state_dev;
      a.dog_sc = (double *)((char *)gen->opt_state_dev + mivi_dog_state_bytes(c) - 16);
      (void)hipMemsetAsync(part, 0xFF, fr_rows_part_bytes(c, n_steps), c->stream);   // (NaN: not delivered yet)
    }
  }
  const int nwg = (d + 3) / 4;   // two row pairs per workgroup
  a.nch = MM >= 8 ? MM / 8 : 1;
  const size_t extras = (8 * kRowsMaxM + 32 + 2 * 256 + 32) * sizeof(float), slab = (size_t)d * Mp * sizeof(float);
  const bool db = 2 * slab + extras <= c->lds_max;
  const size_t lds = (db ? 2 : 1) * slab + extras;
  bool launched = true;
  auto go = [&](auto kern) {
    // the LDS the kernel asks for must be granted, and -- DoG / DoWG: two norm partials exchanged grid-wide every step by spin-wait -- every
    // workgroup must be resident at once: both checked against the device, not assumed (false: the caller takes the graph of launches)
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        (rule >= 2 && !grid_resident(c, reinterpret_cast<const void *>(kern), kRowsNT, lds, nwg))) {
      (void)hipGetLastError();
      launched = false;
      return;
    }
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(kRowsNT), lds, c->stream, a);
  };
  const bool general = gen && (gen->op == 2 || gen->averager == 1);
  auto pick = [&](auto r, auto dbl) {
    constexpr int R = decltype(r)::value;
    constexpr bool D = decltype(dbl)::value;
    if (R < 2 && !general) {
      switch (MM) {
        case 1: go(k_fr_rows_loop<R, 1, D, false>); break;
        case 2: go(k_fr_rows_loop<R, 2, D, false>); break;
        case 4: go(k_fr_rows_loop<R, 4, D, false>); break;
        default: go(k_fr_rows_loop<R, 8, D, false>); break;
      }
    } else {
      switch (MM) {
        case 1: go(k_fr_rows_loop<R, 1, D, true>); break;
        case 2: go(k_fr_rows_loop<R, 2, D, true>); break;
        case 4: go(k_fr_rows_loop<R, 4, D, true>); break;
        default: go(k_fr_rows_loop<R, 8, D, true>); break;
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
  if (rule == 0) { if (db) pick(I0{}, std::true_type{}); else pick(I0{}, std::false_type{}); }
  else if (rule == 1) { if (db) pick(I1{}, std::true_type{}); else pick(I1{}, std::false_type{}); }
  else if (rule == 2) { if (db) pick(I2{}, std::true_type{}); else pick(I2{}, std::false_type{}); }
  else { if (db) pick(I3{}, std::true_type{}); else pick(I3{}, std::false_type{}); }
  if (!launched) return false;
  hipLaunchKernelGGL(k_fr_rows_value, dim3(n_steps), dim3(256), 0, c->stream, d, nwg, M, c->M_total, c->cfg.entropy, c->t_const, (const double *)hist, elbo,
                     (float *)value, n_steps, (int *)c->status.p);
  return true;
}

}  // namespace mivi
