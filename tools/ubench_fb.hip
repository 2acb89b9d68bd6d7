// Developer micro-benchmark of the batch-engine kernels (csrc/kernels_fullrank_batch.hip) with work-skipping knock-outs (-DMIVI_DEV):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DMIVI_DEV tools/ubench_fb.hip -o tools/bin/ubench_fb.exe && tools/bin/ubench_fb.exe [L]
// knock bits: 1 no operand DMA after the prologue, 2 no MFMAs, 4 every stage read from ring slot 0, 8 no gradient stores (VJP), 16 no epilogue
#ifndef UB_WJ
#define UB_WJ 2
#endif
#ifndef UB_PF
#define UB_PF 1
#endif
#include "../advancedvi.jl_amd/csrc/kernels_fullrank_batch.hip"
#include <cstdio>
namespace mivi { void invalidate_graph(mivi_ctx *) {} }
using namespace mivi;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
int main(int argc, char **argv) {
  const int L = argc > 1 ? atoi(argv[1]) : 20, d = 1024, M = 256;
  mivi_ctx c{};
  c.cfg.d = d; c.cfg.n_mc = M; c.cfg.family = MIVI_FULLRANK; c.cfg.dtype = MIVI_F32; c.cfg.entropy = 0; c.M_total = M;
  const size_t plen = (size_t)d + (size_t)d * d, pw = fb_plane_words(&c, M) * 4;
  float *params, *tm, *tis;
  CK(hipMalloc(&params, plen * 4)); CK(hipMalloc(&tm, d * 4)); CK(hipMalloc(&tis, d * 4));
  std::vector<float> hp(plen, 0.f), hv(d, 1.f);
  for (int i = 0; i < d; ++i) { hp[d + (size_t)i * d + i] = 1.f; for (int j = 0; j < i; ++j) hp[d + (size_t)j * d + i] = 0.01f * ((i * 7 + j * 3) % 11 - 5); }
  CK(hipMemcpy(params, hp.data(), plen * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(tm, hv.data(), d * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(tis, hv.data(), d * 4, hipMemcpyHostToDevice));
  c.t_mean.p = tm; c.t_istd.p = tis;
  FbTables &t = c.fb;
  CK(hipMalloc(&t.CA.p, fb_cplane_words(&c) * 4)); CK(hipMalloc(&t.epsP.p, L * pw)); CK(hipMalloc(&t.epsV.p, L * pw)); CK(hipMalloc(&t.WV.p, L * pw));
  CK(hipMalloc(&t.ell.p, (size_t)L * 32 * 8 * 8)); CK(hipMalloc(&t.he.p, (size_t)L * 16 * 8 * 8)); CK(hipMalloc(&t.ld.p, 64 * 8 + 64));
  CK(hipMalloc(&t.grads.p, (size_t)L * plen * 4)); CK(hipMalloc(&t.values.p, L * 4 + 64)); CK(hipMalloc(&c.status.p, 64));
  CK(hipMemset(t.grads.p, 0, (size_t)L * plen * 4));
  const FbTab *tab = fb_prepare(&c, M, L);
  hipStream_t st; CK(hipStreamCreate(&st)); c.stream = st;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("L = %d lanes, d = %d, M = %d: %d product tiles, %d VJP tiles\n", L, d, M, tab->n_prod, tab->n_vjp);
  {   // clock ramp: ~0.5 s of the two products before anything is timed
    FbArgs a = fb_args(&c, params, M);
    a.L = L; a.grads = (float *)t.grads.p; a.grad_stride = (long long)plen; a.values = (float *)t.values.p; a.value_stride = 1; a.lane_last = -1;
    a.rng.seed = 1; a.rng.idx_base = 5;
    hipLaunchKernelGGL(k_fb_cplanes, dim3(((d / 32) * (d / 16) + 3) / 4), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_fb_eps, dim3((d / 64) * (M / 32), L), dim3(512), 0, st, a);
    for (int r = 0; r < 4000; ++r) {
      a.work = (const int4 *)tab->prod.p; a.n_work = tab->n_prod; hipLaunchKernelGGL((k_fb_prod<UB_WJ, UB_PF, FB_DIAG>), dim3(tab->n_prod), dim3(512 / UB_WJ), 0, st, a);
      a.work = (const int4 *)tab->vjp.p; a.n_work = tab->n_vjp; hipLaunchKernelGGL((k_fb_vjp<UB_WJ, UB_PF>), dim3(tab->n_vjp), dim3(512 / UB_WJ), 0, st, a);
    }
    CK(hipStreamSynchronize(st));
  }
#ifdef MIVI_DEV
  const int knocks[] = {0, 16, 1 | 16, 1 | 16 | 32, 1 | 16 | 64, 1 | 16 | 32 | 64, 1 | 2 | 16, 1 | 2 | 16 | 32, 1 | 2 | 16 | 32 | 64};
#else
  const int knocks[] = {0, 0};
#endif
  for (int kn : knocks) {
    FbArgs a = fb_args(&c, params, M);
    a.L = L; a.grads = (float *)t.grads.p; a.grad_stride = (long long)plen; a.values = (float *)t.values.p; a.value_stride = 1; a.lane_last = -1;
    a.rng.seed = 1; a.rng.idx_base = 5; a.knock = kn;
    hipLaunchKernelGGL(k_fb_cplanes, dim3(((d / 32) * (d / 16) + 3) / 4), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_fb_eps, dim3((d / 64) * (M / 32), L), dim3(512), 0, st, a);
    float ms[4] = {0, 0, 0, 0};
    const int reps = 200;
    for (int which = 0; which < 4; ++which) {
      for (int r = -3; r < reps; ++r) {
        if (r == 0) CK(hipEventRecord(e0, st));
        if (which == 0) hipLaunchKernelGGL(k_fb_eps, dim3((d / 64) * (M / 32), L), dim3(512), 0, st, a);
        if (which == 1) { a.work = (const int4 *)tab->prod.p; a.n_work = tab->n_prod; hipLaunchKernelGGL((k_fb_prod<UB_WJ, UB_PF, FB_DIAG>), dim3(tab->n_prod), dim3(512 / UB_WJ), 0, st, a); }
        if (which == 2) { a.work = (const int4 *)tab->vjp.p; a.n_work = tab->n_vjp; hipLaunchKernelGGL((k_fb_vjp<UB_WJ, UB_PF>), dim3(tab->n_vjp), dim3(512 / UB_WJ), 0, st, a); }
        if (which == 3) hipLaunchKernelGGL(k_fb_value, dim3(L), dim3(256), 0, st, a);
      }
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[which], e0, e1));
      ms[which] /= reps;
    }
    printf("knock %2d: eps %7.2f us  prod %7.2f us  vjp %7.2f us  value %6.2f us   (per estimate: %.3f us)\n", kn, ms[0] * 1e3, ms[1] * 1e3, ms[2] * 1e3, ms[3] * 1e3,
           (ms[0] + ms[1] + ms[2] + ms[3]) * 1e3 / L);
    if (kn == 0) {   // every lane reading lane 0's operand planes (timing only: what the memory side costs)
      a.plane_stride = 0;
      for (int which = 1; which <= 2; ++which) {
        for (int r = -3; r < reps; ++r) {
          if (r == 0) CK(hipEventRecord(e0, st));
          if (which == 1) { a.work = (const int4 *)tab->prod.p; a.n_work = tab->n_prod; hipLaunchKernelGGL((k_fb_prod<UB_WJ, UB_PF, FB_DIAG>), dim3(tab->n_prod), dim3(512 / UB_WJ), 0, st, a); }
          if (which == 2) { a.work = (const int4 *)tab->vjp.p; a.n_work = tab->n_vjp; hipLaunchKernelGGL((k_fb_vjp<UB_WJ, UB_PF>), dim3(tab->n_vjp), dim3(512 / UB_WJ), 0, st, a); }
        }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[which], e0, e1));
      }
      printf("  all lanes on lane 0's planes: prod %7.2f us  vjp %7.2f us\n", ms[1] / reps * 1e3, ms[2] / reps * 1e3);
    }
  }
#ifndef MIVI_DEV
  return 0;
#endif
  // placement + timeline of one launch of each product (knock 0): which CU ran which workgroup, when
  long long *dbg; CK(hipMalloc(&dbg, (4096 * 8 + 8) * 8));
  std::vector<long long> hd(4096 * 8 + 8);
  for (int which = 1; which <= 2; ++which) {
    FbArgs a = fb_args(&c, params, M);
    a.L = L; a.grads = (float *)t.grads.p; a.grad_stride = (long long)plen; a.values = (float *)t.values.p; a.value_stride = 1; a.lane_last = -1;
    a.rng.seed = 1; a.rng.idx_base = 5; a.dbg = dbg;
    const int n = which == 1 ? tab->n_prod : tab->n_vjp;
    a.work = (const int4 *)(which == 1 ? tab->prod.p : tab->vjp.p); a.n_work = n;
    CK(hipMemset(dbg, 0, (4096 * 8 + 8) * 8));
    if (which == 1) hipLaunchKernelGGL((k_fb_prod<UB_WJ, UB_PF, FB_DIAG>), dim3(n), dim3(512 / UB_WJ), 0, st, a); else hipLaunchKernelGGL((k_fb_vjp<UB_WJ, UB_PF>), dim3(n), dim3(512 / UB_WJ), 0, st, a);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(hd.data(), dbg, (4096 * 8 + 8) * 8, hipMemcpyDeviceToHost));
    printf("shader clock of block 0 over its main loop: %.0f MHz\n", (double)(hd[8 * 4096 + 2] - hd[8 * 4096 + 1]) / (double)(hd[2] - hd[1]) * 100.0);
    std::vector<int4> wk(n); CK(hipMemcpy(wk.data(), a.work, (size_t)n * 16, hipMemcpyDeviceToHost));
    long long t0 = hd[1];
    for (int b = 0; b < n; ++b) t0 = std::min(t0, hd[b * 8 + 1]);
    printf("%s: block: xcc se cu | rb cb lane | start main end (us, 100 MHz clock)\n", which == 1 ? "prod" : "vjp");
    std::vector<int> percu(8 * 8 * 16, 0);
    for (int b = 0; b < n; ++b) {
      const unsigned hw = (unsigned)hd[b * 8], xcc = (unsigned)(hd[b * 8] >> 32) & 15;
      const int cu = (hw >> 8) & 15, se = (hw >> 13) & 7;
      percu[(xcc * 8 + se) * 16 + cu]++;
      if (b < 48 || b % 16 == 0 || b >= n - 16)
        printf("%4d: %u %d %2d | %d %d %2d | %7.2f %7.2f %7.2f | epi %6.2f %6.2f\n", b, xcc, se, cu, wk[b].y & 0xffff, wk[b].y >> 16, wk[b].x, (hd[b * 8 + 1] - t0) * 0.01, (hd[b * 8 + 2] - t0) * 0.01,
               (hd[b * 8 + 3] - t0) * 0.01, hd[b * 8 + 4] ? (hd[b * 8 + 4] - t0) * 0.01 : 0.0, hd[b * 8 + 5] ? (hd[b * 8 + 5] - t0) * 0.01 : 0.0);
    }
    int used = 0, mx = 0; for (int v : percu) { used += v > 0; mx = std::max(mx, v); }
    printf("%s: %d workgroups on %d distinct CUs, at most %d per CU\n", which == 1 ? "prod" : "vjp", n, used, mx);
  }
  return 0;
}
