// Internal declarations shared by the libmivi translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/mivi.h"
#include "philox.h"

namespace mivi {

constexpr double kLog2Pi = 1.8378770664093454835606594728112;

// --------------------------------------------------------------------------------------------
// Kernel argument blocks (plain structs passed by value)
// --------------------------------------------------------------------------------------------
struct RngArgs {
  uint64_t seed;
  uint64_t idx_base;        // estimate index = idx_base + (idx_ptr ? *idx_ptr : 0)
  const uint64_t *idx_ptr;  // device counter (graph replay) or nullptr
  int m_offset;             // first GLOBAL sample column of this context
};

enum TargetKind : int {
  TGT_NONE = 0,
  TGT_DIAG_GAUSS = 1,
  TGT_DENSE_GAUSS = 2,
  TGT_LOGREG = 3,
  TGT_FUNNEL = 4,
  TGT_CALLBACK = 5
};

// Fused funnel target (mean-field): the only cross-row coupling of Neal's funnel is S_m = sum_i x_im^2 per sample, needed by
// row 0's gradient and by ell -- and every use of it is LINEAR in S_m with a per-column weight any workgroup can re-derive
// from the eps stream (exp(-2 e1_m), exp(-2 e1_m) eps_0m).  So the main kernel's workgroups (rows >= 1) leave two scalar
// partials each,  A = sum_{i,m} x_im^2 exp(-2 e1_m)  and  B = sum_{i,m} x_im^2 exp(-2 e1_m) eps_0m  over their own rows, fold
// -A/2 into their ell partial, and whoever assembles the value adds the O(M) per-column terms and finishes row 0.
struct FunnelFin {
  const double *ab;    // [2][n_part] the A and B partials of the main kernel's workgroups  (nullptr = no funnel work)
  int n_part;
  const void *params;  // [mu; sigma]
  RngArgs rng;
  int d4, M;
  double sigma_v;
  const void *row0;    // nullptr: row 0's (mu, sigma) are params[0], params[d]; else {mu_0, sigma_0} published by the row-0 workgroup of the SAME
                       // launch (k_mf_funnel_sgd_loop): read with agent-scope atomic loads
  const unsigned *wait_word;   // k_mf_funnel_sgd_loop: the A / B (and every other) partial of this step is complete once wait_word[b] >= wait_val
  unsigned wait_val;           // for every row quad b < wait_n (one flag per workgroup: 512 read-modify-writes on ONE counter took 8 us);
  int wait_n, wait_budget;     // waited for (bounded, status bit 8) AFTER the per-column work, which needs none of them
  double *mirror;              // ... then copied in ONE batch of loads (one memory round trip instead of one per partial kind) into this LDS image,
  const double *mirror_src;    // which the partial pointers of the ValueIn (and `ab`) then point into
  int mirror_n;
};

// reduction inputs of the objective value, summed in a fixed order by one workgroup
struct ValueIn {
  FunnelFin fn;
  const double *ell_part;  // per-workgroup partial sums of sum_m ell_m (variable part)
  int n_ell_part;
  const double *ell_part2; // second set (the launching kernel's own partials)
  int n_ell_part2;
  const void *ell;         // per-sample ell (T), generic target route
  int n_ell;
  const double *he_part;   // partial sums of sum_m 0.5|eps_m|^2
  int n_he_part;
  const double *ld_part;   // [2][n_ld_part]: partial sums of log C_ii, then counts of non-positive C_ii (or nullptr)
  int n_ld_part;
  double ell_const;        // constant added per sample (target normaliser)
};

// how results are emitted
struct OutArgs {
  void *grad;        // T[params_len]  (final mode)
  void *value;       // T[1]           (final mode)
  void *partials;    // T[params_len+2] (partials mode, un-normalised)
  int partials_mode; // 0 final, 1 partials
  long long scalars_off;  // index of [sum ell, sum 0.5 eps^2] inside the partials buffer
  int ent_kind;
  int M_total;       // global n_samples (normaliser)
  int M_local;
  int *status;       // device status word: bit0 nonfinite value, bit1 non-positive scale diag
  double *elbo_rec;  // optional: elbo_rec[rec_slot] = -value (optimize loop)
  int rec_slot;
  // partials mode on the peer-to-peer route, DIRECT: every entry of the partial vector is stored straight into its OWNER's staging area
  // (kernels_p2p.hip: no ring slot, no push pass).  p2p_direct: device-resident P2PDirectTab; (p2p_gi, p2p_v): the exchange group this estimate
  // belongs to, counted from the exchange kernel's start, and its place in the group.  Second-generation full-rank f32 kernels only.
  const void *p2p_direct;
  int p2p_gi, p2p_v;
};
struct P2PDirectTab {   // where the ranks' staging areas are mapped in this process (lane 0 of the exchange)
  char *stage[8];
  long long n;          // slice length (elements)
  int R, rank, GV, pad;
  const unsigned *ctr;  // the exchange lane's counters: ctr[0] = exchanges completed (the next group's epoch is ctr[0] + 1 + p2p_gi)
};

// Optimiser step fused into the VJP epilogue (device-resident optimisation loop, full-rank f32): every lower-triangle
// tile applies Descent / Adam (+ ClipScale on the diagonal) to its own parameters right where the gradient entry is
// produced, so the separate update kernel and the gradient round trip through HBM disappear.  rule < 0: off.
struct FusedUpdate {
  int rule = -1;              // 0 Descent, 1 Adam (optim_rules.h: bitwise the same arithmetic as kernels_update.hip)
  void *params;               // the parameter vector being optimised (same buffer the estimate reads)
  void *state;                // Adam: [m (params_len); v (params_len)]
  const long long *t_ptr;     // Adam step count = t_base + *t_ptr
  long long t_base;
  double eta, b1, b2, eps, clip_eps;
  int do_clip = 0;            // ClipScale on the scale diagonal after the step (explicit: epsilon <= 0 is a legal ClipScale)
};

template <typename T>
struct MfArgs {
  int d;
  int M;               // local samples processed by this launch
  int n_cc;            // column chunks (gridDim.y)
  int cols_per_cc;
  const T *params;     // [mu; sigma]
  RngArgs rng;
  int target;          // TGT_DIAG_GAUSS (fused) or TGT_NONE (W from G buffer)
  const T *t_mean;     // diag gauss mean[d]
  const T *t_istd;     // 1/std[d]
  const T *G;          // generic route: d x M gradient of log pi (ld = d)
  int want_grad;
  // scratch
  double *row_part;    // [n_cc][2*d4*4] partial row sums when n_cc > 1
  double *sc_part;     // [4 or 6][n_blocks] scalar partials: ell, 0.5 eps^2, log sigma, #bad sigma (+ funnel A, B)
  ValueIn vin;
  OutArgs out;
  long long *dbg;      // optional timeline: dbg[block*8 + k] = wall_clock64() stamps (nullptr = off)
  int has_prev;        // extra workgroup (blockIdx.x == ceil(d/4)) assembles the previous estimate's value
  ValueIn prev_vin;
  OutArgs prev_out;
};

template <typename T>
struct SampleArgs {  // rand(rng, q, M) -> Z (and eps)
  int d, M;
  const T *params;
  RngArgs rng;
  T *Z;       // d x M, ld = d
  T *eps;     // d x M, ld = ld_eps (or nullptr)
  int ld_eps;
  T *epsT;    // M x d transposed, epsT[m + k*ld_epsT] (or nullptr)
  int ld_epsT;
  double *he_part;  // per-block partial of sum 0.5 eps^2 (or nullptr)
};

template <typename T>
struct FrArgs {
  int d, M, dP, MP;
  const T *params;     // [mu; vec C]
  const T *eps;        // eps[i + m*dP]
  const T *epsT;       // epsT[m + k*MP]
  T *Z;                // Z[i + m*d] or nullptr
  T *W;                // W[i + m*d]: grad log pi (+ STL term)
  T *RT;               // (Z - t_mean)^T: RT[m + k*MP]  (dense target)
  int fused_target;    // TGT_NONE: write Z; TGT_DIAG_GAUSS: write W + ell partial; TGT_DENSE_GAUSS: write Z,RT
  const T *t_mean;
  const T *t_istd;
  const T *t_prec;     // precision matrix, ld = dP, zero padded
  double *ell_part;    // written by sample/target kernels
  double *ld_part;     // [2][nb]: per diagonal block sum log C_ii, then #non-positive C_ii (VJP kernel)
  ValueIn vin;
  OutArgs out;
  long long *dbg;      // optional timeline (nullptr = off)
  // work distribution + heterogeneous workgroups (MFMA kernel only)
  const int2 *work_tab;
  int n_work;          // index of the value workgroup (sample kernel) or INT_MAX
  int n_pre;           // leading workgroups that generate the next estimate's eps (VJP kernel)
  SampleArgs<T> next_eps;
  ValueIn prev_vin;
  OutArgs prev_out;
  FusedUpdate upd;     // VJP kernel, final mode only
  const T *adam_cc;    // LDS: the two Adam bias corrections of this step (set by the kernel)
};

struct ValueJob {      // deferred objective-value assembly of the previous estimate
  ValueIn vin;
  OutArgs out;
};
struct EpsJob {        // eps generation for the next estimate
  RngArgs rng;
  int parity;
};

template <typename T>
struct ColTargetArgs {  // standalone per-column targets: Z -> (ell, G)
  int d, M, kind;
  const T *Z;
  T *G;
  T *ell;
  const T *t_mean;
  const T *t_istd;
  double sigma_v;
  int want_grad;
  int constrained;   // funnel: theta_1 = s itself (no built-in exp bijector / log-Jacobian)
};

// --------------------------------------------------------------------------------------------
// Context
// --------------------------------------------------------------------------------------------
struct DevBuf {
  void *p = nullptr;
  size_t bytes = 0;
};

// third-generation batch engine (kernels_fullrank_batch.hip): work tables per lane count, per-lane work buffers (base + lane * stride)
struct FbTab {
  DevBuf prod, vjp, prod2, prod3;         // work tables: the draw's product, the VJP, the dense target's product, the sticking-the-landing product
  int n_prod = 0, n_vjp = 0, n_prod2 = 0, L = 0, M = 0;
};
struct FbTables {
  FbTab tab[4];                           // per lane count (the full step's, the last shorter step's, other batch lengths'), round robin
  int next_tab = 0;
  DevBuf CA, epsP, WV, ell, he, ld, grads, values;   // operand planes (tril(C) once per call; eps -- ONE orientation since round 6 -- and W per lane)
  DevBuf cscale, pscale, tscale, winv, rinv;               // power-of-two scales of the planes (fr_planes.h): rows of tril(C) / P / C^-T [2][d]; W per lane [M / 128][d]; R per lane [d / 128][M]
  DevBuf PA, RP;                          // dense-Gaussian target: planes of P (once per target), R = Z - m per lane
  bool PA_valid = false;
  DevBuf Tinv, TA, Eye;                   // sticking-the-landing estimators: C^-T (f32), its planes (once per call), the identity the solve takes
  DevBuf parts;                           // sharded batches: the lanes' shard-additive partial vectors (fb_part_len floats each), what one all-reduce sums
  int cap_L = 0, cap_M = 0, cap_LR = 0, cap_LP = 0;
};
struct FbStep {
  const void *params;
  int M, L;
  RngArgs rng;                            // lane l draws estimate rng_index(rng) + l
  void *grads; long long grad_stride;     // lane l's gradient (elements)
  void *values; long long value_stride;
  void *grad_last, *value_last;           // lane_last writes these instead (nullptr: none)
  int lane_last;
  int write_upper;                        // lanes write the exact zeros above the diagonal (0: their buffers hold them already)
  int dense;                              // dense-Gaussian target: two products per lane
  int stl;                                // sticking-the-landing estimators: W += C^-T eps as one more product per lane
  const FbTab *tab;
  int obj;                                // objective mode (mivi_estimate_objective): the lanes are consecutive blocks of n_mc samples of ONE estimate index; values only
  int ent_kind;                           // objective mode: the entropy estimator of the value
  int values_only;                        // no gradient is wanted (mivi_estimate_gradient_each without grads; objective mode): no VJP tile runs
  void *parts; long long part_stride;     // sharded batches: the VJP leaves lane l's UNNORMALISED partial vector at parts + l part_stride instead of a gradient
};

struct GraphCache {
  hipGraphExec_t exec = nullptr;
  int count = 0;
  const void *params = nullptr;
  void *value = nullptr;
  void *grad = nullptr;
  int kind = 0;
  double p0 = 0, p1 = 0;
  void *aux0 = nullptr, *aux1 = nullptr;
  mivi_loop_t loop{};   // kind 9: the configuration the captured loop was built for
};

}  // namespace mivi

struct mivi_ctx {
  mivi_config_t cfg;
  hipStream_t stream = nullptr;
  hipStream_t cap_stream = nullptr;   // internal stream used only for graph capture
  bool own_stream = false;
  std::string err;
  int target = mivi::TGT_NONE;
  int M_total = 0;
  size_t esize = 4;

  // target data
  mivi::DevBuf t_mean, t_istd, t_prec;
  double t_const = 0.0;  // per-sample constant of log pi
  double funnel_sigma_v = 1.5;
  int funnel_constrained = 0;
  // Stacked bijector (mivi_set_bijector_stacked): per-coordinate kind (0 identity, 1 exp) and per-column sum of eta over exp rows
  mivi::DevBuf bij_mask, bij_ld;
  // collective behind the C ABI (mivi_comm_init): RCCL communicator + padded partial / packed-final buffers
  void *comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  mivi::DevBuf dist_P, dist_S, dist_F;
  bool bij_on = false;
  // Pipelined exchange (mivi_estimate_gradient_dist_n): the collective of estimate t runs on comm_stream under the kernels of t + 1;
  // partial vectors double-buffered (dist_P, dist_P2), event pairs per parity
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_part[2] = {nullptr, nullptr}, ev_comm[2] = {nullptr, nullptr};
  mivi::DevBuf dist_P2, dist_ring[6];       // ring of partial vectors (two for the RCCL pipeline, eight = two groups of four for the peer-to-peer pipeline)
  hipStream_t comm_stream2 = nullptr;
  // the batch engine's sharded batches (api_batch.hip fb_batch, dist): the all-reduce + finalisation of step s on fb_comm_stream under the
  // kernels of step s + 1 (streams of their own: comm_stream carries the CU mask the peer-to-peer pipeline needs)
  hipStream_t fb_comm_stream = nullptr;
  hipEvent_t fb_ev_part[2] = {nullptr, nullptr}, fb_ev_comm[2] = {nullptr, nullptr};
  int p2p_pipe_state = 1;   // persistent peer-to-peer pipeline in batched calls: 1 on, -1 off (mivi_p2p_set_pipeline: serial steps)
  int dist_route = 0;   // mivi_comm_set_route: 0 by size, 1 ncclAllReduce, 2 ncclReduceScatter / ncclAllGather, 3 peer-to-peer kernel
  // peer-to-peer exchange over xGMI (kernels_p2p.hip): one uncached allocation per rank [stage | final | flags], mapped into the peers by IPC
  void *p2p_buf = nullptr;
  size_t p2p_bytes = 0;
  void *p2p_peer[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // mapped bases (own entry = p2p_buf)
  bool p2p_opened[8] = {false, false, false, false, false, false, false, false};                   // hipIpcOpenMemHandle'd (to be closed)
  mivi::DevBuf rows_eps;   // kernels_fullrank_rows.hip: eps of all steps of a device-resident loop call
  mivi::DevBuf gen_scratch;  // kernels_meanfield.hip k_mf_gen_loop: DoG / DoWG partial norms of every step, arrival flags
  mivi::DevBuf p2p_tab, p2p_ctr, p2p_scratch, p2p_direct;   // (p2p_direct: P2PDirectTab, what the partial kernels need to store straight into the owners' staging areas)
  bool p2p_on = false;
  bool p2p_distinct = false;     // the attached exchange areas include one of ANOTHER device (not only this process's own contexts on this device)
  bool p2p_verified = false;     // mivi_p2p_selfcheck passed on every rank: only then does the automatic route take the peer-to-peer kernel at world > 1
  int p2p_rank = 0, p2p_world = 1, p2p_G = 1, p2p_vs = 0, p2p_spin = 1 << 21;
  long long p2p_n = 0, p2p_cn = 0;
  size_t p2p_lane_bytes = 0, p2p_off_fin = 0, p2p_off_arr = 0, p2p_off_farr = 0;
  // logreg
  const void *lr_X = nullptr;
  const uint8_t *lr_y = nullptr;
  mivi::DevBuf lr_X_own, lr_y_own, lr_scratch, lr_part, lr_Xrm, lr_xmax, lr_XA, lr_XB, lr_ZP;   // (lr_xmax: bits of max |X|, the scale of X's f16 splits)
  // minibatch view (AdvancedVI.subsample, docs/src/tutorials/subsampling.md:99-102): the full data set stays resident,
  // mivi_logreg_select_rows gathers the batch rows into lr_Xsub / lr_ysub / lr_Xrm_sub and points the active fields at them
  const void *lr_X_full = nullptr;
  const uint8_t *lr_y_full = nullptr;
  int64_t lr_n_full = 0;
  double lr_likeadj_full = 1.0;
  const void *lr_Xrm_act = nullptr;      // row-major copy the MFMA kernels read (full or batch)
  mivi::DevBuf lr_Xsub, lr_ysub, lr_Xrm_sub, lr_idx;
  int lr_route = 0;                        // 0 auto (by problem size), 1 matrix-core kernels, 2 VALU kernels (mivi_set_logreg_route)
  int64_t lr_n = 0;
  int lr_variant = 0;
  double lr_likeadj = 1.0;
  // callback
  mivi_logdensity_and_gradient_fn cb_grad = nullptr;
  mivi_logdensity_fn cb_value = nullptr;
  void *cb_user = nullptr;
  mivi_logdensity_gradient_and_hessian_fn cb_hess = nullptr;   // second-order plugin (mivi_gauss_expected_grad_hess2)
  void *cb_hess_user = nullptr;
  std::vector<char> h_Z, h_G, h_ell;

  // work buffers (sized for `cap_M` samples)
  int cap_M = 0;
  // buffers indexed [cur] are double-buffered so that, inside a captured graph, eps generation of estimate t+1
  // and the value assembly of estimate t overlap the contractions of the neighbouring estimates
  mivi::DevBuf eps[2], epsT[2], ell_part[2], he_part[2], sc_part[2], ld_part[2];
  mivi::DevBuf tabA, tabB, tabD;   // XCD-aware work tables of the MFMA kernels
  mivi::DevBuf stl_CT, stl_Dinv;   // transposed scale + inverted diagonal blocks (full-rank STL, f32)
  mivi::DevBuf stl_F;              // second-generation STL solve: packed operands (stl_dinv.h: pivot inverses + off-diagonal blocks, fragment order)
  bool want_stl_pack = false, stl_pack_done = false;   // Stein estimator: ask the sampling kernel to carry the solve's riders
  mivi::DevBuf stl_X;              // second-generation STL solve: X of the lower half + updated right-hand side of the upper half
  // second-generation full-rank kernels (kernels_fullrank_lds.hip): VJP work lists (32 x 32 and 64 x 64 tiles)
  mivi::DevBuf lds_tabV, lds_tabV64, lds_tabS;   // (lds_tabS: the strips of k_fr_vjp32s)
  int lds_nV = 0, lds_nV64 = 0, lds_nS = 0, lds_M = -1;
  bool d_idx_valid = false;          // the device-side estimate counter (d_idx[0]) is known to hold d_idx_expect
  uint64_t d_idx_expect = 0;
  mivi::DevBuf lds_tabSt;            // Stein accumulation stage: the full square of 64 x 64 tiles
  int lds_nSt = 0, lds_st_d = -1;
  mivi::ValueJob *defer_value = nullptr;   // non-null: run_estimate_lds(stop_after_target) hands its value job over instead of launching it
  bool value_deferred = false;
  int he_n[2] = {0, 0};            // number of sum-0.5-eps^2 partials behind he_part[parity] (depends on who drew eps)
  int nA = 0, nB = 0, nD = 0, tab_M = -1;
  int cur = 0;
  int mf_nblk = 0;
  mivi::DevBuf Z, W, RT, ell, X;
  mivi::DevBuf row_part, status, d_idx, acc, tmp_params, tmp_out;
  mivi::DevBuf obj_vals;   // mivi_estimate_objective on the batch engine: the lanes' values
  mivi::DevBuf stein_A, stein_g;   // Stein estimator: eps G^T accumulator (dP x dP, T) and the f64 column sums of G
  mivi::DevBuf h2_acc;             // second-order branch of the logistic-regression / funnel targets: f64 sums (kernels_hess2.hip)
  mivi::DevBuf dog_part;   // DoG / DoWG on large parameter vectors: 512 x 2 partial norms + the step size
  const uint64_t *idx_src = nullptr;   // mivi_set_index_source
  // speculative eps prefetch across single calls: the VJP kernel of estimate (seed, idx) also generates eps of
  // (seed, idx + 1) into the other parity; a following call for exactly that estimate skips its eps kernel
  bool pre_valid = false;
  mivi::RngArgs pre_rng{};
  int pre_M = 0, pre_parity = 0;
  int pre_capturing = 0;
  unsigned long long pre_capture_id = 0;
  long long *dbg = nullptr;   // timeline buffer supplied through mivi_debug_timeline (tools only)
  int dP = 0, MP = 0;

  mivi::GraphCache graph;
  mivi::FbTables fb;   // batch engine buffers (mivi_estimate_gradient_n / _each on the third-generation route)

  // Interleaved chains (mivi_estimate_gradient_n, full-rank + Gaussian targets): estimates at fixed parameters are independent, so a
  // batch is dealt round-robin onto `1 + n_kids` chains -- this context and child contexts with their own work buffers and streams --
  // whose kernels overlap on the device (the product of one chain's estimate runs beside the VJP of another's).
  int idx_stride = 1;            // estimate-index step between consecutive estimates of THIS context's chain
  bool is_child = false;         // target buffers are borrowed from the parent
  int n_cu = 0;                  // compute units of the device, LDS bytes a workgroup may take: what the launch-free loops with a grid-wide
  size_t lds_max = 64 * 1024;    // exchange per step check their grids against (hipOccupancyMaxActiveBlocksPerMultiprocessor x n_cu >= grid)
  bool exchange_lost = false;    // a grid-wide exchange expired once on this context (status bit 8): those loops are not taken again
  int last_status_bits = 0;      // the sticky device flags read_status saw last
  mivi::DevBuf snap;             // parameters / optimiser state / average before a loop with a grid-wide exchange (restored if it is lost)
  static constexpr int kMaxKids = 15;   // up to sixteen contexts (eight are used); at most FOUR graph branches (a forked graph with five branches crashed inside hipGraphLaunch,
                                       // hip::Graph::UpdateStreams, after a re-capture on ROCm 7.0's runtime; four is also the number of hardware queues)
  mivi_ctx *kids[kMaxKids] = {};
  int n_kids = 0;
  unsigned long long target_gen = 0, kid_gen = ~0ull;   // parent: bumped whenever a captured graph is invalidated; child: the generation it mirrors
  hipEvent_t ev_fork = nullptr, ev_join[kMaxKids] = {};
  bool dist_capture_refused = false;   // the sharded batch could not be captured into a hipGraph (RCCL route): issued eagerly from then on
  bool dist_direct = false;      // sharded estimates on the peer-to-peer route: the partial kernels store straight into the owners' staging areas
  bool dist_lane4 = false;       // pipelined sharded batches: the compute chain is lane-batched (four contexts per launch)
  void *value_sink = nullptr;    // ... and launch_value_only (a chain's closing value kernel) into the value sink
  void *eps_sink = nullptr;      // ... and launch_eps (a chain's first draw) into the eps sink
  void *stl_sink = nullptr;      // ... and launch_stl2 into stl_sink[lane_id]
  void *lane_sink = nullptr;     // lane-batched estimates: the launchers of the two second-generation kernels record into sink[lane_id] instead of launching
  int lane_id = 0;
  mivi::DevBuf kid_out[kMaxKids];       // value (16 bytes) + gradient of the child chains that do not hold the batch's last estimate
};

namespace mivi {

// kernels_meanfield.hip
struct ValueJob;
void launch_mf_main(mivi_ctx *c, const void *params, const RngArgs &rng, int M, int want_grad, const void *G,
                    const ValueIn &vin, const OutArgs &out, const ValueJob *prev = nullptr);
void launch_mf_sgd_loop(mivi_ctx *c, void *params, void *opt_state, uint64_t idx0, long long t0, int n_steps, int rule,
                        double eta, double clip_eps, double *hist, double *elbo, void *grad_out = nullptr, void *lane_scratch = nullptr,
                        void *value_last = nullptr);   // value_last: the last step's objective value as an element of T (written by the value kernel)
bool fr_rows_loop_ok(const mivi_ctx *c);    // kernels_fullrank_rows.hip: f32, n_mc <= 32, d <= 1024, diagonal-Gaussian target, not STL
size_t fr_rows_eps_bytes(const mivi_ctx *c, int n_steps);
size_t fr_rows_part_bytes(const mivi_ctx *c, int n_steps);
bool launch_fr_rows_loop(mivi_ctx *c, void *params, void *opt_state, uint64_t idx0, long long t0, int n_steps, int rule, double eta, double clip_eps,
                         float *eps_all, double *hist, double *elbo, void *value, const mivi_loop_t *gen = nullptr, double *part = nullptr);
bool lr_small_loop_ok(const mivi_ctx *c);   // kernels_logreg_small.hip: the logistic-regression target, d <= 64, (d - 1) n_mc <= 256, n (d - 1) n_mc <= 2^20: one workgroup per 2^14 of them
size_t lr_small_part_bytes(const mivi_ctx *c, int n_steps);   // 0: one workgroup, no exchange
bool launch_lr_small_loop(mivi_ctx *c, void *params, const mivi_loop_t &l, double *elbo, void *value, double *part);
bool mf_gen_loop_ok(const mivi_ctx *c, int rule);   // kernels_meanfield.hip: every rule x operator x averager, mean-field + diagonal-Gaussian target
size_t mf_gen_loop_scratch_bytes(const mivi_ctx *c, int n_steps);
bool launch_mf_gen_loop(mivi_ctx *c, void *params, const mivi_loop_t &l, double *hist, double *elbo, char *scratch);
bool fr_small_loop_ok(const mivi_ctx *c);   // kernels_fullrank_small.hip: d <= 32, n_mc <= 64, diagonal-Gaussian target
void launch_fr_small_loop(mivi_ctx *c, void *params, void *opt_state, uint64_t idx0, long long t0, int n_steps, int rule, double eta,
                          double clip_eps, double *elbo, void *value, const mivi_loop_t *gen = nullptr);
bool launch_mf_funnel_sgd_loop(mivi_ctx *c, void *params, void *opt_state, uint64_t idx0, long long t0, int n_steps, int rule, double eta,
                               double clip_eps, double *hist, unsigned *sync, void *pub, void *gtmp, double *elbo, void *value);
void launch_mf_funnel_loop(mivi_ctx *c, const void *params, uint64_t idx0, int n_steps, double *hist, double *elbo, void *scratch,
                           void *value, void *grad, void *lane_scratch, void *e0_tab = nullptr);
int mf_loop_lanes(const mivi_ctx *c, int n_steps);   // estimate lanes of the launch-free batches (lane_scratch: lanes * 2 d elements)
void launch_sample_mf(mivi_ctx *c, const void *params, const RngArgs &rng, int M, void *Z, void *eps, int ld_eps,
                      double *he_part);

// kernels_fullrank.hip
int launch_eps(mivi_ctx *c, const RngArgs &rng, int M);   // returns the number of he_part entries the draw leaves
void launch_fr_sample(mivi_ctx *c, const void *params, int M, int fused_target, void *Z, const ValueJob *prev = nullptr);
void launch_fr_vjp(mivi_ctx *c, const void *params, int M, const OutArgs &out, const EpsJob *next = nullptr,
                   const ValueJob *self = nullptr, const FusedUpdate *upd = nullptr);
int fr_ld_blocks(const mivi_ctx *c);
bool f64_valu();   // MIVI_F64_VALU=1: keep the f64 full-rank tiles on the vector ALU
void prepare_tables(mivi_ctx *c, int M);   // build + upload the MFMA work tables (no-op for f64 / mean-field)
void launch_fr_dense_target(mivi_ctx *c, int M, int want_grad);
void launch_rt_from_z(mivi_ctx *c, int M);
void launch_fr_stl(mivi_ctx *c, const void *params, int M, const void *rhs = nullptr, void *out = nullptr);
void launch_stein_outer(mivi_ctx *c, int M, void *A, double *gsum, int first, double scale);
void launch_stein_finish(mivi_ctx *c, double n, const double *gsum, const double *ell_sum, const void *ell_single, void *grad, void *logpi_avg);
void launch_stein_gsum(mivi_ctx *c, int M, double *gsum, int first);   // gsum (+)= G 1 (the second-order branch: no eps G^T product)
void launch_const_hess(mivi_ctx *c, void *hess);                      // hess = the built-in Gaussian targets' constant Hessian
// kernels_hess2.hip: the sample average of the Hessians of the built-in logistic-regression / funnel targets (second-order branch)
bool target_has_hess2(const mivi_ctx *c);
size_t target_hess2_bytes(const mivi_ctx *c);
bool target_hess2_begin(mivi_ctx *c);
void target_hess2_accumulate(mivi_ctx *c, int Mc);
void target_hess2_finish(mivi_ctx *c, int n_samples, void *hess);
int fr_sample_blocks(const mivi_ctx *c, int M);
int fr_dense_blocks(const mivi_ctx *c, int M);
int eps_blocks(const mivi_ctx *c, int M);

// kernels_fullrank_lds.hip (f32, d and M multiples of 64)
enum LdsReduceMode : int { R_DIAG = 0, R_DENSE_R = 1, R_DENSE_G = 2, R_PLAIN = 3 };
bool lds_path_shape_ok(const mivi_ctx *c, int M);
bool lds_prepare(mivi_ctx *c, int M);          // work lists + slab buffer for M samples per launch (false: allocation failed)
int lds_ld_blocks(const mivi_ctx *c);
int lds_eps_blocks(const mivi_ctx *c, int M);
void launch_lds_prod32(mivi_ctx *c, const void *params, int M, bool dense, int mode, void *Z, const EpsJob *next, bool want_ld,
                       bool with_dinv = false);   // with_dinv: trailing workgroups invert the 64x64 diagonal blocks of C (STL)
int lds_prod32_tiles(const mivi_ctx *c, int M);
void launch_lds_prod64(mivi_ctx *c, const void *params, int M, bool dense, int mode, void *Z, const EpsJob *next, bool want_ld);
int lds_prod64_tiles(const mivi_ctx *c, int M);
bool lds_use_prod64(const mivi_ctx *c, int M);   // large shapes: unsplit 64 x 64 tiles
int lds_prod32_eps_blocks(const mivi_ctx *c, int M);
bool lds_use_prod32(const mivi_ctx *c, int M);
struct LaneSink;
LaneSink *lane_sinks_alloc(int n);
void lane_sinks_free(LaneSink *s);
void lane_sink_reset(LaneSink *s, int lane);
int lane_sink_counts(const LaneSink *s, int lane);                       // products recorded * 16 + VJPs recorded
bool launch_lanes_prod(mivi_ctx *c, LaneSink *s, int lanes, int which);   // one launch for all lanes (blockIdx.y = lane)
bool launch_lanes_vjp(mivi_ctx *c, LaneSink *s, int lanes);
struct ValueSink;
ValueSink *value_sink_alloc();
void value_sink_free(ValueSink *s);
void launch_lanes_value(mivi_ctx *c, const void *params, ValueSink *s);   // the recorded closing value kernels as one launch
struct EpsSink;
EpsSink *eps_sink_alloc();
void eps_sink_free(EpsSink *s);
void eps_sink_reset(EpsSink *s);
void launch_lanes_eps(mivi_ctx *c, EpsSink *s, int lanes);               // the lanes' recorded first draws as one launch
struct StlSink;
StlSink *stl_sinks_alloc(int n);
void stl_sinks_free(StlSink *s);
void stl_sink_reset(StlSink *s, int lane);
int stl_sink_count(const StlSink *s, int lane);
bool launch_lanes_stl(mivi_ctx *c, StlSink *s, int lanes, bool with_F);  // one solve launch with all lanes' jobs + one combining product
bool lds_bf16x3();   // products on the bf16 matrix cores (three-way exact operand split); MIVI_FR_F32MFMA=1 turns it off
void launch_lds_vjp(mivi_ctx *c, const void *params, int M, const OutArgs &out, const ValueJob *self, const FusedUpdate *upd);
bool lds_stein_ok(const mivi_ctx *c, int M);
void launch_lds_stein_outer(mivi_ctx *c, int M, void *A, double *gsum, int first, double scale, double n, void *grad, void *logpi,
                            const ValueJob *self);
void invalidate_graph(mivi_ctx *c);            // api_core.hip: drop the cached hipGraphExec and the eps speculation

// kernels_fullrank_batch.hip (f32, diagonal-Gaussian target, d % 128 == 0, M % 128 == 0): L estimates at the same parameters per launch
// every workgroup of `grid` resident at once?  (loops whose workgroups exchange partials every step by spin-wait: checked at launch, never assumed)
bool grid_resident(const mivi_ctx *c, const void *kernel, int block, size_t dyn_lds, long long grid);
bool fb_shape_ok(const mivi_ctx *c, int M);
bool fb_whole_tiles(const mivi_ctx *c, int M);        // d and M multiples of 128: no padding (what the dense target, the STL term and the sharded batches need)
mivi_status_t fb_objective(mivi_ctx *c, const void *params, uint64_t idx, int lanes, int entropy, void *values);   // api_batch.hip: lanes x n_mc samples of estimate idx on the batch engine (MIVI_ERR_UNSUPPORTED: not an engine configuration)
const FbTab *fb_prepare(mivi_ctx *c, int M, int L);   // work tables for L lanes (nullptr: allocation failed)
size_t fb_plane_words(const mivi_ctx *c, int M);      // 4-byte words of one lane's operand planes (eps in one orientation, W)
size_t fb_cplane_words(const mivi_ctx *c);            // ... of tril(C)'s
void fb_launch_eps(mivi_ctx *c, const FbStep &s, bool with_cplanes, hipStream_t stream);   // a step's draws (+ tril(C)'s planes, once per call)
void fb_launch_pplanes(mivi_ctx *c, hipStream_t stream);
void fb_launch_tplanes(mivi_ctx *c, hipStream_t stream);
size_t fb_part_len(const mivi_ctx *c);                                     // floats of a lane's partial vector on the engine (a multiple of four)
void fb_launch_finalize_parts(mivi_ctx *c, const FbStep &s, hipStream_t stream);   // (summed) partial vectors -> the lanes' values and gradients
void fb_launch_compute(mivi_ctx *c, const FbStep &s, hipStream_t stream, int which = 15);  // product(s) + target -> VJP + values (which: bit 0 product, 1 VJP, 2 the dense target's product, 3 the sticking-the-landing product)

// kernels_stl.hip (f32, d in {256, 512, 1024, 2048}, M % 32 == 0)
bool stl2_shape_ok(const mivi_ctx *c, int M);
void launch_stl2(mivi_ctx *c, const void *params, int M, bool dinv_done = false, const void *rhs = nullptr, void *out = nullptr,
                 bool overwrite = false);   // rhs (ld dP) / out (ld d) default to eps / W; overwrite: out = X instead of out += X   // W += C^-T eps: dinv64 -> solve (lower half) -> update -> solve (upper half)

// kernels_targets.hip
void launch_col_target(mivi_ctx *c, int M, int want_grad);
void launch_bij_forward(mivi_ctx *c, int M);                        // Z <- binv(Z) in place, bij_ld[m] = sum_exp eta
void launch_bij_backward(mivi_ctx *c, int M, int want_grad, bool add_to_ell);   // G <- J' G + 1_exp; ell[m] += bij_ld[m]
bool launch_logreg_target(mivi_ctx *c, int M, int want_grad);   // false: scratch allocation failed
int logreg_kernel_bits(const mivi_ctx *c, int M);         // mivi_logreg_kernels
bool logreg_uses_mfma(const mivi_ctx *c, int M);          // the matrix-core route (needs Z^T staged in RT)
bool logreg_reserve(mivi_ctx *c, int M);                        // size the scratch ahead of a graph capture
bool logreg_prepare_f32(mivi_ctx *c);   // row-major padded copy of X for the MFMA route (false: allocation failed)

// kernels_p2p.hip: phases bit 0 push, 1 reduce + finalise, 2 unpack (7 = the whole exchange in one launch)
void launch_p2p_handover4(mivi_ctx *c, unsigned *ready, unsigned ready_val, const unsigned *const *freed, const unsigned *freed_min, int n);
void launch_p2p_exchange(mivi_ctx *c, const void *params, const void *const *P, int ring, void *value, void *grad, int phases, int lane, int lanes,
                         int count, const unsigned *ready, unsigned *freed, bool direct = false);   // one lane: estimates lane, lane + lanes, ... < count; P[t % ring]
void launch_p2p_handover(mivi_ctx *c, unsigned *ready, unsigned ready_val, const unsigned *freed, unsigned freed_min);

// kernels_update.hip
void launch_finalize(mivi_ctx *c, const void *params, const void *partials, void *value, void *grad);
void launch_finalize_slice(mivi_ctx *c, const void *params, const void *sum, long long g0, long long n, void *fin);
void launch_unpack_final(mivi_ctx *c, const void *fin, void *value, void *grad);
void launch_prox(mivi_ctx *c, void *params, double stepsize, const void *dog_state, int dog_kind);
void launch_poly_average(mivi_ctx *c, void *avg, const void *params, double avg_eta, const long long *t_ptr, long long t_base);
void launch_dog_update(mivi_ctx *c, void *params, const void *grad, void *state, int kind);
bool launch_dog_update_fused(mivi_ctx *c, void *params, const void *grad, void *state, int kind, double clip_eps, void *avg,
                             double avg_eta, const long long *t_ptr, long long t_base);
void launch_logreg_gather(mivi_ctx *c, int64_t b);   // batch rows lr_idx[0..b) of the full data set -> lr_Xsub / lr_ysub / lr_Xrm_sub
void launch_value_only(mivi_ctx *c, const void *params, const ValueIn &vin, const OutArgs &out);   // on c->stream
void launch_clip(mivi_ctx *c, void *params, double epsilon);
void launch_descent(mivi_ctx *c, void *params, const void *grad, double eta, double clip_eps = NAN);   // NaN: no ClipScale
void launch_adam(mivi_ctx *c, void *params, const void *grad, void *state, const int64_t *t_ptr, int64_t t_base,
                 double eta, double b1, double b2, double eps, double clip_eps = NAN);
void launch_cocob(mivi_ctx *c, void *params, const void *grad, void *state, double alpha, double clip_eps = (double)NAN);
void launch_bump(mivi_ctx *c, uint64_t *ctr, uint64_t by);

}  // namespace mivi
