// The gradient exchange of the sharded estimator written for xGMI (no counterpart in the reference, which is a
// single task: src/algorithms/repgradelbo.jl:84-86 is a mean over samples, so what crosses GPUs is a SUM of partial vectors).
//
// RCCL's ring all-reduce of the 2.1 MB north-star partial vector costs 2 (R - 1) dependent hops (50-70 us of latency at R = 8 against
// 14 us of compute).  xGMI is point to point -- every GPU has a direct link to every other one -- so the exchange is ONE kernel per
// rank with two one-hop phases, all seven links busy in both:
//
//   phase 1  push        every rank stores slice s of its partial vector straight into rank s's staging area (peer stores)
//   phase 2  reduce      rank s sums the R contributions of ITS slice in rank order (f64), finalises it (-1/M, entropy diagonal
//                        terms; the owner of the two scalars also assembles the objective value) and stores the packed final
//                        slice into EVERY rank's final area (peer stores)
//   phase 3  unpack      every rank expands the packed final vector into value + dense gradient (exact zeros above the diagonal)
//
// Every slice is finalised by exactly one rank from contributions summed in rank order: all ranks hold bit-identical results and the
// sum is independent of arrival order.
//
// Synchronisation is in the DATA ("LL" layout, as RCCL's low-latency protocol): every 32-bit payload word travels as an 8-byte pair
// (word, epoch) written by ONE store, so a reader that finds the epoch of this exchange in a pair has the word -- no separate flag,
// no store acknowledge to wait for, no flag poll: a phase costs one (repeated until complete) load round trip instead of
// {store acknowledge, flag round trip, data round trip} (measured on one GPU, DESIGN.md 7: the flag version of this kernel 29-40 us per exchange, this one 33 us).  Twice the
// bytes; the exchange is latency bound (2 MB).
//
// Slices are cut into G chunks; workgroup g of every rank handles chunk g of every slice in all three phases, so workgroup g only
// ever depends on workgroup g of its peers: no grid-wide barrier, no dependency cycle (phase 1 never waits), and the areas can be
// double-buffered by epoch parity without acknowledgements -- workgroup g of rank a reaches epoch e + 2 only after it unpacked
// epoch e + 1, which needed workgroup g of every owner to have reduced epoch e + 1, hence to be done with epoch e.
//
// The kernel is PERSISTENT over a batch of `count` estimates (mivi_estimate_gradient_dist_n): it runs on its own stream beside the
// compute chain and is handed each partial vector through two device words -- `ready` (set by the compute chain when the partial
// vector of estimate t is complete) and `freed[slot]` (bumped by every workgroup once it has read its part of the vector in ring slot `slot`; the compute
// chain checks it before that ring slot is overwritten).  No stream events, no graph fork / join per estimate.  A batch is served by
// `lanes` such kernels (estimate t by lane t mod lanes, each lane with its own areas and epochs): an exchange is a chain of memory round
// trips, two in flight hide each other's waits.
//
// Memory: one fine-grained allocation per rank, mapped into its peers through HIP IPC (mivi_p2p_export / mivi_p2p_attach).  Every access
// to it is system scope (sc0 sc1): stores write through, loads never hit a line an XCD's L2 kept from two epochs ago.  No cache-wide
// fence is issued (the compute kernels running beside the exchange keep their L2-resident operands).  Every spin is bounded: a lost
// peer sets status bit 8 and the kernel leaves (the host reports it; nothing hangs).
#include "device_common.h"

namespace mivi {

constexpr int kP2PLanes = 2, kP2PRing = 4;

struct P2PTable {   // device resident: where every rank's exchange areas are mapped in THIS process, per lane
  unsigned long long *stage[kP2PLanes][8];   // [2][R][n W] pairs: stage[s] = rank s's staging area (contribution of rank `src` to slice s at [parity][src])
  unsigned long long *fin[kP2PLanes][8];     // [2][R n W] pairs : rank s's packed final vector
};

template <typename T>
struct P2PArgs {
  int d, family, ent_kind, M_total;
  long long L, n, cn;        // partial length; slice length (multiple of 4); chunk length (multiple of 4)
  int rank, world, G, vs;    // vs = the rank whose slice holds the two scalars (sum ell, sum 0.5|eps|^2)
  const P2PTable *tab;
  unsigned *ctr;             // this lane's counters: [0] exchanges completed, [1] exit ticket
  const T *P[kP2PRing];      // this rank's partial vectors: estimate t of the batch sits in P[t % ring], zero padded to world * n
  int ring;
  const T *params;
  T *value, *grad;           // results of the batch's LAST estimate
  T *scratch;                // value (4 slots) + gradient of the estimates before it (this lane's own: every estimate is fully written)
  int *status;
  int phases;                // bit 0 push, bit 1 reduce, bit 2 unpack (all three = the exchange; single phases: host-sequenced tests, count = 1)
  int spin_budget;
  int lane, lanes, count;    // this launch serves estimates t = lane, lane + lanes, ... < count
  const unsigned *ready;     // batch hand-over (nullptr: the partial vector is complete at launch): estimate t may start when *ready >= t + 1
  unsigned *freed;           // [ring]: += 1 by every chunk workgroup once its part of the vector in that ring slot has been read
};

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void store16_sys(void *p, u32x4_t r) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ void store8_sys(void *p, u32x2_t r) {
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(r) : "memory");
}
// eight independent 16-byte system-scope loads in flight, ONE wait (such a load is a full memory round trip: issued one per loop
// iteration the exchange was a chain of ~1.5 us latencies)
__device__ __forceinline__ void ld16x8_sys(const void *const (&p)[8], u32x4_t (&o)[8]) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc0 sc1\n\t"
      "global_load_dwordx4 %1, %9, off sc0 sc1\n\t"
      "global_load_dwordx4 %2, %10, off sc0 sc1\n\t"
      "global_load_dwordx4 %3, %11, off sc0 sc1\n\t"
      "global_load_dwordx4 %4, %12, off sc0 sc1\n\t"
      "global_load_dwordx4 %5, %13, off sc0 sc1\n\t"
      "global_load_dwordx4 %6, %14, off sc0 sc1\n\t"
      "global_load_dwordx4 %7, %15, off sc0 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
      : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
      : "memory");
}
__device__ __forceinline__ unsigned long long ld8_sys(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Eight LL units (16 bytes = two (word, epoch) pairs each) until every pair the caller needs carries `epoch`.  need[u]: bit 0 / 1 =
// the first / second pair of unit u matters (0: the slot is padding).  Bounded: false = a word never arrived.
__device__ __forceinline__ bool ll_load8(const void *const (&p)[8], const unsigned (&need)[8], unsigned epoch, int budget, u32x4_t (&o)[8]) {
  for (;;) {
    ld16x8_sys(p, o);
    bool ok = true;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if ((need[u] & 1u) && o[u][1] != epoch) ok = false;
      if ((need[u] & 2u) && o[u][3] != epoch) ok = false;
    }
    if (ok) return true;
    if (--budget <= 0) return false;
    __builtin_amdgcn_s_sleep(2);
  }
}
// one LL pair (bounded)
__device__ __forceinline__ bool ll_load1(const unsigned long long *p, unsigned epoch, int budget, unsigned &word) {
  for (;;) {
    const unsigned long long v = ld8_sys(p);
    if ((unsigned)(v >> 32) == epoch) { word = (unsigned)v; return true; }
    if (--budget <= 0) { word = 0; return false; }
    __builtin_amdgcn_s_sleep(2);
  }
}

template <typename T> struct Words;
template <> struct Words<float> {
  static constexpr int W = 1;
  static __device__ __forceinline__ float make(unsigned w0, unsigned) { return __builtin_bit_cast(float, w0); }
  static __device__ __forceinline__ void split(float x, unsigned &w0, unsigned &w1) { w0 = __builtin_bit_cast(unsigned, x); w1 = 0u; }
};
template <> struct Words<double> {
  static constexpr int W = 2;
  static __device__ __forceinline__ double make(unsigned w0, unsigned w1) {
    return __builtin_bit_cast(double, (unsigned long long)w0 | ((unsigned long long)w1 << 32));
  }
  static __device__ __forceinline__ void split(double x, unsigned &w0, unsigned &w1) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    w0 = (unsigned)b;
    w1 = (unsigned)(b >> 32);
  }
};

// packed index (>= d, full-rank) -> (column j, row i) of the lower triangle: e2 = j d - j (j - 1) / 2 + (i - j)
__device__ __forceinline__ void packed_col_row(long long gi, int d, long long &j, long long &i) {
  const long long e2 = gi - d;
  const double b = 2.0 * d + 1.0;
  j = (long long)((b - sqrt(b * b - 8.0 * (double)e2)) * 0.5);
  if (j < 0) j = 0;
  if (j > d - 1) j = d - 1;
  while (j > 0 && j * d - (j * (j - 1)) / 2 > e2) --j;
  while (j + 1 < d && (j + 1) * d - ((j + 1) * j) / 2 <= e2) ++j;
  i = j + (e2 - (j * d - (j * (j - 1)) / 2));
}

// packed final entry -> its finalised value: -(1/M) sum, the diagonal entries of the scale carry the entropy term
template <typename T>
__device__ __forceinline__ double p2p_finalise(const P2PArgs<T> &a, long long gi, double sum, double invM, double direct) {
  const int d = a.d;
  double v = -sum * invM;
  if (gi < d) return v;
  if (a.family == MIVI_MEANFIELD) return v - direct / (double)a.params[gi];
  long long j, i;
  packed_col_row(gi, d, j, i);
  if (i == j) v -= direct / (double)a.params[d + (size_t)j * d + j];
  return v;
}

// phase 2 for R <= K sources (K in {1, 2, 4, 8}): 8 / K units of this thread x K sources per batch of eight loads.
// A unit = 16 bytes = two payload words = EPU elements (float: 2, double: 1).
template <typename T, int K, int NT>
__device__ __forceinline__ bool p2p_reduce_chunk(const P2PArgs<T> &a, const P2PTable &tb, int p, unsigned epoch, long long c0, long long clen) {
  constexpr int W = Words<T>::W, EPU = 2 / W, NV = 8 / K;
  const int tid = threadIdx.x, R = a.world;
  const long long n = a.n, tri_end = a.L - 2;
  const double invM = 1.0 / (double)a.M_total, direct = direct_entropy_coeff(a.ent_kind);
  const unsigned long long *st = tb.stage[a.lane][a.rank] + ((size_t)(p * R) * n + c0) * W;   // + src * n * W + 2 * unit
  const long long g0 = (long long)a.rank * n + c0;
  const long long units = clen / EPU;
  bool all_ok = true;
  for (long long base = 0; base < units; base += (long long)NV * NT) {
    const void *ptr[8];
    unsigned need[8];
    u32x4_t raw[8];
    long long un[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      un[v] = base + (long long)v * NT + tid;
      const bool in = un[v] < units;
      const long long uc = in ? un[v] : 0;
      // (elements at or beyond tri_end -- the two scalars, the padding -- are never pushed as final values and are not needed here
      //  either: the scalars are summed by the value workgroup)
      unsigned nd = 0;
      if (in) {
        const long long gi = g0 + uc * EPU;
        if (EPU == 2) nd = (gi < tri_end ? 1u : 0u) | (gi + 1 < tri_end ? 2u : 0u);
        else nd = gi < tri_end ? 3u : 0u;
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
        ptr[v * K + k] = st + (size_t)(k < R ? k : R - 1) * n * W + 2 * uc;
        need[v * K + k] = k < R ? nd : 0u;
      }
    }
    if (!ll_load8(ptr, need, epoch, a.spin_budget, raw)) all_ok = false;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (un[v] >= units) continue;
      const long long gi0 = g0 + un[v] * EPU;
      if (gi0 >= tri_end) continue;
      double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) {   // rank order: the sum does not depend on who arrived first
        if (k < R) {
          unsigned wd[4];
          __builtin_memcpy(wd, &raw[v * K + k], 16);   // {word, epoch, word, epoch}
          if (EPU == 2) {
            acc0 += (double)__builtin_bit_cast(float, wd[0]);
            acc1 += (double)__builtin_bit_cast(float, wd[2]);
          } else {
            acc0 += Words<double>::make(wd[0], wd[2]);
          }
        }
      }
      unsigned ow0, ow1, dummy;
      bool both = true;
      if (EPU == 2) {
        Words<float>::split((float)p2p_finalise(a, gi0, acc0, invM, direct), ow0, dummy);
        both = gi0 + 1 < tri_end;
        ow1 = 0u;
        if (both) Words<float>::split((float)p2p_finalise(a, gi0 + 1, acc1, invM, direct), ow1, dummy);
      } else {
        Words<double>::split(p2p_finalise(a, gi0, acc0, invM, direct), ow0, ow1);
      }
      const size_t off = ((size_t)p * R * n + (size_t)gi0) * W;   // pair index inside a final area
      if (both) {
        const u32x4_t ov = {ow0, epoch, ow1, epoch};
        for (int k = 0; k < R; ++k) store16_sys(tb.fin[a.lane][(a.rank + 1 + k) % R] + off, ov);
      } else {   // (the last element below the scalars at an even index: one pair)
        const u32x2_t ov = {ow0, epoch};
        for (int k = 0; k < R; ++k) store8_sys(tb.fin[a.lane][(a.rank + 1 + k) % R] + off, ov);
      }
    }
  }
  return all_ok;
}

template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_p2p_exchange(P2PArgs<T> a) {   // (<= 128 VGPRs: the spinning workgroups must leave the compute kernels their registers)
  constexpr int NT = 256, W = Words<T>::W, EPU = 2 / W;
  __shared__ int sh_ok;
  __shared__ double red[4];
  const int tid = threadIdx.x, g = blockIdx.x, R = a.world, G = a.G, ln = a.lane;
  unsigned epoch = a.ctr[0];
  const long long n = a.n;
  const P2PTable &tb = *a.tab;
  const bool value_wg = (g == G);
  const long long c0 = value_wg ? 0 : (long long)g * a.cn;
  const long long clen = value_wg ? 0 : ((c0 + a.cn <= n ? a.cn : (n > c0 ? n - c0 : 0)));
  const long long tri_end = a.L - 2;
  const int d = a.d;
  bool lost = false;

  for (int t = a.lane; t < a.count; t += a.lanes) {
    ++epoch;
    const int p = (int)(epoch & 1u);
    const int slot = t % a.ring;
    const T *P = a.P[slot];
    const bool last = (t == a.count - 1);
    T *out_v = last ? a.value : a.scratch, *out_g = last ? a.grad : a.scratch + 4;
    if (a.ready) {   // hand-over from the compute chain: the partial vector of estimate t is complete
      if (tid == 0) {
        int budget = a.spin_budget;
        sh_ok = 1;
        while ((int)__hip_atomic_load(a.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < t + 1) {
          if (--budget <= 0) { sh_ok = 0; break; }
          __builtin_amdgcn_s_sleep(4);
        }
      }
      __syncthreads();
      if (!sh_ok) lost = true;
      __syncthreads();
    }

    // ---- phase 1: push chunk g of every slice to its owner (LL pairs) -----------------------------------------------------------------
    if ((a.phases & 1) && !value_wg) {
      const long long vecs = clen * W / 4;   // 16-byte vectors of payload words in a chunk (clen is a multiple of 4)
      for (long long base = 0; base < vecs; base += 8LL * NT) {
        for (int k = 0; k < R; ++k) {
          const int s = (a.rank + 1 + k) % R;   // start with the neighbour: the links fill evenly, the local copy comes last
          const unsigned *src = (const unsigned *)(P + (size_t)s * n + c0);
          unsigned long long *dst = tb.stage[ln][s] + ((size_t)(p * R + a.rank) * n + c0) * W;
          const void *ptr[8];
          u32x4_t r[8];
          long long vi[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            vi[u] = base + (long long)u * NT + tid;
            ptr[u] = src + 4 * (vi[u] < vecs ? vi[u] : 0);
          }
          ld16x8_sys(ptr, r);   // (system scope: the vector was written by the compute kernels' XCDs and this kernel never restarts)
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (vi[u] >= vecs) continue;
            const u32x4_t lo = {r[u][0], epoch, r[u][1], epoch}, hi = {r[u][2], epoch, r[u][3], epoch};
            store16_sys(dst + 4 * vi[u], lo);
            store16_sys(dst + 4 * vi[u] + 2, hi);
          }
        }
      }
      if (a.freed) {   // this workgroup is done reading the partial vector in this ring slot
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(a.freed + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }

    // ---- phase 3a (needs nothing from anybody): exact zeros above the diagonal of the dense gradient -------------------------------------
    if ((a.phases & 4) && !value_wg && a.family == MIVI_FULLRANK) {
      T *gc = out_g + d;
      for (int j = g + 1; j < d; j += G)
        for (int i = tid; i < j; i += NT) gc[(size_t)j * d + i] = T(0);
    }

    // ---- phase 2: reduce + finalise chunk g of MY slice, push the final chunk to every rank -------------------------------------------
    if ((a.phases & 2) && !value_wg) {
      bool ok;
      if (R == 1) ok = p2p_reduce_chunk<T, 1, NT>(a, tb, p, epoch, c0, clen);
      else if (R == 2) ok = p2p_reduce_chunk<T, 2, NT>(a, tb, p, epoch, c0, clen);
      else if (R <= 4) ok = p2p_reduce_chunk<T, 4, NT>(a, tb, p, epoch, c0, clen);
      else ok = p2p_reduce_chunk<T, 8, NT>(a, tb, p, epoch, c0, clen);
      if (!ok) lost = true;
    }
    if ((a.phases & 2) && value_wg && a.rank == a.vs) {   // the objective value: sum ell, sum 0.5|eps|^2 of all ranks + the parameter-only terms
      double s_ld = 0.0, bad = 0.0;
      for (int i = tid; i < d; i += NT) {
        const double c = (double)(a.family == MIVI_MEANFIELD ? a.params[d + i] : a.params[d + (size_t)i * d + i]);
        if (!(c > 0.0)) bad = 1.0;
        s_ld += log(c);
      }
      s_ld = block_sum<double, NT>(s_ld, red);
      bad = block_sum<double, NT>(bad, red);
      if (tid == 0) {
        const long long o0 = tri_end - (long long)a.vs * n;   // offset of the first scalar inside my slice
        double sc[2] = {0.0, 0.0};
        for (int src = 0; src < R; ++src)
          for (int q = 0; q < 2; ++q) {
            unsigned w[2] = {0u, 0u};
            for (int h = 0; h < W; ++h)
              if (!ll_load1(tb.stage[ln][a.rank] + ((size_t)(p * R + src) * n + o0 + q) * W + h, epoch, a.spin_budget, w[h])) lost = true;
            sc[q] += (double)Words<T>::make(w[0], w[1]);
          }
        const double Mt = (double)a.M_total;
        const double ent = (ent_is_closed(a.ent_kind) ? 0.5 * d * (1.0 + kLog2Pi) : sc[1] / Mt + 0.5 * d * kLog2Pi) + s_ld;
        const double value = -(sc[0] / Mt + ent);
        int stt = 0;
        if (!isfinite(value)) stt |= 1;
        if (bad > 0.0) stt |= 2;
        unsigned wv[2], ws[2];
        Words<T>::split((T)value, wv[0], wv[1]);
        Words<T>::split((T)stt, ws[0], ws[1]);
        for (int s = 0; s < R; ++s) {
          unsigned long long *dst = tb.fin[ln][s] + ((size_t)p * R * n + tri_end) * W;
          for (int h = 0; h < W; ++h) {
            const u32x2_t v1 = {wv[h], epoch}, v2 = {ws[h], epoch};
            store8_sys(dst + h, v1);
            store8_sys(dst + W + h, v2);
          }
        }
      }
    }

    // ---- phase 3: unpack chunk g of every final slice ------------------------------------------------------------------------------------
    if (a.phases & 4) {
      const unsigned long long *fin = tb.fin[ln][a.rank] + (size_t)p * R * n * W;
      if (value_wg) {
        if (tid == 0) {
          unsigned wv[2] = {0u, 0u}, ws[2] = {0u, 0u};
          for (int h = 0; h < W; ++h) {
            if (!ll_load1(fin + (size_t)tri_end * W + h, epoch, a.spin_budget, wv[h])) lost = true;
            if (!ll_load1(fin + (size_t)(tri_end + 1) * W + h, epoch, a.spin_budget, ws[h])) lost = true;
          }
          *out_v = Words<T>::make(wv[0], wv[1]);
          const int stt = (int)Words<T>::make(ws[0], ws[1]);
          if (stt && a.status) atomicOr(a.status, stt);
        }
      } else {
        const long long units = clen / EPU, total = units * R;   // unit index x = s * units + u
        for (long long base = 0; base < total; base += 8LL * NT) {
          const void *ptr[8];
          unsigned need[8];
          u32x4_t raw[8];
          long long gi0[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const long long x = base + (long long)u * NT + tid;
            const bool in = x < total;
            const long long s = in ? x / units : 0, uu = in ? x % units : 0;
            gi0[u] = s * n + c0 + uu * EPU;
            ptr[u] = fin + (size_t)gi0[u] * W;
            need[u] = 0;
            if (in) {
              if (EPU == 2) need[u] = (gi0[u] < tri_end ? 1u : 0u) | (gi0[u] + 1 < tri_end ? 2u : 0u);
              else need[u] = gi0[u] < tri_end ? 3u : 0u;
            }
          }
          if (!ll_load8(ptr, need, epoch, a.spin_budget, raw)) lost = true;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (!need[u]) continue;
            unsigned wd[4];
            __builtin_memcpy(wd, &raw[u], 16);   // {word, epoch, word, epoch}
            T x0, x1 = T(0);
            if (EPU == 2) {
              x0 = Words<T>::make(wd[0], 0u);
              x1 = Words<T>::make(wd[2], 0u);
            } else {
              x0 = Words<T>::make(wd[0], wd[2]);
            }
#pragma unroll
            for (int c = 0; c < EPU; ++c) {
              const long long gi = gi0[u] + c;
              if (gi >= tri_end) continue;
              long long di = gi;
              if (a.family != MIVI_MEANFIELD && gi >= d) {
                long long j, i;
                packed_col_row(gi, d, j, i);
                di = d + j * d + i;
              }
              out_g[di] = c == 0 ? x0 : x1;
            }
          }
        }
      }
    }
  }
  if (lost && tid == 0 && a.status) atomicOr(a.status, 8);

  // ---- exit ticket: the last workgroup out publishes the lane's epoch (every workgroup has read it by then) ------------------------------------
  if (a.phases & 4) {
    __syncthreads();
    if (tid == 0) {
      const unsigned tk = __hip_atomic_fetch_add(a.ctr + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (tk == (unsigned)G) {
        __hip_atomic_store(a.ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.ctr, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// hand-over words of the pipelined batch on the COMPUTE chain (one thread): announce a complete partial vector (ready = ready_val), then
// hold the chain until the exchange has read the ring slot the NEXT estimate's kernels are going to overwrite (*freed >= freed_min)
__global__ void k_p2p_handover(unsigned *ready, unsigned ready_val, const unsigned *freed, unsigned freed_min, int budget, int *status) {
  if (ready) __hip_atomic_store(ready, ready_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  if (freed) {
    while ((int)(__hip_atomic_load(freed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - freed_min) < 0) {
      if (--budget <= 0) { if (status) atomicOr(status, 8); break; }
      __builtin_amdgcn_s_sleep(4);
    }
  }
}

// host side -------------------------------------------------------------------------------------------------------------------------
void launch_p2p_handover(mivi_ctx *c, unsigned *ready, unsigned ready_val, const unsigned *freed, unsigned freed_min) {
  hipLaunchKernelGGL(k_p2p_handover, dim3(1), dim3(1), 0, c->stream, ready, ready_val, freed, freed_min, c->p2p_spin, (int *)c->status.p);
}

// one lane of the exchange on c->stream: estimates t = lane, lane + lanes, ... < count with partial vectors P[t % ring]
void launch_p2p_exchange(mivi_ctx *c, const void *params, const void *const *P, int ring, void *value, void *grad, int phases, int lane, int lanes,
                         int count, const unsigned *ready, unsigned *freed) {
  auto fill = [&](auto &a) {
    a.d = c->cfg.d; a.family = c->cfg.family; a.ent_kind = c->cfg.entropy; a.M_total = c->M_total;
    a.L = mivi_partials_len(c); a.n = c->p2p_n; a.cn = c->p2p_cn;
    a.rank = c->p2p_rank; a.world = c->p2p_world; a.G = c->p2p_G; a.vs = c->p2p_vs;
    a.tab = (const P2PTable *)c->p2p_tab.p;
    a.ctr = (unsigned *)c->p2p_ctr.p + 16 * lane;
    a.status = (int *)c->status.p;
    a.phases = phases;
    a.spin_budget = c->p2p_spin;
    a.lane = lane; a.lanes = lanes; a.count = count;
    a.ring = ring;
    a.ready = ready;
    a.freed = freed;
  };
  const size_t plen4 = (size_t)mivi_params_len(c) + 4;
  if (c->cfg.dtype == MIVI_F32) {
    P2PArgs<float> a{};
    fill(a);
    for (int k = 0; k < ring; ++k) a.P[k] = (const float *)P[k];
    a.params = (const float *)params; a.value = (float *)value; a.grad = (float *)grad;
    a.scratch = (float *)c->p2p_scratch.p + plen4 * lane;
    hipLaunchKernelGGL(k_p2p_exchange<float>, dim3(c->p2p_G + 1), dim3(256), 0, c->stream, a);
  } else {
    P2PArgs<double> a{};
    fill(a);
    for (int k = 0; k < ring; ++k) a.P[k] = (const double *)P[k];
    a.params = (const double *)params; a.value = (double *)value; a.grad = (double *)grad;
    a.scratch = (double *)c->p2p_scratch.p + plen4 * lane;
    hipLaunchKernelGGL(k_p2p_exchange<double>, dim3(c->p2p_G + 1), dim3(256), 0, c->stream, a);
  }
}

}  // namespace mivi
