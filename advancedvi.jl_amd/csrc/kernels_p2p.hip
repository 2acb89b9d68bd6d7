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
// Hand-over: payload words travel as they are (4 / 8 bytes each, 16-byte write-through system-scope stores); a workgroup that has
// pushed its chunk waits for its own stores to be acknowledged (s_waitcnt vmcnt(0)) and then stores ONE flag per destination carrying
// the exchange's epoch number, (source rank, chunk) -> arr, (owner rank, chunk) -> farr; the consumer polls the R flags it needs and reads
// the payload with system-scope loads.  Twice per exchange {store acknowledge, flag round trip, data round trip}.  (An "LL" variant --
// every 32-bit word as an 8-byte (word, epoch) pair, no flags -- saves the acknowledge + flag round trips but doubles the bytes: on one
// GPU 33-46 us per exchange against 29-40 us, and in the pipelined batches, where the lanes hide latency and the system-scope path's
// bytes per second are what is left, 32 us per estimate against the figure in DESIGN.md 7; over xGMI the bytes are the bound outright.)
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

constexpr int kP2PLanes = 1, kP2PRing = 8, kP2PGroup = 4;   // ring = two groups of estimates: one being exchanged, one being computed

struct P2PTable {   // device resident: where every rank's exchange areas are mapped in THIS process, per lane
  char *stage[kP2PLanes][8];      // [2][V][R][n] T : stage[s] = rank s's staging area (contribution of rank `src` to slice s of the group's vector v at [parity][v][src])
  char *fin[kP2PLanes][8];        // [2][V][R n] T  : rank s's packed final vectors of the group
  unsigned *arr[kP2PLanes][8];    // [2][R][G]   : arrival flags (source rank, chunk) in rank s's memory
  unsigned *farr[kP2PLanes][8];   // [2][R][G+1] : final-chunk flags (owner rank, chunk) in rank s's memory; [vs][G] = the two scalars
};

template <typename T>
struct P2PArgs {
  int d, family, ent_kind, M_total;
  long long L, n, cn;        // partial length; slice length (multiple of 4); chunk length (multiple of 4)
  int rank, world, G, vs;    // vs = the rank whose slice holds the two scalars (sum ell, sum 0.5|eps|^2)
  const P2PTable *tab;
  unsigned *ctr;             // this lane's counters: [0] exchanges completed, [1] exit ticket, [2] the last epoch whose value this rank has consumed
  const T *P[kP2PRing];      // this rank's partial vectors: estimate t of the batch sits in P[t % ring], zero padded to world * n
  int ring;
  const T *params;
  T *value, *grad;           // results of the batch's LAST estimate
  T *scratch;                // [V] x {value (4 slots) + gradient} of the estimates before it (this lane's own: every estimate is fully written)
  long long scratch_stride;  // elements between the group's scratch outputs
  int *status;
  int phases;                // bit 0 push, bit 1 reduce, bit 2 unpack (all three = the exchange; single phases: host-sequenced tests, count = 1)
  int spin_budget;
  int lane, lanes, count;    // this launch serves the GROUPS gi = lane, lane + lanes, ... of V consecutive estimates (t = gi V ... < count)
  int direct;                // the partial kernels stored every vector straight into the owners' staging areas (OutArgs::p2p_direct): phase 1 only
                             // raises the arrival flags, and a ring slot = a staging slot, released (`freed`) once this rank has the FINAL chunks of
                             // its epoch -- which the owners stored after they had read their staging areas
  unsigned long long *stats; // diagnostics, accumulated by workgroup 0: ticks of the 100 MHz wall clock spent waiting for [0] the compute chain's hand-over,
                             // [1] the arrival flags of its chunk (the peers' pushes), [2] the final flags (the owners' reduced chunks); [3] groups served
  const unsigned *ready;     // batch hand-over (nullptr: the partial vector is complete at launch): estimate t may start when *ready >= t + 1
  unsigned *freed;           // [ring]: += 1 by every chunk workgroup once its part of the vector in that ring slot has been read
};

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store16_sys(void *p, u32x4_t r) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(r) : "memory");
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
template <typename T>
__device__ __forceinline__ T ld_sys(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
template <typename T>
__device__ __forceinline__ void st_sys(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// A flag is stored after this workgroup's payload stores have been ACKNOWLEDGED (the caller's s_waitcnt vmcnt(0) + barrier): they are
// write-through system-scope stores, so nothing of them sits in a cache that a release fence would have to write back -- and no
// cache-wide fence is issued (the compute kernels running beside the exchange keep their L2-resident operands).
__device__ __forceinline__ void flag_store(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// thread k < count polls flags[stride * k] until it carries `want` (bounded); result uniform over the workgroup.  The payload behind a
// flag is read with system-scope loads issued after the flag has been seen (in-order issue + the barrier below).
template <int NT>
__device__ __forceinline__ bool wait_flags(const unsigned *flags, int count, int stride, unsigned want, int budget, int *sh_ok) {
  if (threadIdx.x == 0) *sh_ok = 1;
  __syncthreads();
  for (int k = threadIdx.x; k < count; k += NT) {
    int b = budget;
    while (ld_sys(flags + (size_t)k * stride) != want) {
      if (--b <= 0) { atomicAnd(sh_ok, 0); break; }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  const bool ok = *sh_ok != 0;
  __syncthreads();
  return ok;
}

// packed index (>= d, full-rank) -> (column j, row i) of the lower triangle: e2 = j d - j (j - 1) / 2 + (i - j).  An f32 estimate of the
// column, made exact by the two integer loops (a few f64 instructions per ELEMENT -- sqrt, the loops' multiplies -- were 20 us of a 50 us
// exchange: reduce 25.8 -> and unpack 27.5 -> us; callers now resolve one index per 16-byte vector and walk from there).
__device__ __forceinline__ void packed_col_row(long long gi, int d, long long &j, long long &i) {
  const int e2 = (int)(gi - d);   // (< d (d + 1) / 2 < 2^31: 32-bit arithmetic -- 64-bit integer multiplies / divides are multi-instruction sequences here)
  const float b = 2.f * (float)d + 1.f;
  const float rad = b * b - 8.f * (float)e2;
  int jj = (int)((b - sqrtf(rad > 0.f ? rad : 0.f)) * 0.5f);
  if (jj < 0) jj = 0;
  if (jj > d - 1) jj = d - 1;
  while (jj > 0 && jj * d - (jj * (jj - 1)) / 2 > e2) --jj;
  while (jj + 1 < d && (jj + 1) * d - ((jj + 1) * jj) / 2 <= e2) ++jj;
  j = jj;
  i = jj + (e2 - (jj * d - (jj * (jj - 1)) / 2));
}
// the packed entry after (j, i) (column major, lower triangle)
__device__ __forceinline__ void packed_next(int d, long long &j, long long &i) {
  if (++i == d) { ++j; i = j; }
}

// packed final entry -> its finalised value: -(1/M) sum, the diagonal entries of the scale carry the entropy term.
// (j, i): the entry's place in the triangle (full-rank, gi >= d; unused otherwise)
template <typename T>
__device__ __forceinline__ double p2p_finalise(const P2PArgs<T> &a, long long gi, long long j, long long i, double sum, double invM, double direct) {
  const int d = a.d;
  double v = -sum * invM;
  if (gi < d) return v;
  if (a.family == MIVI_MEANFIELD) return v - direct / (double)a.params[gi];
  if (i == j) v -= direct / (double)a.params[d + (size_t)j * d + j];
  return v;
}

// phase 2 for R <= K sources (K in {1, 2, 4, 8}): 8 / K vectors of this thread x K sources per batch of eight loads.
// A vector = 16 bytes = V elements (float: 4, double: 2).  The caller has waited for the R arrival flags of this chunk.
template <typename T, int K, int NT>
__device__ __forceinline__ void p2p_reduce_chunk(const P2PArgs<T> &a, const P2PTable &tb, int pv, long long c0, long long clen) {   // pv = parity * V + vector
  constexpr int V = 16 / (int)sizeof(T), NV = 8 / K;
  const int tid = threadIdx.x, R = a.world;
  const long long n = a.n, tri_end = a.L - 2;
  const double invM = 1.0 / (double)a.M_total, direct = direct_entropy_coeff(a.ent_kind);
  const T *st = (const T *)tb.stage[a.lane][a.rank] + (size_t)(pv * R) * n + c0;   // + src * n + 16-byte vector
  const long long g0 = (long long)a.rank * n + c0;
  const int vecs = (int)(clen / V);
  for (int base = 0; base < vecs; base += NV * NT) {
    const void *ptr[8];
    u32x4_t raw[8];
    int vn[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      vn[v] = base + v * NT + tid;
      const int vc = vn[v] < vecs ? vn[v] : 0;
#pragma unroll
      for (int k = 0; k < K; ++k) ptr[v * K + k] = st + (size_t)(k < R ? k : R - 1) * n + V * vc;
    }
    ld16x8_sys(ptr, raw);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (vn[v] >= vecs) continue;
      const long long gi0 = g0 + (long long)vn[v] * V;
      if (gi0 >= tri_end) continue;   // (the two scalars belong to the value workgroup, the rest of the slice is padding)
      double acc[V];
#pragma unroll
      for (int c = 0; c < V; ++c) acc[c] = 0.0;
#pragma unroll
      for (int k = 0; k < K; ++k) {   // rank order: the sum does not depend on who arrived first
        if (k < R) {
          T e[V];
          __builtin_memcpy(e, &raw[v * K + k], 16);
#pragma unroll
          for (int c = 0; c < V; ++c) acc[c] += (double)e[c];
        }
      }
      T o[V];
      long long pj = 0, pi = 0;   // place of the vector's first packed entry; the others by walking
      const bool tri = a.family != MIVI_MEANFIELD && gi0 + V > a.d;
      if (tri) packed_col_row(gi0 > a.d ? gi0 : (long long)a.d, a.d, pj, pi);
#pragma unroll
      for (int c = 0; c < V; ++c) {
        const long long gi = gi0 + c;
        o[c] = gi < tri_end ? (T)p2p_finalise(a, gi, pj, pi, acc[c], invM, direct) : T(0);
        if (tri && gi >= a.d) packed_next(a.d, pj, pi);
      }
      const size_t off = (size_t)pv * R * n + (size_t)gi0;   // element index inside a final area
      if (gi0 + V <= tri_end) {
        u32x4_t ov;
        __builtin_memcpy(&ov, o, 16);
        for (int k = 0; k < R; ++k) store16_sys((T *)tb.fin[a.lane][(a.rank + 1 + k) % R] + off, ov);
      } else {   // the vector that holds the scalars is stored without them: the value workgroup writes those, with their own flag
        for (int k = 0; k < R; ++k) {
          T *dst = (T *)tb.fin[a.lane][(a.rank + 1 + k) % R] + off;
          for (int c = 0; c < V; ++c)
            if (gi0 + c < tri_end) st_sys(dst + c, o[c]);
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_p2p_exchange(P2PArgs<T> a) {   // (<= 128 VGPRs: the spinning workgroups must leave the compute kernels their registers)
  constexpr int NT = 256, V = 16 / (int)sizeof(T);   // elements per 16-byte vector
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

  // One EPOCH of the exchange serves a GROUP of up to GV consecutive estimates: an exchange is a chain of ~9 memory round trips whatever it
  // carries, so the group's vectors share every flag and every wait (per estimate: 23 us alone, ~7 us in a group of four on one GPU).
  constexpr int GV = kP2PGroup;
  for (int gi = a.lane; gi * GV < a.count; gi += a.lanes) {
    ++epoch;
    const int p = (int)(epoch & 1u);
    const int t0 = gi * GV, nv = a.count - t0 < GV ? a.count - t0 : GV;
    const bool first_group = (gi == a.lane);
    const bool timed = a.stats && g == 0 && tid == 0;
    long long tk0 = timed ? (long long)wall_clock64() : 0;
    if (a.ready) {   // hand-over from the compute chain: the partial vectors of the group's estimates are complete
      if (tid == 0) {
        int budget = lost ? 64 : a.spin_budget;   // (after a lost peer every further wait of this launch gives up at once: a dead batch ends in milliseconds)
        sh_ok = 1;
        while ((int)__hip_atomic_load(a.ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < t0 + nv) {
          if (--budget <= 0) { sh_ok = 0; break; }
          __builtin_amdgcn_s_sleep(4);
        }
      }
      __syncthreads();
      if (!sh_ok) lost = true;
      __syncthreads();
    }

    if (timed) { const long long tk1 = (long long)wall_clock64(); a.stats[0] += (unsigned long long)(tk1 - tk0); a.stats[3] += 1ull; }
    // ---- phase 1: push chunk g of every slice of every vector to its owner, then one arrival flag per owner ---------------------------
    if ((a.phases & 1) && !value_wg) {
      // The areas are double-buffered by epoch parity with no acknowledgements: safe for the chunk workgroups because workgroup g of
      // epoch e + 2 depends, through the flags it waits for, on workgroup g of every peer having finished epoch e.  The VALUE workgroup
      // sits outside that chain: the two scalars it reads (stage area, rank vs) and writes (final areas) live in the chunks ga / gb of
      // slice vs.  So the chunk workgroups ga / gb of EVERY rank do not start pushing epoch e + 2 before their own rank's value
      // workgroup has consumed epoch e (ctr[2], a device-local word): that value was read behind rank vs's final flag, which rank vs
      // stores after reading its stage scalars -- both parities of both hazards are closed by this one wait.
      if ((a.phases & 4) && a.rank >= 0 && !a.direct) {
        const long long o0 = tri_end - (long long)a.vs * n;
        const int ga = (int)(o0 / a.cn), gb = (int)((o0 + 1) / a.cn);
        if (g == ga || g == gb) {
          if (tid == 0) {
            int budget = lost ? 64 : a.spin_budget;
            sh_ok = 1;
            while ((int)(epoch - __hip_atomic_load(a.ctr + 2, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) > 2) {
              if (--budget <= 0) { sh_ok = 0; break; }
              __builtin_amdgcn_s_sleep(4);
            }
          }
          __syncthreads();
          if (!sh_ok) lost = true;
          __syncthreads();
        }
      }
      const int vecs = (int)(clen / V);   // (clen is a multiple of 4)
      for (int v = 0; v < (a.direct ? 0 : nv); ++v) {   // (direct: the vectors are in the staging areas already)
        const T *P = a.P[(t0 + v) % a.ring];
        for (int base = 0; base < vecs; base += 8 * NT) {
          for (int k = 0; k < R; ++k) {
            const int s = (a.rank + 1 + k) % R;   // start with the neighbour: the links fill evenly, the local copy comes last
            const T *src = P + (size_t)s * n + c0;
            T *dst = (T *)tb.stage[ln][s] + (size_t)((p * GV + v) * R + a.rank) * n + c0;
            const void *ptr[8];
            u32x4_t r[8];
            int vi[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              vi[u] = base + u * NT + tid;
              ptr[u] = src + V * (vi[u] < vecs ? vi[u] : 0);
            }
            ld16x8_sys(ptr, r);   // (system scope: the vector was written by the compute kernels' XCDs and this kernel never restarts)
#pragma unroll
            for (int u = 0; u < 8; ++u)
              if (vi[u] < vecs) store16_sys(dst + V * vi[u], r[u]);
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every payload store of this wave has been acknowledged ...
      __syncthreads();                                    // ... and of this workgroup
      if (tid < R) flag_store(tb.arr[ln][tid] + (size_t)(p * R + a.rank) * G + g, epoch);
      if (a.freed && tid < nv && !a.direct)   // this workgroup is done reading the group's partial vectors in their ring slots
        __hip_atomic_fetch_add(a.freed + (t0 + tid) % a.ring, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    // ---- phase 3a (needs nothing from anybody): exact zeros above the diagonal of the dense gradients -------------------------------------
    // (a scratch gradient keeps its zeros from this launch's first group: only the unpack below writes it, and only below the diagonal;
    //  the caller's buffer gets them with the batch's last estimate)
    if ((a.phases & 4) && !value_wg && a.family == MIVI_FULLRANK) {
      for (int v = 0; v < nv; ++v) {
        const bool last = (t0 + v == a.count - 1);
        if (!(last || first_group)) continue;
        T *gc = (last ? a.grad : a.scratch + (size_t)v * a.scratch_stride + 4) + d;
        for (int j = g + 1; j < d; j += G)
          for (int i = tid; i < j; i += NT) gc[(size_t)j * d + i] = T(0);
      }
    }

    // ---- phase 2: reduce + finalise chunk g of MY slice of every vector, push the final chunks to every rank, one final flag per rank ------
    if ((a.phases & 2) && !value_wg) {
      tk0 = timed ? (long long)wall_clock64() : 0;
      if (!wait_flags<NT>(tb.arr[ln][a.rank] + (size_t)(p * R) * G + g, R, G, epoch, lost ? 64 : a.spin_budget, &sh_ok)) lost = true;
      if (timed) a.stats[1] += (unsigned long long)((long long)wall_clock64() - tk0);
      for (int v = 0; v < nv; ++v) {
        const int pv = p * GV + v;
        if (R == 1) p2p_reduce_chunk<T, 1, NT>(a, tb, pv, c0, clen);
        else if (R == 2) p2p_reduce_chunk<T, 2, NT>(a, tb, pv, c0, clen);
        else if (R <= 4) p2p_reduce_chunk<T, 4, NT>(a, tb, pv, c0, clen);
        else p2p_reduce_chunk<T, 8, NT>(a, tb, pv, c0, clen);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid < R) flag_store(tb.farr[ln][tid] + (size_t)(p * R + a.rank) * (G + 1) + g, epoch);
    }
    if ((a.phases & 2) && value_wg && a.rank == a.vs) {   // the objective values: sum ell, sum 0.5|eps|^2 of all ranks + the parameter-only terms
      const long long o0 = tri_end - (long long)a.vs * n, o1 = o0 + 1;   // offsets of the two scalars inside my slice
      const int ga = (int)(o0 / a.cn), gb = (int)(o1 / a.cn);
      if (!wait_flags<NT>(tb.arr[ln][a.rank] + (size_t)(p * R) * G + ga, R, G, epoch, lost ? 64 : a.spin_budget, &sh_ok)) lost = true;
      if (gb != ga && !wait_flags<NT>(tb.arr[ln][a.rank] + (size_t)(p * R) * G + gb, R, G, epoch, lost ? 64 : a.spin_budget, &sh_ok)) lost = true;
      double s_ld = 0.0, bad = 0.0;
      for (int i = tid; i < d; i += NT) {
        const double c = (double)(a.family == MIVI_MEANFIELD ? a.params[d + i] : a.params[d + (size_t)i * d + i]);
        if (!(c > 0.0)) bad = 1.0;
        s_ld += log(c);
      }
      s_ld = block_sum<double, NT>(s_ld, red);
      bad = block_sum<double, NT>(bad, red);
      if (tid < nv) {
        const int v = tid;
        const T *st = (const T *)tb.stage[ln][a.rank] + (size_t)((p * GV + v) * R) * n;
        double sum_ell = 0.0, s_he = 0.0;
        for (int src = 0; src < R; ++src) {
          sum_ell += (double)ld_sys(st + (size_t)src * n + o0);
          s_he += (double)ld_sys(st + (size_t)src * n + o1);
        }
        const double Mt = (double)a.M_total;
        const double ent = (ent_is_closed(a.ent_kind) ? 0.5 * d * (1.0 + kLog2Pi) : s_he / Mt + 0.5 * d * kLog2Pi) + s_ld;
        const double value = -(sum_ell / Mt + ent);
        int stt = 0;
        if (!isfinite(value)) stt |= 1;
        if (bad > 0.0) stt |= 2;
        for (int s = 0; s < R; ++s) {
          T *dst = (T *)tb.fin[ln][s] + (size_t)(p * GV + v) * R * n;
          st_sys(dst + tri_end, (T)value);
          st_sys(dst + tri_end + 1, (T)stt);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid < R) flag_store(tb.farr[ln][tid] + (size_t)(p * R + a.vs) * (G + 1) + G, epoch);
    }

    // ---- phase 3: unpack chunk g of every final slice of every vector ---------------------------------------------------------------------
    if (a.phases & 4) {
      if (value_wg) {
        if (!wait_flags<NT>(tb.farr[ln][a.rank] + (size_t)(p * R + a.vs) * (G + 1) + G, 1, 1, epoch, lost ? 64 : a.spin_budget, &sh_ok)) lost = true;
        if (tid < nv) {
          const int v = tid;
          const T *fin = (const T *)tb.fin[ln][a.rank] + (size_t)(p * GV + v) * R * n;
          T *out_v = (t0 + v == a.count - 1) ? a.value : a.scratch + (size_t)v * a.scratch_stride;
          *out_v = ld_sys(fin + tri_end);
          const int stt = (int)ld_sys(fin + tri_end + 1);
          if (stt && a.status) atomicOr(a.status, stt);
        }
        __syncthreads();   // (the group's scalars have been read: the chunk workgroups ga / gb may go two epochs ahead, see phase 1)
        if (tid == 0) __hip_atomic_store(a.ctr + 2, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        // chunk g of every owner's slice: R final flags
        tk0 = timed ? (long long)wall_clock64() : 0;
        if (!wait_flags<NT>(tb.farr[ln][a.rank] + (size_t)(p * R) * (G + 1) + g, R, G + 1, epoch, lost ? 64 : a.spin_budget, &sh_ok)) lost = true;
        if (timed) a.stats[2] += (unsigned long long)((long long)wall_clock64() - tk0);
        const int vecs = (int)(clen / V), total = vecs * R;   // vector index x = s * vecs + u  (32-bit: 64-bit divides are long sequences)
        for (int v = 0; v < nv; ++v) {
          const T *fin = (const T *)tb.fin[ln][a.rank] + (size_t)(p * GV + v) * R * n;
          T *out_g = (t0 + v == a.count - 1) ? a.grad : a.scratch + (size_t)v * a.scratch_stride + 4;
          for (int base = 0; base < total; base += 8 * NT) {
            const void *ptr[8];
            u32x4_t raw[8];
            long long gi0[8];
            bool in[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int x = base + u * NT + tid;
              in[u] = x < total;
              const int sx = in[u] ? x / vecs : 0, uu = in[u] ? x - sx * vecs : 0;
              gi0[u] = (long long)sx * n + c0 + (long long)uu * V;
              if (gi0[u] >= tri_end) in[u] = false;   // (scalars / padding: nothing of this vector is part of the gradient)
              ptr[u] = fin + (size_t)(in[u] ? gi0[u] : 0);
            }
            ld16x8_sys(ptr, raw);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              if (!in[u]) continue;
              T e[V];
              __builtin_memcpy(e, &raw[u], 16);
              long long pj = 0, pi = 0;
              const bool tri = a.family != MIVI_MEANFIELD && gi0[u] + V > d;
              if (tri) packed_col_row(gi0[u] > d ? gi0[u] : (long long)d, d, pj, pi);
#pragma unroll
              for (int c = 0; c < V; ++c) {
                const long long gidx = gi0[u] + c;
                if (gidx >= tri_end) continue;
                long long di = gidx;
                if (tri && gidx >= d) {
                  di = d + pj * d + pi;
                  packed_next(d, pj, pi);
                }
                out_g[di] = e[c];
              }
            }
          }
        }
      }
    }
    // direct mode: this workgroup has seen the final flags of its chunk (the value workgroup: of the scalars) from every owner, i.e. every
    // owner is done with this epoch's staging data of the chunk -- the compute chain may store the epoch after next into these staging slots
    if (a.direct && (a.phases & 4) && a.freed) {
      __syncthreads();
      if (tid < nv) __hip_atomic_fetch_add(a.freed + (t0 + tid) % a.ring, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// hand-over words of the pipelined batch on the COMPUTE chain (one thread): announce complete partial vectors (ready = ready_val), then
// hold the chain until the exchange has read the ring slots the NEXT launches are going to overwrite (*freed[k] >= min[k], up to four)
struct P2PHandover {
  unsigned *ready;
  unsigned ready_val;
  const unsigned *freed[4];
  unsigned freed_min[4];
  int n_freed, budget;
  int *status;
};
__global__ void k_p2p_handover(P2PHandover h) {
  // (relaxed: the partial vectors were written -- through, sc1 -- by kernels that completed before this one started; a release here is an
  //  L2 write-back of whatever the exchange kernel running beside the chain has dirtied)
  if (h.ready) __hip_atomic_store(h.ready, h.ready_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int budget = h.budget;
  for (int k = 0; k < h.n_freed; ++k) {
    while ((int)(__hip_atomic_load(h.freed[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - h.freed_min[k]) < 0) {
      if (--budget <= 0) { if (h.status) atomicOr(h.status, 8); return; }
      __builtin_amdgcn_s_sleep(4);
    }
  }
}

// host side -------------------------------------------------------------------------------------------------------------------------
void launch_p2p_handover(mivi_ctx *c, unsigned *ready, unsigned ready_val, const unsigned *freed, unsigned freed_min) {
  P2PHandover h{};
  h.ready = ready; h.ready_val = ready_val; h.budget = c->p2p_spin; h.status = (int *)c->status.p;
  if (freed) { h.freed[0] = freed; h.freed_min[0] = freed_min; h.n_freed = 1; }
  hipLaunchKernelGGL(k_p2p_handover, dim3(1), dim3(1), 0, c->stream, h);
}
void launch_p2p_handover4(mivi_ctx *c, unsigned *ready, unsigned ready_val, const unsigned *const *freed, const unsigned *freed_min, int n) {
  P2PHandover h{};
  h.ready = ready; h.ready_val = ready_val; h.budget = c->p2p_spin; h.status = (int *)c->status.p;
  for (int k = 0; k < n && k < 4; ++k) { h.freed[k] = freed[k]; h.freed_min[k] = freed_min[k]; }
  h.n_freed = n < 4 ? n : 4;
  hipLaunchKernelGGL(k_p2p_handover, dim3(1), dim3(1), 0, c->stream, h);
}

// one lane of the exchange on c->stream: estimates t = lane, lane + lanes, ... < count with partial vectors P[t % ring]
void launch_p2p_exchange(mivi_ctx *c, const void *params, const void *const *P, int ring, void *value, void *grad, int phases, int lane, int lanes,
                         int count, const unsigned *ready, unsigned *freed, bool direct) {
  auto fill = [&](auto &a) {
    a.direct = direct ? 1 : 0;
    a.d = c->cfg.d; a.family = c->cfg.family; a.ent_kind = c->cfg.entropy; a.M_total = c->M_total;
    a.L = mivi_partials_len(c); a.n = c->p2p_n; a.cn = c->p2p_cn;
    a.rank = c->p2p_rank; a.world = c->p2p_world; a.G = c->p2p_G; a.vs = c->p2p_vs;
    a.tab = (const P2PTable *)c->p2p_tab.p;
    a.ctr = (unsigned *)c->p2p_ctr.p + 16 * lane;
    a.stats = (unsigned long long *)((unsigned *)c->p2p_ctr.p + 96);   // (words 96 .. 103 of the counter block: mivi_p2p_stats)
    a.status = (int *)c->status.p;
    a.phases = phases;
    a.spin_budget = c->p2p_spin;
    a.lane = lane; a.lanes = lanes; a.count = count;
    a.ring = ring;
    a.ready = ready;
    a.freed = freed;
  };
  const size_t plen4 = (size_t)mivi_params_len(c) + 4;   // one scratch output {value (4 slots), gradient}; kP2PGroup of them per lane
  if (c->cfg.dtype == MIVI_F32) {
    P2PArgs<float> a{};
    fill(a);
    for (int k = 0; k < ring; ++k) a.P[k] = (const float *)P[k];
    a.params = (const float *)params; a.value = (float *)value; a.grad = (float *)grad;
    a.scratch = (float *)c->p2p_scratch.p + plen4 * kP2PGroup * lane; a.scratch_stride = (long long)plen4;
    hipLaunchKernelGGL(k_p2p_exchange<float>, dim3(c->p2p_G + 1), dim3(256), 0, c->stream, a);
  } else {
    P2PArgs<double> a{};
    fill(a);
    for (int k = 0; k < ring; ++k) a.P[k] = (const double *)P[k];
    a.params = (const double *)params; a.value = (double *)value; a.grad = (double *)grad;
    a.scratch = (double *)c->p2p_scratch.p + plen4 * kP2PGroup * lane; a.scratch_stride = (long long)plen4;
    hipLaunchKernelGGL(k_p2p_exchange<double>, dim3(c->p2p_G + 1), dim3(256), 0, c->stream, a);
  }
}

}  // namespace mivi
