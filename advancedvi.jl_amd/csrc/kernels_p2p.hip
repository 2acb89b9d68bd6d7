// The gradient exchange of the sharded estimator written for xGMI (no counterpart in the reference, which is a
// single task: src/algorithms/repgradelbo.jl:84-86 is a mean over samples, so what crosses GPUs is a SUM of partial vectors).
//
// RCCL's ring all-reduce of the 2.1 MB north-star partial vector costs 2 (R - 1) dependent hops (50-70 us of latency at R = 8 against
// 14 us of compute).  xGMI is point to point -- every GPU has a direct link to every other one -- so the exchange is written as ONE
// kernel per rank with two one-hop phases, all seven links busy in both:
//
//   phase 1  push        every rank stores slice s of its partial vector straight into rank s's staging area (peer stores), flag
//   phase 2  reduce      rank s sums the R contributions of ITS slice in rank order (f64), finalises it (-1/M, entropy diagonal
//                        terms; the slice that holds the two scalars also assembles the objective value) and stores the packed
//                        final slice into EVERY rank's final buffer (peer stores), flag
//   phase 3  unpack      every rank expands the packed final vector into value + dense gradient (exact zeros above the diagonal)
//
// Every slice is finalised by exactly one rank from contributions summed in rank order: all ranks hold bit-identical results and the
// sum is independent of arrival order.  Slices are cut into G chunks; workgroup g of every rank handles chunk g of every slice and
// synchronises only with workgroup g of its peers through (source rank, chunk) flags carrying the exchange's epoch number -- no
// grid-wide barrier, no dependency cycle (phase 1 never waits).  Staging / final / flag buffers are double-buffered by epoch parity:
// an exchange can only complete on a rank after every peer finished reducing the previous one, so epoch e + 2 never overwrites data
// epoch e still needs.  Memory: one fine-grained allocation per rank, mapped into its peers through HIP IPC
// (mivi_p2p_export / mivi_p2p_attach); stores to peers are system-scope write-through, flags are released / acquired at system scope.
// Every spin is bounded: a lost peer sets status bit 8 and the kernel leaves (the host reports it; nothing hangs).
#include "device_common.h"

namespace mivi {

struct P2PTable {   // device resident: where every rank's exchange areas are mapped in THIS process
  char *stage[8];       // [2][R][n] T : stage[s] = rank s's staging area (contribution of rank `src` to slice s at [parity][src])
  char *fin[8];         // [2][R n] T  : rank s's packed final vector
  unsigned *arr[8];     // [2][R][G]   : arrival flags of (source rank, chunk)
  unsigned *farr[8];    // [2][R][G+1] : final-slice arrival flags of (owner rank, chunk); slot G of the value owner = the two scalars
};

template <typename T>
struct P2PArgs {
  int d, family, ent_kind, M_total;
  long long L, n, cn;        // partial length; slice length (multiple of 4); chunk length (multiple of 4)
  int rank, world, G, vs;    // vs = the rank whose slice holds the two scalars (sum ell, sum 0.5|eps|^2)
  const P2PTable *tab;
  unsigned *ctr;             // [0] exchanges completed on this rank, [1] exit ticket
  const T *partials;         // this rank's partial vector, zero padded to world * n
  const T *params;
  T *value, *grad;
  int *status;
  int phases;                // bit 0 push, bit 1 reduce, bit 2 unpack (all three = the exchange; single phases: host-sequenced tests)
  int spin_budget;
};

__device__ __forceinline__ void store16_sys(void *p, const void *src16) {   // 16-byte system-scope write-through store
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t r = *(const u32x4_t *)src16;
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(r) : "memory");
}
// What a peer (or another XCD of this GPU: every XCD has its own L2, and they are not coherent with each other) stored into the
// exchange areas is read with SYSTEM-scope loads (sc0 sc1): they never hit a stale line this XCD's L2 kept from the exchange two epochs
// ago -- plain loads did (found on one GPU: a staging area re-used across epochs / allocations returned the previous contents to the
// XCDs that had read them before, although memory held the new data).
template <typename T>
__device__ __forceinline__ T ld_sys(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void flag_release(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
// wait until *p == want (bounded); returns false on a lost peer
__device__ __forceinline__ bool flag_wait(const unsigned *p, unsigned want, int budget) {
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != want) {
    if (--budget <= 0) return false;
    __builtin_amdgcn_s_sleep(8);
  }
  return true;
}

// one workgroup waits for `count` flags flags[stride * k] (k < count): thread k polls flag k; result uniform
template <int NT>
__device__ __forceinline__ bool wait_flags(const unsigned *flags, int count, int stride, unsigned want, int budget, int *sh_ok) {
  if (threadIdx.x == 0) *sh_ok = 1;
  __syncthreads();
  for (int k = threadIdx.x; k < count; k += NT)
    if (!flag_wait(flags + (size_t)k * stride, want, budget)) atomicAnd(sh_ok, 0);
  __syncthreads();
  const bool ok = *sh_ok != 0;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope: what the flags' writers stored before releasing them is visible
  __syncthreads();
  return ok;
}

template <typename T>
__global__ __launch_bounds__(256) void k_p2p_exchange(P2PArgs<T> a) {
  constexpr int NT = 256, V = 16 / sizeof(T);   // elements per 16-byte vector
  __shared__ int sh_ok;
  __shared__ double red[4];
  const int tid = threadIdx.x, g = blockIdx.x, R = a.world, G = a.G;
  const unsigned epoch = a.ctr[0] + 1u;
  const int p = (int)(epoch & 1u);
  const long long n = a.n, Lp = n * R;
  const P2PTable &tb = *a.tab;
  const bool value_wg = (g == G);
  const long long c0 = value_wg ? 0 : (long long)g * a.cn;
  const long long clen = value_wg ? 0 : ((c0 + a.cn <= n ? a.cn : (n > c0 ? n - c0 : 0)));
  bool lost = false;

  // ---- phase 1: push chunk g of every slice to its owner ---------------------------------------------------------------------
  if ((a.phases & 1) && !value_wg) {
    for (int k = 0; k < R; ++k) {
      const int s = (a.rank + 1 + k) % R;   // start with the neighbour: the links fill evenly, the local copy comes last
      const T *src = a.partials + (size_t)s * n + c0;
      T *dst = (T *)tb.stage[s] + ((size_t)(p * R + a.rank)) * n + c0;
      for (long long e = (long long)tid * V; e < clen; e += (long long)NT * V) store16_sys(dst + e, src + e);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < R) flag_release(tb.arr[tid] + ((size_t)(p * R + a.rank)) * G + g, epoch);
  }

  // ---- phase 2: reduce + finalise chunk g of MY slice, push the final chunk to every rank -------------------------------------------
  const double invM = 1.0 / (double)a.M_total;
  const double direct = direct_entropy_coeff(a.ent_kind);
  const long long tri_end = a.L - 2;
  const int d = a.d;
  if ((a.phases & 2) && !value_wg) {
    if (!wait_flags<NT>(tb.arr[a.rank] + (size_t)(p * R) * G + g, R, G, epoch, a.spin_budget, &sh_ok)) lost = true;
    const T *st = (const T *)tb.stage[a.rank] + (size_t)(p * R) * n + c0;
    const long long g0 = (long long)a.rank * n + c0;
    for (long long e = (long long)tid * V; e < clen; e += (long long)NT * V) {
      double acc[V];
#pragma unroll
      for (int c = 0; c < V; ++c) acc[c] = 0.0;
      for (int src = 0; src < R; ++src) {   // rank order: the sum does not depend on who arrived first
#pragma unroll
        for (int c = 0; c < V; ++c) acc[c] += (double)ld_sys(st + (size_t)src * n + e + c);
      }
      T o[V] __attribute__((aligned(16)));
#pragma unroll
      for (int c = 0; c < V; ++c) {
        const long long gi = g0 + e + c;
        double v = -acc[c] * invM;
        if (gi >= tri_end) {
          v = 0.0;   // the two scalars belong to the value workgroup (which stores them itself), the rest is padding
        } else if (gi >= d) {
          if (a.family == MIVI_MEANFIELD) {
            v -= direct / (double)a.params[gi];
          } else {   // packed entry e2 = j d - j (j - 1) / 2 + (i - j): the diagonal entries carry the entropy term
            const long long e2 = gi - d;
            const double b = 2.0 * d + 1.0;
            long long j = (long long)((b - sqrt(b * b - 8.0 * (double)e2)) * 0.5);
            if (j < 0) j = 0;
            if (j > d - 1) j = d - 1;
            while (j > 0 && j * d - (j * (j - 1)) / 2 > e2) --j;
            while (j + 1 < d && (j + 1) * d - ((j + 1) * j) / 2 <= e2) ++j;
            if (e2 == j * d - (j * (j - 1)) / 2) v -= direct / (double)a.params[d + (size_t)j * d + j];
          }
        }
        o[c] = (T)v;
      }
      // (the vector that holds the scalars is stored without them: elements >= tri_end only ever sit in the value owner's slice, and
      //  the value workgroup writes them with their own flag)
      const long long gi0 = g0 + e;
      if (gi0 + V <= tri_end || gi0 >= a.L) {
        for (int k = 0; k < R; ++k) {
          const int s = (a.rank + 1 + k) % R;
          store16_sys((T *)tb.fin[s] + (size_t)p * Lp + gi0, o);
        }
      } else {
        for (int k = 0; k < R; ++k) {
          const int s = (a.rank + 1 + k) % R;
          T *dst = (T *)tb.fin[s] + (size_t)p * Lp + gi0;
          for (int c = 0; c < V; ++c)
            if (gi0 + c < tri_end) __hip_atomic_store(dst + c, o[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < R) flag_release(tb.farr[tid] + ((size_t)(p * R + a.rank)) * (G + 1) + g, epoch);
  }
  if ((a.phases & 2) && value_wg && a.rank == a.vs) {   // the objective value: sum ell, sum 0.5|eps|^2 of all ranks + the parameter-only terms
    const long long o0 = tri_end - (long long)a.vs * n, o1 = o0 + 1;   // offsets of the two scalars inside my slice
    const int ga = (int)(o0 / a.cn), gb = (int)(o1 / a.cn);
    if (!wait_flags<NT>(tb.arr[a.rank] + (size_t)(p * R) * G + ga, R, G, epoch, a.spin_budget, &sh_ok)) lost = true;
    if (gb != ga && !wait_flags<NT>(tb.arr[a.rank] + (size_t)(p * R) * G + gb, R, G, epoch, a.spin_budget, &sh_ok)) lost = true;
    double s_ld = 0.0, bad = 0.0;
    for (int i = tid; i < d; i += NT) {
      const double c = (double)(a.family == MIVI_MEANFIELD ? a.params[d + i] : a.params[d + (size_t)i * d + i]);
      if (!(c > 0.0)) bad = 1.0;
      s_ld += log(c);
    }
    s_ld = block_sum<double, NT>(s_ld, red);
    bad = block_sum<double, NT>(bad, red);
    if (tid == 0) {
      const T *st = (const T *)tb.stage[a.rank] + (size_t)(p * R) * n;
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
        T *dst = (T *)tb.fin[s] + (size_t)p * Lp;
        __hip_atomic_store(dst + tri_end, (T)value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(dst + tri_end + 1, (T)stt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      for (int s = 0; s < R; ++s) flag_release(tb.farr[s] + ((size_t)(p * R + a.vs)) * (G + 1) + G, epoch);
    }
  }

  // ---- phase 3: unpack the packed final vector ---------------------------------------------------------------------------------------
  if (a.phases & 4) {
    const T *fin = (const T *)tb.fin[a.rank] + (size_t)p * Lp;
    if (value_wg) {
      if (!wait_flags<NT>(tb.farr[a.rank] + ((size_t)(p * R + a.vs)) * (G + 1) + G, 1, 1, epoch, a.spin_budget, &sh_ok)) lost = true;
      if (tid == 0) {
        *a.value = ld_sys(fin + tri_end);
        const int stt = (int)ld_sys(fin + tri_end + 1);
        if (stt && a.status) atomicOr(a.status, stt);
      }
    } else {
      // my share of the dense gradient touches chunks of every slice: wait for all R x G final chunks
      bool ok = true;
      for (int s = 0; s < R; ++s)
        if (!wait_flags<NT>(tb.farr[a.rank] + ((size_t)(p * R + s)) * (G + 1), G, 1, epoch, a.spin_budget, &sh_ok)) ok = false;
      if (!ok) lost = true;
      const long long plen = a.family == MIVI_MEANFIELD ? 2 * (long long)d : (long long)d + (long long)d * d;
      const long long pc = ((plen + G - 1) / G + 3) & ~3LL;
      const long long t0 = (long long)g * pc, t1 = t0 + pc < plen ? t0 + pc : plen;
      for (long long t = t0 + tid; t < t1; t += NT) {
        T v;
        if (a.family == MIVI_MEANFIELD || t < d) {
          v = ld_sys(fin + t);
        } else {
          const long long e2 = t - d, j = e2 / d, i = e2 - j * d;
          v = (j > i) ? T(0) : ld_sys(fin + d + j * d - (j * (j - 1)) / 2 + (i - j));
        }
        a.grad[t] = v;
      }
    }
  }
  if (lost && tid == 0 && a.status) atomicOr(a.status, 8);

  // ---- exit ticket: the last workgroup out advances the epoch (every workgroup has read it by then) ------------------------------------
  if (a.phases & 4) {
    __syncthreads();
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(a.ctr + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (t == (unsigned)G) {
        __hip_atomic_store(a.ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.ctr, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

// host side -------------------------------------------------------------------------------------------------------------------------
void launch_p2p_exchange(mivi_ctx *c, const void *params, const void *partials, void *value, void *grad, int phases) {
  auto fill = [&](auto &a) {
    a.d = c->cfg.d; a.family = c->cfg.family; a.ent_kind = c->cfg.entropy; a.M_total = c->M_total;
    a.L = mivi_partials_len(c); a.n = c->p2p_n; a.cn = c->p2p_cn;
    a.rank = c->p2p_rank; a.world = c->p2p_world; a.G = c->p2p_G; a.vs = c->p2p_vs;
    a.tab = (const P2PTable *)c->p2p_tab.p;
    a.ctr = (unsigned *)c->p2p_ctr.p;
    a.status = (int *)c->status.p;
    a.phases = phases;
    a.spin_budget = c->p2p_spin;
  };
  if (c->cfg.dtype == MIVI_F32) {
    P2PArgs<float> a{};
    fill(a);
    a.partials = (const float *)partials; a.params = (const float *)params; a.value = (float *)value; a.grad = (float *)grad;
    hipLaunchKernelGGL(k_p2p_exchange<float>, dim3(c->p2p_G + 1), dim3(256), 0, c->stream, a);
  } else {
    P2PArgs<double> a{};
    fill(a);
    a.partials = (const double *)partials; a.params = (const double *)params; a.value = (double *)value; a.grad = (double *)grad;
    hipLaunchKernelGGL(k_p2p_exchange<double>, dim3(c->p2p_G + 1), dim3(256), 0, c->stream, a);
  }
}

}  // namespace mivi
