// Device-side helpers shared by the kernels: wave64 / workgroup reductions and the objective-value
// finalisation (negative ELBO assembly of src/algorithms/repgradelbo.jl:112-118,142-149).
#pragma once
#include "mivi_internal.h"
#include "philox.h"
#include <type_traits>

namespace mivi {

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// f32 wave64 sum on the DPP crossbar (no LDS traffic): quad_perm, quad_perm, row_half_mirror, row_mirror
// leave every lane with its 16-lane row sum; four v_readlane finish.  Result is wave-uniform.
__device__ __forceinline__ float wave_sum_f32(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  const int iv = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
}
__device__ __forceinline__ double wave_sum_fast(float v) { return (double)wave_sum_f32(v); }
__device__ __forceinline__ double wave_sum_fast(double v) { return wave_sum(v); }

// Sum `v` over the workgroup; result valid in every thread. `red` must hold NT/64 elements.
// Fixed reduction tree => bitwise reproducible for a given launch geometry.
template <typename T, int NT>
__device__ __forceinline__ T block_sum(T v, T *red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  T s = red[0];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) s += red[i];
  return s;
}

// block_sum with barriers that order LDS traffic only (lds_barrier): the caller's outstanding global stores are NOT drained.  For
// sums taken right behind write-through stores (their acknowledge is ~1.5 us: a __syncthreads() there puts that wait in front
// of whatever the workgroup does next -- a rider, its exit).  Same tree: bitwise the same result as block_sum.
__device__ __forceinline__ void lds_barrier();
template <typename T, int NT>
__device__ __forceinline__ T block_sum_nodrain(T v, T *red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  v = wave_sum(v);
  lds_barrier();
  if (lane == 0) red[w] = v;
  lds_barrier();
  T s = red[0];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) s += red[i];
  return s;
}

// The same for an f32 per-thread value: wave totals on the DPP crossbar in f32 (wave_sum_f32: ~60 cycles against ~900 for six rounds
// of two ds_bpermute + v_add_f64), the NT / 64 wave totals summed in f64.  Fixed tree.
template <int NT>
__device__ __forceinline__ double block_sum_nodrain_f32(float v, double *red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const double sv = (double)wave_sum_f32(v);
  lds_barrier();
  if (lane == 0) red[w] = sv;
  lds_barrier();
  double s = red[0];
#pragma unroll
  for (int i = 1; i < NT / 64; ++i) s += red[i];
  return s;
}

// Sum N values over the workgroup in ONE pass: the N wave reductions are independent shuffle chains (they pipeline), one LDS
// exchange, one barrier pair -- instead of N x {6-step shuffle chain, two barriers} one after the other, which is what made the
// one-workgroup value assembly a 6 us latency chain.  Same tree per value as block_sum (bitwise the same results).
// `red` holds N * NT / 64 elements.  Results valid in every thread.
template <typename T, int NT, int N>
__device__ __forceinline__ void block_sum_n(T (&v)[N], T *red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = wave_sum(v[k]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < N; ++k) red[k * (NT / 64) + w] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) {
    T s = red[k * (NT / 64)];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) s += red[k * (NT / 64) + i];
    v[k] = s;
  }
}

// compile-time loop: f(integral_constant<int, B>) ... f(integral_constant<int, E - 1>) -- indices usable as template arguments
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// 16-byte WRITE-THROUGH store (sc1): the line does not stay dirty in the XCD's L2.
//   * Dirty lines left in L2 are written back when the kernel ENDS, on its critical path: 1 MB of eps(t+1) written by rider
//     workgroups early in the sampling kernel cost +0.9 us at its end, wherever in the kernel it was produced; written through, it
//     has left long before.  The next kernel cannot use the line anyway (the per-XCD L2s are not coherent: its consumers may sit
//     on another XCD and read memory / the memory-side cache).
//   * It also keeps outputs nobody re-reads (the 4 MB dense gradient) from pushing the operands out of the 4 MB L2.
// The s_nop is the gfx9 store-data hazard: a VMEM store of more than 8 bytes still reads its data registers for two cycles
// after issue, and the compiler's hazard recognizer does not look inside inline asm -- without it the next VALU write into one of
// those registers reached memory instead of the value.
template <typename V16>
__device__ __forceinline__ void store16_wt(void *p, const V16 &v) {
  static_assert(sizeof(V16) == 16, "16-byte vector");
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t r = __builtin_bit_cast(u32x4_t, v);   // (HIP's float4 is a struct: the asm wants a register tuple)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(r) : "memory");
}

__device__ __forceinline__ void store4_wt(float *p, float v) {   // (4-byte stores are outside the store-data hazard)
  asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// timeline stamps: region `kind` (0 mean-field / sample, 1 vjp, 2 dense) x 4096 blocks x 8 slots
// Developer instrumentation exists only in -DMIVI_DEV builds (csrc/Makefile: `make DEV=1`): the per-workgroup timeline stamps
// (mivi_debug_timeline) and the work-skipping knock-outs (MIVI_KNOCK) are compiled OUT of the release library -- no environment
// variable can make it skip operand loads, MFMAs, epilogues or stores.
#ifdef MIVI_DEV
#define MIVI_STAMP_K(dbgp, kind, slot) do { if ((dbgp) && threadIdx.x == 0 && blockIdx.x < 4096) (dbgp)[((size_t)(kind) * 4096 + blockIdx.x) * 8 + (slot)] = wall_clock64(); } while (0)
#define MIVI_KNOCKED(args, bits) ((args).knock & (bits))
#define MIVI_DEV_ONLY(...) __VA_ARGS__
#else
#define MIVI_STAMP_K(dbgp, kind, slot) do { } while (0)
#define MIVI_KNOCKED(args, bits) false
#define MIVI_DEV_ONLY(...)
#endif
#define MIVI_STAMP(dbgp, slot) MIVI_STAMP_K(dbgp, 0, slot)

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain outstanding global
// stores (vmcnt), so fire-and-forget stores do not put an HBM round trip on a loop's critical path.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ uint64_t rng_index(const RngArgs &r) {
  return r.idx_base + (r.idx_ptr ? *r.idx_ptr : 0ull);
}

__device__ __forceinline__ double direct_entropy_coeff(int ent_kind) {
  // coefficient of diag(1/C_ii) in d(entropy estimate)/dC  
  switch (ent_kind) {
    case MIVI_ENT_CLOSED_FORM: return 1.0;
    case MIVI_ENT_CLOSED_FORM_ZERO_GRAD: return 0.0;
    case MIVI_ENT_MONTE_CARLO: return 1.0;
    case MIVI_ENT_STL: return 0.0;
    default: return -1.0;  // STL zero-gradient
  }
}

__device__ __forceinline__ bool ent_is_stl(int k) { return k == MIVI_ENT_STL || k == MIVI_ENT_STL_ZERO_GRAD; }
__device__ __forceinline__ bool ent_is_closed(int k) {
  return k == MIVI_ENT_CLOSED_FORM || k == MIVI_ENT_CLOSED_FORM_ZERO_GRAD;
}

// A gradient entry from its row sum: -(1/M) sum, the scale rows also carry the entropy term -direct / sigma.  ONE body, compiled without
// fused multiply-add contraction, for every kernel that writes mean-field gradient entries (k_mf_main, k_mf_colreduce, the launch-free
// loops): they must agree to the bit, and left to the optimiser one instantiation fused the product into the subtraction and another
// did not (1 ulp apart in f64; found on the GPU).
template <typename T>
__device__ __forceinline__ T mf_grad_entry(double row_sum, double invM, bool scale_row, double direct, double sigma) {
#pragma clang fp contract(off)
  const double m = -row_sum * invM;
  const double e = direct / sigma;
  return (T)(scale_row ? m - e : m);
}

template <bool ATOMIC>
__device__ __forceinline__ double ld_f64(const double *p) {
  if (ATOMIC) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}

// Row 0 and the column-only part of ell of the fused funnel target (see FunnelFin): e1 / eps_0 per column from the eps stream, the
// A / B partials of the main kernel in index order.  Per-THREAD shares of sum_m ell_m, sum_m w_0m and sum_m w_0m eps_0m (the
// caller reduces them together with its other sums and writes the two gradient / shard-partial entries of row 0;
// kernels_targets.hip k_col_target / oracle FunnelStackedTarget).
template <typename T, int NT, bool ATOMIC = false>
__device__ __forceinline__ void funnel_finish(int d, const FunnelFin &f, const OutArgs &out, double &s_ell, double &sW, double &sWe) {
  const int tid = threadIdx.x;
  const T *params = (const T *)f.params;
  const double mu0 = f.row0 ? (double)__hip_atomic_load((const T *)f.row0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (double)params[0];
  const double sg0 = f.row0 ? (double)__hip_atomic_load((const T *)f.row0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (double)params[d];
  const uint64_t idx = rng_index(f.rng);
  const bool stl = ent_is_stl(out.ent_kind);
  const double n = (double)(d - 1), sv2 = f.sigma_v * f.sigma_v;
  // (floating-point contraction off, the one intended fused multiply-add explicit: this body is instantiated inside several kernels, and
  //  what the optimiser would fuse differed between them -- the row-0 gradient of k_mf_funnel_sgd_loop came out one ulp from the single
  //  calls' once in a few steps)
  for (int m = tid; m < f.M; m += NT) {
#pragma clang fp contract(off)
    T e[4];
    eps_block<T>(f.rng.seed, idx, (uint64_t)(f.rng.m_offset + m) * (uint64_t)f.d4, e);   // row-quad 0 of column m
    const double e0 = (double)e[0];
    const double e1 = (double)fma((T)sg0, e[0], (T)mu0);   // funnel_column's e1: one fused multiply-add in T
    s_ell += (-e1 - e1 * e1 / (2.0 * sv2)) + (-n * e1) + e1;
    const double w = (-1.0 - e1 / sv2) + (-n) + 1.0 + (stl ? e0 / sg0 : 0.0);
    sW += w;
    sWe += w * e0;
  }
  if (f.wait_word) {   // the other workgroups' partials of this step (k_mf_funnel_sgd_loop): fresh addresses, complete behind this wait
    for (int i = tid; i < f.wait_n; i += NT) {
      int budget = f.wait_budget;
      while ((int)(__hip_atomic_load(f.wait_word + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - f.wait_val) < 0) {
        if (--budget <= 0) { if (out.status) atomicOr(out.status, 8); break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    asm volatile("" ::: "memory");
    if (f.mirror) {
      constexpr int UN = 8;
      for (int base = 0; base < f.mirror_n; base += UN * NT) {
        double v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int i = base + u * NT + tid;
          v[u] = i < f.mirror_n ? f.mirror_src[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int i = base + u * NT + tid;
          if (i < f.mirror_n) f.mirror[i] = v[u];
        }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < f.n_part; i += NT) {   // (ATOMIC: partials left by other workgroups of the SAME launch)
    sW += ld_f64<ATOMIC>(f.ab + i);
    sWe += ld_f64<ATOMIC>(f.ab + f.n_part + i);
  }
}

// Entry idx of this rank's partial vector (partials mode): into the ring slot out.partials, or -- peer-to-peer route, direct mode -- straight
// into its OWNER's staging area: owner s = idx / n, place ((parity G_V + v) R + rank) n + (idx - s n) there (kernels_p2p.hip's layout), as a
// system-scope write-through store.  f32 only (the second-generation full-rank kernels).
struct PartialDst {
  float *flat;
  const P2PDirectTab *tab;
  long long n, off;
  float inv_n;
};
__device__ __forceinline__ PartialDst partial_dst(const OutArgs &out) {
  PartialDst pd;
  pd.flat = (float *)out.partials;
  pd.tab = (const P2PDirectTab *)out.p2p_direct;
  pd.n = 1; pd.off = 0; pd.inv_n = 1.f;
  if (pd.tab) {
    const unsigned epoch = __hip_atomic_load(pd.tab->ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u + (unsigned)out.p2p_gi;
    pd.n = pd.tab->n;
    pd.off = ((long long)((int)(epoch & 1u) * pd.tab->GV + out.p2p_v) * pd.tab->R + pd.tab->rank) * pd.n;
    pd.inv_n = 1.f / (float)pd.n;
  }
  return pd;
}
__device__ __forceinline__ void partial_store(const PartialDst &pd, long long idx, float v) {
  if (!pd.tab) { pd.flat[idx] = v; return; }
  int s = (int)((float)idx * pd.inv_n);
  if (s > pd.tab->R - 1) s = pd.tab->R - 1;
  while ((long long)s * pd.n > idx) --s;
  while ((long long)(s + 1) * pd.n <= idx) ++s;
  float *dst = (float *)pd.tab->stage[s] + pd.off + (idx - (long long)s * pd.n);
  __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One whole workgroup (NT threads) assembles the objective value (or the two scalar partials).
//   value = -( sum_ell / M_total + entropy_estimate )
//   closed-form estimators: d/2 (1 + log 2pi) + sum_i log C_ii            location_scale.jl:52-57
//   MC / STL estimators   : mean_m 0.5|eps_m|^2 + d/2 log 2pi + sum_i log C_ii   (C^-1 (z_m - mu) == eps_m)
// `scale_diag(i)` returns C_ii. `red` holds 6 * NT/64 doubles.
// FUNNEL: also finish the fused funnel target (funnel_finish).  Only k_value_funnel and the funnel instantiation of k_mf_main
// carry it: the other kernels never take that path.
template <typename T, int NT, bool ATOMIC, bool FUNNEL = false, typename DiagFn>
__device__ void finalize_value_block(int d, const ValueIn &vin, const OutArgs &out, int64_t plen, DiagFn scale_diag,
                                     double *red) {
  const int tid = threadIdx.x;
  double s_ell = 0.0, s_he = 0.0, s_ld = 0.0, bad = 0.0, sW = 0.0, sWe = 0.0;
  const bool funnel = FUNNEL && vin.fn.ab;
  if (funnel) funnel_finish<T, NT, ATOMIC>(d, vin.fn, out, s_ell, sW, sWe);
  for (int i = tid; i < vin.n_ell_part; i += NT) s_ell += ld_f64<ATOMIC>(vin.ell_part + i);
  for (int i = tid; i < vin.n_ell_part2; i += NT) s_ell += ld_f64<ATOMIC>(vin.ell_part2 + i);
  for (int i = tid; i < vin.n_ell; i += NT) s_ell += (double)((const T *)vin.ell)[i];
  for (int i = tid; i < vin.n_he_part; i += NT) s_he += ld_f64<ATOMIC>(vin.he_part + i);
  if (!out.partials_mode) {
    if (vin.ld_part) {   // per-workgroup partials of sum_i log C_ii and of the non-positive-diagonal count
      for (int i = tid; i < vin.n_ld_part; i += NT) {
        s_ld += ld_f64<ATOMIC>(vin.ld_part + i);
        bad += ld_f64<ATOMIC>(vin.ld_part + vin.n_ld_part + i);
      }
    } else {
      for (int i = tid; i < d; i += NT) {
        const double c = (double)scale_diag(i);
        if (!(c > 0.0)) bad = 1.0;
        s_ld += log(c);
      }
    }
  }
  if (FUNNEL) {
    double v[6] = {s_ell, s_he, s_ld, bad, sW, sWe};
    block_sum_n<double, NT, 6>(v, red);
    s_ell = v[0]; s_he = v[1]; s_ld = v[2]; bad = v[3]; sW = v[4]; sWe = v[5];
    if (funnel && tid == 0) {   // row 0 of the fused funnel target
      const double sg0 = vin.fn.row0 ? (double)__hip_atomic_load((const T *)vin.fn.row0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                     : (double)((const T *)vin.fn.params)[d];
      if (out.partials_mode) {
        T *p = (T *)out.partials;
        p[0] = (T)sW;
        p[d] = (T)sWe;
      } else {
        T *gr = (T *)out.grad;
        const double invM = 1.0 / (double)out.M_total;
        gr[0] = mf_grad_entry<T>(sW, invM, false, 0.0, 1.0);   // (one body without contraction: every instantiation of this block must
        gr[d] = mf_grad_entry<T>(sWe, invM, true, direct_entropy_coeff(out.ent_kind), sg0);   //  round these two entries identically)
      }
    }
  } else {
    double v[4] = {s_ell, s_he, s_ld, bad};
    block_sum_n<double, NT, 4>(v, red);
    s_ell = v[0]; s_he = v[1]; s_ld = v[2]; bad = v[3];
  }
  if (tid == 0) {
    const double sum_ell = s_ell + (double)out.M_local * vin.ell_const;
    if (out.partials_mode) {
      (void)plen;
      if (out.p2p_direct) {   // (f32 by construction: see PartialDst)
        const PartialDst pd = partial_dst(out);
        partial_store(pd, out.scalars_off, (float)sum_ell);
        partial_store(pd, out.scalars_off + 1, (float)s_he);
      } else {
        T *p = (T *)out.partials;
        p[out.scalars_off] = (T)sum_ell;
        p[out.scalars_off + 1] = (T)s_he;
      }
    } else {
      const double Mt = (double)out.M_total;
      const double ent = (ent_is_closed(out.ent_kind) ? 0.5 * d * (1.0 + kLog2Pi) : s_he / Mt + 0.5 * d * kLog2Pi) + s_ld;
      const double value = -(sum_ell / Mt + ent);
      *(T *)out.value = (T)value;
      int st = 0;
      if (!isfinite(value)) st |= 1;
      if (bad > 0.0) st |= 2;
      if (st && out.status) atomicOr(out.status, st);
      if (out.elbo_rec) out.elbo_rec[out.rec_slot] = -value;
    }
  }
}

}  // namespace mivi
