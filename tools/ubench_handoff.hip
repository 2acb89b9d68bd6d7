// Micro-benchmark (developer tool): producer -> consumer hand-off between workgroups INSIDE one kernel, same XCD or across XCDs
// (each XCD has its own L2; the L2s are not coherent with each other).  The question behind the dataflow-fused estimator kernel:
// which store / flag / load flavours make a 32 KiB block written by one workgroup visible to another, how long the hand-off takes,
// and what the flavour costs the consumer's LDS-DMA reads.
//   producer: 8 x 16-byte stores per thread (flavour ST), s_waitcnt vmcnt(0), barrier, flag (atomic add, agent scope)
//   consumer: poll the flag (load sc1), barrier, LDS-DMA the block (flavour LD), verify every word, acknowledge
// Every spin is bounded (a lost hand-off is reported, the kernel never hangs).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_handoff.hip -o tools/bin/ubench_handoff.exe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { F_PLAIN = 0, F_SC1 = 1, F_SC01 = 2, F_SC0 = 3 };

template <int ST>
__device__ __forceinline__ void store16(void *p, u32x4 v) {
  if (ST == F_PLAIN) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
  else if (ST == F_SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else if (ST == F_SC0) asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
template <int LD>
__device__ __forceinline__ void glds16(const void *g, void *l) {
  constexpr int aux = LD == F_PLAIN ? 0 : (LD == F_SC1 ? 16 : (LD == F_SC0 ? 1 : 17));
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0,
                                   aux);
}
__device__ __forceinline__ unsigned poll_sc1(const unsigned *p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }   // s_memtime: one clock for the chip? (checked below)
__device__ __forceinline__ unsigned long long wall() { return wall_clock64(); }                // 100 MHz, chip-wide

struct Args {
  float *data;          // [npairs][iters or 1][8192]
  unsigned *flag, *ack; // [npairs] (64-byte spaced)
  int npairs, coff, iters;
  size_t iter_stride;   // floats (0: the same block every iteration -> the consumer's L2 holds last iteration's lines)
  long long *t_flag, *t_seen, *t_read, *t_st;   // [npairs][iters]
  unsigned *err;        // [npairs]: mismatching words | 0x80000000 lost
  unsigned *xcc;        // [2 npairs]
};

template <int ST, int LD>
__global__ __launch_bounds__(256) void k_handoff(Args a) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  __shared__ unsigned s_ok;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int b = blockIdx.x;
  if (tid == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    a.xcc[b] = id;
  }
  if (b < a.npairs) {   // producer of pair b
    float *blk = a.data + (size_t)b * (a.iter_stride ? a.iter_stride * a.iters : 8192);
    for (int it = 1; it <= a.iters; ++it) {
      if (tid == 0) {
        int budget = 1 << 20;
        while (poll_sc1(a.ack + 16 * b) < (unsigned)(it - 1) && --budget) __builtin_amdgcn_s_sleep(2);
        s_ok = budget > 0;
      }
      __syncthreads();
      if (!s_ok) return;
      float *dst = blk + (size_t)(it - 1) * a.iter_stride;
      if (tid == 0) a.t_st[(size_t)b * a.iters + it - 1] = (long long)wall();
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const unsigned idx = (unsigned)(p * 1024 + tid * 4);
        const unsigned base = (unsigned)it * 2654435761u + (unsigned)b * 40503u + idx;
        u32x4 v = {base, base + 1, base + 2, base + 3};
        store16<ST>(dst + idx, v);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        a.t_flag[(size_t)b * a.iters + it - 1] = (long long)wall();
        __hip_atomic_fetch_add(a.flag + 16 * b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
  const int pr = (b - a.npairs + a.coff) % a.npairs;   // consumer of pair pr
  const float *blk = a.data + (size_t)pr * (a.iter_stride ? a.iter_stride * a.iters : 8192);
  unsigned bad = 0;
  for (int it = 1; it <= a.iters; ++it) {
    if (tid == 0) {
      int budget = 1 << 20;
      while (poll_sc1(a.flag + 16 * pr) < (unsigned)it && --budget) __builtin_amdgcn_s_sleep(1);
      s_ok = budget > 0;
      a.t_seen[(size_t)pr * a.iters + it - 1] = (long long)wall();
    }
    __syncthreads();
    if (!s_ok) {
      if (tid == 0) a.err[pr] = bad | 0x80000000u;
      return;
    }
    const float *src = blk + (size_t)(it - 1) * a.iter_stride;
#pragma unroll
    for (int p = 0; p < 8; ++p) glds16<LD>(src + (w * 8 + p) * 256 + lane * 4, lds + (w * 8 + p) * 256);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) a.t_read[(size_t)pr * a.iters + it - 1] = (long long)wall();
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const unsigned idx = (unsigned)(p * 1024 + tid * 4);
      const unsigned base = (unsigned)it * 2654435761u + (unsigned)pr * 40503u + idx;
      const u32x4 v = *(const u32x4 *)(lds + idx);
      bad += (v.x != base) + (v.y != base + 1) + (v.z != base + 2) + (v.w != base + 3);
    }
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(a.ack + 16 * pr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
  if (lane == 0 && bad) atomicAdd(a.err + pr, bad);
}

template <int ST, int LD>
void run(const char *name, int npairs, int coff, int iters, bool fresh) {
  Args a{};
  a.npairs = npairs;
  a.coff = coff;
  a.iters = iters;
  a.iter_stride = fresh ? 8192 : 0;
  const size_t nd = (size_t)npairs * (fresh ? iters : 1) * 8192;
  hipMalloc(&a.data, nd * 4);
  hipMemset(a.data, 0, nd * 4);
  hipMalloc(&a.flag, npairs * 64);
  hipMalloc(&a.ack, npairs * 64);
  hipMemset(a.flag, 0, npairs * 64);
  hipMemset(a.ack, 0, npairs * 64);
  const size_t nt = (size_t)npairs * iters;
  hipMalloc(&a.t_flag, nt * 8);
  hipMalloc(&a.t_seen, nt * 8);
  hipMalloc(&a.t_read, nt * 8);
  hipMalloc(&a.t_st, nt * 8);
  hipMalloc(&a.err, npairs * 4);
  hipMemset(a.err, 0, npairs * 4);
  hipMalloc(&a.xcc, 2 * npairs * 4);
  hipLaunchKernelGGL((k_handoff<ST, LD>), dim3(2 * npairs), dim3(256), 0, 0, a);
  hipError_t e = hipDeviceSynchronize();
  std::vector<long long> tf(nt), ts(nt), tr(nt), t0(nt);
  std::vector<unsigned> err(npairs), xcc(2 * npairs);
  hipMemcpy(tf.data(), a.t_flag, nt * 8, hipMemcpyDeviceToHost);
  hipMemcpy(ts.data(), a.t_seen, nt * 8, hipMemcpyDeviceToHost);
  hipMemcpy(tr.data(), a.t_read, nt * 8, hipMemcpyDeviceToHost);
  hipMemcpy(t0.data(), a.t_st, nt * 8, hipMemcpyDeviceToHost);
  hipMemcpy(err.data(), a.err, npairs * 4, hipMemcpyDeviceToHost);
  hipMemcpy(xcc.data(), a.xcc, 2 * npairs * 4, hipMemcpyDeviceToHost);
  unsigned long long bad = 0, lost = 0;
  for (unsigned v : err) {
    bad += v & 0x7fffffffu;
    lost += v >> 31;
  }
  double hand = 0, rd = 0, ack = 0;
  for (size_t i = 0; i < nt; ++i) {
    hand += (double)(ts[i] - tf[i]);
    rd += (double)(tr[i] - ts[i]);
    ack += (double)(tf[i] - t0[i]);
  }
  int same = 0, map_ok = 0;
  for (int p = 0; p < npairs; ++p) {
    const int cb = npairs + ((p - coff) % npairs + npairs) % npairs;
    same += xcc[p] == xcc[cb];
    map_ok += (int)xcc[p] == p % 8;
  }
  printf("%-34s pairs %3d off %d %s: err %s bad words %llu lost %llu | store+ack %.0f ns, hand-off %.0f ns, 32 KiB read %.0f ns | same-XCD pairs %d, xcc==b%%8 %d/%d\n",
         name, npairs, coff, fresh ? "fresh blocks" : "same block  ", hipGetErrorString(e), bad, lost, 10.0 * ack / nt, 10.0 * hand / nt, 10.0 * rd / nt, same,
         map_ok, npairs);
  hipFree(a.data); hipFree(a.flag); hipFree(a.ack); hipFree(a.t_flag); hipFree(a.t_seen); hipFree(a.t_read); hipFree(a.t_st); hipFree(a.err); hipFree(a.xcc);
}

int main() {
  for (int coff : {0, 1}) {
    for (int fresh : {1, 0}) {
      run<F_PLAIN, F_PLAIN>("store plain, load plain", 128, coff, 50, fresh);
      run<F_SC1, F_PLAIN>("store sc1,   load plain", 128, coff, 50, fresh);
      run<F_SC1, F_SC1>("store sc1,   load sc1", 128, coff, 50, fresh);
      run<F_SC01, F_SC01>("store sc0sc1, load sc0sc1", 128, coff, 50, fresh);
      run<F_PLAIN, F_SC1>("store plain, load sc1", 128, coff, 50, fresh);
      run<F_PLAIN, F_SC0>("store plain, load sc0", 128, coff, 50, fresh);
      run<F_SC0, F_SC0>("store sc0,   load sc0", 128, coff, 50, fresh);
      run<F_SC1, F_SC0>("store sc1,   load sc0", 128, coff, 50, fresh);
    }
  }
  return 0;
}
