// Developer probe: what does ds_read_b64_tr_b16 return?  Every LDS halfword holds its own index; each lane supplies an address; print what each lane gets.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const unsigned *addr_in, unsigned long long *out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const unsigned a = (unsigned)(uintptr_t)lds + addr_in[threadIdx.x];
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  out[threadIdx.x] = v;
}
int main() {
  unsigned h[64]; unsigned long long o[64];
  unsigned *d; unsigned long long *dout;
  hipMalloc(&d, sizeof h); hipMalloc(&dout, sizeof o);
  for (int mode = 0; mode < 3; ++mode) {
    // mode 0: lane l -> byte address 8 l (linear).  mode 1: lane l -> 8 * (63 - l) (reversed).  mode 2: lane l -> row stride 64 B: ((l & 15) >> 2) * 64 + (l & 3) * 8 + (l >> 4) * 1024
    for (int l = 0; l < 64; ++l) h[l] = mode == 0 ? 8 * l : (mode == 1 ? 8 * (63 - l) : (((l & 15) >> 2) * 64 + (l & 3) * 8 + (l >> 4) * 1024));
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout);
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d (addr %4u = half %4u): %4llu %4llu %4llu %4llu\n", l, h[l], h[l] / 2, o[l] & 0xffff, (o[l] >> 16) & 0xffff, (o[l] >> 32) & 0xffff, (o[l] >> 48) & 0xffff);
    }
  }
  return 0;
}
