// Micro-benchmark (developer tool): cycles per s_barrier / per LDS round trip for a workgroup shaped like the tile kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NT, int LDSF>
__global__ __launch_bounds__(NT) void k(long long *clk, float *out, int iters) {
  __shared__ float lds[LDSF];
  const int tid = threadIdx.x;
  lds[tid] = tid;
  __syncthreads();
  long long c0 = clock64();
  for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_barrier();
  long long c1 = clock64();
  float v = 0.f;
  for (int i = 0; i < iters; ++i) { v += lds[(tid + i) & (NT - 1)]; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
  long long c2 = clock64();
  for (int i = 0; i < iters; ++i) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); v += lds[(tid + i) & (NT - 1)]; }
  long long c3 = clock64();
  out[blockIdx.x * NT + tid] = v;
  if (tid == 0) { clk[blockIdx.x * 3] = c1 - c0; clk[blockIdx.x * 3 + 1] = c2 - c1; clk[blockIdx.x * 3 + 2] = c3 - c2; }
}
template <int NT, int LDSF>
void run(int grid) {
  long long *clk; float *out;
  hipMalloc(&clk, grid * 24); hipMalloc(&out, grid * NT * 4);
  const int iters = 64;
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<NT, LDSF>), dim3(grid), dim3(NT), 0, 0, clk, out, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(grid * 3);
  hipMemcpy(h.data(), clk, grid * 24, hipMemcpyDeviceToHost);
  double a = 0, b = 0, c = 0;
  for (int i = 0; i < grid; ++i) { a += h[3 * i]; b += h[3 * i + 1]; c += h[3 * i + 2]; }
  printf("NT %4d LDS %6d B grid %4d: s_barrier %6.1f cyc   dependent ds_read %6.1f cyc   waitcnt+barrier+ds_read %6.1f cyc\n", NT, LDSF * 4, grid,
         a / grid / iters, b / grid / iters, c / grid / iters);
}
int main() {
  run<256, 1024>(256); run<512, 1024>(256); run<256, 16384>(256); run<512, 16384>(256); run<512, 20480>(256); run<256, 16384>(512);
  return 0;
}
