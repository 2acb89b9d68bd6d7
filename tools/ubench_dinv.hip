// Micro-benchmark (developer tool): shader cycles of stl_dinv64_block (inverse of a 64 x 64 diagonal block by recursive doubling in
// LDS + the plane-split output) with 256 and 512 threads, one workgroup.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I advancedvi.jl_amd/csrc tools/ubench_dinv.hip -o tools/bin/ubench_dinv.exe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "stl_dinv.h"

template <int NT>
__global__ __launch_bounds__(NT) void k(int d, const float *C, unsigned *pack, long long *clk) {
  __shared__ float sm[3 * 64 * 65];
  const long long c0 = clock64();
  mivi::stl_dinv64_block<NT>(d, C, pack, blockIdx.x, sm);
  __syncthreads();
  if (threadIdx.x == 0) clk[blockIdx.x] = clock64() - c0;
}

int main() {
  const int d = 1024;
  std::vector<float> h((size_t)d * d, 0.f);
  for (int j = 0; j < d; ++j)
    for (int i = j; i < d; ++i) h[(size_t)j * d + i] = i == j ? 1.0f + 0.001f * i : 0.01f * ((i * 7 + j * 3) % 11 - 5);
  float *C;
  unsigned *pack;
  long long *clk, hc[16];
  hipMalloc(&C, h.size() * 4);
  hipMalloc(&pack, mivi::stl_pack_units(d) * 4);
  hipMalloc(&clk, 16 * 8);
  hipMemcpy(C, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<256>, dim3(16), dim3(256), 0, 0, d, C, pack, clk);
    hipMemcpy(hc, clk, sizeof hc, hipMemcpyDeviceToHost);
    printf("256 threads: %lld cycles (block 0), %lld (block 15)\n", hc[0], hc[15]);
    hipLaunchKernelGGL(k<512>, dim3(16), dim3(512), 0, 0, d, C, pack, clk);
    hipMemcpy(hc, clk, sizeof hc, hipMemcpyDeviceToHost);
    printf("512 threads: %lld cycles (block 0), %lld (block 15)\n", hc[0], hc[15]);
  }
  return 0;
}
