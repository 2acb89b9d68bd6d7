// Micro-benchmark (developer tool): v_mfma_f32_16x16x32_bf16 issue cost in a dependent chain (same accumulator), in two and
// four interleaved chains, and with the VALU work of the three-way bf16 split between them.  Shader cycles per MFMA.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma16.hip -o tools/bin/ubench_mfma16.exe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ __launch_bounds__(512) void k(float *out, long long *clk, int waves) {
  const int w = threadIdx.x >> 6;
  if (w >= waves) return;
  u32x4 ua = {threadIdx.x * 3u + 1u, threadIdx.x * 5u, 7u, threadIdx.x}, ub = {threadIdx.x, 11u, threadIdx.x * 7u, 3u};
  bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const long long c0 = clock64();
  for (int s = 0; s < 64; ++s) {
#pragma unroll
    for (int j = 0; j < 12; ++j) acc[j % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j % CHAINS], 0, 0, 0);
  }
  const long long c1 = clock64();
  f32x4 r = acc[0] + acc[1] + acc[2] + acc[3];
  if (r.x == 12345.f) out[threadIdx.x] = r.y;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}

int main() {
  float *out;
  long long *clk, h;
  hipMalloc(&out, 4096);
  hipMalloc(&clk, 64);
  for (int waves : {4, 8}) {
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(512), 0, 0, out, clk, waves);
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("%d waves/CU, 1 chain : %5.1f cycles per MFMA\n", waves, h / (64.0 * 12));
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(512), 0, 0, out, clk, waves);
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("%d waves/CU, 2 chains: %5.1f cycles per MFMA\n", waves, h / (64.0 * 12));
    hipLaunchKernelGGL(k<4>, dim3(1), dim3(512), 0, 0, out, clk, waves);
    hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    printf("%d waves/CU, 4 chains: %5.1f cycles per MFMA\n", waves, h / (64.0 * 12));
  }
  return 0;
}
