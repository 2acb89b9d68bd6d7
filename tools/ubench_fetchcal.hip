// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS library's access patterns (MI355X_MICROARCH.md, HBM: "FETCH_SIZE
// reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths are uncalibrated: calibrate on a known
// byte count in your own access pattern").  Three kernels that each move a KNOWN number of bytes once, from a buffer far larger than
// the 256 MiB Infinity Cache:
//   k_cal_lds16   16 B / lane direct-to-LDS loads (global_load_lds_dwordx4: the staging of k_fr_prod32 / k_fr_vjp32 / k_stl_*)
//   k_cal_ld16    16 B / lane register loads (global_load_dwordx4: the logistic-regression kernels, the epilogues)
//   k_cal_ld4      4 B / lane loads (the first-generation tile kernels, scalar epilogue reads)
// and one store kernel (16 B / lane write-through stores, store16_wt's flavour).  Run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/bin/ubench_fetchcal.exe      (and again with --pmc WRITE_SIZE)
// tools/pmc_calibrate.py divides the counter by the bytes printed here -> profiles/pmc_calibration.json.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_fetchcal.hip -o tools/bin/ubench_fetchcal.exe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_cal_lds16(const float *src, size_t n_vec, float *sink) {
  __shared__ __attribute__((aligned(16))) float lds[256 * 4 * 4];
  const size_t stride = (size_t)gridDim.x * 256;
  float acc = 0.f;
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n_vec; v += 4 * stride) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t vv = v + u * stride < n_vec ? v + u * stride : v;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 4 * vv),
                                       (__attribute__((address_space(3))) void *)(lds + u * 1024 + (threadIdx.x & ~63) * 4), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += lds[threadIdx.x];
  }
  if (acc == 12345.678f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_cal_ld16(const float *src, size_t n_vec, float *sink) {
  const size_t stride = (size_t)gridDim.x * 256;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n_vec; v += stride) acc += *(const f32x4 *)(src + 4 * v);
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
__global__ __launch_bounds__(256) void k_cal_ld4(const float *src, size_t n, float *sink) {
  const size_t stride = (size_t)gridDim.x * 256;
  float acc = 0.f;
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n; v += stride) acc += src[v];
  if (acc == 12345.678f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_cal_st16(float *dst, size_t n_vec) {
  const size_t stride = (size_t)gridDim.x * 256;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 r = {1u, 2u, 3u, 4u};
  for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < n_vec; v += stride)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst + 4 * v), "v"(r) : "memory");
}

int main() {
  const size_t bytes = (size_t)2 << 30;   // 2 GiB: eight times the Infinity Cache
  float *buf = nullptr, *sink = nullptr;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 0, bytes);
  hipDeviceSynchronize();
  const size_t n_vec = bytes / 16;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_cal_lds16, dim3(2048), dim3(256), 0, 0, buf, n_vec, sink);
    hipLaunchKernelGGL(k_cal_ld16, dim3(2048), dim3(256), 0, 0, buf, n_vec, sink);
    hipLaunchKernelGGL(k_cal_ld4, dim3(2048), dim3(256), 0, 0, buf, bytes / 4, sink);
    hipLaunchKernelGGL(k_cal_st16, dim3(2048), dim3(256), 0, 0, buf, n_vec);
  }
  hipDeviceSynchronize();
  printf("{\"bytes_per_launch\": %zu}\n", bytes);
  return 0;
}
