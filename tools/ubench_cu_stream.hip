// Micro-benchmark (developer tool): what one CU can pull from L2 / HBM when only a few workgroups run (the STL chain kernel's
// situation: 16 workgroups, each streaming ~450 KB once).  Variants: plain 16-byte loads vs LDS-DMA, number of waves that issue,
// loads in flight per wave, every workgroup reading the same bytes or its own.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_cu_stream.hip -o tools/bin/ubench_cu_stream.exe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GLDS16(gptr, lptr)                                                                            \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gptr),            \
                                   (__attribute__((address_space(3))) void *)(lptr), 16, 0, 0)

// each issuing wave streams `kb_per_wave` KiB in 1 KiB pieces, DEPTH pieces in flight
template <int DEPTH, bool LDSDMA>
__global__ __launch_bounds__(1024) void k_stream(const float *src, float *out, int waves, int kb_per_wave, size_t wg_stride) {
  __shared__ __attribute__((aligned(16))) float lds[16 * 8 * 256];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (w >= waves) return;
  const float *p = src + blockIdx.x * wg_stride + (size_t)w * kb_per_wave * 256 + lane * 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (LDSDMA) {
    float *ring = lds + w * 8 * 256;
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) GLDS16(p + i * 256, ring + (i % 8) * 256);
    for (int i = 0; i < kb_per_wave; ++i) {
      if (i + DEPTH <= kb_per_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc += *(const f32x4 *)(ring + (i % 8) * 256 + lane * 4);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (i + DEPTH < kb_per_wave) GLDS16(p + (size_t)(i + DEPTH) * 256, ring + ((i + DEPTH) % 8) * 256);
    }
  } else {
    f32x4 buf[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) buf[i] = *(const f32x4 *)(p + i * 256);
    for (int i = 0; i < kb_per_wave; i += DEPTH) {
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) {
        acc += buf[j];
        if (i + j + DEPTH < kb_per_wave) buf[j] = *(const f32x4 *)(p + (size_t)(i + j + DEPTH) * 256);
      }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[threadIdx.x] = acc.x;
}

template <typename F>
double time_us(F launch, hipStream_t st) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int t = 0; t < 6; ++t) {
    hipEventRecord(e0, st);
    for (int r = 0; r < 20; ++r) launch();
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best * 1e3 / 20;
}

int main() {
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  float *buf, *out;
  const size_t total = 256u << 20;
  hipMalloc(&buf, total);
  hipMalloc(&out, 1 << 20);
  hipMemset(buf, 0, total);
  const int KB = 448;
  for (int same = 1; same >= 0; --same) {
    const size_t stride = same ? 0 : (size_t)KB * 256;   // floats: own region (spread wide) or the same bytes
    for (int wgs : {256}) {
      if (!same && (size_t)wgs * KB * 1024 > total) continue;   // (own regions must fit the buffer)
      for (int waves : {4, 8, 16}) {
        const int kbw = KB / waves;
        double t;
        t = time_us([&] { hipLaunchKernelGGL((k_stream<8, false>), dim3(wgs), dim3(1024), 0, st, buf, out, waves, kbw, stride); }, st);
        printf("%s bytes, %2d WGs, %d waves, plain x8 : %6.2f us  -> %6.1f GB/s per CU\n", same ? "same" : "own ", wgs, waves, t, KB * 1024.0 / t / 1e3);
        t = time_us([&] { hipLaunchKernelGGL((k_stream<16, false>), dim3(wgs), dim3(1024), 0, st, buf, out, waves, kbw, stride); }, st);
        printf("%s bytes, %2d WGs, %d waves, plain x16: %6.2f us  -> %6.1f GB/s per CU\n", same ? "same" : "own ", wgs, waves, t, KB * 1024.0 / t / 1e3);
        t = time_us([&] { hipLaunchKernelGGL((k_stream<8, true>), dim3(wgs), dim3(1024), 0, st, buf, out, waves, kbw, stride); }, st);
        printf("%s bytes, %2d WGs, %d waves, glds  x8 : %6.2f us  -> %6.1f GB/s per CU\n", same ? "same" : "own ", wgs, waves, t, KB * 1024.0 / t / 1e3);
        t = time_us([&] { hipLaunchKernelGGL((k_stream<8, true>), dim3(wgs), dim3(1024), 0, st, buf, out, waves, kbw, stride); }, st);
        printf("%s bytes, %2d WGs, %d waves, glds  x8b: %6.2f us  -> %6.1f GB/s per CU\n", same ? "same" : "own ", wgs, waves, t, KB * 1024.0 / t / 1e3);
      }
    }
  }
  return 0;
}
