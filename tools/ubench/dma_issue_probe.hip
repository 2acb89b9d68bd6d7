// Developer probe: what do LDS-DMA requests, LDS fragment reads and MFMAs cost a SIMD when they share it?  One workgroup of 8 waves per CU
// (two waves per SIMD, the batch engine's shape); every wave runs G iterations of {P LDS-DMA pieces of 1 KiB, R ds_read_b128, M MFMAs
// 32x32x16 f16 on two alternating accumulators}, no barrier; the operands come from a 128 KiB L2-resident buffer.  Prints ns and SIMD cycles per
// iteration for a table of (P, R, M, kind): kind 0 = global_load_lds_dwordx4 (64-bit per-lane addresses), 1 = buffer_load_dwordx4 ... lds
// (resource in SGPRs + 32-bit per-lane offsets).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/dma_issue_probe.hip -o tools/bin/dma_issue_probe.exe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int P, int R, int M, int KIND, int NW>
__global__ __launch_bounds__(64 * NW) void k_probe(const unsigned *src, float *out, int G, size_t wg_stride, int span) {
  __shared__ __attribute__((aligned(16))) unsigned lds[NW * 4 * 256 + 8 * 1024];   // per wave: 4 pieces of 1 KiB; + a 32 KiB read area
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned *ring = lds + w * 4 * 256;
  const unsigned *rd = lds + NW * 4 * 256 + 4 * lane;
  const unsigned *g0 = src + (size_t)blockIdx.x * wg_stride + (size_t)w * 256 + 4 * lane;   // wave w's piece of a 8 KiB row; rows 8 KiB apart (span rows, then wrap)
  f32x16 acc0, acc1;
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  u32x4 fa = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}, fb = fa;
  // buffer resource over the whole source buffer
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, 0x7fffffff, 0x00020000);
  const unsigned voff0 = (unsigned)(((size_t)blockIdx.x * wg_stride + (size_t)w * 256) * 4 + 16 * lane);
  unsigned x = 0;
  for (int g = 0; g < G; ++g) {
    if (P > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int piece = ((g * P + p) % span) * 8;   // (a stage = the 8 waves' pieces of P consecutive rows)
      unsigned *dst = ring + ((g * P + p) & 3) * 256;
      if (KIND == 0) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g0 + piece * 256),
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)dst, 16, voff0 + piece * 1024, 0, 0, 0);
      }
    }
    u32x4 t[R > 0 ? R : 1];
#pragma unroll
    for (int r = 0; r < R; ++r) t[r] = *(const u32x4 *)(rd + ((g + r) & 7) * 256);
#pragma unroll
    for (int m = 0; m < M; ++m) {
      if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa), __builtin_bit_cast(f16x8, fb), acc1, 0, 0, 0);
      else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa), __builtin_bit_cast(f16x8, fb), acc0, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) x ^= t[r][0] ^ t[r][3];
    if (R > 0) fa[0] ^= (x & 1u);   // the reads feed the next MFMAs (kept alive)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
  if (s == 12345.f || x == 0x12345u) out[threadIdx.x] = s;
}

template <int P, int R, int M, int KIND, int NW>
void run(const unsigned *src, float *out, const char *name, size_t wg_stride = 0, int span = 16) {
  const int G = 4000, NCU = 256;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int t = 0; t < 5; ++t) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_probe<P, R, M, KIND, NW>), dim3(NCU), dim3(64 * NW), 0, 0, src, out, G, wg_stride, span);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double ns = best * 1e6 / G;
  printf("%-44s waves/CU %2d  P=%d R=%d M=%d: %7.1f ns per iteration = %6.0f cycles at 2.1 GHz (MFMA alone would be %d per SIMD)\n", name, NW, P, R, M, ns, ns * 2.1,
         M * 32 * NW / 4);
}

int main() {
  unsigned *src;
  float *out;
  const size_t total = (size_t)1 << 30;   // 1 GiB
  (void)hipMalloc(&src, total);
  (void)hipMemset(src, 0, total);
  (void)hipMalloc(&out, 4096);
  run<0, 0, 6, 0, 8>(src, out, "MFMA only");
  run<2, 0, 0, 0, 8>(src, out, "DMA only, global_load_lds");
  run<2, 0, 0, 1, 8>(src, out, "DMA only, buffer_load lds");
  run<0, 6, 0, 0, 8>(src, out, "ds_read_b128 only");
  run<2, 0, 6, 0, 8>(src, out, "MFMA + DMA global");
  run<2, 0, 6, 1, 8>(src, out, "MFMA + DMA buffer");
  run<0, 6, 6, 0, 8>(src, out, "MFMA + reads");
  run<2, 6, 6, 0, 8>(src, out, "MFMA + reads + DMA global (the engine's group)");
  run<2, 6, 6, 1, 8>(src, out, "MFMA + reads + DMA buffer");
  run<2, 6, 0, 0, 8>(src, out, "reads + DMA global");
  run<1, 6, 6, 0, 8>(src, out, "MFMA + reads + ONE piece per wave");
  run<2, 6, 6, 0, 16>(src, out, "16 waves: MFMA + reads + DMA global");
  run<0, 0, 6, 0, 16>(src, out, "16 waves: MFMA only");
  run<2, 8, 12, 0, 4>(src, out, "4 waves of 64 x 64: 12 MFMA + 8 reads + 2 pieces");
  run<4, 8, 12, 0, 4>(src, out, "4 waves of 64 x 64: 12 MFMA + 8 reads + 4 pieces");
  // where the operands come from: every workgroup its own region (words): 128 KiB (L2: 32 MiB in all), 1 MiB (256 MiB in all: the memory-side cache),
  // 4 MiB (1 GiB in all: HBM), streamed row by row (span = rows of 8 KiB before the wrap)
  run<2, 0, 0, 0, 8>(src, out, "DMA only, own 128 KiB per workgroup", (size_t)32 * 1024, 16);
  run<2, 6, 6, 0, 8>(src, out, "group, own 128 KiB per workgroup (L2)", (size_t)32 * 1024, 16);
  run<2, 0, 0, 0, 8>(src, out, "DMA only, own 1 MiB per workgroup", (size_t)256 * 1024, 128);
  run<2, 6, 6, 0, 8>(src, out, "group, own 1 MiB per workgroup (MALL)", (size_t)256 * 1024, 128);
  run<2, 0, 0, 0, 8>(src, out, "DMA only, own 4 MiB per workgroup", (size_t)1024 * 1024, 512);
  run<2, 6, 6, 0, 8>(src, out, "group, own 4 MiB per workgroup (HBM)", (size_t)1024 * 1024, 512);
  return 0;
}
