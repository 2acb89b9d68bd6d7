// Micro-benchmark (developer tool): how fast does one wave per SIMD issue v_mfma_f32_32x32x2_f32 when its operands come
// from LDS in different software-pipelining shapes?  Prints shader cycles per MFMA and the shader clock.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma.hip -o /tmp/ubench_mfma && /tmp/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define N_STAGE 32   // "stages" of 16 MFMAs

template <int VAR>
__global__ __launch_bounds__(256) void k(float *out, long long *clk, int nwaves_work) {
  __shared__ __attribute__((aligned(16))) float lds[2 * (32 * 64 + 64 * 32)];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  for (int i = tid; i < 2 * (32 * 64 + 64 * 32); i += 256) lds[i] = (float)((i * 7 + 3) % 13) * 0.01f;
  __syncthreads();
  f32x16 acc, acc2;
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
  const float *As = lds + (tid >> 6 & 1) * 32 + l31;   // As[k][64]
  const float *Bs = lds + 32 * 64 + l31 * 32;          // Bs[n][32]
  const long long c0 = clock64();
  const long long w0 = wall_clock64();
  if (VAR == 0) {   // pure chain, operands in registers
    float a = As[0], b = Bs[0];
    for (int s = 0; s < N_STAGE; ++s) {
#pragma unroll
      for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  } else if (VAR == 1) {   // as the compiler schedules the straightforward loop
    for (int s = 0; s < N_STAGE; ++s) {
      const float *A2 = As + (s & 1) * 4096, *B2 = Bs + (s & 1) * 4096;
#pragma unroll
      for (int s8 = 0; s8 < 4; ++s8) {
        const int kb = 8 * s8 + 4 * h;
        const f32x4 bq = *(const f32x4 *)(B2 + kb);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A2[(kb + j) * 64], bq[j], acc, 0, 0, 0);
      }
    }
  } else if (VAR == 2) {   // all 16 A + 4 B fragment reads of a stage up front, then 16 MFMAs
    for (int s = 0; s < N_STAGE; ++s) {
      const float *A2 = As + (s & 1) * 4096, *B2 = Bs + (s & 1) * 4096;
      float av[16];
      f32x4 bq[4];
#pragma unroll
      for (int s8 = 0; s8 < 4; ++s8) {
        bq[s8] = *(const f32x4 *)(B2 + 8 * s8 + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) av[4 * s8 + j] = A2[(8 * s8 + 4 * h + j) * 64];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s8 = 0; s8 < 4; ++s8)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[4 * s8 + j], bq[s8][j], acc, 0, 0, 0);
    }
  } else if (VAR == 3) {   // register double buffer: fragments of stage s+1 are read while stage s's MFMAs run
    float av[16], aw[16];
    f32x4 bq[4], bw[4];
    auto rd = [&](int s, float (&a_)[16], f32x4 (&b_)[4]) {
      const float *A2 = As + (s & 1) * 4096, *B2 = Bs + (s & 1) * 4096;
#pragma unroll
      for (int s8 = 0; s8 < 4; ++s8) {
        b_[s8] = *(const f32x4 *)(B2 + 8 * s8 + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) a_[4 * s8 + j] = A2[(8 * s8 + 4 * h + j) * 64];
      }
    };
    rd(0, av, bq);
    for (int s = 0; s < N_STAGE; s += 2) {
      rd(s + 1, aw, bw);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bq[i >> 2][i & 3], acc, 0, 0, 0);
      rd(s + 2, av, bq);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[i], bw[i >> 2][i & 3], acc, 0, 0, 0);
    }
  } else if (VAR == 4) {   // two accumulators (two independent chains), straightforward reads
    for (int s = 0; s < N_STAGE; ++s) {
      const float *A2 = As + (s & 1) * 4096, *B2 = Bs + (s & 1) * 4096;
#pragma unroll
      for (int s8 = 0; s8 < 4; ++s8) {
        const int kb = 8 * s8 + 4 * h;
        const f32x4 bq = *(const f32x4 *)(B2 + kb);
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A2[(kb + j) * 64], bq[j], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(A2[(kb + j + 1) * 64], bq[j + 1], acc2, 0, 0, 0);
        }
      }
    }
  } else if (VAR == 5) {   // 16x16x4 MFMAs, 4 independent accumulators (same flops per stage: 64 x 16x16x4)
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v c[4];
    for (int q = 0; q < 4; ++q) c[q] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float a = As[0], b = Bs[0];
    for (int s = 0; s < N_STAGE; ++s) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[q], 0, 0, 0);
      }
    }
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 4; ++r) acc[4 * q + r] = c[q][r];
  }
  const long long c1 = clock64();
  const long long w1 = wall_clock64();
  float sacc = 0.f;
  for (int r = 0; r < 16; ++r) sacc += acc[r] + acc2[r];
  out[blockIdx.x * 256 + tid] = sacc;
  if (lane == 0) {
    clk[(blockIdx.x * 4 + (tid >> 6)) * 2] = c1 - c0;
    clk[(blockIdx.x * 4 + (tid >> 6)) * 2 + 1] = w1 - w0;
  }
}

template <int VAR>
void run(const char *name, int grid) {
  float *out;
  long long *clk;
  hipMalloc(&out, grid * 256 * 4);
  hipMalloc(&clk, grid * 4 * 2 * 8);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<VAR>, dim3(grid), dim3(256), 0, 0, out, clk, 4);
  hipDeviceSynchronize();
  std::vector<long long> h(grid * 8);
  hipMemcpy(h.data(), clk, grid * 8 * 8, hipMemcpyDeviceToHost);
  double cyc = 0, wall = 0;
  for (int i = 0; i < grid * 4; ++i) { cyc += h[2 * i]; wall += h[2 * i + 1]; }
  cyc /= grid * 4; wall /= grid * 4;
  const int nm = N_STAGE * 16;
  printf("%-46s grid %4d: %7.1f cycles / MFMA-equivalent (%.0f cycles, %.2f us wall => shader clock %.2f GHz)\n", name, grid,
         cyc / nm, cyc, wall * 0.01, cyc / (wall * 10.0) );
  hipFree(out);
  hipFree(clk);
}

int main() {
  for (int grid : {256, 512}) {
    run<0>("0 pure chain (registers)", grid);
    run<1>("1 LDS reads, compiler schedule", grid);
    run<2>("2 LDS reads: whole stage up front + wait", grid);
    run<3>("3 LDS reads: register double buffer", grid);
    run<4>("4 two accumulators", grid);
    run<5>("5 16x16x4, four accumulators (registers)", grid);
  }
  return 0;
}
