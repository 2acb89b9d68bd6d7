// Microbenchmark / probe for the f16 two-way operand split (round 5): (1) does v_mfma_f32_32x32x16_f16 honour f16 SUBNORMAL inputs,
// (2) accuracy of hi/lo f16 split products (3 MFMAs) against f64 for N(0,1) operands scaled by 2^11, (3) MFMA issue rate f16 vs bf16.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_f16split.hip -o tools/bin/ubench_f16split.exe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// one wave: C(32x32) = A(32x16) B(16x32): lane (row = lane % 32, h = lane / 32) holds k = 8 h .. 8 h + 7 of its row (A) / column (B)
__global__ void k_probe(const _Float16 *A, const _Float16 *B, float *C) {
  const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = A[l31 * 16 + 8 * h + e]; b[e] = B[l31 * 16 + 8 * h + e]; }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[(8 * (r / 4) + 4 * h + r % 4) * 32 + l31] = c[r];
}
// split product: x = hi + lo (f16 each, after scaling by sa / sb); 3 MFMAs; result / (sa sb)
__global__ void k_split(const float *A, const float *B, float *C, float sa, float sb, int K) {   // A: 32 x K row-major, B: 32 (cols) x K
  const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
  f32x16 c = {0};
  for (int k0 = 0; k0 < K; k0 += 16) {
    f16x8 ah, al, bh, bl;
    for (int e = 0; e < 8; ++e) {
      const float x = A[l31 * K + k0 + 8 * h + e] * sa, y = B[l31 * K + k0 + 8 * h + e] * sb;
      const _Float16 xh = (_Float16)x, yh = (_Float16)y;
      ah[e] = xh; al[e] = (_Float16)(x - (float)xh);
      bh[e] = yh; bl[e] = (_Float16)(y - (float)yh);
    }
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, c, 0, 0, 0);
  }
  const float inv = 1.f / (sa * sb);
  for (int r = 0; r < 16; ++r) C[(8 * (r / 4) + 4 * h + r % 4) * 32 + l31] = c[r] * inv;
}
template <int F16>
__global__ __launch_bounds__(256) void k_rate(float *out, int iters) {
  f32x16 c[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  f16x8 a; bf16x8 ab;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); ab[e] = (__bf16)(threadIdx.x * 0.001f + e); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (F16) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c[i], 0, 0, 0);
      else c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, ab, c[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
  if (s == 123.456f) out[0] = s;
}
int main() {
  // (1) subnormal inputs: A = 2^-20 (f16 subnormal: min normal 2^-14), B = 2^10  -> exact product 2^-10 per k, sum over 16 = 2^-6
  std::vector<_Float16> hA(32 * 16), hB(32 * 16);
  for (auto &x : hA) x = (_Float16)ldexpf(1.f, -20);
  for (auto &x : hB) x = (_Float16)1024.f;
  _Float16 *dA, *dB; float *dC;
  hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dC, 32 * 32 * 4);
  hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
  k_probe<<<1, 64>>>(dA, dB, dC);
  float c0; hipMemcpy(&c0, dC, 4, hipMemcpyDeviceToHost);
  printf("subnormal-A probe: got %.9g, exact %.9g  -> f16 subnormal INPUTS %s\n", c0, ldexp(1.0, -6), c0 == (float)ldexp(1.0, -6) ? "HONOURED" : "FLUSHED (or wrong)");
  // (2) accuracy, K = 1024, N(0,1) x N(0,1); scales 2^11 / 2^11 and 1 / 1
  const int K = 1024;
  std::vector<float> fA(32 * K), fB(32 * K);
  srand(1);
  auto nrm = []() { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return (float)(sqrt(-2 * log(u)) * cos(6.283185307179586 * v)); };
  for (auto &x : fA) x = nrm();
  for (auto &x : fB) x = nrm();
  float *gA, *gB; hipMalloc(&gA, fA.size() * 4); hipMalloc(&gB, fB.size() * 4);
  hipMemcpy(gA, fA.data(), fA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(gB, fB.data(), fB.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> hC(32 * 32);
  for (int sc = 0; sc < 3; ++sc) {
    const float s = sc == 0 ? 1.f : (sc == 1 ? 2048.f : 1.f / 1024.f);
    k_split<<<1, 64>>>(gA, gB, dC, s, s, K);
    hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    double num = 0, den = 0, mx = 0, f32num = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double ref = 0; float f32 = 0.f;
      for (int k = 0; k < K; ++k) { ref += (double)fA[i * K + k] * fB[j * K + k]; f32 = fmaf(fA[i * K + k], fB[j * K + k], f32); }
      const double e = hC[i * 32 + j] - ref;
      num += e * e; den += ref * ref; mx = fmax(mx, fabs(e)); f32num += (f32 - ref) * (f32 - ref);
    }
    printf("split product K=%d scale %g: rel l2 err %.3e (plain f32 fma chain: %.3e), max abs %.3e\n", K, s, sqrt(num / den), sqrt(f32num / den), mx);
  }
  // (3) rate
  float *dO; hipMalloc(&dO, 4);
  for (int f = 0; f < 2; ++f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 8;
    if (f) k_rate<1><<<blocks, 256>>>(dO, 100); else k_rate<0><<<blocks, 256>>>(dO, 100);
    hipEventRecord(e0);
    if (f) k_rate<1><<<blocks, 256>>>(dO, iters); else k_rate<0><<<blocks, 256>>>(dO, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * 4 * iters * 4 * 32768.0;
    printf("%s 32x32x16 MFMA rate: %.1f TFLOP/s\n", f ? "f16 " : "bf16", fl / (ms * 1e-3) / 1e12);
  }
  return 0;
}
