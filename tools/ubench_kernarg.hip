// Micro-benchmark (developer tool): what the kernel-argument fetch costs at kernel entry, and what SGPR preloading of the leading
// arguments (-mllvm -amdgpu-kernarg-preload-count=N: the command processor delivers the first N dwords of the kernarg segment in
// SGPRs at wave launch) saves.  Build twice and compare "entry -> first data":
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_kernarg.hip -o /tmp/ka0.exe
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=16 tools/ubench_kernarg.hip -o /tmp/ka1.exe
// The kernel is launched inside a hipGraph chain behind a kernel that rewrites the data (so L2 is cold, as between the estimator's
// kernels), 256 workgroups x 512 threads; stamps from wall_clock64 (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(512) void k_touch(float *A, int n) {
  for (int i = blockIdx.x * 512 + threadIdx.x; i < n; i += gridDim.x * 512) A[i] += 1.f;
}
__global__ __launch_bounds__(512) void k_probe(const float *A, int lda, float *out, long long *stamps) {
  const long long t0 = wall_clock64();
  const float v = A[threadIdx.x + (size_t)blockIdx.x * lda];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = wall_clock64();
  if (v == 12345.f) out[threadIdx.x] = v;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t0;
    stamps[2 * blockIdx.x + 1] = t1;
  }
}

int main() {
  const int n = 1 << 20;
  float *A, *out;
  long long *st;
  hipMalloc(&A, n * 4);
  hipMalloc(&out, 4096);
  hipMalloc(&st, 256 * 2 * 8);
  hipMemset(A, 0, n * 4);
  hipStream_t s;
  hipStreamCreate(&s);
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int r = 0; r < 20; ++r) {
    hipLaunchKernelGGL(k_touch, dim3(256), dim3(512), 0, s, A, n);
    hipLaunchKernelGGL(k_probe, dim3(256), dim3(512), 0, s, (const float *)A, 4096, out, st);
  }
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  std::vector<double> med;
  for (int rep = 0; rep < 20; ++rep) {
    hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    std::vector<long long> h(512);
    hipMemcpy(h.data(), st, 512 * 8, hipMemcpyDeviceToHost);
    std::vector<double> dd;
    long long tmin = h[0];
    for (int b = 0; b < 256; ++b) tmin = std::min(tmin, h[2 * b]);
    for (int b = 0; b < 256; ++b) dd.push_back((double)(h[2 * b + 1] - h[2 * b]) * 10.0);
    std::sort(dd.begin(), dd.end());
    med.push_back(dd[128]);
  }
  std::sort(med.begin(), med.end());
  printf("entry -> first data: median over workgroups %.0f ns (median of 20 replays; min %.0f, max %.0f)\n", med[10], med[0], med[19]);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0, s);
  for (int r = 0; r < 50; ++r) hipGraphLaunch(ge, s);
  hipEventRecord(e1, s);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("touch + probe pair: %.2f us\n", ms * 1e3 / (50 * 20));
  return 0;
}
