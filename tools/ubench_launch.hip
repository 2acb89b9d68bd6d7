// Micro-benchmark (developer tool): cost of a dependent kernel boundary inside a hipGraph on this box, by kernel shape:
// grid size, bytes left dirty, argument block size, a dependent load chain at entry.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_launch.hip -o tools/bin/ubench_launch.exe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Big { int pad[200]; };

__global__ void k_empty(float *p) { if (p == nullptr) p[0] = 1.f; }
__global__ void k_write(float *p, int n4) {   // every thread stores n4 float4
  float4 *q = (float4 *)p;
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x, T = gridDim.x * (size_t)blockDim.x;
  for (int i = 0; i < n4; ++i) q[t + i * T] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void k_read(const float *p, float *o, int n4) {
  const float4 *q = (const float4 *)p;
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x, T = gridDim.x * (size_t)blockDim.x;
  float s = 0.f;
  for (int i = 0; i < n4; ++i) { float4 v = q[t + i * T]; s += v.x + v.y + v.z + v.w; }
  if (s == 12345.f) o[t] = s;
}
__global__ void k_bigarg(float *p, Big b) { if (p == nullptr) p[0] = (float)b.pad[threadIdx.x % 200]; }
__global__ void k_chain(const int *tab, float *p) {   // dependent loads at entry: scalar table -> vector load -> store
  const int i = tab[blockIdx.x];
  const float v = p[i + threadIdx.x];
  if (v == 12345.f) p[0] = v;
}

template <typename F>
double time_graph(F launch, int reps, hipStream_t st) {
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int r = 0; r < reps; ++r) launch();
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int t = 0; t < 5; ++t) {
    hipEventRecord(e0, st);
    hipGraphLaunch(ge, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  hipGraphExecDestroy(ge);
  hipGraphDestroy(g);
  return best * 1e3 / reps;
}

int main() {
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  float *buf, *out;
  int *tab;
  hipMalloc(&buf, 64 << 20);
  hipMalloc(&out, 64 << 20);
  hipMalloc(&tab, 4096 * 4);
  hipMemset(buf, 0, 64 << 20);
  hipMemset(tab, 0, 4096 * 4);
  const int R = 200;
  for (int grid : {1, 64, 256, 512, 1024}) {
    for (int thr : {256, 512}) {
      double t = time_graph([&] { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(thr), 0, st, buf); }, R, st);
      printf("empty           grid %4d x %3d : %6.2f us / kernel\n", grid, thr, t);
    }
  }
  Big b{};
  printf("big arg (800 B) grid  256 x 256 : %6.2f us / kernel\n", time_graph([&] { hipLaunchKernelGGL(k_bigarg, dim3(256), dim3(256), 0, st, buf, b); }, R, st));
  printf("entry load chain grid 256 x 256 : %6.2f us / kernel\n", time_graph([&] { hipLaunchKernelGGL(k_chain, dim3(256), dim3(256), 0, st, tab, buf); }, R, st));
  for (int mb : {1, 2, 4, 8, 16}) {
    const int n4 = mb * (1 << 20) / 16 / (256 * 256);
    printf("write %2d MB      grid  256 x 256 : %6.2f us / kernel\n", mb, time_graph([&] { hipLaunchKernelGGL(k_write, dim3(256), dim3(256), 0, st, buf, n4); }, R, st));
  }
  for (int mb : {1, 4, 16}) {
    const int n4 = mb * (1 << 20) / 16 / (256 * 256);
    printf("read  %2d MB      grid  256 x 256 : %6.2f us / kernel\n", mb, time_graph([&] { hipLaunchKernelGGL(k_read, dim3(256), dim3(256), 0, st, buf, out, n4); }, R, st));
  }
  // producer -> consumer pair: 4 MB written then read by the next kernel
  {
    const int n4 = 4 * (1 << 20) / 16 / (256 * 256);
    printf("write 4 MB -> read 4 MB pair     : %6.2f us / pair\n", time_graph([&] {
             hipLaunchKernelGGL(k_write, dim3(256), dim3(256), 0, st, buf, n4);
             hipLaunchKernelGGL(k_read, dim3(256), dim3(256), 0, st, buf, out, n4);
           }, R / 2, st));
  }
  return 0;
}
