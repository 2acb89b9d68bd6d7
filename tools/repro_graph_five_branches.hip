// Minimal stand-alone attempt at the crash that caps libmivi's forked graphs at FOUR branches (mivi_internal.h kMaxKids):
// inside the library, `hipGraphLaunch` of a stream-captured graph with FIVE forked branches segfaulted in hip::Graph::UpdateStreams
// after the graph had been re-captured once (7 of 8 runs on ROCm 7.2.0 / MI355X; three and four branches: 12 of 12 clean).
//
// What this program does, per round: capture `branches` forked branches (event fork from the origin stream, `depth` small kernels per
// branch on a stream of its own, event join), instantiate, launch `launches` times on a THIRD stream (the library replays on the
// caller's stream, not on the capture stream), destroy, and capture again with the same streams -- the re-capture is what the library
// does when a batch length changes.  Branch streams are created with a full CU mask like the library's lane streams
// (hipExtStreamCreateWithCUMask: a hardware queue of their own).
//
//   hipcc --offload-arch=gfx950 -O2 tools/repro_graph_five_branches.hip -o /tmp/repro5 && /tmp/repro5 5 8 6 20
//   argv: branches (default 5), rounds of capture/launch/destroy (8), kernels per branch (6), launches per round (20)
// Exit code 0 and "OK" = no crash in this configuration; a segfault is the bug.  The outcome measured on the GPU box is recorded in
// DESIGN.md 9 next to the cap.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHK(x)                                                                                  \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) {                                                                     \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));       \
      exit(2);                                                                                  \
    }                                                                                           \
  } while (0)

__global__ void k_touch(float *p, int n, float a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * a + 1.0f;
}

int main(int argc, char **argv) {
  const int branches = argc > 1 ? atoi(argv[1]) : 5;
  const int rounds = argc > 2 ? atoi(argv[2]) : 8;
  const int depth = argc > 3 ? atoi(argv[3]) : 6;
  const int launches = argc > 4 ? atoi(argv[4]) : 20;
  const int n = 1 << 16;
  hipStream_t cap, user;
  CHK(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
  CHK(hipStreamCreateWithFlags(&user, hipStreamNonBlocking));
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  std::vector<uint32_t> mask((prop.multiProcessorCount + 31) / 32, 0xffffffffu);
  std::vector<hipStream_t> bs(branches);
  std::vector<hipEvent_t> join(branches);
  std::vector<float *> buf(branches);
  for (int b = 0; b < branches; ++b) {
    CHK(hipExtStreamCreateWithCUMask(&bs[b], (uint32_t)mask.size(), mask.data()));
    CHK(hipEventCreateWithFlags(&join[b], hipEventDisableTiming));
    CHK(hipMalloc(&buf[b], n * sizeof(float)));
    CHK(hipMemset(buf[b], 0, n * sizeof(float)));
  }
  hipEvent_t fork;
  CHK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  for (int r = 0; r < rounds; ++r) {
    const int d = depth + (r & 1);   // a different graph shape on every re-capture
    CHK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
    CHK(hipEventRecord(fork, cap));
    for (int b = 0; b < branches; ++b) {
      hipStream_t s = b == 0 ? cap : bs[b];
      if (b) CHK(hipStreamWaitEvent(s, fork, 0));
      for (int k = 0; k < d; ++k) k_touch<<<n / 256, 256, 0, s>>>(buf[b], n, 0.5f);
      if (b) {
        CHK(hipEventRecord(join[b], s));
        CHK(hipStreamWaitEvent(cap, join[b], 0));
      }
    }
    hipGraph_t g;
    CHK(hipStreamEndCapture(cap, &g));
    hipGraphExec_t e;
    CHK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
    CHK(hipGraphDestroy(g));
    for (int l = 0; l < launches; ++l) CHK(hipGraphLaunch(e, user));
    CHK(hipStreamSynchronize(user));
    CHK(hipGraphExecDestroy(e));
    printf("round %d: %d branches x %d kernels, %d launches: ok\n", r, branches, d, launches);
    fflush(stdout);
  }
  printf("OK\n");
  return 0;
}
