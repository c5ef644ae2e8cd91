// Development calibration: aggregate kernel dispatch rate of S streams each replaying a hipGraph of K dependent launches of
// a trivial kernel (G workgroups of 64 threads) -- the command processor's side of "20 batches in flight, 60 launches each".
// GPU_MAX_HW_QUEUES=20 ./dispatch_rate [S] [K] [G]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void tiny(float* p) {
  if (p && threadIdx.x == 9999) p[0] = 1.f;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 20, K = argc > 2 ? atoi(argv[2]) : 60, G = argc > 3 ? atoi(argv[3]) : 1;
  std::vector<hipStream_t> st(S);
  std::vector<hipGraphExec_t> ge(S);
  for (int s = 0; s < S; ++s) {
    CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
    hipGraph_t g;
    CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < K; ++k) hipLaunchKernelGGL(tiny, dim3(G), dim3(64), 0, st[s], (float*)nullptr);
    CK(hipStreamEndCapture(st[s], &g));
    CK(hipGraphInstantiate(&ge[s], g, nullptr, nullptr, 0));
  }
  for (int rep = 0; rep < 3; ++rep) {
    const int rounds = 50;
    for (int s = 0; s < S; ++s) CK(hipGraphLaunch(ge[s], st[s]));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st[0]));
    for (int r = 0; r < rounds; ++r)
      for (int s = 0; s < S; ++s) CK(hipGraphLaunch(ge[s], st[s]));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e1, st[0]));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double n = (double)rounds * S * K;
    printf("S=%d K=%d G=%d: %.1f k launches/s, %.2f us per launch chip-wide, %.3f ms per graph\n", S, K, G, n / ms, ms * 1e3 / n,
           ms / rounds / S * S);
  }
  return 0;
}
