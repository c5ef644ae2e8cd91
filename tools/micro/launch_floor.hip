// Development calibration: per-kernel floor of dependent kernels replayed from a hipGraph, by grid size,
// dynamic LDS size and a minimal amount of per-wave work.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int WORK>
__global__ __launch_bounds__(256) void k(float* p, int n) {
  extern __shared__ float sm[];
  float v = p[threadIdx.x & 63];
  if (WORK) {
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int i = 0; i < WORK; ++i) v = v * 1.0001f + sm[(threadIdx.x + i) & 255];
  }
  if (v == 12345.f) p[threadIdx.x] = v;
}

template <int WORK>
static void run(float* d, int blocks, size_t lds, const char* name) {
  hipStream_t s;
  (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<WORK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int N = 90;
  hipGraph_t g;
  hipGraphExec_t ge;
  (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k<WORK>, dim3(blocks), dim3(256), lds, s, d, 0);
  (void)hipStreamEndCapture(s, &g);
  (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(ge, s);
  (void)hipStreamSynchronize(s);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < 20; ++i) (void)hipGraphLaunch(ge, s);
  (void)hipEventRecord(e1, s);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s blocks %5d lds %6zu: %6.2f us/kernel\n", name, blocks, lds, ms * 1e3 / (20 * N));
}

int main() {
  float* d;
  (void)hipMalloc(&d, 1 << 20);
  (void)hipMemset(d, 0, 1 << 20);
  run<0>(d, 1, 0, "empty");
  run<0>(d, 200, 0, "empty");
  run<0>(d, 200, 72 * 1024, "empty");
  run<0>(d, 800, 72 * 1024, "empty");
  run<0>(d, 2048, 72 * 1024, "empty");
  run<0>(d, 200, 16 * 1024, "empty");
  run<64>(d, 200, 72 * 1024, "64 dependent fma+lds per thread");
  run<512>(d, 200, 72 * 1024, "512 dependent fma+lds per thread");
  run<512>(d, 800, 72 * 1024, "512 dependent fma+lds per thread");
  return 0;
}
