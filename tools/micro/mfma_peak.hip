// Development calibration: achievable v_mfma_f32_16x16x32_bf16 rate on gfx950 with (a) registers only,
// (b) + A fragments from LDS, (c) + B fragments from L1-resident global memory.
// hipcc -O3 --offload-arch=gfx950 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int MODE, int MT, int NT>
__global__ __launch_bounds__(256) void k(const bf16x8* __restrict__ g, float* out, int iters) {
  __shared__ bf16x8 lds[4096];  // 64 KB
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 256) lds[i] = g[i];
  __syncthreads();
  f32x4 acc[MT][NT];
  for (int m = 0; m < MT; ++m)
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
  bf16x8 a[MT], b[NT];
  for (int m = 0; m < MT; ++m) a[m] = g[lane + 64 * m];
  for (int n = 0; n < NT; ++n) b[n] = g[lane + 64 * (n + 8)];
  const bf16x8* gp = g + lane + (blockIdx.x & 7) * 64;
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 1) {
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = lds[((it * MT + m) * 64 + lane) & 4095];
    }
    if (MODE >= 2) {
#pragma unroll
      for (int n = 0; n < NT; ++n) b[n] = gp[((it * NT + n) & 63) * 64];
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
  }
  float s = 0;
  for (int m = 0; m < MT; ++m)
    for (int n = 0; n < NT; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
  if (s == 12345.f) out[tid] = s;
}

template <int MODE, int MT, int NT>
void run(const bf16x8* g, float* out, int blocks, const char* name) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, MT, NT>), dim3(blocks), dim3(256), 0, 0, g, out, iters);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MODE, MT, NT>), dim3(blocks), dim3(256), 0, 0, g, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double fl = 2.0 * 16 * 16 * 32 * MT * NT * (double)iters * 4 * blocks;
  printf("%-28s MT%d NT%d blocks %5d: %8.3f ms  %7.1f TF/s\n", name, MT, NT, blocks, ms, fl / ms / 1e9);
}

int main() {
  bf16x8* g;
  float* out;
  hipMalloc(&g, 1 << 20);
  hipMemset(g, 0, 1 << 20);
  hipMalloc(&out, 4096);
  for (int blocks : {256, 512, 1024}) {
    run<0, 4, 4>(g, out, blocks, "regs only");
    run<1, 4, 4>(g, out, blocks, "A from LDS");
    run<2, 4, 4>(g, out, blocks, "A from LDS, B from L1");
    run<1, 4, 2>(g, out, blocks, "A from LDS");
    run<2, 4, 2>(g, out, blocks, "A from LDS, B from L1");
    run<1, 2, 4>(g, out, blocks, "A from LDS");
    run<2, 2, 4>(g, out, blocks, "A from LDS, B from L1");
  }
  return 0;
}
