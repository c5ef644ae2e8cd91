// Development calibration: L1/TA throughput of wave-wide loads by access pattern (L1-resident data).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) int i32x4;

// PAT 0: b128 contiguous (lane*16)                       1 KB contiguous
// PAT 1: b128, (lane&15)*128 + (lane>>4)*16             conv B fragment, Cin=64
// PAT 2: b128, (lane&15)*256 + (lane>>4)*16             conv B fragment, Cin=128
// PAT 3: b128, (lane&15)*32 + (lane>>4)*... Cin=16: (lane&15)*32 + (lane>>4&1)*16 (half the lanes dup)
// PAT 4: b32 contiguous   PAT 5: b32 stride 2   PAT 6: b64 contiguous
// PAT 7: b128, (lane>>2)*128 + (lane&3)*16              4 lanes share a 64 B chunk (quad-contiguous)
template <int PAT>
__global__ __launch_bounds__(256) void k(const char* __restrict__ g, int* out, int iters) {
  const int lane = threadIdx.x & 63;
  int off;
  if (PAT == 0) off = lane * 16;
  if (PAT == 1) off = (lane & 15) * 128 + (lane >> 4) * 16;
  if (PAT == 2) off = (lane & 15) * 256 + (lane >> 4) * 16;
  if (PAT == 3) off = (lane & 15) * 32 + ((lane >> 4) & 1) * 16;
  if (PAT == 4) off = lane * 4;
  if (PAT == 5) off = lane * 8;
  if (PAT == 6) off = lane * 8;
  if (PAT == 7) off = (lane >> 2) * 128 + (lane & 3) * 16;
  if (PAT == 8) off = (lane & 15) * 16 + (lane >> 4) * 1024;         // planar: 256 B runs in 4 planes
  if (PAT == 9) off = (lane & 15) * 16 + (lane >> 4) * 1024 + 16;    // ... shifted by one pixel
  if (PAT == 10) off = (lane >> 2) * 128 + (lane & 3) * 16 + 16;      // quad straddles 64 B inside a 128 B line
  if (PAT == 11) off = (lane >> 2) * 128 + (lane & 3) * 16 + 80;      // quad straddles two 128 B lines
  if (PAT == 12) off = (lane & 15) * 16 + (lane >> 4) * 1024 + 48;    // planar shifted by 3 pixels
  const char* p = g + off + (threadIdx.x >> 6) * 4096;
  i32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      int o = 0;
      asm volatile("" : "+v"(o));
      const char* q = p + o;
      if (PAT <= 3 || PAT >= 7) {
        i32x4 v = *reinterpret_cast<const i32x4*>(q);
        acc += v;
      } else if (PAT == 6) {
        const int2 v = *reinterpret_cast<const int2*>(q);
        acc[0] += v.x;
        acc[1] += v.y;
      } else {
        acc[0] += *reinterpret_cast<const int*>(q);
      }
      asm volatile("" ::: "memory");
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345) out[threadIdx.x] = 1;
}

template <int PAT>
void run(const char* g, int* out, const char* name) {
  const int iters = 4000, blocks = 256 * 2;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<PAT>), dim3(blocks), dim3(256), 0, 0, g, out, iters);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<PAT>), dim3(blocks), dim3(256), 0, 0, g, out, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  // wave-loads per CU = 2 blocks * 4 waves * iters * 8
  const double wl = 2.0 * 4 * iters * 8;
  printf("%-44s %8.3f ms  %6.1f cycles/wave-load/CU @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / wl);
}

int main() {
  char* g;
  int* out;
  (void)hipMalloc(&g, 1 << 20);
  (void)hipMemset(g, 0, 1 << 20);
  (void)hipMalloc(&out, 4096);
  run<0>(g, out, "b128 contiguous");
  run<1>(g, out, "b128 conv-B pattern pixel stride 128 B");
  run<2>(g, out, "b128 conv-B pattern pixel stride 256 B");
  run<3>(g, out, "b128 conv-B pattern pixel stride 32 B");
  run<7>(g, out, "b128 quad-contiguous 64 B, stride 128 B");
  run<8>(g, out, "b128 planar 4 x 256 B aligned");
  run<9>(g, out, "b128 planar shifted +16 B");
  run<12>(g, out, "b128 planar shifted +48 B");
  run<10>(g, out, "b128 quad straddles 64 B in a line");
  run<11>(g, out, "b128 quad straddles two lines");
  run<4>(g, out, "b32 contiguous");
  run<5>(g, out, "b32 stride 2");
  run<6>(g, out, "b64 contiguous");
  return 0;
}
