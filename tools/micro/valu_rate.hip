// Development calibration: VALU issue cost of the activation epilogue on gfx950.  Per wave and instruction, cycles =
// 1024 SIMDs x clock x time / (waves x instructions).  Cases: v_fma_f32, v_pk_fma_f32, v_exp_f32, v_rcp_f32, the SiLU the
// kernels use (mul, exp2, add, rcp, mul), and a SiLU with the division by Newton steps on a bit-trick seed (one
// transcendental).  hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ float silu_ref(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * v));
}
__device__ __forceinline__ float silu_nr(float v) {  // rcp(d) by two Newton steps from an integer seed
  const float d = 1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * v);
  float r = __uint_as_float(0x7EF311C7u - __float_as_uint(d));
  r = r * (2.0f - d * r);
  r = r * (2.0f - d * r);
  return v * r;
}

__device__ __forceinline__ float silu_h(float v) {  // the sigmoid's exp2 and rcp in f16 (11 bits), the products in fp32
  const _Float16 e = __builtin_exp2f16((_Float16)(-1.44269504088896340736f * v));
  return v * (float)__builtin_amdgcn_rcph((_Float16)1.0f + e);
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  constexpr int N = 16;  // independent chains per lane
  float x[N];
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = seed + 0.001f * (threadIdx.x + i);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (MODE == 0) x[i] = __builtin_fmaf(x[i], 0.999f, 0.001f);
      if (MODE == 2) x[i] = __builtin_amdgcn_exp2f(x[i]);
      if (MODE == 3) x[i] = __builtin_amdgcn_rcpf(x[i]);
      if (MODE == 4) x[i] = silu_ref(x[i]) + 1.0f;
      if (MODE == 5) x[i] = silu_nr(x[i]) + 1.0f;
      if (MODE == 6) x[i] = (float)__builtin_exp2f16((_Float16)x[i]);
      if (MODE == 7) x[i] = (float)__builtin_amdgcn_rcph((_Float16)x[i]);
      if (MODE == 8) x[i] = silu_h(x[i]) + 1.0f;
    }
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < N; i += 2) {
        f32x2 v = {x[i], x[i + 1]};
        v = __builtin_elementwise_fma(v, f32x2{0.999f, 0.999f}, f32x2{0.001f, 0.001f});
        x[i] = v[0];
        x[i + 1] = v[1];
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) s += x[i];
  if (s == 12345.f) out[threadIdx.x] = s;
}

template <int MODE>
void run(float* out, const char* name, double ops_per_elem) {
  const int iters = 4000, blocks = 256 * 8;  // 8 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f);
  hipEventRecord(e0, 0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 3;
  const double elems = 16.0 * iters * 256.0 * blocks;          // lane-level results
  const double wave_results = elems / 64.0;
  const double simd_ns = ms * 1e6 * 1024.0;                    // SIMD-nanoseconds available
  printf("%-34s %8.3f ms  %8.1f G results/s   %6.2f ns of one SIMD per wave-result (= %5.1f clk at 2.4 GHz)\n", name, ms,
         elems / ms / 1e6, simd_ns / wave_results, simd_ns / wave_results * 2.4);
}

int main() {
  float* out;
  hipMalloc(&out, 4096);
  run<0>(out, "v_fma_f32", 1);
  run<1>(out, "v_pk_fma_f32 (per value)", 1);
  run<2>(out, "v_exp_f32", 1);
  run<3>(out, "v_rcp_f32", 1);
  run<4>(out, "silu (mul exp add rcp mul) + add", 6);
  run<5>(out, "silu, rcp by 2 Newton steps + add", 11);
  run<6>(out, "cvt + v_exp_f16 + cvt", 3);
  run<7>(out, "cvt + v_rcp_f16 + cvt", 3);
  run<8>(out, "silu with f16 exp2 / rcp + add", 8);
  return 0;
}
