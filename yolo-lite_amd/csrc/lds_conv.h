// Shared by the kernels that run 3x3 convs on an activation map held in LDS (resident.hip, conv3_band.hip): the k-loop with the
// B fragments from a GUARDED LDS map (the HBM activation layout cut down to a tile: a 3x3 tap is a constant byte offset that
// lands on data or on a zero) and the A fragments straight from L2 into registers, a few k-steps ahead.
#pragma once
#include "conv_common.h"

namespace yl {

// first tile / tile count of part `i` of `parts` when `ntiles` are dealt as evenly as possible (the first ntiles % parts get one more)
__device__ __forceinline__ void even_split(int ntiles, int parts, int i, int& t0, int& cnt) {
  const int base = ntiles / parts, rem = ntiles - base * parts;
  t0 = i * base + (i < rem ? i : rem);
  cnt = base + (i < rem ? 1 : 0);
}

constexpr int RC_D = 3;  // k-steps of filter fragments in flight per wave (resident.hip; conv3_band.hip passes its own depth)

// the first D k-steps of a 3x3 filter (this wave's two tiles of the four-tile block): issued BEFORE the barrier in front of the step
template <int D>
__device__ __forceinline__ void preload3(bf16x8 (&wa)[D][2], const bf16x8* __restrict__ wp) {
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int m = 0; m < 2; ++m) wa[d][m] = wp[(d * 4 + m) * 64];
}

// One 3x3 conv with 64 output channels from a guarded LDS map of 32 * KKN input channels: this wave's M-half (tiles 2 mh,
// 2 mh + 1 of the 4-tile filter block; wp already points at them) x NT pixel tiles.  src: LDS map + this lane's k-group plane
// (g * apl); tb[nt]: byte offset of the lane's pixel in a plane.  K order = conv_kernel's (tap-major, 32 channels per k-step).
template <int NT, int KKN = 2, int D = RC_D>
__device__ __forceinline__ void conv3_from_lds(f32x4 (&acc)[2][NT], bf16x8 (&wa)[D][2], const bf16x8* __restrict__ wp,
                                               const unsigned char* src, const unsigned (&tb)[NT], int W1, int apl) {
  constexpr int NKS = 9 * KKN;
  bf16x8 xb[2][NT];
  auto load_b = [&](int ks, bf16x8 (&dst)[NT]) __attribute__((always_inline)) {
    const int tap = ks / KKN, kk = ks - tap * KKN;
    const int kh = tap / 3, kw = tap - 3 * kh;
    const int off = ((kh - 1) * W1 + (kw - 1)) * 16 + kk * 4 * apl;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) dst[nt] = *reinterpret_cast<const bf16x8*>(src + (int)tb[nt] + off);
  };
  load_b(0, xb[0]);
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    if (ks + 1 < NKS) load_b(ks + 1, xb[(ks + 1) & 1]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int m = 0; m < 2; ++m)
        acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks % D][m], xb[ks & 1][nt], acc[m][nt], 0, 0, 0);
    if (ks + D < NKS) {
#pragma unroll
      for (int m = 0; m < 2; ++m) wa[ks % D][m] = wp[((ks + D) * 4 + m) * 64];
    }
  }
}

}  // namespace yl
