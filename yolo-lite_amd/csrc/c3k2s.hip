// The body of a narrow C3k2 block (yolo11.yaml layer 2 of YOLO11n: c = 16) in ONE launch:
//     m = Bottleneck(c, c, e = 0.5):  c_out = b + SiLU(cv2(SiLU(cv1(b))))      two 3x3 convs, c -> c/2 -> c
//     y = SiLU(C3k2.cv2(cat(a, b, c_out)))                                       1x1 conv, 3c -> c2
// reference nn/modules/block.py:220-235 (C2f.forward after cv1), :330-343 (Bottleneck), conv.py:35-53.
// [a | b] = the two halves of C3k2.cv1's output (planes 0..c/4-1 of the block's concat buffer).
//
// Why: at 160x160 these three convs move 0.39 GB per batch of 32 (b is read three times, the c/2-channel
// intermediate and c_out are written and re-read, cat(a, b, c_out) is re-read) for 52 MB of input and 105 MB of
// output; their matrix work is trivial.  Here b, the intermediate and c_out live in LDS tiles.
//
// One workgroup (256 threads) = TR x TC = 4 x 32 output pixels of one image:
//   0. b with a halo of 2 (8 x 36 pixels, 2 planes) and a (4 x 32, 2 planes) -> LDS.  Positions outside the map read
//      the layout's zero guards or a neighbouring row / image; those values only ever feed intermediate positions
//      that phase A forces to zero, so no bounds logic is needed on the loads.
//   A. mid = SiLU(cv1(b)) on the 6 x 34 pixels phase B needs, ZERO outside the map (it is cv2's zero padding).
//   B. c_out = b + SiLU(cv2(mid)) on the tile.
//   C. y = SiLU(cv2'(a, b, c_out)): K = 48 channels from three LDS tiles, 64 output channels, 16-byte stores.
// Filters arrive as conv_pack_weights fragments (the whole block is 88 KB of bf16 registers-worth: 15 fragments per
// lane) and the k order, accumulation and rounding points are those of conv_kernel: bit-identical to three launches.
#include "conv_common.h"

namespace yl {

constexpr int C3_TR = 4, C3_TC = 32;
constexpr int C3_BR = C3_TR + 4, C3_BC = C3_TC + 4;   // b tile with halo 2
constexpr int C3_MR = C3_TR + 2, C3_MC = C3_TC + 2;   // intermediate with halo 1
constexpr int C3_KA = 5, C3_KB = 3, C3_KC = 2;        // k-steps: 9*16/32, 9*8/32, 48/32 (rounded up)

size_t c3k2s_lds_bytes() {
  return (size_t)(2 * C3_BR * C3_BC + 2 * C3_TR * C3_TC + C3_MR * C3_MC + 2 * C3_TR * C3_TC) * 16;
}

__global__ __launch_bounds__(256) void c3k2s_kernel(C3k2sArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char c3mem[];
  unsigned char* const tb = c3mem;                                  // [2][C3_BR][C3_BC] x 16 B
  unsigned char* const ta = tb + 2 * C3_BR * C3_BC * 16;            // [2][C3_TR][C3_TC]
  unsigned char* const tm = ta + 2 * C3_TR * C3_TC * 16;            // [C3_MR][C3_MC]
  unsigned char* const tc = tm + C3_MR * C3_MC * 16;                // [2][C3_TR][C3_TC]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int b = bid / a.tiles_y;
  const int y0 = ty * C3_TR, x0 = tx * C3_TC;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.tin.base, (short)0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)a.yout.base, (short)0, a.out_bytes, 0x00020000);
  const unsigned in_plane = (unsigned)(a.tin.plane * 2);

  // ---- 0. a (planes 0, 1) and b with halo (planes 2, 3) -> LDS; 832 sixteen-byte pixels, 4 per thread in flight
  {
    constexpr int NB = 2 * C3_BR * C3_BC, NA = 2 * C3_TR * C3_TC;
    bf16x8 v[4];
    unsigned dst[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * 256;
      int pl, yy, xx, d;
      if (i < NB) {
        pl = i / (C3_BR * C3_BC);
        const int r = i - pl * (C3_BR * C3_BC);
        const int ry = r / C3_BC, rx = r - ry * C3_BC;
        yy = y0 - 2 + ry;
        xx = x0 - 2 + rx;
        d = i * 16;
        pl += 2;
      } else {
        const int j = i - NB;
        pl = j / (C3_TR * C3_TC);
        const int r = j - pl * (C3_TR * C3_TC);
        const int ry = r / C3_TC, rx = r - ry * C3_TC;
        yy = y0 + ry;
        xx = x0 + rx;
        d = (int)(ta - tb) + j * 16;
      }
      // guarded index; may be negative for image 0 (-> huge unsigned offset -> out of range -> zeros)
      const long long pix = (long long)a.tin.g0 + (long long)b * a.tin.img + (long long)yy * a.tin.W1 + xx;
      const bool ok = i < NB + NA && pix >= 0 && pix < a.plane_px;
      v[u] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                            rin, ok ? (unsigned)pl * in_plane + (unsigned)pix * 16u : OOB_LANE, 0, 0));
      dst[u] = i < NB + NA ? (unsigned)d : 0xffffffffu;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (dst[u] != 0xffffffffu) *reinterpret_cast<bf16x8*>(tb + dst[u]) = v[u];
  }
  // filter fragments and biases (conv_pack_weights order, MT = 1 / 1 / 4)
  bf16x8 wA[C3_KA], wB[C3_KB], wC[C3_KC][4];
#pragma unroll
  for (int ks = 0; ks < C3_KA; ++ks) wA[ks] = reinterpret_cast<const bf16x8*>(a.wA)[ks * 64 + lane];
#pragma unroll
  for (int ks = 0; ks < C3_KB; ++ks) wB[ks] = reinterpret_cast<const bf16x8*>(a.wB)[ks * 64 + lane];
#pragma unroll
  for (int ks = 0; ks < C3_KC; ++ks)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) wC[ks][mt] = reinterpret_cast<const bf16x8*>(a.wC)[(ks * 4 + mt) * 64 + lane];
  const f32x4 biasA = *reinterpret_cast<const f32x4*>(a.bA + g * 4);  // chan_in_block(1, 0, g, r) = 4g + r
  const f32x4 biasB = *reinterpret_cast<const f32x4*>(a.bB + g * 4);
  f32x4 biasC[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) biasC[mt] = *reinterpret_cast<const f32x4*>(a.bC + chan_in_block(4, mt, g, 0));
  __syncthreads();

  // ---- A. mid = SiLU(cv1(b)), 16 -> 8 channels, on C3_MR x C3_MC; local (r, c) <-> map (y0 - 1 + r, x0 - 1 + c)
  {
    int boff[C3_KA];  // this lane's B fragment of k-step ks relative to the pixel's corner in tb
#pragma unroll
    for (int ks = 0; ks < C3_KA; ++ks) {
      const int k8 = ks * 4 + g;
      const int tap = min(k8 >> 1, 8), c8 = k8 & 1;  // k groups past tap 8 have zero filters: any finite data
      const int kh = tap / 3, kw = tap - kh * 3;
      boff[ks] = ((c8 * C3_BR + kh) * C3_BC + kw + n) * 16;
    }
    constexpr int TPR = (C3_MC + 15) / 16;  // 3 tiles per row, the last 2 pixels wide
    for (int t = wave; t < C3_MR * TPR; t += 4) {
      const int r = t / TPR, ct = t - r * TPR;  // scalar
      const int c = ct * 16 + n;
      const int cl = min(c, C3_MC - 1);
      const unsigned char* corner = tb + (r * C3_BC + ct * 16) * 16 - (c - cl) * 16;
      f32x4 acc = biasA;
#pragma unroll
      for (int ks = 0; ks < C3_KA; ++ks)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wA[ks], *reinterpret_cast<const bf16x8*>(corner + boff[ks]), acc, 0, 0, 0);
      const int my = y0 - 1 + r, mx = x0 - 1 + cl;
      const bool inside = (unsigned)my < (unsigned)a.H && (unsigned)mx < (unsigned)a.W;
      bf16x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = inside ? (__bf16)silu_f(acc[q]) : (__bf16)0.f;
      if (g < 2 && c < C3_MC) *reinterpret_cast<bf16x4*>(tm + (r * C3_MC + c) * 16 + g * 8) = o;
    }
  }
  __syncthreads();

  // ---- B. c_out = b + SiLU(cv2(mid)), 8 -> 16 channels, on the tile
  {
    int boff[C3_KB];
#pragma unroll
    for (int ks = 0; ks < C3_KB; ++ks) {
      const int tap = min(ks * 4 + g, 8);  // one k group (8 channels) per tap
      const int kh = tap / 3, kw = tap - kh * 3;
      boff[ks] = (kh * C3_MC + kw + n) * 16;
    }
    constexpr int TPR = C3_TC / 16;
    for (int t = wave; t < C3_TR * TPR; t += 4) {
      const int ly = t / TPR, ct = t - ly * TPR;  // scalar
      const int lx = ct * 16 + n;
      const unsigned char* corner = tm + (ly * C3_MC + ct * 16) * 16;
      f32x4 acc = biasB;
#pragma unroll
      for (int ks = 0; ks < C3_KB; ++ks)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wB[ks], *reinterpret_cast<const bf16x8*>(corner + boff[ks]), acc, 0, 0, 0);
      // residual: channels 4g..4g+3 of b at the pixel (conv_kernel's epilogue order: SiLU, then += residual)
      const bf16x4 rb = *reinterpret_cast<const bf16x4*>(tb + (((g >> 1) * C3_BR + ly + 2) * C3_BC + lx + 2) * 16 + (g & 1) * 8);
      bf16x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = (__bf16)(silu_f(acc[q]) + (float)rb[q]);
      *reinterpret_cast<bf16x4*>(tc + (((g >> 1) * C3_TR + ly) * C3_TC + lx) * 16 + (g & 1) * 8) = o;
    }
  }
  __syncthreads();

  // ---- C. y = SiLU(cv2'(cat(a, b, c_out))): k groups 0, 1 = a; 2, 3 = b; 4, 5 = c_out; 6, 7 = zero filters
  {
    // per lane: source tile, plane and row pitch of its k group in k-steps 0 and 1
    const unsigned char* src0 = g < 2 ? ta + g * C3_TR * C3_TC * 16 : tb + ((g - 2) * C3_BR + 2) * C3_BC * 16 + 2 * 16;
    const int pitch0 = (g < 2 ? C3_TC : C3_BC) * 16;
    const unsigned char* src1 = tc + min(g, 1) * C3_TR * C3_TC * 16;  // groups 6, 7: any finite data
    const unsigned out_plane = (unsigned)(a.yout.plane * 2);
    constexpr int TPR = C3_TC / 16;
    for (int t = wave; t < C3_TR * TPR; t += 4) {
      const int ly = t / TPR, ct = t - ly * TPR;  // scalar
      const int lx = ct * 16 + n;
      const bf16x8 x0f = *reinterpret_cast<const bf16x8*>(src0 + ly * pitch0 + lx * 16);
      const bf16x8 x1f = *reinterpret_cast<const bf16x8*>(src1 + (ly * C3_TC + lx) * 16);
      f32x4 acc[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wC[0][mt], x0f, biasC[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wC[1][mt], x1f, acc[mt], 0, 0, 0);
      }
      const int oy = y0 + ly, ox = x0 + lx;
      const unsigned po = (oy < a.H && ox < a.W) ? (unsigned)(a.yout.g0 + b * a.yout.img + oy * a.yout.W1 + ox) * 16u : OOB_LANE;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = silu_f(acc[2 * p][r]);
          v[4 + r] = silu_f(acc[2 * p + 1][r]);
        }
        __builtin_amdgcn_raw_buffer_store_b128(
            u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])},
            rout, (unsigned)(p * 4 + g) * out_plane + po, 0, 0);
      }
    }
  }
}

int c3k2s_launch(const C3k2sArgs& a0, hipStream_t s) {
  C3k2sArgs a = a0;
  a.tiles_x = (a.W + C3_TC - 1) / C3_TC;
  a.tiles_y = (a.H + C3_TR - 1) / C3_TR;
  hipLaunchKernelGGL(c3k2s_kernel, dim3((unsigned)(a.B * a.tiles_y * a.tiles_x)), dim3(256), c3k2s_lds_bytes(), s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

}  // namespace yl
