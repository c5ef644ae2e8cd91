// A whole Bottleneck(c, c, k = (3, 3), e = 1.0) of a C3k block in ONE launch:  y = x + SiLU(cv2(SiLU(cv1(x))))
// (reference nn/modules/block.py:330-343, 731-739: C3k.m; conv.py:35-53), c = 32 (layer 6 of YOLO11n, 40x40) or
// 64 (layers 8 and 22, 20x20).
//
// Why: these are the 12 smallest 3x3 launches of the network (8-12 us each for 0.5-1 GFLOP): at 20x20 / 40x40 a
// launch is all latency -- start-up, filter staging, first loads, store tail.  Doing both convs of a Bottleneck in
// one launch pays that once, and the c-channel tensor between them lives in LDS.  x and y must be different
// buffers (the lowering ping-pongs the two Bottlenecks of a C3k between the block's buffer and a scratch buffer):
// a workgroup reads a halo of x that belongs to its neighbours' tiles.
//
// One workgroup (512 threads) = TR x TC = 4 x 32 outputs of one image:
//   0. x with a halo of 2 (8 x 36 pixels, CP = c/8 planes) and cv1's filter slab -> LDS.  Halo positions outside the
//      map read the layout's zero guards or a neighbouring row / image: they only feed intermediate positions that
//      phase A forces to zero.
//   A. mid = SiLU(cv1(x)) on 6 x 34 (zero outside the map: cv2's zero padding) -> LDS, two 16-pixel tiles per wave
//      and trip so that an A fragment read from the slab serves two MFMAs per channel tile.
//   -  cv2's slab replaces cv1's.
//   B. y = x + SiLU(cv2(mid)) on the tile, residual from the x tile, 16-byte stores.
// Packed filters, k order (tap-major, 8-channel groups), accumulation and rounding points are conv_kernel's:
// bit-identical to the two launches (tests/test_gpu_net.py).
#include "conv_common.h"

namespace yl {

constexpr int BN_TR = 4, BN_TC = 32;
constexpr int BN_XR = BN_TR + 4, BN_XC = BN_TC + 4;
constexpr int BN_MR = BN_TR + 2, BN_MC = BN_TC + 2;

template <int CP>
constexpr size_t bneck2_lds() {
  return (size_t)(9 * CP / 4) * (CP / 2) * 1024 + (size_t)CP * (BN_XR * BN_XC + BN_MR * BN_MC) * 16;
}

constexpr int BN_THREADS = 512;  // 8 waves, two per SIMD: one wave per SIMD left every LDS / MFMA dependency exposed
constexpr int BN_WAVES = BN_THREADS / 64;

template <int CP>
__global__ __launch_bounds__(BN_THREADS) __attribute__((amdgpu_waves_per_eu(2))) void bneck2_kernel(Bneck2Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bnmem[];
  constexpr int MT = CP / 2;            // channel tiles of 16
  constexpr int KS = 9 * CP / 4;        // k-steps of a 3x3 conv with c input channels (c % 32 == 0)
  bf16x8* const slab = reinterpret_cast<bf16x8*>(bnmem);               // [KS][MT][64]
  unsigned char* const tx = bnmem + (size_t)KS * MT * 1024;            // [CP][BN_XR][BN_XC] x 16 B
  unsigned char* const tm = tx + (size_t)CP * BN_XR * BN_XC * 16;      // [CP][BN_MR][BN_MC]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx_i = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty_i = bid % a.tiles_y;
  const int b = bid / a.tiles_y;
  const int y0 = ty_i * BN_TR, x0 = tx_i * BN_TC;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.xin.base, (short)0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)a.yout.base, (short)0, a.out_bytes, 0x00020000);
  const unsigned in_plane = (unsigned)(a.xin.plane * 2);

  auto stage_slab = [&](const __bf16* w) {
    const bf16x8* src = reinterpret_cast<const bf16x8*>(w);
    constexpr int N16 = KS * MT * 64;
    for (int i0 = tid; i0 < N16; i0 += BN_THREADS * 8) {
      bf16x8 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * BN_THREADS < N16) t[u] = src[i0 + u * BN_THREADS];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * BN_THREADS < N16) slab[i0 + u * BN_THREADS] = t[u];
    }
  };

  // ---- 0. x with halo -> LDS (all loads of a thread in flight at once), cv1's filters -> slab
  {
    constexpr int NX = CP * BN_XR * BN_XC;
    constexpr int NL = (NX + BN_THREADS - 1) / BN_THREADS;
    bf16x8 v[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int i = tid + u * BN_THREADS;
      const int pl = i / (BN_XR * BN_XC);
      const int r = i - pl * (BN_XR * BN_XC);
      const int ry = r / BN_XC, rx = r - ry * BN_XC;
      // guarded index; may be negative for image 0 (-> out of range -> zeros)
      const long long pix = (long long)a.xin.g0 + (long long)b * a.xin.img + (long long)(y0 - 2 + ry) * a.xin.W1 + (x0 - 2 + rx);
      const bool ok = i < NX && pix >= 0 && pix < a.plane_px;
      v[u] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                            rin, ok ? (unsigned)pl * in_plane + (unsigned)pix * 16u : OOB_LANE, 0, 0));
    }
    stage_slab(a.w1);
#pragma unroll
    for (int u = 0; u < NL; ++u)
      if (tid + u * BN_THREADS < NX) *reinterpret_cast<bf16x8*>(tx + (size_t)(tid + u * BN_THREADS) * 16) = v[u];
  }
  f32x4 bias1[MT], bias2[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    bias1[mt] = *reinterpret_cast<const f32x4*>(a.b1 + chan_in_block(MT, mt, g, 0));
    bias2[mt] = *reinterpret_cast<const f32x4*>(a.b2 + chan_in_block(MT, mt, g, 0));
  }
  __syncthreads();

  // one 3x3 conv over two 16-pixel tiles whose corners (pixel 0, tap (0, 0)) are c0 / c1 in a [CP][R][C] LDS tile
  auto conv2 = [&](const unsigned char* c0, const unsigned char* c1, int rows, int cols, const f32x4 (&bias)[MT],
                   f32x4 (&acc)[MT][2]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt][0] = acc[mt][1] = bias[mt];
#pragma unroll 3
    for (int ks = 0; ks < KS; ++ks) {
      const int k8 = ks * 4 + g;
      const int tap = k8 / CP, c8 = k8 % CP;  // c % 32 == 0: every k group is a real (tap, plane) pair
      const int kh = tap / 3, kw = tap - kh * 3;
      const int off = ((c8 * rows + kh) * cols + kw + n) * 16;
      const bf16x8 xb0 = *reinterpret_cast<const bf16x8*>(c0 + off);
      const bf16x8 xb1 = *reinterpret_cast<const bf16x8*>(c1 + off);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const bf16x8 wa = slab[(ks * MT + mt) * 64 + lane];
        acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb0, acc[mt][0], 0, 0, 0);
        acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb1, acc[mt][1], 0, 0, 0);
      }
    }
  };

  // ---- A. mid = SiLU(cv1(x)) on BN_MR x BN_MC; local (r, c) <-> map (y0 - 1 + r, x0 - 1 + c)
  {
    constexpr int TPR = (BN_MC + 15) / 16;  // 3 tiles per row, the last 2 pixels wide
    constexpr int NTILE = BN_MR * TPR;      // 18
    for (int t0 = 2 * wave; t0 < NTILE; t0 += 2 * BN_WAVES) {
      int rr[2], cc[2];
      const unsigned char* corner[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = min(t0 + u, NTILE - 1);  // scalar
        const int r = t / TPR, ct = t - r * TPR;
        const int c = ct * 16 + n, cl = min(c, BN_MC - 1);
        rr[u] = r;
        cc[u] = (t0 + u < NTILE && c < BN_MC) ? c : -1;  // -1: nothing to store
        corner[u] = tx + (r * BN_XC + ct * 16) * 16 - (c - cl) * 16;
      }
      f32x4 acc[MT][2];
      conv2(corner[0], corner[1], BN_XR, BN_XC, bias1, acc);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int c = cc[u];
        const int my = y0 - 1 + rr[u], mx = x0 - 1 + c;
        const bool inside = (unsigned)my < (unsigned)a.H && (unsigned)mx < (unsigned)a.W;
#pragma unroll
        for (int p = 0; p < MT / 2; ++p) {  // tile pair p = the 8 channels of plane p*4 + g
          bf16x8 o;
          const f32x4 s0 = silu4(acc[2 * p][u]), s1 = silu4(acc[2 * p + 1][u]);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            o[q] = inside ? (__bf16)s0[q] : (__bf16)0.f;
            o[4 + q] = inside ? (__bf16)s1[q] : (__bf16)0.f;
          }
          if (c >= 0) *reinterpret_cast<bf16x8*>(tm + (((p * 4 + g) * BN_MR + rr[u]) * BN_MC + c) * 16) = o;
        }
      }
    }
  }
  __syncthreads();  // mid complete; every wave is done with cv1's slab
  stage_slab(a.w2);
  __syncthreads();

  // ---- B. y = x + SiLU(cv2(mid)) on the tile
  {
    constexpr int TPR = BN_TC / 16;
    constexpr int NTILE = BN_TR * TPR;  // 8: one pair for each of four waves
    const unsigned out_plane = (unsigned)(a.yout.plane * 2);
    for (int t0 = 2 * wave; t0 < NTILE; t0 += 2 * BN_WAVES) {
      int ly[2], lx[2];
      const unsigned char* corner[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = t0 + u;  // scalar
        ly[u] = t / TPR;
        lx[u] = (t - ly[u] * TPR) * 16 + n;
        corner[u] = tm + (ly[u] * BN_MC + (t - ly[u] * TPR) * 16) * 16;
      }
      f32x4 acc[MT][2];
      conv2(corner[0], corner[1], BN_MR, BN_MC, bias2, acc);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int oy = y0 + ly[u], ox = x0 + lx[u];
        const unsigned po = (oy < a.H && ox < a.W) ? (unsigned)(a.yout.g0 + b * a.yout.img + oy * a.yout.W1 + ox) * 16u : OOB_LANE;
#pragma unroll
        for (int p = 0; p < MT / 2; ++p) {
          const bf16x8 rx = *reinterpret_cast<const bf16x8*>(tx + (((p * 4 + g) * BN_XR + ly[u] + 2) * BN_XC + lx[u] + 2) * 16);
          float v[8];
          const f32x4 s0 = silu4(acc[2 * p][u]), s1 = silu4(acc[2 * p + 1][u]);
#pragma unroll
          for (int q = 0; q < 4; ++q) {  // conv_kernel's epilogue order: SiLU, then += residual
            v[q] = s0[q] + (float)rx[q];
            v[4 + q] = s1[q] + (float)rx[4 + q];
          }
          store_b128_nt(u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])},
                        rout, (unsigned)(p * 4 + g) * out_plane + po, a.st_nt);
        }
      }
    }
  }
}

template <int CP>
static int bneck2_launch_t(const Bneck2Args& a, hipStream_t s) {
  auto kern = bneck2_kernel<CP>;
  constexpr size_t lds = bneck2_lds<CP>();
  if (lds > 64 * 1024)
    if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), (int)(160 * 1024))) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * a.tiles_y * a.tiles_x)), dim3(BN_THREADS), lds, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

int bneck2_launch(const Bneck2Args& a0, hipStream_t s) {
  Bneck2Args a = a0;
  a.st_nt = st_nt_policy(a.out_bytes);
  a.tiles_x = (a.W + BN_TC - 1) / BN_TC;
  a.tiles_y = (a.H + BN_TR - 1) / BN_TR;
  if (a.c == 32) return bneck2_launch_t<4>(a, s);
  if (a.c == 64) return bneck2_launch_t<8>(a, s);
  return fail(-3, "bneck2: supported widths are 32 and 64 channels");
}

}  // namespace yl
