// The body of a narrow C3k2 block (yolo11.yaml layers 2, 4 and 16 of YOLO11n: c = 16 or 32) in ONE launch:
//     m = Bottleneck(c, c, e = 0.5):  c_out = b + SiLU(cv2(SiLU(cv1(b))))      two 3x3 convs, c -> c/2 -> c
//     y = SiLU(C3k2.cv2(cat(a, b, c_out)))                                       1x1 conv, 3c -> c2
// reference nn/modules/block.py:220-235 (C2f.forward after cv1), :330-343 (Bottleneck), conv.py:35-53.
// [a | b] = the two halves of C3k2.cv1's output (planes 0..c/4-1 of the block's concat buffer).
//
// Why: at 160x160 these three convs move 0.39 GB per batch of 32 (b is read three times, the c/2-channel
// intermediate and c_out are written and re-read, cat(a, b, c_out) is re-read) for 52 MB of input and 105 MB of
// output; their matrix work is trivial.  Here b, the intermediate and c_out live in LDS tiles.
//
// One workgroup (256 threads) = TR x TC = 8 x 32 output pixels of one image (sizes for c = 16; 4 x 32 tiles took
// 126 us for the three blocks, 8 x 32 take 117 us: less halo per output):
//   0. b with a halo of 2 (12 x 36 pixels, 2 planes) and a (8 x 32, 2 planes) -> LDS.  Positions outside the map read
//      the layout's zero guards or a neighbouring row / image; those values only ever feed intermediate positions
//      that phase A forces to zero, so no bounds logic is needed on the loads.
//   A. mid = SiLU(cv1(b)) on the 10 x 34 pixels phase B needs, ZERO outside the map (it is cv2's zero padding).
//   B. c_out = b + SiLU(cv2(mid)) on the tile.
//   C. y = SiLU(cv2'(a, b, c_out)): K = 48 channels from three LDS tiles, 64 output channels, 16-byte stores.
// Filters arrive as conv_pack_weights fragments (the whole block is 88 KB of bf16 registers-worth: 15 fragments per
// lane) and the k order, accumulation and rounding points are those of conv_kernel: bit-identical to three launches.
#include "conv_common.h"

namespace yl {

constexpr int C3_TR = 8, C3_TC = 32;  // (4 x 32: 41 instead of 71 KiB of LDS at c = 32, but 2 x the halo: 47.4-48.4 k against 48.9 k images/s)
constexpr int C3_BR = C3_TR + 4, C3_BC = C3_TC + 4;   // b tile with halo 2
constexpr int C3_MR = C3_TR + 2, C3_MC = C3_TC + 2;   // intermediate with halo 1

// CP = c / 8 planes per half (2: c = 16, layer 2 of YOLO11n; 4: c = 32, layers 4 and 16); NCB = blocks of 64 output channels
template <int CP>
constexpr size_t c3k2s_lds() {
  return (size_t)(CP * C3_BR * C3_BC + CP * C3_TR * C3_TC + (CP / 2) * C3_MR * C3_MC + CP * C3_TR * C3_TC) * 16;
}

template <int CP, int NCB>
// (waves_per_eu(2): MFMA results in VGPRs, see stem2_kernel)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void c3k2s_kernel(C3k2sArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char c3mem[];
  constexpr int MP = CP / 2;                       // planes of the intermediate
  constexpr int KA = (9 * CP + 3) / 4;             // k-steps of Bottleneck.cv1 (c -> c/2), one tile of 16 rows
  constexpr int KB = (9 * MP + 3) / 4;             // Bottleneck.cv2 (c/2 -> c)
  constexpr int MTB = CP / 2;                      // its channel tiles: 1 (c = 16) or 2 (c = 32)
  constexpr int KC = (3 * CP + 3) / 4;             // C3k2.cv2 (3c -> 64 * NCB), 4 tiles per block
  unsigned char* const tb = c3mem;                                   // [CP][C3_BR][C3_BC] x 16 B
  unsigned char* const ta = tb + CP * C3_BR * C3_BC * 16;            // [CP][C3_TR][C3_TC]
  unsigned char* const tm = ta + CP * C3_TR * C3_TC * 16;            // [MP][C3_MR][C3_MC]
  unsigned char* const tc = tm + MP * C3_MR * C3_MC * 16;            // [CP][C3_TR][C3_TC]
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

  // ---- 0. a (planes 0..CP-1) and b with halo (planes CP..2CP-1) -> LDS, every load of a thread in flight at once
  {
    constexpr int NB = CP * C3_BR * C3_BC, NA = CP * C3_TR * C3_TC;
    constexpr int NL = (NB + NA + 255) / 256;
    bf16x8 v[NL];
    unsigned dst[NL];
    const int pix0 = (int)a.tin.g0 + b * (int)a.tin.img, w1 = (int)a.tin.W1, ppx = (int)a.plane_px;
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int i = tid + u * 256;
      int pl, yy, xx, d;
      if (i < NB) {
        pl = i / (C3_BR * C3_BC);
        const int r = i - pl * (C3_BR * C3_BC);
        const int ry = r / C3_BC, rx = r - ry * C3_BC;
        yy = y0 - 2 + ry;
        xx = x0 - 2 + rx;
        d = i * 16;
        pl += CP;
      } else {
        const int j = i - NB;
        pl = j / (C3_TR * C3_TC);
        const int r = j - pl * (C3_TR * C3_TC);
        const int ry = r / C3_TC, rx = r - ry * C3_TC;
        yy = y0 + ry;
        xx = x0 + rx;
        d = (int)(ta - tb) + j * 16;
      }
      // guarded index; may be negative for image 0 (-> out of range -> zeros).  32-bit: a view is at most 1 GiB (engine.hip)
      const int pix = pix0 + yy * w1 + xx;
      const bool ok = i < NB + NA && pix >= 0 && pix < ppx;
      v[u] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                            rin, ok ? (unsigned)pl * in_plane + (unsigned)pix * 16u : OOB_LANE, 0, 0));
      dst[u] = i < NB + NA ? (unsigned)d : 0xffffffffu;
    }
#pragma unroll
    for (int u = 0; u < NL; ++u)
      if (dst[u] != 0xffffffffu) *reinterpret_cast<bf16x8*>(tb + dst[u]) = v[u];
  }
  // filter fragments of the two 3x3 convs (conv_pack_weights order) stay in registers; the 1x1's are fetched per block
  bf16x8 wA[KA], wB[KB][MTB];
#pragma unroll
  for (int ks = 0; ks < KA; ++ks) wA[ks] = reinterpret_cast<const bf16x8*>(a.wA)[ks * 64 + lane];
#pragma unroll
  for (int ks = 0; ks < KB; ++ks)
#pragma unroll
    for (int mt = 0; mt < MTB; ++mt) wB[ks][mt] = reinterpret_cast<const bf16x8*>(a.wB)[(ks * MTB + mt) * 64 + lane];
  const f32x4 biasA = *reinterpret_cast<const f32x4*>(a.bA + g * 4);  // chan_in_block(1, 0, g, r) = 4g + r
  f32x4 biasB[MTB];
#pragma unroll
  for (int mt = 0; mt < MTB; ++mt) biasB[mt] = *reinterpret_cast<const f32x4*>(a.bB + chan_in_block(MTB, mt, g, 0));
  __syncthreads();

  // ---- A. mid = SiLU(cv1(b)), c -> c/2 channels, on C3_MR x C3_MC; local (r, c) <-> map (y0 - 1 + r, x0 - 1 + c)
  {
    int boff[KA];  // this lane's B fragment of k-step ks relative to the pixel's corner in tb
#pragma unroll
    for (int ks = 0; ks < KA; ++ks) {
      const int k8 = ks * 4 + g;
      const int tap = min(k8 / CP, 8), c8 = k8 % CP;  // k groups past tap 8 have zero filters: any finite data
      const int kh = tap / 3, kw = tap - kh * 3;
      boff[ks] = ((c8 * C3_BR + kh) * C3_BC + kw + n) * 16;
    }
    // 16-pixel tiles: two per row (columns 0 .. 31) and the two remaining columns of ALL rows gathered into tiles of their own
    // (pixel q of them = row q / 2, column 32 + q % 2) -- as a third, 2-pixel-wide tile per row they were a third of this phase's
    // MFMAs, SiLUs and stores for a sixteenth of its pixels (the kernel is bound by instruction issue, NOTEBOOK section 13)
    constexpr int TPRF = C3_MC / 16, NFULL = C3_MR * TPRF;
    constexpr int NRAG = (C3_MC - TPRF * 16) * C3_MR, NTILES = NFULL + (NRAG + 15) / 16;
    static_assert(C3_MC - TPRF * 16 == 2, "two ragged columns");
    for (int t = wave; t < NTILES; t += 4) {
      int r, c;     // the lane's pixel of the intermediate
      bool live;
      if (t < NFULL) {  // scalar branch
        const int rs = t / TPRF, ct = t - rs * TPRF;
        r = rs;
        c = ct * 16 + n;
        live = true;
      } else {
        const int q = (t - NFULL) * 16 + n;
        live = q < NRAG;
        const int qc = live ? q : NRAG - 1;
        r = qc >> 1;
        c = TPRF * 16 + (qc & 1);
      }
      const unsigned char* corner = tb + (r * C3_BC + c - n) * 16;  // (boff carries the lane's + n)
      f32x4 acc = biasA;
#pragma unroll
      for (int ks = 0; ks < KA; ++ks)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wA[ks], *reinterpret_cast<const bf16x8*>(corner + boff[ks]), acc, 0, 0, 0);
      const int my = y0 - 1 + r, mx = x0 - 1 + c;
      const bool inside = (unsigned)my < (unsigned)a.H && (unsigned)mx < (unsigned)a.W;
      bf16x4 o;
      const f32x4 sa = silu4(acc);
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = inside ? (__bf16)sa[q] : (__bf16)0.f;
      // rows 4g..4g+3 = channels 4g..4g+3 of the c/2 = 4*CP the conv has
      if (g < CP && live)
        *reinterpret_cast<bf16x4*>(tm + (((g >> 1) * C3_MR + r) * C3_MC + c) * 16 + (g & 1) * 8) = o;
    }
  }
  __syncthreads();

  // ---- B. c_out = b + SiLU(cv2(mid)), c/2 -> c channels, on the tile
  {
    int boff[KB];
#pragma unroll
    for (int ks = 0; ks < KB; ++ks) {
      const int k8 = ks * 4 + g;
      const int tap = min(k8 / MP, 8), c8 = k8 % MP;
      const int kh = tap / 3, kw = tap - kh * 3;
      boff[ks] = ((c8 * C3_MR + kh) * C3_MC + kw + n) * 16;
    }
    constexpr int TPR = C3_TC / 16;
    for (int t = wave; t < C3_TR * TPR; t += 4) {
      const int ly = t / TPR, ct = t - ly * TPR;  // scalar
      const int lx = ct * 16 + n;
      const unsigned char* corner = tm + (ly * C3_MC + ct * 16) * 16;
      f32x4 acc[MTB];
#pragma unroll
      for (int mt = 0; mt < MTB; ++mt) acc[mt] = biasB[mt];
#pragma unroll
      for (int ks = 0; ks < KB; ++ks) {
        const bf16x8 xb = *reinterpret_cast<const bf16x8*>(corner + boff[ks]);
#pragma unroll
        for (int mt = 0; mt < MTB; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wB[ks][mt], xb, acc[mt], 0, 0, 0);
      }
      // conv_kernel's epilogue order: SiLU, then += residual (b at the pixel), then the bf16 rounding
      if constexpr (MTB == 1) {  // lane = channels 4g..4g+3: half a pixel of plane g >> 1
        const bf16x4 rb = *reinterpret_cast<const bf16x4*>(tb + (((g >> 1) * C3_BR + ly + 2) * C3_BC + lx + 2) * 16 + (g & 1) * 8);
        bf16x4 o;
        const f32x4 s0 = silu4(acc[0]);
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (__bf16)(s0[q] + (float)rb[q]);
        *reinterpret_cast<bf16x4*>(tc + (((g >> 1) * C3_TR + ly) * C3_TC + lx) * 16 + (g & 1) * 8) = o;
      } else {                   // two tiles: channels 8g..8g+7 (chan_in_block(2, ..)) = the pixel of plane g
        const bf16x8 rb = *reinterpret_cast<const bf16x8*>(tb + ((g * C3_BR + ly + 2) * C3_BC + lx + 2) * 16);
        bf16x8 o;
        const f32x4 s0 = silu4(acc[0]), s1 = silu4(acc[MTB - 1]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          o[q] = (__bf16)(s0[q] + (float)rb[q]);
          o[4 + q] = (__bf16)(s1[q] + (float)rb[4 + q]);
        }
        *reinterpret_cast<bf16x8*>(tc + ((g * C3_TR + ly) * C3_TC + lx) * 16) = o;
      }
    }
  }
  // the first block of 1x1 filters is requested before the barrier
  bf16x8 wC[KC][4];
#pragma unroll
  for (int ks = 0; ks < KC; ++ks)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) wC[ks][mt] = reinterpret_cast<const bf16x8*>(a.wC)[(ks * 4 + mt) * 64 + lane];
  __syncthreads();

  // ---- C. y = SiLU(cv2'(cat(a, b, c_out))): k groups [0, CP) = a, [CP, 2CP) = b, [2CP, 3CP) = c_out, beyond: zero filters
  {
    const unsigned char* src[KC];  // per lane: its k group's tile / plane in k-step ks, and that tile's row pitch
    int pitch[KC];
#pragma unroll
    for (int ks = 0; ks < KC; ++ks) {
      const int cl = min(ks * 4 + g, 3 * CP - 1);  // groups past the last: any finite data
      src[ks] = cl < CP ? ta + cl * C3_TR * C3_TC * 16
                        : cl < 2 * CP ? tb + (((cl - CP) * C3_BR + 2) * C3_BC + 2) * 16 : tc + (cl - 2 * CP) * C3_TR * C3_TC * 16;
      pitch[ks] = (cl >= CP && cl < 2 * CP ? C3_BC : C3_TC) * 16;
    }
    const unsigned out_plane = (unsigned)(a.yout.plane * 2);
    constexpr int TPR = C3_TC / 16;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      f32x4 biasC[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) biasC[mt] = *reinterpret_cast<const f32x4*>(a.bC + cb * 64 + chan_in_block(4, mt, g, 0));
      for (int t = wave; t < C3_TR * TPR; t += 4) {
        const int ly = t / TPR, ct = t - ly * TPR;  // scalar
        const int lx = ct * 16 + n;
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = biasC[mt];
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) {
          const bf16x8 xf = *reinterpret_cast<const bf16x8*>(src[ks] + ly * pitch[ks] + lx * 16);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wC[ks][mt], xf, acc[mt], 0, 0, 0);
        }
        const int oy = y0 + ly, ox = x0 + lx;
        const unsigned po = (oy < a.H && ox < a.W) ? (unsigned)(a.yout.g0 + b * a.yout.img + oy * a.yout.W1 + ox) * 16u : OOB_LANE;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          float v[8];
          const f32x4 s0 = silu4(acc[2 * p]), s1 = silu4(acc[2 * p + 1]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = s0[r];
            v[4 + r] = s1[r];
          }
          store_b128_nt(u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])},
                        rout, (unsigned)(cb * 8 + p * 4 + g) * out_plane + po, a.st_nt);
        }
      }
      if (cb + 1 < NCB) {
#pragma unroll
        for (int ks = 0; ks < KC; ++ks)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
            wC[ks][mt] = reinterpret_cast<const bf16x8*>(a.wC)[(((cb + 1) * KC + ks) * 4 + mt) * 64 + lane];
      }
    }
  }
}

template <int CP, int NCB>
static int c3k2s_launch_t(const C3k2sArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((c3k2s_kernel<CP, NCB>), dim3((unsigned)(a.B * a.tiles_y * a.tiles_x)), dim3(256), c3k2s_lds<CP>(), s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

int c3k2s_launch(const C3k2sArgs& a0, hipStream_t s) {
  C3k2sArgs a = a0;
  a.st_nt = (a.cacheable || !tune_int("YL_ST_NT_FUSED", 1)) ? 0 : st_nt_policy(a.out_bytes);
  a.tiles_x = (a.W + C3_TC - 1) / C3_TC;
  a.tiles_y = (a.H + C3_TR - 1) / C3_TR;
  switch (a.c * 1000 + a.cout) {
    case 16064: return c3k2s_launch_t<2, 1>(a, s);
    case 32064: return c3k2s_launch_t<4, 1>(a, s);
    case 32128: return c3k2s_launch_t<4, 2>(a, s);
  }
  return fail(-3, "c3k2: supported widths are c = 16 -> 64 and c = 32 -> 64 / 128 output channels");
}

}  // namespace yl
