// 3x3 stride-1 Conv 64 -> 64 + (folded BN) bias + SiLU, filter in registers, output channels split between wave pairs -- and the
// activation fragments REUSED ACROSS ROWS ("strip walk", round 5; reference nn/modules/conv.py:35-53, nn/modules/head.py:29-57).
//
// conv3_cs_kernel gives a wave 32 of the 64 output channels and a 64-position run of a LINEAR tile: every one of its 18 k-steps
// reads its activation fragments from LDS, 0.5 reads of 1 KiB per MFMA -- with two such waves per SIMD the LDS array is exactly
// as busy as the matrix pipe (4608 cycles each per CU and tile pair), and each runs at half its rate (NOTEBOOK section 12:
// "52 % of peak while it runs").  The three row taps of a 3x3 conv read the SAME fragments for vertically neighbouring outputs,
// so here a wave owns a STRIP -- 16 pixels wide -- and walks down R rows of it: the fragments of three input rows (3 column taps
// x 2 channel halves x 3 rows = 18 fragments, 72 registers) are a rolling window, one new row (6 reads) per output row of 36
// MFMAs: 0.17 reads per MFMA (0.23 with the two halo rows of a walk) -- the k-loop is bound by the matrix pipe, not by LDS.
// The accumulators of one output row are 8 registers, its epilogue 8 SiLUs and one 16-byte store per lane: it runs row by row
// between the MFMA blocks, under the MFMAs of the workgroup that shares the CU.
//
//   workgroup (4 waves) = strip x 2 R rows x 64 channels: wave (h, b) = channel half h, row band b (R rows each)
//   LDS: [8 planes][2 R + 2 rows][18 pixels] x 16 B = 27 KiB per tile (the strip, one pixel left / right, one row above / below;
//        in the guarded layout those are data or zeros), TWO buffers: the next tile's 27 LDS-DMA pieces (per-lane source
//        addresses, the destination is linear) are issued before this tile is computed
//   persistent workgroups, two per CU (110 KiB of LDS, <= 256 registers)
// Same operands, same k order (tap-major, 32 channels per k-step), same accumulation chain from the bias, same epilogue
// arithmetic as conv_kernel: bit-identical results (tests/test_gpu_kernels.py).
// Takes maps whose width is a multiple of 16 and whose height is a multiple of 2 R (80x80, 160x160 ...); others stay on
// conv3_cs_kernel.
#include "conv_common.h"

namespace yl {

typedef __attribute__((address_space(3))) void lds_void_w;

constexpr int SW_R = 5;                       // output rows per wave and tile
constexpr int SW_ROWS = 2 * SW_R + 2;         // staged rows
constexpr int SW_PX = 18;                     // staged pixels per row
constexpr int SW_POS = 8 * SW_ROWS * SW_PX;   // 16-byte positions per buffer (1728)
constexpr int SW_NP = SW_POS / 64;            // 1 KiB DMA pieces per tile (27)
constexpr int SW_PW = (SW_NP + 3) / 4;        // pieces per wave (7; wave 3: 6)
static_assert(SW_POS % 64 == 0, "whole pieces");

template <bool ACT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv3_sw_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char swm[];
  constexpr int KS = 18, BUF = SW_POS * 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = wave >> 1, b = wave & 1;  // channel half, row band
  const int g = lane >> 4, n = lane & 15;
  const int G = gridDim.x;
  const int wg = xcd_remap(blockIdx.x, G);
  const int nmine = (a.t_ntiles - wg + G - 1) / G;
  const int nstrips = a.t_W >> 4, tpi = nstrips * (a.t_H / (2 * SW_R));  // tiles per image
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, (short)0, (a.dbg & 1) ? 0 : a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, a.out_bytes, 0x00020000);

  // this lane's source offset of its pieces relative to the tile's first staged position (plane 0, row -1, pixel -1):
  // piece p = wave + 4 j covers positions 64 p .. 64 p + 63 of the linear [plane][row][pixel] buffer
  unsigned rel[SW_PW];
#pragma unroll
  for (int j = 0; j < SW_PW; ++j) {
    const int i = (wave + 4 * j) * 64 + lane;
    const int pl = i / (SW_ROWS * SW_PX), r = (i / SW_PX) % SW_ROWS, px = i % SW_PX;
    rel[j] = (unsigned)pl * a.in_plane + (unsigned)(r * a.in_w1 + px) * 16u;
  }
  // tile t = (image, band pair, strip): byte offset of its first staged position in plane 0 (>= 0: g0 = W1 + 1)
  auto tile_base = [&](int t, int& img, int& y0, int& x0) __attribute__((always_inline)) {
    img = t / tpi;
    const int rem = t - img * tpi;
    const int k = rem / nstrips;
    x0 = (rem - k * nstrips) * 16;
    y0 = k * (2 * SW_R);
    return (unsigned)(a.in_g0 + img * a.in_img + (y0 - 1) * a.in_w1 + (x0 - 1)) * 16u;
  };
  auto stage = [&](int t, int buf) __attribute__((always_inline)) {
    int img, y0, x0;
    const unsigned base = tile_base(t, img, y0, x0);
#pragma unroll
    for (int j = 0; j < SW_PW; ++j) {
      const int p = wave + 4 * j;  // scalar
      if (p < SW_NP)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void_w*)(swm + buf * BUF + p * 1024), 16, (int)(rel[j] + base), 0, 0, 0);
    }
  };
  if (nmine > 0) stage(wg, 0);
  // (measured: a start delay of 1 / 2 / 4 / 8 x 1024 clocks for the second half of the grid -- the workgroups that share a CU with
  // the first half -- changes nothing: 27.1 us either way; the phases of this kernel add up for another reason than lock step)

  // this wave's half of the filter (conv_pack_weights order, 4 tiles per k-step: tiles 2h, 2h + 1) and its bias
  bf16x8 wf[KS][2];
  {
    const bf16x8* wsrc = reinterpret_cast<const bf16x8*>(a.wpk) + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) wf[ks][mt] = wsrc[(ks * 4 + 2 * h + mt) * 64];
  }
  f32x4 bias_r[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) bias_r[mt] = *reinterpret_cast<const f32x4*>(a.bias + chan_in_block(4, 2 * h + mt, g, 0));
  // the lane's fragment of (plane g, staged row b R, pixel n): window row 0 of this wave = one row above its first output row
  const int lane_off = ((g * SW_ROWS + b * SW_R) * SW_PX + n) * 16;
  constexpr int HALF = 4 * SW_ROWS * SW_PX * 16;  // planes 4..7 = input channels 32..63
  const unsigned oplane = (unsigned)(4 * h + g) * a.out_plane;  // the lane's output plane (8 channels)

  for (int i = 0; i < nmine; ++i) {
    const int t = wg + i * G;
    // my pieces of this tile (issued before the previous tile was computed: older than its SW_R epilogue stores, which stay in flight)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SW_R) : "memory");
    __builtin_amdgcn_s_barrier();                     // the tile has landed; everybody has left the other buffer
    if (i + 1 < nmine) stage(t + G, (i + 1) & 1);
    int img, y0, x0;
    (void)tile_base(t, img, y0, x0);
    const unsigned char* wb = swm + (i & 1) * BUF + lane_off;
    unsigned opos = (unsigned)(a.out_g0 + img * a.out_img + (y0 + b * SW_R) * a.out_w1 + x0 + n) * 16u + oplane;
    // rolling window of fragments: Wn[row % 3][dx][channel half]
    bf16x8 Wn[3][3][2];
    auto load_row = [&](int r) __attribute__((always_inline)) {  // window row r (0 .. R + 1) of this wave
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
          Wn[r % 3][dx][ch] = *reinterpret_cast<const bf16x8*>(wb + ch * HALF + (r * SW_PX + dx) * 16);
    };
    load_row(0);
    load_row(1);
#pragma unroll
    for (int r = 0; r < SW_R; ++r) {
      load_row(r + 2);
      f32x4 acc[2] = {bias_r[0], bias_r[1]};
      if (!(a.dbg & 4)) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
              const int ks = 2 * (3 * dy + dx) + ch;
#pragma unroll
              for (int mt = 0; mt < 2; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][mt], Wn[(r + dy) % 3][dx][ch], acc[mt], 0, 0, 0);
            }
      }
      if (!(a.dbg & 2)) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[q] = acc[0][q];
          v[4 + q] = acc[1][q];
        }
        if constexpr (ACT) silu8(v);
        store_b128_nt(u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])},
                      rout, opos, a.st_nt);
      }
      opos += (unsigned)a.out_w1 * 16u;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (my reads of this buffer are done before the next barrier lets it be refilled)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA of mine may still be in flight when the LDS is released
}

// ------------------------------------------------------------------------------------ host
// sets cfg->sw for plain 64 -> 64 3x3 stride-1 convs on maps that tile into strips (width % 16 == 0, height % (2 R) == 0) with
// enough tiles for two persistent workgroups per CU
void conv3_sw_choose(int batch, int Cin, int Cout, int k, int stride, int Ho, int Wo, int post, ConvConfig* c) {
  c->sw = 0;
  // OPT-IN (YL_CONV_SW=1).  Measured (batch 32): stand-alone with cold inputs 26.7 us against 27.7 us for conv3_cs_kernel @80x80 and
  // 79.0 against 88.8 us @160x160 (765 TFLOP/s = 0.31 of the bf16 peak) -- but IN THE NETWORK, where layer 16's output is still in
  // the Infinity Cache, `model.23.cv2.0.0` takes 27.7 us against 25.4 us: the strip's 288-byte rows fetch whole 128-byte lines for
  // 18 pixels and arrive by 27 gather pieces per 160 outputs, and that costs more than the saved LDS reads bring at 2.5 tiles per
  // workgroup.  Phase ablations (YL_CONV_DBG) of both kernels add up to their totals: MFMA loop 5.9-6.1 us, epilogue 6.8-7.7,
  // staging 4.2-6.0, skeleton 6 (NOTEBOOK section 15).
  const int enable = getenv("YL_CONV_SW") ? atoi(getenv("YL_CONV_SW")) : 0;
  const int min_tiles = getenv("YL_CONV_SW_MIN") ? atoi(getenv("YL_CONV_SW_MIN")) : 512;
  if (!enable || k != 3 || stride != 1 || post || Cin != 64 || Cout != 64 || c->ncb != 1 || c->MT != 4) return;
  if (Wo % 16 || Ho % (2 * SW_R)) return;
  const long long ntiles = (long long)batch * (Wo / 16) * (Ho / (2 * SW_R));
  if (ntiles < min_tiles || ntiles >= (1LL << 30)) return;
  c->sw = 1;
  c->cs = c->reg = c->tile = c->pair = 0;
  c->name = "conv3x3_sw_mfma";
}

template <bool ACT>
static int launch_sw(const ConvArgs& a, int grid, hipStream_t s) {
  auto kern = conv3_sw_kernel<ACT>;
  const size_t lds = (size_t)2 * SW_POS * 16;
  if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), (int)(160 * 1024))) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

int conv3_sw_launch(const ConvConfig& cfg, ConvArgs a, int batch, hipStream_t s) {
  if (a.stride != 1 || a.out_f32 || a.in2 || a.wpk2 || a.out8 || a.res || a.in_w1 != a.out_w1 || a.in_img != a.out_img ||
      a.in_g0 != a.out_g0 || a.in_w1 != a.Wo + 1 || a.Cin != 64 || a.Cout != 64 || a.Wo % 16 || a.Ho % (2 * SW_R))
    return fail(-3, "conv3_sw: needs a plain 64 -> 64 3x3 stride-1 conv without residual on a map of whole strips");
  a.t_B = batch;
  a.t_H = a.Ho;
  a.t_W = a.Wo;
  a.t_ntiles = batch * (a.Wo / 16) * (a.Ho / (2 * SW_R));
  const long long lim = 1LL << 30;
  const long long inb = 8LL * a.in_plane, outb = 8LL * a.out_plane;
  if (inb > lim || outb > lim) return fail(-3, "conv3_sw: a tensor slice exceeds 1 GiB; lower the batch size");
  a.in_bytes = (unsigned)inb;
  a.out_bytes = (unsigned)outb;
  a.dbg = getenv("YL_CONV_DBG") ? atoi(getenv("YL_CONV_DBG")) : 0;
  const int resident = 2 * device_cu_count();  // two persistent workgroups per CU
  const int grid = a.t_ntiles < resident ? a.t_ntiles : resident;
  return a.act ? launch_sw<true>(a, grid, s) : launch_sw<false>(a, grid, s);
}

}  // namespace yl
