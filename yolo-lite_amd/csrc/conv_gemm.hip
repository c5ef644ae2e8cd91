// GEMM-class 1x1 Conv + (folded BN) bias + SiLU [+ residual] for WIDE layers (Cin, Cout >= 128) on gfx950.
//
// Same arithmetic as conv_kernel (reference nn/modules/conv.py:35-53 with the BN folded per utils/torch_utils.py:182-209;
// the residual is the PSABlock / Bottleneck shortcut of nn/modules/block.py:341-343, 947-950): same packed filters
// (conv_pack_weights order, rows permuted so that a lane's two adjacent tiles are one 16-byte pixel), same k order, same
// fp32 accumulation chain from the bias, same epilogue -- results are bit-identical to conv_kernel's.  What changes is the
// operand delivery.  conv_kernel gives every wave a 64 x 64 tile whose activation fragments it fetches itself from the planes
// (texture path) while its four waves share one filter slab: on wide layers the activations are re-read once per 64 output
// channels and the step is bound by the texture addresser + LDS together.  Here a workgroup owns a
// macro tile of BM = 128, 192 or 256 output channels x BN = 128 pixels and BOTH operands are staged through LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction, no VGPRs, no ds_write) through a ring of 3-4 K-slices of 32
// channels; the four waves form a 2 x 2 grid of (BM / 2) x 64 register tiles, so every activation byte is fetched once per
// BM channels and shared by two waves, every filter byte once per workgroup and shared by two waves.
//
//   * A 1x1 stride-1 conv is pointwise in the guarded planar layout (yl_internal.h): output position q depends on input
//     position q only, so the pixel tile is simply 128 CONSECUTIVE guarded positions [q0, q0 + 128) of every plane -- guard
//     pixels included; they are computed and not stored (1-10 % of the positions).  A B stage is 4 planes x 2 KiB, each
//     plane's run two contiguous 1 KiB DMA pieces; in LDS it is [plane][128 px][16 B] = MFMA B-fragment order, read with
//     conflict-free ds_read_b128 (16 consecutive lanes = 256 contiguous bytes; the four k-groups of a fragment sit 2 KiB
//     apart = the same banks, but in different lane groups of the instruction).
//   * An A stage is one k-step x BM / 16 tiles of the packed filter = one contiguous run of BM / 16 KiB in memory, copied
//     linearly; fragment order, conflict-free.
//   * One raw s_barrier per K-slice: a COUNTED wait for my own DMA pieces of slice i (the pieces of the NS - 2 younger
//     slices stay in flight), barrier (everybody's pieces landed, everybody has finished reading slice i - 1), refill that
//     buffer with slice i + NS - 1, compute slice i.  Why a ring and why big tiles (measured, first version: 128 x 128,
//     two buffers of 64 channels, 570-860 TFLOP/s): under load an L2 -> LDS piece takes ~2500 cycles to land, so the rate
//     is (bytes in flight per CU) / latency -- 64 KiB in flight gave ~22 B/clk per CU -- and only fewer bytes per flop
//     (bigger macro tiles) or more bytes in flight (deeper rings) raise it.
//     Two workgroups share a CU (64-80 KiB of LDS, <= 256 registers), so one's epilogue (SiLU is quarter rate) and
//     barrier waits run under the other's MFMAs.
#include <type_traits>

#include "conv_common.h"

#ifndef YL_GEMM_ABL
#define YL_GEMM_ABL 0  // development build switch: timing ablations of the main loop (wrong results): 2 no operand traffic, 4 no fragment
                       // reads, 8 nothing stored, 16 activations always from the first 8192 positions (L2 hits)
#endif
#ifndef YL_GEMM_LD_AUX
#define YL_GEMM_LD_AUX 0
#endif
namespace yl {

typedef __attribute__((address_space(3))) void lds_void_g;

// Ring shapes.  Four waves (2 x 2, BN = 128 pixels), two workgroups per CU: 128 / 192-channel blocks take two K-slices of 64
// channels (64 / 80 KiB; fewer barriers: 384 -> 384 @320x320 x 8: 664 against 597 TFLOP/s with four slices of 32),
// 256-channel blocks three slices of 32 (72 KiB); a K that is not a multiple of 64 takes four slices of 32.
// Eight waves (2 x 4, BN = 256 pixels), one workgroup per CU: 1.3-1.5 x the MACs per staged byte.
// 64-channel blocks (MTW = 2, round 5: layers that are all activation traffic -- 256 -> 64 @80x80 behind Upsample + Concat is
// 92 MB for 6.7 GFLOP): a slice is 24 KiB (8 KiB of filter, 16 KiB of activations) and a ring of three of them keeps 48 KiB per
// workgroup in flight; a K that is not a multiple of 64 takes four slices of 32.
#ifndef YL_GM2_KS
#define YL_GM2_KS 2  // development build switches: ring shape of the 64-channel blocks (even K)
#endif
#ifndef YL_GM2_NS
#define YL_GM2_NS 3
#endif
#ifndef YL_GM6_KS
#define YL_GM6_KS 2  // development build switches: ring shapes of the 192-channel blocks (even K), four waves / eight waves
#endif
#ifndef YL_GM6_NS
#define YL_GM6_NS 2
#endif
#ifndef YL_GM6W_KS
#define YL_GM6W_KS 2
#endif
#ifndef YL_GM6W_NS
#define YL_GM6W_NS 2
#endif
#ifndef YL_GM4_KS
#define YL_GM4_KS 2  // ... and of the 128-channel blocks (four waves, even K)
#endif
#ifndef YL_GM4_NS
#define YL_GM4_NS 2
#endif
constexpr int gm_ks(int MTW, int NWN, bool even) {
  if (MTW == 4 && NWN == 2 && even) return YL_GM4_KS;
  if (MTW == 6 && even) return NWN == 4 ? YL_GM6W_KS : YL_GM6_KS;
  return NWN == 4 ? (MTW == 6 ? 2 : 1) : (MTW == 2 && even ? YL_GM2_KS : ((MTW == 8 || !even) ? 1 : 2));
}
constexpr int gm_ns(int MTW, int NWN, bool even) {
  if (MTW == 4 && NWN == 2 && even) return YL_GM4_NS;
  if (MTW == 6 && even) return NWN == 4 ? YL_GM6W_NS : YL_GM6_NS;
  return NWN == 4 ? (MTW == 6 ? 2 : 3) : (MTW == 8 ? 3 : (even ? (MTW == 2 ? YL_GM2_NS : 2) : 4));
}
size_t conv_gemm_lds_bytes(int MTW, int NWN, bool even) {
  return (size_t)gm_ns(MTW, NWN, even) * gm_ks(MTW, NWN, even) * ((size_t)MTW * 2048 + (size_t)4 * 64 * NWN * 16);
}

// MTW: 16-channel tiles per wave (4, 6 or 8): BM = 2 * MTW * 16 output channels per workgroup.  NWN: waves along the pixels
// (2 or 4; 64 pixels each): BN = 64 * NWN.  KS: k-steps (32 channels) per ring slice; NS: slices in the ring.
// O8: the epilogue also stores the e4m3 copy of channels [out8_c0, out8_c0 + out8_n) (yl_op.out8, fp8 programs), as conv_kernel's does
// UP: the first 32 * ks_split input channels are a nearest-neighbour 2x upsampling of a low-resolution map (nn.Upsample + Concat in
// front of the conv, yolo11.yaml:31-36) and are read from that map itself at pixel (y / 2, x / 2): the LDS-DMA source address is
// per lane (the LDS destination stays linear), so the gather costs one address per lane and tile.
// GEO: 0 = 1x1 conv; 1 = 3x3 stride 1; 2 = 3x3 stride 2 (round 3).  A 3x3 conv is nine shifted 1x1 GEMMs accumulated: in the
// guarded layout a tap is a constant offset of the source position (stride 1) or of a per-lane source position (stride 2:
// the lane's output pixel (y, x) reads input pixel (2y + dy, 2x + dx); one offset per lane and tile, as for UP).  K order:
// k-step = (block of 4 planes, tap), the nine taps of a plane block back to back -- a workgroup's input footprint is swept
// once per plane block, not once per tap (the reuse distance that made the tap-major conv_kernel fetch layer 3 twice).
template <int MTW, int NWN, int KS, int NS, bool RES, bool ACT, bool O8, bool UP = false, int GEO = 0>
__global__ __launch_bounds__(128 * NWN) __attribute__((amdgpu_waves_per_eu(2))) void conv_gemm_kernel(ConvArgs a) {
  static_assert(GEO == 0 || (!UP && !O8), "3x3: plain bf16 convs");
  extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
  constexpr int NW = 2 * NWN;                  // waves
  constexpr int BN = 64 * NWN;                 // pixels per workgroup
  constexpr int BSTEP = 4 * BN * 16;           // bytes of the B operand of one k-step (4 planes)
  constexpr int MT = 2 * MTW;                  // tiles per workgroup
  constexpr int ASTAGE = KS * MT * 1024;       // bytes of one A slice (KS k-steps of the packed filter block: contiguous)
  constexpr int BSTAGE = KS * BSTEP;
  constexpr int STAGE = ASTAGE + BSTAGE;
  constexpr int NPA = ASTAGE / 1024, NPB = BSTAGE / 1024, NP = NPA + NPB;  // 1 KiB DMA pieces per slice
  constexpr int PW = NP / NW;                  // pieces per wave and slice
  static_assert(NP % NW == 0 && NPA % NW == 0, "pieces divide over the waves");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  const int wm = wave / NWN, wn = wave % NWN;
  const int nk = a.ksteps / KS;                // K-slices per tile

  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, (short)0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, (short)0, (unsigned)a.ncb * a.ksteps * MT * 1024u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rin2 =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.in2, (short)0, UP ? a.in2_bytes : 0, 0x00020000);
  const unsigned lane16 = (unsigned)lane * 16u;
  const int split_planes = UP ? 4 * a.ks_split : 0;

  // Persistent workgroups: the grid is min(tiles, resident workgroups) and workgroup w takes tile j * grid + remap(w) in
  // round j.  The ring of K-slices runs THROUGH the tile boundaries -- the last NS - 1 refills of a tile fetch the first
  // slices of the next one -- so a tile's epilogue (SiLU + stores) runs with the next tile's operands in flight instead of
  // in front of a cold prologue.  (Measured on 384 / 768 / 1536 -> 768 @160x160 x 8 with one tile per workgroup:
  // 191 / 279 / 461 us = 1.34 PFLOP/s marginal + 103 us that do not depend on K.)
  // Within a round the tiles of one XCD are consecutive (the channel blocks of a pixel tile share its L2).
  const int G = (int)gridDim.x;
  auto tile_of = [&](int j) __attribute__((always_inline)) {
    const int left = a.nblocks - j * G;
    const int cnt = left < G ? left : G;
    return (int)blockIdx.x < cnt ? j * G + xcd_remap((int)blockIdx.x, cnt) : -1;
  };
  // Where a tile's operands come from: q0 = first guarded position (from the start of the plane), wbase = byte offset of the
  // filter block, past = OOB_K for "no tile" (the pieces then move nothing but still count in vmcnt).
  // UP: off2 = byte offset of this lane's pixel in a plane of the low-resolution source.  Piece j = wave + NW * i of a slice
  // always covers pixel part h = wave % NWN of the tile (NW and the A piece count are multiples of NWN): one offset per lane.
  struct Src {
    int q0, cb;
    unsigned wbase, past, off2;
  };
  auto src_of = [&](int tile) __attribute__((always_inline)) {
    Src x;
    const int t = tile < 0 ? 0 : tile;
    x.cb = t % a.ncb;
    x.q0 = (t / a.ncb) * BN;
    x.wbase = (unsigned)x.cb * a.ksteps * MT * 1024u;
    x.past = tile < 0 ? OOB_K : 0u;
    x.off2 = 0;
    if constexpr (GEO == 2) {  // byte offset of input pixel (2y, 2x) of this lane's output position, in a plane
      const int rb = x.q0 - a.out_g0 + wn * 64;                     // the wave's 64 positions (wave-uniform start)
      const int b0 = __builtin_amdgcn_readfirstlane((rb < 0 ? 0 : rb) / a.out_img);
      const int r = rb + lane;
      int t2 = r - b0 * a.out_img;                                  // < img + 64: at most one wrap (host: img >= 128)
      const int wrap = t2 >= a.out_img;
      const int b = b0 + wrap;
      t2 -= wrap * a.out_img;
      const int y = (int)__umulhi((unsigned)(t2 < 0 ? 0 : t2), a.magic_w1);
      const int xx = t2 - y * a.out_w1;
      const bool ok = r >= 0 && b < a.t_B && y < a.t_H && xx < a.t_W;
      x.off2 = ok ? (unsigned)(a.in_g0 + b * a.in_img + 2 * y * a.in_w1 + 2 * xx) * 16u : 0u;  // (position 0 is a zero guard)
    }
    if constexpr (UP) {
      const int rb = x.q0 - a.in_g0;                                // tile start relative to pixel (0, 0, 0)
      const int b0 = __builtin_amdgcn_readfirstlane((rb < 0 ? 0 : rb) / a.in_img);
      const int r = rb + wn * 64 + lane;
      int t2 = r - b0 * a.in_img;                                   // < img + BN: at most one wrap (host: img >= BN)
      const int wrap = t2 >= a.in_img;
      const int b = b0 + wrap;
      t2 -= wrap * a.in_img;
      const int y = (int)__umulhi((unsigned)(t2 < 0 ? 0 : t2), a.magic_w1);
      const int xx = t2 - y * a.in_w1;
      const bool ok = r >= 0 && b < a.t_B && y < a.t_H && xx < a.t_W;
      x.off2 = ok ? (unsigned)(a.in2_g0 + b * a.in2_img + (y >> 1) * a.in2_w1 + (xx >> 1)) * 16u : 0u;  // (position 0 is a zero guard)
    }
    return x;
  };

  // K-slice `it` of the tile at `x` -> ring buffer `buf`: pieces wave, wave + NW, ... (wave-uniform addressing, the lane adds
  // its 16 bytes).
  auto issue = [&](const Src& x, int it, int buf) __attribute__((always_inline)) {
    const unsigned sbase = (unsigned)buf * STAGE;
    const unsigned past = (YL_GEMM_ABL & 2) ? OOB_K : x.past;  // (ABL 2: timing ablation, no operand traffic)
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int j = wave + NW * i;  // scalar
      if (i < NPA / NW) {  // (NPA is a multiple of NW: the choice is made at compile time)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_g*)(gsm + sbase + (unsigned)j * 1024u), 16,
                                                 (int)(lane16 + x.wbase + (unsigned)it * ASTAGE + (unsigned)j * 1024u + past), 0, 0, 0);
      } else {
        const int jb = j - NPA, p = jb / NWN, h = jb % NWN;   // plane of the slice (4 per k-step), 64-pixel part of the run
        if constexpr (GEO != 0) {
          const int ksg = it * KS + (p >> 2);                                // k-step of the conv: (plane block, tap)
          const int pb = (int)__umulhi((unsigned)ksg, 477218589u);            // ksg / 9
          const int tapoff = tap_offset<3>(a, ksg - 9 * pb);                  // bytes, may be negative (wraps out of range -> zeros)
          const unsigned src = (unsigned)(pb * 4 + (p & 3)) * a.in_plane + (unsigned)tapoff;
          if constexpr (GEO == 1) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void_g*)(gsm + sbase + ASTAGE + (unsigned)jb * 1024u), 16,
                                                     (int)(lane16 + src + (unsigned)(x.q0 + h * 64) * 16u + past), 0, 0, 0);
          } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void_g*)(gsm + sbase + ASTAGE + (unsigned)jb * 1024u), 16,
                                                     (int)(x.off2 + src + past), 0, 0, 0);
          }
          continue;
        }
        const int gp = it * (4 * KS) + p;                     // plane of the whole input (scalar)
        if (UP && gp < split_planes) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rin2, (lds_void_g*)(gsm + sbase + ASTAGE + (unsigned)jb * 1024u), 16,
                                                   (int)(x.off2 + (unsigned)gp * a.in2_plane + past), 0, 0, 0);
        } else {
          // (ABL 16: timing ablation, every pixel tile reads the first 8192 positions' activations -- all L2 hits)
          const unsigned src = (unsigned)(gp - split_planes) * a.in_plane + (unsigned)(((YL_GEMM_ABL & 16) ? (x.q0 & 0x1fff) : x.q0) + h * 64) * 16u;
          // (YL_GEMM_LD_AUX, development build switch: cache-policy bits of the activation pieces when one channel block covers the
          // layer -- every activation byte is then read exactly once)
          if (YL_GEMM_LD_AUX && a.ncb == 1)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void_g*)(gsm + sbase + ASTAGE + (unsigned)jb * 1024u), 16,
                                                     (int)(lane16 + src + past), 0, 0, YL_GEMM_LD_AUX);
          else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void_g*)(gsm + sbase + ASTAGE + (unsigned)jb * 1024u), 16,
                                                   (int)(lane16 + src + past), 0, 0, 0);
        }
      }
    }
  };
  int tile = tile_of(0);  // (the grid never exceeds the tile count)
  Src cur = src_of(tile);
#pragma unroll
  for (int st = 0; st < NS - 1; ++st) issue(cur, st, st);  // (host: at least NS - 1 slices per tile)
  int gs0 = 0;    // ring position of the tile's slice 0
  int round = 0;
#ifdef YL_STAMPS
  // diagnostic build: shader clocks a wave spends in the counted wait for its own pieces / in the barrier behind it, over the whole
  // launch (tools/gemm_stamps.py); st_t0 = the wave's start
  unsigned long long st_wait = 0, st_bar = 0, st_n = 0;
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
#endif

  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, a.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.res, (short)0, RES ? a.res_bytes : 0, 0x00020000);
  const unsigned char* const abase = gsm + ((unsigned)(wm * MTW) * 64u + (unsigned)lane) * 16u;
  const unsigned char* const bbase = gsm + ASTAGE + ((unsigned)g * BN + (unsigned)(wn * 64 + n)) * 16u;
  struct Frags {
    bf16x8 wa[MTW], xb[4];
  };
#pragma clang loop unroll(disable)
  while (tile >= 0) {
  const int ntile = tile_of(++round);
  const Src nxt = src_of(ntile);
  const int cb = cur.cb, q0 = cur.q0;
  // per-lane bias of its 4 rows of every tile (the accumulators start from it, as in conv_kernel)
  f32x4 acc[MTW][4];
  {
    const float* bp = a.bias + cb * MT * 16;
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bp + chan_in_block(MT, wm * MTW + mt, g, 0));
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = bv;
    }
  }
  // output positions of this lane's four pixel tiles: valid pixels only (guards are computed, never stored)
  unsigned po[4];
  {
    // (output geometry: the same as the input's except behind a stride-2 conv)
    const int r0 = q0 + wn * 64 - a.out_g0;                     // position of the wave's first pixel relative to pixel (0, 0, 0)
    const int b0 = __builtin_amdgcn_readfirstlane((r0 < 0 ? 0 : r0) / a.out_img);  // wave-uniform: one division per tile
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int loc = nt * 16 + n;
      const int r = r0 + loc;                                    // < 0: in front of the first image (guard row)
      int t = r - b0 * a.out_img;                                // position inside image b0, < img + 64 (host: img >= 128)
      const int wrap = t >= a.out_img;
      const int b = b0 + wrap;
      t -= wrap * a.out_img;
      const int y = (int)__umulhi((unsigned)(t < 0 ? 0 : t), a.magic_w1);
      const int x = t - y * a.out_w1;
      const bool ok = r >= 0 && b < a.t_B && y < a.t_H && x < a.t_W;
      po[nt] = (ok && !(YL_GEMM_ABL & 8)) ? (unsigned)(q0 + wn * 64 + loc) * 16u : OOB_LANE;  // (ABL 8: nothing stored)
    }
  }

  // Entering slice i of the tile (ring position gs0 + i): a COUNTED wait for my own DMA pieces of it (the NS - 2 younger
  // slices stay in flight; epilogue stores of the previous tile that are still on their way only make the wait stricter),
  // all my LDS reads done, barrier -- now everybody's pieces of slice i have landed and nobody reads the slice before it
  // any more -- and that slice's buffer is refilled with slice i + NS - 1: of this tile, or of the next one.
  // (Round 5, measured and not kept: a tile's first wait as vmcnt((NS - 2) * PW + the previous epilogue's stores) -- its pieces are
  // older than those stores -- changed nothing: 365.7 against 368.0 us on 384 -> 384 @320x320 x 8; the stores' cost, 50-70 us of that
  // launch by ablation, is not the drain at the tile boundary.)
  auto enter = [&](int i) __attribute__((always_inline)) {
#ifdef YL_STAMPS
    const unsigned long long e0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * PW) : "memory");
    const unsigned long long e1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_barrier();
    const unsigned long long e2 = __builtin_amdgcn_s_memtime();
    st_wait += e1 - e0;
    st_bar += e2 - e1;
    st_n += 1;
#else
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * PW) : "memory");
    __builtin_amdgcn_s_barrier();
#endif
    const int sl = i + NS - 1;
    const bool here = sl < nk;  // (wave-uniform: scalar selects)
    Src x;
    x.q0 = here ? cur.q0 : nxt.q0;
    x.wbase = here ? cur.wbase : nxt.wbase;
    x.past = here ? cur.past : nxt.past;
    x.off2 = here ? cur.off2 : nxt.off2;
    issue(x, here ? sl : sl - nk, (gs0 + sl) % NS);
  };
  auto load = [&](int i, int ks, Frags& f) __attribute__((always_inline)) {
    const unsigned char* ab = abase + ((gs0 + i) % NS) * STAGE + ks * (MT * 1024);
    const unsigned char* bb = bbase + ((gs0 + i) % NS) * STAGE + ks * BSTEP;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) f.xb[nt] = *reinterpret_cast<const bf16x8*>(bb + nt * 256);
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) f.wa[mt] = *reinterpret_cast<const bf16x8*>(ab + mt * 1024);
  };
  auto mma = [&](const Frags& f) __attribute__((always_inline)) {
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wa[mt], f.xb[nt], acc[mt][nt], 0, 0, 0);
  };
  // One k-step on the fragments in F while the fragments of the NEXT k-step (slice i, k-step ks of it) are read from LDS
  // INTO THE SAME REGISTERS: a filter fragment is reloaded two rows of four MFMAs behind its last use, the four activation
  // fragments one by one behind their last use in the last row.  A k-step's first MFMA never waits for LDS and the
  // prefetch costs no second register set (two sets made hipcc ping-pong the accumulators between two register sets as
  // well: 256 registers and spills at 8 tiles per wave).  ENTER: the next k-step lies in a new slice -- the wait + barrier
  // + ring refill stand behind the first two rows, so that the reads issued at the end of the previous k-step have had
  // 8 MFMAs to land before lgkmcnt(0).  The order is pinned (sched_barrier around the slice entry, sched_group_barrier
  // for the MFMA / LDS-read interleave).
  auto row = [&](const Frags& f, int mt) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wa[mt], f.xb[nt], acc[mt][nt], 0, 0, 0);
  };
  auto step = [&](Frags& f, int i, int ks, auto enter_tag) __attribute__((always_inline)) {
    constexpr bool ENTER = decltype(enter_tag)::value;
    const unsigned char* ab = abase + ((gs0 + i) % NS) * STAGE + ks * (MT * 1024);
    const unsigned char* bb = bbase + ((gs0 + i) % NS) * STAGE + ks * BSTEP;
    if constexpr (MTW == 2) {
      // two tiles per wave: eight MFMAs per k-step, too few to thread the six fragment reads through -- the next k-step's
      // fragments go into a second register set (24 registers; this instance has them to spare) while this one's MFMAs run
      if constexpr (ENTER) enter(i);
      Frags nf;
      if constexpr ((YL_GEMM_ABL & 4) == 0) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) nf.xb[nt] = *reinterpret_cast<const bf16x8*>(bb + nt * 256);
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) nf.wa[mt] = *reinterpret_cast<const bf16x8*>(ab + mt * 1024);
      } else {
        nf = f;
      }
      mma(f);
      f = nf;
    } else if constexpr ((YL_GEMM_ABL & 4) != 0) {  // (timing ablation: no fragment reads, the MFMAs re-use the registers)
      if constexpr (ENTER) enter(i);
      mma(f);
    } else {
    row(f, 0);
    row(f, 1);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ENTER) enter(i);  // (s_setprio 1 around the MFMA rows: measured, no change)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 2; mt < MTW - 1; ++mt) {
      f.wa[mt - 2] = *reinterpret_cast<const bf16x8*>(ab + (mt - 2) * 1024);
      row(f, mt);
    }
    f.wa[MTW - 3] = *reinterpret_cast<const bf16x8*>(ab + (MTW - 3) * 1024);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      acc[MTW - 1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.wa[MTW - 1], f.xb[nt], acc[MTW - 1][nt], 0, 0, 0);
      f.xb[nt] = *reinterpret_cast<const bf16x8*>(bb + nt * 256);
    }
    f.wa[MTW - 2] = *reinterpret_cast<const bf16x8*>(ab + (MTW - 2) * 1024);
    f.wa[MTW - 1] = *reinterpret_cast<const bf16x8*>(ab + (MTW - 1) * 1024);
#pragma unroll
    for (int mt = 2; mt < MTW - 1; ++mt) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    }
  };
  Frags F;
  enter(0);
  load(0, 0, F);
  if constexpr (KS == 2) {
#pragma clang loop unroll(disable)
    for (int i = 0; i + 1 < nk; ++i) {
      step(F, i, 1, std::false_type{});
      step(F, i + 1, 0, std::true_type{});
    }
    step(F, nk - 1, 1, std::false_type{});
    mma(F);
  } else {
#pragma clang loop unroll(disable)
    for (int i = 0; i + 1 < nk; ++i) step(F, i + 1, 0, std::true_type{});
    mma(F);
  }

  // epilogue: SiLU, residual, one 16-byte pixel of plane (pair, g) per lane and pixel tile
  const unsigned plane0 = (unsigned)(cb * (MT * 2) + wm * (MTW * 2) + g);  // first output plane of this lane (8 channels per plane)
#pragma unroll
  for (int p = 0; p < MTW / 2; ++p) {
    bf16x8 rv[4];
    if constexpr (RES) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        rv[nt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                rres, po[nt] == OOB_LANE ? OOB_LANE : (plane0 + 4u * p) * a.res_plane + po[nt], 0, 0));
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[2 * p][nt][r];
        v[4 + r] = acc[2 * p + 1][nt][r];
      }
      if constexpr (ACT) {
        silu8(v);
      }
      if constexpr (RES) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += (float)rv[nt][r];
      }
      store_b128_nt(u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])}, rout,
                    po[nt] == OOB_LANE ? OOB_LANE : (plane0 + 4u * p) * a.out_plane + po[nt], a.st_nt);
      if constexpr (O8) {  // this lane's 8 channels = half a 16-byte pixel of an e4m3 plane (16 channels)
        const __amdgpu_buffer_rsrc_t r8 = __builtin_amdgcn_make_buffer_rsrc(a.out8, (short)0, a.out8_bytes, 0x00020000);
        const int c = (int)(plane0 + 4u * p) * 8 - a.out8_c0;
        const bool ok8 = po[nt] != OOB_LANE && c >= 0 && c < a.out8_n;
        const float is = a.out8_inv_scale;
        const unsigned v8 = ok8 ? (unsigned)(c >> 4) * a.out8_plane + po[nt] + (unsigned)(c & 15) : OOB_LANE;
        store_b64_nt(u32x2{pack_f8x4(v[0] * is, v[1] * is, v[2] * is, v[3] * is), pack_f8x4(v[4] * is, v[5] * is, v[6] * is, v[7] * is)}, r8, v8,
                     a.st_nt & 2);
      }
    }
  }
  tile = ntile;
  cur = nxt;
  gs0 = (gs0 + nk) % NS;
  }  // tiles
#ifdef YL_STAMPS
  if (a.stamps && lane == 0) {
    unsigned long long* p = a.stamps + ((size_t)blockIdx.x * NW + wave) * 8;
    p[0] = st_wait;
    p[1] = st_bar;
    p[2] = st_n;
    p[3] = __builtin_amdgcn_s_memtime() - st_t0;
    p[4] = (unsigned long long)round;
  }
#endif
}

// ------------------------------------------------------------------------------------ host
static unsigned gm_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull / (unsigned)d) + 1ull); }

// Takes over a conv_choose()n tiling when the layer is wide enough: sets cfg->gemm (tiles per wave) and the packing it needs.
// Only plain bf16 1x1 stride-1 convs (no fused follow-up, second input or fp32 output: the caller checks the views).
void conv_gemm_choose(int batch, int Cin, int Cout, int k, int stride, int Ho, int Wo, ConvConfig* c, int cin2) {
  c->gemm = 0;
  // (read at every engine build, not cached: the tests switch them)
  const int enable = env_int("YL_CONV_GEMM", 1);
  const int min_c = tune_int("YL_CONV_GEMM_MINC", 128);
  // smallest grid that takes it (384 before the workgroups were persistent; 192: YOLO11s batch 64 24.3 -> 25.0 k images/s --
  // its 20x20 maps have 200-400 tiles --, YOLO11n and YOLO11x unchanged; 96: the same)
  const int min_blocks = env_int("YL_CONV_GEMM_BLOCKS", 192);
  // 3x3 convs (stride 1 or 2; YL_CONV_GEMM3=0 turns them off): K = 9 x Cin, so 64 input channels are deep enough
  const int enable3 = env_int("YL_CONV_GEMM3", 1);
  const bool k3 = k == 3 && enable3 && (stride == 1 || stride == 2) && !cin2 && Cin >= 64;
  // 64 output channels (round 5, YL_CONV_GEMM_M64=0: off): 1x1 convs with at least 128 input channels on maps of at least
  // m64_px pixels per batch -- traffic-bound layers, where LDS-DMA keeps more bytes in flight than conv_kernel's register ring
  const int m64 = tune_int("YL_CONV_GEMM_M64", 1);
  const long long m64_px = env_int("YL_CONV_GEMM_M64_PX", 100000);
  const bool c64 = m64 && Cout == 64 && Cin % 32 == 0 && (long long)batch * Ho * Wo >= m64_px &&
                   ((k == 1 && stride == 1 && Cin >= min_c) || (m64 >= 2 && k3));
  if (!enable || !(k3 || (k == 1 && stride == 1 && Cin >= min_c)) || Cin % 32 || (Cout < min_c && !c64)) return;
  int MTW = 0;
  if (c64) MTW = 2;
  else if (Cout % 256 == 0) MTW = 8;
  else if (Cout % 192 == 0) MTW = 6;
  else if (Cout % 128 == 0) MTW = 4;
  else return;
  if ((enable == 4 || enable == 6) && Cout % (32 * enable) == 0) MTW = enable;  // development: force the channel block
  const Geo gg = geo_of(batch, Ho, Wo);
  const long long npos = (long long)gg.g0 + (long long)batch * gg.img;
  const int ncb = Cout / (32 * MTW);
  if ((long long)gg.img * gg.W1 >= (1LL << 32) || gg.img < 128) return;  // exactness of the in-kernel magic division by W1; one image wrap per tile
  // Eight waves (256 pixels per workgroup, one workgroup per CU): measured (tools/gemm_bench.sh) +3-10 % on 256-channel
  // blocks (512 -> 512 @20x20 x 64: 24.7 against 27.0 us; 768 -> 768 @160x160 x 8: 909 against 877 TFLOP/s), -4-8 % on
  // 192-channel blocks, equal on 128.  YL_CONV_GEMM_W8: smallest grid (in 256-pixel workgroups) that takes it; 0: never;
  // negative: every tile size from that grid up (development).
  const int w8_blocks = env_int("YL_CONV_GEMM_W8", 128);
  int NWN = 2;
  const long long b8 = (npos + 255) / 256 * ncb;
  if (w8_blocks > 0 && MTW == 8 && b8 >= w8_blocks) NWN = 4;
  if (w8_blocks < 0 && b8 >= -w8_blocks && (MTW != 6 || c->ksteps % 2 == 0) && MTW != 2) NWN = 4;  // (64-channel blocks: four waves only)
  const int BN = 64 * NWN;
  if (cin2) {  // upsampled second input: whole ring slices on either side of the split, one image wrap per pixel tile
    const bool even = c->ksteps % 2 == 0;
    const int ks_per_slice = NWN == 4 ? (MTW == 6 ? 2 : 1) : ((MTW == 8 || !even) ? 1 : 2);  // = gm_ks
    if (cin2 % 32 || (cin2 / 32) % ks_per_slice || gg.img < BN) return;
  }
  const long long ntiles = (npos + BN - 1) / BN;
  if (ntiles * ncb < min_blocks) return;                       // a grid that cannot fill 256 CUs: conv_kernel's finer tiles
  c->gemm = MTW;
  c->gemm_nwn = NWN;
  c->kord = k3 ? 2 : 0;  // 3x3: k-step = (block of 4 planes, tap), see conv_pack_weights
  c->uniform = 1;
  c->MT = 2 * MTW;
  c->NT = 4;
  c->ncb = ncb;
  c->kchunk = c->ksteps;
  c->post = 0;
  c->reg = c->cs = 0;
  c->ppw = 1;
  c->lds_bytes = conv_gemm_lds_bytes(MTW, NWN, c->ksteps % 2 == 0);
  c->wpk_elems = (size_t)ncb * c->ksteps * c->MT * 64 * 8;
  c->gws = (int)ntiles;
  c->name = k3 ? (stride == 2 ? "conv3x3s2_gemm_mfma" : "conv3x3_gemm_mfma") : "conv1x1_gemm_mfma";
}

template <int MTW, int NWN, bool EVEN, bool RES, bool ACT, bool O8 = false, bool UP = false, int GEO = 0>
static int launch_gemm(const ConvArgs& a, size_t lds, hipStream_t s) {
  if constexpr (GEO == 0 && !O8 && !UP) {
    if (a.ntaps == 9) {
      if (a.out8 || a.in2) return fail(-3, "conv_gemm: a 3x3 conv takes neither an e4m3 copy nor a second input");
      if constexpr (!RES) {  // (no stride-2 conv carries a residual)
        if (a.stride == 2) return launch_gemm<MTW, NWN, EVEN, RES, ACT, false, false, 2>(a, lds, s);
      }
      if (a.stride == 2) return fail(-3, "conv_gemm: a stride-2 conv with a residual");
      return launch_gemm<MTW, NWN, EVEN, RES, ACT, false, false, 1>(a, lds, s);
    }
  }
  if constexpr (!O8 && GEO == 0) {
    if (a.out8) return launch_gemm<MTW, NWN, EVEN, RES, ACT, true, UP>(a, lds, s);
  }
  if constexpr (!UP && !RES && GEO == 0) {  // (the convs behind Upsample + Concat carry no residual)
    if (a.in2) return launch_gemm<MTW, NWN, EVEN, RES, ACT, O8, true>(a, lds, s);
  }
  if (a.in2 && !UP) return fail(-3, "conv_gemm: an upsampled second input cannot be combined with a residual");
  auto kern = conv_gemm_kernel<MTW, NWN, gm_ks(MTW, NWN, EVEN), gm_ns(MTW, NWN, EVEN), RES, ACT, O8, UP, GEO>;
  if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), (int)(160 * 1024))) return rc;
  if (a.ksteps / gm_ks(MTW, NWN, EVEN) < gm_ns(MTW, NWN, EVEN) - 1) return fail(-3, "conv_gemm: too few K-slices for the ring");
  // persistent: as many workgroups as stay resident (every CU x two 4-wave or one 8-wave workgroup), each walks its tiles
  static const int persist = tune_int("YL_CONV_GEMM_PERSIST", 1);
  static const int m64_wpc = tune_int("YL_CONV_GEMM_M64_WPC", 2);  // development: workgroups per CU of the 64-channel blocks
  const int resident = device_cu_count() * (MTW == 2 ? m64_wpc : (NWN == 2 && lds <= 80 * 1024 ? 2 : 1));
  const int grid = persist && a.nblocks > resident ? resident : a.nblocks;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(128 * NWN), lds, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

int conv_gemm_launch(const ConvConfig& cfg, ConvArgs a, int batch, hipStream_t s) {
  const int MTW = cfg.gemm;
  const bool k3 = cfg.ntaps == 9;
  if ((MTW != 2 && MTW != 4 && MTW != 6 && MTW != 8) || (cfg.ntaps != 1 && !k3) || (a.stride != 1 && !(k3 && a.stride == 2)) || a.out_f32 ||
      a.wpk2 || !cfg.uniform || (k3 && cfg.kord != 2) || a.out_w1 != a.Wo + 1 || a.Cout != cfg.ncb * 32 * MTW ||
      (a.stride == 1 && (a.in_w1 != a.out_w1 || a.in_img != a.out_img || a.in_g0 != a.out_g0)))
    return fail(-3, "conv_gemm: needs a plain bf16 1x1 or 3x3 conv, Cin % 32 == 0, Cout a multiple of the channel block");
  a.ntaps = cfg.ntaps;
  a.ncb = cfg.ncb;
  a.cg = cfg.cg;
  a.ksteps = cfg.ksteps;
  a.t_B = batch;
  a.t_H = a.Ho;
  a.t_W = a.Wo;
  a.magic_w1 = gm_magic(a.out_w1);  // (tiles run over OUTPUT positions)
  const long long lim = 1LL << 30;
  const long long inb = (long long)(cfg.cg - (a.in2 ? 4 * a.ks_split : 0)) * a.in_plane, outb = (long long)(a.Cout / 8) * a.out_plane;
  if (a.in2) {
    const long long in2b = (long long)(4 * a.ks_split) * a.in2_plane;
    if (in2b > (1LL << 30)) return fail(-3, "conv_gemm: a tensor slice exceeds 1 GiB (32-bit buffer offsets); lower the batch size");
    a.in2_bytes = (unsigned)in2b;
  }
  const long long resb = a.res ? (long long)(a.Cout / 8) * a.res_plane : 0;
  if (inb > lim || outb > lim || resb > lim || (long long)cfg.ncb * cfg.ksteps * cfg.MT * 1024 > lim)
    return fail(-3, "conv_gemm: a tensor slice exceeds 1 GiB (32-bit buffer offsets); lower the batch size");
  a.in_bytes = (unsigned)inb;
  a.out_bytes = a.out8 && a.out8_only ? 0u : (unsigned)outb;  // (copy only: a descriptor of zero bytes drops every bf16 store)
  a.res_bytes = (unsigned)resb;
  a.nblocks = cfg.gws * cfg.ncb;
  a.dbg = 0;
  a.stamps = g_stamp_buf;
  a.st_nt = st_nt_policy(outb);
  const bool even = cfg.ksteps % 2 == 0;
  const int NWN = cfg.gemm_nwn;
  if ((NWN != 2 && NWN != 4) || (NWN == 4 && MTW == 6 && !even)) return fail(-3, "conv_gemm: unsupported wave grid");
  const size_t lds = conv_gemm_lds_bytes(MTW, NWN, even);
#define YL_GM3(M, W, E) (a.res ? (a.act ? launch_gemm<M, W, E, true, true>(a, lds, s) : launch_gemm<M, W, E, true, false>(a, lds, s)) \
                            : (a.act ? launch_gemm<M, W, E, false, true>(a, lds, s) : launch_gemm<M, W, E, false, false>(a, lds, s)))
  if (NWN == 4) return MTW == 4 ? YL_GM3(4, 4, true) : MTW == 6 ? YL_GM3(6, 4, true) : YL_GM3(8, 4, true);
  if (MTW == 8) return YL_GM3(8, 2, true);
  if (MTW == 2) return even ? YL_GM3(2, 2, true) : YL_GM3(2, 2, false);
  if (MTW == 4) return even ? YL_GM3(4, 2, true) : YL_GM3(4, 2, false);
  return even ? YL_GM3(6, 2, true) : YL_GM3(6, 2, false);
#undef YL_GM3
}

}  // namespace yl
