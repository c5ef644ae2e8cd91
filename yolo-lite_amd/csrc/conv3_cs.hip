// 3x3 stride-1 Conv 64 -> 64 + (folded BN) bias + SiLU [+ residual] with the filter in registers and the OUTPUT CHANNELS
// SPLIT between wave pairs ("Cout split", round 3; reference nn/modules/conv.py:35-53, nn/modules/head.py:29-57, block.py:341-343).
//
// conv3_reg_kernel keeps the whole 64 -> 64 filter (288 registers) in ONE wave per SIMD; its SiLU epilogue (quarter-rate
// v_exp / v_rcp) then has no partner wave to hide behind, and 7.7 of its ~30 us are that epilogue (NOTEBOOK.md section 4).
// Here a wave owns 32 of the 64 output channels: 18 k-steps x 2 tiles x 4 registers = 144 VGPRs of filter, 32 of
// accumulators -- under 256 registers, so TWO workgroups (4 waves each) share a CU, plain MFMA builtins (no AGPR operands,
// no inline asm), and one workgroup's epilogue and LDS-DMA wait run under the other's MFMAs.  No partial sums, no exchange
// (conv3_pair_kernel split K and paid for one).  Waves (h, c) of a workgroup: h = channel half, c = 64-position column of the
// 128-position tile; both halves read the same activation fragments from LDS.
//
// Same operands, same k order (tap-major, 32 channels per k-step), same accumulation chain from the bias and the same
// epilogue as conv_kernel: bit-identical results.
//
// Input staging as conv3_reg / conv3_tile: in the guarded planar layout a 3x3 tap is a constant offset of the linear
// guarded index, so a tile needs ONE contiguous run of L + 2 (W + 1) + 2 positions of each of the 8 input planes, copied by
// LDS-DMA in 1 KiB pieces.  Persistent workgroups (two per CU) load their filter half once and walk tiles t = wg, wg + G, ..:
//   wait for the tile's pieces + barrier -> 144 MFMAs per wave -> barrier (everybody has left the buffer) -> the NEXT tile's
//   pieces are issued -> epilogue of this tile (its stores and the DMA's flight overlap).
#include "conv_common.h"

#ifndef YL_CS_DPP
#define YL_CS_DPP 0  // build switch (round 5 experiment): 1 = column taps by DPP shifts of ONE row fragment, 0 = every tap its own LDS read.
                     // MEASURED SLOWER: 64 -> 64 @80x80 batch 32 34.1 us against 27.5 us (12.5 against 10.5 us @40x40), bit-identical --
                     // the k-loop is LDS-bound (0.5 fragment reads per MFMA), and 384 DPP moves per 144 MFMAs in the same wave cost
                     // more issue time than the 42 LDS reads they replace: vector work between a wave's own MFMAs does not hide
                     // behind them here (the same finding as conv3_reg's interleaved epilogue, NOTEBOOK section 4)
#endif
namespace yl {

typedef __attribute__((address_space(3))) void lds_void_c;

// A B fragment (16 positions x 4 k-groups: lane = (position n, k-group g), one 16-lane row per k-group) shifted by S positions:
// lane n takes position n + S of `lo` (row_shl:S), the last S lanes of a row the first S positions of `hi`, the next 16
// positions (row_shr:16-S).  Two DPP moves per register instead of an LDS read: in the guarded layout the three column taps
// of a row are ONE run of positions, and the k-loop of this kernel is bound by the LDS array (two 32-channel waves read every
// fragment once each: 0.5 reads per MFMA), not by the vector ALU, which idles while it runs.
template <int S>
__device__ __forceinline__ bf16x8 frag_shift(const bf16x8& lo, const bf16x8& hi) {
  static_assert(S == 1 || S == 2, "column taps");
  const u32x4 a = __builtin_bit_cast(u32x4, lo), b = __builtin_bit_cast(u32x4, hi);
  u32x4 r;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    // (bound_ctrl off: a lane whose source lies outside its row keeps `old`)
    const int edge = __builtin_amdgcn_update_dpp(0, (int)b[q], 0x110 + (16 - S), 0xf, 0xf, false);   // row_shr:16-S
    r[q] = (unsigned)__builtin_amdgcn_update_dpp(edge, (int)a[q], 0x100 + S, 0xf, 0xf, false);        // row_shl:S
  }
  return __builtin_bit_cast(bf16x8, r);
}

constexpr int CS_L = 128;  // positions per tile (two 64-position columns)

static __host__ __device__ inline int cs_PL(int W) { return (CS_L + 2 * (W + 1) + 2 + 63) & ~63; }  // staged positions per plane

template <bool RES, bool ACT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv3_cs_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char csm[];
  constexpr int KS = 18, NT = 4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = wave >> 1, c = wave & 1;  // channel half, column of the tile
  const int g = lane >> 4, n = lane & 15;
  const int G = gridDim.x;
  const int wg = xcd_remap(blockIdx.x, G);
  const int PL = a.t_PL, PLB = PL * 16;
  const int ppp = PL >> 6;                       // 1 KiB pieces per plane
  const int nmine = (a.t_ntiles - wg + G - 1) / G;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, (short)0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, a.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.res, (short)0, RES ? a.res_bytes : 0, 0x00020000);
  const unsigned lane16 = (unsigned)lane * 16u;

  // tile t -> LDS: wave w copies planes w and w + 4 (ppp pieces each); a run past the slice reads zeros
  auto stage = [&](int t) __attribute__((always_inline)) {
    const int src0 = (a.in_g0 + t * CS_L - a.in_w1 - 1) * 16;  // first staged guarded position (>= 0)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int plane = wave + 4 * q;
      for (int j = 0; j < ppp; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void_c*)(csm + plane * PLB + j * 1024), 16,
                                                 (int)(lane16 + (unsigned)(plane * (int)a.in_plane + src0 + j * 1024)), 0, 0, 0);
    }
  };
  if (nmine > 0 && !(a.dbg & 1)) stage(wg);  // (a.dbg: timing ablations, YL_CONV_DBG -- wrong results)

  // this wave's half of the filter, conv_pack_weights order (4 tiles per k-step): tiles 2h, 2h + 1
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
  const unsigned char* const tb = csm + (g * PL + c * 64 + n) * 16;  // the lane's position 0 of plane g (taps add offsets)
  const int row_bytes = a.in_w1 * 16, half_bytes = 4 * PLB;           // planes 4..7 = input channels 32..63
  const unsigned oplane = (unsigned)(4 * h + g);                      // the lane's output plane (8 channels)

  for (int i = 0; i < nmine; ++i) {
    const int t = wg + i * G;
    // my pieces of this tile have landed: they are OLDER than the previous tile's NT epilogue stores (issued behind them), so a
    // counted wait covers them and leaves the stores in flight (vmcnt(0) stood still for the stores' round trip once per tile)
#ifndef YL_CS_WAIT_ALL
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NT) : "memory");
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    __builtin_amdgcn_s_barrier();
    f32x4 acc[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = bias_r[mt];
#if YL_CS_DPP
    // Two passes over two 16-position tiles each (the filter sits in registers: a pass costs no operand traffic).  Per row of taps
    // and channel half THREE fragments are read -- the run of 48 positions that starts one position left of the tiles -- and the
    // column taps dx = 1, 2 are its shifts: 6 LDS reads and 64 DPP moves per 24 MFMAs (0.25 reads per MFMA; 0.5 before).  The
    // MFMAs keep conv_kernel's k order (tap-major, channel half inside): bit-identical.
    if (!(a.dbg & 4))
#pragma unroll
    for (int hp = 0; hp < NT / 2; ++hp)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        bf16x8 F[2][3];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
          for (int j = 0; j < 3; ++j)
            F[ch][j] = *reinterpret_cast<const bf16x8*>(tb + ch * half_bytes + dy * row_bytes + (2 * hp + j) * 256);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int ch = 0; ch < 2; ++ch) {
            const int ks = 2 * (3 * dy + dx) + ch;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const bf16x8 xf = dx == 0 ? F[ch][u] : (dx == 1 ? frag_shift<1>(F[ch][u], F[ch][u + 1]) : frag_shift<2>(F[ch][u], F[ch][u + 1]));
#pragma unroll
              for (int mt = 0; mt < 2; ++mt)
                acc[mt][2 * hp + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][mt], xf, acc[mt][2 * hp + u], 0, 0, 0);
            }
          }
      }
#else
    if (!(a.dbg & 4))
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int tap = ks >> 1, ch = ks & 1, dy = tap / 3, dx = tap % 3;
      const unsigned char* rb = tb + ch * half_bytes + dy * row_bytes + dx * 16;
      bf16x8 xb[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) xb[nt] = *reinterpret_cast<const bf16x8*>(rb + nt * 256);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][mt], xb[nt], acc[mt][nt], 0, 0, 0);
    }
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // everybody has read the tile: the buffer may be refilled
    if (i + 1 < nmine && !(a.dbg & 1)) stage(t + G);
    if (a.dbg & 2) continue;
    // epilogue: valid positions only (guards are computed, never stored)
    const int r0 = t * CS_L + c * 64;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int r = r0 + nt * 16 + n;  // position relative to g0
      const int b = (int)__umulhi((unsigned)r, a.magic_img);
      const int rr = r - b * a.in_img;
      const int y = (int)__umulhi((unsigned)rr, a.magic_w1);
      const int x = rr - y * a.in_w1;
      const unsigned po = (b < a.t_B && y < a.t_H && x < a.t_W) ? (unsigned)(a.in_g0 + r) * 16u : OOB_LANE;
      float v[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] = acc[0][nt][q];
        v[4 + q] = acc[1][nt][q];
      }
      if constexpr (ACT) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = silu_f(v[q]);
      }
      if constexpr (RES) {
        const bf16x8 rv = __builtin_bit_cast(
            bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rres, po == OOB_LANE ? OOB_LANE : oplane * a.res_plane + po, 0, 0));
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += (float)rv[q];
      }
      store_b128_nt(u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])},
                    rout, po == OOB_LANE ? OOB_LANE : oplane * a.out_plane + po, a.st_nt);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The tile body shared by the kernels below.  (Round 5 also built a SEGMENT form -- one contiguous run per workgroup, staged at once,
// no barrier between tiles -- and an eight-wave form with the filter through LDS: 25.6 and 24.5 us in the network against 23.5 us for
// the tile walk with a balanced first round; both were deleted in round 6, NOTEBOOK.md section 15 has their measurements.)
struct CsNoMid {
  __device__ __forceinline__ void operator()() const {}
};
// One tile of NT fragments per wave: the k-loop, `mid()` (the tile-walking kernels free the LDS buffer and request the next tile
// there), the epilogue.
// POST (round 6): this conv is Detect's cv2[i][1] and the 1x1 conv cv2[i][2] + the box decode follow in the same launch (reference
// nn/modules/head.py:36-40, 95-126): what conv_kernel's POST + DFL instantiation does for the generic tiling, on this kernel's.  A wave
// holds 32 of the 64 channels of its 16 * NT positions, the 1x1 needs all 64: after SiLU the waves put their bf16 halves into the
// tile's t2 buffer in LDS (8 planes x 128 positions x 16 bytes: B-fragment order), a barrier, then the two waves of a column split
// its fragments (h = 0 takes the even ones, h = 1 the odd ones): two B fragments from t2, the 1x1 filter's eight A fragments from
// LDS (8 KiB, staged once per workgroup), eight MFMAs from the bias -- conv_body's POST order -- and dfl_box on the fp32
// accumulators; lane g = 0 stores the anchor's (cx, cy, w, h), the bf16 logits go to `out` only with a.dfl_keep.  Bit-identical
// to the conv_kernel form.
struct CsPost {
  unsigned char* t2;          // LDS: the tile's t2, [8 planes][128 positions][16 bytes]
  const unsigned char* w2;    // LDS: the 1x1 filter, conv_pack_weights order [2 k-steps][4 tiles][64 lanes][16 bytes]
  const float* bias2;         // global: its bias, padded to 64
  int h, c, pos0;             // channel half, column; the tile's first position (relative to g0)
};

template <int NT, bool RES, bool ACT, typename Mid = CsNoMid, bool POST = false>
__device__ __forceinline__ void cs2_tile(const ConvArgs& a, const bf16x8 (&wf)[18][2], const f32x4 (&bias_r)[2],
                                         const unsigned char* tb, int r0, int n, unsigned oplane, int row_bytes, int half_bytes,
                                         const __amdgpu_buffer_rsrc_t& rout, const __amdgpu_buffer_rsrc_t& rres, int wave, int lane,
                                         int stamp_id, Mid mid = Mid(), const CsPost* post = nullptr) {
  f32x4 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = bias_r[mt];
  if (!(a.dbg & 4))
#pragma unroll
  for (int ks = 0; ks < 18; ++ks) {
    const int tap = ks >> 1, ch = ks & 1, dy = tap / 3, dx = tap % 3;
    const unsigned char* rb = tb + ch * half_bytes + dy * row_bytes + dx * 16;
    bf16x8 xb[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) xb[nt] = *reinterpret_cast<const bf16x8*>(rb + nt * 256);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][mt], xb[nt], acc[mt][nt], 0, 0, 0);
  }
  YL_STAMP(a.stamps, (int)(blockDim.x >> 6), wave, lane, stamp_id);
  mid();
  if (a.dbg & 2) return;
  if constexpr (POST) {
    static_assert(!POST || (ACT && !RES), "the box branch's second conv: SiLU, no residual");
    const int g = lane >> 4;
    // 1. SiLU + bf16: the lane's 8 channels (plane oplane) of its position -> t2
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[q] = acc[0][nt][q];
        v[4 + q] = acc[1][nt][q];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = silu_f(v[q]);
      *reinterpret_cast<u32x4*>(post->t2 + ((int)oplane * 128 + post->c * 16 * NT + nt * 16 + n) * 16) =
          u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the tile's t2 is complete (the next tile writes it behind two more barriers)
    // 2. the 1x1 conv + the box decode of this wave's fragments of the column
    const __amdgpu_buffer_rsrc_t rout2 = rout;
#pragma unroll
    for (int f = 0; f < NT; ++f) {
      if ((f & 1) != post->h) continue;  // (wave-uniform)
      const int pt = post->c * 16 * NT + f * 16 + n;  // position inside the tile
      f32x4 t[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) t[mt] = *reinterpret_cast<const f32x4*>(post->bias2 + chan_in_block(4, mt, g, 0));
#pragma unroll
      for (int p2 = 0; p2 < 2; ++p2) {
        const bf16x8 xb2 = *reinterpret_cast<const bf16x8*>(post->t2 + ((4 * p2 + g) * 128 + pt) * 16);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const bf16x8 w2 = *reinterpret_cast<const bf16x8*>(post->w2 + ((p2 * 4 + mt) * 64 + lane) * 16);
          t[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, xb2, t[mt], 0, 0, 0);
        }
      }
      const int r = post->pos0 + pt;  // position relative to g0
      const int b = (int)__umulhi((unsigned)r, a.magic_img);
      const int rr = r - b * a.in_img;
      const int y = (int)__umulhi((unsigned)rr, a.magic_w1);
      const int x = rr - y * a.in_w1;
      const bool ok = b < a.t_B && y < a.t_H && x < a.t_W;
      const float4 bx = dfl_box(t, g, (float)x + 0.5f, (float)y + 0.5f, a.dfl_stride);
      if (g == 0 && ok) *reinterpret_cast<float4*>(a.dfl_out + 4LL * ((long long)(b * a.t_H + y) * a.t_W + x)) = bx;
      if (a.dfl_keep) {  // the bf16 logits too (parity tests): the lane's 8 consecutive channels of tile pair p = one pixel of plane 4 p + g
        const unsigned po = ok ? (unsigned)(a.in_g0 + r) * 16u : OOB_LANE;
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2)
          store_b128_nt(u32x4{pack_bf16x2(t[2 * p2][0], t[2 * p2][1]), pack_bf16x2(t[2 * p2][2], t[2 * p2][3]),
                              pack_bf16x2(t[2 * p2 + 1][0], t[2 * p2 + 1][1]), pack_bf16x2(t[2 * p2 + 1][2], t[2 * p2 + 1][3])},
                        rout2, po == OOB_LANE ? OOB_LANE : (unsigned)(4 * p2 + g) * a.out_plane + po, a.st_nt);
      }
    }
    return;
  }
  // (Round 5, measured and NOT kept: the (row, column) of fragments 1.. by carrying from fragment 0's instead of two magic
  // divisions each, and silu8's packed multiplies / adds -- 100 of ~445 issue cycles per fragment fewer, and 28.9-29.3 us against
  // 27.0-27.5 us stand-alone (each change alone ~+1 us): the epilogue is bound by its dependency chains, not by issue; the
  // independent per-fragment form gives the scheduler more to interleave)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int r = r0 + nt * 16 + n;  // position relative to g0
    const int b = (int)__umulhi((unsigned)r, a.magic_img);
    const int rr = r - b * a.in_img;
    const int y = (int)__umulhi((unsigned)rr, a.magic_w1);
    const int x = rr - y * a.in_w1;
    const unsigned po = (b < a.t_B && y < a.t_H && x < a.t_W) ? (unsigned)(a.in_g0 + r) * 16u : OOB_LANE;
    float v[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[q] = acc[0][nt][q];
      v[4 + q] = acc[1][nt][q];
    }
    if constexpr (ACT) {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = silu_f(v[q]);
    }
    if constexpr (RES) {
      const bf16x8 rv = __builtin_bit_cast(
          bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rres, po == OOB_LANE ? OOB_LANE : oplane * a.res_plane + po, 0, 0));
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] += (float)rv[q];
    }
    store_b128_nt(u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])},
                  rout, po == OOB_LANE ? OOB_LANE : oplane * a.out_plane + po, a.st_nt);
  }
}

// The tile-walking form with a BALANCED last round (round 5).  conv3_cs_kernel gives workgroup wg the tiles wg, wg + G, ..: 1640
// tiles of 128 positions on 512 workgroups are three rounds and a fourth for 104 workgroups -- the launch lasts four tile periods.
// Here the positions that do not fill whole rounds (13 344 of 209 952) are spread over ALL workgroups as one SHORT tile each (32
// positions = one fragment per wave, 417 workgroups), walked FIRST (the launch's first MFMA needs 196 staged positions instead of
// 292), then the whole rounds of 128-position tiles: 13 fragments per wave on the longest path instead of 16.  Buffer, staging,
// counted wait and barriers as conv3_cs_kernel; tiles by cs2_tile: bit-identical.
// (block / G: this workgroup's index among the G workgroups that walk `a`'s tiles -- the whole grid, or one member's share of a
// group launch, conv3_csg_kernel)
template <bool RES, bool ACT, bool POST = false>
__device__ __forceinline__ void cs1b_body(const ConvArgs& a, int block, int G) {
  extern __shared__ __attribute__((aligned(16))) unsigned char csm[];
  constexpr int KS = 18;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = wave >> 1, c = wave & 1;  // channel half, column of the tile
  const int g = lane >> 4, n = lane & 15;
  const int wg = xcd_remap(block, G);
  const int PL = a.t_PL, PLB = PL * 16;
  const int Lr = a.t_px, nsmall = a.t_ppp, S = nsmall * Lr;  // the short round: tiles of Lr positions, nsmall of them, S positions in all
  const int has_small = wg < nsmall ? 1 : 0;
  const int nfull = a.t_ntiles > wg ? (a.t_ntiles - wg + G - 1) / G : 0;  // this workgroup's 128-position tiles
  const int ntl = has_small + nfull;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, (short)0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, a.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.res, (short)0, RES ? a.res_bytes : 0, 0x00020000);
  const unsigned lane16 = (unsigned)lane * 16u;
  // tile i of this workgroup: first position (relative to g0) and fragments per wave
  auto tile_of = [&](int i, int& nt) __attribute__((always_inline)) -> int {
    if (has_small) {
      if (i == 0) {
        nt = Lr >> 5;
        return wg * Lr;
      }
      --i;
    }
    nt = 4;
    return S + (wg + i * G) * CS_L;
  };
  // a tile -> LDS: wave w copies planes w and w + 4; a run past the slice reads zeros
  auto stage = [&](int pos, int nt) __attribute__((always_inline)) {
    const int src0 = (a.in_g0 + pos - a.in_w1 - 1) * 16;  // first staged guarded position (>= 0)
    const int np = (32 * nt + 2 * a.in_w1 + 2 + 63) >> 6;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int plane = wave + 4 * q;
      for (int j = 0; j < np; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void_c*)(csm + plane * PLB + j * 1024), 16,
                                                 (int)(lane16 + (unsigned)(plane * (int)a.in_plane + src0 + j * 1024)), 0, 0, 0);
    }
  };
  int nt_cur = 4, pos_cur = 0;
  if (ntl > 0) {
    pos_cur = tile_of(0, nt_cur);
    if (!(a.dbg & 1)) stage(pos_cur, nt_cur);
  }
  // POST: [input tile: 8 planes][t2: 16 KiB][the 1x1 filter: 8 KiB, copied once -- the first tile's barrier publishes it]
  unsigned char* const t2buf = csm + 8 * PLB;
  unsigned char* const w2buf = t2buf + 8 * 128 * 16;
  if constexpr (POST) {
    const u32x4* src = reinterpret_cast<const u32x4*>(a.wpk2);
    for (int i = tid; i < 2 * 4 * 64; i += 256) reinterpret_cast<u32x4*>(w2buf)[i] = src[i];
  }
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
  const int row_bytes = a.in_w1 * 16, half_bytes = 4 * PLB;  // planes 4..7 = input channels 32..63
  const unsigned oplane = (unsigned)(4 * h + g);              // the lane's output plane (8 channels)
  int nt_prev = 0;
  for (int i = 0; i < ntl; ++i) {
    // my pieces of this tile are OLDER than the previous tile's epilogue stores (nt_prev of them; with a residual its loads too, which
    // only makes the count conservative): a counted wait covers the pieces and leaves the stores in flight
    if constexpr (POST) {
      // (the previous tile's stores: one per fragment this wave decoded -- ceil or floor of nt_prev / 2 -- when the logits are not kept)
      const int st_prev = a.dfl_keep ? 0 : ((nt_prev + 1 - h) >> 1);
      switch (st_prev) {
        case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
      }
    } else {
    switch (nt_prev) {
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    }
    __builtin_amdgcn_s_barrier();
    int nt_next = 4, pos_next = 0;
    const bool more = i + 1 < ntl;
    if (more) pos_next = tile_of(i + 1, nt_next);
    auto mid = [&]() __attribute__((always_inline)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // everybody has read the tile: the buffer may be refilled
      if (more && !(a.dbg & 1)) stage(pos_next, nt_next);
    };
    const unsigned char* const tb = csm + (g * PL + c * 16 * nt_cur + n) * 16;  // the lane's position 0 of plane g (taps add offsets)
    const int r0 = pos_cur + c * 16 * nt_cur;
    const CsPost post{t2buf, w2buf, a.bias2, h, c, pos_cur};
    switch (nt_cur) {
      case 4: cs2_tile<4, RES, ACT, decltype(mid), POST>(a, wf, bias_r, tb, r0, n, oplane, row_bytes, half_bytes, rout, rres, wave, lane, 4, mid, &post); break;
      case 3: cs2_tile<3, RES, ACT, decltype(mid), POST>(a, wf, bias_r, tb, r0, n, oplane, row_bytes, half_bytes, rout, rres, wave, lane, 4, mid, &post); break;
      case 2: cs2_tile<2, RES, ACT, decltype(mid), POST>(a, wf, bias_r, tb, r0, n, oplane, row_bytes, half_bytes, rout, rres, wave, lane, 4, mid, &post); break;
      default: cs2_tile<1, RES, ACT, decltype(mid), POST>(a, wf, bias_r, tb, r0, n, oplane, row_bytes, half_bytes, rout, rres, wave, lane, 4, mid, &post); break;
    }
    nt_prev = (a.dbg & 2) ? 0 : nt_cur;
    nt_cur = nt_next;
    pos_cur = pos_next;
  }
}

template <bool RES, bool ACT, bool POST = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv3_cs1b_kernel(ConvArgs a) {
  cs1b_body<RES, ACT, POST>(a, blockIdx.x, gridDim.x);
}

// The three Detect levels' cv2[i][1] + cv2[i][2] + decode in ONE launch (round 6; reference head.py:59-62 `for i in range(self.nl)`):
// the resident workgroups are dealt to the levels in proportion to their tiles (a workgroup's filter half lives in its registers:
// it stays with one level), each share walks its level's tiles as conv3_cs1b_kernel does.  On their own the 40x40 / 20x20 copies
// are latency-bound launches (29 us for 5.2 GFLOP as a pair on conv_group_kernel); here they fill the gaps of level 0's walk.
struct CsGroup {
  ConvArgs m[3];
  int nwg[3];  // workgroups of each member (unused members: 0)
};
template <bool POST>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv3_csg_kernel(CsGroup G) {
  int block = blockIdx.x, mi = 0;
  if (block >= G.nwg[0]) {
    block -= G.nwg[0];
    mi = 1;
    if (block >= G.nwg[1]) {
      block -= G.nwg[1];
      mi = 2;
    }
  }
  cs1b_body<false, true, POST>(G.m[mi], block, G.nwg[mi]);
}

// ------------------------------------------------------------------------------------ host
static unsigned cs_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull / (unsigned)d) + 1ull); }

// sets cfg->cs for 64 -> 64 3x3 stride-1 convs with enough tiles for two persistent workgroups per CU
void conv3_cs_choose(int batch, int Cin, int Cout, int k, int stride, int Ho, int Wo, int post, ConvConfig* c) {
  c->cs = 0;
  const int enable = env_int("YL_CONV_CS", 1);
  // (512 tiles = one per resident workgroup: YOLO11s's 64 -> 64 Bottleneck convs @40x40 batch 64 have 840, +0.5-1.1 % end to end
  // against conv3_reg_kernel; YOLO11n batch 32 has none between 512 and 1024)
  const int min_tiles = env_int("YL_CONV_CS_MIN", 512);
  if (!enable || k != 3 || stride != 1 || post || Cin != 64 || Cout != 64 || c->ncb != 1 || c->MT != 4) return;
  const long long img = (long long)(Ho + 1) * (Wo + 1), nq = (long long)batch * img;
  if (nq * img >= (1LL << 32) || img * (Wo + 1) >= (1LL << 32)) return;  // exactness of the magic divisions
  if ((nq + CS_L - 1) / CS_L < min_tiles || (size_t)8 * cs_PL(Wo) * 16 > 80 * 1024) return;
  c->cs = 1;
  c->reg = 0;
  c->name = "conv3x3_cs_mfma";
}

template <bool RES, bool ACT>
static int launch_cs(const ConvArgs& a, int grid, size_t lds, hipStream_t s) {
  auto kern = conv3_cs_kernel<RES, ACT>;
  if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), (int)(160 * 1024))) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

template <bool RES, bool ACT, bool POST = false>
static int launch_cs1b(const ConvArgs& a, int grid, size_t lds, hipStream_t s) {
  auto kern = conv3_cs1b_kernel<RES, ACT, POST>;
  if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), (int)(160 * 1024))) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

// cfg->cs = 2: Detect's cv2[i][1] (64 -> 64 3x3) with cv2[i][2] (1x1, 64 -> 64 DFL logits) and the box decode behind it in the same
// launch (conv3_cs1b_kernel's POST form) -- called by the engine for an op with post_weight and dfl_level that runs on its own
bool conv3_cs_choose_post(int batch, int Cin, int Cout1, int Cout2, int k, int stride, int Ho, int Wo, ConvConfig* c) {
  const int enable = env_int("YL_CONV_CS", 1);
  const int min_tiles = env_int("YL_CONV_CS_MIN", 512);
  if (!enable || k != 3 || stride != 1 || Cin != 64 || Cout1 != 64 || Cout2 != 64 || c->ncb != 1 || c->MT != 4) return false;
  const long long img = (long long)(Ho + 1) * (Wo + 1), nq = (long long)batch * img;
  if (nq * img >= (1LL << 32) || img * (Wo + 1) >= (1LL << 32)) return false;
  if ((nq + CS_L - 1) / CS_L < min_tiles || (size_t)8 * cs_PL(Wo) * 16 + 24 * 1024 > 80 * 1024) return false;
  c->cs = 2;
  c->reg = 0;
  c->kord = 0;
  c->name = "conv3x3_cs+1x1+dfl_mfma";
  return true;
}

// Fills the tile-walk parameters of `a` for `resident` workgroups (geometry, magic numbers, the balanced short first round where the
// positions do not fill whole rounds); *grid = workgroups to launch, *balanced = 0 when conv3_cs_kernel's plain walk applies (the
// POST form runs conv3_cs1b_kernel either way, with no short round then).
static int cs_prepare(ConvArgs& a, int batch, int resident, bool post, size_t* lds, int* grid, int* balanced) {
  if (a.stride != 1 || a.out_f32 || a.in2 || (a.wpk2 != nullptr) != post || a.out8 || a.in_w1 != a.out_w1 || a.in_img != a.out_img ||
      a.in_g0 != a.out_g0 || a.in_w1 != a.Wo + 1 || a.Cin != 64 || a.Cout != 64)
    return fail(-3, "conv3_cs: needs a plain 64 -> 64 3x3 stride-1 conv between planar tensors of one geometry");
  if (post && (!a.dfl_out || !a.bias2 || a.post_cout != 64 || a.post_act || !a.act || a.res))
    return fail(-3, "conv3_cs: the fused 1x1 form is the box branch's cv2[i][1] + cv2[i][2] + decode (SiLU, no residual, DFL epilogue)");
  a.t_B = batch;
  a.t_H = a.Ho;
  a.t_W = a.Wo;
  a.t_PL = cs_PL(a.Wo);
  a.magic_img = cs_magic(a.in_img);
  a.magic_w1 = cs_magic(a.in_w1);
  const long long nq = (long long)batch * a.in_img;
  a.t_ntiles = (int)((nq + CS_L - 1) / CS_L);
  const long long lim = 1LL << 30;
  const long long inb = 8LL * a.in_plane, outb = 8LL * a.out_plane, resb = a.res ? 8LL * a.res_plane : 0;
  if (inb > lim || outb > lim || resb > lim) return fail(-3, "conv3_cs: a tensor slice exceeds 1 GiB; lower the batch size");
  a.in_bytes = (unsigned)inb;
  a.out_bytes = (unsigned)outb;
  a.res_bytes = (unsigned)resb;
  a.dbg = tune_int("YL_CONV_DBG", 0);
  a.stamps = g_stamp_buf;
  *lds = (size_t)8 * a.t_PL * 16 + (post ? 24 * 1024 : 0);  // (+ the tile's t2 and the 1x1 filter)
  // tile-walking form with a balanced (short, first) round
  const int bal = tune_int("YL_CS_BAL", 1);
  const long long nq32 = (nq + 31) & ~31LL;
  const long long per_round = (long long)resident * CS_L;
  const long long rounds = nq32 / per_round, rem = nq32 - rounds * per_round;
  const int Lr = rem > 0 ? (int)(((rem + resident - 1) / resident + 31) & ~31LL) : 0;
  if (bal && Lr > 0 && Lr < CS_L) {
    const int nsmall = (int)((rem + Lr - 1) / Lr);
    const long long S = (long long)nsmall * Lr;
    const int nfull_tiles = nq32 > S ? (int)((nq32 - S + CS_L - 1) / CS_L) : 0;
    a.t_px = Lr;
    a.t_ppp = nsmall;
    a.t_ntiles = nfull_tiles;
    int gridb = nsmall > nfull_tiles ? nsmall : nfull_tiles;
    *grid = gridb > resident ? resident : gridb;
    *balanced = 1;
    return 0;
  }
  *grid = a.t_ntiles < resident ? a.t_ntiles : resident;
  *balanced = 0;
  a.t_px = 32;  // (conv3_cs1b_kernel without a short round)
  a.t_ppp = 0;
  return 0;
}

int conv3_cs_launch(const ConvConfig& cfg, ConvArgs a, int batch, hipStream_t s) {
  const bool post = cfg.cs == 2;
  size_t lds = 0;
  int grid = 0, balanced = 0;
  if (int rc = cs_prepare(a, batch, 2 * device_cu_count() /* two persistent workgroups per CU */, post, &lds, &grid, &balanced)) return rc;
  if (post) return launch_cs1b<false, true, true>(a, grid, lds, s);
  if (balanced) {
    if (a.res) return a.act ? launch_cs1b<true, true>(a, grid, lds, s) : launch_cs1b<true, false>(a, grid, lds, s);
    return a.act ? launch_cs1b<false, true>(a, grid, lds, s) : launch_cs1b<false, false>(a, grid, lds, s);
  }
  if (a.res) return a.act ? launch_cs<true, true>(a, grid, lds, s) : launch_cs<true, false>(a, grid, lds, s);
  return a.act ? launch_cs<false, true>(a, grid, lds, s) : launch_cs<false, false>(a, grid, lds, s);
}

// The Detect levels' cv2[i][1] + cv2[i][2] + decode (cfg.cs == 2 each) as ONE launch: conv3_csg_kernel
bool conv3_cs_group_fits(const ConvConfig* cfg, int n, int batch, const int* Ho, const int* Wo) {
  if (n < 2 || n > 3) return false;
  long long tiles = 0;
  for (int i = 0; i < n; ++i) {
    if (cfg[i].MT != 4 || cfg[i].ncb != 1 || !cfg[i].post) return false;
    const long long img = (long long)(Ho[i] + 1) * (Wo[i] + 1), nq = (long long)batch * img;
    if (nq * img >= (1LL << 32) || img * (Wo[i] + 1) >= (1LL << 32) || (size_t)8 * cs_PL(Wo[i]) * 16 + 24 * 1024 > 80 * 1024) return false;
    tiles += (nq + CS_L - 1) / CS_L;
  }
  return env_int("YL_CONV_CS", 1) && tiles >= env_int("YL_CONV_CS_MIN", 512);
}

int conv3_cs_group_launch(const ConvArgs* args, int n, int batch, hipStream_t s) {
  if (n < 2 || n > 3) return fail(-3, "conv3_cs group: two or three members");
  CsGroup G{};
  const int resident = 2 * device_cu_count();
  long long tiles[3] = {0, 0, 0}, total = 0;
  for (int i = 0; i < n; ++i) {
    tiles[i] = ((long long)batch * args[i].in_img + CS_L - 1) / CS_L;
    total += tiles[i];
  }
  // the resident workgroups in proportion to the members' tiles, at least one each
  int left = resident, grid = 0;
  size_t lds = 0;
  for (int i = 0; i < n; ++i) {
    int share = i == n - 1 ? left : (int)((resident * tiles[i] + total / 2) / total);
    if (share < 1) share = 1;
    if (share > left - (n - 1 - i)) share = left - (n - 1 - i);
    left -= share;
    G.m[i] = args[i];
    size_t l = 0;
    int g_i = 0, bal = 0;
    if (int rc = cs_prepare(G.m[i], batch, share, true, &l, &g_i, &bal)) return rc;
    G.nwg[i] = g_i;
    grid += g_i;
    lds = l > lds ? l : lds;
  }
  auto kern = conv3_csg_kernel<true>;
  if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), (int)(160 * 1024))) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, G);
  YL_HIP(hipGetLastError());
  return 0;
}

}  // namespace yl
