// 3x3 stride-1 Conv + (folded BN) bias + SiLU [+ residual] with the FILTER IN REGISTERS (gfx950).
//
// Same GEMM mapping, packed filters and epilogue as conv_kernel / conv3_tile_kernel (reference
// nn/modules/conv.py:35-53, block.py:341-343); what changes is where the operands live while the matrix pipe runs:
//
//   * gfx950 gives one wave 512 registers per lane (256 VGPR + 256 AGPR, one file; an MFMA reads A / B from either).
//     A 64 -> 64 3x3 filter is 18 k-steps x 4 channel tiles x one 16-byte fragment per lane = 288 registers: every
//     wave of a 4-wave workgroup (one wave per SIMD) keeps the WHOLE filter in its registers for the life of the
//     kernel.  The k-loop then reads only the activation fragments from LDS: NT ds_read_b128 per 4 x NT MFMAs (25 % of
//     the LDS rate at NT = 4) instead of 4 + NT with the filter slab in LDS (50 %, the measured bound of
//     conv3_tile_kernel: its MFMA loop runs at half the pipe rate).
//   * The input arrives as in conv3_tile_kernel: in the guarded planar layout a 3x3 tap is a constant offset of the
//     linear guarded index q, so a workgroup stages ONE contiguous run of L + 2 (W + 1) + 2 positions per plane by
//     LDS-DMA (global_load_lds_dwordx4) into one of two buffers; with no filter slab the whole 160 KiB of LDS holds
//     activations (all Cin / 8 planes of a tile at once, no channel phases).  The DMA pieces of tile i + 1 are issued
//     one per k-step BETWEEN the MFMAs of tile i (a lone wave per SIMD has nobody else to hide their issue cost).
//   * Persistent workgroups (one per CU) walk the tiles t = wg, wg + G, ...; one barrier per tile.
#include <type_traits>

#include "conv_common.h"

namespace yl {

typedef __attribute__((address_space(3))) void lds_void_r;
typedef const __attribute__((address_space(1))) void glb_void_r;

template <int NT>
__device__ __forceinline__ void reg_tile_opix(const ConvArgs& a, int r0, int n, unsigned (&po)[NT]) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int r = r0 + nt * 16 + n;  // position relative to g0
    const int b = (int)__umulhi((unsigned)r, a.magic_img);
    const int rr = r - b * a.in_img;
    const int y = (int)__umulhi((unsigned)rr, a.magic_w1);
    const int x = rr - y * a.in_w1;
    po[nt] = (b < a.t_B && y < a.t_H && x < a.t_W) ? (unsigned)(a.in_g0 + r) * 16u : OOB_LANE;
  }
}

template <int MT, int NT, bool RES>
__device__ __forceinline__ void reg_epilogue(const ConvArgs& a, __amdgpu_buffer_rsrc_t rout, __amdgpu_buffer_rsrc_t rres,
                                             int g, const unsigned (&po)[NT], const f32x4 (&acc)[MT][NT]) {
  bf16x8 rv[MT / 2][NT];
  if constexpr (RES) {
#pragma unroll
    for (int p = 0; p < MT / 2; ++p)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        rv[p][nt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                   rres, (unsigned)(p * 4 + g) * a.res_plane + po[nt], 0, 0));
  }
#pragma unroll
  for (int p = 0; p < MT / 2; ++p)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[2 * p][nt][r];
        v[4 + r] = acc[2 * p + 1][nt][r];
      }
      if (a.act && !(a.dbg & 8)) {
        silu8(v);
      }
      if constexpr (RES) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += (float)rv[p][nt][r];
      }
      store_b128_nt(u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])},
                    rout, (unsigned)(p * 4 + g) * a.out_plane + po[nt], a.st_nt);
    }
}

// The k-loop's MFMAs are inline asm so that the filter fragments can be read IN PLACE from the accumulator-register
// half of the file ("a" constraint): hipcc parks values in AGPRs only behind v_accvgpr_read copies (16 extra VALU
// instructions per k-step here) and insists on AGPR accumulators for its own MFMAs, which would cost 128 more copies
// per tile at every builtin / asm boundary -- hence ALL of them are asm.  Hazards (cdna_hip_programming.md 5.7: hipcc
// pads nothing inside or around an asm statement):
//   * operands: A = filter registers written once in the prologue; B = ds_read results (hipcc's own s_waitcnt covers
//     asm operands); C of a tile's first k-step = the bias registers (written in the prologue), so no VALU write is
//     ever close in front of an MFMA that reads it;
//   * D is consumed only by the next MFMA of the same accumulator chain (whole register as C: 0 wait states) until the
//     tile is done; then `mfma_drain` (s_nop 15 = 16 states, more than any XDL write -> VALU read needs) ties every
//     accumulator so that the epilogue's reads are ordered behind it.
template <bool AREG>
__device__ __forceinline__ void mfma_acc(f32x4& acc, const u32x4& wa, const bf16x8& xb) {
  if constexpr (AREG)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(wa), "v"(xb));
  else
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(xb));
}
template <bool AREG>
__device__ __forceinline__ void mfma_first(f32x4& acc, const u32x4& wa, const bf16x8& xb, const f32x4& c) {
  if constexpr (AREG)
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "a"(wa), "v"(xb), "v"(c));
  else
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(wa), "v"(xb), "v"(c));
}
// The previous tile's epilogue as single VALU instructions in volatile asm, so that they stay exactly where the source puts
// them BETWEEN the (equally volatile) MFMAs: an MFMA holds the SIMD's issue port for 8 of its 16 cycles, one v_exp /
// v_rcp (8 cycles) or two plain VALU instructions fit in the rest.  Left to itself hipcc emits the epilogue as one block
// of ~45 instructions per store, during which the matrix pipe idles (measured: slower than no overlap at all).
// Same instructions and operand order as silu_f() + the (__bf16) casts compile to, so results are bit-identical.
// A transcendental's result must not be read by the very next instruction (gfx940 trans forwarding hazard): the
// stream below always has an independent instruction of the partner value in between.
__device__ __forceinline__ void e_mul(float& t, float x) { asm volatile("v_mul_f32 %0, 0xbfb8aa3b, %1" : "=v"(t) : "v"(x)); }
__device__ __forceinline__ void e_exp(float& t) { asm volatile("v_exp_f32 %0, %0" : "+v"(t)); }
__device__ __forceinline__ void e_add1(float& t) { asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(t)); }
__device__ __forceinline__ void e_rcp(float& t) { asm volatile("v_rcp_f32 %0, %0" : "+v"(t)); }
__device__ __forceinline__ void e_mulx(float& t, float x) { asm volatile("v_mul_f32 %0, %1, %0" : "+v"(t) : "v"(x)); }
__device__ __forceinline__ void e_cvt(unsigned& d, float lo, float hi) {
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(lo), "v"(hi));
}
struct EpiRegs {
  float t[8];
  unsigned d[4];
};
// The stream is cut into UNITS of at most 8 issue cycles -- what fits beside one MFMA: per pair of values (A, B)
//   [mul A, mul B] [exp A] [exp B] [add A, add B] [rcp A] [rcp B] [mul A, mul B] [pack (+ store)]
// i.e. 8 units per pair, 32 per 16-byte store.  (A first version spread single operations evenly, 1.2 per MFMA: gaps
// holding two transcendentals overran their MFMA by 8 cycles and nothing gave that back -- no gain at all.)
constexpr int EPI_UNITS = 32;

template <int MT, int NT>
__device__ __forceinline__ void mfma_drain(f32x4 (&acc)[MT][NT]) {
  asm volatile("s_nop 15");
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(acc[mt][nt]));
}

// MT * 16 = Cout (32 or 64), NPH * 32 = Cin (32 or 64); a wave owns NT * 16 consecutive guarded positions of the
// workgroup's tile of L = 64 * NT.
template <int MT, int NT, int NPH, bool RES, bool ACT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv3_reg_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int L = 64 * NT;
  constexpr int KS = 9 * NPH;     // k-steps of 32
  constexpr int NPL = 4 * NPH;    // input planes
  constexpr int NPW = NPL / 4;    // planes each wave stages: wave, wave + 4
  constexpr int NSTORE = (MT / 2) * NT;  // 16-byte stores per lane and tile
  constexpr int EOFF = KS >= 18 ? 4 * MT * NT : 0;  // MFMAs of a tile before the interleaved epilogue stream starts (4 k-steps)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  const int G = gridDim.x;
  const int wg = xcd_remap(blockIdx.x, G);
  const __amdgpu_buffer_rsrc_t rout =
      __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (a.dbg & 2) ? 0 : a.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.res, (short)0, a.res ? a.res_bytes : 0, 0x00020000);
  // LDS-DMA source: the input slice as a raw buffer, so that a run past its end reads zeros and "no next tile" is a
  // descriptor of size 0 (no memory access at all) instead of a branch around every piece
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, (short)0, (a.dbg & 1) ? 0 : a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rnull = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, (short)0, 0, 0x00020000);
  const int PLB = a.t_PL * 16;          // bytes per staged plane (t_PL = t_ppp * 64 positions: whole 1 KiB pieces)
  const int buf_bytes = NPL * PLB;
  const int nmine = (a.t_ntiles - wg + G - 1) / G;  // tiles wg, wg + G, ... (> 0: grid <= tiles)
  const int ppw = NPW * a.t_ppp;        // DMA pieces per wave and tile
  const unsigned lane16 = (unsigned)lane * 16u;
  const bool early_dma = ppw <= EOFF / NT && !(a.dbg & 512);  // all pieces go out in the first EOFF MFMAs (one per row)

  // Piece j of a wave = 64 positions (1 KiB) of plane wave + 4 * (j / t_ppp); the (plane, piece) counters live in SGPRs.
  struct Dma {
    int pp, src, dst;  // piece inside the plane; byte offset of the piece in the input slice / in LDS
  };
  auto dma_begin = [&](int t, int buf) __attribute__((always_inline)) {
    Dma d;
    d.pp = 0;
    d.src = wave * (int)a.in_plane + (a.in_g0 + t * L - a.in_w1 - 1) * 16;  // first staged guarded position (>= 0)
    d.dst = buf * buf_bytes + wave * PLB;
    return d;
  };
  auto dma_piece = [&](const __amdgpu_buffer_rsrc_t& r, Dma& d) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_r*)(smem + d.dst), 16, (int)(lane16 + (unsigned)d.src), 0, 0, 0);
    d.pp += 1;
    d.src += 1024;
    d.dst += 1024;
    if (d.pp == a.t_ppp) {  // next plane of this wave (scalar branch)
      d.pp = 0;
      d.src += 4 * (int)a.in_plane - a.t_ppp * 1024;
      d.dst += 4 * PLB - a.t_ppp * 1024;
    }
  };
  {
    Dma d = dma_begin(wg, 0);
    for (int j = 0; j < ppw; ++j) dma_piece(rin, d);
  }

  // the filter: this lane's A fragment of every (k-step, channel tile), conv_pack_weights order.  K-steps
  // [0, KA1) live in AGPRs (at most 240 of the 256), the rest in VGPRs.
  constexpr int KA1 = KS * MT * 4 <= 240 ? KS : 240 / (MT * 4);
  u32x4 wr[KS][MT];
  {
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.wpk) + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) wr[ks][mt] = wsrc[(ks * MT + mt) * 64];
  }
  f32x4 bias_r[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
    bias_r[mt] = *reinterpret_cast<const f32x4*>(a.bias + chan_in_block(MT, mt, g, 0));
  const int lane_px = (g * a.t_PL + wave * (16 * NT) + n) * 16;  // byte offset of the lane's position 0 in a buffer
  const int row_bytes = a.in_w1 * 16;
  const int half_bytes = 4 * PLB;  // planes 4 .. 7 (second half of a 64-channel input)

  auto load_b = [&](const unsigned char* tb, int ks, bf16x8 (&xb)[NT]) __attribute__((always_inline)) {
    const int tap = ks / NPH, c = ks % NPH;
    const int dy = tap / 3, dx = tap % 3;
    const unsigned char* rb = tb + c * half_bytes + dy * row_bytes;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) xb[nt] = *reinterpret_cast<const bf16x8*>(rb + dx * 16 + nt * 256);
  };

  // Unit u of the interleaved epilogue stream of one finished tile (accumulators ac, output offsets po): store
  // c = u / EPI_UNITS covers tile pair p = c / NT and column tile nt = c % NT; values j0, j1 of its 8 go through
  // t = -log2(e) x; t = 2^t; t = 1 + t; t = 1 / t; t = x t; then one pack, the fourth pack brings the store.
  EpiRegs E;
  auto epi_op = [&](int u, const f32x4 (&ac)[MT][NT], const unsigned (&po)[NT], EpiRegs& e) __attribute__((always_inline)) {
    const int c = u / EPI_UNITS, r = u % EPI_UNITS, p = c / NT, nt = c % NT;
    const int pair = r / 8, k = r % 8, j0 = 2 * pair, j1 = j0 + 1;
    const float x0 = j0 < 4 ? ac[2 * p][nt][j0] : ac[2 * p + 1][nt][j0 - 4];
    const float x1 = j1 < 4 ? ac[2 * p][nt][j1] : ac[2 * p + 1][nt][j1 - 4];
    if (k == 0) {
      e_mul(e.t[j0], x0);
      e_mul(e.t[j1], x1);
    } else if (k == 1) e_exp(e.t[j0]);
    else if (k == 2) e_exp(e.t[j1]);
    else if (k == 3) {
      e_add1(e.t[j0]);
      e_add1(e.t[j1]);
    } else if (k == 4) e_rcp(e.t[j0]);
    else if (k == 5) e_rcp(e.t[j1]);
    else if (k == 6) {
      e_mulx(e.t[j0], x0);
      e_mulx(e.t[j1], x1);
    } else {
      e_cvt(e.d[pair], e.t[j0], e.t[j1]);
      if (pair == 3 && !(a.dbg & 64))
        store_b128_nt(u32x4{e.d[0], e.d[1], e.d[2], e.d[3]}, rout, (unsigned)(p * 4 + g) * a.out_plane + po[nt], a.st_nt);
    }
  };

  // One tile: barrier, k-loop into `acc` with the next tile's DMA pieces and -- PIPE -- the PREVIOUS tile's epilogue
  // (accumulators `accp`, output offsets `pop`) spread over the k-steps, one store-sized chunk at a time: a lone wave
  // per SIMD has no partner whose MFMAs could cover its SiLU (v_exp + v_rcp are quarter rate), so it covers it itself.
  auto tile_body = [&](auto has_prev, int i, f32x4 (&acc)[MT][NT], const f32x4 (&accp)[MT][NT],
                       const unsigned (&pop)[NT]) __attribute__((always_inline)) {
    constexpr bool HP = decltype(has_prev)::value;
    const int t = wg + i * G;
    const int buf = i & 1;
    // This tile's DMA pieces have landed (every wave waits for its own VMEM operations, then the barrier); every wave
    // has left the previous tile's k-loop -> the other buffer is free.
    // The NSTORE youngest VMEM operations are the previous tile's stores (un-pipelined: issued behind the k-loop;
    // pipelined: the stream starts after the last DMA piece when the pieces fit the first EOFF MFMAs) and may stay in flight.
    if (!HP || early_dma) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool more = i + 1 < nmine;
    const __amdgpu_buffer_rsrc_t rnext = more ? rin : rnull;
    Dma d = dma_begin(t + G, buf ^ 1);
    const unsigned char* tb = smem + buf * buf_bytes + lane_px;
    bf16x8 xb[2][NT];
    load_b(tb, 0, xb[0]);
    int jd = 0;
    const int dpk = a.dbg & 512 ? 1 : MT;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) load_b(tb, ks + 1, xb[(ks + 1) & 1]);  // next step's fragments fly under this step's MFMAs
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          // (the loop is fully unrolled: these conditions fold at compile time)
          if (ks == 0) mfma_first<(0 < KA1)>(acc[mt][nt], wr[0][mt], xb[0][nt], bias_r[mt]);
          else if (ks < KA1) mfma_acc<true>(acc[mt][nt], wr[ks][mt], xb[ks & 1][nt]);
          else mfma_acc<false>(acc[mt][nt], wr[ks][mt], xb[ks & 1][nt]);
          if constexpr (HP) {
            // this MFMA's share of the previous tile's epilogue stream: units [o0, o1) of NSTORE * EPI_UNITS
            // (one per MFMA where the tile has that many MFMAs -- Cin = 64 --, up to three where it has not)
            // The stream starts EOFF MFMAs into the tile: by then every DMA piece of the next tile has been issued (4 per
            // k-step, see below), so all NSTORE stores are younger than all pieces and may stay in flight across the
            // next barrier (counted vmcnt, in-order retirement).
            constexpr int S = KS * MT * NT - EOFF, NU = NSTORE * EPI_UNITS;
            static_assert(NU <= 3 * S, "at most three epilogue units per MFMA");
            const int slot = (ks * MT + mt) * NT + nt - EOFF;
            const int o0 = slot < 0 ? 0 : slot * NU / S, o1 = slot < 0 ? 0 : (slot + 1) * NU / S;
            // (no inner loop: loops are unrolled innermost first, and only straight-line code folds with the outer ones)
            if (o0 < o1) epi_op(o0, accp, pop, E);
            if (o0 + 1 < o1) epi_op(o0 + 1, accp, pop, E);
            if (o0 + 2 < o1) epi_op(o0 + 2, accp, pop, E);
          }
        }
        // The next tile's DMA pieces go out EARLY in this tile, DPK per k-step behind the first two MFMA rows: the last
        // piece then has (KS - ppw / DPK) k-steps to land before the next barrier needs it.  (One piece per k-step left
        // the last ones ~4 k-steps = 0.5 us: SQ_WAIT_ANY showed a quarter of every wave's life parked at the barrier.)
        if (mt < dpk && jd < ppw) {
          dma_piece(rnext, d);
          ++jd;
        }
      }
    }
    for (; jd < ppw; ++jd) dma_piece(rnext, d);
    mfma_drain<MT, NT>(acc);
  };

  const int ntile_run = (a.dbg & 128) ? 0 : nmine;
  constexpr bool PIPE = !RES && ACT;
  if constexpr (PIPE) {
    // two accumulator sets alternate: tile i computes into one while the other one's tile is being written out
    f32x4 accA[MT][NT], accB[MT][NT];
    unsigned poA[NT], poB[NT];
    int i = 0;
    if (ntile_run > 0) {
      tile_body(std::false_type{}, 0, accA, accB, poB);
      reg_tile_opix<NT>(a, wg * L + wave * (16 * NT), n, poA);
      for (i = 1; i + 1 < ntile_run; i += 2) {
        tile_body(std::true_type{}, i, accB, accA, poA);
        reg_tile_opix<NT>(a, (wg + i * G) * L + wave * (16 * NT), n, poB);
        tile_body(std::true_type{}, i + 1, accA, accB, poB);
        reg_tile_opix<NT>(a, (wg + (i + 1) * G) * L + wave * (16 * NT), n, poA);
      }
      if (i < ntile_run) {  // one tile left: into B, then both epilogues below
        tile_body(std::true_type{}, i, accB, accA, poA);
        reg_tile_opix<NT>(a, (wg + i * G) * L + wave * (16 * NT), n, poB);
        if (!(a.dbg & 64)) reg_epilogue<MT, NT, false>(a, rout, rres, g, poB, accB);
      } else {
        if (!(a.dbg & 64)) reg_epilogue<MT, NT, false>(a, rout, rres, g, poA, accA);
      }
    }
  } else {
    for (int i = 0; i < ntile_run; ++i) {
      f32x4 acc[MT][NT];
      unsigned po[NT];
      tile_body(std::false_type{}, i, acc, acc, po);
      reg_tile_opix<NT>(a, (wg + i * G) * L + wave * (16 * NT), n, po);
      if (!(a.dbg & 64)) reg_epilogue<MT, NT, RES>(a, rout, rres, g, po, acc);
    }
  }
}

// ------------------------------------------------------------------------------------ host
static unsigned magic_of_r(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull / (unsigned)d) + 1ull); }
static const size_t LDS_MAX_R = 160 * 1024;
static const int NUM_CU_R = 256;

static int reg_PL(int NT, int W) { return (64 * NT + 2 * (W + 1) + 2 + 63) & ~63; }  // whole 64-position DMA pieces

void conv3_reg_choose(int batch, int Cin, int Cout, int k, int stride, int Ho, int Wo, int post, ConvConfig* c) {
  c->reg = 0;
  const int enable = env_int("YL_CONV_REG", 1);
  const int min_tiles = env_int("YL_CONV_REG_MIN", 512);
  const int want_nt = env_int("YL_CONV_REG_NT", 4);
  if (!enable || k != 3 || stride != 1 || post) return;
  if ((Cin != 32 && Cin != 64) || (Cout != 32 && Cout != 64) || c->ncb != 1 || Cout != c->MT * 16) return;
  const long long nq = (long long)batch * (Ho + 1) * (Wo + 1);
  // Workgroups are persistent, one per CU, so a launch takes ceil(tiles / 256) rounds of NT units each: pick the tile
  // width with the fewest units (64 -> 64 @80x80, batch 32: NT = 4 needs 4 rounds = 16 units, NT = 2 seven = 14), the
  // wider one on a tie (less halo: L + 2 W + 4 positions are staged per L outputs).
  int best = 0;
  long long best_units = 0;
  for (int NT = 4; NT >= 2; NT >>= 1) {
    if (NT > want_nt) continue;
    const long long ntiles = (nq + 64 * NT - 1) / (64 * NT);
    const size_t lds = (size_t)2 * (Cin / 8) * reg_PL(NT, Wo) * 16;
    if (ntiles < min_tiles || lds > LDS_MAX_R) continue;
    const long long units = (ntiles + NUM_CU_R - 1) / NUM_CU_R * NT;
    if (!best || units < best_units) {
      best = NT;
      best_units = units;
    }
  }
  if (best) {
    c->reg = best;
    c->name = "conv3x3_reg_mfma";
  }
}

template <int MT, int NT, int NPH, bool RES, bool ACT>
static int launch_reg_r(const ConvArgs& a, int grid, size_t lds, hipStream_t s) {
  auto kern = conv3_reg_kernel<MT, NT, NPH, RES, ACT>;
  if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), (int)((int)LDS_MAX_R))) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

template <int MT, int NT, int NPH>
static int launch_reg(const ConvArgs& a, int grid, size_t lds, hipStream_t s) {
  // (the interleaved epilogue stream exists for the common case: SiLU, no residual)
  // Opt-in (YL_CONV_REG_PIPE=1): bit-identical and measured EQUAL in time to the plain form on MI355X (64 -> 64 @80x80:
  // 26-30 us either way, box to box): VALU work placed between one wave's own MFMAs did not overlap with them here.
  const int pipe = tune_int("YL_CONV_REG_PIPE", 0);
  if (a.res) return launch_reg_r<MT, NT, NPH, true, false>(a, grid, lds, s);
  return (a.act && pipe) ? launch_reg_r<MT, NT, NPH, false, true>(a, grid, lds, s)
                         : launch_reg_r<MT, NT, NPH, false, false>(a, grid, lds, s);
}

int conv3_reg_launch(const ConvConfig& cfg, ConvArgs a, int batch, hipStream_t s) {
  const int NT = cfg.reg, L = 64 * NT;
  if (a.stride != 1 || a.out_f32 || a.in2 || a.wpk2 || a.in_w1 != a.out_w1 || a.in_img != a.out_img ||
      a.in_g0 != a.out_g0 || a.in_w1 != a.Wo + 1)
    return fail(-3, "conv3_reg: needs a plain 3x3 stride-1 conv between planar tensors of one geometry");
  a.ncb = 1;
  a.cg = cfg.cg;
  a.ksteps = cfg.ksteps;
  a.t_B = batch;
  a.t_H = a.Ho;
  a.t_W = a.Wo;
  a.t_px = L + 2 * a.in_w1 + 2;
  a.t_PL = reg_PL(NT, a.Wo);
  a.t_ppp = (a.t_px + 63) / 64;
  a.magic_ppp = magic_of_r(a.t_ppp);
  a.magic_img = magic_of_r(a.in_img);
  a.magic_w1 = magic_of_r(a.in_w1);
  const long long nq = (long long)batch * a.in_img;
  a.t_ntiles = (int)((nq + L - 1) / L);
  a.t_plane_px = (int)(a.in_plane / 16);
  const long long lim = 1LL << 30;
  const long long outb = (long long)((a.Cout + 7) / 8) * a.out_plane;
  const long long resb = a.res ? (long long)((a.Cout + 7) / 8) * a.res_plane : 0;
  if ((long long)cfg.cg * a.in_plane > lim || outb > lim || resb > lim)
    return fail(-3, "conv3_reg: a tensor slice exceeds 1 GiB (32-bit buffer offsets); lower the batch size");
  a.in_bytes = (unsigned)((long long)cfg.cg * a.in_plane);
  a.out_bytes = (unsigned)outb;
  a.res_bytes = (unsigned)resb;
  {
    static const int dbg = tune_int("YL_CONV_DBG", 0);  // ablation, dev only
    a.dbg = dbg;
  }
  const size_t lds = (size_t)2 * cfg.cg * a.t_PL * 16;
  if (lds > LDS_MAX_R) return fail(-3, "conv3_reg: LDS budget exceeded");
  static const int max_grid = tune_int("YL_CONV_REG_GRID", NUM_CU_R);
  const int grid = a.t_ntiles < max_grid ? a.t_ntiles : max_grid;
  const int nph = cfg.cg / 4;
  switch (cfg.MT * 100 + NT * 10 + nph) {
    case 442: return launch_reg<4, 4, 2>(a, grid, lds, s);
    case 422: return launch_reg<4, 2, 2>(a, grid, lds, s);
    case 441: return launch_reg<4, 4, 1>(a, grid, lds, s);
    case 421: return launch_reg<4, 2, 1>(a, grid, lds, s);
    case 242: return launch_reg<2, 4, 2>(a, grid, lds, s);
    case 222: return launch_reg<2, 2, 2>(a, grid, lds, s);
    case 241: return launch_reg<2, 4, 1>(a, grid, lds, s);
    case 221: return launch_reg<2, 2, 1>(a, grid, lds, s);
  }
  return fail(-3, "conv3_reg: unsupported tile shape");
}

}  // namespace yl
