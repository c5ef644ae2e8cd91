// e4m3 3x3 stride-1 Conv 48 -> 48 / 96 -> 96 (three / six planes of 16 channels) with the INPUT TILE IN LDS and the FILTER IN REGISTERS
// (round 5).
//
// Reference semantics and quantisation exactly as conv_f8.hip (nn/modules/conv.py:35-53 with the BN folded, Bottleneck add
// nn/modules/block.py:341-343; BASELINE.json config 5): same packed filters (conv_f8_pack_weights order), same k-steps of eight
// 16-channel groups in the same order (tap-major or plane-major, ConvArgs::kord), same fp32 accumulation chain, same epilogue --
// results are bit-identical to conv_f8_kernel's.  What changes is the operand delivery.  conv_f8_kernel fetches every B fragment
// from the planes (two 16-byte loads per lane, pixel tile and k-step, two k-steps ahead): on the 48-channel Bottleneck convs of
// YOLO11x @1280 (320x320 maps, 34 GFLOP, 79-197 MB each) that is 62-125 us a launch = 0.27-0.55 PFLOP/s -- four k-steps per
// pixel patch are too short a loop to hide an L2 round trip behind, whatever the prefetch depth (NOTEBOOK section 12).  Here a
// workgroup stages a run of 256 + 2 (W + 1) + 2 guarded positions of the three input planes by LDS-DMA (the whole tile is ONE
// round trip, requested while the previous tile's epilogue runs), every wave keeps the whole 48 x 432 filter in 96 registers, and
// the k-loop reads its fragments from LDS: 48 MFMAs (16x16x128) per wave and tile behind 32 ds_read_b128.
//
// Tiles, staging, counted wait, barriers and the balanced short first round as conv3_cs1b_kernel (conv3_cs.hip).  48 -> 48: four waves =
// four columns of up to four 16-position fragments (tiles of 256 positions, two workgroups per CU).  96 -> 96 (864 x 96 filter): eight
// waves = two channel blocks of 48 x four columns of up to two fragments (tiles of 128 positions, one workgroup per CU); a wave holds its
// block's 48 x 864 filter in 168 registers.  In the guarded layout a tap is a constant offset of the linear position; a k-group
// (tap, plane) is therefore one constant LDS offset per lane and k-step, computed once per kernel.
#include "conv_common.h"
#include "f8_epilogue.h"

namespace yl {

typedef int f8t_v8i __attribute__((ext_vector_type(8)));
typedef int f8t_v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t8;

constexpr int F8T_NCOL = 4;  // columns of a tile = waves per channel block
constexpr int f8t_ntm(int CG) { return CG == 3 ? 4 : 2; }            // fragments per wave of a full tile
constexpr int f8t_L(int CG) { return F8T_NCOL * f8t_ntm(CG) * 16; }  // positions of a full tile (256 / 128)

static __host__ __device__ inline int f8t_PL(int L, int W1) { return (L + 2 * W1 + 2 + 63) & ~63; }  // staged positions per plane

// One tile of NT fragments per wave: k-loop from LDS, `mid()` (free the buffer, request the next tile), epilogue (conv_f8_kernel's).
template <int CG, int NT, typename Mid>
__device__ __forceinline__ void f8t_tile(const ConvArgs& a, const f8t_v8i (&wf)[(9 * CG + 7) / 8][3], const unsigned (&koff)[(9 * CG + 7) / 8],
                                         const unsigned char* tb, int r0, int n, int g, int cb, const float* lab,
                                         const __amdgpu_buffer_rsrc_t& rout, const __amdgpu_buffer_rsrc_t& rres, Mid mid) {
  constexpr int KSN = (9 * CG + 7) / 8, MT = 3;  // a wave's channel block: 48 output channels
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < KSN; ++ks) {
    f8t_v8i xb[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const f8t_v4i lo = *reinterpret_cast<const f8t_v4i*>(tb + nt * 256 + (koff[ks] & 0xffffu));
      const f8t_v4i hi = *reinterpret_cast<const f8t_v4i*>(tb + nt * 256 + (koff[ks] >> 16));
      xb[nt] = f8t_v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[ks][mt], xb[nt], acc[mt][nt], 0, 0, 0, 0, 0, 0);
  }
  mid();
  // ---- epilogue: dequantise, bias, SiLU, residual, store (valid positions only: guards are computed, never stored)
  int opix[NT];
  const int nqv = a.t_B * a.in_img;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int r = r0 + nt * 16 + n;  // position relative to g0
    // (row of the whole batch first: r / W1 is exact by magic number as long as positions x W1 < 2^32 -- r / img is not on 320x320 maps)
    const int R = (int)__umulhi((unsigned)r, a.magic_w1);
    const int x = r - R * a.in_w1;
    const int b = (int)__umulhi((unsigned)R, a.magic_img);  // magic of H + 1 (rows per image incl. the guard row)
    const int y = R - b * (a.t_H + 1);
    opix[nt] = (r < nqv && y < a.t_H && x < a.t_W) ? a.in_g0 + r : -1;
  }
  f8_epilogue_any<MT, NT>(a, acc, opix, cb, g, lab, rres, rout);
}

// NCBW: channel blocks (of 48 output channels) per workgroup = CG / 3 (one workgroup holds the whole conv: eight waves at 96 channels) or
// 1 (96 channels: TWO four-wave workgroups walk the same tiles, one per channel block, each staging the tile for itself -- half a CU
// per workgroup, which is what twenty batches in flight want)
template <int CG, int NCBW>
__global__ __launch_bounds__(64 * F8T_NCOL * NCBW) __attribute__((amdgpu_waves_per_eu(2))) void conv3_f8t_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tsm[];
  constexpr int KSN = (9 * CG + 7) / 8, MT = 3, NCB = CG / 3, NW = NCBW * F8T_NCOL, NTM = f8t_ntm(CG), L = f8t_L(CG);
  constexpr int NSPLIT = NCB / NCBW;  // workgroups per tile walk
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = wave % F8T_NCOL;  // column of the tile
  const int g = lane >> 4, n = lane & 15;
  const int G = gridDim.x / NSPLIT;
  const int wgraw = xcd_remap(blockIdx.x, gridDim.x);
  const int wg = wgraw / NSPLIT;
  const int cb = wave / F8T_NCOL + (wgraw % NSPLIT) * NCBW;  // channel block of this wave
  const int PL = a.t_PL, PLB = PL * 16;
  const int Lr = a.t_px, nsmall = a.t_ppp, S = nsmall * Lr;  // the short round: tiles of Lr positions, nsmall of them, S positions in all
  const int has_small = wg < nsmall ? 1 : 0;
  const int nfull = a.t_ntiles > wg ? (a.t_ntiles - wg + G - 1) / G : 0;  // this workgroup's full tiles
  const int ntl = has_small + nfull;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, (short)0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, a.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.res, (short)0, a.res ? a.res_bytes : 0, 0x00020000);
  const unsigned lane16 = (unsigned)lane * 16u;
  // tile i of this workgroup: first position (relative to g0) and fragments per wave
  auto tile_of = [&](int i, int& nt) __attribute__((always_inline)) -> int {
    if (has_small) {
      if (i == 0) {
        nt = Lr / (16 * F8T_NCOL);
        return wg * Lr;
      }
      --i;
    }
    nt = NTM;
    return S + (wg + i * G) * L;
  };
  // a tile -> LDS: pieces of 1 KiB (64 positions of one plane), j = wave, wave + NW, .. of every plane; a run past the slice reads zeros
  auto stage = [&](int pos, int nt) __attribute__((always_inline)) {
    const int src0 = (a.in_g0 + pos - a.in_w1 - 1) * 16;  // first staged guarded position (>= 0)
    const int np = (16 * F8T_NCOL * nt + 2 * a.in_w1 + 2 + 63) >> 6;
#pragma unroll
    for (int plane = 0; plane < CG; ++plane)
      for (int j = wave; j < np; j += NW)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (lds_void_t8*)(tsm + plane * PLB + j * 1024), 16,
                                                 (int)(lane16 + (unsigned)(plane * (int)a.in_plane + src0 + j * 1024)), 0, 0, 0);
  };
  int nt_cur = NTM, pos_cur = 0;
  if (ntl > 0) {
    pos_cur = tile_of(0, nt_cur);
    stage(pos_cur, nt_cur);
  }
  // the epilogue's alpha / bias of the workgroup's channel blocks -> LDS behind the tile (f8_epilogue.h; visible behind the first
  // barrier of the tile loop): [block of the workgroup][alpha 48, bias 48]
  float* const lab_all = reinterpret_cast<float*>(tsm + (size_t)CG * PLB);
  if (tid < NCBW * 96) {
    const int blk = tid / 96, i = tid - blk * 96;
    const int cbq = blk + (wgraw % NSPLIT) * NCBW;
    lab_all[tid] = (i < 48 ? a.alpha : a.bias)[cbq * 48 + (i < 48 ? i : i - 48)];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const float* const lab = lab_all + (wave / F8T_NCOL) * 96;
  // this wave's channel block of the filter, conv_f8_pack_weights order [cb][k-step][MT][half][64 lanes] x 16 bytes
  f8t_v8i wf[KSN][MT];
  {
    const f8t_v4i* wsrc = reinterpret_cast<const f8t_v4i*>(a.wpk) + (size_t)cb * KSN * MT * 128 + lane;
#pragma unroll
    for (int ks = 0; ks < KSN; ++ks)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f8t_v4i lo = wsrc[((ks * MT + mt) * 2) * 64], hi = wsrc[((ks * MT + mt) * 2 + 1) * 64];
        wf[ks][mt] = f8t_v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
  }
  // LDS byte offset of this lane's two k-groups of every k-step, relative to the FIRST staged position of its column in plane 0 (tap
  // (kh, kw) = kh rows + kw positions further), two 16-bit halves of one register (host: the tile is at most 64 KB).  A group past the
  // end of K has zero filters: it may read any staged bytes (offset 0).
  unsigned koff[KSN];
#pragma unroll
  for (int ks = 0; ks < KSN; ++ks) {
    unsigned two[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int q = ks * 8 + 2 * g + h;
      int tap, c16;
      if (a.kord) {  // plane-major: k-group = plane * 9 + tap
        c16 = (int)__umulhi((unsigned)q, 477218589u);  // q / 9
        tap = q - 9 * c16;
      } else {
        tap = q / CG;
        c16 = q - tap * CG;
      }
      const bool real = q < 9 * CG;
      const int kh = (tap * 11) >> 5;
      two[h] = real ? (unsigned)(c16 * PLB + (kh * a.in_w1 + (tap - 3 * kh)) * 16) : 0u;
    }
    koff[ks] = two[0] | (two[1] << 16);
  }
  // the lane's position inside the staged run: the run starts W1 + 1 positions in front of the tile (tap (0, 0) of the tile's position 0)
  const unsigned char* const tb0 = tsm + n * 16;
  int nt_prev = 0;
  for (int i = 0; i < ntl; ++i) {
    // my pieces of this tile are OLDER than the previous tile's epilogue stores (at most 2 MT nt_prev of them, fewer without the e4m3
    // copy; a residual's loads have been consumed): a counted wait covers the pieces and leaves the plain stores in flight
    switch (nt_prev) {
      case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * MT) : "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * MT) : "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * MT) : "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * MT) : "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    __builtin_amdgcn_s_barrier();
    int nt_next = NTM, pos_next = 0;
    const bool more = i + 1 < ntl;
    if (more) pos_next = tile_of(i + 1, nt_next);
    auto mid = [&]() __attribute__((always_inline)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // everybody has read the tile: the buffer may be refilled
      if (more) stage(pos_next, nt_next);
    };
    const unsigned char* const tb = tb0 + col * 16 * nt_cur * 16;
    const int r0 = pos_cur + col * 16 * nt_cur;
    if constexpr (NTM == 4) {
      switch (nt_cur) {
        case 4: f8t_tile<CG, 4>(a, wf, koff, tb, r0, n, g, cb, lab, rout, rres, mid); break;
        case 3: f8t_tile<CG, 3>(a, wf, koff, tb, r0, n, g, cb, lab, rout, rres, mid); break;
        case 2: f8t_tile<CG, 2>(a, wf, koff, tb, r0, n, g, cb, lab, rout, rres, mid); break;
        default: f8t_tile<CG, 1>(a, wf, koff, tb, r0, n, g, cb, lab, rout, rres, mid); break;
      }
    } else {
      if (nt_cur == 2) f8t_tile<CG, 2>(a, wf, koff, tb, r0, n, g, cb, lab, rout, rres, mid);
      else f8t_tile<CG, 1>(a, wf, koff, tb, r0, n, g, cb, lab, rout, rres, mid);
    }
    nt_prev = nt_cur;
    nt_cur = nt_next;
    pos_cur = pos_next;
  }
}

// ------------------------------------------------------------------------------------ host
static unsigned f8t_magic(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull / (unsigned)d) + 1ull); }

// may this conv take the tile kernel?  (3x3 stride 1, 48 -> 48 or 96 -> 96 in channel blocks of 48, enough positions, the tile fits the LDS)
bool conv3_f8t_fits(const ConvConfig& cfg, int batch, int Cin, int Cout, int k, int stride, int Ho, int Wo) {
  // YL_F8_TILE: 0 never, 1 the 48-channel form only, 2 (default since the epilogue stopped waiting on its own stores) the 96-channel
  // form too (a pair of four-wave workgroups per tile walk; YL_F8_TILE_W8=1: one eight-wave workgroup).  Config 5, A / B on one box.
  // Round 5 (old epilogue): 48 only 1 117 -> 1 129 images/s (+1 %); with the 96-channel form 1 095-1 105 (-2..-2.7 %, although the
  // launch alone was 8-13 % shorter).  Round 6 (f8_epilogue.h): 1 264-1 267 -> 1 271 (+0.5 %), one batch at a time 1 038-1 044 -> 1 050-1 058.
  const int enable = env_int("YL_F8_TILE", 2);
  if (!enable || k != 3 || stride != 1 || Cin != Cout || (Cin != 48 && !(Cin == 96 && enable >= 2)) || cfg.MT != 3 || cfg.ncb != Cout / 48 ||
      cfg.cg != Cin / 16)
    return false;
  const long long img = (long long)(Ho + 1) * (Wo + 1), nq = (long long)batch * img;
  const long long rows = (long long)batch * (Ho + 1);
  if ((nq + 512) * (Wo + 1) >= (1LL << 32) || rows * (Ho + 1) >= (1LL << 32) || nq >= (1LL << 30)) return false;  // exactness of the magic divisions
  const long long min_pos = env_int("YL_F8_TILE_MIN", 65536);
  if (nq < min_pos) return false;
  const int CG = Cin / 16;
  return (size_t)CG * f8t_PL(CG == 3 ? f8t_L(3) : f8t_L(6), Wo + 1) * 16 <= 64 * 1024 - 64;  // (16-bit tap offsets)
}

template <int CG, int NCBW>
static int f8t_launch_t(ConvArgs a, int batch, hipStream_t s) {
  constexpr int L = f8t_L(CG), U = 16 * F8T_NCOL;  // full tile, unit of the short round
  constexpr int NSPLIT = (CG / 3) / NCBW;          // workgroups per tile walk
  a.t_PL = f8t_PL(L, a.in_w1);
  const long long nq = (long long)batch * a.in_img;
  const long long nqu = (nq + U - 1) / U * U;
  // tile walks that stay resident: two four-wave workgroups or one eight-wave workgroup per CU
  const int resident = (NCBW * F8T_NCOL == 4 ? 2 : 1) * device_cu_count() / NSPLIT;
  const long long per_round = (long long)resident * L;
  const long long rounds = nqu / per_round, rem = nqu - rounds * per_round;
  int Lr = rem > 0 ? (int)(((rem + resident - 1) / resident + U - 1) / U * U) : 0;
  int nsmall = 0;
  if (Lr >= L) Lr = 0;  // the remainder fills whole tiles: no short round
  if (Lr > 0) nsmall = (int)((rem + Lr - 1) / Lr);
  const long long S = (long long)nsmall * Lr;
  const int nfull_tiles = nqu > S ? (int)((nqu - S + L - 1) / L) : 0;
  a.t_px = Lr;
  a.t_ppp = nsmall;
  a.t_ntiles = nfull_tiles;
  int grid = nsmall > nfull_tiles ? nsmall : nfull_tiles;
  if (grid > resident) grid = resident;
  if (grid < 1) grid = 1;
  const size_t lds = (size_t)CG * a.t_PL * 16 + (size_t)NCBW * 96 * 4;  // the tile + alpha / bias of the workgroup's channel blocks
  auto kern = conv3_f8t_kernel<CG, NCBW>;
  if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), (int)(160 * 1024))) return rc;
  hipLaunchKernelGGL(kern, dim3(grid * NSPLIT), dim3(64 * F8T_NCOL * NCBW), lds, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

int conv3_f8t_launch(const ConvConfig& cfg, ConvArgs a, int batch, hipStream_t s) {
  if (a.stride != 1 || a.in_w1 != a.out_w1 || a.in_img != a.out_img || a.in_g0 != a.out_g0 || a.in_w1 != a.Wo + 1 || a.Cin != a.Cout ||
      (a.Cin != 48 && a.Cin != 96) || cfg.ntaps != 9)
    return fail(-3, "conv3_f8t: needs a 48 -> 48 or 96 -> 96 3x3 stride-1 conv between planar tensors of one geometry");
  a.t_B = batch;
  a.t_H = a.Ho;
  a.t_W = a.Wo;
  a.magic_img = f8t_magic(a.Ho + 1);  // (rows per image: the kernel divides position -> batch row -> image)
  a.magic_w1 = f8t_magic(a.in_w1);
  if (a.Cin == 48) return f8t_launch_t<3, 1>(a, batch, s);
  const int wide = env_int("YL_F8_TILE_W8", 0);  // 1: the eight-wave form (one workgroup per CU)
  return wide ? f8t_launch_t<6, 2>(a, batch, s) : f8t_launch_t<6, 1>(a, batch, s);
}

}  // namespace yl
