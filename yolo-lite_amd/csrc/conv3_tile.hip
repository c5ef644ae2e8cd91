// 3x3 stride-1 Conv + (folded BN) bias + SiLU [+ residual] with the INPUT TILE IN LDS (gfx950).
//
// Same GEMM mapping, packed filters and epilogue as conv_kernel (conv_mfma.hip; reference
// nn/modules/conv.py:35-53, block.py:341-343), different operand delivery.  conv_kernel fetches every B
// fragment (16 pixels x 32 channels) of every tap straight from L1/L2: nine tap-shifted reads of the same
// pixels through the texture addresser, which at 64 B/clk per CU is exactly as fast as the MFMA pipe
// consumes them at Cout = 64 -- and in practice latency-bound well below that.
// Here a workgroup stages the pixels ONCE:
//
//   * In the guarded planar layout (yl_internal.h) a 3x3 tap is a constant offset in the linear guarded
//     pixel index q, so the convolution is one-dimensional in q: outputs [q0, q0 + L) need inputs
//     [q0 - W1 - 1, q0 + L + W1 + 1) of each plane -- ONE contiguous run per plane, copied global -> LDS
//     by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, no VGPR staging).  Guard positions
//     inside the run are computed like pixels (1.2 % of the MFMA work at 80x80) and simply not stored.
//   * 512-thread workgroups, one per CU, persistent over tiles t = wg, wg + G, ...: the filter slab is
//     staged once per workgroup; the input arrives in PHASES of 32 channels (4 planes) into two LDS
//     buffers, the DMA of phase s + 1 running under the 9 k-steps (144 MFMAs per wave) of phase s.
//   * B fragments: ds_read_b128 at (plane g, pixel) = 16 lanes x 16 B contiguous per plane; with the
//     plane pitch a multiple of 256 B the four lane groups of a read hit 64 distinct banks for every tap
//     shift.  Row bases live in 3 VGPRs, dx and the column tile are instruction offsets: no address
//     arithmetic in the k-loop.  A fragments come from the slab exactly as in conv_kernel.
//   * LDS at 64 -> 64, 80x80: slab 72 KiB + 2 x 4 planes x 688 px x 16 B = 158 KiB of 160.
//
// Measured on MI355X (64 -> 64 @80x80, batch 32, tools/tile_bench.sh / tile_abl.sh): 28.6 us against 34.9 us
// for conv_kernel (21 % vs 17.6 % of the bf16 MFMA peak); at 40x40 the two are equal, so conv3_tile_choose
// takes only layers with >= 300 tiles.  Ablations of the 28.6 us: launch of 256 x 512 threads with 160 KiB
// LDS 3.3 us; + barriers, bias, loop shell 5.4; MFMA loop 12 (two rounds of tiles: 411 tiles on 256 CUs);
// stores 3-4 (all workgroups finish their tiles together, so the 26 MB leave as chip-wide bursts); SiLU 1-2;
// first DMA + slab exposed in the prologue 2.  rocprofv3: SQ_LDS_BANK_CONFLICT = 0, SQ_WAIT_ANY 37 % of wave
// cycles.  Variants measured and dropped this round (all bit-identical, 7/7 tests): two wave groups
// alternating compute and epilogue/DMA intervals on three rotating buffers (30 us: the single active wave of
// a SIMD and its partner's epilogue slow each other down); epilogue of tile i interleaved into the MFMAs of
// tile i + 1 by sched_group_barrier (32 us: 256 VGPRs, 24 spilled); a hand-pinned ds_read pipeline (+15 %).
#include "conv_common.h"

namespace yl {

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

template <int NTHREADS>
__device__ __forceinline__ void copy_slab_n(const bf16x8* __restrict__ src, bf16x8* dst, int n16, int tid) {
  for (int i0 = tid; i0 < n16; i0 += NTHREADS * 8) {
    bf16x8 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * NTHREADS;
      if (i < n16) t[u] = src[i];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * NTHREADS;
      if (i < n16) dst[i] = t[u];
    }
  }
}

// Position -> output pixel: which of a wave's guarded positions are pixels (b < B, y < H, x < W); the rest
// (guard columns / rows, the run past the last image) is computed but never stored.
template <int NT>
__device__ __forceinline__ void tile_opix(const ConvArgs& a, int r0, int n, unsigned (&po)[NT]) {
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

// Epilogue for planar bf16 outputs with Cout == MT*16 (MT even) and SiLU: residual, bf16 pack, one 16-byte
// store per (tile pair, column tile).  Straight-line code: dead positions resolve to out-of-range buffer
// offsets instead of branches (the generic epilogue of conv_common.h branches per store).
template <int MT, int NT, bool RES>
__device__ __forceinline__ void epilogue_planar(const ConvArgs& a, __amdgpu_buffer_rsrc_t rout,
                                                __amdgpu_buffer_rsrc_t rres, int g, const unsigned (&po)[NT],
                                                const f32x4 (&acc)[MT][NT]) {
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
      if (!(a.dbg & 8)) {
        silu8(v);  // the kernel is only launched for act == SiLU
      }
      if constexpr (RES) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += (float)rv[p][nt][r];
      }
      store_b128_nt(u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])},
                    rout, (unsigned)(p * 4 + g) * a.out_plane + po[nt], a.st_nt);
    }
}

// NPH = Cin / 32 input phases per tile (1 or 2); a wave owns NT*16 consecutive guarded positions of the
// workgroup's tile of L = 128*NT.  Per phase: one barrier (this phase's DMA pieces have landed -- every wave
// waits for its own before the barrier -- and every wave is done with the other buffer), the DMA of the
// next phase, 9 k-steps.
template <int MT, int NT, int NPH, bool RES>
__global__ __launch_bounds__(512) void conv3_tile_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int L = 128 * NT;
  if (a.dbg & 32) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  const int G = gridDim.x;
  const int wg = xcd_remap(blockIdx.x, G);
  const __amdgpu_buffer_rsrc_t rout =
      __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (a.dbg & 2) ? 0 : a.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.res, (short)0, a.res ? a.res_bytes : 0, 0x00020000);
  const int slab_bytes = a.ksteps * MT * 1024;
  const int buf_bytes = 4 * a.t_PL * 16;
  unsigned char* const tile0 = smem + slab_bytes;
  const int nmine = (a.t_ntiles - wg + G - 1) / G;  // tiles wg, wg + G, ... (> 0: grid <= tiles)

  // LDS-DMA of one phase: 4 planes x t_px pixels in pieces of 64 pixels (1 KiB per wave instruction),
  // dealt round-robin to the eight waves
  auto issue_dma = [&](int t, int c, int buf) {
    if (a.dbg & 1) return;
    const int qs = a.in_g0 + t * L - a.in_w1 - 1;  // first staged guarded position (>= 0)
    const int npieces = 4 * a.t_ppp;
    for (int p = wave; p < npieces; p += 8) {
      const int pl = (int)__umulhi((unsigned)p, a.magic_ppp);
      const int pp = p - pl * a.t_ppp;
      const int px = pp * 64 + lane;
      const int q = min(qs + px, a.t_plane_px - 1);  // past the plane's end only positions no valid output reads
      const char* src = reinterpret_cast<const char*>(a.in) + (size_t)(c * 4 + pl) * a.in_plane + (size_t)q * 16;
      unsigned char* dst = tile0 + buf * buf_bytes + (pl * a.t_PL + pp * 64) * 16;  // wave-uniform
      if (px < a.t_px) __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)dst, 16, 0, 0);
    }
  };

  issue_dma(wg, 0, 0);
  if (!(a.dbg & 4))
    copy_slab_n<512>(reinterpret_cast<const bf16x8*>(a.wpk), reinterpret_cast<bf16x8*>(smem), a.ksteps * MT * 64, tid);
  f32x4 bias_r[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
    bias_r[mt] = *reinterpret_cast<const f32x4*>(a.bias + chan_in_block(MT, mt, g, 0));
  const bf16x8* wl = reinterpret_cast<const bf16x8*>(smem) + lane;
  const int lane_px = (g * a.t_PL + wave * (16 * NT) + n) * 16;  // byte offset of the lane's position 0 in a buffer
  const int row_bytes = a.in_w1 * 16;

  for (int i = 0; i < nmine; ++i) {
    const int t = wg + i * G;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = bias_r[mt];
#pragma unroll
    for (int c = 0; c < NPH; ++c) {
      const int buf = NPH == 2 ? c : (i & 1);
      __syncthreads();  // with an LDS-DMA pending this is s_waitcnt vmcnt(0) + s_barrier
      if (c + 1 < NPH) issue_dma(t, c + 1, buf ^ 1);
      else if (i + 1 < nmine) issue_dma(t + G, 0, buf ^ 1);
      if (a.dbg & 16) continue;
      const unsigned char* tb = tile0 + buf * buf_bytes + lane_px;
      // Row bases in registers, dx and the column tile as instruction offsets.  The ds_read order is left
      // to the compiler: a hand-pinned pipeline (next tap's fragments requested one per MFMA at the head
      // of a tap, __builtin_amdgcn_sched_group_barrier) measured 15 % SLOWER with two waves per SIMD.
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap % 3;
        bf16x8 wa[MT], xb[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) wa[mt] = wl[((tap * NPH + c) * MT + mt) * 64];
        const unsigned char* rb = tb + dy * row_bytes;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) xb[nt] = *reinterpret_cast<const bf16x8*>(rb + dx * 16 + nt * 256);
        mma_step<MT, NT>(wa, xb, acc);
      }
    }
    unsigned po[NT];
    tile_opix<NT>(a, t * L + wave * (16 * NT), n, po);
    if (!(a.dbg & 64)) epilogue_planar<MT, NT, RES>(a, rout, rres, g, po, acc);
  }
}

// ------------------------------------------------------------------------------------ host
static unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull / (unsigned)d) + 1ull); }
static const size_t LDS_MAX = 160 * 1024;
static const int NUM_CU = 256;

static int tile_PL(int NT, int W) { return (128 * NT + 2 * (W + 1) + 2 + 15) & ~15; }

void conv3_tile_choose(int batch, int Cin, int Cout, int k, int stride, int Ho, int Wo, int post, ConvConfig* c) {
  c->tile = 0;
  // read at every engine build (not cached) so that tests can force the kernel onto small maps
  // Opt-in (YL_CONV_TILE=1): faster per layer and in one-batch-at-a-time latency, but a 160 KiB workgroup
  // owns its CU, and with four batches in flight (bench.py) the kernels of the other streams can no longer
  // fill in beside it: 38.2-39.2 k images/s with the tile kernel against 39.1-39.5 k without.
  const int enable = getenv("YL_CONV_TILE") ? atoi(getenv("YL_CONV_TILE")) : 0;
  const int min_tiles = getenv("YL_CONV_TILE_MIN") ? atoi(getenv("YL_CONV_TILE_MIN")) : 300;
  const int max_nt = getenv("YL_CONV_TILE_NT") ? atoi(getenv("YL_CONV_TILE_NT")) : 4;
  if (!enable || k != 3 || stride != 1 || post) return;
  if ((Cin != 32 && Cin != 64) || c->ncb != 1 || (c->MT != 2 && c->MT != 4) || Cout != c->MT * 16) return;
  const long long nq = (long long)batch * (Ho + 1) * (Wo + 1);
  for (int NT = 4; NT >= 1; NT >>= 1) {
    if (NT > max_nt) continue;
    const long long ntiles = (nq + 128 * NT - 1) / (128 * NT);
    const size_t lds = (size_t)c->ksteps * c->MT * 1024 + (size_t)2 * 4 * tile_PL(NT, Wo) * 16;
    if (ntiles >= min_tiles && lds <= LDS_MAX) {
      c->tile = NT;
      // (this kernel stages the whole filter itself, whatever slab budget conv_choose chunked it for)
      c->name = "conv3x3_tile_mfma";
      return;
    }
  }
}

template <int MT, int NT, int NPH, bool RES>
static int launch_tile_r(const ConvArgs& a, int grid, size_t lds, hipStream_t s) {
  auto kern = conv3_tile_kernel<MT, NT, NPH, RES>;
  if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), (int)((int)LDS_MAX))) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

template <int MT, int NT, int NPH>
static int launch_tile(const ConvArgs& a, int grid, size_t lds, hipStream_t s) {
  return a.res ? launch_tile_r<MT, NT, NPH, true>(a, grid, lds, s) : launch_tile_r<MT, NT, NPH, false>(a, grid, lds, s);
}

int conv3_tile_launch(const ConvConfig& cfg, ConvArgs a, int batch, hipStream_t s) {
  const int NT = cfg.tile, L = 128 * NT;
  if (a.stride != 1 || !a.act || a.out_f32 || a.in2 || a.wpk2 || a.in_w1 != a.out_w1 || a.in_img != a.out_img ||
      a.in_g0 != a.out_g0 || a.in_w1 != a.Wo + 1)
    return fail(-3, "conv3_tile: needs a plain 3x3 stride-1 conv between planar tensors of one geometry");
  a.ncb = 1;
  a.cg = cfg.cg;
  a.ksteps = cfg.ksteps;
  a.t_B = batch;
  a.t_H = a.Ho;
  a.t_W = a.Wo;
  a.t_px = L + 2 * a.in_w1 + 2;
  a.t_PL = tile_PL(NT, a.Wo);
  a.t_ppp = (a.t_px + 63) / 64;
  a.magic_ppp = magic_of(a.t_ppp);
  a.magic_img = magic_of(a.in_img);
  a.magic_w1 = magic_of(a.in_w1);
  const long long nq = (long long)batch * a.in_img;
  a.t_ntiles = (int)((nq + L - 1) / L);
  a.t_plane_px = (int)(a.in_plane / 16);
  const long long lim = 1LL << 30;
  const long long outb = (long long)((a.Cout + 7) / 8) * a.out_plane;
  const long long resb = a.res ? (long long)((a.Cout + 7) / 8) * a.res_plane : 0;
  if ((long long)cfg.cg * a.in_plane > lim || outb > lim || resb > lim)
    return fail(-3, "conv3_tile: a tensor slice exceeds 1 GiB (32-bit buffer offsets); lower the batch size");
  a.out_bytes = (unsigned)outb;
  a.res_bytes = (unsigned)resb;
  {
    static const int dbg = getenv("YL_CONV_DBG") ? atoi(getenv("YL_CONV_DBG")) : 0;  // ablation, dev only
    a.dbg = dbg;
  }
  const size_t lds = (size_t)cfg.ksteps * cfg.MT * 1024 + (size_t)2 * 4 * a.t_PL * 16;
  if (lds > LDS_MAX) return fail(-3, "conv3_tile: LDS budget exceeded");
  static const int max_grid = getenv("YL_CONV_TILE_GRID") ? atoi(getenv("YL_CONV_TILE_GRID")) : NUM_CU;
  const int grid = a.t_ntiles < max_grid ? a.t_ntiles : max_grid;
  const int nph = cfg.cg / 4;
  switch (cfg.MT * 100 + NT * 10 + nph) {
    case 441: return launch_tile<4, 4, 1>(a, grid, lds, s);
    case 442: return launch_tile<4, 4, 2>(a, grid, lds, s);
    case 421: return launch_tile<4, 2, 1>(a, grid, lds, s);
    case 422: return launch_tile<4, 2, 2>(a, grid, lds, s);
    case 411: return launch_tile<4, 1, 1>(a, grid, lds, s);
    case 412: return launch_tile<4, 1, 2>(a, grid, lds, s);
    case 241: return launch_tile<2, 4, 1>(a, grid, lds, s);
    case 242: return launch_tile<2, 4, 2>(a, grid, lds, s);
    case 221: return launch_tile<2, 2, 1>(a, grid, lds, s);
    case 222: return launch_tile<2, 2, 2>(a, grid, lds, s);
    case 211: return launch_tile<2, 1, 1>(a, grid, lds, s);
    case 212: return launch_tile<2, 1, 2>(a, grid, lds, s);
  }
  return fail(-3, "conv3_tile: unsupported tile shape");
}

}  // namespace yl
