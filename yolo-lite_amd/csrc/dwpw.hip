// DWConv 3x3 + SiLU -> 1x1 Conv [+ a further 1x1] in one launch, input staged through LDS (gfx950).
//
// The Detect class branch is Sequential(DWConv(c, c, 3), Conv(c, c3, 1)) twice, then nn.Conv2d(c3, nc, 1)
// (reference nn/modules/head.py:41-48).  Same fusion and results as dwpw_kernel (conv_mfma.hip); what changes is how the
// depthwise conv's operands reach the matrix pipe.  dwpw_kernel fetches every 3x3 tap of every pixel from global memory
// (nine 16-byte loads per pixel and plane): its timeline on MI355X shows the depthwise phase bound by the vector L1's
// 64 bytes per clock, and the phases of the ~5 resident workgroups per CU in lock step (L1-bound phase, then HBM-bound
// store phase, two rounds of that).  Here
//   * work is tiled over GUARDED positions like conv3_reg_kernel: in the guarded planar layout a 3x3 tap is a constant
//     offset of the linear position q, so a workgroup stages ONE contiguous run of L + 2 (W + 1) + 2 positions per plane
//     by LDS-DMA and reads the nine taps from LDS (128 bytes per clock, no address generation in the texture path);
//     L1 traffic drops from 10 x to (L + 2 W + 4) / L x the input;
//   * two planes (16 channels = one MFMA row tile of the block-diagonal depthwise GEMM, see dwpw_kernel) are staged per
//     step into a ring of D buffers: the DMA of steps j + 1 .. j + D - 1 is in flight while step j computes (with D = 2
//     every step waited a full memory round trip, ~1 us measured, for ~0.2 us of work); the ring runs across tile
//     boundaries, so the next tile's first steps are staged under the current tile's GEMM and stores;
//   * workgroups are persistent (tiles wg, wg + G, ...): filter slabs, depthwise taps and biases are copied to LDS once --
//     the loop issues no vector-memory LOAD at all, so the only s_waitcnt vmcnt is the counted one of the ring;
//   * a wave computes the depthwise output of exactly the 16 * NT positions its GEMM consumes, so the bf16 intermediate
//     tile in LDS is wave-private and needs no barrier.
// Guard positions are computed like any other and not stored (their offsets resolve to the out-of-range lane).
#include "conv_common.h"

namespace yl {

typedef __attribute__((address_space(3))) void lds_void_d;

// development aid (YL_DWPW_DBG & 32): per-workgroup phase timestamps of the first tile in 10 ns units (yl_debug_timestamps)
constexpr int DBG_TS_WG = 4096;
__device__ unsigned long long g_dbg_ts[DBG_TS_WG * 8];
#define DWPW_STAMP(k) \
  if ((a.dbg & 32) && tid == 0 && blockIdx.x < DBG_TS_WG && i == 0) g_dbg_ts[blockIdx.x * 8 + (k)] = wall_clock64();

template <int MT, int NT, bool PW2, int D>
__global__ __launch_bounds__(256) void dwpw_tile_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int L = 64 * NT;  // guarded positions per tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  const int G = gridDim.x;
  const int wg = xcd_remap(blockIdx.x, G);
  const int npp = a.cg >> 1;  // 16-channel blocks = staging steps per tile
  const int PL = a.t_PL;      // staged positions per plane (whole 64-position DMA pieces)
  const int ppp = a.t_ppp;    // pieces per plane
  const int tile_planes = PW2 ? max(a.cg, a.cg2) : a.cg;
  // LDS: pointwise slab | further 1x1's slab | intermediate tile [planes][L] | depthwise taps [npp][64] x 16 B |
  //      depthwise bias [C] | pointwise bias [MT * 16] | further 1x1's bias [MT * 16] | stage [D buffers][2 planes][PL] | 1 KiB scratch
  bf16x8* slab = reinterpret_cast<bf16x8*>(smem);
  bf16x8* slab2 = slab + a.ksteps * MT * 64;
  bf16x8* tile = slab2 + (PW2 ? a.ksteps2 * MT * 64 : 0);
  u32x4* dwt = reinterpret_cast<u32x4*>(tile + tile_planes * L);
  float* dwb = reinterpret_cast<float*>(dwt + npp * 64);
  float* pwb = dwb + a.Cin;
  float* pwb2 = pwb + MT * 16;
  const int stage_off = (int)(reinterpret_cast<unsigned char*>(pwb2 + MT * 16) - smem);
  const int buf_bytes = 2 * PL * 16;
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.in, (short)0, (a.dbg & 1) ? 0 : a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rnull = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, (short)0, 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, a.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)nullptr, (short)0, 0, 0x00020000);
  const unsigned lane16 = (unsigned)lane * 16u;

  // Stage planes 2 pp, 2 pp + 1 of tile t into buffer bsel: piece c = wave, wave + 4, ... of the 2 * ppp.  Every wave
  // issues exactly K = ceil(2 ppp / 4) operations per step (surplus ones through the null descriptor: no access), so
  // "step j has landed" is the same counted s_waitcnt in every wave.
  const int K = (2 * ppp + 3) >> 2;
  auto dma_step = [&](const __amdgpu_buffer_rsrc_t& r, int t, int pp, int bsel) __attribute__((always_inline)) {
    for (int k = 0; k < K; ++k) {
      const int c = wave + 4 * k;
      const bool live = c < 2 * ppp;
      const int pl = c >= ppp ? 1 : 0, pc = live ? c - pl * ppp : 0;
      const unsigned src = (unsigned)(2 * pp + pl) * a.in_plane + (unsigned)(a.in_g0 + t * L - a.in_w1 - 1 + pc * 64) * 16u;
      // (a surplus operation may still write its zeros: it gets a 1 KiB scratch piece behind the ring)
      const int dst = stage_off + (live ? bsel * buf_bytes + (pl * PL + pc * 64) * 16 : D * buf_bytes);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(live ? r : rnull, (lds_void_d*)(smem + dst), 16, (int)(lane16 + src), 0, 0, 0);
    }
  };
  // step j -> (tile, plane pair); steps past the workgroup's last tile stage nothing
  auto dma_at = [&](int jj) __attribute__((always_inline)) {
    const int ii = jj / npp, pp = jj - ii * npp, t = wg + ii * G;
    dma_step(t < a.t_ntiles ? rin : rnull, t < a.t_ntiles ? t : 0, pp, jj % D);
  };
  // all but the youngest (D - 2) * K vector-memory operations of this wave have completed (in-order retirement; output
  // stores issued since only make the wait stricter)
  auto wait_step = [&]() __attribute__((always_inline)) {
    if (K == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * 1) : "memory");
    else if (K == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * 2) : "memory");
    else if (K == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * 3) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  const int nmine = (a.t_ntiles - wg + G - 1) / G;  // > 0: grid <= tiles
  copy_slab(reinterpret_cast<const bf16x8*>(a.wpk), slab, a.ksteps * MT * 64, tid);
  if constexpr (PW2) copy_slab(reinterpret_cast<const bf16x8*>(a.wpk2), slab2, a.ksteps2 * MT * 64, tid);
  for (int i = tid; i < npp * 64; i += 256) dwt[i] = reinterpret_cast<const u32x4*>(a.dw_wpk)[i];
  for (int i = tid; i < a.Cin; i += 256) dwb[i] = a.dw_b[i];
  if (tid < MT * 16) pwb[tid] = a.bias[tid];
  if (PW2 && tid < MT * 16) pwb2[tid] = a.bias2[tid];
  // (the prologue's loads are consumed -- written to LDS -- before the first ring operation is issued: the counted
  // waits below see ring operations and output stores only)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int jj = 0; jj < D - 1; ++jj) dma_at(jj);

  // byte offset of the lane's B fragment of k-step s inside a stage buffer: k-group q = 4 s + g = (plane q / 9, tap q % 9)
  // of position n; the staged run starts W + 2 positions before the tile, so tap (kh, kw) of position p is p + kh (W + 1) + kw.
  // k-groups 18, 19 (zero filter entries) read tap 0: finite data.
  int boff[5];
#pragma unroll
  for (int s5 = 0; s5 < 5; ++s5) {
    const int q = 4 * s5 + g, ql = q < 18 ? q : 0, sel = ql >= 9, tap = ql - 9 * sel;
    boff[s5] = (sel * PL + (tap / 3) * a.in_w1 + (tap % 3) + wave * (16 * NT) + n) * 16;
  }
  const int e = n & 7;  // the lane's A row is channel n of the block: element n % 8 of its k-group
  unsigned selm[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) selm[d] = (e >> 1) == d ? ((e & 1) ? 0xffff0000u : 0x0000ffffu) : 0u;
  unsigned char* tb = reinterpret_cast<unsigned char*>(tile);

  int j = 0;  // staging step counter: buffer j & 1
  for (int i = 0; i < nmine; ++i) {
    const int t = wg + i * G;
    DWPW_STAMP(0)
    // ---- depthwise 3x3 + bias + SiLU on the matrix pipe, 16 channels per step
    for (int pp = 0; pp < npp; ++pp, ++j) {
      // this step's pieces have landed (each wave waits for its own, then the barrier); every wave has left the
      // previous step's reads, so that buffer is free for step j + D - 1.  (First step: also the prologue's LDS writes.)
      wait_step();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (pp == 0) { DWPW_STAMP(1) }
      dma_at(j + D - 1);
      if (a.dbg & 2) continue;
      const unsigned char* sb = smem + stage_off + (j % D) * buf_bytes;
      const u32x4 tw = dwt[pp * 64 + lane];  // k-steps 0..4 as bf16 halves: {s0 | s1 << 16, s2 | s3 << 16, s4, -}
      const f32x4 bv = *reinterpret_cast<const f32x4*>(dwb + pp * 16 + 4 * g);
      bf16x8 xb[NT][5];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5) xb[nt][s5] = *reinterpret_cast<const bf16x8*>(sb + boff[s5] + nt * 256);
      f32x4 dacc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) dacc[nt] = bv;
#pragma unroll
      for (int s5 = 0; s5 < 5; ++s5) {
        const unsigned w16 = (s5 & 1) ? (tw[s5 >> 1] >> 16) : (tw[s5 >> 1] & 0xffffu);
        const unsigned both = w16 | (w16 << 16);
        u32x4 wv;
#pragma unroll
        for (int d = 0; d < 4; ++d) wv[d] = both & selm[d];
        const bf16x8 wa = __builtin_bit_cast(bf16x8, wv);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) dacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb[nt][s5], dacc[nt], 0, 0, 0);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        // (one wave-uniform branch around the four SiLUs: a select per value made hipcc branch around each of them)
        if (a.dw_act) {
          dacc[nt] = silu4(dacc[nt]);
        }
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (__bf16)dacc[nt][r];
        *reinterpret_cast<bf16x4*>(tb + ((size_t)(2 * pp + (g >> 1)) * L + wave * 16 * NT + nt * 16 + n) * 16 + (g & 1) * 8) = o;
      }
    }
    DWPW_STAMP(2)

    // ---- output offsets of the wave's positions (guard positions and the ragged end store nothing)
    int opix[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int r = t * L + wave * (16 * NT) + nt * 16 + n;  // position relative to g0
      const int b = (int)__umulhi((unsigned)r, a.magic_img);
      const int rr = r - b * a.in_img;
      const int y = (int)__umulhi((unsigned)rr, a.magic_w1);
      const int x = rr - y * a.in_w1;
      opix[nt] = (b < a.t_B && y < a.t_H && x < a.t_W) ? a.in_g0 + r : -1;
    }

    // ---- 1x1 GEMM from the wave's own columns of the tile
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 bq = *reinterpret_cast<const f32x4*>(pwb + chan_in_block(MT, mt, g, 0));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = bq;
    }
    for (int ks = 0; ks < ((a.dbg & 4) ? 0 : a.ksteps); ++ks) {
      bf16x8 wa[MT], xb[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) wa[mt] = slab[(ks * MT + mt) * 64 + lane];
      // k-groups past the last plane meet zero filter entries (conv_pack_weights pads): any finite operand will do.
      // (An UNCONDITIONAL load: hipcc drops the alias metadata of a select-of-load, and SIInsertWaitcnts then puts
      // s_waitcnt vmcnt(0) -- the whole staging ring -- in front of it.)
      const int k8 = min(ks * 4 + g, a.cg - 1);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) xb[nt] = tile[k8 * L + wave * 16 * NT + nt * 16 + n];
      mma_step<MT, NT>(wa, xb, acc);
    }
    DWPW_STAMP(3)
    if constexpr (!PW2) {
      if (!(a.dbg & 8)) epilogue<MT, NT>(a, rout, rres, 0, g, opix, acc);
    } else {
      // pointwise result (SiLU, bf16: exactly what an unfused kernel would store) -> the wave's columns of the tile,
      // as planes of 8 channels: rows of tile mt are channels chan_in_block(MT, mt, g, 0..3) = half a 16-byte position
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int ch = chan_in_block(MT, mt, g, 0);
        if (ch < a.cg2 * 8) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if (a.act) {
              acc[mt][nt] = silu4(acc[mt][nt]);
            }
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (__bf16)acc[mt][nt][r];
            *reinterpret_cast<bf16x4*>(tb + ((size_t)(ch >> 3) * L + wave * 16 * NT + nt * 16 + n) * 16 + (ch & 7) * 2) = o;
          }
        }
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f32x4 bq = *reinterpret_cast<const f32x4*>(pwb2 + chan_in_block(MT, mt, g, 0));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = bq;
      }
      for (int ks = 0; ks < a.ksteps2; ++ks) {
        bf16x8 wa[MT], xb[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) wa[mt] = slab2[(ks * MT + mt) * 64 + lane];
        const int k8 = min(ks * 4 + g, a.cg2 - 1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) xb[nt] = tile[k8 * L + wave * 16 * NT + nt * 16 + n];
        mma_step<MT, NT>(wa, xb, acc);
      }
      ConvArgs a2 = a;
      a2.Cout = a.post_cout;
      a2.act = a.post_act;
      if (!(a.dbg & 8)) epilogue<MT, NT>(a2, rout, rres, 0, g, opix, acc);
    }
    DWPW_STAMP(4)
  }
  // the last steps staged nothing (null descriptor), but their LDS-DMA operations must retire before the LDS is released
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ------------------------------------------------------------------------------------ host
static unsigned magic_of_d(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull / (unsigned)d) + 1ull); }
static const size_t LDS_MAX_D = 160 * 1024;
static const int NUM_CU_D = 256;

#ifndef DW_RING_DEPTH
#define DW_RING_DEPTH 4
#endif
constexpr int DW_RING = DW_RING_DEPTH;  // staging ring depth
static int dw_PL(int NT, int W) { return (64 * NT + 2 * (W + 1) + 2 + 63) & ~63; }

static size_t dwpw_tile_lds(const ConvConfig& cfg, const ConvArgs& a, int NT) {
  const int planes = a.wpk2 && a.cg2 > cfg.cg ? a.cg2 : cfg.cg;
  return (size_t)cfg.ksteps * cfg.MT * 1024 + (a.wpk2 ? (size_t)a.ksteps2 * cfg.MT * 1024 : 0) + (size_t)planes * 64 * NT * 16 +
         (size_t)(cfg.cg / 2) * 1024 + (size_t)a.Cin * 4 + (size_t)2 * cfg.MT * 64 + (size_t)DW_RING * 2 * dw_PL(NT, a.Wo) * 16 + 1024;
}

template <int MT, int NT, bool PW2>
static int launch_dt(const ConvArgs& a, int grid, size_t lds, hipStream_t s) {
  auto kern = dwpw_tile_kernel<MT, NT, PW2, DW_RING>;
  if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), (int)((int)LDS_MAX_D))) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

template <int MT, int NT>
static int launch_dt2(const ConvArgs& a, int grid, size_t lds, hipStream_t s) {
  return a.wpk2 ? launch_dt<MT, NT, true>(a, grid, lds, s) : launch_dt<MT, NT, false>(a, grid, lds, s);
}

// 1 = handled.  `a` as dwpw_launch (conv_mfma.hip) prepared it.
int dwpw_tile_launch(const ConvConfig& cfg, ConvArgs a, int batch, hipStream_t s, int* handled) {
  *handled = 0;
  // Opt-in: measured SLOWER than dwpw_kernel on MI355X in every configuration tried (batch 32; 64 -> 80 @80x80: 39.4 us
  // against 32.7; 80 -> 80 @80x80: 49.6 against 37.4; 128 -> 80 @40x40: 24.4 against 16.4; ring depth 2 or 4, 64 or 128
  // positions per tile, 1 to 4 workgroups per CU).  The ring and the slabs leave room for 2-3 workgroups per CU, whose four
  // waves meet at a barrier every step; dwpw_kernel's 5-6 independent workgroups per CU hide more latency than the
  // staging saves in L1 traffic.  Kept for the record and for the timeline tool (tools/dwpw_timeline.py).
  const int enable = env_int("YL_DWPW_TILE", 0);
  static const int want_nt = tune_int("YL_DWPW_TILE_NT", 0);
  static const int wg_per_cu = tune_int("YL_DWPW_TILE_WPC", 0);
  if (!enable || a.pf_count || a.in_w1 != a.Wo + 1 || a.in_w1 != a.out_w1 || a.in_img != a.out_img || a.in_g0 != a.out_g0 || a.out_f32)
    return 0;
  const long long nq = (long long)batch * a.in_img;
  if (nq >= (1LL << 27)) return 0;
  const int NT = want_nt ? want_nt : cfg.NT;
  if (NT != 1 && NT != 2) return 0;
  const size_t lds = dwpw_tile_lds(cfg, a, NT);
  if (lds > LDS_MAX_D) return 0;
  const int L = 64 * NT;
  a.t_B = batch;
  a.t_H = a.Ho;
  a.t_W = a.Wo;
  a.t_PL = dw_PL(NT, a.Wo);
  a.t_ppp = a.t_PL / 64;
  a.magic_img = magic_of_d(a.in_img);
  a.magic_w1 = magic_of_d(a.in_w1);
  a.t_ntiles = (int)((nq + L - 1) / L);
  // persistent workgroups: as many per CU as the LDS holds (at most 4 = one wave per SIMD and workgroup, 16 per CU),
  // then the smallest grid that keeps the same number of rounds
  int per_cu = (int)(LDS_MAX_D / lds);
  if (per_cu > 4) per_cu = 4;
  if (wg_per_cu) per_cu = wg_per_cu;
  int grid = NUM_CU_D * per_cu;
  if (grid > a.t_ntiles) grid = a.t_ntiles;
  const int rounds = (a.t_ntiles + grid - 1) / grid;
  grid = (a.t_ntiles + rounds - 1) / rounds;
  *handled = 1;
  switch (cfg.MT * 10 + NT) {
    case 11: return launch_dt2<1, 1>(a, grid, lds, s);
    case 12: return launch_dt2<1, 2>(a, grid, lds, s);
    case 21: return launch_dt2<2, 1>(a, grid, lds, s);
    case 22: return launch_dt2<2, 2>(a, grid, lds, s);
    case 41: return launch_dt2<4, 1>(a, grid, lds, s);
    case 42: return launch_dt2<4, 2>(a, grid, lds, s);
    case 51: return launch_dt2<5, 1>(a, grid, lds, s);
    case 52: return launch_dt2<5, 2>(a, grid, lds, s);
  }
  *handled = 0;
  return 0;
}

}  // namespace yl

// development aid, not part of the documented ABI: copies the timestamps dwpw_tile_kernel recorded under YL_DWPW_DBG & 32
extern "C" __attribute__((visibility("default"))) int yl_debug_timestamps(unsigned long long* dst, int n) {
  if (n > yl::DBG_TS_WG * 8) n = yl::DBG_TS_WG * 8;
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  return hipMemcpyFromSymbol(dst, HIP_SYMBOL(yl::g_dbg_ts), (size_t)n * 8) == hipSuccess ? 0 : -2;
}
