// 3x3 stride-1 Conv 64 -> 64 + (folded BN) bias + SiLU with the filter in registers, TWO waves per SIMD (gfx950).
//
// conv3_reg_kernel (conv3_reg.hip) keeps the whole 64 -> 64 filter (288 registers) in one wave per SIMD; its matrix
// pipe idles while that lone wave runs its epilogue (SiLU = v_exp + v_rcp at quarter rate: 7.7 of 26 us), and VALU work
// placed between a wave's own MFMAs did not overlap with them (measured there).  Two DIFFERENT waves of one SIMD do overlap
// matrix and vector work (MI355X_MICROARCH.md, "Two waves per SIMD"), but two waves have 256 registers each.  Hence:
//
//   * K is split between the two waves of a SIMD (waves w and w + 4 of a 512-thread workgroup): wave role 0 holds the
//     filter fragments of k-steps 0..8, role 1 those of k-steps 9..17 -- 144 AGPRs each, read in place by asm MFMAs
//     exactly as in conv3_reg.hip.  Both compute partial sums of the SAME 32 positions x 64 channels (a half-tile).
//   * After its 9 k-steps the HELPER of the half-tile parks its 32 accumulator registers in LDS (lane-contiguous 16-byte
//     writes) and goes straight on to the next half-tile's MFMAs; behind one workgroup barrier the FINISHER adds them to
//     its own and runs the epilogue (SiLU, bf16, stores) -- beside its partner's MFMAs.  The roles swap every half-tile,
//     so per half-tile a SIMD sees  [MFMAs of one wave || epilogue of the other]  then  [MFMAs of the other]:
//     the matrix pipe is the only serial resource.
//   * Input staging, tile walk and descriptors are conv3_reg's: one contiguous run of L + 2 (W + 1) + 2 guarded
//     positions per plane by LDS-DMA into one of two buffers, pieces of the next tile issued between this tile's MFMAs,
//     persistent workgroups (one per CU).  A workgroup tile is L = 256 positions = 4 pairs x 2 half-tiles x 32.
//
// Accumulation order: (bias + k-steps 0..8) + (k-steps 9..17) instead of one chain of 18 -- results agree with
// conv_kernel to fp32 rounding, i.e. to one bf16 ulp after the output rounding, not bit for bit
// (tests/test_gpu_kernels.py::test_conv3_pair_vs_torch_and_direct).
#include <type_traits>

#include "conv_common.h"

namespace yl {

typedef __attribute__((address_space(3))) void lds_void_p;

// AREG: the filter fragment lives in an AGPR (k-steps 0..7 of a wave's nine: 128 AGPRs); the ninth k-step's four fragments
// stay in VGPRs -- 128 + 128 is the split hipcc picks for a 256-register wave, and a fragment it cannot keep in an AGPR
// it would copy there with v_accvgpr_write right in front of the asm MFMA, a hazard nobody pads.
template <bool AREG>
__device__ __forceinline__ void pair_mfma(f32x4& acc, const u32x4& wa, const bf16x8& xb) {
  if constexpr (AREG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(wa), "v"(xb));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(xb));
}
__device__ __forceinline__ void pair_mfma_first(f32x4& acc, const u32x4& wa, const bf16x8& xb, const f32x4& c) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "a"(wa), "v"(xb), "v"(c));
}

// Cin = Cout = 64: MT = 4 channel tiles, 18 k-steps, 8 input planes; NT = 2 column tiles per half-tile.
template <bool RES>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3_pair_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int MT = 4, NT = 2, KH = 9, NPL = 8, L = 256;
  constexpr int NSTORE = (MT / 2) * NT;  // 16-byte stores per lane and half-tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pair = wave & 3, role = wave >> 2;  // waves w and w + 4 share a SIMD
  const int g = lane >> 4, n = lane & 15;
  const int G = gridDim.x;
  const int wg = xcd_remap(blockIdx.x, G);
  const __amdgpu_buffer_rsrc_t rout =
      __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (a.dbg & 2) ? 0 : a.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.res, (short)0, a.res ? a.res_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, (short)0, (a.dbg & 1) ? 0 : a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rnull = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, (short)0, 0, 0x00020000);
  const int PLB = a.t_PL * 16;  // bytes per staged plane (whole 1 KiB pieces)
  const int buf_bytes = NPL * PLB;
  unsigned char* const xch = smem + 2 * buf_bytes + pair * (MT * NT * 64 * 16);  // this pair's exchange area (8 KiB)
  const int nmine = (a.t_ntiles - wg + G - 1) / G;
  const int ppw = a.t_ppp;  // DMA pieces per wave and tile: wave w stages plane w
  const unsigned lane16 = (unsigned)lane * 16u;

  auto dma_piece = [&](const __amdgpu_buffer_rsrc_t& r, int t, int buf, int j) __attribute__((always_inline)) {
    const int src = wave * (int)a.in_plane + (a.in_g0 + t * L - a.in_w1 - 1) * 16 + j * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_p*)(smem + buf * buf_bytes + wave * PLB + j * 1024), 16,
                                             (int)(lane16 + (unsigned)src), 0, 0, 0);
  };
  for (int j = 0; j < ppw; ++j) dma_piece(rin, wg, 0, j);

  // this wave's half of the filter: k-steps role * 9 .. role * 9 + 8, every channel tile -> 144 AGPRs
  u32x4 wr[KH][MT];
  {
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.wpk) + lane + (size_t)role * KH * MT * 64;
#pragma unroll
    for (int ks = 0; ks < KH; ++ks)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) wr[ks][mt] = wsrc[(ks * MT + mt) * 64];
  }
  // C of the first MFMA: the bias for role 0, zero for role 1 (one uniform select each; no VALU write later)
  f32x4 c0[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + chan_in_block(MT, mt, g, 0));
    const float m = role == 0 ? 1.0f : 0.0f;
    c0[mt] = f32x4{b[0] * m, b[1] * m, b[2] * m, b[3] * m};
  }
  const int row_bytes = a.in_w1 * 16;
  const int half_bytes = 4 * PLB;
  // byte offset of this lane's position 0 of half-tile h inside a buffer: plane g, pair, half, column n
  const int lane_px = (g * a.t_PL + pair * 64 + n) * 16;

  // B fragments of local k-step ks (global k-step role * 9 + ks = tap * 2 + channel half)
  auto load_b = [&](const unsigned char* tb, int ks, bf16x8 (&xb)[NT]) __attribute__((always_inline)) {
    const int kg = role * KH + ks;  // wave-uniform
    const int tap = kg >> 1, c = kg & 1;
    const int dy = (tap * 11) >> 5, dx = tap - 3 * dy;
    const unsigned char* rb = tb + c * half_bytes + dy * row_bytes + dx * 16;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) xb[nt] = *reinterpret_cast<const bf16x8*>(rb + nt * 256);
  };

  auto epilogue = [&](int t, int h, f32x4 (&acc)[MT][NT]) __attribute__((always_inline)) {
    // partner's partial sums
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(xch + ((mt * NT + nt) * 64 + lane) * 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mt][nt][r] += p[r];
      }
    unsigned po[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int r = t * L + pair * 64 + h * 32 + nt * 16 + n;  // position relative to g0
      const int b = (int)__umulhi((unsigned)r, a.magic_img);
      const int rr = r - b * a.in_img;
      const int y = (int)__umulhi((unsigned)rr, a.magic_w1);
      const int x = rr - y * a.in_w1;
      po[nt] = (b < a.t_B && y < a.t_H && x < a.t_W) ? (unsigned)(a.in_g0 + r) * 16u : OOB_LANE;
    }
    if (a.dbg & 64) return;
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
          const bf16x8 rv = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                           rres, (unsigned)(p * 4 + g) * a.res_plane + po[nt], 0, 0));
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] += (float)rv[r];
        }
        store_b128_nt(u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])},
                      rout, (unsigned)(p * 4 + g) * a.out_plane + po[nt], a.st_nt);
      }
  };

  // the first tile's DMA pieces have landed and everybody knows it
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const int ntile_run = (a.dbg & 128) ? 0 : nmine;
  for (int i = 0; i < ntile_run; ++i) {
    const int t = wg + i * G;
    const int buf = i & 1;
    const __amdgpu_buffer_rsrc_t rnext = (i + 1 < nmine) ? rin : rnull;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const unsigned char* tb = smem + buf * buf_bytes + lane_px + h * 512;
      f32x4 acc[MT][NT];
      bf16x8 xb[2][NT];
      load_b(tb, 0, xb[0]);
#pragma unroll
      for (int ks = 0; ks < KH; ++ks) {
        if (ks + 1 < KH) load_b(tb, ks + 1, xb[(ks + 1) & 1]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if (ks == 0) pair_mfma_first(acc[mt][nt], wr[0][mt], xb[0][nt], c0[mt]);
            else if (ks < KH - 1) pair_mfma<true>(acc[mt][nt], wr[ks][mt], xb[ks & 1][nt]);
            else pair_mfma<false>(acc[mt][nt], wr[ks][mt], xb[ks & 1][nt]);
          }
          // the next tile's DMA pieces: one per MFMA row of the first half-tile (4 per k-step: all out within 3 k-steps at 80 x 80)
          if (h == 0) {
            const int j = ks * MT + mt;
            if (j < ppw) dma_piece(rnext, t + G, buf ^ 1, j);
          }
        }
      }
      if (h == 0)
        for (int j = KH * MT; j < ppw; ++j) dma_piece(rnext, t + G, buf ^ 1, j);
      // D is read below by LDS stores or VALU: let the matrix pipe drain (conv3_reg.hip, mfma_drain)
      asm volatile("s_nop 15");
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(acc[mt][nt]));
      const bool finisher = (h == role);  // role 0 finishes the first half-tile of every tile, role 1 the second
      if (!finisher) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4*>(xch + ((mt * NT + nt) * 64 + lane) * 16) = acc[mt][nt];
      }
      // Barrier of the half-tile: the helper's partial sums are in LDS (lgkmcnt), and -- second half -- this wave's DMA
      // pieces of the next tile have landed.  VMEM operations retire in order: role 0 issued its NSTORE stores of the first
      // half-tile AFTER the pieces and may leave them in flight, role 1 has issued nothing behind its pieces.
      if (h == 1) {
        if (role == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (finisher) epilogue(t, h, acc);
    }
  }
}

// ------------------------------------------------------------------------------------ host
static unsigned magic_of_p(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull / (unsigned)d) + 1ull); }
static const size_t LDS_MAX_P = 160 * 1024;
static int pair_PL(int W) { return (256 + 2 * (W + 1) + 2 + 63) & ~63; }
static size_t pair_lds(int W) { return (size_t)2 * 8 * pair_PL(W) * 16 + (size_t)4 * 8192; }

void conv3_pair_choose(int batch, int Cin, int Cout, int k, int stride, int Ho, int Wo, int post, ConvConfig* c) {
  c->pair = 0;
  // Opt-in (YL_CONV_PAIR=1).  Measured on MI355X (64 -> 64 @80x80, batch 32, same box): 31.2 us against 30.4 us for
  // conv3_reg_kernel and 34.5 us for conv_kernel.  Ablations: without epilogue 22.2 us, without epilogue and DMA 19.4 us
  // (the two barriers and the exchange per tile cost what the second wave per SIMD buys); the epilogue still adds 9 us:
  // the finisher's VALU stream gets the issue slots its partner's MFMAs leave (about half), and it sits on the
  // finisher's own critical path -- its next MFMAs wait for it while the helper waits at the barrier.
  const int enable = getenv("YL_CONV_PAIR") ? atoi(getenv("YL_CONV_PAIR")) : 0;
  const int min_tiles = getenv("YL_CONV_PAIR_MIN") ? atoi(getenv("YL_CONV_PAIR_MIN")) : 512;
  if (!enable || k != 3 || stride != 1 || post || Cin != 64 || Cout != 64 || c->ncb != 1 || c->MT != 4) return;
  const long long nq = (long long)batch * (Ho + 1) * (Wo + 1);
  if ((nq + 255) / 256 < min_tiles || pair_lds(Wo) > LDS_MAX_P) return;
  c->pair = 1;
  c->reg = 0;
  c->tile = 0;
  c->name = "conv3x3_pair_mfma";
}

template <bool RES>
static int launch_pair(const ConvArgs& a, int grid, size_t lds, hipStream_t s) {
  auto kern = conv3_pair_kernel<RES>;
  if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), (int)((int)LDS_MAX_P))) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

int conv3_pair_launch(const ConvConfig& cfg, ConvArgs a, int batch, hipStream_t s) {
  constexpr int L = 256;
  if (a.stride != 1 || a.out_f32 || a.in2 || a.wpk2 || a.in_w1 != a.out_w1 || a.in_img != a.out_img ||
      a.in_g0 != a.out_g0 || a.in_w1 != a.Wo + 1 || a.Cin != 64 || a.Cout != 64)
    return fail(-3, "conv3_pair: needs a plain 64 -> 64 3x3 stride-1 conv between planar tensors of one geometry");
  a.ncb = 1;
  a.cg = cfg.cg;
  a.ksteps = cfg.ksteps;
  a.t_B = batch;
  a.t_H = a.Ho;
  a.t_W = a.Wo;
  a.t_px = L + 2 * a.in_w1 + 2;
  a.t_PL = pair_PL(a.Wo);
  a.t_ppp = a.t_PL / 64;
  a.magic_img = magic_of_p(a.in_img);
  a.magic_w1 = magic_of_p(a.in_w1);
  const long long nq = (long long)batch * a.in_img;
  a.t_ntiles = (int)((nq + L - 1) / L);
  a.t_plane_px = (int)(a.in_plane / 16);
  const long long lim = 1LL << 30;
  const long long outb = (long long)8 * a.out_plane;
  const long long resb = a.res ? (long long)8 * a.res_plane : 0;
  if ((long long)cfg.cg * a.in_plane > lim || outb > lim || resb > lim)
    return fail(-3, "conv3_pair: a tensor slice exceeds 1 GiB (32-bit buffer offsets); lower the batch size");
  a.in_bytes = (unsigned)((long long)cfg.cg * a.in_plane);
  a.out_bytes = (unsigned)outb;
  a.res_bytes = (unsigned)resb;
  {
    static const int dbg = getenv("YL_CONV_DBG") ? atoi(getenv("YL_CONV_DBG")) : 0;  // ablation, dev only
    a.dbg = dbg;
  }
  const size_t lds = pair_lds(a.Wo);
  if (lds > LDS_MAX_P) return fail(-3, "conv3_pair: LDS budget exceeded");
  static const int max_grid = getenv("YL_CONV_PAIR_GRID") ? atoi(getenv("YL_CONV_PAIR_GRID")) : 256;
  const int grid = a.t_ntiles < max_grid ? a.t_ntiles : max_grid;
  return a.res ? launch_pair<true>(a, grid, lds, s) : launch_pair<false>(a, grid, lds, s);
}

}  // namespace yl
