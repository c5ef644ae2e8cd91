// Layers 0 and 1 of yolo11.yaml in ONE launch: Conv(3 -> c0, 3x3, s2) + SiLU, then Conv(c0 -> c1, 3x3, s2) +
// SiLU [+ the 1x1 conv fused behind it], reference nn/modules/conv.py:35-53 twice, fed by the predictor's input
// conversion (engine/predictor.py:82-84).
//
// Why: the stem's output (c0 = 16 channels at half resolution, 105 MB at batch 32 x 640^2) is the largest tensor
// of the network and has exactly one reader.  stem_kernel + conv_kernel write and re-read it through HBM
// (0.21 GB of the ~2.3 GB a batch moves); here it only ever exists as a bf16 tile in LDS.
//
// One workgroup (256 threads) = TR x TC = 5 x 32 output pixels of layer 1 for one image:
//   1. the 23 x 132 input window (3 channels) is read with aligned dword buffer loads (zeros outside the image),
//      converted (/255 for uint8, true division) and rounded to bf16 into LDS;
//   2. the 11 x 65 stem outputs the tile needs are computed from it exactly as stem_kernel does (MFMA with K = 27
//      taps padded to 32, im2col gathered from LDS), SiLU'd, rounded to bf16 and parked in LDS as C8 planes; stem
//      positions outside the stem map (layer 1's zero padding) are written as zeros;
//   3. layer 1 runs as conv_kernel would -- same packed filters, same k order (tap-major, 8-channel groups), same
//      epilogue and follow-up 1x1 -- with its B fragments read from the LDS planes (stride-2 pixels).
// Same operands, same MFMA sequence: bit-identical to the two-launch path (tests/test_gpu_net.py).
// Recompute: 11 stem rows per 5 output rows and 65 columns per 32 (1.12x of a 0.9 GFLOP layer).
#include <type_traits>

#include "conv_common.h"

#ifndef YL_STEM2_ABL
#define YL_STEM2_ABL 0  // development build switch
#endif
namespace yl {

#ifndef YL_S2_TR
#define YL_S2_TR 5  // development build switch: rows of the layer-1 output tile
#endif
constexpr int S2_TR = YL_S2_TR, S2_TC = 32;                 // layer-1 output tile (round 4: 2 x 80 before -- the squarer tile recomputes
                                                     // 715 stem pixels per 160 outputs instead of 805, and a window row is four
                                                     // 16-pixel tiles + one column: one tile column per wave)
constexpr int S2_SR = 2 * S2_TR + 1;                 // stem rows (11)
constexpr int S2_SC = 2 * S2_TC + 1;                 // stem columns (65)
constexpr int S2_SP = S2_SC + 1;                     // stem tile pitch in pixels
constexpr int S2_IR = 2 * S2_SR + 1;                 // input rows (23)
constexpr int S2_IC = 2 * S2_SC + 2;                 // input columns incl. one alignment column (132)
constexpr int S2_IP = 160;                           // input tile pitch in elements (>= S2_IC; 32 mod 64: rows alternate LDS bank halves)

size_t stem2_lds_bytes(int C0) {
  return (size_t)3 * S2_IR * S2_IP * 2 + (size_t)(C0 / 8) * S2_SR * S2_SP * 16;
}

template <typename T, int MT, bool POST, int CG0>
// (waves_per_eu(2): at most 256 registers per lane, so the MFMA results are VGPRs -- without it hipcc parks them in AGPRs
// and copies every one back with v_accvgpr_read before the SiLU)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void stem2_kernel(Stem2Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s2mem[];
  __bf16* img = reinterpret_cast<__bf16*>(s2mem);                          // [3][S2_IR][S2_IP]
  unsigned char* stile = s2mem + (size_t)3 * S2_IR * S2_IP * 2;            // [C0/8][S2_SR][S2_SP] x 16 B
  constexpr int VW = 4 / (int)sizeof(T);                                   // elements per dword
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: tile rows / columns below stay in SGPRs
  const int n = lane & 15, g = lane >> 4;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tx = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int ty = bid % a.tiles_y;
  const int b = bid / a.tiles_y;
  const int y0 = ty * S2_TR, x0 = tx * S2_TC;
#define S2_STAMP(id) YL_STAMP(a.stamps, 4, wave, lane, id)
  S2_STAMP(0);
  if (a.zero_n && blockIdx.x == 0)  // the step's NMS candidate counters (Stem2Args::zero_ptr)
    for (int i = tid; i < a.zero_n; i += 256) a.zero_ptr[i] = 0;

  // ---- 1. input window -> LDS (bf16).  Local (ir, ic) <-> image (4*y0 - 3 + ir, 4*x0 - 4 + ic).
  {
    const T* in = reinterpret_cast<const T*>(a.in) + (long long)b * 3 * a.H * a.W;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)in, (short)0, (int)((long long)3 * a.H * a.W * (long long)sizeof(T)), 0x00020000);
    constexpr int DW = S2_IC / VW;  // dwords per row
    constexpr int TOTAL = 3 * S2_IR * DW;
    if constexpr (sizeof(T) == 1 && 6 * DW <= 256 && !(YL_STEM2_ABL & 2)) {
      // uint8 (the predictor's input): a thread = one dword column of one channel, one half of its rows, all of them in flight at
      // once -- the (row, column) arithmetic once per thread instead of twice per dword (the kernel is bound by instruction
      // issue, and the generic loop below spent ~45 instructions per dword, a fifth of the kernel)
      constexpr int RH = (S2_IR + 1) / 2;  // rows per half
      const int col = tid % (3 * DW), half = tid / (3 * DW);
      const int ch = col / DW, d = col - ch * DW;
      const int ix = 4 * x0 - 4 + d * 4;
      const bool mine = tid < 6 * DW;
      const bool okx = mine && ix >= 0 && ix + 4 <= a.W;
      const int ir0 = half * RH, iy0 = 4 * y0 - 3 + ir0;
      unsigned v[RH];
#pragma unroll
      for (int i = 0; i < RH; ++i) {
        const int iy = iy0 + i;
        v[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, (okx && ir0 + i < S2_IR && (unsigned)iy < (unsigned)a.H) ? (unsigned)((ch * a.H + iy) * a.W + ix) : 0x80000000u, 0, 0);
      }
      if (mine) {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
        const float k = 1.0f / 255.0f;  // (see below: exact after the bf16 rounding)
        __bf16* dst = img + (ch * S2_IR + ir0) * S2_IP + d * 4;
#pragma unroll
        for (int i = 0; i < RH; ++i) {
          const bf16x4_t o = {(__bf16)((float)(v[i] & 0xffu) * k), (__bf16)((float)((v[i] >> 8) & 0xffu) * k),
                              (__bf16)((float)((v[i] >> 16) & 0xffu) * k), (__bf16)((float)(v[i] >> 24) * k)};
          if (ir0 + i < S2_IR) *reinterpret_cast<bf16x4_t*>(dst + i * S2_IP) = o;
        }
      }
    } else {
    // 12 loads in flight per thread: the whole uint8 window (2673 dwords) is ONE memory round trip
    constexpr int CH = 12;
    for (int i0 = tid; i0 < TOTAL; i0 += 256 * CH) {
      unsigned v[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int idx = i0 + u * 256;
        const int rowi = idx / DW, d = idx - rowi * DW;
        const int ch = rowi / S2_IR, ir = rowi - ch * S2_IR;
        const int iy = 4 * y0 - 3 + ir, ix = 4 * x0 - 4 + d * VW;
        const bool ok = idx < TOTAL && (unsigned)iy < (unsigned)a.H && ix >= 0 && ix + VW <= a.W;
        const unsigned off = ok ? (unsigned)(((ch * a.H + iy) * a.W + ix) * (int)sizeof(T)) : 0x80000000u;
        v[u] = __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int idx = i0 + u * 256;
        if (idx < TOTAL) {
          const int rowi = idx / DW, d = idx - rowi * DW;
          __bf16* dst = img + rowi * S2_IP + d * VW;
          if constexpr ((YL_STEM2_ABL & 2) != 0) {  // (timing ablation: no conversion)
            *reinterpret_cast<unsigned*>(dst) = v[u];
            *reinterpret_cast<unsigned*>(dst + 2) = v[u];
          } else if constexpr (sizeof(T) == 1) {
            // byte / 255 (predictor.py:84), rounded to bf16 at once.  A true fp32 division costs ~12 VALU
            // instructions per pixel and made this kernel VALU bound; byte * (1 / 255.f) differs from it in the last
            // fp32 bit for 126 of the 256 byte values, but NEVER after the bf16 rounding
            // (tests/test_host_api.py::test_uint8_scale_by_reciprocal_is_exact_in_bf16 checks all 256).
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
            const float k = 1.0f / 255.0f;
            const bf16x4_t o = {(__bf16)((float)(v[u] & 0xffu) * k), (__bf16)((float)((v[u] >> 8) & 0xffu) * k),
                                (__bf16)((float)((v[u] >> 16) & 0xffu) * k), (__bf16)((float)(v[u] >> 24) * k)};
            *reinterpret_cast<bf16x4_t*>(dst) = o;
          } else if constexpr (sizeof(T) == 2) {
            typedef __attribute__((ext_vector_type(2))) _Float16 h2;
            const h2 h = __builtin_bit_cast(h2, v[u]);
            dst[0] = (__bf16)(float)h[0];
            dst[1] = (__bf16)(float)h[1];
          } else {
            dst[0] = (__bf16)__uint_as_float(v[u]);
          }
        }
      }
    }
    }
  }
  S2_STAMP(1);
  __syncthreads();
  S2_STAMP(2);

  // ---- 2. stem outputs of the window -> LDS planes.  Local (r, c) <-> stem pixel (2*y0 - 1 + r, 2*x0 - 1 + c);
  //         it reads input rows 2r + kh and columns 2c + kw + 1 of the window.
  // The kernel is bound by instruction ISSUE (in-kernel stamps, NOTEBOOK section 13: 2 100 instructions per wave, half of them
  // here, 75 per 16-pixel tile for ONE MFMA -- and SiLU's two transcendentals are quarter rate), so this phase counts
  // instructions: read pointers built once with the tile as an immediate offset (they were rebuilt, shifts included, for every
  // tile), the pixels of the window's last column as ONE tile, the zero-padding test only for tiles at the stem map's edge.
  {
    // LDS byte offset of tap k = (ch*3 + kh)*3 + kw relative to the pixel's (2r, 2c) corner; taps >= 27 (zero
    // filter values) read tap 0's element, which is initialised and finite
    unsigned koffb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = g * 8 + j;
      const int kk = k < 27 ? k : 0;
      const int ch = kk / 9, kh = (kk - ch * 9) / 3, kw = kk - ch * 9 - kh * 3;
      koffb[j] = (unsigned)(((ch * S2_IR + kh) * S2_IP + kw + 1) * 2);
    }
    constexpr int TPRF = S2_SC / 16;          // full 16-pixel tiles per window row
    static_assert(S2_SC == TPRF * 16 + 1 && S2_SR <= 16 && TPRF == 4, "four tile columns = four waves, + one column with a lane per row");
    constexpr int ROWB = 2 * S2_IP * 2;       // bytes between the input rows of consecutive stem rows
    const unsigned char* const imgb = reinterpret_cast<const unsigned char*>(img);
    // Tile (r, ct) = window row r, columns 16 ct .. 16 ct + 15; wave w owns tile column ct = w.  Its eight read pointers are built
    // ONCE; the row is an immediate offset of the LDS instructions (the loop over the rows is unrolled).
    const unsigned char* rp[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) rp[j] = imgb + koffb[j] + 4u * (unsigned)n + 64u * (unsigned)wave;
    const int sx0 = 2 * x0 - 1 + wave * 16;
    const bool col_in = sx0 >= 0 && sx0 + 15 < a.W0;  // scalar: no pixel of this tile column is layer 1's zero padding
    // (round 5) The 16-channel blocks of the stem (one for YOLO11n, two for YOLO11s) share a tile's im2col fragment: it is gathered
    // ONCE per tile and feeds one MFMA per block -- the gather (8 ds_read_u16 + 4 v_perm) is the larger part of a tile's instructions.
    constexpr int NMB = CG0 / 2;
    bf16x8 af[NMB];
    f32x4 bv[NMB];
    unsigned char* sbase[NMB];
    unsigned char* wp[NMB];
#pragma unroll
    for (int mbi = 0; mbi < NMB; ++mbi) {
      const int mb = 16 * mbi;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = g * 8 + j;
        af[mbi][j] = (__bf16)(k < 27 ? a.w0[(mb + n) * 27 + k] : 0.f);
      }
      bv[mbi] = *reinterpret_cast<const f32x4*>(a.b0 + mb + g * 4);
      const int ch = mb + g * 4;
      sbase[mbi] = stile + (size_t)(ch >> 3) * S2_SR * S2_SP * 16 + (ch & 7) * 2;
      wp[mbi] = sbase[mbi] + (size_t)((wave * 16 + n) * 16);
    }
    {
      // (the SiLU flag is a compile-time constant inside: left as a runtime test of a.act0, hipcc branched around every
      // single SiLU and shuttled the accumulators through AGPRs between the branches)
      auto stem_tiles = [&](auto act_tag) __attribute__((always_inline)) {
        constexpr bool ACT0 = decltype(act_tag)::value;
        const bf16x4 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        // two rows per trip: both gathers are requested before the first MFMA, so the LDS latency and the MFMA -> SiLU
        // dependency of one tile hide behind the other
#pragma unroll
        for (int r0 = 0; r0 < S2_SR; r0 += 2) {
          constexpr int NU = 2;
          bf16x8 bf[NU];
#pragma unroll
          for (int u = 0; u < NU; ++u) {
            const int r = r0 + u < S2_SR ? r0 + u : S2_SR - 1;  // (a missing second row repeats the last one, nothing stored)
#if YL_STEM2_ABL & 1  // (timing ablation, wrong results: one 16-byte read instead of the eight-element im2col gather)
            bf[u] = *reinterpret_cast<const bf16x8*>(imgb + ((r * ROWB + 64u * wave + 4u * n) & ~15u));
#else
#pragma unroll
            for (int j = 0; j < 8; ++j) bf[u][j] = *reinterpret_cast<const __bf16*>(rp[j] + r * ROWB);
#endif
          }
#pragma unroll
          for (int mbi = 0; mbi < NMB; ++mbi) {
            f32x4 acc[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mbi], bf[u], bv[mbi], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
              const int r = r0 + u;
              if (r >= S2_SR) continue;
              bf16x4 o;
              const f32x4 sv = (ACT0 && !(YL_STEM2_ABL & 4)) ? silu4(acc[u]) : acc[u];
#pragma unroll
              for (int q = 0; q < 4; ++q) o[q] = (__bf16)sv[q];
              const bool row_in = (unsigned)(2 * y0 - 1 + r) < (unsigned)a.H0;  // scalar
              if (!(row_in && col_in)) {  // (scalar branch) stem positions outside the stem map are layer 1's zero padding
                if (!(row_in && (unsigned)(sx0 + n) < (unsigned)a.W0)) o = z;
              }
              *reinterpret_cast<bf16x4*>(wp[mbi] + r * (S2_SP * 16)) = o;
            }
          }
        }
        // the window's last column (c = S2_SC - 1), one lane per row: a tile of its own
        if (wave == 3) {
          const int r = min(n, S2_SR - 1), c = S2_SC - 1;
          const unsigned char* corner = imgb + (2 * r * S2_IP + 2 * c) * 2;
          bf16x8 bl;
#pragma unroll
          for (int j = 0; j < 8; ++j) bl[j] = *reinterpret_cast<const __bf16*>(corner + koffb[j]);
          const bool inside = (unsigned)(2 * y0 - 1 + r) < (unsigned)a.H0 && (unsigned)(2 * x0 - 1 + c) < (unsigned)a.W0;
#pragma unroll
          for (int mbi = 0; mbi < NMB; ++mbi) {
            const f32x4 al = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mbi], bl, bv[mbi], 0, 0, 0);
            bf16x4 o;
            const f32x4 sl = (ACT0 && !(YL_STEM2_ABL & 4)) ? silu4(al) : al;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = (__bf16)sl[q];
            if (n < S2_SR) *reinterpret_cast<bf16x4*>(sbase[mbi] + (size_t)(r * S2_SP + c) * 16) = inside ? o : z;
          }
        }
      };
      if (a.act0) stem_tiles(std::true_type{});
      else stem_tiles(std::false_type{});
    }
  }
  S2_STAMP(3);
  __syncthreads();
  S2_STAMP(4);

  // ---- 3. layer 1 from the LDS planes: 16-pixel tiles aligned to the output rows (TC / 16 per row)
  constexpr int cg0 = CG0;                 // planes of the stem's output (2: 16 channels, YOLO11n; 4: 32, YOLO11s)
  constexpr int KS = (9 * cg0 + 3) / 4;    // k-steps of the second conv (5 or 9)
  bf16x8 wa[KS][MT];  // the whole filter in registers
  int boff[KS];       // byte offset of this lane's B fragment of k-step ks relative to the tile's corner
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) wa[ks][mt] = reinterpret_cast<const bf16x8*>(a.wpk)[(ks * MT + mt) * 64 + lane];
    const int k8 = ks * 4 + g;
    const int tap = min(k8 / cg0, 8);  // k groups past the last tap have zero filters: any finite data will do
    const int c8 = k8 - (k8 / cg0) * cg0;
    const int kh = tap / 3, kw = tap - kh * 3;
    boff[ks] = ((c8 * S2_SR + kh) * S2_SP + kw + 2 * n) * 16;
  }
  f32x4 bias_r[MT], bias2_r[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    bias_r[mt] = *reinterpret_cast<const f32x4*>(a.bias + chan_in_block(MT, mt, g, 0));
    if constexpr (POST) bias2_r[mt] = *reinterpret_cast<const f32x4*>(a.bias2 + chan_in_block(MT, mt, g, 0));
  }
  bf16x8 w2[MT / 2][MT];
  if constexpr (POST) {
#pragma unroll
    for (int p = 0; p < MT / 2; ++p)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) w2[p][mt] = reinterpret_cast<const bf16x8*>(a.wpk2)[(p * MT + mt) * 64 + lane];
  }
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out.base, (short)0, a.out_bytes, 0x00020000);
  const unsigned out_plane = (unsigned)(a.out.plane * 2);
  const int act_last = POST ? a.post_act : a.act;
  constexpr int TPR1 = S2_TC / 16;
  S2_STAMP(5);
  for (int t = wave; t < S2_TR * TPR1; t += 4) {
    const int ly = t / TPR1, ct = t - ly * TPR1;  // scalar
    const int oy = y0 + ly, ox = x0 + ct * 16 + n;
    const unsigned char* corner = stile + (size_t)(2 * ly * S2_SP + 2 * ct * 16) * 16;
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = bias_r[mt];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 xb = *reinterpret_cast<const bf16x8*>(corner + boff[ks]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks][mt], xb, acc[mt], 0, 0, 0);
    }
    if constexpr (POST) {
      // the follow-up 1x1 on the packed accumulators, exactly as conv_kernel<..., POST> does
      bf16x8 xb2[MT / 2];
#pragma unroll
      for (int p = 0; p < MT / 2; ++p) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[2 * p][r];
          v[4 + r] = acc[2 * p + 1][r];
        }
        if (a.act && !(YL_STEM2_ABL & 8)) {
          const f32x4 s0 = silu4(acc[2 * p]), s1 = silu4(acc[2 * p + 1]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = s0[r];
            v[4 + r] = s1[r];
          }
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) xb2[p][r] = (__bf16)v[r];
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt] = bias2_r[mt];
#pragma unroll
      for (int p = 0; p < MT / 2; ++p)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[p][mt], xb2[p], acc[mt], 0, 0, 0);
    }
    // epilogue as conv_common.h's (SiLU, bf16 pack, one 16-byte pixel of plane p*4+g per lane), branch-free
    const unsigned po = (oy < a.H1 && ox < a.W1) ? (unsigned)(a.out.g0 + b * a.out.img + oy * a.out.W1 + ox) * 16u : OOB_LANE;
    const int cout_last = POST ? a.post_cout : a.Cout;
#pragma unroll
    for (int p = 0; p < MT / 2; ++p) {
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[2 * p][r];
        v[4 + r] = acc[2 * p + 1][r];
      }
      if (act_last && !(YL_STEM2_ABL & 8)) {
        const f32x4 s0 = silu4(acc[2 * p]), s1 = silu4(acc[2 * p + 1]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = s0[r];
          v[4 + r] = s1[r];
        }
      }
      const int co = (p * 4 + g) * 8;
      store_b128_nt(u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])},
                    rout, co < cout_last ? (unsigned)(p * 4 + g) * out_plane + po : OOB_LANE, a.st_nt);
    }
  }
  S2_STAMP(6);
#undef S2_STAMP
}

template <typename T, int CG0>
static int stem2_launch_c(const Stem2Args& a, int MT, size_t lds, hipStream_t s) {
  const dim3 grid((unsigned)(a.B * a.tiles_y * a.tiles_x));
  const bool post = a.wpk2 != nullptr;
  auto go = [&](auto kern) {
    if (lds > 64 * 1024)
      if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    YL_HIP(hipGetLastError());
    return 0;
  };
  if (MT == 2 && post) return go(stem2_kernel<T, 2, true, CG0>);
  if (MT == 2) return go(stem2_kernel<T, 2, false, CG0>);
  if (MT == 4 && post) return go(stem2_kernel<T, 4, true, CG0>);
  if (MT == 4) return go(stem2_kernel<T, 4, false, CG0>);
  return fail(-3, "stem2: the second conv needs 32 or 64 output channels");
}

template <typename T>
static int stem2_launch_t(const Stem2Args& a, int MT, size_t lds, hipStream_t s) {
  return a.C0 == 16 ? stem2_launch_c<T, 2>(a, MT, lds, s) : stem2_launch_c<T, 4>(a, MT, lds, s);
}

int stem2_launch(const Stem2Args& a0, int MT, int in_dtype, hipStream_t s) {
  Stem2Args a = a0;
  a.stamps = g_stamp_buf;
  a.st_nt = (a.cacheable || !tune_int("YL_ST_NT_FUSED", 1)) ? 0 : st_nt_policy(a.out_bytes);
  if ((a.C0 != 16 && a.C0 != 32) || a.ksteps != (9 * (a.C0 / 8) + 3) / 4)
    return fail(-3, "stem2: the stem must have 16 or 32 output channels");
  const int esz = in_dtype == YL_F32 ? 4 : in_dtype == YL_F16 ? 2 : 1;
  if ((a.W * esz) % 4) return fail(-3, "stem2: input rows must be a multiple of 4 bytes (set YL_FUSE_STEM=0)");
  if ((long long)3 * a.H * a.W * esz >= (1LL << 30)) return fail(-3, "stem2: one input image exceeds 1 GiB");
  a.tiles_x = (a.W1 + S2_TC - 1) / S2_TC;
  a.tiles_y = (a.H1 + S2_TR - 1) / S2_TR;
  const size_t lds = stem2_lds_bytes(a.C0);
  if (in_dtype == YL_F32) return stem2_launch_t<float>(a, MT, lds, s);
  if (in_dtype == YL_F16) return stem2_launch_t<_Float16>(a, MT, lds, s);
  if (in_dtype == YL_U8) return stem2_launch_t<uint8_t>(a, MT, lds, s);
  return fail(-3, "stem2: unsupported input dtype");
}

}  // namespace yl
