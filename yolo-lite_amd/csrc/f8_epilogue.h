// Epilogue of the e4m3 conv kernels (conv_f8.hip conv_f8_kernel, conv3_f8t.hip conv3_f8t_kernel).
#pragma once
#include "conv_common.h"

namespace yl {

// Dequantise, bias, SiLU, residual, store of one wave's MT x NT accumulator tiles (channel block cb; lab = alpha[MT * 16], bias[MT * 16]
// of that block in LDS).
// Round 6: no wait in here covers a store.  alpha / bias come from LDS (staged once per workgroup behind the slab: the old per-row
// global loads were each followed by s_waitcnt vmcnt(0), which also waits for every store before it and for the next patch's
// prefetched fragments); the residual of row mt + 1 is requested before row mt's stores are issued, so the wait for it is a
// counted one that leaves the stores in flight (it was one load - wait - store round trip per tile, 24 per patch with
// MT x NT = 6 x 4 -- YOLO11x 1280x1280: a Bottleneck's cv2 took 57-77 us beside cv1's 31-46).  RES (0 none, 1 bf16, 2 e4m3) and
// OUT (0 e4m3, 1 bf16, 2 bf16 + e4m3 copy) as template arguments: with the kinds as run-time branches the compiler's wait-count
// pass gives up at the joins and falls back to vmcnt(0) everywhere (and the kernel needs 13 more registers); -1 = read ConvArgs.
// Addresses: byte offset = pixel * 16 + [lane's plane and channel offset] + [row mt: 2 (bf16) or 1 (e4m3) planes per 16
// channels]; a dead pixel is OOB_LANE, which no sum of the other terms brings back into range (conv_common.h).  Cout is a
// multiple of MT * 16 here (conv_f8_choose), so every channel of the block exists.
template <int MT, int NT, int RES, int OUT>
__device__ __forceinline__ void f8_epilogue(const ConvArgs& a, const f32x4 (&acc)[MT][NT], const int (&opix)[NT], int cb, int g,
                                            const float* lab, __amdgpu_buffer_rsrc_t rres, __amdgpu_buffer_rsrc_t rout) {
  const bool has_res = RES < 0 ? a.res != nullptr : RES != 0;
  const bool res_f8 = RES < 0 ? a.res_f8 != 0 : RES == 2;
  const bool out_f8 = OUT < 0 ? a.out_f8 != 0 : OUT == 0;
  const bool out8 = OUT < 0 ? a.out8 != nullptr : OUT == 2;
  // (g through an empty asm: the per-row lane offsets below are patch-invariant, and hoisted out of the kernel's patch loop they
  // are a dozen registers that the k-loop's allocation spills -- every reload in here would be a vmcnt(0) again)
  asm volatile("" : "+v"(g));
  unsigned pix16[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) pix16[nt] = opix[nt] >= 0 ? (unsigned)opix[nt] * 16u : OOB_LANE;
  const unsigned lane_bf = (unsigned)(g >> 1), sub_bf = (unsigned)(g & 1) * 8u, sub_f8 = (unsigned)g * 4u;  // plane within the 16 channels, bytes inside the pixel
  const unsigned res_lane = res_f8 ? (unsigned)(cb * MT) * a.res_plane + sub_f8 : ((unsigned)(cb * MT * 2) + lane_bf) * a.res_plane + sub_bf;
  const unsigned res_row = res_f8 ? a.res_plane : 2u * a.res_plane;
  u32x2 rnext[NT];
  auto res_load = [&](int mt) __attribute__((always_inline)) {
    const unsigned row = res_lane + (unsigned)mt * res_row;
    if (res_f8) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) rnext[nt] = u32x2{__builtin_amdgcn_raw_buffer_load_b32(rres, pix16[nt] + row, 0, 0), 0u};
    } else {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) rnext[nt] = __builtin_amdgcn_raw_buffer_load_b64(rres, pix16[nt] + row, 0, 0);
    }
  };
  if (has_res) res_load(0);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const f32x4 alpha_r = *reinterpret_cast<const f32x4*>(lab + mt * 16 + g * 4);
    const f32x4 bias_r = *reinterpret_cast<const f32x4*>(lab + MT * 16 + mt * 16 + g * 4);
    u32x2 rcur[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) rcur[nt] = rnext[nt];
    if (has_res && mt + 1 < MT) res_load(mt + 1);
    float v[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[nt][r] = __builtin_fmaf(acc[mt][nt][r], alpha_r[r], bias_r[r]);
    if (a.act) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[nt][r] = silu_f(v[nt][r]);
    }
    if (has_res) {
      if (res_f8) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int rv = (int)rcur[nt][0];
          v[nt][0] += a.res_scale * __builtin_amdgcn_cvt_f32_fp8(rv, 0);
          v[nt][1] += a.res_scale * __builtin_amdgcn_cvt_f32_fp8(rv, 1);
          v[nt][2] += a.res_scale * __builtin_amdgcn_cvt_f32_fp8(rv, 2);
          v[nt][3] += a.res_scale * __builtin_amdgcn_cvt_f32_fp8(rv, 3);
        }
      } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const bf16x4 rv = __builtin_bit_cast(bf16x4, rcur[nt]);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[nt][r] += (float)rv[r];
        }
      }
    }
    if (out_f8) {
      const float is = a.out_inv_scale;
      const unsigned row = (unsigned)(cb * MT + mt) * a.out_plane + sub_f8;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        store_b32_nt(pack_f8x4(v[nt][0] * is, v[nt][1] * is, v[nt][2] * is, v[nt][3] * is), rout, pix16[nt] + row, a.st_nt & 2);
    } else {
      const unsigned row = ((unsigned)((cb * MT + mt) * 2) + lane_bf) * a.out_plane + sub_bf;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        store_b64_nt(u32x2{pack_bf16x2(v[nt][0], v[nt][1]), pack_bf16x2(v[nt][2], v[nt][3])}, rout, pix16[nt] + row, a.st_nt);
      if (out8) {  // e4m3 copy of the bf16 result for the next fp8 conv (channels [out8_c0, out8_c0 + out8_n) of this conv's output)
        const __amdgpu_buffer_rsrc_t r8 = __builtin_amdgcn_make_buffer_rsrc(a.out8, (short)0, a.out8_bytes, 0x00020000);
        const int c = (cb * MT + mt) * 16 + g * 4 - a.out8_c0;
        const float is = a.out8_inv_scale;
        const unsigned row8 = (c >= 0 && c < a.out8_n) ? (unsigned)(c >> 4) * a.out8_plane + (unsigned)(c & 15) : OOB_LANE;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          store_b32_nt(pack_f8x4(v[nt][0] * is, v[nt][1] * is, v[nt][2] * is, v[nt][3] * is), r8,
                       (pix16[nt] | row8) < OOB_LANE ? pix16[nt] + row8 : OOB_LANE, a.st_nt & 2);
      }
    }
  }
}

// The kinds the programs use as compiled forms, anything else through the run-time form.
template <int MT, int NT>
__device__ __forceinline__ void f8_epilogue_any(const ConvArgs& a, const f32x4 (&acc)[MT][NT], const int (&opix)[NT], int cb, int g,
                                                const float* lab, __amdgpu_buffer_rsrc_t rres, __amdgpu_buffer_rsrc_t rout) {
  const int rk = !a.res ? 0 : (a.res_f8 ? 2 : 1), ok = a.out_f8 ? 0 : (a.out8 ? 2 : 1);
  if (rk == 0 && ok == 0) f8_epilogue<MT, NT, 0, 0>(a, acc, opix, cb, g, lab, rres, rout);
  else if (rk == 1 && ok == 1) f8_epilogue<MT, NT, 1, 1>(a, acc, opix, cb, g, lab, rres, rout);
  else if (rk == 1 && ok == 2) f8_epilogue<MT, NT, 1, 2>(a, acc, opix, cb, g, lab, rres, rout);
  else if (rk == 0 && ok == 1) f8_epilogue<MT, NT, 0, 1>(a, acc, opix, cb, g, lab, rres, rout);
  else if (rk == 0 && ok == 2) f8_epilogue<MT, NT, 0, 2>(a, acc, opix, cb, g, lab, rres, rout);
  else f8_epilogue<MT, NT, -1, -1>(a, acc, opix, cb, g, lab, rres, rout);
}

}  // namespace yl
