// Device helpers shared by the MFMA conv kernels (conv_mfma.hip, conv3_tile.hip): SiLU, the output-channel
// permutation of the packed filters, buffer-addressing constants, fragment fetch and the common epilogue.
#pragma once
#include "yl_internal.h"

namespace yl {

__device__ __forceinline__ int fastdiv(int x, int d, unsigned magic) {
  return d == 1 ? x : (int)__umulhi((unsigned)x, magic);
}

// SiLU with the hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32; ~1 ulp each, far inside bf16 output
// rounding): an IEEE divide here costs ~10 VALU instructions per element and made the epilogue, not
// memory, the bottleneck of the streaming layers.
__device__ __forceinline__ float silu_f(float v) {
#ifdef YL_ABL_NO_SILU  // (development build switch, wrong results: what the activations' VALU work costs end to end)
  return v;
#else
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * v));
#endif
}
// Four SiLUs with the three full-rate steps as PACKED fp32 instructions (v_pk_mul_f32 / v_pk_add_f32: two values per lane and
// instruction; hipcc leaves adjacent scalar silu_f calls unpacked): 6 + 8 instructions per four values instead of 12 + 8, the
// same IEEE operations on the same operands -- bit-identical to silu_f.  The fused small-channel kernels are bound by
// instruction issue (NOTEBOOK section 13).
typedef __attribute__((ext_vector_type(4))) float silu_f32x4;
__device__ __forceinline__ silu_f32x4 silu4(silu_f32x4 v) {
#ifdef YL_ABL_NO_SILU
  return v;
#else
  // (the constants live in registers the compiler cannot see through: a packed instruction takes no 32-bit literal, and with a
  // visible constant hipcc falls back to four scalar multiplies / adds)
  float kc = -1.44269504088896340736f, one = 1.0f;
  asm volatile("" : "+v"(kc), "+v"(one));
  const silu_f32x4 kv = {kc, kc, kc, kc}, ov = {one, one, one, one};
  const silu_f32x4 t = v * kv;
  silu_f32x4 e;
#pragma unroll
  for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(t[i]);
  const silu_f32x4 d = e + ov;
  silu_f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = __builtin_amdgcn_rcpf(d[i]);
  return v * r;
#endif
}
__device__ __forceinline__ void silu8(float (&v)[8]) {
  const silu_f32x4 a = silu4(silu_f32x4{v[0], v[1], v[2], v[3]}), b = silu4(silu_f32x4{v[4], v[5], v[6], v[7]});
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] = a[i];
    v[4 + i] = b[i];
  }
}

// Output-channel order inside a block of MT*16 channels.  The MFMA leaves lane (pixel n, quad g)
// with rows g*4+r of every 16-row tile; the filter rows are permuted at pack time so that one lane's
// rows of two adjacent tiles are 8 CONSECUTIVE channels = one pixel of one plane: 16-byte stores,
// 256 contiguous bytes per 16 lanes.
__host__ __device__ inline int chan_in_block(int MT, int mt, int g, int r) {
  const int even = MT & ~1;
  return mt < even ? ((((mt >> 1) * 4 + g) * 2 + (mt & 1)) * 4 + r) : (even * 16 + g * 4 + r);
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
// (st_nt: bit 0 = the bf16 stores carry the hint, bit 1 = the e4m3 stores -- fp8 outputs and the e4m3 copies of bf16 outputs -- too)
// A 16-byte (8-byte) buffer store with or without the non-temporal hint (the cache-policy operand is an immediate: a wave-uniform
// branch picks the instruction).
__device__ __forceinline__ void store_b128_nt(u32x4 d, __amdgpu_buffer_rsrc_t r, unsigned off, int nt) {
  if (nt) __builtin_amdgcn_raw_buffer_store_b128(d, r, off, 0, 2);
  else __builtin_amdgcn_raw_buffer_store_b128(d, r, off, 0, 0);
}
__device__ __forceinline__ void store_b32_nt(unsigned d, __amdgpu_buffer_rsrc_t r, unsigned off, int nt) {
  if (nt) __builtin_amdgcn_raw_buffer_store_b32(d, r, off, 0, 2);
  else __builtin_amdgcn_raw_buffer_store_b32(d, r, off, 0, 0);
}
__device__ __forceinline__ void store_b64_nt(u32x2 d, __amdgpu_buffer_rsrc_t r, unsigned off, int nt) {
  if (nt) __builtin_amdgcn_raw_buffer_store_b64(d, r, off, 0, 2);
  else __builtin_amdgcn_raw_buffer_store_b64(d, r, off, 0, 0);
}
// Buffer addressing (raw_buffer_load/store with a 32-bit byte offset): an offset beyond the
// descriptor's size makes a load return zeros and a store vanish, so ragged patches and k tails need
// no branches.  Views are limited to 2^30 bytes (host-checked) so that the sums below can never wrap
// back into range: dead lane 2^31 (+/- a tap offset), k tail 2^30, both 3 * 2^30.
constexpr unsigned OOB_LANE = 0x80000000u;
constexpr unsigned OOB_K = 0x40000000u;

template <int NT>
struct Patch {
  unsigned base[NT];  // byte offset of the (stride-scaled) centre pixel in the lane's first input plane
  unsigned base2[NT]; // UP only: the same for the low-resolution source (pixel (y/2, x/2))
  int opix[NT];       // output pixel: guarded plane index (bf16) or linear NHWC index (fp32); -1 = none
};

template <int NT, bool UP = false>
__device__ __forceinline__ void patch_setup(const ConvArgs& a, int patch, int n, unsigned lane_off, Patch<NT>& P,
                                            unsigned lane_off2 = 0) {
  const int first = patch * (16 * NT);  // wave-uniform
  const int hw = a.Ho * a.Wo;
  const int b0 = first / hw;
  const int rem = first - b0 * hw;
  const int oy0 = rem / a.Wo;
  const int ox0 = rem - oy0 * a.Wo;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int t = ox0 + nt * 16 + n;
    const int q = fastdiv(t, a.Wo, a.magic_wo);
    const int ox = t - q * a.Wo;
    const int oyt = oy0 + q;
    const int qb = fastdiv(oyt, a.Ho, a.magic_ho);
    const int oy = oyt - qb * a.Ho;
    const int b = b0 + qb;
    const int gp = first + nt * 16 + n;
    const bool valid = gp < a.npix;
    const int pi = a.in_g0 + b * a.in_img + oy * a.stride * a.in_w1 + ox * a.stride;
    const int po = a.out_f32 ? gp : a.out_g0 + b * a.out_img + oy * a.out_w1 + ox;
    P.base[nt] = valid ? (unsigned)pi * 16u + lane_off : OOB_LANE;
    if (UP) {
      const int pi2 = a.in2_g0 + b * a.in2_img + (oy >> 1) * a.in2_w1 + (ox >> 1);
      P.base2[nt] = valid ? (unsigned)pi2 * 16u + lane_off2 : OOB_LANE;
    }
    P.opix[nt] = valid ? po : -1;
  }
}

template <int KS>
__device__ __forceinline__ int tap_offset(const ConvArgs& a, int tap) {  // bytes
  if (KS == 1) return 0;
  const int kh = (tap * 11) >> 5;
  return ((kh - 1) * a.in_w1 + (tap - 3 * kh - 1)) * 16;
}

// B-fragment fetch of k-step `ks` (wave-uniform).  In the UNIFORM case (Cin % 32 == 0) a k-step lies
// inside one tap: tap, channel block and byte offset are scalars (the lane's k-group plane is already
// in Patch::base) and the per-lane work is ONE add per load; otherwise (tap, channel group) are derived
// per lane.  K-steps past the end (ring refills) resolve to the out-of-range offset and return zeros.
// UP (1x1 convs reading a Concat whose first channels are a nearest-upsampled low-resolution map,
// yolo11.yaml:31-36): k-steps below ks_split read the low-resolution source itself at pixel (y/2, x/2),
// so the 4x larger upsampled copy is never written or read.
template <int KS, int NT, bool UNIFORM, bool UP = false>
__device__ __forceinline__ void fetch_b(const ConvArgs& a, __amdgpu_buffer_rsrc_t rin, int ks, int g,
                                        const Patch<NT>& P, bf16x8 (&xb)[NT],
                                        __amdgpu_buffer_rsrc_t rin2 = __amdgpu_buffer_rsrc_t()) {
  if constexpr (UP) {
    static_assert(!UP || (KS == 1 && UNIFORM), "UP is a 1x1, Cin % 32 == 0 feature");
    if (ks < a.ks_split) {
      const unsigned off2 = (unsigned)(ks * 4) * a.in2_plane;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        xb[nt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rin2, P.base2[nt] + off2, 0, 0));
      return;
    }
    ks -= a.ks_split;  // k-step within the full-resolution source
    const unsigned off1 = ks < a.ksteps - a.ks_split ? (unsigned)(ks * 4) * a.in_plane : OOB_K;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      xb[nt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rin, P.base[nt] + off1, 0, 0));
    return;
  }
  unsigned off;
  if (UNIFORM) {
    const int tap = fastdiv(ks, a.kkn, a.magic_kkn);
    const int kk = ks - tap * a.kkn;
    off = ks < a.ksteps ? (unsigned)tap_offset<KS>(a, tap) + (unsigned)(kk * 4) * a.in_plane : OOB_K;
  } else {
    const int k8 = ks * 4 + g;
    int tap, c8;
    if (KS == 3 && a.kord) {  // plane-major: the nine taps of a plane are consecutive k-groups (wave-uniform choice)
      c8 = (int)__umulhi((unsigned)k8, 477218589u);  // k8 / 9
      tap = k8 - 9 * c8;
    } else {
      tap = KS == 1 ? 0 : fastdiv(k8, a.cg, a.magic_cg);
      c8 = k8 - tap * a.cg;
    }
    off = k8 < a.ntaps * a.cg ? (unsigned)tap_offset<KS>(a, tap) + (unsigned)c8 * a.in_plane : OOB_K;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
    xb[nt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rin, P.base[nt] + off, 0, 0));
}

template <int MT, int NT>
__device__ __forceinline__ void mma_step(const bf16x8 (&wa)[MT], const bf16x8 (&xb)[NT], f32x4 (&acc)[MT][NT]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[mt], xb[nt], acc[mt][nt], 0, 0, 0);
}

// Global -> LDS copy of a packed filter slab: 8 independent 16-byte loads per thread are issued
// before the first LDS store, so the copy costs one memory round trip per 32 KB, not one per 4 KB.
__device__ __forceinline__ void copy_slab(const bf16x8* __restrict__ src, bf16x8* dst, int n16, int tid) {
  for (int i0 = tid; i0 < n16; i0 += 256 * 8) {
    bf16x8 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 256;
      if (i < n16) t[u] = src[i];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 256;
      if (i < n16) dst[i] = t[u];
    }
  }
}

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  const bf16x2 t = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, t);
}

// four floats -> four OCP e4m3 bytes, saturating (v_cvt_pk_fp8_f32 rounds to nearest even)
__device__ __forceinline__ unsigned pack_f8x4(float a, float b, float c, float d) {
  constexpr float M = 448.0f;  // largest finite e4m3
  a = __builtin_fminf(__builtin_fmaxf(a, -M), M);
  b = __builtin_fminf(__builtin_fmaxf(b, -M), M);
  c = __builtin_fminf(__builtin_fmaxf(c, -M), M);
  d = __builtin_fminf(__builtin_fmaxf(d, -M), M);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (unsigned)r;
}

// fp32 NHWC store of W consecutive channels starting at `co` (Detect level maps, head.py:64-65).
template <int W>
__device__ __forceinline__ void store_f32(const ConvArgs& a, __amdgpu_buffer_rsrc_t rout, bool ok, int pix, int co,
                                          const float (&v)[W]) {
  const bool vec = ((a.out_pitch | a.out_coff) & 3) == 0;
#pragma unroll
  for (int q = 0; q < W; q += 4) {
    const unsigned vo = (unsigned)pix * ((unsigned)a.out_pitch * 4u) + (unsigned)(a.out_coff + co + q) * 4u;
    if (vec && co + q + 4 <= a.Cout) {
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[q]), __float_as_uint(v[q + 1]),
                                                   __float_as_uint(v[q + 2]), __float_as_uint(v[q + 3])},
                                             rout, ok ? vo : OOB_LANE, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q + r]), rout,
                                              (ok && co + q + r < a.Cout) ? vo + 4u * r : OOB_LANE, 0, 0);
    }
  }
}

// The residual form of the epilogue below for bf16 planes (activated accumulators in, even MT): straight-line code in which the
// residual of tile pair p + 1 is requested before pair p's stores are issued, so the wait for it is a counted one that leaves the
// stores in flight.  (As one run-time branch per tile the compiler put `s_waitcnt vmcnt(0)` behind every residual load -- on CDNA4
// that also waits for every store before it: a load - wait - store round trip per tile; round 6, found in conv_f8_kernel first.)
template <int MT, int NT>
__device__ __forceinline__ void epilogue_res(const ConvArgs& a, __amdgpu_buffer_rsrc_t rout, __amdgpu_buffer_rsrc_t rres, int cb, int g,
                                             const int (&opix)[NT], const f32x4 (&acc)[MT][NT]) {
  static_assert(MT % 2 == 0, "tile pairs");
  unsigned pix16[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) pix16[nt] = opix[nt] >= 0 ? (unsigned)opix[nt] * 16u : OOB_LANE;
  u32x4 rnext[NT];
  auto res_load = [&](int p) __attribute__((always_inline)) {
    const int co = cb * MT * 16 + (p * 4 + g) * 8;
    const unsigned row = co < a.Cout ? (unsigned)(co >> 3) * a.res_plane : OOB_K;  // (dead pixel 2^31, dead channel 2^30: no sum is in range)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) rnext[nt] = __builtin_amdgcn_raw_buffer_load_b128(rres, pix16[nt] + row, 0, 0);
  };
  res_load(0);
#pragma unroll
  for (int p = 0; p < MT / 2; ++p) {
    u32x4 rcur[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) rcur[nt] = rnext[nt];
    if (p + 1 < MT / 2) res_load(p + 1);
    const int co = cb * MT * 16 + (p * 4 + g) * 8;
    const unsigned row = co < a.Cout ? (unsigned)(co >> 3) * a.out_plane : OOB_K;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const bf16x8 rv = __builtin_bit_cast(bf16x8, rcur[nt]);
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[2 * p][nt][r] + (float)rv[r];
        v[4 + r] = acc[2 * p + 1][nt][r] + (float)rv[4 + r];
      }
      store_b128_nt(u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])},
                    rout, pix16[nt] + row, a.st_nt);
    }
  }
}

// Epilogue (the accumulators already contain the bias): SiLU, residual, 16-byte buffer stores.
template <int MT, int NT, bool O8 = false>
__device__ __forceinline__ void epilogue(const ConvArgs& a, __amdgpu_buffer_rsrc_t rout, __amdgpu_buffer_rsrc_t rres,
                                         int cb, int g, const int (&opix)[NT], f32x4 (&acc)[MT][NT]) {
  const int cbase = cb * MT * 16;
  if (a.act && !(a.dbg & 8)) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = silu4(acc[mt][nt]);
  }
  if constexpr (!O8 && MT % 2 == 0) {
    if (a.res && !a.out_f32) {
      epilogue_res<MT, NT>(a, rout, rres, cb, g, opix, acc);
      return;
    }
  }
#pragma unroll
  for (int p = 0; p < MT / 2; ++p) {
    const int co = cbase + (p * 4 + g) * 8;
    const bool cok = co < a.Cout;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const bool ok = cok && opix[nt] >= 0;
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[2 * p][nt][r];
        v[4 + r] = acc[2 * p + 1][nt][r];
      }
      if (a.out_f32) {
        store_f32<8>(a, rout, ok, opix[nt], co, v);
        continue;
      }
      if constexpr (O8 || MT % 2 != 0) {  // (even MT without the e4m3 copy: epilogue_res above)
        if (a.res) {
          const unsigned ro = ok ? (unsigned)(co >> 3) * a.res_plane + (unsigned)opix[nt] * 16u : OOB_LANE;
          const bf16x8 rv = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rres, ro, 0, 0));
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] += (float)rv[r];
        }
      }
      const unsigned vo = ok ? (unsigned)(co >> 3) * a.out_plane + (unsigned)opix[nt] * 16u : OOB_LANE;
      store_b128_nt(u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])},
                    rout, vo, a.st_nt);
      if constexpr (O8) {  // e4m3 copy for an fp8 conv downstream (conv_f8.hip): this lane's 8 channels = half a 16-byte pixel
        const __amdgpu_buffer_rsrc_t r8 = __builtin_amdgcn_make_buffer_rsrc(a.out8, (short)0, a.out8_bytes, 0x00020000);
        const int c = co - a.out8_c0;
        const bool ok8 = ok && c >= 0 && c < a.out8_n;
        const float is = a.out8_inv_scale;
        const unsigned v8 = ok8 ? (unsigned)(c >> 4) * a.out8_plane + (unsigned)opix[nt] * 16u + (unsigned)(c & 15) : OOB_LANE;
        store_b64_nt(u32x2{pack_f8x4(v[0] * is, v[1] * is, v[2] * is, v[3] * is), pack_f8x4(v[4] * is, v[5] * is, v[6] * is, v[7] * is)}, r8, v8,
                     a.st_nt & 2);
      }
    }
  }
  if (MT & 1) {
    constexpr int mt = MT - 1;
    const int co = cbase + (MT & ~1) * 16 + g * 4;
    const bool cok = co < a.Cout;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const bool ok = cok && opix[nt] >= 0;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[mt][nt][r];
      if (a.out_f32) {
        store_f32<4>(a, rout, ok, opix[nt], co, v);
        continue;
      }
      const unsigned po = (unsigned)opix[nt] * 16u + (unsigned)(co & 7) * 2u;
      if (a.res) {
        const unsigned ro = ok ? (unsigned)(co >> 3) * a.res_plane + po : OOB_LANE;
        const bf16x4 rv = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rres, ro, 0, 0));
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
      }
      const unsigned vo = ok ? (unsigned)(co >> 3) * a.out_plane + po : OOB_LANE;
      __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])}, rout, vo, 0, 0);
    }
  }
}

// Detect's box decode in the epilogue of the conv that computes a level's 64 DFL logits (ConvArgs::dfl_out; reference
// nn/modules/head.py:95-126 _inference, block.py:51-69 DFL.forward, utils/tal.py:341-350 dist2bbox(xywh=True), head.py:105-106
// `dbox * strides`) -- on the fp32 ACCUMULATORS: the logits never pass through bf16, which was the whole 0.2-0.3 stride box error
// of the stored-logits form.  `acc` = one block of four channel tiles (64 channels, bias included) x NT pixel tiles.  By
// chan_in_block a lane (pixel n, quad g) holds in tile pair p the 8 consecutive channels (4 p + g) * 8 ..: side 2 p + (g >> 1)
// (l, t, r, b = 0..3; channel = side * 16 + bin), bins 8 (g & 1) .. + 7.  The lane pair (g, g ^ 1) shares a side: maximum over the
// 16 bins by one exchange, each lane exponentiates its 8 bins, the two partial sums / weighted sums are exchanged and added as
// (bins 0..7) + (bins 8..15) in both lanes.  Lanes g = 0, 1 then hold (l, r), lanes g = 2, 3 (t, b): centre and extent per axis
// with box_xywh's arithmetic, one exchange across 32 lanes, and lane g = 0 stores the anchor's 16 bytes; 16 lanes = 256 contiguous
// bytes.  first = index (over the batch, patch_setup's order = b * Ho * Wo + y * Wo + x) of the wave's first pixel.
// (dfl_box: one 16-pixel tile -- t[mt] = the four channel tiles' accumulators of the tile; (ax, ay) = the anchor coordinates of the
// lane's pixel, tal.py:326-338 make_anchors: (x + 0.5, y + 0.5).  The box is valid in the lanes g == 0; every lane of the wave must call.)
__device__ __forceinline__ float4 dfl_box(const f32x4 (&t)[4], int g, float ax, float ay, float stride) {
  const float j0 = (float)((g & 1) * 8);
  float d[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    float v[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = t[2 * p][r];
      v[4 + r] = t[2 * p + 1][r];
    }
    float m = v[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) m = fmaxf(m, v[j]);
    m = fmaxf(m, __shfl_xor(m, 16));
    float sum = 0.f, ws = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float e = expf(v[j] - m);
      sum += e;
      ws += e * (j0 + (float)j);
    }
    const float sum2 = __shfl_xor(sum, 16), ws2 = __shfl_xor(ws, 16);
    const bool hi = (g & 1) != 0;
    d[p] = ((hi ? ws2 : ws) + (hi ? ws : ws2)) / ((hi ? sum2 : sum) + (hi ? sum : sum2));
  }
  const float ac = (g >> 1) ? ay : ax;
  const float lo = ac - d[0], hi2 = ac + d[1];
  const float ctr = ((lo + hi2) / 2.0f) * stride, ext = (hi2 - lo) * stride;
  const float cy = __shfl_xor(ctr, 32), eh = __shfl_xor(ext, 32);
  return make_float4(ctr, cy, ext, eh);
}

template <int NT>
__device__ __forceinline__ void dfl_epilogue(const ConvArgs& a, int first, int g, int n, const f32x4 (&acc)[4][NT]) {
  const int hw = a.Ho * a.Wo;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const f32x4 t[4] = {acc[0][nt], acc[1][nt], acc[2][nt], acc[3][nt]};
    const int gp = first + nt * 16 + n;
    const int r = gp - (gp / hw) * hw, oy = r / a.Wo, ox = r - oy * a.Wo;
    const float4 bx = dfl_box(t, g, (float)ox + 0.5f, (float)oy + 0.5f, a.dfl_stride);
    if (g == 0 && gp < a.npix) *reinterpret_cast<float4*>(a.dfl_out + 4LL * gp) = bx;
  }
}

}  // namespace yl
