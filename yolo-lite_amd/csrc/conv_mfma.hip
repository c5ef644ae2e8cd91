// Fused Conv2d + (folded BN) bias + SiLU [+ residual] as an MFMA implicit GEMM for gfx950.
//
// Replaces reference nn/modules/conv.py:35-53 (Conv.forward = act(bn(conv(x)))) with the BN folded
// per utils/torch_utils.py:182-209, and the Bottleneck shortcut add of nn/modules/block.py:341-343.
// Reading a channel slice of a wider NHWC buffer and writing into a channel slice implement
// torch.chunk / torch.cat of C2f/C3/SPPF/C2PSA/Concat (block.py:231-235 etc.) without copies.
//
// GEMM mapping (v_mfma_f32_16x16x32_bf16, wave64):
//   A (16 x 32)  = packed weights, row = output channel, k = (tap, cin) in groups of 8
//   B (32 x 16)  = activations,    col = output pixel,   read from an LDS halo tile [pixel][cch]
//   D (16 x 16)  : lane holds 4 consecutive output channels of ONE pixel -> 8-byte NHWC stores.
// One workgroup = 4 waves; a tile = TH x TW output pixels (4*NT*16) x MT*16 output channels.
//
// Two kernels share the epilogue:
//   conv_direct_kernel  the block's whole filter slab is staged into LDS once; activations are
//                       fetched as MFMA fragments straight from global memory / L1 (no halo
//                       staging, no barrier in the pixel loop).  Used whenever the slab fits.
//   conv_tile_kernel    fallback for huge filters: one LDS halo tile per block, the filter chunk
//                       is restaged with every channel chunk.
// (A persistent weight-stationary LDS-halo variant was measured slower than both at batch 32 --
//  one workgroup per CU cannot hide HBM latency -- and was dropped; see DESIGN.md.)
#include "yl_internal.h"

namespace yl {

constexpr int MAXPRE = 12;  // 16-byte prefetch registers per thread (halo stage)

__device__ __forceinline__ int fastdiv(int x, int d, unsigned magic) {
  return d == 1 ? x : (int)__umulhi((unsigned)x, magic);
}

// SiLU with the hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32; ~1 ulp each, far inside bf16 output
// rounding): an IEEE divide here costs ~10 VALU instructions per element and made the epilogue, not
// memory, the bottleneck of the streaming layers (measured: 32->32 1x1 @160^2 43 us -> see DESIGN.md).
__device__ __forceinline__ float silu_f(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * v));
}

// Per-lane output pixels of the wave's NT pixel tiles: LDS halo offset and output pixel index.
template <int NT>
__device__ __forceinline__ void lane_pixels(const ConvArgs& a, int wave, int n, int oy0, int ox0, int (&pixoff)[NT],
                                            int (&opix)[NT]) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int idx = (wave * NT + nt) * 16 + n;
    bool valid = idx < a.TH * a.TW;
    const int idc = valid ? idx : 0;
    const int ty = fastdiv(idc, a.TW, a.magic_tw);
    const int tx = idc - ty * a.TW;
    const int oy = oy0 + ty, ox = ox0 + tx;
    valid = valid && oy < a.Ho && ox < a.Wo;
    pixoff[nt] = ((ty * a.stride) * a.HW + tx * a.stride) * a.pp;
    opix[nt] = valid ? oy * a.Wo + ox : -1;
  }
}

// Halo stage, split in two so the global loads fly during the previous stage's MFMAs.
__device__ __forceinline__ void stage_load(const ConvArgs& a, const __bf16* src /* image + coff + chunk */, int iy0,
                                           int ix0, int sg, int sp0, int ppass, int HP, bf16x8 (&pre)[MAXPRE]) {
#pragma unroll
  for (int j = 0; j < MAXPRE; ++j) {
    const int pix = sp0 + j * ppass;
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (pix < HP && sg < a.cg) {
      const int hy = fastdiv(pix, a.HW, a.magic_hw);
      const int hx = pix - hy * a.HW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
        v = *reinterpret_cast<const bf16x8*>(src + (long long)(iy * a.W + ix) * a.in_pitch + sg * 8);
    }
    pre[j] = v;
  }
}

__device__ __forceinline__ void stage_store(const ConvArgs& a, unsigned char* dst, int sg, int sp0, int ppass, int HP,
                                            const bf16x8 (&pre)[MAXPRE]) {
  if (sg >= a.cg) return;
#pragma unroll
  for (int j = 0; j < MAXPRE; ++j) {
    const int pix = sp0 + j * ppass;
    if (pix < HP) *reinterpret_cast<bf16x8*>(dst + pix * a.pp + sg * 16) = pre[j];
  }
}

// K loop of one channel chunk: both operands from LDS.
template <int KS, int MT, int NT, bool UNIFORM>
__device__ __forceinline__ void kloop(const ConvArgs& a, const unsigned char* halo, const bf16x8* wl /* + lane */,
                                      int g, const int (&pixoff)[NT], f32x4 (&acc)[MT][NT]) {
  if (UNIFORM) {
    const int kkn = a.cg >> 2;
    int ks = 0;
#pragma unroll
    for (int tap = 0; tap < KS * KS; ++tap) {
      const int tapoff = ((tap / KS) * a.HW + (tap % KS)) * a.pp + g * 16;
      for (int kk = 0; kk < kkn; ++kk, ++ks) {
        bf16x8 wa[MT], xb[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) wa[mt] = wl[(ks * MT + mt) * 64];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          xb[nt] = *reinterpret_cast<const bf16x8*>(halo + pixoff[nt] + tapoff + kk * 64);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[mt], xb[nt], acc[mt][nt], 0, 0, 0);
      }
    }
  } else {
    for (int ks = 0; ks < a.ksteps; ++ks) {
      const int k8 = ks * 4 + g;
      int tap = fastdiv(k8, a.cg, a.magic_cg);
      int c8 = k8 - tap * a.cg;
      if (tap >= KS * KS) {  // zero-weight padding group: read any valid finite data
        tap = KS * KS - 1;
        c8 = 0;
      }
      const int kh = (tap * 11) >> 5, kw = tap - 3 * ((tap * 11) >> 5);
      const int off = (KS == 1 ? 0 : (kh * a.HW + kw) * a.pp) + c8 * 16;
      bf16x8 wa[MT], xb[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) wa[mt] = wl[(ks * MT + mt) * 64];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) xb[nt] = *reinterpret_cast<const bf16x8*>(halo + pixoff[nt] + off);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[mt], xb[nt], acc[mt][nt], 0, 0, 0);
    }
  }
}

// Output-channel order inside a block of MT*16 channels.  The MFMA leaves lane (pixel n, quad g)
// with rows g*4+r of every 16-row tile; the filter rows are permuted at pack time so that one lane's
// rows of two adjacent tiles are 8 CONSECUTIVE channels: 16-byte NHWC stores (64 contiguous bytes
// per pixel per store instruction) instead of 8-byte ones.
__host__ __device__ inline int chan_in_block(int MT, int mt, int g, int r) {
  const int even = MT & ~1;
  return mt < even ? ((((mt >> 1) * 4 + g) * 2 + (mt & 1)) * 4 + r) : (even * 16 + g * 4 + r);
}

template <int W>
__device__ __forceinline__ void store_group(const ConvArgs& a, int b, int pix, int co, float (&v)[W]) {
  // W = 8 or 4 consecutive channels starting at co, all < Cout when `vec` alignment holds
  const bool vec = ((a.Cout | a.out_pitch | a.out_coff | a.res_pitch | a.res_coff) & (W - 1)) == 0;
  if (vec) {
    if (a.res) {
      const __bf16* rp = a.res + (long long)b * a.res_img + (long long)pix * a.res_pitch + a.res_coff + co;
      if (W == 8) {
        const bf16x8 rv = *reinterpret_cast<const bf16x8*>(rp);
#pragma unroll
        for (int r = 0; r < W; ++r) v[r] += (float)rv[r];
      } else {
        const bf16x4 rv = *reinterpret_cast<const bf16x4*>(rp);
#pragma unroll
        for (int r = 0; r < W; ++r) v[r] += (float)rv[r];
      }
    }
    const long long o = (long long)b * a.out_img + (long long)pix * a.out_pitch + a.out_coff + co;
    if (a.out_f32) {
      float* op = reinterpret_cast<float*>(a.out) + o;
#pragma unroll
      for (int q = 0; q < W; q += 4) *reinterpret_cast<f32x4*>(op + q) = f32x4{v[q], v[q + 1], v[q + 2], v[q + 3]};
    } else {
      __bf16* op = reinterpret_cast<__bf16*>(a.out) + o;
      if (W == 8) {
        bf16x8 t;
#pragma unroll
        for (int r = 0; r < 8; ++r) t[r] = (__bf16)v[r];
        *reinterpret_cast<bf16x8*>(op) = t;
      } else {
        *reinterpret_cast<bf16x4*>(op) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < W; ++r) {
      if (co + r >= a.Cout) continue;
      float x = v[r];
      if (a.res) x += (float)a.res[(long long)b * a.res_img + (long long)pix * a.res_pitch + a.res_coff + co + r];
      const long long o = (long long)b * a.out_img + (long long)pix * a.out_pitch + a.out_coff + co + r;
      if (a.out_f32)
        reinterpret_cast<float*>(a.out)[o] = x;
      else
        reinterpret_cast<__bf16*>(a.out)[o] = (__bf16)x;
    }
  }
}

// + bias, SiLU, + residual, NHWC store.
template <int MT, int NT>
__device__ __forceinline__ void epilogue(const ConvArgs& a, int cb, int b, int g, const int (&opix)[NT],
                                         const f32x4 (&acc)[MT][NT]) {
  const int cbase = cb * MT * 16;
#pragma unroll
  for (int p = 0; p < MT / 2; ++p) {
    const int co = cbase + (p * 4 + g) * 8;
    if (co >= a.Cout) continue;
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + co);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(a.bias + co + 4);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (opix[nt] < 0) continue;
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float x0 = acc[2 * p][nt][r] + b0[r], x1 = acc[2 * p + 1][nt][r] + b1[r];
        v[r] = (a.act && !(a.dbg & 8)) ? silu_f(x0) : x0;
        v[4 + r] = (a.act && !(a.dbg & 8)) ? silu_f(x1) : x1;
      }
      store_group<8>(a, b, opix[nt], co, v);
    }
  }
  if (MT & 1) {
    constexpr int mt = MT - 1;
    const int co = cbase + (MT & ~1) * 16 + g * 4;
    if (co < a.Cout) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (opix[nt] < 0) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = acc[mt][nt][r] + bv[r];
          v[r] = a.act ? silu_f(x) : x;
        }
        store_group<4>(a, b, opix[nt], co, v);
      }
    }
  }
}

// XCD-aware, bijective block remap: blocks that share halo rows / weights land on one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

struct TilePos {
  int b, oy0, ox0;
};
__device__ __forceinline__ TilePos tile_pos(const ConvArgs& a, int t) {
  TilePos p;
  const int tx_i = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty_i = t % a.tiles_y;
  p.b = t / a.tiles_y;
  p.oy0 = ty_i * a.TH;
  p.ox0 = tx_i * a.TW;
  return p;
}

// ------------------------------------------------------------------ direct (LDS-free activations)
// Every wave owns a run of NT*16 CONSECUTIVE output pixels (row-major over the whole batch, so a
// patch is one contiguous run of NHWC memory: long DRAM bursts for the loads of a 1x1 conv and for
// every store) and MT*16 output channels.  Its B fragments (activations) come straight from global
// memory / L1 -- 16 contiguous bytes per lane, the tap shift is an address offset, the zero padding
// a per-lane predicate -- so there is no halo staging and no barrier in the pixel loop.  Only the
// block's filter slab lives in LDS (staged once, fragment order, conflict-free reads).
// K = (tap, cin) flattened in groups of 8 channels.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
// Buffer addressing (raw_buffer_load/store with a 32-bit byte offset): an offset beyond the
// descriptor's size makes a load return zeros and a store vanish, so zero padding, ragged patches
// and channel tails need no branches and no zero-fill moves -- just a select on the offset.
constexpr unsigned OOB = 0x80000000u;

template <int NT>
struct Patch {
  unsigned base[NT];  // byte offset of the (stride-scaled) centre pixel inside the input slice
  int gp[NT];         // global output pixel index, -1 if outside the tensor
  unsigned mask[NT];  // bit t: tap t of this pixel lies inside the image
};

template <int KS, int NT>
__device__ __forceinline__ void patch_setup(const ConvArgs& a, int patch, int n, Patch<NT>& P) {
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
    const int cy = oy * a.stride, cx = ox * a.stride;
    P.base[nt] = (unsigned)(((b * a.H + cy) * a.W + cx) * a.in_pitch) * 2u;
    unsigned m = 0;
    if (KS == 1) {
      m = valid ? 1u : 0u;
    } else {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int iy = cy + tp / 3 - 1, ix = cx + tp % 3 - 1;
        m |= ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) ? (1u << tp) : 0u;
      }
      m = valid ? m : 0u;
    }
    P.mask[nt] = m;
    P.gp[nt] = valid ? gp : -1;
  }
}

// B-fragment fetch of k-step `ks` (wave-uniform).  In the UNIFORM case (Cin % 32 == 0) a k-step lies
// inside one tap, so tap / channel block / tap byte offset are scalars and the per-lane work is a bit
// test, a select and an add per load; otherwise (tap, channel group) are derived per lane.  K-steps
// past the end (ring refills, padding) select the out-of-range offset and return zeros.
template <int KS>
__device__ __forceinline__ int tap_offset(const ConvArgs& a, int tap) {  // bytes
  if (KS == 1) return 0;
  const int kh = (tap * 11) >> 5;
  return ((kh - 1) * a.W + (tap - 3 * kh - 1)) * a.in_pitch * 2;
}

template <int KS, int NT, bool UNIFORM>
__device__ __forceinline__ void fetch_b(const ConvArgs& a, __amdgpu_buffer_rsrc_t rin, int ks, int g,
                                        const Patch<NT>& P, bf16x8 (&xb)[NT]) {
  if (UNIFORM) {
    const int tap = fastdiv(ks, a.kkn, a.magic_kkn);
    const int kk = ks - tap * a.kkn;
    const unsigned off = (unsigned)(tap_offset<KS>(a, tap) + kk * 64) + (unsigned)(g * 16);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const unsigned vo = ((P.mask[nt] >> tap) & 1u) ? P.base[nt] + off : OOB;
      xb[nt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rin, vo, 0, 0));
    }
  } else {
    const int k8 = ks * 4 + g;
    const int tap = KS == 1 ? 0 : fastdiv(k8, a.cg, a.magic_cg);
    const int c8 = k8 - tap * a.cg;
    const bool kvalid = KS == 1 ? (k8 < a.cg) : (tap < KS * KS);
    const unsigned off = (unsigned)(tap_offset<KS>(a, tap) + c8 * 16);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const unsigned vo = (kvalid && ((P.mask[nt] >> tap) & 1u)) ? P.base[nt] + off : OOB;
      xb[nt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rin, vo, 0, 0));
    }
  }
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

// Epilogue of the direct kernel (the accumulators already contain the bias): SiLU, residual, 16-byte
// buffer stores.  Requires every channel count / pitch / offset to be a multiple of 8 (host-checked,
// otherwise the generic epilogue is used).
template <int MT, int NT>
__device__ __forceinline__ void epilogue_buf(const ConvArgs& a, __amdgpu_buffer_rsrc_t rout,
                                             __amdgpu_buffer_rsrc_t rres, int cb, int g, const int (&gp)[NT],
                                             f32x4 (&acc)[MT][NT]) {
  const int cbase = cb * MT * 16;
  const unsigned esz = a.out_f32 ? 4u : 2u;
  if (a.act && !(a.dbg & 8)) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mt][nt][r] = silu_f(acc[mt][nt][r]);
  }
#pragma unroll
  for (int p = 0; p < MT / 2; ++p) {
    const int co = cbase + (p * 4 + g) * 8;
    const bool cok = co < a.Cout;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const bool ok = cok && gp[nt] >= 0;
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[2 * p][nt][r];
        v[4 + r] = acc[2 * p + 1][nt][r];
      }
      if (a.res) {
        const unsigned ro = ok ? (unsigned)gp[nt] * (unsigned)(a.res_pitch * 2) + (unsigned)((a.res_coff + co) * 2) : OOB;
        const bf16x8 rv = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rres, ro, 0, 0));
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += (float)rv[r];
      }
      const unsigned vo = ok ? (unsigned)gp[nt] * ((unsigned)a.out_pitch * esz) + (unsigned)(a.out_coff + co) * esz : OOB;
      if (a.out_f32) {
        __builtin_amdgcn_raw_buffer_store_b128(
            u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, rout,
            vo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(
            u32x4{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])}, rout,
            ok ? vo + 16u : OOB, 0, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]),
                                                     pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])},
                                               rout, vo, 0, 0);
      }
    }
  }
  if (MT & 1) {
    constexpr int mt = MT - 1;
    const int co = cbase + (MT & ~1) * 16 + g * 4;
    const bool cok = co < a.Cout;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const bool ok = cok && gp[nt] >= 0;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[mt][nt][r];
      if (a.res) {
        const unsigned ro = ok ? (unsigned)gp[nt] * (unsigned)(a.res_pitch * 2) + (unsigned)((a.res_coff + co) * 2) : OOB;
        const u32x2 rw = __builtin_amdgcn_raw_buffer_load_b64(rres, ro, 0, 0);
        const bf16x4 rv = __builtin_bit_cast(bf16x4, rw);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
      }
      const unsigned vo = ok ? (unsigned)gp[nt] * ((unsigned)a.out_pitch * esz) + (unsigned)(a.out_coff + co) * esz : OOB;
      if (a.out_f32)
        __builtin_amdgcn_raw_buffer_store_b128(
            u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, rout,
            vo, 0, 0);
      else
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])}, rout, vo, 0, 0);
    }
  }
}

template <int KS, int MT, int NT, bool UNIFORM>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  const int bid = xcd_remap(blockIdx.x, a.nblocks);
  const int cb = bid % a.ncb;
  const int pblk = bid / a.ncb;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.in + a.in_coff), (short)0, (a.dbg & 1) ? 0 : a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout =
      __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (a.dbg & 2) ? 0 : a.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.res, (short)0, a.res ? a.res_bytes : 0, 0x00020000);
  // Activation fragments ride a ring of D k-steps (D*NT = 8 sixteen-byte loads in flight per lane,
  // refilled right after the MFMAs that consumed a slot); filter fragments come from LDS one step
  // ahead.  The first ring fill of a patch is issued BEFORE the filter slab is staged (first patch)
  // or before the previous patch's epilogue stores, so its latency is always covered.
  constexpr int D = NT >= 4 ? 2 : (NT == 2 ? 4 : 8);
  bf16x8 xr[D][NT], wv[2][MT];
  Patch<NT> P;
  // wave-uniform values are forced into SGPRs so the k-loop bookkeeping stays on the scalar unit
  int patch = __builtin_amdgcn_readfirstlane(pblk * a.ppw * 4 + wave);
  bool live = patch < a.npatches;
  if (live) {
    patch_setup<KS, NT>(a, patch, n, P);
#pragma unroll
    for (int d = 0; d < D; ++d) fetch_b<KS, NT, UNIFORM>(a, rin, d, g, P, xr[d]);
  }
  // per-lane bias of its 4 rows of every channel tile (the accumulators start from it)
  f32x4 bias_r[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
    bias_r[mt] = *reinterpret_cast<const f32x4*>(a.bias + cb * MT * 16 + chan_in_block(MT, mt, g, 0));
  if (!(a.dbg & 4))
    copy_slab(reinterpret_cast<const bf16x8*>(a.wpk) + (size_t)cb * a.ksteps * MT * 64,
              reinterpret_cast<bf16x8*>(smem), a.ksteps * MT * 64, tid);
  __syncthreads();
  const bf16x8* wl = reinterpret_cast<const bf16x8*>(smem) + lane;

  for (int pp = 0; pp < a.ppw && live; ++pp) {
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = bias_r[mt];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) wv[0][mt] = wl[mt * 64];
    // Branch-free main loop over whole groups of D k-steps: ring refills past the end resolve to
    // out-of-range offsets (zeros, no memory access) and the `ks + 1` filter prefetch is clamped, so
    // the body is straight-line code and the compiler keeps counted vmcnt waits (the D-deep ring
    // really stays in flight).  The < D leftover steps run in a predicated tail.
    const int klast = a.ksteps - 1;
    int ks0 = 0;
    for (; ks0 + D <= a.ksteps; ks0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const int ks = ks0 + d;
        const int kn = min(ks + 1, klast);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) wv[(d + 1) & 1][mt] = wl[(kn * MT + mt) * 64];
        mma_step<MT, NT>(wv[d & 1], xr[d], acc);
        fetch_b<KS, NT, UNIFORM>(a, rin, ks + D, g, P, xr[d]);
      }
    }
#pragma unroll
    for (int d = 0; d < D - 1; ++d) {
      const int ks = ks0 + d;
      if (ks < a.ksteps) {
        const int kn = min(ks + 1, klast);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) wv[(d + 1) & 1][mt] = wl[(kn * MT + mt) * 64];
        mma_step<MT, NT>(wv[d & 1], xr[d], acc);
      }
    }
    // next patch: descriptors + first ring fill go out before this patch's stores
    int opix[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) opix[nt] = P.gp[nt];
    patch += 4;
    live = (pp + 1 < a.ppw) && patch < a.npatches;
    if (live) {
      patch_setup<KS, NT>(a, patch, n, P);
#pragma unroll
      for (int d = 0; d < D; ++d) fetch_b<KS, NT, UNIFORM>(a, rin, d, g, P, xr[d]);
    }
    epilogue_buf<MT, NT>(a, rout, rres, cb, g, opix, acc);
  }
}

// ------------------------------------------------------------------- one-tile-per-block fallback
template <int KS, int MT, int NT, bool UNIFORM>
__global__ __launch_bounds__(256) void conv_tile_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  const int bid = xcd_remap(blockIdx.x, a.nblocks);
  const int cb = bid % a.ncb;
  const TilePos tp = tile_pos(a, bid / a.ncb);
  const int iy0 = tp.oy0 * a.stride - a.pad, ix0 = tp.ox0 * a.stride - a.pad;

  int pixoff[NT], opix[NT];
  lane_pixels<NT>(a, wave, n, tp.oy0, tp.ox0, pixoff, opix);
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int HP = a.HH * a.HW;
  const int ppass = 256 >> a.lg2;
  const int sg = tid & ((1 << a.lg2) - 1), sp0 = tid >> a.lg2;
  const __bf16* inb = a.in + (long long)tp.b * a.in_img + a.in_coff;
  unsigned char* halo = smem + a.halo_lds_off;
  const int wstride = a.ksteps * MT * 64;
  bf16x8 pre[MAXPRE];
  for (int chunk = 0; chunk < a.nchunks; ++chunk) {
    if (chunk) __syncthreads();
    stage_load(a, inb + chunk * a.cch, iy0, ix0, sg, sp0, ppass, HP, pre);
    copy_slab(reinterpret_cast<const bf16x8*>(a.wpk) + ((size_t)cb * a.nchunks + chunk) * wstride,
              reinterpret_cast<bf16x8*>(smem), wstride, tid);
    stage_store(a, halo, sg, sp0, ppass, HP, pre);
    __syncthreads();
    kloop<KS, MT, NT, UNIFORM>(a, halo, reinterpret_cast<const bf16x8*>(smem) + lane, g, pixoff, acc);
  }
  epilogue<MT, NT>(a, cb, tp.b, g, opix, acc);
}

// ------------------------------------------------------------------------------------ host
static unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull / (unsigned)d) + 1ull); }

static const size_t LDS_MAX = 160 * 1024;     // per CU and per workgroup on gfx950
static const size_t W_STATIONARY_MAX = 80 * 1024;
static const int NUM_CU = 256;

int conv_choose(int batch, int Cin, int Cout, int k, int stride, int Ho, int Wo, int aligned8, ConvConfig* c) {
  if ((k != 1 && k != 3) || (stride != 1 && stride != 2) || Cin % 8 || Cin <= 0 || Cout <= 0)
    return fail(-3, "conv: unsupported geometry (k in {1,3}, stride in {1,2}, Cin % 8 == 0)");
  const int cout16 = (Cout + 15) / 16;
  const int ntaps = k * k;
  c->ntaps = ntaps;
  const long long pixels = (long long)batch * Ho * Wo;
  auto halo = [&](int th, int tw, int* hh, int* hw) {
    *hh = (th - 1) * stride + k;
    *hw = (tw - 1) * stride + k;
  };
  // channel-chunk candidates: divisors of Cin that are multiples of 8, at most 128, largest first;
  // multiples of 32 first (wave-uniform tap per k-step), then the rest
  std::vector<int> cands;
  for (int pass = 0; pass < 2; ++pass)
    for (int d = Cin < 128 ? Cin : 128; d >= 8; d -= 8)
      if (Cin % d == 0 && ((d % 32 == 0) == (pass == 0))) cands.push_back(d);
  if (Cin <= 128 && Cin % 32) cands.insert(cands.begin(), Cin);  // small odd Cin: a single chunk

  int mts[4], nmt = 0;
  if (cout16 % 4 == 0) mts[nmt++] = 4;
  if (cout16 % 5 == 0) mts[nmt++] = 5;
  if (cout16 % 2 == 0) mts[nmt++] = 2;
  mts[nmt++] = 1;

  // ---- preferred: direct kernel (activations straight from global/L1, filter slab in LDS)
  {
    static const char* env = getenv("YL_CONV_MODE");  // "tile" forces the LDS-halo kernel (A/B tests)
    const size_t slab_max = getenv("YL_CONV_SLAB_KB") ? (size_t)atoi(getenv("YL_CONV_SLAB_KB")) * 1024 : 72 * 1024;
    const int cg = Cin / 8;
    const int ksteps = (ntaps * cg + 3) / 4;
    // the direct kernel stores 16 bytes per lane: every channel count / pitch / offset must be a
    // multiple of 8 (true for all YOLO11 scales unless nc is odd-sized); otherwise the tile kernel runs
    for (int mi = 0; mi < nmt && aligned8 && !(env && !strcmp(env, "tile")); ++mi) {
      const int MT = mts[mi];
      if ((size_t)ksteps * MT * 1024 > slab_max) continue;
      const int ncb = cout16 / MT;
      const long long npix = (long long)batch * Ho * Wo;
      auto npatches = [&](int nt) { return (npix + 16 * nt - 1) / (16 * nt); };
      int NT = 1;
      static const int nt4_blocks = getenv("YL_CONV_NT4") ? atoi(getenv("YL_CONV_NT4")) : NUM_CU;
      static const int nt2_blocks = getenv("YL_CONV_NT2") ? atoi(getenv("YL_CONV_NT2")) : NUM_CU / 2;
      if (npatches(4) * ncb / 4 >= nt4_blocks) NT = 4;
      else if (npatches(2) * ncb / 4 >= nt2_blocks) NT = 2;
      if (MT == 5 && NT == 4) NT = 2;
      c->ws = 2;  // direct
      c->MT = MT;
      c->ncb = ncb;
      c->NT = NT;
      c->TH = 1;
      c->TW = 16 * NT;
      c->HH = c->HW = 0;
      c->tiles_x = c->tiles_y = 0;
      c->cch = Cin;
      c->cg = cg;
      c->lg2 = 0;
      c->nchunks = 1;
      c->uniform = (cg % 4 == 0);
      c->kloop = ksteps;
      c->ksteps = ksteps;
      c->pp = 0;
      c->halo_bytes = 0;
      c->halo_lds_off = 0;
      c->lds_bytes = (size_t)c->ksteps * MT * 1024;
      c->wpk_elems = (size_t)ncb * c->ksteps * MT * 64 * 8;
      c->ntiles = (int)npatches(NT);
      // patches per wave: amortise the slab load while keeping >= ~8 blocks per CU in the grid
      long long blocks1 = ((long long)c->ntiles + 3) / 4 * ncb;
      static const int ppw_div = getenv("YL_CONV_BPC") ? atoi(getenv("YL_CONV_BPC")) : 8;  // target blocks per CU
      int ppw = (int)(blocks1 / (ppw_div * NUM_CU));
      if (ppw < 1) ppw = 1;
      if (ppw > 32) ppw = 32;
      c->ppw = ppw;
      c->gws = (int)(((long long)c->ntiles + 4 * ppw - 1) / (4 * ppw));
      c->name = k == 3 ? (stride == 2 ? "conv3x3s2_mfma" : "conv3x3_mfma") : "conv1x1_mfma";
      return 0;
    }
  }

  // ---- fallback: LDS-halo tile kernel with per-chunk filter restaging
  for (int ws = 0; ws >= 0; --ws) {
    for (int mi = 0; mi < nmt; ++mi) {
      const int MT = mts[mi];
      const int ncb = cout16 / MT;
      int NT0 = 1;
      if ((pixels / 256) * ncb >= 2 * NUM_CU) NT0 = 4;
      else if ((pixels / 128) * ncb >= 2 * NUM_CU) NT0 = 2;
      if (MT == 5 && NT0 == 4) NT0 = 2;
      for (int NT = NT0; NT >= 1; NT >>= 1) {
        const int P = 4 * NT * 16;
        for (int cch : cands) {
          const int cg = cch / 8, nchunks = Cin / cch;
          int lg2 = 0;
          while ((1 << lg2) < cg) ++lg2;
          const int ksteps = (ntaps * cg + 3) / 4;
          const size_t wchunk = (size_t)ksteps * MT * 1024;
          const size_t wbytes = ws ? wchunk * nchunks : wchunk;
          if (ws && wbytes > W_STATIONARY_MAX) continue;
          if (!ws && wbytes > 64 * 1024) continue;
          const int pp = cch * 2 + 16;
          double best = 1e30;
          int bth = 0, btw = 0;
          for (int tw = 4; tw <= P && tw <= ((Wo + 3) / 4) * 4 + 12; ++tw) {
            int th = P / tw;
            if (th < 1) break;
            if (th > Ho) th = Ho;
            int hh, hw;
            halo(th, tw, &hh, &hw);
            const size_t hb = ((size_t)hh * hw * pp + 15) & ~(size_t)15;
            if (wbytes + (ws ? 2 : 1) * hb > LDS_MAX - 1024) continue;
            if ((hh * hw + (256 >> lg2) - 1) / (256 >> lg2) > MAXPRE) continue;
            const long long tiles = (long long)((Ho + th - 1) / th) * ((Wo + tw - 1) / tw);
            const double cost = (double)tiles * ((double)hh * hw + 0.75 * P);
            if (cost < best) {
              best = cost;
              bth = th;
              btw = tw;
            }
          }
          if (!bth) continue;
          c->ws = ws;
          c->MT = MT;
          c->ncb = ncb;
          c->NT = NT;
          c->TH = bth;
          c->TW = btw;
          halo(c->TH, c->TW, &c->HH, &c->HW);
          c->tiles_x = (Wo + c->TW - 1) / c->TW;
          c->tiles_y = (Ho + c->TH - 1) / c->TH;
          c->cch = cch;
          c->cg = cg;
          c->lg2 = lg2;
          c->nchunks = nchunks;
          c->uniform = (cch % 32 == 0);
          c->ksteps = ksteps;
          c->pp = pp;
          c->halo_bytes = (int)(((size_t)c->HH * c->HW * pp + 15) & ~(size_t)15);
          c->halo_lds_off = (int)wbytes;
          c->lds_bytes = wbytes + (size_t)(ws ? 2 : 1) * c->halo_bytes;
          c->wpk_elems = (size_t)ncb * nchunks * ksteps * MT * 64 * 8;
          c->ntiles = batch * c->tiles_x * c->tiles_y;
          if (ws) {
            const int bpc = (int)(LDS_MAX / c->lds_bytes) >= 2 ? 2 : 1;
            int cap = NUM_CU * bpc / ncb;
            if (cap < 1) cap = 1;
            const int rounds = (c->ntiles + cap - 1) / cap;
            c->gws = (c->ntiles + rounds - 1) / rounds;
          } else {
            c->gws = c->ntiles;
          }
          c->name = k == 3 ? (stride == 2 ? "conv3x3s2_mfma" : "conv3x3_mfma") : "conv1x1_mfma";
          return 0;
        }
      }
    }
  }
  return fail(-3, "conv: no tiling fits LDS");
}

static inline uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

void conv_pack_weights(const float* w, int Cout, int Cin_w, int Cin_pad, int k, const ConvConfig& c,
                       std::vector<uint16_t>* out) {
  (void)Cin_pad;
  out->assign(c.wpk_elems, 0);
  size_t o = 0;
  for (int cb = 0; cb < c.ncb; ++cb)
    for (int chunk = 0; chunk < c.nchunks; ++chunk)
      for (int ks = 0; ks < c.ksteps; ++ks)
        for (int mt = 0; mt < c.MT; ++mt)
          for (int lane = 0; lane < 64; ++lane) {
            const int co = cb * c.MT * 16 + chan_in_block(c.MT, mt, (lane & 15) >> 2, lane & 3);
            const int k8 = ks * 4 + (lane >> 4);
            const int tap = k8 / c.cg, c8 = k8 % c.cg;
            for (int j = 0; j < 8; ++j, ++o) {
              const int ci = chunk * c.cch + c8 * 8 + j;
              if (co < Cout && tap < c.ntaps && ci < Cin_w) {
                const int kh = tap / k, kw = tap % k;
                (*out)[o] = f2bf(w[(((size_t)co * Cin_w + ci) * k + kh) * k + kw]);
              }
            }
          }
}

template <int KS, int MT, int NT, bool U>
static int launch_t(const ConvConfig& cfg, const ConvArgs& a, hipStream_t s) {
  if (cfg.ws == 2) {
    auto kern = conv_direct_kernel<KS, MT, NT, U>;
    static size_t attr = 0;  // per instantiation: largest dynamic LDS size enabled so far
    if (cfg.lds_bytes > attr) {
      YL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)LDS_MAX));
      attr = LDS_MAX;
    }
    hipLaunchKernelGGL(kern, dim3(a.nblocks), dim3(256), cfg.lds_bytes, s, a);
  } else {
    auto kern = conv_tile_kernel<KS, MT, NT, U>;
    static size_t attr = 0;
    if (cfg.lds_bytes > attr) {
      YL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)LDS_MAX));
      attr = LDS_MAX;
    }
    hipLaunchKernelGGL(kern, dim3(a.nblocks), dim3(256), cfg.lds_bytes, s, a);
  }
  YL_HIP(hipGetLastError());
  return 0;
}

template <int KS, int MT, int NT>
static int launch_u(const ConvConfig& cfg, const ConvArgs& a, hipStream_t s) {
  return cfg.uniform ? launch_t<KS, MT, NT, true>(cfg, a, s) : launch_t<KS, MT, NT, false>(cfg, a, s);
}

template <int KS, int MT>
static int launch_n(const ConvConfig& cfg, const ConvArgs& a, hipStream_t s) {
  return cfg.NT == 4 ? launch_u<KS, MT, 4>(cfg, a, s)
                     : cfg.NT == 2 ? launch_u<KS, MT, 2>(cfg, a, s) : launch_u<KS, MT, 1>(cfg, a, s);
}

template <int KS>
static int launch_m(const ConvConfig& cfg, const ConvArgs& a, hipStream_t s) {
  switch (cfg.MT) {
    case 1: return launch_n<KS, 1>(cfg, a, s);
    case 2: return launch_n<KS, 2>(cfg, a, s);
    case 4: return launch_n<KS, 4>(cfg, a, s);
    case 5: return cfg.NT == 1 ? launch_u<KS, 5, 1>(cfg, a, s) : launch_u<KS, 5, 2>(cfg, a, s);
  }
  return fail(-3, "conv: bad MT");
}

int conv_launch(const ConvConfig& cfg, ConvArgs a, int batch, hipStream_t s) {
  a.TH = cfg.TH;
  a.TW = cfg.TW;
  a.tiles_x = cfg.tiles_x;
  a.tiles_y = cfg.tiles_y;
  a.ncb = cfg.ncb;
  a.HH = cfg.HH;
  a.HW = cfg.HW;
  a.cch = cfg.cch;
  a.cg = cfg.cg;
  a.lg2 = cfg.lg2;
  a.nchunks = cfg.nchunks;
  a.ksteps = cfg.ksteps;
  a.kloop = cfg.kloop;
  a.kkn = cfg.cg >> 2;
  a.magic_kkn = magic_of(cfg.cg >> 2);
  a.pp = cfg.pp;
  a.ntaps = cfg.ntaps;
  a.halo_lds_off = cfg.halo_lds_off;
  a.halo_bytes = cfg.halo_bytes;
  a.ntiles = cfg.ntiles;
  a.npatches = cfg.ntiles;
  a.ppw = cfg.ppw;
  a.gws = cfg.gws;
  a.magic_hw = magic_of(cfg.HW);
  a.magic_tw = magic_of(cfg.TW);
  a.magic_cg = magic_of(cfg.cg);
  a.magic_wo = magic_of(a.Wo);
  a.magic_ho = magic_of(a.Ho);
  a.npix = batch * a.Ho * a.Wo;
  {
    const long long inb = ((long long)batch * a.in_img - a.in_coff) * 2;
    const long long outb = (long long)batch * a.out_img * (a.out_f32 ? 4 : 2);
    const long long resb = (long long)batch * a.res_img * 2;
    if (cfg.ws == 2 && (inb >= (1LL << 31) || outb >= (1LL << 31) || resb >= (1LL << 31)))
      return fail(-3, "conv: tensor too large for 32-bit buffer offsets");
    a.in_bytes = (unsigned)inb;
    a.out_bytes = (unsigned)outb;
    a.res_bytes = (unsigned)resb;
  }
  a.nblocks = cfg.gws * cfg.ncb;
  {
    static const int dbg = getenv("YL_CONV_DBG") ? atoi(getenv("YL_CONV_DBG")) : 0;  // ablation, dev only
    a.dbg = dbg;
  }
  return cfg.ntaps == 9 ? launch_m<3>(cfg, a, s) : launch_m<1>(cfg, a, s);
}

}  // namespace yl
