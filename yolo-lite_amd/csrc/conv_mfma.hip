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
// One workgroup = 4 waves = TH x TW output pixels (4*NT*16) x MT*16 output channels.
#include "yl_internal.h"

namespace yl {

__device__ __forceinline__ int fastdiv(int x, int d, unsigned magic) {
  return d == 1 ? x : (int)__umulhi((unsigned)x, magic);
}

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }

template <int KS, int MT, int NT, bool UNIFORM>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;

  // XCD-aware, bijective block remap: blocks that share halo rows / weights land on one L2.
  int bid = blockIdx.x;
  {
    const int q = a.nblocks >> 3, r = a.nblocks & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int cb = bid % a.ncb;
  int t = bid / a.ncb;
  const int tx_i = t % a.tiles_x;
  t /= a.tiles_x;
  const int ty_i = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int oy0 = ty_i * a.TH, ox0 = tx_i * a.TW;
  const int iy0 = oy0 * a.stride - a.pad, ix0 = ox0 * a.stride - a.pad;

  // Per-lane output pixels (one per pixel tile) and their halo offsets in LDS.
  int pixoff[NT];
  int opix[NT];  // oy*Wo+ox or -1
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

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int HP = a.HH * a.HW;
  const int ppass = 256 >> a.lg2;
  const int sg = tid & ((1 << a.lg2) - 1), sp0 = tid >> a.lg2;
  const __bf16* inb = a.in + (long long)b * a.in_img + a.in_coff;

  for (int chunk = 0; chunk < a.nchunks; ++chunk) {
    if (chunk) __syncthreads();
    // ---- stage the halo tile of this channel chunk: global NHWC -> LDS [pixel][cch] (+16 B pad)
    if (sg < a.cg) {
      for (int pix = sp0; pix < HP; pix += ppass) {
        const int hy = fastdiv(pix, a.HW, a.magic_hw);
        const int hx = pix - hy * a.HW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
          v = *reinterpret_cast<const bf16x8*>(inb + (long long)(iy * a.W + ix) * a.in_pitch +
                                               chunk * a.cch + sg * 8);
        *reinterpret_cast<bf16x8*>(smem + pix * a.pp + sg * 16) = v;
      }
    }
    __syncthreads();

    const bf16x8* wp = reinterpret_cast<const bf16x8*>(a.wpk) +
                       ((size_t)(cb * a.nchunks + chunk) * a.ksteps) * MT * 64 + lane;
    if (UNIFORM) {
      const int kkn = a.cg >> 2;
      int ks = 0;
#pragma unroll
      for (int tap = 0; tap < KS * KS; ++tap) {
        const int tapoff = ((tap / KS) * a.HW + (tap % KS)) * a.pp + g * 16;
        for (int kk = 0; kk < kkn; ++kk, ++ks) {
          bf16x8 wa[MT], xb[NT];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) wa[mt] = wp[(ks * MT + mt) * 64];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            xb[nt] = *reinterpret_cast<const bf16x8*>(smem + pixoff[nt] + tapoff + kk * 64);
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
        for (int mt = 0; mt < MT; ++mt) wa[mt] = wp[(ks * MT + mt) * 64];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) xb[nt] = *reinterpret_cast<const bf16x8*>(smem + pixoff[nt] + off);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[mt], xb[nt], acc[mt][nt], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: + bias, SiLU, + residual, store 4 consecutive channels of one pixel
  const bool vec = ((a.Cout | a.out_pitch | a.out_coff | a.res_pitch | a.res_coff) & 3) == 0;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int co = (cb * MT + mt) * 16 + g * 4;
    if (co >= a.Cout) continue;
    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (opix[nt] < 0) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = acc[mt][nt][r] + bv[r];
        v[r] = a.act ? silu_f(x) : x;
      }
      if (vec) {
        if (a.res) {
          const bf16x4 rv = *reinterpret_cast<const bf16x4*>(
              a.res + (long long)b * a.res_img + (long long)opix[nt] * a.res_pitch + a.res_coff + co);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
        }
        if (a.out_f32) {
          float* op = reinterpret_cast<float*>(a.out) + (long long)b * a.out_img +
                      (long long)opix[nt] * a.out_pitch + a.out_coff + co;
          *reinterpret_cast<f32x4*>(op) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
          __bf16* op = reinterpret_cast<__bf16*>(a.out) + (long long)b * a.out_img +
                       (long long)opix[nt] * a.out_pitch + a.out_coff + co;
          *reinterpret_cast<bf16x4*>(op) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (co + r >= a.Cout) continue;
          float x = v[r];
          if (a.res)
            x += (float)a.res[(long long)b * a.res_img + (long long)opix[nt] * a.res_pitch + a.res_coff + co + r];
          const long long o = (long long)b * a.out_img + (long long)opix[nt] * a.out_pitch + a.out_coff + co + r;
          if (a.out_f32)
            reinterpret_cast<float*>(a.out)[o] = x;
          else
            reinterpret_cast<__bf16*>(a.out)[o] = (__bf16)x;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------ host
static unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull / (unsigned)d) + 1ull); }

int conv_choose(int batch, int Cin, int Cout, int k, int stride, int Ho, int Wo, ConvConfig* c) {
  if ((k != 1 && k != 3) || (stride != 1 && stride != 2) || Cin % 8 || Cin <= 0 || Cout <= 0)
    return fail(-3, "conv: unsupported geometry (k in {1,3}, stride in {1,2}, Cin % 8 == 0)");
  const int cout16 = (Cout + 15) / 16;
  int MT = 1;
  if (cout16 % 4 == 0) MT = 4;
  else if (cout16 % 5 == 0) MT = 5;
  else if (cout16 % 2 == 0) MT = 2;
  c->MT = MT;
  c->ncb = cout16 / MT;
  const long long pixels = (long long)batch * Ho * Wo;
  // 256-pixel tiles when that still gives >= ~4 blocks per CU, else 128.
  int NT = (pixels / 256) * c->ncb >= 1024 ? 4 : 2;
  if (MT == 5 && NT == 4) NT = 2;  // keep accumulators <= 64 VGPRs
  c->NT = NT;
  const int P = 4 * NT * 16;
  c->ntaps = k * k;
  const int pad = k / 2;
  (void)pad;
  // channel chunk
  int cch;
  if (Cin <= 128) cch = Cin;
  else if (Cin % 64 == 0) cch = 64;
  else if (Cin % 32 == 0) cch = 32;
  else if (Cin % 16 == 0) cch = 16;
  else cch = 8;
  // tile search: minimise staged halo pixels per output pixel, then the block count
  auto halo = [&](int th, int tw, int* hh, int* hw) {
    *hh = (th - 1) * stride + k;
    *hw = (tw - 1) * stride + k;
  };
  for (;;) {
    double best = 1e30;
    int bth = 0, btw = 0;
    for (int tw = 4; tw <= P && tw <= ((Wo + 3) / 4) * 4 + 12; ++tw) {
      int th = P / tw;
      if (th < 1) break;
      if (th > Ho) th = Ho;
      int hh, hw;
      halo(th, tw, &hh, &hw);
      const size_t lds = (size_t)hh * hw * (cch * 2 + 16);
      if (lds > 64 * 1024) continue;
      const long long tiles = (long long)((Ho + th - 1) / th) * ((Wo + tw - 1) / tw);
      // cost: staged pixels + a penalty for idle MFMA slots (P per tile regardless of use)
      const double cost = (double)tiles * ((double)hh * hw * (k == 1 ? 1.0 : 1.0) + 0.75 * P);
      if (cost < best) {
        best = cost;
        bth = th;
        btw = tw;
      }
    }
    if (bth) {
      c->TH = bth;
      c->TW = btw;
      break;
    }
    // does not fit: halve the chunk (must still divide Cin and be a multiple of 8)
    int nc = cch / 2;
    while (nc >= 8 && (Cin % nc || nc % 8)) nc -= 8;
    if (nc < 8) return fail(-3, "conv: no tile fits LDS");
    cch = nc;
  }
  halo(c->TH, c->TW, &c->HH, &c->HW);
  c->tiles_x = (Wo + c->TW - 1) / c->TW;
  c->tiles_y = (Ho + c->TH - 1) / c->TH;
  c->cch = cch;
  c->cg = cch / 8;
  c->lg2 = 0;
  while ((1 << c->lg2) < c->cg) ++c->lg2;
  c->nchunks = Cin / cch;
  c->uniform = (cch % 32 == 0);
  c->ksteps = (c->ntaps * c->cg + 3) / 4;
  c->pp = cch * 2 + 16;
  c->lds_bytes = (size_t)c->HH * c->HW * c->pp;
  c->wpk_elems = (size_t)c->ncb * c->nchunks * c->ksteps * c->MT * 64 * 8;
  c->name = k == 3 ? (stride == 2 ? "conv3x3s2_mfma" : "conv3x3_mfma") : "conv1x1_mfma";
  return 0;
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
            const int co = (cb * c.MT + mt) * 16 + (lane & 15);
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
  auto kern = conv_mfma_kernel<KS, MT, NT, U>;
  if (cfg.lds_bytes > 48 * 1024) {
    static bool done = false;  // per instantiation
    if (!done) {
      YL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 64 * 1024));
      done = true;
    }
  }
  hipLaunchKernelGGL(kern, dim3(a.nblocks), dim3(256), cfg.lds_bytes, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

template <int KS, int MT, int NT>
static int launch_u(const ConvConfig& cfg, const ConvArgs& a, hipStream_t s) {
  return cfg.uniform ? launch_t<KS, MT, NT, true>(cfg, a, s) : launch_t<KS, MT, NT, false>(cfg, a, s);
}

template <int KS, int MT>
static int launch_n(const ConvConfig& cfg, const ConvArgs& a, hipStream_t s) {
  return cfg.NT == 4 ? launch_u<KS, MT, 4>(cfg, a, s) : launch_u<KS, MT, 2>(cfg, a, s);
}

template <int KS>
static int launch_m(const ConvConfig& cfg, const ConvArgs& a, hipStream_t s) {
  switch (cfg.MT) {
    case 1: return launch_n<KS, 1>(cfg, a, s);
    case 2: return launch_n<KS, 2>(cfg, a, s);
    case 4: return launch_n<KS, 4>(cfg, a, s);
    case 5: return launch_u<KS, 5, 2>(cfg, a, s);
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
  a.pp = cfg.pp;
  a.ntaps = cfg.ntaps;
  a.magic_hw = magic_of(cfg.HW);
  a.magic_tw = magic_of(cfg.TW);
  a.magic_cg = magic_of(cfg.cg);
  a.nblocks = batch * cfg.tiles_x * cfg.tiles_y * cfg.ncb;
  return cfg.ntaps == 9 ? launch_m<3>(cfg, a, s) : launch_m<1>(cfg, a, s);
}

}  // namespace yl
