// fp8 (OCP e4m3) Conv2d + bias + SiLU [+ residual] as an MFMA implicit GEMM on the block-scaled matrix pipe of gfx950:
// v_mfma_f32_16x16x128_f8f6f4 (K = 128 per instruction, twice the bf16 rate per clock; scales unused = 1.0).
//
// Same reference semantics as conv_mfma.hip (nn/modules/conv.py:35-53 with BN folded per utils/torch_utils.py:182-209,
// Bottleneck add block.py:341-343), computed on quantised operands (BASELINE.json config 5: "e4m3 weights +
// activations, fp32 accumulate"):
//     x_q = e4m3(x / s_in)          per-tensor activation scale, calibrated by the host (nn/lower.py::apply_fp8)
//     w_q = e4m3(w[c] / s_w[c])     per-output-channel filter scale, chosen at pack time (absmax / 448)
//     y   = act(alpha[c] * sum(w_q * x_q) + bias[c]) [+ res],   alpha[c] = s_in * s_w[c],  fp32 accumulation
// and y leaves either as bf16 planes (the layout of every other kernel) or as e4m3(y / s_out) fp8 planes.
//
// fp8 activation layout ("C16-planar, guarded"): buffer[C/16][plane_px][16] -- the bf16 layout's geometry (yl_internal.h:
// one zero guard pixel after every row, one guard row after every image, 16 bytes per pixel) with 16 channels per
// plane instead of 8, so taps are the same constant byte offsets and a wave's fragment is again runs of 256 contiguous
// bytes.
//
// GEMM mapping: A (16 x 128) = packed filters from an LDS slab, row = output channel (natural order); B (128 x 16) =
// activations straight from the planes, column = pixel.  K = (tap, cin) in groups of 16 channels; one k-step is 8 groups;
// lane group g = lane >> 4 supplies groups 2g and 2g + 1 of a step for A and B alike (32 bytes each: two 16-byte loads),
// so whatever order the instruction gives the 32 bytes of a lane inside K, filters and activations agree on it.
// D: lane (pixel n, group g) holds channels 16 mt + 4 g + r of tile mt: 4-byte (fp8) or 8-byte (bf16) stores, which a
// wave still issues as 256 contiguous bytes per tile.
#include "conv_common.h"
#include "f8_epilogue.h"

#include <cmath>

#ifndef YL_F8_D
#define YL_F8_D 2  // build switch (development): prefetch depth of conv_f8_kernel for MT <= 3
#endif
namespace yl {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr float F8_MAX = 448.0f;  // largest finite e4m3

// ---------------------------------------------------------------------------------------- bf16 planes -> fp8 planes
// One thread = one pixel of one 16-channel plane (two bf16 planes in, 16 bytes out).  Runs over whole planes, guards
// included: a guard is zero on both sides.
__global__ __launch_bounds__(256) void quant_f8_kernel(const bf16x8* __restrict__ in, v4i* __restrict__ out, unsigned plane_px,
                                                      int planes16, float inv_scale) {
  const unsigned px = blockIdx.x * 256u + threadIdx.x;
  const int p = blockIdx.y;
  if (px >= plane_px || p >= planes16) return;
  const bf16x8 lo = in[(size_t)(2 * p) * plane_px + px];
  const bf16x8 hi = in[(size_t)(2 * p + 1) * plane_px + px];
  float v[16];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    v[j] = (float)lo[j] * inv_scale;
    v[8 + j] = (float)hi[j] * inv_scale;
  }
  v4i o;
#pragma unroll
  for (int q = 0; q < 4; ++q) o[q] = (int)pack_f8x4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
  out[(size_t)p * plane_px + px] = o;
}

int quant_f8_launch(const PView& in, void* out, unsigned plane_px, int C, float scale, hipStream_t s) {
  if (C % 16 || !(scale > 0.f)) return fail(-3, "quantize: channel count must be a multiple of 16 and the scale positive");
  hipLaunchKernelGGL(quant_f8_kernel, dim3((plane_px + 255) / 256, C / 16), dim3(256), 0, s,
                     reinterpret_cast<const bf16x8*>(in.base), reinterpret_cast<v4i*>(out), plane_px, C / 16, 1.0f / scale);
  YL_HIP(hipGetLastError());
  return 0;
}

// --------------------------------------------------------------------------------------------------- conv kernel
// B fragments of k-step ks: two 16-byte loads per pixel tile (k-groups 2g and 2g + 1).  UNIFORM (Cin % 128 == 0): a
// step lies inside one tap, offsets are wave-uniform and Patch::base already contains the lane group's plane.
template <int KS, int NT, bool UNIFORM>
__device__ __forceinline__ void fetch_b8(const ConvArgs& a, __amdgpu_buffer_rsrc_t rin, int ks, int g, const Patch<NT>& P,
                                         v8i (&xb)[NT]) {
  unsigned off0, off1;
  if (UNIFORM) {
    const int tap = fastdiv(ks, a.kkn, a.magic_kkn);
    const int kk = ks - tap * a.kkn;
    off0 = ks < a.ksteps ? (unsigned)tap_offset<KS>(a, tap) + (unsigned)(kk * 8) * a.in_plane : OOB_K;
    off1 = off0 + a.in_plane;
  } else {
    const int q0 = ks * 8 + 2 * g, q1 = q0 + 1;
    const int nq = a.ntaps * a.cg;
    int t0, t1, c0, c1;
    if (KS == 3 && a.kord) {  // plane-major (see conv_f8_choose): k-group = plane * 9 + tap
      c0 = (int)__umulhi((unsigned)q0, 477218589u);  // q / 9
      c1 = (int)__umulhi((unsigned)q1, 477218589u);
      t0 = q0 - 9 * c0;
      t1 = q1 - 9 * c1;
    } else {
      t0 = KS == 1 ? 0 : fastdiv(q0, a.cg, a.magic_cg);
      t1 = KS == 1 ? 0 : fastdiv(q1, a.cg, a.magic_cg);
      c0 = q0 - t0 * a.cg;
      c1 = q1 - t1 * a.cg;
    }
    off0 = q0 < nq ? (unsigned)tap_offset<KS>(a, t0) + (unsigned)c0 * a.in_plane : OOB_K;
    off1 = q1 < nq ? (unsigned)tap_offset<KS>(a, t1) + (unsigned)c1 * a.in_plane : OOB_K;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(rin, P.base[nt] + off0, 0, 0);
    const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(rin, P.base[nt] + off1, 0, 0);
    xb[nt] = v8i{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  }
}

// slab: [k-step][MT][half][64 lanes] x 16 bytes (a lane's two halves 1 KiB apart: conflict-free ds_read_b128)
template <int MT, int NT>
__device__ __forceinline__ void mma_step8(const v4i* wl, int row, const v8i (&xb)[NT], f32x4 (&acc)[MT][NT]) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const v4i lo = wl[((row * MT + mt) * 2) * 64], hi = wl[((row * MT + mt) * 2 + 1) * 64];
    const v8i wa = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wa, xb[nt], acc[mt][nt], 0, 0, 0, 0, 0, 0);
  }
}

template <int KS, int MT, int NT, bool UNIFORM, int D>
__device__ __forceinline__ void run_ksteps8(const ConvArgs& a, __amdgpu_buffer_rsrc_t rin, int g, const Patch<NT>& P,
                                            const v4i* wl, int kbeg, int nk, v8i (&xr)[D][NT], f32x4 (&acc)[MT][NT]) {
  int i0 = 0;
  for (; i0 + D <= nk; i0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      mma_step8<MT, NT>(wl, i0 + d, xr[d], acc);
      fetch_b8<KS, NT, UNIFORM>(a, rin, kbeg + i0 + d + D, g, P, xr[d]);
    }
  }
#pragma unroll
  for (int d = 0; d < D - 1; ++d)
    if (i0 + d < nk) mma_step8<MT, NT>(wl, i0 + d, xr[d], acc);
}

template <int KS, int MT, int NT, bool UNIFORM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv_f8_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  const int bid = xcd_remap(blockIdx.x, a.nblocks);
  const int cb = bid % a.ncb;
  const int pblk = bid / a.ncb;
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, (short)0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, a.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.res, (short)0, a.res ? a.res_bytes : 0, 0x00020000);
  constexpr int D = MT <= 3 ? YL_F8_D : 2;  // k-steps of activation fragments in flight (narrow tiles can afford more)
  v8i xr[D][NT];
  Patch<NT> P;
  const unsigned lane_off = UNIFORM ? (unsigned)(2 * g) * a.in_plane : 0u;
  int patch = __builtin_amdgcn_readfirstlane(pblk * a.ppw * 4 + wave);
  bool live = patch < a.npatches;
  if (live) {
    patch_setup<NT>(a, patch, n, lane_off, P);
#pragma unroll
    for (int d = 0; d < D; ++d) fetch_b8<KS, NT, UNIFORM>(a, rin, d, g, P, xr[d]);
  }
  const bf16x8* wsrc = reinterpret_cast<const bf16x8*>(a.wpk) + (size_t)cb * a.ksteps * MT * 128;  // 16-byte units
  const v4i* wl = reinterpret_cast<const v4i*>(smem8) + lane;
  const bool multi = a.kchunk < a.ksteps;
  if (tid < 2 * MT * 16) {  // the epilogue's alpha / bias of this channel block -> LDS (visible behind the first slab barrier)
    float* lab = reinterpret_cast<float*>(smem8 + a.ab_off);
    const int i = tid < MT * 16 ? tid : tid - MT * 16;
    lab[tid] = (tid < MT * 16 ? a.alpha : a.bias)[cb * MT * 16 + i];
  }
  bool first = true;
  for (int pp = 0; pp < a.ppw; ++pp) {
    if (!first && !live) break;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (a.slab2, round 5: a chunked filter alternates between two chunk buffers filled by LDS-DMA -- conv_mfma.hip conv_body: one raw
    // barrier per chunk behind a counted wait that leaves the activation ring, D k-steps x 2 NT loads, in flight; one k-loop call site)
    typedef __attribute__((address_space(3))) void lds_void_8;
    const bool async = multi && a.slab2;
    const int chunk_bytes = a.kchunk * MT * 2048;
    auto dma_chunk = [&](int kc, int buf) __attribute__((always_inline)) {
      const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, (short)0, (unsigned)a.ksteps * MT * 2048u, 0x00020000);
      const int np = min(a.kchunk, a.ksteps - kc) * MT * 2;  // 1 KiB pieces
      for (int pc = wave; pc < np; pc += 4)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_8*)(smem8 + buf * chunk_bytes + pc * 1024), 16,
                                                 (kc * MT * 2 + pc) * 1024 + lane * 16, 0, 0, 0);
    };
    int c = 0;
    for (int kc = 0; kc < a.ksteps; kc += a.kchunk, ++c) {
      const int nk = min(a.kchunk, a.ksteps - kc);
      const bool more = kc + a.kchunk < a.ksteps;
      if (async) {
        if (first) {
          dma_chunk(0, 0);
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          first = false;
        }
        if (more) dma_chunk(kc + a.kchunk, (c + 1) & 1);
      } else if (multi || first) {
        if (!first) __syncthreads();
        copy_slab(wsrc + (size_t)kc * MT * 128, reinterpret_cast<bf16x8*>(smem8), nk * MT * 128, tid);
        __syncthreads();
        first = false;
      }
      if (live) run_ksteps8<KS, MT, NT, UNIFORM, D>(a, rin, g, P, wl + ((async && (c & 1)) ? chunk_bytes / 16 : 0), kc, nk, xr, acc);
      if (async && more) {
        if (live) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(D * 2 * NT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
    if (!live) break;
    int opix[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) opix[nt] = P.opix[nt];
    patch += 4;
    live = (pp + 1 < a.ppw) && patch < a.npatches;
    // the next patch's first fragments are requested before the epilogue; with six channel tiles only its first k-step (the
    // accumulators, two steps of fragments and the epilogue's own registers do not fit 256: the spills' reloads each cost a
    // vmcnt(0) in the middle of the stores), the second step behind the epilogue
    constexpr int DPRE = MT == 6 ? 1 : D;
    if (live) {
      patch_setup<NT>(a, patch, n, lane_off, P);
#pragma unroll
      for (int d = 0; d < DPRE; ++d) fetch_b8<KS, NT, UNIFORM>(a, rin, d, g, P, xr[d]);
    }
    // ---- epilogue (f8_epilogue.h): the hot combinations of residual / output kind are compiled as straight-line code
    f8_epilogue_any<MT, NT>(a, acc, opix, cb, g, reinterpret_cast<const float*>(smem8 + a.ab_off), rres, rout);
    if (DPRE < D && live) {
#pragma unroll
      for (int d = DPRE; d < D; ++d) fetch_b8<KS, NT, UNIFORM>(a, rin, d, g, P, xr[d]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------ host
static unsigned magic8(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull / (unsigned)d) + 1ull); }

// float -> OCP e4m3fn, round to nearest even, saturating (what v_cvt_pk_fp8_f32 does for in-range values)
uint8_t f32_to_e4m3(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint8_t sign = (uint8_t)((u >> 24) & 0x80);
  if (std::isnan(f)) return (uint8_t)(sign | 0x7f);
  float a = std::fabs(f);
  if (a >= F8_MAX) return (uint8_t)(sign | 0x7e);
  int ex = a > 0.f ? std::ilogb(a) : -6;
  if (ex < -6) ex = -6;                           // subnormals share the exponent of the smallest normal
  const float step = std::ldexp(1.0f, ex - 3);    // spacing of representable values in this binade
  const float q = std::nearbyint(a / step);       // RNE in the default rounding mode
  const float v = q * step;
  if (v == 0.f) return sign;
  if (v >= F8_MAX) return (uint8_t)(sign | 0x7e);
  if (v < std::ldexp(1.0f, -6)) return (uint8_t)(sign | (uint8_t)std::lrint(v / std::ldexp(1.0f, -9)));
  const int e2 = std::ilogb(v);
  const int mant = (int)std::lrint(v / std::ldexp(1.0f, e2 - 3)) - 8;
  return (uint8_t)(sign | ((e2 + 7) << 3) | mant);
}

int conv_f8_choose(int batch, int Cin, int Cout, int k, int stride, int Ho, int Wo, ConvConfig* c) {
  if ((k != 1 && k != 3) || (stride != 1 && stride != 2) || Cin % 16 || Cout % 16 || Cin <= 0 || Cout <= 0)
    return fail(-3, "fp8 conv: k in {1,3}, stride in {1,2}, Cin % 16 == 0, Cout % 16 == 0");
  const int cout16 = Cout / 16;
  int MT = cout16 % 4 == 0 ? 4 : (cout16 % 3 == 0 ? 3 : (cout16 % 2 == 0 ? 2 : 0));
  if (!MT) return fail(-3, "fp8 conv: Cout must be a multiple of 32 or 48");
  // six tiles per wave (96 output channels) instead of four on wide layers: a third fewer re-reads of the activations
  // (YOLO11x 1280x1280: the stride-2 3x3 convs 1.56 -> 1.43 ms).  Not instead of three tiles (the 96-channel Bottleneck convs
  // get slower: 250+ VGPRs with a few spills do not pay on short k-loops).  YL_F8_MT6=0 turns it off.
  const int mt6 = tune_int("YL_F8_MT6", 1);
  const long long npix = (long long)batch * Ho * Wo;
  // ... and not on maps so small that six tiles x 64 pixels per wave leave CUs idle (YOLO11x 1280x1280 batch 8: the 192 -> 192
  // convs at 40x40 are 400 waves = 100 workgroups for 256 CUs): YL_F8_MT6_MINWG = fewest workgroups the six-tile form must fill
  const long long wg6 = (npix + 63) / 64 * (cout16 / 6) / 4;
  if (mt6 && MT == 4 && cout16 % 6 == 0 && Cin >= 128 && wg6 >= tune_int("YL_F8_MT6_MINWG", 0)) MT = 6;
  const int ntaps = k * k, cg = Cin / 16;
  const int ksteps = (ntaps * cg + 7) / 8;
  const size_t slab_max = 72 * 1024, step_bytes = (size_t)MT * 2048;
  int kchunk = ksteps;
  if ((size_t)ksteps * step_bytes > slab_max) kchunk = (int)(slab_max / step_bytes) & ~1;  // multiple of the ring depth
  const int ncb = cout16 / MT;
  const int force_nt = tune_int("YL_F8_NT", 0);  // development: pixel tiles per wave for MT <= 4
  const int NT = (force_nt == 2 || force_nt == 4) && MT != 6 ? force_nt : ((npix + 63) / 64 * ncb / 4 >= 256 || MT == 6 ? 4 : 2);
  c->MT = MT;
  c->NT = NT;
  c->ncb = ncb;
  c->cg = cg;
  c->ntaps = ntaps;
  c->uniform = cg % 8 == 0;
  // Plane-major K order (k-group = plane * 9 + tap) for the 3x3 convs on big maps: in tap-major order a workgroup sweeps its
  // input footprint once per tap, nine times, with a reuse distance beyond the 4 MB of an XCD's L2 -- rocprofv3
  // (profiles/x8fp8_r03_summary.md, YOLO11x 1280x1280 batch 8) shows the stride-2 convs fetching 424 MB per launch for
  // 80-160 MB of input and `model.23.cv2.0.0` 402 MB for 79 MB.  Same cure as conv_mfma.hip's: one plane's nine taps back to
  // back (the per-lane tap / plane arithmetic of the non-UNIFORM fetch).  YL_F8_KORD = smallest map (pixels per image) that
  // takes it; 0: never.
  const int kord_px = env_int("YL_F8_KORD", 1600);  // (read at every engine build: the tests switch it)
  c->kord = (kord_px > 0 && k == 3 && (long long)Ho * Wo >= kord_px) ? 1 : 0;
  if (c->kord) c->uniform = 0;
  c->gemm = 0;
  c->reg = c->cs = 0;
  c->ksteps = ksteps;
  // (YL_F8_SLAB2: 1 = a chunked filter alternates between two buffers of half the k-steps, filled by LDS-DMA; 2 = two buffers of the
  // old chunk size; 0 = restaged through registers between two barriers)
  c->slab2 = 0;
  {
    const int mode = tune_int("YL_F8_SLAB2", 1);
    if (mode > 0 && kchunk < ksteps) {
      if (mode == 1) {
        kchunk = (kchunk / 2) & ~1;
        if (kchunk < 2) kchunk = 2;
      }
      c->slab2 = 1;
    }
  }
  c->kchunk = kchunk;
  c->lds_bytes = (size_t)(c->slab2 ? 2 * kchunk : kchunk) * step_bytes + (size_t)MT * 16 * 8;  // + the block's alpha / bias (conv_f8_kernel's epilogue)
  c->post = 0;
  c->wpk_elems = (size_t)ncb * ksteps * MT * 2048;  // bytes
  c->npatches = (int)((npix + 16 * NT - 1) / (16 * NT));
  long long blocks1 = ((long long)c->npatches + 3) / 4 * ncb;
  // Patches per wave so that a launch has ~3 workgroups per CU: each workgroup stages its 40-80 KB filter slab once and the
  // k-loops are short (7-14 steps), so fewer, longer workgroups cost less of the chip per launch.  YOLO11x 1280x1280 batch 8
  // with 20 batches in flight: 8 per CU 1 042-1 053 images/s, 4: 1 058, 3: 1 072, 2: 1 066-1 070 (but one batch at a time
  // 867 -> 820), 1: 1 066.  YL_F8_PPW_DIV overrides.
  static const int ppw_div = std::max(1, tune_int("YL_F8_PPW_DIV", 3));
  int ppw = (int)(blocks1 / (ppw_div * 256));
  if (ppw < 1) ppw = 1;
  if (ppw > 32) ppw = 32;
  if (kchunk < ksteps) ppw = 1;
  c->ppw = ppw;
  c->gws = (int)(((long long)c->npatches + 4 * ppw - 1) / (4 * ppw));
  c->name = k == 3 ? (stride == 2 ? "conv3x3s2_f8_mfma" : "conv3x3_f8_mfma") : "conv1x1_f8_mfma";
  c->f8t = conv3_f8t_fits(*c, batch, Cin, Cout, k, stride, Ho, Wo) ? 1 : 0;
  if (c->f8t) c->name = "conv3x3_f8_tile_mfma";
  return 0;
}

// Folded fp32 OIHW filters -> e4m3 in fragment order [cb][k-step][MT][half][64 lanes][16], and the per-channel scales.
void conv_f8_pack_weights(const float* w, int Cout, int Cin, int k, const ConvConfig& c, std::vector<uint8_t>* out,
                          std::vector<float>* wscale) {
  const int kk = k * k;
  wscale->assign((size_t)c.ncb * c.MT * 16 + 16, 1.0f);
  for (int co = 0; co < Cout; ++co) {
    float m = 0.f;
    for (size_t i = 0; i < (size_t)Cin * kk; ++i) m = std::fmax(m, std::fabs(w[(size_t)co * Cin * kk + i]));
    (*wscale)[co] = m > 0.f ? m / F8_MAX : 1.0f;
  }
  out->assign(c.wpk_elems, 0);
  size_t o = 0;
  for (int cb = 0; cb < c.ncb; ++cb)
    for (int ks = 0; ks < c.ksteps; ++ks)
      for (int mt = 0; mt < c.MT; ++mt)
        for (int h = 0; h < 2; ++h)
          for (int lane = 0; lane < 64; ++lane) {
            const int co = cb * c.MT * 16 + mt * 16 + (lane & 15);
            const int q = ks * 8 + 2 * (lane >> 4) + h;
            const int tap = c.kord ? q % c.ntaps : q / c.cg, c16 = c.kord ? q / c.ntaps : q % c.cg;
            for (int j = 0; j < 16; ++j, ++o) {
              if (co < Cout && tap < c.ntaps && c16 < c.cg) {  // (plane-major: the k tail lies past the last plane)
                const int ci = c16 * 16 + j, kh = tap / k, kw = tap % k;
                (*out)[o] = f32_to_e4m3(w[(((size_t)co * Cin + ci) * k + kh) * k + kw] / (*wscale)[co]);
              }
            }
          }
}

template <int KS, int MT, int NT, bool U>
static int launch8(const ConvConfig& cfg, const ConvArgs& a, hipStream_t s) {
  auto kern = conv_f8_kernel<KS, MT, NT, U>;
  if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), (int)(160 * 1024))) return rc;
  hipLaunchKernelGGL(kern, dim3(a.nblocks), dim3(256), cfg.lds_bytes, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

template <int KS, int MT>
static int launch8_n(const ConvConfig& cfg, const ConvArgs& a, hipStream_t s) {
  if (cfg.NT == 4) return cfg.uniform ? launch8<KS, MT, 4, true>(cfg, a, s) : launch8<KS, MT, 4, false>(cfg, a, s);
  return cfg.uniform ? launch8<KS, MT, 2, true>(cfg, a, s) : launch8<KS, MT, 2, false>(cfg, a, s);
}

template <int KS>
static int launch8_m(const ConvConfig& cfg, const ConvArgs& a, hipStream_t s) {
  switch (cfg.MT) {
    case 2: return launch8_n<KS, 2>(cfg, a, s);
    case 3: return launch8_n<KS, 3>(cfg, a, s);
    case 4: return launch8_n<KS, 4>(cfg, a, s);
    case 6: return cfg.uniform ? launch8<KS, 6, 4, true>(cfg, a, s) : launch8<KS, 6, 4, false>(cfg, a, s);
  }
  return fail(-3, "fp8 conv: bad MT");
}

// `a` arrives with pointers, plane sizes, geometry, Cin / Cout / stride / act, alpha / bias, out_f8 / res_f8 and scales.
int conv_f8_launch(const ConvConfig& cfg, ConvArgs a, int batch, hipStream_t s) {
  a.ncb = cfg.ncb;
  a.cg = cfg.cg;
  a.ksteps = cfg.ksteps;
  a.kchunk = cfg.kchunk;
  a.slab2 = cfg.slab2;
  a.ntaps = cfg.ntaps;
  a.kord = cfg.kord;
  a.kkn = cfg.cg >> 3;
  a.magic_kkn = magic8(cfg.cg >> 3);
  a.magic_cg = magic8(cfg.cg);
  a.magic_wo = magic8(a.Wo);
  a.magic_ho = magic8(a.Ho);
  a.npatches = cfg.npatches;
  a.ppw = cfg.ppw;
  a.npix = batch * a.Ho * a.Wo;
  a.out_f32 = 0;
  const long long lim = 1LL << 30;
  const long long inb = (long long)cfg.cg * a.in_plane;
  const long long outb = (long long)(a.out_f8 ? a.Cout / 16 : a.Cout / 8) * a.out_plane;
  const long long resb = a.res ? (long long)(a.res_f8 ? a.Cout / 16 : a.Cout / 8) * a.res_plane : 0;
  if (inb > lim || outb > lim || resb > lim)
    return fail(-3, "fp8 conv: a tensor slice exceeds 1 GiB (32-bit buffer offsets); lower the batch size");
  a.in_bytes = (unsigned)inb;
  a.out_bytes = a.out8 && a.out8_only ? 0u : (unsigned)outb;  // (copy only: a descriptor of zero bytes drops every bf16 store)
  a.res_bytes = (unsigned)resb;
  a.nblocks = cfg.gws * cfg.ncb;
  a.dbg = 0;
  a.st_nt = st_nt_policy(outb);
  a.ab_off = (int)(cfg.lds_bytes - (size_t)cfg.MT * 16 * 8);
  if (cfg.f8t) return conv3_f8t_launch(cfg, a, batch, s);
  return cfg.ntaps == 9 ? launch8_m<3>(cfg, a, s) : launch8_m<1>(cfg, a, s);
}

}  // namespace yl
