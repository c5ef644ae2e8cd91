// Bandwidth-bound bf16 kernels of the YOLO11 path over the C8-planar guarded layout (yl_internal.h):
// input conversion, fused stem, depthwise 3x3, SPPF max-pool chain, nearest upsample / slice copy,
// C2PSA attention, and the Detect decode.  Every thread moves 16-byte pixels (8 x bf16 of one plane)
// and neighbouring lanes touch neighbouring pixels of the same plane.
#include "conv_common.h"

namespace yl {

__device__ __forceinline__ float silu_g(float v) {  // hardware exp2 / rcp, see conv_mfma.hip
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * v));
}

__device__ __forceinline__ int pv_pix(const PView& v, int b, int y, int x) { return v.g0 + b * v.img + y * v.W1 + x; }
__device__ __forceinline__ __bf16* pv_ptr(const PView& v, int c8, int pix) {
  return v.base + (long long)c8 * v.plane + (long long)pix * 8;
}

// ---------------------------------------------------------- input NCHW -> one bf16 plane (C = 8)
// engine/predictor.py:67-85 (preprocess: .float(), /255 for uint8 sources); channels 3..7 are zero.
// Only used when layer 0 is not the standard 3x3 stride-2 stem (otherwise see stem_kernel).
template <typename T>
__global__ void convert_input_kernel(const T* __restrict__ in, PView out, int H, int W, float mul, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int HW = H * W;
  const int b = (int)(i / HW);
  const int p = (int)(i - (long long)b * HW);
  const int y = p / W, x = p - y * W;
  const T* src = in + (long long)b * 3 * HW + p;
  bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
  // `mul` is 255 for uint8 sources: a true fp32 division like `im /= 255` (predictor.py:84)
  v[0] = (__bf16)((float)src[0] / mul);
  v[1] = (__bf16)((float)src[HW] / mul);
  v[2] = (__bf16)((float)src[2 * (long long)HW] / mul);
  *reinterpret_cast<bf16x8*>(pv_ptr(out, 0, pv_pix(out, b, y, x))) = v;
}

int convert_input_launch(const void* in, int in_dtype, const PView& out, int batch, int H, int W, hipStream_t s) {
  const long long total = (long long)batch * H * W;
  const int blocks = (int)((total + 255) / 256);
  if (in_dtype == YL_F32)
    hipLaunchKernelGGL(convert_input_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)in, out, H, W, 1.0f,
                       total);
  else if (in_dtype == YL_F16)
    hipLaunchKernelGGL(convert_input_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, (const _Float16*)in, out, H, W,
                       1.0f, total);
  else if (in_dtype == YL_U8)
    hipLaunchKernelGGL(convert_input_kernel<uint8_t>, dim3(blocks), dim3(256), 0, s, (const uint8_t*)in, out, H, W,
                       255.0f, total);
  else
    return fail(-3, "convert_input: unsupported input dtype");
  YL_HIP(hipGetLastError());
  return 0;
}

// --------------------------------------------------------------------------------- fused stem
// Layer 0 of yolo11.yaml (Conv 3 -> c, k 3, s 2; conv.py:35-53) fused with the predictor's input
// conversion (predictor.py:67-85): reads the NCHW network input directly (fp32 / fp16 / uint8) and
// writes NHWC bf16, so the 8-channel padded NHWC copy of the input is never written or re-read.
//
// One workgroup = one output row segment of one image (up to 64 pixels per wave).
//   load   : the 3 channels x 3 input rows the segment needs are read as contiguous element PAIRS
//            (ix = 2ox, 2ox+1: fully coalesced; stride-2 dword loads run at about one lane per clock on
//            this TA), rounded to bf16 and parked in LDS with the left halo column and zero padding.
//   compute: implicit GEMM on v_mfma_f32_16x16x32_bf16 with K = 27 taps padded to 32: A = filters
//            (rows = 16 output channels), B = im2col patch gathered from LDS (cols = 16 pixels).
//   store  : D gives each lane 4 consecutive channels of one pixel -> 8-byte stores; the two lane
//            quads of a plane fill whole 16-byte pixels, 256 contiguous bytes per plane.
// Numerics as the generic MFMA conv (bf16 operands, fp32 accumulate).  Bandwidth bound: 157 MB in,
// 105 MB out at batch 32 x 640^2 (fp32 input, 16 channels).
constexpr int STEM_MAXW = 8;  // waves per workgroup -> up to 512 output pixels per row segment

template <typename T>
struct StemPair;
template <>
struct StemPair<float> {
  typedef __attribute__((ext_vector_type(2))) float type;
};
template <>
struct StemPair<_Float16> {
  typedef __attribute__((ext_vector_type(2))) _Float16 type;
};
template <>
struct StemPair<uint8_t> {
  typedef __attribute__((ext_vector_type(2))) uint8_t type;
};

template <typename T>
__device__ __forceinline__ float stem_cvt(T v) {
  float f = (float)v;
  if constexpr (sizeof(T) == 1) f = f / 255.0f;  // true division, as predictor.py:84
  return f;
}

template <typename T>
__global__ __launch_bounds__(64 * STEM_MAXW) void stem_kernel(StemArgs a, const float* __restrict__ w,
                                                              const float* __restrict__ bias) {
  extern __shared__ __align__(16) unsigned char stem_smem[];
  __bf16* tile = reinterpret_cast<__bf16*>(stem_smem);  // [9 (c,kh)][row] ; col' = ix - 2*ox0 + 2
  typedef typename StemPair<T>::type pair_t;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int row = 2 * nthr + 8;  // bf16 elements per LDS row (even)
  // XCD remap: neighbouring output rows share an input row -- keep them on one L2 (measured with
  // FETCH_SIZE: 1.5x the algorithmic reads when consecutive rows were dealt to different XCDs)
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int seg = bid % a.nseg;
  const int oy = (bid / a.nseg) % a.Ho;
  const int b = bid / (a.nseg * a.Ho);
  const int ox0 = seg * nthr;
  const T* in = reinterpret_cast<const T*>(a.in) + (long long)b * 3 * a.H * a.W;

  // ---- load phase: 9 coalesced pair loads per thread, then the halo column
  {
    const int ix0 = 2 * (ox0 + tid);
    pair_t v[9];
    const long long img_bytes = 3LL * a.H * a.W * (long long)sizeof(T);
    if ((a.W & 1) == 0 && img_bytes < (1LL << 30)) {
      // usual case: branch-free buffer loads; rows above / below the image and pairs right of it
      // resolve to out-of-range offsets (zeros)
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc((void*)in, (short)0, (int)img_bytes, 0x00020000);
      const unsigned lane_off = ix0 < a.W ? (unsigned)ix0 * (unsigned)sizeof(T) : 0x80000000u;
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        const int c = r / 3, kh = r % 3;
        const int iy = 2 * oy + kh - 1;  // block-uniform
        const unsigned row_off =
            (unsigned)iy < (unsigned)a.H ? (unsigned)((c * a.H + iy) * a.W) * (unsigned)sizeof(T) : 0x40000000u;
        if constexpr (sizeof(T) == 4)
          v[r] = __builtin_bit_cast(pair_t, __builtin_amdgcn_raw_buffer_load_b64(rs, row_off + lane_off, 0, 0));
        else if constexpr (sizeof(T) == 2)
          v[r] = __builtin_bit_cast(pair_t, __builtin_amdgcn_raw_buffer_load_b32(rs, row_off + lane_off, 0, 0));
        else
          v[r] = __builtin_bit_cast(pair_t, __builtin_amdgcn_raw_buffer_load_b16(rs, row_off + lane_off, 0, 0));
      }
    } else {
#pragma unroll
      for (int r = 0; r < 9; ++r) {
        const int c = r / 3, kh = r % 3;
        const int iy = 2 * oy + kh - 1;
        const T* rp = in + ((long long)c * a.H + iy) * a.W;
        pair_t q = {(T)0, (T)0};
        if ((unsigned)iy < (unsigned)a.H) {
          if (ix0 < a.W) q[0] = rp[ix0];
          if (ix0 + 1 < a.W) q[1] = rp[ix0 + 1];
        }
        v[r] = q;
      }
    }
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
      const bf16x2 o = {(__bf16)stem_cvt<T>(v[r][0]), (__bf16)stem_cvt<T>(v[r][1])};
      *reinterpret_cast<bf16x2*>(tile + r * row + 2 * tid + 2) = o;
    }
    if (tid < 9) {
      const int c = tid / 3, kh = tid % 3;
      const int iy = 2 * oy + kh - 1, ix = 2 * ox0 - 1;
      float f = 0.f;
      if ((unsigned)iy < (unsigned)a.H && ix >= 0) f = stem_cvt<T>(in[((long long)c * a.H + iy) * a.W + ix]);
      tile[tid * row + 1] = (__bf16)f;
    }
  }
  __syncthreads();

  // ---- compute phase
  const int lane = tid & 63, wave = tid >> 6;
  const int p = lane & 15, kg = lane >> 4;
  const int px0 = wave * 64 + p;  // this lane's pixel in n-tile 0; n-tile g adds 16 pixels
  // LDS element of tap k for pixel px0 (k = (c*3 + kh)*3 + kw; k / 3 = (k * 11) >> 5 for k < 32);
  // k >= 27 reads any initialised (finite) element, its filter value is 0.  The n-tile offset is a
  // constant, so the eight gathers of a tile are ds_read_u16 with immediate offsets.
  const __bf16* tp[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = kg * 8 + j;
    const int r = (k * 11) >> 5;
    tp[j] = tile + (k < 27 ? r * row + (k - 3 * r) + 1 : 2) + 2 * px0;
  }
  const int ox_lane = ox0 + px0;
  const int pix0 = pv_pix(a.out, b, oy, ox_lane);
  for (int mb = 0; mb < a.Cout; mb += 16) {
    bf16x8 af;
    if (a.wpk) {  // (round 5: one 16-byte load per lane -- the eight dword gathers below touch ~16 cache lines per instruction, 48 of
                  // them per wave at 96 output channels: a third of the YOLO11x stem's time)
      af = *reinterpret_cast<const bf16x8*>(a.wpk + ((size_t)(mb >> 4) * 64 + lane) * 8);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = kg * 8 + j;
        af[j] = (__bf16)(k < 27 ? w[(mb + p) * 27 + k] : 0.f);
      }
    }
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + mb + kg * 4);
    const int ch = mb + kg * 4;
    __bf16* op0 = pv_ptr(a.out, ch >> 3, pix0) + (ch & 7);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x8 bf;
#pragma unroll
      for (int j = 0; j < 8; ++j) bf[j] = tp[j][32 * g];
      f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, bv, 0, 0, 0);
      if (ox_lane + 16 * g < a.Wo) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (__bf16)(a.act ? silu_g(acc[r]) : acc[r]);
        if (!a.out8_only) *reinterpret_cast<bf16x4*>(op0 + 128 * g) = o;
        if (a.out8) {  // e4m3 copy (of the bf16-rounded values, as a quantize pass over the stored tensor would see them)
          const float is = a.out8_inv_scale;
          *reinterpret_cast<unsigned*>(a.out8 + (long long)(ch >> 4) * a.out8_plane + (long long)(pix0 + 16 * g) * 16 + (ch & 15)) =
              pack_f8x4((float)o[0] * is, (float)o[1] * is, (float)o[2] * is, (float)o[3] * is);
        }
      }
    }
  }
}

int stem_launch(const StemArgs& a0, int in_dtype, hipStream_t s) {
  if (a0.Cout % 16) return fail(-3, "stem: output channels must be a multiple of 16");
  StemArgs a = a0;
  // row segments of equal size: the fewest that fit STEM_MAXW waves, then the fewest waves that cover one (640 output columns
  // used to be 512 + 128 -- the second workgroup of every row ran six of its eight waves on nothing: 2 x 5 waves now)
  const int nseg0 = (a.Wo + 64 * STEM_MAXW - 1) / (64 * STEM_MAXW);
  const int waves = std::max(1, ((a.Wo + nseg0 - 1) / nseg0 + 63) / 64);
  const int nthr = 64 * waves;
  a.nseg = (a.Wo + nthr - 1) / nthr;
  const unsigned blocks = (unsigned)(a.B * a.Ho * a.nseg);
  const size_t lds = (size_t)9 * (2 * nthr + 8) * sizeof(__bf16);
  if (in_dtype == YL_F32)
    hipLaunchKernelGGL(stem_kernel<float>, dim3(blocks), dim3(nthr), lds, s, a, a.w, a.bias);
  else if (in_dtype == YL_F16)
    hipLaunchKernelGGL(stem_kernel<_Float16>, dim3(blocks), dim3(nthr), lds, s, a, a.w, a.bias);
  else if (in_dtype == YL_U8)
    hipLaunchKernelGGL(stem_kernel<uint8_t>, dim3(blocks), dim3(nthr), lds, s, a, a.w, a.bias);
  else
    return fail(-3, "stem: unsupported input dtype");
  YL_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------ depthwise 3x3
// nn/modules/conv.py:100-105 (DWConv, groups = C) with folded BN; optional SiLU; optional
// accumulate (x = attn_out + pe(v), nn/modules/block.py:913) and the per-head v channel map.
// One thread = one plane (8 channels) x 4 consecutive output pixels of a row: the 3 x 6 input window
// and the 9 x 8 filter taps are loaded once and reused (4.5 sixteen-byte loads per output instead of
// 9).  The zero padding comes from the guard pixels of the layout: no bounds tests on the taps.
// Thread order: pixel runs fastest, planes slowest -> a wave reads contiguous memory.
template <int DW_PX>
__global__ __launch_bounds__(256) void dwconv3_kernel(DwArgs a) {
  const unsigned wq = (unsigned)(a.W + DW_PX - 1) / DW_PX;
  const unsigned per_plane = (unsigned)a.batch * a.H * wq;  // < 2^31 (host-checked)
  // XCD remap: neighbouring row runs (which share input rows) stay on one L2
  unsigned p = (unsigned)xcd_remap(blockIdx.x, gridDim.x) * 256u + threadIdx.x;
  if (p >= per_plane) return;
  const int g8 = blockIdx.y;  // plane: wave-uniform, so the filter / bias loads are scalar
  const unsigned row = p / wq;
  const int x0 = (int)(p - row * wq) * DW_PX;
  const int b = (int)(row / (unsigned)a.H);
  const int y = (int)(row - (unsigned)b * a.H);
  const int c = g8 * 8;
  const int cin = a.grp ? (c / a.grp) * a.grp_stride + a.grp_off + (c % a.grp) : c;
  const __bf16* inb = pv_ptr(a.in, cin >> 3, pv_pix(a.in, b, y, x0));
  // all 3 x (DW_PX + 2) input pixels are requested before the first use: one memory round trip
  bf16x8 v[3][DW_PX + 2];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int t = 0; t < DW_PX + 2; ++t) {
      // columns beyond the guard (ragged last run only) are clamped onto the guard: zeros
      const int dx = min(t - 1, a.W - x0);
      v[kh][t] = *reinterpret_cast<const bf16x8*>(inb + ((kh - 1) * a.in.W1 + dx) * 8);
    }
  // (round 5) channel PAIRS as two-element vectors: __builtin_elementwise_fma on them is ONE v_pk_fma_f32 per pair -- the same IEEE
  // fused multiply-add on the same operands in the same order as 72 scalar v_fma_f32 per pixel, in half the instructions
  typedef __attribute__((ext_vector_type(2))) float dw_f2;
  dw_f2 acc[DW_PX][4];
  {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + c), b1 = *reinterpret_cast<const f32x4*>(a.bias + c + 4);
#pragma unroll
    for (int q = 0; q < DW_PX; ++q) {
      acc[q][0] = dw_f2{b0[0], b0[1]};
      acc[q][1] = dw_f2{b0[2], b0[3]};
      acc[q][2] = dw_f2{b1[0], b1[1]};
      acc[q][3] = dw_f2{b1[2], b1[3]};
    }
  }
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    dw_f2 wrow[3][4];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const float* wp = a.w + (kh * 3 + kw) * a.C + c;
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp), w1 = *reinterpret_cast<const f32x4*>(wp + 4);
      wrow[kw][0] = dw_f2{w0[0], w0[1]};
      wrow[kw][1] = dw_f2{w0[2], w0[3]};
      wrow[kw][2] = dw_f2{w1[0], w1[1]};
      wrow[kw][3] = dw_f2{w1[2], w1[3]};
    }
#pragma unroll
    for (int t = 0; t < DW_PX + 2; ++t) {  // input column x0 - 1 + t feeds outputs q = t - kw
      dw_f2 f[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) f[j] = dw_f2{(float)v[kh][t][2 * j], (float)v[kh][t][2 * j + 1]};
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int q = t - kw;
        if (q >= 0 && q < DW_PX) {
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[q][j] = __builtin_elementwise_fma(f[j], wrow[kw][j], acc[q][j]);
        }
      }
    }
  }
  __bf16* ob = pv_ptr(a.out, g8, pv_pix(a.out, b, y, x0));
#pragma unroll
  for (int q = 0; q < DW_PX; ++q) {
    if (x0 + q >= a.W) break;
    __bf16* op = ob + q * 8;
    bf16x8 o;
    if (a.accumulate) {
      const bf16x8 prev = *reinterpret_cast<const bf16x8*>(op);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (__bf16)((a.act ? silu_g(acc[q][j >> 1][j & 1]) : acc[q][j >> 1][j & 1]) + (float)prev[j]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (__bf16)(a.act ? silu_g(acc[q][j >> 1][j & 1]) : acc[q][j >> 1][j & 1]);
    }
    *reinterpret_cast<bf16x8*>(op) = o;
  }
}

template <int PX>
static void dw_launch_px(const DwArgs& a, hipStream_t s) {
  const long long per_plane = (long long)a.batch * a.H * ((a.W + PX - 1) / PX);
  hipLaunchKernelGGL(dwconv3_kernel<PX>, dim3((unsigned)((per_plane + 255) / 256), a.C / 8), dim3(256), 0, s, a);
}

int dwconv3_launch(const DwArgs& a, hipStream_t s) {
  if (a.C % 8 || (a.grp && (a.grp % 8 || a.grp_stride % 8 || a.grp_off % 8)))
    return fail(-3, "dwconv3: channel counts must be multiples of 8");
  if ((long long)a.batch * a.H * a.W >= (1LL << 31) || a.C / 8 > 65535) return fail(-3, "dwconv3: tensor too large");
  static const int px = tune_int("YL_DW_PX", 2);
  if (px == 1) dw_launch_px<1>(a, s);
  else if (px == 2) dw_launch_px<2>(a, s);
  else dw_launch_px<4>(a, s);
  YL_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------- SPPF pooling chain
// nn/modules/block.py:180-184: y1 = m(x), y2 = m(y1), y3 = m(y2) with m = MaxPool2d(5,1,2).
// Chained 5x5 max pools with -inf padding equal single 5/9/13 windows clipped to the image (the zero
// guards must NOT take part: SiLU outputs can be negative).  Pool k of plane g8 goes to plane
// k * C/8 + g8 of the same buffer.
__global__ void sppf_pool_kernel(PoolArgs a) {
  const int cg = a.C >> 3;
  const long long per_plane = (long long)a.batch * a.H * a.W;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per_plane * cg) return;
  const int g8 = (int)(i / per_plane);
  long long p = i - g8 * per_plane;
  const int x = (int)(p % a.W);
  p /= a.W;
  const int y = (int)(p % a.H);
  const int b = (int)(p / a.H);
  const float ninf = -__builtin_huge_valf();
  float m5[8], m9[8], m13[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) m5[j] = m9[j] = m13[j] = ninf;
  for (int dy = -6; dy <= 6; ++dy) {
    const int iy = y + dy;
    if ((unsigned)iy >= (unsigned)a.H) continue;
    const int ady = dy < 0 ? -dy : dy;
    for (int dx = -6; dx <= 6; ++dx) {
      const int ix = x + dx;
      if ((unsigned)ix >= (unsigned)a.W) continue;
      const int adx = dx < 0 ? -dx : dx;
      const int r = ady > adx ? ady : adx;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(pv_ptr(a.v, g8, pv_pix(a.v, b, iy, ix)));
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)v[j];
        m13[j] = fmaxf(m13[j], f);
        if (r <= 4) m9[j] = fmaxf(m9[j], f);
        if (r <= 2) m5[j] = fmaxf(m5[j], f);
      }
    }
  }
  const int po = pv_pix(a.v, b, y, x);
  bf16x8 o5, o9, o13;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    o5[j] = (__bf16)m5[j];
    o9[j] = (__bf16)m9[j];
    o13[j] = (__bf16)m13[j];
  }
  *reinterpret_cast<bf16x8*>(pv_ptr(a.v, cg + g8, po)) = o5;
  *reinterpret_cast<bf16x8*>(pv_ptr(a.v, 2 * cg + g8, po)) = o9;
  *reinterpret_cast<bf16x8*>(pv_ptr(a.v, 3 * cg + g8, po)) = o13;
}

// LDS version: one workgroup = one image x one plane; the H x W map is staged once (16 bytes per
// pixel), row maxima for the 5/9/13 windows are built from it, then column maxima of those.
// 13 + 27 LDS reads per pixel instead of 169 global loads.
__device__ __forceinline__ bf16x8 max8(bf16x8 a, bf16x8 b) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (float)a[j] > (float)b[j] ? a[j] : b[j];
  return o;
}

// bf16 -> int16 key that orders like the value (sign-magnitude -> two's complement: flip the magnitude bits of
// negatives); its own inverse.  A maximum of bf16 values is then ONE v_pk_max_i16 per two values instead of two
// conversions, a compare and a select per value: the fp32-compare form made this kernel VALU bound (~30
// instructions per 8-channel maximum, 40 maxima per pixel).  -0 orders below +0 (a float compare calls them
// equal; the values are the same); NaNs are not expected in activations.
typedef short i16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ i16x8 ord_key(i16x8 x) { return x ^ ((x >> 15) & (short)0x7fff); }
__device__ __forceinline__ i16x8 kmax(i16x8 a, i16x8 b) { return __builtin_elementwise_max(a, b); }

__global__ __launch_bounds__(256) void sppf_pool_lds_kernel(PoolArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
  const int HW = a.H * a.W;
  i16x8* sx = reinterpret_cast<i16x8*>(psm);  // [HW] input, as ordered keys
  i16x8* sh = sx + HW;                        // [3][HW] horizontal maxima
  const int cg = a.C >> 3;
  const int b = blockIdx.x / cg, g8 = blockIdx.x % cg;
  for (int p = threadIdx.x; p < HW; p += 256) {
    const int y = p / a.W, x = p - y * a.W;
    sx[p] = ord_key(*reinterpret_cast<const i16x8*>(pv_ptr(a.v, g8, pv_pix(a.v, b, y, x))));
  }
  __syncthreads();
  for (int p = threadIdx.x; p < HW; p += 256) {
    const int y = p / a.W, x = p - y * a.W;
    const i16x8 m = sx[p];
    i16x8 m5 = m, m9 = m, m13 = m;
#pragma unroll
    for (int d = 1; d <= 6; ++d) {
      i16x8 t = m;
      if (x - d >= 0) t = kmax(t, sx[p - d]);
      if (x + d < a.W) t = kmax(t, sx[p + d]);
      m13 = kmax(m13, t);
      if (d <= 4) m9 = kmax(m9, t);
      if (d <= 2) m5 = kmax(m5, t);
    }
    sh[p] = m5;
    sh[HW + p] = m9;
    sh[2 * HW + p] = m13;
  }
  __syncthreads();
  for (int p = threadIdx.x; p < HW; p += 256) {
    const int y = p / a.W, x = p - y * a.W;
    i16x8 o[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int r = 2 + 2 * k;
      const i16x8* hk = sh + k * HW;
      i16x8 m = hk[p];
      for (int d = 1; d <= r; ++d) {
        if (y - d >= 0) m = kmax(m, hk[p - d * a.W]);
        if (y + d < a.H) m = kmax(m, hk[p + d * a.W]);
      }
      o[k] = ord_key(m);
    }
    const int po = pv_pix(a.v, b, y, x);
#pragma unroll
    for (int k = 0; k < 3; ++k) *reinterpret_cast<i16x8*>(pv_ptr(a.v, (k + 1) * cg + g8, po)) = o[k];
  }
}

int sppf_pool_launch(const PoolArgs& a, hipStream_t s) {
  if (a.C % 8) return fail(-3, "sppf_pool: channels must be a multiple of 8");
  const size_t lds = (size_t)a.H * a.W * 16 * 4;
  if (lds <= 150 * 1024) {
    static size_t attr = 0;
    if (lds > attr) {
      YL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sppf_pool_lds_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr = 160 * 1024;
    }
    hipLaunchKernelGGL(sppf_pool_lds_kernel, dim3(a.batch * (a.C / 8)), dim3(256), lds, s, a);
  } else {
    const long long total = (long long)a.batch * a.H * a.W * (a.C / 8);
    hipLaunchKernelGGL(sppf_pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
  }
  YL_HIP(hipGetLastError());
  return 0;
}

// -------------------------------------------------------------- nearest upsample / slice copy
// nn.Upsample(None, 2, "nearest") of yolo11.yaml:31,35 written straight into the Concat buffer.
__global__ void upsample_kernel(UpArgs a) {
  const unsigned Ho = a.H * a.factor, Wo = a.W * a.factor;
  const unsigned per_plane = (unsigned)a.batch * Ho * Wo;  // < 2^31 (host-checked)
  const unsigned p = blockIdx.x * 256u + threadIdx.x;
  if (p >= per_plane) return;
  const int g8 = blockIdx.y;
  const unsigned row = p / Wo;
  const int x = (int)(p - row * Wo);
  const int b = (int)(row / Ho);
  const int y = (int)(row - (unsigned)b * Ho);
  const bf16x8 v =
      *reinterpret_cast<const bf16x8*>(pv_ptr(a.in, g8, pv_pix(a.in, b, y / a.factor, x / a.factor)));
  *reinterpret_cast<bf16x8*>(pv_ptr(a.out, g8, pv_pix(a.out, b, y, x))) = v;
}

int upsample_launch(const UpArgs& a, hipStream_t s) {
  if (a.C % 8) return fail(-3, "upsample: channels must be a multiple of 8");
  const long long per_plane = (long long)a.batch * a.H * a.factor * a.W * a.factor;
  if (per_plane >= (1LL << 31) || a.C / 8 > 65535) return fail(-3, "upsample: tensor too large");
  hipLaunchKernelGGL(upsample_kernel, dim3((unsigned)((per_plane + 255) / 256), a.C / 8), dim3(256), 0, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------- attention
// nn/modules/block.py:905-912: per head, attn = softmax((q^T k) * scale) over keys;
// out[:, i] = sum_j v[:, j] * attn[i, j].  qkv channel layout per head: [q(32) | k(32) | v(64)].
//
// MFMA flash-style kernel (v_mfma_f32_16x16x32_bf16), one workgroup per (image, head, query half):
//   S^T tile  = K_tile (16 keys x 32) . Q^T (32 x 16 queries)   -> lane holds 4 keys of ITS query
//   O^T tile += V^T (16 d x 32 keys) . P^T (32 keys x 16 queries)
// The S^T accumulators of two 16-key tiles are, after exp(), exactly the B fragment of the second
// product when MFMA k-slot 8g+j is read as key (j<4 ? 4g+j : 16+4g+j-4); V^T is staged in LDS with
// the keys contiguous so the matching A fragment is two 8-byte LDS reads.  Online softmax per
// query column: the running max is shared by the 4 lanes of a column through two shuffles.
constexpr int AT_KD = 32, AT_HD = 64;
constexpr int AT_STAGE = 512;          // keys resident in LDS at a time
constexpr int AT_KP = AT_KD + 8;       // K row pitch (elements): 80-byte rows
constexpr int AT_VP = AT_STAGE + 8;    // V^T row pitch (elements)
constexpr int AT_WAVES = 8;  // waves per workgroup: 4 x 8 = 32 query-tile slots per (image, head) cover the 25 tiles of a
                             // 20x20 map in ONE round (4 waves needed two, the second 9/16 full)

// R = query tiles per wave.  The key / value stages are the OUTER loop and a wave carries the softmax state of its R query
// tiles through them, so K and V^T are staged once per workgroup however many rounds it has (round 3; before, every round
// restaged every stage: 16 stagings per workgroup at 40x40 = 1600 tokens, YOLO11x at 1280x1280: 187 us).  Per query tile
// the keys are visited in the same order as before: results are bit-identical.
template <int R>
__global__ __launch_bounds__(64 * AT_WAVES) __attribute__((amdgpu_waves_per_eu(2))) void attention_kernel(AttnArgs a, int qsplit) {
  extern __shared__ __attribute__((aligned(16))) unsigned char asm_[];
  __bf16* sK = reinterpret_cast<__bf16*>(asm_);   // [AT_STAGE][AT_KP]
  __bf16* sVt = sK + AT_STAGE * AT_KP;            // [64][AT_VP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  int bid = blockIdx.x;
  const int qs = bid % qsplit;
  bid /= qsplit;
  const int head = bid % a.heads;
  const int b = bid / a.heads;
  const int hstride = 2 * AT_KD + AT_HD;
  const int hplane = head * hstride / 8;  // first plane of this head's [q | k | v] block
  // token t of image b -> pixel index inside a plane
  auto tok = [&](int t) {
    const int y = t / a.W;
    return pv_pix(a.in, b, y, t - y * a.W);
  };
  const int nqt = (a.N + 15) / 16;                       // query tiles
  const int nstage = (a.N + AT_STAGE - 1) / AT_STAGE;
  const int wslot = qs * AT_WAVES + wave;                // this wave takes q tiles wslot, wslot + AT_WAVES * qsplit, ...
  const float ninf = -__builtin_huge_valf();

  bool active[R];
  int q[R];
  bf16x8 qf[R];
  float m[R], l[R];
  f32x4 acc[R][4];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int qt = wslot + r * AT_WAVES * qsplit;
    active[r] = qt < nqt;
    q[r] = qt * 16 + n;
    qf[r] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (active[r] && q[r] < a.N) qf[r] = *reinterpret_cast<const bf16x8*>(pv_ptr(a.in, hplane + g, tok(q[r])));
    m[r] = ninf;
    l[r] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[r][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  for (int st = 0; st < nstage; ++st) {
    const int k0s = st * AT_STAGE;
    const int nk = min(AT_STAGE, a.N - k0s);
    const int nk32 = (nk + 31) & ~31;
    if (st) __syncthreads();  // previous users of the stage buffers are done
    // loads are issued in batches of 8 per thread before the first LDS write (one memory round
    // trip per batch instead of one per vector)
    for (int i0 = tid; i0 < nk32 * 4; i0 += 64 * AT_WAVES * 8) {
      bf16x8 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 64 * AT_WAVES, key = i >> 2, c8 = i & 3;
        v[u] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (i < nk32 * 4 && key < nk)
          v[u] = *reinterpret_cast<const bf16x8*>(pv_ptr(a.in, hplane + AT_KD / 8 + c8, tok(k0s + key)));
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 64 * AT_WAVES, key = i >> 2, c8 = i & 3;
        if (i < nk32 * 4) *reinterpret_cast<bf16x8*>(sK + key * AT_KP + c8 * 8) = v[u];
      }
    }
    // V^T: a thread takes 8 CONSECUTIVE keys of one plane (8 channels), transposes the 8 x 8 block in registers and writes
    // eight 16-byte rows (before: 64 two-byte LDS writes per thread and batch)
    for (int it = tid; it < nk32; it += 64 * AT_WAVES) {  // item = (key group of 8, plane)
      const int kg = it >> 3, c8 = it & 7;
      bf16x8 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int key = kg * 8 + u;
        v[u] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (key < nk) v[u] = *reinterpret_cast<const bf16x8*>(pv_ptr(a.in, hplane + 2 * AT_KD / 8 + c8, tok(k0s + key)));
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bf16x8 wv = {v[0][j], v[1][j], v[2][j], v[3][j], v[4][j], v[5][j], v[6][j], v[7][j]};
        *reinterpret_cast<bf16x8*>(sVt + (c8 * 8 + j) * AT_VP + kg * 8) = wv;
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (!active[r]) continue;
      for (int k0 = 0; k0 < nk32; k0 += 32) {
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(sK + (k0 + n) * AT_KP + g * 8);
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(sK + (k0 + 16 + n) * AT_KP + g * 8);
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, qf[r], z, 0, 0, 0);
        const f32x4 s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, qf[r], z, 0, 0, 0);
        float v[8];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          v[rr] = (k0 + 4 * g + rr < nk) ? s0[rr] * a.scale : ninf;
          v[4 + rr] = (k0 + 16 + 4 * g + rr < nk) ? s1[rr] * a.scale : ninf;
        }
        float lm = v[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) lm = fmaxf(lm, v[j]);
        lm = fmaxf(lm, __shfl_xor(lm, 16));
        lm = fmaxf(lm, __shfl_xor(lm, 32));
        const float mn = fmaxf(m[r], lm);  // finite: every processed step has at least one real key
        const float alpha = __expf(m[r] - mn);
        float ps = 0.f;
        bf16x8 pb;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float e = __expf(v[j] - mn);
          ps += e;
          pb[j] = (__bf16)e;
        }
        l[r] = l[r] * alpha + ps;
        m[r] = mn;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const __bf16* vr = sVt + (dt * 16 + n) * AT_VP + k0 + 4 * g;
          const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vr);
          const bf16x4 hi = *reinterpret_cast<const bf16x4*>(vr + 16);
          const bf16x8 av = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) acc[r][dt][rr] *= alpha;
          acc[r][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, pb, acc[r][dt], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (!active[r]) continue;
    float lr = l[r];
    lr += __shfl_xor(lr, 16);
    lr += __shfl_xor(lr, 32);
    if (q[r] < a.N) {
      const float inv = 1.0f / lr;
      const int qy = q[r] / a.W;
      const int po = pv_pix(a.out, b, qy, q[r] - qy * a.W);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *reinterpret_cast<bf16x4*>(pv_ptr(a.out, (head * AT_HD + dt * 16 + 4 * g) >> 3, po) + ((4 * g) & 7)) =
            bf16x4{(__bf16)(acc[r][dt][0] * inv), (__bf16)(acc[r][dt][1] * inv), (__bf16)(acc[r][dt][2] * inv),
                   (__bf16)(acc[r][dt][3] * inv)};
    }
  }
}

template <int R>
static int launch_attention(const AttnArgs& a, int qsplit, size_t lds, hipStream_t s) {
  auto kern = attention_kernel<R>;
  if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), (int)(160 * 1024))) return rc;
  hipLaunchKernelGGL(kern, dim3(a.batch * a.heads * qsplit), dim3(64 * AT_WAVES), lds, s, a, qsplit);
  YL_HIP(hipGetLastError());
  return 0;
}

int attention_launch(const AttnArgs& a, hipStream_t s) {
  const size_t lds = ((size_t)AT_STAGE * AT_KP + (size_t)AT_HD * AT_VP) * 2;
  // workgroups per (image, head): four; more only when a wave would carry more than four query tiles (every workgroup
  // stages ALL keys and values of its head: 40x40 tokens, YOLO11x batch 8: 4 workgroups 105 us per launch, 7: 150 us)
  const int nqt = (a.N + 15) / 16;
  static const int qs_env = tune_int("YL_ATTN_QSPLIT", 0);  // development
  int qsplit = qs_env > 0 ? qs_env : 4;
  auto rounds_of = [&](int qs) { return (nqt + AT_WAVES * qs - 1) / (AT_WAVES * qs); };
  while (qs_env <= 0 && rounds_of(qsplit) > 4 && qsplit < 16) ++qsplit;
  const int rounds = rounds_of(qsplit);
  if (rounds <= 1) return launch_attention<1>(a, qsplit, lds, s);
  if (rounds == 2) return launch_attention<2>(a, qsplit, lds, s);
  if (rounds <= 4) return launch_attention<4>(a, qsplit, lds, s);
  if (rounds <= 8) return launch_attention<8>(a, qsplit, lds, s);
  return fail(-3, "attention: more than 16384 tokens per image");
}

// ---------------------------------------------------------------------------- Detect decode
// nn/modules/head.py:95-126 (_inference): DFL (block.py:66-69: softmax over 16 bins, expectation),
// dist2bbox xywh (utils/tal.py:341-350) on anchors (x+0.5, y+0.5) (tal.py:326-338), times stride,
// and sigmoid of the class logits.  Level maps are bf16 planes (yl_level); arithmetic and output are
// fp32: y[B][4+nc][A].  One thread = one anchor: its logits are 8 + nc8/8 sixteen-byte pixels, lanes of
// a wave read neighbouring pixels of one plane and write neighbouring columns of one row of y.
constexpr int MAX_LEVELS = 8;
struct DecodeArgs {
  DecodeLevel lv[MAX_LEVELS];
  int n_levels, batch, nc, A;
  float* y;
};

__global__ __launch_bounds__(256) void decode_kernel(DecodeArgs a) {
  const int an = blockIdx.x * 256 + threadIdx.x;  // anchor inside the image
  const int b = blockIdx.y;
  if (an >= a.A) return;
  int l = 0;
#pragma unroll
  for (int i = 1; i < MAX_LEVELS; ++i)
    if (i < a.n_levels && an >= a.lv[i].a0) l = i;
  const DecodeLevel L = a.lv[l];
  const int p = an - L.a0, py = p / L.W, px = p - py * L.W;
  const int pix = pv_pix(L.box, b, py, px);
  float4 bx;
  if (L.xywh) {  // decoded by the epilogue of the conv that computed the DFL logits (yl_level.offset_xywh)
    bx = level_xywh(L, b, p);
  } else {
    float dist[4];
#pragma unroll
    for (int side = 0; side < 4; ++side) {
      float v[16];
      float lo[8], hi[8];
      level_load8(L.box, 2 * side, pix, lo);
      level_load8(L.box, 2 * side + 1, pix, hi);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] = lo[j];
        v[8 + j] = hi[j];
      }
      dist[side] = dfl_expect(v);
    }
    bx = box_xywh((float)px + 0.5f, (float)py + 0.5f, dist[0], dist[1], dist[2], dist[3], L.stride);
  }
  float* yb = a.y + (long long)b * (4 + a.nc) * a.A + an;
  yb[0] = bx.x;
  yb[(long long)a.A] = bx.y;
  yb[2LL * a.A] = bx.z;
  yb[3LL * a.A] = bx.w;
  for (int c8 = 0; c8 * 8 < a.nc; ++c8) {
    float v[8];
    level_load8(L.cls, c8, pix, v);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (c8 * 8 + j < a.nc) yb[(long long)(4 + c8 * 8 + j) * a.A] = sigmoid_score(v[j]);
  }
}

int decode_launch(const DecodeLevel* lv, int n_levels, int batch, int nc, int A, float* y, hipStream_t s) {
  if (n_levels > MAX_LEVELS) return fail(-3, "decode: too many levels");
  DecodeArgs a;
  for (int i = 0; i < n_levels; ++i) a.lv[i] = lv[i];
  for (int i = n_levels; i < MAX_LEVELS; ++i) {
    a.lv[i] = lv[0];
    a.lv[i].a0 = 1 << 30;
  }
  a.n_levels = n_levels;
  a.batch = batch;
  a.nc = nc;
  a.A = A;
  a.y = y;
  hipLaunchKernelGGL(decode_kernel, dim3((unsigned)((A + 255) / 256), batch), dim3(256), 0, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

}  // namespace yl
