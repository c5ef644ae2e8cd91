// Bandwidth-bound NHWC bf16 kernels of the YOLO11 path: input conversion, depthwise 3x3, SPPF
// max-pool chain, nearest upsample / slice copy, C2PSA attention and the Detect decode.
// Every thread moves 16-byte (8 x bf16) channel groups so that global accesses are coalesced.
#include "yl_internal.h"

namespace yl {

__device__ __forceinline__ float silu_g(float v) { return v / (1.0f + __expf(-v)); }

// ------------------------------------------------------------------ input NCHW -> NHWC bf16 (C=8)
// engine/predictor.py:67-85 (preprocess: .float(), /255 for uint8 sources); channels 3..7 are zero.
template <typename T>
__global__ void convert_input_kernel(const T* __restrict__ in, __bf16* __restrict__ out, int HW, long long out_img,
                                     int pitch, float mul, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int b = (int)(i / HW);
  const int p = (int)(i - (long long)b * HW);
  const T* src = in + (long long)b * 3 * HW + p;
  bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
  // `div` is 255 for uint8 sources: a true fp32 division like `im /= 255` (predictor.py:84)
  v[0] = (__bf16)((float)src[0] / mul);
  v[1] = (__bf16)((float)src[HW] / mul);
  v[2] = (__bf16)((float)src[2 * (long long)HW] / mul);
  *reinterpret_cast<bf16x8*>(out + (long long)b * out_img + (long long)p * pitch) = v;
}

int convert_input_launch(const void* in, int in_dtype, __bf16* out, int batch, int H, int W, int pitch,
                         long long out_img, hipStream_t s) {
  const long long total = (long long)batch * H * W;
  const int blocks = (int)((total + 255) / 256);
  if (pitch != 8) return fail(-3, "convert_input: input view pitch must be 8");
  if (in_dtype == YL_F32)
    hipLaunchKernelGGL(convert_input_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)in, out, H * W,
                       out_img, pitch, 1.0f, total);
  else if (in_dtype == YL_F16)
    hipLaunchKernelGGL(convert_input_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, (const _Float16*)in, out,
                       H * W, out_img, pitch, 1.0f, total);
  else if (in_dtype == YL_U8)
    hipLaunchKernelGGL(convert_input_kernel<uint8_t>, dim3(blocks), dim3(256), 0, s, (const uint8_t*)in, out, H * W,
                       out_img, pitch, 255.0f, total);
  else
    return fail(-3, "convert_input: unsupported input dtype");
  YL_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------ depthwise 3x3
// nn/modules/conv.py:100-105 (DWConv, groups = C) with folded BN; optional SiLU; optional
// accumulate (x = attn_out + pe(v), nn/modules/block.py:913) and the per-head v channel map.
__global__ void dwconv3_kernel(DwArgs a) {
  const int cg = a.C >> 3;
  const long long total = (long long)a.batch * a.H * a.W * cg;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int g8 = (int)(i % cg);
  long long p = i / cg;
  const int x = (int)(p % a.W);
  p /= a.W;
  const int y = (int)(p % a.H);
  const int b = (int)(p / a.H);
  const int c = g8 * 8;
  const int cin = a.grp ? (c / a.grp) * a.grp_stride + a.grp_off + (c % a.grp) : c;
  const __bf16* inb = a.in + (long long)b * a.in_img + a.in_coff + cin;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = a.bias[c + j];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int iy = y + kh - 1;
    if ((unsigned)iy >= (unsigned)a.H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int ix = x + kw - 1;
      if ((unsigned)ix >= (unsigned)a.W) continue;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(inb + (long long)(iy * a.W + ix) * a.in_pitch);
      const float* wp = a.w + (kh * 3 + kw) * a.C + c;
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(wp + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] += (float)v[j] * w0[j];
        acc[4 + j] += (float)v[4 + j] * w1[j];
      }
    }
  }
  __bf16* op = a.out + (long long)b * a.out_img + (long long)(y * a.W + x) * a.out_pitch + a.out_coff + c;
  bf16x8 o;
  if (a.accumulate) {
    const bf16x8 prev = *reinterpret_cast<const bf16x8*>(op);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (__bf16)((a.act ? silu_g(acc[j]) : acc[j]) + (float)prev[j]);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (__bf16)(a.act ? silu_g(acc[j]) : acc[j]);
  }
  *reinterpret_cast<bf16x8*>(op) = o;
}

int dwconv3_launch(const DwArgs& a, hipStream_t s) {
  if (a.C % 8 || a.in_pitch % 8 || a.in_coff % 8 || a.out_pitch % 8 || a.out_coff % 8 ||
      (a.grp && (a.grp % 8 || a.grp_stride % 8 || a.grp_off % 8)))
    return fail(-3, "dwconv3: channels/pitches must be multiples of 8");
  const long long total = (long long)a.batch * a.H * a.W * (a.C / 8);
  hipLaunchKernelGGL(dwconv3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------- SPPF pooling chain
// nn/modules/block.py:180-184: y1 = m(x), y2 = m(y1), y3 = m(y2) with m = MaxPool2d(5,1,2).
// Chained 5x5 max pools with -inf padding equal single 5/9/13 windows clipped to the image.
__global__ void sppf_pool_kernel(PoolArgs a) {
  const int cg = a.C >> 3;
  const long long total = (long long)a.batch * a.H * a.W * cg;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int g8 = (int)(i % cg);
  long long p = i / cg;
  const int x = (int)(p % a.W);
  p /= a.W;
  const int y = (int)(p % a.H);
  const int b = (int)(p / a.H);
  const __bf16* inb = a.in + (long long)b * a.img + a.coff + g8 * 8;
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
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(inb + (long long)(iy * a.W + ix) * a.pitch);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)v[j];
        m13[j] = fmaxf(m13[j], f);
        if (r <= 4) m9[j] = fmaxf(m9[j], f);
        if (r <= 2) m5[j] = fmaxf(m5[j], f);
      }
    }
  }
  __bf16* op = a.out + (long long)b * a.img + (long long)(y * a.W + x) * a.pitch + a.coff + g8 * 8;
  bf16x8 o5, o9, o13;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    o5[j] = (__bf16)m5[j];
    o9[j] = (__bf16)m9[j];
    o13[j] = (__bf16)m13[j];
  }
  *reinterpret_cast<bf16x8*>(op + a.C) = o5;
  *reinterpret_cast<bf16x8*>(op + 2 * a.C) = o9;
  *reinterpret_cast<bf16x8*>(op + 3 * a.C) = o13;
}

int sppf_pool_launch(const PoolArgs& a, hipStream_t s) {
  if (a.C % 8 || a.pitch % 8 || a.coff % 8) return fail(-3, "sppf_pool: channels must be multiples of 8");
  const long long total = (long long)a.batch * a.H * a.W * (a.C / 8);
  hipLaunchKernelGGL(sppf_pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

// -------------------------------------------------------------- nearest upsample / slice copy
// nn.Upsample(None, 2, "nearest") of yolo11.yaml:31,35 written straight into the Concat buffer.
__global__ void upsample_kernel(UpArgs a) {
  const int cg = a.C >> 3;
  const int Ho = a.H * a.factor, Wo = a.W * a.factor;
  const long long total = (long long)a.batch * Ho * Wo * cg;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int g8 = (int)(i % cg);
  long long p = i / cg;
  const int x = (int)(p % Wo);
  p /= Wo;
  const int y = (int)(p % Ho);
  const int b = (int)(p / Ho);
  const int iy = y / a.factor, ix = x / a.factor;
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(a.in + (long long)b * a.in_img +
                                                    (long long)(iy * a.W + ix) * a.in_pitch + a.in_coff + g8 * 8);
  *reinterpret_cast<bf16x8*>(a.out + (long long)b * a.out_img + (long long)(y * Wo + x) * a.out_pitch + a.out_coff +
                             g8 * 8) = v;
}

int upsample_launch(const UpArgs& a, hipStream_t s) {
  if (a.C % 8 || a.in_pitch % 8 || a.in_coff % 8 || a.out_pitch % 8 || a.out_coff % 8)
    return fail(-3, "upsample: channels must be multiples of 8");
  const long long total = (long long)a.batch * a.H * a.factor * a.W * a.factor * (a.C / 8);
  hipLaunchKernelGGL(upsample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------------- attention
// nn/modules/block.py:905-912: per head, attn = softmax((q^T k) * scale) over keys;
// out[:, i] = sum_j v[:, j] * attn[i, j].  qkv channel layout per head: [q(32) | k(32) | v(64)].
// One workgroup = 64 queries of one (image, head); 4 waves split the keys (online softmax each)
// and are merged through LDS.  K/V rows are read as LDS broadcasts.  fp32 VALU (about 1 % of the
// network's MACs); the MFMA version is future work.
constexpr int AT_KD = 32, AT_HD = 64, AT_KT = 256;  // key tile staged per pass

__global__ __launch_bounds__(256) void attention_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) __bf16 sk[AT_KT * AT_KD];
  __shared__ __attribute__((aligned(16))) __bf16 sv[AT_KT * AT_HD];
  __shared__ float sm[4][64], sl[4][64];
  const int tid = threadIdx.x, lane = tid & 63, part = tid >> 6;
  const int qblocks = (a.N + 63) / 64;
  int bid = blockIdx.x;
  const int qb = bid % qblocks;
  bid /= qblocks;
  const int head = bid % a.heads;
  const int b = bid / a.heads;
  const int hstride = 2 * AT_KD + AT_HD;
  const __bf16* base = a.qkv + (long long)b * a.in_img + a.in_coff + head * hstride;
  const int q = qb * 64 + lane;
  const bool qv = q < a.N;
  float qr[AT_KD];
  {
    const __bf16* qp = base + (long long)(qv ? q : 0) * a.in_pitch;
#pragma unroll
    for (int c = 0; c < AT_KD; c += 8) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(qp + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) qr[c + j] = (float)v[j] * a.scale;
    }
  }
  float m = -__builtin_huge_valf(), l = 0.f;
  float acc[AT_HD];
#pragma unroll
  for (int d = 0; d < AT_HD; ++d) acc[d] = 0.f;

  for (int k0 = 0; k0 < a.N; k0 += AT_KT) {
    const int kn = min(AT_KT, a.N - k0);
    __syncthreads();
    // stage K [kn][32] and V [kn][64]
    for (int i = tid; i < kn * (AT_KD / 8); i += 256) {
      const int key = i / (AT_KD / 8), c8 = i % (AT_KD / 8);
      *reinterpret_cast<bf16x8*>(sk + key * AT_KD + c8 * 8) =
          *reinterpret_cast<const bf16x8*>(base + (long long)(k0 + key) * a.in_pitch + AT_KD + c8 * 8);
    }
    for (int i = tid; i < kn * (AT_HD / 8); i += 256) {
      const int key = i / (AT_HD / 8), c8 = i % (AT_HD / 8);
      *reinterpret_cast<bf16x8*>(sv + key * AT_HD + c8 * 8) =
          *reinterpret_cast<const bf16x8*>(base + (long long)(k0 + key) * a.in_pitch + 2 * AT_KD + c8 * 8);
    }
    __syncthreads();
    for (int key = part; key < kn; key += 4) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < AT_KD; c += 8) {
        const bf16x8 kv = *reinterpret_cast<const bf16x8*>(sk + key * AT_KD + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += qr[c + j] * (float)kv[j];
      }
      const float mn = fmaxf(m, s);
      const float alpha = __expf(m - mn);
      const float pexp = __expf(s - mn);
      l = l * alpha + pexp;
      m = mn;
#pragma unroll
      for (int d = 0; d < AT_HD; d += 8) {
        const bf16x8 vv = *reinterpret_cast<const bf16x8*>(sv + key * AT_HD + d);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[d + j] = acc[d + j] * alpha + pexp * (float)vv[j];
      }
    }
  }
  // merge the 4 key partitions
  sm[part][lane] = m;
  __syncthreads();
  const float mg = fmaxf(fmaxf(sm[0][lane], sm[1][lane]), fmaxf(sm[2][lane], sm[3][lane]));
  const float f = (m == -__builtin_huge_valf()) ? 0.f : __expf(m - mg);
  sl[part][lane] = l * f;
  __syncthreads();
  const float lg = sl[0][lane] + sl[1][lane] + sl[2][lane] + sl[3][lane];
  // reuse sv as fp32 scratch [4][64 q][64 d] = 64 KB? too large: reduce in 4 rounds through sk/sv
  float* red = reinterpret_cast<float*>(sv);  // 256*64*2 B = 32 KB = 8192 floats = [2][64][64]
  __syncthreads();
  for (int round = 0; round < 2; ++round) {
    // parts (2*round, 2*round+1) deposit, everybody accumulates afterwards
    if ((part >> 1) == round) {
      float* dst = red + ((part & 1) * 64 + lane) * AT_HD;
#pragma unroll
      for (int d = 0; d < AT_HD; ++d) dst[(d + lane) & (AT_HD - 1)] = acc[d] * f;  // skewed: conflict-free
    }
    __syncthreads();
    if (part == 0) {
      if (round == 0) {
#pragma unroll
        for (int d = 0; d < AT_HD; ++d) acc[d] = 0.f;
      }
      const float* s0 = red + lane * AT_HD;
      const float* s1 = red + (64 + lane) * AT_HD;
#pragma unroll
      for (int d = 0; d < AT_HD; ++d) acc[d] += s0[(d + lane) & (AT_HD - 1)] + s1[(d + lane) & (AT_HD - 1)];
    }
    __syncthreads();
  }
  if (part == 0 && qv) {
    const float inv = 1.0f / lg;
    __bf16* op = a.out + (long long)b * a.out_img + (long long)q * a.out_pitch + a.out_coff + head * AT_HD;
#pragma unroll
    for (int d = 0; d < AT_HD; d += 8) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (__bf16)(acc[d + j] * inv);
      *reinterpret_cast<bf16x8*>(op + d) = o;
    }
  }
}

int attention_launch(const AttnArgs& a, hipStream_t s) {
  if (a.in_pitch % 8 || a.in_coff % 8 || a.out_pitch % 8 || a.out_coff % 8)
    return fail(-3, "attention: channel offsets must be multiples of 8");
  const int qblocks = (a.N + 63) / 64;
  hipLaunchKernelGGL(attention_kernel, dim3(a.batch * a.heads * qblocks), dim3(256), 0, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------- Detect decode
// nn/modules/head.py:95-126 (_inference): DFL (block.py:66-69: softmax over 16 bins, expectation),
// dist2bbox xywh (utils/tal.py:341-350) on anchors (x+0.5, y+0.5) (tal.py:326-338), times stride,
// and sigmoid of the class logits.  fp32 in, fp32 out: y[B][4+nc][A].
constexpr int MAX_LEVELS = 8;
struct DecodeArgs {
  DecodeLevel lv[MAX_LEVELS];
  int n_levels, batch, nc, A;
  float* y;
};

__global__ void decode_kernel(DecodeArgs a) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)a.batch * a.A) return;
  const int b = (int)(i / a.A);
  const int an = (int)(i - (long long)b * a.A);
  int l = 0;
#pragma unroll
  for (int k = 1; k < MAX_LEVELS; ++k)
    if (k < a.n_levels && an >= a.lv[k].a0) l = k;
  const DecodeLevel L = a.lv[l];
  const int p = an - L.a0;
  const int no = 64 + a.nc;
  const float* raw = L.raw + ((long long)b * L.H * L.W + p) * no;
  float dist[4];
#pragma unroll
  for (int side = 0; side < 4; ++side) {
    float v[16];
    float mx = -__builtin_huge_valf();
    if ((no & 3) == 0) {
#pragma unroll
      for (int k = 0; k < 16; k += 4) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(raw + side * 16 + k);
        v[k] = t[0]; v[k + 1] = t[1]; v[k + 2] = t[2]; v[k + 3] = t[3];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = raw[side * 16 + k];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) mx = fmaxf(mx, v[k]);
    float sum = 0.f, ws = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float e = expf(v[k] - mx);
      sum += e;
      ws += e * (float)k;
    }
    dist[side] = ws / sum;
  }
  const float ax = (float)(p % L.W) + 0.5f, ay = (float)(p / L.W) + 0.5f;
  const float x1 = ax - dist[0], y1 = ay - dist[1], x2 = ax + dist[2], y2 = ay + dist[3];
  float* yb = a.y + (long long)b * (4 + a.nc) * a.A + an;
  yb[0] = ((x1 + x2) / 2.0f) * L.stride;
  yb[(long long)a.A] = ((y1 + y2) / 2.0f) * L.stride;
  yb[2LL * a.A] = (x2 - x1) * L.stride;
  yb[3LL * a.A] = (y2 - y1) * L.stride;
  for (int c = 0; c < a.nc; ++c) yb[(long long)(4 + c) * a.A] = 1.0f / (1.0f + expf(-raw[64 + c]));
}

int decode_launch(const DecodeLevel* lv, int n_levels, int batch, int nc, int A, float* y, hipStream_t s) {
  if (n_levels > MAX_LEVELS) return fail(-3, "decode: too many levels");
  DecodeArgs a;
  for (int i = 0; i < n_levels; ++i) a.lv[i] = lv[i];
  for (int i = n_levels; i < MAX_LEVELS; ++i) a.lv[i] = DecodeLevel{nullptr, 0, 0, 1 << 30, 0.f};
  a.n_levels = n_levels;
  a.batch = batch;
  a.nc = nc;
  a.A = A;
  a.y = y;
  const long long total = (long long)batch * A;
  hipLaunchKernelGGL(decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

}  // namespace yl
