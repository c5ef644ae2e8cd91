// Internal declarations shared by the kernel and engine translation units of libyl_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/yl_amd.h"

namespace yl {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

void set_error(const std::string& s);
int fail(int code, const std::string& s);

#define YL_HIP(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return ::yl::fail(-2, std::string(#expr) + ": " + hipGetErrorString(_e));              \
  } while (0)

// ----------------------------------------------------------------------------- conv (MFMA)
// Implicit-GEMM view: rows = output channels (A operand, packed weights), columns = output
// pixels (B operand, LDS-staged NHWC halo tile), K = taps x input channels in groups of 8.
struct ConvArgs {
  const __bf16* in;
  void* out;
  const __bf16* res;   // nullptr: none
  const __bf16* wpk;   // [cb][chunk][kstep][MT][64 lanes][8]
  const float* bias;   // padded to ncb*MT*16
  long long in_img, out_img, res_img;  // elements per image in the underlying buffers
  int H, W, Ho, Wo;
  int in_pitch, in_coff, out_pitch, out_coff, res_pitch, res_coff;
  int Cin, Cout, stride, pad, act, out_f32;
  int TH, TW, tiles_x, tiles_y, ncb;   // output tile / grid decomposition
  int HH, HW;                           // halo dims
  int cch, cg, lg2, nchunks, ksteps, pp;  // chunk channels, groups of 8, log2(pow2>=cg), ...
  int ntaps;
  int halo_lds_off, halo_bytes;  // LDS: [filter slab][halo buffer(s)]
  int ntiles, gws;               // pixel tiles in total / blocks per output-channel block
  int npatches, ppw;             // direct kernel: wave patches in total / patches per wave
  int kloop, kkn;                // direct kernel: padded k-step count of the main loop, k-steps per tap
  unsigned magic_kkn;
  unsigned magic_hw, magic_tw, magic_cg, magic_wo, magic_ho;  // ceil(2^32 / x) for small-number division
  int npix;                      // direct kernel: batch * Ho * Wo
  unsigned in_bytes, out_bytes, res_bytes;  // buffer-descriptor sizes (bytes past the base pointers)
  int nblocks;
  int dbg;  // development ablation flags (YL_CONV_DBG), 0 in production
};

struct ConvConfig {
  int MT, NT, uniform, ws;
  int TH, TW, tiles_x, tiles_y, ncb;
  int HH, HW, cch, cg, lg2, nchunks, ksteps, pp, ntaps, halo_lds_off, halo_bytes, ntiles, gws, ppw, kloop;
  size_t lds_bytes;
  size_t wpk_elems;  // packed weight elements
  const char* name;
};

// Chooses the tiling for one conv; returns <0 when unsupported.
int conv_choose(int batch, int Cin, int Cout, int k, int stride, int Ho, int Wo, int aligned8, ConvConfig* cfg);
// Packs folded fp32 OIHW weights into the fragment order cfg describes (host).
void conv_pack_weights(const float* w_oihw, int Cout, int Cin_w, int Cin_pad, int k, const ConvConfig& cfg,
                       std::vector<uint16_t>* out);
int conv_launch(const ConvConfig& cfg, ConvArgs a, int batch, hipStream_t s);

// ------------------------------------------------------------------------ elementwise-ish
struct DwArgs {
  const __bf16* in;
  __bf16* out;
  const float* w;     // [9][C]
  const float* bias;  // [C]
  long long in_img, out_img;
  int H, W, C, in_pitch, in_coff, out_pitch, out_coff;
  int act, accumulate, grp, grp_stride, grp_off;
  int batch;
};
int dwconv3_launch(const DwArgs& a, hipStream_t s);

struct PoolArgs {
  const __bf16* in;  // slice 0 (cv1 output) -- the buffer also receives the three pools
  __bf16* out;       // same buffer base, slices 1..3 follow at coff + C, +2C, +3C
  long long img;
  int H, W, C, pitch, coff, batch;
};
int sppf_pool_launch(const PoolArgs& a, hipStream_t s);

struct UpArgs {
  const __bf16* in;
  __bf16* out;
  long long in_img, out_img;
  int H, W, C, in_pitch, in_coff, out_pitch, out_coff, factor, batch;  // H, W = input dims
};
int upsample_launch(const UpArgs& a, hipStream_t s);

struct AttnArgs {
  const __bf16* qkv;
  __bf16* out;
  long long in_img, out_img;
  int N, heads, in_pitch, in_coff, out_pitch, out_coff, batch;
  float scale;
};
int attention_launch(const AttnArgs& a, hipStream_t s);

struct StemArgs {
  const void* in;     // NCHW [B][3][H][W]
  __bf16* out;
  const float* w;     // [Cout][27] (bf16-rounded values)
  const float* bias;
  long long out_img;
  int B, H, W, Ho, Wo, Cout, out_pitch, out_coff, act, nseg;
};
int stem_launch(const StemArgs& a, int in_dtype, hipStream_t s);

int convert_input_launch(const void* in, int in_dtype, __bf16* out, int batch, int H, int W, int pitch,
                         long long out_img, hipStream_t s);

struct DecodeLevel {
  const float* raw;  // [B][H][W][64+nc]
  int H, W, a0;      // a0: first anchor index of this level
  float stride;
};
int decode_launch(const DecodeLevel* lv, int n_levels, int batch, int nc, int A, float* y, hipStream_t s);

// Decode arithmetic shared by the plain decode kernel and the fused decode+candidate kernel, so both
// produce bit-identical boxes and scores (head.py:95-126, block.py:66-69, tal.py:341-350).
__device__ __forceinline__ float dfl_expect(const float* v) {  // softmax over 16 bins, expectation
  float mx = v[0];
#pragma unroll
  for (int j = 1; j < 16; ++j) mx = fmaxf(mx, v[j]);
  float sum = 0.f, ws = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float e = expf(v[j] - mx);
    sum += e;
    ws += e * (float)j;
  }
  return ws / sum;
}
__device__ __forceinline__ float4 box_xywh(float ax, float ay, float dl, float dt, float dr, float db, float stride) {
  const float x1 = ax - dl, y1 = ay - dt, x2 = ax + dr, y2 = ay + db;
  return make_float4(((x1 + x2) / 2.0f) * stride, ((y1 + y2) / 2.0f) * stride, (x2 - x1) * stride,
                     (y2 - y1) * stride);
}
__device__ __forceinline__ float sigmoid_score(float logit) { return 1.0f / (1.0f + expf(-logit)); }

}  // namespace yl
