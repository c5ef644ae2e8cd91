// Internal declarations shared by the kernel and engine translation units of libyl_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/yl_amd.h"
#include "../../include/yl_amd_diag.h"

namespace yl {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// Switches.  env_int: a RUNTIME knob -- every one is listed in INTEGRATION.md (section 5); tests use most of them to force a
// kernel choice on small shapes or to build the unfused twin of a fused launch.  tune_int: a DEVELOPMENT constant (tile shapes,
// thresholds, ablation flags whose A / B is recorded in NOTEBOOK.md): the production library returns the default and never looks
// at the environment; `make EXTRA=-DYL_TUNING` builds a library that does.
inline int env_int(const char* name, int def) {
  const char* v = getenv(name);
  return v ? atoi(v) : def;
}
#ifdef YL_TUNING
inline int tune_int(const char* name, int def) { return env_int(name, def); }
#else
inline int tune_int(const char*, int def) { return def; }
#endif

void set_error(const std::string& s);
int fail(int code, const std::string& s);

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute of a kernel: set once per (device, kernel), so a process
// that drives a second GPU launches there with the attribute in place too (engine.hip).
int allow_big_lds(const void* kernel, int bytes);
// compute units of the current device (cached per device); the persistent-grid sizes derive from it
int device_cu_count();

// In-kernel phase stamps of a DIAGNOSTIC build (make EXTRA=-DYL_STAMPS; include/yl_amd_diag.h yl_diag_set_stamps): lane 0 of
// every wave stores {shader clock, 100 MHz clock} at phase boundaries into a buffer no other code reads.  The production
// build compiles the macro away.
// Non-temporal epilogue stores (round 5): YL_ST_NT=1 (default) always, 0 never, -1 when the stored tensor has at least YL_ST_NT_BYTES
// bytes (default 48 MB).  Measured on conv_gemm_kernel: 384 -> 384 @320x320 x 8 365 -> 353 us, 128 -> 128 @80x80 x 64 52.1 -> 49.4 us
// stand-alone; on the conv / GEMM / e4m3 epilogues end to end (one box, two rounds): YOLO11s batch 64 25.7 -> 26.1 k, YOLO11x 1280 fp8
// 1 092 -> 1 112, YOLO11n batch 32 56.0 -> 57.5 k (one batch at a time 32.3 -> 32.9 k) -- with twenty batches in flight no cache holds a
// layer's output until its reader runs, and a store that does not allocate leaves the caches to the operands that are re-read.
int st_nt_policy(long long out_bytes);
extern unsigned long long* g_stamp_buf;
#ifdef YL_STAMPS
#define YL_STAMP(buf, waves, wave, lane, id)                                                             \
  do {                                                                                                   \
    if ((buf) && (lane) == 0) {                                                                          \
      unsigned long long* _p = (buf) + (((size_t)blockIdx.x * (waves) + (wave)) * 32 + (id)) * 2;        \
      _p[0] = __builtin_amdgcn_s_memtime();                                                              \
      _p[1] = __builtin_amdgcn_s_memrealtime();                                                          \
    }                                                                                                    \
  } while (0)
#else
#define YL_STAMP(buf, waves, wave, lane, id) do { } while (0)
#endif

#define YL_HIP(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return ::yl::fail(-2, std::string(#expr) + ": " + hipGetErrorString(_e));              \
  } while (0)

// ------------------------------------------------------------------- activation layout in HBM
// bf16 activations are "C8-planar, guarded":  buffer[C/8][plane_px][8]
//   * one PLANE per group of 8 channels; a pixel of a plane is 16 bytes (one MFMA k-group), so the 16
//     pixels x 4 k-groups a wave fetches as a B fragment are 4 runs of 256 contiguous bytes -- the
//     texture addresser serves that at its full 64 B/clk, whereas NHWC (pixels C*2 bytes apart) costs
//     one clock per lane, 4x slower (measured: tools/micro/ta_rate.hip);
//   * inside a plane the whole batch is one array of rows, each row followed by one zero GUARD pixel,
//     each image followed by one zero guard row (plus one guard row in front):
//         pixel(b, y, x) = g0 + b * img + y * W1 + x,   W1 = W + 1, img = (H + 1) * W1, g0 = W1 + 1
//     so every 3x3 tap of every pixel is a plain offset (dy * W1 + dx) that lands on real data or on a
//     zero guard: no per-tap predicates, no halo staging.  Guards are zeroed when a workspace is first
//     used and never written afterwards (buffers are not recycled between tensors).
// A channel-slice view (torch.chunk / torch.cat / Concat) is a sub-range of planes.
// The Detect level maps are ordinary planar buffers (yl_level); fp32 NHWC conv output still exists as
// an epilogue option (yl_view dtype YL_F32) but the lowering no longer uses it.
struct Geo {
  int W1, img, g0;
  unsigned plane_px;  // pixels per plane (multiple of 16)
};
inline Geo geo_of(int batch, int H, int W) {
  Geo g;
  g.W1 = W + 1;
  g.img = (H + 1) * g.W1;
  g.g0 = g.W1 + 1;
  const long long n = (long long)g.g0 + (long long)batch * g.img + 1;
  g.plane_px = (unsigned)((n + 15) & ~15LL);
  return g;
}
inline long long plane_bytes(const Geo& g) { return (long long)g.plane_px * 16; }

// XCD-aware, bijective block remap: consecutive work items (blocks that share input rows / filters)
// land on one XCD and therefore one L2; hardware deals workgroups to the 8 XCDs round-robin.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ----------------------------------------------------------------------------- conv (MFMA)
// Implicit-GEMM view: rows = output channels (A operand, packed weights staged in LDS), columns =
// output pixels (B operand, fetched straight from the planes), K = taps x input channels in groups of 8.
struct ConvArgs {
  const __bf16* in;    // first plane of the input slice
  void* out;           // bf16: first plane of the output slice; fp32: NHWC buffer base
  const __bf16* res;   // first plane of the residual slice, nullptr: none
  const __bf16* in2;   // 1x1 only: low-resolution source of the first ks_split*32 input channels (nearest x2), or nullptr
  unsigned in2_plane, in2_bytes;
  int in2_w1, in2_img, in2_g0, ks_split;
  const __bf16* wpk;   // [cb][kstep][MT][64 lanes][8]
  const __bf16* wpk2;  // fused follow-up 1x1: [MT/2 k-steps][MT][64 lanes][8], nullptr: none
  const float* bias2;  // its bias, padded to MT*16
  int post_cout, post_act;
  int ksteps2, cg2;    // dwpw_kernel with a further 1x1 (wpk2): its k-steps and input planes
  const void* dw_wpk;  // dwpw_kernel: depthwise taps as block-diagonal MFMA A fragments (dw_pack_mfma), bias [C], SiLU flag
  // dwpw_kernel with a fused tail that writes a Detect level's class logits: the NMS candidate pre-filter in its epilogue
  // (PrefilterSink; pf_count == nullptr: off).  pf_anchor0 = index of the level's first anchor.
  int32_t* pf_count;
  int32_t* pf_list;
  int pf_sub_cap, pf_anchor0;
  float pf_logit_lo;
  const float* dw_b;
  int dw_act;
  const float* bias;   // padded to ncb*MT*16
  unsigned in_plane, out_plane, res_plane;  // bytes per plane
  int Ho, Wo;
  int in_w1, in_img, in_g0, out_w1, out_img, out_g0;  // guarded geometry (pixels); res shares the output's
  int out_pitch, out_coff;                  // fp32 NHWC output only
  int Cin, Cout, stride, act, out_f32;
  int ncb, cg, ksteps, kchunk, ntaps;       // k-steps in total / per LDS slab
  int slab2;                                // 1: a chunked filter (kchunk < ksteps) is restaged asynchronously through TWO chunk buffers (conv_body)
  int kord;                                 // 1: plane-major K order (3x3 only; per-lane tap / plane, the non-UNIFORM fetch)
  int npatches, ppw;                        // wave patches in total / patches per wave
  int kkn;                                  // k-steps per tap (uniform case)
  unsigned magic_kkn, magic_cg, magic_wo, magic_ho;  // ceil(2^32 / x) for small-number division
  int npix;                                 // batch * Ho * Wo
  unsigned in_bytes, out_bytes, res_bytes;  // buffer-descriptor sizes (bytes past the base pointers)
  int nblocks;
  int dbg;  // development ablation flags (YL_CONV_DBG), 0 in production
  int st_nt;  // 1: the epilogue's stores carry the non-temporal hint (large outputs that no cache holds until their reader runs)
  // The conv that produces a Detect level's 64 DFL logits (yl_op.dfl_level): dfl_out = the level's decoded boxes, fp32
  // [npix][4] = (cx, cy, w, h) * stride, written by dfl_epilogue (conv_common.h) from the fp32 accumulators; the bf16 logits go
  // to `out` only with dfl_keep.  nullptr: an ordinary conv.
  float* dfl_out;
  float dfl_stride;
  int dfl_keep;
  unsigned long long* stamps;  // diagnostic builds (-DYL_STAMPS): dwpw_body's phase stamps
  // the LDS-tile kernels (conv3_reg / conv3_cs / conv3_f8t / dwpw tile form): tile geometry in guarded-pixel units
  int t_px, t_PL, t_ppp, t_ntiles, t_plane_px, t_H, t_W, t_B;  // pixels staged / LDS plane pitch / DMA pieces per plane
  unsigned magic_ppp, magic_img, magic_w1;
  // conv_f8_kernel (conv_f8.hip): in / wpk are e4m3 (planes of 16 channels; cg, kkn count groups of 16 / steps of 128)
  const float* alpha;          // s_in * s_w[c] per output channel, padded like bias
  float out_inv_scale;         // fp8 output: 1 / s_out
  float res_scale;             // fp8 residual: its scale
  int out_f8, res_f8;          // output / residual are fp8 planes (else bf16 planes)
  // second, e4m3 copy of output channels [out8_c0, out8_c0 + out8_n) (conv_kernel without follow-up conv, conv_f8_kernel)
  void* out8;
  unsigned out8_plane, out8_bytes;
  float out8_inv_scale;
  int out8_c0, out8_n;
  int out8_only;               // the e4m3 copy is the ONLY output (every reader is an fp8 conv): the launchers give the bf16 descriptor zero bytes, its stores vanish
  int ab_off;                  // conv_f8_kernel: byte offset in LDS of the channel block's alpha / bias (behind the filter slab)
};

struct ConvConfig {
  int MT, NT, uniform, post, ncb, cg, ksteps, kchunk, ntaps, npatches, gws, ppw;
  int f8t;           // 1: conv3_f8t_kernel (e4m3 48 -> 48 3x3: input tile in LDS, filter in registers) instead of conv_f8_kernel
  int reg;           // > 0: conv3_reg_kernel (filter in registers) with this many 16-pixel column tiles per wave
  int cs;            // 1: conv3_cs_kernel (64 -> 64 3x3: filter in registers, output channels split between wave pairs); 2: ... with the
                     //    fused 1x1 + box decode of Detect's box branch behind it (conv3_cs_choose_post)
  int slab2;         // 1: kchunk < ksteps and the chunks alternate between two LDS buffers, filled by LDS-DMA (conv_kernel, round 5)
  int kord;          // 1: K runs plane-major (k-group = plane * 9 + tap) instead of tap-major (tap * planes + plane), see conv_choose
  int gemm;          // > 0: conv_gemm_kernel (wide 1x1: both operands staged through LDS) with this many 16-channel tiles per wave
  int gemm_nwn;      // ... and this many waves along the pixels (2: 128-pixel tiles, 4: 256-pixel tiles)
  size_t lds_bytes;
  size_t wpk_elems;  // packed weight elements
  const char* name;
};

// Chooses the tiling for one conv; returns <0 when unsupported.
int conv_choose(int batch, int Cin, int Cout, int k, int stride, int Ho, int Wo, int post, ConvConfig* cfg);
int conv_choose_even(int batch, int Cin, int Cout, int k, int stride, int Ho, int Wo, ConvConfig* cfg);
// Packs folded fp32 OIHW weights into the fragment order cfg describes (host).
void conv_pack_weights(const float* w_oihw, int Cout, int Cin_w, int k, const ConvConfig& cfg,
                       std::vector<uint16_t>* out);
int conv_launch(const ConvConfig& cfg, ConvArgs a, int batch, hipStream_t s);
// conv3_reg.hip: 3x3 stride-1 convs with Cin, Cout in {32, 64}: the filter lives in the registers of every wave
void conv3_reg_choose(int batch, int Cin, int Cout, int k, int stride, int Ho, int Wo, int post, ConvConfig* cfg);
int conv3_reg_launch(const ConvConfig& cfg, ConvArgs a, int batch, hipStream_t s);
// conv3_cs.hip: 3x3 stride-1 64 -> 64 convs, output channels split between wave pairs, two workgroups per CU
void conv3_cs_choose(int batch, int Cin, int Cout, int k, int stride, int Ho, int Wo, int post, ConvConfig* cfg);
int conv3_cs_launch(const ConvConfig& cfg, ConvArgs a, int batch, hipStream_t s);
// ... with Detect's cv2[i][2] (1x1) and the box decode behind it in the same launch (sets cfg->cs = 2)
bool conv3_cs_choose_post(int batch, int Cin, int Cout1, int Cout2, int k, int stride, int Ho, int Wo, ConvConfig* cfg);
// ... of all Detect levels in one launch (conv3_csg_kernel): whether the members fit / the launch (args as conv_group_launch's)
bool conv3_cs_group_fits(const ConvConfig* cfg, int n, int batch, const int* Ho, const int* Wo);
int conv3_cs_group_launch(const ConvArgs* args, int n, int batch, hipStream_t s);
// conv_gemm.hip: wide 1x1 convs (Cin, Cout >= 128) as a 128/192 x 128 macro-tile GEMM, both operands through LDS by LDS-DMA
void conv_gemm_choose(int batch, int Cin, int Cout, int k, int stride, int Ho, int Wo, ConvConfig* cfg, int cin2 = 0);
int conv_gemm_launch(const ConvConfig& cfg, ConvArgs a, int batch, hipStream_t s);
int dwpw_launch(const ConvConfig& cfg, ConvArgs a, int batch, hipStream_t s);
// group launches (the three Detect levels in one launch, conv_mfma.hip)
void conv_retile_for_group(ConvConfig* c, int NT, int batch, int Ho, int Wo);
bool conv_group_compatible(const ConvConfig* cfg, const ConvArgs* a, int n);
int conv_group_launch(const ConvConfig* cfg, const ConvArgs* args, int n, int batch, hipStream_t s);
int dwpw_group_launch(const ConvConfig* cfg, const ConvArgs* args, int n, int batch, hipStream_t s);
int head_group_launch(const ConvConfig* ccfg, const ConvArgs* cargs, int nc, const ConvConfig* dcfg, const ConvArgs* dargs, int nd,
                      int batch, hipStream_t s);
void dw_pack_mfma(const float* w, int C, std::vector<uint16_t>* out);
int dwpw_tile_launch(const ConvConfig& cfg, ConvArgs a, int batch, hipStream_t s, int* handled);
// conv_f8.hip: e4m3 x e4m3 -> fp32 on v_mfma_f32_16x16x128_f8f6f4
int conv_f8_choose(int batch, int Cin, int Cout, int k, int stride, int Ho, int Wo, ConvConfig* cfg);
void conv_f8_pack_weights(const float* w_oihw, int Cout, int Cin, int k, const ConvConfig& cfg, std::vector<uint8_t>* out,
                          std::vector<float>* wscale);
int conv_f8_launch(const ConvConfig& cfg, ConvArgs a, int batch, hipStream_t s);
// conv3_f8t.hip: the LDS-tile form of the 48 -> 48 e4m3 3x3 conv (same packed filters, bit-identical to conv_f8_kernel)
bool conv3_f8t_fits(const ConvConfig& cfg, int batch, int Cin, int Cout, int k, int stride, int Ho, int Wo);
int conv3_f8t_launch(const ConvConfig& cfg, ConvArgs a, int batch, hipStream_t s);
uint8_t f32_to_e4m3(float f);

// ------------------------------------------------------------------------ elementwise-ish
// PView: a channel slice of a planar buffer as the kernels see it.
struct PView {
  __bf16* base;     // first plane of the slice
  long long plane;  // elements per plane
  int W1, img, g0;
};

// bf16 planes (8 channels) -> e4m3 planes (16 channels), x / scale, whole planes incl. guards
int quant_f8_launch(const PView& in, void* out, unsigned plane_px, int C, float scale, hipStream_t s);

struct DwArgs {
  PView in, out;
  const float* w;     // [9][C]
  const float* bias;  // [C]
  int H, W, C;
  int act, accumulate, grp, grp_stride, grp_off;
  int batch;
};
int dwconv3_launch(const DwArgs& a, hipStream_t s);

struct PoolArgs {
  PView v;  // slice 0 (cv1 output); the three pools are written to the C/8 planes after each other
  int H, W, C, batch;
};
int sppf_pool_launch(const PoolArgs& a, hipStream_t s);

struct UpArgs {
  PView in, out;
  int H, W, C, factor, batch;  // H, W = input dims
};
int upsample_launch(const UpArgs& a, hipStream_t s);

struct AttnArgs {
  PView in, out;
  int H, W, N, heads, batch;
  float scale;
};
int attention_launch(const AttnArgs& a, hipStream_t s);

struct StemArgs {
  const void* in;     // NCHW [B][3][H][W]
  PView out;
  const float* w;     // [Cout][27] (bf16-rounded values)
  const __bf16* wpk;  // the same in MFMA A-fragment order [Cout / 16][64 lanes][8] (taps 27..31 zero); nullptr: gather from w
  const float* bias;
  int B, H, W, Ho, Wo, Cout, act, nseg;
  unsigned char* out8;   // e4m3 copy of the output (planes of 16 channels, same pixel geometry), nullptr: none
  long long out8_plane;  // bytes per plane
  float out8_inv_scale;
  int out8_only;         // store only the copy
};
int stem_launch(const StemArgs& a, int in_dtype, hipStream_t s);

int convert_input_launch(const void* in, int in_dtype, const PView& out, int batch, int H, int W, hipStream_t s);

// stem2.hip: layers 0 and 1 in one launch (the stem's output lives only in LDS)
struct Stem2Args {
  const void* in;        // NCHW [B][3][H][W]
  int B, H, W;
  int H0, W0, C0;        // stem output (never materialised)
  int H1, W1;            // output of the second conv
  const float* w0;       // stem filters [C0][27] (bf16-rounded values), bias [C0]
  const float* b0;
  int act0;
  const __bf16* wpk;     // second conv: conv_pack_weights order [kstep][MT][64][8]; bias padded to MT*16
  const float* bias;
  int ksteps, Cout, act;
  const __bf16* wpk2;    // fused follow-up 1x1 (nullptr: none), as ConvArgs
  const float* bias2;
  int post_cout, post_act;
  PView out;
  unsigned out_bytes;
  int st_nt;             // non-temporal stores (st_nt_policy), filled by stem2_launch
  int cacheable;         // 1: never (yl_op.reserved3 bit 8: the latency profile's reader runs next)
  int tiles_x, tiles_y;
  int32_t* zero_ptr;           // zero_n words the first workgroup clears (the NMS candidate counters of this step:
  int zero_n;                  // PrefilterSink::zero_ptr -- one launch less in front of the network); 0: none
  unsigned long long* stamps;  // diagnostic builds (-DYL_STAMPS)
};
int stem2_launch(const Stem2Args& a, int MT, int in_dtype, hipStream_t s);

// c3k2s.hip: the body of a narrow C3k2 block (Bottleneck 16 -> 8 -> 16 + residual, then cv2 48 -> 64) in one launch
struct C3k2sArgs {
  PView tin;             // first plane of the block's concat buffer: planes [a0 a1 | b0 b1 | (c_out, not used)]
  PView yout;            // first plane of the 64-channel output slice
  int B, H, W;
  int c, cout;           // half width (16 or 32) and output channels (64 or 128)
  const __bf16* wA;      // Bottleneck.cv1 (c -> c/2, 3x3), conv_pack_weights order, MT = 1
  const float* bA;       // padded to 16
  const __bf16* wB;      // Bottleneck.cv2 (8 -> 16, 3x3), MT = 1, 3 k-steps
  const float* bB;
  const __bf16* wC;      // C3k2.cv2 (48 -> 64, 1x1), MT = 4, 2 k-steps
  const float* bC;
  unsigned in_bytes, out_bytes;
  int st_nt;             // non-temporal stores (st_nt_policy), filled by c3k2s_launch
  int cacheable;         // 1: never (yl_op.reserved3 bit 8)
  long long plane_px;    // pixels per input plane
  int tiles_x, tiles_y;
};
int c3k2s_launch(const C3k2sArgs& a, hipStream_t s);

// bneck2.hip: a whole Bottleneck (3x3 c -> c, 3x3 c -> c, + x) in one launch, c = 32 or 64; x and y in different buffers
struct Bneck2Args {
  PView xin, yout;
  int B, H, W, c;
  const __bf16* w1;      // conv_pack_weights order, MT = c / 16, 9 * c / 32 k-steps
  const float* b1;       // padded to MT * 16
  const __bf16* w2;
  const float* b2;
  unsigned in_bytes, out_bytes;
  int st_nt;             // non-temporal stores (st_nt_policy), filled by bneck2_launch
  long long plane_px;
  int tiles_x, tiles_y;
};
int bneck2_launch(const Bneck2Args& a, hipStream_t s);

// resident.hip: the body of a C3k block (inner width 64) in one launch, one workgroup per image, the maps in LDS
struct C3kArgs {
  const __bf16* x;  // first plane of the block's input slice (128 channels)
  __bf16* y;        // first plane of its output slice (128 channels)
  unsigned x_plane, y_plane;  // bytes per plane in HBM
  unsigned x_bytes, y_bytes;  // buffer-descriptor sizes
  int B, H, W, W1, img, g0;   // guarded geometry of the map in HBM (input and output share it)
  const __bf16* w[6];         // packed filters: cv1|cv2, m0.cv1, m0.cv2, m1.cv1, m1.cv2, cv3  ([cb][k-step][4 tiles][64 lanes][8])
  const float* bias[6];       // padded to 64-channel blocks
  int ntiles;                 // 16-pixel tiles per image
  int xpl, apl;               // LDS bytes per dense plane / per guarded plane
  int r1;                     // LDS byte offset of region 1
  int scratch;                // LDS byte offset of a 64-byte scratch area (target of the filter warm-up loads)
  int dbg;                    // development switches (YL_RES_DBG), 0 in production
  int split, Hs;              // workgroups per image and the rows each works on (filled by c3k_resident_launch)
  unsigned long long* stamps; // diagnostic builds (-DYL_STAMPS): [workgroup][wave][32] x {s_memtime, s_memrealtime}; else unused
};
bool c3k_resident_fits(int c_, int H, int W);
int c3k_resident_launch(C3kArgs a, hipStream_t s);

// conv3_band.hip: 3x3 stride-1 conv, 64 output channels, 64 / 128 / 256 input channels, a band of rows per workgroup with its
// input in LDS and the filter streamed from L2 (Detect's first box convs on the small levels)
struct BandArgs {
  const __bf16* x;
  __bf16* y;
  unsigned x_plane, y_plane, x_bytes, y_bytes;
  int H, W, W1, img, g0;  // guarded geometry (input and output share it)
  int Cin, act;
  const __bf16* w;        // conv_pack_weights order, one block of four tiles: [k-step][4][64 lanes][8]
  const float* bias;      // padded to 64
  int TR, nbands, ntiles, tpl;  // filled by conv3_band_launch: rows per band, bands per image, 16-pixel tiles per band, LDS bytes per plane
  int st_nt;                    // non-temporal stores (st_nt_policy), filled by the launchers
  unsigned long long* stamps;   // diagnostic builds (-DYL_STAMPS)
  // conv3_band64_kernel with a fused follow-up 1x1 conv 64 -> 64 (conv_kernel's POST): its filter ([2 k-steps][4 tiles][64 lanes][8]),
  // bias (padded to 64) and activation flag; `y` is then the follow-up conv's output
  const __bf16* w2;
  const float* bias2;
  int post_act;
};
bool conv3_band_plan(int Cin, int Cout, int H, int W, BandArgs* a, bool single = false);  // single: the one-conv-per-launch form (512 input channels too)
int conv3_band_launch(const BandArgs* args, int n, int batch, hipStream_t s);
// ... one conv per launch (128 / 256 / 512 -> 64; conv3_band_plan(..., post = true) is its plan)
int conv3_band1_launch(const BandArgs& arg, int batch, hipStream_t s);

// resident.hip: the PSABlock of a C2PSA layer (c = 128, two heads) in one launch, one workgroup per image
struct PsaArgs {
  const __bf16* b;          // first plane of the block's residual stream in the INPUT buffer (128 channels; never written)
  __bf16* b_out;            // ... in the output buffer
  const __bf16* a_in;       // first plane of the other half of C2PSA.cv1's output (128 channels): copied to a_out
  __bf16* a_out;
  unsigned b_plane, b_bytes;
  int B, H, W, W1, img, g0;
  float scale;              // attention score scale (key_dim ** -0.5)
  const __bf16* w_qkv;      // packed [4 blocks][4 k-steps][4 tiles][64 lanes][8]; bias padded to 256
  const float* b_qkv;
  const float* w_pe;        // depthwise taps [9][128] fp32, bias [128]
  const float* b_pe;
  const __bf16* w_tail;     // proj (32 KiB) | ffn.0 (64 KiB) | ffn.1 (64 KiB), each packed in 64-channel blocks
  const float* b_proj;
  const float* b_ffn0;
  const float* b_ffn1;
  unsigned long long* stamps;  // diagnostic builds (-DYL_STAMPS)
  int split;                   // workgroups per image (filled by psa_resident_launch)
};
bool psa_resident_fits(int c, int heads, int key_dim, int head_dim, int H, int W);
int psa_resident_launch(const PsaArgs& a, hipStream_t s);

// psa_tail.hip: the pointwise tail of a C2PSA layer (proj, ffn.0, ffn.1 of its PSABlock, then C2PSA.cv2) in one launch, pixels of
// the whole batch as one list
struct PsaTailArgs {
  const __bf16* o;          // first plane of the attention's output + pe (128 channels)
  const __bf16* u;          // first plane of cat(a, b) = C2PSA.cv1's output (256 channels; b = planes 16 .. 31)
  __bf16* y;                // first plane of the layer's output (256 channels)
  unsigned o_plane, u_plane, y_plane, o_bytes, u_bytes, y_bytes;
  int B, H, W, W1, img, g0;
  const __bf16* w_tail;     // proj (32 KiB) | ffn.0 (64) | ffn.1 (64) | cv2 (128), each packed in 64-channel blocks
  const float* b_proj;
  const float* b_ffn0;
  const float* b_ffn1;
  const float* b_cv2;
  int act;                  // cv2's activation (SiLU)
  unsigned long long* stamps;  // diagnostic builds (-DYL_STAMPS)
};
bool psa_tail_fits(int c, int c2, int n_blocks);
int psa_tail_launch(const PsaTailArgs& a, hipStream_t s);

// sppf_resident.hip: an SPPF layer (256 -> 128, pools, 512 -> 256) and optionally the 1x1 conv 256 -> 256 that alone consumes its
// output, in one launch, one or two workgroups per image
struct SppfArgs {
  const __bf16* x;          // first plane of the layer's input (256 channels)
  __bf16* y;                // first plane of the output (256 channels): the tail conv's output, or SPPF's own without a tail
  unsigned x_plane, y_plane, x_bytes, y_bytes;
  int B, H, W, W1, img, g0;
  const __bf16* w1;         // SPPF.cv1, packed [2 blocks][8 k-steps][4 tiles][64 lanes][8]
  const __bf16* w2;         // SPPF.cv2, packed [4][16][4][64][8] (256 KiB)
  const __bf16* w3;         // the follow-up conv, packed [4][8][4][64][8] (128 KiB); null: none
  const float *b1, *b2, *b3;
  unsigned long long* stamps;  // diagnostic builds (-DYL_STAMPS)
  int split;                   // workgroups per image (filled by sppf_resident_launch)
  int dbg;                     // development switches (YL_SPPF_DBG), 0 in production
};
bool sppf_resident_fits(int c1, int c2, int tail, int H, int W);
int sppf_resident_launch(const SppfArgs& a, hipStream_t s);

struct DecodeLevel {
  const float4* xywh;  // != nullptr: the level's boxes decoded by the producing conv's epilogue (yl_level.offset_xywh), [B][H * W]; `box` unused
  PView box;     // 8 planes: DFL logits (head.py:64 cv2)
  PView cls;     // nc8 / 8 planes: class logits (cv3); padding channels hold large negative values
  int H, W, a0;  // a0: first anchor index of this level
  float stride;
};
inline PView level_view(const void* workspace, int64_t offset, int batch, int H, int W) {
  const Geo g = geo_of(batch, H, W);
  PView p;
  p.base = reinterpret_cast<__bf16*>((char*)workspace + offset);
  p.plane = (long long)g.plane_px * 8;
  p.W1 = g.W1;
  p.img = g.img;
  p.g0 = g.g0;
  return p;
}
inline DecodeLevel decode_level(const void* workspace, const yl_level& L, int batch, int a0) {
  DecodeLevel d;
  d.xywh = L.offset_xywh >= 0 ? reinterpret_cast<const float4*>((const char*)workspace + L.offset_xywh) : nullptr;
  d.box = level_view(workspace, L.offset >= 0 ? L.offset : L.offset_cls, batch, L.H, L.W);  // (geometry is shared with `cls`)
  d.cls = level_view(workspace, L.offset_cls, batch, L.H, L.W);
  d.H = L.H;
  d.W = L.W;
  d.a0 = a0;
  d.stride = L.stride;
  return d;
}
// 8 consecutive channels of plane c8 at pixel pix, as fp32
__device__ __forceinline__ void level_load8(const PView& v, int c8, int pix, float (&o)[8]) {
  const bf16x8 t = *reinterpret_cast<const bf16x8*>(v.base + (long long)c8 * v.plane + (long long)pix * 8);
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (float)t[j];
}
bool conv_pf_supported(const ConvConfig& cfg);  // conv_mfma.hip: conv_kernel can feed the NMS pre-filter from this conv's epilogue
int decode_launch(const DecodeLevel* lv, int n_levels, int batch, int nc, int A, float* y, hipStream_t s);
// nms.hip: where the class branch's last launch appends the anchors whose best class logit could clear the threshold
// (the first stage of yl_decode_nms, decode_candidates_sparse_kernel, done in the producing kernel's epilogue)
constexpr int PF_NSUB = 16;  // sub-lists per image (= nms.hip NSUB)
struct PrefilterSink {
  int32_t* pre_count;  // [B][PF_NSUB], zeroed by prefilter_sink_prepare
  int32_t* pre_list;   // [B][PF_NSUB][sub_cap]
  int sub_cap;
  float logit_lo;
  int32_t* zero_ptr;   // defer_zero: zero_n counters the network's FIRST launch has to clear (nullptr: already cleared)
  int zero_n;
};
// Lays out `work` as yl_decode_nms does, zeroes the counters on `stream` and fills `sink`; *active = 0 when the dense
// decode path applies (conf < 0.05): nothing is done then and the caller must not skip the pre-filter.
// defer_zero: no zeroing launch -- sink->zero_ptr / zero_n name the counters, and the caller promises that a launch in front
// of every class-logit launch clears them (stem2_kernel does: Stem2Args::zero_ptr).
int prefilter_sink_prepare(const yl_level* levels, int n_levels, int batch, int max_cand, float conf_thres, void* work,
                           void* stream, PrefilterSink* sink, int* active, bool defer_zero = false);
// nms.hip: yl_decode_nms with optional stage events (profiling); prefiltered != 0: the counters were zeroed and the
// candidate lists filled by the network's own launches (prefilter_sink_prepare + ConvArgs::pf_*)
int decode_nms_stages(const void* workspace, const yl_level* levels, int n_levels, int batch, int nc, float conf_thres,
                      float iou_thres, int max_det, int agnostic, const int32_t* classes, int n_classes, int max_cand,
                      float max_wh, const float* scale, float* out, int32_t* out_count, void* work, void* stream,
                      hipEvent_t* ev, int prefiltered = 0);

// nrows contiguous rows of `n` floats -> LDS rows of `pitch` words starting at column c0 (256 threads).
// Loads are issued in batches of 8 per thread before the first LDS write: one memory round trip per
// 32 KB instead of one per 4 KB.
__device__ __forceinline__ void stage_rows(float* dst, int pitch, int c0, const float* __restrict__ src, int n,
                                           int nrows, int tid) {
  if ((n & 3) == 0) {  // 16-byte loads over the whole run
    const int n4 = nrows * n / 4;
    for (int i0 = tid; i0 < n4; i0 += 256 * 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 256;
        if (i < n4) v[u] = *reinterpret_cast<const f32x4*>(src + 4LL * i);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 256;
        if (i < n4) {
          const int e = 4 * i, r = e / n, c = e - r * n;
          float* d = dst + r * pitch + c0 + c;
          d[0] = v[u][0];
          d[1] = v[u][1];
          d[2] = v[u][2];
          d[3] = v[u][3];
        }
      }
    }
  } else {
    for (int i = tid; i < nrows * n; i += 256) {
      const int r = i / n, c = i - r * n;
      dst[r * pitch + c0 + c] = src[i];
    }
  }
}

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
// (cx, cy, w, h) of level-local anchor p of image b from the level's decoded boxes (DecodeLevel::xywh)
__device__ __forceinline__ float4 level_xywh(const DecodeLevel& L, int b, int p) {
  return L.xywh[(long long)b * (L.H * L.W) + p];
}

}  // namespace yl
