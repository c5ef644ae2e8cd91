/* yl_amd.h -- C ABI of the MI355X-native YOLO11 detect inference path (libyl_amd.so).
 *
 * The reference (dongyunjinyu/YOLO-Lite @ 2025-03-03) has no FFI / plugin interface: its boundary
 * is the Python object API.  This header is the boundary underneath our
 * Python mirror of that API; each entry point names the reference code it replaces (paths relative
 * to /root/reference/yololite).  Plain pointers and sizes only -- no torch types.  Device pointers
 * come from the caller (PyTorch-ROCm allocations); `stream` is a hipStream_t passed as void*.
 *
 * All functions return 0 on success and a negative code on failure; yl_last_error() returns a
 * thread-local message.  Handles are not thread-safe individually (the reference predictor
 * serialises calls with a lock, engine/predictor.py:65,159, and so does ours).
 */
#ifndef YL_AMD_H
#define YL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YL_ABI_VERSION 21

/* element types */
enum { YL_BF16 = 0, YL_F32 = 1, YL_F16 = 2, YL_U8 = 3, YL_FP8 = 4 /* OCP e4m3 */ };

/* op kinds of an engine program */
enum {
  YL_OP_CONV = 1,     /* dense 1x1 / 3x3, stride 1 / 2, +bias, optional SiLU, optional residual  */
  YL_OP_DWCONV3 = 2,  /* depthwise 3x3 stride 1, +bias, optional SiLU, optional accumulate        */
  YL_OP_SPPF_POOL = 3,/* three chained MaxPool2d(5,1,2): writes pool5/9/13 after the input slice  */
  YL_OP_UPSAMPLE = 4, /* nearest upsample by `stride` (1 = plain copy) into a channel slice        */
  YL_OP_ATTENTION = 5,/* softmax(q^T k * scale) applied to v, per head (C2PSA)                     */
  YL_OP_STEM = 6,     /* first layer fused with the input conversion: 3x3 stride-2 conv on the 3-channel
                         NCHW network input (fp32 / fp16 / uint8, /255 for uint8) -> bf16 planes; must be
                         op 0; weight [Cout][3][3][3]                                              */
  YL_OP_QUANT = 9,    /* in (bf16 planes) -> out (fp8 planes, same dims): out = e4m3(in / out_scale), saturating, round to
                         nearest even; feeds the fp8 convs (BASELINE.json config 5: e4m3 weights + activations)        */
  YL_OP_BNECK2 = 8,   /* a whole Bottleneck(c, c, k = (3, 3), e = 1) in one launch (nn/modules/block.py:330-343; the
                         Bottlenecks of a C3k block): out = in + SiLU(cv2(SiLU(cv1(in)))), c = in.C = out.C = 32 or 64;
                         weight / bias = cv1 [c][c][3][3], post_weight / post_bias = cv2 (post_cin = 9c).  `in` and `out`
                         must be different buffers (a workgroup reads a halo of `in`).  The tensor between the convs lives
                         in LDS; bit-identical to the two YL_OP_CONV ops it replaces.                                */
  YL_OP_C3K = 10,     /* body of a C3k block (nn/modules/block.py:731-739 C3k; :245-259 C3.forward; :330-343 Bottleneck) with inner
                         width c_ = 64 on a map small enough for one workgroup's LDS (20x20 at 640x640: layers 8 and 22 of
                         YOLO11n), ONE launch, one workgroup per image (csrc/resident.hip):
                           u = SiLU(cv1|cv2(in))  (weight / bias = C3.cv1 and C3.cv2 stacked, [2 c_][in.C], 1x1),
                           a = u[0:c_] through two Bottlenecks a + SiLU(cv2(SiLU(cv1(a)))) (extra_weight[0..3] = m.0.cv1, m.0.cv2,
                           m.1.cv1, m.1.cv2, each [c_][c_][3][3]), out = SiLU(cv3(cat(a, u[c_:2 c_]))) (extra_weight[4],
                           [out.C][2 c_], 1x1).  in.C = out.C = 2 c_ = 128; in and out have the same H, W.  Every tensor between
                           `in` and `out` lives in LDS or registers; bit-identical to the six YL_OP_CONV ops it replaces.  */
  YL_OP_PSA = 11,     /* a PSABlock (nn/modules/block.py:919-953; Attention :863-916) with c = 128 channels, 2 heads, key_dim 32,
                         head_dim 64 on a map of at most 400 pixels (layer 10 of YOLO11n at 640x640), ONE launch, one workgroup per
                         image or two (csrc/resident.hip): in = cat(a, b) = C2PSA.cv1's output (256 channels), out = cat(a, b') in
                         a DIFFERENT buffer (the workgroups of an image read all of b at times of their own, so the input is
                         never written; a is copied), b = the block's residual stream:
                           qkv = conv1x1(b) (weight / bias [256][128], no activation); per head o = v . softmax(q^T k * scale)^T;
                           o += dwconv3x3(v) (extra_weight[0] [128][1][3][3], the positional encoding); b += conv1x1(o)
                           (extra_weight[1] [128][128]); b += conv1x1(SiLU(conv1x1(b))) (extra_weight[2] [256][128],
                           extra_weight[3] [128][256]); heads / key_dim / head_dim / scale as YL_OP_ATTENTION.  Bit-identical to the
                           six ops it replaces (YL_OP_CONV, YL_OP_ATTENTION, YL_OP_DWCONV3 with accumulate, three YL_OP_CONV). */
  YL_OP_SPPF = 12,    /* a whole SPPF layer (nn/modules/block.py:165-184) with 256 channels in and out (hidden width 128: YOLO11n's
                         layer 9) on a map of at most 448 pixels and, optionally, the 1x1 conv that alone consumes its output (C2PSA.cv1,
                         block.py:1009-1038, 256 -> 256), ONE launch, one or two workgroups per image (csrc/sppf_resident.hip):
                           x = SiLU(cv1(in)) (weight / bias [128][256]); y5, y9, y13 = the three chained 5x5 max pools of x;
                           s = SiLU(cv2(cat(x, y5, y9, y13))) (extra_weight[0] [256][512]); out = s, or with n_extra = 2
                           out = SiLU(conv1x1(s)) (extra_weight[1] [256][256]) and s is never stored.  x and the pooled maps live in
                           LDS; bit-identical to the YL_OP_CONV, YL_OP_SPPF_POOL, YL_OP_CONV [, YL_OP_CONV] ops it replaces.        */
  YL_OP_PSA_TAIL = 13,/* the pointwise tail of a C2PSA layer with ONE PSABlock of width c = 128 and 256 output channels (layer 10 of
                           YOLO11n; nn/modules/block.py:919-953, 1035-1038): in = o, the attention's output + pe (128 channels);
                           res = cat(a, b), C2PSA.cv1's output (256 channels, read only); out = the layer's output:
                           b += conv1x1(o) (weight / bias [128][128], `proj`); b += conv1x1(SiLU(conv1x1(b))) (extra_weight[0]
                           [256][128], extra_weight[1] [128][256], `ffn`); out = act(conv1x1(cat(a, b))) (extra_weight[2] [256][256],
                           C2PSA.cv2).  Pixels of the whole batch as one list, nothing between the convs is stored; bit-identical
                           to the four YL_OP_CONV ops it replaces.                                                          */
  YL_OP_C3K2S = 7     /* body of a narrow C3k2 block in one launch (nn/modules/block.py:220-235, 330-343), half width
                         c = 16 or 32: `in` = the block's concat buffer [a | b | ...] (a, b = halves of C3k2.cv1's output,
                         c channels each); c_out = b + SiLU(cv2(SiLU(cv1(b)))) with the Bottleneck's two 3x3 convs
                         (weight / bias = cv1 [c/2][c][3][3]; post_weight / post_bias = cv2 [c][c/2][3][3], post_cin =
                         9c/2); out = SiLU(C3k2.cv2(cat(a, b, c_out))) (stem_weight / stem_bias = that 1x1 conv,
                         [out.C][3c]; out.C = 64, or 128 with c = 32).  The intermediates live in LDS; bit-identical to
                         the three YL_OP_CONV ops it replaces.                                                      */
};

/* A channel-slice view of a buffer inside the workspace arena.
 *   YL_F32  : NHWC, buffer[B][H][W][pitch] (conv output option; not used by the YOLO11 lowering).
 *   YL_BF16 : "C8-planar, guarded": buffer[pitch / 8][plane_px][8] -- one plane per 8 channels, and
 *             inside a plane the whole batch as one array of rows, every row followed by one zero
 *             guard pixel, every image by one zero guard row (plus one guard row in front):
 *                 pixel(b, y, x) = g0 + b * (H + 1) * (W + 1) + y * (W + 1) + x,   g0 = W + 2
 *                 plane_px = (g0 + B * (H + 1) * (W + 1) + 1) rounded up to a multiple of 16
 *             so a 3x3 tap is a constant offset that lands on data or on a zero, and the 16 pixels x
 *             4 channel groups of an MFMA B fragment are 4 runs of 256 contiguous bytes.
 *             C, pitch and coff must be multiples of 8; a view is the plane range
 *             [coff / 8, (coff + C) / 8).  Guards are cleared by yl_engine_prepare_workspace and are
 *             never written by any kernel; buffers must not overlap. */
typedef struct yl_view {
  int64_t offset; /* byte offset of the buffer from the workspace base; < 0 = absent        */
  int32_t H, W, C;/* logical dims per image                                                  */
  int32_t pitch;  /* channels of the underlying buffer                                       */
  int32_t coff;   /* first channel of this view inside the buffer                            */
  int32_t dtype;  /* YL_BF16, YL_F32 or YL_FP8                                               */
} yl_view;
/* YL_FP8 views: the bf16 layout with 16 channels per plane -- buffer[C/16][plane_px][16] e4m3 bytes, same guarded
 * pixel geometry, 16 bytes per pixel; C, pitch and coff multiples of 16.  A real value is e4m3 x scale of its tensor. */

/* One op.  Weights are HOST pointers to BN-folded fp32 data; yl_engine_create packs them
 * (bf16, MFMA fragment order) into device memory it owns.
 *   conv   : weight [Cout][Cin][k][k], bias [Cout]      (reference nn/modules/conv.py:35-53 with
 *            BN folded as utils/torch_utils.py:182-209; Bottleneck residual block.py:341-343)
 *   dwconv : weight [C][1][3][3],   bias [C]            (conv.py:100-105)
 *   pool   : nn/modules/block.py:180-184                 upsample: nn.Upsample in yolo11.yaml:31,35
 *   attn   : nn/modules/block.py:905-913                 (in = qkv map, out = attended v)        */
typedef struct yl_op {
  int32_t kind;
  int32_t k;          /* kernel size (1 or 3)                                        */
  int32_t stride;     /* conv stride (1|2) or upsample factor (1|2)                  */
  int32_t act;        /* 1 = SiLU                                                    */
  int32_t accumulate; /* dwconv: out += result                                       */
  int32_t heads, key_dim, head_dim; /* attention geometry                            */
  float scale;        /* attention score scale                                       */
  /* dwconv input channel map: in_channel(c) = (c / grp) * grp_stride + grp_off + c % grp;
     grp == 0 means identity.  Used for Attention.pe on the v channels (block.py:913). */
  int32_t grp, grp_stride, grp_off;
  int32_t cin_weight; /* conv: input channels of `weight` if fewer than in.C (stem: 3 of 8); 0 = in.C */
  int32_t lane;       /* 0 = main stream.  k > 0: side lane k -- the op starts once every earlier op of the
                         program has finished, runs concurrently with LATER main-lane ops and is joined
                         before the decode (the three Detect levels' box / class branches)          */
  yl_view in, out, res; /* res.offset < 0: no residual; res has the geometry of out           */
  const float* weight;
  const float* bias;
  /* conv only: a 1x1 stride-1 conv fused behind this one (its only consumer), applied in the epilogue
     so that the intermediate tensor never exists.  post_cin = output channels of THIS conv (32 or 64,
     `weight` is [post_cin][Cin][k][k]); post_weight [out.C][post_cin], post_bias [out.C]; `act` belongs
     to this conv, post_act to the fused one; out is the fused conv's output; no residual.
     post_weight == NULL: plain conv. */
  int32_t post_cin, post_act;
  const float* post_weight;
  const float* post_bias;
  /* 1x1 stride-1 conv only: in2.offset >= 0 makes the conv read cat(upsample_nearest(in2, up), in) along the
     channels (yolo11.yaml:31-36 nn.Upsample + Concat) without that tensor existing: the first in2.C input
     channels come from in2 at pixel (y / up, x / up).  up must be 2; in2.C and in.C multiples of 32;
     `weight` is [Cout][in2.C + in.C]. */
  yl_view in2;
  int32_t up;
  int32_t group;      /* > 0: consecutive ops with the same value that are the same conv on different maps (the three Detect
                         levels, head.py:59-62) may run as ONE launch; 0 = on its own.  A hint: the results are the same */
  /* 3x3 stride-2 conv that is op 0 only: stem_weight != NULL fuses the network's first layer in front of it
     (Conv(3 -> in.C, 3x3, s2) on the NCHW network input, weight [in.C][3][3][3], bias [in.C], SiLU if stem_act):
     `in` then describes a tensor that is never materialised (its offset is ignored); in.C must be 16 or 32
     (YOLO11n / s) and this conv's own output channels 32 or 64.  Results are bit-identical to YL_OP_STEM followed by the plain conv. */
  const float* stem_weight;
  const float* stem_bias;
  int32_t stem_act, reserved2;
  /* YL_OP_CONV with in.dtype == YL_FP8 (csrc/conv_f8.hip; plain conv: no post / in2 / stem fusion, in.C % 16 == 0,
     out.C % 32 == 0 or % 48 == 0): the filters are quantised to e4m3 per output channel at create time, `in` holds
     e4m3(x / in_scale); out may be bf16 planes or, with out.dtype == YL_FP8, e4m3(y / out_scale); res (bf16, or fp8
     with res_scale) is added after the activation as in the bf16 conv.  YL_OP_QUANT uses out_scale. */
  float in_scale, out_scale, res_scale;
  int32_t out8_c0;
  /* YL_OP_CONV (plain, bf16 or fp8 input, bf16 output) and YL_OP_STEM: out8.offset >= 0 makes the epilogue ALSO store
     e4m3(y / out8_scale) of output channels [out8_c0, out8_c0 + out8.C) into the fp8 view out8 (same H, W) -- the copy an
     fp8 conv reads -- so no YL_OP_QUANT pass over the tensor is needed.  out8_c0 % 8 == 0, out8.C % 16 == 0.
     out8_only != 0 (YL_OP_STEM, and a plain YL_OP_CONV whose copy covers all output channels): the op stores nothing but that copy --
     every reader of its output is an fp8 conv (the lowering proves that no op of the program reads the bf16 planes). */
  yl_view out8;
  float out8_scale;
  int32_t out8_only;
  /* ops made of more than three convs (YL_OP_C3K): n_extra further BN-folded fp32 filters / biases, arrays of HOST pointers
     (read by yl_engine_create only, like weight / bias) */
  int32_t n_extra, reserved3;        /* reserved3: YL_OP_C3K: workgroups per image (2: each takes half of the rows + a halo of four -- the
                                        latency profile of a caller that runs one batch at a time; 0 / 1: one).
                                        YL_OP_CONV with stem_weight and YL_OP_C3K2S: bit 8 (0x100) = keep the output cacheable --
                                        no non-temporal hint on its stores: its reader is the next launch of a caller that runs
                                        one batch at a time (the latency profile); results are the same either way */
  const float* const* extra_weight;
  const float* const* extra_bias;
  /* YL_OP_CONV that produces the 64 DFL logits of a Detect level (a 1x1 conv with out.C == 64, alone or as the fused follow-up
     `post_weight` of the 3x3 conv in front of it; head.py:36-40 cv2[i][2]): dfl_level = 1 + index of that level in the engine's
     `levels` (0 = off).  The epilogue then runs the level's box decode -- DFL (block.py:51-69: softmax over the 16 bins of each
     side, expectation) and dist2bbox(xywh) * stride (tal.py:341-350, head.py:105-106) -- ON THE fp32 ACCUMULATORS and stores
     (cx, cy, w, h) fp32 at yl_level.offset_xywh; the 64 logits are rounded to bf16 and stored at `out` only with dfl_keep != 0
     (debug / parity tests), otherwise `out` is not written.  ABI v21. */
  int32_t dfl_level, dfl_keep;
  /* YL_OP_DWCONV3 with post_weight (depthwise 3x3 -> pointwise 1x1 in one launch) AND stem_weight: a further 1x1
     conv behind the pointwise one (the nn.Conv2d(c3, nc, 1) that ends Detect's class branch, head.py:47), also in
     that launch: reserved2 = output channels of the pointwise conv (post_weight [reserved2][in.C]), stem_weight
     [out.C][reserved2], stem_bias [out.C], SiLU if stem_act; `out` is the last conv's output. */
} yl_op;

/* Detect-level raw maps (reference head.py:64-65 torch.cat(cv2, cv3)): two bf16 buffers in the planar
 * guarded activation layout (see yl_view) -- the 64 DFL logits at `offset` (8 planes) and the class
 * logits at `offset_cls` (nc8 / 8 planes; nc8 = nc rounded up to a multiple of 8, the padding channels
 * hold large negative logits).  Decode and NMS read them as fp32 (bf16 -> fp32 is exact). */
typedef struct yl_level {
  int64_t offset;
  int64_t offset_cls;
  int32_t H, W;
  float stride;
  int32_t nc8;
  /* ABI v21: >= 0: the level's boxes arrive DECODED -- fp32 [B][H * W][4] = (cx, cy, w, h) in input pixels (head.py:105-106
     dbox * strides), written by the epilogue of the conv that computes the DFL logits (yl_op.dfl_level) from its fp32
     accumulators; decode and NMS read 16 bytes per anchor instead of 64 bf16 logits, and the box error of bf16-stored logits
     (up to 0.3 stride) is gone.  `offset` may then be < 0 (no logits stored).  < 0: logits at `offset`, decoded by the readers. */
  int64_t offset_xywh;
} yl_level;

typedef struct yl_engine yl_engine;

const char* yl_last_error(void);
int yl_abi_version(void);
/* Number of visible HIP devices, or a negative code.  Does not create a context. */
int yl_device_count(void);

/* Build an engine for a fixed (batch, in_h, in_w): packs weights, picks kernel configs.
 * Replaces nn/tasks.py:118-145 (BaseModel._predict_once layer loop) + nn/autobackend.py:34-124.
 * `input_view` is where the converted bf16 input (one plane, channels padded to 8) is written when
 * op 0 is not YL_OP_STEM. */
int yl_engine_create(const yl_op* ops, int n_ops, int batch, int in_h, int in_w,
                     const yl_view* input_view, const yl_level* levels, int n_levels, int nc,
                     int64_t workspace_bytes, yl_engine** out);
void yl_engine_destroy(yl_engine* e);

/* Clears a workspace (all guard pixels must read as zero) and remembers its address; forward calls
 * do this themselves the first time they see a workspace pointer.  Call it again after anything
 * else has written to the workspace. */
int yl_engine_prepare_workspace(yl_engine* e, void* workspace, void* stream);

/* Network forward: input NCHW (YL_F32 / YL_F16 in [0,1], or YL_U8 in [0,255] which is divided by
 * 255 as engine/predictor.py:82-84 does) -> raw level maps in the workspace (bf16) -> y[B][4+nc][A] fp32
 * decoded as nn/modules/head.py:95-126 (+ block.py:51-69 DFL, utils/tal.py:326-350).
 * y may be NULL (raw maps only).  Enqueues on `stream`; no host sync. */
int yl_engine_forward(yl_engine* e, const void* input, int in_dtype, void* workspace, float* y,
                      void* stream);
/* Same, replayed from a hipGraph captured on first use for this (input, workspace, y) triple. */
int yl_engine_forward_graph(yl_engine* e, const void* input, int in_dtype, void* workspace,
                            float* y, void* stream);
/* Times each op with HIP events on `stream` (`iters` launches each, after one warm-up pass).
 * ms[n_ops + 2]: [0] input conversion, [1..n_ops] ops, [n_ops+1] decode. Synchronises. */
int yl_engine_profile(yl_engine* e, const void* input, int in_dtype, void* workspace, float* y,
                      void* stream, int iters, float* ms);
/* The hot path as the benchmark runs it (network ops in program order on one stream, then the fused decode + NMS of
 * yl_decode_nms), timed launch by launch with HIP events, averaged over `iters` passes after one warm-up.
 *   ms_ops[n_ops]: the network ops;  ms_post[5]: {zero counters + class pre-filter (decode_candidates_*), exact
 *   decode of the survivors, per-image sort, suppression mask, scan + scale_boxes + output}
 * i.e. the kernels that replace nn/modules/head.py:95-126 and utils/ops.py:138-273.  Synchronises. */
int yl_engine_profile_detect(yl_engine* e, const void* input, int in_dtype, void* workspace, float conf_thres,
                             float iou_thres, int max_det, int max_cand, float* out, int32_t* out_count, void* work,
                             void* stream, int iters, float* ms_ops, float* ms_post);
int yl_engine_num_ops(const yl_engine* e);
/* 1 when yl_engine_detect_graph at this confidence threshold runs the NMS candidate pre-filter (first stage of
 * yl_decode_nms) in the epilogue of the launches that write the Detect levels' class logits: one launch fewer and the
 * class planes are never re-read; 0 when it stays a launch of its own (conf < 0.05, unfused class branch, YL_FUSE_PREFILTER=0). */
int yl_engine_prefilter_fused(const yl_engine* e, float conf_thres);
/* Algorithmic work of op i: flops (2*MAC), bytes read+written once; name is a static string.  The network input's dtype is a
 * call-time argument: the op that reads it prices it at 4 bytes per value (callers that know the dtype subtract the difference,
 * as HipEngine.profile_detect does). */
int yl_engine_op_info(const yl_engine* e, int i, double* flops, double* bytes, const char** kernel_name);

/* Detect decode alone (device pointers): raw level maps -> y[B][4+nc][A].  head.py:95-126. */
int yl_detect_decode(const void* workspace, const yl_level* levels, int n_levels, int batch, int nc,
                     float* y, void* stream);

/* Non-maximum suppression.  Replaces utils/ops.py:138-273 (single-label branch) including the
 * torchvision.ops.nms call at ops.py:265, and optionally utils/ops.py:66-98 scale_boxes +
 * ops.py:276-295 clip_boxes when `scale` is non-NULL.
 *   pred      [B][4+nc][A] fp32 (cx, cy, w, h, class scores); NOT modified
 *   classes   optional device int32[n_classes] filter (ops.py:247-248), NULL = all
 *   scale     optional device float[B][5] = {pad_x, pad_y, gain, orig_w, orig_h}
 *   out       [B][max_det][6] fp32 (x1, y1, x2, y2, conf, cls), score-descending
 *   out_count [B] int32
 *   work      device scratch of yl_nms_workspace_bytes(B, A, max_cand) bytes
 * max_cand caps candidates per image (reference max_nms = 30000, ops.py:254-255; top scores kept). */
int64_t yl_nms_workspace_bytes(int batch, int anchors, int max_cand);
int yl_nms(const float* pred, int batch, int nc, int anchors, float conf_thres, float iou_thres,
           int max_det, int agnostic, const int32_t* classes, int n_classes, int max_cand,
           float max_wh, const float* scale, float* out, int32_t* out_count, void* work, void* stream);

/* Multi-label NMS: the validator's postprocess (engine/validator.py:281-291 calls non_max_suppression with
 * multi_label=True; utils/ops.py:208, 239-241): EVERY (anchor, class) pair with score > conf_thres is a
 * candidate row, capped to the max_cand best by score, then the same class-aware NMS.  Same arguments as
 * yl_nms; `work` must hold yl_nms_ml_workspace_bytes(B, nc, A, max_cand) bytes (keys for up to nc * A pairs
 * per image).  With nc == 1 it degenerates to yl_nms, as the reference does.  Equal scores keep the
 * reference's row order (anchor-major, then class). */
int64_t yl_nms_ml_workspace_bytes(int batch, int nc, int anchors, int max_cand);
int yl_nms_ml(const float* pred, int batch, int nc, int anchors, float conf_thres, float iou_thres,
              int max_det, int agnostic, const int32_t* classes, int n_classes, int max_cand,
              float max_wh, const float* scale, float* out, int32_t* out_count, void* work, void* stream);

/* Fused Detect decode + NMS straight from the raw level maps in the workspace: identical results
 * to yl_detect_decode followed by yl_nms (same decode arithmetic, same candidate order), but
 * y[B][4+nc][A] (reference head.py:126) is never written or re-read.  `work` must hold
 * yl_nms_workspace_bytes(B, A, max_cand) bytes with A = sum of level H*W. */
int yl_decode_nms(const void* workspace, const yl_level* levels, int n_levels, int batch, int nc,
                  float conf_thres, float iou_thres, int max_det, int agnostic, const int32_t* classes,
                  int n_classes, int max_cand, float max_wh, const float* scale, float* out,
                  int32_t* out_count, void* work, void* stream);

/* The whole hot path -- yl_engine_forward (raw maps only) followed by yl_decode_nms -- replayed from
 * one hipGraph captured on first use for this exact argument tuple (every pointer and threshold is
 * baked in; the contents of input / classes / scale may change between calls).  One launch, no host
 * round trip between the network and the NMS. */
int yl_engine_detect_graph(yl_engine* e, const void* input, int in_dtype, void* workspace,
                           float conf_thres, float iou_thres, int max_det, int agnostic,
                           const int32_t* classes, int n_classes, int max_cand, float max_wh,
                           const float* scale, float* out, int32_t* out_count, void* work, void* stream);

/* The graph of yl_engine_detect_graph for this argument tuple WITHOUT launching it (captured on first use; that first call runs
 * the path once eagerly on `stream`): *graph_out is a handle for yl_graph_launch_many, owned by the engine and valid until
 * yl_engine_destroy.  Replaces nothing in the reference (its predictor finishes a batch before it touches the next,
 * engine/predictor.py:178-212); it exists so that a serving loop can issue many prepared steps from native code. */
int yl_engine_detect_graph_prepare(yl_engine* e, const void* input, int in_dtype, void* workspace, float conf_thres,
                                   float iou_thres, int max_det, int agnostic, const int32_t* classes, int n_classes,
                                   int max_cand, float max_wh, const float* scale, float* out, int32_t* out_count, void* work,
                                   void* stream, void** graph_out);
/* yl_engine_detect_graph for a caller whose input address changes from call to call -- the reference API hands the predictor a
 * fresh tensor per batch (engine/predictor.py:178-196, data/loaders.py:415-477) -- without a staging copy: ONE captured graph per
 * argument tuple regardless of `input`; before a launch with a new address the kernel nodes that read the input get it written
 * into their parameters (hipGraphExecKernelNodeSetParams).  Contract: no earlier launch of the same tuple is still pending
 * (one workspace slot = one batch at a time); `input` must stay valid until the launch has finished. */
int yl_engine_detect_graph_rebind(yl_engine* e, const void* input, int in_dtype, void* workspace, float conf_thres,
                                  float iou_thres, int max_det, int agnostic, const int32_t* classes, int n_classes, int max_cand,
                                  float max_wh, const float* scale, float* out, int32_t* out_count, void* work, void* stream);

/* Issues n prepared steps back to back: graphs[i] (from yl_engine_detect_graph_prepare) on HIP stream streams[i].  The host
 * side of the reference's `for batch in dataset:` loop (engine/predictor.py:178) for batches whose inputs are already staged:
 * one native call per round of batches in flight instead of one trip through Python per batch (~80 us each). */
int yl_graph_launch_many(void* const* graphs, void* const* streams, int n);

/* uint8 HWC BGR letterboxed images -> float NCHW RGB / 255 (engine/predictor.py:77-84). */
int yl_preprocess_u8(const uint8_t* bgr_hwc, int batch, int h, int w, float* out_nchw, void* stream);

/* LetterBox on the device (data/augment.py:636-681 LetterBox.__call__, then engine/predictor.py:77-81:
 * BGR -> RGB, HWC -> CHW) for ONE uint8 HWC BGR source image (device pointer, row pitch in bytes) into
 * dst_chw = the [3][H][W] uint8 slot of that image in the NCHW batch.  The image is resized to
 * (new_h, new_w) -- bilinear in OpenCV's 8-bit fixed-point convention, the same integer arithmetic as the
 * CPU restatement data/augment.py::resize_linear_u8 -- placed at (top, left), the rest filled with 114.
 * xtab / ytab: device arrays of new_w / new_h int32[4] = {i0, i1, c0, c1} (source indices and 11-bit
 * coefficients per output column / row, computed by the host exactly as the CPU restatement does); unused
 * when the size does not change.  The geometry (new size, pads) is LetterBox.geometry's. */
int yl_letterbox_u8(const uint8_t* src_bgr_hwc, int src_h, int src_w, int src_pitch, uint8_t* dst_chw, int H,
                    int W, int top, int left, int new_h, int new_w, const int32_t* xtab, const int32_t* ytab,
                    void* stream);

/* ---- multi-GPU exchange of the sharded path (BASELINE.json config 4) ---------------------------------
 * One process per GPU, weights replicated, images sharded; the ONLY exchanges are root -> ranks input slices and ranks ->
 * root packed result records (`b * max_det * 6` fp32 box words + `b` int32 counts per rank and step).  RCCL over xGMI, grouped
 * send / recv (the root's sends leave on all its links at once); no reduction anywhere.  Replaces nothing in the reference:
 * it has no distributed inference (utils/torch_utils.py:142-159 select_device ends on cuda:0 for predict; the per-image
 * independence the sharding rests on is utils/ops.py:219, the loop over images in non_max_suppression).  RCCL is loaded on the
 * first yl_comm_* call (librccl.so.1; YL_RCCL_LIB overrides the path); without it these calls fail, nothing else does.
 * A communicator belongs to the HIP device that was current in yl_comm_create; calls on one communicator must be issued in
 * the same order on every rank (RCCL's rule) and are not thread-safe.  `stream` is a hipStream_t: the exchange is enqueued
 * there and ordered with the engine launches on that stream like any kernel.  Operations of ONE communicator EXECUTE in the
 * order they were issued whatever streams they name (an operation that crosses a link waits for an event its predecessor on
 * the communicator recorded: RCCL matches sends to one peer in execution order and orders nothing between streams itself), so
 * a binder may put each exchange on the stream of the step it belongs to.  One that wants the scatters of step k + n to
 * overlap the gathers of step k creates TWO communicators, one per direction (yololite_amd/dist.py NativeShardedRunner does).
 * The calls record and wait on events of their own: issue them on live streams, not inside a stream capture (the engine's graphs
 * hold the compute; the exchanges go between graph launches).
 * STATUS: EXPERIMENTAL.  Exercised with a world of one on the GPU (tests/test_gpu_dist.py) and never yet with a peer on the
 * other end of a link -- no multi-GPU machine was available to this build; a two-process test is in place and skips itself
 * below two devices.  bench.py --gpus N uses torch.distributed (backend nccl = RCCL) unless YL_BENCH_COMM=native. */
#define YL_COMM_ID_BYTES 128
typedef struct yl_comm yl_comm;
/* One rank (any) creates the 128-byte id and hands it to the others by its own means (a file, a socket, MPI, a
 * torch.distributed store): the rendezvous itself is the binder's business. */
int yl_comm_get_unique_id(void* id_out);
int yl_comm_create(const void* id, int rank, int world, yl_comm** out);
void yl_comm_destroy(yl_comm* c);
int yl_comm_rank(const yl_comm* c);
int yl_comm_world(const yl_comm* c);
/* root: `send` = world slices of `bytes` (device memory, rank order); every rank receives its slice into `recv` (the root by
 * a device copy; recv may alias the root's own slice).  `send` is ignored on the other ranks. */
int yl_comm_scatter(yl_comm* c, const void* send, void* recv, int64_t bytes, int root, void* stream);
/* every rank sends `bytes` from `send`; the root receives world slices into `recv`, rank order.  `recv` is ignored elsewhere. */
int yl_comm_gather(yl_comm* c, const void* send, void* recv, int64_t bytes, int root, void* stream);
/* Everything enqueued on `stream` before the call has completed on EVERY rank when it returns (a one-word gather to rank 0
 * and a one-word scatter back, then a host wait on the stream): the bracket of a timed region. */
int yl_comm_barrier(yl_comm* c, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YL_AMD_H */
