// Batched class-aware NMS on the GPU: candidate filter -> sort -> wavefront-ballot IoU mask ->
// greedy scan.  Replaces reference utils/ops.py:138-273 (non_max_suppression, single-label branch)
// including its call to torchvision.ops.nms (ops.py:265), and optionally ops.py:66-98 scale_boxes.
//
// Exactness: every comparison the reference makes is reproduced in fp32 with the same operation
// order -- xywh2xyxy (ops.py:386-389), the class offset add `boxes + cls * max_wh` (ops.py:258,264;
// the boxes are deliberately quantised by that add), IoU = inter / (a_i + a_j - inter) with strict
// `>` (torchvision semantics), strict `conf > thres` (ops.py:203,244), first-max class (ops.py:243).
// __f*_rn intrinsics stop the compiler from contracting mul+add into FMAs.  Equal scores are
// ordered by anchor index (the reference leaves ties to the sort implementation).
// The reference's wall-clock bail-out (ops.py:269-271) is a load-dependent defect and is not kept.
#include "yl_internal.h"

namespace yl {

struct NmsWs {
  int32_t* count;       // [B]
  int32_t* nfinal;      // [B]
  unsigned long long* keys;  // [B][KP]   KP = pow2 >= A
  int32_t* cls_of;      // [B][A]
  float4* box_off;      // [B][M]
  float4* box_raw;      // [B][M]
  float* conf;          // [B][M]
  int32_t* cls;         // [B][M]
  unsigned long long* mask;  // [B][M][MW]
  float4* cbox;         // [B][A] (cx, cy, w, h) of candidate anchors (fused decode path only)
  int KP, M, MW;
};

static inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }
static int pow2_ge(int x) {
  int p = 64;
  while (p < x) p <<= 1;
  return p;
}

static int64_t nms_layout(int B, int A, int max_cand, char* base, NmsWs* w) {
  const int M = max_cand < A ? max_cand : A;
  const int MW = (M + 63) / 64;
  const int KP = pow2_ge(A);
  int64_t o = 0;
  auto take = [&](int64_t bytes) {
    int64_t r = o;
    o += align256(bytes);
    return r;
  };
  const int64_t o_count = take(4LL * B), o_nf = take(4LL * B), o_keys = take(8LL * B * KP),
                o_clsof = take(4LL * B * A), o_boff = take(16LL * B * M), o_braw = take(16LL * B * M),
                o_conf = take(4LL * B * M), o_cls = take(4LL * B * M), o_mask = take(8LL * B * M * MW),
                o_cbox = take(16LL * B * A);
  if (w) {
    w->cbox = (float4*)(base + o_cbox);
    w->count = (int32_t*)(base + o_count);
    w->nfinal = (int32_t*)(base + o_nf);
    w->keys = (unsigned long long*)(base + o_keys);
    w->cls_of = (int32_t*)(base + o_clsof);
    w->box_off = (float4*)(base + o_boff);
    w->box_raw = (float4*)(base + o_braw);
    w->conf = (float*)(base + o_conf);
    w->cls = (int32_t*)(base + o_cls);
    w->mask = (unsigned long long*)(base + o_mask);
    w->KP = KP;
    w->M = M;
    w->MW = MW;
  }
  return o;
}

// ---- 1. candidates: max class score > conf (strict), first max wins, optional class filter
__global__ void nms_candidates_kernel(const float* __restrict__ pred, int nc, int A, float conf_thres,
                                      const int32_t* __restrict__ classes, int n_classes, NmsWs w) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (a >= A) return;
  const float* p = pred + (long long)b * (4 + nc) * A + 4LL * A + a;
  float best = p[0];
  int bi = 0;
  for (int c = 1; c < nc; ++c) {
    const float v = p[(long long)c * A];
    if (v > best) {
      best = v;
      bi = c;
    }
  }
  if (!(best > conf_thres)) return;
  if (classes) {
    bool ok = false;
    for (int i = 0; i < n_classes; ++i) ok |= (classes[i] == bi);
    if (!ok) return;
  }
  const int slot = atomicAdd(&w.count[b], 1);
  const unsigned sb = __float_as_uint(best);  // positive floats order like their bit patterns
  w.keys[(long long)b * w.KP + slot] = ((unsigned long long)(~sb) << 32) | (unsigned)a;
  w.cls_of[(long long)b * A + a] = bi;
}

// ---- 2. sort by (score desc, anchor asc); gather boxes in xyxy + class offset
constexpr int SORT_LDS_KEYS = 8192;

__device__ __forceinline__ void bitonic_sort(unsigned long long* k, int np, int tid, int nthreads) {
  for (int size = 2; size <= np; size <<= 1) {
    for (int j = size >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < np; i += nthreads) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long x = k[i], y = k[ixj];
          const bool asc = (i & size) == 0;
          if ((x > y) == asc) {
            k[i] = y;
            k[ixj] = x;
          }
        }
      }
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(1024) void nms_sort_kernel(const float* __restrict__ pred, int nc, int A, int agnostic,
                                                         float max_wh, NmsWs w) {
  __shared__ unsigned long long sk[SORT_LDS_KEYS];
  const int b = blockIdx.x, tid = threadIdx.x;
  int n = w.count[b];
  if (n > A) n = A;
  int np = 64;
  while (np < n) np <<= 1;
  unsigned long long* gk = w.keys + (long long)b * w.KP;
  unsigned long long* k = (np <= SORT_LDS_KEYS) ? sk : gk;
  if (k == sk) {
    for (int i = tid; i < np; i += 1024) sk[i] = i < n ? gk[i] : ~0ull;
  } else {
    for (int i = n + tid; i < np; i += 1024) gk[i] = ~0ull;
  }
  __syncthreads();
  if (n > 1) bitonic_sort(k, np, tid, 1024);
  const int nf = n < w.M ? n : w.M;  // keep the top max_cand by score (ops.py:254-255)
  if (tid == 0) w.nfinal[b] = nf;
  const float* pb = pred ? pred + (long long)b * (4 + nc) * A : nullptr;
  for (int i = tid; i < nf; i += 1024) {
    const unsigned long long key = k[i];
    const int a = (int)(key & 0xffffffffu);
    const float score = __uint_as_float(~(unsigned)(key >> 32));
    const int cls = w.cls_of[(long long)b * A + a];
    float cx, cy, bw, bh;
    if (pb) {
      cx = pb[a], cy = pb[A + a], bw = pb[2LL * A + a], bh = pb[3LL * A + a];
    } else {
      const float4 t = w.cbox[(long long)b * A + a];
      cx = t.x, cy = t.y, bw = t.z, bh = t.w;
    }
    const float hw = __fdiv_rn(bw, 2.0f), hh = __fdiv_rn(bh, 2.0f);  // ops.py:386
    const float x1 = __fsub_rn(cx, hw), y1 = __fsub_rn(cy, hh), x2 = __fadd_rn(cx, hw), y2 = __fadd_rn(cy, hh);
    const float c = __fmul_rn((float)cls, agnostic ? 0.0f : max_wh);  // ops.py:258
    const long long o = (long long)b * w.M + i;
    w.box_raw[o] = make_float4(x1, y1, x2, y2);
    w.box_off[o] = make_float4(__fadd_rn(x1, c), __fadd_rn(y1, c), __fadd_rn(x2, c), __fadd_rn(y2, c));
    w.conf[o] = score;
    w.cls[o] = cls;
  }
}

// ---- 3. suppression mask: one wavefront per row i, one ballot per 64 columns j > i
__device__ __forceinline__ bool iou_gt(const float4 a, float area_a, const float4 b, float thr) {
  const float area_b = __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y));
  const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
  const float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
  const float iw = fmaxf(0.0f, __fsub_rn(xx2, xx1)), ih = fmaxf(0.0f, __fsub_rn(yy2, yy1));
  const float inter = __fmul_rn(iw, ih);
  const float uni = __fsub_rn(__fadd_rn(area_a, area_b), inter);
  return __fdiv_rn(inter, uni) > thr;
}

__global__ __launch_bounds__(256) void nms_mask_kernel(float iou_thres, NmsWs w) {
  const int b = blockIdx.y;
  const int n = w.nfinal[b];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwords = (n + 63) >> 6;
  const float4* bo = w.box_off + (long long)b * w.M;
  for (int i = blockIdx.x * 4 + wave; i < n; i += gridDim.x * 4) {
    const float4 bi = bo[i];
    const float area_i = __fmul_rn(__fsub_rn(bi.z, bi.x), __fsub_rn(bi.w, bi.y));
    unsigned long long* row = w.mask + ((long long)b * w.M + i) * w.MW;
    for (int wd = i >> 6; wd < nwords; ++wd) {
      const int j = wd * 64 + lane;
      bool sup = false;
      if (j < n && j > i) sup = iou_gt(bi, area_i, bo[j], iou_thres);
      const unsigned long long bits = __ballot(sup);
      if (lane == 0) row[wd] = bits;
    }
  }
}

// ---- 4. greedy scan over the sorted candidates, 64 at a time, one wavefront per image
constexpr int SCAN_T = 8;  // words per lane -> up to 64*64*8 = 32768 candidates

__global__ __launch_bounds__(64) void nms_scan_kernel(int max_det, const float* __restrict__ scale, float* out,
                                                       int32_t* out_count, NmsWs w) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int n = w.nfinal[b];
  const int nwords = (n + 63) >> 6;
  const unsigned long long* mask = w.mask + (long long)b * w.M * w.MW;
  unsigned long long rem[SCAN_T];
#pragma unroll
  for (int t = 0; t < SCAN_T; ++t) rem[t] = 0ull;
  float padx = 0.f, pady = 0.f, gain = 1.f, w0 = 0.f, h0 = 0.f;
  if (scale) {
    padx = scale[b * 5 + 0];
    pady = scale[b * 5 + 1];
    gain = scale[b * 5 + 2];
    w0 = scale[b * 5 + 3];
    h0 = scale[b * 5 + 4];
  }
  int total = 0;
  for (int blk = 0; blk < nwords && total < max_det; ++blk) {
    unsigned long long own = 0ull;
#pragma unroll
    for (int t = 0; t < SCAN_T; ++t)
      if (t == (blk >> 6)) own = rem[t];
    const unsigned long long cur = __shfl(own, blk & 63);
    const int i = blk * 64 + lane;
    const int nvalid = n - blk * 64;
    const unsigned long long vmask = nvalid >= 64 ? ~0ull : ((1ull << nvalid) - 1ull);
    unsigned long long alive = ~cur & vmask;
    const unsigned long long d = (i < n) ? mask[(long long)i * w.MW + blk] : 0ull;
    unsigned long long kept = 0ull;
    int room = max_det - total;
    while (alive && room > 0) {
      const int bit = __builtin_ctzll(alive);
      kept |= 1ull << bit;
      --room;
      const unsigned long long dw = __shfl(d, bit);
      alive &= ~dw;
      alive &= ~(1ull << bit);
    }
    // OR the rows of the kept candidates into the pending words of later blocks
    unsigned long long kk = kept;
    while (kk) {
      const int bit = __builtin_ctzll(kk);
      kk &= kk - 1;
      const unsigned long long* row = mask + (long long)(blk * 64 + bit) * w.MW;
#pragma unroll
      for (int t = 0; t < SCAN_T; ++t) {
        const int wd = lane + 64 * t;
        if (wd > blk && wd < nwords) rem[t] |= row[wd];
      }
    }
    // emit: lane `bit` writes its own detection at rank order
    if ((kept >> lane) & 1ull) {
      const int slot = total + __popcll(kept & ((1ull << lane) - 1ull));
      const long long o = (long long)b * w.M + i;
      float4 bx = w.box_raw[o];
      if (scale) {  // ops.py:66-98 scale_boxes + ops.py:276-295 clip_boxes
        bx.x = __fdiv_rn(__fsub_rn(bx.x, padx), gain);
        bx.y = __fdiv_rn(__fsub_rn(bx.y, pady), gain);
        bx.z = __fdiv_rn(__fsub_rn(bx.z, padx), gain);
        bx.w = __fdiv_rn(__fsub_rn(bx.w, pady), gain);
        bx.x = fminf(fmaxf(bx.x, 0.f), w0);
        bx.y = fminf(fmaxf(bx.y, 0.f), h0);
        bx.z = fminf(fmaxf(bx.z, 0.f), w0);
        bx.w = fminf(fmaxf(bx.w, 0.f), h0);
      }
      float* op = out + ((long long)b * max_det + slot) * 6;
      op[0] = bx.x;
      op[1] = bx.y;
      op[2] = bx.z;
      op[3] = bx.w;
      op[4] = w.conf[o];
      op[5] = (float)w.cls[o];
    }
    total += __popcll(kept);
  }
  if (lane == 0) out_count[b] = total;
}

}  // namespace yl

using namespace yl;

extern "C" int64_t yl_nms_workspace_bytes(int batch, int anchors, int max_cand) {
  if (batch <= 0 || anchors <= 0 || max_cand <= 0) return -1;
  return nms_layout(batch, anchors, max_cand, nullptr, nullptr);
}

extern "C" int yl_nms(const float* pred, int batch, int nc, int anchors, float conf_thres, float iou_thres,
                      int max_det, int agnostic, const int32_t* classes, int n_classes, int max_cand, float max_wh,
                      const float* scale, float* out, int32_t* out_count, void* work, void* stream) {
  if (!pred || !out || !out_count || !work) return fail(-1, "yl_nms: null pointer");
  if (batch <= 0 || nc <= 0 || anchors <= 0 || max_det <= 0 || max_cand <= 0)
    return fail(-1, "yl_nms: bad sizes");
  if (!(conf_thres >= 0.f && conf_thres <= 1.f) || !(iou_thres >= 0.f && iou_thres <= 1.f))
    return fail(-4, "yl_nms: thresholds must lie in [0, 1]");  // ops.py:186-187 AssertionError
  NmsWs w;
  nms_layout(batch, anchors, max_cand, (char*)work, &w);
  if (w.M > 64 * 64 * SCAN_T) return fail(-3, "yl_nms: max_cand above 32768 is not supported");
  hipStream_t s = (hipStream_t)stream;
  YL_HIP(hipMemsetAsync(w.count, 0, 4 * (size_t)batch, s));
  hipLaunchKernelGGL(nms_candidates_kernel, dim3((anchors + 255) / 256, batch), dim3(256), 0, s, pred, nc, anchors,
                     conf_thres, classes, classes ? n_classes : 0, w);
  hipLaunchKernelGGL(nms_sort_kernel, dim3(batch), dim3(1024), 0, s, pred, nc, anchors, agnostic, max_wh, w);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(64, batch), dim3(256), 0, s, iou_thres, w);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(batch), dim3(64), 0, s, max_det, scale, out, out_count, w);
  YL_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- fused Detect decode + candidates
// Same arithmetic as decode_kernel (shared device functions), but instead of materialising
// y[B][4+nc][A] (reference head.py:126, 90 MB at batch 32) only the anchors whose best class score
// clears the threshold are kept: key + class + (cx, cy, w, h).  One workgroup = 64 anchors.
namespace yl {
constexpr int DC_ANCH = 64;
struct DecCandArgs {
  DecodeLevel lv[8];
  int blk0[9];
  int n_levels, batch, nc, A;
  float conf_thres;
  const int32_t* classes;
  int n_classes;
};

__global__ __launch_bounds__(256) void decode_candidates_kernel(DecCandArgs a, NmsWs w) {
  extern __shared__ __attribute__((aligned(16))) float sraw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nb_img = a.blk0[a.n_levels];
  const int b = blockIdx.x / nb_img;
  const int k = blockIdx.x - b * nb_img;
  int l = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i)
    if (i < a.n_levels && k >= a.blk0[i]) l = i;
  const DecodeLevel L = a.lv[l];
  const int HW = L.H * L.W;
  const int p0 = (k - a.blk0[l]) * DC_ANCH;
  const int nv = min(DC_ANCH, HW - p0);
  const int no = 64 + a.nc, pitch = no + 1;
  float* sdist = sraw + DC_ANCH * pitch;       // [4][64]
  float* sbest = sdist + 4 * DC_ANCH;          // [4][64] best score per wave
  int* sbidx = reinterpret_cast<int*>(sbest + 4 * DC_ANCH);  // [4][64]
  const float* raw = L.raw + ((long long)b * HW + p0) * no;
  if ((no & 3) == 0) {  // the chunk is one contiguous run: 16-byte loads
    const int n4 = nv * no / 4;
    for (int i = tid; i < n4; i += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(raw + 4LL * i);
      const int e = 4 * i, r = e / no, c = e - r * no;
      float* d = sraw + r * pitch + c;
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
  } else {
    for (int r = wave; r < nv; r += 4)
      for (int c = lane; c < no; c += 64) sraw[r * pitch + c] = raw[(long long)r * no + c];
  }
  __syncthreads();
  float dist = 0.f;
  if (lane < nv) dist = dfl_expect(sraw + lane * pitch + wave * 16);
  sdist[wave * DC_ANCH + lane] = dist;
  // best class among c = wave, wave + 4, ...: strict '>' keeps the first maximum (ops.py:243)
  float best = -1.f;
  int bi = 0x7fffffff;
  if (lane < nv)
    for (int c = wave; c < a.nc; c += 4) {
      const float sc = sigmoid_score(sraw[lane * pitch + 64 + c]);
      if (sc > best) {
        best = sc;
        bi = c;
      }
    }
  sbest[wave * DC_ANCH + lane] = best;
  sbidx[wave * DC_ANCH + lane] = bi;
  __syncthreads();
  if (wave == 0 && lane < nv) {
#pragma unroll
    for (int q = 1; q < 4; ++q) {
      const float s2 = sbest[q * DC_ANCH + lane];
      const int i2 = sbidx[q * DC_ANCH + lane];
      if (s2 > best || (s2 == best && i2 < bi)) {
        best = s2;
        bi = i2;
      }
    }
    bool keep = best > a.conf_thres;
    if (keep && a.classes) {
      bool ok = false;
      for (int i = 0; i < a.n_classes; ++i) ok |= (a.classes[i] == bi);
      keep = ok;
    }
    if (keep) {
      const int p = p0 + lane, an = L.a0 + p;
      const float ax = (float)(p % L.W) + 0.5f, ay = (float)(p / L.W) + 0.5f;
      const float4 bx = box_xywh(ax, ay, sdist[lane], sdist[DC_ANCH + lane], sdist[2 * DC_ANCH + lane],
                                 sdist[3 * DC_ANCH + lane], L.stride);
      const int slot = atomicAdd(&w.count[b], 1);
      w.keys[(long long)b * w.KP + slot] = ((unsigned long long)(~__float_as_uint(best)) << 32) | (unsigned)an;
      w.cls_of[(long long)b * a.A + an] = bi;
      w.cbox[(long long)b * a.A + an] = bx;
    }
  }
}
}  // namespace yl

extern "C" int yl_decode_nms(const void* workspace, const yl_level* levels, int n_levels, int batch, int nc,
                             float conf_thres, float iou_thres, int max_det, int agnostic, const int32_t* classes,
                             int n_classes, int max_cand, float max_wh, const float* scale, float* out,
                             int32_t* out_count, void* work, void* stream) {
  if (!workspace || !levels || !out || !out_count || !work) return fail(-1, "yl_decode_nms: null pointer");
  if (batch <= 0 || nc <= 0 || n_levels <= 0 || n_levels > 8 || max_det <= 0 || max_cand <= 0)
    return fail(-1, "yl_decode_nms: bad sizes");
  if (!(conf_thres >= 0.f && conf_thres <= 1.f) || !(iou_thres >= 0.f && iou_thres <= 1.f))
    return fail(-4, "yl_decode_nms: thresholds must lie in [0, 1]");
  DecCandArgs a;
  int nb = 0, A = 0;
  for (int i = 0; i < n_levels; ++i) {
    a.lv[i].raw = reinterpret_cast<const float*>((const char*)workspace + levels[i].offset);
    a.lv[i].H = levels[i].H;
    a.lv[i].W = levels[i].W;
    a.lv[i].a0 = A;
    a.lv[i].stride = levels[i].stride;
    a.blk0[i] = nb;
    nb += (levels[i].H * levels[i].W + DC_ANCH - 1) / DC_ANCH;
    A += levels[i].H * levels[i].W;
  }
  for (int i = n_levels; i < 8; ++i) {
    a.lv[i] = DecodeLevel{nullptr, 0, 0, 1 << 30, 0.f};
    a.blk0[i] = 1 << 30;
  }
  a.blk0[n_levels] = nb;
  a.n_levels = n_levels;
  a.batch = batch;
  a.nc = nc;
  a.A = A;
  a.conf_thres = conf_thres;
  a.classes = classes;
  a.n_classes = classes ? n_classes : 0;
  NmsWs w;
  nms_layout(batch, A, max_cand, (char*)work, &w);
  if (w.M > 64 * 64 * SCAN_T) return fail(-3, "yl_decode_nms: max_cand above 32768 is not supported");
  const size_t lds = ((size_t)DC_ANCH * (64 + nc + 1) + 12 * DC_ANCH) * sizeof(float);
  if (lds > 150 * 1024) return fail(-3, "yl_decode_nms: nc too large");
  static size_t attr = 0;
  if (lds > attr) {
    YL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(decode_candidates_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = 160 * 1024;
  }
  hipStream_t s = (hipStream_t)stream;
  YL_HIP(hipMemsetAsync(w.count, 0, 4 * (size_t)batch, s));
  hipLaunchKernelGGL(decode_candidates_kernel, dim3(batch * nb), dim3(256), lds, s, a, w);
  hipLaunchKernelGGL(nms_sort_kernel, dim3(batch), dim3(1024), 0, s, (const float*)nullptr, nc, A, agnostic, max_wh, w);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(64, batch), dim3(256), 0, s, iou_thres, w);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(batch), dim3(64), 0, s, max_det, scale, out, out_count, w);
  YL_HIP(hipGetLastError());
  return 0;
}
