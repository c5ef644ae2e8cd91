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
#include <cmath>

#include "yl_internal.h"

namespace yl {

constexpr int NSUB = PF_NSUB;

// Counters are cleared by a KERNEL, not by hipMemsetAsync: a memset captured into a hipGraph (the hot path's candidate counters)
// was found to stop clearing once other work ran eagerly in the process -- another engine's first run, a detect(graph=False) -- on
// the legacy null stream: the replayed graphs then ACCUMULATED candidates from call to call (counts 50 -> 86 -> 101 per image) or
// ended in a GPU memory fault (tests/test_gpu_streams.py after the allocator's layout changed; ROCm 7.2; NOTEBOOK section 13).
// A kernel node carries its own arguments.
__global__ void zero_i32_kernel(int32_t* __restrict__ p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0;
}
static inline hipError_t zero_i32(int32_t* p, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(zero_i32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, (int)n);
  return hipGetLastError();
}

struct NmsWs {
  int32_t* count;       // [B]
  // fused decode path: anchors that passed the logit pre-filter, in NSUB sub-lists per image (same-address
  // atomics serialise at ~0.3 us each in L2: one counter per image made both kernels atomic-bound)
  int32_t* pre_count;   // [B][NSUB]
  int32_t* pre_list;    // [B][NSUB][sub_cap] anchor indices, unordered
  int sub_cap;          // = slots per sub-list, also the stride of the sub-lists inside keys[b][]
  int fused;            // 1: keys live in the sub-list slots (sentinel ~0 = rejected by the exact test)
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
  int ml_nc;            // > 0: multi-label mode, the low key word is anchor * nc + class (ops.py:239-241)
};

static inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }
static int pow2_ge(int x) {
  int p = 64;
  while (p < x) p <<= 1;
  return p;
}

static int64_t nms_layout(int B, int A, int max_cand, char* base, NmsWs* w, long long pairs = 0) {
  const long long cand = pairs > A ? pairs : A;  // candidate rows an image can have (multi-label: nc * A)
  const int M = max_cand < cand ? max_cand : (int)cand;
  const int MW = (M + 63) / 64;
  const int KP = pow2_ge((int)cand + 128 * NSUB);  // >= NSUB * sub_cap
  int64_t o = 0;
  auto take = [&](int64_t bytes) {
    int64_t r = o;
    o += align256(bytes);
    return r;
  };
  const int sub_cap = (A / (64 * NSUB) + 2) * 64;  // >= 64 * (64-anchor workgroups per sub-list)
  const int64_t o_count = take(4LL * B * (1 + NSUB)), o_nf = take(4LL * B),
                o_pre = take(4LL * B * NSUB * sub_cap), o_keys = take(8LL * B * KP),
                o_clsof = take(4LL * B * A), o_boff = take(16LL * B * M), o_braw = take(16LL * B * M),
                o_conf = take(4LL * B * M), o_cls = take(4LL * B * M), o_mask = take(8LL * B * M * MW),
                o_cbox = take(16LL * B * A);
  if (w) {
    w->cbox = (float4*)(base + o_cbox);
    w->count = (int32_t*)(base + o_count);
    w->pre_count = w->count + B;  // cleared by the same memset
    w->pre_list = (int32_t*)(base + o_pre);
    w->sub_cap = sub_cap;
    w->fused = 0;
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
    w->ml_nc = 0;
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

// ---- 1b. multi-label candidates (the validator's mode, ops.py:239-241): EVERY (anchor, class) pair with
// score > conf is a row.  One thread per anchor walks the classes; the appends of a wavefront are
// aggregated into one atomic per class step (a per-image counter would otherwise see up to nc * A
// same-address atomics).  Key = (~score, anchor * nc + class): equal scores keep torch.where order.
__global__ void nms_candidates_ml_kernel(const float* __restrict__ pred, int nc, int A, float conf_thres,
                                         const int32_t* __restrict__ classes, int n_classes, NmsWs w) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const float* p = pred + (long long)b * (4 + nc) * A + 4LL * A + (a < A ? a : 0);
  for (int c = 0; c < nc; ++c) {
    bool keep = false;
    float v = 0.f;
    if (a < A) {
      v = p[(long long)c * A];
      keep = v > conf_thres;
    }
    if (keep && classes) {
      bool ok = false;
      for (int i = 0; i < n_classes; ++i) ok |= (classes[i] == c);
      keep = ok;
    }
    const unsigned long long m = __ballot(keep);
    if (m == 0) continue;
    int base = 0;
    const int leader = __ffsll((long long)m) - 1;
    if (lane == leader) base = atomicAdd(&w.count[b], __popcll(m));
    base = __shfl(base, leader);
    if (keep) {
      const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
      if (slot < w.KP)
        w.keys[(long long)b * w.KP + slot] =
            ((unsigned long long)(~__float_as_uint(v)) << 32) | (unsigned)(a * nc + c);
    }
  }
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

// Sort of at most 1024 keys held in LDS by RANK: thread i counts the keys smaller than its own (every real key is unique: its
// low word is the candidate's index) and writes it to that slot of `out`; the sentinels ~0 of rejected candidates are equal to
// each other and stay where the pre-fill put them, behind the real keys.  One barrier instead of the bitonic network's
// log^2(n) / 2 (55 at 1024 keys): 9 -> 2 us per image at ~130 candidates (YOLO11n batch 32).  Same total order, same result.
__device__ __forceinline__ void rank_sort(const unsigned long long* k, unsigned long long* out, int n, int tid) {
  if (tid < n) out[tid] = ~0ull;
  __syncthreads();
  if (tid < n) {
    const unsigned long long x = k[tid];
    if (x != ~0ull) {
      int r = 0;
      for (int j = 0; j < n; ++j) r += k[j] < x ? 1 : 0;
      out[r] = x;
    }
  }
  __syncthreads();
}

// (a device function so that the one-launch path, decode_nms_fused_kernel, runs the very same code per image)
__device__ __forceinline__ void nms_sort_body(const float* __restrict__ pred, int nc, int A, int agnostic, float max_wh,
                                              const NmsWs& w, int b, int tid, unsigned long long* sk, int& s_nvalid) {
  unsigned long long* gk = w.keys + (long long)b * w.KP;
  int n, np = 64;
  unsigned long long* k;
  if (w.fused) {
    // keys sit in NSUB sub-lists (slot j of sub-list s at gk[s * sub_cap + j]); rejected anchors left ~0
    int pre[NSUB + 1];
    pre[0] = 0;
#pragma unroll
    for (int q = 0; q < NSUB; ++q) pre[q + 1] = pre[q] + min(max(w.pre_count[b * NSUB + q], 0), w.sub_cap);
    n = pre[NSUB];
    while (np < n) np <<= 1;
    if (np <= SORT_LDS_KEYS) {
      k = sk;
      for (int i = tid; i < np; i += 1024) {
        unsigned long long v = ~0ull;
        if (i < n) {
          int q = 0;
#pragma unroll
          for (int t = 1; t < NSUB; ++t) q += (i >= pre[t]);
          v = gk[q * w.sub_cap + (i - pre[q])];
        }
        sk[i] = v;
      }
    } else {  // too many for the LDS sort: sort the whole slot range in place (sentinels fill unused slots)
      k = gk;
      n = NSUB * w.sub_cap;
      np = 64;
      while (np < n) np <<= 1;
      for (int q = 0; q < NSUB; ++q)
        for (int i = w.pre_count[b * NSUB + q] + tid; i < w.sub_cap; i += 1024) gk[q * w.sub_cap + i] = ~0ull;
      for (int i = n + tid; i < np; i += 1024) gk[i] = ~0ull;
    }
  } else {
    n = w.count[b];
    const int nmax = w.ml_nc ? min(w.KP, A * w.ml_nc) : A;
    if (n > nmax) n = nmax;
    while (np < n) np <<= 1;
    k = (np <= SORT_LDS_KEYS) ? sk : gk;
    if (k == sk) {
      for (int i = tid; i < np; i += 1024) sk[i] = i < n ? gk[i] : ~0ull;
    } else {
      for (int i = n + tid; i < np; i += 1024) gk[i] = ~0ull;
    }
  }
  if (tid == 0) s_nvalid = 0;
  __syncthreads();
  static_assert(SORT_LDS_KEYS >= 2048, "rank_sort writes its output behind the input keys");
  if (n > 1 && n <= 1024 && k == sk) {
    rank_sort(sk, sk + 1024, n, tid);
    k = sk + 1024;
  } else if (n > 1) {
    bitonic_sort(k, np, tid, 1024);
  }
  // sentinels (fused path: anchors the exact test rejected) sort last: count the real keys
  for (int i = tid; i < n; i += 1024)
    if (k[i] != ~0ull && (i + 1 == n || k[i + 1] == ~0ull)) s_nvalid = i + 1;
  __syncthreads();
  n = s_nvalid;
  const int nf = n < w.M ? n : w.M;  // keep the top max_cand by score (ops.py:254-255)
  if (tid == 0) w.nfinal[b] = nf;
  const float* pb = pred ? pred + (long long)b * (4 + nc) * A : nullptr;
  for (int i = tid; i < nf; i += 1024) {
    const unsigned long long key = k[i];
    const unsigned lo = (unsigned)(key & 0xffffffffu);
    const int a = min((int)(w.ml_nc ? lo / (unsigned)w.ml_nc : lo), A - 1);  // never out of range, whatever the key holds
    const float score = __uint_as_float(~(unsigned)(key >> 32));
    const int cls = w.ml_nc ? (int)(lo % (unsigned)w.ml_nc) : w.cls_of[(long long)b * A + a];
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

__global__ __launch_bounds__(1024) void nms_sort_kernel(const float* __restrict__ pred, int nc, int A, int agnostic,
                                                         float max_wh, NmsWs w) {
  __shared__ unsigned long long sk[SORT_LDS_KEYS];
  __shared__ int s_nvalid;
  nms_sort_body(pred, nc, A, agnostic, max_wh, w, blockIdx.x, threadIdx.x, sk, s_nvalid);
}

// ---- 3. suppression mask: one wavefront per row i, one ballot per 64 columns j > i
// (two halves so that a caller can skip the second one for pairs that do not intersect)
__device__ __forceinline__ float iou_inter(const float4 a, const float4 b) {
  const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
  const float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
  const float iw = fmaxf(0.0f, __fsub_rn(xx2, xx1)), ih = fmaxf(0.0f, __fsub_rn(yy2, yy1));
  return __fmul_rn(iw, ih);
}
__device__ __forceinline__ bool iou_gt_of(float inter, float area_a, const float4 b, float thr) {
  const float area_b = __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y));
  const float uni = __fsub_rn(__fadd_rn(area_a, area_b), inter);
  return __fdiv_rn(inter, uni) > thr;
}

// rows row0, row0 + nrows_step, ... of image b, one wavefront per row
// (bo: the image's n class-offset boxes; mask / MW: its mask rows and their pitch in words -- global or LDS)
// (BoxPtr: const float4* in the workspace, or lds_f4_ptr -- a pointer TYPED as LDS: a pointer that may be either becomes a FLAT
// access, and the compiler then folds differently fed calls of this body into one of flat loads, three times slower)
typedef float nms_f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) nms_f32x4* lds_f4_ptr;
__device__ __forceinline__ float4 ld_box(const float4* p, int i) { return p[i]; }
__device__ __forceinline__ float4 ld_box(lds_f4_ptr p, int i) {
  const nms_f32x4 v = p[i];
  return make_float4(v.x, v.y, v.z, v.w);
}
template <class BoxPtr>
__device__ __forceinline__ void nms_mask_body(float iou_thres, int n, BoxPtr bo, unsigned long long* mask, int MW,
                                              int row0, int nrows_step, int lane) {
  const int nwords = (n + 63) >> 6;
  for (int i = row0; i < n; i += nrows_step) {
    const float4 bi = ld_box(bo, i);
    const float area_i = __fmul_rn(__fsub_rn(bi.z, bi.x), __fsub_rn(bi.w, bi.y));
    unsigned long long* row = mask + (long long)i * MW;
    for (int wd = i >> 6; wd < nwords; ++wd) {
      const int j = wd * 64 + lane;
      // inter is +0 (or NaN) for a pair that does not intersect, and then inter / union cannot exceed a threshold in [0, 1]:
      // the division is evaluated only where some lane of the word has an intersection (most words of a real image have none
      // -- boxes of different classes never meet, max_wh apart); the bits are those of the plain test
      float4 bj = bi;
      float inter = 0.0f;
      if (j < n && j > i) {
        bj = ld_box(bo, j);
        inter = iou_inter(bi, bj);
      }
      const bool meets = inter > 0.0f;
      unsigned long long bits = 0ull;
      if (__ballot(meets)) bits = __ballot(meets && iou_gt_of(inter, area_i, bj, iou_thres));
      if (lane == 0) row[wd] = bits;
    }
  }
}

__global__ __launch_bounds__(256) void nms_mask_kernel(float iou_thres, NmsWs w) {
  const int b = blockIdx.y;
  nms_mask_body(iou_thres, w.nfinal[b], w.box_off + (long long)b * w.M, w.mask + (long long)b * w.M * w.MW, w.MW,
                blockIdx.x * 4 + (threadIdx.x >> 6), gridDim.x * 4, threadIdx.x & 63);
}

// ---- 4. greedy scan over the sorted candidates, 64 at a time, one wavefront per image
constexpr int SCAN_T = 8;  // words per lane -> up to 64*64*8 = 32768 candidates

// one kept detection -> out[b][slot] (ops.py:66-98 scale_boxes + ops.py:276-295 clip_boxes when scaled)
// the image's sorted candidates as the scan reads them: global workspace rows, or the one-launch kernel's LDS arrays
struct ScanSrc {
  const float4* box_raw;
  const float* conf;
  const int32_t* cls;
  int n;
};
__device__ __forceinline__ ScanSrc scan_src_of(const NmsWs& w, int b) {
  const long long o = (long long)b * w.M;
  return ScanSrc{w.box_raw + o, w.conf + o, w.cls + o, w.nfinal[b]};
}

__device__ __forceinline__ void scan_emit(const ScanSrc& src, int b, int i, int slot, int max_det, bool scaled, float padx,
                                          float pady, float gain, float w0, float h0, float* out) {
  float4 bx = src.box_raw[i];
  if (scaled) {
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
  op[4] = src.conf[i];
  op[5] = (float)src.cls[i];
}

// a value every lane of the wave holds alike, as a scalar
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

// (scaled, padx .. h0: the image's scale record -- scale_boxes' arguments; scaled = false: boxes stay in network coordinates)
__device__ __forceinline__ void nms_scan_body(int max_det, bool scaled, float padx, float pady, float gain, float w0, float h0,
                                              float* out, int32_t* out_count, const ScanSrc& src, int b, int lane,
                                              const unsigned long long* mask, int MW) {
  const int n = src.n;
  const int nwords = (n + 63) >> 6;
  unsigned long long rem[SCAN_T];
#pragma unroll
  for (int t = 0; t < SCAN_T; ++t) rem[t] = 0ull;
  int total = 0;
  if (nwords <= SCAN_T) {
    // Fast path (up to 512 candidates -- every realistic conf >= 0.05 case): the pending-suppression
    // words are wave-uniform scalars, each lane keeps its own mask row in registers (one batch of loads
    // per 64 candidates), the diagonal word is resolved with readlane, and the rows of the kept
    // candidates are OR-ed across the wave with shuffles instead of one dependent global load each.
    unsigned long long remw[SCAN_T];
#pragma unroll
    for (int t = 0; t < SCAN_T; ++t) remw[t] = 0ull;
    for (int blk = 0; blk < nwords && total < max_det; ++blk) {
      const int i = blk * 64 + lane;
      unsigned long long d[SCAN_T];
#pragma unroll
      for (int t = 0; t < SCAN_T; ++t) d[t] = (t >= blk && t < nwords && i < n) ? mask[(long long)i * MW + t] : 0ull;
      unsigned long long dcur = 0ull, cur = 0ull;
#pragma unroll
      for (int t = 0; t < SCAN_T; ++t)
        if (t == blk) {
          dcur = d[t];
          cur = remw[t];
        }
      const int nvalid = n - blk * 64;
      const unsigned long long vmask = nvalid >= 64 ? ~0ull : ((1ull << nvalid) - 1ull);
      // greedy pass over the block in score order.  Only candidates whose row has a bit inside the block (`hot`) can change
      // anything, and rows hold bits j > i only: visiting the hot survivors in order leaves exactly the sequential scan's
      // survivors (a candidate's fate is settled by the kept hot ones before it), in as many steps as there are overlapping
      // boxes instead of one per kept box
      unsigned long long alive = uniform64(~cur & vmask);
      unsigned long long todo = alive & __ballot(dcur != 0ull);
      while (todo) {
        const int bit = __builtin_ctzll(todo);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)dcur, bit);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(dcur >> 32), bit);
        const unsigned long long gone = (((unsigned long long)hi << 32) | lo);
        alive &= ~gone;
        todo &= ~gone & ~(1ull << bit);
      }
      // max_det (ops.py:265): the first `room` survivors; what the dropped ones would have suppressed lies behind them
      const int room = max_det - total;
      while (__popcll(alive) > room) alive &= ~(1ull << (63 - __clzll(alive)));
      const unsigned long long kept = alive;
      const bool mine = (kept >> lane) & 1ull;
#pragma unroll
      for (int t = 0; t < SCAN_T; ++t)
        if (t > blk && t < nwords) {
          unsigned lo = mine ? (unsigned)d[t] : 0u, hi = mine ? (unsigned)(d[t] >> 32) : 0u;
#pragma unroll
          for (int off = 32; off >= 1; off >>= 1) {
            lo |= __shfl_xor(lo, off);
            hi |= __shfl_xor(hi, off);
          }
          remw[t] |= ((unsigned long long)hi << 32) | lo;
        }
      if (mine) scan_emit(src, b, i, total + __popcll(kept & ((1ull << lane) - 1ull)), max_det, scaled, padx, pady,
                          gain, w0, h0, out);
      total += __popcll(kept);
    }
    if (lane == 0) out_count[b] = total;
    return;
  }
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
    const unsigned long long d = (i < n) ? mask[(long long)i * MW + blk] : 0ull;
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
      const unsigned long long* row = mask + (long long)(blk * 64 + bit) * MW;
#pragma unroll
      for (int t = 0; t < SCAN_T; ++t) {
        const int wd = lane + 64 * t;
        if (wd > blk && wd < nwords) rem[t] |= row[wd];
      }
    }
    // emit: lane `bit` writes its own detection at rank order
    if ((kept >> lane) & 1ull)
      scan_emit(src, b, i, total + __popcll(kept & ((1ull << lane) - 1ull)), max_det, scaled, padx, pady, gain, w0,
                h0, out);
    total += __popcll(kept);
  }
  if (lane == 0) out_count[b] = total;
}

// the same with the scale record read from memory here
__device__ __forceinline__ void nms_scan_body(int max_det, const float* __restrict__ scale, float* out, int32_t* out_count,
                                              const ScanSrc& src, int b, int lane, const unsigned long long* mask, int MW) {
  float padx = 0.f, pady = 0.f, gain = 1.f, w0 = 0.f, h0 = 0.f;
  if (scale) {
    padx = scale[b * 5 + 0];
    pady = scale[b * 5 + 1];
    gain = scale[b * 5 + 2];
    w0 = scale[b * 5 + 3];
    h0 = scale[b * 5 + 4];
  }
  nms_scan_body(max_det, scale != nullptr, padx, pady, gain, w0, h0, out, out_count, src, b, lane, mask, MW);
}

__global__ __launch_bounds__(64) void nms_scan_kernel(int max_det, const float* __restrict__ scale, float* out,
                                                       int32_t* out_count, NmsWs w) {
  nms_scan_body(max_det, scale, out, out_count, scan_src_of(w, blockIdx.x), blockIdx.x, threadIdx.x,
                w.mask + (long long)blockIdx.x * w.M * w.MW, w.MW);
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
  YL_HIP(zero_i32(w.count, (size_t)batch, s));
  hipLaunchKernelGGL(nms_candidates_kernel, dim3((anchors + 255) / 256, batch), dim3(256), 0, s, pred, nc, anchors,
                     conf_thres, classes, classes ? n_classes : 0, w);
  hipLaunchKernelGGL(nms_sort_kernel, dim3(batch), dim3(1024), 0, s, pred, nc, anchors, agnostic, max_wh, w);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(64, batch), dim3(256), 0, s, iou_thres, w);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(batch), dim3(64), 0, s, max_det, scale, out, out_count, w);
  YL_HIP(hipGetLastError());
  return 0;
}

/* multi-label variant (the validator's postprocess, engine/validator.py:281-291 -> ops.py:239-241) */
extern "C" int64_t yl_nms_ml_workspace_bytes(int batch, int nc, int anchors, int max_cand) {
  if (batch <= 0 || nc <= 0 || anchors <= 0 || max_cand <= 0 || (long long)nc * anchors > (1LL << 30)) return -1;
  return nms_layout(batch, anchors, max_cand, nullptr, nullptr, (long long)nc * anchors);
}

extern "C" int yl_nms_ml(const float* pred, int batch, int nc, int anchors, float conf_thres, float iou_thres,
                         int max_det, int agnostic, const int32_t* classes, int n_classes, int max_cand, float max_wh,
                         const float* scale, float* out, int32_t* out_count, void* work, void* stream) {
  if (!pred || !out || !out_count || !work) return fail(-1, "yl_nms_ml: null pointer");
  if (batch <= 0 || nc <= 0 || anchors <= 0 || max_det <= 0 || max_cand <= 0 || (long long)nc * anchors > (1LL << 30))
    return fail(-1, "yl_nms_ml: bad sizes");
  if (!(conf_thres >= 0.f && conf_thres <= 1.f) || !(iou_thres >= 0.f && iou_thres <= 1.f))
    return fail(-4, "yl_nms_ml: thresholds must lie in [0, 1]");  // ops.py:186-187 AssertionError
  NmsWs w;
  nms_layout(batch, anchors, max_cand, (char*)work, &w, (long long)nc * anchors);
  if (w.M > 64 * 64 * SCAN_T) return fail(-3, "yl_nms_ml: max_cand above 32768 is not supported");
  w.ml_nc = nc > 1 ? nc : 0;  // ops.py:208: multi_label &= nc > 1
  hipStream_t s = (hipStream_t)stream;
  YL_HIP(zero_i32(w.count, (size_t)batch, s));
  if (w.ml_nc)
    hipLaunchKernelGGL(nms_candidates_ml_kernel, dim3((anchors + 255) / 256, batch), dim3(256), 0, s, pred, nc, anchors,
                       conf_thres, classes, classes ? n_classes : 0, w);
  else
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
// clears the threshold are kept: key + class + (cx, cy, w, h).
namespace yl {
struct DecCandArgs {
  DecodeLevel lv[8];
  int n_levels, batch, nc, A;
  float conf_thres;
  float logit_lo;  // sparse path: anchors whose best LOGIT is below this cannot clear conf_thres
  const int32_t* classes;
  int n_classes;
};

struct DcAnchor {
  DecodeLevel L;
  int b, p, px, py, pix;  // level-local anchor index, its coordinates, guarded pixel index
};
// NLV: an upper bound of a.n_levels the caller knows (3: the select below touches three level records instead of eight)
template <int NLV = 8>
__device__ __forceinline__ DcAnchor dc_anchor(const DecCandArgs& a, int b, int an) {
  DcAnchor d;
  if constexpr (NLV <= 4) {
    // the record by field-wise selects between the NLV records (scalars): an indexed copy is a per-lane LOAD from the kernel
    // argument segment -- one more dependent memory round trip (~1 us in the one-launch kernel's chain)
    d.L = a.lv[0];
#pragma unroll
    for (int i = 1; i < NLV; ++i) {
      const bool in = i < a.n_levels && an >= a.lv[i].a0;
      d.L.xywh = in ? a.lv[i].xywh : d.L.xywh;
      d.L.box.base = in ? a.lv[i].box.base : d.L.box.base;
      d.L.box.plane = in ? a.lv[i].box.plane : d.L.box.plane;
      d.L.box.W1 = in ? a.lv[i].box.W1 : d.L.box.W1;
      d.L.box.img = in ? a.lv[i].box.img : d.L.box.img;
      d.L.box.g0 = in ? a.lv[i].box.g0 : d.L.box.g0;
      d.L.cls.base = in ? a.lv[i].cls.base : d.L.cls.base;
      d.L.cls.plane = in ? a.lv[i].cls.plane : d.L.cls.plane;
      d.L.cls.W1 = in ? a.lv[i].cls.W1 : d.L.cls.W1;
      d.L.cls.img = in ? a.lv[i].cls.img : d.L.cls.img;
      d.L.cls.g0 = in ? a.lv[i].cls.g0 : d.L.cls.g0;
      d.L.H = in ? a.lv[i].H : d.L.H;
      d.L.W = in ? a.lv[i].W : d.L.W;
      d.L.a0 = in ? a.lv[i].a0 : d.L.a0;
      d.L.stride = in ? a.lv[i].stride : d.L.stride;
    }
  } else {
    int l = 0;
#pragma unroll
    for (int i = 1; i < NLV; ++i)
      if (i < a.n_levels && an >= a.lv[i].a0) l = i;
    d.L = a.lv[l];  // (with eight records the field-by-field select was measured slower than this indexed copy: 42 against 30 us)
  }
  d.b = b;
  d.p = an - d.L.a0;
  d.py = d.p / d.L.W;
  d.px = d.p - d.py * d.L.W;
  d.pix = d.L.box.g0 + b * d.L.box.img + d.py * d.L.box.W1 + d.px;
  return d;
}

// the reference's candidate test (ops.py:203,243-248: score above conf_thres, class in `classes` when given)
__device__ __forceinline__ bool dc_keep(const DecCandArgs& a, float best, int bi) {
  bool keep = best > a.conf_thres;
  if (keep && a.classes) {
    bool ok = false;
    for (int i = 0; i < a.n_classes; ++i) ok |= (a.classes[i] == bi);
    keep = ok;
  }
  return keep;
}

// the candidate test and bookkeeping for one anchor.
// key_slot < 0: append with an atomic (dense kernel); else the key goes to that slot of keys[b][] and a
// rejected anchor leaves the sentinel ~0 there (sparse path: no atomics).
// (bx = the anchor's (cx, cy, w, h): dc_box of its four DFL distances, or the level's decoded box, DecodeLevel::xywh)
__device__ __forceinline__ float4 dc_box(const DcAnchor& d, float dl, float dt, float dr, float db) {
  return box_xywh((float)d.px + 0.5f, (float)d.py + 0.5f, dl, dt, dr, db, d.L.stride);
}
__device__ __forceinline__ void dc_emit(const DecCandArgs& a, const NmsWs& w, const DcAnchor& d, float best, int bi,
                                        float4 bx, int key_slot = -1) {
  if (!dc_keep(a, best, bi)) {
    if (key_slot >= 0) w.keys[(long long)d.b * w.KP + key_slot] = ~0ull;
    return;
  }
  const int an = d.L.a0 + d.p;
  const int slot = key_slot >= 0 ? key_slot : atomicAdd(&w.count[d.b], 1);
  w.keys[(long long)d.b * w.KP + slot] = ((unsigned long long)(~__float_as_uint(best)) << 32) | (unsigned)an;
  w.cls_of[(long long)d.b * a.A + an] = bi;
  w.cbox[(long long)d.b * a.A + an] = bx;
}

// Dense variant (low thresholds, e.g. validation at conf 0.001): every anchor is decoded, one thread each.
// one thread decodes anchor `an` of image b completely (18 independent 16-byte loads: one memory round trip) and applies the
// reference's candidate test; key_slot as in dc_emit
__device__ __forceinline__ void decode_anchor_thread(const DecCandArgs& a, const NmsWs& w, int b, int an, int key_slot) {
  const DcAnchor d = dc_anchor(a, b, an);
  // best class: strict '>' in increasing c keeps the first maximum (ops.py:243)
  float best = -1.f;
  int bi = 0x7fffffff;
  for (int c8 = 0; c8 * 8 < a.nc; ++c8) {
    float v[8];
    level_load8(d.L.cls, c8, d.pix, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float sc = sigmoid_score(v[j]);
      if (c8 * 8 + j < a.nc && sc > best) {
        best = sc;
        bi = c8 * 8 + j;
      }
    }
  }
  if (!(best > a.conf_thres)) {
    if (key_slot >= 0) w.keys[(long long)b * w.KP + key_slot] = ~0ull;  // sparse path: the slot holds a rejected anchor
    return;
  }
  if (d.L.xywh) {  // the producing conv's epilogue decoded the box from its fp32 accumulators
    dc_emit(a, w, d, best, bi, level_xywh(d.L, b, d.p), key_slot);
    return;
  }
  float dist[4];
#pragma unroll
  for (int side = 0; side < 4; ++side) {
    float v[16], lo[8], hi[8];
    level_load8(d.L.box, 2 * side, d.pix, lo);
    level_load8(d.L.box, 2 * side + 1, d.pix, hi);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = lo[j];
      v[8 + j] = hi[j];
    }
    dist[side] = dfl_expect(v);
  }
  dc_emit(a, w, d, best, bi, dc_box(d, dist[0], dist[1], dist[2], dist[3]), key_slot);
}

// The same decode by FOUR adjacent lanes per anchor (lane & 3 = box side): each lane runs the DFL of its side (16 of the 64
// exponentials) and looks at a quarter of the class planes.  Bit-identical to decode_anchor_thread:
//   * sigmoid_score is monotone in the logit, so the best score is sigmoid(max logit) and only classes whose logit lies within
//     0.01 of the maximum can tie it after rounding: the sigmoid is evaluated for those alone (one or two instead of nc),
//     and the first-maximum rule (ops.py:243) becomes "smallest class index among the equal best scores" across the lanes.
//     Should the window itself round to the best score (saturated scores), every class is evaluated as the reference does.
//   * dfl_expect is the same function on the same 16 values.
// `valid` lanes only; all four lanes of a group must be active or inactive together.
// what the four lanes of a quad know about their anchor after the evaluation (every lane holds the same values)
struct DcQuad {
  float best;
  float4 bx;  // (cx, cy, w, h)
  int bi;
};
// (stamps: diagnostic builds, the one-launch kernel's phase stamps 7-10; every outstanding load is waited for before a stamp)
#ifdef YL_STAMPS
#define DQ_STAMP(id)                                                   \
  do {                                                                 \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        \
    YL_STAMP(stamps, 16, (int)(threadIdx.x >> 6), (int)(threadIdx.x & 63u), id); \
  } while (0)
#else
#define DQ_STAMP(id) do { } while (0)
#endif
__device__ __forceinline__ DcQuad decode_anchor_quad_eval(const DecCandArgs& a, const DcAnchor& d, int side,
                                                          unsigned long long* stamps = nullptr) {
  const int nplanes = (a.nc + 7) >> 3;
  DQ_STAMP(7);
  // every load of the lane is requested up front -- its class planes (side, side + 4, ...: up to QP of them are kept in
  // registers for both passes) and its two DFL planes -- so that the quad pays ONE memory round trip instead of three (first
  // class pass, second class pass, box planes); the arithmetic is unchanged
  constexpr int QP = 4;
  float cv[QP][8], l8[8], h8[8];
#pragma unroll
  for (int q = 0; q < QP; ++q) {
    const int c8 = side + 4 * q;
    if (c8 < nplanes) level_load8(d.L.cls, c8, d.pix, cv[q]);
  }
  const bool dec = d.L.xywh != nullptr;  // the box arrives decoded (DecodeLevel::xywh): 16 bytes instead of the lane's two DFL planes
  float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
  if (dec) {
    bx = level_xywh(d.L, d.b, d.p);
  } else {
    level_load8(d.L.box, 2 * side, d.pix, l8);
    level_load8(d.L.box, 2 * side + 1, d.pix, h8);
  }
  DQ_STAMP(8);
  float m = -__builtin_huge_valf();
#pragma unroll
  for (int q = 0; q < QP; ++q) {
    const int c8 = side + 4 * q;
    if (c8 < nplanes) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (c8 * 8 + j < a.nc) m = fmaxf(m, cv[q][j]);
    }
  }
  for (int c8 = side + 4 * QP; c8 < nplanes; c8 += 4) {  // (more than 128 classes: the remaining planes, not cached)
    float v[8];
    level_load8(d.L.cls, c8, d.pix, v);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (c8 * 8 + j < a.nc) m = fmaxf(m, v[j]);
  }
  m = fmaxf(m, __shfl_xor(m, 1));
  m = fmaxf(m, __shfl_xor(m, 2));
  const float top = sigmoid_score(m);
  const float lo = m - 0.01f;
  const bool all = sigmoid_score(lo) == top;  // the window does not separate: evaluate every class
  float best = -1.f;
  int bi = 0x7fffffff;
  auto scan8 = [&](int c8, const float (&v)[8]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c8 * 8 + j;
      if (c < a.nc && (all || v[j] >= lo)) {
        const float sc = sigmoid_score(v[j]);
        if (sc > best) {  // increasing c within a lane: strict '>' keeps the first maximum
          best = sc;
          bi = c;
        }
      }
    }
  };
#pragma unroll
  for (int q = 0; q < QP; ++q) {
    const int c8 = side + 4 * q;
    if (c8 < nplanes) scan8(c8, cv[q]);
  }
  for (int c8 = side + 4 * QP; c8 < nplanes; c8 += 4) {
    float v[8];
    level_load8(d.L.cls, c8, d.pix, v);
    scan8(c8, v);
  }
#pragma unroll
  for (int off = 1; off <= 2; off <<= 1) {
    const float s2 = __shfl_xor(best, off);
    const int i2 = __shfl_xor(bi, off);
    if (s2 > best || (s2 == best && i2 < bi)) {
      best = s2;
      bi = i2;
    }
  }
  DQ_STAMP(9);
  float dist = 0.f;
  if (!dec && best > a.conf_thres) {  // (uniform over the four lanes)
    float v[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = l8[j];
      v[8 + j] = h8[j];
    }
    dist = dfl_expect(v);
  }
  const int l0 = (int)(threadIdx.x & 63u) & ~3;
  DcQuad q;
  q.best = best;
  q.bi = bi;
  const float dl = __shfl(dist, l0), dt = __shfl(dist, l0 + 1), dr = __shfl(dist, l0 + 2), db = __shfl(dist, l0 + 3);
  q.bx = dec ? bx : dc_box(d, dl, dt, dr, db);
  DQ_STAMP(10);
  return q;
}

template <int NLV = 8>
__device__ __forceinline__ void decode_anchor_quad(const DecCandArgs& a, const NmsWs& w, int b, int an, int key_slot, int side) {
  const DcAnchor d = dc_anchor<NLV>(a, b, an);
  const DcQuad q = decode_anchor_quad_eval(a, d, side);
  if (side != 0) return;
  if (!(q.best > a.conf_thres)) {
    if (key_slot >= 0) w.keys[(long long)b * w.KP + key_slot] = ~0ull;
    return;
  }
  dc_emit(a, w, d, q.best, q.bi, q.bx, key_slot);
}

__global__ __launch_bounds__(256) void decode_candidates_dense_kernel(DecCandArgs a, NmsWs w) {
  const int an = blockIdx.x * 256 + threadIdx.x;
  if (an >= a.A) return;
  decode_anchor_thread(a, w, blockIdx.y, an, -1);
}

// Sparse variant (the usual case: ~1-2 % of the anchors clear conf 0.25), two kernels.
// 1. pre-filter: only the class planes are streamed (one thread per anchor, nc8/8 sixteen-byte loads,
//    coalesced per plane); anchors whose best LOGIT could clear the threshold (safety margin, see the
//    host code) are appended to one of NSUB per-image lists.
__global__ __launch_bounds__(256) void decode_candidates_sparse_kernel(DecCandArgs a, NmsWs w, int nc8) {
  const int an = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (an >= a.A) return;
  const DcAnchor d = dc_anchor(a, b, an);
  float m = -__builtin_huge_valf();
  for (int c8 = 0; c8 < nc8 / 8; ++c8) {
    float v[8];
    level_load8(d.L.cls, c8, d.pix, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, v[j]);
  }
  if (m > a.logit_lo) {
    const int sub = b * NSUB + ((an >> 6) & (NSUB - 1));
    const int slot = atomicAdd(&w.pre_count[sub], 1);
    if (slot < w.sub_cap) w.pre_list[(long long)sub * w.sub_cap + slot] = an;
  }
}

// 2. exact decode of the survivors, one wavefront per anchor: the reference arithmetic -- nc sigmoids,
//    first-max class, DFL of the 64 box logits (exponentials in parallel, sums in the reference's
//    sequential order) -- so the results are identical to the dense kernel's.
constexpr int DX_BLOCKS = 4;  // workgroups per sub-list (4 waves each)
// entries j0, j0 + jstep, ... of sub-list `sub` of image b, one wavefront per anchor
__device__ __forceinline__ void decode_exact_body(const DecCandArgs& a, const NmsWs& w, int b, int sub, int j0, int jstep, int lane) {
  const int n = min(w.pre_count[b * NSUB + sub], w.sub_cap);
  for (int j = j0; j < n; j += jstep) {
    const int an = min(max(w.pre_list[(long long)(b * NSUB + sub) * w.sub_cap + j], 0), a.A - 1);
    const DcAnchor d = dc_anchor(a, b, an);
    // lane = box channel lane: plane lane >> 3, element lane & 7
    const bool dec = d.L.xywh != nullptr;  // (wave-uniform: one anchor per wavefront) the level's boxes arrive decoded
    const float v = dec ? 0.f : (float)d.L.box.base[(long long)(lane >> 3) * d.L.box.plane + (long long)d.pix * 8 + (lane & 7)];
    float mx = v;
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    const float e = expf(v - mx);
    // class scores: per-lane strict '>' in increasing c keeps the first maximum (ops.py:243)
    float best = -1.f;
    int bi = 0x7fffffff;
    for (int c = lane; c < a.nc; c += 64) {
      const float lg = (float)d.L.cls.base[(long long)(c >> 3) * d.L.cls.plane + (long long)d.pix * 8 + (c & 7)];
      const float sc = sigmoid_score(lg);
      if (sc > best) {
        best = sc;
        bi = c;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float s2 = __shfl_xor(best, off);
      const int i2 = __shfl_xor(bi, off);
      if (s2 > best || (s2 == best && i2 < bi)) {
        best = s2;
        bi = i2;
      }
    }
    float sum = 0.f, ws = 0.f;  // every lane sums the side it belongs to
#pragma unroll
    for (int j2 = 0; j2 < 16; ++j2) {
      const float ej = __shfl(e, (lane & 48) + j2);
      sum += ej;
      ws += ej * (float)j2;
    }
    const float dist = ws / sum;
    const float dl = __shfl(dist, 0), dt = __shfl(dist, 16), dr = __shfl(dist, 32), db = __shfl(dist, 48);
    if (lane == 0) dc_emit(a, w, d, best, bi, dec ? level_xywh(d.L, b, d.p) : dc_box(d, dl, dt, dr, db), sub * w.sub_cap + j);
  }
}

__global__ __launch_bounds__(256) void decode_exact_kernel(DecCandArgs a, NmsWs w) {
  decode_exact_body(a, w, blockIdx.z, blockIdx.y, blockIdx.x * 4 + (threadIdx.x >> 6), DX_BLOCKS * 4, threadIdx.x & 63);
}

// ---- the hot path's post-processing in ONE launch (decode_nms_fused_kernel, below): one workgroup (16 wavefronts) per image
// runs the exact decode of the pre-filtered anchors, the sort + gather, the suppression and the greedy scan + scale_boxes one after
// the other -- the device functions of the four kernels above, so the results are bit-identical -- with workgroup barriers where the
// four-launch form has kernel boundaries.  Three regimes by the image's candidate count:
//   <= 512   everything in LDS (keys, classes, boxes, sorted rows, mask): nothing but the pre-filter's lists and the level maps is
//            read, nothing but the detections written.  YOLO11n batch 32, ~126 candidates per image: 32 us kernel span (round 3's
//            form, every hand-over through the NMS workspace: 41.7 us; the four launches 56 us in sequence).  Phases by in-kernel
//            stamps (tools/nms_stamps.py): decode 9-11 (three dependent round trips: counts -> list -> level maps), sort 3.3,
//            mask 6.5, scan 3.2.
//   <= 1536  intermediates through the workspace, boxes in LDS, full mask + scan.
//   above    the greedy loop over the kept candidates' rows only (nms_greedy_body): ~6 000 candidates, 300 kept per image: 3.15 ms
//            per batch of 32 (the full mask on one CU: 13.1 ms; the four launches: 3.66 ms).
// Greedy suppression WITHOUT the mask: one workgroup of 16 waves walks the image's sorted candidates; a candidate that is still
// alive is kept, its row -- IoU against every later candidate, 64 per wave and step -- is evaluated right then and OR-ed into
// the `gone` bitmap (LDS, one word per 64 candidates, each word owned by one wave), the next alive candidate is looked up.
// Only the rows of KEPT candidates are ever needed (at most max_det of them): max_det x n pairs instead of n^2 / 2 -- for the
// dense images the one-launch kernel's general path exists for (thousands of candidates, 300 kept) an order of magnitude less
// work than mask + scan, and it stops at max_det.  Same IoU test, same order: the kept set and the output are those of
// nms_mask_body + nms_scan_body.  Two barriers per kept candidate.
template <class BoxPtr>
__device__ __forceinline__ void nms_greedy_body(float iou_thres, int max_det, const float* __restrict__ scale, float* out,
                                                int32_t* out_count, const ScanSrc& src, BoxPtr bo, unsigned long long* gone,
                                                int& s_cur, int b, int tid) {
  const int n = src.n, nwords = (n + 63) >> 6, lane = tid & 63, wave = tid >> 6;
  float padx = 0.f, pady = 0.f, gain = 1.f, w0 = 0.f, h0 = 0.f;
  if (scale) {
    padx = scale[b * 5 + 0];
    pady = scale[b * 5 + 1];
    gain = scale[b * 5 + 2];
    w0 = scale[b * 5 + 3];
    h0 = scale[b * 5 + 4];
  }
  for (int wd = tid; wd < nwords; wd += 1024) gone[wd] = 0ull;
  __syncthreads();
  int total = 0, pos = 0;
  while (true) {
    if (wave == 0) {  // the next candidate nobody suppressed, from `pos` on (every lane computes the same)
      int i = -1;
      for (int wd = pos >> 6; wd < nwords && i < 0; ++wd) {
        const int nv = n - wd * 64;
        unsigned long long alive = ~gone[wd] & (nv >= 64 ? ~0ull : ((1ull << nv) - 1ull));
        if (wd == (pos >> 6)) alive &= ~0ull << (pos & 63);
        if (alive) i = wd * 64 + __builtin_ctzll(alive);
      }
      if (lane == 0) s_cur = i;
    }
    __syncthreads();
    const int i = s_cur;
    if (i < 0) break;
    if (tid == 0) scan_emit(src, b, i, total, max_det, scale != nullptr, padx, pady, gain, w0, h0, out);
    ++total;
    if (total >= max_det) break;
    const float4 bi = ld_box(bo, i);
    const float area_i = __fmul_rn(__fsub_rn(bi.z, bi.x), __fsub_rn(bi.w, bi.y));
    for (int wd = (i >> 6) + wave; wd < nwords; wd += 16) {
      const int j = wd * 64 + lane;
      float4 bj = bi;
      float inter = 0.0f;
      if (j < n && j > i) {
        bj = ld_box(bo, j);
        inter = iou_inter(bi, bj);
      }
      const bool meets = inter > 0.0f;
      unsigned long long bits = 0ull;
      if (__ballot(meets)) bits = __ballot(meets && iou_gt_of(inter, area_i, bj, iou_thres));
      if (lane == 0 && bits) gone[wd] |= bits;
    }
    pos = i + 1;
    __syncthreads();
  }
  if (tid == 0) out_count[b] = total;
}

#define NMS_STAMP(id) YL_STAMP(stamps, 16, wave, lane, id)

// Up to this many pre-filtered anchors per image the one-launch kernel keeps every intermediate of the image in LDS.
constexpr int FUSED_LDS_MAX = 512;

// NLV: see dc_anchor
template <int NLV>
__global__ __launch_bounds__(1024) void decode_nms_fused_kernel(DecCandArgs a, NmsWs w, float iou_thres, int max_det,
                                                                 int agnostic, float max_wh, const float* __restrict__ scale,
                                                                 float* out, int32_t* out_count, int stop_after, int lds_path,
                                                                 unsigned long long* stamps) {
  __shared__ unsigned long long sk[SORT_LDS_KEYS];
  __shared__ int s_nvalid;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  NMS_STAMP(0);
  int pre[NSUB + 1];  // the NSUB sub-lists as one list: all candidates of the image in flight at once
  pre[0] = 0;
#pragma unroll
  for (int q = 0; q < NSUB; ++q) pre[q + 1] = pre[q] + min(max(w.pre_count[b * NSUB + q], 0), w.sub_cap);
  const int side = tid & 3;
  NMS_STAMP(1);
  if (lds_path && pre[NSUB] <= FUSED_LDS_MAX) {
    // ---- the usual case (conf 0.25: ~1-2 % of the anchors): nothing but the pre-filter's lists and the level maps is read
    // from memory and nothing but the detections written -- keys, classes, boxes, the sorted rows and the mask live in LDS
    // (the general path below hands them from phase to phase through the NMS workspace: eight dependent round trips to L2
    // per image on top of the three every path has -- counts, list, maps).  The same device functions in the same order
    // on the same values: bit-identical to the general path and to the four-launch form.
    //   [0, 14 KiB)  candidates in list order: key u64[512] | class i32[512] | (cx, cy, w, h) float4[512]
    //   [0, 32 KiB)  ... overlaid, once sorted, by the suppression mask u64[512][8]
    //   [32, 52 KiB) sorted rows: box_raw float4[512] | box_off float4[512] | conf f32[512] | class i32[512]
    const int n = pre[NSUB];
    unsigned long long* ckey = sk;
    int32_t* ccls = reinterpret_cast<int32_t*>(sk + FUSED_LDS_MAX);
    float4* cbox = reinterpret_cast<float4*>(sk + FUSED_LDS_MAX + FUSED_LDS_MAX / 2);
    unsigned long long* lmask = sk;
    float4* s_raw = reinterpret_cast<float4*>(sk + 4096);
    float4* s_off = s_raw + FUSED_LDS_MAX;
    float* s_conf = reinterpret_cast<float*>(s_off + FUSED_LDS_MAX);
    int32_t* s_cls = reinterpret_cast<int32_t*>(s_conf + FUSED_LDS_MAX);
    static_assert(SORT_LDS_KEYS >= 4096 + (2 * 16 + 4 + 4) * FUSED_LDS_MAX / 8, "sorted rows behind the mask");
    static_assert((8 + 4 + 16) * FUSED_LDS_MAX <= 4096 * 8 && FUSED_LDS_MAX * 8 * 8 <= 4096 * 8,
                  "candidates / mask below the sorted rows");
    float padx = 0.f, pady = 0.f, gain = 1.f, w0 = 0.f, h0 = 0.f;  // (requested now, used by the scan)
    if (scale && wave == 0) {
      padx = scale[b * 5 + 0];
      pady = scale[b * 5 + 1];
      gain = scale[b * 5 + 2];
      w0 = scale[b * 5 + 3];
      h0 = scale[b * 5 + 4];
    }
    for (int i0 = 0; i0 < n; i0 += 256) {
      const int i = i0 + (tid >> 2);
      if (i < n) {
        int sub = 0;
#pragma unroll
        for (int t = 1; t < NSUB; ++t) sub += (i >= pre[t]);
        int j = i;
#pragma unroll
        for (int t = 1; t < NSUB; ++t) j = (t == sub) ? i - pre[t] : j;
        const int an = min(max(w.pre_list[(long long)(b * NSUB + sub) * w.sub_cap + j], 0), a.A - 1);
        DQ_STAMP(6);
        const DcAnchor d = dc_anchor<NLV>(a, b, an);
        const DcQuad q = decode_anchor_quad_eval(a, d, side, stamps);
        if (side == 0) {
          const bool keep = dc_keep(a, q.best, q.bi);
          ckey[i] = keep ? (((unsigned long long)(~__float_as_uint(q.best)) << 32) | (unsigned)an) : ~0ull;
          ccls[i] = q.bi;
          cbox[i] = q.bx;
        }
      }
    }
    NMS_STAMP(2);
    if (stop_after == 1) return;
    __syncthreads();
    // sort by rank (rank_sort's order: every real key is unique) and build the sorted rows straight from the candidate's own
    // slot -- nms_sort_body's gather arithmetic (ops.py:254-258, 386)
    bool valid = false;
    if (tid < n) {
      const unsigned long long x = ckey[tid];
      valid = x != ~0ull;
      if (valid) {
        int r = 0;
        for (int j = 0; j < n; ++j) r += ckey[j] < x ? 1 : 0;
        if (r < w.M) {  // keep the top max_cand by score
          const float score = __uint_as_float(~(unsigned)(x >> 32));
          const int cls = ccls[tid];
          const float4 t = cbox[tid];
          const float hw = __fdiv_rn(t.z, 2.0f), hh = __fdiv_rn(t.w, 2.0f);
          const float x1 = __fsub_rn(t.x, hw), y1 = __fsub_rn(t.y, hh), x2 = __fadd_rn(t.x, hw), y2 = __fadd_rn(t.y, hh);
          const float c = __fmul_rn((float)cls, agnostic ? 0.0f : max_wh);
          s_raw[r] = make_float4(x1, y1, x2, y2);
          s_off[r] = make_float4(__fadd_rn(x1, c), __fadd_rn(y1, c), __fadd_rn(x2, c), __fadd_rn(y2, c));
          s_conf[r] = score;
          s_cls[r] = cls;
        }
      }
    }
    const int nvalid = __syncthreads_count(valid);  // (also the barrier between the candidates and the mask that overlays them)
    const int nf = nvalid < w.M ? nvalid : w.M;
    NMS_STAMP(3);
    if (stop_after == 2) return;
    nms_mask_body(iou_thres, nf, (lds_f4_ptr)(void*)s_off, lmask, 8, wave, 16, lane);
    NMS_STAMP(4);
    if (stop_after == 3) return;
    __syncthreads();
    if (wave == 0) {
      const ScanSrc src{s_raw, s_conf, s_cls, nf};
      nms_scan_body(max_det, scale != nullptr, padx, pady, gain, w0, h0, out, out_count, src, b, lane, lmask, 8);
    }
    NMS_STAMP(5);
    return;
  }
  // ---- general path: exact decode of the pre-filtered anchors by FOUR LANES each (decode_anchor_quad:
  // decode_candidates_dense_kernel's arithmetic, which decode_exact_kernel reproduces bit for bit).  Measured alternatives: a
  // wavefront per anchor, as the four-launch form has it, leaves each of the 16 waves a chain of ~13 dependent memory round
  // trips (115 us for the launch); one thread per anchor evaluates nc precise sigmoids + 64 exponentials serially (30 us for
  // this phase).  Rounds of 256 candidates; a whole quad is active or idle together (shuffles inside).
  for (int i0 = 0; i0 < pre[NSUB]; i0 += 256) {
    const int i = i0 + (tid >> 2);
    if (i < pre[NSUB]) {
      int sub = 0;
#pragma unroll
      for (int t = 1; t < NSUB; ++t) sub += (i >= pre[t]);
      int j = i;
#pragma unroll
      for (int t = 1; t < NSUB; ++t) j = (t == sub) ? i - pre[t] : j;
      const int an = min(max(w.pre_list[(long long)(b * NSUB + sub) * w.sub_cap + j], 0), a.A - 1);
      decode_anchor_quad<NLV>(a, w, b, an, sub * w.sub_cap + j, side);
    }
  }
  NMS_STAMP(2);
  if (stop_after == 1) return;  // (development: timing ablation, YL_NMS_DBG)
  __syncthreads();
  nms_sort_body(nullptr, a.nc, a.A, agnostic, max_wh, w, b, tid, sk, s_nvalid);
  NMS_STAMP(3);
  if (stop_after == 2) return;
  __syncthreads();
  // mask + scan would evaluate n^2 / 2 pairs on this one CU (10 ms for 6 000 candidates); only the kept candidates' rows are
  // needed: nms_greedy_body.  The `gone` bitmap (one bit per candidate) and, up to 3 840 of them, the boxes sit in LDS (the sort's
  // key array is free now); more boxes are read from the workspace.
  const ScanSrc src = scan_src_of(w, b);
  unsigned long long* gone = sk;                            // [512] words = 32 768 candidates
  static_assert(64 * 64 * SCAN_T <= 512 * 64, "one bit per candidate");
  constexpr int LDS_BOXES = (SORT_LDS_KEYS - 512) / 2;
  // (up to ~1 500 candidates the full mask is still the cheaper way: n^2 / 2 pairs spread over 16 waves without a barrier
  // against two barriers per kept candidate -- 600 candidates, 300 kept: ~80 us against ~350 us)
  constexpr int MASK_MAX = 1536;
  if (src.n <= MASK_MAX) {
    float4* lbox = reinterpret_cast<float4*>(sk + 512);
    for (int i = tid; i < src.n; i += 1024) lbox[i] = w.box_off[(long long)b * w.M + i];
    __syncthreads();
    unsigned long long* mk = w.mask + (long long)b * w.M * w.MW;
    nms_mask_body(iou_thres, src.n, (lds_f4_ptr)(void*)lbox, mk, w.MW, wave, 16, lane);
    __syncthreads();
    if (wave == 0) nms_scan_body(max_det, scale, out, out_count, src, b, lane, mk, w.MW);
  } else if (src.n <= LDS_BOXES) {
    float4* lbox = reinterpret_cast<float4*>(sk + 512);
    for (int i = tid; i < src.n; i += 1024) lbox[i] = w.box_off[(long long)b * w.M + i];
    nms_greedy_body(iou_thres, max_det, scale, out, out_count, src, (lds_f4_ptr)(void*)lbox, gone, s_nvalid, b, tid);
  } else {
    nms_greedy_body(iou_thres, max_det, scale, out, out_count, src, (const float4*)(w.box_off + (long long)b * w.M), gone, s_nvalid, b,
                    tid);
  }
  NMS_STAMP(4);
  NMS_STAMP(5);
}
}  // namespace yl

// ev != nullptr (yl_engine_profile_detect): ev[0..5] are recorded on the stream around the five stages
// {zero counters + class pre-filter, exact decode, sort, suppression mask, scan + output}.
int yl::decode_nms_stages(const void* workspace, const yl_level* levels, int n_levels, int batch, int nc,
                          float conf_thres, float iou_thres, int max_det, int agnostic, const int32_t* classes,
                          int n_classes, int max_cand, float max_wh, const float* scale, float* out,
                          int32_t* out_count, void* work, void* stream, hipEvent_t* ev, int prefiltered) {
  if (!workspace || !levels || !out || !out_count || !work) return fail(-1, "yl_decode_nms: null pointer");
  if (batch <= 0 || nc <= 0 || n_levels <= 0 || n_levels > 8 || max_det <= 0 || max_cand <= 0)
    return fail(-1, "yl_decode_nms: bad sizes");
  if (!(conf_thres >= 0.f && conf_thres <= 1.f) || !(iou_thres >= 0.f && iou_thres <= 1.f))
    return fail(-4, "yl_decode_nms: thresholds must lie in [0, 1]");
  DecCandArgs a;
  int A = 0;
  for (int i = 0; i < n_levels; ++i) {
    if (levels[i].nc8 < nc || levels[i].nc8 % 8) return fail(-1, "yl_decode_nms: level nc8 must be nc rounded up to 8");
    a.lv[i] = decode_level(workspace, levels[i], batch, A);
    A += levels[i].H * levels[i].W;
  }
  for (int i = n_levels; i < 8; ++i) {
    a.lv[i] = a.lv[0];
    a.lv[i].a0 = 1 << 30;
  }
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
  // Sparse path unless the threshold keeps a large share of the anchors (validation at conf 0.001).
  // logit(conf) - 0.05: the exact test `1 / (1 + expf(-x)) > conf` cannot pass below it.
  const bool sparse = conf_thres >= 0.05f;
  a.logit_lo = conf_thres >= 1.f ? __builtin_huge_valf()
                                 : (float)(std::log((double)conf_thres / (1.0 - (double)conf_thres)) - 0.05);
  hipStream_t s = (hipStream_t)stream;
  if (prefiltered && !sparse) return fail(-1, "yl_decode_nms: the fused pre-filter exists for the sparse decode path only");
  if (ev) (void)hipEventRecord(ev[0], s);
  if (!prefiltered) YL_HIP(zero_i32(w.count, (size_t)batch * (1 + NSUB), s));  // count + pre_count
  const dim3 grid((unsigned)((A + 255) / 256), batch);
  if (sparse) {
    w.fused = 1;
    if (!prefiltered) hipLaunchKernelGGL(decode_candidates_sparse_kernel, grid, dim3(256), 0, s, a, w, levels[0].nc8);
    if (ev) (void)hipEventRecord(ev[1], s);
    // (one workgroup per image: with few images in the batch the four launches, whose grids spread over the anchors and rows,
    // are quicker -- YOLO11x 1280x1280 batch 8, ~1000 candidates per image: 214 us against 320 us)
    static const int one_launch_env = env_int("YL_NMS_FUSED", -1);
    const bool one_launch = one_launch_env >= 0 ? one_launch_env != 0 : batch >= 16;
    if (one_launch) {  // exact decode + sort + mask + scan in one launch, one workgroup per image
      static const int stop_after = tune_int("YL_NMS_DBG", 0);
      static const int lds_path = env_int("YL_NMS_LDS", 1);  // (0: the general path for every image)
      auto kern = n_levels <= 3 ? decode_nms_fused_kernel<3> : decode_nms_fused_kernel<8>;
      hipLaunchKernelGGL(kern, dim3(batch), dim3(1024), 0, s, a, w, iou_thres, max_det, agnostic, max_wh, scale, out, out_count,
                         stop_after, lds_path, g_stamp_buf);
      if (ev)
        for (int q = 2; q <= 5; ++q) (void)hipEventRecord(ev[q], s);  // stages 3-5 are part of that launch: zero time
      YL_HIP(hipGetLastError());
      return 0;
    }
    hipLaunchKernelGGL(decode_exact_kernel, dim3(DX_BLOCKS, NSUB, batch), dim3(256), 0, s, a, w);
  } else {
    hipLaunchKernelGGL(decode_candidates_dense_kernel, grid, dim3(256), 0, s, a, w);
    if (ev) (void)hipEventRecord(ev[1], s);
  }
  if (ev) (void)hipEventRecord(ev[2], s);
  hipLaunchKernelGGL(nms_sort_kernel, dim3(batch), dim3(1024), 0, s, (const float*)nullptr, nc, A, agnostic, max_wh, w);
  if (ev) (void)hipEventRecord(ev[3], s);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(64, batch), dim3(256), 0, s, iou_thres, w);
  if (ev) (void)hipEventRecord(ev[4], s);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(batch), dim3(64), 0, s, max_det, scale, out, out_count, w);
  if (ev) (void)hipEventRecord(ev[5], s);
  YL_HIP(hipGetLastError());
  return 0;
}

int yl::prefilter_sink_prepare(const yl_level* levels, int n_levels, int batch, int max_cand, float conf_thres, void* work,
                               void* stream, PrefilterSink* sink, int* active, bool defer_zero) {
  *active = 0;
  if (!levels || !work || !sink || n_levels <= 0 || n_levels > 8 || batch <= 0 || max_cand <= 0) return fail(-1, "prefilter sink: bad arguments");
  if (!(conf_thres >= 0.05f) || !(conf_thres <= 1.f)) return 0;  // dense decode path (or an invalid threshold: yl_decode_nms reports it)
  int A = 0;
  for (int i = 0; i < n_levels; ++i) A += levels[i].H * levels[i].W;
  NmsWs w;
  nms_layout(batch, A, max_cand, (char*)work, &w);
  sink->zero_ptr = nullptr;
  sink->zero_n = 0;
  if (defer_zero) {
    sink->zero_ptr = w.count;  // count + pre_count, adjacent
    sink->zero_n = batch * (1 + NSUB);
  } else {
    YL_HIP(zero_i32(w.count, (size_t)batch * (1 + NSUB), (hipStream_t)stream));
  }
  sink->pre_count = w.pre_count;
  sink->pre_list = w.pre_list;
  sink->sub_cap = w.sub_cap;
  // the same bound as decode_nms_stages: logit(conf) - 0.05
  sink->logit_lo = conf_thres >= 1.f ? __builtin_huge_valf() : (float)(std::log((double)conf_thres / (1.0 - (double)conf_thres)) - 0.05);
  *active = 1;
  return 0;
}

extern "C" int yl_decode_nms(const void* workspace, const yl_level* levels, int n_levels, int batch, int nc,
                             float conf_thres, float iou_thres, int max_det, int agnostic, const int32_t* classes,
                             int n_classes, int max_cand, float max_wh, const float* scale, float* out,
                             int32_t* out_count, void* work, void* stream) {
  return yl::decode_nms_stages(workspace, levels, n_levels, batch, nc, conf_thres, iou_thres, max_det, agnostic, classes,
                               n_classes, max_cand, max_wh, scale, out, out_count, work, stream, nullptr);
}
