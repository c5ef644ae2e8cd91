// Engine: a fixed (batch, H, W) program of ops over one workspace arena, all enqueued on a single
// HIP stream (optionally replayed from a captured hipGraph).  Replaces the Python layer loop of the
// reference (nn/tasks.py:118-145 BaseModel._predict_once, ~320 module calls per forward) and the
// device plumbing of nn/autobackend.py:34-151.
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <thread>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <tuple>

#include "yl_internal.h"

namespace yl {

static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
int fail(int code, const std::string& s) {
  g_err = s;
  return code;
}

unsigned long long* g_stamp_buf = nullptr;
int st_nt_policy(long long out_bytes) {
  // (read at every engine build, not cached: tests and tools switch them)
  const int mode = tune_int("YL_ST_NT", 1);  // (-1: by size, YL_ST_NT_BYTES)
  // bit 1: the e4m3 stores too (YL_ST_NT_F8=1; measured on config 5: 1 136 -> 1 131 images/s, off)
  const int f8 = (tune_int("YL_ST_NT_F8", 0)) ? 2 : 0;
  if (mode >= 0) return mode ? (1 | f8) : 0;
  const long long thr = (long long)tune_int("YL_ST_NT_BYTES", 48 << 20);
  return out_bytes >= thr ? (1 | f8) : 0;
}

int allow_big_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  YL_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  if (done.count({dev, kernel})) return 0;
  YL_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.insert({dev, kernel});
  return 0;
}

int device_cu_count() {
  static std::mutex mu;
  static std::map<int, int> cus;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cus.find(dev);
  if (it != cus.end()) return it->second;
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  cus[dev] = n;
  return n;
}

struct OpRec {
  yl_op op;
  ConvConfig cfg{};
  __bf16* d_wpk = nullptr;
  __bf16* d_wpk2 = nullptr;  // fused follow-up 1x1
  float* d_bias2 = nullptr;
  float* d_w = nullptr;     // dwconv weights [9][C]
  uint16_t* d_dwpk = nullptr;  // fused depthwise + pointwise: the depthwise filter as MFMA A fragments (dw_pack_mfma)
  float* d_bias = nullptr;
  float* d_w0 = nullptr;    // fused stem in front of a conv (stem2.hip): filters [C0][27], bias [C0]
  float* d_b0 = nullptr;
  __bf16* d_wpk3 = nullptr;  // YL_OP_C3K2S: the block's 1x1 conv (d_wpk / d_wpk2 = the Bottleneck's two 3x3 convs)
  float* d_bias3 = nullptr;
  float* d_alpha = nullptr;  // fp8 conv: in_scale * filter scale per output channel
  __bf16* d_xw[6] = {};      // YL_OP_C3K: the six packed filters (cv1|cv2, m.0.cv1, m.0.cv2, m.1.cv1, m.1.cv2, cv3) and biases
  float* d_xb[6] = {};       // YL_OP_PSA: d_xw[0] = qkv, d_xw[1] = proj | ffn.0 | ffn.1 (160 KiB); d_xb = qkv, pe, proj, ffn.0, ffn.1 biases; d_w = pe taps
  bool f8 = false;
  int cls_level = -1;  // fused depthwise + pointwise + 1x1 that writes the class logits of this Detect level (else -1)
  int group_n = 0;  // group launch (yl_op.group): 2 or 3 on the first member, -1 on the others (run with it), 0 = on its own
  bool band = false;  // (first member of a group) the group runs on conv3_band_kernel: deep-K 3x3 convs, 64 output channels
  bool band1 = false; // a conv on its own on conv3_band1_kernel (the wider models' first box convs: 128 / 256 / 512 -> 64)
  bool csg = false;   // ... on conv3_csg_kernel: the levels' cv2[i][1] + cv2[i][2] + box decode, filter halves in registers
  double flops = 0, bytes = 0;
  const char* name = "";
};

}  // namespace yl

using namespace yl;

struct yl_engine {
  int batch = 0, in_h = 0, in_w = 0, nc = 0, A = 0;
  bool cls_tails = false;  // every Detect level's class logits come out of a fused dwpw + 1x1 launch: the NMS pre-filter can ride in its epilogue
  int64_t ws_bytes = 0;
  yl_view input_view{};
  std::vector<yl_level> levels;
  std::vector<OpRec> ops;
  std::vector<void*> allocs;
  struct GraphKey {
    const void* in;
    int dt;
    void* ws;
    float* y;
    bool operator<(const GraphKey& o) const {
      return std::tie(in, dt, ws, y) < std::tie(o.in, o.dt, o.ws, o.y);
    }
  };
  // Instantiated graphs, least recently used first out.  An executable may still be running on some slot's stream
  // when the cache is full, so eviction waits for the device and then destroys exactly ONE entry (the stalest);
  // 64 entries = 8-10 slots x two lanes x a few settings before anything is ever evicted.
  template <typename K>
  struct GraphCache {
    struct Ent {
      hipGraphExec_t exec;
      uint64_t used;
      bool pinned;  // handed out by yl_engine_detect_graph_prepare: the caller holds the handle until yl_engine_destroy
    };
    std::map<K, Ent> m;
    uint64_t tick = 0;
    size_t cap = 64;
    hipGraphExec_t find(const K& k, bool pin = false) {
      auto it = m.find(k);
      if (it == m.end()) return nullptr;
      it->second.used = ++tick;
      if (pin) it->second.pinned = true;
      return it->second.exec;
    }
    // Pinned entries are never evicted (a caller may launch them at any time without going through find()); when every
    // entry is pinned the cache simply grows.
    void insert(const K& k, hipGraphExec_t ge, bool pin = false) {
      if (m.size() >= cap) {
        auto lru = m.end();
        for (auto it = m.begin(); it != m.end(); ++it)
          if (!it->second.pinned && (lru == m.end() || it->second.used < lru->second.used)) lru = it;
        if (lru != m.end()) {
          (void)hipDeviceSynchronize();
          (void)hipGraphExecDestroy(lru->second.exec);
          m.erase(lru);
        }
      }
      m[k] = Ent{ge, ++tick, pin};
    }
    void clear() {
      for (auto& kv : m) (void)hipGraphExecDestroy(kv.second.exec);
      m.clear();
    }
  };
  GraphCache<GraphKey> graphs;
  // whole hot path (network + fused decode + NMS) as one graph, keyed by every baked-in argument
  typedef std::tuple<const void*, int, void*, float, float, int, int, const void*, int, int, float, const void*, void*,
                     void*, void*>
      DetectKey;
  GraphCache<DetectKey> dgraphs;
  // The same graphs keyed WITHOUT the input address (yl_engine_detect_graph_rebind): the kernel nodes that read the input
  // get the new address written into their parameters before a launch, so a caller that hands over a fresh tensor per
  // batch replays one captured graph per slot without staging copies.  The source graph is kept: its node handles address
  // the executable's nodes.
  struct Rebindable {
    hipGraph_t g = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<hipGraphNode_t> nodes;  // kernel nodes whose first parameter starts with the input address
    const void* cur = nullptr;
    uint64_t used = 0;
  };
  // Bounded like `dgraphs` (a service that varies conf / iou / classes / max_det would otherwise grow without limit): at
  // `rcap` entries the least recently launched one is destroyed (device synchronised first), graph and executable.
  std::map<DetectKey, Rebindable> rgraphs;
  uint64_t rtick = 0;
  size_t rcap = 64;
  // Per workspace (= per batch in flight): a private capture stream, one non-blocking stream per side
  // lane (yl_op.lane > 0) and a pool of fork/join events.  Batches that are in flight on different
  // caller streams use different workspaces; giving each its own lane streams keeps their graphs from
  // sharing (and serialising on) one hardware queue.
  struct StreamSet {
    hipStream_t cap = nullptr;
    std::vector<hipStream_t> lanes;
    std::vector<hipEvent_t> events;
  };
  std::map<void*, StreamSet> sets;
  int max_lane = 0;
  // Side lanes as SEPARATE graphs (lane_graphs; the one-batch profile of the program): this HIP runtime replays the parallel
  // branches of one captured graph one after the other (yl_graph_branch_probe), so a program with side lanes is cut into
  // SEGMENTS -- maximal runs of ops of one lane -- each captured as a graph of its own: main segments are launched on the
  // caller's stream in program order, a side segment on its lane's stream behind an event recorded after the main segment in
  // front of it, every lane is joined before the decode + NMS graph.  Ops of one lane keep their program order.
  bool lane_graphs = false;
  struct Segment {
    int lane;
    size_t begin, end;  // ops [begin, end)
  };
  std::vector<Segment> segments;
  struct LanePlan {
    std::vector<hipGraphExec_t> seg;  // one executable per segment
    hipGraphExec_t post = nullptr;    // decode + NMS
    hipGraph_t g0 = nullptr;          // source graph of segment 0: its node handles address seg[0]'s nodes (input rebinding)
    std::vector<hipGraphNode_t> in_nodes;
    const void* cur = nullptr;
    uint64_t used = 0;
  };
  static void destroy_plan(LanePlan& p) {
    for (hipGraphExec_t x : p.seg)
      if (x) (void)hipGraphExecDestroy(x);
    if (p.post) (void)hipGraphExecDestroy(p.post);
    if (p.g0) (void)hipGraphDestroy(p.g0);
    p = LanePlan{};
  }
  std::map<DetectKey, LanePlan> lplans;  // keyed without the input address (always rebindable), bounded like rgraphs
  std::vector<void*> zeroed;  // workspaces whose guard pixels this engine has already cleared
};

static int64_t view_elem_size(const yl_view& v) { return v.dtype == YL_F32 ? 4 : (v.dtype == YL_FP8 ? 1 : 2); }

// bf16 views are plane ranges of a C8-planar guarded buffer (yl_internal.h); fp32 views are NHWC.
static int check_view(const yl_view& v, int batch, int64_t ws_bytes, const char* what) {
  if (v.offset < 0) return fail(-1, std::string(what) + ": view is absent");
  if (v.H <= 0 || v.W <= 0 || v.C <= 0 || v.pitch < v.coff + v.C || v.coff < 0)
    return fail(-1, std::string(what) + ": inconsistent view");
  int64_t end;
  if (v.dtype == YL_F32) {
    end = v.offset + (int64_t)batch * v.H * v.W * v.pitch * 4;
  } else if (v.dtype == YL_FP8) {
    if ((v.C | v.pitch | v.coff) & 15)
      return fail(-3, std::string(what) + ": fp8 channel counts / offsets must be multiples of 16 (planar layout)");
    end = v.offset + (int64_t)(v.pitch / 16) * plane_bytes(geo_of(batch, v.H, v.W));
  } else {
    if ((v.C | v.pitch | v.coff) & 7)
      return fail(-3, std::string(what) + ": bf16 channel counts / offsets must be multiples of 8 (planar layout)");
    end = v.offset + (int64_t)(v.pitch / 8) * plane_bytes(geo_of(batch, v.H, v.W));
  }
  if (end > ws_bytes) return fail(-1, std::string(what) + ": view exceeds the workspace");
  if (v.offset % 16) return fail(-1, std::string(what) + ": view offset must be 16-byte aligned");
  return 0;
}

static PView pview(char* ws, const yl_view& v, int batch) {
  const Geo g = geo_of(batch, v.H, v.W);
  PView p;
  p.base = reinterpret_cast<__bf16*>(ws + v.offset + (int64_t)(v.coff / (v.dtype == YL_FP8 ? 16 : 8)) * plane_bytes(g));
  p.plane = (long long)g.plane_px * 8;
  p.W1 = g.W1;
  p.img = g.img;
  p.g0 = g.g0;
  return p;
}

// yl_op.out8: e4m3 copy of output channels [out8_c0, out8_c0 + out8.C) written by the producer's epilogue
static int check_out8(const yl_op& o, int batch, int64_t ws_bytes, const std::string& tag) {
  if (o.out8.offset < 0) return 0;
  if (int rc = check_view(o.out8, batch, ws_bytes, (tag + " out8").c_str())) return rc;
  if (o.out8.dtype != YL_FP8 || o.out8.H != o.out.H || o.out8.W != o.out.W || o.out8_c0 < 0 || (o.out8_c0 & 7) ||
      o.out8_c0 + o.out8.C > o.out.C || !(o.out8_scale > 0.f) || o.out.dtype != YL_BF16)
    return fail(-1, tag + ": out8 must be an fp8 view of the output's dims covering a channel range of a bf16 output");
  if (o.out8_only && (o.out8_c0 != 0 || o.out8.C != o.out.C)) return fail(-1, tag + ": out8_only needs a copy of all output channels");
  return 0;
}

static void fill_out8(ConvArgs& a, const yl_op& o, char* ws, int B) {
  if (o.out8.offset < 0) return;
  a.out8 = ws + o.out8.offset + (int64_t)(o.out8.coff / 16) * plane_bytes(geo_of(B, o.out8.H, o.out8.W));
  a.out8_plane = (unsigned)plane_bytes(geo_of(B, o.out8.H, o.out8.W));
  a.out8_bytes = (unsigned)((int64_t)(o.out8.C / 16) * a.out8_plane);
  a.out8_inv_scale = 1.0f / o.out8_scale;
  a.out8_c0 = o.out8_c0;
  a.out8_n = o.out8.C;
  a.out8_only = o.out8_only;
}

template <typename T>
static int dev_upload(yl_engine* e, const std::vector<T>& h, T** d) {
  void* p = nullptr;
  YL_HIP(hipMalloc(&p, h.size() * sizeof(T) + 64));
  e->allocs.push_back(p);
  YL_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  *d = reinterpret_cast<T*>(p);
  return 0;
}

extern "C" const char* yl_last_error(void) { return g_err.c_str(); }
extern "C" int yl_abi_version(void) { return YL_ABI_VERSION; }
extern "C" int yl_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return fail(-2, "hipGetDeviceCount failed");
  return n;
}

extern "C" void yl_engine_destroy(yl_engine* e) {
  if (!e) return;
  e->graphs.clear();
  e->dgraphs.clear();
  for (auto& kv : e->rgraphs) {
    if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    if (kv.second.g) (void)hipGraphDestroy(kv.second.g);
  }
  e->rgraphs.clear();
  for (auto& kv : e->lplans) yl_engine::destroy_plan(kv.second);
  e->lplans.clear();
  for (auto& kv : e->sets) {
    if (kv.second.cap) (void)hipStreamDestroy(kv.second.cap);
    for (hipStream_t st : kv.second.lanes)
      if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t ev : kv.second.events) (void)hipEventDestroy(ev);
  }
  for (void* p : e->allocs) (void)hipFree(p);
  delete e;
}

extern "C" int yl_engine_create(const yl_op* ops, int n_ops, int batch, int in_h, int in_w,
                                const yl_view* input_view, const yl_level* levels, int n_levels, int nc,
                                int64_t workspace_bytes, yl_engine** out) {
  if (!ops || !out || !input_view || n_ops <= 0 || batch <= 0 || n_levels < 0 || n_levels > 8 ||
      (n_levels > 0 && !levels))
    return fail(-1, "yl_engine_create: bad arguments");
  std::unique_ptr<yl_engine, void (*)(yl_engine*)> e(new yl_engine, yl_engine_destroy);
  e->batch = batch;
  e->in_h = in_h;
  e->in_w = in_w;
  e->nc = nc;
  e->ws_bytes = workspace_bytes;
  e->input_view = *input_view;
  if (int rc = check_view(*input_view, batch, workspace_bytes, "input")) return rc;
  if (input_view->C != 8 || input_view->pitch != 8 || input_view->H != in_h || input_view->W != in_w ||
      input_view->dtype != YL_BF16)
    return fail(-1, "input view must be one bf16 plane [H][W][8]");
  int a0 = 0;
  for (int i = 0; i < n_levels; ++i) {
    const yl_level& L = levels[i];
    const int64_t pb = plane_bytes(geo_of(batch, L.H, L.W));
    if (L.nc8 < nc || L.nc8 % 8) return fail(-1, "level nc8 must be nc rounded up to a multiple of 8");
    if (L.offset_cls < 0 || L.offset_cls % 16 || L.offset_cls + (L.nc8 / 8) * pb > workspace_bytes)
      return fail(-1, "level view exceeds the workspace");
    // the boxes: 64 bf16 DFL logits at `offset` and / or the decoded (cx, cy, w, h) fp32 at `offset_xywh` (ABI v21)
    if (L.offset < 0 && L.offset_xywh < 0) return fail(-1, "level needs its DFL logits (offset) or its decoded boxes (offset_xywh)");
    if (L.offset >= 0 && (L.offset % 16 || L.offset + 8 * pb > workspace_bytes)) return fail(-1, "level view exceeds the workspace");
    if (L.offset_xywh >= 0 && (L.offset_xywh % 16 || L.offset_xywh + (int64_t)batch * L.H * L.W * 16 > workspace_bytes))
      return fail(-1, "level boxes (offset_xywh) exceed the workspace");
    e->levels.push_back(L);
    a0 += L.H * L.W;
  }
  e->A = a0;

  for (int i = 0; i < n_ops; ++i) {
    OpRec r;
    r.op = ops[i];
    const yl_op& o = r.op;
    const std::string tag = "op " + std::to_string(i);
    if (int rc = check_view(o.in, batch, workspace_bytes, (tag + " in").c_str())) return rc;
    if (int rc = check_view(o.out, batch, workspace_bytes, (tag + " out").c_str())) return rc;
    const double px_in = (double)batch * o.in.H * o.in.W, px_out = (double)batch * o.out.H * o.out.W;
    switch (o.kind) {
      case YL_OP_CONV: {
        if (!o.weight || !o.bias) return fail(-1, tag + ": conv needs weight and bias");
        if (o.in.dtype != YL_BF16 && o.in.dtype != YL_FP8) return fail(-1, tag + ": conv input must be bf16 or fp8");
        const int pad = o.k / 2;
        const int Ho = (o.in.H + 2 * pad - o.k) / o.stride + 1, Wo = (o.in.W + 2 * pad - o.k) / o.stride + 1;
        if (Ho != o.out.H || Wo != o.out.W) return fail(-1, tag + ": conv output dims mismatch");
        if (o.in.dtype == YL_FP8) {  // e4m3 x e4m3 on the block-scaled matrix pipe (conv_f8.hip)
          if (o.post_weight || o.in2.offset >= 0 || o.stem_weight || o.cin_weight || o.out.dtype == YL_F32 || o.dfl_level)
            return fail(-1, tag + ": an fp8 conv is a plain conv (no fused follow-up / upsampled input / stem, bf16 or fp8 output)");
          if (!(o.in_scale > 0.f) || (o.out.dtype == YL_FP8 && !(o.out_scale > 0.f)))
            return fail(-1, tag + ": fp8 conv needs positive in_scale (and out_scale for fp8 output)");
          if (o.res.offset >= 0) {
            if (int rc = check_view(o.res, batch, workspace_bytes, (tag + " res").c_str())) return rc;
            if (o.res.H != Ho || o.res.W != Wo || o.res.C != o.out.C || (o.res.dtype != YL_BF16 && o.res.dtype != YL_FP8) ||
                (o.res.dtype == YL_FP8 && !(o.res_scale > 0.f)))
              return fail(-1, tag + ": residual view mismatch");
          }
          if (int rc = check_out8(o, batch, workspace_bytes, tag)) return rc;
          if (int rc = conv_f8_choose(batch, o.in.C, o.out.C, o.k, o.stride, Ho, Wo, &r.cfg)) return rc;
          std::vector<uint8_t> pk;
          std::vector<float> ws_;
          conv_f8_pack_weights(o.weight, o.out.C, o.in.C, o.k, r.cfg, &pk, &ws_);
          uint8_t* d8 = nullptr;
          if (int rc = dev_upload(e.get(), pk, &d8)) return rc;
          r.d_wpk = reinterpret_cast<__bf16*>(d8);
          std::vector<float> alpha(ws_.size(), 0.f), bias(ws_.size(), 0.f);
          for (int c = 0; c < o.out.C; ++c) {
            alpha[c] = o.in_scale * ws_[c];
            bias[c] = o.bias[c];
          }
          if (int rc = dev_upload(e.get(), alpha, &r.d_alpha)) return rc;
          if (int rc = dev_upload(e.get(), bias, &r.d_bias)) return rc;
          r.f8 = true;
          r.flops = 2.0 * px_out * o.out.C * o.in.C * o.k * o.k;
          r.bytes = px_in * o.in.C + px_out * o.out.C * view_elem_size(o.out) +
                    (o.res.offset >= 0 ? px_out * o.out.C * view_elem_size(o.res) : 0) + (double)o.out.C * o.in.C * o.k * o.k;
          r.name = r.cfg.name;
          break;
        }
        if (o.res.offset >= 0) {
          if (int rc = check_view(o.res, batch, workspace_bytes, (tag + " res").c_str())) return rc;
          if (o.res.H != Ho || o.res.W != Wo || o.res.C != o.out.C || o.res.dtype != YL_BF16)
            return fail(-1, tag + ": residual view mismatch");
        }
        const bool up = o.in2.offset >= 0;
        if (up) {
          if (int rc = check_view(o.in2, batch, workspace_bytes, (tag + " in2").c_str())) return rc;
          if (o.up != 2 || o.k != 1 || o.stride != 1 || o.in2.dtype != YL_BF16 || o.in2.H * 2 != o.in.H ||
              o.in2.W * 2 != o.in.W || o.in2.C % 32 || o.in.C % 32 || o.cin_weight || o.post_weight)
            return fail(-1, tag + ": upsampled second input needs a plain 1x1 conv, up == 2, channel counts % 32 == 0");
        }
        const int cin_total = o.in.C + (up ? o.in2.C : 0);
        // weight Cin may be smaller than the (padded) input view channels: stem 3 -> 8
        const int cin_w = o.cin_weight > 0 ? o.cin_weight : cin_total;
        const bool post = o.post_weight != nullptr;
        const int cout1 = post ? o.post_cin : o.out.C;  // channels the main GEMM produces
        if (o.dfl_level) {  // this conv computes a Detect level's DFL logits: the box decode runs in its epilogue (dfl_epilogue)
          if (o.dfl_level < 0 || o.dfl_level > n_levels) return fail(-1, tag + ": dfl_level names no level");
          const yl_level& L = levels[o.dfl_level - 1];
          if (L.offset_xywh < 0 || L.H != o.out.H || L.W != o.out.W || o.out.C != 64 || o.out.dtype != YL_BF16 || cout1 != 64 ||
              (!post && o.k != 1) || up || o.res.offset >= 0 || o.out8.offset >= 0 || o.stem_weight || o.cin_weight || (post ? o.post_act : o.act))
            return fail(-1, tag + ": the DFL epilogue needs a plain 1x1 conv 64 <- c without activation (or such a conv fused behind a "
                              "conv of 64 channels) on a level with offset_xywh");
          if (!o.dfl_keep && L.offset >= 0 && L.offset == o.out.offset)
            return fail(-1, tag + ": the level announces stored DFL logits (offset) but the conv does not keep them (dfl_keep)");
        }
        if (post && (!o.post_bias || o.res.offset >= 0 || o.out.C > o.post_cin))
          return fail(-1, tag + ": fused follow-up conv needs a bias, no residual and out.C <= post_cin");
        if (int rc = conv_choose(batch, cin_total, cout1, o.k, o.stride, Ho, Wo, post, &r.cfg)) return rc;
        // wide plain convs (1x1, and 3x3 with Cin >= 64, Cout % 128 == 0): the GEMM-class kernel (both operands through LDS);
        // it has its own filter packing (BM / 16 tiles).  Its 3x3 route cannot write an e4m3 copy (out8): such a conv
        // stays on conv_kernel.
        if (!post && !o.stem_weight && !o.cin_weight && o.out.dtype == YL_BF16 && !(up && o.res.offset >= 0) &&
            !(o.k == 3 && o.out8.offset >= 0) && !o.dfl_level)
          conv_gemm_choose(batch, cin_total, cout1, o.k, o.stride, Ho, Wo, &r.cfg, up ? o.in2.C : 0);
        if (o.out8_only && o.out8.offset < 0) return fail(-1, tag + ": out8_only without out8");
        if (o.out8.offset >= 0) {
          if (int rc = check_out8(o, batch, workspace_bytes, tag)) return rc;
          if (post || o.stem_weight) return fail(-1, tag + ": out8 needs a plain conv");
          if (r.cfg.MT & 1)  // the copy is written by tile pairs: take the best even tiling instead
            if (int rc = conv_choose_even(batch, cin_total, cout1, o.k, o.stride, Ho, Wo, &r.cfg)) return rc;
        }
        std::vector<uint16_t> pk;
        conv_pack_weights(o.weight, cout1, cin_w, o.k, r.cfg, &pk);
        if (post) {
          ConvConfig c2 = r.cfg;  // same tile count; K = post_cin channels, one tap
          c2.ncb = 1;
          c2.cg = o.post_cin / 8;
          c2.ksteps = r.cfg.MT / 2;
          c2.ntaps = 1;
          c2.wpk_elems = (size_t)c2.ksteps * c2.MT * 64 * 8;
          std::vector<uint16_t> pk2;
          conv_pack_weights(o.post_weight, o.out.C, o.post_cin, 1, c2, &pk2);
          uint16_t* d2 = nullptr;
          if (int rc = dev_upload(e.get(), pk2, &d2)) return rc;
          r.d_wpk2 = reinterpret_cast<__bf16*>(d2);
          std::vector<float> b2((size_t)r.cfg.MT * 16 + 16, 0.f);
          for (int c = 0; c < o.out.C; ++c) b2[c] = o.post_bias[c];
          if (int rc = dev_upload(e.get(), b2, &r.d_bias2)) return rc;
        }
        uint16_t* d = nullptr;
        if (int rc = dev_upload(e.get(), pk, &d)) return rc;
        r.d_wpk = reinterpret_cast<__bf16*>(d);
        std::vector<float> bias((size_t)r.cfg.ncb * r.cfg.MT * 16 + 16, 0.f);
        for (int c = 0; c < cout1; ++c) bias[c] = o.bias[c];
        if (int rc = dev_upload(e.get(), bias, &r.d_bias)) return rc;
        if (o.stem_weight) {  // the network's first layer fused in front (stem2.hip)
          const int sh = (in_h + 2 - 3) / 2 + 1, sw = (in_w + 2 - 3) / 2 + 1;
          if (i != 0 || !o.stem_bias || o.k != 3 || o.stride != 2 || up || o.res.offset >= 0 || o.cin_weight ||
              o.out.dtype != YL_BF16 || o.lane != 0)
            return fail(-1, tag + ": a fused stem needs a plain 3x3 stride-2 conv as op 0");
          if (o.in.H != sh || o.in.W != sw || (o.in.C != 16 && o.in.C != 32) || (cout1 != 32 && cout1 != 64) ||
              r.cfg.ncb != 1 || r.cfg.ksteps > 9)
            return fail(-1, tag + ": fused stem: unsupported widths or dims");
          std::vector<float> wt((size_t)o.in.C * 27);
          for (size_t q = 0; q < wt.size(); ++q) {  // round to bf16 like every other conv filter
            uint32_t u;
            memcpy(&u, &o.stem_weight[q], 4);
            u += 0x7fffu + ((u >> 16) & 1u);
            u &= 0xffff0000u;
            memcpy(&wt[q], &u, 4);
          }
          if (int rc = dev_upload(e.get(), wt, &r.d_w0)) return rc;
          std::vector<float> b0(o.stem_bias, o.stem_bias + o.in.C);
          if (int rc = dev_upload(e.get(), b0, &r.d_b0)) return rc;
        }
        if (o.dfl_level && (r.cfg.MT != 4 || r.cfg.ncb != 1))
          return fail(-3, tag + ": the DFL epilogue needs the conv's 64 channels as four tiles of one wave");
        // Detect's cv2[i][1] + cv2[i][2] + decode on its own (not a member of a level group): the Cout-split filter-in-registers
        // kernel's form of it where the map is large enough (level 0 at 640x640 from batch 8 up); same packed filters
        if (o.dfl_level && post && o.group == 0 && o.lane == 0 && o.out.dtype == YL_BF16 && o.in.dtype == YL_BF16 && !up)
          conv3_cs_choose_post(batch, cin_total, cout1, o.out.C, o.k, o.stride, Ho, Wo, &r.cfg);
        // a deep-K 3x3 conv with 64 output channels on its own (the wider models' cv2[i][0]): the band kernel, one conv per launch
        if (!post && !up && o.group == 0 && o.lane == 0 && o.k == 3 && o.stride == 1 && o.act && o.res.offset < 0 && o.out8.offset < 0 &&
            !o.stem_weight && !o.cin_weight && o.out.dtype == YL_BF16 && o.in.dtype == YL_BF16 && o.out.C == 64 && !r.cfg.gemm &&
            r.cfg.MT == 4 && r.cfg.ncb == 1 && !r.cfg.kord && (o.in.C == 128 || o.in.C == 256 || o.in.C == 512) && env_int("YL_CONV_BAND", 1) &&
            (long long)batch * Ho * Wo >= 4096 && conv3_band_plan(o.in.C, 64, Ho, Wo, nullptr, true)) {
          r.band1 = true;
          r.cfg.reg = r.cfg.cs = 0;
          r.cfg.name = "conv3x3_band_mfma";
        }
        r.flops = 2.0 * px_out * cout1 * cin_w * o.k * o.k + (post ? 2.0 * px_out * o.out.C * o.post_cin : 0.0);
        // (a DFL conv stores 16 bytes of decoded box per pixel; the 64 bf16 logits only with dfl_keep)
        const double out_b = o.dfl_level ? px_out * (16.0 + (o.dfl_keep ? 128.0 : 0.0)) : px_out * o.out.C * view_elem_size(o.out);
        r.bytes = px_in * o.in.C * 2 + (up ? px_in / 4 * o.in2.C * 2 : 0.0) + out_b +
                  (o.res.offset >= 0 ? px_out * o.out.C * 2 : 0) + (double)cout1 * cin_w * o.k * o.k * 2;
        r.name = r.cfg.name;
        if (o.stem_weight) {
          r.flops += 2.0 * px_in * o.in.C * 27;
          r.bytes = (double)batch * 3 * in_h * in_w * 4 + px_out * o.out.C * 2 + (double)cout1 * cin_w * 9 * 2;
          r.name = "stem+conv3x3s2_mfma";
        }
        break;
      }
      case YL_OP_DWCONV3: {
        if (!o.weight || !o.bias) return fail(-1, tag + ": dwconv needs weight and bias");
        if (o.in.H != o.out.H || o.in.W != o.out.W || o.in.dtype != YL_BF16 || o.out.dtype != YL_BF16)
          return fail(-1, tag + ": dwconv dims mismatch");
        const bool pw = o.post_weight != nullptr;  // depthwise 3x3 -> pointwise 1x1 in one launch
        const int C = pw ? o.in.C : o.out.C;
        if (pw) {
          if (!o.post_bias || o.accumulate || o.grp || o.post_cin != C)
            return fail(-1, tag + ": fused depthwise+pointwise needs a plain depthwise conv and post_cin == in.C");
          // a further 1x1 behind the pointwise conv (stem_weight [out.C][mid], mid = reserved2 channels of the pointwise)
          const bool tail = o.stem_weight != nullptr;
          const int pw_cout = tail ? o.reserved2 : o.out.C;
          if (tail && (!o.stem_bias || pw_cout <= 0 || pw_cout % 8))
            return fail(-1, tag + ": depthwise+pointwise+1x1 needs stem_bias and reserved2 = pointwise channels (% 8 == 0)");
          if (int rc = conv_choose(batch, C, pw_cout, 1, 1, o.out.H, o.out.W, 0, &r.cfg)) return rc;
          // 128 pointwise channels (YOLO11s' class branch, head.py:41: c3 = max(ch[0], min(nc, 100)) = 128): conv_choose gives two
          // blocks of four tiles; the fused launch needs every output channel in ONE workgroup (the depthwise tile in LDS is its
          // GEMM's B operand) -- eight tiles per wave, one block, while the filter (C / 32 k-steps x 8 KiB) fits the slab
          if (r.cfg.ncb == 2 && r.cfg.MT == 4 && pw_cout == 128 && !tail && C % 32 == 0 && C <= 256 && !r.cfg.gemm) {
            r.cfg.MT = 8;
            r.cfg.ncb = 1;
            r.cfg.kchunk = r.cfg.ksteps;
            r.cfg.NT = 2;
            r.cfg.npatches = (int)(((long long)batch * o.out.H * o.out.W + 31) / 32);
            r.cfg.wpk_elems = (size_t)r.cfg.ksteps * 8 * 64 * 8;
            r.cfg.lds_bytes = (size_t)r.cfg.ksteps * 8 * 1024;
            r.cfg.reg = r.cfg.cs = r.cfg.kord = 0;
          }
          if (r.cfg.NT > 2) {
            r.cfg.NT = 2;
            r.cfg.npatches = (int)(((long long)batch * o.out.H * o.out.W + 31) / 32);
          }
          // A deep launch on a small map (Detect level 2: 256 channels @20x20 = 16 channel blocks one after the other in every
          // workgroup, 100 workgroups at 32 pixels per wave for batch 32): 16 pixels per wave double the workgroups and halve
          // each one's serial chain.  Launches that join a level group keep the group's tiling.
          static const int small_nt1 = tune_int("YL_DWPW_SMALL_NT1", 1);
          const int small_blocks = small_nt1 > 1 ? small_nt1 : 256;  // (> 1: the workgroup count below which it applies, development)
          if (small_nt1 && o.group == 0 && r.cfg.NT == 2 && C >= 128 && (r.cfg.npatches + 3) / 4 < small_blocks) {
            r.cfg.NT = 1;
            r.cfg.npatches = (int)(((long long)batch * o.out.H * o.out.W + 15) / 16);
          }
          if (r.cfg.ncb != 1 || r.cfg.kchunk < r.cfg.ksteps)
            return fail(-3, tag + ": fused depthwise+pointwise needs Cout <= 80 and a filter that fits the slab");
          std::vector<uint16_t> pk;
          conv_pack_weights(o.post_weight, pw_cout, C, 1, r.cfg, &pk);
          uint16_t* d = nullptr;
          if (int rc = dev_upload(e.get(), pk, &d)) return rc;
          r.d_wpk = reinterpret_cast<__bf16*>(d);
          std::vector<float> b2((size_t)r.cfg.MT * 16 + 16, 0.f);
          for (int c = 0; c < pw_cout; ++c) b2[c] = o.post_bias[c];
          if (int rc = dev_upload(e.get(), b2, &r.d_bias2)) return rc;
          if (tail) {
            ConvConfig c3{};
            if (int rc = conv_choose(batch, pw_cout, o.out.C, 1, 1, o.out.H, o.out.W, 0, &c3)) return rc;
            if (c3.MT != r.cfg.MT || c3.ncb != 1 || c3.kchunk < c3.ksteps)
              return fail(-3, tag + ": depthwise+pointwise+1x1: both 1x1 convs must have the same number of channel tiles");
            std::vector<uint16_t> pk3;
            conv_pack_weights(o.stem_weight, o.out.C, pw_cout, 1, c3, &pk3);
            uint16_t* d3 = nullptr;
            if (int rc = dev_upload(e.get(), pk3, &d3)) return rc;
            r.d_wpk3 = reinterpret_cast<__bf16*>(d3);
            std::vector<float> b3((size_t)c3.MT * 16 + 16, 0.f);
            for (int c = 0; c < o.out.C; ++c) b3[c] = o.stem_bias[c];
            if (int rc = dev_upload(e.get(), b3, &r.d_bias3)) return rc;
            r.cfg.kchunk = c3.ksteps;  // (kchunk == ksteps was checked above) carries the tail's k-steps to run_op
          }
        }
        if (pw) {
          if (C % 16) return fail(-1, tag + ": depthwise + pointwise needs C % 16 == 0");
          std::vector<uint16_t> dpk;
          dw_pack_mfma(o.weight, C, &dpk);
          if (int rc = dev_upload(e.get(), dpk, &r.d_dwpk)) return rc;
        } else {
          std::vector<float> wt((size_t)9 * C);
          for (int c = 0; c < C; ++c)
            for (int t = 0; t < 9; ++t) wt[(size_t)t * C + c] = o.weight[(size_t)c * 9 + t];
          if (int rc = dev_upload(e.get(), wt, &r.d_w)) return rc;
        }
        std::vector<float> bias(o.bias, o.bias + C);
        if (int rc = dev_upload(e.get(), bias, &r.d_bias)) return rc;
        {
          const int pwc = pw && o.stem_weight ? o.reserved2 : o.out.C;
          r.flops = 2.0 * px_out * C * 9 + (pw ? 2.0 * px_out * C * pwc : 0.0) +
                    (pw && o.stem_weight ? 2.0 * px_out * pwc * o.out.C : 0.0);
        }
        r.bytes = pw ? px_out * (C + o.out.C) * 2 : px_out * C * 2 * (o.accumulate ? 3 : 2);
        r.name = pw ? (o.stem_weight ? "dw3x3+pw1x1+1x1_mfma" : "dw3x3+pw1x1_mfma") : "dwconv3x3";
        break;
      }
      case YL_OP_QUANT: {
        if (o.in.dtype != YL_BF16 || o.out.dtype != YL_FP8 || o.in.H != o.out.H || o.in.W != o.out.W || o.in.C != o.out.C ||
            !(o.out_scale > 0.f))
          return fail(-1, tag + ": quantize: bf16 in, fp8 out, same dims, positive out_scale");
        r.flops = 0;
        r.bytes = px_in * o.in.C * 3;
        r.name = "quantize_f8";
        break;
      }
      case YL_OP_BNECK2: {
        if (!o.weight || !o.bias || !o.post_weight || !o.post_bias) return fail(-1, tag + ": bottleneck needs both convs' weights and biases");
        const int c = o.in.C;
        if (o.in.dtype != YL_BF16 || o.out.dtype != YL_BF16 || o.in.H != o.out.H || o.in.W != o.out.W || o.out.C != c ||
            (c != 32 && c != 64) || o.post_cin != 9 * c || o.lane != 0 || o.in.offset == o.out.offset)
          return fail(-1, tag + ": bottleneck: c = 32 or 64, same dims in and out, different buffers");
        const float* ws[2] = {o.weight, o.post_weight};
        const float* bs[2] = {o.bias, o.post_bias};
        __bf16** dw[2] = {&r.d_wpk, &r.d_wpk2};
        float** db[2] = {&r.d_bias, &r.d_bias2};
        for (int q = 0; q < 2; ++q) {
          ConvConfig cc{};
          if (int rc = conv_choose(batch, c, c, 3, 1, o.out.H, o.out.W, 0, &cc)) return rc;
          if (cc.MT != c / 16 || cc.ksteps != 9 * c / 32 || cc.ncb != 1) return fail(-3, tag + ": bottleneck: unexpected tiling");
          std::vector<uint16_t> pk;
          conv_pack_weights(ws[q], c, c, 3, cc, &pk);
          uint16_t* d = nullptr;
          if (int rc = dev_upload(e.get(), pk, &d)) return rc;
          *dw[q] = reinterpret_cast<__bf16*>(d);
          std::vector<float> bias((size_t)cc.MT * 16 + 16, 0.f);
          for (int ch = 0; ch < c; ++ch) bias[ch] = bs[q][ch];
          if (int rc = dev_upload(e.get(), bias, db[q])) return rc;
        }
        r.flops = 2.0 * px_out * 2.0 * 9.0 * c * c;
        r.bytes = px_out * 2.0 * c * 2;
        r.name = "bottleneck_fused_mfma";
        break;
      }
      case YL_OP_C3K: {
        const int c = o.in.C, c_ = c / 2;
        if (!o.weight || !o.bias || o.n_extra != 5 || !o.extra_weight || !o.extra_bias)
          return fail(-1, tag + ": c3k needs the stacked cv1|cv2 filter and five extra filters (two Bottlenecks, cv3)");
        for (int q = 0; q < 5; ++q)
          if (!o.extra_weight[q] || !o.extra_bias[q]) return fail(-1, tag + ": c3k: null extra filter");
        if (o.in.dtype != YL_BF16 || o.out.dtype != YL_BF16 || o.in.H != o.out.H || o.in.W != o.out.W || o.out.C != c || c != 128 ||
            o.lane != 0 || !c3k_resident_fits(c_, o.in.H, o.in.W))
          return fail(-3, tag + ": c3k: needs 128 channels in and out (inner width 64) on a map that fits one workgroup's LDS");
        // every conv as 64-channel blocks of four tiles (conv_pack_weights order), biases padded to the blocks
        const float* ws[6] = {o.weight, o.extra_weight[0], o.extra_weight[1], o.extra_weight[2], o.extra_weight[3], o.extra_weight[4]};
        const float* bs[6] = {o.bias, o.extra_bias[0], o.extra_bias[1], o.extra_bias[2], o.extra_bias[3], o.extra_bias[4]};
        const int cin[6] = {c, c_, c_, c_, c_, c}, cout[6] = {c, c_, c_, c_, c_, c}, ks[6] = {1, 3, 3, 3, 3, 1};
        for (int q = 0; q < 6; ++q) {
          ConvConfig cc{};
          cc.MT = 4;
          cc.ncb = cout[q] / 64;
          cc.cg = cin[q] / 8;
          cc.ntaps = ks[q] * ks[q];
          cc.ksteps = cc.ntaps * cc.cg / 4;
          cc.wpk_elems = (size_t)cc.ncb * cc.ksteps * cc.MT * 64 * 8;
          std::vector<uint16_t> pk;
          conv_pack_weights(ws[q], cout[q], cin[q], ks[q], cc, &pk);
          uint16_t* d = nullptr;
          if (int rc = dev_upload(e.get(), pk, &d)) return rc;
          r.d_xw[q] = reinterpret_cast<__bf16*>(d);
          std::vector<float> bias((size_t)cout[q] + 16, 0.f);
          for (int ch = 0; ch < cout[q]; ++ch) bias[ch] = bs[q][ch];
          if (int rc = dev_upload(e.get(), bias, &r.d_xb[q])) return rc;
        }
        r.flops = 2.0 * px_out * (2.0 * c * c + 4.0 * 9.0 * c_ * c_);
        r.bytes = px_out * 2.0 * c * 2 + 2.0 * (2.0 * c * c + 36.0 * c_ * c_);
        r.name = "c3k_resident_mfma";
        break;
      }
      case YL_OP_SPPF: {
        const bool tail = o.n_extra == 2;
        if (!o.weight || !o.bias || (o.n_extra != 1 && o.n_extra != 2) || !o.extra_weight || !o.extra_bias)
          return fail(-1, tag + ": sppf needs cv1's filter and one or two extra filters (cv2, the follow-up 1x1 conv)");
        for (int q = 0; q < o.n_extra; ++q)
          if (!o.extra_weight[q] || !o.extra_bias[q]) return fail(-1, tag + ": sppf: null extra filter");
        if (o.in.dtype != YL_BF16 || o.out.dtype != YL_BF16 || o.in.H != o.out.H || o.in.W != o.out.W || o.lane != 0 ||
            !sppf_resident_fits(o.in.C, o.out.C, tail ? o.out.C : 0, o.in.H, o.in.W))
          return fail(-3, tag + ": sppf: 256 channels in and out (hidden width 128) on a map of at most 448 pixels");
        const int c1 = o.in.C, c_ = c1 / 2, c2 = o.out.C;
        const float* ws[3] = {o.weight, o.extra_weight[0], tail ? o.extra_weight[1] : nullptr};
        const float* bs[3] = {o.bias, o.extra_bias[0], tail ? o.extra_bias[1] : nullptr};
        const int cin[3] = {c1, 4 * c_, c2}, cout[3] = {c_, c2, c2};
        for (int q = 0; q < (tail ? 3 : 2); ++q) {
          ConvConfig cc{};
          cc.MT = 4;
          cc.ncb = cout[q] / 64;
          cc.cg = cin[q] / 8;
          cc.ntaps = 1;
          cc.ksteps = cin[q] / 32;
          cc.wpk_elems = (size_t)cc.ncb * cc.ksteps * cc.MT * 64 * 8;
          std::vector<uint16_t> pk;
          conv_pack_weights(ws[q], cout[q], cin[q], 1, cc, &pk);
          uint16_t* d = nullptr;
          if (int rc = dev_upload(e.get(), pk, &d)) return rc;
          r.d_xw[q] = reinterpret_cast<__bf16*>(d);
          std::vector<float> bias((size_t)cout[q] + 16, 0.f);
          for (int ch = 0; ch < cout[q]; ++ch) bias[ch] = bs[q][ch];
          if (int rc = dev_upload(e.get(), bias, &r.d_xb[q])) return rc;
        }
        r.flops = 2.0 * px_out * ((double)c1 * c_ + 4.0 * c_ * c2 + (tail ? (double)c2 * c2 : 0.0));
        r.bytes = px_out * 2.0 * (c1 + c2) + 2.0 * ((double)c1 * c_ + 4.0 * c_ * c2 + (tail ? (double)c2 * c2 : 0.0));
        r.name = "sppf_resident_mfma";
        break;
      }
      case YL_OP_PSA: {
        const int c = o.in.C / 2;  // in / out = cat(a, b): the block works on b and copies a
        if (!o.weight || !o.bias || o.n_extra != 4 || !o.extra_weight || !o.extra_bias)
          return fail(-1, tag + ": psa needs the qkv filter and four extra filters (pe, proj, ffn.0, ffn.1)");
        for (int q = 0; q < 4; ++q)
          if (!o.extra_weight[q] || !o.extra_bias[q]) return fail(-1, tag + ": psa: null extra filter");
        if (o.in.dtype != YL_BF16 || o.out.dtype != YL_BF16 || o.in.offset == o.out.offset || o.out.C != 2 * c || o.in.C != 2 * c ||
            o.in.H != o.out.H || o.in.W != o.out.W || o.lane != 0 || !psa_resident_fits(c, o.heads, o.key_dim, o.head_dim, o.in.H, o.in.W))
          return fail(-3, tag + ": psa: in and out are DIFFERENT buffers of 256 channels (cat(a, b)), 2 heads (key_dim 32, head_dim 64), "
                            "at most 400 pixels per image");
        auto pack = [&](const float* w, int cout, int cin, std::vector<uint16_t>* pk) {
          ConvConfig cc{};
          cc.MT = 4;
          cc.ncb = cout / 64;
          cc.cg = cin / 8;
          cc.ntaps = 1;
          cc.ksteps = cin / 32;
          cc.wpk_elems = (size_t)cc.ncb * cc.ksteps * cc.MT * 64 * 8;
          conv_pack_weights(w, cout, cin, 1, cc, pk);
        };
        {
          std::vector<uint16_t> pk;
          pack(o.weight, 2 * c, c, &pk);
          uint16_t* d = nullptr;
          if (int rc = dev_upload(e.get(), pk, &d)) return rc;
          r.d_xw[0] = reinterpret_cast<__bf16*>(d);
          std::vector<uint16_t> tail, part;
          pack(o.extra_weight[1], c, c, &part);
          tail.insert(tail.end(), part.begin(), part.end());
          pack(o.extra_weight[2], 2 * c, c, &part);
          tail.insert(tail.end(), part.begin(), part.end());
          pack(o.extra_weight[3], c, 2 * c, &part);
          tail.insert(tail.end(), part.begin(), part.end());
          if (tail.size() * 2 != 160 * 1024) return fail(-3, tag + ": psa: unexpected tail filter size");
          if (int rc = dev_upload(e.get(), tail, &d)) return rc;
          r.d_xw[1] = reinterpret_cast<__bf16*>(d);
        }
        {
          const float* bs[5] = {o.bias, o.extra_bias[0], o.extra_bias[1], o.extra_bias[2], o.extra_bias[3]};
          const int nb[5] = {2 * c, c, c, 2 * c, c};
          for (int q = 0; q < 5; ++q) {
            std::vector<float> bias((size_t)nb[q] + 16, 0.f);
            for (int ch = 0; ch < nb[q]; ++ch) bias[ch] = bs[q][ch];
            if (int rc = dev_upload(e.get(), bias, &r.d_xb[q])) return rc;
          }
          std::vector<float> wt((size_t)9 * c);  // depthwise taps as dwconv3_kernel reads them: [tap][channel], fp32
          for (int ch = 0; ch < c; ++ch)
            for (int t = 0; t < 9; ++t) wt[(size_t)t * c + ch] = o.extra_weight[0][(size_t)ch * 9 + t];
          if (int rc = dev_upload(e.get(), wt, &r.d_w)) return rc;
        }
        const double n_tok = (double)o.in.H * o.in.W;
        r.flops = 2.0 * px_out * (2.0 * c * c + c * c + 4.0 * c * c + 9.0 * c) + 2.0 * batch * o.heads * n_tok * n_tok * (o.key_dim + o.head_dim);
        r.bytes = px_out * 4.0 * c * 2 + 2.0 * (7.0 * c * c);
        r.name = "psa_resident_mfma";
        break;
      }
      case YL_OP_PSA_TAIL: {
        const int c = o.in.C;
        if (!o.weight || !o.bias || o.n_extra != 3 || !o.extra_weight || !o.extra_bias)
          return fail(-1, tag + ": psa tail needs the proj filter and three extra filters (ffn.0, ffn.1, cv2)");
        for (int q = 0; q < 3; ++q)
          if (!o.extra_weight[q] || !o.extra_bias[q]) return fail(-1, tag + ": psa tail: null extra filter");
        if (o.res.offset < 0) return fail(-1, tag + ": psa tail: res = cat(a, b), C2PSA.cv1's output");
        if (int rc = check_view(o.res, batch, workspace_bytes, (tag + " res").c_str())) return rc;
        if (o.in.dtype != YL_BF16 || o.out.dtype != YL_BF16 || o.res.dtype != YL_BF16 || !psa_tail_fits(c, o.out.C, 1) || o.res.C != 2 * c ||
            o.in.H != o.out.H || o.in.W != o.out.W || o.res.H != o.out.H || o.res.W != o.out.W || o.lane != 0)
          return fail(-3, tag + ": psa tail: o of 128 channels, cat(a, b) and the output of 256 channels, one map size");
        auto pack = [&](const float* w, int cout, int cin, std::vector<uint16_t>* pk) {
          ConvConfig cc{};
          cc.MT = 4;
          cc.ncb = cout / 64;
          cc.cg = cin / 8;
          cc.ntaps = 1;
          cc.ksteps = cin / 32;
          cc.wpk_elems = (size_t)cc.ncb * cc.ksteps * cc.MT * 64 * 8;
          conv_pack_weights(w, cout, cin, 1, cc, pk);
        };
        {
          std::vector<uint16_t> tail, part;
          const float* ws4[4] = {o.weight, o.extra_weight[0], o.extra_weight[1], o.extra_weight[2]};
          const int co4[4] = {c, 2 * c, c, 2 * c}, ci4[4] = {c, c, 2 * c, 2 * c};
          for (int q = 0; q < 4; ++q) {
            pack(ws4[q], co4[q], ci4[q], &part);
            tail.insert(tail.end(), part.begin(), part.end());
          }
          if (tail.size() * 2 != 288 * 1024) return fail(-3, tag + ": psa tail: unexpected filter size");
          uint16_t* d = nullptr;
          if (int rc = dev_upload(e.get(), tail, &d)) return rc;
          r.d_xw[1] = reinterpret_cast<__bf16*>(d);
          const float* bs[4] = {o.bias, o.extra_bias[0], o.extra_bias[1], o.extra_bias[2]};
          for (int q = 0; q < 4; ++q) {
            std::vector<float> bias((size_t)co4[q] + 16, 0.f);
            for (int ch = 0; ch < co4[q]; ++ch) bias[ch] = bs[q][ch];
            if (int rc = dev_upload(e.get(), bias, &r.d_xb[q])) return rc;
          }
        }
        r.flops = 2.0 * px_out * (c * c + 2.0 * c * c * 2 + 4.0 * c * c);
        r.bytes = px_out * 2.0 * (c + 2 * c + 2 * c) + 2.0 * 9.0 * c * c;
        r.name = "psa_tail_mfma";
        break;
      }
      case YL_OP_C3K2S: {
        if (!o.weight || !o.bias || !o.post_weight || !o.post_bias || !o.stem_weight || !o.stem_bias)
          return fail(-1, tag + ": c3k2 needs the three convs' weights and biases");
        const int c = o.post_cin * 2 / 9;  // Bottleneck.cv2 is [c][c/2][3][3]: post_cin = 9 * c / 2
        if (o.in.dtype != YL_BF16 || o.out.dtype != YL_BF16 || o.in.H != o.out.H || o.in.W != o.out.W || o.in.coff != 0 ||
            (c != 16 && c != 32) || o.post_cin * 2 != 9 * c || o.in.C < 2 * c || o.in.pitch < 2 * c ||
            (o.out.C != 64 && !(o.out.C == 128 && c == 32)) || o.lane != 0)
          return fail(-1, tag + ": c3k2: supported are c = 16 (-> 64 output channels) and c = 32 (-> 64 or 128)");
        struct Part { const float* w; const float* b; int cin, cout, k; __bf16** dw; float** db; };
        const Part parts[3] = {{o.weight, o.bias, c, c / 2, 3, &r.d_wpk, &r.d_bias},
                               {o.post_weight, o.post_bias, c / 2, c, 3, &r.d_wpk2, &r.d_bias2},
                               {o.stem_weight, o.stem_bias, 3 * c, o.out.C, 1, &r.d_wpk3, &r.d_bias3}};
        const int cp = c / 8;
        const int want_mt[3] = {1, cp / 2, 4}, want_ks[3] = {(9 * cp + 3) / 4, (9 * (cp / 2) + 3) / 4, (3 * cp + 3) / 4};
        const int want_ncb[3] = {1, 1, o.out.C / 64};
        for (int q = 0; q < 3; ++q) {
          ConvConfig cc{};
          if (int rc = conv_choose(batch, parts[q].cin, parts[q].cout, parts[q].k, 1, o.out.H, o.out.W, 0, &cc)) return rc;
          if (cc.MT != want_mt[q] || cc.ksteps != want_ks[q] || cc.ncb != want_ncb[q])
            return fail(-3, tag + ": c3k2: unexpected tiling");
          std::vector<uint16_t> pk;
          conv_pack_weights(parts[q].w, parts[q].cout, parts[q].cin, parts[q].k, cc, &pk);
          uint16_t* d = nullptr;
          if (int rc = dev_upload(e.get(), pk, &d)) return rc;
          *parts[q].dw = reinterpret_cast<__bf16*>(d);
          std::vector<float> bias((size_t)cc.ncb * cc.MT * 16 + 16, 0.f);
          for (int ch = 0; ch < parts[q].cout; ++ch) bias[ch] = parts[q].b[ch];
          if (int rc = dev_upload(e.get(), bias, parts[q].db)) return rc;
        }
        r.flops = 2.0 * px_out * (9.0 * c * c / 2 * 2 + 3.0 * c * o.out.C);
        r.bytes = px_out * (2 * c + o.out.C) * 2;
        r.name = "c3k2_fused_mfma";
        break;
      }
      case YL_OP_STEM: {
        if (i != 0) return fail(-1, tag + ": the stem must be op 0");
        if (!o.weight || !o.bias) return fail(-1, tag + ": stem needs weight and bias");
        if (o.out.H != (in_h + 2 - 3) / 2 + 1 || o.out.W != (in_w + 2 - 3) / 2 + 1 || o.out.dtype != YL_BF16)
          return fail(-1, tag + ": stem output dims mismatch");
        if (o.out.C % 16) return fail(-1, tag + ": stem needs Cout % 16 == 0");
        if (int rc = check_out8(o, batch, workspace_bytes, tag)) return rc;
        if (o.out8.offset >= 0 && (o.out8_c0 != 0 || o.out8.C != o.out.C)) return fail(-1, tag + ": the stem's out8 covers all channels");
        if (o.out8_only && o.out8.offset < 0) return fail(-1, tag + ": out8_only without out8");
        std::vector<float> wt((size_t)o.out.C * 27);
        for (size_t q = 0; q < wt.size(); ++q) {  // round to bf16 like every other conv filter
          uint32_t u;
          memcpy(&u, &o.weight[q], 4);
          u += 0x7fffu + ((u >> 16) & 1u);
          u &= 0xffff0000u;
          memcpy(&wt[q], &u, 4);
        }
        if (int rc = dev_upload(e.get(), wt, &r.d_w)) return rc;
        {
          // the same filter in MFMA A-fragment order, bf16: [16-channel block][lane][8] -- lane (row p = lane & 15, k-group lane >> 4)
          // holds taps k = 8 (lane >> 4) + j of output channel block * 16 + p, zero from tap 27 on: ONE 16-byte load per lane and
          // block in stem_kernel instead of eight scattered dword loads (round 5)
          std::vector<uint16_t> pk((size_t)(o.out.C / 16) * 64 * 8, 0);
          for (int mb = 0; mb < o.out.C / 16; ++mb)
            for (int lane = 0; lane < 64; ++lane)
              for (int j = 0; j < 8; ++j) {
                const int k = (lane >> 4) * 8 + j;
                if (k >= 27) continue;
                uint32_t u;
                memcpy(&u, &wt[(size_t)(mb * 16 + (lane & 15)) * 27 + k], 4);
                pk[((size_t)mb * 64 + lane) * 8 + j] = (uint16_t)(u >> 16);  // (wt is bf16-rounded: the low half is zero)
              }
          uint16_t* d = nullptr;
          if (int rc = dev_upload(e.get(), pk, &d)) return rc;
          r.d_wpk = reinterpret_cast<__bf16*>(d);
        }
        std::vector<float> bias(o.bias, o.bias + o.out.C);
        if (int rc = dev_upload(e.get(), bias, &r.d_bias)) return rc;
        r.flops = 2.0 * px_out * o.out.C * 27;
        r.bytes = (double)batch * 3 * in_h * in_w * 4 + px_out * o.out.C * 2;
        r.name = "stem_conv3x3s2";
        break;
      }
      case YL_OP_SPPF_POOL:
        if (o.out.pitch != o.in.pitch || o.out.offset != o.in.offset || o.in.pitch < o.in.coff + 4 * o.in.C)
          return fail(-1, tag + ": sppf pool expects one buffer holding [x | p5 | p9 | p13]");
        r.bytes = px_in * o.in.C * 2 * 4;
        r.name = "sppf_maxpool";
        break;
      case YL_OP_UPSAMPLE:
        if (o.stride != 1 && o.stride != 2) return fail(-1, tag + ": upsample factor must be 1 or 2");
        if (o.out.H != o.in.H * o.stride || o.out.W != o.in.W * o.stride || o.out.C != o.in.C)
          return fail(-1, tag + ": upsample dims mismatch");
        r.bytes = px_in * o.in.C * 2 + px_out * o.out.C * 2;
        r.name = "upsample_nearest";
        break;
      case YL_OP_ATTENTION:
        if (o.key_dim != 32 || o.head_dim != 64)
          return fail(-3, tag + ": attention supports key_dim 32 / head_dim 64 (every YOLO11 scale)");
        if (o.in.C != o.heads * 128 || o.out.C != o.heads * 64) return fail(-1, tag + ": attention channel mismatch");
        r.flops = 2.0 * batch * o.heads * (double)(o.in.H * o.in.W) * (o.in.H * o.in.W) * (32 + 64);
        r.bytes = px_in * o.in.C * 2 + px_out * o.out.C * 2;
        r.name = "attention";
        break;
      default:
        return fail(-1, tag + ": unknown op kind");
    }
    if (o.lane < 0 || o.lane > 64) return fail(-1, tag + ": bad lane");
    if (o.lane > e->max_lane) e->max_lane = o.lane;
    e->ops.push_back(r);
  }
  // ---- side lanes as separate graphs (see yl_engine::lane_graphs): segments = maximal runs of ops of one lane; the first one
  // must be a main-lane segment (it holds the network's first launch)
  {
    static const int lg = tune_int("YL_LANE_GRAPHS", 1);
    e->lane_graphs = lg != 0 && e->max_lane > 0 && !e->ops.empty() && e->ops[0].op.lane == 0;
    if (e->lane_graphs) {
      for (size_t i = 0; i < e->ops.size();) {
        size_t j = i + 1;
        while (j < e->ops.size() && e->ops[j].op.lane == e->ops[i].op.lane) ++j;
        e->segments.push_back(yl_engine::Segment{e->ops[i].op.lane, i, j});
        i = j;
      }
    }
  }
  // ---- which launches produce the class logits of the Detect levels (the NMS candidate pre-filter rides in their epilogue)
  {
    size_t found = 0;
    static const int conv_pf = env_int("YL_FUSE_PREFILTER_CONV", 1);
    for (OpRec& r : e->ops) {
      // the fused depthwise + pointwise + 1x1 launch of the class branch (YOLO11n), or -- where the branch is too wide for it --
      // the plain 1x1 conv that ends it (conv_kernel's PF instantiation)
      const bool tail = r.op.kind == YL_OP_DWCONV3 && r.d_wpk && r.d_wpk3;
      const bool last = conv_pf && r.op.kind == YL_OP_CONV && r.op.k == 1 && r.op.stride == 1 && !r.op.act && !r.f8 && !r.d_w0 && !r.d_wpk2 &&
                        r.op.res.offset < 0 && r.op.in2.offset < 0 && r.op.out8.offset < 0 && r.op.in.dtype == YL_BF16 &&
                        r.op.out.dtype == YL_BF16 && conv_pf_supported(r.cfg);  // (1x1 convs never join a group launch)
      if (!(tail || last) || (r.op.lane != 0 && !e->lane_graphs) || r.op.out.coff != 0) continue;
      for (size_t li = 0; li < e->levels.size(); ++li) {
        const yl_level& L = e->levels[li];
        if (r.op.out.offset == L.offset_cls && r.op.out.C == L.nc8 && r.op.out.H == L.H && r.op.out.W == L.W) {
          r.cls_level = (int)li;
          ++found;
        }
      }
    }
    static const int fuse_pf = env_int("YL_FUSE_PREFILTER", 1);
    e->cls_tails = fuse_pf && !e->levels.empty() && found == e->levels.size() && (e->max_lane == 0 || e->lane_graphs);
  }
  // ---- group launches: runs of consecutive ops with the same yl_op.group > 0 (the lowering puts the Detect levels'
  // copies of one conv side by side): up to three convs, then up to three depthwise + pointwise launches, each kind of
  // one tiling.  A run that does not fit simply stays ungrouped.
  for (size_t i = 0; i < e->ops.size();) {
    const int gid = e->ops[i].op.group;
    size_t j = i + 1;
    if (gid > 0)
      while (j < e->ops.size() && e->ops[j].op.group == gid) ++j;
    const int n = (int)(j - i);
    if (gid > 0 && n >= 2 && n <= 6) {
      int nc = 0, nd = 0;
      bool ok = true;
      for (size_t k = i; k < j && ok; ++k) {
        const OpRec& r = e->ops[k];
        const yl_op& o = r.op;
        ok = o.lane == 0 && !r.f8 && !r.d_w0 && o.in.dtype == YL_BF16 && o.out.dtype == YL_BF16 && o.out8.offset < 0;
        if (o.kind == YL_OP_CONV) {
          ok = ok && nd == 0 && o.k == 3 && o.stride == 1 && o.res.offset < 0 && o.in2.offset < 0 && r.cfg.MT == 4 && r.cfg.uniform &&
               r.cfg.ncb == 1 && (r.cfg.post != 0) == (e->ops[i].cfg.post != 0) && (o.dfl_level != 0) == (e->ops[i].op.dfl_level != 0);
          ++nc;
        } else if (o.kind == YL_OP_DWCONV3) {
          const OpRec& d0 = e->ops[i + nc];
          ok = ok && r.d_wpk && r.cfg.MT == 5 && (r.d_wpk3 != nullptr) == (d0.d_wpk3 != nullptr);
          ++nd;
        } else {
          ok = false;
        }
      }
      ok = ok && nc <= 3 && nd <= 3;
      if (ok && nc && nd)  // mixed launch: the convs carry a fused 1x1 exactly when the depthwise members carry a tail
        ok = (e->ops[i].cfg.post != 0) == (e->ops[i + nc].d_wpk3 != nullptr);
      // (a depthwise-only group without the tail: the class branch's first halves of levels 0 and 1 -- opt-in from the lowering,
      // YL_GROUP_CLS0=1; dwpw_group_launch takes both forms)
      // Two plain convs with a deep K and 64 output channels on small maps (Detect's first box convs on the 40x40 and 20x20 levels,
      // 128 -> 64 and 256 -> 64): conv3_band_kernel -- the input band in LDS, the filter streamed from L2 -- instead of
      // conv_group_kernel, which restages the 144 / 288 KB filters per 256 pixels (YL_CONV_BAND=0: off)
      const int band_on = env_int("YL_CONV_BAND", 1);  // (read at every engine build: tests switch it)
      bool band = ok && band_on && nc == 2 && nd == 0 && !e->ops[i].cfg.post;
      for (size_t k = i; k < j && band; ++k) {
        const OpRec& r = e->ops[k];
        band = r.cfg.MT == 4 && r.cfg.ncb == 1 && !r.cfg.kord && r.op.act && r.op.out.C == 64 &&
               r.op.in.C == (k == i ? 128 : 256) && r.op.in.H == r.op.out.H && r.op.in.W == r.op.out.W &&
               conv3_band_plan(r.op.in.C, 64, r.op.out.H, r.op.out.W, nullptr);
      }
      if (band) {
        for (size_t k = i; k < j; ++k) {
          OpRec& r = e->ops[k];
          r.group_n = k == i ? n : -1;
          r.band = k == i;
          r.name = k == i ? "conv3x3_band_mfma[x2]" : "(in the group launch)";
        }
        ok = false;  // (handled)
      }
      // The levels' second box convs with the fused 1x1 + box decode (cv2[i][1] + cv2[i][2], 64 -> 64 -> 64 DFL logits -> boxes) on
      // the Cout-split filter-in-registers kernel, ONE launch for the levels (conv3_csg_kernel, round 6): conv_group_kernel restages
      // each 72 KB filter per 256 pixels and took 65 us in sequence for the three; this form 4x us (profiles/r06b_*)
      if (ok && nc >= 2 && nd == 0 && e->ops[i].cfg.post && e->ops[i].op.dfl_level) {
        ConvConfig cc3[3];
        int hh[3], ww[3];
        bool fit = nc <= 3;
        for (int k = 0; k < nc && fit; ++k) {
          const OpRec& r = e->ops[i + k];
          fit = r.op.dfl_level && r.op.in.C == 64 && r.op.post_cin == 64 && r.op.out.C == 64 && r.op.k == 3 && !r.op.post_act && r.op.act &&
                r.op.in.H == r.op.out.H && r.op.in.W == r.op.out.W;
          cc3[k] = r.cfg;
          hh[k] = r.op.out.H;
          ww[k] = r.op.out.W;
        }
        if (fit && conv3_cs_group_fits(cc3, nc, batch, hh, ww)) {
          static const char* const gn[3] = {"conv3x3_cs+1x1+dfl_mfma[x1]", "conv3x3_cs+1x1+dfl_mfma[x2]", "conv3x3_cs+1x1+dfl_mfma[x3]"};
          for (size_t k = i; k < j; ++k) {
            OpRec& r = e->ops[k];
            r.group_n = k == i ? n : -1;
            r.csg = k == i;
            r.name = k == i ? gn[n - 1] : "(in the group launch)";
          }
          ok = false;  // (handled)
        }
      }
      if (ok) {
        // one pixel-tile count per kind: the largest member's choice (it has most of the workgroups)
        auto largest = [&](size_t k0, size_t k1) {
          size_t best = k0;
          for (size_t k = k0; k < k1; ++k)
            if ((long long)e->ops[k].op.out.H * e->ops[k].op.out.W > (long long)e->ops[best].op.out.H * e->ops[best].op.out.W) best = k;
          return best;
        };
        if (nc) {
          const OpRec& big = e->ops[largest(i, i + nc)];
          int NT = big.cfg.reg ? big.cfg.reg : big.cfg.NT;
          if (NT != 2 && NT != 4) NT = 2;
          // a launch shared with depthwise members: 32 pixels per wave (151-156 registers, three waves per SIMD -- the
          // depthwise bodies run at the conv's register budget; 64 pixels: 187-220 registers, -1.3 % end to end)
          static const int head_nt = tune_int("YL_HEAD_GROUP_NT", 2);
          if (nd && (head_nt == 2 || head_nt == 4)) NT = head_nt;
          for (size_t k = i; k < i + nc; ++k) conv_retile_for_group(&e->ops[k].cfg, NT, batch, e->ops[k].op.out.H, e->ops[k].op.out.W);
        }
        if (nd) {
          int NT = e->ops[largest(i + nc, j)].cfg.NT;
          if (NT > 2) NT = 2;
          for (size_t k = i + nc; k < j; ++k) {
            OpRec& r = e->ops[k];
            r.cfg.NT = NT;
            r.cfg.npatches = (int)(((long long)batch * r.op.out.H * r.op.out.W + 16 * NT - 1) / (16 * NT));
          }
        }
        static const char* const cn[2][3] = {{"conv3x3_mfma[x1]", "conv3x3_mfma[x2]", "conv3x3_mfma[x3]"},
                                             {"conv3x3+1x1_mfma[x1]", "conv3x3+1x1_mfma[x2]", "conv3x3+1x1_mfma[x3]"}};
        static const char* const dn[3] = {"dw3x3+pw1x1+1x1_mfma[x1]", "dw3x3+pw1x1+1x1_mfma[x2]", "dw3x3+pw1x1+1x1_mfma[x3]"};
        static const char* const dn0[3] = {"dw3x3+pw1x1_mfma[x1]", "dw3x3+pw1x1_mfma[x2]", "dw3x3+pw1x1_mfma[x3]"};
        static const char* const hn[2] = {"head:conv3x3|dw3x3+pw1x1_mfma", "head:conv3x3+1x1|dw3x3+pw1x1+1x1_mfma"};
        for (size_t k = i; k < j; ++k) {
          OpRec& r = e->ops[k];
          r.group_n = k == i ? n : -1;
          r.name = k != i ? "(in the group launch)" : (nc && nd) ? hn[r.cfg.post ? 1 : 0] : nc ? cn[r.cfg.post ? 1 : 0][nc - 1] : (e->ops[i].d_wpk3 ? dn[nd - 1] : dn0[nd - 1]);
        }
      }
    }
    i = j;
  }
  *out = e.release();
  return 0;
}

// ConvArgs of a plain (bf16, not the fused stem) YL_OP_CONV for workspace `ws`
static void build_conv_args(const yl_engine* e, const OpRec& r, char* ws, ConvArgs& a) {
  const yl_op& o = r.op;
  const int B = e->batch;
  const Geo gi = geo_of(B, o.in.H, o.in.W), go = geo_of(B, o.out.H, o.out.W);
  a.in = pview(ws, o.in, B).base;
  a.in_plane = (unsigned)plane_bytes(gi);
  a.in_w1 = gi.W1;
  a.in_img = gi.img;
  a.in_g0 = gi.g0;
  a.out_f32 = o.out.dtype == YL_F32;
  if (a.out_f32) {
    a.out = ws + o.out.offset;
    a.out_pitch = o.out.pitch;
    a.out_coff = o.out.coff;
  } else {
    a.out = pview(ws, o.out, B).base;
    a.out_plane = (unsigned)plane_bytes(go);
  }
  a.out_w1 = go.W1;
  a.out_img = go.img;
  a.out_g0 = go.g0;
  if (o.res.offset >= 0) {
    a.res = pview(ws, o.res, B).base;
    a.res_plane = (unsigned)plane_bytes(go);
  }
  if (o.in2.offset >= 0) {
    const Geo g2 = geo_of(B, o.in2.H, o.in2.W);
    a.in2 = pview(ws, o.in2, B).base;
    a.in2_plane = (unsigned)plane_bytes(g2);
    a.in2_w1 = g2.W1;
    a.in2_img = g2.img;
    a.in2_g0 = g2.g0;
    a.ks_split = o.in2.C / 32;
  }
  a.wpk = r.d_wpk;
  a.bias = r.d_bias;
  a.wpk2 = r.d_wpk2;
  a.bias2 = r.d_bias2;
  a.post_cout = o.out.C;
  a.post_act = o.post_act;
  a.Ho = o.out.H;
  a.Wo = o.out.W;
  a.Cin = o.in.C + (o.in2.offset >= 0 ? o.in2.C : 0);
  a.Cout = r.d_wpk2 ? o.post_cin : o.out.C;
  a.stride = o.stride;
  a.act = o.act;
  if (o.dfl_level) {
    const yl_level& L = e->levels[o.dfl_level - 1];
    a.dfl_out = reinterpret_cast<float*>(ws + L.offset_xywh);
    a.dfl_stride = L.stride;
    a.dfl_keep = o.dfl_keep;
  }
  fill_out8(a, o, ws, B);
}

// ConvArgs + tiling of a fused depthwise + pointwise [+ 1x1] YL_OP_DWCONV3 for workspace `ws`
static void build_dwpw_args(const yl_engine* e, const OpRec& r, char* ws, ConvArgs& a, ConvConfig& cfg,
                            const PrefilterSink* sink = nullptr) {
  const yl_op& o = r.op;
  const int B = e->batch;
  if (sink && r.cls_level >= 0) {  // this launch writes a level's class logits: pre-filter them on the way out
    int a0 = 0;
    for (int li = 0; li < r.cls_level; ++li) a0 += e->levels[li].H * e->levels[li].W;
    a.pf_count = sink->pre_count;
    a.pf_list = sink->pre_list;
    a.pf_sub_cap = sink->sub_cap;
    a.pf_logit_lo = sink->logit_lo;
    a.pf_anchor0 = a0;
  }
  const Geo gi = geo_of(B, o.in.H, o.in.W);
  a.in = pview(ws, o.in, B).base;
  a.in_plane = (unsigned)plane_bytes(gi);
  a.in_w1 = gi.W1;
  a.in_img = gi.img;
  a.in_g0 = gi.g0;
  a.out = pview(ws, o.out, B).base;
  a.out_plane = a.in_plane;
  a.out_w1 = gi.W1;
  a.out_img = gi.img;
  a.out_g0 = gi.g0;
  a.wpk = r.d_wpk;
  a.bias = r.d_bias2;
  a.dw_wpk = r.d_dwpk;
  a.dw_b = r.d_bias;
  a.dw_act = o.act;
  a.Ho = o.out.H;
  a.Wo = o.out.W;
  a.Cin = o.in.C;
  a.Cout = o.out.C;
  a.stride = 1;
  a.act = o.post_act;
  cfg = r.cfg;
  if (r.d_wpk3) {  // a further 1x1 behind the pointwise conv
    a.Cout = o.reserved2;
    a.wpk2 = r.d_wpk3;
    a.bias2 = r.d_bias3;
    a.post_cout = o.out.C;
    a.post_act = o.stem_act;
    a.cg2 = o.reserved2 / 8;
    a.ksteps2 = r.cfg.kchunk;
    cfg.kchunk = cfg.ksteps;
  }
}

static int run_op(const yl_engine* e, const OpRec& r, char* ws, hipStream_t s, const void* input = nullptr,
                  int in_dtype = 0, const PrefilterSink* sink = nullptr) {
  const yl_op& o = r.op;
  const int B = e->batch;
  switch (o.kind) {
    case YL_OP_CONV: {
      if (r.d_w0) {  // layers 0 + 1 in one launch
        if (!input) return fail(-1, "run_op: the fused stem needs the network input");
        Stem2Args q{};
        q.in = input;
        q.B = B;
        q.H = e->in_h;
        q.W = e->in_w;
        q.H0 = o.in.H;
        q.W0 = o.in.W;
        q.C0 = o.in.C;
        q.H1 = o.out.H;
        q.W1 = o.out.W;
        q.w0 = r.d_w0;
        q.b0 = r.d_b0;
        q.act0 = o.stem_act;
        q.wpk = r.d_wpk;
        q.bias = r.d_bias;
        q.ksteps = r.cfg.ksteps;
        q.Cout = r.d_wpk2 ? o.post_cin : o.out.C;
        q.act = o.act;
        q.wpk2 = r.d_wpk2;
        q.bias2 = r.d_bias2;
        q.post_cout = o.out.C;
        q.post_act = o.post_act;
        q.out = pview(ws, o.out, B);
        const long long ob = (long long)((o.out.C + 7) / 8) * plane_bytes(geo_of(B, o.out.H, o.out.W));
        if (ob > (1LL << 30)) return fail(-3, "stem2: the output slice exceeds 1 GiB; lower the batch size");
        q.out_bytes = (unsigned)ob;
        q.cacheable = (o.reserved3 & 0x100) ? 1 : 0;
        if (sink && sink->zero_ptr) {  // (detect_eager: this launch clears the step's NMS candidate counters on its way)
          q.zero_ptr = sink->zero_ptr;
          q.zero_n = sink->zero_n;
        }
        return stem2_launch(q, r.cfg.MT, in_dtype, s);
      }
      ConvArgs a{};
      const Geo gi = geo_of(B, o.in.H, o.in.W), go = geo_of(B, o.out.H, o.out.W);
      if (r.f8) {
        a.in = pview(ws, o.in, B).base;
        a.in_plane = (unsigned)plane_bytes(gi);
        a.in_w1 = gi.W1;
        a.in_img = gi.img;
        a.in_g0 = gi.g0;
        a.out = pview(ws, o.out, B).base;
        a.out_plane = (unsigned)plane_bytes(go);
        a.out_w1 = go.W1;
        a.out_img = go.img;
        a.out_g0 = go.g0;
        a.out_f8 = o.out.dtype == YL_FP8;
        a.out_inv_scale = a.out_f8 ? 1.0f / o.out_scale : 1.0f;
        if (o.res.offset >= 0) {
          a.res = pview(ws, o.res, B).base;
          a.res_plane = (unsigned)plane_bytes(go);
          a.res_f8 = o.res.dtype == YL_FP8;
          a.res_scale = o.res_scale;
        }
        a.wpk = r.d_wpk;
        a.alpha = r.d_alpha;
        a.bias = r.d_bias;
        a.Ho = o.out.H;
        a.Wo = o.out.W;
        a.Cin = o.in.C;
        a.Cout = o.out.C;
        a.stride = o.stride;
        a.act = o.act;
        fill_out8(a, o, ws, B);
        return conv_f8_launch(r.cfg, a, B, s);
      }
      if (r.band1) {
        const Geo gq = geo_of(B, o.out.H, o.out.W);
        BandArgs q{};
        q.x = pview(ws, o.in, B).base;
        q.y = pview(ws, o.out, B).base;
        q.x_plane = q.y_plane = (unsigned)plane_bytes(gq);
        const long long ib = (long long)(o.in.C / 8) * plane_bytes(gq), ob = (long long)(o.out.C / 8) * plane_bytes(gq);
        if (ib > (1LL << 30) || ob > (1LL << 30)) return fail(-3, "conv3_band: a tensor slice exceeds 1 GiB; lower the batch size");
        q.x_bytes = (unsigned)ib;
        q.y_bytes = (unsigned)ob;
        q.H = o.out.H;
        q.W = o.out.W;
        q.W1 = gq.W1;
        q.img = gq.img;
        q.g0 = gq.g0;
        q.Cin = o.in.C;
        q.act = o.act;
        q.w = r.d_wpk;
        q.bias = r.d_bias;
        return conv3_band1_launch(q, B, s);
      }
      build_conv_args(e, r, ws, a);
      if (sink && r.cls_level >= 0) {  // a plain 1x1 launch that writes a level's class logits: pre-filter them on the way out
        int a0 = 0;
        for (int li = 0; li < r.cls_level; ++li) a0 += e->levels[li].H * e->levels[li].W;
        a.pf_count = sink->pre_count;
        a.pf_list = sink->pre_list;
        a.pf_sub_cap = sink->sub_cap;
        a.pf_logit_lo = sink->logit_lo;
        a.pf_anchor0 = a0;
      }
      return conv_launch(r.cfg, a, B, s);
    }
    case YL_OP_QUANT: {
      const Geo gq = geo_of(B, o.in.H, o.in.W);
      return quant_f8_launch(pview(ws, o.in, B), pview(ws, o.out, B).base, gq.plane_px, o.in.C, o.out_scale, s);
    }
    case YL_OP_BNECK2: {
      Bneck2Args q{};
      q.xin = pview(ws, o.in, B);
      q.yout = pview(ws, o.out, B);
      q.B = B;
      q.H = o.out.H;
      q.W = o.out.W;
      q.c = o.in.C;
      q.w1 = r.d_wpk;
      q.b1 = r.d_bias;
      q.w2 = r.d_wpk2;
      q.b2 = r.d_bias2;
      const Geo gq = geo_of(B, o.out.H, o.out.W);
      const long long nb = (long long)(q.c / 8) * plane_bytes(gq);
      if (nb > (1LL << 30)) return fail(-3, "bottleneck: a tensor slice exceeds 1 GiB; lower the batch size");
      q.in_bytes = q.out_bytes = (unsigned)nb;
      q.plane_px = gq.plane_px;
      return bneck2_launch(q, s);
    }
    case YL_OP_C3K: {
      C3kArgs q{};
      const Geo gq = geo_of(B, o.out.H, o.out.W);
      const long long nb = (long long)(o.in.C / 8) * plane_bytes(gq);
      if (nb > (1LL << 30)) return fail(-3, "c3k: a tensor slice exceeds 1 GiB; lower the batch size");
      q.x = pview(ws, o.in, B).base;
      q.y = pview(ws, o.out, B).base;
      q.x_plane = q.y_plane = (unsigned)plane_bytes(gq);
      q.x_bytes = q.y_bytes = (unsigned)nb;
      q.B = B;
      q.H = o.out.H;
      q.W = o.out.W;
      q.W1 = gq.W1;
      q.img = gq.img;
      q.g0 = gq.g0;
      for (int k = 0; k < 6; ++k) {
        q.w[k] = r.d_xw[k];
        q.bias[k] = r.d_xb[k];
      }
      q.split = o.reserved3;  // (workgroups per image asked for by the program: 2, or 0 / 1)
      return c3k_resident_launch(q, s);
    }
    case YL_OP_SPPF: {
      SppfArgs q{};
      const Geo gq = geo_of(B, o.out.H, o.out.W);
      const long long ib = (long long)(o.in.C / 8) * plane_bytes(gq), ob = (long long)(o.out.C / 8) * plane_bytes(gq);
      if (ib > (1LL << 30) || ob > (1LL << 30)) return fail(-3, "sppf: a tensor slice exceeds 1 GiB; lower the batch size");
      q.x = pview(ws, o.in, B).base;
      q.y = pview(ws, o.out, B).base;
      q.x_plane = q.y_plane = (unsigned)plane_bytes(gq);
      q.x_bytes = (unsigned)ib;
      q.y_bytes = (unsigned)ob;
      q.B = B;
      q.H = o.out.H;
      q.W = o.out.W;
      q.W1 = gq.W1;
      q.img = gq.img;
      q.g0 = gq.g0;
      q.w1 = r.d_xw[0];
      q.w2 = r.d_xw[1];
      q.w3 = r.d_xw[2];
      q.b1 = r.d_xb[0];
      q.b2 = r.d_xb[1];
      q.b3 = r.d_xb[2];
      return sppf_resident_launch(q, s);
    }
    case YL_OP_PSA: {
      PsaArgs q{};
      const Geo gq = geo_of(B, o.in.H, o.in.W);
      const long long nb = (long long)(o.in.C / 16) * plane_bytes(gq);  // bytes of ONE half (a or b)
      if (2 * nb > (1LL << 30)) return fail(-3, "psa: a tensor slice exceeds 1 GiB; lower the batch size");
      q.a_in = pview(ws, o.in, B).base;
      q.a_out = pview(ws, o.out, B).base;
      q.b = reinterpret_cast<const __bf16*>(reinterpret_cast<const char*>(q.a_in) + nb);
      q.b_out = reinterpret_cast<__bf16*>(reinterpret_cast<char*>(q.a_out) + nb);
      q.b_plane = (unsigned)plane_bytes(gq);
      q.b_bytes = (unsigned)nb;
      q.B = B;
      q.H = o.in.H;
      q.W = o.in.W;
      q.W1 = gq.W1;
      q.img = gq.img;
      q.g0 = gq.g0;
      q.scale = o.scale;
      q.w_qkv = r.d_xw[0];
      q.w_tail = r.d_xw[1];
      q.b_qkv = r.d_xb[0];
      q.b_pe = r.d_xb[1];
      q.b_proj = r.d_xb[2];
      q.b_ffn0 = r.d_xb[3];
      q.b_ffn1 = r.d_xb[4];
      q.w_pe = r.d_w;
      q.split = o.reserved3;  // (workgroups per image asked for by the program; 0: psa_resident_launch's choice)
      return psa_resident_launch(q, s);
    }
    case YL_OP_PSA_TAIL: {
      PsaTailArgs q{};
      const Geo gq = geo_of(B, o.in.H, o.in.W);
      const long long pb = plane_bytes(gq);
      if (32 * pb > (1LL << 30)) return fail(-3, "psa tail: a tensor slice exceeds 1 GiB; lower the batch size");
      q.o = pview(ws, o.in, B).base;
      q.u = pview(ws, o.res, B).base;
      q.y = pview(ws, o.out, B).base;
      q.o_plane = q.u_plane = q.y_plane = (unsigned)pb;
      q.o_bytes = (unsigned)(16 * pb);
      q.u_bytes = q.y_bytes = (unsigned)(32 * pb);
      q.B = B;
      q.H = o.in.H;
      q.W = o.in.W;
      q.W1 = gq.W1;
      q.img = gq.img;
      q.g0 = gq.g0;
      q.w_tail = r.d_xw[1];
      q.b_proj = r.d_xb[0];
      q.b_ffn0 = r.d_xb[1];
      q.b_ffn1 = r.d_xb[2];
      q.b_cv2 = r.d_xb[3];
      q.act = o.act;
      return psa_tail_launch(q, s);
    }
    case YL_OP_C3K2S: {
      C3k2sArgs q{};
      q.tin = pview(ws, o.in, B);
      q.yout = pview(ws, o.out, B);
      q.B = B;
      q.H = o.out.H;
      q.W = o.out.W;
      q.c = o.post_cin * 2 / 9;
      q.cout = o.out.C;
      q.wA = r.d_wpk;
      q.bA = r.d_bias;
      q.wB = r.d_wpk2;
      q.bB = r.d_bias2;
      q.wC = r.d_wpk3;
      q.bC = r.d_bias3;
      const Geo gq = geo_of(B, o.out.H, o.out.W);
      const long long ib = (long long)(q.c / 4) * plane_bytes(gq), ob = (long long)(o.out.C / 8) * plane_bytes(gq);
      if (ib > (1LL << 30) || ob > (1LL << 30)) return fail(-3, "c3k2: a tensor slice exceeds 1 GiB; lower the batch size");
      q.in_bytes = (unsigned)ib;
      q.out_bytes = (unsigned)ob;
      q.cacheable = (o.reserved3 & 0x100) ? 1 : 0;
      q.plane_px = gq.plane_px;
      return c3k2s_launch(q, s);
    }
    case YL_OP_STEM: {
      if (!input) return fail(-1, "run_op: the stem needs the network input");
      StemArgs a{};
      a.in = input;
      a.out = pview(ws, o.out, B);
      a.w = r.d_w;
      a.wpk = r.d_wpk;
      a.bias = r.d_bias;
      a.B = B;
      a.H = e->in_h;
      a.W = e->in_w;
      a.Ho = o.out.H;
      a.Wo = o.out.W;
      a.Cout = o.out.C;
      a.act = o.act;
      if (o.out8.offset >= 0) {
        const Geo g8 = geo_of(B, o.out8.H, o.out8.W);
        a.out8 = reinterpret_cast<unsigned char*>(ws + o.out8.offset + (int64_t)(o.out8.coff / 16) * plane_bytes(g8));
        a.out8_plane = plane_bytes(g8);
        a.out8_inv_scale = 1.0f / o.out8_scale;
        a.out8_only = o.out8_only;
      }
      return stem_launch(a, in_dtype, s);
    }
    case YL_OP_DWCONV3: {
      if (r.d_wpk) {  // fused depthwise + pointwise
        ConvArgs a{};
        ConvConfig cfg;
        build_dwpw_args(e, r, ws, a, cfg, sink);
        return dwpw_launch(cfg, a, B, s);
      }
      DwArgs a{};
      a.in = pview(ws, o.in, B);
      a.out = pview(ws, o.out, B);
      a.w = r.d_w;
      a.bias = r.d_bias;
      a.H = o.out.H;
      a.W = o.out.W;
      a.C = o.out.C;
      a.act = o.act;
      a.accumulate = o.accumulate;
      a.grp = o.grp;
      a.grp_stride = o.grp_stride;
      a.grp_off = o.grp_off;
      a.batch = B;
      return dwconv3_launch(a, s);
    }
    case YL_OP_SPPF_POOL: {
      PoolArgs a{};
      a.v = pview(ws, o.in, B);
      a.H = o.in.H;
      a.W = o.in.W;
      a.C = o.in.C;
      a.batch = B;
      return sppf_pool_launch(a, s);
    }
    case YL_OP_UPSAMPLE: {
      UpArgs a{};
      a.in = pview(ws, o.in, B);
      a.out = pview(ws, o.out, B);
      a.H = o.in.H;
      a.W = o.in.W;
      a.C = o.in.C;
      a.factor = o.stride;
      a.batch = B;
      return upsample_launch(a, s);
    }
    case YL_OP_ATTENTION: {
      AttnArgs a{};
      a.in = pview(ws, o.in, B);
      a.out = pview(ws, o.out, B);
      a.H = o.in.H;
      a.W = o.in.W;
      a.N = o.in.H * o.in.W;
      a.heads = o.heads;
      a.batch = B;
      a.scale = o.scale;
      return attention_launch(a, s);
    }
  }
  return fail(-1, "run_op: unknown kind");
}

// A group launch (yl_op.group): ops [i, i + n) are the same conv on the Detect levels (reference nn/modules/head.py:59-62)
// and / or the independent branches of the head (head.py:61), one launch for all of them (conv_group_kernel /
// dwpw_group_kernel / head_group_kernel).
static int run_group(const yl_engine* e, size_t i, int n, char* ws, hipStream_t s, const PrefilterSink* sink) {
  if (e->ops[i].band) {
    BandArgs ba[3]{};
    for (int k = 0; k < n && k < 3; ++k) {
      const OpRec& r = e->ops[i + k];
      const yl_op& o = r.op;
      const Geo gq = geo_of(e->batch, o.out.H, o.out.W);
      BandArgs& q = ba[k];
      q.x = pview(ws, o.in, e->batch).base;
      q.y = pview(ws, o.out, e->batch).base;
      q.x_plane = q.y_plane = (unsigned)plane_bytes(gq);
      const long long ib = (long long)(o.in.C / 8) * plane_bytes(gq), ob = (long long)(o.out.C / 8) * plane_bytes(gq);
      if (ib > (1LL << 30) || ob > (1LL << 30)) return fail(-3, "conv3_band: a tensor slice exceeds 1 GiB; lower the batch size");
      q.x_bytes = (unsigned)ib;
      q.y_bytes = (unsigned)ob;
      q.H = o.out.H;
      q.W = o.out.W;
      q.W1 = gq.W1;
      q.img = gq.img;
      q.g0 = gq.g0;
      q.Cin = o.in.C;
      q.act = o.act;
      q.w = r.d_wpk;
      q.bias = r.d_bias;
      q.w2 = r.d_wpk2;
      q.bias2 = r.d_bias2;
      q.post_act = o.post_act;
    }
    return conv3_band_launch(ba, n, e->batch, s);
  }
  if (e->ops[i].csg) {
    ConvArgs ga[3]{};
    for (int k = 0; k < n && k < 3; ++k) build_conv_args(e, e->ops[i + k], ws, ga[k]);
    return conv3_cs_group_launch(ga, n, e->batch, s);
  }
  ConvArgs ca[3]{}, da[3]{};
  ConvConfig cc[3], dc[3];
  int nc = 0, nd = 0;
  for (int k = 0; k < n; ++k) {
    const OpRec& r = e->ops[i + k];
    if (r.op.kind == YL_OP_DWCONV3) {
      build_dwpw_args(e, r, ws, da[nd], dc[nd], sink);
      dc[nd].NT = r.cfg.NT;  // (the group's tiling: build_dwpw_args copies r.cfg, which already carries it)
      ++nd;
    } else {
      build_conv_args(e, r, ws, ca[nc]);
      cc[nc] = r.cfg;
      ++nc;
    }
  }
  if (nc && nd) return head_group_launch(cc, ca, nc, dc, da, nd, e->batch, s);
  return nd ? dwpw_group_launch(dc, da, nd, e->batch, s) : conv_group_launch(cc, ca, nc, e->batch, s);
}

// op i or the group that starts at it; *consumed = ops covered (members of a group other than the first are never passed in)
static int run_at(const yl_engine* e, size_t i, char* ws, hipStream_t s, const void* input, int in_dtype, int* consumed,
                  const PrefilterSink* sink = nullptr) {
  const OpRec& r = e->ops[i];
  if (r.group_n >= 2) {
    *consumed = r.group_n;
    return run_group(e, i, r.group_n, ws, s, sink);
  }
  *consumed = 1;
  return run_op(e, r, ws, s, input, in_dtype, sink);
}

static int run_convert(const yl_engine* e, const void* input, int in_dtype, char* ws, hipStream_t s) {
  return convert_input_launch(input, in_dtype, pview(ws, e->input_view, e->batch), e->batch, e->in_h, e->in_w, s);
}

// The guard pixels of the planar layout must read as zero.  Nothing ever writes them, so clearing a
// workspace once, the first time this engine sees its address, is enough.
static int prepare_workspace(yl_engine* e, char* ws, hipStream_t s) {
  for (void* p : e->zeroed)
    if (p == ws) return 0;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &st);
  if (st != hipStreamCaptureStatusNone) return fail(-1, "workspace must be prepared before stream capture");
  YL_HIP(hipMemsetAsync(ws, 0, (size_t)e->ws_bytes, s));
  e->zeroed.push_back(ws);
  return 0;
}

static int run_decode(const yl_engine* e, const char* ws, float* y, hipStream_t s) {
  DecodeLevel lv[8];
  int a0 = 0;
  for (size_t i = 0; i < e->levels.size(); ++i) {
    lv[i] = decode_level(ws, e->levels[i], e->batch, a0);
    a0 += lv[i].H * lv[i].W;
  }
  return decode_launch(lv, (int)e->levels.size(), e->batch, e->nc, e->A, y, s);
}

// Side lanes: an op with lane k > 0 is launched on its own stream.  A fork event recorded on the
// main stream at the op's program position makes it start only after EVERY earlier op (main lane
// in order; earlier ops of the same lane by stream order), which covers its data dependencies and
// the recycling of arena memory; it then overlaps with later main-lane ops.  All lanes are joined
// back into the main stream before the decode.  The same calls are legal under stream capture, where
// they become fork/join edges of the graph.
static int lane_resources(yl_engine* e, yl_engine::StreamSet& ss, size_t n_events) {
  while ((int)ss.lanes.size() <= e->max_lane) ss.lanes.push_back(nullptr);
  // (measured, round 5: neither the lowest stream priority nor a CU mask on the side lanes keeps the neck's small launches from
  // slowing down beside the large Detect branches -- priority changes nothing, a CU-masked stream serialises with the caller's)
  for (int k = 1; k <= e->max_lane; ++k)
    if (!ss.lanes[k]) YL_HIP(hipStreamCreateWithFlags(&ss.lanes[k], hipStreamNonBlocking));
  while (ss.events.size() < n_events) {
    hipEvent_t ev;
    YL_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    ss.events.push_back(ev);
  }
  return 0;
}

static int capture_stream(yl_engine* e, void* workspace, hipStream_t* out) {
  yl_engine::StreamSet& ss = e->sets[workspace];
  if (!ss.cap) YL_HIP(hipStreamCreateWithFlags(&ss.cap, hipStreamNonBlocking));
  *out = ss.cap;
  return 0;
}

extern "C" int yl_engine_prepare_workspace(yl_engine* e, void* workspace, void* stream) {
  if (!e || !workspace) return fail(-1, "yl_engine_prepare_workspace: null pointer");
  return prepare_workspace(e, (char*)workspace, (hipStream_t)stream);
}

static int forward_impl(yl_engine* e, const void* input, int in_dtype, void* workspace, float* y, void* stream,
                        const PrefilterSink* sink);

extern "C" int yl_engine_forward(yl_engine* e, const void* input, int in_dtype, void* workspace, float* y,
                                 void* stream) {
  return forward_impl(e, input, in_dtype, workspace, y, stream, nullptr);
}

// sink != nullptr: the launches that write the Detect levels' class logits also fill the NMS candidate lists
static int forward_impl(yl_engine* e, const void* input, int in_dtype, void* workspace, float* y, void* stream,
                        const PrefilterSink* sink) {
  if (!e || !input || !workspace) return fail(-1, "yl_engine_forward: null pointer");
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  if (int rc = prepare_workspace(e, ws, s)) return rc;
  yl_engine::StreamSet& ss = e->sets[workspace];
  if (e->max_lane > 0)
    if (int rc = lane_resources(e, ss, e->ops.size() + e->max_lane + 1)) return rc;
  const bool stem = !e->ops.empty() && (e->ops[0].op.kind == YL_OP_STEM || e->ops[0].d_w0);
  if (!stem)
    if (int rc = run_convert(e, input, in_dtype, ws, s)) return rc;
  size_t nev = 0;
  std::vector<char> used((size_t)e->max_lane + 1, 0);
  for (size_t oi = 0; oi < e->ops.size(); ++oi) {
    const OpRec& r = e->ops[oi];
    if (r.group_n < 0) continue;  // launched with the first member of its group
    const int lane = r.op.lane;
    if (lane == 0) {
      int consumed = 1;
      static const bool trace = getenv("YL_TRACE_OPS") != nullptr;  // development: name every eager launch and wait for it
      if (trace) fprintf(stderr, "[yl] op %zu kind %d (%s)\n", oi, (int)r.op.kind, r.name);
      if (int rc = run_at(e, oi, ws, s, input, in_dtype, &consumed, sink)) return rc;
      if (trace) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &cs);
        if (cs == hipStreamCaptureStatusNone) {
          YL_HIP(hipStreamSynchronize(s));
          fprintf(stderr, "[yl] op %zu done\n", oi);
        }
      }
    } else {
      hipEvent_t ev = ss.events[nev++];
      YL_HIP(hipEventRecord(ev, s));
      YL_HIP(hipStreamWaitEvent(ss.lanes[lane], ev, 0));
      // (side ops never read the network input; with lane graphs the class branch of a level may run here and feed the sink)
      if (int rc = run_op(e, r, ws, ss.lanes[lane], nullptr, in_dtype, e->lane_graphs ? sink : nullptr)) return rc;
      used[lane] = 1;
    }
  }
  for (int k = 1; k <= e->max_lane; ++k)
    if (used[k]) {
      hipEvent_t ev = ss.events[nev++];
      YL_HIP(hipEventRecord(ev, ss.lanes[k]));
      YL_HIP(hipStreamWaitEvent(s, ev, 0));
    }
  if (y) {
    if (e->levels.empty()) return fail(-1, "yl_engine_forward: engine has no Detect levels to decode");
    if (int rc = run_decode(e, ws, y, s)) return rc;
  }
  return 0;
}

extern "C" int yl_engine_forward_graph(yl_engine* e, const void* input, int in_dtype, void* workspace, float* y,
                                       void* stream) {
  if (!e || !input || !workspace) return fail(-1, "yl_engine_forward_graph: null pointer");
  hipStream_t s = (hipStream_t)stream;
  yl_engine::GraphKey key{input, in_dtype, workspace, y};
  hipGraphExec_t exec = e->graphs.find(key);
  if (!exec) {
    // run once eagerly so lazy one-time setup (function attributes) happens outside the capture
    if (int rc = yl_engine_forward(e, input, in_dtype, workspace, y, stream)) return rc;
    YL_HIP(hipStreamSynchronize(s));
    // Capture on a private stream (the caller's may be the legacy default stream, which cannot
    // capture); the instantiated graph is then launched on the caller's stream.
    hipStream_t cap = nullptr;
    if (int rc = capture_stream(e, workspace, &cap)) return rc;
    hipGraph_t g = nullptr;
    YL_HIP(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
    int rc = yl_engine_forward(e, input, in_dtype, workspace, y, (void*)cap);
    hipError_t ce = hipStreamEndCapture(cap, &g);
    if (rc) {
      if (g) (void)hipGraphDestroy(g);
      return rc;
    }
    if (ce != hipSuccess) return fail(-2, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
    hipGraphExec_t ge = nullptr;
    hipError_t ie = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ie != hipSuccess) return fail(-2, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie));
    e->graphs.insert(key, ge);
    exec = ge;
  }
  YL_HIP(hipGraphLaunch(exec, s));
  return 0;
}

static int detect_eager(yl_engine* e, const void* input, int in_dtype, void* workspace, float conf_thres,
                        float iou_thres, int max_det, int agnostic, const int32_t* classes, int n_classes,
                        int max_cand, float max_wh, const float* scale, float* out, int32_t* out_count, void* work,
                        void* stream) {
  // The candidate pre-filter of yl_decode_nms rides in the epilogue of the launches that write the class logits
  // (ConvArgs::pf_*) when every level has such a launch and the sparse decode path applies: the counters are zeroed up
  // front, the network fills the lists, and the decode starts at its second stage.
  PrefilterSink sink{};
  int fused = 0;
  // (the counters are cleared by the network's first launch where that is the fused stem -- it precedes every class-logit
  // launch in stream order; otherwise by a launch of their own)
  static const bool zero_in_stem = env_int("YL_ZERO_IN_STEM", 1) != 0;
  const bool defer = zero_in_stem && !e->ops.empty() && e->ops[0].d_w0 && e->ops[0].op.lane == 0 && e->ops[0].group_n >= 0;
  if (e->cls_tails)
    if (int rc = prefilter_sink_prepare(e->levels.data(), (int)e->levels.size(), e->batch, max_cand, conf_thres, work, stream, &sink,
                                        &fused, defer))
      return rc;
  if (int rc = forward_impl(e, input, in_dtype, workspace, nullptr, stream, fused ? &sink : nullptr)) return rc;
  return decode_nms_stages(workspace, e->levels.data(), (int)e->levels.size(), e->batch, e->nc, conf_thres, iou_thres, max_det,
                           agnostic, classes, n_classes, max_cand, max_wh, scale, out, out_count, work, stream, nullptr, fused);
}

// ---- side lanes as separate graphs (yl_engine::lane_graphs) ------------------------------------------------------------
// One step = seg[0] on the caller's stream, then in program order: a main segment on the caller's stream; a side segment on
// its lane's stream behind an event recorded on the caller's stream at that point (= after every earlier main-lane op, which
// covers its data dependencies exactly as forward_impl's fork does); finally every used lane is joined and the decode + NMS
// graph follows.  Four to six graph launches and a few events per step instead of one launch -- for a caller that runs ONE
// batch at a time the 80x80 Detect branches then overlap the 40x40 / 20x20 neck instead of queueing behind it.
// The plan is keyed without the input address: the kernel nodes of segment 0 that read the input get a new address written
// into their parameters when it changes (yl_engine_detect_graph_rebind's mechanism), so one plan serves both entry points.
// First call for a new argument tuple: the step runs ONCE EAGERLY (argument checks, workspace preparation, lazy function
// attributes; one hipStreamSynchronize) and its segments are then captured and launched -- the network runs twice in that call, as
// in detect_graph_get.  yl_engine_detect_graph_prepare on such a program captures the lanes as branches of ONE graph instead.
static int run_range(yl_engine* e, size_t begin, size_t end, char* ws, hipStream_t s, const void* input, int in_dtype,
                     const PrefilterSink* sink) {
  for (size_t oi = begin; oi < end; ++oi) {
    if (e->ops[oi].group_n < 0) continue;  // launched with the first member of its group
    int consumed = 1;
    if (int rc = run_at(e, oi, ws, s, input, in_dtype, &consumed, sink)) return rc;
  }
  return 0;
}

static int end_capture(hipStream_t cap, int rc, hipGraph_t* g_out, hipGraphExec_t* x_out) {
  hipGraph_t g = nullptr;
  hipError_t ce = hipStreamEndCapture(cap, &g);
  if (rc) {
    if (g) (void)hipGraphDestroy(g);
    return rc;
  }
  if (ce != hipSuccess) return fail(-2, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
  hipError_t ie = hipGraphInstantiate(x_out, g, nullptr, nullptr, 0);
  if (ie != hipSuccess) {
    (void)hipGraphDestroy(g);
    return fail(-2, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie));
  }
  if (g_out) *g_out = g;
  else (void)hipGraphDestroy(g);
  return 0;
}

static int detect_lanes(yl_engine* e, const void* input, int in_dtype, void* workspace, float conf_thres, float iou_thres,
                        int max_det, int agnostic, const int32_t* classes, int n_classes, int max_cand, float max_wh,
                        const float* scale, float* out, int32_t* out_count, void* work, void* stream) {
  if (!e || !input || !workspace || !out || !out_count || !work) return fail(-1, "yl_engine_detect_graph: null pointer");
  if (e->levels.empty()) return fail(-1, "yl_engine_detect_graph: engine has no Detect levels");
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  yl_engine::DetectKey key(nullptr, in_dtype, workspace, conf_thres, iou_thres, max_det, agnostic, classes, n_classes, max_cand,
                           max_wh, scale, out, out_count, work);
  auto it = e->lplans.find(key);
  if (it == e->lplans.end()) {
    // eager first (real streams and events for the lanes): argument checks, workspace preparation, lazy function attributes
    if (int rc = detect_eager(e, input, in_dtype, workspace, conf_thres, iou_thres, max_det, agnostic, classes, n_classes, max_cand,
                              max_wh, scale, out, out_count, work, stream))
      return rc;
    YL_HIP(hipStreamSynchronize(s));
    hipStream_t cap = nullptr;
    if (int rc = capture_stream(e, workspace, &cap)) return rc;
    yl_engine::LanePlan P;
    struct Guard {
      yl_engine::LanePlan* p;
      ~Guard() {
        if (p) yl_engine::destroy_plan(*p);
      }
    } guard{&P};
    PrefilterSink sink{};
    int fused = 0;
    static const bool zero_in_stem = env_int("YL_ZERO_IN_STEM", 1) != 0;
    const bool defer = zero_in_stem && e->ops[0].d_w0 && e->ops[0].group_n >= 0;
    const bool stem = e->ops[0].op.kind == YL_OP_STEM || e->ops[0].d_w0;
    for (size_t k = 0; k < e->segments.size(); ++k) {
      const yl_engine::Segment& sg = e->segments[k];
      YL_HIP(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
      int rc = 0;
      if (k == 0) {  // (what detect_eager puts in front of the network)
        if (e->cls_tails)
          rc = prefilter_sink_prepare(e->levels.data(), (int)e->levels.size(), e->batch, max_cand, conf_thres, work, (void*)cap, &sink,
                                      &fused, defer);
        if (!rc && !stem) rc = run_convert(e, input, in_dtype, ws, cap);
      }
      if (!rc) rc = run_range(e, sg.begin, sg.end, ws, cap, input, in_dtype, fused ? &sink : nullptr);
      hipGraphExec_t x = nullptr;
      if (int rc2 = end_capture(cap, rc, k == 0 ? &P.g0 : nullptr, &x)) return rc2;
      P.seg.push_back(x);
    }
    YL_HIP(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
    int rc = decode_nms_stages(workspace, e->levels.data(), (int)e->levels.size(), e->batch, e->nc, conf_thres, iou_thres, max_det,
                               agnostic, classes, n_classes, max_cand, max_wh, scale, out, out_count, work, (void*)cap, nullptr, fused);
    if (int rc2 = end_capture(cap, rc, nullptr, &P.post)) return rc2;
    // the kernel nodes of segment 0 that take the input address as their first parameter
    size_t nn = 0;
    YL_HIP(hipGraphGetNodes(P.g0, nullptr, &nn));
    std::vector<hipGraphNode_t> all(nn);
    if (nn) YL_HIP(hipGraphGetNodes(P.g0, all.data(), &nn));
    for (hipGraphNode_t nd : all) {
      hipGraphNodeType ty;
      YL_HIP(hipGraphNodeGetType(nd, &ty));
      if (ty != hipGraphNodeTypeKernel) continue;
      hipKernelNodeParams kp{};
      YL_HIP(hipGraphKernelNodeGetParams(nd, &kp));
      if (!kp.kernelParams || !kp.kernelParams[0]) continue;
      const void* first = nullptr;
      memcpy(&first, kp.kernelParams[0], sizeof(first));
      if (first == input) P.in_nodes.push_back(nd);
    }
    if (P.in_nodes.empty()) return fail(-3, "lane graphs: no kernel node of the first segment reads the input address");
    P.cur = input;
    if (e->lplans.size() >= e->rcap) {  // evict the least recently launched tuple
      auto lru = e->lplans.begin();
      for (auto jt = e->lplans.begin(); jt != e->lplans.end(); ++jt)
        if (jt->second.used < lru->second.used) lru = jt;
      (void)hipDeviceSynchronize();
      yl_engine::destroy_plan(lru->second);
      e->lplans.erase(lru);
    }
    guard.p = nullptr;
    it = e->lplans.emplace(key, P).first;
  }
  yl_engine::LanePlan& P = it->second;
  P.used = ++e->rtick;
  if (P.cur != input) {
    for (hipGraphNode_t nd : P.in_nodes) {
      hipKernelNodeParams kp{};
      YL_HIP(hipGraphKernelNodeGetParams(nd, &kp));
      memcpy(kp.kernelParams[0], &input, sizeof(input));
      YL_HIP(hipGraphExecKernelNodeSetParams(P.seg[0], nd, &kp));
    }
    P.cur = input;
  }
  yl_engine::StreamSet& ss = e->sets[workspace];
  if (int rc = lane_resources(e, ss, e->segments.size() + (size_t)e->max_lane + 1)) return rc;
  size_t nev = 0;
  std::vector<char> used((size_t)e->max_lane + 1, 0);
  for (size_t k = 0; k < e->segments.size(); ++k) {
    const int lane = e->segments[k].lane;
    if (lane == 0) {
      YL_HIP(hipGraphLaunch(P.seg[k], s));
    } else {
      hipEvent_t ev = ss.events[nev++];
      YL_HIP(hipEventRecord(ev, s));
      YL_HIP(hipStreamWaitEvent(ss.lanes[lane], ev, 0));
      YL_HIP(hipGraphLaunch(P.seg[k], ss.lanes[lane]));
      used[lane] = 1;
    }
  }
  for (int k = 1; k <= e->max_lane; ++k)
    if (used[k]) {
      hipEvent_t ev = ss.events[nev++];
      YL_HIP(hipEventRecord(ev, ss.lanes[k]));
      YL_HIP(hipStreamWaitEvent(s, ev, 0));
    }
  YL_HIP(hipGraphLaunch(P.post, s));
  return 0;
}

// the captured graph of one argument tuple (captured on first use); launch != 0: also launched on `stream`
static int detect_graph_get(yl_engine* e, const void* input, int in_dtype, void* workspace, float conf_thres, float iou_thres,
                            int max_det, int agnostic, const int32_t* classes, int n_classes, int max_cand, float max_wh,
                            const float* scale, float* out, int32_t* out_count, void* work, void* stream, int launch,
                            hipGraphExec_t* exec_out, bool pin = false) {
  if (!e || !input || !workspace || !out || !out_count || !work)
    return fail(-1, "yl_engine_detect_graph: null pointer");
  if (e->levels.empty()) return fail(-1, "yl_engine_detect_graph: engine has no Detect levels");
  hipStream_t s = (hipStream_t)stream;
  yl_engine::DetectKey key(input, in_dtype, workspace, conf_thres, iou_thres, max_det, agnostic, classes, n_classes,
                           max_cand, max_wh, scale, out, out_count, work);
  hipGraphExec_t exec = e->dgraphs.find(key, pin);
  if (!exec) {
    // eager first: argument checks, workspace preparation and lazy function attributes happen here
    if (int rc = detect_eager(e, input, in_dtype, workspace, conf_thres, iou_thres, max_det, agnostic, classes,
                              n_classes, max_cand, max_wh, scale, out, out_count, work, stream))
      return rc;
    YL_HIP(hipStreamSynchronize(s));
    hipStream_t cap = nullptr;
    if (int rc = capture_stream(e, workspace, &cap)) return rc;
    hipGraph_t g = nullptr;
    YL_HIP(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
    int rc = detect_eager(e, input, in_dtype, workspace, conf_thres, iou_thres, max_det, agnostic, classes, n_classes,
                          max_cand, max_wh, scale, out, out_count, work, (void*)cap);
    hipError_t ce = hipStreamEndCapture(cap, &g);
    if (rc) {
      if (g) (void)hipGraphDestroy(g);
      return rc;
    }
    if (ce != hipSuccess) return fail(-2, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
    hipGraphExec_t ge = nullptr;
    hipError_t ie = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ie != hipSuccess) return fail(-2, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie));
    e->dgraphs.insert(key, ge, pin);
    exec = ge;
  }
  if (exec_out) *exec_out = exec;
  if (launch) YL_HIP(hipGraphLaunch(exec, s));
  return 0;
}

extern "C" int yl_engine_detect_graph(yl_engine* e, const void* input, int in_dtype, void* workspace,
                                      float conf_thres, float iou_thres, int max_det, int agnostic,
                                      const int32_t* classes, int n_classes, int max_cand, float max_wh,
                                      const float* scale, float* out, int32_t* out_count, void* work, void* stream) {
  if (e && e->lane_graphs)  // a program with side lanes: one graph per segment (detect_lanes)
    return detect_lanes(e, input, in_dtype, workspace, conf_thres, iou_thres, max_det, agnostic, classes, n_classes, max_cand, max_wh,
                        scale, out, out_count, work, stream);
  return detect_graph_get(e, input, in_dtype, workspace, conf_thres, iou_thres, max_det, agnostic, classes, n_classes, max_cand,
                          max_wh, scale, out, out_count, work, stream, 1, nullptr);
}

// The same graph WITHOUT launching it: a handle for yl_graph_launch_many.  (The first call for an argument tuple runs the
// path once eagerly on `stream` -- argument checks, workspace preparation -- and captures it.)
extern "C" int yl_engine_detect_graph_prepare(yl_engine* e, const void* input, int in_dtype, void* workspace,
                                              float conf_thres, float iou_thres, int max_det, int agnostic,
                                              const int32_t* classes, int n_classes, int max_cand, float max_wh,
                                              const float* scale, float* out, int32_t* out_count, void* work, void* stream,
                                              void** graph_out) {
  if (!graph_out) return fail(-1, "yl_engine_detect_graph_prepare: null pointer");
  // A program with side lanes (lane_graphs) normally replays as one graph per segment on a stream per lane; a single handle
  // cannot carry that, so here its lanes are captured as fork/join branches of ONE graph (what detect_eager records under
  // capture): the same results, the branches serialised by this runtime -- the price of the launch_many form for such a program.
  hipGraphExec_t ex = nullptr;
  const int rc = detect_graph_get(e, input, in_dtype, workspace, conf_thres, iou_thres, max_det, agnostic, classes, n_classes,
                                  max_cand, max_wh, scale, out, out_count, work, stream, 0, &ex, /*pin=*/true);
  *graph_out = (void*)ex;
  return rc;
}

// yl_engine_detect_graph for callers whose input address changes from call to call (the predictor API: a fresh tensor per
// batch).  One graph per argument tuple WITHOUT the input address; before a launch with a new address the parameters of the
// kernel nodes that read the input (the stem launch) are updated in the executable (hipGraphExecKernelNodeSetParams).
// Contract: no earlier launch of the same tuple is still pending when this is called (one slot = one batch at a time).
extern "C" int yl_engine_detect_graph_rebind(yl_engine* e, const void* input, int in_dtype, void* workspace,
                                             float conf_thres, float iou_thres, int max_det, int agnostic,
                                             const int32_t* classes, int n_classes, int max_cand, float max_wh,
                                             const float* scale, float* out, int32_t* out_count, void* work, void* stream) {
  if (!e || !input || !workspace || !out || !out_count || !work)
    return fail(-1, "yl_engine_detect_graph_rebind: null pointer");
  if (e->levels.empty()) return fail(-1, "yl_engine_detect_graph_rebind: engine has no Detect levels");
  if (e->lane_graphs)
    return detect_lanes(e, input, in_dtype, workspace, conf_thres, iou_thres, max_det, agnostic, classes, n_classes, max_cand, max_wh,
                        scale, out, out_count, work, stream);
  hipStream_t s = (hipStream_t)stream;
  yl_engine::DetectKey key(nullptr, in_dtype, workspace, conf_thres, iou_thres, max_det, agnostic, classes, n_classes,
                           max_cand, max_wh, scale, out, out_count, work);
  auto it = e->rgraphs.find(key);
  if (it == e->rgraphs.end()) {
    if (int rc = detect_eager(e, input, in_dtype, workspace, conf_thres, iou_thres, max_det, agnostic, classes,
                              n_classes, max_cand, max_wh, scale, out, out_count, work, stream))
      return rc;
    YL_HIP(hipStreamSynchronize(s));
    hipStream_t cap = nullptr;
    if (int rc = capture_stream(e, workspace, &cap)) return rc;
    yl_engine::Rebindable rb;
    YL_HIP(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
    int rc = detect_eager(e, input, in_dtype, workspace, conf_thres, iou_thres, max_det, agnostic, classes, n_classes,
                          max_cand, max_wh, scale, out, out_count, work, (void*)cap);
    hipError_t ce = hipStreamEndCapture(cap, &rb.g);
    // every exit below that does not hand rb to the table destroys the captured graph (and the executable, if any)
    struct Guard {
      yl_engine::Rebindable* r;
      ~Guard() {
        if (!r) return;
        if (r->exec) (void)hipGraphExecDestroy(r->exec);
        if (r->g) (void)hipGraphDestroy(r->g);
      }
    } guard{&rb};
    if (rc) return rc;
    if (ce != hipSuccess) return fail(-2, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
    size_t nn = 0;
    YL_HIP(hipGraphGetNodes(rb.g, nullptr, &nn));
    std::vector<hipGraphNode_t> all(nn);
    if (nn) YL_HIP(hipGraphGetNodes(rb.g, all.data(), &nn));
    for (hipGraphNode_t nd : all) {
      hipGraphNodeType ty;
      YL_HIP(hipGraphNodeGetType(nd, &ty));
      if (ty != hipGraphNodeTypeKernel) continue;
      hipKernelNodeParams kp{};
      YL_HIP(hipGraphKernelNodeGetParams(nd, &kp));
      if (!kp.kernelParams || !kp.kernelParams[0]) continue;
      const void* first = nullptr;
      memcpy(&first, kp.kernelParams[0], sizeof(first));  // every input-reading launch takes the address first
      if (first == input) rb.nodes.push_back(nd);
    }
    if (rb.nodes.empty())
      return fail(-3, "yl_engine_detect_graph_rebind: no kernel node of the captured graph reads the input address");
    hipError_t ie = hipGraphInstantiate(&rb.exec, rb.g, nullptr, nullptr, 0);
    if (ie != hipSuccess) return fail(-2, std::string("hipGraphInstantiate: ") + hipGetErrorString(ie));
    rb.cur = input;
    if (e->rgraphs.size() >= e->rcap) {  // evict the least recently launched tuple
      auto lru = e->rgraphs.begin();
      for (auto jt = e->rgraphs.begin(); jt != e->rgraphs.end(); ++jt)
        if (jt->second.used < lru->second.used) lru = jt;
      (void)hipDeviceSynchronize();
      if (lru->second.exec) (void)hipGraphExecDestroy(lru->second.exec);
      if (lru->second.g) (void)hipGraphDestroy(lru->second.g);
      e->rgraphs.erase(lru);
    }
    guard.r = nullptr;
    it = e->rgraphs.emplace(key, rb).first;
  }
  yl_engine::Rebindable& rb = it->second;
  rb.used = ++e->rtick;
  if (rb.cur != input) {
    for (hipGraphNode_t nd : rb.nodes) {
      hipKernelNodeParams kp{};
      YL_HIP(hipGraphKernelNodeGetParams(nd, &kp));
      memcpy(kp.kernelParams[0], &input, sizeof(input));  // the node's own copy of its first parameter
      YL_HIP(hipGraphExecKernelNodeSetParams(rb.exec, nd, &kp));
    }
    rb.cur = input;
  }
  YL_HIP(hipGraphLaunch(rb.exec, s));
  return 0;
}

// n prepared steps issued back to back: graphs[i] on streams[i].  One native call instead of n trips through the host
// language: a step of the hot path is ONE graph launch (19-28 us of host time for its 42 kernel nodes); issued from Python it
// costs ~80 us, which in a short run (20 steps on 20 streams) is a fill ramp of 1.6 ms in front of 12 ms of GPU work.
//
// Round 4: the launches are dealt to a small pool of issue threads BY STREAM (every launch of one stream -- and so of one graph
// executable -- stays on one thread, in order): issued from one thread the twentieth batch of a 20-step run started 0.4-0.55 ms
// after the first one, 4 % of the run.  The pool is created on first use and lives as long as the process.
namespace {
struct IssuePool {
  static constexpr int NW = 3;  // workers beside the calling thread
  struct Job {
    void* const* graphs = nullptr;
    void* const* streams = nullptr;
    const int* idx = nullptr;  // indices into graphs / streams, in issue order
    int n = 0;
    int device = 0;
  };
  std::mutex m;
  std::condition_variable cv_go, cv_done;
  Job job[NW];
  uint64_t gen = 0;         // bumped per dispatch
  int pending = 0;          // workers still running this dispatch
  hipError_t err = hipSuccess;
  std::thread th[NW];
  bool started = false;

  static hipError_t run(const Job& j) {
    for (int k = 0; k < j.n; ++k) {
      const int i = j.idx[k];
      const hipError_t e = hipGraphLaunch((hipGraphExec_t)j.graphs[i], (hipStream_t)j.streams[i]);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
  void worker(int w) {
    uint64_t seen = 0;
    int dev = -1;
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(m);
        cv_go.wait(lk, [&] { return gen != seen; });
        seen = gen;
        j = job[w];
      }
      hipError_t e = hipSuccess;
      if (j.n > 0) {
        if (dev != j.device) {
          e = hipSetDevice(j.device);
          dev = j.device;
        }
        if (e == hipSuccess) e = run(j);
      }
      {
        std::lock_guard<std::mutex> lk(m);
        if (e != hipSuccess && err == hipSuccess) err = e;
        if (--pending == 0) cv_done.notify_one();
      }
    }
  }
  void start() {
    if (started) return;
    for (int w = 0; w < NW; ++w) {
      th[w] = std::thread([this, w] { worker(w); });
      th[w].detach();
    }
    started = true;
  }
};
IssuePool* g_issue_pool = nullptr;  // (never destroyed: its threads wait on it until the process ends)
std::mutex g_issue_mu;              // one dispatch at a time
}  // namespace

extern "C" int yl_graph_launch_many(void* const* graphs, void* const* streams, int n) {
  if (!graphs || !streams || n < 0) return fail(-1, "yl_graph_launch_many: bad arguments");
  for (int i = 0; i < n; ++i)
    if (!graphs[i]) return fail(-1, "yl_graph_launch_many: null graph");
  static const int threads_env = tune_int("YL_ISSUE_THREADS", 4);
  const int parts = threads_env < 1 ? 1 : threads_env > IssuePool::NW + 1 ? IssuePool::NW + 1 : threads_env;
  // distinct streams in order of first appearance; stream k goes to part k % parts
  std::vector<void*> seen;
  std::vector<std::vector<int>> idx((size_t)parts);
  if (parts > 1 && n >= 8) {
    for (int i = 0; i < n; ++i) {
      size_t k = 0;
      while (k < seen.size() && seen[k] != streams[i]) ++k;
      if (k == seen.size()) seen.push_back(streams[i]);
      idx[k % (size_t)parts].push_back(i);
    }
  }
  if (parts == 1 || n < 8 || seen.size() < 2) {
    for (int i = 0; i < n; ++i) YL_HIP(hipGraphLaunch((hipGraphExec_t)graphs[i], (hipStream_t)streams[i]));
    return 0;
  }
  std::lock_guard<std::mutex> one(g_issue_mu);
  if (!g_issue_pool) g_issue_pool = new IssuePool();
  IssuePool& P = *g_issue_pool;
  int dev = 0;
  YL_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lk(P.m);
    P.start();
    for (int w = 0; w < IssuePool::NW; ++w) {
      IssuePool::Job& j = P.job[w];
      const bool has = w + 1 < parts;
      j.graphs = graphs;
      j.streams = streams;
      j.idx = has ? idx[(size_t)w + 1].data() : nullptr;
      j.n = has ? (int)idx[(size_t)w + 1].size() : 0;
      j.device = dev;
    }
    P.err = hipSuccess;
    P.pending = IssuePool::NW;
    ++P.gen;
  }
  P.cv_go.notify_all();
  IssuePool::Job mine;
  mine.graphs = graphs;
  mine.streams = streams;
  mine.idx = idx[0].data();
  mine.n = (int)idx[0].size();
  const hipError_t e0 = IssuePool::run(mine);
  hipError_t e1;
  {
    std::unique_lock<std::mutex> lk(P.m);
    P.cv_done.wait(lk, [&] { return P.pending == 0; });
    e1 = P.err;
  }
  if (e0 != hipSuccess) return fail(-2, std::string("hipGraphLaunch: ") + hipGetErrorString(e0));
  if (e1 != hipSuccess) return fail(-2, std::string("hipGraphLaunch (issue thread): ") + hipGetErrorString(e1));
  return 0;
}

extern "C" int yl_engine_num_ops(const yl_engine* e) { return e ? (int)e->ops.size() : -1; }

// 1 when yl_engine_detect_graph at this confidence threshold runs the NMS candidate pre-filter inside the launches that
// write the class logits (one launch fewer, the class planes are not re-read), 0 when it is a launch of its own
extern "C" int yl_engine_prefilter_fused(const yl_engine* e, float conf_thres) {
  return e && e->cls_tails && conf_thres >= 0.05f && conf_thres <= 1.f ? 1 : 0;
}

extern "C" int yl_engine_op_info(const yl_engine* e, int i, double* flops, double* bytes, const char** name) {
  if (!e || i < 0 || i >= (int)e->ops.size()) return fail(-1, "yl_engine_op_info: bad index");
  if (flops) *flops = e->ops[i].flops;
  if (bytes) *bytes = e->ops[i].bytes;
  if (name) *name = e->ops[i].name;
  return 0;
}

extern "C" int yl_engine_profile(yl_engine* e, const void* input, int in_dtype, void* workspace, float* y,
                                 void* stream, int iters, float* ms) {
  if (!e || !input || !workspace || !ms || iters <= 0) return fail(-1, "yl_engine_profile: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  if (int rc = yl_engine_forward(e, input, in_dtype, workspace, y, stream)) return rc;
  YL_HIP(hipStreamSynchronize(s));
  // In-sequence timing: one event between every pair of launches of a whole forward pass, so each
  // op sees the cache state it has in the real run (its input was just produced by its
  // predecessor), not a warm re-run of itself.  Averaged over `iters` passes.
  const int n = (int)e->ops.size();
  std::vector<hipEvent_t> ev((size_t)n + 3);
  for (auto& x : ev) YL_HIP(hipEventCreate(&x));
  std::vector<double> acc((size_t)n + 2, 0.0);
  int rc = 0;
  for (int it = 0; it < iters && !rc; ++it) {
    (void)hipEventRecord(ev[0], s);
    if (e->ops[0].op.kind != YL_OP_STEM && !e->ops[0].d_w0) rc = run_convert(e, input, in_dtype, ws, s);
    (void)hipEventRecord(ev[1], s);
    for (int i = 0; i < n && !rc; ++i) {
      int consumed = 1;
      if (e->ops[i].group_n >= 0) rc = run_at(e, (size_t)i, ws, s, input, in_dtype, &consumed);  // (a group's time lands on its first member)
      (void)hipEventRecord(ev[i + 2], s);
    }
    if (y && !e->levels.empty() && !rc) rc = run_decode(e, ws, y, s);
    (void)hipEventRecord(ev[n + 2], s);
    if (hipEventSynchronize(ev[n + 2]) != hipSuccess) rc = fail(-2, "hipEventSynchronize failed");
    for (int i = 0; i < n + 2 && !rc; ++i) {
      float t = 0.f;
      (void)hipEventElapsedTime(&t, ev[i], ev[i + 1]);
      acc[i] += t;
    }
  }
  for (int i = 0; i < n + 2; ++i) ms[i] = (float)(acc[i] / iters);
  for (auto& x : ev) (void)hipEventDestroy(x);
  return rc;
}

// The hot path as bench.py runs it, timed launch by launch: every network op in program order on ONE stream (side
// lanes folded into it), then the five stages of yl_decode_nms.  ms_ops[n_ops], ms_post[5].
extern "C" int yl_engine_profile_detect(yl_engine* e, const void* input, int in_dtype, void* workspace, float conf_thres,
                                        float iou_thres, int max_det, int max_cand, float* out, int32_t* out_count,
                                        void* work, void* stream, int iters, float* ms_ops, float* ms_post) {
  if (!e || !input || !workspace || !out || !out_count || !work || !ms_ops || !ms_post || iters <= 0)
    return fail(-1, "yl_engine_profile_detect: bad arguments");
  if (e->levels.empty()) return fail(-1, "yl_engine_profile_detect: engine has no Detect levels");
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  if (int rc = detect_eager(e, input, in_dtype, workspace, conf_thres, iou_thres, max_det, 0, nullptr, 0, max_cand, 7680.f,
                            nullptr, out, out_count, work, stream))
    return rc;
  YL_HIP(hipStreamSynchronize(s));
  const int n = (int)e->ops.size();
  std::vector<hipEvent_t> ev((size_t)n + 1), pv(6);
  for (auto& x : ev) YL_HIP(hipEventCreate(&x));
  for (auto& x : pv) YL_HIP(hipEventCreate(&x));
  std::vector<double> acc((size_t)n, 0.0), pacc(5, 0.0);
  int rc = 0;
  const bool stem = e->ops[0].op.kind == YL_OP_STEM || e->ops[0].d_w0;
  for (int it = 0; it < iters && !rc; ++it) {
    if (!stem) rc = run_convert(e, input, in_dtype, ws, s);
    PrefilterSink sink{};
    int fused = 0;  // as detect_eager: the pre-filter in the class launches' epilogue (its stage then times ~0)
    if (!rc && e->cls_tails)
      rc = prefilter_sink_prepare(e->levels.data(), (int)e->levels.size(), e->batch, max_cand, conf_thres, work, stream, &sink, &fused);
    (void)hipEventRecord(ev[0], s);
    for (int i = 0; i < n && !rc; ++i) {
      int consumed = 1;
      if (e->ops[i].group_n >= 0)  // (a group's time lands on its first member)
        rc = run_at(e, (size_t)i, ws, s, input, in_dtype, &consumed, fused ? &sink : nullptr);
      (void)hipEventRecord(ev[i + 1], s);
    }
    if (!rc)
      rc = decode_nms_stages(workspace, e->levels.data(), (int)e->levels.size(), e->batch, e->nc, conf_thres, iou_thres,
                             max_det, 0, nullptr, 0, max_cand, 7680.f, nullptr, out, out_count, work, stream, pv.data(), fused);
    if (!rc && hipEventSynchronize(pv[5]) != hipSuccess) rc = fail(-2, "hipEventSynchronize failed");
    for (int i = 0; i < n && !rc; ++i) {
      float t = 0.f;
      (void)hipEventElapsedTime(&t, ev[i], ev[i + 1]);
      acc[i] += t;
    }
    for (int i = 0; i < 5 && !rc; ++i) {
      float t = 0.f;
      (void)hipEventElapsedTime(&t, pv[i], pv[i + 1]);
      pacc[i] += t;
    }
  }
  for (int i = 0; i < n; ++i) ms_ops[i] = (float)(acc[i] / iters);
  for (int i = 0; i < 5; ++i) ms_post[i] = (float)(pacc[i] / iters);
  for (auto& x : ev) (void)hipEventDestroy(x);
  for (auto& x : pv) (void)hipEventDestroy(x);
  return rc;
}

// ---- CU-masked streams: a HIP stream whose dispatches are confined to a subset of the compute units
// (hipExtStreamCreateWithCUMask).  Built to give every batch in flight a part of the chip of its own; measured: a queue
// cannot be confined to one XCD and static slices are far slower than the hardware's own packing.
extern "C" int yl_stream_create_masked(const uint32_t* cu_mask, int n_words, void** stream_out) {
  if (!cu_mask || n_words <= 0 || !stream_out) return fail(-1, "yl_stream_create_masked: bad arguments");
  hipStream_t s = nullptr;
  YL_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, cu_mask));
  *stream_out = (void*)s;
  return 0;
}

extern "C" int yl_stream_destroy(void* stream) {
  if (!stream) return fail(-1, "yl_stream_destroy: null stream");
  YL_HIP(hipStreamDestroy((hipStream_t)stream));
  return 0;
}

// Which compute units does a stream's work land on?  Every workgroup adds one to the counter of the (XCD, shader
// engine, CU) it runs on: bin = xcc * 64 + se * 16 + cu (HW_REG_XCC_ID, HW_REG_HW_ID).  hist: 512 x uint32 on the device,
// zeroed by the caller.
__global__ void xcc_census_kernel(unsigned* hist) {
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11)) & 7u;
    const unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | ((32 - 1) << 11));
    const unsigned cu = (hw >> 8) & 15u, se = (hw >> 13) & 3u;
    atomicAdd(hist + xcc * 64 + se * 16 + cu, 1u);
    // keep the workgroup alive for a moment so that a probe launch spreads over every unit it may use
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 500) __builtin_amdgcn_s_sleep(4);
  }
}

extern "C" int yl_xcc_census(unsigned* hist8_device /* 512 bins */, int n_workgroups, void* stream) {
  if (!hist8_device || n_workgroups <= 0) return fail(-1, "yl_xcc_census: bad arguments");
  hipLaunchKernelGGL(xcc_census_kernel, dim3((unsigned)n_workgroups), dim3(64), 0, (hipStream_t)stream, hist8_device);
  YL_HIP(hipGetLastError());
  return 0;
}

// Do two parallel branches of a captured graph really run concurrently on this runtime?  (The hot path's graph has
// exactly two; whether the runtime maps them to two hardware queues depends on DEBUG_HIP_FORCE_GRAPH_QUEUES being in
// the environment before HIP initialised -- _lib.apply_runtime_knobs.)  Two one-wave kernels that each spin for
// `spin_us` of the constant 100 MHz clock are captured as a fork / join; the replay takes ~spin_us when they overlap
// and ~2 x spin_us when the runtime serialises them.
__global__ void spin_kernel(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// Diagnostic builds only (-DYL_STAMPS): where the instrumented kernels store their phase stamps (nullptr: off)
extern "C" int yl_diag_set_stamps(void* device_buffer) {
  g_stamp_buf = reinterpret_cast<unsigned long long*>(device_buffer);
#ifdef YL_STAMPS
  return 1;
#else
  return 0;
#endif
}

extern "C" int yl_graph_branch_probe(void* stream, int spin_us, float* ms_graph, int* concurrent) {
  if (!ms_graph || !concurrent || spin_us <= 0 || spin_us > 100000) return fail(-1, "yl_graph_branch_probe: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipStream_t cap = nullptr, side = nullptr;
  hipEvent_t fork = nullptr, join = nullptr, t0 = nullptr, t1 = nullptr;
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  int rc = 0;
  const unsigned long long ticks = (unsigned long long)spin_us * 100ull;
  auto cleanup = [&]() {
    if (ge) (void)hipGraphExecDestroy(ge);
    if (g) (void)hipGraphDestroy(g);
    for (hipEvent_t ev : {fork, join, t0, t1})
      if (ev) (void)hipEventDestroy(ev);
    if (cap) (void)hipStreamDestroy(cap);
    if (side) (void)hipStreamDestroy(side);
  };
#define YL_PROBE(expr)                                                                 \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      rc = fail(-2, std::string(#expr) + ": " + hipGetErrorString(_e));                \
      cleanup();                                                                       \
      return rc;                                                                       \
    }                                                                                  \
  } while (0)
  YL_PROBE(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
  YL_PROBE(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  YL_PROBE(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  YL_PROBE(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  YL_PROBE(hipEventCreate(&t0));
  YL_PROBE(hipEventCreate(&t1));
  YL_PROBE(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
  YL_PROBE(hipEventRecord(fork, cap));
  YL_PROBE(hipStreamWaitEvent(side, fork, 0));
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, cap, ticks);
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, side, ticks);
  YL_PROBE(hipEventRecord(join, side));
  YL_PROBE(hipStreamWaitEvent(cap, join, 0));
  YL_PROBE(hipStreamEndCapture(cap, &g));
  YL_PROBE(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  YL_PROBE(hipGraphLaunch(ge, s));  // warm-up
  YL_PROBE(hipStreamSynchronize(s));
  float best = 1e30f;
  for (int it = 0; it < 3; ++it) {
    YL_PROBE(hipEventRecord(t0, s));
    YL_PROBE(hipGraphLaunch(ge, s));
    YL_PROBE(hipEventRecord(t1, s));
    YL_PROBE(hipEventSynchronize(t1));
    float t = 0.f;
    YL_PROBE(hipEventElapsedTime(&t, t0, t1));
    if (t < best) best = t;
  }
#undef YL_PROBE
  *ms_graph = best;
  *concurrent = best < 1.5f * (float)spin_us * 1e-3f ? 1 : 0;
  cleanup();
  return 0;
}

extern "C" int yl_detect_decode(const void* workspace, const yl_level* levels, int n_levels, int batch, int nc,
                                float* y, void* stream) {
  if (!workspace || !levels || !y || n_levels <= 0 || n_levels > 8) return fail(-1, "yl_detect_decode: bad arguments");
  DecodeLevel lv[8];
  int a0 = 0;
  for (int i = 0; i < n_levels; ++i) {
    lv[i] = decode_level(workspace, levels[i], batch, a0);
    a0 += lv[i].H * lv[i].W;
  }
  return decode_launch(lv, n_levels, batch, nc, a0, y, (hipStream_t)stream);
}

// engine/predictor.py:77-84: BGR HWC uint8 -> RGB CHW float / 255
__global__ void preprocess_u8_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int HW, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long b = i / HW;
  const int p = (int)(i - b * HW);
  const uint8_t* s = in + i * 3;
  float* o = out + b * 3 * HW + p;
  o[0] = (float)s[2] / 255.0f;
  o[HW] = (float)s[1] / 255.0f;
  o[2LL * HW] = (float)s[0] / 255.0f;
}

extern "C" int yl_preprocess_u8(const uint8_t* bgr_hwc, int batch, int h, int w, float* out_nchw, void* stream) {
  if (!bgr_hwc || !out_nchw || batch <= 0 || h <= 0 || w <= 0) return fail(-1, "yl_preprocess_u8: bad arguments");
  const long long total = (long long)batch * h * w;
  hipLaunchKernelGGL(preprocess_u8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     bgr_hwc, out_nchw, h * w, total);
  YL_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------- letterbox on the device
// Reference data/augment.py:636-681 (LetterBox.__call__: cv2.resize INTER_LINEAR + copyMakeBorder(114))
// followed by predictor.py:77-81 (stack, BGR -> RGB, HWC -> CHW), for ONE source image into slot b of the
// uint8 NCHW batch.  The bilinear arithmetic is OpenCV's published 8-bit fixed-point scheme, restated in
// data/augment.py::resize_linear_u8 (the CPU oracle of this kernel; cv2 itself is absent, so pixel parity
// with cv2 stays unpinned): horizontal pass with 11-bit coefficients, >> 4, vertical pass (b * s) >> 16,
// + 2, >> 2.  The coefficient tables come from the host (x: [x0, x1, a0, a1] per output column, y likewise)
// so the float -> fixed conversion is the CPU restatement's own and the integer part is bit-exact.
__global__ void letterbox_u8_kernel(const uint8_t* __restrict__ src, int sh, int sw, int spitch,
                                    uint8_t* __restrict__ dst, int H, int W, int top, int left, int nh, int nw,
                                    const int4* __restrict__ xt, const int4* __restrict__ yt, int resize) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const int HW = H * W;
  uint8_t* o = dst + y * W + x;
  const int ux = x - left, uy = y - top;
  if (ux < 0 || ux >= nw || uy < 0 || uy >= nh) {
    o[0] = o[HW] = o[2 * HW] = 114;
    return;
  }
  if (!resize) {  // same size: copy (augment.py:665 skips cv2.resize)
    const uint8_t* p = src + (long long)uy * spitch + ux * 3;
    o[0] = p[2], o[HW] = p[1], o[2 * HW] = p[0];
    return;
  }
  const int4 cx = xt[ux], cy = yt[uy];
  const uint8_t* r0 = src + (long long)cy.x * spitch;
  const uint8_t* r1 = src + (long long)cy.y * spitch;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int h0 = r0[cx.x * 3 + c] * cx.z + r0[cx.y * 3 + c] * cx.w;  // scaled by 2^11
    const int h1 = r1[cx.x * 3 + c] * cx.z + r1[cx.y * 3 + c] * cx.w;
    int v = (((cy.z * (h0 >> 4)) >> 16) + ((cy.w * (h1 >> 4)) >> 16) + 2) >> 2;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    o[(2 - c) * HW] = (uint8_t)v;  // BGR -> RGB
  }
}

extern "C" int yl_letterbox_u8(const uint8_t* src_bgr_hwc, int src_h, int src_w, int src_pitch, uint8_t* dst_chw,
                               int H, int W, int top, int left, int new_h, int new_w, const int32_t* xtab,
                               const int32_t* ytab, void* stream) {
  if (!src_bgr_hwc || !dst_chw || src_h <= 0 || src_w <= 0 || H <= 0 || W <= 0 || new_h <= 0 || new_w <= 0 ||
      src_pitch < 3 * src_w)
    return fail(-1, "yl_letterbox_u8: bad arguments");
  if (top < 0 || left < 0 || top + new_h > H || left + new_w > W)
    return fail(-1, "yl_letterbox_u8: the resized image does not fit the output");
  const int resize = !(new_h == src_h && new_w == src_w);
  if (resize && (!xtab || !ytab)) return fail(-1, "yl_letterbox_u8: coefficient tables missing");
  hipLaunchKernelGGL(letterbox_u8_kernel, dim3((W + 255) / 256, H), dim3(256), 0, (hipStream_t)stream, src_bgr_hwc,
                     src_h, src_w, src_pitch, dst_chw, H, W, top, left, new_h, new_w,
                     reinterpret_cast<const int4*>(xtab), reinterpret_cast<const int4*>(ytab), resize);
  YL_HIP(hipGetLastError());
  return 0;
}
