// 3x3 stride-1 Conv + (folded BN) bias + SiLU with 64 output channels and a DEEP K (128 or 256 input channels) on small maps:
// the first convs of Detect's box branch on the 40x40 and 20x20 levels (reference nn/modules/head.py:36-40 cv2[i][0],
// conv.py:35-53), 128 -> 64 and 256 -> 64 in YOLO11n.
//
// Why a kernel of its own: conv_kernel gives a workgroup 256 pixels and restages the whole filter (144 / 288 KB) through its
// 48 KB LDS slab in chunks with barriers for them; on the 20x20 level that is 50 workgroups each walking 72 k-steps serially
// -- 29.5 us for 3.8 GFLOP (5 % of the matrix peak), 42 us for the two levels in one launch.  Here the roles are swapped, as in
// resident.hip: a workgroup owns a BAND of rows of one image, stages its INPUT (with the halo rows) once into LDS as a guarded
// map -- one contiguous run per plane in the guarded HBM layout, so a tap is a constant byte offset and the B fragments are
// plain ds_read_b128 -- and streams the filter through a small LDS ring by LDS-DMA (buffer_load ... lds: no registers, three
// slices of two k-steps in flight), every filter byte fetched ONCE per workgroup.  (First version: filter fragments straight
// from L2 into registers, as resident.hip does.  With 224 workgroups reading the same 4 KB per k-step at the same moment, four
// waves each, the L2 channels that hold those lines were the bottleneck: 630-750 cycles per k-step against 224-320 of MFMA
// work -- stamps in tools/band_micro.py, NOTEBOOK.md section 13.)  Eight waves = M-half x N-quarter: 2 filter reads and NT
// activation reads from LDS per 2 NT MFMAs.
// Same packed filters, k order (tap-major), accumulation chain from the bias, SiLU and rounding as conv_kernel:
// BIT-IDENTICAL results (tests/test_gpu_kernels.py::test_conv3_band_vs_conv_kernel).
#include "lds_conv.h"

namespace yl {

constexpr int BD_THREADS = 512;
#ifndef YL_BAND_SCHED
#define YL_BAND_SCHED 0  // 1: one LDS read pinned behind every MFMA (sched_group_barrier): 128 -> 64 @40x40 9.2 -> 8.5 us, 256 -> 64 @20x20 9.9 -> 10.1 us
#endif
#ifndef YL_BAND_ABL
#define YL_BAND_ABL 0  // development build switch: timing ablations of the k-loop (wrong results): 1 no barrier, 2 no ring refill, 4 no
                      // filter-fragment reads, 8 no activation-fragment reads, 16 no counted wait
#endif

// LDS bytes per plane of the band tile: rows y0 - 1 .. y0 + TR of W1 pixels (+ the pixel in front of the first row and one
// behind the last), rounded to 16 pixels (plane pitch a multiple of 256 bytes: conflict-free ds_read_b128)
static inline int band_tpl(int TR, int W) { return ((((TR + 2) * (W + 1) + 2) + 15) & ~15) * 16; }

typedef __attribute__((address_space(3))) void lds_void_b;
constexpr int BD_NS = 6;                 // ring slices: one free (being refilled), one whose fragments are in use, one being read, three in flight (an L2 -> LDS
                                         // piece takes ~2500 cycles to land with every workgroup streaming: 16 KB in flight fed a slice per 1250 cycles,
                                         // twice the MFMA time of a 20-tile band)
constexpr int BD_SLICE = 2 * 4 * 1024;   // bytes of a slice: two k-steps x four tiles of the packed filter = 8 pieces of 1 KiB, one per wave

// piece `wave` of filter slice `sl` -> ring (slices past the end move nothing but still count in vmcnt)
__device__ __forceinline__ void ring_issue(__amdgpu_buffer_rsrc_t rw, unsigned char* ring, int sl, int nsl, int wave, int lane) {
  const unsigned off = sl < nsl ? (unsigned)sl * BD_SLICE + (unsigned)wave * 1024u + (unsigned)lane * 16u : OOB_LANE;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_b*)(ring + (sl % BD_NS) * BD_SLICE + wave * 1024), 16, (int)off, 0, 0, 0);
}

// POST: a 1x1 conv 64 -> 64 that consumes ONLY this conv's output (Detect's cv2[i][1] -> cv2[i][2], head.py:36-40) runs behind it
// in the same launch: the 3x3 conv's output tile goes to LDS (bf16, dense planes = B-fragment order), one workgroup barrier,
// two k-steps from there; that tensor is never stored.  Same arithmetic as conv_kernel's POST epilogue.
template <int KKN, int NT, bool POST = false>
__device__ __forceinline__ void band_compute(const BandArgs& a, unsigned char* lds, unsigned char* ring, __amdgpu_buffer_rsrc_t rw, int b,
                                             int y0, int mh, int t0, int n, int g, int lane, int wave) {
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, (short)0, a.y_bytes, 0x00020000);
  constexpr int NKS = 9 * KKN, NSL = NKS / 2;
  static_assert(NKS % 2 == 0, "slices of two k-steps");
  unsigned tb[NT], opix[NT];
  const int npx = a.TR * a.W;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int i = (t0 + nt) * 16 + n;
    const int ic = i < npx ? i : npx - 1;
    const int r = ic / a.W, x = ic - r * a.W;
    tb[nt] = (unsigned)((r + 1) * a.W1 + x + 1) * 16u;
    opix[nt] = (i < npx && y0 + r < a.H) ? (unsigned)(a.g0 + b * a.img + (y0 + r) * a.W1 + x) * 16u : OOB_LANE;
  }
  f32x4 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + chan_in_block(4, 2 * mh + m, g, 0));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[m][nt] = bv;
  }
  const unsigned char* src = lds + g * a.tpl;
  auto load_b = [&](int ks, bf16x8 (&dst)[NT]) __attribute__((always_inline)) {
    const int tap = ks / KKN, kk = ks - tap * KKN;
    const int kh = tap / 3, kw = tap - 3 * kh;
    const int off = ((kh - 1) * a.W1 + (kw - 1)) * 16 + kk * 4 * a.tpl;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) dst[nt] = *reinterpret_cast<const bf16x8*>(src + (int)tb[nt] + off);
  };
  bf16x8 xb[2][NT];  // the activation fragments of a slice's two k-steps: read one slice ahead (they do not depend on the ring)
  load_b(0, xb[0]);
  load_b(1, xb[1]);
  // filter fragments of a slice: read from the ring ONE SLICE AHEAD as well, so that the barrier does not stand between a
  // slice's landing and its first MFMA (slices 0 .. BD_NS - 2 were requested before the input was staged and have landed:
  // the staging ends with a full wait and a barrier)
  auto load_a = [&](int sl, bf16x8 (&dst)[2][2]) __attribute__((always_inline)) {
    const unsigned char* ab = ring + (sl % BD_NS) * BD_SLICE + (2 * mh) * 1024 + lane * 16;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int m = 0; m < 2; ++m) dst[kk][m] = *reinterpret_cast<const bf16x8*>(ab + (kk * 4 + m) * 1024);
  };
  bf16x8 wa[2][2][2];
  load_a(0, wa[0]);
#pragma unroll
  for (int sl = 0; sl < NSL; ++sl) {
    // my own piece of slice sl + 1 has landed (the younger ones stay in flight); after the barrier everybody's has, and
    // nobody reads slice sl - 1 any more (its fragments were consumed by the MFMAs of the previous iteration): its buffer takes
    // slice sl + BD_NS - 1
    if (!(YL_BAND_ABL & 16)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(BD_NS - 3) : "memory");
    if (!(YL_BAND_ABL & 1)) __builtin_amdgcn_s_barrier();
    if (!(YL_BAND_ABL & 2)) ring_issue(rw, ring, sl + BD_NS - 1, NSL, wave, lane);
    if (sl + 1 < NSL && !(YL_BAND_ABL & 4)) load_a(sl + 1, wa[(sl + 1) & 1]);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int m = 0; m < 2; ++m)
          acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[sl & 1][kk][m], xb[kk][nt], acc[m][nt], 0, 0, 0);
      if (2 * sl + 2 + kk < NKS && !(YL_BAND_ABL & 8)) load_b(2 * sl + 2 + kk, xb[kk]);
    }
#if YL_BAND_SCHED
    // pin the interleave: one LDS read (of the NEXT slice's fragments) behind every MFMA, so that the reads are issued in the
    // shadow of the matrix pipe instead of in a block of their own (the two waves of a SIMD run this loop in lockstep)
#pragma unroll
    for (int q = 0; q < 4 * NT; ++q) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one DS read
    }
#endif
  }
  // the ring's last refills moved nothing (slices past the end resolve to the out-of-range offset), but an LDS-DMA operation
  // that is still in flight when the workgroup ends would land -- as zeros -- in LDS that already belongs to another workgroup:
  // they must retire here (found by tests/test_gpu_net.py::test_twenty_batches_in_flight: boxes of OTHER batches moved)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  YL_STAMP(a.stamps, 8, wave, lane, 3);
  if constexpr (POST) {
    unsigned char* const mid = ring + BD_NS * BD_SLICE;  // [8 planes][ntiles * 16 px][16 B]
    const int mpl = a.ntiles * 256;
    // the follow-up filter's fragments (two k-steps x this wave's two tiles) are on their way while the tile is written
    bf16x8 w2[2][2];
    const bf16x8* wp2 = reinterpret_cast<const bf16x8*>(a.w2) + (2 * mh) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int m = 0; m < 2; ++m) w2[ks][m] = wp2[(ks * 4 + m) * 64];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      bf16x8 o;
      const f32x4 s0 = a.act ? silu4(acc[0][nt]) : acc[0][nt], s1 = a.act ? silu4(acc[1][nt]) : acc[1][nt];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[r] = (__bf16)s0[r];
        o[4 + r] = (__bf16)s1[r];
      }
      *reinterpret_cast<bf16x8*>(mid + (mh * 4 + g) * mpl + ((t0 + nt) * 16 + n) * 16) = o;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias2 + chan_in_block(4, 2 * mh + m, g, 0));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[m][nt] = bv;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const bf16x8 xm = *reinterpret_cast<const bf16x8*>(mid + (ks * 4 + g) * mpl + ((t0 + nt) * 16 + n) * 16);
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[ks][m], xm, acc[m][nt], 0, 0, 0);
      }
  }
  const bool act_out = POST ? a.post_act != 0 : a.act != 0;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    float v[8];
    const f32x4 s0 = act_out ? silu4(acc[0][nt]) : acc[0][nt], s1 = act_out ? silu4(acc[1][nt]) : acc[1][nt];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = s0[r];
      v[4 + r] = s1[r];
    }
    store_b128_nt(u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])},
                  rout, opix[nt] == OOB_LANE ? OOB_LANE : (unsigned)(mh * 4 + g) * a.y_plane + opix[nt], a.st_nt);
  }
}

// KKN = input channels / 32; NTQ = ceil(tiles per band / 4): tiles of the largest pixel quarter (the others have NTQ or NTQ - 1)
// NTQ == 0: the tile count of this wave is dispatched at run time over 1 .. 5 (the 64-channel members: short k-loops, small code)
template <int KKN, int NTQ, bool POST = false>
__device__ __forceinline__ void band_body(const BandArgs& a, int block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bsm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int b = block / a.nbands, band = block - b * a.nbands;
  const int y0 = band * a.TR;
  const int mh = wave & 1;
  YL_STAMP(a.stamps, 8, wave, lane, 0);
  // the filter ring sits behind the input tile; its first slices are requested before the input is staged
  unsigned char* const ring = bsm + 4 * KKN * a.tpl;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, (unsigned)(9 * KKN) * 4096u, 0x00020000);
#pragma unroll
  for (int sl = 0; sl < BD_NS - 1; ++sl) ring_issue(rw, ring, sl, 9 * KKN / 2, wave, lane);
  // ---- the band's input rows (with halo) -> LDS: per plane ONE contiguous run of the guarded layout, starting at pixel
  // (y0 - 1, -1) of image b = plane index b * img + y0 * W1 (g0 == W1 + 1); 8 sixteen-byte loads in flight per thread
  {
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, a.x_bytes, 0x00020000);
    const int tpx = a.tpl / 16, total = 4 * KKN * tpx;
    const unsigned p0 = (unsigned)(b * a.img + y0 * a.W1) * 16u;
    for (int c0 = tid; c0 < total; c0 += BD_THREADS * 8) {
      bf16x8 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + u * BD_THREADS;
        const int pl = c / tpx, t = c - pl * tpx;
        v[u] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                              rin, c < total ? (unsigned)pl * a.x_plane + p0 + (unsigned)t * 16u : OOB_LANE, 0, 0));
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + u * BD_THREADS;
        if (c < total) *reinterpret_cast<bf16x8*>(bsm + c * 16) = v[u];
      }
    }
  }
  YL_STAMP(a.stamps, 8, wave, lane, 1);
  __syncthreads();
  YL_STAMP(a.stamps, 8, wave, lane, 2);
  int t0, cnt;
  even_split(a.ntiles, 4, wave >> 1, t0, cnt);
  // (every wave takes part in the ring's barriers: a wave without tiles still runs the loop shell)
  if constexpr (NTQ == 0) {
    switch (cnt) {
      case 5: band_compute<KKN, 5, POST>(a, bsm, ring, rw, b, y0, mh, t0, n, g, lane, wave); break;
      case 4: band_compute<KKN, 4, POST>(a, bsm, ring, rw, b, y0, mh, t0, n, g, lane, wave); break;
      case 3: band_compute<KKN, 3, POST>(a, bsm, ring, rw, b, y0, mh, t0, n, g, lane, wave); break;
      case 2: band_compute<KKN, 2, POST>(a, bsm, ring, rw, b, y0, mh, t0, n, g, lane, wave); break;
      default: band_compute<KKN, 1, POST>(a, bsm, ring, rw, b, y0, mh, t0, n, g, lane, wave); break;
    }
  } else {
    if (cnt == NTQ) band_compute<KKN, (NTQ > 0 ? NTQ : 1), POST>(a, bsm, ring, rw, b, y0, mh, t0, n, g, lane, wave);
    else band_compute<KKN, (NTQ > 1 ? NTQ - 1 : 1), POST>(a, bsm, ring, rw, b, y0, mh, t0, n, g, lane, wave);
  }
  YL_STAMP(a.stamps, 8, wave, lane, 4);
}

struct BandGroup {
  BandArgs m[2];
  int nb0;  // workgroups of member 0 (the rest belong to member 1)
};

template <int KA, int NA, int KB, int NB>
__global__ __launch_bounds__(BD_THREADS) void conv3_band_kernel(BandGroup G) {
  const int block = blockIdx.x;
  if (block < G.nb0) band_body<KA, NA>(G.m[0], block);
  else band_body<KB, NB>(G.m[1], block - G.nb0);
}

// ONE conv per launch (round 6: the first box convs of the wider models -- YOLO11s 128 -> 64 @80x80, 256 -> 64 @40x40, 512 -> 64 @20x20,
// reference head.py:36-40 cv2[i][0] -- which ran on conv_kernel at 0.10-0.23 of the matrix peak with 1.5 x their algorithmic traffic:
// its tap-major sweep of a 100 MB input does not fit any cache).  The wave's tile count is dispatched at run time.
template <int KKN>
__global__ __launch_bounds__(BD_THREADS) void conv3_band1_kernel(BandArgs a) {
  band_body<KKN, 0>(a, blockIdx.x);
}

// Tiling of one member: the largest band (rows) whose pixel tiles fit NTQ <= 5 per wave quarter and whose input fits the LDS budget
// (post = true: the single-member form -- 512 input channels too, and a band may hold as little as one tile per wave quarter)
bool conv3_band_plan(int Cin, int Cout, int H, int W, BandArgs* a, bool post) {
  const bool single = post;
  post = false;
  if (Cout != 64 || (Cin != 64 && Cin != 128 && Cin != 256 && !(single && Cin == 512))) return false;
  const int planes = Cin / 8;
  int best = 0;
  for (int TR = 1; TR <= H; ++TR) {
    const int ntiles = (TR * W + 15) / 16;
    if (ntiles > 20 || (size_t)planes * band_tpl(TR, W) + (size_t)BD_NS * BD_SLICE + (post ? (size_t)8 * ntiles * 256 : 0) > 160 * 1024) break;
    best = TR;
  }
  if (best < 1) return false;
  // even out the bands (e.g. 20 rows: two bands of 10 instead of 15 + 5)
  const int nbands = (H + best - 1) / best;
  const int TR = (H + nbands - 1) / nbands;
  const int ntiles = (TR * W + 15) / 16, ntq = (ntiles + 3) / 4;
  if ((ntq < 2 && !single) || ntq > 5) return false;  // (the instantiated tile counts; at least one tile per wave)
  if (a) {
    a->TR = TR;
    a->nbands = nbands;
    a->ntiles = ntiles;
    a->tpl = band_tpl(TR, W);
  }
  return true;
}

template <int KA, int NA, int KB, int NB>
static int band_launch_t(const BandGroup& G, int grid, size_t lds, hipStream_t s) {
  auto kern = conv3_band_kernel<KA, NA, KB, NB>;
  if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BD_THREADS), lds, s, G);
  YL_HIP(hipGetLastError());
  return 0;
}

// the two convs (members) in one launch; args[i] carries pointers / geometry / filter, the plan is made here
int conv3_band_launch(const BandArgs* args, int n, int batch, hipStream_t s) {
  if (n < 1 || n > 2) return fail(-1, "conv3_band: one or two members");
  BandGroup G{};
  size_t lds = 0;
  int grid = 0, kk[2] = {2, 2}, nq[2] = {4, 4};
  for (int i = 0; i < n; ++i) {
    G.m[i] = args[i];
    G.m[i].stamps = g_stamp_buf;
    G.m[i].st_nt = st_nt_policy(args[i].y_bytes);
    if (!conv3_band_plan(args[i].Cin, 64, args[i].H, args[i].W, &G.m[i], false)) return fail(-3, "conv3_band: member does not fit the band tiling");
    kk[i] = args[i].Cin / 32;
    nq[i] = (G.m[i].ntiles + 3) / 4;
    const size_t l = (size_t)(args[i].Cin / 8) * G.m[i].tpl + (size_t)BD_NS * BD_SLICE;
    lds = l > lds ? l : lds;
    const int nb = batch * G.m[i].nbands;
    if (i == 0) G.nb0 = nb;
    grid += nb;
  }
  if (n != 2 || kk[0] != 4 || kk[1] != 8) return fail(-3, "conv3_band: a launch is the pair (128 -> 64, 256 -> 64)");
#define YL_BAND_CASE(NA, NB) \
  case NA * 8 + NB: return band_launch_t<4, NA, 8, NB>(G, grid, lds, s);
#define YL_BAND_ROW(NA) YL_BAND_CASE(NA, 2) YL_BAND_CASE(NA, 3) YL_BAND_CASE(NA, 4) YL_BAND_CASE(NA, 5)
  switch (nq[0] * 8 + nq[1]) {
    YL_BAND_ROW(2)
    YL_BAND_ROW(3)
    YL_BAND_ROW(4)
    YL_BAND_ROW(5)
  }
#undef YL_BAND_ROW
#undef YL_BAND_CASE
  return fail(-3, "conv3_band: unsupported member tiling");
}

// one conv on conv3_band1_kernel (plan with `single`)
int conv3_band1_launch(const BandArgs& arg, int batch, hipStream_t s) {
  BandArgs a = arg;
  a.stamps = g_stamp_buf;
  a.st_nt = st_nt_policy(arg.y_bytes);
  if (!conv3_band_plan(a.Cin, 64, a.H, a.W, &a, true)) return fail(-3, "conv3_band: the conv does not fit the band tiling");
  const size_t lds = (size_t)(a.Cin / 8) * a.tpl + (size_t)BD_NS * BD_SLICE;
  const int grid = batch * a.nbands;
#define YL_BAND1(KK)                                                                                      \
  case KK: {                                                                                              \
    auto kern = conv3_band1_kernel<KK>;                                                                   \
    if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;             \
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BD_THREADS), lds, s, a);                          \
    break;                                                                                                \
  }
  switch (a.Cin / 32) {
    YL_BAND1(4)
    YL_BAND1(8)
    YL_BAND1(16)
    default: return fail(-3, "conv3_band: 128, 256 or 512 input channels");
  }
#undef YL_BAND1
  YL_HIP(hipGetLastError());
  return 0;
}

}  // namespace yl
