// sppf_resident_kernel: a whole SPPF layer -- and the 1x1 conv that alone consumes its output -- in ONE launch, per image.
//
// Reference: nn/modules/block.py:165-184 SPPF (cv1, three chained 5x5 max pools, cv2 over cat(x, y1, y2, y3)); the follow-up
// conv is C2PSA.cv1 (block.py:1009-1038) in YOLO11 (layer 9 -> layer 10); conv.py:35-53 Conv with the BN folded
// (utils/torch_utils.py:182-209).  Channels as in YOLO11n: 256 -> 128, pools, 512 -> 256 [, 256 -> 256].
//
//     x  = SiLU(cv1(in))                    1x1, 256 -> 128
//     y5, y9, y13 = max over the 5x5 / 9x9 / 13x13 window around a pixel, clipped to the image (three chained 5x5 pools)
//     s  = SiLU(cv2(cat(x, y5, y9, y13)))   1x1, 512 -> 256
//     out = SiLU(tail(s))                   1x1, 256 -> 256 (optional; then s is never stored)
//
// Four launches of the unfused program (52 us in sequence for 5.9 GFLOP per batch of 32 at 20x20) become one.  Workgroups of
// eight waves, `split` of them per image:
//   * cv1 runs over ALL pixels of the image in every workgroup of it (a wave owns an eighth of the 16-pixel tiles), its output
//     goes to LDS as ORDERED KEYS (sppf_pool_lds_kernel's trick, ops_misc.hip: a bf16 as an int16 that orders like the value,
//     so a maximum is one v_pk_max_i16 per two values): [16 planes][pixels][16 B], dense.
//   * cv2 / tail: a wave owns its pixel tiles (at most two) and ALL 256 output channels, so cv2's accumulators are the tail's
//     B fragments (conv_kernel's POST trick) and nothing between them leaves the registers.  k-steps in the unfused conv's
//     order: x (fragments = the map), then the pooled thirds.
//   * the pools: the map is turned IN PLACE (through registers, two barriers per pass) into its row maxima and those into the
//     5x5 maxima Y5; a y9 fragment is then the maximum of Y5 at the four window corners (y +- 2, x +- 2, clamped to the
//     image -- the union of the four clipped 5x5 windows is the clipped 9x9 window), a y13 fragment of Y5 at the nine points
//     (y + {-4, 0, 4}, x + {-4, 0, 4}): the same set of values as the chained pools, hence the same maximum.
//   * the filters of cv2 and the tail (256 + 128 KB) stream through a three-slot LDS ring by LDS-DMA, one k-step (all 16 tiles,
//     16 KB) per slot, each byte fetched once per workgroup; one barrier per k-step (conv3_band.hip's protocol).
// Arithmetic is conv_kernel's (packed filters, k order, fp32 chain from the bias, silu_f, one rounding to bf16 per layer) and
// sppf_pool_lds_kernel's: BIT-IDENTICAL to the launches it replaces (tests/test_gpu_resident.py).
#include "lds_conv.h"

namespace yl {

typedef __attribute__((address_space(3))) void lds_void_s;
typedef short i16x8 __attribute__((ext_vector_type(8)));

constexpr int SP_THREADS = 512;
constexpr int SP_WAVES = SP_THREADS / 64;
constexpr int SP_SLICE = 16 * 1024;  // one k-step of a 256-output-channel filter: [4 blocks][4 tiles][64 lanes][16 B]
constexpr int SP_NS = 3;

__device__ __forceinline__ i16x8 sp_key(i16x8 x) { return x ^ ((x >> 15) & (short)0x7fff); }  // its own inverse
__device__ __forceinline__ i16x8 sp_max(i16x8 a, i16x8 b) { return __builtin_elementwise_max(a, b); }

template <int NT>
struct SpTiles {
  int pix[NT];        // dense pixel index of the lane's pixel (clamped into the image)
  unsigned gpix[NT];  // byte offset of the lane's pixel in a plane of the HBM layout (OOB_LANE: none)
  int ty[NT], tx[NT];
};
template <int NT>
__device__ __forceinline__ void sp_tiles(const SppfArgs& a, int b, int t0, int cnt, int n, SpTiles<NT>& T) {
  const int N = a.H * a.W;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int i = (t0 + nt) * 16 + n;
    const bool ok = nt < cnt && i < N;
    T.pix[nt] = ok ? i : N - 1;
    const int y = T.pix[nt] / a.W, x = T.pix[nt] - y * a.W;
    T.ty[nt] = y;
    T.tx[nt] = x;
    T.gpix[nt] = ok ? (unsigned)(a.g0 + b * a.img + y * a.W1 + x) * 16u : OOB_LANE;
  }
}

// slice sl of the filter stream -> ring slot sl % 3: k-step sl of cv2 (sl < 16), k-step sl - 16 of the tail.  This wave's two of
// the slice's sixteen 1-KiB pieces (piece = 64-channel block * 4 + tile).  Past the end: an out-of-range offset (moves zeros
// into a slot nobody reads any more, and keeps the per-wave operation count of the waits uniform).
template <bool TAIL>
__device__ __forceinline__ void sp_ring_issue(const SppfArgs& a, __amdgpu_buffer_rsrc_t rw2, __amdgpu_buffer_rsrc_t rw3, unsigned char* ring,
                                              int sl, int wave, int lane) {
  constexpr int NSL = TAIL ? 24 : 16;
  const bool t = TAIL && sl >= 16;
  const int ks = t ? sl - 16 : sl, nks = t ? 8 : 16;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int piece = wave * 2 + q, cb = piece >> 2, m = piece & 3;
    const unsigned off = sl < NSL ? (unsigned)(((cb * nks + ks) * 4 + m) * 1024 + lane * 16) : OOB_LANE;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(t ? rw3 : rw2, (lds_void_s*)(ring + (sl % SP_NS) * SP_SLICE + piece * 1024), 16, (int)off, 0, 0, 0);
  }
}

// one k-step of a 256-output-channel 1x1 conv for this wave's NT tiles: wait for the slice, refill the slot behind it, 16 x NT MFMAs
template <int NT, bool TAIL>
__device__ __forceinline__ void sp_step(const SppfArgs& a, __amdgpu_buffer_rsrc_t rw2, __amdgpu_buffer_rsrc_t rw3, unsigned char* ring, int sl,
                                        int wave, int lane, f32x4 (&acc)[16][NT], const bf16x8 (&frag)[NT]) {
  // my own pieces of slice sl have landed (slice sl + 1's two operations may still be in flight); after the barrier everybody's
  // have, and nobody reads slice sl - 1 any more: its slot takes slice sl + 2
  if (!(a.dbg & 1)) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  if (!(a.dbg & 2)) __builtin_amdgcn_s_barrier();
  if (!(a.dbg & 1)) sp_ring_issue<TAIL>(a, rw2, rw3, ring, sl + 2, wave, lane);
  const unsigned char* slot = ring + ((a.dbg & 4) ? 0 : (sl % SP_NS) * SP_SLICE) + lane * 16;
  // the A fragments of 64-channel block cb + 1 are requested BEFORE block cb's MFMAs: a wave keeps the matrix pipe busy by itself
  // (with one set of registers the reads followed the MFMAs that freed it, and the two waves of a SIMD, in lockstep behind the
  // barrier, both waited for LDS at the same time: 0.93 us per k-step against 0.43 of MFMA time)
  bf16x8 wa[2][4];
#pragma unroll
  for (int m = 0; m < 4; ++m) wa[0][m] = *reinterpret_cast<const bf16x8*>(slot + m * 1024);
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    if (cb + 1 < 4) {
#pragma unroll
      for (int m = 0; m < 4; ++m) wa[(cb + 1) & 1][m] = *reinterpret_cast<const bf16x8*>(slot + ((cb + 1) * 4 + m) * 1024);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[cb * 4 + m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cb & 1][m], frag[nt], acc[cb * 4 + m][nt], 0, 0, 0);
  }
}

template <int NT>
__device__ __forceinline__ void sp_bias16(f32x4 (&acc)[16][NT], const float* bias, int g) {
#pragma unroll
  for (int m = 0; m < 16; ++m) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + (m >> 2) * 64 + chan_in_block(4, m & 3, g, 0));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[m][nt] = bv;
  }
}

// SiLU + one rounding to bf16: the accumulators of tile pair (2 q, 2 q + 1) = the 8 channels of plane 4 q' + g the lane holds
template <int NT>
__device__ __forceinline__ void sp_pack16(const f32x4 (&acc)[16][NT], bf16x8 (&pk)[8][NT]) {
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pk[q][nt][r] = (__bf16)silu_f(acc[2 * q][nt][r]);
        pk[q][nt][4 + r] = (__bf16)silu_f(acc[2 * q + 1][nt][r]);
      }
}

// NT1: tiles per wave of cv1 (all pixels of the image, an eighth per wave); NT2: tiles per wave of this workgroup's own share
template <int NT1, int NT2, bool TAIL>
__global__ __launch_bounds__(SP_THREADS) void sppf_resident_kernel(SppfArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ssm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int N = a.H * a.W, ntiles = (N + 15) / 16;
  const int mpl = ntiles * 256;  // bytes per plane of the LDS map
  unsigned char* const map = ssm;
  unsigned char* const ring = ssm + 16 * mpl;
  // cv1's filter half for channels 64-127 lies in the map's planes 0-7 while those channels' epilogue writes planes 8-15; on a
  // small map (planes 0-7 shorter than 32 KB) it gets an area of its own behind the ring instead (sp_lds_bytes)
  unsigned char* const wtop = 8 * mpl >= 32768 ? ssm : ring + SP_NS * SP_SLICE;
  const int b = blockIdx.x / a.split, part = blockIdx.x - b * a.split;
#define SP_STAMP(id) YL_STAMP(a.stamps, SP_WAVES, wave, lane, id)
  SP_STAMP(0);
  const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2, (short)0, 256u * 1024u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw3 = __builtin_amdgcn_make_buffer_rsrc((void*)(TAIL ? a.w3 : a.w2), (short)0, 128u * 1024u, 0x00020000);

  // ---- cv1 over the whole image, 64 output channels at a time.  Its filter (64 KB) is staged by LDS-DMA -- streamed from L2 by
  // every wave it would be 512 KB of L1 traffic per workgroup, as long as the MFMAs take: the half for channels 64-127 into the
  // map's first planes, the half for channels 0-63 into the ring (idle until the k-loop of cv2).  Channels 64-127 go first: their
  // epilogue writes planes 8-15, clear of the filter half the other waves may still be reading; one barrier, then channels 0-63
  // from the ring, writing planes 0-7.  The B fragments (the layer's input, from HBM) are all requested up front.
  {
    const __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w1, (short)0, 64u * 1024u, 0x00020000);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int piece = wave * 4 + q;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw1, (lds_void_s*)(wtop + piece * 1024), 16, (int)(32768 + piece * 1024 + lane * 16), 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw1, (lds_void_s*)(ring + piece * 1024), 16, (int)(piece * 1024 + lane * 16), 0, 0, 0);
    }
    int t0a, cnta;
    even_split(ntiles, SP_WAVES, wave, t0a, cnta);
    SpTiles<NT1> TA;
    sp_tiles<NT1>(a, b, t0a, cnta, n, TA);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, a.x_bytes, 0x00020000);
    bf16x8 xb[8][NT1];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int nt = 0; nt < NT1; ++nt)
        xb[ks][nt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                    rx, TA.gpix[nt] == OOB_LANE ? OOB_LANE : (unsigned)(ks * 4 + g) * a.x_plane + TA.gpix[nt], 0, 0));
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * NT1) : "memory");  // my pieces of the filter (the B fragments may still be on their way)
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int cb = 1 - h;
      const unsigned char* wsrc = (h == 0 ? wtop : ring) + lane * 16;  // [ks 8][4 tiles][64 lanes][16 B]
      f32x4 acc[4][NT1];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b1 + cb * 64 + chan_in_block(4, m, g, 0));
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) acc[m][nt] = bv;
      }
      bf16x8 wa[2][4];
#pragma unroll
      for (int m = 0; m < 4; ++m) wa[0][m] = *reinterpret_cast<const bf16x8*>(wsrc + m * 1024);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks + 1 < 8) {
#pragma unroll
          for (int m = 0; m < 4; ++m) wa[(ks + 1) & 1][m] = *reinterpret_cast<const bf16x8*>(wsrc + ((ks + 1) * 4 + m) * 1024);
        }
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
          for (int m = 0; m < 4; ++m) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks & 1][m], xb[ks][nt], acc[m][nt], 0, 0, 0);
      }
      if (h == 1) __syncthreads();  // (before the first write to planes 0-7: every wave is done with the filter half that lies there)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) {
          bf16x8 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o[r] = (__bf16)silu_f(acc[2 * p][nt][r]);
            o[4 + r] = (__bf16)silu_f(acc[2 * p + 1][nt][r]);
          }
          if (TA.gpix[nt] != OOB_LANE)
            *reinterpret_cast<i16x8*>(map + ((cb * 2 + p) * 4 + g) * mpl + TA.pix[nt] * 16) = sp_key(__builtin_bit_cast(i16x8, o));
        }
    }
  }
  SP_STAMP(1);
  __syncthreads();  // the map is complete; nobody reads cv1's filter any more: the ring takes the first two slices of cv2's
  sp_ring_issue<TAIL>(a, rw2, rw3, ring, 0, wave, lane);
  sp_ring_issue<TAIL>(a, rw2, rw3, ring, 1, wave, lane);
  SP_STAMP(2);

  // ---- this workgroup's share of the image, this wave's tiles of it
  int p0, pc, t0, cnt;
  even_split(ntiles, a.split, part, p0, pc);
  even_split(pc, SP_WAVES, wave, t0, cnt);
  SpTiles<NT2> T;
  sp_tiles<NT2>(a, b, p0 + t0, cnt, n, T);
  // cv2's k-steps 0-3 read x: this wave's fragments of it leave the map before the pools change it (the k-loop starts behind the
  // pools: sixteen accumulator tiles live across them would spill)
  bf16x8 xf[4][NT2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt)
      xf[j][nt] = __builtin_bit_cast(bf16x8, sp_key(*reinterpret_cast<const i16x8*>(map + (4 * j + g) * mpl + T.pix[nt] * 16)));
  __syncthreads();  // every wave has its x fragments: the map may change

  // ---- the map -> its 5x5 maxima, in place: rows, then columns; a thread's items pass through its registers
  {
    // a thread = one pixel x 16 / tpp planes (tpp threads per pixel, a power of two: 1 for the 400 pixels of a 20x20 map); no index
    // arithmetic per item -- the first version dealt (plane, pixel) items round-robin and spent its time in integer divisions
    int sh = 0;
    while (sh < 4 && (N << (sh + 1)) <= SP_THREADS) ++sh;
    const int p = tid >> sh, q = tid & ((1 << sh) - 1), npl = 16 >> sh;
    const bool act = p < N;
    const int pc = act ? p : 0;
    const int y = pc / a.W, x = pc - y * a.W;
    unsigned char* const base = map + (q * npl) * mpl + pc * 16;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int c = pass == 0 ? x : y, lim = pass == 0 ? a.W : a.H, st = pass == 0 ? 16 : a.W * 16;
      i16x8 r[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j < npl) {
          const unsigned char* o = base + j * mpl;
          i16x8 m = *reinterpret_cast<const i16x8*>(o);
#pragma unroll
          for (int d = 1; d <= 2; ++d) {
            if (c - d >= 0) m = sp_max(m, *reinterpret_cast<const i16x8*>(o - d * st));
            if (c + d < lim) m = sp_max(m, *reinterpret_cast<const i16x8*>(o + d * st));
          }
          r[j] = m;
        }
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 16; ++j)
        if (j < npl && act) *reinterpret_cast<i16x8*>(base + j * mpl) = r[j];
      __syncthreads();
    }
  }
  SP_STAMP(3);
  f32x4 acc2[16][NT2];
  sp_bias16<NT2>(acc2, a.b2, g);
#pragma unroll
  for (int j = 0; j < 4; ++j) sp_step<NT2, TAIL>(a, rw2, rw3, ring, j, wave, lane, acc2, xf[j]);
  SP_STAMP(4);
  // k-steps 4-7: y5 = the map; 8-11: y9 = four corners; 12-15: y13 = nine points
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bf16x8 frag[NT2];
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt)
      frag[nt] = __builtin_bit_cast(bf16x8, sp_key(*reinterpret_cast<const i16x8*>(map + (4 * j + g) * mpl + T.pix[nt] * 16)));
    sp_step<NT2, TAIL>(a, rw2, rw3, ring, 4 + j, wave, lane, acc2, frag);
  }
  {
    int c9[NT2][4];
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) {
      const int ym = max(T.ty[nt] - 2, 0), yp = min(T.ty[nt] + 2, a.H - 1), xm = max(T.tx[nt] - 2, 0), xp = min(T.tx[nt] + 2, a.W - 1);
      c9[nt][0] = (ym * a.W + xm) * 16;
      c9[nt][1] = (ym * a.W + xp) * 16;
      c9[nt][2] = (yp * a.W + xm) * 16;
      c9[nt][3] = (yp * a.W + xp) * 16;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x8 frag[NT2];
#pragma unroll
      for (int nt = 0; nt < NT2; ++nt) {
        const unsigned char* pl = map + (4 * j + g) * mpl;
        i16x8 m = *reinterpret_cast<const i16x8*>(pl + c9[nt][0]);
#pragma unroll
        for (int q = 1; q < 4; ++q) m = sp_max(m, *reinterpret_cast<const i16x8*>(pl + c9[nt][q]));
        frag[nt] = __builtin_bit_cast(bf16x8, sp_key(m));
      }
      sp_step<NT2, TAIL>(a, rw2, rw3, ring, 8 + j, wave, lane, acc2, frag);
    }
  }
  {
    int ry[NT2][3], cx[NT2][3];
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) {
      ry[nt][0] = max(T.ty[nt] - 4, 0) * a.W * 16;
      ry[nt][1] = T.ty[nt] * a.W * 16;
      ry[nt][2] = min(T.ty[nt] + 4, a.H - 1) * a.W * 16;
      cx[nt][0] = max(T.tx[nt] - 4, 0) * 16;
      cx[nt][1] = T.tx[nt] * 16;
      cx[nt][2] = min(T.tx[nt] + 4, a.W - 1) * 16;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x8 frag[NT2];
#pragma unroll
      for (int nt = 0; nt < NT2; ++nt) {
        const unsigned char* pl = map + (4 * j + g) * mpl;
        i16x8 m = *reinterpret_cast<const i16x8*>(pl + ry[nt][0] + cx[nt][0]);
#pragma unroll
        for (int q = 1; q < 9; ++q) m = sp_max(m, *reinterpret_cast<const i16x8*>(pl + ry[nt][q / 3] + cx[nt][q % 3]));
        frag[nt] = __builtin_bit_cast(bf16x8, sp_key(m));
      }
      sp_step<NT2, TAIL>(a, rw2, rw3, ring, 12 + j, wave, lane, acc2, frag);
    }
  }
  SP_STAMP(5);
  // ---- SiLU, one rounding: cv2's output as planes of this wave's pixels = the tail's B fragments
  bf16x8 pk[8][NT2];
  sp_pack16<NT2>(acc2, pk);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, (short)0, a.y_bytes, 0x00020000);
  if constexpr (TAIL) {
    __builtin_amdgcn_sched_barrier(0);
    sp_bias16<NT2>(acc2, a.b3, g);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) sp_step<NT2, TAIL>(a, rw2, rw3, ring, 16 + ks, wave, lane, acc2, pk[ks]);
    SP_STAMP(6);
    sp_pack16<NT2>(acc2, pk);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q)
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pk[q][nt]), rout,
                                             T.gpix[nt] == OOB_LANE ? OOB_LANE : (unsigned)(q * 4 + g) * a.y_plane + T.gpix[nt], 0, 0);
  // (the ring's last refills moved nothing, but an LDS-DMA operation in flight at the end of the workgroup would land in LDS that
  // already belongs to another one: they retire here)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SP_STAMP(7);
#undef SP_STAMP
}

static int sp_lds_bytes(int H, int W) {
  const int mpl = ((H * W + 15) / 16) * 256;
  return 16 * mpl + SP_NS * SP_SLICE + (8 * mpl >= 32768 ? 0 : 32768);
}

// Whether the resident kernel takes an SPPF layer c1 -> c2 (hidden width c1 / 2) on an H x W map, with a follow-up 1x1 conv to
// `tail` channels (0: none)
bool sppf_resident_fits(int c1, int c2, int tail, int H, int W) {
  const int N = H * W, ntiles = (N + 15) / 16;
  return c1 == 256 && c2 == 256 && (tail == 0 || tail == 256) && N >= 1 && ntiles <= 28 && sp_lds_bytes(H, W) <= 160 * 1024;
}

template <int NT1, int NT2, bool TAIL>
static int sp_launch_t(const SppfArgs& a, size_t lds, hipStream_t s) {
  auto kern = sppf_resident_kernel<NT1, NT2, TAIL>;
  if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * a.split)), dim3(SP_THREADS), lds, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

int sppf_resident_launch(const SppfArgs& a0, hipStream_t s) {
  SppfArgs a = a0;
  a.stamps = g_stamp_buf;
  static const int dbg = tune_int("YL_SPPF_DBG", 0);  // development: timing ablations (wrong results)
  a.dbg = dbg;
  if (!sppf_resident_fits(256, 256, a.w3 ? 256 : 0, a.H, a.W)) return fail(-3, "sppf_resident: the map does not fit one workgroup's LDS");
  const int ntiles = (a.H * a.W + 15) / 16;
  // workgroups per image: a wave holds all 256 output channels of its tiles (64 accumulator registers per tile), two tiles at most
  a.split = ntiles > 16 ? 2 : 1;
  const int nt1 = (ntiles + 7) / 8, nt2 = ((ntiles + a.split - 1) / a.split + 7) / 8;
  const size_t lds = (size_t)sp_lds_bytes(a.H, a.W);
  const bool tail = a.w3 != nullptr;
#define SP_CASE(A, B)                                                                        \
  case A * 10 + B:                                                                           \
    return tail ? sp_launch_t<A, B, true>(a, lds, s) : sp_launch_t<A, B, false>(a, lds, s);
  switch (nt1 * 10 + nt2) {
    SP_CASE(1, 1)
    SP_CASE(2, 2)
    SP_CASE(3, 2)
    SP_CASE(4, 2)
  }
#undef SP_CASE
  return fail(-3, "sppf_resident: unsupported map size");
}

}  // namespace yl
