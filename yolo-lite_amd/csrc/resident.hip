// Per-image RESIDENT stage kernels for the small maps (20x20 at 640x640): one workgroup owns one image's map in LDS and runs a
// whole chain of convs on it, workgroup barriers between the layers instead of kernel boundaries.
//
// c3k_resident_kernel: the body of a C3k block (reference nn/modules/block.py:731-739 C3k(C3), :245-259 C3.forward,
// :330-343 Bottleneck, conv.py:35-53 Conv with the BN folded per utils/torch_utils.py:182-209), c_ = 64 channels inside:
//
//     u  = SiLU(cv1|cv2(x))            1x1, 128 -> 128 (C3.cv1 and C3.cv2 stacked: both read x; a = u[0:64], b = u[64:128])
//     a  = a + SiLU(m0.cv2(SiLU(m0.cv1(a))))     two 3x3 convs 64 -> 64 and the shortcut add
//     a  = a + SiLU(m1.cv2(SiLU(m1.cv1(a))))
//     y  = SiLU(cv3(cat(a, b)))        1x1, 128 -> 128
//
// Six launches of the unfused program (layers 8 and 22 of YOLO11n: 70 us in sequence for 4.6 GFLOP per batch of 32, every one
// of them a 9-14 us launch whatever its work) become one.  Eight waves (two per SIMD); x, a and the tensor between a
// Bottleneck's convs live in LDS, b never leaves the registers of the wave that computed it:
//
//   LDS   region 0: x, dense [16 planes][HWp px][16 B]  (HWp = pixels rounded up to 16)     -> later T, guarded [8 planes]
//         region 1: A, guarded [8 planes][(H + 2) * (W + 1) + 2 px][16 B]
//   The guarded plane is the activation layout of HBM cut down to one image (yl_internal.h): a 3x3 tap is a constant byte
//   offset that lands on data or on a zero, so the B fragments of the 3x3 convs are plain ds_read_b128 of 16 consecutive
//   pixels (256 contiguous bytes per lane group: conflict-free), no predicates.
//   Filters come straight from L2 into registers (packed fragment order, conv_pack_weights with four tiles per block),
//   three k-steps ahead; all 32 workgroups of a batch read the same bytes.
//   1x1 convs: a wave owns an EIGHTH of the pixel tiles and every output channel (64 at a time): the b half of u stays in
//             its registers as the B fragments of cv3's last two k-steps.
//   3x3 convs: a wave owns a QUARTER of the pixel tiles and HALF of the output channels (M-half x N-quarter: 2 filter loads
//             and 7 LDS reads per 14 MFMAs), the two waves of a SIMD taking different quarters.
// Arithmetic is conv_kernel's: the same packed filters, k order (tap-major, 32 channels per k-step), fp32 accumulation
// chain starting from the bias, SiLU (silu_f), residual add after the activation, one rounding to bf16 per layer --
// results are BIT-IDENTICAL to the six launches (tests/test_gpu_resident.py).
#include "lds_conv.h"

namespace yl {

typedef __attribute__((address_space(3))) void lds_void_r;

constexpr int RC_THREADS = 512;
constexpr int RC_WAVES = RC_THREADS / 64;

struct PixMap {
  int H, W, W1, HW;
  int o0, o1;  // rows of this map the workgroup stores (the whole map unless the image is split between workgroups)
  int goff;    // pixels between the map's first row and the image's first row in the HBM plane (row offset * W1)
};
// dense pixel index -> guarded index inside one image's plane (0 = the pixel above-left of (0, 0)); clamped for i >= HW
__device__ __forceinline__ int guarded_index(const PixMap& m, int i, bool& valid) {
  valid = i < m.HW;
  const int ic = valid ? i : m.HW - 1;
  const int y = ic / m.W, x = ic - y * m.W;
  return m.W1 + 1 + y * m.W1 + x;
}

// ---- one 3x3 conv 64 -> 64 from a guarded LDS map: this wave's M-half (tiles 2 mh, 2 mh + 1 of the 4-tile filter block) x NT
// pixel tiles.  src: LDS region + this lane's k-group plane (g * apl).  tb[nt]: byte offset of the lane's pixel in a plane.
// A 3x3 step for this wave: acc = bias; conv from `src` map; SiLU; optional residual from map `res` (same plane / pixel the lane
// writes); bf16; 16-byte LDS store into map `dst`.
template <int NT, int NTQ>
__device__ __forceinline__ void step3(const C3kArgs& a, const PixMap& pm, unsigned char* lds, int src_off, int dst_off, int res_off,
                                      const __bf16* w, const float* bias, int mh, const unsigned (&tbq)[NTQ], unsigned okmask, int g,
                                      int lane, bf16x8 (&wa)[RC_D][2]) {
  unsigned tb[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) tb[nt] = tbq[nt];
  f32x4 acc[2][NT];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + chan_in_block(4, 2 * mh + m, g, 0));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[m][nt] = bv;
  }
  conv3_from_lds<NT>(acc, wa, reinterpret_cast<const bf16x8*>(w) + (2 * mh) * 64 + lane, lds + src_off + g * a.apl, tb, pm.W1, a.apl);
  const int pl = (mh * 4 + g) * a.apl;  // the lane's 8 output channels = plane mh * 4 + g
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    float v[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = silu_f(acc[0][nt][r]);
      v[4 + r] = silu_f(acc[1][nt][r]);
    }
    if (res_off >= 0) {
      const bf16x8 rv = *reinterpret_cast<const bf16x8*>(lds + res_off + pl + (int)tb[nt]);
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] += (float)rv[r];
    }
    bf16x8 o;
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = (__bf16)v[r];
    if ((okmask >> nt) & 1u) *reinterpret_cast<bf16x8*>(lds + dst_off + pl + (int)tb[nt]) = o;
  }
}

// The 1x1 convs (128 -> 128) run as the sequence j = (channel block cb, k-step ks) = 0 .. 7 with the filter fragments of three
// j ahead in registers; ring1 is filled BEFORE the barrier in front of the step.
constexpr int RC_D1 = 3;
__device__ __forceinline__ void preload1(bf16x8 (&wa)[RC_D1][4], const bf16x8* __restrict__ wp) {
#pragma unroll
  for (int d = 0; d < RC_D1; ++d)
#pragma unroll
    for (int m = 0; m < 4; ++m) wa[d][m] = wp[(d * 4 + m) * 64];
}

// cv1|cv2 (1x1, 128 -> 128) on this wave's NT tiles: channels [0, 64) -> map A (region 1), [64, 128) -> breg (B fragments of cv3)
template <int NT, int NTE>
__device__ __forceinline__ void step_in(const C3kArgs& a, const PixMap& pm, unsigned char* lds, int t0, int n, int g, int lane,
                                        bf16x8 (&breg)[2][NTE], bf16x8 (&wa)[RC_D1][4]) {
  // x fragments: read per k-step, one step ahead (both channel blocks read them: 2 x 4 LDS reads per tile instead of 64 registers)
  bf16x8 xb[2][NT];
  auto load_x = [&](int ks, bf16x8 (&dst)[NT]) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      dst[nt] = *reinterpret_cast<const bf16x8*>(lds + (ks * 4 + g) * a.xpl + ((t0 + nt) * 16 + n) * 16);
  };
  load_x(0, xb[0]);
  const bf16x8* wp = reinterpret_cast<const bf16x8*>(a.w[0]) + lane;
  int gi[NT];
  bool ok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) gi[nt] = guarded_index(pm, (t0 + nt) * 16 + n, ok[nt]);
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    f32x4 acc[4][NT];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias[0] + cb * 64 + chan_in_block(4, m, g, 0));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[m][nt] = bv;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int j = cb * 4 + ks;
      if (j + 1 < 8) load_x((ks + 1) & 3, xb[(j + 1) & 1]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j % RC_D1][m], xb[j & 1][nt], acc[m][nt], 0, 0, 0);
      if (j + RC_D1 < 8) {
#pragma unroll
        for (int m = 0; m < 4; ++m) wa[j % RC_D1][m] = wp[((j + RC_D1) * 4 + m) * 64];
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        bf16x8 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = (__bf16)silu_f(acc[2 * p][nt][r]);
          o[4 + r] = (__bf16)silu_f(acc[2 * p + 1][nt][r]);
        }
        if (cb == 0) {
          if (ok[nt]) *reinterpret_cast<bf16x8*>(lds + a.r1 + (p * 4 + g) * a.apl + gi[nt] * 16) = o;
        } else {
          breg[p][nt] = o;
        }
      }
  }
}

// cv3 (1x1, 128 -> 128): k-steps 0, 1 from map A, k-steps 2, 3 from breg; 16-byte stores into the output planes in HBM
template <int NT, int NTE>
__device__ __forceinline__ void step_out(const C3kArgs& a, const PixMap& pm, const unsigned char* lds, int t0, int n, int g, int lane,
                                         int b, const bf16x8 (&breg)[2][NTE], bf16x8 (&wa)[RC_D1][4]) {
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, (short)0, a.y_bytes, 0x00020000);
  bf16x8 xa[2][NT];
  unsigned opix[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    bool ok;
    const int pi = (t0 + nt) * 16 + n;
    const int gi = guarded_index(pm, pi, ok);
    const int py = pi / pm.W;
    ok = ok && py >= pm.o0 && py < pm.o1;
    opix[nt] = ok ? (unsigned)(b * a.img + pm.goff + gi) * 16u : OOB_LANE;  // (g0 of the HBM layout == W1 + 1 == the LDS map's origin)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) xa[ks][nt] = *reinterpret_cast<const bf16x8*>(lds + a.r1 + (ks * 4 + g) * a.apl + gi * 16);
  }
  const bf16x8* wp = reinterpret_cast<const bf16x8*>(a.w[5]) + lane;
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    f32x4 acc[4][NT];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias[5] + cb * 64 + chan_in_block(4, m, g, 0));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[m][nt] = bv;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int j = cb * 4 + ks;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const bf16x8 xf = ks < 2 ? xa[ks & 1][nt] : breg[ks & 1][nt];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[j % RC_D1][m], xf, acc[m][nt], 0, 0, 0);
      }
      if (j + RC_D1 < 8) {
#pragma unroll
        for (int m = 0; m < 4; ++m) wa[j % RC_D1][m] = wp[((j + RC_D1) * 4 + m) * 64];
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = silu_f(acc[2 * p][nt][r]);
          v[4 + r] = silu_f(acc[2 * p + 1][nt][r]);
        }
        __builtin_amdgcn_raw_buffer_store_b128(
            u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])}, rout,
            opix[nt] == OOB_LANE ? OOB_LANE : (unsigned)(cb * 8 + p * 4 + g) * a.y_plane + opix[nt], 0, 0);
      }
  }
}

// NTQ = ceil(ntiles / 4): tiles of the largest pixel quarter (the others have NTQ or NTQ - 1); NTE = ceil(ntiles / 8)
template <int NTQ>
__global__ __launch_bounds__(RC_THREADS) void c3k_resident_kernel(C3kArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rsm[];
  constexpr int NTE = (NTQ + 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  // a.split workgroups per image: each runs the block on a.Hs rows of it -- its share of the output rows + a halo of four (four
  // chained 3x3 convs; the rows nearer to the cut than their conv's depth see zeros for their missing neighbours and are never
  // stored) -- as if those rows were the image.  Same arithmetic per stored pixel: bit-identical to one workgroup per image.
  const int b = blockIdx.x / a.split, part = blockIdx.x - b * a.split;
  const int half = (a.H + 1) / 2;
  const int ys = a.split == 1 ? 0 : (part == 0 ? 0 : a.H - a.Hs);
  PixMap pm{a.Hs, a.W, a.W1, a.Hs * a.W, a.split == 1 ? 0 : (part == 0 ? 0 : half - ys), a.split == 1 ? a.Hs : (part == 0 ? half : a.Hs),
            ys * a.W1};
  const int HWp = a.ntiles * 16;
#define RC_STAMP(id) YL_STAMP(a.stamps, RC_WAVES, wave, lane, id)
  RC_STAMP(0);

  // ---- 0. x -> region 0 (dense planes; one pixel per thread, its 16 planes in flight together); A's guard pixels := 0.
  // Behind the x loads every filter byte of the block is touched once (one dword per 128-byte line, by LDS-DMA into a
  // scratch word: no registers, nothing waits for it): the filters were last read a whole network step ago and have left
  // L2 by now; the k-loops below then find them there instead of paying an HBM round trip per k-step.
  const int mh = wave & 1;
  int tq0, tqcnt;
  even_split(a.ntiles, 4, wave >> 1, tq0, tqcnt);
  unsigned tbq[NTQ], okq = 0;  // this wave's pixel tiles of the 3x3 steps: byte offset of the lane's pixel in a guarded plane
#pragma unroll
  for (int nt = 0; nt < NTQ; ++nt) {
    bool ok;
    tbq[nt] = (unsigned)guarded_index(pm, (tq0 + nt) * 16 + n, ok) * 16u;
    okq |= (ok && nt < tqcnt) ? (1u << nt) : 0u;
  }
  bf16x8 wa1[RC_D1][4];
  preload1(wa1, reinterpret_cast<const bf16x8*>(a.w[0]) + lane);
  {
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, (short)0, a.x_bytes, 0x00020000);
    bool ok;
    const int gi = guarded_index(pm, tid, ok);
    const unsigned src = ok ? (unsigned)(b * a.img + pm.goff + gi) * 16u : OOB_LANE;
    bf16x8 v[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
      v[p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rin, src == OOB_LANE ? OOB_LANE : (unsigned)p * a.x_plane + src, 0, 0));
    if (!(a.dbg & 1)) {
      // warm-up: filter q has wbytes[q] bytes; thread t touches lines t, t + 512, ...
#pragma unroll 1
      for (int q = 0; q < 6; ++q) {
        const int nlines = (q == 0 || q == 5 ? 32768 : 73728) / 128;
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w[q], (short)0, (unsigned)nlines * 128u, 0x00020000);
        for (int l = tid; l < nlines; l += RC_THREADS)  // (the LDS destination is M0-based and the same for every lane group: a scratch area)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_r*)(rsm + a.scratch), 4, (unsigned)l * 128u, 0, 0, 0);
      }
    }
    // guard pixels of map A (everything of a guarded plane that is not pixel (y, x) of the image)
    for (int q = tid; q * 16 < a.apl; q += RC_THREADS) {
      const int r = q - (pm.W1 + 1), y = r / pm.W1, x = r - y * pm.W1;
      if (!(r >= 0 && y < pm.H && x < pm.W)) {
        const bf16x8 z = {};
#pragma unroll
        for (int p = 0; p < 8; ++p) *reinterpret_cast<bf16x8*>(rsm + a.r1 + p * a.apl + q * 16) = z;
      }
    }
    if (tid < HWp) {
#pragma unroll
      for (int p = 0; p < 16; ++p) *reinterpret_cast<bf16x8*>(rsm + p * a.xpl + tid * 16) = v[p];
    }
  }
  RC_STAMP(1);
  __syncthreads();
  RC_STAMP(2);

  // ---- 1. u = SiLU(cv1|cv2(x)): a -> map A, b -> registers
  int te0, tecnt;
  even_split(a.ntiles, RC_WAVES, wave, te0, tecnt);
  bf16x8 breg[2][NTE];
  if (tecnt == NTE) step_in<NTE, NTE>(a, pm, rsm, te0, n, g, lane, breg, wa1);
  else if (NTE > 1 && tecnt == NTE - 1) step_in<(NTE > 1 ? NTE - 1 : 1), NTE>(a, pm, rsm, te0, n, g, lane, breg, wa1);
  bf16x8 wa[RC_D][2];
  preload3<RC_D>(wa, reinterpret_cast<const bf16x8*>(a.w[1]) + (2 * mh) * 64 + lane);
  RC_STAMP(3);
  __syncthreads();  // everybody is done with x
  RC_STAMP(4);

  // ---- region 0 becomes map T: its guard pixels must read as zero (the interior is written by the first 3x3 step)
  for (int q = tid; q * 16 < a.apl; q += RC_THREADS) {
    const int r = q - (pm.W1 + 1), y = r / pm.W1, x = r - y * pm.W1;
    if (!(r >= 0 && y < pm.H && x < pm.W)) {
      const bf16x8 z = {};
#pragma unroll
      for (int p = 0; p < 8; ++p) *reinterpret_cast<bf16x8*>(rsm + p * a.apl + q * 16) = z;
    }
  }
  RC_STAMP(5);

  // ---- 2-5. the two Bottlenecks: A -> T, T (+ A) -> A, twice.  The next step's first filter fragments are requested before
  // the barrier that ends a step.
  auto run3 = [&](int s) __attribute__((always_inline)) {
    const int src_off = (s & 1) ? 0 : a.r1, dst_off = (s & 1) ? a.r1 : 0, res_off = (s & 1) ? a.r1 : -1;
    if (tqcnt == NTQ) step3<NTQ, NTQ>(a, pm, rsm, src_off, dst_off, res_off, a.w[1 + s], a.bias[1 + s], mh, tbq, okq, g, lane, wa);
    else if (NTQ > 1 && tqcnt == NTQ - 1)
      step3<(NTQ > 1 ? NTQ - 1 : 1), NTQ>(a, pm, rsm, src_off, dst_off, res_off, a.w[1 + s], a.bias[1 + s], mh, tbq, okq, g, lane, wa);
  };
#pragma unroll 1
  for (int s = 0; s < 3; ++s) {
    run3(s);
    preload3<RC_D>(wa, reinterpret_cast<const bf16x8*>(a.w[2 + s]) + (2 * mh) * 64 + lane);
    RC_STAMP(6 + 2 * s);
    __syncthreads();
    RC_STAMP(7 + 2 * s);
  }
  run3(3);
  bf16x8 wa6[RC_D1][4];
  preload1(wa6, reinterpret_cast<const bf16x8*>(a.w[5]) + lane);
  RC_STAMP(12);
  __syncthreads();
  RC_STAMP(13);

  // ---- 6. y = SiLU(cv3(cat(a, b)))
  if (tecnt == NTE) step_out<NTE, NTE>(a, pm, rsm, te0, n, g, lane, b, breg, wa6);
  else if (NTE > 1 && tecnt == NTE - 1) step_out<(NTE > 1 ? NTE - 1 : 1), NTE>(a, pm, rsm, te0, n, g, lane, b, breg, wa6);
  RC_STAMP(14);
#undef RC_STAMP
}

// ================================================================================================== C2PSA's PSABlock, resident
// psa_resident_kernel: the PSABlock of a C2PSA layer (reference nn/modules/block.py:919-953 PSABlock, :863-916 Attention, with
// c = 128 channels, 2 heads, key_dim 32, head_dim 64: layer 10 of YOLO11n) on ONE image per workgroup, six launches -> one:
//
//     qkv = conv1x1(b)  (128 -> 256, no activation)                    per head [q(32) | k(32) | v(64)]
//     o   = v . softmax(q^T k * scale)^T  per head                     block.py:905-912
//     o   = o + dwconv3x3(v)                                           the positional encoding `pe` (block.py:913)
//     b   = b + conv1x1(o)           (`proj`, 128 -> 128)              PSABlock.forward: x = x + attn(x)
//     b   = b + conv1x1(SiLU(conv1x1(b)))  (`ffn`: 128 -> 256 -> 128)  x = x + ffn(x)
// Input: cat(a, b) = C2PSA.cv1's output; output: cat(a, b') in a second buffer (the input is never written, see psa_body).
//
// Eight waves; wave w owns an EIGHTH of the image's 16-pixel tiles from the first conv to the last store -- everything after
// the attention is pointwise, so the tensors between proj, ffn.0 and ffn.1 never leave that wave's registers (the accumulator
// layout of one conv IS the B-fragment layout of the next, conv_kernel's POST trick), and the residual stream b is its
// B fragments of the first conv.  Only the attention needs other pixels: per head the keys and the transposed values are
// written to LDS by the qkv conv's own epilogue in attention_kernel's layout (ops_misc.hip), and the attention loop is that
// kernel's, instruction for instruction: same key order, same online softmax, same roundings.  The depthwise `pe` reads v
// from dense LDS planes (bounds tests instead of guard pixels) with dwconv3_kernel's tap order.  The three tail convs'
// filters (160 KB) are staged into the whole LDS at once after the attention -- streamed from L2 by every wave they would be
// 2.5 MB of L1 traffic per workgroup.  Results are BIT-IDENTICAL to the six launches (tests/test_gpu_resident.py).
constexpr int PS_KP = 40;    // sK row pitch in elements (attention_kernel's AT_KP)

struct PsaLds {
  int sk, svt, vpl, omap, zslot, vp, hwp, nk32, total;
};
__host__ __device__ static inline PsaLds psa_lds_plan(int N) {
  PsaLds L;
  L.hwp = (N + 15) / 16 * 16;
  L.nk32 = (N + 31) & ~31;
  L.vp = L.nk32 + 8;
  L.sk = 0;
  L.svt = L.sk + L.nk32 * PS_KP * 2;
  L.vpl = L.svt + 64 * L.vp * 2;
  L.omap = 0;  // aliases sK / sVt once a head's attention is done
  L.zslot = L.vpl + 8 * L.hwp * 16;  // sixteen zero bytes: what an out-of-range pe tap reads
  const int front = L.zslot + 16;
  const int tail = 160 * 1024;  // the three tail filters: 32 + 64 + 64 KiB
  L.total = front > tail ? front : tail;
  return L;
}

// this wave's tiles t0 .. t0 + NT - 1 (tiles past `cnt` are inactive: clamped addresses, nothing stored)
template <int NT>
struct PsaTiles {
  int pix[NT];        // dense pixel index of the lane's pixel (clamped into the image)
  bool ok[NT];        // the lane's pixel exists
  unsigned gpix[NT];  // byte offset of the lane's pixel in a plane of the HBM layout (OOB_LANE: none)
  int ty[NT], tx[NT];
};
template <int NT>
__device__ __forceinline__ void psa_tiles(const PsaArgs& a, int b, int t0, int cnt, int n, PsaTiles<NT>& T) {
  const int N = a.H * a.W;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int i = (t0 + nt) * 16 + n;
    T.ok[nt] = nt < cnt && i < N;
    T.pix[nt] = T.ok[nt] ? i : N - 1;
    const int y = T.pix[nt] / a.W, x = T.pix[nt] - y * a.W;
    T.ty[nt] = y;
    T.tx[nt] = x;
    T.gpix[nt] = T.ok[nt] ? (unsigned)(a.g0 + b * a.img + y * a.W1 + x) * 16u : OOB_LANE;
  }
}

// MTW tiles (m0 .. m0 + MTW - 1 of one 64-channel block of a packed 128-input-channel filter) x NT pixel tiles, K = 128
template <int MTW, int NT>
__device__ __forceinline__ void psa_conv128(f32x4 (&acc)[MTW][NT], const bf16x8* __restrict__ wblk, int m0, const float* bias, int g,
                                            const bf16x8 (&bfr)[4][NT]) {
#pragma unroll
  for (int m = 0; m < MTW; ++m) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + chan_in_block(4, m0 + m, g, 0));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[m][nt] = bv;
  }
  bf16x8 wa[4][MTW];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int m = 0; m < MTW; ++m) wa[ks][m] = wblk[(ks * 4 + m0 + m) * 64];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int m = 0; m < MTW; ++m) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ks][m], bfr[ks][nt], acc[m][nt], 0, 0, 0);
}

// NTA: tiles per wave of the key / value conv, which every workgroup of an image runs over ALL its pixels (a wave owns an eighth
// of them); NTB: tiles per wave of this workgroup's OWN share of the image (queries, pe, proj, ffn), an eighth of that share.
template <int NTA, int NTB>
__device__ __forceinline__ void psa_body(const PsaArgs& a, unsigned char* lds, const PsaLds& L, int t0a, int cnta, int t0, int cnt, int b,
                                         int tid, int lane, int wave, int n, int g) {
  constexpr int NT = NTB;
  const int N = a.H * a.W;
  PsaTiles<NTA> TA;
  PsaTiles<NT> T;
  psa_tiles<NTA>(a, b, t0a, cnta, n, TA);
  psa_tiles<NT>(a, b, t0, cnt, n, T);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)a.b, (short)0, a.b_bytes, 0x00020000);
  // (the loads of b repeat -- both heads' convs and the residual stream: an opaque zero in the scalar offset keeps the compiler
  // from merging them into ONE set of registers that stays live through the whole kernel)
  auto load_b = [&](auto& dst, const auto& TT) __attribute__((always_inline)) {
    int zero = 0;
    asm volatile("" : "+s"(zero));
    constexpr int NX = sizeof(TT.pix) / sizeof(int);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int nt = 0; nt < NX; ++nt)
        dst[ks][nt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                     rb, TT.gpix[nt] == OOB_LANE ? OOB_LANE : (unsigned)(ks * 4 + g) * a.b_plane + TT.gpix[nt], zero, 0));
  };
  __bf16* const sK = reinterpret_cast<__bf16*>(lds + L.sk);
  __bf16* const sVt = reinterpret_cast<__bf16*>(lds + L.svt);
  unsigned char* const vpl = lds + L.vpl;
  unsigned char* const omap = lds + L.omap;
  const int vplane = L.hwp * 16;
  const float ninf = -__builtin_huge_valf();
  bf16x8 oh[2][2][NT];  // [head][k-step within the head][tile]: attention output + pe as proj's B fragments
#define PS_STAMP(id) YL_STAMP(a.stamps, RC_WAVES, wave, lane, id)
  PS_STAMP(0);
  // The block's input is never written: the workgroups of one image all read the whole residual stream b (keys and values) at
  // times of their own, so the result goes to a SECOND buffer -- and the untouched half `a` of C2PSA.cv1's output is copied
  // there for this workgroup's pixels, so that C2PSA.cv2 finds cat(a, b) in one place (block.py:1035-1038).
  {
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)a.a_in, (short)0, a.b_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rao = __builtin_amdgcn_make_buffer_rsrc((void*)a.a_out, (short)0, a.b_bytes, 0x00020000);
    bf16x8 av[4][NT];  // lane (n, g): planes g, 4 + g, 8 + g, 12 + g of its pixel
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        av[q][nt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                   ra, T.gpix[nt] == OOB_LANE ? OOB_LANE : (unsigned)(q * 4 + g) * a.b_plane + T.gpix[nt], 0, 0));
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, av[q][nt]), rao,
                                               T.gpix[nt] == OOB_LANE ? OOB_LANE : (unsigned)(q * 4 + g) * a.b_plane + T.gpix[nt], 0, 0);
  }

#pragma unroll
  for (int h = 0; h < 2; ++h) {  // (unrolled: oh[h] must stay in registers)
    // ---- the head's part of the qkv conv (channel blocks 2 h = q | k and 2 h + 1 = v of the packed filter, K = 128): keys and
    // values for ALL pixels (this wave: an eighth of them), queries for this workgroup's own pixels
    bf16x8 qf[NT];
    const bf16x8* const wqk = reinterpret_cast<const bf16x8*>(a.w_qkv) + (size_t)(2 * h) * 4 * 4 * 64 + lane;
    const bf16x8* const wv = wqk + 4 * 4 * 64;
    {
      bf16x8 bfr[4][NTA];
      load_b(bfr, TA);
      f32x4 acck[2][NTA];
      psa_conv128<2, NTA>(acck, wqk, 2, a.b_qkv + (2 * h) * 64, g, bfr);
      if (h == 1) __syncthreads();  // (B1) everybody is done with head 0's keys / values / output map
#pragma unroll
      for (int nt = 0; nt < NTA; ++nt) {
        bf16x8 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = (__bf16)acck[0][nt][r];
          o[4 + r] = (__bf16)acck[1][nt][r];
        }
        if (TA.ok[nt]) *reinterpret_cast<bf16x8*>(sK + TA.pix[nt] * PS_KP + g * 8) = o;  // k: row = key, channels 8 g ..
      }
      f32x4 accv[4][NTA];
      psa_conv128<4, NTA>(accv, wv, 0, a.b_qkv + (2 * h + 1) * 64, g, bfr);
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int nt = 0; nt < NTA; ++nt) {
          bf16x8 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o[r] = (__bf16)accv[2 * p][nt][r];
            o[4 + r] = (__bf16)accv[2 * p + 1][nt][r];
          }
          if (TA.ok[nt]) {
            *reinterpret_cast<bf16x8*>(vpl + (p * 4 + g) * vplane + TA.pix[nt] * 16) = o;  // v as planes (for pe)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) sVt[((p * 4 + g) * 8 + jj) * L.vp + TA.pix[nt]] = o[jj];  // and transposed (for P.V)
          }
        }
    }
    {
      bf16x8 bfr[4][NT];
      load_b(bfr, T);
      f32x4 accq[2][NT];
      psa_conv128<2, NT>(accq, wqk, 0, a.b_qkv + (2 * h) * 64, g, bfr);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        bf16x8 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = (__bf16)accq[0][nt][r];
          o[4 + r] = (__bf16)accq[1][nt][r];
        }
        qf[nt] = T.ok[nt] ? o : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};  // q: channels 8 g .. of the lane's query
      }
    }
    PS_STAMP(1 + 5 * h);
    __syncthreads();  // (B2) keys and values of head h complete
    PS_STAMP(2 + 5 * h);
    __builtin_amdgcn_sched_barrier(0);

    // ---- attention of head h for this wave's query tiles: attention_kernel's loop (ops_misc.hip), one stage (N <= 512)
    {
      float mrun[NT], lrun[NT];
      f32x4 acc[NT][4];
#pragma unroll
      for (int r = 0; r < NT; ++r) {
        mrun[r] = ninf;
        lrun[r] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[r][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const int nk = N;
#pragma unroll
      for (int r = 0; r < NT; ++r) {
        if (r >= cnt) continue;
        for (int k0 = 0; k0 < L.nk32; k0 += 32) {
          const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(sK + (k0 + n) * PS_KP + g * 8);
          const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(sK + (k0 + 16 + n) * PS_KP + g * 8);
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
          const float mn = fmaxf(mrun[r], lm);
          const float alpha = __expf(mrun[r] - mn);
          float ps = 0.f;
          bf16x8 pb;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float e = __expf(v[j] - mn);
            ps += e;
            pb[j] = (__bf16)e;
          }
          lrun[r] = lrun[r] * alpha + ps;
          mrun[r] = mn;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            const __bf16* vr = sVt + (dt * 16 + n) * L.vp + k0 + 4 * g;
            const bf16x4 lo = *reinterpret_cast<const bf16x4*>(vr);
            const bf16x4 hi = *reinterpret_cast<const bf16x4*>(vr + 16);
            const bf16x8 av = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) acc[r][dt][rr] *= alpha;
            acc[r][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, pb, acc[r][dt], 0, 0, 0);
          }
        }
      }
      PS_STAMP(3 + 5 * h);
      __syncthreads();  // (B3) nobody reads the keys / transposed values of head h any more: their place takes the output map
      PS_STAMP(4 + 5 * h);
#pragma unroll
      for (int r = 0; r < NT; ++r) {
        float lr = lrun[r];
        lr += __shfl_xor(lr, 16);
        lr += __shfl_xor(lr, 32);
        if (T.ok[r]) {
          const float inv = 1.0f / lr;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<bf16x4*>(omap + ((dt * 16 + 4 * g) >> 3) * vplane + T.pix[r] * 16 + ((4 * g) & 7) * 2) =
                bf16x4{(__bf16)(acc[r][dt][0] * inv), (__bf16)(acc[r][dt][1] * inv), (__bf16)(acc[r][dt][2] * inv),
                       (__bf16)(acc[r][dt][3] * inv)};
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- o += pe(v) on this wave's own pixels (dwconv3_kernel's arithmetic: bias first, taps in row-major order, then the
    // stored attention output), straight into proj's B fragments.  The map entries read here were written by this wave.
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int pl = ks * 4 + g;              // plane of the head's 64 channels
      const int c = h * 64 + pl * 8;          // first of the lane's 8 channels
      float acc8[NT][8];
      {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.b_pe + c), b1 = *reinterpret_cast<const f32x4*>(a.b_pe + c + 4);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc8[nt][j] = b0[j];
            acc8[nt][4 + j] = b1[j];
          }
      }
#pragma unroll 1
      for (int t = 0; t < 9; ++t) {  // (a real loop: the body is small, unrolled it only adds register pressure)
        const float* wq = a.w_pe + t * 128 + c;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wq), w1 = *reinterpret_cast<const f32x4*>(wq + 4);
        const int dy = t / 3 - 1, dx = t - 3 * (t / 3) - 1;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const bool in = (unsigned)(T.ty[nt] + dy) < (unsigned)a.H && (unsigned)(T.tx[nt] + dx) < (unsigned)a.W;
          // an out-of-range tap reads a zero (what dwconv3_kernel reads from the guard pixels) and goes through the same fma
          const bf16x8 vv = *reinterpret_cast<const bf16x8*>(in ? vpl + pl * vplane + (T.pix[nt] + dy * a.W + dx) * 16 : lds + L.zslot);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc8[nt][j] = __builtin_fmaf((float)vv[j], w0[j], acc8[nt][j]);
            acc8[nt][4 + j] = __builtin_fmaf((float)vv[4 + j], w1[j], acc8[nt][4 + j]);
          }
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const bf16x8 prev = *reinterpret_cast<const bf16x8*>(omap + pl * vplane + T.pix[nt] * 16);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)(acc8[nt][j] + (float)prev[j]);
        oh[h][ks][nt] = o;
      }
    }
    PS_STAMP(5 + 5 * h);
  }

  // ---- the three tail filters -> LDS (all of it), each wave its share of the 160 1-KiB pieces
  __builtin_amdgcn_sched_barrier(0);  // (phase fences for the instruction scheduler: without them it hoists the later phases'
                                      // bias and filter loads across the unrolled head code and spills > 1000 registers)
  __syncthreads();
  {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_tail, (short)0, 160u * 1024u, 0x00020000);
#pragma unroll 4
    for (int pc = wave; pc < 160; pc += RC_WAVES)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_r*)(lds + pc * 1024), 16, (unsigned)(pc * 1024 + lane * 16), 0, 0, 0);
  }
  bf16x8 bp[4][NT];  // the residual stream b of this wave's pixels: first the block's input, then b + proj(o)
  load_b(bp, T);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PS_STAMP(11);
  __syncthreads();
  PS_STAMP(12);
  const bf16x8* const wproj = reinterpret_cast<const bf16x8*>(lds) + lane;              // [cb 2][ks 4][4][64]
  const bf16x8* const wf0 = reinterpret_cast<const bf16x8*>(lds + 32 * 1024) + lane;   // [cb 4][ks 4][4][64]
  const bf16x8* const wf1 = reinterpret_cast<const bf16x8*>(lds + 96 * 1024) + lane;   // [cb 2][ks 8][4][64]
  // ---- b = b + proj(o)
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    f32x4 acc[4][NT];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b_proj + cb * 64 + chan_in_block(4, m, g, 0));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[m][nt] = bv;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 wa[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) wa[m] = wproj[((cb * 4 + ks) * 4 + m) * 64];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[m], oh[ks >> 1][ks & 1][nt], acc[m][nt], 0, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const bf16x8 rv = bp[cb * 2 + p][nt];
        bf16x8 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = (__bf16)(acc[2 * p][nt][r] + (float)rv[r]);
          o[4 + r] = (__bf16)(acc[2 * p + 1][nt][r] + (float)rv[4 + r]);
        }
        bp[cb * 2 + p][nt] = o;
      }
  }
  PS_STAMP(13);
  // ---- b = b + ffn.1(SiLU(ffn.0(b))), two pixel tiles at a time: ffn.0's 64-channel blocks feed ffn.1's k-steps in order
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)a.b_out, (short)0, a.b_bytes, 0x00020000);
#pragma unroll
  for (int n0 = 0; n0 < NT; n0 += 2) {  // (unrolled: bp[..][n0 + q] must stay in registers)
    __builtin_amdgcn_sched_barrier(0);
    constexpr int NP2 = 2;
    f32x4 acc1[8][NP2];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b_ffn1 + (m >> 2) * 64 + chan_in_block(4, m & 3, g, 0));
#pragma unroll
      for (int q = 0; q < NP2; ++q) acc1[m][q] = bv;
    }
#pragma unroll 1
    for (int mb = 0; mb < 4; ++mb) {  // (a real loop: fully unrolled, the scheduler hoists the filter reads of all blocks and spills)
      f32x4 acc0[4][NP2];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b_ffn0 + mb * 64 + chan_in_block(4, m, g, 0));
#pragma unroll
        for (int q = 0; q < NP2; ++q) acc0[m][q] = bv;
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 wa[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) wa[m] = wf0[((mb * 4 + ks) * 4 + m) * 64];
#pragma unroll
        for (int q = 0; q < NP2; ++q)
#pragma unroll
          for (int m = 0; m < 4; ++m)
            acc0[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[m], bp[ks][n0 + q < NT ? n0 + q : NT - 1], acc0[m][q], 0, 0, 0);
      }
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        bf16x8 f[NP2];
#pragma unroll
        for (int q = 0; q < NP2; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            f[q][r] = (__bf16)silu_f(acc0[2 * p][q][r]);
            f[q][4 + r] = (__bf16)silu_f(acc0[2 * p + 1][q][r]);
          }
        const int ks1 = 2 * mb + p;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          const bf16x8 w1 = wf1[(((m >> 2) * 8 + ks1) * 4 + (m & 3)) * 64];
#pragma unroll
          for (int q = 0; q < NP2; ++q) acc1[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, f[q], acc1[m][q], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2)  // tile pair m2 = 64-channel block m2 / 2, pair m2 % 2 -> plane m2 * 4 + g
#pragma unroll
      for (int q = 0; q < NP2; ++q) {
        if (n0 + q >= NT) continue;
        const bf16x8 rv = bp[m2][n0 + q];
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc1[2 * m2][q][r] + (float)rv[r];
          v[4 + r] = acc1[2 * m2 + 1][q][r] + (float)rv[4 + r];
        }
        const unsigned gp = T.gpix[n0 + q];
        __builtin_amdgcn_raw_buffer_store_b128(
            u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])}, rout,
            gp == OOB_LANE ? OOB_LANE : (unsigned)(m2 * 4 + g) * a.b_plane + gp, 0, 0);
      }
  }
  PS_STAMP(14);
#undef PS_STAMP
}

template <int NTA, int NTB>
__global__ __launch_bounds__(RC_THREADS) void psa_resident_kernel(PsaArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  const int N = a.H * a.W;
  const PsaLds L = psa_lds_plan(N);
  // the transposed values' padding columns (keys N .. nk32 - 1) and the zero slot must read as zero: clear the front once
  {
    const bf16x8 z = {};
    for (int i = tid * 16; i < L.zslot + 16; i += RC_THREADS * 16) *reinterpret_cast<bf16x8*>(psm + i) = z;
  }
  __syncthreads();
  // a.split workgroups per image: each runs the key / value conv over the whole image and everything else on its share of the tiles
  const int b = blockIdx.x / a.split, part = blockIdx.x - b * a.split;
  const int ntiles = (N + 15) / 16;
  int t0a, cnta, p0, pc, t0, cnt;
  even_split(ntiles, RC_WAVES, wave, t0a, cnta);
  even_split(ntiles, a.split, part, p0, pc);
  even_split(pc, RC_WAVES, wave, t0, cnt);
  psa_body<NTA, NTB>(a, psm, L, t0a, cnta, p0 + t0, cnt, b, tid, lane, wave, n, g);
}

bool psa_resident_fits(int c, int heads, int key_dim, int head_dim, int H, int W) {
  const int N = H * W;
  return c == 128 && heads == 2 && key_dim == 32 && head_dim == 64 && N >= 8 && N <= 400 && psa_lds_plan(N).total <= 160 * 1024;
}

template <int NTA, int NTB>
static int psa_launch_t(const PsaArgs& a, size_t lds, hipStream_t s) {
  auto kern = psa_resident_kernel<NTA, NTB>;
  if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * a.split)), dim3(RC_THREADS), lds, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

int psa_resident_launch(const PsaArgs& a0, hipStream_t s) {
  PsaArgs a = a0;
  a.stamps = g_stamp_buf;
  if (!psa_resident_fits(128, 2, 32, 64, a.H, a.W)) return fail(-3, "psa_resident: the map does not fit one workgroup");
  const PsaLds L = psa_lds_plan(a.H * a.W);
  const int ntiles = (a.H * a.W + 15) / 16;
  // Workgroups per image.  One workgroup per image is the cheapest in chip time (the key / value conv runs once), but a batch
  // of 32 then occupies 32 of the 256 CUs for ~90 us; two halve that time for ~10 % more work (YL_PSA_SPLIT overrides).
  const int split_env = tune_int("YL_PSA_SPLIT", 0);  // (read per launch: the tests switch it)
  int split = split_env > 0 ? split_env : a0.split > 0 ? a0.split : (ntiles > 16 ? 2 : 1);  // (a0.split: asked for by the program, yl_op.reserved3)
  if (split > ntiles) split = ntiles;
  if (split < 1 || split > 4) split = 1;
  while (split < 4 && ((ntiles + split - 1) / split + 7) / 8 > 2) ++split;  // the own share: at most two tiles per wave
  a.split = split;
  const int nta = (ntiles + 7) / 8, ntb = ((ntiles + split - 1) / split + 7) / 8;
  switch (nta * 10 + ntb) {
    case 11: return psa_launch_t<1, 1>(a, (size_t)L.total, s);
    case 21: return psa_launch_t<2, 1>(a, (size_t)L.total, s);
    case 22: return psa_launch_t<2, 2>(a, (size_t)L.total, s);
    case 31: return psa_launch_t<3, 1>(a, (size_t)L.total, s);
    case 32: return psa_launch_t<3, 2>(a, (size_t)L.total, s);
    case 41: return psa_launch_t<4, 1>(a, (size_t)L.total, s);
    case 42: return psa_launch_t<4, 2>(a, (size_t)L.total, s);
  }
  return fail(-3, "psa_resident: unsupported map size");
}

size_t c3k_resident_lds(int H, int W, int* xpl_out, int* apl_out, int* r1_out, int* ntiles_out) {
  const int ntiles = (H * W + 15) / 16;
  const int xpl = ntiles * 16 * 16;
  const int apx = (((H + 2) * (W + 1) + 2) + 15) & ~15;
  const int apl = apx * 16;
  const int r0 = 16 * xpl > 8 * apl ? 16 * xpl : 8 * apl;
  if (xpl_out) *xpl_out = xpl;
  if (apl_out) *apl_out = apl;
  if (r1_out) *r1_out = r0;
  if (ntiles_out) *ntiles_out = ntiles;
  return (size_t)r0 + (size_t)8 * apl + 256;  // + the scratch line of the filter warm-up (one dword per lane)
}

// Whether the resident kernel takes a C3k block of inner width c_ on an H x W map
bool c3k_resident_fits(int c_, int H, int W) {
  if (c_ != 64) return false;
  int nt = 0;
  const size_t lds = c3k_resident_lds(H, W, nullptr, nullptr, nullptr, &nt);
  return lds <= 160 * 1024 && nt >= 1 && nt <= 28 && nt * 16 <= RC_THREADS;
}

template <int NTQ>
static int c3k_launch_t(const C3kArgs& a, size_t lds, hipStream_t s) {
  auto kern = c3k_resident_kernel<NTQ>;
  if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), 160 * 1024)) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * a.split)), dim3(RC_THREADS), lds, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

int c3k_resident_launch(C3kArgs a, hipStream_t s) {
  if (!c3k_resident_fits(64, a.H, a.W)) return fail(-3, "c3k_resident: the map does not fit one workgroup's LDS");
  // Workgroups per image (YL_C3K_SPLIT, read per launch: the tests switch it).  Two: each takes half of the rows + a halo of four
  // = 0.7 of the map on a 20x20 image: 0.83 of the latency (47.4 -> 39.4 us) for 1.4 x the chip time.  The program asks for it
  // (yl_op.reserved3 = 2: nn/lower.py one_batch=True, the predictor's one-batch-at-a-time path); with 20 batches in flight it costs
  // 1.5 % of the throughput.
  const int want = env_int("YL_C3K_SPLIT", a.split);
  a.split = (want == 2 && (a.H + 1) / 2 + 4 < a.H) ? 2 : 1;
  a.Hs = a.split == 2 ? (a.H + 1) / 2 + 4 : a.H;
  const size_t lds = c3k_resident_lds(a.Hs, a.W, &a.xpl, &a.apl, &a.r1, &a.ntiles);
  a.stamps = g_stamp_buf;
  a.scratch = a.r1 + 8 * a.apl;
  static const int dbg = tune_int("YL_RES_DBG", 0);  // development: 1 = no filter warm-up
  a.dbg = dbg;
  switch ((a.ntiles + 3) / 4) {
    case 1: return c3k_launch_t<1>(a, lds, s);
    case 2: return c3k_launch_t<2>(a, lds, s);
    case 3: return c3k_launch_t<3>(a, lds, s);
    case 4: return c3k_launch_t<4>(a, lds, s);
    case 5: return c3k_launch_t<5>(a, lds, s);
    case 6: return c3k_launch_t<6>(a, lds, s);
    case 7: return c3k_launch_t<7>(a, lds, s);
  }
  return fail(-3, "c3k_resident: unsupported map size");
}

}  // namespace yl
