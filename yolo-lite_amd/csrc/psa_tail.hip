// psa_tail_kernel: the POINTWISE tail of a C2PSA layer with one PSABlock -- everything behind the attention -- in ONE launch:
//
//     b   = b + conv1x1(o)                   `proj`,  128 -> 128   PSABlock.forward: x = x + attn(x)   (block.py:919-953)
//     b   = b + conv1x1(SiLU(conv1x1(b)))    `ffn`,   128 -> 256 -> 128                x = x + ffn(x)
//     out = SiLU(conv1x1(cat(a, b)))         C2PSA.cv2, 256 -> 256                     (block.py:1035-1038)
//
// o = the attention's output + positional encoding (block.py:905-913), cat(a, b) = C2PSA.cv1's output; channels as in
// YOLO11n (layer 10).  Four launches of the unfused program (proj, ffn.0, ffn.1, cv2: 42 us in sequence at batch 32 on 20x20
// maps for 0.9 us of matrix work -- four launch skeletons) become one.  All four convs are pointwise, so NOTHING ties a
// workgroup to an image: the pixels of the whole batch are one list, a wave owns two 16-pixel tiles of it from the first load
// to the last store, and the tensors between the convs never leave its registers (the accumulator layout of one conv IS the
// B-fragment layout of the next: conv_kernel's POST trick, as in psa_resident_kernel, whose proj / ffn code this is).  A
// workgroup = four waves = 128 pixels (64 while that leaves CUs idle: one tile per wave); the filters come through LDS: proj |
// ffn.0 | ffn.1 (160 KB, the whole LDS) by LDS-DMA at the start, cv2's 128 KB over them once every wave is through ffn.1.  b after
// ffn.1 is only cv2's input and is not stored.  Measured (batch 32, 20x20 maps, in-kernel stamps): 18.2 us kernel span with one
// tile per wave on 200 workgroups (22.5 us with two on 100) -- filters staged 3.4, proj 1.1, ffn 5.8 (64 SiLUs per lane: VALU),
// cv2's filter 2.9, cv2 + stores 4.5 -- 23.5 us in sequence against 41.8 us for the four launches.
// Arithmetic is conv_kernel's (packed filters in 64-channel blocks, k order, fp32 chain from the bias, one rounding to bf16 per
// layer, silu_f / silu4): BIT-IDENTICAL to the four launches (tests/test_gpu_resident.py).  The program's latency profile
// (nn/lower.py lower(one_batch=True)) uses it; with batches in flight the one-launch PSABlock (psa_resident_kernel) is cheaper.
#include "conv_common.h"

namespace yl {

typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int PT_THREADS = 256;
constexpr int PT_WAVES = PT_THREADS / 64;
constexpr int PT_LDS = 160 * 1024;

// NT: 16-pixel tiles per wave (2: a filter fragment read from LDS feeds two MFMAs; 1: twice the workgroups)
template <int NT>
__global__ __launch_bounds__(PT_THREADS) void psa_tail_kernel(PsaTailArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ptm[];
  unsigned char* const lds = ptm;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
#define PT_STAMP(id) YL_STAMP(a.stamps, PT_WAVES, wave, lane, id)
  PT_STAMP(0);
  // ---- proj | ffn.0 | ffn.1 -> LDS, each wave its share of the 160 1-KiB pieces
  {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_tail, (short)0, 288u * 1024u, 0x00020000);
#pragma unroll 4
    for (int pc = wave; pc < 160; pc += PT_WAVES)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(lds + pc * 1024), 16, (unsigned)(pc * 1024 + lane * 16), 0, 0, 0);
  }
  // ---- this lane's pixels: dense index over the batch -> byte offset inside a plane of the guarded layout
  const int HW = a.H * a.W, P = a.B * HW;
  unsigned gpix[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int i = ((blockIdx.x * PT_WAVES + wave) * NT + nt) * 16 + n;
    const int ic = i < P ? i : P - 1;
    const int b = ic / HW, r = ic - b * HW;
    const int y = r / a.W, x = r - y * a.W;
    gpix[nt] = i < P ? (unsigned)(a.g0 + b * a.img + y * a.W1 + x) * 16u : OOB_LANE;
  }
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)a.o, (short)0, a.o_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)a.u, (short)0, a.u_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, (short)0, a.y_bytes, 0x00020000);
  auto load4 = [&](bf16x8 (&dst)[4][NT], __amdgpu_buffer_rsrc_t rs, unsigned plane, int p0) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        dst[ks][nt] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                                                     rs, gpix[nt] == OOB_LANE ? OOB_LANE : (unsigned)(p0 + ks * 4 + g) * plane + gpix[nt], 0, 0));
  };
  bf16x8 oh[4][NT];  // proj's B fragments: attention output + pe
  bf16x8 bp[4][NT];  // the residual stream b: first C2PSA.cv1's second half, then b + proj(o)
  load4(oh, ro, a.o_plane, 0);
  load4(bp, ru, a.u_plane, 16);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PT_STAMP(1);
  __syncthreads();
  PT_STAMP(2);
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
        for (int m = 0; m < 4; ++m) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[m], oh[ks][nt], acc[m][nt], 0, 0, 0);
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
  PT_STAMP(3);
  // ---- b = b + ffn.1(SiLU(ffn.0(b))): ffn.0's 64-channel blocks feed ffn.1's k-steps in order
  bf16x8 b2[4][NT];  // b after the block = cv2's k-steps 4 .. 7
  {
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc1[8][NT];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b_ffn1 + (m >> 2) * 64 + chan_in_block(4, m & 3, g, 0));
#pragma unroll
      for (int q = 0; q < NT; ++q) acc1[m][q] = bv;
    }
#pragma unroll 1
    for (int mb = 0; mb < 4; ++mb) {  // (a real loop: fully unrolled, the scheduler hoists the filter reads of all blocks and spills)
      f32x4 acc0[4][NT];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b_ffn0 + mb * 64 + chan_in_block(4, m, g, 0));
#pragma unroll
        for (int q = 0; q < NT; ++q) acc0[m][q] = bv;
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 wa[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) wa[m] = wf0[((mb * 4 + ks) * 4 + m) * 64];
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
          for (int m = 0; m < 4; ++m) acc0[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[m], bp[ks][q], acc0[m][q], 0, 0, 0);
      }
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        bf16x8 f[NT];
#pragma unroll
        for (int q = 0; q < NT; ++q)
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
          for (int q = 0; q < NT; ++q) acc1[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, f[q], acc1[m][q], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2)  // tile pair m2 = 64-channel block m2 / 2, pair m2 % 2 -> plane m2 * 4 + g of b
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        const bf16x8 rv = bp[m2][q];
        bf16x8 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = (__bf16)(acc1[2 * m2][q][r] + (float)rv[r]);
          o[4 + r] = (__bf16)(acc1[2 * m2 + 1][q][r] + (float)rv[4 + r]);
        }
        b2[m2][q] = o;
      }
  }
  PT_STAMP(4);
  // ---- cv2's filter over the three: every wave is through ffn.1 first.  Its other input, a, is requested in front of the barrier.
  bf16x8 af[4][NT];
  load4(af, ru, a.u_plane, 0);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w_tail, (short)0, 288u * 1024u, 0x00020000);
#pragma unroll 4
    for (int pc = wave; pc < 128; pc += PT_WAVES)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(lds + pc * 1024), 16, (unsigned)((160 + pc) * 1024 + lane * 16), 0, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PT_STAMP(5);
  __syncthreads();
  PT_STAMP(6);
  // ---- out = SiLU(cv2(cat(a, b))), two 64-channel blocks at a time; k-steps 0 .. 3 = a, 4 .. 7 = b
  const bf16x8* const wc = reinterpret_cast<const bf16x8*>(lds) + lane;  // [cb 4][ks 8][4][64]
#pragma unroll 1
  for (int ch = 0; ch < 2; ++ch) {
    f32x4 acc[8][NT];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b_cv2 + (ch * 2 + (m >> 2)) * 64 + chan_in_block(4, m & 3, g, 0));
#pragma unroll
      for (int q = 0; q < NT; ++q) acc[m][q] = bv;
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      bf16x8 wa[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) wa[m] = wc[(((ch * 2 + (m >> 2)) * 8 + ks) * 4 + (m & 3)) * 64];
#pragma unroll
      for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int m = 0; m < 8; ++m)
          acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[m], ks < 4 ? af[ks & 3][q] : b2[ks & 3][q], acc[m][q], 0, 0, 0);
    }
    if (a.act) {
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[m][q] = silu4(acc[m][q]);
    }
#pragma unroll
    for (int m2 = 0; m2 < 4; ++m2)  // tile pair m2 of this half -> plane (ch * 2 + m2 / 2) * 8 + (m2 % 2) * 4 + g
#pragma unroll
      for (int q = 0; q < NT; ++q) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[2 * m2][q][r];
          v[4 + r] = acc[2 * m2 + 1][q][r];
        }
        const unsigned plane = (unsigned)((ch * 2 + (m2 >> 1)) * 8 + (m2 & 1) * 4 + g);
        __builtin_amdgcn_raw_buffer_store_b128(
            u32x4{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])}, ry,
            gpix[q] == OOB_LANE ? OOB_LANE : plane * a.y_plane + gpix[q], 0, 0);
      }
  }
  PT_STAMP(7);
#undef PT_STAMP
}

// Whether the tail kernel takes a C2PSA layer: n PSABlocks of width c, output channels c2
bool psa_tail_fits(int c, int c2, int n_blocks) { return c == 128 && c2 == 256 && n_blocks == 1; }

int psa_tail_launch(const PsaTailArgs& a0, hipStream_t s) {
  PsaTailArgs a = a0;
  a.stamps = g_stamp_buf;
  const long long P = (long long)a.B * a.H * a.W;
  if (P <= 0 || P > (1LL << 28)) return fail(-3, "psa_tail: bad map size");
  // one tile per wave while that still leaves CUs idle (a workgroup holds the whole LDS of its CU)
  static const int nt_env = tune_int("YL_PSA_TAIL_NT", 0);
  const int nt = nt_env == 1 || nt_env == 2 ? nt_env : (P <= (long long)device_cu_count() * PT_WAVES * 16 ? 1 : 2);
  const int per_wg = PT_WAVES * nt * 16;
  auto kern = nt == 1 ? psa_tail_kernel<1> : psa_tail_kernel<2>;
  if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), PT_LDS)) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)((P + per_wg - 1) / per_wg)), dim3(PT_THREADS), PT_LDS, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

}  // namespace yl
