// Fused Conv2d + (folded BN) bias + SiLU [+ residual] as an MFMA implicit GEMM for gfx950.
//
// Replaces reference nn/modules/conv.py:35-53 (Conv.forward = act(bn(conv(x)))) with the BN folded
// per utils/torch_utils.py:182-209, and the Bottleneck shortcut add of nn/modules/block.py:341-343.
// Reading a plane range of a wider buffer and writing into a plane range implement torch.chunk /
// torch.cat of C2f/C3/SPPF/C2PSA/Concat (block.py:231-235 etc.) without copies.
//
// GEMM mapping (v_mfma_f32_16x16x32_bf16, wave64):
//   A (16 x 32)  = packed weights, row = output channel, k = (tap, cin) in groups of 8   (from LDS)
//   B (32 x 16)  = activations,    col = output pixel                                    (from L1/L2)
//   D (16 x 16)  : lane (pixel n, quad g) holds rows g*4..g*4+3 of every 16-channel tile.
// One workgroup = 4 waves; a wave owns NT*16 CONSECUTIVE output pixels (row-major over the whole
// batch) times MT*16 output channels.
//
// Activations live in the C8-planar guarded layout (yl_internal.h): a B fragment is 4 runs of 256
// contiguous bytes (one per k-group plane), every 3x3 tap is a constant byte offset, and the zero
// padding comes from the guard pixels -- the k-loop issues one add and one buffer load per fragment,
// no predicates, no halo staging, no barrier.  Only the block's filter slab lives in LDS (staged
// once, fragment order, conflict-free reads); filters larger than the slab budget are restaged in
// chunks of k-steps.
// (Measured alternatives that lost -- NHWC fragments at one TA clock per lane, an LDS halo-tile
//  kernel, a persistent weight-stationary kernel -- are recorded in DESIGN.md.)
#include "conv_common.h"

#ifndef YL_RING_NT4
#define YL_RING_NT4 2  // k-steps of activation fragments in flight per wave at 64 pixels per wave (16-byte loads per lane: 4 x this).
                       // 4 (209-254 registers, still two waves per SIMD): layer 3 58.8 -> 54.8 us, but 48.3-48.9 k against
                       // 49.2-49.7 k images/s with 20 batches in flight
#endif
#ifndef YL_RING_NT2
#define YL_RING_NT2 4  // ... at 32 / 16 pixels per wave (round 5: 8 / 16 measured, no launch changes by more than its noise)
#endif
#ifndef YL_RING_NT1
#define YL_RING_NT1 8
#endif
#ifndef YL_CONV_WPE
#define YL_CONV_WPE 2  // conv_kernel: at least this many waves per SIMD (caps its registers at 512 / YL_CONV_WPE).  3 (the
                       // 64-pixel tiles then spill from 177-203 to 168 registers): 47.0-47.1 k images/s against 48.2-48.9 k
#endif

#ifndef YL_DWPW_PF2
#define YL_DWPW_PF2 1  // the two-register-set prefetch of the depthwise phase for 32 pixels per wave too, in the launches with a fused tail (50 KB of LDS: three workgroups per CU with or without the second set; 68.0 -> 66.4 us for the three levels)
#endif
#ifndef YL_DWPW_WPE
#define YL_DWPW_WPE 2
#endif
namespace yl {

// K-steps [kbeg, kbeg + nk) of one patch against the slab currently in LDS (slab row = ks - kbeg).
// Activation fragments ride a ring of D k-steps (D*NT = 8 sixteen-byte loads in flight per lane,
// refilled right after the MFMAs that consumed a slot); filter fragments come from LDS one step ahead.
// Branch-free main loop over whole groups of D k-steps: ring refills past the end resolve to
// out-of-range offsets (zeros, no memory access) and the filter prefetch is clamped, so the body is
// straight-line code and the compiler keeps counted vmcnt waits (the ring really stays in flight).
// kbeg must be a multiple of D.
template <int KS, int MT, int NT, bool UNIFORM, int D, bool UP = false>
__device__ __forceinline__ void run_ksteps(const ConvArgs& a, __amdgpu_buffer_rsrc_t rin, int g, const Patch<NT>& P,
                                           const bf16x8* wl, int kbeg, int nk, bf16x8 (&xr)[D][NT],
                                           f32x4 (&acc)[MT][NT],
                                           __amdgpu_buffer_rsrc_t rin2 = __amdgpu_buffer_rsrc_t()) {
  bf16x8 wv[2][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) wv[0][mt] = wl[mt * 64];
  const int last = nk - 1;
  int i0 = 0;
  for (; i0 + D <= nk; i0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int i = i0 + d;
      const int in_ = min(i + 1, last);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) wv[(d + 1) & 1][mt] = wl[(in_ * MT + mt) * 64];
      mma_step<MT, NT>(wv[d & 1], xr[d], acc);
      fetch_b<KS, NT, UNIFORM, UP>(a, rin, kbeg + i + D, g, P, xr[d], rin2);
    }
  }
#pragma unroll
  for (int d = 0; d < D - 1; ++d) {
    const int i = i0 + d;
    if (i < nk) {
      const int in_ = min(i + 1, last);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) wv[(d + 1) & 1][mt] = wl[(in_ * MT + mt) * 64];
      mma_step<MT, NT>(wv[d & 1], xr[d], acc);
    }
  }
}

// The NMS candidate pre-filter (nms.hip decode_candidates_sparse_kernel) in the epilogue of the launch that produces a Detect
// level's class logits: an anchor whose best STORED logit (bf16, as the level map holds it) could clear the confidence
// threshold is appended to its image's sub-list -- the class planes (43 MB per batch of 32 for YOLO11n) are then never re-read
// for it.  acc: the final logits of `patch` (all `cout` class channels in this wave: ncb == 1); lane (n, g) holds 4 rows of
// each of the MT channel tiles of pixel n.
template <int MT, int NT>
__device__ __forceinline__ void pf_sink(const ConvArgs& a, const f32x4 (&acc)[MT][NT], int patch, int g, int n, int cout) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    float m = -__builtin_huge_valf();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (chan_in_block(MT, mt, g, r) < cout) m = fmaxf(m, (float)(__bf16)acc[mt][nt][r]);
      }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    const int gp = patch * (16 * NT) + nt * 16 + n;  // pixel index over the batch (patch_setup's order)
    if (g == 0 && gp < a.npix && m > a.pf_logit_lo) {
      const int hw = a.Ho * a.Wo;
      const int b = gp / hw;
      const int an = a.pf_anchor0 + (gp - b * hw);
      const int sub = b * PF_NSUB + ((an >> 6) & (PF_NSUB - 1));
      const int slot = atomicAdd(&a.pf_count[sub], 1);
      if (slot < a.pf_sub_cap) a.pf_list[(long long)sub * a.pf_sub_cap + slot] = an;
    }
  }
}

// POST (MT even, all MT*16 channels of the first conv in this wave, i.e. ncb == 1): a 1x1 conv that
// consumes ONLY this conv's output is applied in the epilogue.  After bias + SiLU + bf16 rounding the
// accumulators of tile pair p ARE the B fragment of k-step p of that 1x1 conv (same channel permutation
// that makes the stores 16 bytes wide), so the intermediate tensor is never written or read.
// O8: the epilogue also stores the e4m3 copy of the output (yl_op.out8, fp8 mode only; a separate instantiation so that the bf16
// programs run exactly the code they ran before the copy existed).
// PF: this conv writes a Detect level's class logits (act == 0, ncb == 1) and feeds the NMS candidate pre-filter on its way out
// (pf_sink; the networks whose class branch ends in a plain 1x1 launch: YOLO11s / m / l / x, where the branch is too wide for
// dwpw_kernel); a separate instantiation for the same reason.
// DFL: this conv (its fused follow-up 1x1 with POST) computes a Detect level's 64 DFL logits: the box decode runs on the fp32
// accumulators (dfl_epilogue, conv_common.h) and the bf16 logits are stored only with a.dfl_keep; separate instantiations again.
template <int KS, int MT, int NT, bool UNIFORM, bool POST, bool UP = false, bool O8 = false, bool PF = false, bool DFL = false>
__device__ __forceinline__ void conv_body(const ConvArgs& a, int block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  const int bid = xcd_remap(block, a.nblocks);
  const int cb = bid % a.ncb;
  const int pblk = bid / a.ncb;
  const __amdgpu_buffer_rsrc_t rin =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.in, (short)0, (a.dbg & 1) ? 0 : a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout =
      __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, (a.dbg & 2) ? 0 : a.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.res, (short)0, a.res ? a.res_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rin2 =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.in2, (short)0, UP ? a.in2_bytes : 0, 0x00020000);
  constexpr int D = NT >= 4 ? YL_RING_NT4 : (NT == 2 ? YL_RING_NT2 : YL_RING_NT1);
  bf16x8 xr[D][NT];
  Patch<NT> P;
  const unsigned lane_off = UNIFORM ? (unsigned)g * a.in_plane : 0u;
  const unsigned lane_off2 = UP ? (unsigned)g * a.in2_plane : 0u;
  // wave-uniform values are forced into SGPRs so the k-loop bookkeeping stays on the scalar unit
  int patch = __builtin_amdgcn_readfirstlane(pblk * a.ppw * 4 + wave);
  bool live = patch < a.npatches;
  // The first ring fill of a patch is issued BEFORE the filter slab is staged (first patch) or before
  // the previous patch's epilogue stores, so its latency is always covered.
  if (live) {
    patch_setup<NT, UP>(a, patch, n, lane_off, P, lane_off2);
#pragma unroll
    for (int d = 0; d < D; ++d) fetch_b<KS, NT, UNIFORM, UP>(a, rin, d, g, P, xr[d], rin2);
  }
  // per-lane bias of its 4 rows of every channel tile (the accumulators start from it)
  f32x4 bias_r[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
    bias_r[mt] = *reinterpret_cast<const f32x4*>(a.bias + cb * MT * 16 + chan_in_block(MT, mt, g, 0));
  const bf16x8* wsrc = reinterpret_cast<const bf16x8*>(a.wpk) + (size_t)cb * a.ksteps * MT * 64;
  const bf16x8* wl = reinterpret_cast<const bf16x8*>(smem) + lane;

  // One loop serves both regimes (a single accumulator set keeps the register count at two waves
  // per SIMD).  Usual case: the whole filter fits the slab -- staged once, then ppw patches per wave.
  // Big filters (multi): ppw == 1 and the slab is restaged every kchunk k-steps (kchunk % 8 == 0);
  // waves without a patch still take part in the copies and barriers.
  const bool multi = a.kchunk < a.ksteps;
  bool first = true;
  for (int pp = 0; pp < a.ppw; ++pp) {
    if (!first && !live) break;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = bias_r[mt];
    // A filter larger than the slab budget is restaged in chunks of kchunk k-steps.  a.slab2 (round 5): ASYNCHRONOUSLY -- two chunk
    // buffers, the next chunk travels to LDS by LDS-DMA (no registers, nobody waits) while this one is computed, ONE raw barrier
    // per chunk.  (Copied through registers behind a barrier of its own, with a second barrier behind it, every restage stood
    // still for a memory round trip.)  The DMA pieces of chunk c + 1 are this wave's OLDEST vector-memory operations when chunk
    // c ends: the D x NT ring loads of the next k-steps are younger, so a counted wait that leaves exactly those in flight
    // covers the pieces; a wave without a patch drains everything.  One k-loop call site serves both forms (two inlined copies
    // of it cost every instantiation ~36 registers).
    typedef __attribute__((address_space(3))) void lds_void_s;
    const bool async = multi && a.slab2;
    const int chunk_bytes = a.kchunk * MT * 1024;
    auto dma_chunk = [&](int kc, int buf) __attribute__((always_inline)) {
      const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
          (void*)wsrc, (short)0, (a.dbg & 4) ? 0u : (unsigned)a.ksteps * MT * 1024u, 0x00020000);
      const int np = min(a.kchunk, a.ksteps - kc) * MT;  // 1 KiB pieces
      for (int pc = wave; pc < np; pc += 4)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_s*)(smem + buf * chunk_bytes + pc * 1024), 16,
                                                 (kc * MT + pc) * 1024 + lane * 16, 0, 0, 0);
    };
    int c = 0;
    for (int kc = 0; kc < a.ksteps; kc += a.kchunk, ++c) {
      const int nk = min(a.kchunk, a.ksteps - kc);
      const bool more = kc + a.kchunk < a.ksteps;
      if (async) {
        if (first) {  // (once per workgroup: the first chunk, and the follow-up 1x1's filter behind the two chunk buffers)
          dma_chunk(0, 0);
          if (POST)
            copy_slab(reinterpret_cast<const bf16x8*>(a.wpk2), reinterpret_cast<bf16x8*>(smem + 2 * chunk_bytes), (MT / 2) * MT * 64, tid);
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          first = false;
        }
        if (more) dma_chunk(kc + a.kchunk, (c + 1) & 1);  // (its buffer was last read in chunk c - 1: everybody has passed that barrier)
      } else if (multi || first) {
        if (!first) __syncthreads();  // every wave is done with the previous slab
        if (!(a.dbg & 4)) copy_slab(wsrc + (size_t)kc * MT * 64, reinterpret_cast<bf16x8*>(smem), nk * MT * 64, tid);
        if (POST && first)  // the follow-up 1x1's filter: MT/2 k-steps x MT tiles, right behind the slab
          copy_slab(reinterpret_cast<const bf16x8*>(a.wpk2),
                    reinterpret_cast<bf16x8*>(smem) + min(a.kchunk, a.ksteps) * MT * 64, (MT / 2) * MT * 64, tid);
        __syncthreads();
        first = false;
      }
      if (live) run_ksteps<KS, MT, NT, UNIFORM, D, UP>(a, rin, g, P, wl + ((async && (c & 1)) ? chunk_bytes / 16 : 0), kc, nk, xr, acc, rin2);
      if (async && more) {
        if (live) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(D * NT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
    if (!live) break;
    // next patch: addresses + first ring fill go out before this patch's stores
    int opix[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) opix[nt] = P.opix[nt];
    const int patch_done = patch;
    patch += 4;
    live = (pp + 1 < a.ppw) && patch < a.npatches;
    if (live) {
      patch_setup<NT, UP>(a, patch, n, lane_off, P, lane_off2);
#pragma unroll
      for (int d = 0; d < D; ++d) fetch_b<KS, NT, UNIFORM, UP>(a, rin, d, g, P, xr[d], rin2);
    }
    if constexpr (POST) {
      static_assert(!POST || (MT % 2 == 0), "POST needs an even tile count");
      // epilogue of the first conv in registers, then the 1x1 on the packed fragments
      bf16x8 xb2[MT / 2][NT];
#pragma unroll
      for (int p = 0; p < MT / 2; ++p)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[2 * p][nt][r];
            v[4 + r] = acc[2 * p + 1][nt][r];
          }
          if (a.act) {
            silu8(v);
          }
#pragma unroll
          for (int r = 0; r < 8; ++r) xb2[p][nt][r] = (__bf16)v[r];
        }
      const bf16x8* wl2 = wl + ((multi && a.slab2) ? 2 * a.kchunk : min(a.kchunk, a.ksteps)) * MT * 64;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const f32x4 b2 = *reinterpret_cast<const f32x4*>(a.bias2 + chan_in_block(MT, mt, g, 0));
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = b2;
      }
#pragma unroll
      for (int p = 0; p < MT / 2; ++p)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const bf16x8 w2 = wl2[(p * MT + mt) * 64];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, xb2[p][nt], acc[mt][nt], 0, 0, 0);
        }
      ConvArgs a2 = a;
      a2.Cout = a.post_cout;
      a2.act = a.post_act;
      if constexpr (DFL) {
        static_assert(!DFL || MT == 4, "the DFL epilogue takes one block of 64 channels");
        dfl_epilogue<NT>(a, patch_done * (16 * NT), g, n, acc);
        if (a.dfl_keep) epilogue<MT, NT>(a2, rout, rres, cb, g, opix, acc);
      } else {
        epilogue<MT, NT>(a2, rout, rres, cb, g, opix, acc);
      }
    } else {
      if constexpr (PF) {
        if (a.pf_count) pf_sink<MT, NT>(a, acc, patch_done, g, n, a.Cout);
      }
      if constexpr (DFL) {
        static_assert(!DFL || MT == 4, "the DFL epilogue takes one block of 64 channels");
        dfl_epilogue<NT>(a, patch_done * (16 * NT), g, n, acc);
        if (a.dfl_keep) epilogue<MT, NT, O8>(a, rout, rres, cb, g, opix, acc);
      } else {
        epilogue<MT, NT, O8>(a, rout, rres, cb, g, opix, acc);
      }
    }
  }
}

template <int KS, int MT, int NT, bool UNIFORM, bool POST, bool UP = false, bool O8 = false, bool PF = false, bool DFL = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(YL_CONV_WPE))) void conv_kernel(ConvArgs a) {
  conv_body<KS, MT, NT, UNIFORM, POST, UP, O8, PF, DFL>(a, blockIdx.x);
}

// Up to three convs of ONE tiling in one launch ("horizontal" fusion): the three Detect levels run the same box / class
// branch on 80x80, 40x40 and 20x20 maps (reference nn/modules/head.py:59-62: `for i in range(self.nl)`), and a launch on
// a 20x20 map cannot fill 256 CUs by itself.  Workgroup -> (conv, block inside it) by the running block counts; every
// conv keeps its own arguments (pointers, geometry, k-steps, slab chunking), only MT / NT / the LDS size are shared.
struct ConvGroup {
  ConvArgs a[3];
  int nb[3];  // workgroups of each member (unused members: 0)
};

template <int KS, int MT, int NT, bool UNIFORM, bool POST, bool DFL = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(YL_CONV_WPE))) void conv_group_kernel(ConvGroup G) {
  int block = blockIdx.x, m = 0;
  if (block >= G.nb[0]) {
    block -= G.nb[0];
    m = 1;
    if (block >= G.nb[1]) {
      block -= G.nb[1];
      m = 2;
    }
  }
  conv_body<KS, MT, NT, UNIFORM, POST, false, false, false, DFL>(G.a[m], block);
}

// ------------------------------------------------------- depthwise 3x3 -> pointwise 1x1 in one launch
// The Detect class branch is Sequential(DWConv(c, c, 3), Conv(c, c3, 1)) twice (head.py:41-48).  The
// depthwise output is consumed by that 1x1 conv only, so a workgroup computes it for its 64*NT pixels
// straight into LDS (bf16, [plane][pixel][8] = MFMA B-fragment order) and runs the 1x1 GEMM from there:
// the intermediate tensor is never written to or read from HBM.  Needs all output channels in one
// workgroup (Cout <= 80 here: MT <= 5, ncb == 1).
// PW2: a further 1x1 conv that consumes ONLY the pointwise conv's output (head.py:47: the nn.Conv2d(c3, nc, 1) that
// ends the class branch) runs behind it from LDS: the pointwise result is written into the depthwise tile's place in
// B-fragment order and multiplied by the second filter slab; that tensor is never stored either.
template <int MT, int NT, bool PW2 = false>
__device__ __forceinline__ void dwpw_body(const ConvArgs& a, int block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int P = 64 * NT;  // pixels per workgroup
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, n = lane & 15;
  const int pblk = xcd_remap(block, a.nblocks);
  bf16x8* slab = reinterpret_cast<bf16x8*>(smem);
  bf16x8* tile = slab + a.ksteps * MT * 64;                       // [max(cg, cg2)][P]
  const int tile_planes = PW2 ? max(a.cg, a.cg2) : a.cg;
  bf16x8* slab2 = tile + tile_planes * P;                          // PW2: [ksteps2][MT][64]
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, (short)0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, a.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)nullptr, (short)0, 0, 0x00020000);

#define DW_STAMP(id) YL_STAMP(a.stamps, 4, wave, lane, id)
  DW_STAMP(0);
  // The pointwise filters are not needed before the depthwise phase is over: they travel to LDS by LDS-DMA (no registers, nobody
  // waits) while that phase runs, and are waited for in front of the barrier that ends it.  (Copied through registers with a barrier
  // of their own in front of the depthwise phase they were 2.0 of a workgroup's 15.9 us: tools/dwpw_stamps.py.)
  {
    typedef __attribute__((address_space(3))) void lds_void_d;
    const int n1 = a.ksteps * MT, n2 = PW2 ? a.ksteps2 * MT : 0;  // 1-KiB pieces
    const __amdgpu_buffer_rsrc_t rw1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk, (short)0, (unsigned)n1 * 1024u, 0x00020000);
    for (int pc = wave; pc < n1; pc += 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw1, (lds_void_d*)(reinterpret_cast<unsigned char*>(slab) + pc * 1024), 16, pc * 1024 + lane * 16, 0, 0, 0);
    if constexpr (PW2) {
      const __amdgpu_buffer_rsrc_t rw2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk2, (short)0, (unsigned)n2 * 1024u, 0x00020000);
      for (int pc = wave; pc < n2; pc += 4)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw2, (lds_void_d*)(reinterpret_cast<unsigned char*>(slab2) + pc * 1024), 16, pc * 1024 + lane * 16, 0, 0, 0);
    }
  }
  DW_STAMP(1);
  DW_STAMP(2);

  // ---- depthwise 3x3 + bias + SiLU on the MATRIX pipe, each wave for the 16 * NT pixels its GEMM will consume.
  // A depthwise conv of 16 channels (two planes) is the GEMM out[16][px] = A[16][K] * B[K][px] with K = 2 planes x 9 taps
  // x 8 channels = 144 (padded to 5 k-steps of 32) and A block-diagonal: row r holds tap weight w[r][tap] at
  // k = (plane r / 8, tap, channel r % 8) and zeros elsewhere.  The B fragments are the same 16-byte tap loads a VALU
  // form makes (lane (n, g) of k-step s reads k-group q = 4 s + g = (plane q / 9, tap q % 9) of pixel n), but the
  // 144 cvt + fma per lane and pixel become 5 MFMAs per 16 pixels: 94 % of the multiplies hit zeros and the matrix pipe
  // is still far cheaper than the VALU here because it was idle.  Each lane's A fragment has ONE non-zero element
  // (row r, element r % 8, when q / 9 == r / 8), so the host packs just that value per (16-channel block, lane, k-step)
  // (dw_pack_mfma: 16 bytes per lane) and the fragment is rebuilt with four ANDs.  Filter taps are bf16 on this path,
  // like every dense filter.  No per-item index arithmetic: the pixel offsets are patch_setup's.
  const int patch = pblk * 4 + wave;
  const bool live = patch < a.npatches;
  Patch<NT> Pp;
  patch_setup<NT>(a, live ? patch : 0, n, 0u, Pp);
  if (!(a.dbg & 2)) {
    unsigned toff[5];
#pragma unroll
    for (int s5 = 0; s5 < 5; ++s5) {
      const int q = 4 * s5 + g, sel = q >= 9, tap = q - 9 * sel;
      toff[s5] = q < 18 ? (unsigned)(sel * (int)a.in_plane + ((tap / 3 - 1) * a.in_w1 + (tap % 3 - 1)) * 16) : OOB_K;
    }
    const int e = n & 7;  // the lane's A row is channel n of the block: element n % 8 of its k-group
    unsigned selm[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) selm[d] = (e >> 1) == d ? ((e & 1) ? 0xffff0000u : 0x0000ffffu) : 0u;
    const u32x4* dwtab = reinterpret_cast<const u32x4*>(a.dw_wpk);
    unsigned char* tb = reinterpret_cast<unsigned char*>(tile);
    const int npp = live ? (a.cg >> 1) : 0;
    // NT == 1: the tap fragments, the packed taps and the bias of channel block pp + 1 are requested BEFORE block pp is
    // computed (two named register sets): a deep launch on a small map (256 channels @20x20: 16 blocks per workgroup, the
    // chip half empty) is a chain of memory round trips otherwise (24.1 -> 21.2 us).  A block index past the end is
    // clamped: the loads repeat the last block's.
    struct DwIn {
      u32x4 t;       // k-steps 0..4 as bf16 halves: {s0 | s1 << 16, s2 | s3 << 16, s4, -}
      f32x4 bv;
      bf16x8 xb[NT][5];
    };
    auto request = [&](int pp, DwIn& q) __attribute__((always_inline)) {
      pp = min(pp, npp - 1);
      q.t = dwtab[pp * 64 + lane];
      q.bv = *reinterpret_cast<const f32x4*>(a.dw_b + pp * 16 + 4 * g);
      const unsigned poff = (unsigned)(2 * pp) * a.in_plane;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s5 = 0; s5 < 5; ++s5)
          q.xb[nt][s5] = __builtin_bit_cast(
              bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rin, ((a.dbg & 1) ? OOB_LANE : Pp.base[nt]) + poff + toff[s5], 0, 0));
    };
    auto compute = [&](int pp, const DwIn& q) __attribute__((always_inline)) {
      f32x4 dacc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) dacc[nt] = q.bv;
#pragma unroll
      for (int s5 = 0; s5 < 5; ++s5) {
        const unsigned w16 = (s5 & 1) ? (q.t[s5 >> 1] >> 16) : (q.t[s5 >> 1] & 0xffffu);
        const unsigned both = w16 | (w16 << 16);
        u32x4 wv;
#pragma unroll
        for (int d = 0; d < 4; ++d) wv[d] = both & selm[d];
        const bf16x8 wa = __builtin_bit_cast(bf16x8, wv);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) dacc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, q.xb[nt][s5], dacc[nt], 0, 0, 0);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        // (one wave-uniform branch around the four SiLUs: a select per value made hipcc branch around each of them)
        if (a.dw_act) {
          dacc[nt] = silu4(dacc[nt]);
        }
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (__bf16)dacc[nt][r];
        *reinterpret_cast<bf16x4*>(tb + ((size_t)(2 * pp + (g >> 1)) * P + wave * 16 * NT + nt * 16 + n) * 16 + (g & 1) * 8) = o;
      }
    };
    if constexpr (NT == 1 || (PW2 && YL_DWPW_PF2)) {  // (16 pixels per wave = the deep launches on small maps, see yl_engine_create)
      if (npp > 0) {
        DwIn q0, q1;
        request(0, q0);
        for (int pp = 0; pp < npp; pp += 2) {
          request(pp + 1, q1);
          compute(pp, q0);
          if (pp + 1 < npp) {
            request(pp + 2, q0);
            compute(pp + 1, q1);
          }
        }
      }
    } else {  // 32 pixels per wave: the chip-filling launches are bound by L1 tap traffic, and the second register set costs
              // a wave per SIMD (64 -> 80 @80x80: 28.4 -> 31.3 us with the prefetch)
      for (int pp = 0; pp < npp; ++pp) {
        DwIn q;
        request(pp, q);
        compute(pp, q);
      }
    }
  }
  DW_STAMP(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this wave's pieces of the filter slabs)
  __syncthreads();
  DW_STAMP(4);

  // ---- 1x1 GEMM from LDS
  if (!PW2 && !live) return;
  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + chan_in_block(MT, mt, g, 0));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = bv;
  }
  const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int ks = 0; ks < ((a.dbg & 4) ? 0 : a.ksteps); ++ks) {
    bf16x8 wa[MT], xb[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) wa[mt] = slab[(ks * MT + mt) * 64 + lane];
    const int k8 = ks * 4 + g;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) xb[nt] = k8 < a.cg ? tile[k8 * P + wave * 16 * NT + nt * 16 + n] : zero;
    mma_step<MT, NT>(wa, xb, acc);
  }
  DW_STAMP(5);
  if constexpr (!PW2) {
    if (a.dbg & 8) return;
    epilogue<MT, NT>(a, rout, rres, 0, g, Pp.opix, acc);
    DW_STAMP(9);
  } else {
    // pointwise result (SiLU, bf16: exactly what the unfused kernel stores) -> the tile, as planes of 8 channels:
    // rows of tile mt are channels chan_in_block(MT, mt, g, 0..3), i.e. half a 16-byte pixel of one plane
    __syncthreads();  // every wave has read its depthwise operands
    unsigned char* tb = reinterpret_cast<unsigned char*>(tile);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int ch = chan_in_block(MT, mt, g, 0);
      if (ch < a.cg2 * 8) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if (a.act) {
            acc[mt][nt] = silu4(acc[mt][nt]);
          }
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (__bf16)acc[mt][nt][r];
          *reinterpret_cast<bf16x4*>(tb + ((size_t)(ch >> 3) * P + wave * 16 * NT + nt * 16 + n) * 16 + (ch & 7) * 2) = o;
        }
      }
    }
    DW_STAMP(6);
    __syncthreads();
    DW_STAMP(7);
    if (!live) return;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias2 + chan_in_block(MT, mt, g, 0));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = bv;
    }
    for (int ks = 0; ks < a.ksteps2; ++ks) {
      bf16x8 wa[MT], xb[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) wa[mt] = slab2[(ks * MT + mt) * 64 + lane];
      const int k8 = ks * 4 + g;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) xb[nt] = k8 < a.cg2 ? tile[k8 * P + wave * 16 * NT + nt * 16 + n] : zero;
      mma_step<MT, NT>(wa, xb, acc);
    }
    ConvArgs a2 = a;
    a2.Cout = a.post_cout;
    a2.act = a.post_act;
    if (a.pf_count && a.post_act) {  // (the activation once, here, under ONE branch: a select per value would compute 40 SiLUs for nothing)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = silu4(acc[mt][nt]);
      a2.act = 0;
    }
    if (a.pf_count) pf_sink<MT, NT>(a, acc, patch, g, n, a.post_cout);
    DW_STAMP(8);
    epilogue<MT, NT>(a2, rout, rres, 0, g, Pp.opix, acc);
    DW_STAMP(9);
  }
#undef DW_STAMP
}

template <int MT, int NT, bool PW2 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(YL_DWPW_WPE))) void dwpw_kernel(ConvArgs a) {
  dwpw_body<MT, NT, PW2>(a, blockIdx.x);
}

// Deep launches on small maps (Detect level 2 of YOLO11n: 256 channels @20x20, 12 800 pixels per batch of 32): dwpw_kernel
// gives every wave ALL channel blocks of its 16 pixels -- 16 dependent rounds of tap loads + 5 MFMAs -- and fills 200
// workgroups.  Here a workgroup owns ONE 16-pixel patch and its four waves split the work: wave w the channel blocks
// w, w + 4, ... of the depthwise conv (into the LDS tile), then the channel-tile PAIR w of the pointwise GEMM (its filter
// fragments straight from global memory, requested before the depthwise phase; no 40 KB slab per workgroup).  Four times
// the workgroups, a quarter of the serial chain.  Same MFMAs in the same order as dwpw_kernel: bit-identical.
// A pair of tiles (2 w, 2 w + 1) of an MT-tile block stores exactly what block w of a two-tile tiling stores, the odd last
// tile what block MT - 1 of a one-tile tiling stores (chan_in_block): the shared epilogue is called that way.
template <int MT>
__global__ __launch_bounds__(256) void dwpw_deep_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int P = 16;
  constexpr int UNITS = (MT + 1) / 2;
  static_assert(UNITS <= 4, "one pair of channel tiles per wave");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, n = lane & 15;
  const int patch = xcd_remap(blockIdx.x, a.npatches);
  bf16x8* tile = reinterpret_cast<bf16x8*>(smem);  // [cg][16]
  const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, (short)0, a.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, (short)0, a.out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)nullptr, (short)0, 0, 0x00020000);
  Patch<1> Pp;
  patch_setup<1>(a, patch, n, 0u, Pp);

  const bool has = wave < UNITS;
  const int mt0 = 2 * wave;
  const bool two = mt0 + 1 < MT;
  const bf16x8* wsrc = reinterpret_cast<const bf16x8*>(a.wpk);
  const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  bf16x8 w[8][2];
  auto loadw = [&](int ks0) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ks = min(ks0 + j, a.ksteps - 1);
      w[j][0] = wsrc[(ks * MT + mt0) * 64 + lane];
      w[j][1] = two ? wsrc[(ks * MT + mt0 + 1) * 64 + lane] : zero;
    }
  };
  if (has) loadw(0);

  {  // ---- depthwise 3x3 + bias + SiLU on the matrix pipe (dwpw_body's scheme), this wave's channel blocks
    unsigned toff[5];
#pragma unroll
    for (int s5 = 0; s5 < 5; ++s5) {
      const int q = 4 * s5 + g, sel = q >= 9, tap = q - 9 * sel;
      toff[s5] = q < 18 ? (unsigned)(sel * (int)a.in_plane + ((tap / 3 - 1) * a.in_w1 + (tap % 3 - 1)) * 16) : OOB_K;
    }
    const int e = n & 7;
    unsigned selm[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) selm[d] = (e >> 1) == d ? ((e & 1) ? 0xffff0000u : 0x0000ffffu) : 0u;
    const u32x4* dwtab = reinterpret_cast<const u32x4*>(a.dw_wpk);
    unsigned char* tb = reinterpret_cast<unsigned char*>(tile);
    const int npp = a.cg >> 1;
    struct DwIn {
      u32x4 t;
      f32x4 bv;
      bf16x8 xb[5];
    };
    auto request = [&](int pp, DwIn& q) __attribute__((always_inline)) {
      pp = min(pp, npp - 1);
      q.t = dwtab[pp * 64 + lane];
      q.bv = *reinterpret_cast<const f32x4*>(a.dw_b + pp * 16 + 4 * g);
      const unsigned poff = (unsigned)(2 * pp) * a.in_plane;
#pragma unroll
      for (int s5 = 0; s5 < 5; ++s5)
        q.xb[s5] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rin, Pp.base[0] + poff + toff[s5], 0, 0));
    };
    auto compute = [&](int pp, const DwIn& q) __attribute__((always_inline)) {
      f32x4 dacc = q.bv;
#pragma unroll
      for (int s5 = 0; s5 < 5; ++s5) {
        const unsigned w16 = (s5 & 1) ? (q.t[s5 >> 1] >> 16) : (q.t[s5 >> 1] & 0xffffu);
        const unsigned both = w16 | (w16 << 16);
        u32x4 wv;
#pragma unroll
        for (int d = 0; d < 4; ++d) wv[d] = both & selm[d];
        dacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv), q.xb[s5], dacc, 0, 0, 0);
      }
      if (a.dw_act) {
        dacc = silu4(dacc);
      }
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (__bf16)dacc[r];
      *reinterpret_cast<bf16x4*>(tb + ((size_t)(2 * pp + (g >> 1)) * P + n) * 16 + (g & 1) * 8) = o;
    };
    // four channel blocks requested at once (four register sets: 256 channels = ONE round of loads per wave)
    for (int pp0 = wave; pp0 < npp; pp0 += 16) {
      DwIn q[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) request(pp0 + 4 * j, q[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (pp0 + 4 * j < npp) compute(pp0 + 4 * j, q[j]);
    }
  }
  __syncthreads();
  if (!has) return;

  // ---- 1x1 GEMM: B from the LDS tile, this wave's pair of channel tiles
  f32x4 acc[2][1];
  acc[0][0] = *reinterpret_cast<const f32x4*>(a.bias + chan_in_block(MT, mt0, g, 0));
  acc[1][0] = two ? *reinterpret_cast<const f32x4*>(a.bias + chan_in_block(MT, mt0 + 1, g, 0)) : f32x4{0, 0, 0, 0};
  for (int ks0 = 0; ks0 < a.ksteps; ks0 += 8) {
    if (ks0) loadw(ks0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (ks0 + j < a.ksteps) {
        const int k8 = (ks0 + j) * 4 + g;
        const bf16x8 xb = k8 < a.cg ? tile[k8 * P + n] : zero;
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][0], xb, acc[0][0], 0, 0, 0);
        if (two) acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][1], xb, acc[1][0], 0, 0, 0);
      }
    }
  }
  if (two) {
    epilogue<2, 1>(a, rout, rres, wave, g, Pp.opix, acc);
  } else {
    f32x4 acc1[1][1];
    acc1[0][0] = acc[0][0];
    epilogue<1, 1>(a, rout, rres, MT - 1, g, Pp.opix, acc1);
  }
}

// the same for up to three launches of one tiling (the class branch's second half is identical on every Detect level)
template <int MT, int NT, bool PW2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(YL_DWPW_WPE))) void dwpw_group_kernel(ConvGroup G) {
  int block = blockIdx.x, m = 0;
  if (block >= G.nb[0]) {
    block -= G.nb[0];
    m = 1;
    if (block >= G.nb[1]) {
      block -= G.nb[1];
      m = 2;
    }
  }
  dwpw_body<MT, NT, PW2>(G.a[m], block);
}

// Convs AND depthwise + pointwise launches in one: the box branch and the class branch of the Detect head are
// independent of each other (head.py:61: cat(cv2[i](x), cv3[i](x))), so their launches can share the chip as well as
// the levels do.  Workgroups [0, n_conv) run conv_body on the members of GC, the rest dwpw_body on the members of GD.
// The register budget is the larger of the two bodies' (the conv's), the LDS size the larger of the members'.
template <int NTC, bool POST, int NTD, bool PW2, bool DFL = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void head_group_kernel(ConvGroup GC, ConvGroup GD, int n_conv) {
  int block = blockIdx.x;
  if (block < n_conv) {
    int m = 0;
    if (block >= GC.nb[0]) {
      block -= GC.nb[0];
      m = 1;
      if (block >= GC.nb[1]) {
        block -= GC.nb[1];
        m = 2;
      }
    }
    conv_body<3, 4, NTC, true, POST, false, false, false, DFL>(GC.a[m], block);
  } else {
    block -= n_conv;
    int m = 0;
    if (block >= GD.nb[0]) {
      block -= GD.nb[0];
      m = 1;
      if (block >= GD.nb[1]) {
        block -= GD.nb[1];
        m = 2;
      }
    }
    dwpw_body<5, NTD, PW2>(GD.a[m], block);
  }
}

// ------------------------------------------------------------------------------------ host
static unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull / (unsigned)d) + 1ull); }

static const size_t LDS_MAX = 160 * 1024;  // per CU and per workgroup on gfx950
static const int NUM_CU = 256;

// conv_choose_even: tilings with an even number of channel tiles per wave only (the e4m3 epilogue copy, yl_op.out8, is written
// by tile pairs)
static thread_local bool g_even_only = false;
int conv_choose_even(int batch, int Cin, int Cout, int k, int stride, int Ho, int Wo, ConvConfig* c) {
  g_even_only = true;
  const int rc = conv_choose(batch, Cin, Cout, k, stride, Ho, Wo, 0, c);
  g_even_only = false;
  if (rc == 0 && (c->MT & 1)) return fail(-3, "conv: no tiling with an even number of channel tiles for this channel count");
  return rc;
}

int conv_choose(int batch, int Cin, int Cout, int k, int stride, int Ho, int Wo, int post, ConvConfig* c) {
  if ((k != 1 && k != 3) || (stride != 1 && stride != 2) || Cin % 8 || Cin <= 0 || Cout <= 0)
    return fail(-3, "conv: unsupported geometry (k in {1,3}, stride in {1,2}, Cin % 8 == 0)");
  const int cout16 = (Cout + 15) / 16;
  const int ntaps = k * k;
  const int cg = Cin / 8;
  const int ksteps = (ntaps * cg + 3) / 4;
  // Slab budget.  72 KiB (two workgroups per CU side by side, a 64 -> 64 3x3 filter in one piece) was the best choice
  // while 8 batches were in flight; with 20 (round 2) the CUs are shared by more launches at a time and a smaller
  // slab -- bigger filters are restaged in chunks -- leaves them room: 40-56 KiB all give 47.6-49.0 k images/s against
  // 45.2-46.3 k for 64-72 KiB and 46.1-46.4 k for 100 KiB (tools/env_ab.sh, alternating runs on one box).
  static const size_t slab_max =
      (size_t)tune_int("YL_CONV_SLAB_KB", 48) * 1024;
  // output-channel tiles per wave: as many as divide the layer (B-fragment traffic goes with 1/MT).
  // YL_CONV_CHUNK=0 restores the older policy of shrinking MT until the whole filter fits the slab.
  static const int allow_chunk = tune_int("YL_CONV_CHUNK", 1);
  int mts[5], nmt = 0;
  if (cout16 % 4 == 0) mts[nmt++] = 4;
  if (cout16 % 5 == 0 && !g_even_only) mts[nmt++] = 5;
  if (cout16 % 3 == 0 && !g_even_only) mts[nmt++] = 3;  // 48-channel layers (YOLO11x Bottlenecks) would otherwise run one tile per wave
  if (cout16 % 2 == 0) mts[nmt++] = 2;
  mts[nmt++] = 1;
  int MT = mts[0];
  // six tiles per wave (96 output channels): channel counts that are multiples of 96 but not of 64 (YOLO11x / m) would otherwise
  // fall to two tiles; YL_CONV_MT6=2 also prefers it to four tiles on wide layers (fewer re-reads of the activations)
  const int mt6 = tune_int("YL_CONV_MT6", 2);
  if (!post && cout16 % 6 == 0 && Cin % 32 == 0 && ((mt6 >= 1 && cout16 % 4 != 0) || (mt6 >= 2 && Cin >= 128))) MT = 6;
  // eight tiles x two pixel tiles per wave (128 channels x 32 pixels): half the activation re-reads of 4 x 4 at twice the filter
  // reads from LDS -- a trade that pays while the texture path, not LDS, is the busier one (YL_CONV_MT8 = smallest Cin that takes it)
  const int mt8 = env_int("YL_CONV_MT8", 128);
  const int mt8_cout = tune_int("YL_CONV_MT8_COUT", 256);  // smallest Cout that takes it
  const int mt8_blocks = env_int("YL_CONV_MT8_BLOCKS", 512);
  if (mt8 > 0 && !post && cout16 % 8 == 0 && Cin % 32 == 0 && Cin >= mt8 && Cout >= mt8_cout && MT == 4 &&
      ((long long)batch * Ho * Wo + 31) / 32 / 4 * (cout16 / 8) >= mt8_blocks)  // still at least two workgroups per CU
    MT = 8;
  if (!allow_chunk)
    for (int mi = 0; mi < nmt; ++mi) {
      MT = mts[mi];
      if ((size_t)ksteps * MT * 1024 <= slab_max) break;
    }
  int kchunk = ksteps;
  if ((size_t)ksteps * MT * 1024 > slab_max) {
    kchunk = (int)(slab_max / ((size_t)MT * 1024)) & ~7;
    if (kchunk < 8) kchunk = 8;  // the smallest chunk the k-loop takes (eight tiles per wave: a 64 KiB slab whatever the budget)
    if ((size_t)kchunk * MT * 1024 > 96 * 1024) return fail(-3, "conv: filter slab budget too small");
  }
  const int ncb = cout16 / MT;
  const long long npix = (long long)batch * Ho * Wo;
  auto npatches = [&](int nt) { return (npix + 16 * nt - 1) / (16 * nt); };
  int NT = 1;
  static const int nt4_blocks = tune_int("YL_CONV_NT4", NUM_CU / 2);
  static const int nt2_blocks = tune_int("YL_CONV_NT2", NUM_CU / 4);
  if (npatches(4) * ncb / 4 >= nt4_blocks) NT = 4;
  else if (npatches(2) * ncb / 4 >= nt2_blocks) NT = 2;
  if (MT == 5 && NT == 4) NT = 2;
  if (MT == 6 && NT == 1) NT = 2;
  if (MT == 8) NT = 2;
  c->MT = MT;
  c->NT = NT;
  c->ncb = ncb;
  c->cg = cg;
  c->ntaps = ntaps;
  c->uniform = (cg % 4 == 0);
  c->ksteps = ksteps;
  c->kchunk = kchunk;
  // A chunked filter alternates between TWO chunk buffers filled by LDS-DMA (conv_body; YL_CONV_SLAB2: 1 = two buffers inside the
  // same budget -- chunks of half the k-steps --, 2 = two buffers of the old chunk size, 0 = one buffer restaged through
  // registers between two barriers, the form of rounds 1-4).  Chunks stay multiples of 4 k-steps: the activation ring of 32- and
  // 64-pixel tiles is 4 / 2 k-steps deep (16-pixel tiles, ring of 8, keep the synchronous form).
  c->slab2 = 0;
  {
    const int mode = tune_int("YL_CONV_SLAB2", 1);
    if (mode > 0 && kchunk < ksteps && NT >= 2 && YL_RING_NT4 <= 4 && YL_RING_NT2 <= 4) {
      if (mode == 1) {
        kchunk = (kchunk / 2) & ~3;
        if (kchunk < 4) kchunk = 4;
      }
      c->kchunk = kchunk;
      c->slab2 = 1;
    }
  }
  c->lds_bytes = (size_t)(kchunk < ksteps ? (c->slab2 ? 2 * kchunk : kchunk) : ksteps) * MT * 1024;
  c->post = post;
  if (post) {
    // follow-up 1x1 in the epilogue: this wave must hold every channel of the first conv
    if (ncb != 1 || (MT != 2 && MT != 4) || Cout != MT * 16)
      return fail(-3, "conv: a fused follow-up 1x1 needs Cout in {32, 64} of the first conv");
    c->lds_bytes += (size_t)(MT / 2) * MT * 1024;
  }
  c->wpk_elems = (size_t)ncb * ksteps * MT * 64 * 8;
  c->npatches = (int)npatches(NT);
  // patches per wave: amortise the slab load while keeping >= ~8 blocks per CU in the grid
  long long blocks1 = ((long long)c->npatches + 3) / 4 * ncb;
  static const int ppw_div = tune_int("YL_CONV_BPC", 32);  // target blocks per CU (8 until several batches ran in flight: +0.7 %)
  int ppw = (int)(blocks1 / (ppw_div * NUM_CU));
  if (ppw < 1) ppw = 1;
  if (ppw > 32) ppw = 32;
  if (kchunk < ksteps) ppw = 1;
  c->ppw = ppw;
  c->gws = (int)(((long long)c->npatches + 4 * ppw - 1) / (4 * ppw));
  c->name = k == 3 ? (stride == 2 ? "conv3x3s2_mfma" : "conv3x3_mfma") : "conv1x1_mfma";
  c->gemm = 0;
  c->cs = 0;
  // K order of the stride-2 3x3 convs with >= 64 input channels (layers 3, 5, 7, 17, 20): PLANE-major.  In tap-major order a
  // workgroup sweeps its whole input footprint (8-32 planes) once per tap, nine times in all, with a reuse distance far beyond
  // L1 and -- 64 workgroups per XCD -- beyond the 4 MB L2: rocprofv3 (profiles/r03_summary.md) shows model.3 fetching 225 MB for
  // a 105 MB input.  Plane-major touches one plane's footprint with all nine taps back to back and never again.  Costs the
  // per-lane (tap, plane) arithmetic of the non-UNIFORM fetch (~10 VALU per k-step); the sum order changes (same operands).
  static const int kord_min_cin = tune_int("YL_CONV_KORD", 64);  // 0: never
  static const int kord_s1_cin = tune_int("YL_CONV_KORD_S1", 0);  // stride 1 too, from this Cin (0: never)
  c->kord = (kord_min_cin > 0 && k == 3 && stride == 2 && Cin >= kord_min_cin && MT <= 5 && !g_even_only) ? 1 : 0;
  if (kord_s1_cin > 0 && k == 3 && stride == 1 && Cin >= kord_s1_cin && MT <= 5 && !g_even_only && Cin % 8 == 0) c->kord = 1;
  if (c->kord) c->uniform = 0;
  conv3_reg_choose(batch, Cin, Cout, k, stride, Ho, Wo, post, c);   // ... or with the filter in registers
  conv3_cs_choose(batch, Cin, Cout, k, stride, Ho, Wo, post, c);    // ... 64 -> 64: output channels split, two workgroups per CU
  if (c->kord && (c->reg || c->cs)) {  // those kernels read the filter tap-major
    c->kord = 0;
    c->uniform = (cg % 4 == 0);
  }
  return 0;
}

static inline uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// dwpw_kernel's depthwise filter for its MFMA form: [C / 16][64 lanes][8] bf16.  Lane (row r = lane % 16, k-group
// g = lane / 16) holds, for k-steps s = 0..4, the ONE non-zero element of its A fragment: tap weight w[r][q % 9] when
// k-group q = 4 s + g belongs to the row's plane (q < 18 and q / 9 == r / 8), else 0.  w: [C][9] fp32 (folded BN).
void dw_pack_mfma(const float* w, int C, std::vector<uint16_t>* out) {
  out->assign((size_t)(C / 16) * 64 * 8, 0);
  for (int pp = 0; pp < C / 16; ++pp)
    for (int lane = 0; lane < 64; ++lane)
      for (int s = 0; s < 5; ++s) {
        const int row = lane & 15, q = 4 * s + (lane >> 4);
        if (q < 18 && q / 9 == row / 8) (*out)[((size_t)pp * 64 + lane) * 8 + s] = f2bf(w[(size_t)(pp * 16 + row) * 9 + q % 9]);
      }
}

void conv_pack_weights(const float* w, int Cout, int Cin_w, int k, const ConvConfig& c, std::vector<uint16_t>* out) {
  out->assign(c.wpk_elems, 0);
  size_t o = 0;
  for (int cb = 0; cb < c.ncb; ++cb)
    for (int ks = 0; ks < c.ksteps; ++ks)
      for (int mt = 0; mt < c.MT; ++mt)
        for (int lane = 0; lane < 64; ++lane) {
          const int co = cb * c.MT * 16 + chan_in_block(c.MT, mt, (lane & 15) >> 2, lane & 3);
          const int k8 = ks * 4 + (lane >> 4);
          int tap = c.kord ? k8 % c.ntaps : k8 / c.cg, c8 = c.kord ? k8 / c.ntaps : k8 % c.cg;
          if (c.kord == 2) {  // conv_gemm_kernel's 3x3 order: k-step = (block of 4 planes, tap)
            tap = ks % c.ntaps;
            c8 = (ks / c.ntaps) * 4 + (lane >> 4);
          }
          for (int j = 0; j < 8; ++j, ++o) {
            const int ci = c8 * 8 + j;
            if (co < Cout && tap < c.ntaps && ci < Cin_w) {
              const int kh = tap / k, kw = tap % k;
              (*out)[o] = f2bf(w[(((size_t)co * Cin_w + ci) * k + kh) * k + kw]);
            }
          }
        }
}

// Whether conv_kernel can carry the NMS candidate pre-filter for a conv of this tiling (launch_pf: 80 classes = five channel
// tiles in one wave, plain 1x1, not on the GEMM kernel)
bool conv_pf_supported(const ConvConfig& cfg) {
  return cfg.ntaps == 1 && cfg.MT == 5 && cfg.ncb == 1 && cfg.uniform && !cfg.post && !cfg.gemm && (cfg.NT == 1 || cfg.NT == 2);
}

// a conv that computes a Detect level's DFL logits (ConvArgs::dfl_out): the 1x1 itself, or the 3x3 in front of it with the 1x1 fused
template <int KS, int NT, bool U, bool POST>
static int launch_dfl(const ConvConfig& cfg, const ConvArgs& a, hipStream_t s) {
  if (a.out8 || a.in2 || a.res || a.out_f32 || a.pf_count || cfg.ncb != 1 || (POST ? a.post_cout : a.Cout) != 64)
    return fail(-3, "conv: the DFL epilogue needs a plain conv whose 64 output channels sit in one wave");
  auto kern = conv_kernel<KS, 4, NT, U, POST, false, false, false, true>;
  static size_t attr = 0;
  if (cfg.lds_bytes > attr) {
    YL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX));
    attr = LDS_MAX;
  }
  hipLaunchKernelGGL(kern, dim3(a.nblocks), dim3(256), cfg.lds_bytes, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

template <int KS, int MT, int NT, bool U, bool POST, bool O8 = false>
static int launch_p(const ConvConfig& cfg, const ConvArgs& a, hipStream_t s) {
  if (a.dfl_out) {
    if constexpr (MT == 4 && !O8 && (POST || KS == 1)) return launch_dfl<KS, NT, U, POST>(cfg, a, s);
    return fail(-3, "conv: the DFL epilogue needs a 1x1 conv (alone or fused behind a 3x3) with four channel tiles per wave");
  }
  if constexpr (!O8 && !POST && (MT == 2 || MT == 4 || MT == 6 || MT == 8)) {
    if (a.out8) return launch_p<KS, MT, NT, U, POST, true>(cfg, a, s);
  }
  if (a.out8 && !O8) return fail(-3, "conv: an e4m3 copy of the output needs 2, 4 or 6 channel tiles per wave and no fused follow-up conv");
  auto kern = conv_kernel<KS, MT, NT, U, POST, false, O8>;
  static size_t attr = 0;  // per instantiation: largest dynamic LDS size enabled so far
  if (cfg.lds_bytes > attr) {
    YL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)LDS_MAX));
    attr = LDS_MAX;
  }
  hipLaunchKernelGGL(kern, dim3(a.nblocks), dim3(256), cfg.lds_bytes, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

// a 1x1 conv that writes a Detect level's class logits and feeds the NMS candidate pre-filter (conv_pf_supported)
template <int MT, int NT>
static int launch_pf(const ConvConfig& cfg, const ConvArgs& a, hipStream_t s) {
  if (!conv_pf_supported(cfg) || a.in2 || a.out8 || a.wpk2 || a.act || a.out_f32)
    return fail(-3, "conv: the pre-filter sink needs a plain 1x1 conv without activation, all output channels in one wave");
  auto kern = conv_kernel<1, MT, NT, true, false, false, false, true>;
  static size_t attr = 0;
  if (cfg.lds_bytes > attr) {
    YL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)LDS_MAX));
    attr = LDS_MAX;
  }
  hipLaunchKernelGGL(kern, dim3(a.nblocks), dim3(256), cfg.lds_bytes, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

template <int MT, int NT, bool O8 = false>
static int launch_up(const ConvConfig& cfg, const ConvArgs& a, hipStream_t s) {
  if constexpr (!O8 && (MT == 2 || MT == 4 || MT == 6 || MT == 8)) {
    if (a.out8) return launch_up<MT, NT, true>(cfg, a, s);
  }
  if (a.out8 && !O8) return fail(-3, "conv: an e4m3 copy of the output needs 2, 4 or 6 channel tiles per wave");
  auto kern = conv_kernel<1, MT, NT, true, false, true, O8>;
  static size_t attr = 0;
  if (cfg.lds_bytes > attr) {
    YL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)LDS_MAX));
    attr = LDS_MAX;
  }
  hipLaunchKernelGGL(kern, dim3(a.nblocks), dim3(256), cfg.lds_bytes, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

template <int KS, int MT, int NT, bool U>
static int launch_t(const ConvConfig& cfg, const ConvArgs& a, hipStream_t s) {
  if (a.in2) {
    if constexpr (KS == 1 && U) {
      if (cfg.post) return fail(-3, "conv: upsampled second input and fused follow-up conv cannot be combined");
      return launch_up<MT, NT>(cfg, a, s);
    }
    return fail(-3, "conv: an upsampled second input needs a 1x1 conv with Cin % 32 == 0");
  }
  if constexpr (MT == 2 || MT == 4) {
    if (cfg.post) return launch_p<KS, MT, NT, U, true>(cfg, a, s);
  }
  if (cfg.post) return fail(-3, "conv: fused follow-up conv needs MT in {2, 4}");
  return launch_p<KS, MT, NT, U, false>(cfg, a, s);
}

template <int KS, int MT, int NT>
static int launch_u(const ConvConfig& cfg, const ConvArgs& a, hipStream_t s) {
  return cfg.uniform ? launch_t<KS, MT, NT, true>(cfg, a, s) : launch_t<KS, MT, NT, false>(cfg, a, s);
}

template <int KS, int MT>
static int launch_n(const ConvConfig& cfg, const ConvArgs& a, hipStream_t s) {
  return cfg.NT == 4 ? launch_u<KS, MT, 4>(cfg, a, s)
                     : cfg.NT == 2 ? launch_u<KS, MT, 2>(cfg, a, s) : launch_u<KS, MT, 1>(cfg, a, s);
}

template <int KS>
static int launch_m(const ConvConfig& cfg, const ConvArgs& a, hipStream_t s) {
  switch (cfg.MT) {
    case 1: return launch_n<KS, 1>(cfg, a, s);
    case 2: return launch_n<KS, 2>(cfg, a, s);
    case 3: return launch_n<KS, 3>(cfg, a, s);
    case 4: return launch_n<KS, 4>(cfg, a, s);
    case 5:
      if constexpr (KS == 1) {
        if (a.pf_count) return cfg.NT == 1 ? launch_pf<5, 1>(cfg, a, s) : launch_pf<5, 2>(cfg, a, s);
      }
      return cfg.NT == 1 ? launch_u<KS, 5, 1>(cfg, a, s) : launch_u<KS, 5, 2>(cfg, a, s);
    case 8:
      if (cfg.uniform && !cfg.post && cfg.NT == 2) return launch_t<KS, 8, 2, true>(cfg, a, s);
      break;
    case 6:
      if (cfg.uniform && !cfg.post)
        return cfg.NT == 4 ? launch_t<KS, 6, 4, true>(cfg, a, s) : launch_t<KS, 6, 2, true>(cfg, a, s);
      break;
  }
  return fail(-3, "conv: bad MT");
}

template <int MT, int NT, bool PW2>
static int launch_dwpw2(const ConvConfig& cfg, const ConvArgs& a, size_t lds, hipStream_t s) {
  auto kern = dwpw_kernel<MT, NT, PW2>;
  static size_t attr = 0;
  if (lds > attr) {
    YL_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)LDS_MAX));
    attr = LDS_MAX;
  }
  const int blocks = (cfg.npatches + 3) / 4;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, a);
  YL_HIP(hipGetLastError());
  return 0;
}

template <int MT, int NT>
static int launch_dwpw(const ConvConfig& cfg, const ConvArgs& a, size_t lds, hipStream_t s) {
  return a.wpk2 ? launch_dwpw2<MT, NT, true>(cfg, a, lds, s) : launch_dwpw2<MT, NT, false>(cfg, a, lds, s);
}

// dwpw_kernel's arguments and LDS size from the tiling; cfg may be retiled (development override of NT)
static int dwpw_prepare(ConvConfig& cfg, ConvArgs& a, int batch, size_t* lds_out) {
  static const int force_nt = tune_int("YL_DWPW_NT", 0);  // development: 1 or 2 positions tiles per wave
  if (force_nt == 1 || force_nt == 2) {
    cfg.NT = force_nt;
    cfg.npatches = (int)(((long long)batch * a.Ho * a.Wo + 16 * force_nt - 1) / (16 * force_nt));
  }
  if (cfg.ncb != 1 || cfg.kchunk < cfg.ksteps || cfg.NT > 2)
    return fail(-3, "dwpw: needs all output channels in one workgroup and NT <= 2");
  if (a.Cin % 16 || !a.dw_wpk) return fail(-3, "dwpw: the depthwise conv runs on 16-channel MFMA tiles (C % 16 == 0)");
  a.ncb = 1;
  a.cg = cfg.cg;
  a.ksteps = cfg.ksteps;
  a.kchunk = cfg.kchunk;
  a.ntaps = 1;
  a.magic_wo = magic_of(a.Wo);
  a.magic_ho = magic_of(a.Ho);
  a.npatches = cfg.npatches;
  a.ppw = 1;
  a.npix = batch * a.Ho * a.Wo;
  const long long lim = 1LL << 30;
  const int cout_stored = a.wpk2 ? a.post_cout : a.Cout;
  const long long inb = (long long)cfg.cg * a.in_plane, outb = (long long)((cout_stored + 7) / 8) * a.out_plane;
  if (inb > lim || outb > lim) return fail(-3, "dwpw: a tensor slice exceeds 1 GiB; lower the batch size");
  a.in_bytes = (unsigned)inb;
  a.out_bytes = (unsigned)outb;
  a.st_nt = st_nt_policy(outb);
  a.nblocks = (cfg.npatches + 3) / 4;
  static const int dwpw_dbg = tune_int("YL_DWPW_DBG", 0);  // timing ablations (wrong results)
  a.dbg = dwpw_dbg;
  a.stamps = g_stamp_buf;
  const int planes = a.wpk2 && a.cg2 > cfg.cg ? a.cg2 : cfg.cg;
  *lds_out = (size_t)cfg.ksteps * cfg.MT * 1024 + (size_t)planes * 64 * cfg.NT * 16 +
             (a.wpk2 ? (size_t)a.ksteps2 * cfg.MT * 1024 : 0);
  if (*lds_out > LDS_MAX) return fail(-3, "dwpw: LDS budget exceeded");
  return 0;
}

// depthwise 3x3 + pointwise 1x1 [+ a further 1x1] (see dwpw_kernel); cfg from conv_choose(Cin = C, k = 1)
int dwpw_launch(const ConvConfig& cfg0, ConvArgs a, int batch, hipStream_t s) {
  ConvConfig cfg = cfg0;
  size_t lds = 0;
  if (int rc = dwpw_prepare(cfg, a, batch, &lds)) return rc;
  {  // input staged through LDS (dwpw.hip) where the geometry allows
    int handled = 0;
    const int rc = dwpw_tile_launch(cfg, a, batch, s, &handled);
    if (rc || handled) return rc;
  }
  {  // deep launches on small maps: one 16-pixel patch per workgroup, the waves split channels (dwpw_deep_kernel)
    const int deep = env_int("YL_DWPW_DEEP", 1);  // (read per launch: the tests switch it)
    if (deep && cfg.NT == 1 && !a.wpk2 && cfg.cg >= 16 && !a.dbg) {
      a.nblocks = a.npatches;
      const size_t lds_deep = (size_t)cfg.cg * 256;
      switch (cfg.MT) {
        case 2: hipLaunchKernelGGL(dwpw_deep_kernel<2>, dim3(a.npatches), dim3(256), lds_deep, s, a); break;
        case 4: hipLaunchKernelGGL(dwpw_deep_kernel<4>, dim3(a.npatches), dim3(256), lds_deep, s, a); break;
        case 5: hipLaunchKernelGGL(dwpw_deep_kernel<5>, dim3(a.npatches), dim3(256), lds_deep, s, a); break;
        case 8: hipLaunchKernelGGL(dwpw_deep_kernel<8>, dim3(a.npatches), dim3(256), lds_deep, s, a); break;
        default: return fail(-3, "dwpw: unsupported tile shape");
      }
      YL_HIP(hipGetLastError());
      return 0;
    }
  }
  switch (cfg.MT * 10 + cfg.NT) {
    case 11: return launch_dwpw<1, 1>(cfg, a, lds, s);
    case 12: return launch_dwpw<1, 2>(cfg, a, lds, s);
    case 21: return launch_dwpw<2, 1>(cfg, a, lds, s);
    case 22: return launch_dwpw<2, 2>(cfg, a, lds, s);
    case 41: return launch_dwpw<4, 1>(cfg, a, lds, s);
    case 42: return launch_dwpw<4, 2>(cfg, a, lds, s);
    case 51: return launch_dwpw<5, 1>(cfg, a, lds, s);
    case 52: return launch_dwpw<5, 2>(cfg, a, lds, s);
    case 81: return launch_dwpw2<8, 1, false>(cfg, a, lds, s);  // (128 pointwise channels: no fused tail)
    case 82: return launch_dwpw2<8, 2, false>(cfg, a, lds, s);
  }
  return fail(-3, "dwpw: unsupported tile shape");
}

template <int MT, int NT, bool PW2 = true>
static int launch_dwpw_group(const ConvGroup& G, int grid, size_t lds, hipStream_t s) {
  auto kern = dwpw_group_kernel<MT, NT, PW2>;
  if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), (int)((int)LDS_MAX))) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, G);
  YL_HIP(hipGetLastError());
  return 0;
}

// n = 2 or 3 depthwise + pointwise + 1x1 launches of one tiling (five channel tiles, one NT) in one launch
int dwpw_group_launch(const ConvConfig* cfg0, const ConvArgs* args, int n, int batch, hipStream_t s) {
  if (n < 2 || n > 3) return fail(-3, "dwpw group: two or three members");
  ConvGroup G{};
  size_t lds = 0;
  int grid = 0, nt0 = 0;
  for (int i = 0; i < n; ++i) {
    ConvConfig cfg = cfg0[i];
    G.a[i] = args[i];
    size_t l = 0;
    if (int rc = dwpw_prepare(cfg, G.a[i], batch, &l)) return rc;  // (may retile: YL_DWPW_NT)
    if (i == 0) nt0 = cfg.NT;
    if (cfg.MT != 5 || cfg.NT != nt0 || (G.a[i].wpk2 != nullptr) != (G.a[0].wpk2 != nullptr))
      return fail(-3, "dwpw group: members need one tiling and all of them, or none, a fused tail");
    G.nb[i] = G.a[i].nblocks;
    grid += G.nb[i];
    lds = l > lds ? l : lds;
  }
  if (!G.a[0].wpk2) return nt0 == 1 ? launch_dwpw_group<5, 1, false>(G, grid, lds, s) : launch_dwpw_group<5, 2, false>(G, grid, lds, s);
  return nt0 == 1 ? launch_dwpw_group<5, 1>(G, grid, lds, s) : launch_dwpw_group<5, 2>(G, grid, lds, s);
}

// conv_kernel's arguments derived from the tiling (everything conv_launch used to fill in place)
static int conv_prepare(const ConvConfig& cfg, ConvArgs& a, int batch) {
  a.ncb = cfg.ncb;
  a.cg = cfg.cg;
  a.ksteps = cfg.ksteps;
  a.kchunk = cfg.kchunk;
  a.slab2 = cfg.slab2;
  a.ntaps = cfg.ntaps;
  a.kord = cfg.kord;
  a.kkn = cfg.cg >> 2;
  a.magic_kkn = magic_of(cfg.cg >> 2);
  a.magic_cg = magic_of(cfg.cg);
  a.magic_wo = magic_of(a.Wo);
  a.magic_ho = magic_of(a.Ho);
  a.npatches = cfg.npatches;
  a.ppw = cfg.ppw;
  a.npix = batch * a.Ho * a.Wo;
  {
    const long long lim = 1LL << 30;
    const long long inb = (long long)(cfg.cg - (a.in2 ? a.ks_split * 4 : 0)) * a.in_plane;
    const int cout_out = a.wpk2 ? a.post_cout : a.Cout;  // channels of the tensor that is actually stored
    const long long outb = a.out_f32 ? (long long)a.npix * a.out_pitch * 4 : (long long)((cout_out + 7) / 8) * a.out_plane;
    const long long resb = a.res ? (long long)((cout_out + 7) / 8) * a.res_plane : 0;
    if (inb > lim || outb > lim || resb > lim)
      return fail(-3, "conv: a tensor slice exceeds 1 GiB (32-bit buffer offsets); lower the batch size");
    if (a.in2) {
      const long long in2b = (long long)(a.ks_split * 4) * a.in2_plane;
      if (in2b > lim) return fail(-3, "conv: a tensor slice exceeds 1 GiB (32-bit buffer offsets); lower the batch size");
      a.in2_bytes = (unsigned)in2b;
    }
    a.in_bytes = (unsigned)inb;
    a.out_bytes = a.out8 && a.out8_only ? 0u : (unsigned)outb;  // (copy only: a descriptor of zero bytes drops every bf16 store)
    a.res_bytes = (unsigned)resb;
    a.st_nt = a.out_f32 ? 0 : st_nt_policy(outb);
  }
  a.nblocks = cfg.gws * cfg.ncb;
  {
    static const int dbg = tune_int("YL_CONV_DBG", 0);  // ablation, dev only
    a.dbg = dbg;
  }
  return 0;
}

// `a` arrives with pointers, plane sizes, geometry, Cin/Cout/stride/act/out_f32 filled in.
int conv_launch(const ConvConfig& cfg, ConvArgs a, int batch, hipStream_t s) {
  // The register-filter variants (conv3_reg, conv3_cs) are chosen from (Cin, Cout, k, stride, post) alone and take only a plain stride-1
  // conv between bf16 planar tensors of ONE geometry; they share conv_kernel's filter packing, so any other view (fp32 or
  // pitched output, an upsampled second input, an e4m3 copy) simply runs on conv_kernel.
  const bool plain = a.stride == 1 && !a.out_f32 && !a.in2 && !a.wpk2 && !a.out8 && a.in_w1 == a.out_w1 &&
                     a.in_img == a.out_img && a.in_g0 == a.out_g0 && a.in_w1 == a.Wo + 1;
  if (cfg.gemm) return conv_gemm_launch(cfg, a, batch, s);  // chosen by the engine with the views known: no fallback needed
  if (cfg.cs == 1 && plain) return conv3_cs_launch(cfg, a, batch, s);
  if (cfg.cs == 2 && a.wpk2 && a.dfl_out && a.stride == 1 && !a.out_f32 && !a.in2 && !a.out8 && a.in_w1 == a.out_w1 &&
      a.in_img == a.out_img && a.in_g0 == a.out_g0 && a.in_w1 == a.Wo + 1)
    return conv3_cs_launch(cfg, a, batch, s);
  if (cfg.reg && plain) return conv3_reg_launch(cfg, a, batch, s);
  if (int rc = conv_prepare(cfg, a, batch)) return rc;
  return cfg.ntaps == 9 ? launch_m<3>(cfg, a, s) : launch_m<1>(cfg, a, s);
}

// The tiling of a conv that joins a group launch: plain conv_kernel (no register / tile variants), NT forced to the
// group's common pixel-tile count, one patch per wave.
void conv_retile_for_group(ConvConfig* c, int NT, int batch, int Ho, int Wo) {
  c->reg = c->cs = 0;
  c->NT = NT;
  c->npatches = (int)(((long long)batch * Ho * Wo + 16 * NT - 1) / (16 * NT));
  c->ppw = 1;
  c->gws = (c->npatches + 3) / 4;
  c->name = c->ntaps == 9 ? "conv3x3_mfma" : "conv1x1_mfma";
}

bool conv_group_compatible(const ConvConfig* cfg, const ConvArgs* a, int n) {
  if (n < 2 || n > 3) return false;
  for (int i = 0; i < n; ++i) {
    if (cfg[i].ntaps != 9 || cfg[i].MT != 4 || cfg[i].uniform != cfg[0].uniform || cfg[i].post != cfg[0].post || cfg[i].NT != cfg[0].NT ||
        (cfg[i].NT != 2 && cfg[i].NT != 4) || a[i].in2 || a[i].out8 || a[i].out_f32 || a[i].res || a[i].stride != 1 ||
        (a[i].dfl_out != nullptr) != (a[0].dfl_out != nullptr) || (a[i].dfl_out && (!cfg[i].post || a[i].post_cout != 64)) ||
        cfg[i].reg || cfg[i].cs)
      return false;
  }
  return true;
}

template <int NT, bool POST, bool U = true, bool DFL = false>
static int launch_group_r(const ConvGroup& G, int grid, size_t lds, hipStream_t s) {
  if constexpr (POST && !DFL) {
    if (G.a[0].dfl_out) return launch_group_r<NT, POST, U, true>(G, grid, lds, s);  // (all members or none: conv_group_compatible)
  }
  if (G.a[0].dfl_out && !DFL) return fail(-3, "conv group: the DFL epilogue rides on the fused follow-up 1x1");
  auto kern = conv_group_kernel<3, 4, NT, U, POST, DFL>;
  if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), (int)((int)LDS_MAX))) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, G);
  YL_HIP(hipGetLastError());
  return 0;
}

// n = 2 or 3 stride-1 3x3 convs with 64 output channels (conv_group_compatible) in one launch
int conv_group_launch(const ConvConfig* cfg, const ConvArgs* args, int n, int batch, hipStream_t s) {
  if (!conv_group_compatible(cfg, args, n)) return fail(-3, "conv group: members need one tiling (3x3, stride 1, 64 output channels)");
  ConvGroup G{};
  size_t lds = 0;
  int grid = 0;
  for (int i = 0; i < n; ++i) {
    G.a[i] = args[i];
    if (int rc = conv_prepare(cfg[i], G.a[i], batch)) return rc;
    G.nb[i] = G.a[i].nblocks;
    grid += G.nb[i];
    lds = cfg[i].lds_bytes > lds ? cfg[i].lds_bytes : lds;
  }
  const bool post = cfg[0].post != 0;
  if (!cfg[0].uniform) {  // (plane-major K order or a channel count that is not a multiple of 32)
    if (cfg[0].NT == 4) return post ? launch_group_r<4, true, false>(G, grid, lds, s) : launch_group_r<4, false, false>(G, grid, lds, s);
    return post ? launch_group_r<2, true, false>(G, grid, lds, s) : launch_group_r<2, false, false>(G, grid, lds, s);
  }
  if (cfg[0].NT == 4) return post ? launch_group_r<4, true>(G, grid, lds, s) : launch_group_r<4, false>(G, grid, lds, s);
  return post ? launch_group_r<2, true>(G, grid, lds, s) : launch_group_r<2, false>(G, grid, lds, s);
}

template <int NTC, bool POST, int NTD, bool DFL = false>
static int launch_head_group(const ConvGroup& GC, const ConvGroup& GD, int n_conv, int grid, size_t lds, hipStream_t s) {
  if constexpr (POST && !DFL) {
    if (GC.a[0].dfl_out) return launch_head_group<NTC, POST, NTD, true>(GC, GD, n_conv, grid, lds, s);
  }
  if (GC.a[0].dfl_out && !DFL) return fail(-3, "head group: the DFL epilogue rides on the fused follow-up 1x1");
  auto kern = head_group_kernel<NTC, POST, NTD, POST, DFL>;
  if (int rc = allow_big_lds(reinterpret_cast<const void*>(kern), (int)((int)LDS_MAX))) return rc;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, GC, GD, n_conv);
  YL_HIP(hipGetLastError());
  return 0;
}

// nc (1..3) compatible convs and nd (1..3) depthwise + pointwise launches of one tiling in ONE launch (head_group_kernel).
// The convs all carry a fused 1x1 and the depthwise launches a fused tail, or none of them does.
int head_group_launch(const ConvConfig* ccfg, const ConvArgs* cargs, int nc, const ConvConfig* dcfg0, const ConvArgs* dargs, int nd,
                      int batch, hipStream_t s) {
  if (nc < 1 || nc > 3 || nd < 1 || nd > 3) return fail(-3, "head group: one to three members of each kind");
  if (nc >= 2 && !conv_group_compatible(ccfg, cargs, nc)) return fail(-3, "head group: the convs need one tiling");
  ConvGroup GC{}, GD{};
  size_t lds = 0;
  int n_conv = 0, grid = 0;
  for (int i = 0; i < nc; ++i) {
    GC.a[i] = cargs[i];
    if (int rc = conv_prepare(ccfg[i], GC.a[i], batch)) return rc;
    GC.nb[i] = GC.a[i].nblocks;
    n_conv += GC.nb[i];
    lds = ccfg[i].lds_bytes > lds ? ccfg[i].lds_bytes : lds;
  }
  const bool post = ccfg[0].post != 0;
  int ntd0 = 0;
  for (int i = 0; i < nd; ++i) {
    ConvConfig cfg = dcfg0[i];
    GD.a[i] = dargs[i];
    size_t l = 0;
    if (int rc = dwpw_prepare(cfg, GD.a[i], batch, &l)) return rc;
    if (i == 0) ntd0 = cfg.NT;
    if (cfg.MT != 5 || cfg.NT != ntd0 || (GD.a[i].wpk2 != nullptr) != post)
      return fail(-3, "head group: the depthwise members need one tiling, with a fused tail exactly when the convs carry a fused 1x1");
    GD.nb[i] = GD.a[i].nblocks;
    grid += GD.nb[i];
    lds = l > lds ? l : lds;
  }
  grid += n_conv;
  const int ntc = ccfg[0].NT, ntd = ntd0;
  if (ccfg[0].ntaps != 9 || ccfg[0].MT != 4 || !ccfg[0].uniform || (ntc != 2 && ntc != 4) || (ntd != 1 && ntd != 2))
    return fail(-3, "head group: unsupported tiling");
#define YL_HG(NTC, NTD) \
  (post ? launch_head_group<NTC, true, NTD>(GC, GD, n_conv, grid, lds, s) : launch_head_group<NTC, false, NTD>(GC, GD, n_conv, grid, lds, s))
  if (ntc == 4) return ntd == 2 ? YL_HG(4, 2) : YL_HG(4, 1);
  return ntd == 2 ? YL_HG(2, 2) : YL_HG(2, 1);
#undef YL_HG
}

}  // namespace yl
