// Two-blocks-per-CU implicit-GEMM convolution for the wide bf16 layers: forward and stride-1 data gradient of nn.Conv2d inside Conv /
// Bottleneck / Detect (ultralytics/nn/modules/conv.py:38-55, block.py:553-565, head.py:40-46).
//
// conv_v4.hip (one 8-wave block per CU, two wave groups one barrier apart) reaches the MFMA rate of the CDNA4 playbook's best GEMM
// loop, but a conv layer is a short GEMM: 18-36 K-steps per tile, then 64-128 KiB of output.  tools/v4_diag measured what that costs
// with one block per CU: all 256 CUs reach their epilogues together, the store burst (33 MB) runs at the HBM write rate while every
// MFMA pipe idles (12-15 us of a 70 us tile), then every block pays its address set-up and first DMA round trip, and 400 tiles on 256
// CUs leave the second round half empty.  This kernel keeps the per-wave work of v4 (128 x 64 wave tile, v_mfma_f32_16x16x32_bf16,
// transposed product so that a lane holds 4 consecutive output channels, LDS-DMA with out-of-range offsets for padding) and changes
// the block shape so that TWO blocks share a CU:
//   * 256 x 128 tile on 4 waves (2 x 2), K-step 32: 24 KiB per stage, three stages = 72 KiB per block, 256 VGPRs per wave;
//   * one barrier per K-step: wait (counted vmcnt: the next stage stays in flight) -> barrier -> refill the stage two steps ahead ->
//     12 x ds_read_b128 + 32 MFMAs;
//   * the two blocks of a CU are independent: one block's epilogue, set-up and barrier waits overlap the other block's K-loop, and
//     tiles are handed out at twice the granularity (the tail round costs half as much).
// LDS rows are 64 bytes (32 bf16); 16-byte slot s of row r holds chunk s ^ (((r >> 2) & 2) ? 3 : 0): the four 16-lane groups of a
// ds_read_b128 (16x16x32 operand: lane = row l&15, chunk l>>4) then hit 16 different slots of the 256-byte bank row.
#include <stdlib.h>
#include <type_traits>
#include "dy_common.h"
#include "conv_epilogue.h"
#include "../../include/dedark_yolo.h"

namespace v5 {

constexpr int BM = 256, BK = 32;
constexpr int A_BYTES = BM * 64, NSTAGE = 3;
// bit 31 of an activation offset marks a padded lane: beyond any source extent (checked by the dispatcher: < 2 GiB)
constexpr unsigned B_ROW_OOB = 0x40000000u;         // weight extent <= 1 GiB: row-invalid + any k offset stays out of range

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__device__ unsigned long long g_stamps[16];       // diagnostics (DY_ABLATE & 32): s_memtime (100 MHz) of wave 0, [0..7] first block, [8..15] a late one

__device__ inline void stamp(int ablate, int nblk, int i) {
  if ((ablate & 32) && threadIdx.x == 0 && (blockIdx.x == 0 || (int)blockIdx.x == (nblk / 8) * 7)) {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    g_stamps[(blockIdx.x == 0 ? 0 : 8) + i] = t;
  }
}

struct P {
  const char* src;
  const char* w;
  char* dst;
  unsigned src_bytes, w_bytes;
  long src_ld, dst_ld, dst_row, dst_img;
  int Hs, Ws, Cs, Hd, Wd, Cd;
  int stride;
  int KH, KW;
  int dh0, dhs, dw0, dws;        // window tap (th, tw) reads source pixel (oh*stride + dh0 + dhs*th, ow*stride + dw0 + dws*tw)
  int kh0, khs, kw0, kws, KWf;   // ... and weight tap (kh0 + khs*th, kw0 + kws*tw) of a KHf x KWf pack
  long w_row;                    // elements per output-channel row of the weight pack
  const float* scale;
  const float* shift;
  int act;
  double* stats;
  int accumulate;
  long M;
  int a_min;                     // most negative window-tap byte offset (<= 0): folded into the activation descriptor's base so that
                                 // the per-K-step SGPR offsets of conv_kernel are >= 0 (see conv_v4.hip)
  int nk;                        // K-steps = KH*KW*Cs/32
  int tiles_n, nblk;
  const char* add_src;           // optional addend view of a data gradient (dy_conv_desc.add_src)
  long add_src_ld;
  int ablate;                    // DY_ABLATE (make DIAG=1 only): 1 A-operand DMA of taps != 0 out of range (no fetch, zeros land),
                                 // 2 every A DMA out of range, 4 every B DMA out of range, 8 no MFMA, 16 no fragment reads,
                                 // 64 taps != 0 issue NO A DMA at all (wait switches to vmcnt(0))
  // > 1: the parity classes of a stride-2 data gradient in ONE conv_kernel launch (see conv_v4.hip: cls[c].blk0 = first slot of class c in
  // every XCD's block sequence, cls[c]._r = its slots per XCD)
  int ncls;
  DyParityCls cls[4];
};

__device__ inline int xcd_remap(int bid, int nblk) {
  int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

__device__ inline long dst_offset(const P& p, long m) {
  if (p.dst_row == 0) return m * p.dst_ld;
  const long HWd = (long)p.Hd * p.Wd;
  const long img = m / HWd;
  const int rem = (int)(m - img * HWd);
  const int oh = rem / p.Wd, ow = rem - oh * p.Wd;
  return img * p.dst_img + (long)oh * p.dst_row + (long)ow * p.dst_ld;
}

// Shared tail of the kernels of this file: affine + activation on the accumulators, bf16 / f16 image in LDS, coalesced stores, optional
// per-channel statistics for BatchNorm (f64 atomics into the tile's replica).
template <int BN, typename T, int MB>
__device__ __forceinline__ void epilogue(const P& p, f32x4 (&acc)[MB][4], char* smem, int tid, int lane, int wave, int wm, int wn, long m0,
                                         int n0, int tile_m) {
  constexpr int WN = BN / 64, WM = 4 / WN;
  // ---- epilogue: bf16 image [pixel][channel] (conv_epilogue.h: store_rows) -> 16-byte stores, 256 contiguous bytes per quarter-wave
  constexpr int PT = dy_epi::row_pitch<BN>();
  const int cl = lane & 15, g = lane >> 4;
  const bool plain = !p.scale && !p.shift && p.act == DY_ACT_NONE;   // raw output: training forward (BatchNorm follows), data gradients
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c0 = 64 * wn + 16 * j + 4 * g;               // this lane's 4 channels of the block
    if (plain) {
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const int px = 16 * MB * wm + 16 * i + cl;
        uint2 w2 = {dy_epi::pack2<T>(acc[i][j][0], acc[i][j][1]), dy_epi::pack2<T>(acc[i][j][2], acc[i][j][3])};
        *reinterpret_cast<uint2*>(smem + px * PT + c0 * 2) = w2;
      }
    } else {
      float sc[4], sf[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + c0 + e;
        const bool nok = n < p.Cd;
        sc[e] = (nok && p.scale) ? p.scale[n] : 1.f;
        sf[e] = (nok && p.shift) ? p.shift[n] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const int px = 16 * MB * wm + 16 * i + cl;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float u = acc[i][j][e] * sc[e] + sf[e];
          if (p.act == DY_ACT_SILU) u = u * dy_sigmoid(u);
          else if (p.act == DY_ACT_LEAKY) u = u > 0.f ? u : 0.1f * u;
          v[e] = u;
        }
        uint2 w2 = {dy_epi::pack2<T>(v[0], v[1]), dy_epi::pack2<T>(v[2], v[3])};
        *reinterpret_cast<uint2*>(smem + px * PT + c0 * 2) = w2;
      }
    }
  }
  __syncthreads();
  stamp(DY_ABLATE_OF(p), p.nblk, 3);
  dy_epi::store_rows<BM, BN, 4>(smem, lane, wave, m0, n0, p.M, p.Cd, p.accumulate, reinterpret_cast<T*>(p.dst),
                                [&](long m) { return dst_offset(p, m); }, reinterpret_cast<const T*>(p.add_src), p.add_src_ld);
  stamp(DY_ABLATE_OF(p), p.nblk, 4);
  if (p.stats) {
    // per-channel sum / sum of squares of the f32 accumulators.  Rows beyond M and channels beyond Cd were fed zeros by the DMA (out
    // of range offsets), so their accumulators are exactly 0 and need no predicate.
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);          // [WM][BN][2]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
          const float a = acc[i][j][e];
          s1 += a;
          s2 += a * a;
        }
        s1 = row16_sum(s1);                                  // over the 16 pixels on the lanes of a row group
        s2 = row16_sum(s2);
        if (cl == 0) {
          const int col = 64 * wn + 16 * j + 4 * g + e;
          red[(wm * BN + col) * 2] = s1;
          red[(wm * BN + col) * 2 + 1] = s2;
        }
      }
    __syncthreads();
    if (tid < BN) {
      const int n = n0 + tid;
      if (n < p.Cd) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          s1 += red[(w * BN + tid) * 2];
          s2 += red[(w * BN + tid) * 2 + 1];
        }
        double* st = p.stats + (long)(tile_m % DY_STATS_REPLICAS) * 2 * p.Cd;
        atomic_add_f64(st + n, (double)s1);
        atomic_add_f64(st + p.Cd + n, (double)s2);
      }
    }
  }
}

// BN = 128: waves 2 (M) x 2 (N), wave tile 128 x 64.  BN = 64 (the 64-channel layers): waves 4 x 1, wave tile 64 x 64.
// SLIM (BN = 64 only): a two-stage ring of single K-steps (40 KiB) and FOUR co-resident blocks per CU instead of four stages in pairs
// (80 KiB) and two blocks: the 64-wide tiles of the high-resolution layers are 0.5 us of MFMA work behind ~10 us of fixed per-tile cost
template <int BN, typename T = bf16_t, bool SLIM = false>
__global__ __launch_bounds__(256, SLIM ? 4 : 2) void conv_kernel(const P pk) {
  constexpr int WN = BN / 64, WM = 4 / WN, MB = BM / WM / 16, NB = 4, STAGE = A_BYTES + BN * 64, B_LD = BN / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  P p = pk;
  int bid;
  if (pk.ncls > 1) {                 // several problems in one launch: this block's class replaces the launch-wide geometry (conv_v4.hip)
    const int x = blockIdx.x & 7, i = (int)blockIdx.x >> 3;
    int c = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (k < pk.ncls && i >= pk.cls[k].blk0) c = k;
    const DyParityCls& k = pk.cls[c];
    bid = x * k._r + (i - k.blk0);
    if (bid >= (int)((k.M + BM - 1) / BM) * pk.tiles_n) return;        // (block-uniform: up to 7 surplus blocks per class)
    p.dst = k.dst; p.M = k.M; p.Hd = k.Hd; p.Wd = k.Wd; p.KH = k.KH; p.KW = k.KW;
    p.dh0 = -k.pad; p.dw0 = -k.pad; p.kh0 = k.kh0; p.kw0 = k.kw0;
    p.nk = k.Ktot / BK;
    p.a_min = -(k.pad * pk.Ws + k.pad) * (int)pk.src_ld * 2;
  } else {
    bid = xcd_remap(blockIdx.x, p.nblk);
  }
  stamp(DY_ABLATE_OF(p), p.nblk, 0);
  const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
  const long m0 = (long)tile_m * BM;
  const int n0 = tile_n * BN;

  // records of the two descriptors: 0 behind the last K-step (every lane out of range: zeros land in a stage nobody reads again)
  unsigned nrec_a = p.nk > 0 ? p.src_bytes - p.a_min : 0, nrec_b = p.nk > 0 ? p.w_bytes : 0;

  // ---- DMA bookkeeping: wave instruction idx = wave + 4j fills rows 16*idx .. +15 of the A (j < 4) / B (j < 2) tile.  As in conv_v4.hip:
  // a lane's pixel offset is the instruction's VGPR offset for the whole K loop, tap and channel chunk travel in its SGPR offset, the
  // padding bits of the lane's row are rotated once per K-step so that the current tap's bit is bit 31 (= out of range when set)
  const int lrow = lane >> 2, slot = lane & 3;
  const int chunk = slot ^ ((lane & 32) ? 3 : 0);          // logical 16-byte chunk (8 channels of the 32-deep step) this lane fetches
  unsigned a_off[4], a_pad[4], b_off[B_LD];
  {
    const DyTileWalk walk(m0, p.Hd, p.Wd);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 16 * (wave + 4 * j) + lrow;
      const bool ok = m0 + r < p.M;
      int img, oh, ow;
      walk.at(r, img, oh, ow);
      const int sh0 = oh * p.stride, sw0 = ow * p.stride;
      a_off[j] = (unsigned)((((long)img * p.Hs + sh0) * p.Ws + sw0) * p.src_ld * 2 + chunk * 16);
      unsigned wb = 0, mk = 0;                             // bit th * KW + tw: window tap (th, tw) of this row lies inside the image
      for (int tw = 0; tw < p.KW; ++tw) {
        const int sw = sw0 + p.dw0 + p.dws * tw;
        if (sw >= 0 && sw < p.Ws) wb |= 1u << tw;
      }
      for (int th = 0; th < p.KH; ++th) {
        const int sh = sh0 + p.dh0 + p.dhs * th;
        if (ok && sh >= 0 && sh < p.Hs) mk |= wb << (th * p.KW);
      }
      if (DY_ABLATE_OF(p) & 2) mk = 0;
      else if (DY_ABLATE_OF(p) & 1) mk &= 1u;
      a_pad[j] = __builtin_amdgcn_alignbit(~mk, ~mk, 1);
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      const int n = n0 + 16 * (wave + 4 * j) + lrow;
      b_off[j] = (n < p.Cd && !(DY_ABLATE_OF(p) & 4)) ? (unsigned)((long)n * p.w_row * 2 + chunk * 16) : B_ROW_OOB;
    }
  }
  // K-step being issued (wave-uniform).  K order: all taps of one BK-channel chunk, then the next chunk.  The taps re-read the SAME
  // pixels (shifted windows), so with the taps innermost a tile's live set is (tile + halo) x BK channels (~40 KB; 32 CUs x 40 KB sit
  // in an XCD's 4 MiB L2) and the 2nd .. 9th reads are L2 hits; tap-major order cycled through the tile's full channel depth between
  // two reads of a line (150 KB per CU: more than its L2 share -> every tap went back to the Infinity Cache, 2.4x fabric traffic).
  const int ntaps = p.KH * p.KW;
  int vtap_a, vtap_b;                                      // lane t = byte offsets of tap t
  {
    const int t = lane < ntaps ? lane : 0;
    const int th = t / p.KW, tw = t - th * p.KW;
    vtap_a = ((p.dh0 + p.dhs * th) * p.Ws + p.dw0 + p.dws * tw) * (int)p.src_ld * 2 - p.a_min;      // >= 0
    vtap_b = (int)((((long)((p.kh0 + p.khs * th) * p.KWf + p.kw0 + p.kws * tw)) * p.Cs) * 2);
  }
  int sk = 0, s_tap = 0, s_ci2 = 0;
  int a_koff = __builtin_amdgcn_readlane(vtap_a, 0), b_koff = __builtin_amdgcn_readlane(vtap_b, 0);
  auto advance = [&]() {
    ++sk;
    const bool wrap = s_tap + 1 == ntaps;
    s_tap = wrap ? 0 : s_tap + 1;
    s_ci2 += wrap ? 2 * BK : 0;
    a_koff = __builtin_amdgcn_readlane(vtap_a, s_tap) + s_ci2;
    b_koff = __builtin_amdgcn_readlane(vtap_b, s_tap) + s_ci2;
    const int rot = wrap ? (33 - ntaps) & 31 : 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) a_pad[j] = __builtin_amdgcn_alignbit(a_pad[j], a_pad[j], rot);
    if (sk >= p.nk) { nrec_a = 0; nrec_b = 0; a_koff = 0; b_koff = 0; }
  };
  auto issue = [&](int base) {                 // one stage: 4 + 2 DMA instructions per wave, ONE unconditional load per lane each
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.src + p.a_min), 0, nrec_a, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, nrec_b, 0x00020000);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if ((DY_ABLATE_OF(p) & 64) && s_tap) continue;
      const unsigned v = (a_pad[j] & 0x80000000u) | a_off[j];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(smem + base + (wave + 4 * j) * 1024), 16, (int)v, a_koff, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (lds_ptr_t)(smem + base + A_BYTES + (wave + 4 * j) * 1024), 16, (int)b_off[j], b_koff, 0,
                                               0);
  };

  // ---- fragment read addresses (16x16x32 operand: lane = row l&15, chunk l>>4)
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fq ^ ((fr & 8) ? 3 : 0)) * 16;
  const int a_rd = (16 * MB * wm + fr) * 64 + sw;        // + 1024 * m-block
  const int b_rd = A_BYTES + (64 * wn + fr) * 64 + sw;   // + 1024 * n-block

  f32x4 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // one staged K-step: 4 + MB fragment reads, MB x 4 MFMAs; `mid` (the K-walk bookkeeping of the NEXT issue) runs inside the cluster
  auto compute = [&](int base, auto mid) {
    u32x4 bfr[NB], afr[MB];
    if (DY_ABLATE_OF(p) & 16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = u32x4{0x3f803f80u + (unsigned)base, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
      for (int i = 0; i < MB; ++i) afr[i] = u32x4{0x3f803f80u, 0x3f803f80u + (unsigned)lane, 0x3f803f80u, 0x3f803f80u};
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const u32x4*>(smem + base + b_rd + 1024 * j);
#pragma unroll
      for (int i = 0; i < MB; ++i) afr[i] = *reinterpret_cast<const u32x4*>(smem + base + a_rd + 1024 * i);
    }
    if (DY_ABLATE_OF(p) & 8) {
#pragma unroll
      for (int i = 0; i < MB; ++i) acc[i][0][0] += __builtin_bit_cast(float, afr[i][0] ^ bfr[i & 3][1]);
      mid();
      return;
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MB; ++i) {
#pragma unroll
      for (int j = 0; j < NB; ++j)
        // transposed product (rows = output channels, columns = pixels): a lane ends up with 4 consecutive CHANNELS of one pixel
        acc[i][j] = mfma_16x16x32<T>(bfr[j], afr[i], acc[i][j]);
      if (i == 0) mid();
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto nothing = []() {};
  auto walk = [&]() { advance(); };
  stamp(DY_ABLATE_OF(p), p.nblk, 1);
  if constexpr (BN >= 128) {
    issue(0);
    advance();
    issue(STAGE);
    advance();
    int cur = 0, fill = 2 * STAGE;
    // (K-steps behind the last one are not issued: a dead DMA fetches nothing but takes the full round trip to retire, and the wait in
    //  front of the epilogue -- which reuses the ring -- paid for it once per tile; the last step's wait covers everything in flight)
    for (int kt = 0; kt < p.nk; ++kt) {
      if ((DY_ABLATE_OF(p) & 64) || kt + 1 >= p.nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + B_LD) : "memory");    // everything but the youngest stage: K-step kt has landed
      __builtin_amdgcn_s_barrier();                        // ... for every wave; and every wave is done reading K-step kt - 1
      __builtin_amdgcn_sched_barrier(0);
      if (sk < p.nk) issue(fill);                          // K-step kt + 2 into the stage K-step kt - 1 occupied
      __builtin_amdgcn_sched_barrier(0);
      compute(cur, walk);
      __builtin_amdgcn_sched_barrier(0);
      cur = cur == 2 * STAGE ? 0 : cur + STAGE;
      fill = fill == 2 * STAGE ? 0 : fill + STAGE;
    }
  } else if constexpr (SLIM) {
    issue(0);
    advance();
    int cur = 0;
    for (int kt = 0; kt < p.nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // K-step kt has landed (nothing younger is in flight)
      __builtin_amdgcn_s_barrier();                        // ... for every wave; and every wave is done reading K-step kt - 1
      __builtin_amdgcn_sched_barrier(0);
      if (sk < p.nk) issue(cur ? 0 : STAGE);               // K-step kt + 1 into the other stage
      __builtin_amdgcn_sched_barrier(0);
      compute(cur, walk);
      __builtin_amdgcn_sched_barrier(0);
      cur = cur ? 0 : STAGE;
    }
  } else {
    // 64-wide tiles: a wave has only 16 MFMAs per K-step, less than the fixed cost of a step (wait + barrier + DMA issue), so the
    // steps go in PAIRS: four stages, one wait + barrier per two K-steps (32 MFMAs), the next pair in flight meanwhile.  An odd
    // last step computes on a zero-filled stage.
    issue(0);
    advance();
    issue(STAGE);
    advance();
    int cur = 0;
    for (int kt = 0; kt < p.nk; kt += 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this pair has landed (nothing younger is in flight)
      __builtin_amdgcn_s_barrier();                        // ... for every wave; and every wave is done reading the previous pair
      __builtin_amdgcn_sched_barrier(0);
      const int nxt = cur ^ (2 * STAGE);
      if (sk < p.nk) {
        issue(nxt);                                        // next pair into the two stages the previous pair occupied
        advance();
        issue(nxt + STAGE);                                // (the second step of an odd tail: zero fill, computed on)
      } else {
        advance();
      }
      __builtin_amdgcn_sched_barrier(0);
      compute(cur, walk);
      compute(cur + STAGE, nothing);
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (nothing is in flight any more unless the tile has a single K-step)
  __builtin_amdgcn_s_barrier();
  stamp(DY_ABLATE_OF(p), p.nblk, 2);

  epilogue<BN, T, MB>(p, acc, smem, tid, lane, wave, wm, wn, m0, n0, tile_m);
#ifdef DY_DIAG
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (stamp 5 = stores acknowledged; the shipped kernel ends with them in flight)
#endif
  stamp(DY_ABLATE_OF(p), p.nblk, 5);
}

// ---- 3x3 / stride 1 / pad 1 ("same") layers: activation BAND kernel.
// tools/conv_bench stamps (DY_ABLATE=32, make DIAG=1) showed what bounds conv_kernel's K loop: not the MFMA pipe, the LDS reads or the
// fetches, but the LDS-DMA round trip.  A stage needs ~3,000 cycles from issue to landed even when every lane is out of range, ~4,600
// under load, and with two of three stages in flight a K-step cannot take less than half of that: 2,300 cycles per step against
// 1,024 cycles of MFMA work of the CU's two blocks.  Bytes in flight are bounded by LDS, so the way to more K-steps in flight is fewer
// bytes per step: the three kw taps of one kernel row read the SAME pixels shifted by one, so a band of 258 (272) consecutive
// flattened pixels x 32 channels serves three K-steps (17 KiB instead of 3 x 16 KiB), the weight tiles (8 / 4 KiB per step) get their
// own deeper ring, and the DMA addresses need no (n, h, w) arithmetic at all: band row i is pixel m0 - 1 + i + dh * W of the flattened
// [N*H*W] source, anything outside the buffer lands as zeros.  What the flattening gets wrong -- taps that cross the left / right /
// top / bottom border of an image read a real neighbour pixel -- is repaired at the fragment read: a lane whose (pixel, tap) is
// padding reads a zero row instead (one v_cndmask on the address per fragment, masks precomputed per lane: 9 taps x MB pixels).
//   BN = 128: 2 bands + 4 weight tiles (66 KiB), weights 3 steps ahead, two blocks per CU.  BN = 64: the same rings (51 KiB) and THREE
//   co-resident blocks per CU -- a 256 x 64 tile is 0.5 us of MFMA work behind ~10 us of row decode, first DMA round trip and epilogue,
//   so a third block in flight is worth more than a deeper ring (3 bands + 7 weight tiles, 79 KiB, 6 steps ahead, two blocks per CU:
//   64->64 at 160x160 217 / 180 us forward / data gradient against 200 / 167; DY_BAND64_NBAND=3 in a DIAG build selects it).
//   Step t = 3g + tw issues [band g + NBAND - 1 if tw == 0] + weight tile t + DB, DB = 3 (NBAND - 1); its wait leaves exactly the
//   issues of steps t - DB + 1 .. t - 1 in flight (counted vmcnt, a constant per tw).
constexpr int BAND_BYTES = 17 * 1024;               // 272 rows of 64 bytes

template <int BN, typename T = bf16_t, int NBAND_ = (BN == 64 ? 3 : 2)>
__global__ __launch_bounds__(256, (BN == 64 && NBAND_ == 2) ? 3 : 2) void band_kernel(const P p) {
  constexpr int WN = BN / 64, WM = 4 / WN, MB = BM / WM / 16, B_LD = BN / 64, B_BYTES = BN * 64;
  constexpr int NBAND = NBAND_, DB = 3 * (NBAND - 1), NBBUF = DB + 1;
  constexpr int OFF_B = NBAND * BAND_BYTES, OFF_ZERO = OFF_B + NBBUF * B_BYTES;
  constexpr int N0 = (DB - 1) * B_LD + 5 * (NBAND - 2), N1 = (DB - 1) * B_LD + 5 * (NBAND - 1);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int bid = xcd_remap(blockIdx.x, p.nblk);
  stamp(DY_ABLATE_OF(p), p.nblk, 0);
  const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
  const long m0 = (long)tile_m * BM;
  const int n0 = tile_n * BN;

  if (tid < 16) reinterpret_cast<unsigned*>(smem + OFF_ZERO)[tid] = 0u;      // the zero row (visible after the first barrier)

  // ---- DMA bookkeeping: wave instruction idx = wave + 4j fills band rows 16*idx .. +15 (j < 4) / weight rows (j < B_LD); the 17th
  // band block (rows 256 .. 271, of which 256 and 257 are read) is filled by all four waves, 16 lanes each
  // (band rows are read at three alignments -- row r + 0 / 1 / 2 for the three column taps -- so the band kernel keys the slot swizzle
  // on bit 2 of the row, chunk c of row r in slot c ^ ((r & 4) ? 2 : 0): the four rows r, r+4, r+8, r+12 of a ds_read_b128 lane group that
  // share a quarter of the 256-byte bank row then sit in four different slots for EVERY starting row, where conv_kernel's key on bit 3
  // is conflict-free only for 16-aligned reads: 2-way on the shifted taps, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.33 / 0.23)
  const int lrow = lane >> 2, slot = lane & 3;
  const int chunk = slot ^ ((lane & 16) ? 2 : 0);
  const bool tail_lane = (lane >> 4) == wave;
  unsigned a_off[5], b_off[B_LD];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int brow = j < 4 ? 16 * (wave + 4 * j) + lrow : 256 + lrow;
    a_off[j] = (unsigned)((m0 - 1 + brow) * p.src_ld * 2 + chunk * 16);     // mod 2^32: a negative pixel index wraps out of range
  }
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    const int n = n0 + 16 * (wave + 4 * j) + lrow;
    b_off[j] = n < p.Cd ? (unsigned)((long)n * p.w_row * 2 + chunk * 16) : B_ROW_OOB;
  }
  const int ng = p.nk / 3;                     // bands: (32-channel chunk, kernel row)
  int ga_n = 0, ga_th = 0, ga_ci = 0;          // band being issued (wave-uniform)
  // (a dead band / weight tile -- beyond the last one -- uses a descriptor of zero records: every lane out of range, zeros land; the
  // band's kernel-row offset is added per lane: a band may start one pixel in front of the tensor and end up to a row behind it, and only
  // the full 32-bit sum is range-checked by the hardware; the weight tile's offset travels in the instruction's SGPR offset)
  auto issue_band = [&](int base) {
    const bool live = ga_n < ng && !(DY_ABLATE_OF(p) & 2);
    const unsigned koff = (unsigned)((p.dh0 + p.dhs * ga_th) * p.Ws * (int)p.src_ld * 2 + ga_ci * 2);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, live ? p.src_bytes : 0u, 0x00020000);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + base + (wave + 4 * j) * 1024), 16, (int)(a_off[j] + koff), 0, 0, 0);
    if (tail_lane)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + base + 16 * 1024), 16, (int)(a_off[4] + koff), 0, 0, 0);
    ++ga_n;
    if (++ga_th == 3) { ga_th = 0; ga_ci += BK; }
  };
  int kb = 0, kb_tp = 0, kb_ci = 0;            // weight tile being issued: K order = chunk, kernel row, kernel column
  auto issue_b = [&](int base) {
    const bool live = kb < p.nk && !(DY_ABLATE_OF(p) & 4);
    const int koff = live ? (int)(((long)kb_tp * p.Cs + kb_ci) * 2) : 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, live ? p.w_bytes : 0u, 0x00020000);
#pragma unroll
    for (int j = 0; j < B_LD; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + base + (wave + 4 * j) * 1024), 16, (int)b_off[j], koff, 0, 0);
    ++kb;
    if (++kb_tp == 9) { kb_tp = 0; kb_ci += BK; }
  };

  // ---- fragment read addresses (16x16x32 operand: lane = row l&15, chunk l>>4); band row of output pixel r for column tap tw:
  // r + 1 + dw(tw)
  const int fr = lane & 15, fq = lane >> 4;
  const int r0 = 16 * MB * wm + fr;
  int a_rd[3];
#pragma unroll
  for (int tw = 0; tw < 3; ++tw) {
    const int brow = r0 + 1 + p.dw0 + p.dws * tw;
    a_rd[tw] = brow * 64 + ((fq ^ ((brow & 4) ? 2 : 0)) * 16);             // + 1024 * m-block
  }
  const int b_rd = (64 * wn + fr) * 64 + ((fq ^ ((fr & 4) ? 2 : 0)) * 16); // + 1024 * n-block
  // padding masks: bit 8*tw + i of tm[th] = tap (th, tw) of this lane's pixel of m-block i lies inside the image
  unsigned tm[3] = {0u, 0u, 0u};
  {
    const long m = m0 + r0;
    int img, h, w;
    DyTileWalk(m0, p.Hd, p.Wd).at(r0, img, h, w);
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const bool ok = m + 16 * i < p.M;
#pragma unroll
      for (int th = 0; th < 3; ++th) {
        const bool hv = (unsigned)(h + p.dh0 + p.dhs * th) < (unsigned)p.Hs;
#pragma unroll
        for (int tw = 0; tw < 3; ++tw) {
          const bool wv = (unsigned)(w + p.dw0 + p.dws * tw) < (unsigned)p.Ws;
          if (ok && hv && wv) tm[th] |= 1u << (8 * tw + i);
        }
      }
      w += 16;                                 // Wd >= 16 (launcher): at most one row wrap
      if (w >= p.Wd) {
        w -= p.Wd;
        if (++h == p.Hd) h = 0;
      }
    }
  }

  f32x4 acc[MB][4];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int band_base, int b_base, unsigned bits, int ard) {
    u32x4 bfr[4], afr[MB];
    const int av = band_base + ard;
    if (DY_ABLATE_OF(p) & 16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = u32x4{0x3f803f80u + (unsigned)b_base, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
      for (int i = 0; i < MB; ++i) afr[i] = u32x4{0x3f803f80u, 0x3f803f80u + (bits & 1u), 0x3f803f80u, 0x3f803f80u};
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const u32x4*>(smem + b_base + b_rd + 1024 * j);
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const int sel = ((bits >> i) & 1u) ? av : OFF_ZERO - 1024 * i;
        afr[i] = *reinterpret_cast<const u32x4*>(smem + sel + 1024 * i);
      }
    }
    if (DY_ABLATE_OF(p) & 8) {
#pragma unroll
      for (int i = 0; i < MB; ++i) acc[i][0][0] += __builtin_bit_cast(float, afr[i][0] ^ bfr[i & 3][1]);
      return;
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = mfma_16x16x32<T>(bfr[j], afr[i], acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
  };
  stamp(DY_ABLATE_OF(p), p.nblk, 1);

  // virtual steps -DB .. -1: bands 0 .. NBAND-2, weight tiles 0 .. DB-1
  int band_fill = 0, b_fill = 0;
#pragma unroll
  for (int s = 0; s < DB; ++s) {
    if (s % 3 == 0) {
      issue_band(band_fill);
      band_fill += BAND_BYTES;
    }
    issue_b(OFF_B + b_fill);
    b_fill += B_BYTES;
  }
  int band_cur = 0, b_cur = 0, th = 0;
  auto next_b = [&]() {
    b_cur = b_cur == DB * B_BYTES ? 0 : b_cur + B_BYTES;
    b_fill = b_fill == DB * B_BYTES ? 0 : b_fill + B_BYTES;
  };
  // With two bands (DB = 3) everything the LAST band group would issue is dead -- band ng and weight tiles nk .. nk + 2 -- and a dead
  // DMA fetches nothing but still takes the full round trip to retire, which the wait in front of the epilogue (it reuses the ring) paid
  // for once per tile.  The last group issues nothing and its waits shrink with what is no longer in flight: weight tiles t + 1, t + 2
  // (tw = 0), t + 1 (tw = 1), nothing (tw = 2).
  // (the last group is its own straight-line copy of the body: the same waits chosen by branches inside one loop body cost
  //  band_kernel<128> 36 spilled registers)
  auto group = [&](auto tailc) {
    constexpr bool tail = decltype(tailc)::value;
    const unsigned tmc = th == 0 ? tm[0] : (th == 1 ? tm[1] : tm[2]);
    // ---- tw = 0: also refills the band slot the previous band group left
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N0) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!tail) {
      issue_band(band_fill);
      issue_b(OFF_B + b_fill);
    }
    __builtin_amdgcn_sched_barrier(0);
    compute(band_cur, OFF_B + b_cur, tmc, a_rd[0]);
    __builtin_amdgcn_sched_barrier(0);
    next_b();
    // ---- tw = 1
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(tail ? B_LD : N1) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!tail) issue_b(OFF_B + b_fill);
    __builtin_amdgcn_sched_barrier(0);
    compute(band_cur, OFF_B + b_cur, tmc >> 8, a_rd[1]);
    __builtin_amdgcn_sched_barrier(0);
    next_b();
    // ---- tw = 2
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(tail ? 0 : N1) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!tail) issue_b(OFF_B + b_fill);
    __builtin_amdgcn_sched_barrier(0);
    compute(band_cur, OFF_B + b_cur, tmc >> 16, a_rd[2]);
    __builtin_amdgcn_sched_barrier(0);
    next_b();
    band_cur = band_cur == (NBAND - 1) * BAND_BYTES ? 0 : band_cur + BAND_BYTES;
    band_fill = band_fill == (NBAND - 1) * BAND_BYTES ? 0 : band_fill + BAND_BYTES;
    th = th == 2 ? 0 : th + 1;
  };
  constexpr bool peel = NBAND == 2;
  for (int g = 0; g < ng - (peel ? 1 : 0); ++g) group(std::false_type{});
  if (peel) group(std::true_type{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (three bands: the zero fills beyond the last step have landed)
  __builtin_amdgcn_s_barrier();
  stamp(DY_ABLATE_OF(p), p.nblk, 2);
  epilogue<BN, T, MB>(p, acc, smem, tid, lane, wave, wm, wn, m0, n0, tile_m);
#ifdef DY_DIAG
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  stamp(DY_ABLATE_OF(p), p.nblk, 5);
}


}  // namespace v5

extern "C" int dy_debug_conv5_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(v5::g_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : 1;
}

bool dy_conv_v5_eligible(const dy_conv_desc* d, int mode) {
  static const bool off = dy_env("DY_NO_CONV_V5") != nullptr;
  if (off || (d->dtype != DY_BF16 && d->dtype != DY_F16)) return false;
  if (!(d->Cs % 32 == 0 && d->KH * d->KW <= 25)) return false;
  if (mode == 1 && d->stride != 1) return false;
  if ((d->src_ld * 2) % 16 != 0 || (d->dst_ld * 2) % 16 != 0 || ((uintptr_t)d->dst) % 16 != 0) return false;
  const long M = (long)d->N * d->Hd * d->Wd;
  const long src_bytes = (((long)d->N * d->Hs * d->Ws - 1) * d->src_ld + d->Cs) * 2;
  const long w_row = d->KHf > 0 ? (long)d->KHf * d->KWf * d->Cs : (long)d->KH * d->KW * d->Cs;
  const long w_bytes = (long)d->Cd * w_row * 2;
  // (activation extent + the most negative tap offset folded into the descriptor's base stay below 2^31, the bit of a padded lane)
  const long halo = ((long)d->KH * d->dil * d->Ws + (long)d->KW * d->dil) * d->src_ld * 2;
  if (!(src_bytes + halo <= 0x7fffffffL && w_bytes <= 0x3fffffffL && M < (1L << 31))) return false;
  const long tiles_m = (M + 255) / 256;
  const long tn = (d->Cd + 127) / 128;
  if (d->Cd >= 96 && tn * 128 * 4 <= (long)d->Cd * 5 && tiles_m * tn >= 256) return true;
  return d->Cd >= 48 && d->Cd <= 64 && tiles_m >= 256;          // 64-wide tiles for the 64-channel layers
}

static int v5_launch(const dy_conv_desc* d, int mode, void* stream, const dy_conv_desc* classes, int ncls);

int dy_conv_v5_launch(const dy_conv_desc* d, int mode, void* stream) { return v5_launch(d, mode, stream, nullptr, 0); }

// The parity classes of a stride-2 data gradient (conv.hip: dgrad_dispatch; heaviest class first) as one conv_kernel launch
// (DY_V5_CLASSES=0 in a DIAG build: one launch per class)
bool dy_conv_v5_classes_eligible(const dy_conv_desc* c, int ncls) {
  static const bool off = dy_env("DY_NO_CONV_V5") != nullptr || (dy_env("DY_V5_CLASSES") && atoi(dy_env("DY_V5_CLASSES")) == 0);
  if (off || ncls < 2 || ncls > 4) return false;
  const dy_conv_desc* d = &c[0];
  if (d->dtype != DY_BF16 && d->dtype != DY_F16) return false;
  if (!(d->Cs % 32 == 0 && d->KHf > 0 && d->KHf * d->KWf <= 25 && d->stride == 1)) return false;
  const long src_bytes = (((long)d->N * d->Hs * d->Ws - 1) * d->src_ld + d->Cs) * 2;
  const long w_bytes = (long)d->Cd * d->KHf * d->KWf * d->Cs * 2;
  const long halo = ((long)d->KHf * d->Ws + d->KWf) * d->src_ld * 2;
  if (!(src_bytes + halo <= 0x7fffffffL && w_bytes <= 0x3fffffffL)) return false;
  const long tn = (d->Cd + 127) / 128;
  const bool wide = d->Cd >= 96 && tn * 128 * 4 <= (long)d->Cd * 5, narrow = d->Cd >= 48 && d->Cd <= 64;
  if (!wide && !narrow) return false;
  long tiles = 0;
  for (int i = 0; i < ncls; ++i) {
    const dy_conv_desc& q = c[i];
    if (q.src != d->src || q.w != d->w || q.Cs != d->Cs || q.Cd != d->Cd || q.dtype != d->dtype || q.stride != 1 || q.dil != 1 || q.KHf != d->KHf ||
        q.KWf != d->KWf || q.kh_step != d->kh_step || q.kw_step != d->kw_step || q.dst_ld != d->dst_ld || q.dst_row_stride != d->dst_row_stride ||
        q.dst_img_stride != d->dst_img_stride || q.src_ld != d->src_ld || q.accumulate != d->accumulate || q.scale || q.shift || q.stats ||
        q.act != DY_ACT_NONE)
      return false;
    if ((q.dst_ld * 2) % 16 != 0 || ((uintptr_t)q.dst) % 16 != 0 || (q.src_ld * 2) % 16 != 0) return false;
    if ((long)q.N * q.Hd * q.Wd >= (1L << 31)) return false;
    tiles += (((long)q.N * q.Hd * q.Wd + 255) / 256) * (wide ? tn : 1);
  }
  return tiles >= 256;
}

int dy_conv_v5_launch_classes(const dy_conv_desc* c, int ncls, void* stream) { return v5_launch(&c[0], 0, stream, c, ncls); }

static int v5_launch(const dy_conv_desc* d, int mode, void* stream, const dy_conv_desc* classes, int ncls) {
  v5::P p;
  p.src = (const char*)d->src; p.w = (const char*)d->w; p.dst = (char*)d->dst;
  p.src_ld = d->src_ld; p.dst_ld = d->dst_ld;
  p.src_bytes = (unsigned)((((long)d->N * d->Hs * d->Ws - 1) * d->src_ld + d->Cs) * 2);
  p.Hs = d->Hs; p.Ws = d->Ws; p.Cs = d->Cs; p.Hd = d->Hd; p.Wd = d->Wd; p.Cd = d->Cd;
  p.KH = d->KH; p.KW = d->KW;
  if (mode == 0) {
    p.stride = d->stride; p.dh0 = -d->pad; p.dhs = d->dil; p.dw0 = -d->pad; p.dws = d->dil;
  } else {               // stride-1 data gradient: dx[h] += dz[h + pad - kh*dil] * w[kh]
    p.stride = 1; p.dh0 = d->pad; p.dhs = -d->dil; p.dw0 = d->pad; p.dws = -d->dil;
  }
  if (d->KHf > 0) {
    p.kh0 = d->kh0; p.khs = d->kh_step; p.kw0 = d->kw0; p.kws = d->kw_step; p.KWf = d->KWf;
    p.w_row = (long)d->KHf * d->KWf * d->Cs;
  } else {
    p.kh0 = 0; p.khs = 1; p.kw0 = 0; p.kws = 1; p.KWf = d->KW;
    p.w_row = (long)d->KH * d->KW * d->Cs;
  }
  p.w_bytes = (unsigned)((long)d->Cd * p.w_row * 2);
  {
    const int dh_lo = p.dhs < 0 ? p.dh0 + p.dhs * (p.KH - 1) : p.dh0, dw_lo = p.dws < 0 ? p.dw0 + p.dws * (p.KW - 1) : p.dw0;
    const long lo = ((long)dh_lo * p.Ws + dw_lo) * p.src_ld * 2;
    p.a_min = lo < 0 ? (int)lo : 0;
  }
  p.scale = d->scale; p.shift = d->shift; p.act = d->act; p.stats = d->stats; p.accumulate = d->accumulate;
  // (bit 1: an output beyond 128 MB is streamed past the L2 with non-temporal stores -- its lines would evict the operand lines the
  //  taps re-read, and whoever reads it next streams it from memory anyway)
  if ((long)d->N * d->Hd * d->Wd * d->Cd * 2 > (128L << 20)) p.accumulate |= 2;
  p.M = (long)d->N * d->Hd * d->Wd;
  p.add_src = mode == 1 ? (const char*)d->add_src : nullptr; p.add_src_ld = d->add_src_ld;
  p.nk = d->KH * d->KW * d->Cs / v5::BK;
  static const int ablate = dy_env("DY_ABLATE") ? atoi(dy_env("DY_ABLATE")) : 0;
  p.ablate = ablate;
  p.dst_row = d->dst_row_stride;
  p.dst_img = d->dst_img_stride ? d->dst_img_stride : (long)d->Hd * d->dst_row_stride;
  const int bn = d->Cd <= 64 ? 64 : 128;
  p.tiles_n = dy_cdiv(d->Cd, bn);
  p.nblk = dy_cdiv(p.M, v5::BM) * p.tiles_n;
  p.ncls = 0;
  if (ncls > 1) {
    DY_CHECK(ncls <= 4 && mode == 0, "conv_v5: at most 4 forward-style classes");
    p.ncls = ncls;
    int slot = 0;
    for (int c = 0; c < ncls; ++c) {
      const dy_conv_desc& q = classes[c];
      DyParityCls& k = p.cls[c];
      k.dst = (char*)q.dst; k.M = (long)q.N * q.Hd * q.Wd; k.Hd = q.Hd; k.Wd = q.Wd; k.KH = q.KH; k.KW = q.KW; k.pad = q.pad;
      k.kh0 = q.kh0; k.kw0 = q.kw0; k.Ktot = q.KH * q.KW * q.Cs;
      const int tiles = (int)dy_cdiv(k.M, (long)v5::BM) * p.tiles_n;
      k.blk0 = slot;
      k._r = dy_cdiv(tiles, 8);
      slot += k._r;
    }
    p.nblk = 8 * slot;
  }
  constexpr int RING128 = v5::NSTAGE * (v5::A_BYTES + 128 * 64), EPI128 = dy_epi::row_image_bytes<v5::BM, 128>();
  constexpr int RING64 = 4 * (v5::A_BYTES + 64 * 64), EPI64 = dy_epi::row_image_bytes<v5::BM, 64>();
  constexpr int SH128 = RING128 > EPI128 ? RING128 : EPI128, SH64 = RING64 > EPI64 ? RING64 : EPI64;
  constexpr int RING64S = 2 * (v5::A_BYTES + 64 * 64), SH64S = RING64S > EPI64 ? RING64S : EPI64;
  static_assert(4 * SH64S <= 160 * 1024, "four blocks per CU");
  static const bool slim64 = !(dy_env("DY_CONV64_SLIM") && atoi(dy_env("DY_CONV64_SLIM")) == 0);
  constexpr int BAND128 = 2 * v5::BAND_BYTES + 4 * 128 * 64 + 64, BAND64 = 3 * v5::BAND_BYTES + 7 * 64 * 64 + 64;
  constexpr int SB128 = BAND128 > EPI128 ? BAND128 : EPI128, SB64 = BAND64 > EPI64 ? BAND64 : EPI64;
  // 64-wide tiles with the 128-wide variant's shallower rings (2 bands + 4 weight tiles = 51 KiB): THREE co-resident blocks per CU
  constexpr int BAND64S = 2 * v5::BAND_BYTES + 4 * 64 * 64 + 64, SB64S = BAND64S > EPI64 ? BAND64S : EPI64;
  static_assert(3 * SB64S <= 160 * 1024, "three blocks per CU");
  static const int band64_nband = dy_env("DY_BAND64_NBAND") ? atoi(dy_env("DY_BAND64_NBAND")) : 2;
  static_assert(2 * SH128 <= 160 * 1024 && 2 * SH64 <= 160 * 1024 && 2 * SB128 <= 160 * 1024 && 2 * SB64 <= 160 * 1024, "two blocks per CU");
  // 3x3 / stride 1 / pad 1 on an unchanged pixel grid: the band kernel
  static const bool no_band = dy_env("DY_NO_CONV_BAND") != nullptr;
  const bool band = ncls <= 1 && !no_band && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->dil == 1 && d->KHf <= 0 && d->Hs == d->Hd &&
                    d->Ws == d->Wd && d->Ws >= 16;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&v5::conv_kernel<128, bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, SH128);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&v5::conv_kernel<64, bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, SH64);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&v5::conv_kernel<128, f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, SH128);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&v5::conv_kernel<64, f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, SH64);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&v5::conv_kernel<64, bf16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SH64S);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&v5::conv_kernel<64, f16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SH64S);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&v5::band_kernel<128, bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, SB128);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&v5::band_kernel<64, bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, SB64);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&v5::band_kernel<128, f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, SB128);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&v5::band_kernel<64, f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, SB64);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&v5::band_kernel<64, bf16_t, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, SB64S);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&v5::band_kernel<64, f16_t, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, SB64S);
    if (e != hipSuccess) {
      dy_set_error("conv_v5: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 3;
    }
    configured = true;
  }
  const bool f16 = d->dtype == DY_F16;
  if (band && bn == 128) {
    dy_note_kernel("v5::band_kernel<128>");
    if (f16) v5::band_kernel<128, f16_t><<<p.nblk, 256, SB128, (hipStream_t)stream>>>(p);
    else v5::band_kernel<128, bf16_t><<<p.nblk, 256, SB128, (hipStream_t)stream>>>(p);
  } else if (band && band64_nband == 2) {
    dy_note_kernel("v5::band_kernel<64>");
    if (f16) v5::band_kernel<64, f16_t, 2><<<p.nblk, 256, SB64S, (hipStream_t)stream>>>(p);
    else v5::band_kernel<64, bf16_t, 2><<<p.nblk, 256, SB64S, (hipStream_t)stream>>>(p);
  } else if (band) {
    dy_note_kernel("v5::band_kernel<64>");
    if (f16) v5::band_kernel<64, f16_t><<<p.nblk, 256, SB64, (hipStream_t)stream>>>(p);
    else v5::band_kernel<64, bf16_t><<<p.nblk, 256, SB64, (hipStream_t)stream>>>(p);
  } else if (bn == 128) {
    dy_note_kernel("v5::conv_kernel<128>");
    if (f16) v5::conv_kernel<128, f16_t><<<p.nblk, 256, SH128, (hipStream_t)stream>>>(p);
    else v5::conv_kernel<128, bf16_t><<<p.nblk, 256, SH128, (hipStream_t)stream>>>(p);
  } else if (slim64) {
    dy_note_kernel("v5::conv_kernel<64>");
    if (f16) v5::conv_kernel<64, f16_t, true><<<p.nblk, 256, SH64S, (hipStream_t)stream>>>(p);
    else v5::conv_kernel<64, bf16_t, true><<<p.nblk, 256, SH64S, (hipStream_t)stream>>>(p);
  } else {
    dy_note_kernel("v5::conv_kernel<64>");
    if (f16) v5::conv_kernel<64, f16_t><<<p.nblk, 256, SH64, (hipStream_t)stream>>>(p);
    else v5::conv_kernel<64, bf16_t><<<p.nblk, 256, SH64, (hipStream_t)stream>>>(p);
  }
  DY_LAUNCH_CHECK();
  return 0;
}
