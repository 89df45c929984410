// Phase-interleaved implicit-GEMM convolution for the wide bf16 layers (>= 256 output channels): forward and stride-1 data gradient
// of nn.Conv2d inside Conv / Bottleneck / Detect (ultralytics/nn/modules/conv.py:38-55, block.py:553-565, head.py:40-46).
//
// conv_v2.hip keeps one barrier per 64-deep K-step and lets both waves of a SIMD run the same program in lockstep: the counters
// showed 22-30 % MFMA-busy with 38-56 % of the wave cycles parked at the barrier / vmcnt(0).  This kernel is built around the
// other structure the CDNA4 playbook describes for one block per CU:
//   * 256 x 256 block tile, K-step 64, 8 waves = 2 groups of 4 (one wave of each group per SIMD).  The groups run the SAME
//     program one barrier apart (group 1 executes one extra s_barrier before the loop, group 0 one after it): while one wave
//     of a SIMD issues its 16 MFMAs of a phase the other one issues the LDS reads + DMA of its next phase;
//   * a K-step is 4 phases, each = one 64 x 32 quadrant of the wave's 128 x 64 output (16 x v_mfma_f32_16x16x32_bf16) and ONE
//     16 KiB half-tile of DMA (2 x buffer_load_dwordx4 ... lds per lane).  A wave's rows / columns are split over BOTH halves of
//     the A / B tile (rows 64*wr + [0,64) of each 128-row half) so that a half-tile is dead after ONE phase and can be refilled
//     while the K-step is still being computed: order of reads  P1: B-half 0 (4 x b128, retired before the barrier by a
//     counted lgkmcnt(8)) + A-half 0 (8),  P2: B-half 1 (4),  P3: A-half 1 (8),  P4: none; order of refills
//     P1: A-half 1 of step k+1,  P2: B-half 0,  P3: A-half 0,  P4: B-half 1 of step k+2;
//   * the DMA is never drained inside the loop: one s_waitcnt vmcnt(6) per K-step (in P4, before its barrier) leaves the three
//     youngest half-tiles in flight across the barriers; a buffer is read from the phase AFTER the wait that retires it;
//   * the implicit-GEMM gather costs ONE VALU per activation DMA and none per weight DMA: every lane keeps the 32-bit byte offset of
//     its output pixel (the instruction's VGPR offset, constant over the K loop); the window tap and channel chunk of the K-step are
//     wave-uniform and travel in the instruction's SGPR offset; a per-lane word of "tap t is padding" bits is rotated once per K-step
//     (inside the MFMA cluster, where a VALU slot is free) so that the current tap's bit sits in bit 31 and one v_and_or puts it on top
//     of the offset: padding taps, rows beyond M and channels beyond Cd then lie outside the buffer descriptor, for which the hardware
//     writes zeros (no zero page, no branches); K-steps beyond the last one use a descriptor of zero records.  (The first version
//     spent 4 dependent VALU per activation piece and 1 per weight piece in the load part of a phase -- next to the partner wave's
//     MFMA cluster, where a VALU instruction of the other wave waits ~16 cycles for an issue slot.)
//   * LDS rows are 128 bytes (64 bf16) with the 16-byte slot XOR-ed by (row>>1)&7 on the DMA source side and on the read side:
//     the 16-lane groups of ds_read_b128 hit 16 different slots of the 256-byte bank row for the 16x16x32 operand layout.
// Epilogue as in conv_epilogue.h (transposed bf16 image, ds_read_b64_tr_b16, 16-byte stores, BatchNorm sums), written for the
// 16x16 accumulator layout.
#include <stdlib.h>
#include <type_traits>
#include "dy_common.h"
#include "conv_epilogue.h"
#include "../../include/dedark_yolo.h"

namespace v4 {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF = 128 * 128;                     // one half-tile: 128 rows x 64 bf16
constexpr int OFF_A0 = 0, OFF_A1 = HALF, OFF_B0 = 2 * HALF, OFF_B1 = 3 * HALF, BUF = 4 * HALF;
// bit 31 of an activation offset marks a padded lane: beyond any source extent (checked by the dispatcher: < 2 GiB)
constexpr unsigned B_ROW_OOB = 0x40000000u;         // weight extent <= 1 GiB: row-invalid + any k offset stays out of range

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

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
                                 // the per-K-step SGPR offsets are >= 0
  int nk;                        // K-steps = KH*KW*Cs/64
  int tiles_n, nblk;
  int n_full, rb_tail;           // blocks [0, n_full) are 256-row tiles; the rest are tail tiles of 64 * rb_tail rows (see v4_launch)
  const char* add_src;           // optional addend view of a data gradient (dy_conv_desc.add_src)
  long add_src_ld;
  // > 1: the parity classes of a stride-2 data gradient in ONE launch (forward-style problems over the same dz and weights that differ
  // in destination offset, grid extent, tap subset and pad).  cls[c].blk0 = first slot of class c in every XCD's block sequence,
  // cls[c]._r = its slots per XCD: each XCD works through its eighth of class 0 (the one with the most taps), then of class 1, ...
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

// ABL: compile-time ablation mask for tools/v4_diag (the library only instantiates ABL = 0): 1 no DMA inside the loop, 2 no MFMA,
// 4 no LDS fragment reads, 8 no stagger between the wave groups, 16 no A-side DMA, 32 no B-side DMA, 64 no epilogue stores,
// 128 no s_setprio around the MFMA clusters, 128 + 256 priority to the loading wave instead, 2048 phases merged in pairs (timing only).
// RB = 16-row blocks per wave and A-half: 4 = the full 256-row tile; 3 / 2 = TAIL tiles of 192 / 128 rows.  A tail tile keeps the
// LDS layout and the DMA / phase schedule of the full tile, but only the first RG = 16 * RB rows of every 64-row group hold pixels
// (LDS row r <-> tile row (r >> 6) * RG + (r & 63)); the other rows are fed by out-of-range DMAs (zeros, no traffic) and their MFMAs,
// fragment reads and stores are not issued.  Why: 400 tiles on 256 CUs are two rounds with 44 % of the second one idle; cutting the last
// 144 tiles' rows into 192 shorter tiles fills the CUs of that round with less work each (v4_launch picks the split).
template <int ABL, typename T, int RB>
__device__ __forceinline__ void conv_tile(const P& p, char* smem, const long m0, const int n0, const int tile_m) {
  constexpr int RG = 16 * RB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // records of the two descriptors: set to 0 behind the last K-step (every lane out of range: zeros land in a buffer nobody reads)
  unsigned nrec_a = p.src_bytes - p.a_min, nrec_b = p.w_bytes;

  // ---- DMA bookkeeping: instruction j of this wave fills rows 8*(wave + 8j) .. +7 of a half-tile; lane -> (row, 16-byte slot)
  const int lrow = lane >> 3, slot = lane & 7;
  const int chunk = slot ^ ((4 * wave + (lane >> 4)) & 7);      // logical 16-byte chunk of the row this lane fetches
  unsigned a_off[4], a_pad[4], b_off[4];                         // index = 2 * half + j
  {
    const DyTileWalk walk(m0, p.Hd, p.Wd);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rl = 128 * (i >> 1) + 8 * (wave + 8 * (i & 1)) + lrow;       // LDS row
      const int r = RB == 4 ? rl : (rl >> 6) * RG + (rl & 63);               // tile row
      const bool ok = (RB == 4 || (rl & 63) < RG) && m0 + r < p.M;
      int img, oh, ow;
      walk.at(r, img, oh, ow);
      const int sh0 = oh * p.stride, sw0 = ow * p.stride;
      a_off[i] = (unsigned)((((long)img * p.Hs + sh0) * p.Ws + sw0) * p.src_ld * 2 + chunk * 16);
      unsigned wb = 0, mk = 0;                             // bit th * KW + tw: window tap (th, tw) of this row lies inside the image
      for (int tw = 0; tw < p.KW; ++tw) {
        const int sw = sw0 + p.dw0 + p.dws * tw;
        if (sw >= 0 && sw < p.Ws) wb |= 1u << tw;
      }
      for (int th = 0; th < p.KH; ++th) {
        const int sh = sh0 + p.dh0 + p.dhs * th;
        if (ok && sh >= 0 && sh < p.Hs) mk |= wb << (th * p.KW);
      }
      a_pad[i] = __builtin_amdgcn_alignbit(~mk, ~mk, 1);          // "tap t is padding" bits, rotated so that tap 0's bit is bit 31
      const int n = n0 + 128 * (i >> 1) + 8 * (wave + 8 * (i & 1)) + lrow;
      b_off[i] = n < p.Cd ? (unsigned)((long)n * p.w_row * 2 + chunk * 16) : B_ROW_OOB;
    }
  }
  // K-step whose half-tiles are being issued (all wave-uniform).  K order: all taps of one BK-channel chunk, then the next chunk.  The
  // taps re-read the SAME pixels (shifted windows), so with the taps innermost a tile's live set is (tile + halo) x BK channels (~40 KB;
  // 32 CUs x 40 KB sit in an XCD's 4 MiB L2) and the 2nd .. 9th reads are L2 hits; tap-major order cycled through the tile's full
  // channel depth between two reads of a line (150 KB per CU: more than its L2 share -> every tap went back to the Infinity Cache,
  // 2.4x fabric traffic).
  // The per-tap byte offsets sit in one VGPR each (lane t = tap t <= 25) and advance() picks them with v_readlane: the walk used to
  // recompute them with a dozen dependent s_mul / s_add and two branches per K-step, inside P1's load part that the partner group's
  // MFMA cluster waits for (s_memtime stamps, tools/v4_diag: P1 ran ~280 cycles longer than P3, which issues the same reads and DMA).
  // advance() itself (two v_readlane, four mask rotations) now runs in the middle of P1's MFMA cluster.
  const int ntaps = p.KH * p.KW;
  int vtap_a, vtap_b;
  {
    const int t = lane < ntaps ? lane : 0;
    const int th = t / p.KW, tw = t - th * p.KW;
    vtap_a = ((p.dh0 + p.dhs * th) * p.Ws + p.dw0 + p.dws * tw) * (int)p.src_ld * 2 - p.a_min;      // >= 0
    vtap_b = (int)((((long)((p.kh0 + p.khs * th) * p.KWf + p.kw0 + p.kws * tw)) * p.Cs) * 2);
  }
  int sk = 0, s_tap = 0, s_ci2 = 0;
  int a_koff = __builtin_amdgcn_readlane(vtap_a, 0);
  int b_koff = __builtin_amdgcn_readlane(vtap_b, 0);
  auto advance = [&]() {
    ++sk;
    const bool wrap = s_tap + 1 == ntaps;
    s_tap = wrap ? 0 : s_tap + 1;
    s_ci2 += wrap ? 2 * BK : 0;
    a_koff = __builtin_amdgcn_readlane(vtap_a, s_tap) + s_ci2;
    b_koff = __builtin_amdgcn_readlane(vtap_b, s_tap) + s_ci2;
    const int rot = wrap ? (33 - ntaps) & 31 : 1;                  // next tap's padding bit to bit 31
#pragma unroll
    for (int i = 0; i < 4; ++i) a_pad[i] = __builtin_amdgcn_alignbit(a_pad[i], a_pad[i], rot);
    if (sk >= p.nk) { nrec_a = 0; nrec_b = 0; a_koff = 0; b_koff = 0; }
  };
  if (p.nk < 1) { nrec_a = 0; nrec_b = 0; a_koff = 0; b_koff = 0; }
  bool in_loop = false;
  auto stage_a = [&](int buf, int h) {
    if ((ABL & 1) && in_loop) return;
    if (ABL & 16) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.src + p.a_min), 0, nrec_a, 0x00020000);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = 2 * h + j;
      const unsigned v = (a_pad[i] & 0x80000000u) | a_off[i];       // padding -> beyond any extent (a_off < 2^31)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + buf * BUF + (h ? OFF_A1 : OFF_A0) + (wave + 8 * j) * 1024), 16,
                                               (int)v, a_koff, 0, 0);
    }
  };
  auto stage_b = [&](int buf, int h) {
    if ((ABL & 1) && in_loop) return;
    if (ABL & 32) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, nrec_b, 0x00020000);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = 2 * h + j;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + buf * BUF + (h ? OFF_B1 : OFF_B0) + (wave + 8 * j) * 1024), 16,
                                               (int)b_off[i], b_koff, 0, 0);
    }
  };

  // ---- fragment read addresses (16x16x32: lane = row (lane&15), k chunk (lane>>4) of a 32-deep block)
  const int fr = lane & 15, fq = lane >> 4, key = (fr >> 1) & 7;
  const int off0 = ((((key >> 2) << 2) | ((fq ^ key) & 3)) << 4);
  const int a_rd = (64 * wr + fr) * 128 + off0;          // + half base + 2048 * m-block ; k block 1: ^ 64
  const int b_rd = (32 * wc + fr) * 128 + off0;

  f32x4 acc[2][2][RB][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 afr[RB][2], bfr[2][2][2];

  auto rd = [&](int byte) {
    if (ABL & 4) return u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    return *reinterpret_cast<const u32x4*>(smem + byte);
  };
  auto read_a = [&](int buf, int h) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      afr[i][0] = rd(buf * BUF + (h ? OFF_A1 : OFF_A0) + a_rd + 2048 * i);
      afr[i][1] = rd(buf * BUF + (h ? OFF_A1 : OFF_A0) + (a_rd ^ 64) + 2048 * i);
    }
  };
  // `rs` = register set the fragments go to.  The two sets swap roles every K-step: in a K-step of parity b, B-half 0 lives in set b and
  // B-half 1 in set b ^ 1 -- so that P4, whose MFMAs use half 0 only, can already read half 0 of the NEXT K-step into the set half 1 has
  // just vacated (see kstep: the load parts then carry 8 / 4 / 8 / 4 fragment reads instead of 12 / 4 / 8 / 0)
  auto read_b = [&](int buf, int h, int rs) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bfr[rs][j][0] = rd(buf * BUF + (h ? OFF_B1 : OFF_B0) + b_rd + 2048 * j);
      bfr[rs][j][1] = rd(buf * BUF + (h ? OFF_B1 : OFF_B0) + (b_rd ^ 64) + 2048 * j);
    }
  };
  // `mid` runs between the two k-halves of the cluster (P1: the K-walk bookkeeping, whose VALU then issues between this wave's own MFMAs)
  auto mma = [&](int ah, int bh, int rs, auto mid) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (ABL & 256) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    if (ABL & 2) {
#pragma unroll
      for (int i = 0; i < RB; ++i) { asm volatile("" ::"v"(afr[i][0])); asm volatile("" ::"v"(afr[i][1])); }
#pragma unroll
      for (int j = 0; j < 2; ++j) { asm volatile("" ::"v"(bfr[rs][j][0])); asm volatile("" ::"v"(bfr[rs][j][1])); }
      mid();
      return;
    }
    if (!(ABL & 128)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          // transposed product (rows = output channels, columns = pixels): a lane ends up with 4 consecutive CHANNELS of one pixel
          acc[ah][bh][i][j] = mfma_16x16x32<T>(bfr[rs][j][kb], afr[i][kb], acc[ah][bh][i][j]);
      if (kb == 0) {
        __builtin_amdgcn_sched_barrier(0);
        mid();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!(ABL & 128)) __builtin_amdgcn_s_setprio(0);
    if (ABL & 256) __builtin_amdgcn_s_setprio(1);       // (experiment: the loading wave gets the priority)
    __builtin_amdgcn_sched_barrier(0);
  };
  auto nothing = []() {};

  // ---- prologue: K-step 0 complete + three half-tiles of K-step 1 in flight
  stage_b(0, 0); stage_a(0, 0); stage_b(0, 1); stage_a(0, 1);
  advance();
  stage_b(1, 0); stage_a(1, 0); stage_b(1, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_b(0, 0, 0);                                                   // B-half 0 of K-step 0 (later ones are read in P4 of the step before)
  if (!(ABL & 8) && wr == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind group 0
  in_loop = true;

  // ABL & 512 (tools/v4_diag only): s_memtime stamps at the three points of a phase where no LDS read is in flight -- phase start,
  // after the first barrier + lgkmcnt(0), after the MFMA cluster -- summed per phase over the K-loop; lane 0 of waves 0 and 4 of
  // block 0 write the sums to the buffer passed as `stats`.
  unsigned long long t_sum[4][3] = {};
  unsigned long long t_prev = 0;
  auto stamp = [&](int ph, int which) {
    if (!(ABL & 512)) return;
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    if (which >= 0) t_sum[ph][which] += t - t_prev;
    t_prev = t;
  };
  auto kstep = [&](auto bufc) {
    constexpr int b = decltype(bufc)::value;
    // P1
    stamp(3, 2);                                   // (closes the previous phase: second barrier of P4)
    read_a(b, 0);
    __builtin_amdgcn_sched_barrier(0);
    stage_a(b ^ 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    stamp(0, 0);
    mma(0, 0, b, [&]() { advance(); });
    stamp(0, 1);
    if (!(ABL & 2048)) __builtin_amdgcn_s_barrier();
    // P2
    stamp(0, 2);
    read_b(b, 1, b ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    stage_b(b, 0);                                 // B-half 0 of this buffer was read in P4 of the previous K-step: free since then
    __builtin_amdgcn_sched_barrier(0);
    if (!(ABL & 2048)) __builtin_amdgcn_s_barrier();
    stamp(1, 0);
    mma(0, 1, b ^ 1, nothing);
    stamp(1, 1);
    __builtin_amdgcn_s_barrier();
    // P3
    stamp(1, 2);
    read_a(b, 1);
    __builtin_amdgcn_sched_barrier(0);
    stage_a(b, 0);
    __builtin_amdgcn_sched_barrier(0);
    // B-half 0 of K-step k+1 (issued in P2 of K-step k-1) has landed for this wave: everything but the 10 youngest pieces
    // (A0 / B1 / A1 of k+1, B0 / A0 of k+2).  Behind this phase's barriers all waves have been here, so P4 may read it.
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stamp(2, 0);
    mma(1, 1, b ^ 1, nothing);
    stamp(2, 1);
    if (!(ABL & 2048)) __builtin_amdgcn_s_barrier();
    // P4
    stamp(2, 2);
    read_b(b ^ 1, 0, b ^ 1);                       // next K-step's B-half 0 into the register set B-half 1 has just vacated
    __builtin_amdgcn_sched_barrier(0);
    stage_b(b, 1);
    __builtin_amdgcn_sched_barrier(0);
    // everything but the three youngest half-tiles: K-step k+1 is complete
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if (!(ABL & 2048)) __builtin_amdgcn_s_barrier();
    stamp(3, 0);
    mma(1, 0, b, nothing);
    stamp(3, 1);
    __builtin_amdgcn_s_barrier();
  };

  stamp(0, -1);
  for (int kt = 0; kt < p.nk; kt += 2) {
    kstep(std::integral_constant<int, 0>{});
    if (kt + 1 < p.nk) kstep(std::integral_constant<int, 1>{});
  }
  if ((ABL & 512) && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4)) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.stats) + (wave ? 12 : 0);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 3; ++j) o[3 * i + j] = t_sum[i][j];
  }
  if (!(ABL & 8) && wr == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the zero fills of the steps beyond the last one have landed
  __builtin_amdgcn_s_barrier();

  // ---- epilogue: bf16 image [pixel][channel] (conv_epilogue.h: store_rows) -> 16-byte stores, 256 contiguous bytes per quarter-wave
  constexpr int PT = dy_epi::row_pitch<BN>();
  const int cl = lane & 15, g = lane >> 4;
  const bool plain = !p.scale && !p.shift && p.act == DY_ACT_NONE;   // raw output: training forward (BatchNorm follows), data gradients
#pragma unroll
  for (int bh = 0; bh < 2; ++bh)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c0 = 128 * bh + 32 * wc + 16 * j + 4 * g;       // this lane's 4 channels of the block
      if (plain) {
#pragma unroll
        for (int ah = 0; ah < 2; ++ah)
#pragma unroll
          for (int i = 0; i < RB; ++i) {
            const int px = 128 * ah + 64 * wr + 16 * i + cl;
            uint2 w2 = {dy_epi::pack2<T>(acc[ah][bh][i][j][0], acc[ah][bh][i][j][1]), dy_epi::pack2<T>(acc[ah][bh][i][j][2], acc[ah][bh][i][j][3])};
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
        for (int ah = 0; ah < 2; ++ah)
#pragma unroll
          for (int i = 0; i < RB; ++i) {
            const int px = 128 * ah + 64 * wr + 16 * i + cl;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float u = acc[ah][bh][i][j][e] * sc[e] + sf[e];
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
  auto off_fn = [&](long m) { return dst_offset(p, m); };
  if (!(ABL & 64))
    dy_epi::store_rows<BM, BN, 8, T, decltype(off_fn), RG>(smem, lane, wave, m0, n0, p.M, p.Cd, p.accumulate, reinterpret_cast<T*>(p.dst), off_fn,
                                                           reinterpret_cast<const T*>(p.add_src), p.add_src_ld);
  if (p.stats && !(ABL & 512)) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);        // [2 (wr)][BN][2]
#pragma unroll
    for (int bh = 0; bh < 2; ++bh)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // rows beyond M and channels beyond Cd were fed zeros by the DMA: their accumulators are exactly 0, no predicate needed
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int ah = 0; ah < 2; ++ah)
#pragma unroll
            for (int i = 0; i < RB; ++i) {
              const float a = acc[ah][bh][i][j][e];
              s1 += a;
              s2 += a * a;
            }
          s1 = row16_sum(s1);                                  // over the 16 pixels on the lanes of a row group
          s2 = row16_sum(s2);
          if (cl == 0) {
            const int col = 128 * bh + 32 * wc + 16 * j + 4 * g + e;
            red[(wr * BN + col) * 2] = s1;
            red[(wr * BN + col) * 2 + 1] = s2;
          }
        }
    __syncthreads();
    if (tid < BN) {
      const int n = n0 + tid;
      if (n < p.Cd) {
        const float s1 = red[tid * 2] + red[(BN + tid) * 2];
        const float s2 = red[tid * 2 + 1] + red[(BN + tid) * 2 + 1];
        double* st = p.stats + (long)(tile_m % DY_STATS_REPLICAS) * 2 * p.Cd;
        atomic_add_f64(st + n, (double)s1);
        atomic_add_f64(st + p.Cd + n, (double)s2);
      }
    }
  }
}

template <int ABL, typename T = bf16_t>
__global__ __launch_bounds__(512) void conv_kernel(const P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (p.ncls > 1) {
    // several problems in one launch: this block's class replaces the launch-wide geometry.  Four launches of 1-, 2-, 2- and 4-tap
    // problems each end in their own partial round of tiles (100-400 tiles on 256 CUs, four times); together the classes fill the
    // rounds, and the long tiles go first.  The XCD gets an equal share of EVERY class (a plain xcd_remap over the class-major order
    // would hand the 4-tap tiles to two XCDs and the 1-tap tiles to two others).
    const int x = blockIdx.x & 7, i = (int)blockIdx.x >> 3;
    int c = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (k < p.ncls && i >= p.cls[k].blk0) c = k;
    const DyParityCls& k = p.cls[c];
    const int tile = x * k._r + (i - k.blk0);
    const int tiles = (int)((k.M + BM - 1) / BM) * p.tiles_n;
    if (tile >= tiles) return;                       // (block-uniform: up to 7 surplus blocks per class)
    P q = p;
    q.dst = k.dst; q.M = k.M; q.Hd = k.Hd; q.Wd = k.Wd; q.KH = k.KH; q.KW = k.KW;
    q.dh0 = -k.pad; q.dw0 = -k.pad; q.kh0 = k.kh0; q.kw0 = k.kw0;
    q.nk = k.Ktot / BK;
    q.a_min = -(k.pad * p.Ws + k.pad) * (int)p.src_ld * 2;
    const int tile_m = tile / p.tiles_n, tile_n = tile - tile_m * p.tiles_n;
    conv_tile<ABL, T, 4>(q, smem, (long)tile_m * BM, tile_n * BN, tile_m);
    return;
  }
  // full tiles first (they are dispatched first), each population spread over the XCDs on its own
  const bool tail = (int)blockIdx.x >= p.n_full;
  const int bid = tail ? p.n_full + xcd_remap(blockIdx.x - p.n_full, p.nblk - p.n_full) : xcd_remap(blockIdx.x, p.n_full);
  const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
  const int n0 = tile_n * BN;
  if (!tail) {
    conv_tile<ABL, T, 4>(p, smem, (long)tile_m * BM, n0, tile_m);
  } else {
    const int full_m = p.n_full / p.tiles_n;
    const long m0 = (long)full_m * BM + (long)(tile_m - full_m) * 64 * p.rb_tail;
    if (p.rb_tail == 3) conv_tile<ABL, T, 3>(p, smem, m0, n0, tile_m);
    else conv_tile<ABL, T, 2>(p, smem, m0, n0, tile_m);
  }
}

}  // namespace v4

// Taken when the 256-wide channel tiles carry <= 20 % padding, the reduction is long enough to amortise the 128 KiB epilogue of a
// tile (K >= 512; shorter ones and the narrower layers go to conv_v5.hip, two smaller blocks per CU) and the tiles fill most of
// the chip (one block per CU).
static int v4_variant(const dy_conv_desc* d, int mode) {
  if (d->dtype != DY_BF16 && d->dtype != DY_F16) return 0;
  if (!(d->Cs % 64 == 0 && d->KH * d->KW <= 25)) return 0;
  if (mode == 1 && d->stride != 1) return 0;
  if ((d->src_ld * 2) % 16 != 0 || (d->dst_ld * 2) % 16 != 0 || ((uintptr_t)d->dst) % 16 != 0) return 0;
  const long src_bytes = (((long)d->N * d->Hs * d->Ws - 1) * d->src_ld + d->Cs) * 2;
  const long w_row = d->KHf > 0 ? (long)d->KHf * d->KWf * d->Cs : (long)d->KH * d->KW * d->Cs;
  const long w_bytes = (long)d->Cd * w_row * 2;
  // activation descriptor: extent + the most negative tap offset folded into its base (v4_launch: a_min) stays below 2^31, the bit
  // that marks a padded lane
  const long halo = ((long)d->KH * d->dil * d->Ws + (long)d->KW * d->dil) * d->src_ld * 2;
  if (!(src_bytes + halo <= 0x7fffffffL && w_bytes <= 0x3fffffffL)) return 0;
  const long tiles_m = ((long)d->N * d->Hd * d->Wd + 255) / 256;
  const long t256 = (d->Cd + 255) / 256;
  // K >= 512.  (K >= 256 is 11 % faster on the 1024->256 1x1 data gradient at 80x80 -- 487 -> 434 us -- and equal on 256->256; it was
  // measured and not adopted: 0.1 ms of a 68 ms step, and it mixes 4-K-step launches into this kernel's roofline population.)
  static const long min_k = dy_env("DY_V4_MINK") ? atol(dy_env("DY_V4_MINK")) : 512;
  if (d->Cd >= 192 && t256 * 256 * 4 <= (long)d->Cd * 5 && tiles_m * t256 >= 192 && (long)d->KH * d->KW * d->Cs >= min_k) return 256;
  return 0;
}

#ifdef DY_V4_DIAG_BUILD
static
#endif
bool dy_conv_v4_eligible(const dy_conv_desc* d, int mode) {
  static const bool off = dy_env("DY_NO_CONV_V4") != nullptr;
  return !off && v4_variant(d, mode) != 0;
}

// Tile quantisation: T tiles on the chip's CUs (one block per CU) run in ceil(T / CUs) rounds, and a last round that is mostly empty
// costs as much as a full one (256->256 at 40x40, batch 64: 400 tiles = 1.56 rounds -> 78 % at best).  The whole rounds keep 256-row
// tiles; the rows of the last partial round are re-cut into tail tiles of 192 or 128 rows when that lets the round finish sooner.  The
// relative cost of a tail tile (its B half of the K loop does not shrink) is measured, not guessed: tools/conv_bench, DESIGN.md section 4.
static int v4_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  }
  return n;
}

static void v4_split(long M, int tiles_n, int& n_full, int& rb_tail, int& nblk) {
  const int cus = v4_cus();
  const long tiles_m = (M + v4::BM - 1) / v4::BM;
  const long T = tiles_m * tiles_n;
  n_full = (int)T;
  rb_tail = 4;
  nblk = (int)T;
  static const char* force = dy_env("DY_V4_TAIL");          // diagnostics builds: 0 = off, 2 / 3 = force that tail height
  if (force && atoi(force) == 0) return;
  if (T <= cus || T % cus == 0) return;
  const long full_blocks = (T / cus) * cus;
  const long full_m = full_blocks / tiles_n;                 // M-tiles of the whole rounds
  if (full_m * tiles_n != full_blocks) return;
  const long rows_left = M - full_m * v4::BM;
  // cost of one round of tail tiles relative to a round of full tiles, measured (tools/gpu/v4_tail.sh, round 3: 256->256 3x3 at 40x40:
  // 192-row round 62 us against 65.5; 128-row rounds 45 us; 512->512: 0.875 / 0.78).  The K loop is bound by LDS-DMA bytes, and a tail
  // tile only sheds rows of the A half-tiles: (16 + 16 * RB / 4) / 32 of the bytes -> 0.875 / 0.75, which is what the clock shows.
  const double cost[5] = {0, 0, 0.75, 0.92, 1.0};
  double best = 1e9;
  int best_rb = 4;
  for (int rb = 4; rb >= 2; --rb) {
    if (force && atoi(force) != rb && rb != 4) continue;
    const long t = ((rows_left + 64 * rb - 1) / (64 * rb)) * tiles_n;
    const double c = (double)((t + cus - 1) / cus) * cost[rb];
    if (c < best - 1e-9) { best = c; best_rb = rb; }
  }
  if (force && atoi(force) >= 2 && atoi(force) <= 3) best_rb = atoi(force);
  if (best_rb == 4) return;
  n_full = (int)full_blocks;
  rb_tail = best_rb;
  nblk = (int)(full_blocks + ((rows_left + 64 * best_rb - 1) / (64 * best_rb)) * tiles_n);
}

template <int ABL, typename T = bf16_t>
static int v4_launch(const dy_conv_desc* d, int mode, void* stream, int force_variant = 0, const dy_conv_desc* classes = nullptr, int ncls = 0) {
  const int variant = force_variant ? force_variant : (ncls > 1 ? 256 : v4_variant(d, mode));
  DY_CHECK(variant != 0, "conv_v4: problem not eligible");
  DY_CHECK((long)d->N * d->Hd * d->Wd < (1L << 31), "conv_v4: too many output pixels");
  v4::P p;
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
  p.nk = d->KH * d->KW * d->Cs / v4::BK;
  p.dst_row = d->dst_row_stride;
  p.dst_img = d->dst_img_stride ? d->dst_img_stride : (long)d->Hd * d->dst_row_stride;
  p.tiles_n = dy_cdiv(d->Cd, variant);
  v4_split(p.M, p.tiles_n, p.n_full, p.rb_tail, p.nblk);
  p.ncls = 0;
  if (ncls > 1) {
    DY_CHECK(ncls <= 4 && mode == 0, "conv_v4: at most 4 forward-style classes");
    p.ncls = ncls;
    int slot = 0;
    for (int c = 0; c < ncls; ++c) {
      const dy_conv_desc& q = classes[c];
      DyParityCls& k = p.cls[c];
      k.dst = (char*)q.dst; k.M = (long)q.N * q.Hd * q.Wd; k.Hd = q.Hd; k.Wd = q.Wd; k.KH = q.KH; k.KW = q.KW; k.pad = q.pad;
      k.kh0 = q.kh0; k.kw0 = q.kw0; k.Ktot = q.KH * q.KW * q.Cs;
      DY_CHECK(k.M < (1L << 31), "conv_v4: too many output pixels");
      const int tiles = (int)dy_cdiv(k.M, (long)v4::BM) * p.tiles_n;
      k.blk0 = slot;
      k._r = dy_cdiv(tiles, 8);
      slot += k._r;
    }
    p.nblk = 8 * slot;
    p.n_full = p.nblk; p.rb_tail = 4;
  }
  constexpr int EPI256 = dy_epi::row_image_bytes<256, 256>();
  constexpr int SH256 = 2 * v4::BUF > EPI256 ? 2 * v4::BUF : EPI256;
  static_assert(SH256 <= 160 * 1024, "LDS budget");
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&v4::conv_kernel<ABL, T>), hipFuncAttributeMaxDynamicSharedMemorySize, SH256);
    if (e != hipSuccess) {
      dy_set_error("conv_v4: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 3;
    }
    configured = true;
  }
  dy_note_kernel("v4::conv_kernel");
  v4::conv_kernel<ABL, T><<<p.nblk, 512, SH256, (hipStream_t)stream>>>(p);
  DY_LAUNCH_CHECK();
  return 0;
}

#ifndef DY_V4_DIAG_BUILD
int dy_conv_v4_launch(const dy_conv_desc* d, int mode, void* stream) {
  return d->dtype == DY_F16 ? v4_launch<0, f16_t>(d, mode, stream) : v4_launch<0, bf16_t>(d, mode, stream);
}

// The parity classes of a stride-2 data gradient (conv.hip: dgrad_dispatch; heaviest class first) as one launch: same dz, same weight
// pack, same channel counts; taken when the 256-wide channel tiles fit, the heaviest class has a K loop worth a 256 x 256 tile and the
// classes together fill the chip.  (DY_V4_CLASSES=0 in a DIAG build: one launch per class.)
bool dy_conv_v4_classes_eligible(const dy_conv_desc* c, int ncls) {
  static const bool off = dy_env("DY_NO_CONV_V4") != nullptr || (dy_env("DY_V4_CLASSES") && atoi(dy_env("DY_V4_CLASSES")) == 0);
  if (off || ncls < 2 || ncls > 4) return false;
  const dy_conv_desc* d = &c[0];
  if (d->dtype != DY_BF16 && d->dtype != DY_F16) return false;
  if (!(d->Cs % 64 == 0 && d->KHf > 0 && d->KHf * d->KWf <= 25 && d->stride == 1)) return false;
  const long src_bytes = (((long)d->N * d->Hs * d->Ws - 1) * d->src_ld + d->Cs) * 2;
  const long w_bytes = (long)d->Cd * d->KHf * d->KWf * d->Cs * 2;
  const long halo = ((long)d->KHf * d->Ws + d->KWf) * d->src_ld * 2;
  if (!(src_bytes + halo <= 0x7fffffffL && w_bytes <= 0x3fffffffL)) return false;
  const long t256 = (d->Cd + 255) / 256;
  if (!(d->Cd >= 192 && t256 * 256 * 4 <= (long)d->Cd * 5)) return false;
  long tiles = 0, kmax = 0;
  for (int i = 0; i < ncls; ++i) {
    const dy_conv_desc& q = c[i];
    if (q.src != d->src || q.w != d->w || q.Cs != d->Cs || q.Cd != d->Cd || q.dtype != d->dtype || q.stride != 1 || q.dil != 1 || q.KHf != d->KHf ||
        q.KWf != d->KWf || q.kh_step != d->kh_step || q.kw_step != d->kw_step || q.dst_ld != d->dst_ld || q.dst_row_stride != d->dst_row_stride ||
        q.dst_img_stride != d->dst_img_stride || q.src_ld != d->src_ld || q.accumulate != d->accumulate || q.scale || q.shift || q.stats ||
        q.act != DY_ACT_NONE)
      return false;
    if ((q.dst_ld * 2) % 16 != 0 || ((uintptr_t)q.dst) % 16 != 0 || (q.src_ld * 2) % 16 != 0) return false;
    tiles += (((long)q.N * q.Hd * q.Wd + 255) / 256) * t256;
    const long k = (long)q.KH * q.KW * q.Cs;
    kmax = k > kmax ? k : kmax;
  }
  return tiles >= 192 && kmax >= 512;
}

int dy_conv_v4_launch_classes(const dy_conv_desc* c, int ncls, void* stream) {
  return c[0].dtype == DY_F16 ? v4_launch<0, f16_t>(&c[0], 0, stream, 0, c, ncls) : v4_launch<0, bf16_t>(&c[0], 0, stream, 0, c, ncls);
}
#endif
