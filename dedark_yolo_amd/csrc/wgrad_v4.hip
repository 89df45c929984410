// Phase-interleaved weight gradient for the wide bf16 layers (>= 192 output channels): nn.Conv2d weight.grad of Conv / Bottleneck /
// Detect (ultralytics/nn/modules/conv.py:38-55, block.py:553-565, head.py:40-46).
//
//   gw[co][kh][kw][ci] = sum over output pixels m of  dz[m][co] * x[gather(m, kh, kw)][ci]
//
// GEMM view: D[k'][co] = sum_m X[m][k'] * dZ[m][co],  k' = (kh*KW + kw)*Cin + ci: the reduction runs over PIXELS.  Same structure as
// conv_v4.hip (256 x 256 block tile, 8 waves = two groups one barrier apart, a K-step of 64 pixels = 4 phases of 16 x
// v_mfma_f32_16x16x32_bf16 and one 16 KiB unit of LDS DMA each, counted vmcnt(6) once per K-step, never drained in the loop) with
// the operand images of wgrad_v2.hip: both operands sit in HBM pixel-major, so a unit is a [64 pixels][128 channels] image with
// 256-byte rows (16-byte chunk c of row r at slot c ^ (((r&3)<<2) | ((r>>2)&3))) and the fragments are read with
// ds_read_b64_tr_b16 (the CDNA4 transposing LDS read: 4 pixels x 16 channels -> each lane the 4 pixels of ITS channel), conflict-free
// for the 16x16x32 operand (the two 16-lane groups of a half read blocks 8 pixels apart).
//   * every lane's 16-byte chunk belongs to ONE (tap, channel) for the whole kernel; per K-step a lane only advances the (row, column)
//     of its two pixel rows (mixed-radix add of 64 pixels, no divisions; the byte offset moves by three wave-uniform deltas selected by
//     the carries) and tests the tap against the image border; padding taps and everything beyond M use an offset outside the buffer
//     descriptor (the hardware writes zeros).  ALL of that arithmetic runs in the middle of the wave's own MFMA clusters and leaves four
//     ready DMA offsets in registers: the load part of a phase -- the part the partner wave group's cluster waits for -- issues its
//     LDS-DMA pieces without a single VALU instruction (the dz pieces carry the K-step in their descriptor's base address);
//   * the pixel range is split over blocks (one round of blocks over the chip); consecutive logical blocks = the k' x co tiles of ONE
//     pixel range and the XCD-aware block order keeps them on one XCD: dz and x of that range are fetched from HBM once and shared
//     through that XCD's L2 by its tiles;
//   * partial tiles go to `scratch` co-major ([co][k'], 16-byte stores along k') and a second kernel adds them in a fixed order
//     (deterministic) into the OIHW f32 master gradient: one thread per (co, ci) reads k' coalesced and writes its KH*KW taps
//     contiguously.
#include <stdlib.h>
#include <type_traits>
#include "dy_common.h"
#include "../../include/dedark_yolo.h"

namespace wg4 {

constexpr int BP = 256, BQ = 256, BKP = 64;
constexpr int UNIT = 64 * 256;                       // one image: 64 pixels x 128 channels bf16
constexpr int OFF_A0 = 0, OFF_A1 = UNIT, OFF_B0 = 2 * UNIT, OFF_B1 = 3 * UNIT, BUF = 4 * UNIT;
constexpr unsigned A_OOB = 0x80000000u;              // operand extents <= 2 GiB (checked by the launcher)
constexpr unsigned B_COL_OOB = 0x40000000u;          // dz extent <= 1 GiB: invalid column + any pixel offset stays out of range

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

struct P {
  const char* x;
  const char* dz;
  unsigned x_bytes, dz_bytes;
  long x_ld, dz_ld;
  int N, Hi, Wi, Cin;           // Cin = padded input channels (k' = tap*Cin + ci)
  int Ho, Wo, Cout;             // Cout = padded output channels
  int KH, KW, stride, pad, dil;
  float* part;                  // [splits][tiles][BQ co][BP k']
  long M;
  int Ktot, tiles_q, tiles, nblk;
  long chunk;                   // pixels per split (multiple of 64)
  int q64_w, r64_w;             // 64 = q64_w * Wo + r64_w
  int q_h, r_h;                 // q64_w = q_h * Ho + r_h
};

__device__ inline int xcd_remap(int bid, int nblk) {
  int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

// POINTWISE: 1x1 / stride 1 / no padding: x rows are the output pixels themselves (no gather arithmetic at all).
template <bool POINTWISE, typename T = bf16_t>
__global__ __launch_bounds__(512) void wgrad_kernel(const P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int bid = xcd_remap(blockIdx.x, p.nblk);           // logical id = split * tiles + tile
  const int split = bid / p.tiles, tile = bid - split * p.tiles;
  const int tile_p = tile / p.tiles_q, tile_q = tile - tile_p * p.tiles_q;
  const int kp0 = tile_p * BP, q0 = tile_q * BQ;
  // the upper 128 output channels of the tile do not exist (Cout <= 128 layers): their dz unit is still issued (out of range: zeros,
  // no fetch -- the vmcnt bookkeeping stays the same) but neither read nor multiplied nor stored
  const bool has_b1 = q0 + 128 < p.Cout;
  const bool has_a1 = kp0 + 128 < p.Ktot;                   // likewise the upper 128 k' of the last k' tile (K = 1152: 4.5 tiles)
  const long m_begin = (long)split * p.chunk;
  const long m_end = (m_begin + p.chunk < p.M) ? m_begin + p.chunk : p.M;
  const int nk = m_begin < m_end ? (int)((m_end - m_begin + BKP - 1) / BKP) : 0;

  // records of the two descriptors: 0 behind the block's last K-step (every lane out of range: zeros land in a unit nobody reads)
  unsigned nrec_a = nk > 0 ? p.x_bytes : 0, nrec_b = nk > 0 ? p.dz_bytes : 0;

  // ---- DMA bookkeeping: wave instruction idx = wave + 8j (j = 0, 1) fills rows 4*idx .. 4*idx+3 of a unit; lane -> (row, slot)
  const int lrow = lane >> 4, slot = lane & 15;
  const int chunk = slot ^ ((lrow << 2) | (wave & 3));       // (idx & 3) == (wave & 3) for both j
  // x side: this lane's chunk of unit u is channel block (tap, ci) of k' = kp0 + 128u + 8*chunk
  int a_dh[2], a_dw[2];
  unsigned a_tap[2];             // byte offset of (tap, ci) relative to the output pixel's source position
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k = kp0 + 128 * u + 8 * chunk;
    const bool kok = k < p.Ktot;
    const int kk = kok ? k : 0;
    const int tap = kk / p.Cin, ci = kk - tap * p.Cin;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    a_dh[u] = kh * p.dil - p.pad;
    a_dw[u] = kw * p.dil - p.pad;
    a_tap[u] = (unsigned)(((long)a_dh[u] * p.Wi + a_dw[u]) * p.x_ld * 2 + ci * 2);
    if (!kok) a_dh[u] = 0x40000000;           // k' beyond K: the row test below fails for every pixel
  }
  // pixel rows of this lane: m_begin + 4*(wave + 8j) + lrow (+ 64 per K-step); kept as SOURCE coordinates (output row / column times
  // the stride) so that the border test needs no multiplication
  int ph[2], pw[2];
  unsigned a_pix[2];             // byte offset of the source position of the row's output pixel at tap (0,0) without padding shift
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long m = m_begin + 4 * (wave + 8 * j) + lrow;
    if (POINTWISE) {
      a_pix[j] = (unsigned)(m * p.x_ld * 2);
      ph[j] = pw[j] = 0;
    } else {
      const unsigned mu = (unsigned)m;
      const unsigned HWo = (unsigned)(p.Ho * p.Wo);
      const int pn = (int)(mu / HWo);
      const unsigned rem = mu - (unsigned)pn * HWo;
      const int oh = (int)(rem / (unsigned)p.Wo), ow = (int)(rem - (unsigned)oh * (unsigned)p.Wo);
      ph[j] = oh * p.stride;
      pw[j] = ow * p.stride;
      a_pix[j] = (unsigned)((((long)pn * p.Hi + ph[j]) * p.Wi + pw[j]) * p.x_ld * 2);
    }
  }
  // dz side: row m, channels q0 + 128u + 8*chunk
  unsigned b_off[2][2];          // [u][j]
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int co = q0 + 128 * u + 8 * chunk;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      b_off[u][j] = co < p.Cout ? (unsigned)((long)(4 * (wave + 8 * j) + lrow) * p.dz_ld * 2 + co * 2) : B_COL_OOB;
  }
  // K-step being issued (wave-uniform): the dz descriptor's BASE moves with it and its records shrink by as much, so that the rows
  // beyond M of the last K-step are out of range by their VGPR offset alone (an SGPR offset is not part of the hardware's range check)
  int b_koff = nk > 0 ? (int)(m_begin * p.dz_ld * 2) : 0;
  if (nk > 0) nrec_b -= (unsigned)b_koff;
  int sk = 0;
  auto advance_scalar = [&]() {
    ++sk;
    b_koff += (int)(BKP * p.dz_ld * 2);
    const unsigned step = (unsigned)(BKP * p.dz_ld * 2);
    nrec_b = nrec_b > step ? nrec_b - step : 0;
    if (sk >= nk) { nrec_a = 0; nrec_b = 0; b_koff = 0; }
  };
  // + 64 pixels: mixed-radix add on (row, column) with the carries selecting the byte delta.  offset(n, h, w) = ((n*Hi + h*s)*Wi + w*s)*ld:
  // per output column e_w = s*ld, per output row e_h = s*Wi*ld, per image e_n = Hi*Wi*ld
  const int e_w = p.stride * (int)p.x_ld * 2, e_h = p.stride * p.Wi * (int)p.x_ld * 2, e_n = p.Hi * p.Wi * (int)p.x_ld * 2;
  const int d_base = p.q_h * e_n + p.r_h * e_h + p.r64_w * e_w, d_cw = e_h - p.Wo * e_w, d_ch = e_n - p.Ho * e_h;
  const int s_rw = p.r64_w * p.stride, s_rh = p.r_h * p.stride, s_Wo = p.Wo * p.stride, s_Ho = p.Ho * p.stride;
  auto advance_rows = [&]() {
    if (POINTWISE) {
#pragma unroll
      for (int j = 0; j < 2; ++j) a_pix[j] += (unsigned)(BKP * p.x_ld * 2);
      return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {                            // (selects, no branches: this sits between MFMAs)
      int w = pw[j] + s_rw, h = ph[j] + s_rh;
      const bool cw = w >= s_Wo;
      w -= cw ? s_Wo : 0;
      h += cw ? p.stride : 0;
      int d = d_base + (cw ? d_cw : 0);
      const bool c1 = h >= s_Ho;
      h -= c1 ? s_Ho : 0;
      d += c1 ? d_ch : 0;
      const bool c2 = h >= s_Ho;                             // r_h + carry can pass Ho once more only when r_h == Ho - 1 and carry
      h -= c2 ? s_Ho : 0;
      d += c2 ? d_ch : 0;
      pw[j] = w; ph[j] = h;
      a_pix[j] += (unsigned)d;
    }
  };
  // the ready DMA offsets of unit u for the lane's two rows: ONE unconditional DMA per lane with a selected offset (an exec-masked DMA
  // would leave stale bytes in LDS instead of zeros)
  unsigned a_v[2][2];            // [u][j]
  auto offsets_of = [&](int u) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bool ok = a_dh[u] < 0x40000000;
      if (!POINTWISE) {
        const int ih = ph[j] + a_dh[u], iw = pw[j] + a_dw[u];
        ok = ((unsigned)ih < (unsigned)p.Hi) & ((unsigned)iw < (unsigned)p.Wi);
      }
      a_v[u][j] = ok ? a_pix[j] + a_tap[u] : A_OOB;
    }
  };
  auto stage_a = [&](int buf, int u) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, nrec_a, 0x00020000);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + buf * BUF + (u ? OFF_A1 : OFF_A0) + (wave + 8 * j) * 1024), 16,
                                               (int)a_v[u][j], 0, 0, 0);
  };
  auto stage_b = [&](int buf, int u) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dz + b_koff), 0, nrec_b, 0x00020000);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + buf * BUF + (u ? OFF_B1 : OFF_B0) + (wave + 8 * j) * 1024), 16,
                                               (int)b_off[u][j], 0, 0, 0);
  };

  // ---- transposed-read addresses.  16-lane group gg = lane>>4 reads pixels 8*gg + 4*t + (0..3) (t = 0, 1) of a 32-pixel block;
  // lane 4q+p of the group addresses pixel row q, channels 4p .. 4p+3 of the 16-channel tile.
  const int gg = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  int a_adr[2][4], b_adr[2][2];                              // [t][16-channel tile of the wave]
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = 8 * gg + 4 * t + tq;                     // + 32 per k block: the key below does not change
    const int key = (tq << 2) | ((2 * gg + t) & 3);
    const int base = row * 256 + 8 * (tp & 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) a_adr[t][i] = base + 16 * ((2 * (4 * wr + i) + (tp >> 1)) ^ key);
#pragma unroll
    for (int j = 0; j < 2; ++j) b_adr[t][j] = base + 16 * ((2 * (2 * wc + j) + (tp >> 1)) ^ key);
  }

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 afr[4][2], bfr[2][2][2];

  // Transposing reads as inline asm: hipcc (ROCm 7.2) orders the ds_read_tr16 builtin behind every pending LDS DMA with a
  // s_waitcnt vmcnt(0) at the head of each K-step, which would drain the pipeline this kernel is built around.  The counts are ours:
  // lgkmcnt(8) in P1, lgkmcnt(0) + sched_barrier in front of every MFMA cluster.
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  unsigned a_vad[2][4], b_vad[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a_vad[t][i] = lds0 + (unsigned)a_adr[t][i];
#pragma unroll
    for (int j = 0; j < 2; ++j) b_vad[t][j] = lds0 + (unsigned)b_adr[t][j];
  }
#define WG4_TR(dst, vaddr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(vaddr), "n"(OFF) : "memory")
#define WG4_FRAG(fr, vad0, vad1, OFF)                      \
  do {                                                      \
    uint2 lo_, hi_;                                         \
    WG4_TR(lo_, vad0, OFF);                                 \
    WG4_TR(hi_, vad1, OFF);                                 \
    fr = u32x4{lo_.x, lo_.y, hi_.x, hi_.y};                 \
  } while (0)
#define WG4_READ_A2(BASE, I0)                                                         \
  do {                                                                                \
    WG4_FRAG(afr[I0][0], a_vad[0][I0], a_vad[1][I0], (BASE));                          \
    WG4_FRAG(afr[I0][1], a_vad[0][I0], a_vad[1][I0], (BASE) + 8192);                   \
    WG4_FRAG(afr[I0 + 1][0], a_vad[0][I0 + 1], a_vad[1][I0 + 1], (BASE));              \
    WG4_FRAG(afr[I0 + 1][1], a_vad[0][I0 + 1], a_vad[1][I0 + 1], (BASE) + 8192);       \
  } while (0)
#define WG4_READ_B(BASE, U)                                                           \
  do {                                                                                \
    WG4_FRAG(bfr[U][0][0], b_vad[0][0], b_vad[1][0], (BASE));                          \
    WG4_FRAG(bfr[U][0][1], b_vad[0][0], b_vad[1][0], (BASE) + 8192);                   \
    WG4_FRAG(bfr[U][1][0], b_vad[0][1], b_vad[1][1], (BASE));                          \
    WG4_FRAG(bfr[U][1][1], b_vad[0][1], b_vad[1][1], (BASE) + 8192);                   \
  } while (0)
  // `mid` runs between the two k-halves of the cluster: the K-walk arithmetic of the phases to come, whose VALU instructions then issue
  // in the gaps of this wave's own MFMAs instead of in a load part (`live` = the tile half exists; the bookkeeping runs either way)
  // LIVE = the tile half exists (the bookkeeping runs either way); GAP = VALU instructions of `mid` the scheduler places behind each MFMA
  auto mma = [&](int ah, int bh, auto livec, auto gapc, auto mid) {
    constexpr bool LIVE = decltype(livec)::value;
    constexpr int GAP = decltype(gapc)::value;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    if constexpr (LIVE) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[ah][bh][i][j] = mfma_16x16x32<T>(afr[i][kb], bfr[bh][j][kb], acc[ah][bh][i][j]);
        if (kb == 0) mid();
      }
      if constexpr (GAP > 0) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA ...
          __builtin_amdgcn_sched_group_barrier(0x002, GAP, 0);    // ... then GAP VALU of the bookkeeping in its shadow
        }
      }
    } else {
      mid();
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // pins the K-walk state behind the point where it is called: without it the compiler starts the arithmetic of a `mid` block early,
  // in the load part in front of the cluster's barrier
  auto pin_rows = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(pw[j]), "+v"(ph[j]), "+v"(a_pix[j]));
  };
  auto pin_offsets = [&](int u) {                            // ... and its results in front of the point where they are first used
#pragma unroll
    for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(a_v[u][j]));
  };

  // ---- prologue: K-step 0 complete + three units of K-step 1 in flight
  offsets_of(0); offsets_of(1);
  stage_b(0, 0); stage_a(0, 0); stage_b(0, 1); stage_a(0, 1);
  advance_scalar();
  advance_rows();
  offsets_of(0); offsets_of(1);                         // rows of K-step 1: unit 0 goes out here, unit 1 in P1 of the first K-step
  stage_b(1, 0); stage_a(1, 0); stage_b(1, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();            // group 1 runs one barrier behind group 0

  // One K-step on buffer `cur` (0 / 1).  The read addresses carry the buffer base (the ds offset field is 16 bits) and are
  // flipped in the last cluster of every K-step; DMA destinations are scalar.
  int cur = 0;
  using yes = std::true_type;
  auto kloop = [&](auto a1c, auto b1c) {
  constexpr bool A1 = decltype(a1c)::value, B1 = decltype(b1c)::value;
  for (int kt = 0; kt < nk; ++kt) {
    const int b = cur, bo = cur ^ 1;
    // P1: B unit 0 (8 transposing reads, retired before the barrier), A unit 0 (16); refill A unit 1 of the next K-step
    // (lgkmcnt is a 4-bit counter: the B reads are waited for with 8 A reads behind them, the other 8 A reads follow)
    WG4_READ_B(OFF_B0, 0);
    WG4_READ_A2(OFF_A0, 0);
    __builtin_amdgcn_sched_barrier(0);
    stage_a(bo, 1);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    WG4_READ_A2(OFF_A0, 2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    mma(0, 0, yes{}, std::integral_constant<int, 0>{}, [&]() { advance_scalar(); });
    __builtin_amdgcn_s_barrier();
    // P2: B unit 1; refill B unit 0 two K-steps ahead; in the cluster the rows move on by 64 pixels (K-step kt + 2) and unit 0's
    // offsets for P3 are made
    if (B1) WG4_READ_B(OFF_B1, 1);
    __builtin_amdgcn_sched_barrier(0);
    stage_b(b, 0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    mma(0, 1, std::bool_constant<B1>{}, std::integral_constant<int, POINTWISE ? 1 : 3>{},
        [&]() { pin_rows(); advance_rows(); offsets_of(0); pin_rows(); pin_offsets(0); });
    __builtin_amdgcn_s_barrier();
    // P3: A unit 1; refill A unit 0; in the cluster unit 1's offsets for P1 of the next K-step
    if (A1) {
      WG4_READ_A2(OFF_A1, 0);
      WG4_READ_A2(OFF_A1, 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    stage_a(b, 0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    mma(1, 1, std::bool_constant<A1 && B1>{}, std::integral_constant<int, 1>{}, [&]() { pin_rows(); offsets_of(1); pin_offsets(1); });
    __builtin_amdgcn_s_barrier();
    // P4: refill B unit 1; everything but the three youngest units has landed; in the cluster the read addresses flip to the other buffer
    stage_b(b, 1);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
      const int delta = cur ? -BUF : BUF;
      auto mid = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { a_vad[t][i] += delta; asm volatile("" : "+v"(a_vad[t][i])); }
#pragma unroll
          for (int j = 0; j < 2; ++j) { b_vad[t][j] += delta; asm volatile("" : "+v"(b_vad[t][j])); }
        }
      };
      mma(1, 0, std::bool_constant<A1>{}, std::integral_constant<int, 1>{}, mid);
    }
    __builtin_amdgcn_s_barrier();
    cur ^= 1;
  }
  };
  // (block-uniform: one specialisation of the loop per combination of existing tile halves, so that a cluster is one basic block)
  if (has_a1) {
    if (has_b1) kloop(std::true_type{}, std::true_type{});
    else kloop(std::true_type{}, std::false_type{});
  } else {
    if (has_b1) kloop(std::false_type{}, std::true_type{});
    else kloop(std::false_type{}, std::false_type{});
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();

  // ---- partial tile -> scratch, co-major: part[bid][co][k'], a lane's 4 consecutive k' (accumulator registers) = one 16-byte store
  float* out = p.part + (long)bid * (BP * BQ);
  const int cl = lane & 15, g = lane >> 4;
#pragma unroll
  for (int bh = 0; bh < 2; ++bh)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (bh == 1 && !has_b1) continue;                     // (the reduce kernel never reads channels >= Cout)
      const int co = 128 * bh + 32 * wc + 16 * j + cl;
#pragma unroll
      for (int ah = 0; ah < 2; ++ah)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (ah == 1 && !has_a1) continue;
          const int k = 128 * ah + 64 * wr + 16 * i + 4 * g;
          *reinterpret_cast<f32x4*>(out + (long)co * BP + k) = acc[ah][bh][i][j];
        }
    }
}

// g[co][ci][kh][kw] = sum over splits of part[split][tile(k', co)][co % 256][k' % 256].  One thread per (co, ci, tap): it adds the
// splits in a fixed order (k' = tap*Cin_pad + ci is contiguous over the threads of a block: coalesced reads).  The tap is a grid
// dimension (one thread per (co, ci) walking its taps left the chip at 4 waves per CU: 21 us for 66 MB of slabs).
__global__ __launch_bounds__(256) void reduce_kernel(const float* __restrict__ part, int splits, int tiles, int tiles_q, int Cout, int Cin,
                                                     int Cin_pad, int taps, float* __restrict__ g) {
  const int ci = blockIdx.x * 256 + threadIdx.x;
  const int co = blockIdx.y;
  const int t = blockIdx.z;
  if (ci >= Cin) return;
  const int tq = co / BQ, cq = co - tq * BQ;
  const long sstride = (long)tiles * BP * BQ;
  const int k = t * Cin_pad + ci;
  const float* src = part + ((long)((k / BP) * tiles_q + tq) * BQ + cq) * BP + (k % BP);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int s = 0;
  for (; s + 3 < splits; s += 4) {
    a0 += src[(long)s * sstride];
    a1 += src[(long)(s + 1) * sstride];
    a2 += src[(long)(s + 2) * sstride];
    a3 += src[(long)(s + 3) * sstride];
  }
  for (; s < splits; ++s) a0 += src[(long)s * sstride];
  g[((long)co * Cin + ci) * taps + t] = (a0 + a1) + (a2 + a3);
}

}  // namespace wg4

bool dy_wgrad_v4_eligible(int dtype, int Cin_pad, int Cout_pad, int KH, int KW, long M, int N, int Hi, int Wi, int Ho, int Wo, long x_ld,
                          long dz_ld, long scratch_elems) {
  static const bool off = dy_env("DY_NO_WGRAD_V4") != nullptr;
  if (off || (dtype != DY_BF16 && dtype != DY_F16)) return false;
  if (Cin_pad % 8 != 0 || Cout_pad % 8 != 0 || (x_ld * 2) % 16 != 0 || (dz_ld * 2) % 16 != 0) return false;
  const long Ktot = (long)KH * KW * Cin_pad;
  const long tq = (Cout_pad + 255) / 256, tp = (Ktot + 255) / 256;
  // 256-wide tiles must not be mostly padding (Cout >= 192 within 20 %, K within 20 %) and the pixel loop must be long enough to
  // amortise a slab -- except against the 128 x 128 kernel on long pixel loops: there even a quarter-filled tile wins (C3, B = 64:
  // 256->64 3x3 at 80x80 672 -> 365 us, 512->64 at 40x40 338 -> 196, 320->128 1x1 at 160x160 436 -> 321, 64->128 3x3 s2 at 320x320
  // 829 -> 521), so from 65,536 pixels on any Cout >= 64 and K >= 192 is taken
  const bool tight = Cout_pad >= 192 && tq * 256 * 4 <= (long)Cout_pad * 5 && tp * 256 * 4 <= Ktot * 5;
  // ... as long as the tile is at least a quarter full (32->64 3x3 s2 at 160x160, B = 32: K = 288, 14 % full, 48 -> 135 us: not taken)
  const bool long_loop = Cout_pad >= 64 && M >= 65536 && Ktot * Cout_pad * 4 >= tp * 256 * tq * 256;
  if (!(Ktot >= 192 && (tight || long_loop))) return false;
  if (M < 16384 || M >= (1L << 31)) return false;
  const long x_bytes = (((long)N * Hi * Wi - 1) * x_ld + Cin_pad) * 2, dz_bytes = ((M - 1) * dz_ld + Cout_pad) * 2;
  if (x_bytes > 0x7fffffffL || dz_bytes > 0x3fffffffL) return false;
  return scratch_elems >= tp * tq * 65536L * 2;        // at least two splits' worth of slabs
}

int dy_wgrad_v4_launch(const void* x, long x_ld, int N, int Hi, int Wi, int Cin_pad, const void* dz, long dz_ld, int Ho, int Wo,
                       int Cout_pad, int KH, int KW, int stride, int pad, int dil, int Cout, int Cin, float* scratch,
                       long scratch_elems, float* g_oihw, int dtype, void* stream) {
  using namespace wg4;
  P p;
  p.x = (const char*)x; p.dz = (const char*)dz; p.x_ld = x_ld; p.dz_ld = dz_ld;
  p.N = N; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin_pad; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout_pad;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil;
  p.M = (long)N * Ho * Wo;
  p.Ktot = KH * KW * Cin_pad;
  p.x_bytes = (unsigned)((((long)N * Hi * Wi - 1) * x_ld + Cin_pad) * 2);
  p.dz_bytes = (unsigned)(((p.M - 1) * dz_ld + Cout_pad) * 2);
  p.part = scratch;
  p.tiles_q = dy_cdiv(Cout_pad, BQ);
  p.tiles = dy_cdiv(p.Ktot, BP) * p.tiles_q;
  // one round of blocks over the 256 CUs, at least 8 K-steps each, slabs must fit the workspace
  long splits = 256 / p.tiles;
  const long max_splits = (p.M + 8L * BKP - 1) / (8L * BKP);
  if (splits > max_splits) splits = max_splits;
  const long fit = scratch_elems / ((long)p.tiles * BP * BQ);
  if (splits > fit) splits = fit;
  if (splits < 1) splits = 1;
  long chunk = (p.M + splits - 1) / splits;
  chunk = (chunk + BKP - 1) / BKP * BKP;
  splits = (p.M + chunk - 1) / chunk;
  p.chunk = chunk;
  p.nblk = (int)(splits * p.tiles);
  p.q64_w = BKP / Wo; p.r64_w = BKP % Wo;
  p.q_h = p.q64_w / Ho; p.r_h = p.q64_w % Ho;
  const bool pointwise = KH == 1 && KW == 1 && stride == 1 && pad == 0;
  constexpr int SHMEM = 2 * BUF;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<false, bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<true, bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<false, f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<true, f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM);
    if (e != hipSuccess) {
      dy_set_error("wgrad_v4: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 3;
    }
    configured = true;
  }
  hipStream_t st = (hipStream_t)stream;
  dy_note_kernel("wg4::wgrad_kernel+reduce_kernel");
  if (dtype == DY_F16) {
    if (pointwise) wgrad_kernel<true, f16_t><<<p.nblk, 512, SHMEM, st>>>(p);
    else wgrad_kernel<false, f16_t><<<p.nblk, 512, SHMEM, st>>>(p);
  } else {
    if (pointwise) wgrad_kernel<true, bf16_t><<<p.nblk, 512, SHMEM, st>>>(p);
    else wgrad_kernel<false, bf16_t><<<p.nblk, 512, SHMEM, st>>>(p);
  }
  DY_LAUNCH_CHECK();
  reduce_kernel<<<dim3(dy_cdiv(Cin, 256), Cout, KH * KW), 256, 0, st>>>(scratch, (int)splits, p.tiles, p.tiles_q, Cout, Cin, Cin_pad, KH * KW, g_oihw);
  DY_LAUNCH_CHECK();
  return 0;
}
