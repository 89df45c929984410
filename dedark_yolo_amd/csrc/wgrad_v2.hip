// Pipelined weight-gradient kernel for the bf16 layers (nn.Conv2d weight.grad of ultralytics/nn/modules/conv.py:38-55).
//
//   gw[co][kh][kw][ci] = sum over output pixels m of  dz[m][co] * x[gather(m, kh, kw)][ci]
//
// GEMM view: D[k'][co] = sum_m  X[m][k'] * dZ[m][co],  k' = (kh*KW + kw)*Cin + ci.  The reduction runs over PIXELS, but both
// operands sit in HBM pixel-major (NHWC): an MFMA lane needs 8 consecutive pixels of one channel.  The register-staged kernel
// of conv.hip transposes 8x8 blocks with v_perm before writing LDS (64 perm + 16 ds_write per unit, two barriers per 64-pixel
// step, one LDS buffer) and reaches 10-13 % of the MFMA peak.  Here:
//   * tiles go global -> LDS untransposed with global_load_lds (16 B per lane, the source address does the im2col gather,
//     padding reads a zero page): image = [64 pixels][128 channels] bf16, 256-byte rows, 16-byte chunk c of row r stored at
//     slot c ^ (((r&3)<<2) | ((r>>2)&3));
//   * fragments are read with ds_read_b64_tr_b16 (CDNA4 transposing LDS read: a 16-lane group fetches 4 pixels x 16 channels
//     and each lane receives the 4 pixels of ITS channel), two reads = one 32x32x16 operand; the swizzle above is
//     conflict-free for these reads;
//   * LDS ring with one raw barrier per step (as conv_v2.hip); shipped shape: 128 (k') x 128 (co) block tile on 4 waves, 2 stages =
//     64 KiB so that two blocks share a CU (the 256 x 128 x 3-stage single-block shape is DY_WG2_EXP=1); 16 MFMA per wave and step;
//   * the pixel range is split over gridDim.y; partial tiles go to `scratch` and a second kernel adds them in a fixed order
//     (deterministic) while scattering to the OIHW f32 master-gradient layout.
#include <stdlib.h>
#include "dy_common.h"
#include "../../include/dedark_yolo.h"

namespace wg2 {

constexpr int BKP = 64;
constexpr int IMG = BKP * 256;                 // one [64][128ch] image: 16 KiB

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

__device__ __attribute__((aligned(16))) unsigned char g_zero_page[16];

struct P {
  const char* x;
  long x_ld;
  int N, Hi, Wi, Cin;          // Cin = padded input channels (k' = tap*Cin + ci)
  const char* dz;
  long dz_ld;
  int Ho, Wo, Cout;            // Cout = padded output channels
  int KH, KW, stride, pad, dil;
  float* part;                 // [splits][tiles][BP][BQ]
  long M;
  int Ktot, tiles_q;
  long chunk;                  // pixels per split (multiple of BKP)
  int pointwise;
};

// BP (k') x 128 (co) block tile on (BP/64) x 2 waves, 64 x 64 outputs per wave, NSTAGE-deep ring of (BP/128 + 1) images.
template <int BP, int BQ, int NSTAGE, typename T = bf16_t>
__global__ __launch_bounds__(BP * 2) void wgrad_kernel(P p) {
  constexpr int TM = 2, TN = BQ / 64;            // wave tile 64 (k') x BQ/2 (co)
  constexpr int NA = BP / 128;                   // x images per stage
  constexpr int NBI = BQ / 128;                  // dz images per stage
  constexpr int NW = BP / 32;                    // waves
  constexpr int NROW = 16 / NW;                  // distinct tile rows per lane inside one image (4 rows per wave instruction)
  constexpr int A_LD = NA * NROW, B_LD = NBI * NROW;   // global_load_lds per lane and stage
  constexpr int STAGE = (NA + NBI) * IMG;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile_p = blockIdx.x / p.tiles_q, tile_q = blockIdx.x - tile_p * p.tiles_q;
  const int kp0 = tile_p * BP, q0 = tile_q * BQ;
  const long m_begin = (long)blockIdx.y * p.chunk;
  const long m_end = (m_begin + p.chunk < p.M) ? m_begin + p.chunk : p.M;
  const int nsteps = m_begin < m_end ? (int)((m_end - m_begin + BKP - 1) / BKP) : 0;

  // ---- DMA bookkeeping.  Wave instruction a (0..31 for the two x images, 0..15 for dz) fills rows 4*(a&15) .. +3 of its image.
  const int lrow = lane >> 4, slot = lane & 15;
  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  int a_kh[A_LD], a_kw[A_LD];
  long a_coff[A_LD];                 // byte offset of the channel chunk inside a pixel
  bool a_ok[A_LD];
  int row_a[NROW];                   // distinct tile rows of this lane's x loads (instruction j uses row j % NROW of image j / NROW)
#pragma unroll
  for (int j = 0; j < A_LD; ++j) {
    const int sub = j / NROW, idx = wave + NW * (j % NROW);
    const int row = 4 * idx + lrow;
    if (j < NROW) row_a[j] = row;
    const int chunk = slot ^ ((lrow << 2) | (idx & 3));
    const int k = kp0 + sub * 128 + 8 * chunk;
    a_ok[j] = k < p.Ktot;
    const int kk = a_ok[j] ? k : 0;
    const int tap = kk / p.Cin, ci = kk - tap * p.Cin;
    a_kh[j] = (tap / p.KW) * p.dil - p.pad;
    a_kw[j] = (tap - (tap / p.KW) * p.KW) * p.dil - p.pad;
    a_coff[j] = (long)ci * 2;
  }
  long b_off[B_LD];                  // byte offset of (row, co chunk) relative to dz + m_step * dz_ld * 2
  bool b_ok[B_LD];
  int row_b[B_LD];
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    const int idx = wave + NW * (j % NROW);
    row_b[j] = 4 * idx + lrow;
    const int chunk = slot ^ ((lrow << 2) | (idx & 3));
    const int co = q0 + (j / NROW) * 128 + 8 * chunk;
    b_ok[j] = co < p.Cout;
    b_off[j] = ((long)row_b[j] * p.dz_ld + (b_ok[j] ? co : 0)) * 2;
  }
  // pixel coordinates of this lane's x rows at the step being issued
  int pi[NROW], poh[NROW], pow_[NROW];
  const long HWo = (long)p.Ho * p.Wo;
#pragma unroll
  for (int r = 0; r < NROW; ++r) {
    long m = m_begin + row_a[r];
    if (m >= p.M) m = p.M - 1;                  // masked below through m_issue
    pi[r] = (int)(m / HWo);
    const int rem = (int)(m - (long)pi[r] * HWo);
    poh[r] = rem / p.Wo;
    pow_[r] = rem - poh[r] * p.Wo;
  }
  long m_issue = m_begin;                       // first pixel of the step being issued

  auto issue = [&](int buf) {
    char* stage = smem + buf * STAGE;
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      const int r = j % NROW;
      const bool live = a_ok[j] && (m_issue + row_a[r] < m_end);
      const char* g = zero;
      if (p.pointwise) {
        if (live) g = p.x + (m_issue + row_a[r]) * p.x_ld * 2 + a_coff[j];
      } else {
        const int ih = poh[r] * p.stride + a_kh[j], iw = pow_[r] * p.stride + a_kw[j];
        if (live && ih >= 0 && ih < p.Hi && iw >= 0 && iw < p.Wi)
          g = p.x + (((long)pi[r] * p.Hi + ih) * p.Wi + iw) * p.x_ld * 2 + a_coff[j];
      }
      __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(stage + (j / NROW) * IMG + (wave + NW * r) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      const bool live = b_ok[j] && (m_issue + row_b[j] < m_end);
      const char* g = live ? p.dz + m_issue * p.dz_ld * 2 + b_off[j] : zero;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(stage + (NA + j / NROW) * IMG + (wave + NW * (j % NROW)) * 1024), 16, 0, 0);
    }
    m_issue += BKP;
    if (!p.pointwise) {
#pragma unroll
      for (int r = 0; r < NROW; ++r) {
        pow_[r] += BKP;
        while (pow_[r] >= p.Wo) { pow_[r] -= p.Wo; ++poh[r]; }
        while (poh[r] >= p.Ho) { poh[r] -= p.Ho; ++pi[r]; }
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- transposed-read addresses: lane 4q+p of a 16-lane group addresses pixel row q of the 4-row block, channels 4p..4p+3
  const int g4 = lane >> 4, h = g4 >> 1, blk = g4 & 1, tq = (lane & 15) >> 2, tp = lane & 3;
  int a_adr[TM][2], b_adr[TN][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int rowoff = 256 * (8 * h + 4 * t + tq);
    const int key = (tq << 2) | (2 * h + t);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int mrow = wm * 64 + i * 32;                     // first k' of the MFMA tile inside the 256-wide block tile
      const int sub = mrow >> 7, tile32 = (mrow & 127) >> 5;
      a_adr[i][t] = sub * IMG + rowoff + 16 * ((4 * tile32 + 2 * blk + (tp >> 1)) ^ key) + 8 * (tp & 1);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int ncol = wn * (BQ / 2) + j * 32;               // first co of the MFMA tile inside the BQ-wide block tile
      const int tile32 = (ncol & 127) >> 5;
      b_adr[j][t] = (NA + (ncol >> 7)) * IMG + rowoff + 16 * ((4 * tile32 + 2 * blk + (tp >> 1)) ^ key) + 8 * (tp & 1);
    }
  }
  // MFMA tiles that lie completely in the channel padding of the block tile are skipped (wave-uniform)
  bool a_live[TM], b_live[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_live[i] = kp0 + wm * 64 + i * 32 < p.Ktot;
#pragma unroll
  for (int j = 0; j < TN; ++j) b_live[j] = q0 + wn * (BQ / 2) + j * 32 < p.Cout;

  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  if (nsteps > 0) issue(0);
  if (NSTAGE > 2 && nsteps > 1) issue(1);
  for (int s = 0; s < nsteps; ++s) {
    if (NSTAGE > 2 && s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LD + B_LD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s + NSTAGE - 1 < nsteps) issue((s + NSTAGE - 1) % NSTAGE);
    // Transposing reads as inline asm: hipcc (ROCm 7.2) orders the ds_read_tr16 builtin behind every pending LDS DMA with an
    // s_waitcnt vmcnt(0) -- right behind the issue() above, so the stage just sent out had to land before the current one was computed
    // and the ring never had anything in flight.  The counts are ours: lgkmcnt(0) in front of the MFMAs of a 16-pixel slice.
    const unsigned stage = lds0 + (unsigned)((s % NSTAGE) * STAGE);
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {                         // 16 pixels per MFMA
      uint2 alo[TM], ahi[TM], blo[TN], bhi[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(alo[i]) : "v"(stage + (unsigned)a_adr[i][0]), "n"(sl * 4096) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ahi[i]) : "v"(stage + (unsigned)a_adr[i][1]), "n"(sl * 4096) : "memory");
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(blo[j]) : "v"(stage + (unsigned)b_adr[j][0]), "n"(sl * 4096) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(bhi[j]) : "v"(stage + (unsigned)b_adr[j][1]), "n"(sl * 4096) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      u32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = u32x4{alo[i].x, alo[i].y, ahi[i].x, ahi[i].y};
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = u32x4{blo[j].x, blo[j].y, bhi[j].x, bhi[j].y};
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          if (a_live[i] && b_live[j])
            acc[i][j] = mfma_32x32x16<T>(af[i], bf[j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- partial tile -> scratch.  D layout of the 32x32 MFMA: col (co) = lane&31, row (k') = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* out = p.part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (BP * BQ);
  const int cl = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (!(a_live[i] && b_live[j])) continue;               // the reduce kernel never reads padding tiles
      const int col = wn * (BQ / 2) + j * 32 + cl;
      const int row0 = wm * 64 + i * 32 + 4 * hh;
#pragma unroll
      for (int r = 0; r < 16; ++r) out[(row0 + (r & 3) + 8 * (r >> 2)) * BQ + col] = acc[i][j][r];
    }
}

// g[co][ci][kh][kw] = sum over splits of part[split][tile(k', co)][k' % BP][co % BQ].  Block = (32 consecutive co, one k'); eight
// split-lanes walk the slabs with four loads in flight each and are added in a fixed order (deterministic).
__global__ __launch_bounds__(256) void reduce_kernel(const float* __restrict__ part, int splits, int tiles, int tiles_q, int BP, int BQ,
                                                     int Cout, int Cin, int Cin_pad, int KH, int KW, int Ktot,
                                                     float* __restrict__ g) {
  __shared__ float red[8][33];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int co = blockIdx.x * 32 + cl;
  const int k = blockIdx.y;
  const int tap = k / Cin_pad, ci = k - tap * Cin_pad;
  if (ci >= Cin) return;                                     // block-uniform
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (co < Cout) {
    const int tile = (k / BP) * tiles_q + co / BQ;
    const float* src = part + ((long)tile * BP + (k % BP)) * BQ + (co % BQ);
    const long sstride = (long)tiles * BP * BQ;
    int s = sl;
    for (; s + 24 < splits; s += 32) {
      const float v0 = src[(long)s * sstride], v1 = src[(long)(s + 8) * sstride], v2 = src[(long)(s + 16) * sstride],
                  v3 = src[(long)(s + 24) * sstride];
      a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; s < splits; s += 8) a0 += src[(long)s * sstride];
  }
  red[sl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sl == 0 && co < Cout) {
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) t += red[u][cl];
    const int kh = tap / KW, kw = tap - kh * KW;
    g[(((long)co * Cin + ci) * KH + kh) * KW + kw] = t;
  }
}

}  // namespace wg2

bool dy_wgrad_v2_eligible(int dtype, int Cin_pad, int Cout_pad, int KH, int KW, long M, long x_ld, long dz_ld) {
  static const bool off = dy_env("DY_NO_WGRAD_V2") != nullptr;
  if (off) return false;
  const long Ktot = (long)KH * KW * Cin_pad;
  // Cout <= 32 stays on the register-staged kernel (32->32 3x3 at 80x80: 59 us there, 66 us here: 3/4 of the co tile would be
  // padding); DY_WG2_NARROW=<min Cout> overrides for experiments.
  static const int exp_narrow = dy_env("DY_WG2_NARROW") ? atoi(dy_env("DY_WG2_NARROW")) : 0;
  const int min_co = exp_narrow > 0 ? exp_narrow : 64;
  return (dtype == DY_BF16 || dtype == DY_F16) && Cin_pad % 8 == 0 && Cout_pad % 8 == 0 && Cout_pad >= min_co && Ktot >= 64 && M >= 4096 && M < (1L << 31) &&
         (x_ld * 2) % 16 == 0 && (dz_ld * 2) % 16 == 0;
}

int dy_wgrad_v2_launch(const void* x, long x_ld, int N, int Hi, int Wi, int Cin_pad, const void* dz, long dz_ld, int Ho, int Wo,
                       int Cout_pad, int KH, int KW, int stride, int pad, int dil, int Cout, int Cin, float* scratch,
                       long scratch_elems, float* g_oihw, int dtype, void* stream) {
  using namespace wg2;
  // Tile choice (tools/conv_bench, B = 64).  The kernel streams its operands from L2 / Infinity Cache every step, so the tile's
  // flop-per-byte decides: 256 x 256 (128 flop/B, 128 KiB, one block per CU) when the layer has >= 256 output channels
  // (256->256 3x3 at 40x40: 271 -> 200 us = 604 TF; 256->512 3x3 s2: 567 -> 386 us), otherwise 128 x 128 on 4 waves with two
  // co-resident blocks (64 KiB each), which beats 256 x 128 x 3 stages.  DY_WG2_EXP = 1 / 2 / 3 forces 256x128x3 / 256x256 / 128x128.
  static const int exp_env = dy_env("DY_WG2_EXP") ? atoi(dy_env("DY_WG2_EXP")) : 0;
  const int exp_mode = exp_env == 3 ? 0 : (exp_env > 0 ? exp_env : (Cout_pad >= 256 ? 2 : 0));
  P p;
  p.x = (const char*)x; p.x_ld = x_ld; p.N = N; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin_pad;
  p.dz = (const char*)dz; p.dz_ld = dz_ld; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout_pad;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil; p.part = scratch;
  p.M = (long)N * Ho * Wo;
  p.Ktot = KH * KW * Cin_pad;
  p.pointwise = (KH == 1 && KW == 1 && stride == 1 && pad == 0) ? 1 : 0;
  const int bp = (exp_mode == 1 || exp_mode == 2 || exp_mode == 4) ? 256 : 128;
  const int BQ = exp_mode == 2 ? 256 : 128;
  const int nstage = exp_mode == 1 ? 3 : 2;      // exp_mode 4: 256 x 128 x 2 stages (96 KiB)
  const int shmem = nstage * (bp / 128 + BQ / 128) * IMG;
  const bool f16 = dtype == DY_F16;
#define WG2_FN(...) (f16 ? reinterpret_cast<const void*>(&wgrad_kernel<__VA_ARGS__, f16_t>) : reinterpret_cast<const void*>(&wgrad_kernel<__VA_ARGS__, bf16_t>))
  const void* fn = exp_mode == 1 ? WG2_FN(256, 128, 3) : exp_mode == 2 ? WG2_FN(256, 256, 2) : exp_mode == 4 ? WG2_FN(256, 128, 2) : WG2_FN(128, 128, 2);
#undef WG2_FN
  static int configured = 0;          // bit mask of the variants whose LDS limit has been raised
  const int cfg_bit = 1 << (exp_mode + (f16 ? 8 : 0));
  if (!(configured & cfg_bit)) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, shmem);
    if (e != hipSuccess) {
      dy_set_error("wgrad_v2: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 3;
    }
    configured |= cfg_bit;
  }
  const int tiles_p = dy_cdiv(p.Ktot, bp);
  p.tiles_q = dy_cdiv(Cout_pad, BQ);
  const int tiles = tiles_p * p.tiles_q;
  // about one wave of blocks over the chip (two 128x128 blocks per CU), at least 8 steps per block, and the slabs must fit the
  // scratch buffer.  1,024 blocks were no faster on the YOLOv8-n shapes (sum of 19 layers 913 us vs 908 us) and write + re-read
  // twice the partial tiles (64 KiB per block).
  static const long env_target = dy_env("DY_WG2_TARGET") ? atol(dy_env("DY_WG2_TARGET")) : 0;
  const long target = env_target > 0 ? env_target : 512;
  long splits = (target + tiles - 1) / tiles;
  const long max_splits = (p.M + 8L * BKP - 1) / (8L * BKP);
  if (splits > max_splits) splits = max_splits;
  const long fit = scratch_elems / ((long)tiles * bp * BQ);
  DY_CHECK(fit >= 1, "dy_conv2d_wgrad: scratch too small (%ld floats, need %ld)", scratch_elems, (long)tiles * bp * BQ);
  if (splits > fit) splits = fit;
  if (splits < 1) splits = 1;
  if (splits > 65535) splits = 65535;
  long chunk = (p.M + splits - 1) / splits;
  chunk = (chunk + BKP - 1) / BKP * BKP;
  splits = (p.M + chunk - 1) / chunk;
  p.chunk = chunk;
  hipStream_t st = (hipStream_t)stream;
  dy_note_kernel(exp_mode == 2 ? "wg2::wgrad_kernel<256, 256, 2>+reduce_kernel" : (exp_mode == 0 ? "wg2::wgrad_kernel<128, 128, 2>+reduce_kernel" : "wg2::wgrad_kernel<256, 128>+reduce_kernel"));
#define WG2_GO(NT_, ...)                                                                              \
  do {                                                                                                 \
    if (f16) wgrad_kernel<__VA_ARGS__, f16_t><<<dim3(tiles, (unsigned)splits), NT_, shmem, st>>>(p);   \
    else wgrad_kernel<__VA_ARGS__, bf16_t><<<dim3(tiles, (unsigned)splits), NT_, shmem, st>>>(p);      \
  } while (0)
  if (exp_mode == 1) WG2_GO(512, 256, 128, 3);
  else if (exp_mode == 2) WG2_GO(512, 256, 256, 2);
  else if (exp_mode == 4) WG2_GO(512, 256, 128, 2);
  else WG2_GO(256, 128, 128, 2);
#undef WG2_GO
  DY_LAUNCH_CHECK();
  reduce_kernel<<<dim3(dy_cdiv(Cout, 32), p.Ktot), 256, 0, st>>>(scratch, (int)splits, tiles, p.tiles_q, bp, BQ, Cout, Cin, Cin_pad,
                                                                            KH, KW, p.Ktot, g_oihw);
  DY_LAUNCH_CHECK();
  return 0;
}
