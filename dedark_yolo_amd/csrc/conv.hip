// Implicit-GEMM convolution on MFMA for gfx950 (CDNA4): forward, data-gradient, weight-gradient.
//
// Replaces nn.Conv2d as used by Conv (reference ultralytics/nn/modules/conv.py:38-55), add_conv (block.py:24-45),
// ConvBlock (common.py:9-23), Detect heads (head.py:40-46), RFBblock (block.py:703-734), extractor Linear (common.py:65-66).
//
// GEMM view (forward):  D[m][n] = sum_k A[m][k] * Wp[n][k]
//   m = (image, ho, wo) output pixel, n = output channel, k = (kh, kw, ci) with ci fastest (NHWC gather, 16-byte vectors).
// Tile: BM x BN x (128 bytes of K) per step, 256 threads = 4 waves (WM x WN), each wave TM x TN MFMA 32x32 tiles.
//   bf16: v_mfma_f32_32x32x16_bf16  (K step 64 elements);  f32: v_mfma_f32_32x32x2_f32 (exact f32, K step 32 elements).
// LDS: A tile [BM][144 B], B tile [BN][144 B]  (128 B of K + 16 B pad: conflict-free ds_read_b128 / ds_write_b128).
// Pipeline: global -> registers prefetch of step s+1 overlaps the MFMAs of step s (register staging; single LDS buffer).
// Block ids are remapped so that each XCD (private L2) owns a contiguous range of output tiles (3x3 halo rows and the
// weight panel stay L2-resident).
#include <stdlib.h>
#include "dy_common.h"
#include "conv_epilogue.h"
#include "../../include/dedark_yolo.h"

// pipelined wide-channel bf16 kernel (conv_v2.hip)
bool dy_conv_v2_eligible(const dy_conv_desc* d);
int dy_conv_v2_launch(const dy_conv_desc* d, int mode, void* stream);
int dy_conv_v2_launch_classes(const dy_conv_desc* classes, int ncls, void* stream);
// band kernel for 3x3 / stride-1 bf16 convs (conv_v3.hip)
bool dy_conv_v5_eligible(const dy_conv_desc* d, int mode);
int dy_conv_v5_launch(const dy_conv_desc* d, int mode, void* stream);
bool dy_conv_v4_eligible(const dy_conv_desc* d, int mode);
int dy_conv_v4_launch(const dy_conv_desc* d, int mode, void* stream);
bool dy_conv_v5_classes_eligible(const dy_conv_desc* classes, int ncls);
int dy_conv_v5_launch_classes(const dy_conv_desc* classes, int ncls, void* stream);
bool dy_conv_v4_classes_eligible(const dy_conv_desc* classes, int ncls);
int dy_conv_v4_launch_classes(const dy_conv_desc* classes, int ncls, void* stream);
bool dy_conv_v3_eligible(const dy_conv_desc* d);
int dy_conv_v3_launch(const dy_conv_desc* d, int mode, void* stream);
// pipelined bf16 weight gradient (wgrad_v2.hip)
bool dy_wgrad_v2_eligible(int dtype, int Cin_pad, int Cout_pad, int KH, int KW, long M, long x_ld, long dz_ld);
int dy_wgrad_v2_launch(const void* x, long x_ld, int N, int Hi, int Wi, int Cin_pad, const void* dz, long dz_ld, int Ho, int Wo,
                       int Cout_pad, int KH, int KW, int stride, int pad, int dil, int Cout, int Cin, float* scratch,
                       long scratch_elems, float* g_oihw, int dtype, void* stream);
// whole-input windows = fully connected layers (dense.hip)
bool dy_dense_fwd_eligible(const dy_conv_desc* d);
int dy_dense_fwd_launch(const dy_conv_desc* d, void* stream);
bool dy_dense_dgrad_eligible(const dy_conv_desc* d);
int dy_dense_dgrad_launch(const dy_conv_desc* d, void* stream);
bool dy_dense_wgrad_eligible(int Hi, int Wi, int Ho, int Wo, int KH, int KW, int pad, int dil);
int dy_dense_wgrad_launch(const void* x, long x_ld, int N, int Hi, int Wi, int Cin_pad, const void* dz, long dz_ld, int Cout, int Cin,
                          float* g_oihw, int dtype, void* stream);
// band weight gradient for the 64-channel 3x3 layers (wgrad_v3.hip)
bool dy_wgrad_v4_eligible(int dtype, int Cin_pad, int Cout_pad, int KH, int KW, long M, int N, int Hi, int Wi, int Ho, int Wo, long x_ld,
                          long dz_ld, long scratch_elems);
int dy_wgrad_v4_launch(const void* x, long x_ld, int N, int Hi, int Wi, int Cin_pad, const void* dz, long dz_ld, int Ho, int Wo,
                       int Cout_pad, int KH, int KW, int stride, int pad, int dil, int Cout, int Cin, float* scratch,
                       long scratch_elems, float* g_oihw, int dtype, void* stream);
bool dy_wgrad_v3_eligible(int dtype, int Cin_pad, int Cout_pad, int KH, int KW, int stride, int pad, int dil, int N, int Hi, int Wi,
                          long x_ld, long dz_ld, long scratch_elems);
int dy_wgrad_v3_launch(const void* x, long x_ld, int N, int Hi, int Wi, int Cin_pad, const void* dz, long dz_ld, int Cout_pad, int Cout,
                       int Cin, float* scratch, long scratch_elems, float* g_oihw, int dtype, void* stream);
// direct stem kernels (conv_small.hip)
bool dy_conv_px_eligible(const dy_conv_desc* d);
bool dy_conv_px_has_shape(const dy_conv_desc* d);
int dy_conv_px_launch(const dy_conv_desc* d, void* stream);
bool dy_conv_stem_fwd_eligible(const dy_conv_desc* d);
int dy_conv_stem_fwd_launch(const dy_conv_desc* d, void* stream);
bool dy_conv_small_dgrad_eligible(const dy_conv_desc* d);
int dy_conv_small_dgrad_launch(const dy_conv_desc* d, void* stream);

namespace {

constexpr int ROWB = 144;       // LDS bytes per tile row
constexpr int NTHREADS = 256;

struct ConvP {
  const char* src;
  long src_ld;
  int N, Hs, Ws, Cs;
  const char* w;
  char* dst;
  long dst_ld;
  int Hd, Wd, Cd;
  int KH, KW, stride, pad, dil;
  const float* scale;
  const float* shift;
  int act;
  double* stats;
  int accumulate;
  long M;     // N*Hd*Wd
  int Ktot;   // KH*KW*Cs
  int tiles_n;
  int nblk;
  int ablate;   // DY_ABLATE env (diagnostics only): 1 skip global loads, 2 skip MFMA, 4 skip stores, 8 skip LDS restage
  long dst_row, dst_img;          // destination row / image strides in elements (dst_row == 0: dense, offset = m * dst_ld)
  int kh0, khs, kw0, kws, KWf;    // window tap (th, tw) -> weight tap (kh0 + khs*th, kw0 + kws*tw) of a KWf-wide pack
  long w_row;                     // elements per output-channel row of the weight pack (KHf*KWf*Cs)
  int ncls;                       // > 1: parity classes of a stride-2 data gradient in one launch
  DyParityCls cls[4];
};

template <typename PP>
__device__ inline long dst_offset(const PP& p, long m) {
  if (p.dst_row == 0) return m * p.dst_ld;
  const long HWd = (long)p.Hd * p.Wd;
  const long img = m / HWd;
  const int rem = (int)(m - img * HWd);
  const int oh = rem / p.Wd, ow = rem - oh * p.Wd;
  return img * p.dst_img + (long)oh * p.dst_row + (long)ow * p.dst_ld;
}

__device__ inline int xcd_remap(int bid, int nblk) {
  // bijective: blocks b, b+8, ... share an XCD; give each XCD a contiguous chunk of tile ids
  int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

// ---- MFMA over one staged K-step. LDS rows hold 128 bytes of K. ------------------------------------------------
template <typename T, int TM, int TN>
__device__ inline void mma_step(const char* As, const char* Bs, int a_row0, int b_row0, int lane, f32x16 (&acc)[TM][TN]) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    u32x4 af[TM], bfr[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
      af[i] = *reinterpret_cast<const u32x4*>(As + (a_row0 + i * 32 + r) * ROWB + kk * 32 + h * 16);
#pragma unroll
    for (int j = 0; j < TN; ++j)
      bfr[j] = *reinterpret_cast<const u32x4*>(Bs + (b_row0 + j * 32 + r) * ROWB + kk * 32 + h * 16);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (sizeof(T) == 2) {
          acc[i][j] = mfma_32x32x16<T>(af[i], bfr[j], acc[i][j]);
        } else {
          // lane half h holds k = kk*8 + 4h + q; A and B use the same k permutation, so the sum is exact
          const f32x4 fa = __builtin_bit_cast(f32x4, af[i]);
          const f32x4 fb = __builtin_bit_cast(f32x4, bfr[j]);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], fb[0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1], fb[1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2], fb[2], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[3], fb[3], acc[i][j], 0, 0, 0);
        }
      }
  }
}

// ---- forward / dgrad kernel --------------------------------------------------------------------------------------
// MODE 0: src coord = o*stride - pad + k*dil            (forward gather)
// MODE 1: t = o + pad - k*dil ; valid iff t % stride == 0 ; src coord = t / stride   (data gradient gather)
// The 128x32 tile (narrow n-scale layers) is latency-bound: six co-resident blocks per CU (78-80 VGPRs, 23 KB LDS each) instead
// of four cut the narrow forward / dgrad layers by 0.19 ms per C2 step.
template <typename T, int BM, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__(NTHREADS, (BN <= 32 ? 6 : 1)) void conv_igemm_kernel(const ConvP pk) {
  ConvP p = pk;
  constexpr int VE = DT<T>::VE;
  constexpr int BK = 8 * VE;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int AR = BM / 32, BR = BN / 32;   // vectors per thread for A / B
  static_assert(WM * WN == 4, "4 waves");
  __shared__ __attribute__((aligned(16))) char smem[(BM + BN) * ROWB];
  char* As = smem;
  char* Bs = smem + BM * ROWB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int bid = xcd_remap(blockIdx.x, p.nblk);
  if (pk.ncls > 1) {                       // several problems in one launch: this block's class replaces the launch-wide geometry
    int c = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
      if (i < pk.ncls && bid >= pk.cls[i].blk0) c = i;
    p.dst = pk.cls[c].dst; p.M = pk.cls[c].M; p.Hd = pk.cls[c].Hd; p.Wd = pk.cls[c].Wd; p.KH = pk.cls[c].KH; p.KW = pk.cls[c].KW;
    p.pad = pk.cls[c].pad; p.kh0 = pk.cls[c].kh0; p.kw0 = pk.cls[c].kw0; p.Ktot = pk.cls[c].Ktot;
    bid -= pk.cls[c].blk0;
  }
  const int tile_m = bid / p.tiles_n, tile_n = bid % p.tiles_n;
  const long m0 = (long)tile_m * BM;
  const int n0 = tile_n * BN;

  const int kv = tid & 7, r0 = tid >> 3;
  // per-row source bases
  const char* a_base[AR];
  int a_h[AR], a_w[AR];
  bool a_ok[AR];
  const DyTileWalk walk(m0, p.Hd, p.Wd);
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int r = r0 + 32 * i;
    a_ok[i] = m0 + r < p.M;
    int img, oh, ow;
    walk.at(r, img, oh, ow);
    a_base[i] = p.src + (long)img * p.Hs * p.Ws * p.src_ld * (long)sizeof(T);
    if (MODE == 0) {
      a_h[i] = oh * p.stride - p.pad;
      a_w[i] = ow * p.stride - p.pad;
    } else {
      a_h[i] = oh + p.pad;
      a_w[i] = ow + p.pad;
    }
  }
  // k state of this thread's vector column: k = k0 + kv*VE -> (kh, kw, ci)
  int kcur = kv * VE;
  int ci, kh, kw;
  {
    int tap = kcur / p.Cs;
    ci = kcur - tap * p.Cs;
    kh = tap / p.KW;
    kw = tap - kh * p.KW;
  }
  const char* b_ptr[BR];
  bool b_ok[BR];
#pragma unroll
  for (int j = 0; j < BR; ++j) {
    int n = n0 + r0 + 32 * j;
    b_ok[j] = n < p.Cd;
    b_ptr[j] = p.w + (long)(b_ok[j] ? n : 0) * p.w_row * (long)sizeof(T);
  }

  u32x4 ra[AR], rb[BR];
  auto load_step = [&]() {
    const bool kvalid = kcur < p.Ktot;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      bool ok = kvalid && a_ok[i];
      int sh, sw;
      if (MODE == 0) {
        sh = a_h[i] + kh * p.dil;
        sw = a_w[i] + kw * p.dil;
      } else {
        int th = a_h[i] - kh * p.dil, tw = a_w[i] - kw * p.dil;
        ok = ok && th >= 0 && tw >= 0;
        if (p.stride == 1) {
          sh = th;
          sw = tw;
        } else {
          sh = th / p.stride;
          sw = tw / p.stride;
          ok = ok && (sh * p.stride == th) && (sw * p.stride == tw);
        }
      }
      ok = ok && sh >= 0 && sh < p.Hs && sw >= 0 && sw < p.Ws;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ok) v = *reinterpret_cast<const u32x4*>(a_base[i] + (((long)sh * p.Ws + sw) * p.src_ld + ci) * (long)sizeof(T));
      ra[i] = v;
    }
    const long wk =((long)((p.kh0 + p.khs * kh) * p.KWf + p.kw0 + p.kws * kw) * p.Cs + ci) * (long)sizeof(T);
#pragma unroll
    for (int j = 0; j < BR; ++j) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (kvalid && b_ok[j]) v = *reinterpret_cast<const u32x4*>(b_ptr[j] + wk);
      rb[j] = v;
    }
  };
  auto advance = [&]() {
    kcur += BK;
    ci += BK;
    while (ci >= p.Cs) {
      ci -= p.Cs;
      if (++kw == p.KW) { kw = 0; ++kh; }
    }
  };
  auto store_step = [&]() {
#pragma unroll
    for (int i = 0; i < AR; ++i) *reinterpret_cast<u32x4*>(As + (r0 + 32 * i) * ROWB + kv * 16) = ra[i];
#pragma unroll
    for (int j = 0; j < BR; ++j) *reinterpret_cast<u32x4*>(Bs + (r0 + 32 * j) * ROWB + kv * 16) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nsteps = (p.Ktot + BK - 1) / BK;
  load_step();
  store_step();
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const bool more = s + 1 < nsteps;
    if (more && !(DY_ABLATE_OF(p) & 1)) {
      advance();
      load_step();
    }
    if (!(DY_ABLATE_OF(p) & 2)) mma_step<T, TM, TN>(As, Bs, wm * (BM / WM), wn * (BN / WN), lane, acc);
    if (more && (DY_ABLATE_OF(p) & 8)) continue;
    __syncthreads();
    if (more) {
      store_step();
      __syncthreads();
    }
  }

  // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int cl = lane & 31, hh = lane >> 5;
  float csum[TN], csq[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) { csum[j] = 0.f; csq[j] = 0.f; }
  T* dst = reinterpret_cast<T*>(p.dst);
  if constexpr (sizeof(T) == 2) {
    // bf16: transposed LDS image + ds_read_b64_tr_b16 -> 16-byte stores (conv_epilogue.h); the per-lane 2-byte stores of the
    // scalar path below wrote 64-byte row fragments and made the narrow n-scale layers store-bound
    static_assert(dy_epi::image_bytes<BM, BN>() <= (BM + BN) * ROWB, "epilogue image must fit the staging buffers");
    if (!(DY_ABLATE_OF(p) & 4))
      dy_epi::store_tile<BM, BN, WM, WN, TM, TN>(smem, acc, wm, wn, lane, wave, m0, n0, p.M, p.Cd, p.scale, p.shift, p.act,
                                                 p.accumulate, dst,
                                                 [&](long m) { return dst_offset(p, m); }, csum, csq);
  } else {
    int nn[TN];
    bool nok[TN];
    float sc[TN], sf[TN];
  #pragma unroll
    for (int j = 0; j < TN; ++j) {
      nn[j] = n0 + wn * (BN / WN) + j * 32 + cl;
      nok[j] = nn[j] < p.Cd;
      sc[j] = (nok[j] && p.scale) ? p.scale[nn[j]] : 1.f;
      sf[j] = (nok[j] && p.shift) ? p.shift[nn[j]] : 0.f;
    }
  #pragma unroll
    for (int i = 0; i < TM; ++i) {
  #pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (m >= p.M) continue;
        T* orow = dst + dst_offset(p, m);          // one pixel decode per row (strided destinations of the parity-split dgrad)
  #pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (nok[j]) {
            float a = acc[i][j][r];
            csum[j] += a;
            csq[j] += a * a;
            float v = dy_act(p.act, a * sc[j] + sf[j]);
            T* o = orow + nn[j];
            if (p.accumulate) v += DT<T>::ld(o);
            if (!(DY_ABLATE_OF(p) & 4)) DT<T>::st(o, v);
          }
        }
      }
    }
  }
  if (p.stats) {
    __syncthreads();                         // smem reuse
    float* red = reinterpret_cast<float*>(smem);   // [WM][BN][2]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s1 = csum[j] + __shfl_xor(csum[j], 32, 64);
      float s2 = csq[j] + __shfl_xor(csq[j], 32, 64);
      if (hh == 0) {
        int c = wn * (BN / WN) + j * 32 + cl;
        red[(wm * BN + c) * 2] = s1;
        red[(wm * BN + c) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    if (tid < BN) {
      int n = n0 + tid;
      if (n < p.Cd) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
          s1 += red[(w * BN + tid) * 2];
          s2 += red[(w * BN + tid) * 2 + 1];
        }
        // DY_STATS_REPLICAS copies of the accumulators, chosen by tile row: thousands of blocks adding into the same two
        // cache lines serialise at the memory-side atomic unit (measured: +100 us per 160x160 layer)
        double* st = p.stats + (long)(tile_m % DY_STATS_REPLICAS) * 2 * p.Cd;
        atomic_add_f64(st + n, (double)s1);
        atomic_add_f64(st + p.Cd + n, (double)s2);
      }
    }
  }
}

// ---- thin layers: Cs in {16, 32, 48}, Cd <= 32 (bf16) ----------------------------------------------------------------------
// With so few channels a K-step of the tiled kernel above is one or two taps and the block spends its life in barriers and load
// latency (16->16 3x3 at 160x160, B = 32: 45 us for 52 MB of operands = 1.2 TB/s).  Here NOTHING of the activation goes through
// LDS: the A fragment of a 32x32x16 MFMA is, per lane, 8 consecutive channels of one pixel -- exactly one 16-byte load from the
// NHWC tensor (lane&31 = pixel of the tile, lane>>5 = channel half of the 16-channel slab), shifted per tap; the 9 taps re-read
// the same lines from L1 / L2.  The weights (<= 27 KB) are staged once per block in LDS in fragment order.  No barriers in the
// main loop; a wave keeps 2 pixel tiles x 3 taps of loads in flight.  MODE as above (MODE 1 only with stride 1).
constexpr int THIN_BM = 256;                 // 4 waves x 2 tiles x 32 pixels
constexpr int THIN_MAX_SLABS = 27;           // 9 taps x 48 channels

template <int CS16, int MODE>
__global__ __launch_bounds__(NTHREADS, (CS16 == 3 ? 3 : 4)) void conv_thin_kernel(const ConvP pk) {
  ConvP p = pk;
  __shared__ __attribute__((aligned(16))) char smem[THIN_MAX_SLABS * 1024];
  static_assert(dy_epi::image_bytes<THIN_BM, 32>() <= THIN_MAX_SLABS * 1024, "epilogue image reuses the weight buffer");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int bid = xcd_remap(blockIdx.x, p.nblk);
  if (pk.ncls > 1) {
    int c = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
      if (i < pk.ncls && bid >= pk.cls[i].blk0) c = i;
    p.dst = pk.cls[c].dst; p.M = pk.cls[c].M; p.Hd = pk.cls[c].Hd; p.Wd = pk.cls[c].Wd; p.KH = pk.cls[c].KH; p.KW = pk.cls[c].KW;
    p.pad = pk.cls[c].pad; p.kh0 = pk.cls[c].kh0; p.kw0 = pk.cls[c].kw0; p.Ktot = pk.cls[c].Ktot;
    bid -= pk.cls[c].blk0;
  }
  const long m0 = (long)bid * THIN_BM;
  const int ntaps = p.KH * p.KW;

  // ---- weights -> LDS in fragment order: slab s = tap * CS16 + c16, lane slot l = (n = l & 31, channel half = l >> 5)
  for (int idx = tid; idx < ntaps * CS16 * 64; idx += NTHREADS) {
    const int sl = idx >> 6, l = idx & 63;
    const int tap = sl / CS16, c16 = sl - tap * CS16;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int n = l & 31, half = l >> 5;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (n < p.Cd) {
      const long wk = (long)((p.kh0 + p.khs * kh) * p.KWf + p.kw0 + p.kws * kw) * p.Cs + c16 * 16 + half * 8;
      v = *reinterpret_cast<const u32x4*>(p.w + ((long)n * p.w_row + wk) * 2);
    }
    *reinterpret_cast<u32x4*>(smem + (long)idx * 16) = v;
  }
  __syncthreads();

  // ---- this lane's pixel of each of the wave's two tiles
  const int row = lane & 31, half = lane >> 5;
  const DyTileWalk walk(m0, p.Hd, p.Wd);
  const char* a_base[2];
  int a_h[2], a_w[2];
  bool a_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wave * 64 + i * 32 + row;
    a_ok[i] = m0 + r < p.M;
    int img, oh, ow;
    walk.at(r, img, oh, ow);
    a_base[i] = p.src + ((long)img * p.Hs * p.Ws * p.src_ld + half * 8) * 2;
    if (MODE == 0) {
      a_h[i] = oh * p.stride - p.pad;
      a_w[i] = ow * p.stride - p.pad;
    } else {
      a_h[i] = oh + p.pad;
      a_w[i] = ow + p.pad;
    }
  }
  f32x16 acc[2][1];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

  typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
  for (int kh = 0; kh < p.KH; ++kh) {
    u32x4 af[3][2][CS16];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int sh = MODE == 0 ? a_h[i] + kh : a_h[i] - kh;
        const int sw = MODE == 0 ? a_w[i] + kw : a_w[i] - kw;
        const bool ok = kw < p.KW && a_ok[i] && sh >= 0 && sh < p.Hs && sw >= 0 && sw < p.Ws;
        const char* src = a_base[i] + ((long)sh * p.Ws + sw) * p.src_ld * 2;
#pragma unroll
        for (int c = 0; c < CS16; ++c) {
          u32x4 v = {0u, 0u, 0u, 0u};
          if (ok) v = *reinterpret_cast<const u32x4*>(src + c * 32);
          af[kw][i][c] = v;
        }
      }
    }
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      if (kw < p.KW) {
        const char* bsl = smem + ((long)((kh * p.KW + kw) * CS16) * 64 + lane) * 16;
#pragma unroll
        for (int c = 0; c < CS16; ++c) {
          const u32x4 bf = *reinterpret_cast<const u32x4*>(bsl + c * 1024);
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[kw][i][c]), __builtin_bit_cast(bf16x8, bf),
                                                               acc[i][0], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue (conv_epilogue.h; its leading barrier also ends the reads of the weight image)
  float csum[1], csq[1];
  dy_epi::store_tile<THIN_BM, 32, 4, 1, 2, 1>(smem, acc, wave, 0, lane, wave, m0, 0, p.M, p.Cd, p.scale, p.shift, p.act, p.accumulate,
                                             reinterpret_cast<bf16_t*>(p.dst), [&](long m) { return dst_offset(p, m); }, csum, csq);
  if (p.stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);   // [4 waves][32][2]
    const float s1 = csum[0] + __shfl_xor(csum[0], 32, 64);
    const float s2 = csq[0] + __shfl_xor(csq[0], 32, 64);
    if (half == 0) {
      red[(wave * 32 + row) * 2] = s1;
      red[(wave * 32 + row) * 2 + 1] = s2;
    }
    __syncthreads();
    if (tid < 32 && tid < p.Cd) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        t1 += red[(w * 32 + tid) * 2];
        t2 += red[(w * 32 + tid) * 2 + 1];
      }
      double* st = p.stats + (long)(bid % DY_STATS_REPLICAS) * 2 * p.Cd;
      atomic_add_f64(st + tid, (double)t1);
      atomic_add_f64(st + p.Cd + tid, (double)t2);
    }
  }
}

// ---- weight gradient ------------------------------------------------------------------------------------------------
// dW[co][k] += sum_m dz[m][co] * X[m][k],  k = (kh,kw,ci).  GEMM rows = co, cols = k, reduction over pixels m.
// Both operands are pixel-major in HBM, MFMA wants the reduction index contiguous per lane: each thread loads a
// VE x VE (pixels x channels) block with 16-byte loads, transposes it in registers and writes pixel-contiguous
// 16-byte rows to LDS.  Grid = (co tiles * k tiles, pixel splits); partial tiles are added with f32 atomics
// (32 consecutive floats per half-wave = the full-rate atomic shape).
struct WgP {
  const char* x;
  long x_ld;
  int N, Hi, Wi, Cin;
  const char* dz;
  long dz_ld;
  int Ho, Wo, Cout;
  int KH, KW, stride, pad, dil;
  float* part;   // [splits][tiles][BM][BN] partial tiles (plain stores; summed by wgrad_reduce_kernel)
  long M;
  int Ktot;
  int tiles_k;
  int tiles;
  long chunk;    // pixels per split (multiple of MK)
  int pointwise;
};

template <typename T, int VE> struct Transposer;
template <> struct Transposer<float, 4> {
  // in[e] = 4 channels of pixel e ; out[c] = 4 pixels of channel c
  __device__ static inline void run(const u32x4 (&in)[4], u32x4 (&out)[4]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      u32x4 o = {in[0][c], in[1][c], in[2][c], in[3][c]};
      out[c] = o;
    }
  }
};
template <> struct Transposer<bf16_t, 8> {
  // in[e] = 8 bf16 channels (4 dwords) of pixel e ; out[c] = 8 pixels of channel c
  __device__ static inline void run(const u32x4 (&in)[8], u32x4 (&out)[8]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      u32x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t lo = in[2 * q][c >> 1], hi = in[2 * q + 1][c >> 1];
        // select half (c&1) of lo into bits 0-15 and of hi into bits 16-31
        o[q] = (c & 1) ? __builtin_amdgcn_perm(hi, lo, 0x07060302u) : __builtin_amdgcn_perm(hi, lo, 0x05040100u);
      }
      out[c] = o;
    }
  }
};

template <> struct Transposer<f16_t, 8> : Transposer<bf16_t, 8> {};      // 16-bit payloads: the same shuffle

// The small tiles are latency-bound (one global-load round trip per 64-pixel step): cap them at 128 VGPRs so that four blocks
// share a CU instead of three (they compile to 132 without the bound).
template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(NTHREADS, (BM * BN <= 4096 ? 4 : 1)) void conv_wgrad_kernel(WgP p) {
  constexpr int VE = DT<T>::VE;
  constexpr int MK = 8 * VE;                 // pixels per step
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int NA = 8 * (BM / VE), NB = 8 * (BN / VE);    // VE x VE units in the A (dz) / B (x) tiles
  constexpr int UPT = (NA + NB + NTHREADS - 1) / NTHREADS;
  __shared__ __attribute__((aligned(16))) char smem[(BM + BN) * ROWB];
  char* As = smem;
  char* Bs = smem + BM * ROWB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tile_c = blockIdx.x / p.tiles_k, tile_k = blockIdx.x % p.tiles_k;
  const int c0 = tile_c * BM, k0 = tile_k * BN;
  const long m_begin = (long)blockIdx.y * p.chunk;
  const long m_end = (m_begin + p.chunk < p.M) ? m_begin + p.chunk : p.M;
  if (m_begin >= m_end) return;

  // static unit assignment
  bool u_isA[UPT], u_on[UPT], u_chan_ok[UPT];
  int u_pg[UPT], u_cg[UPT];       // pixel group (0..7), channel group
  int u_kh[UPT], u_kw[UPT], u_ci[UPT];
#pragma unroll
  for (int q = 0; q < UPT; ++q) {
    int u = tid + q * NTHREADS;
    u_on[q] = u < NA + NB;
    u_isA[q] = u < NA;
    u_kh[q] = u_kw[q] = u_ci[q] = 0;
    if (u_isA[q]) {
      u_cg[q] = u % (BM / VE);
      u_pg[q] = u / (BM / VE);
      u_chan_ok[q] = (c0 + u_cg[q] * VE) < p.Cout;
    } else {
      int v = u - NA;
      u_cg[q] = v % (BN / VE);
      u_pg[q] = (v / (BN / VE)) & 7;
      int k = k0 + u_cg[q] * VE;
      u_chan_ok[q] = k < p.Ktot;
      int kk = u_chan_ok[q] ? k : 0;
      int tap = kk / p.Cin;
      u_ci[q] = kk - tap * p.Cin;
      u_kh[q] = tap / p.KW;
      u_kw[q] = tap - u_kh[q] * p.KW;
    }
  }
  const long HWo = (long)p.Ho * p.Wo;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  u32x4 reg[UPT][VE];
  auto load_step = [&](long mbase) {
#pragma unroll
    for (int q = 0; q < UPT; ++q) {
      const long mfirst = mbase + u_pg[q] * VE;
      if (u_isA[q]) {
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          long m = mfirst + e;
          u32x4 v = {0u, 0u, 0u, 0u};
          if (u_on[q] && u_chan_ok[q] && m < m_end)
            v = *reinterpret_cast<const u32x4*>(p.dz + (m * p.dz_ld + c0 + u_cg[q] * VE) * (long)sizeof(T));
          reg[q][e] = v;
        }
      } else if (p.pointwise) {
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          long m = mfirst + e;
          u32x4 v = {0u, 0u, 0u, 0u};
          if (u_on[q] && u_chan_ok[q] && m < m_end)
            v = *reinterpret_cast<const u32x4*>(p.x + (m * p.x_ld + u_ci[q]) * (long)sizeof(T));
          reg[q][e] = v;
        }
      } else {
        long mm = mfirst < p.M ? mfirst : 0;
        int img = (int)(mm / HWo);
        int rem = (int)(mm - (long)img * HWo);
        int oh = rem / p.Wo, ow = rem - oh * p.Wo;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          long m = mfirst + e;
          int ih = oh * p.stride - p.pad + u_kh[q] * p.dil;
          int iw = ow * p.stride - p.pad + u_kw[q] * p.dil;
          bool ok = u_on[q] && u_chan_ok[q] && m < m_end && ih >= 0 && ih < p.Hi && iw >= 0 && iw < p.Wi;
          u32x4 v = {0u, 0u, 0u, 0u};
          if (ok)
            v = *reinterpret_cast<const u32x4*>(p.x + ((((long)img * p.Hi + ih) * p.Wi + iw) * p.x_ld + u_ci[q]) * (long)sizeof(T));
          reg[q][e] = v;
          if (++ow == p.Wo) {
            ow = 0;
            if (++oh == p.Ho) { oh = 0; ++img; }
          }
        }
      }
    }
  };
  auto store_step = [&]() {
#pragma unroll
    for (int q = 0; q < UPT; ++q) {
      if (!u_on[q]) continue;
      u32x4 t[VE];
      Transposer<T, VE>::run(reg[q], t);
      char* base = (u_isA[q] ? As : Bs) + (u_cg[q] * VE) * ROWB + u_pg[q] * 16;
#pragma unroll
      for (int c = 0; c < VE; ++c) *reinterpret_cast<u32x4*>(base + c * ROWB) = t[c];
    }
  };

  load_step(m_begin);
  store_step();
  __syncthreads();
  for (long mb = m_begin; mb < m_end; mb += MK) {
    const bool more = mb + MK < m_end;
    if (more) load_step(mb + MK);
    mma_step<T, TM, TN>(As, Bs, wm * (BM / WM), wn * (BN / WN), lane, acc);
    __syncthreads();
    if (more) {
      store_step();
      __syncthreads();
    }
  }
  // partial tile -> scratch slab of this (split, tile): 32 consecutive floats per half-wave, no atomics, deterministic
  const int cl = lane & 31, hh = lane >> 5;
  float* slab = p.part + ((long)blockIdx.y * p.tiles + blockIdx.x) * (BM * BN);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = wn * (BN / WN) + j * 32 + cl;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        slab[row * BN + col] = acc[i][j][r];
      }
  }
}

// sum the split slabs of one output channel and write its OIHW gradient row (contiguous): one block per co.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int splits, int tiles, int tiles_k,
                                                            int bm, int bn, int Cin, int Cin_pad, int KH, int KW, int Ktot,
                                                            float* __restrict__ g) {
  // block = (one output channel, 32 consecutive k); 8 split-lanes per k walk the slabs with 8 loads in flight each
  __shared__ float red[8][33];
  const int co = blockIdx.x;
  const int tile_c = co / bm, r = co - tile_c * bm;
  const int kl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int k = blockIdx.y * 32 + kl;
  float a = 0.f;
  if (k < Ktot) {
    const int tile_k = k / bn, c = k - tile_k * bn;
    const float* src = part + ((long)(tile_c * tiles_k + tile_k) * bm + r) * bn + c;
    const long sstride = (long)tiles * bm * bn;
    int s = sl;
    for (; s + 56 < splits; s += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(long)(s + 8 * u) * sstride];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; s < splits; s += 8) a += src[(long)s * sstride];
  }
  red[sl][kl] = a;
  __syncthreads();
  if (sl == 0 && k < Ktot) {
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) t += red[u][kl];
    const int tap = k / Cin_pad, ci = k - tap * Cin_pad;
    if (ci < Cin) {
      const int kh = tap / KW, kw = tap - kh * KW;
      g[(((long)co * Cin + ci) * KH + kh) * KW + kw] = t;
    }
  }
}

// ---- weight (un)packing -----------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cout_pad, int Cin,
                                   int Cin_pad, int KH, int KW, int transposed) {
  long total = (long)Cout_pad * KH * KW * Cin_pad;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int co, kh, kw, ci;
    long r = i;
    if (!transposed) {       // [Cout_pad][KH][KW][Cin_pad]
      ci = (int)(r % Cin_pad); r /= Cin_pad;
      kw = (int)(r % KW); r /= KW;
      kh = (int)(r % KH); r /= KH;
      co = (int)r;
    } else {                 // [Cin_pad][KH][KW][Cout_pad]
      co = (int)(r % Cout_pad); r /= Cout_pad;
      kw = (int)(r % KW); r /= KW;
      kh = (int)(r % KH); r /= KH;
      ci = (int)r;
    }
    float v = (ci < Cin && co < Cout) ? w[(((long)co * Cin + ci) * KH + kh) * KW + kw] : 0.f;
    DT<T>::st(out + i, v);
  }
}

// all weights of a model in one launch: block b belongs to the item with first_block <= b < next.first_block.
// A block = one 32 (output channels) x 32 (input channels) tile of one weight, all taps: the OIHW source rows of the tile are
// contiguous runs of 32 * KH * KW floats (read coalesced into LDS), the packed destination -- [Cout_pad][KH][KW][Cin_pad] or the
// transposed [Cin_pad][KH][KW][Cout_pad] -- is written in runs of 32 consecutive elements read back from LDS.  (The first version
// computed one output element per thread and gathered its source with a 36-byte stride: 2.5 GB of fetch traffic per launch by
// PMC for 0.35 GB of weights.  The launch itself still takes ~0.5 ms per C3 step with either version -- not yet understood.)
// Windows of more than PACK_TAPS taps (the extractor's 8x8 fully connected layer) go through in tap chunks.
constexpr int PACK_TILE = 32, PACK_TAPS = 9;

__device__ inline void pack_store(const dy_pack_item& it, long o, float v) {
  if (it.dtype == DY_F32) ((float*)it.packed)[o] = v;
  else if (it.dtype == DY_F16) ((f16_t*)it.packed)[o] = (f16_t)v;
  else ((bf16_t*)it.packed)[o] = f32_to_bf16(v);
}

// one chunk of TC taps [t0, t0 + TC) of the block's 32 x 32 tile.  TC is a compile-time constant for the windows that matter
// (1x1, 3x3), so that the index arithmetic has no run-time divisions.  TC == 0: run-time `tc`.
template <int TC>
__device__ inline void pack_chunk(const dy_pack_item& it, float (&tile)[PACK_TILE][PACK_TILE * PACK_TAPS + 1], int co0, int ci0, int T,
                                  int t0, int tc_rt) {
  const int tc = TC ? TC : tc_rt;
  const int tid = threadIdx.x;
  const int run = PACK_TILE * tc;
  {   // read: 8 threads per source row (row co of the tile = `run` consecutive floats when tc == T)
    const int r = tid >> 3, co = co0 + r;
    const bool row_ok = co < it.Cout;
    const float* src = it.w + ((long)co * it.Cin + ci0) * T + t0;
    for (int k = tid & 7; k < run; k += 8) {
      const int cl = k / tc, t = k - cl * tc;
      const int ci = ci0 + cl;
      float v = 0.f;
      if (row_ok && ci < it.Cin) v = src[(long)cl * T + t];
      tile[r][cl * PACK_TAPS + t] = v;
    }
  }
  __syncthreads();
  // write: 16 bytes per thread (8 consecutive elements of the destination's contiguous dimension in the 16-bit dtypes, 4 in f32; the
  // padded extents are multiples of that vector, so a group is either whole or absent).  One 2-byte store per thread made this phase
  // 36 store instructions per thread and tile.
  const int VE = it.dtype == DY_F32 ? 4 : 8, GP = PACK_TILE / VE;           // elements per store, groups per 32-wide run
  const int n_vec = PACK_TILE * GP * tc;
  for (int i = tid; i < n_vec; i += 256) {
    const int g = i % GP, q = i / GP;
    const int a = q / tc, t = q - a * tc;                  // a = tile row (co) of a plain pack, tile column (ci) of a transposed one
    float v[8];
    long o;
    bool ok;
    if (!it.transposed) {                                  // packed[co][tap][ci]
      const int co = co0 + a, ci = ci0 + g * VE;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = e < VE ? tile[a][(g * VE + e) * PACK_TAPS + t] : 0.f;
      ok = co < it.Cout_pad && ci < it.Cin_pad;
      o = ((long)co * T + t0 + t) * it.Cin_pad + ci;
    } else {                                               // packed[ci][tap][co]
      const int ci = ci0 + a, co = co0 + g * VE;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = e < VE ? tile[g * VE + e][a * PACK_TAPS + t] : 0.f;
      ok = co < it.Cout_pad && ci < it.Cin_pad;
      o = ((long)ci * T + t0 + t) * it.Cout_pad + co;
    }
    if (!ok) continue;
    if (it.dtype == DY_F32) stvec<float>((float*)it.packed + o, v);
    else if (it.dtype == DY_F16) stvec<f16_t>((f16_t*)it.packed + o, v);
    else stvec<bf16_t>((bf16_t*)it.packed + o, v);
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void pack_multi_kernel(const dy_pack_item* __restrict__ items, int n_items) {
  __shared__ float tile[PACK_TILE][PACK_TILE * PACK_TAPS + 1];          // [co][ci * taps + t]; +1: conflict-free transposed reads
  // the block's item: last one with first_block <= b; the first_block values go through LDS in one round trip
  __shared__ long fb[1024];
  const long b = blockIdx.x;
  const bool cached = n_items <= 1024;
  if (cached) {
    for (int i = threadIdx.x; i < n_items; i += 256) fb[i] = items[i].first_block;
    __syncthreads();
  }
  int lo = 0, hi = n_items - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((cached ? fb[mid] : items[mid].first_block) <= b) lo = mid; else hi = mid - 1;
  }
  const dy_pack_item it = items[lo];
  const int tiles_ci = (it.Cin_pad + PACK_TILE - 1) / PACK_TILE;
  const int tb = (int)(b - it.first_block);
  const int co0 = (tb / tiles_ci) * PACK_TILE, ci0 = (tb % tiles_ci) * PACK_TILE;
  const int T = it.KH * it.KW;
  if (T == 9) pack_chunk<9>(it, tile, co0, ci0, 9, 0, 9);
  else if (T == 1) pack_chunk<1>(it, tile, co0, ci0, 1, 0, 1);
  else
    for (int t0 = 0; t0 < T; t0 += PACK_TAPS) pack_chunk<0>(it, tile, co0, ci0, T, t0, T - t0 < PACK_TAPS ? T - t0 : PACK_TAPS);
}

__global__ void unpack_wgrad_kernel(const float* __restrict__ dwp, float* __restrict__ g, int Cout, int Cin, int Cin_pad,
                                    int KH, int KW) {
  long total = (long)Cout * Cin * KH * KW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i;
    int kw = (int)(r % KW); r /= KW;
    int kh = (int)(r % KH); r /= KH;
    int ci = (int)(r % Cin); r /= Cin;
    int co = (int)r;
    g[i] = dwp[(((long)co * KH + kh) * KW + kw) * Cin_pad + ci];
  }
}

void fill_convp(ConvP& p, const dy_conv_desc* d, const dy_conv_desc* classes, int ncls) {
  p.ncls = 0;
  if (ncls > 1) {
    p.ncls = ncls;
    for (int c = 0; c < ncls; ++c) {
      const dy_conv_desc& q = classes[c];
      DyParityCls& k = p.cls[c];
      k.dst = (char*)q.dst; k.M = (long)q.N * q.Hd * q.Wd; k.Hd = q.Hd; k.Wd = q.Wd; k.KH = q.KH; k.KW = q.KW; k.pad = q.pad;
      k.kh0 = q.kh0; k.kw0 = q.kw0; k.Ktot = q.KH * q.KW * q.Cs; k.blk0 = 0; k._r = 0;
    }
  }
  p.src = (const char*)d->src; p.src_ld = d->src_ld; p.N = d->N; p.Hs = d->Hs; p.Ws = d->Ws; p.Cs = d->Cs;
  p.w = (const char*)d->w; p.dst = (char*)d->dst; p.dst_ld = d->dst_ld; p.Hd = d->Hd; p.Wd = d->Wd; p.Cd = d->Cd;
  p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
  p.scale = d->scale; p.shift = d->shift; p.act = d->act; p.stats = d->stats; p.accumulate = d->accumulate;
  p.M = (long)d->N * d->Hd * d->Wd;
  p.Ktot = d->KH * d->KW * d->Cs;
  static const int ablate = dy_env("DY_ABLATE") ? atoi(dy_env("DY_ABLATE")) : 0;
  p.ablate = ablate;
  p.dst_row = d->dst_row_stride;
  p.dst_img = d->dst_img_stride ? d->dst_img_stride : (long)d->Hd * d->dst_row_stride;
  if (d->KHf > 0) {
    p.kh0 = d->kh0; p.khs = d->kh_step; p.kw0 = d->kw0; p.kws = d->kw_step; p.KWf = d->KWf;
    p.w_row = (long)d->KHf * d->KWf * d->Cs;
  } else {
    p.kh0 = 0; p.khs = 1; p.kw0 = 0; p.kws = 1; p.KWf = d->KW;
    p.w_row = p.Ktot;
  }
}

template <typename T, int MODE>
int launch_conv(const dy_conv_desc* d, hipStream_t st, const dy_conv_desc* classes = nullptr, int ncls = 0) {
  ConvP p;
  fill_convp(p, d, classes, ncls);
  constexpr int BM = 128;
  const int tiles_m = dy_cdiv(p.M, BM);
  auto blocks = [&](int bn) {                 // tiles_n / nblk (+ the block ranges of the classes)
    p.tiles_n = dy_cdiv(d->Cd, bn);
    p.nblk = tiles_m * p.tiles_n;
    if (p.ncls > 1) {
      long acc = 0;
      for (int c = 0; c < p.ncls; ++c) {
        p.cls[c].blk0 = (int)acc;
        acc += dy_cdiv(p.cls[c].M, BM) * p.tiles_n;
      }
      p.nblk = (int)acc;
    }
  };
  dy_note_kernel(d->Cd <= 32 ? "conv_igemm_kernel<BN=32>" : (d->Cd <= 64 ? "conv_igemm_kernel<BN=64>" : "conv_igemm_kernel<BN=128>"));
  if (d->Cd <= 32) {
    blocks(32);
    conv_igemm_kernel<T, BM, 32, 4, 1, MODE><<<p.nblk, NTHREADS, 0, st>>>(p);
  } else if (d->Cd <= 64) {
    blocks(64);
    conv_igemm_kernel<T, BM, 64, 2, 2, MODE><<<p.nblk, NTHREADS, 0, st>>>(p);
  } else {
    blocks(128);
    conv_igemm_kernel<T, BM, 128, 2, 2, MODE><<<p.nblk, NTHREADS, 0, st>>>(p);
  }
  DY_LAUNCH_CHECK();
  return 0;
}

// thin-layer kernel: bf16, 16 / 32 / 48 source channels, <= 32 destination channels, window <= 3x3, no dilation
// Measured on the C2 layers (B = 32, single stream): a win for 16 source channels with a 3x3 window (16->16 at 160x160: 57 -> 49 us)
// and for the parity classes of a stride-2 data gradient (16->32 at 320x320: 144 -> 116 us); NOT for 32 channels at stride 1
// (32->32 3x3 at 80x80: 26 -> 34 us) or 1x1 windows (48->32: 34 -> 48 us) -- one round of loads per tap row and the per-block
// weight staging / epilogue still dominate a 256-pixel block, so those stay on the tiled kernel (`all_shapes` = tests / tuning).
bool thin_eligible(const dy_conv_desc* d, int mode, bool parity_class = false) {
  static const bool off = dy_env("DY_NO_CONV_THIN") != nullptr;
  static const bool all_shapes = getenv("DY_CONV_THIN_ALL") != nullptr;
  if (off || d->dtype != DY_BF16) return false;
  if (!all_shapes && !parity_class && !(d->Cs == 16 && d->KH * d->KW > 1)) return false;
  if (!(d->Cs == 16 || d->Cs == 32 || d->Cs == 48) || d->Cd > 32 || d->Cd % 8 != 0) return false;
  if (d->KH > 3 || d->KW > 3 || d->dil != 1 || (mode == 1 && d->stride != 1)) return false;
  if ((d->src_ld * 2) % 16 != 0 || (d->dst_ld * 2) % 16 != 0 || (long)d->N * d->Hd * d->Wd < 1024) return false;
  return true;
}

int launch_thin(const dy_conv_desc* d, int mode, hipStream_t st, const dy_conv_desc* classes = nullptr, int ncls = 0) {
  ConvP p;
  fill_convp(p, d, classes, ncls);
  p.tiles_n = 1;
  p.nblk = dy_cdiv(p.M, THIN_BM);
  if (p.ncls > 1) {
    long acc = 0;
    for (int c = 0; c < p.ncls; ++c) {
      p.cls[c].blk0 = (int)acc;
      acc += dy_cdiv(p.cls[c].M, THIN_BM);
    }
    p.nblk = (int)acc;
  }
  dy_note_kernel("conv_thin_kernel");
#define THIN(C_, M_) conv_thin_kernel<C_, M_><<<p.nblk, NTHREADS, 0, st>>>(p)
  const int c16 = d->Cs / 16;
  if (mode == 0) { if (c16 == 1) THIN(1, 0); else if (c16 == 2) THIN(2, 0); else THIN(3, 0); }
  else { if (c16 == 1) THIN(1, 1); else if (c16 == 2) THIN(2, 1); else THIN(3, 1); }
#undef THIN
  DY_LAUNCH_CHECK();
  return 0;
}

int check_conv(const dy_conv_desc* d, const char* who) {
  DY_CHECK(d && d->src && d->w && (d->dst || d->dst_planar), "%s: null pointer", who);
  DY_CHECK(d->dtype == DY_F32 || d->dtype == DY_BF16 || d->dtype == DY_F16, "%s: bad dtype %d", who, d->dtype);
  const int ve = d->dtype == DY_F32 ? 4 : 8, es = d->dtype == DY_F32 ? 4 : 2;
  DY_CHECK(d->Cs > 0 && d->Cs % ve == 0, "%s: Cs=%d must be a positive multiple of %d", who, d->Cs, ve);
  DY_CHECK(d->src_ld >= d->Cs && (d->src_ld * es) % 16 == 0, "%s: src_ld=%ld not 16-byte aligned", who, (long)d->src_ld);
  DY_CHECK(((uintptr_t)d->src) % 16 == 0 && ((uintptr_t)d->w) % 16 == 0, "%s: src/w pointer not 16-byte aligned", who);
  DY_CHECK(d->dst_ld >= d->Cd && d->Cd > 0, "%s: dst_ld=%ld < Cd=%d", who, (long)d->dst_ld, d->Cd);
  DY_CHECK(d->N > 0 && d->Hs > 0 && d->Ws > 0 && d->Hd > 0 && d->Wd > 0, "%s: empty geometry", who);
  DY_CHECK(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->dil > 0 && d->pad >= 0, "%s: bad window", who);
  DY_CHECK(d->act >= 0 && d->act <= 2, "%s: bad act", who);
  return 0;
}

}  // namespace

extern "C" int dy_conv2d_fwd(const dy_conv_desc* d, void* stream) {
  if (int e = check_conv(d, "dy_conv2d_fwd")) return e;
  const int ho = (d->Hs + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
  const int wo = (d->Ws + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
  DY_CHECK(ho == d->Hd && wo == d->Wd, "dy_conv2d_fwd: dst %dx%d does not match conv output %dx%d", d->Hd, d->Wd, ho, wo);
  if (dy_dense_fwd_eligible(d)) return dy_dense_fwd_launch(d, stream);
  if (dy_conv_stem_fwd_eligible(d)) return dy_conv_stem_fwd_launch(d, stream);
  if (dy_conv_px_eligible(d) && dy_conv_px_has_shape(d) && !d->accumulate && !d->add_src) return dy_conv_px_launch(d, stream);
  if (dy_conv_v4_eligible(d, 0)) return dy_conv_v4_launch(d, 0, stream);
  if (dy_conv_v5_eligible(d, 0)) return dy_conv_v5_launch(d, 0, stream);
  if (dy_conv_v3_eligible(d)) return dy_conv_v3_launch(d, 0, stream);
  if (dy_conv_v2_eligible(d)) return dy_conv_v2_launch(d, 0, stream);
  hipStream_t st = (hipStream_t)stream;
  if (thin_eligible(d, 0)) return launch_thin(d, 0, st);
  return d->dtype == DY_F32 ? launch_conv<float, 0>(d, st) : (d->dtype == DY_F16 ? launch_conv<f16_t, 0>(d, st) : launch_conv<bf16_t, 0>(d, st));
}

static int dgrad_dispatch(const dy_conv_desc* d, void* stream, bool* added) {
  if (int e = check_conv(d, "dy_conv2d_dgrad")) return e;
  // src = dz (conv output geometry), dst = dx (conv input geometry)
  const int ho = (d->Hd + 2 * d->pad - d->dil * (d->KH - 1) - 1) / d->stride + 1;
  const int wo = (d->Wd + 2 * d->pad - d->dil * (d->KW - 1) - 1) / d->stride + 1;
  DY_CHECK(ho == d->Hs && wo == d->Ws, "dy_conv2d_dgrad: dz %dx%d does not match conv output %dx%d", d->Hs, d->Ws, ho, wo);
  DY_CHECK(d->stats == nullptr, "dy_conv2d_dgrad: stats unsupported");
  DY_CHECK(d->KHf == 0 && d->dst_row_stride == 0, "dy_conv2d_dgrad: tap subsets / strided destinations are forward-only");
  if (dy_dense_dgrad_eligible(d)) return dy_dense_dgrad_launch(d, stream);
  if (dy_conv_small_dgrad_eligible(d)) return dy_conv_small_dgrad_launch(d, stream);
  DY_CHECK(d->dst_planar == nullptr, "dy_conv2d_dgrad: dst_planar is only supported by the direct stem kernel (bf16, 3x3 s2 p1, Cd == 8)");
  static const bool no_parity = dy_env("DY_NO_PARITY_DGRAD") != nullptr;
  if (d->stride == 2 && d->dil == 1 && !no_parity) {
    // Stride-2 data gradient = 4 independent dense problems, one per parity class (ph, pw) of the output pixel: only the
    // taps kh == (ph + pad) mod 2 reach dx[2*hq + ph], and they read dz[hq + (ph + pad - kh)/2]: a stride-1 correlation over dz
    // with a 1- or 2-tap window per axis.  The masked formulation visits all KH*KW taps for every pixel (75 % zero work).
    dy_conv_desc c[4];
    int nc = 0;
    bool ok = true;
    for (int ph = 0; ph < 2 && ok; ++ph)
      for (int pw = 0; pw < 2 && ok; ++pw) {
        int geo[2][4];                       // per axis: n taps, first weight tap, pad', extent of the class grid
        for (int ax = 0; ax < 2; ++ax) {
          const int par = ax == 0 ? ph : pw, K = ax == 0 ? d->KH : d->KW, ext = ax == 0 ? d->Hd : d->Wd;
          const int first = (par + d->pad) & 1;                    // smallest valid tap
          const int n = first < K ? (K - 1 - first) / 2 + 1 : 0;
          const int last = first + 2 * (n - 1);                    // largest valid tap = smallest source offset
          const int omin = (par + d->pad - last) / 2;              // exact: same parity
          geo[ax][0] = n; geo[ax][1] = last; geo[ax][2] = -omin; geo[ax][3] = (ext - par + 1) / 2;
        }
        if (geo[0][3] <= 0 || geo[1][3] <= 0) continue;            // no pixel of this parity
        if (geo[0][0] == 0 || geo[1][0] == 0 || geo[0][2] != geo[1][2] || geo[0][2] < 0) { ok = false; break; }
        dy_conv_desc& q = c[nc++];
        q = *d;
        const long es = d->dtype == DY_F32 ? 4 : 2;
        q.dst = (char*)d->dst + ((long)ph * d->Wd + pw) * d->dst_ld * es;
        q.dst_ld = 2 * d->dst_ld;
        q.dst_row_stride = 2L * d->Wd * d->dst_ld;
        q.dst_img_stride = (long)d->Hd * d->Wd * d->dst_ld;
        q.Hd = geo[0][3]; q.Wd = geo[1][3];
        q.KHf = d->KH; q.KWf = d->KW;
        q.KH = geo[0][0]; q.KW = geo[1][0];
        q.kh0 = geo[0][1]; q.kh_step = -2; q.kw0 = geo[1][1]; q.kw_step = -2;
        q.stride = 1; q.pad = geo[0][2];
      }
    if (ok) {
      // (running the four classes side by side on auxiliary streams was measured: 128->128 3x3 s2 at 40x40 58 -> 72 us, the
      //  training step 10.4 -> 10.75 ms: the cross-stream waits cost more than the small launches gain)
      // All classes in ONE launch (blocks ordered heaviest class first): four back-to-back launches of 1-, 2-, 2- and 4-tap
      // problems each paid their own launch floor and tail.  DY_PARITY_MULTI=0 keeps the separate launches.
      static const bool multi = !(dy_env("DY_PARITY_MULTI") && atoi(dy_env("DY_PARITY_MULTI")) == 0);
      if (multi && nc > 1) {
        dy_conv_desc r[4];
        bool all_v2 = true, none_v2 = true;
        for (int i = 0; i < nc; ++i) {
          r[i] = c[nc - 1 - i];
          const bool e2 = dy_conv_v2_eligible(&r[i]);
          all_v2 = all_v2 && e2;
          none_v2 = none_v2 && !e2;
        }
        // ... where the classes are small: once a class alone fills the chip (C3: 256->512 at 80x80, B = 64, 400 tiles of
        // 256x256 per class) one launch is SLOWER than four (560 vs 449 us) -- the short-K classes' blocks then crowd out the long ones
        const long class_tiles = dy_cdiv((long)r[0].N * r[0].Hd * r[0].Wd, 256);
        // the large-tile kernels: ONE launch whose block order hands every XCD its share of every class, longest tiles first (four
        // launches each end in their own partial round of tiles: 512->512 at 40x40, B = 64: 286 -> 127 us, tools/gpu/s2_fanout.sh)
        if (dy_conv_v4_classes_eligible(r, nc)) return dy_conv_v4_launch_classes(r, nc, stream);
        if (dy_conv_v5_classes_eligible(r, nc)) return dy_conv_v5_launch_classes(r, nc, stream);
        if (all_v2 && class_tiles <= 256) return dy_conv_v2_launch_classes(r, nc, stream);
        if (none_v2) {
          bool thin = true;
          for (int i = 0; i < nc; ++i) thin = thin && thin_eligible(&r[i], 0, true);
          if (thin) return launch_thin(&r[0], 0, (hipStream_t)stream, r, nc);
          if (class_tiles <= 256)
            return d->dtype == DY_F32 ? launch_conv<float, 0>(&r[0], (hipStream_t)stream, r, nc)
                 : (d->dtype == DY_F16 ? launch_conv<f16_t, 0>(&r[0], (hipStream_t)stream, r, nc)
                                       : launch_conv<bf16_t, 0>(&r[0], (hipStream_t)stream, r, nc));
        }
      }
      for (int i = nc - 1; i >= 0; --i) {    // heaviest class (most taps) first
        const dy_conv_desc* q = &c[i];
        int e;
        if (dy_conv_v4_eligible(q, 0)) e = dy_conv_v4_launch(q, 0, stream);
        else if (dy_conv_v5_eligible(q, 0)) e = dy_conv_v5_launch(q, 0, stream);
        else if (dy_conv_v2_eligible(q)) e = dy_conv_v2_launch(q, 0, stream);
        else e = q->dtype == DY_F32 ? launch_conv<float, 0>(q, (hipStream_t)stream)
                                    : (q->dtype == DY_F16 ? launch_conv<f16_t, 0>(q, (hipStream_t)stream) : launch_conv<bf16_t, 0>(q, (hipStream_t)stream));
        if (e) return e;
      }
      return 0;
    }
  }
  // the pixel-streaming 1x1 kernel and the two large-tile kernels add the optional `add_src` view in their epilogue
  if (dy_conv_px_eligible(d) && dy_conv_px_has_shape(d)) { *added = true; return dy_conv_px_launch(d, stream); }
  if (dy_conv_v4_eligible(d, 1)) { *added = true; return dy_conv_v4_launch(d, 1, stream); }
  if (dy_conv_v5_eligible(d, 1)) { *added = true; return dy_conv_v5_launch(d, 1, stream); }
  if (dy_conv_v3_eligible(d)) return dy_conv_v3_launch(d, 1, stream);
  if (dy_conv_v2_eligible(d)) return dy_conv_v2_launch(d, 1, stream);
  hipStream_t st = (hipStream_t)stream;
  if (thin_eligible(d, 1)) return launch_thin(d, 1, st);
  return d->dtype == DY_F32 ? launch_conv<float, 1>(d, st) : (d->dtype == DY_F16 ? launch_conv<f16_t, 1>(d, st) : launch_conv<bf16_t, 1>(d, st));
}

extern "C" int dy_conv2d_dgrad(const dy_conv_desc* d, void* stream) {
  bool added = false;
  if (int e = dgrad_dispatch(d, stream, &added)) return e;
  if (d->add_src && !added) {           // route without the fused addend: dst += add_src as its own pass
    DY_CHECK(d->dst && d->dst_planar == nullptr, "dy_conv2d_dgrad: add_src needs an NHWC destination");
    return dy_copy2d(d->add_src, d->add_src_ld, d->dst, d->dst_ld, (int64_t)d->N * d->Hd * d->Wd, d->Cd, 1, d->dtype, stream);
  }
  return 0;
}

namespace {
template <typename T>
int launch_wgrad(WgP p, float* scratch, long scratch_elems, float* g_oihw, int Cout_real, int Cin_real, hipStream_t st) {
  constexpr int MK = 8 * DT<T>::VE;
  int bm, bn;
  bn = p.Ktot <= 32 ? 32 : (p.Ktot <= 64 ? 64 : 128);
  if (bn == 32) bm = 128;
  else if (bn == 64) bm = p.Cout <= 64 ? 64 : 128;
  else bm = p.Cout <= 32 ? 32 : (p.Cout <= 64 ? 64 : 128);
  const int tiles_c = dy_cdiv(p.Cout, bm);
  p.tiles_k = dy_cdiv(p.Ktot, bn);
  const int tiles = tiles_c * p.tiles_k;
  p.tiles = tiles;
  // enough splits to fill the chip (~8 blocks per CU: swept with tools/wgrad_sweep.py on the n-scale shapes, 2048 blocks beat
  // 1024 by 7 % and 256 lose 55 %), at least 4 steps of pixels per split, and the slabs must fit
  static const int env_blocks = dy_env("DY_WGRAD_BLOCKS") ? atoi(dy_env("DY_WGRAD_BLOCKS")) : 0;     // tuning aids
  static const int env_steps = dy_env("DY_WGRAD_STEPS") ? atoi(dy_env("DY_WGRAD_STEPS")) : 0;
  const int min_steps = env_steps > 0 ? env_steps : 4;
  long max_splits = (p.M + (long)min_steps * MK - 1) / ((long)min_steps * MK);
  long want = ((env_blocks > 0 ? env_blocks : 2048) + tiles - 1) / tiles;
  long splits = want < max_splits ? want : max_splits;
  const long fit = scratch_elems / ((long)tiles * bm * bn);
  DY_CHECK(fit >= 1, "dy_conv2d_wgrad: scratch too small (%ld floats, need %ld)", scratch_elems, (long)tiles * bm * bn);
  if (splits > fit) splits = fit;
  if (splits < 1) splits = 1;
  if (splits > 65535) splits = 65535;
  long chunk = (p.M + splits - 1) / splits;
  chunk = (chunk + MK - 1) / MK * MK;
  splits = (p.M + chunk - 1) / chunk;
  p.chunk = chunk;
  p.part = scratch;
  dim3 grid(tiles, (unsigned)splits);
  dy_note_kernel("conv_wgrad_kernel+wgrad_reduce_kernel");
#define WG(BM_, BN_, WM_, WN_) conv_wgrad_kernel<T, BM_, BN_, WM_, WN_><<<grid, NTHREADS, 0, st>>>(p)
  if (bm == 128 && bn == 128) WG(128, 128, 2, 2);
  else if (bm == 64 && bn == 128) WG(64, 128, 1, 4);
  else if (bm == 32 && bn == 128) WG(32, 128, 1, 4);
  else if (bm == 128 && bn == 64) WG(128, 64, 4, 1);
  else if (bm == 64 && bn == 64) WG(64, 64, 2, 2);
  else WG(128, 32, 4, 1);
#undef WG
  DY_LAUNCH_CHECK();
  wgrad_reduce_kernel<<<dim3(Cout_real, dy_cdiv(p.Ktot, 32)), 256, 0, st>>>(scratch, (int)splits, tiles, p.tiles_k, bm, bn,
                                                                            Cin_real, p.Cin, p.KH, p.KW, p.Ktot, g_oihw);
  DY_LAUNCH_CHECK();
  return 0;
}
}  // namespace

extern "C" int dy_conv2d_wgrad(const void* x, int64_t x_ld, int N, int Hi, int Wi, int Cin_pad, const void* dz, int64_t dz_ld,
                               int Ho, int Wo, int Cout_pad, int KH, int KW, int stride, int pad, int dil, int Cout, int Cin,
                               float* scratch, int64_t scratch_elems, float* g_oihw, int dtype, void* stream) {
  DY_CHECK(x && dz && scratch && g_oihw, "dy_conv2d_wgrad: null pointer");
  DY_CHECK(dtype == DY_F32 || dtype == DY_BF16 || dtype == DY_F16, "dy_conv2d_wgrad: bad dtype");
  const int ve = dtype == DY_F32 ? 4 : 8, es = dtype == DY_F32 ? 4 : 2;
  DY_CHECK(Cin_pad % ve == 0 && Cout_pad % ve == 0, "dy_conv2d_wgrad: Cin=%d / Cout=%d must be multiples of %d", Cin_pad, Cout_pad, ve);
  DY_CHECK(Cout > 0 && Cout <= Cout_pad && Cin > 0 && Cin <= Cin_pad, "dy_conv2d_wgrad: bad real channel counts");
  DY_CHECK((x_ld * es) % 16 == 0 && (dz_ld * es) % 16 == 0, "dy_conv2d_wgrad: ld not 16-byte aligned");
  DY_CHECK(((uintptr_t)x) % 16 == 0 && ((uintptr_t)dz) % 16 == 0, "dy_conv2d_wgrad: pointer not 16-byte aligned");
  const int ho = (Hi + 2 * pad - dil * (KH - 1) - 1) / stride + 1, wo = (Wi + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  DY_CHECK(ho == Ho && wo == Wo, "dy_conv2d_wgrad: dz %dx%d does not match conv output %dx%d", Ho, Wo, ho, wo);
  if (dy_dense_wgrad_eligible(Hi, Wi, Ho, Wo, KH, KW, pad, dil))
    return dy_dense_wgrad_launch(x, x_ld, N, Hi, Wi, Cin_pad, dz, dz_ld, Cout, Cin, g_oihw, dtype, stream);
  if (dy_wgrad_v3_eligible(dtype, Cin_pad, Cout_pad, KH, KW, stride, pad, dil, N, Hi, Wi, x_ld, dz_ld, scratch_elems))
    return dy_wgrad_v3_launch(x, x_ld, N, Hi, Wi, Cin_pad, dz, dz_ld, Cout_pad, Cout, Cin, scratch, scratch_elems, g_oihw, dtype, stream);
  if (dy_wgrad_v4_eligible(dtype, Cin_pad, Cout_pad, KH, KW, (long)N * Ho * Wo, N, Hi, Wi, Ho, Wo, x_ld, dz_ld, scratch_elems))
    return dy_wgrad_v4_launch(x, x_ld, N, Hi, Wi, Cin_pad, dz, dz_ld, Ho, Wo, Cout_pad, KH, KW, stride, pad, dil, Cout, Cin, scratch,
                              scratch_elems, g_oihw, dtype, stream);
  if (dy_wgrad_v2_eligible(dtype, Cin_pad, Cout_pad, KH, KW, (long)N * Ho * Wo, x_ld, dz_ld))
    return dy_wgrad_v2_launch(x, x_ld, N, Hi, Wi, Cin_pad, dz, dz_ld, Ho, Wo, Cout_pad, KH, KW, stride, pad, dil, Cout, Cin, scratch,
                              scratch_elems, g_oihw, dtype, stream);
  WgP p;
  p.x = (const char*)x; p.x_ld = x_ld; p.N = N; p.Hi = Hi; p.Wi = Wi; p.Cin = Cin_pad;
  p.dz = (const char*)dz; p.dz_ld = dz_ld; p.Ho = Ho; p.Wo = Wo; p.Cout = Cout_pad;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.dil = dil; p.part = scratch;
  p.M = (long)N * Ho * Wo;
  p.Ktot = KH * KW * Cin_pad;
  p.pointwise = (KH == 1 && KW == 1 && stride == 1 && pad == 0) ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  return dtype == DY_F32 ? launch_wgrad<float>(p, scratch, scratch_elems, g_oihw, Cout, Cin, st)
                         : (dtype == DY_F16 ? launch_wgrad<f16_t>(p, scratch, scratch_elems, g_oihw, Cout, Cin, st)
                                            : launch_wgrad<bf16_t>(p, scratch, scratch_elems, g_oihw, Cout, Cin, st));
}

extern "C" int dy_pack_weight(const float* w, void* packed, int Cout, int Cout_pad, int Cin, int Cin_pad, int KH, int KW,
                              int transposed, int dtype, void* stream) {
  DY_CHECK(w && packed && Cin_pad >= Cin && Cout_pad >= Cout, "dy_pack_weight: bad args");
  long total = (long)Cout_pad * KH * KW * Cin_pad;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DY_F32) pack_weight_kernel<float><<<blocks, 256, 0, st>>>(w, (float*)packed, Cout, Cout_pad, Cin, Cin_pad, KH, KW, transposed);
  else if (dtype == DY_F16) pack_weight_kernel<f16_t><<<blocks, 256, 0, st>>>(w, (f16_t*)packed, Cout, Cout_pad, Cin, Cin_pad, KH, KW, transposed);
  else pack_weight_kernel<bf16_t><<<blocks, 256, 0, st>>>(w, (bf16_t*)packed, Cout, Cout_pad, Cin, Cin_pad, KH, KW, transposed);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int64_t dy_pack_item_blocks(int Cout_pad, int Cin_pad, int KH, int KW) {
  (void)KH; (void)KW;                       // one block per 32 x 32 channel tile, whatever the window
  return (int64_t)((Cout_pad + PACK_TILE - 1) / PACK_TILE) * ((Cin_pad + PACK_TILE - 1) / PACK_TILE);
}

extern "C" int dy_pack_weights_multi(const dy_pack_item* items_dev, int n_items, int64_t n_blocks, void* stream) {
  DY_CHECK(items_dev && n_items > 0 && n_blocks > 0 && n_blocks < (1LL << 31), "dy_pack_weights_multi: bad args");
  pack_multi_kernel<<<(unsigned)n_blocks, 256, 0, (hipStream_t)stream>>>(items_dev, n_items);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_unpack_wgrad(const float* dwp, float* g, int Cout, int Cin, int Cin_pad, int KH, int KW, void* stream) {
  DY_CHECK(dwp && g, "dy_unpack_wgrad: null");
  long total = (long)Cout * Cin * KH * KW;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  unpack_wgrad_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(dwp, g, Cout, Cin, Cin_pad, KH, KW);
  DY_LAUNCH_CHECK();
  return 0;
}

// ---- merged entries: the same launches as the separate calls, one foreign-function call instead of two or three.  The step of
// BASELINE configs[1] issues ~700 launches from Python at ~10 us of interpreter + ctypes time each and had become host-bound
// (tools/host_time.py: 9.6 ms to issue a step the GPU finishes in 9.9 ms).
extern "C" int dy_conv2d_bn_act_fwd(const dy_conv_desc* d, int64_t count, const float* gamma, const float* beta, float* running_mean,
                                    float* running_var, float momentum, float eps, float* aff, int act, const void* residual,
                                    int64_t res_ld, void* y, int64_t y_ld, void* stream) {
  DY_CHECK(d && d->stats && d->dst && aff, "dy_conv2d_bn_act_fwd: needs a raw-output buffer, statistics and the affine buffer");
  if (int e = dy_conv2d_fwd(d, stream)) return e;
  const int C = d->Cd;
  if (int e = dy_bn_finalize(d->stats, count, gamma, beta, running_mean, running_var, momentum, eps, aff, aff + C, aff + 2 * C,
                             aff + 3 * C, C, stream))
    return e;
  return dy_bn_act_fwd(d->dst, d->dst_ld, aff, aff + C, act, residual, res_ld, y, y_ld, (int64_t)d->N * d->Hd * d->Wd, C, d->dtype, stream);
}

extern "C" int dy_conv2d_wgrad_forked(void* wait_for, const void* x, int64_t x_ld, int N, int Hi, int Wi, int Cin_pad, const void* dz,
                                      int64_t dz_ld, int Ho, int Wo, int Cout_pad, int KH, int KW, int stride, int pad, int dil,
                                      int Cout, int Cin, float* scratch, int64_t scratch_elems, float* g_oihw, int dtype,
                                      void* stream) {
  if (wait_for != stream)                      // (a null handle is a stream too: the default stream)
    if (int e = dy_stream_fork(wait_for, stream)) return e;
  return dy_conv2d_wgrad(x, x_ld, N, Hi, Wi, Cin_pad, dz, dz_ld, Ho, Wo, Cout_pad, KH, KW, stride, pad, dil, Cout, Cin, scratch,
                         scratch_elems, g_oihw, dtype, stream);
}
