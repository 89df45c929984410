// Pipelined implicit-GEMM convolution for the wide-channel bf16 layers (YOLOv8-L stages): forward and data-gradient.
//
// Measured on MI355X with tools/conv_bench: the register-staged kernel of conv.hip spends 45 % of its time outside both the
// MFMA pipe and HBM (ds_write staging pass, two barriers per K-step, address VALU).  This kernel removes that pass:
//   * A (gathered activations) and B (packed weights) tiles go global -> LDS directly (global_load_lds, 16 B per lane, the
//     per-lane SOURCE address does the implicit-GEMM gather; padding taps read a zero page), no VGPR staging, no ds_write;
//   * NSTAGE-deep LDS ring, ONE raw s_barrier per K-step, counted s_waitcnt vmcnt(N) (with 3 stages the next stage stays in
//     flight across the barrier); the shipped shapes use 2 stages so that TWO blocks are co-resident per CU (see the launcher);
//   * BM x BN block tile on (BM/64) x 2 waves, K-step 64 bf16 = 128-byte LDS rows, XOR swizzle (16-byte slot ^ (row>>1)&7)
//     applied on the source side (the DMA image is lane-linear) and on the ds_read_b128 side: conflict-free fragments;
//   * v_mfma_f32_32x32x16_bf16, f32 accumulate; BatchNorm batch statistics from the accumulators (replicated f64 atomics).
// Requirements (checked by the dispatcher): bf16, Cs % 64 == 0 (a K-step never straddles two taps), dense 16-byte aligned views.
#include <stdlib.h>
#include "dy_common.h"
#include "conv_epilogue.h"
#include "../../include/dedark_yolo.h"

namespace v2 {

constexpr int BK = 64;
constexpr int ROW = 128;                 // bytes per LDS row (64 bf16)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __attribute__((aligned(16))) unsigned char g_zero_page[16];
__device__ unsigned long long g_stamps[16];       // diagnostics (DY_ABLATE & 32): s_memtime at phase boundaries, block 0 / wave 0

__device__ inline void stamp(int ablate, int i) {
  if ((ablate & 32) && blockIdx.x == 0 && threadIdx.x == 0) {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    g_stamps[i] = t;
  }
}

struct P {
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
  long M;
  int Ktot;
  int tiles_n, nblk;
  int ablate;     // DY_ABLATE (diagnostics): 1 no loads after the prologue, 2 no MFMA, 4 no stores, 16 no LDS fragment reads
  long dst_row, dst_img;          // destination row / image strides in elements (dst_row == 0: dense)
  int kh0, khs, kw0, kws, KWf;    // window tap -> weight tap mapping (tap subsets of a parity-split data gradient)
  long w_row;                     // elements per output-channel row of the weight pack
  int ncls;                       // > 1: parity classes of a stride-2 data gradient in one launch
  int f16;                        // payload is IEEE half instead of bf16 (host side: selects the instantiation)
  DyParityCls cls[4];
};

__device__ inline long dst_offset(const P& p, long m) {
  if (p.dst_row == 0) return m * p.dst_ld;
  const long HWd = (long)p.Hd * p.Wd;
  const long img = m / HWd;
  const int rem = (int)(m - img * HWd);
  const int oh = rem / p.Wd, ow = rem - oh * p.Wd;
  return img * p.dst_img + (long)oh * p.dst_row + (long)ow * p.dst_ld;
}

__device__ inline int xcd_remap(int bid, int nblk) {
  int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

// SMALLC: Cs is not a multiple of 64 (the n-scale layers, 8..48 channels): a 64-wide K-step then spans several taps, so every
// lane derives (tap, channel) of ITS 16-byte chunk; K = KH*KW*Cs is padded to the step with zero-page loads.
// BM x BN block tile on (BM/64) x 2 waves (wave tile 64 x BN/2); NSTAGE-deep LDS ring.
template <int BM, int BN, int MODE, bool SMALLC, int NSTAGE, typename T = bf16_t>
__global__ __launch_bounds__(BM * 2) void conv_kernel(const P pk) {
  P p = pk;
  constexpr int WN = 2, WM = BM / 64, NW = WM * WN, NT = 64 * NW;
  constexpr int TM = BM / WM / 32;          // 2
  constexpr int TN = BN / WN / 32;          // 2 (BN=128) or 1 (BN=64)
  constexpr int A_LD = BM * 8 / NT;         // glds per thread for A per stage (4)
  constexpr int B_LD = BN * 8 / NT;         // 2 or 1
  constexpr int STAGE = (BM + BN) * ROW;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  stamp(DY_ABLATE_OF(p), 0);
  const int wm = wave >> 1, wn = wave & 1;
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
  const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
  const long m0 = (long)tile_m * BM;
  const int n0 = tile_n * BN;

  // ---- DMA bookkeeping: instruction j of this wave fills rows 8*(wave + 8j) .. +7 of the tile, lane -> (row, slot)
  const int lrow = lane >> 3, slot = lane & 7;
  const int chunk = slot ^ (((4 * wave) + (lane >> 4)) & 7);       // logical 16-byte chunk of the row this lane fetches
  const char* a_base[A_LD];
  int a_h[A_LD], a_w[A_LD];
  bool a_ok[A_LD];
  const DyTileWalk walk(m0, p.Hd, p.Wd);      // (64-bit divisions per row cost more than the 3 K-steps of the stem layer's tile)
#pragma unroll
  for (int j = 0; j < A_LD; ++j) {
    const int r = 8 * (wave + NW * j) + lrow;
    a_ok[j] = m0 + r < p.M;
    int img, oh, ow;
    walk.at(r, img, oh, ow);
    a_base[j] = p.src + ((long)img * p.Hs * p.Ws * p.src_ld + (SMALLC ? 0 : chunk * 8)) * 2;
    if (MODE == 0) {
      a_h[j] = oh * p.stride - p.pad;
      a_w[j] = ow * p.stride - p.pad;
    } else {
      a_h[j] = oh + p.pad;
      a_w[j] = ow + p.pad;
    }
  }
  const char* b_ptr[B_LD];
  bool b_ok[B_LD];
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    int n = n0 + 8 * (wave + NW * j) + lrow;
    b_ok[j] = n < p.Cd;
    b_ptr[j] = p.w + ((long)(b_ok[j] ? n : 0) * p.w_row + (SMALLC ? 0 : chunk * 8)) * 2;
  }
  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  int kh = 0, kw = 0, ci = 0;              // tap / channel offset of the NEXT stage to issue
  int kstep = 0;

  auto issue = [&](int buf) {
    char* stage = smem + buf * STAGE;
    bool kvalid = true;
    if (SMALLC) {                            // per-lane tap of this lane's chunk
      const int kk = kstep * BK + chunk * 8;
      kvalid = kk < p.Ktot;
      const int tap = kvalid ? kk / p.Cs : 0;
      ci = kvalid ? kk - tap * p.Cs : 0;
      kh = tap / p.KW;
      kw = tap - kh * p.KW;
      ++kstep;
    }
    // otherwise the tap geometry is uniform for the whole K-step (Cs % 64 == 0)
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
      int sh, sw;
      bool ok = a_ok[j] && kvalid;
      if (MODE == 0) {
        sh = a_h[j] + kh * p.dil;
        sw = a_w[j] + kw * p.dil;
      } else {
        int th = a_h[j] - kh * p.dil, tw = a_w[j] - kw * p.dil;
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
      const char* g = ok ? a_base[j] + (((long)sh * p.Ws + sw) * p.src_ld + ci) * 2 : zero;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(stage + (wave + NW * j) * 1024), 16, 0, 0);
    }
    const long wk = ((long)((p.kh0 + p.khs * kh) * p.KWf + p.kw0 + p.kws * kw) * p.Cs + ci) * 2;
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      const char* g = (b_ok[j] && kvalid) ? b_ptr[j] + wk : zero;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(stage + BM * ROW + (wave + NW * j) * 1024), 16, 0, 0);
    }
    if (!SMALLC) {
      ci += BK;
      if (ci >= p.Cs) {
        ci = 0;
        if (++kw == p.KW) { kw = 0; ++kh; }
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

  const int nsteps = (p.Ktot + BK - 1) / BK;
  const int fr = lane & 31, fh = lane >> 5;
  // fragment row byte offsets (swizzle key (row>>1)&7 is per row)
  int a_off[TM], a_key[TM], b_off[TN], b_key[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    int row = wm * (BM / WM) + i * 32 + fr;
    a_off[i] = row * ROW;
    a_key[i] = (row >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    int row = wn * (BN / WN) + j * 32 + fr;
    b_off[j] = BM * ROW + row * ROW;
    b_key[j] = (row >> 1) & 7;
  }

  stamp(DY_ABLATE_OF(p), 1);
  issue(0);
  if (NSTAGE > 2 && nsteps > 1) issue(1);
  stamp(DY_ABLATE_OF(p), 2);
  for (int s = 0; s < nsteps; ++s) {
    if (s == 1) stamp(DY_ABLATE_OF(p), 3);
    if (s == 9) stamp(DY_ABLATE_OF(p), 4);
    if (NSTAGE > 2 && s + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_LD + B_LD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s + NSTAGE - 1 < nsteps && !(DY_ABLATE_OF(p) & 1)) issue((s + NSTAGE - 1) % NSTAGE);
    const char* stage = smem + (s % NSTAGE) * STAGE;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int c = 2 * kk + fh;
      u32x4 af[TM], bf[TN];
      if (DY_ABLATE_OF(p) & 16) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = u32x4{(unsigned)s, 1u, 2u, 3u};
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = u32x4{(unsigned)kk, 1u, 2u, 3u};
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const u32x4*>(stage + a_off[i] + ((c ^ a_key[i]) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const u32x4*>(stage + b_off[j] + ((c ^ b_key[j]) << 4));
      }
      if (DY_ABLATE_OF(p) & 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(af[i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(bf[j]));
        continue;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = mfma_32x32x16<T>(af[i], bf[j], acc[i][j]);
    }
  }

  stamp(DY_ABLATE_OF(p), 5);
  // ---- epilogue (conv_epilogue.h): accumulators -> transposed bf16 image in the idle ring -> 16-byte stores through
  // ds_read_b64_tr_b16; csum / csq = per-column sums of the raw accumulators for the BatchNorm statistics below
  const int cl = lane & 31, hh = lane >> 5;
  float csum[TN], csq[TN];
  if (!(DY_ABLATE_OF(p) & 4))
    dy_epi::store_tile<BM, BN, WM, WN, TM, TN>(smem, acc, wm, wn, lane, wave, m0, n0, p.M, p.Cd, p.scale, p.shift, p.act, p.accumulate,
                                   reinterpret_cast<T*>(p.dst), [&](long m) { return dst_offset(p, m); }, csum, csq);
  stamp(DY_ABLATE_OF(p), 6);
  stamp(DY_ABLATE_OF(p), 7);
  if (p.stats) {
    __syncthreads();                              // the bf16 image has been consumed: reuse LDS for the column sums
    float* red = reinterpret_cast<float*>(smem);  // [WM][BN][2]
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
        double* st = p.stats + (long)(tile_m % DY_STATS_REPLICAS) * 2 * p.Cd;
        atomic_add_f64(st + n, (double)s1);
        atomic_add_f64(st + p.Cd + n, (double)s2);
      }
    }
  }
}

template <int BN, int MODE, bool SMALLC, int NSTAGE, int BM, typename T>
int launch_t(P& p, hipStream_t st) {
  constexpr int RING = NSTAGE * (BM + BN) * ROW, EPI = dy_epi::image_bytes<BM, BN>();
  constexpr int SHMEM = RING > EPI ? RING : EPI;            // the epilogue image reuses the ring
  static_assert(SHMEM <= 160 * 1024, "LDS budget");
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_kernel<BM, BN, MODE, SMALLC, NSTAGE, T>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SHMEM);
    if (e != hipSuccess) {
      dy_set_error("conv_v2: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 3;
    }
    configured = true;
  }
  p.tiles_n = dy_cdiv(p.Cd, BN);
  p.nblk = dy_cdiv(p.M, BM) * p.tiles_n;
  if (p.ncls > 1) {
    long acc = 0;
    for (int c = 0; c < p.ncls; ++c) {
      p.cls[c].blk0 = (int)acc;
      acc += dy_cdiv(p.cls[c].M, BM) * p.tiles_n;
    }
    p.nblk = (int)acc;
  }
  static char name[64];
  if (!name[0]) snprintf(name, sizeof(name), "v2::conv_kernel<%d, %d, %d, %s, %d>", BM, BN, MODE, SMALLC ? "true" : "false", NSTAGE);
  dy_note_kernel(name);
  conv_kernel<BM, BN, MODE, SMALLC, NSTAGE, T><<<p.nblk, BM * 2, SHMEM, st>>>(p);
  DY_LAUNCH_CHECK();
  return 0;
}

template <int BN, int MODE, bool SMALLC, int NSTAGE = 3, int BM = 256>
int launch(P& p, hipStream_t st) {
  return p.f16 ? launch_t<BN, MODE, SMALLC, NSTAGE, BM, f16_t>(p, st) : launch_t<BN, MODE, SMALLC, NSTAGE, BM, bf16_t>(p, st);
}

}  // namespace v2

// true when the pipelined kernel can take this problem (the dispatcher in conv.hip falls back to the generic kernel otherwise)
extern "C" int dy_debug_conv_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(v2::g_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : 1;
}

// 256 x 256 tiles pay off only when they still fill the chip (256 CUs, one block each): 256->256 3x3 at 20x20 with B = 64 is just
// 100 such tiles and ran at 93 us against 56 us on the band kernel.
bool dy_conv_prefers_256(const dy_conv_desc* d) {
  const long M = (long)d->N * d->Hd * d->Wd;
  const long tn = (d->Cd + 255) / 256;
  // ... and when the channel tiles are not mostly padding (Cd = 320: 2 x 256 covers 1.6x the channels; 1x1 128->320 dgrad at 160x160
  // runs 1031 us with 256-wide tiles, 830 us with 128-wide ones)
  return d->Cd >= 256 && ((M + 255) / 256) * tn >= 192 && tn * 256 * 4 <= (long)d->Cd * 5;
}

bool dy_conv_v2_eligible(const dy_conv_desc* d) {
  static const bool off = dy_env("DY_NO_CONV_V2") != nullptr;
  if (off) return false;
  const long M = (long)d->N * d->Hd * d->Wd;
  if (!((d->dtype == DY_BF16 || d->dtype == DY_F16) && M >= 2048 && (d->src_ld * 2) % 16 == 0)) return false;
  if (d->Cs % 64 == 0) return d->Cd >= 64;
  // narrow SOURCE channels (per-lane tap decode, SMALLC).  Measured on the n-scale layers: a win only when the destination is
  // at least one 64-wide tile (96->64 1x1: 50 -> 40 us, 32->64 3x3 s2: 61 -> 54 us); with Cd <= 32 the 256x64 tile is 50-75 %
  // padding and its epilogue-bound blocks lose to the register-staged kernel (32->32 3x3: 31 -> 49 us).  DY_V2_SMALLC=1 forces it.
  static const bool force = dy_env("DY_V2_SMALLC") != nullptr;
  return d->Cs % 8 == 0 && d->Cd % 8 == 0 && (d->Cd >= 64 || (force && d->Cd >= 8));
}

static int v2_launch_impl(const dy_conv_desc* d, int mode, const dy_conv_desc* classes, int ncls, void* stream);

int dy_conv_v2_launch(const dy_conv_desc* d, int mode, void* stream) { return v2_launch_impl(d, mode, nullptr, 0, stream); }

// `classes[0..ncls)`: forward-style problems that differ only in destination offset, grid extent, tap subset and pad (the parity
// classes of a stride-2 data gradient); one launch, blocks ordered class by class.
int dy_conv_v2_launch_classes(const dy_conv_desc* classes, int ncls, void* stream) {
  return v2_launch_impl(&classes[0], 0, classes, ncls, stream);
}

static int v2_launch_impl(const dy_conv_desc* d, int mode, const dy_conv_desc* classes, int ncls, void* stream) {
  v2::P p;
  p.f16 = d->dtype == DY_F16;
  p.ncls = 0;
  if (ncls > 1) {
    DY_CHECK(ncls <= 4, "conv_v2: at most 4 classes");
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
  hipStream_t st = (hipStream_t)stream;
  const bool wide = d->Cd > 64;
  if (d->Cs % 64 != 0) {
    if (mode == 0) return wide ? v2::launch<128, 0, true, 2, 128>(p, st) : v2::launch<64, 0, true, 2, 256>(p, st);
    return wide ? v2::launch<128, 1, true, 2, 128>(p, st) : v2::launch<64, 1, true, 2, 256>(p, st);
  }
  // Tile shapes (swept with tools/conv_bench, DY_V2_EXP): two CO-RESIDENT blocks per CU beat one bigger block with a deeper
  // ring -- their barriers and epilogue bursts interleave.  Wide outputs: 128x128 tile on 4 waves, 2 stages = 64 KB (2 blocks /
  // CU; 256->256 3x3 254 -> 239 us, 1280->512 1x1 302 -> 286 us vs 256x128 x 3 stages).  Cd <= 64: 256x64 on 8 waves, 2 stages =
  // 80 KB (64->64 3x3 at 160x160: 433 -> 333 us, also 20 % faster than the band kernel's 64-wide variant).
  static const int exp_mode = dy_env("DY_V2_EXP") ? atoi(dy_env("DY_V2_EXP")) : 0;
#define DY_V2_GO(BN_, NS_, BM_) (mode == 0 ? v2::launch<BN_, 0, false, NS_, BM_>(p, st) : v2::launch<BN_, 1, false, NS_, BM_>(p, st))
  if (exp_mode == 1) return wide ? DY_V2_GO(128, 3, 256) : DY_V2_GO(64, 3, 256);        // the first version: one block per CU
  // >= 256 output channels: 256 x 256 tile (one block per CU, 2 stages = 128 KiB).  These kernels stream both operands from L2 /
  // Infinity Cache every step, so the tile's flop-per-byte (128 vs 64 for 128 x 128) outweighs co-residency here.
  if (dy_conv_prefers_256(d) && exp_mode != 2) return DY_V2_GO(256, 2, 256);
  if (exp_mode == 3 && wide) return DY_V2_GO(128, 2, 512);      // experiment: 512 x 128 tile on 16 waves, exactly 160 KiB
  // (few output pixels, wide: a 4-stage ring, 64-row tiles or 64x64 tiles were all slower or equal for the 100-block launches of
  //  YOLOv8-n's 20x20 layers -- 128->128 3x3, B = 32: 29.1 us as is, 29.9 / 29.6 / 38.4 us)
  // Launches with less than one block per CU (YOLOv8-n's 20x20 layers at B = 32: 100 blocks) were tried with a 3-stage ring,
  // with 64-row tiles and with 128x64 tiles for Cd <= 64: all equal or slower (128->128 3x3: 29.2 us as is, 29.9 / 29.6 us;
  // 256->64 3x3: 45.7 us as is, 50.1 / 65.5 us), so the shapes below stay.
  return wide ? DY_V2_GO(128, 2, 128) : DY_V2_GO(64, 2, 256);
#undef DY_V2_GO
}
