// 3x3 / stride-1 / pad-1 convolution with spatial reuse in LDS ("band" kernel), bf16, forward and data-gradient.
//
// Why: in-kernel s_memtime stamps of the pipelined implicit-GEMM kernel (conv_v2.hip) show 2,900 cycles per K-step against
// 1,024 cycles of MFMA work: a CU ingests only ~14-16 B/clk from L2 / Infinity Cache, and an implicit-GEMM tile re-fetches every
// input pixel once per tap (9x).  For a stride-1 3x3 conv on an NHWC tensor the pixels needed by BM consecutive output pixels
// are ONE contiguous range of the flattened (n,h,w) index: [m0-(W+1), m0+BM+(W+1)).  This kernel stages that band once per
// 64-channel chunk and reads all 9 taps out of LDS (fragment row = output row + kh*W + kw); taps that fall outside the image
// are zeroed in registers by a per-row 9-bit validity mask.  Per 9 K-steps a CU now ingests band (BM+2W+2)*128 B + 9 weight
// tiles instead of 9*(BM+BN)*128 B: 2.3x less for the stage-3 shapes.
//   * band chunks are double-buffered: the next chunk's band is fetched in 8 slices, one per tap step, behind counted vmcnt waits;
//   * weight tiles: 3-stage ring as in conv_v2; one raw s_barrier per step; global_load_lds 16 B, XOR-swizzled 128-byte rows;
//   * 256 x BN tile, 8 waves (4 x 2), v_mfma_f32_32x32x16_bf16; epilogue through LDS with 16-byte stores; BN batch statistics.
// Data gradient (MODE 1) is the same walk with the taps mirrored: dx[m] += dz[m + (1-kh)*W + (1-kw)] * Wt[kh][kw].
#include <stdlib.h>
#include "dy_common.h"
#include "conv_epilogue.h"
#include "../../include/dedark_yolo.h"

namespace v3 {

constexpr int BM = 256, BK = 64, NT = 512, NB = 3;
constexpr int ROW = 128;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __attribute__((aligned(16))) unsigned char g_zero_page[16];
__device__ unsigned long long g_stamps[16];       // diagnostics (DY_ABLATE & 32)

__device__ inline void stamp(int ablate, int i) {
  if ((ablate & 32) && blockIdx.x == 0 && threadIdx.x == 0) {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    g_stamps[i] = t;
  }
}

struct P {
  const char* src;      // [N*H*W][src_ld] bf16
  long src_ld;
  int N, H, W, Cs;
  const char* w;        // packed [Cd][3][3][Cs]
  char* dst;
  long dst_ld;
  int Cd;
  const float* scale;
  const float* shift;
  int act;
  double* stats;
  int accumulate;
  long M;               // N*H*W
  int Ktot;             // 9*Cs
  int tiles_n, nblk;
  int L;                // band rows = BM + 2*(W+1)
  int a_bytes;          // bytes of one band buffer (L rounded up to 64 rows)
  int nslices;          // ceil(L / 64)
  int ablate;           // DY_ABLATE diagnostics: 1 no band prefetch, 64 no weight-tile loads in the loop, 2 no MFMA, 32 stamps
  int f16;              // payload is IEEE half instead of bf16 (host side: selects the instantiation)
};

__device__ inline int xcd_remap(int bid, int nblk) {
  int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

template <int BN, int MODE, typename T = bf16_t>
__global__ __launch_bounds__(NT) void conv3x3_kernel(P p) {
  constexpr int WN = 2, WM = 4;
  constexpr int TM = BM / WM / 32;          // 2
  constexpr int TN = BN / WN / 32;          // 2 or 1
  constexpr int B_LD = BN * 8 / NT;         // weight-tile glds per thread per stage
  constexpr int BSTAGE = BN * ROW;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  stamp(DY_ABLATE_OF(p), 0);
  const int wm = wave >> 1, wn = wave & 1;
  const int bid = xcd_remap(blockIdx.x, p.nblk);
  const int tile_m = bid / p.tiles_n, tile_n = bid - tile_m * p.tiles_n;
  const long m0 = (long)tile_m * BM;
  const int n0 = tile_n * BN;
  const int W = p.W;
  const int nchunks = p.Cs / BK;
  char* Abuf0 = smem;
  char* Abuf1 = smem + p.a_bytes;
  char* Bring = smem + (nchunks > 1 ? 2 : 1) * p.a_bytes;

  // ---- DMA lane geometry: one instruction fills 8 rows x 128 B; lane -> (row-in-group, 16-byte slot); source chunk swizzled
  const int lrow = lane >> 3, slot = lane & 7;
  const int chunk = slot ^ (((4 * wave) + (lane >> 4)) & 7);
  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  const long band0 = m0 - (W + 1);          // flattened input pixel of band row 0

  // slice q of a band: rows [64q, 64q+64); this thread's row = 64q + 8*wave + lrow
  auto issue_band_slice = [&](char* abuf, int q, int ci) {
    const int row = 64 * q + 8 * wave + lrow;
    const long g = band0 + row;
    const bool ok = row < p.L && g >= 0 && g < p.M;
    const char* src = ok ? p.src + (g * p.src_ld + ci + chunk * 8) * 2 : zero;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(abuf + (64 * q + 8 * wave) * ROW), 16, 0, 0);
  };
  const char* b_ptr[B_LD];
  bool b_ok[B_LD];
#pragma unroll
  for (int j = 0; j < B_LD; ++j) {
    int n = n0 + 8 * (wave + 8 * j) + lrow;
    b_ok[j] = n < p.Cd;
    b_ptr[j] = p.w + ((long)(b_ok[j] ? n : 0) * p.Ktot + chunk * 8) * 2;
  }
  // weights for step s = (chunk c, tap t): k offset = t*Cs + c*64
  auto issue_b = [&](int c, int t) {
    char* stage = Bring + (t % NB) * BSTAGE;
    const long koff = ((long)t * p.Cs + c * BK) * 2;
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
      const char* g = b_ok[j] ? b_ptr[j] + koff : zero;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(stage + (wave + 8 * j) * 1024), 16, 0, 0);
    }
  };

  // ---- fragment geometry
  const int fr = lane & 31, fh = lane >> 5;
  int a_row[TM];          // tile row of this lane's A fragments
  unsigned a_mask[TM];    // bit t set: tap t of that output pixel reads a real input pixel
  const long HW = (long)p.H * W;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    a_row[i] = wm * (BM / WM) + i * 32 + fr;
    const long m = m0 + a_row[i];
    unsigned mk = 0;
    if (m < p.M) {
      const int rem = (int)(m % HW);
      const int h = rem / W, w = rem - h * W;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int dh = (MODE == 0) ? t / 3 - 1 : 1 - t / 3, dw = (MODE == 0) ? t % 3 - 1 : 1 - t % 3;
        if (h + dh >= 0 && h + dh < p.H && w + dw >= 0 && w + dw < W) mk |= 1u << t;
      }
    }
    a_mask[i] = mk;
  }
  int b_off[TN], b_key[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    int row = wn * (BN / WN) + j * 32 + fr;
    b_off[j] = row * ROW;
    b_key[j] = (row >> 1) & 7;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- prologue: whole band of chunk 0, weight tiles of steps 0 and 1
  stamp(DY_ABLATE_OF(p), 1);
  for (int q = 0; q < p.nslices; ++q) issue_band_slice(Abuf0, q, 0);
  issue_b(0, 0);
  issue_b(0, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  stamp(DY_ABLATE_OF(p), 2);

  // Ping-pong schedule.  Every step has a memory half (issue the DMA of tile +2 / one band slice, read the 16 fragments of
  // this step from LDS, wait for the loads of the previous step) and a matrix half (16 MFMAs), each closed by a barrier.
  // Waves 4-7 share their SIMDs with waves 0-3 and run half a step late (one extra leading barrier), so one partner is in
  // its matrix half while the other issues memory / LDS work: with every wave in lock-step (one barrier per step) the
  // matrix pipe idled ~55 % of a step (stamps: 2,200 cycles per step against 1,024 cycles of MFMA).
  // The 9 taps of a chunk are unrolled: tap, ring slot (9 % 3 == 0), tap shift and mask bit are compile-time constants.
  // Hand-off rule: after the barrier that closes a wave's memory half of step q, that wave's parts of tile q+1 have landed
  // (counted vmcnt), so tile q is complete for both groups when its first reader (group A, step q) starts.
  const bool late = wave >= 4;
  if (late) __builtin_amdgcn_s_barrier();
  const int zrow_off = p.L * ROW;          // a band row that is always zero (the DMA pads the band to a 64-row multiple > L)
  for (int c = 0; c < nchunks; ++c) {
    const bool more = c + 1 < nchunks;
    const char* abuf = (c & 1) ? Abuf1 : Abuf0;
    char* anext = (c & 1) ? Abuf0 : Abuf1;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (c == 0 && t == 1) stamp(DY_ABLATE_OF(p), 3);
      if (c == 1 && t == 0) stamp(DY_ABLATE_OF(p), 4);
      // -------- memory half
      bool tile_issued = false, slice_issued = false;
      if (!(DY_ABLATE_OF(p) & 64)) {
        if (t + 2 < 9) { issue_b(c, t + 2); tile_issued = true; }           // ring slot (t+2)%3: last read in step-1
        else if (more) { issue_b(c + 1, t + 2 - 9); tile_issued = true; }
      }
      if (more && t < 8 && !(DY_ABLATE_OF(p) & 1)) {
        issue_band_slice(anext, t < p.nslices ? t : p.nslices - 1, (c + 1) * BK);
        slice_issued = true;
      }
      const char* bst = Bring + (t % NB) * BSTAGE;
      const int shift = (MODE == 0) ? (t / 3) * W + (t % 3) : (2 - t / 3) * W + (2 - t % 3);
      int a_off[TM], a_key[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int br = a_row[i] + shift;
        const bool ok = (a_mask[i] >> t) & 1u;
        a_off[i] = ok ? br * ROW : zrow_off;               // taps outside the image read the zero row
        a_key[i] = ok ? (br >> 1) & 7 : 0;
      }
      u32x4 af[4][TM], bf[4][TN];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int cc = 2 * kk + fh;
        if (DY_ABLATE_OF(p) & 16) {
#pragma unroll
          for (int i = 0; i < TM; ++i) af[kk][i] = u32x4{(unsigned)c, 1u, 2u, 3u};
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[kk][j] = u32x4{(unsigned)kk, 1u, 2u, 3u};
          continue;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) af[kk][i] = *reinterpret_cast<const u32x4*>(abuf + a_off[i] + ((cc ^ a_key[i]) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[kk][j] = *reinterpret_cast<const u32x4*>(bst + b_off[j] + ((cc ^ b_key[j]) << 4));
      }
      // everything older than what this step just issued must have landed (it is read by the other group one phase from now)
      if (tile_issued && slice_issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(B_LD + 1) : "memory");
      else if (tile_issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(B_LD) : "memory");
      else if (slice_issued) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // -------- matrix half
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (DY_ABLATE_OF(p) & 2) {
#pragma unroll
          for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(af[kk][i]));
#pragma unroll
          for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(bf[kk][j]));
          continue;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = mfma_32x32x16<T>(af[kk][i], bf[kk][j], acc[i][j]);
      }
      __builtin_amdgcn_s_barrier();
    }
  }
  if (!late) __builtin_amdgcn_s_barrier();

  stamp(DY_ABLATE_OF(p), 5);
  // ---- epilogue (conv_epilogue.h): accumulators -> transposed bf16 image in the idle ring -> 16-byte stores through
  // ds_read_b64_tr_b16; csum / csq = per-column sums of the raw accumulators for the BatchNorm statistics below
  const int cl = lane & 31, hh = lane >> 5;
  float csum[TN], csq[TN];
  if (!(DY_ABLATE_OF(p) & 4))
    dy_epi::store_tile<BM, BN, 4, 2, TM, TN>(smem, acc, wm, wn, lane, wave, m0, n0, p.M, p.Cd, p.scale, p.shift, p.act, p.accumulate,
                                   reinterpret_cast<T*>(p.dst), [&](long m) { return m * p.dst_ld; }, csum, csq);
  stamp(DY_ABLATE_OF(p), 6);
  stamp(DY_ABLATE_OF(p), 7);
  if (p.stats) {
    __syncthreads();
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

constexpr int LDS_LIMIT = 160 * 1024;

template <int BN>
int shmem_bytes(const P& p) {
  const int nchunks = p.Cs / BK;
  int ring = (nchunks > 1 ? 2 : 1) * p.a_bytes + NB * BN * ROW;
  int epi = dy_epi::image_bytes<BM, BN>();
  return ring > epi ? ring : epi;
}

template <int BN, int MODE, typename T>
int launch_t(P& p, hipStream_t st) {
  const int shm = shmem_bytes<BN>(p);
  static int configured = 0;
  if (shm > configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_kernel<BN, MODE, T>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_LIMIT);
    if (e != hipSuccess) {
      dy_set_error("conv_v3: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 3;
    }
    configured = LDS_LIMIT;
  }
  p.tiles_n = dy_cdiv(p.Cd, BN);
  p.nblk = dy_cdiv(p.M, BM) * p.tiles_n;
  dy_note_kernel(BN == 128 ? (MODE ? "v3::conv3x3_kernel<128, 1>" : "v3::conv3x3_kernel<128, 0>") : (MODE ? "v3::conv3x3_kernel<64, 1>" : "v3::conv3x3_kernel<64, 0>"));
  conv3x3_kernel<BN, MODE, T><<<p.nblk, NT, shm, st>>>(p);
  DY_LAUNCH_CHECK();
  return 0;
}

template <int BN, int MODE>
int launch(P& p, hipStream_t st) {
  return p.f16 ? launch_t<BN, MODE, f16_t>(p, st) : launch_t<BN, MODE, bf16_t>(p, st);
}

}  // namespace v3

static void v3_fill(const dy_conv_desc* d, v3::P& p) {
  p.src = (const char*)d->src; p.src_ld = d->src_ld; p.N = d->N; p.H = d->Hs; p.W = d->Ws; p.Cs = d->Cs;
  p.w = (const char*)d->w; p.dst = (char*)d->dst; p.dst_ld = d->dst_ld; p.Cd = d->Cd;
  p.scale = d->scale; p.shift = d->shift; p.act = d->act; p.stats = d->stats; p.accumulate = d->accumulate;
  p.M = (long)d->N * d->Hs * d->Ws;
  p.f16 = d->dtype == DY_F16;
  p.Ktot = 9 * d->Cs;
  p.L = v3::BM + 2 * (d->Ws + 1);
  p.nslices = (p.L + 64) / 64;      // >= one spare row past L: the always-zero row used for out-of-image taps
  p.a_bytes = p.nslices * 64 * v3::ROW;
  static const int ablate = dy_env("DY_ABLATE") ? atoi(dy_env("DY_ABLATE")) : 0;
  p.ablate = ablate;
}

extern "C" int dy_debug_conv3_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(v3::g_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : 1;
}

// the band kernel takes 3x3 / stride 1 / pad 1 / dil 1 bf16 convs whose band (and its double buffer) fit in LDS
bool dy_conv_prefers_256(const dy_conv_desc* d);      // conv_v2.hip

bool dy_conv_v3_eligible(const dy_conv_desc* d) {
  static const bool off = dy_env("DY_NO_CONV_V3") != nullptr;
  if (off) return false;
  if (!((d->dtype == DY_BF16 || d->dtype == DY_F16) && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->dil == 1)) return false;
  if (d->KHf != 0 || d->dst_row_stride != 0) return false;    // tap subsets / strided destinations: generic kernels only
  // Cd <= 64 goes to conv_v2's 256x64 two-blocks-per-CU configuration (20 % faster than the 64-wide band variant) and
  // Cd >= 256 with enough tiles to its 256x256 tile (256->256 at 40x40: 190 us against 221 us here)
  if (dy_conv_prefers_256(d)) return false;
  if (!(d->Cs % 64 == 0 && d->Cd > 64 && d->Hs == d->Hd && d->Ws == d->Wd && (d->src_ld * 2) % 16 == 0)) return false;
  if ((long)d->N * d->Hs * d->Ws < 2048) return false;
  v3::P p;
  v3_fill(d, p);
  const int nchunks = d->Cs / v3::BK;
  if (nchunks > 1 && p.nslices > 8) return false;      // the next band must arrive in the 8 prefetch slices
  const int bn = d->Cd > 64 ? 128 : 64;
  const int ring = (nchunks > 1 ? 2 : 1) * p.a_bytes + v3::NB * bn * v3::ROW;
  return ring <= v3::LDS_LIMIT;
}

int dy_conv_v3_launch(const dy_conv_desc* d, int mode, void* stream) {
  v3::P p;
  v3_fill(d, p);
  hipStream_t st = (hipStream_t)stream;
  const bool wide = d->Cd > 64;
  if (mode == 0) return wide ? v3::launch<128, 0>(p, st) : v3::launch<64, 0>(p, st);
  return wide ? v3::launch<128, 1>(p, st) : v3::launch<64, 1>(p, st);
}
