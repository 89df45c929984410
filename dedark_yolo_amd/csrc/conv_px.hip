// Pixel-streaming 1x1 convolution for the short-K layers of the high-resolution stages (C2f cv1 / cv2 of yolov8*.yaml at 160x160 and their
// data gradients; reference ultralytics/nn/modules/conv.py:38-55, block.py:373-393):  y[m][n] = sum_k x[m][k] * w[n][k],  K <= 320.
//
// These layers move 0.8-1.5 GB for 50-130 GFLOP: their roofline is the HBM, and the tiled kernels ran them at 2.3-2.9 TB/s -- a
// 256 x 128 tile with a K of 128 is ~1 us of MFMA work behind ~10 us of row decode, first DMA round trip and LDS epilogue
// (tools/gpu/v5_tile_cost.sh).  Here nothing is tiled over pixels (the structure of conv_small.hip's stem kernel):
//   * the whole weight matrix [N][K] sits in LDS for the block's life (rows padded by 16 bytes; the 32 channels of a step are laid out so
//     that the 16 rows of one MFMA block are consecutive LDS rows);
//   * a wave walks 32-pixel groups with a grid stride; its pixel operands are loaded straight from the NHWC tensor (one lane = 8
//     consecutive channels of one pixel = the 8 k of a 16x16x32 step: 16 bytes, K/32 loads per 16 pixels), no LDS, no barrier;
//   * per 32 output channels: two MFMA row blocks whose rows are permuted (c = 32 s + 8 g + 4 u + e for accumulator e of block u in lane
//     group g) so that a lane owns 8 consecutive channels of its pixel: one 16-byte store, the four lane groups covering 64 contiguous
//     bytes; optional `dst +=` / addend view; BatchNorm sums of the raw output stay in registers (N <= 128).
#include <stdlib.h>
#include "dy_common.h"
#include "../../include/dedark_yolo.h"

namespace px {

constexpr int NT = 512;
// weight rows in LDS are padded by TWO 16-byte slots: lane (col, g) of a ds_read_b128 then reads slot 2 col + g (K = 128, 256) or
// 10 col + g (K = 64, 320) of the 16-slot bank row, and each of the instruction's 16-lane groups ({cols 0-3, 12-15} of k-chunk g with
// {cols 4-11} of chunk g ^ 1) covers all 16 slots; with one slot of padding (slot col + g) every group had one pair on the same slot
// (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50 on every px instance, profiles/r03_lds_conflicts.txt)
constexpr int ROW_PAD = 32;

struct P {
  const char* x;
  long x_ld;
  const char* w;                 // [N][K] (forward pack of a 1x1 conv, or the transposed pack for its data gradient)
  char* y;
  long y_ld;
  long M;
  int K, N;
  int chunks;                    // > 1: the output channels are cut into chunks of 32 * NS; block b works on chunk b % chunks with its own
                                 // weight rows in LDS and re-reads the pixels (wide-N data gradients: the re-reads are L2 / Infinity Cache hits)
  int accumulate;
  const char* add;
  long add_ld;
  double* stats;
};

// LDS row of channel c: step (c >> 5) * 32 + 16 * u + 4 * g + e  with  c & 31 = 8 g + 4 u + e
__device__ inline int lds_row(int c) { return (c & ~31) + 16 * ((c >> 2) & 1) + 4 * ((c >> 3) & 3) + (c & 3); }

template <int KB, int NS, typename T, bool STATS>
__global__ __launch_bounds__(NT) void px1x1_kernel(const P p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int K = 32 * KB, PITCH = 2 * K + px::ROW_PAD, UNR = NS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, g = lane >> 4;
  const T* x = reinterpret_cast<const T*>(p.x);
  T* y = reinterpret_cast<T*>(p.y);
  const T* add = reinterpret_cast<const T*>(p.add);
  const int chunk = p.chunks > 1 ? (int)(blockIdx.x % p.chunks) : 0;
  const int n0 = chunk * 32 * NS;                          // first output channel of this block
  const int ns_here = (p.N - n0) / 32 < NS ? (p.N - n0) / 32 : NS;
  const long blk = p.chunks > 1 ? blockIdx.x / p.chunks : blockIdx.x, nblk = p.chunks > 1 ? gridDim.x / p.chunks : gridDim.x;
  for (int i = tid; i < 32 * ns_here * KB * 4; i += NT) {  // weights -> LDS, 16-byte chunks
    const int c = i / (KB * 4), ch = i - c * (KB * 4);
    *reinterpret_cast<u32x4*>(smem + lds_row(c) * PITCH + ch * 16) = *reinterpret_cast<const u32x4*>(p.w + ((long)(n0 + c) * K + ch * 8) * 2);
  }
  __syncthreads();
  const int a_base = col * PITCH + 16 * g;                 // + (32 s + 16 u) * PITCH + 64 kb
  float s1[STATS ? NS : 1][8], s2[STATS ? NS : 1][8];
  if (STATS) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) s1[s][e] = s2[s][e] = 0.f;
  }
  const long groups = (p.M + 31) / 32;
  for (long grp = blk * (NT / 64) + wave; grp < groups; grp += nblk * (NT / 64)) {
    u32x4 xf[2][KB];
    long m[2];
    bool live[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      m[q] = grp * 32 + 16 * q + col;
      live[q] = m[q] < p.M;
      const T* src = x + (live[q] ? m[q] : 0) * p.x_ld + 8 * g;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) xf[q][kb] = live[q] ? *reinterpret_cast<const u32x4*>(src + 32 * kb) : u32x4{0u, 0u, 0u, 0u};
    }
    // (one 32-channel step at a time: fully unrolled and freely scheduled, the compiler hoists every step's weight fragments to the top
    //  of the group and spills; the statistics variant indexes its register sums by s and stays unrolled behind scheduling barriers)
#pragma unroll UNR
    for (int s = 0; s < NS; ++s) {
      __builtin_amdgcn_sched_barrier(0);
      if (s >= ns_here) break;                             // (block-uniform: the last chunk of a wide layer may be shorter)
      f32x4 acc[2][2];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[q][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          const u32x4 a = *reinterpret_cast<const u32x4*>(smem + a_base + (32 * s + 16 * u) * PITCH + 64 * kb);
#pragma unroll
          for (int q = 0; q < 2; ++q) acc[q][u] = mfma_16x16x32<T>(a, xf[q][kb], acc[q][u]);
        }
      // lane (pixel col, group g) holds channels 32 s + 8 g + 4 u + e
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[q][e >> 2][e & 3];
        if (STATS) {                                       // (pixels beyond M were fed zeros: exactly 0, no predicate)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s1[s][e] += v[e];
            s2[s][e] += v[e] * v[e];
          }
        }
        if (live[q]) {
          T* o = y + m[q] * p.y_ld + n0 + 32 * s + 8 * g;
          if (p.accumulate || add) {
            float t[8];
            if (p.accumulate) {
              ldvec<T>(o, t);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += t[e];
            }
            if (add) {
              ldvec<T>(add + m[q] * p.add_ld + n0 + 32 * s + 8 * g, t);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += t[e];
            }
          }
          stvec<T>(o, v);
        }
      }
    }
  }
  if (STATS) {
    __syncthreads();                                       // every wave is done with the weight image: reuse it
    float* red = reinterpret_cast<float*>(smem);           // [wave][channel][2]
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = row16_sum(s1[s][e]), b = row16_sum(s2[s][e]);
        if (col == 0) {
          red[(wave * 32 * NS + 32 * s + 8 * g + e) * 2] = a;
          red[(wave * 32 * NS + 32 * s + 8 * g + e) * 2 + 1] = b;
        }
      }
    __syncthreads();
    if (tid < 32 * NS) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < NT / 64; ++w8) {
        a += red[(w8 * 32 * NS + tid) * 2];
        b += red[(w8 * 32 * NS + tid) * 2 + 1];
      }
      double* st = p.stats + (long)(blockIdx.x % DY_STATS_REPLICAS) * 2 * p.N;
      atomic_add_f64(st + tid, (double)a);
      atomic_add_f64(st + p.N + tid, (double)b);
    }
  }
}

}  // namespace px

// Raw-output 1x1 / stride-1 layers on an unchanged pixel grid, K in {64, 128, 320}, N a multiple of 32, the weight image within LDS;
// BatchNorm sums only up to 128 output channels (they live in registers); long pixel ranges (the layers of the 160x160 / 320x320 stages);
// K = 256 with up to 1024 output channels in chunks of 256 (the wide 1x1 data gradients of the 80x80 stage; no sums).
bool dy_conv_px_eligible(const dy_conv_desc* d) {
  static const bool off = dy_env("DY_NO_CONV_PX") != nullptr;
  if (off || (d->dtype != DY_BF16 && d->dtype != DY_F16)) return false;
  if (!(d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->KHf == 0 && d->dst_row_stride == 0 && d->dst && !d->dst_planar &&
        d->Hs == d->Hd && d->Ws == d->Wd && !d->scale && !d->shift && d->act == DY_ACT_NONE))
    return false;
  if (d->Cd % 32 != 0 || d->Cd < 64) return false;
  if (d->Cs == 256) {                                  // wide data gradients of the 80x80 stage: output channels in chunks of 256
    if (d->stats || d->Cd > 1024) return false;
  } else {
    if (!(d->Cs == 64 || d->Cs == 128 || d->Cs == 320) || d->Cd > 320) return false;
    if ((long)d->Cd * (2 * d->Cs + px::ROW_PAD) > 150 * 1024) return false;
  }
  if (d->stats && (d->Cd > 128 || d->accumulate || d->add_src)) return false;
  if ((d->src_ld * 2) % 16 != 0 || (d->dst_ld * 2) % 16 != 0 || ((uintptr_t)d->dst) % 16 != 0 || ((uintptr_t)d->src) % 16 != 0) return false;
  if (d->add_src && ((d->add_src_ld * 2) % 16 != 0 || ((uintptr_t)d->add_src) % 16 != 0)) return false;
  return (long)d->N * d->Hd * d->Wd >= 262144;
}

// (the statistics variant keeps 2 x 8 x NS register sums: it exists for NS <= 4 only -- dy_conv_px_eligible admits BatchNorm sums up to 128
//  output channels -- and the wider instantiations, 125-227 spilled registers each, are not built)
template <int KB, int NS, typename T>
static int px_go(const px::P& p, bool stats, size_t shm, unsigned blocks, hipStream_t st) {
  constexpr bool HAS_STATS = NS <= 4;
  static bool configured_s = false, configured_p = false;
  bool& configured = stats ? configured_s : configured_p;
  const void* fn = reinterpret_cast<const void*>(&px::px1x1_kernel<KB, NS, T, false>);
  if constexpr (HAS_STATS) {
    if (stats) fn = reinterpret_cast<const void*>(&px::px1x1_kernel<KB, NS, T, true>);
  } else if (stats) {
    dy_set_error("conv_px: no statistics variant for %d output channels per block", 32 * NS);
    return 3;
  }
  if (!configured) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
      dy_set_error("conv_px: hipFuncSetAttribute failed");
      return 3;
    }
    configured = true;
  }
  if constexpr (HAS_STATS) {
    if (stats) {
      px::px1x1_kernel<KB, NS, T, true><<<blocks, px::NT, shm, st>>>(p);
      return 0;
    }
  }
  px::px1x1_kernel<KB, NS, T, false><<<blocks, px::NT, shm, st>>>(p);
  return 0;
}

int dy_conv_px_launch(const dy_conv_desc* d, void* stream) {
  px::P p;
  p.x = (const char*)d->src; p.x_ld = d->src_ld; p.w = (const char*)d->w; p.y = (char*)d->dst; p.y_ld = d->dst_ld;
  p.M = (long)d->N * d->Hd * d->Wd; p.K = d->Cs; p.N = d->Cd; p.accumulate = d->accumulate;
  p.add = (const char*)d->add_src; p.add_ld = d->add_src_ld; p.stats = d->stats;
  const int chunk_n = d->Cs == 256 ? 256 : d->Cd;          // output channels per block
  p.chunks = (d->Cd + chunk_n - 1) / chunk_n;
  const int rows = d->Cd < chunk_n ? d->Cd : chunk_n;
  const size_t w_bytes = (size_t)rows * (2 * d->Cs + px::ROW_PAD), red_bytes = (size_t)(px::NT / 64) * d->Cd * 8;
  const size_t shm = w_bytes > red_bytes ? w_bytes : red_bytes;
  const long groups = (p.M + 31) / 32;
  long blocks = 256;                                       // one block (8 waves) per CU
  if (blocks * (px::NT / 64) > groups) blocks = (groups + px::NT / 64 - 1) / (px::NT / 64);
  if (p.chunks > 1) blocks = (256 / p.chunks) * p.chunks;  // a whole number of blocks per chunk
  hipStream_t st = (hipStream_t)stream;
  const bool stats = d->stats != nullptr;
  const bool f16 = d->dtype == DY_F16;
  dy_note_kernel("px1x1_kernel");
  int rc = 4;
#define PX(KB_, NS_) rc = f16 ? px_go<KB_, NS_, f16_t>(p, stats, shm, (unsigned)blocks, st) : px_go<KB_, NS_, bf16_t>(p, stats, shm, (unsigned)blocks, st)
  const int kb = d->Cs / 32, ns = d->Cd / 32;
  if (kb == 2 && ns == 2) PX(2, 2);
  else if (kb == 4 && ns == 2) PX(4, 2);
  else if (kb == 4 && ns == 4) PX(4, 4);
  else if (kb == 10 && ns == 4) PX(10, 4);
  else if (kb == 4 && ns == 10) PX(4, 10);
  else if (kb == 2 && ns == 4) PX(2, 4);
  else if (kb == 8) PX(8, 8);                              // chunks of 256 channels (the last one may be shorter)
#undef PX
  DY_CHECK(rc != 4, "conv_px: no instantiation for K=%d N=%d", d->Cs, d->Cd);
  if (rc) return rc;
  DY_LAUNCH_CHECK();
  return 0;
}

bool dy_conv_px_has_shape(const dy_conv_desc* d) {
  const int kb = d->Cs / 32, ns = d->Cd / 32;
  return (kb == 2 && ns == 2) || (kb == 4 && ns == 2) || (kb == 4 && ns == 4) || (kb == 10 && ns == 4) || (kb == 4 && ns == 10) || (kb == 2 && ns == 4) ||
         kb == 8;
}
