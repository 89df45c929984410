// Direct (non-MFMA) kernels for the network stem, where the implicit-GEMM tiles are mostly padding.
//
// dgrad of a 3x3 / stride 2 / pad 1 convolution into a <= 8-channel image (Conv(3, c, 3, 2) of yolov8*.yaml layer 1, the first
// ConvBlock of the parameter extractor, reference ultralytics/nn/modules/conv.py:38-55 / common.py:9-23):
//   dx[n, h, w, ci] = sum_{kh, kw, co} dz[n, (h+1-kh)/2, (w+1-kw)/2, co] * w[co, ci, kh, kw]      (terms with odd h+1-kh dropped)
// One thread owns a 2x2 quad of output pixels: the quad touches exactly the 2x2 block of dz pixels (qh..qh+1, qw..qw+1) and
// every one of the 9 taps once.  dz pixels stay packed (bf16 pairs) in VGPRs, the weights are wave-uniform and come through
// the scalar cache straight into the SGPR operand of v_dot2c_f32_bf16 (2 MACs per lane per instruction, f32 accumulate).
// The kernel is HBM-bound: it reads dz once and writes dx once.  The MFMA formulation (128x32 tiles for 8 output channels,
// K = 16..64 per parity class) took 1.08 ms on the 32x640x640 stem; this kernel is bounded by ~315 MB of traffic.
#include "dy_common.h"
#include "../../include/dedark_yolo.h"

namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

// two 16-bit MACs per lane and instruction, f32 accumulate: v_dot2c_f32_bf16 / v_dot2_f32_f16 (the reference's AMP dtype, BASELINE configs[4])
template <typename T> __device__ inline float dot2(uint32_t a, uint32_t b, float c);
template <> __device__ inline float dot2<bf16_t>(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
}
template <> __device__ inline float dot2<f16_t>(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a), __builtin_bit_cast(f16x2, b), c, false);
}

template <int CS, int NCI, typename T>
__global__ __launch_bounds__(256) void dgrad3x3s2_small_kernel(const T* __restrict__ dz, long dz_ld,
                                                                const uint32_t* __restrict__ wt, T* __restrict__ dx,
                                                                long dx_ld, int N, int Hd, int Wd, int Hs, int Ws,
                                                                int accumulate, T* __restrict__ planar, int nplanes) {
  constexpr int NP = CS / 2;                         // packed pairs per pixel
  const int QH = (Hd + 1) >> 1, QW = (Wd + 1) >> 1;
  const long q = blockIdx.x * 256L + threadIdx.x;
  if (q >= (long)N * QH * QW) return;
  const int qw = (int)(q % QW);
  const long t = q / QW;
  const int qh = (int)(t % QH), n = (int)(t / QH);

  uint32_t P[2][2][NP];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int r = qh + a, c = qw + b;
      const bool ok = r < Hs && c < Ws;
      const u32x4* src = reinterpret_cast<const u32x4*>(dz + (((long)n * Hs + (ok ? r : 0)) * Ws + (ok ? c : 0)) * dz_ld);
#pragma unroll
      for (int v = 0; v < NP / 4; ++v) {
        u32x4 x = {0u, 0u, 0u, 0u};
        if (ok) x = src[v];
        P[a][b][4 * v + 0] = x[0]; P[a][b][4 * v + 1] = x[1]; P[a][b][4 * v + 2] = x[2]; P[a][b][4 * v + 3] = x[3];
      }
    }

  float acc[2][2][NCI];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ci = 0; ci < NCI; ++ci) acc[i][j][ci] = 0.f;

  // output row parity dh: even rows see tap kh = 1 (source row qh), odd rows taps kh = 0 (source qh+1) and kh = 2 (source qh)
#pragma unroll
  for (int ci = 0; ci < NCI; ++ci)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int dh = kh == 1 ? 0 : 1, a = kh == 0 ? 1 : 0;
        const int dw = kw == 1 ? 0 : 1, b = kw == 0 ? 1 : 0;
        const uint32_t* w = wt + ((ci * 3 + kh) * 3 + kw) * NP;       // wave-uniform -> s_load
        float s = acc[dh][dw][ci];
#pragma unroll
        for (int p = 0; p < NP; ++p) s = dot2<T>(P[a][b][p], w[p], s);
        acc[dh][dw][ci] = s;
      }

  if (planar) {
    // planar [N, nplanes, Hd, Wd] output (the front-end's layout): 6 B per pixel instead of the 16 B NHWC8 vector
#pragma unroll
    for (int dh = 0; dh < 2; ++dh) {
      const int h = 2 * qh + dh;
      if (h >= Hd) continue;
#pragma unroll
      for (int ci = 0; ci < NCI; ++ci) {
        if (ci >= nplanes) break;
        T* o = planar + (((long)n * nplanes + ci) * Hd + h) * Wd + 2 * qw;
#pragma unroll
        for (int dw = 0; dw < 2; ++dw)
          if (2 * qw + dw < Wd) {
            float v = acc[dh][dw][ci];
            if (accumulate) v += DT<T>::ld(o + dw);
            DT<T>::st(o + dw, v);
          }
      }
    }
    return;
  }
#pragma unroll
  for (int dh = 0; dh < 2; ++dh)
#pragma unroll
    for (int dw = 0; dw < 2; ++dw) {
      const int h = 2 * qh + dh, w = 2 * qw + dw;
      if (h < Hd && w < Wd) {
        T* o = dx + (((long)n * Hd + h) * Wd + w) * dx_ld;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        if (accumulate) ldvec<T>(o, v);
#pragma unroll
        for (int ci = 0; ci < NCI; ++ci) v[ci] += acc[dh][dw][ci];
        stvec<T>(o, v);
      }
    }
}

// dgrad of a 1x1 convolution with 8 (padded) output channels -- the ASFF weight_level convs Conv(c, 8, 1) of
// ultralytics/nn/modules/block.py:47-52:  dx[m][ci] (+)= sum_{co<8} dz[m][co] * w[co][ci].  Pure write bandwidth; the MFMA
// tile kernel spent 525 us on 64x80x80x256 (K = 8 of a 64-wide step, 2-byte stores), this one moves 16 B per lane.
constexpr int THIN_PX = 4;
template <typename T>
__global__ __launch_bounds__(256) void dgrad1x1_thin_kernel(const T* __restrict__ dz, long dz_ld, const T* __restrict__ wt,
                                                             T* __restrict__ dx, long dx_ld, long M, int Cd, int accumulate) {
  const int CG = Cd >> 3;
  const long t = blockIdx.x * 256L + threadIdx.x;
  const int cg = (int)(t % CG);
  const long m0 = (t / CG) * THIN_PX;
  if (m0 >= M) return;
  u32x4 w[8];                                        // rows ci = 8*cg + e of the [Cd][8] transposed pack
#pragma unroll
  for (int e = 0; e < 8; ++e) w[e] = *reinterpret_cast<const u32x4*>(wt + ((long)cg * 8 + e) * 8);
#pragma unroll
  for (int px = 0; px < THIN_PX; ++px) {
    const long m = m0 + px;
    if (m >= M) break;
    const u32x4 z = *reinterpret_cast<const u32x4*>(dz + m * dz_ld);
    T* o = dx + m * dx_ld + cg * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (accumulate) ldvec<T>(o, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s = v[e];
#pragma unroll
      for (int q = 0; q < 4; ++q) s = dot2<T>(z[q], w[e][q], s);
      v[e] = s;
    }
    stvec<T>(o, v);
  }
}

// The same data gradient on the matrix cores, for 64 gradient channels (the L / C5 stem: 64 x 320 x 320 x 64 of dz): the dot2 kernel above
// needs 864 v_dot2 per thread and was VALU-bound at 2.1 TB/s.  Per 16 consecutive quads of one quad row a wave loads the 2 x 2 x 17 block
// of dz pixels they touch as MFMA column operands (lane = quad j and 8 channels of one pixel: 16 bytes straight from memory), keeps the
// nine weight taps [ci][tap][64] as row operands in registers (rows = input channels: 3 of 16 used -- the matrix pipe has 20x the headroom
// this layer needs) and issues 18 v_mfma_f32_16x16x32: parity class (dh, dw) accumulates its 1 / 2 / 2 / 4 taps.  A lane of group 0 ends
// up with channels 0..3 of its quad's four pixels; the two pixels of a row go out as one packed store per plane (64 contiguous bytes per
// 16 lanes) or as 8-byte NHWC stores.
template <typename T>
__global__ __launch_bounds__(256) void stem_dgrad64_kernel(const T* __restrict__ dz, long dz_ld, const T* __restrict__ wt, T* __restrict__ dx,
                                                           long dx_ld, int N, int Hd, int Wd, int Hs, int Ws, int accumulate,
                                                           T* __restrict__ planar, int nplanes) {
  constexpr int CS = 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int QH = (Hd + 1) >> 1, QW = (Wd + 1) >> 1, GW = (QW + 15) >> 4;
  // weight fragments [tap][k step]: lane (row ci = j, k group g) = 8 consecutive co of tap t: wt[ci][t][32*kb + 8*g ..]
  u32x4 wf[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
      wf[t][kb] = j < 8 ? *reinterpret_cast<const u32x4*>(wt + ((long)j * 9 + t) * CS + 32 * kb + 8 * g) : u32x4{0u, 0u, 0u, 0u};
  const long groups = (long)N * QH * GW;
  for (long grp = (long)blockIdx.x * 4 + wave; grp < groups; grp += (long)gridDim.x * 4) {
    const int gw = (int)(grp % GW);
    const long t_ = grp / GW;
    const int qh = (int)(t_ % QH), n = (int)(t_ / QH);
    const int qw = gw * 16 + j;
    u32x4 xf[2][2][2];                                   // [a][b][k step]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int r = qh + a, c = qw + b;
        const bool ok = r < Hs && c < Ws;
        const T* src = dz + (((long)n * Hs + (ok ? r : 0)) * Ws + (ok ? c : 0)) * dz_ld + 8 * g;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) xf[a][b][kb] = ok ? *reinterpret_cast<const u32x4*>(src + 32 * kb) : u32x4{0u, 0u, 0u, 0u};
      }
    f32x4 acc[2][2];
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) acc[dh][dw] = f32x4{0.f, 0.f, 0.f, 0.f};
    // output row parity dh: even rows see tap kh = 1 (source row qh), odd rows taps kh = 0 (source qh+1) and kh = 2 (source qh)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int dh = kh == 1 ? 0 : 1, a = kh == 0 ? 1 : 0;
        const int dw = kw == 1 ? 0 : 1, b = kw == 0 ? 1 : 0;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) acc[dh][dw] = mfma_16x16x32<T>(wf[kh * 3 + kw][kb], xf[a][b][kb], acc[dh][dw]);
      }
    // lane (quad j, group g) holds channels 4g .. 4g+3 of the quad's four pixels
    if (planar) {
      if (g < 2) {
#pragma unroll
        for (int dh = 0; dh < 2; ++dh) {
          const int h = 2 * qh + dh;
          if (h >= Hd || 2 * qw >= Wd) continue;
#pragma unroll
          for (int ci = 0; ci < 4; ++ci) {
            if (4 * g + ci >= nplanes) break;
            T* o = planar + (((long)n * nplanes + 4 * g + ci) * Hd + h) * Wd + 2 * qw;
            float v0 = acc[dh][0][ci], v1 = acc[dh][1][ci];
            const bool two = 2 * qw + 1 < Wd;
            if (accumulate) {
              v0 += DT<T>::ld(o);
              if (two) v1 += DT<T>::ld(o + 1);
            }
            if (two && (Wd & 1) == 0) {                 // even row length: the pair is 4-byte aligned
              if constexpr (__is_same(T, f16_t)) {
                typedef __attribute__((ext_vector_type(2))) _Float16 h2;
                *reinterpret_cast<h2*>(o) = h2{(_Float16)v0, (_Float16)v1};
              } else {
                *reinterpret_cast<uint32_t*>(o) = pack_bf16x2(v0, v1);
              }
            } else {
              DT<T>::st(o, v0);
              if (two) DT<T>::st(o + 1, v1);
            }
          }
        }
      }
      continue;
    }
    if (g < 2) {
#pragma unroll
      for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int dw = 0; dw < 2; ++dw) {
          const int h = 2 * qh + dh, w = 2 * qw + dw;
          if (h < Hd && w < Wd) {
            T* o = dx + (((long)n * Hd + h) * Wd + w) * dx_ld + 4 * g;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[dh][dw][e] + (accumulate ? DT<T>::ld(o + e) : 0.f);
            if constexpr (__is_same(T, f16_t)) {
              typedef __attribute__((ext_vector_type(4))) _Float16 h4;
              *reinterpret_cast<h4*>(o) = h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            } else {
              *reinterpret_cast<uint2*>(o) = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            }
          }
        }
    }
  }
}

// Forward of the network stem, Conv(3, c, 3, 2) on the zero-padded NHWC8 image (yolov8*.yaml layer 1: 3 -> 64 at 640x640; reference
// ultralytics/nn/modules/conv.py:38-55), raw output + BatchNorm sums (training) -- 2.6 GFLOP over 1.26 GB: the layer is HBM-bound, and
// the tiled kernels, whose 256 x 64 tile has a K of 72 behind a full prologue / LDS epilogue, ran it at 2.3 TB/s.
//   out^T[c][px] = sum_k W[c][k] * X[k][px],  k = tap * 8 + ci  (K = 72, padded to 3 MFMA steps of 32 = 4 taps x 8 channels each)
// One v_mfma_f32_16x16x32 step takes, per lane, 8 consecutive k of one row / column -- exactly ONE NHWC8 pixel (16 bytes) of one tap:
// the pixel operand is loaded straight from the image (per-lane address, zeros for padding / taps 9..11), the weight operand straight
// from the forward pack [c][tap][8] (its row IS the k order); no LDS, no barrier.  Rows of a 16-channel block are permuted so that lane
// group g = lane >> 4 owns channels 4*CB*g .. +4*CB-1 across the CB row blocks: its accumulators are consecutive channels of one
// pixel and go out as 16-byte stores, the four groups covering the pixel's whole row.  A wave walks 16-pixel groups with a grid
// stride and keeps the per-channel sums in registers.
template <int CB, typename T>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const T* __restrict__ x, long x_ld, const T* __restrict__ w, T* __restrict__ y,
                                                       long y_ld, int N, int Hs, int Ws, int Hd, int Wd, int Cd, double* stats) {
  constexpr int CPG = 4 * CB;                       // channels per lane group
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, g = lane >> 4;
  // weight fragments [row block][k step]: lane (row r = col, k group g) = 8 consecutive k of channel c(rb, r) = CPG*(r>>2) + 4*rb + (r&3)
  u32x4 wf[CB][3];
#pragma unroll
  for (int rb = 0; rb < CB; ++rb) {
    const int c = CPG * (col >> 2) + 4 * rb + (col & 3);
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) {
      const int tap = 4 * kb + g;
      wf[rb][kb] = (tap < 9 && c < Cd) ? *reinterpret_cast<const u32x4*>(w + ((long)c * 9 + tap) * 8) : u32x4{0u, 0u, 0u, 0u};
    }
  }
  float s1[CB][4], s2[CB][4];
#pragma unroll
  for (int rb = 0; rb < CB; ++rb)
#pragma unroll
    for (int e = 0; e < 4; ++e) s1[rb][e] = s2[rb][e] = 0.f;
  const long M = (long)N * Hd * Wd, groups = (M + 15) / 16;
  const long gstride = (long)gridDim.x * 4;
  // this lane's tap of each k step: (dh, dw) relative to the window's top-left source pixel
  int tdh[3], tdw[3];
#pragma unroll
  for (int kb = 0; kb < 3; ++kb) {
    const int tap = 4 * kb + g;
    tdh[kb] = tap / 3;
    tdw[kb] = tap - 3 * tdh[kb];
  }
  for (long grp = (long)blockIdx.x * 4 + wave; grp < groups; grp += gstride) {
    const long m = grp * 16 + col;
    const bool live = m < M;
    const unsigned mu = (unsigned)(live ? m : 0);
    const unsigned hw = (unsigned)(Hd * Wd);
    const int n = (int)(mu / hw);
    const unsigned rem = mu - (unsigned)n * hw;
    const int oh = (int)(rem / (unsigned)Wd), ow = (int)(rem - (unsigned)oh * (unsigned)Wd);
    const int h0 = 2 * oh - 1, w0 = 2 * ow - 1;
    u32x4 xf[3];
#pragma unroll
    for (int kb = 0; kb < 3; ++kb) {
      const int sh = h0 + tdh[kb], sw = w0 + tdw[kb];
      const bool ok = live && 4 * kb + g < 9 && (unsigned)sh < (unsigned)Hs && (unsigned)sw < (unsigned)Ws;
      xf[kb] = ok ? *reinterpret_cast<const u32x4*>(x + (((long)n * Hs + sh) * Ws + sw) * x_ld) : u32x4{0u, 0u, 0u, 0u};
    }
    f32x4 acc[CB];
#pragma unroll
    for (int rb = 0; rb < CB; ++rb) {
      acc[rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 3; ++kb) acc[rb] = mfma_16x16x32<T>(wf[rb][kb], xf[kb], acc[rb]);
    }
    // lane (pixel col, group g) holds channels CPG*g + 4*rb + e of its pixel
    if (live) {
      T* o = y + m * y_ld + CPG * g;
      float v[8];
#pragma unroll
      for (int half = 0; half < (CPG + 7) / 8; ++half) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int ch = 8 * half + e;
          v[e] = ch < CPG ? acc[ch >> 2][ch & 3] : 0.f;
        }
        if (CPG >= 8) {
          if (CPG * g + 8 * half < Cd) stvec<T>(o + 8 * half, v);
        } else if (CPG * g < Cd) {                    // CB == 1: four channels = 8 bytes
          if constexpr (__is_same(T, f16_t)) {
            typedef __attribute__((ext_vector_type(4))) _Float16 h4;
            *reinterpret_cast<h4*>(o) = h4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
          } else {
            *reinterpret_cast<uint2*>(o) = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          }
        }
      }
    }
    if (stats) {                                      // (pixels beyond M were fed zeros: exactly 0, no predicate)
#pragma unroll
      for (int rb = 0; rb < CB; ++rb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s1[rb][e] += acc[rb][e];
          s2[rb][e] += acc[rb][e] * acc[rb][e];
        }
    }
  }
  if (stats) {
    __shared__ float red[4][16 * CB][2];              // [wave][channel][sum, sum of squares]
#pragma unroll
    for (int rb = 0; rb < CB; ++rb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = row16_sum(s1[rb][e]), b = row16_sum(s2[rb][e]);
        if (col == 0) {
          red[wave][CPG * g + 4 * rb + e][0] = a;
          red[wave][CPG * g + 4 * rb + e][1] = b;
        }
      }
    __syncthreads();
    if ((int)threadIdx.x < 16 * CB && (int)threadIdx.x < Cd) {
      const int c = threadIdx.x;
      const float a = (red[0][c][0] + red[1][c][0]) + (red[2][c][0] + red[3][c][0]);
      const float b = (red[0][c][1] + red[1][c][1]) + (red[2][c][1] + red[3][c][1]);
      double* st = stats + (long)(blockIdx.x % DY_STATS_REPLICAS) * 2 * Cd;
      atomic_add_f64(st + c, (double)a);
      atomic_add_f64(st + Cd + c, (double)b);
    }
  }
}

template <int CS, typename T>
int launch_small_dgrad(const dy_conv_desc* d, hipStream_t st) {
  const long quads = (long)d->N * ((d->Hd + 1) / 2) * ((d->Wd + 1) / 2);
  const unsigned grid = (unsigned)((quads + 255) / 256);
  const int nci = d->dst_valid_channels > 0 && d->dst_valid_channels <= 4 ? 4 : 8;
  dy_note_kernel("dgrad3x3s2_small_kernel");
#define GO(NCI)                                                                                                              \
  dgrad3x3s2_small_kernel<CS, NCI, T><<<grid, 256, 0, st>>>((const T*)d->src, d->src_ld, (const uint32_t*)d->w, (T*)d->dst,       \
                                                            d->dst_ld, d->N, d->Hd, d->Wd, d->Hs, d->Ws, d->accumulate,         \
                                                            (T*)d->dst_planar, d->dst_valid_channels)
  if (nci == 4) GO(4); else GO(8);
#undef GO
  DY_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// Taken for the raw-output forward (training: BatchNorm follows; also plain inference convs without affine / activation)
bool dy_conv_stem_fwd_eligible(const dy_conv_desc* d) {
  static const bool off = dy_env("DY_NO_CONV_SMALL") != nullptr;
  return !off && (d->dtype == DY_BF16 || d->dtype == DY_F16) && d->KH == 3 && d->KW == 3 && d->stride == 2 && d->pad == 1 && d->dil == 1 &&
         d->Cs == 8 && d->KHf == 0 && d->dst_row_stride == 0 && d->dst && (d->Cd == 16 || d->Cd == 32 || d->Cd == 64) && !d->scale &&
         !d->shift && d->act == DY_ACT_NONE && !d->accumulate && (d->src_ld * 2) % 16 == 0 && (d->dst_ld * 2) % 16 == 0 &&
         ((uintptr_t)d->dst) % 16 == 0 && (long)d->N * d->Hd * d->Wd < (1L << 31);
}

int dy_conv_stem_fwd_launch(const dy_conv_desc* d, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const long groups = ((long)d->N * d->Hd * d->Wd + 15) / 16;
  long blocks = (groups + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;            // 8 blocks (32 waves) per CU, each wave walks its pixel groups with a grid stride
  dy_note_kernel("stem_fwd_kernel");
#define GO(CB_, T_)                                                                                                              \
  stem_fwd_kernel<CB_, T_><<<(unsigned)blocks, 256, 0, st>>>((const T_*)d->src, d->src_ld, (const T_*)d->w, (T_*)d->dst, d->dst_ld, d->N, \
                                                            d->Hs, d->Ws, d->Hd, d->Wd, d->Cd, d->stats)
  if (d->dtype == DY_F16) {
    if (d->Cd == 16) GO(1, f16_t); else if (d->Cd == 32) GO(2, f16_t); else GO(4, f16_t);
  } else {
    if (d->Cd == 16) GO(1, bf16_t); else if (d->Cd == 32) GO(2, bf16_t); else GO(4, bf16_t);
  }
#undef GO
  DY_LAUNCH_CHECK();
  return 0;
}

static bool thin_eligible(const dy_conv_desc* d) {
  return (d->dtype == DY_BF16 || d->dtype == DY_F16) && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && d->Cs == 8 && d->Cd % 8 == 0 &&
         d->KHf == 0 && d->dst_row_stride == 0 && d->dst && !d->dst_planar && (d->src_ld * 2) % 16 == 0 && (d->dst_ld * 2) % 16 == 0 &&
         d->Hs == d->Hd && d->Ws == d->Wd;
}

bool dy_conv_small_dgrad_eligible(const dy_conv_desc* d) {
  static const bool off = dy_env("DY_NO_CONV_SMALL") != nullptr;
  if (off) return false;
  if (thin_eligible(d)) return true;
  return (d->dtype == DY_BF16 || d->dtype == DY_F16) && d->KH == 3 && d->KW == 3 && d->stride == 2 && d->pad == 1 && d->dil == 1 && d->Cd == 8 &&
         (d->Cs == 16 || d->Cs == 32 || d->Cs == 64) && d->KHf == 0 && d->dst_row_stride == 0 && (d->src_ld * 2) % 16 == 0 &&
         ((d->dst_planar && d->dst_valid_channels > 0 && d->dst_valid_channels <= 8) || (d->dst && (d->dst_ld * 2) % 16 == 0));
}

int dy_conv_small_dgrad_launch(const dy_conv_desc* d, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (thin_eligible(d)) {
    const long M = (long)d->N * d->Hd * d->Wd;
    const long threads = (M + THIN_PX - 1) / THIN_PX * (d->Cd / 8);
    dy_note_kernel("dgrad1x1_thin_kernel");
    const unsigned grid = (unsigned)((threads + 255) / 256);
    if (d->dtype == DY_F16)
      dgrad1x1_thin_kernel<f16_t><<<grid, 256, 0, st>>>((const f16_t*)d->src, d->src_ld, (const f16_t*)d->w, (f16_t*)d->dst, d->dst_ld, M, d->Cd,
                                                        d->accumulate);
    else
      dgrad1x1_thin_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)d->src, d->src_ld, (const bf16_t*)d->w, (bf16_t*)d->dst, d->dst_ld, M,
                                                         d->Cd, d->accumulate);
    DY_LAUNCH_CHECK();
    return 0;
  }
  static const bool no_mfma = dy_env("DY_STEM_DGRAD_DOT2") != nullptr;
  if (d->Cs == 64 && !no_mfma) {                       // the L stem: matrix cores (the dot2 kernel is VALU-bound there)
    const long groups = (long)d->N * ((d->Hd + 1) / 2) * (((d->Wd + 1) / 2 + 15) / 16);
    long blocks = (groups + 3) / 4;
    if (blocks > 256 * 6) blocks = 256 * 6;
    dy_note_kernel("stem_dgrad64_kernel");
    if (d->dtype == DY_F16)
      stem_dgrad64_kernel<f16_t><<<(unsigned)blocks, 256, 0, st>>>((const f16_t*)d->src, d->src_ld, (const f16_t*)d->w, (f16_t*)d->dst, d->dst_ld,
                                                                   d->N, d->Hd, d->Wd, d->Hs, d->Ws, d->accumulate, (f16_t*)d->dst_planar,
                                                                   d->dst_valid_channels);
    else
      stem_dgrad64_kernel<bf16_t><<<(unsigned)blocks, 256, 0, st>>>((const bf16_t*)d->src, d->src_ld, (const bf16_t*)d->w, (bf16_t*)d->dst,
                                                                    d->dst_ld, d->N, d->Hd, d->Wd, d->Hs, d->Ws, d->accumulate,
                                                                    (bf16_t*)d->dst_planar, d->dst_valid_channels);
    DY_LAUNCH_CHECK();
    return 0;
  }
  if (d->dtype == DY_F16) {
    if (d->Cs == 16) return launch_small_dgrad<16, f16_t>(d, st);
    if (d->Cs == 32) return launch_small_dgrad<32, f16_t>(d, st);
    return launch_small_dgrad<64, f16_t>(d, st);
  }
  if (d->Cs == 16) return launch_small_dgrad<16, bf16_t>(d, st);
  if (d->Cs == 32) return launch_small_dgrad<32, bf16_t>(d, st);
  return launch_small_dgrad<64, bf16_t>(d, st);
}
