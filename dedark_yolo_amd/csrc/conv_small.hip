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
  if (d->dtype == DY_F16) {
    if (d->Cs == 16) return launch_small_dgrad<16, f16_t>(d, st);
    if (d->Cs == 32) return launch_small_dgrad<32, f16_t>(d, st);
    return launch_small_dgrad<64, f16_t>(d, st);
  }
  if (d->Cs == 16) return launch_small_dgrad<16, bf16_t>(d, st);
  if (d->Cs == 32) return launch_small_dgrad<32, bf16_t>(d, st);
  return launch_small_dgrad<64, bf16_t>(d, st);
}
