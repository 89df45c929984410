// Convolutions whose window covers the whole input (Hout = Wout = 1, pad 0): the fully connected layers of the parameter
// extractor, nn.Linear(2048, 64) on the flattened 32x8x8 map (reference ultralytics/nn/modules/common.py:65-78), expressed by
// the host as an 8x8 convolution.  As an implicit GEMM this is M = batch (32..64) rows: ONE 128-row tile that walks K = 2048 in
// 32..64 dependent steps on one CU (119 us forward / 176 us dgrad for 8 MFLOP).  Here every output element gets its own wave
// (forward) or thread (gradients); the three kernels are latency-trivial and together move < 2 MB.
#include "dy_common.h"
#include "../../include/dedark_yolo.h"

namespace {

// y[b][n] = act((sum_k x[b][k] * w[n][k]) * scale[n] + shift[n]);  one wave per (b, n)
template <typename T>
__global__ __launch_bounds__(256) void dense_fwd_kernel(const T* __restrict__ x, long x_img, const T* __restrict__ w, T* __restrict__ y,
                                                        long y_ld, int B, int K, int Cd, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int act) {
  constexpr int VE = DT<T>::VE;
  const int lane = threadIdx.x & 63;
  const long o = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (o >= (long)B * Cd) return;
  const int b = (int)(o / Cd), n = (int)(o - (long)b * Cd);
  const T* xr = x + b * x_img;
  const T* wr = w + (long)n * K;
  float s = 0.f;
  for (int k = lane * VE; k < K; k += 64 * VE) {
    float a[VE], c[VE];
    ldvec<T>(xr + k, a);
    ldvec<T>(wr + k, c);
#pragma unroll
    for (int e = 0; e < VE; ++e) s += a[e] * c[e];
  }
  s = wave_sum(s);
  if (lane == 0) {
    const float v = dy_act(act, s * (scale ? scale[n] : 1.f) + (shift ? shift[n] : 0.f));
    DT<T>::st(y + b * y_ld + n, v);
  }
}

// dx[b][h][w][ci] (+)= sum_co dz[b][co] * wt[ci][h][w][co];  one thread per output element, ci fastest
template <typename T>
__global__ __launch_bounds__(256) void dense_dgrad_kernel(const T* __restrict__ dz, long dz_ld, const T* __restrict__ wt,
                                                          T* __restrict__ dx, long dx_ld, int B, int HW, int Cd, int Cs,
                                                          int accumulate) {
  constexpr int VE = DT<T>::VE;
  const long t = blockIdx.x * 256L + threadIdx.x;
  if (t >= (long)B * HW * Cd) return;
  const int ci = (int)(t % Cd);
  const long r = t / Cd;
  const int hw = (int)(r % HW), b = (int)(r / HW);
  const T* zr = dz + b * dz_ld;
  const T* wr = wt + ((long)ci * HW + hw) * Cs;
  float s = 0.f;
  for (int co = 0; co < Cs; co += VE) {
    float a[VE], c[VE];
    ldvec<T>(zr + co, a);
    ldvec<T>(wr + co, c);
#pragma unroll
    for (int e = 0; e < VE; ++e) s += a[e] * c[e];
  }
  T* o = dx + ((long)b * HW + hw) * dx_ld + ci;
  if (accumulate) s += DT<T>::ld(o);
  DT<T>::st(o, s);
}

// g[co][ci][kh][kw] = sum_b dz[b][co] * x[b][kh][kw][ci];  one thread per (co, hw, ci), ci fastest (coalesced x reads)
template <typename T>
__global__ __launch_bounds__(256) void dense_wgrad_kernel(const T* __restrict__ x, long x_ld, const T* __restrict__ dz, long dz_ld,
                                                          float* __restrict__ g, int B, int HW, int Cin, int Cin_pad, int Cout) {
  const long t = blockIdx.x * 256L + threadIdx.x;
  if (t >= (long)Cout * HW * Cin_pad) return;
  const int ci = (int)(t % Cin_pad);
  const long r = t / Cin_pad;
  const int hw = (int)(r % HW), co = (int)(r / HW);
  if (ci >= Cin) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += DT<T>::ld(dz + b * dz_ld + co) * DT<T>::ld(x + ((long)b * HW + hw) * x_ld + ci);
  g[((long)co * Cin + ci) * HW + hw] = s;
}

}  // namespace

bool dy_dense_fwd_eligible(const dy_conv_desc* d) {
  return d->Hd == 1 && d->Wd == 1 && d->KH == d->Hs && d->KW == d->Ws && d->pad == 0 && d->dil == 1 && d->src_ld == d->Cs &&
         d->stats == nullptr && !d->accumulate && d->KHf == 0 && d->dst_row_stride == 0 && d->KH * d->KW > 1 && d->dst;
}

int dy_dense_fwd_launch(const dy_conv_desc* d, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int K = d->KH * d->KW * d->Cs;
  const long x_img = (long)d->Hs * d->Ws * d->src_ld;
  const unsigned grid = (unsigned)(((long)d->N * d->Cd + 3) / 4);
  if (d->dtype == DY_F32)
    dense_fwd_kernel<float><<<grid, 256, 0, st>>>((const float*)d->src, x_img, (const float*)d->w, (float*)d->dst, d->dst_ld, d->N, K, d->Cd,
                                                  d->scale, d->shift, d->act);
  else if ((d->dtype) == DY_F16)
    dense_fwd_kernel<f16_t><<<grid, 256, 0, st>>>((const f16_t*)d->src, x_img, (const f16_t*)d->w, (f16_t*)d->dst, d->dst_ld, d->N, K,
                                                   d->Cd, d->scale, d->shift, d->act);
  else
    dense_fwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)d->src, x_img, (const bf16_t*)d->w, (bf16_t*)d->dst, d->dst_ld, d->N, K,
                                                   d->Cd, d->scale, d->shift, d->act);
  DY_LAUNCH_CHECK();
  return 0;
}

bool dy_dense_dgrad_eligible(const dy_conv_desc* d) {
  return d->Hs == 1 && d->Ws == 1 && d->KH == d->Hd && d->KW == d->Wd && d->pad == 0 && d->dil == 1 && d->stride == 1 && d->KHf == 0 &&
         d->dst_row_stride == 0 && d->KH * d->KW > 1 && d->dst && !d->dst_planar;
}

int dy_dense_dgrad_launch(const dy_conv_desc* d, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int HW = d->Hd * d->Wd;
  const unsigned grid = (unsigned)(((long)d->N * HW * d->Cd + 255) / 256);
  if (d->dtype == DY_F32)
    dense_dgrad_kernel<float><<<grid, 256, 0, st>>>((const float*)d->src, d->src_ld, (const float*)d->w, (float*)d->dst, d->dst_ld, d->N, HW,
                                                    d->Cd, d->Cs, d->accumulate);
  else if ((d->dtype) == DY_F16)
    dense_dgrad_kernel<f16_t><<<grid, 256, 0, st>>>((const f16_t*)d->src, d->src_ld, (const f16_t*)d->w, (f16_t*)d->dst, d->dst_ld, d->N,
                                                     HW, d->Cd, d->Cs, d->accumulate);
  else
    dense_dgrad_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)d->src, d->src_ld, (const bf16_t*)d->w, (bf16_t*)d->dst, d->dst_ld, d->N,
                                                     HW, d->Cd, d->Cs, d->accumulate);
  DY_LAUNCH_CHECK();
  return 0;
}

bool dy_dense_wgrad_eligible(int Hi, int Wi, int Ho, int Wo, int KH, int KW, int pad, int dil) {
  return Ho == 1 && Wo == 1 && KH == Hi && KW == Wi && pad == 0 && dil == 1 && KH * KW > 1;
}

int dy_dense_wgrad_launch(const void* x, long x_ld, int N, int Hi, int Wi, int Cin_pad, const void* dz, long dz_ld, int Cout, int Cin,
                          float* g_oihw, int dtype, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int HW = Hi * Wi;
  const unsigned grid = (unsigned)(((long)Cout * HW * Cin_pad + 255) / 256);
  if (dtype == DY_F32)
    dense_wgrad_kernel<float><<<grid, 256, 0, st>>>((const float*)x, x_ld, (const float*)dz, dz_ld, g_oihw, N, HW, Cin, Cin_pad, Cout);
  else if ((dtype) == DY_F16)
    dense_wgrad_kernel<f16_t><<<grid, 256, 0, st>>>((const f16_t*)x, x_ld, (const f16_t*)dz, dz_ld, g_oihw, N, HW, Cin, Cin_pad, Cout);
  else
    dense_wgrad_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)x, x_ld, (const bf16_t*)dz, dz_ld, g_oihw, N, HW, Cin, Cin_pad, Cout);
  DY_LAUNCH_CHECK();
  return 0;
}
