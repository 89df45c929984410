// Low-light enhancement front-end: lowlight_recovery.forward (reference ultralytics/nn/modules/llie.py:17-54) and the
// filters of ultralytics/nn/modules/filtersB.py (DeDark :190-216, ImprovedWhiteBalance :246-259, Gamma :227-233,
// Contrast :296-303 with rgb2lum util_filters.py:270-273, Usm :151-175).  All fp32 math, HBM-bound.
//
//   x[B,3,H,W] --pointwise chain--> s4 --separable 25-tap gaussian (reflect halo 12) + unsharp combine--> out
// The reference launches ~12 full-image passes (clone x2, pad, 3 dense 625-tap conv2d, cat, ...); here: one pointwise
// kernel (1R+1W), one USM kernel (1R + 1W f32 + optional NHWC8 copy for the stem conv).  The blur is done separably in
// LDS (50 taps instead of 625).  Must-reproduce quirk: `lum` of the contrast filter is a per-(b,c,row) scalar taken from
// pixel columns 0,1,2 of the gamma-filtered image.
#include "dy_common.h"
#include "../../include/dedark_yolo.h"

namespace {

constexpr int R = 12;           // gaussian radius
__constant__ float c_taps[R + 1];   // k[|d|], sigma 5, normalised (filtersB.py:152-161)

struct FParams { float omega, s[3], gamma, alpha, lam; };

__device__ inline FParams load_params(const float* params, int b) {
  FParams p;
  const float* q = params + b * 8;
  p.omega = q[0]; p.s[0] = q[1]; p.s[1] = q[2]; p.s[2] = q[3]; p.gamma = q[4]; p.alpha = q[5]; p.lam = q[6];
  return p;
}

// chain up to the gamma filter for one pixel
__device__ inline float chain_s3(float x, float A, float I, const FParams& p, int c, float* s1_out, float* s2_out) {
  float tx = fmaxf(1.f - p.omega * I, 0.01f);
  float s1 = (x - A) / tx + A;
  float s2 = s1 * p.s[c];
  if (s1_out) *s1_out = s1;
  if (s2_out) *s2_out = s2;
  return powf(fmaxf(s2, 1e-4f), p.gamma);
}

__device__ inline float contrast_gain(float lum, float alpha) {
  float cl = -cosf(3.14159265358979323846f * lum) * 0.5f + 0.5f;
  return (1.f - alpha) + alpha * cl / (lum + 1e-6f);
}

// ---- image relayout (+ optional bilinear resize, align_corners=False) ------------------------------------------------
template <typename T>
__global__ void image_to_nhwc8_kernel(const float* __restrict__ x, int B, int H, int W, T* __restrict__ y, int Ho, int Wo) {
  const long total = (long)B * Ho * Wo;
  const bool resize = (Ho != H) || (Wo != W);
  const float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int wo = (int)(i % Wo);
    long t = i / Wo;
    int ho = (int)(t % Ho);
    int b = (int)(t / Ho);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (!resize) {
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = x[(((long)b * 3 + c) * H + ho) * W + wo];
    } else {
      float hr = fmaxf(sh * (ho + 0.5f) - 0.5f, 0.f), wr = fmaxf(sw * (wo + 0.5f) - 0.5f, 0.f);
      int h1 = (int)hr, w1 = (int)wr;
      int hp = h1 < H - 1 ? 1 : 0, wp = w1 < W - 1 ? 1 : 0;
      float hl = hr - h1, wl = wr - w1;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* pl = x + (((long)b * 3 + c) * H + h1) * W + w1;
        float top = (1.f - wl) * pl[0] + wl * pl[wp];
        float bot = (1.f - wl) * pl[(long)hp * W] + wl * pl[(long)hp * W + wp];
        v[c] = (1.f - hl) * top + hl * bot;
      }
    }
    if constexpr (sizeof(T) == 4) {
      stvec<T>(y + i * 8, v);
      stvec<T>(y + i * 8 + 4, v + 4);
    } else {
      stvec<T>(y + i * 8, v);
    }
  }
}

__global__ void resize_bwd_kernel(const float* __restrict__ dy, int dy_ld, int B, int H, int W, int Ho, int Wo, float* dx) {
  const long total = (long)B * Ho * Wo;
  const float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int wo = (int)(i % Wo);
    long t = i / Wo;
    int ho = (int)(t % Ho);
    int b = (int)(t / Ho);
    float hr = fmaxf(sh * (ho + 0.5f) - 0.5f, 0.f), wr = fmaxf(sw * (wo + 0.5f) - 0.5f, 0.f);
    int h1 = (int)hr, w1 = (int)wr;
    int hp = h1 < H - 1 ? 1 : 0, wp = w1 < W - 1 ? 1 : 0;
    float hl = hr - h1, wl = wr - w1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float g = dy[i * dy_ld + c];
      float* pl = dx + (((long)b * 3 + c) * H + h1) * W + w1;
      atomic_add_f32(pl, (1.f - hl) * (1.f - wl) * g);
      atomic_add_f32(pl + wp, (1.f - hl) * wl * g);
      atomic_add_f32(pl + (long)hp * W, hl * (1.f - wl) * g);
      atomic_add_f32(pl + (long)hp * W + wp, hl * wl * g);
    }
  }
}

// ---- feat[15] -> params[8] ------------------------------------------------------------------------------------------
__global__ void filter_params_fwd_kernel(const float* __restrict__ feat, int feat_ld, float* __restrict__ params, int B) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* f = feat + (long)b * feat_ld;
  float* p = params + b * 8;
  p[0] = tanhf(f[0]) * 0.45f + 0.55f;                       // tanh_range(0.1, 1.0)
  float e0 = expf(tanhf(f[1] * 0.f) * 0.5f), e1 = expf(tanhf(f[2]) * 0.5f), e2 = expf(tanhf(f[3]) * 0.5f);
  float D = 1e-5f + 0.27f * e0 + 0.67f * e1 + 0.06f * e2;
  p[1] = e0 / D; p[2] = e1 / D; p[3] = e2 / D;
  p[4] = expf(tanhf(f[4]) * 1.0986122886681098f);           // ln 3
  p[5] = tanhf(f[13]);
  p[6] = tanhf(f[14]) * 2.5f + 2.5f;                         // tanh_range(0, 5)
  p[7] = 0.f;
}

__global__ void filter_params_bwd_kernel(const float* __restrict__ feat, int feat_ld, const float* __restrict__ dp,
                                         float* __restrict__ df, int B) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* f = feat + (long)b * feat_ld;
  const float* g = dp + b * 8;
  float* o = df + (long)b * feat_ld;
  for (int i = 0; i < feat_ld; ++i) o[i] = 0.f;
  float t0 = tanhf(f[0]);
  o[0] = g[0] * 0.45f * (1.f - t0 * t0);
  float tt[3] = {0.f, tanhf(f[2]), tanhf(f[3])};
  float e[3] = {expf(0.f), expf(tt[1] * 0.5f), expf(tt[2] * 0.5f)};
  const float cf[3] = {0.27f, 0.67f, 0.06f};
  float D = 1e-5f + cf[0] * e[0] + cf[1] * e[1] + cf[2] * e[2];
  float dot = g[1] * e[0] + g[2] * e[1] + g[3] * e[2];
  for (int j = 1; j < 3; ++j) {                              // R slot is masked: zero gradient
    float de = g[1 + j] / D - cf[j] * dot / (D * D);
    o[1 + j] = de * e[j] * 0.5f * (1.f - tt[j] * tt[j]);
  }
  float t4 = tanhf(f[4]);
  float gam = expf(t4 * 1.0986122886681098f);
  o[4] = g[4] * gam * 1.0986122886681098f * (1.f - t4 * t4);
  float t13 = tanhf(f[13]);
  o[13] = g[5] * (1.f - t13 * t13);
  float t14 = tanhf(f[14]);
  o[14] = g[6] * 2.5f * (1.f - t14 * t14);
}

// ---- pointwise chain: one block per (b, c, row) ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void pointwise_fwd_kernel(const float* __restrict__ x, const float* __restrict__ params,
                                                             const float* __restrict__ A, const float* __restrict__ IcA,
                                                             float* __restrict__ s4, int B, int H, int W) {
  __shared__ float s_l[3];
  const int row = blockIdx.x;                // (b*3 + c)*H + h
  const int h = row % H, bc = row / H, c = bc % 3, b = bc / 3;
  const FParams p = load_params(params, b);
  const float Ac = A ? A[b * 3 + c] : 0.8f;
  const float* xr = x + (long)row * W;
  const float* ir = IcA ? IcA + ((long)b * H + h) * W : nullptr;
  if (threadIdx.x < 3) s_l[threadIdx.x] = chain_s3(xr[threadIdx.x], Ac, ir ? ir[threadIdx.x] : 0.5f, p, c, nullptr, nullptr);
  __syncthreads();
  const float lum = fminf(fmaxf(0.27f * s_l[0] + 0.67f * s_l[1] + 0.06f * s_l[2], 0.f), 1.f);
  const float K = contrast_gain(lum, p.alpha);
  float* o = s4 + (long)row * W;
  for (int w = threadIdx.x; w < W; w += blockDim.x) {
    float s3 = chain_s3(xr[w], Ac, ir ? ir[w] : 0.5f, p, c, nullptr, nullptr);
    // lerp(img, img/(lum+1e-6)*cl, alpha) written as the reference does (util_filters.py:316-317)
    float cl = -cosf(3.14159265358979323846f * lum) * 0.5f + 0.5f;
    float ci = s3 / (lum + 1e-6f) * cl;
    o[w] = (1.f - p.alpha) * s3 + p.alpha * ci;
    (void)K;
  }
}

__global__ __launch_bounds__(256) void pointwise_bwd_kernel(const float* __restrict__ x, const float* __restrict__ params,
                                                             const float* __restrict__ A, const float* __restrict__ IcA,
                                                             const float* __restrict__ ds4, float* __restrict__ dx,
                                                             float* dparams, int B, int H, int W, int accumulate) {
  __shared__ float s_l[3];
  __shared__ float sm[20];
  const int row = blockIdx.x;
  const int h = row % H, bc = row / H, c = bc % 3, b = bc / 3;
  const FParams p = load_params(params, b);
  const float Ac = A ? A[b * 3 + c] : 0.8f;
  const float* xr = x + (long)row * W;
  const float* ir = IcA ? IcA + ((long)b * H + h) * W : nullptr;
  const float* gr = ds4 + (long)row * W;
  if (threadIdx.x < 3) s_l[threadIdx.x] = chain_s3(xr[threadIdx.x], Ac, ir ? ir[threadIdx.x] : 0.5f, p, c, nullptr, nullptr);
  __syncthreads();
  const float lraw = 0.27f * s_l[0] + 0.67f * s_l[1] + 0.06f * s_l[2];
  const float lum = fminf(fmaxf(lraw, 0.f), 1.f);
  const float PI = 3.14159265358979323846f;
  const float cl = -cosf(PI * lum) * 0.5f + 0.5f;
  const float q = cl / (lum + 1e-6f);
  const float K = (1.f - p.alpha) + p.alpha * q;
  // phase 1: dK_row = sum_w d4*s3 (also d alpha)
  float part = 0.f;
  for (int w = threadIdx.x; w < W; w += blockDim.x)
    part += gr[w] * chain_s3(xr[w], Ac, ir ? ir[w] : 0.5f, p, c, nullptr, nullptr);
  const float dK = block_sum(part, sm);
  const float d_alpha = dK * (q - 1.f);
  // d q / d lum, gated by clamp(lum, 0, 1) (inclusive bounds pass the gradient, as torch.clamp does)
  float dlum = 0.f;
  if (lraw >= 0.f && lraw <= 1.f) {
    float dcl = 0.5f * PI * sinf(PI * lum);
    dlum = dK * p.alpha * (dcl * (lum + 1e-6f) - cl) / ((lum + 1e-6f) * (lum + 1e-6f));
  }
  // phase 2
  float a_gamma = 0.f, a_wb = 0.f, a_om = 0.f;
  float* dxr = dx + (long)row * W;
  for (int w = threadIdx.x; w < W; w += blockDim.x) {
    float I = ir ? ir[w] : 0.5f;
    float s1, s2;
    float s3 = chain_s3(xr[w], Ac, I, p, c, &s1, &s2);
    float d3 = gr[w] * K;
    if (w == 0) d3 += 0.27f * dlum;
    else if (w == 1) d3 += 0.67f * dlum;
    else if (w == 2) d3 += 0.06f * dlum;
    float base = fmaxf(s2, 1e-4f);
    a_gamma += d3 * s3 * logf(base);
    float d2 = (s2 >= 1e-4f) ? d3 * p.gamma * powf(base, p.gamma - 1.f) : 0.f;
    a_wb += d2 * s1;
    float d1 = d2 * p.s[c];
    float txr = 1.f - p.omega * I;
    float tx = fmaxf(txr, 0.01f);
    if (txr >= 0.01f) a_om += d1 * (xr[w] - Ac) * I / (tx * tx);
    float g = d1 / tx;
    if (dx) dxr[w] = accumulate ? dxr[w] + g : g;
  }
  a_gamma = block_sum(a_gamma, sm);
  a_wb = block_sum(a_wb, sm);
  a_om = block_sum(a_om, sm);
  if (threadIdx.x == 0) {
    float* dp = dparams + b * 8;
    atomic_add_f32(dp + 0, a_om);
    atomic_add_f32(dp + 1 + c, a_wb);
    atomic_add_f32(dp + 4, a_gamma);
    atomic_add_f32(dp + 5, d_alpha);
  }
}

// ---- USM: separable gaussian through LDS ----------------------------------------------------------------------------
constexpr int TH = 16, TW = 64;
constexpr int LH = TH + 2 * R, LW = TW + 2 * R;

__device__ inline int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

template <typename T>
__global__ __launch_bounds__(256) void usm_fwd_kernel(const float* __restrict__ s4, const float* __restrict__ params,
                                                       float* __restrict__ out, T* __restrict__ out8, float* __restrict__ hp,
                                                       int B, int H, int W) {
  __shared__ float tile[LH][LW + 1];
  __shared__ float tmp[LH][TW + 1];
  const int b = blockIdx.z, y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  const float lam = params[b * 8 + 6];
  float res[3][4];
  for (int c = 0; c < 3; ++c) {
    const float* pl = s4 + ((long)b * 3 + c) * H * W;
    __syncthreads();
    for (int i = tid; i < LH * LW; i += 256) {
      int r = i / LW, q = i - r * LW;
      int yy = reflect(y0 + r - R, H), xx = reflect(x0 + q - R, W);
      yy = min(max(yy, 0), H - 1);
      xx = min(max(xx, 0), W - 1);
      tile[r][q] = pl[(long)yy * W + xx];
    }
    __syncthreads();
    for (int i = tid; i < LH * TW; i += 256) {
      int r = i / TW, q = i - r * TW;
      float a = c_taps[0] * tile[r][q + R];
#pragma unroll
      for (int d = 1; d <= R; ++d) a += c_taps[d] * (tile[r][q + R - d] + tile[r][q + R + d]);
      tmp[r][q] = a;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int r = ty + 4 * j;
      float a = c_taps[0] * tmp[r + R][tx];
#pragma unroll
      for (int d = 1; d <= R; ++d) a += c_taps[d] * (tmp[r + R - d][tx] + tmp[r + R + d][tx]);
      float v = tile[r + R][tx + R];
      float hi = v - a;
      float o = hi * lam + v;
      res[c][j] = o;
      int yy = y0 + r, xx = x0 + tx;
      if (yy < H && xx < W) {
        long idx = (((long)b * 3 + c) * H + yy) * W + xx;
        if (out) out[idx] = o;
        if (hp) hp[idx] = hi;
      }
    }
  }
  if (out8) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int yy = y0 + ty + 4 * j, xx = x0 + tx;
      if (yy < H && xx < W) {
        float v[8] = {res[0][j], res[1][j], res[2][j], 0.f, 0.f, 0.f, 0.f, 0.f};
        T* o = out8 + (((long)b * H + yy) * W + xx) * 8;
        if constexpr (sizeof(T) == 4) {
          stvec<T>(o, v);
          stvec<T>(o + 4, v + 4);
        } else {
          stvec<T>(o, v);
        }
      }
    }
  }
}

// adjoint weight of the reflect-padded blur along one axis: d(blur[i]) / d(s[m]) for i = m + d
// `taps` must point to LDS: the index differs per lane, and a divergent index into __constant__ memory is executed as a
// waterfall loop over the distinct values (this made the first version of usm_bwd 10x slower than usm_fwd).
__device__ inline float adj_w(const float* taps, int m, int d, int n) {
  int ad = d < 0 ? -d : d;
  float w = taps[ad];
  if (m >= 1 && m <= R) {
    int t = 2 * m + d;
    t = t < 0 ? -t : t;
    if (t <= R) w += taps[t];
  }
  if (m >= n - 1 - R && m <= n - 2) {
    int t = 2 * (n - 1) - 2 * m - d;
    t = t < 0 ? -t : t;
    if (t <= R) w += taps[t];
  }
  return w;
}

template <typename T>
__global__ __launch_bounds__(256) void usm_bwd_kernel(const float* __restrict__ dout, const T* __restrict__ dout8, int ld8,
                                                       const float* __restrict__ hp, const float* __restrict__ params,
                                                       float* __restrict__ ds4, float* dparams, int B, int H, int W) {
  // all three channels of the tile are staged at once: one 16-byte load per pixel of the NHWC gradient instead of three
  // 2-byte loads at a 16-byte stride (the first version of this kernel spent 3.6 ms there)
  __shared__ float tile[3][LH][LW + 1];
  __shared__ float tmp[LH][TW + 1];
  __shared__ float sm[20];
  __shared__ float s_taps[R + 1];
  if (threadIdx.x <= R) s_taps[threadIdx.x] = c_taps[threadIdx.x];
  constexpr int VE = DT<T>::VE;
  const int b = blockIdx.z, y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  const float lam = params[b * 8 + 6];
  float dl = 0.f;
  for (int i = tid; i < LH * LW; i += 256) {
    int r = i / LW, q = i - r * LW;
    int yy = y0 + r - R, xx = x0 + q - R;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
      if (dout) {
        const float* pl = dout + (((long)b * 3) * H + yy) * W + xx;
        v0 = pl[0]; v1 = pl[(long)H * W]; v2 = pl[2L * H * W];
      } else if (ld8 == VE) {
        float v[VE];
        ldvec<T>(dout8 + (((long)b * H + yy) * W + xx) * VE, v);
        v0 = v[0]; v1 = v[1]; v2 = v[2];
      } else {
        const T* pp = dout8 + (((long)b * H + yy) * W + xx) * ld8;
        v0 = DT<T>::ld(pp); v1 = DT<T>::ld(pp + 1); v2 = DT<T>::ld(pp + 2);
      }
    }
    tile[0][r][q] = v0; tile[1][r][q] = v1; tile[2][r][q] = v2;
  }
  __syncthreads();
  for (int c = 0; c < 3; ++c) {
    for (int i = tid; i < LH * TW; i += 256) {
      int r = i / TW, q = i - r * TW;
      int m = x0 + q;
      float a = 0.f;
      if (m < W) {
        const bool border = (m <= R) || (m >= W - 1 - R);
        if (!border) {
          a = c_taps[0] * tile[c][r][q + R];
#pragma unroll
          for (int d = 1; d <= R; ++d) a += c_taps[d] * (tile[c][r][q + R - d] + tile[c][r][q + R + d]);
        } else {
          for (int d = -R; d <= R; ++d) a += adj_w(s_taps, m, d, W) * tile[c][r][q + R + d];
        }
      }
      tmp[r][q] = a;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int r = ty + 4 * j;
      int m = y0 + r, xx = x0 + tx;
      if (m < H && xx < W) {
        float a = 0.f;
        const bool border = (m <= R) || (m >= H - 1 - R);
        if (!border) {
          a = c_taps[0] * tmp[r + R][tx];
#pragma unroll
          for (int d = 1; d <= R; ++d) a += c_taps[d] * (tmp[r + R - d][tx] + tmp[r + R + d][tx]);
        } else {
          for (int d = -R; d <= R; ++d) a += adj_w(s_taps, m, d, H) * tmp[r + R + d][tx];
        }
        float g = tile[c][r + R][tx + R];
        long idx = (((long)b * 3 + c) * H + m) * W + xx;
        ds4[idx] = g * (1.f + lam) - lam * a;
        dl += g * hp[idx];
      }
    }
    __syncthreads();
  }
  dl = block_sum(dl, sm);
  if (tid == 0) atomic_add_f32(dparams + b * 8 + 6, dl);
}

bool g_taps_ready = false;
int ensure_taps() {
  if (g_taps_ready) return 0;
  float k[2 * R + 1];
  float sum = 0.f;
  for (int i = -R; i <= R; ++i) {
    float xv = (float)i / 5.0f;
    k[i + R] = expf(-0.5f * (xv * xv));
    sum += k[i + R];
  }
  float taps[R + 1];
  for (int d = 0; d <= R; ++d) taps[d] = k[R + d] / sum;
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_taps), taps, sizeof(taps));
  if (e != hipSuccess) {
    dy_set_error("frontend: hipMemcpyToSymbol failed: %s", hipGetErrorString(e));
    return 3;
  }
  g_taps_ready = true;
  return 0;
}

inline int ew_blocks(long total) {
  long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int dy_frontend_init(void) { return ensure_taps(); }

extern "C" int dy_image_to_nhwc8(const float* x, int B, int H, int W, void* y, int Ho, int Wo, int dtype, void* stream) {
  DY_CHECK(x && y && B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "dy_image_to_nhwc8: bad args");
  const int blocks = ew_blocks((long)B * Ho * Wo);
  if (dtype == DY_F32) image_to_nhwc8_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>(x, B, H, W, (float*)y, Ho, Wo);
  else image_to_nhwc8_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>(x, B, H, W, (bf16_t*)y, Ho, Wo);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_resize_bwd(const float* dy, int dy_ld, int B, int H, int W, int Ho, int Wo, float* dx, void* stream) {
  DY_CHECK(dy && dx && dy_ld >= 3, "dy_resize_bwd: bad args");
  resize_bwd_kernel<<<ew_blocks((long)B * Ho * Wo), 256, 0, (hipStream_t)stream>>>(dy, dy_ld, B, H, W, Ho, Wo, dx);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_filter_params_fwd(const float* feat, int feat_ld, float* params, int B, void* stream) {
  DY_CHECK(feat && params && B > 0 && feat_ld >= 15, "dy_filter_params_fwd: bad args");
  filter_params_fwd_kernel<<<dy_cdiv(B, 64), 64, 0, (hipStream_t)stream>>>(feat, feat_ld, params, B);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_filter_params_bwd(const float* feat, int feat_ld, const float* dparams, float* dfeat, int B, void* stream) {
  DY_CHECK(feat && dparams && dfeat && B > 0 && feat_ld >= 15, "dy_filter_params_bwd: bad args");
  filter_params_bwd_kernel<<<dy_cdiv(B, 64), 64, 0, (hipStream_t)stream>>>(feat, feat_ld, dparams, dfeat, B);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_filters_pointwise_fwd(const float* x, const float* params, const float* A, const float* IcA, float* s4, int B,
                                        int H, int W, void* stream) {
  DY_CHECK(x && params && s4 && B > 0 && H > 0 && W >= 3, "dy_filters_pointwise_fwd: bad args (W must be >= 3)");
  pointwise_fwd_kernel<<<B * 3 * H, 256, 0, (hipStream_t)stream>>>(x, params, A, IcA, s4, B, H, W);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_filters_pointwise_bwd(const float* x, const float* params, const float* A, const float* IcA,
                                        const float* ds4, float* dx, float* dparams, int B, int H, int W, int accumulate,
                                        void* stream) {
  DY_CHECK(x && params && ds4 && dparams && B > 0 && H > 0 && W >= 3, "dy_filters_pointwise_bwd: bad args");
  pointwise_bwd_kernel<<<B * 3 * H, 256, 0, (hipStream_t)stream>>>(x, params, A, IcA, ds4, dx, dparams, B, H, W, accumulate);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_usm_fwd(const float* s4, const float* params, float* out_nchw, void* out_nhwc8, float* hp, int B, int H, int W,
                          int dtype, void* stream) {
  DY_CHECK(s4 && params && B > 0, "dy_usm_fwd: bad args");
  DY_CHECK(H > R && W > R, "dy_usm_fwd: reflect padding needs H, W > %d", R);
  if (int e = ensure_taps()) return e;
  dim3 grid(dy_cdiv(W, TW), dy_cdiv(H, TH), B);
  if (dtype == DY_F32) usm_fwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(s4, params, out_nchw, (float*)out_nhwc8, hp, B, H, W);
  else usm_fwd_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>(s4, params, out_nchw, (bf16_t*)out_nhwc8, hp, B, H, W);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_usm_bwd(const float* dout_nchw, const void* dout_nhwc8, int dout_ld, const float* hp, const float* params,
                          float* ds4, float* dparams, int B, int H, int W, int dtype, void* stream) {
  DY_CHECK(dout_nhwc8 == nullptr || dout_ld >= 3, "dy_usm_bwd: bad dout_ld");
  DY_CHECK((dout_nchw != nullptr) != (dout_nhwc8 != nullptr), "dy_usm_bwd: exactly one of dout_nchw / dout_nhwc8");
  DY_CHECK(hp && params && ds4 && dparams && B > 0 && H > R && W > R, "dy_usm_bwd: bad args");
  if (int e = ensure_taps()) return e;
  dim3 grid(dy_cdiv(W, TW), dy_cdiv(H, TH), B);
  if (dtype == DY_F32)
    usm_bwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(dout_nchw, (const float*)dout_nhwc8, dout_ld, hp, params, ds4, dparams, B, H, W);
  else
    usm_bwd_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>(dout_nchw, (const bf16_t*)dout_nhwc8, dout_ld, hp, params, ds4, dparams, B, H, W);
  DY_LAUNCH_CHECK();
  return 0;
}
