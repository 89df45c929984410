// Low-light enhancement front-end: lowlight_recovery.forward (reference ultralytics/nn/modules/llie.py:17-54) and the
// filters of ultralytics/nn/modules/filtersB.py (DeDark :190-216, ImprovedWhiteBalance :246-259, Gamma :227-233,
// Contrast :296-303 with rgb2lum util_filters.py:270-273, Usm :151-175).  All fp32 math, HBM-bound.
//
//   x[B,3,H,W] --pointwise chain--> s4 --separable 25-tap gaussian (reflect halo 12) + unsharp combine (usm.hip)--> out
// The reference launches ~12 full-image passes (clone x2, pad, 3 dense 625-tap conv2d, cat, ...); here: one pointwise
// kernel (1R+1W), one USM kernel (1R + 1W f32 + optional NHWC8 copy for the stem conv).  Must-reproduce quirk: `lum` of the contrast filter is a per-(b,c,row) scalar taken from
// pixel columns 0,1,2 of the gamma-filtered image.
#include "dy_common.h"
#include "../../include/dedark_yolo.h"

namespace {

struct FParams { float omega, s[3], gamma, alpha, lam; };

__device__ inline FParams load_params(const float* params, int b) {
  FParams p;
  const float* q = params + b * 8;
  p.omega = q[0]; p.s[0] = q[1]; p.s[1] = q[2]; p.s[2] = q[3]; p.gamma = q[4]; p.alpha = q[5]; p.lam = q[6];
  return p;
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

__global__ void filter_params_bwd_kernel(const float* __restrict__ feat, int feat_ld, const double* __restrict__ dp,
                                         float* __restrict__ df, int B) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* f = feat + (long)b * feat_ld;
  float g[8];
  for (int i = 0; i < 8; ++i) g[i] = (float)dp[b * 8 + i];        // the f64 sums of the f32 block partials, rounded once
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

// ---- pointwise chain: one WAVE per (b, c, row), four rows per block ---------------------------------------------------
// pow(x, g) = exp2(g * log2(x)) on the hardware transcendentals (v_log_f32 / v_exp_f32, ~1e-6 relative for x >= 1e-4,
// g <= 3): the libm powf/logf sequence made both kernels ALU-bound (1.1 ms backward for 471 MB of traffic).  Reductions are
// wave shuffles; the block only meets in LDS to merge its four rows' parameter gradients into one set of atomics.
constexpr int PW_ROWS = 4;
constexpr float LN2 = 0.69314718055994530942f;

struct Px { float s1, s2, base, L, s3, rtx, txr; };

// FAST: hardware transcendentals (throughput / bf16 mode); otherwise libm (f32 parity mode: bit-comparable with torch.pow)
template <bool FAST>
__device__ inline Px chain_px(float x, float A, float I, const FParams& p, int c) {
  Px r;
  r.txr = 1.f - p.omega * I;
  const float tx = fmaxf(r.txr, 0.01f);
  r.rtx = FAST ? __builtin_amdgcn_rcpf(tx) : 1.f / tx;
  r.s1 = FAST ? (x - A) * r.rtx + A : (x - A) / tx + A;
  r.s2 = r.s1 * p.s[c];
  r.base = fmaxf(r.s2, 1e-4f);
  if (FAST) {
    r.L = __builtin_amdgcn_logf(r.base);           // v_log_f32 = log2
    r.s3 = __builtin_amdgcn_exp2f(p.gamma * r.L);  // v_exp_f32 = 2^x
  } else {
    r.L = log2f(r.base);
    r.s3 = powf(r.base, p.gamma);
  }
  return r;
}

template <bool FAST>
__device__ inline float row_lum(const float* xr, const float* ir, float Ac, const FParams& p, int c, int lane, int W, float* lraw_out) {
  float v = 0.f;
  if (lane < 3 && lane < W) v = chain_px<FAST>(xr[lane], Ac, ir ? ir[lane] : 0.5f, p, c).s3;
  const float lraw = 0.27f * __shfl(v, 0, 64) + 0.67f * __shfl(v, 1, 64) + 0.06f * __shfl(v, 2, 64);
  *lraw_out = lraw;
  return fminf(fmaxf(lraw, 0.f), 1.f);
}

template <bool FAST>
__global__ __launch_bounds__(256) void pointwise_fwd_kernel(const float* __restrict__ x, const float* __restrict__ params,
                                                             const float* __restrict__ A, const float* __restrict__ IcA,
                                                             float* __restrict__ s4, int B, int H, int W) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * PW_ROWS + wave;                // (b*3 + c)*H + h
  if (row >= (long)B * 3 * H) return;
  const int h = (int)(row % H), bc = (int)(row / H), c = bc % 3, b = bc / 3;
  const FParams p = load_params(params, b);
  const float Ac = A ? A[b * 3 + c] : 0.8f;
  const float* xr = x + row * W;
  const float* ir = IcA ? IcA + ((long)b * H + h) * W : nullptr;
  float lraw;
  const float lum = row_lum<FAST>(xr, ir, Ac, p, c, lane, W, &lraw);
  // lerp(img, img/(lum+1e-6)*cl, alpha) (util_filters.py:316-317) = img * K with a per-row K
  const float cl = -cosf(3.14159265358979323846f * lum) * 0.5f + 0.5f;
  const float K = (1.f - p.alpha) + p.alpha * (cl / (lum + 1e-6f));
  float* o = s4 + row * W;
  for (int w = lane; w < W; w += 64) o[w] = chain_px<FAST>(xr[w], Ac, ir ? ir[w] : 0.5f, p, c).s3 * K;
}

struct PwAcc { float gamma, wb, om; };

// backward of the chain for one pixel given d(loss)/d(s3); returns d(loss)/dx and accumulates the parameter gradients
template <bool FAST>
__device__ inline float chain_bwd_px(const Px& q, float x, float A, float I, float d3, const FParams& p, int c, PwAcc& a) {
  a.gamma += d3 * q.s3 * (q.L * LN2);
  const float d2 = (q.s2 >= 1e-4f) ? d3 * p.gamma * (FAST ? q.s3 * __builtin_amdgcn_rcpf(q.base) : q.s3 / q.base) : 0.f;
  a.wb += d2 * q.s1;
  const float d1 = d2 * p.s[c];
  if (q.txr >= 0.01f) a.om += d1 * (x - A) * I * q.rtx * q.rtx;
  return d1 * q.rtx;
}

template <bool FAST>
__global__ __launch_bounds__(256) void pointwise_bwd_kernel(const float* __restrict__ x, const float* __restrict__ params,
                                                             const float* __restrict__ A, const float* __restrict__ IcA,
                                                             const float* __restrict__ ds4, float* __restrict__ dx,
                                                             double* dparams, int B, int H, int W, int accumulate) {
  __shared__ float s_part[PW_ROWS][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * PW_ROWS + wave;
  const bool live = row < (long)B * 3 * H;
  const bool merged = (H % PW_ROWS) == 0;                            // then the block's rows share (b, c)
  float t_om = 0.f, t_wb = 0.f, t_gamma = 0.f, t_alpha = 0.f;
  int b = 0, c = 0;
  if (live) {
    const int h = (int)(row % H), bc = (int)(row / H);
    c = bc % 3;
    b = bc / 3;
    const FParams p = load_params(params, b);
    const float Ac = A ? A[b * 3 + c] : 0.8f;
    const float* xr = x + row * W;
    const float* ir = IcA ? IcA + ((long)b * H + h) * W : nullptr;
    const float* gr = ds4 + row * W;
    float* dxr = dx ? dx + row * W : nullptr;
    float lraw;
    const float lum = row_lum<FAST>(xr, ir, Ac, p, c, lane, W, &lraw);
    const float PI = 3.14159265358979323846f;
    const float cl = -cosf(PI * lum) * 0.5f + 0.5f;
    const float q = cl / (lum + 1e-6f);
    const float K = (1.f - p.alpha) + p.alpha * q;
    // every pixel with d3 = d4 * K; the three pixels that feed `lum` get their extra term afterwards
    PwAcc acc = {0.f, 0.f, 0.f};
    float dK = 0.f;
    for (int w = lane; w < W; w += 64) {
      const float I = ir ? ir[w] : 0.5f, xv = xr[w], g4 = gr[w];
      const Px px = chain_px<FAST>(xv, Ac, I, p, c);
      dK += g4 * px.s3;
      const float g = chain_bwd_px<FAST>(px, xv, Ac, I, g4 * K, p, c, acc);
      if (dxr) dxr[w] = accumulate ? dxr[w] + g : g;
    }
    dK = wave_sum(dK);
    t_alpha = dK * (q - 1.f);
    // d q / d lum, gated by clamp(lum, 0, 1) (inclusive bounds pass the gradient, as torch.clamp does)
    float dlum = 0.f;
    if (lraw >= 0.f && lraw <= 1.f) {
      const float dcl = 0.5f * PI * sinf(PI * lum);
      dlum = dK * p.alpha * (dcl * (lum + 1e-6f) - cl) / ((lum + 1e-6f) * (lum + 1e-6f));
    }
    if (lane < 3 && lane < W) {
      const float coef = lane == 0 ? 0.27f : (lane == 1 ? 0.67f : 0.06f);
      const float I = ir ? ir[lane] : 0.5f, xv = xr[lane];
      const Px px = chain_px<FAST>(xv, Ac, I, p, c);
      const float g = chain_bwd_px<FAST>(px, xv, Ac, I, coef * dlum, p, c, acc);
      if (dxr) dxr[lane] += g;                                        // same lane wrote dxr[lane] in the loop above
    }
    t_om = wave_sum(acc.om);
    t_wb = wave_sum(acc.wb);
    t_gamma = wave_sum(acc.gamma);
  }
  if (merged) {
    if (lane == 0) {
      s_part[wave][0] = t_om; s_part[wave][1] = t_wb; s_part[wave][2] = t_gamma; s_part[wave][3] = t_alpha;
    }
    __syncthreads();
    if (threadIdx.x < 4 && (long)blockIdx.x * PW_ROWS < (long)B * 3 * H) {
      const int bc0 = (int)(((long)blockIdx.x * PW_ROWS) / H);
      const int k = threadIdx.x;
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < PW_ROWS; ++r) v += s_part[r][k];
      // f64 atomics: the sum of ~500 f32 block partials per image does not depend on their arrival order beyond 2^-52 relative
      // (f32 atomics here made every run of a training step differ from the last one in the regressor's gradients)
      double* dp = dparams + (bc0 / 3) * 8;
      atomic_add_f64(dp + (k == 0 ? 0 : (k == 1 ? 1 + bc0 % 3 : (k == 2 ? 4 : 5))), (double)v);
    }
  } else if (live && lane == 0) {
    double* dp = dparams + b * 8;
    atomic_add_f64(dp + 0, (double)t_om);
    atomic_add_f64(dp + 1 + c, (double)t_wb);
    atomic_add_f64(dp + 4, (double)t_gamma);
    atomic_add_f64(dp + 5, (double)t_alpha);
  }
}

inline int ew_blocks(long total) {
  long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int dy_image_to_nhwc8(const float* x, int B, int H, int W, void* y, int Ho, int Wo, int dtype, void* stream) {
  DY_CHECK(x && y && B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "dy_image_to_nhwc8: bad args");
  const int blocks = ew_blocks((long)B * Ho * Wo);
  if (dtype == DY_F32) image_to_nhwc8_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>(x, B, H, W, (float*)y, Ho, Wo);
  else if ((dtype) == DY_F16) image_to_nhwc8_kernel<f16_t><<<blocks, 256, 0, (hipStream_t)stream>>>(x, B, H, W, (f16_t*)y, Ho, Wo);
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

extern "C" int dy_filter_params_bwd(const float* feat, int feat_ld, const double* dparams, float* dfeat, int B, void* stream) {
  DY_CHECK(feat && dparams && dfeat && B > 0 && feat_ld >= 15, "dy_filter_params_bwd: bad args");
  filter_params_bwd_kernel<<<dy_cdiv(B, 64), 64, 0, (hipStream_t)stream>>>(feat, feat_ld, dparams, dfeat, B);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_filters_pointwise_fwd(const float* x, const float* params, const float* A, const float* IcA, float* s4, int B,
                                        int H, int W, int fast_math, void* stream) {
  DY_CHECK(x && params && s4 && B > 0 && H > 0 && W >= 3, "dy_filters_pointwise_fwd: bad args (W must be >= 3)");
  const unsigned grid = dy_cdiv((long)B * 3 * H, PW_ROWS);
  if (fast_math) pointwise_fwd_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(x, params, A, IcA, s4, B, H, W);
  else pointwise_fwd_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(x, params, A, IcA, s4, B, H, W);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_filters_pointwise_bwd(const float* x, const float* params, const float* A, const float* IcA,
                                        const float* ds4, float* dx, double* dparams, int B, int H, int W, int accumulate,
                                        int fast_math, void* stream) {
  DY_CHECK(x && params && ds4 && dparams && B > 0 && H > 0 && W >= 3, "dy_filters_pointwise_bwd: bad args");
  const unsigned grid = dy_cdiv((long)B * 3 * H, PW_ROWS);
  if (fast_math) pointwise_bwd_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(x, params, A, IcA, ds4, dx, dparams, B, H, W, accumulate);
  else pointwise_bwd_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(x, params, A, IcA, ds4, dx, dparams, B, H, W, accumulate);
  DY_LAUNCH_CHECK();
  return 0;
}
