// Pooling / resampling / concat / ASFF blend kernels (NHWC, 16-byte vectors along C). All HBM-bound.
// Replaces MaxPool2d in SPPF (reference ultralytics/nn/modules/block.py:331-338) and ASFF (block.py:58,85-86),
// nn.Upsample / F.interpolate(nearest) (yolov8.yaml head; block.py:91,97,99), torch.cat / chunk (conv.py:473,
// block.py:385-387), the ASFF softmax blend (block.py:103-111).
#include "dy_common.h"
#include "../../include/dedark_yolo.h"

namespace {

inline int ew_blocks(long total) {
  long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// ---- max pool: first maximum in (kh, kw) scan order wins (ATen max_pool2d semantics); arg = kh*k + kw ---------------
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, long x_ld, T* __restrict__ y, long y_ld,
                                   uint8_t* __restrict__ arg, int N, int H, int W, int C, int k, int stride, int pad, int Ho,
                                   int Wo) {
  constexpr int VE = DT<T>::VE;
  const int CG = C / VE;
  const long total = (long)N * Ho * Wo * CG;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long px = i / CG;
    int c = (int)(i - px * CG) * VE;
    int wo = (int)(px % Wo);
    long t = px / Wo;
    int ho = (int)(t % Ho);
    int n = (int)(t / Ho);
    float best[VE];
    int bi[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    bool first = true;
    for (int kh = 0; kh < k; ++kh) {
      int h = ho * stride - pad + kh;
      if (h < 0 || h >= H) continue;
      for (int kw = 0; kw < k; ++kw) {
        int w = wo * stride - pad + kw;
        if (w < 0 || w >= W) continue;
        float v[VE];
        ldvec<T>(x + (((long)n * H + h) * W + w) * x_ld + c, v);
#pragma unroll
        for (int e = 0; e < VE; ++e)
          if (first || v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = kh * k + kw; }
        first = false;
      }
    }
    stvec<T>(y + px * y_ld + c, best);
    if (arg) {
#pragma unroll
      for (int e = 0; e < VE; ++e) arg[px * C + c + e] = (uint8_t)bi[e];
    }
  }
}

// gather form of the adjoint: each input pixel sums dy of the windows whose argmax points at it (no atomics)
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, long dy_ld, const uint8_t* __restrict__ arg, T* __restrict__ dx,
                                   long dx_ld, int N, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo,
                                   int accumulate) {
  constexpr int VE = DT<T>::VE;
  const int CG = C / VE;
  const long total = (long)N * H * W * CG;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long px = i / CG;
    int c = (int)(i - px * CG) * VE;
    int w = (int)(px % W);
    long t = px / W;
    int h = (int)(t % H);
    int n = (int)(t / H);
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
    for (int kh = 0; kh < k; ++kh) {
      int th = h + pad - kh;
      if (th < 0 || th % stride) continue;
      int ho = th / stride;
      if (ho >= Ho) continue;
      for (int kw = 0; kw < k; ++kw) {
        int tw = w + pad - kw;
        if (tw < 0 || tw % stride) continue;
        int wo = tw / stride;
        if (wo >= Wo) continue;
        long opx = ((long)n * Ho + ho) * Wo + wo;
        float g[VE];
        ldvec<T>(dy + opx * dy_ld + c, g);
        const uint8_t* a = arg + opx * C + c;
#pragma unroll
        for (int e = 0; e < VE; ++e)
          if (a[e] == kh * k + kw) acc[e] += g[e];
      }
    }
    T* o = dx + px * dx_ld + c;
    if (accumulate) {
      float old[VE];
      ldvec<T>(o, old);
#pragma unroll
      for (int e = 0; e < VE; ++e) acc[e] += old[e];
    }
    stvec<T>(o, acc);
  }
}

template <typename T>
__global__ void upsample_fwd_kernel(const T* __restrict__ x, long x_ld, T* __restrict__ y, long y_ld, int N, int H, int W,
                                    int C, int s) {
  constexpr int VE = DT<T>::VE;
  const int CG = C / VE, Ho = H * s, Wo = W * s;
  const long total = (long)N * Ho * Wo * CG;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long px = i / CG;
    int c = (int)(i - px * CG) * VE;
    int wo = (int)(px % Wo);
    long t = px / Wo;
    int ho = (int)(t % Ho);
    int n = (int)(t / Ho);
    u32x4 v = *reinterpret_cast<const u32x4*>(x + (((long)n * H + ho / s) * W + wo / s) * x_ld + c);
    *reinterpret_cast<u32x4*>(y + px * y_ld + c) = v;
  }
}

template <typename T>
__global__ void upsample_bwd_kernel(const T* __restrict__ dy, long dy_ld, T* __restrict__ dx, long dx_ld, int N, int H, int W,
                                    int C, int s, int accumulate) {
  constexpr int VE = DT<T>::VE;
  const int CG = C / VE, Ho = H * s, Wo = W * s;
  const long total = (long)N * H * W * CG;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long px = i / CG;
    int c = (int)(i - px * CG) * VE;
    int w = (int)(px % W);
    long t = px / W;
    int h = (int)(t % H);
    int n = (int)(t / H);
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
    for (int a = 0; a < s; ++a)
      for (int b = 0; b < s; ++b) {
        float g[VE];
        ldvec<T>(dy + (((long)n * Ho + h * s + a) * Wo + w * s + b) * dy_ld + c, g);
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[e] += g[e];
      }
    T* o = dx + px * dx_ld + c;
    if (accumulate) {
      float old[VE];
      ldvec<T>(o, old);
#pragma unroll
      for (int e = 0; e < VE; ++e) acc[e] += old[e];
    }
    stvec<T>(o, acc);
  }
}

template <typename T>
__global__ void copy2d_kernel(const T* __restrict__ src, long src_ld, T* __restrict__ dst, long dst_ld, long pixels, int C,
                              int accumulate) {
  constexpr int VE = DT<T>::VE;
  const int CG = C / VE;
  const long total = pixels * CG;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long px = i / CG;
    int c = (int)(i - px * CG) * VE;
    if (accumulate) {
      float a[VE], b[VE];
      ldvec<T>(src + px * src_ld + c, a);
      ldvec<T>(dst + px * dst_ld + c, b);
#pragma unroll
      for (int e = 0; e < VE; ++e) a[e] += b[e];
      stvec<T>(dst + px * dst_ld + c, a);
    } else {
      *reinterpret_cast<u32x4*>(dst + px * dst_ld + c) = *reinterpret_cast<const u32x4*>(src + px * src_ld + c);
    }
  }
}

template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    DT<D>::st(dst + i, DT<S>::ld(src + i));
}

// ---- ASFF blend: one wave per pixel, lanes stride over channel vectors ---------------------------------------------
template <typename T>
__global__ void asff_fwd_kernel(const T* __restrict__ x0, long ld0, const T* __restrict__ x1, long ld1,
                                const T* __restrict__ x2, long ld2, const T* __restrict__ lg, long ldl, T* __restrict__ out,
                                long ldo, long pixels, int C) {
  constexpr int VE = DT<T>::VE;
  const int CG = C / VE;
  const int lane = threadIdx.x & 63;
  const long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  for (long px = wave; px < pixels; px += nwaves) {
    // x2 == nullptr: two-level fusion (AsffDoubLevel), the softmax runs over two logits
    float l0 = DT<T>::ld(lg + px * ldl), l1 = DT<T>::ld(lg + px * ldl + 1), l2 = x2 ? DT<T>::ld(lg + px * ldl + 2) : -INFINITY;
    float mx = fmaxf(l0, fmaxf(l1, l2));
    float e0 = expf(l0 - mx), e1 = expf(l1 - mx), e2 = x2 ? expf(l2 - mx) : 0.f;
    float inv = 1.f / (e0 + e1 + e2);
    float w0 = e0 * inv, w1 = e1 * inv, w2 = e2 * inv;
    for (int g = lane; g < CG; g += 64) {
      float a[VE], b[VE], c[VE];
      ldvec<T>(x0 + px * ld0 + g * VE, a);
      ldvec<T>(x1 + px * ld1 + g * VE, b);
      if (x2) ldvec<T>(x2 + px * ld2 + g * VE, c);
      else { for (int e = 0; e < VE; ++e) c[e] = 0.f; }
#pragma unroll
      for (int e = 0; e < VE; ++e) a[e] = a[e] * w0 + b[e] * w1 + c[e] * w2;
      stvec<T>(out + px * ldo + g * VE, a);
    }
  }
}

template <typename T>
__global__ void asff_bwd_kernel(const T* __restrict__ dout, long lddo, const T* __restrict__ x0, long ld0,
                                const T* __restrict__ x1, long ld1, const T* __restrict__ x2, long ld2,
                                const T* __restrict__ lg, long ldl, T* __restrict__ dx0, long ldd0, T* __restrict__ dx1,
                                long ldd1, T* __restrict__ dx2, long ldd2, T* __restrict__ dlg, long lddl, long pixels, int C,
                                int acc0, int acc1, int acc2, int lg_width) {
  constexpr int VE = DT<T>::VE;
  const int CG = C / VE;
  const int lane = threadIdx.x & 63;
  const long wave = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  for (long px = wave; px < pixels; px += nwaves) {
    float l0 = DT<T>::ld(lg + px * ldl), l1 = DT<T>::ld(lg + px * ldl + 1), l2 = x2 ? DT<T>::ld(lg + px * ldl + 2) : -INFINITY;
    float mx = fmaxf(l0, fmaxf(l1, l2));
    float e0 = expf(l0 - mx), e1 = expf(l1 - mx), e2 = x2 ? expf(l2 - mx) : 0.f;
    float inv = 1.f / (e0 + e1 + e2);
    float w0 = e0 * inv, w1 = e1 * inv, w2 = e2 * inv;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
    for (int g = lane; g < CG; g += 64) {
      float go[VE], a[VE], b[VE], c[VE], o[VE];
      ldvec<T>(dout + px * lddo + g * VE, go);
      ldvec<T>(x0 + px * ld0 + g * VE, a);
      ldvec<T>(x1 + px * ld1 + g * VE, b);
      if (x2) ldvec<T>(x2 + px * ld2 + g * VE, c);
      else { for (int e = 0; e < VE; ++e) c[e] = 0.f; }
#pragma unroll
      for (int e = 0; e < VE; ++e) { d0 += go[e] * a[e]; d1 += go[e] * b[e]; d2 += go[e] * c[e]; }
      T* p0 = dx0 + px * ldd0 + g * VE;
      T* p1 = dx1 + px * ldd1 + g * VE;
      T* p2 = dx2 + px * ldd2 + g * VE;
      if (acc0) ldvec<T>(p0, o); else { for (int e = 0; e < VE; ++e) o[e] = 0.f; }
#pragma unroll
      for (int e = 0; e < VE; ++e) o[e] += go[e] * w0;
      stvec<T>(p0, o);
      if (acc1) ldvec<T>(p1, o); else { for (int e = 0; e < VE; ++e) o[e] = 0.f; }
#pragma unroll
      for (int e = 0; e < VE; ++e) o[e] += go[e] * w1;
      stvec<T>(p1, o);
      if (x2) {
        if (acc2) ldvec<T>(p2, o); else { for (int e = 0; e < VE; ++e) o[e] = 0.f; }
#pragma unroll
        for (int e = 0; e < VE; ++e) o[e] += go[e] * w2;
        stvec<T>(p2, o);
      }
    }
    d0 = wave_sum(d0); d1 = wave_sum(d1); d2 = wave_sum(d2);
    if (lane == 0) {
      float dot = w0 * d0 + w1 * d1 + w2 * d2;
      DT<T>::st(dlg + px * lddl, w0 * (d0 - dot));
      DT<T>::st(dlg + px * lddl + 1, w1 * (d1 - dot));
      if (2 < lg_width) DT<T>::st(dlg + px * lddl + 2, w2 * (d2 - dot));        // (0 in the two-level mode: w2 == 0)
      for (int j = 3; j < lg_width; ++j) DT<T>::st(dlg + px * lddl + j, 0.f);
    }
  }
}

int check_view(const char* who, const void* p, long ld, int C, int dtype) {
  const int ve = dtype == DY_F32 ? 4 : 8, es = dtype == DY_F32 ? 4 : 2;
  DY_CHECK(p != nullptr, "%s: null pointer", who);
  DY_CHECK(C > 0 && C % ve == 0, "%s: C=%d must be a multiple of %d", who, C, ve);
  DY_CHECK(ld >= C && (ld * es) % 16 == 0 && ((uintptr_t)p) % 16 == 0, "%s: view not 16-byte aligned (ld=%ld)", who, ld);
  return 0;
}

}  // namespace

#define DISPATCH(dtype, KERNEL, grid, ...)                                              \
  do {                                                                                  \
    if ((dtype) == DY_F32) KERNEL<float><<<grid, 256, 0, (hipStream_t)stream>>>(__VA_ARGS__); \
    else if ((dtype) == DY_F16) KERNEL<f16_t><<<grid, 256, 0, (hipStream_t)stream>>>(__VA_ARGS__);\
    else KERNEL<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>(__VA_ARGS__);                  \
    DY_LAUNCH_CHECK();                                                                  \
  } while (0)

extern "C" int dy_maxpool_fwd(const void* x, int64_t x_ld, void* y, int64_t y_ld, uint8_t* argmax, int N, int H, int W, int C,
                              int k, int stride, int pad, int Ho, int Wo, int dtype, void* stream) {
  if (int e = check_view("dy_maxpool_fwd(x)", x, x_ld, C, dtype)) return e;
  if (int e = check_view("dy_maxpool_fwd(y)", y, y_ld, C, dtype)) return e;
  DY_CHECK(k >= 1 && k <= 15 && stride >= 1 && pad >= 0 && 2 * pad <= k, "dy_maxpool_fwd: bad window");
  DY_CHECK(Ho == (H + 2 * pad - k) / stride + 1 && Wo == (W + 2 * pad - k) / stride + 1, "dy_maxpool_fwd: bad output size");
  const int ve = dtype == DY_F32 ? 4 : 8;
  const int blocks = ew_blocks((long)N * Ho * Wo * (C / ve));
  if (dtype == DY_F32)
    maxpool_fwd_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>((const float*)x, x_ld, (float*)y, y_ld, argmax, N, H, W, C,
                                                                       k, stride, pad, Ho, Wo);
  else if ((dtype) == DY_F16)
    maxpool_fwd_kernel<f16_t><<<blocks, 256, 0, (hipStream_t)stream>>>((const f16_t*)x, x_ld, (f16_t*)y, y_ld, argmax, N, H, W,
                                                                        C, k, stride, pad, Ho, Wo);
  else
    maxpool_fwd_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, x_ld, (bf16_t*)y, y_ld, argmax, N, H, W,
                                                                        C, k, stride, pad, Ho, Wo);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_maxpool_bwd(const void* dy, int64_t dy_ld, const uint8_t* argmax, void* dx, int64_t dx_ld, int N, int H,
                              int W, int C, int k, int stride, int pad, int Ho, int Wo, int accumulate, int dtype,
                              void* stream) {
  if (int e = check_view("dy_maxpool_bwd(dy)", dy, dy_ld, C, dtype)) return e;
  if (int e = check_view("dy_maxpool_bwd(dx)", dx, dx_ld, C, dtype)) return e;
  DY_CHECK(argmax != nullptr, "dy_maxpool_bwd: null argmax");
  const int ve = dtype == DY_F32 ? 4 : 8;
  const int blocks = ew_blocks((long)N * H * W * (C / ve));
  if (dtype == DY_F32)
    maxpool_bwd_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>((const float*)dy, dy_ld, argmax, (float*)dx, dx_ld, N, H, W,
                                                                       C, k, stride, pad, Ho, Wo, accumulate);
  else if ((dtype) == DY_F16)
    maxpool_bwd_kernel<f16_t><<<blocks, 256, 0, (hipStream_t)stream>>>((const f16_t*)dy, dy_ld, argmax, (f16_t*)dx, dx_ld, N, H,
                                                                        W, C, k, stride, pad, Ho, Wo, accumulate);
  else
    maxpool_bwd_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)dy, dy_ld, argmax, (bf16_t*)dx, dx_ld, N, H,
                                                                        W, C, k, stride, pad, Ho, Wo, accumulate);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_upsample_nearest_fwd(const void* x, int64_t x_ld, void* y, int64_t y_ld, int N, int H, int W, int C, int scale,
                                       int dtype, void* stream) {
  if (int e = check_view("dy_upsample_nearest_fwd(x)", x, x_ld, C, dtype)) return e;
  if (int e = check_view("dy_upsample_nearest_fwd(y)", y, y_ld, C, dtype)) return e;
  DY_CHECK(scale >= 1, "dy_upsample_nearest_fwd: bad scale");
  const int ve = dtype == DY_F32 ? 4 : 8;
  const int blocks = ew_blocks((long)N * H * scale * W * scale * (C / ve));
  if (dtype == DY_F32)
    upsample_fwd_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>((const float*)x, x_ld, (float*)y, y_ld, N, H, W, C, scale);
  else if ((dtype) == DY_F16)
    upsample_fwd_kernel<f16_t><<<blocks, 256, 0, (hipStream_t)stream>>>((const f16_t*)x, x_ld, (f16_t*)y, y_ld, N, H, W, C, scale);
  else
    upsample_fwd_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, x_ld, (bf16_t*)y, y_ld, N, H, W, C, scale);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_upsample_nearest_bwd(const void* dy, int64_t dy_ld, void* dx, int64_t dx_ld, int N, int H, int W, int C,
                                       int scale, int accumulate, int dtype, void* stream) {
  if (int e = check_view("dy_upsample_nearest_bwd(dy)", dy, dy_ld, C, dtype)) return e;
  if (int e = check_view("dy_upsample_nearest_bwd(dx)", dx, dx_ld, C, dtype)) return e;
  const int ve = dtype == DY_F32 ? 4 : 8;
  const int blocks = ew_blocks((long)N * H * W * (C / ve));
  if (dtype == DY_F32)
    upsample_bwd_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>((const float*)dy, dy_ld, (float*)dx, dx_ld, N, H, W, C, scale,
                                                                        accumulate);
  else if ((dtype) == DY_F16)
    upsample_bwd_kernel<f16_t><<<blocks, 256, 0, (hipStream_t)stream>>>((const f16_t*)dy, dy_ld, (f16_t*)dx, dx_ld, N, H, W, C,
                                                                         scale, accumulate);
  else
    upsample_bwd_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)dy, dy_ld, (bf16_t*)dx, dx_ld, N, H, W, C,
                                                                         scale, accumulate);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_copy2d(const void* src, int64_t src_ld, void* dst, int64_t dst_ld, int64_t pixels, int C, int accumulate,
                         int dtype, void* stream) {
  if (int e = check_view("dy_copy2d(src)", src, src_ld, C, dtype)) return e;
  if (int e = check_view("dy_copy2d(dst)", dst, dst_ld, C, dtype)) return e;
  const int ve = dtype == DY_F32 ? 4 : 8;
  const int blocks = ew_blocks(pixels * (C / ve));
  if (dtype == DY_F32)
    copy2d_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>((const float*)src, src_ld, (float*)dst, dst_ld, pixels, C, accumulate);
  else if ((dtype) == DY_F16)
    copy2d_kernel<f16_t><<<blocks, 256, 0, (hipStream_t)stream>>>((const f16_t*)src, src_ld, (f16_t*)dst, dst_ld, pixels, C,
                                                                   accumulate);
  else
    copy2d_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)src, src_ld, (bf16_t*)dst, dst_ld, pixels, C,
                                                                   accumulate);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream) {
  DY_CHECK(src && dst && n >= 0, "dy_cast: bad args");
  const int blocks = ew_blocks(n);
  hipStream_t st = (hipStream_t)stream;
  DY_CHECK(src_dtype == dst_dtype || src_dtype == DY_F32 || dst_dtype == DY_F32, "dy_cast: bf16 <-> f16 goes through f32");
  if (src_dtype == DY_F32 && dst_dtype == DY_BF16) cast_kernel<float, bf16_t><<<blocks, 256, 0, st>>>((const float*)src, (bf16_t*)dst, n);
  else if (src_dtype == DY_BF16 && dst_dtype == DY_F32) cast_kernel<bf16_t, float><<<blocks, 256, 0, st>>>((const bf16_t*)src, (float*)dst, n);
  else if (src_dtype == DY_F32 && dst_dtype == DY_F16) cast_kernel<float, f16_t><<<blocks, 256, 0, st>>>((const float*)src, (f16_t*)dst, n);
  else if (src_dtype == DY_F16 && dst_dtype == DY_F32) cast_kernel<f16_t, float><<<blocks, 256, 0, st>>>((const f16_t*)src, (float*)dst, n);
  else if (src_dtype == DY_F32 && dst_dtype == DY_F32) cast_kernel<float, float><<<blocks, 256, 0, st>>>((const float*)src, (float*)dst, n);
  else cast_kernel<bf16_t, bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)src, (bf16_t*)dst, n);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_asff_fuse_fwd(const void* x0, int64_t ld0, const void* x1, int64_t ld1, const void* x2, int64_t ld2,
                                const void* logits, int64_t ldl, void* out, int64_t ldo, int64_t pixels, int C, int dtype,
                                void* stream) {
  if (int e = check_view("dy_asff_fuse_fwd(x0)", x0, ld0, C, dtype)) return e;
  if (int e = check_view("dy_asff_fuse_fwd(x1)", x1, ld1, C, dtype)) return e;
  if (x2)
    if (int e = check_view("dy_asff_fuse_fwd(x2)", x2, ld2, C, dtype)) return e;
  if (int e = check_view("dy_asff_fuse_fwd(out)", out, ldo, C, dtype)) return e;
  DY_CHECK(logits && ldl >= (x2 ? 3 : 2), "dy_asff_fuse_fwd: bad logits");
  long waves = pixels;
  int blocks = (int)((waves + 3) / 4);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (dtype == DY_F32)
    asff_fwd_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>((const float*)x0, ld0, (const float*)x1, ld1, (const float*)x2, ld2,
                                                                    (const float*)logits, ldl, (float*)out, ldo, pixels, C);
  else if ((dtype) == DY_F16)
    asff_fwd_kernel<f16_t><<<blocks, 256, 0, (hipStream_t)stream>>>((const f16_t*)x0, ld0, (const f16_t*)x1, ld1,
                                                                     (const f16_t*)x2, ld2, (const f16_t*)logits, ldl,
                                                                     (f16_t*)out, ldo, pixels, C);
  else
    asff_fwd_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>((const bf16_t*)x0, ld0, (const bf16_t*)x1, ld1,
                                                                     (const bf16_t*)x2, ld2, (const bf16_t*)logits, ldl,
                                                                     (bf16_t*)out, ldo, pixels, C);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_asff_fuse_bwd(const void* dout, int64_t lddo, const void* x0, int64_t ld0, const void* x1, int64_t ld1,
                                const void* x2, int64_t ld2, const void* logits, int64_t ldl, void* dx0, int64_t ldd0, void* dx1,
                                int64_t ldd1, void* dx2, int64_t ldd2, void* dlogits, int64_t lddl, int64_t pixels, int C,
                                int acc0, int acc1, int acc2, int dtype, void* stream) {
  if (int e = check_view("dy_asff_fuse_bwd(dout)", dout, lddo, C, dtype)) return e;
  if (int e = check_view("dy_asff_fuse_bwd(x0)", x0, ld0, C, dtype)) return e;
  if (int e = check_view("dy_asff_fuse_bwd(x1)", x1, ld1, C, dtype)) return e;
  if (x2)
    if (int e = check_view("dy_asff_fuse_bwd(x2)", x2, ld2, C, dtype)) return e;
  if (int e = check_view("dy_asff_fuse_bwd(dx0)", dx0, ldd0, C, dtype)) return e;
  if (int e = check_view("dy_asff_fuse_bwd(dx1)", dx1, ldd1, C, dtype)) return e;
  if (x2)
    if (int e = check_view("dy_asff_fuse_bwd(dx2)", dx2, ldd2, C, dtype)) return e;
  DY_CHECK(logits && dlogits && ldl >= (x2 ? 3 : 2) && lddl >= (x2 ? 3 : 2), "dy_asff_fuse_bwd: bad logits");
  int blocks = (int)((pixels + 3) / 4);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  const int lgw = (int)(lddl < 8 ? lddl : 8);
  if (dtype == DY_F32)
    asff_bwd_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>(
        (const float*)dout, lddo, (const float*)x0, ld0, (const float*)x1, ld1, (const float*)x2, ld2, (const float*)logits, ldl,
        (float*)dx0, ldd0, (float*)dx1, ldd1, (float*)dx2, ldd2, (float*)dlogits, lddl, pixels, C, acc0, acc1, acc2, lgw);
  else if ((dtype) == DY_F16)
    asff_bwd_kernel<f16_t><<<blocks, 256, 0, (hipStream_t)stream>>>(
        (const f16_t*)dout, lddo, (const f16_t*)x0, ld0, (const f16_t*)x1, ld1, (const f16_t*)x2, ld2, (const f16_t*)logits,
        ldl, (f16_t*)dx0, ldd0, (f16_t*)dx1, ldd1, (f16_t*)dx2, ldd2, (f16_t*)dlogits, lddl, pixels, C, acc0, acc1, acc2, lgw);
  else
    asff_bwd_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>(
        (const bf16_t*)dout, lddo, (const bf16_t*)x0, ld0, (const bf16_t*)x1, ld1, (const bf16_t*)x2, ld2, (const bf16_t*)logits,
        ldl, (bf16_t*)dx0, ldd0, (bf16_t*)dx1, ldd1, (bf16_t*)dx2, ldd2, (bf16_t*)dlogits, lddl, pixels, C, acc0, acc1, acc2, lgw);
  DY_LAUNCH_CHECK();
  return 0;
}
