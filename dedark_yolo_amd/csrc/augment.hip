// Device-side input pipeline: the pixel work of the reference's host augmentation chain.
//
// Reference, per sample on dataloader worker CPUs: ultralytics/data/base.py:142-169 load_image (cv2.resize), ultralytics/data/augment.py
// Mosaic._mosaic4 :158-195 (numpy canvas of four images) -> RandomPerspective :323-345 (cv2.warpAffine, border 114) -> RandomHSV :486-499
// (cv2.cvtColor BGR<->HSV + three cv2.LUT) -> RandomFlip x2 :527-532 -> Format._format_img :745-751 (HWC BGR -> CHW RGB); validation:
// LetterBox :559-591 (cv2.resize + cv2.copyMakeBorder); trainer: DarkChannel / AtmLight / DarkIcA (models/yolo/detect/train.py:42-68,
// a Python loop per image behind a device->host copy).
//
// Here: dy_aug_mosaic_warp renders a whole batch in ONE launch -- a thread per output pixel walks flips -> warpAffine's fixed-point
// inverse map (AB_BITS 10, INTER_BITS 5, 15-bit bilinear weights) -> the four mosaic placement rectangles (the 2s x 2s canvas is
// never materialised: each tap is looked up in the source image that owns that canvas pixel, else grey 114) -> cv2's 8-bit BGR->HSV,
// the three lookup tables, HSV->BGR -> planar RGB.  All arithmetic is integer / explicitly rounded f32, bit-identical to
// oracle/augment.py (OpenCV itself is absent from the image: its algorithms are restated from the published source, parity unpinned).
#include "dy_common.h"
#include "../../include/dedark_yolo.h"

// HIP's __fmul_rn / __fsub_rn / __dmul_rn are plain operators on AMD targets: they document the intent; what keeps hipcc from contracting
// `1 - s * h` or `m1 * y + m2` into a fused multiply-add (one rounding instead of two: off by one level at exact .5 ties) is
// -ffp-contract=off for this file (csrc/Makefile)

namespace {

__device__ inline int rint_i(double v) { return (int)rint(v); }          // saturate_cast<int>(double) = cvRound (ties to even)

// ---- cv::resize, INTER_LINEAR, 8-bit (resize.cpp: 11-bit coefficients, the >>4 / >>16 / +2 >>2 vertical pass) ---------------------
struct ResizeAxis {
  int s0, s1, a0, a1;
};
__device__ inline ResizeAxis resize_axis(int d, int src_n, int dst_n, bool zero_frac_at_edges) {
  const double scale = (double)src_n / (double)dst_n;
  float f = (float)__dsub_rn(__dmul_rn((double)d + 0.5, scale), 0.5);          // (no fused multiply-add: one rounding per operation)
  int s = (int)floorf(f);
  f = __fsub_rn(f, (float)s);
  if (zero_frac_at_edges && (s < 0 || s >= src_n - 1)) f = 0.f;          // columns (resize.cpp: fx = 0 when the tap pair is clamped)
  ResizeAxis r;
  r.s0 = min(max(s, 0), src_n - 1);
  r.s1 = min(max(s + 1, 0), src_n - 1);
  r.a1 = (int)rintf(__fmul_rn(f, 2048.f));
  r.a0 = (int)rintf(__fmul_rn(__fsub_rn(1.f, f), 2048.f));
  return r;
}
__device__ inline int resize_px(const uint8_t* src, long pitch, const ResizeAxis& ax, const ResizeAxis& ay, int c) {
  const uint8_t* r0 = src + (long)ay.s0 * pitch;
  const uint8_t* r1 = src + (long)ay.s1 * pitch;
  const int h0 = r0[ax.s0 * 3 + c] * ax.a0 + r0[ax.s1 * 3 + c] * ax.a1;
  const int h1 = r1[ax.s0 * 3 + c] * ax.a0 + r1[ax.s1 * 3 + c] * ax.a1;
  const int v = (((ay.a0 * (h0 >> 4)) >> 16) + ((ay.a1 * (h1 >> 4)) >> 16) + 2) >> 2;
  return min(max(v, 0), 255);
}

__global__ __launch_bounds__(256) void resize_kernel(const uint8_t* __restrict__ src, int sh, int sw, long spitch, uint8_t* __restrict__ dst,
                                                     int dh, int dw, long dpitch) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh) return;
  const ResizeAxis ax = resize_axis(x, sw, dw, true), ay = resize_axis(y, sh, dh, false);
#pragma unroll
  for (int c = 0; c < 3; ++c) dst[(long)y * dpitch + x * 3 + c] = (uint8_t)resize_px(src, spitch, ax, ay, c);
}

// LetterBox + Format: resize to (nh, nw), constant border 114 around it, HWC BGR -> CHW RGB
__global__ __launch_bounds__(256) void letterbox_kernel(const uint8_t* __restrict__ src, int sh, int sw, long spitch, int nh, int nw, int top,
                                                        int left, int oh, int ow, uint8_t* __restrict__ out) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= ow || y >= oh) return;
  const int rx = x - left, ry = y - top;
  int bgr[3] = {114, 114, 114};
  if (rx >= 0 && rx < nw && ry >= 0 && ry < nh) {
    const ResizeAxis ax = resize_axis(rx, sw, nw, true), ay = resize_axis(ry, sh, nh, false);
#pragma unroll
    for (int c = 0; c < 3; ++c) bgr[c] = resize_px(src, spitch, ax, ay, c);
  }
  const long plane = (long)oh * ow, o = (long)y * ow + x;
  out[o] = (uint8_t)bgr[2];
  out[plane + o] = (uint8_t)bgr[1];
  out[2 * plane + o] = (uint8_t)bgr[0];
}

// ---- cv::cvtColor 8-bit BGR <-> HSV (color_hsv.cpp: RGB2HSV_b's 12-bit division tables, HSV2RGB_b's float path) ---------------------
__device__ inline void bgr2hsv(int b, int g, int r, const int* sdiv, const int* hdiv, int& h, int& s, int& v) {
  v = max(max(b, g), r);
  const int vmin = min(min(b, g), r);
  const int diff = v - vmin;
  s = (diff * sdiv[v] + (1 << 11)) >> 12;
  int hh = (v == r) ? (g - b) : (v == g) ? (b - r + 2 * diff) : (r - g + 4 * diff);
  hh = (hh * hdiv[diff] + (1 << 11)) >> 12;
  h = hh + (hh < 0 ? 180 : 0);
}
__device__ inline void hsv2bgr(int H, int S, int V, int& b, int& g, int& r) {
  float h = __fmul_rn((float)H, (float)(6.0 / 180.0));
  const float s = __fmul_rn((float)S, (float)(1.0 / 255.0)), v = __fmul_rn((float)V, (float)(1.0 / 255.0));
  float fb, fg, fr;
  if (S == 0) {
    fb = fg = fr = v;
  } else {
    if (h >= 6.f) h = __fsub_rn(h, 6.f);
    int sector = (int)floorf(h);
    h = __fsub_rn(h, (float)sector);
    if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
    float tab[4];
    tab[0] = v;
    tab[1] = __fmul_rn(v, __fsub_rn(1.f, s));
    tab[2] = __fmul_rn(v, __fsub_rn(1.f, __fmul_rn(s, h)));
    tab[3] = __fmul_rn(v, __fsub_rn(1.f, __fmul_rn(s, __fsub_rn(1.f, h))));
    const int sb[6] = {1, 1, 3, 0, 0, 2}, sg[6] = {3, 0, 0, 2, 1, 1}, sr[6] = {0, 2, 1, 1, 3, 0};
    fb = tab[sb[sector]]; fg = tab[sg[sector]]; fr = tab[sr[sector]];
  }
  b = min(max((int)rintf(__fmul_rn(fb, 255.f)), 0), 255);
  g = min(max((int)rintf(__fmul_rn(fg, 255.f)), 0), 255);
  r = min(max((int)rintf(__fmul_rn(fr, 255.f)), 0), 255);
}

// ---- mosaic canvas lookup + cv::warpAffine (imgwarp.cpp) + HSV gains + flips + planar RGB ---------------------------------------
__device__ inline void canvas_px(const dy_aug_sample& a, int cx, int cy, int* bgr) {
  bgr[0] = bgr[1] = bgr[2] = 114;
  if (cx < 0 || cy < 0 || cx >= a.canvas_w || cy >= a.canvas_h) return;
  for (int i = a.n_src - 1; i >= 0; --i) {                   // later images were pasted over earlier ones (augment.py:188)
    const int* r = a.rect[i];
    if (cx >= r[0] && cx < r[2] && cy >= r[1] && cy < r[3]) {
      const uint8_t* p = a.src[i] + (long)(cy - r[1] + r[5]) * a.pitch[i] + (long)(cx - r[0] + r[4]) * 3;
      bgr[0] = p[0]; bgr[1] = p[1]; bgr[2] = p[2];
      return;
    }
  }
}

__global__ __launch_bounds__(256) void mosaic_warp_kernel(const dy_aug_sample* __restrict__ samples, int oh, int ow, uint8_t* __restrict__ out) {
  __shared__ int sdiv[256], hdiv[256];
  __shared__ uint8_t lut[3][256];
  const dy_aug_sample& a = samples[blockIdx.z];
  {
    const int i = threadIdx.x;
    sdiv[i] = i ? rint_i((double)(255 << 12) / (1.0 * i)) : 0;
    hdiv[i] = i ? rint_i((double)(180 << 12) / (6.0 * i)) : 0;
    lut[0][i] = a.lut[0][i]; lut[1][i] = a.lut[1][i]; lut[2][i] = a.lut[2][i];
  }
  __syncthreads();
  const int ox = blockIdx.x * 64 + (threadIdx.x & 63), oy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (ox >= ow || oy >= oh) return;
  const int wx = a.fliplr ? ow - 1 - ox : ox, wy = a.flipud ? oh - 1 - oy : oy;        // np.flipud then np.fliplr of the warped image
  // inverse map in fixed point: X = (round(m0 x 1024) + round((m1 y + m2) 1024) + 16) >> 5
  const int X = (rint_i(__dmul_rn(__dmul_rn(a.minv[0], (double)wx), 1024.0)) +
                 rint_i(__dmul_rn(__dadd_rn(__dmul_rn(a.minv[1], (double)wy), a.minv[2]), 1024.0)) + 16) >> 5;
  const int Y = (rint_i(__dmul_rn(__dmul_rn(a.minv[3], (double)wx), 1024.0)) +
                 rint_i(__dmul_rn(__dadd_rn(__dmul_rn(a.minv[4], (double)wy), a.minv[5]), 1024.0)) + 16) >> 5;
  const int sx = min(max(X >> 5, -32768), 32767), sy = min(max(Y >> 5, -32768), 32767);
  const int fx = X & 31, fy = Y & 31;
  const int w00 = 32 * (32 - fx) * (32 - fy), w01 = 32 * fx * (32 - fy), w10 = 32 * (32 - fx) * fy, w11 = 32 * fx * fy;
  int p00[3], p01[3], p10[3], p11[3];
  canvas_px(a, sx, sy, p00);
  canvas_px(a, sx + 1, sy, p01);
  canvas_px(a, sx, sy + 1, p10);
  canvas_px(a, sx + 1, sy + 1, p11);
  int bgr[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int v = (p00[c] * w00 + p01[c] * w01 + p10[c] * w10 + p11[c] * w11 + (1 << 14)) >> 15;
    bgr[c] = min(max(v, 0), 255);
  }
  if (a.hsv) {
    int h, s, v;
    bgr2hsv(bgr[0], bgr[1], bgr[2], sdiv, hdiv, h, s, v);
    hsv2bgr(lut[0][h], lut[1][s], lut[2][v], bgr[0], bgr[1], bgr[2]);
  }
  const long plane = (long)oh * ow;
  uint8_t* o = out + (long)blockIdx.z * 3 * plane + (long)oy * ow + ox;
  o[0] = (uint8_t)bgr[2];
  o[plane] = (uint8_t)bgr[1];
  o[2 * plane] = (uint8_t)bgr[0];
}

// ---- dark-channel prior of the trainer (models/yolo/detect/train.py:42-68), deterministic -----------------------------------------------
// One block per image.  u8 = (uint8)(x * 255) per channel (train.py:81), dark = channel minimum, numpx = max(HW / 1000, 1) brightest-dark
// pixels with ties broken by pixel index (stable ascending order: the later pixel ranks higher), the first of them left out like the
// reference's `range(1, numpx)`, A = sum / numpx; IcA rows 0..2 as the reference writes them (row r divided by A[r]), rows >= 3 by the
// per-channel formula (oracle/augment.py:dark_ica).
constexpr int DCP_T = 1024;

__device__ inline int px_u8(const float* p) { return (int)(__fmul_rn(*p, 255.f)) & 255; }

__global__ __launch_bounds__(DCP_T) void dark_prior_kernel(const float* __restrict__ img, int H, int W, float* __restrict__ A_out,
                                                          float* __restrict__ ica) {
  __shared__ int hist[256];
  __shared__ int s_t, s_need, s_base;
  __shared__ unsigned long long s_sum[3];
  __shared__ int s_scan[DCP_T];
  __shared__ double s_A[3];
  const int b = blockIdx.x, tid = threadIdx.x;
  const long HW = (long)H * W;
  const float* p0 = img + (long)b * 3 * HW;
  const int numpx = max((int)(HW / 1000), 1);
  if (tid < 256) hist[tid] = 0;
  if (tid < 3) s_sum[tid] = 0ull;
  __syncthreads();
  for (long i = tid; i < HW; i += DCP_T) {
    const int d = min(min(px_u8(p0 + i), px_u8(p0 + HW + i)), px_u8(p0 + 2 * HW + i));
    atomicAdd(&hist[d], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int cum = 0, t = 255;
    for (; t > 0; --t) {
      if (cum + hist[t] >= numpx) break;
      cum += hist[t];
    }
    s_t = t;
    s_need = numpx - cum;          // pixels of value t that belong to the top set (>= 1)
    s_base = 0;
  }
  __syncthreads();
  const int t = s_t, need = s_need;
  // all pixels brighter than t
  unsigned long long loc[3] = {0ull, 0ull, 0ull};
  for (long i = tid; i < HW; i += DCP_T) {
    const int r = px_u8(p0 + i), g = px_u8(p0 + HW + i), bl = px_u8(p0 + 2 * HW + i);
    if (min(min(r, g), bl) > t) { loc[0] += r; loc[1] += g; loc[2] += bl; }
  }
  // pixels of value t, taken from the END of the image (highest indices first); rank need-1 is the one the reference's loop skips
  for (long hi = HW; hi > 0 && s_base < need; hi -= DCP_T) {
    const long i = hi - 1 - tid;                      // thread 0 looks at the last pixel of the window
    int r = 0, g = 0, bl = 0, flag = 0;
    if (i >= 0) {
      r = px_u8(p0 + i); g = px_u8(p0 + HW + i); bl = px_u8(p0 + 2 * HW + i);
      flag = min(min(r, g), bl) == t;
    }
    s_scan[tid] = flag;
    __syncthreads();
    for (int off = 1; off < DCP_T; off <<= 1) {       // inclusive scan (Hillis-Steele)
      const int v = tid >= off ? s_scan[tid - off] : 0;
      __syncthreads();
      s_scan[tid] += v;
      __syncthreads();
    }
    const int rank = s_base + s_scan[tid] - flag;     // t-valued pixels after this one
    if (flag && rank < need - 1) { loc[0] += r; loc[1] += g; loc[2] += bl; }
    __syncthreads();
    if (tid == DCP_T - 1) s_base += s_scan[tid];
    __syncthreads();
  }
  atomicAdd(&s_sum[0], loc[0]);
  atomicAdd(&s_sum[1], loc[1]);
  atomicAdd(&s_sum[2], loc[2]);
  __syncthreads();
  if (tid < 3) {
    s_A[tid] = (double)s_sum[tid] / (double)numpx;
    A_out[b * 3 + tid] = (float)s_A[tid];
  }
  __syncthreads();
  const double A0 = s_A[0], A1 = s_A[1], A2 = s_A[2];
  for (long i = tid; i < HW; i += DCP_T) {
    const int y = (int)(i / W);
    const int c0 = px_u8(p0 + i), c1 = px_u8(p0 + HW + i), c2 = px_u8(p0 + 2 * HW + i);
    double d0, d1, d2;
    if (y < 3) {
      const double Ar = y == 0 ? A0 : y == 1 ? A1 : A2;
      d0 = c0 / Ar; d1 = c1 / Ar; d2 = c2 / Ar;
    } else {
      d0 = c0 / A0; d1 = c1 / A1; d2 = c2 / A2;
    }
    auto u8 = [](double q) { return isfinite(q) ? (int)((long long)trunc(q) & 255) : 0; };
    ica[(long)b * HW + i] = (float)min(min(u8(d0), u8(d1)), u8(d2));
  }
}

}  // namespace

extern "C" int dy_aug_resize_u8(const uint8_t* src, int sh, int sw, int64_t src_pitch, uint8_t* dst, int dh, int dw, int64_t dst_pitch,
                                void* stream) {
  DY_CHECK(src && dst && sh > 0 && sw > 0 && dh > 0 && dw > 0 && src_pitch >= 3L * sw && dst_pitch >= 3L * dw, "dy_aug_resize_u8: bad arguments");
  resize_kernel<<<dim3(dy_cdiv(dw, 64), dy_cdiv(dh, 4)), 256, 0, (hipStream_t)stream>>>(src, sh, sw, src_pitch, dst, dh, dw, dst_pitch);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_aug_letterbox(const uint8_t* src, int sh, int sw, int64_t src_pitch, int new_h, int new_w, int top, int left, int out_h,
                                int out_w, uint8_t* out, void* stream) {
  DY_CHECK(src && out && sh > 0 && sw > 0 && new_h > 0 && new_w > 0 && out_h > 0 && out_w > 0, "dy_aug_letterbox: bad arguments");
  DY_CHECK(top >= 0 && left >= 0 && top + new_h <= out_h && left + new_w <= out_w, "dy_aug_letterbox: the resized image does not fit");
  letterbox_kernel<<<dim3(dy_cdiv(out_w, 64), dy_cdiv(out_h, 4)), 256, 0, (hipStream_t)stream>>>(src, sh, sw, src_pitch, new_h, new_w, top, left,
                                                                                               out_h, out_w, out);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_aug_mosaic_warp(const dy_aug_sample* samples, int B, int out_h, int out_w, uint8_t* out, void* stream) {
  DY_CHECK(B >= 0 && out_h > 0 && out_w > 0 && (B == 0 || (samples && out)), "dy_aug_mosaic_warp: bad arguments");
  DY_CHECK(B <= 65535, "dy_aug_mosaic_warp: at most 65535 samples per launch");
  if (B == 0) return 0;
  mosaic_warp_kernel<<<dim3(dy_cdiv(out_w, 64), dy_cdiv(out_h, 4), B), 256, 0, (hipStream_t)stream>>>(samples, out_h, out_w, out);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_dark_channel_prior(const float* img, int B, int H, int W, float* A, float* ica, void* stream) {
  DY_CHECK(B >= 0 && H > 0 && W > 0 && (B == 0 || (img && A && ica)), "dy_dark_channel_prior: bad arguments");
  if (B == 0) return 0;
  dark_prior_kernel<<<B, DCP_T, 0, (hipStream_t)stream>>>(img, H, W, A, ica);
  DY_LAUNCH_CHECK();
  return 0;
}
