// Host/device math shared by the loss kernels and the CPU-side math harness (tests/test_host_math.py).
// CIoU follows bbox_iou(xywh=False, CIoU=True) of the reference (ultralytics/utils/metrics.py:75-128):
// eps is added to h only, alpha is a constant in the backward (computed under no_grad).
#pragma once
#include <math.h>

#ifdef __HIPCC__
#define DY_HD __host__ __device__
#else
#define DY_HD
#endif

DY_HD inline float dy_fmaxf(float a, float b) { return a > b ? a : b; }
DY_HD inline float dy_fminf(float a, float b) { return a < b ? a : b; }

// CIoU(b1, b2), boxes xyxy
DY_HD inline float dy_ciou(const float* b1, const float* b2) {
  const float eps = 1e-7f;
  float x1 = b1[0], y1 = b1[1], x2 = b1[2], y2 = b1[3];
  float X1 = b2[0], Y1 = b2[1], X2 = b2[2], Y2 = b2[3];
  float w1 = x2 - x1, h1 = y2 - y1 + eps, w2 = X2 - X1, h2 = Y2 - Y1 + eps;
  float iw = dy_fminf(x2, X2) - dy_fmaxf(x1, X1), ih = dy_fminf(y2, Y2) - dy_fmaxf(y1, Y1);
  float inter = dy_fmaxf(iw, 0.f) * dy_fmaxf(ih, 0.f);
  float uni = w1 * h1 + w2 * h2 - inter + eps;
  float iou = inter / uni;
  float cw = dy_fmaxf(x2, X2) - dy_fminf(x1, X1), ch = dy_fmaxf(y2, Y2) - dy_fminf(y1, Y1);
  float c2 = cw * cw + ch * ch + eps;
  float sx = X1 + X2 - x1 - x2, sy = Y1 + Y2 - y1 - y2;
  float rho2 = (sx * sx + sy * sy) / 4.f;
  float da = atanf(w2 / h2) - atanf(w1 / h1);
  float v = 0.40528473456935116f * da * da;          // 4/pi^2
  float alpha = v / (v - iou + (1.f + eps));
  return iou - (rho2 / c2 + v * alpha);
}

// gradient weight of max(a,b) wrt a (ties split in half, as torch.maximum's backward does)
DY_HD inline float dy_gmax(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }
DY_HD inline float dy_gmin(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }

// returns CIoU(b1,b2) and g[k] = d CIoU / d b1[k]
DY_HD inline float dy_ciou_grad(const float* b1, const float* b2, float* g) {
  const float eps = 1e-7f;
  float x1 = b1[0], y1 = b1[1], x2 = b1[2], y2 = b1[3];
  float X1 = b2[0], Y1 = b2[1], X2 = b2[2], Y2 = b2[3];
  float w1 = x2 - x1, h1 = y2 - y1 + eps, w2 = X2 - X1, h2 = Y2 - Y1 + eps;
  float iw = dy_fminf(x2, X2) - dy_fmaxf(x1, X1), ih = dy_fminf(y2, Y2) - dy_fmaxf(y1, Y1);
  float iwc = dy_fmaxf(iw, 0.f), ihc = dy_fmaxf(ih, 0.f);
  float inter = iwc * ihc;
  float uni = w1 * h1 + w2 * h2 - inter + eps;
  float iou = inter / uni;
  float cw = dy_fmaxf(x2, X2) - dy_fminf(x1, X1), ch = dy_fmaxf(y2, Y2) - dy_fminf(y1, Y1);
  float c2 = cw * cw + ch * ch + eps;
  float sx = X1 + X2 - x1 - x2, sy = Y1 + Y2 - y1 - y2;
  float rho2 = (sx * sx + sy * sy) / 4.f;
  float at1 = atanf(w1 / h1);
  float da = atanf(w2 / h2) - at1;
  float v = 0.40528473456935116f * da * da;
  float alpha = v / (v - iou + (1.f + eps));
  // per-coordinate derivatives, order (x1, y1, x2, y2)
  float diw[4] = {iw >= 0.f ? -dy_gmax(x1, X1) : 0.f, 0.f, iw >= 0.f ? dy_gmin(x2, X2) : 0.f, 0.f};
  float dih[4] = {0.f, ih >= 0.f ? -dy_gmax(y1, Y1) : 0.f, 0.f, ih >= 0.f ? dy_gmin(y2, Y2) : 0.f};
  const float dw1[4] = {-1.f, 0.f, 1.f, 0.f}, dh1[4] = {0.f, -1.f, 0.f, 1.f};
  float dcw[4] = {-dy_gmin(x1, X1), 0.f, dy_gmax(x2, X2), 0.f};
  float dch[4] = {0.f, -dy_gmin(y1, Y1), 0.f, dy_gmax(y2, Y2)};
  const float drho[4] = {-sx / 2.f, -sy / 2.f, -sx / 2.f, -sy / 2.f};
  float den = w1 * w1 + h1 * h1;
  for (int k = 0; k < 4; ++k) {
    float dinter = diw[k] * ihc + iwc * dih[k];
    float duni = dw1[k] * h1 + w1 * dh1[k] - dinter;
    float diou = (dinter * uni - inter * duni) / (uni * uni);
    float dc2 = 2.f * cw * dcw[k] + 2.f * ch * dch[k];
    float dpen = (drho[k] * c2 - rho2 * dc2) / (c2 * c2);
    float dat1 = (h1 * dw1[k] - w1 * dh1[k]) / den;
    float dv = 0.40528473456935116f * 2.f * da * (-dat1);
    g[k] = diou - dpen - alpha * dv;
  }
  return iou - (rho2 / c2 + v * alpha);
}

// Every mode of the reference's bbox_iou (ultralytics/utils/metrics.py:75-128): kind 0 IoU, 1 GIoU, 2 DIoU, 3 CIoU; xywh != 0: boxes are
// (cx, cy, w, h) and w, h enter the areas / aspect term as given (no eps), otherwise (x1, y1, x2, y2) with eps added to h only.
// g (nullable) = d value / d box1 in box1's OWN coordinates; CIoU's alpha is a constant in the backward (no_grad there).
DY_HD inline float dy_box_iou_any(const float* b1, const float* b2, int xywh, int kind, float eps, float* g) {
  float x1, y1, x2, y2, X1, Y1, X2, Y2, w1, h1, w2, h2;
  if (xywh) {
    w1 = b1[2]; h1 = b1[3]; w2 = b2[2]; h2 = b2[3];
    const float hw1 = w1 / 2.f, hh1 = h1 / 2.f, hw2 = w2 / 2.f, hh2 = h2 / 2.f;
    x1 = b1[0] - hw1; x2 = b1[0] + hw1; y1 = b1[1] - hh1; y2 = b1[1] + hh1;
    X1 = b2[0] - hw2; X2 = b2[0] + hw2; Y1 = b2[1] - hh2; Y2 = b2[1] + hh2;
  } else {
    x1 = b1[0]; y1 = b1[1]; x2 = b1[2]; y2 = b1[3];
    X1 = b2[0]; Y1 = b2[1]; X2 = b2[2]; Y2 = b2[3];
    w1 = x2 - x1; h1 = y2 - y1 + eps; w2 = X2 - X1; h2 = Y2 - Y1 + eps;
  }
  const float iw = dy_fminf(x2, X2) - dy_fmaxf(x1, X1), ih = dy_fminf(y2, Y2) - dy_fmaxf(y1, Y1);
  const float iwc = dy_fmaxf(iw, 0.f), ihc = dy_fmaxf(ih, 0.f);
  const float inter = iwc * ihc;
  const float uni = w1 * h1 + w2 * h2 - inter + eps;
  const float iou = inter / uni;
  const float cw = dy_fmaxf(x2, X2) - dy_fminf(x1, X1), ch = dy_fmaxf(y2, Y2) - dy_fminf(y1, Y1);
  const float c2 = cw * cw + ch * ch + eps, c_area = cw * ch + eps;
  const float sx = X1 + X2 - x1 - x2, sy = Y1 + Y2 - y1 - y2;
  const float rho2 = (sx * sx + sy * sy) / 4.f;
  float v = 0.f, alpha = 0.f, da = 0.f;
  if (kind == 3) {
    da = atanf(w2 / h2) - atanf(w1 / h1);
    v = 0.40528473456935116f * da * da;          // 4/pi^2
    alpha = v / (v - iou + (1.f + eps));
  }
  const float val = kind == 0 ? iou : kind == 1 ? iou - (c_area - uni) / c_area : kind == 2 ? iou - rho2 / c2 : iou - (rho2 / c2 + v * alpha);
  if (!g) return val;
  // partial derivatives wrt the corners (x1, y1, x2, y2) with w1, h1 held fixed, and wrt w1, h1 with the corners held fixed
  const float diw[4] = {iw >= 0.f ? -dy_gmax(x1, X1) : 0.f, 0.f, iw >= 0.f ? dy_gmin(x2, X2) : 0.f, 0.f};
  const float dih[4] = {0.f, ih >= 0.f ? -dy_gmax(y1, Y1) : 0.f, 0.f, ih >= 0.f ? dy_gmin(y2, Y2) : 0.f};
  const float dcw[4] = {-dy_gmin(x1, X1), 0.f, dy_gmax(x2, X2), 0.f};
  const float dch[4] = {0.f, -dy_gmin(y1, Y1), 0.f, dy_gmax(y2, Y2)};
  const float drho[4] = {-sx / 2.f, -sy / 2.f, -sx / 2.f, -sy / 2.f};
  float gc[4];
  for (int k = 0; k < 4; ++k) {
    const float dinter = diw[k] * ihc + iwc * dih[k];
    const float diou = (dinter * uni + inter * dinter) / (uni * uni);          // d uni = -d inter at fixed w1, h1
    float d = diou;
    if (kind == 1) {
      const float dca = dcw[k] * ch + cw * dch[k];
      d += (-dinter * c_area - uni * dca) / (c_area * c_area);
    } else if (kind >= 2) {
      const float dc2 = 2.f * cw * dcw[k] + 2.f * ch * dch[k];
      d -= (drho[k] * c2 - rho2 * dc2) / (c2 * c2);
    }
    gc[k] = d;
  }
  // wrt w1 / h1: they enter through uni = w1 h1 + ... and, for CIoU, through atan(w1 / h1)
  const float den = w1 * w1 + h1 * h1;
  float gw = -inter * h1 / (uni * uni), gh = -inter * w1 / (uni * uni);
  if (kind == 1) { gw += h1 / c_area; gh += w1 / c_area; }
  if (kind == 3) {
    const float k2 = 0.40528473456935116f * 2.f * da * alpha;                 // - alpha dv = + k2 d atan(w1 / h1)
    gw += k2 * (h1 / den);
    gh += k2 * (-w1 / den);
  }
  if (xywh) {          // corners = centre -/+ half extent
    g[0] = gc[0] + gc[2];
    g[1] = gc[1] + gc[3];
    g[2] = (gc[2] - gc[0]) / 2.f + gw;
    g[3] = (gc[3] - gc[1]) / 2.f + gh;
  } else {             // w1 = x2 - x1, h1 = y2 - y1 + eps
    g[0] = gc[0] - gw;
    g[1] = gc[1] - gh;
    g[2] = gc[2] + gw;
    g[3] = gc[3] + gh;
  }
  return val;
}

// softmax over n logits -> probabilities p, returns expectation sum_i i*p_i
DY_HD inline float dy_softmax_expect(const float* x, int n, float* p) {
  float mx = x[0];
  for (int i = 1; i < n; ++i) mx = dy_fmaxf(mx, x[i]);
  float s = 0.f;
  for (int i = 0; i < n; ++i) { p[i] = expf(x[i] - mx); s += p[i]; }
  float inv = 1.f / s, e = 0.f;
  for (int i = 0; i < n; ++i) { p[i] *= inv; e += p[i] * (float)i; }
  return e;
}

// DFL for one side: CE at floor(t) and floor(t)+1 (reference ultralytics/utils/loss.py:75-84), logits x[16]
DY_HD inline float dy_dfl_side(const float* x, float t, float* wl_out, int* tl_out) {
  float mx = x[0];
  for (int i = 1; i < 16; ++i) mx = dy_fmaxf(mx, x[i]);
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += expf(x[i] - mx);
  float lse = mx + logf(s);
  int tl = (int)t;
  float wl = (float)(tl + 1) - t, wr = 1.f - wl;
  if (wl_out) *wl_out = wl;
  if (tl_out) *tl_out = tl;
  return (lse - x[tl]) * wl + (lse - x[tl + 1]) * wr;
}

// BCE-with-logits element (torch formula): (1-t)*x + max(-x,0) + log(exp(-m) + exp(-x-m)), m = max(-x,0)
DY_HD inline float dy_bce(float x, float t) {
  float m = dy_fmaxf(-x, 0.f);
  return (1.f - t) * x + m + logf(expf(-m) + expf(-x - m));
}
