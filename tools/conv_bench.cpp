// Standalone micro-benchmark of the implicit-GEMM conv kernels through the C-ABI (no torch): times fwd / dgrad / wgrad on
// the YOLOv8-L stage shapes with hipEvents and prints TFLOP/s and the fraction of the dense bf16 MFMA peak.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/conv_bench.cpp -Ldedark_yolo_amd/lib -ldedark_yolo -o gpurun_out/conv_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "../include/dedark_yolo.h"
extern "C" int dy_debug_conv_stamps(unsigned long long* out);
extern "C" int dy_debug_conv3_stamps(unsigned long long* out);
extern "C" int dy_debug_conv5_stamps(unsigned long long* out);

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int B, Cin, Cout, H, W, k, s, p; };

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 20;
  int B = argc > 2 ? atoi(argv[2]) : 64;
  std::vector<Shape> shapes = {
      {"s3 3x3 256->256 @40", B, 256, 256, 40, 40, 3, 1, 1},  {"s2 3x3 128->128 @80", B, 128, 128, 80, 80, 3, 1, 1},
      {"s1 3x3 64->64 @160", B, 64, 64, 160, 160, 3, 1, 1},   {"s4 3x3 512->512 @20", B, 512, 512, 20, 20, 3, 1, 1},
      {"1x1 1280->512 @40", B, 1280, 512, 40, 40, 1, 1, 0},   {"1x1 256->256 @80", B, 256, 256, 80, 80, 1, 1, 0},
      {"3x3s2 256->512 @80", B, 256, 512, 80, 80, 3, 2, 1},   {"3x3s2 64->128 @320", B, 64, 128, 320, 320, 3, 2, 1},
      {"1x1 320->128 @160", B, 320, 128, 160, 160, 1, 1, 0},  {"1x1 1024->256 @80", B, 1024, 256, 80, 80, 1, 1, 0},
      {"1x1 2048->512 @40", B, 2048, 512, 40, 40, 1, 1, 0},   {"s3 3x3 256->256 @80", B, 256, 256, 80, 80, 3, 1, 1},
      {"s4 3x3 512->512 @40", B, 512, 512, 40, 40, 3, 1, 1},
      {"d 3x3 256->64 @80", B, 256, 64, 80, 80, 3, 1, 1},      {"d 3x3 512->64 @40", B, 512, 64, 40, 40, 3, 1, 1},       // Detect box-branch stems
      {"s4 3x3 256->256 @20", B, 256, 256, 20, 20, 3, 1, 1},   {"s4 3x3 128->128 @40", B, 128, 128, 40, 40, 3, 1, 1},
      {"1x1 128->128 @160", B, 128, 128, 160, 160, 1, 1, 0},   {"3x3s2 128->256 @160", B, 128, 256, 160, 160, 3, 2, 1},
      {"3x3s2 512->512 @40", B, 512, 512, 40, 40, 3, 2, 1},    {"3x3s2 256->256 @80", B, 256, 256, 80, 80, 3, 2, 1},
      {"stem 3x3s2 8->64 @640", B, 8, 64, 640, 640, 3, 2, 1},
      {"1x1 64->64 @160", B, 64, 64, 160, 160, 1, 1, 0},
  };
  if (argc > 3 && argv[3][0] == 'n') {       // YOLOv8-n layers with few output pixels (B = 32: 12,800 / 51,200 rows)
    shapes = {
        {"n 3x3 128->128 @20", B, 128, 128, 20, 20, 3, 1, 1}, {"n 3x3 256->64 @20", B, 256, 64, 20, 20, 3, 1, 1},
        {"n 1x1 384->256 @20", B, 384, 256, 20, 20, 1, 1, 0}, {"n 1x1 256->256 @20", B, 256, 256, 20, 20, 1, 1, 0},
        {"n 1x1 512->256 @20", B, 512, 256, 20, 20, 1, 1, 0}, {"n 3x3s2 128->256 @40", B, 128, 256, 40, 40, 3, 2, 1},
        {"n 3x3 64->64 @40", B, 64, 64, 40, 40, 3, 1, 1},     {"n 1x1 192->128 @40", B, 192, 128, 40, 40, 1, 1, 0},
        {"n 3x3s2 128->128 @40", B, 128, 128, 40, 40, 3, 2, 1}, {"n 3x3 128->64 @40", B, 128, 64, 40, 40, 3, 1, 1},
        {"n 1x1 256->128 @20", B, 256, 128, 20, 20, 1, 1, 0},
    };
  }
  const int warm = getenv("CB_WARM") ? atoi(getenv("CB_WARM")) : 40;
  const char* only = getenv("CB_ONLY");          // substring filter on the shape name
  const bool check = getenv("CB_CHECK") != nullptr;   // sampled CPU check of fwd / dgrad / wgrad outputs
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float* scratch;
  const long scratch_elems = 32L << 20;
  CK(hipMalloc(&scratch, scratch_elems * 4));
  for (auto& s : shapes) {
    if (only && !strstr(s.name, only)) continue;
    const int Ho = (s.H + 2 * s.p - s.k) / s.s + 1, Wo = (s.W + 2 * s.p - s.k) / s.s + 1;
    const long nx = (long)s.B * s.H * s.W * s.Cin, ny = (long)s.B * Ho * Wo * s.Cout, nw = (long)s.Cout * s.k * s.k * s.Cin;
    std::vector<unsigned short> hx(nx), hw(nw), hy(ny);
    unsigned seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hx) v = f2bf(rnd());
    for (auto& v : hw) v = f2bf(rnd() * 0.05f);
    for (auto& v : hy) v = f2bf(rnd());
    unsigned short *dx, *dw, *dwt, *dy, *dz;
    float* gw;
    double* stats;
    CK(hipMalloc(&dx, nx * 2)); CK(hipMalloc(&dw, nw * 2)); CK(hipMalloc(&dwt, nw * 2)); CK(hipMalloc(&dy, ny * 2));
    CK(hipMalloc(&dz, nx * 2)); CK(hipMalloc(&gw, nw * 4)); CK(hipMalloc(&stats, 64 * 2 * s.Cout * 8));
    CK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dwt, hw.data(), nw * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dy, hy.data(), ny * 2, hipMemcpyHostToDevice));
    CK(hipMemset(stats, 0, 64 * 2 * s.Cout * 8));
    dy_conv_desc f = {};
    f.src = dx; f.src_ld = s.Cin; f.N = s.B; f.Hs = s.H; f.Ws = s.W; f.Cs = s.Cin; f.w = dw; f.dst = dy; f.dst_ld = s.Cout;
    f.Hd = Ho; f.Wd = Wo; f.Cd = s.Cout; f.KH = f.KW = s.k; f.stride = s.s; f.pad = s.p; f.dil = 1; f.stats = stats; f.dtype = DY_BF16;
    dy_conv_desc g = {};
    g.src = dy; g.src_ld = s.Cout; g.N = s.B; g.Hs = Ho; g.Ws = Wo; g.Cs = s.Cout; g.w = dwt; g.dst = dz; g.dst_ld = s.Cin;
    g.Hd = s.H; g.Wd = s.W; g.Cd = s.Cin; g.KH = g.KW = s.k; g.stride = s.s; g.pad = s.p; g.dil = 1; g.dtype = DY_BF16;
    if (getenv("CB_ACC")) g.accumulate = 1;                 // the data gradient as the graph issues it for a fan-out: dx += ...
    if (getenv("CB_ADD")) { g.add_src = dx; g.add_src_ld = s.Cin; }        // ... or dx = dgrad + another gradient view
    const double flops = 2.0 * s.B * Ho * Wo * (double)s.Cout * s.k * s.k * s.Cin;
    float ms[3];
    for (int which = 0; which < 3; ++which) {
      auto run = [&]() {
        int rc = 0;
        if (which == 0) rc = dy_conv2d_fwd(&f, st);
        else if (which == 1) rc = dy_conv2d_dgrad(&g, st);
        else rc = dy_conv2d_wgrad(dx, s.Cin, s.B, s.H, s.W, s.Cin, dy, s.Cout, Ho, Wo, s.Cout, s.k, s.k, s.s, s.p, 1, s.Cout, s.Cin,
                                  scratch, scratch_elems, gw, DY_BF16, st);
        if (rc) { printf("call failed: %s\n", dy_last_error()); exit(1); }
      };
      for (int i = 0; i < warm; ++i) run();          // the clock needs ~10 ms of load to settle: 3 launches read 10-15 % slow
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) run();
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms[which], e0, e1));
      ms[which] /= iters;
    }
    if (getenv("DY_ABLATE") && (atoi(getenv("DY_ABLATE")) & 32) && getenv("CB_V5")) {       // conv_v5 stamps: first block and a late one
      unsigned long long t[16];
      for (int which = 0; which < 2; ++which) {
        CK(hipDeviceSynchronize());
        if (which == 0) dy_conv2d_fwd(&f, st); else dy_conv2d_dgrad(&g, st);
        CK(hipDeviceSynchronize());
        dy_debug_conv5_stamps(t);
        for (int b = 0; b < 16; b += 8)
          printf("   %s %s block (shader clocks): start +%llu | prologue %llu | K loop %llu | lds image %llu | stores %llu | stats %llu | total %llu\n", which ? "dgrad" : "fwd",
                 b ? "late" : "first", t[b] - t[0], t[b + 1] - t[b], t[b + 2] - t[b + 1], t[b + 3] - t[b + 2], t[b + 4] - t[b + 3], t[b + 5] - t[b + 4], t[b + 5] - t[b]);
      }
    } else if (getenv("DY_ABLATE") && (atoi(getenv("DY_ABLATE")) & 32)) {
      unsigned long long t[16];
      CK(hipDeviceSynchronize());
      dy_conv2d_fwd(&f, st);
      CK(hipDeviceSynchronize());
      if (s.k == 3 && s.s == 1) dy_debug_conv3_stamps(t); else dy_debug_conv_stamps(t);
      printf("   fwd block0 cycles: prologue %llu | issue01 %llu | step0 %llu | steps1-8 %llu (per step %llu) | steps9-end %llu | lds image %llu | stores %llu\n",
             t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], (t[4] - t[3]) / 8, t[5] - t[4], t[6] - t[5], t[7] - t[6]);
    }
    if (check) {
      // sampled reference: fwd y[m][co], dgrad dx[pixel][ci] (weights read as the transposed pack [Cin][KH][KW][Cout]), wgrad g[co][ci][kh][kw]
      auto bf = [](unsigned short v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return (double)f; };
      std::vector<unsigned short> oy(ny), odx(nx);
      std::vector<float> ogw(nw);
      f.stats = nullptr;
      CK(hipMemcpy(dy, hy.data(), ny * 2, hipMemcpyHostToDevice));
      if (dy_conv2d_dgrad(&g, st)) { printf("dgrad failed: %s\n", dy_last_error()); exit(1); }
      if (dy_conv2d_wgrad(dx, s.Cin, s.B, s.H, s.W, s.Cin, dy, s.Cout, Ho, Wo, s.Cout, s.k, s.k, s.s, s.p, 1, s.Cout, s.Cin, scratch, scratch_elems, gw, DY_BF16, st)) { printf("wgrad failed\n"); exit(1); }
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(odx.data(), dz, nx * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(ogw.data(), gw, nw * 4, hipMemcpyDeviceToHost));
      if (dy_conv2d_fwd(&f, st)) { printf("fwd failed: %s\n", dy_last_error()); exit(1); }
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(oy.data(), dy, ny * 2, hipMemcpyDeviceToHost));
      double e_f = 0, e_d = 0, e_w = 0, m_f = 0, m_d = 0, m_w = 0;
      unsigned sd = 777;
      auto rr = [&](long n) { sd = sd * 1664525u + 1013904223u; return (long)((sd >> 4) % (unsigned long)n); };
      const long Mo = (long)s.B * Ho * Wo;
      for (int t = 0; t < 512; ++t) {
        long m = t < 8 ? t : (t < 16 ? Mo - 1 - (t - 8) : rr(Mo));
        int co = t < 16 ? (t * 37) % s.Cout : (int)rr(s.Cout);
        int b = (int)(m / ((long)Ho * Wo)), oh = (int)((m / Wo) % Ho), ow = (int)(m % Wo);
        double a = 0;
        for (int kh = 0; kh < s.k; ++kh) for (int kw = 0; kw < s.k; ++kw) {
          int ih = oh * s.s - s.p + kh, iw = ow * s.s - s.p + kw;
          if (ih < 0 || ih >= s.H || iw < 0 || iw >= s.W) continue;
          const unsigned short* xr = &hx[(((long)b * s.H + ih) * s.W + iw) * s.Cin];
          const unsigned short* wr_ = &hw[(((long)co * s.k + kh) * s.k + kw) * s.Cin];
          for (int c = 0; c < s.Cin; ++c) a += bf(xr[c]) * bf(wr_[c]);
        }
        double got = bf(oy[m * s.Cout + co]);
        e_f = fmax(e_f, fabs(got - a)); m_f = fmax(m_f, fabs(a));
      }
      for (int t = 0; t < 512; ++t) {
        long pix = t < 8 ? t : (t < 16 ? (long)s.B * s.H * s.W - 1 - (t - 8) : rr((long)s.B * s.H * s.W));
        int ci = (int)rr(s.Cin);
        int b = (int)(pix / ((long)s.H * s.W)), h = (int)((pix / s.W) % s.H), w_ = (int)(pix % s.W);
        double a = 0;
        for (int kh = 0; kh < s.k; ++kh) for (int kw = 0; kw < s.k; ++kw) {
          int th = h + s.p - kh, tw = w_ + s.p - kw;
          if (th < 0 || tw < 0 || th % s.s || tw % s.s) continue;
          int oh = th / s.s, ow = tw / s.s;
          if (oh >= Ho || ow >= Wo) continue;
          const unsigned short* zr = &hy[(((long)b * Ho + oh) * Wo + ow) * s.Cout];
          const unsigned short* wr_ = &hw[(((long)ci * s.k + kh) * s.k + kw) * s.Cout];     // transposed pack [Cin][KH][KW][Cout]
          for (int c = 0; c < s.Cout; ++c) a += bf(zr[c]) * bf(wr_[c]);
        }
        double got = bf(odx[pix * s.Cin + ci]);
        e_d = fmax(e_d, fabs(got - a)); m_d = fmax(m_d, fabs(a));
      }
      for (int t = 0; t < 48; ++t) {
        int co = (int)rr(s.Cout), ci = (int)rr(s.Cin), kh = (int)rr(s.k), kw = (int)rr(s.k);
        double a = 0;
        for (int b = 0; b < s.B; ++b) for (int oh = 0; oh < Ho; ++oh) for (int ow = 0; ow < Wo; ++ow) {
          int ih = oh * s.s - s.p + kh, iw = ow * s.s - s.p + kw;
          if (ih < 0 || ih >= s.H || iw < 0 || iw >= s.W) continue;
          a += bf(hy[(((long)b * Ho + oh) * Wo + ow) * s.Cout + co]) * bf(hx[(((long)b * s.H + ih) * s.W + iw) * s.Cin + ci]);
        }
        double got = ogw[(((long)co * s.Cin + ci) * s.k + kh) * s.k + kw];
        e_w = fmax(e_w, fabs(got - a)); m_w = fmax(m_w, fabs(a));
      }
      printf("   check (max abs err / max |ref|): fwd %.3g / %.3g  dgrad %.3g / %.3g  wgrad %.3g / %.3g  %s\n", e_f, m_f, e_d, m_d, e_w, m_w,
             (e_f <= 1e-2 * m_f && e_d <= 1e-2 * m_d && e_w <= 2e-3 * m_w) ? "OK" : "MISMATCH");
    }
    printf("%-22s B=%d GF=%7.1f | fwd %7.1f us %6.1f TF (%4.1f%%) | dgrad %7.1f us %6.1f TF | wgrad %7.1f us %6.1f TF\n", s.name, s.B,
           flops / 1e9, ms[0] * 1e3, flops / ms[0] / 1e9, flops / ms[0] / 1e9 / 25.0, ms[1] * 1e3, flops / ms[1] / 1e9, ms[2] * 1e3,
           flops / ms[2] / 1e9);
    fflush(stdout);
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dwt)); CK(hipFree(dy)); CK(hipFree(dz)); CK(hipFree(gw)); CK(hipFree(stats));
  }
  return 0;
}
