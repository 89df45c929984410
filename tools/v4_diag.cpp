// Ablation timing of the phase-interleaved conv kernel (csrc/conv_v4.hip compiled into this tool with its ABL template variants;
// the library ships ABL = 0 only).  Prints us / TF per variant for one layer shape: which resource the K-loop is waiting for.
//   tools/bin/v4_diag [B=64] [C=256] [HW=40] [k=3]
#define DY_V4_DIAG_BUILD 1
#include "../dedark_yolo_amd/csrc/conv_v4.hip"
#include <stdio.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

template <int ABL>
static void run(const char* what, dy_conv_desc& f, double flops, hipStream_t st, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) if (v4_launch<ABL>(&f, 0, st)) { printf("launch failed: %s\n", dy_last_error()); exit(1); }
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) v4_launch<ABL>(&f, 0, st);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= iters;
  printf("  ABL %3d  %-46s %8.1f us  %7.1f TF\n", ABL, what, ms * 1e3, flops / ms / 1e9);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, Cc = argc > 2 ? atoi(argv[2]) : 256, HW = argc > 3 ? atoi(argv[3]) : 40, k = argc > 4 ? atoi(argv[4]) : 3;
  // (the tile variant follows the channel count: 256 -> 256 x 256 tile, 128 -> 256 x 128 tile)
  const int pad = k / 2;
  const long nx = (long)B * HW * HW * Cc, nw = (long)Cc * k * k * Cc;
  std::vector<unsigned short> hx(nx), hw(nw);
  unsigned seed = 12345;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hx) v = f2bf(rnd());
  for (auto& v : hw) v = f2bf(rnd() * 0.05f);
  unsigned short *dx, *dw, *dy;
  CK(hipMalloc(&dx, nx * 2)); CK(hipMalloc(&dw, nw * 2)); CK(hipMalloc(&dy, nx * 2));
  CK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  dy_conv_desc f = {};
  f.src = dx; f.src_ld = Cc; f.N = B; f.Hs = HW; f.Ws = HW; f.Cs = Cc; f.w = dw; f.dst = dy; f.dst_ld = Cc;
  f.Hd = HW; f.Wd = HW; f.Cd = Cc; f.KH = f.KW = k; f.stride = 1; f.pad = pad; f.dil = 1; f.dtype = DY_BF16;
  const double flops = 2.0 * B * HW * HW * (double)Cc * k * k * Cc;
  printf("v4 ablations: %dx%d %d->%d @%d B=%d  (%.1f GF, %ld tiles)\n", k, k, Cc, Cc, HW, B, flops / 1e9, ((long)B * HW * HW + 255) / 256 * ((Cc + 255) / 256));
  const int it = 20;
  run<0>("shipped kernel", f, flops, st, it);
  run<1>("no DMA inside the loop", f, flops, st, it);
  run<16>("no A-side DMA", f, flops, st, it);
  run<32>("no B-side DMA", f, flops, st, it);
  run<4>("no LDS fragment reads", f, flops, st, it);
  run<2>("no MFMA", f, flops, st, it);
  run<8>("no stagger (groups in lockstep)", f, flops, st, it);
  run<64>("no epilogue stores", f, flops, st, it);
  run<128>("no s_setprio around the MFMA clusters", f, flops, st, it);
  run<128 | 256>("priority to the loading wave", f, flops, st, it);
  run<0>("shipped kernel (again)", f, flops, st, it);
  run<128>("no s_setprio (again)", f, flops, st, it);
  run<1 | 4>("MFMA + barriers only", f, flops, st, it);
  run<1 | 4 | 2048>("MFMA + HALF the barriers (phases merged in pairs; timing only)", f, flops, st, it);
  run<2048>("everything, phases merged in pairs (WRONG results; timing only)", f, flops, st, it);
  run<1 | 2>("LDS reads + barriers only", f, flops, st, it);
  run<2 | 4>("DMA + barriers only", f, flops, st, it);
  run<1 | 2 | 4>("barriers + scalar bookkeeping only", f, flops, st, it);
  run<0>("shipped kernel (again)", f, flops, st, it);
  if (Cc >= 192) {        // phase stamps (256 x 256 kernel): cycles per K-step of [reads + DMA issue + barrier | MFMA cluster | second barrier]
    unsigned long long* dbg;
    CK(hipMalloc(&dbg, 24 * 8));
    CK(hipMemset(dbg, 0, 24 * 8));
    f.stats = (double*)dbg;
    run<512>("stamped build (not a timing)", f, flops, st, 1);
    unsigned long long h[24];
    CK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
    const double nks = (double)(k * k * Cc / 64) * 4.0;     // 3 warm-up launches + 1 timed one accumulate into the same sums... per launch reset below
    (void)nks;
    for (int w = 0; w < 2; ++w) {
      printf("  group %d (wave %d) cycles per K-step and phase  [load part + wait | MFMA cluster | second barrier]\n", w, 4 * w);
      double tot = 0;
      for (int ph = 0; ph < 4; ++ph) {
        const double a = h[12 * w + 3 * ph] / (double)(k * k * Cc / 64), b = h[12 * w + 3 * ph + 1] / (double)(k * k * Cc / 64),
                     c = h[12 * w + 3 * ph + 2] / (double)(k * k * Cc / 64);
        printf("    P%d  %7.0f | %7.0f | %7.0f\n", ph + 1, a, b, c);
        tot += a + b + c;
      }
      printf("    sum %7.0f cycles per K-step (1024 cycles of MFMA per wave; the stamps themselves cost ~40 each)\n", tot);
    }
    f.stats = nullptr;
  }
  return 0;
}
