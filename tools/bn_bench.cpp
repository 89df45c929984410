// Standalone micro-benchmark of the BatchNorm + activation streaming kernels through the C-ABI (no torch): times
// dy_bn_act_fwd / dy_bn_act_bwd_reduce / dy_bn_act_bwd_apply on the YOLOv8-n (C2) and YOLOv8-L (C3) activation shapes and
// prints the achieved algorithmic GB/s (bytes that must move: fwd z->y, reduce dy+z, apply dy+z->dz).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bn_bench.cpp -Ldedark_yolo_amd/lib -ldedark_yolo -o tools/bin/bn_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../include/dedark_yolo.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define DK(x) do { if ((x) != 0) { printf("dy error: %s (line %d)\n", dy_last_error(), __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int B, C, H; };

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 20;
  int act = argc > 2 ? atoi(argv[2]) : DY_ACT_SILU;
  std::vector<Shape> shapes = {
      {"n P1 16@320", 32, 16, 320}, {"n P2 32@160", 32, 32, 160}, {"n P3 64@80", 32, 64, 80},   {"n P4 128@40", 32, 128, 40},
      {"n P5 256@20", 32, 256, 20}, {"L P1 64@320", 64, 64, 320}, {"L P2 128@160", 64, 128, 160}, {"L P3 256@80", 64, 256, 80},
      {"L P4 512@40", 64, 512, 40}, {"L P5 512@20", 64, 512, 20}, {"n 64@40", 32, 64, 40}, {"n 128@20", 32, 128, 20}, {"n 64@20", 32, 64, 20},
      {"n 32@80", 32, 32, 80},
      {"L 128@80", 64, 128, 80}, {"L 256@40", 64, 256, 40}, {"L 64@160", 64, 64, 160}, {"L 256@20", 64, 256, 20},
  };
  const char* only = getenv("BB_ONLY");
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (auto& s : shapes) {
    if (only && !strstr(s.name, only)) continue;
    const long pixels = (long)s.B * s.H * s.H, n = pixels * s.C;
    unsigned short *z, *y, *dy, *dz;
    float *scale, *shift, *mean, *invstd, *gamma, *dgamma, *dbeta;
    double* sums;
    CK(hipMalloc(&z, n * 2)); CK(hipMalloc(&y, n * 2)); CK(hipMalloc(&dy, n * 2)); CK(hipMalloc(&dz, n * 2));
    CK(hipMemset(z, 0x3c, n * 2)); CK(hipMemset(dy, 0x3b, n * 2));
    std::vector<float> ones(s.C, 1.0f), zeros(s.C, 0.0f);
    float** fp[] = {&scale, &shift, &mean, &invstd, &gamma, &dgamma, &dbeta};
    for (auto p : fp) { CK(hipMalloc(p, s.C * 4)); CK(hipMemcpy(*p, ones.data(), s.C * 4, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&sums, DY_BN_BWD_REPLICAS * 2 * s.C * 8));
    CK(hipMemset(sums, 0, DY_BN_BWD_REPLICAS * 2 * s.C * 8));
    float ms[3];
    for (int k = 0; k < 3; ++k) {
      for (int it = -3; it < iters; ++it) {
        if (it == 0) CK(hipEventRecord(e0, st));
        if (k == 0) DK(dy_bn_act_fwd(z, s.C, scale, shift, act, nullptr, 0, y, s.C, pixels, s.C, DY_BF16, st));
        if (k == 1) DK(dy_bn_act_bwd_reduce(dy, s.C, z, s.C, scale, shift, mean, invstd, act, 1, sums, pixels, s.C, DY_BF16, st));
        if (k == 2) DK(dy_bn_act_bwd_apply(dy, s.C, z, s.C, scale, shift, mean, invstd, gamma, act, 1, sums, dz, s.C, dgamma, dbeta, pixels, s.C, DY_BF16, st));
      }
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms[k], e0, e1));
      ms[k] /= iters;
    }
    const double gb = n * 2 / 1e9;
    printf("%-14s B=%d %7.1f MB/tensor | fwd %7.1f us %6.0f GB/s | reduce %7.1f us %6.0f GB/s | apply %7.1f us %6.0f GB/s\n", s.name, s.B,
           gb * 1e3, ms[0] * 1e3, 2 * gb / (ms[0] * 1e-3), ms[1] * 1e3, 2 * gb / (ms[1] * 1e-3), ms[2] * 1e3, 3 * gb / (ms[2] * 1e-3));
    CK(hipFree(z)); CK(hipFree(y)); CK(hipFree(dy)); CK(hipFree(dz)); CK(hipFree(sums));
    for (auto p : fp) CK(hipFree(*p));
  }
  return 0;
}
