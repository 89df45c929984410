// BatchNorm (training statistics) + activation, forward and backward, NHWC, vectorised 16 B per lane.
// Replaces nn.BatchNorm2d + SiLU / LeakyReLU(0.1) + residual add of the reference (ultralytics/nn/modules/conv.py:49-51,
// block.py:24-45,565; eps / momentum set at utils/torch_utils.py:263-265).  All kernels are HBM-bound streaming kernels.
#include "dy_common.h"
#include "../../include/dedark_yolo.h"

namespace {

__global__ void bn_finalize_kernel(const double* __restrict__ stats, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* running_mean, float* running_var, float momentum,
                                   float eps, float* scale, float* shift, float* mean_out, float* invstd_out, int C) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double m = stats[c] / count;
  double var = stats[C + c] / count - m * m;      // biased (normalisation)
  if (var < 0) var = 0;
  float invstd = (float)(1.0 / sqrt(var + (double)eps));
  float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  float sc = g * invstd;
  scale[c] = sc;
  shift[c] = b - (float)m * sc;
  mean_out[c] = (float)m;
  invstd_out[c] = invstd;
  if (running_mean) {
    double unbiased = count > 1 ? var * count / (count - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

__global__ void bn_fold_eval_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                    float* scale, float* shift, int C) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float sc = gamma[c] / sqrtf(rv[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

template <typename T>
__global__ void bn_act_fwd_kernel(const T* __restrict__ z, long z_ld, const float* __restrict__ scale,
                                  const float* __restrict__ shift, int act, const T* __restrict__ res, long res_ld,
                                  T* __restrict__ y, long y_ld, long pixels, int C) {
  constexpr int VE = DT<T>::VE;
  const int CG = C / VE;
  const long total = pixels * CG;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long px = i / CG;
    int c = (int)(i - px * CG) * VE;
    float v[VE], r[VE];
    ldvec<T>(z + px * z_ld + c, v);
    if (res) ldvec<T>(res + px * res_ld + c, r);
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      float u = v[e] * (scale ? scale[c + e] : 1.f) + (shift ? shift[c + e] : 0.f);
      float o = dy_act(act, u);
      if (res) o += r[e];
      v[e] = o;
    }
    stvec<T>(y + px * y_ld + c, v);
  }
}

// backward pass 1: per-channel sums of g and g*zhat
template <typename T>
__global__ void bn_act_bwd_reduce_kernel(const T* __restrict__ dy, long dy_ld, const T* __restrict__ z, long z_ld,
                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                         const float* __restrict__ mean, const float* __restrict__ invstd, int act,
                                         int has_bn, double* sums, long pixels, int C, int cgb, int rows_per_block) {
  constexpr int VE = DT<T>::VE;
  extern __shared__ float sred[];          // [2][cgb*VE]
  const int CG = C / VE;
  const int tid = threadIdx.x;
  const int cg_local = tid % cgb, prow = tid / cgb;
  const int cg = blockIdx.y * cgb + cg_local;
  const bool active = prow < rows_per_block && cg < CG;
  for (int i = tid; i < 2 * cgb * VE; i += blockDim.x) sred[i] = 0.f;
  __syncthreads();
  float s1[VE], s2[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  if (active) {
    const int c = cg * VE;
    float sc[VE], sh[VE], mu[VE], is[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      sc[e] = scale ? scale[c + e] : 1.f;
      sh[e] = shift ? shift[c + e] : 0.f;
      mu[e] = has_bn ? mean[c + e] : 0.f;
      is[e] = has_bn ? invstd[c + e] : 0.f;
    }
    for (long px = (long)blockIdx.x * rows_per_block + prow; px < pixels; px += (long)gridDim.x * rows_per_block) {
      float g[VE], zz[VE];
      ldvec<T>(dy + px * dy_ld + c, g);
      ldvec<T>(z + px * z_ld + c, zz);
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        float u = zz[e] * sc[e] + sh[e];
        float ge = g[e] * dy_dact(act, u);
        s1[e] += ge;
        s2[e] += ge * (zz[e] - mu[e]) * is[e];
      }
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      atomicAdd(&sred[cg_local * VE + e], s1[e]);
      if (has_bn) atomicAdd(&sred[cgb * VE + cg_local * VE + e], s2[e]);
    }
  }
  __syncthreads();
  for (int i = tid; i < cgb * VE; i += blockDim.x) {
    int c = blockIdx.y * cgb * VE + i;
    if (c < C) {
      atomic_add_f64(sums + c, (double)sred[i]);
      if (has_bn) atomic_add_f64(sums + C + c, (double)sred[cgb * VE + i]);
    }
  }
}

template <typename T>
__global__ void bn_act_bwd_apply_kernel(const T* __restrict__ dy, long dy_ld, const T* __restrict__ z, long z_ld,
                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                        const float* __restrict__ mean, const float* __restrict__ invstd,
                                        const float* __restrict__ gamma, int act, int has_bn,
                                        const double* __restrict__ sums, T* __restrict__ dz, long dz_ld, float* dgamma,
                                        float* dbeta, long pixels, int C) {
  constexpr int VE = DT<T>::VE;
  const int CG = C / VE;
  const long total = pixels * CG;
  const float invM = 1.f / (float)pixels;
  if (blockIdx.x == 0) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      if (dbeta) dbeta[c] = (float)sums[c];
      if (dgamma && has_bn) dgamma[c] = (float)sums[C + c];
    }
  }
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long px = i / CG;
    int c = (int)(i - px * CG) * VE;
    float g[VE], zz[VE];
    ldvec<T>(dy + px * dy_ld + c, g);
    ldvec<T>(z + px * z_ld + c, zz);
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      float u = zz[e] * (scale ? scale[c + e] : 1.f) + (shift ? shift[c + e] : 0.f);
      float ge = g[e] * dy_dact(act, u);
      if (has_bn) {
        float zh = (zz[e] - mean[c + e]) * invstd[c + e];
        float a = (float)sums[c + e] * invM, b = (float)sums[C + c + e] * invM;
        ge = (gamma ? gamma[c + e] : 1.f) * invstd[c + e] * (ge - a - zh * b);
      }
      g[e] = ge;
    }
    stvec<T>(dz + px * dz_ld + c, g);
  }
}

inline int ew_blocks(long total) {
  long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

int check_view(const char* who, const void* p, long ld, int C, int dtype) {
  const int ve = dtype == DY_F32 ? 4 : 8, es = dtype == DY_F32 ? 4 : 2;
  DY_CHECK(p != nullptr, "%s: null pointer", who);
  DY_CHECK(C > 0 && C % ve == 0, "%s: C=%d must be a multiple of %d", who, C, ve);
  DY_CHECK(ld >= C && (ld * es) % 16 == 0 && ((uintptr_t)p) % 16 == 0, "%s: view not 16-byte aligned (ld=%ld)", who, ld);
  return 0;
}

}  // namespace

extern "C" int dy_bn_finalize(const double* stats, int64_t count, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, float momentum, float eps, float* scale, float* shift, float* mean,
                              float* invstd, int C, void* stream) {
  DY_CHECK(stats && scale && shift && mean && invstd && C > 0 && count > 0, "dy_bn_finalize: bad args");
  bn_finalize_kernel<<<dy_cdiv(C, 128), 128, 0, (hipStream_t)stream>>>(stats, (double)count, gamma, beta, running_mean,
                                                                       running_var, momentum, eps, scale, shift, mean,
                                                                       invstd, C);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_bn_fold_eval(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                               float* scale, float* shift, int C, void* stream) {
  DY_CHECK(gamma && beta && rm && rv && scale && shift && C > 0, "dy_bn_fold_eval: bad args");
  bn_fold_eval_kernel<<<dy_cdiv(C, 128), 128, 0, (hipStream_t)stream>>>(gamma, beta, rm, rv, eps, scale, shift, C);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_bn_act_fwd(const void* z, int64_t z_ld, const float* scale, const float* shift, int act,
                             const void* residual, int64_t res_ld, void* y, int64_t y_ld, int64_t pixels, int C, int dtype,
                             void* stream) {
  if (int e = check_view("dy_bn_act_fwd(z)", z, z_ld, C, dtype)) return e;
  if (int e = check_view("dy_bn_act_fwd(y)", y, y_ld, C, dtype)) return e;
  if (residual) if (int e = check_view("dy_bn_act_fwd(res)", residual, res_ld, C, dtype)) return e;
  const int ve = dtype == DY_F32 ? 4 : 8;
  const int blocks = ew_blocks(pixels * (C / ve));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DY_F32)
    bn_act_fwd_kernel<float><<<blocks, 256, 0, st>>>((const float*)z, z_ld, scale, shift, act, (const float*)residual, res_ld,
                                                     (float*)y, y_ld, pixels, C);
  else
    bn_act_fwd_kernel<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)z, z_ld, scale, shift, act, (const bf16_t*)residual,
                                                      res_ld, (bf16_t*)y, y_ld, pixels, C);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_bn_act_bwd_reduce(const void* dy, int64_t dy_ld, const void* z, int64_t z_ld, const float* scale,
                                    const float* shift, const float* mean, const float* invstd, int act, int has_bn,
                                    double* sums, int64_t pixels, int C, int dtype, void* stream) {
  if (int e = check_view("dy_bn_act_bwd_reduce(dy)", dy, dy_ld, C, dtype)) return e;
  if (int e = check_view("dy_bn_act_bwd_reduce(z)", z, z_ld, C, dtype)) return e;
  DY_CHECK(sums && (!has_bn || (mean && invstd)), "dy_bn_act_bwd_reduce: null stats");
  const int ve = dtype == DY_F32 ? 4 : 8;
  const int CG = C / ve;
  const int cgb = CG < 256 ? CG : 256;
  const int rows = 256 / cgb;
  long want = (pixels + rows - 1) / rows;
  int gx = (int)(want > 512 ? 512 : (want < 1 ? 1 : want));
  dim3 grid(gx, dy_cdiv(CG, cgb));
  size_t shm = 2 * (size_t)cgb * ve * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DY_F32)
    bn_act_bwd_reduce_kernel<float><<<grid, 256, shm, st>>>((const float*)dy, dy_ld, (const float*)z, z_ld, scale, shift, mean,
                                                            invstd, act, has_bn, sums, pixels, C, cgb, rows);
  else
    bn_act_bwd_reduce_kernel<bf16_t><<<grid, 256, shm, st>>>((const bf16_t*)dy, dy_ld, (const bf16_t*)z, z_ld, scale, shift,
                                                             mean, invstd, act, has_bn, sums, pixels, C, cgb, rows);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_bn_act_bwd_apply(const void* dy, int64_t dy_ld, const void* z, int64_t z_ld, const float* scale,
                                   const float* shift, const float* mean, const float* invstd, const float* gamma, int act,
                                   int has_bn, const double* sums, void* dz, int64_t dz_ld, float* dgamma, float* dbeta,
                                   int64_t pixels, int C, int dtype, void* stream) {
  if (int e = check_view("dy_bn_act_bwd_apply(dy)", dy, dy_ld, C, dtype)) return e;
  if (int e = check_view("dy_bn_act_bwd_apply(z)", z, z_ld, C, dtype)) return e;
  if (int e = check_view("dy_bn_act_bwd_apply(dz)", dz, dz_ld, C, dtype)) return e;
  DY_CHECK(sums && (!has_bn || (mean && invstd)), "dy_bn_act_bwd_apply: null stats");
  const int ve = dtype == DY_F32 ? 4 : 8;
  const int blocks = ew_blocks(pixels * (C / ve));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DY_F32)
    bn_act_bwd_apply_kernel<float><<<blocks, 256, 0, st>>>((const float*)dy, dy_ld, (const float*)z, z_ld, scale, shift, mean,
                                                           invstd, gamma, act, has_bn, sums, (float*)dz, dz_ld, dgamma, dbeta,
                                                           pixels, C);
  else
    bn_act_bwd_apply_kernel<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)dy, dy_ld, (const bf16_t*)z, z_ld, scale, shift,
                                                            mean, invstd, gamma, act, has_bn, sums, (bf16_t*)dz, dz_ld,
                                                            dgamma, dbeta, pixels, C);
  DY_LAUNCH_CHECK();
  return 0;
}
