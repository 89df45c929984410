// BatchNorm (training statistics) + activation, forward and backward, NHWC, vectorised 16 B per lane.
// Replaces nn.BatchNorm2d + SiLU / LeakyReLU(0.1) + residual add of the reference (ultralytics/nn/modules/conv.py:49-51,
// block.py:24-45,565; eps / momentum set at utils/torch_utils.py:263-265).  All kernels are HBM-bound streaming kernels.
//
// Thread mapping: a thread owns ONE 16-byte channel group for its whole life (per-channel constants live in registers) and
// walks pixels with a grid stride, two pixels in flight per iteration; a wave covers 64 consecutive channel groups of one
// pixel row (1 KiB contiguous when C >= 512 bf16) or several adjacent pixels for narrower tensors.
#include "dy_common.h"
#include "../../include/dedark_yolo.h"

namespace {

constexpr int NT = 256;

__global__ void bn_finalize_kernel(const double* __restrict__ stats, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* running_mean, float* running_var, float momentum,
                                   float eps, float* scale, float* shift, float* mean_out, float* invstd_out, int C) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int r = 0; r < DY_STATS_REPLICAS; ++r) {
    s1 += stats[(long)r * 2 * C + c];
    s2 += stats[(long)r * 2 * C + C + c];
  }
  double m = s1 / count;
  double var = s2 / count - m * m;      // biased (normalisation)
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

struct Map {           // thread -> (channel group, first pixel, pixel stride)
  int c;               // first channel of the group
  long first, step;
  bool active;
};

template <int VE>
__device__ inline Map make_map(int C, int cgb, int rows) {
  Map m;
  const int CG = C / VE;
  const int cg_local = threadIdx.x % cgb, prow = threadIdx.x / cgb;
  const int cg = blockIdx.y * cgb + cg_local;
  m.active = prow < rows && cg < CG;
  m.c = cg * VE;
  m.first = (long)blockIdx.x * rows + prow;
  m.step = (long)gridDim.x * rows;
  return m;
}

template <typename T>
__global__ __launch_bounds__(NT) void bn_act_fwd_kernel(const T* __restrict__ z, long z_ld, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int act, const T* __restrict__ res,
                                                         long res_ld, T* __restrict__ y, long y_ld, long pixels, int C, int cgb,
                                                         int rows) {
  constexpr int VE = DT<T>::VE;
  const Map m = make_map<VE>(C, cgb, rows);
  if (!m.active) return;
  float sc[VE], sh[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    sc[e] = scale ? scale[m.c + e] : 1.f;
    sh[e] = shift ? shift[m.c + e] : 0.f;
  }
  for (long pa = m.first; pa < pixels; pa += 2 * m.step) {
    const long pb = pa + m.step;
    const bool hb = pb < pixels;
    float va[VE], vb[VE], ra[VE], rb[VE];
    ldvec<T>(z + pa * z_ld + m.c, va);
    if (hb) ldvec<T>(z + pb * z_ld + m.c, vb);
    if (res) {
      ldvec<T>(res + pa * res_ld + m.c, ra);
      if (hb) ldvec<T>(res + pb * res_ld + m.c, rb);
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      float o = dy_act(act, va[e] * sc[e] + sh[e]);
      va[e] = res ? o + ra[e] : o;
    }
    stvec<T>(y + pa * y_ld + m.c, va);
    if (hb) {
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        float o = dy_act(act, vb[e] * sc[e] + sh[e]);
        vb[e] = res ? o + rb[e] : o;
      }
      stvec<T>(y + pb * y_ld + m.c, vb);
    }
  }
}

// backward pass 1: per-channel sums of g and g*zhat
template <typename T>
__global__ __launch_bounds__(NT) void bn_act_bwd_reduce_kernel(const T* __restrict__ dy, long dy_ld, const T* __restrict__ z,
                                                                long z_ld, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, int act, int has_bn,
                                                                double* sums, long pixels, int C, int cgb, int rows) {
  constexpr int VE = DT<T>::VE;
  extern __shared__ float sred[];          // [2][cgb*VE]
  const int tid = threadIdx.x;
  const Map m = make_map<VE>(C, cgb, rows);
  const int cg_local = tid % cgb;
  for (int i = tid; i < 2 * cgb * VE; i += NT) sred[i] = 0.f;
  __syncthreads();
  if (m.active) {
    float sc[VE], sh[VE], mu[VE], is[VE], s1[VE], s2[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      sc[e] = scale ? scale[m.c + e] : 1.f;
      sh[e] = shift ? shift[m.c + e] : 0.f;
      mu[e] = has_bn ? mean[m.c + e] : 0.f;
      is[e] = has_bn ? invstd[m.c + e] : 0.f;
      s1[e] = 0.f;
      s2[e] = 0.f;
    }
    for (long pa = m.first; pa < pixels; pa += 2 * m.step) {
      const long pb = pa + m.step;
      const bool hb = pb < pixels;
      float ga[VE], za[VE], gb[VE], zb[VE];
      ldvec<T>(dy + pa * dy_ld + m.c, ga);
      ldvec<T>(z + pa * z_ld + m.c, za);
      if (hb) {
        ldvec<T>(dy + pb * dy_ld + m.c, gb);
        ldvec<T>(z + pb * z_ld + m.c, zb);
      }
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        float ge = ga[e] * dy_dact(act, za[e] * sc[e] + sh[e]);
        s1[e] += ge;
        s2[e] += ge * (za[e] - mu[e]) * is[e];
      }
      if (hb) {
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          float ge = gb[e] * dy_dact(act, zb[e] * sc[e] + sh[e]);
          s1[e] += ge;
          s2[e] += ge * (zb[e] - mu[e]) * is[e];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      atomicAdd(&sred[cg_local * VE + e], s1[e]);
      if (has_bn) atomicAdd(&sred[cgb * VE + cg_local * VE + e], s2[e]);
    }
  }
  __syncthreads();
  for (int i = tid; i < cgb * VE; i += NT) {
    int c = blockIdx.y * cgb * VE + i;
    if (c < C) {
      atomic_add_f64(sums + c, (double)sred[i]);
      if (has_bn) atomic_add_f64(sums + C + c, (double)sred[cgb * VE + i]);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void bn_act_bwd_apply_kernel(const T* __restrict__ dy, long dy_ld, const T* __restrict__ z,
                                                               long z_ld, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                               int act, int has_bn, const double* __restrict__ sums,
                                                               T* __restrict__ dz, long dz_ld, float* dgamma, float* dbeta,
                                                               long pixels, long count, int C, int cgb, int rows) {
  constexpr int VE = DT<T>::VE;
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < cgb * VE; i += NT) {
      int c = blockIdx.y * cgb * VE + i;
      if (c < C) {
        if (dbeta) dbeta[c] = (float)sums[c];
        if (dgamma && has_bn) dgamma[c] = (float)sums[C + c];
      }
    }
  }
  const Map m = make_map<VE>(C, cgb, rows);
  if (!m.active) return;
  const float invM = 1.f / (float)count;
  float sc[VE], sh[VE], mu[VE], is[VE], k1[VE], ka[VE], kb[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    sc[e] = scale ? scale[m.c + e] : 1.f;
    sh[e] = shift ? shift[m.c + e] : 0.f;
    mu[e] = has_bn ? mean[m.c + e] : 0.f;
    is[e] = has_bn ? invstd[m.c + e] : 0.f;
    k1[e] = has_bn ? (gamma ? gamma[m.c + e] : 1.f) * is[e] : 1.f;
    ka[e] = has_bn ? (float)sums[m.c + e] * invM : 0.f;
    kb[e] = has_bn ? (float)sums[C + m.c + e] * invM : 0.f;
  }
  for (long pa = m.first; pa < pixels; pa += 2 * m.step) {
    const long pb = pa + m.step;
    const bool hb = pb < pixels;
    float ga[VE], za[VE], gb[VE], zb[VE];
    ldvec<T>(dy + pa * dy_ld + m.c, ga);
    ldvec<T>(z + pa * z_ld + m.c, za);
    if (hb) {
      ldvec<T>(dy + pb * dy_ld + m.c, gb);
      ldvec<T>(z + pb * z_ld + m.c, zb);
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      float ge = ga[e] * dy_dact(act, za[e] * sc[e] + sh[e]);
      ga[e] = k1[e] * (ge - ka[e] - (za[e] - mu[e]) * is[e] * kb[e]);
    }
    stvec<T>(dz + pa * dz_ld + m.c, ga);
    if (hb) {
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        float ge = gb[e] * dy_dact(act, zb[e] * sc[e] + sh[e]);
        gb[e] = k1[e] * (ge - ka[e] - (zb[e] - mu[e]) * is[e] * kb[e]);
      }
      stvec<T>(dz + pb * dz_ld + m.c, gb);
    }
  }
}

struct Geo { int cgb, rows; dim3 grid; };

Geo geometry(long pixels, int C, int ve, int max_x) {
  Geo g;
  const int CG = C / ve;
  g.cgb = CG < NT ? CG : NT;
  g.rows = NT / g.cgb;
  long want = (pixels + 2L * g.rows - 1) / (2L * g.rows);       // two pixels per thread per iteration
  int gx = (int)(want > max_x ? max_x : (want < 1 ? 1 : want));
  g.grid = dim3(gx, dy_cdiv(CG, g.cgb));
  return g;
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
  if (pixels <= 0) return 0;
  const Geo g = geometry(pixels, C, dtype == DY_F32 ? 4 : 8, 4096);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DY_F32)
    bn_act_fwd_kernel<float><<<g.grid, NT, 0, st>>>((const float*)z, z_ld, scale, shift, act, (const float*)residual, res_ld,
                                                    (float*)y, y_ld, pixels, C, g.cgb, g.rows);
  else
    bn_act_fwd_kernel<bf16_t><<<g.grid, NT, 0, st>>>((const bf16_t*)z, z_ld, scale, shift, act, (const bf16_t*)residual, res_ld,
                                                     (bf16_t*)y, y_ld, pixels, C, g.cgb, g.rows);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_bn_act_bwd_reduce(const void* dy, int64_t dy_ld, const void* z, int64_t z_ld, const float* scale,
                                    const float* shift, const float* mean, const float* invstd, int act, int has_bn,
                                    double* sums, int64_t pixels, int C, int dtype, void* stream) {
  if (int e = check_view("dy_bn_act_bwd_reduce(dy)", dy, dy_ld, C, dtype)) return e;
  if (int e = check_view("dy_bn_act_bwd_reduce(z)", z, z_ld, C, dtype)) return e;
  DY_CHECK(sums && (!has_bn || (mean && invstd)), "dy_bn_act_bwd_reduce: null stats");
  if (pixels <= 0) return 0;
  const int ve = dtype == DY_F32 ? 4 : 8;
  const Geo g = geometry(pixels, C, ve, 1024);
  size_t shm = 2 * (size_t)g.cgb * ve * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DY_F32)
    bn_act_bwd_reduce_kernel<float><<<g.grid, NT, shm, st>>>((const float*)dy, dy_ld, (const float*)z, z_ld, scale, shift, mean,
                                                             invstd, act, has_bn, sums, pixels, C, g.cgb, g.rows);
  else
    bn_act_bwd_reduce_kernel<bf16_t><<<g.grid, NT, shm, st>>>((const bf16_t*)dy, dy_ld, (const bf16_t*)z, z_ld, scale, shift,
                                                              mean, invstd, act, has_bn, sums, pixels, C, g.cgb, g.rows);
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
  // pixels == 0: only the parameter gradients (dgamma / dbeta) are written
  const Geo g = geometry(pixels > 0 ? pixels : 1, C, dtype == DY_F32 ? 4 : 8, 4096);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DY_F32)
    bn_act_bwd_apply_kernel<float><<<g.grid, NT, 0, st>>>((const float*)dy, dy_ld, (const float*)z, z_ld, scale, shift, mean, invstd,
                                                          gamma, act, has_bn, sums, (float*)dz, dz_ld, dgamma, dbeta, pixels,
                                                          pixels > 0 ? pixels : 1, C, g.cgb, g.rows);
  else
    bn_act_bwd_apply_kernel<bf16_t><<<g.grid, NT, 0, st>>>((const bf16_t*)dy, dy_ld, (const bf16_t*)z, z_ld, scale, shift, mean,
                                                           invstd, gamma, act, has_bn, sums, (bf16_t*)dz, dz_ld, dgamma, dbeta,
                                                           pixels, pixels > 0 ? pixels : 1, C, g.cgb, g.rows);
  DY_LAUNCH_CHECK();
  return 0;
}
