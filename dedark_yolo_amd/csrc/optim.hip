// Fused optimizer step on flat f32 parameter ranges: gradient clipping coefficient, SGD(nesterov)/AdamW update and the
// EMA lerp in ONE pass (reference: BaseTrainer.optimizer_step ultralytics/engine/trainer.py:459-467, build_optimizer
// :611-665, ModelEMA.update ultralytics/utils/torch_utils.py:360-371).  Pure HBM streaming: 4-5 reads + 3 writes / element.
#include "dy_common.h"
#include "../../include/dedark_yolo.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, double* acc) {
  __shared__ float sm[20];
  float s = 0.f;
  const long n4 = n >> 2;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 v = g4[i];
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { float v = g[(n4 << 2) + threadIdx.x]; s += v * v; }
  s = block_sum(s, sm);
  if (threadIdx.x == 0) atomic_add_f64(acc, (double)s);
}

__device__ inline float clip_coef(const double* sumsq, float max_norm) {
  if (!sumsq) return 1.f;
  float nrm = (float)sqrt(*sumsq);
  float c = max_norm / (nrm + 1e-6f);
  return c < 1.f ? c : 1.f;
}

__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, float* __restrict__ ema,
                           float lr, float mom, float wd, int nesterov, float ed, const double* sumsq, float max_norm, long n) {
  const float cc = clip_coef(sumsq, max_norm);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float w = p[i];
    float d = g[i] * cc + wd * w;
    float b = mom * buf[i] + d;
    buf[i] = b;
    d = nesterov ? d + mom * b : b;
    w -= lr * d;
    p[i] = w;
    if (ema) ema[i] = ed * ema[i] + (1.f - ed) * w;
  }
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m1, float* __restrict__ m2,
                             float* __restrict__ ema, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2,
                             float ed, const double* sumsq, float max_norm, long n) {
  const float cc = clip_coef(sumsq, max_norm);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float w = p[i] * (1.f - lr * wd);
    float gi = g[i] * cc;
    float a = b1 * m1[i] + (1.f - b1) * gi;
    float v = b2 * m2[i] + (1.f - b2) * gi * gi;
    m1[i] = a;
    m2[i] = v;
    float denom = sqrtf(v) / sqrtf(bc2) + eps;
    w -= (lr / bc1) * a / denom;
    p[i] = w;
    if (ema) ema[i] = ed * ema[i] + (1.f - ed) * w;
  }
}

inline int ew_blocks(long n) {
  long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" int dy_sumsq(const float* g, int64_t n, double* acc, void* stream) {
  DY_CHECK(g && acc && n >= 0 && ((uintptr_t)g) % 16 == 0, "dy_sumsq: bad args");
  sumsq_kernel<<<ew_blocks(n / 4 + 1), 256, 0, (hipStream_t)stream>>>(g, n, acc);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_sgd_step(float* p, const float* g, float* mom_buf, float* ema, float lr, float momentum, float weight_decay,
                           int nesterov, float ema_decay, const double* sumsq, float max_norm, int64_t n, void* stream) {
  DY_CHECK(p && g && mom_buf && n >= 0, "dy_sgd_step: bad args");
  if (n == 0) return 0;
  sgd_kernel<<<ew_blocks(n), 256, 0, (hipStream_t)stream>>>(p, g, mom_buf, ema, lr, momentum, weight_decay, nesterov, ema_decay,
                                                            sumsq, max_norm, n);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_adamw_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, float* ema, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int step, float ema_decay, const double* sumsq,
                             float max_norm, int64_t n, void* stream) {
  DY_CHECK(p && g && exp_avg && exp_avg_sq && n >= 0 && step >= 1, "dy_adamw_step: bad args");
  if (n == 0) return 0;
  float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  adamw_kernel<<<ew_blocks(n), 256, 0, (hipStream_t)stream>>>(p, g, exp_avg, exp_avg_sq, ema, lr, beta1, beta2, eps, weight_decay,
                                                              bc1, bc2, ema_decay, sumsq, max_norm, n);
  DY_LAUNCH_CHECK();
  return 0;
}
