// Fused optimizer step on ONE flat f32 parameter buffer: gradient clipping coefficient, SGD(nesterov)/AdamW update with
// per-element parameter-group hyper-parameters and the EMA lerp in one pass (reference: BaseTrainer.optimizer_step
// ultralytics/engine/trainer.py:459-467, build_optimizer :611-665, ModelEMA.update ultralytics/utils/torch_utils.py:360-371).
// Pure HBM streaming: 4-5 reads + 3 writes per element instead of ~230 x (5-8) small launches.
#include "dy_common.h"
#include "../../include/dedark_yolo.h"

namespace {

struct Hyp { float lr[4]; float wd[4]; };

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

// Coefficient applied to every gradient element: clip factor min(1, max_norm / (|g| + 1e-6)) on the TRUE norm, times 1 / loss_scale
// when the gradients were produced from a scaled loss (fp16, reference GradScaler: unscale_ -> clip -> step).  *skip = the scaled
// gradients hold an inf / NaN: the reference's scaler.step() then leaves parameters and optimizer state untouched.
__device__ inline float step_coef(const double* sumsq, float max_norm, const float* loss_scale, bool* skip) {
  const float inv = loss_scale ? 1.f / loss_scale[0] : 1.f;
  *skip = false;
  if (!sumsq) return inv;
  const double ss = *sumsq;
  if (loss_scale && !(ss < (double)INFINITY)) {
    *skip = true;
    return 0.f;
  }
  const float nrm = (float)sqrt(ss) * inv;
  const float c = max_norm / (nrm + 1e-6f);
  return (c < 1.f ? c : 1.f) * inv;
}

__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, float* __restrict__ ema,
                           const uint8_t* __restrict__ gid, Hyp h, float mom, int nesterov, float ed, const double* sumsq,
                           float max_norm, float gscale, const float* loss_scale, long n) {
  bool skip;
  const float cc = step_coef(sumsq, max_norm, loss_scale, &skip) * gscale;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    if (skip) {
      if (ema) ema[i] = ed * ema[i] + (1.f - ed) * p[i];
      continue;
    }
    const int k = gid ? (gid[i] & 3) : 0;
    float w = p[i];
    float d = g[i] * cc + h.wd[k] * w;
    float b = mom * buf[i] + d;
    buf[i] = b;
    d = nesterov ? d + mom * b : b;
    w -= h.lr[k] * d;
    p[i] = w;
    if (ema) ema[i] = ed * ema[i] + (1.f - ed) * w;
  }
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m1, float* __restrict__ m2,
                             float* __restrict__ ema, const uint8_t* __restrict__ gid, Hyp h, float b1, float b2, float eps,
                             int step, float ed, const double* sumsq, float max_norm, float gscale,
                             const float* loss_scale, long n) {
  bool skip;
  const float cc = step_coef(sumsq, max_norm, loss_scale, &skip) * gscale;
  // bias correction counts the steps the optimizer really TOOK: GradScaler.step does not call optimizer.step() after an overflow, so
  // torch's Adam `step` does not advance there; loss_scale[2] = overflowed steps so far (dy_loss_scale_update)
  const float eff = (float)(step - (loss_scale ? (int)loss_scale[2] : 0));
  const float bc1 = 1.f - powf(b1, eff), bc2 = 1.f - powf(b2, eff);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    if (skip) {
      if (ema) ema[i] = ed * ema[i] + (1.f - ed) * p[i];
      continue;
    }
    const int k = gid ? (gid[i] & 3) : 0;
    float w = p[i] * (1.f - h.lr[k] * h.wd[k]);
    float gi = g[i] * cc;
    float a = b1 * m1[i] + (1.f - b1) * gi;
    float v = b2 * m2[i] + (1.f - b2) * gi * gi;
    m1[i] = a;
    m2[i] = v;
    float denom = sqrtf(v) / sqrtf(bc2) + eps;
    w -= (h.lr[k] / bc1) * a / denom;
    p[i] = w;
    if (ema) ema[i] = ed * ema[i] + (1.f - ed) * w;
  }
}

// dynamic loss scale (torch.cuda.amp.GradScaler.update: growth 2, backoff 0.5, growth_interval 2000):
// st = {scale, consecutive finite steps, overflowed (skipped) steps in total}
__global__ void loss_scale_update_kernel(float* st, const double* sumsq, float growth, float backoff, int interval) {
  if (threadIdx.x || blockIdx.x) return;
  if (!(*sumsq < (double)INFINITY)) {
    st[0] *= backoff;
    st[1] = 0.f;
    st[2] += 1.f;
  } else {
    const float good = st[1] + 1.f;
    if (good >= (float)interval) {
      st[0] *= growth;
      st[1] = 0.f;
    } else {
      st[1] = good;
    }
  }
}

__global__ void ema_lerp_kernel(float* __restrict__ ema, const float* __restrict__ src, float d, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    ema[i] = d * ema[i] + (1.f - d) * src[i];
}

inline int ew_blocks(long n) {
  long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" int dy_sumsq(const float* g, int64_t n, double* acc, void* stream) {
  DY_CHECK(g && acc && n >= 0 && ((uintptr_t)g) % 16 == 0, "dy_sumsq: bad args");
  // every block ends with ONE f64 atomic on the same address (~8 ns each, serialised): 256 blocks, not 2048 (30 -> ~8 us at 3 M)
  int blocks = ew_blocks(n / 4 + 1);
  if (blocks > 256) blocks = 256;
  sumsq_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(g, n, acc);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_sgd_step_scaled(float* p, const float* g, float* mom_buf, float* ema, const uint8_t* group_id, float lr0, float lr1,
                                  float lr2, float wd0, float wd1, float wd2, float momentum, int nesterov, float ema_decay,
                                  const double* sumsq, float max_norm, float grad_scale, const float* loss_scale, int64_t n,
                                  void* stream) {
  DY_CHECK(p && g && mom_buf && n >= 0 && (!loss_scale || sumsq), "dy_sgd_step: bad args");
  if (n == 0) return 0;
  Hyp h = {{lr0, lr1, lr2, lr2}, {wd0, wd1, wd2, wd2}};
  sgd_kernel<<<ew_blocks(n), 256, 0, (hipStream_t)stream>>>(p, g, mom_buf, ema, group_id, h, momentum, nesterov, ema_decay, sumsq,
                                                            max_norm, grad_scale, loss_scale, n);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_sgd_step(float* p, const float* g, float* mom_buf, float* ema, const uint8_t* group_id, float lr0, float lr1,
                           float lr2, float wd0, float wd1, float wd2, float momentum, int nesterov, float ema_decay,
                           const double* sumsq, float max_norm, float grad_scale, int64_t n, void* stream) {
  return dy_sgd_step_scaled(p, g, mom_buf, ema, group_id, lr0, lr1, lr2, wd0, wd1, wd2, momentum, nesterov, ema_decay, sumsq, max_norm,
                            grad_scale, nullptr, n, stream);
}

extern "C" int dy_adamw_step_scaled(float* p, const float* g, float* exp_avg, float* exp_avg_sq, float* ema, const uint8_t* group_id,
                                    float lr0, float lr1, float lr2, float wd0, float wd1, float wd2, float beta1, float beta2,
                                    float eps, int step, float ema_decay, const double* sumsq, float max_norm, float grad_scale,
                                    const float* loss_scale, int64_t n, void* stream) {
  DY_CHECK(p && g && exp_avg && exp_avg_sq && n >= 0 && step >= 1 && (!loss_scale || sumsq), "dy_adamw_step: bad args");
  if (n == 0) return 0;
  Hyp h = {{lr0, lr1, lr2, lr2}, {wd0, wd1, wd2, wd2}};
  adamw_kernel<<<ew_blocks(n), 256, 0, (hipStream_t)stream>>>(p, g, exp_avg, exp_avg_sq, ema, group_id, h, beta1, beta2, eps, step,
                                                              ema_decay, sumsq, max_norm, grad_scale, loss_scale, n);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_adamw_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, float* ema, const uint8_t* group_id,
                             float lr0, float lr1, float lr2, float wd0, float wd1, float wd2, float beta1, float beta2, float eps,
                             int step, float ema_decay, const double* sumsq, float max_norm, float grad_scale, int64_t n,
                             void* stream) {
  return dy_adamw_step_scaled(p, g, exp_avg, exp_avg_sq, ema, group_id, lr0, lr1, lr2, wd0, wd1, wd2, beta1, beta2, eps, step,
                              ema_decay, sumsq, max_norm, grad_scale, nullptr, n, stream);
}

extern "C" int dy_loss_scale_update(float* state, const double* sumsq, float growth, float backoff, int interval, void* stream) {
  DY_CHECK(state && sumsq && growth >= 1.f && backoff > 0.f && backoff <= 1.f && interval >= 1, "dy_loss_scale_update: bad args");
  loss_scale_update_kernel<<<1, 64, 0, (hipStream_t)stream>>>(state, sumsq, growth, backoff, interval);
  DY_LAUNCH_CHECK();
  return 0;
}

namespace {
__global__ void grad_accumulate_kernel(float* __restrict__ acc, const float* __restrict__ g, long n) {
  const long n4 = n >> 2;
  f32x4* a4 = reinterpret_cast<f32x4*>(acc);
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) a4[i] = a4[i] + g4[i];
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) acc[(n4 << 2) + threadIdx.x] += g[(n4 << 2) + threadIdx.x];
}
}  // namespace

extern "C" int dy_grad_accumulate(float* acc, const float* g, int64_t n, void* stream) {
  DY_CHECK(acc && g && n >= 0 && ((uintptr_t)acc % 16 == 0) && ((uintptr_t)g % 16 == 0), "dy_grad_accumulate: bad args");
  if (n == 0) return 0;
  grad_accumulate_kernel<<<ew_blocks(n), 256, 0, (hipStream_t)stream>>>(acc, g, n);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_ema_lerp(float* ema, const float* src, float decay, int64_t n, void* stream) {
  DY_CHECK(ema && src && n >= 0, "dy_ema_lerp: bad args");
  if (n == 0) return 0;
  ema_lerp_kernel<<<ew_blocks(n), 256, 0, (hipStream_t)stream>>>(ema, src, decay, n);
  DY_LAUNCH_CHECK();
  return 0;
}
