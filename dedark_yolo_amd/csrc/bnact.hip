// BatchNorm (training statistics) + activation, forward and backward, NHWC, vectorised 16 B per lane.
// Replaces nn.BatchNorm2d + SiLU / LeakyReLU(0.1) + residual add of the reference (ultralytics/nn/modules/conv.py:49-51,
// block.py:24-45,565; eps / momentum set at utils/torch_utils.py:263-265).  All kernels are HBM-bound streaming kernels.
//
// Thread mapping: a thread owns ONE 16-byte channel group for its whole life (per-channel constants live in registers) and
// walks pixels with a grid stride, two pixels in flight per iteration; a wave covers 64 consecutive channel groups of one
// pixel row (1 KiB contiguous when C >= 512 bf16) or several adjacent pixels for narrower tensors.
#include <stdlib.h>
#include <type_traits>
#include "dy_common.h"
#include "../../include/dedark_yolo.h"

// tensors beyond this many MiB are streamed with non-temporal loads / stores by the forward pass (tools/gpu: variant builds with
// -DDY_BN_NT_FWD_MB=n)
#ifndef DY_BN_NT_FWD_MB
#define DY_BN_NT_FWD_MB 128
#endif

namespace {

constexpr int NT = 256;
// NTS = non-temporal loads and stores: tensors beyond ~128 MB stream through the caches once (+8-12 % on 210-840 MB tensors, same
// box); smaller ones were written by the producing kernel a moment ago and are still in the Infinity Cache -- non-temporal reads of a
// 105 MB tensor are 9-13 % SLOWER (tools/gpu/bn_ab.sh, profiles/r03_bn_nontemporal_ab.txt)
template <typename T, bool NTS> __device__ inline void ld_s(const T* p, float* out) {
  if constexpr (NTS) ldvec_nt<T>(p, out); else ldvec<T>(p, out);
}
template <typename T, bool NTS> __device__ inline void st_s(T* p, const float* in) {
  if constexpr (NTS) stvec_nt<T>(p, in); else stvec<T>(p, in);
}

// one wave per channel: lane r sums replica r, then a wave reduction (the old one-thread-per-channel loop over the 64
// replicas was a chain of 128 dependent-latency loads: 10 us for a 64-channel layer)
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ stats, double count, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* running_mean, float* running_var,
                                                           float momentum, float eps, float* scale, float* shift, float* mean_out,
                                                           float* invstd_out, int C) {
  static_assert(DY_STATS_REPLICAS == 64, "one replica per lane");
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= C) return;
  double s1 = stats[(long)lane * 2 * C + c];
  double s2 = stats[(long)lane * 2 * C + C + c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if (lane != 0) return;
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

// v_exp_f32 / v_rcp_f32 based sigmoid (about 1e-6 relative error): the IEEE expf + division sequence made the SiLU
// kernels ALU-bound (4.1 TB/s against 5.9 TB/s for the LeakyReLU variant of the same kernel).
__device__ inline float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// The f32 instantiation is the parity path and keeps the IEEE expf / division of dy_common.h (a 1e-6 relative sigmoid error
// is amplified past the 1e-4 loss-item bound by the 126 small-batch BatchNorm layers of the repo-L golden case).
template <typename T, int ACT> __device__ inline float act_f(float u) {
  if (sizeof(T) == 4) return dy_act(ACT, u);
  if (ACT == DY_ACT_SILU) return u * fast_sigmoid(u);
  if (ACT == DY_ACT_LEAKY) return u > 0.f ? u : 0.1f * u;
  return u;
}
template <typename T, int ACT> __device__ inline float dact_f(float u) {
  if (sizeof(T) == 4) return dy_dact(ACT, u);
  if (ACT == DY_ACT_SILU) {
    const float s = fast_sigmoid(u);
    return s * (1.0f + u * (1.0f - s));
  }
  if (ACT == DY_ACT_LEAKY) return u > 0.f ? 1.0f : 0.1f;
  return 1.0f;
}
// The activation is a launch parameter, the element loops want it at compile time: with a run-time switch inside them every ELEMENT went
// through four scalar branches and its exp -> add -> rcp -> mul chain ran alone in its basic block (s_nop between the transcendentals, no
// second element to overlap with).  The loop is instantiated once per activation inside the kernel and picked in front of it.
template <typename F> __device__ inline void with_act(int act, F&& f) {
  if (act == DY_ACT_SILU) f(std::integral_constant<int, DY_ACT_SILU>{});
  else if (act == DY_ACT_LEAKY) f(std::integral_constant<int, DY_ACT_LEAKY>{});
  else f(std::integral_constant<int, DY_ACT_NONE>{});
}

struct Map {           // thread -> (channel group, first pixel, pixel stride)
  int c, cl;           // first channel of the group (global / within the block's channel range)
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
  m.cl = cg_local * VE;
  m.first = (long)blockIdx.x * rows + prow;
  m.step = (long)gridDim.x * rows;
  return m;
}


// Per-channel constants are staged once per block in LDS (coalesced loads by the first threads) and copied to registers.
// U = pixels in flight per thread.  More is not better: the 7 per-channel constants x VE already take 56 VGPRs in the backward
// kernels, and occupancy beats per-thread ILP for these streams (bn_bench: apply on 64ch x 6.5 M px 531 us at U=4, 487 us at U=1).
template <typename T, int U, bool NTS>
__global__ __launch_bounds__(NT) void bn_act_fwd_kernel(const T* __restrict__ z, long z_ld, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int act, const T* __restrict__ res,
                                                         long res_ld, T* __restrict__ y, long y_ld, long pixels, int C, int cgb,
                                                         int rows) {
  constexpr int VE = DT<T>::VE;
  extern __shared__ float lds[];               // [2][nch]
  const int nch = cgb * VE, c0 = blockIdx.y * nch;
  for (int i = threadIdx.x; i < nch; i += NT) {
    const int c = c0 + i;
    lds[i] = (scale && c < C) ? scale[c] : 1.f;
    lds[nch + i] = (shift && c < C) ? shift[c] : 0.f;
  }
  __syncthreads();
  const Map m = make_map<VE>(C, cgb, rows);
  if (!m.active) return;
  float sc[VE], sh[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    sc[e] = lds[m.cl + e];
    sh[e] = lds[nch + m.cl + e];
  }
  with_act(act, [&](auto actc) {
    constexpr int ACT = decltype(actc)::value;
    auto loop = [&](auto resc) {
      constexpr bool RES = decltype(resc)::value;
      for (long p0 = m.first; p0 < pixels; p0 += U * m.step) {
        float v[U][VE], r[U][VE];
#pragma unroll
        for (int k = 0; k < U; ++k) {
          const long p = p0 + k * m.step;
          if (p < pixels) {
            ld_s<T, NTS>(z + p * z_ld + m.c, v[k]);
            if (RES) ld_s<T, NTS>(res + p * res_ld + m.c, r[k]);
          }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
          const long p = p0 + k * m.step;
          if (p < pixels) {
#pragma unroll
            for (int e = 0; e < VE; ++e) {
              const float o = act_f<T, ACT>(v[k][e] * sc[e] + sh[e]);
              v[k][e] = RES ? o + r[k][e] : o;
            }
            st_s<T, NTS>(y + p * y_ld + m.c, v[k]);
          }
        }
      }
    };
    if (res) loop(std::true_type{});
    else loop(std::false_type{});
  });
}

// backward pass 1: per-channel sums of g and g*zhat
template <typename T, int U, bool NTS>
__global__ __launch_bounds__(NT) void bn_act_bwd_reduce_kernel(const T* __restrict__ dy, long dy_ld, const T* __restrict__ z,
                                                                long z_ld, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, int act, int has_bn,
                                                                double* sums, long pixels, int C, int cgb, int rows) {
  constexpr int VE = DT<T>::VE;
  extern __shared__ float lds[];               // constants [4][nch], then the per-thread partial sums [2][rows][nch]
  const int tid = threadIdx.x;
  const int nch = cgb * VE, c0 = blockIdx.y * nch;
  float* part = lds + 4 * nch;
  for (int i = tid; i < nch; i += NT) {
    const int c = c0 + i;
    const bool ok = c < C;
    lds[i] = (scale && ok) ? scale[c] : 1.f;
    lds[nch + i] = (shift && ok) ? shift[c] : 0.f;
    lds[2 * nch + i] = (has_bn && ok) ? mean[c] : 0.f;
    lds[3 * nch + i] = (has_bn && ok) ? invstd[c] : 0.f;
  }
  __syncthreads();
  const Map m = make_map<VE>(C, cgb, rows);
  if (m.active) {
    float sc[VE], sh[VE], mu[VE], is[VE], s1[VE], s2[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      sc[e] = lds[m.cl + e];
      sh[e] = lds[nch + m.cl + e];
      mu[e] = lds[2 * nch + m.cl + e];
      is[e] = lds[3 * nch + m.cl + e];
      s1[e] = 0.f;
      s2[e] = 0.f;
    }
    with_act(act, [&](auto actc) {
      constexpr int ACT = decltype(actc)::value;
      for (long p0 = m.first; p0 < pixels; p0 += U * m.step) {
        float g[U][VE], zz[U][VE];
#pragma unroll
        for (int k = 0; k < U; ++k) {
          const long p = p0 + k * m.step;
          if (p < pixels) {
            ld_s<T, NTS>(dy + p * dy_ld + m.c, g[k]);
            ld_s<T, NTS>(z + p * z_ld + m.c, zz[k]);
          }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
          const long p = p0 + k * m.step;
          if (p < pixels) {
#pragma unroll
            for (int e = 0; e < VE; ++e) {
              const float ge = g[k][e] * dact_f<T, ACT>(zz[k][e] * sc[e] + sh[e]);
              s1[e] += ge;
              s2[e] += ge * ((zz[k][e] - mu[e]) * is[e]);
            }
          }
        }
      }
    });
    // block reduction without atomics (the LDS float atomics serialised up to `rows` deep on every channel and made the order of
    // the additions, hence the last bits of the sums, depend on wave arrival): each thread parks its VE partial sums, the first
    // 2 * nch threads then add the `rows` values of one channel in a fixed order
    float* mine = part + (threadIdx.x / cgb) * nch + m.cl;
#pragma unroll
    for (int e = 0; e < VE; e += 4) {
      *reinterpret_cast<f32x4*>(mine + e) = f32x4{s1[e], s1[e + 1], s1[e + 2], s1[e + 3]};
      *reinterpret_cast<f32x4*>(mine + rows * nch + e) = f32x4{s2[e], s2[e + 1], s2[e + 2], s2[e + 3]};
    }
  }
  __syncthreads();
  double* dst = sums + (long)(blockIdx.x % DY_BN_BWD_REPLICAS) * 2 * C;
  const int CGl = C / VE - blockIdx.y * cgb;                      // channel groups of this block that exist
  for (int i = tid; i < 2 * nch; i += NT) {
    const int which = i >= nch, ch = i - which * nch;
    const int c = c0 + ch;
    if (c < C && ch / VE < CGl && (which == 0 || has_bn)) {
      const float* col = part + which * rows * nch + ch;
      float acc = 0.f;
      for (int r = 0; r < rows; ++r) acc += col[r * nch];
      atomic_add_f64(dst + which * C + c, (double)acc);
    }
  }
}

// backward pass 2: dz = k1*g - (K2*zhat + K3) with k1 = gamma*invstd, K2 = k1*sum(g*zhat)/M, K3 = k1*sum(g)/M
template <typename T, int U, bool NTS>
__global__ __launch_bounds__(NT) void bn_act_bwd_apply_kernel(const T* __restrict__ dy, long dy_ld, const T* __restrict__ z,
                                                               long z_ld, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                               int act, int has_bn, const double* __restrict__ sums,
                                                               T* __restrict__ dz, long dz_ld, float* dgamma, float* dbeta,
                                                               long pixels, long count, int C, int cgb, int rows) {
  constexpr int VE = DT<T>::VE;
  extern __shared__ float lds[];               // [7][nch]: sc, sh, mu, is, k1, K2, K3
  const int nch = cgb * VE, c0 = blockIdx.y * nch;
  const float invM = 1.f / (float)count;
  for (int i = threadIdx.x; i < nch; i += NT) {
    const int c = c0 + i;
    const bool ok = c < C;
    double t1 = 0.0, t2 = 0.0;
    if (ok) {
#pragma unroll
      for (int r = 0; r < DY_BN_BWD_REPLICAS; ++r) {
        t1 += sums[(long)r * 2 * C + c];
        if (has_bn) t2 += sums[(long)r * 2 * C + C + c];
      }
      if (blockIdx.x == 0) {
        if (dbeta) dbeta[c] = (float)t1;
        if (dgamma && has_bn) dgamma[c] = (float)t2;
      }
    }
    const float isv = (has_bn && ok) ? invstd[c] : 0.f;
    const float k1 = has_bn ? ((gamma && ok) ? gamma[c] : 1.f) * isv : 1.f;
    lds[i] = (scale && ok) ? scale[c] : 1.f;
    lds[nch + i] = (shift && ok) ? shift[c] : 0.f;
    lds[2 * nch + i] = (has_bn && ok) ? mean[c] : 0.f;
    lds[3 * nch + i] = isv;
    lds[4 * nch + i] = k1;
    lds[5 * nch + i] = has_bn ? k1 * ((float)t2 * invM) : 0.f;
    lds[6 * nch + i] = has_bn ? k1 * ((float)t1 * invM) : 0.f;
  }
  __syncthreads();
  const Map m = make_map<VE>(C, cgb, rows);
  if (!m.active) return;
  float sc[VE], sh[VE], mu[VE], is[VE], k1[VE], k2[VE], k3[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    sc[e] = lds[m.cl + e];
    sh[e] = lds[nch + m.cl + e];
    mu[e] = lds[2 * nch + m.cl + e];
    is[e] = lds[3 * nch + m.cl + e];
    k1[e] = lds[4 * nch + m.cl + e];
    k2[e] = lds[5 * nch + m.cl + e];
    k3[e] = lds[6 * nch + m.cl + e];
  }
  with_act(act, [&](auto actc) {
    constexpr int ACT = decltype(actc)::value;
    for (long p0 = m.first; p0 < pixels; p0 += U * m.step) {
      float g[U][VE], zz[U][VE];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const long p = p0 + k * m.step;
        if (p < pixels) {
          ld_s<T, NTS>(dy + p * dy_ld + m.c, g[k]);
          ld_s<T, NTS>(z + p * z_ld + m.c, zz[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const long p = p0 + k * m.step;
        if (p < pixels) {
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            const float ge = g[k][e] * dact_f<T, ACT>(zz[k][e] * sc[e] + sh[e]);
            g[k][e] = k1[e] * ge - (k2[e] * ((zz[k][e] - mu[e]) * is[e]) + k3[e]);
          }
          st_s<T, NTS>(dz + p * dz_ld + m.c, g[k]);
        }
      }
    }
  });
}

struct Geo { int cgb, rows; dim3 grid; };

// Grid policy (swept with tools/bn_bench on the YOLOv8-L shapes): every block pays a prologue proportional to the channels it
// covers (per-channel constants; the backward kernels also fold DY_BN_BWD_REPLICAS x 2 doubles per channel) and the reduce
// kernel ends with 2 atomics per channel, so wide tensors want FEWER, longer-running blocks: about 2^18 / C of them (2048 at
// C <= 128 ... 512 at C = 512), never fewer than 4 pixels per thread.  E.g. apply on 512ch x 102,400 px: 6,400 blocks 118 us,
// 1,024 blocks 82 us; reduce on 64ch x 6.5 M px: 1,024 blocks 429 us, 2,048 blocks 363 us.
Geo geometry(long pixels, int C, int ve, int kind /*0 fwd, 1 bwd reduce, 2 bwd apply*/) {
  Geo g;
  const int CG = C / ve;
  g.cgb = CG < NT ? CG : NT;
  g.rows = NT / g.cgb;
  static const int env_tgt = dy_env("DY_BN_BLOCKS") ? atoi(dy_env("DY_BN_BLOCKS")) : 0;       // tuning aid
  long target = (1L << 18) / (C > 0 ? C : 1);
  target = target > 2048 ? 2048 : (target < 256 ? 256 : target);
  const long bytes = pixels * C * (ve == 8 ? 2 : 4);
  // Backward kernels below ~128 MB per tensor (re-swept in round 2 with tools/bn_bench after the reduce kernel lost its LDS float
  // atomics, whose per-block tail had favoured few blocks): 1,024 blocks from 16 MB on (reduce 128ch x 409,600 px 44.9 -> 37.0 us,
  // 256ch x 102,400 px 25.9 -> 22.9, 512ch x 102,400 px 48.0 -> 39.5; apply 3-6 %), 512 below (reduce 128ch x 51,200 px 11.3 ->
  // 8.6 us against the former 256); 512-channel tensors up to 32 MB keep 512 (1,024 atomics per block: 18.4 vs 19.2 us).
  if (kind != 0 && bytes <= (128L << 20)) {
    const bool big = bytes > (16L << 20) && !(C >= 512 && bytes <= (32L << 20));
    target = big ? 1024 : 512;
  }
  if (env_tgt > 0) target = env_tgt;
  const int gy = dy_cdiv(CG, g.cgb);
  long gx = (target + gy - 1) / gy;
  const long cap = (pixels + 4L * g.rows - 1) / (4L * g.rows);          // at least 4 pixels per thread
  if (gx > cap) gx = cap;
  if (gx < 1) gx = 1;
  g.grid = dim3((unsigned)gx, gy);
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
  bn_finalize_kernel<<<dy_cdiv(C, 4), 256, 0, (hipStream_t)stream>>>(stats, (double)count, gamma, beta, running_mean,
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
  const int ve = dtype == DY_F32 ? 4 : 8;
  const Geo g = geometry(pixels, C, ve, 0);
  const size_t shm = 2 * (size_t)g.cgb * ve * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  const bool big = pixels * C * (dtype == DY_F32 ? 4 : 2) > ((long)DY_BN_NT_FWD_MB << 20);
  dy_note_kernel("bn_act_fwd_kernel");
#define FWD(T_, U_) bn_act_fwd_kernel<T_, U_, (U_ == 4)><<<g.grid, NT, shm, st>>>((const T_*)z, z_ld, scale, shift, act, (const T_*)residual, res_ld, \
                                                                    (T_*)y, y_ld, pixels, C, g.cgb, g.rows)
  if (dtype == DY_F32) { if (big) FWD(float, 4); else FWD(float, 2); }
  else if ((dtype) == DY_F16) { if (big) FWD(f16_t, 4); else FWD(f16_t, 2); }
  else { if (big) FWD(bf16_t, 4); else FWD(bf16_t, 2); }
#undef FWD
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
  const Geo g = geometry(pixels, C, ve, 1);
  size_t shm = (4 + 2 * (size_t)g.rows) * g.cgb * ve * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  dy_note_kernel("bn_act_bwd_reduce_kernel");
  const bool big = pixels * C * (dtype == DY_F32 ? 4 : 2) > (128L << 20);
#define REDUCE(T_, N_) bn_act_bwd_reduce_kernel<T_, 2, N_><<<g.grid, NT, shm, st>>>((const T_*)dy, dy_ld, (const T_*)z, z_ld, scale, shift, mean, invstd, \
                                                                                   act, has_bn, sums, pixels, C, g.cgb, g.rows)
  if (dtype == DY_F32) { if (big) REDUCE(float, true); else REDUCE(float, false); }
  else if (dtype == DY_F16) { if (big) REDUCE(f16_t, true); else REDUCE(f16_t, false); }
  else { if (big) REDUCE(bf16_t, true); else REDUCE(bf16_t, false); }
#undef REDUCE
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
  const int ve = dtype == DY_F32 ? 4 : 8;
  const Geo g = geometry(pixels > 0 ? pixels : 1, C, ve, 2);
  const size_t shm = 7 * (size_t)g.cgb * ve * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  const bool big = pixels * C * (dtype == DY_F32 ? 4 : 2) > (128L << 20);
  dy_note_kernel("bn_act_bwd_apply_kernel");
#define APPLY(T_, U_) bn_act_bwd_apply_kernel<T_, U_, (U_ == 1)><<<g.grid, NT, shm, st>>>((const T_*)dy, dy_ld, (const T_*)z, z_ld, scale, shift, mean, invstd, \
                                                                            gamma, act, has_bn, sums, (T_*)dz, dz_ld, dgamma, dbeta, pixels, \
                                                                            pixels > 0 ? pixels : 1, C, g.cgb, g.rows)
  if (dtype == DY_F32) { if (big) APPLY(float, 1); else APPLY(float, 2); }
  else if ((dtype) == DY_F16) { if (big) APPLY(f16_t, 1); else APPLY(f16_t, 2); }
  else { if (big) APPLY(bf16_t, 1); else APPLY(bf16_t, 2); }
#undef APPLY
  DY_LAUNCH_CHECK();
  return 0;
}

// both backward passes behind one call (aff = [scale | shift | mean | invstd], C floats each)
extern "C" int dy_bn_act_bwd(const void* dy, int64_t dy_ld, const void* z, int64_t z_ld, const float* aff, const float* gamma, int act,
                             double* sums, void* dz, int64_t dz_ld, float* dgamma, float* dbeta, int64_t pixels, int C, int dtype,
                             void* stream) {
  DY_CHECK(aff, "dy_bn_act_bwd: null affine buffer");
  if (int e = dy_bn_act_bwd_reduce(dy, dy_ld, z, z_ld, aff, aff + C, aff + 2 * C, aff + 3 * C, act, 1, sums, pixels, C, dtype, stream))
    return e;
  return dy_bn_act_bwd_apply(dy, dy_ld, z, z_ld, aff, aff + C, aff + 2 * C, aff + 3 * C, gamma, act, 1, sums, dz, dz_ld, dgamma, dbeta,
                             pixels, C, dtype, stream);
}
