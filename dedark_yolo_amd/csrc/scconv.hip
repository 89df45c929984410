// The non-conv parts of SCConv (reference ultralytics/nn/modules/conv.py:323-440: GroupBatchnorm2d, SRU, CRU), the building block of
// MFRU (block.py:164-217, cfg/models/v8/yolov8-3.yaml).  The six convolutions of a CRU run on the conv kernels; here:
//   * per-image channel moments (sum x, sum x^2)                     -> group statistics of GroupBatchnorm2d, global average pool of CRU
//   * SRU: group norm (unbiased std, eps OUTSIDE the root) + gate sigmoid(gn * gamma / sum(gamma)) >= 0.5 + cross reconstruction
//   * CRU tail: softmax over the 2C pooled channels, scale, fold the halves
// and their backward passes.  Layout NHWC views (pointer + pixel stride), f32 / bf16 / f16; all sums in f32 per thread, a fixed-order
// LDS reduction over the rows of a block and f64 atomics across blocks.  One block = pixels of ONE image (blockIdx.y), so every sum
// is keyed (image, channel).  These kernels are HBM-bound elementwise / reduction passes (2 reads + 1 write or 2 reads).
#include "dy_common.h"
#include "../../include/dedark_yolo.h"

namespace {

constexpr int NT = 256;

// device-scope f64 add (global_atomic_add_f64)
__device__ inline void add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }

struct Lay {          // thread -> (vector slot, row) of a block; a slot is VE consecutive channels
  int slot, row, slots, rows;
};
__device__ inline Lay layout(int nslots) {
  Lay l;
  l.slots = nslots < NT ? nslots : NT;
  l.rows = NT / l.slots;
  l.slot = threadIdx.x % l.slots;
  l.row = threadIdx.x / l.slots;
  return l;
}

// acc[Q][VE] of every thread -> dst[(n * nkeys + key0 + e) * Q + q] over the block (rows added in a fixed order, then one f64 atomic)
template <int Q, int VE>
__device__ inline void block_sums_to_global(float (&acc)[Q][VE], const Lay& l, bool active, float* lds, double* dst, long key0_of_slot_stride,
                                            int key_base) {
  // lds: [rows][slots][Q][VE]
  if (active && l.row < l.rows) {
    float* mine = lds + ((l.row * l.slots + l.slot) * Q) * VE;
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int e = 0; e < VE; ++e) mine[q * VE + e] = acc[q][e];
  }
  __syncthreads();
  const int per_row = l.slots * Q * VE;
  for (int i = threadIdx.x; i < per_row; i += NT) {
    float s = 0.f;
    for (int r = 0; r < l.rows; ++r) s += lds[r * per_row + i];
    const int slot = i / (Q * VE), q = (i / VE) % Q, e = i % VE;
    add_f64(dst + ((long)(key_base + slot * key0_of_slot_stride + e)) * Q + q, (double)s);
  }
  __syncthreads();
}

// ---- channel moments: out[n][c][2] += (sum x, sum x^2) over the pixels of image n --------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void chan_moments_kernel(const T* __restrict__ x, long ld, long HW, int C, double* __restrict__ out) {
  constexpr int VE = DT<T>::VE;
  extern __shared__ float lds[];
  const int n = blockIdx.y, CV = C / VE;
  for (int s0 = 0; s0 < CV; s0 += NT) {                 // C <= NT * VE in practice: one pass
    const int nslots = (CV - s0) < NT ? (CV - s0) : NT;
    const Lay l = layout(nslots);
    const bool active = l.row < l.rows;
    float acc[2][VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
    if (active) {
      const T* base = x + (long)n * HW * ld + (s0 + l.slot) * VE;
      for (long p = (long)blockIdx.x * l.rows + l.row; p < HW; p += (long)gridDim.x * l.rows) {
        float v[VE];
        ldvec<T>(base + p * ld, v);
#pragma unroll
        for (int e = 0; e < VE; ++e) { acc[0][e] += v[e]; acc[1][e] += v[e] * v[e]; }
      }
    }
    block_sums_to_global<2, VE>(acc, l, active, lds, out, VE, n * C + s0 * VE);
  }
}

// group statistics of image n from the channel moments: mean, 1 / (std_unbiased + eps) per group; LDS arrays of G floats
__device__ inline void group_stats(const double* __restrict__ mom, int n, int C, int G, long HW, float eps, float* g_mean, float* g_inv,
                                   float* g_std) {
  const int cpg = C / G;
  for (int g = threadIdx.x; g < G; g += NT) {
    double s1 = 0.0, s2 = 0.0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      s1 += mom[((long)n * C + c) * 2];
      s2 += mom[((long)n * C + c) * 2 + 1];
    }
    const double cnt = (double)cpg * (double)HW;
    const double mean = s1 / cnt;
    double var = (s2 - cnt * mean * mean) / (cnt - 1.0);            // torch.std: unbiased
    if (var < 0.0) var = 0.0;
    const float sd = (float)sqrt(var);
    g_mean[g] = (float)mean;
    g_std[g] = sd;
    g_inv[g] = 1.f / (sd + eps);
  }
}

__device__ inline bool sru_gate(float gn, float wg) {             // sigmoid(gn * w_gamma) >= 0.5, evaluated the way torch does in f32
  const float t = gn * wg;
  return 1.f / (1.f + expf(-t)) >= 0.5f;
}

// ---- SRU forward: y[c] = (m[c] ? gn[c] : 0) + (m[c'] ? 0 : gn[c']),  c' = c +- C/2,  gn = (x - mean_g) / (std_g + eps) * gamma + beta ------
template <typename T>
__global__ __launch_bounds__(NT) void sru_fwd_kernel(const T* __restrict__ x, long x_ld, T* __restrict__ y, long y_ld, long HW, int C, int G,
                                                     const double* __restrict__ mom, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps) {
  constexpr int VE = DT<T>::VE;
  __shared__ float g_mean[64], g_inv[64], g_std[64];
  __shared__ float wsum_s;
  const int n = blockIdx.y;
  group_stats(mom, n, C, G, HW, eps, g_mean, g_inv, g_std);
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += gamma[c];
    wsum_s = s;
  }
  __syncthreads();
  const float wsum = wsum_s;
  const int half = C / 2, PV = half / VE, cpg = C / G;
  const long total = HW * PV;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const long p = i / PV;
    const int c = (int)(i - p * PV) * VE, c2 = c + half;
    const T* xp = x + ((long)n * HW + p) * x_ld;
    float a[VE], b[VE], oa[VE], ob[VE];
    ldvec<T>(xp + c, a);
    ldvec<T>(xp + c2, b);
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      const int ga = (c + e) / cpg, gb = (c2 + e) / cpg;
      const float gna = (a[e] - g_mean[ga]) * g_inv[ga] * gamma[c + e] + beta[c + e];
      const float gnb = (b[e] - g_mean[gb]) * g_inv[gb] * gamma[c2 + e] + beta[c2 + e];
      const bool ma = sru_gate(gna, gamma[c + e] / wsum), mb = sru_gate(gnb, gamma[c2 + e] / wsum);
      oa[e] = (ma ? gna : 0.f) + (mb ? 0.f : gnb);
      ob[e] = (mb ? gnb : 0.f) + (ma ? 0.f : gna);
    }
    T* yp = y + ((long)n * HW + p) * y_ld;
    stvec<T>(yp + c, oa);
    stvec<T>(yp + c2, ob);
  }
}

// ---- SRU backward, pass 1: red[n][c][2] += (sum dgn, sum dgn * xhat) with dgn[c] = m[c] ? dy[c] : dy[c'] -------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void sru_bwd_reduce_kernel(const T* __restrict__ x, long x_ld, const T* __restrict__ dy, long dy_ld, long HW,
                                                            int C, int G, const double* __restrict__ mom, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, double* __restrict__ red) {
  constexpr int VE = DT<T>::VE;
  extern __shared__ float lds[];
  __shared__ float g_mean[64], g_inv[64], g_std[64];
  __shared__ float wsum_s;
  const int n = blockIdx.y;
  group_stats(mom, n, C, G, HW, eps, g_mean, g_inv, g_std);
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += gamma[c];
    wsum_s = s;
  }
  __syncthreads();
  const float wsum = wsum_s;
  const int half = C / 2, PV = half / VE, cpg = C / G;
  for (int s0 = 0; s0 < PV; s0 += NT) {
    const int nslots = (PV - s0) < NT ? (PV - s0) : NT;
    const Lay l = layout(nslots);
    const bool active = l.row < l.rows;
    float acc[4][VE];                                   // (dgn, dgn*xhat) of c, then of c'
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < VE; ++e) acc[q][e] = 0.f;
    const int c = (s0 + l.slot) * VE, c2 = c + half;
    if (active) {
      for (long p = (long)blockIdx.x * l.rows + l.row; p < HW; p += (long)gridDim.x * l.rows) {
        const T* xp = x + ((long)n * HW + p) * x_ld;
        const T* dp = dy + ((long)n * HW + p) * dy_ld;
        float a[VE], b[VE], da[VE], db[VE];
        ldvec<T>(xp + c, a);
        ldvec<T>(xp + c2, b);
        ldvec<T>(dp + c, da);
        ldvec<T>(dp + c2, db);
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          const int ga = (c + e) / cpg, gb = (c2 + e) / cpg;
          const float xha = (a[e] - g_mean[ga]) * g_inv[ga], xhb = (b[e] - g_mean[gb]) * g_inv[gb];
          const float gna = xha * gamma[c + e] + beta[c + e], gnb = xhb * gamma[c2 + e] + beta[c2 + e];
          const bool ma = sru_gate(gna, gamma[c + e] / wsum), mb = sru_gate(gnb, gamma[c2 + e] / wsum);
          const float dga = ma ? da[e] : db[e], dgb = mb ? db[e] : da[e];
          acc[0][e] += dga;
          acc[1][e] += dga * xha;
          acc[2][e] += dgb;
          acc[3][e] += dgb * xhb;
        }
      }
    }
    // two key ranges (c and c'): write them as two 2-quantity reductions
    float lo[2][VE], hi[2][VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) { lo[0][e] = acc[0][e]; lo[1][e] = acc[1][e]; hi[0][e] = acc[2][e]; hi[1][e] = acc[3][e]; }
    block_sums_to_global<2, VE>(lo, l, active, lds, red, VE, n * C + s0 * VE);
    block_sums_to_global<2, VE>(hi, l, active, lds, red, VE, n * C + half + s0 * VE);
  }
}

// ---- SRU backward, pass 2: dx = (dxhat - mean_g(dxhat)) / (std + eps) - (x - mean) * S_g / ((std + eps)^2 (cnt - 1) std) ---------------
// (std = sqrt(sum (x - mean)^2 / (cnt - 1)): d std / d x_i = (x_i - mean) / ((cnt - 1) std); the eps sits outside the root)
// dxhat = dgn * gamma;  mean_g(dxhat) = sum_c gamma[c] red[n][c][0] / cnt;  S_g = sum_j dxhat_j (x_j - mean) = (std + eps) sum_c gamma[c] red[n][c][1]
template <typename T>
__global__ __launch_bounds__(NT) void sru_bwd_apply_kernel(const T* __restrict__ x, long x_ld, const T* __restrict__ dy, long dy_ld,
                                                           T* __restrict__ dx, long dx_ld, long HW, int C, int G,
                                                           const double* __restrict__ mom, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, const double* __restrict__ red) {
  constexpr int VE = DT<T>::VE;
  __shared__ float g_mean[64], g_inv[64], g_std[64], g_m1[64], g_k[64];
  __shared__ float wsum_s;
  const int n = blockIdx.y, cpg = C / G;
  group_stats(mom, n, C, G, HW, eps, g_mean, g_inv, g_std);
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += gamma[c];
    wsum_s = s;
  }
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += NT) {
    double s1 = 0.0, s2 = 0.0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
      s1 += (double)gamma[c] * red[((long)n * C + c) * 2];
      s2 += (double)gamma[c] * red[((long)n * C + c) * 2 + 1];
    }
    const double cnt = (double)cpg * (double)HW;
    g_m1[g] = (float)(s1 / cnt);
    // coefficient of xhat in dx: (x - mean) S_g / ((std+eps)^2 (cnt-1) std) with x - mean = xhat (std+eps), S_g = (std+eps) s2  ->  xhat s2 / ((cnt-1) std)
    const float sd = g_std[g];
    g_k[g] = sd > 0.f ? (float)(s2 / ((cnt - 1.0) * (double)sd)) : 0.f;
  }
  __syncthreads();
  const float wsum = wsum_s;
  const int half = C / 2, PV = half / VE;
  const long total = HW * PV;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const long p = i / PV;
    const int c = (int)(i - p * PV) * VE, c2 = c + half;
    const T* xp = x + ((long)n * HW + p) * x_ld;
    const T* dp = dy + ((long)n * HW + p) * dy_ld;
    float a[VE], b[VE], da[VE], db[VE], oa[VE], ob[VE];
    ldvec<T>(xp + c, a);
    ldvec<T>(xp + c2, b);
    ldvec<T>(dp + c, da);
    ldvec<T>(dp + c2, db);
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      const int ga = (c + e) / cpg, gb = (c2 + e) / cpg;
      const float xha = (a[e] - g_mean[ga]) * g_inv[ga], xhb = (b[e] - g_mean[gb]) * g_inv[gb];
      const float gna = xha * gamma[c + e] + beta[c + e], gnb = xhb * gamma[c2 + e] + beta[c2 + e];
      const bool ma = sru_gate(gna, gamma[c + e] / wsum), mb = sru_gate(gnb, gamma[c2 + e] / wsum);
      const float dxa = (ma ? da[e] : db[e]) * gamma[c + e], dxb = (mb ? db[e] : da[e]) * gamma[c2 + e];
      oa[e] = (dxa - g_m1[ga]) * g_inv[ga] - xha * g_k[ga];
      ob[e] = (dxb - g_m1[gb]) * g_inv[gb] - xhb * g_k[gb];
    }
    T* op = dx + ((long)n * HW + p) * dx_ld;
    stvec<T>(op + c, oa);
    stvec<T>(op + c2, ob);
  }
}

// softmax over the K = 2C pooled means of image n (mom holds sums; first moment only), into LDS s[K]
__device__ inline void pooled_softmax(const double* __restrict__ mom, int n, int K, long HW, float* s, float* scratch) {
  float mx = -INFINITY;
  for (int k = threadIdx.x; k < K; k += NT) {
    const float v = (float)(mom[((long)n * K + k) * 2] / (double)HW);
    s[k] = v;
    mx = fmaxf(mx, v);
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
  __syncthreads();
  float sum = 0.f;
  for (int k = threadIdx.x; k < K; k += NT) {
    const float e = expf(s[k] - mx);
    s[k] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = sum;
  __syncthreads();
  sum = scratch[0] + scratch[1] + scratch[2] + scratch[3];
  const float inv = 1.f / sum;
  for (int k = threadIdx.x; k < K; k += NT) s[k] *= inv;
  __syncthreads();
}

// ---- CRU tail forward: res[c] = o[c] * s[c] + o[c + C] * s[c + C],  s = softmax_k(mean_pixels o[k]) over the 2C channels ---------------
template <typename T>
__global__ __launch_bounds__(NT) void cru_fuse_fwd_kernel(const T* __restrict__ o, long o_ld, T* __restrict__ res, long r_ld, long HW, int C,
                                                          const double* __restrict__ mom) {
  constexpr int VE = DT<T>::VE;
  extern __shared__ float lds[];                         // s[2C]
  __shared__ float scratch[4];
  const int n = blockIdx.y;
  pooled_softmax(mom, n, 2 * C, HW, lds, scratch);
  const int PV = C / VE;
  const long total = HW * PV;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const long p = i / PV;
    const int c = (int)(i - p * PV) * VE;
    const T* op = o + ((long)n * HW + p) * o_ld;
    float a[VE], b[VE], r[VE];
    ldvec<T>(op + c, a);
    ldvec<T>(op + C + c, b);
#pragma unroll
    for (int e = 0; e < VE; ++e) r[e] = a[e] * lds[c + e] + b[e] * lds[C + c + e];
    stvec<T>(res + ((long)n * HW + p) * r_ld + c, r);
  }
}

// ---- CRU tail backward, pass 1: ds[n][k][0] += sum_pixels dres[c(k)] * o[k]  (second slot unused, kept for the shared reduction) --------
template <typename T>
__global__ __launch_bounds__(NT) void cru_fuse_bwd_reduce_kernel(const T* __restrict__ o, long o_ld, const T* __restrict__ dres, long d_ld,
                                                                 long HW, int C, double* __restrict__ ds) {
  constexpr int VE = DT<T>::VE;
  extern __shared__ float lds[];
  const int n = blockIdx.y, PV = C / VE;
  for (int s0 = 0; s0 < PV; s0 += NT) {
    const int nslots = (PV - s0) < NT ? (PV - s0) : NT;
    const Lay l = layout(nslots);
    const bool active = l.row < l.rows;
    float lo[2][VE], hi[2][VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) { lo[0][e] = lo[1][e] = hi[0][e] = hi[1][e] = 0.f; }
    const int c = (s0 + l.slot) * VE;
    if (active) {
      for (long p = (long)blockIdx.x * l.rows + l.row; p < HW; p += (long)gridDim.x * l.rows) {
        const T* op = o + ((long)n * HW + p) * o_ld;
        float a[VE], b[VE], d[VE];
        ldvec<T>(op + c, a);
        ldvec<T>(op + C + c, b);
        ldvec<T>(dres + ((long)n * HW + p) * d_ld + c, d);
#pragma unroll
        for (int e = 0; e < VE; ++e) { lo[0][e] += d[e] * a[e]; hi[0][e] += d[e] * b[e]; }
      }
    }
    block_sums_to_global<2, VE>(lo, l, active, lds, ds, VE, n * 2 * C + s0 * VE);
    block_sums_to_global<2, VE>(hi, l, active, lds, ds, VE, n * 2 * C + C + s0 * VE);
  }
}

// ---- CRU tail backward, pass 2: do[k] = dres[c(k)] * s[k] + s[k] (ds[k] - sum_j s[j] ds[j]) / HW -----------------------------------------
template <typename T>
__global__ __launch_bounds__(NT) void cru_fuse_bwd_apply_kernel(const T* __restrict__ dres, long d_ld, T* __restrict__ dout, long do_ld, long HW,
                                                                int C, const double* __restrict__ mom, const double* __restrict__ ds) {
  constexpr int VE = DT<T>::VE;
  extern __shared__ float lds[];                         // s[2C], then pool-gradient term t[2C]
  __shared__ float scratch[4];
  const int n = blockIdx.y, K = 2 * C;
  float* s = lds;
  float* t = lds + K;
  pooled_softmax(mom, n, K, HW, s, scratch);
  float dot = 0.f;
  for (int k = threadIdx.x; k < K; k += NT) dot += s[k] * (float)ds[((long)n * K + k) * 2];
  dot = wave_sum(dot);
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = dot;
  __syncthreads();
  dot = scratch[0] + scratch[1] + scratch[2] + scratch[3];
  for (int k = threadIdx.x; k < K; k += NT) t[k] = s[k] * ((float)ds[((long)n * K + k) * 2] - dot) / (float)HW;
  __syncthreads();
  const int PV = C / VE;
  const long total = HW * PV;
  for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
    const long p = i / PV;
    const int c = (int)(i - p * PV) * VE;
    float d[VE], a[VE], b[VE];
    ldvec<T>(dres + ((long)n * HW + p) * d_ld + c, d);
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      a[e] = d[e] * s[c + e] + t[c + e];
      b[e] = d[e] * s[C + c + e] + t[C + c + e];
    }
    T* op = dout + ((long)n * HW + p) * do_ld;
    stvec<T>(op + c, a);
    stvec<T>(op + C + c, b);
  }
}

int check(const char* who, const void* p, long ld, int C, int dtype) {
  const int ve = dtype == DY_F32 ? 4 : 8, es = dtype == DY_F32 ? 4 : 2;
  DY_CHECK(p != nullptr, "%s: null pointer", who);
  DY_CHECK(dtype == DY_F32 || dtype == DY_BF16 || dtype == DY_F16, "%s: bad dtype", who);
  DY_CHECK(C > 0 && C % ve == 0, "%s: C=%d must be a multiple of %d", who, C, ve);
  DY_CHECK(ld >= C && (ld * es) % 16 == 0 && ((uintptr_t)p) % 16 == 0, "%s: view not 16-byte aligned (ld=%ld)", who, ld);
  return 0;
}

inline dim3 grid_for(long HW, int N, int vectors_per_pixel) {
  long work = HW * vectors_per_pixel;
  long bx = (work + NT * 4 - 1) / (NT * 4);
  if (bx < 1) bx = 1;
  if (bx > 1024) bx = 1024;
  return dim3((unsigned)bx, (unsigned)N);
}

#define DY_SC_DISPATCH(dtype, ...)                                      \
  do {                                                                  \
    if ((dtype) == DY_F32) { using T = float; __VA_ARGS__; }            \
    else if ((dtype) == DY_F16) { using T = f16_t; __VA_ARGS__; }       \
    else { using T = bf16_t; __VA_ARGS__; }                             \
    DY_LAUNCH_CHECK();                                                  \
  } while (0)

}  // namespace

extern "C" int dy_chan_moments(const void* x, int64_t ld, int N, int64_t HW, int C, double* out, int dtype, void* stream) {
  if (int e = check("dy_chan_moments", x, ld, C, dtype)) return e;
  DY_CHECK(out && N > 0 && HW > 0, "dy_chan_moments: bad args");
  const int ve = dtype == DY_F32 ? 4 : 8;
  const dim3 g = grid_for(HW, N, C / ve);
  const size_t shm = (size_t)NT * 2 * ve * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  DY_SC_DISPATCH(dtype, chan_moments_kernel<T><<<g, NT, shm, st>>>((const T*)x, ld, HW, C, out));
  return 0;
}

extern "C" int dy_sru_fwd(const void* x, int64_t x_ld, void* y, int64_t y_ld, int N, int64_t HW, int C, int groups, const double* moments,
                          const float* gamma, const float* beta, float eps, int dtype, void* stream) {
  if (int e = check("dy_sru_fwd(x)", x, x_ld, C, dtype)) return e;
  if (int e = check("dy_sru_fwd(y)", y, y_ld, C, dtype)) return e;
  const int ve = dtype == DY_F32 ? 4 : 8;
  DY_CHECK(moments && gamma && beta && groups > 0 && groups <= 64 && C % groups == 0 && (C / 2) % ve == 0 && HW * (C / groups) > 1,
           "dy_sru_fwd: bad group layout");
  const dim3 g = grid_for(HW, N, C / 2 / ve);
  hipStream_t st = (hipStream_t)stream;
  DY_SC_DISPATCH(dtype, sru_fwd_kernel<T><<<g, NT, 0, st>>>((const T*)x, x_ld, (T*)y, y_ld, HW, C, groups, moments, gamma, beta, eps));
  return 0;
}

extern "C" int dy_sru_bwd(const void* x, int64_t x_ld, const void* dy, int64_t dy_ld, void* dx, int64_t dx_ld, int N, int64_t HW, int C,
                          int groups, const double* moments, const float* gamma, const float* beta, float eps, double* red, int dtype,
                          void* stream) {
  if (int e = check("dy_sru_bwd(x)", x, x_ld, C, dtype)) return e;
  if (int e = check("dy_sru_bwd(dy)", dy, dy_ld, C, dtype)) return e;
  if (int e = check("dy_sru_bwd(dx)", dx, dx_ld, C, dtype)) return e;
  const int ve = dtype == DY_F32 ? 4 : 8;
  DY_CHECK(moments && gamma && beta && red && groups > 0 && groups <= 64 && C % groups == 0 && (C / 2) % ve == 0, "dy_sru_bwd: bad args");
  const dim3 g = grid_for(HW, N, C / 2 / ve);
  const size_t shm = (size_t)NT * 2 * ve * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  DY_SC_DISPATCH(dtype, sru_bwd_reduce_kernel<T><<<g, NT, shm, st>>>((const T*)x, x_ld, (const T*)dy, dy_ld, HW, C, groups, moments, gamma,
                                                                    beta, eps, red));
  DY_SC_DISPATCH(dtype, sru_bwd_apply_kernel<T><<<g, NT, 0, st>>>((const T*)x, x_ld, (const T*)dy, dy_ld, (T*)dx, dx_ld, HW, C, groups, moments,
                                                                  gamma, beta, eps, red));
  return 0;
}

extern "C" int dy_cru_fuse_fwd(const void* o, int64_t o_ld, void* res, int64_t r_ld, int N, int64_t HW, int C, const double* moments, int dtype,
                               void* stream) {
  if (int e = check("dy_cru_fuse_fwd(o)", o, o_ld, 2 * C, dtype)) return e;
  if (int e = check("dy_cru_fuse_fwd(res)", res, r_ld, C, dtype)) return e;
  DY_CHECK(moments && C <= 2048, "dy_cru_fuse_fwd: bad args");
  const int ve = dtype == DY_F32 ? 4 : 8;
  const dim3 g = grid_for(HW, N, C / ve);
  const size_t shm = (size_t)2 * C * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  DY_SC_DISPATCH(dtype, cru_fuse_fwd_kernel<T><<<g, NT, shm, st>>>((const T*)o, o_ld, (T*)res, r_ld, HW, C, moments));
  return 0;
}

extern "C" int dy_cru_fuse_bwd(const void* o, int64_t o_ld, const void* dres, int64_t d_ld, void* dout, int64_t do_ld, int N, int64_t HW, int C,
                               const double* moments, double* ds, int dtype, void* stream) {
  if (int e = check("dy_cru_fuse_bwd(o)", o, o_ld, 2 * C, dtype)) return e;
  if (int e = check("dy_cru_fuse_bwd(dres)", dres, d_ld, C, dtype)) return e;
  if (int e = check("dy_cru_fuse_bwd(dout)", dout, do_ld, 2 * C, dtype)) return e;
  DY_CHECK(moments && ds && C <= 2048, "dy_cru_fuse_bwd: bad args");
  const int ve = dtype == DY_F32 ? 4 : 8;
  const dim3 g = grid_for(HW, N, C / ve);
  hipStream_t st = (hipStream_t)stream;
  const size_t shm_r = (size_t)NT * 2 * ve * sizeof(float), shm_a = (size_t)4 * C * sizeof(float);
  DY_SC_DISPATCH(dtype, cru_fuse_bwd_reduce_kernel<T><<<g, NT, shm_r, st>>>((const T*)o, o_ld, (const T*)dres, d_ld, HW, C, ds));
  DY_SC_DISPATCH(dtype, cru_fuse_bwd_apply_kernel<T><<<g, NT, shm_a, st>>>((const T*)dres, d_ld, (T*)dout, do_ld, HW, C, moments, ds));
  return 0;
}
