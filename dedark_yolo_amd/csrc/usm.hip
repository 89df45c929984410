// UsmFilter of the low-light front-end (reference ultralytics/nn/modules/filtersB.py:144-175): 25x25 gaussian (sigma 5) on a
// reflect-padded image, out = (img - blur) * lambda + img, forward and backward.  fp32 math.
//
// The reference runs three dense 625-tap conv2d calls; here the blur is separable (2 x 25 taps) and lives in LDS:
//   stage  (TH+24) x (TW+24) reflected tile of one channel, stored TRANSPOSED (tileT[col][row]) so that
//   hpass  lane = tile row, each thread produces 4 adjacent columns from a 28-value register window (7 LDS reads per output
//          instead of 25), conflict-free because consecutive lanes read consecutive rows of one column; result -> tmp[row][col]
//   vpass  lane = column, each thread produces 4 adjacent rows from a 28-value window of tmp, fused with the unsharp combine.
//
// Backward: for a symmetric kernel, A = (reflect-pad + blur) and its adjoint differ only near the image edges:
//   (A^T g)[m] = (A g)[m] + k[m] g[0]   for 1 <= m <= 12,      (A^T g)[0] = (A g)[0] - sum_{i=1..12} k[i] g[i]   (same at n-1)
// (the mirrored copies of g that reflect padding adds are the adjoint's fold-back terms except for the edge sample itself).
// So the backward pass is the forward machinery plus a wave-uniform correction on the 13 border rows / columns; the first
// version of this kernel evaluated per-lane adjoint weights and took 1.3 ms against 0.33 ms for the forward pass.
#include "dy_common.h"
#include "../../include/dedark_yolo.h"

namespace {

constexpr int R = 12;                       // gaussian radius
constexpr int TH = 40, TW = 64;             // output tile
constexpr int LH = TH + 2 * R, LW = TW + 2 * R;   // 64 x 88 staged
constexpr int PT = LH + 1;                  // tileT pitch (floats), odd -> conflict-free transposed stores
constexpr int PM = TW + 1;                  // tmp pitch
constexpr int NTH = 256;
static_assert(LH == 64, "hpass maps one lane to one staged row");

__constant__ float c_taps[R + 1];           // k[|d|], normalised (filtersB.py:152-161)

__device__ inline int reflect(int i, int n) {
  i = i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i);
  return min(max(i, 0), n - 1);               // only reached by tiles hanging over the image edge (masked at the store)
}

// (A^T g)[m] - (A g)[m] along one axis for an output sample m within R of an image edge.  `line` points at this lane's
// element of staged sample 0 (image coordinate o0 - R), consecutive samples are `pitch` floats apart.
//   m = 0:            - sum_{i=1..R} k[i] g[i]          1 <= m <= R:          + k[m] g[0]
//   m = n-1:          - sum_{i=1..R} k[i] g[n-1-i]      n-1-R <= m <= n-2:    + k[n-1-m] g[n-1]
// (not inlined: inlining it four times per pass pushed usm_bwd to 224 VGPRs / scratch spills that tripled its HBM traffic)
__device__ __noinline__ float adj_edge(const float* line, int pitch, int m, int o0, int n) {
  float c = 0.f;
  if (m == 0) {
    for (int i = 1; i <= R; ++i) c -= c_taps[i] * line[(i - o0 + R) * pitch];
  } else if (m <= R) {
    c += c_taps[m] * line[(0 - o0 + R) * pitch];
  }
  if (m == n - 1) {
    for (int i = 1; i <= R; ++i) c -= c_taps[i] * line[(n - 1 - i - o0 + R) * pitch];
  } else if (m >= n - 1 - R && m <= n - 2) {
    c += c_taps[n - 1 - m] * line[(n - 1 - o0 + R) * pitch];
  }
  return c;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// The 25-tap sums run on PAIRS of independent groups held as float2 (v_pk_add_f32 / v_pk_fma_f32: two results per VALU
// instruction); every element still sees the same sequence k0*w[R], += k[d]*(w[R-d] + w[R+d]) as the scalar code, so the
// results are bit-identical.  A pair shares nothing but the instruction stream: .x is group g, .y is group g2.
__device__ inline void taps4x2(const f32x2 (&win)[28], f32x2 (&acc)[4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f32x2 a = c_taps[0] * win[e + R];
#pragma unroll
    for (int d = 1; d <= R; ++d) a += c_taps[d] * (win[e + R - d] + win[e + R + d]);
    acc[e] = a;
  }
}

// horizontal pass over all 64 staged rows; ADJ: add the adjoint's edge terms for the columns within R of the image border
template <bool ADJ>
__device__ inline void hpass(const float* __restrict__ tileT, float* __restrict__ tmp, int x0, int W) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NWV = NTH / 64;
  static_assert((TW / 4) % (2 * NWV) == 0, "column groups pair up evenly");
  for (int g = wave; g < TW / 4; g += 2 * NWV) {
    const int g2 = g + NWV;
    f32x2 win[28];
#pragma unroll
    for (int j = 0; j < 28; ++j) {
      win[j].x = tileT[(4 * g + j) * PT + lane];
      win[j].y = tileT[(4 * g2 + j) * PT + lane];
    }
    f32x2 acc[4];
    taps4x2(win, acc);
    if (ADJ) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = x0 + 4 * g + e, m2 = x0 + 4 * g2 + e;      // wave-uniform
        if (m < W && (m <= R || m >= W - 1 - R)) acc[e].x += adj_edge(tileT + lane, PT, m, x0, W);
        if (m2 < W && (m2 <= R || m2 >= W - 1 - R)) acc[e].y += adj_edge(tileT + lane, PT, m2, x0, W);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      tmp[lane * PM + 4 * g + e] = acc[e].x;
      tmp[lane * PM + 4 * g2 + e] = acc[e].y;
    }
  }
}

// vertical pass for the 4 output rows of group G at column `lane`; returns the blurred values
template <bool ADJ>
__device__ inline void vpass4(const float* __restrict__ tmp, int G, int lane, int y0, int H, float (&acc)[4]) {
  float win[28];
#pragma unroll
  for (int j = 0; j < 28; ++j) win[j] = tmp[(4 * G + j) * PM + lane];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float a = c_taps[0] * win[e + R];
#pragma unroll
    for (int d = 1; d <= R; ++d) a += c_taps[d] * (win[e + R - d] + win[e + R + d]);
    acc[e] = a;
  }
  if (ADJ) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = y0 + 4 * G + e;                   // wave-uniform
      if (m < H && (m <= R || m >= H - 1 - R)) acc[e] += adj_edge(tmp + lane, PM, m, y0, H);
    }
  }
}

// the same for two row groups at once (.x = G, .y = G2)
template <bool ADJ>
__device__ inline void vpass4x2(const float* __restrict__ tmp, int G, int G2, int lane, int y0, int H, f32x2 (&acc)[4]) {
  f32x2 win[28];
#pragma unroll
  for (int j = 0; j < 28; ++j) {
    win[j].x = tmp[(4 * G + j) * PM + lane];
    win[j].y = tmp[(4 * G2 + j) * PM + lane];
  }
  taps4x2(win, acc);
  if (ADJ) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = y0 + 4 * G + e, m2 = y0 + 4 * G2 + e;          // wave-uniform
      if (m < H && (m <= R || m >= H - 1 - R)) acc[e].x += adj_edge(tmp + lane, PM, m, y0, H);
      if (m2 < H && (m2 <= R || m2 >= H - 1 - R)) acc[e].y += adj_edge(tmp + lane, PM, m2, y0, H);
    }
  }
}

// Row groups of the vertical pass: wave w owns w and w + 4 (a pair) and, for w < NG - 8, w + 8 (alone).
template <bool ADJ>
__device__ inline void vpass_all(const float* __restrict__ tmp, int wave, int lane, int y0, int H, float (&a)[3][4]) {
  constexpr int NG = TH / 4;
  static_assert(NG > 7 && NG <= 12, "two full groups per wave, at most one more");
  f32x2 p[4];
  vpass4x2<ADJ>(tmp, wave, wave + 4, lane, y0, H, p);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    a[0][e] = p[e].x;
    a[1][e] = p[e].y;
    a[2][e] = 0.f;
  }
  if (wave + 8 < NG) vpass4<ADJ>(tmp, wave + 8, lane, y0, H, a[2]);
}

// Stages the (TH+24) x (TW+24) reflected tile of one PLANAR plane (H x W, f32 or the compute dtype) into tileT (transposed):
// 4 columns per step with one 16- / 8-byte load where the chunk lies inside the image (x0 - R is a multiple of 4), element-wise
// reflection on the border chunks.  (The first version computed a div / mod and two reflections per ELEMENT: more VALU work
// than the two blur passes together.)
template <typename S>
__device__ inline void stage_planar(const S* __restrict__ pl, float* __restrict__ tileT, int y0, int x0, int H, int W) {
  constexpr int CH = LW / 4;
  static_assert(LW % 4 == 0 && R % 4 == 0 && TW % 4 == 0, "4-column chunks");
  const bool vec_ok = (W & 3) == 0;
  for (int i = threadIdx.x; i < LH * CH; i += NTH) {
    const int r = i / CH, cq = i - r * CH;
    const int yy = reflect(y0 + r - R, H);
    const int gx = x0 - R + 4 * cq;
    const S* row = pl + (long)yy * W;
    float v[4];
    if (vec_ok && gx >= 0 && gx + 3 < W) {
      if constexpr (sizeof(S) == 4) {
        const float4 t = *reinterpret_cast<const float4*>(row + gx);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      } else {
        const uint2 t = *reinterpret_cast<const uint2*>(row + gx);
        if constexpr (__is_same(S, f16_t)) {
          v[0] = cvt32<f16_t>((uint16_t)t.x); v[1] = cvt32<f16_t>((uint16_t)(t.x >> 16));
          v[2] = cvt32<f16_t>((uint16_t)t.y); v[3] = cvt32<f16_t>((uint16_t)(t.y >> 16));
        } else {
          v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
          v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = DT<S>::ld(row + reflect(gx + j, W));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) tileT[(4 * cq + j) * PT + r] = v[j];
  }
}

template <typename T>
__global__ __launch_bounds__(NTH, 4) void usm_fwd_kernel(const float* __restrict__ s4, const float* __restrict__ params,
                                                       float* __restrict__ out, T* __restrict__ out8, float* __restrict__ hp, int B,
                                                       int H, int W) {
  __shared__ float tileT[LW * PT];
  __shared__ float tmp[LH * PM];
  const int b = blockIdx.z, y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float lam = params[b * 8 + 6];
  constexpr int NG = TH / 4;                     // 10 row groups, wave w owns w, w+4, w+8
  float res[3][3][4];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* pl = s4 + ((long)b * 3 + c) * H * W;
    __syncthreads();                             // previous channel's vpass still reads tileT / tmp
    stage_planar<float>(pl, tileT, y0, x0, H, W);
    __syncthreads();
    hpass<false>(tileT, tmp, x0, W);
    __syncthreads();
    float a[3][4];
    vpass_all<false>(tmp, wave, lane, y0, H, a);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int G = wave + 4 * k;
      if (G < NG) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int t = 4 * G + e;
          const float v = tileT[(lane + R) * PT + t + R];
          const float hi = v - a[k][e];
          const float o = hi * lam + v;
          res[c][k][e] = o;
          const int yy = y0 + t, xx = x0 + lane;
          if (yy < H && xx < W) {
            const long idx = (((long)b * 3 + c) * H + yy) * W + xx;
            if (out) out[idx] = o;
            if (hp) hp[idx] = hi;
          }
        }
      }
    }
  }
  if (out8) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int G = wave + 4 * k;
      if (G < NG) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int yy = y0 + 4 * G + e, xx = x0 + lane;
          if (yy < H && xx < W) {
            float v[8] = {res[0][k][e], res[1][k][e], res[2][k][e], 0.f, 0.f, 0.f, 0.f, 0.f};
            T* o = out8 + (((long)b * H + yy) * W + xx) * 8;
            if constexpr (sizeof(T) == 4) {
              stvec<T>(o, v);
              stvec<T>(o + 4, v + 4);
            } else {
              stvec<T>(o, v);
            }
          }
        }
      }
    }
  }
}

// (3 blocks per CU: the pair windows plus the adjoint's edge calls need ~150 VGPRs; at the 128 of 4 blocks per CU the kernel
//  spilled 15 VGPRs + 64 SGPRs and went from 394 to 493 us)
template <typename T>
__global__ __launch_bounds__(NTH, 3) void usm_bwd_kernel(const float* __restrict__ dout, const T* __restrict__ dout8, int ld8,
                                                       const float* __restrict__ hp, const float* __restrict__ params,
                                                       float* __restrict__ ds4, double* dparams, int B, int H, int W) {
  __shared__ float tileT[LW * PT];
  __shared__ float tmp[LH * PM];
  __shared__ float sm[20];
  constexpr int VE = DT<T>::VE;
  const int b = blockIdx.z, y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float lam = params[b * 8 + 6];
  constexpr int NG = TH / 4;
  float dl = 0.f;
  for (int c = 0; c < 3; ++c) {
    __syncthreads();
    // the gradient of one channel, reflected like the forward input (the neighbouring blocks' copies come from L2)
    if (dout) {
      stage_planar<float>(dout + ((long)b * 3 + c) * H * W, tileT, y0, x0, H, W);
    } else if (ld8 == 0) {                                     // planar [B,3,H,W] in the compute dtype (direct stem dgrad)
      stage_planar<T>(dout8 + ((long)b * 3 + c) * H * W, tileT, y0, x0, H, W);
    } else {
      for (int i = tid; i < LH * LW; i += NTH) {
        const int r = i / LW, q = i - r * LW;
        const int yy = reflect(y0 + r - R, H), xx = reflect(x0 + q - R, W);
        float v;
        if (ld8 == VE) {
          float t[VE];
          ldvec<T>(dout8 + (((long)b * H + yy) * W + xx) * VE, t);
          v = c == 0 ? t[0] : (c == 1 ? t[1] : t[2]);
        } else {
          v = DT<T>::ld(dout8 + (((long)b * H + yy) * W + xx) * ld8 + c);
        }
        tileT[q * PT + r] = v;
      }
    }
    __syncthreads();
    hpass<true>(tileT, tmp, x0, W);
    __syncthreads();
    float a[3][4];
    vpass_all<true>(tmp, wave, lane, y0, H, a);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int G = wave + 4 * k;
      if (G < NG) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int t = 4 * G + e;
          const int yy = y0 + t, xx = x0 + lane;
          if (yy < H && xx < W) {
            const float g = tileT[(lane + R) * PT + t + R];
            const long idx = (((long)b * 3 + c) * H + yy) * W + xx;
            ds4[idx] = g * (1.f + lam) - lam * a[k][e];
            dl += g * hp[idx];
          }
        }
      }
    }
  }
  dl = block_sum(dl, sm);
  if (tid == 0) atomic_add_f64(dparams + b * 8 + 6, (double)dl);      // f64: order-free sum of the block partials (see pointwise_bwd_kernel)
}

bool g_taps_ready = false;
int ensure_taps() {
  if (g_taps_ready) return 0;
  float k[2 * R + 1];
  float sum = 0.f;
  for (int i = -R; i <= R; ++i) {
    float xv = (float)i / 5.0f;
    k[i + R] = expf(-0.5f * (xv * xv));
    sum += k[i + R];
  }
  float taps[R + 1];
  for (int d = 0; d <= R; ++d) taps[d] = k[R + d] / sum;
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_taps), taps, sizeof(taps));
  if (e != hipSuccess) {
    dy_set_error("usm: hipMemcpyToSymbol failed: %s", hipGetErrorString(e));
    return 3;
  }
  g_taps_ready = true;
  return 0;
}

}  // namespace

extern "C" int dy_frontend_init(void) { return ensure_taps(); }

extern "C" int dy_usm_fwd(const float* s4, const float* params, float* out_nchw, void* out_nhwc8, float* hp, int B, int H, int W,
                          int dtype, void* stream) {
  DY_CHECK(s4 && params && B > 0, "dy_usm_fwd: bad args");
  DY_CHECK(H > R && W > R, "dy_usm_fwd: reflect padding needs H, W > %d", R);
  if (int e = ensure_taps()) return e;
  dim3 grid(dy_cdiv(W, TW), dy_cdiv(H, TH), B);
  if (dtype == DY_F32) usm_fwd_kernel<float><<<grid, NTH, 0, (hipStream_t)stream>>>(s4, params, out_nchw, (float*)out_nhwc8, hp, B, H, W);
  else if ((dtype) == DY_F16) usm_fwd_kernel<f16_t><<<grid, NTH, 0, (hipStream_t)stream>>>(s4, params, out_nchw, (f16_t*)out_nhwc8, hp, B, H, W);
  else usm_fwd_kernel<bf16_t><<<grid, NTH, 0, (hipStream_t)stream>>>(s4, params, out_nchw, (bf16_t*)out_nhwc8, hp, B, H, W);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_usm_bwd(const float* dout_nchw, const void* dout_nhwc8, int dout_ld, const float* hp, const float* params,
                          float* ds4, double* dparams, int B, int H, int W, int dtype, void* stream) {
  DY_CHECK(dout_nhwc8 == nullptr || dout_ld >= 3 || dout_ld == 0, "dy_usm_bwd: bad dout_ld");
  DY_CHECK((dout_nchw != nullptr) != (dout_nhwc8 != nullptr), "dy_usm_bwd: exactly one of dout_nchw / dout_nhwc8");
  DY_CHECK(hp && params && ds4 && dparams && B > 0 && H > R && W > R, "dy_usm_bwd: bad args");
  if (int e = ensure_taps()) return e;
  dim3 grid(dy_cdiv(W, TW), dy_cdiv(H, TH), B);
  if (dtype == DY_F32)
    usm_bwd_kernel<float><<<grid, NTH, 0, (hipStream_t)stream>>>(dout_nchw, (const float*)dout_nhwc8, dout_ld, hp, params, ds4, dparams, B, H, W);
  else if ((dtype) == DY_F16)
    usm_bwd_kernel<f16_t><<<grid, NTH, 0, (hipStream_t)stream>>>(dout_nchw, (const f16_t*)dout_nhwc8, dout_ld, hp, params, ds4, dparams, B, H, W);
  else
    usm_bwd_kernel<bf16_t><<<grid, NTH, 0, (hipStream_t)stream>>>(dout_nchw, (const bf16_t*)dout_nhwc8, dout_ld, hp, params, ds4, dparams, B, H, W);
  DY_LAUNCH_CHECK();
  return 0;
}
