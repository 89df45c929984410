// Common device/host helpers for the Dedark-YOLO gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define DY_F32 0
#define DY_BF16 1
#define DY_F16 2

// Diagnostics -- kernel ablations (`ablate` bits), s_memtime stamps and dispatch overrides read from the environment -- exist only
// in `make DIAG=1` builds (-DDY_DIAG): the shipped library compiles them out (the ablate field reads as the constant 0, so the
// branches and their registers disappear) and its dispatch does not depend on environment variables.  One test hook stays a real
// getenv: DY_CONV_THIN_ALL (routes every eligible shape to the thin-layer kernel; tests/test_gpu_conv_kernels.py).
#include <stdlib.h>
#ifdef DY_DIAG
#define DY_ABLATE_OF(p) ((p).ablate)
static inline const char* dy_env(const char* name) { return getenv(name); }
#else
#define DY_ABLATE_OF(p) 0
static inline const char* dy_env(const char*) { return nullptr; }
#endif

#define DY_ACT_NONE 0
#define DY_ACT_SILU 1
#define DY_ACT_LEAKY 2   // LeakyReLU(0.1)

typedef uint16_t bf16_t;   // raw bf16 bits
typedef _Float16 f16_t;    // IEEE half (the reference's AMP dtype, BASELINE configs[4])
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

// ---- error plumbing (C-ABI returns int, message via dy_last_error) -------------------------------------------
extern "C" const char* dy_last_error(void);
void dy_set_error(const char* fmt, ...);
void dy_note_kernel(const char* name);   // kernel symbol of the launch (error.cpp: dy_last_kernel)

#define DY_CHECK(cond, ...)                      \
  do {                                           \
    if (!(cond)) {                               \
      dy_set_error(__VA_ARGS__);                 \
      return 1;                                  \
    }                                            \
  } while (0)

#define DY_LAUNCH_CHECK()                                                    \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      dy_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return 2;                                                              \
    }                                                                        \
  } while (0)

static inline int dy_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- bf16 <-> f32 --------------------------------------------------------------------------------------------
__host__ __device__ inline float bf16_to_f32(bf16_t v) {
  union { uint32_t u; float f; } c;
  c.u = ((uint32_t)v) << 16;
  return c.f;
}
__device__ inline bf16_t f32_to_bf16(float f) {
  // plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950); round-to-nearest-even
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}

// One parity class of a stride-2 data gradient (conv.hip: dy_conv2d_dgrad): the conv kernels take up to four of them in one
// launch, block ranges [blk0, next blk0) select the class and the fields below replace the launch-wide ones.
struct DyParityCls {
  char* dst;
  long M;
  int Hd, Wd, KH, KW, pad, kh0, kw0, Ktot, blk0, _r;
};

template <typename T> struct DT;
template <> struct DT<float> {
  static constexpr int id = DY_F32;
  static constexpr int VE = 4;   // elements per 16-byte vector
  __device__ static inline float ld(const float* p) { return *p; }
  __device__ static inline void st(float* p, float v) { *p = v; }
};
template <> struct DT<bf16_t> {
  static constexpr int id = DY_BF16;
  static constexpr int VE = 8;
  __device__ static inline float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static inline void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

template <> struct DT<f16_t> {
  static constexpr int id = DY_F16;
  static constexpr int VE = 8;
  __device__ static inline float ld(const f16_t* p) { return (float)*p; }
  __device__ static inline void st(f16_t* p, float v) { *p = (f16_t)v; }
};

// f32 -> 16-bit storage bits of T / back (the pipelined kernels move 16-bit payloads as raw bits)
template <typename T> __device__ inline uint16_t cvt16(float f);
template <> __device__ inline uint16_t cvt16<bf16_t>(float f) { return f32_to_bf16(f); }
template <> __device__ inline uint16_t cvt16<f16_t>(float f) { return __builtin_bit_cast(uint16_t, (f16_t)f); }
template <typename T> __device__ inline float cvt32(uint16_t v);
template <> __device__ inline float cvt32<bf16_t>(uint16_t v) { return bf16_to_f32(v); }
template <> __device__ inline float cvt32<f16_t>(uint16_t v) { return (float)__builtin_bit_cast(f16_t, v); }

// the two MFMA shapes of the 16-bit kernels, by storage type (8 K-elements per lane and operand, f32 accumulate)
template <typename T, typename V> __device__ inline f32x16 mfma_32x32x16(V a, V b, f32x16 c) {
  static_assert(sizeof(V) == 16, "8 x 16-bit operand");
  if constexpr (sizeof(T) == 2 && !__is_same(T, f16_t))
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <typename T, typename V> __device__ inline f32x4 mfma_16x16x32(V a, V b, f32x4 c) {
  static_assert(sizeof(V) == 16, "8 x 16-bit operand");
  if constexpr (sizeof(T) == 2 && !__is_same(T, f16_t))
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// 16-byte vector load/store of VE elements into/out of float registers
template <typename T> __device__ inline void ldvec(const T* p, float* out);
template <> __device__ inline void ldvec<float>(const float* p, float* out) {
  f32x4 v = *reinterpret_cast<const f32x4*>(p);
  out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
}
template <> __device__ inline void ldvec<bf16_t>(const bf16_t* p, float* out) {
  u32x4 v = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    out[2 * i] = __builtin_bit_cast(float, v[i] << 16);
    out[2 * i + 1] = __builtin_bit_cast(float, v[i] & 0xffff0000u);
  }
}
template <> __device__ inline void ldvec<f16_t>(const f16_t* p, float* out) {
  const f16x8 v = *reinterpret_cast<const f16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = (float)v[i];
}
template <typename T> __device__ inline void stvec(T* p, const float* in);
template <> __device__ inline void stvec<float>(float* p, const float* in) {
  f32x4 v = {in[0], in[1], in[2], in[3]};
  *reinterpret_cast<f32x4*>(p) = v;
}
// two floats -> one dword of bf16 (v_cvt_pk_bf16_f32: one instruction per PAIR, round-to-nearest-even, NaN stays NaN)
__device__ inline uint32_t pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_{lo, hi}, bf16x2_));
}
template <> __device__ inline void stvec<bf16_t>(bf16_t* p, const float* in) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack_bf16x2(in[2 * i], in[2 * i + 1]);
  *reinterpret_cast<u32x4*>(p) = v;
}

template <> __device__ inline void stvec<f16_t>(f16_t* p, const float* in) {
  f16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (f16_t)in[i];
  *reinterpret_cast<f16x8*>(p) = v;
}

// streaming forms (the BatchNorm passes): non-temporal loads / stores of the same 16 bytes
template <typename T> __device__ inline void ldvec_nt(const T* p, float* out) {
  if constexpr (sizeof(T) == 4) {
    // (as a u32x4 load whose lanes are bit-cast one by one, hipcc of ROCm 7.2 split this into four dword loads from the SAME address)
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
  } else {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    if constexpr (__is_same(T, f16_t)) {
      const f16x8 h = __builtin_bit_cast(f16x8, v);
#pragma unroll
      for (int i = 0; i < 8; ++i) out[i] = (float)h[i];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        out[2 * i] = __builtin_bit_cast(float, v[i] << 16);
        out[2 * i + 1] = __builtin_bit_cast(float, v[i] & 0xffff0000u);
      }
    }
  }
}
template <typename T> __device__ inline void stvec_nt(T* p, const float* in) {
  if constexpr (sizeof(T) == 4) {
    const f32x4 v = {in[0], in[1], in[2], in[3]};
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
  } else {
    u32x4 v;
    if constexpr (__is_same(T, f16_t)) {
      f16x8 h;
#pragma unroll
      for (int i = 0; i < 8; ++i) h[i] = (f16_t)in[i];
      v = __builtin_bit_cast(u32x4, h);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = pack_bf16x2(in[2 * i], in[2 * i + 1]);
    }
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
  }
}

// ---- activations ---------------------------------------------------------------------------------------------
__host__ __device__ inline float dy_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__host__ __device__ inline float dy_act(int act, float u) {
  if (act == DY_ACT_SILU) return u * dy_sigmoid(u);
  if (act == DY_ACT_LEAKY) return u > 0.f ? u : 0.1f * u;
  return u;
}
// d act(u) / du
__host__ __device__ inline float dy_dact(int act, float u) {
  if (act == DY_ACT_SILU) {
    float s = dy_sigmoid(u);
    return s * (1.0f + u * (1.0f - s));
  }
  if (act == DY_ACT_LEAKY) return u > 0.f ? 1.0f : 0.1f;
  return 1.0f;
}

// ---- output pixel -> (image, row, column) for the rows of one tile ---------------------------------------------------------------
// The integer division pair is done once, on the wave-uniform first pixel of the tile; a lane's row offset r (< 65,536) is folded in
// with float reciprocals: floor((x + 0.5) * (1 / d)) is exact for x < 2^16 (the true value is at least 0.5 / d away from an
// integer, the float error is below (x + 0.5) / d * 2^-22).  Four 32-bit divisions per lane cost more than the rest of a conv tile's
// address set-up together.
struct DyTileWalk {
  int img0, oh0, ow0, Hd, Wd;
  float inv_h, inv_w;
  __device__ inline DyTileWalk(long m0, int Hd_, int Wd_) : Hd(Hd_), Wd(Wd_) {
    const unsigned HWd = (unsigned)(Hd_ * Wd_);
    const unsigned mm = (unsigned)m0;                    // m0 < 2^31 (launchers)
    const unsigned img = mm / HWd;
    const unsigned rem = mm - img * HWd;
    const unsigned oh = rem / (unsigned)Wd_;
    img0 = (int)img;
    oh0 = (int)oh;
    ow0 = (int)(rem - oh * (unsigned)Wd_);
    inv_h = 1.0f / (float)Hd_;
    inv_w = 1.0f / (float)Wd_;
  }
  __device__ inline void at(int r, int& img, int& oh, int& ow) const {
    const int x = ow0 + r;
    const int q = (int)(((float)x + 0.5f) * inv_w);
    ow = x - q * Wd;
    const int y = oh0 + q;
    const int q2 = (int)(((float)y + 0.5f) * inv_h);
    oh = y - q2 * Hd;
    img = img0 + q2;
  }
};

// ---- wave / block reductions (wave = 64 lanes) ------------------------------------------------------------------
// sum over each aligned group of 16 lanes (a DPP row), result in every lane of the group: four v_add_f32 with a DPP operand
// (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror) instead of four ds_bpermute round trips through LDS
__device__ inline float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
  return v;
}
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sum over a block of up to 1024 threads; result valid in every thread. `sm` must hold >= 17 floats.
__device__ inline float block_sum(float v, float* sm) {
  v = wave_sum(v);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += sm[i];
  return r;
}

__device__ inline void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }
__device__ inline void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }
