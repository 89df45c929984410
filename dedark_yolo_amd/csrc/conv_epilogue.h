// Shared epilogue of the 16-bit (bf16 / f16) conv kernels (conv.hip, conv_v2.hip, conv_v3.hip): f32 accumulators -> affine + activation ->
// bf16 tile in HBM with 16-byte stores, plus the per-channel (sum, sum of squares) of the RAW accumulators for BatchNorm.
//
// The 32x32 MFMA leaves lane (col = lane&31, half hh = lane>>5) with rows (r&3) + 8*(r>>2) + 4*hh of ONE column: four
// consecutive registers are four consecutive ROWS.  The first version wrote a row-major [BM][BN] image with 64 two-byte
// ds_write per lane (15.8 k cycles of a 108 k-cycle block, measured with s_memtime).  Here the image is stored TRANSPOSED,
// imageT[col][row], so the four rows of a register quad are one 8-byte ds_write_b64 (16 per lane), and the store phase reads
// it back with ds_read_b64_tr_b16: a 16-lane group fetches 4 channels x 16 pixels and every lane receives the 4 channels of
// ITS pixel; two reads = the 16-byte (8-channel) vector of one pixel.  A wave covers 16 pixels x 32 channels per store
// instruction (64 contiguous bytes per pixel).
#pragma once
#include "dy_common.h"

namespace dy_epi {

template <int BM>
constexpr int pitch() { return BM * 2 + 8; }      // bytes per imageT row (BM rows x 2 B + 8 B pad)

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

template <int BM, int BN>
constexpr int image_bytes() { return BN * pitch<BM>(); }

template <typename T16 = bf16_t>
__device__ inline uint32_t pack2(float a, float b) {
  if constexpr (__is_same(T16, bf16_t)) return pack_bf16x2(a, b);           // one v_cvt_pk_bf16_f32 for the pair
  else return (uint32_t)cvt16<T16>(a) | ((uint32_t)cvt16<T16>(b) << 16);
}

// Store phase: the transposed image [BN cols][BM rows] (pitch<BM>() bytes per column) -> HBM.  Unit = 16 pixels x 32 channels
// per wave instruction; NW waves share the units.  The caller has synchronised the block after writing the image.
template <int BM, int BN, int NW, typename T16, typename OffFn>
__device__ inline void store_image(const char* smem, int lane, int wave, long m0, int n0, long M, int Cd, int accumulate, T16* dst,
                                   OffFn off) {
  constexpr int PT = pitch<BM>();
  constexpr int UP = BM / 16, UC = BN / 32;
  const int g = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  // a wave walks the channel tiles of ITS pixel tiles: consecutive store instructions of one wave complete the 128-byte lines
  // of the same 16 pixels (adjacent 64-byte pieces), instead of eight waves each writing one piece of every line
  static_assert(UP % NW == 0, "pixel tiles per wave");
#pragma unroll 2
  for (int u = 0; u < (UP / NW) * UC; ++u) {
    const int pt = wave + NW * (u / UC), ct = u % UC;
    const int cb = ct * 32 + 8 * g;                // first channel of this lane's 8-channel vector
    const char* base = smem + (cb + tq) * PT + (pt * 16 + 4 * tp) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(base + 4 * PT));
    const long m = m0 + pt * 16 + li;
    const int n = n0 + cb;
    if (m < M && n < Cd) {
      u32x4 v;
      v[0] = (uint32_t)(uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
      v[1] = (uint32_t)(uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
      v[2] = (uint32_t)(uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
      v[3] = (uint32_t)(uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
      T16* o = dst + off(m) + n;
      if (n + 8 <= Cd) {
        if (accumulate) {
          float x[8], y[8];
          ldvec<T16>(o, x);
          ldvec<T16>(reinterpret_cast<const T16*>(&v), y);
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] += y[e];
          stvec<T16>(o, x);
        } else {
          *reinterpret_cast<u32x4*>(o) = v;
        }
      } else {                                     // ragged channel tail (never happens for padded views)
        const T16* e = reinterpret_cast<const T16*>(&v);
        for (int q = 0; q < 8 && n + q < Cd; ++q) {
          if (accumulate) DT<T16>::st(o + q, DT<T16>::ld(o + q) + DT<T16>::ld(e + q));
          else o[q] = e[q];
        }
      }
    }
  }
}

// ---- Row-major variant for kernels whose accumulators hold CHANNELS in the registers and PIXELS on the lanes (the conv_v4
// kernels compute the transposed product: MFMA rows = output channels).  A lane's 4 consecutive channels of one pixel are one
// ds_write_b64 into image[pixel][channel] (pitch BN*2 + 8 bytes: 16 pixels x 8 bytes hit 32 different banks); the store phase
// reads 16 bytes per lane (two ds_read_b64, conflict-free) so that each 16-lane quarter of a store instruction writes 256
// CONTIGUOUS bytes of one pixel (the [col][row] image above gives 16 pixels x 64 bytes per instruction: a quarter-wave then touches
// 16 different lines and the store phase ran at 1-2 TB/s chip-wide).
template <int BN>
constexpr int row_pitch() { return BN * 2 + 8; }

template <int BM, int BN>
constexpr int row_image_bytes() { return BM * row_pitch<BN>(); }

// `add` (optional): a second [pixel][channel] view (pixel stride add_ld) whose values are added on the way out -- dst = [dst +]
// tile + add (the shortcut gradient of a Bottleneck joins the data gradient of its first conv here instead of in a copy pass).
// RG (rows per 64-row group, default 64 = every row): a tile whose row groups are only partly populated (conv_v4's shorter tail tiles) --
// image row px holds output pixel m0 + (px >> 6) * RG + (px & 63) when (px & 63) < RG and nothing otherwise.
template <int BM, int BN, int NW, typename T16, typename OffFn, int RG = 64>
__device__ inline void store_rows(const char* smem, int lane, int wave, long m0, int n0, long M, int Cd, int accumulate, T16* dst, OffFn off,
                                  const T16* __restrict__ add = nullptr, long add_ld = 0) {
  constexpr int PT = row_pitch<BN>();
  constexpr int LPP = BN >= 128 ? 16 : BN / 8;      // lanes (16-byte chunks) per pixel and instruction: 16 (256 B) or 8 (BN = 64: 128 B)
  constexpr int PPI = 64 / LPP;                     // pixels per wave instruction
  constexpr int ITERS = BM / PPI / NW;
  constexpr int HALVES = BN >= 128 ? BN / 128 : 1;
  const int pl = lane / LPP, chunk = lane % LPP;
  const bool nt = accumulate & 2;                   // bit 1 (set by the launcher): plain stores of an output beyond 128 MB go non-temporal
  accumulate &= 1;
#pragma unroll 2
  for (int u = 0; u < ITERS; ++u) {
    const int px = PPI * (wave + NW * u) + pl;
    if (RG < 64 && (px & 63) >= RG) continue;
    const long m = RG < 64 ? m0 + (px >> 6) * RG + (px & 63) : m0 + px;
#pragma unroll
    for (int h = 0; h < HALVES; ++h) {
      const int c = 128 * h + 8 * chunk;
      const uint2 lo = *reinterpret_cast<const uint2*>(smem + px * PT + c * 2);
      const uint2 hi = *reinterpret_cast<const uint2*>(smem + px * PT + c * 2 + 8);
      const int n = n0 + c;
      if (m < M && n < Cd) {
        u32x4 v = {lo.x, lo.y, hi.x, hi.y};
        T16* o = dst + off(m) + n;
        if (n + 8 <= Cd) {
          if (accumulate || add) {
            float x[8], y[8];
            ldvec<T16>(reinterpret_cast<const T16*>(&v), y);
            if (accumulate) {
              ldvec<T16>(o, x);
#pragma unroll
              for (int e = 0; e < 8; ++e) y[e] += x[e];
            }
            if (add) {
              ldvec<T16>(add + m * add_ld + n, x);
#pragma unroll
              for (int e = 0; e < 8; ++e) y[e] += x[e];
            }
            if (nt) {
              u32x4 w;
              stvec<T16>(reinterpret_cast<T16*>(&w), y);
              asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(o), "v"(w) : "memory");
            } else {
              stvec<T16>(o, y);
            }
          } else if (nt) {
            // (as inline asm: with the builtin the two branches differ only in their !nontemporal metadata and SimplifyCFG sinks them into
            //  ONE plain store)
            asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(o), "v"(v) : "memory");
          } else {
            *reinterpret_cast<u32x4*>(o) = v;
          }
        } else {
          const T16* e = reinterpret_cast<const T16*>(&v);
          for (int q = 0; q < 8 && n + q < Cd; ++q) {
            float t = DT<T16>::ld(e + q);
            if (accumulate) t += DT<T16>::ld(o + q);
            if (add) t += DT<T16>::ld(add + m * add_ld + n + q);
            DT<T16>::st(o + q, t);
          }
        }
      }
    }
  }
}

// Block tile BM x BN on WM x WN waves (wave tile 32*TM x 32*TN).  `off(m)` = element offset of output pixel m.
template <int BM, int BN, int WM, int WN, int TM, int TN, typename T16, typename OffFn>
__device__ inline void store_tile(char* smem, f32x16 (&acc)[TM][TN], int wm, int wn, int lane, int wave, long m0, int n0, long M, int Cd,
                                  const float* scale, const float* shift, int act, int accumulate, T16* dst, OffFn off,
                                  float (&csum)[TN], float (&csq)[TN]) {
  constexpr int PT = pitch<BM>();
  static_assert(BM / WM == 32 * TM && BN / WN == 32 * TN, "wave tiling");
  const int cl = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int j = 0; j < TN; ++j) { csum[j] = 0.f; csq[j] = 0.f; }
  __builtin_amdgcn_s_barrier();                   // every wave is done reading the ring
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = wn * (BN / WN) + j * 32 + cl;
    const int n = n0 + col;
    const bool nok = n < Cd;
    const float sc = (nok && scale) ? scale[n] : 1.f;
    const float sf = (nok && shift) ? shift[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row0 = wm * (BM / WM) + i * 32 + 4 * hh;
      const long mrem = M - (m0 + row0);           // rows with (r&3)+8*(r>>2) < mrem are real pixels
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = acc[i][j][4 * rq + e];
          if (nok && 8 * rq + e < mrem) {
            csum[j] += a;
            csq[j] += a * a;
          }
          float u = a * sc + sf;
          if (act == DY_ACT_SILU) u = u * dy_sigmoid(u);
          else if (act == DY_ACT_LEAKY) u = u > 0.f ? u : 0.1f * u;
          v[e] = u;
        }
        uint2 w = {pack2<T16>(v[0], v[1]), pack2<T16>(v[2], v[3])};
        *reinterpret_cast<uint2*>(smem + col * PT + (row0 + 8 * rq) * 2) = w;
      }
    }
  }
  __syncthreads();
  store_image<BM, BN, WM * WN>(smem, lane, wave, m0, n0, M, Cd, accumulate, dst, off);
}

}  // namespace dy_epi
