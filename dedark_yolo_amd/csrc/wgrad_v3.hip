// Band weight gradient for the 64- and 128-channel 3x3 / stride 1 / pad 1 layers (Bottleneck convs of the C2f blocks,
// reference ultralytics/nn/modules/block.py:553-565), bf16.
//
// These layers are bandwidth-bound (64->64 at 160x160, B = 64: 420 MB of operands for 121 GFLOP), yet the tiled kernels re-read
// x once per tap and dz once per k'-tile through L2 (4-5x the algorithmic traffic by the PMC counters) and run at 190-210 TF.
// Here a block walks over consecutive rows of ONE image and keeps, in a zero-padded pixel space,
//     x rows h-1, h, h+1 (+ the row being prefetched)  in a 4-slot ring of LDS row buffers [pixel][64 ci]
//     dz row h (+ the next one)                        in 2 row buffers               [pixel][64 co]
// so that every operand byte is fetched from HBM exactly once and all 9 taps are shifted windows of the same LDS rows:
//     D[(kh,kw,ci)][co] += sum_px  x_pad[h+kh-1][px+kw-1][ci] * dz_pad[h][px][co]
// The zero padding (one pixel left / right, zero rows above / below the image) is produced by the DMA itself (padding lanes
// fetch a zero page), so there are no border masks anywhere.  Operands are read with ds_read_b64_tr_b16 (pixel-major LDS rows
// -> 8 consecutive pixels of one channel per lane).  6 waves: wave w owns k' tiles 3w..3w+2 (of 18 = 9 taps x 2 ci tiles) x both
// co tiles = 6 accumulator tiles; 10 transposing reads per 6 MFMA, issued for the NEXT 16-pixel slice between the MFMAs of the current one
// (two fragment sets); the row buffers are bank-swizzled (swizzle_key).  One barrier per image row.  With 128 input channels there are
// 36 k' tiles on 12 waves; every block covers 64 output channels (blockIdx.y selects the 64-wide slice of dz).
// Partial sums go to `scratch` as [co slice][block][9*CI][64] f32 and a second kernel adds them in a fixed order.
#include <stdlib.h>
#include "dy_common.h"
#include "../../include/dedark_yolo.h"

// diagnostics only (tools/gpu/wg3_ablate.sh builds variant libraries with -DWG3_ABLATE=n; results are wrong, only the time matters):
// 1 = DMA + one barrier per row only, 2 = DMA + MFMAs on register constants (no fragment reads), 3 = the bare MFMA loop (no DMA either)
#ifndef WG3_ABLATE
#define WG3_ABLATE 0
#endif

namespace wg3 {

constexpr int CO = 64;                    // output channels per block
constexpr int ZPB = CO * 2;               // bytes per dz pixel in LDS

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

__device__ __attribute__((aligned(16))) unsigned char g_zero_page[16];

struct P {
  const char* x;
  long x_ld;
  const char* dz;
  long dz_ld;
  int N, H, W;
  int PW;          // padded pixels per row processed by the MFMAs (multiple of 16, >= W + 2)
  int XW;          // pixel slots per x row buffer (PW + 8): slot j holds padded column j - 1, i.e. image column j - 2
  int rb, nseg;    // rows per block, row segments per image
  int SW, nstrip;  // column strips (images too wide for the LDS row buffers): strip width, strips per row segment; one strip = whole rows
  int co_stride;   // unused padding guard (kept zero)
  float* part;     // [co slices][blocks][9*CI][64]
};

// Bank swizzle of the row buffers.  A 32-lane half of ds_read_b64_tr_b16 reads 4 consecutive pixels x 32 channels (64 bytes per pixel)
// and the LDS bank row is 256 bytes: with plain [pixel][channel] rows the four pixels sit a multiple of 256 bytes apart (128 channels:
// 4-way conflict) or two of them do (64 channels: 2-way) -- SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.69 on 128->128 at 80x80,
// exactly 6 x reads at 4x + 4 dz reads at 2x per slice, and the LDS array, not the MFMA pipe, set the pace.  So the 64-byte quarter q of
// slot pixel P is stored at quarter q ^ key(P), with key chosen so that 4 consecutive pixels land in 4 different quarters of the bank
// row: 128 channels (one pixel per bank row) key = P & 3; 64 channels (two pixels per bank row) key = (P >> 1) & 1.  The DMA is
// lane-linear in LDS, so the permutation is applied to the SOURCE chunk a lane fetches.
template <int C>
__device__ inline int swizzle_key(int P) {
  return C == 128 ? (P & 3) : ((P >> 1) & 1);
}

// one image row (or one column strip of it) -> one LDS row buffer; `shift` = slot index of image column col0 (2 for x, 1 for dz); rows
// outside the image and image columns outside [lo, hi) come from the zero page.  x rows are loaded with one halo column on each side of
// the strip (real neighbours, zeros only at the image border); dz rows with exactly the strip's columns, so that strips partition the sum.
// C = channels per LDS pixel (64 or 128), NW = waves of the block, c0 = first channel fetched
template <int C, int NW>
__device__ inline void load_row(const char* src, long ld, int c0, int n, int h, int H, int W, char* buf, int slots, int shift, int wave,
                                int lane, const char* zero, int col0 = 0, int lo = 0, int hi = 1 << 30) {
  if (hi > W) hi = W;
  constexpr int PPI = 512 / C;                          // pixels per wave instruction (1 KiB)
  constexpr int CPP = C / 8;                            // 16-byte chunks per pixel
  const int ninstr = slots / PPI;
  const bool rowok = h >= 0 && h < H;
  const int pl = lane / CPP;
  const int chunk = (lane % CPP) ^ swizzle_key<C>(pl) * 4;         // PPI is a multiple of 4: the key of slot PPI*i + pl is that of pl
  // wave-uniform row base (scalar registers) + one 32-bit offset per lane: `wave` is uniform (readfirstlane), so the instruction loop, the
  // LDS destination (M0) and the 64-bit row arithmetic stay on the scalar unit
  const int ldb = (int)ld * 2;
  const char* row = src + ((((long)n * H + h) * W + col0) * ld + c0) * 2;
  const int voff = (pl - shift) * ldb + chunk * 16;
  const int wl = col0 + pl - shift;
  for (int i = wave; i < ninstr; i += NW) {
    const int w = wl + PPI * i;
    const char* g = (rowok && w >= lo && w < hi) ? row + (long)(PPI * i) * ldb + voff : zero;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(buf + i * 1024), 16, 0, 0);
  }
}

template <int CI, typename T = bf16_t>
__global__ __launch_bounds__(CI * 6) void wgrad_kernel(P p) {
  constexpr int PXB = CI * 2;                           // bytes per x pixel in LDS
  constexpr int CT = CI / 32;                           // ci tiles per tap
  constexpr int KT = 9 * CT;                            // k' tiles of 32
  constexpr int NW = KT / 3;                            // waves (3 k' tiles x 2 co tiles each)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int XB = p.XW * PXB, ZB = p.PW * ZPB;           // bytes per x / dz row buffer
  const int co0 = blockIdx.y * CO;
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_ptr_t)smem;
  char* xring = smem;                                   // 4 x rows
  char* zring = smem + 4 * XB;                          // 2 dz rows
  const char* zero = reinterpret_cast<const char*>(g_zero_page);
  const int bseg = blockIdx.x / p.nstrip, strip = blockIdx.x - bseg * p.nstrip;
  const int n = bseg / p.nseg, seg = bseg - n * p.nseg;
  const int h0 = seg * p.rb, h1 = min(h0 + p.rb, p.H);
  const int col0 = strip * p.SW, xlo = max(col0 - 1, 0), xhi = col0 + p.SW + 1, zhi = col0 + p.SW;

  f32x16 acc[3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // per-lane part of the transposed-read addresses: lane 4q+p of a 16-lane group addresses pixel q of a 4-pixel block, channels
  // 4p..4p+3 of the group's 16-channel block; lanes 0-31 / 32-63 take pixels 0-7 / 8-15 of the 16-pixel slice
  const int g4 = lane >> 4, hh = g4 >> 1, blk = g4 & 1, tq = (lane & 15) >> 2, tp = lane & 3;
  const int lane_x = (8 * hh + tq) * PXB + blk * 32 + tp * 8;
  const int lane_z = (8 * hh + tq) * ZPB + blk * 32 + tp * 8;
  int a_kh[3], a_off[3];                                // tile -> ring row selector and byte offset (tap column + ci tile)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int kt = 3 * wave + i, tap = kt / CT, ct = kt - tap * CT;
    a_kh[i] = tap / 3;
    const int kw = tap - 3 * a_kh[i];
    a_off[i] = kw * PXB + (ct ^ swizzle_key<CI>(kw + tq)) * 64 + lane_x;        // x slot index = px + kw (px = 16 s + 8 hh + tq, + 4)
  }
  const int kz = swizzle_key<CO>(tq);
  const int b_off[2] = {kz * 64 + lane_z, (1 ^ kz) * 64 + lane_z};

  // slot of image row r in the x ring: (r + 1) & 3
  load_row<CI, NW>(p.x, p.x_ld, 0, n, h0 - 1, p.H, p.W, xring + ((h0 + 0) & 3) * XB, p.XW, 2, wave, lane, zero, col0, xlo, xhi);
  load_row<CI, NW>(p.x, p.x_ld, 0, n, h0, p.H, p.W, xring + ((h0 + 1) & 3) * XB, p.XW, 2, wave, lane, zero, col0, xlo, xhi);
  load_row<CI, NW>(p.x, p.x_ld, 0, n, h0 + 1, p.H, p.W, xring + ((h0 + 2) & 3) * XB, p.XW, 2, wave, lane, zero, col0, xlo, xhi);
  load_row<CO, NW>(p.dz, p.dz_ld, co0, n, h0, p.H, p.W, zring + (h0 & 1) * ZB, p.PW, 1, wave, lane, zero, col0, col0, zhi);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const int nslices = p.PW >> 4;
  for (int h = h0; h < h1; ++h) {
    if (h + 1 < h1 && WG3_ABLATE != 3) {                 // prefetch what the next row needs
      load_row<CI, NW>(p.x, p.x_ld, 0, n, h + 2, p.H, p.W, xring + ((h + 3) & 3) * XB, p.XW, 2, wave, lane, zero, col0, xlo, xhi);
      load_row<CO, NW>(p.dz, p.dz_ld, co0, n, h + 1, p.H, p.W, zring + ((h + 1) & 1) * ZB, p.PW, 1, wave, lane, zero, col0, col0, zhi);
    }
    // Transposing reads as inline asm: hipcc (ROCm 7.2) orders the ds_read_tr16 builtin behind every pending LDS DMA with an
    // s_waitcnt vmcnt(0) -- here in front of the first read of every slice, i.e. the prefetch of the next row had to land completely
    // before any MFMA of the current row (an exposed HBM round trip per image row: ~0.5 us of MFMA work per row waited 1-2 us).  The
    // counts are ours: lgkmcnt(0) in front of the MFMAs of a slice, vmcnt(0) + barrier at the end of the row.
    unsigned xa[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) xa[i] = lds0 + (unsigned)(((h + a_kh[i]) & 3) * XB + a_off[i]);       // x row h + kh - 1
    const unsigned zb = lds0 + (unsigned)(4 * XB + (h & 1) * ZB);
    // Two fragment sets; the 10 reads of slice s + 1 go into the gaps between the first MFMAs of slice s (4 + 4 + 2 behind MFMA 0, 1, 2:
    // the last read has three MFMAs to land in), not as a burst in front of them -- the bare MFMA loop of this kernel takes 95 us on
    // 128->128 at 80x80 (B = 64), the reads as a burst added 30 us on top of it (tools/gpu/wg3_ablate.sh).
    uint2 alo[2][3], ahi[2][3], blo[2][2], bhi[2][2];
    auto read_a = [&](int set, int s, int i) {
      if (WG3_ABLATE >= 1) {
        alo[set][i] = ahi[set][i] = uint2{0x3f803f80u + (unsigned)s, 0x3f803f80u + (unsigned)lane};
        return;
      }
      const unsigned so = s * 16 * PXB;
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(alo[set][i]) : "v"(xa[i] + so) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ahi[set][i]) : "v"(xa[i] + so), "n"(4 * PXB) : "memory");
    };
    auto read_b = [&](int set, int s, int j) {
      if (WG3_ABLATE >= 1) {
        blo[set][j] = bhi[set][j] = uint2{0x3f803f80u, 0x3f803f80u + (unsigned)s};
        return;
      }
      const unsigned sz = s * 16 * ZPB;
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(blo[set][j]) : "v"(zb + b_off[j] + sz) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(bhi[set][j]) : "v"(zb + b_off[j] + sz), "n"(4 * ZPB) : "memory");
    };
    auto mfma = [&](int set, int i, int j) {
      if (WG3_ABLATE == 1) return;
      const u32x4 af = u32x4{alo[set][i].x, alo[set][i].y, ahi[set][i].x, ahi[set][i].y};
      const u32x4 bf = u32x4{blo[set][j].x, blo[set][j].y, bhi[set][j].x, bhi[set][j].y};
      acc[i][j] = mfma_32x32x16<T>(af, bf, acc[i][j]);
    };
    // one slice: the MFMAs of fragment set `cur`, slice sn read into the other set on the way (behind the last slice of the row the
    // last slice once more, unused: one loop body for every slice)
    auto slice = [&](int cur, int sn) {
      if (WG3_ABLATE == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      mfma(cur, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      read_a(cur ^ 1, sn, 0);
      read_a(cur ^ 1, sn, 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma(cur, 0, 1);
      __builtin_amdgcn_sched_barrier(0);
      read_b(cur ^ 1, sn, 0);
      read_b(cur ^ 1, sn, 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma(cur, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
      read_a(cur ^ 1, sn, 2);
      __builtin_amdgcn_sched_barrier(0);
      mfma(cur, 1, 1);
      mfma(cur, 2, 0);
      mfma(cur, 2, 1);
      __builtin_amdgcn_sched_barrier(0);
    };
    read_a(0, 0, 0); read_a(0, 0, 1); read_a(0, 0, 2); read_b(0, 0, 0); read_b(0, 0, 1);
    const int last = nslices - 1;
    for (int s = 0; s < nslices; s += 2) {
      slice(0, min(s + 1, last));
      if (s + 1 < nslices) slice(1, min(s + 2, last));
    }
    if (WG3_ABLATE == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the unused reads behind the last slice
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // D layout of the 32x32 MFMA: col (co) = lane&31, row (k') = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* out = p.part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (KT * 32 * CO);
  const int cl = lane & 31, h5 = lane >> 5;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row0 = (3 * wave + i) * 32 + 4 * h5, col = j * 32 + cl;
#pragma unroll
      for (int r = 0; r < 16; ++r) out[(row0 + (r & 3) + 8 * (r >> 2)) * CO + col] = acc[i][j][r];
    }
}

// g[co][ci][kh][kw] = sum over blocks of part[co / 64][block][(kh*3+kw)*CI + ci][co % 64]; threads run along co (contiguous reads)
__global__ __launch_bounds__(256) void reduce_kernel(const float* __restrict__ part, int nblk, int CI, int Cout, int Cin,
                                                     float* __restrict__ g) {
  // block = (32 consecutive co, one k); eight split-lanes walk the slabs, added in a fixed order
  __shared__ float red[8][33];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int co = blockIdx.x * 32 + cl;
  const int k = blockIdx.y;
  const int tap = k / CI, ci = k - tap * CI;
  if (ci >= Cin) return;                                     // block-uniform
  const long slab = (long)9 * CI * CO;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (co < Cout) {
    const float* src = part + (long)(co / CO) * nblk * slab + (long)k * CO + (co % CO);
    int s = sl;
    for (; s + 24 < nblk; s += 32) {
      const float v0 = src[(long)s * slab], v1 = src[(long)(s + 8) * slab], v2 = src[(long)(s + 16) * slab], v3 = src[(long)(s + 24) * slab];
      a0 += v0; a1 += v1; a2 += v2; a3 += v3;
    }
    for (; s < nblk; s += 8) a0 += src[(long)s * slab];
  }
  red[sl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sl == 0 && co < Cout) {
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) t += red[u][cl];
    g[((long)co * Cin + ci) * 9 + tap] = t;
  }
}

}  // namespace wg3

// column strips per row so that 4 x rows + 2 dz rows of one strip (+ halo) fit the 160 KiB of LDS; 0 = does not fit in <= 4 strips.
// (1280x1280 inputs: 128 channels at W = 160 and 64 channels at W = 320 need two strips; everything at 640x640 takes whole rows)
static int wg3_strips(int Wi, int Cin_pad) {
  for (int ns = 1; ns <= 4; ++ns) {
    const int SW = ((Wi + ns - 1) / ns + 15) / 16 * 16;
    const int PW = (SW + 2 + 15) / 16 * 16;
    if (4 * (PW + 8) * Cin_pad * 2 + 2 * PW * wg3::ZPB <= 160 * 1024) return ns;
  }
  return 0;
}

bool dy_wgrad_v3_eligible(int dtype, int Cin_pad, int Cout_pad, int KH, int KW, int stride, int pad, int dil, int N, int Hi, int Wi,
                          long x_ld, long dz_ld, long scratch_elems) {
  static const bool off = dy_env("DY_NO_WGRAD_V3") != nullptr;
  if (off) return false;
  if (!((dtype == DY_BF16 || dtype == DY_F16) && (Cin_pad == 64 || Cin_pad == 128) && Cout_pad % 64 == 0 && Cout_pad <= 128 && KH == 3 && KW == 3 && stride == 1 &&
        pad == 1 && dil == 1))
    return false;
  if ((x_ld * 2) % 16 != 0 || (dz_ld * 2) % 16 != 0) return false;
  if (wg3_strips(Wi, Cin_pad) == 0) return false;       // the six row buffers must fit LDS, in up to four column strips
  // worth it when the pixel loop is long enough to amortise the slab per block (64->64 at 40x40, B = 32 is not)
  static const long min_m = dy_env("DY_WG3_MINM") ? atol(dy_env("DY_WG3_MINM")) : 131072;
  return (long)N * Hi * Wi >= min_m &&
         scratch_elems >= (long)N * wg3_strips(Wi, Cin_pad) * (Cout_pad / wg3::CO) * 9 * Cin_pad * wg3::CO;     // >= 1 block per image and strip
}

int dy_wgrad_v3_launch(const void* x, long x_ld, int N, int Hi, int Wi, int Cin_pad, const void* dz, long dz_ld, int Cout_pad, int Cout,
                       int Cin, float* scratch, long scratch_elems, float* g_oihw, int dtype, void* stream) {
  using namespace wg3;
  P p;
  p.x = (const char*)x; p.x_ld = x_ld; p.dz = (const char*)dz; p.dz_ld = dz_ld;
  p.N = N; p.H = Hi; p.W = Wi;
  p.nstrip = wg3_strips(Wi, Cin_pad);
  DY_CHECK(p.nstrip >= 1, "wgrad_v3: image rows too wide for the LDS row buffers");
  p.SW = p.nstrip == 1 ? Wi : ((Wi + p.nstrip - 1) / p.nstrip + 15) / 16 * 16;
  p.nstrip = (Wi + p.SW - 1) / p.SW;
  p.PW = (p.SW + 2 + 15) / 16 * 16;
  p.XW = p.PW + 8;
  p.co_stride = 0;
  p.part = scratch;
  const int ny = Cout_pad / CO;
  // ONE round of blocks over the 256 CUs (1 block per CU: the row buffers take 131 KB), whole rows of one image each: every block
  // parks a 295 / 147 KB slab, so a second round costs more in slab traffic than its finer tail saves (B = 64: 128->128 at 80x80
  // 203 -> 159 us, 64->64 at 160x160 183 -> 170 us with 256 instead of 512 blocks); and the slabs must fit the scratch buffer
  const long slab = (long)9 * Cin_pad * CO;
  const long maxblk = scratch_elems / (slab * ny);
  static const int target_blocks = dy_env("DY_WG3_BLOCKS") ? atoi(dy_env("DY_WG3_BLOCKS")) : 256;
  int nseg = target_blocks / (N * ny * p.nstrip);
  if ((long)N * nseg * p.nstrip > maxblk) nseg = (int)(maxblk / ((long)N * p.nstrip));
  if (nseg < 1) nseg = 1;
  if (nseg > Hi) nseg = Hi;
  p.rb = (Hi + nseg - 1) / nseg;
  p.nseg = (Hi + p.rb - 1) / p.rb;
  const int nblk = N * p.nseg * p.nstrip;
  DY_CHECK((long)nblk * ny * slab <= scratch_elems, "dy_conv2d_wgrad: scratch too small (%ld floats, need %ld)", scratch_elems,
           (long)nblk * ny * slab);
  const int shmem = 4 * p.XW * Cin_pad * 2 + 2 * p.PW * ZPB;
  static int configured = 0;
  const bool f16 = dtype == DY_F16;
  const int bit = (Cin_pad == 64 ? 1 : 2) << (f16 ? 2 : 0);
  if (!(configured & bit)) {
    const void* fn = Cin_pad == 64 ? (f16 ? reinterpret_cast<const void*>(&wgrad_kernel<64, f16_t>) : reinterpret_cast<const void*>(&wgrad_kernel<64, bf16_t>))
                                   : (f16 ? reinterpret_cast<const void*>(&wgrad_kernel<128, f16_t>) : reinterpret_cast<const void*>(&wgrad_kernel<128, bf16_t>));
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      dy_set_error("wgrad_v3: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return 3;
    }
    configured |= bit;
  }
  hipStream_t st = (hipStream_t)stream;
  dy_note_kernel(Cin_pad == 64 ? "wg3::wgrad_kernel<64>+reduce_kernel" : "wg3::wgrad_kernel<128>+reduce_kernel");
  if (Cin_pad == 64) {
    if (f16) wgrad_kernel<64, f16_t><<<dim3(nblk, ny), 384, shmem, st>>>(p);
    else wgrad_kernel<64, bf16_t><<<dim3(nblk, ny), 384, shmem, st>>>(p);
  } else {
    if (f16) wgrad_kernel<128, f16_t><<<dim3(nblk, ny), 768, shmem, st>>>(p);
    else wgrad_kernel<128, bf16_t><<<dim3(nblk, ny), 768, shmem, st>>>(p);
  }
  DY_LAUNCH_CHECK();
  reduce_kernel<<<dim3(dy_cdiv(Cout, 32), 9 * Cin_pad), 256, 0, st>>>(scratch, nblk, Cin_pad, Cout, Cin, g_oihw);
  DY_LAUNCH_CHECK();
  return 0;
}
