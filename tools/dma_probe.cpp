// LDS-DMA delivery probe (gfx950): how many bytes per clock a CU takes in through `buffer_load_dwordx4 ... lds` in the shape of the
// conv kernels' K loops -- no MFMA, no LDS reads, only issue + counted vmcnt + barrier per step -- as a function of
//   rows   : rows per 1 KiB wave-instruction (16 x 64 B as conv_v5, 8 x 128 B as conv_v4, 4 x 256 B as wgrad_v4, 1 x 1024 B)
//   stride : bytes between consecutive rows in memory (a pixel's channel record: 256 B = 128 channels, 512 B = 256 channels, ...)
//   foot   : bytes a block cycles through (its live set: 32 KiB stays in L2 / L1, 4 MiB per block streams from HBM / Infinity Cache)
//   blocks : co-resident blocks per CU (4 waves each), stages in flight per block
// The conv kernels reach ~15 B/clk per CU with real fetches and ~28 B/clk with out-of-range (zero-fill) offsets (DESIGN.md, round-2
// findings); this tool asks which of the above that number depends on.  Build: make -C tools; run on the GPU box: tools/bin/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct Args {
  const char* src;
  unsigned src_bytes;
  unsigned row_stride;     // bytes
  unsigned foot_rows;      // rows a block cycles through (power of two)
  int nsteps;
  int oob;                 // 1: every offset out of range (zero fill, no fetch)
};

// IPS = DMA instructions per wave and step (stage = 4 waves x IPS KiB), INFLIGHT = stages in flight behind the one being "consumed"
template <int ROWS, int IPS, int INFLIGHT>
__global__ __launch_bounds__(256, 2) void probe(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 4 * IPS * 1024, NST = INFLIGHT + 1;
  constexpr int LPR = 64 / ROWS;                  // lanes per row
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, a.src_bytes, 0x00020000);
  const unsigned base_row = blockIdx.x * a.foot_rows;          // every block its own region
  const unsigned lrow = lane / LPR, lcol = (lane % LPR) * 16;
  unsigned row = (wave * IPS) * ROWS + lrow;                    // advances by 4 * IPS * ROWS rows per step
  auto issue = [&](int st) {
#pragma unroll
    for (int j = 0; j < IPS; ++j) {
      const unsigned r = (row + j * ROWS) & (a.foot_rows - 1);
      const unsigned off = a.oob ? 0x80000000u : (base_row + r) * a.row_stride + lcol;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + st * STAGE + (wave * IPS + j) * 1024), 16, (int)off, 0, 0, 0);
    }
    row += 4 * IPS * ROWS;
  };
  int fill = 0;
#pragma unroll
  for (int s = 0; s < INFLIGHT; ++s) issue(fill++);
  for (int k = 0; k < a.nsteps; ++k) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPS * (INFLIGHT - 1)) : "memory");
    __builtin_amdgcn_s_barrier();
    issue(fill);
    fill = fill == NST - 1 ? 0 : fill + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (a.nsteps < 0) *(volatile char*)a.src = smem[tid];         // keep LDS alive
}

template <int ROWS, int IPS, int INFLIGHT>
static void run(const char* label, const char* buf, size_t bytes, unsigned stride, unsigned foot_bytes, int blocks_per_cu, int oob) {
  constexpr int STAGE = 4 * IPS * 1024, NST = INFLIGHT + 1;
  int shmem = STAGE * NST;
  if (blocks_per_cu == 1 && shmem < 81 * 1024) shmem = 81 * 1024;       // force one block per CU
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<ROWS, IPS, INFLIGHT>), hipFuncAttributeMaxDynamicSharedMemorySize, shmem));
  Args a;
  a.src = buf;
  a.row_stride = stride;
  unsigned foot_rows = foot_bytes / (1024 / ROWS);
  unsigned p2 = 1;
  while (p2 * 2 <= foot_rows) p2 *= 2;
  a.foot_rows = p2;
  a.nsteps = 2000;
  a.oob = oob;
  const int nblk = 256 * blocks_per_cu;
  const size_t need = (size_t)nblk * a.foot_rows * stride + 1024;
  if (need > bytes || need > 0x7fffffffUL) { printf("%-44s skipped (needs %zu MB)\n", label, need >> 20); return; }
  a.src_bytes = (unsigned)need;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) probe<ROWS, IPS, INFLIGHT><<<nblk, 256, shmem>>>(a);
  CK(hipEventRecord(e0));
  const int reps = 5;
  for (int w = 0; w < reps; ++w) probe<ROWS, IPS, INFLIGHT><<<nblk, 256, shmem>>>(a);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double bytes_moved = (double)nblk * (a.nsteps + INFLIGHT) * STAGE;
  const double per_cu = bytes_moved / 256 / (ms * 1e-3);
  printf("%-44s rows/instr %2d stride %5u foot %7u KiB  %d blk/CU x %d KiB x %d in flight%s: %7.1f GB/s per CU = %5.1f B/clk @2.4GHz, %6.2f TB/s chip\n", label,
         ROWS, stride, (unsigned)((size_t)a.foot_rows * (1024 / ROWS) >> 10), blocks_per_cu, STAGE >> 10, INFLIGHT, oob ? " OOB" : "", per_cu / 1e9,
         per_cu / 2.4e9, bytes_moved / (ms * 1e-3) / 1e12);
}

int main() {
  const size_t bytes = (size_t)2000 << 20;
  char* buf;
  CK(hipMalloc(&buf, bytes));
  CK(hipMemset(buf, 1, bytes));
  // ---- the conv_v5 shape: 16 rows x 64 B, 24 KiB stages (IPS = 6), 2 in flight, 2 blocks per CU
  run<16, 6, 2>("v5-like, zero fill", buf, bytes, 256, 32 << 10, 2, 1);
  run<16, 6, 2>("v5-like, L2-resident 32 KiB/block", buf, bytes, 256, 32 << 10, 2, 0);
  run<16, 6, 2>("v5-like, 512 KiB/block", buf, bytes, 256, 512 << 10, 2, 0);
  run<16, 6, 2>("16 x 64 B dense, streaming 2 MiB/block", buf, bytes, 64, 2 << 20, 2, 0);
  // ---- row shape at the same bytes
  run<8, 6, 2>("8 x 128 B rows, 32 KiB/block", buf, bytes, 256, 32 << 10, 2, 0);
  run<4, 6, 2>("4 x 256 B rows, 32 KiB/block", buf, bytes, 256, 32 << 10, 2, 0);
  run<1, 6, 2>("1 x 1 KiB rows, 32 KiB/block", buf, bytes, 1024, 32 << 10, 2, 0);
  run<8, 6, 2>("8 x 128 B dense, streaming", buf, bytes, 128, 2 << 20, 2, 0);
  run<4, 6, 2>("4 x 256 B dense, streaming", buf, bytes, 256, 2 << 20, 2, 0);
  run<1, 6, 2>("1 x 1 KiB rows, streaming", buf, bytes, 1024, 2 << 20, 2, 0);
  // ---- row stride (channel record size)
  run<16, 6, 2>("16 x 64 B, stride 512", buf, bytes, 512, 32 << 10, 2, 0);
  run<16, 6, 2>("16 x 64 B, stride 1024", buf, bytes, 1024, 32 << 10, 2, 0);
  run<16, 6, 2>("16 x 64 B, stride 2304 (weights)", buf, bytes, 2304, 32 << 10, 2, 0);
  run<16, 6, 2>("16 x 64 B, stride 64 (dense)", buf, bytes, 64, 32 << 10, 2, 0);
  // ---- depth / occupancy
  run<16, 6, 1>("1 stage in flight", buf, bytes, 256, 32 << 10, 2, 0);
  run<16, 3, 5>("12 KiB stages, 5 in flight", buf, bytes, 256, 32 << 10, 2, 0);
  run<16, 2, 8>("8 KiB stages, 8 in flight", buf, bytes, 256, 32 << 10, 2, 0);
  run<16, 6, 2>("one block per CU", buf, bytes, 256, 32 << 10, 1, 0);
  run<16, 6, 5>("one block per CU, 5 in flight (144 KiB)", buf, bytes, 256, 32 << 10, 1, 0);
  run<16, 6, 5>("one block per CU, 5 in flight, zero fill", buf, bytes, 256, 32 << 10, 1, 1);
  run<1, 6, 5>("one block per CU, 5 in flight, 1 KiB rows", buf, bytes, 1024, 32 << 10, 1, 0);
  return 0;
}
