// Batched non-max suppression for the detect validator / predictor (reference ultralytics/utils/ops.py:144-278, called
// from ultralytics/models/yolo/detect/val.py:62-70).  The reference loops over images in Python and hands each image's
// candidates to torchvision.ops.nms; here the whole batch is three launches:
//
//   1. dy_nms_candidates  every (anchor, class) pair with score > conf (multi_label) or every anchor's best class emits a
//                         64-bit key  (~score_bits << 32) | (anchor * nc + class).  Scores are positive floats, so the
//                         ascending key order is "score descending, candidate order ascending": exactly the stable
//                         descending sort of the reference's candidate list (torch.where order = row-major (anchor, class)).
//   2. dy_nms_sort        one segmented radix sort (rocPRIM) over the B per-image segments.
//   3. dy_nms_greedy      one block per image: the first min(n, max_nms) candidates are materialised as class-offset xyxy
//                         boxes, then the greedy scan keeps the next live box and kills the boxes it overlaps (IoU > thr).
//                         Only rows of KEPT boxes are ever evaluated (<= max_det rows x n columns) instead of the n x n
//                         mask matrix of the classic GPU NMS, because the reference truncates to max_det anyway.
//
// IoU arithmetic is torchvision's CPU kernel's: class offset added in f32 (ops.py:259), areas / intersection / quotient in f32 with
// one rounding per operation, quotient widened for the comparison with the double threshold.
#include <hipcub/hipcub.hpp>

#include "dy_common.h"

namespace {

constexpr int NMS_THREADS = 1024;

__global__ void nms_candidates_kernel(const float* __restrict__ pred, int nc, int A, float conf, int multi_label,
                                      unsigned long long* __restrict__ keys, int* __restrict__ counts, long cap) {
  const int b = blockIdx.y;
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= A) return;
  const float* p = pred + (long)b * (4 + nc) * A + 4L * A + a;
  unsigned long long* kb = keys + (long)b * cap;
  if (multi_label) {
    for (int j = 0; j < nc; ++j) {
      const float s = p[(long)j * A];
      if (s > conf) {
        const int slot = atomicAdd(&counts[b], 1);
        if (slot < cap) kb[slot] = ((unsigned long long)(~__float_as_uint(s)) << 32) | (unsigned)(a * nc + j);
      }
    }
  } else {
    float best = p[0];
    int bj = 0;
    for (int j = 1; j < nc; ++j) {
      const float s = p[(long)j * A];
      if (s > best) { best = s; bj = j; }        // first maximum, like torch.max
    }
    if (best > conf) {
      const int slot = atomicAdd(&counts[b], 1);
      if (slot < cap) kb[slot] = ((unsigned long long)(~__float_as_uint(best)) << 32) | (unsigned)(a * nc + bj);
    }
  }
}

__global__ void nms_offsets_kernel(const int* __restrict__ counts, int* __restrict__ seg_begin, int* __restrict__ seg_end, int B,
                                   long cap) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const long c = counts[b] < cap ? counts[b] : cap;
  seg_begin[b] = (int)(b * cap);
  seg_end[b] = (int)(b * cap + c);
}

// (HIP's __f*_rn intrinsics are plain operators on AMD targets: csrc/Makefile builds this file with -ffp-contract=off)
// torchvision's CPU kernel (nms_kernel_impl<float>) evaluates areas, intersection and the quotient in the INPUT dtype, every
// operation rounded on its own (no FMA in its generic x86-64 build), and widens only the quotient for the comparison against the
// double threshold.  The explicit _rn intrinsics keep hipcc from contracting or re-associating.
__device__ inline float nms_area(const float4 b) { return __fmul_rn(__fsub_rn(b.z, b.x), __fsub_rn(b.w, b.y)); }

__device__ inline bool nms_overlaps(const float4 bi, const float area_i, const float4 bj, const double thr) {
  const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
  const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
  const float w = fmaxf(0.f, __fsub_rn(xx2, xx1)), h = fmaxf(0.f, __fsub_rn(yy2, yy1));
  const float inter = __fmul_rn(w, h);
  const float ovr = __fdiv_rn(inter, __fsub_rn(__fadd_rn(area_i, nms_area(bj)), inter));
  return (double)ovr > thr;
}

__global__ __launch_bounds__(NMS_THREADS) void nms_greedy_kernel(const float* __restrict__ pred, const unsigned long long* __restrict__ keys,
                                                                 const int* __restrict__ counts, int nc, int A, long cap, double iou_thr,
                                                                 int max_nms, int max_det, float max_wh, int agnostic,
                                                                 float4* __restrict__ boxes_ws, unsigned char* __restrict__ dead_ws,
                                                                 float* __restrict__ out, long long* __restrict__ keep_idx,
                                                                 int* __restrict__ out_counts) {
  const int b = blockIdx.x, tid = threadIdx.x;
  long n = counts[b] < cap ? counts[b] : cap;
  if (n > max_nms) n = max_nms;
  const unsigned long long* kb = keys + (long)b * cap;
  const float* pb = pred + (long)b * (4 + nc) * A;
  float4* boxes = boxes_ws + (long)b * max_nms;
  unsigned char* dead = dead_ws + (long)b * max_nms;
  // class-offset boxes in sorted order (ops.py:236 xywh2xyxy, :259 boxes + cls * max_wh, all f32)
  for (long i = tid; i < n; i += NMS_THREADS) {
    const unsigned idx = (unsigned)(kb[i] & 0xffffffffull);
    const int a = idx / nc, j = idx % nc;
    const float x = pb[a], y = pb[A + a], hw = pb[2L * A + a] / 2, hh = pb[3L * A + a] / 2;
    const float off = agnostic ? 0.f : (float)j * max_wh;
    boxes[i] = make_float4((x - hw) + off, (y - hh) + off, (x + hw) + off, (y + hh) + off);
    dead[i] = 0;
  }
  __syncthreads();
  __shared__ int s_next;
  int kept = 0;
  long cur = 0;
  while (kept < max_det && cur < n) {
    // next live candidate at or after cur
    if (tid == 0) s_next = 0x7fffffff;
    __syncthreads();
    for (long base = cur; base < n; base += NMS_THREADS) {
      const long i = base + tid;
      if (i < n && !dead[i]) atomicMin(&s_next, (int)i);
      __syncthreads();
      const int found = s_next;
      __syncthreads();                             // nobody may start the next window's atomicMin before all have read
      if (found != 0x7fffffff) break;
    }
    const int i = s_next;
    __syncthreads();
    if (i == 0x7fffffff) break;
    const float4 bi = boxes[i];
    if (tid == 0) {
      const unsigned idx = (unsigned)(kb[i] & 0xffffffffull);
      const int a = idx / nc, j = idx % nc;
      const float x = pb[a], y = pb[A + a], hw = pb[2L * A + a] / 2, hh = pb[3L * A + a] / 2;
      float* o = out + ((long)b * max_det + kept) * 6;
      o[0] = x - hw; o[1] = y - hh; o[2] = x + hw; o[3] = y + hh;
      o[4] = pb[(4L + j) * A + a];
      o[5] = (float)j;
      keep_idx[(long)b * max_det + kept] = idx;
    }
    const float area_i = nms_area(bi);
    for (long j = i + 1 + tid; j < n; j += NMS_THREADS)
      if (!dead[j] && nms_overlaps(bi, area_i, boxes[j], iou_thr)) dead[j] = 1;
    ++kept;
    cur = i + 1;
    __syncthreads();
  }
  if (tid == 0) out_counts[b] = kept;
}

}  // namespace

extern "C" int dy_nms_candidates(const float* pred, int B, int nc, int A, float conf_thres, int multi_label, uint64_t* keys,
                                 int* counts, int64_t cap, void* stream) {
  DY_CHECK(B >= 0 && nc > 0 && A > 0 && cap > 0, "dy_nms_candidates: bad sizes B=%d nc=%d A=%d cap=%ld", B, nc, A, (long)cap);
  DY_CHECK((int64_t)A * nc < (1LL << 32) && (int64_t)B * cap < (1LL << 31), "dy_nms_candidates: index range");
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(counts, 0, sizeof(int) * B, st);
  nms_candidates_kernel<<<dim3(dy_cdiv(A, 256), B), 256, 0, st>>>(pred, nc, A, conf_thres, multi_label && nc > 1,
                                                                   (unsigned long long*)keys, counts, cap);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_nms_sort(const uint64_t* keys, uint64_t* keys_sorted, const int* counts, int B, int64_t cap, void* workspace,
                           size_t* workspace_bytes, void* stream) {
  // workspace layout: [2*B ints segment offsets | rocPRIM temporary storage]; query with workspace == NULL
  hipStream_t st = (hipStream_t)stream;
  const size_t head = ((size_t)2 * B * sizeof(int) + 255) & ~(size_t)255;
  size_t tmp = 0;
  hipError_t e = hipcub::DeviceSegmentedRadixSort::SortKeys(nullptr, tmp, (const unsigned long long*)keys, (unsigned long long*)keys_sorted,
                                                            (int)(B * cap), B, (const int*)nullptr, (const int*)nullptr, 0, 64, st);
  DY_CHECK(e == hipSuccess, "dy_nms_sort: size query failed: %s", hipGetErrorString(e));
  if (workspace == nullptr) {
    *workspace_bytes = head + tmp;
    return 0;
  }
  DY_CHECK(*workspace_bytes >= head + tmp, "dy_nms_sort: workspace %zu < %zu bytes", *workspace_bytes, head + tmp);
  if (B == 0) return 0;
  int* seg_begin = (int*)workspace;
  int* seg_end = seg_begin + B;
  nms_offsets_kernel<<<dy_cdiv(B, 64), 64, 0, st>>>(counts, seg_begin, seg_end, B, cap);
  DY_LAUNCH_CHECK();
  e = hipcub::DeviceSegmentedRadixSort::SortKeys((char*)workspace + head, tmp, (const unsigned long long*)keys,
                                                 (unsigned long long*)keys_sorted, (int)(B * cap), B, seg_begin, seg_end, 0, 64, st);
  DY_CHECK(e == hipSuccess, "dy_nms_sort: sort failed: %s", hipGetErrorString(e));
  return 0;
}

extern "C" int dy_nms_greedy(const float* pred, const uint64_t* keys_sorted, const int* counts, int B, int nc, int A, int64_t cap,
                             double iou_thres, int max_nms, int max_det, float max_wh, int agnostic, float* boxes_ws,
                             uint8_t* dead_ws, float* out, int64_t* keep_idx, int* out_counts, void* stream) {
  DY_CHECK(B >= 0 && nc > 0 && A > 0 && cap > 0 && max_nms > 0 && max_det > 0, "dy_nms_greedy: bad sizes");
  DY_CHECK(((uintptr_t)boxes_ws & 15) == 0, "dy_nms_greedy: boxes workspace must be 16-byte aligned");
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  nms_greedy_kernel<<<B, NMS_THREADS, 0, st>>>(pred, (const unsigned long long*)keys_sorted, counts, nc, A, cap, iou_thres, max_nms,
                                               max_det, max_wh, agnostic, (float4*)boxes_ws, dead_ws, out, (long long*)keep_idx,
                                               out_counts);
  DY_LAUNCH_CHECK();
  return 0;
}
