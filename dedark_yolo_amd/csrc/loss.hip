// Detection loss + task-aligned assigner on the GPU.
// Replaces v8DetectionLoss / RcoveryDetectionLoss (reference ultralytics/utils/loss.py:103-193,388-416), BboxLoss
// (:51-84), TaskAlignedAssigner (ultralytics/utils/tal.py:12-243), bbox_iou (ultralytics/utils/metrics.py:75-128),
// make_anchors / dist2bbox (tal.py:246-271), Detect's eval decode (ultralytics/nn/modules/head.py:66-93) and the tensor
// part of preprocess_batch (ultralytics/models/yolo/detect/train.py:70-111).
//
// The Detect maps are NHWC, i.e. already [B, anchors, 64+nc] per level: the reference's cat/split/permute copies vanish.
// Integer outputs (target_gt_idx, fg_mask) follow the CPU reference's tie-breaking: argmax = first maximum; top-10 =
// libstdc++ std::partial_sort (heap-select) as torch.topk uses on CPU for dim >= 64*k, emulated exactly by one wave.
#include "dy_common.h"
#include "dy_lossmath.h"
#include "../../include/dedark_yolo.h"

namespace {

constexpr int REG = 16;
constexpr int TOPK = 10;

struct Maps {
  const char* map[3];
  long ld[3];
  int h[3], w[3], off[4];
  float stride[3];
  int B, nc, nl, A;
  // decoded-input mode of the assigner (dy_tal_assign_decoded = TaskAlignedAssigner.forward's own arguments): class probabilities
  // [B, A, nc] f32 and anchor points [A, 2] in pixels; the predicted boxes are then in pixels too.  Both null otherwise.
  const float* dec_scores;
  const float* dec_anchors;
};

__device__ inline void anchor_of(const Maps& m, int a, int& lvl, int& cell, float& ax, float& ay) {
  lvl = (a >= m.off[1]) + (a >= m.off[2] && m.nl > 2);
  if (m.nl == 1) lvl = 0;
  cell = a - m.off[lvl];
  int yy = cell / m.w[lvl], xx = cell - yy * m.w[lvl];
  ax = xx + 0.5f;
  ay = yy + 0.5f;
}

template <typename T>
__device__ inline const T* row_ptr(const Maps& m, int b, int lvl, int cell) {
  return reinterpret_cast<const T*>(m.map[lvl]) + ((long)b * m.h[lvl] * m.w[lvl] + cell) * m.ld[lvl];
}

// n consecutive channels of one anchor row as floats: 16-byte loads where the row allows it (rows start 16-byte aligned and
// c0 is a multiple of the vector width everywhere below); the per-element 2-byte loads made the loss kernels instruction-bound
template <typename T>
__device__ inline void load_run(const T* r, int c0, int n, long ld, float* out) {
  constexpr int VE = DT<T>::VE;
  int k = 0;
  if ((c0 % VE) == 0)
    for (; k + VE <= n && c0 + k + VE <= ld; k += VE) ldvec<T>(r + c0 + k, out + k);
  for (; k < n; ++k) out[k] = DT<T>::ld(r + c0 + k);
}

// ---- targets ---------------------------------------------------------------------------------------------------------
__global__ void prepare_targets_kernel(const float* __restrict__ bidx, const float* __restrict__ cls,
                                       const float* __restrict__ bb, int n, int B, int n_max, float W, float H,
                                       float* __restrict__ gt, int* __restrict__ counts) {
  for (int i = threadIdx.x; i < B * n_max * 5; i += blockDim.x) gt[i] = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) counts[i] = 0;
  __syncthreads();
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    int b = (int)bidx[t];
    if (b < 0 || b >= B) continue;
    int rank = 0;
    for (int u = 0; u < t; ++u) rank += ((int)bidx[u] == b);
    atomicAdd(&counts[b], 1);
    if (rank >= n_max) continue;
    float cx = bb[t * 4] * W, cy = bb[t * 4 + 1] * H, bw = bb[t * 4 + 2] * W, bh = bb[t * 4 + 3] * H;
    float* o = gt + ((long)b * n_max + rank) * 5;
    o[0] = cls[t];
    o[1] = cx - bw / 2;
    o[2] = cy - bh / 2;
    o[3] = cx + bw / 2;
    o[4] = cy + bh / 2;
  }
}

// ---- decode -------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void decode_kernel(Maps m, float* __restrict__ pred) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)m.B * m.A) return;
  int b = (int)(i / m.A), a = (int)(i - (long)b * m.A);
  int lvl, cell;
  float ax, ay;
  anchor_of(m, a, lvl, cell, ax, ay);
  const T* r = row_ptr<T>(m, b, lvl, cell);
  float d[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float x[REG], p[REG];
    load_run<T>(r, s * REG, REG, m.ld[lvl], x);
    d[s] = dy_softmax_expect(x, REG, p);
  }
  float* o = pred + i * 4;
  o[0] = ax - d[0];
  o[1] = ay - d[1];
  o[2] = ax + d[2];
  o[3] = ay + d[3];
}

// ---- assigner step 1: metrics + exact top-10 per (b, gt) ------------------------------------------------------------------
__device__ inline bool topk_comp(float x, float y) { return ((x != x) && !(y != y)) || (x > y); }

// libstdc++ __adjust_heap + __push_heap on (hv, hi)[0..len)
__device__ inline void heap_adjust(float* hv, int* hi, int hole, int len, float v, int idx) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (topk_comp(hv[child], hv[child - 1])) child--;
    hv[hole] = hv[child]; hi[hole] = hi[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    hv[hole] = hv[child - 1]; hi[hole] = hi[child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && topk_comp(hv[parent], v)) {
    hv[hole] = hv[parent]; hi[hole] = hi[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  hv[hole] = v; hi[hole] = idx;
}

// ---- torch.topk on rows shorter than 64 * k takes std::nth_element (ATen TopKImpl.h: partial_sort only when k * 64 <= n), whose
// treatment of equal values (the zero-metric anchors of a ground truth with fewer than 10 positive candidates) decides which of
// them become positives.  What follows is libstdc++'s introselect on (value, index) pairs, statement for statement as documented in
// <bits/stl_algo.h> / <bits/stl_heap.h> (__introselect, __unguarded_partition_pivot, __move_median_to_first, __unguarded_partition,
// __insertion_sort, __heap_select), run by one thread on LDS copies of the row: the first k entries afterwards are torch's top-k.
constexpr int SMALL_A = 64 * TOPK;                    // rows shorter than this take nth_element in torch

__device__ inline void nth_swap(float* v, int* x, int a, int b) {
  const float tv = v[a]; v[a] = v[b]; v[b] = tv;
  const int tx = x[a]; x[a] = x[b]; x[b] = tx;
}

__device__ inline void nth_median_to_first(float* v, int* x, int result, int a, int b, int c) {
  if (topk_comp(v[a], v[b])) {
    if (topk_comp(v[b], v[c])) nth_swap(v, x, result, b);
    else if (topk_comp(v[a], v[c])) nth_swap(v, x, result, c);
    else nth_swap(v, x, result, a);
  } else if (topk_comp(v[a], v[c])) nth_swap(v, x, result, a);
  else if (topk_comp(v[b], v[c])) nth_swap(v, x, result, c);
  else nth_swap(v, x, result, b);
}

__device__ inline int nth_partition(float* v, int* x, int first, int last, int pivot) {
  while (true) {
    while (topk_comp(v[first], v[pivot])) ++first;
    --last;
    while (topk_comp(v[pivot], v[last])) --last;
    if (!(first < last)) return first;
    nth_swap(v, x, first, last);
    ++first;
  }
}

__device__ inline void nth_insertion_sort(float* v, int* x, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    const float val = v[i];
    const int idx = x[i];
    if (topk_comp(val, v[first])) {
      for (int j = i; j > first; --j) { v[j] = v[j - 1]; x[j] = x[j - 1]; }
      v[first] = val; x[first] = idx;
    } else {
      int lastp = i, next = i - 1;
      while (topk_comp(val, v[next])) { v[lastp] = v[next]; x[lastp] = x[next]; lastp = next; --next; }
      v[lastp] = val; x[lastp] = idx;
    }
  }
}

// __heap_select(first, middle, last) on the sub-array starting at `first` (only reached when introselect runs out of depth)
__device__ inline void nth_heap_select(float* v, int* x, int first, int middle, int last) {
  const int len = middle - first;
  if (len >= 2)
    for (int parent = (len - 2) / 2; parent >= 0; --parent) heap_adjust(v + first, x + first, parent, len, v[first + parent], x[first + parent]);
  for (int i = middle; i < last; ++i)
    if (topk_comp(v[i], v[first])) {
      const float val = v[i];
      const int idx = x[i];
      v[i] = v[first]; x[i] = x[first];
      heap_adjust(v + first, x + first, 0, len, val, idx);
    }
}

__device__ inline void nth_element_pairs(float* v, int* x, int n, int nth) {
  int first = 0, last = n;
  int depth = 0;
  for (int t = n; t > 1; t >>= 1) ++depth;             // std::__lg(n)
  depth *= 2;
  while (last - first > 3) {
    if (depth == 0) {
      nth_heap_select(v, x, first, nth + 1, last);
      nth_swap(v, x, first, nth);
      return;
    }
    --depth;
    const int mid = first + (last - first) / 2;
    nth_median_to_first(v, x, first, first + 1, mid, last - 1);
    const int cut = nth_partition(v, x, first + 1, last, first);
    if (cut <= nth) first = cut;
    else last = cut;
  }
  nth_insertion_sort(v, x, first, last);
}

template <typename T>
__global__ __launch_bounds__(256) void tal_metrics_kernel(Maps m, const float* __restrict__ pred, const float* __restrict__ gt,
                                                           int n_max, float* __restrict__ align, float* __restrict__ overl,
                                                           int* __restrict__ cand, uint8_t* __restrict__ mask_pos) {
  __shared__ float hv[TOPK];
  __shared__ int hi[TOPK];
  __shared__ float s_nv[SMALL_A];                      // row copy for the nth_element path (rows shorter than 640 anchors)
  __shared__ int s_ni[SMALL_A];
  __shared__ int s_wave_cnt[4];
  __shared__ int s_total;
  __shared__ float s_v0;
  const int j = blockIdx.x, b = blockIdx.y;
  const float* g = gt + ((long)b * n_max + j) * 5;
  const float gx1 = g[1], gy1 = g[2], gx2 = g[3], gy2 = g[4];
  if (!(gx1 + gy1 + gx2 + gy2 > 0.f)) return;          // mask_gt false: row contributes nothing (loss.py:168)
  const int label = (int)g[0];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long base = ((long)b * n_max + j) * m.A;
  float* al = align + base;
  float* ov = overl + base;
  const float gb[4] = {gx1, gy1, gx2, gy2};
  for (int a = tid; a < m.A; a += 256) {
    int lvl = 0, cell = 0;
    float px, py, st = 1.f;
    if (m.dec_anchors) {
      px = m.dec_anchors[2 * a];
      py = m.dec_anchors[2 * a + 1];
    } else {
      float ax, ay;
      anchor_of(m, a, lvl, cell, ax, ay);
      st = m.stride[lvl];
      px = ax * st;
      py = ay * st;
    }
    float dmin = fminf(fminf(px - gx1, py - gy1), fminf(gx2 - px, gy2 - py));
    float metric = 0.f, o = 0.f;
    if (dmin > 1e-9f) {
      const float* pb = pred + ((long)b * m.A + a) * 4;
      float pbox[4] = {pb[0] * st, pb[1] * st, pb[2] * st, pb[3] * st};
      o = fmaxf(dy_ciou(gb, pbox), 0.f);
      const int lc = label < 0 ? 0 : (label >= m.nc ? m.nc - 1 : label);
      float sc;
      if (m.dec_scores) sc = m.dec_scores[((long)b * m.A + a) * m.nc + lc];
      else sc = dy_sigmoid(DT<T>::ld(row_ptr<T>(m, b, lvl, cell) + 4 * REG + lc));
      metric = powf(sc, 0.5f) * powf(o, 6.0f);
    }
    al[a] = metric;
    ov[a] = o;
  }
  __syncthreads();
  if (m.A < SMALL_A) {                                  // block-uniform
    for (int a = tid; a < m.A; a += 256) { s_nv[a] = al[a]; s_ni[a] = a; }
    __syncthreads();
    if (tid == 0) nth_element_pairs(s_nv, s_ni, m.A, TOPK - 1);
    __syncthreads();
    if (tid < TOPK) {
      const int a = s_ni[tid];
      float px, py;
      if (m.dec_anchors) {
        px = m.dec_anchors[2 * a];
        py = m.dec_anchors[2 * a + 1];
      } else {
        int lvl, cell;
        float ax, ay;
        anchor_of(m, a, lvl, cell, ax, ay);
        px = ax * m.stride[lvl];
        py = ay * m.stride[lvl];
      }
      const float dmin = fminf(fminf(px - gx1, py - gy1), fminf(gx2 - px, gy2 - py));
      if (dmin > 1e-9f) mask_pos[base + a] = 1;
    }
    return;
  }
  // initial heap = first TOPK elements (std::__make_heap), v0 = its minimum (heap top)
  if (tid == 0) {
    for (int i = 0; i < TOPK; ++i) { hv[i] = al[i]; hi[i] = i; }
    for (int parent = (TOPK - 2) / 2; parent >= 0; --parent) {
      float v = hv[parent]; int id = hi[parent];
      heap_adjust(hv, hi, parent, TOPK, v, id);
    }
    s_v0 = hv[0];
    s_total = 0;
  }
  __syncthreads();
  // ordered compaction of the elements that can ever enter the heap: comp(v, v0)
  const float v0 = s_v0;
  int* cl = cand + base;
  // (thread t owns the contiguous anchors [TOPK + t*chunk, +chunk): count, one block-wide exclusive scan, write -- the ordered
  //  compaction used to be 33 rounds of ballot + three block barriers each)
  {
    const int chunk = (m.A - TOPK + 255) / 256;
    const int a_lo = TOPK + tid * chunk, a_hi = min(a_lo + chunk, m.A);
    int cnt = 0;
    for (int a = a_lo; a < a_hi; ++a) cnt += topk_comp(al[a], v0) ? 1 : 0;
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (lane >= o) incl += up;
    }
    if (lane == 63) s_wave_cnt[wave] = incl;
    __syncthreads();
    int off = incl - cnt;
    for (int w = 0; w < wave; ++w) off += s_wave_cnt[w];
    for (int a = a_lo; a < a_hi; ++a)
      if (topk_comp(al[a], v0)) cl[off++] = a;
    if (tid == 255) s_total = off;                        // the last thread's end = total
    __syncthreads();
  }
  // sequential heap-select over the candidates (wave 0; lanes broadcast candidates, lane 0 mutates the heap).  The heap minimum
  // only grows, so a candidate that does not beat the minimum at the start of its 64-wide chunk never will: only the others are
  // visited (a large box has thousands of candidates and 10 winners).
  if (wave == 0) {
    const int total = s_total;
    for (int c0 = 0; c0 < total; c0 += 64) {
      int idx = (c0 + lane < total) ? cl[c0 + lane] : -1;
      float val = idx >= 0 ? al[idx] : 0.f;
      unsigned long long todo = __ballot(idx >= 0 && topk_comp(val, hv[0]));
      while (todo) {
        const int t = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        float vt = __shfl(val, t, 64);
        int it = __shfl(idx, t, 64);
        if (topk_comp(vt, hv[0])) {
          if (lane == 0) heap_adjust(hv, hi, 0, TOPK, vt, it);   // __pop_heap: replace the top with the new value
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      }
    }
    // mask_pos = topk & in_gts & mask_gt (tal.py:129-139)
    if (lane < TOPK) {
      int a = hi[lane];
      float px, py;
      if (m.dec_anchors) {
        px = m.dec_anchors[2 * a];
        py = m.dec_anchors[2 * a + 1];
      } else {
        int lvl, cell;
        float ax, ay;
        anchor_of(m, a, lvl, cell, ax, ay);
        px = ax * m.stride[lvl];
        py = ay * m.stride[lvl];
      }
      float dmin = fminf(fminf(px - gx1, py - gy1), fminf(gx2 - px, gy2 - py));
      if (dmin > 1e-9f) mask_pos[base + a] = 1;
    }
  }
}

// ---- assigner step 2: anchors claimed by several gts keep the max-CIoU one; target_gt_idx / fg_mask ---------------------
__global__ void tal_resolve_kernel(int B, int A, int n_max, const int* __restrict__ counts, const float* __restrict__ overl,
                                   uint8_t* __restrict__ mask_pos, int* __restrict__ gt_idx, uint8_t* __restrict__ fg) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)B * A) return;
  int b = (int)(i / A), a = (int)(i - (long)b * A);
  int n = counts[b] < n_max ? counts[b] : n_max;
  int cnt = 0, first = 0;
  for (int j = n - 1; j >= 0; --j)
    if (mask_pos[((long)b * n_max + j) * A + a]) { cnt++; first = j; }
  if (cnt > 1) {
    float best = -INFINITY;
    int bj = 0;
    for (int j = 0; j < n; ++j) {
      float o = overl[((long)b * n_max + j) * A + a];
      if (o > best) { best = o; bj = j; }
    }
    for (int j = 0; j < n; ++j) mask_pos[((long)b * n_max + j) * A + a] = (j == bj);
    first = bj;
    cnt = 1;
  }
  gt_idx[i] = cnt ? first : 0;
  fg[i] = cnt ? 1 : 0;
}

// ---- assigner step 3: per-gt maxima of align*mask and overlap*mask ---------------------------------------------------------
__global__ __launch_bounds__(256) void tal_posmax_kernel(int A, int n_max, const float* __restrict__ align,
                                                          const float* __restrict__ overl, const uint8_t* __restrict__ mask_pos,
                                                          float* __restrict__ pos) {
  __shared__ float sm[2][4];
  const int j = blockIdx.x, b = blockIdx.y;
  const long base = ((long)b * n_max + j) * A;
  float ma = 0.f, mo = 0.f;
  for (int a = threadIdx.x; a < A; a += 256)
    if (mask_pos[base + a]) { ma = fmaxf(ma, align[base + a]); mo = fmaxf(mo, overl[base + a]); }
  ma = wave_max(ma); mo = wave_max(mo);
  if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = ma; sm[1][threadIdx.x >> 6] = mo; }
  __syncthreads();
  if (threadIdx.x == 0) {
    pos[((long)b * n_max + j) * 2] = fmaxf(fmaxf(sm[0][0], sm[0][1]), fmaxf(sm[0][2], sm[0][3]));
    pos[((long)b * n_max + j) * 2 + 1] = fmaxf(fmaxf(sm[1][0], sm[1][1]), fmaxf(sm[1][2], sm[1][3]));
  }
}

// ---- assigner step 4: normalised score, label and box per anchor ---------------------------------------------------------------
__global__ void tal_targets_kernel(int B, int A, int n_max, const int* __restrict__ counts, const float* __restrict__ gt,
                                   const float* __restrict__ align, const uint8_t* __restrict__ mask_pos,
                                   const float* __restrict__ pos, const int* __restrict__ gt_idx, const uint8_t* __restrict__ fg,
                                   float* __restrict__ norm, int* __restrict__ label, float* __restrict__ tbox) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)B * A) return;
  int b = (int)(i / A), a = (int)(i - (long)b * A);
  int n = counts[b] < n_max ? counts[b] : n_max;
  float nm = 0.f;
  for (int j = 0; j < n; ++j) {
    long r = (long)b * n_max + j;
    if (mask_pos[r * A + a]) nm = fmaxf(nm, align[r * A + a] * pos[r * 2 + 1] / (pos[r * 2] + 1e-9f));
  }
  norm[i] = fg[i] ? nm : 0.f;
  const float* g = gt + ((long)b * n_max + gt_idx[i]) * 5;
  int lb = n_max > 0 ? (int)g[0] : 0;
  label[i] = lb < 0 ? 0 : lb;
#pragma unroll
  for (int k = 0; k < 4; ++k) tbox[i * 4 + k] = n_max > 0 ? g[1 + k] : 0.f;
}

// ---- loss forward ------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void loss_fwd_kernel(Maps m, const float* __restrict__ pred, const uint8_t* __restrict__ fg,
                                                        const float* __restrict__ norm, const int* __restrict__ label,
                                                        const float* __restrict__ tbox, double* acc) {
  __shared__ float sm[20];
  float a_ts = 0.f, a_bce = 0.f, a_iou = 0.f, a_dfl = 0.f;
  // grid-stride: a few hundred blocks instead of one per 256 anchors -- every block ends with four f64 atomics on the SAME four
  // addresses, and 1,050 of them in a row were most of this kernel's time
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (long)m.B * m.A; i += (long)gridDim.x * blockDim.x) {
    int b = (int)(i / m.A), a = (int)(i - (long)b * m.A);
    int lvl, cell;
    float ax, ay;
    anchor_of(m, a, lvl, cell, ax, ay);
    const T* r = row_ptr<T>(m, b, lvl, cell);
    const bool f = fg[i] != 0;
    const float wgt = f ? norm[i] : 0.f;
    const int lb = label[i];
    a_ts += wgt;
    for (int c0 = 0; c0 < m.nc; c0 += 8) {
      float xc[8];
      const int n = m.nc - c0 < 8 ? m.nc - c0 : 8;
      load_run<T>(r, 4 * REG + c0, n, m.ld[lvl], xc);
      for (int e = 0; e < n; ++e) a_bce += dy_bce(xc[e], (f && c0 + e == lb) ? wgt : 0.f);
    }
    if (f) {
      const float st = m.stride[lvl];
      float tb[4] = {tbox[i * 4] / st, tbox[i * 4 + 1] / st, tbox[i * 4 + 2] / st, tbox[i * 4 + 3] / st};
      float pb[4] = {pred[i * 4], pred[i * 4 + 1], pred[i * 4 + 2], pred[i * 4 + 3]};
      a_iou += (1.f - dy_ciou(pb, tb)) * wgt;
      float tgt[4] = {ax - tb[0], ay - tb[1], tb[2] - ax, tb[3] - ay};
      float d = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float x[REG];
        load_run<T>(r, s * REG, REG, m.ld[lvl], x);
        float t = fminf(fmaxf(tgt[s], 0.f), (float)(REG - 1) - 0.01f);
        d += dy_dfl_side(x, t, nullptr, nullptr);
      }
      a_dfl += d * 0.25f * wgt;
    }
  }
  a_ts = block_sum(a_ts, sm);
  a_bce = block_sum(a_bce, sm);
  a_iou = block_sum(a_iou, sm);
  a_dfl = block_sum(a_dfl, sm);
  if (threadIdx.x == 0) {
    atomic_add_f64(acc + 0, (double)a_ts);
    atomic_add_f64(acc + 1, (double)a_bce);
    atomic_add_f64(acc + 2, (double)a_iou);
    atomic_add_f64(acc + 3, (double)a_dfl);
  }
}

__global__ void loss_finish_kernel(const double* acc, const float* recovery, float hb, float hc, float hd, float lrl, int B,
                                   float* loss_out, float* items) {
  if (threadIdx.x || blockIdx.x) return;
  float tss = fmaxf((float)acc[0], 1.f);
  float box = (float)acc[2] / tss * hb, cls = (float)acc[1] / tss * hc, dfl = (float)acc[3] / tss * hd;
  float rec = recovery ? recovery[0] : 0.f;
  float total = (box + cls + dfl) * (float)B;
  if (recovery) { total += lrl * rec; cls += lrl * rec; }
  loss_out[0] = total;
  items[0] = box; items[1] = cls; items[2] = dfl;
}

// ---- loss backward: d loss / d maps ----------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void loss_bwd_kernel(Maps m, char* d0, char* d1, char* d2, long dl0, long dl1, long dl2,
                                                        const float* __restrict__ pred, const uint8_t* __restrict__ fg,
                                                        const float* __restrict__ norm, const int* __restrict__ label,
                                                        const float* __restrict__ tbox, const double* __restrict__ acc,
                                                        const float* __restrict__ grad_out, float hb, float hc, float hd,
                                                        int pad_to) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)m.B * m.A) return;
  int b = (int)(i / m.A), a = (int)(i - (long)b * m.A);
  int lvl, cell;
  float ax, ay;
  anchor_of(m, a, lvl, cell, ax, ay);
  const T* r = row_ptr<T>(m, b, lvl, cell);
  char* dbase = lvl == 0 ? d0 : (lvl == 1 ? d1 : d2);
  long dld = lvl == 0 ? dl0 : (lvl == 1 ? dl1 : dl2);
  T* o = reinterpret_cast<T*>(dbase) + ((long)b * m.h[lvl] * m.w[lvl] + cell) * dld;
  const float tss = fmaxf((float)acc[0], 1.f);
  const float go = (grad_out ? grad_out[0] : 1.f) * (float)m.B / tss;
  const bool f = fg[i] != 0;
  const float wgt = f ? norm[i] : 0.f;
  const int lb = label[i];
  constexpr int VE = DT<T>::VE;
  const bool vec_ok = (pad_to % VE) == 0 && (dld * (long)sizeof(T)) % 16 == 0;      // always true for our padded NHWC gradient maps
  // class logits, then zeros up to the padded width: VE channels per 16-byte store
  for (int c0 = 4 * REG; c0 < pad_to; c0 += VE) {
    float x[VE], g[VE];
    const int n = 4 * REG + m.nc - c0;                  // real channels left in this vector (may be <= 0)
    load_run<T>(r, c0, n < VE ? (n > 0 ? n : 0) : VE, m.ld[lvl], x);
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      const int c = c0 + e - 4 * REG;
      g[e] = (e < n) ? go * hc * (dy_sigmoid(x[e]) - ((f && c == lb) ? wgt : 0.f)) : 0.f;
    }
    if (vec_ok) {
      stvec<T>(o + c0, g);
    } else {
      for (int e = 0; e < VE && c0 + e < pad_to; ++e) DT<T>::st(o + c0 + e, g[e]);
    }
  }
  if (!f) {
    float z[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) z[e] = 0.f;
    for (int c = 0; c < 4 * REG; c += VE) {
      if (vec_ok) stvec<T>(o + c, z);
      else
        for (int e = 0; e < VE; ++e) DT<T>::st(o + c + e, 0.f);
    }
    return;
  }
  const float st = m.stride[lvl];
  float tb[4] = {tbox[i * 4] / st, tbox[i * 4 + 1] / st, tbox[i * 4 + 2] / st, tbox[i * 4 + 3] / st};
  float pb[4] = {pred[i * 4], pred[i * 4 + 1], pred[i * 4 + 2], pred[i * 4 + 3]};
  float gc[4];
  dy_ciou_grad(pb, tb, gc);
  // L_box = (1 - ciou) * w / tss ; pred = (ax - l, ay - t, ax + r, ay + b)
  const float kb = -go * hb * wgt;
  float dd[4] = {-kb * gc[0], -kb * gc[1], kb * gc[2], kb * gc[3]};   // d L / d (l, t, r, b)
  float tgt[4] = {ax - tb[0], ay - tb[1], tb[2] - ax, tb[3] - ay};
  const float kd = go * hd * wgt * 0.25f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float x[REG], p[REG];
    load_run<T>(r, s * REG, REG, m.ld[lvl], x);
    float e = dy_softmax_expect(x, REG, p);
    float t = fminf(fmaxf(tgt[s], 0.f), (float)(REG - 1) - 0.01f);
    int tl = (int)t;
    float wl = (float)(tl + 1) - t, wr = 1.f - wl;
    float go_[REG];
#pragma unroll
    for (int k = 0; k < REG; ++k) {
      float gdfl = p[k] - (k == tl ? wl : 0.f) - (k == tl + 1 ? wr : 0.f);
      float gbox = p[k] * ((float)k - e) * dd[s];
      go_[k] = kd * gdfl + gbox;
    }
    if (vec_ok) {
#pragma unroll
      for (int k = 0; k < REG; k += VE) stvec<T>(o + s * REG + k, go_ + k);
    } else {
#pragma unroll
      for (int k = 0; k < REG; ++k) DT<T>::st(o + s * REG + k, go_[k]);
    }
  }
}

// ---- Detect eval decode ----------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void detect_decode_kernel(Maps m, float* __restrict__ y) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)m.B * m.A) return;
  int b = (int)(i / m.A), a = (int)(i - (long)b * m.A);
  int lvl, cell;
  float ax, ay;
  anchor_of(m, a, lvl, cell, ax, ay);
  const T* r = row_ptr<T>(m, b, lvl, cell);
  float d[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float x[REG], p[REG];
    load_run<T>(r, s * REG, REG, m.ld[lvl], x);
    d[s] = dy_softmax_expect(x, REG, p);
  }
  const float st = m.stride[lvl];
  float x1 = ax - d[0], y1 = ay - d[1], x2 = ax + d[2], y2 = ay + d[3];
  float* o = y + (long)b * (4 + m.nc) * m.A + a;
  o[0] = (x1 + x2) / 2 * st;
  o[(long)m.A] = (y1 + y2) / 2 * st;
  o[2L * m.A] = (x2 - x1) * st;
  o[3L * m.A] = (y2 - y1) * st;
  for (int c = 0; c < m.nc; ++c) o[(long)(4 + c) * m.A] = dy_sigmoid(DT<T>::ld(r + 4 * REG + c));
}

// uint8 pixels take 256 values: every block tabulates clean = v/255 and img = clean^gamma once (the exact libm powf of the
// scalar version, so results are unchanged) and then streams 16 pixels per thread and iteration (one 16-byte load, four
// 16-byte stores per output).  The per-pixel powf made this kernel ALU-bound (0.19 ms for 39 MB in / 315 MB out).
__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ img, float* __restrict__ img_out,
                                                          float* __restrict__ clean_out, float gamma, int lowlight, int dedark,
                                                          double* mse_acc, long n) {
  __shared__ float sm[20];
  __shared__ float t_img[256], t_clean[256], t_d2[256];
  {
    const int v = threadIdx.x;
    float clean = (float)v / 255.f;
    float im;
    if (dedark && lowlight) { clean = powf(clean, gamma); im = clean; }
    else if (lowlight) im = powf(clean, gamma);
    else im = clean;
    t_img[v] = im;
    t_clean[v] = clean;
    t_d2[v] = (im - clean) * (im - clean);
  }
  __syncthreads();
  float part = 0.f;
  const bool aligned = (uintptr_t)img % 16 == 0 && (uintptr_t)img_out % 16 == 0 && (!clean_out || (uintptr_t)clean_out % 16 == 0);
  const long nvec = aligned ? n / 16 : 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    const u32x4 raw = reinterpret_cast<const u32x4*>(img)[i];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t w = raw[q];
      const int v0 = w & 255, v1 = (w >> 8) & 255, v2 = (w >> 16) & 255, v3 = w >> 24;
      reinterpret_cast<f32x4*>(img_out)[4 * i + q] = f32x4{t_img[v0], t_img[v1], t_img[v2], t_img[v3]};
      if (clean_out) reinterpret_cast<f32x4*>(clean_out)[4 * i + q] = f32x4{t_clean[v0], t_clean[v1], t_clean[v2], t_clean[v3]};
      part += (t_d2[v0] + t_d2[v1]) + (t_d2[v2] + t_d2[v3]);
    }
  }
  for (long i = nvec * 16 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int v = img[i];
    img_out[i] = t_img[v];
    if (clean_out) clean_out[i] = t_clean[v];
    part += t_d2[v];
  }
  part = block_sum(part, sm);
  if (threadIdx.x == 0 && mse_acc) atomic_add_f64(mse_acc, (double)part);
}

int make_maps(const dy_det_maps* d, Maps& m, const char* who) {
  DY_CHECK(d && d->n_levels >= 1 && d->n_levels <= 3, "%s: bad maps", who);
  DY_CHECK(d->dtype == DY_F32 || d->dtype == DY_BF16 || d->dtype == DY_F16, "%s: bad dtype", who);
  m.B = d->B; m.nc = d->nc; m.nl = d->n_levels;
  m.dec_scores = nullptr; m.dec_anchors = nullptr;
  int off = 0;
  for (int l = 0; l < 3; ++l) {
    m.off[l] = off;
    if (l < d->n_levels) {
      DY_CHECK(d->map[l] && d->h[l] > 0 && d->w[l] > 0 && d->map_ld[l] >= 4 * REG + d->nc, "%s: bad level %d", who, l);
      m.map[l] = (const char*)d->map[l]; m.ld[l] = d->map_ld[l]; m.h[l] = d->h[l]; m.w[l] = d->w[l]; m.stride[l] = d->stride[l];
      off += d->h[l] * d->w[l];
    } else {
      m.map[l] = nullptr; m.ld[l] = 0; m.h[l] = 1; m.w[l] = 1; m.stride[l] = 1.f;
    }
  }
  m.off[3] = off;
  if (d->n_levels < 3) m.off[2] = off;
  if (d->n_levels < 2) m.off[1] = off;
  m.A = off;
  DY_CHECK(m.B > 0 && m.nc > 0 && m.A >= TOPK, "%s: empty problem (A=%d)", who, m.A);
  return 0;
}

}  // namespace

extern "C" int dy_loss_prepare_targets(const float* batch_idx, const float* cls, const float* bboxes, int n_targets, int B,
                                       int n_max, float img_w, float img_h, float* gt, int32_t* counts, void* stream) {
  DY_CHECK(gt && counts && B > 0 && n_max >= 0 && n_targets >= 0, "dy_loss_prepare_targets: bad args");
  DY_CHECK(n_targets == 0 || (batch_idx && cls && bboxes), "dy_loss_prepare_targets: null targets");
  prepare_targets_kernel<<<1, 256, 0, (hipStream_t)stream>>>(batch_idx, cls, bboxes, n_targets, B, n_max, img_w, img_h, gt, counts);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_loss_decode(const dy_det_maps* d, float* pred_boxes, void* stream) {
  Maps m;
  if (int e = make_maps(d, m, "dy_loss_decode")) return e;
  DY_CHECK(pred_boxes, "dy_loss_decode: null");
  int blocks = dy_cdiv((long)m.B * m.A, 256);
  if (d->dtype == DY_F32) decode_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>(m, pred_boxes);
  else if ((d->dtype) == DY_F16) decode_kernel<f16_t><<<blocks, 256, 0, (hipStream_t)stream>>>(m, pred_boxes);
  else decode_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>(m, pred_boxes);
  DY_LAUNCH_CHECK();
  return 0;
}

namespace {
int run_assigner(const Maps& m, int dtype, const float* pred_boxes, const float* gt, const int32_t* counts, int n_max, float* work_f,
                 int32_t* work_i, uint8_t* work_b, int32_t* target_gt_idx, uint8_t* fg_mask, float* norm, int32_t* target_label,
                 float* target_box, hipStream_t st) {
  const long BA = (long)m.B * m.A;
  if (n_max == 0) {          // tal.py:106-110
    (void)hipMemsetAsync(target_gt_idx, 0, BA * 4, st);
    (void)hipMemsetAsync(fg_mask, 0, BA, st);
    (void)hipMemsetAsync(norm, 0, BA * 4, st);
    (void)hipMemsetAsync(target_label, 0, BA * 4, st);
    (void)hipMemsetAsync(target_box, 0, BA * 16, st);
    return 0;
  }
  DY_CHECK(work_f && work_i && work_b, "dy_tal_assign: null work buffers");
  const long R = (long)m.B * n_max * m.A;
  float* align = work_f;
  float* overl = work_f + R;
  float* pos = work_f + 2 * R;       // [B*n_max*2]
  (void)hipMemsetAsync(work_b, 0, R, st);
  (void)hipMemsetAsync(work_f, 0, (2 * R + 2L * m.B * n_max) * sizeof(float), st);
  dim3 grid(n_max, m.B);
  if (dtype == DY_F32) tal_metrics_kernel<float><<<grid, 256, 0, st>>>(m, pred_boxes, gt, n_max, align, overl, work_i, work_b);
  else if ((dtype) == DY_F16) tal_metrics_kernel<f16_t><<<grid, 256, 0, st>>>(m, pred_boxes, gt, n_max, align, overl, work_i, work_b);
  else tal_metrics_kernel<bf16_t><<<grid, 256, 0, st>>>(m, pred_boxes, gt, n_max, align, overl, work_i, work_b);
  DY_LAUNCH_CHECK();
  int blocks = dy_cdiv(BA, 256);
  tal_resolve_kernel<<<blocks, 256, 0, st>>>(m.B, m.A, n_max, counts, overl, work_b, target_gt_idx, fg_mask);
  DY_LAUNCH_CHECK();
  tal_posmax_kernel<<<grid, 256, 0, st>>>(m.A, n_max, align, overl, work_b, pos);
  DY_LAUNCH_CHECK();
  tal_targets_kernel<<<blocks, 256, 0, st>>>(m.B, m.A, n_max, counts, gt, align, work_b, pos, target_gt_idx, fg_mask, norm,
                                             target_label, target_box);
  DY_LAUNCH_CHECK();
  return 0;
}
}  // namespace

extern "C" int dy_tal_assign(const dy_det_maps* d, const float* pred_boxes, const float* gt, const int32_t* counts, int n_max,
                             float* work_f, int32_t* work_i, uint8_t* work_b, int32_t* target_gt_idx, uint8_t* fg_mask, float* norm,
                             int32_t* target_label, float* target_box, void* stream) {
  Maps m;
  if (int e = make_maps(d, m, "dy_tal_assign")) return e;
  DY_CHECK(pred_boxes && gt && counts && target_gt_idx && fg_mask && norm && target_label && target_box, "dy_tal_assign: null");
  return run_assigner(m, d->dtype, pred_boxes, gt, counts, n_max, work_f, work_i, work_b, target_gt_idx, fg_mask, norm, target_label,
                      target_box, (hipStream_t)stream);
}

extern "C" int dy_tal_assign_decoded(const float* pd_scores, const float* pd_bboxes, const float* anc_points, const float* gt,
                                     const int32_t* counts, int B, int A, int nc, int n_max, float* work_f, int32_t* work_i,
                                     uint8_t* work_b, int32_t* target_gt_idx, uint8_t* fg_mask, float* norm, int32_t* target_label,
                                     float* target_box, void* stream) {
  DY_CHECK(pd_scores && pd_bboxes && anc_points && gt && counts && target_gt_idx && fg_mask && norm && target_label && target_box,
           "dy_tal_assign_decoded: null");
  DY_CHECK(B > 0 && nc > 0 && A >= TOPK && n_max >= 0, "dy_tal_assign_decoded: empty problem (A=%d)", A);
  Maps m;
  for (int l = 0; l < 3; ++l) { m.map[l] = nullptr; m.ld[l] = 0; m.h[l] = 1; m.w[l] = 1; m.stride[l] = 1.f; m.off[l] = l ? A : 0; }
  m.off[3] = A;
  m.B = B; m.nc = nc; m.nl = 1; m.A = A;
  m.dec_scores = pd_scores; m.dec_anchors = anc_points;
  return run_assigner(m, DY_F32, pd_bboxes, gt, counts, n_max, work_f, work_i, work_b, target_gt_idx, fg_mask, norm, target_label,
                      target_box, (hipStream_t)stream);
}

extern "C" int dy_loss_fwd(const dy_det_maps* d, const float* pred_boxes, const uint8_t* fg_mask, const float* norm,
                           const int32_t* target_label, const float* target_box, double* acc, void* stream) {
  Maps m;
  if (int e = make_maps(d, m, "dy_loss_fwd")) return e;
  DY_CHECK(pred_boxes && fg_mask && norm && target_label && target_box && acc, "dy_loss_fwd: null");
  int blocks = dy_cdiv((long)m.B * m.A, 256);
  if (blocks > 256) blocks = 256;
  if (d->dtype == DY_F32) loss_fwd_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>(m, pred_boxes, fg_mask, norm, target_label, target_box, acc);
  else if ((d->dtype) == DY_F16) loss_fwd_kernel<f16_t><<<blocks, 256, 0, (hipStream_t)stream>>>(m, pred_boxes, fg_mask, norm, target_label, target_box, acc);
  else loss_fwd_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>(m, pred_boxes, fg_mask, norm, target_label, target_box, acc);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_loss_finish(const double* acc, const float* recovery, float hyp_box, float hyp_cls, float hyp_dfl, float lrl,
                              int B, float* loss_out, float* items, void* stream) {
  DY_CHECK(acc && loss_out && items, "dy_loss_finish: null");
  loss_finish_kernel<<<1, 64, 0, (hipStream_t)stream>>>(acc, recovery, hyp_box, hyp_cls, hyp_dfl, lrl, B, loss_out, items);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_loss_bwd(const dy_det_maps* d, void* const dmap[3], const int64_t dmap_ld[3], const float* pred_boxes,
                           const uint8_t* fg_mask, const float* norm, const int32_t* target_label, const float* target_box,
                           const double* acc, const float* grad_out, float hyp_box, float hyp_cls, float hyp_dfl, void* stream) {
  Maps m;
  if (int e = make_maps(d, m, "dy_loss_bwd")) return e;
  DY_CHECK(dmap && dmap_ld && pred_boxes && fg_mask && norm && target_label && target_box && acc, "dy_loss_bwd: null");
  char* dp[3] = {nullptr, nullptr, nullptr};
  long dl[3] = {0, 0, 0};
  int pad_to = 1 << 30;
  for (int l = 0; l < m.nl; ++l) {
    DY_CHECK(dmap[l] && dmap_ld[l] >= 4 * REG + m.nc, "dy_loss_bwd: bad dmap %d", l);
    dp[l] = (char*)dmap[l];
    dl[l] = dmap_ld[l];
    if (dmap_ld[l] < pad_to) pad_to = (int)dmap_ld[l];
  }
  int blocks = dy_cdiv((long)m.B * m.A, 256);
  if (d->dtype == DY_F32)
    loss_bwd_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>(m, dp[0], dp[1], dp[2], dl[0], dl[1], dl[2], pred_boxes, fg_mask,
                                                                    norm, target_label, target_box, acc, grad_out, hyp_box, hyp_cls,
                                                                    hyp_dfl, pad_to);
  else if ((d->dtype) == DY_F16)
    loss_bwd_kernel<f16_t><<<blocks, 256, 0, (hipStream_t)stream>>>(m, dp[0], dp[1], dp[2], dl[0], dl[1], dl[2], pred_boxes, fg_mask,
                                                                     norm, target_label, target_box, acc, grad_out, hyp_box,
                                                                     hyp_cls, hyp_dfl, pad_to);
  else
    loss_bwd_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>(m, dp[0], dp[1], dp[2], dl[0], dl[1], dl[2], pred_boxes, fg_mask,
                                                                     norm, target_label, target_box, acc, grad_out, hyp_box,
                                                                     hyp_cls, hyp_dfl, pad_to);
  DY_LAUNCH_CHECK();
  return 0;
}

// ---- the two box-loss terms as stand-alone entries (reference API: metrics.bbox_iou(b1, b2, xywh=False, CIoU=True) and
// BboxLoss._df_loss(pred_dist, target)); the fused loss kernels above inline the same dy_lossmath.h functions
__global__ __launch_bounds__(256) void ciou_pairs_kernel(const float* __restrict__ b1, const float* __restrict__ b2, long n,
                                                         float* __restrict__ out, float* __restrict__ grad_b1) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a[4], b[4], g[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { a[k] = b1[i * 4 + k]; b[k] = b2[i * 4 + k]; }
  if (grad_b1) {
    out[i] = dy_ciou_grad(a, b, g);
#pragma unroll
    for (int k = 0; k < 4; ++k) grad_b1[i * 4 + k] = g[k];
  } else {
    out[i] = dy_ciou(a, b);
  }
}

__global__ __launch_bounds__(256) void iou_pairs_kernel(const float* __restrict__ b1, const float* __restrict__ b2, long n, int xywh, int kind,
                                                        float eps, float* __restrict__ out, float* __restrict__ grad_b1) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a[4], b[4], g[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { a[k] = b1[i * 4 + k]; b[k] = b2[i * 4 + k]; }
  out[i] = dy_box_iou_any(a, b, xywh, kind, eps, grad_b1 ? g : nullptr);
  if (grad_b1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) grad_b1[i * 4 + k] = g[k];
  }
}

// one thread per (box, side): CE at floor(t) / floor(t)+1 of 16 logits; out[box] = mean over the 4 sides (sides add up through
// a wave shuffle: 4 neighbouring lanes), grad = d sum(out) / d logits
__global__ __launch_bounds__(256) void dfl_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, long n_sides,
                                                  float* __restrict__ out, float* __restrict__ grad) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  float v = 0.f;
  if (i < n_sides) {
    float x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = pred[i * 16 + k];
    float wl;
    int tl;
    v = dy_dfl_side(x, tgt[i], &wl, &tl);
    if (grad) {
      float p[16];
      dy_softmax_expect(x, 16, p);
#pragma unroll
      for (int k = 0; k < 16; ++k) grad[i * 16 + k] = 0.25f * (p[k] - (k == tl ? wl : 0.f) - (k == tl + 1 ? 1.f - wl : 0.f));
    }
  }
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  if (i < n_sides && (i & 3) == 0) out[i >> 2] = 0.25f * v;
}

extern "C" int dy_bbox_ciou(const float* b1, const float* b2, int64_t n, float* out, float* grad_b1, void* stream) {
  DY_CHECK(n >= 0 && (n == 0 || (b1 && b2 && out)), "dy_bbox_ciou: null operand");
  if (n == 0) return 0;
  ciou_pairs_kernel<<<dy_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(b1, b2, n, out, grad_b1);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_bbox_iou(const float* b1, const float* b2, int64_t n, int xywh, int kind, float eps, float* out, float* grad_b1,
                           void* stream) {
  DY_CHECK(n >= 0 && (n == 0 || (b1 && b2 && out)), "dy_bbox_iou: null operand");
  DY_CHECK(kind >= 0 && kind <= 3, "dy_bbox_iou: kind %d (0 IoU, 1 GIoU, 2 DIoU, 3 CIoU)", kind);
  if (n == 0) return 0;
  iou_pairs_kernel<<<dy_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(b1, b2, n, xywh, kind, eps, out, grad_b1);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_dfl_loss(const float* pred_dist, const float* target, int64_t n_boxes, float* out, float* grad, void* stream) {
  DY_CHECK(n_boxes >= 0 && (n_boxes == 0 || (pred_dist && target && out)), "dy_dfl_loss: null operand");
  if (n_boxes == 0) return 0;
  dfl_kernel<<<dy_cdiv(n_boxes * 4, 256), 256, 0, (hipStream_t)stream>>>(pred_dist, target, n_boxes * 4, out, grad);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_detect_decode(const dy_det_maps* d, float* y, void* stream) {
  Maps m;
  if (int e = make_maps(d, m, "dy_detect_decode")) return e;
  DY_CHECK(y, "dy_detect_decode: null");
  int blocks = dy_cdiv((long)m.B * m.A, 256);
  if (d->dtype == DY_F32) detect_decode_kernel<float><<<blocks, 256, 0, (hipStream_t)stream>>>(m, y);
  else if ((d->dtype) == DY_F16) detect_decode_kernel<f16_t><<<blocks, 256, 0, (hipStream_t)stream>>>(m, y);
  else detect_decode_kernel<bf16_t><<<blocks, 256, 0, (hipStream_t)stream>>>(m, y);
  DY_LAUNCH_CHECK();
  return 0;
}

extern "C" int dy_preprocess_batch(const uint8_t* img, float* img_out, float* clean_out, float dark_param, int lowlight,
                                   int dedark, double* mse_acc, int64_t n, void* stream) {
  DY_CHECK(img && img_out && n > 0, "dy_preprocess_batch: bad args");
  long b = (n + 255) / 256;
  int blocks = (int)(b > 1024 ? 1024 : b);           // one f64 atomic per block on one address at the end
  preprocess_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(img, img_out, clean_out, dark_param, lowlight, dedark, mse_acc, n);
  DY_LAUNCH_CHECK();
  return 0;
}
