// Inference-side box selection on the GPU: score threshold + ordered compaction per head, then the cross-head class-wise greedy NMS.
// Follows YOLOv3PostProcessor._filter_single_head_boxes / apply_nms / _apply_nms / _cal_iou
// (/root/reference/yolov3/yolov3_post_process.py:45-77, 79-106, 109-131, 134-162); results are bit-identical to that NumPy code:
// float32 score arithmetic, float64 IoU arithmetic without fma contraction, np.where order, stable descending sort.
#include "common.h"
#include <math.h>

namespace {

constexpr int FB_THREADS = 1024;
constexpr int NMS_THREADS = 1024;
constexpr int NMS_MAX = 2048;   // candidates per image over the three heads that one workgroup sorts in LDS

// one workgroup per image: walks the H*W*B predictions in flat order, FB_THREADS at a time, and appends every prediction whose
// score exceeds the threshold (np.where order) as a row [x0/W, y0/H, x1/W, y1/H, conf, class prob, class index, score]
__global__ __launch_bounds__(FB_THREADS) void filter_boxes_kernel(const float* __restrict__ pred /*[N][P][L]*/, const float* __restrict__ boxes /*[N][P][4]*/,
                                                                  int P, int H, int W, int L, float thresh, int cap, int* __restrict__ counts /*[N]*/,
                                                                  float* __restrict__ rows /*[N][cap][8]*/, int* __restrict__ index /*[N][cap]*/) {
  __shared__ int wave_cnt[FB_THREADS / 64];
  __shared__ int base_s;
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* pr = pred + (size_t)n * P * L;
  const float* bx = boxes + (size_t)n * P * 4;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int p0 = 0; p0 < P; p0 += FB_THREADS) {
    const int i = p0 + tid;
    float conf = 0.f, prob = 1.f, score = 0.f;
    int arg = 0;
    bool hit = false;
    if (i < P) {
      const float* t = pr + (size_t)i * L;
      conf = t[4];
      score = conf;
      if (L > 5) {                                  // np.max / np.argmax over the class probabilities: first maximum wins
        prob = t[5];
        for (int k = 1; k < L - 5; ++k) {
          const float v = t[5 + k];
          if (v > prob) { prob = v; arg = k; }
        }
        score = prob * conf;                        // float32 product, as all_class_prob * all_score (:59)
      }
      hit = score > thresh;
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wave_cnt[wv] = __popcll(m);
    __syncthreads();
    int off = base_s;
    for (int k = 0; k < wv; ++k) off += wave_cnt[k];
    if (hit) {
      const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
      if (pos < cap) {
        float* r = rows + ((size_t)n * cap + pos) * 8;
        const float4 b = *reinterpret_cast<const float4*>(bx + (size_t)i * 4);
        r[0] = b.x / (float)W; r[1] = b.y / (float)H; r[2] = b.z / (float)W; r[3] = b.w / (float)H;   // float32 divisions (:66-69)
        r[4] = conf; r[5] = prob; r[6] = (float)arg; r[7] = score;
        index[(size_t)n * cap + pos] = i;
      }
    }
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int k = 0; k < FB_THREADS / 64; ++k) tot += wave_cnt[k];
      base_s += tot;
    }
    __syncthreads();
  }
  if (tid == 0) counts[n] = base_s;                 // may exceed cap: the caller checks
}

__device__ __forceinline__ uint32_t orderable(float f) {   // monotone float -> uint32
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// IoU exactly as _cal_iou / _overlap (:134-162) in float64, operation by operation
#pragma clang fp contract(off)
__device__ __forceinline__ bool iou_exceeds(const float* a, const float* b, double thresh) {
  const double a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
  const double w = fmin(a2, b2) - fmax(a0, b0);
  const double h = fmin(a3, b3) - fmax(a1, b1);
  double iou = 0.0;
  if (!(w <= 0.0 || h <= 0.0)) {
    const double inter = w * h;
    const double uni = (a2 - a0) * (a3 - a1) + (b2 - b0) * (b3 - b1) - inter;
    iou = inter / uni;
  }
  return iou > thresh;
}

// one workgroup per image.  Candidates = concat(head0, head1, head2) rows; ids are per-head positions (the reference never advances
// start_index, :81-89) or global positions when fixed_indices.  Sort by score descending (ties keep concatenation order), greedy
// suppression within a class, then a row is kept iff its id is among the survivors' ids (:100-104).
__global__ __launch_bounds__(NMS_THREADS) void nms_heads_kernel(const float* __restrict__ rows0, const float* __restrict__ rows1, const float* __restrict__ rows2,
                                                                const int* __restrict__ cnt0, const int* __restrict__ cnt1, const int* __restrict__ cnt2,
                                                                int cap, double thresh, int fixed_indices, uint8_t* __restrict__ keep0,
                                                                uint8_t* __restrict__ keep1, uint8_t* __restrict__ keep2, int* __restrict__ status) {
  __shared__ unsigned long long key[NMS_MAX];
  __shared__ float box[NMS_MAX][4];
  __shared__ int cls[NMS_MAX];
  __shared__ uint8_t alive[NMS_MAX];
  __shared__ uint8_t id_kept[NMS_MAX];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int c0 = cnt0[n], c1 = cnt1[n], c2 = cnt2[n];
  const int K = c0 + c1 + c2;
  uint8_t* k0 = keep0 + (size_t)n * cap;
  uint8_t* k1 = keep1 + (size_t)n * cap;
  uint8_t* k2 = keep2 + (size_t)n * cap;
  if (c0 > cap || c1 > cap || c2 > cap || K > NMS_MAX) {       // fail loudly: the host raises on a non-zero status
    if (tid == 0) atomicMax(status, K > NMS_MAX ? K : (c0 > c1 ? (c0 > c2 ? c0 : c2) : (c1 > c2 ? c1 : c2)));
    return;
  }
  int P2 = 1;
  while (P2 < K) P2 <<= 1;
  auto row_of = [&](int j) -> const float* {
    return j < c0 ? rows0 + ((size_t)n * cap + j) * 8 : j < c0 + c1 ? rows1 + ((size_t)n * cap + (j - c0)) * 8 : rows2 + ((size_t)n * cap + (j - c0 - c1)) * 8;
  };
  for (int j = tid; j < P2; j += NMS_THREADS) {
    unsigned long long kv = ~0ull;
    if (j < K) kv = ((unsigned long long)(~orderable(row_of(j)[7])) << 32) | (unsigned)j;
    key[j] = kv;
    id_kept[j] = 0;
  }
  __syncthreads();
  for (int k = 2; k <= P2; k <<= 1)                            // bitonic sort, ascending in (inverted score, position)
    for (int s = k >> 1; s > 0; s >>= 1) {
      for (int t = tid; t < P2; t += NMS_THREADS) {
        const int u = t ^ s;
        if (u > t) {
          const unsigned long long a = key[t], b = key[u];
          const bool up = (t & k) == 0;
          if ((a > b) == up) { key[t] = b; key[u] = a; }
        }
      }
      __syncthreads();
    }
  for (int r = tid; r < K; r += NMS_THREADS) {                 // sorted rank r <- candidate j
    const int j = (int)(key[r] & 0xffffffffu);
    const float* rw = row_of(j);
    box[r][0] = rw[0]; box[r][1] = rw[1]; box[r][2] = rw[2]; box[r][3] = rw[3];
    cls[r] = (int)rw[6];
    alive[r] = 1;
  }
  __syncthreads();
  for (int i = 0; i + 1 < K; ++i) {
    if (alive[i]) {                                            // uniform: written before the barrier that ended iteration i-1
      const int ci = cls[i];
      for (int j = i + 1 + tid; j < K; j += NMS_THREADS)
        if (alive[j] && cls[j] == ci && iou_exceeds(box[i], box[j], thresh)) alive[j] = 0;
    }
    __syncthreads();
  }
  for (int r = tid; r < K; r += NMS_THREADS)
    if (alive[r]) {
      const int j = (int)(key[r] & 0xffffffffu);
      const int id = fixed_indices ? j : (j < c0 ? j : j < c0 + c1 ? j - c0 : j - c0 - c1);
      id_kept[id] = 1;
    }
  __syncthreads();
  for (int j = tid; j < K; j += NMS_THREADS) {
    if (j < c0) k0[j] = id_kept[j];
    else if (j < c0 + c1) k1[j - c0] = id_kept[fixed_indices ? j : j - c0];
    else k2[j - c0 - c1] = id_kept[fixed_indices ? j : j - c0 - c1];
  }
}

}  // namespace

extern "C" int yolo_filter_boxes(const float* prediction, const float* boxes, int N, int H, int W, int B, int L, float score_thresh, int cap,
                                 int* counts, float* rows, int* index, void* stream) {
  YOLO_CHECK_ARG(prediction && boxes && counts && rows && index, "yolo_filter_boxes: null pointer");
  YOLO_CHECK_ARG(N > 0 && H > 0 && W > 0 && B > 0 && L >= 5 && cap > 0, "yolo_filter_boxes: bad shape");
  YOLO_CHECK_ARG((long long)H * W * B < (1ll << 30), "yolo_filter_boxes: head too large");
  hipLaunchKernelGGL(filter_boxes_kernel, dim3(N), dim3(FB_THREADS), 0, (hipStream_t)stream, prediction, boxes, H * W * B, H, W, L, score_thresh, cap,
                     counts, rows, index);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}

extern "C" int yolo_nms_max_candidates(void) { return NMS_MAX; }

extern "C" int yolo_nms_heads(const float* rows8, const float* rows16, const float* rows32, const int* count8, const int* count16, const int* count32,
                              int N, int cap, double nms_thresh, int fixed_indices, uint8_t* keep8, uint8_t* keep16, uint8_t* keep32, int* status,
                              void* stream) {
  YOLO_CHECK_ARG(rows8 && rows16 && rows32 && count8 && count16 && count32 && keep8 && keep16 && keep32 && status, "yolo_nms_heads: null pointer");
  YOLO_CHECK_ARG(N > 0 && cap > 0, "yolo_nms_heads: bad shape");
  hipLaunchKernelGGL(nms_heads_kernel, dim3(N), dim3(NMS_THREADS), 0, (hipStream_t)stream, rows8, rows16, rows32, count8, count16, count32, cap,
                     nms_thresh, fixed_indices, keep8, keep16, keep32, status);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}
