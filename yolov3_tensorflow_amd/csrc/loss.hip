// YOLOv3 loss forward + backward on gfx950, float32 throughout (compiled with -ffp-contract=off so the IoU / BCE
// arithmetic keeps the reference's operation order).
//
// Replaces, for the training step, yolov3/yolov3_decoder.py:119-192 (decode), yolov3/label_decoder.py:44-60 and
// yolov3/yolov3_loss.py:81-369 (IoU assignment, masks, xy/wh/conf/class terms, rectified prior loss) plus the TF autodiff of
// that graph, and removes the reference's sequential per-image tf.map_fn (yolov3_loss.py:111): every prediction of every image
// is one lane.
//
// Launch sequence of yolo_loss_fwd_bwd (all on the caller's stream):
//   1. assign  : one workgroup per image; per (GT, head) the response cell, the IoU of its B predicted boxes with the GT, the
//                first arg-max anchor (yolov3_loss.py:269-302) and the cross-head ">=" selection (:203-208).
//   2. main    : one lane per prediction: decode, max IoU over the image's GTs (:275-294), object / background masks
//                (:328-332), every loss term this prediction takes part in and the COMPLETE gradient of its L logits;
//                wave-cooperative coalesced stores of d(logits) (float32 and/or bf16); 6 block partial sums.
//   3. finalize: batch mean -> terms[6][3] (rows xy, wh, noobj, obj, class, rectified; columns /8,/16,/32), total, and the
//                rectified-image counter update (:125-130,152).
#include "common.h"

namespace {

struct LossCfg {
  yolo_loss_config c;
  int P[3];       // predictions per image per head = H*W*B
  float area[3];  // H*W as float
  int nb[3];      // workgroups of loss_main_kernel per image and head (its grid.x is their sum)
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float clipf_(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// IoU of a predicted corner box with a GT corner box, in the reference's operation order (yolov3_loss.py:276-293)
__device__ __forceinline__ float iou_ref(float px0, float py0, float px1, float py1, float parea, float tx0, float ty0, float tx1,
                                         float ty1, float tarea, int tiou) {
  const float iw = fmaxf(fminf(px1, tx1) - fmaxf(px0, tx0), 0.f);
  const float ih = fmaxf(fminf(py1, ty1) - fmaxf(py0, ty0), 0.f);
  const float inter = iw * ih;
  float iou = inter / (parea + tarea - inter);
  if (tiou) iou = iou * inter / tarea;
  return iou;
}

struct Box { float x0, y0, x1, y1, area, w, h, cx, cy; };

__device__ __forceinline__ Box decode_box(const float* t, int col, int row, float aw, float ah, float eps_lo, float eps_hi) {
  Box b;
  b.cx = clipf_(sigmoidf_(t[0]), eps_lo, eps_hi) + (float)col;  // yolov3_decoder.py:153-155
  b.cy = clipf_(sigmoidf_(t[1]), eps_lo, eps_hi) + (float)row;
  b.w = expf(t[2]) * aw;                                         // :167-168
  b.h = expf(t[3]) * ah;
  const float hw = b.w / 2, hh = b.h / 2;
  b.x0 = b.cx - hw; b.y0 = b.cy - hh; b.x1 = b.cx + hw; b.y1 = b.cy + hh;  // :137-139
  b.area = b.w * b.h;                                            // yolov3_loss.py:267
  return b;
}

// ---------------------------------------------------------------------------------------------------------------- 1. assign
__global__ __launch_bounds__(64) void loss_assign_kernel(LossCfg cfg, const float* __restrict__ l0, const float* __restrict__ l1,
                                                         const float* __restrict__ l2, const float* __restrict__ labels,
                                                         int* __restrict__ assign /*[N][T][3]*/, float* __restrict__ resp_iou /*[N][T][3]*/) {
  extern __shared__ float sh[];  // [T][3] response IoU, then [T][3] prediction index (as int)
  const yolo_loss_config& c = cfg.c;
  const int n = blockIdx.x, T = c.T;
  float* s_iou = sh;
  int* s_idx = reinterpret_cast<int*>(sh + T * 3);
  const float eps_lo = c.eps, eps_hi = 1.f - c.eps;
  for (int k = threadIdx.x; k < T * 3; k += blockDim.x) {
    const int t = k / 3, h = k - t * 3;
    const float* lab = labels + ((size_t)n * T + t) * 5;
    float best = -INFINITY;
    int bidx = -1;
    if (lab[0] >= 0.f) {  // yolov3_loss.py:239
      const int H = c.H[h], W = c.W[h], B = c.B[h];
      const float tx = lab[0] * (float)W, ty = lab[1] * (float)H;  // label_decoder.py:53
      const float tw = lab[2] * (float)W, th = lab[3] * (float)H;  // :54
      const float thw = tw / 2, thh = th / 2;
      const float tx0 = tx - thw, ty0 = ty - thh, tx1 = tx + thw, ty1 = ty + thh;  // :58-59
      const float tarea = tw * th;                                                 // yolov3_loss.py:273
      int col = (int)floorf(tx), row = (int)floorf(ty);                           // :269-270
      col = min(max(col, 0), W - 1);  // documented divergence: clamp instead of a failing gather_nd
      row = min(max(row, 0), H - 1);
      const float* base = (h == 0 ? l0 : (h == 1 ? l1 : l2)) + ((size_t)(n * H + row) * W + col) * c.ldc[h];
      int barg = 0;
      for (int b = 0; b < B; ++b) {
        const Box p = decode_box(base + b * c.L, col, row, c.anchor_w[h][b], c.anchor_h[h][b], eps_lo, eps_hi);
        const float v = iou_ref(p.x0, p.y0, p.x1, p.y1, p.area, tx0, ty0, tx1, ty1, tarea, c.is_tiou_recall);
        if (b == 0 || v > best) { best = v; barg = b; }  // first maximum (tf.arg_max)
      }
      bidx = (row * W + col) * B + barg;
    }
    s_iou[k] = best;
    s_idx[k] = bidx;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const float i0 = s_iou[t * 3], i1 = s_iou[t * 3 + 1], i2 = s_iou[t * 3 + 2];
    const bool valid = s_idx[t * 3] >= 0;
    const bool a0 = valid && i0 >= i1 && i0 >= i2;  // yolov3_loss.py:203-208 (ties go to several heads)
    const bool a1 = valid && i1 >= i0 && i1 >= i2;
    const bool a2 = valid && i2 >= i0 && i2 >= i1;
    int* o = assign + ((size_t)n * T + t) * 3;
    o[0] = a0 ? s_idx[t * 3] : -1;
    o[1] = a1 ? s_idx[t * 3 + 1] : -1;
    o[2] = a2 ? s_idx[t * 3 + 2] : -1;
    if (resp_iou) {
      float* r = resp_iou + ((size_t)n * T + t) * 3;
      r[0] = i0; r[1] = i1; r[2] = i2;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- 2. main
constexpr int LM_THREADS = 256;

// LDS budget (floats) of loss_main_kernel for T label slots and channel stride ldc
__host__ __device__ inline int lm_lds_floats(int T, int ldc) { return T * 10 + ldc / 2 + 4 * 64 * 8 + 4 * 6; }

// One wave = 64/B whole cells (lane -> (cell, anchor)), so the gradient block a wave produces is a run of complete logits rows
// and is written back with 16-byte (float32) / 8-byte (bf16) stores: lane -> 4 consecutive channels, via an LDS table
// channel -> (anchor, field).
__global__ __launch_bounds__(LM_THREADS) void loss_main_kernel(LossCfg cfg, const float* __restrict__ l0, const float* __restrict__ l1,
                                                               const float* __restrict__ l2, const float* __restrict__ labels,
                                                               const int* __restrict__ assign, const int* __restrict__ current_num,
                                                               float* __restrict__ d0, float* __restrict__ d1, float* __restrict__ d2,
                                                               bf16_t* __restrict__ e0, bf16_t* __restrict__ e1, bf16_t* __restrict__ e2,
                                                               float* __restrict__ partial /*[N][3][nbx][6]*/, int nbx, float inv_n) {
  extern __shared__ float sh[];
  const yolo_loss_config& c = cfg.c;
  // grid.x runs over the workgroups of head 0, then 1, then 2 (a (max, 3, N) grid left 55 % of the workgroups without cells at 416^2)
  const int h = (int)blockIdx.x < cfg.nb[0] ? 0 : ((int)blockIdx.x < cfg.nb[0] + cfg.nb[1] ? 1 : 2);
  const int bx = (int)blockIdx.x - (h == 0 ? 0 : (h == 1 ? cfg.nb[0] : cfg.nb[0] + cfg.nb[1]));
  const int n = blockIdx.y, T = c.T, L = c.L;
  const int H = c.H[h], W = c.W[h], B = c.B[h], ldc = c.ldc[h], HW = H * W;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // LDS: GT table [T][8] (x0,y0,x1,y1,area,valid,-,-), assignment [T], GT class [T], channel table [ldc] (u16),
  //      wave scratch [4][64][8] (g0..g4, softmax max, softmax sum, #responsible GTs or -1), reduction [4][6]
  float* s_gt = sh;
  int* s_as = reinterpret_cast<int*>(sh + T * 8);
  int* s_cls = reinterpret_cast<int*>(sh + T * 9);
  unsigned short* s_lut = reinterpret_cast<unsigned short*>(sh + T * 10);
  float* s_w = sh + T * 10 + ldc / 2 + wave * 64 * 8;
  float* s_red = sh + T * 10 + ldc / 2 + 4 * 64 * 8;
  const float fW = (float)W, fH = (float)H;
  const int C = L - 5;
  for (int t = threadIdx.x; t < T; t += LM_THREADS) {
    const float* lab = labels + ((size_t)n * T + t) * 5;
    const float tx = lab[0] * fW, ty = lab[1] * fH, tw = lab[2] * fW, th = lab[3] * fH;
    const float thw = tw / 2, thh = th / 2;
    s_gt[t * 8 + 0] = tx - thw; s_gt[t * 8 + 1] = ty - thh; s_gt[t * 8 + 2] = tx + thw; s_gt[t * 8 + 3] = ty + thh;
    s_gt[t * 8 + 4] = tw * th;
    s_gt[t * 8 + 5] = lab[0] >= 0.f ? 1.f : 0.f;
    s_as[t] = assign[((size_t)n * T + t) * 3 + h];
    s_cls[t] = (int)lab[4];
  }
  for (int ch = threadIdx.x; ch < ldc; ch += LM_THREADS) {
    unsigned short e = 0xFFFF;  // padding channel
    if (ch < B * L) {
      const int bb = ch / L, j = ch - bb * L;
      e = (unsigned short)((bb << 8) | (j < 5 ? j : 5));
    }
    s_lut[ch] = e;
  }
  __syncthreads();

  const float* lg = (h == 0 ? l0 : (h == 1 ? l1 : l2)) + (size_t)n * HW * ldc;
  float* dg = (h == 0 ? d0 : (h == 1 ? d1 : d2));
  bf16_t* eg = (h == 0 ? e0 : (h == 1 ? e1 : e2));
  if (dg) dg += (size_t)n * HW * ldc;
  if (eg) eg += (size_t)n * HW * ldc;
  const bool rect = c.rectified_coord_num >= 0 && current_num[0] <= c.rectified_coord_num;  // yolov3_loss.py:125
  const float eps_lo = c.eps, eps_hi = 1.f - c.eps;
  const float w_xy = c.w_xy[h], w_wh = c.w_wh[h], w_no = c.w_noobj[h], w_obj = c.w_obj[h], w_cls = c.w_cls[h], w_r = c.w_rect[h];

  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // xy, wh, noobj, obj, class, rectified (un-normalised sums)
  const int cpw = 64 / B;                            // whole cells per wave
  const int cell_base = (bx * (LM_THREADS / 64) + wave) * cpw;
  if (cell_base < HW) {  // wave-uniform
    const int cl = lane / B, b = lane - cl * B;
    const int cell = cell_base + cl;
    const bool pv = cl < cpw && cell < HW;
    const int pid = cell * B + b;
    float g[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float mx = 0.f, se = 1.f, nvalid = -1.f;
    bool resp = false;
    if (pv) {
      const int row = cell / W, col = cell - row * W;
      const float* t = lg + (size_t)cell * ldc + b * L;
      const float t4[5] = {t[0], t[1], t[2], t[3], t[4]};
      const Box p = decode_box(t4, col, row, c.anchor_w[h][b], c.anchor_h[h][b], eps_lo, eps_hi);
      const float sx = sigmoidf_(t4[0]), sy = sigmoidf_(t4[1]);
      const float sc = sigmoidf_(t4[4]);
      const float conf = clipf_(sc, eps_lo, eps_hi);          // yolov3_decoder.py:178-179
      const bool conf_pass = sc >= eps_lo && sc <= eps_hi;    // tf.clip_by_value gradient
      float max_iou = -INFINITY;
      int nresp = 0;
      for (int k = 0; k < T; ++k) {
        if (s_gt[k * 8 + 5] == 0.f) continue;
        const float v = iou_ref(p.x0, p.y0, p.x1, p.y1, p.area, s_gt[k * 8], s_gt[k * 8 + 1], s_gt[k * 8 + 2], s_gt[k * 8 + 3],
                                s_gt[k * 8 + 4], c.is_tiou_recall);
        max_iou = fmaxf(max_iou, v);                          // yolov3_loss.py:294
        if (s_as[k] == pid) {                                 // this prediction is responsible for GT k (:341-342)
          ++nresp;
          const float* lab = labels + ((size_t)n * T + k) * 5;
          const float tx = lab[0] * fW, ty = lab[1] * fH, tw = lab[2] * fW, th = lab[3] * fH;
          // obj (:344-347)
          float lo = -logf(conf), go;
          if (c.is_focal_loss) {
            const float om = 1.f - conf;
            lo = lo * (powf(om, c.focal_gamma) * c.focal_alpha);
            go = c.focal_alpha * (-powf(om, c.focal_gamma) / conf + c.focal_gamma * powf(om, c.focal_gamma - 1.f) * logf(conf)) * (sc * (1.f - sc));
          } else {
            go = -(1.f / conf) * (sc * (1.f - sc));
          }
          acc[3] += lo;
          if (conf_pass) g[4] += w_obj * go;
          // xy / wh (:350-359)
          const float scale = 2.f - tw * th / cfg.area[h];
          const float cix = floorf(tx), ciy = floorf(ty);
          const float txf = tx - cix, tyf = ty - ciy;
          const float pxf = p.cx - cix, pyf = p.cy - ciy;
          acc[0] += scale * (-(txf * logf(pxf) + (1.f - txf) * logf(1.f - pxf))) + scale * (-(tyf * logf(pyf) + (1.f - tyf) * logf(1.f - pyf)));
          if (sx >= eps_lo && sx <= eps_hi) g[0] += w_xy * scale * (-txf / pxf + (1.f - txf) / (1.f - pxf)) * (sx * (1.f - sx));
          if (sy >= eps_lo && sy <= eps_hi) g[1] += w_xy * scale * (-tyf / pyf + (1.f - tyf) / (1.f - pyf)) * (sy * (1.f - sy));
          const float dw_ = logf(tw) - logf(p.w), dh_ = logf(th) - logf(p.h);
          acc[1] += scale * (dw_ * dw_) + scale * (dh_ * dh_);
          g[2] += w_wh * scale * (-2.f * dw_);
          g[3] += w_wh * scale * (-2.f * dh_);
        }
      }
      // background (:331-338)
      if (nresp == 0 && max_iou < c.iou_thresh) {
        float ln = -logf(1.f - conf), gn;
        if (c.is_focal_loss) {
          ln = ln * powf(conf, c.focal_gamma);
          gn = (powf(conf, c.focal_gamma) / (1.f - conf) - c.focal_gamma * powf(conf, c.focal_gamma - 1.f) * logf(1.f - conf)) * (sc * (1.f - sc));
        } else {
          gn = (1.f / (1.f - conf)) * (sc * (1.f - sc));
        }
        acc[2] += ln;
        if (conf_pass) g[4] += w_no * gn;
      }
      if (rect) {  // yolov3_loss.py:153-162
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[5] += t4[j] * t4[j]; g[j] += w_r * 2.f * t4[j]; }
      }
#pragma unroll
      for (int j = 0; j < 5; ++j) g[j] *= inv_n;
      resp = nresp > 0 && C > 0;
    }
    // class term of a responsible prediction (:361-364): softmax statistics + loss here, gradient in the store pass.  Responsible lanes
    // are rare (<= T per image and head) and each needs max / sum over its C class logits: the WAVE reads them, 64 classes at a time
    // (a lane walking its own 80 logits twice kept its whole wave waiting for 160 dependent loads: 18 of the kernel's 64 us)
    for (unsigned long long todo = __ballot(resp); todo; todo &= todo - 1) {
      const int src = __ffsll((long long)todo) - 1;
      const int scl = src / B;
      const float* tc = lg + (size_t)(cell_base + scl) * ldc + (src - scl * B) * L + 5;
      float m_ = -INFINITY;
      for (int k = lane; k < C; k += 64) m_ = fmaxf(m_, tc[k]);
      m_ = wave_max(m_);
      float s_ = 0.f;
      for (int k = lane; k < C; k += 64) s_ += expf(tc[k] - m_);
      s_ = wave_sum(s_);
      if (lane == src) { mx = m_; se = s_; }
    }
    if (pv) {
      const float* t = lg + (size_t)cell * ldc + b * L;
      if (resp) {
        int nv = 0;
        for (int k = 0; k < T; ++k) {
          if (s_as[k] != pid) continue;
          const int cls = s_cls[k];
          if (cls >= 0 && cls < C) {
            const float pc = expf(t[5 + cls] - mx) / se;
            acc[4] += -logf(clipf_(pc, eps_lo, eps_hi));
            if (pc >= eps_lo && pc <= eps_hi) ++nv;            // tf.clip_by_value (yolov3_decoder.py:191) passes no gradient once the target's
          }                                                    // probability is clipped: that target then moves none of the class logits
        }
        nvalid = (float)nv;
      }
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) s_w[lane * 8 + j] = g[j];
    s_w[lane * 8 + 5] = mx; s_w[lane * 8 + 6] = se; s_w[lane * 8 + 7] = nvalid;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // same-wave LDS write -> read by other lanes
    __builtin_amdgcn_wave_barrier();
    // ---- store pass: ncell complete rows of ldc channels, 4 channels per lane ----
    const int ncell = min(cpw, HW - cell_base);
    const int q = ldc >> 2;  // lanes per row (ldc = 64 * 2^k)
    const float gscale = w_cls * inv_n;
    // ldc == 256 (the 80-class heads): a lane keeps its 4 channels for every row -- table entries read once, and a lane none of whose
    // channels is a box / confidence logit has nothing to read from LDS in a row without a responsible anchor (nearly all rows)
    const bool keep = q == 64;
    unsigned e_keep[4] = {0xFFFFu, 0xFFFFu, 0xFFFFu, 0xFFFFu};
    bool small = true;
    if (keep) {
      small = false;
#pragma unroll
      for (int u = 0; u < 4; ++u) { e_keep[u] = s_lut[lane * 4 + u]; small = small || (e_keep[u] != 0xFFFFu && (e_keep[u] & 0xFF) < 5); }
    }
    for (int idx = lane; idx < ncell * q; idx += 64) {
      const int clx = keep ? (idx >> 6) : idx / q, c4 = keep ? lane : idx - clx * q;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      bool any = false;                                        // does any anchor of this cell carry a class gradient?  (broadcast reads)
      for (int bb = 0; bb < B; ++bb) any = any || s_w[(clx * B + bb) * 8 + 7] >= 0.f;
      if (any || small)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ch = c4 * 4 + u;
        const unsigned e = keep ? e_keep[u] : (unsigned)s_lut[ch];
        if (e != 0xFFFFu) {
          const int bb = e >> 8, j = e & 0xFF;
          const int pl = clx * B + bb;
          if (j < 5) {
            v[u] = s_w[pl * 8 + j];
          } else if (any && s_w[pl * 8 + 7] >= 0.f) {        // class logit of a responsible prediction: w*(n*softmax - counts)/N
            const int k = ch - bb * L - 5;
            const int rpid = (cell_base + clx) * B + bb;
            const float sm = expf(lg[(size_t)(cell_base + clx) * ldc + ch] - s_w[pl * 8 + 5]) / s_w[pl * 8 + 6];
            float cnt = 0.f;
            if (sm >= eps_lo && sm <= eps_hi)                // targets of class k whose (this) probability is not clipped
              for (int t2 = 0; t2 < T; ++t2) cnt += (s_as[t2] == rpid && s_cls[t2] == k) ? 1.f : 0.f;
            v[u] = gscale * (s_w[pl * 8 + 7] * sm - cnt);
          }
        }
      }
      const size_t off = (size_t)(cell_base + clx) * ldc + c4 * 4;
      if (dg) *reinterpret_cast<float4*>(dg + off) = make_float4(v[0], v[1], v[2], v[3]);
      if (eg) {
        uint2 o;
        const float gs16 = c.grad_scale16;            // loss scaling of the 16-bit gradient copy that feeds the backward pass
        o.x = pack_bf2(v[0] * gs16, v[1] * gs16);
        o.y = pack_bf2(v[2] * gs16, v[3] * gs16);
        *reinterpret_cast<uint2*>(eg + off) = o;
      }
    }
  }
  // ---- block partial sums (un-weighted sums; weights and 1/N applied in finalize) ----
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float s = wave_sum(acc[k]);
    if (lane == 0) s_red[wave * 6 + k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const float s = s_red[threadIdx.x] + s_red[6 + threadIdx.x] + s_red[12 + threadIdx.x] + s_red[18 + threadIdx.x];
    partial[(((size_t)n * 3 + h) * nbx + bx) * 6 + threadIdx.x] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------- 3. finalize
// output o = term k = o / 3 of head h = o % 3, reduced over all images and blocks
__global__ __launch_bounds__(1024) void loss_finalize_kernel(LossCfg cfg, const float* __restrict__ partial, int N, int nbx, float inv_n,
                                                            int batch_global, int* __restrict__ current_num, float* __restrict__ terms /*[6][3]*/,
                                                            float* __restrict__ total) {
  __shared__ float s_w[16][18];
  __shared__ float s_t[18];
  const yolo_loss_config& c = cfg.c;
  const bool rect = c.rectified_coord_num >= 0 && current_num[0] <= c.rectified_coord_num;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // This launch is latency, not work: a thread takes whole rows (image, head, block) of 6 terms, so that every load of the launch is in
  // flight after one or two trips (18 waves walking one output each needed ~8 dependent round trips: 11.9 us).  Rows of head h beyond
  // its own nb[h] workgroups are never written and never read.
  const int r0 = N * cfg.nb[0], r1 = r0 + N * cfg.nb[1], r2 = r1 + N * cfg.nb[2];
  float acc[3][6];
#pragma unroll
  for (int h = 0; h < 3; ++h)
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[h][k] = 0.f;
  for (int r = threadIdx.x; r < r2; r += 1024) {
    const int h = r < r0 ? 0 : (r < r1 ? 1 : 2);
    const int i = r - (h == 0 ? 0 : (h == 1 ? r0 : r1));
    const int nbh = cfg.nb[h];
    const int n = i / nbh, b = i - n * nbh;
    const float2* src = reinterpret_cast<const float2*>(partial + (((size_t)n * 3 + h) * nbx + b) * 6);
    const float2 a = src[0], b2 = src[1], d = src[2];
    const float v[6] = {a.x, a.y, b2.x, b2.y, d.x, d.y};
#pragma unroll
    for (int k = 0; k < 6; ++k) {           // (static indexing: the accumulators stay in registers)
      if (h == 0) acc[0][k] += v[k];
      else if (h == 1) acc[1][k] += v[k];
      else acc[2][k] += v[k];
    }
  }
#pragma unroll
  for (int h = 0; h < 3; ++h)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const float s = wave_sum(acc[h][k]);
      if (lane == 0) s_w[wave][k * 3 + h] = s;
    }
  __syncthreads();
  if (threadIdx.x < 18) {
    const int o = threadIdx.x, k = o / 3, h = o - k * 3;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += s_w[w][o];
    const float wt = (k == 0 ? c.w_xy[h] : k == 1 ? c.w_wh[h] : k == 2 ? c.w_noobj[h] : k == 3 ? c.w_obj[h] : k == 4 ? c.w_cls[h] : c.w_rect[h]);
    float v = wt * s * inv_n;
    if (k == 5 && !rect) v = 0.f;
    s_t[o] = v;
    terms[o] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 18; ++i) t += s_t[i];
    total[0] = t;
    if (rect) current_num[0] += batch_global;  // yolov3_loss.py:152 (advances only while the rectified branch is taken)
  }
}

// ---------------------------------------------------------------------------------------------------------------- decode (inference)
// YOLOv3Decoder._decode_single_head (yolov3_decoder.py:119-192) for one head: one lane per prediction.
__global__ __launch_bounds__(256) void decode_head_kernel(const float* __restrict__ logits, int N, int H, int W, int B, int L, int ldc,
                                                          const float* __restrict__ anchors /*[B][2] grid units (w,h)*/, float eps,
                                                          float* __restrict__ decoded /*[N][H][W][B][L]*/, float* __restrict__ boxes /*[.][4]*/,
                                                          float* __restrict__ score /*[N][H][W][B]*/, int* __restrict__ cls_idx) {
  const size_t total = (size_t)N * H * W * B;
  const float eps_lo = eps, eps_hi = 1.f - eps;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int b = (int)(i % B);
    const size_t cellg = i / B;
    const int col = (int)(cellg % W);
    const int row = (int)((cellg / W) % H);
    const float* t = logits + cellg * ldc + (size_t)b * L;
    const Box p = decode_box(t, col, row, anchors[b * 2], anchors[b * 2 + 1], eps_lo, eps_hi);
    const float conf = clipf_(sigmoidf_(t[4]), eps_lo, eps_hi);
    float* d = decoded ? decoded + i * L : nullptr;
    if (d) { d[0] = p.cx; d[1] = p.cy; d[2] = p.w; d[3] = p.h; d[4] = conf; }
    if (boxes) { float* bx = boxes + i * 4; bx[0] = p.x0; bx[1] = p.y0; bx[2] = p.x1; bx[3] = p.y1; }
    float best = 1.f;      // class probability defaults to 1, class 0 (yolov3_post_process.py:53-55)
    int arg = 0;
    const int C = L - 5;
    if (C > 0) {
      float mx = -INFINITY;
      for (int k = 0; k < C; ++k) mx = fmaxf(mx, t[5 + k]);
      float se = 0.f;
      for (int k = 0; k < C; ++k) se += expf(t[5 + k] - mx);
      best = -1.f;
      for (int k = 0; k < C; ++k) {
        const float pr = clipf_(expf(t[5 + k] - mx) / se, eps_lo, eps_hi);   // yolov3_decoder.py:189-191
        if (d) d[5 + k] = pr;
        if (pr > best) { best = pr; arg = k; }                               // np.argmax: first maximum
      }
    }
    if (score) score[i] = C > 0 ? best * conf : conf;                          // yolov3_post_process.py:57-59
    if (cls_idx) cls_idx[i] = arg;
  }
}

inline int loss_nbx(const yolo_loss_config* c) {
  int nbx = 1;
  for (int h = 0; h < 3; ++h) {
    const int cells_per_block = (LM_THREADS / 64) * (64 / c->B[h]);
    const int nb = (c->H[h] * c->W[h] + cells_per_block - 1) / cells_per_block;
    nbx = nb > nbx ? nb : nbx;
  }
  return nbx;
}

}  // namespace

extern "C" int64_t yolo_loss_workspace_bytes(const yolo_loss_config* c, int N) {
  if (!c || N <= 0 || c->T <= 0) return YOLO_ERR_INVALID_ARG;
  for (int h = 0; h < 3; ++h)
    if (c->B[h] <= 0 || c->B[h] > YOLO_MAX_ANCHORS || c->H[h] <= 0 || c->W[h] <= 0) return YOLO_ERR_INVALID_ARG;
  const int nbx = loss_nbx(c);
  return (int64_t)N * c->T * 3 * 4 /*assign*/ + (int64_t)N * 3 * nbx * 6 * 4 /*partials*/;
}

extern "C" int yolo_loss_fwd_bwd(const yolo_loss_config* c, int N, int batch_global, const float* logits8, const float* logits16,
                                 const float* logits32, const float* labels, float* dlogits8, float* dlogits16, float* dlogits32,
                                 void* dlogits8_bf16, void* dlogits16_bf16, void* dlogits32_bf16, int* current_num, float* terms,
                                 float* total, int* assign_out, float* resp_iou_out, void* workspace, void* stream) {
  YOLO_CHECK_ARG(c && N > 0 && batch_global >= N, "bad config / batch");
  YOLO_CHECK_ARG(logits8 && logits16 && logits32 && labels && current_num && terms && total && workspace, "null pointer");
  YOLO_CHECK_ARG(c->T > 0 && c->T <= 512 && c->L >= 5 && c->L <= 4096, "bad T / L");
  LossCfg cfg;
  cfg.c = *c;
  if (cfg.c.grad_scale16 == 0.f) cfg.c.grad_scale16 = 1.f;
  int max_ldc = 0;
  for (int h = 0; h < 3; ++h) {
    YOLO_CHECK_ARG(c->H[h] > 0 && c->W[h] > 0 && c->B[h] > 0 && c->B[h] <= YOLO_MAX_ANCHORS, "bad head geometry");
    YOLO_CHECK_ARG(c->ldc[h] >= c->B[h] * c->L, "ldc smaller than B*L");
    YOLO_CHECK_ARG(c->ldc[h] >= 64 && (c->ldc[h] & (c->ldc[h] - 1)) == 0, "ldc must be a power of two >= 64");
    cfg.P[h] = c->H[h] * c->W[h] * c->B[h];
    cfg.area[h] = (float)(c->H[h] * c->W[h]);
    max_ldc = max_ldc > c->ldc[h] ? max_ldc : c->ldc[h];
  }
  const int nbx = loss_nbx(c);
  for (int h = 0; h < 3; ++h) {
    const int cells_per_block = (LM_THREADS / 64) * (64 / c->B[h]);
    cfg.nb[h] = (c->H[h] * c->W[h] + cells_per_block - 1) / cells_per_block;
  }
  int* assign = assign_out ? assign_out : reinterpret_cast<int*>(workspace);
  float* partial = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + (size_t)N * c->T * 3 * 4);
  hipStream_t st = (hipStream_t)stream;
  const float inv_n = 1.f / (float)N;
  hipLaunchKernelGGL(loss_assign_kernel, dim3(N), dim3(64), (size_t)c->T * 3 * 2 * 4, st, cfg, logits8, logits16, logits32, labels, assign,
                     resp_iou_out);
  YOLO_LAUNCH_CHECK();
  const size_t lds = (size_t)lm_lds_floats(c->T, max_ldc) * 4;
  YOLO_CHECK_ARG(lds <= 64 * 1024, "T / ldc too large for the loss kernel's LDS budget");
  hipLaunchKernelGGL(loss_main_kernel, dim3(cfg.nb[0] + cfg.nb[1] + cfg.nb[2], N), dim3(LM_THREADS), lds, st, cfg, logits8, logits16, logits32,
                     labels, assign, current_num, dlogits8, dlogits16, dlogits32, (bf16_t*)dlogits8_bf16, (bf16_t*)dlogits16_bf16,
                     (bf16_t*)dlogits32_bf16, partial, nbx, inv_n);
  YOLO_LAUNCH_CHECK();
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, st, cfg, partial, N, nbx, inv_n, batch_global, current_num, terms, total);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}

extern "C" int yolo_decode_head(const float* logits, int N, int H, int W, int B, int L, int ldc, const float* anchors_grid, float eps,
                                float* decoded, float* boxes, float* score, int* cls_idx, void* stream) {
  YOLO_CHECK_ARG(logits && anchors_grid && N > 0 && H > 0 && W > 0 && B > 0 && L >= 5 && ldc >= B * L, "bad argument");
  YOLO_CHECK_ARG(decoded || boxes || score || cls_idx, "no output requested");
  const size_t total = (size_t)N * H * W * B;
  size_t blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(decode_head_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, N, H, W, B, L, ldc, anchors_grid, eps,
                     decoded, boxes, score, cls_idx);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}
