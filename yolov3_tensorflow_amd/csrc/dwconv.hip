// Mixed depthwise convolution ("MixConv") of the reference's MixNet-18 block, gfx950, NHWC bf16, bandwidth-bound.
//
// Replaces, per block, the Lambda channel slices + 4 x keras DepthwiseConv2D (k = 3/5/7/9, stride 1, 'same', no bias) +
// Concatenate of backbone/mixnet18.py:38-45 (factories backbone/basic_backbone.py:45-66) and their TF gradients: the slice and
// the concat are pure addressing (channel group -> kernel size), so one launch covers the whole tensor.
//   fwd / dgrad : one lane = a strip of 8 output pixels x 8 channels (16-byte vectors); blockIdx.y = channel group, so a wave runs one
//                 kernel size; input row segments and weight rows are loaded once per strip and the window slides in registers.
//                 dgrad is the same kernel with flipped taps (stride 1, symmetric padding).
//   wgrad       : dW[tap][c] = sum over pixels of x(shifted) * dy; per workgroup a range of image rows, per kernel row the K taps are
//                 accumulated in registers over the strips, then a block reduction through LDS and one float atomic per (tap, channel).
#include "common.h"

namespace {

struct MixP {
  int N, H, W, C;
  int split[5];
  int ksize[4];
};

constexpr int DW_THREADS = 256;

__device__ __forceinline__ uint4 ld16(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }

constexpr int SP = 8;   // output pixels per lane (a horizontal strip)

// fwd / dgrad body for one kernel size: a lane owns a strip of SP consecutive output pixels of one image row and 8 channels; per kernel
// row it loads the SP + K - 1 input chunks and the K weight chunks once and slides the window in registers (K*(SP+K-1)/SP loads per
// output instead of K*K).
template <int K>
__device__ __forceinline__ void dw_strip(const MixP& p, const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ y,
                                         int c0, int cg, int cv, int flip, int accumulate) {
  constexpr int PAD = K / 2;
  const int strips = (p.W + SP - 1) / SP;
  const size_t total = (size_t)p.N * p.H * strips * cv;
  for (size_t i = (size_t)blockIdx.x * DW_THREADS + threadIdx.x; i < total; i += (size_t)gridDim.x * DW_THREADS) {
    const int chunk = (int)(i % cv);
    size_t t = i / cv;
    const int sx = (int)(t % strips);
    t /= strips;
    const int hq = (int)(t % p.H);
    const int n = (int)(t / p.H);
    const int c = c0 + chunk * 8, w0 = sx * SP;
    float acc[SP][8];
#pragma unroll
    for (int o = 0; o < SP; ++o)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[o][j] = 0.f;
    for (int dh = 0; dh < K; ++dh) {
      const int hh = hq + dh - PAD;
      if (hh < 0 || hh >= p.H) continue;
      float wt[K][8];
      const int wr = flip ? (K - 1 - dh) : dh;
#pragma unroll
      for (int dw = 0; dw < K; ++dw) unpack_bf8(ld16(w + (size_t)(wr * K + (flip ? (K - 1 - dw) : dw)) * cg + chunk * 8), wt[dw]);
      const bf16_t* xrow = x + ((size_t)(n * p.H + hh) * p.W) * p.C + c;
#pragma unroll
      for (int ic = 0; ic < SP + K - 1; ++ic) {
        const int ww = w0 - PAD + ic;
        if (ww < 0 || ww >= p.W) continue;
        float xv[8];
        unpack_bf8(ld16(xrow + (size_t)ww * p.C), xv);
#pragma unroll
        for (int dw = 0; dw < K; ++dw) {
          const int o = ic - dw;             // output pixel this (input column, tap) pair feeds
          if (o >= 0 && o < SP) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[o][j] += xv[j] * wt[dw][j];
          }
        }
      }
    }
    bf16_t* yrow = y + ((size_t)(n * p.H + hq) * p.W) * p.C + c;
#pragma unroll
    for (int o = 0; o < SP; ++o) {
      if (w0 + o < p.W) {
        bf16_t* yo = yrow + (size_t)(w0 + o) * p.C;
        if (accumulate) {
          float old[8];
          unpack_bf8(ld16(yo), old);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[o][j] += old[j];
        }
        *reinterpret_cast<uint4*>(yo) = pack_bf8(acc[o]);
      }
    }
  }
}

__global__ __launch_bounds__(DW_THREADS) void dwconv_mix_kernel(MixP p, const bf16_t* __restrict__ x, const bf16_t* __restrict__ w0,
                                                                const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2,
                                                                const bf16_t* __restrict__ w3, bf16_t* __restrict__ y, int flip, int accumulate) {
  const int grp = blockIdx.y;
  const int c0 = p.split[grp], cg = p.split[grp + 1] - c0, cv = cg >> 3;
  if (cv == 0) return;
  const bf16_t* w = grp == 0 ? w0 : (grp == 1 ? w1 : (grp == 2 ? w2 : w3));
  switch (p.ksize[grp]) {   // blockIdx.y-uniform
    case 1: dw_strip<1>(p, x, w, y, c0, cg, cv, flip, accumulate); break;
    case 3: dw_strip<3>(p, x, w, y, c0, cg, cv, flip, accumulate); break;
    case 5: dw_strip<5>(p, x, w, y, c0, cg, cv, flip, accumulate); break;
    case 7: dw_strip<7>(p, x, w, y, c0, cg, cv, flip, accumulate); break;
    default: dw_strip<9>(p, x, w, y, c0, cg, cv, flip, accumulate); break;
  }
}

// wgrad body for one kernel size: per kernel row dh a lane accumulates the K taps of that row for 8 channels over its strips
// (x row segment and dy strip loaded once per (strip, dh)), then the block reduces over lanes through LDS and issues one float atomic
// per (tap, channel).
template <int K>
__device__ __forceinline__ void dw_wgrad_strip(const MixP& p, const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                               float* __restrict__ dw_out, int c0, int cg, int cv, int rows_per_block, float* red) {
  constexpr int PAD = K / 2;
  const int npl = DW_THREADS / cv;
  const int chunk = threadIdx.x % cv, pl = threadIdx.x / cv;
  const bool active = pl < npl;
  const int strips = (p.W + SP - 1) / SP;
  const int nrows = p.N * p.H;
  const int rbeg = blockIdx.x * rows_per_block, rend = min(nrows, rbeg + rows_per_block);
  const int c = c0 + chunk * 8;
  for (int dh = 0; dh < K; ++dh) {
    float acc[K][8];
#pragma unroll
    for (int d = 0; d < K; ++d)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[d][j] = 0.f;
    if (active) {
      for (int it = (rbeg * strips) + pl; it < rend * strips; it += npl) {
        const int row = it / strips, sx = it - row * strips;
        const int hq = row % p.H, n = row / p.H;
        const int hh = hq + dh - PAD;
        if (hh < 0 || hh >= p.H) continue;
        const int w0 = sx * SP;
        float g[SP][8];
        const bf16_t* grow = dy + ((size_t)row * p.W) * p.C + c;
#pragma unroll
        for (int o = 0; o < SP; ++o) {
          if (w0 + o < p.W) unpack_bf8(ld16(grow + (size_t)(w0 + o) * p.C), g[o]);
          else {
#pragma unroll
            for (int j = 0; j < 8; ++j) g[o][j] = 0.f;
          }
        }
        const bf16_t* xrow = x + ((size_t)(n * p.H + hh) * p.W) * p.C + c;
#pragma unroll
        for (int ic = 0; ic < SP + K - 1; ++ic) {
          const int ww = w0 - PAD + ic;
          if (ww < 0 || ww >= p.W) continue;
          float xv[8];
          unpack_bf8(ld16(xrow + (size_t)ww * p.C), xv);
#pragma unroll
          for (int d = 0; d < K; ++d) {
            const int o = ic - d;
            if (o >= 0 && o < SP) {
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[d][j] += xv[j] * g[o][j];
            }
          }
        }
      }
    }
#pragma unroll
    for (int d = 0; d < K; ++d) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = acc[d][j];
      __syncthreads();
      for (int u = threadIdx.x; u < cg; u += DW_THREADS) {
        float s = 0.f;
        for (int q = 0; q < npl; ++q) s += red[(q * cv + (u >> 3)) * 8 + (u & 7)];
        atomicAdd(dw_out + (size_t)(dh * K + d) * cg + u, s);
      }
    }
  }
}

__global__ __launch_bounds__(DW_THREADS) void dwconv_mix_wgrad_kernel(MixP p, const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                      float* __restrict__ d0, float* __restrict__ d1, float* __restrict__ d2,
                                                                      float* __restrict__ d3, int rows_per_block) {
  __shared__ float red[DW_THREADS * 8];
  const int grp = blockIdx.y;
  const int c0 = p.split[grp], cg = p.split[grp + 1] - c0, cv = cg >> 3;
  if (cv == 0) return;
  float* dw_out = grp == 0 ? d0 : (grp == 1 ? d1 : (grp == 2 ? d2 : d3));
  switch (p.ksize[grp]) {
    case 1: dw_wgrad_strip<1>(p, x, dy, dw_out, c0, cg, cv, rows_per_block, red); break;
    case 3: dw_wgrad_strip<3>(p, x, dy, dw_out, c0, cg, cv, rows_per_block, red); break;
    case 5: dw_wgrad_strip<5>(p, x, dy, dw_out, c0, cg, cv, rows_per_block, red); break;
    case 7: dw_wgrad_strip<7>(p, x, dy, dw_out, c0, cg, cv, rows_per_block, red); break;
    default: dw_wgrad_strip<9>(p, x, dy, dw_out, c0, cg, cv, rows_per_block, red); break;
  }
}

int check_mix(const yolo_mixconv_problem* p) {
  YOLO_CHECK_ARG(p != nullptr, "null problem");
  YOLO_CHECK_ARG(p->N > 0 && p->H > 0 && p->W > 0 && p->C > 0 && p->C % 8 == 0, "bad dims");
  YOLO_CHECK_ARG(p->split[0] == 0 && p->split[4] == p->C, "split must cover [0, C)");
  for (int g = 0; g < 4; ++g) {
    const int cg = p->split[g + 1] - p->split[g];
    YOLO_CHECK_ARG(cg >= 0 && cg % 8 == 0, "group sizes must be multiples of 8");
    YOLO_CHECK_ARG(cg == 0 || ((cg / 8) <= 64 && (((cg / 8) & ((cg / 8) - 1)) == 0)), "group size / 8 must be a power of two <= 64");
    YOLO_CHECK_ARG(p->ksize[g] == 1 || p->ksize[g] == 3 || p->ksize[g] == 5 || p->ksize[g] == 7 || p->ksize[g] == 9, "kernel sizes: 1,3,5,7,9");
  }
  YOLO_CHECK_ARG((size_t)p->N * p->H * p->W < (1ull << 31), "too many pixels");
  return YOLO_OK;
}

MixP to_dev(const yolo_mixconv_problem* p) {
  MixP m;
  m.N = p->N; m.H = p->H; m.W = p->W; m.C = p->C;
  for (int i = 0; i < 5; ++i) m.split[i] = p->split[i];
  for (int i = 0; i < 4; ++i) m.ksize[i] = p->ksize[i];
  return m;
}

int launch_mix(const yolo_mixconv_problem* p, const void* x, const void* w0, const void* w1, const void* w2, const void* w3, void* y, int flip,
               int accumulate, void* stream) {
  int rc = check_mix(p);
  if (rc) return rc;
  YOLO_CHECK_ARG(x && y && w0 && w1 && w2 && w3, "null pointer");
  const size_t items = (size_t)p->N * p->H * ((p->W + SP - 1) / SP) * (p->C / 16);      // strips x chunks of the largest group
  size_t b = (items + DW_THREADS - 1) / DW_THREADS;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  hipLaunchKernelGGL(dwconv_mix_kernel, dim3((unsigned)b, 4), dim3(DW_THREADS), 0, (hipStream_t)stream, to_dev(p), (const bf16_t*)x,
                     (const bf16_t*)w0, (const bf16_t*)w1, (const bf16_t*)w2, (const bf16_t*)w3, (bf16_t*)y, flip, accumulate);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}

}  // namespace

extern "C" int yolo_dwconv_mix_fwd(const yolo_mixconv_problem* p, const void* x, const void* w0, const void* w1, const void* w2,
                                   const void* w3, void* y, void* stream) {
  return launch_mix(p, x, w0, w1, w2, w3, y, 0, 0, stream);
}

extern "C" int yolo_dwconv_mix_dgrad(const yolo_mixconv_problem* p, const void* dy, const void* w0, const void* w1, const void* w2,
                                     const void* w3, void* dx, int accumulate, void* stream) {
  return launch_mix(p, dy, w0, w1, w2, w3, dx, 1, accumulate, stream);
}

extern "C" int yolo_dwconv_mix_wgrad(const yolo_mixconv_problem* p, const void* x, const void* dy, float* dw0, float* dw1, float* dw2,
                                     float* dw3, void* stream) {
  int rc = check_mix(p);
  if (rc) return rc;
  YOLO_CHECK_ARG(x && dy && dw0 && dw1 && dw2 && dw3, "null pointer");
  const int nrows = p->N * p->H;
  int blocks = nrows < 256 ? nrows : 256;
  if (blocks < 1) blocks = 1;
  const int ppb = (nrows + blocks - 1) / blocks;      // image rows per workgroup
  hipLaunchKernelGGL(dwconv_mix_wgrad_kernel, dim3(blocks, 4), dim3(DW_THREADS), 0, (hipStream_t)stream, to_dev(p), (const bf16_t*)x,
                     (const bf16_t*)dy, dw0, dw1, dw2, dw3, ppb);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}
