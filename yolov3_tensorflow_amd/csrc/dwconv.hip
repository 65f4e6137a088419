// Mixed depthwise convolution ("MixConv") of the reference's MixNet-18 block, gfx950, NHWC bf16, bandwidth-bound.
//
// Replaces, per block, the Lambda channel slices + 4 x keras DepthwiseConv2D (k = 3/5/7/9, stride 1, 'same', no bias) +
// Concatenate of backbone/mixnet18.py:38-45 (factories backbone/basic_backbone.py:45-66) and their TF gradients: the slice and
// the concat are pure addressing (channel group -> kernel size), so one launch covers the whole tensor.
//   fwd / dgrad : one lane = one pixel x 8 channels (16-byte vectors); blockIdx.y = channel group, so a wave runs one kernel size.
//                 dgrad is the same kernel with flipped taps (stride 1, symmetric padding).
//   wgrad       : dW[tap][c] = sum over pixels of x(shifted) * dy; per workgroup a pixel range, per tap a block reduction through
//                 LDS and one float atomic per (tap, channel).
#include "common.h"

namespace {

struct MixP {
  int N, H, W, C;
  int split[5];
  int ksize[4];
};

constexpr int DW_THREADS = 256;

__device__ __forceinline__ uint4 ld16(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }

__global__ __launch_bounds__(DW_THREADS) void dwconv_mix_kernel(MixP p, const bf16_t* __restrict__ x, const bf16_t* __restrict__ w0,
                                                                const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2,
                                                                const bf16_t* __restrict__ w3, bf16_t* __restrict__ y, int flip, int accumulate) {
  const int grp = blockIdx.y;
  const int c0 = p.split[grp], cg = p.split[grp + 1] - c0, cv = cg >> 3;
  if (cv == 0) return;
  const int k = p.ksize[grp], pad = k >> 1;
  const bf16_t* w = grp == 0 ? w0 : (grp == 1 ? w1 : (grp == 2 ? w2 : w3));
  const size_t total = (size_t)p.N * p.H * p.W * cv;
  for (size_t i = (size_t)blockIdx.x * DW_THREADS + threadIdx.x; i < total; i += (size_t)gridDim.x * DW_THREADS) {
    const int chunk = (int)(i % cv);
    size_t pix = i / cv;
    const int wq = (int)(pix % p.W);
    size_t t = pix / p.W;
    const int hq = (int)(t % p.H);
    const int n = (int)(t / p.H);
    const int c = c0 + chunk * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int dh = 0; dh < k; ++dh) {
      const int hh = hq + dh - pad;
      if (hh < 0 || hh >= p.H) continue;
      for (int dw = 0; dw < k; ++dw) {
        const int ww = wq + dw - pad;
        if (ww < 0 || ww >= p.W) continue;
        float xv[8], wv[8];
        unpack_bf8(ld16(x + ((size_t)(n * p.H + hh) * p.W + ww) * p.C + c), xv);
        const int tap = flip ? ((k - 1 - dh) * k + (k - 1 - dw)) : (dh * k + dw);
        unpack_bf8(ld16(w + (size_t)tap * cg + chunk * 8), wv);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += xv[j] * wv[j];
      }
    }
    bf16_t* yo = y + pix * p.C + c;
    if (accumulate) {
      float o[8];
      unpack_bf8(ld16(yo), o);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += o[j];
    }
    *reinterpret_cast<uint4*>(yo) = pack_bf8(acc);
  }
}

__global__ __launch_bounds__(DW_THREADS) void dwconv_mix_wgrad_kernel(MixP p, const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                      float* __restrict__ d0, float* __restrict__ d1, float* __restrict__ d2,
                                                                      float* __restrict__ d3, int pix_per_block) {
  __shared__ float red[DW_THREADS * 8];
  const int grp = blockIdx.y;
  const int c0 = p.split[grp], cg = p.split[grp + 1] - c0, cv = cg >> 3;
  if (cv == 0) return;
  const int k = p.ksize[grp], pad = k >> 1;
  float* dw_out = grp == 0 ? d0 : (grp == 1 ? d1 : (grp == 2 ? d2 : d3));
  const int npl = DW_THREADS / cv;                // pixel lanes (cv is a power of two <= 64 here; surplus threads idle)
  const int chunk = threadIdx.x % cv, pl = threadIdx.x / cv;
  const bool active = pl < npl;
  const int M = p.N * p.H * p.W;
  const int pbeg = blockIdx.x * pix_per_block, pend = min(M, pbeg + pix_per_block);
  const int c = c0 + chunk * 8;
  for (int tap = 0; tap < k * k; ++tap) {
    const int dh = tap / k - pad, dwc = tap % k - pad;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (active) {
      for (int pix = pbeg + pl; pix < pend; pix += npl) {
        const int wq = pix % p.W;
        const int t = pix / p.W;
        const int hq = t % p.H, n = t / p.H;
        const int hh = hq + dh, ww = wq + dwc;
        if (hh < 0 || hh >= p.H || ww < 0 || ww >= p.W) continue;
        float xv[8], gv[8];
        unpack_bf8(ld16(x + ((size_t)(n * p.H + hh) * p.W + ww) * p.C + c), xv);
        unpack_bf8(ld16(dy + (size_t)pix * p.C + c), gv);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += xv[j] * gv[j];
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = acc[j];
    __syncthreads();
    // thread u < cg sums channel u over the pixel lanes:  red[(pl * cv + u / 8) * 8 + u % 8]
    for (int u = threadIdx.x; u < cg; u += DW_THREADS) {
      float s = 0.f;
      for (int q = 0; q < npl; ++q) s += red[(q * cv + (u >> 3)) * 8 + (u & 7)];
      atomicAdd(dw_out + (size_t)tap * cg + u, s);
    }
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
    YOLO_CHECK_ARG(p->ksize[g] >= 1 && p->ksize[g] <= 9 && (p->ksize[g] & 1), "kernel sizes must be odd and <= 9");
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
  const size_t items = (size_t)p->N * p->H * p->W * (p->C / 16);
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
  const int M = p->N * p->H * p->W;
  int blocks = (M + 2047) / 2048;
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  const int ppb = (M + blocks - 1) / blocks;
  hipLaunchKernelGGL(dwconv_mix_wgrad_kernel, dim3(blocks, 4), dim3(DW_THREADS), 0, (hipStream_t)stream, to_dev(p), (const bf16_t*)x,
                     (const bf16_t*)dy, dw0, dw1, dw2, dw3, ppb);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}
