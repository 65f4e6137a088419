// Input pipeline on the GPU: decoded uint8 RGB images of arbitrary size -> the network's input batch in one pass:
// letterbox with nearest-neighbour resize, x * 1/255, RGB -> BGR (/root/reference/dataset/file_util.py:54-59) and the augmentation menu
// (/root/reference/dataset/dataset_util.py:29-99: salt-and-pepper / gaussian noise, then brightness, saturation, contrast in one of three
// orders, clip), written as float32 NHWC3 (the tensor the reference feeds to keras) and / or directly as the bf16 NHWC8 conv input.
// The per-pixel random numbers come from a counter-based Philox4x32-10 block (counter = pixel, image; key = seed), so the contrast
// pass can RECOMPUTE the noisy pixel instead of materialising an intermediate image: pass 1 reduces the per-channel means that
// adjust_contrast needs, pass 2 recomputes, finishes the chain and writes.  Arithmetic is float32 operation by operation (no fma
// contraction), mirroring oracle/dataset.py.
#include "common.h"
#include <math.h>

namespace {

constexpr int DS_THREADS = 256;
constexpr int DS_SUM_BLOCKS = 64;   // partial sums per image

struct float3_ { float r, g, b; };   // channel order as stored (BGR after the reversal)

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1, uint32_t out[4]) {
  uint32_t c[4] = {c0, c1, 0u, 0u};
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

__device__ __forceinline__ float3_ adjust_saturation(float3_ p, float factor) {   // tf.image.adjust_saturation, per pixel
  const float r = p.r, g = p.g, b = p.b;
  const float v = fmaxf(fmaxf(r, g), b);
  const float range = v - fminf(fminf(r, g), b);
  float s = v > 0.f ? range / v : 0.f;
  const float norm = 1.0f / (6.0f * range);
  float hh;
  if (r == v) hh = norm * (g - b);
  else if (g == v) hh = norm * (b - r) + (float)(2.0 / 6.0);
  else hh = norm * (r - g) + (float)(4.0 / 6.0);
  if (!(range > 0.f)) hh = 0.f;
  if (hh < 0.f) hh = hh + 1.f;
  s = fminf(fmaxf(s * factor, 0.f), 1.f);
  const float c = s * v, m = v - c, dh = hh * 6.f;
  const float fm = dh - 2.f * floorf(dh / 2.f);
  const float x = c * (1.f - fabsf(fm - 1.f));
  const int cat = (int)dh;
  float rr = 0.f, gg = 0.f, bb = 0.f;
  switch (cat) {
    case 0: rr = c; gg = x; break;
    case 1: rr = x; gg = c; break;
    case 2: gg = c; bb = x; break;
    case 3: gg = x; bb = c; break;
    case 4: rr = x; bb = c; break;
    case 5: rr = c; bb = x; break;
    default: break;
  }
  return {rr + m, gg + m, bb + m};
}

// pixel (y, x) of image n up to (not including) adjust_contrast
__device__ __forceinline__ float3_ pre_contrast(const uint8_t* __restrict__ src, const yolo_image_desc& d, int n, int y, int x, int W, int augment) {
  float3_ p = {0.f, 0.f, 0.f};
  const int yy = y - d.top, xx = x - d.left;
  if (yy >= 0 && yy < d.nh && xx >= 0 && xx < d.nw) {
    const float hs = (float)d.h / (float)d.nh, ws = (float)d.w / (float)d.nw;      // ResizeNearestNeighbor: float32 scale, floor, clamp
    int sy = (int)floorf((float)yy * hs), sx = (int)floorf((float)xx * ws);
    sy = sy < d.h - 1 ? sy : d.h - 1;
    sx = sx < d.w - 1 ? sx : d.w - 1;
    const uint8_t* q = src + d.offset + ((size_t)sy * d.w + sx) * 3;
    const float k = (float)(1.0 / 255);                                            // convert_image_dtype: multiply by float32(1/255)
    p.r = (float)q[2] * k; p.g = (float)q[1] * k; p.b = (float)q[0] * k;           // RGB -> BGR
  }
  if (!augment) return p;
  if (d.noise == 0 || d.noise == 1) {
    uint32_t r[4];
    philox4x32_10((uint32_t)(y * W + x), (uint32_t)n, d.seed0, d.seed1, r);
    if (d.noise == 0) {                                                            // salt and pepper: whole pixel -> 0 or 1
      const float sel = ((float)(r[0] >> 8) * 0x1p-24f) < 0.01f ? 1.f : 0.f;
      const float val = ((float)(r[1] >> 8) * 0x1p-24f) < 0.5f ? 1.f : 0.f;
      p.r = p.r * (1.f - sel) + val * sel; p.g = p.g * (1.f - sel) + val * sel; p.b = p.b * (1.f - sel) + val * sel;
    } else {                                                                       // gaussian, sigma 0.01 per element (Box-Muller)
      float u[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) u[i] = ((float)(r[i] >> 8) + 1.0f) * 0x1p-24f;
      const float rad0 = sqrtf(-2.0f * logf(u[0])), rad1 = sqrtf(-2.0f * logf(u[2]));
      const float two_pi = 6.283185307179586f;
      p.r = p.r + rad0 * cosf(two_pi * u[1]) * 0.01f;
      p.g = p.g + rad0 * sinf(two_pi * u[1]) * 0.01f;
      p.b = p.b + rad1 * cosf(two_pi * u[3]) * 0.01f;
    }
  }
  if (d.color_order == 0) {
    p.r += d.brightness_delta; p.g += d.brightness_delta; p.b += d.brightness_delta;
    p = adjust_saturation(p, d.saturation_factor);
  } else if (d.color_order == 1) {
    p = adjust_saturation(p, d.saturation_factor);
    p.r += d.brightness_delta; p.g += d.brightness_delta; p.b += d.brightness_delta;
  } else if (d.color_order == 2) {
    p = adjust_saturation(p, d.saturation_factor);
  }
  return p;
}

__global__ __launch_bounds__(DS_THREADS) void image_channel_sum_kernel(const uint8_t* __restrict__ src, const yolo_image_desc* __restrict__ desc, int H, int W,
                                                                       double* __restrict__ partial /*[N][DS_SUM_BLOCKS][3]*/) {
  __shared__ double red[DS_THREADS / 64][3];
  const int n = blockIdx.y;
  const yolo_image_desc d = desc[n];
  if (d.color_order < 0 || d.color_order > 2) return;          // no contrast step: no mean needed
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  const int P = H * W;
  for (int i = blockIdx.x * DS_THREADS + threadIdx.x; i < P; i += DS_SUM_BLOCKS * DS_THREADS) {
    const float3_ p = pre_contrast(src, d, n, i / W, i % W, W, 1);
    s0 += p.r; s1 += p.g; s2 += p.b;
  }
  double a = s0, b = s1, c = s2;
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); c += __shfl_xor(c, o, 64); }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a; red[threadIdx.x >> 6][1] = b; red[threadIdx.x >> 6][2] = c; }
  __syncthreads();
  if (threadIdx.x < 3) {
    double t = 0.0;
    for (int k = 0; k < DS_THREADS / 64; ++k) t += red[k][threadIdx.x];
    partial[((size_t)n * DS_SUM_BLOCKS + blockIdx.x) * 3 + threadIdx.x] = t;
  }
}

__global__ __launch_bounds__(DS_THREADS) void letterbox_augment_kernel(const uint8_t* __restrict__ src, const yolo_image_desc* __restrict__ desc, int H, int W,
                                                                       int augment, const double* __restrict__ partial, float* __restrict__ out_f32,
                                                                       bf16_t* __restrict__ out_bf16) {
  __shared__ float mean_s[3];
  const int n = blockIdx.y;
  const yolo_image_desc d = desc[n];
  const bool contrast = augment && d.color_order >= 0 && d.color_order <= 2;
  if (contrast && threadIdx.x < 3) {
    double t = 0.0;
    for (int k = 0; k < DS_SUM_BLOCKS; ++k) t += partial[((size_t)n * DS_SUM_BLOCKS + k) * 3 + threadIdx.x];   // fixed order: deterministic
    mean_s[threadIdx.x] = (float)(t / (double)((size_t)H * W));
  }
  __syncthreads();
  const int P = H * W;
  for (int i = blockIdx.x * DS_THREADS + threadIdx.x; i < P; i += gridDim.x * DS_THREADS) {
    float3_ p = pre_contrast(src, d, n, i / W, i % W, W, augment);
    if (contrast) {                                            // adjust_contrast: (x - mean) * factor + mean, per channel
      p.r = (p.r - mean_s[0]) * d.contrast_factor + mean_s[0];
      p.g = (p.g - mean_s[1]) * d.contrast_factor + mean_s[1];
      p.b = (p.b - mean_s[2]) * d.contrast_factor + mean_s[2];
      if (d.color_order == 2) { p.r += d.brightness_delta; p.g += d.brightness_delta; p.b += d.brightness_delta; }
    }
    if (augment) {                                             // clip_by_value(image, 0, 1) (dataset_util.py:98)
      p.r = fminf(fmaxf(p.r, 0.f), 1.f); p.g = fminf(fmaxf(p.g, 0.f), 1.f); p.b = fminf(fmaxf(p.b, 0.f), 1.f);
    }
    const size_t o = (size_t)n * P + i;
    if (out_f32) { out_f32[o * 3 + 0] = p.r; out_f32[o * 3 + 1] = p.g; out_f32[o * 3 + 2] = p.b; }
    if (out_bf16) {
      const float v[8] = {p.r, p.g, p.b, 0.f, 0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<uint4*>(out_bf16 + o * 8) = pack_bf8(v);
    }
  }
}

}  // namespace

extern "C" int64_t yolo_letterbox_workspace_bytes(int N) { return N > 0 ? (int64_t)N * DS_SUM_BLOCKS * 3 * (int64_t)sizeof(double) : YOLO_ERR_INVALID_ARG; }

extern "C" int yolo_letterbox_augment(const uint8_t* src, const yolo_image_desc* desc, int N, int H, int W, int augment, void* workspace,
                                      float* out_f32, void* out_bf16x8, void* stream) {
  YOLO_CHECK_ARG(src && desc && (out_f32 || out_bf16x8), "yolo_letterbox_augment: null pointer");
  YOLO_CHECK_ARG(N > 0 && H > 0 && W > 0 && (long long)H * W < (1ll << 30), "yolo_letterbox_augment: bad shape");
  YOLO_CHECK_ARG(!augment || workspace, "yolo_letterbox_augment: augmentation needs the workspace of yolo_letterbox_workspace_bytes");
  if (augment) {
    hipLaunchKernelGGL(image_channel_sum_kernel, dim3(DS_SUM_BLOCKS, N), dim3(DS_THREADS), 0, (hipStream_t)stream, src, desc, H, W, (double*)workspace);
    YOLO_LAUNCH_CHECK();
  }
  int bx = (H * W + DS_THREADS * 4 - 1) / (DS_THREADS * 4);
  bx = bx < 1 ? 1 : bx > 256 ? 256 : bx;
  hipLaunchKernelGGL(letterbox_augment_kernel, dim3(bx, N), dim3(DS_THREADS), 0, (hipStream_t)stream, src, desc, H, W, augment, (const double*)workspace,
                     out_f32, (bf16_t*)out_bf16x8);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}
