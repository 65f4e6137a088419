// Stem convolution of the three backbones: Conv2D 3x3 stride 2 'same' on the RGB image, 3 -> 64 channels (reference resnet18.py:59,
// resnet18_v2.py:61, mixnet18.py:72; TF 'same' at stride 2 on an even size pads bottom / right only).  M = N*Ho*Wo = 1.38 M pixels, K = 27:
// 4.8 GFLOP but 88 MB in + 177 MB out at 416^2 / batch 32 -- a pure HBM stream.  The implicit-GEMM kernel spent 132 us on it (one
// 128-pixel tile per workgroup: prologue, two K-steps behind barriers, three-sync epilogue, 10816 workgroups); this kernel walks image rows:
//   * a workgroup owns `rows` consecutive output rows of one image and all 64 output channels; per output row it needs input rows
//     2ho, 2ho+1, 2ho+2, kept in a 3-slot LDS ring (slot = row % 3): two new rows per iteration, requested into registers one
//     iteration ahead, the zero pad column (x = W) and the zero bottom row (y = H) live in LDS
//   * K is laid out as tap * 4 + channel (the packed input pixel is [c0 c1 c2 0 0 0 0 0]: one 8-byte LDS read per tap): taps 0..7 fill one
//     v_mfma_f32_16x16x32 K-step, tap 8 a quarter of a second one; wave w owns output channels 16w..16w+15 and keeps its weights in 8 VGPRs
//   * the output row is staged in LDS and written as whole 128-byte NHWC pixels; BatchNorm partial statistics (of the values as stored)
//     accumulate in registers over all rows of the workgroup: ONE statistics row per workgroup (832 instead of 10816 rows for bn_finalize)
#include "common.h"

namespace {

constexpr int ST_THREADS = 256;

struct StemArgs {
  const bf16_t* x;      // [N][H][W][8]
  const bf16_t* w;      // [64][3][3][8]
  bf16_t* y;            // [N][Ho][Wo][64]
  float* stat_sum; float* stat_sq;   // [grid][64] or null
  int H, W, Ho, Wo, rows, groups;    // groups = row groups per image
};

__device__ __forceinline__ uint2 lds_read8(const char* p) { return *reinterpret_cast<const uint2*>(p); }

__global__ __launch_bounds__(ST_THREADS) void stem_conv3x3s2_kernel(StemArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int px = lane & 15, q = lane >> 4;
  const int n = blockIdx.x / a.groups, grp = blockIdx.x - n * a.groups;
  const int ho0 = grp * a.rows, ho1 = min(ho0 + a.rows, a.Ho);
  const int W = a.W, rowb = (W + 1) * 16;                 // bytes of one LDS input row (with the pad column)
  char* ring = smem;                                      // 3 rows
  char* outb = smem + 3 * rowb;                           // Wo pixels * OSTR bytes (16-byte pad per pixel: conflict-free 8-byte writes)
  constexpr int OSTR = 144;
  const bf16_t* xin = a.x + (size_t)n * a.H * W * 8;

  // weights of this wave's 16 channels: K-step 0 = taps 2q, 2q+1 (4 channels each), K-step 1 = tap 8 for q == 0
  const int co = wave * 16 + px;
  union { bf16x8_t v; uint2 u[2]; } wa0, wa1;
  wa0.u[0] = *reinterpret_cast<const uint2*>(a.w + ((size_t)co * 9 + 2 * q) * 8);
  wa0.u[1] = *reinterpret_cast<const uint2*>(a.w + ((size_t)co * 9 + 2 * q + 1) * 8);
  wa1.u[0] = q == 0 ? *reinterpret_cast<const uint2*>(a.w + ((size_t)co * 9 + 8) * 8) : make_uint2(0u, 0u);
  wa1.u[1] = make_uint2(0u, 0u);
  // LDS offsets of this lane's taps relative to (row slot base, pixel 2*wo): tap t = (r, s) -> r selects the slot, s * 16 bytes
  const int t0 = 2 * q, t1 = 2 * q + 1;
  const int r0 = t0 / 3, s0 = t0 - 3 * r0, r1 = t1 / 3, s1 = t1 - 3 * r1;

  // zero the pad column of the three slots; load the first input row
  if (tid < 3) *reinterpret_cast<uint4*>(ring + tid * rowb + W * 16) = make_uint4(0u, 0u, 0u, 0u);
  const int chunks = W;                                   // 16-byte chunks per input row
  for (int c = tid; c < chunks; c += ST_THREADS)
    *reinterpret_cast<uint4*>(ring + ((2 * ho0) % 3) * rowb + c * 16) = *reinterpret_cast<const uint4*>(xin + ((size_t)(2 * ho0) * W + c) * 8);

  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  constexpr int MAXPF = 6;                                // prefetch registers: 2 rows * W chunks / 256 threads <= 6 (W <= 768)
  // the two new input rows of an iteration are requested ONE ITERATION AHEAD into registers (they fly under the MFMA and store phases of
  // the row before) and written to the ring at the top of their iteration, when nobody reads the slots they replace any more.
  // (Requested and awaited at the top of the same iteration, as this kernel did until round 3, every row exposed an HBM round trip:
  // 71 -> 63 us alone at 416^2 / batch 32.)
  uint4 pf[MAXPF];
  auto request_rows = [&](int ho) {                       // input rows 2ho+1, 2ho+2 (row H is the zero pad)
    const int rA = 2 * ho + 1;
#pragma unroll
    for (int k = 0; k < MAXPF; ++k) {
      const int c = tid + k * ST_THREADS;
      const int rr = rA + (c >= chunks ? 1 : 0), cc = c >= chunks ? c - chunks : c;
      pf[k] = (c < 2 * chunks && rr < a.H) ? *reinterpret_cast<const uint4*>(xin + ((size_t)rr * W + cc) * 8) : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  request_rows(ho0);
  for (int ho = ho0; ho < ho1; ++ho) {
    const int rA = 2 * ho + 1;
#pragma unroll
    for (int k = 0; k < MAXPF; ++k) {
      const int c = tid + k * ST_THREADS;
      if (c < 2 * chunks) {
        const int rr = rA + (c >= chunks ? 1 : 0), cc = c >= chunks ? c - chunks : c;
        *reinterpret_cast<uint4*>(ring + (rr % 3) * rowb + cc * 16) = pf[k];
      }
    }
    __syncthreads();                                      // rows 2ho .. 2ho+2 are in the ring
    if (ho + 1 < ho1) request_rows(ho + 1);
    const char* b0 = ring + ((2 * ho + r0) % 3) * rowb + s0 * 16;
    const char* b1 = ring + ((2 * ho + r1) % 3) * rowb + s1 * 16;
    const char* b8 = ring + ((2 * ho + 2) % 3) * rowb + 2 * 16;
    for (int t = 0; t < a.Wo / 16; ++t) {
      const int wo = t * 16 + px;
      union { bf16x8_t v; uint2 u[2]; } xb0, xb1;
      xb0.u[0] = lds_read8(b0 + wo * 32);
      xb0.u[1] = lds_read8(b1 + wo * 32);
      xb1.u[0] = q == 0 ? lds_read8(b8 + wo * 32) : make_uint2(0u, 0u);
      xb1.u[1] = make_uint2(0u, 0u);
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
      acc = YOLO_MFMA_16x16x32(wa0.v, xb0.v, acc);
      acc = YOLO_MFMA_16x16x32(wa1.v, xb1.v, acc);
      uint2 o;
      o.x = pack_bf2(acc[0], acc[1]);
      o.y = pack_bf2(acc[2], acc[3]);
      const float v0 = lo2f(o.x), v1 = hi2f(o.x), v2 = lo2f(o.y), v3 = hi2f(o.y);   // statistics of the values as stored
      ssum[0] += v0; ssum[1] += v1; ssum[2] += v2; ssum[3] += v3;
      ssq[0] += v0 * v0; ssq[1] += v1 * v1; ssq[2] += v2 * v2; ssq[3] += v3 * v3;
      *reinterpret_cast<uint2*>(outb + wo * OSTR + (wave * 16 + q * 4) * 2) = o;
    }
    __syncthreads();                                      // the output row is staged; everybody is done reading rows 2ho, 2ho+1
    bf16_t* yrow = a.y + ((size_t)(n * a.Ho + ho) * a.Wo) * 64;
    for (int c = tid; c < a.Wo * 8; c += ST_THREADS)
      *reinterpret_cast<uint4*>(yrow + (size_t)c * 8) = *reinterpret_cast<const uint4*>(outb + (c >> 3) * OSTR + (c & 7) * 16);
    // (the next iteration's ring writes target the slots of rows 2ho and 2ho+1 only; its first barrier orders them against the staged
    //  row's global stores above, which read outb -- a different region)
  }
  if (a.stat_sum) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float s = row16_sum(ssum[j]), sq = row16_sum(ssq[j]);
      if (px == 0) {
        a.stat_sum[(size_t)blockIdx.x * 64 + wave * 16 + q * 4 + j] = s;
        a.stat_sq[(size_t)blockIdx.x * 64 + wave * 16 + q * 4 + j] = sq;
      }
    }
  }
}

int g_stem_direct = 1;

inline int stem_rows(int N, int Ho) {      // output rows per workgroup: aim at ~512 workgroups (2 resident per CU), all in one round
  int rows = (N * Ho + 511) / 512;
  if (rows < 1) rows = 1;
  if (rows > Ho) rows = Ho;
  return rows;
}

}  // namespace

bool yolo_stem_applies(const yolo_conv_problem* p) {
  return g_stem_direct && p->Cin == 8 && p->C0 == 0 && p->Cout == 64 && p->R == 3 && p->S == 3 && p->stride == 2 && p->pad_t == 0 && p->pad_l == 0 &&
         p->H % 2 == 0 && p->W % 2 == 0 && p->Ho == p->H / 2 && p->Wo == p->W / 2 && p->Wo % 16 == 0 && p->W <= 768;
}

int yolo_stem_set_direct(int on) { g_stem_direct = on ? 1 : 0; return YOLO_OK; }

int yolo_stem_stat_rows(const yolo_conv_problem* p) {
  const int rows = stem_rows(p->N, p->Ho);
  return p->N * ((p->Ho + rows - 1) / rows);
}

int yolo_stem_fwd(const yolo_conv_problem* p, const void* x, const void* w, void* y, float* stat_sum, float* stat_sq, void* stream) {
  StemArgs a;
  a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.y = (bf16_t*)y; a.stat_sum = stat_sum; a.stat_sq = stat_sq;
  a.H = p->H; a.W = p->W; a.Ho = p->Ho; a.Wo = p->Wo;
  a.rows = stem_rows(p->N, p->Ho);
  a.groups = (p->Ho + a.rows - 1) / a.rows;
  const size_t lds = 3 * (size_t)(p->W + 1) * 16 + (size_t)p->Wo * 144;
  static size_t lds_allowed = 64 * 1024;
  if (lds > lds_allowed) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_conv3x3s2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { yolo_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
    lds_allowed = lds;
  }
  hipLaunchKernelGGL(stem_conv3x3s2_kernel, dim3(p->N * a.groups), dim3(ST_THREADS), lds, (hipStream_t)stream, a);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}
