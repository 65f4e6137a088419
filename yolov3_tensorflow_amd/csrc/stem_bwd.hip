// Backward pass of the stem, from the pooled gradient straight to the stem convolution's weight gradient:
//   conv3x3 s2 (RGB -> 64) -> [BatchNorm] -> max-pool 3x3 s2 'same' -> [ReLU]        (reference resnet18.py:59-61, mixnet18.py:72,
//   resnet18_v2.py:61-62; TF differentiates the chain op by op)
// The stem convolution has no data gradient (its input is the image), so the pre-pool gradient dy[N][H][W][64] is consumed by the weight
// gradient ONLY.  The two-kernel form wrote it (177 MB at 416^2 / batch 32) and read it back: 151 us (un-pool + BatchNorm apply) + 77 us
// (implicit-GEMM weight gradient, K = 27) serial at the very end of the step, with nothing left to overlap them.  Here a workgroup owns
// 8 x 16 pre-pool pixels at a time:
//   1. the <= 6 x 10 pooled pixels whose windows touch the tile are staged in LDS (ReLU-masked gradient + arg-max bytes, 24 B per
//      8-channel chunk); every pre-pool chunk then GATHERS from the 1..4 windows covering it: g[j] += d[j] where the window's arg-max
//      byte names this position -- in the window-class order (dh == 2, dw == 2) of bn_pool_bwd_apply_scatter_kernel, so the float32
//      sums are the same.  (A scatter through ds_add_f32 into a float32 LDS tile, four barrier-separated classes, cost 88 of 216 us.)
//   2. BatchNorm apply dy = a (g - k1 - xhat k2) = A g + B y + D (or dy = g without BN) in registers, rounded to the 16-bit activation
//      type as the stored dy was, into a [64 pixel rows][128 columns] LDS image (column = 64 * (tile row >> 2) + channel) laid out for ds_read_b64_tr_b16
//   3. dW[co][tap * 4 + ch] += sum over the 128 pixels dy[p][co] * x[2h + r][2w + s][ch]: v_mfma_f32_16x16x32 with the pixels as K; wave w
//      takes pixels 32w .. 32w+31 for all 4 x 3 output tiles.  A (dy^T) comes from the transposed LDS read, B from a 17 x 33 pixel patch
//      of the packed image (4 channels x 2 bytes per pixel in LDS) gathered with eight 2-byte reads per fragment
// The grid is persistent (3 workgroups per CU); every workgroup keeps its 64 x 48 float32 partial in registers over its tiles and writes
// ONE slab [64][3][3][8]; the gradient bucket's summing launch (yolo_wgrad_reduce_batched) adds the slabs.  No atomics.
#include "common.h"

namespace {

constexpr int SB_THREADS = 256;
constexpr int TH = 8, TW = 16, TPIX = TH * TW;          // pre-pool tile
constexpr int PR = TH / 2 + 2, PC = TW / 2 + 2;          // pooled positions whose windows can touch it
constexpr int C = 64, CV = 8;
constexpr int MAXP = (PR * PC * CV + SB_THREADS - 1) / SB_THREADS;      // pooled chunks per thread (2)
constexpr int MAXQ = TPIX * CV / SB_THREADS;                            // pre-pool chunks per thread (4)
constexpr int XR = 2 * TH + 1, XC = 2 * TW + 1;          // image patch (pixels)
constexpr int MAXX = (XR * XC + SB_THREADS - 1) / SB_THREADS;           // patch pixels per thread (3)
constexpr int NPOOL = PR * PC * CV;                      // staged pooled chunks
constexpr int SG_BYTES = NPOOL * 16, SA_BYTES = NPOOL * 8;
constexpr int IMG_BYTES = 64 * 256;                      // dy image for the transposed reads
constexpr int XP_BYTES = (XR * XC * 8 + 15) / 16 * 16;
constexpr int CST_BYTES = 3 * C * 4;                     // dy = A g + B y + D per channel (in LDS, not in 24+ loop-invariant registers per thread)
constexpr int SB_LDS = SG_BYTES + SA_BYTES + IMG_BYTES + XP_BYTES + CST_BYTES;
constexpr int SB_WG_PER_CU = 3;

struct StemBwdArgs {
  const bf16_t* dout; const bf16_t* out; const uint8_t* argmax;   // pooled [N][Ho][Wo][64]; out == null: no ReLU
  const bf16_t* y;                                                 // pre-pool conv output [N][H][W][64] (read only with BatchNorm)
  const float* a1; const float* mean; const float* rstd; const float* k1; const float* k2;   // a1 == null: no BatchNorm
  const bf16_t* x;                                                 // packed image [N][Hi][Wi][8]
  float* slabs;                                                    // [grid][64][3][3][8]
  int N, H, W, Ho, Wo, pt, pl, Hi, Wi, tiles_h, tiles_w, ntiles;
};

__device__ __forceinline__ uint4 ld16g(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }

// chunk swizzle of the [64][128 x 16 bit] image with plain 256-byte rows (cdna guide T10, image (b)): conflict-free for the 16-byte row
// writes of step 2 and for the transposed reads of step 3
__device__ __forceinline__ int img_f(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ bf16x8_t tr_frag(const char* img, int p0, int col0, int lane) {
  // lane l (g = l >> 4, i = l & 15) receives image[p0 + 8g + j][col0 + i], j = 0..7 (two 4 x 16 transposed block reads)
  const int gq = lane >> 4, i = lane & 15;
  const int r0 = p0 + 8 * gq + (i >> 2), r1 = r0 + 4;
  const int ch = (col0 >> 3) + ((i & 3) >> 1), hb = (i & 1) << 3;
  typedef s16x4_t __attribute__((address_space(3))) * lds_ptr_t;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(img + r0 * 256 + ((ch ^ img_f(r0)) << 4) + hb));
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(img + r1 * 256 + ((ch ^ img_f(r1)) << 4) + hb));
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  s16x8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, r);
}

__global__ __launch_bounds__(SB_THREADS, SB_WG_PER_CU) void stem_pool_bwd_wgrad_kernel(StemBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* sG = reinterpret_cast<uint4*>(smem);                 // [NPOOL] masked pooled gradient, 8 channels
  uint2* sA = reinterpret_cast<uint2*>(smem + SG_BYTES);      // [NPOOL] arg-max codes, 8 bytes (0xff: no window here)
  char* img = smem + SG_BYTES + SA_BYTES;
  char* xp = img + IMG_BYTES;                                 // [XR][XC] x 8 bytes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* cst = reinterpret_cast<float*>(xp + XP_BYTES);        // [3][64]: A = a, B = -a rstd k2, D = a (mean rstd k2 - k1)
  if (a.a1 && tid < C) {
    const float aa = a.a1[tid], rk = a.rstd[tid] * a.k2[tid];
    cst[tid] = aa; cst[C + tid] = -aa * rk; cst[2 * C + tid] = aa * (a.mean[tid] * rk - a.k1[tid]);
  }                                                           // (visible after the first tile's first barrier)

  f32x4_t acc[4][3];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[ct][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // B-fragment geometry of this lane (fixed over tiles): column n of column tile t = tap * 4 + ch; pixels 32 * wave + 8 * (lane >> 4) + j
  const int half = wave >> 1, p0 = (wave & 1) * 32, gq = lane >> 4;
  const int hl_b = half * 4 + (p0 >> 4) + (gq >> 1), wl_b = 8 * (gq & 1);
  int boff[3];
  bool bval[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int col = t * 16 + (lane & 15), tap = col >> 2, ch = col & 3;
    const int r = tap / 3, s = tap - 3 * r;
    bval[t] = tap < 9;
    boff[t] = bval[t] ? ((2 * hl_b + r) * XC + 2 * wl_b + s) * 8 + ch * 2 : 0;
  }
  const int cv = tid % CV, c = cv * 8;                        // SB_THREADS is a multiple of CV: a thread keeps its channel chunk

  // tiles are dealt so that the workgroups of one XCD (blockIdx.x & 7: workgroups go to the XCDs round-robin) walk a CONTIGUOUS range of
  // tiles: neighbouring tiles share pooled halo positions and image patch columns, which then hit that XCD's L2 (PMC: 493 MB fetched per
  // launch for ~330 MB of tiles with plain round-robin)
  int t_first = blockIdx.x, t_end = a.ntiles, t_step = gridDim.x;
  if ((gridDim.x & 7) == 0) {
    const int xcd = blockIdx.x & 7;
    t_first = (int)((long long)a.ntiles * xcd / 8) + (blockIdx.x >> 3);
    t_end = (int)((long long)a.ntiles * (xcd + 1) / 8);
    t_step = gridDim.x >> 3;
  }
  for (int tile = t_first; tile < t_end; tile += t_step) {
    int b = tile;
    const int tw = b % a.tiles_w; b /= a.tiles_w;
    const int th = b % a.tiles_h;
    const int n = b / a.tiles_h;
    const int h0 = th * TH, w0 = tw * TW;
    const int ho0 = max(0, (h0 + a.pt - 1) >> 1), wo0 = max(0, (w0 + a.pl - 1) >> 1);

    // ---- A: request everything this tile needs
    uint4 pg[MAXP], po[MAXP];
    uint2 pa[MAXP];
#pragma unroll
    for (int q = 0; q < MAXP; ++q) {
      const int i = tid + q * SB_THREADS, pp = i / CV;
      const int ho = ho0 + pp / PC, wo = wo0 + pp % PC;
      pg[q] = po[q] = make_uint4(0u, 0u, 0u, 0u);
      pa[q] = make_uint2(0xffffffffu, 0xffffffffu);
      if (i < NPOOL && ho < a.Ho && wo < a.Wo) {
        const size_t o = ((size_t)(n * a.Ho + ho) * a.Wo + wo) * C + c;
        pg[q] = ld16g(a.dout + o);
        pa[q] = *reinterpret_cast<const uint2*>(a.argmax + o);
        if (a.out) po[q] = ld16g(a.out + o);
      }
    }
    uint4 yv[MAXQ];
    bool okq[MAXQ];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
      const int pix = (tid + q * SB_THREADS) / CV;
      const int h = h0 + pix / TW, w = w0 + pix % TW;
      okq[q] = h < a.H && w < a.W;
      yv[q] = (okq[q] && a.a1) ? ld16g(a.y + ((size_t)(n * a.H + h) * a.W + w) * C + c) : make_uint4(0u, 0u, 0u, 0u);
    }
    uint2 xv[MAXX];
#pragma unroll
    for (int q = 0; q < MAXX; ++q) {
      const int i = tid + q * SB_THREADS;
      const int r = i / XC, cc = i - r * XC;
      const int hi = 2 * h0 + r, wi = 2 * w0 + cc;                    // 'same' at stride 2 on an even size pads bottom / right only
      xv[q] = (i < XR * XC && hi < a.Hi && wi < a.Wi) ? *reinterpret_cast<const uint2*>(a.x + ((size_t)(n * a.Hi + hi) * a.Wi + wi) * 8)
                                                      : make_uint2(0u, 0u);
    }
    // ---- B: the pooled chunks, ReLU-masked, into LDS (the previous tile's gather pass is behind its second barrier)
#pragma unroll
    for (int q = 0; q < MAXP; ++q) {
      const int i = tid + q * SB_THREADS;
      if (i < NPOOL) {
        uint4 g = pg[q];
        if (a.out) {
          const unsigned ow[4] = {po[q].x, po[q].y, po[q].z, po[q].w};
          unsigned gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const unsigned lo = ow[k] & 0xffffu, hi = ow[k] >> 16;
            gw[k] = ((lo != 0u && lo < 0x8000u) ? (gw[k] & 0xffffu) : 0u) | ((hi != 0u && hi < 0x8000u) ? (gw[k] & 0xffff0000u) : 0u);
          }
          g = make_uint4(gw[0], gw[1], gw[2], gw[3]);
        }
        sG[i] = g;
        sA[i] = pa[q];
      }
    }
    __syncthreads();      // pooled chunks staged; every wave has left the previous tile's MFMA phase (image / patch may be rewritten)
    // ---- C: gather + BatchNorm apply -> dy image; image patch -> LDS
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
      const int pix = (tid + q * SB_THREADS) / CV;
      const int hl = pix / TW, wl = pix % TW;
      float r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = 0.f;
      if (okq[q]) {
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = 0.f;
        const int hn = h0 + hl + a.pt, wn = w0 + wl + a.pl;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {                      // window class along h: dh in {0, 1}, then dh == 2 (even hn only)
          const int ho = (hn >> 1) - kh, dh = hn - 2 * ho;
          if ((kh && (hn & 1)) || ho < 0 || ho >= a.Ho) continue;
#pragma unroll
          for (int kw = 0; kw < 2; ++kw) {
            const int wo = (wn >> 1) - kw, dw = wn - 2 * wo;
            if ((kw && (wn & 1)) || wo < 0 || wo >= a.Wo) continue;
            const int li = ((ho - ho0) * PC + (wo - wo0)) * CV + cv;
            const uint2 am = sA[li];
            const uint4 dv = sG[li];
            const unsigned code = (unsigned)(dh * 3 + dw);
            const unsigned dw4[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const unsigned aj = ((j < 4 ? (am.x >> (8 * j)) : (am.y >> (8 * (j - 4)))) & 0xffu);
              const float d = (j & 1) ? hi2f(dw4[j >> 1]) : lo2f(dw4[j >> 1]);
              g[j] += aj == code ? d : 0.f;
            }
          }
        }
        if (a.a1) {
          float v[8];
          unpack_bf8(yv[q], v);
#pragma unroll
          for (int j = 0; j < 8; ++j) r[j] = cst[c + j] * g[j] + (cst[C + c + j] * v[j] + cst[2 * C + c + j]);
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) r[j] = g[j];
        }
      }
      const int row = (hl & 3) * 16 + wl, chunk = (hl >> 2) * 8 + cv;
      *reinterpret_cast<uint4*>(img + row * 256 + ((chunk ^ img_f(row)) << 4)) = pack_bf8(r);
    }
#pragma unroll
    for (int q = 0; q < MAXX; ++q) {
      const int i = tid + q * SB_THREADS;
      if (i < XR * XC) *reinterpret_cast<uint2*>(xp + i * 8) = xv[q];
    }
    __syncthreads();
    // ---- D: dW += dy^T x over this wave's 32 pixels (every lane takes part: the transposed reads need EXEC all ones)
    bf16x8_t bfrag[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      typedef unsigned short u16x8_t __attribute__((ext_vector_type(8)));
      u16x8_t v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const unsigned short*>(xp + boff[t] + j * 16);
      if (!bval[t]) v = u16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
      bfrag[t] = __builtin_bit_cast(bf16x8_t, v);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const bf16x8_t af = tr_frag(img, p0, half * 64 + ct * 16, lane);
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[ct][t] = YOLO_MFMA_16x16x32(af, bfrag[t], acc[ct][t]);
    }
  }

  // ---- the four waves' partials meet in LDS, one wave at a time; then the slab in the device weight layout [co][tap][8]
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);                  // [64 co][48]
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int idx = (ct * 16 + (lane >> 4) * 4 + j) * 48 + t * 16 + (lane & 15);
            red[idx] = (w == 0 ? 0.f : red[idx]) + acc[ct][t][j];
          }
    }
    __syncthreads();
  }
  float* slab = a.slabs + (size_t)blockIdx.x * (64 * 72);
  for (int i = tid; i < 64 * 72; i += SB_THREADS) {
    const int co = i / 72, rem = i - co * 72, tap = rem >> 3, ch = rem & 7;
    slab[i] = ch < 3 ? red[co * 48 + tap * 4 + ch] : 0.f;
  }
}

inline bool stem_bwd_ok(const yolo_conv_problem* p, int Cpool, int Ho, int Wo, int pt, int pl) {
  return p && p->Cin == 8 && p->C0 == 0 && p->Cout == 64 && Cpool == 64 && p->R == 3 && p->S == 3 && p->stride == 2 && p->pad_t == 0 &&
         p->pad_l == 0 && p->H % 2 == 0 && p->W % 2 == 0 && p->Ho == p->H / 2 && p->Wo == p->W / 2 && pt >= 0 && pt <= 1 && pl >= 0 && pl <= 1 &&
         Ho > 0 && Wo > 0 && (size_t)p->N * p->H * p->W * 8 < (1ull << 31);
}

inline int stem_bwd_grid(const yolo_conv_problem* p) {
  const int tiles = p->N * ((p->Ho + TH - 1) / TH) * ((p->Wo + TW - 1) / TW);
  return tiles < 256 * SB_WG_PER_CU ? tiles : 256 * SB_WG_PER_CU;
}

static_assert(SB_WG_PER_CU * (SB_LDS + 1024) <= 160 * 1024, "workgroups per CU (LDS, with allocation granularity)");

}  // namespace

// number of [64][3][3][8] float32 slabs yolo_stem_pool_bwd_wgrad writes for this stem (0: the fused form does not apply)
extern "C" int yolo_stem_pool_bwd_slabs(const yolo_conv_problem* p, int pooled_channels, int Ho, int Wo, int pad_t, int pad_l) {
  if (!stem_bwd_ok(p, pooled_channels, Ho, Wo, pad_t, pad_l)) return 0;
  return stem_bwd_grid(p);
}

extern "C" int yolo_stem_pool_bwd_wgrad(const yolo_conv_problem* p, const void* x, const void* dout, const void* out, const uint8_t* argmax,
                                        int relu, const void* y, const float* a1, const float* mean, const float* rstd, const float* k1,
                                        const float* k2, int Ho, int Wo, int pad_t, int pad_l, float* slabs, size_t slab_bytes, void* stream) {
  YOLO_CHECK_ARG(stem_bwd_ok(p, 64, Ho, Wo, pad_t, pad_l), "not a stem this kernel covers (yolo_stem_pool_bwd_slabs)");
  YOLO_CHECK_ARG(x && dout && argmax && slabs, "null pointer");
  YOLO_CHECK_ARG(!relu || out, "relu needs out");
  YOLO_CHECK_ARG(!a1 || (y && mean && rstd && k1 && k2), "BN branch incomplete");
  StemBwdArgs a;
  a.dout = (const bf16_t*)dout; a.out = relu ? (const bf16_t*)out : nullptr; a.argmax = argmax;
  a.y = (const bf16_t*)y; a.a1 = a1; a.mean = mean; a.rstd = rstd; a.k1 = k1; a.k2 = k2;
  a.x = (const bf16_t*)x; a.slabs = slabs;
  a.N = p->N; a.H = p->Ho; a.W = p->Wo; a.Ho = Ho; a.Wo = Wo; a.pt = pad_t; a.pl = pad_l; a.Hi = p->H; a.Wi = p->W;
  a.tiles_h = (p->Ho + TH - 1) / TH; a.tiles_w = (p->Wo + TW - 1) / TW;
  a.ntiles = p->N * a.tiles_h * a.tiles_w;
  const int grid = stem_bwd_grid(p);
  YOLO_CHECK_ARG(slab_bytes >= (size_t)grid * 64 * 72 * sizeof(float), "slab region too small (yolo_stem_pool_bwd_slabs)");
  hipLaunchKernelGGL(stem_pool_bwd_wgrad_kernel, dim3(grid), dim3(SB_THREADS), SB_LDS, (hipStream_t)stream, a);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}
