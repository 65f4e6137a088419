// Implicit-GEMM convolution for gfx950 (MI355X): forward, data-gradient and weight-gradient, NHWC bf16, fp32 accumulate.
//
// Replaces keras.layers.Conv2D (reference backbone/basic_backbone.py:42, yolov3/yolov3_detector.py:98-150) and its TF
// autodiff gradients.  One gather routine feeds all three passes: a "row" is a pixel of the row space (output pixels for
// fwd, input pixels for dgrad, output pixels for wgrad) and a "k-chunk" is 8 consecutive channels (16 bytes) of one
// kernel tap of the source tensor; the (optional) nearest 2x upsample + channel concat of the FPN necks
// (yolov3_detector.py:115-116,140-141) is resolved inside the gather, so the concatenated tensor never exists.
//
// fwd/dgrad kernel : 128 pixels x {128|64} channels per 256-thread workgroup (4 wave64), BK = 64, register-staged
//                    global->LDS with an XOR-swizzled image (conflict-free ds_read_b128), double-buffered LDS,
//                    v_mfma_f32_16x16x32_bf16 with the WEIGHT tile as the A operand so that every lane ends up holding
//                    4 consecutive output channels of one pixel (8-byte bf16 / 16-byte f32 NHWC stores).
//                    Epilogue options: bias, f32 output, accumulate (dgrad fan-in), BatchNorm partial statistics.
// wgrad kernel     : dW[co][kcol] += sum_pixels dY[pix][co] * X[pix][kcol]; both operands are pixel-major in memory,
//                    so fragments are read with ds_read_b64_tr_b16 (hardware transpose); split over pixel ranges,
//                    fp32 atomics into dW.
#include "common.h"

namespace {

struct Gather {
  const bf16_t* src0;  // half-resolution source of the first C0 channels (nullptr if C0 == 0)
  const bf16_t* src1;  // full-resolution source of the remaining C1 channels
  int Hs, Ws;          // spatial size of the (virtual, concatenated) source
  int C0, C1;
  int lgC8;            // log2((C0 + C1) / 8)
  int Ho, Wo;          // row space
  int S, RS;           // kernel width, taps
  int smul, pad_h, pad_w, den;  // src coord = (row * smul - pad + tap) / den  (valid iff divisible and in range)
  int M;               // rows
  int Kg;              // GEMM K = RS * (C0 + C1)
};

struct RowInfo { int n, hb, wb; };

__device__ __forceinline__ RowInfo decode_row(const Gather& g, int m) {
  RowInfo r;
  if (m >= g.M) { r.n = 0; r.hb = -(1 << 28); r.wb = -(1 << 28); return r; }
  int hw = g.Ho * g.Wo;
  r.n = m / hw;
  int rem = m - r.n * hw;
  int ho = rem / g.Wo;
  int wo = rem - ho * g.Wo;
  r.hb = ho * g.smul - g.pad_h;
  r.wb = wo * g.smul - g.pad_w;
  return r;
}

// One 16-byte k-chunk of one row; zero outside the image / outside K.
__device__ __forceinline__ uint4 gather_chunk(const Gather& g, const RowInfo& r, int tap_r, int tap_s, int c, bool kvalid) {
  int hn = r.hb + tap_r, wn = r.wb + tap_s;
  bool ok = kvalid & (hn >= 0) & (wn >= 0);
  if (g.den == 2) { ok = ok & (((hn | wn) & 1) == 0); hn >>= 1; wn >>= 1; }
  ok = ok & (hn < g.Hs) & (wn < g.Ws);
  uint4 v = make_uint4(0, 0, 0, 0);
  if (ok) {
    const bf16_t* p;
    if (c < g.C0) p = g.src0 + ((size_t)(r.n * (g.Hs >> 1) + (hn >> 1)) * (g.Ws >> 1) + (wn >> 1)) * g.C0 + c;
    else          p = g.src1 + ((size_t)(r.n * g.Hs + hn) * g.Ws + wn) * g.C1 + (c - g.C0);
    v = *reinterpret_cast<const uint4*>(p);
  }
  return v;
}

// XOR-swizzled [rows][64 bf16] image: 128-byte rows, 16-byte chunk index ^= row & 7.
__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

constexpr int BM = 128;  // pixels per workgroup tile
constexpr int BK = 64;   // K elements per stage

template <int BN, bool OUT_F32>
__global__ __launch_bounds__(256) void igemm_fwd_kernel(Gather g, const bf16_t* __restrict__ Wt, const float* __restrict__ bias,
                                                        void* __restrict__ Yv, int ldy, int accumulate,
                                                        float* __restrict__ stat_sum, float* __restrict__ stat_sq,
                                                        int Kout, int tiles_n) {
  constexpr int WM = (BN == 128) ? 2 : 4;  // waves along pixels
  constexpr int WN = 4 / WM;               // waves along channels
  constexpr int PT = BM / WM / 16;         // 16-pixel MFMA tiles per wave
  constexpr int CT = BN / WN / 16;         // 16-channel MFMA tiles per wave (= 4)
  constexpr int A_BYTES = BM * BK * 2;
  constexpr int B_BYTES = BN * BK * 2;
  constexpr int B_ROWS_PER_THREAD = BN / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int ccol = tid & 7;   // k-chunk column of this thread inside a stage
  const int rbase = tid >> 3; // first row handled (then +32 per i)
  RowInfo rows[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) rows[i] = decode_row(g, m0 + rbase + 32 * i);

  f32x4_t acc[CT][PT];
#pragma unroll
  for (int a = 0; a < CT; ++a)
#pragma unroll
    for (int b = 0; b < PT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = (g.Kg + BK - 1) / BK;
  const int cmask = (1 << g.lgC8) - 1;
  uint4 ra[4], rb[B_ROWS_PER_THREAD];

  auto load_stage = [&](int kt) {
    int q = kt * (BK / 8) + ccol;
    int tap = q >> g.lgC8;
    int c = (q & cmask) << 3;
    bool kvalid = tap < g.RS;
    int tr = tap / g.S, ts = tap - tr * g.S;
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = gather_chunk(g, rows[i], tr, ts, c, kvalid);
    int k = kt * BK + ccol * 8;
    bool kv = k < g.Kg;
#pragma unroll
    for (int i = 0; i < B_ROWS_PER_THREAD; ++i) {
      rb[i] = make_uint4(0, 0, 0, 0);
      if (kv) rb[i] = *reinterpret_cast<const uint4*>(Wt + (size_t)(n0 + rbase + 32 * i) * g.Kg + k);
    }
  };
  auto store_stage = [&](int buf) {
    char* sA = smem + buf * (A_BYTES + B_BYTES);
    char* sB = sA + A_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(sA + swz(rbase + 32 * i, ccol)) = ra[i];
#pragma unroll
    for (int i = 0; i < B_ROWS_PER_THREAD; ++i) *reinterpret_cast<uint4*>(sB + swz(rbase + 32 * i, ccol)) = rb[i];
  };
  auto compute_stage = [&](int buf) {
    const char* sA = smem + buf * (A_BYTES + B_BYTES);
    const char* sB = sA + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t wf[CT], pf[PT];
      const int ch = ks * 4 + (lane >> 4);
#pragma unroll
      for (int a = 0; a < CT; ++a) wf[a] = *reinterpret_cast<const bf16x8_t*>(sB + swz(wn * (CT * 16) + a * 16 + (lane & 15), ch));
#pragma unroll
      for (int b = 0; b < PT; ++b) pf[b] = *reinterpret_cast<const bf16x8_t*>(sA + swz(wm * (PT * 16) + b * 16 + (lane & 15), ch));
#pragma unroll
      for (int a = 0; a < CT; ++a)
#pragma unroll
        for (int b = 0; b < PT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[a], pf[b], acc[a][b], 0, 0, 0);
    }
  };

  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_stage(kt + 1);
    compute_stage(kt & 1);
    if (kt + 1 < nk) store_stage((kt + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds channels co..co+3 (rows of D) of pixel (column of D) ----
  const int cq = (lane >> 4) * 4;
  float ssum[CT][4], ssq[CT][4];
#pragma unroll
  for (int a = 0; a < CT; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) { ssum[a][j] = 0.f; ssq[a][j] = 0.f; }

#pragma unroll
  for (int b = 0; b < PT; ++b) {
    const int m = m0 + wm * (PT * 16) + b * 16 + (lane & 15);
    const bool mv = m < g.M;
#pragma unroll
    for (int a = 0; a < CT; ++a) {
      const int co = n0 + wn * (CT * 16) + a * 16 + cq;
      float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
      if (bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += bias[co + j];
      }
      if (mv) {
        if constexpr (OUT_F32) {
          float* y = reinterpret_cast<float*>(Yv) + (size_t)m * ldy + co;
          if (accumulate) { float4 o = *reinterpret_cast<float4*>(y); v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
          *reinterpret_cast<float4*>(y) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          bf16_t* y = reinterpret_cast<bf16_t*>(Yv) + (size_t)m * ldy + co;
          if (accumulate) {
            uint2 o = *reinterpret_cast<uint2*>(y);
            v[0] += __uint_as_float(o.x << 16); v[1] += __uint_as_float(o.x & 0xffff0000u);
            v[2] += __uint_as_float(o.y << 16); v[3] += __uint_as_float(o.y & 0xffff0000u);
          }
          uint2 o;
          o.x = pack_bf2(v[0], v[1]);
          o.y = pack_bf2(v[2], v[3]);
          *reinterpret_cast<uint2*>(y) = o;
          if (stat_sum) {  // statistics of the values as stored (bf16-rounded)
            float r0 = __uint_as_float(o.x << 16), r1 = __uint_as_float(o.x & 0xffff0000u);
            float r2 = __uint_as_float(o.y << 16), r3 = __uint_as_float(o.y & 0xffff0000u);
            ssum[a][0] += r0; ssum[a][1] += r1; ssum[a][2] += r2; ssum[a][3] += r3;
            ssq[a][0] += r0 * r0; ssq[a][1] += r1 * r1; ssq[a][2] += r2 * r2; ssq[a][3] += r3 * r3;
          }
        }
      }
    }
  }
  if (stat_sum) {
    const int prow = tile_m * WM + wm;
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = ssum[a][j], q = ssq[a][j];
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        if ((lane & 15) == 0) {
          const int co = n0 + wn * (CT * 16) + a * 16 + cq + j;
          stat_sum[(size_t)prow * Kout + co] = s;
          stat_sq[(size_t)prow * Kout + co] = q;
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// wgrad: D[co][kcol] = sum_p dY[p][co] * X[p][kcol]
// ------------------------------------------------------------------------------------------------------------------
constexpr int WG_BCO = 128;   // rows of D per workgroup (output channels)
constexpr int WG_BKC = 128;   // columns of D per workgroup (k-columns = (tap, ci))
constexpr int WG_BP = 64;     // pixels per stage (2 MFMA k-steps)
constexpr int WG_LD = 272;    // LDS row stride in bytes: 256 + 16 pad, 16-byte aligned

__device__ __forceinline__ bf16x8_t tr_frag(const char* img, int p0, int col0, int lane) {
  // lane l (g = l>>4, i = l&15) receives image[p0 + 8g + j][col0 + i], j = 0..7 (two 4x16 transposed block reads)
  const int gq = lane >> 4, i = lane & 15;
  const char* a = img + (p0 + 8 * gq + (i >> 2)) * WG_LD + (col0 + 4 * (i & 3)) * 2;
  typedef s16x4_t __attribute__((address_space(3))) * lds_ptr_t;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(a));
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(a + 4 * WG_LD));
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  s16x8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, r);
}

__global__ __launch_bounds__(256) void igemm_wgrad_kernel(Gather g, const bf16_t* __restrict__ dY, int ldy,
                                                          float* __restrict__ dW, int Kout, int steps_per_split) {
  constexpr int IMG = WG_BP * WG_LD;  // bytes of one [64 pix][128 col] image
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;  // wave tile: co [wr*64, +64) x kcol [wc*64, +64)
  const int kc0 = blockIdx.x * WG_BKC, co0 = blockIdx.y * WG_BCO;
  const int nsteps = (g.M + WG_BP - 1) / WG_BP;
  const int s_begin = blockIdx.z * steps_per_split;
  const int s_end = min(nsteps, s_begin + steps_per_split);
  if (s_begin >= s_end) return;

  const int ccol = tid & 15;  // 16-byte chunk column (of 16) in both images
  const int rbase = tid >> 4; // pixel row (then +16 per i)
  // this thread's fixed k-chunk of the X gather
  const int q = (kc0 >> 3) + ccol;
  const int tap = q >> g.lgC8;
  const int xc = (q & ((1 << g.lgC8) - 1)) << 3;
  const bool kvalid = tap < g.RS;
  const int tr = tap / g.S, ts = tap - tr * g.S;
  const int yc = co0 + ccol * 8;
  const bool yvalid = yc < Kout;

  f32x4_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  uint4 rx[4], ry[4];
  auto load_stage = [&](int st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = st * WG_BP + rbase + 16 * i;
      RowInfo r = decode_row(g, m);
      rx[i] = gather_chunk(g, r, tr, ts, xc, kvalid);
      ry[i] = make_uint4(0, 0, 0, 0);
      if (yvalid && m < g.M) ry[i] = *reinterpret_cast<const uint4*>(dY + (size_t)m * ldy + yc);
    }
  };
  auto store_stage = [&](int buf) {
    char* sX = smem + buf * 2 * IMG;
    char* sY = sX + IMG;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<uint4*>(sX + (rbase + 16 * i) * WG_LD + ccol * 16) = rx[i];
      *reinterpret_cast<uint4*>(sY + (rbase + 16 * i) * WG_LD + ccol * 16) = ry[i];
    }
  };
  auto compute_stage = [&](int buf) {
    const char* sX = smem + buf * 2 * IMG;
    const char* sY = sX + IMG;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t yf[4], xf[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) yf[a] = tr_frag(sY, ks * 32, wr * 64 + a * 16, lane);
#pragma unroll
      for (int b = 0; b < 4; ++b) xf[b] = tr_frag(sX, ks * 32, wc * 64 + b * 16, lane);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[a], xf[b], acc[a][b], 0, 0, 0);
    }
  };

  load_stage(s_begin);
  store_stage(0);
  __syncthreads();
  for (int st = s_begin; st < s_end; ++st) {
    const int buf = (st - s_begin) & 1;
    if (st + 1 < s_end) load_stage(st + 1);
    compute_stage(buf);
    if (st + 1 < s_end) store_stage(buf ^ 1);
    __syncthreads();
  }

  // D[row = co][col = kcol]: lane holds rows 4*(lane>>4)+j, column lane&15
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int kc = kc0 + wc * 64 + b * 16 + (lane & 15);
      if (kc < g.Kg) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int co = co0 + wr * 64 + a * 16 + (lane >> 4) * 4 + j;
          if (co < Kout) atomicAdd(dW + (size_t)co * g.Kg + kc, acc[a][b][j]);
        }
      }
    }
}

// [Cout][RS][Cin] -> [Cin][RS flipped][Cout], 32x32 tiles through LDS.
__global__ void repack_dgrad_kernel(const bf16_t* __restrict__ wf, bf16_t* __restrict__ wd, int Cout, int RS, int Cin) {
  __shared__ bf16_t tile[32][33];
  const int tap = blockIdx.z;
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += 8) {
    int co = co0 + r, ci = ci0 + threadIdx.x;
    tile[r][threadIdx.x] = (co < Cout && ci < Cin) ? wf[((size_t)co * RS + tap) * Cin + ci] : (bf16_t)0;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {
    int ci = ci0 + r, co = co0 + threadIdx.x;
    if (ci < Cin && co < Cout) wd[((size_t)ci * RS + (RS - 1 - tap)) * Cout + co] = tile[threadIdx.x][r];
  }
}

int ilog2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return ((1 << l) == v) ? l : -1;
}

int check_problem(const yolo_conv_problem* p) {
  YOLO_CHECK_ARG(p != nullptr, "null problem");
  YOLO_CHECK_ARG(p->N > 0 && p->H > 0 && p->W > 0 && p->Ho > 0 && p->Wo > 0, "non-positive dims");
  YOLO_CHECK_ARG(p->Cin > 0 && p->Cin % 8 == 0 && ilog2_exact(p->Cin / 8) >= 0, "Cin/8 must be a power of two");
  YOLO_CHECK_ARG(p->Cout > 0 && p->Cout % 64 == 0, "Cout must be a multiple of 64 (pad)");
  YOLO_CHECK_ARG(p->C0 >= 0 && p->C0 < p->Cin && p->C0 % 8 == 0, "bad C0");
  YOLO_CHECK_ARG(p->C0 == 0 || (p->H % 2 == 0 && p->W % 2 == 0), "upsample-concat needs even H, W");
  YOLO_CHECK_ARG(p->R >= 1 && p->S >= 1 && p->R <= 9 && p->S <= 9, "bad kernel size");
  YOLO_CHECK_ARG(p->stride == 1 || p->stride == 2, "stride must be 1 or 2");
  YOLO_CHECK_ARG(p->pad_t >= 0 && p->pad_l >= 0 && p->pad_t < p->R && p->pad_l < p->S, "bad padding");
  // every output pixel must map inside the padded input
  YOLO_CHECK_ARG((p->Ho - 1) * p->stride - p->pad_t < p->H && (p->Wo - 1) * p->stride - p->pad_l < p->W, "Ho/Wo too large");
  YOLO_CHECK_ARG((size_t)p->N * p->H * p->W * p->Cin < (1ull << 31) && (size_t)p->N * p->Ho * p->Wo * p->Cout < (1ull << 31),
                 "tensor too large for 32-bit row indexing");
  return YOLO_OK;
}

Gather fwd_gather(const yolo_conv_problem* p, const void* src0, const void* src1) {
  Gather g;
  g.src0 = (const bf16_t*)src0; g.src1 = (const bf16_t*)src1;
  g.Hs = p->H; g.Ws = p->W; g.C0 = p->C0; g.C1 = p->Cin - p->C0;
  g.lgC8 = ilog2_exact(p->Cin / 8);
  g.Ho = p->Ho; g.Wo = p->Wo; g.S = p->S; g.RS = p->R * p->S;
  g.smul = p->stride; g.pad_h = p->pad_t; g.pad_w = p->pad_l; g.den = 1;
  g.M = p->N * p->Ho * p->Wo; g.Kg = p->R * p->S * p->Cin;
  return g;
}

template <bool F32>
int launch_fwd(const Gather& g, const void* w, const float* bias, void* y, int ldy, int accumulate, float* ssum, float* ssq,
               int Kout, hipStream_t st) {
  const int tiles_m = (g.M + BM - 1) / BM;
  if (Kout % 128 == 0) {
    const int tn = Kout / 128;
    const size_t lds = 2 * (BM * BK * 2 + 128 * BK * 2);
    hipLaunchKernelGGL((igemm_fwd_kernel<128, F32>), dim3(tiles_m * tn), dim3(256), lds, st, g, (const bf16_t*)w, bias, y, ldy,
                       accumulate, ssum, ssq, Kout, tn);
  } else {
    const int tn = Kout / 64;
    const size_t lds = 2 * (BM * BK * 2 + 64 * BK * 2);
    hipLaunchKernelGGL((igemm_fwd_kernel<64, F32>), dim3(tiles_m * tn), dim3(256), lds, st, g, (const bf16_t*)w, bias, y, ldy,
                       accumulate, ssum, ssq, Kout, tn);
  }
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}

}  // namespace

extern "C" int yolo_conv2d_stat_rows(const yolo_conv_problem* p) {
  if (!p || p->Cout % 64 != 0) return YOLO_ERR_INVALID_ARG;
  const int tiles_m = (p->N * p->Ho * p->Wo + BM - 1) / BM;
  return tiles_m * ((p->Cout % 128 == 0) ? 2 : 4);
}

extern "C" int yolo_conv2d_fwd(const yolo_conv_problem* p, const void* src0, const void* src1, const void* w_fwd,
                               const float* bias, void* y, int y_is_f32, float* stat_sum, float* stat_sq, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  YOLO_CHECK_ARG(src1 && w_fwd && y, "null pointer");
  YOLO_CHECK_ARG(p->C0 == 0 || src0, "C0 > 0 needs src0");
  YOLO_CHECK_ARG((stat_sum == nullptr) == (stat_sq == nullptr), "stat_sum and stat_sq go together");
  YOLO_CHECK_ARG(!(y_is_f32 && stat_sum), "statistics are defined on bf16 outputs only");
  Gather g = fwd_gather(p, src0, src1);
  if (y_is_f32) return launch_fwd<true>(g, w_fwd, bias, y, p->Cout, 0, nullptr, nullptr, p->Cout, (hipStream_t)stream);
  return launch_fwd<false>(g, w_fwd, bias, y, p->Cout, 0, stat_sum, stat_sq, p->Cout, (hipStream_t)stream);
}

extern "C" int yolo_conv2d_dgrad(const yolo_conv_problem* p, const void* dy, const void* w_dgrad, void* dx, int accumulate,
                                 void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  YOLO_CHECK_ARG(dy && w_dgrad && dx, "null pointer");
  YOLO_CHECK_ARG(p->Cin % 64 == 0, "dgrad needs Cin % 64 == 0");
  YOLO_CHECK_ARG(ilog2_exact(p->Cout / 8) >= 0, "dgrad needs Cout/8 to be a power of two");
  // conv-transpose as a forward gather over dy with flipped taps: src = (row - (R-1-pad) + tap') / stride
  Gather g;
  g.src0 = nullptr; g.src1 = (const bf16_t*)dy;
  g.Hs = p->Ho; g.Ws = p->Wo; g.C0 = 0; g.C1 = p->Cout;
  g.lgC8 = ilog2_exact(p->Cout / 8);
  g.Ho = p->H; g.Wo = p->W; g.S = p->S; g.RS = p->R * p->S;
  g.smul = 1; g.pad_h = p->R - 1 - p->pad_t; g.pad_w = p->S - 1 - p->pad_l; g.den = p->stride;
  g.M = p->N * p->H * p->W; g.Kg = p->R * p->S * p->Cout;
  return launch_fwd<false>(g, w_dgrad, nullptr, dx, p->Cin, accumulate, nullptr, nullptr, p->Cin, (hipStream_t)stream);
}

extern "C" int yolo_conv2d_wgrad(const yolo_conv_problem* p, const void* src0, const void* src1, const void* dy, float* dw,
                                 int split_k, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  YOLO_CHECK_ARG(src1 && dy && dw, "null pointer");
  YOLO_CHECK_ARG(p->C0 == 0 || src0, "C0 > 0 needs src0");
  Gather g = fwd_gather(p, src0, src1);
  const int tiles_k = (g.Kg + WG_BKC - 1) / WG_BKC;
  const int tiles_c = (p->Cout + WG_BCO - 1) / WG_BCO;
  const int nsteps = (g.M + WG_BP - 1) / WG_BP;
  if (split_k <= 0) {  // aim at ~3 workgroups per CU, at least 8 pixel-steps per workgroup
    split_k = (768 + tiles_k * tiles_c - 1) / (tiles_k * tiles_c);
    const int max_split = (nsteps + 7) / 8;
    if (split_k > max_split) split_k = max_split;
    if (split_k < 1) split_k = 1;
  }
  if (split_k > nsteps) split_k = nsteps;
  const int sps = (nsteps + split_k - 1) / split_k;
  split_k = (nsteps + sps - 1) / sps;
  YOLO_CHECK_ARG(split_k <= 65535, "split_k too large");
  const size_t lds = 2 * 2 * WG_BP * WG_LD;
  hipLaunchKernelGGL(igemm_wgrad_kernel, dim3(tiles_k, tiles_c, split_k), dim3(256), lds, (hipStream_t)stream, g,
                     (const bf16_t*)dy, p->Cout, dw, p->Cout, sps);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}

extern "C" int yolo_repack_dgrad_weights(const void* w_fwd, void* w_dgrad, int Cout, int R, int S, int Cin, void* stream) {
  YOLO_CHECK_ARG(w_fwd && w_dgrad && Cout > 0 && Cin > 0 && R > 0 && S > 0, "bad argument");
  hipLaunchKernelGGL(repack_dgrad_kernel, dim3((Cin + 31) / 32, (Cout + 31) / 32, R * S), dim3(32, 8), 0, (hipStream_t)stream,
                     (const bf16_t*)w_fwd, (bf16_t*)w_dgrad, Cout, R * S, Cin);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}
