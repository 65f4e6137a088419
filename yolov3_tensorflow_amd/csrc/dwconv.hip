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
#ifndef DW_MIN_WAVES
#define DW_MIN_WAVES 2
#endif

__device__ __forceinline__ uint4 ld16(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }

constexpr int SP = 8;   // output pixels per lane (a horizontal strip)

// Both kernels work on LDS-resident row tiles: a workgroup stages TH (+ K - 1 halo) image rows of cvb 8-channel chunks, zero-padded to
// [-K/2, 8*strips + K/2) columns, as 16-byte chunks [row][column][chunk] with a row pitch == cvb (mod 16) chunks, so that lanes that
// differ in (row, chunk) hit different bank quads.  The first version read every input chunk K*(SP+K-1)/SP times from global memory
// (18x for K = 9) and, in the weight gradient, made one pass over the data per kernel row with an 81-round serial block reduction:
// 100 / 490 us per launch on the 104 x 104 map against ~25 us of HBM time.
struct TilePlan { int TH, cvb, xpitch, W8, nsub, per_sub; };   // rows per tile, chunks per workgroup, x row pitch (chunks), 8*strips,
                                                                // channel sub-blocks of the group, workgroups per sub-block

__host__ __device__ inline TilePlan plan_tile(int H, int W, int K, int cv, int blocks) {
  TilePlan t;
  t.W8 = (W + SP - 1) / SP * SP;
  t.cvb = t.W8 <= 32 ? (cv < 4 ? cv : 4) : (t.W8 <= 64 ? (cv < 2 ? cv : 2) : 1);
  const int wp = (t.W8 + K - 1) * t.cvb;
  t.xpitch = (wp + 15) / 16 * 16 + t.cvb;
  const int budget = 4096;                                   // 64 KiB of 16-byte chunks for x + dy / x alone
  int th = (budget - (K - 1) * t.xpitch) / (t.W8 * t.cvb + t.xpitch);
  t.TH = th < 1 ? 1 : (th > H ? H : th);
  t.nsub = cv / t.cvb;
  t.per_sub = blocks / t.nsub < 1 ? 1 : blocks / t.nsub;
  return t;
}

__device__ __forceinline__ void stage_rows(uint4* dst, int pitch, const bf16_t* __restrict__ src, int n, int h_first, int nrows, int col_first,
                                           int ncols, int H, int W, int C, int c, int cvb) {
  // dst[row][col][cb] = src[n][h_first + row][col_first + col][c + 8 cb .. +7], zeros outside the image
  const int total = nrows * ncols * cvb;
  for (int i = threadIdx.x; i < total; i += DW_THREADS) {
    const int cb = i % cvb;
    int t = i / cvb;
    const int col = t % ncols, row = t / ncols;
    const int h = h_first + row, w = col_first + col;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (h >= 0 && h < H && w >= 0 && w < W) v = ld16(src + ((size_t)(n * H + h) * W + w) * C + c + cb * 8);
    dst[row * pitch + col * cvb + cb] = v;
  }
}

// fwd / dgrad for one kernel size: lane = (row, strip of SP pixels, chunk); per kernel row it reads the SP + K - 1 input chunks of its
// strip from LDS once and slides the window in registers.  dgrad is the same kernel with flipped taps (stride 1, symmetric padding).
template <int K>
__device__ __forceinline__ void dw_tile(const MixP& p, const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ y,
                                        int c0, int cg, int cv, int flip, int accumulate, uint4* smem) {
  constexpr int PAD = K / 2;
  const TilePlan t = plan_tile(p.H, p.W, K, cv, gridDim.x);
  const int strips = t.W8 / SP, ncols = t.W8 + K - 1;
  uint4* sw = smem;                                   // [K*K][cvb] weights of this sub-block (taps already flipped for dgrad)
  uint4* sx = smem + K * K * t.cvb;
  const int sub = blockIdx.x % t.nsub, idx = blockIdx.x / t.nsub;
  if (idx >= t.per_sub) return;
  const int cs = sub * t.cvb;                         // first chunk of the sub-block inside the group
  for (int i = threadIdx.x; i < K * K * t.cvb; i += DW_THREADS) {
    const int cb = i % t.cvb, tap = i / t.cvb;
    const int r = tap / K, q = tap - r * K;
    sw[i] = ld16(w + (size_t)((flip ? K - 1 - r : r) * K + (flip ? K - 1 - q : q)) * cg + (cs + cb) * 8);
  }
  const int tiles_h = (p.H + t.TH - 1) / t.TH, ntiles = p.N * tiles_h;
  for (int tile = idx; tile < ntiles; tile += t.per_sub) {
    const int n = tile / tiles_h, h0 = (tile - n * tiles_h) * t.TH;
    __syncthreads();                                  // previous tile fully consumed (and the weights staged)
    stage_rows(sx, t.xpitch, x, n, h0 - PAD, t.TH + K - 1, -PAD, ncols, p.H, p.W, p.C, c0 + cs * 8, t.cvb);
    __syncthreads();
    const int rows = min(t.TH, p.H - h0);
    for (int it = threadIdx.x; it < rows * strips * t.cvb; it += DW_THREADS) {
      const int cb = it % t.cvb;
      const int r = (it / t.cvb) % rows, sidx = it / (t.cvb * rows);
      float acc[SP][8];
#pragma unroll
      for (int o = 0; o < SP; ++o)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[o][j] = 0.f;
#pragma unroll 1
      for (int dh = 0; dh < K; ++dh) {          // a real loop: unrolled, the compiler hoists all K*K weight chunks and spills
        float wt[K][8];
#pragma unroll
        for (int q = 0; q < K; ++q) unpack_bf8(sw[(dh * K + q) * t.cvb + cb], wt[q]);
        const uint4* xr = sx + (r + dh) * t.xpitch + (sidx * SP) * t.cvb + cb;
#pragma unroll
        for (int ic = 0; ic < SP + K - 1; ++ic) {
          float xv[8];
          unpack_bf8(xr[ic * t.cvb], xv);
#pragma unroll
          for (int q = 0; q < K; ++q) {
            const int o = ic - q;                     // output pixel this (input column, tap) pair feeds
            if (o >= 0 && o < SP) {
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[o][j] += xv[j] * wt[q][j];
            }
          }
        }
      }
      bf16_t* yrow = y + ((size_t)(n * p.H + h0 + r) * p.W) * p.C + c0 + (cs + cb) * 8;
#pragma unroll
      for (int o = 0; o < SP; ++o) {
        const int wq = sidx * SP + o;
        if (wq < p.W) {
          bf16_t* yo = yrow + (size_t)wq * p.C;
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
}

__global__ __launch_bounds__(DW_THREADS, DW_MIN_WAVES) void dwconv_mix_kernel(MixP p, const bf16_t* __restrict__ x, const bf16_t* __restrict__ w0,
                                                                const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2,
                                                                const bf16_t* __restrict__ w3, bf16_t* __restrict__ y, int flip, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) uint4 dw_smem[];
  const int grp = blockIdx.y;
  const int c0 = p.split[grp], cg = p.split[grp + 1] - c0, cv = cg >> 3;
  if (cv == 0) return;
  const bf16_t* w = grp == 0 ? w0 : (grp == 1 ? w1 : (grp == 2 ? w2 : w3));
  switch (p.ksize[grp]) {   // blockIdx.y-uniform
    case 1: dw_tile<1>(p, x, w, y, c0, cg, cv, flip, accumulate, dw_smem); break;
    case 3: dw_tile<3>(p, x, w, y, c0, cg, cv, flip, accumulate, dw_smem); break;
    case 5: dw_tile<5>(p, x, w, y, c0, cg, cv, flip, accumulate, dw_smem); break;
    case 7: dw_tile<7>(p, x, w, y, c0, cg, cv, flip, accumulate, dw_smem); break;
    default: dw_tile<9>(p, x, w, y, c0, cg, cv, flip, accumulate, dw_smem); break;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Tiled forward / data gradient (the default).  The row-tile kernel above gives every workgroup ONE 8-channel chunk of wide maps (16-byte
// accesses at a 128-byte stride), the same 256 workgroups to every kernel size although k = 9 carries 9x the work per channel of k = 3, and
// compiles all sizes into one register budget (256 VGPRs: one wave per SIMD): 100 us on the 104 x 104 x 64 map against ~20 us of HBM time.
// Here a workgroup owns an 8 x 16 output tile of one image for a SLAB of 64 consecutive channels (all kernel sizes that fall into it):
//   * the halo'd input tile is staged as whole 128-byte pixels (coalesced), zero outside the image; the slab's weights are staged as
//     float32 (flipped for the data gradient), so lanes read them with broadcast LDS reads instead of unpacking them again and again
//   * work is handed out per WAVE: a wave-task is up to 64 lanes of ONE kernel size -- lane = (chunk, row, strip of SP pixels), SP = 8 / 4 / 2
//     for k = 3 / 5 / 7,9, so that every kernel size fills a wave -- and the tasks are dealt to the four waves longest-first (the schedule
//     is a few integers computed by every thread from the channel split: no host tables)
//   * outputs go straight to global memory, 16 bytes per (pixel, chunk); lanes of a task are ordered chunk-fastest
// ------------------------------------------------------------------------------------------------------------------
constexpr int MT_TH = 8, MT_TW = 16, MT_WAVES = 4;

__host__ __device__ inline int mt_sp(int K, bool pure_small) { return K >= 7 ? 2 : (K == 5 ? 4 : (pure_small ? 4 : 8)); }
__host__ __device__ inline int mt_group_of(const MixP& p, int channel) {
  return channel >= p.split[3] ? 3 : (channel >= p.split[2] ? 2 : (channel >= p.split[1] ? 1 : 0));
}

template <int K, int SP>
__device__ __forceinline__ void mt_task(const MixP& p, const uint4* __restrict__ sx, int pitch, int ppx, int pad, const float* __restrict__ swf,
                                        int first, int units, bf16_t* __restrict__ y, int n, int h0, int w0, int chunk0, int accumulate, int lane) {
  constexpr int STRIPS = MT_TW / SP;
  const int items = units * MT_TH * STRIPS;
  const int off = pad - K / 2;                                  // this kernel size's window inside the slab's halo
  for (int i = lane; i < items; i += 64) {
    const int u = i % units, rs = i / units;
    const int r = rs / STRIPS, sidx = rs - r * STRIPS;
    const int j = first + u;                                    // chunk inside the slab
    const float* wj = swf + (size_t)u * K * K * 8;              // this chunk's float32 taps [K][K][8]
    float acc[SP][8];
#pragma unroll
    for (int o = 0; o < SP; ++o)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[o][c] = 0.f;
#pragma unroll 1
    for (int dh = 0; dh < K; ++dh) {            // a real loop: unrolled, the compiler hoists all K*K weight rows and spills
      float wt[K][8];
#pragma unroll
      for (int q = 0; q < K; ++q) {
        const float4 wa = *reinterpret_cast<const float4*>(wj + (dh * K + q) * 8), wb = *reinterpret_cast<const float4*>(wj + (dh * K + q) * 8 + 4);
        wt[q][0] = wa.x; wt[q][1] = wa.y; wt[q][2] = wa.z; wt[q][3] = wa.w; wt[q][4] = wb.x; wt[q][5] = wb.y; wt[q][6] = wb.z; wt[q][7] = wb.w;
      }
      const uint4* xr = sx + (r + dh + off) * pitch + (sidx * SP + off) * ppx + j;
#pragma unroll
      for (int ic = 0; ic < SP + K - 1; ++ic) {
        float xv[8];
        unpack_bf8(xr[ic * ppx], xv);
#pragma unroll
        for (int q = 0; q < K; ++q) {
          const int o = ic - q;
          if (o >= 0 && o < SP) {
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[o][c] += xv[c] * wt[q][c];
          }
        }
      }
    }
    const int h = h0 + r;
    if (h < p.H) {
      bf16_t* yrow = y + ((size_t)(n * p.H + h) * p.W) * p.C + (size_t)(chunk0 + j) * 8;
#pragma unroll
      for (int o = 0; o < SP; ++o) {
        const int wq = w0 + sidx * SP + o;
        if (wq < p.W) {
          bf16_t* yo = yrow + (size_t)wq * p.C;
          if (accumulate) {
            float old[8];
            unpack_bf8(ld16(yo), old);
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[o][c] += old[c];
          }
          *reinterpret_cast<uint4*>(yo) = pack_bf8(acc[o]);
        }
      }
    }
  }
}

__global__ __launch_bounds__(MT_WAVES * 64, 2) void dwconv_mix_tiled_kernel(MixP p, const bf16_t* __restrict__ x, const bf16_t* __restrict__ w0,
                                                                          const bf16_t* __restrict__ w1, const bf16_t* __restrict__ w2,
                                                                          const bf16_t* __restrict__ w3, bf16_t* __restrict__ y, int flip,
                                                                          int accumulate, int tiles_w, int tiles_h) {
  extern __shared__ __attribute__((aligned(16))) uint4 dw_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cvs = p.C >> 3;
  const int chunk0 = blockIdx.y * 8, nch = min(8, cvs - chunk0);
  // kernel sizes of the slab's chunks -> halo and the wave-task table, ONCE per workgroup (the kernel-argument lookups behind p.ksize[group]
  // are dependent scalar loads: recomputed per tile they cost more than the tile's arithmetic)
  __shared__ int t_K[16], t_first[16], t_units[16], t_woff[16], t_owner[16], t_meta[4];
  if (tid == 0) {
    int pad_ = 0, nt_ = 0, woff = 0, total = 0;
    bool small = true;
    for (int j = 0; j < nch; ++j) {
      const int K = p.ksize[mt_group_of(p, (chunk0 + j) * 8)];
      pad_ = max(pad_, K / 2);
      small = small && K <= 3;
      total += K * K;
    }
    for (int j = 0; j < nch;) {                                 // runs of equal kernel size cut into <= 64-lane pieces
      const int K = p.ksize[mt_group_of(p, (chunk0 + j) * 8)];
      int run = 1;
      while (j + run < nch && p.ksize[mt_group_of(p, (chunk0 + j + run) * 8)] == K) ++run;
      const int upw = max(1, 64 / (MT_TH * (MT_TW / mt_sp(K, small))));          // units (chunks) per wave pass
      for (int f = 0; f < run; f += upw, ++nt_) {
        t_K[nt_] = K; t_first[nt_] = j + f; t_units[nt_] = min(upw, run - f); t_woff[nt_] = woff + f * K * K * 8;
      }
      woff += run * K * K * 8;
      j += run;
    }
    for (int k = 0; k < nt_; ++k) {                             // dealt in snake order from the most expensive end (kernel sizes grow with the
      const int ph = (nt_ - 1 - k) % (2 * MT_WAVES);            // chunk index): 0 1 2 3 3 2 1 0 ...
      t_owner[k] = ph < MT_WAVES ? ph : 2 * MT_WAVES - 1 - ph;
    }
    t_meta[0] = pad_; t_meta[1] = nt_; t_meta[2] = small ? 1 : 0; t_meta[3] = total;
  }
  __syncthreads();
  const int pad = t_meta[0], nt = t_meta[1];
  const bool pure_small = t_meta[2] != 0;
  // ---- stage: float32 weights of the slab (taps flipped for the data gradient), then the halo'd input tile ----
  const int trows = MT_TH + 2 * pad, tcols = MT_TW + 2 * pad;
  const int ppx = nch + 1;                                      // chunks per LDS pixel: one pad chunk, so that lanes 2 / 4 / 8 pixels apart
  const int pitch = tcols * ppx + 1;                            // (and rows) land on different bank quads
  uint4* sx = dw_smem;
  float* swf = reinterpret_cast<float*>(dw_smem + trows * pitch);
  {
    // one flat pass over all taps of all chunks (a loop per chunk would wait for global memory once per chunk)
    const int total = t_meta[3];
    for (int t = tid; t < total; t += MT_WAVES * 64) {
      int j = 0, base = 0;
      for (;;) {
        const int K = p.ksize[mt_group_of(p, (chunk0 + j) * 8)];
        if (t < base + K * K) break;
        base += K * K; ++j;
      }
      const int g = mt_group_of(p, (chunk0 + j) * 8), K = p.ksize[g], cg = p.split[g + 1] - p.split[g];
      const bf16_t* w = g == 0 ? w0 : (g == 1 ? w1 : (g == 2 ? w2 : w3));
      const int cl = (chunk0 + j) * 8 - p.split[g];            // first channel of the chunk inside its group
      const int tap = t - base, r = tap / K, q = tap - r * K;
      float f[8];
      unpack_bf8(ld16(w + (size_t)((flip ? K - 1 - r : r) * K + (flip ? K - 1 - q : q)) * cg + cl), f);
      float* d = swf + (size_t)t * 8;
#pragma unroll
      for (int c = 0; c < 8; ++c) d[c] = f[c];
    }
  }
  const int ntiles = p.N * tiles_h * tiles_w;
  for (int tile_id = blockIdx.x; tile_id < ntiles; tile_id += gridDim.x) {   // persistent over tiles: weights and schedule are per workgroup
  int tile = tile_id;
  const int tx = tile % tiles_w; tile /= tiles_w;
  const int ty = tile % tiles_h;
  const int n = tile / tiles_h;
  const int h0 = ty * MT_TH, wc0 = tx * MT_TW;
  __syncthreads();                                              // the previous tile's readers are done (and the weights are staged)
  {
    // thread = (chunk j fixed, pixel q advancing by threads / nch): row / column by carry, no divisions in the loop
    const int step = (MT_WAVES * 64) / nch;                     // nch divides 256 (1, 2, 4 or 8 chunks)
    const int j = tid % nch;
    int q = tid / nch;
    int row = q / tcols, col = q - row * tcols;
    const int drow = step / tcols, dcol = step - drow * tcols;
    for (; row < trows; ) {
      const int h = h0 - pad + row, wq = wc0 - pad + col;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (h >= 0 && h < p.H && wq >= 0 && wq < p.W) v = ld16(x + ((size_t)(n * p.H + h) * p.W + wq) * p.C + (size_t)(chunk0 + j) * 8);
      sx[row * pitch + col * ppx + j] = v;
      row += drow; col += dcol;
      if (col >= tcols) { col -= tcols; ++row; }
    }
  }
  __syncthreads();
  // ---- wave-tasks ----
  for (int k = 0; k < nt; ++k) {
    if (t_owner[k] != wave) continue;                           // wave-uniform
    const int K = t_K[k], first = t_first[k], units = t_units[k];
    const float* wk = swf + t_woff[k];
#define MT_RUN(K_, SP_) mt_task<K_, SP_>(p, sx, pitch, ppx, pad, wk, first, units, y, n, h0, wc0, chunk0, accumulate, lane)
    switch (K) {
      case 1: MT_RUN(1, 8); break;
      case 3: if (pure_small) MT_RUN(3, 4); else MT_RUN(3, 8); break;
      case 5: MT_RUN(5, 4); break;
      case 7: MT_RUN(7, 2); break;
      default: MT_RUN(9, 2); break;
    }
#undef MT_RUN
  }
  }   // tiles
}

// wgrad for one kernel size, ONE pass over the data: thread = (pixel lane, kernel row dh, chunk); per (row, strip) item it reads the dy
// strip (shared by the K kernel-row threads: LDS broadcast) and the x row segment of its kernel row and accumulates the K taps of that
// row for 8 channels in registers across all tiles of the workgroup; one LDS reduction over the pixel lanes at the very end, and the
// workgroup's totals go to its slab slot with plain stores (a second kernel sums the slabs: deterministic, no atomics).
template <int K>
__device__ __forceinline__ void dw_wgrad_tile(const MixP& p, const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ slab0,
                                              int c0, int cg, int cv, uint4* smem) {
  constexpr int PAD = K / 2;
  const TilePlan t = plan_tile(p.H, p.W, K, cv, gridDim.x);
  const int strips = t.W8 / SP, ncols = t.W8 + K - 1;
  uint4* sx = smem;
  uint4* sdy = smem + (t.TH + K - 1) * t.xpitch;
  const int dpitch = t.W8 * t.cvb;
  const int sub = blockIdx.x % t.nsub, idx = blockIdx.x / t.nsub;
  if (idx >= t.per_sub) return;
  const int cs = sub * t.cvb;
  const int L = DW_THREADS / (K * t.cvb);             // pixel lanes
  const int cb = threadIdx.x % t.cvb, dh = (threadIdx.x / t.cvb) % K, li = threadIdx.x / (t.cvb * K);
  float acc[K][8];
#pragma unroll
  for (int d = 0; d < K; ++d)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[d][j] = 0.f;
  const int tiles_h = (p.H + t.TH - 1) / t.TH, ntiles = p.N * tiles_h;
  for (int tile = idx; tile < ntiles; tile += t.per_sub) {
    const int n = tile / tiles_h, h0 = (tile - n * tiles_h) * t.TH;
    __syncthreads();
    stage_rows(sx, t.xpitch, x, n, h0 - PAD, t.TH + K - 1, -PAD, ncols, p.H, p.W, p.C, c0 + cs * 8, t.cvb);
    stage_rows(sdy, dpitch, dy, n, h0, t.TH, 0, t.W8, p.H, p.W, p.C, c0 + cs * 8, t.cvb);
    __syncthreads();
    const int rows = min(t.TH, p.H - h0);
    if (li < L) {
      for (int it = li; it < rows * strips; it += L) {
        const int r = it / strips, sidx = it - r * strips;
        float g[SP][8];
        const uint4* gr = sdy + r * dpitch + (sidx * SP) * t.cvb + cb;
#pragma unroll
        for (int o = 0; o < SP; ++o) unpack_bf8(gr[o * t.cvb], g[o]);
        const uint4* xr = sx + (r + dh) * t.xpitch + (sidx * SP) * t.cvb + cb;
#pragma unroll
        for (int ic = 0; ic < SP + K - 1; ++ic) {
          float xv[8];
          unpack_bf8(xr[ic * t.cvb], xv);
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
  }
  // totals over the pixel lanes: red[thread][8] per tap column d, summed by thread u = (dh, cb, j)
  float* red = reinterpret_cast<float*>(smem);
  float* slab = slab0 + (size_t)idx * K * K * cg;
#pragma unroll
  for (int d = 0; d < K; ++d) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = li < L ? acc[d][j] : 0.f;
    __syncthreads();
    for (int u = threadIdx.x; u < K * t.cvb * 8; u += DW_THREADS) {
      const int j = u & 7, cbu = (u >> 3) % t.cvb, dhu = (u >> 3) / t.cvb;
      float sum = 0.f;
      for (int q = 0; q < L; ++q) sum += red[(((q * K) + dhu) * t.cvb + cbu) * 8 + j];
      slab[(size_t)(dhu * K + d) * cg + (cs + cbu) * 8 + j] = sum;
    }
  }
}

__global__ __launch_bounds__(DW_THREADS, DW_MIN_WAVES) void dwconv_mix_wgrad_kernel(MixP p, const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                      float* __restrict__ s0, float* __restrict__ s1, float* __restrict__ s2,
                                                                      float* __restrict__ s3) {
  extern __shared__ __attribute__((aligned(16))) uint4 dw_smem[];
  const int grp = blockIdx.y;
  const int c0 = p.split[grp], cg = p.split[grp + 1] - c0, cv = cg >> 3;
  if (cv == 0) return;
  float* slab = grp == 0 ? s0 : (grp == 1 ? s1 : (grp == 2 ? s2 : s3));
  switch (p.ksize[grp]) {
    case 1: dw_wgrad_tile<1>(p, x, dy, slab, c0, cg, cv, dw_smem); break;
    case 3: dw_wgrad_tile<3>(p, x, dy, slab, c0, cg, cv, dw_smem); break;
    case 5: dw_wgrad_tile<5>(p, x, dy, slab, c0, cg, cv, dw_smem); break;
    case 7: dw_wgrad_tile<7>(p, x, dy, slab, c0, cg, cv, dw_smem); break;
    default: dw_wgrad_tile<9>(p, x, dy, slab, c0, cg, cv, dw_smem); break;
  }
}

// dW_g[i] (+)= sum over the per_sub slabs of group g (blockIdx.y)
struct SlabSet { const float* slab[4]; float* dw[4]; int n[4], count[4]; };
__global__ __launch_bounds__(256) void dwconv_mix_wgrad_reduce_kernel(SlabSet s, int accumulate) {
  const int g = blockIdx.y;
  const float* src = g == 0 ? s.slab[0] : (g == 1 ? s.slab[1] : (g == 2 ? s.slab[2] : s.slab[3]));
  float* dst = g == 0 ? s.dw[0] : (g == 1 ? s.dw[1] : (g == 2 ? s.dw[2] : s.dw[3]));
  const int n = g == 0 ? s.n[0] : (g == 1 ? s.n[1] : (g == 2 ? s.n[2] : s.n[3]));
  const int cnt = g == 0 ? s.count[0] : (g == 1 ? s.count[1] : (g == 2 ? s.count[2] : s.count[3]));
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // 8 loads in flight per lane: the loop is pure latency
    int q = 0;
    for (; q + 7 < cnt; q += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += src[(size_t)(q + u) * n + i];
    }
    for (; q < cnt; ++q) a[0] += src[(size_t)q * n + i];
    dst[i] = (accumulate ? dst[i] : 0.f) + (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7])));
  }
}

constexpr int DW_BLOCKS = 256;   // workgroups per channel group (grid.x); every one loops over its share of the row tiles
constexpr size_t DW_LDS = 4096 * 16 + 81 * 4 * 16;   // tile budget + the largest weight block (K = 9, cvb = 4)

int g_dw_grid = 512;
int g_dw_tiled = 1;              // "dw_tiled": 1 (default) tiled forward / data gradient, 0 the row-tile kernel

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

// the tile budget is above the 64 KiB default limit of dynamic LDS
int allow_big_lds() {
  static int rc = -1;
  if (rc < 0) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv_mix_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DW_LDS);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv_mix_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DW_LDS);
    if (e != hipSuccess) { yolo_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
    rc = 0;
  }
  return rc;
}

int launch_mix(const yolo_mixconv_problem* p, const void* x, const void* w0, const void* w1, const void* w2, const void* w3, void* y, int flip,
               int accumulate, void* stream) {
  int rc = check_mix(p);
  if (rc) return rc;
  YOLO_CHECK_ARG(x && y && w0 && w1 && w2 && w3, "null pointer");
  if ((rc = allow_big_lds())) return rc;
  if (g_dw_tiled) {
    const int tiles_h = (p->H + MT_TH - 1) / MT_TH, tiles_w = (p->W + MT_TW - 1) / MT_TW, slabs = (p->C / 8 + 7) / 8;
    // LDS: halo'd tile of the widest kernel (pad 4) x 8 chunks + float32 taps of 8 chunks of the largest kernel
    const size_t lds = (size_t)(MT_TH + 8) * ((MT_TW + 8) * 9 + 1) * 16 + (size_t)8 * 81 * 8 * 4;
    static bool attr = false;
    if (!attr) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dwconv_mix_tiled_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) { yolo_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
      attr = true;
    }
    const int ntile = p->N * tiles_h * tiles_w, gx = ntile < g_dw_grid ? ntile : g_dw_grid;      // persistent: a few workgroups per CU and slab loop over the tiles
    hipLaunchKernelGGL(dwconv_mix_tiled_kernel, dim3(gx, slabs), dim3(MT_WAVES * 64), lds, (hipStream_t)stream, to_dev(p),
                       (const bf16_t*)x, (const bf16_t*)w0, (const bf16_t*)w1, (const bf16_t*)w2, (const bf16_t*)w3, (bf16_t*)y, flip, accumulate,
                       tiles_w, tiles_h);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
  }
  hipLaunchKernelGGL(dwconv_mix_kernel, dim3(DW_BLOCKS, 4), dim3(DW_THREADS), DW_LDS, (hipStream_t)stream, to_dev(p), (const bf16_t*)x,
                     (const bf16_t*)w0, (const bf16_t*)w1, (const bf16_t*)w2, (const bf16_t*)w3, (bf16_t*)y, flip, accumulate);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}

// slab counts / sizes of the 4 groups for the two-phase weight gradient
void wgrad_slabs(const yolo_mixconv_problem* p, int (&count)[4], int (&n)[4]) {
  for (int g = 0; g < 4; ++g) {
    const int cg = p->split[g + 1] - p->split[g], K = p->ksize[g];
    n[g] = K * K * cg;
    count[g] = cg ? plan_tile(p->H, p->W, K, cg / 8, DW_BLOCKS).per_sub : 0;
  }
}

}  // namespace

int yolo_dw_set_tiled(int on) { if (on > 1) { g_dw_grid = on; g_dw_tiled = 1; } else g_dw_tiled = on ? 1 : 0; return YOLO_OK; }   // > 1: persistent grid per slab

extern "C" int yolo_dwconv_mix_fwd(const yolo_mixconv_problem* p, const void* x, const void* w0, const void* w1, const void* w2,
                                   const void* w3, void* y, void* stream) {
  return launch_mix(p, x, w0, w1, w2, w3, y, 0, 0, stream);
}

extern "C" int yolo_dwconv_mix_dgrad(const yolo_mixconv_problem* p, const void* dy, const void* w0, const void* w1, const void* w2,
                                     const void* w3, void* dx, int accumulate, void* stream) {
  return launch_mix(p, dy, w0, w1, w2, w3, dx, 1, accumulate, stream);
}

extern "C" size_t yolo_dwconv_mix_wgrad_workspace_bytes(const yolo_mixconv_problem* p) {
  if (check_mix(p)) return 0;
  int count[4], n[4];
  wgrad_slabs(p, count, n);
  size_t f = 0;
  for (int g = 0; g < 4; ++g) f += (size_t)count[g] * n[g];
  return f * sizeof(float);
}

extern "C" int yolo_dwconv_mix_wgrad(const yolo_mixconv_problem* p, const void* x, const void* dy, float* dw0, float* dw1, float* dw2,
                                     float* dw3, void* workspace, size_t workspace_bytes, int accumulate, void* stream) {
  int rc = check_mix(p);
  if (rc) return rc;
  YOLO_CHECK_ARG(x && dy && dw0 && dw1 && dw2 && dw3 && workspace, "null pointer");
  YOLO_CHECK_ARG(workspace_bytes >= yolo_dwconv_mix_wgrad_workspace_bytes(p), "workspace too small (yolo_dwconv_mix_wgrad_workspace_bytes)");
  if ((rc = allow_big_lds())) return rc;
  SlabSet s;
  wgrad_slabs(p, s.count, s.n);
  float* ws = (float*)workspace;
  float* dws[4] = {dw0, dw1, dw2, dw3};
  int nmax = 1;
  for (int g = 0; g < 4; ++g) {
    s.slab[g] = ws;
    s.dw[g] = dws[g];
    ws += (size_t)s.count[g] * s.n[g];
    if (s.n[g] > nmax) nmax = s.n[g];
  }
  hipLaunchKernelGGL(dwconv_mix_wgrad_kernel, dim3(DW_BLOCKS, 4), dim3(DW_THREADS), DW_LDS, (hipStream_t)stream, to_dev(p), (const bf16_t*)x,
                     (const bf16_t*)dy, (float*)s.slab[0], (float*)s.slab[1], (float*)s.slab[2], (float*)s.slab[3]);
  YOLO_LAUNCH_CHECK();
  hipLaunchKernelGGL(dwconv_mix_wgrad_reduce_kernel, dim3((nmax + 255) / 256, 4), dim3(256), 0, (hipStream_t)stream, s, accumulate);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}
