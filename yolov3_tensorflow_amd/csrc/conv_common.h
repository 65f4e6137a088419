// Pieces shared by the convolution kernels of conv_igemm.hip, conv_stream.hip and conv_s32.hip (included into each translation unit; everything lives
// in an anonymous namespace): the gather description, the BatchNorm epilogue arguments, address helpers and the tile epilogue.
#pragma once
#include "common.h"
#include <type_traits>
#include <string.h>

namespace yoloconv {

struct Gather {
  const bf16_t* src0;  // half-resolution source of the first C0 channels (nullptr if C0 == 0)
  const bf16_t* src1;  // full-resolution source of the remaining C1 channels
  int Hs, Ws;          // spatial size of the (virtual, concatenated) source
  int C0, C1;
  int lgC8;            // log2((C0 + C1) / 8)
  int Ho, Wo;          // row space
  int S, RS;           // kernel width, taps
  int smul, pad_h, pad_w, den;  // src coord = (row * smul - pad + tap) / den  (valid iff divisible and in range)
  int M;               // rows (< 2^24: row decode uses float reciprocals)
  int Kg;              // GEMM K = RS * (C0 + C1)
  float rhw, rw;       // 1 / (Ho * Wo), 1 / Wo
  int magicS;          // tap / S == (tap * magicS) >> 16 for tap < 128
  // ---- stride-2 data gradient as 4 parity classes (blockIdx.y = 2 * (h & 1) + (w & 1) of the output pixel): each class is a dense
  // stride-1 correlation over dY with 1 or 2 of the 3 taps per dimension, written to every other row / column of dX; the kernel
  // specialises its copy of this struct per class (s2 == 0: everything below is unused)
  int role;            // 0 = forward launch, 1 = data gradient (kernel selection only: the data gradient runs beside the weight-gradient stream)
  int bnepi;           // 1 = the launch carries the fused BatchNorm-backward reduce (kernel selection only; must agree between the rows query and the launch)
  int s2;
  int s2_ny;           // classes launched (grid y): 4, or 1 = only the even / even class (1x1 stride-2: the other positions get nothing)
  int N, S_full;       // images; tap columns of the (flipped) weight tensor
  int wKg;             // weight row stride in elements (9 * C)
  struct Dim { int n, pad, size, t0, t1; } rowd[2], cold[2];   // per parity: taps, padding, class grid size, flipped tap indices
  int OH, OW;          // dX spatial size
};

// per-launch view of the class a workgroup works on
struct ClassView { int on, wbase, wdr, wds, two, Hc, Wc, OH, OW, ph, pw; float rhw, rw; };

// BatchNorm-backward reduce fused into the data-gradient epilogue: the tile being written IS dL/d(out) of a BN(+ReLU)(+residual) unit, so
// the epilogue masks it with the unit's ReLU sign bits (mask: one byte per 8-channel chunk, or null), stores the masked gradient g and
// leaves the per-channel partial sums  sum g,  sum g xhat  (and  sum g xhat2  of a shortcut BN) of its tile in partial[row][3][ld] --
// the separate reduce pass over dout / out / y (and the grid barrier of the single-launch form) disappears.  partial == null: off.
struct BnEpi {
  const uint8_t* mask;
  const bf16_t* y;  const float* mean;  const float* rstd;
  const bf16_t* y2; const float* mean2; const float* rstd2;
  float* partial;
  const bf16_t* addend;   // fan-in source of an accumulating data gradient when it is not the output buffer itself (any instantiation)
  // exact accumulators instead of partial rows (common.h yolo_acc_*): the fused reduce adds its 2 / 3 tile sums there when `partial` is null;
  // a FORWARD launch (no fused reduce) with stat_sum == null adds its sum / sum of squares there (Q = 2)
  long long* acc;
  // two-level partial rows: group > 0 = a row buffer holds [P = ceil(rows / group) group rows][rows raw rows][counters];
  // the workgroup that completes a group of `group` consecutive pixel tiles folds their raw rows into the group row (conv_common.h rows_fold),
  // so that the BatchNorm kernel that consumes the statistics sums <= ~85 rows in its own prologue and no finalize launch is needed
  int group, rows;
};
struct Epi { float* ssum; float* ssq; BnEpi bn; };

}  // namespace yoloconv

// ---- weights-in-registers streaming kernel of the 64-channel 3x3 / stride-1 layers (conv_stream.hip), dispatched from conv_igemm.hip ----
struct StreamPlanOut { int span, nx, ny; size_t lds; };
// 0 = the streaming kernel does not take this problem, else the pixels per workgroup (statistics / partial rows = ceil(M / that))
int yolo_stream_plan(const yoloconv::Gather& g, int Kout, bool f32, StreamPlanOut* out);
int yolo_stream_launch(const yoloconv::Gather& g, const void* w, void* y, int ldy, int accumulate, const yoloconv::Epi& e, int Kout, hipStream_t st);
extern int g_stream;                  // "stream" tuning (conv_stream.hip)

// ---- 32x32x16 / 64 x 64 wave-tile 3x3 stride-1 kernel (conv_s32.hip), dispatched from conv_igemm.hip ----
struct S32PlanOut { int id, bm, bn, tiles; size_t lds; };
// 0 = the kernel does not take this problem, else the pixels per tile (statistics / partial rows = ceil(M / that))
int yolo_s32_plan(const yoloconv::Gather& g, int Kout, bool f32, S32PlanOut* out);
int yolo_s32_launch(const yoloconv::Gather& g, const void* w, void* y, int ldy, int accumulate, const yoloconv::Epi& e, int Kout, hipStream_t st);
extern int g_s32;                     // "s32" tuning (conv_s32.hip)

// ---- stationary-output weight gradient of the 3x3 stride-1 layers (conv_wgrad9.hip), dispatched from conv_igemm.hip ----
// 0 = the kernel does not take this problem, else the number of pixel splits (= slabs)
int yolo_wgrad9_plan(const yolo_conv_problem* p, int* sps_out);
int yolo_wgrad9_launch(const yolo_conv_problem* p, const void* x, const void* dy, float* out, long long slab, hipStream_t stream);
extern int g_wgrad9, g_wgrad9_wgs;    // "wgrad9" / "wgrad9_wgs" tuning (conv_wgrad9.hip)

namespace {
using yoloconv::Gather; using yoloconv::ClassView; using yoloconv::BnEpi; using yoloconv::Epi;

struct RowInfo { int n, hb, wb; };

// full-rate 24-bit multiply-add (operands < 2^24; the 32-bit v_mul_lo_u32 is quarter rate)
__device__ __forceinline__ unsigned mad24(unsigned a, unsigned b, unsigned c) { return __umul24(a, b) + c; }

// q = m / d, r = m % d with a float reciprocal + one correction step (exact for m < 2^24): ~8 instructions instead of the ~40 of an
// integer division -- the tile prologue decodes 4 rows per lane and was dominated by divisions
__device__ __forceinline__ void fast_divmod(int m, int d, float rd, int& q, int& r) {
  q = (int)((float)m * rd);
  r = m - q * d;
  if (r < 0) { --q; r += d; }
  if (r >= d) { ++q; r -= d; }
}

__device__ __forceinline__ RowInfo decode_row(const Gather& g, int m) {
  RowInfo r;
  if (m >= g.M) { r.n = 0; r.hb = -(1 << 28); r.wb = -(1 << 28); return r; }
  int rem, ho, wo;
  fast_divmod(m, g.Ho * g.Wo, g.rhw, r.n, rem);
  fast_divmod(rem, g.Wo, g.rw, ho, wo);
  r.hb = ho * g.smul - g.pad_h;
  r.wb = wo * g.smul - g.pad_w;
  return r;
}

// XOR-swizzled [rows][64 bf16] image: 128-byte rows, 16-byte chunk index ^= row & 7.
__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// 16 bytes of zeros in global memory: the source of every out-of-image / out-of-K chunk of the LDS-DMA gathers
__device__ __attribute__((aligned(16))) uint4 g_zero16 = {0u, 0u, 0u, 0u};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA of 16 bytes per lane through a raw buffer descriptor over [base, base + bytes): lanes whose byte offset is out of range
// (e.g. 0x80000000) write zeros -- probed on gfx950 (tools/probes/buffer_lds_probe.hip).  The descriptor is wave-uniform (4 SGPRs).
__device__ __forceinline__ void buffer_load_lds16(const void* base, unsigned bytes, void* lds, unsigned voffset) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)lds, 16, voffset, 0, 0, 0);
}

// address of one 16-byte k-chunk of one row, or the zero chunk
__device__ __forceinline__ const bf16_t* gather_addr(const Gather& g, const RowInfo& r, int tap_r, int tap_s, int c, bool kvalid) {
  int hn = r.hb + tap_r, wn = r.wb + tap_s;
  bool ok = kvalid & (hn >= 0) & (wn >= 0);
  if (g.den == 2) { ok = ok & (((hn | wn) & 1) == 0); hn >>= 1; wn >>= 1; }
  ok = ok & (hn < g.Hs) & (wn < g.Ws);
  const bf16_t* p = reinterpret_cast<const bf16_t*>(&g_zero16);
  if (ok) {
    if (c < g.C0) p = g.src0 + ((size_t)(r.n * (g.Hs >> 1) + (hn >> 1)) * (g.Ws >> 1) + (wn >> 1)) * g.C0 + c;
    else          p = g.src1 + ((size_t)(r.n * g.Hs + hn) * g.Ws + wn) * g.C1 + (c - g.C0);
  }
  return p;
}

// ---- two-level partial rows ---------------------------------------------------------------------------------------------------------
// Row buffer layout for nq quantities q[k] (separate arrays, or one array with a quantity stride), row stride rs floats:
//   rows [0, P)          group rows, P = ceil(R / G)              (what the consumer reads)
//   rows [P, P + R)      raw rows, one per pixel tile             (written by every workgroup with write-through stores)
//   after row P + R of q[0]: P * tiles_n int32 arrival counters   (zero between launches: the last arriver resets its counter)
// A workgroup (pixel tile `row`, channel tile `tile_n`, channels [n0, n0 + cols)) calls rows_fold after its raw-row stores.  The one whose
// arrival completes its group sums the group's raw rows IN ROW ORDER -- whoever does it, the sum is the same bits: run-to-run deterministic,
// unlike float atomics -- and stores the group row with plain stores for the NEXT kernel.  No workgroup ever waits for another one.
// Visibility follows the cdna guide's write-through form (Guideline 16 / microarch 'Valid forms', counter row): raw rows stored with agent-scope
// relaxed atomic stores (global_store ... sc1), every storing wave drains vmcnt(0), workgroup barrier, ONE lane adds to the counter
// (relaxed, agent), the last arriver learns it from the value its add returned and reads the rows with sc1 loads behind a workgroup barrier.
constexpr int YOLO_ROW_GROUP_MAX = 64;
__host__ __device__ inline int yolo_row_groups(int R, int G) { return (R + G - 1) / G; }
__device__ __forceinline__ void row_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <int NT>
__device__ __forceinline__ void rows_fold(float* q0, float* q1, float* q2, int nq, size_t rs, int R, int G, int row, int tile_n, int tiles_n,
                                          int n0, int cols, int tid, int* s_flag) {
  const int P = yolo_row_groups(R, G), grp = row / G;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's raw-row stores have left
  __syncthreads();
  if (tid == 0) {
    int* cnt = reinterpret_cast<int*>(q0 + (size_t)(P + R) * rs) + grp * tiles_n + tile_n;
    const int expect = min(G, R - grp * G);
    const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == expect - 1 ? 1 : 0;
    if (last) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // self-cleaning: zero again for the next launch
    *s_flag = last;
  }
  __syncthreads();
  if (!*s_flag) return;
  const int r0 = grp * G, nr = min(R, r0 + G) - r0;
  for (int e = tid; e < nq * cols; e += NT) {
    const int k = e / cols, cl = e - k * cols;
    float* const base = (k == 0 ? q0 : (k == 1 ? q1 : q2)) + n0 + cl;
    float t = 0.f;
    int r = 0;
    for (; r + 8 <= nr; r += 8) {                          // 8 loads in flight, summed in row order
      float f[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) f[u] = __hip_atomic_load(base + (size_t)(P + r0 + r + u) * rs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int u = 0; u < 8; ++u) t += f[u];
    }
    for (; r < nr; ++r) t += __hip_atomic_load(base + (size_t)(P + r0 + r) * rs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    base[(size_t)grp * rs] = t;
  }
}

// XCD-aware tile order (MI355X deals consecutive workgroups round-robin over its 8 XCDs, each with a private L2): give every XCD a
// contiguous run of tiles so that the tiles sharing a pixel panel / weight panel hit the same L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// Weight-gradient grids: logical (x = k tile, y = co tile, z = pixel split).  Every workgroup of one split reads the same pixels of X and
// dY, so with xcd != 0 the grid is launched 1-D and each XCD gets a contiguous run of logical ids (x fastest, then y, then z): the
// tiles of a split land on one XCD close in time and share its L2 instead of fetching the range once per XCD (PMC: the 64-channel
// 104 x 104 weight gradient fetched 259 MB for 88 MB of operands -- one read per kernel row).
struct Bid3 { int x, y, z; };
__device__ __forceinline__ Bid3 wgrad_block(int gx, int gy, int xcd) {
  if (!xcd) return {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
  const int l = xcd_remap(blockIdx.x, gridDim.x);
  const int x = l % gx, t = l / gx;
  return {x, t % gy, t / gy};
}

constexpr int BK = 64;  // K elements per stage

// Tile epilogue shared by the conv kernels: lane holds channels co..co+3 (rows of D) of pixel (column of D); acc[a][b] = channel tile a x
// pixel tile b of this wave (wave grid WM pixels x WN channels).  bf16 outputs are staged through LDS (the operand buffers are free by
// then), optional BatchNorm partial statistics go to one row per pixel tile.
// (SG > 1: the waves of SG k-groups share the pixel / channel positions (wm, wn); group sg stages only its pixel fragments [b_lo, b_hi) and
// the statistics of the groups are summed)
template <int BM, int BN, int NW, int WM, int WN, int PT, int CT, bool OUT_F32, bool BNEPI = false, int SG = 1>
__device__ __forceinline__ void tile_epilogue(f32x4_t (&acc)[CT][PT], char* smem, int M, int m0, int n0, int tile_m, const float* __restrict__ bias,
                                              void* __restrict__ Yv, int ldy, int accumulate, float* __restrict__ stat_sum,
                                              float* __restrict__ stat_sq, int Kout, int tid, int lane, int wm, int wn, const ClassView& cv,
                                              const BnEpi& bn, int prow, int sg = 0, int b_lo = 0, int b_hi = PT) {
  const int cq = (lane >> 4) * 4;
  const bool fwd_acc = !BNEPI && !stat_sum && bn.acc;        // forward statistics into accumulators
  const bool want_stats = stat_sum || fwd_acc;
  float ssum[CT][4], ssq[CT][4];
#pragma unroll
  for (int a = 0; a < CT; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) { ssum[a][j] = 0.f; ssq[a][j] = 0.f; }

  if constexpr (OUT_F32) {   // float32 logits (+ bias): 16 bytes per lane, written straight from the accumulators
#pragma unroll
    for (int b = 0; b < PT; ++b) {
      const int m = m0 + wm * (PT * 16) + b * 16 + (lane & 15);
      if (m < M) {
#pragma unroll
        for (int a = 0; a < CT; ++a) {
          const int co = n0 + wn * (CT * 16) + a * 16 + cq;
          float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
          if (bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += bias[co + j];
          }
          float* y = reinterpret_cast<float*>(Yv) + (size_t)m * ldy + co;
          if (accumulate) { float4 o = *reinterpret_cast<float4*>(y); v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
          *reinterpret_cast<float4*>(y) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
  } else {
    // bf16 activations: the tile is transposed through LDS (free now) so that every store instruction writes whole 128/256-byte
    // NHWC rows -- straight from the accumulators each instruction wrote 16 scattered 32-byte segments, which cost more than the
    // MFMAs of the whole tile on the 64-channel layers (ablation: 39 of 70 us)
    constexpr int OLD = BN * 2 + 16;                  // LDS row stride in bytes (16-byte pad: conflict-free 8-byte writes)
    // fused BatchNorm reduce: geometry of the write loop (a thread keeps its channel chunk and walks rows) and the global reads of its
    // first NB rows (y, the fan-in addend, the sign byte), requested NOW so that they fly under the staging of the tile
    // (BM need not be a multiple of the E_RG rows a pass covers: rows >= BM of the last pass are skipped)
    constexpr int E_CPR = BN / 8, E_RG = NW * 64 / E_CPR, E_ITER = (BM + E_RG - 1) / E_RG, E_NB = E_ITER < 4 ? E_ITER : 4;
    const int e_ch = tid % E_CPR, e_rg = tid / E_CPR, e_c = n0 + e_ch * 8;
    uint4 e_yv[E_NB], e_ev[E_NB];
    unsigned e_mk[E_NB], e_off[E_NB];
    auto e_load = [&](int it0) {
#pragma unroll
      for (int k = 0; k < E_NB; ++k) {
        const int row = e_rg + (it0 + k) * E_RG, m = m0 + row;
        e_off[k] = 0xffffffffu;                        // element offsets (the host refuses tensors of 2^31 elements or more)
        if (m < M && row < BM) {
          int mo = m;
          if (cv.on) {
            int n_, rem, hh, ww;
            fast_divmod(m, cv.Hc * cv.Wc, cv.rhw, n_, rem);
            fast_divmod(rem, cv.Wc, cv.rw, hh, ww);
            mo = (n_ * cv.OH + 2 * hh + cv.ph) * cv.OW + 2 * ww + cv.pw;
          }
          e_off[k] = (unsigned)mo * (unsigned)ldy + (unsigned)e_c;
          e_yv[k] = *reinterpret_cast<const uint4*>(bn.y + e_off[k]);
          if (accumulate) e_ev[k] = *reinterpret_cast<const uint4*>((bn.addend ? bn.addend : reinterpret_cast<const bf16_t*>(Yv)) + e_off[k]);
          e_mk[k] = bn.mask ? (unsigned)bn.mask[e_off[k] >> 3] : 0xffu;
        }
      }
    };
    if constexpr (BNEPI) e_load(0);
    __syncthreads();                                  // every wave is done with the operand ring
#pragma unroll
    for (int b = 0; b < PT; ++b) {
      if (b < b_lo || b >= b_hi) continue;            // (a k-group stages the fragments it holds the final sums of)
      const int pl = wm * (PT * 16) + b * 16 + (lane & 15);
#pragma unroll
      for (int a = 0; a < CT; ++a) {
        const int cl = wn * (CT * 16) + a * 16 + cq;
        uint2 o;
        o.x = pack_bf2(acc[a][b][0], acc[a][b][1]);
        o.y = pack_bf2(acc[a][b][2], acc[a][b][3]);
        *reinterpret_cast<uint2*>(smem + pl * OLD + cl * 2) = o;
        if (want_stats && !accumulate && m0 + pl < M) {   // statistics of the values as stored (bf16-rounded)
          const float r0 = lo2f(o.x), r1 = hi2f(o.x);
          const float r2 = lo2f(o.y), r3 = hi2f(o.y);
          ssum[a][0] += r0; ssum[a][1] += r1; ssum[a][2] += r2; ssum[a][3] += r3;
          ssq[a][0] += r0 * r0; ssq[a][1] += r1 * r1; ssq[a][2] += r2 * r2; ssq[a][3] += r3 * r3;
        }
      }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;                       // 16-byte chunks per row
    bf16_t* Y = reinterpret_cast<bf16_t*>(Yv);
    if constexpr (BNEPI) {   // data gradient of a BatchNorm unit's output: mask, store g, partial sums of g and g xhat
      constexpr int RG = E_RG, ITER = E_ITER, NB = E_NB;
      static_assert(RG * BN * 4 <= BM * (BN * 2 + 16), "epilogue geometry");
      const int ch = e_ch, rg = e_rg, c = e_c;
      float mu[8], rs[8], s0[8], s1[8], s2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { mu[j] = bn.mean[c + j]; rs[j] = bn.rstd[c + j]; s0[j] = s1[j] = s2[j] = 0.f; }
#pragma unroll
      for (int it0 = 0; it0 < ITER; it0 += NB) {
        if (it0 > 0) e_load(it0);                     // (256 x 128 tiles only: the second batch of rows)
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          if (e_off[k] != 0xffffffffu) {
            const int row = rg + (it0 + k) * RG;
            uint4 v = *reinterpret_cast<const uint4*>(smem + row * OLD + ch * 16);
            float g8[8], y8[8];
            if (accumulate) {                         // gradient fan-in: float32 add, one rounding (as the plain path below)
              unpack_bf8(v, g8);
              unpack_bf8(e_ev[k], y8);
#pragma unroll
              for (int j = 0; j < 8; ++j) g8[j] += y8[j];
              v = pack_bf8(g8);
            }
            unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
              w4[q] = (((e_mk[k] >> (2 * q)) & 1u) ? (w4[q] & 0xffffu) : 0u) | (((e_mk[k] >> (2 * q + 1)) & 1u) ? (w4[q] & 0xffff0000u) : 0u);
            v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
            *reinterpret_cast<uint4*>(Y + e_off[k]) = v;
            unpack_bf8(v, g8);
            unpack_bf8(e_yv[k], y8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s0[j] += g8[j]; s1[j] += g8[j] * ((y8[j] - mu[j]) * rs[j]); }
            if (bn.y2) {                              // shortcut BatchNorm of a down-sampling block (3 units per step): loaded in place
              unpack_bf8(*reinterpret_cast<const uint4*>(bn.y2 + e_off[k]), y8);
#pragma unroll
              for (int j = 0; j < 8; ++j) s2[j] += g8[j] * ((y8[j] - bn.mean2[c + j]) * bn.rstd2[c + j]);
            }
          }
        }
      }
      float* red = reinterpret_cast<float*>(smem);    // [RG][BN], one quantity at a time
      const int nq = bn.y2 ? 3 : 2;
      const bool grouped = bn.group > 0 && bn.partial;
      const int rrow = grouped ? yolo_row_groups(bn.rows, bn.group) + prow : prow;      // (grouped: the raw row behind the group rows)
      for (int q = 0; q < nq; ++q) {
        __syncthreads();                              // the staged tile (q = 0) / the previous quantity has been read
#pragma unroll
        for (int j = 0; j < 8; ++j) red[rg * BN + ch * 8 + j] = q == 0 ? s0[j] : (q == 1 ? s1[j] : s2[j]);
        __syncthreads();
        for (int cl = tid; cl < BN; cl += NW * 64) {
          float t = 0.f;
#pragma unroll 8
          for (int r = 0; r < RG; ++r) t += red[r * BN + cl];
          if (grouped) row_store(bn.partial + ((size_t)rrow * 3 + q) * ldy + n0 + cl, t);
          else if (bn.partial) bn.partial[((size_t)prow * 3 + q) * ldy + n0 + cl] = t;
          else yolo_acc_add(bn.acc, 3, ldy, prow % YOLO_ACC_NB, q, n0 + cl, t);
        }
      }
      if (grouped) {
        __syncthreads();                              // (red is free: its first word takes the last-arriver flag)
        rows_fold<NW * 64>(bn.partial, bn.partial + ldy, bn.partial + 2 * (size_t)ldy, nq, (size_t)3 * ldy, bn.rows, bn.group, prow, n0 / BN,
                           Kout / BN, n0, BN, tid, reinterpret_cast<int*>(smem));
      }
      return;
    }
    for (int i = tid; i < BM * CPR; i += NW * 64) {
      const int row = i / CPR, ch = i - row * CPR;
      const int m = m0 + row;
      if (m < M) {
        uint4 v = *reinterpret_cast<const uint4*>(smem + row * OLD + ch * 16);
        int mo = m;
        if (cv.on) {                                  // parity class: pixel (n, h', w') of the class grid -> (n, 2h' + ph, 2w' + pw) of dX
          int n_, rem, hh, ww;
          fast_divmod(m, cv.Hc * cv.Wc, cv.rhw, n_, rem);
          fast_divmod(rem, cv.Wc, cv.rw, hh, ww);
          mo = (n_ * cv.OH + 2 * hh + cv.ph) * cv.OW + 2 * ww + cv.pw;
        }
        bf16_t* yp = Y + (size_t)mo * ldy + n0 + ch * 8;
        if (accumulate) {                             // gradient fan-in: y += tile (float32 add, one rounding)
          float a8[8], b8[8];
          unpack_bf8(v, a8);
          unpack_bf8(*reinterpret_cast<const uint4*>(bn.addend ? bn.addend + ((size_t)mo * ldy + n0 + ch * 8) : yp), b8);
#pragma unroll
          for (int j = 0; j < 8; ++j) a8[j] += b8[j];
          v = pack_bf8(a8);
        }
        *reinterpret_cast<uint4*>(yp) = v;
      }
    }
  }

  if (want_stats) {  // one partial row per workgroup: lanes -> 16-lane groups by shuffle, waves along the pixel axis through LDS
    __syncthreads();                                  // every wave is done with the staged output tile
    float* red = reinterpret_cast<float*>(smem);      // [2][SG * WM][BN]
    constexpr int WMS = SG * WM;
    const int wms = sg * WM + wm;
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = ssum[a][j], q = ssq[a][j];
        s = row16_sum(s);
        q = row16_sum(q);
        if ((lane & 15) == 0) {
          const int cl = wn * (CT * 16) + a * 16 + cq + j;
          red[wms * BN + cl] = s;
          red[(WMS + wms) * BN + cl] = q;
        }
      }
    __syncthreads();
    const bool grouped = bn.group > 0 && stat_sum;
    for (int cl = tid; cl < BN; cl += NW * 64) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < WMS; ++w) { s += red[w * BN + cl]; q += red[(WMS + w) * BN + cl]; }
      if (grouped) {
        const size_t rrow = (size_t)yolo_row_groups(bn.rows, bn.group) + tile_m;
        row_store(stat_sum + rrow * Kout + n0 + cl, s);
        row_store(stat_sq + rrow * Kout + n0 + cl, q);
      } else if (stat_sum) {
        stat_sum[(size_t)tile_m * Kout + n0 + cl] = s;
        stat_sq[(size_t)tile_m * Kout + n0 + cl] = q;
      } else {
        yolo_acc_add(bn.acc, 2, Kout, tile_m % YOLO_ACC_NB, 0, n0 + cl, s);
        yolo_acc_add(bn.acc, 2, Kout, tile_m % YOLO_ACC_NB, 1, n0 + cl, q);
      }
    }
    if (grouped) {
      __syncthreads();
      rows_fold<NW * 64>(stat_sum, stat_sq, nullptr, 2, (size_t)Kout, bn.rows, bn.group, tile_m, n0 / BN, Kout / BN, n0, BN, tid,
                         reinterpret_cast<int*>(smem));
    }
  }
}

}  // namespace
