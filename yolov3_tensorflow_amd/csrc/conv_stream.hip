// 3x3 / stride-1 / SAME convolution of the 64-channel layers (forward and data gradient): the WEIGHTS stay in registers and the pixels
// stream past them, for gfx950 (MI355X).
//
// Replaces keras.layers.Conv2D (reference backbone/basic_backbone.py:20-43 via resnet18.py:29-32) and its TF autodiff data gradient on the
// 64 -> 64 channel layers of the 104 x 104 maps (416 x 416 input).  With C = 64 the whole K extent is 9 taps x 64 channels = 576: a
// tile kernel (conv3x3_strip_kernel) runs 9 K steps per tile and spends most of a workgroup's life in its prologue, its epilogue and the
// barrier of every K step (measured 420-470 TFLOP/s on these layers against 700-830 on the deeper ones).  Here:
//
// * ONE 512-thread workgroup per CU walks a CONTIGUOUS range of `span` pixels (M / 256 rounded up to 64) in steps of 64 pixels.
// * The 64 x 576 weight tile is read ONCE per workgroup (LDS-DMA into a swizzled image, then 36 ds_read_b128 per lane) and lives in
//   registers for the whole range: wave (nh, mi) holds the 32 output channels of half nh for all 18 k-substeps (144 VGPRs) and computes
//   the 16-pixel fragment mi of every step: 36 MFMAs (16x16x32) against 18 fragment reads, no weight traffic at all in the loop.
// * The pixels live in a ring of 512 LDS rows (64 KiB) addressed by the global pixel index & 511 (the XOR swizzle chunk ^= row & 7 needs
//   no base: pieces are 8-row aligned in pixel space): every step each wave appends ONE 8-row piece by LDS-DMA, two steps ahead of its
//   use, waited on with a counted s_waitcnt vmcnt -- one barrier per step.  SAME padding / row wrap / image boundaries: per-lane 9-bit
//   tap mask, masked taps read a zero row.
// * The output fragment goes bf16 through a double-buffered LDS tile; the NEXT step's first instructions (all 512 threads: 64 pixels x 8
//   chunks) store it in whole 128-byte NHWC rows, accumulate the BatchNorm statistics (forward) or run the fused BatchNorm-backward
//   reduce (data gradient: ReLU mask, fan-in addend, sums of g and g xhat) -- the same arithmetic as conv_common.h tile_epilogue --
//   with their global reads requested one step earlier.  One statistics / partial row per workgroup.
#include "conv_common.h"

namespace {

struct StreamArgs {
  const bf16_t* src; unsigned src_bytes;   // NHWC activations (or dY for the data gradient), C = 64
  const bf16_t* wt;  unsigned wt_bytes;    // [Kout][9][64]
  int H, W, M;                             // M = N*H*W
  int span, ny;                            // pixels per workgroup (multiple of 64), channel tiles (Kout / 64)
  int e0;                                  // 128 + roundup8(W + 1): the ring is filled up to pixel P0 + e0 before the first step
  float rhw, rw;
};

constexpr int ST_RING = 512;                         // ring rows
constexpr int ST_ZERO = ST_RING * 128;               // 128 zero bytes
constexpr int ST_WT = ST_ZERO + 128;                 // weight image: 9 taps x [64 channels][64 k] swizzled (72 KiB), dead after the prologue
constexpr int ST_OLD = 144;                          // staged output row stride (128 + 16: conflict-free 16-byte writes)
constexpr int ST_STAGE = ST_WT;                      // 2 staged tiles of 64 x 144 bytes (alias the weight image)
constexpr int ST_RED = ST_WT + 2 * 64 * ST_OLD;      // final reduction: 3 x [8][64] floats
constexpr int ST_LDS = ST_WT + 9 * 8192;

typedef unsigned st_v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x8_t st_lds_frag_t;

// EPI: 0 = plain (forward: statistics of the stored values; plain data gradient), 1 = fused BatchNorm-backward reduce, 2 = the same with a
// second (shortcut) BatchNorm.  ACC: the output is added to `addend` (or to the output buffer itself).  The hot loop is ONE basic block:
// every out-of-range access of the store side goes through buffer descriptors (dropped stores, zero loads) instead of branches.
template <int EPI, bool ACC>
__global__ __launch_bounds__(512) void conv3x3_stream_kernel(StreamArgs a, void* __restrict__ Yv, int ldy, float* __restrict__ stat_sum,
                                                             float* __restrict__ stat_sq, int Kout, BnEpi bn) {
  constexpr bool BNEPI = EPI != 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nh = wave >> 2, mi = wave & 3;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = tile % a.ny, wg_m = tile / a.ny;
  const int n0 = tile_n * 64;
  const int P0 = wg_m * a.span;
  const int P1 = min(P0 + a.span, a.M);
  const int niter = (P1 - P0 + 63) >> 6;
  if (tid < 8) *reinterpret_cast<uint4*>(smem + ST_ZERO + tid * 16) = make_uint4(0u, 0u, 0u, 0u);

  // ---- prologue: ring pieces [P0 - (W+1), P0 + e0) and the weight image -------------------------------------------------------
  const int lrow = lane >> 3;
  const int cchunk = (lane & 7) ^ lrow;
  const int lane_src = lrow * 128 + cchunk * 16;                  // byte offset of this lane inside an 8-pixel piece of the source
  {
    const int j_first = (P0 - a.W - 1 + 1024) / 8 - 128;          // floor((P0 - W - 1) / 8)
    const int j_end = (P0 + a.e0) >> 3;
    for (int j = j_first + wave; j < j_end; j += 8)
      buffer_load_lds16(a.src, a.src_bytes, smem + ((j * 8) & (ST_RING - 1)) * 128, (unsigned)(j * 1024 + lane_src));
    const unsigned wrow = (unsigned)(((n0 + wave * 8 + lrow) * 576 + cchunk * 8) * 2);
#pragma unroll
    for (int t = 0; t < 9; ++t) buffer_load_lds16(a.wt, a.wt_bytes, smem + ST_WT + t * 8192 + wave * 1024, wrow + t * 128);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int kq = lane >> 4;
  bf16x8_t wreg[2][18];
  {
    const int r = lane & 15;
#pragma unroll
    for (int cf = 0; cf < 2; ++cf) {
      const int chl = nh * 32 + 8 * (r >> 2) + 4 * cf + (r & 3);   // MFMA row r of chain cf: so that a lane ends up with 8 consecutive channels
#pragma unroll
      for (int ks = 0; ks < 18; ++ks)
        wreg[cf][ks] = *reinterpret_cast<const bf16x8_t*>(smem + ST_WT + (ks >> 1) * 8192 + swz(chl, (ks & 1) * 4 + kq));
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();                                                // the weight image is dead: its space now stages the output tiles

  // ---- per-lane state of the compute side: pixel P0 + 64 it + 16 mi + (lane & 15) ------------------------------------------------
  int px = P0 + 16 * mi + (lane & 15);
  int py_, px_;                                                   // (row, column) of that pixel inside its image
  {
    int n_, rem;
    fast_divmod(px, a.H * a.W, a.rhw, n_, rem);
    fast_divmod(rem, a.W, a.rw, py_, px_);
  }
  const int zero_addr = ST_ZERO + kq * 16;
  const int stage_wr = ST_STAGE + (16 * mi + (lane & 15)) * ST_OLD + (nh * 32 + 8 * kq) * 2;
  int tad[9];                                                     // unmasked LDS address of this lane's fragment row of tap t (k chunk kq)
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int row = (px + (t / 3) * a.W + (t % 3) - a.W - 1) & (ST_RING - 1);
    tad[t] = row * 128 + (((row ^ kq) & 7) << 4);
  }

  // ---- per-thread state of the store side: pixel slot tid >> 3, channel chunk tid & 7 --------------------------------------------
  const int spx = tid >> 3, sch = tid & 7, sc = n0 + sch * 8;
  const int stage_rd = ST_STAGE + spx * ST_OLD + sch * 16;
  const bool fwd_acc = !BNEPI && !stat_sum && bn.acc;
  float s0[8], s1[8], s2[8], mu[8], rs[8], mu2[8], rs2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s0[j] = s1[j] = s2[j] = 0.f; mu[j] = rs[j] = mu2[j] = rs2[j] = 0.f; }
  if constexpr (BNEPI) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { mu[j] = bn.mean[sc + j]; rs[j] = bn.rstd[sc + j]; }
  }
  if constexpr (EPI == 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { mu2[j] = bn.mean2[sc + j]; rs2[j] = bn.rstd2[sc + j]; }
  }
  const unsigned ybytes = (unsigned)a.M * (unsigned)ldy * 2u;     // (< 2^31: checked by the host)
  const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(Yv, 0, (int)ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rAdd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(bn.addend ? bn.addend : reinterpret_cast<const bf16_t*>(Yv)), 0, (int)ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBy = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(bn.y), 0, BNEPI ? (int)ybytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rBy2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(bn.y2), 0, EPI == 2 ? (int)ybytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rMk = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(bn.mask), 0, (BNEPI && bn.mask) ? (int)(ybytes >> 4) : 0, 0x00020000);
  const unsigned nomask = (BNEPI && bn.mask) ? 0u : 0xffu;
  st_v4u e_yv = {0u, 0u, 0u, 0u}, e_ev = e_yv, e_y2 = e_yv;
  unsigned e_mk = 0xffu, e_off = 0xfffffff0u;                     // byte offset of this thread's 16 bytes of the block (out of range: none)
  // global reads of the store side for block `it` (requested one step before they are used); out-of-range offsets read zeros
  auto epi_load = [&](int it) {
    const int m = P0 + 64 * it + spx;
    e_off = m < P1 ? ((unsigned)m * (unsigned)ldy + (unsigned)sc) * 2u : 0xfffffff0u;
    if constexpr (BNEPI) {
      e_yv = __builtin_amdgcn_raw_buffer_load_b128(rBy, (int)e_off, 0, 0);
      e_mk = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rMk, (int)(e_off >> 4), 0, 0) | nomask;
    }
    if constexpr (EPI == 2) e_y2 = __builtin_amdgcn_raw_buffer_load_b128(rBy2, (int)e_off, 0, 0);
    if constexpr (ACC) e_ev = __builtin_amdgcn_raw_buffer_load_b128(rAdd, (int)e_off, 0, 0);
  };
  // store side of one block (staged by the previous step): the arithmetic of conv_common.h tile_epilogue, branch-free
  auto store_block = [&](st_v4u sv) {
    uint4 v = make_uint4(sv.x, sv.y, sv.z, sv.w);
    float g8[8], y8[8];
    if constexpr (ACC) {                              // gradient fan-in: float32 add, one rounding
      unpack_bf8(v, g8);
      unpack_bf8(make_uint4(e_ev.x, e_ev.y, e_ev.z, e_ev.w), y8);
#pragma unroll
      for (int j = 0; j < 8; ++j) g8[j] += y8[j];
      v = pack_bf8(g8);
    }
    if constexpr (BNEPI) {
      unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q)
        w4[q] = (((e_mk >> (2 * q)) & 1u) ? (w4[q] & 0xffffu) : 0u) | (((e_mk >> (2 * q + 1)) & 1u) ? (w4[q] & 0xffff0000u) : 0u);
      v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    }
    const st_v4u ov = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(ov, rY, (int)e_off, 0, 0);      // (out-of-range offsets are dropped)
    const bool live = e_off != 0xfffffff0u;
    unpack_bf8(v, g8);
#pragma unroll
    for (int j = 0; j < 8; ++j) g8[j] = live ? g8[j] : 0.f;
    if constexpr (BNEPI) {
      unpack_bf8(make_uint4(e_yv.x, e_yv.y, e_yv.z, e_yv.w), y8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s0[j] += g8[j]; s1[j] += g8[j] * ((y8[j] - mu[j]) * rs[j]); }
      if constexpr (EPI == 2) {
        unpack_bf8(make_uint4(e_y2.x, e_y2.y, e_y2.z, e_y2.w), y8);
#pragma unroll
        for (int j = 0; j < 8; ++j) s2[j] += g8[j] * ((y8[j] - mu2[j]) * rs2[j]);
      }
    } else if constexpr (!ACC) {                      // statistics of the values as stored (bf16-rounded)
#pragma unroll
      for (int j = 0; j < 8; ++j) { s0[j] += g8[j]; s1[j] += g8[j] * g8[j]; }
    }
  };

  const int npiece0 = ((P0 + a.e0) >> 3) + wave;                  // this wave's piece of step 0
  // one step: the store side of the previous block, the ring piece of step it + 2, 9 taps x (2 fragment reads, 4 MFMAs)
  auto step = [&](int it, bool compute) {
    // the pieces of steps <= it - 2 and the store side's reads of block it - 1 have landed (outstanding: the piece of step it - 1);
    // every wave has finished step it - 1 (its ring reads, its staged fragment)
    asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const st_v4u staged = *reinterpret_cast<const st_v4u*>(smem + stage_rd + ((it + 1) & 1) * (64 * ST_OLD));     // block it - 1 (it = 0: unused)
    if (!compute) { store_block(staged); return; }
    // tap addresses: masked taps (SAME padding, row wrap, image boundary, pixels beyond M) read the zero row
    const unsigned cb = (px_ > 0 ? 1u : 0u) | 2u | (px_ < a.W - 1 ? 4u : 0u);
    unsigned ok = (py_ > 0 ? cb : 0u) | (cb << 3) | (py_ < a.H - 1 ? (cb << 6) : 0u);
    ok = px < a.M ? ok : 0u;
    int addr[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int m = __builtin_amdgcn_sbfe((int)ok, t, 1);
      addr[t] = (tad[t] & m) | (zero_addr & ~m);
      tad[t] = (tad[t] + 64 * 128) & (ST_RING * 128 - 1);          // the next step's pixel: 64 ring rows on (64 % 8 == 0: same swizzle)
    }
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    {
      const int j = npiece0 + 8 * it;
      buffer_load_lds16(a.src, a.src_bytes, smem + ((j * 8) & (ST_RING - 1)) * 128, (unsigned)(j * 1024 + lane_src));
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const bf16x8_t p0 = *reinterpret_cast<const st_lds_frag_t*>(addr[t]);            // (dynamic LDS starts at address 0: no base add)
      const bf16x8_t p1 = *reinterpret_cast<const st_lds_frag_t*>(addr[t] ^ 64);
      acc0 = YOLO_MFMA_16x16x32(wreg[0][2 * t], p0, acc0);
      acc1 = YOLO_MFMA_16x16x32(wreg[1][2 * t], p0, acc1);
      acc0 = YOLO_MFMA_16x16x32(wreg[0][2 * t + 1], p1, acc0);
      acc1 = YOLO_MFMA_16x16x32(wreg[1][2 * t + 1], p1, acc1);
      if (t == 1) { store_block(staged); epi_load(it); }           // (their VALU work goes under the MFMAs of the taps around them)
    }
    st_v4u o;
    o.x = pack_bf2(acc0[0], acc0[1]); o.y = pack_bf2(acc0[2], acc0[3]);
    o.z = pack_bf2(acc1[0], acc1[1]); o.w = pack_bf2(acc1[2], acc1[3]);
    *reinterpret_cast<st_v4u*>(smem + stage_wr + (it & 1) * (64 * ST_OLD)) = o;
    px += 64;
    px_ += 64;
    if (px_ >= a.W) { px_ -= a.W; py_ = py_ + 1 == a.H ? 0 : py_ + 1; }     // (W >= 64: one wrap at most)
  };
  for (int it = 0; it < niter; ++it) step(it, true);               // (step 0's store side sees e_off out of range: nothing is stored or summed)
  step(niter, false);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the last two pieces are never read; nothing may be in flight into LDS at the end)

  // ---- one statistics / partial row per workgroup: threads -> 8-slot groups, then the 8 groups ------------------------------------
  const int nq = BNEPI ? (EPI == 2 ? 3 : 2) : ((!ACC && (stat_sum || fwd_acc)) ? 2 : 0);
  if (nq == 0) return;
  float* const red = reinterpret_cast<float*>(smem + ST_RED);       // [3][8][64]
#pragma unroll
  for (int j = 0; j < 8; ++j) {                                      // lanes of a wave that share (lane & 7): 8 pixel slots
    float v0 = s0[j], v1 = s1[j], v2 = s2[j];
#pragma unroll
    for (int d = 8; d < 64; d <<= 1) {
      v0 += __shfl_xor(v0, d);
      v1 += __shfl_xor(v1, d);
      if constexpr (EPI == 2) v2 += __shfl_xor(v2, d);
    }
    if (lane < 8) {
      red[(0 * 8 + wave) * 64 + lane * 8 + j] = v0;
      red[(1 * 8 + wave) * 64 + lane * 8 + j] = v1;
      if constexpr (EPI == 2) red[(2 * 8 + wave) * 64 + lane * 8 + j] = v2;
    }
  }
  __syncthreads();
  if (tid < nq * 64) {
    const int q = tid >> 6, cl = tid & 63;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t += red[(q * 8 + w) * 64 + cl];
    if constexpr (BNEPI) {
      if (bn.partial) bn.partial[((size_t)wg_m * 3 + q) * ldy + n0 + cl] = t;
      else yolo_acc_add(bn.acc, 3, ldy, wg_m % YOLO_ACC_NB, q, n0 + cl, t);
    } else {
      if (stat_sum) (q == 0 ? stat_sum : stat_sq)[(size_t)wg_m * Kout + n0 + cl] = t;
      else yolo_acc_add(bn.acc, 2, Kout, wg_m % YOLO_ACC_NB, q, n0 + cl, t);
    }
  }
}

bool st_eligible(const yoloconv::Gather& g, int Kout, bool f32) {
  if (f32 || g.den != 1 || g.C0 != 0 || g.S != 3 || g.RS != 9 || g.smul != 1 || g.pad_h != 1 || g.pad_w != 1 || g.s2) return false;
  if (g.Hs != g.Ho || g.Ws != g.Wo || g.C1 != 64 || Kout % 64 != 0) return false;
  if (g.Wo < 64 || ((g.Wo + 1 + 7) / 8 * 8) + g.Wo + 1 > 320) return false;           // one column wrap per step; the ring's reach
  if ((size_t)g.M * 128 >= (1ull << 31) || (size_t)Kout * 576 * 2 >= (1ull << 31) || (size_t)g.M * Kout * 2 >= (1ull << 31)) return false;
  return true;
}

}  // namespace

int g_stream = -1;        // "stream" tuning: -1 auto, 0 never, 1 wherever it fits

// 0 = the streaming kernel does not take this problem, else the pixels per workgroup (statistics / partial rows = ceil(M / that))
int yolo_stream_plan(const yoloconv::Gather& g, int Kout, bool f32, StreamPlanOut* out) {
  if (g_stream == 0 || !st_eligible(g, Kout, f32)) return 0;
  const int ny = Kout / 64;
  const int nxt = ny >= 256 ? 1 : 256 / ny;
  int span = ((g.M + nxt - 1) / nxt + 63) / 64 * 64;
  if (g_stream < 0 && span < 512) return 0;            // auto: at least 8 steps behind one weight load
  if (span < 64) span = 64;
  if (out) { out->span = span; out->nx = (g.M + span - 1) / span; out->ny = ny; out->lds = ST_LDS; }
  return span;
}

namespace {
template <int EPI, bool ACC>
int st_launch_e(const yoloconv::Gather& g, const StreamPlanOut& pl, const void* w, void* y, int ldy, const yoloconv::Epi& e, int Kout, hipStream_t st) {
  StreamArgs a;
  a.src = g.src1;
  a.src_bytes = (unsigned)((size_t)g.M * 128);
  a.wt = (const bf16_t*)w;
  a.wt_bytes = (unsigned)((size_t)Kout * 576 * 2);
  a.H = g.Ho; a.W = g.Wo; a.M = g.M;
  a.span = pl.span; a.ny = pl.ny;
  a.e0 = 128 + (g.Wo + 1 + 7) / 8 * 8;
  a.rhw = g.rhw; a.rw = g.rw;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_stream_kernel<EPI, ACC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err != hipSuccess) { yolo_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(err)); return (int)err; }
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3x3_stream_kernel<EPI, ACC>), dim3(pl.nx * pl.ny), dim3(512), pl.lds, st, a, y, ldy, e.ssum, e.ssq, Kout, e.bn);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}
template <int EPI>
int st_launch_a(const yoloconv::Gather& g, const StreamPlanOut& pl, const void* w, void* y, int ldy, int accumulate, const yoloconv::Epi& e, int Kout, hipStream_t st) {
  if (accumulate) return st_launch_e<EPI, true>(g, pl, w, y, ldy, e, Kout, st);
  return st_launch_e<EPI, false>(g, pl, w, y, ldy, e, Kout, st);
}
}  // namespace

int yolo_stream_launch(const yoloconv::Gather& g, const void* w, void* y, int ldy, int accumulate, const yoloconv::Epi& e, int Kout, hipStream_t st) {
  StreamPlanOut pl;
  if (!yolo_stream_plan(g, Kout, false, &pl) || ldy != Kout) { yolo_set_error("%s:%d: no streaming plan", __FILE__, __LINE__); return YOLO_ERR_INVALID_ARG; }
  if (e.bn.y && e.bn.y2) return st_launch_a<2>(g, pl, w, y, ldy, accumulate, e, Kout, st);
  if (e.bn.y) return st_launch_a<1>(g, pl, w, y, ldy, accumulate, e, Kout, st);
  return st_launch_a<0>(g, pl, w, y, ldy, accumulate, e, Kout, st);
}
