// Implicit-GEMM convolution for gfx950 (MI355X): forward, data-gradient and weight-gradient, NHWC bf16, fp32 accumulate.
//
// Replaces keras.layers.Conv2D (reference backbone/basic_backbone.py:42, yolov3/yolov3_detector.py:98-150) and its TF
// autodiff gradients.  One gather routine feeds all three passes: a "row" is a pixel of the row space (output pixels for
// fwd, input pixels for dgrad, output pixels for wgrad) and a "k-chunk" is 8 consecutive channels (16 bytes) of one
// kernel tap of the source tensor; the (optional) nearest 2x upsample + channel concat of the FPN necks
// (yolov3_detector.py:115-116,140-141) is resolved inside the gather, so the concatenated tensor never exists.
//
// igemm_fwd_kernel     : {128,64} pixels x {128,64} channels per workgroup (8 waves on 128 x 128, else 4), BK = 64; both operands arrive by
//                        LDS-DMA (global_load_lds_dwordx4 / buffer_load ... lds) into a 2-stage ring with the XOR swizzle applied on the
//                        SOURCE address (the DMA writes lane-linear), one raw s_barrier + counted s_waitcnt per K-step,
//                        v_mfma_f32_16x16x32 with the WEIGHT tile as the A operand so that a lane holds 4 consecutive channels of a pixel.
//                        Stride-2 data gradients run as four dense parity classes in one launch.
// conv3x3_strip_kernel : 3x3 / stride-1 layers: the pixel strip of a tile is loaded once per 64-channel slice, the nine taps are nine
//                        shifted reads of that LDS image; only the weight tile streams through a ring.
// epilogue (shared, conv_common.h): bf16 tile transposed through LDS -> whole NHWC row stores (also the fan-in accumulate path),
//                        BatchNorm partial statistics (one row per pixel tile), or the BatchNorm-backward reduce of a data gradient;
//                        float32 logits + bias are written directly.
// wgrad kernels        : dW[co][kcol] = sum_pixels dY[pix][co] * X[pix][kcol]; both operands are pixel-major in memory, so fragments are
//                        read with ds_read_b64_tr_b16 (hardware transpose); split over pixel ranges into per-layer SLABS (plain stores,
//                        no atomics) that one launch per gradient bucket sums (wgrad_reduce_batched_kernel).
#include "conv_common.h"

namespace {


// Forward / data-gradient implicit GEMM.  BM pixels x BN channels per 256-thread workgroup; a 3-stage LDS ring filled by LDS-DMA
// (global_load_lds_dwordx4: 16 B per lane, no VGPR staging), counted vmcnt so that one stage stays in flight across the single raw
// s_barrier of each K-step; the XOR swizzle of the LDS image is applied on the SOURCE address (LDS-DMA writes lane-linear).
// FAST gather (den == 1, single source, <= 32 taps): address = row base + tap offset, validity = one bit of a per-row tap mask, both
// computed once per tile; the generic path (stride-2 data gradient, fused upsample+concat) recomputes coordinates per K-step.
template <int BM, int BN, int NSTAGE, bool OUT_F32, bool FAST, int NW, bool BNEPI>
__global__ __launch_bounds__(NW * 64) void igemm_fwd_kernel(Gather g, const bf16_t* __restrict__ Wt, const float* __restrict__ bias,
                                                        void* __restrict__ Yv, int ldy, int accumulate,
                                                        float* __restrict__ stat_sum, float* __restrict__ stat_sq,
                                                        int Kout, int tiles_n, BnEpi bnepi) {
  constexpr int WN = (BN == 128) ? 2 : 1;  // waves along channels
  constexpr int WM = NW / WN;              // waves along pixels
  constexpr int PT = BM / WM / 16;         // 16-pixel MFMA tiles per wave
  constexpr int CT = BN / WN / 16;         // 16-channel MFMA tiles per wave (= 4)
  constexpr int A_BYTES = BM * BK * 2;
  constexpr int B_BYTES = BN * BK * 2;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int A_INSTR = BM / (8 * NW);   // LDS-DMA instructions per wave per stage (8 rows x 128 B each)
  constexpr int B_INSTR = BN / (8 * NW);
  extern __shared__ __attribute__((aligned(16))) char smem[];

  ClassView cv = {};
  int wKg = g.Kg;
  if (g.s2) {   // uniform: specialise the gather for this workgroup's parity class
    const int ph = blockIdx.y >> 1, pw = blockIdx.y & 1;
    Gather::Dim rd, cd;                        // field-wise selects: no dynamic indexing, no aggregate select (both end up in scratch)
    rd.n = ph ? g.rowd[1].n : g.rowd[0].n;       cd.n = pw ? g.cold[1].n : g.cold[0].n;
    rd.pad = ph ? g.rowd[1].pad : g.rowd[0].pad; cd.pad = pw ? g.cold[1].pad : g.cold[0].pad;
    rd.size = ph ? g.rowd[1].size : g.rowd[0].size; cd.size = pw ? g.cold[1].size : g.cold[0].size;
    rd.t0 = ph ? g.rowd[1].t0 : g.rowd[0].t0;    cd.t0 = pw ? g.cold[1].t0 : g.cold[0].t0;
    rd.t1 = ph ? g.rowd[1].t1 : g.rowd[0].t1;    cd.t1 = pw ? g.cold[1].t1 : g.cold[0].t1;
    wKg = g.wKg;
    g.S = cd.n; g.RS = rd.n * cd.n; g.pad_h = rd.pad; g.pad_w = cd.pad; g.Ho = rd.size; g.Wo = cd.size;
    g.M = g.N * rd.size * cd.size; g.Kg = g.RS * g.C1;
    g.rhw = 1.0f / (float)(rd.size * cd.size); g.rw = 1.0f / (float)cd.size; g.magicS = 65536 / cd.n + 1;
    cv.on = 1; cv.Hc = rd.size; cv.Wc = cd.size; cv.OH = g.OH; cv.OW = g.OW; cv.ph = ph; cv.pw = pw; cv.rhw = g.rhw; cv.rw = g.rw;
    // class tap t = r' * S' + s' -> tap of the full (flipped) weight tensor = wbase + r' * wdr + s' * wds (arithmetic, not a table: a
    // 4-entry select chain is turned into a scratch lookup table by the compiler)
    cv.two = cd.n == 2;
    cv.wbase = rd.t0 * g.S_full + cd.t0;
    cv.wdr = (rd.t1 - rd.t0) * g.S_full;
    cv.wds = cd.t1 - cd.t0;
  }

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = tile % tiles_n, tile_m = tile / tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (m0 >= g.M) return;                     // a smaller parity class has fewer pixel tiles than the grid (whole workgroup, before any barrier)

  // LDS-DMA lane geometry: lane -> row (lane >> 3) of the instruction's 8 rows, LDS slot (lane & 7); the slot holds global chunk
  // slot ^ (row & 7) (row & 7 == lane >> 3 because instruction row blocks are 8-aligned)
  const int lrow = lane >> 3;
  const int cchunk = (lane & 7) ^ lrow;
  RowInfo rows[A_INSTR];
  int rbase[A_INSTR];
  unsigned vmask[A_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    rows[j] = decode_row(g, m0 + (wave * A_INSTR + j) * 8 + lrow);
    if constexpr (FAST) {
      rbase[j] = ((rows[j].n * g.Hs + rows[j].hb) * g.Ws + rows[j].wb) * g.C1;
      // tap (tr, ts) is inside the image iff row tr and column ts are: 3 + 3 range tests per row instead of 9 x 4; bits 0..15 = rows,
      // bits 16..31 = columns (R, S <= 9)
      unsigned mk = 0;
      for (int t = 0; t * g.S < g.RS; ++t) {
        const int hn = rows[j].hb + t;
        mk |= ((hn >= 0) & (hn < g.Hs)) ? (1u << t) : 0u;
      }
      for (int t = 0; t < g.S; ++t) {
        const int wn = rows[j].wb + t;
        mk |= ((wn >= 0) & (wn < g.Ws)) ? (0x10000u << t) : 0u;
      }
      vmask[j] = mk;
    }
  }
  const bf16_t* wrow[B_INSTR];
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) wrow[j] = Wt + (size_t)(n0 + (wave * B_INSTR + j) * 8 + lrow) * wKg + cchunk * 8;

  f32x4_t acc[CT][PT];
#pragma unroll
  for (int a = 0; a < CT; ++a)
#pragma unroll
    for (int b = 0; b < PT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = (g.Kg + BK - 1) / BK;
  const int cmask = (1 << g.lgC8) - 1;

  auto issue_stage = [&](int kt, int buf) {
    char* sA = smem + buf * STAGE + wave * (A_INSTR * 1024);
    char* sB = smem + buf * STAGE + A_BYTES + wave * (B_INSTR * 1024);
    const int q = kt * (BK / 8) + cchunk;
    const int tap = q >> g.lgC8;
    const int c = (q & cmask) << 3;
    const bool kvalid = tap < g.RS;
    const int tr = (tap * g.magicS) >> 16, ts = tap - tr * g.S;
    if constexpr (FAST) {
      const int toff = (tr * g.Ws + ts) * g.C1 + c;
      const unsigned tbits = kvalid ? ((1u << tr) | (0x10000u << ts)) : 0xffffffffu;   // ~0 never matches: both bits must be set
#pragma unroll
      for (int j = 0; j < A_INSTR; ++j) {
        const bf16_t* p = (kvalid && (vmask[j] & tbits) == tbits) ? g.src1 + (rbase[j] + toff) : reinterpret_cast<const bf16_t*>(&g_zero16);
        __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(sA + j * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < A_INSTR; ++j)
        __builtin_amdgcn_global_load_lds((gptr_t)gather_addr(g, rows[j], tr, ts, c, kvalid), (lptr_t)(sA + j * 1024), 16, 0, 0);
    }
    const bool kv = kt * BK + cchunk * 8 < g.Kg;
    int wk = kt * BK;
    if (cv.on) {                              // class tap -> tap of the full weight tensor (a K-step never straddles taps: C1 % 64 == 0)
      const int lgC = g.lgC8 + 3, t = wk >> lgC;
      const int r1 = cv.two ? t >> 1 : t, s1 = cv.two ? t & 1 : 0;
      const int wt = cv.wbase + r1 * cv.wdr + s1 * cv.wds;
      wk = wt * g.C1 + (wk - (t << lgC));
    }
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)(kv ? wrow[j] + wk : reinterpret_cast<const bf16_t*>(&g_zero16)), (lptr_t)(sB + j * 1024), 16, 0, 0);
  };
  auto compute_stage = [&](int buf) {
    const char* sA = smem + buf * STAGE;
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
        for (int b = 0; b < PT; ++b) acc[a][b] = YOLO_MFMA_16x16x32(wf[a], pf[b], acc[a][b]);
    }
  };

  constexpr int LPS = A_INSTR + B_INSTR;   // LDS-DMA loads per stage per wave
#pragma unroll
  for (int st = 0; st < NSTAGE - 1; ++st)
    if (st < nk) issue_stage(st, st);
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt has landed once at most the loads of stages kt+1 .. kt+NSTAGE-2 are outstanding (vmcnt counts in issue order);
    // in the last NSTAGE-2 iterations fewer stages are in flight, so drain completely there
    // lgkmcnt(0): this wave's LDS reads of stage kt-1 must have COMPLETED before the barrier lets other waves' LDS-DMA refill that
    // buffer -- the compiler happily sinks the lgkmcnt wait (and the last MFMAs of the previous stage) below a raw s_barrier, and a
    // DMA write does not queue behind another wave's pending ds_read (seen as a rare corrupted half tile)
    if (kt + NSTAGE - 2 < nk) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NSTAGE - 2) * LPS) : "memory");
    else                      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // every wave's part of stage kt is visible; stage kt-1 is no longer read
    asm volatile("" ::: "memory");
    if (kt + NSTAGE - 1 < nk) issue_stage(kt + NSTAGE - 1, (kt + NSTAGE - 1) % NSTAGE);   // refills the buffer of stage kt-1
    compute_stage(kt % NSTAGE);
  }

  // (partial row of a fused BatchNorm reduce: one per pixel tile and parity class; classes smaller than the grid leave theirs untouched = 0)
  // accumulate == 2: only the even / even parity class has a previous contribution (a 1x1 stride-2 gradient written at those positions only)
  const int acc_eff = accumulate == 2 ? ((cv.on && cv.ph == 0 && cv.pw == 0) ? 1 : 0) : accumulate;
  tile_epilogue<BM, BN, NW, WM, WN, PT, CT, OUT_F32, BNEPI>(acc, smem, g.M, m0, n0, tile_m, bias, Yv, ldy, acc_eff, stat_sum, stat_sq, Kout, tid, lane, wm, wn,
                                                     cv, bnepi, (int)(blockIdx.y * (gridDim.x / tiles_n)) + tile_m);
}

// ------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / SAME convolution with the input strip resident in LDS
// ------------------------------------------------------------------------------------------------------------------
// The implicit-GEMM kernel above re-gathers the pixel operand for each of the 9 taps, and on the large feature maps it runs at the
// L2 -> LDS gather rate (~14 TB/s over all CUs at 64 FLOP/B for 128 x 128 tiles), not at the MFMA rate.  Here a workgroup owns BM
// consecutive pixels (linear NHW index) x BN output channels and, per 64-channel slice, loads the strip of pixels
// [m0 - (W+1), m0 + BM + (W+1)) ONCE; tap (tr, ts) of pixel p is strip row p + tr*W + ts, so the 9 taps are 9 shifted reads of the same
// LDS image (the XOR swizzle chunk ^= row & 7 is conflict-free for ds_read_b128 at every row alignment).  Taps that fall off the image
// (SAME padding, row wrap, image boundary inside the strip) are redirected lane by lane to a zero row.  Only the weight tile
// (BN x 64 per tap and slice) streams through a 2-stage ring.  L2 -> LDS bytes per tile drop 2.6-3.3x for 256-pixel tiles.
struct StripArgs {
  const bf16_t* src; unsigned src_bytes;   // NHWC activations (or dY for the stride-1 data gradient)
  const bf16_t* wt;  unsigned wt_bytes;    // [Kout][9][C]
  int H, W, C, M, Kg;                      // M = N*H*W, Kg = 9*C
  int E8;                                  // strip rows, multiple of 8 (>= BM + 2W + 2)
  float rhw, rw;
  int xsplit;                              // 0: tiles dealt to the XCDs in contiguous runs; G = 2 / 4: XCD x takes channel-tile group x % G of pixel part x / G
};

template <int BM, int BN, int NW, int WS, bool BNEPI>
__global__ __launch_bounds__(NW * 64) void conv3x3_strip_kernel(StripArgs a, const float* __restrict__ bias, void* __restrict__ Yv, int ldy,
                                                            int accumulate, float* __restrict__ stat_sum, float* __restrict__ stat_sq,
                                                            int Kout, int tiles_n, BnEpi bnepi) {
  constexpr int WN = (BN == 128 || (BM == 64 && BN == 64)) ? 2 : 1;   // (64 x 64: 2 x 2 waves of 32 x 32 -- 4 fragment reads per 4 MFMAs instead of 5)
  constexpr int WM = NW / WN;
  constexpr int PT = BM / WM / 16;
  constexpr int CT = BN / WN / 16;
  constexpr int B_INSTR = BN / (8 * NW);   // weight LDS-DMA instructions per wave per k-step (8 rows x 128 B each)
  constexpr int W_STAGE = BN * 128;
  static_assert(WS == 2 || WS == 3, "weight ring depth");
  static_assert(B_INSTR >= 1, "tile too small for the wave count");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sStrip = smem;
  char* const sZero = smem + a.E8 * 128;
  char* const sRing = sZero + 128;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int tile_n, tile_m;
  if (a.xsplit) {
    // XCD-aware rectangle: with contiguous runs every XCD's L2 streams ALL weights and 1 / 8 of the pixels; here XCD x (= blockIdx & 7)
    // owns the channel tiles of group x % G and the pixel tiles of part x / G: 1 / G of the weights, G / 8 of the pixels per L2.  The grid is
    // padded to the largest part (workgroups past their part leave before any barrier).
    const int G = a.xsplit, parts = 8 / G, x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int tiles_m = (a.M + BM - 1) / BM, nn = tiles_n / G;
    const int mp = x / G, m_lo = mp * tiles_m / parts, m_hi = (mp + 1) * tiles_m / parts;
    tile_n = (x % G) * nn + j % nn;
    tile_m = m_lo + j / nn;
    if (tile_m >= m_hi) return;
  } else {
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    tile_n = tile % tiles_n; tile_m = tile / tiles_n;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (tid < 8) *reinterpret_cast<uint4*>(sZero + tid * 16) = make_uint4(0u, 0u, 0u, 0u);

  // LDS-DMA lane geometry (both images): lane -> row (lane >> 3) of the instruction's 8 rows, slot lane & 7 holds chunk slot ^ row
  const int lrow = lane >> 3;
  const int cchunk = (lane & 7) ^ lrow;
  // strip: instruction i covers strip rows 8i .. 8i+7 = pixels m0 - (W+1) + 8i + lrow; out-of-tensor pixels are out of the buffer
  // range (negative offsets wrap above 2^31) and arrive as zeros
  const int strip_off0 = ((m0 - (a.W + 1) + wave * 8 + lrow) * a.C + cchunk * 8) * 2;
  const int strip_step = NW * 8 * a.C * 2;
  const int n_strip_instr = a.E8 >> 3;
  unsigned wbase[B_INSTR];
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) wbase[j] = (unsigned)(((n0 + (wave * B_INSTR + j) * 8 + lrow) * a.Kg + cchunk * 8) * 2);

  // per pixel tile of this wave: tile-local pixel index and the validity of its 3 tap rows / 3 tap columns
  int prow[PT];
  unsigned pmask[PT];
#pragma unroll
  for (int b = 0; b < PT; ++b) {
    prow[b] = wm * (PT * 16) + b * 16 + (lane & 15);
    const int m = m0 + prow[b];
    unsigned mk = 0;
    if (m < a.M) {
      int n, rem, y, x;
      fast_divmod(m, a.H * a.W, a.rhw, n, rem);
      fast_divmod(rem, a.W, a.rw, y, x);
      mk = (y > 0 ? 1u : 0u) | 2u | (y < a.H - 1 ? 4u : 0u) | (x > 0 ? 8u : 0u) | 16u | (x < a.W - 1 ? 32u : 0u);
    }
    pmask[b] = mk;
  }
  const int kq = lane >> 4;                 // k group of the MFMA operand layout: channels 8*kq .. +7 of a 32-channel substep
  const int zero_addr = a.E8 * 128 + kq * 16;

  f32x4_t acc[CT][PT];
#pragma unroll
  for (int a_ = 0; a_ < CT; ++a_)
#pragma unroll
    for (int b = 0; b < PT; ++b) acc[a_][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nchunk = a.C >> 6;
  const int nk = nchunk * 9;

  auto issue_weights = [&](int cc, int tap, int stage) {
    char* sB = sRing + stage * W_STAGE + wave * (B_INSTR * 1024);
    const unsigned koff = (unsigned)((tap * a.C + cc * 64) * 2);
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) buffer_load_lds16(a.wt, a.wt_bytes, sB + j * 1024, wbase[j] + koff);
  };
  auto issue_strip = [&](int cc) {
    int off = strip_off0 + cc * 128;
    for (int i = wave; i < n_strip_instr; i += NW) {
      buffer_load_lds16(a.src, a.src_bytes, sStrip + i * 1024, (unsigned)off);
      off += strip_step;
    }
  };
  auto compute = [&](int tr, int ts, int stage) {
    const char* sB = sRing + stage * W_STAGE;
    const int toff = tr * a.W + ts;
    int baddr[PT];
#pragma unroll
    for (int b = 0; b < PT; ++b) {
      const int row = prow[b] + toff;
      const bool ok = ((pmask[b] >> tr) & (pmask[b] >> (3 + ts)) & 1u) != 0u;
      baddr[b] = ok ? row * 128 + ((kq ^ (row & 7)) << 4) : zero_addr;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t wf[CT], pf[PT];
      const int ch = ks * 4 + kq;
#pragma unroll
      for (int a_ = 0; a_ < CT; ++a_) wf[a_] = *reinterpret_cast<const bf16x8_t*>(sB + swz(wn * (CT * 16) + a_ * 16 + (lane & 15), ch));
#pragma unroll
      for (int b = 0; b < PT; ++b) pf[b] = *reinterpret_cast<const bf16x8_t*>(smem + (baddr[b] ^ (ks << 6)));   // chunk ^ 4 (zero row: still zeros)
#pragma unroll
      for (int a_ = 0; a_ < CT; ++a_)
#pragma unroll
        for (int b = 0; b < PT; ++b) acc[a_][b] = YOLO_MFMA_16x16x32(wf[a_], pf[b], acc[a_][b]);
    }
  };

  // weight ring of WS stages: the tiles of K-steps kk+1 .. kk+WS-1 are in flight while K-step kk is computed (WS = 3 where the LDS
  // budget keeps the same number of workgroups per CU: one K-step of compute is shorter than an L2 round trip)
  int icc = 0, itap = 0, istage = 0;                       // next (slice, tap) to issue and its ring stage
  auto issue_next = [&]() {
    issue_weights(icc, itap, istage);
    if (++itap == 9) { itap = 0; ++icc; }
    if (++istage == WS) istage = 0;
  };
#pragma unroll
  for (int q = 0; q < WS - 1; ++q)
    if (q < nk) issue_next();
  int kk = 0, cstage = 0;
  for (int cc = 0; cc < nchunk; ++cc) {
    if (cc > 0) {                                       // every wave has finished reading the previous slice's strip (reads completed)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    issue_strip(cc);
    for (int tr = 0; tr < 3; ++tr)
      for (int ts = 0; ts < 3; ++ts, ++kk) {
        // weight stage kk (and, on the first tap of a slice, the strip issued after the prefetched stages) must have landed
        // (lgkmcnt(0): the reads of the stage that is refilled after this barrier have completed -- see igemm_fwd_kernel)
        if (WS == 2 || (tr | ts) == 0 || kk + WS - 2 >= nk) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else                                                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((WS - 2) * B_INSTR) : "memory");
        __builtin_amdgcn_s_barrier();                  // strip + weight stage kk visible; the stage of K-step kk-1 is no longer read
        asm volatile("" ::: "memory");
        if (kk + WS - 1 < nk) issue_next();
        compute(tr, ts, cstage);
        if (++cstage == WS) cstage = 0;
      }
  }
  __syncthreads();   // all waves are done with strip / ring before the epilogue reuses the LDS (tile_epilogue syncs only for bf16 outputs)
  const ClassView cv = {};
  tile_epilogue<BM, BN, NW, WM, WN, PT, CT, false, BNEPI>(acc, smem, a.M, m0, n0, tile_m, bias, Yv, ldy, accumulate, stat_sum, stat_sq, Kout, tid, lane,
                                                   wm, wn, cv, bnepi, tile_m);
}

// ------------------------------------------------------------------------------------------------------------------
// wgrad: D[co][kcol] = sum_p dY[p][co] * X[p][kcol]
// ------------------------------------------------------------------------------------------------------------------
constexpr int WG_BKC = 128;   // columns of D per workgroup (k-columns = (tap, ci))
constexpr int WG_BP = 64;     // pixels per stage (2 MFMA k-steps)

// LDS image [64 pixels][128 columns] bf16 with plain 256-byte rows and the 16-byte chunk index XOR-ed with
// f(row) = ((row & 3) << 2) | ((row >> 2) & 3): filled lane-linearly by LDS-DMA (swizzle applied on the source side) and read
// transposed with ds_read_b64_tr_b16 without bank conflicts (cdna guide T10, image (b)).
__device__ __forceinline__ int wg_f(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ bf16x8_t tr_frag(const char* img, int p0, int col0, int lane) {
  // lane l (g = l>>4, i = l&15) receives image[p0 + 8g + j][col0 + i], j = 0..7 (two 4x16 transposed block reads)
  const int gq = lane >> 4, i = lane & 15;
  const int r0 = p0 + 8 * gq + (i >> 2), r1 = r0 + 4;
  const int ch = (col0 >> 3) + ((i & 3) >> 1), hb = (i & 1) << 3;
  typedef s16x4_t __attribute__((address_space(3))) * lds_ptr_t;
  s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(img + r0 * 256 + ((ch ^ wg_f(r0)) << 4) + hb));
  s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(img + r1 * 256 + ((ch ^ wg_f(r1)) << 4) + hb));
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  s16x8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, r);
}

// per-stage advance of the pixel cursor (64 pixels): 64 = dn * (Ho*Wo) + dh * Wo + dw, and the byte sizes of the two streamed tensors
struct WgradStep { int dn, dh, dw; unsigned x_bytes, y_bytes; long long slab; int gx, gy, xcd; };   // slab > 0: split z stores to dW + z * slab (no atomics)

// Weight gradient dW[co][kcol] = sum over pixels dY[pix][co] * X[pix][kcol] (kcol = (tap, ci)): NW waves = 2 along co x NW/2 along
// kcol, split-K over pixel ranges (blockIdx.z), fp32 atomics into dW.
// CAT = false (single source): the gather runs division-free -- every lane keeps the (n, ho, wo) cursor of its rows and advances it by
// the constant 64-pixel step, and the loads go through buffer descriptors (buffer_load ... lds), whose range check supplies the zeros of
// padding / tails (offset 0x80000000 = out of range) instead of a selected zero-page pointer: ~55 VALU per stage instead of ~200 with
// a dozen quarter-rate 32-bit multiplies, which had made the kernel VALU-bound (SQ_INSTS_VALU / SQ_INSTS_MFMA = 11.7).
// CAT = true (fused upsample + concat source) keeps the generic pointer path.
template <int BCO, int NW, bool CAT>
__global__ __launch_bounds__(NW * 64) void igemm_wgrad_kernel(Gather g, const bf16_t* __restrict__ dY, int ldy,
                                                          float* __restrict__ dW, int Kout, int steps_per_split, WgradStep ws) {
  constexpr int IMG = WG_BP * 256;          // bytes of one [64 pix][128 col] image
  constexpr int COT = BCO / 32;             // 16-row co tiles per wave
  constexpr int WC = NW / 2;                // waves along kcol
  constexpr int XT = WG_BKC / WC / 16;      // 16-column kcol tiles per wave
  constexpr int IPW = 16 / NW;              // LDS-DMA instructions per wave per image (4 rows x 256 B each)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;  // wave tile: co [wr*BCO/2, +BCO/2) x kcol [wc*16*XT, +16*XT)
  const Bid3 bid = wgrad_block(ws.gx, ws.gy, ws.xcd);
  const int kc0 = bid.x * WG_BKC, co0 = bid.y * BCO;
  const int nsteps = (g.M + WG_BP - 1) / WG_BP;
  const int s_begin = bid.z * steps_per_split;
  const int s_end = min(nsteps, s_begin + steps_per_split);
  if (s_begin >= s_end) return;

  // LDS-DMA geometry: instruction j of this wave covers image rows (wave*IPW + j)*4 .. +3; lane -> row +(lane >> 4), slot lane & 15,
  // which holds chunk slot ^ f(row) with f(row) = ((lane >> 4) << 2) | ((wave*IPW + j) & 3)
  const int lr = lane >> 4, slot = lane & 15;
  int x_tr[IPW], x_ts[IPW], x_c[IPW], y_c[IPW];
  bool x_kv[IPW], y_cv[IPW];
#pragma unroll
  for (int j = 0; j < IPW; ++j) {
    const int ch = slot ^ ((lr << 2) | ((wave * IPW + j) & 3));
    const int q = (kc0 >> 3) + ch;
    const int tap = q >> g.lgC8;
    x_c[j] = (q & ((1 << g.lgC8) - 1)) << 3;
    x_kv[j] = tap < g.RS;
    x_tr[j] = (min(tap, 127) * g.magicS) >> 16;
    x_ts[j] = tap - x_tr[j] * g.S;
    y_c[j] = co0 + ch * 8;
    y_cv[j] = (BCO == 128 || ch < 8) && y_c[j] < Kout;
  }
  const int hw = g.Ho * g.Wo;

  f32x4_t acc[COT][XT];
#pragma unroll
  for (int a = 0; a < COT; ++a)
#pragma unroll
    for (int b = 0; b < XT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // ---- single-source cursor state
  constexpr unsigned OOB = 0x80000000u;
  int cn[IPW], chh[IPW], cww[IPW], cm[IPW], th[IPW], tw[IPW];
  unsigned yoff[IPW];
  if constexpr (!CAT) {
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
      cm[j] = s_begin * WG_BP + (wave * IPW + j) * 4 + lr;
      int rem;
      fast_divmod(cm[j], hw, g.rhw, cn[j], rem);
      fast_divmod(rem, g.Wo, g.rw, chh[j], cww[j]);
      th[j] = x_tr[j] - g.pad_h;
      tw[j] = x_ts[j] - g.pad_w;
      yoff[j] = ((unsigned)cm[j] * (unsigned)ldy + (unsigned)y_c[j]) * 2u;
    }
  }
  const unsigned ystep = (unsigned)(WG_BP * ldy * 2);

  auto issue_stage = [&](int st, int buf) {
    char* sX = smem + buf * 2 * IMG + wave * (IPW * 1024);
    char* sY = sX + IMG;
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
      if constexpr (!CAT) {
        const bool inm = cm[j] < g.M;
        const int hn = (int)__umul24((unsigned)chh[j], (unsigned)g.smul) + th[j];
        const int wn = (int)__umul24((unsigned)cww[j], (unsigned)g.smul) + tw[j];
        const bool ok = inm & x_kv[j] & ((unsigned)hn < (unsigned)g.Hs) & ((unsigned)wn < (unsigned)g.Ws);
        const unsigned pix = mad24(mad24((unsigned)cn[j], (unsigned)g.Hs, (unsigned)hn), (unsigned)g.Ws, (unsigned)wn);
        const unsigned xo = mad24(pix, (unsigned)g.C1, (unsigned)x_c[j]) * 2u;
        buffer_load_lds16(g.src1, ws.x_bytes, sX + j * 1024, ok ? xo : OOB);
        buffer_load_lds16(dY, ws.y_bytes, sY + j * 1024, (inm & y_cv[j]) ? yoff[j] : OOB);
        // advance the cursor by 64 pixels
        cm[j] += WG_BP;
        yoff[j] += ystep;
        const int w2 = cww[j] + ws.dw;
        const int cw = w2 >= g.Wo;
        cww[j] = w2 - (cw ? g.Wo : 0);
        const int h2 = chh[j] + ws.dh + cw;
        const int chc = h2 >= g.Ho;
        chh[j] = h2 - (chc ? g.Ho : 0);
        cn[j] += ws.dn + chc;
      } else {
        const int m = st * WG_BP + (wave * IPW + j) * 4 + lr;
        const bf16_t* px = reinterpret_cast<const bf16_t*>(&g_zero16);
        const bf16_t* py = px;
        if (m < g.M) {
          int n, rem, ho, wo;
          fast_divmod(m, hw, g.rhw, n, rem);
          fast_divmod(rem, g.Wo, g.rw, ho, wo);
          RowInfo r;
          r.n = n; r.hb = ho * g.smul - g.pad_h; r.wb = wo * g.smul - g.pad_w;
          px = gather_addr(g, r, x_tr[j], x_ts[j], x_c[j], x_kv[j]);
          if (y_cv[j]) py = dY + (size_t)m * ldy + y_c[j];
        }
        __builtin_amdgcn_global_load_lds((gptr_t)px, (lptr_t)(sX + j * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)py, (lptr_t)(sY + j * 1024), 16, 0, 0);
      }
    }
  };
  auto compute_stage = [&](int buf) {
    const char* sX = smem + buf * 2 * IMG;
    const char* sY = sX + IMG;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t yf[COT], xf[XT];
#pragma unroll
      for (int a = 0; a < COT; ++a) yf[a] = tr_frag(sY, ks * 32, wr * (BCO / 2) + a * 16, lane);
#pragma unroll
      for (int b = 0; b < XT; ++b) xf[b] = tr_frag(sX, ks * 32, wc * (XT * 16) + b * 16, lane);
#pragma unroll
      for (int a = 0; a < COT; ++a)
#pragma unroll
        for (int b = 0; b < XT; ++b) acc[a][b] = YOLO_MFMA_16x16x32(yf[a], xf[b], acc[a][b]);
    }
  };

  issue_stage(s_begin, 0);
  for (int st = s_begin; st < s_end; ++st) {
    const int buf = (st - s_begin) & 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // loads landed; this wave's reads of the buffer refilled next completed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (st + 1 < s_end) issue_stage(st + 1, buf ^ 1);
    compute_stage(buf);
  }

  // D[row = co][col = kcol]: lane holds rows 4*(lane>>4)+j, column lane&15
#pragma unroll
  for (int a = 0; a < COT; ++a)
#pragma unroll
    for (int b = 0; b < XT; ++b) {
      const int kc = kc0 + wc * (XT * 16) + b * 16 + (lane & 15);
      if (kc < g.Kg) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int co = co0 + wr * (BCO / 2) + a * 16 + (lane >> 4) * 4 + j;
          if (co < Kout) {
            if (ws.slab) dW[(size_t)bid.z * (size_t)ws.slab + (size_t)co * g.Kg + kc] = acc[a][b][j];
            else         atomicAdd(dW + (size_t)co * g.Kg + kc, acc[a][b][j]);
          }
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of 3x3 / stride 1 / SAME convolutions with the three taps of one kernel row sharing one X image
// ------------------------------------------------------------------------------------------------------------------
// The generic kernel above gathers X once per tap.  Here a workgroup owns (kernel row tr, 64 input channels cc, BCO output channels):
// D[co][s][ci] for s = 0..2, and per 64-pixel stage loads the X strip of pixels [m0 + (tr-1)W - 1, +66) ONCE (128-byte rows = the 64
// channels); tap s of pixel k is strip row k + s, so the three taps are three shifted transposed reads of one LDS image.  The bytes
// fetched per FLOP drop from 1/64 (128 x 128 tile) to 1/126 (128 x 192), 1/43 -> 1/92 for 64-channel layers.
//  - vertical validity (y + tr - 1 outside the image, which in the linear strip is the neighbouring image) depends only on the X
//    pixel's own row: those strip rows are loaded as zeros (buffer range check);
//  - horizontal validity: tap s = 0 is invalid for output pixels with x == 0, s = 2 for x == W-1; the lanes that hold such pixels read
//    a zero row instead (per-lane address select).
// X image swizzle: 32-byte group index ^= f(row), f(row) = bit1(row) | bit3(row) << 1 -- conflict-free for ds_read_b64_tr_b16 at every
// row shift (brute-forced over all alignments).
// Diagnostic builds only (csrc/Makefile target `diag`, -DWG_STAMPS; tools/probes/wgrad_stamps.py): per wave, s_memtime cycles of the pipelined
// strip weight gradient summed over its stages in {counted wait, barrier, stage body} and the spans outside the loop.
#ifdef WG_STAMPS
__device__ unsigned long long* g_wg_stamps = nullptr;
#define WG_T(var)                                                                          \
  do {                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");          \
    __builtin_amdgcn_sched_barrier(0);                                                     \
  } while (0)
#else
#define WG_T(var) do {} while (0)
#endif

struct WgradStripArgs {
  const bf16_t* x; unsigned x_bytes;
  const bf16_t* dy; unsigned y_bytes;
  int H, W, C, Cout, M, Kg;
  int dh, dw;                 // 64 % (H*W) = dh * W + dw
  int d4, d32, d36;           // 4 % W, 32 % W, 36 % W
  float rhw, rw;
  long long slab;             // > 0: split z stores to out + z * slab; 0: float atomics
  int steps_per_split;
  int gx, gy, xcd;
};

__device__ __forceinline__ int wgx_f(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

template <int BCO, int NST, bool PIPE = false>
__global__ __launch_bounds__(512, 4) void wgrad3x3_strip_kernel(WgradStripArgs a, float* __restrict__ out) {
  constexpr int COT = BCO / 32;               // 16-row co tiles per wave (8 waves: 2 along co x 4 along ci)
  constexpr int XROWS = 72;                   // 66 needed, 9 LDS-DMA instructions of 8 rows
  constexpr int XIMG = XROWS * 128, YIMG = WG_BP * 256, STAGE = XIMG + YIMG;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef WG_STAMPS
  unsigned long long W0 = 0, W1 = 0, W2 = 0, W3 = 0, wa = 0, wb = 0, wc_ = 0, wd = 0, sw_wait = 0, sw_bar = 0, sw_body = 0;
#endif
  WG_T(W0);
  const int wr = wave >> 2, wc = wave & 3;    // wave tile: co [wr*BCO/2, +BCO/2) x ci [wc*16, +16) x 3 taps
  const Bid3 bid = wgrad_block(a.gx, a.gy, a.xcd);
  const int tr = bid.x % 3, cc = bid.x / 3;
  const int co0 = bid.y * BCO;
  const int nsteps = (a.M + WG_BP - 1) / WG_BP;
  const int s_begin = bid.z * a.steps_per_split;
  const int s_end = min(nsteps, s_begin + a.steps_per_split);
  if (s_begin >= s_end) return;
  // zero rows (redirect target of invalid taps); with NST == 3 also the landing zone (1 KB of zeros) of the dummy out-of-range load
  // that keeps the number of loads per stage the same on every wave, so that s_waitcnt vmcnt can leave one stage in flight
  char* const sZero = smem + NST * STAGE;
  if (tid < 64) *reinterpret_cast<uint4*>(sZero + tid * 16) = make_uint4(0u, 0u, 0u, 0u);
  constexpr unsigned OOB = 0x80000000u;
  const int hw = a.H * a.W;

  // ---- X strip loads: instruction ix covers strip rows 8 ix .. +7 (wave w issues ix = w, wave 0 also ix = 8); lane -> row + (lane >> 3),
  // 16-byte slot lane & 7 holding source chunk slot ^ (f(row) << 1); strip row e = pixel st*64 + (tr-1)*W - 1 + e
  const int xl_row = lane >> 3;
  int xe[2], xoff[2], xy[2], xx[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    xe[u] = (u == 0 ? wave : 8) * 8 + xl_row;
    const int px = s_begin * WG_BP + (tr - 1) * a.W - 1 + xe[u];
    const int chunk = (lane & 7) ^ (wgx_f(xe[u]) << 1);
    xoff[u] = (px * a.C + cc * 64 + chunk * 8) * 2;          // negative / beyond the tensor -> out of the buffer range -> zeros
    int n_, rem;
    fast_divmod(px + hw, hw, a.rhw, n_, rem);               // px >= -(W + 1) > -H*W
    fast_divmod(rem, a.W, a.rw, xy[u], xx[u]);
  }
  const int bad_y = tr == 0 ? a.H - 1 : (tr == 2 ? 0 : -1);   // strip rows of this image row belong to the neighbouring image / padding
  const int xstep = WG_BP * a.C * 2;
  // ---- dY loads (as in the generic kernel): instruction j covers image rows (wave*2 + j)*4 .. +3, 256-byte rows
  const int lr = lane >> 4, slot = lane & 15;
  unsigned yoff[2];
  bool y_cv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ch = slot ^ ((lr << 2) | ((wave * 2 + j) & 3));
    const int m = s_begin * WG_BP + (wave * 2 + j) * 4 + lr;
    y_cv[j] = (BCO == 128 || ch < 8);
    yoff[j] = ((unsigned)m * (unsigned)a.Cout + (unsigned)(co0 + ch * 8)) * 2u;   // m >= M is beyond the buffer -> zeros
  }
  const unsigned ystep = (unsigned)(WG_BP * a.Cout * 2);

  // ---- transposed-read geometry: lane (g = lane >> 4, i = lane & 15) reads pixels pk = 8g + (i >> 2) + {0, 4, 32, 36}; X address of
  // (pk, tap s) = strip row pk + s, 32-byte group wc, 8-byte piece i & 3
  const int gq = lane >> 4, i16 = lane & 15;
  const int pbase = 8 * gq + (i16 >> 2);
  int xaddr[4][3];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int row = pbase + (k & 1) * 4 + (k >> 1) * 32 + s;
      xaddr[k][s] = row * 128 + ((wc ^ wgx_f(row)) << 5) + (i16 & 3) * 8;
    }
  const int zaddr = NST * STAGE;                       // relative to smem
  // x coordinate of output pixel s_begin*64 + pbase
  int x0;
  {
    int n_, rem, y_;
    fast_divmod(s_begin * WG_BP + pbase, hw, a.rhw, n_, rem);
    fast_divmod(rem, a.W, a.rw, y_, x0);
  }

  f32x4_t acc[COT][3];
#pragma unroll
  for (int c = 0; c < COT; ++c)
#pragma unroll
    for (int s = 0; s < 3; ++s) acc[c][s] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto issue_stage = [&](int buf) {
    char* sX = smem + buf * STAGE;
    char* sY = sX + XIMG + wave * 2048;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (NST == 3 && u == 1 && wave != 0) {
        buffer_load_lds16(a.x, a.x_bytes, sZero, OOB);      // count-keeping dummy: zeros onto the zero rows
      } else if (u == 0 || wave == 0) {
        buffer_load_lds16(a.x, a.x_bytes, sX + (u == 0 ? wave : 8) * 1024, xy[u] == bad_y ? OOB : (unsigned)xoff[u]);
        xoff[u] += xstep;
        const int w2 = xx[u] + a.dw;
        const int cw = w2 >= a.W;
        xx[u] = w2 - (cw ? a.W : 0);
        const int h2 = xy[u] + a.dh + cw;
        xy[u] = h2 - (h2 >= a.H ? a.H : 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      buffer_load_lds16(a.dy, a.y_bytes, sY + j * 1024, y_cv[j] ? yoff[j] : OOB);
      yoff[j] += ystep;
    }
  };
  auto compute_stage = [&](int buf) {
    const int xb = buf * STAGE;
    const char* sY = smem + xb + XIMG;
    // horizontal validity of the 4 pixels of this lane
    int xk[4];
    xk[0] = x0;
    xk[1] = x0 + a.d4;  xk[1] -= xk[1] >= a.W ? a.W : 0;
    xk[2] = x0 + a.d32; xk[2] -= xk[2] >= a.W ? a.W : 0;
    xk[3] = x0 + a.d36; xk[3] -= xk[3] >= a.W ? a.W : 0;
    typedef s16x4_t __attribute__((address_space(3))) * lds_ptr_t;
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t yf[COT], xf[3];
#pragma unroll
      for (int c = 0; c < COT; ++c) yf[c] = tr_frag(sY, ks * 32, wr * (BCO / 2) + c * 16, lane);
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        int a0 = xb + xaddr[2 * ks][s], a1 = xb + xaddr[2 * ks + 1][s];
        if (s == 0) { a0 = xk[2 * ks] == 0 ? zaddr : a0; a1 = xk[2 * ks + 1] == 0 ? zaddr : a1; }
        if (s == 2) { a0 = xk[2 * ks] == a.W - 1 ? zaddr : a0; a1 = xk[2 * ks + 1] == a.W - 1 ? zaddr : a1; }
        s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(smem + a0));
        s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(smem + a1));
        s16x8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        xf[s] = __builtin_bit_cast(bf16x8_t, r);
      }
#pragma unroll
      for (int c = 0; c < COT; ++c)
#pragma unroll
        for (int s = 0; s < 3; ++s) acc[c][s] = YOLO_MFMA_16x16x32(yf[c], xf[s], acc[c][s]);
    }
    // advance the lane's pixel cursor by 64
    x0 += a.dw;
    x0 -= x0 >= a.W ? a.W : 0;
  };

  if constexpr (PIPE) {
    // Software-pipelined variant (NST == 2).  The waves of a workgroup run in lockstep behind the per-stage barrier, so with the plain
    // loop every wave on a SIMD is in its VALU phase (cursor updates, tap-validity selects, address adds: ~100 VALU per stage) or in its
    // MFMA phase (24 MFMAs) at the same time and the two phases add up instead of overlapping (measured: the kernel is insensitive to
    // ring depth and to LDS-read ILP, and 1.2x faster with this body at equal occupancy).  Here the per-lane LDS addresses are
    // stage-invariant registers (buffer, half-step and pixel-group offsets are compile-time constants that fold into the DS offset
    // field), and the only per-stage VALU work -- the loads and the tap-validity selects of the NEXT stage -- is branch-free, independent
    // of the current stage's MFMAs and in the same scheduling region, so the scheduler slots it between them.
    static_assert(NST == 2, "pipelined variant is double-buffered");
    typedef s16x4_t __attribute__((address_space(3))) * lds_ptr_t;
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    int ya[COT][2];                                        // dY fragment addresses of half step 0 (half step 1: + 32 rows, same swizzle)
#pragma unroll
    for (int c = 0; c < COT; ++c) {
      const int col0 = wr * (BCO / 2) + c * 16;
      const int r0 = 8 * gq + (i16 >> 2), r1 = r0 + 4;
      const int ch = (col0 >> 3) + ((i16 & 3) >> 1), hb = (i16 & 1) << 3;
      ya[c][0] = XIMG + r0 * 256 + ((ch ^ wg_f(r0)) << 4) + hb;
      ya[c][1] = XIMG + r1 * 256 + ((ch ^ wg_f(r1)) << 4) + hb;
    }
    // X fragment addresses: xaddr[k][s] with k = 2 * ks + lohi; half step 1 = half step 0 + 32 strip rows (bits 1 and 3 of the row, which
    // the swizzle uses, do not change): only xaddr[0..1][s] are kept
    int xs[2][2][2];                                       // selected addresses of the side taps (s = 0, 2) of the stage being read: [ks][side][lohi]
    auto select_x = [&](auto bufc) {                       // for the stage whose first pixel has column x0; advances x0 by 64 pixels
      constexpr int B = decltype(bufc)::value;
      constexpr int zrel = NST * STAGE - B * STAGE;        // the zero rows, seen from this buffer's base
      int xk[4];
      xk[0] = x0;
      xk[1] = x0 + a.d4;  xk[1] -= xk[1] >= a.W ? a.W : 0;
      xk[2] = x0 + a.d32; xk[2] -= xk[2] >= a.W ? a.W : 0;
      xk[3] = x0 + a.d36; xk[3] -= xk[3] >= a.W ? a.W : 0;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int col = xk[2 * ks + h];
          xs[ks][0][h] = col == 0 ? zrel : xaddr[h][0] + ks * 4096;
          xs[ks][1][h] = col == a.W - 1 ? zrel : xaddr[h][2] + ks * 4096;
        }
      x0 += a.dw;
      x0 -= x0 >= a.W ? a.W : 0;
    };
    // branch-free loads of the next stage: every wave issues 4 instructions (the ninth strip instruction belongs to wave 0; the other
    // waves aim theirs out of range at the zero rows); past the last stage of this split the loads fetch the next split's pixels (or
    // zeros beyond the tensor) into the buffer nobody reads any more
    const unsigned x1_mask = wave == 0 ? 0u : OOB;
    auto issue_next = [&](auto bufc) {
      constexpr int B = decltype(bufc)::value;
      char* sX = smem + B * STAGE;
      buffer_load_lds16(a.x, a.x_bytes, sX + wave * 1024, xy[0] == bad_y ? OOB : (unsigned)xoff[0]);
      buffer_load_lds16(a.x, a.x_bytes, wave == 0 ? sX + 8 * 1024 : sZero, (xy[1] == bad_y ? OOB : (unsigned)xoff[1]) | x1_mask);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        xoff[u] += xstep;
        const int w2 = xx[u] + a.dw;
        const int cw = w2 >= a.W;
        xx[u] = w2 - (cw ? a.W : 0);
        const int h2 = xy[u] + a.dh + cw;
        xy[u] = h2 - (h2 >= a.H ? a.H : 0);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        buffer_load_lds16(a.dy, a.y_bytes, sX + XIMG + wave * 2048 + j * 1024, y_cv[j] ? yoff[j] : OOB);
        yoff[j] += ystep;
      }
    };
    auto rd = [&](const char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)p); };
    auto pack = [&](s16x4_t lo, s16x4_t hi) {
      s16x8_t r = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      return __builtin_bit_cast(bf16x8_t, r);
    };
    auto body = [&](auto bufc) {
      constexpr int B = decltype(bufc)::value;
      WG_T(wa);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this stage landed; this wave's reads of the other buffer completed
      WG_T(wb);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      WG_T(wc_);
      const char* base = smem + B * STAGE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t yf[COT], xf[3];
#pragma unroll
        for (int c = 0; c < COT; ++c) yf[c] = pack(rd(base + ya[c][0] + ks * 8192), rd(base + ya[c][1] + ks * 8192));
        xf[0] = pack(rd(base + xs[ks][0][0]), rd(base + xs[ks][0][1]));
        xf[1] = pack(rd(base + xaddr[0][1] + ks * 4096), rd(base + xaddr[1][1] + ks * 4096));
        xf[2] = pack(rd(base + xs[ks][1][0]), rd(base + xs[ks][1][1]));
        if (ks == 0) issue_next(std::integral_constant<int, (B ^ 1)>{});
        else         select_x(std::integral_constant<int, (B ^ 1)>{});
#pragma unroll
        for (int c = 0; c < COT; ++c)
#pragma unroll
          for (int s = 0; s < 3; ++s) acc[c][s] = YOLO_MFMA_16x16x32(yf[c], xf[s], acc[c][s]);
      }
#ifdef WG_STAMPS
      WG_T(wd);
      sw_wait += wb - wa; sw_bar += wc_ - wb; sw_body += wd - wc_;
#endif
    };
    issue_next(std::integral_constant<int, 0>{});
    select_x(std::integral_constant<int, 0>{});
    WG_T(W1);
    for (int st = s_begin; st < s_end; st += 2) {
      body(std::integral_constant<int, 0>{});
      if (st + 1 < s_end) body(std::integral_constant<int, 1>{});
    }
    WG_T(W2);
  } else if (NST == 2) {
    issue_stage(0);
    for (int st = s_begin; st < s_end; ++st) {
      const int buf = (st - s_begin) & 1;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // loads landed; this wave's reads of the buffer refilled next completed
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (st + 1 < s_end) issue_stage(buf ^ 1);
      compute_stage(buf);
    }
  } else {
    // three stages: stage st+1 stays in flight (4 loads per wave) while stage st is awaited; stage st+2 refills the buffer of stage st-1
    issue_stage(0);
    if (s_begin + 1 < s_end) issue_stage(1);
    int buf = 0;
    for (int st = s_begin; st < s_end; ++st) {
      if (st + 1 < s_end) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (st + 2 < s_end) issue_stage(buf == 0 ? 2 : buf - 1);
      compute_stage(buf);
      buf = buf == 2 ? 0 : buf + 1;
    }
  }

  // D[row = co][col = ci]: lane holds rows 4*(lane>>4)+j, column lane&15
  float* dst = out + (a.slab ? (size_t)bid.z * (size_t)a.slab : 0);
#pragma unroll
  for (int c = 0; c < COT; ++c)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int kc = (tr * 3 + s) * a.C + cc * 64 + wc * 16 + (lane & 15);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int co = co0 + wr * (BCO / 2) + c * 16 + (lane >> 4) * 4 + j;
        if (a.slab) dst[(size_t)co * a.Kg + kc] = acc[c][s][j];
        else        atomicAdd(dst + (size_t)co * a.Kg + kc, acc[c][s][j]);
      }
    }
#ifdef WG_STAMPS
  if constexpr (PIPE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WG_T(W3);
    if (g_wg_stamps && lane == 0) {
      unsigned long long* o = g_wg_stamps + ((size_t)blockIdx.x * 8 + wave) * 16;
      o[0] = W0; o[1] = W1 - W0; o[2] = W2 - W1; o[3] = W3 - W2; o[4] = sw_wait; o[5] = sw_bar; o[6] = sw_body; o[7] = (unsigned long long)(s_end - s_begin); o[9] = W3;
    }
  }
#endif
}

// dW[i] (+)= sum over the split slabs: 256 threads = 64 float4 columns x 4 slab groups (coalesced 1 KiB rows, 4-deep unrolled loads),
// the groups meet in LDS.  Plain stores + this pass replace the float atomics of the one-pass kernel, which had become 44 % of the
// weight-gradient time (~0.47 G lane-atomics/us device-wide, while the same bytes as plain stores are nearly free).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float4* __restrict__ part, int nslab, long long slab4, float4* __restrict__ dW,
                                                           int n4, int accumulate) {
  __shared__ float4 red[4][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + col;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const float4* p = part + i;
    int z = grp;
    for (; z + 12 < nslab; z += 16) {
      const float4 a = p[(size_t)z * slab4], b = p[(size_t)(z + 4) * slab4], c = p[(size_t)(z + 8) * slab4], d = p[(size_t)(z + 12) * slab4];
      s.x += (a.x + b.x) + (c.x + d.x); s.y += (a.y + b.y) + (c.y + d.y);
      s.z += (a.z + b.z) + (c.z + d.z); s.w += (a.w + b.w) + (c.w + d.w);
    }
    for (; z < nslab; z += 4) {
      const float4 a = p[(size_t)z * slab4];
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
  }
  red[grp][col] = s;
  __syncthreads();
  if (grp == 0 && i < n4) {
    float4 r = red[0][col];
#pragma unroll
    for (int k = 1; k < 4; ++k) { r.x += red[k][col].x; r.y += red[k][col].y; r.z += red[k][col].z; r.w += red[k][col].w; }
    if (accumulate) { const float4 o = dW[i]; r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w; }
    dW[i] = r;
  }
}

// The same summation for MANY layers in one launch (a whole gradient bucket): every layer keeps its slabs in a private region of one
// arena until its bucket is complete, then one grid sums them all -- 31 small launches per step (each 4-70 us, mostly ramp and tail)
// become 3.  tab[e] = {first float4 of dW in `grads`, first float4 of the slabs in `arena`, float4s per slab, slabs, first workgroup}.
__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const long long* __restrict__ tab, int n, const float4* __restrict__ arena,
                                                                   float4* __restrict__ grads, int total_blocks) {
  __shared__ float4 red[256];
  // the grid may be smaller than the table's block count ("reduce_wgs" tuning): a workgroup then walks blocks b, b + grid, ... -- the same
  // sums block by block, but a launch that leaves compute units free for the kernels of the other stream
  for (long long vb = blockIdx.x; vb < total_blocks; vb += gridDim.x) {
    int e = 0;
    for (int k = 1; k < n; ++k) e = (vb >= tab[k * 5 + 4]) ? k : e;      // n <= a few dozen, uniform: scalar loads
    const long long dst4 = tab[e * 5], src4 = tab[e * 5 + 1], n4 = tab[e * 5 + 2];
    const int nslab = (int)tab[e * 5 + 3];
    // a workgroup covers 64 float4 columns with 4 slab lanes, or -- layers with many slabs (YOLO_REDUCE_WIDE_SLABS and more: small layers
    // split over hundreds of pixel ranges, the stem's per-workgroup slabs) -- 16 columns with 16 lanes: 4x the loads in flight per column
    const bool wide = nslab >= YOLO_REDUCE_WIDE_SLABS;
    const int ncol = wide ? 16 : 64, nl = wide ? 16 : 4;
    const int col = threadIdx.x % ncol, grp = threadIdx.x / ncol;
    const long long i = (vb - tab[e * 5 + 4]) * ncol + col;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
      const float4* p = arena + src4 + i;
      int z = grp;
      // (slabs are read exactly once: non-temporal loads keep them from displacing the L2 lines of the convolution running beside this launch)
      auto ldslab = [&](int zz) {
        typedef float f32x4nt_t __attribute__((ext_vector_type(4)));
        const f32x4nt_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4nt_t*>(p + (size_t)zz * n4));
        return make_float4(v.x, v.y, v.z, v.w);
      };
      for (; z + 3 * nl < nslab; z += 4 * nl) {
        const float4 a = ldslab(z), b = ldslab(z + nl), c = ldslab(z + 2 * nl), d = ldslab(z + 3 * nl);
        s.x += (a.x + b.x) + (c.x + d.x); s.y += (a.y + b.y) + (c.y + d.y);
        s.z += (a.z + b.z) + (c.z + d.z); s.w += (a.w + b.w) + (c.w + d.w);
      }
      for (; z < nslab; z += nl) {
        const float4 a = ldslab(z);
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
      }
    }
    red[grp * ncol + col] = s;
    __syncthreads();
    if (grp == 0 && i < n4) {
      float4 r = red[col];
      for (int k = 1; k < nl; ++k) { const float4 o = red[k * ncol + col]; r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w; }
      grads[dst4 + i] = r;
    }
    __syncthreads();      // (red is rewritten by the next block)
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

// all layers in one launch: table[l] = {src_off, dst_off, Cout, RS, Cin, tile_begin, tiles_ci, tiles_co} (element offsets into the flat
// bf16 buffers); tile order inside a layer: tap-major, then co tile, then ci tile
__global__ void repack_dgrad_batched_kernel(const bf16_t* __restrict__ wf, bf16_t* __restrict__ wd, const int* __restrict__ table, int nlayers) {
  __shared__ bf16_t tile[32][33];
  // the layer of this tile: last l with tile_begin[l] <= blockIdx.x.  Binary search: the linear scan was up to 30 DEPENDENT scalar loads per
  // workgroup, several microseconds for 16 K workgroups that move 2 KB each
  int lo = 0, hi = nlayers - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= table[mid * 8 + 5]) lo = mid; else hi = mid - 1;
  }
  const int* t = table + lo * 8;
  const int Cout = t[2], RS = t[3], Cin = t[4], tci = t[6], tco = t[7];
  int id = blockIdx.x - t[5];
  const int tap = id / (tci * tco);
  id -= tap * tci * tco;
  const int co0 = (id / tci) * 32, ci0 = (id % tci) * 32;
  const bf16_t* src = wf + t[0];
  bf16_t* dst = wd + t[1];
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + threadIdx.x;
    tile[r][threadIdx.x] = (co < Cout && ci < Cin) ? src[((size_t)co * RS + tap) * Cin + ci] : (bf16_t)0;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + threadIdx.x;
    if (ci < Cin && co < Cout) dst[((size_t)ci * RS + (RS - 1 - tap)) * Cout + co] = tile[threadIdx.x][r];
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
  YOLO_CHECK_ARG((size_t)p->N * p->H * p->W < (1u << 24) && (size_t)p->N * p->Ho * p->Wo < (1u << 24), "row decode needs N*H*W < 2^24");
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
  g.rhw = 1.0f / (float)(g.Ho * g.Wo); g.rw = 1.0f / (float)g.Wo; g.magicS = 65536 / g.S + 1;
  g.role = 0; g.bnepi = 0;
  g.s2 = 0; g.N = p->N; g.S_full = p->S; g.wKg = g.Kg; g.OH = p->Ho; g.OW = p->Wo;
  return g;
}

// tile choice: 128 x 128 for wide layers, 128 x 64 for 64-channel outputs; 64-pixel tiles when 128-pixel tiles would leave
// most of the 256 CUs idle (the 13x13 / 26x26 layers at batch 32)
struct TileCfg { int bm, bn; };
TileCfg pick_tile(int M, int Kout) {
  TileCfg t;
  t.bn = (Kout % 128 == 0) ? 128 : 64;
  const long tiles128 = (long)((M + 127) / 128) * (Kout / t.bn);
  t.bm = tiles128 < 512 ? 64 : 128;
  // still fewer than 2 workgroups per CU with 64 x 128 tiles (the 13 x 13 layers at batch 32): 64 x 64 tiles (measured +8 %)
  if (t.bm == 64 && t.bn == 128 && (long)((M + 63) / 64) * (Kout / 128) < 512) t.bn = 64;
  return t;
}
// strip kernel plan: 0 = not eligible, else the pixel tile BM (and the channel tile through *bnp)
// tuning overrides (yolo_set_tuning): strip_bm = -1 auto, 0 = never use the strip kernel, 64 / 128 / 256 = force; strip_bn = 0 auto
int g_strip_bm = -1, g_strip_bn = 0;
int g_wgrad_pipe = 1;    // "wgrad_pipe": software-pipelined stage body of the strip weight gradient (next stage's VALU work under the MFMAs)
int g_wgrad_ring = 2;    // "wgrad_ring": stages of the strip weight-gradient's operand ring (2 or 3)
int g_wgrad_xcd = 1;     // "wgrad_xcd": 1 = 1-D weight-gradient grids with one contiguous run of logical blocks per XCD, 0 = plain 3-D grid
int g_strip_ws = 0;      // "strip_ws": 0 auto, 2 / 3 force the weight-ring depth of the strip kernel
// "strip_xsplit": 0 = contiguous tile runs per XCD, 2 / 4 = XCD-aware rectangles in the strip kernel, -1 (default) = 2 where that lowers the fetch
// (launch_strip_ws_e).  Alone the 13 x 13 / 26 x 26 launches
// take the same time either way (they are not bound by their fetch); in the step, beside the weight-gradient stream, 2 gains +0.5 % (5 of 5
// pairs, 8484 against 8444 images/s; 4: +0.2 %; profiles/r04_strip_xsplit_ab.txt) -- fewer weight bytes cross the fabric the two streams share
int g_strip_xsplit = -1;
// workgroups aimed at by the two-phase path: 1.5 per CU measured best on the whole step (256 / 384 / 512 tried: fewer slabs to sum
// and less competition with the main stream's kernels outweigh the shorter pixel ranges of 512)
int g_wgrad_target = 384;   // "wgrad_target" tuning
int g_s2_classes = 1;    // "s2_classes" = 0 keeps the stride-2 data gradient on the strided (den = 2) gather
int g_wgrad_strip = 1;   // "wgrad_strip" = 0 keeps the weight gradient of 3x3 stride-1 layers on the generic kernel
int pick_strip(const Gather& g, int Kout, bool f32, int* bnp = nullptr) {
  if (f32 || g.den != 1 || g.C0 != 0 || g.S != 3 || g.RS != 9 || g.smul != 1 || g.pad_h != 1 || g.pad_w != 1) return 0;
  if (g.Hs != g.Ho || g.Ws != g.Wo || g.C1 % 64 != 0 || Kout % 64 != 0) return 0;
  const size_t nimg = (size_t)g.M / ((size_t)g.Ho * g.Wo);
  if (nimg * g.Hs * g.Ws * g.C1 * 2 >= (1ull << 31) || (size_t)Kout * g.Kg * 2 >= (1ull << 31)) return 0;
  const int env = g_strip_bm;
  if (env == 0) return 0;
  // measured on the ResNet18-YOLOv3 layers at batch 32 (tools/conv_bench.py): 256-pixel tiles on the widest maps (halo 2W+2 rows per
  // tile), 128 otherwise, 64 on the 13 x 13 maps; 64-channel tiles (3 workgroups per CU) unless that makes > 2048 workgroups
  int bm = g.Wo >= 80 ? 256 : (g.M >= 8192 ? 128 : 64);
  int bn = (Kout % 128 == 0 && (long)((g.M + bm - 1) / bm) * (Kout / 64) > 2048) ? 128 : 64;
  if (env > 0) {
    bm = env;
    if (g_strip_bn == 64 || (g_strip_bn == 128 && Kout % 128 == 0)) bn = g_strip_bn;
  }
  if (bm != 64 && bm != 128 && bm != 256) return 0;
  if (bnp) *bnp = bn;
  const size_t lds = (size_t)((bm + 2 * g.Wo + 2 + 7) / 8 * 8) * 128 + 128 + 2 * (size_t)bn * 128;
  if (lds > 160 * 1024) return 0;
  return bm;
}
// the 32x32x16 kernel takes a problem when forced ("s32" > 0), or by its automatic rule while the strip kernel's tile choice is automatic too
// ("strip_bm" = 0 / 64 / 128 / 256 asks for the implicit-GEMM / a specific strip kernel: tests and A/B runs)
int s32_plan(const Gather& g, int Kout, bool f32, S32PlanOut* out) {
  if (g_s32 <= 0 && g_strip_bm != -1) return 0;
  return yolo_s32_plan(g, Kout, f32, out);
}
int stat_rows_for(const Gather& g, int Kout) {
  if (const int sp = yolo_stream_plan(g, Kout, false, nullptr)) return (g.M + sp - 1) / sp;
  if (const int s3 = s32_plan(g, Kout, false, nullptr)) return (g.M + s3 - 1) / s3;
  const int sb = pick_strip(g, Kout, false);
  if (sb) return (g.M + sb - 1) / sb;
  const TileCfg t = pick_tile(g.M, Kout);
  return (g.M + t.bm - 1) / t.bm;      // one partial row per pixel tile
}

// two-level partial rows (conv_common.h rows_fold): group size for R raw rows, and the rows a caller allocates for them
int g_reduce_wgs = 0;    // "reduce_wgs" tuning: most workgroups of the batched slab sum (0 = one per table block)
int g_row_group = 16;    // "row_group" tuning: raw rows folded per group (0 = the *_g entry points keep raw rows)
int row_group_for(const Gather& g, int R) {
  if (g_row_group <= 0 || g.s2 || R <= 0) return 0;
  int G = g_row_group;
  while (G < YOLO_ROW_GROUP_MAX && (R + G - 1) / G > 64) G *= 2;      // at most ~64 group rows for the consumer's prologue
  return G;
}
void row_layout(int R, int G, int K, int rs, int32_t* info) {          // {rows to allocate, group rows, group size, raw rows}
  if (G <= 0) { info[0] = R; info[1] = R; info[2] = 0; info[3] = R; return; }
  const int P = yolo_row_groups(R, G);
  const long counters = (long)P * (K / 64);                            // one per (group, 64-channel tile): the narrowest channel tile of any kernel
  info[0] = P + R + (int)((counters + rs - 1) / rs) + 1;
  info[1] = P; info[2] = G; info[3] = R;
}

template <int BM, int BN, int NW, int WS, bool BNEPI>
int launch_strip_ws_e(const Gather& g, const void* w, void* y, int ldy, int accumulate, const Epi& e, int Kout, hipStream_t st) {
  StripArgs a;
  a.src = g.src1;
  a.src_bytes = (unsigned)((size_t)g.M / ((size_t)g.Ho * g.Wo) * g.Hs * g.Ws * g.C1 * 2);
  a.wt = (const bf16_t*)w;
  a.wt_bytes = (unsigned)((size_t)Kout * g.Kg * 2);
  a.H = g.Ho; a.W = g.Wo; a.C = g.C1; a.M = g.M; a.Kg = g.Kg;
  a.E8 = (BM + 2 * g.Wo + 2 + 7) / 8 * 8;
  a.rhw = g.rhw; a.rw = g.rw;
  const size_t lds_main = (size_t)a.E8 * 128 + 128 + WS * (size_t)BN * 128, lds_out = (size_t)BM * (BN * 2 + 16);
  const size_t lds = lds_main > lds_out ? lds_main : lds_out;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_strip_kernel<BM, BN, NW, WS, BNEPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { yolo_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  const int tiles_m = (g.M + BM - 1) / BM, tn = Kout / BN;
  int grid = tiles_m * tn;
  a.xsplit = 0;
  // "strip_xsplit": -1 (default) = 2 x 4 rectangles where they lower the bytes the eight L2s fetch together -- 8 W / G + X G for weights W and
  // input X, against 8 W + X for contiguous runs: G = 2 wins iff X < 4 W (the 13 x 13 layers: 5.5 MB of pixels, 2.4-4.7 MB of weights; not the
  // 26 x 26 ones: 11 MB against 1.2-2.4) --, 0 never, 2 / 4 always
  int G = g_strip_xsplit;
  if (G < 0) G = (size_t)g.M * g.C1 < 4 * (size_t)Kout * g.Kg ? 2 : 0;
  if (G && tn % G == 0 && tiles_m >= 8) {
    const int parts = 8 / G;
    a.xsplit = G;
    grid = 8 * ((tiles_m + parts - 1) / parts) * (tn / G);
  }
  hipLaunchKernelGGL((conv3x3_strip_kernel<BM, BN, NW, WS, BNEPI>), dim3(grid), dim3(NW * 64), lds, st, a, nullptr, y, ldy, accumulate, e.ssum, e.ssq,
                     Kout, tn, e.bn);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}

template <int BM, int BN, int NW, int WS>
int launch_strip_ws(const Gather& g, const void* w, void* y, int ldy, int accumulate, const Epi& e, int Kout, hipStream_t st) {
  if (e.bn.y) return launch_strip_ws_e<BM, BN, NW, WS, true>(g, w, y, ldy, accumulate, e, Kout, st);      // (y: the fused BatchNorm reduce is on)
  return launch_strip_ws_e<BM, BN, NW, WS, false>(g, w, y, ldy, accumulate, e, Kout, st);
}

// third weight stage where three workgroups per CU still fit in the 160 KiB of LDS (measured: +7..14 % on the 26 x 26 / 13 x 13 maps,
// -25 % where it drops the 104 x 104 maps to one workgroup per CU)
template <int BM, int BN, int NW>
int launch_strip(const Gather& g, const void* w, void* y, int ldy, int accumulate, const Epi& e, int Kout, hipStream_t st) {
  const size_t strip = (size_t)((BM + 2 * g.Wo + 2 + 7) / 8 * 8) * 128 + 128, out = (size_t)BM * (BN * 2 + 16);
  auto lds = [&](int ws) { const size_t m = strip + (size_t)ws * BN * 128; return m > out ? m : out; };
  const size_t cap = 160 * 1024;
  if (g_strip_ws != 2 && (g_strip_ws == 3 || cap / lds(3) >= 3) && lds(3) <= cap)
    return launch_strip_ws<BM, BN, NW, 3>(g, w, y, ldy, accumulate, e, Kout, st);
  return launch_strip_ws<BM, BN, NW, 2>(g, w, y, ldy, accumulate, e, Kout, st);
}

template <int BM, int BN, int NS, bool F32, bool FAST, int NW, bool BNEPI>
int launch_tile3_e(const Gather& g, const void* w, const float* bias, void* y, int ldy, int accumulate, const Epi& e, int Kout,
                hipStream_t st) {
  const int tiles_m = (g.M + BM - 1) / BM, tn = Kout / BN;
  constexpr size_t lds_ring = NS * (BM * BK * 2 + BN * BK * 2), lds_out = (size_t)BM * (BN * 2 + 16);
  constexpr size_t lds = lds_ring > lds_out ? lds_ring : lds_out;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_fwd_kernel<BM, BN, NS, F32, FAST, NW, BNEPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { yolo_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  hipLaunchKernelGGL((igemm_fwd_kernel<BM, BN, NS, F32, FAST, NW, BNEPI>), dim3(tiles_m * tn, g.s2 ? g.s2_ny : 1), dim3(NW * 64), lds, st, g, (const bf16_t*)w, bias, y, ldy, accumulate,
                     e.ssum, e.ssq, Kout, tn, e.bn);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}

template <int BM, int BN, int NS, bool F32, bool FAST, int NW>
int launch_tile3(const Gather& g, const void* w, const float* bias, void* y, int ldy, int accumulate, const Epi& e, int Kout,
                 hipStream_t st) {
  if constexpr (!F32) {
    if (e.bn.y) return launch_tile3_e<BM, BN, NS, F32, FAST, NW, true>(g, w, bias, y, ldy, accumulate, e, Kout, st);
  }
  return launch_tile3_e<BM, BN, NS, F32, FAST, NW, false>(g, w, bias, y, ldy, accumulate, e, Kout, st);
}

// Ring depth: 2 stages (64 KB for 128 x 128: two workgroups per CU; 48 KB for 128 x 64 / 64 x 128: three).  Measured on MI355X
// (tools/conv_bench.py, batch 32): 3 / 4 stages at fewer workgroups per CU are 20-45 % slower -- waves per SIMD matter more than lookahead
// on these short K loops -- and a ring of finer 32-wide units (64-byte rows) is 30-80 % slower: half-line LDS-DMA pieces double the L2
// requests, rows must stay whole 128-byte lines.
template <int BM, int BN, bool F32, bool FAST>
int launch_tile2(const Gather& g, const void* w, const float* bias, void* y, int ldy, int accumulate, const Epi& e, int Kout,
                 hipStream_t st) {
  // 128 x 128 tiles run with 8 waves (each 32 pixels x 64 channels): same LDS, twice the waves per SIMD; measured 128-channel layer
  // fwd 56 -> 49 us, dgrad 48 -> 38 us; on 128 x 64 tiles it is mixed (fwd slower), so those keep 4 waves
  if constexpr (BM == 128 && BN == 128) return launch_tile3<BM, BN, 2, F32, FAST, 8>(g, w, bias, y, ldy, accumulate, e, Kout, st);
  else return launch_tile3<BM, BN, 2, F32, FAST, 4>(g, w, bias, y, ldy, accumulate, e, Kout, st);
}

template <int BM, int BN, bool F32>
int launch_tile(const Gather& g, const void* w, const float* bias, void* y, int ldy, int accumulate, const Epi& e, int Kout,
                hipStream_t st) {
  const bool fast = g.den == 1 && g.C0 == 0 && g.S <= 9 && g.RS <= 81 &&
                    (size_t)g.Hs * g.Ws * g.C1 * (size_t)(g.M / (g.Ho * g.Wo) + 1) < (1ull << 31);
  if (fast) return launch_tile2<BM, BN, F32, true>(g, w, bias, y, ldy, accumulate, e, Kout, st);
  return launch_tile2<BM, BN, F32, false>(g, w, bias, y, ldy, accumulate, e, Kout, st);
}

template <bool F32>
int launch_fwd(const Gather& g, const void* w, const float* bias, void* y, int ldy, int accumulate, const Epi& e,
               int Kout, hipStream_t st) {
  int sbn = 0;
  if constexpr (!F32) {
    if (!bias && accumulate != 2 && yolo_stream_plan(g, Kout, false, nullptr)) return yolo_stream_launch(g, w, y, ldy, accumulate, e, Kout, st);
    if (!bias && (accumulate != 2 || g.s2) && s32_plan(g, Kout, false, nullptr)) return yolo_s32_launch(g, w, y, ldy, accumulate, e, Kout, st);
  }
  if (const int sb = pick_strip(g, Kout, F32, &sbn)) {
    const bool wide = sbn == 128;
    if (sb == 64) return wide ? launch_strip<64, 128, 4>(g, w, y, ldy, accumulate, e, Kout, st)
                              : launch_strip<64, 64, 4>(g, w, y, ldy, accumulate, e, Kout, st);
    if (sb == 256) return wide ? launch_strip<256, 128, 8>(g, w, y, ldy, accumulate, e, Kout, st)
                               : launch_strip<256, 64, 8>(g, w, y, ldy, accumulate, e, Kout, st);
    return wide ? launch_strip<128, 128, 8>(g, w, y, ldy, accumulate, e, Kout, st)
                : launch_strip<128, 64, 4>(g, w, y, ldy, accumulate, e, Kout, st);
  }
  const TileCfg t = pick_tile(g.M, Kout);
  if (t.bm == 128 && t.bn == 128) return launch_tile<128, 128, F32>(g, w, bias, y, ldy, accumulate, e, Kout, st);
  if (t.bm == 128 && t.bn == 64) return launch_tile<128, 64, F32>(g, w, bias, y, ldy, accumulate, e, Kout, st);
  if (t.bm == 64 && t.bn == 128) return launch_tile<64, 128, F32>(g, w, bias, y, ldy, accumulate, e, Kout, st);
  return launch_tile<64, 64, F32>(g, w, bias, y, ldy, accumulate, e, Kout, st);
}

}  // namespace

extern int g_fused_min_chunks;
extern int g_ew_nt;                  // eltwise.hip
extern int64_t g_acc_stream_elems;   // eltwise.hip
extern int64_t g_rows_stream_elems;  // eltwise.hip
extern int g_rows_grid;              // eltwise.hip
extern int g_pool_scatter;
extern int g_bwd_fin_small;     // eltwise.hip
extern int g_reduce_cap;        // eltwise.hip
extern int g_fused_small_chunks;   // eltwise.hip
extern int g_opt_wgs;              // optim.hip
extern int g_s32_s2;               // conv_s32.hip

extern "C" int yolo_set_tuning(const char* name, int value) {
  YOLO_CHECK_ARG(name != nullptr, "null name");
  if (!strcmp(name, "stem_direct")) return yolo_stem_set_direct(value);
  if (!strcmp(name, "dw_tiled")) return yolo_dw_set_tiled(value);
  if (!strcmp(name, "strip_bm")) { YOLO_CHECK_ARG(value == -1 || value == 0 || value == 64 || value == 128 || value == 256, "strip_bm"); g_strip_bm = value; }
  else if (!strcmp(name, "wgrad_strip")) { YOLO_CHECK_ARG(value == 0 || value == 1, "wgrad_strip"); g_wgrad_strip = value; }
  else if (!strcmp(name, "bn_fused_min_chunks")) { YOLO_CHECK_ARG(value >= 1 && value <= 12, "bn_fused_min_chunks"); g_fused_min_chunks = value; }
  else if (!strcmp(name, "s2_classes")) { YOLO_CHECK_ARG(value == 0 || value == 1, "s2_classes"); g_s2_classes = value; }
  else if (!strcmp(name, "wgrad_target")) { YOLO_CHECK_ARG(value >= 64 && value <= 4096, "wgrad_target"); g_wgrad_target = value; }
  else if (!strcmp(name, "bn_fused_small_grid")) { YOLO_CHECK_ARG(value == 0 || (value >= 16 && value <= 255), "bn_fused_small_grid"); g_fused_small_chunks = value; }
  else if (!strcmp(name, "reduce_cap")) { YOLO_CHECK_ARG(value >= 64 && value <= 4096, "reduce_cap"); g_reduce_cap = value; }
  else if (!strcmp(name, "pool_scatter")) { YOLO_CHECK_ARG(value == 0 || value == 1, "pool_scatter"); g_pool_scatter = value; }
  else if (!strcmp(name, "wgrad_pipe")) { YOLO_CHECK_ARG(value == 0 || value == 1, "wgrad_pipe"); g_wgrad_pipe = value; }
  else if (!strcmp(name, "wgrad_ring")) { YOLO_CHECK_ARG(value == 2 || value == 3, "wgrad_ring"); g_wgrad_ring = value; }
  else if (!strcmp(name, "wgrad_xcd")) { YOLO_CHECK_ARG(value == 0 || value == 1, "wgrad_xcd"); g_wgrad_xcd = value; }
  else if (!strcmp(name, "strip_xsplit")) { YOLO_CHECK_ARG(value == -1 || value == 0 || value == 2 || value == 4, "strip_xsplit"); g_strip_xsplit = value; }
  else if (!strcmp(name, "strip_ws")) { YOLO_CHECK_ARG(value == 0 || value == 2 || value == 3, "strip_ws"); g_strip_ws = value; }
  else if (!strcmp(name, "ew_nt")) { YOLO_CHECK_ARG(value >= 0 && value <= 3, "ew_nt"); g_ew_nt = value; }
  else if (!strcmp(name, "acc_stream_kelems")) { YOLO_CHECK_ARG(value >= 0, "acc_stream_kelems"); g_acc_stream_elems = (int64_t)value * 1000; }
  else if (!strcmp(name, "bwd_fin_small")) { YOLO_CHECK_ARG(value == 0 || value == 1, "bwd_fin_small"); g_bwd_fin_small = value; }
  else if (!strcmp(name, "stream")) { YOLO_CHECK_ARG(value >= -1 && value <= 2, "stream"); g_stream = value; }
  else if (!strcmp(name, "rows_stream_kelems")) { YOLO_CHECK_ARG(value >= 0, "rows_stream_kelems"); g_rows_stream_elems = (int64_t)value * 1000; }
  else if (!strcmp(name, "rows_grid")) { YOLO_CHECK_ARG(value >= 64 && value <= 4096, "rows_grid"); g_rows_grid = value; }
  else if (!strcmp(name, "row_group")) { YOLO_CHECK_ARG(value == 0 || value == 8 || value == 16 || value == 32 || value == 64, "row_group"); g_row_group = value; }
  else if (!strcmp(name, "reduce_wgs")) { YOLO_CHECK_ARG(value == 0 || (value >= 16 && value <= 65536), "reduce_wgs"); g_reduce_wgs = value; }
  else if (!strcmp(name, "s32_s2")) { YOLO_CHECK_ARG(value == 0 || value == 1, "s32_s2"); g_s32_s2 = value; }
  else if (!strcmp(name, "opt_wgs")) { YOLO_CHECK_ARG(value >= 16 && value <= 2048, "opt_wgs"); g_opt_wgs = value; }
  else if (!strcmp(name, "wgrad9_wgs")) { YOLO_CHECK_ARG(value >= 16 && value <= 1024, "wgrad9_wgs"); g_wgrad9_wgs = value; }
  else if (!strcmp(name, "wgrad9")) { YOLO_CHECK_ARG(value >= -1 && value <= 1, "wgrad9"); g_wgrad9 = value; }
  else if (!strcmp(name, "s32")) { YOLO_CHECK_ARG(value >= -1 && value <= 16, "s32"); g_s32 = value; }
  else if (!strcmp(name, "strip_bn")) { YOLO_CHECK_ARG(value == 0 || value == 64 || value == 128, "strip_bn"); g_strip_bn = value; }
  else YOLO_CHECK_ARG(false, "unknown tuning name");
  return YOLO_OK;
}

extern "C" int yolo_conv2d_stat_rows(const yolo_conv_problem* p) {
  if (!p || p->Cout % 64 != 0 || p->N <= 0 || p->Ho <= 0 || p->Wo <= 0) return YOLO_ERR_INVALID_ARG;
  if (check_problem(p)) return YOLO_ERR_INVALID_ARG;
  if (yolo_stem_applies(p)) return yolo_stem_stat_rows(p);
  static const char dummy = 0;
  return stat_rows_for(fwd_gather(p, &dummy, &dummy), p->Cout);
}

extern "C" int yolo_conv2d_fwd_plan(const yolo_conv_problem* p, int32_t* info) {
  YOLO_CHECK_ARG(info != nullptr, "null info");
  int rc = check_problem(p);
  if (rc) return rc;
  for (int i = 0; i < 8; ++i) info[i] = 0;
  if (yolo_stem_applies(p)) { info[0] = 3; info[4] = yolo_stem_stat_rows(p); return YOLO_OK; }
  static const char dummy = 0;
  const Gather g = fwd_gather(p, &dummy, &dummy);
  StreamPlanOut sp;
  if (yolo_stream_plan(g, p->Cout, false, &sp)) {
    info[0] = 4; info[1] = 64; info[2] = 64; info[3] = sp.span; info[4] = sp.nx * sp.ny; info[5] = (int32_t)sp.lds;
    return YOLO_OK;
  }
  S32PlanOut s3;
  if (s32_plan(g, p->Cout, false, &s3)) {
    info[0] = 5; info[1] = s3.bm; info[2] = s3.bn; info[3] = s3.bm; info[4] = s3.tiles; info[5] = (int32_t)s3.lds; info[6] = s3.id;
    return YOLO_OK;
  }
  int sbn = 0;
  if (const int sb = pick_strip(g, p->Cout, false, &sbn)) {
    info[0] = 1; info[1] = sb; info[2] = sbn; info[3] = sb; info[4] = (g.M + sb - 1) / sb * (p->Cout / sbn);
    return YOLO_OK;
  }
  const TileCfg t = pick_tile(g.M, p->Cout);
  info[1] = t.bm; info[2] = t.bn; info[3] = t.bm; info[4] = (g.M + t.bm - 1) / t.bm * (p->Cout / t.bn);
  return YOLO_OK;
}

extern "C" int yolo_conv2d_fwd(const yolo_conv_problem* p, const void* src0, const void* src1, const void* w_fwd,
                               const float* bias, void* y, int y_is_f32, float* stat_sum, float* stat_sq, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  YOLO_CHECK_ARG(src1 && w_fwd && y, "null pointer");
  YOLO_CHECK_ARG(p->C0 == 0 || src0, "C0 > 0 needs src0");
  YOLO_CHECK_ARG((stat_sum == nullptr) == (stat_sq == nullptr), "stat_sum and stat_sq go together");
  YOLO_CHECK_ARG(!(y_is_f32 && stat_sum), "statistics are defined on bf16 outputs only");
  if (!y_is_f32 && !bias && yolo_stem_applies(p)) return yolo_stem_fwd(p, src1, w_fwd, y, stat_sum, stat_sq, stream);   // RGB stem: row-walking kernel
  Gather g = fwd_gather(p, src0, src1);
  if (y_is_f32) return launch_fwd<true>(g, w_fwd, bias, y, p->Cout, 0, Epi{}, p->Cout, (hipStream_t)stream);
  return launch_fwd<false>(g, w_fwd, bias, y, p->Cout, 0, Epi{stat_sum, stat_sq, {}}, p->Cout, (hipStream_t)stream);
}

// Two-level statistics rows (conv_common.h rows_fold).  info4 = {rows the caller allocates (zeroed ONCE: the kernels keep the counters zero
// between launches), group rows P the consumer reads (rows [0, P) of stat_sum / stat_sq), group size, raw rows}; group size 0 = this problem
// keeps plain rows (the RGB stem's row-walking kernel) and yolo_conv2d_fwd_g behaves like yolo_conv2d_fwd.
extern "C" int yolo_conv2d_stat_group_layout(const yolo_conv_problem* p, int32_t* info4) {
  YOLO_CHECK_ARG(info4 != nullptr, "null info");
  int rc = check_problem(p);
  if (rc) return rc;
  YOLO_CHECK_ARG(p->Cout % 64 == 0, "Cout must be a multiple of 64");
  if (yolo_stem_applies(p)) { row_layout(yolo_stem_stat_rows(p), 0, p->Cout, p->Cout, info4); return YOLO_OK; }
  static const char dummy = 0;
  const Gather g = fwd_gather(p, &dummy, &dummy);
  const int R = stat_rows_for(g, p->Cout);
  row_layout(R, row_group_for(g, R), p->Cout, p->Cout, info4);
  return YOLO_OK;
}

// yolo_conv2d_fwd (16-bit output, no bias, statistics) with the rows laid out as yolo_conv2d_stat_group_layout says
extern "C" int yolo_conv2d_fwd_g(const yolo_conv_problem* p, const void* src0, const void* src1, const void* w_fwd, void* y, float* stat_sum,
                                 float* stat_sq, void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  YOLO_CHECK_ARG(src1 && w_fwd && y && stat_sum && stat_sq, "null pointer");
  YOLO_CHECK_ARG(p->C0 == 0 || src0, "C0 > 0 needs src0");
  if (yolo_stem_applies(p)) return yolo_stem_fwd(p, src1, w_fwd, y, stat_sum, stat_sq, stream);
  Gather g = fwd_gather(p, src0, src1);
  Epi e = {stat_sum, stat_sq, {}};
  e.bn.rows = stat_rows_for(g, p->Cout);
  e.bn.group = row_group_for(g, e.bn.rows);
  return launch_fwd<false>(g, w_fwd, nullptr, y, p->Cout, 0, e, p->Cout, (hipStream_t)stream);
}

// yolo_conv2d_fwd (16-bit output, no bias) whose BatchNorm statistics go into an exact accumulator block (common.h yolo_acc_*: Q = 2,
// C = Cout, yolo_acc_words(2, Cout) int64 words, zeroed by the caller before the launch) instead of one partial row per pixel tile: the
// consumer sums YOLO_ACC_NB buckets whatever the tile count, so no finalize launch is needed between this kernel and the BatchNorm apply
extern "C" int yolo_conv2d_fwd_acc(const yolo_conv_problem* p, const void* src0, const void* src1, const void* w_fwd, void* y, int64_t* stat_acc,
                                   void* stream) {
  int rc = check_problem(p);
  if (rc) return rc;
  YOLO_CHECK_ARG(src1 && w_fwd && y && stat_acc, "null pointer");
  YOLO_CHECK_ARG(p->C0 == 0 || src0, "C0 > 0 needs src0");
  YOLO_CHECK_ARG(!yolo_stem_applies(p), "the RGB stem kernel writes statistics rows (yolo_conv2d_fwd)");
  Gather g = fwd_gather(p, src0, src1);
  Epi e = {};
  e.bn.acc = (long long*)stat_acc;
  return launch_fwd<false>(g, w_fwd, nullptr, y, p->Cout, 0, e, p->Cout, (hipStream_t)stream);
}

extern "C" int64_t yolo_acc_words(int Q, int C) { return (Q > 0 && C > 0) ? (int64_t)yolo_acc_block_words(Q, C) : 0; }

namespace {
// conv-transpose as a forward gather over dy with flipped taps: src = (row - (R-1-pad) + tap') / stride
int dgrad_gather(const yolo_conv_problem* p, const void* dy, Gather* gp, bool even_only = false) {
  int rc = check_problem(p);
  if (rc) return rc;
  YOLO_CHECK_ARG(p->Cin % 64 == 0, "dgrad needs Cin % 64 == 0");
  YOLO_CHECK_ARG(ilog2_exact(p->Cout / 8) >= 0, "dgrad needs Cout/8 to be a power of two");
  Gather& g = *gp;
  g.src0 = nullptr; g.src1 = (const bf16_t*)dy;
  g.Hs = p->Ho; g.Ws = p->Wo; g.C0 = 0; g.C1 = p->Cout;
  g.lgC8 = ilog2_exact(p->Cout / 8);
  g.Ho = p->H; g.Wo = p->W; g.S = p->S; g.RS = p->R * p->S;
  g.smul = 1; g.pad_h = p->R - 1 - p->pad_t; g.pad_w = p->S - 1 - p->pad_l; g.den = p->stride;
  g.M = p->N * p->H * p->W; g.Kg = p->R * p->S * p->Cout;
  g.rhw = 1.0f / (float)(g.Ho * g.Wo); g.rw = 1.0f / (float)g.Wo; g.magicS = 65536 / g.S + 1;
  g.role = 1; g.bnepi = 0;
  g.s2 = 0; g.s2_ny = 4; g.N = p->N; g.S_full = p->S; g.wKg = g.Kg; g.OH = p->H; g.OW = p->W;
  YOLO_CHECK_ARG(g.M < (1 << 24), "row decode needs N*H*W < 2^24");
  const bool k3 = p->R == 3 && p->S == 3;
  const bool k1 = even_only && p->R == 1 && p->S == 1 && p->pad_t == 0 && p->pad_l == 0;
  YOLO_CHECK_ARG(!even_only || (k1 && p->stride == 2 && p->Cout % 64 == 0), "even-only data gradient: 1x1, stride 2, no padding, Cout % 64 == 0");
  if ((g_s2_classes || k1) && p->stride == 2 && (k3 || k1) && p->H >= 2 && p->W >= 2 && p->Cout % 64 == 0) {
    // dX[h] = sum_r dY[(h + pad - r) / 2] W[r] over the r with (h + pad - r) even: for h = 2h' + ph the taps r = (ph + pad) mod 2 (+ 2),
    // taken in descending r (= ascending flipped index R-1-r) they read dY rows h' - pad', h' - pad' + 1: a stride-1 correlation.
    // The strided gather (den = 2) instead walks all 9 taps and fetches zeros for 27 of every 36 (pixel, tap) pairs.
    // (1x1: the even / even class has its one tap, the other three have none -- only that class is launched)
    const int R = p->R;
    auto dim = [R](int pad, int size, Gather::Dim (&d)[2]) {
      for (int par = 0; par < 2; ++par) {
        const int r_lo = (par + pad) & 1;                // taps r_lo, r_lo + 2 (< R)
        const int r_hi = r_lo + 2 < R ? r_lo + 2 : r_lo;
        d[par].n = r_lo < R ? (r_hi > r_lo ? 2 : 1) : 0;
        d[par].t0 = R - 1 - r_hi;                        // flipped index of the first (largest r) tap
        d[par].t1 = R - 1 - r_lo;
        d[par].pad = -((par + pad - r_hi) / 2);          // (par + pad - r_hi) is even and <= 0
        d[par].size = (size - par + 1) / 2;
      }
    };
    dim(p->pad_t, p->H, g.rowd);
    dim(p->pad_l, p->W, g.cold);
    g.s2 = 1;
    g.s2_ny = k1 ? 1 : 4;
    g.wKg = g.Kg;
    g.M = p->N * g.rowd[0].size * g.cold[0].size;        // the largest class: tile choice and grid size
    g.den = 1;
  }
  return YOLO_OK;
}
}  // namespace

extern "C" int yolo_conv2d_dgrad(const yolo_conv_problem* p, const void* dy, const void* w_dgrad, void* dx, int accumulate,
                                 void* stream) {
  YOLO_CHECK_ARG(dy && w_dgrad && dx, "null pointer");
  Gather g;
  int rc = dgrad_gather(p, dy, &g);
  if (rc) return rc;
  YOLO_CHECK_ARG(accumulate != 2 || (g.s2 && g.s2_ny == 4), "accumulate = 2 needs the parity-class data gradient (yolo_conv2d_dgrad_classed)");
  return launch_fwd<false>(g, w_dgrad, nullptr, dx, p->Cin, accumulate, Epi{}, p->Cin, (hipStream_t)stream);
}

// 1x1 / stride-2 data gradient that writes ONLY the positions it contributes to, dx[:, ::2, ::2, :] (=|+=): the other three quarters of dx
// receive nothing from this layer.  For the first writer of a fan-in whose other writer is a 3x3 stride-2 data gradient launched with
// accumulate = 2 (previous contribution at the even / even positions only): neither the zeros of this layer nor their read-back happen.
extern "C" int yolo_conv2d_dgrad_even(const yolo_conv_problem* p, const void* dy, const void* w_dgrad, void* dx, int accumulate, void* stream) {
  YOLO_CHECK_ARG(dy && w_dgrad && dx, "null pointer");
  Gather g;
  int rc = dgrad_gather(p, dy, &g, true);
  if (rc) return rc;
  return launch_fwd<false>(g, w_dgrad, nullptr, dx, p->Cin, accumulate ? 1 : 0, Epi{}, p->Cin, (hipStream_t)stream);
}

// 1 if yolo_conv2d_dgrad runs this problem as parity classes (3x3, stride 2): accumulate = 2 is accepted then
extern "C" int yolo_conv2d_dgrad_classed(const yolo_conv_problem* p) {
  Gather g;
  static const char dummy = 0;
  if (!p || dgrad_gather(p, &dummy, &g)) return 0;
  return g.s2 && g.s2_ny == 4 ? 1 : 0;
}

// dx = addend + conv_transpose(dy, w): the fan-in add of yolo_conv2d_dgrad(accumulate = 1) with the other contribution read from ITS buffer
// (e.g. the masked gradient of the residual unit that feeds dx's tensor through an identity shortcut) -- that contribution is then never
// copied into dx first
extern "C" int yolo_conv2d_dgrad_add(const yolo_conv_problem* p, const void* dy, const void* w_dgrad, void* dx, const void* addend,
                                     void* stream) {
  YOLO_CHECK_ARG(dy && w_dgrad && dx && addend, "null pointer");
  Gather g;
  int rc = dgrad_gather(p, dy, &g);
  if (rc) return rc;
  Epi e = {};
  e.bn.addend = (const bf16_t*)addend;
  return launch_fwd<false>(g, w_dgrad, nullptr, dx, p->Cin, 1, e, p->Cin, (hipStream_t)stream);
}

// rows of the [rows][3][Cin] partial-sum buffer yolo_conv2d_dgrad_bn fills (one per pixel tile, x 4 parity classes on the stride-2 path);
// rows a launch does not write (classes smaller than the grid) must stay zero: allocate the buffer zeroed, nothing else writes it
extern "C" int yolo_conv2d_dgrad_bn_rows(const yolo_conv_problem* p) {
  Gather g;
  static const char dummy = 0;
  if (!p || dgrad_gather(p, &dummy, &g)) return YOLO_ERR_INVALID_ARG;
  if ((size_t)p->N * p->H * p->W * p->Cin >= (1ull << 31)) return YOLO_ERR_INVALID_ARG;     // 32-bit element offsets in the epilogue
  g.bnepi = 1;
  return (g.s2 ? 4 : 1) * stat_rows_for(g, p->Cin);
}

// Data gradient whose output is dL/d(out) of a BatchNorm(+ReLU)(+residual / shortcut BN) unit, with that unit's backward REDUCE in the
// epilogue (see BnEpi): dx receives the masked gradient g (relu_mask: the unit's sign bytes from yolo_bn_act_fwd_mask, or null for a
// linear unit), partial[rows][3][Cin] the tile sums of g, g xhat(y, mean, rstd) and -- if y2 is given -- g xhat(y2, mean2, rstd2).
// yolo_bn_bwd_finalize over `partial` and yolo_bn_act_bwd_apply with relu = 0 on dx complete the unit's backward pass.
extern "C" int yolo_conv2d_dgrad_bn(const yolo_conv_problem* p, const void* dy, const void* w_dgrad, void* dx, int accumulate, const void* addend,
                                    const void* relu_mask, const void* y, const float* mean, const float* rstd, const void* y2,
                                    const float* mean2, const float* rstd2, float* partial, void* stream) {
  return yolo_conv2d_dgrad_bn_acc(p, dy, w_dgrad, dx, accumulate, addend, relu_mask, y, mean, rstd, y2, mean2, rstd2, partial, nullptr, stream);
}

// two-level partial rows for yolo_conv2d_dgrad_bn_g: info4 as yolo_conv2d_stat_group_layout, rows of [3][Cin] floats (stride-2 problems, whose
// parity classes leave rows unwritten, keep plain rows: group size 0)
extern "C" int yolo_conv2d_dgrad_bn_group_layout(const yolo_conv_problem* p, int32_t* info4) {
  YOLO_CHECK_ARG(info4 != nullptr, "null info");
  Gather g;
  static const char dummy = 0;
  int rc = dgrad_gather(p, &dummy, &g);
  if (rc) return rc;
  YOLO_CHECK_ARG((size_t)p->N * p->H * p->W * p->Cin < (1ull << 31), "the fused reduce addresses dx with 32-bit element offsets");
  g.bnepi = 1;
  const int R = (g.s2 ? 4 : 1) * stat_rows_for(g, p->Cin);
  row_layout(R, row_group_for(g, R), p->Cin, 3 * p->Cin, info4);
  return YOLO_OK;
}

namespace {
int dgrad_bn_impl(const yolo_conv_problem* p, const void* dy, const void* w_dgrad, void* dx, int accumulate, const void* addend, const void* relu_mask,
                  const void* y, const float* mean, const float* rstd, const void* y2, const float* mean2, const float* rstd2, float* partial,
                  int64_t* acc, bool grouped, void* stream);
}
extern "C" int yolo_conv2d_dgrad_bn_g(const yolo_conv_problem* p, const void* dy, const void* w_dgrad, void* dx, int accumulate, const void* addend,
                                      const void* relu_mask, const void* y, const float* mean, const float* rstd, const void* y2,
                                      const float* mean2, const float* rstd2, float* partial, void* stream) {
  return dgrad_bn_impl(p, dy, w_dgrad, dx, accumulate, addend, relu_mask, y, mean, rstd, y2, mean2, rstd2, partial, nullptr, true, stream);
}

// the same with the tile sums added into an exact accumulator block (Q = 3, C = Cin; common.h yolo_acc_*) when partial is null
extern "C" int yolo_conv2d_dgrad_bn_acc(const yolo_conv_problem* p, const void* dy, const void* w_dgrad, void* dx, int accumulate, const void* addend,
                                        const void* relu_mask, const void* y, const float* mean, const float* rstd, const void* y2,
                                        const float* mean2, const float* rstd2, float* partial, int64_t* acc, void* stream) {
  return dgrad_bn_impl(p, dy, w_dgrad, dx, accumulate, addend, relu_mask, y, mean, rstd, y2, mean2, rstd2, partial, acc, false, stream);
}

namespace {
int dgrad_bn_impl(const yolo_conv_problem* p, const void* dy, const void* w_dgrad, void* dx, int accumulate, const void* addend, const void* relu_mask,
                  const void* y, const float* mean, const float* rstd, const void* y2, const float* mean2, const float* rstd2, float* partial,
                  int64_t* acc, bool grouped, void* stream) {
  YOLO_CHECK_ARG(dy && w_dgrad && dx, "null pointer");
  YOLO_CHECK_ARG(y && mean && rstd && ((partial != nullptr) != (acc != nullptr)), "the fused reduce needs y, mean, rstd and either partial rows or an accumulator block");
  YOLO_CHECK_ARG(!y2 || (mean2 && rstd2), "y2 needs mean2 and rstd2");
  YOLO_CHECK_ARG((size_t)p->N * p->H * p->W * p->Cin < (1ull << 31), "the fused reduce addresses dx with 32-bit element offsets");
  Gather g;
  int rc = dgrad_gather(p, dy, &g);
  if (rc) return rc;
  g.bnepi = 1;
  Epi e = {};
  e.bn.mask = (const uint8_t*)relu_mask;
  e.bn.y = (const bf16_t*)y; e.bn.mean = mean; e.bn.rstd = rstd;
  e.bn.y2 = (const bf16_t*)y2; e.bn.mean2 = mean2; e.bn.rstd2 = rstd2;
  e.bn.partial = partial;
  e.bn.acc = (long long*)acc;
  e.bn.addend = (const bf16_t*)addend;             // non-null: the fan-in source instead of dx itself (implies accumulate)
  if (grouped && partial) {
    e.bn.rows = stat_rows_for(g, p->Cin);
    e.bn.group = row_group_for(g, e.bn.rows);
  }
  YOLO_CHECK_ARG(accumulate != 2 || (g.s2 && g.s2_ny == 4 && !addend), "accumulate = 2 needs the parity-class data gradient and no addend");
  return launch_fwd<false>(g, w_dgrad, nullptr, dx, p->Cin, accumulate == 2 ? 2 : ((accumulate || addend) ? 1 : 0), e, p->Cin, (hipStream_t)stream);
}
}  // namespace

namespace {
struct WgradPlan { Gather g; int bco, tiles_k, tiles_c, split_k, sps; bool strip, w9; };

// split-K plan: at most `target` workgroups (2 per CU: 64 KiB of LDS each) so that the whole grid is resident in one round -- one
// workgroup more than the slots costs a second, almost empty round -- and at least 8 pixel-steps per workgroup
int plan_wgrad(const yolo_conv_problem* p, const void* src0, const void* src1, int split_k, int target, WgradPlan* pl) {
  int rc = check_problem(p);
  if (rc) return rc;
  YOLO_CHECK_ARG(p->C0 == 0 || src0, "C0 > 0 needs src0");
  pl->g = fwd_gather(p, src0, src1);
  YOLO_CHECK_ARG(pl->g.M < (1 << 24), "wgrad row decode needs N*Ho*Wo < 2^24");
  pl->bco = (p->Cout % 128 == 0) ? 128 : 64;
  pl->strip = g_wgrad_strip && p->R == 3 && p->S == 3 && p->stride == 1 && p->pad_t == 1 && p->pad_l == 1 && p->Ho == p->H &&
              p->Wo == p->W && p->C0 == 0 && p->Cin % 64 == 0 && (size_t)p->N * p->H * p->W * p->Cin * 2 < (1ull << 31) &&
              (size_t)pl->g.M * p->Cout * 2 < (1ull << 31);
  pl->w9 = false;
  if (split_k <= 0 && pl->strip) {                    // the stationary-output kernel plans its own pixel splits (one workgroup per CU)
    int sps9 = 0;
    if (const int ns = yolo_wgrad9_plan(p, &sps9)) {
      pl->w9 = true; pl->split_k = ns; pl->sps = sps9; pl->tiles_k = pl->tiles_c = 1;
      return YOLO_OK;
    }
  }
  pl->tiles_k = pl->strip ? 3 * (p->Cin / 64) : (pl->g.Kg + WG_BKC - 1) / WG_BKC;
  pl->tiles_c = (p->Cout + pl->bco - 1) / pl->bco;
  const int nsteps = (pl->g.M + WG_BP - 1) / WG_BP;
  if (split_k <= 0) {
    split_k = target / (pl->tiles_k * pl->tiles_c);
    const int max_split = (nsteps + 7) / 8;
    if (split_k > max_split) split_k = max_split;
    if (split_k < 1) split_k = 1;
  }
  if (split_k > nsteps) split_k = nsteps;
  pl->sps = (nsteps + split_k - 1) / split_k;
  pl->split_k = (nsteps + pl->sps - 1) / pl->sps;     // every z owns at least one step
  YOLO_CHECK_ARG(pl->split_k <= 65535, "split_k too large");
  return YOLO_OK;
}

// slab = 0: float atomics into out; slab > 0: split z stores its partial tile to out + z * slab
void launch_wgrad(const yolo_conv_problem* p, const WgradPlan& pl, const void* dy, float* out, long long slab, hipStream_t stream) {
  const Gather& g = pl.g;
  if (pl.w9) { (void)yolo_wgrad9_launch(p, g.src1, dy, out, slab, stream); return; }
  if (pl.strip) {
    WgradStripArgs a;
    a.x = g.src1; a.x_bytes = (unsigned)((size_t)p->N * p->H * p->W * p->Cin * 2);
    a.dy = (const bf16_t*)dy; a.y_bytes = (unsigned)((size_t)g.M * p->Cout * 2);
    a.H = p->H; a.W = p->W; a.C = p->Cin; a.Cout = p->Cout; a.M = g.M; a.Kg = g.Kg;
    const int hw = p->H * p->W;
    a.dh = (WG_BP % hw) / p->W; a.dw = (WG_BP % hw) % p->W;
    a.d4 = 4 % p->W; a.d32 = 32 % p->W; a.d36 = 36 % p->W;
    a.rhw = g.rhw; a.rw = g.rw; a.slab = slab; a.steps_per_split = pl.sps;
    a.gx = pl.tiles_k; a.gy = pl.tiles_c; a.xcd = g_wgrad_xcd;
    const dim3 grid = g_wgrad_xcd ? dim3(pl.tiles_k * pl.tiles_c * pl.split_k) : dim3(pl.tiles_k, pl.tiles_c, pl.split_k);
    const int nst = g_wgrad_ring == 3 ? 3 : 2;
    const size_t lds = (size_t)nst * (72 * 128 + WG_BP * 256) + 1024;
    if (nst == 3) {
      static bool attr = false;
      if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_strip_kernel<128, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_strip_kernel<64, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
      }
      if (pl.bco == 128) hipLaunchKernelGGL((wgrad3x3_strip_kernel<128, 3>), grid, dim3(512), lds, stream, a, out);
      else               hipLaunchKernelGGL((wgrad3x3_strip_kernel<64, 3>), grid, dim3(512), lds, stream, a, out);
      return;
    }
    if (g_wgrad_pipe) {
      if (pl.bco == 128) hipLaunchKernelGGL((wgrad3x3_strip_kernel<128, 2, true>), grid, dim3(512), lds, stream, a, out);
      else               hipLaunchKernelGGL((wgrad3x3_strip_kernel<64, 2, true>), grid, dim3(512), lds, stream, a, out);
      return;
    }
    if (pl.bco == 128) hipLaunchKernelGGL((wgrad3x3_strip_kernel<128, 2>), grid, dim3(512), lds, stream, a, out);
    else               hipLaunchKernelGGL((wgrad3x3_strip_kernel<64, 2>), grid, dim3(512), lds, stream, a, out);
    return;
  }
  const size_t lds = 2 * 2 * WG_BP * 256;
  // 64 pixels = dn images + dh rows + dw columns of the output grid
  WgradStep ws;
  const int hw = p->Ho * p->Wo;
  ws.dn = WG_BP / hw;
  ws.dh = (WG_BP % hw) / p->Wo;
  ws.dw = (WG_BP % hw) % p->Wo;
  ws.slab = slab;
  const size_t xb = (size_t)p->N * p->H * p->W * (size_t)(p->Cin - p->C0) * 2, yb = (size_t)g.M * p->Cout * 2;
  const bool cat = p->C0 > 0 || xb >= (1ull << 31) || yb >= (1ull << 31) || (size_t)p->N * p->H * p->W >= (1u << 24);
  ws.x_bytes = cat ? 0u : (unsigned)xb;
  ws.y_bytes = cat ? 0u : (unsigned)yb;
  ws.gx = pl.tiles_k; ws.gy = pl.tiles_c; ws.xcd = g_wgrad_xcd;
  const dim3 grid = g_wgrad_xcd ? dim3(pl.tiles_k * pl.tiles_c * pl.split_k) : dim3(pl.tiles_k, pl.tiles_c, pl.split_k);
#define YOLO_WGRAD_LAUNCH(BCO_, CAT_)                                                                                          \
  hipLaunchKernelGGL((igemm_wgrad_kernel<BCO_, 8, CAT_>), grid, dim3(512), lds, stream, g, (const bf16_t*)dy, p->Cout, out, p->Cout, \
                     pl.sps, ws)
  if (pl.bco == 128) { if (cat) YOLO_WGRAD_LAUNCH(128, true); else YOLO_WGRAD_LAUNCH(128, false); }
  else               { if (cat) YOLO_WGRAD_LAUNCH(64, true);  else YOLO_WGRAD_LAUNCH(64, false); }
#undef YOLO_WGRAD_LAUNCH
}

}  // namespace

#ifdef WG_STAMPS
extern "C" int yolo_debug_wg_stamps(void* buf) {
  unsigned long long* p = (unsigned long long*)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wg_stamps), &p, sizeof(p));
}
#endif

extern "C" int yolo_conv2d_wgrad(const yolo_conv_problem* p, const void* src0, const void* src1, const void* dy, float* dw,
                                 int split_k, void* stream) {
  YOLO_CHECK_ARG(p && src1 && dy && dw, "null pointer");
  WgradPlan pl;
  int rc = plan_wgrad(p, src0, src1, split_k, 384, &pl);
  if (rc) return rc;
  launch_wgrad(p, pl, dy, dw, 0, (hipStream_t)stream);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}

extern "C" size_t yolo_conv2d_wgrad_workspace_bytes(const yolo_conv_problem* p) {
  WgradPlan pl;
  static const char dummy = 0;
  if (!p || plan_wgrad(p, &dummy, &dummy, 0, g_wgrad_target, &pl)) return 0;
  return pl.split_k > 1 ? (size_t)pl.split_k * (size_t)p->Cout * (size_t)pl.g.Kg * sizeof(float) : 0;
}

extern "C" int yolo_conv2d_wgrad_reduce(const yolo_conv_problem* p, const void* src0, const void* src1, const void* dy, float* dw,
                                        void* workspace, size_t workspace_bytes, int accumulate, void* stream) {
  YOLO_CHECK_ARG(p && src1 && dy && dw, "null pointer");
  WgradPlan pl;
  int rc = plan_wgrad(p, src0, src1, 0, g_wgrad_target, &pl);
  if (rc) return rc;
  const size_t n = (size_t)p->Cout * (size_t)pl.g.Kg;
  if (pl.split_k == 1) {                              // one workgroup per tile: straight into dw (plain stores, or atomics to add)
    launch_wgrad(p, pl, dy, dw, accumulate ? 0 : (long long)n, (hipStream_t)stream);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
  }
  YOLO_CHECK_ARG(workspace && workspace_bytes >= (size_t)pl.split_k * n * sizeof(float), "workspace too small (yolo_conv2d_wgrad_workspace_bytes)");
  YOLO_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && (reinterpret_cast<uintptr_t>(dw) & 15) == 0, "dw / workspace must be 16-byte aligned");
  launch_wgrad(p, pl, dy, (float*)workspace, (long long)n, (hipStream_t)stream);
  const int n4 = (int)(n / 4);                         // Cin % 8 == 0
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((n4 + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const float4*)workspace, pl.split_k,
                     (long long)(n / 4), (float4*)dw, n4, accumulate);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}

extern "C" int yolo_conv2d_wgrad_splits(const yolo_conv_problem* p) {
  WgradPlan pl;
  static const char dummy = 0;
  if (!p || plan_wgrad(p, &dummy, &dummy, 0, g_wgrad_target, &pl)) return YOLO_ERR_INVALID_ARG;
  return pl.split_k;
}

extern "C" int yolo_conv2d_wgrad_slabs(const yolo_conv_problem* p, const void* src0, const void* src1, const void* dy, float* dw, float* slabs,
                                       size_t slab_bytes, void* stream) {
  YOLO_CHECK_ARG(p && src1 && dy && dw, "null pointer");
  WgradPlan pl;
  int rc = plan_wgrad(p, src0, src1, 0, g_wgrad_target, &pl);
  if (rc) return rc;
  const size_t n = (size_t)p->Cout * (size_t)pl.g.Kg;
  // the caller sized the slab region (and the batched summing launch's table) from yolo_conv2d_wgrad_splits at planning time: a tuning
  // change since then (yolo_set_tuning "wgrad_target" / "wgrad_strip") must not silently sum stale slabs or overwrite a direct result
  if (pl.split_k == 1) {                              // one workgroup per tile: plain stores straight into dw, nothing to sum
    YOLO_CHECK_ARG(slab_bytes == 0, "the plan has one split now but the caller planned slabs: re-plan after yolo_set_tuning");
    launch_wgrad(p, pl, dy, dw, (long long)n, (hipStream_t)stream);
    YOLO_LAUNCH_CHECK();
    return YOLO_OK;
  }
  YOLO_CHECK_ARG(slabs && slab_bytes == (size_t)pl.split_k * n * sizeof(float),
                 "slab region does not match the current split plan (yolo_conv2d_wgrad_splits; re-plan after yolo_set_tuning)");
  YOLO_CHECK_ARG((reinterpret_cast<uintptr_t>(slabs) & 15) == 0, "slabs must be 16-byte aligned");
  launch_wgrad(p, pl, dy, slabs, (long long)n, (hipStream_t)stream);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}

extern "C" int yolo_wgrad_reduce_batched(const int64_t* table_dev, int nentries, int total_blocks, const float* arena, float* grads, void* stream) {
  YOLO_CHECK_ARG(table_dev && arena && grads && nentries > 0 && total_blocks > 0, "bad argument");
  YOLO_CHECK_ARG((reinterpret_cast<uintptr_t>(arena) & 15) == 0 && (reinterpret_cast<uintptr_t>(grads) & 15) == 0, "arena / grads must be 16-byte aligned");
  const int grid = (g_reduce_wgs > 0 && total_blocks > g_reduce_wgs) ? g_reduce_wgs : total_blocks;
  hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const long long*)table_dev, nentries,
                     (const float4*)arena, (float4*)grads, total_blocks);
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

extern "C" int yolo_repack_dgrad_weights_batched(const void* w_fwd_flat, void* w_dgrad_flat, const int32_t* table_dev, int nlayers,
                                                 int total_tiles, void* stream) {
  YOLO_CHECK_ARG(w_fwd_flat && w_dgrad_flat && table_dev && nlayers > 0 && total_tiles > 0, "bad argument");
  hipLaunchKernelGGL(repack_dgrad_batched_kernel, dim3(total_tiles), dim3(32, 8), 0, (hipStream_t)stream, (const bf16_t*)w_fwd_flat,
                     (bf16_t*)w_dgrad_flat, (const int*)table_dev, nlayers);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}
