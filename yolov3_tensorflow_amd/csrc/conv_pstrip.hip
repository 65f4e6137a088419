// 3x3 / stride-1 / SAME convolution (forward and data gradient), one LARGE tile per compute unit, for gfx950 (MI355X).
//
// Replaces keras.layers.Conv2D (reference backbone/basic_backbone.py:20-43 via resnet18.py:29-32, yolov3_detector.py:96,113,121,146) and
// its TF autodiff data gradient on the >= 128-channel 3x3 layers.  conv3x3_strip_kernel (conv_igemm.hip) runs 2-4 small workgroups per
// CU and loses 25-40 % of a workgroup's life in its prologue / epilogue and 12-30 % of the slots in the grid's last round; its K loop
// spends ~7 VALU instructions per pixel-fragment read on tap masks and swizzled addresses.  Here:
//
// * ONE 512-thread workgroup per CU owns a tile sized so that the layer is a single round of <= 256 tiles
//   (352 pixels x 64 channels, or 176 x 128): 8 waves = KG k-groups x MW pixel-waves x NV channel-waves, every wave an
//   11 x 2 arrangement of 16 x 16 accumulators (176 pixels x 32 channels).  The k-groups split each 64-deep K step (one tap of one
//   64-channel slice) into its two 32-deep MFMA sub-steps -- the two waves that share a SIMD run the same instruction stream on the two
//   halves of K -- and their partial sums meet once, through LDS, after the K loop.
// * The pixel strip lives in LDS in PADDED coordinates: an image row occupies Wp = roundup(W + 1, 8) LDS rows (the pad columns and one
//   separator line per image are zeros, filled by out-of-range LDS-DMA lanes), so SAME padding, row wrap and image boundaries need no
//   per-lane masks, and since Wp is a multiple of the 8-row swizzle period the tap row (tr) is a lane-uniform byte offset.  A lane
//   keeps 3 addresses per pixel fragment (ts = -1, 0, +1) for the whole tile: ONE v_add per fragment read in the K loop.
// * Strip slices are double-buffered (slice c+1 streams in behind the K steps of slice c), the weight tiles run through a ring of
//   RING stages with counted s_waitcnt vmcnt, the first fragments of K step s+1 are read before the barrier that ends K step s.
// * Epilogue: conv_common.h tile_epilogue (bf16 tile through LDS, BatchNorm statistics / fused BatchNorm-backward reduce).
namespace {
// Diagnostic builds only (make EXTRA_conv_pstrip=-DPS_STAMPS; tools/probes/pstrip_stamps.py): s_memtime stamps of wave 0 of every workgroup.
// In the product build no stamp executes and the symbol below does not exist.
#ifdef PS_STAMPS
__device__ unsigned long long* g_ps_stamps = nullptr;
#define PS_STAMP(i)                                                                                      \
  do {                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    unsigned long long t_;                                                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    if (g_ps_stamps && threadIdx.x == 0) g_ps_stamps[blockIdx.x * 32 + (i)] = t_;                        \
  } while (0)
#else
#define PS_STAMP(i) do {} while (0)
#endif

}  // namespace
#include "conv_common.h"

namespace {

struct PStripArgs {
  const bf16_t* src; unsigned src_bytes;   // NHWC activations (or dY for the data gradient)
  const bf16_t* wt;  unsigned wt_bytes;    // [Kout][9][C]
  int H, W, C, M, Kg, N;                   // M = N*H*W, Kg = 9*C
  int dq8, dr8;                            // 8 / (Wp / 8), 8 % (Wp / 8): how a wave's next strip piece (8 further) advances in (line, column group)
  int tstride;                             // pixels a tile owns (<= BM: the rows above it are computed and dropped), tile t starts at t * tstride
  int Wp;                                  // LDS rows per image line: multiple of 8, >= W + 1
  int npieces;                             // 1 KiB LDS-DMA pieces per strip slice: 1 (lead pad) + lines * Wp / 8
  int strip_bytes;                         // npieces * 1024
  float rhw, rw, rh1;                      // 1 / (H*W), 1 / W, 1 / (H+1)
};

constexpr int PS_NPW = 10;                 // strip pieces per wave and slice (10 x 8 waves x 1 KiB >= any strip that fits)
// strip pieces of the NEXT slice a wave issues in the K step of tap t: all ten by tap 4, so that the counted wait at the end of tap 6 (the
// weights of tap 8, issued in tap 5 or 6) also covers the whole slice and the barrier there publishes it before tap 8 reads ahead into it
__host__ __device__ constexpr int ps_ns(int t) { return t < 5 ? 2 : 0; }

// LDS-DMA instructions a wave has issued BEHIND its pieces of K step s+1 when it waits for them (end of the K step of tap t): the strip
// pieces of tap t-(ring-2) .. t and the weights of the ring-2 K steps in between (nwp pieces each)
__host__ __device__ constexpr int ps_behind(int t, int ring, int nwp) {
  int n = (ring - 2) * nwp;
  for (int j = 0; j <= ring - 2; ++j) n += ps_ns((t + 9 - j) % 9);
  return n;
}

template <int N_> __device__ __forceinline__ void ps_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

template <int KG, int MW, int NV, int PT, int CT, int RING, bool BNEPI>
__global__ __launch_bounds__(512) void conv3x3_pstrip_kernel(PStripArgs a, void* __restrict__ Yv, int ldy, int accumulate,
                                                            float* __restrict__ stat_sum, float* __restrict__ stat_sq, int Kout, int tiles_n,
                                                            BnEpi bnepi) {
  static_assert(KG * MW * NV == 8 && KG == 2, "8 waves in two k-groups");
  static_assert(RING == 3 || RING == 4, "weight ring depth");
  constexpr int BM = MW * PT * 16, BN = NV * CT * 16;
  constexpr int NWP = BN / 64;                       // weight pieces (8 rows x 128 B) per wave and K step
  constexpr int W_STAGE = BN * 128;
  static_assert(NWP >= 1, "channel tile too small");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS: [strip buffer 0][strip buffer 1][weight ring].  Padding pieces (they keep the vmcnt arithmetic uniform) land in the lead pad
  // of strip buffer 0: out-of-range sources write zeros, which is what that KiB holds anyway.
  const int SB = a.strip_bytes;
  const int ring0 = 2 * SB, dump0 = 0;

  PS_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave / (MW * NV), wr = wave % (MW * NV), wm = wr / NV, wn = wr % NV;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = tile % tiles_n, tile_m = tile / tiles_n;
  const int m0 = tile_m * a.tstride, n0 = tile_n * BN;
  const int m_end = m0 + a.tstride < a.M ? m0 + a.tstride : a.M;
  const int HW = a.H * a.W, H1 = a.H + 1;
  const int nchunk = a.C >> 6;
  const int nstage = nchunk * 9;
  const int WpB = a.Wp * 128;

  // extended line of the tile's first pixel: e = n * (H + 1) + y (one separator line per image); strip line l holds extended line e0 - 1 + l
  int n_first, y_first;
  {
    int n, rem, y, x;
    fast_divmod(m0, HW, a.rhw, n, rem);
    fast_divmod(rem, a.W, a.rw, y, x);
    n_first = __builtin_amdgcn_readfirstlane(n);
    y_first = __builtin_amdgcn_readfirstlane(y);
  }
  const int e0 = n_first * H1 + y_first;

  // LDS-DMA lane geometry: lane -> row lrow of the piece's 8 rows, slot lane & 7 receives chunk slot ^ lrow (rows are 8-aligned)
  const int lrow = lane >> 3;
  const int cchunk = (lane & 7) ^ lrow;
  const int wp8 = a.Wp >> 3;

  unsigned wbase[NWP];
#pragma unroll
  for (int j = 0; j < NWP; ++j) wbase[j] = (unsigned)(((n0 + (wave * NWP + j) * 8 + lrow) * a.Kg + cchunk * 8) * 2);

  auto issue_weights = [&](int s, int cc, int tap, int slot) {      // K step s = (slice cc, tap) -> ring slot; past the end: padding pieces
    const bool real = s < nstage;
    const unsigned koff = real ? (unsigned)((tap * a.C + cc * 64) * 2) : 0x80000000u;
#pragma unroll
    for (int j = 0; j < NWP; ++j) {
      char* dst = real ? smem + ring0 + slot * W_STAGE + (wave * NWP + j) * 1024 : smem + dump0;
      buffer_load_lds16(a.wt, a.wt_bytes, dst, real ? wbase[j] + koff : 0x80000000u);
    }
  };
  auto issue_strip = [&](int k, unsigned off, int cc, int buf) {   // piece k of this wave for slice cc (cc >= nchunk: padding piece)
    const int i = wave + 8 * k;
    const bool real = (i < a.npieces) && (cc < nchunk);
    char* dst = real ? smem + buf * SB + i * 1024 : smem + dump0;
    buffer_load_lds16(a.src, a.src_bytes, dst, real ? off + (unsigned)(cc * 128) : 0x80000000u);
  };
  auto rd = [&](int addr) -> bf16x8_t { return *reinterpret_cast<const bf16x8_t*>(smem + addr); };

  // ---- prologue: K steps 0 .. RING-2 of the weights and slice 0 of the strip are requested as early as their addresses are known ----
  {
    int cc = 0, tap = 0;
#pragma unroll
    for (int q = 0; q < RING - 1; ++q) {
      issue_weights(q, cc, tap, q);
      if (++tap == 9) { tap = 0; ++cc; }
    }
  }
  // source offset (slice 0) of this lane in each of the wave's strip pieces, or an out-of-range offset (-> zeros): piece i = wave + 8 k
  // holds LDS rows 8 (i - 1) .. + 7 = columns 8 xg .. + 7 of strip line l.  (l, xg) and the line's (image, row) advance piece by piece
  // in scalar registers; the lane adds its column and chunk.
  unsigned spoff[PS_NPW];
  {
    int l = 0, xg = wave - 1;                          // piece 0 (wave 0, k = 0) is the lead pad: xg = -1 marks it
    while (xg >= wp8) { xg -= wp8; ++l; }
    int n = n_first, y = y_first - 1 + l;              // extended line e0 - 1 + l
    if (y < 0) { y += H1; --n; }
    while (y >= H1) { y -= H1; ++n; }
    const unsigned lane_part = (unsigned)((lrow * a.C + cchunk * 8) * 2);
#pragma unroll
    for (int k = 0; k < PS_NPW; ++k) {
      const int x = xg * 8 + lrow;
      const bool line_ok = xg >= 0 && n >= 0 && n < a.N && y < a.H && (wave + 8 * k) < a.npieces;
      const unsigned base = (unsigned)((((n * a.H + y) * a.W + xg * 8) * a.C) * 2);
      spoff[k] = (line_ok && x < a.W) ? base + lane_part : 0x80000000u;
      issue_strip(k, spoff[k], 0, 0);
      // next piece of this wave: 8 pieces further
      if (xg < 0) { xg += 8; } else { xg += a.dr8; l += a.dq8; y += a.dq8; }
      while (xg >= wp8) { xg -= wp8; ++l; ++y; }
      while (y >= H1) { y -= H1; ++n; }
    }
  }
  PS_STAMP(1);

  // pixel fragments: LDS byte address of (pixel, k chunk of this lane) for the three tap columns, tap row 0, strip buffer 0.  Each
  // thread decodes ONE pixel of the tile into its LDS row (a table in strip buffer 1, which nothing uses before K step 0), every lane
  // then picks up the rows of its PT pixels.
  {
    int* const rowtab = reinterpret_cast<int*>(smem + SB);
    if (tid < BM) {
      int m = m0 + tid;
      m = m < m_end ? m : m_end - 1;                    // rows past the tile's pixels compute garbage that is never stored
      int n, rem, y, x;
      fast_divmod(m, HW, a.rhw, n, rem);
      fast_divmod(rem, a.W, a.rw, y, x);
      const int l = n * H1 + y - e0 + 1;                // strip line of the pixel (>= 1)
      rowtab[tid] = 8 + (l - 1) * a.Wp + x - 1;         // LDS row of tap (tr = 0, ts = 0)
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  const int kq = lane >> 4;
  const int kc0 = kq + ((KG == 2 && g == 1) ? 4 : 0);   // 16-byte chunk of the 64-channel row: sub-step * 4 + k group of the operand layout
  int pa[3][PT];
#pragma unroll
  for (int b = 0; b < PT; ++b) {
    const int r0 = reinterpret_cast<const int*>(smem + SB)[(wm * PT + b) * 16 + (lane & 15)];
#pragma unroll
    for (int ts = 0; ts < 3; ++ts) {
      const int r = r0 + ts;
      pa[ts][b] = r * 128 + (((r ^ kc0) & 7) << 4);
    }
  }
  int wa[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) wa[c] = ring0 + swz(wn * (CT * 16) + c * 16 + (lane & 15), kc0);

  f32x4_t acc[CT][PT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int b = 0; b < PT; ++b) acc[c][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (lgkmcnt: the table reads above -- slice 1 overwrites the table from K step 0 on)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  PS_STAMP(2);
  // ---- K loop: two phases per K step, the two k-groups half a step apart (the 8-phase GEMM structure of the CDNA guide) ----
  // LOAD phase: the wave reads the K step's 2 weight + PT pixel fragments into registers and issues its share of the LDS-DMA for two K
  // steps ahead; MFMA phase: 2 * PT MFMAs back to back at raised priority, no LDS or memory instruction in between.  A raw s_barrier
  // ends every phase.  Group 1 runs one barrier behind group 0, so on every SIMD one wave is in its MFMA cluster while the other loads:
  //   group 0:  LOAD(0) | MFMA(0) | LOAD(1) | MFMA(1) | ...            (phase 2s, 2s+1)
  //   group 1:          | LOAD(0) | MFMA(0) | LOAD(1) | MFMA(1) | ...  (phase 2s+1, 2s+2)
  // Ring slot (s % RING) is read in phases 2s and 2s+1; K step s+2 is issued in LOAD(s) into the slot of K step s-1 (RING = 3), behind the
  // barrier that ended phase 2s-1; every wave waits for its own pieces of K step s+1 before the barrier that starts phase 2s+2 (group 0 at
  // the end of MFMA(s), group 1 at the end of LOAD(s): the same counted vmcnt either way).  The next strip slice is issued in the LOAD
  // phases of taps 0..4 and is covered by those waits before tap 8 ends (vmcnt retires in order).
  int icc = (RING - 1) / 9, itap = (RING - 1) % 9, islot = RING - 1;   // K step being issued (RING-1 ahead) and its slot
  int cslot = 0;                                                       // slot of the K step being loaded
  if (g == 1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
  int s = 0;
  for (int cc = 0; cc < nchunk; ++cc) {
    const int sbuf = (cc & 1) * SB;
    auto stage = [&](auto TC) {
      constexpr int T = decltype(TC)::value;
      constexpr int tr = T / 3, ts = T % 3;
      constexpr int k0 = 2 * T;                         // first strip piece of this tap (ps_ns: 2,2,2,2,2,0,0,0,0)
      // -------- LOAD --------
      if (cc == 1 && T == 4) PS_STAMP(8);
      const int soff = sbuf + tr * WpB;
      const int woff = cslot * W_STAGE;
      bf16x8_t wf[CT], pf[PT];
#pragma unroll
      for (int c = 0; c < CT; ++c) wf[c] = rd(wa[c] + woff);
#pragma unroll
      for (int b = 0; b < PT; ++b) pf[b] = rd(pa[ts][b] + soff);
#ifndef PS_NODMA
      issue_weights(s + RING - 1, icc, itap, islot);
      if (++itap == 9) { itap = 0; ++icc; }
      if (++islot == RING) islot = 0;
#pragma unroll
      for (int q = 0; q < ps_ns(T); ++q) issue_strip(k0 + q, spoff[k0 + q], cc + 1, (cc + 1) & 1);
      if (g == 1) ps_wait_vmcnt<ps_behind(T, RING, NWP)>();
#endif
      __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): every fragment is in its registers (the MFMA cluster waits for nothing)
      if (cc == 1 && T == 4) PS_STAMP(9);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (cc == 1 && T == 4) PS_STAMP(10);
      // -------- MFMA --------
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int b = 0; b < PT; ++b)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c][b] = YOLO_MFMA_16x16x32(wf[c], pf[b], acc[c][b]);
      __builtin_amdgcn_s_setprio(0);
#ifndef PS_NODMA
      if (g == 0) ps_wait_vmcnt<ps_behind(T, RING, NWP)>();
#endif
      if (cc == 1 && T == 4) PS_STAMP(11);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (cc == 1 && T == 4) PS_STAMP(12);
      if (++cslot == RING) cslot = 0;
      ++s;
    };
    stage(std::integral_constant<int, 0>{});
    stage(std::integral_constant<int, 1>{});
    stage(std::integral_constant<int, 2>{});
    stage(std::integral_constant<int, 3>{});
    stage(std::integral_constant<int, 4>{});
    stage(std::integral_constant<int, 5>{});
    stage(std::integral_constant<int, 6>{});
    stage(std::integral_constant<int, 7>{});
    stage(std::integral_constant<int, 8>{});
    if (cc == 0) PS_STAMP(3);
  }
  if (g == 0) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
  PS_STAMP(4);
  // every piece (padding pieces included) has landed, every wave is done with strip and ring: LDS is free for the reduction / epilogue
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();

  PS_STAMP(5);
  // the two k-groups hold partial sums of the same outputs: group 0 finishes pixel fragments [0, PH), group 1 [PH, PT); each hands the
  // other's fragments over through LDS (one float4 per lane, fragment and channel tile: conflict-free 16-byte accesses)
  constexpr int PH = (PT + 1) / 2;
  const int b_lo = g == 0 ? 0 : PH, b_hi = g == 0 ? PH : PT;
  {
    float4* const sx = reinterpret_cast<float4*>(smem);
#pragma unroll
    for (int b = 0; b < PT; ++b)
      if (b < b_lo || b >= b_hi) {
#pragma unroll
        for (int c = 0; c < CT; ++c)
          sx[((wr * PT + b) * CT + c) * 64 + lane] = make_float4(acc[c][b][0], acc[c][b][1], acc[c][b][2], acc[c][b][3]);
      }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < PT; ++b)
      if (b >= b_lo && b < b_hi) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const float4 v = sx[((wr * PT + b) * CT + c) * 64 + lane];
          acc[c][b][0] += v.x; acc[c][b][1] += v.y; acc[c][b][2] += v.z; acc[c][b][3] += v.w;
        }
      }
    __syncthreads();
  }
  PS_STAMP(6);
  const ClassView cv = {};
  tile_epilogue<BM, BN, 8, MW, NV, PT, CT, false, BNEPI, KG>(acc, smem, m_end, m0, n0, tile_m, nullptr, Yv, ldy, accumulate, stat_sum, stat_sq, Kout, tid,
                                                           lane, wm, wn, cv, bnepi, tile_m, g, b_lo, b_hi);
  PS_STAMP(7);
}

// ---- host side ----------------------------------------------------------------------------------------------------
struct PsVariant { int kg, mw, nv, pt, ct; };
constexpr PsVariant kVariants[] = {{2, 2, 2, 11, 2}, {2, 1, 4, 11, 2}, {2, 4, 1, 6, 4}, {2, 2, 2, 6, 4}};   // 352 x 64, 176 x 128, 384 x 64, 192 x 128
constexpr int kNumVariants = 4;

struct PsPlan { int variant, bm, bn, tstride, ring, lines, wp, npieces, tiles; size_t lds; double eff; };

// strip lines a tile of bm pixels needs (max over the tiles of the layer, tile t = pixels [t * bm, (t + 1) * bm)): extended lines
// first .. last, plus one above and one below
int ps_lines(const yoloconv::Gather& g, int bm) {
  const int H = g.Ho, W = g.Wo, HW = H * W;
  static thread_local struct { int H, W, M, bm, lines; } memo[8] = {};
  static thread_local int memo_next = 0;
  for (const auto& e : memo)
    if (e.lines && e.H == H && e.W == W && e.M == g.M && e.bm == bm) return e.lines;
  int worst = 0;
  for (long m0 = 0; m0 < g.M; m0 += bm) {
    const long m1 = (m0 + bm < g.M ? m0 + bm : g.M) - 1;
    const long e0 = (m0 / HW) * (H + 1) + (m0 % HW) / W, e1 = (m1 / HW) * (H + 1) + (m1 % HW) / W;
    const int l = (int)(e1 - e0) + 3;
    worst = l > worst ? l : worst;
  }
  memo[memo_next] = {H, W, g.M, bm, worst};
  memo_next = (memo_next + 1) & 7;
  return worst;
}

bool ps_eligible(const yoloconv::Gather& g, int Kout, bool f32) {
  if (f32 || g.den != 1 || g.C0 != 0 || g.S != 3 || g.RS != 9 || g.smul != 1 || g.pad_h != 1 || g.pad_w != 1 || g.s2) return false;
  if (g.Hs != g.Ho || g.Ws != g.Wo || g.C1 % 64 != 0 || Kout % 64 != 0) return false;
  const size_t nimg = (size_t)g.M / ((size_t)g.Ho * g.Wo);
  if (nimg * g.Hs * g.Ws * g.C1 * 2 >= (1ull << 31) || (size_t)Kout * g.Kg * 2 >= (1ull << 31)) return false;
  return true;
}

bool ps_plan_variant(const yoloconv::Gather& g, int Kout, int v, PsPlan* out) {
  const PsVariant& V = kVariants[v];
  PsPlan p;
  p.variant = v;
  p.bm = V.mw * V.pt * 16;
  p.bn = V.nv * V.ct * 16;
  if (Kout % p.bn != 0) return false;
  p.wp = (g.Wo + 1 + 7) / 8 * 8;
  const int tn = Kout / p.bn;
  const int ntm_min = (g.M + p.bm - 1) / p.bm;
  const int rounds = (ntm_min * tn + 255) / 256;
  p.eff = (double)g.M * Kout / ((double)rounds * 256 * p.bm * p.bn);
  // pixels per tile: as many tiles as these rounds hold (shorter tiles -> fewer strip lines), whole image rows where that still fits
  int ntm = rounds * 256 / tn;
  ntm = ntm < ntm_min ? ntm_min : ntm;
  const int even = (g.M + ntm - 1) / ntm;
  const int cand[3] = {(even + g.Wo - 1) / g.Wo * g.Wo, even, p.bm};
  bool have = false;
  for (int c = 0; c < 3 && !have; ++c) {
    const int ts = cand[c];
    if (ts > p.bm || ts < 1) continue;
    if ((long)((g.M + ts - 1) / ts) * tn > (long)rounds * 256) continue;
    p.tstride = ts;
    p.lines = ps_lines(g, ts);
    p.npieces = 1 + p.lines * (p.wp / 8);
    if (p.npieces > 8 * PS_NPW) continue;
    const size_t strip = (size_t)p.npieces * 1024, out_tile = (size_t)p.bm * (p.bn * 2 + 16);
    const size_t xchg = (size_t)V.mw * V.nv * V.pt * V.ct * 1024;      // partial sums of the k-groups, exchanged after the K loop
    auto lds = [&](int ring) {
      size_t m = 2 * strip + (size_t)ring * p.bn * 128;
      m = m > out_tile ? m : out_tile;
      return m > xchg ? m : xchg;
    };
    p.ring = lds(4) <= 160 * 1024 ? 4 : 3;
    p.lds = lds(p.ring);
    if (p.lds > 160 * 1024) continue;
    have = true;
  }
  if (!have) return false;
  p.tiles = (g.M + p.tstride - 1) / p.tstride * tn;
  *out = p;
  return true;
}

}  // namespace

int g_ps_depth = 3;      // (unused; kept for the tuning name)
int g_pstrip = 0;         // "pstrip" tuning: -1 auto, 0 never (default while the kernel is being built up), 1 + v = force variant v where it fits

// plan for a problem: 0 = the big-tile kernel is not used, else the pixels per tile (the statistics / partial rows are ceil(M / that))
int yolo_pstrip_plan(const yoloconv::Gather& g, int Kout, bool f32, PsPlanOut* out) {
  if (g_pstrip == 0 || !ps_eligible(g, Kout, f32)) return 0;
  PsPlan best;
  bool have = false;
  // measured order of preference on MI355X (tools/probes/conv_one.py): 384 x 64 (4 channel tiles per wave: fewest LDS bytes per MFMA), then
  // 352 x 64, 192 x 128, 176 x 128
  static const int order[kNumVariants] = {2, 0, 3, 1};
  for (int o = 0; o < kNumVariants; ++o) {
    const int v = order[o];
    PsPlan p;
    if (g_pstrip > 0 && v != g_pstrip - 1) continue;
    if (!ps_plan_variant(g, Kout, v, &p)) continue;
    if (g_pstrip < 0 && (p.tiles > 256 || p.eff < 0.80)) continue;   // auto: one well-filled round of tiles only
    if (!have) { best = p; have = true; }
  }
  if (!have) return 0;
  if (g_pstrip < 0 && g.C1 < 128) return 0;            // auto: deep K loops only (>= 18 K steps per prologue / epilogue)
  if (out) { out->variant = best.variant; out->bm = best.bm; out->bn = best.bn; out->tstride = best.tstride; out->ring = best.ring; out->wp = best.wp; out->npieces = best.npieces; out->tiles = best.tiles; out->lds = best.lds; }
  return best.tstride;
}

namespace {

template <int KG, int MW, int NV, int PT, int CT, int RING, bool BNEPI>
int ps_launch_e(const yoloconv::Gather& g, const PsPlanOut& pl, const void* w, void* y, int ldy, int accumulate, const yoloconv::Epi& e, int Kout, hipStream_t st) {
  PStripArgs a;
  a.src = g.src1;
  a.N = (int)((size_t)g.M / ((size_t)g.Ho * g.Wo));
  a.src_bytes = (unsigned)((size_t)a.N * g.Hs * g.Ws * g.C1 * 2);
  a.wt = (const bf16_t*)w;
  a.wt_bytes = (unsigned)((size_t)Kout * g.Kg * 2);
  a.H = g.Ho; a.W = g.Wo; a.C = g.C1; a.M = g.M; a.Kg = g.Kg;
  a.dq8 = 8 / (pl.wp / 8); a.dr8 = 8 % (pl.wp / 8);
  a.tstride = pl.tstride; a.Wp = pl.wp; a.npieces = pl.npieces; a.strip_bytes = pl.npieces * 1024;
  a.rhw = g.rhw; a.rw = g.rw; a.rh1 = 1.0f / (float)(g.Ho + 1);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pstrip_kernel<KG, MW, NV, PT, CT, RING, BNEPI>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err != hipSuccess) { yolo_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(err)); return (int)err; }
    attr_set = true;
  }
  const int tn = Kout / pl.bn;
  hipLaunchKernelGGL((conv3x3_pstrip_kernel<KG, MW, NV, PT, CT, RING, BNEPI>), dim3(pl.tiles), dim3(512), pl.lds, st, a, y, ldy, accumulate, e.ssum, e.ssq, Kout, tn,
                     e.bn);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}

template <int KG, int MW, int NV, int PT, int CT>
int ps_launch_v(const yoloconv::Gather& g, const PsPlanOut& pl, const void* w, void* y, int ldy, int accumulate, const yoloconv::Epi& e, int Kout, hipStream_t st) {
  if (pl.ring == 4) {
    if (e.bn.y) return ps_launch_e<KG, MW, NV, PT, CT, 4, true>(g, pl, w, y, ldy, accumulate, e, Kout, st);
    return ps_launch_e<KG, MW, NV, PT, CT, 4, false>(g, pl, w, y, ldy, accumulate, e, Kout, st);
  }
  if (e.bn.y) return ps_launch_e<KG, MW, NV, PT, CT, 3, true>(g, pl, w, y, ldy, accumulate, e, Kout, st);
  return ps_launch_e<KG, MW, NV, PT, CT, 3, false>(g, pl, w, y, ldy, accumulate, e, Kout, st);
}

}  // namespace

int yolo_pstrip_launch(const yoloconv::Gather& g, const void* w, void* y, int ldy, int accumulate, const yoloconv::Epi& e, int Kout, hipStream_t st) {
  PsPlanOut pl;
  if (!yolo_pstrip_plan(g, Kout, false, &pl)) { yolo_set_error("%s:%d: no big-tile plan", __FILE__, __LINE__); return YOLO_ERR_INVALID_ARG; }
  switch (pl.variant) {
    case 0: return ps_launch_v<2, 2, 2, 11, 2>(g, pl, w, y, ldy, accumulate, e, Kout, st);
    case 1: return ps_launch_v<2, 1, 4, 11, 2>(g, pl, w, y, ldy, accumulate, e, Kout, st);
    case 2: return ps_launch_v<2, 4, 1, 6, 4>(g, pl, w, y, ldy, accumulate, e, Kout, st);
    default: return ps_launch_v<2, 2, 2, 6, 4>(g, pl, w, y, ldy, accumulate, e, Kout, st);
  }
}

#ifdef PS_STAMPS
extern "C" int yolo_debug_ps_stamps(void* buf) {
  unsigned long long* p = (unsigned long long*)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ps_stamps), &p, sizeof(p));
}
#endif
