// 3x3 / stride-1 / SAME convolution (forward and data gradient) on v_mfma_f32_32x32x16 with 64 x 64 wave tiles, for gfx950 (MI355X).
//
// Replaces keras.layers.Conv2D (reference backbone/basic_backbone.py:20-43 via resnet18.py:29-32, yolov3_detector.py:96-150) and its TF
// autodiff data gradient, like conv3x3_strip_kernel (conv_igemm.hip), whose outer structure it keeps: a workgroup owns BM consecutive
// pixels (linear NHW index) x BN output channels, loads the pixel strip [m0 - (W+1), m0 + BM + (W+1)) ONCE per 64-channel slice and reads
// the nine taps as nine shifted views of that LDS image; only the weight tile of a tap (BN x 64) streams through a 3-stage ring.
// What changed is what round 3 measured as the limiter of that kernel -- vector-instruction issue (profiles/HISTORY.md):
//  * v_mfma_f32_32x32x16 instead of 16x16x32: an MFMA holds its SIMD's vector issue for 8 of 32 cycles instead of 8 of 16, and one lane
//    address feeds twice the FLOPs;
//  * wave tile 64 pixels x 64 channels (2 x 2 blocks of 32 x 32): 16 ds_read_b128 per 16 MFMAs of 32 cycles per tap -- two thirds of the
//    LDS bytes per FLOP of the 32 x 64 tiles on 16x16x32;
//  * the nine (masked) tap addresses of a lane are computed ONCE per tile (they do not depend on the channel slice): per tap and pixel block
//    the loop issues the three XORs of the k-substeps and nothing else -- no v_cmp / v_cndmask, no row arithmetic;
//  * the weight fragment addresses are loop constants, the ring stage is an immediate offset (9 taps, 3 stages: the stage of a tap is static);
//  * optionally the K extent of a tap is split over WK wave groups (the k-substeps of a 64-channel slice are dealt to them; partial sums meet
//    in LDS once per tile): 8- or 4-block tiles keep the 64 x 64 wave tile where the layer is too small for 16-block tiles on 256 CUs;
//  * the epilogue stages the bf16 tile through LDS and walks it in whole NHWC rows -- stores, fan-in add, BatchNorm statistics or the fused
//    BatchNorm-backward reduce all happen there (the same arithmetic as conv_common.h tile_epilogue's BNEPI branch).
#include "conv_common.h"

namespace {
// Diagnostic builds only (make EXTRA_conv_s32="-fno-slp-vectorize -DS32_STAMPS"; tools/probes/s32_stamps.py): per wave, s_memtime cycles summed over
// all taps in {counted vmcnt wait, barrier, LDS-DMA issue, fragment reads + MFMAs} and the spans outside the tap loop.  In the product build
// no stamp executes and yolo_debug_s32_stamps does not exist.
#ifdef S32_STAMPS
__device__ unsigned long long* g_s32_stamps = nullptr;
#define S32_T(var)                                                                         \
  do {                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");          \
    __builtin_amdgcn_sched_barrier(0);                                                     \
  } while (0)
#define S32_ACC(sum, a_, b_) sum += (b_) - (a_)
#else
#define S32_T(var) do {} while (0)
#define S32_ACC(sum, a_, b_) do {} while (0)
#endif

struct S32Args {
  const bf16_t* src; unsigned src_bytes;   // NHWC activations (or dY for the stride-1 data gradient)
  const bf16_t* wt;  unsigned wt_bytes;    // [Kout][9][C]
  int H, W, C, M, Kg;                      // M = N*H*W, Kg = 9*C
  int E8;                                  // strip rows, multiple of 8 (>= BM + 2W + 2)
  float rhw, rw;
  int OH, OW;                              // MODE 1: the output (dX) grid, 2H x 2W
};

typedef __attribute__((address_space(3))) bf16x8_t s32_lds_frag_t;

// LDS images are [rows][64 x 16 bit] with the 16-byte chunk index XOR-ed with bits 1-3 of the ABSOLUTE LDS row (byte address >> 7): 32
// consecutive rows read with ds_read_b128 hit 16 distinct 16-byte slots in every lane group, at every row alignment
__device__ __forceinline__ int s32_sw(int row) { return (row >> 1) & 7; }

// bit `bit` of v ? a : b, as v_bfe_i32 + v_bfi_b32 (no VCC round trip)
__device__ __forceinline__ int s32_select_bit(unsigned v, int bit, int a, int b) {
  int d;
  asm("v_bfe_i32 %0, %1, %2, 1\n\tv_bfi_b32 %0, %0, %3, %4" : "=&v"(d) : "v"(v), "s"(bit), "v"(a), "v"(b));
  return d;
}

// MODE 1 = data gradient of a 3x3 / STRIDE-2 convolution (TF SAME on an even map: padding (0, 1)) as four parity classes of dX: class (ph, pw)
// holds dX[2h' + ph][2w' + pw] = sum over the taps r = ph (mod 2), s = pw (mod 2) of dY[h' - (r >> 1)][w' - (s >> 1)] W[r][s] -- a stride-1
// correlation over dY with the taps (dr, ds) in {-1, 0}^2 of which class (0,0) has 4, (0,1) and (1,0) 2, (1,1) 1 (9 in all: no multiply
// by a structural zero).  Here `a.src` is dY on its own H x W grid (= the class grid), the channel tiles of the launch are (class, block)
// pairs, a slice runs a fixed schedule of 4 K-step slots = the taps t in {0, 1, 3, 4} of the stride-1 numbering (so the address / mask
// table below is the stride-1 one), a workgroup SKIPS the slots its class does not have (their weight pieces are requested out of range),
// and the epilogue writes pixel (2h' + ph, 2w' + pw) of the OH x OW grid.  (reference: TF autodiff of the stride-2 Conv2D of
// backbone/resnet18.py:29-32 via basic_backbone.py:20-43; implicit-GEMM counterpart: conv_igemm.hip, ClassView.)
template <int WM, int WN, int WK, int PB, int CB, bool BNEPI, int MODE = 0>
__global__ __launch_bounds__(WM * WN * WK * 64) void conv3x3_s32_kernel(S32Args a, void* __restrict__ Yv, int ldy, int accumulate,
                                                                       float* __restrict__ stat_sum, float* __restrict__ stat_sq, int Kout,
                                                                       int tiles_n, BnEpi bn) {
  constexpr int NW = WM * WN * WK, NT = NW * 64;
  constexpr int BM = WM * PB * 32, BN = WN * CB * 32;
  constexpr int NTAP = MODE ? 4 : 9;             // K-step slots per 64-channel slice
  constexpr int WS = MODE ? 4 : 3;               // weight ring stages (NTAP is a multiple of WS: the stage of slot q is q % WS)
  constexpr int NS = 4 / WK;                     // k-substeps (16 channels each) of a tap that one wave computes
  constexpr int B_INSTR = BN / (8 * NW);         // weight LDS-DMA instructions per wave per tap (8 rows x 128 B each)
  constexpr int W_STAGE = BN * 128;
  static_assert(NW % 2 == 0 && B_INSTR >= 1 && BN % (8 * NW) == 0, "tile too small for the wave count");
  static_assert(WK == 1 || WK == 2 || WK == 4, "K split");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int ring0 = a.E8 * 128;
  const int zero0 = ring0 + WS * W_STAGE;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef S32_STAMPS
  unsigned long long T0 = 0, ta = 0, tb = 0, tc = 0, td = 0, te = 0, s_wait = 0, s_bar = 0, s_iss = 0, s_cmp = 0, s_slice = 0, T1 = 0, T2 = 0, T3 = 0;
#endif
  S32_T(T0);
  const int wk = wave % WK, wsp = wave / WK, wn = wsp % WN, wm = wsp / WN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = tile % tiles_n, tile_m = tile / tiles_n;
  const int nblk = MODE ? tiles_n >> 2 : tiles_n;                   // channel blocks per class
  const int cls = MODE ? tile_n / nblk : 0, ph = cls >> 1, pw = cls & 1;
  const int m0 = tile_m * BM, n0 = (MODE ? tile_n - cls * nblk : tile_n) * BN;
  // MODE 1, per slot q = (dr, ds) in order (-1,-1) (-1,0) (0,-1) (0,0): does this class have it, and which tap of the flipped [Cin][9][Cout]
  // weight tensor is it (kernel row r = ph ? 1 : (dr ? 2 : 0), flipped index 2 - r)
  bool vq[4] = {true, true, true, true};
  int wtq[4] = {0, 1, 3, 4};
  if constexpr (MODE == 1) {
    vq[0] = !ph && !pw; vq[1] = !ph; vq[2] = !pw;
    const int fr_m = ph ? 1 : 0, fr_0 = ph ? 1 : 2, fs_m = pw ? 1 : 0, fs_0 = pw ? 1 : 2;      // flipped row / column of dr (ds) = -1 and 0
    wtq[0] = 3 * fr_m + fs_m; wtq[1] = 3 * fr_m + fs_0; wtq[2] = 3 * fr_0 + fs_m; wtq[3] = 3 * fr_0 + fs_0;
  }
  auto wtap = [&](int q) { return MODE ? (q == 0 ? wtq[0] : (q == 1 ? wtq[1] : (q == 2 ? wtq[2] : wtq[3]))) : q; };      // (uniform selects)
  auto wvalid = [&](int q) { return MODE ? (q == 0 ? vq[0] : (q == 1 ? vq[1] : (q == 2 ? vq[2] : true))) : true; };
  if (tid < 8) *reinterpret_cast<uint4*>(smem + zero0 + tid * 16) = make_uint4(0u, 0u, 0u, 0u);

  // ---- LDS-DMA lane geometry: a piece is 8 rows x 128 bytes; lane -> row lane >> 3, 16-byte slot lane & 7, which holds chunk slot ^ sw(row)
  const int lrow = lane >> 3;
  // strip piece i = wave + k NW covers strip rows 8i .. 8i+7 = pixels m0 - (W+1) + 8i + lrow; sw = 4 (i & 1) + (lrow >> 1), i & 1 == wave & 1;
  // out-of-tensor pixels are out of the buffer range (negative offsets wrap above 2^31) and arrive as zeros
  const int strip_off0 = ((m0 - (a.W + 1) + wave * 8 + lrow) * a.C + (((lane & 7) ^ (4 * (wave & 1) + (lrow >> 1))) << 3)) * 2;
  const int strip_step = NW * 8 * a.C * 2;
  const int n_strip_instr = a.E8 >> 3;
  // weight piece q = wave B_INSTR + j covers ring rows 8q .. 8q+7 of a stage; ring row R holds output channel n0 + (R & ~31) + perm(R & 31),
  // perm(r) = 16 ((r >> 2) & 1) + 4 (r >> 3) + (r & 3): MFMA row r of a 32-channel block -- a lane's 16 accumulators are then 16 CONSECUTIVE channels
  unsigned wbase[B_INSTR];
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) {
    const int q = wave * B_INSTR + j, R = 8 * q + lrow, r = R & 31;
    const int ch = n0 + (R & ~31) + 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);
    const int sw = 4 * (((a.E8 >> 3) + q) & 1) + (lrow >> 1);
    wbase[j] = (unsigned)((ch * a.Kg + (((lane & 7) ^ sw) << 3)) * 2);
  }

  const int nchunk = a.C >> 6;
  const int nk = nchunk * NTAP;
  auto issue_weights = [&](int cc, int tap, int stage) {
    char* sB = smem + ring0 + stage * W_STAGE + wave * (B_INSTR * 1024);
    const unsigned koff = (unsigned)((wtap(tap) * a.C + cc * 64) * 2) | (wvalid(tap) ? 0u : 0x80000000u);
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) buffer_load_lds16(a.wt, a.wt_bytes, sB + j * 1024, wbase[j] + koff);
  };
  auto issue_strip = [&](int cc) {
    int off = strip_off0 + cc * 128;
    for (int i = wave; i < n_strip_instr; i += NW) {
      buffer_load_lds16(a.src, a.src_bytes, smem + i * 1024, (unsigned)off);
      off += strip_step;
    }
  };
  // prologue: the weight tiles of K-steps 0 and 1 and the first slice's strip are requested BEFORE the per-lane address tables below are
  // computed (two float-reciprocal divisions and 9 selects per pixel block: ~1.5 k cycles that the loads now fly under)
  issue_weights(0, 0, 0);
  if (nk > 1) issue_weights(0, 1, 1);
  issue_strip(0);
  __builtin_amdgcn_sched_barrier(0);

  // ---- fragment addresses.  32x32x16 operand layout: lane (r = lane & 31, kh = lane >> 5) holds row / column r, k = 8 kh .. 8 kh + 7
  const int p32 = lane & 31, kh = lane >> 5;
  const int s0 = wk * NS;                                           // this wave's first k-substep
  int tad[PB][9];                                                   // masked LDS address of tap t of this lane's pixel of block b, substep s0
#pragma unroll
  for (int b = 0; b < PB; ++b) {
    const int pl = (wm * PB + b) * 32 + p32, m = m0 + pl;
    int n_, rem, y_, x_;
    fast_divmod(min(m, a.M - 1), a.H * a.W, a.rhw, n_, rem);
    fast_divmod(rem, a.W, a.rw, y_, x_);
    // SAME padding, row wrap, image boundary inside the strip: bit t of ok = tap t reads a real pixel (sign-bit arithmetic)
    const int c0 = (int)((unsigned)(-x_) >> 31), c2 = (int)((unsigned)(x_ - (a.W - 1)) >> 31);
    const int cb = c0 | 2 | (c2 << 2);
    const int r0 = (-y_) >> 31, r2 = (y_ - (a.H - 1)) >> 31;
    const unsigned ok = (unsigned)(((cb & r0) | (cb << 3) | ((cb << 6) & r2)) & ((m - a.M) >> 31));
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (MODE == 1 && !(t == 0 || t == 1 || t == 3 || t == 4)) { tad[b][t] = 0; continue; }
      const int row = pl + (t / 3) * a.W + (t % 3);
      const int ad = row * 128 + (((2 * s0 + kh) ^ s32_sw(row)) << 4);
      tad[b][t] = s32_select_bit(ok, t, ad, zero0 + kh * 16);
    }
  }
  int wad[CB][NS];                                                  // weight fragment addresses in stage 0
#pragma unroll
  for (int c = 0; c < CB; ++c) {
    const int R = (wn * CB + c) * 32 + p32;
#pragma unroll
    for (int s = 0; s < NS; ++s) wad[c][s] = ring0 + R * 128 + (((2 * (s0 + s) + kh) ^ s32_sw(a.E8 + R)) << 4);
  }

  f32x16_t acc[CB][PB];
#pragma unroll
  for (int c = 0; c < CB; ++c)
#pragma unroll
    for (int b = 0; b < PB; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[c][b][i] = 0.f;

  // One tap = one K-step of 64 channels.  Hand-placed (stamps: profiles/HISTORY.md, round 4): what a wave issues besides its MFMAs sits BETWEEN
  // them -- an MFMA holds the vector issue for 8 of its 32 cycles -- instead of in front of them:
  //  * the pixel fragments of tap t + 1 do not depend on the barrier (the strip is resident for the whole slice): they are requested under
  //    the MFMAs of tap t, into the other of two register sets;
  //  * the LDS-DMA pieces of the weight tile of tap t + 2 (60-100 cycles of issue each) go one per MFMA group;
  //  * only the weight fragments are read behind the barrier.
  // The pieces are issued branch-free: past the last K-step their offset is out of range (zeros land in a stage nobody reads), so every
  // tap waits with the same counted vmcnt.
  bf16x8_t wf[CB][NS], pfa[PB][NS], pfb[PB][NS];
  int kk = 0;
  auto tap = [&](auto tc_, int cc, bf16x8_t (&cur)[PB][NS], bf16x8_t (&nxt)[PB][NS]) {
    constexpr int t = decltype(tc_)::value;                        // slot of the slice; its tap in the stride-1 numbering is GT(t)
    constexpr int so = (t % WS) * W_STAGE;
    constexpr int t2 = (t + 2) % NTAP, st2 = (t + 2) % WS;
    S32_T(ta);
    // weight stage kk (and, on the first tap of a slice, the strip) has landed; lgkmcnt(0): this wave's reads of the stage that is
    // refilled after this barrier have completed (an LDS-DMA write does not queue behind another wave's pending ds_read)
    if constexpr (t == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else                  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(B_INSTR) : "memory");
    S32_T(tb);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    S32_T(tc);
    S32_T(td);
    const unsigned koff = (unsigned)((wtap(t2) * a.C + (t + 2 >= NTAP ? cc + 1 : cc) * 64) * 2);
    const unsigned oob = (kk + 2 < nk && wvalid(t2)) ? 0u : 0x80000000u;
    char* const sB = smem + ring0 + st2 * W_STAGE + wave * (B_INSTR * 1024);
    if constexpr (t == 0) {
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int b = 0; b < PB; ++b) cur[b][s] = *reinterpret_cast<const s32_lds_frag_t*>(tad[b][0] ^ (s << 5));
    }
    if constexpr (MODE == 1) {
      if (!wvalid(t)) {                                 // (uniform) this class does not have the slot: keep the schedule, skip the work
        char* const sBs = smem + ring0 + st2 * W_STAGE + wave * (B_INSTR * 1024);
#pragma unroll
        for (int j = 0; j < B_INSTR; ++j) buffer_load_lds16(a.wt, a.wt_bytes, sBs + j * 1024, (wbase[j] + koff) | oob);
        if constexpr (t < NTAP - 1) {
          constexpr int tn_ = t + 1 < 2 ? t + 1 : t + 2;
#pragma unroll
          for (int r = 0; r < PB * NS; ++r) nxt[r % PB][r / PB] = *reinterpret_cast<const s32_lds_frag_t*>(tad[r % PB][tn_] ^ ((r / PB) << 5));
        }
        ++kk;
        return;
      }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int c = 0; c < CB; ++c) wf[c][s] = *reinterpret_cast<const s32_lds_frag_t*>(wad[c][s] + so);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int NM = NS * CB * PB;                  // MFMAs of this tap; slot i sits behind MFMA i
    constexpr int STEP = NM / B_INSTR;                // piece j sits in slot j * STEP, the next tap's PB * NS fragment reads in the other slots
    static_assert(NM % B_INSTR == 0 && NM - B_INSTR >= PB * NS, "interleave slots");
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      const int s = i / (CB * PB), c = (i / PB) % CB, b = i % PB;
      acc[c][b] = YOLO_MFMA_32x32x16(wf[c][s], cur[b][s], acc[c][b]);
      __builtin_amdgcn_sched_barrier(0);              // (the MFMA first: what follows issues under it)
      if (i % STEP == 0) {
        const int j = i / STEP;
        buffer_load_lds16(a.wt, a.wt_bytes, sB + j * 1024, (wbase[j] + koff) | oob);
      } else if (t < NTAP - 1) {
        constexpr int tq = t < NTAP - 1 ? t + 1 : NTAP - 1;      // (last slot: dead code, keeps the index in range for the compiler's bounds check)
        constexpr int tn = MODE ? (tq < 2 ? tq : tq + 1) : tq;
        const int r = i - (i / STEP + 1);             // fragment reads issued in earlier slots
        if (r < PB * NS) nxt[r % PB][r / PB] = *reinterpret_cast<const s32_lds_frag_t*>(tad[r % PB][tn] ^ ((r / PB) << 5));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    S32_T(te);
    S32_ACC(s_wait, ta, tb); S32_ACC(s_bar, tb, tc); S32_ACC(s_iss, tc, td); S32_ACC(s_cmp, td, te);
    ++kk;
  };

  S32_T(T1);
  for (int cc = 0; cc < nchunk; ++cc) {
    S32_T(ta);
    if (cc > 0) {                                       // every wave has finished reading the previous slice's strip (reads completed)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue_strip(cc);
    }
    S32_T(tb);
    S32_ACC(s_slice, ta, tb);
    tap(std::integral_constant<int, 0>{}, cc, pfa, pfb);
    tap(std::integral_constant<int, 1>{}, cc, pfb, pfa);
    tap(std::integral_constant<int, 2>{}, cc, pfa, pfb);
    tap(std::integral_constant<int, 3>{}, cc, pfb, pfa);
    if constexpr (MODE == 0) {
      tap(std::integral_constant<int, 4>{}, cc, pfa, pfb);
      tap(std::integral_constant<int, 5>{}, cc, pfb, pfa);
      tap(std::integral_constant<int, 6>{}, cc, pfa, pfb);
      tap(std::integral_constant<int, 7>{}, cc, pfb, pfa);
      tap(std::integral_constant<int, 8>{}, cc, pfa, pfb);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the out-of-range pieces of the last two taps: nothing may be in flight into LDS below)
  S32_T(T2);

  // ---- epilogue ------------------------------------------------------------------------------------------------------------------
  constexpr int OLD = BN * 2 + 16;                    // staged row stride in bytes (16-byte pad: conflict-free 16-byte writes)
  constexpr int CPR = BN / 8, RG = NT / CPR, ITER = (BM + RG - 1) / RG, NB = ITER <= 4 ? ITER : (ITER <= 8 ? 4 : (ITER + 1) / 2);
  static_assert(NT % CPR == 0, "row-walk geometry");
  const int e_ch = tid % CPR, e_rg = tid / CPR, e_c = n0 + e_ch * 8;
  if constexpr (MODE == 1) accumulate = accumulate == 2 ? (cls == 0 ? 1 : 0) : accumulate;     // 2: only the even / even class has a previous contribution
  const int prow = MODE ? cls * ((int)gridDim.x / tiles_n) + tile_m : tile_m;                  // partial row of the fused reduce: one per (class, pixel tile)
  bf16_t* const Y = reinterpret_cast<bf16_t*>(Yv);
  const bf16_t* const addp = bn.addend ? bn.addend : Y;
  constexpr int NBAT = (ITER + NB - 1) / NB;            // batches of NB rows per thread; the global reads of a batch are requested one batch ahead
  static_assert(NBAT <= 2, "two register sets of epilogue reads");
  uint4 e_yv[NBAT][NB], e_ev[NBAT][NB];
  unsigned e_mk[NBAT][NB], e_off[NBAT][NB];
  // global reads of rows it0 .. it0 + NB - 1 of this thread (the fused reduce's y and sign byte, the fan-in addend) into register set `set`
  auto e_load = [&](auto set_, int it0) {
    constexpr int set = decltype(set_)::value;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int row = e_rg + (it0 + k) * RG, m = m0 + row;
      e_off[set][k] = 0xffffffffu;                     // element offsets (the host refuses tensors of 2^31 elements or more)
      if (m < a.M && row < BM) {
        int mo = m;
        if constexpr (MODE == 1) {                     // pixel (2 y + ph, 2 x + pw) of the OH x OW grid
          int n_, rem, y_, x_;
          fast_divmod(m, a.H * a.W, a.rhw, n_, rem);
          fast_divmod(rem, a.W, a.rw, y_, x_);
          mo = (n_ * a.OH + 2 * y_ + ph) * a.OW + 2 * x_ + pw;
        }
        e_off[set][k] = (unsigned)mo * (unsigned)ldy + (unsigned)e_c;
        if constexpr (BNEPI) {
          e_yv[set][k] = *reinterpret_cast<const uint4*>(bn.y + e_off[set][k]);
          e_mk[set][k] = bn.mask ? (unsigned)bn.mask[e_off[set][k] >> 3] : 0xffu;
        }
        if (accumulate) e_ev[set][k] = *reinterpret_cast<const uint4*>(addp + e_off[set][k]);
      }
    }
  };
  e_load(std::integral_constant<int, 0>{}, 0);        // requested now: they fly under the staging of the tile
  __syncthreads();                                    // every wave is done with strip / ring
  if constexpr (WK > 1) {                             // partial sums of the k-groups meet in LDS: groups 1 .. WK-1 publish, group 0 adds
    float4* const red = reinterpret_cast<float4*>(smem);
    if (wk > 0) {
#pragma unroll
      for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int b = 0; b < PB; ++b)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            red[(((wsp * (WK - 1) + wk - 1) * (CB * PB) + c * PB + b) * 4 + q) * 64 + lane] =
                make_float4(acc[c][b][4 * q], acc[c][b][4 * q + 1], acc[c][b][4 * q + 2], acc[c][b][4 * q + 3]);
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int g = 0; g < WK - 1; ++g)
#pragma unroll
        for (int c = 0; c < CB; ++c)
#pragma unroll
          for (int b = 0; b < PB; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 v = red[(((wsp * (WK - 1) + g) * (CB * PB) + c * PB + b) * 4 + q) * 64 + lane];
              acc[c][b][4 * q] += v.x; acc[c][b][4 * q + 1] += v.y; acc[c][b][4 * q + 2] += v.z; acc[c][b][4 * q + 3] += v.w;
            }
    }
    __syncthreads();                                  // the partial sums have been read: the staged tile may overwrite them
  }
  if (wk == 0) {
    // accumulator i of lane (pixel, h) is MFMA row (i & 3) + 8 (i >> 2) + 4 h = channel 16 h + i of the block (the row permutation of the
    // weight image): 16 consecutive channels, two 16-byte writes into the staged tile
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      const int pl = (wm * PB + b) * 32 + p32;
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        const int cl = (wn * CB + c) * 32 + 16 * kh;
        uint4 o0, o1;
        o0.x = pack_bf2(acc[c][b][0], acc[c][b][1]);   o0.y = pack_bf2(acc[c][b][2], acc[c][b][3]);
        o0.z = pack_bf2(acc[c][b][4], acc[c][b][5]);   o0.w = pack_bf2(acc[c][b][6], acc[c][b][7]);
        o1.x = pack_bf2(acc[c][b][8], acc[c][b][9]);   o1.y = pack_bf2(acc[c][b][10], acc[c][b][11]);
        o1.z = pack_bf2(acc[c][b][12], acc[c][b][13]); o1.w = pack_bf2(acc[c][b][14], acc[c][b][15]);
        *reinterpret_cast<uint4*>(smem + pl * OLD + cl * 2) = o0;
        *reinterpret_cast<uint4*>(smem + pl * OLD + cl * 2 + 16) = o1;
      }
    }
  }
  __syncthreads();
  if constexpr (NBAT > 1) e_load(std::integral_constant<int, 1>{}, NB);       // (the accumulators are dead: registers to spare)

  // row walk: thread = (16-byte channel chunk, row group); whole 128 / 256-byte NHWC rows per store instruction
  const bool fwd_acc = !BNEPI && !stat_sum && bn.acc;
  const bool want_stats = !BNEPI && !accumulate && (stat_sum || fwd_acc);
  float mu[8], rs[8], s0_[8], s1_[8], s2_[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s0_[j] = s1_[j] = s2_[j] = 0.f; mu[j] = rs[j] = 0.f; }
  if constexpr (BNEPI) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { mu[j] = bn.mean[e_c + j]; rs[j] = bn.rstd[e_c + j]; }
  }
#pragma unroll
  for (int bt = 0; bt < NBAT; ++bt) {
    const int it0 = bt * NB;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      if (e_off[bt][k] != 0xffffffffu) {
        const int row = e_rg + (it0 + k) * RG;
        uint4 v = *reinterpret_cast<const uint4*>(smem + row * OLD + e_ch * 16);
        float g8[8], y8[8];
        if (accumulate) {                             // gradient fan-in: float32 add, one rounding
          unpack_bf8(v, g8);
          unpack_bf8(e_ev[bt][k], y8);
#pragma unroll
          for (int j = 0; j < 8; ++j) g8[j] += y8[j];
          v = pack_bf8(g8);
        }
        if constexpr (BNEPI) {                        // the unit's ReLU mask: one sign byte per 8-channel chunk
          unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int q = 0; q < 4; ++q)
            w4[q] = (((e_mk[bt][k] >> (2 * q)) & 1u) ? (w4[q] & 0xffffu) : 0u) | (((e_mk[bt][k] >> (2 * q + 1)) & 1u) ? (w4[q] & 0xffff0000u) : 0u);
          v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
        *reinterpret_cast<uint4*>(Y + e_off[bt][k]) = v;
        if constexpr (BNEPI) {
          unpack_bf8(v, g8);
          unpack_bf8(e_yv[bt][k], y8);
#pragma unroll
          for (int j = 0; j < 8; ++j) { s0_[j] += g8[j]; s1_[j] += g8[j] * ((y8[j] - mu[j]) * rs[j]); }
          if (bn.y2) {                                // shortcut BatchNorm of a down-sampling block: loaded in place
            unpack_bf8(*reinterpret_cast<const uint4*>(bn.y2 + e_off[bt][k]), y8);
#pragma unroll
            for (int j = 0; j < 8; ++j) s2_[j] += g8[j] * ((y8[j] - bn.mean2[e_c + j]) * bn.rstd2[e_c + j]);
          }
        } else if (want_stats) {                      // statistics of the values as stored (16-bit rounded)
          unpack_bf8(v, g8);
#pragma unroll
          for (int j = 0; j < 8; ++j) { s0_[j] += g8[j]; s1_[j] += g8[j] * g8[j]; }
        }
      }
    }
  }
  const int nq = BNEPI ? (bn.y2 ? 3 : 2) : (want_stats ? 2 : 0);
#ifdef S32_STAMPS
  S32_T(T3);
  if (g_s32_stamps && lane == 0) {
    unsigned long long* o = g_s32_stamps + ((size_t)blockIdx.x * NW + wave) * 16;
    o[0] = T0; o[1] = T1 - T0; o[2] = T2 - T1; o[3] = T3 - T2; o[4] = s_wait; o[5] = s_bar; o[6] = s_iss; o[7] = s_cmp; o[8] = s_slice; o[9] = T3;
  }
#endif
  if (nq == 0) return;
  float* const red = reinterpret_cast<float*>(smem);  // [RG][BN], one quantity at a time
  static_assert(RG * BN * 4 <= BM * OLD, "reduction scratch");
  const bool grouped = bn.group > 0 && (BNEPI ? bn.partial != nullptr : stat_sum != nullptr);       // two-level partial rows (conv_common.h rows_fold)
  const size_t rrow = grouped ? (size_t)yolo_row_groups(bn.rows, bn.group) + tile_m : (size_t)tile_m;
  for (int q = 0; q < nq; ++q) {
    __syncthreads();                                  // the staged tile (q = 0) / the previous quantity has been read
#pragma unroll
    for (int j = 0; j < 8; ++j) red[e_rg * BN + e_ch * 8 + j] = q == 0 ? s0_[j] : (q == 1 ? s1_[j] : s2_[j]);
    __syncthreads();
    for (int cl = tid; cl < BN; cl += NT) {
      float t = 0.f;
#pragma unroll 8
      for (int r = 0; r < RG; ++r) t += red[r * BN + cl];
      if constexpr (BNEPI) {
        if (grouped) row_store(bn.partial + (rrow * 3 + q) * ldy + n0 + cl, t);
        else if (bn.partial) bn.partial[((size_t)prow * 3 + q) * ldy + n0 + cl] = t;
        else yolo_acc_add(bn.acc, 3, ldy, tile_m % YOLO_ACC_NB, q, n0 + cl, t);
      } else {
        if (grouped) row_store((q == 0 ? stat_sum : stat_sq) + rrow * Kout + n0 + cl, t);
        else if (stat_sum) (q == 0 ? stat_sum : stat_sq)[(size_t)tile_m * Kout + n0 + cl] = t;
        else yolo_acc_add(bn.acc, 2, Kout, tile_m % YOLO_ACC_NB, q, n0 + cl, t);
      }
    }
  }
  if (grouped) {
    __syncthreads();
    if constexpr (BNEPI) rows_fold<NT>(bn.partial, bn.partial + ldy, bn.partial + 2 * (size_t)ldy, nq, (size_t)3 * ldy, bn.rows, bn.group, tile_m,
                                       tile_n, tiles_n, n0, BN, tid, reinterpret_cast<int*>(smem));
    else rows_fold<NT>(stat_sum, stat_sq, nullptr, 2, (size_t)Kout, bn.rows, bn.group, tile_m, tile_n, tiles_n, n0, BN, tid, reinterpret_cast<int*>(smem));
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
struct S32Cfg { int wm, wn, wk, pb, cb; };
// configuration ids (yolo_set_tuning "s32" = 1 + id forces one where it fits)
constexpr S32Cfg kCfg[] = {
    {2, 2, 1, 2, 2},   // 0: 128 x 128, 4 waves
    {4, 1, 1, 2, 2},   // 1: 256 x 64, 4 waves
    {2, 1, 2, 2, 2},   // 2: 128 x 64, 4 waves, K split 2
    {1, 2, 2, 2, 2},   // 3: 64 x 128, 4 waves, K split 2
    {1, 1, 4, 2, 2},   // 4: 64 x 64, 4 waves, K split 4
    {4, 2, 1, 2, 2},   // 5: 256 x 128, 8 waves
    {2, 2, 2, 2, 2},   // 6: 128 x 128, 8 waves, K split 2
    {4, 1, 2, 2, 2},   // 7: 256 x 64, 8 waves, K split 2
    {4, 2, 1, 3, 2},   // 8: 384 x 128, 8 waves (96 x 64 wave tiles)
};
constexpr int kNCfg = (int)(sizeof(kCfg) / sizeof(kCfg[0]));

size_t s32_lds(const S32Cfg& c, int W, bool s2 = false) {
  const int bm = c.wm * c.pb * 32, bn = c.wn * c.cb * 32;
  const size_t e8 = (size_t)(bm + 2 * W + 2 + 7) / 8 * 8;
  const size_t main_ = e8 * 128 + (s2 ? 4 : 3) * (size_t)bn * 128 + 128;
  const size_t out = (size_t)bm * (bn * 2 + 16);
  const size_t red = c.wk > 1 ? (size_t)c.wm * c.wn * (c.wk - 1) * c.pb * c.cb * 4096 : 0;
  size_t m = main_ > out ? main_ : out;
  return m > red ? m : red;
}

// the stride-2 data gradient as parity classes (dgrad_gather, conv_igemm.hip): 3x3, TF SAME on an even map (padding (0, 1): the class of even
// rows / columns has the taps r = 2, 0 reading dY rows h' - 1, h'; the odd one the tap r = 1 reading row h'), every class H/2 x W/2
bool s32_s2_eligible(const yoloconv::Gather& g, int Kout, bool f32) {
  if (f32 || !g.s2 || g.s2_ny != 4 || g.S != 3 || g.RS != 9 || g.C0 != 0 || g.C1 % 64 != 0 || Kout % 64 != 0) return false;
  if ((g.OH & 1) || (g.OW & 1) || g.Hs != g.OH / 2 || g.Ws != g.OW / 2) return false;
  for (int d = 0; d < 2; ++d) {
    const yoloconv::Gather::Dim* dm = d ? g.cold : g.rowd;
    const int size = d ? g.OW / 2 : g.OH / 2;
    if (dm[0].n != 2 || dm[0].pad != 1 || dm[0].t0 != 0 || dm[0].t1 != 2 || dm[0].size != size) return false;
    if (dm[1].n != 1 || dm[1].pad != 0 || dm[1].t0 != 1 || dm[1].size != size) return false;
  }
  const size_t nimg = (size_t)g.N;
  if (nimg * g.Hs * g.Ws * g.C1 * 2 >= (1ull << 31) || (size_t)Kout * g.Kg * 2 >= (1ull << 31) || nimg * g.OH * g.OW * Kout * 2 >= (1ull << 31)) return false;
  return true;
}

bool s32_eligible(const yoloconv::Gather& g, int Kout, bool f32) {
  if (f32 || g.den != 1 || g.C0 != 0 || g.S != 3 || g.RS != 9 || g.smul != 1 || g.pad_h != 1 || g.pad_w != 1 || g.s2) return false;
  if (g.Hs != g.Ho || g.Ws != g.Wo || g.C1 % 64 != 0 || Kout % 64 != 0) return false;
  const size_t nimg = (size_t)g.M / ((size_t)g.Ho * g.Wo);
  if (nimg * g.Hs * g.Ws * g.C1 * 2 >= (1ull << 31) || (size_t)Kout * g.Kg * 2 >= (1ull << 31) || (size_t)g.M * Kout * 2 >= (1ull << 31)) return false;
  return true;
}

template <int WM, int WN, int WK, int PB, int CB, bool BNEPI, int MODE>
int s32_launch_e(const yoloconv::Gather& g, const void* w, void* y, int ldy, int accumulate, const yoloconv::Epi& e, int Kout, hipStream_t st) {
  constexpr int BM = WM * PB * 32, BN = WN * CB * 32, NT = WM * WN * WK * 64;
  // MODE 1: the source is dY on its own grid (Hs x Ws = the class grid), g.M = pixels of one class, the output grid is OH x OW
  const int H = MODE ? g.Hs : g.Ho, W = MODE ? g.Ws : g.Wo;
  S32Args a;
  a.src = g.src1;
  a.src_bytes = (unsigned)((size_t)g.M * g.C1 * 2);
  a.wt = (const bf16_t*)w;
  a.wt_bytes = (unsigned)((size_t)Kout * g.Kg * 2);
  a.H = H; a.W = W; a.C = g.C1; a.M = g.M; a.Kg = g.Kg;
  a.E8 = (BM + 2 * W + 2 + 7) / 8 * 8;
  a.rhw = 1.0f / (float)(H * W); a.rw = 1.0f / (float)W;
  a.OH = g.OH; a.OW = g.OW;
  const S32Cfg c = {WM, WN, WK, PB, CB};
  const size_t lds = s32_lds(c, W, MODE == 1);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_s32_kernel<WM, WN, WK, PB, CB, BNEPI, MODE>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err != hipSuccess) { yolo_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(err)); return (int)err; }
    attr_set = true;
  }
  const int tiles_m = (g.M + BM - 1) / BM, tn = (MODE ? 4 : 1) * (Kout / BN);
  hipLaunchKernelGGL((conv3x3_s32_kernel<WM, WN, WK, PB, CB, BNEPI, MODE>), dim3(tiles_m * tn), dim3(NT), lds, st, a, y, ldy, accumulate, e.ssum,
                     e.ssq, Kout, tn, e.bn);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}
template <int WM, int WN, int WK, int PB, int CB>
int s32_launch_c(const yoloconv::Gather& g, const void* w, void* y, int ldy, int accumulate, const yoloconv::Epi& e, int Kout, hipStream_t st) {
  if (g.s2) {      // (the stride-2 classes: configurations 1 and 2 are instantiated)
    if constexpr ((WM == 4 && WN == 1 && WK == 1) || (WM == 2 && WN == 1 && WK == 2)) {
      if (e.bn.y) return s32_launch_e<WM, WN, WK, PB, CB, true, 1>(g, w, y, ldy, accumulate, e, Kout, st);
      return s32_launch_e<WM, WN, WK, PB, CB, false, 1>(g, w, y, ldy, accumulate, e, Kout, st);
    } else {
      yolo_set_error("%s:%d: no stride-2 instantiation of this s32 configuration", __FILE__, __LINE__);
      return YOLO_ERR_INVALID_ARG;
    }
  }
  if (e.bn.y) return s32_launch_e<WM, WN, WK, PB, CB, true, 0>(g, w, y, ldy, accumulate, e, Kout, st);
  return s32_launch_e<WM, WN, WK, PB, CB, false, 0>(g, w, y, ldy, accumulate, e, Kout, st);
}

}  // namespace

// "s32" tuning: -1 = automatic choice (default), 0 = never, 1 + id = force configuration id where it fits
int g_s32 = -1;

// 0 = this kernel does not take the problem, else the pixel tile (statistics / partial rows = ceil(M / that))
int g_s32_s2 = 1;       // "s32_s2" tuning: 1 = the stride-2 data gradient's parity classes on this kernel (default), 0 = on the implicit GEMM

int yolo_s32_plan(const yoloconv::Gather& g, int Kout, bool f32, S32PlanOut* out) {
  if (g.s2) {
    // 256 x 64 tiles (one class-pixel tile x one 64-channel block of one class), 128 x 64 with a K split where that leaves the grid short
    if (g_s32 == 0 || !g_s32_s2 || !s32_s2_eligible(g, Kout, f32)) return 0;
    const int id = g.M >= 16384 ? 1 : 2;
    const S32Cfg& c = kCfg[id];
    const int bm = c.wm * c.pb * 32, bn = c.wn * c.cb * 32;
    const size_t lds = s32_lds(c, g.Ws, true);
    if (lds > 160 * 1024) return 0;
    if (out) { out->id = id; out->bm = bm; out->bn = bn; out->tiles = (g.M + bm - 1) / bm * 4 * (Kout / bn); out->lds = lds; }
    return bm;
  }
  if (g_s32 == 0 || !s32_eligible(g, Kout, f32)) return 0;
  int id = -1;
  if (g_s32 > 0) id = g_s32 - 1;
  else {
    // automatic: where the kernel shortens the training STEP (kernel trace of the headline step with "s32" = -1 / 0, profiles/r04_s32_ab_*;
    // us in the step, this kernel / strip kernel).  Timed alone in a loop (tools/probes/s32_sweep.py) it also wins 5-30 % on the 26 x 26 and
    // 13 x 13 layers, but in the step those launches take the same time on either kernel (27.6 / 27.6, 33.4 / 33.0, 30.6 / 29.1), and in the backward
    // pass, beside the weight-gradient stream, neither kernel runs at its stand-alone speed -- so the rule is the measured one, not the sweep's:
    //  * 40-79 columns (52 x 52 x 128 -> 128: 28.2 / 32.9; 128 -> 256: 50.1 / 55.6): 256 x 64 tiles, forward and data gradient;
    //  * 20-39 columns: forward launches onto >= 512 channels only (26 x 26 x 256 -> 512: 46.2 / 49.9), 256 x 64 tiles;
    //  * everything else stays on the strip / streaming kernels (80 columns and more: 9 taps in all against ~8 k cycles of setup + epilogue
    //    per tile, 48 / 30 us alone on 104 x 104 x 64).
    if (g.M < 2048 || g.Wo >= 80 || g.Wo < 20) return 0;
    if (g.Wo >= 40) id = 1;
    else if (g.role == 0 && !g.bnepi && Kout >= 512) id = 1;
    else return 0;
  }
  if (id < 0 || id >= kNCfg) return 0;
  const S32Cfg& c = kCfg[id];
  const int bm = c.wm * c.pb * 32, bn = c.wn * c.cb * 32;
  if (Kout % bn != 0) return 0;
  const size_t lds = s32_lds(c, g.Wo);
  if (lds > 160 * 1024) return 0;
  if (out) { out->id = id; out->bm = bm; out->bn = bn; out->tiles = (g.M + bm - 1) / bm * (Kout / bn); out->lds = lds; }
  return bm;
}

#ifdef S32_STAMPS
extern "C" int yolo_debug_s32_stamps(void* buf) {
  unsigned long long* p = (unsigned long long*)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_s32_stamps), &p, sizeof(p));
}
#endif

int yolo_s32_launch(const yoloconv::Gather& g, const void* w, void* y, int ldy, int accumulate, const yoloconv::Epi& e, int Kout, hipStream_t st) {
  S32PlanOut pl;
  if (!yolo_s32_plan(g, Kout, false, &pl) || ldy != Kout) { yolo_set_error("%s:%d: no s32 plan", __FILE__, __LINE__); return YOLO_ERR_INVALID_ARG; }
  switch (pl.id) {
    case 0: return s32_launch_c<2, 2, 1, 2, 2>(g, w, y, ldy, accumulate, e, Kout, st);
    case 1: return s32_launch_c<4, 1, 1, 2, 2>(g, w, y, ldy, accumulate, e, Kout, st);
    case 2: return s32_launch_c<2, 1, 2, 2, 2>(g, w, y, ldy, accumulate, e, Kout, st);
    case 3: return s32_launch_c<1, 2, 2, 2, 2>(g, w, y, ldy, accumulate, e, Kout, st);
    case 4: return s32_launch_c<1, 1, 4, 2, 2>(g, w, y, ldy, accumulate, e, Kout, st);
    case 5: return s32_launch_c<4, 2, 1, 2, 2>(g, w, y, ldy, accumulate, e, Kout, st);
    case 6: return s32_launch_c<2, 2, 2, 2, 2>(g, w, y, ldy, accumulate, e, Kout, st);
    case 7: return s32_launch_c<4, 1, 2, 2, 2>(g, w, y, ldy, accumulate, e, Kout, st);
    case 8: return s32_launch_c<4, 2, 1, 3, 2>(g, w, y, ldy, accumulate, e, Kout, st);
  }
  yolo_set_error("%s:%d: bad s32 configuration", __FILE__, __LINE__);
  return YOLO_ERR_INVALID_ARG;
}
