// Weight gradient of 3x3 / stride-1 / SAME convolutions with the OUTPUT tile stationary in registers, for gfx950 (MI355X).
//
// Replaces TF autodiff of keras.layers.Conv2D w.r.t. its kernel (reference backbone/basic_backbone.py:42 via resnet18.py:29-32,
// yolov3_detector.py:96-150), like wgrad3x3_strip_kernel (conv_igemm.hip), whose stamps (profiles/r04_wgrad_stamps.txt) show a loop that is
// not waiting for memory: per 64-pixel stage a wave spends ~1750 cycles in the stage body and ~300 at the barrier for 384 cycles of MFMA, the
// matrix pipe is ~70 % busy at a shader clock of ~1.35 GHz, and every workgroup pays setup + a 96 KB slab store for 22 stages.  What costs there
// is what a wave issues per MFMA (14 transposed reads per 12 MFMAs, per-lane tap-validity selects, cursor arithmetic) and the bytes moved per
// FLOP (the X strip of a stage is re-fetched with its halo for every kernel row and output tile).  Here:
//  * a workgroup owns dW[64 co][9 taps][64 ci] = 36 blocks of 32 x 32 (v_mfma_f32_32x32x16) for a RANGE of pixels and keeps all of it in
//    registers: 8 waves = 2 K-groups x (2 co blocks x 2 ci blocks), 9 blocks (144 accumulator registers) per wave; one workgroup per CU;
//  * pixels run in PADDED coordinates (one zero column behind every image row, one zero row behind every image): every tap of every position
//    is then a constant offset -- no tap masks, no validity selects; the pad positions are zeros in both operands and add nothing
//    (K grows by (H+1)(W+1)/(HW): 1.9 % on 104 x 104, 3.9 % on 52 x 52, 7.8 % on 26 x 26);
//  * X streams ONCE through a ring of LDS rows (position -> row, 128 bytes = the 64 channels of the unit), dY through a 3-stage ring of
//    64-position stages; both arrive by LDS-DMA through buffer descriptors (pad / out-of-tensor positions: out-of-range offsets -> zeros),
//    two stages ahead of their use, one barrier per stage.  L2 -> LDS bytes per FLOP: 1/288 (strip kernel: 1/126);
//  * operands are read transposed (ds_read_b64_tr_b16: pixel-major memory, pixels are the K index): per 16 positions a wave reads the dY^T
//    fragment (2 reads) and, per kernel row, 12 consecutive positions of its 32 ci (3 reads); the three tap columns are that register window
//    shifted by 0 / 1 / 2 positions -- 4 v_alignbit_b32 for the odd one -- so 11 reads feed 9 MFMAs (strip kernel: 14 reads per 12 half-size MFMAs);
//    64-byte-half swizzle by bit 1 of the row: conflict-free at every row alignment (brute-forced over the lane groups);
//  * the two K-groups take alternate 16-position steps of a stage and meet once, in LDS, before the slab store.
// Splits over pixel ranges store their partial tiles as slabs (plain stores) that the bucket's summing launch adds, like the strip kernel.
#include "conv_common.h"

namespace {

struct Wg9Args {
  const bf16_t* x;  unsigned x_bytes;     // [N*H*W][Cin]
  const bf16_t* dy; unsigned y_bytes;     // [N*H*W][Cout]
  int H, W, Hp, Wp, N, Cin, Cout;
  int Q;                                  // padded positions N*Hp*Wp
  int R;                                  // ring rows (multiple of 8) = 192 + 2 D
  int D;                                  // X lead in positions: roundup8(Wp + 1)
  int units_ci, units;                    // Cin / 64, (Cin / 64) * (Cout / 64)
  int sps, nstages;                       // stages (64 positions) per split, stages in all
  int dn, dh, dw, m64;                    // 64 positions = dn images + dh rows + dw columns of the padded grid; the same step in pixel indices
  float rhwp, rwp;                        // 1 / (Hp*Wp), 1 / Wp
  long long slab;                         // floats per slab (Cout * 9 * Cin); split z stores to out + z * slab
};

typedef s16x4_t __attribute__((address_space(3))) * w9_lds_tr_t;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"      // (host pass only: LDS addresses are 32-bit on the device)
__device__ __forceinline__ s16x4_t w9_tr(int addr) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((w9_lds_tr_t)addr); }
#pragma clang diagnostic pop
typedef short w9_s16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned w9_u32x2_t __attribute__((ext_vector_type(2)));

// (hi:lo) >> 16: elements 1, 2 of the three 16-bit elements starting at lo's low half
__device__ __forceinline__ unsigned w9_align(unsigned hi, unsigned lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }

__device__ __forceinline__ bf16x8_t w9_frag(unsigned a, unsigned b, unsigned c, unsigned d) {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  const u32x4_t v = {a, b, c, d};
  return __builtin_bit_cast(bf16x8_t, v);
}

constexpr int W9_YSTAGE = 64 * 128;          // bytes of one dY stage: 64 positions x 64 co
constexpr int W9_NT = 512;

// LDS-DMA of 16 bytes per lane (one 1 KiB piece per wave) issued from inline asm: hipcc's waitcnt pass treats ds_read_b64_tr_b16 as a possible reader
// of any LDS-DMA it has seen and drains vmcnt(0) in front of it (found in this kernel's first build: every stage waited out the pieces it had just
// requested); an asm statement is opaque to that pass, the counted s_waitcnt vmcnt of the stage loop is the only ordering -- as intended.
// lds = wave-uniform LDS byte address (M0 is written in the same statement that uses it), voff = per-lane byte offset (out of range: zeros).
__device__ __forceinline__ void w9_dma(__amdgpu_buffer_rsrc_t r, int lds, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds), "v"(voff), "s"(r) : "memory");
}

__global__ __launch_bounds__(512) void wgrad9_kernel(Wg9Args a, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // (the only LDS object: byte address 0)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave >> 2, cb = wave & 1, cib = (wave >> 1) & 1;
  const int l = xcd_remap(blockIdx.x, gridDim.x);
  const int unit = l % a.units, split = l / a.units;       // the units of one pixel range sit on one XCD: X and dY are fetched once per L2
  const int ci0 = (unit % a.units_ci) * 64, co0 = (unit / a.units_ci) * 64;
  const int s_begin = split * a.sps;
  const int nst = min(a.nstages, s_begin + a.sps) - s_begin;
  if (nst <= 0) return;
  const int Q0 = s_begin * 64;
  const int RB = a.R * 128;                                 // ring bytes; rows [R, R + 8) mirror rows [0, 8)
  const int YB = RB + 1024;                                 // dY stages
  const int hpwp = a.Hp * a.Wp;
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.dy), 0, (int)a.y_bytes, 0x00020000);

  // ---- LDS-DMA geometry: a piece is 8 positions x 128 bytes; lane -> position lane >> 3, 16-byte slot lane & 7 holding chunk slot ^ 4 bit1(row)
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ (((lrow >> 1) & 1) << 2);
  const unsigned xcol = (unsigned)((ci0 + lchunk * 8) * 2), ycol = (unsigned)((co0 + lchunk * 8) * 2);
  // a piece's positions are consecutive from a wave-uniform base: the base's padded coordinates live in scalar registers, the lane adds its row
  struct Cur { int x, y, n, m; };                           // column, row, image of the base position in the padded grid; its pixel index
  auto decode = [&](int pos) {                              // pos >= -hpwp
    Cur c;
    int rem;
    fast_divmod(pos + hpwp, hpwp, a.rhwp, c.n, rem);
    c.n -= 1;
    fast_divmod(rem, a.Wp, a.rwp, c.y, c.x);
    c.m = (c.n * a.H + c.y) * a.W + c.x;
    c.x = __builtin_amdgcn_readfirstlane(c.x); c.y = __builtin_amdgcn_readfirstlane(c.y);
    c.n = __builtin_amdgcn_readfirstlane(c.n); c.m = __builtin_amdgcn_readfirstlane(c.m);
    return c;
  };
  auto advance = [&](Cur& c) {                              // + 64 positions (scalar arithmetic)
    c.x += a.dw;
    const int cx = c.x >= a.Wp ? 1 : 0;
    c.x -= cx ? a.Wp : 0;
    c.y += a.dh + cx;
    const int cy = c.y >= a.Hp ? 1 : 0;
    c.y -= cy ? a.Hp : 0;
    c.n += a.dn + cy;
    c.m += a.m64 - cx - (cy ? a.W : 0);
  };
  constexpr unsigned OOB = 0x80000000u;
  // byte offset of this lane's 16 bytes of the piece at base c in a tensor with rowbytes bytes per pixel (pad / outside: out of range); branch-free
  auto lane_off = [&](const Cur& c, unsigned rowbytes, unsigned col) {
    int x = c.x + lrow;
    const int c1 = x >= a.Wp ? 1 : 0;
    x -= c1 ? a.Wp : 0;
    int y = c.y + c1;
    const int c2 = y >= a.Hp ? 1 : 0;
    y -= c2 ? a.Hp : 0;
    const int n = c.n + c2;
    const int m = c.m + lrow - c1 - (c2 ? a.W : 0);
    const bool ok = (x < a.W) & (y < a.H) & ((unsigned)n < (unsigned)a.N);
    const unsigned off = __umul24((unsigned)m, rowbytes) + col;          // (m < 2^24, rowbytes < 2^24: full-rate 24-bit multiply; garbage when !ok)
    return ok ? off : OOB;
  };
  const unsigned xrowb = (unsigned)(a.Cin * 2), yrowb = (unsigned)(a.Cout * 2);
  // Most pieces lie inside one image row, clear of the pad column: then the lane's offset is a scalar (the base pixel) times the row size plus a
  // loop-invariant per-lane constant -- one VALU add instead of the ~22 of lane_off.  The condition is wave-uniform (the cursor lives in scalar
  // registers).
  const unsigned xlane = (unsigned)lrow * xrowb + xcol, ylane = (unsigned)lrow * yrowb + ycol;
  auto piece_off = [&](const Cur& c, unsigned rowbytes, unsigned col, unsigned lane_const) {
    if ((c.x + 7 < a.W) & (c.y < a.H) & ((unsigned)c.n < (unsigned)a.N)) return (unsigned)c.m * rowbytes + lane_const;
    return lane_off(c, rowbytes, col);
  };

  // ---- prologue: X pieces [0, (2 D + 128) / 8) = positions [Q0 - D, Q0 + D + 128) (they do not wrap: 2 D + 128 <= R), dY stages 0 and 1
  const int npro = (2 * a.D + 128) >> 3;
  for (int pi = wave; pi < npro; pi += 8) {
    const Cur c = decode(Q0 - a.D + 8 * pi);
    const unsigned off = lane_off(c, xrowb, xcol);
    w9_dma(rX, pi * 1024, off);
    if (pi == 0) w9_dma(rX, RB, off);                       // rows [R, R + 8) mirror rows [0, 8)
  }
  // steady state: wave w issues the X piece at positions q_s + D + 128 + 8 w and dY piece w of stage s + 2
  Cur cx = decode(Q0 + a.D + 128 + 8 * wave), cy = decode(Q0 + 8 * wave);
  w9_dma(rY, YB + wave * 1024, lane_off(cy, yrowb, ycol));
  advance(cy);
  w9_dma(rY, YB + W9_YSTAGE + wave * 1024, nst > 1 ? lane_off(cy, yrowb, ycol) : OOB);
  advance(cy);
  int xrow = 2 * a.D + 128 + 8 * wave;                      // ring row of this wave's next X piece
  xrow -= xrow >= a.R ? a.R : 0;

  // ---- transposed-read geometry: lane (g = lane >> 4, i = lane & 15): 16-column half g & 1 of the 32-column block, k half g >> 1,
  // row q = i >> 2 of a 4-row block, 8-byte column group p = i & 3
  const int g4 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int kh = g4 >> 1;
  const int colb = (g4 & 1) * 32 + p4 * 8;
  // dY^T fragment of k-step k of a stage: rows 16 k + 8 kh + 4 rd + q, block cb; bit1(row) = bit1(q)
  const int ya = YB + (8 * kh + q4) * 128 + ((cb ^ ((q4 >> 1) & 1)) << 6) + colb + kg * 2048;       // (this K-group's first step k = kg: + 16 rows)
  // X rows of kernel row tr: ring row of position q_s + 16 k + 8 kh + 4 rd + q + (tr - 1) Wp - 1; per lane and kernel row the WRAPPED byte
  // address of (rd = 0, this K-group's first step), advanced by 64 rows per stage; the 64-byte half is swizzled by bit 1 of the row, which
  // neither the stage advance nor rd / kh / k change
  int xa[3];
#pragma unroll
  for (int tr = 0; tr < 3; ++tr) {
    int row = a.D + (tr - 1) * a.Wp - 1 + 16 * kg + 8 * kh + q4;           // relative to the ring origin (position Q0 - D): >= 0, < R
    xa[tr] = row * 128 + ((cib ^ ((row >> 1) & 1)) << 6) + colb;
  }

  f32x16_t acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // raw operands of one 16-position step: dY^T (2 reads) and, per kernel row, 12 positions (3 reads)
  struct Raw { s16x4_t y0, y1, x[3][3]; };
  auto load_raw = [&](Raw& r, int yaddr, const int (&xad)[3]) {
    r.y0 = w9_tr(yaddr);
    r.y1 = w9_tr(yaddr + 512);
#pragma unroll
    for (int tr = 0; tr < 3; ++tr) {
      r.x[tr][0] = w9_tr(xad[tr]);
      r.x[tr][1] = w9_tr(xad[tr] + 512);
      r.x[tr][2] = w9_tr(xad[tr] + 1024);
    }
  };
  auto wrap = [&](int v) { const unsigned x = (unsigned)v, y = x - (unsigned)RB; return (int)(x < y ? x : y); };   // v in [0, 2 RB): v mod RB
  auto frag_y = [&](const Raw& r) {
    w9_s16x8_t yv = {r.y0[0], r.y0[1], r.y0[2], r.y0[3], r.y1[0], r.y1[1], r.y1[2], r.y1[3]};
    return __builtin_bit_cast(bf16x8_t, yv);
  };
  // the three MFMAs of kernel row tr: positions 0 .. 11 of the row are the dwords d0.x d0.y d1.x d1.y d2.x d2.y; tap column ts = the window of 8
  // positions that starts at position ts
  auto mfma_row = [&](const bf16x8_t& af, const Raw& r, int tr) {
    const w9_u32x2_t d0 = __builtin_bit_cast(w9_u32x2_t, r.x[tr][0]), d1 = __builtin_bit_cast(w9_u32x2_t, r.x[tr][1]),
                     d2 = __builtin_bit_cast(w9_u32x2_t, r.x[tr][2]);
    acc[tr * 3 + 0] = YOLO_MFMA_32x32x16(af, w9_frag(d0[0], d0[1], d1[0], d1[1]), acc[tr * 3 + 0]);
    acc[tr * 3 + 2] = YOLO_MFMA_32x32x16(af, w9_frag(d0[1], d1[0], d1[1], d2[0]), acc[tr * 3 + 2]);
    acc[tr * 3 + 1] = YOLO_MFMA_32x32x16(af, w9_frag(w9_align(d0[1], d0[0]), w9_align(d1[0], d0[1]), w9_align(d1[1], d1[0]), w9_align(d2[0], d1[1])),
                                         acc[tr * 3 + 1]);
  };

  // ---- stage loop: K-group kg takes the 16-position steps k = kg and kg + 2 of every stage.  Hand-placed: all 22 reads of the stage first, the
  // two LDS-DMA pieces of stage s + 2 and the address arithmetic of the next stage between the MFMAs
  int ystage = 0;
  for (int s = 0; s < nst; ++s) {
    // stage s (dY) and the X pieces up to its halo have landed: everything but the 2 pieces issued during stage s - 1 (the X mirror piece, when
    // there is one, only makes this wait for one more); lgkmcnt(0): this wave's reads of what the pieces below overwrite have completed
    // (s == 0: the last TWO loads issued are dY stage 0 and dY stage 1 -- only stage 1 may still fly)
    if (s + 1 >= nst) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if (s == 0)  asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    else              asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int yb = ya + ystage * W9_YSTAGE;
    Raw r0, r1;
    load_raw(r0, yb, xa);
    int xb[3];
#pragma unroll
    for (int tr = 0; tr < 3; ++tr) xb[tr] = wrap(xa[tr] + 32 * 128);      // this K-group's second step: 32 positions on
    load_raw(r1, yb + 4096, xb);
    __builtin_amdgcn_sched_barrier(0);
    const bool more = s + 2 < nst;
    const unsigned offx = more ? piece_off(cx, xrowb, xcol, xlane) : OOB;          // (past the range: zeros into rows nobody reads; same vmcnt arithmetic)
    const bf16x8_t a0 = frag_y(r0);
    mfma_row(a0, r0, 0);
    __builtin_amdgcn_sched_barrier(0);
    w9_dma(rX, xrow * 128, offx);
    if (xrow == 0) w9_dma(rX, RB, offx);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned offy = more ? piece_off(cy, yrowb, ycol, ylane) : OOB;
    mfma_row(a0, r0, 1);
    __builtin_amdgcn_sched_barrier(0);
    {
      int st2 = ystage + 2; st2 -= st2 >= 3 ? 3 : 0;
      w9_dma(rY, YB + st2 * W9_YSTAGE + wave * 1024, offy);
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma_row(a0, r0, 2);
    advance(cx);
    advance(cy);
    xrow += 64; xrow -= xrow >= a.R ? a.R : 0;
    const bf16x8_t a1 = frag_y(r1);
    mfma_row(a1, r1, 0);
#pragma unroll
    for (int tr = 0; tr < 3; ++tr) xa[tr] = wrap(xa[tr] + 64 * 128);      // next stage
    mfma_row(a1, r1, 1);
    mfma_row(a1, r1, 2);
    ystage = ystage == 2 ? 0 : ystage + 1;
  }

  // ---- epilogue: the two K-groups meet in LDS (the rings are dead) and the sums are laid out there as the slab itself, [64 co][9 taps][64 ci]
  // float32, so that all 512 threads store it in 16-byte pieces of whole 256-byte (tap, ci) rows.  D[row = co][col = ci]: a lane holds column
  // lane & 31 and rows (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
  __syncthreads();
  float4* const red = reinterpret_cast<float4*>(smem);
  float* const tile = reinterpret_cast<float*>(smem);
  const int wq = wave & 3;
  if (kg == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        red[((wq * 9 + t) * 4 + q) * 64 + lane] = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
  }
  __syncthreads();
  if (kg == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = red[((wq * 9 + t) * 4 + q) * 64 + lane];
        acc[t][4 * q] += v.x; acc[t][4 * q + 1] += v.y; acc[t][4 * q + 2] += v.z; acc[t][4 * q + 3] += v.w;
      }
  }
  __syncthreads();                                          // every partial sum has been read: the tile may overwrite them
  if (kg == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = 32 * cb + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
        tile[(co * 9 + t) * 64 + 32 * cib + (lane & 31)] = acc[t][i];
      }
  }
  __syncthreads();
  {
    float* const dst = out + (a.slab ? (size_t)split * (size_t)a.slab : 0);
    const int Kg = 9 * a.Cin;
    // 64 x 9 rows of 64 floats = 16 float4 each: 9216 float4 over 512 threads
    for (int e = tid; e < 64 * 9 * 16; e += W9_NT) {
      const int rowi = e >> 4, c4 = e & 15;
      const int co = rowi / 9, t = rowi - co * 9;
      const float4 v = *reinterpret_cast<const float4*>(tile + rowi * 64 + c4 * 4);
      float* d = dst + (size_t)(co0 + co) * Kg + t * a.Cin + ci0 + c4 * 4;
      if (a.slab) *reinterpret_cast<float4*>(d) = v;
      else { atomicAdd(d, v.x); atomicAdd(d + 1, v.y); atomicAdd(d + 2, v.z); atomicAdd(d + 3, v.w); }
    }
  }
}

}  // namespace

// "wgrad9" tuning: -1 automatic (default), 0 never, 1 wherever it fits
int g_wgrad9 = -1;
// "wgrad9_wgs": workgroups (= CUs: one 512-thread workgroup owns a CU) a launch aims at.  The kernel runs on the weight-gradient stream beside the
// main stream's data gradients and BatchNorm kernels, which are the step's critical path: a grid below 256 leaves whole CUs to them
int g_wgrad9_wgs = 128;     // (in-step sweep, profiles/r04_wgrad9_wgs_sweep.txt: 256 -> 7.71 k images/s, 192 -> 8.00, 160 -> 8.04, 128 -> 8.08, 96 -> 8.08, 64 -> 7.60; strip kernel 7.78)

// 0 = this kernel does not take the problem, else the number of pixel splits (slabs); *sps_out = stages per split
int yolo_wgrad9_plan(const yolo_conv_problem* p, int* sps_out) {
  if (g_wgrad9 == 0) return 0;
  if (!(p->R == 3 && p->S == 3 && p->stride == 1 && p->pad_t == 1 && p->pad_l == 1 && p->Ho == p->H && p->Wo == p->W && p->C0 == 0)) return 0;
  if (p->Cin % 64 != 0 || p->Cout % 64 != 0) return 0;
  const long Hp = p->H + 1, Wp = p->W + 1, Q = (long)p->N * Hp * Wp;
  if (p->W < 7 || Hp * Wp < (Wp + 1 + 7) / 8 * 8) return 0;       // (one column wrap per 8-position piece; the first piece's positions decode from >= -Hp*Wp)
  if (Q + Hp * Wp + 4096 >= (1 << 24) || (size_t)p->N * p->H * p->W * p->Cin * 2 >= (1ull << 31) || (size_t)p->N * p->H * p->W * p->Cout * 2 >= (1ull << 31)) return 0;
  const int D = (int)((Wp + 1 + 7) / 8 * 8), R = 192 + 2 * D;
  if ((size_t)(R + 8) * 128 + 3 * 8192 > 160 * 1024) return 0;
  if (g_wgrad9 < 0) {
    // automatic: the padded grid costs (H+1)(W+1)/(HW) of K; below 20 columns (13 x 13: +16 %) the strip kernel keeps the layer
    if (p->W < 20 || p->H < 20) return 0;
  }
  const int units = (p->Cin / 64) * (p->Cout / 64);
  const int nstages = (int)((Q + 63) / 64);
  int nsplit = units >= g_wgrad9_wgs ? 1 : g_wgrad9_wgs / units;
  if (nsplit > nstages) nsplit = nstages;
  const int sps = (nstages + nsplit - 1) / nsplit;
  nsplit = (nstages + sps - 1) / sps;
  if (sps_out) *sps_out = sps;
  return nsplit;
}

int yolo_wgrad9_launch(const yolo_conv_problem* p, const void* x, const void* dy, float* out, long long slab, hipStream_t stream) {
  int sps = 0;
  const int nsplit = yolo_wgrad9_plan(p, &sps);
  if (!nsplit) { yolo_set_error("%s:%d: no wgrad9 plan", __FILE__, __LINE__); return YOLO_ERR_INVALID_ARG; }
  Wg9Args a;
  a.x = (const bf16_t*)x; a.x_bytes = (unsigned)((size_t)p->N * p->H * p->W * p->Cin * 2);
  a.dy = (const bf16_t*)dy; a.y_bytes = (unsigned)((size_t)p->N * p->H * p->W * p->Cout * 2);
  a.H = p->H; a.W = p->W; a.Hp = p->H + 1; a.Wp = p->W + 1; a.N = p->N; a.Cin = p->Cin; a.Cout = p->Cout;
  a.Q = p->N * a.Hp * a.Wp;
  a.D = (a.Wp + 1 + 7) / 8 * 8;
  a.R = 192 + 2 * a.D;
  a.units_ci = p->Cin / 64; a.units = a.units_ci * (p->Cout / 64);
  a.sps = sps; a.nstages = (a.Q + 63) / 64;
  const int hpwp = a.Hp * a.Wp;
  a.dn = 64 / hpwp; a.dh = (64 % hpwp) / a.Wp; a.dw = (64 % hpwp) % a.Wp;
  a.m64 = a.dn * p->H * p->W + a.dh * p->W + a.dw;
  a.rhwp = 1.0f / (float)hpwp; a.rwp = 1.0f / (float)a.Wp;
  a.slab = slab;
  const size_t lds_ring = (size_t)(a.R + 8) * 128 + 3 * 8192, lds_red = 4 * 9 * 4 * 64 * 16;
  const size_t lds = lds_ring > lds_red ? lds_ring : lds_red;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad9_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err != hipSuccess) { yolo_set_error("hipFuncSetAttribute failed: %s", hipGetErrorString(err)); return (int)err; }
    attr_set = true;
  }
  hipLaunchKernelGGL(wgrad9_kernel, dim3(a.units * nsplit), dim3(W9_NT), lds, stream, a, out);
  YOLO_LAUNCH_CHECK();
  return YOLO_OK;
}
