// 3x3 / stride-1 / SAME convolution of the 64-channel layers (forward and data gradient): the WEIGHTS stay in registers and the pixels
// stream past them, for gfx950 (MI355X).
//
// Replaces keras.layers.Conv2D (reference backbone/basic_backbone.py:20-43 via resnet18.py:29-32) and its TF autodiff data gradient on the
// 64 -> 64 channel layers of the 104 x 104 maps (416 x 416 input).  With C = 64 the whole K extent is 9 taps x 64 channels = 576: a
// tile kernel (conv3x3_strip_kernel) runs 9 K steps per tile and spends most of a workgroup's life in its prologue, its epilogue and the
// barrier of every K step (41 us alone for the benchmark layer).  Here:
//
// * ONE 512-thread workgroup per CU walks a CONTIGUOUS range of `span` pixels (M / 256 rounded up to 128) in steps of 128 pixels.
// * The 64 x 576 weight tile is read ONCE per workgroup (LDS-DMA into a swizzled, row-permuted image, then 36 ds_read_b128 per lane) and
//   lives in registers for the whole range: wave (nh, mi) holds the 32 output channels of half nh for all 36 k-substeps (144 VGPRs) and
//   computes the 32-pixel fragment mi of every step with 36 v_mfma_f32_32x32x16 against 36 fragment reads -- no weight traffic in the
//   loop.  (32x32x16 rather than 16x16x32: an MFMA holds the SIMD's vector issue for 8 cycles of its 32 instead of 8 of its 16, and one
//   lane address serves twice the flops; the first version of this kernel, on 16x16x32, was bound by vector issue at 2800 cycles per 64
//   pixels.)
// * The pixels live in a ring of 896 LDS rows (112 KiB) addressed relative to the range start: every step each wave appends TWO 8-row
//   pieces by LDS-DMA, three steps ahead of their use, waited on with a counted s_waitcnt vmcnt -- one barrier per step, and the first
//   fragments of a step are requested before the barrier that opens it.  SAME padding / row wrap / image boundaries: a per-lane 9-bit tap
//   mask; masked taps read a zero row.  Lane addresses advance by a constant per step (128 rows: the same swizzle phase).
// * The output fragment goes bf16 through a double-buffered LDS tile; during the NEXT step all 512 threads (128 pixels x 8 chunks, two per
//   thread) store it in whole 128-byte NHWC rows, accumulate the BatchNorm statistics (forward) or run the fused BatchNorm-backward
//   reduce (data gradient: ReLU mask, fan-in addend, sums of g and g xhat) -- the same arithmetic as conv_common.h tile_epilogue -- with
//   their global reads requested one step earlier, all of it spread under the MFMAs.  One statistics / partial row per workgroup.
// * The kernel owns every register and all LDS of its CU.  Alone that is its strength (31 us against 41); beside the weight-gradient
//   stream of the backward pass its workgroups wait for whole CUs to drain, so the automatic selection takes it for FORWARD launches only
//   (yolo_stream_plan; measured +1.1 % on the training step).  The data-gradient instantiations (EPI 1 / 2, ACC) are correct and tested
//   (tests/test_stream_gpu.py forces them) but not tuned: with the store side's BatchNorm operands on top of the 144 weight registers they
//   spill 30-120 VGPRs; a backward-pass form needs <= 128 VGPRs and two workgroups per CU (profiles/HISTORY.md, round 3).
#include "conv_common.h"

namespace {
// Diagnostic builds only (make EXTRA_conv_stream=-DST_STAMPS; tools/probes/stream_stamps.py): s_memtime stamps of wave 0 of every workgroup.
// In the product build no stamp executes and yolo_debug_st_stamps does not exist.
#ifdef ST_STAMPS
__device__ unsigned long long* g_st_stamps = nullptr;
#define ST_STAMP(i)                                                                                      \
  do {                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    unsigned long long t_;                                                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    if (g_st_stamps && threadIdx.x == 0) g_st_stamps[blockIdx.x * 64 + (i)] = t_;                        \
  } while (0)
// per-wave stamps (lane 0 of every wave), behind the per-workgroup block
#define ST_WSTAMP(i)                                                                                     \
  do {                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    unsigned long long t_;                                                                               \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    if (g_st_stamps && (threadIdx.x & 63) == 0) g_st_stamps[gridDim.x * 64 + (blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + (i)] = t_; \
  } while (0)
#else
#define ST_STAMP(i) do {} while (0)
#define ST_WSTAMP(i) do {} while (0)
#endif

struct StreamArgs {
  const bf16_t* src; unsigned src_bytes;   // NHWC activations (or dY for the data gradient), C = 64
  const bf16_t* wt;  unsigned wt_bytes;    // [Kout][9][64]
  int H, W, M;                             // M = N*H*W
  int span, ny;                            // pixels per workgroup (multiple of 128), channel tiles (Kout / 64)
  int e0;                                  // 256 + roundup8(W + 1): the ring is filled up to pixel P0 + e0 before the first step
  float rhw, rw;
};

constexpr int ST_ROWS = 896;                         // ring rows (a multiple of the 16-row swizzle period)
constexpr int ST_RING_BYTES = ST_ROWS * 128;         // 112 KiB
constexpr int ST_STEP = 128;                         // pixels per step
constexpr int ST_ZERO = ST_RING_BYTES;               // 128 zero bytes
constexpr int ST_OLD = 144;                          // staged output row stride (128 + 16)
constexpr int ST_STAGE = ST_ZERO + 128;              // 2 staged tiles of 128 x 144 bytes
constexpr int ST_LDS = 160 * 1024;
constexpr int ST_WT = ST_LDS - 9 * 8192;             // weight image (prologue only): over the ring's last rows, the zero row and the staged tiles
static_assert(ST_STAGE + 2 * ST_STEP * ST_OLD <= ST_LDS, "LDS budget");

typedef unsigned st_v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x8_t st_lds_frag_t;

// row swizzle of the [rows][64 bf16] images read by the 32x32x16 operand loads (32 consecutive rows, one 16-byte chunk each): the chunk index
// is XOR-ed with bits 1-3 of the row -- conflict-free for ds_read_b128's lane groups at every row alignment
__device__ __forceinline__ int st_sw(int row) { return (row >> 1) & 7; }

// the bitwise select (m & a) | (~m & b) with m = bit `bit` of v as a mask (0 / -1): kept opaque so that the compiler does not turn the pair into
// v_cmp + v_cndmask through VCC (which costs s_nop wait states beside the MFMAs)
__device__ __forceinline__ int st_select_bit(unsigned v, int bit, int a, int b) {      // bit `bit` of v ? a : b
  int d;
  asm("v_bfe_i32 %0, %1, %2, 1\n\tv_bfi_b32 %0, %0, %3, %4" : "=&v"(d) : "v"(v), "s"(bit), "v"(a), "v"(b));
  return d;
}
template <int N>
__device__ __forceinline__ void st_wait_vmcnt_upto(int n) {      // s_waitcnt vmcnt(n) for a wave-uniform n <= N (the immediate must be a constant)
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else {
    if (n >= N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    else st_wait_vmcnt_upto<N - 1>(n);
  }
}

// EPI: 0 = plain (forward: statistics of the stored values; plain data gradient), 1 = fused BatchNorm-backward reduce, 2 = the same with a
// second (shortcut) BatchNorm.  ACC: the output is added to `addend` (or to the output buffer itself).  The hot loop is ONE basic block:
// every out-of-range access of the store side goes through buffer descriptors (dropped stores, zero loads) instead of branches.
template <int EPI, bool ACC>
__global__ __launch_bounds__(512) void conv3x3_stream_kernel(StreamArgs a, void* __restrict__ Yv, int ldy, float* __restrict__ stat_sum,
                                                             float* __restrict__ stat_sq, int Kout, BnEpi bn) {
  constexpr bool BNEPI = EPI != 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (wave-uniform: scalar registers)
  const int nh = wave >> 2, mi = wave & 3;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = tile % a.ny, wg_m = tile / a.ny;
  const int n0 = tile_n * 64;
  const int P0 = wg_m * a.span;
  const int P1 = min(P0 + a.span, a.M);
  const int niter = (P1 - P0 + ST_STEP - 1) / ST_STEP;
  const int Pbase = ((P0 - a.W - 1 + 1024) / 8 - 128) * 8;       // ring row 0 = pixel 8 * floor((P0 - W - 1) / 8): every pixel the range needs is >= it
  ST_STAMP(0);

  // ---- prologue: ring pieces [Pbase, P0 + e0) and the weight image ---------------------------------------------------------------
  // LDS-DMA lane geometry: a piece is 8 rows x 128 bytes; lane -> row lane >> 3, 16-byte slot lane & 7, which holds chunk slot ^ st_sw(row);
  // for row 8k + lrow that is 4 (k & 1) + (lrow >> 1)
  const int lrow = lane >> 3;
  auto lane_src = [&](int parity) { return lrow * 128 + (((lane & 7) ^ (4 * parity + (lrow >> 1))) << 4); };
  const int K0 = (P0 + a.e0 - Pbase) >> 3;                        // pieces before the first step (< 96: they do not wrap)
  {
    // weight image first: row R = 32 h + r of tap t holds channel n0 + 32 h + perm(r) -- MFMA row r of the half-h waves (perm: see the accumulators)
    const int R = wave * 8 + lrow, r = R & 31;
    const int ch = n0 + (R & 32) + 16 * ((r >> 2) & 1) + 4 * (r >> 3) + (r & 3);
    const unsigned wrow = (unsigned)((ch * 576 + ((lane & 7) ^ st_sw(R)) * 8) * 2);
#pragma unroll
    for (int t = 0; t < 9; ++t) buffer_load_lds16(a.wt, a.wt_bytes, smem + ST_WT + t * 8192 + wave * 1024, wrow + t * 128);
    const int ls = lane_src(wave & 1);
    for (int k = wave; k < K0; k += 8)
      buffer_load_lds16(a.src, a.src_bytes, smem + k * 1024, (unsigned)((Pbase + 8 * k) * 128 + ls));
  }
  st_wait_vmcnt_upto<12>((K0 - wave + 7) >> 3);                    // the weight pieces have landed (the ring pieces, issued after them, may still fly)
  __syncthreads();
  ST_STAMP(1);

  const int kh = lane >> 5;                                       // k half of the 32x32x16 operand layout: 8 channels 8 kh .. +7 of a 16-channel substep
  bf16x8_t wreg[36];
  {
    const int R = nh * 32 + (lane & 31);
#pragma unroll
    for (int s_ = 0; s_ < 36; ++s_)
      wreg[s_] = *reinterpret_cast<const bf16x8_t*>(smem + ST_WT + (s_ >> 2) * 8192 + R * 128 + (((2 * (s_ & 3) + kh) ^ st_sw(R)) << 4));
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();                                                // the weight image is dead; the prologue's ring pieces have landed
  // the zero row; and zeros in the staged tiles, so that step 0's store side (nothing staged yet, offsets out of range) adds zeros to the sums
  for (int o = tid * 16; o < 128 + 2 * ST_STEP * ST_OLD; o += 512 * 16) *reinterpret_cast<uint4*>(smem + ST_ZERO + o) = make_uint4(0u, 0u, 0u, 0u);
  ST_STAMP(2);

  // ---- per-lane state of the compute side: pixel P0 + 128 it + 32 mi + (lane & 31) -----------------------------------------------
  int px = P0 + 32 * mi + (lane & 31);
  int py_, px_;                                                   // (row, column) of that pixel inside its image
  {
    int n_, rem;
    fast_divmod(px, a.H * a.W, a.rhw, n_, rem);
    fast_divmod(rem, a.W, a.rw, py_, px_);
  }
  const int zero_addr = ST_ZERO + kh * 16;
  const int stage_wr = ST_STAGE + (32 * mi + (lane & 31)) * ST_OLD + nh * 64 + kh * 32;
  unsigned tad[9];                                                // unmasked LDS address of this lane's row of tap t, chunk kh (substep s: ^ (s << 5))
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int rr = px + (t / 3) * a.W + (t % 3) - a.W - 1 - Pbase;     // 0 <= rr < 768 in the first step
    tad[t] = (unsigned)(rr * 128 + ((kh ^ st_sw(rr)) << 4));
  }

  // ---- per-thread state of the store side: pixel slots tid >> 3 and 64 + (tid >> 3), channel chunk tid & 7 -----------------------
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
  const st_v4u zero4 = {0u, 0u, 0u, 0u};
  st_v4u e_yv[2] = {zero4, zero4}, e_ev[2] = {zero4, zero4}, e_y2[2] = {zero4, zero4};
  unsigned e_mk[2] = {0xffu, 0xffu}, e_off[2] = {0xfffffff0u, 0xfffffff0u};     // byte offsets of this thread's 2 x 16 bytes of the block (out of range: none)
  // global reads of the store side for block `it` (requested one step before they are used); out-of-range offsets read zeros
  auto epi_load = [&](int it) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int m = P0 + ST_STEP * it + 64 * b + spx;
      e_off[b] = m < P1 ? ((unsigned)m * (unsigned)ldy + (unsigned)sc) * 2u : 0xfffffff0u;
      if constexpr (BNEPI) {
        e_yv[b] = __builtin_amdgcn_raw_buffer_load_b128(rBy, (int)e_off[b], 0, 0);
        e_mk[b] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rMk, (int)(e_off[b] >> 4), 0, 0) | nomask;
      }
      if constexpr (EPI == 2) e_y2[b] = __builtin_amdgcn_raw_buffer_load_b128(rBy2, (int)e_off[b], 0, 0);
      if constexpr (ACC) e_ev[b] = __builtin_amdgcn_raw_buffer_load_b128(rAdd, (int)e_off[b], 0, 0);
    }
  };
  // store side of one half block (staged by the previous step): the arithmetic of conv_common.h tile_epilogue, branch-free
  auto store_block = [&](int b, st_v4u sv) {
    uint4 v = make_uint4(sv.x, sv.y, sv.z, sv.w);
    float g8[8], y8[8];
    if constexpr (ACC) {                              // gradient fan-in: float32 add, one rounding
      unpack_bf8(v, g8);
      unpack_bf8(make_uint4(e_ev[b].x, e_ev[b].y, e_ev[b].z, e_ev[b].w), y8);
#pragma unroll
      for (int j = 0; j < 8; ++j) g8[j] += y8[j];
      v = pack_bf8(g8);
    }
    if constexpr (BNEPI) {
      unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q = 0; q < 4; ++q)
        w4[q] = (((e_mk[b] >> (2 * q)) & 1u) ? (w4[q] & 0xffffu) : 0u) | (((e_mk[b] >> (2 * q + 1)) & 1u) ? (w4[q] & 0xffff0000u) : 0u);
      v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
    }
    const st_v4u ov = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(ov, rY, (int)e_off[b], 0, 0);      // (out-of-range offsets are dropped)
    unpack_bf8(v, g8);                                // (pixels beyond the range: every tap was masked, the staged value is 0)
    if constexpr (BNEPI) {
      unpack_bf8(make_uint4(e_yv[b].x, e_yv[b].y, e_yv[b].z, e_yv[b].w), y8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s0[j] += g8[j]; s1[j] += g8[j] * ((y8[j] - mu[j]) * rs[j]); }
      if constexpr (EPI == 2) {
        unpack_bf8(make_uint4(e_y2[b].x, e_y2[b].y, e_y2[b].z, e_y2[b].w), y8);
#pragma unroll
        for (int j = 0; j < 8; ++j) s2[j] += g8[j] * ((y8[j] - mu2[j]) * rs2[j]);
      }
    } else if constexpr (!ACC) {                      // statistics of the values as stored (bf16-rounded)
#pragma unroll
      for (int j = 0; j < 8; ++j) { s0[j] += g8[j]; s1[j] += g8[j] * g8[j]; }
    }
  };

  // this wave's two ring pieces per step: pieces K0 + 16 (it + 1) + 2 wave + {0, 1} -- the pixels of step it + 3
  int pdst = ((K0 + 2 * wave) * 1024) % ST_RING_BYTES;
  int psrc = (Pbase + 8 * (K0 + 2 * wave)) * 128;
  const int ls0 = lane_src(K0 & 1), ls1 = lane_src((K0 + 1) & 1);
  const int src_end = (P1 + a.W + 1) * 128;                       // first byte no step of this range reads: pieces beyond it are not fetched
  auto issue_piece = [&](int j) {                                 // (they are still issued, out of range: zeros, the same vmcnt arithmetic)
    if (j == 0) { buffer_load_lds16(a.src, a.src_bytes, smem + pdst, psrc < src_end ? (unsigned)(psrc + ls0) : 0x80000000u); return; }
    const int d1 = pdst + 1024 >= ST_RING_BYTES ? pdst + 1024 - ST_RING_BYTES : pdst + 1024;
    buffer_load_lds16(a.src, a.src_bytes, smem + d1, psrc + 1024 < src_end ? (unsigned)(psrc + 1024 + ls1) : 0x80000000u);
    pdst = pdst + 16 * 1024 >= ST_RING_BYTES ? pdst + 16 * 1024 - ST_RING_BYTES : pdst + 16 * 1024;
    psrc += 16 * 1024;
  };
  issue_piece(0);                                                  // the pixels of step 2 (the prologue brought those of steps 0 and 1)
  issue_piece(1);
  // tap masks: SAME padding, row wrap, image boundary, pixels beyond M read the zero row
  auto tap_bits = [&]() {                                          // (sign-bit arithmetic: no VCC round trips)
    const int c0 = (int)((unsigned)(-px_) >> 31);                  // column > 0
    const int c2 = (int)((unsigned)(px_ - (a.W - 1)) >> 31);       // column < W - 1
    const int cb = c0 | 2 | (c2 << 2);
    const int r0 = (-py_) >> 31, r2 = (py_ - (a.H - 1)) >> 31;     // row > 0, row < H - 1 (as masks)
    const int ok_ = (cb & r0) | (cb << 3) | ((cb << 6) & r2);
    return (unsigned)(ok_ & ((px - a.M) >> 31));
  };
  unsigned ok = tap_bits();
  // the masked address of tap t for the coming step; the unmasked one moves 128 ring rows per step (128 % 16 == 0: same swizzle)
  auto tap_addr = [&](int t) {
    const int ad = st_select_bit(ok, t, (int)tad[t], zero_addr);
    const unsigned x = tad[t] + ST_STEP * 128, y = tad[t] + ST_STEP * 128 - ST_RING_BYTES;
    tad[t] = x < y ? x : y;
    return ad;
  };
  __syncthreads();                                                 // the zero row is written
  // the first three fragments of a step are requested at the END of the step before it (before the barrier between them: the pieces a
  // step reads have landed one step earlier), so that the MFMAs resume right behind the barrier
  bf16x8_t fr[4];
  int cur = tap_addr(0), nxt = 0;
  fr[0] = *reinterpret_cast<const st_lds_frag_t*>(cur);            // (dynamic LDS starts at address 0: no base add)
  fr[1] = *reinterpret_cast<const st_lds_frag_t*>(cur ^ 32);
  fr[2] = *reinterpret_cast<const st_lds_frag_t*>(cur ^ 64);
  // one step: the store side of the previous block, the ring pieces of step it + 3, 36 fragment reads and 36 MFMAs (32x32x16)
  auto step = [&](int it, bool compute, auto phase) {
    constexpr int PH = decltype(phase)::value;        // the two waves of a SIMD (halves nh = 0 / 1) place the store side under different MFMAs
    // the pieces of steps <= it + 1 and the store side's reads of block it - 1 have landed (outstanding: the two pieces issued in step
    // it - 1); every wave has finished step it - 1 (its ring reads, its staged fragment: the 3 youngest LDS operations are the fragment reads)
    if (it < 32) ST_STAMP(4 + it);
    if (compute) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(3)" ::: "memory");
    else         asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    if (it == 5) ST_WSTAMP(0);
    if (it == 6) ST_WSTAMP(2);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (it == 5) ST_WSTAMP(1);
    if (it == 6) ST_WSTAMP(3);
    const int sbuf = ((it + 1) & 1) * (ST_STEP * ST_OLD);                                              // block it - 1 (it = 0: zeros)
    st_v4u stagedA, stagedB;
    if (!compute) {
      stagedA = *reinterpret_cast<const st_v4u*>(smem + stage_rd + sbuf);
      stagedB = *reinterpret_cast<const st_v4u*>(smem + stage_rd + sbuf + 64 * ST_OLD);
      store_block(0, stagedA);
      store_block(1, stagedB);
      return;
    }
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // software pipeline, fenced so that the compiler keeps it: the fragment of substep s + 3 is requested BEFORE the MFMA of substep s, and the
    // step's other work (next tap address, ring pieces, store side, its global reads) is spread under the MFMAs
#pragma unroll
    for (int s_ = 0; s_ < 36; ++s_) {
      if (s_ + 3 < 36) {
        const int base = ((s_ + 3) >> 2) == (s_ >> 2) ? cur : nxt;
        fr[(s_ + 3) & 3] = *reinterpret_cast<const st_lds_frag_t*>(base ^ (((s_ + 3) & 3) << 5));
      }
      __builtin_amdgcn_sched_barrier(0);
      if ((s_ & 3) == 0 && (s_ >> 2) + 1 < 9) nxt = tap_addr((s_ >> 2) + 1);
      acc = YOLO_MFMA_32x32x16(wreg[s_], fr[s_ & 3], acc);
      if (s_ == 1 + 8 * PH) issue_piece(0);
      if (s_ == 2 + 8 * PH) issue_piece(1);
      // (the staged tile is read INSIDE the fenced stages, a few MFMAs before its use: hoisted to the barrier, its unpacking sat before the first MFMA)
      if (s_ == 2 + 14 * PH) stagedA = *reinterpret_cast<const st_v4u*>(smem + stage_rd + sbuf);
      if (s_ == 5 + 14 * PH) store_block(0, stagedA);
      if (s_ == 10 + 14 * PH) stagedB = *reinterpret_cast<const st_v4u*>(smem + stage_rd + sbuf + 64 * ST_OLD);
      if (s_ == 13 + 14 * PH) store_block(1, stagedB);
      if (s_ == 21 + 12 * PH) epi_load(it);
      if (s_ == 33) {                                               // (after the step's last tap_addr: the masks of the NEXT step)
        px += ST_STEP;
        px_ += ST_STEP;
#pragma unroll
        for (int k = 0; k < 2; ++k) {                               // (W >= 64: two column wraps at most)
          const int q = px_ - a.W, w1 = q >> 31;
          px_ = q + (a.W & w1);
          py_ += 1 + w1;
        }
        { const int q = py_ - a.H, w1 = q >> 31; py_ = q + (a.H & w1); }       // (H >= 2, or H == 1 handled by the second line)
        { const int q = py_ - a.H, w1 = q >> 31; py_ = q + (a.H & w1); }
        ok = tap_bits();
      }
      if ((s_ & 3) == 3) cur = nxt;
      if (s_ == 35) cur = tap_addr(0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // accumulator i of lane (pixel, h) is MFMA row (i & 3) + 8 (i >> 2) + 4 h = channel 16 h + i of this wave's 32 (the row permutation of the
    // weight image): 16 consecutive channels, two 16-byte writes into the staged tile
    st_v4u o0, o1;
    o0.x = pack_bf2(acc[0], acc[1]);   o0.y = pack_bf2(acc[2], acc[3]);   o0.z = pack_bf2(acc[4], acc[5]);   o0.w = pack_bf2(acc[6], acc[7]);
    o1.x = pack_bf2(acc[8], acc[9]);   o1.y = pack_bf2(acc[10], acc[11]); o1.z = pack_bf2(acc[12], acc[13]); o1.w = pack_bf2(acc[14], acc[15]);
    char* const sw_ = smem + stage_wr + (it & 1) * (ST_STEP * ST_OLD);
    *reinterpret_cast<st_v4u*>(sw_) = o0;
    *reinterpret_cast<st_v4u*>(sw_ + 16) = o1;
    __builtin_amdgcn_sched_barrier(0);
    fr[0] = *reinterpret_cast<const st_lds_frag_t*>(cur);            // the next step's first fragments (the last step's are never used)
    fr[1] = *reinterpret_cast<const st_lds_frag_t*>(cur ^ 32);
    fr[2] = *reinterpret_cast<const st_lds_frag_t*>(cur ^ 64);
  };
  ST_STAMP(3);
  if (nh == 0) {                                                    // (step 0's store side sees e_off out of range and a zero tile)
    for (int it = 0; it < niter; ++it) step(it, true, std::integral_constant<int, 0>());
  } else {
    for (int it = 0; it < niter; ++it) step(it, true, std::integral_constant<int, 1>());
  }
  step(niter, false, std::integral_constant<int, 0>());
  ST_STAMP(40);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the last pieces are never read; nothing may be in flight into LDS at the end)

  // ---- one statistics / partial row per workgroup: [64 pixel slots][64 channels] through the (dead) ring, 8 slots per thread, then 8 groups
  const int nq = BNEPI ? (EPI == 2 ? 3 : 2) : ((!ACC && (stat_sum || fwd_acc)) ? 2 : 0);
  if (nq == 0) return;
  __syncthreads();                                                 // every wave is out of the loop: the ring is dead
  ST_STAMP(41);
  float* const red = reinterpret_cast<float*>(smem);               // [3][64][64], then [3][8][64] behind it
  float* const red2 = red + 3 * 64 * 64;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[(0 * 64 + spx) * 64 + sch * 8 + j] = s0[j];
    red[(1 * 64 + spx) * 64 + sch * 8 + j] = s1[j];
    if constexpr (EPI == 2) red[(2 * 64 + spx) * 64 + sch * 8 + j] = s2[j];
  }
  __syncthreads();
  {
    const int cl = tid & 63, part = tid >> 6;
    for (int q = 0; q < nq; ++q) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += red[(q * 64 + part * 8 + k) * 64 + cl];
      red2[(q * 8 + part) * 64 + cl] = t;
    }
  }
  __syncthreads();
  const bool grouped = bn.group > 0 && (BNEPI ? bn.partial != nullptr : stat_sum != nullptr);       // two-level partial rows (conv_common.h rows_fold)
  const size_t rrow = grouped ? (size_t)yolo_row_groups(bn.rows, bn.group) + wg_m : (size_t)wg_m;
  if (tid < nq * 64) {
    const int q = tid >> 6, cl = tid & 63;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) t += red2[(q * 8 + w) * 64 + cl];
    if constexpr (BNEPI) {
      if (grouped) row_store(bn.partial + (rrow * 3 + q) * ldy + n0 + cl, t);
      else if (bn.partial) bn.partial[((size_t)wg_m * 3 + q) * ldy + n0 + cl] = t;
      else yolo_acc_add(bn.acc, 3, ldy, wg_m % YOLO_ACC_NB, q, n0 + cl, t);
    } else {
      if (grouped) row_store((q == 0 ? stat_sum : stat_sq) + rrow * Kout + n0 + cl, t);
      else if (stat_sum) (q == 0 ? stat_sum : stat_sq)[(size_t)wg_m * Kout + n0 + cl] = t;
      else yolo_acc_add(bn.acc, 2, Kout, wg_m % YOLO_ACC_NB, q, n0 + cl, t);
    }
  }
  if (grouped) {
    __syncthreads();
    if constexpr (BNEPI) rows_fold<512>(bn.partial, bn.partial + ldy, bn.partial + 2 * (size_t)ldy, nq, (size_t)3 * ldy, bn.rows, bn.group, wg_m,
                                        tile_n, a.ny, n0, 64, tid, reinterpret_cast<int*>(smem));
    else rows_fold<512>(stat_sum, stat_sq, nullptr, 2, (size_t)Kout, bn.rows, bn.group, wg_m, tile_n, a.ny, n0, 64, tid, reinterpret_cast<int*>(smem));
  }
  ST_STAMP(42);
}

bool st_eligible(const yoloconv::Gather& g, int Kout, bool f32) {
  if (f32 || g.den != 1 || g.C0 != 0 || g.S != 3 || g.RS != 9 || g.smul != 1 || g.pad_h != 1 || g.pad_w != 1 || g.s2) return false;
  if (g.Hs != g.Ho || g.Ws != g.Wo || g.C1 != 64 || Kout % 64 != 0) return false;
  const int r8 = (g.Wo + 1 + 7) / 8 * 8;
  if (g.Wo < 64 || 512 + r8 + g.Wo + 1 > ST_ROWS) return false;                          // two column wraps per step at most; the ring's reach
  if ((g.Wo + 9 + 256 + r8) * 128 > ST_WT) return false;                                  // the prologue's pieces stay below the weight image
  if ((size_t)g.M * 128 >= (1ull << 31) || (size_t)Kout * 576 * 2 >= (1ull << 31) || (size_t)g.M * Kout * 2 >= (1ull << 31)) return false;
  return true;
}

}  // namespace

// "stream" tuning: -1 auto = forward launches with >= 512 pixels per workgroup (the kernel owns every register and all LDS of its CU: beside
// the weight-gradient stream of the backward pass its workgroups wait for whole CUs to drain -- measured 78-156 us per launch in the step
// against 33 alone), 0 never, 1 wherever it fits (tests), 2 = auto for the data gradient as well
int g_stream = -1;

// 0 = the streaming kernel does not take this problem, else the pixels per workgroup (statistics / partial rows = ceil(M / that))
int yolo_stream_plan(const yoloconv::Gather& g, int Kout, bool f32, StreamPlanOut* out) {
  if (g_stream == 0 || !st_eligible(g, Kout, f32)) return 0;
  const int ny = Kout / 64;
  const int nxt = ny >= 256 ? 1 : 256 / ny;
  int span = ((g.M + nxt - 1) / nxt + ST_STEP - 1) / ST_STEP * ST_STEP;
  if (g_stream != 1 && span < 512) return 0;           // auto: at least 4 steps behind one weight load
  if (g_stream < 0 && g.role != 0) return 0;
  if (span < ST_STEP) span = ST_STEP;
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
  a.e0 = 256 + (g.Wo + 1 + 7) / 8 * 8;
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

#ifdef ST_STAMPS
extern "C" int yolo_debug_st_stamps(void* buf) {
  unsigned long long* p = (unsigned long long*)buf;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_st_stamps), &p, sizeof(p));
}
#endif
