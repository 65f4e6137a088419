/* yolov3_amd.h -- C-ABI of the MI355X-native (gfx950) kernels behind the YOLOv3 training hot path.
 *
 * The reference (zheng-yuwei/YOLOv3-tensorflow) has no FFI layer of its own: its hot path is reached through Python
 * module surfaces and executed by TensorFlow built-in ops.  Each entry point below replaces the
 * TensorFlow op(s) instantiated at the cited reference call site; the Python facades in yolov3_tensorflow_amd/ bind
 * them with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; nothing is allocated, freed or synchronised here;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued on it, so every call is
 *     capturable into a hipGraph;
 *   - activations are NHWC bf16 (uint16 storage); logits, statistics, master weights and gradients are float32;
 *   - conv weights (bf16 compute copies): forward/wgrad layout [Cout][R][S][Cin], dgrad layout [Cin][R][S][Cout] with the
 *     taps flipped (made by yolo_repack_dgrad_weights);
 *   - return value: 0 (YOLO_OK) on success, YOLO_ERR_INVALID_ARG (<0) for a rejected argument, or a positive
 *     hipError_t; yolo_last_error() returns a thread-local message for the last non-zero status.
 */
#ifndef YOLOV3_AMD_H_
#define YOLOV3_AMD_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YOLO_OK 0
#define YOLO_ERR_INVALID_ARG (-1)
#define YOLO_ABI_VERSION 1
#define YOLO_DTYPE_BF16 0
#define YOLO_DTYPE_FP16 1

int yolo_abi_version(void);
/* 16-bit element type of this build: YOLO_DTYPE_BF16 (libyolov3_amd.so) or YOLO_DTYPE_FP16 (libyolov3_amd_fp16.so, the same sources
 * compiled with -DYOLO_FP16: wherever this header says "bf16" that library stores IEEE half and multiplies with the f16 MFMA). */
int yolo_abi_dtype(void);
const char* yolo_last_error(void);
/* ------------------------------------------------------------------------------------------------------------------
 * Launch sequencer.  The reference runs a training step as ONE session.run of a graph TensorFlow built once
 * (/root/reference/yolov3/trainer.py:84,113: compile + fit); here the step is a fixed list of ~270 kernel launches over static buffers on
 * two or three streams.  Between yolo_seq_begin() and yolo_seq_end() every launch made through this library (and every yolo_seq_fork) is
 * executed AND recorded -- kernel, grid, block, LDS bytes, stream, a copy of the argument values; yolo_seq_run(id, begin, end) re-issues
 * items [begin, end) with one call (the host cost of a step drops from ~2.8 ms of Python + ctypes to the bare HIP launches).
 * yolo_seq_mark() = number of items recorded so far (segment boundary for host work that must happen between launches, e.g. a collective).
 * yolo_seq_fork(a, b): stream b waits for everything queued on stream a at this point (event record + stream wait; usable outside a
 * recording too).  yolo_seq_fork_local(a, b): the same edge for the case where ONLY kernels of this device wait behind it (main <->
 * weight-gradient stream): its event is created with hipEventDisableSystemFence -- the kernels' own agent-scope release / acquire make
 * their stores visible to each other, the system-scope writeback of a default event record is for copy engines, peers and the host
 * (+1.1 % on the step, profiles/r04_fork_fence_ab.txt; env YOLO_FORK_FENCE=system | device selects the default / device-scope event).
 * Use yolo_seq_fork in front of anything that is not a kernel of this device (a collective, a copy).
 * One recording at a time, single-threaded; the caller re-records when any buffer address or launch decision changes.
 * ------------------------------------------------------------------------------------------------------------------ */
int yolo_seq_begin(void);                 /* -> sequence id */
int yolo_seq_mark(void);
int yolo_seq_end(void);                   /* -> number of items */
int yolo_seq_fork(void* from_stream, void* to_stream);
int yolo_seq_fork_local(void* from_stream, void* to_stream);
int yolo_seq_run(int seq, int begin, int end);
int yolo_seq_free(int seq);
/* CRC-32C (Castagnoli) of a host buffer, continuing from `seed` (0 to start): the checksum of the TensorFlow checkpoint files the
 * reference reads / writes through TensorFlow (/root/reference/yolov3/trainer.py:47-67,90-91; utils/tf_checkpoint.py here). */
uint32_t yolo_crc32c(const void* data, size_t n, uint32_t seed);

/* ------------------------------------------------------------------------------------------------------------------
 * Convolution as implicit GEMM on MFMA (v_mfma_f32_16x16x32_bf16), NHWC.
 * Replaces keras.layers.Conv2D at /root/reference/backbone/basic_backbone.py:42 (all backbone/neck convs) and
 * /root/reference/yolov3/yolov3_detector.py:98-100,123-125,148-150 (detection convs, bias, float32 logits), plus the
 * TF autodiff gradients of those ops.  UpSampling2D + concatenate (yolov3_detector.py:115-116,140-141) is folded into
 * the operand gather of the following 1x1 convolution (C0 > 0).
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t N, H, W;     /* input batch and spatial size */
  int32_t Cin;         /* input channels, multiple of 8 and (Cin/8) a power of two (the RGB stem is padded 3 -> 8) */
  int32_t C0;          /* > 0: the input is concat(upsample2x(src0[N,H/2,W/2,C0]), src1[N,H,W,Cin-C0]); 0: src1 only */
  int32_t Cout;        /* output channels, padded to 64 * 2^k (255 -> 256, 170 -> 256) */
  int32_t R, S;        /* kernel height / width */
  int32_t stride;      /* 1 or 2 (same in both directions) */
  int32_t pad_t, pad_l;/* top/left zero padding (TF 'same': (0,0) for k=3 s=2 on even sizes, (1,1) for k=3 s=1) */
  int32_t Ho, Wo;      /* output spatial size */
} yolo_conv_problem;

/* y[N,Ho,Wo,Cout] = conv(x, w) (+ bias).  y is bf16 unless y_is_f32.  If stat_sum/stat_sq are non-NULL the kernel also
 * writes per-channel partial sums / sums of squares of the bf16-rounded outputs: arrays [stat_rows][Cout] where
 * stat_rows = yolo_conv2d_stat_rows(p) (reduced later by yolo_bn_finalize). */
int yolo_conv2d_stat_rows(const yolo_conv_problem* p);
/* Which kernel yolo_conv2d_fwd would launch for p under the current tuning, without launching (tests, tools): info[0] = family (0 implicit
 * GEMM, 1 LDS-resident strip, 2 retired (the big-tile kernel of round 3), 3 RGB stem, 4 weights-in-registers streaming = conv_stream.hip, 5 = 32x32x16 / 64 x 64 wave-tile strip kernel = conv_s32.hip), info[1] / info[2] = pixel /
 * channel tile, info[3] = pixels a tile owns (streaming: pixels per workgroup), info[4] = workgroups, info[5] = dynamic LDS bytes
 * (streaming, family 5), info[6] = configuration id (family 5),
 * info[7] = reserved.  The data gradient of p is planned like the forward pass of the problem with Cin and Cout swapped. */
int yolo_conv2d_fwd_plan(const yolo_conv_problem* p, int32_t* info8);
/* Test / benchmark hook: override a kernel-selection heuristic.  "strip_bm": -1 auto (default), 0 never use the LDS-resident strip
 * kernel for 3x3 stride-1 convolutions, 64 / 128 / 256 force its pixel tile; "strip_bn": 0 auto, 64 / 128; "wgrad_strip": 0 / 1;
 * "stem_direct": 1 (default) / 0 the RGB stem (Cin 8, Cout 64, 3x3 stride 2) on its row-walking kernel or on the implicit GEMM (changes yolo_conv2d_stat_rows);
 * "dw_tiled": 1 (default) / 0 the mixed depthwise forward / data gradient on its tiled kernel or on the row-tile kernel, > 1 = workgroups per
 * 64-channel slab of the tiled kernel's persistent grid (default 512);
 * "stream": -1 (default) the weights-in-registers streaming kernel (conv_stream.hip) for FORWARD 3x3 stride-1 launches with 64 input channels and
 * >= 512 pixels per workgroup / 0 never / 1 wherever it fits, data gradients included / 2 = the automatic rule for data gradients too (changes
 * yolo_conv2d_stat_rows and yolo_conv2d_dgrad_bn_rows); "bwd_fin_small": 0 (default) / 1 yolo_bn_bwd_finalize on 1024- or 256-thread workgroups (bit-identical results);
 * "s32": -1 (default) the automatic rule of the 32x32x16 strip kernel (conv_s32.hip: 3x3 stride-1 layers below 80 columns; it yields while "strip_bm" is not -1) / 0 never /
 * 1 + id force tile configuration id (0: 128 x 128, 1: 256 x 64, 2: 128 x 64 K split 2, 3: 64 x 128 K split 2, 4: 64 x 64 K split 4, 5: 256 x 128 on 8 waves, 6: 128 x 128 on 8 waves
 * K split 2, 7: 256 x 64 on 8 waves K split 2, 8: 384 x 128 on 8 waves with 96 x 64 wave tiles) where it fits (changes yolo_conv2d_stat_rows and yolo_conv2d_dgrad_bn_rows);
 * "strip_xsplit": -1 (default: 2 where the input is smaller than 4x the weights -- the 13 x 13 layers --, else 0) / 2 / 4 / 0: the strip kernel deals its tiles to the XCDs as rectangles -- XCD x owns channel-tile group x % G of pixel part x / G, so an
 * XCD's L2 sees 1 / G of the weights and G / 8 of the pixels -- or (0) as contiguous runs (all weights, 1 / 8 of the pixels per L2); same tiles, bit-identical results;
 * "s32_s2": 1 (default) the four parity classes of a 3x3 / stride-2 data gradient on an even map run on the 32x32x16 strip kernel (conv_s32.hip MODE 1:
 * four K-step slots per slice, a class skips the slots it does not have) / 0 on the implicit GEMM (changes yolo_conv2d_dgrad_bn_rows);
 * "wgrad9": -1 (default) the stationary-output weight gradient (conv_wgrad9.hip) for 3x3 stride-1 layers of 20 x 20 pixels and more / 0 never / 1 wherever it
 * fits (changes yolo_conv2d_wgrad_splits and the slab sizes: re-plan);
 * "wgrad9_wgs": workgroups of that kernel's grid (16..1024, default 128: one workgroup per CU on half the CUs, the other half stays free for the main stream);
 * "reduce_wgs": 0 (default: one workgroup per table block) or the most workgroups of yolo_wgrad_reduce_batched, "opt_wgs": the most workgroups (16..2048,
 * default 2048) of the optimizer / cast launches -- both measured as CU partitions for the side stream (profiles/r04_side_grid_caps_ab.txt: no gain);
 * "ew_nt": bit mask, default 3: non-temporal loads of the streamed-once operands of the BatchNorm backward (1) / forward (2) apply kernels;
 * "acc_stream_kelems": tensors from this many thousand elements take the streaming form of the accumulator-fed BatchNorm launches;
 * "strip_ws": 0 auto / 2 / 3 weight-ring stages; "s2_classes": 0 / 1 stride-2 data gradient as four dense parity classes;
 * "wgrad_target": workgroups the split-K plan aims at (64..4096, default 384; changes yolo_conv2d_wgrad_workspace_bytes);
 * "wgrad_xcd": 0 / 1 XCD-chunked 1-D weight-gradient grids; "wgrad_ring": 2 / 3 operand stages and "wgrad_pipe": 0 / 1 software-pipelined
 * stage body of the strip weight gradient; "bn_fused_min_chunks": smallest per-thread chunk count (1..12, default 3) served by the
 * full grid of yolo_bn_act_bwd_fused; "bn_fused_small_grid": 0 (default: smaller tensors use the three-kernel path) or the fixed
 * workgroup count (16..255) of a second, smaller cooperative grid with its own barrier counters -- set it once, before the first
 * fused launch on a given sync_words buffer.  Tile choices change yolo_conv2d_stat_rows(): set them before sizing statistics buffers.
 * Process-wide, not thread-safe: meant for A/B runs and tests. */
int yolo_set_tuning(const char* name, int value);
int yolo_conv2d_fwd(const yolo_conv_problem* p, const void* src0, const void* src1, const void* w_fwd, const float* bias,
                    void* y, int y_is_f32, float* stat_sum, float* stat_sq, void* stream);
/* dx[N,H,W,Cin] (=|+=) conv_transpose(dy[N,Ho,Wo,Cout], w).  Cin must be a multiple of 64.  accumulate != 0 adds into dx. */
int yolo_conv2d_dgrad(const yolo_conv_problem* p, const void* dy, const void* w_dgrad, void* dx, int accumulate,
                      void* stream);
/* accumulate = 2 (yolo_conv2d_dgrad / _bn, 3x3 stride-2 problems run as four parity classes: yolo_conv2d_dgrad_classed(p) != 0): only the
 * even / even positions of dx hold a previous contribution -- the one a 1x1 stride-2 shortcut convolution's data gradient wrote with
 * yolo_conv2d_dgrad_even, which touches dx[:, ::2, ::2, :] only (the reference's down-sampling blocks: resnet18.py:31-52 feed one tensor to
 * a 3x3 / stride-2 and a 1x1 / stride-2 convolution; TF adds the two gradients densely).  Three quarters of the shortcut gradient are
 * structural zeros: they are neither written nor read back. */
int yolo_conv2d_dgrad_classed(const yolo_conv_problem* p);
int yolo_conv2d_dgrad_even(const yolo_conv_problem* p, const void* dy, const void* w_dgrad, void* dx, int accumulate, void* stream);
/* dx = addend + conv_transpose(dy, w): accumulate = 1 with the other contribution left where it was produced (no copy into dx first). */
int yolo_conv2d_dgrad_add(const yolo_conv_problem* p, const void* dy, const void* w_dgrad, void* dx, const void* addend, void* stream);
/* The same with the BatchNorm-backward REDUCE of the unit whose output gradient dx is, in the epilogue (the reference differentiates
 * Conv2D -> BatchNormalization -> ReLU chains by autodiff, basic_backbone.py:68-90; this is that chain's backward pass without the
 * separate statistics pass): dx receives the MASKED gradient g = (accumulated) dx where the unit's ReLU was positive (relu_mask: the sign
 * bytes of yolo_bn_act_fwd_mask, null for a linear unit), and partial[rows][3][Cin] (rows = yolo_conv2d_dgrad_bn_rows(p), allocated
 * ZEROED by the caller) the per-tile sums of g, g*xhat(y, mean, rstd) and, with y2, g*xhat(y2, mean2, rstd2).  yolo_bn_bwd_finalize over
 * `partial` and yolo_bn_act_bwd_apply(relu = 0) on dx finish the unit.  No grid barrier: safe next to collective kernels.
 * addend (may be NULL): the fan-in contribution is read from this buffer instead of dx (implies accumulate), see yolo_conv2d_dgrad_add.
 * yolo_conv2d_dgrad_bn_rows < 0: this problem cannot take the fused form (N*H*W*Cin >= 2^31). */
int yolo_conv2d_dgrad_bn_rows(const yolo_conv_problem* p);
/* Two-level partial rows (round 4): the convolution epilogues fold their per-tile rows in groups of 16-64 tiles -- the workgroup whose arrival
 * completes a group sums that group's rows in row order (deterministic; no workgroup waits for another) -- so that the BatchNorm kernel consuming
 * the statistics derives its constants from <= ~85 rows in its own prologue and the finalize launch between them disappears
 * (yolo_bn_finalize_act_fwd / yolo_bn_bwd_finalize_apply take that form for large tensors).  Replaces the same tf.keras BatchNormalization
 * statistics (basic_backbone.py:68-78) as the functions above.
 * *_group_layout: info4 = {rows the caller allocates, ZEROED ONCE (the kernels leave their arrival counters zero after every launch), group rows
 * P = rows [0, P) that hold the folded sums, group size (0: this problem keeps plain rows -- the RGB stem, stride-2 parity classes -- and the
 * *_g entry point behaves like the plain one), raw rows}.  yolo_conv2d_fwd_g = yolo_conv2d_fwd for a 16-bit output with statistics;
 * yolo_conv2d_dgrad_bn_g = yolo_conv2d_dgrad_bn; both with their rows laid out as the layout query says.  "row_group" tuning: 16 (default) / 0 / 8 / 32 / 64. */
int yolo_conv2d_stat_group_layout(const yolo_conv_problem* p, int32_t* info4);
int yolo_conv2d_fwd_g(const yolo_conv_problem* p, const void* src0, const void* src1, const void* w_fwd, void* y, float* stat_sum, float* stat_sq,
                      void* stream);
int yolo_conv2d_dgrad_bn_group_layout(const yolo_conv_problem* p, int32_t* info4);
int yolo_conv2d_dgrad_bn_g(const yolo_conv_problem* p, const void* dy, const void* w_dgrad, void* dx, int accumulate, const void* addend,
                           const void* relu_mask, const void* y, const float* mean, const float* rstd, const void* y2, const float* mean2,
                           const float* rstd2, float* partial, void* stream);
int yolo_conv2d_dgrad_bn(const yolo_conv_problem* p, const void* dy, const void* w_dgrad, void* dx, int accumulate, const void* addend,
                         const void* relu_mask, const void* y, const float* mean, const float* rstd, const void* y2,
                         const float* mean2, const float* rstd2, float* partial, void* stream);
/* dw[Cout][R][S][Cin] += x^T * dy (float32 atomics; the caller zeroes dw once per step).  split_k <= 0 = auto. */
int yolo_conv2d_wgrad(const yolo_conv_problem* p, const void* src0, const void* src1, const void* dy, float* dw,
                      int split_k, void* stream);
/* The same gradient without atomics (the training path): every pixel split stores its partial [Cout][R][S][Cin] slab into
 * `workspace` (>= yolo_conv2d_wgrad_workspace_bytes(p) bytes, 16-byte aligned, may be shared by consecutive calls on one stream),
 * a second launch sums the slabs: dw = (accumulate ? dw : 0) + sum.  Deterministic for a given problem. */
size_t yolo_conv2d_wgrad_workspace_bytes(const yolo_conv_problem* p);
int yolo_conv2d_wgrad_reduce(const yolo_conv_problem* p, const void* src0, const void* src1, const void* dy, float* dw,
                             void* workspace, size_t workspace_bytes, int accumulate, void* stream);
/* The training step's form: the slab pass alone into a PRIVATE region of a slab arena (yolo_conv2d_wgrad_splits(p) slabs of
 * Cout*R*S*Cin floats; with one split the kernel stores straight into dw and `slabs` is not touched), and ONE summing launch per gradient
 * bucket for all its layers once the bucket's backward pass is complete (the gradient exchange / optimizer of the bucket follow it).
 * table_dev: device int64 [nentries][5] = {first float4 of the layer's dw in `grads`, first float4 of its slabs in `arena`, float4s per
 * slab, number of slabs, first workgroup}; a workgroup covers 64 float4s of a layer, or 16 when the layer has YOLO_REDUCE_WIDE_SLABS or more
 * slabs (16 slab lanes instead of 4): total_blocks = sum of ceil(float4s / 64 or 16).  Below that threshold: the summation order of
 * yolo_conv2d_wgrad_reduce. */
#define YOLO_REDUCE_WIDE_SLABS 64
int yolo_conv2d_wgrad_splits(const yolo_conv_problem* p);
int yolo_conv2d_wgrad_slabs(const yolo_conv_problem* p, const void* src0, const void* src1, const void* dy, float* dw, float* slabs,
                            size_t slab_bytes, void* stream);
int yolo_wgrad_reduce_batched(const int64_t* table_dev, int nentries, int total_blocks, const float* arena, float* grads, void* stream);
/* bf16 [Cout][R][S][Cin] -> bf16 [Cin][R][S][Cout] with flipped taps (operand layout of yolo_conv2d_dgrad). */
int yolo_repack_dgrad_weights(const void* w_fwd, void* w_dgrad, int Cout, int R, int S, int Cin, void* stream);
/* the same for every layer in ONE launch.  table_dev: device int32 [nlayers][8] = {src element offset into w_fwd_flat, dst element
 * offset into w_dgrad_flat, Cout, R*S, Cin, first tile index, ceil(Cin/32), ceil(Cout/32)}; total_tiles = sum of R*S*tiles. */
int yolo_repack_dgrad_weights_batched(const void* w_fwd_flat, void* w_dgrad_flat, const int32_t* table_dev, int nlayers, int total_tiles,
                                      void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * BatchNorm (training mode) + ReLU + residual add, stem BN -> max-pool -> ReLU, and their backward passes.
 * Replace keras BatchNormalization (/root/reference/backbone/basic_backbone.py:75-77, momentum .9, eps 1e-5),
 * Activation('relu') (:89), layers.add (:124), MaxPooling2D(3, 2, 'same') (/root/reference/backbone/resnet18.py:60) and
 * their TF autodiff.  Tensors are bf16 [M][C] (M = N*H*W), C/8 a power of two <= 256; per-channel vectors are float32.
 * ------------------------------------------------------------------------------------------------------------------ */
/* number of partial rows the reduction kernels (yolo_bn_stats, *_bwd_reduce) write for an [M][C] tensor */
int yolo_reduce_rows(int M, int C);
/* partial[rows][2][C] = per-workgroup (sum x, sum x^2) of x[M][C]; for BatchNorms whose input is not a conv output */
int yolo_bn_stats(const void* x, int M, int C, float* partial, void* stream);
/* reduce P partial rows (row stride in floats) -> mean, rstd, scale = gamma*rstd, shift = beta - mean*scale; update the
 * moving statistics (NULL = skip).  gamma/beta NULL = 1/0. */
int yolo_bn_finalize(const float* psum, const float* psq, int P, int64_t row_stride, int C, float count, const float* gamma,
                     const float* beta, float eps, float momentum, float* moving_mean, float* moving_var, float* scale,
                     float* shift, float* mean, float* rstd, void* stream);
/* Small maps (a few hundred partial rows): yolo_bn_finalize + yolo_bn_act_fwd (out = act(y*scale + shift + res), relu_mask as
 * yolo_bn_act_fwd_mask, NULL = none) in ONE launch -- each of the two is nothing but the ~5 us floor of a dependent launch there.
 * Every workgroup reduces the partial rows of its 32 channels itself, so P should stay below a few hundred; C % 32 == 0; gamma / beta
 * are required.  Replaces the BatchNormalization + Activation(+ add) pair of reference backbone/basic_backbone.py:68-90,102-125. */
int yolo_bn_finalize_act_fwd(const float* psum, const float* psq, int P, int64_t row_stride, int C, float count, const float* gamma,
                             const float* beta, float eps, float momentum, float* moving_mean, float* moving_var, float* scale,
                             float* shift, float* mean, float* rstd, const void* y, const void* res, void* out, uint8_t* relu_mask,
                             int64_t M, int relu, void* stream);
/* Exact cross-workgroup accumulators: BatchNorm statistics WITHOUT partial rows and without a finalize launch.  A block holds, for Q
 * quantities of C channels, YOLO_ACC_NB = 8 buckets of two int64 limbs each (value * 2^20 = hi + lo * 2^-40) plus two trailing words (the first is the
 * non-finite / out-of-range flag); kernels add their
 * workgroup sums with 64-bit integer atomics, which are associative: the totals are bit-reproducible whatever the arrival order, unlike
 * float atomics.  yolo_acc_words(Q, C) = size of a block in 8-byte words; the caller zeroes all blocks once per step (yolo_zero_words)
 * before the first kernel that adds to them.  Replaces, like the functions above, the statistics of tf.keras BatchNormalization
 * (backbone/basic_backbone.py:68-78) and their TF autodiff. */
int64_t yolo_acc_words(int Q, int C);
int yolo_zero_words(int64_t* words, int64_t n, void* stream);
/* yolo_conv2d_fwd (16-bit output, no bias, not the RGB stem) adding sum / sum of squares of the stored outputs to stat_acc (Q = 2, C = Cout) */
int yolo_conv2d_fwd_acc(const yolo_conv_problem* p, const void* src0, const void* src1, const void* w_fwd, void* y, int64_t* stat_acc, void* stream);
/* yolo_bn_finalize_act_fwd reading the statistics from such a block: any number of pixel tiles upstream */
int yolo_bn_finalize_act_fwd_acc(const int64_t* stat_acc, int C, float count, const float* gamma, const float* beta, float eps, float momentum,
                                 float* moving_mean, float* moving_var, float* scale, float* shift, float* mean, float* rstd, const void* y,
                                 const void* res, void* out, uint8_t* relu_mask, int64_t M, int relu, void* stream);
/* yolo_conv2d_dgrad_bn with the tile sums added to an accumulator block (Q = 3, C = Cin) when partial is NULL, and yolo_bn_bwd_finalize_apply
 * reading such a block */
int yolo_conv2d_dgrad_bn_acc(const yolo_conv_problem* p, const void* dy, const void* w_dgrad, void* dx, int accumulate, const void* addend,
                             const void* relu_mask, const void* y, const float* mean, const float* rstd, const void* y2, const float* mean2,
                             const float* rstd2, float* partial, int64_t* acc, void* stream);
int yolo_bn_bwd_finalize_apply_acc(const int64_t* acc, int C, float count, float* dgamma, float* dbeta, float* k1, float* k2, const void* g,
                                   const void* y, const float* a1, const float* mean, const float* rstd, void* dy, int acc_dy, void* dres,
                                   int acc_dres, int64_t M, void* stream);
/* The same finalize for up to 4 BatchNorms over consecutive channel groups of ONE tensor (MixNet's grouped BN: shared statistics work
 * vectors, separate gamma / beta / moving statistics / gradient slots): split[0..ngroups] (host) = group boundaries, the pointer arrays
 * (host arrays of device pointers) have ngroups entries.  One launch instead of one per group. */
int yolo_bn_finalize_grouped(const float* psum, const float* psq, int P, int64_t row_stride, int C, float count, int ngroups,
                             const int32_t* split, const float* const* gamma, const float* const* beta, float eps, float momentum,
                             float* const* moving_mean, float* const* moving_var, float* scale, float* shift, float* mean, float* rstd,
                             void* stream);
int yolo_bn_bwd_finalize_grouped(const float* partial, int P, int64_t row_stride, int64_t q_stride, int C, int which, float count,
                                 int ngroups, const int32_t* split, float* const* dgamma, float* const* dbeta, float* k1, float* k2,
                                 void* stream);
/* out = act(y*scale + shift + T); T = 0 (res NULL), res (res_scale NULL) or res*res_scale + res_shift; scale NULL = identity */
int yolo_bn_act_fwd(const void* y, const float* scale, const float* shift, const void* res, const float* res_scale,
                    const float* res_shift, void* out, int64_t M, int C, int relu, void* stream);
/* The ReLU form that also leaves the activation's sign as a BYTE MASK, relu_mask[M][C/8] (bit j of byte (m, c/8) = channel 8*(c/8)+j of pixel m
 * is positive).  The backward entry points below accept it in place of `out` with relu = 2: they then read 1 byte instead of 16 per chunk
 * (BatchNorm backward is pure HBM traffic: dout + out + y in, dy out; the mask removes a quarter of it). */
int yolo_bn_act_fwd_mask(const void* y, const float* scale, const float* shift, const void* res, const float* res_scale,
                         const float* res_shift, void* out, uint8_t* relu_mask, int64_t M, int C, void* stream);
/* out[N,Ho,Wo,C] = act(maxpool3x3s2(y*scale + shift)); argmax[N,Ho,Wo,C] = window position 0..8 of the first maximum */
int yolo_bn_pool_fwd(const void* y, const float* scale, const float* shift, void* out, uint8_t* argmax, int N, int H, int W, int C,
                     int Ho, int Wo, int pad_t, int pad_l, int relu, void* stream);
/* g = dout * (out > 0 if relu); partial[rows][3][C] = (sum g, sum g*xhat(y), sum g*xhat(y2)) */
int yolo_bn_act_bwd_reduce(const void* dout, const void* out, int relu, const void* y, const float* mean, const float* rstd,
                           const void* y2, const float* mean2, const float* rstd2, int M, int C, float* partial, void* stream);
/* reduce partial rows (row_stride floats apart; quantity q starts q*q_stride floats into a row; the caller may offset `partial` to a
 * channel sub-range): dgamma = sum g*xhat, dbeta = sum g (NULL = skip), k1 = dbeta/count, k2 = dgamma/count; which = 1 (y) or 2 (y2) */
int yolo_bn_bwd_finalize(const float* partial, int P, int64_t row_stride, int64_t q_stride, int C, int which, float count,
                         float* dgamma, float* dbeta, float* k1, float* k2, void* stream);
/* Small maps: yolo_bn_bwd_finalize (which = 1) + yolo_bn_act_bwd_apply on an already masked gradient g (relu = 0), in ONE launch:
 * dgamma / dbeta (NULL = skip), k1, k2 from the [P][..] partial rows, then dy (=|+=) a1*(g - k1 - xhat*k2) and the optional shortcut
 * copy dres (=|+=) g.  C % 32 == 0.  (TF autodiff of the same reference lines.) */
int yolo_bn_bwd_finalize_apply(const float* partial, int P, int64_t row_stride, int64_t q_stride, int C, float count, float* dgamma,
                               float* dbeta, float* k1, float* k2, const void* g, const void* y, const float* a1, const float* mean,
                               const float* rstd, void* dy, int acc_dy, void* dres, int acc_dres, int64_t M, void* stream);
/* dy (=|+=) a1*(g - k1 - xhat*k2) (a1 NULL: dy = g); optional second BN branch -> dy2; optional dres (=|+=) g */
int yolo_bn_act_bwd_apply(const void* dout, const void* out, int relu, const void* y, const float* a1, const float* mean,
                          const float* rstd, const float* k1, const float* k2, void* dy, int acc_dy, const void* y2, const float* a2,
                          const float* mean2, const float* rstd2, const float* k1b, const float* k2b, void* dy2, void* dres,
                          int acc_dres, int64_t M, int C, void* stream);
/* The three calls above in ONE launch for a BN-carrying main branch (training hot path): the grid is resident (one 1024-thread
 * workgroup per CU), every workgroup keeps the masked gradient of its slice in registers, the last one to arrive finalizes
 * (dgamma / dbeta written, NULL = skip) and the apply runs from the held values, so dout / out are read once.  workspace: at least
 * yolo_bn_bwd_fused_workspace_floats(C) floats; sync_words: yolo_bn_bwd_fused_sync_words() ints zeroed once by the caller (grid-barrier
 * counters that only ever grow, plus a count of spin time-outs: read it with yolo_bn_fused_timeouts).  Returns 1 without launching when M*C is too large for the register-resident scheme (the caller
 * falls back to reduce / finalize / apply), 0 on success. */
int64_t yolo_bn_bwd_fused_workspace_floats(int C);
int yolo_bn_bwd_fused_sync_words(void);
int yolo_bn_act_bwd_fused(const void* dout, const void* out, int relu, int64_t M, int C, const void* y, const float* a1,
                          const float* mean, const float* rstd, float* dgamma, float* dbeta, void* dy, int acc_dy, const void* y2,
                          const float* a2, const float* mean2, const float* rstd2, float* dgamma2, float* dbeta2, void* dy2,
                          void* dres, int acc_dres, float* workspace, int* sync_words, void* stream);
/* the same with a grouped BatchNorm as the main branch (dgamma / dbeta: host arrays of ngroups device pointers, split as above) */
int yolo_bn_act_bwd_fused_grouped(const void* dout, const void* out, int relu, int64_t M, int C, const void* y, const float* a1,
                                  const float* mean, const float* rstd, int ngroups, const int32_t* split, float* const* dgamma,
                                  float* const* dbeta, void* dy, int acc_dy, const void* y2, const float* a2, const float* mean2,
                                  const float* rstd2, float* dgamma2, float* dbeta2, void* dy2, void* dres, int acc_dres,
                                  float* workspace, int* sync_words, void* stream);
int yolo_bn_fused_timeouts(const int* sync_words, int* host_out);
/* Register a HOST-visible word (pinned, device-accessible memory; NULL removes it): a grid-barrier time-out additionally stores 1 there with
 * system scope, so the owner can refuse to enqueue the next step without synchronising the device first. */
int yolo_bn_fused_set_host_flag(int* sync_words, int* host_flag);
/* the same two passes through the stem's max-pool (rows = pre-pool pixels N*H*W) */
/* (with relu and non-NULL gamma/beta the sums are taken over the pooled map: xhat = (out - beta) / gamma, y/argmax are not read) */
int yolo_bn_pool_bwd_reduce(const void* dout, const void* out, const uint8_t* argmax, int relu, const void* y, const float* mean,
                            const float* rstd, const float* gamma, const float* beta, int N, int H, int W, int C, int Ho, int Wo,
                            int pad_t, int pad_l, float* partial, void* stream);
int yolo_bn_pool_bwd_apply(const void* dout, const void* out, const uint8_t* argmax, int relu, const void* y, const float* a1,
                           const float* mean, const float* rstd, const float* k1, const float* k2, void* dy, int N, int H, int W, int C,
                           int Ho, int Wo, int pad_t, int pad_l, void* stream);
/* The stem's whole backward pass behind the reduce / finalize pair above, WITHOUT materialising the pre-pool gradient: un-pooling of
 * dout (ReLU-masked by `out` if relu), BatchNorm apply (a1 == NULL: no BatchNorm, dy = un-pooled gradient) and the stem convolution's
 * weight gradient in one persistent kernel (csrc/stem_bwd.hip).  The stem convolution p (RGB padded to 8 channels -> 64, 3x3, stride 2,
 * even size) has no data gradient, so nothing else reads dy.  slabs: yolo_stem_pool_bwd_slabs(...) float32 slabs of [64][3][3][8]
 * (0 = this stem is not covered: use yolo_bn_pool_bwd_apply + yolo_conv2d_wgrad_slabs), summed by yolo_wgrad_reduce_batched like any
 * other layer's.  x: the packed image [N][p->H][p->W][8]; y: the stem convolution's output [N][p->Ho][p->Wo][64]; dout / out / argmax:
 * the pooled map [N][Ho][Wo][64].  Replaces autodiff of /root/reference/backbone/resnet18.py:59-61, resnet18_v2.py:61-62, mixnet18.py:72. */
int yolo_stem_pool_bwd_slabs(const yolo_conv_problem* p, int pooled_channels, int Ho, int Wo, int pad_t, int pad_l);
int yolo_stem_pool_bwd_wgrad(const yolo_conv_problem* p, const void* x, const void* dout, const void* out, const uint8_t* argmax, int relu,
                             const void* y, const float* a1, const float* mean, const float* rstd, const float* k1, const float* k2,
                             int Ho, int Wo, int pad_t, int pad_l, float* slabs, size_t slab_bytes, void* stream);
/* gradient of concat(upsample2x(a), b): da[N,H/2,W/2,C0] (=|+=) 2x2 sums of dcat[..., :C0]; db[N,H,W,C1] (=|+=) dcat[..., C0:]
 * (/root/reference/yolov3/yolov3_detector.py:115-116,140-141) */
int yolo_upcat_split_bwd(const void* dcat, void* da, int acc_a, void* db, int acc_b, int N, int H, int W, int C0, int C1, void* stream);
/* out[c] = sum over P partial rows (row stride in floats) of column c (bias gradient of the detection convs) */
int yolo_reduce_partials(const float* partial, int P, int64_t row_stride, int C, float* out, void* stream);
/* inference-mode BatchNorm (keras learning_phase False, /root/reference/run.py:21-24): scale/shift from the moving statistics */
int yolo_bn_eval_scale_shift(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var, float eps,
                             float* scale, float* shift, int C, void* stream);
/* float32 NHWC images (C = 3, [0,1], BGR: /root/reference/dataset/file_util.py:58-59) -> bf16 NHWC8, channels 3..7 zero */
int yolo_pack_input(const float* images, void* out, int64_t npix, int Cimg, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Mixed depthwise convolution of MixNet-18: channel groups split[g]..split[g+1] use a ksize[g] x ksize[g] depthwise kernel
 * (stride 1, 'same').  Replaces the Lambda slices + 4 DepthwiseConv2D + Concatenate of /root/reference/backbone/mixnet18.py:38-45
 * (factory /root/reference/backbone/basic_backbone.py:45-66) and their TF gradients.  x, y: bf16 [N,H,W,C]; w_g: bf16
 * [k][k][C_g]; dw_g: float32 = (accumulate ? dw_g : 0) + gradient; the weight gradient is two-phase (per-workgroup slabs in `workspace`,
 * >= yolo_dwconv_mix_wgrad_workspace_bytes(p) bytes, then one summing launch): deterministic, no atomics.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t N, H, W, C;
  int32_t split[5];    /* 0 = split[0] <= ... <= split[4] = C, group sizes multiples of 8 (and /8 a power of two) */
  int32_t ksize[4];    /* odd, <= 9 */
} yolo_mixconv_problem;
int yolo_dwconv_mix_fwd(const yolo_mixconv_problem* p, const void* x, const void* w0, const void* w1, const void* w2, const void* w3,
                        void* y, void* stream);
int yolo_dwconv_mix_dgrad(const yolo_mixconv_problem* p, const void* dy, const void* w0, const void* w1, const void* w2, const void* w3,
                          void* dx, int accumulate, void* stream);
size_t yolo_dwconv_mix_wgrad_workspace_bytes(const yolo_mixconv_problem* p);
int yolo_dwconv_mix_wgrad(const yolo_mixconv_problem* p, const void* x, const void* dy, float* dw0, float* dw1, float* dw2, float* dw3,
                          void* workspace, size_t workspace_bytes, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * YOLOv3 loss forward + backward.  Replaces YOLOv3Decoder.decode (/root/reference/yolov3/yolov3_decoder.py:62-192),
 * LabelDecoder.decode (/root/reference/yolov3/label_decoder.py:26-60), YOLOv3Loss.loss
 * (/root/reference/yolov3/yolov3_loss.py:81-369) and the TF gradient of the loss w.r.t. the head logits.
 * ------------------------------------------------------------------------------------------------------------------ */
#define YOLO_MAX_ANCHORS 8
typedef struct {
  int32_t H[3], W[3], B[3];          /* grid size and anchors per head, order /8, /16, /32 */
  int32_t ldc[3];                    /* channel stride of the logits rows (>= B*L; conv outputs are padded) */
  float anchor_w[3][YOLO_MAX_ANCHORS], anchor_h[3][YOLO_MAX_ANCHORS];   /* anchors in grid units (yolov3_decoder.py:38-40) */
  int32_t L;                         /* box_len = 5 + class_num (configs.py:44) */
  int32_t T;                         /* label slots per image; labels are float32 [N][T][5], padded with -1 */
  float iou_thresh;                  /* configs.py:50 */
  float w_xy[3], w_wh[3], w_noobj[3], w_obj[3], w_cls[3];   /* configs.py:52, transposed per head */
  float w_rect[3];                   /* rectified_loss_weight (configs.py:59) */
  int32_t rectified_coord_num;       /* configs.py:58; -1 disables */
  int32_t is_focal_loss;
  float focal_alpha, focal_gamma;
  int32_t is_tiou_recall;
  float eps;                         /* K.epsilon() = 1e-8 (run.py:26) */
  float grad_scale16;                /* factor on the 16-bit d(logits) copies only (static loss scaling for fp16 training; the float32
                                        d(logits), the loss terms and the total are never scaled); 0 is read as 1 */
} yolo_loss_config;

int64_t yolo_loss_workspace_bytes(const yolo_loss_config* c, int N);
/* logits*: float32 [N][H][W][ldc].  dlogits* (float32, may be NULL) and dlogits*_bf16 (may be NULL) get d(total)/d(logits)
 * in the same layout (padding channels are written as zeros).  current_num: device int32 rectified-image counter (advances by
 * batch_global while active).  terms: float32 [6][3] (rows xy, wh, noobj, obj, class, rectified); total: float32 [1].
 * assign_out (int32 [N][T][3], may be NULL): flat index (row*W+col)*B+anchor of the responsible prediction or -1;
 * resp_iou_out (float32 [N][T][3], may be NULL). */
int yolo_loss_fwd_bwd(const yolo_loss_config* c, int N, int batch_global, const float* logits8, const float* logits16,
                      const float* logits32, const float* labels, float* dlogits8, float* dlogits16, float* dlogits32,
                      void* dlogits8_bf16, void* dlogits16_bf16, void* dlogits32_bf16, int* current_num, float* terms, float* total,
                      int* assign_out, float* resp_iou_out, void* workspace, void* stream);

/* inference-side decode of one head (YOLOv3Decoder._decode_single_head, /root/reference/yolov3/yolov3_decoder.py:119-192, plus the score /
 * arg-max class of /root/reference/yolov3/yolov3_post_process.py:53-59).  anchors_grid: device float32 [B][2] (w, h) in grid units.
 * Outputs (device, any may be NULL): decoded [N][H][W][B][L], boxes [N][H][W][B][4] corners, score [N][H][W][B], cls_idx int32. */
int yolo_decode_head(const float* logits, int N, int H, int W, int B, int L, int ldc, const float* anchors_grid, float eps, float* decoded,
                     float* boxes, float* score, int* cls_idx, void* stream);

/* inference-side box selection, per head: YOLOv3PostProcessor._filter_single_head_boxes
 * (/root/reference/yolov3/yolov3_post_process.py:45-77).  prediction: decoded float32 [N][H][W][B][L] (what yolo_decode_head writes),
 * boxes: corners [N][H][W][B][4] in grid units.  For each image, every prediction with score = conf * max class prob (conf alone
 * when L == 5) > score_thresh is appended in flat (row, col, anchor) order -- np.where order -- to rows[n][k][8] =
 * {x0/W, y0/H, x1/W, y1/H, conf, class prob, class index, score} (float32 arithmetic as in the reference) and its flat index to
 * index[n][k]; counts[n] = number of hits, which may exceed cap (only the first cap are stored: the caller must check). */
int yolo_filter_boxes(const float* prediction, const float* boxes, int N, int H, int W, int B, int L, float score_thresh, int cap,
                      int* counts, float* rows, int* index, void* stream);
/* cross-head class-wise greedy NMS: YOLOv3PostProcessor.apply_nms / _apply_nms / _cal_iou (yolov3_post_process.py:79-162), one
 * workgroup per image over the rows of yolo_filter_boxes for the three heads: stable sort by descending score over the
 * concatenation /8, /16, /32; a box is suppressed by a better box of the same class whose IoU (float64, no fma) > nms_thresh;
 * keepX[n][k] = 1 iff row k's id is among the survivors' ids.  Ids are per-head positions, as in the reference (start_index is
 * never advanced, :81-89), unless fixed_indices.  If an image has more than yolo_nms_max_candidates() rows in total, or a count
 * exceeds cap, nothing is written for it and *status (device int32, zero it first) receives the offending count. */
int yolo_nms_max_candidates(void);
int yolo_nms_heads(const float* rows8, const float* rows16, const float* rows32, const int* count8, const int* count16, const int* count32,
                   int N, int cap, double nms_thresh, int fixed_indices, uint8_t* keep8, uint8_t* keep16, uint8_t* keep32, int* status,
                   void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Input pipeline (the row before the hot path).  Replaces, for one batch of decoded images,
 * tf.image.resize_image_with_pad(NEAREST) + convert_image_dtype + tf.reverse (/root/reference/dataset/file_util.py:54-59) and
 * DatasetUtil._augment (/root/reference/dataset/dataset_util.py:29-99).  JPEG decoding stays on the host.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
  int64_t offset;            /* byte offset of this image's uint8 RGB HWC pixels in src */
  int32_t h, w;              /* decoded size */
  int32_t nh, nw, top, left; /* letterbox geometry: resized size and position inside the H x W canvas (bars are zero) */
  int32_t noise;             /* 0 salt-and-pepper p = 0.01, 1 gaussian sigma = 0.01, anything else none (dataset_util.py:47-56) */
  int32_t color_order;       /* 0 brightness,saturation,contrast; 1 saturation,brightness,contrast; 2 saturation,contrast,brightness;
                                anything else none (dataset_util.py:58-96) */
  float brightness_delta, saturation_factor, contrast_factor;   /* the scalar draws of tf.image.random_* (host side) */
  uint32_t seed0, seed1;     /* Philox4x32-10 key of the per-pixel noise; counter = (pixel index, image index, 0, 0) */
  int32_t reserved;
} yolo_image_desc;

int64_t yolo_letterbox_workspace_bytes(int N);
/* src: device uint8; desc: device array [N].  out_f32 (float32 [N][H][W][3], BGR in [0,1]) and / or out_bf16x8 (the packed conv input
 * [N][H][W][8], channels 3..7 zero) -- at least one.  augment = 0 skips noise / colour / clip (test and predict modes). */
int yolo_letterbox_augment(const uint8_t* src, const yolo_image_desc* desc, int N, int H, int W, int augment, void* workspace,
                           float* out_f32, void* out_bf16x8, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * RAdam + L2 regularisation over the flat parameter buffer.  Replaces RAdam.get_updates
 * (/root/reference/utils/radam.py:56-107) and the Keras L2 regularisers (/root/reference/backbone/basic_backbone.py:41,64,76).
 * sched: device float32 [4] = {lr (host-set), lr_t, rho_t, update rule of this step (0 first moment only = RAdam warm-up, 1 adaptive,
 * 2 / 3 SGD momentum with / without Nesterov)}; iterations: device int64 [1].
 * ------------------------------------------------------------------------------------------------------------------ */
int yolo_radam_schedule(float* sched, int64_t* iterations, float beta1, float beta2, float decay, float warmup_coef, void* stream);
/* The reference trainer's two other optimizers (/root/reference/yolov3/trainer.py:70-73): kind 1 = keras Adam (amsgrad when the step gets a
 * vhat buffer), 2 = keras SGD momentum + Nesterov (beta1 = the momentum, m = the velocity), 3 = keras SGD momentum; 0 = RAdam with
 * warmup_coef 1.  Fills the same sched block, whose 4th word selects the update rule inside yolo_radam_l2_step. */
int yolo_optimizer_schedule(float* sched, int64_t* iterations, int kind, float beta1, float beta2, float decay, void* stream);
int yolo_radam_l2_blocks(int64_t n);   /* length of l2_partial */
/* n (multiple of 256) elements; l2_table[n/256] = lambda of each 256-element chunk; grads are multiplied by grad_scale and
 * zeroed afterwards if zero_grad; params_bf16 (may be NULL) receives the bf16 copy; vhat (may be NULL) enables AMSGrad;
 * l2_partial (may be NULL) receives per-workgroup sums of lambda*p^2 at the pre-update weights; nonfinite (device int32, may be NULL):
 * gradient elements that are inf / NaN (fp16 overflow) are not applied (treated as 0) and *nonfinite is incremented once per wave
 * that saw one -- the caller checks it and lowers the loss scale / fails loudly. */
int yolo_radam_l2_step(float* params, float* grads, float* m, float* v, float* vhat, void* params_bf16, const float* l2_table,
                       int64_t n, const float* sched, float beta1, float beta2, float eps, float grad_scale, int zero_grad,
                       float* l2_partial, int* nonfinite, void* stream);
int yolo_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream);
/* out[0] = sum of partial[0..n) (+ add[0] if add != NULL); out_plain[0] (may be NULL) = the sum without `add`: the reported loss (YOLOv3 loss +
 * L2 terms) and the L2 terms alone from one launch */
int yolo_sum_partials(const float* partial, int n, const float* add, float* out, float* out_plain, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLOV3_AMD_H_ */
