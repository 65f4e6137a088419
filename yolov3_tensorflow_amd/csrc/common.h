// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the YOLOv3 training hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/yolov3_amd.h"
#include "launch.h"

// The 16-bit activation / compute-copy element.  Default build: bfloat16.  -DYOLO_FP16 (libyolov3_amd_fp16.so): IEEE half -- same
// kernels, same layouts; only the conversions below and the MFMA opcode differ.  The names keep "bf16" in both builds.
typedef uint16_t bf16_t;  // storage type of one element
#ifdef YOLO_FP16
typedef _Float16 bf16x8_t __attribute__((ext_vector_type(8)));
#define YOLO_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define YOLO_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#else
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
#define YOLO_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define YOLO_MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

#define YOLO_WAVE 64

// ---- error plumbing (C-ABI returns int status; message retrievable with yolo_last_error) ----
void yolo_set_error(const char* fmt, ...);
#define YOLO_CHECK_ARG(cond, msg)                                   \
  do {                                                              \
    if (!(cond)) {                                                  \
      yolo_set_error("%s:%d: %s", __FILE__, __LINE__, msg);        \
      return YOLO_ERR_INVALID_ARG;                                  \
    }                                                               \
  } while (0)
#define YOLO_LAUNCH_CHECK()                                                              \
  do {                                                                                   \
    hipError_t e_ = hipGetLastError();                                                   \
    if (e_ != hipSuccess) {                                                              \
      yolo_set_error("%s:%d: launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return (int)e_;                                                                    \
    }                                                                                    \
  } while (0)

// ---- dedicated stem convolution (stem.hip), dispatched from yolo_conv2d_fwd / yolo_conv2d_stat_rows ----
bool yolo_stem_applies(const yolo_conv_problem* p);
int yolo_stem_stat_rows(const yolo_conv_problem* p);
int yolo_stem_fwd(const yolo_conv_problem* p, const void* x, const void* w, void* y, float* stat_sum, float* stat_sq, void* stream);
int yolo_stem_set_direct(int on);
int yolo_dw_set_tiled(int on);       // dwconv.hip

// ---- 16-bit element <-> f32 (round-to-nearest-even; bf16: v_cvt_pk_bf16_f32, fp16: v_cvt_f16_f32 / v_cvt_f32_f16) ----
#ifdef YOLO_FP16
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  _Float16 h = (_Float16)f;
  return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ float lo2f(uint32_t w) { return bf2f((bf16_t)(w & 0xffffu)); }
__device__ __forceinline__ float hi2f(uint32_t w) { return bf2f((bf16_t)(w >> 16)); }
#else
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float lo2f(uint32_t w) { return __uint_as_float(w << 16); }          // element in the low / high half of a dword
__device__ __forceinline__ float hi2f(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
#endif
#ifdef YOLO_FP16
__device__ __forceinline__ uint32_t pack_bf2(float a, float b) { return (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16); }
#else
// one v_cvt_pk_bf16_f32 for the pair (the scalar form above costs two conversions and an SDWA or)
__device__ __forceinline__ uint32_t pack_bf2(float a, float b) {
  typedef float yolo_f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 yolo_bf16x2_t __attribute__((ext_vector_type(2)));
  const yolo_f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, yolo_bf16x2_t));
}
#endif
__device__ __forceinline__ void unpack_bf8(const uint4& v, float* f) {
  f[0] = lo2f(v.x); f[1] = hi2f(v.x);
  f[2] = lo2f(v.y); f[3] = hi2f(v.y);
  f[4] = lo2f(v.z); f[5] = hi2f(v.z);
  f[6] = lo2f(v.w); f[7] = hi2f(v.w);
}
__device__ __forceinline__ uint4 pack_bf8(const float* f) {
  uint4 v;
  v.x = pack_bf2(f[0], f[1]); v.y = pack_bf2(f[2], f[3]); v.z = pack_bf2(f[4], f[5]); v.w = pack_bf2(f[6], f[7]);
  return v;
}

// ---- exact cross-workgroup accumulators (BatchNorm statistics without a finalize launch) ----
// A float value is added as TWO int64 limbs (value * 2^20 = hi + lo * 2^-40, lo in [0, 2^40)): integer atomics are associative, so the
// total does not depend on the order in which workgroups arrive -- bit-reproducible, unlike float atomics -- and carries 60 fractional
// bits (the float32 partial sums are represented exactly unless they are below 2^-36).  Layout of one accumulator block:
// [YOLO_ACC_NB = 8 buckets][Q quantities][2 limbs][C channels] int64, then two words, the first a flag (non-zero: a non-finite value, or one whose
// magnitude the fixed-point limbs cannot hold, |v| >= 2^42, was added; the consumer then produces NaN -- the float path would have produced inf / NaN or
// a sum of no significance).  Buckets (workgroup index mod YOLO_ACC_NB) spread the same-address contention.
#define YOLO_ACC_NB 8
__host__ __device__ inline size_t yolo_acc_block_words(int Q, int C) { return (size_t)YOLO_ACC_NB * Q * 2 * C + 2; }
__device__ __forceinline__ void yolo_acc_add(long long* block, int Q, int C, int bucket, int q, int c, float v) {
  // inf / NaN, or beyond the limbs' range: v * 2^20 must stay far inside int64 (2^63) also after thousands of adds into one bucket
  if (!(fabsf(v) < 4398046511104.0f)) {                // 2^42
    atomicOr(reinterpret_cast<unsigned long long*>(block + (size_t)YOLO_ACC_NB * Q * 2 * C), 1ull);
    return;
  }
  const double d = (double)v * 1048576.0;              // 2^20
  const double fl = floor(d);
  const long long hi = (long long)fl, lo = (long long)((d - fl) * 1099511627776.0);    // 2^40
  long long* p = block + ((size_t)(bucket * Q + q) * 2) * C + c;
  atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)hi);
  atomicAdd(reinterpret_cast<unsigned long long*>(p + C), (unsigned long long)lo);
}
// the total of quantity q, channel c over all buckets (NaN if the flag is raised)
__device__ __forceinline__ double yolo_acc_total(const long long* block, int Q, int C, int q, int c) {
  long long hi = 0, lo = 0;
#pragma unroll 4
  for (int b = 0; b < YOLO_ACC_NB; ++b) {
    const long long* p = block + ((size_t)(b * Q + q) * 2) * C + c;
    hi += p[0];
    lo += p[C];
  }
  const double t = (double)hi * (1.0 / 1048576.0) + (double)lo * (1.0 / 1152921504606846976.0);     // 2^-20, 2^-60
  return block[(size_t)YOLO_ACC_NB * Q * 2 * C] ? __builtin_nan("") : t;
}

// ---- wave / block reductions (64-wide wavefronts) ----
// sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), result in every lane: quad butterflies, then half-row and row mirrors
// (each step is one v_add_f32 with a DPP modifier -- no LDS traffic, unlike __shfl_xor -> ds_bpermute_b32)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
