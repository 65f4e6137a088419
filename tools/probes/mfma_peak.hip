// Sustained dense bf16 MFMA rate of the whole chip with operands in registers (no memory traffic): what "100 % MFMA" is at the clock the
// part actually holds under that load.   hipcc --offload-arch=gfx950 -O3 -o mfma_peak.bin mfma_peak.hip && ./mfma_peak.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int KIND>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float seed) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + threadIdx.x * 0.001f + j); b[j] = (__bf16)(seed - j * 0.5f); }
  if (KIND == 0) {
    f32x4 acc[8];
    bf16x8 av[2] = {a, b}, bv[2] = {b, a};          // distinct operand pairs and start values: the accumulators must stay separate registers
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{(float)t, 0.f, seed, 0.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int t = 0; t < 8; ++t) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(av[t & 1]), "v"(bv[t >> 2]));   // (in place: the builtin let hipcc rotate the accumulators through overlapping registers)
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else if (KIND == 2) {       // 16x16x32 with dst != srcC (disjoint, aligned register groups): does renaming cost anything?
    f32x4 acc[8], acc2[8];
    bf16x8 av[2] = {a, b}, bv[2] = {b, a};
    for (int t = 0; t < 8; ++t) { acc[t] = f32x4{(float)t, 0.f, seed, 0.f}; acc2[t] = acc[t]; }
    for (int i = 0; i < iters; i += 2) {
#pragma unroll
      for (int t = 0; t < 8; ++t) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=a"(acc2[t]) : "v"(av[t & 1]), "v"(bv[t >> 2]), "a"(acc[t]));
#pragma unroll
      for (int t = 0; t < 8; ++t) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=a"(acc[t]) : "v"(av[t & 1]), "v"(bv[t >> 2]), "a"(acc2[t]));
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int j = 0; j < 16; ++j) s += acc[t][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}

int main() {
  float* out;
  const int blocks = 256 * 4;          // 4 workgroups of 4 waves per CU: 4 waves per SIMD
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int kind = 0; kind < 3; ++kind) {
    for (int rep = 0; rep < 4; ++rep) {
      const int iters = 20000;
      hipEventRecord(e0, 0);
      if (kind == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
      else if (kind == 2) hipLaunchKernelGGL(mfma_loop<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
      else           hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      const double mf = kind != 1 ? 8.0 * 2 * 16 * 16 * 32 : 4.0 * 2 * 32 * 32 * 16;      // FLOP per wave per iteration
      const double flops = mf * iters * (double)blocks * 4;
      printf("%s: %.3f ms  %.1f TFLOP/s  (%.2f GHz-equivalent of the 2.5 PFLOP/s at 2.4 GHz)\n", kind == 0 ? "16x16x32 in place" : (kind == 2 ? "16x16x32 dst != srcC" : "32x32x16"), ms,
             flops / ms / 1e9, flops / ms / 1e9 / 2500.0 * 2.4);
    }
  }
  return 0;
}
