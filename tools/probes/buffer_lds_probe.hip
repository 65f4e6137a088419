// Probe: does `buffer_load_dwordx4 ... lds` (LDS-DMA through a buffer descriptor) write zeros for out-of-range lanes on gfx950?
// Build: hipcc --offload-arch=gfx950 -O3 -o buffer_lds_probe buffer_lds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lptr_t;
__global__ void k(const char* src, int nbytes, int* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) ((int*)smem)[i] = 0x7f7f7f7f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int off = (wave * 64 + lane) * 16;
  if (lane % 3 == 1) off = -1;            // far out of range
  if (lane % 3 == 2) off = nbytes - 8;    // straddles the end
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(smem + wave * 1024), 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = ((int*)smem)[i];
}
int main() {
  const int n = 4096;
  std::vector<int> h(n / 4);
  for (int i = 0; i < n / 4; ++i) h[i] = i + 1;
  char* d; int* o;
  hipMalloc(&d, n); hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
  k<<<1, 256, 4096>>>(d, n, o);
  std::vector<int> r(1024);
  hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 256; ++t) {
    int lane = t & 63;
    for (int j = 0; j < 4; ++j) {
      int got = r[t * 4 + j], want;
      if (lane % 3 == 0) want = t * 4 + j + 1;
      else if (lane % 3 == 1) want = 0;
      else want = -2;   // report only
      if (want == -2) { if (t < 6) printf("straddle lane %d word %d: %d\n", t, j, got); }
      else if (got != want) { if (bad < 8) printf("MISMATCH t %d j %d got %d want %d\n", t, j, got, want); ++bad; }
    }
  }
  printf("bad = %d\n", bad);
  return bad != 0;
}
