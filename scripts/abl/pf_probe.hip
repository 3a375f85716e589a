// Which form of an LDS-DMA "prefetch" is legal?  One variant per process (argv[1]); prints "ok <variant>".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int V>
__global__ void k(const float* buf, unsigned* out) {
  extern __shared__ __align__(16) unsigned char lds[];
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + (V == 4 ? 131072u : 0u));
  const unsigned stride = (V == 0) ? 16u : 512u;
  const unsigned off = (blockIdx.x * 4 + wave) * 32768u + lane * stride;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(dst) : "memory");
  if (V == 0 || V == 1 || V == 4) asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(buf) : "memory");
  if (V == 2) asm volatile("global_load_lds_dwordx4 %0, %1 offset:128" ::"v"(off), "s"(buf) : "memory");
  if (V == 3) asm volatile("global_load_lds_dword %0, %1 offset:128" ::"v"(off), "s"(buf) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = reinterpret_cast<unsigned*>(lds)[V == 4 ? 32768 : 0];
}
int main(int argc, char** argv) {
  const int v = argc > 1 ? atoi(argv[1]) : 0;
  float* buf; unsigned* out;
  hipMalloc(&buf, 64u << 20); hipMemset(buf, 0, 64u << 20); hipMalloc(&out, 4096);
  const size_t ldsb = 150 * 1024;
#define RUN(V) { hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb); hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), ldsb, 0, buf, out); }
  switch (v) { case 0: RUN(0) break; case 1: RUN(1) break; case 2: RUN(2) break; case 3: RUN(3) break; default: RUN(4) break; }
  hipError_t e = hipDeviceSynchronize();
  printf("%s variant %d (%s)\n", e == hipSuccess ? "ok" : "FAILED", v, hipGetErrorString(e));
  return e != hipSuccess;
}
