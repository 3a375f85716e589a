// Micro-benchmark: v_mfma_f32_32x32x16_f16 issue rate vs operand values
// (zeros / normal fp16 / fp16 subnormals).  Build: hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k(const _Float16* __restrict__ src, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x;
  h16x8 a = *reinterpret_cast<const h16x8*>(src + (size_t)lane * 16);
  h16x8 b = *reinterpret_cast<const h16x8*>(src + (size_t)lane * 16 + 8);
  f32x16 c0, c1, c2, c3;
  for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c3[r] = 0.f;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  const int iters = 4096, blocks = 256 * 2;
  _Float16* d; float* o;
  hipMalloc(&d, 256 * 16 * 2); hipMalloc(&o, blocks * 256 * 4);
  const char* names[] = {"zeros", "normal", "subnormal", "mixed hi*lo"};
  for (int mode = 0; mode < 4; ++mode) {
    std::vector<_Float16> h(256 * 16);
    for (size_t i = 0; i < h.size(); ++i) {
      float u = (rand() / (float)RAND_MAX) * 2.f - 1.f;
      float v = mode == 0 ? 0.f : mode == 1 ? u : mode == 2 ? u * 3e-5f : ((i / 8) & 1 ? u * 3e-5f : u);
      h[i] = (_Float16)v;
    }
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, o, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 4 * 32 * 32 * 16 * 2;
    printf("%-12s %.3f ms  %.1f TFLOP/s\n", names[mode], ms, flops / ms * 1e-9);
  }
  return 0;
}
