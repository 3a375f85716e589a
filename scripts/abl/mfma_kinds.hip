// Micro-benchmark (round 3): issue cost of the MFMA shapes the KPConv phase 1 could use, one wave per SIMD and two:
//   v_mfma_f32_16x16x4_f32 (exact f32, what k_kpconv_ring uses), v_mfma_f32_16x16x16_f16 (legacy K = 16),
//   v_mfma_f32_16x16x32_f16; and the same loops with 8 independent v_fma_f32 per MFMA from the SAME wave
//   (does vector work hide under matrix work inside a wave?).
// Build: hipcc -O3 --offload-arch=gfx950 mfma_kinds.hip -o mfma_kinds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int KIND, int VALU>
__global__ __launch_bounds__(512) void k(const float* __restrict__ src, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63;
  float a = src[lane], b = src[64 + lane];
  h4 a4 = {(_Float16)a, (_Float16)b, (_Float16)a, (_Float16)b};
  h8 a8 = {(_Float16)a, (_Float16)b, (_Float16)a, (_Float16)b, (_Float16)a, (_Float16)b, (_Float16)a, (_Float16)b};
  f4 c[4];
  for (int j = 0; j < 4; ++j) c[j] = (f4){0.f, 0.f, 0.f, 0.f};
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = a + j;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (KIND == 0) c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[j], 0, 0, 0);
      if (KIND == 1) c[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, a4, c[j], 0, 0, 0);
      if (KIND == 2) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, a8, c[j], 0, 0, 0);
      if (VALU) {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_fmaf(v[u], 1.0001f, b);
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
  for (int u = 0; u < 8; ++u) s += v[u];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int VALU>
void run(const char* name, int threads, float* d, float* o) {
  const int iters = 2048, blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, VALU>), dim3(blocks), dim3(threads), 0, 0, d, o, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, VALU>), dim3(blocks), dim3(threads), 0, 0, d, o, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // one workgroup per CU; waves per SIMD = threads / 256; MFMAs per wave = 4 * iters
  const double ns_per_mfma_per_simd = ms * 1e6 / (4.0 * iters * (threads / 256));
  printf("%-34s waves/SIMD %d  %.3f ms  -> %.1f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz)\n", name, threads / 256, ms,
         ns_per_mfma_per_simd, ns_per_mfma_per_simd * 2.4);
}

int main() {
  float *d, *o;
  hipMalloc(&d, 128 * 4); hipMalloc(&o, 256 * 512 * 4);
  float h[128]; for (int i = 0; i < 128; ++i) h[i] = 0.001f * (i % 37) - 0.01f;
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int threads : {256, 512}) {
    if (threads == 256) {
      run<0, 0>("16x16x4 f32", 256, d, o); run<1, 0>("16x16x16 f16", 256, d, o); run<2, 0>("16x16x32 f16", 256, d, o);
      run<0, 1>("16x16x4 f32 + 8 v_fma each", 256, d, o); run<1, 1>("16x16x16 f16 + 8 v_fma each", 256, d, o);
      run<2, 1>("16x16x32 f16 + 8 v_fma each", 256, d, o);
    } else {
      run<0, 0>("16x16x4 f32", 512, d, o); run<1, 0>("16x16x16 f16", 512, d, o); run<2, 0>("16x16x32 f16", 512, d, o);
      run<0, 1>("16x16x4 f32 + 8 v_fma each", 512, d, o); run<1, 1>("16x16x16 f16 + 8 v_fma each", 512, d, o);
      run<2, 1>("16x16x32 f16 + 8 v_fma each", 512, d, o);
    }
  }
  return 0;
}
