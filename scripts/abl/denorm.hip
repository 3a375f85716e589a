#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float* out, float aval, float bval) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)aval; b[j] = (_Float16)bval; }
  f32x16 c; for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
  float* d; hipMalloc(&d, 4); float h;
  const float vals[][2] = {{1e-6f, 1.f}, {3e-8f, 1.f}, {1e-6f, 1e-6f}, {1e-3f, 1e-6f}, {6e-5f, 1.f}};
  for (auto& v : vals) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, v[0], v[1]);
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("a=%g b=%g  fp16(a)=%g  mfma sum16=%g  expected=%g\n", v[0], v[1], (float)(_Float16)v[0], h, 16.0 * (float)(_Float16)v[0] * (float)(_Float16)v[1]);
  }
}
