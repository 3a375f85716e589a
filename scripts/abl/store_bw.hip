// Micro-benchmark: HBM write bandwidth of three store patterns over a [M x N] f32 matrix.
//   a: MFMA C-layout dword stores (lane = column, register = row): 2 rows x 128 B per instruction
//   b: dwordx4 stores, 16 lanes per row (256 B contiguous per row, 4 rows per instruction)
//   c: fully linear dwordx4
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void ka(float* out, int M, int N) {
  // block = 128 x 64 tile, 4 waves x (32 x 64)
  const int gx = N / 64;
  const int m0 = (blockIdx.x / gx) * 128 + (threadIdx.x >> 6) * 32, n0 = (blockIdx.x % gx) * 64;
  const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
  for (int j = 0; j < 2; ++j)
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < M) out[(size_t)row * N + n0 + 32 * j + l31] = (float)r;
    }
}
__global__ __launch_bounds__(256) void kb(float* out, int M, int N) {
  const int gx = N / 64;
  const int m0 = (blockIdx.x / gx) * 128 + (threadIdx.x >> 6) * 32, n0 = (blockIdx.x % gx) * 64;
  const int lane = threadIdx.x & 63, c = lane & 15, rr = lane >> 4;
  for (int i = 0; i < 8; ++i) {
    const int row = m0 + 4 * i + rr;
    if (row < M) *reinterpret_cast<float4*>(out + (size_t)row * N + n0 + 4 * c) = make_float4(1, 2, 3, 4);
  }
}
__global__ __launch_bounds__(256) void kc(float* out, size_t n4) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n4; i += stride) reinterpret_cast<float4*>(out)[i] = make_float4(1, 2, 3, 4);
}
int main() {
  const int M = 61745, N = 768;
  float* o; hipMalloc(&o, (size_t)(M + 256) * N * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = ((M + 127) / 128) * (N / 64);
  for (int v = 0; v < 3; ++v) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      if (v == 0) hipLaunchKernelGGL(ka, dim3(blocks), dim3(256), 0, 0, o, M, N);
      if (v == 1) hipLaunchKernelGGL(kb, dim3(blocks), dim3(256), 0, 0, o, M, N);
      if (v == 2) hipLaunchKernelGGL(kc, dim3(256 * 8), dim3(256), 0, 0, o, (size_t)M * N / 4);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("pattern %c: %.1f us  %.2f TB/s\n", 'a' + v, best * 1e3, (double)M * N * 4 / best * 1e-9);
  }
  return 0;
}
