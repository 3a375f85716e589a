// Round 5 (design probe for a split-fp16 KPConv phase 1, DESIGN.md "Design note for the next round"):
//   A  today's form: per two ring items (16 neighbours x 64 channels) 16 v_mfma_f32_16x16x4_f32, B operand by 16
//      ds_read_b32 of f32 rows;
//   B  proposed:     12 v_mfma_f32_16x16x16_f16 (hi x hi, lo x hi, hi x lo over 4 channel blocks), B operand by 8
//      ds_read_b64_tr_b16 of rows stored as fp16 hi | lo halves.
// Prints cycles per two items per wave (8 waves per CU = two per SIMD, like the ring kernel) and checks B's result
// against a host computation (validates the transposed-read addressing).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_a(const float* __restrict__ rows, float* __restrict__ out, int iters,
                                           unsigned long long* cyc) {
  __shared__ __align__(16) float X[16 * 64];                 // 16 neighbours x 64 channels, f32
  for (int i = threadIdx.x; i < 16 * 64; i += blockDim.x) X[i] = rows[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, n = lane & 15, kg = lane >> 4;
  f32x4 acc[4] = {};
  float a = 0.001f * (lane + 1);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)                              // 4 k-steps of 4 neighbours
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {                          // 4 channel blocks
        const float b = X[(4 * ks + kg) * 64 + 16 * nb + n];   // ds_read_b32
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[nb], 0, 0, 0);
      }
    a += 1e-7f;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  float s = 0.f;
  for (int nb = 0; nb < 4; ++nb) s += acc[nb][0] + acc[nb][1] + acc[nb][2] + acc[nb][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__device__ __forceinline__ h16x4 tr_read(unsigned addr) {
  h16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ h16x4 tr_read_nw(unsigned addr) {
  h16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

// rows_h: 16 rows x [hi 64 halves | lo 64 halves]
__global__ __launch_bounds__(512) void k_b(const _Float16* __restrict__ rows_h, const float* __restrict__ infl,
                                           float* __restrict__ out, float* __restrict__ wf, int iters,
                                           unsigned long long* cyc) {
  extern __shared__ __align__(16) unsigned char lds[];
  _Float16* X = reinterpret_cast<_Float16*>(lds);              // 16 rows x 128 halves (256 B per row)
  for (int i = threadIdx.x; i < 16 * 128; i += blockDim.x) X[i] = rows_h[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, m = lane & 15, kg = lane >> 4;
  // A operand: lane (kernel point m, k-group kg) holds the influences of neighbours 4 kg .. 4 kg + 3
  h16x4 ah, al;
  for (int j = 0; j < 4; ++j) {
    const float w = infl[m * 16 + 4 * kg + j];
    ah[j] = (_Float16)w;
    al[j] = (_Float16)(w - (float)ah[j]);
  }
  // transposed read: lane 4 q + p of the 16-lane group kg supplies row 4 kg + q, columns 4 p .. 4 p + 3 of the block
  const int q = (lane & 15) >> 2, p = lane & 3;
  const unsigned base = (unsigned)(uintptr_t)X + (unsigned)((4 * kg + q) * 256 + p * 8);
  f32x4 acc[4] = {};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    h16x4 bh[4], bl[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      bh[nb] = tr_read_nw(base + nb * 32);
      bl[nb] = tr_read_nw(base + 128 + nb * 32);
    }
    // (the wait names the registers it makes valid: the compiler may not move their first use above it)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(bh[0]), "+v"(bh[1]), "+v"(bh[2]), "+v"(bh[3]), "+v"(bl[0]), "+v"(bl[1]), "+v"(bl[2]), "+v"(bl[3])
                 :: "memory");
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh[nb], acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl[nb], acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh[nb], acc[nb], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  // C layout: lane (column n = lane & 15, rows 4 (lane >> 4) + r)
  if (threadIdx.x < 64 && blockIdx.x == 0)
    for (int nb = 0; nb < 4; ++nb)
      for (int r = 0; r < 4; ++r) wf[(4 * kg + r) * 64 + 16 * nb + m] = acc[nb][r];
  float s = 0.f;
  for (int nb = 0; nb < 4; ++nb) s += acc[nb][0] + acc[nb][1] + acc[nb][2] + acc[nb][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
// C: K = 32 neighbours (four ring items) per v_mfma_f32_16x16x32_f16: lane (n, g) needs rows 8 g .. 8 g + 7 of its column:
// two transposed reads per plane and channel block.  rows_h: 32 rows x [hi 64 | lo 64].
__global__ __launch_bounds__(512) void k_c(const _Float16* __restrict__ rows_h, const float* __restrict__ infl,
                                           float* __restrict__ out, float* __restrict__ wf, int iters,
                                           unsigned long long* cyc) {
  extern __shared__ __align__(16) unsigned char lds[];
  _Float16* X = reinterpret_cast<_Float16*>(lds);              // 32 rows x 128 halves
  for (int i = threadIdx.x; i < 32 * 128; i += blockDim.x) X[i] = rows_h[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, m = lane & 15, kg = lane >> 4;
  h16x8 ah, al;
  for (int j = 0; j < 8; ++j) {
    const float w = infl[m * 32 + 8 * kg + j];
    ah[j] = (_Float16)w;
    al[j] = (_Float16)(w - (float)ah[j]);
  }
  const int q = (lane & 15) >> 2, p = lane & 3;
  const unsigned base = (unsigned)(uintptr_t)X + (unsigned)((8 * kg + q) * 256 + p * 8);
  f32x4 acc[4] = {};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    h16x4 b0h[4], b1h[4], b0l[4], b1l[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      b0h[nb] = tr_read_nw(base + nb * 32);
      b1h[nb] = tr_read_nw(base + 4 * 256 + nb * 32);
      b0l[nb] = tr_read_nw(base + 128 + nb * 32);
      b1l[nb] = tr_read_nw(base + 4 * 256 + 128 + nb * 32);
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(b0h[0]), "+v"(b0h[1]), "+v"(b0h[2]), "+v"(b0h[3]), "+v"(b1h[0]), "+v"(b1h[1]), "+v"(b1h[2]), "+v"(b1h[3]),
                   "+v"(b0l[0]), "+v"(b0l[1]), "+v"(b0l[2]), "+v"(b0l[3]), "+v"(b1l[0]), "+v"(b1l[1]), "+v"(b1l[2]), "+v"(b1l[3])
                 :: "memory");
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const h16x8 bh = {b0h[nb][0], b0h[nb][1], b0h[nb][2], b0h[nb][3], b1h[nb][0], b1h[nb][1], b1h[nb][2], b1h[nb][3]};
      const h16x8 bl = {b0l[nb][0], b0l[nb][1], b0l[nb][2], b0l[nb][3], b1l[nb][0], b1l[nb][1], b1l[nb][2], b1l[nb][3]};
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[nb], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  if (threadIdx.x < 64 && blockIdx.x == 0)
    for (int nb = 0; nb < 4; ++nb)
      for (int r = 0; r < 4; ++r) wf[(4 * kg + r) * 64 + 16 * nb + m] = acc[nb][r];
  float s = 0.f;
  for (int nb = 0; nb < 4; ++nb) s += acc[nb][0] + acc[nb][1] + acc[nb][2] + acc[nb][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// D: as C, software pipelined -- the sixteen transposed reads of the NEXT K-step are in flight under the twelve MFMAs
// of the current one (what the ring kernel would do).
__global__ __launch_bounds__(512) void k_d(const _Float16* __restrict__ rows_h, const float* __restrict__ infl,
                                           float* __restrict__ out, int iters, unsigned long long* cyc) {
  extern __shared__ __align__(16) unsigned char lds[];
  _Float16* X = reinterpret_cast<_Float16*>(lds);
  for (int i = threadIdx.x; i < 32 * 128; i += blockDim.x) X[i] = rows_h[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, m = lane & 15, kg = lane >> 4;
  h16x8 ah, al;
  for (int j = 0; j < 8; ++j) {
    const float w = infl[m * 32 + 8 * kg + j];
    ah[j] = (_Float16)w;
    al[j] = (_Float16)(w - (float)ah[j]);
  }
  const int q = (lane & 15) >> 2, p = lane & 3;
  const unsigned base = (unsigned)(uintptr_t)X + (unsigned)((8 * kg + q) * 256 + p * 8);
  f32x4 acc[4] = {};
  h16x4 c0h[4], c1h[4], c0l[4], c1l[4], n0h[4], n1h[4], n0l[4], n1l[4];
  auto issue = [&](h16x4 (&a0)[4], h16x4 (&a1)[4], h16x4 (&b0)[4], h16x4 (&b1)[4]) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      a0[nb] = tr_read_nw(base + nb * 32);
      a1[nb] = tr_read_nw(base + 4 * 256 + nb * 32);
      b0[nb] = tr_read_nw(base + 128 + nb * 32);
      b1[nb] = tr_read_nw(base + 4 * 256 + 128 + nb * 32);
    }
  };
  issue(c0h, c1h, c0l, c1l);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(c0h[0]), "+v"(c0h[1]), "+v"(c0h[2]), "+v"(c0h[3]), "+v"(c1h[0]), "+v"(c1h[1]), "+v"(c1h[2]), "+v"(c1h[3]),
                   "+v"(c0l[0]), "+v"(c0l[1]), "+v"(c0l[2]), "+v"(c0l[3]), "+v"(c1l[0]), "+v"(c1l[1]), "+v"(c1l[2]), "+v"(c1l[3])
                 :: "memory");
    issue(n0h, n1h, n0l, n1l);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const h16x8 bh = {c0h[nb][0], c0h[nb][1], c0h[nb][2], c0h[nb][3], c1h[nb][0], c1h[nb][1], c1h[nb][2], c1h[nb][3]};
      const h16x8 bl = {c0l[nb][0], c0l[nb][1], c0l[nb][2], c0l[nb][3], c1l[nb][0], c1l[nb][1], c1l[nb][2], c1l[nb][3]};
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[nb], 0, 0, 0);
    }
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      c0h[nb] = n0h[nb]; c1h[nb] = n1h[nb]; c0l[nb] = n0l[nb]; c1l[nb] = n1l[nb];
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  float s = 0.f;
  for (int nb = 0; nb < 4; ++nb) s += acc[nb][0] + acc[nb][1] + acc[nb][2] + acc[nb][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// E: as D on the conflict-free image (b) of cdna_hip_programming.md T10: the 16-byte chunk ch of row r lives at chunk
// ch ^ (((r & 3) << 2) | ((r >> 2) & 3)) of its 256-byte row (with plain rows every row starts on bank 0 and the four
// rows a 16-lane group reads collide: C and D above are 8-way conflicted).
__device__ __forceinline__ unsigned sw_off(unsigned row, unsigned ch) { return 256u * row + 16u * (ch ^ (((row & 3u) << 2) | ((row >> 2) & 3u))); }
__global__ __launch_bounds__(512) void k_e(const _Float16* __restrict__ rows_h, const float* __restrict__ infl,
                                           float* __restrict__ out, float* __restrict__ wf, int iters,
                                           unsigned long long* cyc) {
  extern __shared__ __align__(16) unsigned char lds[];
  for (int i = threadIdx.x; i < 32 * 16; i += blockDim.x) {      // 16-byte chunks
    const unsigned row = i >> 4, ch = i & 15;
    *reinterpret_cast<uint4*>(lds + sw_off(row, ch)) = reinterpret_cast<const uint4*>(rows_h)[i];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, m = lane & 15, kg = lane >> 4;
  h16x8 ah, al;
  for (int j = 0; j < 8; ++j) {
    const float w = infl[m * 32 + 8 * kg + j];
    ah[j] = (_Float16)w;
    al[j] = (_Float16)(w - (float)ah[j]);
  }
  const int q = (lane & 15) >> 2, p = lane & 3;
  const unsigned lb = (unsigned)(uintptr_t)lds;
  unsigned a0h[4], a1h[4], a0l[4], a1l[4];                     // per-lane addresses (loop invariant)
  for (int nb = 0; nb < 4; ++nb) {
    a0h[nb] = lb + sw_off(8 * kg + q, 2 * nb + (p >> 1)) + 8 * (p & 1);
    a1h[nb] = lb + sw_off(8 * kg + 4 + q, 2 * nb + (p >> 1)) + 8 * (p & 1);
    a0l[nb] = lb + sw_off(8 * kg + q, 8 + 2 * nb + (p >> 1)) + 8 * (p & 1);
    a1l[nb] = lb + sw_off(8 * kg + 4 + q, 8 + 2 * nb + (p >> 1)) + 8 * (p & 1);
  }
  f32x4 acc[4] = {};
  h16x4 c0h[4], c1h[4], c0l[4], c1l[4], n0h[4], n1h[4], n0l[4], n1l[4];
  auto issue = [&](h16x4 (&x0)[4], h16x4 (&x1)[4], h16x4 (&y0)[4], h16x4 (&y1)[4]) {
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      x0[nb] = tr_read_nw(a0h[nb]);
      x1[nb] = tr_read_nw(a1h[nb]);
      y0[nb] = tr_read_nw(a0l[nb]);
      y1[nb] = tr_read_nw(a1l[nb]);
    }
  };
  issue(c0h, c1h, c0l, c1l);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(c0h[0]), "+v"(c0h[1]), "+v"(c0h[2]), "+v"(c0h[3]), "+v"(c1h[0]), "+v"(c1h[1]), "+v"(c1h[2]), "+v"(c1h[3]),
                   "+v"(c0l[0]), "+v"(c0l[1]), "+v"(c0l[2]), "+v"(c0l[3]), "+v"(c1l[0]), "+v"(c1l[1]), "+v"(c1l[2]), "+v"(c1l[3])
                 :: "memory");
    issue(n0h, n1h, n0l, n1l);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const h16x8 bh = {c0h[nb][0], c0h[nb][1], c0h[nb][2], c0h[nb][3], c1h[nb][0], c1h[nb][1], c1h[nb][2], c1h[nb][3]};
      const h16x8 bl = {c0l[nb][0], c0l[nb][1], c0l[nb][2], c0l[nb][3], c1l[nb][0], c1l[nb][1], c1l[nb][2], c1l[nb][3]};
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[nb], 0, 0, 0);
      acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[nb], 0, 0, 0);
    }
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      c0h[nb] = n0h[nb]; c1h[nb] = n1h[nb]; c0l[nb] = n0l[nb]; c1l[nb] = n1l[nb];
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  if (threadIdx.x < 64 && blockIdx.x == 0)
    for (int nb = 0; nb < 4; ++nb)
      for (int r = 0; r < 4; ++r) wf[(4 * kg + r) * 64 + 16 * nb + m] = acc[nb][r];
  float s = 0.f;
  for (int nb = 0; nb < 4; ++nb) s += acc[nb][0] + acc[nb][1] + acc[nb][2] + acc[nb][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  const int iters = 4000;
  std::vector<float> X(16 * 64), W(16 * 16);
  srand(1);
  for (auto& v : X) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
  for (auto& v : W) v = rand() / (float)RAND_MAX;
  std::vector<_Float16> Xh(16 * 128);
  for (int r = 0; r < 16; ++r)
    for (int c = 0; c < 64; ++c) {
      const _Float16 h = (_Float16)X[r * 64 + c];
      Xh[r * 128 + c] = h;
      Xh[r * 128 + 64 + c] = (_Float16)(X[r * 64 + c] - (float)h);
    }
  float *dX, *dW, *dout, *dwf; _Float16* dXh; unsigned long long* dc;
  (void)hipMalloc(&dX, X.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&dXh, Xh.size() * 2);
  hipMalloc(&dout, 256 * 512 * 4); hipMalloc(&dwf, 16 * 64 * 4); hipMalloc(&dc, 8);
  hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dXh, Xh.data(), Xh.size() * 2, hipMemcpyHostToDevice);
  unsigned long long c = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_a, dim3(256), dim3(512), 0, 0, dX, dout, iters, dc);
    hipDeviceSynchronize(); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    if (rep) printf("A  f32 16x16x4 x16 + 16 ds_read_b32      : %7.1f cycles per two items per wave (2 waves/SIMD)\n", (double)c / iters);
    hipLaunchKernelGGL(k_b, dim3(256), dim3(512), 4096, 0, dXh, dW, dout, dwf, iters, dc);
    hipDeviceSynchronize(); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    if (rep) printf("B  f16 16x16x16 x12 + 8 ds_read_b64_tr_b16: %7.1f cycles per two items per wave (2 waves/SIMD)\n", (double)c / iters);
  }
  std::vector<float> wf(16 * 64);
  hipLaunchKernelGGL(k_b, dim3(1), dim3(512), 4096, 0, dXh, dW, dout, dwf, 1, dc);      // one pass: the result itself
  hipDeviceSynchronize();
  hipMemcpy(wf.data(), dwf, wf.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0, scale = 0;
  for (int kp = 0; kp < 16; ++kp)
    for (int ch = 0; ch < 64; ++ch) {
      double ref = 0;
      for (int nb = 0; nb < 16; ++nb) ref += (double)W[kp * 16 + nb] * X[nb * 64 + ch];
      worst = fmax(worst, fabs(ref - wf[kp * 64 + ch]));
      scale = fmax(scale, fabs(ref));
    }
  printf("B  result vs float64: max error %.3e of scale %.3e (%s)\n", worst, scale, worst < 1e-5 * scale ? "transposed-read addressing OK" : "MISMATCH");
  // ---- C: 32 neighbours per MFMA
  std::vector<float> X2(32 * 64), W2(16 * 32);
  for (auto& v : X2) v = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
  for (auto& v : W2) v = rand() / (float)RAND_MAX;
  std::vector<_Float16> X2h(32 * 128);
  for (int r = 0; r < 32; ++r)
    for (int c2 = 0; c2 < 64; ++c2) {
      const _Float16 h = (_Float16)X2[r * 64 + c2];
      X2h[r * 128 + c2] = h;
      X2h[r * 128 + 64 + c2] = (_Float16)(X2[r * 64 + c2] - (float)h);
    }
  float* dW2; _Float16* dX2h;
  hipMalloc(&dW2, W2.size() * 4); hipMalloc(&dX2h, X2h.size() * 2);
  hipMemcpy(dW2, W2.data(), W2.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dX2h, X2h.data(), X2h.size() * 2, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_c, dim3(256), dim3(512), 8192, 0, dX2h, dW2, dout, dwf, iters, dc);
    hipDeviceSynchronize(); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    if (rep) printf("C  f16 16x16x32 x12 + 16 ds_read_b64_tr_b16: %7.1f cycles per FOUR items per wave (2 waves/SIMD) = %.1f per two\n", (double)c / iters, (double)c / iters / 2);
  }
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_d, dim3(256), dim3(512), 8192, 0, dX2h, dW2, dout, iters, dc);
    hipDeviceSynchronize(); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    if (rep) printf("D  as C, reads of the next K-step under the MFMAs : %7.1f cycles per FOUR items per wave = %.1f per two\n", (double)c / iters, (double)c / iters / 2);
  }
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_e, dim3(256), dim3(512), 8192, 0, dX2h, dW2, dout, dwf, iters, dc);
    hipDeviceSynchronize(); hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    if (rep) printf("E  as D on the swizzled (conflict-free) image     : %7.1f cycles per FOUR items per wave = %.1f per two\n", (double)c / iters, (double)c / iters / 2);
  }
  {
    hipLaunchKernelGGL(k_e, dim3(1), dim3(512), 8192, 0, dX2h, dW2, dout, dwf, 1, dc);
    hipDeviceSynchronize();
    hipMemcpy(wf.data(), dwf, wf.size() * 4, hipMemcpyDeviceToHost);
    double w2 = 0, s2 = 0;
    for (int kp = 0; kp < 16; ++kp)
      for (int ch = 0; ch < 64; ++ch) {
        double ref = 0;
        for (int nb = 0; nb < 32; ++nb) ref += (double)W2[kp * 32 + nb] * X2[nb * 64 + ch];
        // k_e ran ONE iteration = the prologue's reads only once more: acc holds exactly one K-step
        w2 = fmax(w2, fabs(ref - wf[kp * 64 + ch]));
        s2 = fmax(s2, fabs(ref));
      }
    printf("E  result vs float64: max error %.3e of scale %.3e (%s)\n", w2, s2, w2 < 1e-5 * s2 ? "swizzled addressing OK" : "MISMATCH");
  }
  hipLaunchKernelGGL(k_c, dim3(1), dim3(512), 8192, 0, dX2h, dW2, dout, dwf, 1, dc);
  hipDeviceSynchronize();
  hipMemcpy(wf.data(), dwf, wf.size() * 4, hipMemcpyDeviceToHost);
  worst = 0; scale = 0;
  for (int kp = 0; kp < 16; ++kp)
    for (int ch = 0; ch < 64; ++ch) {
      double ref = 0;
      for (int nb = 0; nb < 32; ++nb) ref += (double)W2[kp * 32 + nb] * X2[nb * 64 + ch];
      worst = fmax(worst, fabs(ref - wf[kp * 64 + ch]));
      scale = fmax(scale, fabs(ref));
    }
  printf("C  result vs float64: max error %.3e of scale %.3e (%s)\n", worst, scale, worst < 1e-5 * scale ? "addressing OK" : "MISMATCH");
  hipError_t e = hipGetLastError();
  return e != hipSuccess;
}
