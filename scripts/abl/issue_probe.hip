// Micro-benchmark (round 5): what one SIMD of gfx950 does with a stream of v_mfma_f32_32x32x16_f16 and vector
// instructions -- the question behind the attention core (csrc/attention.hip): how much vector work hides under the
// matrix pipe, inside ONE wave (hand-placed stream) and ACROSS the 1-3 waves of a SIMD (hardware arbitration), for
// evenly interleaved and for phase-structured streams, with and without s_setprio around the matrix phase.
// Every loop body is ONE asm statement (the compiler cannot reorder it).  Cycles from s_memtime per wave (median over
// waves), wall time from HIP events (effective clock = cycles / wall).
// Build: hipcc -O3 --offload-arch=gfx950 issue_probe.hip -o issue_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define M  "v_mfma_f32_32x32x16_f16 %[acc], %[a], %[b], %[acc]\n"
#define M2 "v_mfma_f32_32x32x16_f16 %[acc2], %[a], %[b], %[acc2]\n"
#define F0 "v_fma_f32 %[v0], %[v0], %[c1], %[c2]\n"
#define F1 "v_fma_f32 %[v1], %[v1], %[c1], %[c2]\n"
#define F2 "v_fma_f32 %[v2], %[v2], %[c1], %[c2]\n"
#define F3 "v_fma_f32 %[v3], %[v3], %[c1], %[c2]\n"
#define F4 "v_fma_f32 %[v4], %[v4], %[c1], %[c2]\n"
#define F5 "v_fma_f32 %[v5], %[v5], %[c1], %[c2]\n"
#define F6 "v_fma_f32 %[v6], %[v6], %[c1], %[c2]\n"
#define F7 "v_fma_f32 %[v7], %[v7], %[c1], %[c2]\n"
#define E0 "v_exp_f32 %[e0], %[c1]\n"
#define E1 "v_exp_f32 %[e1], %[c2]\n"
#define E2 "v_exp_f32 %[e2], %[c1]\n"
#define E3 "v_exp_f32 %[e3], %[c2]\n"
#define D0 "v_dot2c_f32_f16 %[v0], %[c1], %[c2]\n"
#define D1 "v_dot2c_f32_f16 %[v1], %[c2], %[c1]\n"
#define D2 "v_dot2c_f32_f16 %[v2], %[c1], %[c2]\n"
#define D3 "v_dot2c_f32_f16 %[v3], %[c2], %[c1]\n"
#define C0 "v_cvt_pkrtz_f16_f32 %[e0], %[c1], %[c2]\n"
#define C1 "v_cvt_pkrtz_f16_f32 %[e1], %[c2], %[c1]\n"
#define C2 "v_cvt_pkrtz_f16_f32 %[e2], %[c1], %[c2]\n"
#define C3 "v_cvt_pkrtz_f16_f32 %[e3], %[c2], %[c1]\n"
#define X0 "v_fma_mixlo_f16 %[e0], %[c1], -1.0, %[c2] op_sel_hi:[1,0,0]\n"
#define X1 "v_fma_mixhi_f16 %[e1], %[c1], -1.0, %[c2] op_sel_hi:[1,0,0]\n"
#define X2 "v_fma_mixlo_f16 %[e2], %[c2], -1.0, %[c1] op_sel_hi:[1,0,0]\n"
#define X3 "v_fma_mixhi_f16 %[e3], %[c2], -1.0, %[c1] op_sel_hi:[1,0,0]\n"
#define A0 "v_add_f32 %[v0], %[v0], %[c1]\n"
#define A1 "v_add_f32 %[v1], %[v1], %[c2]\n"
#define A2 "v_add_f32 %[v2], %[v2], %[c1]\n"
#define A3 "v_add_f32 %[v3], %[v3], %[c2]\n"
#define P0 "v_pk_add_f16 %[e0], %[c1], %[c2]\n"
#define P1 "v_pk_add_f16 %[e1], %[c2], %[c1]\n"
#define P2 "v_pk_add_f16 %[e2], %[c1], %[c2]\n"
#define P3 "v_pk_add_f16 %[e3], %[c2], %[c1]\n"
#define X30 "v_max3_f32 %[e0], %[c1], %[c2], %[v0]\n"
#define X31 "v_max3_f32 %[e1], %[c2], %[c1], %[v1]\n"
#define X32 "v_max3_f32 %[e2], %[c1], %[c2], %[v2]\n"
#define X33 "v_max3_f32 %[e3], %[c2], %[c1], %[v3]\n"
#define Q0 "v_pk_add_f32 %[w0], %[w0], %[w1]\n"
#define Q1 "v_pk_add_f32 %[w1], %[w1], %[w0]\n"
#define G0 "v_fma_mix_f32 %[v0], %[c1], 1.0, %[v0] op_sel_hi:[1,0,0]\n"
#define G1 "v_fma_mix_f32 %[v1], %[c1], 1.0, %[v1] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define G2 "v_fma_mix_f32 %[v2], %[c2], 1.0, %[v2] op_sel_hi:[1,0,0]\n"
#define G3 "v_fma_mix_f32 %[v3], %[c2], 1.0, %[v3] op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define FX2 F0 F1
#define FX4 F0 F1 F2 F3
#define FX5 F0 F1 F2 F3 F4
#define FX6 F0 F1 F2 F3 F4 F5
#define FX8 F0 F1 F2 F3 F4 F5 F6 F7
#define R2(x) x x
#define R3(x) x x x
#define R4(x) x x x x
#define R6(x) x x x x x x
#define R8(x) x x x x x x x x
#define R12(x) R4(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)
#define PRIO1 "s_setprio 3\n"
#define PRIO0 "s_setprio 0\n"

// variant table: name, MFMAs per iteration, body
#define VARIANTS(X)                                                                                            \
  X(0, "mfma only (one chain)", 8, R8(M))                                                                      \
  X(1, "mfma + 2 fma", 8, R8(M FX2))                                                                           \
  X(2, "mfma + 4 fma", 8, R8(M FX4))                                                                           \
  X(3, "mfma + 5 fma", 8, R8(M FX5))                                                                           \
  X(4, "mfma + 6 fma", 8, R8(M FX6))                                                                           \
  X(5, "mfma + 8 fma", 8, R8(M FX8))                                                                           \
  X(6, "mfma + 1 exp + 2 fma", 8, R8(M E0 FX2))                                                                \
  X(7, "mfma + 1 exp + 4 fma", 8, R8(M E0 FX4))                                                                \
  X(8, "mfma + 2 exp + 2 fma", 8, R8(M E0 F0 E1 F1))                                                           \
  X(9, "mfma + 2 exp + 4 fma", 8, R8(M E0 F0 F1 E1 F2 F3))                                                     \
  X(10, "mfma + 3 exp", 8, R8(M E0 E1 E2))                                                                     \
  X(11, "fma only (8)", 0, R8(FX8))                                                                            \
  X(12, "exp only (8)", 0, R2(E0 E1 E2 E3))                                                                    \
  X(13, "phased: 12 mfma | 72 fma", 12, R12(M) R8(FX8) F0)                                                     \
  X(14, "phased + setprio: 12 mfma | 72 fma", 12, PRIO1 R12(M) PRIO0 R8(FX8) F0)                               \
  X(15, "interleaved: 12 x (mfma + 6 fma)", 12, R12(M FX6))                                                    \
  X(16, "attn-like phased: 12 mfma | 28 fma | 12 x (mfma + 2.67 exp + 7 fma)", 24,                             \
    R12(M) R3(FX8) FX4 R4(M2 E0 E1 E2 FX4 F4 F5 F6 M2 E0 E1 E2 FX4 F4 F5 F6 M2 E0 E1 FX4 F4 F5 F6))            \
  X(17, "attn-like phased + setprio on the score mfmas", 24,                                                   \
    PRIO1 R12(M) PRIO0 R3(FX8) FX4 R4(M2 E0 E1 E2 FX4 F4 F5 F6 M2 E0 E1 E2 FX4 F4 F5 F6 M2 E0 E1 FX4 F4 F5 F6)) \
  X(18, "attn-like even: 24 x (mfma + 1.33 exp + 4.67 fma)", 24,                                               \
    R8(M E0 FX4 F4 M2 E1 FX4 M E2 E3 FX4 F4))                                                                  \
  X(19, "attn-like lean even: 20 mfma, 32 exp, 44 fma", 20,                                                    \
    R4(M E0 E1 F0 F1 M2 E2 E3 F2 F3 M E0 E1 F0 F1 M2 E2 F2 F3 M E3 F0 F1 F2))                                  \
  X(20, "attn-like lean phased: 12 mfma | 8 x (mfma + 4 exp + 5.25 fma)", 20,                                  \
    R12(M) R8(M2 E0 E1 E2 E3 FX5) F0 F1)                                                                       \
  X(21, "attn-like lean phased + setprio", 20,                                                                 \
    PRIO1 R12(M) PRIO0 R8(M2 E0 E1 E2 E3 FX5) F0 F1)                                                           \
  X(30, "64 v_dot2c_f32_f16", 0, R16(D0 D1 D2 D3))                                                             \
  X(31, "64 v_cvt_pkrtz", 0, R16(C0 C1 C2 C3))                                                                 \
  X(32, "64 v_fma_mixlo/hi_f16", 0, R16(X0 X1 X2 X3))                                                          \
  X(33, "64 v_add_f32", 0, R16(A0 A1 A2 A3))                                                                   \
  X(34, "64 v_pk_add_f16", 0, R16(P0 P1 P2 P3))                                                                \
  X(35, "64 v_max3_f32", 0, R16(X30 X31 X32 X33))                                                              \
  X(36, "64 v_pk_add_f32", 0, R16(Q0 Q1 Q0 Q1))                                                                \
  X(37, "64 v_fma_mix_f32 (f16 src)", 0, R16(G0 G1 G2 G3))                                                     \
  X(38, "64 v_exp_f32", 0, R16(E0 E1 E2 E3))                                                                   \
  X(40, "8 x (mfma + 4 dot2c)", 8, R8(M D0 D1 D2 D3))                                                          \
  X(41, "8 x (mfma + 4 cvt_pkrtz)", 8, R8(M C0 C1 C2 C3))                                                      \
  X(42, "8 x (mfma + 4 fma_mix)", 8, R8(M X0 X1 X2 X3))                                                        \
  X(43, "8 x (mfma + 4 add)", 8, R8(M A0 A1 A2 A3))                                                            \
  X(44, "8 x (mfma + 2 pk_add_f32)", 8, R8(M Q0 Q1))                                                           \
  X(45, "8 x (mfma + 4 fma_mix_f32)", 8, R8(M G0 G1 G2 G3))                                                    \
  X(46, "8 x (mfma + 6 dot2c)", 8, R8(M D0 D1 D2 D3 D0 D1))                                                    \
  X(47, "8 x (mfma + 6 add)", 8, R8(M A0 A1 A2 A3 A0 A1))

template <int V>
__global__ __launch_bounds__(1024) void k(const float* __restrict__ src, float* __restrict__ out,
                                          unsigned long long* __restrict__ cyc, int iters) {
  const int lane = threadIdx.x & 63;
  float c1 = src[lane] * 1e-3f, c2 = src[64 + lane] * 1e-3f;
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(src[(lane + j) & 127]); b[j] = (_Float16)(src[(lane + 3 * j) & 127]); }
  f16v acc, acc2;
  for (int j = 0; j < 16; ++j) { acc[j] = 0.f; acc2[j] = 0.f; }
  float v0 = c1, v1 = c2, v2 = c1 + 1, v3 = c2 + 1, v4 = c1 + 2, v5 = c2 + 2, v6 = c1 + 3, v7 = c2 + 3;
  float e0 = 0, e1 = 0, e2 = 0, e3 = 0;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 w0 = {c1, c2}, w1 = {c2, c1};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#define X(ID, NAME, NM, BODY)                                                                                         \
    if constexpr (V == ID)                                                                                            \
      asm volatile(BODY                                                                                               \
                   : [acc] "+v"(acc), [acc2] "+v"(acc2), [v0] "+v"(v0), [v1] "+v"(v1), [v2] "+v"(v2), [v3] "+v"(v3),   \
                     [v4] "+v"(v4), [v5] "+v"(v5), [v6] "+v"(v6), [v7] "+v"(v7), [e0] "+v"(e0), [e1] "+v"(e1),        \
                     [e2] "+v"(e2), [e3] "+v"(e3), [w0] "+v"(w0), [w1] "+v"(w1)                                       \
                   : [a] "v"(a), [b] "v"(b), [c1] "v"(c1), [c2] "v"(c2));
    VARIANTS(X)
#undef X
  }
  asm volatile("s_nop 7\ns_nop 7\ns_nop 7" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + e0 + e1 + e2 + e3 + w0[0] + w0[1] + w1[0] + w1[1];
  for (int j = 0; j < 16; ++j) s += acc[j] + acc2[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int V>
void run(const char* name, int nm, int wps, float* d, float* o, unsigned long long* cyc) {
  const int iters = 4096, blocks = 256, threads = 256 * wps;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<V>), dim3(blocks), dim3(threads), 0, 0, d, o, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<V>), dim3(blocks), dim3(threads), 0, 0, d, o, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 4 * wps);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double cy = (double)h[h.size() / 2] / iters;       // cycles per iteration, one wave
  // SIMD-cycles per iteration-of-one-wave = cycles per iteration / waves per SIMD (the waves run side by side)
  printf("v%-2d %-70s w/SIMD %d  %8.1f cyc/iter/wave  %8.1f SIMD-cyc per wave-iter", V, name, wps, cy, cy / wps);
  if (nm) printf("  (%5.1f per mfma)", cy / wps / nm);
  printf("  wall %.3f ms = %7.1f ns per wave-iter per SIMD  (s_memtime clock %.2f GHz)\n", ms, ms * 1e6 / iters / wps,
         cy * iters / (ms * 1e6));
}

int main(int argc, char** argv) {
  float *d, *o; unsigned long long* cyc;
  hipMalloc(&d, 128 * 4); hipMalloc(&o, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
  float h[128]; for (int i = 0; i < 128; ++i) h[i] = 0.37f * ((i * 7919) % 41) - 7.3f;
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  const int lo_id = argc > 1 ? atoi(argv[1]) : 0;
  for (int wps : {1, 2, 3}) {
#define X(ID, NAME, NM, BODY) if (ID >= lo_id) run<ID>(NAME, NM, wps, d, o, cyc);
    VARIANTS(X)
#undef X
  }
  return 0;
}
