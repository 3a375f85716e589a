// Shared host/device helpers for libspr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstring>

#include "../../include/spr.h"

namespace spr {

constexpr int kWave = 64;

void set_error(const char* fmt, ...);

#define SPR_HIP_CHECK(expr)                                                   \
  do {                                                                        \
    hipError_t _e = (expr);                                                   \
    if (_e != hipSuccess) {                                                   \
      ::spr::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                       __FILE__, __LINE__);                                   \
      return 1;                                                               \
    }                                                                         \
  } while (0)

#define SPR_LAUNCH_CHECK()                                                  \
  do {                                                                      \
    hipError_t _e = hipGetLastError();                                      \
    if (_e != hipSuccess) {                                                 \
      ::spr::set_error("kernel launch failed: %s (%s:%d)",                  \
                       hipGetErrorString(_e), __FILE__, __LINE__);          \
      return 1;                                                             \
    }                                                                       \
  } while (0)

#define SPR_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      ::spr::set_error(__VA_ARGS__);  \
      return 2;                       \
    }                                 \
  } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Bump allocator over a caller supplied workspace.
struct Workspace {
  char* base;
  size_t size;
  size_t off;
  Workspace(void* p, size_t n) : base((char*)p), size(n), off(0) {}
  template <typename T>
  T* take(size_t count) {
    size_t bytes = align_up(count * sizeof(T), 256);
    if (off + bytes > size) return nullptr;
    T* r = (T*)(base + off);
    off += bytes;
    return r;
  }
};

// ---- optional per-launch HIP-event timing (bench.py roofline legs) ----------
// code: fused KPConv = cin * 100000 + cout; attention core (k_attn_h3 / k_attn) = -1.
// Records are appended under a mutex (launches may come from several host
// threads, one per stream); see spr_prof_enable / spr_prof_read in include/spr.h.
bool prof_enabled();
void prof_push(hipEvent_t beg, hipEvent_t end, int code, int n);
struct ProfScope {
  hipStream_t stream;
  bool on;
  hipEvent_t beg, end;
  int code, n;
  ProfScope(hipStream_t s, int code_, int n_) : stream(s), on(prof_enabled()), code(code_), n(n_) {
    if (!on) return;
    if (hipEventCreate(&beg) != hipSuccess || hipEventCreate(&end) != hipSuccess) {
      on = false;
      return;
    }
    (void)hipEventRecord(beg, stream);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(end, stream);
    prof_push(beg, end, code, n);
  }
};

// Per-device one-time kernel attribute (dynamic LDS above 64 KB) and CU count; keyed by the
// current device, mutex protected (launches come from several host threads).  core.hip.
int ensure_dyn_lds(const void* kernel, int bytes);
int device_cu_count();

// max |x| over a [rows, cols] view with row stride `stride` (floats) -> parts[kAmaxParts]
// (device); see block_absmax.  Defined in core.hip.
int launch_absmax(const float* x, long rows, int cols, long stride, float* parts, hipStream_t stream);
// two tensors in one launch (an activation and the weights it meets)
int launch_absmax2(const float* x0, long rows0, int cols0, long stride0, float* parts0, const float* x1,
                   long rows1, int cols1, long stride1, float* parts1, hipStream_t stream);

// ---- device helpers --------------------------------------------------------
#ifdef __HIPCC__
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Largest s in [0, nseg) with cu[s] <= i  (cu[0] = 0, cu non-decreasing).
__device__ __forceinline__ int find_segment(const int* __restrict__ cu, int nseg,
                                            int i) {
  int lo = 0, hi = nseg;  // invariant: cu[lo] <= i < cu[hi]
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (cu[mid] <= i)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

// XCD-aware remap of a linear workgroup id (MI355X: 8 XCDs, each with its own
// L2; the dispatcher deals workgroups round-robin over them, so ids b and b+8
// share an L2).  Returns a bijective logical id such that CONSECUTIVE logical
// ids run on the same XCD -- tiles that share an operand panel then hit one L2
// instead of eight.  Speed only: correctness never depends on placement.
__device__ __forceinline__ int xcd_swizzle(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = b & 7, idx = b >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// Split-fp16 operand: hi = fp16(x) (RNE, packed convert), lo = fp16(x - hi) with
// the subtraction and the narrowing done by ONE mixed-precision FMA per
// element (v_fma_mixlo/hi_f16 widen the fp16 operand inside the ALU).
// hi_u / lo_u hold (a, b) as packed halves.  UNSCALED form: only for operands
// whose magnitude is known to sit in fp16's comfortable range (softmax
// probabilities); everything else goes through split_pk_s.
typedef _Float16 spr_h16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pk(float a, float b, unsigned int& hi_u, unsigned int& lo_u) {
  const spr_h16x2 hi = {(_Float16)a, (_Float16)b};
  hi_u = __builtin_bit_cast(unsigned int, hi);
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(lo_u)
      : "v"(hi_u), "v"(a), "v"(b));
}

// Range-safe split: the operand is first multiplied by a power of two s (exact)
// chosen per TENSOR so that max |x| s lies in [2^14, 2^15) (pow2_scale_for):
//   hi = fp16(x s),  lo = fp16(x s - hi)      (one v_fma_mix*_f16 each)
// Every element with |x| >= 2^-18 max|x| then has a normal lo (22 significand
// bits together with hi); smaller elements keep an ABSOLUTE error <= 2^-25 in
// scaled units = 2^-39 max|x| -- far below one fp32 ulp of the tensor's large
// entries.  No overflow is possible (max |x s| < 2^15 < 65504).  The product of
// two scaled operands is unscaled in the epilogue by the exact factor
// 1 / (s_a s_b).
__device__ __forceinline__ void split_pk_s(float a, float b, float s, unsigned int& hi_u,
                                           unsigned int& lo_u) {
  asm("v_fma_mixlo_f16 %0, %1, %3, 0 op_sel_hi:[0,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %2, %3, 0 op_sel_hi:[0,0,0]"
      : "=&v"(hi_u)
      : "v"(a), "v"(b), "v"(s));
  asm("v_fma_mixlo_f16 %0, %2, %4, -%1 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %3, %4, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(lo_u)
      : "v"(hi_u), "v"(a), "v"(b), "v"(s));
}

// The two halves of split_pk_s as separate calls (same instructions, same bits): for callers that place the work in
// the issue shadow of matrix instructions a few cycles at a time (xenc.hip).
__device__ __forceinline__ void split_pk_s_hi(float a, float b, float s, unsigned int& hi_u) {
  asm("v_fma_mixlo_f16 %0, %1, %3, 0 op_sel_hi:[0,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %2, %3, 0 op_sel_hi:[0,0,0]"
      : "=&v"(hi_u)
      : "v"(a), "v"(b), "v"(s));
}
__device__ __forceinline__ void split_pk_s_lo(float a, float b, float s, unsigned int hi_u, unsigned int& lo_u) {
  asm("v_fma_mixlo_f16 %0, %2, %4, -%1 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %3, %4, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(lo_u)
      : "v"(hi_u), "v"(a), "v"(b), "v"(s));
}

// Exponent k of the power-of-two scale 2^k that brings `amax` into [2^14, 2^15);
// clamped to [-60, 60] (so that 2^-(ka+kb) stays a normal float); 0 for a zero,
// negative or non-finite bound.
__host__ __device__ __forceinline__ int pow2_exp_for(float amax) {
  if (!(amax > 0.f) || !(amax < 3.0e38f)) return 0;
  unsigned int u;
#ifdef __HIP_DEVICE_COMPILE__
  u = __float_as_uint(amax);
#else
  memcpy(&u, &amax, 4);
#endif
  const int e = (int)((u >> 23) & 0xff) - 127;   // floor(log2 amax) (denormals: -127)
  int k = 14 - e;
  if (k > 60) k = 60;
  if (k < -60) k = -60;
  return k;
}
__host__ __device__ __forceinline__ float pow2f(int k) {   // 2^k, |k| <= 126
  const unsigned int u = (unsigned int)(127 + k) << 23;
  float f;
#ifdef __HIP_DEVICE_COMPILE__
  f = __uint_as_float(u);
#else
  memcpy(&f, &u, 4);
#endif
  return f;
}

// ---- per-tensor max |x| as kAmaxParts per-block partials ---------------------
// launch_absmax writes exactly kAmaxParts floats (unused blocks write 0); the
// consumer kernel reduces them in its prologue with block_absmax -- no atomics,
// no memset, no host round trip.
constexpr int kAmaxParts = 512;
// slots of a range published by a producer kernel (atomic max; LayerNorm, InstanceNorm, max-pool)
constexpr int kRangeSlots = 64;
// Reduces parts[0..kAmaxParts) over the workgroup; `sh` = 17 floats of LDS.
// Contains two __syncthreads().  Result returned to every thread.
// n: number of partials (kAmaxParts for a measured range; producers that publish the range of
// their own output -- LayerNorm, the ReLU GEMM, the attention core -- write one partial per
// workgroup, or a single bound).
__device__ __forceinline__ float block_absmax(const float* __restrict__ parts, float* sh, int n = kAmaxParts) {
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, parts[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = sh[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t = fmaxf(t, sh[w]);
    sh[16] = t;
  }
  __syncthreads();
  return sh[16];
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
#endif

}  // namespace spr
