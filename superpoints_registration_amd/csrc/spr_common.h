// Shared host/device helpers for libspr_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdint>

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
// hi_u / lo_u hold (a, b) as packed halves.
typedef _Float16 spr_h16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pk(float a, float b, unsigned int& hi_u, unsigned int& lo_u) {
  const spr_h16x2 hi = {(_Float16)a, (_Float16)b};
  hi_u = __builtin_bit_cast(unsigned int, hi);
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(lo_u)
      : "v"(hi_u), "v"(a), "v"(b));
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
