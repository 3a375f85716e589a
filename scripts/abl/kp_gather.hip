// Pure-gather micro-benchmark for the KPConv neighbour gather (task 1a).
//
// Question: what does one MI355X CU sustain when it gathers the KPConv's neighbour rows
// (128 / 256 / 512-byte feature rows selected by the REAL neighbour matrices of the bench
// pyramid) into LDS, with no influence math and no MFMA beside it?  The guide's table
// (MI355X_MICROARCH.md "Indexed rows: gather into LDS") is for 1,152-byte rows: 66-73 GB/s
// per CU from L2, 33.5 from the Infinity Cache, 23-24 from HBM.
//
// Build:  hipcc -O3 --offload-arch=gfx950 -shared -fPIC kp_gather.hip -o libkp_gather.so
// Driven by scripts/kp_gather_bench.py (ctypes; device pointers from torch).
//
// Variants (mode):
//   0  LDS-DMA   global_load_lds_dwordx4, per-lane source address, D KiB in flight per wave
//   1  register  global_load_dwordx4 -> ds_write_b128, D loads in flight per wave
//   2  register  global_load_dwordx4 -> xor into a register (no LDS write)
// Work split: one wave = one query at a time (its neighbour row is read with one coalesced
// load one query ahead); a wave-instruction fetches 1 KiB = 1024 / row_bytes rows.
// Tile walk: persistent workgroups; xcd_chunk = 1 gives every XCD (blockIdx % 8) a contiguous
// eighth of the query order, so that with a spatial order the workgroups of one XCD share rows
// in its L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace {

__device__ __forceinline__ void wait_vm_dyn(int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 23: asm volatile("s_waitcnt vmcnt(23)" ::: "memory"); break;
    case 31: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

typedef float f4 __attribute__((ext_vector_type(4)));

// MODE as above; D = 1-KiB pieces in flight per wave; LPR = lanes per row (row_bytes / 16).
// No compiler-visible global load anywhere in the loop (hipcc would drain the asm loads with a
// vmcnt(0) at its first use): the neighbour rows of a wave's next batch of QB queries also arrive by
// LDS-DMA, one batch ahead, into a private double buffer, and every wait is a counted one.
constexpr int QB = 8;
__device__ __forceinline__ void glds16(const void* src, unsigned dst_s) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst_s) : "memory");
}
__device__ __forceinline__ void glds4(const void* src, unsigned dst_s) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst_s) : "memory");
}

template <int MODE, int D, int LPR>
__global__ __launch_bounds__(1024) void k_gather(const int* __restrict__ nbr, int nq, int stride, int kmax,
                                                 const float* __restrict__ x, int ns, int xcd_chunk,
                                                 unsigned long long* __restrict__ rows_done,
                                                 float* __restrict__ sink) {
  extern __shared__ __align__(16) unsigned char lds[];
  constexpr int RPI = 64 / LPR;                       // rows per wave-instruction
  constexpr int WAVE_LDS = D * 1024 + 2 * QB * 256;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwave = blockDim.x >> 6;
  unsigned char* ring = lds + (size_t)wave * WAVE_LDS;
  int* ibuf = reinterpret_cast<int*>(ring + D * 1024);          // [2][QB][64]
  const unsigned ring_s = (unsigned)(uintptr_t)ring, ibuf_s = (unsigned)(uintptr_t)ibuf;
  const int sub = lane / LPR, piece = lane % LPR;
  // batches of QB consecutive queries; waves of the chip take batches round robin (xcd_chunk = 0), or
  // every XCD takes a contiguous eighth of the batch list and its waves go round robin inside it
  const int nbatch = (nq + QB - 1) / QB;
  int b_begin, b_step, b_end;
  if (xcd_chunk) {
    const int xcd = blockIdx.x & 7, per = (nbatch + 7) / 8;
    const int wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    b_begin = xcd * per + wg_in_xcd * nwave + wave;
    b_step = wgs_per_xcd * nwave;
    b_end = min(nbatch, (xcd + 1) * per);
  } else {
    b_begin = blockIdx.x * nwave + wave;
    b_step = gridDim.x * nwave;
    b_end = nbatch;
  }
  const size_t last_ok = (size_t)nq * stride - 1;
  auto fetch_batch = [&](int b, int buf) {   // QB index rows -> ibuf[buf]; lanes beyond the row read the next row's head
    for (int i = 0; i < QB; ++i) {
      size_t e = (size_t)min(b * QB + i, nq - 1) * stride + lane;
      e = e < last_ok ? e : last_ok;
      glds4(nbr + e, __builtin_amdgcn_readfirstlane(ibuf_s + (buf * QB + i) * 256));
    }
  };
  int c = 0;                 // pieces issued by this wave
  unsigned long long rows = 0;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  f4 stage[MODE == 0 ? 1 : D];
  if (b_begin < b_end) fetch_batch(b_begin, 0);
  int buf = 0;
  bool need_drain = true;    // a batch's index rows have landed once D pieces were issued behind them
  for (int b = b_begin; b < b_end; b += b_step, buf ^= 1) {
    if (need_drain) wait_vm_dyn(0);
    if (b + b_step < b_end) fetch_batch(b + b_step, buf ^ 1);
    int issued = 0;
    for (int i = 0; i < QB; ++i) {
      if (b * QB + i >= nq) break;
      int idx = ibuf[(buf * QB + i) * 64 + lane];
      if (lane >= kmax) idx = ns;
      const int v = __popcll(__ballot(idx >= 0 && idx < ns));   // rows are distance sorted: valid prefix
      rows += v;
      for (int k0 = 0; k0 < v; k0 += RPI) {
        const int kk = k0 + sub;
        int id = __shfl(idx, min(kk, 63), 64);
        if (kk >= v) id = __shfl(idx, 0, 64);                   // pad with the first (valid) row
        const float* src = x + (size_t)id * (LPR * 4) + piece * 4;
        const int slot = c % D;
        if (MODE == 0) {
          wait_vm_dyn(D - 1);
          glds16(src, __builtin_amdgcn_readfirstlane(ring_s + slot * 1024));
        } else {
          // software ring of D register slots: consume the oldest, then refill it
#pragma unroll
          for (int d = 0; d < D; ++d) {
            if (d == slot) {
              if (c >= D) {
                wait_vm_dyn(D - 1);
                asm volatile("" : "+v"(stage[d]));
                if (MODE == 1) *reinterpret_cast<f4*>(ring + d * 1024 + lane * 16) = stage[d];
                else acc += stage[d];
              }
              asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(stage[d]) : "v"(src) : "memory");
            }
          }
        }
        ++c;
        ++issued;
      }
    }
    need_drain = issued < D;
  }
  wait_vm_dyn(0);
  if (MODE != 0) {
#pragma unroll
    for (int d = 0; d < D; ++d) { asm volatile("" : "+v"(stage[d])); acc += stage[d]; }
  }
  if (MODE != 2) acc += *reinterpret_cast<f4*>(ring + lane * 16);
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
  rows = __shfl(rows, 0, 64);
  if (lane == 0) atomicAdd(rows_done, rows);
}

template <int MODE, int D, int LPR>
int launch(const int* nbr, int nq, int stride, int kmax, const float* x, int ns, int xcd_chunk,
           int waves_per_wg, int wgs_per_cu, unsigned long long* rows_done, float* sink, hipStream_t st) {
  const size_t ldsb = (size_t)waves_per_wg * (D * 1024 + 2 * QB * 256);
  auto kern = k_gather<MODE, D, LPR>;
  if (ldsb > 160 * 1024) return -2;
  if (ldsb > 64 * 1024) hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(kern, dim3(256 * wgs_per_cu), dim3(64 * waves_per_wg), ldsb, st, nbr, nq, stride, kmax, x, ns,
                     xcd_chunk, rows_done, sink);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

#define DISPATCH_D(MODE, LPR)                                                                                     \
  switch (depth) {                                                                                               \
    case 2: rc = launch<MODE, 2, LPR>(nbr, nq, stride, kmax, x, ns, xcd_chunk, wpw, wpc, rows, sink, st); break;   \
    case 4: rc = launch<MODE, 4, LPR>(nbr, nq, stride, kmax, x, ns, xcd_chunk, wpw, wpc, rows, sink, st); break;   \
    case 8: rc = launch<MODE, 8, LPR>(nbr, nq, stride, kmax, x, ns, xcd_chunk, wpw, wpc, rows, sink, st); break;   \
    case 16: rc = launch<MODE, 16, LPR>(nbr, nq, stride, kmax, x, ns, xcd_chunk, wpw, wpc, rows, sink, st); break; \
    default: rc = -3;                                                                                            \
  }
#define DISPATCH_L(MODE)                                                  \
  switch (row_bytes) {                                                    \
    case 128: DISPATCH_D(MODE, 8) break;                                  \
    case 256: DISPATCH_D(MODE, 16) break;                                 \
    case 512: DISPATCH_D(MODE, 32) break;                                 \
    default: rc = -4;                                                     \
  }

// Returns the average kernel time in ms over `reps` launches (after one warm launch) in *ms_out and
// the rows gathered per launch in *rows_out.
extern "C" int kpg_run(int mode, int depth, int row_bytes, int wpw, int wpc, const int* nbr, int nq, int stride,
                       int kmax, const float* x, int ns, int xcd_chunk, int reps, float* ms_out,
                       unsigned long long* rows_out) {
  if (kmax > 64) return -5;
  hipStream_t st = 0;
  unsigned long long* rows;
  float* sink;
  hipMalloc(&rows, 8);
  hipMalloc(&sink, 64);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  int rc = 0;
  (void)hipGetLastError();   // a failed earlier call must not poison this one
  for (int r = -1; r < reps && rc == 0; ++r) {
    if (r == 0) hipEventRecord(a, st);
    if (r <= 0) hipMemsetAsync(rows, 0, 8, st);
    if (mode == 0) { DISPATCH_L(0) } else if (mode == 1) { DISPATCH_L(1) } else if (mode == 2) { DISPATCH_L(2) } else rc = -6;
  }
  hipEventRecord(b, st);
  if (hipStreamSynchronize(st) != hipSuccess) rc = -7;
  float ms = 0.f;
  if (rc == 0) hipEventElapsedTime(&ms, a, b);
  *ms_out = ms / (float)reps;
  unsigned long long h = 0;
  hipMemcpy(&h, rows, 8, hipMemcpyDeviceToHost);
  *rows_out = h / (unsigned long long)(reps > 0 ? reps : 1);
  hipFree(rows);
  hipFree(sink);
  hipEventDestroy(a);
  hipEventDestroy(b);
  return rc;
}
