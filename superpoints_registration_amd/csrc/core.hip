// Error reporting, version and an on-device self test of the two f32 MFMA
// fragment layouts every kernel in this library relies on.
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "spr_common.h"

namespace spr {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- per-launch timing records (ProfScope, spr_common.h) ----------------------
struct ProfRec {
  hipEvent_t beg, end;
  int code, n;
};
static std::vector<ProfRec> g_prof;
static std::mutex g_prof_mu;
static std::atomic<bool> g_prof_on{false};
bool prof_enabled() { return g_prof_on.load(std::memory_order_relaxed); }
void prof_push(hipEvent_t beg, hipEvent_t end, int code, int n) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back(ProfRec{beg, end, code, n});
}

namespace {
// max |x| of a [rows, cols] view (row stride `stride` floats) as kAmaxParts
// per-block partials.  Fixed grid of kAmaxParts blocks: every slot is written.
// cols4 = cols / 4 when the view is float4-addressable (cols, stride % 4 == 0 and
// 16-byte aligned base), else 0 -> scalar path.
struct AbsmaxJob {
  const float* x;
  long rows, stride;
  int cols, cols4;
  float* parts;
};
// blockIdx.y selects the tensor (a second, usually small one -- the weights -- rides along in
// the same launch)
// |a|, with NaN mapped to +inf: a range that "saw" a NaN must not read as finite (the consumers treat a
// non-finite range as "this tensor is broken" -- e.g. the fixed-point scatter-adds of backward.hip turn
// their whole output into NaN instead of silently dropping the NaN contribution).
__device__ __forceinline__ float amag(float a) { return a != a ? INFINITY : fabsf(a); }

__global__ __launch_bounds__(256) void k_absmax(AbsmaxJob j0, AbsmaxJob j1) {
  const AbsmaxJob j = blockIdx.y == 0 ? j0 : j1;
  const float* __restrict__ x = j.x;
  const long rows = j.rows, stride = j.stride;
  const int cols = j.cols, cols4 = j.cols4;
  float* __restrict__ parts = j.parts;
  __shared__ float sh[4];
  float m = 0.f;
  const long nthr = (long)gridDim.x * 256, t0 = (long)blockIdx.x * 256 + threadIdx.x;
  if (cols4 > 0) {
    const long total = rows * cols4;
    if (stride == cols) {
      const float4* p = reinterpret_cast<const float4*>(x);
      long i = t0;
      for (; i + 3 * nthr < total; i += 4 * nthr) {   // four 16-byte loads in flight
        const float4 a = p[i], b = p[i + nthr], c = p[i + 2 * nthr], d = p[i + 3 * nthr];
        m = fmaxf(m, fmaxf(fmaxf(amag(a.x), amag(a.y)), fmaxf(amag(a.z), amag(a.w))));
        m = fmaxf(m, fmaxf(fmaxf(amag(b.x), amag(b.y)), fmaxf(amag(b.z), amag(b.w))));
        m = fmaxf(m, fmaxf(fmaxf(amag(c.x), amag(c.y)), fmaxf(amag(c.z), amag(c.w))));
        m = fmaxf(m, fmaxf(fmaxf(amag(d.x), amag(d.y)), fmaxf(amag(d.z), amag(d.w))));
      }
      for (; i < total; i += nthr) {
        const float4 a = p[i];
        m = fmaxf(m, fmaxf(fmaxf(amag(a.x), amag(a.y)), fmaxf(amag(a.z), amag(a.w))));
      }
    } else {
      for (long i = t0; i < total; i += nthr) {
        const long r = i / cols4;
        const int c = (int)(i - r * cols4);
        const float4 a = *reinterpret_cast<const float4*>(x + r * stride + 4 * c);
        m = fmaxf(m, fmaxf(fmaxf(amag(a.x), amag(a.y)), fmaxf(amag(a.z), amag(a.w))));
      }
    }
  } else {
    const long total = rows * cols;
    for (long i = t0; i < total; i += nthr) {
      const long r = i / cols;
      m = fmaxf(m, amag(x[r * stride + (i - r * cols)]));
    }
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) parts[blockIdx.x] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// D[16x16] = A[16x4] B[4x16] with A[l&15][l>>4], B[l>>4][l&15];
// D: col = l&15, row = 4*(l>>4) + r
__global__ void k_test_16x16x4(const float* A, const float* B, float* D) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)],
                                             acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}
// D[32x32] = A[32x2] B[2x32] with A[l&31][l>>5], B[l>>5][l&31];
// D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
__global__ void k_test_32x32x2(const float* A, const float* B, float* D) {
  const int l = threadIdx.x;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * 2 + (l >> 5)], B[(l >> 5) * 32 + (l & 31)],
                                             acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r)
    D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
}  // namespace
}  // namespace spr

using namespace spr;

namespace {
std::mutex g_dev_mu;
std::map<std::pair<int, const void*>, int> g_lds_attr;   // (device, kernel) -> bytes granted
std::map<int, int> g_cu_count;                            // device -> CUs
}  // namespace

int spr::ensure_dyn_lds(const void* kernel, int bytes) {
  if (bytes <= 64 * 1024) return 0;
  int dev = 0;
  SPR_HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_dev_mu);
  auto key = std::make_pair(dev, kernel);
  auto it = g_lds_attr.find(key);
  if (it != g_lds_attr.end() && it->second >= bytes) return 0;
  SPR_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  g_lds_attr[key] = bytes;
  return 0;
}

int spr::device_cu_count() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  std::lock_guard<std::mutex> lk(g_dev_mu);
  auto it = g_cu_count.find(dev);
  if (it != g_cu_count.end()) return it->second;
  hipDeviceProp_t prop;
  int n = 256;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    n = prop.multiProcessorCount;
  g_cu_count[dev] = n;
  return n;
}

namespace {
int make_job(const float* x, long rows, int cols, long stride, float* parts, AbsmaxJob* j) {
  SPR_REQUIRE(x != nullptr && parts != nullptr && rows >= 0 && cols >= 1 && stride >= cols,
              "absmax: bad arguments (rows=%ld cols=%d stride=%ld)", rows, cols, stride);
  const bool v4 = cols % 4 == 0 && stride % 4 == 0 && ((uintptr_t)x & 15) == 0;
  *j = AbsmaxJob{x, rows, stride, cols, v4 ? cols / 4 : 0, parts};
  return 0;
}
}  // namespace

int spr::launch_absmax(const float* x, long rows, int cols, long stride, float* parts, hipStream_t stream) {
  AbsmaxJob j;
  if (int rc = make_job(x, rows, cols, stride, parts, &j)) return rc;
  hipLaunchKernelGGL(k_absmax, dim3(kAmaxParts, 1), dim3(256), 0, stream, j, j);
  SPR_LAUNCH_CHECK();
  return 0;
}

int spr::launch_absmax2(const float* x0, long rows0, int cols0, long stride0, float* parts0, const float* x1,
                        long rows1, int cols1, long stride1, float* parts1, hipStream_t stream) {
  AbsmaxJob j0, j1;
  if (int rc = make_job(x0, rows0, cols0, stride0, parts0, &j0)) return rc;
  if (int rc = make_job(x1, rows1, cols1, stride1, parts1, &j1)) return rc;
  hipLaunchKernelGGL(k_absmax, dim3(kAmaxParts, 2), dim3(256), 0, stream, j0, j1);
  SPR_LAUNCH_CHECK();
  return 0;
}

// max |x| of MANY contiguous tensors in one launch (the weights of a model at the start of a training step: ~150
// tensors of 10^2 .. 10^6 elements, each of which used to cost its own 512-workgroup launch).  jobs_dev: njobs records
// {pointer, element count}; parts_out [njobs][parts_per_job].
struct AbsmaxMultiJob {
  const float* x;
  long long n;
};
namespace {
__global__ __launch_bounds__(256) void k_absmax_multi(const AbsmaxMultiJob* __restrict__ jobs, int parts_per_job,
                                                      float* __restrict__ parts_out) {
  const AbsmaxMultiJob j = jobs[blockIdx.y];
  __shared__ float sh[4];
  const long chunk = (j.n + parts_per_job - 1) / parts_per_job;
  const long i0 = (long)blockIdx.x * chunk, i1 = i0 + chunk < j.n ? i0 + chunk : j.n;
  float m = 0.f;
  for (long i = i0 + threadIdx.x; i < i1; i += 256) m = fmaxf(m, amag(j.x[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) parts_out[(size_t)blockIdx.y * parts_per_job + blockIdx.x] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
}  // namespace
extern "C" int spr_absmax_multi(const void* jobs_dev, int njobs, int parts_per_job, float* parts_out, void* stream_) {
  SPR_REQUIRE(jobs_dev != nullptr && parts_out != nullptr && njobs >= 1 && njobs <= 65535 && parts_per_job >= 1 &&
                  parts_per_job <= 1024, "absmax_multi: bad arguments");
  hipLaunchKernelGGL(k_absmax_multi, dim3(parts_per_job, njobs), dim3(256), 0, (hipStream_t)stream_,
                     (const AbsmaxMultiJob*)jobs_dev, parts_per_job, parts_out);
  SPR_LAUNCH_CHECK();
  return 0;
}

// Range of a tensor that does not change between calls (weights): measured once by the caller
// and handed to spr_linear_r / spr_kpconv_fwd_r as w_range (spr_range_parts() floats).
extern "C" int spr_range_parts(void) { return spr::kAmaxParts; }
extern "C" int spr_absmax(const float* x, long rows, int cols, long stride, float* parts, void* stream_) {
  SPR_REQUIRE(x != nullptr && parts != nullptr && rows >= 1 && cols >= 1 && stride >= cols, "absmax: bad arguments");
  return spr::launch_absmax(x, rows, cols, stride, parts, (hipStream_t)stream_);
}

extern "C" int spr_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof) {
    (void)hipEventDestroy(r.beg);
    (void)hipEventDestroy(r.end);
  }
  g_prof.clear();
  g_prof_on = on != 0;
  return 0;
}

extern "C" int spr_prof_read(int max_records, int* codes, int* nqs, float* ms) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  int n = 0;
  for (auto& r : g_prof) {
    if (n >= max_records) break;
    if (hipEventSynchronize(r.end) != hipSuccess) break;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.beg, r.end) != hipSuccess) break;
    codes[n] = r.code;
    nqs[n] = r.n;
    ms[n] = t;
    ++n;
  }
  // the records are consumed: release their events (a long-running caller that enables
  // profiling once would otherwise leak two events per launch)
  for (auto& r : g_prof) {
    (void)hipEventDestroy(r.beg);
    (void)hipEventDestroy(r.end);
  }
  g_prof.clear();
  return n;
}

extern "C" int spr_version(void) { return SPR_VERSION; }
extern "C" const char* spr_last_error(void) { return g_err; }

extern "C" int spr_selftest(int* status_host) {
  *status_host = -1;
  float *dA = nullptr, *dB = nullptr, *dD = nullptr;
  struct Free {   // releases the scratch on every return path
    float **a, **b, **d;
    ~Free() {
      if (*a) (void)hipFree(*a);
      if (*b) (void)hipFree(*b);
      if (*d) (void)hipFree(*d);
    }
  } guard{&dA, &dB, &dD};
  SPR_HIP_CHECK(hipMalloc(&dA, 4096));
  SPR_HIP_CHECK(hipMalloc(&dB, 4096));
  SPR_HIP_CHECK(hipMalloc(&dD, 8192));
  int bad = 0;
  {
    std::vector<float> A(64), B(64), D(256), R(256, 0.f);
    for (int i = 0; i < 16; ++i)
      for (int k = 0; k < 4; ++k) A[i * 4 + k] = (float)(1 + i * 3 + k * 7);   // asymmetric
    for (int k = 0; k < 4; ++k)
      for (int j = 0; j < 16; ++j) B[k * 16 + j] = (float)(2 + k * 5 - j * 2);
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j)
        for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
    SPR_HIP_CHECK(hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice));
    SPR_HIP_CHECK(hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_test_16x16x4, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    SPR_HIP_CHECK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    for (int i = 0; i < 256; ++i)
      if (D[i] != R[i]) bad |= 1;
  }
  {
    std::vector<float> A(64), B(64), D(1024), R(1024, 0.f);
    for (int i = 0; i < 32; ++i)
      for (int k = 0; k < 2; ++k) A[i * 2 + k] = (float)(1 + i * 3 + k * 11);
    for (int k = 0; k < 2; ++k)
      for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)(2 + k * 5 - j * 2);
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j)
        for (int k = 0; k < 2; ++k) R[i * 32 + j] += A[i * 2 + k] * B[k * 32 + j];
    SPR_HIP_CHECK(hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice));
    SPR_HIP_CHECK(hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_test_32x32x2, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    SPR_HIP_CHECK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
    for (int i = 0; i < 1024; ++i)
      if (D[i] != R[i]) bad |= 2;
  }
  *status_host = bad;
  if (bad) set_error("MFMA layout self test failed (mask %d)", bad);
  return bad ? 1 : 0;
}
