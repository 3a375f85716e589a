// a1 -- batched voxel-grid barycentre subsampling on gfx950.
//
// Behaviour contract: batch_grid_subsampling()
//   /root/reference/src/models/backbone_kpconv/cpp_wrappers/cpp_subsampling/
//   grid_subsampling/grid_subsampling.cpp:5-106 (per cloud) and :109-204.
//
// MI355X design (not a translation of the hash-map loop):
//   1. one workgroup per cloud reduces the bounding box -> origin, nx, ny;
//   2. one thread per point builds a 64-bit key (cloud << 40 | voxel key);
//   3. rocPRIM radix sort of (key, point index) -- stable, so every voxel's
//      points stay in original order;
//   4. head flags + scan -> voxel ids; one thread per voxel sums its points
//      IN ORIGINAL ORDER in float32 and scales by (float)(1.0/count), which
//      reproduces the reference barycentre bit for bit;
//   5. output order: canonical (ascending key, spatially coherent -> better
//      gather locality downstream) or REFERENCE order.  The reference emits
//      voxels in libstdc++ std::unordered_map iteration order; that order is
//      re-derived on the GPU by replaying the hashtable's rehash epochs with
//      parallel passes (bucket-first-arrival atomics + scan), one workgroup
//      per cloud -- see k_umap_order.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <mutex>
#include <unordered_map>
#include <vector>

#include "spr_common.h"

namespace spr {
namespace {

constexpr int kKeyBits = 40;
constexpr unsigned long long kKeyMask = (1ull << kKeyBits) - 1;
constexpr int kMaxSched = 40;

struct CloudInfo {
  float org[3];
  unsigned int nx, ny;
};

// The bucket schedule travels as a kernel ARGUMENT (160 bytes): no per-device constant
// upload, no process-wide once-flag -- correct on whichever device the stream belongs to.
struct SchedArg {
  unsigned int v[kMaxSched];
};

// ---- bucket schedule of the platform's libstdc++ ---------------------------
// Measured, not assumed: drive a real std::unordered_map<size_t,char> and
// record every bucket_count() transition (same policy object the reference's
// std::unordered_map<size_t,SampledData> uses).
struct Schedule {
  std::vector<unsigned int> buckets;  // [0] = 1 (initial), then each growth
};
const Schedule& host_schedule() {
  static Schedule s;
  static std::once_flag once;
  std::call_once(once, [] {
    std::unordered_map<size_t, char> m;
    size_t bc = m.bucket_count();
    s.buckets.push_back((unsigned int)bc);
    const size_t kMax = (size_t)1 << 22;  // clouds up to 4M voxels
    for (size_t i = 0; i < kMax; ++i) {
      m.emplace(i, 0);
      if (m.bucket_count() != bc) {
        bc = m.bucket_count();
        s.buckets.push_back((unsigned int)bc);
      }
    }
  });
  return s;
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bbox(const float* __restrict__ xyz,
                                              const int* __restrict__ cu,
                                              float dl, CloudInfo* info) {
  const int c = blockIdx.x;
  const int beg = cu[c], end = cu[c + 1];
  __shared__ float smn[3][256], smx[3][256];
  float mn[3], mx[3];
  if (end > beg) {
    for (int d = 0; d < 3; ++d) mn[d] = mx[d] = xyz[3 * (size_t)beg + d];
  } else {
    for (int d = 0; d < 3; ++d) mn[d] = mx[d] = 0.f;
  }
  for (int i = beg + threadIdx.x; i < end; i += blockDim.x) {
    for (int d = 0; d < 3; ++d) {
      float v = xyz[3 * (size_t)i + d];
      if (v < mn[d]) mn[d] = v;
      if (v > mx[d]) mx[d] = v;
    }
  }
  for (int d = 0; d < 3; ++d) {
    smn[d][threadIdx.x] = mn[d];
    smx[d][threadIdx.x] = mx[d];
  }
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      for (int d = 0; d < 3; ++d) {
        float a = smn[d][threadIdx.x + s];
        if (a < smn[d][threadIdx.x]) smn[d][threadIdx.x] = a;
        float b = smx[d][threadIdx.x + s];
        if (b > smx[d][threadIdx.x]) smx[d][threadIdx.x] = b;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    // grid_subsampling.cpp:27,30-31 -- float32, no contraction.
    const float inv = __fdiv_rn(1.0f, dl);
    CloudInfo ci;
    for (int d = 0; d < 3; ++d)
      ci.org[d] = __fmul_rn(floorf(__fmul_rn(smn[d][0], inv)), dl);
    ci.nx = (unsigned int)floorf(__fdiv_rn(__fsub_rn(smx[0][0], ci.org[0]), dl)) + 1u;
    ci.ny = (unsigned int)floorf(__fdiv_rn(__fsub_rn(smx[1][0], ci.org[1]), dl)) + 1u;
    info[c] = ci;
  }
}

__global__ void k_keys(const float* __restrict__ xyz, const int* __restrict__ cu,
                       int n, int nb, float dl, const CloudInfo* __restrict__ info,
                       unsigned long long* keys, int* vals, int* err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = find_segment(cu, nb, i);
  const CloudInfo ci = info[c];
  // grid_subsampling.cpp:53-56
  const unsigned long long ix =
      (unsigned long long)floorf(__fdiv_rn(__fsub_rn(xyz[3 * (size_t)i + 0], ci.org[0]), dl));
  const unsigned long long iy =
      (unsigned long long)floorf(__fdiv_rn(__fsub_rn(xyz[3 * (size_t)i + 1], ci.org[1]), dl));
  const unsigned long long iz =
      (unsigned long long)floorf(__fdiv_rn(__fsub_rn(xyz[3 * (size_t)i + 2], ci.org[2]), dl));
  const unsigned long long key =
      ix + (unsigned long long)ci.nx * iy + (unsigned long long)ci.nx * ci.ny * iz;
  if (key > kKeyMask) atomicOr(err, 1);
  keys[i] = ((unsigned long long)c << kKeyBits) | (key & kKeyMask);
  vals[i] = i;
}

__global__ void k_flags(const unsigned long long* __restrict__ keys, int n, int* flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// vid = exclusive scan of flags.  Writes voxel start offsets and nvox.
__global__ void k_segstart(const int* __restrict__ flags, const int* __restrict__ vid,
                           int n, int* vstart, int* nvox) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (flags[i]) vstart[vid[i]] = i;
  if (i == n - 1) *nvox = vid[i] + flags[i];
}

__global__ void k_bary(const float* __restrict__ xyz,
                       const unsigned long long* __restrict__ keys,
                       const int* __restrict__ vals, const int* __restrict__ vstart,
                       const int* __restrict__ nvox_p, int n, float* bary,
                       unsigned long long* vkey, int* vfirst) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  const int nvox = *nvox_p;
  if (v >= nvox) return;
  const int beg = vstart[v];
  const int end = (v + 1 < nvox) ? vstart[v + 1] : n;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int j = beg; j < end; ++j) {  // ascending original index (stable sort)
    const size_t p = (size_t)vals[j];
    sx = __fadd_rn(sx, xyz[3 * p + 0]);
    sy = __fadd_rn(sy, xyz[3 * p + 1]);
    sz = __fadd_rn(sz, xyz[3 * p + 2]);
  }
  // grid_subsampling.cpp:87: point * (1.0 / count)  -> float(double division)
  const float a = (float)(1.0 / (double)(end - beg));
  bary[3 * (size_t)v + 0] = __fmul_rn(sx, a);
  bary[3 * (size_t)v + 1] = __fmul_rn(sy, a);
  bary[3 * (size_t)v + 2] = __fmul_rn(sz, a);
  const unsigned long long k = keys[beg];
  vkey[v] = k;
  vfirst[v] = vals[beg];
}

// vcu[c] = first voxel of cloud c (voxels are sorted by (cloud, key)); one
// thread per cloud boundary, binary search instead of contended atomics.
__global__ void k_cloud_lb(const unsigned long long* __restrict__ vkey,
                           const int* __restrict__ nvox_p, int nb, int* vcu) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > nb) return;
  const int nvox = *nvox_p;
  int lo = 0, hi = nvox;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((int)(vkey[mid] >> kKeyBits) < c)
      lo = mid + 1;
    else
      hi = mid;
  }
  vcu[c] = lo;
}

// Single workgroup: per-cloud voxel offsets (vcu), output lens / offsets.
__global__ void k_cloud_offsets(const int* __restrict__ vcu, int nb, int max_p, int* out_cu,
                                int* out_lens, int* out_total) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int b = 0;
  for (int c = 0; c < nb; ++c) {
    out_cu[c] = b;
    const int m = vcu[c + 1] - vcu[c];
    const int keep = (max_p > 0 && m > max_p) ? max_p : m;
    out_lens[c] = keep;
    b += keep;
  }
  out_cu[nb] = b;
  *out_total = b;
}

__global__ void k_emit_canonical(const float* __restrict__ bary,
                                 const unsigned long long* __restrict__ vkey,
                                 const int* __restrict__ nvox_p,
                                 const int* __restrict__ vcu,
                                 const int* __restrict__ out_cu,
                                 const int* __restrict__ out_lens, float* out) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= *nvox_p) return;
  const int c = (int)(vkey[v] >> kKeyBits);
  const int r = v - vcu[c];
  if (r >= out_lens[c]) return;
  const size_t o = (size_t)(out_cu[c] + r);
  out[3 * o + 0] = bary[3 * (size_t)v + 0];
  out[3 * o + 1] = bary[3 * (size_t)v + 1];
  out[3 * o + 2] = bary[3 * (size_t)v + 2];
}

// Insertion sequence of the voxels = ascending (cloud, first point index) = ascending first point
// index (clouds are contiguous): every voxel marks its first point, a scan over the points ranks the
// marks, a compaction emits the voxel ids in that order.  (A 64-bit pair sort did this before: 17
// merge-sort launches per subsampling.)
__global__ void k_mark_first(const int* __restrict__ vfirst, const int* __restrict__ nvox_p, int* __restrict__ mark,
                             int* __restrict__ who) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= *nvox_p) return;
  const int f = vfirst[v];
  mark[f] = 1;
  who[f] = v;
}
__global__ void k_compact_first(const int* __restrict__ mark, const int* __restrict__ pos, const int* __restrict__ who,
                                int n, int* __restrict__ ins) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !mark[i]) return;
  ins[pos[i]] = who[i];
}

// ---------------------------------------------------------------------------
// libstdc++ unordered_map iteration order, replayed per cloud.
//
// State after every rehash epoch is the node list L.  Within an epoch with
// `nbk` buckets the arrival sequence is  A = L_prev ++ (keys inserted in the
// epoch, in insertion order); the resulting list is: buckets in DESCENDING
// order of the arrival position of their first node, each bucket's nodes in
// DESCENDING arrival position (hashtable.h _M_insert_bucket_begin /
// _M_rehash_aux: a node goes to the list head if its bucket is empty, else to
// the head of its bucket's chain).  Verified against std::unordered_map in
// tests/test_oracle_native.py (oracle) and tests/test_preprocess_gpu.py.
//
// Per epoch, in parallel over arrival positions p:
//   first[b] = min p, cnt[b]++, chain push          (atomics)
//   P = inclusive scan of (p is bucket head ? cnt[b] : 0)
//   new position = (total - P[first[b]]) + #{same bucket, pos > p}
// One 1024-thread workgroup per cloud; tables live in the global workspace.
__device__ void block_inclusive_scan(int* data, int m, int* lds /*[1024]*/) {
  const int t = threadIdx.x, T = blockDim.x;
  const int per = (m + T - 1) / T;
  const int beg = min(t * per, m), end = min(beg + per, m);
  int s = 0;
  for (int i = beg; i < end; ++i) s += data[i];
  lds[t] = s;
  __syncthreads();
  for (int o = 1; o < T; o <<= 1) {
    int v = (t >= o) ? lds[t - o] : 0;
    __syncthreads();
    lds[t] += v;
    __syncthreads();
  }
  int run = lds[t] - s;
  for (int i = beg; i < end; ++i) {
    run += data[i];
    data[i] = run;
  }
  __syncthreads();
}

// k % nbk for k < 2^40 (voxel keys), nbk < 2^32: quotient estimate in double (off by at most one), exact fix-up.
__device__ __forceinline__ int fast_mod(unsigned long long k, unsigned int nbk, double inv) {
  const unsigned long long q = (unsigned long long)((double)k * inv);
  long long r = (long long)k - (long long)(q * (unsigned long long)nbk);
  if (r < 0) r += nbk;
  else if (r >= (long long)nbk) r -= nbk;
  return (int)r;
}

// bucket tables (first arrival, count, chain head) of an epoch live in LDS while 3 * nbk ints fit
constexpr int kUmapLdsInts = 31744;   // 124 KB: every epoch up to 10 273 buckets (clouds of <= 10 273 voxels)

__global__ __launch_bounds__(1024) void k_umap_order(
    const unsigned long long* __restrict__ vkey, const int* __restrict__ ins,
    const int* __restrict__ vcu, const int* __restrict__ cu, int nsched, SchedArg sched,
    int* listA, int* listB, int* tfirst, int* tcnt, int* thead, int* nxt,
    int* scan, int* bucket, const float* __restrict__ bary, const int* __restrict__ out_cu,
    const int* __restrict__ out_lens, float* out) {
  __shared__ int lds[1024];
  extern __shared__ int lds_tab[];
  const int c = blockIdx.x;
  const int base = vcu[c];
  const int m = vcu[c + 1] - base;
  if (m <= 0) return;
  // per-cloud slices of the workspace
  int* la = listA + base;
  int* lb = listB + base;
  int* nx = nxt + base;
  int* sc = scan + base;
  int* bk = bucket + base;
  const size_t toff = 3 * (size_t)cu[c] + 16 * (size_t)c;
  const int t = threadIdx.x, T = blockDim.x;

  int done = 0;  // elements already in the list
  int e = 0;
  int* cur = la;
  int* nxtl = lb;
  while (done < m) {
    e += 1;
    if (e >= nsched) break;  // cannot happen: schedule covers 4M voxels
    const int nbk = (int)sched.v[e];
    const int upto = min(m, nbk);
    const int len = upto;  // arrival sequence length = done + (upto - done)
    const double inv = 1.0 / (double)nbk;
    auto epoch = [&](int* tf, int* tc, int* th) {
      for (int b = t; b < nbk; b += T) {
        tf[b] = 0x7fffffff;
        tc[b] = 0;
        th[b] = -1;
      }
      // new arrivals are elements done..upto-1 (insertion order)
      for (int p = done + t; p < upto; p += T) cur[p] = p;
      __syncthreads();
      for (int p = t; p < len; p += T) {
        const int el = cur[p];
        const int b = fast_mod(vkey[ins[base + el]] & kKeyMask, (unsigned int)nbk, inv);
        bk[p] = b;
        atomicMin(&tf[b], p);
        atomicAdd(&tc[b], 1);
        nx[p] = atomicExch(&th[b], p);
      }
      __syncthreads();
      for (int p = t; p < len; p += T) {
        const int b = bk[p];
        sc[p] = (tf[b] == p) ? tc[b] : 0;
      }
      __syncthreads();
      block_inclusive_scan(sc, len, lds);
      const int total = len;  // sum of all bucket counts
      for (int p = t; p < len; p += T) {
        const int b = bk[p];
        int greater = 0;
        for (int q = th[b]; q != -1; q = nx[q]) greater += (q > p) ? 1 : 0;
        const int pos = (total - sc[tf[b]]) + greater;
        nxtl[pos] = cur[p];
      }
      __syncthreads();
    };
    if (3 * nbk <= kUmapLdsInts) epoch(lds_tab, lds_tab + nbk, lds_tab + 2 * nbk);
    else epoch(tfirst + toff, tcnt + toff, thead + toff);
    int* tmp = cur;
    cur = nxtl;
    nxtl = tmp;
    done = upto;
  }
  // emit in list order
  const int keep = out_lens[c];
  const size_t ob = (size_t)out_cu[c];
  for (int p = t; p < keep; p += T) {
    const size_t v = (size_t)ins[base + cur[p]];
    out[3 * (ob + p) + 0] = bary[3 * v + 0];
    out[3 * (ob + p) + 1] = bary[3 * v + 1];
    out[3 * (ob + p) + 2] = bary[3 * v + 2];
  }
}

__global__ void k_check_err(const int* err, int* out_total) {
  if (*err) *out_total = -1;
}

size_t sort_temp_bytes(int n) {
  size_t bytes = 0;
  rocprim::radix_sort_pairs(nullptr, bytes, (unsigned long long*)nullptr,
                            (unsigned long long*)nullptr, (int*)nullptr, (int*)nullptr,
                            (unsigned int)(n > 0 ? n : 1));
  size_t sb = 0;
  rocprim::exclusive_scan(nullptr, sb, (int*)nullptr, (int*)nullptr, 0, (size_t)(n > 0 ? n : 1),
                          rocprim::plus<int>());
  return align_up(bytes > sb ? bytes : sb, 256) + 256;
}

}  // namespace
}  // namespace spr

using namespace spr;

extern "C" size_t spr_grid_subsample_workspace_bytes(int n, int nb) {
  const size_t N = (size_t)(n > 0 ? n : 1), B = (size_t)(nb > 0 ? nb : 1);
  size_t b = 0;
  b += align_up(sizeof(CloudInfo) * B, 256);
  b += 3 * align_up(8 * N, 256);           // keys in/out, vkey
  b += 10 * align_up(4 * N, 256);          // vals in/out, flags, vid, vstart, vfirst, sval x2, listA, listB
  b += 2 * align_up(4 * N, 256);           // nxt, scan
  b += align_up(12 * N, 256);              // bary
  b += 3 * align_up(4 * (3 * N + 16 * B + 64), 256);  // bucket tables
  b += 4 * align_up(4 * (B + 1), 256);     // cloud_cnt, vcu, out_cu, misc
  b += 1024;                               // scalars
  b += sort_temp_bytes(n);
  return b;
}

extern "C" int spr_grid_subsample(const float* xyz, const int* cu, int n, int nb,
                                  float dl, int max_p, int order_mode,
                                  float* out_xyz, int* out_lens, int* out_total,
                                  void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(n >= 0 && nb >= 1, "grid_subsample: bad sizes n=%d nb=%d", n, nb);
  SPR_REQUIRE(dl > 0.f, "grid_subsample: sampleDl must be > 0");
  SPR_REQUIRE(nb < (1 << 23), "grid_subsample: too many clouds");
  // reference: subsampling an empty batch is an error (wrapper.cpp:266-270)
  SPR_REQUIRE(n > 0, "grid_subsample: empty input");
  SPR_REQUIRE(ws_bytes >= spr_grid_subsample_workspace_bytes(n, nb),
              "grid_subsample: workspace too small");
  Workspace w(ws, ws_bytes);
  const size_t N = (size_t)n;
  CloudInfo* info = w.take<CloudInfo>(nb);
  unsigned long long* keys = w.take<unsigned long long>(N);
  unsigned long long* keys2 = w.take<unsigned long long>(N);
  unsigned long long* vkey = w.take<unsigned long long>(N);
  int* vals = w.take<int>(N);
  int* vals2 = w.take<int>(N);
  int* flags = w.take<int>(N);
  int* vid = w.take<int>(N);
  int* vstart = w.take<int>(N);
  int* vfirst = w.take<int>(N);
  int* sval = w.take<int>(N);
  int* sval2 = w.take<int>(N);
  int* listA = w.take<int>(N);
  int* listB = w.take<int>(N);
  int* nxt = w.take<int>(N);
  int* scan = w.take<int>(N);
  float* bary = w.take<float>(3 * N);
  const size_t tsz = 3 * N + 16 * (size_t)nb + 64;
  int* tfirst = w.take<int>(tsz);
  int* tcnt = w.take<int>(tsz);
  int* thead = w.take<int>(tsz);
  int* cloud_cnt = w.take<int>(nb + 1);
  int* vcu = w.take<int>(nb + 1);
  int* out_cu = w.take<int>(nb + 1);
  int* scalars = w.take<int>(64);  // [0]=nvox [1]=err
  size_t temp_bytes = sort_temp_bytes(n);
  void* temp = w.take<char>(temp_bytes);
  SPR_REQUIRE(temp != nullptr, "grid_subsample: workspace carve failed");
  int* nvox = scalars;
  int* err = scalars + 1;

  SPR_HIP_CHECK(hipMemsetAsync(scalars, 0, 64 * sizeof(int), stream));
  SPR_HIP_CHECK(hipMemsetAsync(cloud_cnt, 0, (nb + 1) * sizeof(int), stream));

  const int TB = 256;
  hipLaunchKernelGGL(k_bbox, dim3(nb), dim3(256), 0, stream, xyz, cu, dl, info);
  hipLaunchKernelGGL(k_keys, dim3(cdiv(n, TB)), dim3(TB), 0, stream, xyz, cu, n, nb, dl,
                     info, keys, vals, err);
  SPR_LAUNCH_CHECK();
  size_t tb = temp_bytes;
  // keys = cloud << kKeyBits | voxel: only the bits a batch of nb clouds can set are sorted (6 radix passes instead
  // of 8 for 128 clouds)
  int key_bits = kKeyBits + 1;
  while (key_bits < 64 && (1ull << (key_bits - kKeyBits)) < (unsigned long long)nb) ++key_bits;
  SPR_HIP_CHECK(rocprim::radix_sort_pairs(temp, tb, keys, keys2, vals, vals2,
                                          (unsigned int)n, 0, key_bits, stream));
  hipLaunchKernelGGL(k_flags, dim3(cdiv(n, TB)), dim3(TB), 0, stream, keys2, n, flags);
  tb = temp_bytes;
  SPR_HIP_CHECK(rocprim::exclusive_scan(temp, tb, flags, vid, 0, (size_t)n,
                                        rocprim::plus<int>(), stream));
  hipLaunchKernelGGL(k_segstart, dim3(cdiv(n, TB)), dim3(TB), 0, stream, flags, vid, n,
                     vstart, nvox);
  hipLaunchKernelGGL(k_bary, dim3(cdiv(n, TB)), dim3(TB), 0, stream, xyz, keys2, vals2,
                     vstart, nvox, n, bary, vkey, vfirst);
  hipLaunchKernelGGL(k_cloud_lb, dim3(cdiv(nb + 1, 256)), dim3(256), 0, stream, vkey, nvox, nb, vcu);
  hipLaunchKernelGGL(k_cloud_offsets, dim3(1), dim3(64), 0, stream, vcu, nb, max_p, out_cu,
                     out_lens, out_total);
  SPR_LAUNCH_CHECK();
  if (order_mode == SPR_ORDER_CANONICAL) {
    hipLaunchKernelGGL(k_emit_canonical, dim3(cdiv(n, TB)), dim3(TB), 0, stream, bary,
                       vkey, nvox, vcu, out_cu, out_lens, out_xyz);
  } else {
    const Schedule& s = host_schedule();
    SchedArg sched;
    for (int i = 0; i < kMaxSched; ++i)
      sched.v[i] = i < (int)s.buckets.size() ? s.buckets[i] : 0xffffffffu;
    const int nsched = (int)(s.buckets.size() < (size_t)kMaxSched ? s.buckets.size() : kMaxSched);
    // insertion order: flags / vid / vstart are free again after k_bary (mark, who, rank)
    SPR_HIP_CHECK(hipMemsetAsync(flags, 0, N * sizeof(int), stream));
    hipLaunchKernelGGL(k_mark_first, dim3(cdiv(n, TB)), dim3(TB), 0, stream, vfirst, nvox, flags, vid);
    tb = temp_bytes;
    SPR_HIP_CHECK(rocprim::exclusive_scan(temp, tb, flags, vstart, 0, (size_t)n, rocprim::plus<int>(), stream));
    hipLaunchKernelGGL(k_compact_first, dim3(cdiv(n, TB)), dim3(TB), 0, stream, flags, vstart, vid, n, sval2);
    const int umap_lds = kUmapLdsInts * (int)sizeof(int);
    if (int rc = ensure_dyn_lds((const void*)k_umap_order, umap_lds)) return rc;
    hipLaunchKernelGGL(k_umap_order, dim3(nb), dim3(1024), umap_lds, stream, vkey, sval2, vcu, cu,
                       nsched, sched, listA, listB, tfirst, tcnt, thead, nxt, scan, sval, bary, out_cu,
                       out_lens, out_xyz);
  }
  hipLaunchKernelGGL(k_check_err, dim3(1), dim3(1), 0, stream, err, out_total);
  SPR_LAUNCH_CHECK();
  return 0;
}

// ---- spatial walk order -----------------------------------------------------------------------------
// order[] = the points of every cloud sorted by the Morton code of their `cell`-sized grid cell (clouds in batch
// order).  Not part of any result: operators that gather rows of the PREVIOUS level for every point (max-pool, the
// first convolution) walk their queries in this order so that queries in flight together share neighbours and the
// rows they gather are served by L2 instead of being re-fetched from HBM (the points themselves stay in the
// reference's hash-map order).  The grid origin of a cloud is its first point: any fixed origin does.
namespace spr {
namespace {
__device__ __forceinline__ unsigned int spread10(unsigned int v) {     // 10 bits -> every third bit
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
__global__ void k_order_keys(const float* __restrict__ xyz, const int* __restrict__ cu, int n, int nb, float inv_cell,
                             unsigned long long* __restrict__ keys, int* __restrict__ vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int lo = 0, hi = nb;                               // cloud of point i
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (cu[mid] <= i) lo = mid; else hi = mid;
  }
  const int o = cu[lo];
  unsigned int c[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float d = (xyz[3 * (size_t)i + a] - xyz[3 * (size_t)o + a]) * inv_cell;
    const int q = (int)floorf(fminf(fmaxf(d, -512.f), 511.f)) + 512;
    c[a] = (unsigned int)q;
  }
  const unsigned int m = spread10(c[0]) | (spread10(c[1]) << 1) | (spread10(c[2]) << 2);
  keys[i] = ((unsigned long long)lo << 30) | m;
  vals[i] = i;
}
}  // namespace
}  // namespace spr

extern "C" size_t spr_cell_order_workspace_bytes(int n) {
  const size_t N = (size_t)(n > 0 ? n : 1);
  return 2 * align_up(8 * N, 256) + align_up(4 * N, 256) + sort_temp_bytes(n) + 256;
}

extern "C" int spr_cell_order(const float* xyz, const int* cu, int n, int nb, float cell, int* order, void* ws,
                              size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(n >= 1 && nb >= 1 && nb < (1 << 30) && cell > 0.f, "cell_order: bad arguments");
  SPR_REQUIRE(ws_bytes >= spr_cell_order_workspace_bytes(n), "cell_order: workspace too small");
  Workspace w(ws, ws_bytes);
  unsigned long long* keys = w.take<unsigned long long>(n);
  unsigned long long* keys2 = w.take<unsigned long long>(n);
  int* vals = w.take<int>(n);
  size_t tb = sort_temp_bytes(n);
  void* temp = w.take<char>(tb);
  SPR_REQUIRE(temp != nullptr, "cell_order: workspace carve failed");
  hipLaunchKernelGGL(k_order_keys, dim3(cdiv(n, 256)), dim3(256), 0, stream, xyz, cu, n, nb, 1.0f / cell, keys, vals);
  SPR_LAUNCH_CHECK();
  int bits = 31;
  while (bits < 60 && (1ull << (bits - 30)) < (unsigned long long)nb) ++bits;
  SPR_HIP_CHECK(rocprim::radix_sort_pairs(temp, tb, keys, keys2, vals, order, (unsigned int)n, 0, bits, stream));
  return 0;
}

// ---- voxel pre-downsampling, one point per voxel -----------------------------------
// Replaces the CPU voxel_down_sample of the KITTI loader (data_loaders/kitti_pred.py:12-14, :203-204,
// kiss_icp: voxel = (p / voxel_size) truncated toward zero per axis; the FIRST point that falls
// into a voxel is kept).  The kept SET of points is the reference's; they are emitted in
// ascending original index (kiss_icp emits hash-map order, which nothing downstream depends on).
namespace spr {
namespace {
__global__ void k_vox_keys(const float* __restrict__ xyz, int n, double voxel, unsigned long long* keys,
                           int* vals, int* err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long k = 0;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const double q = (double)xyz[3 * (size_t)i + d] / voxel;   // Eigen: (point / voxel_size).cast<int>()
    const long long c = (long long)q;                                    // truncation toward zero
    if (c < -(1ll << 20) || c >= (1ll << 20)) atomicExch(err, 1);
    k = (k << 21) | (unsigned long long)((c + (1ll << 20)) & 0x1fffff);
  }
  keys[i] = k;
  vals[i] = i;
}
__global__ void k_vox_heads(const unsigned long long* __restrict__ keys, const int* __restrict__ vals, int n,
                            unsigned int* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // stable sort: the first entry of an equal-key run carries the smallest original index
  const bool head = i == 0 || keys[i] != keys[i - 1];
  out[i] = head ? (unsigned int)vals[i] : 0xffffffffu;
}
__global__ void k_vox_count(const unsigned int* __restrict__ sorted, int n, int* out_idx, int* count,
                            const int* err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool ok = sorted[i] != 0xffffffffu;
  out_idx[i] = ok ? (int)sorted[i] : -1;
  if (ok && (i + 1 == n || sorted[i + 1] == 0xffffffffu)) count[0] = err[0] ? -1 : i + 1;
}
size_t vox_temp_bytes(int n) {
  size_t a = 0, b = 0;
  rocprim::radix_sort_pairs(nullptr, a, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (int*)nullptr,
                            (int*)nullptr, (unsigned int)(n > 0 ? n : 1), 0, 63);
  rocprim::radix_sort_keys(nullptr, b, (unsigned int*)nullptr, (unsigned int*)nullptr, (unsigned int)(n > 0 ? n : 1), 0,
                           32);
  return align_up(a > b ? a : b, 256);
}
}  // namespace
}  // namespace spr

extern "C" size_t spr_voxel_downsample_workspace_bytes(int n) {
  const size_t N = (size_t)(n > 0 ? n : 1);
  return 2 * align_up(8 * N, 256) + 4 * align_up(4 * N, 256) + 256 + spr::vox_temp_bytes(n);
}

// out_idx [n] i32: the first out_count[0] entries are the kept original indices (ascending);
// out_count[0] = -1 if a coordinate exceeds +-2^20 voxels.
extern "C" int spr_voxel_downsample(const float* xyz, int n, double voxel_size, int* out_idx, int* out_count, void* ws,
                                    size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(n >= 1 && voxel_size > 0.0 && xyz && out_idx && out_count, "voxel_downsample: bad arguments");
  SPR_REQUIRE(ws && ws_bytes >= spr_voxel_downsample_workspace_bytes(n), "voxel_downsample: workspace too small");
  Workspace w(ws, ws_bytes);
  unsigned long long* keys = w.take<unsigned long long>(n);
  unsigned long long* keys2 = w.take<unsigned long long>(n);
  int* vals = w.take<int>(n);
  int* vals2 = w.take<int>(n);
  unsigned int* sel = w.take<unsigned int>(n);
  unsigned int* sel2 = w.take<unsigned int>(n);
  int* err = w.take<int>(1);
  const size_t temp_bytes = vox_temp_bytes(n);
  void* temp = w.take<char>(temp_bytes);
  SPR_REQUIRE(temp != nullptr, "voxel_downsample: workspace carve failed");
  SPR_HIP_CHECK(hipMemsetAsync(err, 0, sizeof(int), stream));
  SPR_HIP_CHECK(hipMemsetAsync(out_count, 0, sizeof(int), stream));
  const int TBk = 256;
  hipLaunchKernelGGL(k_vox_keys, dim3(cdiv(n, TBk)), dim3(TBk), 0, stream, xyz, n, voxel_size, keys, vals, err);
  size_t tb = temp_bytes;
  SPR_HIP_CHECK(rocprim::radix_sort_pairs(temp, tb, keys, keys2, vals, vals2, (unsigned int)n, 0, 63, stream));
  hipLaunchKernelGGL(k_vox_heads, dim3(cdiv(n, TBk)), dim3(TBk), 0, stream, keys2, vals2, n, sel);
  tb = temp_bytes;
  SPR_HIP_CHECK(rocprim::radix_sort_keys(temp, tb, sel, sel2, (unsigned int)n, 0, 32, stream));
  hipLaunchKernelGGL(k_vox_count, dim3(cdiv(n, TBk)), dim3(TBk), 0, stream, sel2, n, out_idx, out_count, err);
  SPR_LAUNCH_CHECK();
  return 0;
}
