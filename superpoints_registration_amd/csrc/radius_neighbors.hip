// a2 -- batched fixed-radius neighbour search on gfx950.
//
// Behaviour contract: batch_nanoflann_neighbors()
//   /root/reference/src/models/backbone_kpconv/cpp_wrappers/cpp_neighbors/
//   neighbors/neighbors.cpp:211-332  (+ nanoflann.hpp:249 strict `dist <
//   radius`, :432-440 float32 metric accumulated x,y,z, :1287 sort by
//   distance) and the column slice of kpconv.py:258-262.
//
// The reference walks a kd-tree per query on one CPU thread.  Here:
//   1. per support cloud: bounding-box min (one workgroup per cloud);
//   2. supports binned into a uniform grid with cell = r*(1+2^-8) -- the
//      margin absorbs float rounding of the cell coordinate so that every
//      support with d2 < r2 is guaranteed to sit in the 3x3x3 cell
//      neighbourhood of the query (cell coordinates are < 8192 per axis);
//   3. rocPRIM radix sort by (cloud, cz, cy, cx); the sorted copy carries
//      xyz + original index as one 16-byte record, so the candidate scan is a
//      coalescable stream;
//   4. one thread per query: 9 (dz,dy) rows x one contiguous x-run each,
//      located by binary search in the sorted keys; exact reference d2
//      arithmetic (explicit _rn intrinsics, no FMA contraction); the `limit`
//      nearest by (d2, index) are kept in an LDS-resident sorted list
//      (slot-major layout -> conflict-free), rows are padded with ns.
// That is algo 1 (no limit on the clouds' extent).  The default, algo 0, replaces
// steps 3-4 by a dense per-cloud cell table (counting sort, no key search) and a
// scan + sort kernel pair -- see "Fast path" below.  Both give identical rows.
// Ties: the reference's std::sort leaves equal-d2 runs in kd-tree visit order;
// this kernel orders them by index (documented; tests canonicalise tie runs).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "spr_common.h"
#include <type_traits>

namespace spr {
namespace {

struct NbrCloud {
  float mn[3];
  int pad;
};

constexpr int kMaxCell = 8191;

__global__ __launch_bounds__(256) void k_min(const float* __restrict__ xyz,
                                             const int* __restrict__ cu, NbrCloud* info) {
  const int c = blockIdx.x;
  const int beg = cu[c], end = cu[c + 1];
  __shared__ float smn[3][256];
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f};
  for (int i = beg + threadIdx.x; i < end; i += blockDim.x)
    for (int d = 0; d < 3; ++d) mn[d] = fminf(mn[d], xyz[3 * (size_t)i + d]);
  for (int d = 0; d < 3; ++d) smn[d][threadIdx.x] = mn[d];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s)
      for (int d = 0; d < 3; ++d)
        smn[d][threadIdx.x] = fminf(smn[d][threadIdx.x], smn[d][threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    NbrCloud ci;
    for (int d = 0; d < 3; ++d) ci.mn[d] = (end > beg) ? smn[d][0] : 0.f;
    ci.pad = 0;
    info[c] = ci;
  }
}

__device__ __forceinline__ int cell_coord(float p, float mn, float inv_cell) {
  float v = (p - mn) * inv_cell;
  v = fminf(fmaxf(v, -4.0f), 20000.0f);
  return (int)floorf(v);
}

__device__ __forceinline__ unsigned long long pack_key(int c, int cz, int cy, int cx) {
  return ((unsigned long long)c << 48) | ((unsigned long long)cz << 32) |
         ((unsigned long long)cy << 16) | (unsigned long long)cx;
}

__global__ void k_cellkeys(const float* __restrict__ xyz, const int* __restrict__ cu, int n,
                           int nb, float inv_cell, const NbrCloud* __restrict__ info,
                           unsigned long long* keys, int* vals, int* err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = find_segment(cu, nb, i);
  const NbrCloud ci = info[c];
  const int cx = cell_coord(xyz[3 * (size_t)i + 0], ci.mn[0], inv_cell);
  const int cy = cell_coord(xyz[3 * (size_t)i + 1], ci.mn[1], inv_cell);
  const int cz = cell_coord(xyz[3 * (size_t)i + 2], ci.mn[2], inv_cell);
  if (cx > kMaxCell || cy > kMaxCell || cz > kMaxCell || cx < 0 || cy < 0 || cz < 0)
    atomicOr(err, 1);
  keys[i] = pack_key(c, cz & 0xffff, cy & 0xffff, cx & 0xffff);
  vals[i] = i;
}

__global__ void k_gather_sorted(const float* __restrict__ xyz, const int* __restrict__ vals,
                                int n, float4* rec) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int j = vals[i];
  rec[i] = make_float4(xyz[3 * (size_t)j + 0], xyz[3 * (size_t)j + 1], xyz[3 * (size_t)j + 2],
                       __int_as_float(j));
}

__device__ __forceinline__ bool nbr_less(float d2a, int ia, float d2b, int ib) {
  return (d2a < d2b) || (d2a == d2b && ia < ib);
}

// Dynamic LDS: float d2[limit][BLOCK], int id[limit][BLOCK].
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_query(
    const float* __restrict__ q_xyz, const int* __restrict__ q_cu, int nq,
    const int* __restrict__ s_cu, int ns, int nb, const NbrCloud* __restrict__ info,
    const unsigned long long* __restrict__ skeys, const float4* __restrict__ rec, float r2,
    float inv_cell, int limit, int* __restrict__ out, int* max_count) {
  extern __shared__ __align__(16) unsigned char smem[];
  float* l_d2 = (float*)smem;
  int* l_id = (int*)(smem + sizeof(float) * (size_t)limit * BLOCK);
  const int t = threadIdx.x;
  const int i = blockIdx.x * BLOCK + t;
  int total = 0;
  if (i < nq) {
    const int c = find_segment(q_cu, nb, i);
    const NbrCloud ci = info[c];
    const float qx = q_xyz[3 * (size_t)i + 0], qy = q_xyz[3 * (size_t)i + 1],
                qz = q_xyz[3 * (size_t)i + 2];
    const int cx = cell_coord(qx, ci.mn[0], inv_cell);
    const int cy = cell_coord(qy, ci.mn[1], inv_cell);
    const int cz = cell_coord(qz, ci.mn[2], inv_cell);
    const int sbeg = s_cu[c], send = s_cu[c + 1];
    int kept = 0;
    const int xlo = max(cx - 1, 0), xhi = min(cx + 1, kMaxCell);
    if (xhi >= 0 && xlo <= kMaxCell && send > sbeg) {
      for (int dz = -1; dz <= 1; ++dz) {
        const int z = cz + dz;
        if (z < 0 || z > kMaxCell) continue;
        for (int dy = -1; dy <= 1; ++dy) {
          const int y = cy + dy;
          if (y < 0 || y > kMaxCell) continue;
          const unsigned long long klo = pack_key(c, z, y, xlo);
          const unsigned long long khi = pack_key(c, z, y, xhi);
          // lower_bound(klo) in [sbeg, send)
          int lo = sbeg, hi = send;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (skeys[mid] < klo)
              lo = mid + 1;
            else
              hi = mid;
          }
          for (int j = lo; j < send; ++j) {
            if (skeys[j] > khi) break;
            const float4 s = rec[j];
            // nanoflann.hpp:432-440: diff = query - support; result += diff*diff
            const float dx = __fsub_rn(qx, s.x), dy2 = __fsub_rn(qy, s.y),
                        dzz = __fsub_rn(qz, s.z);
            float d2 = __fmul_rn(dx, dx);
            d2 = __fadd_rn(d2, __fmul_rn(dy2, dy2));
            d2 = __fadd_rn(d2, __fmul_rn(dzz, dzz));
            if (d2 < r2) {  // strict, nanoflann.hpp:249
              total++;
              const int sid = __float_as_int(s.w);
              int pos;
              if (kept < limit) {
                pos = kept++;
              } else {
                const int last = limit - 1;
                if (!nbr_less(d2, sid, l_d2[last * BLOCK + t], l_id[last * BLOCK + t]))
                  continue;
                pos = last;
              }
              // insertion: shift larger entries up
              while (pos > 0 &&
                     nbr_less(d2, sid, l_d2[(pos - 1) * BLOCK + t], l_id[(pos - 1) * BLOCK + t])) {
                l_d2[pos * BLOCK + t] = l_d2[(pos - 1) * BLOCK + t];
                l_id[pos * BLOCK + t] = l_id[(pos - 1) * BLOCK + t];
                --pos;
              }
              l_d2[pos * BLOCK + t] = d2;
              l_id[pos * BLOCK + t] = sid;
            }
          }
        }
      }
    }
    int* row = out + (size_t)i * limit;
    for (int k = 0; k < limit; ++k) row[k] = (k < kept) ? l_id[k * BLOCK + t] : ns;
  }
  // block max of total -> one atomic per wave
  int m = total;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
  if ((t & 63) == 0 && m > 0) atomicMax(max_count, m);
}


// ===========================================================================
// Fast path (algo 0): dense per-cloud cell table.
//   k_grid_dims   per cloud: bbox min/max -> grid dims (cells of r*(1+2^-8))
//   k_grid_offsets  prefix of the per-cloud cell counts (overflow -> error)
//   k_cell_count  one integer atomic per support point (spread over ~N cells)
//   rocPRIM exclusive scan over the cell counts -> cell starts
//   k_cell_scatter  counting-sort scatter of (xyz, index) records
//   k_scan_table  one thread per query, no LDS (full occupancy): the 9 (dz,dy)
//                 rows of the 3x3x3 neighbourhood are 9 contiguous record
//                 ranges whose bounds are 18 independent loads; candidates
//                 stream as 16-byte records, 8 in flight; in-range ones are
//                 appended as packed (d2, index) keys to a scratch row of
//                 capacity 2*limit.  Self searches walk the queries in cell order.
//   k_sort_rows   rank sort of each scratch row in LDS, cut to `limit`.
// The order inside a cell depends on atomic arrival order, the OUTPUT does
// not: rows are ordered by (d2, index) and the K-nearest cut uses the same
// total order.
struct GridCloud {
  float mn[3];
  int dim[3];
  long long off;  // first cell of this cloud in the table
};

__global__ __launch_bounds__(256) void k_grid_dims(const float* __restrict__ xyz,
                                                   const int* __restrict__ cu, float inv_cell,
                                                   GridCloud* info, int* err) {
  const int c = blockIdx.x;
  const int beg = cu[c], end = cu[c + 1];
  __shared__ float smn[3][256], smx[3][256];
  float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int i = beg + threadIdx.x; i < end; i += blockDim.x)
    for (int d = 0; d < 3; ++d) {
      const float v = xyz[3 * (size_t)i + d];
      mn[d] = fminf(mn[d], v);
      mx[d] = fmaxf(mx[d], v);
    }
  for (int d = 0; d < 3; ++d) {
    smn[d][threadIdx.x] = mn[d];
    smx[d][threadIdx.x] = mx[d];
  }
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s)
      for (int d = 0; d < 3; ++d) {
        smn[d][threadIdx.x] = fminf(smn[d][threadIdx.x], smn[d][threadIdx.x + s]);
        smx[d][threadIdx.x] = fmaxf(smx[d][threadIdx.x], smx[d][threadIdx.x + s]);
      }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    GridCloud g;
    for (int d = 0; d < 3; ++d) {
      g.mn[d] = (end > beg) ? smn[d][0] : 0.f;
      int n = 1;
      if (end > beg) {
        n = cell_coord(smx[d][0], g.mn[d], inv_cell) + 1;
        if (n < 1) n = 1;
        if (n > kMaxCell + 1) {
          atomicOr(err, 1);
          n = 1;
        }
      }
      g.dim[d] = n;
    }
    g.off = 0;
    info[c] = g;
  }
}

// Table header (ints): [0] error flags, [1] cells in use, [8 .. 8 + kTableSlots) max row count of the
// queries run against the table (k_scan_table raises its slot with atomicMax; zeroed here, at build).
constexpr int kTableSlots = 8;
constexpr int kHdrErr = 0, kHdrCells = 1, kHdrSlot0 = 8;
__global__ void k_grid_offsets(GridCloud* info, int nb, long long cap, int* hdr) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int* err = hdr + kHdrErr;
  for (int k = 0; k < kTableSlots; ++k) hdr[kHdrSlot0 + k] = 0;
  long long off = 0;
  for (int c = 0; c < nb; ++c) {
    info[c].off = off;
    off += (long long)info[c].dim[0] * info[c].dim[1] * info[c].dim[2];
    if (off > cap) {  // table too small for this geometry: flag, keep offsets in range
      atomicOr(err, 2);
      off = 0;
    }
  }
  hdr[kHdrCells] = (int)off;
}

// The cell table is sized for the worst geometry the fast path accepts (64 cells per support point + 2^20),
// a few per cent of which a voxelised cloud uses: 1 M of 34.6 M cells for 32 clouds of 16 384 points.
// Zeroing and scanning the whole capacity (hipMemsetAsync + rocPRIM scan: 0.28 GB of traffic per search
// at that size) is replaced by kernels launched over the capacity whose workgroups leave at once when
// their block of kScanBlock cells lies beyond the cells in use (a device-side count: no host round trip).
constexpr int kScanBlock = 4096;   // cells per workgroup: 256 threads x 16
__global__ __launch_bounds__(256) void k_table_zero(const int* __restrict__ hdr, int* __restrict__ count) {
  const long base = (long)blockIdx.x * kScanBlock;
  if (base > (long)hdr[kHdrCells]) return;          // cells [0, total] are used (entry `total` receives the sum)
  int4* p = reinterpret_cast<int4*>(count + base) + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 4; ++u) p[u * 256] = make_int4(0, 0, 0, 0);
}

__device__ __forceinline__ int block_sum_256(int v, int* sh /*[4]*/) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const int r = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void k_table_bsum(const int* __restrict__ hdr, const int* __restrict__ count,
                                                    int* __restrict__ bsum) {
  __shared__ int sh[4];
  const long base = (long)blockIdx.x * kScanBlock;
  if (base > (long)hdr[kHdrCells]) return;
  const int4* p = reinterpret_cast<const int4*>(count + base) + threadIdx.x;
  int v = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int4 q = p[u * 256];
    v += q.x + q.y + q.z + q.w;
  }
  v = block_sum_256(v, sh);
  if (threadIdx.x == 0) bsum[blockIdx.x] = v;
}

// in-place exclusive scan: count[] -> start[]; the block's offset is the sum of the block sums before it
__global__ __launch_bounds__(256) void k_table_scan(const int* __restrict__ hdr, int* __restrict__ count,
                                                    const int* __restrict__ bsum) {
  __shared__ int sh[4];
  __shared__ int wsum[4];
  const long base = (long)blockIdx.x * kScanBlock;
  if (base > (long)hdr[kHdrCells]) return;
  int off = 0;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) off += bsum[b];
  off = block_sum_256(off, sh);
  // thread t owns cells base + 16 t .. + 15
  int4* p = reinterpret_cast<int4*>(count + base) + 4 * threadIdx.x;
  int4 q[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) q[u] = p[u];
  int mine = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) mine += q[u].x + q[u].y + q[u].z + q[u].w;
  // exclusive prefix of `mine` over the 256 threads
  int incl = mine;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  int run = off + incl - mine;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) run += wsum[w];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    int4 o4;
    o4.x = run; run += q[u].x;
    o4.y = run; run += q[u].y;
    o4.z = run; run += q[u].z;
    o4.w = run; run += q[u].w;
    p[u] = o4;
  }
}

__device__ __forceinline__ long long grid_cell(const GridCloud& g, int cx, int cy, int cz) {
  return g.off + ((long long)cz * g.dim[1] + cy) * g.dim[0] + cx;
}

__global__ void k_cell_count(const float* __restrict__ xyz, const int* __restrict__ cu, int n, int nb,
                             float inv_cell, const GridCloud* __restrict__ info,
                             const int* __restrict__ err, int* count, int* cell_of, int* rank) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || *err) return;
  const int c = find_segment(cu, nb, i);
  const GridCloud g = info[c];
  int cx = cell_coord(xyz[3 * (size_t)i + 0], g.mn[0], inv_cell);
  int cy = cell_coord(xyz[3 * (size_t)i + 1], g.mn[1], inv_cell);
  int cz = cell_coord(xyz[3 * (size_t)i + 2], g.mn[2], inv_cell);
  cx = min(max(cx, 0), g.dim[0] - 1);
  cy = min(max(cy, 0), g.dim[1] - 1);
  cz = min(max(cz, 0), g.dim[2] - 1);
  const long long L = grid_cell(g, cx, cy, cz);
  cell_of[i] = (int)L;
  rank[i] = atomicAdd(&count[L], 1);   // arrival rank inside the cell: the scatter needs no second counter table
}

__global__ void k_cell_scatter(const float* __restrict__ xyz, int n, const int* __restrict__ err,
                               const int* __restrict__ cell_of, const int* __restrict__ start,
                               const int* __restrict__ rank, float4* rec) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || *err) return;
  const int L = cell_of[i];
  const int pos = start[L] + rank[i];
  rec[pos] = make_float4(xyz[3 * (size_t)i + 0], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2],
                         __int_as_float(i));
}

// Scan kernel: no LDS, ~30 VGPRs -> the CU runs at full wave occupancy, which
// is what hides the (L2) latency of the record stream.  Candidates inside the
// radius are appended unsorted to a per-query scratch row in global memory
// (fire-and-forget stores); only a query whose row is already full (more than
// `limit` supports in range) pays for reads: it replaces the current worst
// entry and rescans its row for the new worst.
// The nine record runs of a query live in NAMED registers (a nine-element array indexed by the running
// run number is demoted to scratch memory by the compiler: a scratch load in the address chain of every
// record load).  Run k as a packed (begin, end) pair picked by a select chain.
struct Runs9 {
  int2 r0, r1, r2, r3, r4, r5, r6, r7, r8;
};
__device__ __forceinline__ int2 pick9(int k, const Runs9& a) {
  int2 v = a.r0;
  v = (k == 1) ? a.r1 : v;
  v = (k == 2) ? a.r2 : v;
  v = (k == 3) ? a.r3 : v;
  v = (k == 4) ? a.r4 : v;
  v = (k == 5) ? a.r5 : v;
  v = (k == 6) ? a.r6 : v;
  v = (k == 7) ? a.r7 : v;
  v = (k == 8) ? a.r8 : v;
  return v;
}

// (d2, index) packed into one u64 whose unsigned order IS the neighbour order:
// d2 >= +0 so its bit pattern is monotone, index in the low word breaks ties.
__device__ __forceinline__ unsigned long long nbr_key(float d2, int id) {
  return ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)id;
}

// scratch-row capacity: twice the row width, at most 128 (LDS of k_sort_rows)
static inline int nbr_row_cap(int limit) { return limit * 2 < 128 ? limit * 2 : 128; }

__global__ __launch_bounds__(256) void k_scan_table(
    const float* __restrict__ q_xyz, const int* __restrict__ q_cu, int nq, int nb,
    const GridCloud* __restrict__ info, const int* __restrict__ start,
    const float4* __restrict__ rec, const int* __restrict__ err, float r2, float inv_cell,
    int cap, int limit, int self, unsigned long long* __restrict__ tmp_key, int* __restrict__ kept_out,
    int* __restrict__ qid_out, int* max_count) {
  constexpr int kBins = 16;
  __shared__ unsigned short l_hist[kBins * 256];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (*err) return;
  int total = 0;
  if (t < nq) {
    const int c = find_segment(q_cu, nb, t);
    const GridCloud g = info[c];
    // self search (queries == supports): walk the queries in CELL order -- thread t
    // takes the point of record t -- so the lanes of a wave scan the same cells and
    // their record streams hit the same cache lines; rows are independent, each is
    // written at its own query index
    int i = t;
    float qx, qy, qz;
    if (self) {
      const float4 me = rec[t];
      i = __float_as_int(me.w);
      qx = me.x, qy = me.y, qz = me.z;
    } else {
      qx = q_xyz[3 * (size_t)t + 0], qy = q_xyz[3 * (size_t)t + 1], qz = q_xyz[3 * (size_t)t + 2];
    }
    const int cx = cell_coord(qx, g.mn[0], inv_cell);
    const int cy = cell_coord(qy, g.mn[1], inv_cell);
    const int cz = cell_coord(qz, g.mn[2], inv_cell);
    const int xlo = max(cx - 1, 0), xhi = min(cx + 1, g.dim[0] - 1);
    auto run_of = [&](int k) -> int2 {
      const int z = cz + k / 3 - 1, y = cy + k % 3 - 1;
      const bool in = xlo <= xhi && z >= 0 && z < g.dim[2] && y >= 0 && y < g.dim[1];
      const long long L0 = in ? grid_cell(g, xlo, y, z) : 0;
      const long long L1 = in ? grid_cell(g, xhi, y, z) + 1 : 0;
      const int b = start[L0];
      return make_int2(b, in ? start[L1] : b);
    };
    Runs9 runs;
    runs.r0 = run_of(0); runs.r1 = run_of(1); runs.r2 = run_of(2);
    runs.r3 = run_of(3); runs.r4 = run_of(4); runs.r5 = run_of(5);
    runs.r6 = run_of(6); runs.r7 = run_of(7); runs.r8 = run_of(8);
    // scratch row of `cap` >= limit entries: a query with more than `limit` (but at most cap)
    // supports in range keeps them all and k_sort_rows cuts the row.  A query with MORE than cap
    // in range (dense regions: LiDAR ground near the sensor has 100-150 points within the first
    // level's radius against limits of ~40) is resolved with a second pass instead of the
    // replace-worst rescans of the row (52 rescans of 76 entries each on average for 150 in range:
    // 98 ms per forward on the KITTI-shaped workload): the first pass also histograms d2 of the
    // in-range candidates in kBins equal bins (per-thread counters in LDS); the bin in which the
    // cumulative count reaches `limit` gives a radius that still contains the `limit` nearest, and
    // the second pass collects only what lies inside it.
    // slot-major scratch: entry k of thread t at tmp_key[k * nq + t] -- the lanes of a wave append to
    // neighbouring addresses, and k_sort_rows reads every slot as one coalesced line per wave
    // (row-major rows of cap * 8 bytes cost one cache line per lane and load)
    unsigned long long* const row = tmp_key + t;
    const size_t rs = (size_t)nq;
    int kept = 0;
    unsigned long long worst = 0;   // largest key among the kept entries
    int w_pos = 0;
    const float bin_scale = (float)kBins / r2;
    unsigned short* my_hist = l_hist + threadIdx.x;          // [kBins][256] u16, one column per thread
#pragma unroll
    for (int b = 0; b < kBins; ++b) my_hist[b * 256] = 0;
    int cut_bin = kBins;            // second pass: accept bins <= cut_bin
    bool second = false;
    auto consider = [&](const float4 s) {
      // nanoflann.hpp:432-440: diff = query - support; result += diff*diff
      const float dx = __fsub_rn(qx, s.x), dy = __fsub_rn(qy, s.y), dz = __fsub_rn(qz, s.z);
      float d2 = __fmul_rn(dx, dx);
      d2 = __fadd_rn(d2, __fmul_rn(dy, dy));
      d2 = __fadd_rn(d2, __fmul_rn(dz, dz));
      if (!(d2 < r2)) return;  // strict, nanoflann.hpp:249
      const int bin = min((int)(d2 * bin_scale), kBins - 1);
      if (!second) {
        total++;
        // saturating 16-bit counter: the cut below only asks whether the cumulative count reaches
        // `limit` (<= 128), so a bin pinned at 65535 answers like its true count -- raw dense clouds
        // (tens of thousands of returns inside one radius) must not wrap it
        const unsigned short h = my_hist[bin * 256];
        my_hist[bin * 256] = h == 65535 ? h : (unsigned short)(h + 1);
        if (kept < cap) row[rs * kept++] = nbr_key(d2, __float_as_int(s.w));
        return;
      }
      if (bin > cut_bin) return;
      const unsigned long long key = nbr_key(d2, __float_as_int(s.w));
      if (kept < cap) {
        row[rs * kept] = key;
        if (kept == 0 || key > worst) {
          worst = key;
          w_pos = kept;
        }
        ++kept;
        return;
      }
      // more than cap candidates inside the cut bin (ties / lattices): replace-worst, rare
      if (!(key < worst)) return;
      row[rs * w_pos] = key;
      __threadfence_block();   // our own store must be visible to the rescan below
      worst = key;
      for (int k = 0; k < cap; ++k) {
        const unsigned long long rk = row[rs * k];
        if (rk > worst) {
          worst = rk;
          w_pos = k;
        }
      }
    };
    // the 9 record runs as ONE candidate sequence, 8 records in flight
    auto scan = [&]() {
      int k = 0, j = runs.r0.x, e = runs.r0.y;
      while (k < 9) {
        float4 s8[8];
        bool v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          while (k < 9 && j >= e) {
            ++k;
            const int2 r = pick9(k, runs);
            j = r.x;
            e = r.y;
          }
          v8[u] = k < 9;
          s8[u] = rec[v8[u] ? j : 0];
          j += v8[u] ? 1 : 0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (v8[u]) consider(s8[u]);
      }
    };
    scan();
    if (total > cap) {
      int cum = 0;
      cut_bin = kBins - 1;
      for (int b = 0; b < kBins; ++b) {
        cum += my_hist[b * 256];
        if (cum >= limit) {
          cut_bin = b;
          break;
        }
      }
      second = true;
      kept = 0;
      scan();
    }
    kept_out[t] = kept;
    qid_out[t] = i;
  }
  int m = total;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(max_count, m);
}

// Sort kernel: one thread per query row; keys staged slot-major in LDS
// (conflict-free) and ordered by ranking, four ranks per pass over the row so
// each LDS read feeds four compares.  LDS holds the first `lslots` entries of a row (the launcher passes the
// scratch capacity: whole rows); entries beyond would be re-read from the slot-major scratch.  (Measured
// alternatives: no LDS at all, every pass from the scratch at full occupancy -- 136 vs 92 us per call;
// lslots = limit -- 88 vs 92 us on voxelised clouds, 1.9 ms per call on LiDAR-shaped scans.)
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_sort_rows(const unsigned long long* __restrict__ tmp_key,
                                                     const int* __restrict__ kept_in,
                                                     const int* __restrict__ qid_in,
                                                     const int* __restrict__ err, int nq, int ns,
                                                     int limit, int lslots, int* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned long long* l_key = (unsigned long long*)smem;
  const int t = threadIdx.x;
  const int i = blockIdx.x * BLOCK + t;
  if (*err || i >= nq) return;
  const int kept = kept_in[i];
  const unsigned long long* src = tmp_key + i;          // slot-major scratch of k_scan_table's thread i
  const size_t rs = (size_t)nq;
  const int staged = min(kept, lslots);
  for (int k = 0; k < staged; ++k) l_key[k * BLOCK + t] = src[rs * k];
  auto key_at = [&](int k) { return k < lslots ? l_key[k * BLOCK + t] : src[rs * k]; };
  int* row = out + (size_t)qid_in[i] * limit;
  constexpr unsigned long long kInf = ~0ull;
  for (int a = 0; a < kept; a += 4) {
    const unsigned long long k0 = key_at(a);
    const unsigned long long k1 = a + 1 < kept ? key_at(a + 1) : kInf;
    const unsigned long long k2 = a + 2 < kept ? key_at(a + 2) : kInf;
    const unsigned long long k3 = a + 3 < kept ? key_at(a + 3) : kInf;
    int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    for (int b = 0; b < staged; ++b) {
      const unsigned long long kb = l_key[b * BLOCK + t];
      r0 += kb < k0 ? 1 : 0;
      r1 += kb < k1 ? 1 : 0;
      r2 += kb < k2 ? 1 : 0;
      r3 += kb < k3 ? 1 : 0;
    }
    for (int b = staged; b < kept; ++b) {
      const unsigned long long kb = src[rs * b];
      r0 += kb < k0 ? 1 : 0;
      r1 += kb < k1 ? 1 : 0;
      r2 += kb < k2 ? 1 : 0;
      r3 += kb < k3 ? 1 : 0;
    }
    // ranks are a permutation of 0..kept-1; the row keeps the `limit` nearest
    if (r0 < limit) row[r0] = (int)(unsigned)k0;
    if (a + 1 < kept && r1 < limit) row[r1] = (int)(unsigned)k1;
    if (a + 2 < kept && r2 < limit) row[r2] = (int)(unsigned)k2;
    if (a + 3 < kept && r3 < limit) row[r3] = (int)(unsigned)k3;
  }
  for (int k = kept; k < limit; ++k) row[k] = ns;
}

// Round 5: the same rank sort with the row in REGISTERS (rows of at most N entries; the launcher caps the scan's
// scratch rows at N for limits <= N).  No LDS at all: k_sort_rows needs cap * 8 bytes of LDS per thread (40 KB per
// 64-thread block at the 3DMatch limit), i.e. four waves per CU at best -- and none while a main-stream kernel that
// owns the CU's LDS (the KPConv ring: 160 KB) is resident, which is where its 465 us per call in the round-4 bench
// profile came from (150 us alone).  Every index into the key array is a compile-time constant (the loops are fully
// unrolled; a wave leaves both loops at its longest row through wave-uniform branches), so the array lives in VGPRs.
// Results are those of k_sort_rows, entry for entry.
// ---------------------------------------------------------------------------------------------------
// Round 5: one-pass K-nearest selection, one WAVE per query (neighbors.cpp:272-327 semantics).
// k_scan_table + k_sort_rows(_reg) give every query a thread: the 64 lanes of a wave stream 64 different record
// runs (one 128-byte line per lane and load for a 16-byte record unless neighbouring queries share cells), append
// to a slot-major scratch in global memory, and a dense row (LiDAR ground: 100-150 supports in range against a
// limit of ~40) pays a second scan.  Here the lanes of a wave take 64 CONSECUTIVE candidate records of one query
// (the nine x-runs of its 3 x 3 x 3 cells as one sequence: coalesced 1-KiB loads), and the row lives in the wave's
// registers, sorted by (d2, index), one entry per lane (two for limits > 64):
//   per batch: in-range candidates are ranked against the row and each other by ballots -- for each accepted
//   candidate j (wave-uniform loop over the ballot's set bits): rank_j = #{row entries < key_j} + #{accepted < key_j},
//   and every row entry counts the accepted keys below it -- and the union is written through 1 KiB of LDS per wave
//   into its new order, cut at `limit`.  Once the row is full only candidates below its last entry are accepted, so
//   a dense row costs a few ranks per batch, never a second pass.  No scratch, no sort kernel.
// Queries are handed out in chunks of consecutive indices (device counter): a wave's successive queries of a self
// search are neighbours in cell order and re-read the same record lines.
// Rows are bit for bit those of k_scan_table + k_sort_rows: same keys, same total order, same cut.
constexpr int kKnnChunk = 16;

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l) {
  const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, l);
  const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}

template <bool BIG>
__global__ __launch_bounds__(256) void k_knn_wave(
    const float* __restrict__ q_xyz, const int* __restrict__ q_cu, int nq, int nb,
    const GridCloud* __restrict__ info, const int* __restrict__ start, const float4* __restrict__ rec,
    const int* __restrict__ err, float r2, float inv_cell, int limit, int self, int ns, int* __restrict__ out,
    int* max_count, int* q_ctr) {
  __shared__ unsigned long long s_key[4][128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (*err) return;
  unsigned long long* sk = s_key[wave];
  constexpr unsigned long long kNone = ~0ull;
  int cl_lo = 0, cl_hi = 0;          // query range of the cached cloud
  GridCloud g = info[0];
  int wave_max = 0;
  for (;;) {
    int c0 = 0;
    if (lane == 0) c0 = atomicAdd(q_ctr, kKnnChunk);
    c0 = __builtin_amdgcn_readfirstlane(c0);
    if (c0 >= nq) break;
    const int c1 = min(c0 + kKnnChunk, nq);
    for (int t = c0; t < c1; ++t) {
      if (t < cl_lo || t >= cl_hi) {      // wave uniform
        const int c = find_segment(q_cu, nb, t);
        cl_lo = q_cu[c];
        cl_hi = q_cu[c + 1];
        g = info[c];
      }
      int qi = t;
      float qx, qy, qz;
      if (self) {                         // the query IS record t (cell order)
        const float4 me = rec[t];
        qi = __float_as_int(me.w);
        qx = me.x, qy = me.y, qz = me.z;
      } else {
        qx = q_xyz[3 * (size_t)t + 0], qy = q_xyz[3 * (size_t)t + 1], qz = q_xyz[3 * (size_t)t + 2];
      }
      const int cx = cell_coord(qx, g.mn[0], inv_cell);
      const int cy = cell_coord(qy, g.mn[1], inv_cell);
      const int cz = cell_coord(qz, g.mn[2], inv_cell);
      const int xlo = max(cx - 1, 0), xhi = min(cx + 1, g.dim[0] - 1);
      // lane k < 9: record run k (the x-cells xlo..xhi of one (y, z) row of the neighbourhood)
      int rb = 0, re = 0;
      if (lane < 9) {
        const int z = cz + lane / 3 - 1, y = cy + lane % 3 - 1;
        const bool in = xlo <= xhi && z >= 0 && z < g.dim[2] && y >= 0 && y < g.dim[1];
        if (in) {
          rb = start[grid_cell(g, xlo, y, z)];
          re = start[grid_cell(g, xhi, y, z) + 1];
        }
      }
      // exclusive prefix of the run lengths over lanes 0..8
      const int len = re - rb;
      int pre = len;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        const int v = __shfl_up(pre, o, 64);
        if (lane >= o) pre += v;
      }
      const int n_c = __builtin_amdgcn_readlane(pre, 8);
      const int ex = pre - len;
      int roff[9], rdel[9];               // SGPRs: first candidate index of run k, record index - candidate index
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        roff[k] = __builtin_amdgcn_readlane(ex, k);
        rdel[k] = __builtin_amdgcn_readlane(rb, k) - roff[k];
      }
      unsigned long long lk0 = kNone, lk1 = kNone;   // the row: rank `lane` (and 64 + lane)
      int cnt = 0, total = 0;
      // record of candidate `base + lane`, loaded ONE BATCH AHEAD: the L2 latency of a batch's 1 KiB of records (1-2 us:
      // a wave alone cannot hide it, and a dense row has seven batches) runs under the ranking of the batch before it.
      // (A plain load: the compiler tracks it and waits at the first use in the next iteration.  An inline-asm load is
      // wrong here -- the compiler believes the asm's output is written at once and copies the not-yet-loaded
      // registers into the loop-carried ones.)
      auto fetch = [&](int base) __attribute__((always_inline)) -> float4 {
        const int ci = base + lane;
        int d = rdel[0];
#pragma unroll
        for (int k = 1; k < 9; ++k) d = ci >= roff[k] ? rdel[k] : d;     // runs of length 0 are skipped over
        return rec[ci < n_c ? ci + d : 0];
      };
      float4 sp_next = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n_c > 0) sp_next = fetch(0);
      for (int base = 0; base < n_c; base += 64) {
        const int ci = base + lane;
        const bool valid = ci < n_c;
        const float4 sp = sp_next;
        if (base + 64 < n_c) sp_next = fetch(base + 64);
        // nanoflann.hpp:432-440: diff = query - support; result += diff*diff
        const float dx = __fsub_rn(qx, sp.x), dy = __fsub_rn(qy, sp.y), dz = __fsub_rn(qz, sp.z);
        float d2 = __fmul_rn(dx, dx);
        d2 = __fadd_rn(d2, __fmul_rn(dy, dy));
        d2 = __fadd_rn(d2, __fmul_rn(dz, dz));
        bool in = valid && d2 < r2;       // strict, nanoflann.hpp:249
        const unsigned long long key = nbr_key(d2, __float_as_int(sp.w));
        unsigned long long m = __builtin_amdgcn_ballot_w64(in);
        total += __popcll(m);
        if (cnt == limit) {               // full: only what beats the current last entry can enter
          const unsigned long long worst = (!BIG || limit <= 64) ? readlane_u64(lk0, limit - 1) : readlane_u64(lk1, limit - 65);
          in = in && key < worst;
          m = __builtin_amdgcn_ballot_w64(in);
        }
        if (m == 0) continue;
        int sh0 = 0, sh1 = 0, myrank = 0;
        for (unsigned long long mm = m; mm != 0; mm &= mm - 1) {
          const int jl = __builtin_ctzll(mm);
          const unsigned long long kj = readlane_u64(key, jl);
          int nl = __popcll(__builtin_amdgcn_ballot_w64(lk0 < kj));     // empty slots hold ~0: never below
          sh0 += kj < lk0 ? 1 : 0;
          if (BIG) {
            nl += __popcll(__builtin_amdgcn_ballot_w64(lk1 < kj));
            sh1 += kj < lk1 ? 1 : 0;
          }
          nl += __popcll(__builtin_amdgcn_ballot_w64(in && key < kj));
          if (lane == jl) myrank = nl;
        }
        // the union in its new order (ranks are distinct: keys carry the support index), cut at `limit`
        if (lk0 != kNone && lane + sh0 < limit) sk[lane + sh0] = lk0;
        if (BIG && lk1 != kNone && 64 + lane + sh1 < limit) sk[64 + lane + sh1] = lk1;
        if (in && myrank < limit) sk[myrank] = key;
        cnt = min(cnt + (int)__popcll(m), limit);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        lk0 = lane < cnt ? sk[lane] : kNone;
        if (BIG) lk1 = 64 + lane < cnt ? sk[64 + lane] : kNone;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      int* row = out + (size_t)qi * limit;
      if (lane < limit) row[lane] = lane < cnt ? (int)(unsigned)lk0 : ns;
      if (BIG && 64 + lane < limit) row[64 + lane] = 64 + lane < cnt ? (int)(unsigned)lk1 : ns;
      wave_max = max(wave_max, total);
    }
  }
  if (lane == 0 && wave_max > 0 && wave_max > *reinterpret_cast<volatile int*>(max_count)) atomicMax(max_count, wave_max);
}

// f(integral_constant<int, I>) for I = B .. E-1: every index is a compile-time constant in the callee (a plain
// `#pragma unroll` nest of 64 x 64 iterations is not unrolled by the compiler: the key array then went to scratch)
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

template <int N>
__global__ __launch_bounds__(256) void k_sort_rows_reg(const unsigned long long* __restrict__ tmp_key,
                                                       const int* __restrict__ kept_in, const int* __restrict__ qid_in,
                                                       const int* __restrict__ err, int nq, int ns, int limit,
                                                       int* __restrict__ out) {
  static_assert(N % 8 == 0, "rows in chunks of eight");
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (*err) return;
  const bool act = i < nq;
  const int kept = act ? min(kept_in[i], N) : 0;
  int kmax = kept;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) kmax = max(kmax, __shfl_xor(kmax, o, 64));
  kmax = __builtin_amdgcn_readfirstlane(kmax);
  const unsigned long long* src = tmp_key + (act ? i : 0);
  const size_t rs = (size_t)nq;
  unsigned long long key[N];
  static_for<0, N>([&](auto kc) { key[decltype(kc)::value] = ~0ull; });
  static_for<0, N / 8>([&](auto cc) {
    constexpr int k0 = 8 * decltype(cc)::value;
    if (k0 < kmax) {
      static_for<k0, k0 + 8>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if (k < kept) key[k] = src[rs * k];
      });
    }
  });
  int* row = out + (size_t)(act ? qid_in[i] : 0) * limit;
  static_for<0, N>([&](auto ac) {
    constexpr int a = decltype(ac)::value;
    if (a < kmax) {
      int r = 0;
      static_for<0, N / 8>([&](auto cc) {
        constexpr int b0 = 8 * decltype(cc)::value;
        if (b0 < kmax) {
          static_for<b0, b0 + 8>([&](auto bc) { r += key[decltype(bc)::value] < key[a] ? 1 : 0; });   // padding (~0) is never smaller
        }
      });
      // ranks are a permutation of 0..kept-1; the row keeps the `limit` nearest
      if (a < kept && r < limit) row[r] = (int)(unsigned)key[a];
    }
  });
  if (act)
    for (int k = kept; k < limit; ++k) row[k] = ns;
}

__global__ void k_nbr_err2(const int* hdr, int slot, int* max_count) {
  const int err = hdr[kHdrErr];
  if (err & 1) *max_count = -1;       // extent / radius too large
  else if (err & 2) *max_count = -2;  // cell table overflow -> caller retries with algo 1
  else *max_count = hdr[kHdrSlot0 + slot];
}

size_t table_cap_cells(int ns) { return (size_t)64 * (size_t)(ns > 0 ? ns : 1) + ((size_t)1 << 20); }

// ---- cell table as an object of its own: built once per (supports, radius), queried several times -------
// (the pyramid asks for the conv, the pool and the previous level's up-sampling neighbours against the
// same supports with the same radius: kpconv.py:352, :377, :384 of the reference)
// blob: header int[64] | GridCloud[nb] | start int[cells_alloc] | records float4[ns]
size_t table_cells_alloc(int ns) { return align_up(table_cap_cells(ns) + 1, kScanBlock); }
struct TableView {
  int* hdr;
  GridCloud* ginfo;
  int* start;
  float4* rec;
};
size_t table_bytes(int ns, int nb) {
  return align_up(64 * sizeof(int), 256) + align_up(sizeof(GridCloud) * (size_t)(nb > 0 ? nb : 1), 256) +
         align_up(4 * table_cells_alloc(ns), 256) + align_up(16 * (size_t)(ns > 0 ? ns : 1), 256);
}
TableView table_view(void* blob, int ns, int nb) {
  Workspace w(blob, table_bytes(ns, nb));
  TableView t;
  t.hdr = w.take<int>(64);
  t.ginfo = w.take<GridCloud>(nb > 0 ? nb : 1);
  t.start = w.take<int>(table_cells_alloc(ns));
  t.rec = w.take<float4>(ns > 0 ? ns : 1);
  return t;
}
size_t table_build_ws_bytes(int ns) {
  return 2 * align_up(4 * (size_t)(ns > 0 ? ns : 1), 256) + align_up(4 * (table_cells_alloc(ns) / kScanBlock), 256);
}
size_t table_query_ws_bytes(int nq) {
  const size_t Q = (size_t)(nq > 0 ? nq : 1);
  return align_up(8 * Q * 128, 256) + 2 * align_up(4 * Q, 256);   // unsorted (d2, id) key rows (cap <= 128), kept counts, query ids
}

int table_build(const float* s_xyz, const int* s_cu, int ns, int nb, float radius, void* blob, void* ws,
                size_t ws_bytes, hipStream_t stream) {
  const TableView t = table_view(blob, ns, nb);
  Workspace w(ws, ws_bytes);
  int* cell_of = w.take<int>((size_t)ns);
  int* rank = w.take<int>((size_t)ns);
  const int nblk = (int)(table_cells_alloc(ns) / kScanBlock);
  int* bsum = w.take<int>((size_t)nblk);
  SPR_REQUIRE(bsum != nullptr, "radius table: workspace carve failed");
  const float inv_cell = 1.0f / (radius * (1.0f + 1.0f / 256.0f));
  const int TB = 256;
  SPR_HIP_CHECK(hipMemsetAsync(t.hdr, 0, 8 * sizeof(int), stream));
  hipLaunchKernelGGL(k_grid_dims, dim3(nb), dim3(256), 0, stream, s_xyz, s_cu, inv_cell, t.ginfo, t.hdr + kHdrErr);
  hipLaunchKernelGGL(k_grid_offsets, dim3(1), dim3(64), 0, stream, t.ginfo, nb, (long long)table_cap_cells(ns), t.hdr);
  hipLaunchKernelGGL(k_table_zero, dim3(nblk), dim3(256), 0, stream, t.hdr, t.start);
  hipLaunchKernelGGL(k_cell_count, dim3(cdiv(ns, TB)), dim3(TB), 0, stream, s_xyz, s_cu, ns, nb, inv_cell, t.ginfo,
                     t.hdr + kHdrErr, t.start, cell_of, rank);
  hipLaunchKernelGGL(k_table_bsum, dim3(nblk), dim3(256), 0, stream, t.hdr, t.start, bsum);
  hipLaunchKernelGGL(k_table_scan, dim3(nblk), dim3(256), 0, stream, t.hdr, t.start, bsum);
  hipLaunchKernelGGL(k_cell_scatter, dim3(cdiv(ns, TB)), dim3(TB), 0, stream, s_xyz, ns, t.hdr + kHdrErr, cell_of,
                     t.start, rank, t.rec);
  SPR_LAUNCH_CHECK();
  return 0;
}

// algo: 0 = thread per query (k_scan_table + rank sort), 1 = wave per query (k_knn_wave), -1 = the library's default
// (0, or 1 under SPR_NBR_ALGO=wave).  Rows are identical either way.
int table_query(const float* q_xyz, const int* q_cu, int nq, int self, int ns, int nb, float radius, int limit,
                int slot, const void* blob, int* out_idx, int* max_count, void* ws, size_t ws_bytes,
                hipStream_t stream, int algo = -1) {
  const TableView t = table_view(const_cast<void*>(blob), ns, nb);
  Workspace w(ws, ws_bytes);
  // limits up to kRegRows: the scratch rows are capped there and sorted in registers (k_sort_rows_reg); a query with
  // more candidates in range than the cap takes the scan's second (histogram-cut) pass, as before beyond 2 * limit
  constexpr int kRegRows = 64;
  static const bool reg_sort = [] { const char* e = getenv("SPR_NBR_LDS_SORT"); return e == nullptr || e[0] != '1'; }();
  const bool use_reg = reg_sort && limit <= kRegRows;
  const int rcap = use_reg ? min(nbr_row_cap(limit), kRegRows) : nbr_row_cap(limit);
  unsigned long long* tmp_key = w.take<unsigned long long>((size_t)nq * rcap);
  int* kept = w.take<int>((size_t)nq);
  int* qid = w.take<int>((size_t)nq);
  SPR_REQUIRE(qid != nullptr, "radius query: workspace carve failed");
  const float r2 = radius * radius;  // neighbors.cpp:226 (float32)
  const float inv_cell = 1.0f / (radius * (1.0f + 1.0f / 256.0f));
  // SPR_NBR_ALGO=wave: the one-pass wave-per-query selection (k_knn_wave) instead of scan + rank sort
  static const int env_algo = [] {
    const char* e = getenv("SPR_NBR_ALGO");
    return e == nullptr ? -1 : (e[0] == 'w' ? 1 : 0);     // "wave" / "scan": force one of them (A/B)
  }();
  const bool knn_wave = env_algo >= 0 ? env_algo == 1 : algo == 1;
  if (knn_wave) {
    SPR_HIP_CHECK(hipMemsetAsync(qid, 0, sizeof(int), stream));
    const int nblk = min(cdiv(nq, 4 * kKnnChunk), 8 * device_cu_count());
    if (limit <= 64)
      hipLaunchKernelGGL(k_knn_wave<false>, dim3(nblk), dim3(256), 0, stream, q_xyz, q_cu, nq, nb, t.ginfo, t.start, t.rec,
                         t.hdr + kHdrErr, r2, inv_cell, limit, self, ns, out_idx, t.hdr + kHdrSlot0 + slot, qid);
    else
      hipLaunchKernelGGL(k_knn_wave<true>, dim3(nblk), dim3(256), 0, stream, q_xyz, q_cu, nq, nb, t.ginfo, t.start, t.rec,
                         t.hdr + kHdrErr, r2, inv_cell, limit, self, ns, out_idx, t.hdr + kHdrSlot0 + slot, qid);
    hipLaunchKernelGGL(k_nbr_err2, dim3(1), dim3(1), 0, stream, t.hdr, slot, max_count);
    SPR_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(k_scan_table, dim3(cdiv(nq, 256)), dim3(256), 0, stream, q_xyz, q_cu, nq, nb, t.ginfo, t.start,
                     t.rec, t.hdr + kHdrErr, r2, inv_cell, rcap, limit, self, tmp_key, kept, qid, t.hdr + kHdrSlot0 + slot);
  if (use_reg) {
    hipLaunchKernelGGL(k_sort_rows_reg<kRegRows>, dim3(cdiv(nq, 256)), dim3(256), 0, stream, tmp_key, kept, qid,
                       t.hdr + kHdrErr, nq, ns, limit, out_idx);
    hipLaunchKernelGGL(k_nbr_err2, dim3(1), dim3(1), 0, stream, t.hdr, slot, max_count);
    SPR_LAUNCH_CHECK();
    return 0;
  }
  constexpr int BLOCK = 64;   // cap <= 128 -> at most 64 KB of LDS
  // whole rows in LDS: staging only the first `limit` entries gave 4 % on voxelised clouds (most rows are
  // shorter than `limit`) but 1.9 ms per call on LiDAR-shaped scans, whose rows fill the 2 * limit scratch
  hipLaunchKernelGGL(k_sort_rows<BLOCK>, dim3(cdiv(nq, BLOCK)), dim3(BLOCK), (size_t)rcap * 8 * BLOCK, stream, tmp_key,
                     kept, qid, t.hdr + kHdrErr, nq, ns, limit, rcap, out_idx);
  hipLaunchKernelGGL(k_nbr_err2, dim3(1), dim3(1), 0, stream, t.hdr, slot, max_count);
  SPR_LAUNCH_CHECK();
  return 0;
}

__global__ void k_nbr_err(const int* err, int* max_count) {
  if (*err) *max_count = -1;
}

size_t nbr_sort_temp_bytes(int n) {
  size_t bytes = 0;
  rocprim::radix_sort_pairs(nullptr, bytes, (unsigned long long*)nullptr,
                            (unsigned long long*)nullptr, (int*)nullptr, (int*)nullptr,
                            (unsigned int)(n > 0 ? n : 1));
  return align_up(bytes, 256) + 256;
}

}  // namespace
}  // namespace spr

using namespace spr;

extern "C" size_t spr_radius_neighbors_workspace_bytes(int nq, int ns, int nb) {
  const size_t N = (size_t)(ns > 0 ? ns : 1), B = (size_t)(nb > 0 ? nb : 1);
  size_t b = 0;
  b += align_up(sizeof(NbrCloud) * B, 256);
  b += 2 * align_up(8 * N, 256);
  b += 2 * align_up(4 * N, 256);
  b += align_up(16 * N, 256);
  b += 256;
  b += nbr_sort_temp_bytes(ns);
  // cell-table path: the table itself, its build scratch, the query scratch
  b += table_bytes(ns, nb) + table_build_ws_bytes(ns) + table_query_ws_bytes(nq);
  return b;
}

extern "C" int spr_radius_neighbors(const float* q_xyz, const int* q_cu, int nq,
                                    const float* s_xyz, const int* s_cu, int ns, int nb,
                                    float radius, int limit, int algo, int* out_idx,
                                    int* max_count, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(nq > 0 && ns > 0 && nb >= 1, "radius_neighbors: empty input (nq=%d ns=%d)", nq, ns);
  SPR_REQUIRE(nb < 65536, "radius_neighbors: at most 65535 clouds per call");
  SPR_REQUIRE(radius > 0.f, "radius_neighbors: radius must be > 0");
  SPR_REQUIRE(limit >= 1 && limit <= 128, "radius_neighbors: limit must be in [1,128], got %d", limit);
  SPR_REQUIRE(ws_bytes >= spr_radius_neighbors_workspace_bytes(nq, ns, nb),
              "radius_neighbors: workspace too small");
  Workspace w(ws, ws_bytes);
  const size_t N = (size_t)ns;
  NbrCloud* info = w.take<NbrCloud>(nb);
  unsigned long long* keys = w.take<unsigned long long>(N);
  unsigned long long* keys2 = w.take<unsigned long long>(N);
  int* vals = w.take<int>(N);
  int* vals2 = w.take<int>(N);
  float4* rec = w.take<float4>(N);
  int* err = w.take<int>(16);
  size_t temp_bytes = nbr_sort_temp_bytes(ns);
  void* temp = w.take<char>(temp_bytes);
  SPR_REQUIRE(temp != nullptr, "radius_neighbors: workspace carve failed");

  const int TB = 256;
  if (algo == 0) {
    void* blob = w.take<char>(table_bytes(ns, nb));
    const size_t bws = table_build_ws_bytes(ns), qws = table_query_ws_bytes(nq);
    void* build_ws = w.take<char>(bws);
    void* query_ws = w.take<char>(qws);
    SPR_REQUIRE(query_ws != nullptr, "radius_neighbors: workspace carve failed (table)");
    if (int rc = table_build(s_xyz, s_cu, ns, nb, radius, blob, build_ws, bws, stream)) return rc;
    return table_query(q_xyz, q_cu, nq, (q_xyz == s_xyz && q_cu == s_cu && nq == ns) ? 1 : 0, ns, nb, radius, limit, 0,
                       blob, out_idx, max_count, query_ws, qws, stream);
  }
  const float r2 = radius * radius;  // neighbors.cpp:226 (float32)
  const float inv_cell = 1.0f / (radius * (1.0f + 1.0f / 256.0f));
  SPR_HIP_CHECK(hipMemsetAsync(err, 0, 16 * sizeof(int), stream));
  SPR_HIP_CHECK(hipMemsetAsync(max_count, 0, sizeof(int), stream));
  hipLaunchKernelGGL(k_min, dim3(nb), dim3(256), 0, stream, s_xyz, s_cu, info);
  hipLaunchKernelGGL(k_cellkeys, dim3(cdiv(ns, TB)), dim3(TB), 0, stream, s_xyz, s_cu, ns, nb,
                     inv_cell, info, keys, vals, err);
  SPR_LAUNCH_CHECK();
  size_t tb = temp_bytes;
  SPR_HIP_CHECK(rocprim::radix_sort_pairs(temp, tb, keys, keys2, vals, vals2,
                                          (unsigned int)ns, 0, 64, stream));
  hipLaunchKernelGGL(k_gather_sorted, dim3(cdiv(ns, TB)), dim3(TB), 0, stream, s_xyz, vals2, ns,
                     rec);
  if (limit <= 64) {
    constexpr int BLOCK = 128;
    const size_t lds = (size_t)limit * BLOCK * 8;
    hipLaunchKernelGGL(k_query<BLOCK>, dim3(cdiv(nq, BLOCK)), dim3(BLOCK), lds, stream, q_xyz,
                       q_cu, nq, s_cu, ns, nb, info, keys2, rec, r2, inv_cell, limit, out_idx,
                       max_count);
  } else {
    constexpr int BLOCK = 64;
    const size_t lds = (size_t)limit * BLOCK * 8;
    hipLaunchKernelGGL(k_query<BLOCK>, dim3(cdiv(nq, BLOCK)), dim3(BLOCK), lds, stream, q_xyz,
                       q_cu, nq, s_cu, ns, nb, info, keys2, rec, r2, inv_cell, limit, out_idx,
                       max_count);
  }
  hipLaunchKernelGGL(k_nbr_err, dim3(1), dim3(1), 0, stream, err, max_count);
  SPR_LAUNCH_CHECK();
  return 0;
}

// ---- cell table built once, queried several times (Preprocessor: conv / pool / up-sampling searches that share
// supports and radius).  Results are those of spr_radius_neighbors (algo 0), row for row. ----------------------
extern "C" size_t spr_radius_table_bytes(int ns, int nb) { return table_bytes(ns, nb); }
extern "C" size_t spr_radius_table_build_workspace_bytes(int ns, int nb) { (void)nb; return table_build_ws_bytes(ns); }
extern "C" size_t spr_radius_table_query_workspace_bytes(int nq) { return table_query_ws_bytes(nq); }
extern "C" int spr_radius_table_slots(void) { return kTableSlots; }

extern "C" int spr_radius_table_build(const float* s_xyz, const int* s_cu, int ns, int nb, float radius, void* table,
                                      size_t table_bytes_, void* ws, size_t ws_bytes, void* stream_) {
  SPR_REQUIRE(s_xyz && s_cu && table && ns > 0 && nb >= 1 && nb < 65536, "radius_table_build: bad arguments");
  SPR_REQUIRE(radius > 0.f, "radius_table_build: radius must be > 0");
  SPR_REQUIRE(table_bytes_ >= table_bytes(ns, nb) && ((uintptr_t)table & 255) == 0, "radius_table_build: table too small or misaligned");
  SPR_REQUIRE(ws != nullptr && ws_bytes >= table_build_ws_bytes(ns), "radius_table_build: workspace too small");
  return table_build(s_xyz, s_cu, ns, nb, radius, table, ws, ws_bytes, (hipStream_t)stream_);
}

// self != 0: the queries ARE the table's supports (same array, same cu) -> cell-order walk.  slot in
// [0, spr_radius_table_slots()): one per query call against this table build (its max row count).
// *max_count as in spr_radius_neighbors (-1 extent too large, -2 table overflow: use algo 1).
extern "C" int spr_radius_table_query(const float* q_xyz, const int* q_cu, int nq, int self, int ns, int nb,
                                      float radius, int limit, int slot, const void* table, int* out_idx,
                                      int* max_count, void* ws, size_t ws_bytes, void* stream_) {
  SPR_REQUIRE(q_xyz && q_cu && table && out_idx && max_count && nq > 0 && ns > 0 && nb >= 1, "radius_table_query: bad arguments");
  SPR_REQUIRE(limit >= 1 && limit <= 128, "radius_table_query: limit must be in [1,128], got %d", limit);
  SPR_REQUIRE(slot >= 0 && slot < kTableSlots, "radius_table_query: slot out of range");
  SPR_REQUIRE(!self || nq == ns, "radius_table_query: a self search has nq == ns");
  SPR_REQUIRE(ws != nullptr && ws_bytes >= table_query_ws_bytes(nq), "radius_table_query: workspace too small");
  return table_query(q_xyz, q_cu, nq, self ? 1 : 0, ns, nb, radius, limit, slot, table, out_idx, max_count, ws, ws_bytes,
                     (hipStream_t)stream_);
}

// The same with the selection algorithm chosen by the caller: 0 = one thread per query (scan of its 27 cells into a
// scratch row + rank sort in registers: the cheaper one for sparse rows), 1 = one wave per query (one pass, the row in
// registers, no scratch: the faster one when rows are dense -- more supports in range than `limit`).
extern "C" int spr_radius_table_query_a(const float* q_xyz, const int* q_cu, int nq, int self, int ns, int nb,
                                        float radius, int limit, int slot, const void* table, int* out_idx,
                                        int* max_count, int algo, void* ws, size_t ws_bytes, void* stream_) {
  SPR_REQUIRE(q_xyz && q_cu && table && out_idx && max_count && nq > 0 && ns > 0 && nb >= 1, "radius_table_query: bad arguments");
  SPR_REQUIRE(limit >= 1 && limit <= 128, "radius_table_query: limit must be in [1,128], got %d", limit);
  SPR_REQUIRE(slot >= 0 && slot < kTableSlots, "radius_table_query: slot out of range");
  SPR_REQUIRE(!self || nq == ns, "radius_table_query: a self search has nq == ns");
  SPR_REQUIRE(algo == 0 || algo == 1, "radius_table_query: algo must be 0 (thread per query) or 1 (wave per query)");
  SPR_REQUIRE(ws != nullptr && ws_bytes >= table_query_ws_bytes(nq), "radius_table_query: workspace too small");
  return table_query(q_xyz, q_cu, nq, self ? 1 : 0, ns, nb, radius, limit, slot, table, out_idx, max_count, ws, ws_bytes,
                     (hipStream_t)stream_, algo);
}
