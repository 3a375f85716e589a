// a5 -- per-cloud InstanceNorm (+ residual add + LeakyReLU), strided max
// pooling and row gather on gfx950.
//
// Behaviour contract:
//   BatchNormBlock.forward  kpconv_blocks.py:497-525 (nn.InstanceNorm1d per
//     cloud: biased variance, eps inside the sqrt, no affine, no running stats)
//   LeakyReLU(0.1) after it  kpconv_blocks.py:553-561, :645, :727, :741
//   max_pool                kpconv_blocks.py:127-143
//
// The reference loops over clouds in Python (2B slices x ~30 norms per
// forward).  Here every norm is three stream-ordered launches over the packed
// [sum N, C] tensor, independent of the number of clouds:
//   1. k_in_stats   grid (cloud, split): per-channel sum / sum of squares in
//                   float64 over a row slice (deterministic: fixed slices,
//                   fixed-order final reduction, no atomics)
//   2. k_in_final   mean / rstd per (cloud, channel)
//   3. k_in_apply   float4 streaming pass: normalise, + residual, LeakyReLU
#include "spr_common.h"

namespace spr {
namespace {

// Row slices are relative to each cloud's own start and have a fixed length,
// so a cloud's statistics do not depend on what else is in the batch
// (bitwise batch invariance).
constexpr int kSliceRows = 512;
int in_nsplit(int max_len) {
  int s = cdiv(max_len > 0 ? max_len : 1, kSliceRows);
  return s < 1 ? 1 : s;
}

// grid (cloud, split, channel block of <= 64 channels); block = 256 threads =
// (256 / CW4) row lanes x CW4 float4 columns.  Every thread streams its rows
// with 4 independent 16-byte loads in flight and accumulates in float64; row
// lanes are combined through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void k_in_stats(const float* __restrict__ x,
                                                  const int* __restrict__ cu, int c, int nsplit,
                                                  double* __restrict__ part /*[nb][nsplit][2][c]*/) {
  const int cloud = blockIdx.x, split = blockIdx.y, cb = blockIdx.z * 64;
  const int beg = cu[cloud], end = cu[cloud + 1];
  const int r0 = beg + split * kSliceRows;
  const int r1 = min(r0 + kSliceRows, end);
  const int cw = min(c - cb, 64);
  const int cw4 = cw >> 2;
  const int rl = 256 / cw4;  // row lanes
  const int tc = threadIdx.x % cw4, tr = threadIdx.x / cw4;
  __shared__ double sh[8][256];
  double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
  if (tr < rl) {
    const float* base = x + cb + 4 * tc;
    int r = r0 + tr;
    for (; r + 3 * rl < r1; r += 4 * rl) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        v[u] = *reinterpret_cast<const float4*>(base + (size_t)(r + u * rl) * c);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double a = v[u].x, b = v[u].y, cc = v[u].z, d = v[u].w;
        s[0] += a; ss[0] += a * a;
        s[1] += b; ss[1] += b * b;
        s[2] += cc; ss[2] += cc * cc;
        s[3] += d; ss[3] += d * d;
      }
    }
    for (; r < r1; r += rl) {
      const float4 v = *reinterpret_cast<const float4*>(base + (size_t)r * c);
      const double a = v.x, b = v.y, cc = v.z, d = v.w;
      s[0] += a; ss[0] += a * a;
      s[1] += b; ss[1] += b * b;
      s[2] += cc; ss[2] += cc * cc;
      s[3] += d; ss[3] += d * d;
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    sh[k][threadIdx.x] = s[k];
    sh[4 + k][threadIdx.x] = ss[k];
  }
  __syncthreads();
  if (tr == 0) {
    double* p = part + (((size_t)cloud * nsplit + split) * 2) * c;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double a = s[k], b = ss[k];
      for (int q = 1; q < rl; ++q) {
        a += sh[k][q * cw4 + tc];
        b += sh[4 + k][q * cw4 + tc];
      }
      p[cb + 4 * tc + k] = a;
      p[c + cb + 4 * tc + k] = b;
    }
  }
}

__global__ void k_in_final(const double* __restrict__ part, const int* __restrict__ cu, int nb,
                           int c, int nsplit, float eps, float* __restrict__ mean,
                           float* __restrict__ rstd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nb * c) return;
  const int cloud = i / c, ch = i % c;
  double s = 0.0, ss = 0.0;
  for (int k = 0; k < nsplit; ++k) {
    const double* p = part + (((size_t)cloud * nsplit + k) * 2) * c;
    s += p[ch];
    ss += p[c + ch];
  }
  const int len = cu[cloud + 1] - cu[cloud];
  const double n = len > 0 ? (double)len : 1.0;
  const double m = s / n;
  double var = ss / n - m * m;
  if (var < 0.0) var = 0.0;
  mean[i] = (float)m;
  rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
}

// Operand-range hand-over (spr.h): the workgroup's max |out| joins one of kRangeSlots slots by
// integer atomic max on the bit pattern (non-negative floats order like unsigned ints; the result
// does not depend on the order of arrival).  The caller zero-initialises the slots.  Called by
// every thread of a 256-thread block.
__device__ __forceinline__ void publish_range(float mx, float* __restrict__ slots, int nslots) {
  __shared__ float shr[4];
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) shr[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicMax(reinterpret_cast<unsigned int*>(slots) + (blockIdx.x & (nslots - 1)),
              __float_as_uint(fmaxf(fmaxf(shr[0], shr[1]), fmaxf(shr[2], shr[3]))));
}

// Streaming pass, one float4 (4 channels of one row) per thread and step; every thread takes
// kInApplyUnroll steps a whole grid apart (all loads of a thread issued up front), so that a
// workgroup publishes its range once per kInApplyUnroll * 256 float4s.
constexpr int kInApplyUnroll = 4;
__global__ __launch_bounds__(256) void k_in_apply(const float* __restrict__ x, const int* __restrict__ cu, int n,
                                                  int nb, int c, int norm, const float* __restrict__ mean,
                                                  const float* __restrict__ rstd, const float* __restrict__ add,
                                                  float slope, float* __restrict__ out,
                                                  float* __restrict__ out_range, int nslots) {
  const int c4 = c >> 2;
  const long total = (long)n * c4;
  const long stride = (long)gridDim.x * blockDim.x;
  const long g0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  float4 v[kInApplyUnroll], a[kInApplyUnroll];
#pragma unroll
  for (int u = 0; u < kInApplyUnroll; ++u) {
    const long gid = g0 + u * stride;
    if (gid < total) {
      v[u] = reinterpret_cast<const float4*>(x)[gid];
      if (add) a[u] = reinterpret_cast<const float4*>(add)[gid];
    }
  }
  float mx = 0.f;
#pragma unroll
  for (int u = 0; u < kInApplyUnroll; ++u) {
    const long gid = g0 + u * stride;
    if (gid >= total) continue;
    float4 w = v[u];
    if (norm) {
      const int row = (int)(gid / c4), q = (int)(gid % c4);
      const int cloud = find_segment(cu, nb, row);
      const float4 m = reinterpret_cast<const float4*>(mean + (size_t)cloud * c)[q];
      const float4 r = reinterpret_cast<const float4*>(rstd + (size_t)cloud * c)[q];
      w.x = (w.x - m.x) * r.x;
      w.y = (w.y - m.y) * r.y;
      w.z = (w.z - m.z) * r.z;
      w.w = (w.w - m.w) * r.w;
    }
    if (add) {
      w.x += a[u].x;
      w.y += a[u].y;
      w.z += a[u].z;
      w.w += a[u].w;
    }
    w.x = w.x >= 0.f ? w.x : w.x * slope;
    w.y = w.y >= 0.f ? w.y : w.y * slope;
    w.z = w.z >= 0.f ? w.z : w.z * slope;
    w.w = w.w >= 0.f ? w.w : w.w * slope;
    reinterpret_cast<float4*>(out)[gid] = w;
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(w.x), fabsf(w.y)), fmaxf(fabsf(w.z), fabsf(w.w))));
  }
  if (out_range != nullptr) publish_range(mx, out_range, nslots);
}

// ---- backward of out = lrelu(IN(x) + add, slope) (kpconv_blocks.py:510-525, :553-561, :741) -----
// g = dout * (out >= 0 ? 1 : slope);  d add = g;
// d x = rstd * (g - mean_n(g) - xhat * mean_n(g xhat)),  xhat = (x - mean) rstd   (per cloud, channel)
// Same deterministic slicing as the forward statistics: partial sums of g and g*xhat in float64.
// grid (cloud, split, channel block of <= 64 channels) like k_in_stats: (256 / CW4) row lanes x CW4 float4
// columns, float64 partial sums per thread, row lanes combined through LDS in a fixed order.  (The first
// form -- one thread per channel walking its 512 rows -- used 32..128 threads of a workgroup: 5.5 ms per
// training step.)
__global__ __launch_bounds__(256) void k_in_bwd_stats(const float* __restrict__ x, const float* __restrict__ out,
                                                      const float* __restrict__ dout, const int* __restrict__ cu,
                                                      int c, int nsplit, float slope,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      double* __restrict__ part /*[nb][nsplit][2][c]*/) {
  const int cloud = blockIdx.x, split = blockIdx.y, cb = blockIdx.z * 64;
  const int beg = cu[cloud], end = cu[cloud + 1];
  const int r0 = beg + split * kSliceRows;
  const int r1 = min(r0 + kSliceRows, end);
  const int cw = min(c - cb, 64);
  const int cw4 = cw >> 2;
  const int rl = 256 / cw4;  // row lanes
  const int tc = threadIdx.x % cw4, tr = threadIdx.x / cw4;
  __shared__ double sh[8][256];
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (tr < rl) {
    const int col = cb + 4 * tc;
    const float4 mu = *reinterpret_cast<const float4*>(mean + (size_t)cloud * c + col);
    const float4 rs = *reinterpret_cast<const float4*>(rstd + (size_t)cloud * c + col);
    for (int r = r0 + tr; r < r1; r += rl) {
      const size_t o = (size_t)r * c + col;
      const float4 xv = *reinterpret_cast<const float4*>(x + o);
      const float4 ov = *reinterpret_cast<const float4*>(out + o);
      const float4 dv = *reinterpret_cast<const float4*>(dout + o);
      // (> 0, not >= 0: torch's leaky_relu backward takes the slope branch AT zero, and an output is exactly
      // zero once in ~10^7 elements -- x equal to the rounded mean -- which the BASELINE-size gradient test hits)
      const float g0 = dv.x * (ov.x > 0.f ? 1.f : slope), g1 = dv.y * (ov.y > 0.f ? 1.f : slope);
      const float g2 = dv.z * (ov.z > 0.f ? 1.f : slope), g3 = dv.w * (ov.w > 0.f ? 1.f : slope);
      s1[0] += (double)g0; s2[0] += (double)g0 * (double)((xv.x - mu.x) * rs.x);
      s1[1] += (double)g1; s2[1] += (double)g1 * (double)((xv.y - mu.y) * rs.y);
      s1[2] += (double)g2; s2[2] += (double)g2 * (double)((xv.z - mu.z) * rs.z);
      s1[3] += (double)g3; s2[3] += (double)g3 * (double)((xv.w - mu.w) * rs.w);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    sh[k][threadIdx.x] = s1[k];
    sh[4 + k][threadIdx.x] = s2[k];
  }
  __syncthreads();
  if (tr == 0) {
    double* p = part + (((size_t)cloud * nsplit + split) * 2) * c;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double a = s1[k], b = s2[k];
      for (int q = 1; q < rl; ++q) {
        a += sh[k][q * cw4 + tc];
        b += sh[4 + k][q * cw4 + tc];
      }
      p[cb + 4 * tc + k] = a;
      p[c + cb + 4 * tc + k] = b;
    }
  }
}

__global__ void k_in_bwd_final(const double* __restrict__ part, const int* __restrict__ cu, int nb, int c,
                               int nsplit, double* __restrict__ m1, double* __restrict__ m2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nb * c) return;
  const int cloud = i / c, ch = i % c;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < nsplit; ++k) {
    const double* p = part + (((size_t)cloud * nsplit + k) * 2) * c;
    s1 += p[ch];
    s2 += p[c + ch];
  }
  const int len = cu[cloud + 1] - cu[cloud];
  const double n = len > 0 ? (double)len : 1.0;
  // kept in float64: dx = rstd (g - m1 - xhat m2) sums to zero over a cloud only as exactly as m1 is
  // represented -- a float32 m1 leaves a common-mode residue of 3e-8 |m1| in every row, which a
  // downstream sum over all points (the first KPConv's weight gradient) amplifies by their number
  m1[i] = s1 / n;
  m2[i] = s2 / n;
}

__global__ void k_in_bwd_apply(const float* __restrict__ x, const float* __restrict__ out,
                               const float* __restrict__ dout, const int* __restrict__ cu, int n, int nb, int c,
                               int norm, float slope, const float* __restrict__ mean,
                               const float* __restrict__ rstd, const double* __restrict__ m1,
                               const double* __restrict__ m2, float* __restrict__ dx, float* __restrict__ dadd) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)n * c) return;
  const int row = (int)(gid / c), ch = (int)(gid % c);
  const float g = dout[gid] * (out[gid] > 0.f ? 1.f : slope);
  if (dadd) dadd[gid] = g;
  if (!norm) {
    dx[gid] = g;
    return;
  }
  const int cloud = find_segment(cu, nb, row);
  const size_t s = (size_t)cloud * c + ch;
  const float xh = (x[gid] - mean[s]) * rstd[s];
  dx[gid] = rstd[s] * (float)(((double)g - m1[s]) - (double)xh * m2[s]);
}

// the same, four channels per thread (c % 4 == 0, 16-byte aligned tensors): one segment search and one index
// division per float4 instead of per element; element arithmetic unchanged (bit for bit the results above)
__global__ void k_in_bwd_apply4(const float* __restrict__ x, const float* __restrict__ out,
                                const float* __restrict__ dout, const int* __restrict__ cu, int n, int nb, int c,
                                int norm, float slope, const float* __restrict__ mean,
                                const float* __restrict__ rstd, const double* __restrict__ m1,
                                const double* __restrict__ m2, float* __restrict__ dx, float* __restrict__ dadd) {
  const int c4 = c >> 2;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)n * c4) return;
  const int row = (int)(gid / c4), ch = 4 * (int)(gid % c4);
  const float4 d4 = reinterpret_cast<const float4*>(dout)[gid], o4 = reinterpret_cast<const float4*>(out)[gid];
  float g[4] = {d4.x * (o4.x > 0.f ? 1.f : slope), d4.y * (o4.y > 0.f ? 1.f : slope),
                d4.z * (o4.z > 0.f ? 1.f : slope), d4.w * (o4.w > 0.f ? 1.f : slope)};
  if (dadd) reinterpret_cast<float4*>(dadd)[gid] = make_float4(g[0], g[1], g[2], g[3]);
  if (!norm) {
    reinterpret_cast<float4*>(dx)[gid] = make_float4(g[0], g[1], g[2], g[3]);
    return;
  }
  const int cloud = find_segment(cu, nb, row);
  const size_t s = (size_t)cloud * c + ch;
  const float4 x4 = reinterpret_cast<const float4*>(x)[gid];
  const float xv[4] = {x4.x, x4.y, x4.z, x4.w};
  float r[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float xh = (xv[e] - mean[s + e]) * rstd[s + e];
    r[e] = rstd[s + e] * (float)(((double)g[e] - m1[s + e]) - (double)xh * m2[s + e]);
  }
  reinterpret_cast<float4*>(dx)[gid] = make_float4(r[0], r[1], r[2], r[3]);
}

// order != nullptr (spr_cell_order of the query points): the queries are walked in that order, every XCD a contiguous
// share of it (workgroups b and b + 8 share an L2) -- queries in flight together are neighbours in space, so the
// ~4.5 reads of every support row happen close together and hit L2 (the grid is then a multiple of 8).
__global__ void k_maxpool(const float* __restrict__ x, int ns, int c, const int* __restrict__ idx,
                          int nq, int idx_stride, int k, float* __restrict__ out, float* __restrict__ out_range,
                          int nslots, const int* __restrict__ order) {
  const int c4 = c >> 2;
  long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (order != nullptr) gid = ((long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * blockDim.x + threadIdx.x;
  float mx = 0.f;
  if (gid < (long)nq * c4) {
  const int q = (int)(gid % c4);
  const int row = order != nullptr ? order[gid / c4] : (int)(gid / c4);
  gid = (long)row * c4 + q;
  float4 m = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
  const int* ir = idx + (size_t)row * idx_stride;
  int j = 0;
  // four gathers in flight per thread (the serial one-load-per-iteration loop was latency bound)
  for (; j + 4 <= k; j += 4) {
    int id[4];
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) id[u] = ir[j + u];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);    // shadow row (kpconv_blocks.py:136)
      if (id[u] >= 0 && id[u] < ns) v[u] = reinterpret_cast<const float4*>(x + (size_t)id[u] * c)[q];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      m.x = fmaxf(m.x, v[u].x);
      m.y = fmaxf(m.y, v[u].y);
      m.z = fmaxf(m.z, v[u].z);
      m.w = fmaxf(m.w, v[u].w);
    }
  }
  for (; j < k; ++j) {
    const int id = ir[j];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (id >= 0 && id < ns) v = reinterpret_cast<const float4*>(x + (size_t)id * c)[q];
    m.x = fmaxf(m.x, v.x);
    m.y = fmaxf(m.y, v.y);
    m.z = fmaxf(m.z, v.z);
    m.w = fmaxf(m.w, v.w);
  }
  reinterpret_cast<float4*>(out)[gid] = m;
  mx = fmaxf(fmaxf(fabsf(m.x), fabsf(m.y)), fmaxf(fabsf(m.z), fabsf(m.w)));
  }
  if (out_range != nullptr) publish_range(mx, out_range, nslots);
}

__global__ void k_gather_rows(const float* __restrict__ x, int n_src, int c,
                              const int* __restrict__ idx, int n, float* __restrict__ out) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)n * c) return;
  const int row = (int)(gid / c), ch = (int)(gid % c);
  const int id = idx[row];
  out[gid] = (id >= 0 && id < n_src) ? x[(size_t)id * c + ch] : 0.f;
}

}  // namespace
}  // namespace spr

using namespace spr;

extern "C" size_t spr_instnorm_workspace_bytes(int max_len, int nb, int c) {
  const size_t B = (size_t)(nb > 0 ? nb : 1), C = (size_t)(c > 0 ? c : 1);
  const int ns = in_nsplit(max_len);
  return align_up(B * ns * 2 * C * sizeof(double), 256) + 2 * align_up(B * C * sizeof(float), 256);
}

extern "C" int spr_instnorm(const float* x, const int* cu, int n, int nb, int max_len_host, int c,
                            float eps, int norm, const float* add, float slope, float* out,
                            void* ws, size_t ws_bytes, void* stream_) {
  return spr_instnorm_r(x, cu, n, nb, max_len_host, c, eps, norm, add, slope, out, nullptr, 0, ws, ws_bytes, stream_);
}

// out_range: NULL, or out_range_n (a power of two) zero-initialised floats that receive partial
// maxima of |out| (operand-range hand-over to the consuming GEMM / KPConv); a few hundred slots
// keep the atomic traffic per address low on the largest tensors.
extern "C" int spr_instnorm_r(const float* x, const int* cu, int n, int nb, int max_len_host, int c,
                              float eps, int norm, const float* add, float slope, float* out, float* out_range,
                              int out_range_n, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(out_range == nullptr || (out_range_n >= 1 && (out_range_n & (out_range_n - 1)) == 0),
              "instnorm: out_range_n must be a power of two");
  SPR_REQUIRE(n > 0 && nb >= 1 && c >= 4 && c % 4 == 0, "instnorm: need n>0 and c %% 4 == 0 (c=%d)", c);
  float* mean = nullptr;
  float* rstd = nullptr;
  if (norm) {
    SPR_REQUIRE(max_len_host >= 1 && max_len_host <= n, "instnorm: bad max_len_host=%d", max_len_host);
    SPR_REQUIRE(ws_bytes >= spr_instnorm_workspace_bytes(max_len_host, nb, c), "instnorm: workspace too small");
    Workspace w(ws, ws_bytes);
    const int nsplit = in_nsplit(max_len_host);
    double* part = w.take<double>((size_t)nb * nsplit * 2 * c);
    mean = w.take<float>((size_t)nb * c);
    rstd = w.take<float>((size_t)nb * c);
    SPR_REQUIRE(rstd != nullptr, "instnorm: workspace carve failed");
    hipLaunchKernelGGL(k_in_stats, dim3(nb, nsplit, cdiv(c, 64)), dim3(256), 0, stream, x, cu, c, nsplit, part);
    hipLaunchKernelGGL(k_in_final, dim3(cdiv((long)nb * c, 256)), dim3(256), 0, stream, part, cu,
                       nb, c, nsplit, eps, mean, rstd);
    SPR_LAUNCH_CHECK();
  }
  const long total = (long)n * (c / 4);
  hipLaunchKernelGGL(k_in_apply, dim3(cdiv(total, 256 * kInApplyUnroll)), dim3(256), 0, stream, x, cu, n, nb, c,
                     norm, mean, rstd, add, slope, out, out_range, out_range_n);
  SPR_LAUNCH_CHECK();
  return 0;
}

// Statistics only: mean [nb][c] and rstd [nb][c] of a per-cloud InstanceNorm (the two passes k_in_stats / k_in_final
// of spr_instnorm, bit for bit), for consumers that normalise on load (spr_block_tail_n).
extern "C" int spr_instnorm_stats(const float* x, const int* cu, int n, int nb, int max_len_host, int c, float eps,
                                  float* mean, float* rstd, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(x && cu && mean && rstd && n > 0 && nb >= 1 && c >= 4 && c % 4 == 0, "instnorm_stats: bad arguments (c=%d)", c);
  SPR_REQUIRE(max_len_host >= 1 && max_len_host <= n, "instnorm_stats: bad max_len_host=%d", max_len_host);
  SPR_REQUIRE(ws_bytes >= spr_instnorm_workspace_bytes(max_len_host, nb, c), "instnorm_stats: workspace too small");
  Workspace w(ws, ws_bytes);
  const int nsplit = in_nsplit(max_len_host);
  double* part = w.take<double>((size_t)nb * nsplit * 2 * c);
  SPR_REQUIRE(part != nullptr, "instnorm_stats: workspace carve failed");
  hipLaunchKernelGGL(k_in_stats, dim3(nb, nsplit, cdiv(c, 64)), dim3(256), 0, stream, x, cu, c, nsplit, part);
  hipLaunchKernelGGL(k_in_final, dim3(cdiv((long)nb * c, 256)), dim3(256), 0, stream, part, cu, nb, c, nsplit, eps, mean,
                     rstd);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t spr_instnorm_bwd_workspace_bytes(int max_len, int nb, int c) {
  const size_t B = (size_t)(nb > 0 ? nb : 1), C = (size_t)(c > 0 ? c : 1);
  return spr_instnorm_workspace_bytes(max_len, nb, c) + 2 * align_up(B * C * sizeof(double), 256);
}

// x: the forward input, out: the forward output (sign pattern of the LeakyReLU), dout: its gradient.
// dx [n,c]; dadd [n,c] or NULL.
extern "C" int spr_instnorm_bwd(const float* x, const float* out, const float* dout, const int* cu, int n, int nb,
                                int max_len_host, int c, float eps, int norm, float slope, float* dx, float* dadd,
                                void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(n > 0 && nb >= 1 && c >= 4 && c % 4 == 0, "instnorm_bwd: need n>0 and c %% 4 == 0 (c=%d)", c);
  SPR_REQUIRE(x && out && dout && dx, "instnorm_bwd: null operand");
  float *mean = nullptr, *rstd = nullptr;
  double *m1 = nullptr, *m2 = nullptr;
  if (norm) {
    SPR_REQUIRE(max_len_host >= 1 && max_len_host <= n, "instnorm_bwd: bad max_len_host=%d", max_len_host);
    SPR_REQUIRE(ws_bytes >= spr_instnorm_bwd_workspace_bytes(max_len_host, nb, c), "instnorm_bwd: workspace too small");
    Workspace w(ws, ws_bytes);
    const int nsplit = in_nsplit(max_len_host);
    double* part = w.take<double>((size_t)nb * nsplit * 2 * c);
    mean = w.take<float>((size_t)nb * c);
    rstd = w.take<float>((size_t)nb * c);
    m1 = w.take<double>((size_t)nb * c);
    m2 = w.take<double>((size_t)nb * c);
    SPR_REQUIRE(m2 != nullptr, "instnorm_bwd: workspace carve failed");
    hipLaunchKernelGGL(k_in_stats, dim3(nb, nsplit, cdiv(c, 64)), dim3(256), 0, stream, x, cu, c, nsplit, part);
    hipLaunchKernelGGL(k_in_final, dim3(cdiv((long)nb * c, 256)), dim3(256), 0, stream, part, cu, nb, c, nsplit,
                       eps, mean, rstd);
    hipLaunchKernelGGL(k_in_bwd_stats, dim3(nb, nsplit, cdiv(c, 64)), dim3(256), 0, stream, x, out, dout, cu, c, nsplit, slope,
                       mean, rstd, part);
    hipLaunchKernelGGL(k_in_bwd_final, dim3(cdiv((long)nb * c, 256)), dim3(256), 0, stream, part, cu, nb, c, nsplit,
                       m1, m2);
    SPR_LAUNCH_CHECK();
  }
  if (c % 4 == 0 && (((uintptr_t)x | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dx | (uintptr_t)dadd) & 15) == 0)
    hipLaunchKernelGGL(k_in_bwd_apply4, dim3(cdiv((long)n * (c / 4), 256)), dim3(256), 0, stream, x, out, dout, cu, n, nb,
                       c, norm, slope, mean, rstd, m1, m2, dx, dadd);
  else
    hipLaunchKernelGGL(k_in_bwd_apply, dim3(cdiv((long)n * c, 256)), dim3(256), 0, stream, x, out, dout, cu, n, nb, c,
                       norm, slope, mean, rstd, m1, m2, dx, dadd);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" int spr_maxpool_gather(const float* x, int ns, int c, const int* idx, int nq,
                                  int idx_stride, int k, float* out, void* stream_) {
  return spr_maxpool_gather_r(x, ns, c, idx, nq, idx_stride, k, out, nullptr, 0, stream_);
}

// out_range: as in spr_instnorm_r.
extern "C" int spr_maxpool_gather_r(const float* x, int ns, int c, const int* idx, int nq, int idx_stride, int k,
                                    float* out, float* out_range, int out_range_n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(out_range == nullptr || (out_range_n >= 1 && (out_range_n & (out_range_n - 1)) == 0),
              "maxpool: out_range_n must be a power of two");
  SPR_REQUIRE(nq > 0 && ns > 0 && c % 4 == 0 && k >= 1 && k <= idx_stride, "maxpool: bad arguments");
  const long total = (long)nq * (c / 4);
  hipLaunchKernelGGL(k_maxpool, dim3(cdiv(total, 256)), dim3(256), 0, stream, x, ns, c, idx, nq,
                     idx_stride, k, out, out_range, out_range_n, (const int*)nullptr);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" int spr_maxpool_gather_o(const float* x, int ns, int c, const int* idx, int nq, int idx_stride, int k,
                                    const int* order, float* out, float* out_range, int out_range_n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(out_range == nullptr || (out_range_n >= 1 && (out_range_n & (out_range_n - 1)) == 0),
              "maxpool: out_range_n must be a power of two");
  SPR_REQUIRE(nq > 0 && ns > 0 && c % 4 == 0 && k >= 1 && k <= idx_stride, "maxpool: bad arguments");
  const long total = (long)nq * (c / 4);
  const long nblk = order != nullptr ? align_up((size_t)cdiv(total, 256), 8) : cdiv(total, 256);
  hipLaunchKernelGGL(k_maxpool, dim3((unsigned)nblk), dim3(256), 0, stream, x, ns, c, idx, nq, idx_stride, k, out,
                     out_range, out_range_n, order);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" int spr_gather_rows(const float* x, int n_src, int c, const int* idx, int n, float* out,
                               void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(n > 0 && c >= 1, "gather_rows: bad arguments");
  const long total = (long)n * c;
  hipLaunchKernelGGL(k_gather_rows, dim3(cdiv(total, 256)), dim3(256), 0, stream, x, n_src, c, idx,
                     n, out);
  SPR_LAUNCH_CHECK();
  return 0;
}
