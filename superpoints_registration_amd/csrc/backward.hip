// Backward kernels of the hot path that are not matrix products (those go
// through bgemm.hip) -- SURVEY 8f row 1.  In the reference every one of these
// gradients is produced by torch autograd from the forward expressions cited
// at each kernel; here they are explicit HIP kernels behind the C ABI and the
// Python side wires them into torch.autograd.Function objects (autograd.py).
//
// Scatter-type gradients (max-pool, KPConv dX) accumulate with float atomics;
// reductions over rows (LayerNorm dgamma / dbeta, bias gradients) are
// deterministic two-stage sums.
#include "spr_common.h"

namespace spr {
namespace {

// dy' = dy * act'(y): ReLU (y > 0), sigmoid (y (1 - y)); transformers.py:236, qk_regtr_full.py:249
__global__ void k_act_bwd(const float* __restrict__ y, const float* __restrict__ dy, int act, long n,
                          float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = y[i];
  out[i] = act == SPR_ACT_RELU ? (v > 0.f ? dy[i] : 0.f) : act == SPR_ACT_SIGMOID ? dy[i] * v * (1.f - v) : dy[i];
}

__global__ void k_act_bwd4(const float4* __restrict__ y, const float4* __restrict__ dy, int act, long n4,
                           float4* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 v = y[i], d = dy[i];
  float4 r = d;
  if (act == SPR_ACT_RELU)
    r = make_float4(v.x > 0.f ? d.x : 0.f, v.y > 0.f ? d.y : 0.f, v.z > 0.f ? d.z : 0.f, v.w > 0.f ? d.w : 0.f);
  else if (act == SPR_ACT_SIGMOID)
    r = make_float4(d.x * v.x * (1.f - v.x), d.y * v.y * (1.f - v.y), d.z * v.z * (1.f - v.z), d.w * v.w * (1.f - v.w));
  out[i] = r;
}

// column sums of x [m, n] over a fixed number of row chunks: parts [nchunk][n]
constexpr int kColChunks = 512;
__global__ __launch_bounds__(256) void k_colsum_parts(const float* __restrict__ x, long m, int n,
                                                      float* __restrict__ parts) {
  const int chunk = blockIdx.y;
  const long rows_per = (m + kColChunks - 1) / kColChunks;
  const long r0 = chunk * rows_per, r1 = r0 + rows_per < m ? r0 + rows_per : m;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  float s = 0.f;
  for (long r = r0; r < r1; ++r) s += x[r * n + j];
  parts[(long)chunk * n + j] = s;
}
// the same for n | 1024 (every bias of the model): the block sweeps the matrix as a flat float4 stream -- 4 KB per
// step whatever n is (the column-per-thread form above kept 32 of 256 threads busy for n = 32) -- each thread
// stays on its four columns, threads of equal columns are summed through LDS in a fixed order.
__global__ __launch_bounds__(256) void k_colsum_flat(const float* __restrict__ x, long m, int n,
                                                     float* __restrict__ parts) {
  __shared__ float4 sh[256];
  const long total4 = m * n / 4;
  const long steps = (total4 + 255) / 256;
  const long per = (steps + kColChunks - 1) / kColChunks;
  const long s0 = (long)blockIdx.x * per, s1 = s0 + per < steps ? s0 + per : steps;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  long st = s0;
  for (; st + 1 < s1; st += 2) {
    const long f0 = st * 256 + threadIdx.x, f1 = f0 + 256;
    const float4 u = x4[f0];                       // f0 < total4: st is not the last step
    const float4 v = f1 < total4 ? x4[f1] : make_float4(0.f, 0.f, 0.f, 0.f);
    a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
    b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
  }
  if (st < s1) {
    const long f0 = st * 256 + threadIdx.x;
    if (f0 < total4) {
      const float4 u = x4[f0];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
    }
  }
  sh[threadIdx.x] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  __syncthreads();
  const int ng = n / 4;
  if ((int)threadIdx.x < ng) {
    float4 t = sh[threadIdx.x];
    for (int k = threadIdx.x + ng; k < 256; k += ng) {
      const float4 v = sh[k];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    *reinterpret_cast<float4*>(parts + (long)blockIdx.x * n + 4 * threadIdx.x) = t;
  }
}

// ---- LayerNorm backward (transformers.py:121,:196-197: y = LN(x) gamma + beta; optional second
// output y + pos).  One wave per row; per-block partial sums of dgamma / dbeta over the block's rows.
constexpr int kLnBlocks = 256;
template <int MAXV>
__global__ __launch_bounds__(256) void k_layernorm_bwd(const float* __restrict__ x, int m, int c,
                                                       const float* __restrict__ gamma, float eps,
                                                       const float* __restrict__ dy_a,
                                                       const float* __restrict__ dy_b, float* __restrict__ dx,
                                                       float* __restrict__ dg_parts, float* __restrict__ db_parts) {
  __shared__ float sg[4][64 * MAXV], sb[4][64 * MAXV];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nv = c >> 6;
  float ag[MAXV], ab[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) ag[i] = ab[i] = 0.f;
  for (int row = blockIdx.x * 4 + wave; row < m; row += kLnBlocks * 4) {
    float v[MAXV], g[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const bool ok = i < nv;
      const size_t o = (size_t)row * c + i * 64 + lane;
      v[i] = ok ? x[o] : 0.f;
      g[i] = ok ? (dy_a ? dy_a[o] : 0.f) + (dy_b ? dy_b[o] : 0.f) : 0.f;
      s += v[i];
    }
    const float mean = wave_sum(s) / (float)c;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const float d = i < nv ? v[i] - mean : 0.f;
      ss += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)c + eps);
    float s1 = 0.f, s2 = 0.f;
    float xh[MAXV], dxh[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const bool ok = i < nv;
      xh[i] = ok ? (v[i] - mean) * rstd : 0.f;
      dxh[i] = ok ? g[i] * gamma[i * 64 + lane] : 0.f;
      s1 += dxh[i];
      s2 += dxh[i] * xh[i];
      ag[i] += g[i] * xh[i];
      ab[i] += g[i];
    }
    s1 = wave_sum(s1) / (float)c;
    s2 = wave_sum(s2) / (float)c;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (i < nv) dx[(size_t)row * c + i * 64 + lane] = rstd * (dxh[i] - s1 - xh[i] * s2);
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    sg[wave][i * 64 + lane] = ag[i];
    sb[wave][i * 64 + lane] = ab[i];
  }
  __syncthreads();
  for (int ch = threadIdx.x; ch < c; ch += 256) {
    dg_parts[(size_t)blockIdx.x * c + ch] = sg[0][ch] + sg[1][ch] + sg[2][ch] + sg[3][ch];
    db_parts[(size_t)blockIdx.x * c + ch] = sb[0][ch] + sb[1][ch] + sb[2][ch] + sb[3][ch];
  }
}

// c == 256 (every LayerNorm of the model): one float4 per lane and row, FOUR rows per wave and iteration with all
// their loads issued up front -- the generic kernel above walks one row at a time behind four dependent wave
// reductions and reached 0.9 TB/s (211 us for 61 k tokens).
__global__ __launch_bounds__(256) void k_layernorm_bwd256(const float* __restrict__ x, int m,
                                                          const float* __restrict__ gamma, float eps,
                                                          const float* __restrict__ dy_a,
                                                          const float* __restrict__ dy_b, float* __restrict__ dx,
                                                          float* __restrict__ dg_parts, float* __restrict__ db_parts) {
  __shared__ float4 sg[4][64], sb[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float4 gm = reinterpret_cast<const float4*>(gamma)[lane];
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* a4 = reinterpret_cast<const float4*>(dy_a);
  const float4* b4 = reinterpret_cast<const float4*>(dy_b);
  float4* o4 = reinterpret_cast<float4*>(dx);
  float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = ag;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int base = (blockIdx.x * 4 + wave) * 4; base < m; base += kLnBlocks * 16) {
    float4 v[4], g[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = base + j < m;
      const size_t o = (size_t)(base + j) * 64 + lane;
      v[j] = ok ? x4[o] : z;
      const float4 ga = (ok && dy_a) ? a4[o] : z, gb = (ok && dy_b) ? b4[o] : z;
      g[j] = make_float4(ga.x + gb.x, ga.y + gb.y, ga.z + gb.z, ga.w + gb.w);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float mean = wave_sum((v[j].x + v[j].y) + (v[j].z + v[j].w)) * (1.0f / 256.0f);
      const float4 d = make_float4(v[j].x - mean, v[j].y - mean, v[j].z - mean, v[j].w - mean);
      const float rstd = 1.0f / sqrtf(wave_sum((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w)) * (1.0f / 256.0f) + eps);
      const float4 xh = make_float4(d.x * rstd, d.y * rstd, d.z * rstd, d.w * rstd);
      const float4 dh = make_float4(g[j].x * gm.x, g[j].y * gm.y, g[j].z * gm.z, g[j].w * gm.w);
      const float s1 = wave_sum((dh.x + dh.y) + (dh.z + dh.w)) * (1.0f / 256.0f);
      const float s2 = wave_sum((dh.x * xh.x + dh.y * xh.y) + (dh.z * xh.z + dh.w * xh.w)) * (1.0f / 256.0f);
      ag.x += g[j].x * xh.x; ag.y += g[j].y * xh.y; ag.z += g[j].z * xh.z; ag.w += g[j].w * xh.w;
      ab.x += g[j].x; ab.y += g[j].y; ab.z += g[j].z; ab.w += g[j].w;
      if (base + j < m)
        o4[(size_t)(base + j) * 64 + lane] = make_float4(rstd * (dh.x - s1 - xh.x * s2), rstd * (dh.y - s1 - xh.y * s2),
                                                         rstd * (dh.z - s1 - xh.z * s2), rstd * (dh.w - s1 - xh.w * s2));
    }
  }
  sg[wave][lane] = ag;
  sb[wave][lane] = ab;
  __syncthreads();
  if (threadIdx.x < 64) {
    float4 a = sg[0][lane], b = sb[0][lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float4 p = sg[w][lane], q = sb[w][lane];
      a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
      b.x += q.x; b.y += q.y; b.z += q.z; b.w += q.w;
    }
    reinterpret_cast<float4*>(dg_parts + (size_t)blockIdx.x * 256)[lane] = a;
    reinterpret_cast<float4*>(db_parts + (size_t)blockIdx.x * 256)[lane] = b;
  }
}

// ---- order-independent scatter-adds ----------------------------------------------------------------
// Several queries send a contribution to the same (row, channel) of a gradient (max-pool, row gather, the
// KPConv's neighbour gather).  Float atomics made those sums depend on the arrival order (run-to-run
// differences at rounding level).  Here a contribution is converted to 64-bit FIXED POINT -- scaled by a
// power of two 2^fx derived from the measured maximum of the incoming gradient, so that the largest
// contribution sits near 2^40 -- and added with integer atomics: integer addition is associative, the
// result does not depend on the order.  Resolution 2^-40 of the largest contribution (finer than the
// 2^-24 of a float running sum); 2^22 maximal contributions fit before overflow.  k_fx_to_float converts.
__device__ __forceinline__ int fx_exp(const float* __restrict__ parts, float* sh, float factor) {
  return pow2_exp_for(block_absmax(parts, sh) * factor) + 25;    // max |v| 2^fx in [2^39, 2^40)
}
__device__ __forceinline__ void fx_add(unsigned long long* acc, float v, int fx) {
  // v * 2^fx exactly (two exact power-of-two factors keep the intermediate in float range)
  const float sc = (v * pow2f(fx / 2)) * pow2f(fx - fx / 2);
  atomicAdd(acc, (unsigned long long)__float2ll_rn(sc));
}
__global__ void k_fx_to_float(const unsigned long long* __restrict__ acc, long n, const float* __restrict__ parts,
                              float factor, float* __restrict__ out) {
  __shared__ float sh[17];
  const float amax = block_absmax(parts, sh);
  const int fx = pow2_exp_for(amax * factor) + 25;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (!(amax < 3.0e38f)) {   // the incoming gradient holds an inf or a NaN (k_absmax maps NaN to inf): a float
    out[i] = __builtin_nanf("");   // atomicAdd would have propagated it; the fixed-point sums cannot, so the
    return;                        // whole gradient is marked (a finite-gradient check downstream still fires)
  }
  const double inv = (double)pow2f(-(fx / 2)) * (double)pow2f(-(fx - fx / 2));
  out[i] = (float)((double)(long long)acc[i] * inv);
}

// ---- max-pool backward (kpconv_blocks.py:127-143): dy goes to the arg-max source row of every
// (query, channel); the shadow row (index ns) receives nothing.  First maximum wins, like
// torch.max's index on ties.
__global__ __launch_bounds__(256) void k_maxpool_bwd(const float* __restrict__ x, int ns, int c, const int* __restrict__ idx,
                                                     int nq, int idx_stride, int k, const float* __restrict__ dy,
                                                     const float* __restrict__ parts, unsigned long long* __restrict__ acc) {
  __shared__ float sh[17];
  const int fx = fx_exp(parts, sh, 1.0f);
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)nq * c) return;
  const int row = (int)(gid / c), ch = (int)(gid % c);
  const int* ir = idx + (size_t)row * idx_stride;
  float best = -3.0e38f;
  int bi = -1;
  for (int j = 0; j < k; ++j) {
    const int id = ir[j];
    const bool ok = id >= 0 && id < ns;
    const float v = ok ? x[(size_t)id * c + ch] : 0.f;
    if (v > best) {
      best = v;
      bi = ok ? id : -1;
    }
  }
  if (bi >= 0) fx_add(acc + (size_t)bi * c + ch, dy[gid], fx);
}

// the same with four channels per thread: the neighbour ids are read once per float4 and the gathered rows arrive as
// 16-byte loads, four in flight (the scalar form above walked k dependent 4-byte loads per element)
__global__ __launch_bounds__(256) void k_maxpool_bwd4(const float* __restrict__ x, int ns, int c, const int* __restrict__ idx,
                                                      int nq, int idx_stride, int k, const float* __restrict__ dy,
                                                      const float* __restrict__ parts, unsigned long long* __restrict__ acc) {
  __shared__ float sh[17];
  const int fx = fx_exp(parts, sh, 1.0f);
  const int c4 = c >> 2;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)nq * c4) return;
  const int row = (int)(gid / c4), q = (int)(gid % c4);
  const int* ir = idx + (size_t)row * idx_stride;
  float best[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
  int bi[4] = {-1, -1, -1, -1};
  auto take = [&](int id, const float4& v) {
    const bool ok = id >= 0 && id < ns;
    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (e[u] > best[u]) {         // first maximum wins (torch.max's index on ties), shadow rows read as 0
        best[u] = e[u];
        bi[u] = ok ? id : -1;
      }
  };
  int j = 0;
  for (; j + 4 <= k; j += 4) {
    int id[4];
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) id[u] = ir[j + u];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (id[u] >= 0 && id[u] < ns) v[u] = reinterpret_cast<const float4*>(x + (size_t)id[u] * c)[q];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) take(id[u], v[u]);
  }
  for (; j < k; ++j) {
    const int id = ir[j];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (id >= 0 && id < ns) v = reinterpret_cast<const float4*>(x + (size_t)id * c)[q];
    take(id, v);
  }
  const float4 g = reinterpret_cast<const float4*>(dy)[gid];
  const float ge[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (bi[u] >= 0) fx_add(acc + (size_t)bi[u] * c + 4 * q + u, ge[u], fx);
}

// rows gathered forward (spr_gather_rows) -> scatter-add backward
__global__ __launch_bounds__(256) void k_scatter_rows_add(const float* __restrict__ dy, const int* __restrict__ idx, int n,
                                                          int c, int n_src, const float* __restrict__ parts,
                                                          unsigned long long* __restrict__ acc) {
  __shared__ float sh[17];
  const int fx = fx_exp(parts, sh, 1.0f);
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)n * c) return;
  const int row = (int)(gid / c), ch = (int)(gid % c);
  const int id = idx[row];
  if (id >= 0 && id < n_src) fx_add(acc + (size_t)id * c + ch, dy[gid], fx);
}

// ---- KPConv backward helpers (kpconv_blocks.py:309-412) --------------------------------------
// One wave per query.
//   WF mode : wf[n, p * cin + c] = sum_k infl[p][k] x[idx[n,k], c]     (recomputed forward, un-normalised)
//             cnt[n] = max(1, #{k : sum_c x[idx[n,k], :] > 0})
//   DX mode : dx[idx[n,k], c] += sum_p infl[p][k] dwf[n, p * cin + c]
constexpr int kKPmax = 16;
// The influence weights of up to 64 neighbours are computed side by side (lane =
// neighbour: id, support point, 15 distances) and parked in LDS, then the lanes walk the channels with one
// broadcast read of a neighbour's 16 weights per 15 FMAs -- the per-neighbour dependent chain (id -> point ->
// sqrt -> LDS -> barrier) of the round-3 form is gone: the 16-pair training step 154 -> 145 ms.  HALF (cin == 32): the
// two half-waves take the even / odd neighbours and are summed at the end.  cin == 1: everything stays in the
// neighbour lanes (15 wave sums).
template <bool DX, bool HALF>
__global__ __launch_bounds__(256) void k_kpconv_aux2(const float* __restrict__ q_xyz, int nq,
                                                     const float* __restrict__ s_xyz, int ns,
                                                     const int* __restrict__ nbr, int nbr_stride, int kmax,
                                                     const float* __restrict__ x, int cin,
                                                     const float* __restrict__ kpts, int n_kp, float inv_extent,
                                                     const float* __restrict__ dwf, float* __restrict__ wf,
                                                     float* __restrict__ cnt_out, const float* __restrict__ parts,
                                                     unsigned long long* __restrict__ dx_acc) {
  __shared__ __align__(16) float infl_s[4][64][kKPmax];
  __shared__ int id_s[4][64];
  int fx = 0;
  if (DX) {   // |sum_p infl[p] dwf[p]| <= n_kp max |dwf|
    __shared__ float sh[17];
    fx = fx_exp(parts, sh, (float)n_kp);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + wave;
  if (n >= nq) return;   // whole wave
  const float qx = q_xyz[3 * (size_t)n], qy = q_xyz[3 * (size_t)n + 1], qz = q_xyz[3 * (size_t)n + 2];
  const int half = HALF ? lane >> 5 : 0;
  const int cl = HALF ? lane & 31 : lane;
  const int nchunk = HALF ? 1 : (cin + 63) / 64;
  int cnt = 0;
  float acc1[kKPmax];          // cin == 1: per-neighbour-lane partial sums
#pragma unroll
  for (int p = 0; p < kKPmax; ++p) acc1[p] = 0.f;
  for (int kb = 0; kb < kmax; kb += 64) {
    // ---- phase A: lane = neighbour ----
    const int kk = kb + lane;
    int id = kk < kmax ? nbr[(size_t)n * nbr_stride + kk] : -1;
    const bool ok = id >= 0 && id < ns;
    if (!ok) id = -1;
    float w[kKPmax];
    {
      float px = 0.f, py = 0.f, pz = 0.f;
      if (ok) {
        px = s_xyz[3 * (size_t)id] - qx;
        py = s_xyz[3 * (size_t)id + 1] - qy;
        pz = s_xyz[3 * (size_t)id + 2] - qz;
      }
#pragma unroll
      for (int p = 0; p < kKPmax; ++p) {
        float v = 0.f;
        if (p < n_kp) {
          const float dx_ = px - kpts[3 * p], dy_ = py - kpts[3 * p + 1], dz_ = pz - kpts[3 * p + 2];
          v = ok ? fmaxf(0.f, 1.f - sqrtf(dx_ * dx_ + dy_ * dy_ + dz_ * dz_) * inv_extent) : 0.f;
        }
        w[p] = v;
      }
    }
    if (!DX) {   // neighbour count of the reference: rows whose feature sum is > 0
      float s = 0.f;
      if (ok) {
        const float* xr = x + (size_t)id * cin;
        if ((cin & 3) == 0) {
          for (int c = 0; c < cin; c += 4) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            s += (v.x + v.y) + (v.z + v.w);
          }
        } else {
          for (int c = 0; c < cin; ++c) s += xr[c];
        }
      }
      cnt += __builtin_popcountll(__builtin_amdgcn_ballot_w64(ok && s > 0.f));
    }
    if (cin == 1) {
      if (!DX) {
        const float xv = ok ? x[id] : 0.f;
#pragma unroll
        for (int p = 0; p < kKPmax; ++p) acc1[p] += w[p] * xv;
        continue;
      }
    }
    __builtin_amdgcn_wave_barrier();      // the previous chunk's reads are done (one wave: program order)
#pragma unroll
    for (int p4 = 0; p4 < kKPmax / 4; ++p4)
      *reinterpret_cast<float4*>(&infl_s[wave][lane][4 * p4]) = make_float4(w[4 * p4], w[4 * p4 + 1], w[4 * p4 + 2], w[4 * p4 + 3]);
    id_s[wave][lane] = id;
    __builtin_amdgcn_wave_barrier();
    const int nk = min(64, kmax - kb);
    // ---- phase B: lane = channel ----
    for (int c0 = 0; c0 < nchunk; ++c0) {
      const int c = c0 * 64 + cl;
      const bool cok = c < cin;
      float acc[kKPmax];
#pragma unroll
      for (int p = 0; p < kKPmax; ++p) acc[p] = 0.f;
      if (DX) {
#pragma unroll
        for (int p = 0; p < kKPmax; ++p) acc[p] = (cok && p < n_kp) ? dwf[((size_t)n * n_kp + p) * cin + c] : 0.f;
      }
      const int step = HALF ? 2 : 1;
#pragma unroll 2
      for (int k0 = 0; k0 < nk; k0 += step) {
        const int k = k0 + half;
        const int idk = k < nk ? id_s[wave][k] : -1;
        if (!HALF && idk < 0) continue;     // wave uniform
        const bool use = idk >= 0 && cok;
        float iw[kKPmax];
#pragma unroll
        for (int p4 = 0; p4 < kKPmax / 4; ++p4) {
          const float4 v = *reinterpret_cast<const float4*>(&infl_s[wave][k & 63][4 * p4]);
          iw[4 * p4] = v.x;
          iw[4 * p4 + 1] = v.y;
          iw[4 * p4 + 2] = v.z;
          iw[4 * p4 + 3] = v.w;
        }
        if (DX) {
          float v = 0.f;
#pragma unroll
          for (int p = 0; p < kKPmax; ++p)
            if (p < n_kp) v += iw[p] * acc[p];
          if (use && v != 0.f) fx_add(dx_acc + (size_t)idk * cin + c, v, fx);
        } else {
          const float xv = use ? x[(size_t)idk * cin + c] : 0.f;
#pragma unroll
          for (int p = 0; p < kKPmax; ++p)
            if (p < n_kp) acc[p] += iw[p] * xv;
        }
      }
      if (!DX) {
        if (HALF) {
#pragma unroll
          for (int p = 0; p < kKPmax; ++p) acc[p] += __shfl_xor(acc[p], 32, 64);
        }
        // (kmax <= 64: one chunk, plain store; more: later chunks add to what the first wrote)
        if (cok && half == 0) {
#pragma unroll
          for (int p = 0; p < kKPmax; ++p)
            if (p < n_kp) {
              float* dst = wf + ((size_t)n * n_kp + p) * cin + c;
              *dst = kb == 0 ? acc[p] : *dst + acc[p];
            }
        }
      }
    }
  }
  if (!DX && cin == 1) {
#pragma unroll
    for (int p = 0; p < kKPmax; ++p) {
      const float t = wave_sum(acc1[p]);
      if (p < n_kp && lane == 0) wf[(size_t)n * n_kp + p] = t;
    }
  }
  if (!DX && lane == 0) cnt_out[n] = (float)(cnt > 1 ? cnt : 1);
}

// ---- softmax over the rows of per-batch matrices (attention backward) --------------------------
struct MatDesc {
  long long a_off, b_off, c_off;
  int m, n, k, pad;
};
// in place: mat_b[i, :] = softmax(mat_b[i, :])          (one wave per row)
__global__ void k_softmax_rows(float* __restrict__ mat, const MatDesc* __restrict__ desc) {
  const MatDesc d = desc[blockIdx.y];
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (row >= d.m) return;
  float* r = mat + d.c_off + (size_t)row * d.n;
  float mx = -INFINITY;
  for (int j = lane; j < d.n; j += 64) mx = fmaxf(mx, r[j]);
  mx = wave_max(mx);
  float s = 0.f;
  for (int j = lane; j < d.n; j += 64) s += expf(r[j] - mx);
  s = wave_sum(s);
  const float inv = 1.f / s;
  for (int j = lane; j < d.n; j += 64) r[j] = expf(r[j] - mx) * inv;
}
// in place on dp: ds = p * (dp - sum_j p dp)
__global__ void k_softmax_bwd_rows(const float* __restrict__ p, float* __restrict__ dp,
                                   const MatDesc* __restrict__ desc) {
  const MatDesc d = desc[blockIdx.y];
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (row >= d.m) return;
  const float* pr = p + d.c_off + (size_t)row * d.n;
  float* dr = dp + d.c_off + (size_t)row * d.n;
  float s = 0.f;
  for (int j = lane; j < d.n; j += 64) s += pr[j] * dr[j];
  s = wave_sum(s);
  for (int j = lane; j < d.n; j += 64) dr[j] = pr[j] * (dr[j] - s);
}

}  // namespace
}  // namespace spr

using namespace spr;

extern "C" int spr_act_bwd(const float* y, const float* dy, int act, long n, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(y && dy && out && n >= 1 && act >= 0 && act <= 2, "act_bwd: bad arguments");
  if (n % 4 == 0 && (((uintptr_t)y | (uintptr_t)dy | (uintptr_t)out) & 15) == 0)
    hipLaunchKernelGGL(k_act_bwd4, dim3(cdiv(n / 4, 256)), dim3(256), 0, stream, (const float4*)y, (const float4*)dy, act,
                       n / 4, (float4*)out);
  else
    hipLaunchKernelGGL(k_act_bwd, dim3(cdiv(n, 256)), dim3(256), 0, stream, y, dy, act, n, out);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t spr_colsum_workspace_bytes(int n) { return (size_t)kColChunks * (n > 0 ? n : 1) * sizeof(float); }

extern "C" int spr_colsum(const float* x, long m, int n, float* out, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(x && out && m >= 1 && n >= 1, "colsum: bad arguments");
  SPR_REQUIRE(ws && ws_bytes >= spr_colsum_workspace_bytes(n), "colsum: workspace too small");
  float* parts = (float*)ws;
  if (n >= 4 && 1024 % n == 0 && ((uintptr_t)x & 15) == 0)
    hipLaunchKernelGGL(k_colsum_flat, dim3(kColChunks), dim3(256), 0, stream, x, (long)m, n, parts);
  else
    hipLaunchKernelGGL(k_colsum_parts, dim3(cdiv(n, 256), kColChunks), dim3(256), 0, stream, x, m, n, parts);
  SPR_LAUNCH_CHECK();
  return spr_reduce_parts(parts, kColChunks, n, 1.0f, out, 0, stream_);
}

extern "C" size_t spr_layernorm_bwd_workspace_bytes(int c) {
  return 2 * (size_t)kLnBlocks * (c > 0 ? c : 1) * sizeof(float);
}

extern "C" int spr_layernorm_bwd(const float* x, int m, int c, const float* gamma, float eps, const float* dy_norm,
                                 const float* dy_pos, float* dx, float* dgamma, float* dbeta, void* ws,
                                 size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(m > 0 && c % 64 == 0 && c <= 1024, "layernorm_bwd: c must be a multiple of 64 and <= 1024 (c=%d)", c);
  SPR_REQUIRE(x && gamma && dx && dgamma && dbeta && (dy_norm || dy_pos), "layernorm_bwd: null operand");
  SPR_REQUIRE(ws && ws_bytes >= spr_layernorm_bwd_workspace_bytes(c), "layernorm_bwd: workspace too small");
  float* dg = (float*)ws;
  float* db = dg + (size_t)kLnBlocks * c;
  const bool al16 = (((uintptr_t)x | (uintptr_t)dx | (uintptr_t)gamma | (uintptr_t)dy_norm | (uintptr_t)dy_pos) & 15) == 0;
  if (c == 256 && al16)
    hipLaunchKernelGGL(k_layernorm_bwd256, dim3(kLnBlocks), dim3(256), 0, stream, x, m, gamma, eps, dy_norm, dy_pos, dx,
                       dg, db);
  else if (c <= 256)
    hipLaunchKernelGGL(k_layernorm_bwd<4>, dim3(kLnBlocks), dim3(256), 0, stream, x, m, c, gamma, eps, dy_norm,
                       dy_pos, dx, dg, db);
  else
    hipLaunchKernelGGL(k_layernorm_bwd<16>, dim3(kLnBlocks), dim3(256), 0, stream, x, m, c, gamma, eps, dy_norm,
                       dy_pos, dx, dg, db);
  SPR_LAUNCH_CHECK();
  if (int rc = spr_reduce_parts(dg, kLnBlocks, c, 1.0f, dgamma, 0, stream_)) return rc;
  return spr_reduce_parts(db, kLnBlocks, c, 1.0f, dbeta, 0, stream_);
}

// scratch of the order-independent scatter-adds: the fixed-point accumulators + the range partials of the
// incoming gradient
extern "C" size_t spr_scatter_workspace_bytes(long rows, int c) {
  return align_up((size_t)(rows > 0 ? rows : 1) * (size_t)(c > 0 ? c : 1) * sizeof(unsigned long long), 256) +
         align_up(kAmaxParts * sizeof(float), 256);
}
namespace {
struct FxScratch {
  unsigned long long* acc;
  float* parts;
};
int fx_scratch(void* ws, size_t ws_bytes, long rows, int c, hipStream_t stream, FxScratch* out) {
  SPR_REQUIRE(ws != nullptr && ws_bytes >= spr_scatter_workspace_bytes(rows, c), "scatter: workspace too small");
  Workspace w(ws, ws_bytes);
  out->acc = w.take<unsigned long long>((size_t)rows * c);
  out->parts = w.take<float>(kAmaxParts);
  SPR_REQUIRE(out->parts != nullptr, "scatter: workspace carve failed");
  SPR_HIP_CHECK(hipMemsetAsync(out->acc, 0, (size_t)rows * c * sizeof(unsigned long long), stream));
  return 0;
}
}  // namespace

// dx [ns, c] is fully written (ws: spr_scatter_workspace_bytes(ns, c))
extern "C" int spr_maxpool_bwd(const float* x, int ns, int c, const int* idx, int nq, int idx_stride, int k,
                               const float* dy, float* dx, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(nq > 0 && ns > 0 && c >= 1 && k >= 1 && k <= idx_stride, "maxpool_bwd: bad arguments");
  FxScratch f;
  if (int rc = fx_scratch(ws, ws_bytes, ns, c, stream, &f)) return rc;
  if (int rc = launch_absmax(dy, nq, c, c, f.parts, stream)) return rc;
  if (c % 4 == 0 && (((uintptr_t)x | (uintptr_t)dy) & 15) == 0)
    hipLaunchKernelGGL(k_maxpool_bwd4, dim3(cdiv((long)nq * (c / 4), 256)), dim3(256), 0, stream, x, ns, c, idx, nq,
                       idx_stride, k, dy, f.parts, f.acc);
  else
    hipLaunchKernelGGL(k_maxpool_bwd, dim3(cdiv((long)nq * c, 256)), dim3(256), 0, stream, x, ns, c, idx, nq,
                       idx_stride, k, dy, f.parts, f.acc);
  hipLaunchKernelGGL(k_fx_to_float, dim3(cdiv((long)ns * c, 256)), dim3(256), 0, stream, f.acc, (long)ns * c, f.parts,
                     1.0f, dx);
  SPR_LAUNCH_CHECK();
  return 0;
}

// dx [n_src, c] is fully written (ws: spr_scatter_workspace_bytes(n_src, c))
extern "C" int spr_scatter_rows_add(const float* dy, const int* idx, int n, int c, int n_src, float* dx, void* ws,
                                    size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(n > 0 && c >= 1 && n_src > 0, "scatter_rows_add: bad arguments");
  FxScratch f;
  if (int rc = fx_scratch(ws, ws_bytes, n_src, c, stream, &f)) return rc;
  if (int rc = launch_absmax(dy, n, c, c, f.parts, stream)) return rc;
  hipLaunchKernelGGL(k_scatter_rows_add, dim3(cdiv((long)n * c, 256)), dim3(256), 0, stream, dy, idx, n, c, n_src,
                     f.parts, f.acc);
  hipLaunchKernelGGL(k_fx_to_float, dim3(cdiv((long)n_src * c, 256)), dim3(256), 0, stream, f.acc, (long)n_src * c,
                     f.parts, 1.0f, dx);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" int spr_kpconv_weighted_features(const float* q_xyz, int nq, const float* s_xyz, int ns, const int* nbr,
                                            int nbr_stride, int kmax, const float* x, int cin,
                                            const float* kernel_points, int n_kp, float kp_extent, float* wf,
                                            float* cnt, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(nq > 0 && ns > 0 && cin >= 1 && n_kp >= 1 && n_kp <= kKPmax && kp_extent > 0.f && kmax >= 1 &&
                  kmax <= nbr_stride, "kpconv_weighted_features: bad arguments");
  if (cin == 32)
    hipLaunchKernelGGL((k_kpconv_aux2<false, true>), dim3(cdiv(nq, 4)), dim3(256), 0, stream, q_xyz, nq, s_xyz, ns, nbr,
                       nbr_stride, kmax, x, cin, kernel_points, n_kp, 1.0f / kp_extent, (const float*)nullptr, wf, cnt,
                       (const float*)nullptr, (unsigned long long*)nullptr);
  else
    hipLaunchKernelGGL((k_kpconv_aux2<false, false>), dim3(cdiv(nq, 4)), dim3(256), 0, stream, q_xyz, nq, s_xyz, ns, nbr,
                       nbr_stride, kmax, x, cin, kernel_points, n_kp, 1.0f / kp_extent, (const float*)nullptr, wf, cnt,
                       (const float*)nullptr, (unsigned long long*)nullptr);
  SPR_LAUNCH_CHECK();
  return 0;
}

namespace {
// an externally published range (n partials) as the kAmaxParts partials the fixed-point kernels read
__global__ __launch_bounds__(256) void k_range_to_parts(const float* __restrict__ range, int n, float* __restrict__ parts) {
  __shared__ float sh[17];
  const float m = block_absmax(range, sh, n);
  for (int i = threadIdx.x; i < kAmaxParts; i += 256) parts[i] = i == 0 ? m : 0.f;
}
}  // namespace

static int kpconv_bwd_dx_impl(const float* q_xyz, int nq, const float* s_xyz, int ns, const int* nbr,
                              int nbr_stride, int kmax, int cin, const float* kernel_points, int n_kp,
                              float kp_extent, const float* dwf, const float* dwf_range, int dwf_range_n, float* dx,
                              void* ws, size_t ws_bytes, void* stream_);

// dx [ns, cin] is fully written (ws: spr_scatter_workspace_bytes(ns, cin)); order-independent sums
extern "C" int spr_kpconv_bwd_dx(const float* q_xyz, int nq, const float* s_xyz, int ns, const int* nbr,
                                 int nbr_stride, int kmax, int cin, const float* kernel_points, int n_kp,
                                 float kp_extent, const float* dwf, float* dx, void* ws, size_t ws_bytes,
                                 void* stream_) {
  return kpconv_bwd_dx_impl(q_xyz, nq, s_xyz, ns, nbr, nbr_stride, kmax, cin, kernel_points, n_kp, kp_extent, dwf,
                            nullptr, 0, dx, ws, ws_bytes, stream_);
}
// dwf_range: max |dwf| partials published by the product that wrote dwf (spr_linear_r's out_range): the 1 GB
// tensor is then not scanned again for the fixed-point scale
extern "C" int spr_kpconv_bwd_dx_r(const float* q_xyz, int nq, const float* s_xyz, int ns, const int* nbr,
                                   int nbr_stride, int kmax, int cin, const float* kernel_points, int n_kp,
                                   float kp_extent, const float* dwf, const float* dwf_range, int dwf_range_n, float* dx,
                                   void* ws, size_t ws_bytes, void* stream_) {
  SPR_REQUIRE(dwf_range == nullptr || dwf_range_n >= 1, "kpconv_bwd_dx: a range needs a count");
  return kpconv_bwd_dx_impl(q_xyz, nq, s_xyz, ns, nbr, nbr_stride, kmax, cin, kernel_points, n_kp, kp_extent, dwf,
                            dwf_range, dwf_range_n, dx, ws, ws_bytes, stream_);
}

static int kpconv_bwd_dx_impl(const float* q_xyz, int nq, const float* s_xyz, int ns, const int* nbr,
                              int nbr_stride, int kmax, int cin, const float* kernel_points, int n_kp,
                              float kp_extent, const float* dwf, const float* dwf_range, int dwf_range_n, float* dx,
                              void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(nq > 0 && ns > 0 && cin >= 1 && n_kp >= 1 && n_kp <= kKPmax && kp_extent > 0.f && kmax >= 1 &&
                  kmax <= nbr_stride, "kpconv_bwd_dx: bad arguments");
  FxScratch f;
  if (int rc = fx_scratch(ws, ws_bytes, ns, cin, stream, &f)) return rc;
  if (dwf_range != nullptr) {
    hipLaunchKernelGGL(k_range_to_parts, dim3(1), dim3(256), 0, stream, dwf_range, dwf_range_n, f.parts);
  } else if (int rc = launch_absmax(dwf, nq, n_kp * cin, n_kp * cin, f.parts, stream)) {
    return rc;
  }
  if (cin == 32)
    hipLaunchKernelGGL((k_kpconv_aux2<true, true>), dim3(cdiv(nq, 4)), dim3(256), 0, stream, q_xyz, nq, s_xyz, ns, nbr,
                       nbr_stride, kmax, (const float*)nullptr, cin, kernel_points, n_kp, 1.0f / kp_extent, dwf,
                       (float*)nullptr, (float*)nullptr, f.parts, f.acc);
  else
    hipLaunchKernelGGL((k_kpconv_aux2<true, false>), dim3(cdiv(nq, 4)), dim3(256), 0, stream, q_xyz, nq, s_xyz, ns, nbr,
                       nbr_stride, kmax, (const float*)nullptr, cin, kernel_points, n_kp, 1.0f / kp_extent, dwf,
                       (float*)nullptr, (float*)nullptr, f.parts, f.acc);
  hipLaunchKernelGGL(k_fx_to_float, dim3(cdiv((long)ns * cin, 256)), dim3(256), 0, stream, f.acc, (long)ns * cin,
                     f.parts, (float)n_kp, dx);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" int spr_softmax_rows(float* mat, const void* desc_dev, int nbatch, int max_m, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(mat && desc_dev && nbatch >= 1 && nbatch <= 65535 && max_m >= 1, "softmax_rows: bad arguments");
  hipLaunchKernelGGL(k_softmax_rows, dim3(cdiv((long)max_m * 64, 256), nbatch), dim3(256), 0, stream, mat,
                     (const MatDesc*)desc_dev);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" int spr_softmax_bwd_rows(const float* p, float* dp, const void* desc_dev, int nbatch, int max_m,
                                    void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(p && dp && desc_dev && nbatch >= 1 && nbatch <= 65535 && max_m >= 1, "softmax_bwd_rows: bad arguments");
  hipLaunchKernelGGL(k_softmax_bwd_rows, dim3(cdiv((long)max_m * 64, 256), nbatch), dim3(256), 0, stream, p, dp,
                     (const MatDesc*)desc_dev);
  SPR_LAUNCH_CHECK();
  return 0;
}
