// a5 -- the tail of a ResNet bottleneck block as two row-streaming passes.
//
// Behaviour contract (ResnetBottleneckBlock.forward, kpconv_blocks.py:723-741):
//     x        = unary2(x)                       Linear(no bias) -> InstanceNorm per cloud   (:556-561, :497-525)
//     shortcut = unary_shortcut(shortcut)        same, or the identity
//     return leaky_relu(x + shortcut)            LeakyReLU(0.1)
//
// As separate operators that is, for the 128-channel block on 0.5 M points, 2.9 GB of HBM traffic: each
// projection writes its [N, Cout] output, the statistics pass reads it, the normalising pass reads it
// again and writes it.  The projections themselves are tiny (K = 32 .. 256), so here they are simply
// computed twice and their un-normalised outputs never exist in memory:
//   pass 1  (k_block_tail<.., 1>)  y_a = x_a W_a^T, y_b = x_b W_b^T per 64-row tile; only the per-tile
//           column sums and sums of squares (float64) leave the kernel
//   final   (k_tail_final)        per (cloud, channel): fixed-order sum of the tile partials -> mean, rstd
//   pass 2  (k_block_tail<.., 2>)  the same products again, then
//           out = lrelu((y_a - mean_a) rstd_a + ((y_b - mean_b) rstd_b  |  add)), one [N, Cout] write
// 2.9 GB -> 0.67 GB for that block.  The arithmetic per element is that of the separate operators (split-fp16
// products with per-tensor power-of-two scales, float64 statistics, the same float32 expression for the
// normalisation), and -- tiles being aligned to each cloud's first row -- a cloud's result does not depend
// on its batch mates.
//
// Kernel shape: persistent 8-wave workgroups; a wave owns 16 or 32 output columns and keeps ITS slice of
// both weight matrices in registers for the whole launch (split fp16 hi/lo, fragments of
// v_mfma_f32_16x16x32_f16); the x tile is split once while it is staged into LDS in fragment order
// (conflict-free ds_read_b128 for every wave, XOR-swizzled so that the staging ds_write_b64 spread over
// the banks); the next tile's rows are in flight in registers under the current tile's MFMAs; one
// barrier per tile.  Pass 1 uses the C layout (row = 4 kg + r, col = lane) whose column sums are in-lane;
// pass 2 swaps the MFMA operands (C^T: four consecutive output columns per lane) so that the result
// leaves as 16-byte stores.
#include "spr_common.h"

namespace spr {
int gemm_mode();
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kTailThreads = 512;
constexpr int kTailWaves = 8;

struct TailArgs {
  const float* xa;
  const float* xb;
  const float* wa;
  const float* wb;
  const float* add;
  const int* cu;
  const int* tile_cu;    // [nb + 1] tiles before each cloud
  const int4* tile_desc; // [tile] {first row, valid rows, cloud, 0}
  int nb, n, n_total;
  const float* xa_parts;
  const float* xb_parts;
  const float* wa_parts;
  const float* wb_parts;
  int xa_np, xb_np, wa_np, wb_np;
  double* part;   // [tile][branch][2][n_total]
  float* stats;   // [cloud][branch][2][n_total]  (mean, rstd)
  float slope;
  float* out;
  float* out_range;
  int nslots;
  // round 5: xa is the RAW input of a per-cloud InstanceNorm + LeakyReLU (the KPConv output of a bottleneck block):
  // xa_mean / xa_rstd [nb][ka] (spr_instnorm_stats); the normalisation runs while a tile is staged, so the
  // normalised tensor is never written or read (kpconv_blocks.py:510-525 followed by :553-561 of the reference)
  const float* xa_mean;
  const float* xa_rstd;
  float xa_slope;
};

// slot (16 bytes) of the fragment element (row16, kg) inside the 1-KiB image of k-step ks.
// ds_read_b128 serves a wave in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... : the XOR terms
// (kg in 0..3, 12 for odd k-steps) map each of those row sets onto itself, so reads stay conflict-free,
// while the 8 (kg, k-step parity) pieces of one row land in 8 different 16-byte bank groups for the writer.
__device__ __forceinline__ int frag_slot(int row16, int kg, int ks) { return (row16 ^ kg ^ ((ks & 1) * 12)) + 16 * kg; }

template <int K, int NP>
__device__ __forceinline__ void tail_fetch(const float* __restrict__ x, int row0, int valid, float4 (&p)[NP]) {
  constexpr int Q = K / 4;   // float4 per row
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int e = threadIdx.x + kTailThreads * i;
    const int row = e / Q, j = e % Q;
    p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < valid) p[i] = reinterpret_cast<const float4*>(x + (size_t)(row0 + row) * K)[j];
  }
}

template <int K, int NP, int KS, int KS0>
__device__ __forceinline__ void tail_stage(const float4 (&p)[NP], float s, char* hi_plane, char* lo_plane) {
  constexpr int Q = K / 4;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int e = threadIdx.x + kTailThreads * i;
    const int row = e / Q, j = e % Q;
    const int ks = KS0 + (j >> 3), kg = (j >> 1) & 3, half = j & 1;
    const int off = (((row >> 4) * KS + ks) * 64 + frag_slot(row & 15, kg, ks)) * 16 + half * 8;
    unsigned int h0, l0, h1, l1;
    split_pk_s(p[i].x, p[i].y, s, h0, l0);
    split_pk_s(p[i].z, p[i].w, s, h1, l1);
    *reinterpret_cast<uint2*>(hi_plane + off) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(lo_plane + off) = make_uint2(l0, l1);
  }
}

template <int KA, int KB, int NC, int TR, int PASS>
__global__ __launch_bounds__(kTailThreads) void k_block_tail(const TailArgs a) {
  constexpr int KSA = KA / 32, KSB = KB / 32, KS = KSA + KSB, RT = TR / 16;
  constexpr int NT = NC / (16 * kTailWaves);           // 16-column blocks per wave
  constexpr int NBR = KB > 0 ? 2 : 1;
  constexpr int NPA = TR * KA / 4 / kTailThreads, NPB = KB > 0 ? TR * KB / 4 / kTailThreads : 1;
  constexpr int PLANE = RT * KS * 1024;                // bytes of one fp16 plane of a tile
  constexpr int NSTAT4 = NBR * 2 * NC / 4;             // float4s of a tile's (mean, rstd) image
  static_assert(TR * KA / 4 % kTailThreads == 0 && (KB == 0 || TR * KB / 4 % kTailThreads == 0), "staging shape");
  static_assert(NT >= 1 && NSTAT4 <= kTailThreads, "column shape");
  extern __shared__ __align__(16) char smem[];
  __shared__ float shf[17];
  char* const tile_lds = smem;                          // [2 buffers][hi, lo][PLANE]
  float* const stat_lds = reinterpret_cast<float*>(smem + 4 * PLANE);   // [2 buffers][NBR][2][NC]  (pass 2)

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p16 = lane & 15, kg = lane >> 4;
  const int cg = blockIdx.y;
  const int wcol = wave * 16 * NT;                      // first column of the wave inside the column group
  const int gcol = cg * NC + wcol;

  const int ka = pow2_exp_for(block_absmax(a.xa_parts, shf, a.xa_np));
  const int kwa = pow2_exp_for(block_absmax(a.wa_parts, shf, a.wa_np));
  int kb = 0, kwb = 0;
  if (KB > 0) {
    kb = pow2_exp_for(block_absmax(a.xb_parts, shf, a.xb_np));
    kwb = pow2_exp_for(block_absmax(a.wb_parts, shf, a.wb_np));
  }
  const float sa = pow2f(ka), sb = pow2f(kb);
  const float ua = pow2f(-ka - kwa), ub = pow2f(-kb - kwb);

  // this wave's weight slices: lane (p16, kg) holds w[gcol + 16 nt + p16][32 ks + 8 kg + 0..7]
  f16x8 wh[NT][KS], wl[NT][KS];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bool isb = ks >= KSA;
      const float* src = isb ? a.wb + (size_t)(gcol + 16 * nt + p16) * KB + 32 * (ks - KSA) + 8 * kg
                             : a.wa + (size_t)(gcol + 16 * nt + p16) * KA + 32 * ks + 8 * kg;
      const float sw = isb ? pow2f(kwb) : pow2f(kwa);
      const float4 v0 = reinterpret_cast<const float4*>(src)[0], v1 = reinterpret_cast<const float4*>(src)[1];
      unsigned int h[4], l[4];
      split_pk_s(v0.x, v0.y, sw, h[0], l[0]);
      split_pk_s(v0.z, v0.w, sw, h[1], l[1]);
      split_pk_s(v1.x, v1.y, sw, h[2], l[2]);
      split_pk_s(v1.z, v1.w, sw, h[3], l[3]);
      wh[nt][ks] = __builtin_bit_cast(f16x8, (u32x4){h[0], h[1], h[2], h[3]});
      wl[nt][ks] = __builtin_bit_cast(f16x8, (u32x4){l[0], l[1], l[2], l[3]});
    }

  const int ntiles = a.tile_cu[a.nb];
  float4 pa[NPA], pb[NPB], pst = make_float4(0.f, 0.f, 0.f, 0.f);
  float mx = 0.f;
  // normalise-on-load of xa: every staged element of this thread belongs to the same four channels (the thread
  // count is a multiple of the float4s per row), so one (mean, rstd) float4 pair per tile -- a tile is one cloud
  const bool norm_a = a.xa_mean != nullptr;
  float4 pma = make_float4(0.f, 0.f, 0.f, 0.f), pra = make_float4(1.f, 1.f, 1.f, 1.f);
  int pvalid = 0;
  static_assert(kTailThreads % (KA / 4) == 0, "a thread stages one channel quad");

  // d = {first row, valid rows, cloud} of the tile (one 16-byte record, read one iteration ahead of its use
  // so that no dependent index chain sits in front of the row loads)
  auto fetch = [&](const int4 d) {
    const int row0 = d.x, valid = d.y, cloud = d.z;
    tail_fetch<KA, NPA>(a.xa, row0, valid, pa);
    if (norm_a) {
      const int j = threadIdx.x % (KA / 4);
      pma = reinterpret_cast<const float4*>(a.xa_mean + (size_t)cloud * KA)[j];
      pra = reinterpret_cast<const float4*>(a.xa_rstd + (size_t)cloud * KA)[j];
      pvalid = valid;
    }
    if constexpr (KB > 0) tail_fetch<KB, NPB>(a.xb, row0, valid, pb);
    if constexpr (PASS == 2) {
      if (threadIdx.x < NSTAT4) {
        const int per = NC / 4, brst = threadIdx.x / per, c4 = threadIdx.x % per;   // brst = branch * 2 + stat
        pst = reinterpret_cast<const float4*>(a.stats + ((size_t)cloud * NBR * 2 + brst) * a.n_total + cg * NC)[c4];
      }
    }
  };
  auto stage = [&](int buf) {
    char* hi = tile_lds + buf * 2 * PLANE;
    char* lo = hi + PLANE;
    if (norm_a) {
      // the float operations of k_in_apply (norm_pool.hip), in its order; rows past the cloud's end stay zero
#pragma unroll
      for (int i = 0; i < NPA; ++i) {
        const int row = (threadIdx.x + kTailThreads * i) / (KA / 4);
        float4 w = pa[i];
        w.x = (w.x - pma.x) * pra.x;
        w.y = (w.y - pma.y) * pra.y;
        w.z = (w.z - pma.z) * pra.z;
        w.w = (w.w - pma.w) * pra.w;
        w.x = w.x >= 0.f ? w.x : w.x * a.xa_slope;
        w.y = w.y >= 0.f ? w.y : w.y * a.xa_slope;
        w.z = w.z >= 0.f ? w.z : w.z * a.xa_slope;
        w.w = w.w >= 0.f ? w.w : w.w * a.xa_slope;
        pa[i] = row < pvalid ? w : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    tail_stage<KA, NPA, KS, 0>(pa, sa, hi, lo);
    if constexpr (KB > 0) tail_stage<KB, NPB, KS, KSA>(pb, sb, hi, lo);
    if constexpr (PASS == 2) {
      if (threadIdx.x < NSTAT4) reinterpret_cast<float4*>(stat_lds + buf * NBR * 2 * NC)[threadIdx.x] = pst;
    }
  };

  int t = blockIdx.x;
  const int4 none = make_int4(0, 0, 0, 0);
  int4 d_cur = t < ntiles ? a.tile_desc[t] : none;
  int4 d_next = t + (int)gridDim.x < ntiles ? a.tile_desc[t + gridDim.x] : none;
  if (t < ntiles) {
    fetch(d_cur);
    stage(0);
  }
  __syncthreads();
  for (int it = 0; t < ntiles; ++it) {
    const int tn = t + gridDim.x;
    const bool more = tn < ntiles;
    const int4 d_nn = tn + (int)gridDim.x < ntiles ? a.tile_desc[tn + gridDim.x] : none;
    if (more) fetch(d_next);
    const int buf = it & 1;
    const char* hi = tile_lds + buf * 2 * PLANE;
    const char* lo = hi + PLANE;
    f32x4 acc_a[RT][NT], acc_b[KB > 0 ? RT : 1][NT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        acc_a[rt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (KB > 0) acc_b[rt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int off = ((rt * KS + ks) * 64 + frag_slot(p16, kg, ks)) * 16;
        const f16x8 xh = *reinterpret_cast<const f16x8*>(hi + off);
        const f16x8 xl = *reinterpret_cast<const f16x8*>(lo + off);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          f32x4& c = (KB > 0 && ks >= KSA) ? acc_b[KB > 0 ? rt : 0][nt] : acc_a[rt][nt];
          if constexpr (PASS == 1) {     // C[row = 4 kg + r][col = p16]
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wl[nt][ks], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, wh[nt][ks], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, wh[nt][ks], c, 0, 0, 0);
          } else {                       // C^T[col = 4 kg + r][row = p16]
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[nt][ks], xh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt][ks], xl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[nt][ks], xh, c, 0, 0, 0);
          }
        }
      }
    }
    if constexpr (PASS == 1) {
      // column sums of the tile (rows past the cloud's end were staged as zeros): in-lane over the
      // 4 row tiles x 4 rows, then over the 4 kg groups in a fixed order; float64 throughout.
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int br = 0; br < NBR; ++br) {
          double s = 0.0, ss = 0.0;
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const double v = (double)(br == 0 ? acc_a[rt][nt][r] : acc_b[KB > 0 ? rt : 0][nt][r]);
              s += v;
              ss = __builtin_fma(v, v, ss);
            }
          s += __shfl_xor(s, 16, 64);
          ss += __shfl_xor(ss, 16, 64);
          s += __shfl_xor(s, 32, 64);
          ss += __shfl_xor(ss, 32, 64);
          if (kg == 0) {
            const double u = (double)(br == 0 ? ua : ub);   // exact power of two
            double* p = a.part + ((size_t)t * NBR + br) * 2 * a.n_total + gcol + 16 * nt + p16;
            p[0] = s * u;
            p[a.n_total] = ss * u * u;
          }
        }
    } else {
      const int row0 = d_cur.x, valid = d_cur.y;
      const float* st = stat_lds + buf * NBR * 2 * NC;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int c = wcol + 16 * nt + 4 * kg;          // column inside the group
        const float4 ma = *reinterpret_cast<const float4*>(st + c);
        const float4 ra = *reinterpret_cast<const float4*>(st + NC + c);
        float4 mb = ma, rb = ra;
        if (KB > 0) {
          mb = *reinterpret_cast<const float4*>(st + 2 * NC + c);
          rb = *reinterpret_cast<const float4*>(st + 3 * NC + c);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          const int row = 16 * rt + p16;
          if (row < valid) {
            const size_t o = (size_t)(row0 + row) * a.n_total + cg * NC + c;
            const f32x4 ya = acc_a[rt][nt];
            float4 v;
            v.x = (ya[0] * ua - ma.x) * ra.x;
            v.y = (ya[1] * ua - ma.y) * ra.y;
            v.z = (ya[2] * ua - ma.z) * ra.z;
            v.w = (ya[3] * ua - ma.w) * ra.w;
            if (KB > 0) {
              const f32x4 yb = acc_b[KB > 0 ? rt : 0][nt];
              v.x += (yb[0] * ub - mb.x) * rb.x;
              v.y += (yb[1] * ub - mb.y) * rb.y;
              v.z += (yb[2] * ub - mb.z) * rb.z;
              v.w += (yb[3] * ub - mb.w) * rb.w;
            } else if (a.add != nullptr) {
              const float4 ad = *reinterpret_cast<const float4*>(a.add + o);
              v.x += ad.x;
              v.y += ad.y;
              v.z += ad.z;
              v.w += ad.w;
            }
            v.x = v.x >= 0.f ? v.x : v.x * a.slope;
            v.y = v.y >= 0.f ? v.y : v.y * a.slope;
            v.z = v.z >= 0.f ? v.z : v.z * a.slope;
            v.w = v.w >= 0.f ? v.w : v.w * a.slope;
            *reinterpret_cast<float4*>(a.out + o) = v;
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
          }
        }
      }
    }
    if (more) stage(buf ^ 1);
    __syncthreads();
    t = tn;
    d_cur = d_next;
    d_next = d_nn;
  }
  if constexpr (PASS == 2) {
    if (a.out_range != nullptr) {
      mx = wave_max(mx);
      if (lane == 0) shf[wave] = mx;
      __syncthreads();
      if (threadIdx.x == 0) {
        float m = shf[0];
        for (int w = 1; w < kTailWaves; ++w) m = fmaxf(m, shf[w]);
        atomicMax(reinterpret_cast<unsigned int*>(a.out_range) + ((blockIdx.x + gridDim.x * blockIdx.y) & (a.nslots - 1)),
                  __float_as_uint(m));
      }
    }
  }
}

// mean / rstd per (cloud, branch, channel): the tile partials of a cloud summed in tile order.
__global__ void k_tail_final(const double* __restrict__ part, const int* __restrict__ cu, const int* __restrict__ tile_cu,
                             int nb, int nbr, int n_total, float eps, float* __restrict__ stats) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)nb * nbr * n_total) return;
  const int col = (int)(i % n_total), br = (int)((i / n_total) % nbr), cloud = (int)(i / ((long)n_total * nbr));
  const int t0 = tile_cu[cloud], t1 = tile_cu[cloud + 1];
  double s = 0.0, ss = 0.0;
  const double* p = part + ((size_t)t0 * nbr + br) * 2 * n_total + col;
  const size_t step = (size_t)nbr * 2 * n_total;
  int t = t0;
  for (; t + 4 <= t1; t += 4) {      // four tiles' partials in flight; the order of the additions is fixed
    double v[4], w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = p[u * step];
      w[u] = p[u * step + n_total];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s += v[u];
      ss += w[u];
    }
    p += 4 * step;
  }
  for (; t < t1; ++t) {
    s += p[0];
    ss += p[n_total];
    p += step;
  }
  const int len = cu[cloud + 1] - cu[cloud];
  const double n = len > 0 ? (double)len : 1.0;
  const double m = s / n;
  double var = ss / n - m * m;
  if (var < 0.0) var = 0.0;
  float* o = stats + ((size_t)cloud * nbr + br) * 2 * n_total + col;
  o[0] = (float)m;
  o[n_total] = (float)(1.0 / sqrt(var + (double)eps));
}

// tile_cu[c] = number of tr-row tiles of the clouds before c (tiles start at each cloud's first row)
__global__ __launch_bounds__(1024) void k_tile_cu(const int* __restrict__ cu, int nb, int tr, int* __restrict__ tile_cu) {
  __shared__ int sh[1024];
  int carry = 0;
  for (int base = 0; base < nb; base += 1024) {
    const int c = base + threadIdx.x;
    int v = c < nb ? (cu[c + 1] - cu[c] + tr - 1) / tr : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const int add = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += add;
      __syncthreads();
    }
    if (c < nb) tile_cu[c + 1] = carry + sh[threadIdx.x];
    carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) tile_cu[0] = 0;
}

__global__ void k_tile_desc(const int* __restrict__ cu, const int* __restrict__ tile_cu, int tr, int4* __restrict__ desc) {
  const int c = blockIdx.x;
  const int t0 = tile_cu[c], nt = tile_cu[c + 1] - t0, r0 = cu[c], len = cu[c + 1] - r0;
  for (int i = threadIdx.x; i < nt; i += blockDim.x) desc[t0 + i] = make_int4(r0 + i * tr, min(tr, len - i * tr), c, 0);
}

struct TailShape {
  int nc, tr;
};
// column-group width and tile rows of a supported shape; {0, 0} otherwise
TailShape tail_shape(int ka, int kb, int n_total) {
  if (ka == 32 && kb == 64 && n_total == 128) return {128, 64};
  if (ka == 32 && kb == 0 && n_total == 128) return {128, 64};
  if (ka == 64 && kb == 128 && n_total == 256) return {256, 64};
  if (ka == 64 && kb == 0 && n_total == 256) return {256, 64};
  if (ka == 128 && kb == 256 && n_total == 512) return {128, 32};
  if (ka == 128 && kb == 0 && n_total == 512) return {256, 64};
  return {0, 0};
}

template <int KA, int KB, int NC, int TR>
int launch_tail(const TailArgs& a, float eps, hipStream_t stream) {
  constexpr int KS = (KA + KB) / 32, NBR = KB > 0 ? 2 : 1;
  constexpr int lds = 4 * (TR / 16) * KS * 1024 + 2 * NBR * 2 * NC * (int)sizeof(float);
  const void* k1 = (const void*)k_block_tail<KA, KB, NC, TR, 1>;
  const void* k2 = (const void*)k_block_tail<KA, KB, NC, TR, 2>;
  if (int rc = ensure_dyn_lds(k1, lds)) return rc;
  if (int rc = ensure_dyn_lds(k2, lds)) return rc;
  static int per_cu[2] = {0, 0};
  if (per_cu[0] == 0) {
    int n1 = 0, n2 = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n1, k1, kTailThreads, lds) != hipSuccess || n1 < 1) n1 = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n2, k2, kTailThreads, lds) != hipSuccess || n2 < 1) n2 = 1;
    per_cu[1] = n2 > 2 ? 2 : n2;
    per_cu[0] = n1 > 2 ? 2 : n1;
  }
  const int ncg = a.n_total / NC;
  const int max_tiles = cdiv(a.n, TR) + a.nb;
  auto grid_x = [&](int occ) {
    int g = device_cu_count() * occ / ncg;
    if (g > max_tiles) g = max_tiles;
    return g < 1 ? 1 : g;
  };
  hipLaunchKernelGGL((k_block_tail<KA, KB, NC, TR, 1>), dim3(grid_x(per_cu[0]), ncg), dim3(kTailThreads), lds, stream, a);
  hipLaunchKernelGGL(k_tail_final, dim3(cdiv((long)a.nb * NBR * a.n_total, 256)), dim3(256), 0, stream, a.part, a.cu,
                     a.tile_cu, a.nb, NBR, a.n_total, eps, a.stats);
  hipLaunchKernelGGL((k_block_tail<KA, KB, NC, TR, 2>), dim3(grid_x(per_cu[1]), ncg), dim3(kTailThreads), lds, stream, a);
  SPR_LAUNCH_CHECK();
  return 0;
}

}  // namespace
}  // namespace spr

using namespace spr;

extern "C" int spr_block_tail_tile_rows(int ka, int kb, int n_out) { return tail_shape(ka, kb, n_out).tr; }

// tiles: int32 [spr_block_tail_tiles_len(n, nb, tile_rows)] = tile_cu [nb + 1], padding to a multiple of 4,
// then one {first row, valid rows, cloud, 0} record per tile
static size_t tiles_desc_offset(int nb) { return align_up((size_t)nb + 1, 4); }
extern "C" size_t spr_block_tail_tiles_len(int n, int nb, int tile_rows) {
  return tiles_desc_offset(nb) + 4 * ((size_t)cdiv(n > 0 ? n : 1, tile_rows > 0 ? tile_rows : 64) + (size_t)(nb > 0 ? nb : 1));
}

extern "C" int spr_block_tail_tiles(const int* cu, int n, int nb, int tile_rows, int* tiles, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(cu != nullptr && tiles != nullptr && n >= 1 && nb >= 1 && tile_rows >= 16 && ((uintptr_t)tiles & 15) == 0,
              "block_tail_tiles: bad arguments");
  hipLaunchKernelGGL(k_tile_cu, dim3(1), dim3(1024), 0, stream, cu, nb, tile_rows, tiles);
  hipLaunchKernelGGL(k_tile_desc, dim3(nb), dim3(256), 0, stream, cu, tiles, tile_rows,
                     reinterpret_cast<int4*>(tiles + tiles_desc_offset(nb)));
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t spr_block_tail_workspace_bytes(int n, int nb, int kb, int n_out, int tile_rows) {
  const size_t nbr = kb > 0 ? 2 : 1;
  const size_t tiles = (size_t)cdiv(n > 0 ? n : 1, tile_rows > 0 ? tile_rows : 64) + (size_t)(nb > 0 ? nb : 1);
  return align_up(tiles * nbr * 2 * (size_t)n_out * sizeof(double), 256) +
         align_up((size_t)(nb > 0 ? nb : 1) * nbr * 2 * (size_t)n_out * sizeof(float), 256) +
         4 * align_up(kAmaxParts * sizeof(float), 256);
}

static int block_tail_impl(const float* xa, int ka, const float* wa, const float* xb, int kb, const float* wb,
                           const float* add, const int* cu, const int* tiles, int n, int nb, int n_out, float eps,
                           float slope, float* out, const float* xa_range, int xa_range_n, const float* wa_range,
                           int wa_range_n, const float* xb_range, int xb_range_n, const float* wb_range,
                           int wb_range_n, float* out_range, int out_range_n, const float* xa_mean,
                           const float* xa_rstd, float xa_slope, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(spr::gemm_mode() == 1, "block_tail: only in the split-fp16 product mode (spr_set_gemm_mode(1))");
  const TailShape sh = tail_shape(ka, kb, n_out);
  SPR_REQUIRE(sh.tr > 0, "block_tail: unsupported shape ka=%d kb=%d n_out=%d (spr_block_tail_tile_rows)", ka, kb, n_out);
  SPR_REQUIRE(xa && wa && cu && tiles && out && n > 0 && nb >= 1, "block_tail: null operand or empty input");
  SPR_REQUIRE((kb > 0) == (xb != nullptr) && (kb > 0) == (wb != nullptr), "block_tail: xb / wb must come with kb > 0");
  SPR_REQUIRE(!(kb > 0 && add != nullptr), "block_tail: either a projected shortcut (xb, wb) or a plain one (add)");
  SPR_REQUIRE(out_range == nullptr || (out_range_n >= 1 && (out_range_n & (out_range_n - 1)) == 0),
              "block_tail: out_range_n must be a power of two");
  SPR_REQUIRE(ws != nullptr && ws_bytes >= spr_block_tail_workspace_bytes(n, nb, kb, n_out, sh.tr),
              "block_tail: workspace too small");
  const int nbr = kb > 0 ? 2 : 1;
  Workspace w(ws, ws_bytes);
  TailArgs a;
  a.part = w.take<double>(((size_t)cdiv(n, sh.tr) + nb) * nbr * 2 * n_out);
  a.stats = w.take<float>((size_t)nb * nbr * 2 * n_out);
  float* mp[4];
  for (int i = 0; i < 4; ++i) mp[i] = w.take<float>(kAmaxParts);
  SPR_REQUIRE(mp[3] != nullptr, "block_tail: workspace carve failed");
  a.xa = xa; a.xb = xb; a.wa = wa; a.wb = wb; a.add = add; a.cu = cu; a.tile_cu = tiles;
  a.tile_desc = reinterpret_cast<const int4*>(tiles + tiles_desc_offset(nb));
  a.nb = nb; a.n = n; a.n_total = n_out; a.slope = slope; a.out = out; a.out_range = out_range; a.nslots = out_range_n;
  a.xa_mean = xa_mean; a.xa_rstd = xa_rstd; a.xa_slope = xa_slope;
  SPR_REQUIRE((xa_mean == nullptr) == (xa_rstd == nullptr), "block_tail: xa_mean and xa_rstd come together");
  SPR_REQUIRE(xa_mean == nullptr || xa_range != nullptr,
              "block_tail: a normalised-on-load xa needs the bound of the NORMALISED values as xa_range");
  auto range = [&](const float* x, long rows, int cols, const float* given, int given_n, float* scratch,
                   const float*& parts, int& np) -> int {
    if (given != nullptr) {
      SPR_REQUIRE(given_n >= 1, "block_tail: a range needs a count");
      parts = given;
      np = given_n;
      return 0;
    }
    parts = scratch;
    np = kAmaxParts;
    return launch_absmax(x, rows, cols, cols, scratch, stream);
  };
  if (int rc = range(xa, n, ka, xa_range, xa_range_n, mp[0], a.xa_parts, a.xa_np)) return rc;
  if (int rc = range(wa, n_out, ka, wa_range, wa_range_n, mp[1], a.wa_parts, a.wa_np)) return rc;
  a.xb_parts = a.wb_parts = nullptr;
  a.xb_np = a.wb_np = 0;
  if (kb > 0) {
    if (int rc = range(xb, n, kb, xb_range, xb_range_n, mp[2], a.xb_parts, a.xb_np)) return rc;
    if (int rc = range(wb, n_out, kb, wb_range, wb_range_n, mp[3], a.wb_parts, a.wb_np)) return rc;
  }
#define SPR_TAIL(KA_, KB_, NC_, TR_) \
  if (ka == KA_ && kb == KB_ && sh.nc == NC_ && sh.tr == TR_) return launch_tail<KA_, KB_, NC_, TR_>(a, eps, stream)
  SPR_TAIL(32, 64, 128, 64);
  SPR_TAIL(32, 0, 128, 64);
  SPR_TAIL(64, 128, 256, 64);
  SPR_TAIL(64, 0, 256, 64);
  SPR_TAIL(128, 256, 128, 32);
  SPR_TAIL(128, 0, 256, 64);
#undef SPR_TAIL
  SPR_REQUIRE(false, "block_tail: no kernel for ka=%d kb=%d n_out=%d", ka, kb, n_out);
  return 1;
}

extern "C" int spr_block_tail(const float* xa, int ka, const float* wa, const float* xb, int kb, const float* wb,
                              const float* add, const int* cu, const int* tiles, int n, int nb, int n_out, float eps,
                              float slope, float* out, const float* xa_range, int xa_range_n, const float* wa_range,
                              int wa_range_n, const float* xb_range, int xb_range_n, const float* wb_range,
                              int wb_range_n, float* out_range, int out_range_n, void* ws, size_t ws_bytes,
                              void* stream_) {
  return block_tail_impl(xa, ka, wa, xb, kb, wb, add, cu, tiles, n, nb, n_out, eps, slope, out, xa_range, xa_range_n,
                         wa_range, wa_range_n, xb_range, xb_range_n, wb_range, wb_range_n, out_range, out_range_n,
                         nullptr, nullptr, 1.0f, ws, ws_bytes, stream_);
}

// The same with xa = the RAW input of a per-cloud InstanceNorm + LeakyReLU(xa_slope) whose statistics the caller
// has (spr_instnorm_stats: xa_mean, xa_rstd [nb][ka]): lrelu(IN(lrelu(IN(xa)) wa^T) + ...).  xa_range must bound
// the NORMALISED values (sqrt(longest cloud) is always valid: |x - mean| <= sqrt(n - 1) sigma).
extern "C" int spr_block_tail_n(const float* xa, int ka, const float* wa, const float* xb, int kb, const float* wb,
                                const float* add, const int* cu, const int* tiles, int n, int nb, int n_out, float eps,
                                float slope, float* out, const float* xa_range, int xa_range_n, const float* wa_range,
                                int wa_range_n, const float* xb_range, int xb_range_n, const float* wb_range,
                                int wb_range_n, float* out_range, int out_range_n, const float* xa_mean,
                                const float* xa_rstd, float xa_slope, void* ws, size_t ws_bytes, void* stream_) {
  SPR_REQUIRE(xa_mean != nullptr && xa_rstd != nullptr, "block_tail_n: statistics missing");
  return block_tail_impl(xa, ka, wa, xb, kb, wb, add, cu, tiles, n, nb, n_out, eps, slope, out, xa_range, xa_range_n,
                         wa_range, wa_range_n, xb_range, xb_range_n, wb_range, wb_range_n, out_range, out_range_n,
                         xa_mean, xa_rstd, xa_slope, ws, ws_bytes, stream_);
}
