// Generic batched, strided, exact-f32 MFMA GEMM -- the workhorse of the BACKWARD
// pass.  Every gradient of the hot path that is a matrix
// product is expressed through it:
//   nn.Linear            dX = dY W            dW = dY^T X      (transformers.py:96-104,
//                                                               kpconv_blocks.py:549)
//   KPConv               d(weighted feats) = g W_flat^T,  dW_flat = wf^T g
//                                                              (kpconv_blocks.py:394-406)
//   attention core       S = Q K^T, dV = P^T dO, dP = dO V^T, dQ = dS K, dK = dS^T Q
//                                                              (transformers.py:198-227)
//   InfoNCE / correlation heads                                (feature_loss.py:268-296,
//                                                               qk_regtr_full.py:453)
// which in the reference are produced by torch autograd.
//
//   C_b(i, j) = alpha * sum_k A_b(i, k) B_b(k, j) + beta * C_b(i, j)
//   A_b(i, k) = A[a_off_b + i * sa_i + k * sa_k]          (any strides: N / T operands,
//   B_b(k, j) = B[b_off_b + k * sb_k + j * sb_j]           head slices of [T, 3d] rows, ...)
//   C_b(i, j) = C[c_off_b + i * sc_i + j * sc_j]
// Batches carry their own sizes (varlen segments) in a device descriptor array.
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact f32, a k-ordered fmaf chain per
// output) -- gradients are formed in plain fp32 like the reference's.
//
// Kernel shape: 64 x 64 output tile per workgroup of 4 waves (one 32 x 32
// accumulator each), K slabs of 16 staged k-major in LDS so that the MFMA
// fragments (lane = row/col, k = 2 s + lane/32) are conflict-free reads whatever
// the operand strides; the global side walks whichever operand dimension is
// contiguous with consecutive lanes.
#include "spr_common.h"

#include <cstdlib>

namespace spr {
namespace {

struct BgemmDesc {
  long long a_off, b_off, c_off;
  int m, n, k, pad;   // pad: 0, or the leading dimension of A for this batch
};

constexpr int TB = 64;    // output tile edge
constexpr int KB = 16;    // K slab
constexpr int LDT = TB + 1;

__global__ __launch_bounds__(256) void k_bgemm_f32(const float* __restrict__ A, const float* __restrict__ B,
                                                   float* __restrict__ C, const BgemmDesc* __restrict__ desc,
                                                   long sa_i, long sa_k, long sb_k, long sb_j, long sc_i,
                                                   long sc_j, float alpha, float beta, int tiles_n_max) {
  __shared__ float As[KB * LDT];   // [k][i]
  __shared__ float Bs[KB * LDT];   // [k][j]
  const BgemmDesc d = desc[blockIdx.y];
  // per-batch leading dimension of A (d.pad > 0): replaces whichever A stride is not 1 -- batches whose matrices have
  // their own row length (the per-pair score matrices) then fit one launch
  if (d.pad > 0) {
    if (sa_k == 1) sa_i = d.pad;
    else sa_k = d.pad;
  }
  const int tn = (d.n + TB - 1) / TB, tm = (d.m + TB - 1) / TB;
  const int tile = blockIdx.x;
  if (tile >= tn * tm) return;
  const int i0 = (tile / tn) * TB, j0 = (tile % tn) * TB;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
  const float* Ab = A + d.a_off;
  const float* Bb = B + d.b_off;
  float* Cb = C + d.c_off;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // staging maps: 256 threads cover a [KB x TB] slab in 4 passes; consecutive threads walk
  // the operand's contiguous dimension
  const bool a_k_fast = sa_k == 1;     // rows contiguous along k
  const bool b_k_fast = sb_k == 1;
  for (int k0 = 0; k0 < d.k; k0 += KB) {
#pragma unroll
    for (int p = 0; p < (KB * TB) / 256; ++p) {
      const int e = p * 256 + tid;
      int ia, ka, jb, kb;
      if (a_k_fast) { ka = e % KB; ia = e / KB; } else { ia = e % TB; ka = e / TB; }
      if (b_k_fast) { kb = e % KB; jb = e / KB; } else { jb = e % TB; kb = e / TB; }
      const bool oka = i0 + ia < d.m && k0 + ka < d.k;
      const bool okb = j0 + jb < d.n && k0 + kb < d.k;
      As[ka * LDT + ia] = oka ? Ab[(long)(i0 + ia) * sa_i + (long)(k0 + ka) * sa_k] : 0.f;
      Bs[kb * LDT + jb] = okb ? Bb[(long)(k0 + kb) * sb_k + (long)(j0 + jb) * sb_j] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KB / 2; ++s) {
      const float a = As[(2 * s + lh) * LDT + wi + l31];
      const float b = Bs[(2 * s + lh) * LDT + wj + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  // C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const int j = j0 + wj + l31;
  if (j < d.n) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = i0 + wi + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (i < d.m) {
        float* c = Cb + (long)i * sc_i + (long)j * sc_j;
        const float v = alpha * acc[r];
        *c = beta != 0.f ? v + beta * *c : v;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Large-tile form (round 3).  The 64 x 64 kernel above pays one exposed global-load latency per
// 16-deep K slab for 8 MFMAs per wave and ran the training step's ~1.4 TFLOP of gradient products
// at ~10 TFLOP/s (70 % of a 208 ms step at 4 pairs x 16 384 points).  Same arithmetic -- exact f32,
// v_mfma_f32_32x32x2_f32, the SAME k-ordered accumulation per output, hence bitwise the same
// results -- with
//   * TM x TN = 128 x 128 tiles (each of the 4 waves owns 64 x 64 = 2 x 2 accumulators) or 128 x 32
//     (4 x 1 waves, one accumulator each: the attention backward's d_head-wide outputs),
//   * the next K slab's global loads issued into registers BEFORE the current slab's MFMAs
//     (software pipeline: one exposed latency per tile instead of one per slab),
//   * two LDS slab buffers, one barrier per slab.
template <int TM, int TN>
__global__ __launch_bounds__(256) void k_bgemm_f32_t(const float* __restrict__ A, const float* __restrict__ B,
                                                     float* __restrict__ C, const BgemmDesc* __restrict__ desc,
                                                     long sa_i, long sa_k, long sb_k, long sb_j, long sc_i,
                                                     long sc_j, float alpha, float beta) {
  constexpr int WN = TN == 128 ? 2 : 1;            // waves along n
  constexpr int WM = 4 / WN;                       // waves along m
  constexpr int MI = TM / (32 * WM), NJ = TN / (32 * WN);   // 32 x 32 accumulators per wave
  constexpr int LDA = TM + 1, LDB = TN + 1;
  constexpr int NA = KB * TM / 256, NB = KB * TN / 256;     // staged elements per thread and slab
  __shared__ float As[2][KB * LDA];   // [k][i]
  __shared__ float Bs[2][KB * LDB];   // [k][j]
  const BgemmDesc d = desc[blockIdx.y];
  // per-batch leading dimension of A (d.pad > 0): replaces whichever A stride is not 1 -- batches whose matrices have
  // their own row length (the per-pair score matrices) then fit one launch
  if (d.pad > 0) {
    if (sa_k == 1) sa_i = d.pad;
    else sa_k = d.pad;
  }
  const int tn = (d.n + TN - 1) / TN, tm = (d.m + TM - 1) / TM;
  const int tile = blockIdx.x;
  if (tile >= tn * tm) return;
  const int i0 = (tile / tn) * TM, j0 = (tile % tn) * TN;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wi = (wave / WN) * (32 * MI), wj = (wave % WN) * (32 * NJ);
  const float* Ab = A + d.a_off;
  const float* Bb = B + d.b_off;
  float* Cb = C + d.c_off;
  const bool a_k_fast = sa_k == 1, b_k_fast = sb_k == 1;

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][nj][r] = 0.f;

  float ra[NA], rb[NB];
  auto fetch = [&](int k0) {   // slab k0 .. k0 + KB - 1 -> registers (consecutive threads walk the contiguous dimension)
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      const int e = p * 256 + tid;
      int ia, ka;
      if (a_k_fast) { ka = e % KB; ia = e / KB; } else { ia = e % TM; ka = e / TM; }
      const bool ok = i0 + ia < d.m && k0 + ka < d.k;
      ra[p] = ok ? Ab[(long)(i0 + ia) * sa_i + (long)(k0 + ka) * sa_k] : 0.f;
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      const int e = p * 256 + tid;
      int jb, kb;
      if (b_k_fast) { kb = e % KB; jb = e / KB; } else { jb = e % TN; kb = e / TN; }
      const bool ok = j0 + jb < d.n && k0 + kb < d.k;
      rb[p] = ok ? Bb[(long)(k0 + kb) * sb_k + (long)(j0 + jb) * sb_j] : 0.f;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int p = 0; p < NA; ++p) {
      const int e = p * 256 + tid;
      int ia, ka;
      if (a_k_fast) { ka = e % KB; ia = e / KB; } else { ia = e % TM; ka = e / TM; }
      As[buf][ka * LDA + ia] = ra[p];
    }
#pragma unroll
    for (int p = 0; p < NB; ++p) {
      const int e = p * 256 + tid;
      int jb, kb;
      if (b_k_fast) { kb = e % KB; jb = e / KB; } else { jb = e % TN; kb = e / TN; }
      Bs[buf][kb * LDB + jb] = rb[p];
    }
  };
  fetch(0);
  stash(0);
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < d.k; k0 += KB, buf ^= 1) {
    const bool more = k0 + KB < d.k;
    if (more) fetch(k0 + KB);            // in flight under this slab's MFMAs
#pragma unroll
    for (int s2 = 0; s2 < KB / 2; ++s2) {
      float a[MI], b[NJ];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = As[buf][(2 * s2 + lh) * LDA + wi + 32 * mi + l31];
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) b[nj] = Bs[buf][(2 * s2 + lh) * LDB + wj + 32 * nj + l31];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
          acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[nj], acc[mi][nj], 0, 0, 0);
    }
    if (more) stash(buf ^ 1);            // the other buffer: last read one slab ago, behind a barrier
    __syncthreads();
  }
  // C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
      const int j = j0 + wj + 32 * nj + l31;
      if (j >= d.n) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + wi + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (i < d.m) {
          float* c = Cb + (long)i * sc_i + (long)j * sc_j;
          const float v = alpha * acc[mi][nj][r];
          *c = beta != 0.f ? v + beta * *c : v;
        }
      }
    }
}

// out[j] = scale * sum over parts p of parts[p][j]   (fixed order: deterministic split-K)
// A block = 64 outputs x 4 part lanes: lane q sums the parts p = q (mod 4) in ascending order, four loads
// in flight; the four lane sums are combined as (s0 + s1) + (s2 + s3).  (One thread per output walking all
// parts left a bias gradient -- 256 outputs, hundreds of parts -- on ONE workgroup: 46 us per call.)
__global__ __launch_bounds__(256) void k_reduce_parts(const float* __restrict__ parts, int nparts, long n, float scale,
                                                      float* __restrict__ out, int accumulate) {
  __shared__ float sh[4][64];
  const int q = threadIdx.x >> 6, jl = threadIdx.x & 63;
  const long j = (long)blockIdx.x * 64 + jl;
  float s = 0.f;
  if (j < n) {
    int p = q;
    for (; p + 12 < nparts; p += 16) {
      const float a = parts[(long)p * n + j], b = parts[(long)(p + 4) * n + j];
      const float c = parts[(long)(p + 8) * n + j], d = parts[(long)(p + 12) * n + j];
      s += a;
      s += b;
      s += c;
      s += d;
    }
    for (; p < nparts; p += 4) s += parts[(long)p * n + j];
  }
  sh[q][jl] = s;
  __syncthreads();
  if (q == 0 && j < n) {
    const float t = ((sh[0][jl] + sh[1][jl]) + (sh[2][jl] + sh[3][jl])) * scale;
    out[j] = accumulate ? out[j] + t : t;
  }
}

// ---------------------------------------------------------------------------
// out[nl, nr] = L[rows, nl]^T R[rows, nr] with FLOAT64 accumulation, for small nl * nr and very tall
// operands: the weight gradient of the first KPConv (15 x 1 x 64 entries, a sum over every point of
// the batch).  There the summands nearly cancel -- the output gradient sums to zero over each cloud
// (InstanceNorm's backward) and the weighted features of the constant input are nearly the same for
// every point -- so a float32 running sum over 10^4 - 10^5 rows loses three digits (measured at BASELINE
// size against the float64 oracle: 2.5e-3 of the tensor's scale; every other encoder tensor 2e-6).
// Same policy as the InstanceNorm statistics: long sums in float64, fixed slabs, fixed-order reduction.
constexpr int kTnSlab = 1024;
__global__ __launch_bounds__(256) void k_tn_f64_part(const float* __restrict__ Lm, const float* __restrict__ Rm,
                                                     long rows, int nl, int nr, double* __restrict__ part) {
  const long r0 = (long)blockIdx.x * kTnSlab, r1 = min(r0 + (long)kTnSlab, rows);
  const int nout = nl * nr;
  for (int o0 = 0; o0 < nout; o0 += 256 * 4) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    int il[4], ir[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int o = min(o0 + u * 256 + (int)threadIdx.x, nout - 1);
      il[u] = o / nr;
      ir[u] = o % nr;
    }
    for (long r = r0; r < r1; ++r) {
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] += (double)Lm[r * nl + il[u]] * (double)Rm[r * nr + ir[u]];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int o = o0 + u * 256 + (int)threadIdx.x;
      if (o < nout) part[(long)blockIdx.x * nout + o] = acc[u];
    }
  }
}
__global__ void k_tn_f64_final(const double* __restrict__ part, int nslab, int nout, float* __restrict__ out) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= nout) return;
  double s = 0.0;
  for (int p = 0; p < nslab; ++p) s += part[(long)p * nout + o];
  out[o] = (float)s;
}

}  // namespace
}  // namespace spr

using namespace spr;

extern "C" size_t spr_tn_product_f64_workspace_bytes(long rows, int nl, int nr) {
  return (size_t)cdiv(rows > 0 ? rows : 1, kTnSlab) * (size_t)nl * nr * sizeof(double) + 256;
}

extern "C" int spr_tn_product_f64(const float* Lm, const float* Rm, long rows, int nl, int nr, float* out,
                                  void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(Lm && Rm && out && rows >= 1 && nl >= 1 && nr >= 1, "tn_product_f64: bad arguments");
  SPR_REQUIRE((long)nl * nr <= 65536, "tn_product_f64: meant for small outputs (%d x %d)", nl, nr);
  SPR_REQUIRE(ws != nullptr && ws_bytes >= spr_tn_product_f64_workspace_bytes(rows, nl, nr), "tn_product_f64: workspace too small");
  const int nslab = cdiv(rows, kTnSlab);
  double* part = (double*)ws;
  hipLaunchKernelGGL(k_tn_f64_part, dim3(nslab), dim3(256), 0, stream, Lm, Rm, rows, nl, nr, part);
  hipLaunchKernelGGL(k_tn_f64_final, dim3(cdiv((long)nl * nr, 256)), dim3(256), 0, stream, part, nslab, nl * nr, out);
  SPR_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------
// Weight gradients in the forward's arithmetic (round 3): parts[b][i][j] = sum over the rows t of batch b of
// L[t, i] R[t, j]  (L = dY [rows, nl], R = X [rows, nr], both row-major) with range-scaled split-fp16 operands
// and v_mfma_f32_32x32x16_f16 -- three MFMAs per product, fp32 accumulation -- instead of the exact-f32 MFMA
// of spr_bgemm (1/5 of its matrix-pipe time).  The contraction runs over the ROW index of both operands, so a
// slab of 32 rows is transposed on its way into LDS: a thread loads a 4 (rows) x 4 (columns) block, splits it
// and writes, per column, the 4 consecutive-row halves as one 8-byte word into the [column][row] image the
// MFMA fragments read (8 consecutive k per lane); lanes are mapped row-quad fastest so that the 16 lanes of a
// ds_write_b64 group cover all banks.  128 x 128 outputs per workgroup of 4 waves (64 x 64 each), slab-deep
// register prefetch, one workgroup per (tile, row batch); the batches are summed by spr_reduce_parts.
namespace spr {
namespace {
typedef _Float16 th8 __attribute__((ext_vector_type(8)));
constexpr int TNT = 128;       // output tile edge
constexpr int TNK = 32;        // rows (contraction) per slab
constexpr int TNS = 40;        // LDS row stride in halves (32 + 8 pad: conflict-free ds_read_b128)

__global__ __launch_bounds__(256) void k_gemm_tn_h3(const float* __restrict__ Lm, const float* __restrict__ Rm, long rows,
                                                    int nl, int nr, int chunk, const float* __restrict__ l_parts,
                                                    int n_lp, const float* __restrict__ r_parts, int n_rp,
                                                    float* __restrict__ parts) {
  __shared__ __align__(16) _Float16 Ah[TNT * TNS], Al[TNT * TNS], Bh[TNT * TNS], Bl[TNT * TNS];
  __shared__ float shf[17];
  const int tn = (nr + TNT - 1) / TNT;
  const int i0 = (blockIdx.x / tn) * TNT, j0 = (blockIdx.x % tn) * TNT;
  const long r_beg = (long)blockIdx.y * chunk;
  const long r_end = r_beg + chunk < rows ? r_beg + chunk : rows;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wi = (wave >> 1) * 64, wj = (wave & 1) * 64;
  const int ka = pow2_exp_for(block_absmax(l_parts, shf, n_lp));
  const int kb = pow2_exp_for(block_absmax(r_parts, shf, n_rp));
  const float sa = pow2f(ka), sb = pow2f(kb), unscale = pow2f(-ka - kb);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging role: row quad tq (rows 4 tq .. 4 tq + 3 of the slab), column quad n4 (columns 4 n4 .. + 3 of the tile)
  const int tq = tid & 7, n4 = tid >> 3;
  const int la = min(i0 + 4 * n4, nl - 4), lb = min(j0 + 4 * n4, nr - 4);   // clamped: columns past the edge are never stored
  float4 ra[4], rb[4];
  auto fetch = [&](long t0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long t = t0 + 4 * tq + e;
      ra[e] = make_float4(0.f, 0.f, 0.f, 0.f);
      rb[e] = ra[e];
      if (t < r_end) {
        ra[e] = *reinterpret_cast<const float4*>(Lm + t * nl + la);
        rb[e] = *reinterpret_cast<const float4*>(Rm + t * nr + lb);
      }
    }
  };
  auto put = [&](const float4 (&v)[4], float sc, _Float16* hi, _Float16* lo) {
    const float c[4][4] = {{v[0].x, v[1].x, v[2].x, v[3].x}, {v[0].y, v[1].y, v[2].y, v[3].y},
                           {v[0].z, v[1].z, v[2].z, v[3].z}, {v[0].w, v[1].w, v[2].w, v[3].w}};
#pragma unroll
    for (int q = 0; q < 4; ++q) {       // column 4 n4 + q: its four consecutive rows
      unsigned int h0, l0, h1, l1;
      split_pk_s(c[q][0], c[q][1], sc, h0, l0);
      split_pk_s(c[q][2], c[q][3], sc, h1, l1);
      const int o = (4 * n4 + q) * TNS + 4 * tq;
      *reinterpret_cast<uint2*>(hi + o) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(lo + o) = make_uint2(l0, l1);
    }
  };
  fetch(r_beg);
  for (long t0 = r_beg; t0 < r_end; t0 += TNK) {
    put(ra, sa, Ah, Al);
    put(rb, sb, Bh, Bl);
    __syncthreads();
    if (t0 + TNK < r_end) fetch(t0 + TNK);
#pragma unroll
    for (int s2 = 0; s2 < TNK / 16; ++s2) {
      const int ko = 16 * s2 + 8 * lh;
      th8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const th8*>(Ah + (wi + 32 * i + l31) * TNS + ko);
        al[i] = *reinterpret_cast<const th8*>(Al + (wi + 32 * i + l31) * TNS + ko);
        bh[i] = *reinterpret_cast<const th8*>(Bh + (wj + 32 * i + l31) * TNS + ko);
        bl[i] = *reinterpret_cast<const th8*>(Bl + (wj + 32 * i + l31) * TNS + ko);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  // C layout: col = l31, row = (r & 3) + 8 (r >> 2) + 4 lh
  float* P = parts + (size_t)blockIdx.y * nl * nr;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = j0 + wj + 32 * j + l31;
      if (col >= nr) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = i0 + wi + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < nl) P[(size_t)row * nr + col] = acc[i][j][r] * unscale;
      }
    }
}
}  // namespace
}  // namespace spr

extern "C" size_t spr_tn_product_split_workspace_bytes(void) { return 2 * align_up(kAmaxParts * sizeof(float), 256); }

// parts [nbatch][nl][nr], nbatch = ceil(rows / chunk): the caller sums them (spr_reduce_parts).  l_range / r_range:
// range partials of L / R (NULL: measured here).  nl, nr multiples of 4; chunk a multiple of 16.
extern "C" int spr_tn_product_split(const float* Lm, const float* Rm, long rows, int nl, int nr, int chunk,
                                    const float* l_range, int l_range_n, const float* r_range, int r_range_n,
                                    float* parts, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(Lm && Rm && parts && rows >= 1 && nl >= 4 && nr >= 4 && nl % 4 == 0 && nr % 4 == 0 && chunk >= 16 &&
                  chunk % 16 == 0, "tn_product_split: bad arguments (rows=%ld nl=%d nr=%d chunk=%d)", rows, nl, nr, chunk);
  SPR_REQUIRE(ws != nullptr && ws_bytes >= spr_tn_product_split_workspace_bytes(), "tn_product_split: workspace too small");
  Workspace w(ws, ws_bytes);
  float* lp = w.take<float>(kAmaxParts);
  float* rp = w.take<float>(kAmaxParts);
  const float* lparts = l_range;
  const float* rparts = r_range;
  int nlp = l_range_n, nrp = r_range_n;
  if (l_range == nullptr) {
    if (int rc = launch_absmax(Lm, rows, nl, nl, lp, stream)) return rc;
    lparts = lp;
    nlp = kAmaxParts;
  }
  if (r_range == nullptr) {
    if (int rc = launch_absmax(Rm, rows, nr, nr, rp, stream)) return rc;
    rparts = rp;
    nrp = kAmaxParts;
  }
  SPR_REQUIRE(nlp >= 1 && nrp >= 1, "tn_product_split: a range needs a count");
  const long nbatch = (rows + chunk - 1) / chunk;
  SPR_REQUIRE(nbatch <= 65535, "tn_product_split: too many batches (%ld)", nbatch);
  const int tiles = cdiv(nl, spr::TNT) * cdiv(nr, spr::TNT);
  hipLaunchKernelGGL(spr::k_gemm_tn_h3, dim3(tiles, (unsigned)nbatch), dim3(256), 0, stream, Lm, Rm, rows, nl, nr, chunk,
                     lparts, nlp, rparts, nrp, parts);
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" int spr_bgemm(const float* A, const float* B, float* C, const void* desc_dev, int nbatch,
                         int max_m, int max_n, long sa_i, long sa_k, long sb_k, long sb_j, long sc_i,
                         long sc_j, float alpha, float beta, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(A && B && C && desc_dev && nbatch >= 1 && max_m >= 1 && max_n >= 1, "bgemm: bad arguments");
  SPR_REQUIRE(nbatch <= 65535, "bgemm: too many batches (%d)", nbatch);
  // tile shape by the largest batch entry: big square outputs -> 128 x 128, tall d_head-wide ones
  // -> 128 x 32, everything small stays on the 64 x 64 kernel (all three accumulate identically)
  static const int force = getenv("SPR_BGEMM_TILE") ? atoi(getenv("SPR_BGEMM_TILE")) : 0;   // A/B switch: 64 = old kernel
  if (force != 64 && max_m >= 128 && max_n >= 96) {
    const long tiles = (long)cdiv(max_m, 128) * cdiv(max_n, 128);
    SPR_REQUIRE(tiles < (1l << 31), "bgemm: grid too large");
    if (tiles * nbatch < 160) {
      // fewer tiles than compute units (the per-pair products of the loss heads: 38 tiles): half-height tiles
      const long tiles64 = (long)cdiv(max_m, 64) * cdiv(max_n, 128);
      hipLaunchKernelGGL((k_bgemm_f32_t<64, 128>), dim3((unsigned)tiles64, nbatch), dim3(256), 0, stream, A, B, C,
                         (const BgemmDesc*)desc_dev, sa_i, sa_k, sb_k, sb_j, sc_i, sc_j, alpha, beta);
      SPR_LAUNCH_CHECK();
      return 0;
    }
    hipLaunchKernelGGL((k_bgemm_f32_t<128, 128>), dim3((unsigned)tiles, nbatch), dim3(256), 0, stream, A, B, C,
                       (const BgemmDesc*)desc_dev, sa_i, sa_k, sb_k, sb_j, sc_i, sc_j, alpha, beta);
    SPR_LAUNCH_CHECK();
    return 0;
  }
  if (force != 64 && max_m >= 128 && max_n <= 32) {
    const long tiles = (long)cdiv(max_m, 128) * cdiv(max_n, 32);
    SPR_REQUIRE(tiles < (1l << 31), "bgemm: grid too large");
    hipLaunchKernelGGL((k_bgemm_f32_t<128, 32>), dim3((unsigned)tiles, nbatch), dim3(256), 0, stream, A, B, C,
                       (const BgemmDesc*)desc_dev, sa_i, sa_k, sb_k, sb_j, sc_i, sc_j, alpha, beta);
    SPR_LAUNCH_CHECK();
    return 0;
  }
  const long tiles = (long)cdiv(max_m, TB) * cdiv(max_n, TB);
  SPR_REQUIRE(tiles < (1l << 31), "bgemm: grid too large");
  hipLaunchKernelGGL(k_bgemm_f32, dim3((unsigned)tiles, nbatch), dim3(256), 0, stream, A, B, C,
                     (const BgemmDesc*)desc_dev, sa_i, sa_k, sb_k, sb_j, sc_i, sc_j, alpha, beta, cdiv(max_n, TB));
  SPR_LAUNCH_CHECK();
  return 0;
}

extern "C" int spr_reduce_parts(const float* parts, int nparts, long n, float scale, float* out,
                                int accumulate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  SPR_REQUIRE(parts && out && nparts >= 1 && n >= 1, "reduce_parts: bad arguments");
  hipLaunchKernelGGL(k_reduce_parts, dim3(cdiv(n, 64)), dim3(256), 0, stream, parts, nparts, n, scale, out,
                     accumulate);
  SPR_LAUNCH_CHECK();
  return 0;
}
